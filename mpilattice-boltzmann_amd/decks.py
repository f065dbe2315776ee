"""Input decks: the reference's two text formats and the synthetic benchmark deck.

Formats (reference `d2q9-bgk.c`):
  * parameter file — seven tokens nx, ny, maxIters, reynolds_dim (ints), density, accel, omega
    (floats), in that order (`d2q9-bgk.c:781-800`);
  * obstacle file — one `x y 1` line per blocked cell, duplicates allowed (`d2q9-bgk.c:933-949`).

The synthetic deck is SURVEY.md §8(d)'s: solid walls on the four edges plus interior cells blocked
i.i.d. with probability p from splitmix64 (fixed integer PRNG, so every implementation reads the
same file).  Pure numpy host code; nothing here touches the GPU.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


@dataclass(frozen=True)
class Params:
    """The reference's `t_param` inputs (`d2q9-bgk.c:79-90`) as read from a parameter file."""

    nx: int
    ny: int
    max_iters: int
    reynolds_dim: int
    density: float
    accel: float
    omega: float

    def write(self, path: str) -> None:
        # repr(float) round-trips; the reference reads with %f into a float (`d2q9-bgk.c:793-800`)
        with open(path, "w") as fh:
            fh.write(f"{self.nx}\n{self.ny}\n{self.max_iters}\n{self.reynolds_dim}\n")
            fh.write(f"{self.density!r}\n{self.accel!r}\n{self.omega!r}\n")


def splitmix64(seed: int, n: int, first: int = 0) -> np.ndarray:
    """Outputs first+1 .. first+n of splitmix64 seeded with `seed` (vectorised; uint64 wrap-around)."""
    with np.errstate(over="ignore"):
        idx = np.arange(first + 1, first + n + 1, dtype=np.uint64)
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def synthetic_obstacles(nx: int, ny: int, p: float = 0.005, seed: int = 42, walls: bool = True) -> np.ndarray:
    """(ny, nx) int32 map: 1 = blocked.  Cell (x, y) consumes the (y*nx+x)-th PRNG output; it is
    blocked when the top 53 bits, as a fraction of 2**53, fall below p."""
    # in cache-sized pieces: the 8192 x 8192 deck is 67 M outputs, and whole-array temporaries made this 9 s of memory traffic
    n = nx * ny
    flat = np.empty(n, dtype=np.int32)
    piece = 1 << 18
    for first in range(0, n, piece):
        r = splitmix64(seed, min(piece, n - first), first)
        frac = (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        flat[first:first + r.size] = frac < p
    obst = flat.reshape(ny, nx)
    if walls:
        obst[0, :] = 1
        obst[-1, :] = 1
        obst[:, 0] = 1
        obst[:, -1] = 1
    return obst


def write_obstacles(path: str, obstacles: np.ndarray) -> None:
    """Write the reference's `x y 1` list, row-major (y outer, x inner)."""
    ys, xs = np.nonzero(obstacles)
    out = np.empty((xs.size, 3), dtype=np.int64)
    out[:, 0], out[:, 1], out[:, 2] = xs, ys, 1
    np.savetxt(path, out, fmt="%d")


def write_synthetic_deck(directory: str, name: str, params: Params, p: float = 0.005, seed: int = 42,
                         walls: bool = True) -> tuple[str, str]:
    """Materialise `<directory>/input_<name>.params` and `obstacles_<name>.dat`; returns both paths."""
    os.makedirs(directory, exist_ok=True)
    ppath = os.path.join(directory, f"input_{name}.params")
    opath = os.path.join(directory, f"obstacles_{name}.dat")
    params.write(ppath)
    write_obstacles(opath, synthetic_obstacles(params.nx, params.ny, p, seed, walls))
    return ppath, opath
