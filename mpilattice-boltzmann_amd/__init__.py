"""MI355X-native D2Q9-BGK lattice-Boltzmann timestep path (drop-in for the hot path of
ag14774/MPILattice-Boltzmann): hand-written gfx950 HIP kernels behind a C ABI
(include/lbm_d2q9.h), a CLI shim with the reference's contract, and this Python host mirror.

The directory name contains a hyphen; import it as `mpilattice_boltzmann_amd` (alias module at the
repo root) or via importlib.
"""
import os as _os

# The peer-to-peer transport maps other rank processes' device memory (hipIpcGetMemHandle / hipIpcOpenMemHandle) and
# RCCL does the same; where the host driver only supports dmabuf IPC both need this set before the HIP runtime
# initialises (harmless elsewhere).  Respect a value the caller has chosen.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import _capi, checker, decks  # noqa: E402
from ._capi import EXPORTS, LIB_PATH, LIB_RCCL_PATH, P2P_EXPORTS, RCCL_EXPORTS, LbmError, load_library, load_rccl_library
from .build import CLI as CLI_PATH
from .build import build
from .decks import Params, synthetic_obstacles, write_obstacles, write_synthetic_deck
from .host import (NORTH, SOUTH, HaloExchange, P2PRing, Partition, RcclRing, Simulation, av_velocity_host, av_velocity_obs,
                   count_free_cells, decompose, obstacle_window, plan_groups, plan_steps, rank_layout, tile_layout, choose_rank_grid, read_obstacles, read_params, reynolds,
                   run_partitioned, write_av_vels, write_final_state, write_final_state_obs)

__all__ = [
    "EXPORTS", "RCCL_EXPORTS", "P2P_EXPORTS", "P2PRing", "av_velocity_obs", "obstacle_window", "plan_groups", "plan_steps", "rank_layout", "tile_layout", "choose_rank_grid", "write_final_state_obs", "LIB_PATH", "LIB_RCCL_PATH", "load_rccl_library", "CLI_PATH", "LbmError", "load_library", "build", "Params", "synthetic_obstacles",
    "write_obstacles", "write_synthetic_deck", "NORTH", "SOUTH", "HaloExchange", "Partition", "RcclRing", "Simulation",
    "av_velocity_host", "count_free_cells", "decompose", "read_obstacles", "read_params", "reynolds",
    "run_partitioned", "write_av_vels", "write_final_state", "checker", "decks",
]
