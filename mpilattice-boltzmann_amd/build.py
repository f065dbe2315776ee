"""In-tree build of the native half: liblbm_d2q9.so (HIP kernels + C ABI) and the d2q9-bgk CLI shim.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the
outputs (lib/, bin/) are git-ignored but travel to the GPU box with the tree.  gfx950 only.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "liblbm_d2q9.so")
LIB_RCCL = os.path.join(PKG, "lib", "liblbm_d2q9_rccl.so")
CLI = os.path.join(PKG, "bin", "d2q9-bgk")
LIB_EXPERIMENTS = os.path.join(PKG, "lib", "variants", "experiments.so")

# -ffp-contract=off: keep the reference's unfused float arithmetic (bit parity with gcc -std=c99).
COMMON = ["-O3", "-ffp-contract=off", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP path cannot be built (there is no CPU fallback)")
    return exe


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> dict[str, str]:
    """Compile the shared library and the CLI if missing or older than their sources; experiments=True: the experiment build as well."""
    hipcc = _hipcc()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    headers = [os.path.join(ROOT, "include", "lbm_d2q9.h"), os.path.join(ROOT, "include", "lbm_d2q9_p2p.h"),
               os.path.join(CSRC, "lbm_internal.h"), os.path.join(CSRC, "lbm_p2p_impl.h"), os.path.abspath(__file__)]
    headers += sorted(glob.glob(os.path.join(CSRC, "kernels", "*.h")))       # device code, included by lbm_kernels.hip
    lib_src = [os.path.join(CSRC, "lbm_kernels.hip"), os.path.join(CSRC, "lbm_host.cpp")]
    if force or _stale(LIB, lib_src + headers):
        cmd = [hipcc, "--offload-arch=gfx950", *COMMON, "-fPIC", "-shared", *lib_src, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    rccl_src = [os.path.join(CSRC, "lbm_rccl.cpp")]
    rccl_hdr = [os.path.join(ROOT, "include", "lbm_d2q9_rccl.h")]
    if force or _stale(LIB_RCCL, rccl_src + rccl_hdr + headers + [LIB]):
        # host-only translation unit: the RCCL step loop is a client of the core ABI
        cmd = [hipcc, *COMMON, "-fPIC", "-shared", *rccl_src, "-L", os.path.dirname(LIB), "-llbm_d2q9", "-lrccl",
               "-Wl,-rpath,$ORIGIN", "-o", LIB_RCCL]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    cli_src = [os.path.join(CSRC, "d2q9_bgk_main.c")]
    if force or _stale(CLI, cli_src + headers + [LIB]):
        # the host shim is plain C99 over the C ABI: the system C compiler, no HIP, no C++ runtime of its own
        cc = shutil.which("gcc") or shutil.which("cc")
        if cc is None:
            raise RuntimeError("no C compiler (gcc / cc) for the host shim")
        cmd = [cc, "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-pthread", "-I", os.path.join(ROOT, "include"), *cli_src,
               "-L", os.path.dirname(LIB), "-llbm_d2q9", "-Wl,-rpath,$ORIGIN/../lib", "-o", CLI]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    out = {"lib": LIB, "lib_rccl": LIB_RCCL, "cli": CLI}
    if experiments:
        # the same translation unit with -DLBM_EXPERIMENTS=1: lbm_sweep_kernel and the LDS-staged one-step kernel, which measured slower
        # than what ships and are kept, with their parity tests (tests/experiments_suite.py), outside liblbm_d2q9.so
        os.makedirs(os.path.dirname(LIB_EXPERIMENTS), exist_ok=True)
        if force or _stale(LIB_EXPERIMENTS, lib_src + headers):
            cmd = [hipcc, "--offload-arch=gfx950", *COMMON, "-DLBM_EXPERIMENTS=1", "-fPIC", "-shared", *lib_src, "-o", LIB_EXPERIMENTS]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        out["lib_experiments"] = LIB_EXPERIMENTS
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
