#!/usr/bin/env python3
"""d2q9_bgk.py <paramfile> <obstaclefile> — the reference's command-line contract on 1..N GPUs.

Single GPU:   python d2q9_bgk.py input.params obstacles.dat          (same as bin/d2q9-bgk)
N GPUs:       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
                  --master-port P d2q9_bgk.py input.params obstacles.dat
where it plays the role of `mpirun -np N ./d2q9-bgk` (reference `mpi_submit:63`): rows are
partitioned by the reference's rule (`d2q9-bgk.c:834-862`), halos travel over RCCL, rank 0 prints
the five stdout lines (`:411-415`) and writes final_state.dat / av_vels.dat into the cwd.
"""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def die(message: str) -> None:
    """`die()` of the reference (`d2q9-bgk.c:1145-1151`): "Error at line %d of file %s:\\n%s\\n" on stderr, exit(EXIT_FAILURE)."""
    caller = sys._getframe(1)
    sys.stderr.write(f"Error at line {caller.f_lineno} of file {os.path.basename(__file__)}:\n{message}\n")
    sys.stderr.flush()
    sys.exit(1)


def main(argv) -> int:
    if len(argv) != 3:                                                     # :197-200, :1153-1157
        sys.stderr.write(f"Usage: {argv[0]} <paramfile> <obstaclefile>\n")
        return 1
    import mpilattice_boltzmann_amd as lbm
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "LBM_FORCE_DEVICE" in os.environ:          # testing aid: several ranks on one device (if the communicator allows it)
        local_rank = int(os.environ["LBM_FORCE_DEVICE"])
    # rank 0 parses both files (d2q9-bgk.c:772-803, 917-953) before anything touches the GPU; the parameters
    # travel to the other ranks as the reference's t_param does, the obstacle rows are scattered by Simulation (:966-970)
    params, obstacles, failure = None, None, None
    if rank == 0:
        try:
            params = lbm.read_params(argv[1])
            obstacles, _ = lbm.read_obstacles(argv[2], params.nx, params.ny)
        except lbm.LbmError as e:
            failure = str(e)
            if world == 1:
                die(failure)
    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("LBM_DIST_BACKEND", "nccl")         # "gloo": several ranks may share one GPU (testing aid)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        box = [params, failure]
        dist.broadcast_object_list(box, src=0)
        params, failure = box
    if failure is not None:
        if rank == 0:
            die(failure)
        return 1
    # p2p: direct peer-to-peer halo stores (default); rccl: RCCL send/recv; torch: torch.distributed P2P ops
    exchange = os.environ.get("LBM_EXCHANGE", "auto" if backend == "nccl" or world == 1 else "p2p")
    # the contract path forms every sum|u| term as the reference does (double precision, d2q9-bgk.c:667); LBM_FLAGS=0: the library's default
    flags = int(os.environ.get("LBM_FLAGS", lbm._capi.FLAG_EXACT_AVVELS))
    # LBM_RANK_GRID=PXxPY (PX * PY = the ranks): the tile (2-D) decomposition instead of the reference's row blocks (peer-to-peer loop)
    rank_grid = None
    if os.environ.get("LBM_RANK_GRID") == "auto":             # lbm_choose_rank_grid: row blocks unless the grid is much wider than tall
        rank_grid = lbm.choose_rank_grid(params, world, flags)
        if rank_grid is not None:
            exchange = "p2p"
    elif os.environ.get("LBM_RANK_GRID"):
        try:
            rank_grid = tuple(int(v) for v in os.environ["LBM_RANK_GRID"].lower().split("x"))
            assert len(rank_grid) == 2 and rank_grid[0] * rank_grid[1] == world
        except (ValueError, AssertionError):
            die("LBM_RANK_GRID: expected PXxPY with PX * PY = the number of ranks")
        exchange = "p2p"
    try:
        sim = lbm.Simulation(params, obstacles, device=local_rank, distributed=world > 1, exchange=exchange, flags=flags, rank_grid=rank_grid)
    except lbm.LbmError as e:
        die(str(e))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    tic = time.time()                                                      # :278-279
    av_vels = sim.run(params.max_iters)                                    # :315-396
    torch.cuda.synchronize()
    toc = time.time()                                                      # :397-398
    ru = resource.getrusage(resource.RUSAGE_SELF)
    obs = sim.gather_observables()                                         # rank 0 gets (u_x, u_y, u, pressure) of the whole grid
    if rank == 0:
        print("==done==")                                                  # :411-415
        print("Reynolds number:\t\t%.12E" % sim.reynolds(observables=obs))
        print("Elapsed time:\t\t\t%.6f (s)" % (toc - tic))
        print("Elapsed user CPU time:\t\t%.6f (s)" % ru.ru_utime)
        print("Elapsed system CPU time:\t%.6f (s)" % ru.ru_stime)
        mlups = params.nx * params.ny * params.max_iters / (toc - tic) / 1e6
        print("MLUPS:\t\t\t\t%.1f (%d GPU%s, %s loop%s)" % (mlups, world, "" if world == 1 else "s", sim.loop,
                                                            "" if rank_grid is None else ", %d x %d tiles" % rank_grid))
        if not os.environ.get("LBM_NO_OUTPUT"):                            # :419-421
            sim.write_values(av_vels, ".", observables=obs)
    sim.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
