set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
BIN=$GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk
D=$GRAFT_REPO_ROOT/tests/golden/decks
mkdir -p /tmp/r128 && cd /tmp/r128 && $BIN $D/input_128x128.params $D/obstacles_128x128.dat && sha256sum final_state.dat av_vels.dat
mkdir -p /tmp/r256 && cd /tmp/r256 && $BIN $D/input_256x256.params $D/obstacles_256x256.dat && sha256sum final_state.dat av_vels.dat
mkdir -p /tmp/r1024 && cd /tmp/r1024 && $BIN $D/input_1024x1024.params $D/obstacles_1024x1024.dat && sha256sum final_state.dat av_vels.dat
cd $GRAFT_REPO_ROOT && python - <<'PY'
import mpilattice_boltzmann_amd as lbm, time, numpy as np
p = lbm.Params(8192, 8192, 100, 10, 0.1, 0.005, 1.85)
obst = lbm.synthetic_obstacles(8192, 8192, 0.005, 42, True)
for flags in (0, 2):
    sim = lbm.Simulation(p, obst, flags=flags)
    sim.run(10)
    t=time.time(); sim.run(100); dt=time.time()-t
    print("8192^2 flags",flags, "ms/step", dt*10, "MLUPS", 8192*8192*100/dt/1e6, flush=True)
    sim.close()
PY
