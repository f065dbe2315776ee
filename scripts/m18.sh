cd $GRAFT_REPO_ROOT
python - <<'PY'
import mpilattice_boltzmann_amd as lbm
lbm.write_synthetic_deck("/tmp/deck8192", "8192x8192", lbm.Params(8192, 8192, 200, 10, 0.1, 0.005, 1.85), 0.005, 42, True)
PY
mkdir -p /tmp/run8192 && cd /tmp/run8192
time $GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk /tmp/deck8192/input_8192x8192.params /tmp/deck8192/obstacles_8192x8192.dat
ls -la /tmp/run8192; head -2 /tmp/run8192/final_state.dat; tail -1 /tmp/run8192/av_vels.dat
