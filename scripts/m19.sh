cd $GRAFT_REPO_ROOT
ls -la oracle/_ref/ 2>&1 | head; ldd oracle/_ref/d2q9-bgk_ref 2>&1 | grep -E "mpi|not found"
mkdir -p /tmp/refrun && cd /tmp/refrun && time $GRAFT_REPO_ROOT/oracle/_ref/d2q9-bgk_ref $GRAFT_REPO_ROOT/tests/golden/decks/input_128x128.params $GRAFT_REPO_ROOT/tests/golden/decks/obstacles_128x128.dat; sha256sum final_state.dat
