#!/bin/bash
# Round-3 session E: the full record — parity suite, PMC round, secondary figures.
cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh r03 2>&1 | tail -25 && bash scripts/gpu_extras.sh r03 2>&1 | tail -40
