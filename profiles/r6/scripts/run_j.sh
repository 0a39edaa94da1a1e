#!/bin/bash
# round 6, call J: the full GPU suite + smoke() on the final tree
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6j_smoke.log 2>&1; tail -1 gpurun_out/r6j_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6j_tests.log 2>&1; tail -4 gpurun_out/r6j_tests.log
