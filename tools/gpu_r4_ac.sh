#!/bin/bash
# round 4: tests of the plan builder's forms
O=$GRAFT_REPO_ROOT/gpurun_out/r4ac; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "random_lengths or candidate_table or filter or bpp" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
