#!/bin/bash
# round 4: full GPU suite + smoke of the final tree
O=$GRAFT_REPO_ROOT/gpurun_out/r4af; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -10 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
