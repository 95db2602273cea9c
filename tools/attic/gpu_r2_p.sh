#!/bin/bash
# small groups: does keeping a group's diagonal inside the 256 MB memory-side cache pay?
cd $GRAFT_REPO_ROOT
for g in 400 800 1600 5000; do for gs in 2 4; do echo "== group=$g streams=$gs"; timeout -k 10 200 python tools/run_eval.py 10000 200 3 4 $g group_streams=$gs 2>&1 | tail -1; done; done
