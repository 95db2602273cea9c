#!/bin/bash
cd $GRAFT_REPO_ROOT
for gs in 1 2 3 4; do echo "== group_streams=$gs"; timeout -k 10 200 python tools/run_eval.py 10000 200 3 4 0 group_streams=$gs 2>&1 | tail -1; done
for g in 1250 2500 5000; do echo "== group=$g (2 streams)"; timeout -k 10 200 python tools/run_eval.py 10000 200 3 4 $g 2>&1 | tail -1; done
