#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 3 4 0 group_streams=1 > $O/kt.log 2>&1; tail -2 $O/kt.log
cd $GRAFT_REPO_ROOT && python tools/kstats.py $O/kt $O/kstats.csv && head -8 $O/kstats.csv
