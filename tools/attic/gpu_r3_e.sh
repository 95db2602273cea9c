#!/bin/bash
# round 3, step e: per-kernel times of one stream (group_streams=1: durations do not overlap), phase clocks, counters
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 3 4 0 group_streams=1 > $O/kt.log 2>&1; tail -2 $O/kt.log
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/kt $O/kstats.csv && head -12 $O/kstats.csv
timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases.txt 2>&1; cat $O/phases.txt
rm -rf $O/kt
