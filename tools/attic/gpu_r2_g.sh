#!/bin/bash
# scan profile: kernel trace of load + 2 scans of 4096 x L=300 '(.....)'; load laps of three batches
O=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $O
cd $GRAFT_REPO_ROOT
ELEMDP_TIME=1 timeout -k 10 300 python tools/load_laps.py 10000 300 2>&1 | grep "==\|unfiltered" | head -30
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 4096 300 > $O/kt.log 2>&1; tail -3 $O/kt.log
cd $GRAFT_REPO_ROOT && python tools/kstats.py $O/kt $O/kstats.csv && head -24 $O/kstats.csv
