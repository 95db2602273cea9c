#!/bin/bash
# round 4: kernel stats of one load (10 000 x L=300)
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/load_laps.py 10000 300 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/kt $O/kernel_stats.csv && head -24 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/kt
