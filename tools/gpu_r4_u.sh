#!/bin/bash
# round 4: kernel stats of the streamed train evaluation (30 000 x L=200 in chunks of 10 000, 2 evaluations)
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/stream_60k.py 30000 200 10000 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep "^eval\|^load" $O/kt.log
python tools/kstats.py $O/kt $O/kernel_stats.csv && head -30 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/kt
