#!/bin/bash
# mini-batch iteration, load laps of a mini-batch, kernel stats of the scan at 10k x L=300
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $O
timeout -k 10 200 python tools/minibatch_bench.py 2000 200 60 2>&1 | tail -2
timeout -k 10 200 python tools/minibatch_bench.py 2000 200 60 --joint 2>&1 | tail -2
ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 128 200 "((.*.))" 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 10000 300 > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/kt $O/kstats_scan.csv; tail -1 $O/kt.log; cut -d, -f1-6 $O/kstats_scan.csv | head -30
