#!/bin/bash
# kernel stats of load_batch (three loads of 10 000 x L=300)
O=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/load_laps.py 10000 300 > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/kt $O/kstats_load.csv; cut -d, -f1-6 $O/kstats_load.csv | head -24
