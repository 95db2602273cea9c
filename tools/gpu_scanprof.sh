#!/bin/bash
# kernel stats of the scan path at the config-E shape (2000 x L=300, S=29) and the config-B shape
O=$GRAFT_REPO_ROOT/gpurun_out/scanprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/ktE -o run -- python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 2000 300 '(.....)' > $O/ktE.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/ktE $O/kstats_scanE.csv; tail -1 $O/ktE.log; cut -d, -f1-6 $O/kstats_scanE.csv | head -24
