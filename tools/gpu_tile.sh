#!/bin/bash
# Kernel stats and FETCH_SIZE of the train pipeline with / without the tiled split sums (option "tile").
O=$GRAFT_REPO_ROOT/gpurun_out/tile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/tools/run_eval.py
for T in 0 1; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt$T -o run -- python3 $R 4096 200 2 4 0 tile=$T > $O/kt$T.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/kt$T $O/kstats_tile$T.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch$T -o run -- python3 $R 2048 200 1 4 0 tile=$T > $O/fetch$T.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write$T -o run -- python3 $R 2048 200 1 4 0 tile=$T > $O/write$T.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $O/fetch$T $O/write$T 2048 1 $O/traffic_tile$T.json
done
tail -2 $O/kt0.log $O/kt1.log
