#!/bin/bash
# Round-2 evidence: full bench line, kernel trace of the bench command (serial passes), HBM traffic counters (separate --pmc
# passes, no tracing domains).  Outputs under gpurun_out/r2p/ -- the summaries are copied to profiles/ by hand.
O=$GRAFT_REPO_ROOT/gpurun_out/r2p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --serial-passes > $O/kt.log 2>&1; tail -1 $O/kt.log
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/fetch.log 2>&1; tail -1 $O/fetch.log
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/write.log 2>&1; tail -1 $O/write.log
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/kt $O/kstats.csv && head -8 $O/kstats.csv
python tools/pmc_traffic.py $O/fetch $O/write 2048 1 $O/traffic.json | head -12
