#!/bin/bash
# round 3, step b: where the time and the bytes go with the compact tables (kernel trace, HBM counters, phase clocks)
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 3 4 > $O/kt.log 2>&1; tail -1 $O/kt.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/fetch.log 2>&1; tail -1 $O/fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/write.log 2>&1; tail -1 $O/write.log
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/kt $O/kstats.csv && head -12 $O/kstats.csv
python tools/pmc_traffic.py $O/fetch $O/write 2048 1 $O/traffic.json | head -30
timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases.txt 2>&1; cat $O/phases.txt
rm -rf $O/kt $O/fetch $O/write
