#!/bin/bash
# round 4: the three role orders of the plan in one count and one scatter pass (ranks from the count pass): GPU suite, load times
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "^E " $O/tests.log | head; exit $rc; fi
timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.txt 2>&1; tail -1 $O/scan.txt
timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan2.txt 2>&1; tail -1 $O/scan2.txt
timeout -k 10 200 python tools/run_eval.py 10000 200 2 > $O/eval.txt 2>&1; grep -h "load\|seq/s" $O/eval.txt | tail -3
timeout -k 10 200 python tools/minibatch_bench.py 2000 200 40 > $O/mb.txt 2>&1; tail -1 $O/mb.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 10000 300 "(.....)" > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py $O/kt $O/kstats_scan.csv && grep -h "k_role\|k_plan\|k_permute\|k6_" $O/kstats_scan.csv | cut -c1-110; rm -rf $O/kt
