#!/bin/bash
# round 3, step h: kernel trace of the scan (10 000 x L=300, (.....)) and the new scan test
O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_round3_gpu.py -q -m gpu -x -k "scan_passes" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
S="python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 10000 300 (.....)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/skt -o run -- $S 0 0 1 > $O/skt.log 2>&1; tail -1 $O/skt.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/skt $O/kstats_scan.csv && head -24 $O/kstats_scan.csv
rm -rf $O/skt
