#!/bin/bash
# round 4: per-sequence filter kernels after the dense stems / unconditional unary loads: tests, laps, phase clocks
O=$GRAFT_REPO_ROOT/gpurun_out/r4y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "filter or bpp" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps.txt 2>&1 || { tail -5 $O/laps.txt; exit 1; }
echo "$(grep 'BPP filter' $O/laps.txt | tail -3 | awk '{s+=$2} END {print s}') ms filter; $(grep '== load' $O/laps.txt | tail -1)"
ELEMDP_BPP_PROF=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/prof.txt 2>&1 || { tail -5 $O/prof.txt; exit 1; }
grep "bpp prof" $O/prof.txt | tail -16
