#!/bin/bash
# round 4: phase clocks of the per-sequence filter kernels (thread 0 of every workgroup)
O=$GRAFT_REPO_ROOT/gpurun_out/r4aa; mkdir -p $O
cd $GRAFT_REPO_ROOT
ELEMDP_BPP_PROF=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/prof.txt 2>&1 || { tail -5 $O/prof.txt; exit 1; }
grep "bpp prof" $O/prof.txt | tail -16; grep "== load" $O/prof.txt | tail -1
