#!/bin/bash
# round 4: long sequences: the per-sequence filter kernels with one workgroup per CU (LDS image up to 150 KB) against the launches per diagonal
O=$GRAFT_REPO_ROOT/gpurun_out/r4ao; mkdir -p $O
cd $GRAFT_REPO_ROOT
for L in 700 1000 1400; do
for kb in 80 150; do
  ELEMDP_BPP_SEQ_KB=$kb ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 2000 $L > $O/l.txt 2>&1 || { tail -5 $O/l.txt; exit 1; }
  echo "L=$L cap $kb KB: $(grep 'BPP filter' $O/l.txt | tail -1) ; $(grep '== load' $O/l.txt | tail -1)"
done
done
