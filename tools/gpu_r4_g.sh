#!/bin/bash
# round 4: phase ceilings (knock-outs) and in-kernel phase clocks of the final build, one block per workgroup and three
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
cd $GRAFT_REPO_ROOT
for nb in 1 3; do
  echo "== nblk $nb"
  ELEMDP_NBLK=$nb timeout -k 10 300 python tools/dbg_phases.py 4096 200 0 1 2 4 7 128 2055 1031 > $O/ceil_nblk$nb.txt 2>&1; cat $O/ceil_nblk$nb.txt
  ELEMDP_NBLK=$nb timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases_nblk$nb.txt 2>&1; cat $O/phases_nblk$nb.txt
done
