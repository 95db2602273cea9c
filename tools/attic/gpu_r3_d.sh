#!/bin/bash
# round 3, step d: dynamic instruction counts per phase (phase knock-outs through option dbg: 1 pair phases, 2 item sums, 4 unary)
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 7; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY -d $O/d$dbg -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 0 dbg=$dbg > $O/d$dbg.log 2>&1 || { echo "pass $dbg failed"; tail -3 $O/d$dbg.log; }
  echo "#### dbg $dbg"; python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $O d$dbg | head -24
  rm -rf $O/d$dbg
done
