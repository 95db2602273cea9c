#!/bin/bash
# round 4: which phase of the band kernels asks the memory side for lines (fabric read / write requests with phases switched off)
O=$GRAFT_REPO_ROOT/gpurun_out/r4al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3; do
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/p$dbg -o run -- python3 $GRAFT_REPO_ROOT/tools/pmc_dbg.py 2048 200 $dbg > $O/p$dbg.log 2>&1 || { echo "dbg $dbg failed"; tail -3 $O/p$dbg.log; }
  ( cd $GRAFT_REPO_ROOT && PMC_TABLE_TOP=3 python3 tools/pmc_table.py $O p$dbg > $O/t$dbg.txt 2>&1 )
  echo "---- dbg $dbg: $(grep "^dbg" $O/p$dbg.log)"; grep -A4 "== k4_out<0\|== k4_in<true, false" $O/t$dbg.txt
  rm -rf $O/p$dbg
done
