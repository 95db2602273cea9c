#!/bin/bash
# round 4: where the vector L1's line accesses of the band kernels come from (the L1 looks up ~0.4 lines per clock and CU for
# gathers: tools/bench_micro/gather_bw.hip; k4_out averages 0.42): TCP_TOTAL_CACHE_ACCESSES with phases switched off
O=$GRAFT_REPO_ROOT/gpurun_out/r4ai; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 7 2048 1024; do
  timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum -d $O/p$dbg -o run -- python3 $GRAFT_REPO_ROOT/tools/pmc_dbg.py 2048 200 $dbg > $O/p$dbg.log 2>&1 || { echo "dbg $dbg failed"; tail -3 $O/p$dbg.log; }
  ( cd $GRAFT_REPO_ROOT && PMC_TABLE_TOP=6 python3 tools/pmc_table.py $O p$dbg > $O/t$dbg.txt 2>&1 )
  echo "---- dbg $dbg: $(grep "^dbg" $O/p$dbg.log)"; grep -A5 "== k4_out<0\|== k4_in<true, false" $O/t$dbg.txt
  rm -rf $O/p$dbg
done
