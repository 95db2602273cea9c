#!/bin/bash
# round 4: instruction cache and scalar data cache of the band kernels (k4_in 23.6 KB + k4_out 40.9 KB of code run side by side on two
# streams; the instruction cache holds 64 KB for two CUs)
O=$GRAFT_REPO_ROOT/gpurun_out/r4ae
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" -d $O/$n -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 2 4 > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/$n.log; }; }
run p1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run p2 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run p3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run p4 SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES
run p5 SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES
cd $GRAFT_REPO_ROOT
python3 tools/pmc_table.py $O p1 p2 p3 p4 p5 > $O/table.txt 2>&1
grep -A24 "== k4_out<0\|== k4_in<true, false" $O/table.txt
for p in p1 p2 p3 p4 p5; do rm -rf $O/$p; done
