#!/bin/bash
# round 3, step c: which unit binds the band kernels?  instruction mix, TA / TCP busy, LDS, waits (separate --pmc passes)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" $O/avail.txt | awk '{print $2}' | sort -u > $O/names.txt
wc -l $O/names.txt
run() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" -d $O/$n -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/$n.log; }; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
run p3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM
run p4 TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run p5 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
run p6 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run p7 GRBM_GUI_ACTIVE GRBM_COUNT TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
cd $GRAFT_REPO_ROOT
python3 tools/pmc_table.py $O p1 p2 p3 p4 p5 p6 p7 > $O/table.txt 2>&1
cat $O/table.txt
for p in p1 p2 p3 p4 p5 p6 p7; do rm -rf $O/$p; done
