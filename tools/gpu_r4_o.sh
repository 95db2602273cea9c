#!/bin/bash
# round 4: address translation and L2 counters of the band kernels (which unit do all variants of the schedule share?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" $O/avail.txt | awk '{print $2}' | sort -u > $O/names.txt
grep -i "utcl\|tlb\|xnack\|translat" $O/names.txt | head -40
run() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" -d $O/$n -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/$n.log; }; }
run p1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
run p2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run p3 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run p4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE
run p5 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum
cd $GRAFT_REPO_ROOT
python3 tools/pmc_table.py $O p1 p2 p3 p4 p5 > $O/table.txt 2>&1
cat $O/table.txt
for p in p1 p2 p3 p4 p5; do rm -rf $O/$p; done
