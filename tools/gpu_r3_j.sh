#!/bin/bash
# round 3, final: phase ceilings (knock-outs), in-kernel phase clocks and the SQ / TA / TCP counters of the final build
O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/dbg_phases.py 4096 200 0 1 2 4 16 7 128 32 > $O/ceilings.txt 2>&1; cat $O/ceilings.txt
timeout -k 10 300 python tools/prof_phases.py 4096 200 > $O/phases.txt 2>&1; cat $O/phases.txt
bash tools/gpu_r3_c.sh > $O/c.log 2>&1; tail -3 $O/c.log
cp gpurun_out/r3c/table.txt $O/counters.txt
