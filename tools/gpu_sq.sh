#!/bin/bash
# SQ counters of one train evaluation: is the pipeline waiting (memory) or issuing (VALU)?  Outputs under gpurun_out/sq/.
O=$GRAFT_REPO_ROOT/gpurun_out/sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d $O/run -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/sq_report.py $O/run
