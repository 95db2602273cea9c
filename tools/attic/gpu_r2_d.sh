#!/bin/bash
# quick parity subset + ceilings + kernel stats.  Outputs under gpurun_out/r2d/.
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_round2_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout -k 10 300 python tools/dbg_phases.py 4096 200 > $O/ceil.log 2>&1; cat $O/ceil.log
timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases.log 2>&1; cat $O/phases.log
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 3 4 0 two_streams=0 group_streams=1 > $O/kt.log 2>&1; tail -2 $O/kt.log
cd $GRAFT_REPO_ROOT && python tools/kstats.py $O/kt $O/kstats.csv && head -14 $O/kstats.csv
