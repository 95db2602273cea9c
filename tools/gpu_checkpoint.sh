#!/bin/bash
# One GPU-box call: parity tests, smoke, bench, kernel trace of the bench command.  Outputs under gpurun_out/ckpt/.
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/ckpt
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 && tail -3 $O/pytest.log &&
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log &&
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err && cat $O/bench.json &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --serial-passes > $O/kt.log 2>&1 && tail -1 $O/kt.log
