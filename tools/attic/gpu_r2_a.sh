#!/bin/bash
# Round 2, call A: parity tests (all failures shown), smoke, bench, phase profile, kernel trace of the bench command.
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cat $O/bench.json; tail -3 $O/bench.err
timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases.log 2>&1; cat $O/phases.log
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --serial-passes > $O/kt.log 2>&1; tail -1 $O/kt.log
cd $GRAFT_REPO_ROOT && python tools/kstats.py $O/kt $O/kstats.csv && head -12 $O/kstats.csv
