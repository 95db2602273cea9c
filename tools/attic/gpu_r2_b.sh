#!/bin/bash
# Round 2, quick iteration: parity tests, short bench (no CPU baseline), phase profile.  Outputs under gpurun_out/r2b/.
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2b/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "second", d["second_point"]["value"])
s=d.get("secondary"); print("scan", s and (s["value"], s["load_s"], s["scan_s"]))
print("minibatch", d.get("secondary_default_mode",{}).get("value"))
PY
tail -3 $O/bench.err
timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases.log 2>&1; cat $O/phases.log
