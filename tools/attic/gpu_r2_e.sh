#!/bin/bash
# full parity suite + streamed 60k x L=200 evaluation + quick bench.  Outputs under gpurun_out/r2e/.
O=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
timeout -k 10 600 python tools/stream_60k.py 60000 200 > $O/stream60k.log 2>&1; echo "stream rc $?"; cat $O/stream60k.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2e/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "second", d["second_point"]["value"])
s=d.get("secondary"); print("scan", s and (s["value"], s["load_s"], s["scan_s"]))
print("minibatch", d.get("secondary_default_mode",{}).get("value"))
PY
