#!/bin/bash
# round 4: unpadded compact tables as the default: full GPU suite, bench
O=$GRAFT_REPO_ROOT/gpurun_out/r4ak; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 1000 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-330 $O/bench.json
python - <<'PY'
import json,os
b=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4ak/bench.json")))
s=b["secondary"]; print("scan", round(s["value"]), s["load_s"], s["scan_s"], round(s["resident_rate"]))
print("default", b["secondary_default_mode"]["value"], "streamed", round(b["secondary_streamed"]["value"]), b["secondary_streamed"]["of_resident"], "scanE", round(b["secondary_scan_streamed"]["value"]))
print(b["per_gpu_at_shard"]["train_D"], b["per_gpu_at_shard"]["scan_E"])
PY
