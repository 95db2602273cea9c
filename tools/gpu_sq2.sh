#!/bin/bash
# LDS counters of one train evaluation: are the LDS atomics of the heavy sums serialising?
O=$GRAFT_REPO_ROOT/gpurun_out/sq2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL -d $O/run -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<'PY'
import glob, os, sqlite3
from collections import defaultdict
db = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/sq2/run", "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
acc = defaultdict(lambda: defaultdict(float))
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][cn] += val
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:4]:
    print(k, {cn: "%.3g" % v for cn, v in sorted(acc[k].items())})
PY
