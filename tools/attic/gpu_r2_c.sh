#!/bin/bash
# ceilings with phases off + SQ instruction counters per variant.  Outputs under gpurun_out/r2c/.
O=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/dbg_phases.py 4096 200 > $O/ceil.log 2>&1; cat $O/ceil.log
cd /tmp && export TMPDIR=/tmp
for D in 0 7 3 4; do
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq$D -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 0 dbg=$D > $O/sq$D.log 2>&1 || { tail -5 $O/sq$D.log; exit 1; }
echo "== dbg=$D"; python3 $GRAFT_REPO_ROOT/tools/sq_report.py $O/sq$D | head -4
done
for D in 0 7; do
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/sr$D -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 0 dbg=$D > $O/sr$D.log 2>&1 || { tail -5 $O/sr$D.log; exit 1; }
echo "== dbg=$D (2)"; python3 - <<PY
import glob, os, sqlite3
from collections import defaultdict
db = glob.glob(os.path.join("$O/sr$D", "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
acc = defaultdict(lambda: defaultdict(float))
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][cn] += val
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0))[:4]:
    print(k, {a: "%.3g" % b for a, b in acc[k].items()})
PY
done
