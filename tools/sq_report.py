"""Per-kernel sums of the SQ counters of a rocprofv3 --pmc run (rocpd sqlite): python tools/sq_report.py <dir>"""
import glob, os, sqlite3, sys
from collections import defaultdict
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
acc = defaultdict(lambda: defaultdict(float))
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][cn] += val
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    a = acc[k]
    wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
    print("%-24s wave_cycles %.3g  wait_any %.1f%%  wait_inst %.1f%%  active_inst %.1f%%  active_valu %.1f%%  insts_valu %.3g  insts_lds %.3g" % (
        k, wc, 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * a.get("SQ_ACTIVE_INST_VALU", 0) / wc, a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_LDS", 0)))
