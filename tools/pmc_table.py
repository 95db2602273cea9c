"""Per-kernel sums of every counter of several rocprofv3 --pmc passes (rocpd sqlite): python tools/pmc_table.py <dir> <pass>..."""
import glob, os, sqlite3, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for p in sys.argv[2:]:
    dbs = glob.glob(os.path.join(sys.argv[1], p, "**", "*.db"), recursive=True)
    if not dbs:
        print("pass %s: no output" % p)
        continue
    c = sqlite3.connect(dbs[0])
    for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        k = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][cn] += val
names = sorted({cn for k in acc for cn in acc[k]})
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", max(acc[k].values()))))[:int(os.environ.get("PMC_TABLE_TOP", "4"))]:
    print("==", k)
    for cn in names:
        if cn in acc[k]:
            print("   %-36s %.4g" % (cn, acc[k][cn]))
