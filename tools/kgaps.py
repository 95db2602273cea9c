"""Durations of, and gaps between, consecutive kernels of a rocprofv3 kernel trace (rocpd sqlite): python tools/kgaps.py <dir>"""
import glob, os, sqlite3, sys
from collections import defaultdict
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
def short(n): return n.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
dur, gap = defaultdict(list), defaultdict(list)
prev_end = None
for name, s, e in rows:
    k = short(name)
    dur[k].append(e - s)
    if prev_end is not None:
        gap[k].append(s - prev_end)
    prev_end = e
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap[k] or [0]
    print("%-28s n %5d  dur avg %9.1f us (min %8.1f max %8.1f)  gap-before avg %7.1f us (min %6.1f)" %
          (k, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(g) / len(g) / 1e3, min(g) / 1e3))
