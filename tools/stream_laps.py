"""Sums the ELEMDP_TIME laps of tools/stream_60k.py per evaluation and label.  usage: ELEMDP_TIME=1 STREAM_REPS=3 python tools/stream_60k.py ... 2>&1 | python tools/stream_laps.py"""
import re, sys
from collections import defaultdict
ev, acc = 0, defaultdict(float)
for line in sys.stdin:
    m = re.match(r"\[elemdp\s+([\d.]+) ms\] (.*)", line)
    if m:
        acc[m.group(2).strip()] += float(m.group(1))
    elif line.startswith("eval"):
        print(line.strip())
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
            print("    %8.0f ms  %s" % (v, k))
        acc.clear()
