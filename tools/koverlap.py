"""How much kernels overlap in a rocprofv3 kernel trace (rocpd sqlite): sum of durations vs the union of their intervals.
python tools/koverlap.py <dir>"""
import glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select start, end from kernels order by start").fetchall()
tot = sum(e - s for s, e in rows)
union, cur_s, cur_e = 0, None, None
for s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print("kernels %d  sum of durations %.1f ms  union %.1f ms  (overlap factor %.2f)  first-to-last %.1f ms  GPU idle %.1f %%" % (
    len(rows), tot / 1e6, union / 1e6, tot / union, span / 1e6, 100 * (1 - union / span)))
