"""Time of a rocprofv3 kernel trace (rocpd sqlite) during which only kernels of the given name prefixes run (exposed chains):
python tools/kexposed.py <dir> prefix [prefix ...]"""
import glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
small = tuple(sys.argv[2:])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
def fam(n):
    return n.replace("elemdp::(anonymous namespace)::", "").replace("void ", "")
ev = []
for s, e, n in rows:
    k = 0 if fam(n).startswith(small) else 1
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
cnt = [0, 0]
last = ev[0][0]
only_small = none = both = big_only = 0
for t, d, k in ev:
    dt = t - last
    if cnt[0] > 0 and cnt[1] == 0: only_small += dt
    elif cnt[0] == 0 and cnt[1] == 0: none += dt
    elif cnt[0] > 0: both += dt
    else: big_only += dt
    cnt[k] += d
    last = t
span = ev[-1][0] - ev[0][0]
print("span %.1f ms: only %s running %.1f ms, together with others %.1f ms, others alone %.1f ms, nothing %.1f ms" % (
    span / 1e6, "/".join(small), only_small / 1e6, both / 1e6, big_only / 1e6, none / 1e6))
