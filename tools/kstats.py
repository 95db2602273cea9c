"""Per-kernel totals from a rocprofv3 (rocpd sqlite) output directory: python tools/kstats.py <dir> [out.csv]"""
import csv, glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(scratch_size), max(lds_size) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
def short(n): return n.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
out = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "Scratch", "LDS"]]
for r in rows:
    out.append([short(r[0]), r[1], r[2], "%.1f" % r[3], "%.2f" % (100 * r[2] / tot), r[4], r[5], r[6], r[7], r[8]])
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
