"""Condenses rocprofv3 output directories into the small files committed under profiles/.

  python tools/summarize_profile.py stats <dir> <out.csv>          kernel_stats.csv of a --kernel-trace --stats run
  python tools/summarize_profile.py pmc <out.json> <n_seq> <n_eval> <dir> [<dir> ...]
                                                                    per-kernel sums of every counter in the --pmc runs
"""
import csv, glob, json, os, sys
from collections import defaultdict


def find(d, suffix):
    r = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not r:
        raise SystemExit("no %s under %s" % (suffix, d))
    return r[0]


def short(name):
    name = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


if sys.argv[1] == "stats":
    src = find(sys.argv[2], "_kernel_stats.csv")
    rows = list(csv.reader(open(src)))
    with open(sys.argv[3], "w") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            r[0] = short(r[0])
            w.writerow(r)
    print("wrote", sys.argv[3])
else:
    out, n_seq, n_eval = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for d in sys.argv[5:]:
        for row in csv.DictReader(open(find(d, "_counter_collection.csv"))):
            k = short(row["Kernel_Name"])
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add((d, row["Dispatch_Id"]))
    res = {"n_seq": n_seq, "n_eval": n_eval, "kernels": {}}
    for k in sorted(tot):
        res["kernels"][k] = {"dispatches": len(calls[k]) // max(1, len(sys.argv[5:]) if False else 1), **tot[k]}
    train = [k for k in tot if k.startswith("k3_") and "bpp" not in k]
    res["train_pipeline_sum"] = {c: sum(tot[k].get(c, 0.) for k in train) for c in sorted({c for k in train for c in tot[k]})}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)
