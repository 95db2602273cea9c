"""HBM traffic of the train pipeline from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <n_seq> <n_eval> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  Following MI355X_MICROARCH.md (HBM section), on gfx950 FETCH_SIZE tallies 128-B
read requests at 64 B, so fetched bytes = 2 x FETCH_SIZE for wide coalesced reads (our staged segment loads are 8 B per
lane over contiguous rows: calibrated only to within that factor); WRITE_SIZE is exact for streaming stores.
Both the raw and the corrected figure are written."""
import glob, json, os, sqlite3, sys
from collections import defaultdict


def per_kernel(d, counter):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        k = name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[k][0] += 1
        out[k][1] += val * 1024.0
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
n_seq, n_eval = int(sys.argv[3]), int(sys.argv[4])
rows, tot_f, tot_w = {}, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    rows[k] = {"dispatches": f[0] or w[0], "fetch_size_bytes": f[1], "write_size_bytes": w[1]}
    if k.startswith("k4_") or k == "k_reduce":
        tot_f += f[1]
        tot_w += w[1]
res = {"n_seq": n_seq, "n_eval": n_eval, "seq_len": 200, "pattern": "((.*.))",
       "train_pipeline": {"fetch_size_bytes_per_seq": tot_f / n_seq / n_eval, "write_size_bytes_per_seq": tot_w / n_seq / n_eval},
       "hbm_bytes_per_seq": (2 * tot_f + tot_w) / n_seq / n_eval,
       "hbm_bytes_per_seq_uncorrected": (tot_f + tot_w) / n_seq / n_eval,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/run_eval.py %d 200 %d 4; bytes = 2*FETCH_SIZE + WRITE_SIZE "
                 "(gfx950 correction of MI355X_MICROARCH.md), k4_* kernels + k_reduce" % (n_seq, n_eval),
       "kernels": rows}
json.dump(res, open(sys.argv[5], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for k, v in rows.items():
    print("%-28s %5d  fetch %8.2f GB  write %8.2f GB" % (k, v["dispatches"], v["fetch_size_bytes"] / 1e9, v["write_size_bytes"] / 1e9))
