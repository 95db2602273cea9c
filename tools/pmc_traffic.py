"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite), optionally with the kernel
times of a kernel-trace pass of the same command.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <n_seq> <out.json> [--kt <kernel_trace_dir>] [--seq-len L]
                                [--pattern P] [--S n_states] [--scan] [--merge <existing.json>] [--source "..."]
                                [--bench <bench line .json>]

The number of evaluations (scans) of a run is READ FROM THE RUN: k4_weights is launched once per evaluation / scan, k_mask once per
load_batch -- separately for each of the three passes (they may run different commands).  Kernels of the evaluation are divided by the
evaluations, load-time kernels (k_mask*, k6_*, k_plan_*, k_role_*, k_permute_items) by the loads.  With --bench the per-kernel
times of the train pipeline must add up to at most 1.05 x the ms_per_step of that bench line, or the tool fails: a wrong divisor
shows up there first (round 3 divided 7 evaluations by 6: 17 %).  (The kernel trace runs the groups on ONE stream so that kernel
times add up; the bench overlaps the tails of two streams and is 2 - 3 % faster than that sum.)

FETCH_SIZE / WRITE_SIZE are in KiB.  Following MI355X_MICROARCH.md (HBM section), on gfx950 FETCH_SIZE tallies 128-B read requests
at 64 B, so fetched bytes = 2 x FETCH_SIZE for wide coalesced reads (our row segments are 8 B per lane over contiguous rows:
calibrated to within that factor in profiles/r01_calib_fetch_report.txt); WRITE_SIZE is exact for streaming stores.  Both the
raw and the corrected figure are written.  Per kernel of the evaluation pipeline the file carries
    bytes              measured HBM bytes of one evaluation (2 * FETCH_SIZE + WRITE_SIZE, summed over its launches)
    algorithmic_bytes  its share of SURVEY section 8(d)'s per-unit figure (k4_in: T, k4_out: 4 T per sequence; the scan kernels:
                       their tables, see --scan) times the sequences of the run
    ms                 its time in one evaluation (kernel trace of the same command on one stream, when --kt is given)
    frac               algorithmic_bytes / ms against the 8 TB/s HBM peak
"""
import argparse, glob, json, os, sqlite3
from collections import defaultdict

PEAK = 8.0e12


def short(name):
    return name.replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def per_kernel(d, counter):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        k = short(name)
        out[k][0] += 1
        out[k][1] += val * 1024.0
    return out


def kernel_ms(d):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, n, tot in c.execute("select name, count(*), sum(end-start) from kernels group by name"):
        out[short(name)][0] += n
        out[short(name)][1] += tot / 1e6
    return out


def family(k):
    """kernel name without its template arguments"""
    return k.split("<")[0]


LOAD_FAMILIES = ("k_mask", "k6_", "k_plan_", "k_role_", "k_permute_items")


def counts(tab):
    """(evaluations, loads) of a run from its own dispatch counts"""
    n_eval = sum(v[0] for k, v in tab.items() if family(k) == "k4_weights")
    n_load = sum(v[0] for k, v in tab.items() if family(k) == "k_mask")
    if n_eval <= 0:
        raise SystemExit("no k4_weights dispatch in the run: cannot tell the number of evaluations")
    return n_eval, max(n_load, 1)


def divisor(k, n_eval, n_load):
    return n_load if family(k).startswith(LOAD_FAMILIES) else n_eval


ap = argparse.ArgumentParser()
ap.add_argument("fetch"); ap.add_argument("write"); ap.add_argument("n_seq", type=int); ap.add_argument("out")
ap.add_argument("--kt"); ap.add_argument("--bench")
ap.add_argument("--seq-len", type=int, default=200); ap.add_argument("--pattern", default="((.*.))"); ap.add_argument("--S", type=int, default=22)
ap.add_argument("--scan", action="store_true"); ap.add_argument("--merge"); ap.add_argument("--source", default="")
a = ap.parse_args()
fetch, write = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
kms = kernel_ms(a.kt) if a.kt else {}
(fe, fl), (we, wl) = counts(fetch), counts(write)
ke, kl = counts(kms) if kms else (0, 0)
L, W, S = a.seq_len, 50, a.S
T = (L + 1) * (W + 1) * 7 * S * 8 + (L + 1) * S * 8                      # one table of one sequence
T_trace, T_b = (L + 1) * (W + 1) * 7 * S * 20, (L + 1) * (W + 1) * 7 * 8
# train: 5 T = inside writes T, the outside sweep reads it and writes its own table for both passes of the reference (4 T);
# scan: 7 T + T_trace + 3 T_b = two sum insides (2 T), two sum outsides (4 T), the Viterbi pass (T + its trace), the filter (3 T_b)
# (per kernel INSTANCE: the scan runs two instances of k4_in and of k4_out, each one pass)
alg = {"k4_in": T, "k4_out": 4 * T} if not a.scan else {"k4_in": T, "k4_out": 2 * T, "k5_cyk": T + T_trace, "k6_in": T_b, "k6_out": 2 * T_b,
                                                           "k6_in_seq": T_b, "k6_out_seq": 2 * T_b, "k6_in_tab": T_b, "k6_out_tab": 2 * T_b}
pipeline = ("k4_", "k_reduce") if not a.scan else ("k4_", "k5_", "k6_")
rows, tot_f, tot_w, tot_ms = {}, 0.0, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    fb, wb = f[1] / divisor(k, fe, fl), w[1] / divisor(k, we, wl)         # bytes of one evaluation (one load)
    row = {"dispatches": f[0] or w[0], "per": "load" if family(k).startswith(LOAD_FAMILIES) else "evaluation",
           "fetch_size_bytes": fb, "write_size_bytes": wb, "bytes": 2 * fb + wb, "bytes_per_seq": (2 * fb + wb) / a.n_seq}
    if family(k) in alg:
        row["algorithmic_bytes"] = alg[family(k)] * a.n_seq
        row["traffic_over_algorithmic"] = row["bytes"] / row["algorithmic_bytes"]
    if k in kms:
        row["ms"] = kms[k][1] / divisor(k, ke, kl)
        if "algorithmic_bytes" in row:
            row["frac"] = row["algorithmic_bytes"] / (row["ms"] * 1e-3) / PEAK
        row["measured_GBps"] = row["bytes"] / (row["ms"] * 1e-3) / 1e9
    rows[k] = row
    if k.startswith(pipeline):
        tot_f += fb
        tot_w += wb
        tot_ms += row.get("ms", 0.0)
res = {"n_seq": a.n_seq, "evaluations": {"fetch_pass": fe, "write_pass": we, "kernel_trace": ke},
       "loads": {"fetch_pass": fl, "write_pass": wl, "kernel_trace": kl}, "seq_len": L, "pattern": a.pattern,
       "pipeline": {"fetch_size_bytes_per_seq": tot_f / a.n_seq, "write_size_bytes_per_seq": tot_w / a.n_seq, "ms": tot_ms},
       "hbm_bytes_per_seq": (2 * tot_f + tot_w) / a.n_seq,
       "hbm_bytes_per_seq_uncorrected": (tot_f + tot_w) / a.n_seq,
       "algorithmic_bytes_per_seq": (7 * T + T_trace + 3 * T_b) if a.scan else 5 * T,
       "source": a.source or "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction of "
                             "MI355X_MICROARCH.md); kernels %s" % ", ".join(p + "*" for p in pipeline),
       "kernels": rows}
if a.bench:
    line = [ln for ln in open(a.bench).read().splitlines() if ln.startswith("{")][-1]
    step = json.loads(line)["ms_per_step"]
    res["bench_ms_per_step"] = step
    if kms and tot_ms > 1.05 * step:
        raise SystemExit("per-kernel times of the pipeline add up to %.1f ms per evaluation, more than 1.05 x the bench line's %.1f ms per step: "
                         "wrong divisor or a different workload" % (tot_ms, step))
if a.merge and os.path.exists(a.merge):
    old = json.load(open(a.merge))
    old["scan" if a.scan else "train"] = res
    res = old
json.dump(res, open(a.out, "w"), indent=1)
show = res.get("scan" if a.scan else "train", res)
print(json.dumps({k: v for k, v in show.items() if k != "kernels"}, indent=1))
for k, v in show["kernels"].items():
    if "algorithmic_bytes" in v:
        print("%-30s %5d  %8.2f MB/seq measured, %6.2f algorithmic (x%.2f)%s" % (k, v["dispatches"], v["bytes_per_seq"] / 1e6, v["algorithmic_bytes"] / a.n_seq / 1e6,
              v["traffic_over_algorithmic"], ("  %.1f ms  frac %.3f" % (v["ms"], v["frac"])) if "frac" in v else ""))
