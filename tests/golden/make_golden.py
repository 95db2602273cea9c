#!/usr/bin/env python3
"""Generate tests/golden/* from the compiled reference (oracle/_ref).  Runs only where
/root/reference exists (build container); the GPU box uses the committed fixtures.

    make -C oracle ref && python tests/golden/make_golden.py

Fixtures are data only: copies of the reference's own test *data* files (0.fq, 1.fq, {0..3}.model,
the `ubox` lines of 1.0.ps, material/positive.fa) and outputs of the reference run on stated inputs.
"""
import hashlib
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from rnaelem_amd import synth  # noqa: E402

REF = "/root/reference"
RB = os.path.join(REPO, "oracle", "_ref")
G = os.path.join(REPO, "tests", "golden")


def run(args, **kw):
    r = subprocess.run(args, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (args, r.stderr[-2000:]))
    return r.stdout


def jload(txt):
    return json.loads(txt.replace('"-inf"', "-Infinity").replace('"inf"', "Infinity").replace('"nan"', "NaN"))


def dump(name, obj):
    with open(os.path.join(G, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", name, os.path.getsize(os.path.join(G, name)))


def write_model(path, pattern, rows, lam, softmax=False, max_span=50, max_iloop=30, min_bpp=1e-4, tau=0.1,
                no_rss=0, no_prf=0, no_ene=0):
    """Model text in the reference's format (motif_io.hpp:29-57)."""
    def tab(rows):
        return "[" + ",".join("[" + ",".join("%.17g" % v for v in r) + "]" for r in rows) + "]"
    with open(path, "w") as f:
        f.write("pattern: %s\n" % pattern)
        f.write("%s: %s\n" % ("s" if softmax else "theta", tab(rows)))
        f.write("ene-param: ~T2004~\nmax-span: %d\nmax-internal-loop: %d\n" % (max_span, max_iloop))
        f.write("theta-softmax: %d\n" % int(softmax))
        f.write("%s: 0.1\nrho-lambda: 0.1\ntau: %g\n" % ("rho-s" if softmax else "rho-theta", tau))
        f.write("lambda: [%s]\nlambda-prior: 0\nmin-bpp: %g\n" % (",".join("%.6g" % v for v in lam), min_bpp))
        f.write("no-rss: %d\nno-profile: %d\nno-energy: %d\n" % (no_rss, no_prf, no_ene))


def row_sizes(pattern):
    sizes = [4]
    for c in pattern:
        if c == ".":
            sizes.append(4)
        elif c == ")":
            sizes.append(6)
    return sizes


def parse_scan(txt):
    recs, cur = [], None
    for line in txt.split("\n"):
        if line.startswith("id: "):
            cur = {"id": line[4:]}
            recs.append(cur)
        elif cur is not None and ": " in line or (cur is not None and line.endswith(":")):
            k, _, v = line.partition(":")
            v = v[1:] if v.startswith(" ") else v
            if k in ("start", "end", "inner"):
                cur[k] = [float(x) for x in v.strip("[]").split(",")]
            elif k == "psihat":
                cur[k] = [int(x) for x in v.strip("[]").split(",")]
            elif k == "motif region":
                a, b = v.split(" - ")
                cur["Ys"], cur["Ye"] = int(a), int(b)
            elif k == "exist prob":
                cur["exist_prob"] = float(v)
            elif k in ("seq", "rss", "mot"):
                cur[k] = v
    return recs


def main():
    os.makedirs(G, exist_ok=True)
    # ---- data files of the reference's own tests
    for f in ("0.fq", "1.fq", "0.model", "1.model", "2.model", "3.model"):
        open(os.path.join(G, f), "w").write(open(os.path.join(REF, "RNAelem-test", f)).read())
    with open(os.path.join(G, "rnafold_1_0_ubox.txt"), "w") as f:
        f.write("# `i j sqrt(p) ubox` lines of RNAelem-test/1.0.ps (RNAfold -p --maxBPspan=50, ViennaRNA 2.3.1)\n")
        for line in open(os.path.join(REF, "RNAelem-test", "1.0.ps")):
            a = line.split()
            if len(a) == 4 and a[3] == "ubox" and not a[0].startswith("%"):
                f.write(" ".join(a[:3]) + "\n")
    open(os.path.join(G, "positive.fa"), "w").write(open(os.path.join(REF, "material", "positive.fa")).read())

    # ---- inputs we generate (committed so the GPU box sees identical bytes)
    recs = synth.fasta_to_fastq_records(os.path.join(G, "positive.fa"))
    with open(os.path.join(G, "positive.fq"), "w") as f:
        for rid, s, q in recs:
            f.write("%s\n%s\n+\n%s\n" % (rid, s, q))
    with open(os.path.join(G, "positive_head6.fq"), "w") as f:
        for rid, s, q in recs[:6]:
            f.write("%s\n%s\n+\n%s\n" % (rid, s, q))
    for name, n, L in (("syn_L40_n3", 3, 40), ("syn_L100_n3", 3, 100), ("syn_L150_n8", 8, 150), ("syn_L200_n4", 4, 200)):
        seqs, quals = synth.synth_batch(n, L)
        if name == "syn_L100_n3":
            quals[1][-1] = 5   # one "no motif" record (last quality char != '!')
            quals[2][10:30] = np.arange(20) % 7 + 3  # graded qualities -> non-trivial position weights
        synth.write_fastq(os.path.join(G, name + ".fq"), seqs, quals)
    with open(os.path.join(G, "tiny.fq"), "w") as f:
        f.write("@t0\nGGGAAAUCCCAGGCUUCGGCCAAC\n+\n++++++++++++++++++++++++!\n")
        f.write("@t1\nACGUAACGGGAAACCGUUCGACGU\n+\n+++++,,,,,-----+++++++++&\n")

    rng = np.random.RandomState(20240807)

    def rand_rows(pattern, scale=0.8):
        return [list(np.round(rng.randn(k) * scale - 1.3, 5)) for k in row_sizes(pattern)]

    def unif_rows(pattern):
        return [[float(np.log(1.0 / k))] * k for k in row_sizes(pattern)]

    write_model(os.path.join(G, "trna_x0.model"), "(.....)", unif_rows("(.....)"), [0, 0])
    write_model(os.path.join(G, "trna_a.model"), "(.....)", rand_rows("(.....)"), [0.6, 1.3])
    write_model(os.path.join(G, "syn_x0.model"), "((.*.))", unif_rows("((.*.))"), [0, 0])
    write_model(os.path.join(G, "syn_l1.model"), "((.*.))", unif_rows("((.*.))"), [1, 1])
    write_model(os.path.join(G, "syn_b.model"), "((.*.))", rand_rows("((.*.))"), [0.8, 0.4])
    write_model(os.path.join(G, "syn_sm.model"), "((.*.))", rand_rows("((.*.))"), [0.5, 1.1], softmax=True)
    write_model(os.path.join(G, "tiny_a.model"), "(.*)", rand_rows("(.*)"), [0.7, 1.2], max_span=30, min_bpp=0)
    write_model(os.path.join(G, "tiny_ne.model"), "(.*).", rand_rows("(.*)."), [0.7, 1.2], max_span=30, min_bpp=0, no_ene=1)
    write_model(os.path.join(G, "syn_c12.model"), "(.(.).)", rand_rows("(.(.).)"), [1.0, 0.5], max_span=40, max_iloop=12)

    # ---- pattern automata
    pats = [".", "....", "(.)", "(.*)", ".*.", "(.).(.)", "(.)*(.)", "(.....)", "((.*.))", "..*..", "(.*).", "(.(.).)",
            "((...))", "(.(...).)", "**.(.**.).*", "(.)(.)", "((.)*(.))"]
    plist = [l.strip() for l in open(os.path.join(REF, "pattern_list")) if l.strip()]
    plist = [p for p in plist if re.fullmatch(r"[.()*]+", p)]  # the list holds one malformed entry ("...}")
    pats += [p for k, p in enumerate(plist) if k % 9 == 4 and p not in pats]
    hmm = {}
    for p in pats:
        hmm[p] = jload(run([os.path.join(RB, "ref_dump"), "hmm", p]))
    dump("hmm.json", hmm)
    # sizes only, for every preset pattern
    sizes = {}
    for p in plist:
        h = jload(run([os.path.join(RB, "ref_dump"), "hmm", p]))
        sizes[p] = [h["M"], h["S"], len(h["loop_state"]), len(h["loop_loop"]), sum(map(len, h["right"])),
                    sum(map(len, h["left"])), sum(map(len, h["pair"]))]
    dump("hmm_sizes.json", sizes)

    # ---- energy tables (log Boltzmann weights as parsed by the reference)
    for tag, name in (("T2004", "~T2004~"), ("A2007", "~A2007~")):
        e = jload(run([os.path.join(RB, "ref_dump"), "energy", name]))
        out = {}
        for k, v in e.items():
            if isinstance(v, list) and len(v) > 200:
                raw = b"".join(struct.pack("<d", x) for x in v)
                fin = [x for x in v if np.isfinite(x)]
                out[k] = {"n": len(v), "sha256": hashlib.sha256(raw).hexdigest(), "n_finite": len(fin),
                          "sum_finite": float(np.sum(fin)), "head": v[:60]}
            else:
                out[k] = v
        dump("energy_%s.json" % tag, out)

    # ---- BPP (K1)
    b = jload(run([os.path.join(RB, "ref_dump"), "bpp", os.path.join(G, "1.fq"), "50", "30", "1e-4"]))
    dump("bpp_1fq.json", b)
    b = jload(run([os.path.join(RB, "ref_dump"), "bpp", os.path.join(G, "syn_L100_n3.fq"), "50", "30", "1e-4"]))
    dump("bpp_syn_L100.json", b)

    # ---- fn / gr
    cases = [("0.model", "0.fq"), ("1.model", "0.fq"), ("2.model", "0.fq"), ("3.model", "0.fq"),
             ("trna_x0.model", "positive.fq"), ("trna_a.model", "positive.fq"), ("syn_x0.model", "syn_L150_n8.fq"),
             ("syn_l1.model", "syn_L150_n8.fq"), ("syn_b.model", "syn_L200_n4.fq"), ("syn_sm.model", "syn_L100_n3.fq"),
             ("syn_b.model", "syn_L100_n3.fq"), ("tiny_a.model", "tiny.fq"), ("tiny_ne.model", "tiny.fq"),
             ("syn_c12.model", "syn_L100_n3.fq"), ("syn_c12.model", "syn_L40_n3.fq")]
    ev = []
    for mdl, fq in cases:
        r = jload(run([os.path.join(RB, "ref_dump"), "eval", os.path.join(G, mdl), os.path.join(G, fq)]))
        ev.append({"model": mdl, "fq": fq, "n_seq": r["n_seq"], "x": r["x"], "fn": r["fn"], "gr": r["gr"][:-1],
                   "sum_eff": r["sum_eff"]})
        print(mdl, fq, r["fn"])
    dump("eval.json", ev)

    # ---- --lik-ratio objective (motif_trainer.hpp:156-202) on batches with both labels
    evl = []
    for mdl, fq in (("0.model", "0.fq"), ("syn_b.model", "syn_L100_n3.fq"), ("tiny_a.model", "tiny.fq"), ("1.model", "0.fq")):
        r = jload(run([os.path.join(RB, "ref_dump"), "eval", os.path.join(G, mdl), os.path.join(G, fq), "lik=1"]))
        evl.append({"model": mdl, "fq": fq, "n_seq": r["n_seq"], "x": r["x"], "fn": r["fn"], "gr": r["gr"][:-1], "sum_eff": r["sum_eff"]})
    dump("eval_lik.json", evl)

    # ---- per-sequence DP details
    dps = []
    for mdl, fq, full in (("0.model", "0.fq", 0), ("1.model", "0.fq", 0), ("2.model", "0.fq", 0), ("3.model", "0.fq", 0),
                          ("syn_b.model", "syn_L100_n3.fq", 0), ("syn_sm.model", "syn_L40_n3.fq", 0),
                          ("trna_a.model", "positive_head6.fq", 0), ("tiny_a.model", "tiny.fq", 1),
                          ("syn_c12.model", "syn_L40_n3.fq", 0)):
        r = jload(run([os.path.join(RB, "ref_dump"), "dp", os.path.join(G, fq), os.path.join(G, mdl), "full=%d" % full]))
        dps.append({"model": mdl, "fq": fq, "S": r["S"], "M": r["M"], "seqs": r["seqs"]})
    dump("dp.json", dps)

    # ---- scan records from the reference binary
    sc = []
    for mdl, fq in (("0.model", "0.fq"), ("1.model", "0.fq"), ("3.model", "0.fq"), ("trna_a.model", "positive_head6.fq"),
                    ("syn_b.model", "syn_L100_n3.fq"), ("syn_sm.model", "syn_L40_n3.fq"), ("tiny_a.model", "tiny.fq"),
                    ("syn_c12.model", "syn_L100_n3.fq")):
        out = os.path.join("/tmp", "scan_%s_%s.raw" % (mdl, fq))
        run([os.path.join(RB, "RNAelem"), "scan", "--fastq", os.path.join(G, fq), "--motif-model", os.path.join(G, mdl),
             "--out1", out])
        sc.append({"model": mdl, "fq": fq, "records": parse_scan(open(out).read())})
    dump("scan.json", sc)

    # ---- optimizer-loop traces of `RNAelem train --no-shuffle` (regularised objective after every L-BFGS-B iteration)
    tt = []
    for fq, pattern, iters in (("positive.fq", "(.....)", 12), ("positive_head6.fq", "(.....)", 10)):
        r = subprocess.run([os.path.join(RB, "RNAelem"), "train", "--fastq", os.path.join(G, fq), "--motif-pattern", pattern,
                            "--out1", "/tmp/tt.model", "--max-iter", str(iters), "--no-shuffle", "--batch-size", "-1", "-t", "8",
                            "--lambda-init", "0", "--epsilon", "1e-5"], capture_output=True, text=True)
        f = [float(m.group(2)) for m in re.finditer(r"^iter: (\d+) , f: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
        tt.append({"fq": fq, "pattern": pattern, "max_iter": iters, "epsilon": 1e-5, "lambda_init": 0, "rho_theta": 0.1,
                   "rho_lambda": 0.1, "tau": 0.1, "iter_f": f})
    dump("train_trace.json", tt)

    # ---- shuffled negatives: uShuffle driven as motif_trainer.hpp:145-152, and the Adam trace of the default train mode
    from rnaelem_amd import io as _io, synth as _synth
    sh = []
    for k in (1, 2, 3, 4):
        for it in (0, 1, 7):
            for sq in _synth.synth_batch(5, 60)[0][:3]:
                st = _io.decode_seq(sq)
                sh.append({"seq": st, "k": k, "iter": it,
                           "shuffled": run([os.path.join(RB, "ref_dump"), "shuffle", st, str(k), str(it)]).strip()})
    dump("shuffle.json", sh)
    r = subprocess.run([os.path.join(RB, "RNAelem"), "train", "--fastq", os.path.join(G, "positive_head6.fq"), "--motif-pattern", "(.....)",
                        "--out1", "/tmp/ts.model", "--max-iter", "6", "--batch-size", "-1", "-t", "1", "--lambda-init", "0"],
                       capture_output=True, text=True)
    ys = [float(m.group(2)) for m in re.finditer(r"^iter: (\d+) , y: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
    dump("train_trace_shuffle.json", {"fq": "positive_head6.fq", "pattern": "(.....)", "max_iter": 6, "kmer_shuf": 2, "rho_theta": 0.1,
                                      "rho_lambda": 0.1, "tau": 0.1, "lambda_init": 0, "iter_fn": ys})

    # ---- mini-batches: `--batch-size 2` over the 6 records, 8 evaluations (3 per epoch; epochs reshuffled by mt19937(epoch))
    r = subprocess.run([os.path.join(RB, "RNAelem"), "train", "--fastq", os.path.join(G, "positive_head6.fq"), "--motif-pattern", "(.....)",
                        "--out1", "/tmp/tb.model", "--max-iter", "8", "--batch-size", "2", "-t", "1", "--lambda-init", "0"],
                       capture_output=True, text=True)
    ys = [float(m.group(2)) for m in re.finditer(r"^iter: (\d+) , y: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
    dump("train_trace_minibatch.json", {"fq": "positive_head6.fq", "pattern": "(.....)", "max_iter": 8, "batch_size": 2, "kmer_shuf": 2,
                                        "rho_theta": 0.1, "rho_lambda": 0.1, "tau": 0.1, "lambda_init": 0, "iter_fn": ys})

    # ---- the reference's known-answer cases re-run through its debug configuration
    pc = []
    kat = [(".", "A", "."), (".", "AA", ".."), (".", "CAAAG", "(...)"), (".", "ACAAAGA", ".(...)."),
           (".", "ACACAAAGGA", ".(.(...))."), (".", "ACACAGACAGAAGA", ".(.(.).(.)..)."), (".", "CACAGAG", "(.(.).)"),
           ("(.)", "CAAAG", "(...)"), ("(.)", "CCAAAGG", "((...))"), ("(.*)", "CAAAG", "(...)"),
           ("(.*)", "CCAAAGG", "((...))"), (".*.", "AA", ".."), (".*.", "CAAAG", "(...)"), ("(.).(.)", "CAGACAG", "(.).(.)"),
           ("(.).(.)", "CCAGACAGG", "((.).(.))"), ("(.)*(.)", "CAGCAG", "(.)(.)"), ("(.)*(.)", "CCAGCAGG", "((.)(.))"),
           (".", "CAG", "(.)"), (".", "CACGG", "(...)"), (".", "CAGAU", "(.).."), ("(.*).", "CCAAAGGA", "((...))."),
           ("((.*.))", "GGCAAACAGCC", "((.(...).))")]
    for p, s, r in kat:
        pc.append(jload(run([os.path.join(RB, "ref_dump_dbg"), "pathcount", p, s, r])))
    dump("pathcount.json", pc)


if __name__ == "__main__":
    main()
