#!/usr/bin/env python3
"""Round-3 fixtures from the compiled reference (oracle/_ref); adds to what the earlier generators wrote.

    make -C oracle ref && python tests/golden/make_golden_r3.py

Runs only where /root/reference exists (build container).  Fixtures are data: outputs of the reference binary.

  train_final.json   the binary without a sub-command (train, write the model to --out1, scan the training set to --out2:
                     main.cpp:47-84, what script/elem spawns) with --no-shuffle (L-BFGS-B): objective after every iteration as
                     printed, the parameters of the final model, and the scan records of the training set under that model
  train_converged.json  the same runs to convergence (--max-iter 400): final objective, iterations, parameters, Ys / Ye per record
  dp_A2007.json      fn / gr of RNAelemTrainer::operator() with the Andronescu 2007 energy parameters (~A2007~)
"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from make_golden import G, RB, dump, jload, parse_scan, run  # noqa: E402
from rnaelem_amd import io  # noqa: E402

BIN = os.path.join(RB, "RNAelem")


def main():
    out = []
    for fq, pattern, iters in (("positive_head6.fq", "(.....)", 10), ("positive.fq", "(.....)", 12)):
        m, raw = "/tmp/tf.model", "/tmp/tf.raw"
        r = subprocess.run([BIN, "--fastq", os.path.join(G, fq), "--motif-pattern", pattern, "--out1", m, "--out2", raw, "--max-iter", str(iters),
                            "--no-shuffle", "--batch-size", "-1", "-t", "8", "--lambda-init", "0", "--epsilon", "1e-5"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        f = [float(x.group(2)) for x in re.finditer(r"^iter: (\d+) , f: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
        model = io.read_model(m)
        out.append({"fq": fq, "pattern": pattern, "max_iter": iters, "epsilon": 1e-5, "lambda_init": 0, "iter_f": f,
                    "x": [float(v) for v in model["x"]],
                    "records": [{k: r[k] for k in ("id", "Ys", "Ye", "exist_prob", "rss", "mot", "psihat")} for r in parse_scan(open(raw).read())]})
        print(fq, len(f), "iterations,", len(out[-1]["records"]), "records")
    dump("train_final.json", out)

    # ---- the same runs to convergence (pgtol = --epsilon 1e-5): what an optimizer of another L-BFGS-B version has to reach
    conv = []
    for fq, pattern in (("positive_head6.fq", "(.....)"), ("positive.fq", "(.....)")):
        m, raw = "/tmp/tc.model", "/tmp/tc.raw"
        r = subprocess.run([BIN, "--fastq", os.path.join(G, fq), "--motif-pattern", pattern, "--out1", m, "--out2", raw, "--max-iter", "400",
                            "--no-shuffle", "--batch-size", "-1", "-t", "8", "--lambda-init", "0", "--epsilon", "1e-5"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        txt = r.stdout + r.stderr
        assert "lbfgsb converged" in txt
        f = [float(x.group(2)) for x in re.finditer(r"^iter: (\d+) , f: ([-0-9.e+]+)", txt, re.M)]
        recs = sorted(parse_scan(open(raw).read()), key=lambda q: q["id"])
        conv.append({"fq": fq, "pattern": pattern, "epsilon": 1e-5, "lambda_init": 0, "rho_theta": 0.1, "rho_lambda": 0.1, "tau": 0.1,
                     "n_iter": len(f) - 1, "final_f": float(re.search(r"final value: ([-0-9.e+]+)", txt).group(1)),
                     "x": [float(v) for v in io.read_model(m)["x"]],
                     "records": [{k: q[k] for k in ("id", "Ys", "Ye", "exist_prob")} for q in recs]})
        print(fq, "converged after", conv[-1]["n_iter"], "iterations, f =", conv[-1]["final_f"])
    dump("train_converged.json", conv)

    # ---- fn / gr per sequence under the Andronescu 2007 parameters (energy_model.hpp:155-160): a model our writer wrote
    # (syn_b.model with ene-param ~A2007~), evaluated by the reference's RNAelemTrainer::operator() (ref_dump dp)
    m = io.read_model(os.path.join(G, "syn_b.model"))
    m["ene_param"] = "~A2007~"
    io.write_model(os.path.join(G, "syn_a2007.model"), m)
    r = jload(run([os.path.join(RB, "ref_dump"), "dp", os.path.join(G, "syn_L100_n3.fq"), os.path.join(G, "syn_a2007.model"), "full=0"]))
    keys = ("id", "L", "W", "positive", "bpp_eff", "Zo", "Zari", "Znasi", "f", "ENo", "EHo", "ENx", "EHx")
    dump("dp_A2007.json", [{"model": "syn_a2007.model", "fq": "syn_L100_n3.fq", "S": r["S"], "M": r["M"],
                            "seqs": [{k: q[k] for k in keys if k in q} for q in r["seqs"]]}])


if __name__ == "__main__":
    main()
