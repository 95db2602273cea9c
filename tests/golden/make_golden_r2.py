#!/usr/bin/env python3
"""Round-2 fixtures from the compiled reference (oracle/_ref); adds to what make_golden.py wrote, rewrites nothing of it.

    make -C oracle ref && python tests/golden/make_golden_r2.py

Runs only where /root/reference exists (build container).  Fixtures are data: inputs we synthesize, files our own writer
produced, and outputs of the reference binary run on them.

  syn_L300_n4.fq, scan_e.json      BASELINE config E shape: 4 x L=300 scanned with a '(.....)' model (S = 29)
  scan_raw_*.raw                   raw text of `RNAelem scan` (-t 1) for a byte comparison with io.scan_record
  written_trna_a.model (+ scan)    a model written by io.write_model, read back by RNAelemReader: its scan records
  ref_written.model                a model written by the reference's own writer after two L-BFGS-B iterations
  eval_text.json                   `RNAelem eval` out1 / out2 text (fn: / gr: at 17 digits, motif_eval.hpp:23-54)
  train_trace_mask.json            `RNAelem train --param-set` traces, L-BFGS-B and Adam (motif_mask_trainer.hpp:28-111)
  scan_norss.json                  `RNAelem scan` of the reference's no-rss model 2.model
"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from make_golden import G, RB, dump, parse_scan, run  # noqa: E402
from rnaelem_amd import io, synth  # noqa: E402

BIN = os.path.join(RB, "RNAelem")


def scan(model, fq, out):
    run([BIN, "scan", "--fastq", fq, "--motif-model", model, "--out1", out, "-t", "1"])
    return open(out).read()


def main():
    # ---- config E shape
    seqs, quals = synth.synth_batch(4, 300)
    quals[2][-1] = 7   # one record without the "has motif" flag (the scan ignores the label: ws[L] only masks in training)
    synth.write_fastq(os.path.join(G, "syn_L300_n4.fq"), seqs, quals)
    txt = scan(os.path.join(G, "trna_a.model"), os.path.join(G, "syn_L300_n4.fq"), "/tmp/scan_e.raw")
    dump("scan_e.json", [{"model": "trna_a.model", "fq": "syn_L300_n4.fq", "records": parse_scan(txt)}])

    # ---- raw scan text (single thread: the reference writes records in completion order)
    for mdl, fq in (("0.model", "0.fq"), ("trna_a.model", "positive_head6.fq")):
        txt = scan(os.path.join(G, mdl), os.path.join(G, fq), "/tmp/raw.raw")
        name = "scan_raw_%s_%s.raw" % (mdl.split(".")[0], fq.split(".")[0])
        open(os.path.join(G, name), "w").write(txt)
        print("wrote", name, len(txt))

    # ---- our writer -> the reference's reader
    m = io.read_model(os.path.join(G, "trna_a.model"))
    io.write_model(os.path.join(G, "written_trna_a.model"), m)
    txt = scan(os.path.join(G, "written_trna_a.model"), os.path.join(G, "positive_head6.fq"), "/tmp/w.raw")
    dump("scan_written_model.json", [{"model": "written_trna_a.model", "fq": "positive_head6.fq", "records": parse_scan(txt)}])

    # ---- the reference's writer
    # (the binary without a sub-command = what script/elem spawns: train, write the model to out1, scan to out2, main.cpp:47-84;
    # the `train` sub-command never writes the final model: its writer has no stream id, main.cpp:110-111)
    r = subprocess.run([BIN, "--fastq", os.path.join(G, "positive_head6.fq"), "--motif-pattern", "(.....)", "--out1",
                        os.path.join(G, "ref_written.model"), "--out2", "/tmp/rw.raw", "--max-iter", "2", "--no-shuffle", "--batch-size",
                        "-1", "-t", "1", "--lambda-init", "0.5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([BIN, "--fastq", os.path.join(G, "syn_L40_n3.fq"), "--motif-pattern", "((.*.))", "--out1",
                        os.path.join(G, "ref_written_sm.model"), "--out2", "/tmp/rw2.raw", "--max-iter", "2", "--no-shuffle",
                        "--batch-size", "-1", "-t", "1", "--theta-softmax", "--lambda-init", "0.25"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    # ---- eval sub-command
    ev = []
    for mdl, fq, extra in (("0.model", "0.fq", []), ("1.model", "0.fq", []), ("3.model", "0.fq", []),
                           ("0.model", "0.fq", ["--lik-ratio"])):
        run([BIN, "eval", "--fastq", os.path.join(G, fq), "--motif-model", os.path.join(G, mdl), "--out1", "/tmp/ev1.txt", "--out2",
             "/tmp/ev2.txt", "--no-shuffle", "--batch-size", "-1", "-t", "1"] + extra)
        ev.append({"model": mdl, "fq": fq, "args": extra, "out1": open("/tmp/ev1.txt").read(), "out2": open("/tmp/ev2.txt").read()})
    dump("eval_text.json", ev)

    # ---- mask trainer
    tm = []
    for vary, extra, key in (("0-3,30,31", ["--no-shuffle"], r"^iter: (\d+) , f: ([-0-9.e+]+)"),
                             ("4-7,31", [], r"^iter: (\d+) , y: ([-0-9.e+]+)")):
        r = subprocess.run([BIN, "--fastq", os.path.join(G, "positive_head6.fq"), "--motif-pattern", "(.....)", "--out1",
                            "/tmp/tm.model", "--out2", "/tmp/tm.raw", "--max-iter", "6", "--batch-size", "-1", "-t", "1", "--lambda-init", "0.5",
                            "--param-set", vary, "--epsilon", "1e-5"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        ys = [float(mm.group(2)) for mm in re.finditer(key, r.stdout + r.stderr, re.M)]
        fin = io.read_model("/tmp/tm.model")
        tm.append({"fq": "positive_head6.fq", "pattern": "(.....)", "max_iter": 6, "param_set": vary, "no_shuffle": bool(extra),
                   "lambda_init": 0.5, "epsilon": 1e-5, "trace": ys, "final_x": list(fin["x"])})
        print("mask", vary, ys)
    dump("train_trace_mask.json", tm)

    # ---- scan in --no-rss mode
    txt = scan(os.path.join(G, "2.model"), os.path.join(G, "0.fq"), "/tmp/nr.raw")
    dump("scan_norss.json", [{"model": "2.model", "fq": "0.fq", "records": parse_scan(txt)}])


if __name__ == "__main__":
    main()
