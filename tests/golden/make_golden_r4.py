#!/usr/bin/env python3
"""Round-4 fixtures from the compiled reference (oracle/_ref); adds to what the earlier generators wrote.

    make -C oracle ref && python tests/golden/make_golden_r4.py

Runs only where /root/reference exists (build container).  Fixtures are data: outputs of the reference binary.

  train_trace_lik_shuffle.json   `RNAelem train --lik-ratio` in the DEFAULT mode (Adam, a shuffled negative per record and
                                 iteration: motif_trainer.hpp:156-202 with :145-152), whole batch and mini-batches of 2: the data
                                 term and the gradient norm the optimizer prints at every iteration
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from make_golden import G, RB, dump  # noqa: E402

BIN = os.path.join(RB, "RNAelem")


def main():
    out = []
    for fq, pattern, iters, batch in (("positive_head6.fq", "(.....)", 6, -1), ("positive_head6.fq", "(.....)", 8, 2),
                                      ("0.fq", "((.*.))", 5, -1)):
        m = "/tmp/tl.model"
        r = subprocess.run([BIN, "train", "--fastq", os.path.join(G, fq), "--motif-pattern", pattern, "--out1", m, "--max-iter", str(iters),
                            "--batch-size", str(batch), "-t", "1", "--lambda-init", "0", "--lik-ratio"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        ys = [float(x.group(2)) for x in re.finditer(r"^iter: (\d+) , y: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
        assert len(ys) == iters, (len(ys), r.stdout[-500:])
        gn = [float(x.group(1)) for x in re.finditer(r"^iter: \d+ , y: [-0-9.e+]+ , \|gr\|: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
        out.append({"fq": fq, "pattern": pattern, "max_iter": iters, "batch_size": batch, "kmer_shuf": 2, "rho_theta": 0.1, "rho_lambda": 0.1,
                    "tau": 0.1, "lambda_init": 0, "lik_ratio": 1, "iter_fn": ys, "iter_gnorm": gn})
        print(fq, pattern, batch, ys)
    dump("train_trace_lik_shuffle.json", out)


if __name__ == "__main__":
    main()
