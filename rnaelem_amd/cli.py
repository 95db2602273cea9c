"""Command line of the path: the option names of the reference binary's `train` / `scan` sub-commands
(RNAelem/application.hpp option table; what `script/elem` spawns), driving libelemdp instead of the CPU DP.

    python -m rnaelem_amd.cli train --fastq pos.fq --motif-pattern '((.*.))' --out1 model.txt --no-shuffle [-i 300]
    python -m rnaelem_amd.cli scan  --fastq seqs.fq --motif-model model.txt --out1 scan.raw
    python -m rnaelem_amd.cli       --fastq pos.fq --motif-pattern '((.*.))' --out1 model.txt --out2 scan.raw
                                    # no sub-command = what script/elem spawns: train, write the model, scan (main.cpp:47-84)
    python -m rnaelem_amd.cli eval  --fastq pos.fq --motif-model model.txt --out1 fn.txt --out2 gr.txt     (motif_eval.hpp:23-54)
    python -m rnaelem_amd.cli array-eval --array N --task-id K ... --out4 part   # part K of N: index / range / fn / gr / sum eff
    torchrun --nproc-per-node 8 -m rnaelem_amd.cli train ...      # one rank per GPU, one RCCL all-reduce per evaluation
    torchrun --nproc-per-node 8 -m rnaelem_amd.cli scan ...       # ranks scan contiguous ranges; rank 0 joins them in input order

Implemented: full-batch training -- `--no-shuffle` (L-BFGS-B, the path named by BASELINE.json) and the default mode with
per-iteration shuffled negatives (Adam, `--kmer-shuf`) --, mini-batches (`--batch-size N`), all of them on one or several
GPUs (torchrun), `--lik-ratio`,
`--param-set`, `--theta-softmax`, and `scan`.  Not implemented: grid-engine array jobs (out of scope).
"""
import argparse
import os
import sys

import numpy as np

from . import api, io, train as trainer
from .distributed import ShardedPairs, ShardedShuffledNegatives, ShardedTrainer


def build_parser():
    p = argparse.ArgumentParser(prog="rnaelem_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = p.add_subparsers(dest="cmd", required=True)
    for name in ("train", "scan", "normal", "eval", "array-eval"):
        s = sub.add_parser(name)
        s.add_argument("-f", "--fastq", required=True, help="input FASTQ with pseudo-qualities (L+1 quality characters)")
        s.add_argument("--out1", required=name != "array-eval", help="train / normal: model file; scan: raw records; eval: 'fn:' line")
        s.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK or 0)")
    for name in ("train", "normal"):
        _train_options(sub.choices[name])
    sub.choices["normal"].add_argument("--out2", required=True, help="raw scan records of the training sequences")
    for name in ("scan", "eval", "array-eval"):
        sub.choices[name].add_argument("-q", "--motif-model", required=True)
    for name in ("scan", "normal"):
        sub.choices[name].add_argument("--chunk", type=int, default=20000,
                                       help="sequences resident on the GPU at a time (plan + tables; BASELINE config E = 100 k x L=300 "
                                            "does not fit at once)")
    sub.choices["eval"].add_argument("--out2", required=True, help="'gr:' line")
    a = sub.choices["array-eval"]
    a.add_argument("-a", "--array", type=int, required=True, help="number of parts")
    a.add_argument("--task-id", type=int, default=None, help="1-based part (default: $SGE_TASK_ID)")
    a.add_argument("--out4", required=True, help="result file prefix: <out4>-<task id> (motif_array_trainer.hpp:25)")
    for name in ("eval", "array-eval"):
        sub.choices[name].add_argument("--lik-ratio", action="store_true")
    return p


def _train_options(t):
    t.add_argument("-m", "--motif-pattern", required=True)
    t.add_argument("-i", "--max-iter", type=int, default=300)
    t.add_argument("--energy-param", default="~T2004~")
    t.add_argument("-w", "--max-span", type=int, default=50)
    t.add_argument("-c", "--max-internal-loop", type=int, default=30)
    t.add_argument("--epsilon", type=float, default=1e-3)
    t.add_argument("--rho-s", type=float, default=0.1)
    t.add_argument("--rho-theta", type=float, default=0.1)
    t.add_argument("--rho-lambda", type=float, default=0.1)
    t.add_argument("--tau", type=float, default=0.1)
    t.add_argument("--lambda-init", type=float, default=0.0)
    t.add_argument("--lambda-prior", type=float, default=0.0)
    t.add_argument("-p", "--min-bpp", type=float, default=1e-4)
    t.add_argument("--no-rss", action="store_true")
    t.add_argument("--no-profile", action="store_true")
    t.add_argument("--no-energy", action="store_true")
    t.add_argument("--no-shuffle", action="store_true")
    t.add_argument("--theta-softmax", action="store_true")
    t.add_argument("--lik-ratio", action="store_true")
    t.add_argument("--batch-size", type=int, default=-1)
    t.add_argument("--kmer-shuf", type=int, default=2)
    t.add_argument("--param-set", default=None, help="comma separated indexes of the parameters to fit (the others stay fixed)")
    t.add_argument("--optimizer", choices=["lbfgsb", "adam"], default="lbfgsb")


def parse_param_set(spec):
    """--param-set: comma separated indexes or ranges `a-b` (inclusive), as application.hpp:376-388 reads it"""
    out = []
    for part in spec.split(","):
        se = [int(v) for v in part.split("-")]
        out += [se[0]] if len(se) == 1 else list(range(se[0], se[1] + 1))
    return out


def _rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def cmd_train(a):
    rank, local_rank, world = _rank_world()
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    flags = (api.NO_RSS if a.no_rss else 0) | (api.NO_PROFILE if a.no_profile else 0) | (api.NO_ENERGY if a.no_energy else 0) | \
        (api.THETA_SOFTMAX if a.theta_softmax else 0) | (api.LIK_RATIO if a.lik_ratio else 0)
    pattern = a.motif_pattern.replace("_", ".") if a.no_rss else a.motif_pattern
    par = a.energy_param if a.energy_param in ("~T2004~", "~A2007~") else open(a.energy_param).read()
    device = a.device if a.device is not None else local_rank
    eng = api.Engine(pattern, par, a.max_span, a.max_internal_loop, a.min_bpp, a.tau, flags, device)
    recs = io.read_fastq(a.fastq)
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    sharded_pairs = world > 1 and (not a.no_shuffle or a.batch_size > 0)
    ev = None if (a.batch_size > 0 or sharded_pairs) else ShardedTrainer(eng, seqs, quals, rank, world)
    x0 = eng.initial_params(a.lambda_init)
    log = (lambda msg: print(msg, file=sys.stderr, flush=True)) if rank == 0 else None
    vary = parse_param_set(a.param_set) if a.param_set else None
    optimizer = a.optimizer
    if sharded_pairs:      # several GPUs: records (and their negatives) sharded per evaluation, one all-reduce
        neg = api.Engine(pattern, par, a.max_span, a.max_internal_loop, a.min_bpp, a.tau, flags, device)
        pairs = ShardedPairs.on_engines(eng, neg, rank, world, None if a.no_shuffle else a.kmer_shuf)
        if a.batch_size > 0:
            ev = trainer.MiniBatches(seqs, quals, a.batch_size, None, pairs=pairs)
        else:
            ev = ShardedShuffledNegatives(pairs, seqs, quals)
        optimizer = "lbfgsb" if a.no_shuffle else "adam"
    elif a.batch_size > 0:   # mini-batches: every evaluation loads the next records of the epoch order (motif_trainer.hpp:595-632)
        def eval_batch(s2, q2, x):
            eng.load_batch(s2, q2)
            return eng.train_eval(x) + (eng.seq_stats()[:, 4] != 0,)

        eng2 = api.Engine(pattern, par, a.max_span, a.max_internal_loop, a.min_bpp, a.tau, flags, device)
        # (double-buffered: the next batch loads on one engine while the other evaluates the current one)
        ev = trainer.MiniBatches(seqs, quals, a.batch_size, eval_batch, None if a.no_shuffle else a.kmer_shuf, engines=[eng, eng2])
        optimizer = "lbfgsb" if a.no_shuffle else "adam"
    elif not a.no_shuffle:   # default `elem train`: Adam over positives + per-iteration shuffled negatives (main.cpp:132-152)
        neg = api.Engine(pattern, par, a.max_span, a.max_internal_loop, a.min_bpp, a.tau, flags, device)

        def eval_neg(s2, q2, x):
            neg.load_batch(s2, q2)
            return neg.train_eval(x)

        ev = trainer.ShuffledNegatives(seqs, eng.train_eval, lambda: eng.seq_stats()[:, 4] != 0, eval_neg, a.kmer_shuf)
        optimizer = "adam"
    res = trainer.train(ev, x0, a.rho_s if a.theta_softmax else a.rho_theta, a.rho_lambda, a.max_iter, a.epsilon, optimizer, log, vary)
    if hasattr(ev, "finish"):
        ev.finish()
    if rank == 0:
        d = eng.describe()
        rows, k = [], 0
        for w in d["theta_sizes"]:
            rows.append([0.0] * w)
            k += w
        m = dict(pattern=pattern, rows=rows, lam=[0.0, 0.0], softmax=a.theta_softmax, ene_param=a.energy_param, max_span=a.max_span,
                 max_iloop=a.max_internal_loop, min_bpp=a.min_bpp, tau=a.tau, rho_theta=a.rho_theta, rho_s=a.rho_s,
                 rho_lambda=a.rho_lambda, lambda_prior=a.lambda_prior, no_rss=a.no_rss, no_prf=a.no_profile, no_ene=a.no_energy)
        io.write_model(a.out1, m, res["x"])
        print("%s after %d iterations (%d evaluations); final value: %.6g" % (res["message"], res["n_iter"], res["n_eval"], res["f"]),
              file=sys.stderr)
    if a.cmd == "normal":     # main.cpp:73-81: the trained model scans the training sequences
        m = io.read_model(a.out1) if rank == 0 else None
        if world > 1:
            box = [m]
            dist.broadcast_object_list(box, src=0)
            m = box[0]
        _scan_records(eng, m, recs, a.out2, a.chunk, rank, world, (lambda: dist.barrier()) if world > 1 else (lambda: None))
    if world > 1:
        dist.destroy_process_group()


def sharded_scan(recs, out1, rank, world, scan_part, barrier):
    """Scan needs no collective: rank k scans the contiguous range assigned_range(n, world, k) and writes
    `<out1>.<k>`; after a barrier rank 0 joins the parts in rank order = input order into `out1` (the reference's writer
    emits records as its threads finish, motif_scanner.hpp:237-252; here the order is the input's).  `scan_part(records)`
    yields the record texts of a list of (id, codes, quals)."""
    from .distributed import assigned_range
    lo, hi = assigned_range(len(recs), world, rank)
    out = out1 if world == 1 else "%s.%d" % (out1, rank)
    with open(out, "w") as f:
        for text in scan_part(recs[lo:hi]):
            f.write(text)
    if world == 1:
        return
    barrier()
    if rank == 0:
        with open(out1, "w") as f:
            for k in range(world):
                part = "%s.%d" % (out1, k)
                with open(part) as g:
                    for line in g:
                        f.write(line)
                os.remove(part)
    barrier()


def _scan_records(eng, m, recs, out1, chunk, rank, world, barrier):
    nodes = eng.describe()["node"]
    step = max(1, chunk)

    def scan_part(mine):
        for c0 in range(0, len(mine), step):        # records are independent: chunks in input order
            part = mine[c0:c0 + step]
            eng.load_batch([s for _, s, _ in part], [q for _, _, q in part])
            res, en = eng.scan(m["x"])
            for (rid, codes, _), r in zip(part, res):
                yield io.scan_record(rid, codes, r, nodes)

    sharded_scan(recs, out1, rank, world, scan_part, barrier)


def cmd_scan(a):
    rank, local_rank, world = _rank_world()
    barrier = lambda: None
    if world > 1:     # only for the barrier around the join of the per-rank files: gloo, no device collective
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        barrier = dist.barrier
    m = io.read_model(a.motif_model)
    eng = io.engine_from_model(m, a.device if a.device is not None else local_rank)
    _scan_records(eng, m, io.read_fastq(a.fastq), a.out1, a.chunk, rank, world, barrier)
    if world > 1:
        dist.destroy_process_group()


def eval_text(fn, gr):
    """The two lines `RNAelem eval` prints (datp: ostream precision 17, vectors as [a,b,...]; motif_eval.hpp:46-47,
    util.hpp:98-105, 175-181)."""
    return "fn: %.17g\n" % fn, "gr: [%s]\n" % ",".join("%.17g" % v for v in gr)


def cmd_eval(a):
    """`eval`: fn / gr of a model file over the whole FASTQ (--no-shuffle semantics: the records as they are); `array-eval`:
    the same over part K of N (arrayjob_manager.hpp:143-151) in the key: value layout that collect_fn_gr_eff reads
    (motif_array_trainer.hpp:20-58).  Note: the reference binary's own `eval` prints zeros -- it never calls set_conditions,
    so its reader hands out no record (tests/golden/eval_text.json) -- ; the format is its, the numbers are the path's."""
    rank, local_rank, world = _rank_world()
    m = io.read_model(a.motif_model)
    if a.lik_ratio:
        m["flags"] |= api.LIK_RATIO
    eng = io.engine_from_model(m, a.device if a.device is not None else local_rank)
    recs = io.read_fastq(a.fastq)
    if a.cmd == "array-eval":
        from .distributed import assigned_range
        tid = a.task_id if a.task_id is not None else int(os.environ.get("SGE_TASK_ID", "1"))
        lo, hi = assigned_range(len(recs), a.array, tid - 1)
        fn, gr, eff = 0.0, np.zeros(eng.n_param), 0.0
        if hi > lo:
            eng.load_batch([s for _, s, _ in recs[lo:hi]], [q for _, _, q in recs[lo:hi]])
            fn, gr, eff, _ = eng.train_eval(m["x"])
        l1, l2 = eval_text(fn, gr)
        with open("%s-%d" % (a.out4, tid), "w") as f:
            f.write("index: %d / %d\nrange: %d - %d\n%s%ssum eff: %.17g\n" % (tid, a.array, lo, hi, l1, l2, eff))
        return
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    ev = ShardedTrainer(eng, [s for _, s, _ in recs], [q for _, _, q in recs], rank, world)
    fn, gr, eff, _ = ev(m["x"])
    if rank == 0:
        l1, l2 = eval_text(fn, gr)
        open(a.out1, "w").write(l1)
        open(a.out2, "w").write(l2)
    if world > 1:
        dist.destroy_process_group()


SUBCOMMANDS = ("train", "scan", "normal", "eval", "array-eval")


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in SUBCOMMANDS + ("-h", "--help"):
        argv = ["normal"] + argv          # the binary without a sub-command (PM_NORMAL, application.hpp:306)
    a = build_parser().parse_args(argv)
    {"train": cmd_train, "normal": cmd_train, "scan": cmd_scan, "eval": cmd_eval, "array-eval": cmd_eval}[a.cmd](a)


if __name__ == "__main__":
    main()
