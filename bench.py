#!/usr/bin/env python3
"""Benchmark of the hot path: train-iteration sequences / second (BASELINE.json metric).

One "step" = one evaluation of (fn, gr) over the whole batch = per sequence K2 (inside) + 2 x K3
(outside + expected counts), the cached K1 (BPP filter) excluded, then ONE RCCL all-reduce of the
partial sums when N > 1.  The roofline "launch" is the whole diagonal pipeline of one evaluation (all
k4_* launches of the scaled-linear pipeline, timed with HIP events on the engine's stream:
elemdp_last_timing()[1]); `traffic` is the HBM byte count of the same pipeline from the committed
rocprofv3 --pmc passes (profiles/traffic.json), scaled to this launch.  Workload = BASELINE config C/D: 10 000 synthetic RNAs of L = 200, pattern
'((.*.))', x0 with lambda = (1,1) so the energy terms are exercised; the 10 000 sequences are sharded
over the N ranks (strong scaling, as the metric is quoted).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (a child `torch.distributed.run`,
started before this process touches the GPU); under a launcher WORLD_SIZE must equal --gpus or the run is refused.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PATTERN = "((.*.))"
N_SEQ, SEQ_LEN, MAX_SPAN, MAX_ILOOP = 10000, 200, 50, 30
PEAK_HBM_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)


def algorithmic_bytes_per_seq(L, S, W=MAX_SPAN):
    """5*T per sequence with the BPP filter cached, T = (L+1)(W+1)*7*S*8 + (L+1)*S*8."""
    T = (L + 1) * (W + 1) * 7 * S * 8 + (L + 1) * S * 8
    return 5 * T


def cpu_baseline(seqs, quals, x, n_theta):
    """Reference CPU path (oracle/_ref, kind 'reference') or the oracle port, on bounded samples: all host cores, and one
    thread (SURVEY.md section 8d asks for both)."""
    cores = len(os.sched_getaffinity(0))
    from rnaelem_amd import synth
    ref = os.path.join(REPO, "oracle", "_ref", "ref_dump")

    def run_ref(sample, threads):
        with tempfile.TemporaryDirectory() as td:
            fq, mdl = os.path.join(td, "s.fq"), os.path.join(td, "m.model")
            synth.write_fastq(fq, seqs[:sample], quals[:sample])
            rows, k = [], 0
            for w in [4, 4, 4, 6, 6]:
                rows.append(list(x[k:k + w]))
                k += w
            with open(mdl, "w") as f:
                f.write("pattern: %s\ntheta: [%s]\n" % (PATTERN, ",".join("[" + ",".join("%.17g" % v for v in r) + "]" for r in rows)))
                f.write("ene-param: ~T2004~\nmax-span: %d\nmax-internal-loop: %d\ntheta-softmax: 0\nrho-theta: 0.1\n" % (MAX_SPAN, MAX_ILOOP))
                f.write("rho-lambda: 0.1\ntau: 0.1\nlambda: [%.17g,%.17g]\nlambda-prior: 0\nmin-bpp: 0.0001\n" % (x[-2], x[-1]))
            out = subprocess.run([ref, "time", fq, str(threads), "1", mdl], capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                return None
            return json.loads(out.stdout.strip().split("\n")[-1])["seq_per_sec"]

    def run_port(sample, threads):
        from oracle import pyoracle as po
        o = po.make_oracle(PATTERN, MAX_SPAN, MAX_ILOOP, min_bpp=1e-4, tau=0.1)
        t0 = time.time()
        o.train_eval(x, seqs[:sample], quals[:sample], n_threads=threads)
        return sample / (time.time() - t0)

    kind, run = ("reference", run_ref) if os.path.exists(ref) else ("port", run_port)
    what = "RNAelemTrainer::operator()" if kind == "reference" else "oracle port"
    s_all, s_one = min(len(seqs), max(8, cores * 6)), min(len(seqs), 24)
    v_all = run(s_all, cores)
    if v_all is None:      # the reference binary failed on this box: the port
        kind, run, what = "port", run_port, "oracle port"
        v_all = run(s_all, cores)
    v_one = run(s_one, 1)
    return {"value": v_all, "unit": "seq/s", "cores": cores, "kind": kind,
            "sample": "%d of the %d sequences, 1 eval, %d threads (%s, BPP filter included)" % (s_all, len(seqs), cores, what),
            "one_thread": {"value": v_one, "unit": "seq/s", "cores": 1,
                           "sample": "%d of the %d sequences, 1 eval, 1 thread (%s, BPP filter included)" % (s_one, len(seqs), what)}}


def scan_algorithmic_bytes_per_seq(L, S, W=MAX_SPAN):
    """scan = 7*T + T_trace + 3*T_b per sequence (three inside-type passes write T each, two outside
    passes read + write 2*T each, the Viterbi pass writes 5 int32 per (cell, structural state, state), K1 = 3 plain tables)."""
    T = (L + 1) * (W + 1) * 7 * S * 8 + (L + 1) * S * 8
    T_trace = (L + 1) * (W + 1) * 7 * S * 20
    T_b = (L + 1) * (W + 1) * 7 * 8
    return 7 * T + T_trace + 3 * T_b


class ScanSecondary:
    """BASELINE's secondary metric on the config-E shape: `elem scan` sequences / second of a FRESH batch -- a scan visits
    every sequence once, so the BPP filter + plan (load_batch) belong in it: value = n / (load + scan).  An untimed scan of
    another batch of the same shape comes first (it allocates the table and trace slots: fresh device memory costs ~20 ms / GB,
    a one-off of the process, not of the batch)."""

    def __init__(self, api, synth, device, n=10000, L=300, pattern="(.....)"):
        self.n, self.L, self.pattern, self.synth = n, L, pattern, synth
        self.eng = api.Engine(pattern, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
        self.x = self.eng.initial_params(1.0)
        seqs, quals = synth.synth_batch(n, L, seed=77 + L)
        self.eng.load_batch(seqs, quals)
        self.eng.scan(self.x)

    def measure(self):
        eng, n = self.eng, self.n
        seqs, quals = self.synth.synth_batch(n, self.L)
        t0 = time.perf_counter()
        eng.load_batch(seqs, quals)
        t1 = time.perf_counter()
        eng.scan(self.x)
        t2 = time.perf_counter()
        alg = scan_algorithmic_bytes_per_seq(self.L, eng.n_state) * n
        ach = alg / (t2 - t0) / 1e9
        traffic, traffic_src = None, None      # HBM bytes of the scan's kernels from the committed rocprofv3 --pmc passes
        tf = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tf):
            t = json.load(open(tf)).get("scan")
            if t and t.get("seq_len") == self.L and t.get("pattern") == self.pattern:
                traffic, traffic_src = t["hbm_bytes_per_seq"] * n, t["source"]
        return {"metric": "scan seqs/sec (BPP filter + plan + K4/K5 sum passes + K6 Viterbi parse, fresh batch)", "value": n / (t2 - t0),
                "unit": "seq/s", "load_s": t1 - t0, "scan_s": t2 - t1, "resident_rate": n / (t2 - t1),
                "log_space_fallback_sequences": int(eng.last_timing()[2]),
                "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                             "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_seq": alg // n,
                             "note": "7*T + T_trace + 3*T_b (SURVEY section 8d: the reference's algorithm; this implementation keeps no "
                                     "band trace table, its own traffic is `traffic`) over the wall time of load + scan"},
                "workload": "%d synthetic RNAs L=%d, pattern %s (S=%d), W=%d C=%d" % (n, self.L, self.pattern, eng.n_state, MAX_SPAN, MAX_ILOOP)}


def minibatch_secondary(api, synth, device, n=2000, L=200, iters=40):
    """The reference's default training mode (`elem train`: mini-batches of 64 records + their shuffled negatives, Adam):
    wall time per optimizer iteration, load_batch included (every iteration loads its batch)."""
    from rnaelem_amd import train
    eng = api.Engine(PATTERN, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    seqs, quals = synth.synth_batch(n, L)

    eng2 = api.Engine(PATTERN, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    ev = train.MiniBatches(seqs, quals, 64, None, kmer_shuf=2, engines=[eng, eng2])   # (as rnaelem_amd.cli does)
    x0 = eng.initial_params(0.0)
    rho = train.regularisation(len(x0), 0.1, 0.1)
    # untimed: until both engines have evaluated a batch of theirs (MiniBatches loads `lookahead` = 8 iterations' batches at a
    # time, alternating between the engines) -- their buffers are then at their final sizes
    train.minimize_adam(ev, x0, rho, max_iter=2 * ev.lookahead + 2)
    t0 = time.perf_counter()
    train.minimize_adam(ev, x0, rho, max_iter=iters)
    dt = time.perf_counter() - t0
    ev.finish()
    eng.close()
    eng2.close()
    return {"metric": "default-mode train iteration (64 records + 64 shuffled negatives, load + evaluation)", "value": dt / iters * 1e3,
            "unit": "ms", "seq_per_s": 128 * iters / dt, "workload": "%d synthetic RNAs L=%d, pattern %s" % (n, L, PATTERN)}


def streamed_secondary(api, synth, device, resident_rate, n=60000, L=200, chunk=10000, reps=2):
    """A training set that is evaluated chunk by chunk (option max_resident forces what the handle decides by itself when a batch does
    not fit the device): chunk k is evaluated on one inner handle while the plan of chunk k + 1 is rebuilt on the other; the BPP
    filter's result of every chunk stays on the host after the first pass.  value = sequences / second of an evaluation after the
    first; `of_resident` = its ratio to the resident rate of this run."""
    eng = api.Engine(PATTERN, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    eng.set_option("max_resident", chunk)
    seqs, quals = synth.synth_batch(n, L)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    t0 = time.perf_counter()
    eng.train_eval(x)                      # (first pass: filter + plan of every chunk)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.train_eval(x)
    dt = (time.perf_counter() - t0) / reps
    eng.close()
    return {"metric": "streamed train-iter seqs/sec (chunks of %d, plan rebuilt per chunk and pass)" % chunk, "value": n / dt, "unit": "seq/s",
            "s_per_eval": dt, "first_eval_s": t_first, "of_resident": (n / dt) / resident_rate,
            "workload": "%d synthetic RNAs L=%d, pattern %s" % (n, L, PATTERN)}


def scan_streamed_secondary(api, synth, device, n=100000, L=300, chunk=12500, pattern="(.....)"):
    """BASELINE config E at its own size on ONE GPU: `elem scan` of 100 000 sequences of L = 300, streamed in chunks (option
    max_resident; the BPP filter + plan of chunk k + 1 are built on a second inner engine while chunk k is scanned), records kept
    in input order.  value = sequences / second of load_batch + scan (wall clock).  (On 8 GPUs every rank scans 12 500 of them:
    `per_gpu_at_shard`.)"""
    eng = api.Engine(pattern, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    eng.set_option("max_resident", chunk)
    x = eng.initial_params(1.0)
    # untimed: two chunks of another batch through both inner engines (their plan and table buffers are then allocated: fresh
    # device memory costs ~20 ms / GB, a one-off of the process as in ScanSecondary)
    seqs, quals = synth.synth_batch(2 * chunk + 1, L, seed=98)
    eng.load_batch(seqs, quals)
    eng.scan(x)
    seqs, quals = synth.synth_batch(n, L, seed=99)
    t0 = time.perf_counter()
    eng.load_batch(seqs, quals)
    t1 = time.perf_counter()
    recs, _ = eng.scan(x)
    t2 = time.perf_counter()
    assert len(recs) == n
    eng.close()
    return {"metric": "streamed scan seqs/sec (config E's size: load + scan, chunks of %d)" % chunk, "value": n / (t2 - t0), "unit": "seq/s",
            "load_s": t1 - t0, "scan_s": t2 - t1, "workload": "%d synthetic RNAs L=%d, pattern %s" % (n, L, pattern)}


def shard_projection(api, synth, device, steps=5):
    """What ONE GPU does at the shard sizes of the 8-GPU configurations D and E (10 000 / 8 train, 100 000 / 8 scan): measured on
    this GPU, labelled as a PROJECTION of the per-GPU rate of an 8-GPU run (no 8-GPU node was measured; the collective of D adds
    one all-reduce of ~60 doubles per step, E has none)."""
    eng = api.Engine(PATTERN, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    seqs, quals = synth.synth_batch(N_SEQ // 8, SEQ_LEN)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    eng.train_eval(x)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_eval(x)
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    n_scan, L_scan = 100000 // 8, 300
    sc = api.Engine("(.....)", "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, device)
    xs = sc.initial_params(1.0)
    s2, q2 = synth.synth_batch(n_scan, L_scan, seed=5)
    t0 = time.perf_counter()
    sc.load_batch(s2, q2)
    sc.scan(xs)
    ds_cold = time.perf_counter() - t0      # (a fresh engine: device buffers of plan and table slots allocated on the way, ~20 ms / GB)
    s2, q2 = synth.synth_batch(n_scan, L_scan, seed=6)
    t0 = time.perf_counter()
    sc.load_batch(s2, q2)
    sc.scan(xs)
    ds = time.perf_counter() - t0
    sc.close()
    return {"label": "projection: one GPU at the per-rank shard of the 8-GPU configurations (not an 8-GPU measurement)",
            "train_D": {"shard": N_SEQ // 8, "seq_len": SEQ_LEN, "ms_per_step": dt * 1e3, "per_gpu_seq_per_s": (N_SEQ // 8) / dt,
                        "projected_8gpu_seq_per_s": N_SEQ / dt},
            "scan_E": {"shard": n_scan, "seq_len": L_scan, "s_load_plus_scan": ds, "s_first_batch_of_a_fresh_engine": ds_cold,
                       "per_gpu_seq_per_s": n_scan / ds, "projected_8gpu_seq_per_s": 8 * n_scan / ds}}


def count_gpus_sysfs(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this machine without touching HIP or torch: KFD topology nodes with SIMDs (CPU nodes have simd_count 0), limited
    by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when set.  No KFD topology = no GPU driver = 0; None only when the directory
    exists but cannot be read (the ranks check their own device again)."""
    if not os.path.isdir(root):
        return 0
    try:
        n = 0
        for node in sorted(os.listdir(root)):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(root, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        vis = os.environ.get(var)
        if vis is not None:
            n = min(n, len([t for t in vis.split(",") if t.strip() != ""]))
    return n


def launcher_command(n, argv, port=None):
    """the child that runs the N ranks of `bench.py --gpus N` (one process per GPU, RCCL): python -m torch.distributed.run"""
    if port is None:
        port = 29500 + (os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(REPO, "bench.py")] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-seq", type=int, default=N_SEQ)
    ap.add_argument("--seq-len", type=int, default=SEQ_LEN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the scan measurement (BASELINE's secondary metric)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed evaluation of 60 000 sequences (secondary_streamed)")
    ap.add_argument("--serial-passes", action="store_true",
                    help="everything on one stream (for kernel traces: with concurrent streams kernels overlap and their durations no "
                         "longer add up to the pipeline time)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launch the ranks as a CHILD; this parent never touches torch or HIP (the GPUs are counted from the kernel driver's
        # topology files; every rank checks its own device again)
        have = count_gpus_sysfs()
        if have is not None and have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this machine has %d GPU(s); refusing to report a %d-GPU number" % (args.gpus, have, args.gpus))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:]), env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: refusing to run (the line would report the wrong n_gpus)" % (world, args.gpus))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d device(s))" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from rnaelem_amd import api, synth
    from rnaelem_amd.distributed import ShardedTrainer
    secondary = world == 1 and not args.no_secondary
    eng = api.Engine(PATTERN, "~T2004~", MAX_SPAN, MAX_ILOOP, 1e-4, 0.1, 0, local_rank)
    if args.serial_passes:
        eng.set_option("two_streams", 0)
        eng.set_option("group_streams", 1)
    seqs, quals = synth.synth_batch(args.n_seq, args.seq_len)
    x = eng.initial_params(1.0)
    t_load = time.time()
    trainer = ShardedTrainer(eng, seqs, quals, rank, world)
    t_load = time.time() - t_load
    n_local = trainer.range[1] - trainer.range[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer(x)
    barrier()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        fn, gr, eff, nsk = trainer(x)
        kern_ms.append(eng.last_timing()[1])
    barrier()
    dt = time.perf_counter() - t0
    # second timed point of SURVEY section 8(d): x0 as the CLI builds it, lambda = (0,0)
    x00 = eng.initial_params(0.0)
    trainer(x00)
    barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        trainer(x00)
    barrier()
    dt00 = (time.perf_counter() - t0) / 3
    shard_sizes = [n_local]
    if world > 1:
        t = torch.tensor([dt, dt00], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt00 = float(t[0].item()), float(t[1].item())
        sizes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device="cuda"))
        shard_sizes = [int(v.item()) for v in sizes]
        assert dist.get_world_size() == args.gpus and sum(shard_sizes) == args.n_seq
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    S = eng.n_state
    alg = algorithmic_bytes_per_seq(args.seq_len, S) * n_local
    k_s = float(np.mean(kern_ms)) / 1e3
    achieved = alg / k_s / 1e9
    # HBM traffic per sequence from the committed rocprofv3 --pmc pass (profiles/), scaled to this launch
    traffic, traffic_src = None, None
    tf = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tf):
        t = json.load(open(tf))
        if t.get("seq_len") == args.seq_len and t.get("pattern") == PATTERN:
            traffic, traffic_src = t["hbm_bytes_per_seq"] * n_local, t["source"]
    line = {
        "metric": "train-iter seqs/sec (inside+outside)", "value": args.n_seq * args.steps / dt, "unit": "seq/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "shard_sizes": shard_sizes,
        "second_point": {"x": "x0 of the CLI defaults, lambda = (0,0)", "value": args.n_seq / dt00, "unit": "seq/s", "ms_per_step": dt00 * 1e3,
                         "steps": 3},
        "config": {"workload": "%d synthetic RNAs L=%d, pattern %s, W=%d C=%d min_bpp=1e-4 T2004, x0 lambda=(1,1), one (fn,gr) eval per step" % (
            args.n_seq, args.seq_len, PATTERN, MAX_SPAN, MAX_ILOOP), "n_seq": args.n_seq, "seq_len": args.seq_len, "pattern": PATTERN,
            "sharding": "contiguous ranges per rank, 1 all-reduce of %d doubles per step" % eng.partial_len(),
            "bpp_filter": "cached (computed once at load_batch, %.1f s incl. plan)" % t_load, "fn": fn, "n_skipped": nsk},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k4_* diagonal pipeline of one evaluation (sum of all launches; dominant %s)" % eng.kernel_name(),
                     "log_space_fallback_sequences": int(eng.last_timing()[2]),
                     "kernel_ms": k_s * 1e3, "algorithmic_bytes_per_launch": alg},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(seqs, quals, x, eng.n_param - 2)
    if secondary:
        del trainer, eng
        import gc
        gc.collect()
        sec = ScanSecondary(api, synth, local_rank)      # (untimed: engine, a first batch, the scan that allocates the slots)
        line["secondary"] = sec.measure()
        del sec
        gc.collect()
        line["secondary_default_mode"] = minibatch_secondary(api, synth, local_rank)   # (its buffers persist: 3 untimed iterations)
        if not args.no_streamed:
            gc.collect()
            line["secondary_streamed"] = streamed_secondary(api, synth, local_rank, line["value"])
            gc.collect()
            line["secondary_scan_streamed"] = scan_streamed_secondary(api, synth, local_rank)
        gc.collect()
        line["per_gpu_at_shard"] = shard_projection(api, synth, local_rank)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
