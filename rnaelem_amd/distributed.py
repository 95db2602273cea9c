"""Data-parallel evaluation over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference's only multi-process mechanism is a Sun-Grid-Engine array job whose tasks evaluate the
contiguous ranges `assigned_range(total, n, k)` (RNAelem/arrayjob_manager.hpp:143-151) and exchange
`fn: / gr: / sum eff:` text files that the master sums (RNAelem/motif_array_trainer.hpp:20-58).
Here every rank keeps its range resident on its GPU, computes the local sums
[fn, sum_eff, n_used, n_skipped, ENo, ENx, EHo, EHx] on the device and ONE all-reduce (sum, fp64,
4 + 2*n_theta + 4 doubles ~ 0.5 KB: latency bound, so a single small collective per optimizer step)
replaces the files; every rank then finishes fn / gr identically, so the optimizer needs no broadcast.
"""
import numpy as np


def assigned_range(total, n, k):
    """Range [from, to) of 0-based part k of n: the first `total mod n` parts get one extra element."""
    base, res = divmod(total, n)
    start = k * base + min(k, res)
    return start, start + base + (1 if k < res else 0)


class ShardedTrainer:
    """== RNAelemTrainer::operator() (motif_trainer.hpp:595-633) over a batch sharded across ranks."""

    def __init__(self, engine, seqs, quals, rank=0, world=1, use_device_buffer=True):
        self.engine, self.rank, self.world = engine, rank, world
        self.total = len(seqs)
        a, b = assigned_range(self.total, world, rank)
        self.range = (a, b)
        if b > a:      # (more ranks than records: this rank contributes an all-zero partial vector)
            engine.load_batch(seqs[a:b], quals[a:b])
        self._buf = None
        self._device = use_device_buffer
        if world > 1:
            import torch
            self._torch = torch
            dev = "cuda" if use_device_buffer else "cpu"
            self._buf = torch.zeros(engine.partial_len(), dtype=torch.float64, device=dev)

    def __call__(self, x):
        eng = self.engine
        if self.world == 1:
            return eng.train_eval(x)
        import torch.distributed as dist
        if self.range[1] == self.range[0]:
            self._buf.zero_()
        elif self._device:
            eng.train_partial(x, device_ptr=self._buf.data_ptr())
        else:
            self._buf.copy_(self._torch.from_numpy(eng.train_partial(x)))
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
        return eng.train_finish(self._buf.cpu().numpy(), x=x)


def torch_all_reduce(partial):
    """Sum of a small host vector over the ranks (RCCL when the process group is "nccl", gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.float64).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class ShardedPairs:
    """The default `elem train` evaluation over several ranks: a list of records and, when `kmer_shuf` is set, the k-let
    shuffled negative of every record that was not skipped (motif_trainer.hpp:145-152, 228-245).  Every rank evaluates the
    contiguous part `assigned_range` of the records and the negatives of ITS records (the shuffle is seeded per sequence
    and evaluation, so the result does not depend on the sharding); the two partial vectors are added and ONE all-reduce
    per evaluation follows, as in ShardedTrainer.

    partial_pos(x) -> partial vector of the resident records; skipped_pos() -> bool per resident record;
    partial_batch(seqs, quals, x) -> partial vector of another batch (second engine); finish(total, x) ->
    (fn, gr, sum_eff, n_skipped); load(seqs, quals) makes records resident.  (api.Engine methods on the GPU; the CPU tests
    pass the emulation.)"""

    def __init__(self, load, partial_pos, skipped_pos, partial_batch, finish, partial_len, rank=0, world=1, kmer_shuf=None,
                 all_reduce=None):
        from .api import kmer_shuffle
        self._shuffle = kmer_shuffle
        self.load_fn, self.partial_pos, self.skipped_pos, self.partial_batch, self.finish = load, partial_pos, skipped_pos, partial_batch, finish
        self.rank, self.world, self.k = rank, world, kmer_shuf
        self.all_reduce = all_reduce or (torch_all_reduce if world > 1 else (lambda p: p))
        self.mine, self._n_loaded, self._partial_len = [], 0, partial_len

    def load(self, seqs, quals):
        a, b = assigned_range(len(seqs), self.world, self.rank)
        self.mine = list(seqs[a:b])
        self._n_loaded = b - a
        if b > a:
            self.load_fn(self.mine, list(quals[a:b]))

    def __call__(self, x, count):
        p = None
        if self._n_loaded:
            p = np.array(self.partial_pos(x), dtype=np.float64)
            if self.k is not None:
                negs = [self._shuffle(s, self.k, count) for s, sk in zip(self.mine, self.skipped_pos()) if not sk]
                if negs:
                    quals = [np.r_[np.zeros(len(s), dtype=np.uint8), np.uint8(1)] for s in negs]   # ws = 0, label "no motif"
                    pn = np.array(self.partial_batch(negs, quals, x), dtype=np.float64)
                    pn[1] = pn[2] = 0.       # "considered BP" and its count are statistics of the records themselves
                    p = p + pn
        if p is None:
            p = np.zeros(self._partial_len)
        return self.finish(self.all_reduce(p), x)

    @classmethod
    def on_engines(cls, eng, neg, rank=0, world=1, kmer_shuf=None, all_reduce=None):
        def partial_batch(s2, q2, x):
            neg.load_batch(s2, q2)
            return neg.train_partial(x)
        return cls(eng.load_batch, eng.train_partial, lambda: eng.seq_stats()[:, 4] != 0, partial_batch,
                   lambda total, x: eng.train_finish(total, x=x), eng.partial_len(), rank, world, kmer_shuf, all_reduce)


class ShardedShuffledNegatives:
    """Full-batch default mode over several ranks: the records stay resident, every evaluation gets fresh negatives."""

    def __init__(self, pairs, seqs, quals):
        self.pairs, self.count = pairs, 0
        pairs.load(seqs, quals)

    def __call__(self, x):
        res = self.pairs(x, self.count)
        self.count += 1
        return res


def reduce_and_finish(engine, partial, x):
    """Host-only tail of the sharded evaluation (used by the CPU `gloo` tests): all-reduce a partial
    vector that was produced elsewhere and turn it into (fn, gr, sum_eff, n_skipped)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return engine.train_finish(t.numpy(), x=x)
