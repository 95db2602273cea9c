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
        if self._device:
            eng.train_partial(x, device_ptr=self._buf.data_ptr())
        else:
            self._buf.copy_(self._torch.from_numpy(eng.train_partial(x)))
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
        return eng.train_finish(self._buf.cpu().numpy())


def reduce_and_finish(engine, partial, x):
    """Host-only tail of the sharded evaluation (used by the CPU `gloo` tests): all-reduce a partial
    vector that was produced elsewhere and turn it into (fn, gr, sum_eff, n_skipped)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return engine.train_finish(t.numpy(), x=x)
