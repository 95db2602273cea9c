"""Host optimizer loop around the evaluation path.

Mirrors how `elem train` drives `RNAelemTrainer::operator()`:

* objective handed to the optimizer = fn + sum_i rho_i x_i^2 / 2, gradient + rho_i x_i (L2 regulariser added by the
  optimizer, `Lbfgsb::before_update` / `Adam::before_update`, RNAelem/optimizer.hpp:246-260, 107-122);
  rho = rho_theta (or rho_s with --theta-softmax) for the theta entries, rho_lambda for lambda
  (`set_regularization`, RNAelem/motif_trainer.hpp:527-540);
* bounds: theta unbounded below (lower bound log 0), lambda >= 0 (`set_bounds`, :508-526);
* `--no-shuffle` mode: L-BFGS-B with m = 5, factr = 1e7, pgtol = --epsilon, at most --max-iter iterations, the best
  point seen at any evaluation is the result (`Lbfgsb::minimize`, optimizer.hpp:262-334).  The reference embeds the
  L-BFGS-B 2.1 routine `setulb`; here the same algorithm family comes from SciPy (L-BFGS-B 3.0), so iterates agree with
  the reference's trace to optimizer tolerance, not bitwise (tests/test_train_loop.py);
* otherwise Adam with alpha = 0.1, beta = (0.9, 0.999), eps = 1e-8, moments initialised to 0, bias correction with
  beta^(t+1), convergence `|g|^2 < (y + 1) * 1e-8`, projection onto the bounds after every step
  (`Adam::minimize`, optimizer.hpp:127-160) -- mirrored operation by operation.

`evaluate(x) -> (fn, gr, sum_eff, n_skipped)` is an `api.Engine.train_eval`, a `distributed.ShardedTrainer`, or any
callable with that signature (tests pass the oracle).
"""
import math

import numpy as np


def regularisation(n_param, rho_theta, rho_lambda):
    """rho vector in pack_params order: theta rows flattened, then lambda_0, lambda_1."""
    return np.r_[np.full(n_param - 2, float(rho_theta)), np.full(2, float(rho_lambda))]


def bounds(n_param, x0=None, vary=None):
    """(lower, upper) of every parameter: theta free, lambda >= 0.  `vary` (--param-set, motif_mask_trainer.hpp:66-103):
    indices of the parameters to fit; the others are fixed at x0 by equal bounds."""
    b = [(None, None)] * (n_param - 2) + [(0.0, None)] * 2
    if vary is not None:
        keep = set(int(i) for i in vary)
        b = [b[i] if i in keep else (float(x0[i]), float(x0[i])) for i in range(n_param)]
    return b


class Objective:
    """fn + L2 term as the reference's optimizers see it; keeps the trace and the best point."""

    def __init__(self, evaluate, rho, log=None):
        self.evaluate, self.rho, self.log = evaluate, np.asarray(rho, dtype=np.float64), log
        self.n_eval = 0
        self.best_f, self.best_x = math.inf, None
        self.trace = []          # (n_eval, regularised f, |gr|^2 of the unregularised gradient, sum_eff, fn)

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        fn, gr, eff, nsk = self.evaluate(x)
        y = fn + float(np.sum(self.rho * x * x) / 2.0)
        g = np.asarray(gr, dtype=np.float64) + self.rho * x
        self.n_eval += 1
        self.trace.append((self.n_eval, y, float(np.dot(gr, gr)), eff, fn))
        if y < self.best_f:
            self.best_f, self.best_x = y, x.copy()
        if self.log:
            self.log("eval %d: f %.6g (fn %.6g), |gr|^2 %.6g, considered BP %.6g, skipped %d" % (self.n_eval, y, fn, np.dot(gr, gr), eff, nsk))
        return y, g


def minimize_lbfgsb(evaluate, x0, rho, max_iter=300, epsilon=1e-3, log=None, vary=None):
    """`elem train --no-shuffle`: returns dict(x, f, n_iter, n_eval, trace, iter_f, message)."""
    from scipy.optimize import fmin_l_bfgs_b
    obj = Objective(evaluate, rho, log)
    iter_f = []

    def on_iter(xk):
        iter_f.append(obj.trace[-1][1])

    x, f, info = fmin_l_bfgs_b(obj, np.asarray(x0, dtype=np.float64), bounds=bounds(len(x0), x0, vary), m=5, factr=1e7, pgtol=epsilon,
                               maxiter=max_iter, maxfun=20 * max_iter + 20, callback=on_iter)
    return dict(x=obj.best_x, f=obj.best_f, n_iter=info["nit"], n_eval=obj.n_eval, trace=obj.trace, iter_f=iter_f,
                message=info["task"] if isinstance(info["task"], str) else info["task"].decode())


def minimize_adam(evaluate, x0, rho, max_iter=100, alpha=0.1, beta1=0.9, beta2=0.999, eps=1e-8, log=None, vary=None):
    """`elem train` with shuffled negatives / mini-batches (optimizer.hpp:127-160), operation by operation.  `vary`
    (--param-set): the other parameters are pinned to x0 by bounds of type 3, which Adam::after_update clips to after every
    step (set_mask_bounds, motif_mask_trainer.hpp:66-108)."""
    obj = Objective(evaluate, rho, log)
    x = np.asarray(x0, dtype=np.float64).copy()
    n = len(x)
    lower = np.r_[np.full(n - 2, -np.inf), 0.0, 0.0]
    upper = np.full(n, np.inf)
    if vary is not None:
        fixed = np.ones(n, dtype=bool)
        fixed[[int(i) for i in vary]] = False
        lower[fixed] = upper[fixed] = x[fixed]
    m, v = np.zeros(n), np.zeros(n)
    b1t, b2t = beta1, beta2
    t = 0
    while True:
        t += 1
        y, g = obj(x)
        b1t *= beta1
        b2t *= beta2
        m += (1.0 - beta1) * (g - m)
        v += (1.0 - beta2) * (g * g - v)
        x = x - alpha * (m / (1.0 - b1t)) / (np.sqrt(v / (1.0 - b2t)) + eps)
        x = np.minimum(np.maximum(x, lower), upper)
        if float(np.dot(g, g)) < (y + 1.0) * 1e-8 or t >= max_iter:
            break
    return dict(x=x, f=obj.trace[-1][1], n_iter=t - 1, n_eval=obj.n_eval, trace=obj.trace, message="adam")


class ShuffledNegatives:
    """`elem train` without --no-shuffle (motif_trainer.hpp:145-152, 228-245): every evaluation pairs each sequence with a
    k-let shuffled copy of itself (uShuffle seeded by srand(#occurrences of the first base + evaluation count)) that is
    treated as "no motif" with all position weights 0; a negative is dropped when its positive was skipped.

    evaluate_pos(x) evaluates the resident positives; evaluate_batch(seqs, quals, x) loads and evaluates another batch
    (a second engine on the GPU, or the oracle in tests); skipped_pos() -> bool array after evaluate_pos."""

    def __init__(self, seqs, evaluate_pos, skipped_pos, evaluate_batch, k=2):
        from .api import kmer_shuffle
        self._shuffle = kmer_shuffle
        self.seqs, self.k, self.count = seqs, k, 0
        self.evaluate_pos, self.skipped_pos, self.evaluate_batch = evaluate_pos, skipped_pos, evaluate_batch

    def __call__(self, x):
        fn, gr, eff, nsk = self.evaluate_pos(x)
        skipped = self.skipped_pos()
        negs = [self._shuffle(s, self.k, self.count) for s, sk in zip(self.seqs, skipped) if not sk]
        self.count += 1
        if negs:
            quals = [np.r_[np.zeros(len(s), dtype=np.uint8), np.uint8(1)] for s in negs]   # ws = 0, label "no motif"
            fn2, gr2, _, nsk2 = self.evaluate_batch(negs, quals, x)
            fn, gr, nsk = fn + fn2, np.asarray(gr) + np.asarray(gr2), nsk + nsk2
        return fn, gr, eff, nsk


class MiniBatches:
    """`elem train --batch-size N` (FastqBatchReader, fastq_io.hpp:132-167; RNAelemTrainer::operator(), motif_trainer.hpp:595-632):
    every evaluation reads the next N records of the current epoch order; a rest of fewer than N records is dropped; at
    the end of an epoch the order is shuffled with std::shuffle(mt19937(epoch)) on top of the previous order.  With
    `kmer_shuf` every record also gets its shuffled negative (seed = evaluation count, as in ShuffledNegatives).

    evaluate_batch(seqs, quals, x) -> (fn, gr, sum_eff, n_skipped, skipped flags per sequence); or `pairs`: a
    distributed.ShardedPairs that evaluates a batch and its negatives over several ranks (then evaluate_batch is unused);
    `evaluate_joint(seqs, quals, x, n_records) -> (fn, gr, sum_eff of the records, n_skipped, skipped flags)`: optional,
    one evaluation of records + negatives together; `engines`: two api.Engine objects for the same (double-buffered:
    the next batch loads on one while the other evaluates; then the callables are unused)."""

    def __init__(self, seqs, quals, batch, evaluate_batch, kmer_shuf=None, pairs=None, evaluate_joint=None, engines=None, lookahead=8):
        from .api import epoch_permutation, kmer_shuffle
        self._perm, self._shuffle = epoch_permutation, kmer_shuffle
        self.seqs, self.quals, self.batch = seqs, quals, batch if batch > 0 else len(seqs)
        self.evaluate_batch, self.k, self.pairs, self.evaluate_joint = evaluate_batch, kmer_shuf, pairs, evaluate_joint
        self.order, self.pos, self.n_shuffles, self.count = list(range(len(seqs))), 0, 0, 0
        # `engines` = two api.Engine objects: the batch of the NEXT evaluation (records + negatives: neither depends on x) is
        # loaded on one of them by a thread while the other evaluates the current one
        # `lookahead`: the batches of that many coming evaluations are loaded as ONE batch (neither the reader's order nor the
        # negatives depend on x; a load of 128 sequences is bound by launch latencies and costs about as much as one of 1024) and
        # evaluated range by range (engine options eval_first / eval_count)
        self.engines = list(engines) if engines else None
        self.lookahead = max(1, int(lookahead))
        if self.engines:
            # the joint batch must stay RESIDENT on its engine (a streamed batch, or one on the log-space pipeline, is evaluated as
            # a whole -- the engine refuses a range there): no more than ~2048 sequences at a time, negatives included
            per_eval = self.batch * (2 if kmer_shuf is not None else 1)
            self.lookahead = max(1, min(self.lookahead, 2048 // max(per_eval, 1)))
        self._pending = None
        self._current = None

    def _next_batch(self):
        """advances the reader: (records, qualities, evaluation count) of the next evaluation"""
        n = len(self.seqs)
        if n - self.pos < self.batch:
            self.pos = n                                   # the partial last batch is skipped
        if self.pos == n:                                  # end of the epoch: reshuffle, rewind
            perm = self._perm(n, self.n_shuffles)
            self.order = [self.order[p] for p in perm]
            self.n_shuffles += 1
            self.pos = 0
        idx = self.order[self.pos:self.pos + self.batch]
        self.pos += self.batch
        c = self.count
        self.count += 1
        return [self.seqs[i] for i in idx], [self.quals[i] for i in idx], c

    def _with_negatives(self, s1, q1, c):
        negs = [self._shuffle(s, self.k, c) for s in s1]
        return s1 + negs, q1 + [np.r_[np.zeros(len(s), dtype=np.uint8), np.uint8(1)] for s in negs]

    def _prefetch(self, slot):
        """the next evaluation's batch: reader, negatives and load_batch, all on a thread of their own (the reader and the
        shuffles draw from the process-global rand() of the reference: one thread at a time, in the reference's order)"""
        import threading
        eng = self.engines[slot]
        box = {}

        def work():
            try:
                parts, sa_all, qa_all = [], [], []
                for _ in range(self.lookahead):
                    s1, q1, c = self._next_batch()
                    sa, qa = self._with_negatives(s1, q1, c) if self.k is not None else (s1, q1)
                    parts.append(dict(s1=s1, q1=q1, c=c, first=len(sa_all), count=len(sa)))
                    sa_all += sa
                    qa_all += qa
                box.update(parts=parts, sa_all=sa_all, qa_all=qa_all)
                eng.load_batch(sa_all, qa_all)
            except Exception as e:      # re-raised by the evaluation that needs the batch
                box["error"] = e

        th = threading.Thread(target=work)
        th.start()
        return dict(thread=th, box=box, slot=slot)

    def finish(self):
        """waits for the batch that was prefetched for an evaluation that never came (call before the engines go away)"""
        if self._pending is not None:
            self._pending["thread"].join()
            self._pending = None
        self._current = None      # (its remaining parts belong to a batch that may be gone with the engines)

    def _call_prefetched(self, x):
        if self._current is None or not self._current["parts"]:
            if self._pending is None:
                self._pending = self._prefetch(0)
            cur = self._pending
            cur["thread"].join()
            if "error" in cur["box"]:
                raise cur["box"]["error"]
            cur.update(cur["box"])
            self._current = cur
            self._pending = self._prefetch(1 - cur["slot"])   # the next batches load while these are evaluated
        cur = self._current
        part = cur["parts"].pop(0)
        eng, n_rec, first, count = self.engines[cur["slot"]], len(part["s1"]), part["first"], part["count"]
        eng.set_option("eval_first", first)
        eng.set_option("eval_count", count)
        fn, gr, eff, nsk = eng.train_eval(x)
        skipped = eng.seq_stats()[first:first + count, 4] != 0
        if self.k is None:
            return fn, gr, eff, nsk
        if not np.any(skipped[:n_rec]):
            return fn, gr, float(eng.bpp_eff()[first:first + n_rec].sum()), nsk
        # a record was skipped: its negative must be left out -- the two-step evaluation on the same engine, whose resident
        # batches are put back afterwards (rare: a partition function outside the double range)
        if self._pending is not None:
            self._pending["thread"].join()   # (the shuffles below and the prefetch thread's share one rand())
        eng.load_batch(part["s1"], part["q1"])
        fn, gr, eff, nsk = eng.train_eval(x)
        sk = eng.seq_stats()[:, 4] != 0
        negs = [self._shuffle(s, self.k, part["c"]) for s, dead in zip(part["s1"], sk) if not dead]
        if negs:
            eng.load_batch(negs, [np.r_[np.zeros(len(s), dtype=np.uint8), np.uint8(1)] for s in negs])
            fn2, gr2, _, nsk2 = eng.train_eval(x)
            fn, gr, nsk = fn + fn2, np.asarray(gr) + np.asarray(gr2), nsk + nsk2
        if cur["parts"]:
            eng.load_batch(cur["sa_all"], cur["qa_all"])
        return fn, gr, eff, nsk

    def __call__(self, x):
        if self.engines is not None:
            return self._call_prefetched(x)
        s1, q1, c = self._next_batch()
        if self.pairs is not None:
            self.pairs.load(s1, q1)
            return self.pairs(x, c)
        if self.k is not None and self.evaluate_joint is not None:
            # records and their negatives as ONE batch (a mini-batch evaluation is bound by launch latencies, not by the
            # number of sequences: half the loads and evaluations).  A negative belongs in the sum only if its record was
            # not skipped (non-finite Z: rare) -- then the two-step evaluation below is the answer.
            sa, qa = self._with_negatives(s1, q1, c)
            fn, gr, eff, nsk, skipped = self.evaluate_joint(sa, qa, x, len(s1))
            if not np.any(skipped[:len(s1)]):
                return fn, gr, eff, nsk
        fn, gr, eff, nsk, skipped = self.evaluate_batch(s1, q1, x)
        if self.k is not None:
            negs = [self._shuffle(s, self.k, c) for s, sk in zip(s1, skipped) if not sk]
            if negs:
                quals = [np.r_[np.zeros(len(s), dtype=np.uint8), np.uint8(1)] for s in negs]
                fn2, gr2, _, nsk2, _ = self.evaluate_batch(negs, quals, x)
                fn, gr, nsk = fn + fn2, np.asarray(gr) + np.asarray(gr2), nsk + nsk2
        return fn, gr, eff, nsk


def train(evaluate, x0, rho_theta=0.1, rho_lambda=0.1, max_iter=300, epsilon=1e-3, optimizer="lbfgsb", log=None, vary=None):
    rho = regularisation(len(x0), rho_theta, rho_lambda)
    if vary is not None:   # only the fitted parameters are regularised (set_mask_regularization, motif_mask_trainer.hpp:36-64)
        mask = np.zeros(len(x0))
        mask[[int(i) for i in vary]] = 1.0
        rho = rho * mask
    if optimizer == "lbfgsb":
        return minimize_lbfgsb(evaluate, x0, rho, max_iter, epsilon, log, vary)
    if optimizer == "adam":
        return minimize_adam(evaluate, x0, rho, max_iter, log=log, vary=vary)
    raise ValueError("optimizer must be 'lbfgsb' or 'adam'")
