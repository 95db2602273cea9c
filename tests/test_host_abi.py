"""CPU-side checks of the product: the C-ABI library builds, loads and exports every symbol that
include/elemdp.h declares; the host-only entry points (automaton, x0, gradient assembly) work without a
GPU; every computing entry point fails loudly with ELEMDP_ENODEV instead of falling back to a CPU path;
the multi-rank reduction path is exercised with world_size-2 `gloo` on the CPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from rnaelem_amd import api, io, synth
from rnaelem_amd.distributed import assigned_range
from tests.util import REPO, gload, gpath

HEADER = os.path.join(REPO, "include", "elemdp.h")


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    declared = set(re.findall(r"\b(elemdp_[a-z_0-9]+)\s*\(", open(HEADER).read()))
    declared.discard("elemdp_handle")
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(api.SYMBOLS) == declared
    assert lib.elemdp_abi_version() == 1


def test_product_never_links_or_imports_the_oracle():
    """The oracle is test infrastructure: nothing under rnaelem_amd/ may mention it."""
    for root, _, files in os.walk(os.path.join(REPO, "rnaelem_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                for line in open(os.path.join(root, f), errors="ignore"):
                    code = line.strip()
                    if code.startswith(("#include", "import ", "from ")) or "CDLL" in code or "dlopen" in code:
                        assert "oracle" not in code and "emul" not in code, (f, code)
    out = subprocess.run(["ldd", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "emul" not in out


@pytest.mark.parametrize("pattern", sorted(gload("hmm.json")))
def test_describe_matches_reference_automaton(pattern):
    ref = gload("hmm.json")[pattern]
    got = api.Engine(pattern).describe()
    for k in ("reg_pattern", "M", "S", "node", "theta_id", "theta_sizes", "state", "loop_state", "right", "left", "pair",
              "loop_loop"):
        assert got[k] == ref[k], (pattern, k)


def test_initial_params_equal_reference_x0():
    for c in gload("eval.json"):
        if c["model"] in ("trna_x0.model", "syn_x0.model"):
            m = io.read_model(gpath(c["model"]))
            eng = io.engine_from_model(m)
            assert np.array_equal(eng.initial_params(0.0), np.array(c["x"]))
    e = api.Engine("(.*)", flags=api.THETA_SOFTMAX)
    assert np.array_equal(e.initial_params(0.3), np.r_[np.zeros(e.n_param - 2), 0.3, 0.3])


def test_error_codes_without_compute():
    with pytest.raises(api.ElemdpError) as e:
        api.Engine("(.")
    assert e.value.code == -1
    with pytest.raises(api.ElemdpError):
        api.Engine("(.)", flags=api.NO_RSS)        # pairs are not allowed with --no-rss
    with pytest.raises(api.ElemdpError):
        api.Engine("..", flags=api.NO_RSS | api.NO_PROFILE)
    with pytest.raises(api.ElemdpError):
        api.Engine("(.)", energy_param="# stack\nnot numbers at all\n# Triloops\nCAACGAAAAA 1\n")


@pytest.mark.skipif(have_gpu(), reason="only meaningful on a machine without a GPU")
def test_compute_entry_points_refuse_to_run_without_a_gpu():
    eng = api.Engine("((.*.))")
    s, q = synth.synth_batch(2, 30)
    with pytest.raises(api.ElemdpError) as e:
        eng.load_batch(s, q)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_train_finish_assembles_the_reference_gradient():
    """elemdp_train_finish == motif_trainer.hpp:248-271 incl. the softmax chain rule (host only)."""
    for flags in (0, api.THETA_SOFTMAX):
        eng = api.Engine("(.*)", flags=flags)
        nt = eng.n_param - 2
        rng = np.random.RandomState(3)
        x = rng.randn(eng.n_param)
        ENo, ENx, EH = rng.rand(nt), rng.rand(nt), rng.rand(4)
        red = np.r_[1.25, 0.5, 7, 1, ENo, ENx, EH]
        fn, gr, eff, nsk = eng.train_finish(red, x=x)
        assert (fn, eff, nsk) == (1.25, 0.5, 1)
        d = ENo - ENx
        if flags:
            want, k = [], 0
            for w in eng.describe()["theta_sizes"]:
                s = x[k:k + w]
                p = np.exp(s - np.logaddexp.reduce(s))
                tot = d[k:k + w].sum()
                want += list((1 - p) * d[k:k + w] - p * (tot - d[k:k + w]))
                k += w
            want = np.array(want)
        else:
            want = d
        np.testing.assert_allclose(gr[:-2], want, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(gr[-2:], [EH[0] - EH[2], EH[1] - EH[3]], rtol=1e-15)


def test_assigned_range_matches_reference_array_job_split():
    # arrayjob_manager.hpp:143-151 : the first `total mod n` parts get one extra element
    assert [assigned_range(10, 3, k) for k in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [assigned_range(76, 8, k)[1] - assigned_range(76, 8, k)[0] for k in range(8)] == [10, 10, 10, 10, 9, 9, 9, 9]
    for total, n in ((10000, 8), (7, 8), (1, 1)):
        r = [assigned_range(total, n, k) for k in range(n)]
        assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from oracle import pyoracle as po
from rnaelem_amd import io
from rnaelem_amd.distributed import assigned_range, reduce_and_finish
from tests.emul.pyemul import Emul
from tests.util import gpath

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
m = io.read_model(gpath("syn_sm.model"))
recs = io.read_fastq(gpath("syn_L40_n3.fq")) + io.read_fastq(gpath("syn_L100_n3.fq"))[:2]
a, b = assigned_range(len(recs), world, rank)
# this rank's partial sums [fn, sum_eff, n_used, n_skipped, ENo, ENx, EHo, EHx] (here from the CPU emulation of the
# kernels' rule code; on the GPU box elemdp_train_partial produces the same vector on the device)
emu = Emul(m["pattern"], open(po.DEFAULT_PAR).read(), m["max_span"], m["max_iloop"], m["min_bpp"], m["tau"], m["flags"])
nt = emu.n_param - 2
part = np.zeros(4 + 2 * nt + 4)
for rid, s, q in recs[a:b]:
    r = emu.train_seq(m["x"], s, q)
    if r["skipped"]:
        part[3] += 1
        continue
    part[0] += r["f"]; part[1] += r["bpp_eff"]; part[2] += 1
    part[4:4 + nt] += r["ENo"]; part[4 + nt:4 + 2 * nt] += r["ENx"]
    part[4 + 2 * nt:4 + 2 * nt + 2] += r["EHo"]; part[4 + 2 * nt + 2:] += r["EHx"]
eng = io.engine_from_model(m)          # host-only handle: finish runs through the C ABI
fn, gr, eff, nsk = reduce_and_finish(eng, part, m["x"])
if rank == 0:
    o, x = po.oracle_from_model(gpath("syn_sm.model"))
    fo, go, eo, no = o.train_eval(x, [s for _, s, _ in recs], [q for _, _, q in recs])
    assert abs(fn - fo) <= 1e-10 * max(1, abs(fo)), (fn, fo)
    assert np.allclose(gr, go, rtol=1e-8, atol=1e-10), np.abs(gr - go).max()
    assert abs(eff - eo) < 1e-12 and nsk == no
    print("OK", fn)
dist.destroy_process_group()
'''


def test_two_rank_gloo_all_reduce_path(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


PAIRS_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from oracle import pyoracle as po
from rnaelem_amd import api, io, train
from rnaelem_amd.distributed import ShardedPairs, ShardedShuffledNegatives
from tests.emul.pyemul import Emul
from tests.util import gload, gpath

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = gload(sys.argv[2])
recs = io.read_fastq(gpath(t["fq"]))
seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
emu = Emul(t["pattern"], open(po.DEFAULT_PAR).read(), 50, 30, 1e-4, t["tau"], 0)
eng = api.Engine(t["pattern"], "~T2004~", 50, 30, 1e-4, t["tau"])       # host-only handle: x0 and finish through the C ABI
nt = emu.n_param - 2
state = {}

def partial_of(s2, q2, x):
    """what elemdp_train_partial returns for a resident batch (here from the CPU emulation of the kernels' rule code)"""
    part, skipped = np.zeros(4 + 2 * nt + 4), []
    for s, q in zip(s2, q2):
        r = emu.train_seq(x, s, q)
        skipped.append(bool(r["skipped"]))
        if r["skipped"]:
            part[3] += 1
            continue
        part[0] += r["f"]; part[1] += r["bpp_eff"]; part[2] += 1
        part[4:4 + nt] += r["ENo"]; part[4 + nt:4 + 2 * nt] += r["ENx"]
        part[4 + 2 * nt:4 + 2 * nt + 2] += r["EHo"]; part[4 + 2 * nt + 2:] += r["EHx"]
    return part, np.array(skipped, dtype=bool)

def load(s2, q2):
    state["batch"] = (s2, q2)

def partial_pos(x):
    part, state["skipped"] = partial_of(*state["batch"], x)
    return part

pairs = ShardedPairs(load, partial_pos, lambda: state["skipped"], lambda s2, q2, x: partial_of(s2, q2, x)[0],
                     lambda total, x: eng.train_finish(total, x=x), 4 + 2 * nt + 4, rank, world, t["kmer_shuf"])
if "batch_size" in t:
    ev = train.MiniBatches(seqs, quals, t["batch_size"], None, pairs=pairs)
else:
    ev = ShardedShuffledNegatives(pairs, seqs, quals)
x0 = eng.initial_params(t["lambda_init"])
r = train.minimize_adam(ev, x0, train.regularisation(len(x0), t["rho_theta"], t["rho_lambda"]), max_iter=t["max_iter"])
fn = [row[4] for row in r["trace"]]
assert len(fn) == len(t["iter_fn"]), (fn, t["iter_fn"])
for a, b in zip(fn, t["iter_fn"]):
    assert abs(a - b) <= 2e-5 * abs(b), (fn, t["iter_fn"])
print("OK rank", rank, fn[-1])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("trace,port", [("train_trace_shuffle.json", "29573"), ("train_trace_minibatch.json", "29575")])
def test_two_rank_default_training_modes_reproduce_the_reference_traces(tmp_path, trace, port):
    """Shuffled negatives and mini-batches over two ranks (distributed.ShardedPairs, gloo): records and the negatives of a
    rank's own records are evaluated per rank, one all-reduce per evaluation; the data term of every evaluation equals the
    trace of the reference binary (one process), whatever the sharding."""
    script = tmp_path / "worker.py"
    script.write_text(PAIRS_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, trace], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK rank" in o for o in outs)
