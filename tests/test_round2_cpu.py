"""CPU-side parity of the rows SURVEY.md section 8 marks "next" (formats, eval, mask trainer, no-rss scan, config E shape) and of
the multi-rank plumbing (ordered scan output, bench launcher).  Fixtures: tests/golden/make_golden_r2.py (outputs of the
compiled reference, oracle/_ref)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from rnaelem_amd import api, cli, io, train
from tests.emul.pyemul import Emul
from tests.util import arr, assert_log_close, gload, gpath

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAR = open(po.DEFAULT_PAR).read()


def check_records(o, case, rtol=2e-5):
    byid = {r["id"]: r for r in case["records"]}
    nodes = o.hmm()["node"]
    n = 0
    for rid, seq, qual in po.read_fastq(gpath(case["fq"])):
        r, g = byid[rid], o.scan_seq(seq, qual)
        assert (g["Ys"], g["Ye"]) == (r["Ys"], r["Ye"])
        for k in ("start", "end", "inner"):
            ref, got = arr(r[k]), g[k]
            assert np.array_equal(np.isneginf(ref), np.isneginf(got)), k
            m = ~np.isneginf(ref)
            np.testing.assert_allclose(got[m], ref[m], rtol=rtol, atol=1e-300)
        assert g["exist_prob"] == pytest.approx(r["exist_prob"], rel=rtol)
        assert list(g["psihat"]) == r["psihat"] and g["rss"] == r["rss"]
        assert "".join(" " if (h == 0 or h == o.M - 1) else nodes[h] for h in g["psihat"]) == r["mot"]
        n += 1
    return n


def test_oracle_scan_of_the_config_e_shape_matches_the_reference_binary():
    """BASELINE config E: L = 300 scanned with a '(.....)' model (S = 29) -- records of `RNAelem scan`."""
    case = gload("scan_e.json")[0]
    o, x = po.oracle_from_model(gpath(case["model"]))
    assert o.S == 29
    assert check_records(o, case) == 4


def test_oracle_scan_in_no_rss_mode_matches_the_reference_binary():
    """motif_model.hpp:171-206 under the scanner functors (the reference's own no-rss model 2.model)"""
    case = gload("scan_norss.json")[0]
    o, x = po.oracle_from_model(gpath(case["model"]))
    assert check_records(o, case) == 2


@pytest.mark.parametrize("linear", [False, True])
@pytest.mark.parametrize("prune", [0, 1])
def test_product_rules_scan_in_no_rss_mode(linear, prune):
    """the kernels' rule headers with the pair mask cleared (what Engine::load_batch does under ELEMDP_NO_RSS)"""
    case = gload("scan_norss.json")[0]
    md = po.read_model(gpath(case["model"]))
    o, x = po.oracle_from_model(gpath(case["model"]))
    e = Emul(md["pattern"], PAR, md["max_span"], md["max_iloop"], md["min_bpp"], md["tau"], 1)
    e.set_prune(prune)
    for (rid, seq, qual), r in zip(po.read_fastq(gpath(case["fq"])), case["records"]):
        a, b = o.scan_seq(seq, qual), e.scan_seq(x, seq, qual, linear=linear)
        assert (b["Ys"], b["Ye"], b["rss"], list(b["psihat"])) == (r["Ys"], r["Ye"], r["rss"], r["psihat"])
        for k in ("start", "end", "inner"):
            assert_log_close(b[k], a[k], rtol=1e-9, atol=1e-9, what=k)


@pytest.mark.parametrize("raw,model,fq", [("scan_raw_0_0.raw", "0.model", "0.fq"),
                                          ("scan_raw_trna_a_positive_head6.raw", "trna_a.model", "positive_head6.fq")])
def test_scan_record_text_is_byte_identical_to_the_reference_output(raw, model, fq):
    """io.scan_record (what `cli scan` writes) against the raw text of `RNAelem scan -t 1` (motif_scanner.hpp:240-251,
    6-digit vectors util.hpp:98-105); the numbers come from the oracle."""
    o, x = po.oracle_from_model(gpath(model))
    nodes = o.hmm()["node"]
    recs = io.read_fastq(gpath(fq))
    text = ""
    for (rid, codes, q) in recs:
        g = o.scan_seq(codes, q)
        text += io.scan_record(rid, codes, g, nodes)
    assert text == open(gpath(raw)).read()


def test_model_writer_is_read_by_the_reference_reader():
    """written_trna_a.model was produced by io.write_model and read by RNAelemReader (motif_io.hpp:118-262): the writer still
    produces exactly that file, and the scan records the reference computed from it are those of our reading of it."""
    m = io.read_model(gpath("trna_a.model"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "w.model")
        io.write_model(p, m)
        assert open(p).read() == open(gpath("written_trna_a.model")).read()
    case = gload("scan_written_model.json")[0]
    o, x = po.oracle_from_model(gpath(case["model"]))
    np.testing.assert_array_equal(x, io.read_model(gpath(case["model"]))["x"])
    assert check_records(o, case) == 6


@pytest.mark.parametrize("name", ["ref_written.model", "ref_written_sm.model"])
def test_model_writer_reproduces_files_written_by_the_reference(name, tmp_path):
    """A model file written by the reference's own writer (RNAelemWriter::write), read and written back: identical bytes,
    except that `exp-theta:` (ignored by every reader) is recomputed from the 6-digit theta and may differ in its last digit."""
    m = io.read_model(gpath(name))
    p = str(tmp_path / "w.model")
    io.write_model(p, m)
    got, ref = open(p).read().split("\n"), open(gpath(name)).read().split("\n")
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        if b.startswith("exp-theta:"):
            va, vb = json.loads(a.split(": ", 1)[1]), json.loads(b.split(": ", 1)[1])
            for ra, rb in zip(va, vb):
                np.testing.assert_allclose(ra, rb, rtol=2e-5)
        else:
            assert a == b


def test_eval_text_has_the_reference_layout():
    """`eval` prints `fn: <17 digits>` to out1 and `gr: [..]` to out2 (motif_eval.hpp:46-47).  The shipped binary prints
    zeros there (it never hands a record to the evaluation): the fixture pins the LAYOUT, the numbers are the oracle's."""
    for c in gload("eval_text.json"):
        n = len(json.loads(c["out2"].split(": ", 1)[1]))
        l1, l2 = cli.eval_text(0.0, np.zeros(n))
        assert (l1, l2) == (c["out1"], c["out2"])
    l1, l2 = cli.eval_text(10.354626980956652, np.array([-2.4286081696701274, 0.1, -np.inf]))
    assert l1 == "fn: 10.354626980956652\n" and l2 == "gr: [-2.4286081696701274,0.10000000000000001,-inf]\n"


def test_param_set_spec_accepts_ranges():
    assert cli.parse_param_set("0-3,30,31") == [0, 1, 2, 3, 30, 31]
    assert cli.parse_param_set("4") == [4]


@pytest.mark.parametrize("k", [0, 1])
def test_mask_trainer_reproduces_the_reference_traces(k):
    """--param-set (motif_mask_trainer.hpp:28-111) with L-BFGS-B (--no-shuffle) and with Adam (default mode, shuffled
    negatives): the objective trace of the reference binary, and the parameters outside the set stay at x0."""
    t = gload("train_trace_mask.json")[k]
    recs = io.read_fastq(gpath(t["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    o = po.make_oracle(t["pattern"], 50, 30, min_bpp=1e-4, tau=0.1)
    vary = cli.parse_param_set(t["param_set"])
    x0 = api.Engine(t["pattern"]).initial_params(t["lambda_init"])
    fixed = [i for i in range(len(x0)) if i not in vary]
    if t["no_shuffle"]:
        def ev(x):
            o.set_params(x)
            return o.train_eval(x, seqs, quals, n_threads=4)
        r = train.train(ev, x0, 0.1, 0.1, max_iter=t["max_iter"], epsilon=t["epsilon"], vary=vary)
        got = r["iter_f"]
        y0 = r["trace"][0][1]
        assert y0 == pytest.approx(t["trace"][0], rel=2e-5)
        for a, b in zip(got[:4], t["trace"][1:5]):        # (L-BFGS-B 3.0 vs 2.1: the first iterations agree to the printed digits)
            assert a == pytest.approx(b, rel=2e-4), (got, t["trace"])
    else:
        state = {}

        def ev_pos(x):
            o.set_params(x)
            state["skipped"] = np.array([o.train_seq(s, q)["skipped"] for s, q in zip(seqs, quals)], dtype=bool)
            return o.train_eval(x, seqs, quals, n_threads=4)

        def ev_batch(s2, q2, x):
            o.set_params(x)
            return o.train_eval(x, s2, q2, n_threads=4)

        ev = train.ShuffledNegatives(seqs, ev_pos, lambda: state["skipped"], ev_batch, k=2)
        r = train.train(ev, x0, 0.1, 0.1, max_iter=t["max_iter"], optimizer="adam", vary=vary)
        fn = [row[4] for row in r["trace"]]
        assert len(fn) == len(t["trace"])
        for a, b in zip(fn, t["trace"]):
            assert a == pytest.approx(b, rel=2e-5), (fn, t["trace"])
        np.testing.assert_allclose(r["x"], t["final_x"], rtol=2e-5, atol=2e-6)
    assert np.array_equal(r["x"][fixed], x0[fixed])


SCAN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import pyoracle as po
from rnaelem_amd import cli, io
from tests.util import gpath
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
o, x = po.oracle_from_model(gpath("trna_a.model"))
nodes = o.hmm()["node"]
recs = io.read_fastq(gpath("positive_head6.fq"))[:int(sys.argv[3])]

def scan_part(mine):       # (the CPU stand-in of Engine.scan: records from the oracle)
    for rid, codes, q in mine:
        yield io.scan_record(rid, codes, o.scan_seq(codes, q), nodes)

cli.sharded_scan(recs, sys.argv[2], rank, world, scan_part, dist.barrier)
dist.destroy_process_group()
print("OK")
'''


@pytest.mark.parametrize("n_rec,world,port", [(6, 2, "29581"), (2, 3, "29583")])
def test_multi_rank_scan_output_is_joined_in_input_order(tmp_path, n_rec, world, port):
    """`cli scan` under torchrun: every rank scans its contiguous range (arrayjob_manager.hpp:143-151), rank 0 joins the parts
    in input order; more ranks than records leaves empty parts.  The joined file equals the reference's raw output."""
    script = tmp_path / "worker.py"
    script.write_text(SCAN_WORKER)
    out = str(tmp_path / "scan.raw")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), REPO, out, str(n_rec)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ref = open(gpath("scan_raw_trna_a_positive_head6.raw")).read()
    lines = ref.split("\n")
    assert open(out).read() == "".join(l + "\n" for l in lines[:10 * n_rec])       # (a record = 10 lines)
    assert [f for f in os.listdir(tmp_path) if f.startswith("scan.raw.")] == []


def test_bench_launcher_builds_the_torchrun_command_and_refuses_a_wrong_world():
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-7:] == [os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"]
    # a launch whose world does not match --gpus must fail loudly (exit code != 0, no JSON line)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout) and "{" not in r.stdout
    # the parent counts GPUs from the KFD topology (no torch / HIP call): nodes with SIMDs, capped by the visibility variables
    import tempfile
    with tempfile.TemporaryDirectory() as root:
        for k, simd in enumerate([0, 0, 256, 256, 256]):
            os.makedirs(os.path.join(root, str(k)))
            open(os.path.join(root, str(k), "properties"), "w").write("cpu_cores_count %d\nsimd_count %d\n" % (64 if not simd else 0, simd))
        keep = {v: os.environ.pop(v, None) for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES")}
        try:
            assert bench.count_gpus_sysfs(root) == 3
            os.environ["ROCR_VISIBLE_DEVICES"] = "0,2"
            assert bench.count_gpus_sysfs(root) == 2
        finally:
            os.environ.pop("ROCR_VISIBLE_DEVICES", None)
            for v, val in keep.items():
                if val is not None:
                    os.environ[v] = val
    assert bench.count_gpus_sysfs(os.path.join(REPO, "no-such-dir")) == 0
    # --gpus N on a machine with fewer GPUs: refused before anything is launched
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout) and "{" not in r.stdout
