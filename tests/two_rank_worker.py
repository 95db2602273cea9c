"""Worker of tests/test_round3_gpu.py::test_in_library_collective_with_two_ranks: one rank of a two-GPU world that shards a batch
by assigned_range (arrayjob_manager.hpp:143-151), joins the in-library RCCL communicator (elemdp_comm_init) and evaluates.
args: rank world uid_file out_file"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rnaelem_amd import api, io                      # noqa: E402
from rnaelem_amd.distributed import assigned_range   # noqa: E402
from tests.util import gpath                          # noqa: E402

rank, world, uid_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
m = io.read_model(gpath("syn_b.model"))
recs = io.read_fastq(gpath("syn_L150_n8.fq"))
seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
eng = io.engine_from_model(m, device=rank)
a, b = assigned_range(len(seqs), world, rank)
if b > a:
    eng.load_batch(seqs[a:b], quals[a:b])
if rank == 0:
    uid = api.Engine.comm_unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_file + ".tmp", uid_file)          # (the other rank sees the whole id or nothing)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no communicator id after 120 s" % rank)
        time.sleep(0.05)
    uid = open(uid_file, "rb").read()
eng.comm_init(rank, world, uid)
fn, gr, eff, nsk = eng.train_eval(m["x"])
json.dump({"fn": fn, "gr": [float(v) for v in gr], "eff": eff, "nsk": int(nsk), "range": [a, b]}, open(out_file, "w"))
eng.comm_destroy()
