"""On-disk formats either side of the hot path.

  * FASTQ with pseudo-qualities: 4-line records, quality string of length L+1, Sanger offset 33, last
    character '!' <=> the sequence carries the motif        RNAelem/fastq_io.hpp:64-108
  * model text file                                          RNAelem/motif_io.hpp:29-57 (write), :118-262 (read)
  * 10-line scan record                                      RNAelem/motif_scanner.hpp:240-251
"""
import json
import math

import numpy as np

from . import api

_CODE = np.zeros(256, dtype=np.uint8)
for _c, _v in (("A", 1), ("a", 1), ("C", 2), ("c", 2), ("G", 3), ("g", 3), ("U", 4), ("u", 4), ("T", 4), ("t", 4)):
    _CODE[ord(_c)] = _v
NACGU = "NACGU"


def encode_seq(s):
    """bio_sequence.hpp:28-39: A,C,G,U/T -> 1..4, anything else -> 0."""
    return _CODE[np.frombuffer(s.encode(), dtype=np.uint8)].copy()


def decode_seq(codes):
    return "".join(NACGU[c] for c in codes)


def read_fastq(path, base=33):
    """-> list of (id line incl. '@', code array, quality array).  Like the reference's reader
    (fastq_io.hpp:85-105) a record counts only if all four of its lines are terminated."""
    with open(path) as f:
        lines = f.read().split("\n")
    recs = []
    k = 0
    while k + 4 < len(lines):   # the 4th line must be newline-terminated
        rid, s, _, q = lines[k:k + 4]
        recs.append((rid, encode_seq(s), (np.frombuffer(q.encode(), dtype=np.uint8).astype(np.int16) - base).astype(np.uint8)))
        k += 4
    return recs


REQUIRED = ["pattern", "theta|s", "ene-param", "max-span", "rho", "rho-lambda", "tau", "lambda", "min-bpp",
            "max-internal-loop", "theta-softmax"]


def read_model(path):
    """Parse a model file -> dict(pattern, rows, lam, flags, max_span, max_iloop, min_bpp, tau, rho..., ene_param)."""
    d = {}
    for line in open(path):
        line = line.rstrip("\n")
        p = line.split(": ")
        if len(p) < 2:
            continue
        if len(p) != 2:
            raise ValueError("fail to parse: %s" % path)
        d[p[0].strip()] = p[1].strip()
    have = set(d)
    missing = [k for k in REQUIRED if not ((k == "theta|s" and ({"theta", "s"} & have)) or
                                          (k == "rho" and ({"rho-theta", "rho-s"} & have)) or k in have)]
    if missing:
        raise ValueError("motif file broken: %s %s" % (path, missing))
    softmax = bool(int(d["theta-softmax"]))
    no_rss = bool(int(d.get("no-rss", "0")))
    pattern = d["pattern"].replace("_", ".") if no_rss else d["pattern"]
    rows = json.loads(d["s"] if "s" in d else d["theta"])
    m = dict(pattern=pattern, rows=rows, softmax=softmax, lam=json.loads(d["lambda"]), ene_param=d["ene-param"],
             max_span=int(d["max-span"]), max_iloop=int(d["max-internal-loop"]), min_bpp=float(d["min-bpp"]),
             tau=float(d["tau"]), rho_theta=float(d.get("rho-theta", 0)), rho_s=float(d.get("rho-s", 0)),
             rho_lambda=float(d["rho-lambda"]), lambda_prior=float(d.get("lambda-prior", 0)), no_rss=no_rss,
             no_prf=bool(int(d.get("no-profile", "0"))), no_ene=bool(int(d.get("no-energy", "0"))))
    m["flags"] = (api.NO_RSS if m["no_rss"] else 0) | (api.NO_PROFILE if m["no_prf"] else 0) | \
        (api.NO_ENERGY if m["no_ene"] else 0) | (api.THETA_SOFTMAX if softmax else 0)
    m["x"] = np.array([v for row in rows for v in row] + list(m["lam"]), dtype=np.float64)
    return m


def engine_from_model(m, device=-1):
    ene = m["ene_param"]
    par = ene if ene in ("~T2004~", "~A2007~") else open(ene).read()
    return api.Engine(m["pattern"], par, m["max_span"], m["max_iloop"], m["min_bpp"], m["tau"], m["flags"], device)


def fmt(v):
    """Default ostream formatting of a double (6 significant digits), as util.hpp:98-105 prints vectors."""
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if v == -math.inf:
        return "-inf"
    if v == math.inf:
        return "inf"
    if v != v:
        return "nan"
    return "%.6g" % v


def fmt_vec(v):
    return "[" + ",".join(fmt(x) for x in v) + "]"


def _log_softmax_rows(rows):
    out = []
    for r in rows:
        a = np.array(r, dtype=np.float64)
        mx = a.max()
        out.append(list(a - (mx + math.log(np.exp(a - mx).sum()))))
    return out


def write_model(path, m, x=None):
    """Model file in the reference's layout (RNAelemWriter::write)."""
    rows = m["rows"]
    lam = m["lam"]
    if x is not None:
        k, rows = 0, []
        for r in m["rows"]:
            rows.append(list(x[k:k + len(r)]))
            k += len(r)
        lam = list(x[k:k + 2])
    theta = _log_softmax_rows(rows) if m["softmax"] else rows
    pat = m["pattern"].replace(".", "_") if m["no_rss"] else m["pattern"]
    with open(path, "w") as f:
        f.write("pattern: %s\n" % pat)
        f.write("%s: [%s]\n" % ("s" if m["softmax"] else "theta", ",".join(fmt_vec(r) for r in rows)))
        f.write("exp-theta: [%s]\n" % ",".join(fmt_vec(np.exp(r)) for r in theta))
        f.write("ene-param: %s\nmax-span: %d\nmax-internal-loop: %d\n" % (m["ene_param"], m["max_span"], m["max_iloop"]))
        f.write("theta-softmax: %d\n" % int(m["softmax"]))
        f.write("%s: %s\n" % (("rho-s", fmt(m["rho_s"])) if m["softmax"] else ("rho-theta", fmt(m["rho_theta"]))))
        f.write("rho-lambda: %s\ntau: %s\nlambda: %s\nlambda-prior: %s\nmin-bpp: %s\n" % (
            fmt(m["rho_lambda"]), fmt(m["tau"]), fmt_vec(lam), fmt(m["lambda_prior"]), fmt(m["min_bpp"])))
        f.write("no-rss: %d\nno-profile: %d\nno-energy: %d\n" % (int(m["no_rss"]), int(m["no_prf"]), int(m["no_ene"])))


def scan_record(rid, codes, rec, nodes):
    """The 10-line record of `RNAelem scan` (motif_scanner.hpp:240-251)."""
    M = len(nodes)
    mot = "".join(" " if (h == 0 or h == M - 1) else nodes[h] for h in rec["psihat"])
    return "\n".join([
        "id: " + rid, "start: " + fmt_vec(rec["start"]), "end: " + fmt_vec(rec["end"]), "inner: " + fmt_vec(rec["inner"]),
        "psihat: " + fmt_vec([int(v) for v in rec["psihat"]]), "motif region: %d - %d" % (rec["Ys"], rec["Ye"]),
        "exist prob: " + fmt(rec["exist_prob"]), "seq: " + decode_seq(codes), "rss: " + rec["rss"], "mot: " + mot]) + "\n"
