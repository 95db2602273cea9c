"""Synthetic inputs for the benchmark / parity configs.

Sequences are i.i.d. uniform over ACGU with exact length L, drawn from splitmix64 seeded with
`20240807 + L` (two bits per base, high bits first); qualities are '+' * L + '!' -- exactly what the
reference's `script/kmer-psp.py:53-70` emits without a negative set (q = 10 everywhere, last char
'!' = "has motif").  Ids are "@0" .. "@N-1".
"""
import numpy as np

MASK = (1 << 64) - 1


def splitmix64_stream(seed, n):
    """n successive splitmix64 outputs as a uint64 array."""
    out = np.empty(n, dtype=np.uint64)
    x = seed & MASK
    for k in range(n):
        x = (x + 0x9E3779B97F4A7C15) & MASK
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        out[k] = z ^ (z >> 31)
    return out


def synth_codes(n_seq, L, seed=None):
    """(n_seq, L) uint8 array of base codes 1..4 (A,C,G,U)."""
    if seed is None:
        seed = 20240807 + L
    per = 32  # bases per 64-bit word
    words_per_seq = (L + per - 1) // per
    w = splitmix64_stream(seed, n_seq * words_per_seq).reshape(n_seq, words_per_seq)
    shifts = np.arange(62, -2, -2, dtype=np.uint64)
    bases = ((w[:, :, None] >> shifts[None, None, :]) & np.uint64(3)).astype(np.uint8) + 1
    return bases.reshape(n_seq, words_per_seq * per)[:, :L].copy()


def synth_batch(n_seq, L, seed=None, positive=True):
    """-> (seqs, quals): lists of uint8 arrays; qual has L+1 entries (char-33): 10..10, then 0 / 5."""
    codes = synth_codes(n_seq, L, seed)
    q = np.full(L + 1, 10, dtype=np.uint8)
    q[L] = 0 if positive else 5
    return [codes[i] for i in range(n_seq)], [q.copy() for _ in range(n_seq)]


def write_fastq(path, seqs, quals, ids=None):
    """4-line FASTQ with L+1 quality characters (reference input contract, fastq_io.hpp:64-108)."""
    nacgu = np.frombuffer(b"NACGU", dtype=np.uint8)
    with open(path, "w") as f:
        for n, (s, q) in enumerate(zip(seqs, quals)):
            rid = ids[n] if ids is not None else "@%d" % n
            f.write("%s\n%s\n+\n%s\n" % (rid, nacgu[np.asarray(s)].tobytes().decode(),
                                          (np.asarray(q, dtype=np.uint8) + 33).tobytes().decode()))


def fasta_to_fastq_records(path, base=10):
    """FASTA -> (id, seq string, qual string) like kmer-psp.py without a negative file."""
    recs = []
    ann, seq = None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if ann is not None:
                recs.append((ann, "".join(seq)))
            ann, seq = line, []
        else:
            seq.append(line.strip())
    if ann is not None:
        recs.append((ann, "".join(seq)))
    return [("@" + a[1:], s, chr(33 + base) * len(s) + "!") for a, s in recs]
