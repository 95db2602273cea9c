#!/usr/bin/env python3
"""Canonicalise a ViennaRNA-2.0 energy parameter file into rnaelem_amd/data/*.elempar.

The engine needs the Turner-2004 nearest-neighbour parameters (published thermodynamic data,
distributed in ViennaRNA's `.par` format; the reference vendors them as a C string literal,
`RNAelem/rna_turner2004.par`, and string-includes them at `energy_param.hpp:647-673`).  The GPU box
has no access to /root/reference, so the parameter *data* has to ship with this repo.  This script
is the committed generator: it reads a ViennaRNA-format file (plain, or wrapped in C string-literal
quotes), keeps only the free-energy sections and only the numbers the reference's parser actually
consumes (`energy_param.hpp:159-183` get_array and the read_*dim shift/post ranges used at
`:519-640`), drops enthalpies and comments and writes a compact file in the same `# section`
syntax, so one parser (ours, `rnaelem_amd/csrc/energy_tables.cpp`) reads both the shipped file and
user-supplied ViennaRNA files.

usage:  tools/gen_energy_tables.py /root/reference/RNAelem/rna_turner2004.par \
            rnaelem_amd/data/turner2004.elempar
"""
import re
import sys

# section -> number of leading values the reference reads (None = special)
COUNTS = {
    "stack": 6 * 6,                      # read_2dim(7,7,1,1): 6 rows, first 6 of the 7 columns
    "mismatch_hairpin": 6 * 25,          # read_3dim(7,5,5,1,0,0)
    "mismatch_interior": 6 * 25,
    "mismatch_interior_1n": 6 * 25,
    "mismatch_interior_23": 6 * 25,
    "mismatch_multi": 7 * 25,            # read_3dim_smooth(8,5,5,1,0,0) (7th block lands out of bounds)
    "mismatch_exterior": 7 * 25,
    "dangle5": 7 * 5,                    # read_2dim_smooth(8,5,1,0)
    "dangle3": 7 * 5,
    "int11": 7 * 7 * 25,                 # read_4dim(8,8,5,5,1,1,0,0)
    "int21": 7 * 7 * 125,                # read_5dim(8,8,5,5,5,1,1,0,0,0)
    "int22": 6 * 6 * 256,                # read_6dim(... shifts 1, posts 1,1,0,0,0,0)
    "hairpin": 31,
    "bulge": 31,
    "interior": 31,
}
PER_LINE = {"stack": 6, "int22": 4, "hairpin": 10, "bulge": 10, "interior": 10}  # values per output line
ROWLEN = {"stack": 7}  # stack rows carry a 7th (NN) column that the reference skips


def unwrap(text):
    """Accept either a plain file or one whose lines are C string literals "....\\n"."""
    out = []
    for line in text.splitlines():
        s = line.strip()
        if s.startswith('"') and s.endswith('"'):
            s = s[1:-1]
            if s.endswith("\\n"):
                s = s[:-2]
            out.append(s)
        else:
            out.append(line.rstrip("\n"))
    return out


def numbers_of(line):
    """Words of a data line up to the first comment opener, like get_array()."""
    vals = []
    for w in line.split():
        if "/*" in w:
            break
        vals.append(w)
    return vals


def main(src, dst):
    lines = unwrap(open(src).read())
    sections = {}
    order = []
    i = 0
    while i < len(lines):
        m = re.match(r"^#\s+(\S+)", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        i += 1
        body = []
        while i < len(lines) and not lines[i].startswith("#"):
            body.append(lines[i])
            i += 1
        sections[name] = body
        order.append(name)

    out = ["## RNAfold parameter file v2.0 (elempar: free-energy sections only, canonicalised by",
           "## tools/gen_energy_tables.py; values in dcal/mol exactly as in the source file)", ""]
    for name in order:
        body = sections[name]
        if name in COUNTS:
            need = COUNTS[name]
            vals = []
            for ln in body:
                if len(ln) < 2:      # get_array stops at the first blank line
                    break
                w = numbers_of(ln)
                if name in ROWLEN:
                    w = w[: ROWLEN[name] - 1] if len(w) >= ROWLEN[name] else w
                vals.extend(w)
                if len(vals) >= need:
                    break
            vals = vals[:need]
            assert len(vals) == need, (name, len(vals), need)
            out.append("# " + name)
            per = PER_LINE.get(name, 5)
            for k in range(0, need, per):
                out.append(" ".join("%5s" % v for v in vals[k:k + per]))
            out.append("")
        elif name in ("ML_params", "NINIO", "Misc"):
            for ln in body:
                if ln == "":
                    break
                if "*" in ln:
                    continue
                out.append("# " + name)
                out.append(" ".join(ln.split()))
                out.append("")
                break
        elif name in ("Hexaloops", "Tetraloops", "Triloops"):
            out.append("# " + name)
            for ln in body:
                if ln == "":
                    break
                if "*" in ln:
                    continue
                w = ln.split()
                out.append("%s %s" % (w[0], w[1]))
            out.append("")
    out.append("#END")
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
