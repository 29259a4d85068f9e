#!/usr/bin/env python3
"""Recover the Joe-Kuo direction-number parameters behind the reference's Sobol' generator matrices and write them as
data: pbrt-v3-iile_amd/csrc/host/sobol_params.inc (what libiile_host.so generates its matrices from).

Provenance. The reference's src/core/sobolmatrices.cpp (L. Gruenschloss, 2012) tabulates, for 1024 dimensions,
the 52 columns of the Sobol' generator matrices built from S. Joe and F. Y. Kuo, "Constructing Sobol sequences with
better two-dimensional projections", SIAM J. Sci. Comput. 30 (2008) — their published parameter file
new-joe-kuo-6.21201: per dimension a primitive polynomial over GF(2) (degree s, coefficient bits a) and initial
direction integers m_1..m_s. That file is not in the image and there is no network, but the parameters are determined
by the matrices: column k of dimension d is v_k = m_k << (52 - k) (k = 1..52), and for k > s

    m_k = 2 a_1 m_{k-1} ^ 4 a_2 m_{k-2} ^ ... ^ 2^{s-1} a_{s-1} m_{k-s+1} ^ 2^s m_{k-s} ^ m_{k-s}

(Bratley & Fox, Algorithm 659). This script reads the reference's table HERE (it is run in the build container only),
finds for every dimension the smallest (s, a) whose recurrence reproduces all 52 columns, checks that the matrices
regenerated from (s, a, m_1..m_s) equal the table bit for bit (32-bit and 52-bit forms), and writes the parameters:
16 KB of numbers instead of 27 000 lines of tables. Nothing of the reference's source text is kept.

    python3 tools/make_sobol_data.py [/root/reference/src/core/sobolmatrices.cpp]
"""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/core/sobolmatrices.cpp"
W = 52  # SobolMatrixSize


def parse_table(text, name):
    i = text.index(name)
    i = text.index("{", i)
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    body = text[i:j + 1]
    return [int(x.rstrip("ULul"), 16) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", body)]


def generate(s, a, m_init, n_cols=W):
    """Direction integers m_1..m_n of one dimension (dimension 0: s = 0, all m_k = 1)."""
    if s == 0:
        return [1] * n_cols
    m = list(m_init)
    for k in range(s, n_cols):
        v = m[k - s] ^ (m[k - s] << s)
        for i in range(1, s):
            if (a >> (s - 1 - i)) & 1:
                v ^= m[k - i] << i
        m.append(v)
    return m


def columns52(m):
    return [m[k] << (W - 1 - k) for k in range(len(m))]


def main():
    text = open(SRC).read()
    m64 = parse_table(text, "SobolMatrices64[NumSobolDimensions")
    m32 = parse_table(text, "SobolMatrices32[NumSobolDimensions")
    assert len(m64) == 1024 * W and len(m32) == 1024 * W, (len(m64), len(m32))
    params = []
    for d in range(1024):
        cols = m64[d * W:(d + 1) * W]
        m = []
        for k, v in enumerate(cols):
            sh = W - 1 - k
            assert v & ((1 << sh) - 1) == 0 and (v >> sh) & 1 == 1 and v >> sh < (1 << (k + 1)), (d, k, hex(v))
            m.append(v >> sh)
        found = None
        if all(x == 1 for x in m):
            found = (0, 0)
        else:
            for s in range(1, 20):
                for a in range(1 << max(s - 1, 0)):
                    if generate(s, a, m[:s]) == m:
                        found = (s, a)
                        break
                if found:
                    break
        assert found, f"dimension {d}: no recurrence of degree < 20 reproduces the table"
        s, a = found
        params.append((s, a, m[:s]))
        regen = columns52(generate(s, a, m[:s]))
        assert regen == cols
        assert [c >> (W - 32) for c in regen] == m32[d * W:(d + 1) * W], d
    out = os.path.join(REPO, "pbrt-v3-iile_amd", "csrc", "host", "sobol_params.inc")
    with open(out, "w") as f:
        f.write("// Joe-Kuo (2008, new-joe-kuo-6.21201) direction-number parameters of the first 1024 Sobol' dimensions:\n"
                "// per dimension {s, a, m_1 .. m_s} (degree and inner coefficient bits of the primitive polynomial, initial\n"
                "// direction integers). Data, recovered from the generator matrices by tools/make_sobol_data.py (provenance and\n"
                "// the recurrence are described there); sobol.cpp expands them into the matrices.\n")
        flat = []
        for s, a, mi in params:
            flat += [s, a] + mi
        f.write(f"static const int kSobolParamDims = {len(params)};\n")
        f.write(f"static const unsigned int kSobolParams[{len(flat)}] = {{\n")
        for i in range(0, len(flat), 24):
            f.write("    " + ", ".join(str(x) for x in flat[i:i + 24]) + ",\n")
        f.write("};\n")
    print(f"wrote {out}: {len(params)} dimensions, {len(flat)} numbers, max degree {max(p[0] for p in params)}")


if __name__ == "__main__":
    main()
