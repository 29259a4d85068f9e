#!/usr/bin/env python3
"""Writes tests/golden/sobol_reference.json from the reference's Sobol' tables (run in the build container, where
/root/reference exists; the fixture travels, the reference does not):
  * SHA-256 of SobolMatrices32 and SobolMatrices64 (1024 x 52 entries, little-endian) — what the host's regenerated
    matrices (csrc/host/sobol.cpp from the Joe-Kuo parameters) must hash to;
  * VdCSobolMatrices / VdCSobolMatricesInv rows for log2 resolutions 1 .. 16 — what its GF(2) algebra must produce;
  * known answers: SobolSampleFloat(a, dim) and SobolIntervalToIndex(m, frame, p) evaluated here on the reference's own
    tables with the formulas of src/core/lowdiscrepancy.h:229-274 (integer XORs and one float32 multiply).
"""
import hashlib, json, os, re, struct
import numpy as np

SRC = "/root/reference/src/core/sobolmatrices.cpp"
text = open(SRC).read()


def flat(name):
    i = text.index(name); i = text.index("{", i)
    depth, j = 0, i
    while True:
        depth += text[j] == "{"; depth -= text[j] == "}"
        if depth == 0: break
        j += 1
    return [int(x.rstrip("ULul"), 16) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", text[i:j + 1])]


def rows(name):
    i = text.index(name); i = text.index("=", i); i = text.index("{", i)
    depth, j = 0, i
    while True:
        depth += text[j] == "{"; depth -= text[j] == "}"
        if depth == 0: break
        j += 1
    return [[int(x.rstrip("ULul"), 16) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", r)] for r in re.findall(r"\{([^{}]*)\}", text[i + 1:j])]


m32, m64 = flat("SobolMatrices32[NumSobolDimensions"), flat("SobolMatrices64[NumSobolDimensions")
vdc, inv = rows("VdCSobolMatrices[]"), rows("VdCSobolMatricesInv[]")
assert len(m32) == len(m64) == 1024 * 52


def sample_float(a, dim):
    v, i = 0, dim * 52
    while a:
        if a & 1: v ^= m32[i]
        a >>= 1; i += 1
    return float(min(np.float32(v) * np.float32(2.0 ** -32), np.float32(1.0 - 2.0 ** -24)))


def interval_to_index(m, frame, px, py):
    if m == 0: return 0
    index, delta, c = frame << (2 * m), 0, 0
    while frame:
        if frame & 1: delta ^= vdc[m - 1][c]
        frame >>= 1; c += 1
    b, c = ((px << m) | py) ^ delta, 0
    while b:
        if b & 1: index ^= inv[m - 1][c]
        b >>= 1; c += 1
    return index


rng = np.random.default_rng(2008)
kat_s = [[int(a), int(d), sample_float(int(a), int(d))] for a, d in zip(rng.integers(0, 2 ** 32, 200), rng.integers(0, 1024, 200))]
kat_s += [[a, d, sample_float(a, d)] for a in (0, 1, 2, 3, 255, 256, 8191) for d in (0, 1, 2, 5, 40, 99, 1023)]
kat_i = []
for m in (1, 2, 5, 9, 10, 11, 16):
    for _ in range(12):
        frame = int(rng.integers(0, 2 ** max(1, min(10, 32 - 2 * m))))
        px, py = int(rng.integers(0, 2 ** m)), int(rng.integers(0, 2 ** m))
        kat_i.append([m, frame, px, py, interval_to_index(m, frame, px, py)])
out = {
    "source": "src/core/sobolmatrices.cpp of the reference (tables), src/core/lowdiscrepancy.h:229-274 (formulas)",
    "sha256_SobolMatrices32": hashlib.sha256(struct.pack(f"<{len(m32)}I", *m32)).hexdigest(),
    "sha256_SobolMatrices64": hashlib.sha256(struct.pack(f"<{len(m64)}Q", *m64)).hexdigest(),
    "VdCSobolMatrices": {str(m): [hex(x) for x in vdc[m - 1]] for m in range(1, 17)},
    "VdCSobolMatricesInv": {str(m): [hex(x) for x in inv[m - 1]] for m in range(1, 17)},
    "SobolSampleFloat": kat_s,
    "SobolIntervalToIndex": kat_i,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sobol_reference.json"), "w"), indent=0)
print("wrote sobol_reference.json:", len(kat_s), "samples,", len(kat_i), "indices; VdC row lengths", [len(r) for r in vdc[:4]], [len(r) for r in inv[:4]])
