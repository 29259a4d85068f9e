"""SobolSampler: the host's generator matrices against the reference's tables (checksums and rows in
tests/golden/sobol_reference.json, written by tests/golden/make_sobol_fixture.py from the reference), the oracle's
restatement against known answers evaluated on the reference's own tables, and the reference's own unit tests of the
generator-matrix helpers (src/tests/sampling.cpp:75-138) re-run on the restatement. CPU only."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(REPO, "tests", "golden", "sobol_reference.json")))


def test_host_matrices_equal_the_reference_tables(binding):
    m32, m64 = binding.sobol_matrices(1024)
    assert hashlib.sha256(m32.astype("<u4").tobytes()).hexdigest() == FIX["sha256_SobolMatrices32"]
    assert hashlib.sha256(m64.astype("<u8").tobytes()).hexdigest() == FIX["sha256_SobolMatrices64"]


def test_host_vdc_matrices_equal_the_reference_tables(binding):
    for m in range(1, 17):
        vdc, inv = binding.sobol_vdc(m)
        want_v = [int(x, 16) for x in FIX["VdCSobolMatrices"][str(m)]]
        want_i = [int(x, 16) for x in FIX["VdCSobolMatricesInv"][str(m)]]
        assert len(want_i) == 2 * m
        assert vdc[:len(want_v)].tolist() == want_v, m
        assert inv[:2 * m].tolist() == want_i, m


def test_known_answers_of_the_reference_tables(binding, oracle):
    m32, m64 = binding.sobol_matrices(1024)
    for a, dim, want in FIX["SobolSampleFloat"]:
        got = oracle.lib.oracle_sobol_sample_float(m32.ctypes.data, a, dim, 0)
        assert np.float32(got) == np.float32(want), (a, dim)
    for m, frame, px, py, want in FIX["SobolIntervalToIndex"]:
        vdc, inv = binding.sobol_vdc(m)
        assert oracle.lib.oracle_sobol_interval_to_index(vdc.ctypes.data, inv.ctypes.data, m, frame, px, py) == want


def test_reference_test_LowDiscrepancy_Sobol(binding, oracle):
    """src/tests/sampling.cpp:118-133: float and double variants agree as floats; dimension 0 is the base-2 radical inverse."""
    m32, m64 = binding.sobol_matrices(100)
    lib = oracle.lib
    for i in range(256):
        for dim in range(100):
            f = np.float32(lib.oracle_sobol_sample_float(m32.ctypes.data, i, dim, 0))
            d = np.float32(lib.oracle_sobol_sample_double(m64.ctypes.data, i, dim, 0))
            assert f == d, (i, dim)
    for i in range(8192):
        assert np.float32(lib.oracle_sobol_sample_float(m32.ctypes.data, i, 0, 0)) == \
            np.float32(lib.oracle_reverse_bits32(i)) * np.float32(2.3283064365386963e-10)


def test_reference_test_LowDiscrepancy_GeneratorMatrix(oracle):
    """src/tests/sampling.cpp:75-104."""
    lib = oracle.lib
    C = np.array([1 << i for i in range(32)], np.uint32)
    Crev = np.array([lib.oracle_reverse_bits32(int(c)) for c in C], np.uint32)
    for a in range(128):
        assert lib.oracle_multiply_generator(C.ctypes.data, a) == a
        ri = np.float32(oracle.radical_inverse(0, a))
        assert ri == np.float32(lib.oracle_reverse_bits32(lib.oracle_multiply_generator(C.ctypes.data, a))) * np.float32(2.3283064365386963e-10)
        assert ri == np.float32(lib.oracle_sample_generator_matrix(Crev.ctypes.data, a, 0))
    rng = np.random.default_rng(7)  # "random / goofball generator matrix" (any matrix: the identity tested is linear algebra)
    C = rng.integers(0, 2 ** 32, 32, dtype=np.uint64).astype(np.uint32)
    Crev = np.array([lib.oracle_reverse_bits32(int(c)) for c in C], np.uint32)
    for a in range(1024):
        assert lib.oracle_reverse_bits32(lib.oracle_multiply_generator(C.ctypes.data, a)) == lib.oracle_multiply_generator(Crev.ctypes.data, a)


def test_reference_test_LowDiscrepancy_GrayCodeSample(oracle):
    """src/tests/sampling.cpp:106-116."""
    lib = oracle.lib
    C = np.array([1 << i for i in range(32)], np.uint32)
    v = np.zeros(64, np.float32)
    lib.oracle_gray_code_sample(C.ctypes.data, 64, 0, v.ctypes.data)
    for a in range(64):
        u = np.float32(lib.oracle_multiply_generator(C.ctypes.data, a)) * np.float32(2.3283064365386963e-10)
        assert u in v


def test_sobol_sampler_of_a_scene(binding, oracle):
    """SobolSampler as the scene's sampler: pixelsamples rounded up to a power of two (sobol.h:59-64), every sample of a
    pixel lands in that pixel (SobolIntervalToIndex), and the frame's samples are stratified: one per elementary
    interval of the pixel (the property src/tests/sampling.cpp:138-200 checks for the sampler's first 2D sample)."""
    scene = binding.HostScene(xres=96, yres=80, spp=6, sampler="sobol")
    assert scene.info["spp"] == 8
    for px, py in ((0, 0), (5, 7), (95, 79), (64, 33)):
        pts = []
        for k in range(8):
            idx = oracle.sample_index(scene, px, py, k)
            u = [float(oracle.sample_dimension(scene, idx, d, px, py)) for d in (0, 1)]
            assert 0 <= u[0] < 1 and 0 <= u[1] < 1
            # the unremapped sample really is inside the pixel: SobolSample * resolution floors to the pixel
            m32, _ = binding.sobol_matrices(2)
            for d, p in ((0, px), (1, py)):
                s = np.float32(oracle.lib.oracle_sobol_sample_float(m32.ctypes.data, idx, d, 0))
                assert int(np.floor(s * np.float32(128))) == p
            pts.append(u)
        for i in range(4):  # 8 samples: 2^i x 2^(3-i) elementary intervals hold one sample each
            nx, ny = 1 << i, 1 << (3 - i)
            cells = {(int(u[0] * nx), int(u[1] * ny)) for u in pts}
            assert len(cells) == 8
    # a film rendered with it is a valid estimate of the same image
    film, _ = oracle.render(scene, threads=4)
    ref, _ = oracle.render(binding.HostScene(xres=96, yres=80, spp=8), threads=4)
    a, b = scene.film_to_rgb(film).mean(dtype=np.float64), scene.film_to_rgb(ref).mean(dtype=np.float64)
    assert abs(a - b) / b < 0.03


def test_sobol_index_limit_is_enforced(binding):
    with pytest.raises(RuntimeError, match="32-bit sample index"):
        binding.HostScene(xres=1920, yres=1080, spp=2048, sampler="sobol")
