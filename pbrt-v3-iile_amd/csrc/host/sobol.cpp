// sobol.cpp — the Sobol' generator matrices of the reference's SobolSampler, built on the host like the Halton tables.
//
// The reference ships them as tables (src/core/sobolmatrices.cpp: SobolMatrices32 / SobolMatrices64, 1024 dimensions x 52
// columns, and VdCSobolMatrices / VdCSobolMatricesInv per resolution). They are functions of little data:
//  * dimension d's column k is v_k = m_k << (51 - k) in 52 bits (the 32-bit form keeps its upper 32 bits), with the
//    direction integers m_k from the Joe-Kuo parameters {s, a, m_1..m_s} (sobol_params.inc; Bratley & Fox recurrence);
//  * SobolIntervalToIndex (src/core/lowdiscrepancy.h:229-252) needs, for a film of 2^m x 2^m pixels, (i) the part of
//    (x << m | y) — the pixel the first two dimensions put a sample in, i.e. the upper m bits of each — that index bit
//    2m + c contributes (VdCSobolMatrices[m - 1][c]) and (ii) the inverse of the map from index bits 0 .. 2m - 1 to that
//    pixel vector (VdCSobolMatricesInv[m - 1][c]): linear algebra over GF(2) on the first two dimensions' matrices.
// tests/test_sobol.py holds both against checksums of the reference's tables (tests/golden/sobol_reference.json).
#include <cstdint>
#include <vector>

#include "host_scene.h"

namespace iile {

namespace {
#include "sobol_params.inc"
constexpr int kCols = 52;  // SobolMatrixSize

// direction integers m_1 .. m_52 of one dimension
bool direction_integers(int dim, uint64_t m[kCols]) {
    if (dim < 0 || dim >= kSobolParamDims) return false;
    const unsigned int *p = kSobolParams;
    for (int d = 0; d < dim; ++d) p += 2 + p[0];
    const int s = int(p[0]);
    const unsigned int a = p[1];
    if (s == 0) {  // dimension 0: van der Corput
        for (int k = 0; k < kCols; ++k) m[k] = 1;
        return true;
    }
    for (int k = 0; k < s; ++k) m[k] = p[2 + k];
    for (int k = s; k < kCols; ++k) {
        uint64_t v = m[k - s] ^ (m[k - s] << s);
        for (int i = 1; i < s; ++i)
            if ((a >> (s - 1 - i)) & 1u) v ^= m[k - i] << i;
        m[k] = v;
    }
    return true;
}
}  // namespace

int sobol_num_dimensions() { return kSobolParamDims; }

// SobolMatrices64[dim * 52 + k]
bool sobol_columns64(int dim, uint64_t cols[52]) {
    uint64_t m[kCols];
    if (!direction_integers(dim, m)) return false;
    for (int k = 0; k < kCols; ++k) cols[k] = m[k] << (kCols - 1 - k);
    return true;
}
// SobolMatrices32[dim * 52 + k]: the upper 32 of the 52 bits
bool sobol_columns32(int dim, uint32_t cols[52]) {
    uint64_t c64[kCols];
    if (!sobol_columns64(dim, c64)) return false;
    for (int k = 0; k < kCols; ++k) cols[k] = uint32_t(c64[k] >> (kCols - 32));
    return true;
}

// VdCSobolMatrices[m - 1][0 .. 52 - 2m) and VdCSobolMatricesInv[m - 1][0 .. 2m) for 1 <= m <= 16
bool sobol_vdc(int m, uint64_t vdc[52], uint64_t inv[52]) {
    if (m < 1 || m > 16) return false;
    uint32_t c0[kCols], c1[kCols];
    sobol_columns32(0, c0);
    sobol_columns32(1, c1);
    const int n = 2 * m;
    // what index bit `col` adds to the pixel vector (x << m | y): the upper m bits of both dimensions' columns
    auto pix = [&](int col) { return (uint64_t(c0[col] >> (32 - m)) << m) | uint64_t(c1[col] >> (32 - m)); };
    for (int c = 0; c < kCols; ++c) vdc[c] = inv[c] = 0;
    for (int c = 0; c + n < kCols; ++c) vdc[c] = pix(n + c);
    // Gauss-Jordan over GF(2) on the n x n matrix A (column c = pix(c)), rows as bit masks, augmented with the identity
    std::vector<uint64_t> row(size_t(n), 0), aug(size_t(n), 0);
    for (int c = 0; c < n; ++c) {
        const uint64_t a = pix(c);
        for (int r = 0; r < n; ++r)
            if ((a >> r) & 1) row[size_t(r)] |= uint64_t(1) << c;
    }
    for (int r = 0; r < n; ++r) aug[size_t(r)] = uint64_t(1) << r;
    std::vector<int> pivot_row(size_t(n), -1);
    int r0 = 0;
    for (int c = 0; c < n; ++c) {
        int pr = -1;
        for (int r = r0; r < n; ++r)
            if ((row[size_t(r)] >> c) & 1) {
                pr = r;
                break;
            }
        if (pr < 0) return false;  // (A is invertible: the first two dimensions form a (0, 2)-sequence)
        std::swap(row[size_t(r0)], row[size_t(pr)]);
        std::swap(aug[size_t(r0)], aug[size_t(pr)]);
        for (int r = 0; r < n; ++r)
            if (r != r0 && ((row[size_t(r)] >> c) & 1)) {
                row[size_t(r)] ^= row[size_t(r0)];
                aug[size_t(r)] ^= aug[size_t(r0)];
            }
        pivot_row[size_t(c)] = r0++;
    }
    // index bit c = aug[pivot_row[c]] . b  =>  column r of the inverse collects the index bits that b's bit r sets
    for (int c = 0; c < n; ++c) {
        const uint64_t coef = aug[size_t(pivot_row[size_t(c)])];
        for (int r = 0; r < n; ++r)
            if ((coef >> r) & 1) inv[r] |= uint64_t(1) << c;
    }
    return true;
}

}  // namespace iile
