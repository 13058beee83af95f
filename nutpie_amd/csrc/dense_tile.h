// dense_tile.h — the fp64 matrix-core tile of the dense-precision Gaussian's gradient (BASELINE.json configs[1], variant (ii) of
// "1000-dim correlated Gaussian" with a dense precision matrix).
//
//     logp(x) = -1/2 (x - mu)' P (x - mu),   grad = -P (x - mu),   P symmetric [D][D]
//
// For a batch of chains the gradient is a GEMM  G[chain][j] = -sum_k Z[chain][k] P[j][k]  (Z = X - mu; P symmetric, so both operands
// are read along their rows, K-contiguous).  One wavefront computes a 32 x 32 block of G — 2 x 2 blocks of v_mfma_f64_16x16x4_f64 —
// reading its operand fragments straight from global memory / L2 (no LDS: the kernels that call this have none to spare, and at
// 64 cycles per fp64 MFMA the operand traffic of a 32 x 32 register block is 8 bytes per cycle and wave).
//
// Summation order (include/nphip_spec.h, "dense gradient"; restated by the CPU checker's DenseModel): per output element ONE
// accumulator, starting at +0.0, updated by fused multiply-adds in this order of k:
//     for k0 = 0, 16, 32, ...:  for s = 0..3:  for t = 0..3:   k = k0 + 4 t + s      (k >= D contributes nothing)
// — lane (row m, t = lane >> 4) of a fragment holds the four consecutive elements k0 + 4 t .. k0 + 4 t + 3 of its row (two 16-byte
// loads: a full 128-byte line per row and k0), MFMA number s of a k0-step consumes element s of every lane, and the matrix core adds
// the four products of one instruction in the order of t.  g = -acc.
#pragma once

namespace nphip {

typedef double dg_v4 __attribute__((ext_vector_type(4)));
typedef double dg_v2 __attribute__((ext_vector_type(2)));
typedef dg_v2 __attribute__((aligned(8))) dg_double2_u;   // rows of a dense [n][D] array start on 8-byte boundaries when D is odd

struct DgFrag { double e[4]; };

__device__ __forceinline__ DgFrag dg_load4(const double* row, int64_t kb) {
    const dg_double2_u lo = *(const NPHIP_GLOBAL dg_double2_u*)(row + kb);
    const dg_double2_u hi = *(const NPHIP_GLOBAL dg_double2_u*)(row + kb + 2);
    DgFrag f;
    f.e[0] = lo.x; f.e[1] = lo.y; f.e[2] = hi.x; f.e[3] = hi.y;
    return f;
}
// the same where the row ends inside (or before) the fragment: elements at k >= D are +0.0 and are not read
__device__ __forceinline__ DgFrag dg_load4_guarded(const double* row, int64_t kb, int64_t D) {
    DgFrag f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f.e[e] = (kb + e < D) ? *(const NPHIP_GLOBAL double*)(row + kb + e) : 0.0;
    return f;
}

struct DgStage { DgFrag x[2], p[2], mu; };

// One wavefront's 32 x 32 block.  xrow[r] / prow[c]: this lane's row of X (dense, D elements) in row block r and of the padded
// precision matrix (KP elements, zero beyond D) in column block c — row index lane & 15 of the block.  mu: padded to KP with zeros.
// On return acc[r][c][j] is the ACCUMULATOR (the gradient is its negative) of row (lane >> 4) + 4 j, column lane & 15 of block (r, c).
__device__ __forceinline__ void dense_block_32x32(const double* const (&xrow)[2], const double* const (&prow)[2], const double* mu, int64_t D,
                                                  int lane, dg_v4 (&acc)[2][2]) {
    const int64_t t4 = 4 * (int64_t)(lane >> 4);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = (dg_v4){0.0, 0.0, 0.0, 0.0};
    const int64_t kfull = (D / 16) * 16;   // k0-steps whose sixteen k are all inside the rows
    auto load = [&](int64_t k0) {
        DgStage s;
        const int64_t kb = k0 + t4;
        s.x[0] = dg_load4(xrow[0], kb); s.x[1] = dg_load4(xrow[1], kb);
        s.p[0] = dg_load4(prow[0], kb); s.p[1] = dg_load4(prow[1], kb);
        s.mu = dg_load4(mu, kb);
        return s;
    };
    auto compute = [&](const DgStage& s) {
        double z[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) z[r][e] = s.x[r].e[e] - s.mu.e[e];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[r][e], s.p[c].e[e], acc[r][c], 0, 0, 0);
    };
    // Three stages in flight: the operands of k0 + 32 are requested before the sixteen MFMAs of k0 are issued (2 x 1536 cycles of
    // matrix-core time cover an L2 miss into the Infinity Cache).  The steady-state loop is BRANCH-FREE — every load in it is
    // unconditional: behind a conditional load the compiler can no longer count the loads in flight and waits for all of them
    // (s_waitcnt vmcnt(0)), which exposed a full memory latency per three steps (measured: 129 k against 96 k cycles per round).
    int64_t k0 = 0;
    if (kfull >= 32) {
        DgStage s0 = load(0), s1 = load(16), s2;
        for (; k0 + 80 <= kfull; k0 += 48) {
            s2 = load(k0 + 32);
            compute(s0);
            s0 = load(k0 + 48);
            compute(s1);
            s1 = load(k0 + 64);
            compute(s2);
        }
        // what is left of the full steps: two (already loaded) to four
        const int64_t rem = (kfull - k0) / 16;
        if (rem >= 3) s2 = load(k0 + 32);
        compute(s0);
        if (rem >= 4) s0 = load(k0 + 48);
        compute(s1);
        if (rem >= 3) compute(s2);
        if (rem >= 4) compute(s0);
        k0 = kfull;
    } else {
        for (; k0 + 16 <= kfull; k0 += 16) compute(load(k0));
    }
    if (k0 < D) {   // the last, partial step: X guarded element by element (P and mu are padded with zeros)
        DgStage s;
        const int64_t kb = k0 + t4;
        s.x[0] = dg_load4_guarded(xrow[0], kb, D); s.x[1] = dg_load4_guarded(xrow[1], kb, D);
        s.p[0] = dg_load4(prow[0], kb); s.p[1] = dg_load4(prow[1], kb);
        s.mu = dg_load4(mu, kb);
        compute(s);
    }
}

}  // namespace nphip
