// nutpie-hip: the window estimator of adaptation="low_rank" as ONE kernel (include/nutpie_hip.h: nphip_low_rank_estimate).
//
// What it restates: nutpie_amd/low_rank.py::estimate (the published description of nuts-rs' low-rank mass matrix — reference
// src/wrapper.rs:307-334, python/nutpie/sample.py:921-933, docs/sampling-options.qmd:124-144; the crate itself is not in the tree), for
// the shape every model of up to 256 dimensions has: b <= 32 basis draws, i.e. a subspace of r = 2b <= 64 directions.
//
// Why a kernel of its own (round 6): the torch formulation is ~150 launches plus four batched eigendecompositions whose QL recurrence is
// one wave's sequential work — 7 ms per hand-in whether one chain has stopped or 240, while a 512-chain radon job covers 8 draws per
// millisecond: the warm-up's six hand-ins cost more than the warm-up.  Here: one workgroup per chain (or several chains in turn: the
// launch is capped at the CUs a running engine kernel leaves idle), everything between the window in the trace and (sigma^2, V, lambda)
// in LDS —
//   moments over the window -> scaled basis rows Z (never materialised: rebuilt from the trace where needed) -> Gram matrix ->
//   its PIVOTED CHOLESKY factor L, whose rows are the basis rows in an orthonormal basis Q of their span (Z_perm = L Q') ->
//   projected covariances Cx, Cg from the rows of L -> CHOLESKY Cg = R R' -> the geometric mean S = R^-T (R' Cx R)^1/2 R^-1 ->
//   its spectrum, re-centred on the median -> the k_max directions furthest from 1 outside [1 / cutoff, cutoff] -> V = Z_perm' L^-T W_sel
// — with the two symmetric eigenproblems that are left (the square root, the spectrum of S) solved by a one-sided Jacobi method
// (Hestenes): 32 column pairs per step rotate in parallel, 16 lanes per pair (one DPP row), one barrier per step; no sequential
// recurrence anywhere.  (The first build solved four eigenproblems — Gram matrix and Cg too: 55 Jacobi sweeps per chain instead of 24;
// DESIGN.md 10.1 has the history and the numbers.)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/nutpie_hip.h"

namespace nphip_lrest {

constexpr int R = 64;          // order of the subspace problem (2 x basis draws, zero-padded)
constexpr int LD = 66;         // column stride of a matrix in LDS (even: 16-byte accesses; 66 = 2 mod 32 spreads a row over the banks)
#ifndef NPHIP_LREST_THREADS
#define NPHIP_LREST_THREADS 512
#endif
constexpr int kThreads = NPHIP_LREST_THREADS;   // 512: two waves per SIMD — a Jacobi step is a chain of dependent latencies (LDS, cross-lane sums, rsq)
constexpr int kWaves = kThreads / 64;
constexpr int LP = kThreads / 32;     // lanes per column pair of a Jacobi step
constexpr int NJ = (R / LP) / 2;      // 16-byte pieces of a column per lane
constexpr int TN = 1024 / kThreads;   // columns of a thread's block of a 64 x 64 product (rows: 4)
static_assert(kThreads == 256 || kThreads == 512, "256 or 512 threads");
constexpr int kMaxSweeps = 30;
constexpr double kJacobiTol2 = 1e-28;   // a pair of columns is orthogonal when |g_p . g_q| <= 1e-14 |g_p| |g_q|
constexpr int kMaxDim = 512;   // per-dimension vectors kept in LDS
constexpr int kMaxPick = 32;
constexpr int kMaxK = 16;
constexpr int kScratch = R * R + 16;   // per chain: U' parked, then 16 diagnostic words (sweeps of the four eigenproblems, cycles)

struct Params {
    const double* draws;
    const double* grads;
    long long chain_stride, draw_stride;
    const long long* chains;   // device, or null: chain = block index
    int n, dim, m, b;
    int pick[kMaxPick];
    double gamma, log_cutoff;
    int k_max;
    double* sig2;     // [n][dim]
    double* V;        // [n][k_max][dim]
    double* lam;      // [n][k_max]
    int* k_used;      // [n]
    double* scratch;  // [n][kScratch]
};

#define M_(A, r, c) (A)[(c) * LD + (r)]

// C = op(A)[.., k0:k1] op(B)[k0:k1, ..], all R x R in LDS, C distinct from A and B.  Thread (tx, ty) owns a 4 x 4 block.
template <bool TA, bool TB>
__device__ __forceinline__ void mm(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ B, int k0, int k1, int tid) {
    const int r0 = 4 * (tid & 15), c0 = TN * (tid >> 4);
    double acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;
    for (int k = k0; k < k1; ++k) {
        double a[4], b[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = TA ? M_(A, k, r0 + i) : M_(A, r0 + i, k);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = TB ? M_(B, c0 + j, k) : M_(B, k, c0 + j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) M_(C, r0 + i, c0 + j) = acc[i][j];
}

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += red[w];
    return t;
}

// 1 / sqrt(x) and 1 / x to full precision from the hardware's seeds (two / one Newton steps more than the seed's ~26 bits need: the
// compiler's own sqrt and divide carry scaling for denormals and a correctly rounded last bit this rotation has no use for — its angle
// may be off in the last bits as long as c^2 + s^2 = 1, which the refined rsq gives)
__device__ __forceinline__ double fast_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}

// Eigendecomposition of the symmetric matrix in G (R x R): on return column j of Vm is the eigenvector of w[j] (in no particular
// order), G is destroyed.  One-sided Jacobi: G starts as A = A I, V as I; a rotation of columns p, q of both keeps G = A V and makes
// g_p, g_q orthogonal; at convergence g_j = A v_j = w_j v_j.  Pairs by the round-robin tournament (63 steps of 32 disjoint pairs).
// sum over the LP lanes that share a column pair, returned to all of them.  16 lanes are one DPP row: four v_mov_dpp steps (quad
// permutations, then the half-row and row mirrors — after each step the partial sums are uniform over the group the next one crosses),
// no trip through the LDS crossbar (ds_bpermute: what __shfl_xor compiles to — ~100 cycles of latency per level, twelve levels in a chain)
template <int CTRL> __device__ __forceinline__ double dpp_get(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void pair_sum3(double& a, double& b, double& g) {
    if constexpr (LP == 16) {
        a += dpp_get<0xB1>(a); b += dpp_get<0xB1>(b); g += dpp_get<0xB1>(g);
        a += dpp_get<0x4E>(a); b += dpp_get<0x4E>(b); g += dpp_get<0x4E>(g);
        a += dpp_get<0x141>(a); b += dpp_get<0x141>(b); g += dpp_get<0x141>(g);
        a += dpp_get<0x140>(a); b += dpp_get<0x140>(b); g += dpp_get<0x140>(g);
    } else {
        a += dpp_get<0xB1>(a); b += dpp_get<0xB1>(b); g += dpp_get<0xB1>(g);
        a += dpp_get<0x4E>(a); b += dpp_get<0x4E>(b); g += dpp_get<0x4E>(g);
        a += dpp_get<0x141>(a); b += dpp_get<0x141>(b); g += dpp_get<0x141>(g);
    }
}

// rel_tiny: columns whose squared norm is below rel_tiny |A|_F^2 are zero — a Gram matrix of rank 31 has 33 of them, and rotating rounding
// noise against rounding noise takes twenty sweeps to end.
__device__ int jacobi(double* __restrict__ G, double* __restrict__ Vm, double* __restrict__ w, double* __restrict__ red, int tid, double rel_tiny) {
    for (int idx = tid; idx < R * R; idx += kThreads) { const int r = idx & (R - 1), c = idx >> 6; M_(Vm, r, c) = (r == c) ? 1.0 : 0.0; }
    double fro = 0.0;
    for (int idx = tid; idx < R * R; idx += kThreads) { const double a = M_(G, idx & (R - 1), idx >> 6); fro = fma(a, a, fro); }
    fro = block_sum(fro, red, tid);
    const double tiny2 = fro * rel_tiny + 1e-300;
    const int pr = tid / LP, sub = tid % LP;
    int sweeps = 0;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        int rotated = 0;
        ++sweeps;
        for (int step = 0; step < R - 1; ++step) {
            int p, q;
            if (pr == 0) { p = R - 1; q = step; }
            else { p = step + pr; p -= (p >= R - 1) ? R - 1 : 0; q = step + (R - 1) - pr; q -= (q >= R - 1) ? R - 1 : 0; }
#ifdef NPHIP_LREST_SORT
            if (p > q) { const int t_ = p; p = q; q = t_; }   // p < q: the column of larger norm is kept in p (below)
#endif
            double* gp = G + p * LD + 2 * sub;
            double* gq = G + q * LD + 2 * sub;
            double* vp = Vm + p * LD + 2 * sub;
            double* vq = Vm + q * LD + 2 * sub;
            // every operand of the step is read up front — also the eigenvector columns, which only a rotation needs: behind the
            // branch their LDS round trip would follow the rsq chain instead of running beside it
            double2 xp[NJ], xq[NJ], yp[NJ], yq[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { xp[j] = *(const double2*)(gp + 2 * LP * j); xq[j] = *(const double2*)(gq + 2 * LP * j); }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { yp[j] = *(const double2*)(vp + 2 * LP * j); yq[j] = *(const double2*)(vq + 2 * LP * j); }
            double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                a = fma(xp[j].x, xp[j].x, a); a = fma(xp[j].y, xp[j].y, a);
                b = fma(xq[j].x, xq[j].x, b); b = fma(xq[j].y, xq[j].y, b);
                g = fma(xp[j].x, xq[j].x, g); g = fma(xp[j].y, xq[j].y, g);
            }
            pair_sum3(a, b, g);
            if (a > tiny2 && b > tiny2 && g * g > kJacobiTol2 * (a * b)) {   // |g| > tol sqrt(a b)
                // tan of the rotation angle: t = sign(d) 2g / (|d| + sqrt(d^2 + 4 g^2)), d = b - a;  c = 1 / sqrt(1 + t^2), s = c t
                const double d = b - a, g2 = 2.0 * g;
                const double h2 = fma(d, d, g2 * g2);
                const double h = h2 * fast_rsq(h2);
                const double t = copysign(g2, d * g2) * fast_rcp(fabs(d) + h);
                const double c = fast_rsq(fma(t, t, 1.0)), s = c * t;
                rotated = 1;
                // the rotated columns have squared norms a - t g and b + t g; the larger one goes to the lower index (de Rijk's ordering:
                // a graded spectrum — these matrices span ten orders of magnitude — converges in half the sweeps, and the small
                // eigenvalues come out with the accuracy of their own columns)
#ifdef NPHIP_LREST_SORT
                const bool swap = (b + t * g) > (a - t * g);
#else
                const bool swap = false;
#endif
                double* const gu = swap ? gq : gp; double* const gv = swap ? gp : gq;
                double* const vu = swap ? vq : vp; double* const vv = swap ? vp : vq;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double2 u, v;
                    u.x = fma(c, xp[j].x, -(s * xq[j].x)); u.y = fma(c, xp[j].y, -(s * xq[j].y));
                    v.x = fma(s, xp[j].x, c * xq[j].x);    v.y = fma(s, xp[j].y, c * xq[j].y);
                    *(double2*)(gu + 2 * LP * j) = u; *(double2*)(gv + 2 * LP * j) = v;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double2 u, v;
                    u.x = fma(c, yp[j].x, -(s * yq[j].x)); u.y = fma(c, yp[j].y, -(s * yq[j].y));
                    v.x = fma(s, yp[j].x, c * yq[j].x);    v.y = fma(s, yp[j].y, c * yq[j].y);
                    *(double2*)(vu + 2 * LP * j) = u; *(double2*)(vv + 2 * LP * j) = v;
                }
            }
#ifdef NPHIP_LREST_SORT
            else if (b > a && b > tiny2) {
                // (no rotation needed, but out of order: exchange the columns)
                rotated = 1;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { *(double2*)(gp + 2 * LP * j) = xq[j]; *(double2*)(gq + 2 * LP * j) = xp[j]; *(double2*)(vp + 2 * LP * j) = yq[j]; *(double2*)(vq + 2 * LP * j) = yp[j]; }
            }
#endif
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }
    // w_j = v_j' g_j (the Rayleigh quotient: signed, so a rounding-negative eigenvalue stays negative)
    {
        constexpr int TPC = kThreads / R;   // threads per column
        const int j = tid / TPC, part = tid % TPC;
        double acc = 0.0;
        for (int r = part * (R / TPC); r < (part + 1) * (R / TPC); ++r) acc = fma(M_(Vm, r, j), M_(G, r, j), acc);
#pragma unroll
        for (int off = 1; off < TPC; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (part == 0) w[j] = acc;
    }
    __syncthreads();
    return sweeps;
}


// Cholesky factorisation in place, A = L L' (A symmetric, both triangles valid on entry; on return the lower triangle is L, the rest zero).
// perm != nullptr: with diagonal pivoting and a rank decision — the factorisation of P A P' stops at the first pivot below rel_tol times
// the first (largest) one; perm[k] = the original index at position k; returns the rank (columns beyond it are zero).  The Gram matrix of
// the basis rows is factored this way: Z_perm = L Q' with orthonormal Q — the rows of L ARE the coordinates of the basis rows in an
// orthonormal basis of their span, which is all the estimator needs of it (round 6: this replaces the eigendecomposition of the Gram
// matrix, 15 of the kernel's 55 Jacobi sweeps).  perm == nullptr: plain Cholesky of a positive definite matrix.
__device__ int cholesky(double* __restrict__ A, int* __restrict__ perm, double rel_tol, double* __restrict__ red, int* __restrict__ ired, int tid) {
    if (perm && tid < R) perm[tid] = tid;
    __syncthreads();
    int rank = R;
    double d0 = 0.0;
    for (int k = 0; k < R; ++k) {
        double dk;
        if (perm) {
            if (tid < 64) {   // wave 0: the largest remaining diagonal entry (ties: the lowest index)
                double d = (tid >= k) ? M_(A, tid, tid) : -__builtin_inf();
                if (!(d == d)) d = -__builtin_inf();
                double best = d;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
                const unsigned long long bm = __ballot(d == best);
                if (tid == 0) { ired[0] = __ffsll((long long)bm) - 1; red[0] = best; }
            }
            __syncthreads();
            const int piv = ired[0];
            dk = red[0];
            if (k == 0) d0 = dk;
            if (!(dk > rel_tol * d0) || !(dk > 0.0)) { rank = k; break; }
            if (piv != k) {
                if (tid < R) { const double t = M_(A, k, tid); M_(A, k, tid) = M_(A, piv, tid); M_(A, piv, tid) = t; }
                __syncthreads();
                if (tid < R) { const double t = M_(A, tid, k); M_(A, tid, k) = M_(A, tid, piv); M_(A, tid, piv) = t; }
                if (tid == 0) { const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
            }
            __syncthreads();
        } else {
            dk = M_(A, k, k);
            if (!(dk > 0.0)) dk = 1e-300;   // (positive definite by construction: gamma on the diagonal)
            __syncthreads();
        }
        const double lkk = sqrt(dk), inv = 1.0 / lkk;
        if (tid > k && tid < R) M_(A, tid, k) *= inv;
        if (tid == k) M_(A, k, k) = lkk;
        __syncthreads();
        for (int idx = tid; idx < R * R; idx += kThreads) {
            const int i = idx & (R - 1), j = idx >> 6;
            if (i > k && j > k) M_(A, i, j) = fma(-M_(A, i, k), M_(A, j, k), M_(A, i, j));
        }
        __syncthreads();
    }
    for (int idx = tid; idx < R * R; idx += kThreads) {
        const int i = idx & (R - 1), j = idx >> 6;
        if (j > i || j >= rank) M_(A, i, j) = 0.0;
    }
    __syncthreads();
    return rank;
}

// B <- L^-T B in place: L lower triangular of order n (<= R), B has nc columns.  Back substitution, right-looking.
__device__ void solve_lt(const double* __restrict__ L, double* __restrict__ B, int n, int nc, int tid) {
    for (int i = n - 1; i >= 0; --i) {
        const double inv = 1.0 / M_(L, i, i);
        if (tid < nc) M_(B, i, tid) *= inv;
        __syncthreads();
        for (int idx = tid; idx < i * nc; idx += kThreads) {
            const int j = idx % i, c = idx / i;
            M_(B, j, c) = fma(-M_(L, i, j), M_(B, i, c), M_(B, j, c));
        }
        __syncthreads();
    }
}

__device__ void estimate_one(const Params& P, const int slot, double* lds) {
    double* B0 = lds;
    double* B1 = B0 + R * LD;
    double* B2 = B1 + R * LD;
    double* B3 = B2 + R * LD;
    double* mean_x = B3 + R * LD;        // [kMaxDim]
    double* mean_g = mean_x + kMaxDim;
    double* stds = mean_g + kMaxDim;
    double* em = stds + kMaxDim;         // [R] each
    double* es = em + R;
    double* red = es + R;                // [16]: wave sums; [8] the spectrum's centre
    int* isel = (int*)(red + 16);        // [kMaxK + 4]
    int* perm = isel + kMaxK + 4;        // [R]
    const int tid = threadIdx.x;
    const long long chain = P.chains ? P.chains[slot] : (long long)slot;
    const double* X = P.draws + chain * P.chain_stride;
    const double* Gx = P.grads + chain * P.chain_stride;
    const int D = P.dim, m = P.m, b = P.b, r = 2 * b;
    const double inf = __builtin_inf();
    const long long tk0_ = (long long)__builtin_readcyclecounter();

    // ---- A. moments over the whole window (two passes, as torch.mean / torch.std do) and the diagonal scaling sqrt(std(x) / std(g))
    int bad = 0;
    for (int i = tid; i < D; i += kThreads) {
        double sx = 0.0, sg = 0.0;
        for (int t = 0; t < m; ++t) {
            const double x = X[(long long)t * P.draw_stride + i], g = Gx[(long long)t * P.draw_stride + i];
            if (!(fabs(x) < inf) || !(fabs(g) < inf)) bad = 1;
            sx += x; sg += g;
        }
        mean_x[i] = sx / (double)m; mean_g[i] = sg / (double)m;
    }
    bad = __syncthreads_or(bad);   // a window with a non-finite entry says nothing: the identity for this chain
    for (int i = tid; i < D; i += kThreads) {
        double st = 1.0;
        if (!bad) {
            const double mx = mean_x[i], mg = mean_g[i];
            double vx = 0.0, vg = 0.0;
            for (int t = 0; t < m; ++t) {
                const double dx = X[(long long)t * P.draw_stride + i] - mx, dg = Gx[(long long)t * P.draw_stride + i] - mg;
                vx = fma(dx, dx, vx); vg = fma(dg, dg, vg);
            }
            const double sdx = sqrt(vx / (double)(m - 1)), sdg = sqrt(vg / (double)(m - 1));
            st = sqrt(sdx / sdg);
            if (!(fabs(st) < inf) || !(st > 0.0)) st = 1.0;
            st = fmin(fmax(st, 1e-10), 1e10);
        } else {
            mean_x[i] = 0.0; mean_g[i] = 0.0;
        }
        stds[i] = st;
    }
    __syncthreads();

    // Z(t, i): row t of the scaled basis — draws for t < b, gradients for b <= t < 2b, zero padding beyond
    auto zval = [&](int t, int i) -> double {
        if (bad || t >= r) return 0.0;
        if (t < b) return (X[(long long)P.pick[t] * P.draw_stride + i] - mean_x[i]) / stds[i];
        return (Gx[(long long)P.pick[t - b] * P.draw_stride + i] - mean_g[i]) * stds[i];
    };

    // ---- B. Gram matrix G = Z Z' (tiles of 32 dimensions staged in LDS: B1) -> B0
    {
        const int r0 = 4 * (tid & 15), c0 = TN * (tid >> 4);
        double acc[4][TN];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;
        for (int d0 = 0; d0 < D; d0 += 32) {
            __syncthreads();
            for (int e = tid; e < R * 32; e += kThreads) {
                const int dl = e & 31, t = e >> 5;
                M_(B1, t, dl) = (d0 + dl < D) ? zval(t, d0 + dl) : 0.0;
            }
            __syncthreads();
            for (int k = 0; k < 32; ++k) {
                double a[4], bb[TN];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = M_(B1, r0 + i, k);
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[j] = M_(B1, c0 + j, k);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = fma(a[i], bb[j], acc[i][j]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) M_(B0, r0 + i, c0 + j) = acc[i][j];
        __syncthreads();
    }

    // ---- C. the basis rows in an orthonormal basis of their span: Z_perm = L Q' from the pivoted Cholesky factorisation of the Gram
    //         matrix (rank: pivots above 1e-10 of the first).  L is parked in global memory for the end (V = Q W_sel = Z_perm' L^-T W_sel).
    const long long t0_ = (long long)__builtin_readcyclecounter();
    const int rank = cholesky(B0, perm, 1e-10, red, isel, tid);
    const long long t1_ = (long long)__builtin_readcyclecounter();
    double* Us = P.scratch + (size_t)slot * kScratch;
    for (int idx = tid; idx < R * R; idx += kThreads) Us[idx] = M_(B0, idx & (R - 1), idx >> 6);
    // ---- E. Cx = Px' Px / b + gamma I -> B1, Cg = Pg' Pg / b + gamma I -> B2: row k of L is a draw (perm[k] < b) or a gradient
    {
        const int r0 = 4 * (tid & 15), c0 = TN * (tid >> 4);
        double ax[4][TN], ag[4][TN];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) { ax[i][j] = 0.0; ag[i][j] = 0.0; }
        for (int k = 0; k < R; ++k) {   // (every position: with a rank below r a basis row can sit anywhere; the rows of the zero padding are zero)
            const bool isx = perm[k] < b;
            double a[4], bb[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = M_(B0, k, r0 + i);
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = M_(B0, k, c0 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) { const double pr_ = a[i] * bb[j]; ax[i][j] += isx ? pr_ : 0.0; ag[i][j] += isx ? 0.0 : pr_; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const double dg_ = (r0 + i == c0 + j) ? P.gamma : 0.0;
                M_(B1, r0 + i, c0 + j) = ax[i][j] / (double)b + dg_;
                M_(B2, r0 + i, c0 + j) = ag[i][j] / (double)b + dg_;
            }
    }
    __syncthreads();
    int nonfin = 0;
    double trx = 0.0, trg = 0.0;
    for (int idx = tid; idx < R * R; idx += kThreads) {
        const int rr = idx & (R - 1), c = idx >> 6;
        const double cx = M_(B1, rr, c), cg = M_(B2, rr, c);
        if (!(fabs(cx) < inf) || !(fabs(cg) < inf)) nonfin = 1;
        if (rr == c && rr < rank) { trx += cx; trg += cg; }
    }
    nonfin = __syncthreads_or(nonfin);   // a Gram matrix that overflowed says as little as a non-finite window
    trx = block_sum(trx, red, tid);
    trg = block_sum(trg, red, tid);
    const bool ident = bad || nonfin;
    if (ident) {
        for (int i = tid; i < D; i += kThreads) P.sig2[(size_t)slot * D + i] = 1.0;
        for (int idx = tid; idx < P.k_max * D; idx += kThreads) P.V[(size_t)slot * P.k_max * D + idx] = 0.0;
        if (tid < P.k_max) P.lam[(size_t)slot * P.k_max + tid] = 1.0;
        if (tid == 0) P.k_used[slot] = 0;
        return;
    }

    // ---- F. Cg = Rc Rc' (Cholesky, in place in B2).  The geometric mean S = Cx # Cg^-1 — the symmetric positive solution of
    //         S Cg S = Cx — is  S = Rc^-T (Rc' Cx Rc)^1/2 Rc^-1.
    cholesky(B2, nullptr, 0.0, red, isel, tid);
    // ---- G. M = Rc' Cx Rc -> B0 (symmetrised)
    mm<false, false>(B3, B1, B2, 0, R, tid);    // Cx Rc
    __syncthreads();
    mm<true, false>(B0, B2, B3, 0, R, tid);     // Rc' (Cx Rc)
    __syncthreads();
    for (int idx = tid; idx < R * R; idx += kThreads) {   // (thread owns (i, j) and (j, i), i <= j)
        const int i = idx & (R - 1), j = idx >> 6;
        if (i <= j) { const double v = 0.5 * (M_(B0, i, j) + M_(B0, j, i)); M_(B0, i, j) = v; M_(B0, j, i) = v; }
    }
    __syncthreads();
    // ---- H. M = Um diag(em) Um'  (B0 destroyed, Um -> B3);  I. M^1/2 = (Um em^1/4) (Um em^1/4)' -> B1
    const int sw2 = jacobi(B0, B3, em, red, tid, 1e-40);
    for (int idx = tid; idx < R * R; idx += kThreads) {
        const int rr = idx & (R - 1), c = idx >> 6;
        M_(B3, rr, c) *= sqrt(sqrt(fmax(em[c], 0.0)));
    }
    __syncthreads();
    mm<false, true>(B1, B3, B3, 0, R, tid);
    __syncthreads();
    // S = Rc^-T (Rc^-T M^1/2)'  -> B0
    solve_lt(B2, B1, R, R, tid);
    for (int idx = tid; idx < R * R; idx += kThreads) { const int i = idx & (R - 1), j = idx >> 6; M_(B0, i, j) = M_(B1, j, i); }
    __syncthreads();
    solve_lt(B2, B0, R, R, tid);
    for (int idx = tid; idx < R * R; idx += kThreads) {
        const int i = idx & (R - 1), j = idx >> 6;
        if (i <= j) { const double v = 0.5 * (M_(B0, i, j) + M_(B0, j, i)); M_(B0, i, j) = v; M_(B0, j, i) = v; }
    }
    __syncthreads();
    // ---- J. S = W diag(es) W'  (B0 destroyed, W -> B3: already in the coordinates of Q)
    const int sw3 = jacobi(B0, B3, es, red, tid, 1e-40);
    const int sw0 = 0, sw1 = 0;

    // ---- L. spectrum: clamp to what a geometric mean of these two matrices can have, re-centre on the median of the directions
    //         inside the span, pick the k_max furthest from 1 (log scale) among those outside [1 / cutoff, cutoff].  Wave 0, a lane
    //         per direction.
    if (tid < R) {
        const int j = tid;
        double lv = 0.0;
        for (int i = 0; i < rank; ++i) { const double wij = M_(B3, i, j); lv = fma(wij, wij, lv); }   // (inside the span: coordinates below the rank)
        const bool live = lv > 0.5;
        // (bounds of the exact spectrum, from traces: |Cg| <= tr Cg, |Cx| <= tr Cx — a net for what rounding puts outside, not a rule)
        const double lo = sqrt(P.gamma / fmax(trg, P.gamma)), hi = sqrt(fmax(trx, P.gamma) / P.gamma);
        double e = fmax(fmin(es[j], hi), lo);
        const double le = log(e);
        // lower median of le over the live lanes: the lane whose rank (ties by lane) is (n_live - 1) / 2
        const unsigned long long lm = __ballot(live);
        const int n_live = __popcll(lm);
        int rank = 0;
        for (int i = 0; i < R; ++i) {
            const double li = __shfl(le, i, 64);
            const bool il = (lm >> i) & 1ull;
            rank += (il && (li < le || (li == le && i < j))) ? 1 : 0;
        }
        const bool is_med = live && n_live > 0 && rank == (n_live - 1) / 2;
        const unsigned long long mm_ = __ballot(is_med);
        double med = 0.0;
        if (mm_) med = __shfl(le, __ffsll((long long)mm_) - 1, 64);
        const double centre = exp(med);
        e = e / centre;
        const double score0 = fabs(log(e));
        double score = (live && score0 > P.log_cutoff) ? score0 : -1.0;
        es[j] = e;
        if (j == 0) red[8] = centre;
        // top k_max by score (descending; ties: the lower lane)
        int n_used = 0;
        for (int kk = 0; kk < P.k_max; ++kk) {
            double best = score;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
            const unsigned long long bm = __ballot(score == best);
            const int who = __ffsll((long long)bm) - 1;
            if (best > 0.0) { if (j == 0) isel[kk] = who; n_used = kk + 1; if (j == who) score = -2.0; }
            else { if (j == 0) isel[kk] = -1; }
        }
        if (j == 0) isel[kMaxK] = n_used;
    }
    __syncthreads();
    const int n_used = isel[kMaxK];
    const double centre = red[8];
    for (int i = tid; i < D; i += kThreads) { const double s = stds[i] * sqrt(centre); P.sig2[(size_t)slot * D + i] = s * s; }
    if (tid < P.k_max) P.lam[(size_t)slot * P.k_max + tid] = (tid < n_used) ? es[isel[tid]] : 1.0;
    if (tid == 0) P.k_used[slot] = n_used;

    // ---- M. T = L^-T W[:, sel]  (rank x n_used; rows beyond the rank are zero) -> B0;  N. V = Q W_sel = Z_perm' T
    for (int idx = tid; idx < R * R; idx += kThreads) M_(B1, idx & (R - 1), idx >> 6) = Us[idx];   // L back from global memory
    for (int idx = tid; idx < R * kMaxK; idx += kThreads) {
        const int i = idx & (R - 1), jj = idx >> 6;
        M_(B0, i, jj) = (jj < n_used && i < rank) ? M_(B3, i, isel[jj]) : 0.0;
    }
    __syncthreads();
    solve_lt(B1, B0, rank, kMaxK, tid);
    for (int i = tid; i < D; i += kThreads) {
        double acc[kMaxK];
#pragma unroll
        for (int jj = 0; jj < kMaxK; ++jj) acc[jj] = 0.0;
        if (n_used > 0) {
            for (int t = 0; t < rank; ++t) {
                const double z = zval(perm[t], i);
#pragma unroll
                for (int jj = 0; jj < kMaxK; ++jj) acc[jj] = fma(z, M_(B0, t, jj), acc[jj]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < kMaxK; ++jj)
            if (jj < P.k_max) P.V[((size_t)slot * P.k_max + jj) * D + i] = (jj < n_used) ? acc[jj] : 0.0;
    }
    if (tid == 0) {   // diagnostics (scratch/r6_lr_native.py)
        double* dg = Us + R * R;
        dg[0] = sw0; dg[1] = sw1; dg[2] = sw2; dg[3] = sw3;
        dg[4] = (double)(t0_ - tk0_); dg[5] = (double)(t1_ - t0_); dg[6] = (double)((long long)__builtin_readcyclecounter() - tk0_);
    }
}

// A workgroup takes the chains slot, slot + gridDim.x, ...: the launch is capped at the CUs the running engine kernel leaves idle (a
// workgroup's 146 KB of LDS want a CU to themselves, and an estimator workgroup that waits for a CU takes the one the engine's NEXT launch
// needs — measured: every hand-in then stalled the engine for its whole duration, scratch/r6_lr_wall.py)
__global__ __launch_bounds__(kThreads) void k_lr_estimate(const Params P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int slot = blockIdx.x; slot < P.n; slot += gridDim.x) {
        estimate_one(P, slot, lds);
        __syncthreads();
    }
}

constexpr size_t kLdsBytes = ((size_t)4 * R * LD + 3 * kMaxDim + 2 * R + 16) * sizeof(double) + (kMaxK + 4 + R) * sizeof(int) + 16;

}  // namespace nphip_lrest

extern "C" int nphip_low_rank_estimate_supported(uint64_t dim, uint64_t m, uint64_t n_pick, uint64_t k_max) {
    using namespace nphip_lrest;
    return (dim >= 1 && dim <= (uint64_t)kMaxDim && m >= 2 && n_pick >= 1 && n_pick <= (uint64_t)kMaxPick && n_pick <= m && n_pick < dim + 1 &&
            k_max >= 1 && k_max <= (uint64_t)kMaxK) ? 1 : 0;
}

extern "C" int nphip_low_rank_estimate(uint64_t n, uint64_t dim, uint64_t m, uint64_t n_pick, const int32_t* pick, const double* draws, const double* grads,
                                       int64_t chain_stride, int64_t draw_stride, const int64_t* chains_device, double gamma, double cutoff, uint64_t k_max,
                                       double* sigma2, double* V, double* lambda, int32_t* k_used, double* scratch, uint64_t max_workgroups, void* stream) {
    using namespace nphip_lrest;
    if (n == 0) return NPHIP_OK;
    if (!nphip_low_rank_estimate_supported(dim, m, n_pick, k_max) || !pick || !draws || !grads || !sigma2 || !V || !lambda || !k_used || !scratch || !(cutoff > 1.0) || !(gamma > 0.0))
        return NPHIP_ERR;
    Params P;
    P.draws = draws; P.grads = grads; P.chain_stride = chain_stride; P.draw_stride = draw_stride; P.chains = (const long long*)chains_device;
    P.n = (int)n; P.dim = (int)dim; P.m = (int)m; P.b = (int)n_pick;
    for (int i = 0; i < kMaxPick; ++i) P.pick[i] = i < (int)n_pick ? pick[i] : 0;
    for (int i = 0; i < (int)n_pick; ++i) if (pick[i] < 0 || (uint64_t)pick[i] >= m) return NPHIP_ERR;
    P.gamma = gamma; P.log_cutoff = log(cutoff); P.k_max = (int)k_max;
    P.sig2 = sigma2; P.V = V; P.lam = lambda; P.k_used = k_used; P.scratch = scratch;
    (void)hipFuncSetAttribute((const void*)k_lr_estimate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);   // (146 KB of the CU's 160)
    const uint64_t grid = (max_workgroups > 0 && max_workgroups < n) ? max_workgroups : n;
    hipLaunchKernelGGL(k_lr_estimate, dim3((unsigned)grid), dim3(kThreads), kLdsBytes, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? NPHIP_OK : NPHIP_ERR;
}
