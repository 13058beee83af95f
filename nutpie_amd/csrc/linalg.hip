// nutpie-hip: batched symmetric eigendecomposition on the device (include/nutpie_hip.h: nphip_batched_eigh).
//
// What it is for: the window estimator of adaptation="low_rank" (reference: src/wrapper.rs:307-334, python/nutpie/sample.py:921-933;
// nuts-rs does this with faer's self-adjoint eigendecomposition, Cargo.lock `faer 0.24`) needs four eigendecompositions of order
// 2m <= 128 per chain and window — of the Gram matrix of the window's scaled draws and gradients, of the projected gradient
// covariance, of the geometric-mean problem and of the metric itself (nutpie_amd/low_rank.py::estimate) — for hundreds of chains
// at once.  rocSOLVER's batched syevd takes 90-210 ms per call for 512 problems of order <= 64 and 5-12 ms above (scratch/eigh_time.py).
//
// One workgroup (256 threads) per matrix, the matrix resident in LDS for the whole computation (order 128: 129 KB of the CU's 160 KB;
// rows padded to an odd stride so that a column walk is conflict-free), three classical stages, all in place:
//   1. Householder tridiagonalisation  T = Q' A Q  (the reflectors parked below the sub-diagonal, LAPACK's dsytd2 layout),
//   2. Q formed in place from the reflectors (LAPACK's dorgtr / dorg2r recurrence),
//   3. implicit QL with Wilkinson shifts on T (EISPACK imtql2 / "tqli"), the rotations of a sweep computed by one lane and then
//      applied by one lane per ROW of Q — consecutive rotations of a sweep share a column, a row's owner applies them in order
//      without synchronisation.
// Results per matrix do not depend on what else is in the batch (rocSOLVER's do, in the last bits), so the estimator's output no
// longer depends on which chains are handed in together.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/nutpie_hip.h"

namespace nphip_linalg {

constexpr int kMaxOrder = 128;
constexpr int kThreads = 256;
constexpr int kMaxSweeps = 60;   // QL iterations per eigenvalue before giving up (EISPACK: 30)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the workgroup, returned to every thread (two barriers; `red`: 4 doubles of LDS)
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
    v = wave_sum(v);
    __syncthreads();   // (red may still be read from the previous call)
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max(double v, double* red, int tid) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// lane `l` (uniform) of a value held across wave 0
__device__ __forceinline__ double lane_read(double a, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), l), __builtin_amdgcn_readlane(__double2loint(a), l));
}
__device__ __forceinline__ double lane_get(double a0, double a1, int i) {   // both halves are read, a scalar select picks: no branch
    const double x0 = lane_read(a0, i & 63), x1 = lane_read(a1, i & 63);
    return (i & 64) ? x1 : x0;
}

// status per matrix: 0 ok, 1 QL did not converge
// mode 0: the decomposition; 1 (test hook): stop after stage 1 — W = diagonal of T, row 0 of A = its sub-diagonal; 2: stop after stage 2 — A = Q
__global__ __launch_bounds__(kThreads) void k_batched_eigh(int n, double* __restrict__ A_all, double* __restrict__ W_all, int* __restrict__ status, int mode) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int ld = n | 1;
    double* V = lds;                 // n x ld
    double* d = V + (size_t)n * ld;  // diagonal of T, then the eigenvalues
    double* e = d + n;               // sub-diagonal: e[i] couples i and i + 1; e[n - 1] = 0
    double* tau = e + n;
    double* pv = tau + n;            // the current reflector v (v[0] = 1)
    double* wv = pv + n;             // w = p - K v
    double* part = wv + n;           // 2 x n partial products
    double* cs = part + 2 * n;       // the rotations of a QL sweep
    double* sn = cs + n;
    double* red = sn + n;            // 4
    int* ctl = (int*)(red + 4);      // 4: {state, m, lo, perm scratch flag}
    int* perm = ctl + 4;             // n
    double* A = A_all + (size_t)blockIdx.x * n * n;

    const long long T0 = (long long)__builtin_readcyclecounter();
    long long tA = 0, tB = 0, nrot = 0, nsweep = 0;
    // ---- load, scaled by 1 / max |a_ij| (no over- or underflow in the squares below; the eigenvalues are scaled back)
    double amax = 0.0;
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx - i * n;
        const double a = (j <= i) ? A[idx] : A[(size_t)j * n + i];   // the lower triangle is the matrix
        V[i * ld + j] = a;
        // (fmax drops a NaN operand: a matrix with SOME non-finite entries would pass the test below and spin through every sweep
        //  of every eigenvalue on NaNs — ADVICE r4; anything that is not a finite number makes the maximum infinite instead)
        amax = (fabs(a) <= 1.0e300) ? fmax(amax, fabs(a)) : __builtin_inf();
    }
    amax = block_max(amax, red, tid);
    if (!(amax > 0.0) || !(amax < 1.0e300)) {   // zero (or non-finite) matrix: eigenvalues 0 (or NaN), eigenvectors the unit vectors
        const double fill = (amax == 0.0) ? 0.0 : __builtin_nan("");
        for (int idx = tid; idx < n * n; idx += kThreads) { const int i = idx / n, j = idx - i * n; A[idx] = (i == j) ? 1.0 : 0.0; }
        for (int i = tid; i < n; i += kThreads) W_all[(size_t)blockIdx.x * n + i] = fill;
        if (tid == 0) status[blockIdx.x] = 0;
        return;
    }
    const double inv_amax = 1.0 / amax;
    for (int idx = tid; idx < n * n; idx += kThreads) { const int i = idx / n, j = idx - i * n; V[i * ld + j] *= inv_amax; }
    __syncthreads();

    // ---- 1. tridiagonalisation.  Step k: the reflector H = I - tau v v' (v[0] = 1) that maps x = A[k+1.., k] onto beta e_1;
    //         trailing block A22 <- H A22 H = A22 - v w' - w v',  w = p - (tau/2 v'p) v,  p = tau A22 v.
    const int row = tid & 127, half = tid >> 7;   // two threads per row of the trailing block
    for (int k = 0; k + 2 < n; ++k) {
        const int s = n - k - 1;
        const double* x = V + (size_t)(k + 1) * ld + k;   // x[i] = x[i * ld]
        const double xi = (tid < s) ? x[(size_t)tid * ld] : 0.0;
        const double xnorm2 = block_sum((tid >= 1 && tid < s) ? xi * xi : 0.0, red, tid);
        const double x0 = x[0];
        if (xnorm2 == 0.0) {   // nothing to annihilate
            if (tid == 0) { tau[k] = 0.0; e[k] = x0; d[k] = V[(size_t)k * ld + k]; }
            __syncthreads();
            continue;
        }
        const double beta = -copysign(sqrt(fma(x0, x0, xnorm2)), x0);
        const double tk = (beta - x0) / beta, scale = 1.0 / (x0 - beta);
        if (tid < s) {
            const double vi = (tid == 0) ? 1.0 : xi * scale;
            pv[tid] = vi;
            if (tid >= 1) V[(size_t)(k + 1 + tid) * ld + k] = vi;   // parked for stage 2
        }
        if (tid == 0) { tau[k] = tk; e[k] = beta; d[k] = V[(size_t)k * ld + k]; }
        __syncthreads();
        // p = tau A22 v: the two halves of a row's product
        const int hs = (s + 1) >> 1, j0 = half * hs, j1 = min(s, j0 + hs);
        if (row < s) {
            const double* ar = V + (size_t)(k + 1 + row) * ld + (k + 1);
            double acc = 0.0;
#pragma unroll 8
            for (int j = j0; j < j1; ++j) acc = fma(ar[j], pv[j], acc);   // (unrolled: the LDS reads of several terms in flight)
            part[half * n + row] = acc;
        }
        __syncthreads();
        double pi = 0.0, vi = 0.0;
        if (tid < s) { pi = tk * (part[tid] + part[n + tid]); vi = pv[tid]; }
        const double K = 0.5 * tk * block_sum(pi * vi, red, tid);
        if (tid < s) wv[tid] = pi - K * vi;
        __syncthreads();
        if (row < s) {
            double* ar = V + (size_t)(k + 1 + row) * ld + (k + 1);
            const double vr = pv[row], wr = wv[row];
#pragma unroll 8
            for (int j = j0; j < j1; ++j) ar[j] = ar[j] - (vr * wv[j] + wr * pv[j]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { d[n - 2] = V[(size_t)(n - 2) * ld + (n - 2)]; e[n - 2] = V[(size_t)(n - 1) * ld + (n - 2)]; }
        d[n - 1] = V[(size_t)(n - 1) * ld + (n - 1)];
        e[n - 1] = 0.0;
    }
    __syncthreads();

    const long long T1 = (long long)__builtin_readcyclecounter();
    if (mode == 1) {
        for (int i = tid; i < n; i += kThreads) { W_all[(size_t)blockIdx.x * n + i] = d[i] * amax; A[i] = e[i] * amax; }
        if (tid == 0) status[blockIdx.x] = 0;
        return;
    }

    // ---- 2. Q = H_0 H_1 ... H_{n-3} in place.  The reflectors move one column to the right (reflector k: rows k+2.., column k+1),
    //         row and column 0 become the unit vector, and the trailing (n-1) x (n-1) block is built from its last column backwards.
    if (tid < n) {
        double* r = V + (size_t)tid * ld;
        for (int c = tid - 1; c >= 1; --c) r[c] = r[c - 1];   // (only the entries below the sub-diagonal matter: c <= tid - 1)
        for (int c = tid; c < n; ++c) r[c] = (c == tid) ? 1.0 : 0.0;   // diagonal and everything right of it: unit matrix
        if (tid >= 1) r[0] = 0.0;
    }
    __syncthreads();
    // In terms of Qs = V[1.., 1..] (order m = n - 1): reflector j sits in Qs[j+1.., j]; the last column is already the unit vector.
    for (int j = n - 3; j >= 0; --j) {
        const double tj = tau[j];
        const int rj = j + 1;   // row / column of V
        // apply H_j to the columns right of rj: one thread per column
        if (tid > rj && tid < n) {
            const int c = tid;
            double dot = V[(size_t)rj * ld + c];   // v[rj] = 1
#pragma unroll 8
            for (int r = rj + 1; r < n; ++r) dot = fma(V[(size_t)r * ld + rj], V[(size_t)r * ld + c], dot);
            dot *= tj;
            V[(size_t)rj * ld + c] -= dot;
#pragma unroll 8
            for (int r = rj + 1; r < n; ++r) V[(size_t)r * ld + c] = fma(-dot, V[(size_t)r * ld + rj], V[(size_t)r * ld + c]);
        }
        __syncthreads();
        // column rj itself: H_j e_1 = e_1 - tau v
        if (tid > rj && tid < n) V[(size_t)tid * ld + rj] *= -tj;
        if (tid == rj) V[(size_t)rj * ld + rj] = 1.0 - tj;
        __syncthreads();
    }

    const long long T2 = (long long)__builtin_readcyclecounter();
    if (mode == 2) {
        for (int idx = tid; idx < n * n; idx += kThreads) { const int i = idx / n, j = idx - i * n; A[idx] = V[(size_t)i * ld + j]; }
        for (int i = tid; i < n; i += kThreads) W_all[(size_t)blockIdx.x * n + i] = d[i] * amax;
        if (tid == 0) status[blockIdx.x] = 0;
        return;
    }

    // ---- 3. implicit QL on (d, e), the rotations accumulated into the columns of V.  A sweep's scalar recurrence is sequential; it
    //         runs on wave 0 with every lane computing the same values: the sweep's operands d[i], e[i] are read from a lane-resident
    //         snapshot (v_readlane: no LDS round trip inside the dependent chain; a sweep reads each entry before it writes it), the
    //         negligible sub-diagonal is found by one ballot instead of a walk, 1 / r and r come from one v_rsq_f64 and two Newton steps.
    const bool w0 = tid < 64;
    const double eps = 2.220446049250313e-16;
    // a sub-diagonal below eps times the norm of T is zero.  A test relative to its two neighbours alone never ends inside a cluster
    // that shares an unreduced block with eigenvalues a million times larger — gamma I + a low-rank covariance is exactly that, and
    // so is a Gram matrix of low rank (a cluster of zeros): every sweep over the block commits roundings of size eps |T|
    double anorm = 0.0;
    if (w0) {
        if (tid < n) anorm = fabs(d[tid]) + fabs(e[tid]);
        if (tid + 64 < n) anorm = fmax(anorm, fabs(d[tid + 64]) + fabs(e[tid + 64]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) anorm = fmax(anorm, __shfl_xor(anorm, off, 64));
    }
    int failed = 0;
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        while (true) {
            const long long ta = (long long)__builtin_readcyclecounter();
            if (w0) {
                const int j0 = tid, j1 = tid + 64;
                const double d0 = j0 < n ? d[j0] : 0.0, d1 = j1 < n ? d[j1] : 0.0;
                const double e0 = j0 < n ? e[j0] : 0.0, e1 = j1 < n ? e[j1] : 0.0;
                const double dn0 = j0 + 1 < n ? d[j0 + 1] : 0.0, dn1 = j1 + 1 < n ? d[j1 + 1] : 0.0;
                const bool z0 = j0 >= l && j0 < n - 1 && fabs(e0) <= eps * fmax(fabs(d0) + fabs(dn0), anorm);
                const bool z1 = j1 >= l && j1 < n - 1 && fabs(e1) <= eps * fmax(fabs(d1) + fabs(dn1), anorm);
                const unsigned long long b0 = __ballot(z0), b1 = __ballot(z1);
                const int m = b0 ? __ffsll((long long)b0) - 1 : (b1 ? 63 + __ffsll((long long)b1) : n - 1);
                int state = 1, lo = l;
                if (m == l) state = 0;
                else if (iter >= kMaxSweeps) state = 2;
                else {
                    const double dl = lane_get(d0, d1, l), el = lane_get(e0, e1, l), dm = lane_get(d0, d1, m);
                    double g = (lane_get(d0, d1, l + 1) - dl) / (2.0 * el);
                    const double r0 = sqrt(fma(g, g, 1.0));
                    g = dm - dl + el / (g + copysign(r0, g));
                    double s = 1.0, c = 1.0, p = 0.0, dup = dm;   // dup: d[i + 1] as it was before this sweep
                    bool broke = false;
                    int i = m - 1;
                    for (; i >= l; --i) {
                        const double di = lane_get(d0, d1, i), ei = lane_get(e0, e1, i);
                        const double f = s * ei, b = c * ei;
                        const double x = fma(f, f, g * g);
                        if (x == 0.0) {
                            if (tid == 0) { e[i + 1] = 0.0; d[i + 1] = dup - p; e[m] = 0.0; }
                            broke = true;
                            break;
                        }
                        double y = __builtin_amdgcn_rsq(x);            // 1 / sqrt(x), refined to full precision
                        const double hx = 0.5 * x;
                        y = fma(y, fma(-hx * y, y, 0.5), y);
                        y = fma(y, fma(-hx * y, y, 0.5), y);
                        double rr = x * y;                             // sqrt(x)
                        rr = fma(fma(-rr, rr, x), 0.5 * y, rr);
                        s = f * y;
                        c = g * y;
                        g = dup - p;
                        const double r2 = fma(di - g, s, 2.0 * c * b);
                        p = s * r2;
                        const double dnew = g + p;
                        g = fma(c, r2, -b);
                        if (tid == 0) { e[i + 1] = rr; d[i + 1] = dnew; cs[i] = c; sn[i] = s; }
                        dup = di;
                    }
                    if (broke) lo = i + 1;
                    else if (tid == 0) { d[l] = dl - p; e[l] = g; e[m] = 0.0; }
                }
                if (tid == 0) { ctl[0] = state; ctl[1] = m; ctl[2] = lo; }
            }
            __syncthreads();
            const long long tb = (long long)__builtin_readcyclecounter();
            const int state = ctl[0], m = ctl[1], lo = ctl[2];
            if (state == 1) { nrot += m - lo; ++nsweep; }
            if (state == 1 && tid < n && lo <= m - 1) {
                double* r = V + (size_t)tid * ld;
                double hi = r[m];            // the entry of column i + 1, carried from rotation to rotation
#pragma unroll 8
                for (int i = m - 1; i >= lo; --i) {
                    const double c = cs[i], s = sn[i], zi = r[i];
                    r[i + 1] = fma(s, zi, c * hi);
                    hi = fma(c, zi, -(s * hi));
                }
                r[lo] = hi;
            }
            __syncthreads();   // (ctl, cs, sn, d, e are rewritten by the next sweep)
            tA += tb - ta; tB += (long long)__builtin_readcyclecounter() - tb;
            if (state == 0) break;
            if (state == 2) { failed = 1; break; }
            ++iter;
        }
        if (failed) break;
    }

    // ---- ascending order, write-out (eigenvectors as columns, eigenvalues scaled back)
    if (tid == 0) {
        for (int i = 0; i < n; ++i) perm[i] = i;
        for (int i = 1; i < n; ++i) {   // insertion sort (stable)
            const int pi = perm[i];
            const double key = d[pi];
            int j = i - 1;
            while (j >= 0 && d[perm[j]] > key) { perm[j + 1] = perm[j]; --j; }
            perm[j + 1] = pi;
        }
        status[blockIdx.x] = failed;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx - i * n;
        A[idx] = V[(size_t)i * ld + perm[j]];
    }
    for (int i = tid; i < n; i += kThreads) W_all[(size_t)blockIdx.x * n + i] = d[perm[i]] * amax;
    if (mode == 3 && tid == 0 && n >= 8) {   // (developer aid: cycles of the stages instead of the first eigenvalues)
        double* w = W_all + (size_t)blockIdx.x * n;
        w[0] = (double)(T1 - T0); w[1] = (double)(T2 - T1); w[2] = (double)tA; w[3] = (double)tB; w[4] = (double)nrot; w[5] = (double)nsweep;
        w[6] = (double)((long long)__builtin_readcyclecounter() - T0);
    }
}

size_t lds_bytes(int n) {
    const size_t ld = (size_t)(n | 1);
    return ((size_t)n * ld + 10 * (size_t)n + 4) * sizeof(double) + (4 + (size_t)n) * sizeof(int) + 16;
}

}  // namespace nphip_linalg

extern "C" int nphip_linalg_launch_eigh(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, int* status_device, void* stream, int mode) {
    using namespace nphip_linalg;
    if (order == 0 || order > (uint64_t)kMaxOrder) return (int)hipErrorInvalidValue;
    if (n_batch == 0) return 0;
    hipLaunchKernelGGL(k_batched_eigh, dim3((unsigned)n_batch), dim3(kThreads), lds_bytes((int)order), (hipStream_t)stream, (int)order, a_device, w_device, status_device, mode);
    return (int)hipGetLastError();
}
