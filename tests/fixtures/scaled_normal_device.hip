// scaled_normal_device.hip — test / benchmark fixture (written for this repo).
//
// The density of tests/fixtures/eight_schools.c::scaled_normal_logp — independent normals with scales 0.5, 0.8, ..., 2.3,
// 0.5, ... in any dimension — behind the engine's batched DEVICE callback (nphip_device_logp_fn, include/nutpie_hip.h):
//   scaled_normal_device_seq   one THREAD per chain, the sum over the dimensions in index order: the same IEEE operations in
//                              the same order as the host C function, so a device-callback job can be compared with the CPU
//                              oracle driving that host function BIT FOR BIT (tests/test_gpu_parity.py);
//   scaled_normal_device_fast  one wavefront per chain, lanes stride over the dimensions (a different summation order: for
//                              timing the engine's launch-per-evaluation kernels with a callback that costs almost nothing).
// Compiled with -ffp-contract=off (as the host fixture is).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ __launch_bounds__(64) void k_seq(uint64_t n, uint64_t dim, const double* __restrict__ q, double* __restrict__ grad, double* __restrict__ logp) {
    const uint64_t c = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    const double* x = q + c * dim;
    double* g = grad + c * dim;
    double lp = 0.0;
    for (uint64_t i = 0; i < dim; ++i) {
        const double sd = 0.5 + 0.3 * (double)(i % 7);
        const double z = x[i] / sd;
        lp += -0.5 * z * z;
        g[i] = -z / sd;
    }
    logp[c] = lp;
}

__global__ __launch_bounds__(256) void k_fast(uint64_t n, uint64_t dim, const double* __restrict__ q, double* __restrict__ grad, double* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n) return;
    const double* x = q + c * dim;
    double* g = grad + c * dim;
    double lp = 0.0;
    for (uint64_t i = lane; i < dim; i += 64) {
        const double sd = 0.5 + 0.3 * (double)(i % 7);
        const double z = x[i] / sd;
        lp += -0.5 * z * z;
        g[i] = -z / sd;
    }
    for (int off = 32; off > 0; off >>= 1) lp += __shfl_xor(lp, off, 64);
    if (lane == 0) logp[c] = lp;
}

}  // namespace

extern "C" {

int scaled_normal_device_seq(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp, void* stream, void* user_data) {
    (void)user_data;
    hipLaunchKernelGGL(k_seq, dim3((unsigned)((n_chains + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_chains, dim, q, grad, logp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int scaled_normal_device_fast(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp, void* stream, void* user_data) {
    (void)user_data;
    hipLaunchKernelGGL(k_fast, dim3((unsigned)((n_chains + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_chains, dim, q, grad, logp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
