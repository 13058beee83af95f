// Native device log-density for the radon model of BASELINE.json config 3 — a model written against the engine's
// batched DEVICE callback (nphip_device_logp_fn, include/nutpie_hip.h): the same density as nutpie_amd/radon.py, as
// one HIP kernel per evaluation (one wavefront per chain) instead of ~35 torch kernels.  Test/benchmark fixture: it
// shows what the callback path costs when the model side is not launch-bound.
//
// Unconstrained vector (D = 2n + 3): [intercept, raw(n-1), log sd, floor_effect, craw(n-1), log csd, log sigma]
// (nutpie_amd/radon.py, model of reference README.md:60-88).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr int kMaxCounties = 128;

struct RadonData {
    int n_counties, n_obs;
    const int* county;        // [n_obs] county of each observation
    const double* floor_;     // [n_obs]
    const double* y;          // [n_obs]
    const int* row_start;     // [n_counties + 1] CSR: observations of each county (deterministic scatter-add)
    const int* row_obs;       // [n_obs]
};

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wavefront per chain, 4 chains per block
__global__ __launch_bounds__(256) void radon_logp_grad(RadonData d, uint64_t n_chains, const double* __restrict__ q,
                                                       double* __restrict__ grad, double* __restrict__ logp) {
    extern __shared__ double lds[];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t chain = (uint64_t)blockIdx.x * 4 + wib;
    if (chain >= n_chains) return;
    const int n = d.n_counties, n_obs = d.n_obs, D = 2 * n + 3;
    double* eff = lds + (size_t)wib * (2 * kMaxCounties + d.n_obs);   // [ce(n) | cfe(n)] then w[n_obs]
    double* cfe = eff + kMaxCounties;
    double* w = eff + 2 * kMaxCounties;
    const double* x = q + chain * D;
    double* g = grad + chain * D;
    const int o_raw = 1, o_lsd = n, o_floor = n + 1, o_craw = n + 2, o_lcsd = 2 * n + 1, o_lsig = 2 * n + 2;
    const double intercept = x[0], fe = x[o_floor], lsd = x[o_lsd], lcsd = x[o_lcsd], lsig = x[o_lsig];
    const double sd = exp(lsd), csd = exp(lcsd), sig = exp(lsig), inv_sig = 1.0 / sig;
    const double c1 = 1.0 / (sqrt((double)n) + n), c2 = 1.0 / sqrt((double)n);
    // zero-sum extension of the two raw vectors (PyMC ZeroSumTransform.backward)
    double s_raw = 0.0, s_craw = 0.0, ss = 0.0;
    for (int j = lane; j < n - 1; j += 64) {
        const double a = x[o_raw + j], b = x[o_craw + j];
        s_raw += a; s_craw += b; ss += a * a + b * b;
    }
    s_raw = wave_sum(s_raw); s_craw = wave_sum(s_craw); ss = wave_sum(ss);
    for (int j = lane; j < n; j += 64) {
        const double e = (j < n - 1) ? x[o_raw + j] - s_raw * c1 : -s_raw * c2;
        const double ce = (j < n - 1) ? x[o_craw + j] - s_craw * c1 : -s_craw * c2;
        eff[j] = e * sd;
        cfe[j] = ce * csd;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // observations: residuals, d lp / d mu
    double rr = 0.0, sw = 0.0, swf = 0.0;
    for (int o = lane; o < n_obs; o += 64) {
        const int cty = d.county[o];
        const double fl = d.floor_[o];
        const double mu = intercept + eff[cty] + fl * (fe + cfe[cty]);
        const double r = (d.y[o] - mu) * inv_sig;
        const double wo = r * inv_sig;
        w[o] = wo;
        rr += r * r; sw += wo; swf += wo * fl;
    }
    rr = wave_sum(rr); sw = wave_sum(sw); swf = wave_sum(swf);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // per-county sums of w (county effect) and w * floor (county floor effect): one lane per county, fixed order
    double dot_e = 0.0, dot_c = 0.0, su_e = 0.0, su_c = 0.0;
    double ge_[2] = {0.0, 0.0}, gc_[2] = {0.0, 0.0};
    for (int t = 0, j = lane; j < n; j += 64, ++t) {
        double a = 0.0, b = 0.0;
        for (int k = d.row_start[j]; k < d.row_start[j + 1]; ++k) {
            const int o = d.row_obs[k];
            a += w[o];
            b += w[o] * d.floor_[o];
        }
        ge_[t] = a; gc_[t] = b;
        const double ext = eff[j] / sd, cext = cfe[j] / csd;
        dot_e += ext * a; dot_c += cext * b;
        if (j < n - 1) { su_e += a * sd; su_c += b * csd; }
    }
    dot_e = wave_sum(dot_e); dot_c = wave_sum(dot_c); su_e = wave_sum(su_e); su_c = wave_sum(su_c);
    // last county's (scaled) gradient, needed by the transpose of the extension
    const int last_lane = (n - 1) & 63, last_t = (n - 1) >> 6;
    const double gl_e = __shfl(last_t == 0 ? ge_[0] : ge_[1], last_lane, 64) * sd;
    const double gl_c = __shfl(last_t == 0 ? gc_[0] : gc_[1], last_lane, 64) * csd;
    for (int t = 0, j = lane; j < n - 1; j += 64, ++t) {
        g[o_raw + j] = (ge_[t] * sd - (c1 * su_e + c2 * gl_e)) - x[o_raw + j];
        g[o_craw + j] = (gc_[t] * csd - (c1 * su_c + c2 * gl_c)) - x[o_craw + j];
    }
    if (lane == 0) {
        g[0] = -0.01 * intercept + sw;
        g[o_floor] = -0.25 * fe + swf;
        g[o_lsd] = 1.0 - sd * sd + sd * dot_e;
        g[o_lcsd] = 1.0 - csd * csd + csd * dot_c;
        g[o_lsig] = 1.0 - sig * sig / 2.25 + rr - n_obs;
        logp[chain] = -0.005 * intercept * intercept - 0.125 * fe * fe - 0.5 * ss - 0.5 * sd * sd + lsd - 0.5 * csd * csd + lcsd
                      - (0.5 / 2.25) * sig * sig + lsig - 0.5 * rr - n_obs * lsig;
    }
}

struct Handle {
    RadonData d;
    void* bufs[5];
};

}  // namespace

extern "C" {

// host arrays: county[n_obs] (int32), floor[n_obs], y[n_obs]; returns an opaque handle (user_data of the callback)
void* radon_device_create(int n_counties, int n_obs, const int* county, const double* floor_, const double* y) {
    if (n_counties > kMaxCounties || n_counties > 128) return nullptr;
    Handle* h = (Handle*)calloc(1, sizeof(Handle));
    int* row_start = (int*)calloc(n_counties + 1, sizeof(int));
    int* row_obs = (int*)calloc(n_obs, sizeof(int));
    for (int o = 0; o < n_obs; ++o) row_start[county[o] + 1]++;
    for (int j = 0; j < n_counties; ++j) row_start[j + 1] += row_start[j];
    int* fill = (int*)calloc(n_counties, sizeof(int));
    for (int o = 0; o < n_obs; ++o) row_obs[row_start[county[o]] + fill[county[o]]++] = o;
    const size_t sizes[5] = {n_obs * sizeof(int), n_obs * sizeof(double), n_obs * sizeof(double), (n_counties + 1) * sizeof(int), n_obs * sizeof(int)};
    const void* src[5] = {county, floor_, y, row_start, row_obs};
    for (int k = 0; k < 5; ++k) {
        if (hipMalloc(&h->bufs[k], sizes[k]) != hipSuccess) return nullptr;
        if (hipMemcpy(h->bufs[k], src[k], sizes[k], hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    free(row_start); free(row_obs); free(fill);
    h->d.n_counties = n_counties; h->d.n_obs = n_obs;
    h->d.county = (const int*)h->bufs[0]; h->d.floor_ = (const double*)h->bufs[1]; h->d.y = (const double*)h->bufs[2];
    h->d.row_start = (const int*)h->bufs[3]; h->d.row_obs = (const int*)h->bufs[4];
    return h;
}

void radon_device_free(void* handle) {
    Handle* h = (Handle*)handle;
    if (!h) return;
    for (int k = 0; k < 5; ++k) (void)hipFree(h->bufs[k]);
    free(h);
}

// nphip_device_logp_fn
int radon_device_logp(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp, void* stream, void* user_data) {
    Handle* h = (Handle*)user_data;
    if (!h || dim != (uint64_t)(2 * h->d.n_counties + 3)) return -1;
    const size_t lds_bytes = 4 * (size_t)(2 * kMaxCounties + h->d.n_obs) * sizeof(double);
    hipLaunchKernelGGL(radon_logp_grad, dim3((unsigned)((n_chains + 3) / 4)), dim3(256), lds_bytes, (hipStream_t)stream, h->d, n_chains, q, grad, logp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
