// host.hip — host driver and C-ABI of libnutpie_hip.so (see include/nutpie_hip.h).
//
// Mirrors, for the diag-NUTS path only, what the reference's PyO3 layer does around
// nuts_rs::Sampler (src/wrapper.rs): a settings object with flat attribute names
// (wrapper.rs:210-451, 563-620), three model flavours (src/pymc.rs raw C callback,
// src/pyfunc.rs callable -> batched device callback, fused analytic Gaussian), and a sampler
// handle with wait/pause/resume/abort/inspect semantics (wrapper.rs:1252-1456).
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nutpie_hip.h"
#include "../../include/nphip_spec.h"
#include "engine_types.h"

namespace nphip {
hipError_t launch_advance(const Args& a, const Args* d_args, bool fused, int W, hipStream_t st, const LaunchSlice* slice = nullptr);
hipError_t launch_resume(const Args* d_args, int n, const int64_t* d_chains, const double* d_pos, bool fused, hipStream_t st);
hipError_t launch_set_metric(const Args* d_args, int n, const int64_t* d_chains, int k, const double* sig2, const double* V, const double* lam, int* d_taken, hipStream_t st);
hipError_t launch_stage_metric(const Args* d_args, int n, const int64_t* d_chains, int k, const double* sig2, const double* V, const double* lam, int* d_taken, hipStream_t st);
hipError_t launch_remote(const Args* d_args, int W, int nv, hipStream_t st, const LaunchSlice& sl);
hipError_t launch_dense_resident(const Args* d_args, int nv, int max_evals, hipStream_t st, const LaunchSlice sl);
hipError_t launch_dense_grad(const double* X, const double* Pp, const double* mu, double* G, double* logp, int64_t n, int64_t D, int64_t KP, int W, hipStream_t st);
hipError_t launch_mfma_f64_rate(double* out, int blocks, int iters, hipStream_t st);
hipError_t launch_test_detmath(int fn, uint64_t n, const double* x, double* y, hipStream_t st);
hipError_t launch_test_dot(int W, uint64_t n, const double* x, const double* y, double* out, hipStream_t st);
}  // namespace nphip

using namespace nphip;

namespace {

thread_local std::string t_error;
void set_error(const std::string& e) { t_error = e; }
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(x) do { if (!hip_ok((x), #x)) return false; } while (0)

}  // namespace

// =========================================================================== settings
struct nphip_settings {
    // DiagNutsSettings (reference src/wrapper.rs:19,120,525-533); defaults: SURVEY.md §8c / App. A
    uint64_t seed = 0;
    uint64_t num_tune = 400, num_draws = 1000, num_chains = 6;
    uint64_t maxdepth = 10, mindepth = 0;
    bool check_turning = true;
    bool store_unconstrained = false, store_gradient = false, store_transformed = false, store_divergences = false;
    double max_energy_error = 1000.0;
    // adapt_options
    double early_window = 0.3, step_size_window = 0.15;
    uint64_t mass_matrix_switch_freq = 80, early_mass_matrix_switch_freq = 10, mass_matrix_update_freq = 1;
    bool store_mass_matrix = false, use_grad_based_estimate = true;
    // step_size_settings
    double initial_step = 0.1, target_accept = 0.8;
    double jitter = 0.0;  // 0 => None
    double max_step_size = INFINITY;
    double da_k = 0.75, da_t0 = 10.0, da_gamma = 0.05;
    bool fixed_step = false;
    double adam_learning_rate = 0.05;
    bool adam = false;  // step_size_adapt_method = "adam"
    // engine knobs (not in the reference)
    bool adapt_mass_matrix = true;
    uint64_t num_try_init = 100;
    std::vector<uint64_t> pause_draws;   // host-driven adaptation hook (nphip_settings_set_pause_draws)
    bool low_rank_metric = false;        // the host may replace a chain's metric at the pause draws (nphip_sampler_set_metric)
};

static int unknown_attr(const char* name) {
    set_error(std::string("Unknown settings attribute: ") + name);
    return NPHIP_ERR_UNKNOWN_ATTR;
}
static int not_available(const char* name, const char* adaptation) {
    // wrapper.rs:138-145
    set_error(std::string("Option ") + name + " not available for " + adaptation + " adaptation");
    return NPHIP_ERR_NOT_AVAILABLE;
}
static int bad_value(const std::string& msg) { set_error(msg); return NPHIP_ERR_BAD_VALUE; }

extern "C" {

const char* nphip_last_error(void) { return t_error.c_str(); }
const char* nphip_version(void) { return "0.1.0"; }
int nphip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

nphip_settings_t* nphip_settings_new_diag(uint64_t seed) {
    auto* s = new nphip_settings();
    s->seed = seed;
    return s;
}
nphip_settings_t* nphip_settings_clone(const nphip_settings_t* s) { return new nphip_settings(*s); }
void nphip_settings_free(nphip_settings_t* s) { delete s; }

int nphip_settings_set_f64(nphip_settings_t* s, const char* name, double v) {
    std::string n(name);
    if (n == "initial_step") s->initial_step = v;
    else if (n == "target_accept") s->target_accept = v;
    else if (n == "max_step_size") s->max_step_size = v;
    else if (n == "max_energy_error") s->max_energy_error = v;
    else if (n == "step_size_jitter") {
        if (v < 0.0) return bad_value("step_size_jitter must be positive");  // wrapper.rs:394-396
        s->jitter = v;
    }
    else if (n == "step_size_adam_learning_rate") s->adam_learning_rate = v;
    else if (n == "mass_matrix_eigval_cutoff" || n == "mass_matrix_gamma") return not_available(name, "diag");
    else if (n == "target_integration_time") return bad_value("target_integration_time is not supported by the HIP engine");
    else if (n == "early_window") s->early_window = v;
    else if (n == "step_size_window") s->step_size_window = v;
    else if (n == "da_k") s->da_k = v;
    else if (n == "da_t0") s->da_t0 = v;
    else if (n == "da_gamma") s->da_gamma = v;
    else return unknown_attr(name);
    return NPHIP_OK;
}

int nphip_settings_set_u64(nphip_settings_t* s, const char* name, uint64_t v) {
    std::string n(name);
    if (n == "num_tune") s->num_tune = v;
    else if (n == "num_draws") s->num_draws = v;
    else if (n == "num_chains") s->num_chains = v;
    else if (n == "maxdepth") {
        if (v < 1 || v > (uint64_t)kMaxDepthCap) return bad_value("maxdepth must be between 1 and 16 for the HIP engine");
        s->maxdepth = v;
    }
    else if (n == "mindepth") s->mindepth = v;
    else if (n == "window_switch_freq" || n == "mass_matrix_switch_freq") s->mass_matrix_switch_freq = v;  // wrapper.rs:214-229, 291-303
    else if (n == "early_window_switch_freq") s->early_mass_matrix_switch_freq = v;
    else if (n == "mass_matrix_update_freq") s->mass_matrix_update_freq = v;
    else if (n == "extra_doublings") { if (v != 0) return bad_value("extra_doublings is not supported by the HIP engine"); }
    else if (n == "seed") s->seed = v;
    else if (n == "num_try_init") s->num_try_init = v;
    else return unknown_attr(name);
    return NPHIP_OK;
}

int nphip_settings_set_pause_draws(nphip_settings_t* s, uint64_t n, const uint64_t* draws) {
    if (n > 16) return bad_value("at most 16 pause draws");
    for (uint64_t i = 1; i < n; ++i)
        if (draws[i] <= draws[i - 1]) return bad_value("pause draws must be increasing");
    s->pause_draws.assign(draws, draws + n);
    return NPHIP_OK;
}

int nphip_settings_set_bool(nphip_settings_t* s, const char* name, int v) {
    std::string n(name);
    const bool b = v != 0;
    if (n == "check_turning") s->check_turning = b;
    else if (n == "store_mass_matrix") s->store_mass_matrix = b;
    else if (n == "use_grad_based_mass_matrix") s->use_grad_based_estimate = b;
    else if (n == "store_unconstrained") s->store_unconstrained = b;
    else if (n == "store_gradient") s->store_gradient = b;
    else if (n == "store_transformed") s->store_transformed = b;
    else if (n == "store_divergences") s->store_divergences = b;
    else if (n == "train_on_orbit") return not_available(name, "diag");  // wrapper.rs:331-343
    else if (n == "microcanonical_trajectory" || n == "exact_normal_trajectory") {
        if (b) return bad_value(std::string(name) + " is not supported by the HIP engine");
    }
    else if (n == "adapt_mass_matrix") s->adapt_mass_matrix = b;
    else if (n == "low_rank_metric") s->low_rank_metric = b;
    else return unknown_attr(name);
    return NPHIP_OK;
}

int nphip_settings_set_str(nphip_settings_t* s, const char* name, const char* v) {
    std::string n(name), val(v);
    if (n == "step_size_adapt_method") {  // wrapper.rs:344-376
        if (val == "dual_average") { s->fixed_step = false; s->adam = false; return NPHIP_OK; }
        if (val == "adam") { s->fixed_step = false; s->adam = true; return NPHIP_OK; }
        char* end = nullptr;
        double step = strtod(v, &end);
        if (end == v || *end != '\0' || !(step > 0.0))
            return bad_value("step_size_adapt_method must be a positive float when using fixed step size");
        s->fixed_step = true;
        s->adam = false;
        s->initial_step = step;
        return NPHIP_OK;
    }
    return unknown_attr(name);
}

int nphip_settings_get_f64(const nphip_settings_t* s, const char* name, double* out) {
    std::string n(name);
    if (n == "initial_step") *out = s->initial_step;
    else if (n == "target_accept") *out = s->target_accept;
    else if (n == "max_step_size") *out = s->max_step_size;
    else if (n == "max_energy_error") *out = s->max_energy_error;
    else if (n == "step_size_jitter") *out = s->jitter;
    else if (n == "early_window") *out = s->early_window;
    else if (n == "step_size_window") *out = s->step_size_window;
    else return unknown_attr(name);
    return NPHIP_OK;
}

int nphip_settings_get_u64(const nphip_settings_t* s, const char* name, uint64_t* out) {
    std::string n(name);
    if (n == "num_tune") *out = s->num_tune;
    else if (n == "num_draws") *out = s->num_draws;
    else if (n == "num_chains") *out = s->num_chains;
    else if (n == "maxdepth") *out = s->maxdepth;
    else if (n == "mindepth") *out = s->mindepth;
    else if (n == "seed") *out = s->seed;
    else if (n == "mass_matrix_switch_freq" || n == "window_switch_freq") *out = s->mass_matrix_switch_freq;
    else if (n == "early_window_switch_freq") *out = s->early_mass_matrix_switch_freq;
    else if (n == "mass_matrix_update_freq") *out = s->mass_matrix_update_freq;
    else if (n == "num_try_init") *out = s->num_try_init;
    else if (n == "check_turning") *out = s->check_turning;
    else if (n == "store_mass_matrix") *out = s->store_mass_matrix;
    else if (n == "use_grad_based_mass_matrix") *out = s->use_grad_based_estimate;
    else if (n == "store_unconstrained") *out = s->store_unconstrained;
    else if (n == "store_gradient") *out = s->store_gradient;
    else if (n == "store_divergences") *out = s->store_divergences;
    else if (n == "store_transformed") *out = s->store_transformed;
    else return unknown_attr(name);
    return NPHIP_OK;
}

static std::string jnum(double v) {
    if (!std::isfinite(v)) return "null";
    char b[64];
    snprintf(b, sizeof(b), "%.17g", v);
    return b;
}
static const char* jb(bool v) { return v ? "true" : "false"; }

int64_t nphip_settings_to_json(const nphip_settings_t* s, char* buf, int64_t cap) {
    // nested layout of DiagNutsSettings as serde writes it (field paths: wrapper.rs:217-447)
    std::string j = "{";
    j += "\"num_tune\":" + std::to_string(s->num_tune);
    j += ",\"num_draws\":" + std::to_string(s->num_draws);
    j += ",\"maxdepth\":" + std::to_string(s->maxdepth);
    j += ",\"mindepth\":" + std::to_string(s->mindepth);
    j += std::string(",\"store_gradient\":") + jb(s->store_gradient);
    j += std::string(",\"store_unconstrained\":") + jb(s->store_unconstrained);
    j += std::string(",\"store_transformed\":") + jb(s->store_transformed);
    j += ",\"max_energy_error\":" + jnum(s->max_energy_error);
    j += std::string(",\"store_divergences\":") + jb(s->store_divergences);
    j += ",\"adapt_options\":{";
    j += "\"step_size_settings\":{\"initial_step\":" + jnum(s->initial_step) + ",\"target_accept\":" + jnum(s->target_accept);
    j += ",\"jitter\":" + (s->jitter > 0 ? jnum(s->jitter) : std::string("null"));
    j += ",\"adapt_options\":{\"method\":" + (s->fixed_step ? "{\"fixed\":" + jnum(s->initial_step) + "}" : std::string(s->adam ? "\"adam\"" : "\"dual_average\""));
    j += ",\"dual_average\":{\"k\":" + jnum(s->da_k) + ",\"t0\":" + jnum(s->da_t0) + ",\"gamma\":" + jnum(s->da_gamma) +
         ",\"max_step_size\":" + jnum(s->max_step_size) + "}";
    j += ",\"adam\":{\"learning_rate\":" + jnum(s->adam_learning_rate) + "}}}";
    j += std::string(",\"mass_matrix_options\":{\"store_mass_matrix\":") + jb(s->store_mass_matrix) +
         ",\"use_grad_based_estimate\":" + jb(s->use_grad_based_estimate) + "}";
    j += ",\"early_window\":" + jnum(s->early_window) + ",\"step_size_window\":" + jnum(s->step_size_window);
    j += ",\"mass_matrix_switch_freq\":" + std::to_string(s->mass_matrix_switch_freq);
    j += ",\"early_mass_matrix_switch_freq\":" + std::to_string(s->early_mass_matrix_switch_freq);
    j += ",\"mass_matrix_update_freq\":" + std::to_string(s->mass_matrix_update_freq) + "}";
    j += std::string(",\"check_turning\":") + jb(s->check_turning);
    j += ",\"target_integration_time\":null,\"extra_doublings\":0,\"trajectory_kind\":\"euclidean\"";
    j += ",\"num_chains\":" + std::to_string(s->num_chains);
    j += ",\"seed\":" + std::to_string(s->seed);
    j += "}";
    int64_t need = (int64_t)j.size() + 1;
    if (buf && cap > 0) {
        int64_t n = need <= cap ? need - 1 : cap - 1;
        memcpy(buf, j.data(), (size_t)n);
        buf[n] = 0;
    }
    return need;
}

}  // extern "C"

// =========================================================================== models
// BridgeStan C API (bridgestan.h) as the reference calls it through the `bridgestan` crate (src/stan.rs:454-463)
typedef int (*bs_ldg_fn)(const void* model, bool propto, bool jacobian, const double* theta, double* val, double* grad, char** err);
typedef void (*bs_free_err_fn)(char* err);
struct BsAdapter { void* model; bs_ldg_fn ldg; bs_free_err_fn free_err; };
static int bs_trampoline(uint64_t, const double* x, double* grad, double* logp, void* user) {
    auto* b = static_cast<BsAdapter*>(user);
    char* err = nullptr;
    int rc = b->ldg(b->model, true, true, x, logp, grad, &err);
    if (rc != 0) { if (err && b->free_err) b->free_err(err); return 1; }  // Stan errors are recoverable
    return std::isfinite(*logp) ? 0 : 4;                                     // BadLogp, recoverable
}

// BridgeStan's expand step (src/stan.rs:473-520): bs_param_constrain with one bs_rng per chain (src/stan.rs:787-796)
typedef int (*bs_constrain_fn)(const void* model, bool include_tp, bool include_gq, const double* theta_unc, double* theta, void* rng, char** err);
typedef void* (*bs_rng_construct_fn)(unsigned int seed, char** err);
typedef void (*bs_rng_destruct_fn)(void* rng);
struct BsExpand {
    void* model; bs_constrain_fn constrain; bs_rng_construct_fn rng_new; bs_rng_destruct_fn rng_free; bs_free_err_fn free_err;
    std::vector<uint64_t> perm;   // out[j] = theta[perm[j]] (column-major blocks -> C order); empty = identity
};

// launcher of a runtime-compiled density's resident kernel (kernels.hip part 7: nphip_jit_launch)
typedef int (*nphip_jit_launch_fn)(const Args* d_args, int max_evals, void* stream, const LaunchSlice* sl, uint64_t dyn_lds_bytes);

struct nphip_model {
    int kind = 0;  // 0 fused tridiag, 1 host callback, 2 device callback, 3 runtime-compiled device density (resident kernel)
    nphip_jit_launch_fn jit_launch = nullptr;
    const void* jit_data = nullptr;
    uint64_t jit_lds_bytes = 0;   // LDS scratch per wave
    uint64_t jit_shared_bytes = 0; // LDS shared by the chains of a workgroup
    int jit_nv = 0;   // chunks of 128 dimensions per wave
    int jit_w = 1;    // waves per chain
    bool jit_lr = false;   // the library's resident kernel was built for the low-rank metric (nphip_model_jit_low_rank)
    std::shared_ptr<BsAdapter> bs;
    std::shared_ptr<BsExpand> bs_expand;
    uint64_t dim = 0;
    std::vector<double> mu, a, b;
    bool dense = false;              // kind 2 driven by the engine's own gradient kernels (nphip_model_dense_gaussian): prec = P [dim][dim]
    std::shared_ptr<std::vector<double>> prec;
    nphip_raw_logp_fn host_fn = nullptr;
    nphip_device_logp_fn dev_fn = nullptr;
    void* user = nullptr;
    int n_threads = 0;
    int init_kind = 0;
    std::vector<double> init_points;
    uint64_t n_init_points = 0;
    // expand step (src/pymc.rs:64-95): host row function or batched device function
    uint64_t expanded_dim = 0;
    nphip_raw_expand_fn expand_fn = nullptr;
    nphip_device_expand_fn expand_dev_fn = nullptr;
    void* expand_user = nullptr;
};

extern "C" {

nphip_model_t* nphip_model_tridiag_gaussian(uint64_t dim, const double* mu, const double* diag, const double* offdiag) {
    if (dim == 0 || !diag) { set_error("tridiag model needs dim > 0 and a diagonal"); return nullptr; }
    auto* m = new nphip_model();
    m->kind = 0; m->dim = dim;
    m->mu.assign(dim, 0.0); if (mu) m->mu.assign(mu, mu + dim);
    m->a.assign(diag, diag + dim);
    m->b.assign(dim, 0.0); if (offdiag && dim > 1) std::copy(offdiag, offdiag + dim - 1, m->b.begin());
    return m;
}
nphip_model_t* nphip_model_host_callback(uint64_t dim, nphip_raw_logp_fn fn, void* user_data, int n_threads) {
    if (dim == 0 || !fn) { set_error("host callback model needs dim > 0 and a function"); return nullptr; }
    auto* m = new nphip_model();
    m->kind = 1; m->dim = dim; m->host_fn = fn; m->user = user_data; m->n_threads = n_threads;
    return m;
}
nphip_model_t* nphip_model_bridgestan(uint64_t dim, void* bs_model, void* log_density_gradient, void* free_error_msg, int n_threads) {
    if (dim == 0 || !bs_model || !log_density_gradient) { set_error("bridgestan model needs dim > 0, a model handle and bs_log_density_gradient"); return nullptr; }
    auto* m = new nphip_model();
    m->kind = 1; m->dim = dim; m->n_threads = n_threads;
    m->bs = std::make_shared<BsAdapter>(BsAdapter{bs_model, (bs_ldg_fn)log_density_gradient, (bs_free_err_fn)free_error_msg});
    m->host_fn = bs_trampoline;
    m->user = m->bs.get();
    return m;
}
nphip_model_t* nphip_model_device_callback(uint64_t dim, nphip_device_logp_fn fn, void* user_data) {
    if (dim == 0 || !fn) { set_error("device callback model needs dim > 0 and a function"); return nullptr; }
    auto* m = new nphip_model();
    m->kind = 2; m->dim = dim; m->dev_fn = fn; m->user = user_data;
    return m;
}
// Dense-precision Gaussian: the model's evaluation is the engine's own fp64 MFMA GEMM (kernels.hip part 10, dense_tile.h) behind the
// device-callback path — `dev_fn` is set by the sampler once the matrix is on its device (setup()).
nphip_model_t* nphip_model_dense_gaussian(uint64_t dim, const double* mu, const double* P) {
    if (dim == 0 || !P) { set_error("dense Gaussian model needs dim > 0 and a precision matrix"); return nullptr; }
    for (uint64_t i = 0; i < dim; ++i)
        for (uint64_t j = 0; j < i; ++j)
            if (!(P[i * dim + j] == P[j * dim + i])) {
                set_error("the precision matrix must be symmetric (P[" + std::to_string(i) + "][" + std::to_string(j) + "] != P[" + std::to_string(j) + "][" + std::to_string(i) + "])");
                return nullptr;
            }
    auto* m = new nphip_model();
    m->kind = 2; m->dense = true; m->dim = dim;
    m->mu.assign(dim, 0.0); if (mu) m->mu.assign(mu, mu + dim);
    m->prec = std::make_shared<std::vector<double>>(P, P + dim * dim);
    return m;
}
nphip_model_t* nphip_model_jit_density(uint64_t dim, void* launch_fn, int nv, const void* data_device, uint64_t lds_bytes_per_chain,
                                       uint64_t lds_bytes_shared, int waves_per_chain) {
    const int w = waves_per_chain > 0 ? waves_per_chain : 1;
    if (dim == 0 || !launch_fn) { set_error("a runtime-compiled density needs dim > 0 and its launcher"); return nullptr; }
    if (!(w == 1 || w == 2 || w == 4)) { set_error("a runtime-compiled density runs with 1, 2 or 4 waves per chain"); return nullptr; }
    if (dim > 1024 || nv < 1 || nv > 8 || (uint64_t)nv * 128 * w < dim || (uint64_t)nv != ((dim + 127) / 128 + w - 1) / w) {
        set_error("the resident kernel of a runtime-compiled density holds up to 1024 dimensions (nv chunks of 128 per wave, nv = "
                  "ceil(ceil(dim / 128) / waves)); larger models run through the batched device callback of the same library");
        return nullptr;
    }
    if (lds_bytes_per_chain % 8 != 0 || lds_bytes_shared % 8 != 0) { set_error("LDS scratch sizes must be multiples of 8 bytes"); return nullptr; }
    // LDS of the launch must fit the CU's 160 KB.  One wave per chain: four chains per workgroup — control blocks, reduction
    // scratch, the four rings (4 x 4 KB x nv each), the leaf's position and gradient rows (2 x 1 KB x nv per chain) and the
    // density's scratch.  Several waves per chain: one chain per workgroup, a ring per wave.
    const uint64_t cpb = w == 1 ? 4 : 1, waves = w == 1 ? 4 : (uint64_t)w, ld = (uint64_t)nv * 128 * w;
    const uint64_t fixed = waves * 1200 + 1024 + waves * (uint64_t)nv * 4096 + 64 + 16 * (uint64_t)w * nv + cpb * 2 * ld * 8;
    if (fixed + cpb * lds_bytes_per_chain + lds_bytes_shared > 160 * 1024) {
        set_error("LDS scratch of the density does not fit beside the kernel's own (" + std::to_string(fixed) + " bytes fixed, " +
                  std::to_string(cpb * lds_bytes_per_chain) + " requested for the chains of a workgroup + " + std::to_string(lds_bytes_shared) + " shared, 163840 per CU)");
        return nullptr;
    }
    auto* m = new nphip_model();
    m->kind = 3; m->dim = dim; m->jit_launch = (nphip_jit_launch_fn)launch_fn; m->jit_nv = nv; m->jit_w = w; m->jit_data = data_device;
    m->jit_lds_bytes = lds_bytes_per_chain;
    m->jit_shared_bytes = lds_bytes_shared;
    return m;
}
int nphip_model_jit_low_rank(nphip_model_t* m, int capable) {
    if (!m || m->kind != 3) { set_error("nphip_model_jit_low_rank: not a runtime-compiled density"); return NPHIP_ERR; }
    m->jit_lr = capable != 0;
    return NPHIP_OK;
}

int nphip_model_set_init(nphip_model_t* m, int kind, const double* points, uint64_t n_points) {
    if (kind < 0 || kind > 2) return bad_value("init kind must be 0, 1 or 2");
    if (kind == 2 && (!points || n_points == 0)) return bad_value("explicit init needs points");
    m->init_kind = kind;
    m->init_points.clear();
    m->n_init_points = 0;
    if (kind == 2) { m->init_points.assign(points, points + n_points * m->dim); m->n_init_points = n_points; }
    return NPHIP_OK;
}
int nphip_model_set_expand(nphip_model_t* m, uint64_t expanded_dim, nphip_raw_expand_fn fn, void* user_data) {
    if (!fn || expanded_dim == 0) return bad_value("expand needs a function and expanded_dim > 0");
    m->expanded_dim = expanded_dim; m->expand_fn = fn; m->expand_dev_fn = nullptr; m->expand_user = user_data;
    m->bs_expand.reset();
    return NPHIP_OK;
}
int nphip_model_set_bridgestan_expand(nphip_model_t* m, uint64_t expanded_dim, void* bs_model, void* param_constrain, void* rng_construct,
                                      void* rng_destruct, void* free_error_msg, const uint64_t* perm) {
    if (!bs_model || !param_constrain || !rng_construct || !rng_destruct || expanded_dim == 0)
        return bad_value("the BridgeStan expand step needs a model handle, bs_param_constrain, bs_rng_construct, bs_rng_destruct and expanded_dim > 0");
    auto e = std::make_shared<BsExpand>();
    e->model = bs_model; e->constrain = (bs_constrain_fn)param_constrain; e->rng_new = (bs_rng_construct_fn)rng_construct;
    e->rng_free = (bs_rng_destruct_fn)rng_destruct; e->free_err = (bs_free_err_fn)free_error_msg;
    if (perm) {
        e->perm.assign(perm, perm + expanded_dim);
        std::vector<uint8_t> seen(expanded_dim, 0);
        for (uint64_t j : e->perm) {
            if (j >= expanded_dim || seen[j]) return bad_value("perm must be a permutation of 0 .. expanded_dim - 1");
            seen[j] = 1;
        }
    }
    m->expanded_dim = expanded_dim; m->expand_fn = nullptr; m->expand_dev_fn = nullptr; m->expand_user = nullptr;
    m->bs_expand = e;
    return NPHIP_OK;
}
int nphip_model_set_device_expand(nphip_model_t* m, uint64_t expanded_dim, nphip_device_expand_fn fn, void* user_data) {
    if (!fn || expanded_dim == 0) return bad_value("expand needs a function and expanded_dim > 0");
    m->expanded_dim = expanded_dim; m->expand_dev_fn = fn; m->expand_fn = nullptr; m->expand_user = user_data;
    m->bs_expand.reset();
    return NPHIP_OK;
}
uint64_t nphip_model_expanded_dim(const nphip_model_t* m) { return m->expanded_dim; }
uint64_t nphip_model_dim(const nphip_model_t* m) { return m->dim; }
void nphip_model_free(nphip_model_t* m) { delete m; }

uint64_t nphip_abi_struct_size(int which) {
    return which == 0 ? sizeof(nphip_launch_t) : (which == 1 ? sizeof(nphip_chain_progress_t) : 0);
}

void nphip_launch_defaults(nphip_launch_t* l) {
    memset(l, 0, sizeof(*l));
    l->store_draws = 1;
}

}  // extern "C"

// =========================================================================== sampler
namespace {

// tiny persistent pool for the host-callback flavour: rows of the batch are evaluated
// concurrently, exactly as the reference evaluates chains on `cores` threads (sample.py:856-857)
// CPUs this process may actually use: the affinity mask and the cgroup CPU quota (a GPU box can show 256 logical CPUs and grant 16)
static int usable_cores() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t cs;
    CPU_ZERO(&cs);
    if (sched_getaffinity(0, sizeof(cs), &cs) == 0 && CPU_COUNT(&cs) > 0) n = std::min(n, (int)CPU_COUNT(&cs));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
        char q[64];
        long per = 0;
        if (fscanf(f, "%63s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min(n, (int)((atol(q) + per - 1) / per));
        fclose(f);
    } else {
        long quota = -1, per = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &per) != 1) per = 0; fclose(g); }
        if (quota > 0 && per > 0) n = std::min(n, (int)((quota + per - 1) / per));
    }
    return std::max(1, n);
}

// Is [p, p + bytes) mapped readable and writable in THIS process (/proc/self/maps)?  Fine-grained device memory is, on a large-BAR
// system whose runtime maps it for the CPU; asking first means a box where it is not gets host-memory staging instead of a fault.
static bool cpu_writable(const void* p, size_t bytes) {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return false;
    const unsigned long long lo = (unsigned long long)(uintptr_t)p, hi = lo + bytes;
    char line[512];
    bool ok = false;
    while (fgets(line, sizeof(line), f)) {
        unsigned long long a = 0, b = 0;
        char perms[8] = {0};
        if (sscanf(line, "%llx-%llx %7s", &a, &b, perms) == 3 && a <= lo && hi <= b) { ok = perms[0] == 'r' && perms[1] == 'w'; break; }
    }
    fclose(f);
    return ok;
}

// Evaluates the rows of a batch on `n` threads: the caller and n - 1 workers (the reference: one rayon worker per chain,
// src/pymc.rs:197-215).  A batch of a cheap model is a few microseconds of work, so the workers do not sleep between batches:
// they spin on a generation counter (a dispatch costs a cache-line transfer, not a futex wake-up of tens of microseconds) and
// only go to sleep on a condition variable after ~200 us without work.
struct RowPool {
    std::vector<std::thread> threads;
    alignas(64) std::atomic<uint64_t> gen{0};
    alignas(64) std::atomic<uint64_t> next{0};
    alignas(64) std::atomic<int> acks{0};
    alignas(64) uint64_t n_rows = 0;
    uint64_t chunk = 1;
    const std::function<void(uint64_t)>* job = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> sleepers{0};
    std::atomic<bool> stop{false};
    static constexpr int kSpin = 1 << 13;
    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    int active_workers = 0;   // workers taking part in the current batch (the others only acknowledge it)
    explicit RowPool(int n) {
        for (int i = 1; i < n; ++i) threads.emplace_back([this, i] { loop(i - 1); });
    }
    int size() const { return (int)threads.size() + 1; }
    ~RowPool() {
        { std::lock_guard<std::mutex> lk(mu); stop.store(true); }
        cv.notify_all();
        for (auto& t : threads) t.join();
    }
    void work() {
        for (;;) {
            const uint64_t r0 = next.fetch_add(chunk, std::memory_order_relaxed);
            if (r0 >= n_rows) break;
            const uint64_t r1 = std::min(n_rows, r0 + chunk);
            for (uint64_t r = r0; r < r1; ++r) (*job)(r);
        }
    }
    void loop(int index) {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (gen.load() == seen) {
                if (stop.load(std::memory_order_relaxed)) return;
                if (++spins > kSpin) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1);
                    cv.wait(lk, [&] { return stop.load() || gen.load() != seen; });
                    sleepers.fetch_sub(1);
                    spins = 0;
                } else {
                    relax();
                }
            }
            if (stop.load(std::memory_order_relaxed)) return;
            seen += 1;   // (a new batch starts only after every worker acknowledged the previous one)
            if (index < active_workers) work();
            acks.fetch_add(1, std::memory_order_release);
        }
    }
    // `use` threads (the caller included) share the rows; 0 = all of them
    void run(uint64_t rows, const std::function<void(uint64_t)>& f, int use = 0) {
        if (use <= 0 || use > size()) use = size();
        if (threads.empty() || rows < 2 || use < 2) { for (uint64_t r = 0; r < rows; ++r) f(r); return; }
        job = &f;
        n_rows = rows;
        active_workers = use - 1;
        chunk = std::max<uint64_t>(1, rows / (4 * (uint64_t)use));
        next.store(0, std::memory_order_relaxed);
        acks.store(0, std::memory_order_relaxed);
        gen.fetch_add(1);
        if (sleepers.load() > 0) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
        work();
        const int want = (int)threads.size();
        while (acks.load(std::memory_order_acquire) != want) relax();
    }
};

}  // namespace

struct nphip_sampler {
    nphip_settings set;
    nphip_model model;
    nphip_launch_t launch;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Args args;
    Args* d_args = nullptr;  // device copy, read by the kernels through the constant address space
    int W = 1;
    bool fused = true;
    bool dens = false;   // runtime-compiled device density: the resident kernel of the model's own library, driven like a fused model
    bool zero_copy = false;  // host callbacks: staging buffers in pinned host memory, no copies
    uint64_t n = 0, T = 0, dim = 0;
    std::vector<void*> allocs;
    std::vector<void*> pinned;
    std::vector<double> bpad, bshpad;  // staging of the padded off-diagonal (must outlive the async copy)
    unsigned long long* h_counters = nullptr;  // pinned
    double *h_q = nullptr, *h_g = nullptr, *h_u = nullptr;  // pinned staging (host callback)
    int64_t* h_code = nullptr;
    std::unique_ptr<RowPool> pool;
    struct RowRange { uint64_t lo, cnt; };
    void eval_rows(uint64_t lo, uint64_t cnt);
    void eval_ranges(const RowRange* rg, int nr);
    int host_cores = 1;
    double row_ns = 0.0;     // cheapest timed batch so far, per row
    int timed_batches = 0;
    int eval_threads = 1;

    std::thread th;
    std::mutex mu;       // state
    std::mutex mu_run;   // held by the driver for one launch iteration; by readers while copying
    std::condition_variable cv;
    bool want_pause = false, want_abort = false;
    bool finished = false, failed = false, thread_done = false;
    std::string error;
    std::chrono::steady_clock::time_point t_start;
    std::atomic<double> seconds{0.0};
    std::atomic<uint64_t> launches{0};

    template <class Tp>
    bool dalloc(Tp** p, size_t count, int fill = 0) {
        void* d = nullptr;
        size_t bytes = count * sizeof(Tp);
        if (bytes == 0) bytes = 8;
        if (!hip_ok(hipMalloc(&d, bytes), "hipMalloc")) return false;
        allocs.push_back(d);
        if (!hip_ok(hipMemsetAsync(d, fill, bytes, stream), "hipMemset")) return false;
        *p = reinterpret_cast<Tp*>(d);
        return true;
    }
    template <class Tp>
    bool palloc(Tp** p, size_t count) {
        void* h = nullptr;
        if (!hip_ok(hipHostMalloc(&h, count * sizeof(Tp) + 8, hipHostMallocCoherent), "hipHostMalloc")) return false;
        pinned.push_back(h);
        memset(h, 0, count * sizeof(Tp) + 8);
        *p = reinterpret_cast<Tp*>(h);
        return true;
    }

    // dense-precision Gaussian (model.dense): the padded matrix and mean on this device; the evaluation = two kernel launches
    struct DenseDev { const double* Pp = nullptr; const double* mu = nullptr; int64_t KP = 0; int W = 1; } dense_dev;
    // ... and its resident form (kernels.hip: k_advance<..., DENSEG>): the register-resident leaf with the launch-wide GEMM as the
    // evaluation in its middle — driven like a fused model (a launch runs `evals_per_launch` evaluations of every chain).  A launch whose
    // roll call fails (the device was busy: not every chain resident) has touched nothing and is simply repeated; after three in a row the
    // job goes on with a launch per evaluation.
    bool dg = false, dg_fell_back = false;
    int dg_fail = 0;
    unsigned dg_launch_id = 0;
    unsigned long long dg_fail_seen = 0;
    volatile unsigned long long* h_dg_abort = nullptr;   // pinned [2]: the rendezvous-timed-out word after each launch
    bool dg_check(int slot);
    static int dense_dev_fn(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp, void* stream, void* user) {
        const DenseDev* d = (const DenseDev*)user;
        return launch_dense_grad(q, d->Pp, d->mu, grad, logp, (int64_t)n_chains, (int64_t)dim, d->KP, d->W, (hipStream_t)stream) == hipSuccess ? 0 : -1;
    }
    // hand-ins of the host-driven adaptation hook (nphip_sampler_set_metric / nphip_sampler_resume_at): grow-only device buffers owned by the
    // sampler — a hipMalloc / hipFree pair per call synchronises the whole device (hipFree), i.e. the chains that run while one is handed in
    struct Scratch { void* p = nullptr; size_t bytes = 0; };
    Scratch hand_in[5];   // chain list, taken counter, sigma^2 / positions, V, lambda
    void* hand_in_buf(int which, size_t bytes) {
        Scratch& b = hand_in[which];
        if (b.bytes >= bytes && b.p) return b.p;
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.bytes = 0;
        const size_t want = std::max<size_t>(256, bytes + bytes / 2);
        if (!hip_ok(hipMalloc(&b.p, want), "hipMalloc")) { b.p = nullptr; return nullptr; }
        b.bytes = want;
        return b.p;
    }
    bool setup();
    void run();
    bool manual = false;
    int manual_have = 0;
    // kernel timing (nphip_sampler_step with kernel_ms): one pair of HIP events around every launch, read after the last one
    std::vector<hipEvent_t> tev;
    size_t timed_launches = 0;
    double* kernel_ms_acc = nullptr;  // when set, iterations time their kernel with HIP events
    // fused models: the next launch is enqueued before the counters of the previous one are looked at (no idle device between
    // two launches); the "all chains done" / error check therefore runs one launch behind
    hipEvent_t ev_f[2] = {nullptr, nullptr};
    uint64_t fused_k = 0;
    bool check_counters(int slot, bool& all_done);
    bool launch_kernel(bool fused_, int have);
    bool iteration_fused(bool& all_done);
    bool iteration_graph(bool& all_done);
    hipGraphExec_t cb_graph = nullptr;
    hipEvent_t cb_ev[2] = {nullptr, nullptr};
    int cb_graph_steps = 0;
    uint64_t cb_replays = 0;
    uint64_t cb_polls = 0;
    bool iteration_callback(bool& all_done, int& have);
    bool iteration_callback_groups(bool& all_done, int& have);
    // host callbacks, zero-copy staging: chains in groups, each on its own stream, completion by a flag in pinned memory
    // (no stream synchronisation): the kernel of one group runs while the host evaluates the rows of the other
    static constexpr int kMaxGroups = 8;
    int n_groups = 0;
    hipStream_t grp_stream[kMaxGroups] = {};
    uint64_t grp_lo[kMaxGroups + 1] = {};
    unsigned grp_seq[kMaxGroups] = {};
    bool grp_primed = false;
    int cb_groups = 0;   // device callbacks in groups of chains (launch.host_groups >= 2): group g's (kernel, callback) on its own stream
    volatile unsigned long long* h_grp_flag = nullptr;  // pinned [groups][4]
    // Resident launches (kernels.hip: REMOTE): a group's kernel stays on the device for `persist_evals` evaluations; an evaluation
    // is a rendezvous — the kernel publishes its positions and the sequence number, the host evaluates the rows and answers
    // with one word in pinned memory.  grp_seq[g] is then the sequence number the host waits for next.
    bool remote = false;
    bool remote_fell_back = false;
    // resident launches on a large-BAR system: the results (gradient, logp, code) and the go words live in fine-grained DEVICE
    // memory that the host writes through the PCIe BAR (write-combining stores: 64 rows in ~0.25 us) — the kernel then polls
    // and reads local memory instead of host memory over PCIe (a read round trip each).  The callback itself still writes
    // cached host rows (it may read its own output back; a CPU load from the BAR costs 1.2 us).
    bool bar = false;
    double *b_g = nullptr, *b_u = nullptr;
    int64_t* b_code = nullptr;
    volatile unsigned long long* b_go = nullptr;
    void publish_rows(uint64_t lo, uint64_t cnt);
    int remote_nv = 0;
    int persist_evals = 256;
    volatile unsigned long long* h_grp_go = nullptr;    // pinned [groups][8]
    bool grp_running[kMaxGroups] = {};
    int grp_evals[kMaxGroups] = {};
    bool materialise = false;   // the next callback launches follow resident ones (LaunchSlice::materialise)
    int64_t fall_back_after = 0; // tests (launch.host_persist = -N): leave the resident mode after N evaluations, as a failed roll call would
    int64_t remote_evals = 0;
    bool bar_used = false;   // (diagnostics: the BAR copy was in use at some point of the job)
    double t_wait_ns = 0.0, t_eval_ns = 0.0;   // NPHIP_TIMING=1: where the driver thread's time goes (printed at the end of the job)
    unsigned remote_launch_id = 0;
    bool remote_fresh = false;  // the running launch has not published anything yet (its roll call may still fail)
    int remote_next = 0;        // round-robin start of the poll
    bool launch_remote_all();
    int poll_remote(int g);
    int wait_remote(int only);
    void answer_group(int g, bool last);
    bool drain_groups();
    bool remote_fall_back();
    bool iteration_remote(bool& all_done);
    bool launch_group(int g, int have);
    bool wait_group(int g);
    bool iteration_pipelined(bool& all_done);
    bool sync_all() {
        if (remote && !drain_groups()) return false;
        bool ok = hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
        for (int g = 0; g < n_groups; ++g) ok = hip_ok(hipStreamSynchronize(grp_stream[g]), "hipStreamSynchronize") && ok;
        return ok;
    }
    void fail(const std::string& msg) {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        error = msg;
    }
    std::string chain_error_message();
    void release() {
        for (void* d : allocs) (void)hipFree(d);
        allocs.clear();
        for (auto& b : hand_in) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
        for (void* h : pinned) (void)hipHostFree(h);
        pinned.clear();
        for (auto& e : tev) (void)hipEventDestroy(e);
        tev.clear();
        for (auto& e : ev_f) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (cb_graph) { (void)hipGraphExecDestroy(cb_graph); cb_graph = nullptr; }
        for (auto& e : cb_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (auto& gs : grp_stream) if (gs) { (void)hipStreamDestroy(gs); gs = nullptr; }
        if (own_stream && stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    }
};

// Waves per chain: a function of the dimension ONLY.  The summation geometry (hence every float of a chain) depends on
// it, and a chain's result must not depend on how many other chains run with it or on how they are sharded.  (Measured:
// with 64..256 chains at D = 1000, waves_per_chain = 4 is 20 % faster — available through nphip_launch_t, not chosen
// behind the user's back; 2 waves per chain are slower than 1.)
// Leapfrogs per chain per launch of a fused model: about 10 ms of kernel.  A launch boundary costs every chain a flush and a
// reload of its on-chip state, and the device the tail of the chain that met the most draw ends (measured, bench.py: D = 1000
// 201 / 208 / 210 / 213 M leapfrogs/s with 256 / 512 / 1024 / 2048 per launch; D = 10 000 10.6 / 12.2 / 13.0 with 32 / 128 / 512
// and nothing beyond).  Results do not depend on it.
static int default_evals_per_launch(uint64_t dim) { return dim <= 1024 ? 2048 : (dim <= 4096 ? 1024 : 512); }

static int choose_waves(uint64_t dim) {
    if (dim <= 1024) return 1;
    if (dim <= 2048) return 2;
    // 2048 < D <= 4096: register kernels with the LDS ring; 4096 < D <= 10240: lean register kernels, state in VGPRs + AGPRs.
    // Measured (profiles/r2_grid_b_lean_w8_vs_w4.txt, 1024 chains): 4 waves per chain beat 8 from D = 7000 up (10.2 vs 6.9 M
    // leapfrogs/s at D = 10 000: at 8 waves the 256-VGPR budget spills the state) and tie below.
    // 10240 < D <= 12288 (round 6): the same kernels with 21 .. 24 chunks per wave — they spill, and run 1.3 - 1.6 x the memory-resident ones
    if (dim <= 12288) return 4;
    return 8;  // memory-resident kernels; measured at D = 10 000: W = 8 (5.5 M leapfrogs/s) beats 4 (5.0) and 16 (3.6)
}

// Device-callback models run on the launch-per-evaluation kernels, whose fused leaf holds up to two chunks of 128 dimensions per
// wave in registers (kernels.hip: leaf_cb, NPHIP_CB_CHUNKS): the fewest waves per chain that keep a row inside it — a function of
// the dimension alone, like choose_waves (a chain's floats depend on the number of waves that sum over its row).  Measured at
// D = 1000 x 1024 chains (profiles/r4_callback_kernels.txt): 4 waves x 2 chunks 33.5 us per launch, 2 waves x 4 chunks 36.6.
static int choose_waves_callback(uint64_t dim) {
    const uint64_t nch = (dim + 127) / 128;
    for (int w = 1; w <= 16; w *= 2)
        if (nch <= 2u * (uint64_t)w) return w;
    return 16;
}

bool nphip_sampler::setup() {
    HIP_TRY(hipSetDevice(device));
    if (launch.stream) { stream = (hipStream_t)launch.stream; own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
    dim = model.dim;
    n = launch.n_local_chains ? launch.n_local_chains : set.num_chains;
    T = set.num_tune + set.num_draws;
    fused = model.kind == 0;
    dens = model.kind == 3;
    // (the dense Gaussian up to 1024 dimensions: one wave per chain — the geometry of its resident kernel, whichever form runs)
    W = dens ? model.jit_w : (launch.waves_per_chain ? launch.waves_per_chain : (model.kind == 2 && !(model.dense && dim <= 1024) ? choose_waves_callback(dim) : choose_waves(dim)));
    const bool lrm = set.low_rank_metric;
    if (dens && lrm != model.jit_lr) {
        set_error(lrm ? "this library's resident kernel was not built for the low-rank metric (compile it with -DNPHIP_JIT_LR=1 and say so with nphip_model_jit_low_rank), or use its batched device callback"
                      : "this library's resident kernel was built for the low-rank metric: the job must set low_rank_metric");
        return false;
    }
    // (P-slots carry the velocity as a third vector.  Round 4: fused models keep the register-resident leaf under the metric
    //  (kernels.hip: Machine<..., LR>) in the geometries choose_waves() picks up to D = 4096 — one wave per chain, or two / four
    //  with 5..8 chunks per wave; everything else runs the memory-resident kernels)
    if (lrm && !dens) {
        const uint64_t per_wave = ((dim + 127) / 128 + (uint64_t)W - 1) / (uint64_t)W;
        const bool lr_reg = model.kind == 0 && ((W == 1 && dim <= 1024) || ((W == 2 || W == 4) && per_wave >= 5 && per_wave <= 8));
        if (!lr_reg) launch.no_register_kernel = 1;
    }
    if (dens && set.store_divergences) {
        set_error("store_divergences needs the pre-step state in memory: use the batched device callback of the density's library (launch per evaluation)");
        return false;
    }
    if (!(W == 1 || W == 2 || W == 4 || W == 8 || W == 16)) { set_error("waves_per_chain must be 1, 2, 4, 8 or 16"); return false; }
    if (n == 0 || dim == 0) { set_error("need at least one chain and one dimension"); return false; }
    if (launch.chain_offset + n > set.num_chains) {
        // a mis-sharded launch would reuse global chain ids, i.e. RNG streams, without any visible symptom
        set_error("chain_offset + n_local_chains exceeds settings.num_chains (" + std::to_string(launch.chain_offset) + " + " + std::to_string(n) +
                  " > " + std::to_string(set.num_chains) + ")");
        return false;
    }

    memset(&args, 0, sizeof(args));
    DevSettings& s = args.s;
    s.seed = set.seed;
    s.num_tune = (int64_t)set.num_tune; s.num_draws = (int64_t)set.num_draws;
    s.maxdepth = (int64_t)set.maxdepth; s.mindepth = (int64_t)set.mindepth;
    s.check_turning = set.check_turning; s.use_grad_based = set.use_grad_based_estimate;
    s.adapt_mass_matrix = set.adapt_mass_matrix; s.fixed_step_size = set.fixed_step;
    s.max_energy_error = set.max_energy_error;
    // window bounds
    s.early_end = (int64_t)std::ceil((double)set.num_tune * set.early_window);
    {
        uint64_t ssw = (uint64_t)std::ceil((double)set.num_tune * set.step_size_window);
        s.final_window = (int64_t)((set.num_tune > ssw ? set.num_tune - ssw : 0) + 1);
    }
    s.mm_switch_freq = (int64_t)set.mass_matrix_switch_freq;
    s.early_mm_switch_freq = (int64_t)set.early_mass_matrix_switch_freq;
    s.mm_update_freq = (int64_t)set.mass_matrix_update_freq;
    s.initial_step = set.initial_step; s.target_accept = set.target_accept;
    s.jitter = set.jitter; s.max_step_size = set.max_step_size;
    s.adapt_adam = set.adam ? 1 : 0; s.adam_lr = set.adam_learning_rate;
    s.da_k = set.da_k; s.da_t0 = set.da_t0; s.da_gamma = set.da_gamma;
    s.init_kind = model.init_kind; s.num_try_init = (int32_t)set.num_try_init;
    s.store_draws = launch.store_draws; s.store_gradient = set.store_gradient;
    s.store_mass_matrix = set.store_mass_matrix; s.store_divergences = set.store_divergences;
    // the adaptation hook is driven by the caller (nphip_sampler_waiting / nphip_sampler_resume_at need manual mode): with a
    // driver thread nobody would ever resume a chain that stopped, and wait() would never return.  A pause at or after the
    // end of warm-up would re-enter the initial-point sequence (fresh mass matrix, step-size search) in the sampling phase.
    if (!set.pause_draws.empty()) {
        if (!launch.manual) { set_error("pause draws need a manual-mode sampler (launch.manual = 1): the caller resumes the chains"); return false; }
        for (uint64_t pd : set.pause_draws)
            if (pd == 0 || pd >= set.num_tune) { set_error("pause draws must lie inside the warm-up (0 < draw < num_tune)"); return false; }
    }
    s.n_pause = (int32_t)set.pause_draws.size();
    for (size_t i = 0; i < set.pause_draws.size(); ++i) s.pause_draws[i] = (int64_t)set.pause_draws[i];

    args.n_chains = (int64_t)n;
    args.chain_offset = (int64_t)launch.chain_offset;
    args.dim = (int64_t)dim;
    args.ld = (int64_t)((dim + 127) / 128 * 128);
    // register-resident kernels with several waves per chain (1024 < D <= 4096): every wave owns the same number of
    // chunks, so the leading dimension is padded to a multiple of 128 * W (pads are exact zeros in every reduction)
    const bool fused_model = (model.kind == 0);
    int reg_multi = 0;
    if (fused_model && (W == 2 || W == 4) && !launch.no_register_kernel) {
        const int64_t per_wave = ((int64_t)((dim + 127) / 128) + W - 1) / W;
        if (per_wave >= 1 && per_wave <= 8) { reg_multi = (int)per_wave; args.ld = per_wave * W * 128; }
    }
    // lean register-resident kernels (8 waves per chain, up to 10 chunks per wave: D <= 10240 — the rows of config 5): state in
    // VGPRs, sigma^2 in LDS, merge operands streamed (kernels.hip: leaf_lean).  Same padding rule as above.
    int lean_nc = 0;
    if (fused_model && (W == 8 || W == 4) && !launch.no_register_kernel) {
        const int64_t per_wave = ((int64_t)((dim + 127) / 128) + W - 1) / W;
        if (W == 8 && per_wave >= 1 && per_wave <= 10) { lean_nc = (int)per_wave; args.ld = per_wave * W * 128; }
        // (experimental geometry: 4 waves per chain with the state spread over VGPRs + AGPRs, one wave per SIMD)
        if (W == 4 && per_wave > 8 && per_wave <= 24) { lean_nc = (int)per_wave; args.ld = per_wave * W * 128; }   // (21 .. 24, round 6: the build spills and still beats the memory-resident kernels)
    }
    // host-callback models with several waves per chain (1024 < D <= 4096): the same padding, so that resident launches can
    // run them on the register-resident leaf (8 chunks per wave at most)
    if ((model.kind == 1 || model.kind == 3) && (W == 2 || W == 4)) {
        const int64_t per_wave = ((int64_t)((dim + 127) / 128) + W - 1) / W;
        if (per_wave >= 1 && per_wave <= 8) args.ld = per_wave * W * 128;
    }
    args.cap = (int32_t)set.maxdepth;
    args.npslots = num_pslots(args.cap);
    args.nqpool = num_qpool(args.cap);
    const size_t ld = (size_t)args.ld;
    // register-resident specialisation, one wave per chain: state in VGPRs (dim <= 1024, one instantiation per chunk count)
    args.reg_nv = lean_nc ? lean_nc : reg_multi;
    args.lean = lean_nc ? 1 : 0;
    // (store_divergences does not change the choice: the register kernels rebuild the pre-step state of a failed leapfrog in
    //  the rare path — kernels.hip: replay_divergence)
    if (fused && W == 1 && !launch.no_register_kernel) {
        const int nchunks = (int)(args.ld / 128);  // one kernel instantiation per exact chunk count (straight-line code)
        if (nchunks <= 8 || (getenv("NPHIP_DEV_W1_WIDE") && nchunks <= 12)) args.reg_nv = nchunks;   // (9 .. 12: developer libraries only, kernels.hip: Machine::NORING)
    }
    // memory-resident fused kernel, one wave per chain (store_divergences, no_register_kernel): cache the cursor's
    // (sigma^2, grad, p, rho) in VGPRs between leaves.  With more waves per chain the cache costs occupancy (measured).
    args.stream_cache = (fused && !args.reg_nv && W == 1 && !launch.no_stream_cache && args.ld / 128 <= 8 && !lrm) ? 1 : 0;
    args.pvec = lrm ? 3 : 2;
    args.lr_on = lrm ? 1 : 0;
    s.low_rank_metric = lrm ? 1 : 0;

    // D > 4096 (one chain per CU): sigma^2 of the chain in LDS instead of one more HBM stream per pass
    args.sig_lds = (fused && !args.reg_nv && W >= 8 && args.ld * 8 <= 128 * 1024 && !launch.no_stream_cache) ? 1 : 0;

    if (!dalloc(&args.ctl, n)) return false;
    if (!dalloc(&args.qpool, n * args.nqpool * 2 * ld)) return false;
    if (!dalloc(&args.pslots, n * args.npslots * (size_t)args.pvec * ld)) return false;
    if (lrm && (!dalloc(&args.lr_V, n * (size_t)kLrMax * ld) || !dalloc(&args.lr_lam, n * (size_t)kLrMax) || !dalloc(&args.lr_std, n * ld))) return false;
    if (lrm && (!dalloc(&args.st_V, n * (size_t)kLrMax * ld) || !dalloc(&args.st_lam, n * (size_t)kLrMax) || !dalloc(&args.st_sig2, n * ld))) return false;
    if (!dalloc(&args.sig2, n * ld)) return false;
    if (!dalloc(&args.est, n * 8 * ld)) return false;
    if (!dalloc(&args.counters, 4)) return false;
    if (fused) {
        double *mu = nullptr, *a = nullptr, *b = nullptr;
        if (!dalloc(&mu, ld) || !dalloc(&a, ld) || !dalloc(&b, ld)) return false;
        HIP_TRY(hipMemcpyAsync(mu, model.mu.data(), dim * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(a, model.a.data(), dim * 8, hipMemcpyHostToDevice, stream));
        // b[dim-1 ..] = -0.0: lets the register kernel add boundary terms unconditionally (t + (-0.0) == t).
        // Same stream as the zero-fill of the buffer (the engine stream does not synchronise with the null stream).
        bpad.assign(ld, -0.0);
        std::copy(model.b.begin(), model.b.begin() + (dim > 0 ? dim - 1 : 0), bpad.begin());
        HIP_TRY(hipMemcpyAsync(b, bpad.data(), ld * 8, hipMemcpyHostToDevice, stream));
        args.m_mu = mu; args.m_a = a; args.m_b = b;
        // shifted copy: m_bsh[i] = b_{i-1}, so that (b_{i-1}, b_i) is one aligned 16-byte load
        double* bsh = nullptr;
        if (!dalloc(&bsh, ld + 8)) return false;
        bshpad.assign(ld + 8, -0.0);
        std::copy(model.b.begin(), model.b.begin() + (dim > 0 ? dim - 1 : 0), bshpad.begin() + 1);
        HIP_TRY(hipMemcpyAsync(bsh, bshpad.data(), (ld + 8) * 8, hipMemcpyHostToDevice, stream));
        args.m_bsh = bsh;
    } else {
        // Host callbacks with small batches are latency-bound (five copies + a synchronisation per leapfrog): there the
        // staging buffers ARE the pinned host buffers (coherent, GPU-visible on ROCm) and the kernel reads / writes them
        // over PCIe directly.  Larger batches keep device staging + bulk copies.
        // (up to 4 MB of positions per step; 8 MB where resident launches apply — dim <= 1024, <= 1024 chains — which need them)
        const uint64_t zc_limit = (n <= 1024 && dim <= 1024) ? (8u << 20) : (4u << 20);
        zero_copy = (model.kind == 1) && !(launch.staging_q && launch.staging_grad && launch.staging_logp) && n * dim * 8 <= zc_limit;
        if (launch.staging_q && launch.staging_grad && launch.staging_logp) {
            args.qeval = (double*)launch.staging_q; args.geval = (double*)launch.staging_grad; args.ueval = (double*)launch.staging_logp;
        } else if (!zero_copy && (!dalloc(&args.qeval, n * dim) || !dalloc(&args.geval, n * dim) || !dalloc(&args.ueval, n))) return false;
        if (dens) {
            args.dens_data = model.jit_data;
            args.dens_lds_doubles = (int32_t)(model.jit_lds_bytes / 8);
            args.dens_shared_doubles = (int32_t)(model.jit_shared_bytes / 8);
            args.reg_nv = model.jit_nv;
            if ((int64_t)model.jit_nv * 128 * W != args.ld) { set_error("the density's library was compiled for another dimension (nv chunks)"); return false; }
        }
        if (model.kind == 1) {
            if (!zero_copy && !dalloc(&args.ecode, n)) return false;
            if (!palloc(&h_q, n * dim) || !palloc(&h_g, n * dim) || !palloc(&h_u, n) || !palloc(&h_code, n)) return false;
            if (zero_copy) { args.qeval = h_q; args.geval = h_g; args.ueval = h_u; args.ecode = h_code; }
            // n_threads > 0: that many evaluation threads.  n_threads == 0: threads per batch from the measured cost of a row (eval_ranges)
            host_cores = usable_cores();
            // pipelining: two groups of chains in flight — while the host evaluates the rows of one group the
            // kernel of the other runs.  Zero-copy batches only (they are the latency-bound ones); launch.host_groups = 1 turns it off.
            if (zero_copy && !launch.manual && launch.host_groups != 1 && n >= 2) {
                // measured (profiles/r2_config4_host_callback_pipelining.txt): eight schools, 256 chains: 5.7 -> 6.5 (2 groups) -> 6.7 M
                // leapfrogs/s (4 groups); 1024 chains: 12.3 -> 14.5 -> 16.2.  The rest of a step is fixed latency (launch, the
                // memory-resident kernel's dependent passes, its PCIe reads of the gradients, the flag), which groups overlap with
                // each other but cannot shorten — the resident launches below can (profiles/r2_config4_resident_launches.txt: 11.3
                // and 20 M leapfrogs/s).
                n_groups = (n >= 128) ? 4 : ((n >= 32) ? 2 : 1);
                if (launch.host_groups >= 2 && launch.host_groups <= kMaxGroups) n_groups = (int)std::min<uint64_t>(launch.host_groups, n);
                for (int g = 0; g <= n_groups; ++g) grp_lo[g] = n * (uint64_t)g / (uint64_t)n_groups;
                if (!dalloc(&args.grp_arrive, kMaxGroups)) return false;
                unsigned long long* f = nullptr;
                if (!palloc(&f, 4 * kMaxGroups)) return false;
                h_grp_flag = f;
                args.grp_flag = f;
                // resident launches: one wave per chain with the state in registers (dim <= 1024); every chain of a group must be on
                // the device at once (<= 1024 chains: a quarter of the wave slots); the adaptation hook edits control blocks
                // between launches and the divergence record needs the pre-step state in memory — both keep a launch per evaluation
                unsigned long long* go = nullptr;
                if (!palloc(&go, 8 * kMaxGroups) || !dalloc(&args.grp_go_dev, 16 * kMaxGroups)) return false;
                h_grp_go = go;
                args.grp_go = go;
                remote_nv = (int)(args.ld / 128 / W);
                // (above 4 MB of positions per step the job is bound by PCIe traffic either way, and launches per evaluation were 15 %
                //  faster at 1024 chains x 1000 dimensions: resident only when asked for)
                remote = (W == 1 || W == 2 || W == 4) && remote_nv <= 8 && (int64_t)remote_nv * W * 128 == args.ld && n * (uint64_t)W <= 1024 &&
                         launch.host_persist != 1 && !launch.no_register_kernel && !lrm &&
                         !set.store_divergences && set.pause_draws.empty() && (n * dim * 8 <= (4u << 20) || launch.host_persist > 1);
                persist_evals = launch.host_persist > 1 ? launch.host_persist : 256;
                if (launch.host_persist < 0) { fall_back_after = -(int64_t)launch.host_persist; persist_evals = 7; }
                for (int g = 0; g < kMaxGroups; ++g) grp_seq[g] = remote ? 1u : 0u;
                if (remote && !getenv("NPHIP_NO_BAR")) {
                    int large_bar = 0;
                    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) large_bar = 0;
                    void *pg = nullptr, *pu = nullptr, *pc = nullptr, *po = nullptr;
                    if (large_bar && hipExtMallocWithFlags(&pg, std::max<size_t>(8, n * dim * 8), hipDeviceMallocFinegrained) == hipSuccess &&
                        hipExtMallocWithFlags(&pu, n * 8, hipDeviceMallocFinegrained) == hipSuccess &&
                        hipExtMallocWithFlags(&pc, n * 8, hipDeviceMallocFinegrained) == hipSuccess &&
                        hipExtMallocWithFlags(&po, 8 * kMaxGroups * 8, hipDeviceMallocFinegrained) == hipSuccess &&
                        cpu_writable(pg, std::max<size_t>(8, n * dim * 8)) && cpu_writable(pu, n * 8) && cpu_writable(pc, n * 8) &&
                        cpu_writable(po, 8 * kMaxGroups * 8)) {
                        bar = true;
                        bar_used = true;
                        b_g = (double*)pg; b_u = (double*)pu; b_code = (int64_t*)pc; b_go = (volatile unsigned long long*)po;
                        for (int i = 0; i < 8 * kMaxGroups; ++i) b_go[i] = 0ull;
                        args.geval = b_g; args.ueval = b_u; args.ecode = b_code; args.grp_go = b_go;
                    } else {
                        (void)hipGetLastError();
                    }
                    for (void* q : {pg, pu, pc, po}) if (q) allocs.push_back(q);
                }
                if (remote) {
                    // (a step of a resident group is ~20 us of device latency — PCIe round trips and one L2 write-back — and ~3 us of
                    //  host work per 64 rows: eight groups keep the driver thread busy while seven of them are in flight)
                    if (launch.host_groups == 0) n_groups = (int)std::max<uint64_t>(1, std::min<uint64_t>(8, n / 8));
                    // one wave per chain: a workgroup holds four chains, and they rendezvous as one (kernels.hip: remote_sync) — group
                    // bounds on multiples of 4
                    const uint64_t per_wg = (W == 1) ? 4 : 1, wgs = (n + per_wg - 1) / per_wg;
                    n_groups = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_groups, wgs));
                    for (int g = 1; g < n_groups; ++g) grp_lo[g] = (wgs * (uint64_t)g / (uint64_t)n_groups) * per_wg;
                    grp_lo[n_groups] = n;
                }
                for (int g = 0; g < n_groups; ++g) HIP_TRY(hipStreamCreateWithFlags(&grp_stream[g], hipStreamNonBlocking));
            }
        }
    }
    if (model.dense) {
        // Pp [DP][KP]: rows padded to a multiple of 16 elements and 64 rows with zeros (dense_tile.h reads whole fragments), mu [KP]
        const size_t KP = (dim + 15) / 16 * 16, DP = (dim + 63) / 64 * 64;
        double *dP = nullptr, *dmu = nullptr;
        if (!dalloc(&dP, DP * KP) || !dalloc(&dmu, std::max<size_t>(KP, (size_t)args.ld))) return false;   // (mu: zeros up to the padded row length)
        HIP_TRY(hipMemcpy2DAsync(dP, KP * 8, model.prec->data(), dim * 8, dim * 8, dim, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(dmu, model.mu.data(), dim * 8, hipMemcpyHostToDevice, stream));
        dense_dev.Pp = dP; dense_dev.mu = dmu; dense_dev.KP = (int64_t)KP; dense_dev.W = W;
        model.dev_fn = &nphip_sampler::dense_dev_fn;
        model.user = &dense_dev;
        // the resident form: every chain on the device at once (one workgroup of four chains per CU), state in registers
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 0;
        // ... when the job is large enough for it: a cluster is formed from workgroups of ONE die (8 of them, workgroups dealt round-robin), and
        // its members share the column tiles of the GEMM — with fewer than half as many workgroups per die as there are tiles (64 columns each)
        // a member computes three or more tiles per round and a launch per evaluation is faster (measured, scratch/r6_dense_small.py:
        // D = 1000: 128 chains 0.69 against 1.05 M leapfrogs/s, 256 chains 2.38 against 2.00; D = 256: 16 chains 0.28 against 0.52, 64 chains
        // 1.59 against 1.39).  launch.host_persist > 1 forces the resident form.
        const uint64_t wgs_per_die = std::max<uint64_t>(1, ((n + 3) / 4) / 8), n_tiles = (dim + 63) / 64;
        const uint64_t tiles_per_member = (n_tiles + std::min<uint64_t>(16, wgs_per_die) - 1) / std::min<uint64_t>(16, wgs_per_die);
        dg = W == 1 && dim <= 1024 && (n + 3) / 4 <= (uint64_t)cus && !lrm && !set.store_divergences &&
             set.pause_draws.empty() && !launch.no_register_kernel && launch.host_persist != 1 && launch.host_groups < 2 &&
             (tiles_per_member <= 2 || launch.host_persist > 1);
        if (dg) {
            args.dg_P = dP; args.dg_mu = dmu; args.dg_KP = (int64_t)KP;
            { const char* e = getenv("NPHIP_DG_VARIANT"); args.dg_variant = e ? atoi(e) : 0; }
            args.reg_nv = (int32_t)(args.ld / 128);
            unsigned long long* f = nullptr;
            if (!dalloc(&args.dg_sync, (size_t)kDgSyncWords) || !dalloc(&args.grp_go_dev, 16 * kMaxGroups) || !palloc(&f, 4 * kMaxGroups)) return false;
            args.grp_flag = f;
            h_grp_flag = f;
            unsigned long long* ab = nullptr;
            if (!palloc(&ab, 2)) return false;
            h_dg_abort = ab;
        }
    }
    if (model.kind == 2 && launch.host_groups >= 2 && !launch.manual && n >= 8) {
        // Device callbacks in groups (round 5): the chains in `host_groups` contiguous groups, each with a stream of its
        // own on which its k_advance launch and its callback alternate — while one group's callback runs (a GEMM, a torch function) the
        // other group's engine kernel does.  No flag and no host synchronisation: stream order is the only dependency, the callback
        // gets the group's rows of the staging buffers (pointers offset, n_chains = the group's).  Bounds on whole workgroups.
        const uint64_t per_wg = (W == 1) ? 4 : 1, wgs = (n + per_wg - 1) / per_wg;
        n_groups = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)launch.host_groups, (uint64_t)kMaxGroups, wgs}));
        grp_lo[0] = 0;
        for (int g = 1; g < n_groups; ++g) grp_lo[g] = (wgs * (uint64_t)g / (uint64_t)n_groups) * per_wg;
        grp_lo[n_groups] = n;
        for (int g = 0; g < n_groups; ++g) HIP_TRY(hipStreamCreateWithFlags(&grp_stream[g], hipStreamNonBlocking));
        cb_groups = n_groups;
    }
    if (model.init_kind == 2) {
        if (model.n_init_points < launch.chain_offset + n) { set_error("explicit init points do not cover all chains"); return false; }
        double* ip = nullptr;
        if (!dalloc(&ip, n * dim)) return false;
        HIP_TRY(hipMemcpyAsync(ip, model.init_points.data() + launch.chain_offset * dim, n * dim * 8, hipMemcpyHostToDevice, stream));
        args.init_points = ip;
    }
    const size_t nt = (size_t)n * T;
    if (launch.store_draws && !dalloc(&args.tr_draws, nt * dim)) return false;
    if (set.store_gradient && !dalloc(&args.tr_grad, nt * dim)) return false;
    if (set.store_mass_matrix && !dalloc(&args.tr_mm, nt * dim)) return false;
    if (set.store_divergences)
        for (int k = 0; k < 4; ++k)
            if (!dalloc(&args.tr_div[k], nt * dim, 0xFF)) return false;  // all-ones = NaN
    if (!dalloc(&args.st_depth, nt) || !dalloc(&args.st_nsteps, nt) || !dalloc(&args.st_idx, nt)) return false;
    if (!dalloc(&args.st_diverging, nt) || !dalloc(&args.st_maxdepth, nt) || !dalloc(&args.st_tuning, nt)) return false;
    if (!dalloc(&args.st_energy, nt) || !dalloc(&args.st_energy_error, nt) || !dalloc(&args.st_logp, nt)) return false;
    if (!dalloc(&args.st_step, nt) || !dalloc(&args.st_step_bar, nt)) return false;
    if (!dalloc(&args.st_accept, nt) || !dalloc(&args.st_accept_sym, nt)) return false;
    if (!palloc(&h_counters, 4)) return false;   // two slots of (chains done, chains in error)
    if (!dalloc(&d_args, 1)) return false;
    HIP_TRY(hipMemcpyAsync(d_args, &args, sizeof(Args), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return true;
}

std::string nphip_sampler::chain_error_message() {
    std::vector<Ctl> h(n);
    if (hipMemcpy(h.data(), args.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost) != hipSuccess) return "chain error";
    for (uint64_t i = 0; i < n; ++i) {
        if (h[i].phase == PH_ERROR) {
            std::string c = "chain " + std::to_string(launch.chain_offset + i) + ": ";
            if (h[i].err == CE_INIT_FAILED) return c + "could not find a finite initial point (logp or gradient not finite)";
            if (h[i].err == CE_FATAL_LOGP) return c + "logp function returned a fatal error";
            if (h[i].err == CE_RESUME_FAILED) return c + "the position given to resume_at does not evaluate (logp or gradient not finite)";
            return c + "unknown error";
        }
    }
    return "chain error";
}

bool nphip_sampler::launch_kernel(bool fused_, int have) {
    // (a runtime-compiled density costs microseconds per evaluation: 512 per launch is milliseconds of kernel already, and the
    //  job's tail — chains that are done wait for the launch to end — shrinks with the launch)
    // (the dense Gaussian's resident form: an evaluation round of the whole launch is ~50 us at 1024 chains x 1000 dimensions)
    args.max_evals = (fused_ || dens || dg) ? (launch.evals_per_launch > 0 ? launch.evals_per_launch : (dg ? 256 : (dens ? 512 : default_evals_per_launch(dim)))) : 0;
    args.have_result = have;
    if (kernel_ms_acc) {
        while (tev.size() < 2 * (timed_launches + 1)) {
            hipEvent_t e = nullptr;
            if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) return false;
            tev.push_back(e);
        }
        if (!hip_ok(hipEventRecord(tev[2 * timed_launches], stream), "hipEventRecord")) return false;
    }
    if (dens) {
        LaunchSlice sl;
        memset(&sl, 0, sizeof(sl));
        sl.chain_lo = 0; sl.chain_n = (int)n; sl.grp = -1;
        const uint64_t cpb = W == 1 ? 4 : 1;   // chains per workgroup
        const int rc = model.jit_launch(d_args, args.max_evals, (void*)stream, &sl,
                                        cpb * model.jit_lds_bytes + model.jit_shared_bytes + cpb * 2 * (uint64_t)args.ld * 8);
        if (rc != 0) { set_error(std::string("launch of the runtime-compiled density kernel: ") + hipGetErrorString((hipError_t)rc)); return false; }
    } else if (dg) {
        LaunchSlice sl;
        memset(&sl, 0, sizeof(sl));
        sl.chain_lo = 0; sl.chain_n = (int)n; sl.grp = -1; sl.seq = ++dg_launch_id; sl.n_grp = 1;
        for (int g = 1; g <= kMaxGroups; ++g) sl.grp_lo[g] = (int)n;
        if (!hip_ok(hipMemsetAsync(args.dg_sync, 0, (size_t)kDgSyncWords * 8, stream), "hipMemsetAsync")) return false;
        if (!hip_ok(launch_dense_resident(d_args, args.reg_nv, args.max_evals, stream, sl), "launch k_advance (dense, resident)")) return false;
    } else if (materialise && model.dense) {
        // the launches before this one were resident ones (they defer the first half of a tree leapfrog into the leaf): this one performs it
        LaunchSlice sl;
        memset(&sl, 0, sizeof(sl));
        sl.chain_lo = 0; sl.chain_n = (int)n; sl.grp = -1; sl.materialise = 1;
        materialise = false;
        if (!hip_ok(launch_advance(args, d_args, fused_, W, stream, &sl), "launch k_advance")) return false;
    } else if (!hip_ok(launch_advance(args, d_args, fused_, W, stream), "launch k_advance")) return false;
    if (kernel_ms_acc) {
        if (!hip_ok(hipEventRecord(tev[2 * timed_launches + 1], stream), "hipEventRecord")) return false;
        timed_launches += 1;
    }
    launches.fetch_add(1);
    return true;
}

bool nphip_sampler::check_counters(int slot, bool& all_done) {
    const unsigned long long* hc = h_counters + 2 * slot;
    if (hc[1] > 0) { (void)hipStreamSynchronize(stream); set_error(chain_error_message()); return false; }
    all_done = hc[0] >= n;
    return true;
}

bool nphip_sampler::iteration_fused(bool& all_done) {
    const int slot = (int)(fused_k & 1);
    if (!ev_f[0] && (!hip_ok(hipEventCreateWithFlags(&ev_f[0], hipEventDisableTiming), "hipEventCreate") ||
                     !hip_ok(hipEventCreateWithFlags(&ev_f[1], hipEventDisableTiming), "hipEventCreate"))) return false;
    if (!launch_kernel(!dens && !dg, 0)) return false;
    if (!hip_ok(hipMemcpyAsync(h_counters + 2 * slot, args.counters, 16, hipMemcpyDeviceToHost, stream), "copy counters")) return false;
    if (dg && !hip_ok(hipMemcpyAsync((void*)(h_dg_abort + slot), args.dg_sync + (size_t)128 * 32, 8, hipMemcpyDeviceToHost, stream), "copy abort word")) return false;
    if (!hip_ok(hipEventRecord(ev_f[slot], stream), "hipEventRecord")) return false;
    fused_k += 1;
    if (fused_k < 2) return true;
    // the launch before this one: by now it has the device to itself no longer — its successor is queued behind it
    if (!hip_ok(hipEventSynchronize(ev_f[slot ^ 1]), "hipEventSynchronize")) return false;
    if (dg && !dg_check(slot ^ 1)) return false;
    return check_counters(slot ^ 1, all_done);
}

// The dense Gaussian's resident launches, looked at one launch late (as the counters are): a rendezvous that timed out is an error (it
// cannot happen after a roll call that succeeded, short of a device fault); a roll call that failed is a launch that did nothing.
bool nphip_sampler::dg_check(int slot) {
    if (h_dg_abort[slot] != 0ull) {
        (void)hipStreamSynchronize(stream);
        set_error("dense Gaussian, resident kernel: a rendezvous of the launch-wide gradient GEMM timed out");
        return false;
    }
    if (getenv("NPHIP_DEBUG") && dg_launch_id <= 2) {
        unsigned seats[8];
        (void)hipMemcpy(seats, args.dg_sync + 128 * 32 + 16, sizeof(seats), hipMemcpyDeviceToHost);
        fprintf(stderr, "nphip: dense resident launch: workgroups per die %u %u %u %u %u %u %u %u\n", seats[0], seats[1], seats[2], seats[3], seats[4], seats[5], seats[6], seats[7]);
    }
    const unsigned long long failed_id = h_grp_flag[3];
    if (failed_id != 0ull && failed_id != dg_fail_seen) {
        dg_fail_seen = failed_id;
        if (++dg_fail >= 3) {
            // the device does not hold all chains at once (something else is running on it): a launch per evaluation from here on
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
            dg = false;
            dg_fell_back = true;
            materialise = true;
            manual_have = 0;
            fused_k = 0;
        }
    } else if (failed_id == dg_fail_seen) {
        dg_fail = 0;
    }
    return true;
}

// Host callback on the rows [lo, lo + cnt) of the staging buffers (the reference calls the same function pointer once per
// chain-step from its worker threads: src/pymc.rs:197-215).
void nphip_sampler::eval_rows(uint64_t lo, uint64_t cnt) {
    const RowRange one{lo, cnt};
    eval_ranges(&one, 1);
}

// The rows of several ranges as ONE batch (resident launches: every group that has published by now).  Threads: the model's
// from the measured cost of a row — batches evaluated on this thread are timed — so that every thread gets >= 10 us of rows, at
// most the model's n_threads when given (a batch of 256 rows of 40 ns is faster on one thread than handed out; 32 rows of 1.6 us are not).
void nphip_sampler::eval_ranges(const RowRange* rg, int nr) {
    const uint64_t d = dim;
    uint64_t total = 0;
    for (int k = 0; k < nr; ++k) total += rg[k].cnt;
    const std::function<void(uint64_t)> f = [this, d, rg, nr](uint64_t r) {
        int k = 0;
        while (k + 1 < nr && r >= rg[k].cnt) { r -= rg[k].cnt; ++k; }
        const uint64_t row = rg[k].lo + r;
        double lp = NAN;
        h_code[row] = (int64_t)model.host_fn(d, h_q + row * d, h_g + row * d, &lp, model.user);   // c_int, sign-extended
        h_u[row] = lp;
    };
    // (the model's n_threads is an upper bound: eight schools with n_threads = 16 forced onto 16 threads ran at 6.7 M leapfrogs/s
    //  against 11.7 on one)
    const int cap = model.n_threads > 0 ? model.n_threads : std::max(1, host_cores - 1);
    int want = 1;
    if (timed_batches >= 4 && row_ns >= 150.0)   // (rows cheaper than that are faster on one thread: measured with 35 ns rows)
        want = (int)std::max(1.0, std::min<double>((double)cap, std::floor((double)total * row_ns / 10000.0)));
    if (want >= 2) {
        // sized to the largest batch width asked for so far (every worker of a pool acknowledges every batch: idle workers of a
        // pool sized to the core count would spin through each of the steady batches of a resident job); growing it is rare
        if (!pool || pool->size() < want) pool.reset(new RowPool(want));
        eval_threads = std::max(eval_threads, want);
        pool->run(total, f, want);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t r = 0; r < total; ++r) f(r);
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (double)std::max<uint64_t>(1, total);
    row_ns = (timed_batches == 0) ? ns : std::min(row_ns, ns);
    timed_batches += 1;
}

// Device callbacks in groups: one step of every group.  The first step runs on the main stream (it follows the set-up work there).
bool nphip_sampler::iteration_callback_groups(bool& all_done, int& have) {
    if ((cb_polls & 7) == 0) {
        const int slot = (int)((cb_polls >> 3) & 1);
        if (!cb_ev[slot] && !hip_ok(hipEventCreate(&cb_ev[slot]), "hipEventCreate")) return false;
        if (cb_polls >= 16 && !hip_ok(hipEventSynchronize(cb_ev[slot]), "hipEventSynchronize")) return false;
        if (!hip_ok(hipMemcpyAsync(h_counters, args.counters, 16, hipMemcpyDeviceToHost, grp_stream[n_groups - 1]), "copy counters")) return false;
        if (!hip_ok(hipEventRecord(cb_ev[slot], grp_stream[n_groups - 1]), "hipEventRecord")) return false;
    }
    ++cb_polls;
    volatile unsigned long long* hc = h_counters;
    if (hc[1] > 0) {
        for (int g = 0; g < n_groups; ++g) (void)hipStreamSynchronize(grp_stream[g]);
        set_error(chain_error_message());
        return false;
    }
    if (hc[0] >= n) {
        all_done = true;
        for (int g = 0; g < n_groups; ++g)
            if (!hip_ok(hipStreamSynchronize(grp_stream[g]), "hipStreamSynchronize")) return false;
        return true;
    }
    for (int g = 0; g < n_groups; ++g) {
        LaunchSlice sl;
        memset(&sl, 0, sizeof(sl));
        sl.chain_lo = (int)grp_lo[g];
        sl.chain_n = (int)(grp_lo[g + 1] - grp_lo[g]);
        sl.grp = -1;
        sl.materialise = materialise ? 1 : 0;
        args.max_evals = 0;
        args.have_result = have;
        if (!hip_ok(launch_advance(args, d_args, false, W, grp_stream[g], &sl), "launch k_advance")) return false;
        launches.fetch_add(g == 0 ? 1 : 0);    // (a step of all groups counts once)
        const uint64_t lo = grp_lo[g];
        const int rc = model.dev_fn((uint64_t)sl.chain_n, dim, args.qeval + lo * dim, args.geval + lo * dim, args.ueval + lo, (void*)grp_stream[g], model.user);
        if (rc < 0) {
            for (int o = 0; o < n_groups; ++o) (void)hipStreamSynchronize(grp_stream[o]);
            set_error("device logp callback failed (code " + std::to_string(rc) + ")");
            return false;
        }
    }
    materialise = false;
    have = 1;
    return true;
}

bool nphip_sampler::iteration_callback(bool& all_done, int& have) {
    if (model.kind == 2 && cb_graph_steps > 0 && have == 1 && !kernel_ms_acc) {
        if (cb_groups > 1 && !cb_graph) {   // the steps before the capture ran on the groups' streams: the graph (launched on the main one) follows them
            for (int g = 0; g < n_groups; ++g)
                if (!hip_ok(hipStreamSynchronize(grp_stream[g]), "hipStreamSynchronize")) return false;
        }
        return iteration_graph(all_done);
    }
    if (model.kind == 2 && cb_groups > 1 && !kernel_ms_acc) {
        if (!grp_primed) {   // everything enqueued on the main stream so far (set-up, the chains' start) precedes the groups' first launches
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
            grp_primed = true;
        }
        return iteration_callback_groups(all_done, have);
    }
    if (!launch_kernel(false, have)) return false;
    if (model.kind == 1) {
        // host callback: D2H positions, evaluate rows on the host pool, H2D gradients
        // (pinned hipMemcpyAsync; the reference calls the same function pointer once per
        // chain-step from its worker threads: src/pymc.rs:197-215)
        if (!zero_copy && !hip_ok(hipMemcpyAsync(h_q, args.qeval, n * dim * 8, hipMemcpyDeviceToHost, stream), "D2H q")) return false;
        if (!hip_ok(hipMemcpyAsync(h_counters, args.counters, 16, hipMemcpyDeviceToHost, stream), "copy counters")) return false;
        if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
        if (h_counters[1] > 0) { set_error(chain_error_message()); return false; }
        if (h_counters[0] >= n) { all_done = true; return true; }
        const uint64_t d = dim;
        (void)d;
        eval_rows(0, n);
        if (!zero_copy) {
            if (!hip_ok(hipMemcpyAsync(args.geval, h_g, n * dim * 8, hipMemcpyHostToDevice, stream), "H2D grad")) return false;
            if (!hip_ok(hipMemcpyAsync(args.ueval, h_u, n * 8, hipMemcpyHostToDevice, stream), "H2D logp")) return false;
            if (!hip_ok(hipMemcpyAsync(args.ecode, h_code, n * 8, hipMemcpyHostToDevice, stream), "H2D code")) return false;
        }
    } else {
        // device callback: counters are polled without synchronising (stale values only delay exit), and only every
        // eighth step: the 16-byte copy is a 5 us kernel in the same stream (13 % of a step with a one-kernel model)
        // ... and the host stays at most sixteen steps ahead of the device: without a bound it enqueues thousands of launches while
        // the device works through the first ones, and when the last chain finishes the device still has all of them to run
        // (measured: 11 089 launches for a job of 3 425 evaluation steps)
        if ((cb_polls & 7) == 0) {
            const int slot = (int)((cb_polls >> 3) & 1);
            if (!cb_ev[slot] && !hip_ok(hipEventCreate(&cb_ev[slot]), "hipEventCreate")) return false;
            if (cb_polls >= 16 && !hip_ok(hipEventSynchronize(cb_ev[slot]), "hipEventSynchronize")) return false;
            if (!hip_ok(hipMemcpyAsync(h_counters, args.counters, 16, hipMemcpyDeviceToHost, stream), "copy counters")) return false;
            if (!hip_ok(hipEventRecord(cb_ev[slot], stream), "hipEventRecord")) return false;
        }
        ++cb_polls;
        volatile unsigned long long* hc = h_counters;
        if (hc[1] > 0) {
            (void)hipStreamSynchronize(stream);
            set_error(chain_error_message());
            return false;
        }
        if (hc[0] >= n) { all_done = true; return hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
        int rc = model.dev_fn(n, dim, args.qeval, args.geval, args.ueval, (void*)stream, model.user);
        if (rc < 0) {
            (void)hipStreamSynchronize(stream);
            set_error("device logp callback failed (code " + std::to_string(rc) + ")");
            return false;
        }
    }
    have = 1;
    return true;
}

// Device-callback models, steady state: `cb_graph_steps` x (engine kernel, model callback) captured once in a HIP graph
// and replayed — one host call per run of leapfrogs instead of two launches per leapfrog.  At most two replays are in
// flight (the host must not run ahead of the done / error counters by more than that).
bool nphip_sampler::iteration_graph(bool& all_done) {
    if (!cb_graph) {
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok && cb_groups > 1) {
            // groups of chains as parallel branches of ONE graph (round 5): the capture forks from the main stream into every group's stream,
            // each branch is that group's `cb_graph_steps` x (k_advance slice, callback on the group's rows), and joins again — the device
            // overlaps one group's callback with another's kernel, and the host launches one graph per `cb_graph_steps` steps
            args.max_evals = 0;
            args.have_result = 1;
            hipEvent_t fork = nullptr;
            ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess && hipEventRecord(fork, stream) == hipSuccess;
            std::vector<hipEvent_t> joins((size_t)n_groups, nullptr);
            for (int gi = 0; ok && gi < n_groups; ++gi) {
                ok = hipStreamWaitEvent(grp_stream[gi], fork, 0) == hipSuccess;
                LaunchSlice sl;
                memset(&sl, 0, sizeof(sl));
                sl.chain_lo = (int)grp_lo[gi];
                sl.chain_n = (int)(grp_lo[gi + 1] - grp_lo[gi]);
                sl.grp = -1;
                const uint64_t lo = grp_lo[gi];
                for (int i = 0; ok && i < cb_graph_steps; ++i)
                    ok = launch_advance(args, d_args, false, W, grp_stream[gi], &sl) == hipSuccess &&
                         model.dev_fn((uint64_t)sl.chain_n, dim, args.qeval + lo * dim, args.geval + lo * dim, args.ueval + lo, (void*)grp_stream[gi], model.user) >= 0;
                ok = ok && hipEventCreateWithFlags(&joins[(size_t)gi], hipEventDisableTiming) == hipSuccess &&
                     hipEventRecord(joins[(size_t)gi], grp_stream[gi]) == hipSuccess && hipStreamWaitEvent(stream, joins[(size_t)gi], 0) == hipSuccess;
            }
            ok = (hipStreamEndCapture(stream, &g) == hipSuccess) && ok && g != nullptr;
            if (fork) (void)hipEventDestroy(fork);
            for (hipEvent_t e : joins) if (e) (void)hipEventDestroy(e);
        } else if (ok) {
            args.max_evals = 0;
            args.have_result = 1;
            for (int i = 0; ok && i < cb_graph_steps; ++i) {
                ok = launch_advance(args, d_args, false, W, stream) == hipSuccess &&
                     model.dev_fn(n, dim, args.qeval, args.geval, args.ueval, (void*)stream, model.user) >= 0;
            }
            ok = (hipStreamEndCapture(stream, &g) == hipSuccess) && ok && g != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&cb_graph, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        if (!ok) {  // not capturable (e.g. a callback that allocates): keep stepping the plain way
            if (getenv("NPHIP_DEBUG")) fprintf(stderr, "nphip: graph_steps: the (kernel, callback) sequence could not be captured (%s): stepping without a graph\n", hipGetErrorString(hipGetLastError()));
            (void)hipGetLastError();
            cb_graph = nullptr;
            cb_graph_steps = 0;
            int have = 1;
            return iteration_callback(all_done, have);
        }
        for (auto& e : cb_ev)
            if (!e && !hip_ok(hipEventCreate(&e), "hipEventCreate")) return false;
    }
    const int slot = (int)(cb_replays & 1);
    if (cb_replays >= 2 && !hip_ok(hipEventSynchronize(cb_ev[slot]), "hipEventSynchronize")) return false;
    if (!hip_ok(hipGraphLaunch(cb_graph, stream), "hipGraphLaunch")) return false;
    if (!hip_ok(hipMemcpyAsync(h_counters, args.counters, 16, hipMemcpyDeviceToHost, stream), "copy counters")) return false;
    if (!hip_ok(hipEventRecord(cb_ev[slot], stream), "hipEventRecord")) return false;
    cb_replays += 1;
    launches.fetch_add((uint64_t)cb_graph_steps);
    volatile unsigned long long* hc = h_counters;
    if (hc[1] > 0) {
        (void)hipStreamSynchronize(stream);
        set_error(chain_error_message());
        return false;
    }
    if (hc[0] >= n) { all_done = true; return hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
    return true;
}

bool nphip_sampler::launch_group(int g, int have) {
    LaunchSlice sl;
    sl.chain_lo = (int)grp_lo[g];
    sl.chain_n = (int)(grp_lo[g + 1] - grp_lo[g]);
    sl.grp = g;
    sl.seq = ++grp_seq[g];
    sl.materialise = materialise ? 1 : 0;
    args.max_evals = 0;
    args.have_result = have;
    if (!hip_ok(launch_advance(args, d_args, false, W, grp_stream[g], &sl), "launch k_advance")) return false;
    launches.fetch_add(1);
    return true;
}

bool nphip_sampler::wait_group(int g) {
    // the kernel's last arriver writes the sequence number of the launch into pinned host memory (kernels.hip: k_advance)
    const unsigned long long want = grp_seq[g];
    volatile unsigned long long* f = h_grp_flag + 4 * g;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0; f[0] != want; ++spin) {
        if ((spin & 0x3ff) == 0x3ff) {
            if (hipStreamQuery(grp_stream[g]) == hipSuccess && f[0] != want) {
                // the stream is idle and the flag never came: the launch failed
                if (f[0] != want) { set_error("host-callback group finished without publishing its completion flag"); return false; }
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { set_error("timeout waiting for the engine kernel"); return false; }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

// ---- resident launches ------------------------------------------------------------------------------------------------
// ONE launch covers all groups: HIP multiplexes streams onto a few hardware queues, and a group's kernel queued behind another
// group's resident kernel would never start while the host waits for it.
bool nphip_sampler::launch_remote_all() {
    LaunchSlice sl;
    sl.chain_lo = 0;
    sl.chain_n = (int)n;
    sl.grp = 0;
    sl.seq = ++remote_launch_id;
    sl.materialise = 0;
    sl.n_grp = n_groups;
    if (bar && remote_evals >= 64) {
        // The BAR copy of the results takes PCIe reads off the device's critical path and costs the driver thread the copy.  It
        // pays while the device is what the job waits for; once a round of host work (every group's rows) is longer than a
        // device step (~20 us) the host is the bound and the copy only adds to it: back to results in host memory (launch boundary:
        // nothing is in flight).  Measured: eight schools, 256 chains 10.5 -> 12.2 M leapfrogs/s with the copy, 1024 chains 19.5 -> 16.3.
        const double host_round_us = (double)n_groups * (t_eval_ns / (double)remote_evals + 800.0) * 1e-3;
        if (host_round_us > 25.0) {
            bar = false;
            args.geval = h_g; args.ueval = h_u; args.ecode = h_code; args.grp_go = h_grp_go;
            for (int g = 0; g < n_groups; ++g) h_grp_go[8 * g] = 0ull;
            if (!hip_ok(hipMemcpyAsync(d_args, &args, sizeof(Args), hipMemcpyHostToDevice, grp_stream[0]), "update args")) return false;
        }
    }
    for (int g = 0; g <= kMaxGroups; ++g) sl.grp_lo[g] = (int)grp_lo[std::min(g, n_groups)];
    for (int g = 0; g < kMaxGroups; ++g) sl.grp_seq[g] = grp_seq[g];   // a group's first evaluation carries the number the host waits for
    if (!hip_ok(launch_remote(d_args, W, remote_nv, grp_stream[0], sl), "launch k_advance (resident)")) return false;
    for (int g = 0; g < n_groups; ++g) { grp_running[g] = true; grp_evals[g] = 0; }
    remote_fresh = true;
    return true;
}

// 1: group g has published the evaluation the host waits for; 0: not yet
inline int nphip_sampler::poll_remote(int g) {
    if ((h_grp_flag[4 * g] & kGoSeqMask) != (unsigned long long)grp_seq[g]) return 0;
    std::atomic_thread_fence(std::memory_order_acquire);
    return 1;
}

// waits until some running group (or, with only >= 0, that group) has published; returns its index, -1 on error, -2 when the
// launch's roll call failed (nothing ran)
int nphip_sampler::wait_remote(int only) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 1;; ++spin) {
        for (int g = 0; g < n_groups; ++g) {
            const int o = (remote_next + g) % n_groups;
            if ((only < 0 || o == only) && grp_running[o] && poll_remote(o)) { remote_next = (o + 1) % n_groups; remote_fresh = false; return o; }
        }
        if ((spin & 0xff) == 0) {
            if (remote_fresh && h_grp_flag[3] == (unsigned long long)remote_launch_id) return -2;
            if ((spin & 0xffff) == 0) {
                const hipError_t q = hipStreamQuery(grp_stream[0]);
                if (q != hipSuccess && q != hipErrorNotReady) { (void)hip_ok(q, "resident launch"); return -1; }
                if (q == hipSuccess) {
                    bool any = false;
                    for (int g = 0; g < n_groups; ++g) any = any || (grp_running[g] && poll_remote(g));
                    if (!any && !(remote_fresh && h_grp_flag[3] == (unsigned long long)remote_launch_id)) {
                        set_error("resident host-callback launch ended without publishing its evaluation");
                        return -1;
                    }
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) { set_error("timeout waiting for the engine kernel"); return -1; }
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

static inline void store_fence() {
#if defined(__x86_64__)
    __builtin_ia32_sfence();   // write-combining stores (the BAR) become globally visible, in order with what follows
#else
    std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}

// large BAR: the rows the callback just wrote (cached host memory) go to the device's copy
void nphip_sampler::publish_rows(uint64_t lo, uint64_t cnt) {
    if (!bar) return;
    memcpy(b_g + lo * dim, h_g + lo * dim, cnt * dim * 8);
    memcpy(b_u + lo, h_u + lo, cnt * 8);
    memcpy(b_code + lo, h_code + lo, cnt * 8);
    store_fence();
}

// results of the evaluation grp_seq[g] are in the staging rows: one word to the kernel; `last` ends the group's part of the
// launch at the next boundary
void nphip_sampler::answer_group(int g, bool last) {
    std::atomic_thread_fence(std::memory_order_release);
    (bar ? b_go : h_grp_go)[8 * g] = (unsigned long long)grp_seq[g] | (last ? kGoLast : 0ull);
    if (bar) store_fence();
    std::atomic_thread_fence(std::memory_order_seq_cst);
    grp_seq[g] += 1;
    if (last) grp_running[g] = false;
    launches.fetch_add(1);
}

// Bring the resident launch to its next boundary: one more evaluation of every group still in it, answered with kGoLast.  (A
// chain's state only exists in memory at a boundary — whoever wants to read it, pause, abort or finish goes through here.)
bool nphip_sampler::drain_groups() {
    for (int g = 0; g < n_groups; ++g) {
        if (!grp_running[g]) continue;
        const int w = wait_remote(g);
        if (w == -1) return false;
        if (w == -2) { for (int o = 0; o < n_groups; ++o) grp_running[o] = false; break; }   // nothing ran
        eval_rows(grp_lo[g], grp_lo[g + 1] - grp_lo[g]);
        publish_rows(grp_lo[g], grp_lo[g + 1] - grp_lo[g]);
        answer_group(g, true);
    }
    return true;
}

// The device could not hold all chains at once (roll call, kernels.hip): back to one launch per evaluation, for good.
bool nphip_sampler::remote_fall_back() {
    if (!drain_groups()) return false;
    remote = false;
    remote_fell_back = true;
    for (int g = 0; g < n_groups; ++g)
        if (!hip_ok(hipStreamSynchronize(grp_stream[g]), "hipStreamSynchronize")) return false;
    if (bar) {
        // launches per evaluation read the host's rows directly
        bar = false;
        args.geval = h_g; args.ueval = h_u; args.ecode = h_code; args.grp_go = h_grp_go;
        if (!hip_ok(hipMemcpy(d_args, &args, sizeof(Args), hipMemcpyHostToDevice), "update args")) return false;
    }
    materialise = true;
    grp_primed = false;
    return true;
}

// Every group that has published by now is evaluated as one batch and answered; up to one round of groups per call.  (With a
// cheap callback the groups come one at a time and the device is what the job waits for; with an expensive one all the other
// groups publish while a batch is being evaluated, and the batches grow until the evaluation pool is busy.)
bool nphip_sampler::iteration_remote(bool& all_done) {
    if (!grp_primed) {
        if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;   // set-up copies ran on the main stream
        grp_primed = true;
    }
    for (int served = 0; served < n_groups;) {
        if (fall_back_after > 0 && remote_evals >= fall_back_after) return remote_fall_back();
        bool any = false;
        for (int g = 0; g < n_groups; ++g) any = any || grp_running[g];
        if (!any && !launch_remote_all()) return false;
        const auto tw0 = std::chrono::steady_clock::now();
        const int first = wait_remote(-1);
        const auto tw1 = std::chrono::steady_clock::now();
        t_wait_ns += std::chrono::duration<double, std::nano>(tw1 - tw0).count();
        if (first == -1) return false;
        if (first == -2) { for (int o = 0; o < n_groups; ++o) grp_running[o] = false; return remote_fall_back(); }
        int ready[kMaxGroups];
        RowRange rg[kMaxGroups];
        int nr = 0;
        // (looking at the other groups' flags costs a cache miss each: only when a batch is long enough for them to have published)
        const bool gather = model.n_threads > 1 || (timed_batches >= 4 && row_ns * (double)(grp_lo[first + 1] - grp_lo[first]) > 8000.0);
        for (int k = 0; k < (gather ? n_groups : 1); ++k) {
            const int g = (first + k) % n_groups;
            if (!grp_running[g] || (g != first && !poll_remote(g))) continue;
            const unsigned long long pub = h_grp_flag[4 * g];
            if (pub & kPubError) { (void)sync_all(); set_error(chain_error_message()); return false; }
            if (pub & kPubAllDone) { all_done = true; return sync_all(); }
            ready[nr] = g;
            rg[nr] = RowRange{grp_lo[g], grp_lo[g + 1] - grp_lo[g]};
            nr += 1;
        }
        eval_ranges(rg, nr);
        for (int k = 0; k < nr; ++k) publish_rows(rg[k].lo, rg[k].cnt);
        t_eval_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tw1).count();
        for (int k = 0; k < nr; ++k) {
            const int g = ready[k];
            grp_evals[g] += 1;
            remote_evals += 1;
            answer_group(g, grp_evals[g] >= persist_evals);
        }
        served += nr;
    }
    return true;
}

// One round over the groups: for each, wait for its kernel, evaluate its rows on the host pool, relaunch it.
bool nphip_sampler::iteration_pipelined(bool& all_done) {
    if (!grp_primed) {
        if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;   // set-up copies ran on the main stream
        for (int g = 0; g < n_groups; ++g)
            if (!launch_group(g, 0)) return false;
        grp_primed = true;
        materialise = false;
    }
    const uint64_t d = dim;
    for (int g = 0; g < n_groups; ++g) {
        if (!wait_group(g)) return false;
        volatile unsigned long long* f = h_grp_flag + 4 * g;
        if (f[2] > 0) { (void)sync_all(); set_error(chain_error_message()); return false; }
        if (f[1] >= n) {
            // every chain of the job is done; drain the launches still in flight for the other groups
            for (int o = 0; o < n_groups; ++o)
                if (o != g && !wait_group(o)) return false;
            all_done = true;
            return sync_all();
        }
        const uint64_t lo = grp_lo[g], cnt = grp_lo[g + 1] - lo;
        (void)d;
        eval_rows(lo, cnt);
        std::atomic_thread_fence(std::memory_order_release);
        if (!launch_group(g, 1)) return false;
    }
    return true;
}

void nphip_sampler::run() {
    (void)hipSetDevice(device);
    t_start = std::chrono::steady_clock::now();
    int have = 0;
    bool all_done = false;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu);
            if (want_pause && !want_abort && remote) {
                // a resident launch would sit on the device for the whole pause (and hold up whatever shares its hardware queue):
                // bring it to its boundary first
                lk.unlock();
                { std::lock_guard<std::mutex> run_lk(mu_run); (void)sync_all(); }
                lk.lock();
            }
            cv.wait(lk, [&] { return !want_pause || want_abort; });
            if (want_abort) break;
        }
        bool ok;
        {
            std::lock_guard<std::mutex> run_lk(mu_run);
            ok = (fused || dens || dg) ? iteration_fused(all_done) : (remote ? iteration_remote(all_done) : ((n_groups > 0 && cb_groups == 0) ? iteration_pipelined(all_done) : iteration_callback(all_done, have)));
        }
        seconds.store(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        if (!ok) { fail(t_error); break; }
        if (all_done) break;
    }
    {
        std::lock_guard<std::mutex> run_lk(mu_run);
        (void)sync_all();
    }
    if (remote_evals > 0 && getenv("NPHIP_TIMING"))
        fprintf(stderr, "[nutpie-hip] resident launches: %lld evaluations of a group, driver thread: %.2f us waiting for the device + %.2f us evaluating per evaluation, %d evaluation threads (%.0f ns per row), results through the BAR: %s\n",
                (long long)remote_evals, t_wait_ns / remote_evals * 1e-3, t_eval_ns / remote_evals * 1e-3, eval_threads, row_ns,
                bar ? "yes" : (bar_used ? "at first" : "no"));
    {
        std::lock_guard<std::mutex> lk(mu);
        finished = all_done && !failed;
        thread_done = true;
    }
    cv.notify_all();
}

extern "C" {

nphip_sampler_t* nphip_sampler_create(const nphip_settings_t* set, const nphip_model_t* model, const nphip_launch_t* launch) {
    if (!set || !model) { set_error("null settings or model"); return nullptr; }
    auto* s = new nphip_sampler();
    s->set = *set;
    s->model = *model;
    if (launch) s->launch = *launch; else nphip_launch_defaults(&s->launch);
    s->device = s->launch.device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available: the nutpie-hip engine requires an AMD GPU (no CPU fallback)");
        delete s;
        return nullptr;
    }
    if (!s->setup()) { s->release(); delete s; return nullptr; }
    // (the dense Gaussian's evaluation is two kernel launches of the engine's own: always capturable — 16 steps per graph unless told otherwise)
    s->cb_graph_steps = (s->model.kind == 2 && s->launch.graph_steps > 0) ? s->launch.graph_steps : ((s->model.dense && s->launch.graph_steps == 0) ? 16 : 0);
    s->want_pause = s->launch.start_paused != 0;
    s->manual = s->launch.manual != 0;
    if (!s->manual) s->th = std::thread([s] { s->run(); });
    else s->t_start = std::chrono::steady_clock::now();
    return s;
}

void nphip_sampler_free(nphip_sampler_t* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->want_abort = true; }
    s->cv.notify_all();
    if (s->th.joinable()) s->th.join();
    (void)hipSetDevice(s->device);
    s->pool.reset();
    s->release();
    delete s;
}

int nphip_sampler_step(nphip_sampler_t* s, uint64_t n_launches, double* kernel_ms, uint64_t* launches_done) {
    if (!s->manual) { set_error("nphip_sampler_step needs a sampler created with launch.manual = 1"); return NPHIP_WAIT_ERROR; }
    if (launches_done) *launches_done = 0;
    if (kernel_ms) *kernel_ms = 0.0;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->failed) { set_error(s->error); return NPHIP_WAIT_ERROR; }
        if (s->thread_done) return NPHIP_WAIT_DONE;
    }
    (void)hipSetDevice(s->device);
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    s->kernel_ms_acc = kernel_ms;
    s->timed_launches = 0;
    bool all_done = false, ok = true;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n_launches && !all_done; ++i) {
        ok = (s->fused || s->dens || s->dg) ? s->iteration_fused(all_done) : s->iteration_callback(all_done, s->manual_have);
        if (!ok) break;
        if (launches_done) *launches_done += 1;
    }
    s->kernel_ms_acc = nullptr;
    if (ok) ok = hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
    if (ok && s->dg && s->fused_k > 0) ok = s->dg_check((int)((s->fused_k - 1) & 1));
    if (ok && (s->fused || s->dens || s->dg) && s->fused_k > 0 && !all_done) ok = s->check_counters((int)((s->fused_k - 1) & 1), all_done);   // (the last launch, too)
    if (ok && kernel_ms) {
        for (size_t i = 0; i < s->timed_launches && ok; ++i) {
            float ms = 0.f;
            ok = hip_ok(hipEventElapsedTime(&ms, s->tev[2 * i], s->tev[2 * i + 1]), "hipEventElapsedTime");
            *kernel_ms += (double)ms;
        }
    }
    s->seconds.store(s->seconds.load() + std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (!ok) { s->fail(t_error); return NPHIP_WAIT_ERROR; }
    if (all_done) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->finished = true;
        s->thread_done = true;
    }
    return all_done ? NPHIP_WAIT_DONE : NPHIP_WAIT_TIMEOUT;
}

int nphip_sampler_wait(nphip_sampler_t* s, int64_t timeout_ms) {
    if (s->manual) {
        auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            int rc = nphip_sampler_step(s, 1, nullptr, nullptr);
            if (rc != NPHIP_WAIT_TIMEOUT) return rc;
            if (timeout_ms >= 0 && std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(timeout_ms)) return NPHIP_WAIT_TIMEOUT;
        }
    }
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->want_pause) { s->want_pause = false; s->cv.notify_all(); }  // wait resumes (sample.py:596-608)
    auto done = [&] { return s->thread_done; };
    if (timeout_ms < 0) s->cv.wait(lk, done);
    else if (!s->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), done)) return NPHIP_WAIT_TIMEOUT;
    if (s->failed) { set_error(s->error); return NPHIP_WAIT_ERROR; }
    return NPHIP_WAIT_DONE;
}
int nphip_sampler_pause(nphip_sampler_t* s) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->want_pause = true;
    return NPHIP_OK;
}
int nphip_sampler_resume(nphip_sampler_t* s) {
    { std::lock_guard<std::mutex> lk(s->mu); s->want_pause = false; }
    s->cv.notify_all();
    return NPHIP_OK;
}
int nphip_sampler_abort(nphip_sampler_t* s) {
    { std::lock_guard<std::mutex> lk(s->mu); s->want_abort = true; if (s->manual) s->thread_done = true; }
    s->cv.notify_all();
    if (s->th.joinable()) s->th.join();
    return NPHIP_OK;
}
int nphip_sampler_is_finished(nphip_sampler_t* s) {
    std::lock_guard<std::mutex> lk(s->mu);
    return s->thread_done ? 1 : 0;
}
int nphip_sampler_waves_per_chain(const nphip_sampler_t* s) { return s->W; }
int nphip_default_evals_per_launch(uint64_t dim) { return default_evals_per_launch(dim); }
uint64_t nphip_sampler_num_chains(const nphip_sampler_t* s) { return s->n; }
uint64_t nphip_sampler_dim(const nphip_sampler_t* s) { return s->dim; }
uint64_t nphip_sampler_total_draws(const nphip_sampler_t* s) { return s->T; }
double nphip_sampler_seconds(const nphip_sampler_t* s) { return s->seconds.load(); }
uint64_t nphip_sampler_launches(const nphip_sampler_t* s) { return s->launches.load(); }
int nphip_sampler_host_mode(const nphip_sampler_t* s) {
    if (s->model.dense) return s->dg ? NPHIP_HOST_MODE_RESIDENT : (s->dg_fell_back ? NPHIP_HOST_MODE_FELL_BACK : NPHIP_HOST_MODE_LAUNCH_PER_EVALUATION);
    if (s->model.kind != 1) return NPHIP_HOST_MODE_NONE;
    if (s->remote) return NPHIP_HOST_MODE_RESIDENT;
    if (s->remote_fell_back) return NPHIP_HOST_MODE_FELL_BACK;
    return (s->n_groups > 0 && s->cb_groups == 0) ? NPHIP_HOST_MODE_GROUPS : NPHIP_HOST_MODE_LAUNCH_PER_EVALUATION;
}

static bool read_ctl(nphip_sampler_t* s, std::vector<Ctl>& h) {
    h.resize(s->n);
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return false;
    return hip_ok(hipMemcpy(h.data(), s->args.ctl, s->n * sizeof(Ctl), hipMemcpyDeviceToHost), "copy ctl");
}

int nphip_sampler_progress(nphip_sampler_t* s, uint64_t chain, nphip_chain_progress_t* out) {
    // chain == UINT64_MAX: fill out[0..n) for all local chains
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return NPHIP_ERR;
    const uint64_t lo = chain == UINT64_MAX ? 0 : chain, hi = chain == UINT64_MAX ? s->n : chain + 1;
    if (hi > s->n) { set_error("chain index out of range"); return NPHIP_ERR; }
    const uint64_t ms = (uint64_t)(s->seconds.load() * 1000.0);
    for (uint64_t i = lo; i < hi; ++i) {
        nphip_chain_progress_t& p = out[i - lo];
        p.finished_draws = (uint64_t)h[i].draw;
        p.total_draws = s->T;
        p.divergences = (uint64_t)h[i].n_div;
        p.tuning = h[i].draw < (int64_t)s->set.num_tune;
        p.started = h[i].phase != PH_START;
        p.latest_num_steps = (uint64_t)h[i].latest_steps;
        p.total_num_steps = (uint64_t)h[i].total_steps;
        p.step_size = h[i].step_size;
        p.runtime_ms = ms;
    }
    return NPHIP_OK;
}

int nphip_sampler_finished_draws(nphip_sampler_t* s, uint64_t* finished) {
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return NPHIP_ERR;
    for (uint64_t i = 0; i < s->n; ++i) finished[i] = (uint64_t)h[i].draw;
    return NPHIP_OK;
}

static void* stat_ptr(nphip_sampler_t* s, const std::string& n, size_t* bytes) {
    const Args& a = s->args;
    const size_t nt = (size_t)s->n * s->T, d = s->dim;
    struct E { const char* name; void* p; size_t b; };
    const E table[] = {
        {"draws", a.tr_draws, nt * d * 8}, {"gradient", a.tr_grad, nt * d * 8}, {"mass_matrix_inv", a.tr_mm, nt * d * 8},
        {"divergence_start", a.tr_div[0], nt * d * 8}, {"divergence_end", a.tr_div[1], nt * d * 8},
        {"divergence_momentum", a.tr_div[2], nt * d * 8}, {"divergence_start_gradient", a.tr_div[3], nt * d * 8},
        {"depth", a.st_depth, nt * 8}, {"n_steps", a.st_nsteps, nt * 8}, {"index_in_trajectory", a.st_idx, nt * 8},
        {"diverging", a.st_diverging, nt}, {"maxdepth_reached", a.st_maxdepth, nt}, {"tuning", a.st_tuning, nt},
        {"energy", a.st_energy, nt * 8}, {"energy_error", a.st_energy_error, nt * 8}, {"logp", a.st_logp, nt * 8},
        {"step_size", a.st_step, nt * 8}, {"step_size_bar", a.st_step_bar, nt * 8},
        {"mean_tree_accept", a.st_accept, nt * 8}, {"mean_tree_accept_sym", a.st_accept_sym, nt * 8},
    };
    for (const E& e : table)
        if (n == e.name) { *bytes = e.b; return e.p; }
    return nullptr;
}

int nphip_sampler_copy_expanded(nphip_sampler_t* s, void* host_out, uint64_t nbytes) {
    const nphip_model& m = s->model;
    const uint64_t E = m.expanded_dim, d = s->dim, rows = s->n * s->T;
    if (E == 0) { set_error("the model has no expand function"); return NPHIP_ERR; }
    if (!s->args.tr_draws) { set_error("the expand step needs the stored draws (store_draws)"); return NPHIP_ERR; }
    if (nbytes != rows * E * 8) { set_error("size mismatch for expanded"); return NPHIP_ERR; }
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return NPHIP_ERR;
    double* out = static_cast<double*>(host_out);
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return NPHIP_ERR;
    // blocks of rows: at most ~64 MB of positions per block
    const uint64_t block = std::max<uint64_t>(1, std::min<uint64_t>(rows, (64ull << 20) / (d * 8)));
    std::atomic<int> first_err{0};
    if (m.bs_expand) {
        // BridgeStan: the generated quantities draw from a per-chain bs_rng, so a chain's draws are expanded in order on one
        // thread (src/stan.rs:473-492) and the chains concurrently.  The rng's seed is the chain's own: word 0 of the Philox
        // block (settings.seed; 0, global chain, 0, NPHIP_RNG_EXPAND) — the reference takes `rng.next_u32()` of the chain's
        // generator (src/stan.rs:787-788); independent of sharding either way.
        const BsExpand& B = *m.bs_expand;
        const uint64_t T = s->T, chains_per_block = std::max<uint64_t>(1, block / std::max<uint64_t>(1, T));
        std::vector<double> x(std::min<uint64_t>(s->n, chains_per_block) * T * d);
        int nt = (int)std::min<uint64_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
        if (m.n_threads > 0) nt = m.n_threads;
        RowPool pool(nt > 1 ? nt : 0);
        std::mutex err_mu;
        std::string err_text;
        for (uint64_t c0 = 0; c0 < s->n; c0 += chains_per_block) {
            const uint64_t nc = std::min(chains_per_block, s->n - c0);
            if (!hip_ok(hipMemcpy(x.data(), s->args.tr_draws + c0 * T * d, nc * T * d * 8, hipMemcpyDeviceToHost), "copy draws")) return NPHIP_ERR;
            pool.run(nc, [&](uint64_t r) {
                const uint64_t chain = c0 + r;
                const nphip_u32x4 blk = nphip_philox(s->set.seed, 0u, (uint32_t)(s->launch.chain_offset + chain), 0u, NPHIP_RNG_EXPAND);
                char* em = nullptr;
                void* rng = B.rng_new(blk.v[0], &em);
                std::vector<double> theta(E);
                for (uint64_t draw = 0; draw < T; ++draw) {
                    double* o = out + (chain * T + draw) * E;
                    if (!rng || (int64_t)draw >= h[chain].draw) { for (uint64_t e = 0; e < E; ++e) o[e] = NAN; continue; }
                    const int rc = B.constrain(B.model, true, true, x.data() + (r * T + draw) * d, theta.data(), rng, &em);
                    if (rc != 0) {
                        int z = 0;
                        if (first_err.compare_exchange_strong(z, rc)) { std::lock_guard<std::mutex> lk(err_mu); err_text = em ? em : ""; }
                        if (em && B.free_err) B.free_err(em);
                        em = nullptr;
                        for (uint64_t e = 0; e < E; ++e) o[e] = NAN;
                        continue;
                    }
                    if (B.perm.empty()) memcpy(o, theta.data(), E * 8);
                    else for (uint64_t e = 0; e < E; ++e) o[e] = theta[B.perm[e]];
                }
                if (!rng) { int z = 0; if (first_err.compare_exchange_strong(z, -1)) { std::lock_guard<std::mutex> lk(err_mu); err_text = em ? em : "bs_rng_construct failed"; } if (em && B.free_err) B.free_err(em); }
                else B.rng_free(rng);
            });
        }
        if (first_err.load() != 0) {   // src/stan.rs:493-494
            set_error("Failed to constrain the parameters of the draw" + (err_text.empty() ? std::string() : ": " + err_text));
            return NPHIP_ERR;
        }
        return NPHIP_OK;
    } else if (m.expand_fn) {
        std::vector<double> x(block * d);
        int nt = (int)std::min<uint64_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
        if (m.n_threads > 0) nt = m.n_threads;
        RowPool pool(nt > 1 ? nt : 0);
        for (uint64_t lo = 0; lo < rows; lo += block) {
            const uint64_t nb = std::min(block, rows - lo);
            if (!hip_ok(hipMemcpy(x.data(), s->args.tr_draws + lo * d, nb * d * 8, hipMemcpyDeviceToHost), "copy draws")) return NPHIP_ERR;
            pool.run(nb, [&](uint64_t r) {
                const uint64_t row = lo + r, chain = row / s->T, draw = row % s->T;
                double* o = out + row * E;
                if ((int64_t)draw >= h[chain].draw) { for (uint64_t e = 0; e < E; ++e) o[e] = NAN; return; }
                const int rc = m.expand_fn(d, E, x.data() + r * d, o, m.expand_user);
                if (rc != 0) { int z = 0; first_err.compare_exchange_strong(z, rc); }
            });
        }
    } else {
        double* dout = nullptr;
        if (!hip_ok(hipMalloc((void**)&dout, block * E * 8), "hipMalloc")) return NPHIP_ERR;
        bool ok = true;
        for (uint64_t lo = 0; ok && lo < rows; lo += block) {
            const uint64_t nb = std::min(block, rows - lo);
            const int rc = m.expand_dev_fn(nb, d, E, s->args.tr_draws + lo * d, dout, (void*)s->stream, m.expand_user);
            if (rc != 0) { first_err.store(rc); break; }
            ok = hip_ok(hipStreamSynchronize(s->stream), "sync") && hip_ok(hipMemcpy(out + lo * E, dout, nb * E * 8, hipMemcpyDeviceToHost), "copy expanded");
        }
        (void)hipFree(dout);
        if (!ok) return NPHIP_ERR;
        for (uint64_t row = 0; row < rows; ++row)
            if ((int64_t)(row % s->T) >= h[row / s->T].draw)
                for (uint64_t e = 0; e < E; ++e) out[row * E + e] = NAN;
    }
    if (first_err.load() != 0) { set_error("Expand function returned error code " + std::to_string(first_err.load())); return NPHIP_ERR; }
    return NPHIP_OK;
}

int nphip_sampler_copy_stat(nphip_sampler_t* s, const char* name, void* host_out, uint64_t nbytes) {
    if (std::string(name) == "expanded") return nphip_sampler_copy_expanded(s, host_out, nbytes);
    size_t bytes = 0;
    void* p = stat_ptr(s, name, &bytes);
    if (!p) { set_error(std::string("trace has no array named ") + name); return NPHIP_ERR; }
    if (nbytes != bytes) { set_error("size mismatch for " + std::string(name)); return NPHIP_ERR; }
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return NPHIP_ERR;
    if (!hip_ok(hipMemcpy(host_out, p, bytes, hipMemcpyDeviceToHost), "copy trace")) return NPHIP_ERR;
    return NPHIP_OK;
}

int64_t nphip_sampler_waiting(nphip_sampler_t* s, uint8_t* mask) {
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return -1;
    int64_t cnt = 0;
    for (uint64_t i = 0; i < s->n; ++i) {
        const bool w = h[i].phase == PH_WAIT_HOST;
        if (mask) mask[i] = w ? 1 : ((h[i].phase == PH_DONE || h[i].phase == PH_ERROR) ? 2 : 0);
        cnt += w ? 1 : 0;
    }
    return cnt;
}

int nphip_sampler_resume_at(nphip_sampler_t* s, uint64_t n, const uint64_t* chains, const double* positions, int on_device) {
    if (!s->manual) { set_error("nphip_sampler_resume_at needs a sampler created with launch.manual = 1"); return NPHIP_ERR; }
    if (n == 0) return NPHIP_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (chains[i] >= s->n) { set_error("chain index out of range"); return NPHIP_ERR; }
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return NPHIP_ERR;
    std::vector<int64_t> ch(chains, chains + n);
    int64_t* d_ch = (int64_t*)s->hand_in_buf(0, n * 8);
    double* d_pos = nullptr;
    bool ok = d_ch != nullptr && hip_ok(hipMemcpy(d_ch, ch.data(), n * 8, hipMemcpyHostToDevice), "H2D chains");
    if (ok && !on_device) {
        d_pos = (double*)s->hand_in_buf(2, n * s->dim * 8);
        ok = d_pos != nullptr && hip_ok(hipMemcpy(d_pos, positions, n * s->dim * 8, hipMemcpyHostToDevice), "H2D positions");
    }
    if (ok) ok = hip_ok(launch_resume(s->d_args, (int)n, d_ch, on_device ? positions : d_pos, s->fused, s->stream), "launch k_resume") &&
                 hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
    // callback models: the staged positions changed after the last evaluation — the next launch must not consume its results
    s->manual_have = 0;
    return ok ? NPHIP_OK : NPHIP_ERR;
}

static int hand_in_metric(nphip_sampler_t* s, uint64_t n, const uint64_t* chains, uint64_t k, const double* sig2, const double* V, const double* lam,
                          int on_device, bool staged, uint64_t* n_taken) {
    if (!s->manual) { set_error(std::string(staged ? "nphip_sampler_stage_metric" : "nphip_sampler_set_metric") + " needs a sampler created with launch.manual = 1"); return NPHIP_ERR; }
    if (!s->set.low_rank_metric) { set_error("the sampler was not created for host-supplied metrics (settings: low_rank_metric)"); return NPHIP_ERR; }
    if (k > (uint64_t)kLrMax) { set_error("at most " + std::to_string(kLrMax) + " low-rank columns"); return NPHIP_ERR; }
    if (n == 0) return NPHIP_OK;
    if (!sig2 || (k > 0 && (!V || !lam))) { set_error("set_metric needs sigma^2, and V and lambda when k > 0"); return NPHIP_ERR; }
    for (uint64_t i = 0; i < n; ++i)
        if (chains[i] >= s->n) { set_error("chain index out of range"); return NPHIP_ERR; }
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return NPHIP_ERR;
    std::vector<int64_t> ch(chains, chains + n);
    int taken = 0;
    double *d_s = nullptr, *d_v = nullptr, *d_l = nullptr;
    const size_t bs = n * s->dim * 8, bv = n * k * s->dim * 8, bl = n * k * 8;
    // (the sampler's own grow-only buffers: ADVICE r4 / VERDICT r5 — no allocation, hence no device-wide synchronisation, per hand-in)
    int64_t* d_ch = (int64_t*)s->hand_in_buf(0, n * 8);
    int* d_taken = (int*)s->hand_in_buf(1, sizeof(int));
    bool ok = d_ch != nullptr && d_taken != nullptr && hip_ok(hipMemcpy(d_ch, ch.data(), n * 8, hipMemcpyHostToDevice), "H2D chains") &&
              hip_ok(hipMemset(d_taken, 0, sizeof(int)), "hipMemset");
    if (ok && !on_device) {
        d_s = (double*)s->hand_in_buf(2, bs);
        ok = d_s != nullptr && hip_ok(hipMemcpy(d_s, sig2, bs, hipMemcpyHostToDevice), "H2D sigma^2");
        if (ok && k > 0) {
            d_v = (double*)s->hand_in_buf(3, bv);
            d_l = (double*)s->hand_in_buf(4, bl);
            ok = d_v != nullptr && d_l != nullptr && hip_ok(hipMemcpy(d_v, V, bv, hipMemcpyHostToDevice), "H2D V") && hip_ok(hipMemcpy(d_l, lam, bl, hipMemcpyHostToDevice), "H2D lambda");
        }
    }
    if (ok) ok = hip_ok((staged ? launch_stage_metric : launch_set_metric)(s->d_args, (int)n, d_ch, (int)k, on_device ? sig2 : d_s, on_device ? V : d_v, on_device ? lam : d_l, d_taken, s->stream),
                        staged ? "launch k_stage_metric" : "launch k_set_metric") && hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize") &&
                 hip_ok(hipMemcpy(&taken, d_taken, sizeof(int), hipMemcpyDeviceToHost), "D2H taken");
    if (n_taken) *n_taken = (uint64_t)taken;
    if (staged) return ok ? NPHIP_OK : NPHIP_ERR;   // (the chains run on: nothing of theirs changed yet; a chain past its warm-up is simply not counted)
    s->manual_have = 0;   // (callback models: the staged evaluation belongs to the state before the pause)
    if (ok && (uint64_t)taken != n) {
        // (a metric for a chain that is not stopped used to be dropped without a word)
        set_error(std::to_string(n - (uint64_t)taken) + " of " + std::to_string(n) + " chains were not stopped at a pause draw (nphip_sampler_waiting): they keep their metric" +
                  (taken ? ", the other " + std::to_string(taken) + " took the new one" : ""));
        return NPHIP_ERR;
    }
    return ok ? NPHIP_OK : NPHIP_ERR;
}

int nphip_sampler_set_metric(nphip_sampler_t* s, uint64_t n, const uint64_t* chains, uint64_t k, const double* sig2, const double* V, const double* lam,
                             int on_device) {
    return hand_in_metric(s, n, chains, k, sig2, V, lam, on_device, false, nullptr);
}

int nphip_sampler_stage_metric(nphip_sampler_t* s, uint64_t n, const uint64_t* chains, uint64_t k, const double* sig2, const double* V, const double* lam,
                               int on_device, uint64_t* n_taken) {
    return hand_in_metric(s, n, chains, k, sig2, V, lam, on_device, true, n_taken);
}

// Chains stopped at a pause draw go on exactly as if they had not stopped (nphip_sampler_release): the next launch begins their next draw
// (PH_DRAW_BEGIN: momentum refresh, first doubling) under the metric, step size and adaptation state they stopped with.  One thread per chain.
__global__ void k_release(const Args* __restrict__ Ap, int n, const int64_t* __restrict__ chains, int* __restrict__ taken) {
    const Args& A = *Ap;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    Ctl* c = A.ctl + chains[i];
    if (c->phase != PH_WAIT_HOST) return;
    c->phase = PH_DRAW_BEGIN;
    atomicAdd(taken, 1);
}

int nphip_sampler_release(nphip_sampler_t* s, uint64_t n, const uint64_t* chains) {
    if (!s->manual) { set_error("nphip_sampler_release needs a sampler created with launch.manual = 1"); return NPHIP_ERR; }
    if (n == 0) return NPHIP_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (chains[i] >= s->n) { set_error("chain index out of range"); return NPHIP_ERR; }
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    (void)hipSetDevice(s->device);
    if (!s->sync_all()) return NPHIP_ERR;
    std::vector<int64_t> ch(chains, chains + n);
    int taken = 0;
    int64_t* d_ch = (int64_t*)s->hand_in_buf(0, n * 8);
    int* d_taken = (int*)s->hand_in_buf(1, sizeof(int));
    bool ok = d_ch != nullptr && d_taken != nullptr && hip_ok(hipMemcpy(d_ch, ch.data(), n * 8, hipMemcpyHostToDevice), "H2D chains") &&
              hip_ok(hipMemset(d_taken, 0, sizeof(int)), "hipMemset");
    if (ok) {
        hipLaunchKernelGGL(k_release, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_args, (int)n, d_ch, d_taken);
        ok = hip_ok(hipGetLastError(), "launch k_release") && hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize") &&
             hip_ok(hipMemcpy(&taken, d_taken, sizeof(int), hipMemcpyDeviceToHost), "D2H taken");
    }
    s->manual_have = 0;   // (callback models: the staged evaluation belongs to the state before the pause)
    if (ok && (uint64_t)taken != n) {
        set_error(std::to_string(n - (uint64_t)taken) + " of " + std::to_string(n) + " chains were not stopped at a pause draw (nphip_sampler_waiting)");
        return NPHIP_ERR;
    }
    return ok ? NPHIP_OK : NPHIP_ERR;
}

int nphip_sampler_set_evals_per_launch(nphip_sampler_t* s, int32_t evals) {
    if (!s->manual) { set_error("nphip_sampler_set_evals_per_launch needs a sampler created with launch.manual = 1"); return NPHIP_ERR; }
    if (evals < 0) { set_error("evals_per_launch must be >= 0 (0: the default)"); return NPHIP_ERR; }
    std::lock_guard<std::mutex> run_lk(s->mu_run);
    s->launch.evals_per_launch = evals;
    return NPHIP_OK;
}

int64_t nphip_sampler_chain_draws(nphip_sampler_t* s, int64_t* draws, uint8_t* state) {
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return -1;
    int64_t running = 0;
    for (uint64_t i = 0; i < s->n; ++i) {
        if (draws) draws[i] = h[i].draw;
        const uint8_t st = h[i].phase == PH_WAIT_HOST ? 1 : ((h[i].phase == PH_DONE || h[i].phase == PH_ERROR) ? 2 : 0);
        if (state) state[i] = st;
        running += st == 0 ? 1 : 0;
    }
    return running;
}

int nphip_sampler_profile(nphip_sampler_t* s, int64_t out[16]) {
    std::vector<Ctl> h;
    if (!read_ctl(s, h)) return NPHIP_ERR;
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (auto& c : h) for (int k = 0; k < 16; ++k) out[k] += c.prof[k];
    return NPHIP_OK;
}

void* nphip_sampler_device_ptr(nphip_sampler_t* s, const char* name) {
    size_t bytes = 0;
    return stat_ptr(s, name, &bytes);
}

// ---- dense linear algebra for the low-rank estimator (linalg.hip) ------------------------
extern "C" int nphip_linalg_launch_eigh(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, int* status_device, void* stream, int mode);

static int batched_eigh_mode(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream, int mode);
int nphip_batched_eigh(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream) {
    return batched_eigh_mode(n_batch, order, a_device, w_device, stream, 0);
}
// test hook: the stages of nphip_batched_eigh on their own (mode 1: the tridiagonal form — w = diagonal, row 0 of a = sub-diagonal; 2: a = Q)
int nphip_test_eigh_stage(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream, int mode) {
    return batched_eigh_mode(n_batch, order, a_device, w_device, stream, mode);
}
static int batched_eigh_mode(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream, int mode) {
    if (order == 0 || order > 128) { set_error("nphip_batched_eigh: the order must be 1..128 (the matrix lives in LDS)"); return NPHIP_ERR; }
    if (n_batch == 0) return NPHIP_OK;
    if (!a_device || !w_device) { set_error("nphip_batched_eigh: null device pointer"); return NPHIP_ERR; }
    // The status buffer is kept between calls (one per device, grown on demand): hipFree synchronises the whole device — the estimator's
    // worker thread would wait for the engine's launch on another stream, which is exactly the overlap the low-rank driver is built
    // around.  The mutex serialises callers that share the buffer.
    static std::mutex mu;
    static std::map<int, std::pair<int*, uint64_t>> cache;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto& slot = cache[dev];
    if (slot.second < n_batch) {
        if (slot.first) (void)hipFree(slot.first);
        slot = {nullptr, 0};
        const uint64_t cap = std::max<uint64_t>(n_batch, 1024);
        if (!hip_ok(hipMalloc((void**)&slot.first, cap * sizeof(int)), "hipMalloc")) return NPHIP_ERR;
        slot.second = cap;
    }
    int* d_status = slot.first;
    std::vector<int> st(n_batch, 0);
    bool ok = hip_ok((hipError_t)nphip_linalg_launch_eigh(n_batch, order, a_device, w_device, d_status, stream, mode), "launch k_batched_eigh") &&
              hip_ok(hipMemcpyAsync(st.data(), d_status, n_batch * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H status") &&
              hip_ok(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    if (!ok) return NPHIP_ERR;
    for (uint64_t i = 0; i < n_batch; ++i)
        if (st[i] != 0) { set_error("nphip_batched_eigh: the QL iteration of matrix " + std::to_string(i) + " did not converge"); return NPHIP_ERR; }
    return NPHIP_OK;
}

// ---- test hooks ------------------------------------------------------------------------
int nphip_test_detmath(int device, int fn, uint64_t n, const double* x, double* y) {
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device"); return NPHIP_ERR; }
    double *dx = nullptr, *dy = nullptr;
    const uint64_t nx = fn == 7 ? 4 : n;
    if (!hip_ok(hipMalloc((void**)&dx, nx * 8), "hipMalloc") || !hip_ok(hipMalloc((void**)&dy, n * 8 + 8), "hipMalloc")) return NPHIP_ERR;
    bool ok = hip_ok(hipMemcpy(dx, x, nx * 8, hipMemcpyHostToDevice), "H2D") &&
              hip_ok(launch_test_detmath(fn, n, dx, dy, nullptr), "launch") &&
              hip_ok(hipMemcpy(y, dy, n * 8, hipMemcpyDeviceToHost), "D2H");
    (void)hipFree(dx); (void)hipFree(dy);
    return ok ? NPHIP_OK : NPHIP_ERR;
}

int nphip_test_mfma_f64_rate(int device, double* tflops) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return NPHIP_ERR; }
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return NPHIP_ERR;
    const int blocks = prop.multiProcessorCount * 2, iters = 4096;   // two workgroups of four waves per CU: two waves per SIMD
    double* d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hip_ok(hipMalloc((void**)&d, 64), "hipMalloc") && hip_ok(hipEventCreate(&e0), "event") && hip_ok(hipEventCreate(&e1), "event");
    float best = 0.f;
    for (int rep = 0; ok && rep < 4; ++rep) {   // (the first repetition warms the clocks up)
        ok = hip_ok(hipEventRecord(e0, nullptr), "record") && hip_ok(launch_mfma_f64_rate(d, blocks, iters, nullptr), "launch") &&
             hip_ok(hipEventRecord(e1, nullptr), "record") && hip_ok(hipEventSynchronize(e1), "sync");
        float ms = 0.f;
        ok = ok && hip_ok(hipEventElapsedTime(&ms, e0, e1), "elapsed");
        if (ok && rep > 0 && (best == 0.f || ms < best)) best = ms;
    }
    if (ok) *tflops = (double)blocks * 4.0 * iters * 8.0 * 2048.0 / ((double)best * 1e-3) / 1e12;
    if (ok && getenv("NPHIP_DEBUG")) {
        double h[3] = {0, 0, 0};
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        fprintf(stderr, "nphip: fp64 MFMA rate: %.1f TFLOP/s over %.3f ms; one wave: %.1f shader cycles per MFMA issued (two waves per SIMD), shader clock %.0f MHz\n",
                *tflops, best, h[1] / (iters * 8.0), h[1] / (h[2] / 100.0));
    }
    if (d) (void)hipFree(d);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return ok ? NPHIP_OK : NPHIP_ERR;
}
int nphip_test_dense_grad(int device, int waves, uint64_t n, uint64_t dim, const double* x, const double* mu, const double* P, double* grad, double* logp) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return NPHIP_ERR; }
    const size_t KP = (dim + 15) / 16 * 16, DP = (dim + 63) / 64 * 64;
    double *dx = nullptr, *dg = nullptr, *dl = nullptr, *dP = nullptr, *dmu = nullptr;
    bool ok = hip_ok(hipMalloc((void**)&dx, n * dim * 8), "hipMalloc") && hip_ok(hipMalloc((void**)&dg, n * dim * 8), "hipMalloc") &&
              hip_ok(hipMalloc((void**)&dl, n * 8), "hipMalloc") && hip_ok(hipMalloc((void**)&dP, DP * KP * 8), "hipMalloc") &&
              hip_ok(hipMalloc((void**)&dmu, KP * 8), "hipMalloc");
    ok = ok && hip_ok(hipMemset(dP, 0, DP * KP * 8), "hipMemset") && hip_ok(hipMemset(dmu, 0, KP * 8), "hipMemset") &&
         hip_ok(hipMemcpy2D(dP, KP * 8, P, dim * 8, dim * 8, dim, hipMemcpyHostToDevice), "H2D P") &&
         hip_ok(hipMemcpy(dmu, mu, dim * 8, hipMemcpyHostToDevice), "H2D mu") && hip_ok(hipMemcpy(dx, x, n * dim * 8, hipMemcpyHostToDevice), "H2D x");
    ok = ok && hip_ok(launch_dense_grad(dx, dP, dmu, dg, dl, (int64_t)n, (int64_t)dim, (int64_t)KP, waves, nullptr), "launch dense grad") &&
         hip_ok(hipDeviceSynchronize(), "sync") && hip_ok(hipMemcpy(grad, dg, n * dim * 8, hipMemcpyDeviceToHost), "D2H grad") &&
         hip_ok(hipMemcpy(logp, dl, n * 8, hipMemcpyDeviceToHost), "D2H logp");
    for (double* q : {dx, dg, dl, dP, dmu}) if (q) (void)hipFree(q);
    return ok ? NPHIP_OK : NPHIP_ERR;
}
int nphip_test_dot(int device, int waves, uint64_t n, const double* x, const double* y, double* out) {
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device"); return NPHIP_ERR; }
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    if (!hip_ok(hipMalloc((void**)&dx, n * 8 + 8), "hipMalloc") || !hip_ok(hipMalloc((void**)&dy, n * 8 + 8), "hipMalloc") ||
        !hip_ok(hipMalloc((void**)&dout, 8), "hipMalloc")) return NPHIP_ERR;
    bool ok = hip_ok(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice), "H2D") && hip_ok(hipMemcpy(dy, y, n * 8, hipMemcpyHostToDevice), "H2D") &&
              hip_ok(launch_test_dot(waves, n, dx, dy, dout, nullptr), "launch") && hip_ok(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost), "D2H");
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return ok ? NPHIP_OK : NPHIP_ERR;
}

// Host-side test hook (no GPU involved): `batches` batches of `rows` rows on an evaluation pool of `threads` threads, `use` of
// them per batch (0 = all); row r of batch b adds (b + 1) * (r + 1) into out[r].  Also reports the cores the pool sizing sees.
int nphip_test_rowpool(int threads, uint64_t rows, int batches, int use, uint64_t* out, int* usable_cores_out) {
    if (usable_cores_out) *usable_cores_out = usable_cores();
    RowPool pool(threads);
    for (int b = 0; b < batches; ++b) {
        const std::function<void(uint64_t)> f = [out, b](uint64_t r) { out[r] += (uint64_t)(b + 1) * (r + 1); };
        pool.run(rows, f, use);
        if ((b % 7) == 3) std::this_thread::sleep_for(std::chrono::microseconds(600));   // lets the workers fall asleep in between
    }
    return NPHIP_OK;
}

}  // extern "C"
