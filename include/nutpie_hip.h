/*
 * nutpie_hip.h — C-ABI of libnutpie_hip.so, the MI355X-native NUTS engine.
 *
 * Drop-in boundary: these entry points are what nutpie's PyO3 layer
 * (reference src/wrapper.rs) would bind instead of `nuts_rs::Sampler`.  Every
 * function is `extern "C"`, takes plain pointers/sizes, never throws, and
 * reports failure through an int status + nphip_last_error().
 *
 *   reference interface                                 replaced by
 *   --------------------------------------------------  -------------------------------
 *   PyNutsSettings::Diag / apply_update / as_dict       nphip_settings_*
 *     (src/wrapper.rs:525-533, 563-620, 210-451, 751-769)
 *   LogpFunc / RawLogpFunc  (src/pymc.rs:21-62)         nphip_model_host_callback
 *   ExpandFunc / RawExpandFunc (src/pymc.rs:31-37, 64-95, nphip_model_set_expand, nphip_sampler_copy_expanded
 *     217-286)
 *   PyModel  (src/pyfunc.rs:206-230, 517-570)           nphip_model_device_callback, nphip_model_jit_density
 *   StanModel::logp (src/stan.rs:454-463)               nphip_model_bridgestan (adapter onto the host-callback path)
 *   StanDensity::expand_vector (src/stan.rs:473-520)    nphip_model_set_bridgestan_expand
 *   nuts_rs::Sampler::new (src/wrapper.rs:977-1085)     nphip_sampler_create
 *   PySampler::{wait,pause,resume,abort,is_finished,    nphip_sampler_{wait,pause,resume,abort,
 *     inspect,take_results} (src/wrapper.rs:1252-1456)    is_finished,trace_*}
 *   ChainProgress (src/wrapper.rs:47-104)               nphip_chain_progress_t
 *
 * Threading: every entry point may be called from any host thread; a sampler is
 * guarded by one mutex (as the reference's Mutex<SamplerState>, wrapper.rs:954).
 * The engine owns one host driver thread per sampler; `wait` blocks the caller
 * (the Python binding releases the GIL around it, as wrapper.rs:1305-1330 does).
 */
#ifndef NUTPIE_HIP_H
#define NUTPIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPHIP_OK 0
#define NPHIP_ERR (-1)
/* settings errors mirror the two Python exception classes of wrapper.rs:138-145, 610-614 */
#define NPHIP_ERR_UNKNOWN_ATTR (-2)   /* AttributeError("Unknown settings attribute: ...")            */
#define NPHIP_ERR_NOT_AVAILABLE (-3)  /* ValueError("Option ... not available for ... adaptation")     */
#define NPHIP_ERR_BAD_VALUE (-4)      /* ValueError(...)                                               */

/* nphip_sampler_wait results (SamplerWaitResult, wrapper.rs:1099-1143) */
#define NPHIP_WAIT_DONE 0
#define NPHIP_WAIT_TIMEOUT 1
#define NPHIP_WAIT_ERROR 2

const char* nphip_last_error(void);
const char* nphip_version(void);
int nphip_device_count(void);

/* ------------------------------------------------------------------ settings */
typedef struct nphip_settings nphip_settings_t;

/* PyNutsSettings.Diag(seed) — wrapper.rs:721-723.  adaptation "diag"/"draw_diag" only;
 * LowRank/Flow/MCLMC are out of scope for this engine. */
nphip_settings_t* nphip_settings_new_diag(uint64_t seed);
nphip_settings_t* nphip_settings_clone(const nphip_settings_t*);
void nphip_settings_free(nphip_settings_t*);
/* Flat attribute names exactly as wrapper.rs:213-447, 565-609 accepts them. */
int nphip_settings_set_f64(nphip_settings_t*, const char* name, double v);
int nphip_settings_set_u64(nphip_settings_t*, const char* name, uint64_t v);
int nphip_settings_set_bool(nphip_settings_t*, const char* name, int v);
int nphip_settings_set_str(nphip_settings_t*, const char* name, const char* v);
int nphip_settings_get_f64(const nphip_settings_t*, const char* name, double* out);
int nphip_settings_get_u64(const nphip_settings_t*, const char* name, uint64_t* out);
/* Host-driven adaptation hook (used by the low-rank metric, nutpie_amd/low_rank.py — the N4 row of SURVEY.md §8f): every
 * chain stops between two draws when it has finished exactly draws[i] draws (at most 16 entries, increasing); see
 * nphip_sampler_waiting / nphip_sampler_resume_at. */
int nphip_settings_set_pause_draws(nphip_settings_t*, uint64_t n, const uint64_t* draws);
/* JSON of the nested settings (the "settings" value of as_dict(), wrapper.rs:755-769);
 * returns needed length incl. NUL; writes at most cap bytes. */
int64_t nphip_settings_to_json(const nphip_settings_t*, char* buf, int64_t cap);

/* ------------------------------------------------------------------ models */
typedef struct nphip_model nphip_model_t;

/* The reference's raw C logp callback, verbatim: src/pymc.rs:23-29 /
 * python/nutpie/compile_pymc.py:975-981.  0 ok, >0 recoverable (=> divergence), <0 fatal
 * (src/pymc.rs:166-180).  Must be re-entrant: called concurrently from up to n_threads host threads (n_threads is an upper bound, 0 = the
 * usable cores: the engine evaluates a batch of rows on as many threads as its measured cost is worth, one for cheap rows). */
typedef int (*nphip_raw_logp_fn)(uint64_t dim, const double* x, double* grad_out, double* logp_out, void* user_data);
/* (The return type is the reference's `std::os::raw::c_int`.  A numba cfunc declared `int64(...)`, compile_pymc.py:975-981,
 * leaves its code in the full return register; like the reference, the engine reads the low 32 bits.) */

/* The reference's raw expand callback, verbatim: src/pymc.rs:31-37 (`RawExpandFunc`), numba side
 * python/nutpie/compile_pymc.py:1018-1041.  Maps ONE unconstrained draw x[dim] to the flat vector of all expanded
 * (constrained / deterministic) variables out[expanded_dim]; 0 = ok, anything else is an error
 * ("Expand function returned error code N", src/pymc.rs:276-283).  Re-entrant: rows are expanded concurrently. */
typedef int (*nphip_raw_expand_fn)(uint64_t dim, uint64_t expanded_dim, const double* x, double* out, void* user_data);
/* Batched device form of the expand step: x[n_rows][dim] -> out[n_rows][expanded_dim], both device,
 * fp64, row-major; work enqueued on `stream`.  Return 0, or nonzero for an error. */
typedef int (*nphip_device_expand_fn)(uint64_t n_rows, uint64_t dim, uint64_t expanded_dim, const double* x, double* out,
                                      void* stream, void* user_data);

/* Batched device callback (the GPU form of src/pyfunc.rs:206-230): when called, the engine's
 * staging buffer q[n_chains][dim] (device, fp64, row-major) holds the positions; the callee must
 * fill grad[n_chains][dim] and logp[n_chains] (device) with work enqueued on `stream`.
 * Non-finite logp => recoverable (src/pyfunc.rs:218-220).  Return 0, or <0 for a fatal error. */
typedef int (*nphip_device_logp_fn)(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp,
                                    void* stream, void* user_data);

/* Fused analytic model: logp(x) = -1/2 (x-mu)' L (x-mu), L symmetric tridiagonal (diag[dim],
 * offdiag[dim-1]); mu/offdiag may be NULL.  Covers N(0,I), diagonal and AR(1) Gaussians.
 * Host pointers; copied. */
nphip_model_t* nphip_model_tridiag_gaussian(uint64_t dim, const double* mu, const double* diag, const double* offdiag);
/* Dense-precision Gaussian: logp(x) = -1/2 (x-mu)' P (x-mu), P symmetric [dim][dim] row-major (BASELINE.json configs[1] read as a DENSE
 * correlated Gaussian — SURVEY.md 8d variant (ii)); mu may be NULL.  Host pointers; copied.  The reference has no such model: its user
 * writes the density as a per-chain callable (src/pyfunc.rs:206-230) and nuts-rs calls it once per chain and leapfrog; here the gradients
 * of ALL chains of a step are one fp64 GEMM on the matrix cores, hand-written (csrc/dense_tile.h: v_mfma_f64_16x16x4_f64, fixed summation
 * order — include/nphip_spec.h), inside the engine: no callback, no staging round trip through a framework. */
nphip_model_t* nphip_model_dense_gaussian(uint64_t dim, const double* mu, const double* P);
nphip_model_t* nphip_model_host_callback(uint64_t dim, nphip_raw_logp_fn fn, void* user_data, int n_threads);
nphip_model_t* nphip_model_device_callback(uint64_t dim, nphip_device_logp_fn fn, void* user_data);
/* BridgeStan flavour (reference src/stan.rs:454-463): each row is evaluated as
 *   bs_log_density_gradient(bs_model, propto=true, jacobian=true, theta, &val, grad, &err)
 * on the host pool; any Stan error or a non-finite density is recoverable (src/stan.rs:392-396,
 * 459-461).  `log_density_gradient` / `free_error_msg` are the addresses of the BridgeStan C API
 * functions of the loaded model library (bridgestan.h), `bs_model` the model handle. */
nphip_model_t* nphip_model_bridgestan(uint64_t dim, void* bs_model, void* log_density_gradient, void* free_error_msg, int n_threads);
/* Runtime-compiled device density (the GPU form of a compiled model's logp function: compile_pymc.py:668-871 builds one per
 * model; here the model is HIP source compiled at run time into its own instantiation of the engine's resident kernel —
 * nutpie_amd/density.py).  `launch_fn` = address of `nphip_jit_launch` of the model's library, `nv` = its `nphip_jit_nv()`
 * (chunks of 128 dimensions per wave: dim <= 1024), `waves_per_chain` = its `nphip_jit_w()` (1, 2 or 4 wavefronts evaluate one
 * chain's density together; 0 = 1), `data_device` = the model's data block in device memory (borrowed),
 * `lds_bytes_per_chain` = LDS scratch the density uses per chain, `lds_bytes_shared` = LDS common to the chains of a
 * workgroup (four with one wave per chain, else one), filled once per launch by the library's staging function (the model's
 * data).  The evaluation is a call in the middle of the register-resident leaf: no launch, no memory round trip for the chain
 * state.  The same library exports the density as a batched device callback (`nphip_jit_logp`, an nphip_device_logp_fn) for
 * everything the resident kernel does not cover. */
nphip_model_t* nphip_model_jit_density(uint64_t dim, void* launch_fn, int nv, const void* data_device, uint64_t lds_bytes_per_chain,
                                       uint64_t lds_bytes_shared, int waves_per_chain);
/* The library's resident kernel was compiled for the low-rank metric (-DNPHIP_JIT_LR=1: `nphip_jit_lr()` of the library returns 1;
 * one, two or four waves per chain): a job with `low_rank_metric` (adaptation = "low_rank", src/wrapper.rs:307-334) then keeps the register-resident leaf —
 * the cursor's velocity M^-1 p as a sixth resident vector, the columns of V streamed against it — instead of the batched callback. */
int nphip_model_jit_low_rank(nphip_model_t*, int capable);
/* Initial positions: kind 0 = U(-2,2) (src/pyfunc.rs:540-544), 1 = N(0,1) (src/stan.rs:798-808),
 * 2 = explicit host array points[n_chains_total][dim] indexed by GLOBAL chain id
 * (src/pymc.rs:505-534 evaluates the user's init function per chain on the host). */
int nphip_model_set_init(nphip_model_t*, int kind, const double* points, uint64_t n_points);
/* ExpandFunc::new(dim, expanded_dim, ptr, user_data_ptr, keep_alive) — src/pymc.rs:74-95.  Any model flavour may carry
 * an expand function; the host form is evaluated on the model's host thread pool, the device form on the engine's stream. */
int nphip_model_set_expand(nphip_model_t*, uint64_t expanded_dim, nphip_raw_expand_fn fn, void* user_data);
int nphip_model_set_device_expand(nphip_model_t*, uint64_t expanded_dim, nphip_device_expand_fn fn, void* user_data);
/* BridgeStan flavour of the expand step (reference src/stan.rs:473-520, 787-796): every stored draw is expanded with
 *   bs_param_constrain(bs_model, include_tp = true, include_gq = true, theta_unc, theta, rng, &err)
 * using ONE bs_rng per chain, as the reference does (StanModel::math builds it from the chain's generator; here its seed is
 * word 0 of the Philox block (settings.seed; 0, global chain, 0, NPHIP_RNG_EXPAND)): the draws of a chain are expanded in
 * order on one thread, chains concurrently.  BridgeStan writes every variable as a column-major block; the reference re-orders
 * them to C order (src/stan.rs:507-516, 671-711): out[j] = theta[perm[j]] (perm = NULL: keep BridgeStan's order).  The
 * function addresses are those of the loaded model library (bridgestan.h: bs_param_constrain, bs_rng_construct,
 * bs_rng_destruct, bs_free_error_msg).  A Stan error fails the copy with "Failed to constrain the parameters of the draw". */
int nphip_model_set_bridgestan_expand(nphip_model_t*, uint64_t expanded_dim, void* bs_model, void* param_constrain, void* rng_construct,
                                      void* rng_destruct, void* free_error_msg, const uint64_t* perm);
uint64_t nphip_model_expanded_dim(const nphip_model_t*);  /* 0 = no expand function */
uint64_t nphip_model_dim(const nphip_model_t*);
void nphip_model_free(nphip_model_t*);

/* ------------------------------------------------------------------ sampler */
typedef struct nphip_sampler nphip_sampler_t;

/* ChainProgress, wrapper.rs:47-104 */
typedef struct {
    uint64_t finished_draws;
    uint64_t total_draws;
    uint64_t divergences;
    int32_t tuning;
    int32_t started;
    uint64_t latest_num_steps;
    uint64_t total_num_steps;
    double step_size;
    uint64_t runtime_ms;
} nphip_chain_progress_t;

/* Engine/launch options beyond the reference's settings. */
typedef struct {
    int32_t device;            /* HIP device ordinal */
    int32_t waves_per_chain;   /* 0 = choose from dim; else 1,2,4,8,16 (fixes the reduction order) */
    uint64_t chain_offset;     /* global id of local chain 0 (multi-GPU chain sharding)            */
    uint64_t n_local_chains;   /* 0 = settings.num_chains                                          */
    void* stream;              /* hipStream_t to run on; NULL = engine-owned stream                 */
    int32_t store_draws;       /* keep [chain][draw][dim] positions in HBM (default 1)             */
    int32_t evals_per_launch;  /* fused models: leapfrogs per chain per kernel launch (0 = nphip_default_evals_per_launch(dim)) */
    int32_t start_paused;
    int32_t manual;            /* 1: no driver thread; the caller advances with nphip_sampler_step  */
    /* Device-callback models: caller-owned staging buffers q[n][dim], grad[n][dim], logp[n] (device,
     * fp64).  NULL = engine-owned.  Lets a PyTorch caller hand in tensors it allocated itself. */
    void* staging_q;
    void* staging_grad;
    void* staging_logp;
    int32_t no_register_kernel; /* 1: force the memory-resident kernel even where the register-resident
                                 * specialisation applies (A/B measurements, tests) */
    int32_t no_stream_cache;    /* 1: memory-resident fused kernels reload the cursor state every leapfrog (A/B, tests) */
    int32_t graph_steps;        /* device-callback models: >0 = capture this many (engine kernel + callback) steps in a HIP
                                 * graph and replay it; the callback must then only enqueue work on the given stream */
    int32_t host_groups;        /* host-callback models with zero-copy staging: 0 = default (two groups of chains in flight: the
                                 * kernel of one runs while the host evaluates the rows of the other), 1 = no pipelining,
                                 * 2..8 = that many groups */
    int32_t host_persist;       /* the same models, dim <= 4096, <= 1024 waves: evaluations one kernel launch serves.  0 = default
                                 * (256: the group's kernel stays on the device with the chain state in registers, publishes its
                                 * positions, waits for the host's word in pinned memory and goes on), N > 1 = that many,
                                 * 1 = one launch per evaluation; -N (tests) = leave the resident mode after N evaluations, the
                                 * way a failed roll call does.  The dense-precision Gaussian (nphip_model_dense_gaussian): 0 = its resident
                                 * form when the job fills the die-local clusters of workgroups that share a round's GEMM (otherwise a launch
                                 * per evaluation is faster: DESIGN.md section 12), 1 = a launch per evaluation, N > 1 = the resident form */
    int32_t reserved_;
} nphip_launch_t;

void nphip_launch_defaults(nphip_launch_t*);
/* leapfrogs per chain per launch a fused model of this dimension runs by default (about 10 ms of kernel; results do not depend on it) */
int nphip_default_evals_per_launch(uint64_t dim);
/* sizeof(nphip_launch_t) / sizeof(nphip_chain_progress_t) as the library was built: lets a binding check its struct layouts */
uint64_t nphip_abi_struct_size(int which /* 0: nphip_launch_t, 1: nphip_chain_progress_t */);

/* nuts_rs::Sampler::new — starts the driver thread; sampling begins immediately. */
nphip_sampler_t* nphip_sampler_create(const nphip_settings_t*, const nphip_model_t*, const nphip_launch_t*);
void nphip_sampler_free(nphip_sampler_t*);                 /* aborts if still running */
int nphip_sampler_wait(nphip_sampler_t*, int64_t timeout_ms); /* <0: no timeout; resumes a paused sampler */
/* Manual mode only: perform up to n_launches engine iterations (one kernel launch each; for callback
 * models one leapfrog per chain incl. the callback) on the calling thread and synchronise.  Stops early
 * when all chains are done.  kernel_ms (optional) receives the summed duration of the k_advance kernels
 * measured with HIP events on the engine's stream; launches_done (optional) the count performed.
 * Returns NPHIP_WAIT_DONE when every chain has finished, NPHIP_WAIT_TIMEOUT otherwise, NPHIP_WAIT_ERROR. */
int nphip_sampler_step(nphip_sampler_t*, uint64_t n_launches, double* kernel_ms, uint64_t* launches_done);
int nphip_sampler_pause(nphip_sampler_t*);
int nphip_sampler_resume(nphip_sampler_t*);
int nphip_sampler_abort(nphip_sampler_t*);                 /* stop; partial trace stays readable */
int nphip_sampler_is_finished(nphip_sampler_t*);
int nphip_sampler_progress(nphip_sampler_t*, uint64_t local_chain, nphip_chain_progress_t* out);
int nphip_sampler_waves_per_chain(const nphip_sampler_t*);
uint64_t nphip_sampler_num_chains(const nphip_sampler_t*); /* local */
uint64_t nphip_sampler_dim(const nphip_sampler_t*);
uint64_t nphip_sampler_total_draws(const nphip_sampler_t*); /* num_tune + num_draws */
/* wall-clock seconds spent in the sampling loop (excludes allocation) and kernel launches issued */
double nphip_sampler_seconds(const nphip_sampler_t*);
uint64_t nphip_sampler_launches(const nphip_sampler_t*);
/* How a host-callback model is being driven: not a host-callback model / one kernel launch per evaluation /
 * the same with groups of chains pipelined / resident launches / resident launches that fell back to launches per evaluation
 * because the device could not hold all chains at once. */
enum { NPHIP_HOST_MODE_NONE = 0, NPHIP_HOST_MODE_LAUNCH_PER_EVALUATION = 1, NPHIP_HOST_MODE_GROUPS = 2, NPHIP_HOST_MODE_RESIDENT = 3,
       NPHIP_HOST_MODE_FELL_BACK = 4 };
int nphip_sampler_host_mode(const nphip_sampler_t* s);

/* Trace hand-off (PyTrace / take_results, wrapper.rs:1431-1494).  All copies are D2H into host
 * memory laid out [local_chain][draw](...).  `finished[local_chain]` (optional) receives the number
 * of completed draws per chain (chains may be unequal after abort, python/nutpie/sample.py:192-199).
 * Names: "draws"(f64[dim]) "depth" "n_steps" "index_in_trajectory"(i64) "diverging" "maxdepth_reached"
 * "tuning"(u8) "energy" "energy_error" "logp" "step_size" "step_size_bar" "mean_tree_accept"
 * "mean_tree_accept_sym"(f64) and, when enabled, "gradient" "mass_matrix_inv" "divergence_start"
 * "divergence_end" "divergence_momentum" "divergence_start_gradient"(f64[dim]). */
int nphip_sampler_finished_draws(nphip_sampler_t*, uint64_t* finished);
int nphip_sampler_copy_stat(nphip_sampler_t*, const char* name, void* host_out, uint64_t nbytes);
/* The expand step over the whole stored trace (PyMcModelRef::expand_vector per draw, src/pymc.rs:217-286):
 * host_out[local_chain][draw][expanded_dim], fp64; rows of draws a chain has not finished are NaN.  Needs store_draws.
 * Also reachable as nphip_sampler_copy_stat(s, "expanded", ...). */
int nphip_sampler_copy_expanded(nphip_sampler_t*, void* host_out, uint64_t nbytes);
/* Chains stopped at a pause draw: mask[local_chain] = 1 (2 = finished or failed, 0 = running; mask may be NULL); returns the
 * number of waiting chains, or < 0 on error. */
int64_t nphip_sampler_waiting(nphip_sampler_t*, uint8_t* mask);
/* Resume waiting chains at new positions (positions[n][dim], host memory, or device memory if on_device): each chain
 * re-enters the initial-point sequence there — logp and gradient, mass matrix from the gradient, step-size search — and goes
 * on with its next draw; draw counter, trace and RNG streams continue.  What the host changed in between (e.g. the linear
 * map a wrapped model applies) is the host's business.  Manual-mode samplers only (the caller drives nphip_sampler_step). */
int nphip_sampler_resume_at(nphip_sampler_t*, uint64_t n, const uint64_t* local_chains, const double* positions, int on_device);
/* Low-rank metric (reference: the mass matrix of adaptation="low_rank", src/wrapper.rs:307-334, python/nutpie/sample.py:921-933,
 * docs/sampling-options.qmd:124-144):  M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2  with D = diag(sigma^2), V = k <= 16 orthonormal
 * columns, Lambda their eigenvalues.  With the boolean setting `low_rank_metric` the engine integrates, draws momenta and tests
 * U-turns under such a metric (every model flavour; fused models up to D = 4096 and compiled densities on the register-resident leaf).  The metric of a chain is supplied by the host at
 * the pause draws: sigma2[n][dim], V[n][k][dim] (row j = column j of V), lambda[n][k], host or device memory.  The chain keeps its
 * position, runs a step-size search under the new metric and goes on with its next draw; its own diagonal adaptation is off
 * from then on.  Until the first call a chain runs on the diagonal metric it adapts itself, exactly as without the setting.
 * Manual-mode samplers only.  Every named chain must be stopped at a pause draw (nphip_sampler_waiting): chains that are not keep
 * their metric and the call returns an error naming how many (the stopped ones among them have taken the new one).  The window
 * estimator that produces (sigma2, V, lambda): nphip_low_rank_estimate below (round 6; before that, and for shapes it does not cover,
 * above the C-ABI: nutpie_amd/low_rank.py::estimate).  The engine keeps V in SINGLE precision
 * (rounded to nearest on receipt; all arithmetic on it is fp64): the metric applied is that of the rounded columns. */
int nphip_sampler_set_metric(nphip_sampler_t*, uint64_t n, const uint64_t* local_chains, uint64_t k, const double* sigma2, const double* V,
                             const double* lambda, int on_device);
/* The same hand-in for chains that RUN (round 6): the metric is parked beside the chain (one staging row per chain; a parked metric that
 * has not been taken yet is replaced) and the chain takes it ITSELF at the end of the draw it is working on — same position, a step-size
 * search under the new metric, its own diagonal adaptation off from then on, exactly as after nphip_sampler_set_metric — so no chain
 * ever stops for the host and no pause draws are needed.  Called between two nphip_sampler_step calls of a manual-mode sampler created
 * with `low_rank_metric`.  Chains that have finished, failed, or are within one draw of the end of their warm-up ignore the hand-in;
 * *n_taken (optional) receives the number of chains that parked it.  What nuts-rs does inline on the chain's own thread at an update
 * of the mass matrix (src/wrapper.rs:307-334: mass_matrix_update_freq), with the estimate computed beside the running chains. */
int nphip_sampler_stage_metric(nphip_sampler_t*, uint64_t n, const uint64_t* local_chains, uint64_t k, const double* sigma2, const double* V,
                               const double* lambda, int on_device, uint64_t* n_taken);
/* Draws every local chain has finished (draws[local_chain], optional) and its state (state[local_chain], optional: 0 running,
 * 1 stopped at a pause draw, 2 finished or failed) in one read of the control blocks; returns the number of running chains, < 0 on error.
 * (ChainProgress::finished_draws of every chain, src/wrapper.rs:66-82, for a driver that decides per chain.) */
int64_t nphip_sampler_chain_draws(nphip_sampler_t*, int64_t* draws, uint8_t* state);
/* Chains stopped at a pause draw go on as if they had not stopped: same metric, same step size, their own mass-matrix adaptation as it
 * was (round 6).  What the low-rank driver does with a chain whose window shows no direction outside the eigenvalue cutoff and that has
 * never been handed a metric: the diagonal metric it adapts itself — every draw, on the device — IS the metric the estimator would hand
 * it, only fresher (the reference refreshes the diagonal part every mass_matrix_update_freq draws, src/wrapper.rs:198-240).  Every named
 * chain must be stopped (nphip_sampler_waiting); an error names how many were not. */
int nphip_sampler_release(nphip_sampler_t*, uint64_t n, const uint64_t* local_chains);
/* Manual-mode samplers: evaluations per chain of the launches that follow (fused models, compiled densities, the dense Gaussian's resident
 * form; 0 = the default).  A driver that looks at its chains between launches (low-rank hand-ins) takes short launches while chains can
 * still stop and long ones afterwards; what a chain computes does not depend on where its launches end. */
int nphip_sampler_set_evals_per_launch(nphip_sampler_t*, int32_t evals);
/* Batched symmetric eigendecomposition on the device — the dense-linear-algebra kernel of the low-rank estimator (reference:
 * adaptation="low_rank", src/wrapper.rs:307-334; nuts-rs uses faer's self-adjoint eigendecomposition there, Cargo.lock faer 0.24).
 * a_device: [n_batch][order][order] row-major, the LOWER triangle is the matrix; on return column j of matrix b is the eigenvector
 * of w_device[b][j], eigenvalues ascending.  order <= 128: one workgroup per matrix, the matrix resident in LDS (Householder
 * tridiagonalisation, Q in place, implicit QL — nutpie_amd/csrc/linalg.hip).  Runs on `stream` and returns after it has finished
 * (the convergence status of every matrix is checked).  Used by nutpie_amd/low_rank.py::estimate for its four decompositions. */
int nphip_batched_eigh(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream);

/* The window estimator of adaptation="low_rank" below the ABI (round 6; reference: the low-rank mass matrix of nuts-rs behind
 * PyNutsSettings::LowRank, src/wrapper.rs:307-334, 725-729; python/nutpie/sample.py:921-933; docs/sampling-options.qmd:124-144 — the crate
 * is not in the tree: the estimator follows the published description, as nutpie_amd/low_rank.py::estimate, which it restates, does).
 * For each of n chains: from the window of m draws and their gradients — draws[chain * chain_stride + t * draw_stride + i], t < m, i < dim,
 * device memory (the engine's own trace: nphip_sampler_device_ptr "draws" / "gradient" offset to the window's first draw), chain taken
 * from chains_device[block] (device memory) or the block index when NULL —
 *   sigma2[n][dim]      the diagonal scaling std(x) / std(g) over ALL m draws, re-centred on the median of the projected spectrum
 *   V[n][k_max][dim]    row j = column j of V (orthonormal; zero rows beyond k_used[n])
 *   lambda[n][k_max]    eigenvalues (1 beyond k_used)
 * of  M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2:  the geometric mean of the draw covariance and the inverse gradient covariance
 * (regularised by gamma) projected onto the span of the n_pick basis draws pick[0 .. n_pick) (indices into the window, host memory) and their
 * gradients, eigenvalues outside [1 / cutoff, cutoff] kept, at most k_max <= 16.  One workgroup per chain, LDS-resident, the four symmetric
 * eigenproblems of order 2 n_pick <= 64 by a one-sided Jacobi method (nutpie_amd/csrc/lowrank_est.hip).  scratch: n * 4112 doubles of
 * device memory.  max_workgroups (0 = one per chain): the launch's size — a workgroup wants a CU's LDS to itself, so beside a
 * running engine kernel the caller caps it at the CUs that kernel leaves idle (a workgroup then takes several chains in turn).
 * Asynchronous on `stream`.  nphip_low_rank_estimate_supported: whether a shape is inside what the kernel covers
 * (dim <= 512, n_pick <= 32 and <= dim, k_max <= 16). */
int nphip_low_rank_estimate_supported(uint64_t dim, uint64_t m, uint64_t n_pick, uint64_t k_max);
int nphip_low_rank_estimate(uint64_t n, uint64_t dim, uint64_t m, uint64_t n_pick, const int32_t* pick, const double* draws, const double* grads,
                            int64_t chain_stride, int64_t draw_stride, const int64_t* chains_device, double gamma, double cutoff, uint64_t k_max,
                            double* sigma2, double* V, double* lambda, int32_t* k_used, double* scratch, uint64_t max_workgroups, void* stream);

/* Developer aid: per-section cycle counters summed over chains; all zero unless the library was built
 * with -DNPHIP_PROFILE.  [0] leapfrog cycles [1] tree cycles (hot) [2] draw-end cycles [3..5] their counts. */
int nphip_sampler_profile(nphip_sampler_t*, int64_t out[16]);
/* Device pointer of a trace array (for zero-copy wrapping / RCCL gathers); NULL if absent. */
void* nphip_sampler_device_ptr(nphip_sampler_t*, const char* name);

/* ------------------------------------------------------------------ test hooks */
/* Evaluate the device implementations of include/nphip_spec.h on arrays (parity tests).
 * fn: 0 exp 1 log 2 log1p 3 sin2pi 4 cos2pi 5 sqrt 6 reciprocal 7 normals(seed=x[0],chain=x[1],draw=x[2],purpose=x[3]) */
int nphip_test_detmath(int device, int fn, uint64_t n, const double* x, double* y);
/* the stages of nphip_batched_eigh on their own.  mode 1: Householder tridiagonalisation only — w = the diagonal of T, row 0 of each a = its
 * sub-diagonal (entry i couples i and i + 1); mode 2: also Q, returned in a (T = Q' A Q); mode 3: the decomposition, with w[0..6] of every
 * matrix replaced by cycle counts (tridiagonalisation, Q, QL recurrence, QL application, rotations, sweeps, total) */
int nphip_test_eigh_stage(uint64_t n_batch, uint64_t order, double* a_device, double* w_device, void* stream, int mode);
/* the dense Gaussian's evaluation on n rows (host arrays): grad[n][dim] = -P (x - mu) by the engine's MFMA GEMM, logp[n] = 1/2 (x - mu).grad in
 * the summation order of `waves` waves per chain */
int nphip_test_dense_grad(int device, int waves, uint64_t n, uint64_t dim, const double* x, const double* mu, const double* P, double* grad, double* logp);
/* measured fp64 matrix-core rate of the device (v_mfma_f64_16x16x4_f64 back to back on every SIMD), in TFLOP/s: the peak the dense model's roofline is priced against */
int nphip_test_mfma_f64_rate(int device, double* tflops);
/* dot product in the engine's summation order with W waves */
int nphip_test_dot(int device, int waves, uint64_t n, const double* x, const double* y, double* out);
/* host-only: the evaluation pool of the host-callback path (spin-waiting workers, `use` threads per batch); row r of batch b
 * adds (b + 1) * (r + 1) into out[r]; *usable_cores = the core count the pool is sized against (affinity and cgroup quota) */
int nphip_test_rowpool(int threads, uint64_t rows, int batches, int use, uint64_t* out, int* usable_cores);

#ifdef __cplusplus
}
#endif
#endif
