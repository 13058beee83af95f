// engine_types.h — data layout shared by the HIP kernels and the host driver.
//
// Everything a chain owns lives in HBM in chain-major order ("[chain][slot][vector][ld]"),
// so one workgroup streams contiguous rows.  See DESIGN.md §3 for the layout rationale.
#pragma once

#include <stdint.h>

namespace nphip {

constexpr int kMaxDepthCap = 16;  // settings.maxdepth <= 16 (2^16 leapfrogs per draw)
constexpr int kLrMax = 16;        // columns of the low-rank part of the metric, at most

// ---- phases of the per-chain state machine (Ctl::phase) -------------------------------
enum Phase : int64_t {
    PH_START = 0,      // nothing evaluated yet: generate the first initial point
    PH_INIT_EVAL = 1,  // waiting for logp at an initial point
    PH_SS_FIRST = 2,   // waiting for the first leapfrog of the step-size search
    PH_SS_ITER = 3,    // waiting for an iteration leapfrog of the step-size search
    PH_TREE = 4,       // waiting for a leapfrog inside the NUTS tree
    PH_DONE = 5,
    PH_ERROR = 6,
    PH_WAIT_HOST = 7,  // stopped after a draw listed in DevSettings::pause_draws: the host re-parametrises the chain and resumes it
    PH_RESUME_SS = 8,  // resumed by the host with a new metric at the same position: step-size search, then the next draw
    // launch-per-evaluation (callback) kernels only — a launch lasts as long as its slowest chain, so the end of a draw is cut into
    // slices that take a launch each (the chain sits out the evaluations in between; its trace does not change):
    PH_DRAW_END = 9,    // the tree of the draw is complete (Ctl::pend_end says how): adaptation, trace row, statistics
    PH_DRAW_BEGIN = 10, // momentum refresh and the first leapfrog of the next draw
};

enum ChainError : int64_t {
    CE_NONE = 0,
    CE_INIT_FAILED = 1,  // no finite initial point after num_try_init attempts
    CE_FATAL_LOGP = 2,   // logp callback returned a negative code
    CE_RESUME_FAILED = 3,   // the position given to nphip_sampler_resume_at does not evaluate (logp or gradient not finite)
};

// ---- P-slots: (p, rho) pairs.  Assignment is a pure function of the leaf index ---------
// slot 0            : momentum at the trajectory origin
// slot 1..4         : trajectory ends, END[dirbit][parity]
// slot 5 + k        : FIRST[k], k = 0..cap   (first leaf of an open sub-tree of level k)
// slot 5+cap+1 + k  : LAST[k],  k = 0..cap   (last leaf of a completed sub-tree of level k)
constexpr int kSlotInit = 0;
__host__ __device__ inline int slot_end(int dirbit, int parity) { return 1 + 2 * dirbit + parity; }
__host__ __device__ inline int slot_first(int k) { return 5 + k; }
__host__ __device__ inline int slot_last(int k, int cap) { return 5 + cap + 1 + k; }
__host__ __device__ inline int num_pslots(int cap) { return 5 + 2 * (cap + 1); }
// Q-pool: (q, grad) pairs, allocated by bitmask.  ends(2) + candidate(1) + sub-tree draws(cap) + new leaf(1)
__host__ __device__ inline int num_qpool(int cap) { return cap + 4; }

// ---- per-chain control block: 8-byte words only (copied HBM <-> LDS word-wise) ----------
struct Ctl {
    int64_t phase;
    int64_t err;
    int64_t draw;          // draw in progress (== finished draws)
    int64_t init_attempt;
    int64_t total_steps;   // cumulative leapfrogs incl. step-size search
    int64_t n_div;         // cumulative divergent draws
    int64_t latest_steps;  // n_steps of the last finished draw
    int64_t eval_buf;      // Q-pool index whose q is (being) evaluated
    // hamiltonian
    double step_size;
    // dual averaging [oracle: struct DualAverage]
    double da_log_step, da_log_step_adapted, da_hbar, da_mu;   // Adam: hbar = first moment, mu = second moment
    double adam_b1t, adam_b2t;                                // Adam: running powers of beta1, beta2
    int64_t da_count;
    // adaptation schedule
    int64_t tuning;
    int64_t has_initial_mm;
    int64_t last_update;
    int64_t fg, fg_count, bg_count;  // fg = index (0/1) of the foreground estimator
    // step-size search
    int64_t ss_iter, ss_dir, ss_id;
    // pending leapfrog
    int64_t lf_srcq, lf_srcp, lf_newq, lf_newp, lf_sign;
    // trajectory / tree
    double H0;
    double main_wm;   // multinomial weight of the main tree = main_wm * 2^main_we (nphip_spec.h)
    int64_t main_we;
    int64_t depth, dir, nleaf;
    int64_t idx_left, idx_right, idx_cur;
    int64_t endq[2], endp[2], endpar[2];
    int64_t curq, curp;
    int64_t cand_q, cand_idx;
    double cand_U, cand_E;
    // acceptance collector (sums over the leapfrogs of the current draw; divided by n_steps when read)
    double acc_sum, acc_sym_sum;
    int64_t n_steps;
    // info of the draw being finished (kept across a mid-adapt step-size search)
    int64_t fin_depth, fin_flags;
    double fin_eerr;
    // lean register kernels: (A.first, T.first) of the level-1 merge the next leaf will check, evaluated one leaf early
    int64_t pre_turn;
    // PH_DRAW_END: how the draw ended — bit 0 diverging, bit 1 maxdepth reached, bit 2 store the divergence record, bit 3 ... with its end position
    int64_t pend_end;
    // low-rank metric (nphip_sampler_set_metric): 0 = the chain still runs on the diagonal metric it adapts itself; 1 = the host
    // supplied (sigma^2, V, lambda) — M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 with lr_k columns — and the chain's own
    // mass-matrix adaptation is off
    int64_t host_metric, lr_k;
    // a metric handed in while the chain RUNS (nphip_sampler_stage_metric): k + 1 of the metric waiting in Args::st_* — the chain
    // takes it between two draws of its warm-up (finish_draw) and goes on through a step-size search; 0 = nothing waits
    int64_t staged;
    // resident host-callback launches (k_advance<..., REMOTE>): the evaluation this chain publishes next, whether the host
    // has asked the launch to end at the next boundary, and the group it reports to (set at kernel start; transient)
    int64_t hs_seq, hs_last, hs_grp, hs_n, hs_wgn, hs_box;
    // cycle counters per section (only advanced when built with -DNPHIP_PROFILE): leapfrog, tree, rare, count
    int64_t prof[16];
    // sub-tree stack
    double sub_wm[kMaxDepthCap];
    int64_t sub_we[kMaxDepthCap];
    double sub_U[kMaxDepthCap];
    double sub_E[kMaxDepthCap];
    int64_t sub_q[kMaxDepthCap];
    int64_t sub_idx[kMaxDepthCap];
};
static_assert(sizeof(Ctl) % 8 == 0, "Ctl must be made of 8-byte words");
constexpr int kCtlWords = sizeof(Ctl) / 8;

// ---- settings as the kernels see them -------------------------------------------------
struct DevSettings {
    uint64_t seed;
    int64_t num_tune, num_draws;
    int64_t maxdepth, mindepth;
    int32_t check_turning, use_grad_based, adapt_mass_matrix, fixed_step_size;
    double max_energy_error;
    int64_t early_end, final_window;  // derived window bounds
    int64_t mm_switch_freq, early_mm_switch_freq, mm_update_freq;
    double initial_step, target_accept, jitter, max_step_size;
    double da_k, da_t0, da_gamma;
    double adam_lr;            // step_size_adapt_method = "adam": learning rate; adapt_adam selects it
    int32_t adapt_adam, pad1_;
    int32_t init_kind, num_try_init;
    int32_t store_draws, store_gradient, store_mass_matrix, store_divergences;
    int32_t low_rank_metric, pad3_;   // the host may replace a chain's metric at the pause draws (nphip_sampler_set_metric)
    // host-driven adaptation (low-rank metric as a linear re-parametrisation, nutpie_amd/low_rank.py): a chain stops
    // (PH_WAIT_HOST) when it has finished exactly pause_draws[i] draws
    int32_t n_pause, pad2_;
    int64_t pause_draws[16];
};

// ---- kernel arguments -------------------------------------------------------------------
struct Args {
    DevSettings s;
    int64_t n_chains;      // local
    int64_t chain_offset;  // global id of local chain 0
    int64_t dim, ld;       // ld = dim rounded up to 128
    int32_t cap;           // maxdepth (slot/pool sizing)
    int32_t npslots, nqpool;
    // per-chain state
    Ctl* ctl;
    double* qpool;   // [n][nqpool][2][ld]   (q, grad)
    double* pslots;  // [n][npslots][2][ld]  (p, rho)
    double* sig2;    // [n][ld]
    double* est;     // [n][2][4][ld]        (mean_q, m2_q, mean_g, m2_g) x {est0, est1}
    // model: fused tridiagonal Gaussian
    const double* m_mu;   // [ld]
    const double* m_a;    // [ld]
    const double* m_b;    // [ld]
    const double* m_bsh;  // [ld + 8]: m_bsh[i] = b_{i-1} (m_bsh[0] and the tail are -0.0): aligned (b_{i-1}, b_i) pairs
    // model: callbacks (dense staging, ld = dim)
    double* qeval;   // [n][dim]
    double* geval;   // [n][dim]
    double* ueval;   // [n]   logp values
    int64_t* ecode;  // [n]   per-chain callback codes (host callback) or NULL
    const double* init_points;  // [n][dim] local slice, or NULL
    // trace
    double* tr_draws;  // [n][T][dim] or NULL
    double* tr_grad;   // [n][T][dim] or NULL
    double* tr_mm;     // [n][T][dim] or NULL
    double* tr_div[4]; // start, end, momentum, start_gradient: [n][T][dim] or NULL
    int64_t* st_depth; int64_t* st_nsteps; int64_t* st_idx;
    uint8_t* st_diverging; uint8_t* st_maxdepth; uint8_t* st_tuning;
    double* st_energy; double* st_energy_error; double* st_logp; double* st_step; double* st_step_bar;
    double* st_accept; double* st_accept_sym;
    // launch control
    int32_t reg_nv;       // >0: register-resident kernel, NV = reg_nv chunks of 128 per wave (fused; W = 1, or W = 2/4 with ld = 128 * W * NV)
    int32_t stream_cache; // 1: memory-resident fused kernel with the cursor's loads cached in VGPRs (NV < 0 instantiations)
    int32_t sig_lds;      // 1: memory-resident kernels with W >= 8 keep the chain's sigma^2 in (dynamic) LDS
    int32_t lean;         // 1: lean register-resident kernel (W = 8, reg_nv chunks per wave, sigma^2 in dynamic LDS)
    int32_t max_evals;    // fused: evaluations per chain this launch
    int32_t have_result;  // callbacks: geval/ueval hold the answer to the pending request
    unsigned long long* counters;  // [0] chains done, [1] chains in error, [2] chains that entered PH_WAIT_HOST (cumulative)
    // pipelined host-callback groups (host.hip: iteration_pipelined)
    unsigned int* grp_arrive;               // device [groups]: chains of the group that finished the current launch
    volatile unsigned long long* grp_flag;  // pinned host [groups][4]: evaluation sequence number, chains done, chains in error,
                                            // sequence number of a launch whose roll call failed
    // resident launches: the kernel stays on the device between evaluations and waits for the host's word
    volatile unsigned long long* grp_go;    // pinned host [groups][8]: sequence number whose results are ready | kGoLast
    unsigned long long* grp_go_dev;         // device [groups][16] (a cache line each): [0] the same word, republished by the
                                            // group's last arriver; of group 0 also [1] roll-call verdict, [2] roll-call count
    // runtime-compiled device densities (nphip_model_jit_density): the density is a device function compiled into its own
    // instantiation of k_advance; an evaluation is a call in the middle of the register-resident leaf (kernels.hip: density_eval)
    // low-rank metric (settings.low_rank_metric: memory-resident kernels only): P-slots carry a third vector, the velocity
    // v = M^-1 p of the state, which the U-turn criteria read instead of recomputing sigma^2 p
    int32_t pvec;          // vectors per P-slot: 2 (p, rho) or 3 (p, rho, v)
    int32_t lr_on;         // the job may receive host metrics: lr_V / lr_lam / lr_std are allocated
    float* lr_V;           // [n][kLrMax][ld]  orthonormal columns (rows here), zero beyond dim and beyond lr_k — SINGLE precision in memory (round 5):
                           // the columns' traffic bounds the low-rank leapfrog (four passes per step), every operation on them is fp64 on the
                           // values fp32 holds; the metric that is applied is the one of the rounded columns (include/nphip_spec.h)
    double* lr_lam;        // [n][kLrMax]      eigenvalues
    double* lr_std;        // [n][ld]          sqrt(sigma^2)
    // metrics handed in while the chains run (Ctl::staged): the chain copies its rows into (sigma^2, lr_std, lr_V, lr_lam) itself,
    // between two draws — no chain ever stops for the host
    double* st_sig2;       // [n][ld]
    float* st_V;           // [n][kLrMax][ld]
    double* st_lam;        // [n][kLrMax]
    const void* dens_data;       // the model's data block (device memory; layout defined by the generated prelude of the density source)
    int32_t dens_lds_doubles;    // LDS scratch per wave the density asked for, in doubles (dynamic LDS of the launch)
    int32_t dens_shared_doubles; // LDS shared by the chains of a workgroup (the model's data staged once per launch: nphip_density_stage)
    // dense-precision Gaussian, resident form (kernels.hip: Machine<..., DG>): the padded precision matrix [DP][KP] (DP = dim rounded up to 64,
    // KP to 16; zeros beyond dim), the padded mean [KP], and the rendezvous words of the clusters of workgroups — per cluster two counters
    // (positions written / gradients written) on a 128-byte line each, zeroed by the host before every launch; word 0 of the last line:
    // a rendezvous timed out (everybody leaves)
    const double* dg_P;
    const double* dg_mu;
    int64_t dg_KP;
    unsigned long long* dg_sync;   // kDgSyncWords 64-bit words (layout: kernels.hip, kDg*)
    int32_t dg_variant, dg_pad_;   // measurement switches (NPHIP_DG_VARIANT): bit 0 no fences, bit 1 agent-scope accesses to the exchanged rows, bit 2 no GEMM
};
constexpr int kDgSyncWords = 128 * 32 + 16 + 16 + 8 * 256 / 2;   // = kDgWords of kernels.hip: 128 clusters x 2 lines, abort line, seat counts, seat table
constexpr unsigned long long kGoLast = 1ull << 32;      // finish this evaluation's step, then leave the kernel at the boundary
constexpr unsigned long long kGoSeqMask = 0xffffffffull;
constexpr unsigned long long kPubAllDone = 1ull << 62, kPubError = 1ull << 63;   // resident launches: flags beside the published sequence number
constexpr unsigned long long kRollGo = 1, kRollFail = 2;   // verdict = (launch sequence number << 8) | state

// the chains one launch covers (kernel parameter)
struct LaunchSlice {
    int chain_lo, chain_n;
    int grp;       // >= 0: publish completion in Args::grp_flag[grp] (no stream synchronisation on the host)
    unsigned seq;
    int materialise;   // 1 (callback kernels): the previous launch was a resident one, which defers the first half of a tree
                       // leapfrog into the leaf — perform it now so that this launch finds an evaluation pending
    // resident launches (REMOTE): ONE launch covers every group (streams may share a hardware queue, and a kernel queued behind
    // a resident one would never start); seq is then the launch's id for the roll call
    int n_grp;
    int grp_lo[9];         // group g = chains [grp_lo[g], grp_lo[g + 1])
    unsigned grp_seq[8];   // sequence number of each group's first evaluation in this launch
    // runtime-compiled densities, one wave per chain (set by nphip_jit_launch, read by its kernel only): chains per workgroup.  4 fills
    // every SIMD of a CU; a job with no more chains than the device has CUs gets a CU per chain instead (the chains of a workgroup share
    // its LDS bandwidth, and a density moves ~100 KB through it per evaluation)
    int cpb;
};

}  // namespace nphip
