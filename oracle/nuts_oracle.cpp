/*
 * nuts_oracle.cpp — CPU restatement of nutpie's diag-NUTS hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see nuts_oracle.h: "PARITY UNPINNED").
 *
 * What is restated and where it comes from.  nutpie itself only configures and
 * drives the sampler (reference src/wrapper.rs:957-1095 `PySampler::new` ->
 * `nuts_rs::Sampler::new`); the arithmetic is nuts-rs 0.18.3 (Cargo.lock:2295).
 * Each block below names the nuts-rs routine it restates (SURVEY.md Appendix A
 * section in brackets) and the in-tree file:line that corroborates it.
 *
 * The structure deliberately follows the crate (recursive tree with
 * reference-counted states), NOT the GPU engine (iterative, slot-indexed), so
 * that bit-parity between the two is a meaningful check.
 *
 * Build: see oracle/Makefile  (g++ -O3 -std=c++17 -ffp-contract=off -mfma).
 */
#include "nuts_oracle.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;
std::mutex g_error_mutex;
std::string g_last_error;

// ---------------------------------------------------------------------------
// Deterministic numerics — independent restatement of include/nphip_spec.h
// (table-driven; the header is the unrolled form).
// ---------------------------------------------------------------------------

struct U4 { uint32_t v[4]; };

U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    const uint64_t M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32)};
    uint32_t c[4] = {c0, c1, c2, c3};
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = M0 * c[0];
        uint64_t p1 = M1 * c[2];
        uint32_t n[4];
        n[0] = (uint32_t)(p1 >> 32) ^ c[1] ^ key[0];
        n[1] = (uint32_t)p1;
        n[2] = (uint32_t)(p0 >> 32) ^ c[3] ^ key[1];
        n[3] = (uint32_t)p0;
        memcpy(c, n, sizeof(n));
        key[0] += 0x9E3779B9u;
        key[1] += 0xBB67AE85u;
    }
    U4 o;
    memcpy(o.v, c, sizeof(c));
    return o;
}

double u01(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * std::ldexp(1.0, -53);
}

const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33, INV_LN2 = 0x1.71547652b82fep+0;

static void det_exp_parts(double x, double* p_out, double* k_out);

// Extended-range tree weights w = m * 2^e (restates the "extended-range tree weights" block of nphip_spec.h):
// the multinomial weights nuts-rs carries as log_size [A.3], without leaving the linear domain.
struct Weight { double m = 1.0; int64_t e = 0; };
Weight w_leaf(double neg_energy_error) {
    double x = std::min(std::max(neg_energy_error, -1e9), 1e9);
    Weight w; double k;
    det_exp_parts(x, &w.m, &k);
    w.e = (int64_t)k;
    return w;
}
double w_rel(const Weight& w, int64_t E) {  // w / 2^E, E >= w.e
    const int64_t d = E - w.e;
    return d >= 1000 ? 0.0 : w.m * std::ldexp(1.0, (int)-d);
}
Weight w_add(const Weight& a, const Weight& b) {
    Weight o;
    o.e = std::max(a.e, b.e);
    o.m = w_rel(a, o.e) + w_rel(b, o.e);
    return o;
}

static void det_exp_parts(double x, double* p_out, double* k_out) {
    static const double inv_fact[14] = {1.0, 1.0, 0.5, 0.16666666666666666, 0.041666666666666664,
                                        0.008333333333333333, 0.001388888888888889, 0.0001984126984126984,
                                        2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07,
                                        2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10};
    double k = std::nearbyint(x * INV_LN2);
    double r = std::fma(-k, LN2_HI, x);
    r = std::fma(-k, LN2_LO, r);
    double p = inv_fact[13];
    for (int n = 12; n >= 0; --n) p = std::fma(p, r, inv_fact[n]);
    *p_out = p; *k_out = k;
}

double det_exp(double x) {
    if (std::isnan(x)) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    static const double inv_fact[14] = {1.0, 1.0, 0.5, 0.16666666666666666, 0.041666666666666664,
                                        0.008333333333333333, 0.001388888888888889, 0.0001984126984126984,
                                        2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07,
                                        2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10};
    double k = std::nearbyint(x * INV_LN2);
    double r = std::fma(-k, LN2_HI, x);
    r = std::fma(-k, LN2_LO, r);
    double p = inv_fact[13];
    for (int n = 12; n >= 0; --n) p = std::fma(p, r, inv_fact[n]);
    int ki = (int)k, k1 = ki / 2, k2 = ki - k1;
    return (p * std::ldexp(1.0, k1)) * std::ldexp(1.0, k2);
}

double det_log(double x) {
    if (std::isnan(x)) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (std::isinf(x)) return x;
    int e = 0;
    if (x < std::ldexp(1.0, -1022)) { x *= std::ldexp(1.0, 54); e = -54; }
    int ex;
    double m = std::frexp(x, &ex);  // m in [0.5,1)
    m *= 2.0; ex -= 1;              // m in [1,2)
    e += ex;
    if (m >= 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = 2.0 / 25.0;
    for (int n = 11; n >= 1; --n) R = std::fma(R, z, 2.0 / (double)(2 * n + 1));
    double logm = std::fma(s * z, R, 2.0 * s);
    double de = (double)e;
    return std::fma(de, LN2_HI, std::fma(de, LN2_LO, logm));
}

double det_log1p(double y) {
    double u = 1.0 + y;
    if (u == 1.0) return y;
    return det_log(u) * y / (u - 1.0);
}

// nuts-rs `logaddexp` helper used for tree weights [A.3].
double det_logaddexp(double a, double b) {
    if (a == b) return a + 0x1.62e42fefa39efp-1;
    double diff = a - b;
    if (diff > 0.0) return a + det_log1p(det_exp(-diff));
    if (diff < 0.0) return b + det_log1p(det_exp(diff));
    return diff;
}

void det_sincos2pi(double u, double* sn, double* cs) {
    double t = 4.0 * u;
    double qd = std::floor(t);
    double f = t - qd;
    bool swap = f > 0.5;
    if (swap) f = 1.0 - f;
    double x = f * 0x1.921fb54442d18p+0;
    double x2 = x * x;
    static const double sc[8] = {-0.16666666666666666, 0.008333333333333333, -0.0001984126984126984,
                                 2.7557319223985893e-06, -2.505210838544172e-08, 1.6059043836821613e-10,
                                 -7.647163731819816e-13, 2.8114572543455206e-15};
    static const double cc[9] = {-0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05,
                                 -2.755731922398589e-07, 2.08767569878681e-09, -1.1470745597729725e-11,
                                 4.779477332387385e-14, -1.5619206968586225e-16};
    double S = sc[7];
    for (int n = 6; n >= 0; --n) S = std::fma(S, x2, sc[n]);
    double s = std::fma(x * x2, S, x);
    double C = cc[8];
    for (int n = 7; n >= 0; --n) C = std::fma(C, x2, cc[n]);
    double c = std::fma(x2, C, 1.0);
    if (swap) std::swap(s, c);
    switch ((int)qd) {
        case 0: *sn = s; *cs = c; break;
        case 1: *sn = c; *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}

void normal_pair(const U4& r, double* z0, double* z1) {
    double u1 = u01(r.v[0], r.v[1]);
    double u2 = u01(r.v[2], r.v[3]);
    double rad = std::sqrt(-2.0 * det_log(u1));
    double sn, cs;
    det_sincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

enum { RNG_MOMENTUM = 1, RNG_DIRECTION = 2, RNG_MERGE = 3, RNG_INIT = 4, RNG_SS_MOMENTUM = 5, RNG_JITTER = 6 };

// --- reductions in the engine's summation order (nphip_spec.h "geometry") ----
// 128*W interleaved accumulators, component add, DPP-order butterfly, wave-order sum.
struct Geometry { int W; };

template <class F>
double det_reduce(size_t n, Geometry geo, F term_fma /* (i, acc) -> fma(x_i, y_i, acc) */) {
    const int W = geo.W;  // 1..16
    double acc[128 * 16];
    const size_t P = (size_t)128 * W;
    for (size_t s = 0; s < P; ++s) acc[s] = 0.0;
    // element i accumulates into slot i % P  (= 128*((i/128) % W) + i%128), increasing i
    for (size_t base = 0; base < n; base += P) {
        const size_t m = (n - base < P) ? n - base : P;
        for (size_t s = 0; s < m; ++s) acc[s] = term_fma(base + s, acc[s]);
    }
    double total = 0.0;
    for (int w = 0; w < W; ++w) {
        double lane[64];
        for (int l = 0; l < 64; ++l) lane[l] = acc[(size_t)128 * w + 2 * l] + acc[(size_t)128 * w + 2 * l + 1];
        // stage partners of the engine's DPP reduction: l^1, l^2, mirror in 8, mirror in 16, l^16, lanes {0,32}
        for (int stage = 0; stage < 5; ++stage) {
            double nxt[64];
            for (int l = 0; l < 64; ++l) {
                int partner;
                switch (stage) {
                    case 0: partner = l ^ 1; break;
                    case 1: partner = l ^ 2; break;
                    case 2: partner = (l & ~7) | (7 - (l & 7)); break;
                    case 3: partner = (l & ~15) | (15 - (l & 15)); break;
                    default: partner = l ^ 16; break;
                }
                nxt[l] = lane[l] + lane[partner];
            }
            memcpy(lane, nxt, sizeof(lane));
        }
        const double wsum = lane[0] + lane[32];
        total = (w == 0) ? wsum : total + wsum;
    }
    return total;
}

double det_dot(const double* x, const double* y, size_t n, Geometry geo) {
#ifdef ORACLE_TUNED
    // "tuned" build (libnuts_oracle_tuned.so, bench.py's second cpu_baseline): free summation order, vectorised
    (void)geo;
    double acc = 0.0;
#pragma omp simd reduction(+ : acc)
    for (size_t i = 0; i < n; ++i) acc += x[i] * y[i];
    return acc;
#else
    return det_reduce(n, geo, [&](size_t i, double a) { return std::fma(x[i], y[i], a); });
#endif
}

// ---------------------------------------------------------------------------
// Models
// ---------------------------------------------------------------------------

struct Model {
    size_t dim = 0;
    virtual ~Model() = default;
    // returns the callback code convention of src/pymc.rs:166-180
    virtual int64_t logp(const double* q, double* grad, double* logp_out) = 0;
};

// Fused analytic model of the engine (DESIGN.md "tridiag Gaussian"):
//   z = q - mu ; t_i = a_i z_i (+ b_{i-1} z_{i-1}) (+ b_i z_{i+1}) ; g_i = -t_i ; logp = 0.5 * dot(z, g)
struct TridiagModel : Model {
    std::vector<double> mu, a, b;
    Geometry geo{1};
    std::vector<double> z;
    int64_t logp(const double* q, double* grad, double* logp_out) override {
        z.resize(dim);
#ifdef ORACLE_TUNED
        {   // one vectorised pass: gradient and logp together
            const size_t n = dim;
            double* zz = z.data();
            const double *mm = mu.data(), *aa = a.data(), *bb = b.data();
#pragma omp simd
            for (size_t i = 0; i < n; ++i) zz[i] = q[i] - mm[i];
            double acc = 0.0;
            if (n == 1) { grad[0] = -(aa[0] * zz[0]); *logp_out = 0.5 * zz[0] * grad[0]; return 0; }
            grad[0] = -(aa[0] * zz[0] + bb[0] * zz[1]);
            acc = zz[0] * grad[0];
#pragma omp simd reduction(+ : acc)
            for (size_t i = 1; i < n - 1; ++i) {
                const double t = aa[i] * zz[i] + bb[i - 1] * zz[i - 1] + bb[i] * zz[i + 1];
                grad[i] = -t;
                acc -= zz[i] * t;
            }
            grad[n - 1] = -(aa[n - 1] * zz[n - 1] + bb[n - 2] * zz[n - 2]);
            acc += zz[n - 1] * grad[n - 1];
            *logp_out = 0.5 * acc;
            return 0;
        }
#endif
        for (size_t i = 0; i < dim; ++i) z[i] = q[i] - mu[i];
        for (size_t i = 0; i < dim; ++i) {
            double t = a[i] * z[i];
            if (i > 0) t = std::fma(b[i - 1], z[i - 1], t);
            if (i + 1 < dim) t = std::fma(b[i], z[i + 1], t);
            grad[i] = -t;
        }
        *logp_out = 0.5 * det_dot(z.data(), grad, dim, geo);
        return 0;
    }
};

// Dense-precision Gaussian of the engine (nphip_model_dense_gaussian; nutpie_amd/csrc/dense_tile.h):
//   z = q - mu ; acc_j = sum_k z_k P[j][k] by fused multiply-adds, ONE accumulator per j from +0.0, in the order
//   k = k0 + 4 t + s  for k0 = 0, 16, ... / s = 0..3 / t = 0..3 (k >= dim skipped) ; g_j = -acc_j ; logp = 0.5 * dot(z, g)
// (the order in which the engine's fp64 matrix-core tile consumes a row: include/nphip_spec.h "dense gradient").
void dense_grad(const double* P, size_t dim, const double* z, double* grad) {
#ifdef ORACLE_TUNED
    for (size_t j = 0; j < dim; ++j) {
        const double* row = P + j * dim;
        double acc = 0.0;
#pragma omp simd reduction(+ : acc)
        for (size_t k = 0; k < dim; ++k) acc += z[k] * row[k];
        grad[j] = -acc;
    }
#else
    // eight rows at a time: eight independent accumulator chains (each chain is sequential by contract)
    for (size_t j0 = 0; j0 < dim; j0 += 8) {
        const size_t nj = dim - j0 < 8 ? dim - j0 : 8;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (size_t k0 = 0; k0 < dim; k0 += 16)
            for (size_t s = 0; s < 4; ++s)
                for (size_t t = 0; t < 4; ++t) {
                    const size_t k = k0 + 4 * t + s;
                    if (k >= dim) continue;
                    const double zk = z[k];
                    for (size_t u = 0; u < nj; ++u) acc[u] = std::fma(zk, P[(j0 + u) * dim + k], acc[u]);
                }
        for (size_t u = 0; u < nj; ++u) grad[j0 + u] = -acc[u];
    }
#endif
}

struct DenseModel : Model {
    const double* P = nullptr;   // [dim][dim] row-major, symmetric (borrowed)
    std::vector<double> mu, z;
    Geometry geo{1};
    int64_t logp(const double* q, double* grad, double* logp_out) override {
        z.resize(dim);
        for (size_t i = 0; i < dim; ++i) z[i] = q[i] - mu[i];
        dense_grad(P, dim, z.data(), grad);
        *logp_out = 0.5 * det_dot(z.data(), grad, dim, geo);
        return 0;
    }
};

struct CallbackModel : Model {
    oracle_logp_fn fn = nullptr;
    void* user = nullptr;
    int64_t logp(const double* q, double* grad, double* logp_out) override {
        return fn((uint64_t)dim, q, grad, logp_out, user);
    }
};

// ---------------------------------------------------------------------------
// Hamiltonian state  [A.1]
// ---------------------------------------------------------------------------

struct State {
    std::vector<double> q, p, g, v, psum;
    double U = 0.0, K = 0.0, H0 = 0.0;
    int64_t idx = 0;
    double energy() const { return K + U; }
    double energy_error() const { return energy() - H0; }
};

struct StatePool {
    size_t dim;
    std::vector<State*> free_list;
    explicit StatePool(size_t d) : dim(d) {}
    ~StatePool() { for (auto* s : free_list) delete s; }
    std::shared_ptr<State> get() {
        State* s;
        if (free_list.empty()) {
            s = new State();
            s->q.resize(dim); s->p.resize(dim); s->g.resize(dim); s->v.resize(dim); s->psum.resize(dim);
        } else {
            s = free_list.back();
            free_list.pop_back();
        }
        return std::shared_ptr<State>(s, [this](State* x) { free_list.push_back(x); });
    }
};
using StateP = std::shared_ptr<State>;

// What nuts-rs reports about a divergent transition: the state the failed leapfrog started from and — when the
// divergence is an energy error, not a logp error — the position it ended at.  These feed the optional statistics
// `divergence_start / _end / _momentum / _start_gradient` (python/nutpie/sample.py:631-650, docs/sample-stats.qmd:95-102).
struct DivergenceInfo {
    bool logp_error = false; double energy_error = NAN;
    bool has_start = false, has_end = false;
    std::vector<double> start_q, start_p, start_g, end_q;
    void record_start(const State& s);
};

// AcceptanceRateCollector [A.6]: incremental running means over all leapfrogs of a draw.
struct RunningMean {
    double sum = 0.0; uint64_t count = 0;
    void reset() { sum = 0.0; count = 0; }
    void add(double v) { count += 1; sum += (v - sum) / (double)count; }
};
// Acceptance-statistic means are kept as (sum, count) and divided when read (nuts-rs updates a running mean per
// leapfrog; same value up to rounding, two divisions fewer per leapfrog on the device).
// crate_arithmetic = 1 switches to the crate's own update: an incrementally updated running mean per leapfrog.
struct SumMean {
    uint64_t count = 0; double total = 0.0; bool running = false;
    void reset() { count = 0; total = 0.0; }
    void add(double v) {
        count += 1;
        if (running) total += (v - total) / (double)count;   // total IS the mean in this mode (nuts-rs RunningMean::add)
        else total += v;
    }
    double value() const { return running ? total : (count ? total / (double)count : 0.0); }
};
// ---- experiment knobs (oracle_set_variant): which of the RECALLED details of nuts-rs' warm-up the oracle follows.  The defaults are
// the restatement every parity test runs; the other values exist for the sensitivity study of the reference-held evidence
// (scratch/r5_reference_sensitivity.py, profiles/r5_reference_sensitivity.txt) and are never set by a test of the engine.
struct Variant {
    int min_refresh = 3;     // draws the foreground estimator needs before a mass-matrix refresh reads it
    int search_mode = 2;     // initial step-size search: 0 never, 1 at the initial point only, 2 also after the first data-driven mass matrix
    int late_sym = 1;        // the symmetric acceptance statistic drives dual averaging once no estimator switch can follow (and in the final window)
    int last_bar = 1;        // the last tuning draw sets step_size = step_size_bar
    int floor_windows = 0;   // window bounds as (frac * T) truncated instead of ceil
};
static Variant g_variant;

struct Collector {
    SumMean mean, mean_sym;
    void set_running(bool r) { mean.running = r; mean_sym.running = r; }
    void register_init() { mean.reset(); mean_sym.reset(); }
    void register_leapfrog(const State* end, bool diverged) {
        if (diverged) { mean.add(0.0); mean_sym.add(0.0); return; }
        double e = det_exp(-end->energy_error());
        double a = e < 1.0 ? e : 1.0;
        mean.add(a);
        mean_sym.add(2.0 * a / (1.0 + e));
    }
};

void DivergenceInfo::record_start(const State& s) { has_start = true; has_end = false; start_q = s.q; start_p = s.p; start_g = s.g; }

enum class Leap { Ok, Diverge, Fatal };

struct Hamiltonian {
    Model* model;
    Geometry geo;
    std::vector<double> sig2;  // diagonal of M^-1 ("variance"), nuts-rs DiagMassMatrix
    double step_size = 0.1;
    double max_energy_error = 1000.0;
    StatePool pool;
    Hamiltonian(Model* m, Geometry g) : model(m), geo(g), sig2(m->dim, 1.0), pool(m->dim) {}
    size_t dim() const { return model->dim; }

    // Low-rank metric M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 (nuts-rs LowRankMassMatrix; reference knobs src/wrapper.rs:307-334).
    // Inactive (lr == false): the diagonal metric, v = sig2 * p, exactly as before.
    bool lr = false;
    int lr_k = 0;
    std::vector<double> lr_std, lr_V, lr_lam;   // sqrt(sig2) [n]; k rows of length n; k eigenvalues
    void set_metric(int k, const double* s2, const double* V, const double* lam) {
        const size_t n = dim();
        lr = true; lr_k = k;
        sig2.assign(s2, s2 + n);
        lr_std.resize(n);
        for (size_t i = 0; i < n; ++i) lr_std[i] = std::sqrt(sig2[i]);
        lr_V.assign(V ? V : s2, (V ? V : s2) + (size_t)k * n);
        lr_lam.assign(lam ? lam : s2, (lam ? lam : s2) + k);
    }
    // v = M^-1 p:  u = std p ; d_j = <V_j, u> ; c_j = (lambda_j - 1) d_j ; w = u + sum_j V_j c_j (fma chain, j ascending) ; v = std w
    void velocity(const double* p, double* v) const {
        const size_t n = dim();
        if (!lr) { for (size_t i = 0; i < n; ++i) v[i] = sig2[i] * p[i]; return; }
        std::vector<double> u(n), c(lr_k);
        for (size_t i = 0; i < n; ++i) u[i] = lr_std[i] * p[i];
        for (int j = 0; j < lr_k; ++j) c[j] = (lr_lam[j] - 1.0) * det_dot(lr_V.data() + (size_t)j * n, u.data(), n, geo);
        for (size_t i = 0; i < n; ++i) {
            double w = u[i];
            for (int j = 0; j < lr_k; ++j) w = std::fma(lr_V[(size_t)j * n + i], c[j], w);
            v[i] = lr_std[i] * w;
        }
    }

    // EuclideanHamiltonian::leapfrog [A.5]; operand list corroborated by
    // reference benches/run_tvm_leapfrog.rs_old:81-85 (position, momentum, grad, epsilon, mass_diag).
    Leap leapfrog(const State& s, int sign, Collector* col, StateP* out_state, DivergenceInfo* info) {
        const size_t n = dim();
        StatePool& pl = pool;
        StateP o = pl.get();
        const double eps = (double)sign * step_size;
        const double h = 0.5 * eps;
#ifdef ORACLE_TUNED
        {
            double *op = o->p.data(), *oq = o->q.data();
            const double *sg = s.g.data(), *sp = s.p.data(), *sq = s.q.data(), *s2 = sig2.data();
#pragma omp simd
            for (size_t i = 0; i < n; ++i) {
                const double ph = sp[i] + h * sg[i];
                op[i] = ph;
                oq[i] = sq[i] + eps * (s2[i] * ph);
            }
        }
#else
        for (size_t i = 0; i < n; ++i) o->p[i] = std::fma(h, s.g[i], s.p[i]);
        velocity(o->p.data(), o->v.data());
        for (size_t i = 0; i < n; ++i) o->q[i] = std::fma(eps, o->v[i], s.q[i]);
#endif
        double lp = 0.0;
        int64_t code = model->logp(o->q.data(), o->g.data(), &lp);
        o->idx = s.idx + sign;
        o->H0 = s.H0;
        if (code < 0) { g_error = "logp callback returned fatal code " + std::to_string(code); return Leap::Fatal; }
        if (code > 0 || !std::isfinite(lp)) {
            // recoverable logp error => divergence (src/pymc.rs:166-180, src/stan.rs:392-396,459-461,
            // src/pyfunc.rs:100-116,218-220)
            info->logp_error = true;
            info->record_start(s);
            if (col) col->register_leapfrog(o.get(), true);
            return Leap::Diverge;
        }
        o->U = -lp;
#ifdef ORACLE_TUNED
        {
            double *op = o->p.data(), *ov = o->v.data(), *ops = o->psum.data();
            const double *og = o->g.data(), *s2 = sig2.data(), *sps = s.psum.data();
            const double keep = (o->idx == -1) ? 0.0 : 1.0;
            double kk = 0.0;
#pragma omp simd reduction(+ : kk)
            for (size_t i = 0; i < n; ++i) {
                const double pv = op[i] + h * og[i];
                const double vv = s2[i] * pv;
                op[i] = pv;
                ov[i] = vv;
                kk += pv * vv;
                ops[i] = keep * sps[i] + pv;
            }
            o->K = 0.5 * kk;
        }
#else
        for (size_t i = 0; i < n; ++i) o->p[i] = std::fma(h, o->g[i], o->p[i]);
        velocity(o->p.data(), o->v.data());
        o->K = 0.5 * det_dot(o->p.data(), o->v.data(), n, geo);
        if (o->idx == -1) {
            o->psum = o->p;
        } else {
            for (size_t i = 0; i < n; ++i) o->psum[i] = s.psum[i] + o->p[i];
        }
#endif
        double de = o->energy_error();
        if (de > max_energy_error || !std::isfinite(de)) {
            info->energy_error = de;
            info->record_start(s);
            info->has_end = true;
            info->end_q = o->q;
            if (col) col->register_leapfrog(o.get(), true);
            return Leap::Diverge;
        }
        if (col) col->register_leapfrog(o.get(), false);
        *out_state = o;
        return Leap::Ok;
    }

    // EuclideanHamiltonian::is_turning [A.4]
    bool is_turning(const State& s1, const State& s2) {
        const State* start = &s1; const State* end = &s2;
        if (!(s1.idx < s2.idx)) { start = &s2; end = &s1; }
        const int64_t a = start->idx, b = end->idx;
        const size_t n = dim();
        double t1, t2;
#ifdef ORACLE_TUNED
        {   // both products in one vectorised pass (nuts-rs: scalar_prods2 / scalar_prods3)
            const double *eps_ = end->psum.data(), *sps = start->psum.data(), *sp = start->p.data(), *ep = end->p.data();
            const double *ev = end->v.data(), *sv = start->v.data();
            double x1 = 0.0, x2 = 0.0;
            if (a >= 0 && b >= 0) {
#pragma omp simd reduction(+ : x1, x2)
                for (size_t i = 0; i < n; ++i) { const double t = (eps_[i] - sps[i]) + sp[i]; x1 += t * ev[i]; x2 += t * sv[i]; }
            } else if (b >= 0 && a < 0) {
#pragma omp simd reduction(+ : x1, x2)
                for (size_t i = 0; i < n; ++i) { const double t = eps_[i] + sps[i]; x1 += t * ev[i]; x2 += t * sv[i]; }
            } else {
#pragma omp simd reduction(+ : x1, x2)
                for (size_t i = 0; i < n; ++i) { const double t = (sps[i] - eps_[i]) + ep[i]; x1 += t * ev[i]; x2 += t * sv[i]; }
            }
            return (x1 < 0.0) || (x2 < 0.0);
        }
#endif
        if (a >= 0 && b >= 0) {
            // scalar_prods3(end.p_sum, -start.p_sum, +start.p ; end.v, start.v)
            t1 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma((end->psum[i] - start->psum[i]) + start->p[i], end->v[i], acc); });
            t2 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma((end->psum[i] - start->psum[i]) + start->p[i], start->v[i], acc); });
        } else if (b >= 0 && a < 0) {
            t1 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma(end->psum[i] + start->psum[i], end->v[i], acc); });
            t2 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma(end->psum[i] + start->psum[i], start->v[i], acc); });
        } else {
            t1 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma((start->psum[i] - end->psum[i]) + end->p[i], end->v[i], acc); });
            t2 = det_reduce(n, geo, [&](size_t i, double acc) { return std::fma((start->psum[i] - end->psum[i]) + end->p[i], start->v[i], acc); });
        }
        return (t1 < 0.0) || (t2 < 0.0);
    }

    // initialize_trajectory [A.2]: p = z * inv_std, inv_std = sqrt(1/sig2)   (A8)
    StateP init_trajectory(const State& cur, uint64_t seed, uint32_t chain, uint32_t draw_id, uint32_t purpose) {
        const size_t n = dim();
        StateP o = pool.get();
        o->q = cur.q; o->g = cur.g; o->U = cur.U;
        for (size_t j = 0; 2 * j < n; ++j) {
            double z0, z1;
            normal_pair(philox(seed, (uint32_t)j, chain, draw_id, purpose), &z0, &z1);
            o->p[2 * j] = z0;
            if (2 * j + 1 < n) o->p[2 * j + 1] = z1;
        }
        if (lr) {
            // p ~ N(0, M):  p = (z + sum_j V_j f_j) * sqrt(1 / sig2),  f_j = (1 / sqrt(lambda_j) - 1) <V_j, z>
            std::vector<double> f(lr_k);
            for (int j = 0; j < lr_k; ++j) f[j] = (1.0 / std::sqrt(lr_lam[j]) - 1.0) * det_dot(lr_V.data() + (size_t)j * n, o->p.data(), n, geo);
            for (size_t i = 0; i < n; ++i) {
                double w = o->p[i];
                for (int j = 0; j < lr_k; ++j) w = std::fma(lr_V[(size_t)j * n + i], f[j], w);
                o->p[i] = w * std::sqrt(1.0 / sig2[i]);
            }
        } else {
            for (size_t i = 0; i < n; ++i) o->p[i] = o->p[i] * std::sqrt(1.0 / sig2[i]);
        }
        velocity(o->p.data(), o->v.data());
        o->K = 0.5 * det_dot(o->p.data(), o->v.data(), n, geo);
        o->psum = o->p;
        o->idx = 0;
        o->H0 = o->energy();
        return o;
    }
};

// ---------------------------------------------------------------------------
// NUTS tree  [A.2, A.3]
// ---------------------------------------------------------------------------

struct DrawCtx {
    uint64_t seed; uint32_t chain; uint32_t draw;
    uint32_t doubling_depth = 0;  // depth of the main tree when this doubling started
    uint32_t leaf = 0;            // leaves integrated so far in this doubling
    bool check_turning = true;
    bool log_weights = false;     // crate_arithmetic: tree weights as log_size, merged with logaddexp [A.3 verbatim]
};

enum class Ext { Ok, Turning, Diverging, Fatal };

struct Tree {
    StateP left, right, draw;
    // multinomial weight of the tree, nuts-rs `log_size`, carried as w.m * 2^w.e (nphip_spec.h, "extended-range
    // tree weights")
    Weight w;
    double log_size = 0.0;   // the same weight in the crate's own form (maintained in both modes, used when ctx.log_weights)
    uint64_t depth = 0;
    bool is_main = false;

    // single_step
    Leap single_step(Hamiltonian& H, int dir, Collector& col, DrawCtx& ctx, Tree* out, DivergenceInfo* info) const {
        const State& start = dir > 0 ? *right : *left;
        StateP end;
        Leap r = H.leapfrog(start, dir, &col, &end, info);
        ctx.leaf += 1;
        if (r != Leap::Ok) return r;
        out->left = end; out->right = end; out->draw = end;
        out->depth = 0; out->is_main = false;
        out->w = w_leaf(-end->energy_error());
        out->log_size = -end->energy_error();
        return Leap::Ok;
    }

    // merge_into: multinomial inside sub-trees, biased progressive at the top level
    void merge_into(Tree&& other, int dir, DrawCtx& ctx) {
        if (dir > 0) right = other.right; else left = other.left;
        // sub-trees: accept other's draw w.p. w_other / (w_self + w_other); main tree (biased progressive
        // sampling): w.p. min(1, w_other / w_self)
        const Weight sum = w_add(w, other.w);
        const double log_sum = det_logaddexp(log_size, other.log_size);
        bool take;
        if (ctx.log_weights) {
            // SURVEY A.3 verbatim: self_w = is_main ? self.log_size : log_size;
            //   take if other.log_size >= self_w or rng.bool_with_prob(exp(other.log_size - self_w))
            // (the uniform is the contract's: same Philox block and words as the linear form)
            const double self_w = is_main ? log_size : log_sum;
            take = other.log_size >= self_w;
            if (!take) {
                uint32_t c3 = (uint32_t)RNG_MERGE | (ctx.doubling_depth << 8) | ((uint32_t)(depth >> 1) << 16);
                U4 r = philox(ctx.seed, ctx.leaf, ctx.chain, ctx.draw, c3);
                const int w0 = 2 * (int)(depth & 1);
                take = u01(r.v[w0], r.v[w0 + 1]) < det_exp(other.log_size - self_w);
            }
        } else {
            const double ref = is_main ? w_rel(w, sum.e) : sum.m;
            const double oth = w_rel(other.w, sum.e);
            take = is_main && (oth >= ref);
            if (!take) {
                uint32_t c3 = (uint32_t)RNG_MERGE | (ctx.doubling_depth << 8) | ((uint32_t)(depth >> 1) << 16);
                U4 r = philox(ctx.seed, ctx.leaf, ctx.chain, ctx.draw, c3);
                const int w0 = 2 * (int)(depth & 1);
                take = u01(r.v[w0], r.v[w0 + 1]) * ref < oth;
            }
        }
        if (take) draw = other.draw;
        depth += 1;
        w = sum;
        log_size = log_sum;
    }

    // extend (recursive doubling of `this` in direction dir)
    Ext extend(Hamiltonian& H, int dir, Collector& col, DrawCtx& ctx, DivergenceInfo* info) {
        Tree other;
        Leap r = single_step(H, dir, col, ctx, &other, info);
        if (r == Leap::Fatal) return Ext::Fatal;
        if (r == Leap::Diverge) return Ext::Diverging;
        while (other.depth < depth) {
            Ext e = other.extend(H, dir, col, ctx, info);
            if (e == Ext::Turning) return Ext::Turning;      // `self` unchanged
            if (e == Ext::Diverging) return Ext::Diverging;  // `self` unchanged
            if (e == Ext::Fatal) return Ext::Fatal;
        }
        bool turning = false;
        if (ctx.check_turning) {
            const State& first = dir > 0 ? *left : *other.left;
            const State& last = dir > 0 ? *other.right : *right;
            turning = H.is_turning(first, last);
            if (depth > 0) {
                if (!turning) turning = H.is_turning(*right, *other.right);
                if (!turning) turning = H.is_turning(*left, *other.left);
            }
        }
        merge_into(std::move(other), dir, ctx);
        return turning ? Ext::Turning : Ext::Ok;
    }
};

struct SampleInfo {
    uint64_t depth = 0;
    bool diverging = false;
    bool maxdepth_reached = false;
};

// ---------------------------------------------------------------------------
// Adaptation  [A.7 - A.9]
// ---------------------------------------------------------------------------

struct DualAverage {
    double k, t0, gamma;
    double log_step, log_step_adapted, hbar, mu;
    uint64_t count;
    void init(double initial_step, double k_, double t0_, double gamma_) {
        if (adam) { init_adam(initial_step, lr); return; }
        k = k_; t0 = t0_; gamma = gamma_;
        log_step = det_log(initial_step);
        log_step_adapted = log_step;
        hbar = 0.0;
        mu = det_log(10.0 * initial_step);
        count = 1;
    }
    void advance(double accept, double target) {
        if (adam) { advance_adam(accept, target); return; }
        double w = 1.0 / ((double)count + t0);
        hbar = (1.0 - w) * hbar + w * (target - accept);
        log_step = mu - hbar * std::sqrt((double)count) / gamma;
        double mk = det_exp(-k * det_log((double)count));
        log_step_adapted = mk * log_step + (1.0 - mk) * log_step_adapted;
        count += 1;
    }
    double current() const { return det_exp(log_step); }
    double adapted() const { return det_exp(log_step_adapted); }

    // step_size_adapt_method = "adam": Adam on log(step size), gradient = accept - target (the step grows while the
    // acceptance statistic is above the target); beta1 0.9, beta2 0.999, eps 1e-8; no averaged iterate, so
    // `step_size_bar` equals the current step.  The fields of the dual-averaging state are reused:
    // hbar = first moment, mu = second moment, (b1t, b2t) = running powers of the betas.
    bool adam = false;
    double lr = 0.05, b1t = 1.0, b2t = 1.0;
    void init_adam(double initial_step, double learning_rate) {
        adam = true; lr = learning_rate;
        log_step = det_log(initial_step); log_step_adapted = log_step;
        hbar = 0.0; mu = 0.0; b1t = 1.0; b2t = 1.0; count = 1;
    }
    void advance_adam(double accept, double target) {
        const double g = accept - target;
        hbar = 0.9 * hbar + (1.0 - 0.9) * g;
        mu = 0.999 * mu + (1.0 - 0.999) * (g * g);
        b1t = b1t * 0.9; b2t = b2t * 0.999;
        const double mhat = hbar / (1.0 - b1t), vhat = mu / (1.0 - b2t);
        log_step = log_step + lr * mhat / (std::sqrt(vhat) + 1e-8);
        log_step_adapted = log_step;
        count += 1;
    }
};

// RunningVariance (Welford); `current` = (M2, 1/(n-1))
struct RunningVariance {
    std::vector<double> mean, m2;
    uint64_t count = 0;
    explicit RunningVariance(size_t n) : mean(n, 0.0), m2(n, 0.0) {}
    void reset() { std::fill(mean.begin(), mean.end(), 0.0); std::fill(m2.begin(), m2.end(), 0.0); count = 0; }
    void add(const double* x) {
        count += 1;
        const size_t n = mean.size();
        if (count == 1) { for (size_t i = 0; i < n; ++i) mean[i] = x[i]; return; }
        const double inv = 1.0 / (double)count;
        for (size_t i = 0; i < n; ++i) {
            double diff = x[i] - mean[i];
            mean[i] = std::fma(diff, inv, mean[i]);
            m2[i] = std::fma(diff, x[i] - mean[i], m2[i]);
        }
    }
};

struct Chain {
    const oracle_settings_t& S;
    Model* model;
    Geometry geo;
    Hamiltonian H;
    uint32_t chain_id;  // global
    Collector col;
    DualAverage da;
    RunningVariance fg_q, fg_g, bg_q, bg_g;
    bool has_initial_mm = true;
    bool host_metric = false;   // the metric was handed in (low_rank_metric): the chain's own mass-matrix adaptation is off
    uint64_t last_update = 0;
    double last_accept = 0.0, last_accept_sym = 0.0;
    uint64_t early_end, final_window;
    bool tuning = true;
    StateP cur;  // current point (q, g, U valid)

    Chain(const oracle_settings_t& s, Model* m, uint32_t cid)
        : S(s), model(m), geo{s.waves_per_chain}, H(m, Geometry{s.waves_per_chain}), chain_id(cid),
          fg_q(m->dim), fg_g(m->dim), bg_q(m->dim), bg_g(m->dim) {
        H.max_energy_error = S.max_energy_error;
        da.adam = S.adam != 0;
        da.lr = S.adam_learning_rate;
        // window bounds [A.8]
        early_end = (uint64_t)std::ceil((double)S.num_tune * S.early_window);
        uint64_t ssw = (uint64_t)std::ceil((double)S.num_tune * S.step_size_window);
        final_window = (S.num_tune > ssw ? S.num_tune - ssw : 0) + 1;
        if (g_variant.floor_windows) {
            early_end = (uint64_t)((double)S.num_tune * S.early_window);
            ssw = (uint64_t)((double)S.num_tune * S.step_size_window);
            final_window = S.num_tune > ssw ? S.num_tune - ssw : 0;
        }
    }

    // Model::init_position: src/pyfunc.rs:540-544 (U(-2,2)), src/stan.rs:798-808 (N(0,1))
    void init_position(uint32_t attempt, const double* explicit_init, double* q) {
        const size_t n = model->dim;
        if (S.init_kind == 2) { memcpy(q, explicit_init, n * sizeof(double)); return; }
        for (size_t j = 0; 2 * j < n; ++j) {
            U4 r = philox(S.seed, (uint32_t)j, chain_id, attempt, RNG_INIT);
            double a, b;
            if (S.init_kind == 0) {
                a = std::fma(4.0, u01(r.v[0], r.v[1]), -2.0);
                b = std::fma(4.0, u01(r.v[2], r.v[3]), -2.0);
            } else {
                normal_pair(r, &a, &b);
            }
            q[2 * j] = a;
            if (2 * j + 1 < n) q[2 * j + 1] = b;
        }
    }

    // step-size heuristic search [A.7]
    bool step_size_search(uint32_t search_id) {
        if (S.fixed_step_size) { H.step_size = S.initial_step; da.init(S.initial_step, S.da_k, S.da_t0, S.da_gamma); return true; }
        H.step_size = S.initial_step;
        StateP st = H.init_trajectory(*cur, S.seed, chain_id, search_id, RNG_SS_MOMENTUM);
        Collector c; c.register_init();
        StateP nxt; DivergenceInfo info;
        Leap r = H.leapfrog(*st, +1, &c, &nxt, &info);
        if (r == Leap::Fatal) return false;
        if (r != Leap::Ok) { da.init(S.initial_step, S.da_k, S.da_t0, S.da_gamma); return true; }
        double accept = c.mean.value();
        int dir = accept > S.target_accept ? +1 : -1;
        for (int it = 0; it < 100; ++it) {
            Collector c2; c2.register_init();
            r = H.leapfrog(*st, dir, &c2, &nxt, &info);
            if (r == Leap::Fatal) return false;
            if (r != Leap::Ok) { H.step_size = S.initial_step; da.init(S.initial_step, S.da_k, S.da_t0, S.da_gamma); return true; }
            accept = c2.mean.value();
            if (dir > 0) {
                if (accept <= S.target_accept || H.step_size > 1e5) { da.init(H.step_size, S.da_k, S.da_t0, S.da_gamma); return true; }
                H.step_size *= 2.0;
            } else {
                if (accept >= S.target_accept || H.step_size < 1e-10) { da.init(H.step_size, S.da_k, S.da_t0, S.da_gamma); return true; }
                H.step_size /= 2.0;
            }
        }
        H.step_size = S.initial_step;
        da.init(S.initial_step, S.da_k, S.da_t0, S.da_gamma);
        return true;
    }

    void update_stepsize(uint32_t draw, bool use_best_guess) {
        if (S.fixed_step_size) return;
        double step = use_best_guess ? da.adapted() : da.current();
        if (S.step_size_jitter > 0.0) {
            U4 r = philox(S.seed, 0, chain_id, draw, RNG_JITTER);
            double u = u01(r.v[0], r.v[1]);
            step *= std::fma(2.0 * S.step_size_jitter, u, 1.0 - S.step_size_jitter);
        }
        if (step > S.max_step_size) step = S.max_step_size;
        H.step_size = step;
    }

    // mass-matrix refresh from the foreground estimator [A.9];
    // formulas corroborated in-tree by python/nutpie/normalizing_flow.py:1906-1915.
    bool update_mass_matrix() {
        if (fg_q.count < (uint64_t)g_variant.min_refresh) return false;
        const size_t n = model->dim;
        if (S.use_grad_based_mass_matrix) {
            for (size_t i = 0; i < n; ++i) {
                double val = std::sqrt(fg_q.m2[i] / fg_g.m2[i]);
                if (!std::isfinite(val)) continue;
                val = val < 1e-20 ? 1e-20 : (val > 1e20 ? 1e20 : val);
                H.sig2[i] = val;
            }
        } else {
            const double scale = 1.0 / (double)(fg_q.count - 1);
            for (size_t i = 0; i < n; ++i) {
                double val = fg_q.m2[i] * scale;
                if (!std::isfinite(val)) continue;
                val = val < 1e-20 ? 1e-20 : (val > 1e20 ? 1e20 : val);
                H.sig2[i] = val;
            }
        }
        return true;
    }

    bool init(const double* explicit_init) {
        const size_t n = model->dim;
        cur = H.pool.get();
        bool ok = false;
        for (int attempt = 0; attempt < S.num_try_init; ++attempt) {
            init_position((uint32_t)attempt, explicit_init, cur->q.data());
            double lp;
            int64_t code = model->logp(cur->q.data(), cur->g.data(), &lp);
            if (code < 0) { g_error = "logp callback returned fatal code " + std::to_string(code); return false; }
            if (code == 0 && std::isfinite(lp)) { cur->U = -lp; ok = true; break; }
        }
        if (!ok) { g_error = "could not find a finite initial point"; return false; }
        if (S.adapt_mass_matrix) {
            // mass matrix strategy init: estimators see the initial point; sig2 = 1/clamp(|g|)
            fg_q.add(cur->q.data()); bg_q.add(cur->q.data());
            fg_g.add(cur->g.data()); bg_g.add(cur->g.data());
            for (size_t i = 0; i < n; ++i) {
                double a = std::fabs(cur->g[i]);
                a = a < 1e-20 ? 1e-20 : (a > 1e20 ? 1e20 : a);  // NaN falls through to the isfinite test
                double val = 1.0 / a;
                H.sig2[i] = std::isfinite(val) ? val : 1.0;
            }
        }
        if (g_variant.search_mode == 0) { H.step_size = S.initial_step; da.init(S.initial_step, S.da_k, S.da_t0, S.da_gamma); return true; }
        return step_size_search(0xffffffffu);
    }

    // GlobalStrategy::adapt [A.8]
    bool adapt(uint64_t draw, bool draw_is_good) {
        last_accept = col.mean.value();
        last_accept_sym = col.mean_sym.value();
        if (draw >= S.num_tune) { tuning = false; return true; }
        if (draw < final_window) {
            bool is_early = draw < early_end;
            uint64_t switch_freq = is_early ? S.early_mass_matrix_switch_freq : S.mass_matrix_switch_freq;
            bool did_change = false;
            bool is_late = switch_freq + draw > final_window;
            if (S.adapt_mass_matrix && !host_metric) {
                if (draw_is_good) {
                    fg_q.add(cur->q.data()); fg_g.add(cur->g.data());
                    bg_q.add(cur->q.data()); bg_g.add(cur->g.data());
                }
                bool could_switch = bg_q.count >= switch_freq;
                bool force_update = false;
                if (could_switch && !is_late) {
                    std::swap(fg_q, bg_q); std::swap(fg_g, bg_g);
                    bg_q.reset(); bg_g.reset();
                    force_update = true;
                }
                if (force_update || (draw - last_update >= S.mass_matrix_update_freq)) did_change = update_mass_matrix();
                if (did_change) last_update = draw;
            }
            if (is_late && g_variant.late_sym) da.advance(last_accept_sym, S.target_accept);
            else da.advance(last_accept, S.target_accept);
            if (did_change && has_initial_mm && g_variant.search_mode == 2) {
                has_initial_mm = false;
                if (!step_size_search((uint32_t)draw)) return false;
            } else {
                update_stepsize((uint32_t)draw, false);
            }
            return true;
        }
        da.advance(g_variant.late_sym ? last_accept_sym : last_accept, S.target_accept);
        update_stepsize((uint32_t)draw, g_variant.last_bar && draw == S.num_tune - 1);
        return true;
    }

    // metrics handed in from outside (oracle_settings_t::low_rank_metric): update u applies before draw metric_draws[u]
    bool maybe_set_metric(uint64_t draw_idx, uint64_t local_chain) {
        if (!S.low_rank_metric) return true;
        const size_t n = model->dim, k = (size_t)S.metric_k, nc = (size_t)S.num_chains;
        for (int u = 0; u < S.n_metric_updates; ++u) {
            if (S.metric_draws[u] != draw_idx) continue;
            H.set_metric((int)k, S.metric_sig2 + ((size_t)u * nc + local_chain) * n,
                         k ? S.metric_V + (((size_t)u * nc + local_chain) * k) * n : nullptr, k ? S.metric_lam + ((size_t)u * nc + local_chain) * k : nullptr);
            host_metric = true;
            has_initial_mm = false;
            return step_size_search(0x80000000u | (uint32_t)draw_idx);
        }
        return true;
    }

    // nuts::draw [A.2]
    bool draw(uint64_t draw_idx, SampleInfo* info, StateP* out, DivergenceInfo* div_out = nullptr) {
        StateP init = H.init_trajectory(*cur, S.seed, chain_id, (uint32_t)draw_idx, RNG_MOMENTUM);
        col.set_running((S.crate_arithmetic & 2) != 0);
        col.register_init();
        Tree tree;
        tree.left = init; tree.right = init; tree.draw = init;
        tree.depth = 0; tree.w = Weight{}; tree.log_size = 0.0; tree.is_main = true;
        DrawCtx ctx{S.seed, chain_id, (uint32_t)draw_idx};
        ctx.log_weights = (S.crate_arithmetic & 1) != 0;
        DivergenceInfo dinfo;
        while (tree.depth < S.maxdepth) {
            U4 r = philox(S.seed, (uint32_t)tree.depth, chain_id, (uint32_t)draw_idx, RNG_DIRECTION);
            int dir = (r.v[0] & 1u) ? +1 : -1;
            ctx.doubling_depth = (uint32_t)tree.depth;
            ctx.leaf = 0;
            // mindepth: no U-turn termination before the main tree can reach depth > mindepth
            ctx.check_turning = (S.check_turning != 0) && (tree.depth + 1 > S.mindepth);
            Ext e = tree.extend(H, dir, col, ctx, &dinfo);
            if (e == Ext::Fatal) return false;
            if (e == Ext::Diverging) {
                info->depth = tree.depth; info->diverging = true; *out = tree.draw;
                if (div_out) *div_out = std::move(dinfo);
                return true;
            }
            if (e == Ext::Turning) { info->depth = tree.depth; *out = tree.draw; return true; }
        }
        info->depth = tree.depth;
        info->maxdepth_reached = true;
        *out = tree.draw;
        return true;
    }
};

// ---------------------------------------------------------------------------
// Sampler driver: one chain per task, min(chains, n_threads) workers — the
// reference's `cores` model (python/nutpie/sample.py:856-857, 1061-1070).
// ---------------------------------------------------------------------------

template <class MakeModel>
int run_sampler(const oracle_settings_t* S, uint64_t dim, MakeModel make_model, const double* init_points,
                oracle_trace_t* out, double* seconds) {
    const uint64_t T = S->num_tune + S->num_draws;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        for (;;) {
            uint64_t c = next.fetch_add(1);
            if (c >= S->num_chains || failed.load()) return;
            std::unique_ptr<Model> model = make_model();
            Chain chain(*S, model.get(), (uint32_t)(S->chain_offset + c));
            const double* ip = (S->init_kind == 2 && init_points) ? init_points + c * dim : nullptr;
            if (!chain.init(ip)) {
                std::lock_guard<std::mutex> lk(g_error_mutex);
                g_last_error = g_error; failed.store(1); return;
            }
            for (uint64_t d = 0; d < T; ++d) {
                SampleInfo info; StateP st; DivergenceInfo dinfo;
                if (!chain.maybe_set_metric(d, c) || !chain.draw(d, &info, &st, &dinfo)) {
                    std::lock_guard<std::mutex> lk(g_error_mutex);
                    g_last_error = g_error; failed.store(1); return;
                }
                // DrawGradCollector: a diverging draw that did not move is not fed to the estimators
                bool good = info.diverging ? (st->idx != 0) : true;
                double energy = st->energy(), eerr = st->energy_error();
                int64_t idx = st->idx;
                uint64_t n_steps = chain.col.mean.count;
                double mta = chain.col.mean.value(), mtas = chain.col.mean_sym.value();
                chain.cur = st;
                if (!chain.adapt(d, good)) {
                    std::lock_guard<std::mutex> lk(g_error_mutex);
                    g_last_error = g_error; failed.store(1); return;
                }
                const uint64_t o = c * T + d;
                if (out->draws) memcpy(out->draws + o * dim, st->q.data(), dim * sizeof(double));
                if (out->gradient) memcpy(out->gradient + o * dim, st->g.data(), dim * sizeof(double));
                if (out->mass_matrix_inv) memcpy(out->mass_matrix_inv + o * dim, chain.H.sig2.data(), dim * sizeof(double));
                if (out->divergence_start) {
                    // rows of draws that did not diverge stay NaN (python/nutpie/sample.py:631-650: the columns are nullable)
                    double* rows[4] = {out->divergence_start, out->divergence_end, out->divergence_momentum, out->divergence_start_gradient};
                    for (auto* r : rows) for (uint64_t i = 0; i < dim; ++i) r[o * dim + i] = NAN;
                    if (info.diverging && dinfo.has_start) {
                        memcpy(rows[0] + o * dim, dinfo.start_q.data(), dim * sizeof(double));
                        if (dinfo.has_end) memcpy(rows[1] + o * dim, dinfo.end_q.data(), dim * sizeof(double));
                        memcpy(rows[2] + o * dim, dinfo.start_p.data(), dim * sizeof(double));
                        memcpy(rows[3] + o * dim, dinfo.start_g.data(), dim * sizeof(double));
                    }
                }
                if (out->depth) out->depth[o] = (int64_t)info.depth;
                if (out->n_steps) out->n_steps[o] = (int64_t)n_steps;
                if (out->index_in_trajectory) out->index_in_trajectory[o] = idx;
                if (out->diverging) out->diverging[o] = info.diverging;
                if (out->maxdepth_reached) out->maxdepth_reached[o] = info.maxdepth_reached;
                if (out->tuning) out->tuning[o] = chain.tuning;
                if (out->energy) out->energy[o] = energy;
                if (out->energy_error) out->energy_error[o] = eerr;
                if (out->logp) out->logp[o] = -st->U;
                if (out->step_size) out->step_size[o] = chain.H.step_size;
                if (out->step_size_bar) out->step_size_bar[o] = chain.da.adapted();
                if (out->mean_tree_accept) out->mean_tree_accept[o] = mta;
                if (out->mean_tree_accept_sym) out->mean_tree_accept_sym[o] = mtas;
            }
        }
    };
    int nt = S->n_threads < 1 ? 1 : S->n_threads;
    if ((uint64_t)nt > S->num_chains) nt = (int)S->num_chains;
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? 1 : 0;
}

}  // namespace

extern "C" {

void oracle_default_settings(oracle_settings_t* s) {
    memset(s, 0, sizeof(*s));
    s->seed = 0; s->num_tune = 400; s->num_draws = 1000; s->num_chains = 6;  // docs/_freeze/index (6 chains, 400+1000)
    s->maxdepth = 10; s->mindepth = 0; s->check_turning = 1;                    // sample.py:896-899
    s->use_grad_based_mass_matrix = 1;
    s->max_energy_error = 1000.0;                                               // docs/sampling-options.qmd:73
    s->early_window = 0.3; s->step_size_window = 0.15;
    s->mass_matrix_switch_freq = 80; s->early_mass_matrix_switch_freq = 10; s->mass_matrix_update_freq = 1;
    s->initial_step = 0.1; s->target_accept = 0.8;                              // docs/sampling-options.qmd:71,82
    s->step_size_jitter = 0.0; s->max_step_size = INFINITY;
    s->da_k = 0.75; s->da_t0 = 10.0; s->da_gamma = 0.05;
    s->fixed_step_size = 0; s->adapt_mass_matrix = 1; s->adam = 0; s->adam_learning_rate = 0.05;
    s->init_kind = 0; s->num_try_init = 100;
    s->waves_per_chain = 1; s->n_threads = 1; s->chain_offset = 0;
}

int oracle_sample_tridiag(const oracle_settings_t* s, uint64_t dim, const double* mu, const double* diag,
                          const double* offdiag, const double* init_points, oracle_trace_t* out, double* seconds) {
    auto mk = [&]() {
        auto m = std::make_unique<TridiagModel>();
        m->dim = dim;
        m->geo = Geometry{s->waves_per_chain};
        m->mu.assign(dim, 0.0); if (mu) m->mu.assign(mu, mu + dim);
        m->a.assign(diag, diag + dim);
        m->b.assign(dim, 0.0); if (offdiag && dim > 1) std::copy(offdiag, offdiag + dim - 1, m->b.begin());
        return std::unique_ptr<Model>(std::move(m));
    };
    return run_sampler(s, dim, mk, init_points, out, seconds);
}

int oracle_sample_callback(const oracle_settings_t* s, uint64_t dim, oracle_logp_fn fn, void* user,
                           const double* init_points, oracle_trace_t* out, double* seconds) {
    auto mk = [&]() {
        auto m = std::make_unique<CallbackModel>();
        m->dim = dim; m->fn = fn; m->user = user;
        return std::unique_ptr<Model>(std::move(m));
    };
    return run_sampler(s, dim, mk, init_points, out, seconds);
}

int oracle_sample_dense(const oracle_settings_t* s, uint64_t dim, const double* mu, const double* P, const double* init_points,
                        oracle_trace_t* out, double* seconds) {
    auto mk = [&]() {
        auto m = std::make_unique<DenseModel>();
        m->dim = dim;
        m->geo = Geometry{s->waves_per_chain};
        m->P = P;
        m->mu.assign(dim, 0.0); if (mu) m->mu.assign(mu, mu + dim);
        return std::unique_ptr<Model>(std::move(m));
    };
    return run_sampler(s, dim, mk, init_points, out, seconds);
}

// one evaluation of the dense model on n rows: grad[n][dim], logp[n]
void oracle_dense_grad(uint64_t n, uint64_t dim, const double* x, const double* mu, const double* P, int waves, double* grad, double* logp) {
    DenseModel m;
    m.dim = dim; m.P = P; m.geo = Geometry{waves};
    m.mu.assign(dim, 0.0); if (mu) m.mu.assign(mu, mu + dim);
    for (uint64_t i = 0; i < n; ++i) m.logp(x + i * dim, grad + i * dim, logp + i);
}

void oracle_set_variant(int min_refresh, int search_mode, int late_sym, int last_bar, int floor_windows) {
    g_variant.min_refresh = min_refresh; g_variant.search_mode = search_mode; g_variant.late_sym = late_sym;
    g_variant.last_bar = last_bar; g_variant.floor_windows = floor_windows;
}

const char* oracle_last_error(void) { return g_last_error.c_str(); }

void oracle_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    U4 r = philox(seed, c0, c1, c2, c3);
    memcpy(out, r.v, sizeof(r.v));
}

void oracle_detmath(int fn, uint64_t n, const double* x, double* y) {
    for (uint64_t i = 0; i < n; ++i) {
        double s, c;
        switch (fn) {
            case 0: y[i] = det_exp(x[i]); break;
            case 1: y[i] = det_log(x[i]); break;
            case 2: y[i] = det_log1p(x[i]); break;
            case 3: det_sincos2pi(x[i], &s, &c); y[i] = s; break;
            default: det_sincos2pi(x[i], &s, &c); y[i] = c; break;
        }
    }
}

double oracle_logaddexp(double a, double b) { return det_logaddexp(a, b); }

// extended-range tree weights (unit hooks)
void oracle_w_leaf(double neg_energy_error, double* m, int64_t* e) { Weight w = w_leaf(neg_energy_error); *m = w.m; *e = w.e; }
void oracle_w_add(double m1, int64_t e1, double m2, int64_t e2, double* m, int64_t* e) {
    Weight a; a.m = m1; a.e = e1; Weight b; b.m = m2; b.e = e2;
    Weight o = w_add(a, b); *m = o.m; *e = o.e;
}

void oracle_normals(uint64_t seed, uint32_t chain, uint32_t draw, uint32_t purpose, uint64_t n, double* out) {
    for (uint64_t j = 0; 2 * j < n; ++j) {
        double z0, z1;
        normal_pair(philox(seed, (uint32_t)j, chain, draw, purpose), &z0, &z1);
        out[2 * j] = z0;
        if (2 * j + 1 < n) out[2 * j + 1] = z1;
    }
}

double oracle_dot(const double* x, const double* y, uint64_t n, int waves) { return det_dot(x, y, n, Geometry{waves}); }

double oracle_leapfrog_tridiag(uint64_t dim, const double* mu, const double* diag, const double* offdiag,
                               const double* sig2, double eps, int waves, double* q, double* p, double* g,
                               double* kinetic, double* potential) {
    TridiagModel m;
    m.dim = dim; m.geo = Geometry{waves};
    m.mu.assign(dim, 0.0); if (mu) m.mu.assign(mu, mu + dim);
    m.a.assign(diag, diag + dim);
    m.b.assign(dim, 0.0); if (offdiag && dim > 1) std::copy(offdiag, offdiag + dim - 1, m.b.begin());
    Hamiltonian H(&m, Geometry{waves});
    H.sig2.assign(sig2, sig2 + dim);
    H.step_size = std::fabs(eps);
    H.max_energy_error = INFINITY;
    State s;
    s.q.assign(q, q + dim); s.p.assign(p, p + dim); s.g.assign(g, g + dim);
    s.v.resize(dim); s.psum.assign(p, p + dim);
    s.idx = 0; s.H0 = 0.0;
    StateP o; DivergenceInfo info;
    Leap r = H.leapfrog(s, eps >= 0 ? +1 : -1, nullptr, &o, &info);
    if (r != Leap::Ok) return NAN;
    memcpy(q, o->q.data(), dim * 8); memcpy(p, o->p.data(), dim * 8); memcpy(g, o->g.data(), dim * 8);
    if (kinetic) *kinetic = o->K;
    if (potential) *potential = o->U;
    return o->K + o->U;
}

void oracle_dual_average(double initial_step, double target, double k, double t0, double gamma, uint64_t n,
                         const double* accept, double* step, double* step_bar) {
    DualAverage da;
    da.init(initial_step, k, t0, gamma);
    for (uint64_t i = 0; i < n; ++i) {
        da.advance(accept[i], target);
        step[i] = da.current();
        step_bar[i] = da.adapted();
    }
}

void oracle_welford(uint64_t n, uint64_t dim, const double* samples, double* mean, double* m2) {
    RunningVariance rv(dim);
    for (uint64_t i = 0; i < n; ++i) rv.add(samples + i * dim);
    memcpy(mean, rv.mean.data(), dim * 8);
    memcpy(m2, rv.m2.data(), dim * 8);
}

void oracle_lr_velocity(uint64_t dim, int k, const double* sig2, const double* V, const double* lam, int waves, const double* p, double* v) {
    TridiagModel m; m.dim = dim;
    Hamiltonian H(&m, Geometry{waves});
    H.set_metric(k, sig2, V, lam);
    H.velocity(p, v);
}

int oracle_is_turning(uint64_t dim, const double* sig2, int waves, int64_t idx1, const double* p1,
                      const double* psum1, int64_t idx2, const double* p2, const double* psum2) {
    TridiagModel m; m.dim = dim;
    Hamiltonian H(&m, Geometry{waves});
    H.sig2.assign(sig2, sig2 + dim);
    State a, b;
    a.p.assign(p1, p1 + dim); a.psum.assign(psum1, psum1 + dim); a.v.resize(dim); a.idx = idx1;
    b.p.assign(p2, p2 + dim); b.psum.assign(psum2, psum2 + dim); b.v.resize(dim); b.idx = idx2;
    for (uint64_t i = 0; i < dim; ++i) { a.v[i] = sig2[i] * a.p[i]; b.v[i] = sig2[i] * b.p[i]; }
    return H.is_turning(a, b) ? 1 : 0;
}

}  // extern "C"
