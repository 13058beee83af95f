/*
 * bs_standin.c — TEST FIXTURE: a stand-in for a BridgeStan model library.
 *
 * BridgeStan and the Stan toolchain are not installable in the build image, so the engine's BridgeStan adapters
 * (nphip_model_bridgestan: reference src/stan.rs:454-463; nphip_model_set_bridgestan_expand: src/stan.rs:473-520,
 * 774-796) are exercised against this library, which exports the part of BridgeStan's C API (bridgestan.h) the
 * reference uses, for two tiny models selected by the `data` string of bs_model_construct:
 *
 *   "halfnormal"  the Stan program of the reference's golden-vector test (tests/test_stan.py:282-302):
 *                     parameters { real<lower=0> a; }  model { a ~ normal(0, 1); }
 *                     generated quantities { real b = normal_rng(0, 1) + a; }
 *                 unconstrained x = log a; with jacobian and propto: logp = x - exp(2 x) / 2.
 *   "matrix"      parameters { matrix[2, 3] m; }  transformed parameters { matrix[3, 2] mt = m'; }
 *                 model { to_vector(m) ~ std_normal(); }  generated quantities { real s = sum(m) + normal_rng(0, 1); }
 *                 — outputs are COLUMN-MAJOR, as Stan writes them (the reference transposes: src/stan.rs:507-516, 671-711).
 *
 * Also exports the same density as the reference's raw C callbacks (src/pymc.rs:23-37) for the PyMC flavour of the
 * golden-vector test (tests/test_pymc.py:533-552: HalfNormal("a") on the log scale).
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- raw callbacks: HalfNormal(1) on the log scale, D = 1 */
int halfnormal_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    if (dim != 1) return -1;
    const double e = exp(2.0 * x[0]);
    *logp = x[0] - 0.5 * e;
    grad[0] = 1.0 - e;
    if (!isfinite(*logp)) return 4;   /* python/nutpie/compile_pymc.py:983-1004: non-finite logp is recoverable */
    return 0;
}
int halfnormal_expand(uint64_t dim, uint64_t expanded, const double* x, double* out, void* user) {
    (void)user;
    if (dim != 1 || expanded != 1) return -1;
    out[0] = exp(x[0]);
    return 0;
}

/* ---- BridgeStan C API */
typedef struct { int kind; int unc; int n_out; } bs_model;
typedef struct { uint64_t s; } bs_rng;

static char* dup_msg(const char* m) { char* e = (char*)malloc(strlen(m) + 1); strcpy(e, m); return e; }

bs_model* bs_model_construct(const char* data, unsigned int seed, char** err) {
    (void)seed;
    bs_model* m = (bs_model*)malloc(sizeof(bs_model));
    if (data && strstr(data, "matrix")) { m->kind = 1; m->unc = 6; m->n_out = 13; }
    else if (!data || strstr(data, "halfnormal")) { m->kind = 0; m->unc = 1; m->n_out = 2; }
    else { free(m); if (err) *err = dup_msg("unknown model"); return NULL; }
    return m;
}
void bs_model_destruct(bs_model* m) { free(m); }
int bs_param_unc_num(const bs_model* m) { return m->unc; }
int bs_param_num(const bs_model* m, bool include_tp, bool include_gq) {
    if (m->kind == 0) return 1 + (include_gq ? 1 : 0);
    return 6 + (include_tp ? 6 : 0) + (include_gq ? 1 : 0);
}
const char* bs_param_names(const bs_model* m, bool include_tp, bool include_gq) {
    if (m->kind == 0) return include_gq ? "a,b" : "a";
    if (include_tp && include_gq) return "m.1.1,m.2.1,m.1.2,m.2.2,m.1.3,m.2.3,mt.1.1,mt.2.1,mt.3.1,mt.1.2,mt.2.2,mt.3.2,s";
    if (include_tp) return "m.1.1,m.2.1,m.1.2,m.2.2,m.1.3,m.2.3,mt.1.1,mt.2.1,mt.3.1,mt.1.2,mt.2.2,mt.3.2";
    if (include_gq) return "m.1.1,m.2.1,m.1.2,m.2.2,m.1.3,m.2.3,s";
    return "m.1.1,m.2.1,m.1.2,m.2.2,m.1.3,m.2.3";
}
void bs_free_error_msg(char* e) { free(e); }

int bs_log_density_gradient(const bs_model* m, bool propto, bool jacobian, const double* theta, double* val, double* grad, char** err) {
    if (!propto || !jacobian) { if (err) *err = dup_msg("unsupported flags"); return 1; }
    if (m->kind == 0) {
        const double e = exp(2.0 * theta[0]);
        *val = theta[0] - 0.5 * e;
        grad[0] = 1.0 - e;
        if (theta[0] > 300.0) { if (err) *err = dup_msg("overflow"); return 1; }   /* Stan would throw */
        return 0;
    }
    double lp = 0.0;
    for (int i = 0; i < 6; ++i) { lp -= 0.5 * theta[i] * theta[i]; grad[i] = -theta[i]; }
    *val = lp;
    return 0;
}

/* the generator only has to be deterministic per seed: splitmix64 + Box-Muller */
bs_rng* bs_rng_construct(unsigned int seed, char** err) {
    (void)err;
    bs_rng* r = (bs_rng*)malloc(sizeof(bs_rng));
    r->s = 0x9E3779B97F4A7C15ull * ((uint64_t)seed + 1u);
    return r;
}
void bs_rng_destruct(bs_rng* r) { free(r); }
static double rng_u01(bs_rng* r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((double)(z >> 11) + 0.5) * 0x1.0p-53;
}
static double rng_normal(bs_rng* r) {
    const double u1 = rng_u01(r), u2 = rng_u01(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

int bs_param_constrain(const bs_model* m, bool include_tp, bool include_gq, const double* theta_unc, double* theta, bs_rng* rng, char** err) {
    if (include_gq && !rng) { if (err) *err = dup_msg("generated quantities need an rng"); return 1; }
    if (m->kind == 0) {
        theta[0] = exp(theta_unc[0]);
        if (include_gq) theta[1] = rng_normal(rng) + theta[0];
        return 0;
    }
    /* m is matrix[2, 3]; the unconstrained vector IS its column-major serialisation */
    int o = 0;
    double sum = 0.0;
    for (int i = 0; i < 6; ++i) { theta[o++] = theta_unc[i]; sum += theta_unc[i]; }
    if (include_tp) {
        /* mt = m' is matrix[3, 2], written column-major: mt(r, c) = m(c, r), flat index r + 3 c; m(c, r) sits at c + 2 r */
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 3; ++r) theta[o++] = theta_unc[c + 2 * r];
    }
    if (include_gq) theta[o++] = sum + rng_normal(rng);
    if (theta_unc[0] > 1e6) { if (err) *err = dup_msg("constrain failed"); return 1; }
    return 0;
}
