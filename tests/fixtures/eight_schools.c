/*
 * eight_schools.c — test fixture (written for this repo; not reference code).
 *
 * Non-centred eight-schools model on the unconstrained scale, D = 10:
 *   x = (mu, log_tau, theta_tilde[8]);  tau = exp(log_tau)
 *   mu ~ N(0, 5); tau ~ HalfCauchy(0, 5); theta_tilde ~ N(0, 1); y_j ~ N(mu + tau*theta_tilde_j, sigma_j)
 * Exposed twice:
 *   (a) with the reference's raw C logp callback signature (src/pymc.rs:23-29), and
 *   (b) behind BridgeStan's C API names (bs_log_density_gradient, bs_param_unc_num,
 *       bs_free_error_msg) so the engine's BridgeStan adapter can be exercised without a Stan
 *       toolchain (BASELINE.json config 4: "Stan 8-schools via bridgestan, host logp").
 * Constants are dropped (propto=true); the log-Jacobian of tau = exp(log_tau) is included
 * (jacobian=true), as src/stan.rs:455-458 requests.
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double Y[8] = {28, 8, -3, 7, -1, 1, 18, 12};
static const double S[8] = {15, 10, 16, 11, 9, 11, 10, 18};

static double eval(const double* x, double* g) {
    const double mu = x[0], tau = exp(x[1]);
    double lp = -mu * mu / 50.0 - log1p(tau * tau / 25.0) + x[1];
    double gmu = -mu / 25.0;
    double gtau = -(2.0 * tau / 25.0) / (1.0 + tau * tau / 25.0);
    for (int j = 0; j < 8; ++j) {
        const double t = x[2 + j];
        const double r = (Y[j] - mu - tau * t) / S[j];
        lp += -0.5 * t * t - 0.5 * r * r;
        gmu += r / S[j];
        gtau += r * t / S[j];
        g[2 + j] = -t + r * tau / S[j];
    }
    g[0] = gmu;
    g[1] = gtau * tau + 1.0;
    return lp;
}

/* (a) raw callback: 0 ok, 4 non-finite logp, 3 non-finite gradient (codes of compile_pymc.py:983-1004) */
int64_t eight_schools_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    if (dim != 10) return -1;
    double lp = eval(x, grad);
    *logp = lp;
    for (int i = 0; i < 10; ++i)
        if (!isfinite(grad[i])) return 3;
    if (!isfinite(lp)) return 4;
    return 0;
}

/* any dimension: independent normals with scales 0.5, 0.8, ..., 2.3, 0.5, ... (sums in index order; used for wide rows and many
 * chains through the host-callback path) */
int scaled_normal_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    double lp = 0.0;
    for (uint64_t i = 0; i < dim; ++i) {
        const double sd = 0.5 + 0.3 * (double)(i % 7);
        const double z = x[i] / sd;
        lp += -0.5 * z * z;
        grad[i] = -z / sd;
    }
    *logp = lp;
    return 0;
}

/* the same density behind a numba-style `int64` return whose upper half is NOT clean: the reference reads the return as
 * `c_int` (src/pymc.rs:23-29), i.e. the low 32 bits — so must the engine */
int64_t eight_schools_logp_dirty_high(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    const int64_t rc = eight_schools_logp(dim, x, grad, logp, user);
    return (int64_t)(((uint64_t)0x5eed0000u << 32) | (uint32_t)rc);
}

/* raw expand callback (src/pymc.rs:31-37 / compile_pymc.py:1018-1041): unconstrained draw -> (mu, tau, theta_tilde[8], theta[8]) */
int eight_schools_expand(uint64_t dim, uint64_t expanded, const double* x, double* out, void* user) {
    (void)user;
    if (dim != 10) return -1;
    if (expanded != 18) return -1;
    const double mu = x[0], tau = exp(x[1]);
    out[0] = mu;
    out[1] = tau;
    for (int j = 0; j < 8; ++j) { out[2 + j] = x[2 + j]; out[10 + j] = mu + tau * x[2 + j]; }
    return 0;
}
int failing_expand(uint64_t dim, uint64_t expanded, const double* x, double* out, void* user) {
    (void)dim; (void)expanded; (void)x; (void)out; (void)user;
    return -2;
}

/* a callback that fails on demand: recoverable when x[0] > 3, fatal when x[0] > 1e6 (never reached) */
int64_t failing_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    double lp = 0.0;
    for (uint64_t i = 0; i < dim; ++i) { lp -= 0.5 * x[i] * x[i]; grad[i] = -x[i]; }
    *logp = lp;
    if (x[0] > 1e6) return -2;
    if (x[0] > 2.5) return 1;
    return 0;
}

int64_t fatal_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)dim; (void)x; (void)grad; (void)logp; (void)user;
    return -7;
}

/* (b) BridgeStan C API stand-in */
typedef struct { int dim; int calls; } bs_model;
bs_model* bs_model_construct(const char* data, unsigned int seed, char** err) {
    (void)data; (void)seed; (void)err;
    bs_model* m = (bs_model*)malloc(sizeof(bs_model));
    m->dim = 10; m->calls = 0;
    return m;
}
void bs_model_destruct(bs_model* m) { free(m); }
int bs_param_unc_num(const bs_model* m) { return m->dim; }
void bs_free_error_msg(char* e) { free(e); }
int bs_log_density_gradient(const bs_model* m, bool propto, bool jacobian, const double* theta, double* val, double* grad, char** err) {
    (void)m;
    if (!propto || !jacobian) {
        if (err) { *err = (char*)malloc(32); strcpy(*err, "unsupported flags"); }
        return 1;
    }
    *val = eval(theta, grad);
    if (theta[1] > 20.0) {  /* Stan would throw on overflow: exercise the error path */
        if (err) { *err = (char*)malloc(32); strcpy(*err, "tau overflow"); }
        return 1;
    }
    return 0;
}
