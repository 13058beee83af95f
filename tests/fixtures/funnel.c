/*
 * funnel.c — test fixture (written for this repo; not reference code).
 *
 * Neal's funnel as the reference's documentation defines it (docs/sample-stats.qmd:18-22):
 *   log_sigma ~ Normal(0, 1);  x[5] ~ Normal(0, exp(log_sigma))          D = 6, x = (log_sigma, x_0 .. x_4)
 * with the reference's raw C logp callback signature (src/pymc.rs:23-29).  The reference's frozen docs hold the final step sizes,
 * last-draw gradient counts and divergence counts of 6 chains of nuts-rs on this model (tests/golden/reference_doc_step_sizes.json:
 * "funnel_diag"): tests/test_oracle_reference_pins.py runs the oracle on it.
 */
#include <math.h>
#include <stdint.h>

int64_t funnel_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    if (dim != 6) return -1;
    const double ls = x[0], inv_var = exp(-2.0 * ls);
    double ss = 0.0;
    for (int i = 1; i < 6; ++i) {
        ss += x[i] * x[i];
        grad[i] = -x[i] * inv_var;
    }
    /* -ls^2/2 - 5 ls - sum x^2 / (2 sigma^2)  (constants dropped) */
    *logp = -0.5 * ls * ls - 5.0 * ls - 0.5 * ss * inv_var;
    grad[0] = -ls - 5.0 + ss * inv_var;
    return 0;
}

/* docs/sample-stats.qmd:141-145: x ~ Normal(0, 1); y ~ Normal(x, 0.01); z[100] ~ Normal(y, 1).  D = 102, position (x, y, z_0 .. z_99). */
int64_t correlated_102d_logp(uint64_t dim, const double* q, double* grad, double* logp, void* user) {
    (void)user;
    if (dim != 102) return -1;
    const double x = q[0], y = q[1], r = (y - x) * 1.0e4;       /* (y - x) / 0.01^2 */
    double lp = -0.5 * x * x - 0.5 * (y - x) * r, sz = 0.0;
    for (int i = 0; i < 100; ++i) {
        const double d = q[2 + i] - y;
        lp -= 0.5 * d * d;
        sz += d;
        grad[2 + i] = -d;
    }
    grad[0] = -x + r;
    grad[1] = -r + sz;
    *logp = lp;
    return 0;
}

/* docs/nf-adapt.qmd:60-64: log_sigma ~ Normal(0, 1); x[100] ~ Normal(0, sigma = exp(log_sigma / 2)).  D = 101, position (log_sigma, x_0 .. x_99):
 * logp = -ls^2/2 - 50 ls - sum x^2 exp(-ls) / 2 (constants dropped).  The reference's frozen docs hold, for 6 chains of nuts-rs under the default
 * adaptation (seed 1): the TOTAL number of gradient evaluations incl. warm-up (124 219), the minimum bulk ESS (31.46) and the progress table. */
int64_t funnel_101d_logp(uint64_t dim, const double* x, double* grad, double* logp, void* user) {
    (void)user;
    if (dim != 101) return -1;
    const double ls = x[0], inv_var = exp(-ls);
    double ss = 0.0;
    for (int i = 1; i < 101; ++i) {
        ss += x[i] * x[i];
        grad[i] = -x[i] * inv_var;
    }
    *logp = -0.5 * ls * ls - 50.0 * ls - 0.5 * ss * inv_var;
    grad[0] = -ls - 50.0 + 0.5 * ss * inv_var;
    return 0;
}
