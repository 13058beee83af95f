/*
 * radon_host.c — test fixture (written for this repo): the radon density of tests/fixtures/radon_device.hip as plain C behind
 * the reference's raw logp callback signature (src/pymc.rs:23-29), so that the CPU oracle can sample the SAME model the engine
 * samples through its batched device callback (BASELINE.json config 3).  Same formulas, sequential summation: results agree
 * with the HIP kernel to rounding (the kernel sums across a wavefront), not bit for bit.
 *
 * Unconstrained vector (D = 2n + 3): [intercept, raw(n-1), log sd, floor_effect, craw(n-1), log csd, log sigma].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n_counties, n_obs;
    int* county;
    double *floor_, *y;
} radon_host;

void* radon_host_create(int n_counties, int n_obs, const int* county, const double* floor_, const double* y) {
    radon_host* h = (radon_host*)calloc(1, sizeof(radon_host));
    h->n_counties = n_counties; h->n_obs = n_obs;
    h->county = (int*)malloc(n_obs * sizeof(int)); memcpy(h->county, county, n_obs * sizeof(int));
    h->floor_ = (double*)malloc(n_obs * sizeof(double)); memcpy(h->floor_, floor_, n_obs * sizeof(double));
    h->y = (double*)malloc(n_obs * sizeof(double)); memcpy(h->y, y, n_obs * sizeof(double));
    return h;
}
void radon_host_free(void* p) {
    radon_host* h = (radon_host*)p;
    if (!h) return;
    free(h->county); free(h->floor_); free(h->y); free(h);
}

int radon_host_logp(uint64_t dim, const double* x, double* g, double* logp, void* user) {
    const radon_host* d = (const radon_host*)user;
    const int n = d->n_counties, n_obs = d->n_obs;
    if (dim != (uint64_t)(2 * n + 3)) return -1;
    const int o_raw = 1, o_lsd = n, o_floor = n + 1, o_craw = n + 2, o_lcsd = 2 * n + 1, o_lsig = 2 * n + 2;
    const double intercept = x[0], fe = x[o_floor], lsd = x[o_lsd], lcsd = x[o_lcsd], lsig = x[o_lsig];
    const double sd = exp(lsd), csd = exp(lcsd), sig = exp(lsig), inv_sig = 1.0 / sig;
    const double c1 = 1.0 / (sqrt((double)n) + n), c2 = 1.0 / sqrt((double)n);
    double eff[128], cfe[128], ge[128], gc[128];
    if (n > 128) return -1;
    double s_raw = 0.0, s_craw = 0.0, ss = 0.0;
    for (int j = 0; j < n - 1; ++j) { const double a = x[o_raw + j], b = x[o_craw + j]; s_raw += a; s_craw += b; ss += a * a + b * b; }
    for (int j = 0; j < n; ++j) {
        const double e = (j < n - 1) ? x[o_raw + j] - s_raw * c1 : -s_raw * c2;
        const double ce = (j < n - 1) ? x[o_craw + j] - s_craw * c1 : -s_craw * c2;
        eff[j] = e * sd; cfe[j] = ce * csd; ge[j] = 0.0; gc[j] = 0.0;
    }
    double rr = 0.0, sw = 0.0, swf = 0.0;
    for (int o = 0; o < n_obs; ++o) {
        const int cty = d->county[o];
        const double fl = d->floor_[o];
        const double mu = intercept + eff[cty] + fl * (fe + cfe[cty]);
        const double r = (d->y[o] - mu) * inv_sig;
        const double wo = r * inv_sig;
        rr += r * r; sw += wo; swf += wo * fl;
        ge[cty] += wo; gc[cty] += wo * fl;
    }
    double dot_e = 0.0, dot_c = 0.0, su_e = 0.0, su_c = 0.0;
    for (int j = 0; j < n; ++j) {
        dot_e += (eff[j] / sd) * ge[j]; dot_c += (cfe[j] / csd) * gc[j];
        if (j < n - 1) { su_e += ge[j] * sd; su_c += gc[j] * csd; }
    }
    const double gl_e = ge[n - 1] * sd, gl_c = gc[n - 1] * csd;
    for (int j = 0; j < n - 1; ++j) {
        g[o_raw + j] = (ge[j] * sd - (c1 * su_e + c2 * gl_e)) - x[o_raw + j];
        g[o_craw + j] = (gc[j] * csd - (c1 * su_c + c2 * gl_c)) - x[o_craw + j];
    }
    g[0] = -0.01 * intercept + sw;
    g[o_floor] = -0.25 * fe + swf;
    g[o_lsd] = 1.0 - sd * sd + sd * dot_e;
    g[o_lcsd] = 1.0 - csd * csd + csd * dot_c;
    g[o_lsig] = 1.0 - sig * sig / 2.25 + rr - n_obs;
    *logp = -0.005 * intercept * intercept - 0.125 * fe * fe - 0.5 * ss - 0.5 * sd * sd + lsd - 0.5 * csd * csd + lcsd
            - (0.5 / 2.25) * sig * sig + lsig - 0.5 * rr - n_obs * lsig;
    for (uint64_t i = 0; i < dim; ++i) if (!isfinite(g[i])) return 3;
    if (!isfinite(*logp)) return 4;
    return 0;
}
