/*
 * nphip_spec.h — the deterministic-numerics contract of the nutpie-hip engine.
 *
 * nuts-rs (the crate behind nutpie, Cargo.toml:24) draws its randomness from
 * ChaCha8 + ziggurat normals and evaluates exp/ln through the platform libm.
 * Neither is reproducible on a GPU bit-for-bit, and the crate source is not in
 * the reference tree, so this engine DEFINES its own result contract:
 *
 *   1. a counter-based RNG (Philox4x32-10) keyed by (seed, chain, draw, purpose,
 *      index) — results do not depend on GPU count, chain sharding or on how
 *      chains interleave in time;
 *   2. exp / log / log1p / sin,cos(2*pi*u) built from +,-,*,/,fma and sqrt only
 *      (all correctly rounded IEEE-754 binary64 operations on both x86-64 and
 *      gfx950), so host and device produce identical bits;
 *   3. a fixed summation order for every length-D reduction (see nphip_dot
 *      geometry below).
 *
 * Everything in this file is usable from host (g++) and device (hipcc) code.
 * Both sides MUST be compiled with -ffp-contract=off: fused multiply-adds happen
 * only where this file (or the kernels) call fma() explicitly.
 *
 * The CPU oracle (oracle/) restates these functions independently; the tests
 * compare the two implementations bit-for-bit.
 */
#ifndef NPHIP_SPEC_H
#define NPHIP_SPEC_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define NPHIP_HD __host__ __device__ __forceinline__
#else
#define NPHIP_HD static inline
#endif

/* ------------------------------------------------------------------------- */
/* RNG streams                                                               */
/* ------------------------------------------------------------------------- */

/* Stream purposes (low byte of counter word 3). */
enum {
    NPHIP_RNG_MOMENTUM = 1,    /* c0 = pair index j -> elements 2j,2j+1 ; c2 = draw            */
    NPHIP_RNG_DIRECTION = 2,   /* c0 = tree depth before the doubling   ; c2 = draw            */
    NPHIP_RNG_MERGE = 3,       /* c0 = leaf index in doubling; c3 |= depth<<8 | (level>>1)<<16; the merge at
                                `level` uses words 2*(level&1), 2*(level&1)+1 of the block */
    NPHIP_RNG_INIT = 4,        /* c0 = pair index ; c2 = init attempt                          */
    NPHIP_RNG_SS_MOMENTUM = 5, /* c0 = pair index ; c2 = search id (0xffffffff at chain start, */
                               /*                    else the draw index that triggered it)   */
    NPHIP_RNG_JITTER = 6,      /* c2 = draw                                                    */
    NPHIP_RNG_EXPAND = 7       /* word 0 = seed of the chain's generator for the expand step (BridgeStan's bs_rng, src/stan.rs:787-788) */
};

typedef struct { uint32_t v[4]; } nphip_u32x4;

NPHIP_HD uint32_t nphip_mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

/* Philox4x32-10 (Salmon et al., SC'11).  key = (seed lo, seed hi). */
NPHIP_HD nphip_u32x4 nphip_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = nphip_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = nphip_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    nphip_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

/* 53-bit uniform in the open interval (0,1): (k + 0.5) * 2^-53, k = top 53 bits. */
NPHIP_HD double nphip_u01(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | (uint64_t)lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}

/* ------------------------------------------------------------------------- */
/* Deterministic elementary functions                                        */
/* ------------------------------------------------------------------------- */

NPHIP_HD double nphip_bits2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
NPHIP_HD uint64_t nphip_d2bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* 2^k for -1022 <= k <= 1023 */
NPHIP_HD double nphip_pow2i(int k) { return nphip_bits2d((uint64_t)(k + 1023) << 52); }

/* exp(x): k = rint(x/ln2), r = x - k ln2 (Cody-Waite, fma), Taylor degree 13 in
 * Horner/fma form, scaled by 2^k in two exact steps.  <= 2 ulp. */
/* NPHIP_K(c): a literal of the polynomial kernels.  On the device it is pinned in an SGPR pair at its point of
 * use (an empty asm the optimiser cannot move): otherwise the ~40 coefficients are hoisted out of the sampler's
 * main loop into VGPRs and spilled to scratch, and every Horner step waits on a scratch reload.  Same value,
 * same operation order on both sides. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double nphip_kdev(double c) { asm volatile("" : "+s"(c)); return c; }
#define NPHIP_K(c) nphip_kdev(c)
#else
#define NPHIP_K(c) (c)
#endif

/* (p, k) with exp(x) ~= p * 2^k, p in about [0.70, 1.42]: the reduction and the polynomial of nphip_exp, before
 * the scaling.  |x| <= 1e9 so that k * ln2_hi is exact. */
NPHIP_HD void nphip_exp_parts(double x, double* p_out, double* k_out) {
    double k = rint(x * NPHIP_K(0x1.71547652b82fep+0));
    double r = fma(-k, NPHIP_K(0x1.62e42fee00000p-1), x);
    r = fma(-k, NPHIP_K(0x1.a39ef35793c76p-33), r);
    double p = NPHIP_K(1.6059043836821613e-10);      /* 1/13! */
    p = fma(p, r, NPHIP_K(2.08767569878681e-09));    /* 1/12! */
    p = fma(p, r, NPHIP_K(2.505210838544172e-08));   /* 1/11! */
    p = fma(p, r, NPHIP_K(2.755731922398589e-07));   /* 1/10! */
    p = fma(p, r, NPHIP_K(2.7557319223985893e-06));  /* 1/9!  */
    p = fma(p, r, NPHIP_K(2.48015873015873e-05));    /* 1/8!  */
    p = fma(p, r, NPHIP_K(0.0001984126984126984));   /* 1/7!  */
    p = fma(p, r, NPHIP_K(0.001388888888888889));    /* 1/6!  */
    p = fma(p, r, NPHIP_K(0.008333333333333333));    /* 1/5!  */
    p = fma(p, r, NPHIP_K(0.041666666666666664));    /* 1/4!  */
    p = fma(p, r, NPHIP_K(0.16666666666666666));     /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    *p_out = p;
    *k_out = k;
}

/* exp(x) from its parts (x is the original argument: it decides overflow / underflow) */
NPHIP_HD double nphip_exp_scale(double x, double p, double k) {
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    return (p * nphip_pow2i(k1)) * nphip_pow2i(k2);
}

NPHIP_HD double nphip_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    double p, k;
    nphip_exp_parts(x, &p, &k);
    return nphip_exp_scale(x, p, k);
}

/* ---- extended-range tree weights ---------------------------------------------------------------------------
 * nuts-rs carries the multinomial weight of a (sub)tree as `log_size`, merges with logaddexp and accepts the
 * other side's draw with probability exp(log_size_other - log_size_total).  The same weights are carried here
 * as w = m * 2^e (m > 0 a double, e an integer, so nothing overflows for any finite energy error): a leaf's
 * weight exp(-energy_error) is the (p, k) pair above, sums are aligned additions, and the acceptance test is
 * u * w_total < w_other -- no log, exp or division per merge.  Same distribution, different rounding. */
NPHIP_HD void nphip_w_leaf(double neg_energy_error, double* m, int64_t* e) {
    double x = neg_energy_error;
    if (x > 1e9) x = 1e9;
    if (x < -1e9) x = -1e9;
    double k;
    nphip_exp_parts(x, m, &k);
    *e = (int64_t)k;
}
/* m * 2^(e - E) for E >= e; contributions below 2^-1000 of the larger weight are dropped */
NPHIP_HD double nphip_w_rel(double m, int64_t e, int64_t E) {
    int64_t d = E - e;
    if (d >= 1000) return 0.0;
    return m * nphip_pow2i((int)-d);
}
/* (m1, e1) + (m2, e2), operands in this order */
NPHIP_HD void nphip_w_add(double m1, int64_t e1, double m2, int64_t e2, double* m, int64_t* e) {
    int64_t E = e1 >= e2 ? e1 : e2;
    *m = nphip_w_rel(m1, e1, E) + nphip_w_rel(m2, e2, E);
    *e = E;
}

/* log(x): x = 2^e m, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1);
 * log m = 2s + s z R(z), z = s^2, R = sum_{n=1..12} 2/(2n+1) z^(n-1). <= 2 ulp. */
NPHIP_HD double nphip_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    if (x < 0x1.0p-1022) { x *= 0x1.0p54; e = -54; }
    uint64_t b = nphip_d2bits(x);
    e += (int)(b >> 52) - 1023;
    double m = nphip_bits2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m >= 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = NPHIP_K(0.08);                        /* 2/25 */
    R = fma(R, z, NPHIP_K(0.08695652173913043));     /* 2/23 */
    R = fma(R, z, NPHIP_K(0.09523809523809523));     /* 2/21 */
    R = fma(R, z, NPHIP_K(0.10526315789473684));     /* 2/19 */
    R = fma(R, z, NPHIP_K(0.11764705882352941));     /* 2/17 */
    R = fma(R, z, NPHIP_K(0.13333333333333333));     /* 2/15 */
    R = fma(R, z, NPHIP_K(0.15384615384615385));     /* 2/13 */
    R = fma(R, z, NPHIP_K(0.18181818181818182));     /* 2/11 */
    R = fma(R, z, NPHIP_K(0.2222222222222222));      /* 2/9  */
    R = fma(R, z, NPHIP_K(0.2857142857142857));      /* 2/7  */
    R = fma(R, z, NPHIP_K(0.4));                     /* 2/5  */
    R = fma(R, z, NPHIP_K(0.6666666666666666));      /* 2/3  */
    double logm = fma(s * z, R, 2.0 * s);
    double de = (double)e;
    return fma(de, NPHIP_K(0x1.62e42fee00000p-1), fma(de, NPHIP_K(0x1.a39ef35793c76p-33), logm));
}

/* log(1+y) for y >= 0 (HP-15C trick). */
NPHIP_HD double nphip_log1p(double y) {
    double u = 1.0 + y;
    if (u == 1.0) return y;
    return nphip_log(u) * y / (u - 1.0);
}

/* logaddexp as nuts-rs computes tree weights (restated from the crate's
 * math helpers). */
NPHIP_HD double nphip_logaddexp(double a, double b) {
    if (a == b) return a + 0x1.62e42fefa39efp-1;
    double diff = a - b;
    if (diff > 0.0) return a + nphip_log1p(nphip_exp(-diff));
    if (diff < 0.0) return b + nphip_log1p(nphip_exp(diff));
    return diff; /* NaN */
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1): exact octant reduction, Taylor on [0, pi/4]. */
NPHIP_HD void nphip_sincos2pi(double u, double* sn, double* cs) {
    double t = 4.0 * u;
    double qd = floor(t);
    double f = t - qd;
    int swap = f > 0.5;
    if (swap) f = 1.0 - f;
    double x = f * 0x1.921fb54442d18p+0;
    double x2 = x * x;
    double S = 2.8114572543455206e-15;       /*  1/17! */
    S = fma(S, x2, -7.647163731819816e-13);  /* -1/15! */
    S = fma(S, x2, 1.6059043836821613e-10);  /*  1/13! */
    S = fma(S, x2, -2.505210838544172e-08);  /* -1/11! */
    S = fma(S, x2, 2.7557319223985893e-06);  /*  1/9!  */
    S = fma(S, x2, -0.0001984126984126984);  /* -1/7!  */
    S = fma(S, x2, 0.008333333333333333);    /*  1/5!  */
    S = fma(S, x2, -0.16666666666666666);    /* -1/3!  */
    double s = fma(x * x2, S, x);
    double C = -1.5619206968586225e-16;      /* -1/18! */
    C = fma(C, x2, 4.779477332387385e-14);   /*  1/16! */
    C = fma(C, x2, -1.1470745597729725e-11); /* -1/14! */
    C = fma(C, x2, 2.08767569878681e-09);    /*  1/12! */
    C = fma(C, x2, -2.755731922398589e-07);  /* -1/10! */
    C = fma(C, x2, 2.48015873015873e-05);    /*  1/8!  */
    C = fma(C, x2, -0.001388888888888889);   /* -1/6!  */
    C = fma(C, x2, 0.041666666666666664);    /*  1/4!  */
    C = fma(C, x2, -0.5);
    double c = fma(x2, C, 1.0);
    if (swap) { double tmp = s; s = c; c = tmp; }
    int q = (int)qd;
    if (q == 0) { *sn = s; *cs = c; }
    else if (q == 1) { *sn = c; *cs = -s; }
    else if (q == 2) { *sn = -s; *cs = -c; }
    else { *sn = -c; *cs = s; }
}

/* Two standard normals from one Philox block (Box-Muller). */
NPHIP_HD void nphip_normal_pair(nphip_u32x4 r, double* z0, double* z1) {
    double u1 = nphip_u01(r.v[0], r.v[1]);
    double u2 = nphip_u01(r.v[2], r.v[3]);
    double rad = sqrt(-2.0 * nphip_log(u1));
    double sn, cs;
    nphip_sincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

/* ------------------------------------------------------------------------- */
/* Reduction order ("geometry")                                              */
/* ------------------------------------------------------------------------- */
/*
 * A chain is processed by W wavefronts of 64 lanes; each lane owns pairs of
 * consecutive elements.  Element i lives in chunk c = i / 128; chunk c belongs
 * to wave (c mod W); inside the chunk, lane l = (i mod 128) / 2 holds component
 * i mod 2.  A dot product sum_i x_i*y_i is defined as:
 *   - per (wave, lane, component) accumulator, acc = fma(x_i, y_i, acc) over the
 *     owned elements in increasing i, starting from +0.0;
 *   - lane value = acc[component 0] + acc[component 1];
 *   - five butterfly stages v = v + v_partner with partner(l) = l^1, l^2, (l&~7)|(7-(l&7)),
 *     (l&~15)|(15-(l&15)), l^16  (the cheap DPP patterns quad_perm / row_half_mirror / row_mirror of
 *     gfx950 plus one ds_swizzle), then wave total = v[lane 0] + v[lane 32];
 *   - wave totals summed in wave order: ((w0 + w1) + w2) + ...
 * W is reported by the engine (nphip_sampler_waves_per_chain) and is an input
 * of the oracle.
 */
#define NPHIP_LANES 64
#define NPHIP_CHUNK 128

/* ------------------------------------------------------------------------- */
/* Dense gradient (nphip_model_dense_gaussian)                               */
/* ------------------------------------------------------------------------- */
/*
 * logp(x) = -1/2 (x - mu)' P (x - mu), P symmetric [D][D].  With z = x - mu (one subtraction per element):
 *   acc_j = sum_k z_k * P[j][k]  as ONE chain of fused multiply-adds per output j, starting from +0.0, in the order
 *           for k0 = 0, 16, 32, ...:  for s = 0..3:  for t = 0..3:   k = k0 + 4 t + s        (k >= D is skipped)
 *   g_j   = -acc_j
 *   logp  = 1/2 * dot(z, g)   in the reduction geometry above (W waves per chain)
 * The order of k is the order in which the engine's fp64 matrix-core tile (nutpie_amd/csrc/dense_tile.h) consumes a row:
 * lane (row, t) of an operand fragment holds the four consecutive elements k0 + 4 t .. k0 + 4 t + 3, v_mfma_f64_16x16x4_f64
 * number s of a k0-step takes element s of every lane, and the matrix core adds the four products of one instruction to the
 * accumulator as fused multiply-adds in the order of t (established on the device: tests/test_gpu_dense.py compares the GEMM
 * with a std::fma chain bit for bit).  Which workgroup, wave or launch computes an element does not enter: every output has
 * exactly one accumulator.
 */
#define NPHIP_DENSE_KSTEP 16

#endif /* NPHIP_SPEC_H */
