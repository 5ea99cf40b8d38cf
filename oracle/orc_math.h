/*
 * orc_math.h -- ORACLE (test infrastructure, never shipped, never linked into the product).
 *
 * Scalar fp32 math + RNG used by the CPU restatement of the reference's WGSL kernels.
 *
 * The reference's shaders call WGSL built-ins (sqrt, sin, cos, pow, normalize, length, dot, min, max,
 * f32(u32)) whose precision is backend-defined (wgpu/naga 22.1.0, Cargo.lock:937-939,2033-2035), so the
 * reference pins no bit pattern for them ("parity unpinned" for those built-ins).
 * This file fixes ONE deterministic definition of each: only IEEE-754 binary32 +,-,*,/,sqrt, fma and
 * integer bit operations, in a fixed association order. Every result is therefore a legal WGSL
 * execution, and the HIP kernels (which carry their own, independently written, device copy of the
 * same specification in wavefront_path_tracer_amd/csrc/wfpt_device_math.h) must reproduce it bit for bit.
 *
 * Compile with -ffp-contract=off: the only fused operations are the explicit orc_fma() calls.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <stdint.h>
#include <string.h>

static inline float    orc_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2u(float f)    { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float orc_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float orc_sqrt(float x)                  { return __builtin_sqrtf(x); }
/* minNum/maxNum: a NaN operand yields the other operand (WGSL leaves this case open). */
static inline float orc_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
static inline float orc_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b > a ? b : a)); }

/* ---- integer RNG: generate_rays.wgsl:138-181 (identical copies in shade.wgsl:228-266) ---- */

/* generate_rays.wgsl:173-181 */
static inline uint32_t orc_jenkins_hash(uint32_t x) {
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

/* generate_rays.wgsl:138-141: dot(pixel, (1, resolution.x)) ^ jenkins(frame), hashed again */
static inline uint32_t orc_init_rng(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame) {
    uint32_t seed = (px * 1u + py * res_x) ^ orc_jenkins_hash(frame);
    return orc_jenkins_hash(seed);
}

/* generate_rays.wgsl:146-153: PCG-RXS-M-XS-32 */
static inline uint32_t orc_rng_next_int(uint32_t *state) {
    uint32_t new_state = *state * 747796405u + 2891336453u;
    *state = new_state;
    uint32_t word = ((new_state >> ((new_state >> 28) + 4u)) ^ new_state) * 277803737u;
    return (word >> 22) ^ word;
}

/* generate_rays.wgsl:133-136: f32(x) * 2^-32; u32->f32 is round-to-nearest-even, range [0,1] inclusive */
static inline float orc_rng_next_float(uint32_t *state) {
    uint32_t x = orc_rng_next_int(state);
    return (float)x * 2.3283064365387e-10f;
}

/* generate_rays.wgsl:155-171: LCG skip-ahead AS WRITTEN (accumulates only when delta == 1, not delta & 1) */
static inline void orc_advance(uint32_t *state, uint32_t advance_by) {
    uint32_t acc_mult = 1u, acc_plus = 0u;
    uint32_t cur_mult = 747796405u, cur_plus = 2891336453u;
    uint32_t delta = advance_by;
    while (delta > 0) {
        if (delta == 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1u) * cur_plus;
        cur_mult *= cur_mult;
        delta = delta >> 1;
    }
    *state = *state * acc_mult + acc_plus;
}

/* ---- transcendental definitions (the part WGSL leaves to the backend) ---- */

/* sin and cos of x (|x| < ~1e4; callers pass 2*pi*u, u in [0,1]).
 * k = rint(x * 2/pi); r = x - k*pi/2 by a three-constant Cody-Waite reduction with fma;
 * Cephes single-precision minimax polynomials on [-pi/4, pi/4]; quadrant select. */
static inline void orc_sincos(float x, float *s_out, float *c_out) {
    float k = __builtin_rintf(x * 0.63661975f);
    float r = orc_fma(-k, 1.5703125f, x);
    r = orc_fma(-k, 4.837512969970703125e-4f, r);
    r = orc_fma(-k, 7.54978995489188216e-8f, r);
    float z = r * r;
    /* sin(r) = r + r*z*(S1 + z*(S2 + z*S3)) */
    float ps = orc_fma(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = orc_fma(z, ps, -1.6666654611e-1f);
    float s = orc_fma(r * z, ps, r);
    /* cos(r) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    float pc = orc_fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = orc_fma(z, pc, 4.166664568298827e-2f);
    float c = orc_fma(z * z, pc, orc_fma(z, -0.5f, 1.0f));
    int q = (int)k & 3;
    float sq = (q & 1) ? c : s;
    float cq = (q & 1) ? s : c;
    if (q & 2) sq = -sq;
    if ((q + 1) & 2) cq = -cq;
    *s_out = sq;
    *c_out = cq;
}

/* log2(x) for normal x > 0: x = 2^e * m, m in [sqrt(1/2), sqrt(2)); Cephes logf polynomial in f = m-1;
 * result = e + log2(e)*ln(m) assembled with fma. */
static inline float orc_log2_pos(float x) {
    uint32_t ux = orc_f2u(x);
    int e = (int)(ux >> 23) - 127;
    float m = orc_u2f((ux & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = orc_fma(f, 7.0376836292e-2f, -1.1514610310e-1f);
    p = orc_fma(f, p, 1.1676998740e-1f);
    p = orc_fma(f, p, -1.2420140846e-1f);
    p = orc_fma(f, p, 1.4249322787e-1f);
    p = orc_fma(f, p, -1.6668057665e-1f);
    p = orc_fma(f, p, 2.0000714765e-1f);
    p = orc_fma(f, p, -2.4999993993e-1f);
    p = orc_fma(f, p, 3.3333331174e-1f);
    float ln_m = orc_fma(f * z, p, orc_fma(z, -0.5f, f)); /* f - z/2 + f*z*p */
    return orc_fma(ln_m, 1.44269504f, (float)e);
}

/* 2^x: n = rint(x), g = x - n in [-1/2, 1/2]; degree-6 polynomial for 2^g; scale by 2^n through the
 * exponent bits. x < -125 returns +0 (so no result is ever denormal), x > 127 returns +inf. */
static inline float orc_exp2(float x) {
    if (x != x) return x;
    if (x < -125.0f) return 0.0f;
    if (x > 127.0f) return orc_u2f(0x7f800000u);
    float n = __builtin_rintf(x);
    float g = x - n;
    float p = orc_fma(g, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = orc_fma(g, p, 9.618437357674640e-3f);
    p = orc_fma(g, p, 5.550332471162809e-2f);
    p = orc_fma(g, p, 2.402264791363012e-1f);
    p = orc_fma(g, p, 6.931472028550421e-1f);
    p = orc_fma(g, p, 1.0f);
    int ni = (int)n;
    return p * orc_u2f((uint32_t)(ni + 127) << 23);
}

/* pow(x, y) = exp2(y * log2(x)) for x > 0; pow(0, y>0) = 0; x < 0 gives NaN (WGSL: undefined there).
 * Used by shade.wgsl:120 (pow(u, 0.33333)) and shade.wgsl:161 (pow(1 - cos, 5)). */
static inline float orc_pow(float x, float y) {
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return orc_u2f(0x7fc00000u);
    return orc_exp2(y * orc_log2_pos(x));
}

#endif /* ORC_MATH_H */
