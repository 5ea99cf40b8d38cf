// wfpt_device_math.h -- gfx950 device arithmetic for the wavefront kernels.
//
// The WGSL built-ins the reference's shaders call (sqrt, sin, cos, pow, min, max, normalize, dot,
// f32(u32)) have backend-defined precision, so the build fixes one definition of each, made only of
// IEEE-754 binary32 add/sub/mul/div/sqrt/fma and integer bit operations in a fixed order:
//   * hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt gives correctly rounded `/` and sqrt;
//   * this translation unit is compiled with -ffp-contract=off, the only fused operations are the
//     explicit __builtin_fmaf calls below (v_fma_f32);
//   * fp32 denormals are not flushed on gfx950 (and no result below is ever denormal).
// The CPU oracle carries its own independent statement of the same definitions; the GPU tests compare
// the two bit for bit (wfpt_selftest_math).
//
// Citations: gr = gpu_wavefront_pt/shaders/generate_rays.wgsl, sh = .../shade.wgsl.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wfpt {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_sqrtf(x); }
// minNum / maxNum (v_min_f32 / v_max_f32): a NaN operand yields the other operand
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }

// ---------------- integer RNG (gr:138-181, identical in sh:228-266) ----------------
__device__ __forceinline__ uint32_t jenkins_hash(uint32_t x) { // gr:173-181
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}
__device__ __forceinline__ uint32_t init_rng(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame) { // gr:138-141
    return jenkins_hash((px + py * res_x) ^ jenkins_hash(frame));
}
__device__ __forceinline__ uint32_t rng_next_int(uint32_t &state) { // gr:146-153, PCG-RXS-M-XS-32
    const uint32_t s = state * 747796405u + 2891336453u;
    state = s;
    const uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (word >> 22) ^ word;
}
__device__ __forceinline__ float u32_to_unit_float(uint32_t x) { // gr:133-136: f32(x) * 2^-32, RTNE
    return static_cast<float>(x) * 2.3283064365387e-10f;
}
__device__ __forceinline__ float rng_next_float(uint32_t &state) { return u32_to_unit_float(rng_next_int(state)); }
// gr:155-171: the skip-ahead as written (only the final `delta == 1` step accumulates)
__device__ __forceinline__ uint32_t advance(uint32_t state, uint32_t advance_by) {
    uint32_t acc_mult = 1u, acc_plus = 0u, cur_mult = 747796405u, cur_plus = 2891336453u;
    for (uint32_t delta = advance_by; delta > 0; delta >>= 1) {
        if (delta == 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1u) * cur_plus;
        cur_mult *= cur_mult;
    }
    return state * acc_mult + acc_plus;
}

// ---------------- sin / cos ----------------
// Quadrant reduction k = rint(x * 2/pi), r = x - k*pi/2 with pi/2 split in three (Cody-Waite, fma),
// Cephes single-precision polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_(float x, float &sin_out, float &cos_out) {
    const float k = __builtin_rintf(x * 0.63661975f);
    float r = fma_(-k, 1.5703125f, x);
    r = fma_(-k, 4.837512969970703125e-4f, r);
    r = fma_(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = fma_(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fma_(z, ps, -1.6666654611e-1f);
    const float s = fma_(r * z, ps, r);
    float pc = fma_(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fma_(z, pc, 4.166664568298827e-2f);
    const float c = fma_(z * z, pc, fma_(z, -0.5f, 1.0f));
    const int q = static_cast<int>(k) & 3;
    float sq = (q & 1) ? c : s;
    float cq = (q & 1) ? s : c;
    sq = (q & 2) ? -sq : sq;
    cq = ((q + 1) & 2) ? -cq : cq;
    sin_out = sq;
    cos_out = cq;
}

// ---------------- pow(x, y) = exp2(y * log2(x)) ----------------
__device__ __forceinline__ float log2_pos(float x) { // normal x > 0
    const uint32_t bits = __float_as_uint(x);
    int e = static_cast<int>(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = fma_(f, 7.0376836292e-2f, -1.1514610310e-1f);
    p = fma_(f, p, 1.1676998740e-1f);
    p = fma_(f, p, -1.2420140846e-1f);
    p = fma_(f, p, 1.4249322787e-1f);
    p = fma_(f, p, -1.6668057665e-1f);
    p = fma_(f, p, 2.0000714765e-1f);
    p = fma_(f, p, -2.4999993993e-1f);
    p = fma_(f, p, 3.3333331174e-1f);
    const float ln_m = fma_(f * z, p, fma_(z, -0.5f, f));
    return fma_(ln_m, 1.44269504f, static_cast<float>(e));
}
__device__ __forceinline__ float exp2_(float x) {
    if (x != x) return x;
    if (x < -125.0f) return 0.0f;
    if (x > 127.0f) return __uint_as_float(0x7f800000u);
    const float n = __builtin_rintf(x);
    const float g = x - n;
    float p = fma_(g, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fma_(g, p, 9.618437357674640e-3f);
    p = fma_(g, p, 5.550332471162809e-2f);
    p = fma_(g, p, 2.402264791363012e-1f);
    p = fma_(g, p, 6.931472028550421e-1f);
    p = fma_(g, p, 1.0f);
    return p * __uint_as_float(static_cast<uint32_t>(static_cast<int>(n) + 127) << 23);
}
__device__ __forceinline__ float pow_(float x, float y) { // sh:120 (y = 0.33333), sh:161 (y = 5)
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return __uint_as_float(0x7fc00000u);
    return exp2_(y * log2_pos(x));
}

// ---------------- small vectors: fixed association order, no contraction ----------------
struct float3_ { float x, y, z; };
__device__ __forceinline__ float dot3(float3_ a, float3_ b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// WGSL normalize(v): the built-in's accuracy is that of v / length(v) (2.5 ULP per component); the definition fixed here since round 4
// is v * (1 / length(v)): one IEEE division and three products instead of three divisions (a correctly rounded division is ~11
// instructions, most of them half rate), at most 1.5 ULP from the quotient form. Rounds 1-3 divided each component.
__device__ __forceinline__ float3_ normalize3(float3_ a) {
    const float inv = 1.0f / sqrt_(dot3(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

} // namespace wfpt
