// tools/microbench_valu.hip -- hipcc --offload-arch=gfx950 -O3 -o build/valu tools/microbench_valu.hip && build/valu
//
// What does one SIMD of gfx950 sustain, in SHADER CYCLES per wave64 instruction, for each instruction class of the
// traversal's hot loop, with 1 .. 8 waves per SIMD? Round 2's version timed short launches with hipEvents and reported
// "cycles at the nominal 2.4 GHz"; it could not tell a 2-cycle pipe at a throttled clock from a 3-cycle pipe at full
// clock. This one measures inside the kernel:
//   cycles = delta s_memtime (shader clock ticks) around a loop of 64 INDEPENDENT instructions per trip (16 registers x 4),
//   clock  = delta s_memtime / delta s_memrealtime x 100 MHz (s_memrealtime = wall_clock64(), a constant 100 MHz counter),
// after >= 0.7 s of back-to-back launches of the same kernel so that the chip sits in the DVFS state the load gives it
// (MI355X_MICROARCH.md "DVFS give-back" item 6). Reported per class: cycles per instruction per SIMD (= wave cycles /
// (waves per SIMD x instructions)), the clock held, and ns per instruction per SIMD (what a kernel's wall time sees).
// Every instruction is inline asm on registers: the compiler can neither fuse, pack nor drop any of them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cycles, real, r0, r1; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2))); // a 128-bit VGPR tuple inline asm can name

enum Kind { VISIT_ASM, VISIT_ASM_PK, FMA, MINMAX, MAX3, CNDMASK, CMP, CMP_CND, RCP, SQRT, ADDU, LSHLADD, FMA_DEP, FMA_SALU, FMA_DSREAD, VISIT_OLD, VISIT_NEW,
            MULF, ADDF, ANDB, LSHL, BFE, CVTUB, ADDC, MOV, MED3, ANDOR, CNDVCC, MINE64, PKFMA, FMAMIX, PERM, MINU, MULLO, MULHI, MAD64, DIVSCALE, DIVFMAS, DIVFIXUP, EXPF, LOGF, READLANE, WRITELANE, LSHLADD64, PKMUL, CVTFU, S_ADD, S_AND64, S_CSEL, S_LOADHIT, RFL_CHAIN, FMA_SALU16, FMA_SALU64, FMA_SAND32, MINMAX_SALU32, VISIT_MIX, N_KINDS };

// one instruction of the class on register x (a, b: loop-invariant VGPRs; m: an SGPR pair holding a lane mask)
#define I_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b))
#define I_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_CND(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m))
#define I_CMP(x) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(x), "v"(a))
#define I_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define I_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x))
#define I_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(a))
#define I_MULF(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_ADDF(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_ANDB(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x))
#define I_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x))
#define I_CVTUB(x) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x))
#define I_ADDC(x) asm volatile("v_addc_co_u32 %0, %2, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m))
#define I_MOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a))
#define I_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_CNDVCC(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : )
#define I_MINE64(x) asm volatile("v_min_f32_e64 %0, %0, |%1|" : "+v"(x) : "v"(a))
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(ya), "v"(yb))
#define I_FMAMIX(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(a), "v"(b))
#define I_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a))
#define I_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[k]) : "v"(a), "v"(b) : "vcc")
#define I_DIVSCALE(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(a) : "vcc")
#define I_DIVFMAS(x) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b) : "vcc")
#define I_DIVFIXUP(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define I_EXPF(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define I_LOGF(x) asm volatile("v_log_f32 %0, %0" : "+v"(x))
#define I_READLANE(x) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc) : "v"(x))
#define I_WRITELANE(x) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(x) : "s"(sc))
#define I_LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(y[k]) : "v"(ya))
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[k]) : "v"(ya))
#define I_CVTFU(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x))
#define I_SALU() asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc")
// scalar classes (round 5): independent instructions on 16 scalar registers
#define I_SADD(k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ss[k]) : : "scc")
#define I_SAND64(k) asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm[k]) : "s"(m) : "scc")
#define I_SCSEL(k) asm volatile("s_cselect_b32 %0, %0, %1" : "+s"(ss[k]) : "s"(sc))
#define REPS16(OP) OP(0); OP(1); OP(2); OP(3); OP(4); OP(5); OP(6); OP(7); OP(8); OP(9); OP(10); OP(11); OP(12); OP(13); OP(14); OP(15)
#define REPS8(OP) OP(0); OP(1); OP(2); OP(3); OP(4); OP(5); OP(6); OP(7)

#define REP16(OP) OP(x[0]); OP(x[1]); OP(x[2]); OP(x[3]); OP(x[4]); OP(x[5]); OP(x[6]); OP(x[7]); OP(x[8]); OP(x[9]); OP(x[10]); OP(x[11]); OP(x[12]); OP(x[13]); OP(x[14]); OP(x[15])

template <int KIND> __global__ __launch_bounds__(256) void kern(Stamp *out, float *sink, uint32_t trips, float fa, float fb, const uint32_t *ktab) {
    __shared__ float4 s_tab[1024]; // 16 KB: ds_read_b128 targets (FMA_DSREAD, VISIT_*)
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) s_tab[i] = make_float4(fa, fb, fa, fb);
    __syncthreads();
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = fa + static_cast<float>((threadIdx.x * 16 + k) & 1023) * 1e-3f;
    float a = fa, b = fb;
    v2f y[8], ya = {fa, fb}, yb = {fb, fa};
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = v2f{x[2 * k], x[2 * k + 1]};
    unsigned long long m = 0x5555aaaa3333ccccull, m2 = 0;
    uint32_t sc = 0;
    uint32_t ss[16];
    unsigned long long sm[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) ss[k] = __builtin_amdgcn_readfirstlane(trips + k);
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[k] = m ^ (0x0101010101010101ull * k);
    uint32_t addr = (threadIdx.x * 37u & 255u) * 64u; // per-lane node pair, 64 B apart
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < trips; ++it) {
        if (KIND == FMA) { REP16(I_FMA); REP16(I_FMA); REP16(I_FMA); REP16(I_FMA); }
        if (KIND == MINMAX) { REP16(I_MIN); REP16(I_MAX); REP16(I_MIN); REP16(I_MAX); }
        if (KIND == MAX3) { REP16(I_MAX3); REP16(I_MIN3); REP16(I_MAX3); REP16(I_MIN3); }
        if (KIND == CNDMASK) { REP16(I_CND); REP16(I_CND); REP16(I_CND); REP16(I_CND); }
        if (KIND == CMP) { REP16(I_CMP); REP16(I_CMP); REP16(I_CMP); REP16(I_CMP); }
        if (KIND == CMP_CND) { // compare -> select pairs, the select reads the mask the compare just wrote (32 + 32)
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(x[k & 15]), "v"(a));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[(k + 8) & 15]) : "v"(b), "s"(m2));
            }
        }
        if (KIND == RCP) { REP16(I_RCP); REP16(I_RCP); REP16(I_RCP); REP16(I_RCP); }
        if (KIND == SQRT) { REP16(I_SQRT); REP16(I_SQRT); REP16(I_SQRT); REP16(I_SQRT); }
        if (KIND == ADDU) { REP16(I_ADDU); REP16(I_ADDU); REP16(I_ADDU); REP16(I_ADDU); }
        if (KIND == LSHLADD) { REP16(I_LSHLADD); REP16(I_LSHLADD); REP16(I_LSHLADD); REP16(I_LSHLADD); }
        if (KIND == MULF) { REP16(I_MULF); REP16(I_MULF); REP16(I_MULF); REP16(I_MULF); }
        if (KIND == ADDF) { REP16(I_ADDF); REP16(I_ADDF); REP16(I_ADDF); REP16(I_ADDF); }
        if (KIND == ANDB) { REP16(I_ANDB); REP16(I_ANDB); REP16(I_ANDB); REP16(I_ANDB); }
        if (KIND == LSHL) { REP16(I_LSHL); REP16(I_LSHL); REP16(I_LSHL); REP16(I_LSHL); }
        if (KIND == BFE) { REP16(I_BFE); REP16(I_BFE); REP16(I_BFE); REP16(I_BFE); }
        if (KIND == CVTUB) { REP16(I_CVTUB); REP16(I_CVTUB); REP16(I_CVTUB); REP16(I_CVTUB); }
        if (KIND == ADDC) { REP16(I_ADDC); REP16(I_ADDC); REP16(I_ADDC); REP16(I_ADDC); }
        if (KIND == MOV) { REP16(I_MOV); REP16(I_MOV); REP16(I_MOV); REP16(I_MOV); }
        if (KIND == MED3) { REP16(I_MED3); REP16(I_MED3); REP16(I_MED3); REP16(I_MED3); }
        if (KIND == ANDOR) { REP16(I_ANDOR); REP16(I_ANDOR); REP16(I_ANDOR); REP16(I_ANDOR); }
        if (KIND == CNDVCC) { REP16(I_CNDVCC); REP16(I_CNDVCC); REP16(I_CNDVCC); REP16(I_CNDVCC); }
        if (KIND == MINE64) { REP16(I_MINE64); REP16(I_MINE64); REP16(I_MINE64); REP16(I_MINE64); }
        if (KIND == PKFMA) { // 64 packed FMAs (two fp32 FMAs each) on 8 register pairs
#pragma unroll
            for (int k = 0; k < 64; ++k) I_PKFMA(k & 7);
        }
        if (KIND == FMAMIX) { REP16(I_FMAMIX); REP16(I_FMAMIX); REP16(I_FMAMIX); REP16(I_FMAMIX); }
        if (KIND == PERM) { REP16(I_PERM); REP16(I_PERM); REP16(I_PERM); REP16(I_PERM); }
        if (KIND == MINU) { REP16(I_MINU); REP16(I_MINU); REP16(I_MINU); REP16(I_MINU); }
        if (KIND == MULLO) { REP16(I_MULLO); REP16(I_MULLO); REP16(I_MULLO); REP16(I_MULLO); }
        if (KIND == MULHI) { REP16(I_MULHI); REP16(I_MULHI); REP16(I_MULHI); REP16(I_MULHI); }
        if (KIND == DIVSCALE) { REP16(I_DIVSCALE); REP16(I_DIVSCALE); REP16(I_DIVSCALE); REP16(I_DIVSCALE); }
        if (KIND == DIVFMAS) { REP16(I_DIVFMAS); REP16(I_DIVFMAS); REP16(I_DIVFMAS); REP16(I_DIVFMAS); }
        if (KIND == DIVFIXUP) { REP16(I_DIVFIXUP); REP16(I_DIVFIXUP); REP16(I_DIVFIXUP); REP16(I_DIVFIXUP); }
        if (KIND == EXPF) { REP16(I_EXPF); REP16(I_EXPF); REP16(I_EXPF); REP16(I_EXPF); }
        if (KIND == LOGF) { REP16(I_LOGF); REP16(I_LOGF); REP16(I_LOGF); REP16(I_LOGF); }
        if (KIND == READLANE) { REP16(I_READLANE); REP16(I_READLANE); REP16(I_READLANE); REP16(I_READLANE); }
        if (KIND == WRITELANE) { REP16(I_WRITELANE); REP16(I_WRITELANE); REP16(I_WRITELANE); REP16(I_WRITELANE); }
        if (KIND == CVTFU) { REP16(I_CVTFU); REP16(I_CVTFU); REP16(I_CVTFU); REP16(I_CVTFU); }
        if (KIND == MAD64 || KIND == LSHLADD64 || KIND == PKMUL) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if (KIND == MAD64) I_MAD64(k & 7);
                if (KIND == LSHLADD64) I_LSHLADD64(k & 7);
                if (KIND == PKMUL) I_PKMUL(k & 7);
            }
        }
        if (KIND == FMA_DEP) { // ONE dependent chain: latency, not throughput
#pragma unroll
            for (int k = 0; k < 64; ++k) I_FMA(x[0]);
        }
        if (KIND == FMA_SALU) { // 64 v_fma with a scalar add after every second one (the traversal has 1 SALU per 3 VALU)
#pragma unroll
            for (int k = 0; k < 64; ++k) { I_FMA(x[k & 15]); if (k & 1) I_SALU(); }
        }
        if (KIND == S_ADD) { REPS16(I_SADD); REPS16(I_SADD); REPS16(I_SADD); REPS16(I_SADD); }
        if (KIND == S_AND64) { REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); REPS8(I_SAND64); }
        if (KIND == S_CSEL) { REPS16(I_SCSEL); REPS16(I_SCSEL); REPS16(I_SCSEL); REPS16(I_SCSEL); }
        if (KIND == S_LOADHIT) { // 64 scalar loads of one cached line, waited for every 16
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                asm volatile("s_load_dword %0, %1, 0x0" : "=s"(ss[k & 15]) : "s"(ktab));
                if ((k & 15) == 15) asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
        if (KIND == RFL_CHAIN) { // v_readfirstlane -> s_add -> v_add with the scalar: the uniform() round trip, one dependent chain, 64 links (3 instructions each)
#pragma unroll
            for (int k = 0; k < 64; ++k)
                asm volatile("v_readfirstlane_b32 %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0" : "+v"(x[0]), "+s"(sc) : : "scc");
        }
        if (KIND == FMA_SALU16) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { I_FMA(x[k & 15]); if ((k & 3) == 3) I_SALU(); }
        }
        if (KIND == FMA_SALU64) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { I_FMA(x[k & 15]); I_SADD(k & 15); }
        }
        if (KIND == FMA_SAND32) { // the traversal's scalar work is mostly 64-bit mask logic
#pragma unroll
            for (int k = 0; k < 64; ++k) { I_FMA(x[k & 15]); if (k & 1) I_SAND64((k >> 1) & 7); }
        }
        if (KIND == MINMAX_SALU32) { // half-rate VALU with a scalar instruction after every second one
#pragma unroll
            for (int k = 0; k < 64; ++k) { if (k & 1) { I_MIN(x[k & 15]); } else { I_MAX(x[k & 15]); } if (k & 1) I_SADD(k & 15); }
        }
        if (KIND == VISIT_MIX) { // the static mix of one inner visit of bounce_kernel (profiles/r03_isa_inner_visit.txt): 25 full-rate + 17 half-rate VALU, 32 SALU
                                  // (two thirds of them 64-bit mask logic), interleaved; independent instructions, no LDS: what the issue logic alone allows
#pragma unroll
            for (int k = 0; k < 42; ++k) {
                if (k % 5 < 3) { I_FMA(x[k & 15]); } else if (k & 1) { I_MIN(x[k & 15]); } else { I_CND(x[k & 15]); }
                if (k < 32) { if (k % 3 == 0) { I_SADD(k & 15); } else { I_SAND64(k & 7); } }
            }
        }
        if (KIND == FMA_DSREAD) { // 64 v_fma + 4 ds_read_b128 per 64 (the node pair of one visit), waited for once per trip
            v4f n0, n1, n2, n3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48"
                         : "=v"(n0), "=v"(n1), "=v"(n2), "=v"(n3) : "v"(addr));
            REP16(I_FMA); REP16(I_FMA); REP16(I_FMA); REP16(I_FMA);
            asm volatile("s_waitcnt lgkmcnt(0)");
            asm volatile("" ::"v"(n0), "v"(n1), "v"(n2), "v"(n3));
            addr = (addr + 64u * 7u) & 16383u & ~63u;
        }
        if (KIND == VISIT_ASM || KIND == VISIT_ASM_PK) {
            // The DESCENDING path of descend_asm (wfpt_kernels.hip), straight-line: node pair from LDS -> two slab tests -> mask logic -> the next
            // pair's address from the chosen child word. VISIT_ASM: the shipped 18 v_fma_f32 (35 vector + 11 scalar instructions). VISIT_ASM_PK: the
            // pair stored interleaved (lcx rcx lcy rcy | lcz rcz lhx rhx | lhy rhy lhz rhz | words), 8 v_pk_fma_f32 + 2 v_fma_f32 (27 + 11).
            unsigned long long m_cur, m_l, m_r, m_go, m_t;
            float t1, t2;
            uint32_t lf = addr, trail = 1u, node = 0u, pc = 0u;
            if (KIND == VISIT_ASM) {
                asm volatile(
                    "ds_read_b128 v[32:35], %[lf]\n\tds_read_b128 v[36:39], %[lf] offset:16\n\tds_read_b128 v[40:43], %[lf] offset:32\n\tds_read_b128 v[44:47], %[lf] offset:48\n\t"
                    "s_mov_b64 %[cur], exec\n\ts_waitcnt lgkmcnt(2)\n\t"
                    "v_fma_f32 v32, v32, %[bx], %[nox]\n\tv_fma_f32 v33, v33, %[by], %[noy]\n\tv_fma_f32 v34, v34, %[bz], %[noz]\n\t"
                    "v_fma_f32 %[t1], v36, -|%[bx]|, v32\n\tv_fma_f32 v32, v36, |%[bx]|, v32\n\tv_fma_f32 v36, v37, -|%[by]|, v33\n\t"
                    "v_fma_f32 v33, v37, |%[by]|, v33\n\tv_fma_f32 v37, v38, -|%[bz]|, v34\n\tv_fma_f32 v34, v38, |%[bz]|, v34\n\t"
                    "v_max3_f32 %[t1], %[t1], v36, v37\n\tv_min_f32_e32 v32, v32, v33\n\tv_min3_f32 v32, v32, v34, %[nearest]\n\t"
                    "v_max_f32_e32 v33, 0, %[t1]\n\tv_cmp_le_f32_e64 %[ml], v33, v32\n\ts_waitcnt lgkmcnt(0)\n\t"
                    "v_fma_f32 v40, v40, %[bx], %[nox]\n\tv_fma_f32 v41, v41, %[by], %[noy]\n\tv_fma_f32 v42, v42, %[bz], %[noz]\n\t"
                    "v_fma_f32 %[t2], v44, -|%[bx]|, v40\n\tv_fma_f32 v40, v44, |%[bx]|, v40\n\tv_fma_f32 v44, v45, -|%[by]|, v41\n\t"
                    "v_fma_f32 v41, v45, |%[by]|, v41\n\tv_fma_f32 v45, v46, -|%[bz]|, v42\n\tv_fma_f32 v42, v46, |%[bz]|, v42\n\t"
                    "v_max3_f32 %[t2], %[t2], v44, v45\n\tv_min_f32_e32 v40, v40, v41\n\tv_min3_f32 v40, v40, v42, %[nearest]\n\t"
                    "v_max_f32_e32 v41, 0, %[t2]\n\tv_cmp_le_f32_e64 %[mr], v41, v40\n\tv_cmp_gt_f32_e32 vcc, %[t1], %[t2]\n\t"
                    "s_orn2_b64 %[mt], vcc, %[ml]\n\ts_and_b64 %[mgo], %[mt], %[mr]\n\ts_or_b64 %[mt], %[ml], %[mr]\n\ts_and_b64 %[ml], %[ml], %[mr]\n\t"
                    "s_and_b64 exec, %[cur], %[mt]\n\t"
                    "v_addc_co_u32_e64 %[node], %[mr], 0, %[lf], %[mgo]\n\tv_addc_co_u32_e64 %[trail], %[mr], %[trail], %[trail], %[ml]\n\t"
                    "v_cndmask_b32_e64 %[lf], v35, v43, %[mgo]\n\tv_cndmask_b32_e64 %[pc], v39, v47, %[mgo]\n\t"
                    "s_andn2_b64 exec, %[cur], %[mt]\n\ts_mov_b64 exec, %[cur]\n\tv_cmp_eq_u32_e32 vcc, 0, %[pc]\n\t"
                    : [node] "+v"(node), [lf] "+v"(lf), [pc] "+v"(pc), [trail] "+v"(trail), [cur] "=&s"(m_cur), [ml] "=&s"(m_l), [mr] "=&s"(m_r), [mgo] "=&s"(m_go),
                      [mt] "=&s"(m_t), [t1] "=&v"(t1), [t2] "=&v"(t2)
                    : [bx] "v"(x[0]), [by] "v"(x[1]), [bz] "v"(x[2]), [nox] "v"(x[3]), [noy] "v"(x[4]), [noz] "v"(x[5]), [nearest] "v"(x[6])
                    : "vcc", "scc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            } else {
                // operand pairs: y[0] = (bx, by), y[1] = (bz, ax), y[2] = (ay, az), y[3] = (nox, noy), y[4] = (noz, -)
                asm volatile(
                    "ds_read_b128 v[32:35], %[lf]\n\tds_read_b128 v[36:39], %[lf] offset:16\n\tds_read_b128 v[40:43], %[lf] offset:32\n\tds_read_b128 v[44:47], %[lf] offset:48\n\t"
                    "s_mov_b64 %[cur], exec\n\ts_waitcnt lgkmcnt(3)\n\t"
                    "v_pk_fma_f32 v[32:33], v[32:33], %[bxy], %[noxy] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" // tc_x (l, r) = c * bx + nox
                    "v_pk_fma_f32 v[34:35], v[34:35], %[bxy], %[noxy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t" // tc_y
                    "s_waitcnt lgkmcnt(2)\n\t"
                    "v_pk_fma_f32 v[36:37], v[36:37], %[bza], %[nozz] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" // tc_z
                    "v_fma_f32 %[t1], v38, -|%[bx]|, v32\n\tv_fma_f32 %[t2], v39, -|%[bx]|, v33\n\t"     // in_x (l), (r)
                    "v_pk_fma_f32 v[32:33], v[38:39], %[bza], v[32:33] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" // out_x = h_x * ax + tc_x
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_pk_fma_f32 v[38:39], v[40:41], %[ayz], v[34:35] op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t" // in_y
                    "v_pk_fma_f32 v[34:35], v[40:41], %[ayz], v[34:35] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"                              // out_y
                    "v_pk_fma_f32 v[40:41], v[42:43], %[ayz], v[36:37] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t" // in_z
                    "v_pk_fma_f32 v[36:37], v[42:43], %[ayz], v[36:37] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                              // out_z
                    "v_max3_f32 %[t1], %[t1], v38, v40\n\tv_max3_f32 %[t2], %[t2], v39, v41\n\t"
                    "v_min_f32_e32 v32, v32, v34\n\tv_min3_f32 v32, v32, v36, %[nearest]\n\tv_min_f32_e32 v33, v33, v35\n\tv_min3_f32 v33, v33, v37, %[nearest]\n\t"
                    "v_max_f32_e32 v42, 0, %[t1]\n\tv_cmp_le_f32_e64 %[ml], v42, v32\n\tv_max_f32_e32 v43, 0, %[t2]\n\tv_cmp_le_f32_e64 %[mr], v43, v33\n\t"
                    "v_cmp_gt_f32_e32 vcc, %[t1], %[t2]\n\ts_waitcnt lgkmcnt(0)\n\t"
                    "s_orn2_b64 %[mt], vcc, %[ml]\n\ts_and_b64 %[mgo], %[mt], %[mr]\n\ts_or_b64 %[mt], %[ml], %[mr]\n\ts_and_b64 %[ml], %[ml], %[mr]\n\t"
                    "s_and_b64 exec, %[cur], %[mt]\n\t"
                    "v_addc_co_u32_e64 %[node], %[mr], 0, %[lf], %[mgo]\n\tv_addc_co_u32_e64 %[trail], %[mr], %[trail], %[trail], %[ml]\n\t"
                    "v_cndmask_b32_e64 %[lf], v44, v46, %[mgo]\n\tv_cndmask_b32_e64 %[pc], v45, v47, %[mgo]\n\t"
                    "s_andn2_b64 exec, %[cur], %[mt]\n\ts_mov_b64 exec, %[cur]\n\tv_cmp_eq_u32_e32 vcc, 0, %[pc]\n\t"
                    : [node] "+v"(node), [lf] "+v"(lf), [pc] "+v"(pc), [trail] "+v"(trail), [cur] "=&s"(m_cur), [ml] "=&s"(m_l), [mr] "=&s"(m_r), [mgo] "=&s"(m_go),
                      [mt] "=&s"(m_t), [t1] "=&v"(t1), [t2] "=&v"(t2)
                    : [bxy] "v"(y[0]), [bza] "v"(y[1]), [ayz] "v"(y[2]), [noxy] "v"(y[3]), [nozz] "v"(y[4]), [bx] "v"(x[0]), [nearest] "v"(x[6])
                    : "vcc", "scc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            }
            x[7] += static_cast<float>(node + trail + pc) * 1e-30f; // keeps the results alive
            addr = ((lf >> 3) * 64u + addr + 64u) & 16383u & ~63u;   // the next pair depends on the chosen child word
        }
        if (KIND == VISIT_OLD || KIND == VISIT_NEW) {
            // The instruction mix of one inner-node visit of the LDS traversal, as a DEPENDENT computation the way the kernel
            // has it: 4 ds_read_b128 -> wait -> two slab tests -> ordering -> descent (the address of the next trip depends on
            // the result). OLD: 6 fma + 3 min + 3 max + min3 + max3 per box (round 2). NEW: 9 fma + max3 + min3 per box
            // (centre / half-extent planes, no per-axis min / max). One wave makes ONE visit per trip: with W waves per SIMD this
            // is what the real kernel does, minus divergence.
            v4f n0, n1, n2, n3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(n0), "=v"(n1), "=v"(n2), "=v"(n3) : "v"(addr));
            float tl, tr;
            const float ix = a, iy = b, iz = x[1], nox = x[2], noy = x[3], noz = x[4], nearest = x[5];
            if (KIND == VISIT_OLD) {
                auto box = [&](v4f lo, v4f hi) {
                    float t1 = __builtin_fmaf(lo.x, ix, nox), t2 = __builtin_fmaf(hi.x, ix, nox);
                    float tmin = __builtin_fminf(t1, t2), tmax = __builtin_fmaxf(t1, t2);
                    t1 = __builtin_fmaf(lo.y, iy, noy); t2 = __builtin_fmaf(hi.y, iy, noy);
                    tmin = __builtin_fmaxf(__builtin_fminf(t1, t2), tmin); tmax = __builtin_fminf(__builtin_fmaxf(t1, t2), tmax);
                    t1 = __builtin_fmaf(lo.z, iz, noz); t2 = __builtin_fmaf(hi.z, iz, noz);
                    tmin = __builtin_fmaxf(__builtin_fminf(t1, t2), tmin); tmax = __builtin_fminf(__builtin_fmaxf(t1, t2), tmax);
                    return (tmin > tmax || tmax <= 0.0f || tmin > nearest) ? 3e38f : tmin;
                };
                tl = box(n0, n1); tr = box(n2, n3);
            } else {
                const float ax = __builtin_fabsf(ix), ay = __builtin_fabsf(iy), az = __builtin_fabsf(iz);
                auto box = [&](v4f c, v4f h) {
                    const float cx = __builtin_fmaf(c.x, ix, nox), cy = __builtin_fmaf(c.y, iy, noy), cz = __builtin_fmaf(c.z, iz, noz);
                    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(h.x, -ax, cx), __builtin_fmaf(h.y, -ay, cy)), __builtin_fmaf(h.z, -az, cz));
                    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaf(h.x, ax, cx), __builtin_fmaf(h.y, ay, cy)), __builtin_fmaf(h.z, az, cz));
                    return (tmin > tmax || tmax <= 0.0f || tmin > nearest) ? 3e38f : tmin;
                };
                tl = box(n0, n1); tr = box(n2, n3);
            }
            const bool swap = tl > tr;
            const float tnear = swap ? tr : tl, tfar = swap ? tl : tr;
            uint32_t next = __float_as_uint(swap ? n2.w : n0.w);
            if (tnear > nearest) next ^= 0x40u; // "pop": some other pair
            x[6] = x[6] + ((tfar < nearest) ? 1.0f : 0.0f);
            addr = (next * 64u + addr + 64u) & 16383u & ~63u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < 16; ++k) sc += ss[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) sc += static_cast<uint32_t>(sm[k]);
    float s = static_cast<float>(m2 & 1) + static_cast<float>(sc) + static_cast<float>(addr) + s_tab[threadIdx.x].x; // (keeps s_tab allocated, at LDS offset 0)
#pragma unroll
    for (int k = 0; k < 16; ++k) s += x[k];
    if (s == 123.456f) sink[0] = s; // keeps everything live, never true
    if ((threadIdx.x & 63u) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = {t1 - t0, r1 - r0, r0, r1};
}

struct Row { std::string name; int insts_per_trip; };

static bool quick = false;
template <int KIND> void run(const char *name, double insts_per_trip, Stamp *d_out, float *d_sink, int cus) {
    for (int waves_per_simd : (quick ? std::vector<int>{2, 8} : std::vector<int>{1, 2, 4, 8})) {
        const int blocks = cus * waves_per_simd; // 256-thread blocks = 4 waves = one per SIMD; waves_per_simd blocks per CU
        // pick the trip count for ~8 ms launches, then hold the load for >= 0.7 s before the measured launches
        uint32_t trips = 2000;
        auto launch = [&](uint32_t n) { hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, n, 1.0001f, 0.5f, reinterpret_cast<const uint32_t *>(d_sink)); };
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); launch(trips); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        trips = static_cast<uint32_t>(std::min(4.0e6, std::max(2000.0, trips * 8.0 / std::max(ms, 0.01f))));
        float held = 0;
        while (held < 700.0f) {
            CK(hipEventRecord(e0)); for (int k = 0; k < 8; ++k) launch(trips); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); held += ms;
        }
        CK(hipEventRecord(e0)); launch(trips); CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float launch_ms = 0; CK(hipEventElapsedTime(&launch_ms, e0, e1));
        std::vector<Stamp> h(static_cast<size_t>(blocks) * 4);
        CK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for (const Stamp &s : h) { cyc.push_back(static_cast<double>(s.cycles)); clk.push_back(s.real ? 1e8 * s.cycles / s.real : 0.0); }
        std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
        std::nth_element(clk.begin(), clk.begin() + clk.size() / 2, clk.end());
        const double c = cyc[cyc.size() / 2], f = clk[clk.size() / 2];
        const double per_simd = c / (static_cast<double>(trips) * insts_per_trip * waves_per_simd);
        // were all waves really resident together? fraction of the waves whose [start, end] holds the median mid-point
        std::vector<double> mid;
        for (const Stamp &s : h) mid.push_back(0.5 * (static_cast<double>(s.r0) + static_cast<double>(s.r1)));
        std::nth_element(mid.begin(), mid.begin() + mid.size() / 2, mid.end());
        const double m = mid[mid.size() / 2];
        size_t together = 0;
        for (const Stamp &s : h) together += (static_cast<double>(s.r0) <= m && m <= static_cast<double>(s.r1)) ? 1 : 0;
        // chip-wide rate from the launch's wall time (hipEvents), no assumption about placement at all
        const double chip = static_cast<double>(h.size()) * trips * insts_per_trip / (launch_ms * 1e-3);
        printf("%-34s waves/SIMD %d: %7.3f cycles/instr/SIMD  (one wave: %6.2f cycles/instr)  clock %.3f GHz  %6.3f ns/instr/SIMD | %3.0f%% of the waves "
               "resident together, launch %.2f ms = %.3f T wave-instr/s chip-wide = %.3f per ns per SIMD (%.2f ns each)\n", name,
               waves_per_simd, per_simd, c / (trips * insts_per_trip), f * 1e-9, per_simd / (f * 1e-9), 100.0 * together / h.size(), launch_ms,
               chip * 1e-12, chip * 1e-9 / (4.0 * cus), (4.0 * cus) / (chip * 1e-9));
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main(int argc, char **argv) {
    const bool only_new = argc > 1 && std::string(argv[1]) != "salu" && std::string(argv[1]) != "visit"; // "salu": the scalar classes of round 5; any other argument: only the classes added in round 3
    const bool only_salu = argc > 1 && std::string(argv[1]) == "salu";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Stamp *d_out;
    float *d_sink;
    CK(hipMalloc(&d_out, sizeof(Stamp) * 4 * cus * 8));
    CK(hipMalloc(&d_sink, 64));
    printf("device: %s, %d CUs; cycles = s_memtime ticks, clock = s_memtime / s_memrealtime x 100 MHz, medians over all waves\n", prop.name, cus);
    if (argc > 1 && std::string(argv[1]) == "visit") { // round 5: the hand-written inner visit as shipped, and with packed FMAs on an interleaved node pair
        quick = true;
        run<FMA>("v_fma_f32", 64, d_out, d_sink, cus);
        run<PKFMA>("v_pk_fma_f32 (two FMAs per instruction)", 64, d_out, d_sink, cus);
        run<VISIT_ASM>("descending visit, 18 v_fma (per VISIT)", 1, d_out, d_sink, cus);
        run<VISIT_ASM_PK>("descending visit, 8 v_pk_fma + 2 v_fma (per VISIT)", 1, d_out, d_sink, cus);
        return 0;
    }
    if (only_salu) {
        // Does the scalar unit take issue slots from the vector ALU? Scalar classes alone, then VALU streams with 1/4, 1/2 and 1 scalar
        // instruction per vector instruction (rates are per VALU instruction for the mixed rows, per scalar instruction for the pure ones).
        quick = true;
        run<FMA>("v_fma_f32", 64, d_out, d_sink, cus);
        run<S_ADD>("s_add_u32", 64, d_out, d_sink, cus);
        run<S_AND64>("s_and_b64", 64, d_out, d_sink, cus);
        run<S_CSEL>("s_cselect_b32", 64, d_out, d_sink, cus);
        run<S_LOADHIT>("s_load_dword (scalar-cache hit)", 64, d_out, d_sink, cus);
        run<RFL_CHAIN>("v_readfirstlane -> s_add -> v_add chain (per link)", 64, d_out, d_sink, cus);
        run<FMA_SALU16>("64 v_fma + 16 s_add (per VALU)", 64, d_out, d_sink, cus);
        run<FMA_SALU>("64 v_fma + 32 s_add (per VALU)", 64, d_out, d_sink, cus);
        run<FMA_SALU64>("64 v_fma + 64 s_add (per VALU)", 64, d_out, d_sink, cus);
        run<FMA_SAND32>("64 v_fma + 32 s_and_b64 (per VALU)", 64, d_out, d_sink, cus);
        run<MINMAX>("v_min_f32 / v_max_f32", 64, d_out, d_sink, cus);
        run<MINMAX_SALU32>("64 v_min/max + 32 s_add (per VALU)", 64, d_out, d_sink, cus);
        run<VISIT_MIX>("visit mix: 25 full + 17 half VALU + 32 SALU (per VALU)", 42, d_out, d_sink, cus);
        return 0;
    }
    if (only_new) {
        quick = true;
        run<FMA>("v_fma_f32", 64, d_out, d_sink, cus);
        run<PKFMA>("v_pk_fma_f32 (two FMAs per instruction)", 64, d_out, d_sink, cus);
        run<FMAMIX>("v_fma_mix_f32 (f16 source)", 64, d_out, d_sink, cus);
        run<PERM>("v_perm_b32", 64, d_out, d_sink, cus);
        run<MINU>("v_min_u32", 64, d_out, d_sink, cus);
        run<MULLO>("v_mul_lo_u32", 64, d_out, d_sink, cus);
        run<MULHI>("v_mul_hi_u32", 64, d_out, d_sink, cus);
        run<MAD64>("v_mad_u64_u32", 64, d_out, d_sink, cus);
        run<DIVSCALE>("v_div_scale_f32", 64, d_out, d_sink, cus);
        run<DIVFMAS>("v_div_fmas_f32", 64, d_out, d_sink, cus);
        run<DIVFIXUP>("v_div_fixup_f32", 64, d_out, d_sink, cus);
        run<EXPF>("v_exp_f32", 64, d_out, d_sink, cus);
        run<LOGF>("v_log_f32", 64, d_out, d_sink, cus);
        run<READLANE>("v_readlane_b32", 64, d_out, d_sink, cus);
        run<WRITELANE>("v_writelane_b32", 64, d_out, d_sink, cus);
        run<LSHLADD64>("v_lshl_add_u64", 64, d_out, d_sink, cus);
        run<PKMUL>("v_pk_mul_f32", 64, d_out, d_sink, cus);
        run<CVTFU>("v_cvt_f32_u32", 64, d_out, d_sink, cus);
        return 0;
    }
    run<FMA>("v_fma_f32", 64, d_out, d_sink, cus);
    run<MINMAX>("v_min_f32 / v_max_f32", 64, d_out, d_sink, cus);
    run<MAX3>("v_max3_f32 / v_min3_f32", 64, d_out, d_sink, cus);
    run<CNDMASK>("v_cndmask_b32 (sgpr mask)", 64, d_out, d_sink, cus);
    run<CMP>("v_cmp_gt_f32 -> sgpr", 64, d_out, d_sink, cus);
    run<CMP_CND>("v_cmp -> v_cndmask pairs", 64, d_out, d_sink, cus);
    run<RCP>("v_rcp_f32", 64, d_out, d_sink, cus);
    run<SQRT>("v_sqrt_f32", 64, d_out, d_sink, cus);
    run<ADDU>("v_add_u32", 64, d_out, d_sink, cus);
    run<LSHLADD>("v_lshl_add_u32", 64, d_out, d_sink, cus);
    run<FMA_DEP>("v_fma_f32, ONE dependent chain", 64, d_out, d_sink, cus);
    run<FMA_SALU>("64 v_fma + 32 s_add (per VALU)", 64, d_out, d_sink, cus);
    run<FMA_DSREAD>("64 v_fma + 4 ds_read_b128 (per VALU)", 64, d_out, d_sink, cus);
    run<VISIT_OLD>("visit, min/max planes (per VISIT)", 1, d_out, d_sink, cus);
    run<VISIT_NEW>("visit, centre/half planes (per VISIT)", 1, d_out, d_sink, cus);
    quick = true; // more instruction classes, at 2 and 8 waves per SIMD
    run<MULF>("v_mul_f32", 64, d_out, d_sink, cus);
    run<ADDF>("v_add_f32", 64, d_out, d_sink, cus);
    run<ANDB>("v_and_b32", 64, d_out, d_sink, cus);
    run<LSHL>("v_lshlrev_b32", 64, d_out, d_sink, cus);
    run<BFE>("v_bfe_u32", 64, d_out, d_sink, cus);
    run<CVTUB>("v_cvt_f32_ubyte1", 64, d_out, d_sink, cus);
    run<ADDC>("v_addc_co_u32 (sgpr carry)", 64, d_out, d_sink, cus);
    run<MOV>("v_mov_b32", 64, d_out, d_sink, cus);
    run<MED3>("v_med3_f32", 64, d_out, d_sink, cus);
    run<ANDOR>("v_and_or_b32", 64, d_out, d_sink, cus);
    run<CNDVCC>("v_cndmask_b32_e32 (vcc)", 64, d_out, d_sink, cus);
    run<MINE64>("v_min_f32_e64 (abs modifier)", 64, d_out, d_sink, cus);
    run<PKFMA>("v_pk_fma_f32 (two FMAs per instruction)", 64, d_out, d_sink, cus);
    run<FMAMIX>("v_fma_mix_f32 (f16 source)", 64, d_out, d_sink, cus);
    run<PERM>("v_perm_b32", 64, d_out, d_sink, cus);
    run<MINU>("v_min_u32", 64, d_out, d_sink, cus);
    return 0;
}
