// Shared pieces of the persistent BiLSTM layer kernels (lstm.hip: split-bf16 teams; lstm_f32.hip: exact-fp32 teams).
#pragma once
#include "mdd_internal.h"

namespace mdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

// Gate nonlinearities of the reference-width recurrences (lstm_step_packed_kernel and lstm_layer_f32_kernel share them, which keeps
// the two bit-identical).  An fp32 MFMA runs on the SIMD's fp32 lanes: unlike next to a bf16 MFMA, every vector instruction of the
// cell update is time the matrix products do not get, so the libm forms (~40 instructions per call) are replaced by short ones built
// on the hardware exp2 / rcp (1 ulp each) that stay at fp32 working accuracy in the sense the cell update needs: absolute error
// <= 9e-8 (sigmoid) / 1.3e-7 (tanh) over the whole range; relative: sigmoid <= 1.6 ulp for v >= 0 and ~0.7 |v| ulp of its (small)
// result for v < 0 (the argument product is rounded once), tanh <= 3.9 ulp
// (measured against fp64 in tests/test_gpu_parity.py::test_gate_functions_accuracy).  End to end they cost nothing: with the libm
// forms compiled in instead, the log-probs' distance to a float64 evaluation at B = 64 x T' = 250 was the same 4.65e-7 mean
// (profiles/round3_lstm_f32_notes.txt).
//   sigmoid(v) = 1 / (1 + 2^(-v log2 e)), the reciprocal refined by one Newton step; the exponent is capped so that 1 + e stays finite
//   tanh(v)    = sign(v) (1 - e) / (1 + e), e = 2^(-2 |v| log2 e) for |v| >= 1/4; the odd Taylor polynomial up to v^9 below that
//                (1 - e cancels there; the next term, 1382/155925 v^10, is < 1e-8 relative at 1/4)
__device__ __forceinline__ float gate_sigmoid(float v) {
    const float a = v * -1.44269504088896340736f;
    const float e = __builtin_amdgcn_exp2f(a > 126.f ? 126.f : a);      // (a select, not fminf: a NaN must stay a NaN)
    const float d = 1.f + e, r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.f), r);
}
__device__ __forceinline__ float gate_tanh(float v) {
    const float a = fabsf(v), v2 = v * v;
    const float e = __builtin_amdgcn_exp2f(a * -2.88539008177792681472f);
    const float d = 1.f + e, r0 = __builtin_amdgcn_rcpf(d);
    const float big = (1.f - e) * __builtin_fmaf(r0, __builtin_fmaf(-d, r0, 1.f), r0);
    float p = __builtin_fmaf(v2, 62.f / 2835.f, -17.f / 315.f);
    p = __builtin_fmaf(p, v2, 2.f / 15.f);
    p = __builtin_fmaf(p, v2, -1.f / 3.f);
    const float small = __builtin_fmaf(a, p * v2, a);
    return __builtin_copysignf(a < 0.25f ? small : big, v);
}


struct PersistArgs {
    const float *gx;                 // [T][B][2][4H] permuted gate columns
    SplitPtr whh;                    // Whh' [2][4H][H] row-major hi/lo (split-bf16 kernels)
    const float *whh_f32;            // Whh' in the packed consumer order of lstm_step_packed_kernel (exact-fp32 kernel)
    unsigned short *hx;              // [2 parity][32 teams][tiles][H/4 chunk columns][16 rows] x 16 B: tagged chunks of the state in flight
    unsigned int *sync;              // [16] unused, [16] abort flag  (zeroed before every launch)
    int *err_flag;
    float *out, *out_raw;
    SplitPtr out_split;
    const float *oscale, *oshift;
    int T, B, BG, BGr;               // BG: padded rows per group (multiple of 16), BGr: real rows per group
    long long *dbg;                  // diagnostic builds only: per-workgroup cycle sums of the step phases (null in production)
    const int *seqlen;               // see LstmStepArgs::seqlen (null = every row runs all T steps)
    float *gates_save, *c_save;      // TRAIN instantiations: post-activation i,f,g,o [T][B][2][H][4] and c_t [T][B][2][H] for the backward pass
    int early;                       // diagnostic (MDD_LSTM_EARLY): request the next tile's panel a whole MFMA section too early, so that stale panels and the redo path occur
};

// diagnostic phase stamps (MDD_LSTM_DBG): cycle sums per phase of the step, read by tools/lstm_stamps.py
#define PSTAMP(i) do { if (a.dbg) { long long n_ = __builtin_readcyclecounter(); ph[i] += n_ - tst; tst = n_; } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which on
// gfx950 is s_waitcnt vmcnt(0): every barrier would drain the write-through publish stores, the output stores and the
// in-flight LDS-DMA / sweep requests this kernel deliberately keeps outstanding across phases.  Global memory needs no
// ordering here (the hand-off is data-tagged); where an LDS-DMA result must be visible, an explicit s_waitcnt vmcnt(N)
// precedes the barrier.
__device__ __forceinline__ void wait_vmcnt(int n) {   // s_waitcnt vmcnt(n) for a wave-uniform n (the count is an immediate)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;   // n >= 9: waiting for more than asked is always safe
    }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KB lane-linear at the wave-uniform LDS address l), written as
// inline assembly on purpose: the compiler's wait-count pass treats the builtin form as an LDS write still in flight
// and puts s_waitcnt vmcnt(0) in front of the next LDS read it can relate to it -- here the cell update's slab read and
// the staged sweep's read-back -- which also waits for every store issued in between (the write-through publish alone
// takes ~1 us to acknowledge) and for the sweep request the kernel wants in flight during the cell update.  All waits
// on these transfers are the explicit counted ones below.
// Same, with the global address as a wave-uniform base (SGPR pair) plus a 32-bit per-lane byte offset and the LDS
// address already scalar: no 64-bit vector address arithmetic per piece.
template <bool SC1>
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned la) {
    if (SC1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" :: "v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
}
template <bool SC1>
__device__ __forceinline__ void lds_dma16(const void *g, void *l) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t *)l);
    if (SC1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" :: "v"(g), "s"(la) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(la) : "memory", "m0");
}


static constexpr int kPersistGrid = 256;   // see persistent_grid_fits()

}  // namespace mdd
