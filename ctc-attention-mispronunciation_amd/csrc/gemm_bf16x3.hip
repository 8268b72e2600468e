// Split-bf16 ("bf16x3") NT GEMM on the gfx950 bf16 matrix cores: C[M,N] = A[M,K] . W[N,K]^T (+ bias).
//
// Every fp32 operand x is carried as two bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa
// bits together), and the product is evaluated as  Ah.Wh + Ah.Wl + Al.Wh  with fp32 accumulation in
// v_mfma_f32_32x32x16_bf16.  The dropped Al.Wl term is 2^-16 of a product; measured on the whole model
// (4 x BiLSTM-384, T'=100) the log-probs move by 7.6e-6 against 1.9e-6 for the exact-fp32 MFMA path,
// both far inside the 1e-4 parity tolerance (DESIGN.md "precision modes").  Three bf16 MFMAs cost 3/16 of
// the fp32 MFMA they replace, which is why the time-batched contractions of the path run here:
// BiLSTM input projections (AA/models/model_ctc.py:28-29,44), text-encoder projection (:150,198),
// `score` Linear (:151,201) and the attention scores bmm (:204).
//
// Operands arrive pre-split from their producers (conv1, the LSTM step epilogue, the embedding gather;
// weights are split once at load time), so this kernel moves the same 4 bytes per element an fp32 GEMM
// would and spends no VALU on conversion.
//
// Kernels (all stream their operands HBM -> LDS with LDS-DMA, 16 B per lane, and use v_mfma_f32_16x16x32_bf16 with the WEIGHT
// fragment as the first operand, so that a lane's four accumulator registers are four consecutive C columns): the 8-phase 256x256
// kernel (large projections), the single-barrier 256x256 kernel it is screened against, and the 128x128 kernel (small / batched /
// split-output GEMMs).  An output element is accumulated identically in all of them (batch-size independent bits).  The
// register-staged and 32x32x16 forms of round 1 measured slower and are no longer built.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "mdd_internal.h"

namespace mdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int XBM = 128, XBN = 128, XBK = 32, XROW = 64;   // XROW: bytes per LDS row (32 bf16, unpadded, XOR-swizzled)
constexpr int XPLANE = XBM * XROW;                          // 8192 B
// LDS layout: 16-byte chunk c of row r lives at chunk position c ^ ((r >> 2) & 3): the ds_read_b128 fragment reads
// (16-lane groups {0-3,12-15,20-27}, ..: rows sharing r&3 differ in (r>>2)&3) and the ds_write_b128 staging writes are
// bank-conflict-free.
//
__device__ __forceinline__ bf16x8 x3_frag(const unsigned char *plane, int row, int kbyte) {
    return *reinterpret_cast<const bf16x8 *>(plane + row * XROW + ((((kbyte >> 4) ^ ((row >> 2) & 3))) << 4));
}

__device__ __forceinline__ unsigned short bf16_bits(float x) {   // round-to-nearest-even, NaN-preserving cast
    __bf16 b = (__bf16)x;
    return *reinterpret_cast<unsigned short *>(&b);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// ---- LDS-DMA variant of the 128x128 kernel: tiles go global -> LDS directly (global_load_lds_dwordx4), no staging
// registers and no ds_write pass.  The LDS image is lane-linear per wave-instruction (64 x 16 B = 16 rows of one plane),
// so the XOR swizzle is applied to the per-lane SOURCE address; fragment reads use the same swizzle (x3_frag).
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void x3_glds_tile(const unsigned short *__restrict__ Ah, const unsigned short *__restrict__ Al,
                                             const unsigned short *__restrict__ Wh, const unsigned short *__restrict__ Wl, int lda, int ldw,
                                             int M, int N, int m0, int n0, int k0, unsigned char *stage, int wave, int lane) {
    // wave w issues 8 instructions: plane p = j>>1, 16-row group g = (j&1)*4 + w
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = j >> 1, g = (j & 1) * 4 + wave;
        const int row = g * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
        const unsigned short *base = p == 0 ? Ah : (p == 1 ? Al : (p == 2 ? Wh : Wl));
        const int ld = p < 2 ? lda : ldw, rtot = p < 2 ? M : N, r0 = p < 2 ? m0 : n0;
        const unsigned short *src = base + (size_t)min(r0 + row, rtot - 1) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds(src, (lds_void *)(stage + p * XPLANE + g * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_glds_kernel(const unsigned short *__restrict__ Ah, const unsigned short *__restrict__ Al,
                                                                   const unsigned short *__restrict__ Wh, const unsigned short *__restrict__ Wl,
                                                                   const float *__restrict__ bias, float *__restrict__ C,
                                                                   unsigned short *__restrict__ Ch, unsigned short *__restrict__ Cl, int M, int N,
                                                                   int K, int lda, int ldw, int ldc, long sA, long sW, long sC, int tiles_n) {
    // v_mfma_f32_16x16x32_bf16, per wave 64x64 = 4x4 tiles.  Same shape and product order as the 256x256 kernel, so an
    // output element is accumulated identically whichever kernel the problem size selects (results do not depend on
    // how many utterances share the batch, bit for bit).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][4 * XPLANE];   // one array: [stage][Ah|Al|Wh|Wl]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int nwg = gridDim.x, bid = blockIdx.x;
    int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int m0 = tm * XBM, n0 = tn * XBN;
    Ah += (size_t)blockIdx.z * sA; Al += (size_t)blockIdx.z * sA;
    Wh += (size_t)blockIdx.z * sW; Wl += (size_t)blockIdx.z * sW;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = K / XBK;
    const int l16 = lane & 15, kq16 = (lane >> 4) * 16;
    x3_glds_tile(Ah, Al, Wh, Wl, lda, ldw, M, N, m0, n0, 0, lds[0], wave, lane);
    __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) x3_glds_tile(Ah, Al, Wh, Wl, lda, ldw, M, N, m0, n0, (kt + 1) * XBK, lds[cur ^ 1], wave, lane);
        const unsigned char *st = lds[cur];
        bf16x8 fwh[4], fwl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fwh[j] = x3_frag(st + 2 * XPLANE, wn * 64 + j * 16 + l16, kq16);
            fwl[j] = x3_frag(st + 3 * XPLANE, wn * 64 + j * 16 + l16, kq16);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bf16x8 fah = x3_frag(st, wm * 64 + i * 16 + l16, kq16);
            const bf16x8 fal = x3_frag(st + XPLANE, wm * 64 + i * 16 + l16, kq16);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwl[j], fah, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[j], fal, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[j], fah, acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int q4 = (lane >> 4) * 4;   // W fragment first: D row = 4*(lane>>4) + r = C column, D col = lane&15 = C row
    if (EPI == 0) C += (size_t)blockIdx.z * sC; else { Ch += (size_t)blockIdx.z * sC; Cl += (size_t)blockIdx.z * sC; }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = m0 + wm * 64 + i * 16 + l16;
            if (row >= M) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int col = n0 + wn * 64 + j * 16 + q4 + r;
                if (col >= N) continue;
                const float v = acc[i][j][r] + (bias ? bias[col] : 0.f);
                if (EPI == 0) C[(size_t)row * ldc + col] = v;
                else {
                    const unsigned short h = bf16_bits(v);
                    Ch[(size_t)row * ldc + col] = h;
                    Cl[(size_t)row * ldc + col] = bf16_bits(v - bf16_to_f32(h));
                }
            }
        }
}

// ---- LDS-DMA, 256x256 tile, 8 waves (2 x 4), per wave 128x64: half the L2 -> LDS bytes per flop of the 128x128 tile.
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_glds256_kernel(const unsigned short *__restrict__ Ah, const unsigned short *__restrict__ Al,
                                                                      const unsigned short *__restrict__ Wh, const unsigned short *__restrict__ Wl,
                                                                      const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                                      int lda, int ldw, int ldc, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];   // [2 stages][Ah|Al|Wh|Wl][256 rows x 64 B] = 128 KB
    constexpr int PL = 256 * XROW;                                              // 16 KB per plane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int nwg = gridDim.x, bid = blockIdx.x;
    int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc16[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fill = [&](int k0, unsigned char *stage, int j0 = 0, int j1 = 8) {
#pragma unroll
        for (int j = j0; j < j1; j++) {
            const int id = j * 8 + wave, p = id >> 4, g = id & 15;
            const int row = g * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
            const unsigned short *base = p == 0 ? Ah : (p == 1 ? Al : (p == 2 ? Wh : Wl));
            const int ld = p < 2 ? lda : ldw, rtot = p < 2 ? M : N, r0 = p < 2 ? m0 : n0;
            const unsigned short *src = base + (size_t)min(r0 + row, rtot - 1) * ld + k0 + c * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(stage + p * PL + g * 1024), 16, 0, 0);
        }
    };
    const int nk = K / XBK;
    fill(0, glds_smem);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        unsigned char *nst = glds_smem + (cur ^ 1) * 4 * PL;
        const unsigned char *st = glds_smem + cur * 4 * PL;
        // v_mfma_f32_16x16x32_bf16: one k-step per K-tile; lane (row l&15, k-slice l>>4)
        const int l16 = lane & 15, kq16 = (lane >> 4) * 16;
        bf16x8 fwh[4], fwl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fwh[j] = x3_frag(st + 2 * PL, wn * 64 + j * 16 + l16, kq16);
            fwl[j] = x3_frag(st + 3 * PL, wn * 64 + j * 16 + l16, kq16);
        }
        // the next stage's 8 LDS-DMA pieces go out two at a time between the MFMA groups: a piece costs the issuing
        // wave 100+ cycles, which the SIMD's other wave covers with MFMAs only if the two are not doing it at once
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if ((i & 1) == 0) {
                if (more) fill((kt + 1) * XBK, nst, i, i + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bf16x8 fah = x3_frag(st, wm * 128 + i * 16 + l16, kq16);
            const bf16x8 fal = x3_frag(st + PL, wm * 128 + i * 16 + l16, kq16);
#pragma unroll
            for (int j = 0; j < 4; j++) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwl[j], fah, acc16[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[j], fal, acc16[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[j], fah, acc16[i][j], 0, 0, 0);
            if (i & 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    {   // W fragment first: D row = 4*(lane>>4) + r = C column, D col = lane&15 = C row (see the 8-phase kernel)
        const int l16 = lane & 15, q4 = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = m0 + wm * 128 + i * 16 + l16;
                if (row >= M) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int col = n0 + wn * 64 + j * 16 + q4 + r;
                    if (col < N) C[(size_t)row * ldc + col] = acc16[i][j][r] + (bias ? bias[col] : 0.f);
                }
            }
    }
}

// ---- 256x256 tile, eight barrier-delimited phases per K-tile, two wave groups one barrier apart.
// Same LDS image and the same per-element arithmetic as gemm_bf16x3_glds256_kernel (identical bits); what changes is
// who waits for what.  The K-tile's operands are four half-tiles (A rows 0..127 / 128..255, W rows 0..127 / 128..255;
// hi + lo planes, 16 KB = 16 LDS-DMA pieces = 2 per wave) and a wave's share of C is four 64x32 quadrants (A half a,
// W half b) taken in the order (0,0) (0,1) (1,1) (1,0).  A quadrant is a load phase L (fragment reads for it, the two
// DMA pieces of one half-tile of the NEXT K-tile, a counted s_waitcnt vmcnt(4) that retires the half-tile the next L
// will read while the two issued after it stay in flight) and an MFMA phase M (24 MFMAs), each ended by a raw
// s_barrier.  Waves 4..7 (the other wave of every SIMD) run one barrier behind waves 0..3, so one group's M always
// overlaps the other group's L: the matrix pipe no longer idles while both waves of a SIMD read fragments and wait.
// Hazards: a half-tile issued in phase p is waited for at the end of L(p+2) by every wave, read in L(p+3) -- for the
// late group that is after one more barrier than the early group's wait, as the stagger requires -- and its buffer is
// refilled eight phases later.
template <bool DMA_IN_M, bool STAMP = false, bool WFIRST = true>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_ph8_kernel(const unsigned short *__restrict__ Ah, const unsigned short *__restrict__ Al,
                                                                 const unsigned short *__restrict__ Wh, const unsigned short *__restrict__ Wl,
                                                                 const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                                 int lda, int ldw, int ldc, int tiles_n, long long *stamps = nullptr) {
    // STAMP (diagnostic build only): per wave, cycles spent in load-phase bodies / waiting at their barrier / MFMA-phase bodies /
    // waiting at theirs, summed over the K loop -> stamps[(workgroup * 8 + wave) * 4 + {0,1,2,3}]
    long long sacc[4] = {0, 0, 0, 0}, stt = 0;
#define PH8_T(i_) do { if (STAMP) { const long long n_ = (long long)__builtin_readcyclecounter(); sacc[i_] += n_ - stt; stt = n_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];   // [2 stages][Ah|Al|Wh|Wl][256 rows x 64 B] = 128 KB
    constexpr int PL = 256 * XROW;                                              // 16 KB per plane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = (wave & 1) + 2 * grp;   // SIMD partners w, w+4 sit in different groups
    int nwg = gridDim.x, bid = blockIdx.x;
    int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[2][2][4][2];                                          // [A half][W half][row tile][col tile]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // half-tile ht of the K-tile at k0 -> stage: 0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1; this wave's 2 pieces.
    // Addresses are a wave-uniform base (SGPR pair, advanced by k0) plus a per-lane 32-bit byte offset fixed for the
    // whole K loop, and the LDS address is scalar too: no vector address arithmetic inside the loop (on this chip a
    // SIMD's vector instructions are not covered by its MFMAs).
    const unsigned wave_s = __builtin_amdgcn_readfirstlane((unsigned)wave);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void *)glds_smem;
    unsigned voff[4];
#pragma unroll
    for (int ht = 0; ht < 4; ht++) {
        const bool isA = ht == 0 || ht == 3;
        const int half = (ht >= 2) ? 1 : 0, g = half * 8 + wave;     // 16-row group inside the plane
        const int row = g * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
        const int ld = isA ? lda : ldw, rtot = isA ? M : N, r0 = isA ? m0 : n0;
        voff[ht] = (unsigned)(((size_t)min(r0 + row, rtot - 1) * ld + c * 8) * 2);
    }
    auto issue = [&](int ht, int k0, unsigned stage_off) {
        const bool isA = ht == 0 || ht == 3;
        const unsigned g = ((ht >= 2) ? 8u : 0u) + wave_s;
        const unsigned short *bh = (isA ? Ah : Wh) + k0, *bl = (isA ? Al : Wl) + k0;
        const unsigned la = lds0 + stage_off + g * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff[ht]), "s"(bh), "s"(la + (unsigned)((isA ? 0 : 2) * PL)) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff[ht]), "s"(bl), "s"(la + (unsigned)((isA ? 1 : 3) * PL)) : "memory", "m0");
    };
    const int nk = K / XBK;
    const int l16 = lane & 15, kq16 = (lane >> 4) * 16;
#pragma unroll
    for (int ht = 0; ht < 4; ht++) issue(ht, 0, 0u);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // A0, W0 of K-tile 0
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");            // from here on one barrier behind group 0
    bf16x8 fah[4], fal[4], fwh[2][2], fwl[2][2];                     // A fragments of the current half; W fragments of both halves
#define PH8_END_L() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PH8_T(0); asm volatile("s_barrier" ::: "memory"); PH8_T(1); __builtin_amdgcn_s_setprio(1)
#define PH8_END_M() __builtin_amdgcn_s_setprio(0); PH8_T(2); asm volatile("s_barrier" ::: "memory"); PH8_T(3)
#define PH8_LOAD_A(a_) _Pragma("unroll") for (int i = 0; i < 4; i++) { \
            fah[i] = x3_frag(st, (a_) * 128 + wr * 64 + i * 16 + l16, kq16); fal[i] = x3_frag(st + PL, (a_) * 128 + wr * 64 + i * 16 + l16, kq16); }
#define PH8_LOAD_W(b_) _Pragma("unroll") for (int j = 0; j < 2; j++) { \
            fwh[b_][j] = x3_frag(st + 2 * PL, (b_) * 128 + wc * 32 + j * 16 + l16, kq16); fwl[b_][j] = x3_frag(st + 3 * PL, (b_) * 128 + wc * 32 + j * 16 + l16, kq16); }
#define PH8_MM(x_, y_, c_) (WFIRST ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_, y_, c_, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(y_, x_, c_, 0, 0, 0))
#define PH8_MFMA(a_, b_) _Pragma("unroll") for (int i = 0; i < 4; i++) { \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = PH8_MM(fwl[b_][j], fah[i], acc[a_][b_][i][j]); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = PH8_MM(fwh[b_][j], fal[i], acc[a_][b_][i][j]); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = PH8_MM(fwh[b_][j], fah[i], acc[a_][b_][i][j]); }
    // DMA_IN_M (experiment, not the default): the two LDS-DMA pieces of a phase pair are issued from inside the MFMA phase (behind
    // its first MFMA group) instead of from the load phase; each counted wait then sees one issue less in front of it (vmcnt(2) where
    // the load-phase form has vmcnt(4)).  Stamps (profiles/round2_gemm_phase_stamps.txt): the MFMA phases, not the load phases, set
    // the length of a barrier interval (per K-tile and wave: MFMA bodies 1744 cycles against load bodies 1336), so moving issue work
    // into them lengthens the interval: 5-8 % slower.
#define PH8_MFMA_DMA(a_, b_, ht_) do { if (DMA_IN_M) { \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwl[b_][j], fah[0], acc[a_][b_][0][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[b_][j], fal[0], acc[a_][b_][0][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[b_][j], fah[0], acc[a_][b_][0][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); \
            if (more) issue(ht_, nk0, nst); \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int i = 1; i < 4; i++) { \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwl[b_][j], fah[i], acc[a_][b_][i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[b_][j], fal[i], acc[a_][b_][i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 2; j++) acc[a_][b_][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwh[b_][j], fah[i], acc[a_][b_][i][j], 0, 0, 0); } \
        } else { PH8_MFMA(a_, b_) } } while (0)
    if (STAMP) stt = (long long)__builtin_readcyclecounter();
    for (int kt = 0; kt < nk; kt++) {
        const unsigned char *st = glds_smem + (kt & 1) * 4 * PL;
        const unsigned nst = (unsigned)(((kt & 1) ^ 1) * 4 * PL);   // byte offset of the other stage
        const bool more = kt + 1 < nk;
        const int nk0 = (kt + 1) * XBK;
        // ---- quadrant (0,0)
        PH8_LOAD_A(0) PH8_LOAD_W(0)
        if (DMA_IN_M) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                              // W1 of this K-tile
        } else {
            if (more) issue(0, nk0, nst);
            if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // W1 of this K-tile
        }
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_L();
        PH8_MFMA_DMA(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_M();
        // ---- quadrant (0,1)
        PH8_LOAD_W(1)
        if (DMA_IN_M) {
            if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A1 of this K-tile
        } else {
            if (more) issue(1, nk0, nst);
            if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A1 of this K-tile
        }
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_L();
        PH8_MFMA_DMA(0, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_M();
        // ---- quadrant (1,1)
        PH8_LOAD_A(1)
        if (!DMA_IN_M && more) issue(2, nk0, nst);
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_L();
        PH8_MFMA_DMA(1, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_M();
        // ---- quadrant (1,0): its fragments are already in registers
        if (DMA_IN_M) {
            if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                  // A0, W0 of the next K-tile
        } else if (more) { issue(3, nk0, nst); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }                        // A0, W0 of the next K-tile
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_L();
        PH8_MFMA_DMA(1, 0, 3);
        __builtin_amdgcn_sched_barrier(0);
        PH8_END_M();
    }
#undef PH8_MFMA_DMA
#undef PH8_MM
    if (STAMP && stamps && lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 4; i++) stamps[((size_t)blockIdx.x * 8 + wave) * 4 + i] = sacc[i];
#undef PH8_T
#undef PH8_END_L
#undef PH8_END_M
#undef PH8_LOAD_A
#undef PH8_LOAD_W
#undef PH8_MFMA
    if (grp == 0) asm volatile("s_barrier" ::: "memory");            // pairs with group 1's last barrier
    // The W fragment is the MFMA's first operand (D rows = weight rows = C columns) and the A fragment its second (D columns = C
    // rows): D row = 4*(lane>>4) + r, D col = lane&15, so a lane's four accumulator registers are FOUR CONSECUTIVE COLUMNS of one C
    // row -- the tile leaves as 16-byte stores (32 per wave, 16 rows x 64 B each) instead of 128 scalar ones.
    const int q4 = (lane >> 4) * 4;
    if (!WFIRST) {   // measurement form only (A fragment first: D row = C row, D col = C column; scalar stores)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int col = n0 + b * 128 + wc * 32 + j * 16 + l16;
                        if (col >= N) continue;
                        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = m0 + a * 128 + wr * 64 + i * 16 + q4 + r;
                            if (row < M) C[(size_t)row * ldc + col] = acc[a][b][i][j][r] + bv;
                        }
                    }
        return;
    }
    if (STAMP && stamps == nullptr) {   // diagnostic form: no C stores at all (what the epilogue's HBM writes cost the launch); keep the accumulators alive
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) keep += acc[a][b][i][j][0] + acc[a][b][i][j][1] + acc[a][b][i][j][2] + acc[a][b][i][j][3];
        if (keep == 12345.678f) C[0] = keep;
        return;
    }
    const bool n_vec = (ldc % 4 == 0) && (((size_t)C & 15) == 0);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int row = m0 + a * 128 + wr * 64 + i * 16 + l16;
                    const int col = n0 + b * 128 + wc * 32 + j * 16 + q4;
                    if (row >= M || col >= N) continue;
                    const f32x4 v = acc[a][b][i][j];
                    float *dst = C + (size_t)row * ldc + col;
                    if (n_vec && col + 3 < N) {
                        float4 o = make_float4(v[0], v[1], v[2], v[3]);
                        if (bias) { const float4 bv = *reinterpret_cast<const float4 *>(bias + col); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
                        *reinterpret_cast<float4 *>(dst) = o;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (col + r < N) dst[r] = v[r] + (bias ? bias[col + r] : 0.f);
                    }
                }
}


int init_gemm_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16x3_glds256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16x3_ph8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16x3_ph8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)(gemm_bf16x3_ph8_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)(gemm_bf16x3_ph8_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)(gemm_bf16x3_ph8_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 256 * XROW));
    return MDD_OK;
}

int launch_gemm_bf16x3(const SplitPtr &A, const SplitPtr &W, const float *bias, float *C, const SplitPtr *Csplit, int M, int N,
                       int K, int lda, int ldw, int ldc, int batch, long sA, long sW, long sC, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || K % XBK || lda % 8 || ldw % 8 || sA % 8 || sW % 8) {
        set_error("gemm_bf16x3: bad shape M=%d N=%d K=%d lda=%d ldw=%d (K must be a multiple of %d)", M, N, K, lda, ldw, XBK);
        return MDD_ERR_ARG;
    }
    if (!Csplit && batch == 1 && M >= 1024 && N >= 512) {   // large projection: 256x256 tiles
        const int tn = (N + 255) / 256;
        const dim3 grid(((M + 255) / 256) * tn), block(512);
        const size_t smem = 2 * 4 * 256 * XROW;
        // MDD_GEMM=glds256: the single-barrier kernel; MDD_GEMM_DMA=M: the 8-phase kernel with the LDS-DMA issued from inside the MFMA
        // phases (measured 5-8 % slower than from the load phases: profiles/round2_gemm_phase_stamps.txt)
        static const int form = (getenv("MDD_GEMM") && !strcmp(getenv("MDD_GEMM"), "glds256")) ? 0 : ((getenv("MDD_GEMM_DMA") && !strcmp(getenv("MDD_GEMM_DMA"), "M")) ? 2 : 1);
        if (form == 2) hipLaunchKernelGGL(gemm_bf16x3_ph8_kernel<true>, grid, block, smem, st, A.hi, A.lo, W.hi, W.lo, bias, C, M, N, K, lda, ldw, ldc, tn);
        else if (form == 1) hipLaunchKernelGGL(gemm_bf16x3_ph8_kernel<false>, grid, block, smem, st, A.hi, A.lo, W.hi, W.lo, bias, C, M, N, K, lda, ldw, ldc, tn);
        else hipLaunchKernelGGL(gemm_bf16x3_glds256_kernel, grid, block, smem, st, A.hi, A.lo, W.hi, W.lo, bias, C, M, N, K, lda, ldw, ldc, tn);
        MDD_LAUNCH_CHECK();
        return MDD_OK;
    }
    const int tm = (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN;   // 128x128 tiles (also batched / split output)
    dim3 grid(tm * tn, 1, batch), block(256);
    if (Csplit)
        hipLaunchKernelGGL(gemm_bf16x3_glds_kernel<1>, grid, block, 0, st, A.hi, A.lo, W.hi, W.lo, bias, (float *)nullptr, Csplit->hi, Csplit->lo,
                           M, N, K, lda, ldw, ldc, sA, sW, sC, tn);
    else
        hipLaunchKernelGGL(gemm_bf16x3_glds_kernel<0>, grid, block, 0, st, A.hi, A.lo, W.hi, W.lo, bias, C, (unsigned short *)nullptr,
                           (unsigned short *)nullptr, M, N, K, lda, ldw, ldc, sA, sW, sC, tn);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
namespace mdd {
__global__ void diag_fill_kernel(unsigned short *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (unsigned short)(0x3c00u + (h & 0x3ffu) + ((h >> 10) & 1u) * 0x8000u);   // bf16 of magnitude 0.0078..0.031, random sign
    }
}
__global__ void diag_diff_kernel(const unsigned *a, const unsigned *b, size_t n, unsigned *count) {
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(count, c);
}
}  // namespace mdd

// Race screen for the 8-phase kernel (tests/test_gpu_parity.py): the same pseudo-random split operands through the single-barrier
// kernel once and through BOTH forms of the 8-phase kernel (LDS-DMA issued from the load phases / from the MFMA phases) `reps` times
// each; returns the number of C words that ever differed (all perform the same arithmetic per element, so it must be 0).
// ms_out (nullable): [single-barrier, 8-phase DMA-in-L, 8-phase DMA-in-M] mean kernel time over the repetitions, HIP events.
extern "C" int mdd_diag_gemm_ph8(int M, int N, int K, int reps, unsigned seed, unsigned *mismatches_out, float *ms_out) {
    using namespace mdd;
    if (M <= 0 || N <= 0 || K < XBK || K % XBK || !mismatches_out || reps < 1) { set_error("mdd_diag_gemm_ph8: bad shape"); return MDD_ERR_ARG; }
    static bool at = false; if (!at) { if (int rc = init_gemm_attributes()) return rc; at = true; }
    unsigned short *A = nullptr, *W = nullptr; float *C1 = nullptr, *C2 = nullptr; unsigned *cnt = nullptr;
    MDD_HIP_CHECK(hipMalloc((void **)&A, (size_t)2 * M * K * 2));
    MDD_HIP_CHECK(hipMalloc((void **)&W, (size_t)2 * N * K * 2));
    MDD_HIP_CHECK(hipMalloc((void **)&C1, (size_t)M * N * 4));
    MDD_HIP_CHECK(hipMalloc((void **)&C2, (size_t)M * N * 4));
    MDD_HIP_CHECK(hipMalloc((void **)&cnt, 4));
    MDD_HIP_CHECK(hipMemset(cnt, 0, 4));
    hipEvent_t e0, e1; MDD_HIP_CHECK(hipEventCreate(&e0)); MDD_HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(diag_fill_kernel, dim3(1024), dim3(256), 0, nullptr, A, (size_t)2 * M * K, seed);
    hipLaunchKernelGGL(diag_fill_kernel, dim3(1024), dim3(256), 0, nullptr, W, (size_t)2 * N * K, seed * 7919u + 13u);
    const int tn = (N + 255) / 256;
    const dim3 grid(((M + 255) / 256) * tn), block(512);
    float ms[3] = {0.f, 0.f, 0.f};
    for (int form = 0; form < 3; form++) {
        for (int r = 0; r < reps; r++) {
            float *dst = form == 0 ? C1 : C2;
            if (form) MDD_HIP_CHECK(hipMemsetAsync(C2, 0xff, (size_t)M * N * 4, nullptr));
            MDD_HIP_CHECK(hipEventRecord(e0, nullptr));
            if (form == 0) hipLaunchKernelGGL(gemm_bf16x3_glds256_kernel, grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                                              (const float *)nullptr, dst, M, N, K, K, K, N, tn);
            else if (form == 1) hipLaunchKernelGGL(gemm_bf16x3_ph8_kernel<false>, grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                                                   (const float *)nullptr, dst, M, N, K, K, K, N, tn);
            else hipLaunchKernelGGL(gemm_bf16x3_ph8_kernel<true>, grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                                    (const float *)nullptr, dst, M, N, K, K, K, N, tn);
            MDD_HIP_CHECK(hipEventRecord(e1, nullptr));
            MDD_HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f; MDD_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0 || reps == 1) ms[form] += t;
            if (form) hipLaunchKernelGGL(diag_diff_kernel, dim3(1024), dim3(256), 0, nullptr, reinterpret_cast<const unsigned *>(C1),
                                         reinterpret_cast<const unsigned *>(C2), (size_t)M * N, cnt);
        }
        ms[form] /= (float)(reps > 1 ? reps - 1 : 1);
    }
    MDD_HIP_CHECK(hipMemcpy(mismatches_out, cnt, 4, hipMemcpyDeviceToHost));
    if (ms_out) { ms_out[0] = ms[0]; ms_out[1] = ms[1]; ms_out[2] = ms[2]; }
    if (ms_out && getenv("MDD_GEMM_AFIRST")) {   // the 8-phase kernel with the A fragment as first MFMA operand (scalar C stores): ms_out[11]
        float tot = 0.f;
        for (int r = 0; r < reps; r++) {
            MDD_HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL((gemm_bf16x3_ph8_kernel<false, false, false>), grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                               (const float *)nullptr, C2, M, N, K, K, K, N, tn, (long long *)nullptr);
            MDD_HIP_CHECK(hipEventRecord(e1, nullptr));
            MDD_HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f; MDD_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) tot += t;
        }
        ms_out[11] = tot / (float)(reps - 1);
    }
    if (ms_out && getenv("MDD_GEMM_T128")) {   // ms_out[14]: the 128x128 kernel (two workgroups per CU) on the same problem
        const int tm1 = (M + XBM - 1) / XBM, tn1 = (N + XBN - 1) / XBN;
        float tot = 0.f;
        for (int r = 0; r < reps; r++) {
            MDD_HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(gemm_bf16x3_glds_kernel<0>, dim3(tm1 * tn1), dim3(256), 0, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K, (const float *)nullptr, C2,
                               (unsigned short *)nullptr, (unsigned short *)nullptr, M, N, K, K, K, N, 0l, 0l, 0l, tn1);
            MDD_HIP_CHECK(hipEventRecord(e1, nullptr));
            MDD_HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f; MDD_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) tot += t;
        }
        ms_out[14] = tot / (float)(reps - 1);
    }
    if (ms_out && getenv("MDD_GEMM_NOSTORE")) {   // ms_out[12]: the 8-phase kernel without its C stores
        float tot = 0.f;
        for (int r = 0; r < reps; r++) {
            MDD_HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL((gemm_bf16x3_ph8_kernel<false, true>), grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                               (const float *)nullptr, C2, M, N, K, K, K, N, tn, (long long *)nullptr);
            MDD_HIP_CHECK(hipEventRecord(e1, nullptr));
            MDD_HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f; MDD_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) tot += t;
        }
        ms_out[12] = tot / (float)(reps - 1);
    }
    if (ms_out && getenv("MDD_GEMM_STAMP")) {   // phase stamps of both forms: ms_out[3..6] (DMA in L) and [7..10] (DMA in M), mean cycles per K-tile and wave
        long long *sd = nullptr;
        MDD_HIP_CHECK(hipMalloc((void **)&sd, sizeof(long long) * 256 * 8 * 4));
        std::vector<long long> hs(256 * 8 * 4);
        for (int form = 0; form < 2; form++) {
            MDD_HIP_CHECK(hipMemset(sd, 0, sizeof(long long) * 256 * 8 * 4));
            if (form == 0) hipLaunchKernelGGL((gemm_bf16x3_ph8_kernel<false, true>), grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                                              (const float *)nullptr, C2, M, N, K, K, K, N, tn, sd);
            else hipLaunchKernelGGL((gemm_bf16x3_ph8_kernel<true, true>), grid, block, 2 * 4 * 256 * XROW, nullptr, A, A + (size_t)M * K, W, W + (size_t)N * K,
                                    (const float *)nullptr, C2, M, N, K, K, K, N, tn, sd);
            MDD_HIP_CHECK(hipMemcpy(hs.data(), sd, sizeof(long long) * 256 * 8 * 4, hipMemcpyDeviceToHost));
            const int nwg = std::min(256, (int)grid.x);
            for (int i = 0; i < 4; i++) {
                double tot = 0;
                for (int w = 0; w < nwg * 8; w++) tot += (double)hs[(size_t)w * 4 + i];
                ms_out[3 + form * 4 + i] = (float)(tot / (nwg * 8) / (K / XBK));
            }
        }
        (void)hipFree(sd);
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C1); (void)hipFree(C2); (void)hipFree(cnt); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

namespace mdd {

// fp32 [n] -> hi/lo planes (used for weights at load time and by the tap / test helpers)
__global__ void split_kernel(const float *__restrict__ x, size_t n, unsigned short *__restrict__ hi, unsigned short *__restrict__ lo) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const unsigned short h = bf16_bits(v);
        hi[i] = h;
        lo[i] = bf16_bits(v - bf16_to_f32(h));
    }
}
__global__ void unsplit_kernel(const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, size_t n, float *__restrict__ x) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = bf16_to_f32(hi[i]) + bf16_to_f32(lo[i]);
}

int launch_split(const float *x, size_t n, const SplitPtr &out, hipStream_t st) {
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, st, x, n, out.hi, out.lo);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
int launch_unsplit(const SplitPtr &in, size_t n, float *x, hipStream_t st) {
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(unsplit_kernel, dim3(grid), dim3(256), 0, st, in.hi, in.lo, n, x);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
