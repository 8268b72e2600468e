// "f32x6": fp32-grade NT GEMM on the bf16 matrix cores.  C[M,N] = A[M,K] . W[N,K]^T (+ bias).
//
// gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 rate (it executes on the fp32 vector lanes).  An fp32 value is,
// exactly, the sum of THREE bf16 numbers -- hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 significand bits = the
// 24 of fp32 -- so a product of two fp32 values is the sum of nine bf16 x bf16 products, each exact in fp32.  The six largest
// (hi.hi; hi.mid, mid.hi; hi.lo, lo.hi, mid.mid) carry everything down to 2^-24 of the product, i.e. to fp32's own rounding; the
// three dropped ones (mid.lo, lo.mid, lo.lo) are below it.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA they replace.
// What decides the accuracy is not the dropped terms but the ORDER of accumulation: hi.hi is summed in an accumulator of its own
// (one rounding per 32 products) and the five small products in a second one (their roundings happen at 2^-8 of the scale),
// combined once at the end.  Emulated and measured (tests/test_gpu_parity.py::test_gemm_f32x6_accuracy): closer to the float64
// product than ATen's fp32 GEMM on the CPU and ~3x closer than the exact-fp32 MFMA GEMM, which is one rounding per 2 products.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mdd_internal.h"

namespace mdd {

__device__ __forceinline__ unsigned short bf16_rn(float x) { __bf16 b = (__bf16)x; return *reinterpret_cast<unsigned short *>(&b); }
__device__ __forceinline__ float bf16_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// x [rows][ld] fp32 (K leading columns used) -> three bf16 planes (hi, mid, lo) with hi + mid + lo == x exactly (finite x; overflow /
// NaN land in hi), each plane in the K-TILE-MAJOR order the f32x6 kernel streams: plane[kt][row][32] (kt = k / 32), so that the 16 rows x
// 64 bytes one LDS-DMA instruction moves are 1 KB of CONTIGUOUS memory (eight whole 128-byte lines, every byte used).  With row-major
// planes the same instruction touched 16 half-lines, and the kernel ran at the rate a CU ingests lines from L2 (~25 useful B/clk).
// One wave per (16-row group, K-tile): lane (row = lane / 4, chunk = lane % 4) reads 8 floats, writes 16 bytes per plane.
__global__ void split3_kernel(const float *__restrict__ x, int rows, int K, int ld, unsigned short *__restrict__ planes, size_t plane_elems) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int nkt = K / 32, ngr = (rows + 15) / 16;
    const size_t total = (size_t)ngr * nkt;
    for (size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < total; w += (size_t)gridDim.x * (blockDim.x >> 6)) {
        const int g = (int)(w / nkt), kt = (int)(w - (size_t)g * nkt);
        const int row = g * 16 + (lane >> 2), c = lane & 3;
        if (row >= rows) continue;
        const float4 *src = reinterpret_cast<const float4 *>(x + (size_t)row * ld + kt * 32 + c * 8);
        const float4 v0 = src[0], v1 = src[1];
        const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        u16x8 h, m, l;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            h[k] = bf16_rn(f[k]);
            const float r1 = f[k] - bf16_f32(h[k]);
            m[k] = bf16_rn(r1);
            l[k] = bf16_rn(r1 - bf16_f32(m[k]));
        }
        const size_t off = ((size_t)kt * rows + row) * 32 + c * 8;
        *reinterpret_cast<u16x8 *>(planes + off) = h;
        *reinterpret_cast<u16x8 *>(planes + plane_elems + off) = m;
        *reinterpret_cast<u16x8 *>(planes + 2 * plane_elems + off) = l;
    }
}

// planes: 3 x rows x K elements (hi | mid | lo, each K-tile-major)
int launch_split3(const float *x, int rows, int K, int ld, unsigned short *planes, hipStream_t st) {
    if (rows <= 0 || K <= 0 || K % 32 || ld % 4) { set_error("split3: rows=%d K=%d ld=%d (K a multiple of 32, ld of 4)", rows, K, ld); return MDD_ERR_ARG; }
    const size_t waves = (size_t)((rows + 15) / 16) * (K / 32);
    int grid = (int)((waves + 3) / 4); if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, st, x, rows, K, ld, planes, (size_t)rows * K);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// (prototype only) row-major three-plane split, for the x3 kernel's operand layout
__global__ void split3_rowmajor_kernel(const float *__restrict__ x, size_t n, unsigned short *__restrict__ p0, unsigned short *__restrict__ p1, unsigned short *__restrict__ p2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const unsigned short h = bf16_rn(v);
        const float r1 = v - bf16_f32(h);
        const unsigned short m = bf16_rn(r1);
        p0[i] = h; p1[i] = m; p2[i] = bf16_rn(r1 - bf16_f32(m));
    }
}
static int launch_split3_rowmajor(const float *x, size_t n, unsigned short *planes, hipStream_t st) {
    hipLaunchKernelGGL(split3_rowmajor_kernel, dim3(4096), dim3(256), 0, st, x, n, planes, planes + n, planes + 2 * n);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

__global__ void add3_kernel(float *__restrict__ c, const float *__restrict__ s1, const float *__restrict__ s2, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = c[i] + (s1[i] + s2[i]);
}

// ---- the kernel: 192 x 128 tile of C per workgroup, FOUR waves (one per SIMD, 512 registers each) of 96 x 64, K-tile 32.
// Three accumulator sets of 6 x 4 MFMA tiles per wave -- hh (hi.hi), sm (the five small products), tot (hh flushed into it every
// FLUSH K-tiles, which keeps the hh chain at FLUSH roundings per segment: at K = 1952 one chain of 61 was measured at 1.3x ATen's
// error, segments of 8 are below it) -- are what the one-wave-per-SIMD shape is for: 288 accumulator registers fit 512, not 256.
// Operands stream HBM -> LDS with LDS-DMA (16 B per lane, XOR-swizzled on the source side so that fragment reads are
// conflict-free: the layout of gemm_bf16x3.hip), two stages of 60 KB; a K-tile's 15 pieces per wave go out three at a time between the
// row tiles of the K-tile before; one workgroup barrier per K-tile.  The W fragment is the MFMA's first operand, so a lane's four
// accumulator registers are four consecutive columns of one C row and the tile leaves as 16-byte stores.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void6;
constexpr int X6_RT = 6;                                              // 16-row MFMA tiles per wave: 6 x 4 tiles x 3 accumulator sets = 288 registers (8 would need 384 + fragments: spills)
constexpr int X6_BM = 2 * 16 * X6_RT, X6_BN = 128, X6_BK = 32, X6_ROW = 64;   // X6_ROW: bytes per LDS row (32 bf16)
constexpr int X6_GA = X6_BM / 16 / 4, X6_NP = 3 * X6_GA + 6;         // A row groups (16 rows) per wave; LDS-DMA pieces per wave and K-tile
constexpr int X6_PA = X6_BM * X6_ROW, X6_PW = X6_BN * X6_ROW;          // bytes per A / W plane of a stage
constexpr int X6_STAGE = 3 * X6_PA + 3 * X6_PW;                       // 72 KB
constexpr int X6_FLUSH = 8;

__device__ __forceinline__ bf16x8 x6_frag(const unsigned char *plane, int row, int kbyte) {
    return *reinterpret_cast<const bf16x8 *>(plane + row * X6_ROW + ((((kbyte >> 4) ^ ((row >> 2) & 3))) << 4));
}

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void gemm_f32x6_kernel(const unsigned short *__restrict__ Ap, const unsigned short *__restrict__ Wp, size_t a_plane, size_t w_plane,
                                                            const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K, int ldc, int tiles_n,
                                                            long long *stamps) {
    // STAMP (diagnostic instantiation): per wave, cycles of the K loop spent issuing the next stage + reading the first fragments / in the
    // MFMA section / waiting for the LDS-DMA / at the barrier -> stamps[(workgroup * 4 + wave) * 4 + {0..3}] (tools/gemm_time.py)
    long long sacc[4] = {0, 0, 0, 0}, stt = 0;
#define X6_T(i_) do { if (STAMP) { const long long n_ = (long long)__builtin_readcyclecounter(); sacc[i_] += n_ - stt; stt = n_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int nwg = gridDim.x, bid = blockIdx.x;
    int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);   // an XCD walks consecutive tiles: the A panel stays in its L2
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int m0 = tm * X6_BM, n0 = tn * X6_BN;
    f32x4 hh[X6_RT][4], sm[X6_RT][4], tot[X6_RT][4];
#pragma unroll
    for (int i = 0; i < X6_RT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; sm[i][j] = hh[i][j]; tot[i][j] = hh[i][j]; }
    // LDS-DMA pieces of this wave: A row groups GA*w .. GA*w+GA-1 and W row groups 2w, 2w+1 (16 rows each), three planes each.  Per-lane byte
    // offsets are fixed for the whole K loop; the plane and k0 go into the scalar base.
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void6 *)smem6;
    unsigned voffA[X6_GA], voffW[2];
#pragma unroll
    for (int g = 0; g < X6_GA; g++) {
        const int row = (wave * X6_GA + g) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        voffA[g] = (unsigned)(((size_t)min(m0 + row, M - 1) * 32 + c * 8) * 2);      // K-tile-major planes: 64 bytes per row and K-tile
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int row = (wave * 2 + g) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        voffW[g] = (unsigned)(((size_t)min(n0 + row, N - 1) * 32 + c * 8) * 2);
    }
    auto piece = [&](int idx, int kt_, unsigned stage_off) {   // idx < 3 GA: A (plane idx / GA, group idx % GA); then W (plane (idx - 3 GA) / 2, group (idx - 3 GA) % 2)
        if (idx < 3 * X6_GA) {
            const int p = idx / X6_GA, g = idx % X6_GA;
            const unsigned short *base = Ap + (size_t)p * a_plane + (size_t)kt_ * M * 32;
            const unsigned la = lds0 + stage_off + (unsigned)(p * X6_PA + (wave * X6_GA + g) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[g]), "s"(base), "s"(la) : "memory", "m0");
        } else {
            const int p = (idx - 3 * X6_GA) >> 1, g = (idx - 3 * X6_GA) & 1;
            const unsigned short *base = Wp + (size_t)p * w_plane + (size_t)kt_ * N * 32;
            const unsigned la = lds0 + stage_off + (unsigned)(3 * X6_PA + p * X6_PW + (wave * 2 + g) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffW[g]), "s"(base), "s"(la) : "memory", "m0");
        }
    };
    const int nk = K / X6_BK;
    const int l16 = lane & 15, kq16 = (lane >> 4) * 16;
#pragma unroll
    for (int idx = 0; idx < X6_NP; idx++) piece(idx, 0, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // One K-tile.  Its W fragments (fwc) and the A fragments of its first row tile (fa[0]) are already in registers: they were read
    // under the LAST row tile of the K-tile before, right behind the barrier that declared this stage complete -- so neither the
    // fragment reads' LDS latency (measured 780 cycles with all four waves reading at once behind a K-tile-end barrier) nor the
    // barrier's skew stand between two K-tiles' MFMAs.  The next stage's LDS-DMA pieces go out ONE at a time, one per MFMA group of the
    // first X6_IS row tiles (a burst of pieces stalls the issuing wave, and with one wave per SIMD nobody else feeds the matrix pipe
    // meanwhile; later than that they would not have landed when the last row tile starts).
    // The two MFMA-fed accumulator sets live in the accumulation file for the whole K loop ("+a": written as asm because the compiler
    // otherwise shuttles one set between the files on every K-tile); tot is only touched by the flush's vector adds.
    bf16x8 fa[2][3];
    constexpr int X6_IS = 3;                                             // row tiles over which the next stage is requested
    static_assert(X6_IS * 6 >= X6_NP && (X6_RT % 2) == 0, "one piece per MFMA group at most; fa[0] must be free under the last row tile");
#define X6_MFMA(acc_, w_, a_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(w_), "v"(a_))
    auto ktile = [&](int kt, bf16x8 (&fwc)[3][4], bf16x8 (&fwn)[3][4]) {
        const unsigned char *st = smem6 + (kt & 1) * X6_STAGE;
        const unsigned char *nsp = smem6 + ((kt & 1) ^ 1) * X6_STAGE;
        const unsigned nst = (unsigned)(((kt & 1) ^ 1) * X6_STAGE);
        const bool more = kt + 1 < nk;
#pragma unroll
        for (int i = 0; i < X6_RT; i++) {
            if (i + 1 < X6_RT) {
#pragma unroll
                for (int p = 0; p < 3; p++) fa[(i + 1) & 1][p] = x6_frag(st + p * X6_PA, wr * (16 * X6_RT) + (i + 1) * 16 + l16, kq16);
            } else if (more) {
                X6_T(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of the next stage have landed
                X6_T(1);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's; and every wave has read all it needs of this stage
                X6_T(2);
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int j = 0; j < 4; j++) fwn[p][j] = x6_frag(nsp + 3 * X6_PA + p * X6_PW, wc * 64 + j * 16 + l16, kq16);
#pragma unroll
                for (int p = 0; p < 3; p++) fa[0][p] = x6_frag(nsp + p * X6_PA, wr * (16 * X6_RT) + l16, kq16);
            }
            const bf16x8 ah = fa[i & 1][0], am = fa[i & 1][1], al = fa[i & 1][2];
            // small products first (smallest first), hi.hi last and into its own accumulator; product-major so that consecutive MFMAs
            // write different accumulators
#define X6_PIECE(g_) do { if (i < X6_IS) { const int pi_ = (i * 6 + (g_)) * X6_NP / (X6_IS * 6), pn_ = (i * 6 + (g_) + 1) * X6_NP / (X6_IS * 6); if (more && pn_ > pi_) piece(pi_, kt + 1, nst); } } while (0)
            X6_PIECE(0);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(sm[i][j], fwc[1][j], am);
            X6_PIECE(1);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(sm[i][j], fwc[2][j], ah);
            X6_PIECE(2);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(sm[i][j], fwc[0][j], al);
            X6_PIECE(3);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(sm[i][j], fwc[1][j], ah);
            X6_PIECE(4);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(sm[i][j], fwc[0][j], am);
            X6_PIECE(5);
#pragma unroll
            for (int j = 0; j < 4; j++) X6_MFMA(hh[i][j], fwc[0][j], ah);
#undef X6_PIECE
        }
    };
    bf16x8 fwA[3][4], fwB[3][4];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int j = 0; j < 4; j++) fwA[p][j] = x6_frag(smem6 + 3 * X6_PA + p * X6_PW, wc * 64 + j * 16 + l16, kq16);
#pragma unroll
    for (int p = 0; p < 3; p++) fa[0][p] = x6_frag(smem6 + p * X6_PA, wr * (16 * X6_RT) + l16, kq16);
    if (STAMP) stt = (long long)__builtin_readcyclecounter();
    // Segments of X6_FLUSH K-tiles with the flush of hh into tot BETWEEN them (as a conditional inside one K loop the compiler kept a copy
    // of hh in ordinary registers across every K-tile: 96 v_accvgpr_read + 96 v_accvgpr_write per 144 MFMAs); K-tiles in pairs, so that
    // the two W fragment sets swap roles without a register copy.
    static_assert(X6_FLUSH % 2 == 0, "K-tiles are taken in pairs");
    for (int kt0 = 0; kt0 < nk; kt0 += X6_FLUSH) {
        const int kt1 = min(kt0 + X6_FLUSH, nk);
        for (int kt = kt0; kt < kt1; kt += 2) {
            ktile(kt, fwA, fwB);
            if (kt + 1 < kt1) ktile(kt + 1, fwB, fwA);
            else {   // an odd tail: the loop ends here (kt1 == nk); nothing follows that would read fwA / fwB in the wrong role
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                     // the asm MFMAs' results are read by vector instructions next: their wait states by hand
#pragma unroll
        for (int i = 0; i < X6_RT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { tot[i][j] += hh[i][j]; hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
#undef X6_MFMA
    X6_T(3);
    if (STAMP && stamps && lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 4; i++) stamps[((size_t)blockIdx.x * 4 + wave) * 4 + i] = sacc[i];
#undef X6_T
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const int q4 = (lane >> 4) * 4;   // D row = 4 * (lane >> 4) + r = C column, D col = lane & 15 = C row
#pragma unroll
    for (int i = 0; i < X6_RT; i++) {
        const int row = m0 + wr * (16 * X6_RT) + i * 16 + l16;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = n0 + wc * 64 + j * 16 + q4;
            if (row >= M || col >= N) continue;
            f32x4 v = tot[i][j] + sm[i][j];
            if (bias) { v[0] += bias[col]; if (col + 1 < N) v[1] += bias[col + 1]; if (col + 2 < N) v[2] += bias[col + 2]; if (col + 3 < N) v[3] += bias[col + 3]; }
            float *dst = C + (size_t)row * ldc + col;
            if (col + 3 < N) *reinterpret_cast<f32x4 *>(dst) = v;
            else for (int r = 0; r < 4 && col + r < N; r++) dst[r] = v[r];
        }
    }
}

int init_gemm_x6_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_f32x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X6_STAGE));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_f32x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X6_STAGE));
    return MDD_OK;
}

// A3, W3: three consecutive K-tile-major bf16 planes (hi | mid | lo; launch_split3) of a_plane = M x K / w_plane = N x K elements each
int launch_gemm_f32x6(const unsigned short *A3, size_t a_plane, const unsigned short *W3, size_t w_plane, const float *bias, float *C, int M, int N, int K,
                      int ldc, hipStream_t st, long long *stamps) {
    if (M <= 0 || N <= 0 || K <= 0 || K % X6_BK || ldc % 4 || a_plane != (size_t)M * K || w_plane != (size_t)N * K) {
        set_error("gemm_f32x6: bad shape M=%d N=%d K=%d ldc=%d", M, N, K, ldc);
        return MDD_ERR_ARG;
    }
    const int tn = (N + X6_BN - 1) / X6_BN;
    const dim3 grid(((M + X6_BM - 1) / X6_BM) * tn);
    if (stamps) hipLaunchKernelGGL(gemm_f32x6_kernel<true>, grid, dim3(256), 2 * X6_STAGE, st, A3, W3, a_plane, w_plane, bias, C, M, N, K, ldc, tn, stamps);
    else hipLaunchKernelGGL(gemm_f32x6_kernel<false>, grid, dim3(256), 2 * X6_STAGE, st, A3, W3, a_plane, w_plane, bias, C, M, N, K, ldc, tn, (long long *)nullptr);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd

using namespace mdd;

// Diagnostic / test entry: one GEMM through a chosen arithmetic, fp32 operands and result on the device.
//   mode 0: exact fp32 MFMA (gemm_nt_f32_kernel)      1: split-bf16 x3 (operands split here)
//   mode 3: the f32x6 kernel (operands split into three planes here)
//   mode 2: f32x6 PROTOTYPE -- the six products through three launches of the x3 kernel (hi.hi alone; {hi.lo, mid.mid, hi.mid};
//           {mid.hi, lo.hi}) and one combine pass: the arithmetic of the x6 kernel at none of its speed
extern "C" int mdd_diag_gemm(int mode, const float *A_dev, const float *W_dev, float *C_dev, int M, int N, int K, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!A_dev || !W_dev || !C_dev || M <= 0 || N <= 0 || K <= 0 || K % 32) { set_error("mdd_diag_gemm: bad arguments"); return MDD_ERR_ARG; }
    if (mode == 0) return launch_gemm_nt(A_dev, W_dev, nullptr, C_dev, M, N, K, K, K, N, 1, 0, 0, 0, st);
    const size_t na = (size_t)M * K, nw = (size_t)N * K, nc = (size_t)M * N;
    unsigned short *pa = nullptr, *pw = nullptr, *zero = nullptr;
    float *t1 = nullptr, *t2 = nullptr;
    int rc = MDD_OK;
    const size_t nz = na > nw ? na : nw;
    if (hipMalloc((void **)&pa, 3 * na * 2) != hipSuccess || hipMalloc((void **)&pw, 3 * nw * 2) != hipSuccess || hipMalloc((void **)&zero, nz * 2) != hipSuccess ||
        hipMalloc((void **)&t1, nc * 4) != hipSuccess || hipMalloc((void **)&t2, nc * 4) != hipSuccess) { set_error("mdd_diag_gemm: out of memory"); rc = MDD_ERR_NOMEM; }
    if (!rc && hipMemsetAsync(zero, 0, nz * 2, st) != hipSuccess) rc = MDD_ERR_HIP;
    if (!rc) {
        if (mode == 1) {
            SplitPtr a{pa, pa + na}, w{pw, pw + nw};
            if (!(rc = launch_split(A_dev, na, a, st)) && !(rc = launch_split(W_dev, nw, w, st)))
                rc = launch_gemm_bf16x3(a, w, nullptr, C_dev, nullptr, M, N, K, K, K, N, 1, 0, 0, 0, st);
        } else if (mode == 2) {
            rc = launch_split3_rowmajor(A_dev, na, pa, st);
            if (!rc) rc = launch_split3_rowmajor(W_dev, nw, pw, st);
            // launch_gemm_bf16x3(A = (X, Y), W = (U, V)) computes X.V + Y.U + X.U
            SplitPtr a_hh{pa, zero}, w_hh{pw, zero};                                   // hi.hi
            SplitPtr a_s1{pa, pa + na}, w_s1{pw + nw, pw + 2 * nw};                   // X=Ah Y=Am U=Wm V=Wl: Ah.Wl + Am.Wm + Ah.Wm
            SplitPtr a_s2{pa + 2 * na, pa + na}, w_s2{pw, zero};                      // X=Al Y=Am U=Wh V=0 : Am.Wh + Al.Wh
            if (!rc) rc = launch_gemm_bf16x3(a_hh, w_hh, nullptr, C_dev, nullptr, M, N, K, K, K, N, 1, 0, 0, 0, st);
            if (!rc) rc = launch_gemm_bf16x3(a_s1, w_s1, nullptr, t1, nullptr, M, N, K, K, K, N, 1, 0, 0, 0, st);
            if (!rc) rc = launch_gemm_bf16x3(a_s2, w_s2, nullptr, t2, nullptr, M, N, K, K, K, N, 1, 0, 0, 0, st);
            if (!rc) { hipLaunchKernelGGL(add3_kernel, dim3(2048), dim3(256), 0, st, C_dev, t1, t2, nc); if (hipGetLastError() != hipSuccess) rc = MDD_ERR_HIP; }
        } else if (mode == 3) {
            static bool attr = false;
            if (!attr) { rc = init_gemm_x6_attributes(); attr = true; }
            if (!rc) rc = launch_split3(A_dev, M, K, K, pa, st);
            if (!rc) rc = launch_split3(W_dev, N, K, K, pw, st);
            if (!rc) rc = launch_gemm_f32x6(pa, na, pw, nw, nullptr, C_dev, M, N, K, N, st, nullptr);
        } else { set_error("mdd_diag_gemm: mode %d", mode); rc = MDD_ERR_ARG; }
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(pa); (void)hipFree(pw); (void)hipFree(zero); (void)hipFree(t1); (void)hipFree(t2);
    return rc;
}

namespace mdd {
__global__ void fill_pattern_kernel(float *x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * ((h >> 24) & 1 ? 1.f : 0.05f);
    }
}
}  // namespace mdd

// Timing aid: `reps` launches of one GEMM kernel (operands resident and pre-split; events on the launch stream) -> mean ms.
//   mode 0 exact fp32 MFMA, 1 split-bf16 x3, 3 f32x6
extern "C" int mdd_diag_gemm_time(int mode, int M, int N, int K, int reps, float *ms_out) {
    // MDD_GEMM_STAMP (mode 3): one extra launch of the stamped instantiation; per-K-tile cycle means of the first 1024 workgroups are printed
    const bool want_stamps = mode == 3 && getenv("MDD_GEMM_STAMP") != nullptr;
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 || reps < 1 || !ms_out) { set_error("mdd_diag_gemm_time: bad arguments"); return MDD_ERR_ARG; }
    const size_t na = (size_t)M * K, nw = (size_t)N * K, nc = (size_t)M * N;
    float *A = nullptr, *W = nullptr, *Cm = nullptr;
    unsigned short *pa = nullptr, *pw = nullptr;
    int rc = MDD_OK;
    if (hipMalloc((void **)&A, na * 4) != hipSuccess || hipMalloc((void **)&W, nw * 4) != hipSuccess || hipMalloc((void **)&Cm, nc * 4) != hipSuccess ||
        hipMalloc((void **)&pa, 3 * na * 2) != hipSuccess || hipMalloc((void **)&pw, 3 * nw * 2) != hipSuccess) { set_error("mdd_diag_gemm_time: out of memory"); rc = MDD_ERR_NOMEM; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!rc) {
        hipLaunchKernelGGL(fill_pattern_kernel, dim3(4096), dim3(256), 0, 0, A, na, 1u);
        hipLaunchKernelGGL(fill_pattern_kernel, dim3(4096), dim3(256), 0, 0, W, nw, 2u);
        SplitPtr a{pa, pa + na}, w{pw, pw + nw};
        if (mode == 1) { rc = launch_split(A, na, a, 0); if (!rc) rc = launch_split(W, nw, w, 0); }
        if (mode == 3) { rc = init_gemm_x6_attributes(); if (!rc) rc = launch_split3(A, M, K, K, pa, 0); if (!rc) rc = launch_split3(W, N, K, K, pw, 0); }
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int r = -2; r < reps && !rc; r++) {
            if (r == 0) (void)hipEventRecord(e0, 0);
            if (mode == 0) rc = launch_gemm_nt(A, W, nullptr, Cm, M, N, K, K, K, N, 1, 0, 0, 0, 0);
            else if (mode == 1) rc = launch_gemm_bf16x3(a, w, nullptr, Cm, nullptr, M, N, K, K, K, N, 1, 0, 0, 0, 0);
            else if (mode == 3) rc = launch_gemm_f32x6(pa, na, pw, nw, nullptr, Cm, M, N, K, N, 0, nullptr);
            else { set_error("mdd_diag_gemm_time: mode %d", mode); rc = MDD_ERR_ARG; }
        }
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) rc = rc ? rc : MDD_ERR_HIP;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (want_stamps && !rc) {
            long long *sd = nullptr;
            const size_t ns = (size_t)1024 * 4 * 4;
            if (hipMalloc((void **)&sd, ns * 8) == hipSuccess && hipMemset(sd, 0, ns * 8) == hipSuccess) {
                rc = launch_gemm_f32x6(pa, na, pw, nw, nullptr, Cm, M, N, K, N, 0, sd);
                std::vector<long long> h(ns);
                (void)hipMemcpy(h.data(), sd, ns * 8, hipMemcpyDeviceToHost);
                double sum[4] = {0, 0, 0, 0}; size_t cnt = 0;
                for (size_t w_ = 0; w_ < 1024 * 4; w_++) if (h[w_ * 4 + 0] > 0) { for (int i = 0; i < 4; i++) sum[i] += (double)h[w_ * 4 + i]; cnt++; }
                const double d = (double)cnt * (K / 32);
                if (cnt) printf("  f32x6 stamps, cycles per K-tile and wave: MFMA stream %.0f (ideal %d), LDS-DMA wait %.0f, barrier %.0f\n",
                                (sum[0] + sum[3]) / d, X6_RT * 4 * 6 * 16, sum[1] / d, sum[2] / d);
                fflush(stdout);
            }
            (void)hipFree(sd);
        }
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(Cm); (void)hipFree(pa); (void)hipFree(pw);
    return rc;
}
