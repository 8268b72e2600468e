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

// ---- the kernel: 192 x 128 tile of C per workgroup, FOUR waves (one per SIMD, 512 registers each), K-tile 32.
// A wave owns 48 rows x all 128 columns: 3 x 8 MFMA tiles x THREE accumulator sets -- hh (hi.hi), sm (the five small products), tot (hh
// flushed into it every X6_FLUSH K-tiles: at K = 1952 one chain of 61 roundings was measured at 1.3x ATen's error, segments of 8
// are below it) = 288 accumulator registers, which is what the one-wave-per-SIMD shape is for.
// Operand paths (what the first forms of this kernel taught, profiles/round3_gemm_f32x6_notes.txt):
//  * A: a wave's rows are nobody else's, so its A fragments never touch LDS: nine 16-byte-per-lane global loads per K-tile straight into
//    registers, issued a whole K-tile ahead.  The planes are K-tile-major (launch_split3), so one load instruction is 1 KB of
//    contiguous memory in fragment order.
//  * W: shared by the four waves, streamed HBM/L2 -> LDS by LDS-DMA (24 KB per K-tile, six pieces per wave, XOR-swizzled on the
//    source side so that fragment reads are conflict-free), two stages; fragments are read two column tiles at a time, one pair ahead of
//    the MFMAs that use them -- never in a burst: four waves reading a K-tile's worth of fragments at once behind a barrier took
//    ~780 cycles (LDS bandwidth), a third of the K-tile's MFMA time.
//  * every memory instruction goes out alone between MFMA groups (a burst stalls the issuing wave, and with one wave per SIMD nobody
//    else feeds the matrix pipe meanwhile); one vmcnt(0) + barrier per K-tile, a whole K-tile after the requests.
// The two MFMA-fed accumulator sets live in the accumulation file ("+a" asm MFMAs: the compiler otherwise shuttles a set between the
// files on every K-tile) and the flush sits between two loops, not in a conditional inside one (same reason).  The W fragment is the
// MFMA's first operand, so a lane's four accumulator registers are four consecutive columns of one C row: 16-byte stores.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void6;
constexpr int X6_RT = 3;                                              // 16-row MFMA tiles per wave
constexpr int X6_BM = 4 * 16 * X6_RT, X6_BN = 128, X6_BK = 32, X6_ROW = 64;   // X6_ROW: bytes per LDS row (32 bf16)
constexpr int X6_PW = X6_BN * X6_ROW;                                 // bytes per W plane of a stage (8 KB)
constexpr int X6_STAGE = 3 * X6_PW;                                   // 24 KB
constexpr int X6_NS = 2;                                              // stages
constexpr int X6_NPW = 3 * (X6_BN / 16) / 4;                          // LDS-DMA pieces per wave and K-tile (6)
constexpr int X6_FLUSH = 8;

__device__ __forceinline__ bf16x8 x6_frag(const unsigned char *plane, int row, int kbyte) {
    return *reinterpret_cast<const bf16x8 *>(plane + row * X6_ROW + ((((kbyte >> 4) ^ ((row >> 2) & 3))) << 4));
}

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void gemm_f32x6_kernel(const unsigned short *__restrict__ Ap, const unsigned short *__restrict__ Wp, size_t a_plane, size_t w_plane,
                                                            const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K, int ldc, int tiles_n,
                                                            int ntiles, long long *stamps) {
    // STAMP (diagnostic instantiation): per wave, cycles of the K loop in the MFMA stream / waiting for memory / at the barrier
    // -> stamps[(workgroup * 4 + wave) * 4 + {0, 1, 2}] (tools/gemm_time.py)
    long long sacc[4] = {0, 0, 0, 0}, stt = 0;
#define X6_T(i_) do { if (STAMP) { const long long n_ = (long long)__builtin_readcyclecounter(); sacc[i_] += n_ - stt; stt = n_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PERSISTENT: a workgroup walks the tiles vb = blockIdx.x, + gridDim.x, ... (one workgroup per CU: no dispatch between tiles, and the next
    // tile's first operands are requested before this tile's epilogue).  Tile of a virtual block index: an XCD (vb % 8) walks consecutive
    // tiles, so that the A panel stays in its L2.
    const int tq = ntiles >> 3, trem = ntiles & 7;
    auto tile_of = [&](int vb, int &m0_, int &n0_) {
        const int xcd = vb & 7;
        const int swz = (xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq) + (vb >> 3);
        const int tm = swz / tiles_n, tn = swz % tiles_n;
        m0_ = tm * X6_BM + wave * (16 * X6_RT); n0_ = tn * X6_BN;  // this wave's first row; the workgroup's first column
    };
    int vb = blockIdx.x, m0, n0;
    tile_of(vb, m0, n0);
    f32x4 hh[X6_RT][8], sm[X6_RT][8], tot[X6_RT][8];
    const int l16 = lane & 15, kq = lane >> 4;
    // A fragments straight from the K-tile-major planes: lane (row l16 of row tile i, k-slice kq) reads 16 bytes at ((kt * M + row) * 32 + kq * 8)
    // elements; rows past M are clamped (their C rows are never stored)
    unsigned aoff[X6_RT];
    auto load_a = [&](bf16x8 (&fa)[X6_RT][3], int kt) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const unsigned char *base = reinterpret_cast<const unsigned char *>(Ap + (size_t)p * a_plane + (size_t)kt * M * 32);
#pragma unroll
            for (int i = 0; i < X6_RT; i++) fa[i][p] = *reinterpret_cast<const bf16x8 *>(base + aoff[i]);
        }
    };
    // W: LDS-DMA pieces of this wave = row groups 2w, 2w+1 (16 rows each) of the three planes
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void6 *)smem6;
    unsigned voffW[2];
    auto set_tile = [&]() {   // the per-lane offsets of the tile at (m0, n0)
#pragma unroll
        for (int i = 0; i < X6_RT; i++) aoff[i] = (unsigned)(((size_t)min(m0 + i * 16 + l16, M - 1) * 32 + kq * 8) * 2);
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int row = (wave * 2 + g) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            voffW[g] = (unsigned)(((size_t)min(n0 + row, N - 1) * 32 + c * 8) * 2);
        }
    };
    set_tile();
    auto piece = [&](int idx, int kt_, unsigned stage_off) {   // idx 0..5: plane idx / 2, row group idx % 2
        const int p = idx >> 1, g = idx & 1;
        const unsigned short *base = Wp + (size_t)p * w_plane + (size_t)kt_ * N * 32;
        const unsigned la = lds0 + stage_off + (unsigned)(p * X6_PW + (wave * 2 + g) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffW[g]), "s"(base), "s"(la) : "memory", "m0");
    };
    const int nk = K / X6_BK;
    const int kq16 = kq * 16;
    bf16x8 faA[X6_RT][3], faB[X6_RT][3];
#pragma unroll
    for (int idx = 0; idx < X6_NPW; idx++) piece(idx, 0, 0u);
    load_a(faA, 0);
#define X6_MFMA(acc_, w_, a_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(w_), "v"(a_))
    // One K-tile: fac holds its A fragments, fan receives the next K-tile's.  Column tiles in pairs (jp): the pair's six W fragments are read
    // from LDS one pair ahead; per pair 3 rows x 2 columns x 6 products = 36 MFMAs, product-major (six different accumulators in a row),
    // smallest products first, hi.hi last and into its own accumulator.
    auto ktile = [&](int kt, bf16x8 (&fac)[X6_RT][3], bf16x8 (&fan)[X6_RT][3]) {
        const unsigned char *st = smem6 + (kt % X6_NS) * X6_STAGE;
        const unsigned nst = (unsigned)(((kt + 1) % X6_NS) * X6_STAGE);
        // (the last K-tile requests itself again instead of nothing: a branch in front of each of the 15 memory instructions cost more than
        // one K-tile's worth of redundant, never-read loads per 61)
        const int ktn = min(kt + 1, nk - 1);
        bf16x8 fw[2][3][2];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++) fw[0][p][jj] = x6_frag(st + p * X6_PW, jj * 16 + l16, kq16);
#pragma unroll
        for (int jp = 0; jp < 4; jp++) {
            if (jp + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) fw[(jp + 1) & 1][p][jj] = x6_frag(st + p * X6_PW, ((jp + 1) * 2 + jj) * 16 + l16, kq16);
            }
            // memory instructions of the next K-tile, one per product group: 9 A fragment loads (pairs 0, 1), 6 W pieces (pairs 1, 2)
#define X6_MEM(g_) do { const int s_ = jp * 6 + (g_); \
                if (s_ < 9) { const int p_ = s_ / X6_RT, i_ = s_ % X6_RT; \
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fan[i_][p_]) : "v"(aoff[i_]), "s"(Ap + (size_t)p_ * a_plane + (size_t)ktn * M * 32) : "memory"); } \
                else if (s_ < 9 + X6_NPW) piece(s_ - 9, ktn, nst); } while (0)
#define X6_GROUP(acc_, wp_, ap_) _Pragma("unroll") for (int i = 0; i < X6_RT; i++) { _Pragma("unroll") for (int jj = 0; jj < 2; jj++) X6_MFMA(acc_[i][jp * 2 + jj], fw[jp & 1][wp_][jj], fac[i][ap_]); }
            X6_MEM(0); X6_GROUP(sm, 1, 1)      // mid . mid
            X6_MEM(1); X6_GROUP(sm, 2, 0)      // W lo . A hi
            X6_MEM(2); X6_GROUP(sm, 0, 2)      // W hi . A lo
            X6_MEM(3); X6_GROUP(sm, 1, 0)      // W mid . A hi
            X6_MEM(4); X6_GROUP(sm, 0, 1)      // W hi . A mid
            X6_MEM(5); X6_GROUP(hh, 0, 0)      // hi . hi
#undef X6_GROUP
#undef X6_MEM
        }
        X6_T(0);
        // The next K-tile's A fragments and this wave's W pieces (requested a K-tile ago) are here.  The A loads are asm as well: as C++
        // loads the compiler's wait-count pass, which cannot see the asm LDS-DMA pieces, guarded the next K-tile's first MFMAs with
        // vmcnt(1) / vmcnt(0) that also waited for the loads issued a few instructions earlier (a memory round trip per K-tile).
        // Nothing reads or moves the destination registers before this wait (checked in the ISA: the registers the loads write are the
        // ones the next K-tile's MFMAs read).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X6_T(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's pieces; and every wave is done reading this stage
        X6_T(2);
    };
    for (;;) {   // ---- one tile per iteration
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < X6_RT; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) { hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; sm[i][j] = hh[i][j]; tot[i][j] = hh[i][j]; }
    if (STAMP) stt = (long long)__builtin_readcyclecounter();
    // Segments of X6_FLUSH K-tiles with the flush of hh into tot BETWEEN them; K-tiles in pairs, so that the two A fragment sets swap roles
    // without a register copy.
    static_assert(X6_FLUSH % 2 == 0 && X6_NS == 2, "K-tiles are taken in pairs");
    for (int kt0 = 0; kt0 < nk; kt0 += X6_FLUSH) {
        const int kt1 = min(kt0 + X6_FLUSH, nk);
        for (int kt = kt0; kt < kt1; kt += 2) {
            ktile(kt, faA, faB);
            if (kt + 1 < kt1) ktile(kt + 1, faB, faA);
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                     // the asm MFMAs' results are read by vector instructions next: their wait states by hand
#pragma unroll
        for (int i = 0; i < X6_RT; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) { tot[i][j] += hh[i][j]; hh[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    // the next tile's first W stage and A fragments are requested before this tile's epilogue (stage 0 and faA are free: every wave has passed
    // the last K-tile's barrier)
    const int m0c = m0, n0c = n0, vbn = vb + (int)gridDim.x;
    const bool more = vbn < ntiles;
    if (more) {
        tile_of(vbn, m0, n0);
        set_tile();
#pragma unroll
        for (int idx = 0; idx < X6_NPW; idx++) piece(idx, 0, 0u);
        load_a(faA, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const int q4 = kq * 4;   // D row = 4 * (lane >> 4) + r = C column, D col = lane & 15 = C row
#pragma unroll
    for (int i = 0; i < X6_RT; i++) {
        const int row = m0c + i * 16 + l16;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int col = n0c + j * 16 + q4;
            if (row >= M || col >= N) continue;
            f32x4 v = tot[i][j] + sm[i][j];
            if (bias) { v[0] += bias[col]; if (col + 1 < N) v[1] += bias[col + 1]; if (col + 2 < N) v[2] += bias[col + 2]; if (col + 3 < N) v[3] += bias[col + 3]; }
            float *dst = C + (size_t)row * ldc + col;
            if (col + 3 < N) *reinterpret_cast<f32x4 *>(dst) = v;
            else for (int r = 0; r < 4 && col + r < N; r++) dst[r] = v[r];
        }
    }
    if (!more) break;
    vb = vbn;
    }   // ---- tiles
#undef X6_MFMA
    if (STAMP && stamps && lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 4; i++) stamps[((size_t)blockIdx.x * 4 + wave) * 4 + i] = sacc[i];
#undef X6_T
}

int init_gemm_x6_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_f32x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_NS * X6_STAGE));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_f32x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_NS * X6_STAGE));
    return MDD_OK;
}

// A3, W3: three consecutive K-tile-major bf16 planes (hi | mid | lo; launch_split3) of a_plane = M x K / w_plane = N x K elements each
int launch_gemm_f32x6(const unsigned short *A3, size_t a_plane, const unsigned short *W3, size_t w_plane, const float *bias, float *C, int M, int N, int K,
                      int ldc, hipStream_t st, long long *stamps) {
    if (M <= 0 || N <= 0 || K <= 0 || K % X6_BK || ldc % 4 || a_plane != (size_t)M * K || w_plane != (size_t)N * K) {
        set_error("gemm_f32x6: bad shape M=%d N=%d K=%d ldc=%d", M, N, K, ldc);
        return MDD_ERR_ARG;
    }
    const int tn = (N + X6_BN - 1) / X6_BN, ntiles = ((M + X6_BM - 1) / X6_BM) * tn;
    static const int grid_cap = [] { const char *e = getenv("MDD_GEMM_X6_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();   // one persistent workgroup per CU
    const dim3 grid(ntiles < grid_cap ? ntiles : grid_cap);
    if (stamps) hipLaunchKernelGGL(gemm_f32x6_kernel<true>, grid, dim3(256), X6_NS * X6_STAGE, st, A3, W3, a_plane, w_plane, bias, C, M, N, K, ldc, tn, ntiles, stamps);
    else hipLaunchKernelGGL(gemm_f32x6_kernel<false>, grid, dim3(256), X6_NS * X6_STAGE, st, A3, W3, a_plane, w_plane, bias, C, M, N, K, ldc, tn, ntiles, (long long *)nullptr);
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
                if (cnt) printf("  f32x6 stamps, cycles per K-tile and wave: MFMA stream %.0f (ideal %d), memory wait %.0f, barrier %.0f\n",
                                sum[0] / d, X6_RT * 8 * 6 * 16, sum[1] / d, sum[2] / d);
                fflush(stdout);
            }
            (void)hipFree(sd);
        }
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(Cm); (void)hipFree(pa); (void)hipFree(pw);
    return rc;
}
