// fp32 "NT" GEMM on the gfx950 matrix cores: C[M,N] = A[M,K] . W[N,K]^T (+ bias).
//
// Used for every time-batched dense contraction of the path: the BiLSTM input projections
// (reference: nn.LSTM weight_ih, AA/models/model_ctc.py:28-29,44), the text-encoder projection
// (:150,198), the `score` Linear (:151,201) and the attention scores bmm (:204).
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain), so results stay inside the 1e-4 parity
// budget; it runs at the fp32 matrix peak (157 TFLOP/s), which is the roofline of this kernel.
//
// Tiling: 128x128 block tile, BK=16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA
// tiles of 32x32 (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS with
// the next tile's loads in flight during the MFMAs (two LDS buffers, one barrier per K-tile).
// LDS rows are padded to 17 floats: the per-lane ds_read_b32 of an MFMA operand (lane = row,
// half-wave = k) is then conflict-free.
#include "mdd_internal.h"

namespace mdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDS_LD = BK + 1;

template <bool ALIGNED>
__device__ __forceinline__ void load_tile_regs(const float *__restrict__ P, int ld, int rows_total, int K, int row0,
                                               int k0, int tid, float4 (&r)[2]) {
    // thread -> (row = tid/4 [+64], 4 consecutive k at (tid%4)*4)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int row = row0 + (tid >> 2) + i * 64;
        int k = k0 + (tid & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_total) {
            const float *p = P + (size_t)row * ld + k;
            if (ALIGNED && k + 3 < K) {
                v = *reinterpret_cast<const float4 *>(p);
            } else {
                if (k < K) v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
        r[i] = v;
    }
}

__device__ __forceinline__ void store_tile_lds(float *s, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float *d = s + ((tid >> 2) + i * 64) * LDS_LD + (tid & 3) * 4;
        d[0] = r[i].x; d[1] = r[i].y; d[2] = r[i].z; d[3] = r[i].w;
    }
}

// SEG: two-level accumulation.  An MFMA accumulator is ONE rounding per two products along K; over K = 768 that chain's rounding noise
// was the largest contribution to the attention scores' (and through them the log-probs') distance from a float64 evaluation
// (tools/stage_errors.py: key 4.6e-7 of rms against ATen's 3.5e-7).  With SEG the accumulators are emptied into a second set every four
// K-tiles (32 roundings per segment, then one per segment): measured below ATen's blocked CPU GEMM.  The adds are vector
// instructions an fp32 MFMA does not hide (~6 % of the kernel), so the large input projections (N > 1024), which have their own
// fp32-grade form on the bf16 matrix cores (gemm_bf16x6.hip), keep the single chain; the choice depends on the shape only.
template <bool ALIGNED, bool SEG>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ C, int M,
                                                          int N, int K, int lda, int ldw, int ldc, long sA, long sW,
                                                          long sC, int tiles_n) {
    __shared__ float lds[2][2][BM * LDS_LD];  // [buffer][A|W][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each
    // XCD a contiguous run of tiles (they share A row-panels through that XCD's L2).
    int nwg = gridDim.x, bid = blockIdx.x;
    int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    A += (size_t)blockIdx.z * sA; W += (size_t)blockIdx.z * sW; C += (size_t)blockIdx.z * sC;

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    float4 ra[2], rw[2];
    load_tile_regs<ALIGNED>(A, lda, M, K, m0, 0, tid, ra);
    load_tile_regs<ALIGNED>(W, ldw, N, K, n0, 0, tid, rw);
    store_tile_lds(lds[0][0], tid, ra);
    store_tile_lds(lds[0][1], tid, rw);
    __syncthreads();

    const int nk = (K + BK - 1) / BK;
    const int li = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile_regs<ALIGNED>(A, lda, M, K, m0, (kt + 1) * BK, tid, ra);
            load_tile_regs<ALIGNED>(W, ldw, N, K, n0, (kt + 1) * BK, tid, rw);
        }
        const float *as = lds[cur][0] + (wm * 64 + li) * LDS_LD + lh;
        const float *ws = lds[cur][1] + (wn * 64 + li) * LDS_LD + lh;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = as[kk], a1 = as[32 * LDS_LD + kk];
            float b0 = ws[kk], b1 = ws[32 * LDS_LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile_lds(lds[cur ^ 1][0], tid, ra);
            store_tile_lds(lds[cur ^ 1][1], tid, rw);
        }
        if (SEG && (kt & 3) == 3) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int col = n0 + wn * 64 + j * 32 + li;
            if (col >= N) continue;
            float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) C[(size_t)row * ldc + col] = (SEG ? tot[i][j][r] + acc[i][j][r] : acc[i][j][r]) + bv;
            }
        }
}

int launch_gemm_nt(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int lda, int ldw,
                   int ldc, int batch, long sA, long sW, long sC, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) { set_error("gemm: bad shape %d %d %d x%d", M, N, K, batch); return MDD_ERR_ARG; }
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, batch), block(256);
    bool aligned = (lda % 4 == 0) && (ldw % 4 == 0) && (sA % 4 == 0) && (sW % 4 == 0) &&
                   ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0);
    const bool seg = N <= 1024 && K > 64;
    if (aligned && seg) hipLaunchKernelGGL((gemm_nt_f32_kernel<true, true>), grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    else if (aligned) hipLaunchKernelGGL((gemm_nt_f32_kernel<true, false>), grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    else if (seg) hipLaunchKernelGGL((gemm_nt_f32_kernel<false, true>), grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    else hipLaunchKernelGGL((gemm_nt_f32_kernel<false, false>), grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// General fp32 GEMM of the training step (forward in train mode and every backward contraction):
//   C[m,n] (+)= sum_k opA[m,k] * opB[n,k]      opA[m,k] = TA ? A[k*lda + m] : A[m*lda + k],  opB likewise with TB
// i.e. (TA,TB) = (0,0) is the "NT" product above, (0,1) C = A.B with B stored [K,N] (dX = dY.W), (1,1) C = A^T.B with both
// stored [K,*] (dW = dY^T.X: the contraction runs over the rows of two activation matrices), (1,0) the remaining case.
// Same tiling and the same exact-fp32 MFMA as gemm_nt_f32_kernel; a transposed operand is read along its contiguous
// axis (4 consecutive m per lane) and transposed on its way into the [row][k] LDS tile.  `accumulate` adds into C
// (gradient accumulation over the two directions / several uses of a tensor).
template <bool TR>
__device__ __forceinline__ void load_tile_any(const float *__restrict__ P, int ld, int rows_total, int K, int row0, int k0, int tid,
                                              float4 (&r)[2]) {
    if (!TR) { load_tile_regs<false>(P, ld, rows_total, K, row0, k0, tid, r); return; }
    // thread -> (k = tid/16, 4 consecutive rows at (tid%16)*4 [+64])
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int k = k0 + (tid >> 4), row = row0 + (tid & 15) * 4 + i * 64;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float *p = P + (size_t)k * ld + row;
            if (row < rows_total) v.x = p[0];
            if (row + 1 < rows_total) v.y = p[1];
            if (row + 2 < rows_total) v.z = p[2];
            if (row + 3 < rows_total) v.w = p[3];
        }
        r[i] = v;
    }
}
template <bool TR>
__device__ __forceinline__ void store_tile_any(float *s, int tid, const float4 (&r)[2]) {
    if (!TR) { store_tile_lds(s, tid, r); return; }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float *d = s + ((tid & 15) * 4 + i * 64) * LDS_LD + (tid >> 4);
        d[0] = r[i].x; d[LDS_LD] = r[i].y; d[2 * LDS_LD] = r[i].z; d[3 * LDS_LD] = r[i].w;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                       const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                       int lda, int ldb, int ldc, long sA, long sB, long sC, int tiles_n, int accumulate, int ksplit) {
    __shared__ float lds[2][2][BM * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // ksplit > 0: split-K -- slice z of the grid contracts k in [z*ksplit, (z+1)*ksplit) into its own partial C (stride sC)
    int kbeg = 0;
    if (ksplit > 0) { kbeg = blockIdx.z * ksplit; K = min(K, kbeg + ksplit); C += (size_t)blockIdx.z * sC; }
    else { A += (size_t)blockIdx.z * sA; Bm += (size_t)blockIdx.z * sB; C += (size_t)blockIdx.z * sC; }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float4 ra[2], rb[2];
    load_tile_any<TA>(A, lda, M, K, m0, kbeg, tid, ra);
    load_tile_any<TB>(Bm, ldb, N, K, n0, kbeg, tid, rb);
    store_tile_any<TA>(lds[0][0], tid, ra);
    store_tile_any<TB>(lds[0][1], tid, rb);
    __syncthreads();
    const int nk = (K - kbeg + BK - 1) / BK;
    const int li = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile_any<TA>(A, lda, M, K, m0, kbeg + (kt + 1) * BK, tid, ra);
            load_tile_any<TB>(Bm, ldb, N, K, n0, kbeg + (kt + 1) * BK, tid, rb);
        }
        const float *as = lds[cur][0] + (wm * 64 + li) * LDS_LD + lh;
        const float *ws = lds[cur][1] + (wn * 64 + li) * LDS_LD + lh;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = as[kk], a1 = as[32 * LDS_LD + kk];
            float b0 = ws[kk], b1 = ws[32 * LDS_LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile_any<TA>(lds[cur ^ 1][0], tid, ra);
            store_tile_any<TB>(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int col = n0 + wn * 64 + j * 32 + li;
            if (col >= N) continue;
            float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) {
                    float *c = C + (size_t)row * ldc + col;
                    *c = acc[i][j][r] + bv + (accumulate ? *c : 0.f);
                }
            }
        }
}

int launch_gemm_f32(bool ta, bool tb, const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda, int ldb,
                    int ldc, int batch, long sA, long sB, long sC, bool accumulate, hipStream_t st, int ksplit) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) { set_error("gemm_f32: bad shape %d %d %d x%d", M, N, K, batch); return MDD_ERR_ARG; }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    if (ksplit > 0) batch = (K + ksplit - 1) / ksplit;     // split-K: `batch` partial products, C + z*sC each
    dim3 grid(tiles_m * tiles_n, 1, batch), block(256);
#define GO(TA_, TB_) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_>), grid, block, 0, st, A, B, bias, C, M, N, K, lda, ldb, ldc, sA, sB, sC, tiles_n, accumulate ? 1 : 0, ksplit)
    if (!ta && !tb) GO(false, false); else if (!ta && tb) GO(false, true); else if (ta && !tb) GO(true, false); else GO(true, true);
#undef GO
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
