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

template <bool ALIGNED>
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

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
                if (row < M) C[(size_t)row * ldc + col] = acc[i][j][r] + bv;
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
    if (aligned)
        hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    else
        hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, grid, block, 0, st, A, W, bias, C, M, N, K, lda, ldw, ldc, sA, sW, sC, tiles_n);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
