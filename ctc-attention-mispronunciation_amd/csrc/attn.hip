// Text-side gather and the fused attention / classifier tail.
//
// Reference: AA/models/model_ctc.py:193 (embeds), :204-218:
//   attn = softmax_L(X . key^T)   (no 1/sqrt(d) scale, no padding mask)
//   c    = attn . x1                                   [T',768]
//   out  = log_softmax( Linear(BN1d(cat(X, c))) )       fc = BN(1536) + Linear(1536->C, no bias)
// The scores S = X.key^T come from the batched MFMA GEMM (gemm.hip); this kernel fuses everything
// after them so neither the attention weights, the context, the concatenation nor the logits ever
// touch HBM.  One workgroup = one utterance x 16 posterior frames.
#include "mdd_internal.h"

namespace mdd {

__global__ void embed_kernel(const float *__restrict__ table, int rows, int E, const int64_t *__restrict__ ids, int B,
                             int L, float *__restrict__ out, int *err_flag) {
    // out row m = l*B + b  (time-major, like every other sequence buffer)
    const int m = blockIdx.x, l = m / B, b = m - l * B;
    long id = ids[(size_t)b * L + l];
    if (id < 0 || id >= rows) {  // reference: IndexError from nn.Embedding
        if (threadIdx.x == 0) atomicExch(err_flag, 1);
        id = 0;
    }
    const float *src = table + (size_t)id * E;
    float *dst = out + (size_t)m * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) dst[e] = src[e];
}

int launch_embed(const float *table, int rows, int E, const int64_t *ids, int B, int L, float *out, int *err_flag,
                 hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3(B * L), dim3(128), 0, st, table, rows, E, ids, B, L, out, err_flag);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

constexpr int TT = 16;  // posterior frames per workgroup

// dynamic LDS: attw[TT][L] | y[TT][2*H2] | logits[TT][C]
__global__ __launch_bounds__(256) void attn_tail_kernel(const float *__restrict__ S, int Lp, const float *__restrict__ X,
                                                        const float *__restrict__ V, const float *__restrict__ fscale,
                                                        const float *__restrict__ fshift, const float *__restrict__ wfc,
                                                        float *__restrict__ logp, int Tp, int B, int L, int H2, int C) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *attw = smem;                 // [TT][L]
    float *y = attw + TT * L;           // [TT][2*H2]
    float *lg = y + TT * 2 * H2;        // [TT][C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int nt = min(TT, Tp - t0);
    const int D2 = 2 * H2;

    // 1. softmax over L: one wave per row, 4 rows per wave
    for (int r = wave; r < TT; r += 4) {
        if (r >= nt) { for (int l = lane; l < L; l += 64) attw[r * L + l] = 0.f; continue; }
        const float *srow = S + ((size_t)b * Tp + t0 + r) * Lp;
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, srow[l]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) { float e = expf(srow[l] - mx); attw[r * L + l] = e; sum += e; }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int l = lane; l < L; l += 64) attw[r * L + l] = attw[r * L + l] / sum;
    }
    __syncthreads();

    // 2. y = BN(cat(X, ctx)); thread owns columns d = tid, tid+256, ...
    for (int d = tid; d < H2; d += 256) {
        float acc[TT];
#pragma unroll
        for (int r = 0; r < TT; r++) acc[r] = 0.f;
        for (int l = 0; l < L; l++) {
            const float v = V[((size_t)l * B + b) * H2 + d];
#pragma unroll
            for (int r = 0; r < TT; r++) acc[r] = fmaf(attw[r * L + l], v, acc[r]);
        }
        const float sx = fscale[d], hx = fshift[d], sc = fscale[H2 + d], hc = fshift[H2 + d];
#pragma unroll
        for (int r = 0; r < TT; r++) {
            const float xv = r < nt ? X[((size_t)(t0 + r) * B + b) * H2 + d] : 0.f;
            y[r * D2 + d] = xv * sx + hx;
            y[r * D2 + H2 + d] = acc[r] * sc + hc;
        }
    }
    __syncthreads();

    // 3. logits[r][c] = y[r] . wfc[c]: a wave per (r,c) dot product, lanes over k
    for (int o = wave; o < TT * C; o += 4) {
        const int r = o / C, c = o - r * C;
        const float *wr = wfc + (size_t)c * D2;
        const float *yr = y + r * D2;
        float acc = 0.f;
        for (int k = lane * 4; k < D2; k += 256) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wr + k);
            const float4 y4 = *reinterpret_cast<const float4 *>(yr + k);
            acc = fmaf(w4.x, y4.x, acc); acc = fmaf(w4.y, y4.y, acc);
            acc = fmaf(w4.z, y4.z, acc); acc = fmaf(w4.w, y4.w, acc);
        }
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if (lane == 0) lg[r * C + c] = acc;
    }
    __syncthreads();

    // 4. log-softmax over C, one wave per row
    for (int r = wave; r < nt; r += 4) {
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lg[r * C + c]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(lg[r * C + c] - mx);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float lse = logf(sum);
        float *orow = logp + ((size_t)(t0 + r) * B + b) * C;
        for (int c = lane; c < C; c += 64) orow[c] = (lg[r * C + c] - mx) - lse;
    }
}

int launch_attn_tail(const float *S, int Lp, const float *X, const float *V, const float *fscale, const float *fshift,
                     const float *wfc, float *logp, int Tp, int B, int L, int H2, int C, hipStream_t st) {
    if (H2 % 2 != 0) { set_error("attn_tail: 2H must be even"); return MDD_ERR_ARG; }
    size_t smem = sizeof(float) * ((size_t)TT * L + (size_t)TT * 2 * H2 + (size_t)TT * C);
    if (smem > 160 * 1024) { set_error("attn_tail: L=%d too long for the LDS tile (%zu B)", L, smem); return MDD_ERR_ARG; }
    // y starts 16-byte aligned because TT*L*4 % 16 == 0 for TT = 16.
    dim3 grid((Tp + TT - 1) / TT, B), block(256);
    hipLaunchKernelGGL(attn_tail_kernel, grid, block, smem, st, S, Lp, X, V, fscale, fshift, wfc, logp, Tp, B, L, H2, C);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int init_kernel_attributes() {  // called once from mdd_create (never inside a stream capture)
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)attn_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return MDD_OK;
}

}  // namespace mdd
