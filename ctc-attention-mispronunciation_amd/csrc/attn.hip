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
                             int L, float *__restrict__ out, unsigned short *__restrict__ out_hi,
                             unsigned short *__restrict__ out_lo, int *err_flag) {
    // out row m = l*B + b  (time-major, like every other sequence buffer)
    const int m = blockIdx.x, l = m / B, b = m - l * B;
    long id = ids[(size_t)b * L + l];
    if (id < 0 || id >= rows) {  // reference: IndexError from nn.Embedding
        if (threadIdx.x == 0) atomicExch(err_flag, 1);
        id = 0;
    }
    const float *src = table + (size_t)id * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float v = src[e];
        if (out) out[(size_t)m * E + e] = v;
        if (out_hi) {
            __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
            out_hi[(size_t)m * E + e] = *reinterpret_cast<unsigned short *>(&h);
            out_lo[(size_t)m * E + e] = *reinterpret_cast<unsigned short *>(&l);
        }
    }
}

int launch_embed(const float *table, int rows, int E, const int64_t *ids, int B, int L, float *out, SplitPtr sp,
                 int *err_flag, hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3(B * L), dim3(128), 0, st, table, rows, E, ids, B, L, out, sp.hi, sp.lo, err_flag);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

constexpr int TT = 16;  // posterior frames per workgroup

// dynamic LDS: attw[TT][L] | y[TT][2*H2] | logits[TT][C]
__global__ __launch_bounds__(256) void attn_tail_kernel(const float *__restrict__ S, int Lp, const float *__restrict__ X,
                                                        const float *__restrict__ V, const float *__restrict__ fscale,
                                                        const float *__restrict__ fshift, const float *__restrict__ wfc,
                                                        float *__restrict__ logp, int Tp, int B, int Lmax, int H2, int C,
                                                        const int *__restrict__ llen) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = llen ? min(max(llen[blockIdx.y], 1), Lmax) : Lmax;   // this utterance's own batch's canonical length (fused batches)
    float *attw = smem;                 // [TT][L]
    float *y = attw + TT * Lmax;        // [TT][2*H2]
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

// ---- MFMA variant (4H % 64 == 0): the context product and the classifier run on v_mfma_f32_16x16x4_f32.
//   ctx[16 x 2H]  = attw[16 x L] . V_b[L x 2H]     : wave w owns column tiles w, w+4, ..
//   logits[16x48] = y[16 x 4H] . Wfc^T             : K split over the 4 waves, partials summed through LDS
// Wfc is repacked at weight-load time into the order the lanes consume it (zero rows for n >= C):
//   wfcp[w][nt][j][lane][m] = Wfc[nt*16 + (lane&15)][w*(4H/4) + 16 j + 4 (lane>>4) + m]
// The x half of y = BN(X) never touches LDS: waves 0 and 1 (whose K quarters lie in it) read their A operands
// straight from X, 16 bytes per lane.  Only the context half is staged, so a workgroup holds ~65 KB and two fit a
// CU (the kernel is a chain of global-load latencies: a second resident workgroup is what hides them).  In the
// context product the V values of the next column tile are requested before the current tile's MFMAs.
// dynamic LDS: attw[16][LA] | yc[16][H2 + 4] | part[4][16][48]
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void attn_tail_mfma_kernel(const float *__restrict__ S, int Lp, const float *__restrict__ X,
                                                             const float *__restrict__ V, const float *__restrict__ fscale,
                                                             const float *__restrict__ fshift, const float *__restrict__ wfcp,
                                                             float *__restrict__ logp, int Tp, int B, int Lmax, int H2, int C,
                                                             const int *__restrict__ llen) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // fused batches of different canonical lengths: this utterance attends over the l < llen[b] of its own batch (the same
    // arithmetic, in the same order, as a launch with L = llen[b]); the LDS tile is sized for Lmax
    const int L = llen ? min(max(llen[blockIdx.y], 1), Lmax) : Lmax;
    const int D2 = 2 * H2, LDY = H2 + 4, LA = (L + 3) & ~3;
    float *attw = smem;                  // [16][LA]
    float *yc = attw + 16 * ((Lmax + 3) & ~3);   // [16][LDY]   context half of y
    float *part = yc + 16 * LDY;         // [4][16][48]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * 16;
    const int nt_rows = min(16, Tp - t0);

    // classifier operand addresses
    const int J = D2 / 64;               // float4 groups per wave quarter (J % 4 == 0: D2 % 256 == 0 for H in {256, 384})
    const float4 *wp = reinterpret_cast<const float4 *>(wfcp) + (size_t)wave * 3 * J * 64 + lane;
    const bool xhalf = wave < 2;         // this wave's K quarter [wave*D2/4, +D2/4) lies in the x half (H2 = 2 quarters)
    const int xrow = min(t0 + li, Tp - 1);
    const float *xsrc = X + ((size_t)xrow * B + b) * H2 + wave * (D2 / 4) + 4 * kq;   // + 16 j
    const float *scp = fscale + wave * (D2 / 4) + 4 * kq, *shp = fshift + wave * (D2 / 4) + 4 * kq;

    // 1. softmax over L (one wave per row); padded rows / columns are zero.  With L <= 64 (one element per lane) a wave's
    //    four rows are requested together: four dependent global round trips become one.
    if (L <= 64) {
        float sv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = wave + 4 * i;
            sv[i] = (r < nt_rows && lane < L) ? S[((size_t)b * Tp + t0 + r) * Lp + lane] : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = wave + 4 * i;
            float *arow = attw + r * LA;
            if (r >= nt_rows) { if (lane < LA) arow[lane] = 0.f; continue; }
            float mx = sv[i];
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float e = lane < L ? expf(sv[i] - mx) : 0.f;
            float sum = e;
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane < LA) arow[lane] = lane < L ? e / sum : 0.f;
        }
    } else
    for (int r = wave; r < 16; r += 4) {
        float *arow = attw + r * LA;
        if (r >= nt_rows) { for (int l = lane; l < LA; l += 64) arow[l] = 0.f; continue; }
        const float *srow = S + ((size_t)b * Tp + t0 + r) * Lp;
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, srow[l]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int l = lane; l < LA; l += 64) { float e = l < L ? expf(srow[l] - mx) : 0.f; arow[l] = e; sum += e; }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int l = lane; l < L; l += 64) arow[l] = arow[l] / sum;
    }
    __syncthreads();

    // 2. ctx tiles: A = attw[row li][k], B = V[k = l][col d].  All V values of a tile are requested before the first MFMA
    //    (a lone dependent global load per k-step cost ~500 cycles each), and one tile ahead of the MFMAs.
    {
        auto loadV = [&](float (&vv)[16], int nt, int k0) {
            const int d = nt * 16 + li;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int l = k0 + 4 * i + kq;
                vv[i] = (k0 + 4 * i < LA && nt * 16 < H2) ? V[((size_t)(l < L ? l : L - 1) * B + b) * H2 + (d < H2 ? d : 0)] : 0.f;   // attw is 0 for l >= L
            }
        };
        if (LA <= 64) {   // the usual case (L <= 64): one k block per tile, software-pipelined over the tiles
            float va[16], vb[16];
            loadV(va, wave, 0);
            for (int nt = wave; nt * 16 < H2; nt += 8) {
                loadV(vb, nt + 4, 0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (4 * i < LA) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(attw[li * LA + 4 * i + kq], va[i], acc, 0, 0, 0);
                {
                    const int d = nt * 16 + li;
                    const float sc = fscale[H2 + d], hc = fshift[H2 + d];
#pragma unroll
                    for (int r = 0; r < 4; r++) yc[(kq * 4 + r) * LDY + d] = acc[r] * sc + hc;   // D: row = 4*(lane>>4)+r, col = lane&15
                }
                if ((nt + 4) * 16 < H2) {
                    loadV(va, nt + 8, 0);
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (4 * i < LA) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(attw[li * LA + 4 * i + kq], vb[i], acc2, 0, 0, 0);
                    const int d = (nt + 4) * 16 + li;
                    const float sc = fscale[H2 + d], hc = fshift[H2 + d];
#pragma unroll
                    for (int r = 0; r < 4; r++) yc[(kq * 4 + r) * LDY + d] = acc2[r] * sc + hc;
                }
            }
        } else {
            for (int nt = wave; nt * 16 < H2; nt += 4) {
                const int d = nt * 16 + li;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < LA; k0 += 64) {
                    float vv[16];
                    loadV(vv, nt, k0);
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (k0 + 4 * i < LA) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(attw[li * LA + k0 + 4 * i + kq], vv[i], acc, 0, 0, 0);
                }
                const float sc = fscale[H2 + d], hc = fshift[H2 + d];
#pragma unroll
                for (int r = 0; r < 4; r++) yc[(kq * 4 + r) * LDY + d] = acc[r] * sc + hc;
            }
        }
    }
    __syncthreads();

    // 3. classifier: this wave's quarter of K, three 16-column tiles
    {
        const float *yr = yc + li * LDY + (xhalf ? 0 : wave - 2) * (D2 / 4) + 4 * kq;   // context half (waves 2, 3)
        // two-level accumulation (see gemm.hip, SEG): 16 roundings per segment, then one per segment, instead of a chain of D2 / 16
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, t0 = a0, t1 = a0, t2 = a0;
        for (int j0 = 0; j0 < J; j0 += 4) {
            float4 y4[4], w0[4], w1[4], w2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                w0[u] = wp[(0 * J + j) * 64]; w1[u] = wp[(1 * J + j) * 64]; w2[u] = wp[(2 * J + j) * 64];
                if (xhalf) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xsrc + 16 * j);
                    const float4 sc = *reinterpret_cast<const float4 *>(scp + 16 * j), sh = *reinterpret_cast<const float4 *>(shp + 16 * j);
                    y4[u] = make_float4(xv.x * sc.x + sh.x, xv.y * sc.y + sh.y, xv.z * sc.z + sh.z, xv.w * sc.w + sh.w);
                } else {
                    y4[u] = *reinterpret_cast<const float4 *>(yr + 16 * j);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].x, w0[u].x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].x, w1[u].x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].x, w2[u].x, a2, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].y, w0[u].y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].y, w1[u].y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].y, w2[u].y, a2, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].z, w0[u].z, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].z, w1[u].z, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].z, w2[u].z, a2, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].w, w0[u].w, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].w, w1[u].w, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y4[u].w, w2[u].w, a2, 0, 0, 0);
            }
            t0 += a0; t1 += a1; t2 += a2;
            a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; a2 = a0;
        }
        a0 = t0; a1 = t1; a2 = t2;
        float *pw = part + wave * 16 * 48;
#pragma unroll
        for (int r = 0; r < 4; r++) {   // D: row (time) = 4*(lane>>4)+r, col (class) = lane&15
            pw[(kq * 4 + r) * 48 + li] = a0[r];
            pw[(kq * 4 + r) * 48 + 16 + li] = a1[r];
            pw[(kq * 4 + r) * 48 + 32 + li] = a2[r];
        }
    }
    __syncthreads();

    // 4. sum the four K-quarters, log-softmax over C (one wave per row)
    for (int r = wave; r < nt_rows; r += 4) {
        float v = -INFINITY;
        if (lane < C) v = (part[r * 48 + lane] + part[(16 + r) * 48 + lane]) + (part[(32 + r) * 48 + lane] + part[(48 + r) * 48 + lane]);
        float mx = v;
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = lane < C ? expf(v - mx) : 0.f;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float lse = logf(sum);
        if (lane < C) logp[((size_t)(t0 + r) * B + b) * C + lane] = (v - mx) - lse;
    }
}

int launch_attn_tail(const float *S, int Lp, const float *X, const float *V, const float *fscale, const float *fshift,
                     const float *wfc, const float *wfcp, float *logp, int Tp, int B, int L, int H2, int C, hipStream_t st, const int *llen) {
    if (wfcp && (2 * H2) % 256 == 0 && C <= 48) {
        const int D2 = 2 * H2, LA = (L + 3) & ~3;
        size_t smem = sizeof(float) * ((size_t)16 * LA + (size_t)16 * (H2 + 4) + 4 * 16 * 48);
        if (smem > 160 * 1024) { set_error("attn_tail: L=%d too long for the LDS tile (%zu B)", L, smem); return MDD_ERR_ARG; }
        hipLaunchKernelGGL(attn_tail_mfma_kernel, dim3((Tp + 15) / 16, B), dim3(256), smem, st, S, Lp, X, V, fscale, fshift, wfcp,
                           logp, Tp, B, L, H2, C, llen);
        MDD_LAUNCH_CHECK();
        return MDD_OK;
    }
    if (H2 % 2 != 0) { set_error("attn_tail: 2H must be even"); return MDD_ERR_ARG; }
    size_t smem = sizeof(float) * ((size_t)TT * L + (size_t)TT * 2 * H2 + (size_t)TT * C);
    if (smem > 160 * 1024) { set_error("attn_tail: L=%d too long for the LDS tile (%zu B)", L, smem); return MDD_ERR_ARG; }
    // y starts 16-byte aligned because TT*L*4 % 16 == 0 for TT = 16.
    dim3 grid((Tp + TT - 1) / TT, B), block(256);
    hipLaunchKernelGGL(attn_tail_kernel, grid, block, smem, st, S, Lp, X, V, fscale, fshift, wfc, logp, Tp, B, L, H2, C, llen);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int init_kernel_attributes() {  // called once from mdd_create (never inside a stream capture)
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)attn_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)attn_tail_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return MDD_OK;
}

}  // namespace mdd
