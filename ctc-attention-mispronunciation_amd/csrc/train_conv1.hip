// conv1 of the training step (32 -> 32 channels, 3x3, stride 2x2, pad 1; AA/models/model_ctc.py:59-66 with the 0329 config) as three
// direct kernels on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): forward, input gradient, weight gradient.  They read and
// write the channels-last activations themselves ([B][T][W1][32] in, [B][T/2][W2][32] out); the 9x expanded patch matrix of an
// im2col formulation (562 MB at B = 32 x 10 s, written once and read three times per step) never exists.
//   forward   z1[p, co]      = bias[co] + sum_{tap, ci} a0[patch(p, tap), ci] * W[co][tap][ci]          p = (b, t', w')
//   dgrad     da0[q, ci]     = sum_{taps that reach q} sum_co dz1[p(q, tap), co] * W[co][tap][ci]       q = (b, t, w)
//   wgrad     dW[co][tap][ci] = sum_p dz1[p, co] * a0[patch(p, tap), ci]
// W arrives packed as w1r[co][tap * 32 + ci] (launch_pack_w1).
#include "train.h"

namespace mdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C1 = 32;   // channels (both sides)

// ---------------------------------------------------------------------------------------------------------------- forward
// A workgroup = 128 consecutive output positions x 32 output channels; a wave = 32 positions.  Per tap the 128 input rows (128 B each)
// are gathered through registers into LDS ([pos][ci], stride 33: the MFMA operand of lane (pos, k) is a column read), the weights of
// all taps sit in LDS for the whole workgroup ([co][288], stride 289).
__global__ __launch_bounds__(256) void conv1_fwd_direct_kernel(const float *__restrict__ a0, const float *__restrict__ w1r, const float *__restrict__ bias,
                                                               float *__restrict__ z1, int B, int T, int W1, int W2) {
    __shared__ float Ws[C1 * 289];
    __shared__ float As[128 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int Tp = T / 2;
    const size_t R1 = (size_t)B * Tp * W2, p0 = (size_t)blockIdx.x * 128;
    for (int i = tid; i < C1 * 288; i += 256) { const int co = i / 288, k = i - co * 288; Ws[co * 289 + k] = w1r[i]; }
    // this thread's four gather slots: position p0 + tid/8 + 32 i, channels (tid % 8) * 4 ..
    const int c4 = (tid & 7) * 4;
    int gb[4], gt[4], gw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const size_t p = p0 + (tid >> 3) + 32 * i;
        if (p < R1) { const int wo = (int)(p % W2); const size_t r = p / W2; gw[i] = 2 * wo - 1; gt[i] = 2 * (int)(r % Tp) - 1; gb[i] = (int)(r / Tp); }
        else { gb[i] = -1; gt[i] = gw[i] = 0; }
    }
    auto gather = [&](int tap, float4 (&v)[4]) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ti = gt[i] + kh, wi = gw[i] + kw;
            v[i] = (gb[i] >= 0 && ti >= 0 && ti < T && wi >= 0 && wi < W1) ? *reinterpret_cast<const float4 *>(a0 + (((size_t)gb[i] * T + ti) * W1 + wi) * C1 + c4)
                                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](const float4 (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float *d = As + ((tid >> 3) + 32 * i) * 33 + c4;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float4 v[4];
    gather(0, v);
    park(v);
    __syncthreads();
    for (int tap = 0; tap < 9; tap++) {
        if (tap + 1 < 9) gather(tap + 1, v);                            // the next tap's rows travel while this tap's products run
        const float *as = As + (wave * 32 + li) * 33 + lh, *ws = Ws + li * 289 + tap * 32 + lh;
#pragma unroll
        for (int j = 0; j < 16; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * j], ws[2 * j], acc, 0, 0, 0);
        __syncthreads();
        if (tap + 1 < 9) { park(v); __syncthreads(); }
    }
    const float bv = bias[li];
#pragma unroll
    for (int r = 0; r < 16; r++) {   // D: row (position) = (r & 3) + 8 (r >> 2) + 4 lh, column (co) = li
        const size_t p = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < R1) z1[p * C1 + li] = acc[r] + bv;
    }
}
int launch_conv1_fwd_direct(const float *a0, const float *w1r, const float *bias, float *z1, int B, int T, int W1, int W2, int ch, hipStream_t st) {
    if (ch != C1) { set_error("conv1 direct: channels=%d not built", ch); return MDD_ERR_ARG; }
    const size_t R1 = (size_t)B * (T / 2) * W2;
    hipLaunchKernelGGL(conv1_fwd_direct_kernel, dim3((unsigned)((R1 + 127) / 128)), dim3(256), 0, st, a0, w1r, bias, z1, B, T, W1, W2);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ---------------------------------------------------------------------------------------------------------------- input gradient
// With stride 2 an input row t is reached through kh = 1 only (t even: t' = t/2) or through kh = 0 and 2 (t odd: t' = (t+1)/2 and
// (t-1)/2), and the same along w: four classes of input positions with 1, 2, 2, 4 taps.  A workgroup takes the input rows (b, 2m) and
// (b, 2m+1); wave c takes one class: the positions w = (c & 1) + 2 j, j < ceil((W1 - (c & 1)) / 2), of row 2m + (c >> 1).  For a
// tap the operand rows dz1[b, t', w'_j, :] are CONSECUTIVE output positions (w'_j = j + const), so a wave stages them with contiguous
// 16-byte loads into its own LDS tile and nothing crosses waves after the weights are in.
__global__ __launch_bounds__(256) void conv1_dgrad_direct_kernel(const float *__restrict__ dz1, const float *__restrict__ w1r, float *__restrict__ da0,
                                                                 int B, int T, int W1, int W2) {
    extern __shared__ __attribute__((aligned(16))) float dg_smem[];
    float *Ws = dg_smem;                      // [tap][co][ci]
    float *Asall = dg_smem + 9 * C1 * C1;     // per wave: [j][co], stride 33
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int Tp = T / 2;
    const int b = blockIdx.x / Tp, t = 2 * (blockIdx.x - b * Tp) + (wave >> 1), par = wave & 1;
    for (int i = tid; i < 9 * C1 * C1; i += 256) { const int tap = i / (C1 * C1), r = i - tap * C1 * C1, co = r / C1, ci = r - co * C1; Ws[i] = w1r[co * 288 + tap * C1 + ci]; }
    __syncthreads();
    const int nj = (W1 - par + 1) / 2;                                  // positions of this class in the row (61 at W1 = 122)
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
    float *as = Asall + wave * 64 * 33;
    // this wave's taps (1, 2 or 4), pipelined: the next tap's operand rows are requested before the current tap's products
    int ntap = 0, ttap[4], tsrc_tp[4], twoo[4];
    for (int kh = (t & 1) ? 0 : 1; kh < 3; kh += 2) {
        const int t2 = t + 1 - kh, tp = t2 >> 1;                         // t2 = 2 t'
        if (t2 < 0 || tp >= Tp) continue;
        for (int kw = par ? 0 : 1; kw < 3; kw += 2) { ttap[ntap] = kh * 3 + kw; tsrc_tp[ntap] = tp; twoo[ntap] = (par + 1 - kw) >> 1; ntap++; }   // w'_j = j + woo
    }
    auto gather = [&](int n, float4 (&v)[8]) {   // rows j = 0..63: dz1[b, tp, j + woo, :]  (zero outside [0, W2) and past nj)
        const float *src = dz1 + (((size_t)b * Tp + tsrc_tp[n]) * W2 + twoo[n]) * C1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int q = lane + 64 * i, j = q >> 3, c = (q & 7) * 4;
            v[i] = (j < nj && j + twoo[n] < W2) ? *reinterpret_cast<const float4 *>(src + (size_t)j * C1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](const float4 (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int q = lane + 64 * i, j = q >> 3, c = (q & 7) * 4;
            float *d_ = as + j * 33 + c;
            d_[0] = v[i].x; d_[1] = v[i].y; d_[2] = v[i].z; d_[3] = v[i].w;
        }
    };
    float4 v[8];
    if (ntap > 0) { gather(0, v); park(v); }
    for (int n = 0; n < ntap; n++) {
        __builtin_amdgcn_wave_barrier();                                 // (this wave's own tile: LDS operations of one wave complete in order)
        if (n + 1 < ntap) gather(n + 1, v);
        const float *ws = Ws + ttap[n] * C1 * C1 + lh * C1 + li;         // B[k = co][col = ci]
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const float bv = ws[2 * jj * C1];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[li * 33 + 2 * jj + lh], bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[(32 + li) * 33 + 2 * jj + lh], bv, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        if (n + 1 < ntap) park(v);
    }
    if (t >= T) return;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {   // D: row j = 32 m + (r & 3) + 8 (r >> 2) + 4 lh, column ci = li
            const int j = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (j < nj) da0[(((size_t)b * T + t) * W1 + par + 2 * j) * C1 + li] = acc[m][r];
        }
}
int launch_conv1_dgrad_direct(const float *dz1, const float *w1r, float *da0, int B, int T, int W1, int W2, int ch, hipStream_t st) {
    if (ch != C1 || W1 > 128) { set_error("conv1 direct: channels=%d / width=%d not built", ch, W1); return MDD_ERR_ARG; }
    hipLaunchKernelGGL(conv1_dgrad_direct_kernel, dim3((unsigned)(B * (T / 2))), dim3(256), (9 * C1 * C1 + 4 * 64 * 33) * sizeof(float), st, dz1, w1r, da0, B, T, W1, W2);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
// dW[co][tap][ci]: the contraction runs over the 488 000 output positions.  A workgroup walks tiles of 128 positions (grid-stride);
// per tile the dz1 rows and, tap by tap, the a0 patch rows are staged in LDS exactly as the forward kernel stages them ([pos][c],
// stride 33), and wave w contracts positions 32 w .. 32 w + 31 of the tile: A[co][k = pos] and B[k = pos][ci] are column reads of
// the two tiles.  Nine 32x32 accumulators per wave for the whole walk; at the end the four waves are summed through LDS in a fixed
// order and the workgroup's partial goes to `part` (reduced afterwards in fp64: deterministic).
__global__ __launch_bounds__(256) void conv1_wgrad_direct_kernel(const float *__restrict__ dz1, const float *__restrict__ a0, float *__restrict__ part,
                                                                 int B, int T, int W1, int W2, int ntiles) {
    __shared__ float red[9 * 1024];                                      // the walk's two operand tiles, then the waves' sum
    float *Dz = red, *As = red + 128 * 33;
    static_assert(2 * 128 * 33 <= 9 * 1024, "operand tiles fit the reduction buffer");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int Tp = T / 2, c4 = (tid & 7) * 4;
    const size_t R1 = (size_t)B * Tp * W2;
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[k][r] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t p0 = (size_t)tile * 128;
        int gb[4], gt[4], gw[4];
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const size_t p = p0 + (tid >> 3) + 32 * i;
            if (p < R1) { const int wo = (int)(p % W2); const size_t r = p / W2; gw[i] = 2 * wo - 1; gt[i] = 2 * (int)(r % Tp) - 1; gb[i] = (int)(r / Tp); }
            else { gb[i] = -1; gt[i] = gw[i] = 0; }
            v[i] = p < R1 ? *reinterpret_cast<const float4 *>(dz1 + p * C1 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        auto gather = [&](int tap, float4 (&u)[4]) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ti = gt[i] + kh, wi = gw[i] + kw;
                u[i] = (gb[i] >= 0 && ti >= 0 && ti < T && wi >= 0 && wi < W1) ? *reinterpret_cast<const float4 *>(a0 + (((size_t)gb[i] * T + ti) * W1 + wi) * C1 + c4)
                                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto park = [&](float *dst, const float4 (&u)[4]) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float *d = dst + ((tid >> 3) + 32 * i) * 33 + c4;
                d[0] = u[i].x; d[1] = u[i].y; d[2] = u[i].z; d[3] = u[i].w;
            }
        };
        float4 u[4];
        gather(0, u);
        __syncthreads();                                                 // the previous tile's products are done with Dz / As
        park(Dz, v);
        park(As, u);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap + 1 < 9) gather(tap + 1, u);
            const float *dz = Dz + (wave * 32 + lh) * 33 + li, *as = As + (wave * 32 + lh) * 33 + li;
#pragma unroll
            for (int j = 0; j < 16; j++) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[2 * j * 33], as[2 * j * 33], acc[tap], 0, 0, 0);
            if (tap + 1 < 9) { __syncthreads(); park(As, u); __syncthreads(); }
        }
    }
    __syncthreads();
    // D: row (co) = (r & 3) + 8 (r >> 2) + 4 lh, column (ci) = li  ->  red[tap][co][ci]
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int tap = 0; tap < 9; tap++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float *d_ = red + tap * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * C1 + li;
                    *d_ = (w == 0 ? 0.f : *d_) + acc[tap][r];
                }
        }
        __syncthreads();
    }
    float *dst = part + (size_t)blockIdx.x * (C1 * 288);
    for (int i = tid; i < 9 * 1024; i += 256) { const int tap = i >> 10, co = (i >> 5) & 31, ci = i & 31; dst[co * 288 + tap * C1 + ci] = red[i]; }
}
int conv1_wgrad_parts(int B, int T, int W2) {
    const size_t R1 = (size_t)B * (T / 2) * W2, tiles = (R1 + 127) / 128;
    return (int)std::min<size_t>(512, tiles);
}
int launch_conv1_wgrad_direct(const float *dz1, const float *a0, float *part, int B, int T, int W1, int W2, int ch, hipStream_t st) {
    if (ch != C1) { set_error("conv1 direct: channels=%d not built", ch); return MDD_ERR_ARG; }
    const size_t R1 = (size_t)B * (T / 2) * W2;
    const int parts = conv1_wgrad_parts(B, T, W2);
    hipLaunchKernelGGL(conv1_wgrad_direct_kernel, dim3(parts), dim3(256), 0, st, dz1, a0, part, B, T, W1, W2, (int)((R1 + 127) / 128));
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int init_conv1_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)conv1_dgrad_direct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (9 * C1 * C1 + 4 * 64 * 33) * (int)sizeof(float)));
    return MDD_OK;
}

}  // namespace mdd
