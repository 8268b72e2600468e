// Input stacking and the 2-layer CNN front-end.
//
// Reference: make_context/skip_feat (AA/utils/tools.py:207-227, AA/utils/data_loader.py:138-142) and
// LayerCNN.forward x2 (AA/models/model_ctc.py:73-81; geometry AA/conf/ctc_config.0329.yaml:59-64):
//   conv0: Conv2d(1->ch, k3x3, stride (1,2), pad 1) + bias -> BN2d(eval) -> ReLU
//   conv1: Conv2d(ch->ch, k3x3, stride (2,2), pad 1) + bias -> BN2d(eval) -> ReLU
// followed by the [B,ch,T',W2] -> [T',B,ch*W2] relayout (model_ctc.py:176-181), which conv1's store
// performs directly.  Eval BatchNorm and the conv bias are folded into per-channel scale/shift on
// the host side of the library (y = conv_nobias * scale + shift).
#include "mdd_internal.h"

namespace mdd {

// ---------------------------------------------------------------- A1
__global__ void stack_skip_kernel(const float *__restrict__ raw, int T_raw, int D, int right, int skip, int kept,
                                  int T_out, float *__restrict__ out) {
    const int W = (right + 1) * D;
    const int b = blockIdx.y, i = blockIdx.x;  // one output frame per block
    const float *src = raw + (size_t)b * T_raw * D;
    float *dst = out + ((size_t)b * T_out + i) * W;
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        float v = 0.f;
        if (i < kept) {
            int r = j / D, c = j - r * D;
            int s = i * skip + r;
            if (s > T_raw - 1) s = T_raw - 1;  // last frame replicated at the edge
            v = src[(size_t)s * D + c];
        }
        dst[j] = v;
    }
}

int launch_stack_skip(const float *raw, int B, int T_raw, int D, int right, int skip, int n_down, float *out,
                      hipStream_t st) {
    if (B <= 0 || T_raw <= 0 || D <= 0 || right < 0) { set_error("stack_skip: bad shape"); return MDD_ERR_ARG; }
    if (skip < 1) skip = 1;
    int kept = (T_raw + skip - 1) / skip;
    int T_out = mdd_stack_len(T_raw, skip, n_down);
    hipLaunchKernelGGL(stack_skip_kernel, dim3(T_out, B), dim3(256), 0, st, raw, T_raw, D, right, skip, kept, T_out, out);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ---------------------------------------------------------------- conv0 (Cin = 1)
// one thread per output position (b,t,w), all channels; stores are coalesced over w per channel.
template <int CH>
__global__ __launch_bounds__(256) void conv0_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    float *__restrict__ y, int B, int T, int F, int W1) {
    const int wo = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y, b = blockIdx.z;
    if (wo >= W1) return;
    float in[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
        int ti = t + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            int fi = wo * 2 + kw - 1;
            in[kh][kw] = (ti >= 0 && ti < T && fi >= 0 && fi < F) ? x[((size_t)b * T + ti) * F + fi] : 0.f;
        }
    }
#pragma unroll 4
    for (int c = 0; c < CH; c++) {
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++) acc = fmaf(in[kh][kw], w[c * 9 + kh * 3 + kw], acc);
        float v = acc * scale[c] + shift[c];
        y[(((size_t)b * CH + c) * T + t) * W1 + wo] = v > 0.f ? v : 0.f;
    }
}

int launch_conv0(const float *x, const float *w, const float *scale, const float *shift, float *y0, int B, int T, int F,
                 int ch, hipStream_t st) {
    int W1 = (F + 2 - 3) / 2 + 1;
    dim3 grid((W1 + 127) / 128, T, B), block(128);
    if (ch == 32) hipLaunchKernelGGL(conv0_kernel<32>, grid, block, 0, st, x, w, scale, shift, y0, B, T, F, W1);
    else if (ch == 4) hipLaunchKernelGGL(conv0_kernel<4>, grid, block, 0, st, x, w, scale, shift, y0, B, T, F, W1);
    else { set_error("conv0: channels=%d not built (32 or 4)", ch); return MDD_ERR_ARG; }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ---------------------------------------------------------------- conv1 (CH -> CH, stride 2x2)
// Weights are pre-transposed on load to w_t[ci][kh][kw][co] and staged in LDS; every lane reads the
// same 16-byte slot (broadcast).  A thread produces all CH output channels for two adjacent output
// columns so each broadcast weight read feeds two FMAs.
template <int CH>
__global__ __launch_bounds__(256) void conv1_kernel(const float *__restrict__ y0, const float *__restrict__ w_t,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    float *__restrict__ seq, unsigned short *__restrict__ seq_hi,
                                                    unsigned short *__restrict__ seq_lo, int B, int T, int W1, int W2) {
    __shared__ __attribute__((aligned(16))) float wl[CH * 9 * CH];
    for (int i = threadIdx.x; i < CH * 9 * CH; i += blockDim.x) wl[i] = w_t[i];
    __syncthreads();
    const int Tp = T / 2, pairs = (W2 + 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (output row, pair of output columns)
    const int tp = idx / pairs, b = blockIdx.z;
    const int wo = (idx - tp * pairs) * 2;
    if (tp >= Tp) return;
    float acc0[CH], acc1[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { acc0[c] = 0.f; acc1[c] = 0.f; }
    for (int ci = 0; ci < CH; ci++) {
        const float *plane = y0 + ((size_t)b * CH + ci) * T * W1;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int ti = tp * 2 + kh - 1;
            float in[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                int fi = wo * 2 + j - 1;
                in[j] = (ti >= 0 && ti < T && fi >= 0 && fi < W1) ? plane[(size_t)ti * W1 + fi] : 0.f;
            }
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const float *wr = wl + ((ci * 3 + kh) * 3 + kw) * CH;
#pragma unroll
                for (int c = 0; c < CH; c += 4) {
                    float4 w4 = *reinterpret_cast<const float4 *>(wr + c);
                    acc0[c] = fmaf(in[kw], w4.x, acc0[c]);         acc1[c] = fmaf(in[kw + 2], w4.x, acc1[c]);
                    acc0[c + 1] = fmaf(in[kw], w4.y, acc0[c + 1]); acc1[c + 1] = fmaf(in[kw + 2], w4.y, acc1[c + 1]);
                    acc0[c + 2] = fmaf(in[kw], w4.z, acc0[c + 2]); acc1[c + 2] = fmaf(in[kw + 2], w4.z, acc1[c + 2]);
                    acc0[c + 3] = fmaf(in[kw], w4.w, acc0[c + 3]); acc1[c + 3] = fmaf(in[kw + 2], w4.w, acc1[c + 3]);
                }
            }
        }
    }
    const size_t ob = ((size_t)tp * B + b) * CH * W2;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        float v0 = acc0[c] * scale[c] + shift[c];
        float v1 = acc1[c] * scale[c] + shift[c];
        v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
        const size_t o = ob + c * W2 + wo;
        if (seq) { seq[o] = v0; if (wo + 1 < W2) seq[o + 1] = v1; }
        if (seq_hi) {   // split-bf16 planes for the bf16x3 input projection
            __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
            seq_hi[o] = *reinterpret_cast<unsigned short *>(&h0); seq_lo[o] = *reinterpret_cast<unsigned short *>(&l0);
            if (wo + 1 < W2) { seq_hi[o + 1] = *reinterpret_cast<unsigned short *>(&h1); seq_lo[o + 1] = *reinterpret_cast<unsigned short *>(&l1); }
        }
    }
}

int launch_conv1(const float *y0, const float *w_t, const float *scale, const float *shift, float *seq, SplitPtr sp, int B,
                 int T, int W1, int ch, hipStream_t st) {
    int W2 = (W1 + 2 - 3) / 2 + 1;
    int pairs = (W2 + 1) / 2;
    dim3 grid(((T / 2) * pairs + 255) / 256, 1, B), block(256);
    if (ch == 32) hipLaunchKernelGGL(conv1_kernel<32>, grid, block, 0, st, y0, w_t, scale, shift, seq, sp.hi, sp.lo, B, T, W1, W2);
    else if (ch == 4) hipLaunchKernelGGL(conv1_kernel<4>, grid, block, 0, st, y0, w_t, scale, shift, seq, sp.hi, sp.lo, B, T, W1, W2);
    else { set_error("conv1: channels=%d not built (32 or 4)", ch); return MDD_ERR_ARG; }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
