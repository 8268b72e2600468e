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

// ---------------------------------------------------------------- fused conv0 -> conv1 (split-bf16 MFMA)
// One pass over x produces the BiLSTM input rows directly: conv0 (+BN+ReLU) is recomputed per output row into LDS
// (3 conv0 rows x 122 columns x 32 channels, bf16 hi/lo) and conv1 runs as an implicit GEMM on the bf16 matrix cores
//   out[w', co] = sum_k A[w', k] . W1[co, k],   k = (kh, kw, ci), K = 288,  M = 61 (two 32-row tiles),  N = 32
// with the bf16x3 product form.  The 500 MB-per-batch conv0 activation (y0) never exists in HBM: the kernel reads
// x once (5 rows per output row) and writes the [T'*B, 1952] split-bf16 rows the input projection consumes.
// A workgroup is persistent over output rows (grid-stride), so its W1 fragments stay in registers.
// LDS: x tile [5][248] f32 | y0 hi/lo [3][124 cols][80 B] (32 ch x bf16 + 16 B pad: 2-way-conflict fragment reads)
//      | K-half reduction [2][32x32] f32 | output row [1952] hi, lo.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CF_XLD = 248, CF_COLB = 80, CF_NCOL = 124, CF_YPLANE = 3 * CF_NCOL * CF_COLB;   // 29760 B

// NP = 2: the split-bf16 x3 form (operands as hi + lo, products hi.lo + lo.hi + hi.hi; output: two row-major planes).
// NP = 3: the fp32-grade form of the f32x6 mode (gemm_bf16x6.hip): every operand as hi + mid + lo = all 24 significand bits, the six
// products down to 2^-24 of a product, smallest first; the contractions here are short (9 taps; 288 / 2 per wave), so one accumulator
// does.  Output: three planes in the K-TILE-MAJOR order the f32x6 projection GEMM streams (plane[kt][row][32], kt = column / 32), i.e.
// the conv output never exists as fp32 and needs no split pass.  w1l holds the mid plane followed by the lo plane in that case.
template <int NP>
__global__ __launch_bounds__(256, NP == 2 ? 2 : 1) void conv_fused_kernel(const float *__restrict__ x, const float *__restrict__ w0,
                                                            const float *__restrict__ sc0, const float *__restrict__ sh0,
                                                            const unsigned short *__restrict__ w1h, const unsigned short *__restrict__ w1l,
                                                            const float *__restrict__ sc1, const float *__restrict__ sh1,
                                                            unsigned short *__restrict__ out_hi, unsigned short *__restrict__ out_lo,
                                                            float *__restrict__ out_f32, int B, int T, int Traw, int S, int seg) {
    constexpr int F = 243, W1 = 122, W2 = 61, CH = 32, ROW = CH * W2;   // 1952
    constexpr int D0 = F / 3;                                           // raw feature width when x holds unstacked frames
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);                                   // [5][CF_XLD]
    unsigned char *yh = smem + 5 * CF_XLD * 4, *yl = yh + CF_YPLANE;               // conv0 tile, hi / lo
    unsigned char *ym = yl + CF_YPLANE;                                            // (NP = 3) mid
    float *red = reinterpret_cast<float *>(yl + (NP - 1) * CF_YPLANE);             // [2][1024]
    unsigned short *oh = reinterpret_cast<unsigned short *>(red + 2048), *ol = oh + ROW, *om = ol + ROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave & 1, kh2 = wave >> 1;                                      // M tile, K half of this wave
    const int li = lane & 31, half = lane >> 5;
    const int Tp = T / 2;

    // resident B fragments: W1[co = li][k], this wave's 9 k-steps of 16
    bf16x8 bwh[9], bwl[9], bwm[NP == 3 ? 9 : 1];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int kb = kh2 * 9 + i;
        bwh[i] = *reinterpret_cast<const bf16x8 *>(w1h + (size_t)li * 288 + kb * 16 + half * 8);
        if (NP == 3) {
            bwm[i] = *reinterpret_cast<const bf16x8 *>(w1l + (size_t)li * 288 + kb * 16 + half * 8);
            bwl[i] = *reinterpret_cast<const bf16x8 *>(w1l + (size_t)CH * 288 + (size_t)li * 288 + kb * 16 + half * 8);
        } else bwl[i] = *reinterpret_cast<const bf16x8 *>(w1l + (size_t)li * 288 + kb * 16 + half * 8);
    }
    // conv0 weights as resident A fragments: lane (channel li, k-slice half): taps half*8 .. half*8+7 (taps >= 9 are zero)
    bf16x8 w0h, w0l, w0m;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int tap = half * 8 + j;
        const float wv = tap < 9 ? w0[li * 9 + tap] : 0.f;
        w0h[j] = (__bf16)wv;
        const float r1 = wv - (float)w0h[j];
        if (NP == 3) { w0m[j] = (__bf16)r1; w0l[j] = (__bf16)(r1 - (float)w0m[j]); }
        else { w0l[j] = (__bf16)r1; w0m[j] = (__bf16)0.f; }
    }
    float c0s[16], c0h[16];      // folded BN of the 16 channels this lane receives in D: reg q -> channel (q&3) + 8*(q>>2) + 4*half
#pragma unroll
    for (int q = 0; q < 16; q++) { const int ch = (q & 3) + 8 * (q >> 2) + 4 * half; c0s[q] = sc0[ch]; c0h[q] = sh0[ch]; }
    // zero the left pad column of the conv0 tile once (wcol = -1)
    for (int i = tid; i < 3 * NP * CF_COLB / 4; i += 256) {
        const int r = i / (NP * CF_COLB / 4), rem = i - r * (NP * CF_COLB / 4), pl = rem / (CF_COLB / 4);
        reinterpret_cast<unsigned int *>((pl == 0 ? yh : pl == 1 ? yl : ym) + r * CF_NCOL * CF_COLB)[rem % (CF_COLB / 4)] = 0u;
    }

    // x rows 2tp-2 .. 2tp+2, columns -1 .. 243 (zero outside) of one output row: 5*248 elements, 5 per thread.  The next
    // row's elements are requested before this row's arithmetic starts and parked in registers (a row is otherwise a
    // chain global load -> LDS -> conv0 -> conv1 with the ~1 us load latency exposed every time).
    constexpr int XPT = (5 * CF_XLD + 255) / 256;
    auto load_x = [&](int tp, int b, float (&dst)[XPT]) {
#pragma unroll
        for (int k = 0; k < XPT; k++) {
            const int i = tid + 256 * k, r = i / CF_XLD, c = i - r * CF_XLD - 1, ti = 2 * tp - 2 + r;
            float v = 0.f;
            if (i < 5 * CF_XLD) {
                if (Traw == 0) {
                    if (ti >= 0 && ti < T && c >= 0 && c < F) v = x[((size_t)b * T + ti) * F + c];
                } else {   // x = raw frames [B, Traw, 81]: stacked frame ti = raw frames 2ti, 2ti+1, 2ti+2 (last frame repeated past the
                           // end, tools.py:207-220), frames 0, 2, 4, .. kept (:222-227), zero rows up to an even count (data_loader.py:140-142)
                    const int j = c / D0, fr = min(2 * ti + j, Traw - 1);
                    if (ti >= 0 && 2 * ti < Traw && c >= 0 && c < F) v = x[((size_t)b * Traw + fr) * D0 + (c - j * D0)];
                }
            }
            dst[k] = v;
        }
    };
    // A workgroup walks consecutive output rows tp of ONE utterance (segment sidx of S per utterance): conv0 row 2tp+1 of
    // one output row is conv0 row 2(tp+1)-1 of the next, so after the first row of a segment only two of the three conv0
    // rows are computed; the tile's three row slots rotate (logical row r lives in slot (r + rot) % 3).
    float xnext[XPT];
    const int b = blockIdx.x / S, sidx = blockIdx.x - b * S;
    const int tp_begin = sidx * seg, tp_end = min(Tp, tp_begin + seg);
    if (tp_begin < tp_end) load_x(tp_begin, b, xnext);
    int rot = 0;
    for (int tp = tp_begin; tp < tp_end; tp++) {
        const int rowid = tp * B + b;
        const bool fresh = tp == tp_begin;
        if (!fresh) rot = rot == 0 ? 2 : rot - 1;                                  // (rot + 2) % 3: old logical row 2 becomes row 0
        __syncthreads();                                                           // previous row's LDS fully consumed
#pragma unroll
        for (int k = 0; k < XPT; k++) if (tid + 256 * k < 5 * CF_XLD) xs[tid + 256 * k] = xnext[k];
        if (tp + 1 < tp_end) load_x(tp + 1, b, xnext);   // (all five x rows: skipping the one that only feeds the reused conv0 row costs more in branches than it saves)
        __syncthreads();
        // ---- conv0 on the matrix cores: D[ch, pos] = W0[ch, k] . P[k, pos], k = 9 taps padded to 16 (one MFMA k-step),
        // bf16x3.  A = weights (resident fragments), B = the 3x3 patch of each of the 3*122 = 366 positions (12 tiles of
        // 32, 3 per wave), built from the fp32 x tile.  D puts the position on the lane and 16 channels in the registers
        // (4 groups of 4 consecutive channels), exactly the [r][wcol+1][ci] rows conv1 reads: 8-byte LDS stores.
        for (int pt = wave; pt < (fresh ? 12 : 8); pt += 4) {
            const int p = (fresh ? 0 : W1) + pt * 32 + li, pc = min(p, 3 * W1 - 1);
            const int r = pc / W1, wc = pc - r * W1, ti = 2 * tp - 1 + r;
            const int rs = r + rot >= 3 ? r + rot - 3 : r + rot;                   // row slot in the tile
            const bool rowok = ti >= 0 && ti < T && p < 3 * W1;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int tap = half * 8 + j, kh = tap / 3, kw = tap - kh * 3;           // taps 9..15 are padding
                v[j] = (tap < 9) ? xs[(r + kh) * CF_XLD + 2 * wc + kw] : 0.f;
            }
            bf16x8 bh, bl, bm;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                bh[j] = (__bf16)v[j];
                const float r1 = v[j] - (float)bh[j];
                if (NP == 3) { bm[j] = (__bf16)r1; bl[j] = (__bf16)(r1 - (float)bm[j]); }
                else bl[j] = (__bf16)r1;
            }
            f32x16 d;
#pragma unroll
            for (int q = 0; q < 16; q++) d[q] = 0.f;
            if (NP == 3) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0m, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, bl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0l, bh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0m, bh, d, 0, 0, 0);
            } else {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, bl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0l, bh, d, 0, 0, 0);
            }
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, bh, d, 0, 0, 0);
            if (p < 3 * W1) {
                unsigned char *dh = yh + (rs * CF_NCOL + wc + 1) * CF_COLB, *dl = yl + (rs * CF_NCOL + wc + 1) * CF_COLB;
                unsigned char *dm = ym + (rs * CF_NCOL + wc + 1) * CF_COLB;
#pragma unroll
                for (int g = 0; g < 4; g++) {      // registers 4g..4g+3 = channels 8g + 4*half + 0..3
                    unsigned short hb[4], lb[4], mb[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float a = d[4 * g + e] * c0s[4 * g + e] + c0h[4 * g + e];
                        a = (rowok && a > 0.f) ? a : 0.f;
                        __bf16 h = (__bf16)a;
                        const float r1 = a - (float)h;
                        hb[e] = *reinterpret_cast<unsigned short *>(&h);
                        if (NP == 3) {
                            __bf16 mm_ = (__bf16)r1, l = (__bf16)(r1 - (float)mm_);
                            mb[e] = *reinterpret_cast<unsigned short *>(&mm_); lb[e] = *reinterpret_cast<unsigned short *>(&l);
                        } else { __bf16 l = (__bf16)r1; lb[e] = *reinterpret_cast<unsigned short *>(&l); }
                    }
                    const int cb = (8 * g + 4 * half) * 2;
                    *reinterpret_cast<uint2 *>(dh + cb) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
                    *reinterpret_cast<uint2 *>(dl + cb) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
                    if (NP == 3) *reinterpret_cast<uint2 *>(dm + cb) = make_uint2(mb[0] | ((unsigned)mb[1] << 16), mb[2] | ((unsigned)mb[3] << 16));
                }
            }
        }
        __syncthreads();
        // ---- conv1: this wave = M tile mt (w' = mt*32 + li), K half kh2
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const int m = min(mt * 32 + li, W2 - 1);                                   // rows >= 61 recompute row 60, never stored
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int kb = kh2 * 9 + i, kk = kb >> 1, kh = kk / 3, kw = kk - kh * 3;
            const int khs = kh + rot >= 3 ? kh + rot - 3 : kh + rot;
            const int off = (khs * CF_NCOL + 2 * m + kw) * CF_COLB + ((kb & 1) * 16 + half * 8) * 2;   // col index = (2m+kw-1)+1
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(yh + off);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(yl + off);
            if (NP == 3) {
                const bf16x8 am = *reinterpret_cast<const bf16x8 *>(ym + off);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bwm[i], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bwl[i], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bwh[i], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bwm[i], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bwh[i], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bwl[i], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bwh[i], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bwh[i], acc, 0, 0, 0);
        }
        // ---- reduce the two K halves, BN + ReLU, transpose to the output row through LDS
        if (kh2 == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) red[mt * 1024 + r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kh2 == 0) {
            const float s1 = sc1[li], h1 = sh1[li];                                 // C/D: col = lane&31 = co
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int wo = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;         // row = w'
                if (wo < W2) {
                    float v = (acc[r] + red[mt * 1024 + r * 64 + lane]) * s1 + h1;
                    v = v > 0.f ? v : 0.f;
                    __bf16 hb = (__bf16)v;
                    const float r1 = v - (float)hb;
                    oh[li * W2 + wo] = *reinterpret_cast<unsigned short *>(&hb);
                    if (NP == 3) {
                        __bf16 mb = (__bf16)r1, lb = (__bf16)(r1 - (float)mb);
                        om[li * W2 + wo] = *reinterpret_cast<unsigned short *>(&mb);
                        ol[li * W2 + wo] = *reinterpret_cast<unsigned short *>(&lb);
                    } else { __bf16 lb = (__bf16)r1; ol[li * W2 + wo] = *reinterpret_cast<unsigned short *>(&lb); }
                }
            }
        }
        __syncthreads();
        // ---- row store.  NP = 3: 16-byte chunk q of a plane row = columns 8q .. 8q+7 = K-tile q / 4, offset (q % 4) * 8 of the tile-major
        // plane (out_hi = hi plane; mid and lo follow at plane strides of rows_total * 1952 elements, passed in out_lo as the mid plane)
        if (NP == 3) {
            const size_t mrows = (size_t)Tp * B;
            if (tid < 244) {
                const size_t dst = (((size_t)(tid >> 2) * mrows + rowid) * 32 + (tid & 3) * 8);
                *reinterpret_cast<u32x4 *>(out_hi + dst) = *reinterpret_cast<const u32x4 *>(oh + tid * 8);
                *reinterpret_cast<u32x4 *>(out_lo + dst) = *reinterpret_cast<const u32x4 *>(om + tid * 8);
                *reinterpret_cast<u32x4 *>(out_lo + mrows * ROW + dst) = *reinterpret_cast<const u32x4 *>(ol + tid * 8);
            }
            if (out_f32)
                for (int i = tid; i < ROW; i += 256)
                    out_f32[(size_t)rowid * ROW + i] = (__uint_as_float((unsigned)oh[i] << 16) + __uint_as_float((unsigned)om[i] << 16)) + __uint_as_float((unsigned)ol[i] << 16);
        } else {
            const size_t base = (size_t)rowid * ROW;
            if (tid < 244) *reinterpret_cast<u32x4 *>(out_hi + base + tid * 8) = *reinterpret_cast<const u32x4 *>(oh + tid * 8);
            else if (tid - 244 < 12) {   // 12 threads left in this pass: start the lo plane
                const int q = tid - 244;
                *reinterpret_cast<u32x4 *>(out_lo + base + q * 8) = *reinterpret_cast<const u32x4 *>(ol + q * 8);
            }
            if (tid < 232) {
                const int q = tid + 12;
                *reinterpret_cast<u32x4 *>(out_lo + base + q * 8) = *reinterpret_cast<const u32x4 *>(ol + q * 8);
            }
            if (out_f32)
                for (int i = tid; i < ROW; i += 256)
                    out_f32[base + i] = __uint_as_float((unsigned)oh[i] << 16) + __uint_as_float((unsigned)ol[i] << 16);
        }
    }
}

size_t conv_fused_smem(int np) { return 5 * CF_XLD * 4 + (size_t)np * CF_YPLANE + 2048 * 4 + (size_t)np * 1952 * 2; }

int launch_conv_fused(const float *x, const float *w0, const float *sc0, const float *sh0, SplitPtr w1, const float *sc1,
                      const float *sh1, SplitPtr out, float *out_f32, int B, int T, int Traw, hipStream_t st) {
    // segments of consecutive output rows per utterance: about 1024 workgroups (two rounds of the 512 that fit the chip)
    const int Tp = T / 2;
    if (Tp <= 0 || B <= 0) return MDD_OK;
    int S = (1024 + B - 1) / B;
    S = S < 1 ? 1 : (S > Tp ? Tp : S);
    const int seg = (Tp + S - 1) / S;
    S = (Tp + seg - 1) / seg;                                                      // no empty segments
    hipLaunchKernelGGL(conv_fused_kernel<2>, dim3(B * S), dim3(256), conv_fused_smem(2), st, x, w0, sc0, sh0, w1.hi, w1.lo, sc1, sh1,
                       out.hi, out.lo, out_f32, B, T, Traw, S, seg);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// f32x6 form: w1_3 = conv1 weights [co][kh][kw][ci] as three consecutive row-major planes (hi | mid | lo, 32 * 288 elements each);
// out3 = three consecutive K-tile-major planes of (T/2 * B) x 1952 elements (the A operand of launch_gemm_f32x6)
int launch_conv_fused3(const float *x, const float *w0, const float *sc0, const float *sh0, const unsigned short *w1_3, const float *sc1,
                       const float *sh1, unsigned short *out3, float *out_f32, int B, int T, int Traw, hipStream_t st) {
    const int Tp = T / 2;
    if (Tp <= 0 || B <= 0) return MDD_OK;
    int S = (512 + B - 1) / B;                                                     // one workgroup per CU (114 KB of LDS): two rounds of 256
    S = S < 1 ? 1 : (S > Tp ? Tp : S);
    const int seg = (Tp + S - 1) / S;
    S = (Tp + seg - 1) / seg;
    const size_t plane = (size_t)Tp * B * 1952;
    hipLaunchKernelGGL(conv_fused_kernel<3>, dim3(B * S), dim3(256), conv_fused_smem(3), st, x, w0, sc0, sh0, w1_3, w1_3 + 32 * 288, sc1, sh1,
                       out3, out3 + plane, out_f32, B, T, Traw, S, seg);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int init_conv_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_fused_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    return MDD_OK;
}

}  // namespace mdd
