// Kernels of the training step (train-mode forward pieces and every backward piece that is not a GEMM).
//
// Reference: run_epoch (AA/steps/train_ctc.py:28-105) drives CTC_Model.forward in train mode (AA/models/model_ctc.py:160-223:
// BatchNorm with batch statistics, Dropout(p) after each LayerCNN and BatchRNN) and autograd's backward of it.  Arithmetic is
// exact fp32 (the reference trains in fp32); contractions go through launch_gemm_f32 (gemm.hip).
//
// Activation layouts of the training path (everything is "rows x features", row-major):
//   conv activations   channels-last  [B, T, W, ch]        rows = (b, t, w)
//   sequence buffers   time-major     [T', B, feat]        rows = (t, b)       (what CTC_Model.forward returns)
//   dropout masks      the REFERENCE tensor's layout at that site ([B,ch,T,W] after a LayerCNN, [T',B,2H] after a BatchRNN),
//                      one byte per element, so a test can hand over the very mask the reference drew.
#include "mdd_internal.h"
#include "train.h"

namespace mdd {

// ------------------------------------------------------------------------------------------------ dropout masks
// counter-based generator: mask[i] = hash(seed, site, i) keeps with probability 1 - p
__global__ void dropout_mask_kernel(unsigned char *mask, size_t n, unsigned long long seed, unsigned site, float p) {
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1) + ((unsigned long long)site << 56);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        mask[i] = (unsigned)(z >> 32) >= thresh ? 1 : 0;
    }
}
int launch_dropout_mask(unsigned char *mask, size_t n, unsigned long long seed, unsigned site, float p, hipStream_t st) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, mask, n, seed, site, p);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ conv0 (1 -> ch), direct
// z0[(b,t,w), c] = bias[c] + sum_{kh,kw} x[b, t+kh-1, 2w+kw-1] * w0[c, kh, kw]      (Conv2d k3x3, stride (1,2), pad 1)
template <int CH>
__global__ __launch_bounds__(256) void conv0_train_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                              float *__restrict__ z, int B, int T, int F, int W1) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x, npos = (size_t)B * T * W1;
    if (pos >= npos) return;
    const int wo = (int)(pos % W1), t = (int)((pos / W1) % T), b = (int)(pos / ((size_t)W1 * T));
    float in[9];
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const int ti = t + kh - 1, fi = wo * 2 + kw - 1;
            in[kh * 3 + kw] = (ti >= 0 && ti < T && fi >= 0 && fi < F) ? x[((size_t)b * T + ti) * F + fi] : 0.f;
        }
    float *o = z + pos * CH;
#pragma unroll 4
    for (int c = 0; c < CH; c++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) acc = fmaf(in[k], w[c * 9 + k], acc);
        o[c] = acc + bias[c];
    }
}
int launch_conv0_train_fwd(const float *x, const float *w, const float *bias, float *z, int B, int T, int F, int ch, hipStream_t st) {
    const int W1 = (F + 2 - 3) / 2 + 1;
    const size_t npos = (size_t)B * T * W1;
    dim3 grid((unsigned)((npos + 255) / 256)), block(256);
    if (ch == 32) hipLaunchKernelGGL(conv0_train_fwd_kernel<32>, grid, block, 0, st, x, w, bias, z, B, T, F, W1);
    else if (ch == 4) hipLaunchKernelGGL(conv0_train_fwd_kernel<4>, grid, block, 0, st, x, w, bias, z, B, T, F, W1);
    else { set_error("conv0 (train): channels=%d not built", ch); return MDD_ERR_ARG; }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// dW0[c, k] = sum_pos dz0[pos, c] * patch[pos, k];  db0[c] = sum_pos dz0[pos, c].  Partial sums per workgroup in LDS (fp32), one
// fp64 atomic per (workgroup, entry): the order of the adds moves the result by ~1e-16 relative, far below fp32 rounding.
template <int CH>
__global__ __launch_bounds__(256) void conv0_train_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dz, double *__restrict__ acc /*[CH*10]*/,
                                                              int B, int T, int F, int W1, int rows_per_wave) {
    // A wave walks whole (b, t) rows of W1 output positions.  lane = (position slot, channel): the CH lanes of a slot read one contiguous
    // dz row and the same nine input samples (broadcast); a lane keeps its channel's nine tap sums and the bias sum (k = 9) in registers
    // (rows_per_wave * W1 * CH / 64 terms in fp32), then fp64 across workgroups.
    constexpr int PPW = 64 / CH;
    __shared__ float part[4][CH * 10];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane % CH, slot = lane / CH;
    const int nrow = B * T, row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
    float sum[10];
#pragma unroll
    for (int k = 0; k < 10; k++) sum[k] = 0.f;
    for (int row = row0; row < min(row0 + rows_per_wave, nrow); row++) {
        const int t = row % T;
        const float *xr = x + (size_t)row * F;                      // input row t of utterance b; rows t-1 / t+1 exist unless at the edge
        const bool up = t > 0, dn = t + 1 < T;
        const float *dzr = dz + (size_t)row * W1 * CH;
        for (int wo = slot; wo < W1; wo += PPW) {
            const float dzv = dzr[wo * CH + c];
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int fi = wo * 2 + kw - 1;
                const bool in = fi >= 0 && fi < F;
                const float p0 = (in && up) ? xr[fi - F] : 0.f, p1 = in ? xr[fi] : 0.f, p2 = (in && dn) ? xr[fi + F] : 0.f;
                sum[kw] = fmaf(dzv, p0, sum[kw]); sum[3 + kw] = fmaf(dzv, p1, sum[3 + kw]); sum[6 + kw] = fmaf(dzv, p2, sum[6 + kw]);
            }
            sum[9] += dzv;
        }
    }
#pragma unroll
    for (int k = 0; k < 10; k++)
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) sum[k] += __shfl_xor(sum[k], o);
    if (lane < CH) {
#pragma unroll
        for (int k = 0; k < 10; k++) part[wave][c * 10 + k] = sum[k];
    }
    __syncthreads();
    for (int e = tid; e < CH * 10; e += 256) atomicAdd(&acc[e], (double)((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])));
}
__global__ void conv0_bwd_finish_kernel(const double *acc, float *dw, float *db, int ch) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ch * 10) return;
    const int c = e / 10, k = e - c * 10;
    if (k < 9) dw[c * 9 + k] = (float)acc[e]; else db[c] = (float)acc[e];
}
int launch_conv0_train_bwd(const float *x, const float *dz, double *acc, float *dw, float *db, int B, int T, int F, int ch, hipStream_t st) {
    const int W1 = (F + 2 - 3) / 2 + 1;
    const size_t npos = (size_t)B * T * W1;
    MDD_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(double) * ch * 10, st));
    (void)npos;
    const int rpw = 4;                                             // (b, t) rows per wave: ~500 workgroups at B*T = 8000
    dim3 grid((unsigned)((B * T + 4 * rpw - 1) / (4 * rpw))), block(256);
    if (ch == 32) hipLaunchKernelGGL(conv0_train_bwd_kernel<32>, grid, block, 0, st, x, dz, acc, B, T, F, W1, rpw);
    else if (ch == 4) hipLaunchKernelGGL(conv0_train_bwd_kernel<4>, grid, block, 0, st, x, dz, acc, B, T, F, W1, rpw);
    else { set_error("conv0 (train): channels=%d not built", ch); return MDD_ERR_ARG; }
    hipLaunchKernelGGL(conv0_bwd_finish_kernel, dim3((ch * 10 + 63) / 64), dim3(64), 0, st, acc, dw, db, ch);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ conv1 (ch -> ch, stride 2x2) as im2col + GEMM
// col[(b,t',w'), (kh*3+kw)*ch + ci] = a0[b, 2t'+kh-1, 2w'+kw-1, ci]   (zero outside)
__global__ void im2col1_kernel(const float *__restrict__ a0, float *__restrict__ col, int B, int T, int W1, int W2, int ch) {
    const int Tp = T / 2, Kc = 9 * ch, c4n = ch / 4;
    const size_t n = (size_t)B * Tp * W2 * 9 * c4n;    // one thread per 16 bytes of col, in col's own order: stores (and each tap's loads) are contiguous
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const size_t pt = i / c4n;
        const int tap = (int)(pt % 9);
        const size_t pos = pt / 9;
        const int wo = (int)(pos % W2), tp = (int)((pos / W2) % Tp), b = (int)(pos / ((size_t)W2 * Tp));
        const int ti = 2 * tp + tap / 3 - 1, wi = 2 * wo + tap % 3 - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ti >= 0 && ti < T && wi >= 0 && wi < W1) v = *reinterpret_cast<const float4 *>(a0 + (((size_t)b * T + ti) * W1 + wi) * ch + c);
        *reinterpret_cast<float4 *>(col + pos * Kc + tap * ch + c) = v;
    }
}
// da0[b,t,w,ci] = sum over the (<= 4) output positions / taps that read it of dcol
__global__ void col2im1_kernel(const float *__restrict__ dcol, float *__restrict__ da0, int B, int T, int W1, int W2, int ch) {
    const int Tp = T / 2, Kc = 9 * ch, c4n = ch / 4;
    const size_t n = (size_t)B * T * W1 * c4n;                      // four channels of an input position per thread (the taps' adds in the same order as before)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c4n) * 4;
        const size_t p = i / c4n;
        const int w = (int)(p % W1), t = (int)((p / W1) % T), b = (int)(p / ((size_t)W1 * T));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int tt = t + 1 - kh;
            if (tt < 0 || (tt & 1) || (tt >> 1) >= Tp) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int ww = w + 1 - kw;
                if (ww < 0 || (ww & 1) || (ww >> 1) >= W2) continue;
                const float4 v = *reinterpret_cast<const float4 *>(dcol + (((size_t)b * Tp + (tt >> 1)) * W2 + (ww >> 1)) * Kc + (kh * 3 + kw) * ch + ci);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        *reinterpret_cast<float4 *>(da0 + p * ch + ci) = acc;
    }
}
// W1 [co][ci][kh][kw] <-> W1r [co][(kh*3+kw)*ch + ci]
__global__ void pack_w1_kernel(const float *__restrict__ w, float *__restrict__ wr, int ch, int to_packed) {
    const int n = ch * ch * 9, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = i % 9, ci = (i / 9) % ch, co = i / (9 * ch);
    const int j = (co * 9 + k) * ch + ci;
    if (to_packed) wr[j] = w[i]; else wr[i] = w[j];     // unpack: wr is the reference-layout destination, w the packed source
}
int launch_im2col1(const float *a0, float *col, int B, int T, int W1, int W2, int ch, hipStream_t st) {
    hipLaunchKernelGGL(im2col1_kernel, dim3(4096), dim3(256), 0, st, a0, col, B, T, W1, W2, ch);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
int launch_col2im1(const float *dcol, float *da0, int B, int T, int W1, int W2, int ch, hipStream_t st) {
    hipLaunchKernelGGL(col2im1_kernel, dim3(8192), dim3(256), 0, st, dcol, da0, B, T, W1, W2, ch);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
int launch_pack_w1(const float *src, float *dst, int ch, bool to_packed, hipStream_t st) {
    hipLaunchKernelGGL(pack_w1_kernel, dim3((ch * ch * 9 + 255) / 256), dim3(256), 0, st, src, dst, ch, to_packed ? 1 : 0);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}

// a1 [B,T',W2,ch] (channels-last) <-> seq [T',B, c*W2 + w]   (model_ctc.py:176-181; feature index = c*W2 + w)
__global__ void cnn_seq_kernel(float *__restrict__ a1, float *__restrict__ seq, int B, int Tp, int W2, int ch, int to_seq) {
    const size_t n = (size_t)B * Tp * W2 * ch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        const size_t p = i / ch;
        const int w = (int)(p % W2), t = (int)((p / W2) % Tp), b = (int)(p / ((size_t)W2 * Tp));
        const size_t j = ((size_t)t * B + b) * ((size_t)ch * W2) + (size_t)c * W2 + w;
        if (to_seq) seq[j] = a1[i]; else a1[i] = seq[j];
    }
}
int launch_cnn_seq(float *a1, float *seq, int B, int Tp, int W2, int ch, bool to_seq, hipStream_t st) {
    hipLaunchKernelGGL(cnn_seq_kernel, dim3(4096), dim3(256), 0, st, a1, seq, B, Tp, W2, ch, to_seq ? 1 : 0);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm over rows (train mode)
// Row-major x [R, F]; feature f is normalised over the R rows (BatchNorm2d on channels-last conv rows, BatchNorm1d on the
// [T'*B, F] rows of a sequence buffer: model_ctc.py:41-43, 153).  Column sums in fp64.
// mode 0: v = x                      (forward statistics)
// mode 1: dy = g (plain)             sums of dy and dy * xhat            (BatchNorm backward)
// mode 2: dy = g * mask * scale * (bn(x) > 0)   the same behind ReLU + dropout (LayerCNN: conv -> BN -> ReLU -> Dropout)
__device__ __forceinline__ float bn_site_dy(const BnSite &s, float g, float x, float mean, float invstd, float gamma, float beta, size_t row, int f, int F) {
    const float y = (x - mean) * invstd * gamma + beta;
    const float m = s.mask ? (float)s.mask[row * F + f] * s.scale : 1.f;
    return y > 0.f ? g * m : 0.f;
}
// dropout mask of a conv site, [B][F][T*W] (the reference's tensor order) -> [B][T*W][F] (the order of the channels-last rows)
__global__ __launch_bounds__(256) void mask_rows_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, int F, int TW) {
    __shared__ unsigned char tile[32][33];
    const size_t base = (size_t)blockIdx.z * F * TW;
    const int p0 = blockIdx.x * 32, f0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j, p_ = p0 + tx;
        tile[j][tx] = (f < F && p_ < TW) ? src[base + (size_t)f * TW + p_] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p_ = p0 + j, f = f0 + tx;
        if (p_ < TW && f < F) dst[base + (size_t)p_ * F + f] = tile[tx][j];
    }
}
int launch_mask_rows(const unsigned char *src, unsigned char *dst, int B, int F, int TW, hipStream_t st) {
    hipLaunchKernelGGL(mask_rows_kernel, dim3((TW + 31) / 32, (F + 31) / 32, B), dim3(256), 0, st, src, dst, F, TW);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
template <int MODE>
__global__ __launch_bounds__(256) void col_stats_kernel(const float *__restrict__ x, const float *__restrict__ g, size_t R, int F, int rows_per_wg,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, BnSite site, double *__restrict__ s1, double *__restrict__ s2) {
    __shared__ float p1[4][64], p2[4][64];
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int f = blockIdx.x * 64 + cl;
    const size_t r0 = (size_t)blockIdx.y * rows_per_wg;
    float a1 = 0.f, a2 = 0.f;
    if (f < F) {
        float mu = 0.f, is = 0.f, ga = 1.f, be = 0.f;
        if (MODE != 0) { mu = mean[f]; is = invstd[f]; }
        if (MODE == 2) { ga = gamma[f]; be = beta[f]; }
        for (size_t r = r0 + rl; r < r0 + rows_per_wg && r < R; r += 4) {
            const float xv = x[r * F + f];
            if (MODE == 0) { a1 += xv; a2 = fmaf(xv, xv, a2); }
            else {
                const float dy = MODE == 1 ? g[r * F + f] : bn_site_dy(site, g[r * F + f], xv, mu, is, ga, be, r, f, F);
                a1 += dy; a2 = fmaf(dy, (xv - mu) * is, a2);
            }
        }
    }
    p1[rl][cl] = a1; p2[rl][cl] = a2;
    __syncthreads();
    if (tid < 64 && f < F) {
        atomicAdd(&s1[f], (double)p1[0][tid] + (double)p1[1][tid] + (double)p1[2][tid] + (double)p1[3][tid]);
        atomicAdd(&s2[f], (double)p2[0][tid] + (double)p2[1][tid] + (double)p2[2][tid] + (double)p2[3][tid]);
    }
}
// F = 32 (the conv sites: 2e7 elements in 6e5 rows): a lane owns four channels of a row (one 16-byte load), eight lanes a row, a
// wave eight rows per load instruction (1 KB, contiguous); partial sums per lane in fp32 over <= 32 rows, then fp64.
template <int MODE>
__global__ __launch_bounds__(256) void col_stats32_kernel(const float *__restrict__ x, const float *__restrict__ g, size_t R, int rows_per_wg,
                                                          const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, BnSite site, double *__restrict__ s1, double *__restrict__ s2) {
    constexpr int F = 32;
    __shared__ float q1[4][F], q2[4][F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const size_t r0 = (size_t)blockIdx.x * rows_per_wg, r1 = min(r0 + (size_t)rows_per_wg, R);
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (MODE != 0) { mu[j] = mean[c4 + j]; is[j] = invstd[c4 + j]; }
        if (MODE == 2) { ga[j] = gamma[c4 + j]; be[j] = beta[c4 + j]; }
    }
    for (size_t r = r0 + wave * 8 + rsub; r < r1; r += 32) {
        const float4 xq = *reinterpret_cast<const float4 *>(x + r * F + c4);
        const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) { a1[j] += xv[j]; a2[j] = fmaf(xv[j], xv[j], a2[j]); }
        } else {
            const float4 gq = *reinterpret_cast<const float4 *>(g + r * F + c4);
            const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
            uchar4 mq = make_uchar4(1, 1, 1, 1);
            if (MODE == 2 && site.mask) mq = *reinterpret_cast<const uchar4 *>(site.mask + r * F + c4);
            const unsigned char mv[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float dy = gv[j];
                if (MODE == 2) {
                    const float y = (xv[j] - mu[j]) * is[j] * ga[j] + be[j];
                    const float m = site.mask ? (float)mv[j] * site.scale : 1.f;
                    dy = y > 0.f ? gv[j] * m : 0.f;
                }
                a1[j] += dy; a2[j] = fmaf(dy, (xv[j] - mu[j]) * is[j], a2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { a1[j] += __shfl_xor(a1[j], o); a2[j] += __shfl_xor(a2[j], o); }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) { q1[wave][c4 + j] = a1[j]; q2[wave][c4 + j] = a2[j]; }
    }
    __syncthreads();
    if (tid < F) {
        atomicAdd(&s1[tid], ((double)q1[0][tid] + (double)q1[1][tid]) + ((double)q1[2][tid] + (double)q1[3][tid]));
        atomicAdd(&s2[tid], ((double)q2[0][tid] + (double)q2[1][tid]) + ((double)q2[2][tid] + (double)q2[3][tid]));
    }
}
template <int MODE>
static void launch_col_stats(const float *x, const float *g, size_t R, int F, const float *mean, const float *invstd, const float *gamma, const float *beta,
                             const BnSite &site, double *s1, double *s2, hipStream_t st) {
    if (F == 32 && (((size_t)x | (size_t)g) & 15) == 0) {
        const int per = 1024;
        hipLaunchKernelGGL(col_stats32_kernel<MODE>, dim3((unsigned)((R + per - 1) / per)), dim3(256), 0, st, x, g, R, per, mean, invstd, gamma, beta, site, s1, s2);
        return;
    }
    int per = (int)std::max<size_t>(64, (R + 255) / 256);
    dim3 grid((F + 63) / 64, (unsigned)((R + per - 1) / per));
    hipLaunchKernelGGL(col_stats_kernel<MODE>, grid, dim3(256), 0, st, x, g, R, F, per, mean, invstd, gamma, beta, site, s1, s2);
}
// mean / biased variance -> invstd; running statistics as nn.BatchNorm does (momentum 0.1, unbiased variance)
__global__ void bn_finalize_kernel(const double *s1, const double *s2, size_t R, int F, float eps, float momentum, float *mean, float *invstd,
                                   float *running_mean, float *running_var) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double m = s1[f] / (double)R;
    double var = s2[f] / (double)R - m * m;
    if (var < 0) var = 0;
    mean[f] = (float)m;
    invstd[f] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
        running_mean[f] = (1.f - momentum) * running_mean[f] + momentum * (float)m;
        running_var[f] = (1.f - momentum) * running_var[f] + momentum * (float)unb;
    }
}
// y = bn(x) [-> relu -> dropout]
// Elementwise BatchNorm kernels: the launch picks a grid whose stride (in elements) is a multiple of F, so a thread meets the same V
// features in every iteration and keeps their parameters in registers (bn_grid()).
template <int POST, int V>      // V = 4: F a multiple of 4 and 16-byte aligned buffers: four features of a row per thread
__global__ void bn_fwd_kernel(const float *__restrict__ x, size_t R, int F, const float *__restrict__ mean, const float *__restrict__ invstd,
                              const float *__restrict__ gamma, const float *__restrict__ beta, BnSite site, float *__restrict__ y) {
    const size_t n = R * F / V, q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = (int)((q0 * V) % F);
    float mu[V], is[V], ga[V], be[V];
#pragma unroll
    for (int j = 0; j < V; j++) { mu[j] = mean[f + j]; is[j] = invstd[f + j]; ga[j] = gamma[f + j]; be[j] = beta[f + j]; }
    for (size_t q = q0; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q * V;
        float xv[V], ov[V];
        unsigned char mv[V];
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(x + i); xv[0] = t.x; xv[1 % V] = t.y; xv[2 % V] = t.z; xv[3 % V] = t.w;
            if (POST && site.mask) { const uchar4 m = *reinterpret_cast<const uchar4 *>(site.mask + i); mv[0] = m.x; mv[1 % V] = m.y; mv[2 % V] = m.z; mv[3 % V] = m.w; }
        } else {
            xv[0] = x[i];
            if (POST && site.mask) mv[0] = site.mask[i];
        }
#pragma unroll
        for (int j = 0; j < V; j++) {
            float v = (xv[j] - mu[j]) * is[j] * ga[j] + be[j];
            if (POST) {
                const float m = site.mask ? (float)mv[j] * site.scale : 1.f;
                v = v > 0.f ? v * m : 0.f;
            }
            ov[j] = v;
        }
        if (V == 4) *reinterpret_cast<float4 *>(y + i) = make_float4(ov[0], ov[1 % V], ov[2 % V], ov[3 % V]);
        else y[i] = ov[0];
    }
}
// dx = gamma * invstd * (dy - sum(dy)/R - xhat * sum(dy*xhat)/R);  dgamma = sum(dy*xhat), dbeta = sum(dy)
template <int MODE, int V>
__global__ void bn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, size_t R, int F, const float *__restrict__ mean,
                              const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta, BnSite site,
                              const double *__restrict__ s1, const double *__restrict__ s2, float *__restrict__ dx) {
    const size_t n = R * F / V, q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = (int)((q0 * V) % F);
    float mu[V], is[V], ga[V], be[V], m1[V], m2[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        mu[j] = mean[f + j]; is[j] = invstd[f + j]; ga[j] = gamma[f + j]; be[j] = MODE == 2 ? beta[f + j] : 0.f;
        m1[j] = (float)(s1[f + j] / (double)R); m2[j] = (float)(s2[f + j] / (double)R);
    }
    for (size_t q = q0; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q * V;
        float xv[V], gv[V], ov[V];
        unsigned char mv[V];
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(x + i); xv[0] = t.x; xv[1 % V] = t.y; xv[2 % V] = t.z; xv[3 % V] = t.w;
            const float4 u = *reinterpret_cast<const float4 *>(g + i); gv[0] = u.x; gv[1 % V] = u.y; gv[2 % V] = u.z; gv[3 % V] = u.w;
            if (MODE == 2 && site.mask) { const uchar4 m = *reinterpret_cast<const uchar4 *>(site.mask + i); mv[0] = m.x; mv[1 % V] = m.y; mv[2 % V] = m.z; mv[3 % V] = m.w; }
        } else {
            xv[0] = x[i]; gv[0] = g[i];
            if (MODE == 2 && site.mask) mv[0] = site.mask[i];
        }
#pragma unroll
        for (int j = 0; j < V; j++) {
            float dy = gv[j];
            if (MODE == 2) {   // behind ReLU + dropout (bn_site_dy's arithmetic)
                const float yv = (xv[j] - mu[j]) * is[j] * ga[j] + be[j];
                const float m = site.mask ? (float)mv[j] * site.scale : 1.f;
                dy = yv > 0.f ? gv[j] * m : 0.f;
            }
            const float xh = (xv[j] - mu[j]) * is[j];
            ov[j] = ga[j] * is[j] * (dy - m1[j] - xh * m2[j]);
        }
        if (V == 4) *reinterpret_cast<float4 *>(dx + i) = make_float4(ov[0], ov[1 % V], ov[2 % V], ov[3 % V]);
        else dx[i] = ov[0];
    }
}
// workgroups of 256 threads such that the grid's stride in elements (grid * 256 * V) is a multiple of F
static unsigned bn_grid(int F, int V) {
    int a = F, b = 256 * V;
    while (b) { const int t_ = a % b; a = b; b = t_; }              // a = gcd(F, 256 * V)
    const int unit = F / a;                                         // the grid must be a multiple of this
    return (unsigned)std::max(unit, 4096 / unit * unit);
}
__global__ void bn_param_grads_kernel(const double *s1, const double *s2, int F, float *dgamma, float *dbeta) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) { dgamma[f] = (float)s2[f]; dbeta[f] = (float)s1[f]; }
}

int launch_bn_train_fwd(const float *x, size_t R, int F, const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                        float *running_var, double *s1s2, float *mean, float *invstd, const BnSite *site, float *y, hipStream_t st) {
    MDD_HIP_CHECK(hipMemsetAsync(s1s2, 0, sizeof(double) * 2 * F, st));
    BnSite none{nullptr, 1.f};
    launch_col_stats<0>(x, nullptr, R, F, nullptr, nullptr, nullptr, nullptr, none, s1s2, s1s2 + F, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((F + 127) / 128), dim3(128), 0, st, s1s2, s1s2 + F, R, F, eps, momentum, mean, invstd, running_mean, running_var);
    const bool v4 = F % 4 == 0 && (((size_t)x | (size_t)y) & 15) == 0 && (!site || !site->mask || ((size_t)site->mask & 3) == 0);
    if (site && v4) hipLaunchKernelGGL((bn_fwd_kernel<1, 4>), dim3(bn_grid(F, 4)), dim3(256), 0, st, x, R, F, mean, invstd, gamma, beta, *site, y);
    else if (site) hipLaunchKernelGGL((bn_fwd_kernel<1, 1>), dim3(bn_grid(F, 1)), dim3(256), 0, st, x, R, F, mean, invstd, gamma, beta, *site, y);
    else if (v4) hipLaunchKernelGGL((bn_fwd_kernel<0, 4>), dim3(bn_grid(F, 4)), dim3(256), 0, st, x, R, F, mean, invstd, gamma, beta, none, y);
    else hipLaunchKernelGGL((bn_fwd_kernel<0, 1>), dim3(bn_grid(F, 1)), dim3(256), 0, st, x, R, F, mean, invstd, gamma, beta, none, y);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
int launch_bn_train_bwd(const float *x, const float *g, size_t R, int F, const float *gamma, const float *beta, const float *mean, const float *invstd,
                        const BnSite *site, double *s1s2, float *dx, float *dgamma, float *dbeta, hipStream_t st) {
    MDD_HIP_CHECK(hipMemsetAsync(s1s2, 0, sizeof(double) * 2 * F, st));
    BnSite none{nullptr, 1.f};
    const bool v4 = F % 4 == 0 && (((size_t)x | (size_t)g | (size_t)dx) & 15) == 0 && (!site || !site->mask || ((size_t)site->mask & 3) == 0);
    if (site) {
        launch_col_stats<2>(x, g, R, F, mean, invstd, gamma, beta, *site, s1s2, s1s2 + F, st);
        if (v4) hipLaunchKernelGGL((bn_bwd_kernel<2, 4>), dim3(bn_grid(F, 4)), dim3(256), 0, st, x, g, R, F, mean, invstd, gamma, beta, *site, s1s2, s1s2 + F, dx);
        else hipLaunchKernelGGL((bn_bwd_kernel<2, 1>), dim3(bn_grid(F, 1)), dim3(256), 0, st, x, g, R, F, mean, invstd, gamma, beta, *site, s1s2, s1s2 + F, dx);
    } else {
        launch_col_stats<1>(x, g, R, F, mean, invstd, gamma, beta, none, s1s2, s1s2 + F, st);
        if (v4) hipLaunchKernelGGL((bn_bwd_kernel<1, 4>), dim3(bn_grid(F, 4)), dim3(256), 0, st, x, g, R, F, mean, invstd, gamma, beta, none, s1s2, s1s2 + F, dx);
        else hipLaunchKernelGGL((bn_bwd_kernel<1, 1>), dim3(bn_grid(F, 1)), dim3(256), 0, st, x, g, R, F, mean, invstd, gamma, beta, none, s1s2, s1s2 + F, dx);
    }
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((F + 127) / 128), dim3(128), 0, st, s1s2, s1s2 + F, F, dgamma, dbeta);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
__global__ void col_sum_finish_kernel(const double *s1, int F, float *out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) out[f] = (float)s1[f];
}
// column sums of g [R, F] -> out[F] (bias gradients)
int launch_col_sum(const float *g, size_t R, int F, double *s1s2, float *out, hipStream_t st) {
    MDD_HIP_CHECK(hipMemsetAsync(s1s2, 0, sizeof(double) * 2 * F, st));
    BnSite none{nullptr, 1.f};
    launch_col_stats<0>(g, nullptr, R, F, nullptr, nullptr, nullptr, nullptr, none, s1s2, s1s2 + F, st);
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3((F + 127) / 128), dim3(128), 0, st, s1s2, F, out);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise helpers
// y = x * mask * scale (mask in the same [rows, F] order); in place allowed
__global__ void dropout_rows_kernel(const float *__restrict__ x, const unsigned char *__restrict__ mask, float scale, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = mask ? x[i] * ((float)mask[i] * scale) : x[i];
}
int launch_dropout_rows(const float *x, const unsigned char *mask, float scale, size_t n, float *y, hipStream_t st) {
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(4096), dim3(256), 0, st, x, mask, scale, n, y);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// dst[r, c0 + j] (ld_dst) (+)= src[r, s0 + j] (ld_src), j < width
__global__ void copy_cols_kernel(const float *__restrict__ src, int ld_src, int s0, float *__restrict__ dst, int ld_dst, int c0, size_t R, int width, int add) {
    const size_t n = R * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / width; const int j = (int)(i % width);
        const float v = src[r * ld_src + s0 + j];
        float *d = dst + r * ld_dst + c0 + j;
        *d = add ? *d + v : v;
    }
}
int launch_copy_cols(const float *src, int ld_src, int s0, float *dst, int ld_dst, int c0, size_t R, int width, bool add, hipStream_t st) {
    hipLaunchKernelGGL(copy_cols_kernel, dim3(4096), dim3(256), 0, st, src, ld_src, s0, dst, ld_dst, c0, R, width, add ? 1 : 0);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// rows of length n: y = softmax(x) or log_softmax(x) (one wave per row)
template <int LOG>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ x, size_t R, int n, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *xr = x + r * n;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, xr[j]);
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += expf(xr[j] - m);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    const float ls = logf(s);
    for (int j = lane; j < n; j += 64) y[r * n + j] = LOG ? (xr[j] - m) - ls : expf(xr[j] - m) / s;
}
int launch_softmax_rows(const float *x, size_t R, int n, float *y, bool log, hipStream_t st) {
    dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (log) hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, block, 0, st, x, R, n, y);
    else hipLaunchKernelGGL(softmax_rows_kernel<0>, grid, block, 0, st, x, R, n, y);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// LOG: dx = g - exp(y) * sum(g)   (y = log-probs);   else: dx = y * (g - sum(g * y))   (y = probabilities)
template <int LOG>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float *__restrict__ y, const float *__restrict__ g, size_t R, int n, float *__restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += LOG ? g[r * n + j] : g[r * n + j] * y[r * n + j];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    for (int j = lane; j < n; j += 64) {
        const float yv = y[r * n + j], gv = g[r * n + j];
        dx[r * n + j] = LOG ? gv - expf(yv) * s : yv * (gv - s);
    }
}
int launch_softmax_bwd_rows(const float *y, const float *g, size_t R, int n, float *dx, bool log, hipStream_t st) {
    dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (log) hipLaunchKernelGGL(softmax_bwd_rows_kernel<1>, grid, block, 0, st, y, g, R, n, dx);
    else hipLaunchKernelGGL(softmax_bwd_rows_kernel<0>, grid, block, 0, st, y, g, R, n, dx);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// dE[v, :] = sum over positions m = l*B + b with ids[b, l] == v of g[m, :]   (ascending m: deterministic)
__global__ __launch_bounds__(128) void embed_bwd_kernel(const float *__restrict__ g, const int64_t *__restrict__ ids, int B, int L, int E, float *__restrict__ dE) {
    __shared__ int sid[1024];                                       // the ids of 1024 positions at a time: only the matching ones touch g
    const int v = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x, LB = L * B;
    float acc = 0.f;
    for (int m0 = 0; m0 < LB; m0 += 1024) {
        const int nm = min(1024, LB - m0);
        for (int j = threadIdx.x; j < nm; j += blockDim.x) { const int m = m0 + j, l = m / B, b = m - l * B; sid[j] = (int)ids[(size_t)b * L + l]; }
        __syncthreads();
        if (e < E)
            for (int j = 0; j < nm; j++)
                if (sid[j] == v) acc += g[(size_t)(m0 + j) * E + e];
        __syncthreads();
    }
    if (e < E) dE[(size_t)v * E + e] = acc;
}
int launch_embed_bwd(const float *g, const int64_t *ids, int B, int L, int E, int rows, float *dE, hipStream_t st) {
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(rows, (E + 127) / 128), dim3(128), 0, st, g, ids, B, L, E, dE);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// sum of `parts` partial matrices of n elements (split-K GEMM outputs): out[i] = sum_z part[z*n + i]
__global__ void reduce_parts_kernel(const float *__restrict__ part, int parts, size_t n, float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int z = 0; z < parts; z++) acc += (double)part[(size_t)z * n + i];
        out[i] = (float)acc;
    }
}
int launch_reduce_parts(const float *part, int parts, size_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, part, parts, n, out);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ LSTM weight layouts
// reference W [4H, K] per direction (gate-major rows g*H + u)  <->  packed W' [2][4H][K] with rows u*4 + g
__global__ void pack_gates_kernel(const float *__restrict__ w_fwd, const float *__restrict__ w_rev, float *__restrict__ packed, int H, int K, int to_packed,
                                  float *__restrict__ out_fwd, float *__restrict__ out_rev) {
    const size_t n = (size_t)2 * 4 * H * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const size_t row = i / K;                      // d*4H + u*4 + g
        const int d = (int)(row / (4 * H)), rr = (int)(row % (4 * H)), u = rr >> 2, g = rr & 3;
        const size_t j = ((size_t)g * H + u) * K + k;  // index inside the direction's reference tensor
        if (to_packed) packed[i] = (d ? w_rev : w_fwd)[j];
        else (d ? out_rev : out_fwd)[j] = packed[i];
    }
}
int launch_pack_gates(const float *w_fwd, const float *w_rev, float *packed, int H, int K, hipStream_t st) {
    hipLaunchKernelGGL(pack_gates_kernel, dim3(2048), dim3(256), 0, st, w_fwd, w_rev, packed, H, K, 1, (float *)nullptr, (float *)nullptr);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
int launch_unpack_gates(const float *packed, float *out_fwd, float *out_rev, int H, int K, hipStream_t st) {
    hipLaunchKernelGGL(pack_gates_kernel, dim3(2048), dim3(256), 0, st, (const float *)nullptr, (const float *)nullptr, const_cast<float *>(packed), H, K, 0, out_fwd, out_rev);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
// Whh' [2][4H][H] -> WhhT [2][H][4H]  (the backward step reads Whh' down its columns)
__global__ void transpose_whh_kernel(const float *__restrict__ w, float *__restrict__ wt, int H) {
    const size_t n = (size_t)2 * 4 * H * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % H);
        const size_t row = i / H;
        const int d = (int)(row / (4 * H)), nrow = (int)(row % (4 * H));
        wt[((size_t)d * H + k) * 4 * H + nrow] = w[i];
    }
}
int launch_transpose_whh(const float *w, float *wt, int H, hipStream_t st) {
    hipLaunchKernelGGL(transpose_whh_kernel, dim3(2048), dim3(256), 0, st, w, wt, H);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ LSTM backward (BPTT), one launch per step
// Backward order s = 0..T-1 visits time t_s = T-1-s (forward direction) / s (reverse direction).  Launch s, for its 16 hidden
// units k and 16 batch rows b:
//   dh[b,k]   = dout[t_s, b, d, k] + sum_n DG[t_prev][b, d, n] * Whh'[d][n][k]         (t_prev = the time visited by launch s-1)
//   then the cell backward of (t_s, b, k): with the saved gates i,f,g,o, c_t and c_{t-1},
//     do = dh * tanh(c_t) * o(1-o);   dc = dh * o * (1 - tanh(c_t)^2) + dc_carry
//     di = dc * g * i(1-i);  df = dc * c_{t-1} * f(1-f);  dg = dc * i * (1-g^2);  dc_carry = dc * f
//   DG[t_s][b, d, k*4 + {i,f,g,o}] = pre-activation gradients (what the weight-gradient GEMMs and the next launch read).
// The contraction over n = 4H is split over the four waves of the workgroup (one quarter each, v_mfma_f32_16x16x4_f32 with the
// units on the MFMA row axis) and combined through LDS; the launch boundary is the step-to-step dependency.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// NQ4 > 0: H is a compile-time multiple of 16 and a lane's share of the contraction is NQ4 float4 steps (24 at H = 384, 16 at
// H = 256): ALL of its loads are issued before the first MFMA, so a step pays one L2 round trip instead of one per unrolled group.
template <int NQ4>
__global__ __launch_bounds__(256, 1) void lstm_bwd_step_kernel(LstmBwdArgs a, int s) {
    __shared__ float red[4][16 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kt = blockIdx.x, d = blockIdx.y, b0 = blockIdx.z * 16;
    const int H = a.H, B = a.B, T = a.T, G = 4 * H;
    const int t = d ? s : (T - 1 - s);
    const int tprev = d ? s - 1 : (T - s);                 // time visited by the previous launch
    const int li = lane & 15, kq = lane >> 4;
    // the cell backward's operands (saved gates, cell states, the carried cell gradient, the upstream gradient) do not depend on the
    // product: every thread requests those of its (unit, batch) element before the contraction, one memory round trip per step
    const int ul = tid >> 4, bl = tid & 15;
    const int u = kt * 16 + ul, b = b0 + bl;
    const bool live = u < H && b < B;
    const int uc = min(u, H - 1), bcl = min(b, B - 1);
    const size_t si = (((size_t)t * B + bcl) * 2 + d) * H + uc;
    const float4 gt = *reinterpret_cast<const float4 *>(a.gates + si * 4);
    const float ct = a.cst[si];
    const int tp = d ? t + 1 : t - 1;                       // the forward pass's previous time of this direction
    const float cprev = (tp >= 0 && tp < T) ? a.cst[(((size_t)tp * B + bcl) * 2 + d) * H + uc] : 0.f;
    const size_t ci = ((size_t)d * B + bcl) * H + uc;
    const float dc_in = s > 0 ? a.dc[ci] : 0.f;
    const float dout_v = a.dout[((size_t)t * B + bcl) * 2 * H + d * H + uc];
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        // wave w contracts n in [w*G/4, (w+1)*G/4); inside the wave the four 16-lane groups take contiguous quarters of that range
        const int nq = G / 16, n0 = wave * (G / 4) + kq * nq;
        const int bb = min(b0 + li, B - 1);
        const float *wp = a.whhT + ((size_t)d * H + min(kt * 16 + li, H - 1)) * G + n0;               // A[row = unit li][n]
        const float *gp = a.dg + (((size_t)tprev * B + bb) * 2 + d) * G + n0;                        // B[n][col = batch li]
        if (NQ4 > 0) {
            float4 w4[NQ4 > 0 ? NQ4 : 1], g4[NQ4 > 0 ? NQ4 : 1];
#pragma unroll
            for (int n = 0; n < NQ4; n++) { w4[n] = *reinterpret_cast<const float4 *>(wp + 4 * n); g4[n] = *reinterpret_cast<const float4 *>(gp + 4 * n); }
            __builtin_amdgcn_sched_barrier(0);
            f32x4v a1 = acc, a2 = acc, a3 = acc;           // four chains: the dependent-issue latency of the fp32 MFMA exceeds its issue time
#pragma unroll
            for (int n = 0; n < NQ4; n++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].x, g4[n].x, acc, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].y, g4[n].y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].z, g4[n].z, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].w, g4[n].w, a3, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = (acc[r] + a1[r]) + (a2[r] + a3[r]);
        } else {
            for (int n = 0; n < nq; n++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[n], gp[n], acc, 0, 0, 0);
        }
    }
    // D layout: col = lane&15 (batch), row = 4*(lane>>4) + r (unit within the tile)
#pragma unroll
    for (int r = 0; r < 4; r++) red[wave][(4 * kq + r) * 16 + li] = acc[r];
    __syncthreads();
    // one thread per (unit, batch) of the tile
    if (!live) return;
    const float dh_rec = (red[0][ul * 16 + bl] + red[1][ul * 16 + bl]) + (red[2][ul * 16 + bl] + red[3][ul * 16 + bl]);
    const float dh = dout_v + dh_rec;
    const float th = tanhf(ct);
    const float d_o = dh * th * gt.w * (1.f - gt.w);
    const float dcell = dh * gt.w * (1.f - th * th) + dc_in;
    const float d_i = dcell * gt.z * gt.x * (1.f - gt.x);
    const float d_f = dcell * cprev * gt.y * (1.f - gt.y);
    const float d_g = dcell * gt.x * (1.f - gt.z * gt.z);
    a.dc[ci] = dcell * gt.y;
    *reinterpret_cast<float4 *>(a.dg + (((size_t)t * B + b) * 2 + d) * G + u * 4) = make_float4(d_i, d_f, d_g, d_o);
}
int launch_lstm_bwd(const LstmBwdArgs &a, hipStream_t st) {
    if (a.H % 16 && a.H % 4) { set_error("lstm backward: H must be a multiple of 4"); return MDD_ERR_ARG; }
    dim3 grid((a.H + 15) / 16, 2, (a.B + 15) / 16), block(256);
    for (int s = 0; s < a.T; s++) {
        if (a.H == 384) hipLaunchKernelGGL(lstm_bwd_step_kernel<24>, grid, block, 0, st, a, s);
        else if (a.H == 256) hipLaunchKernelGGL(lstm_bwd_step_kernel<16>, grid, block, 0, st, a, s);
        else hipLaunchKernelGGL(lstm_bwd_step_kernel<0>, grid, block, 0, st, a, s);
    }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ split-bf16 operands for the training GEMMs
// The flagged split-bf16 variant of the training step sends its large contractions through the projection GEMM of the decode path
// (gemm_bf16x3.hip: C = A . W^T with hi/lo bf16 planes, K a multiple of 32).  These two kernels make the planes of an fp32 matrix:
// as it stands (rows x cols, the contraction along the columns, zero-padded to cols_pad) or transposed (the contraction along the ROWS:
// out[c][r] = src[r][c], zero-padded to rows_pad), which turns the NN and TN products of the backward pass into the NT form.
__device__ __forceinline__ void split_store(float v, unsigned short *hi, unsigned short *lo, size_t i) {
    __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
    hi[i] = *reinterpret_cast<unsigned short *>(&h);
    lo[i] = *reinterpret_cast<unsigned short *>(&l);
}
__global__ void split_rows_kernel(const float *__restrict__ src, int ld, size_t rows, int cols, int cols_pad, unsigned short *__restrict__ hi,
                                  unsigned short *__restrict__ lo) {
    const size_t n = rows * cols_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols_pad; const int c = (int)(i % cols_pad);
        split_store(c < cols ? src[r * ld + c] : 0.f, hi, lo, i);
    }
}
__global__ __launch_bounds__(256) void transpose_split_kernel(const float *__restrict__ src, int ld, int rows, int cols, int rows_pad,
                                                              unsigned short *__restrict__ hi, unsigned short *__restrict__ lo) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows_pad) split_store(tile[tx][j], hi, lo, (size_t)c * rows_pad + r);
    }
}
int launch_split_rows(const float *src, int ld, size_t rows, int cols, int cols_pad, unsigned short *hi, unsigned short *lo, hipStream_t st) {
    hipLaunchKernelGGL(split_rows_kernel, dim3(4096), dim3(256), 0, st, src, ld, rows, cols, cols_pad, hi, lo);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}
int launch_transpose_split(const float *src, int ld, int rows, int cols, int rows_pad, unsigned short *hi, unsigned short *lo, hipStream_t st) {
    hipLaunchKernelGGL(transpose_split_kernel, dim3((rows_pad + 31) / 32, (cols + 31) / 32), dim3(256), 0, st, src, ld, rows, cols, rows_pad, hi, lo);
    MDD_LAUNCH_CHECK(); return MDD_OK;
}

// ------------------------------------------------------------------------------------------------ Adam (torch.optim.Adam semantics, L2 weight decay)
// g' = g + wd * p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, size_t n, float lr, float b1,
                            float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] + wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}
int launch_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// The same update over up to 48 tensors in ONE launch (the model has 55 parameter tensors, most of them small: one launch each was a
// quarter millisecond of launches per step).  The table travels in the kernel arguments; workgroup w works on 1024-element chunk
// w - first[t] of tensor t (first[] = running chunk counts).
constexpr int ADAM_MT = 48;
struct AdamTable { float *p[ADAM_MT]; const float *g[ADAM_MT]; float *m[ADAM_MT]; float *v[ADAM_MT]; unsigned long long n[ADAM_MT]; int first[ADAM_MT + 1]; int count; };
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable tb, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    int t = 0;
    while (t + 1 < tb.count && (int)blockIdx.x >= tb.first[t + 1]) t++;            // (uniform per workgroup; <= 48 steps)
    const size_t i0 = (size_t)((int)blockIdx.x - tb.first[t]) * 1024;
    float *p = tb.p[t], *m = tb.m[t], *v = tb.v[t];
    const float *g = tb.g[t];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t i = i0 + threadIdx.x + 256 * k;
        if (i < tb.n[t]) {
            const float gi = g[i] + wd * p[i];
            const float mi = b1 * m[i] + (1.f - b1) * gi;
            const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            m[i] = mi; v[i] = vi;
            p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
    }
}
int launch_adam_multi(float *const *p, float *const *g, float *const *m, float *const *v, const int64_t *numel, int n, float lr, float b1, float b2, float eps,
                      float wd, int step, hipStream_t st) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
    int i = 0;
    while (i < n) {
        AdamTable tb;
        tb.count = 0; tb.first[0] = 0;
        for (; i < n && tb.count < ADAM_MT; i++) {
            if (!p[i] || !g[i] || numel[i] <= 0) continue;
            const int c = tb.count++;
            tb.p[c] = p[i]; tb.g[c] = g[i]; tb.m[c] = m[i]; tb.v[c] = v[i]; tb.n[c] = (unsigned long long)numel[i];
            tb.first[c + 1] = tb.first[c] + (int)((numel[i] + 1023) / 1024);
        }
        if (tb.count == 0) break;
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)tb.first[tb.count]), dim3(256), 0, st, tb, lr, b1, b2, eps, wd, bc1, bc2s);
        MDD_LAUNCH_CHECK();
    }
    return MDD_OK;
}

}  // namespace mdd
