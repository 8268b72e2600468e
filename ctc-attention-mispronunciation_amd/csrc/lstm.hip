// BiLSTM recurrence: one launch per time step, both directions in the same grid.
//
// Reference: torch.nn.LSTM as used by BatchRNN.forward (AA/models/model_ctc.py:36-49) and by the text
// encoder (:150,198): gate order i,f,g,o, zero initial state, every padded step is processed, the
// reverse direction starts at t = T-1.
//
// Per step and direction:  G^T[n,b] = sum_k Whh[n,k] h_prev[b,k]   (n = 4H gate rows, b = batch)
// is computed with v_mfma_f32_16x16x4_f32 with the GATE rows on the MFMA row axis and the batch on
// the column axis.  Gate rows are stored permuted (n' = unit*4 + gate) so that the four accumulator
// registers of a lane are exactly i,f,g,o of one (unit, batch) pair: the cell update needs no
// cross-lane traffic and no LDS.  A wave owns 4 hidden units x 16 batch rows; a workgroup = 4 waves
// = 4 batch tiles of the same 4 units; grid = (H/4 unit tiles, 2 directions, ceil(B/64)).
// The k axis is split over the four 16-lane groups of the wave in contiguous quarters (the MFMA
// only requires A and B to agree on k), so every lane streams H/4 contiguous floats of one Whh row
// and of one h_prev row.
//
// The launch boundary is the step-to-step dependency (all-to-all over hidden units): per the
// MI355X price list a dependent kernel boundary (~1.5 us) is cheaper than any in-launch grid sync.
#include "mdd_internal.h"

namespace mdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

template <bool VEC>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a, int s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = blockIdx.x, d = blockIdx.y;
    const int b0 = (blockIdx.z * 4 + wave) * 16;
    if (b0 >= a.B) return;  // wave-uniform
    const int H = a.H, B = a.B, KQ = H >> 2;
    const int t = d ? (a.T - 1 - s) : s;
    const int li = lane & 15, kq = lane >> 4;
    const int b = b0 + li, bc = b < B ? b : B - 1;
    const float *hprev = a.hbuf + ((size_t)((s & 1) ^ 1) * 2 + d) * B * H;
    float *hnext = a.hbuf + ((size_t)(s & 1) * 2 + d) * B * H;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const float *wp = a.whh + ((size_t)d * 4 * H + ut * 16 + li) * H + kq * KQ;  // A[row li][k quarter kq]
        const float *hp = hprev + (size_t)bc * H + kq * KQ;                            // B[k quarter kq][col li]
        if (VEC) {
#pragma unroll 8
            for (int k = 0; k < KQ; k += 4) {
                float4 w4 = *reinterpret_cast<const float4 *>(wp + k);
                float4 h4 = *reinterpret_cast<const float4 *>(hp + k);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, h4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, h4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, h4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, h4.w, acc, 0, 0, 0);
            }
        } else {
            for (int k = 0; k < KQ; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[k], hp[k], acc, 0, 0, 0);
        }
    }
    // D layout: col = lane&15 (batch), row = 4*(lane>>4) + r  ->  unit = ut*4 + (lane>>4), gate = r
    if (b < B) {
        const int u = ut * 4 + kq;
        const float4 g4 = *reinterpret_cast<const float4 *>(a.gx + (((size_t)t * B + b) * 2 + d) * 4 * H + u * 4);
        const size_t ci = ((size_t)d * B + b) * H + u;
        const float cold = s > 0 ? a.cbuf[ci] : 0.f;
        const float ig = sigmoid_f(acc[0] + g4.x), fg = sigmoid_f(acc[1] + g4.y);
        const float gg = tanhf(acc[2] + g4.z), og = sigmoid_f(acc[3] + g4.w);
        const float cn = fg * cold + ig * gg;
        const float hn = og * tanhf(cn);
        a.cbuf[ci] = cn;
        hnext[(size_t)b * H + u] = hn;
        const size_t oi = ((size_t)t * B + b) * 2 * H + d * H + u;
        if (a.out_raw) a.out_raw[oi] = hn;
        if (a.out != a.out_raw) a.out[oi] = a.oscale ? hn * a.oscale[d * H + u] + a.oshift[d * H + u] : hn;
    }
}

int launch_lstm_layer(const LstmStepArgs &a, hipStream_t st) {
    if (a.H % 4 != 0 || a.T <= 0 || a.B <= 0) { set_error("lstm: bad shape T=%d B=%d H=%d", a.T, a.B, a.H); return MDD_ERR_ARG; }
    dim3 grid(a.H / 4, 2, (a.B + 63) / 64), block(256);
    const bool vec = (a.H % 16 == 0);
    for (int s = 0; s < a.T; s++) {
        if (vec) hipLaunchKernelGGL(lstm_step_kernel<true>, grid, block, 0, st, a, s);
        else hipLaunchKernelGGL(lstm_step_kernel<false>, grid, block, 0, st, a, s);
    }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
