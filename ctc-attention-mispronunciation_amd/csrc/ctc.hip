// CTC forward-backward (alpha/beta lattice) for the training step.
//
// Reference: nn.CTCLoss(reduction='sum') as called at AA/steps/train_ctc.py:72,186 (blank = 0, padded 2-D
// targets, input lengths = (frac * T').long()).  The arithmetic is ATen's; restated from the published
// algorithm (Graves et al. 2006, eq. 6-8, 10-11, 16).  Outputs: per-utterance nll (the reference's loss
// is their sum) and the tensor autograd deposits on the log-probs,
//   grad[t,b,c] = exp(lp) - sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - ll - lp),  t < in_len[b]
// and 0 on padded frames.
//
// ctc_wave_kernel -- one workgroup per utterance, the lattice on wavefronts:
//   * every lane of a wave owns NL consecutive labels of the utterance and, with each, the blank state in front of it
//     (states 2i and 2i+1): a time step needs ONE value from the neighbouring lane (alpha: lane-1's last label state;
//     beta: lane+1's first blank and label state), moved by a DPP wave shift -- no LDS, no barrier inside the scan;
//   * the alpha recurrence (forwards, wave 0) and the beta recurrence (backwards, wave 1) are independent, so they run
//     side by side: the dependent chain is T' steps, not 2 T';
//   * the utterance's log-probs are staged once in LDS ([T'][C] fp32, 45 KB at T'=250, C=45) by all waves; a step reads
//     its two values from there one step ahead of their use;
//   * lattice values are fp64; a log-add is  m + log(sum exp(x - m))  with the differences and the exp/log in fp32 on
//     the hardware transcendental units (v_exp_f32 / v_log_f32, 1 ulp): the terms are <= 1 and the largest is exactly
//     1, so a step's absolute error is ~1e-7 whatever the magnitude of the values (an fp32 lattice, as ATen keeps it,
//     loses 6e-8 RELATIVE to values of several hundred per step);
//   * both rows of every frame go to a workspace (fp64, 16 bytes per lane-label, coalesced); after one barrier all
//     eight waves turn frames into gradient rows: occupancies gamma_t(s) = exp(alpha + beta - ll - lp) are <= 1, so the
//     per-class sums are plain sums -- the blank states by a DPP wave reduction, the label states by a per-class list
//     walk in ascending label position (deterministic: the same bits on every run, no atomics).
// The scan is latency-bound (T' dependent steps per utterance, ~0.1-0.2 us each), not HBM-bound: the bytes it must move
// (log-probs in, gradient out) are 2 x T' x B x C x 4.
//
// ctc_generic_kernel -- the previous form (one thread per state, LDS rows, one barrier per frame): any label length and
// class count, used when an utterance does not fit the wave form (Lmax > 255, C > 256 or T' x C x 4 > 128 KB).
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "mdd_internal.h"

namespace mdd {

__device__ __forceinline__ double lse3(double a, double b, double c) {
    double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}
__device__ __forceinline__ double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = fmax(a, b);
    return m + log(exp(a - m) + exp(b - m));
}

// ---- fp64 value, fp32 increment log-adds (see the header comment)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309417f; }
__device__ __forceinline__ double wlse2(double a, double b) {
    const double m = fmax(a, b);
    const double mm = (m == -INFINITY) ? 0.0 : m;
    const float s = fexp((float)(a - mm)) + fexp((float)(b - mm));
    return mm + (double)flog(s);
}
__device__ __forceinline__ double wlse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    const double mm = (m == -INFINITY) ? 0.0 : m;
    const float s = fexp((float)(a - mm)) + fexp((float)(b - mm)) + fexp((float)(c - mm));
    return mm + (double)flog(s);
}

// lane i <- lane i-1 (SHR) / lane i+1 (SHL) across the whole wave; the edge lane gets -inf
template <int CTRL>
__device__ __forceinline__ double wave_shift(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp((int)0xfff00000, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi2, lo2);
}
static constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;

// LDS traffic of ONE wave is processed in order; this only keeps the compiler from moving accesses across the point
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float x) {   // every lane returns the sum over the 64 lanes
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
}

struct CtcArgs {
    const float *logp; int T, B, C;
    const int64_t *targets; int Lmax;
    const int64_t *in_len, *tgt_len;
    int blank;
    float *nll, *grad;
    double *ws;      // [B][2][T][SP] alpha rows then beta rows (grad != null only)
    int SP;          // row pitch in doubles = 2 * (Lmax + 1)
};

// dynamic LDS: lpt[Tb][C] f32 | lab[LC] i32 | cls_off[C+1] i32 | cls_idx[LC] i32 | gam[8][LC] f32   (LC = 64 * NL)
template <int NL>
__global__ __launch_bounds__(512) void ctc_wave_kernel(CtcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int LC = 64 * NL;
    const int T = a.T, B = a.B, C = a.C, blank = a.blank;
    float *lpt = reinterpret_cast<float *>(sm);
    int *lab_s = reinterpret_cast<int *>(lpt + (size_t)T * C);
    int *cls_off = lab_s + LC;
    int *cls_idx = cls_off + C + 1;
    float *gam = reinterpret_cast<float *>(cls_idx + LC);
    __shared__ double s_fin[2];
    __shared__ int s_bad;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwave = blockDim.x >> 6;
    int Tb = (int)a.in_len[b], L = (int)a.tgt_len[b];
    if (Tb > T) Tb = T;
    if (Tb < 0) Tb = 0;
    if (L > a.Lmax) L = a.Lmax;
    if (L < 0) L = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // labels (a label outside [0, C) would index past the row: the utterance is reported as NaN instead)
    for (int i = tid; i < LC; i += blockDim.x) {
        int l = blank;
        if (i < L) {
            const long long v = a.targets[(size_t)b * a.Lmax + i];
            if (v < 0 || v >= C) { s_bad = 1; } else l = (int)v;
        }
        lab_s[i] = l;
    }
    // stage the utterance's log-probs: rows of C floats, B*C apart
    for (int e = tid; e < Tb * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        lpt[e] = a.logp[((size_t)t * B + b) * C + c];
    }
    if (a.grad)   // padded frames carry zero gradient
        for (int e = Tb * C + tid; e < T * C; e += blockDim.x) {
            const int t = e / C, c = e - t * C;
            a.grad[((size_t)t * B + b) * C + c] = 0.f;
        }
    __syncthreads();
    if (s_bad || Tb == 0) {
        if (tid == 0) a.nll[b] = s_bad ? NAN : ((L == 0) ? 0.f : INFINITY);
        if (a.grad && s_bad)
            for (int e = tid; e < Tb * C; e += blockDim.x) { const int t = e / C, c = e - t * C; a.grad[((size_t)t * B + b) * C + c] = 0.f; }
        return;
    }
    // per-class lists of label positions (ascending), for the gradient's deterministic class sums
    if (a.grad && wave == 2) {
        for (int c = lane; c < C; c += 64) {
            int n = 0;
            for (int i = 0; i < L; i++) n += lab_s[i] == c;
            cls_off[c + 1] = n;
        }
        if (lane == 0) cls_off[0] = 0;
        wave_lds_order();
        if (lane == 0) for (int c = 0; c < C; c++) cls_off[c + 1] += cls_off[c];
        wave_lds_order();
        for (int c = lane; c < C; c += 64) {
            int k = cls_off[c];
            for (int i = 0; i < L; i++) if (lab_s[i] == c) cls_idx[k++] = i;
        }
    }
    double *wsa = a.ws ? a.ws + (size_t)b * 2 * T * a.SP : nullptr;
    double *wsb = wsa ? wsa + (size_t)T * a.SP : nullptr;
    int lab[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) lab[j] = lab_s[lane * NL + j];

    if (wave == 0) {
        // ---------------- alpha, forwards
        bool skip[NL], valid_o[NL];
        const int left_lab = __shfl_up(lab[NL - 1], 1);
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            const int prev = j == 0 ? left_lab : lab[j - 1];
            valid_o[j] = i < L;
            skip[j] = i >= 1 && i < L && lab[j] != prev && lab[j] != blank;
        }
        double ae[NL], ao[NL];
        float lpb = lpt[blank], lpl[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) lpl[j] = lpt[lab[j]];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            ae[j] = i == 0 ? (double)lpb : -INFINITY;
            ao[j] = (i == 0 && L > 0) ? (double)lpl[j] : -INFINITY;
        }
        // lpb / lpl hold the log-probs of row t+1 while row t+1 is being computed; row t+2's are requested first, so their
        // LDS latency lies beside the step's arithmetic, not in front of it
        float nb = 0.f, nl[NL];
        if (Tb > 1) {
            const float *row = lpt + C;
            nb = row[blank];
#pragma unroll
            for (int j = 0; j < NL; j++) nl[j] = row[lab[j]];
        }
        for (int t = 0;; t++) {
            if (wsa) {
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    const int i = lane * NL + j;
                    if (i <= L) *reinterpret_cast<double2 *>(wsa + (size_t)t * a.SP + 2 * i) = make_double2(ae[j], ao[j]);
                }
            }
            if (t + 1 >= Tb) break;
            lpb = nb;
#pragma unroll
            for (int j = 0; j < NL; j++) lpl[j] = nl[j];
            if (t + 2 < Tb) {
                const float *row = lpt + (size_t)(t + 2) * C;
                nb = row[blank];
#pragma unroll
                for (int j = 0; j < NL; j++) nl[j] = row[lab[j]];
            }
            const double from_left = wave_shift<DPP_WAVE_SHR1>(ao[NL - 1]);
            double ne[NL], no[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const double pol = j == 0 ? from_left : ao[j - 1];
                ne[j] = wlse2(ae[j], pol) + (double)lpb;
                no[j] = valid_o[j] ? wlse3(ao[j], ae[j], skip[j] ? pol : -INFINITY) + (double)lpl[j] : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < NL; j++) { ae[j] = ne[j]; ao[j] = no[j]; }
        }
        // log-likelihood: states 2L (blank behind the last label) and 2L-1 (the last label)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            if (i == L) s_fin[0] = ae[j];
            if (i == L - 1) s_fin[1] = ao[j];
        }
        if (L == 0 && lane == 0) s_fin[1] = -INFINITY;
    } else if (wave == 1 && a.grad) {
        // ---------------- beta, backwards
        bool skip[NL];
        const int right_lab = __shfl_down(lab[0], 1);
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            const int next = j == NL - 1 ? right_lab : lab[j + 1];
            skip[j] = i + 1 < L && lab[j] != blank && lab[j] != next;
        }
        double be[NL], bo[NL];
        const float *row0 = lpt + (size_t)(Tb - 1) * C;
        float lpb = row0[blank], lpl[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) lpl[j] = row0[lab[j]];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            be[j] = i == L ? (double)lpb : -INFINITY;
            bo[j] = (i == L - 1) ? (double)lpl[j] : -INFINITY;
        }
        float nb = 0.f, nl[NL];
        if (Tb > 1) {
            const float *row = lpt + (size_t)(Tb - 2) * C;
            nb = row[blank];
#pragma unroll
            for (int j = 0; j < NL; j++) nl[j] = row[lab[j]];
        }
        for (int t = Tb - 1;; t--) {
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int i = lane * NL + j;
                if (i <= L) *reinterpret_cast<double2 *>(wsb + (size_t)t * a.SP + 2 * i) = make_double2(be[j], bo[j]);
            }
            if (t == 0) break;
            lpb = nb;
#pragma unroll
            for (int j = 0; j < NL; j++) lpl[j] = nl[j];
            if (t >= 2) {
                const float *row = lpt + (size_t)(t - 2) * C;
                nb = row[blank];
#pragma unroll
                for (int j = 0; j < NL; j++) nl[j] = row[lab[j]];
            }
            const double er = wave_shift<DPP_WAVE_SHL1>(be[0]), orr = wave_shift<DPP_WAVE_SHL1>(bo[0]);
            double ne[NL], no[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const double ner = j == NL - 1 ? er : be[j + 1], nor_ = j == NL - 1 ? orr : bo[j + 1];
                ne[j] = wlse2(be[j], bo[j]) + (double)lpb;
                no[j] = wlse3(bo[j], ner, skip[j] ? nor_ : -INFINITY) + (double)lpl[j];
            }
#pragma unroll
            for (int j = 0; j < NL; j++) { be[j] = ne[j]; bo[j] = no[j]; }
        }
    }
    __syncthreads();
    const double ll = wlse2(s_fin[0], s_fin[1]);
    if (tid == 0) a.nll[b] = (float)(-ll);
    if (!a.grad) return;
    // ---------------- gradient rows, all waves, frames dealt round-robin
    float *g = gam + wave * LC;
    for (int t = wave; t < Tb; t += nwave) {
        const float *row = lpt + (size_t)t * C;
        const float lpb = row[blank];
        float ge = 0.f;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = lane * NL + j;
            float go = 0.f;
            if (i <= L) {
                const double2 al = *reinterpret_cast<const double2 *>(wsa + (size_t)t * a.SP + 2 * i);
                const double2 be = *reinterpret_cast<const double2 *>(wsb + (size_t)t * a.SP + 2 * i);
                ge += fexp((float)(al.x + be.x - ll - (double)lpb));
                if (i < L) go = fexp((float)(al.y + be.y - ll - (double)row[lab[j]]));
            }
            g[lane * NL + j] = go;
        }
        const float gblank = wave_sum(ge);
        wave_lds_order();     // the walk below sees this frame's values
        for (int c = lane; c < C; c += 64) {
            float acc = c == blank ? gblank : 0.f;
            for (int k = cls_off[c]; k < cls_off[c + 1]; k++) acc += g[cls_idx[k]];
            a.grad[((size_t)t * B + b) * C + c] = fexp(row[c]) - acc;
        }
        wave_lds_order();
    }
}

// dynamic LDS: row[2][Smax] double | ab[Smax] double | lab[Smax] int
__global__ __launch_bounds__(256) void ctc_generic_kernel(const float *__restrict__ logp, int T, int B, int C,
                                                          const int64_t *__restrict__ targets, int Lmax,
                                                          const int64_t *__restrict__ in_len, const int64_t *__restrict__ tgt_len,
                                                          int blank, float *__restrict__ nll_out, float *__restrict__ grad,
                                                          double *__restrict__ alpha_ws, int Smax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    double *row = reinterpret_cast<double *>(sm);      // [2][Smax]
    double *ab = row + 2 * Smax;                       // [Smax]
    int *lab = reinterpret_cast<int *>(ab + Smax);     // [Smax]
    __shared__ double s_ll;
    __shared__ int s_bad;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    int Tb = (int)in_len[b], L = (int)tgt_len[b];
    if (Tb > T) Tb = T;
    if (Tb < 0) Tb = 0;
    if (L > Lmax) L = Lmax;
    if (L < 0) L = 0;
    const int S = 2 * L + 1;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int s = tid; s < S; s += nth) {
        int l = blank;
        if (s & 1) {
            const long long v = targets[(size_t)b * Lmax + (s >> 1)];
            if (v < 0 || v >= C) s_bad = 1; else l = (int)v;
        }
        lab[s] = l;
    }
    if (grad)  // padded frames carry zero gradient
        for (size_t i = (size_t)Tb * C + tid; i < (size_t)T * C; i += nth) {
            const size_t t = i / C, c = i - t * C;
            grad[(t * B + b) * C + c] = 0.f;
        }
    __syncthreads();
    if (s_bad || Tb == 0) {
        if (tid == 0) nll_out[b] = s_bad ? NAN : ((L == 0) ? 0.f : INFINITY);
        if (grad && s_bad)
            for (size_t i = tid; i < (size_t)Tb * C; i += nth) { const size_t t = i / C, c = i - t * C; grad[(t * B + b) * C + c] = 0.f; }
        return;
    }
    double *aw = alpha_ws ? alpha_ws + (size_t)b * T * Smax : nullptr;
#define LP(t, c) ((double)logp[((size_t)(t) * B + b) * C + (c)])
    // ---- alpha
    for (int s = tid; s < S; s += nth) {
        double v = s == 0 ? LP(0, blank) : (s == 1 ? LP(0, lab[1]) : -INFINITY);
        row[s] = v;
        if (aw) aw[s] = v;
    }
    __syncthreads();
    for (int t = 1; t < Tb; t++) {
        const double *prev = row + ((t - 1) & 1) * Smax;
        double *curr = row + (t & 1) * Smax;
        for (int s = tid; s < S; s += nth) {
            const int l = lab[s];
            const double a0 = prev[s];
            const double a1 = s >= 1 ? prev[s - 1] : -INFINITY;
            const double a2 = (s >= 2 && l != blank && l != lab[s - 2]) ? prev[s - 2] : -INFINITY;
            const double m = lse3(a0, a1, a2);
            const double v = (m == -INFINITY) ? -INFINITY : m + LP(t, l);
            curr[s] = v;
            if (aw) aw[(size_t)t * Smax + s] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double *lastrow = row + ((Tb - 1) & 1) * Smax;
        const double ll = lse2(lastrow[S - 1], S > 1 ? lastrow[S - 2] : -INFINITY);
        s_ll = ll;
        nll_out[b] = (float)(-ll);
    }
    __syncthreads();
    if (!grad) return;
    const double ll = s_ll;
    // ---- beta (backwards) fused with the gradient of each frame
    for (int t = Tb - 1; t >= 0; t--) {
        double *curr = row + (t & 1) * Smax;
        const double *nxt = row + ((t + 1) & 1) * Smax;
        for (int s = tid; s < S; s += nth) {
            const int l = lab[s];
            double v;
            if (t == Tb - 1) {
                v = (s == S - 1) ? LP(t, blank) : ((s == S - 2) ? LP(t, l) : -INFINITY);
            } else {
                const double b0 = nxt[s];
                const double b1 = s + 1 < S ? nxt[s + 1] : -INFINITY;
                const double b2 = (s + 2 < S && l != blank && l != lab[s + 2]) ? nxt[s + 2] : -INFINITY;
                const double m = lse3(b0, b1, b2);
                v = (m == -INFINITY) ? -INFINITY : m + LP(t, l);
            }
            curr[s] = v;
            ab[s] = aw[(size_t)t * Smax + s] + v;
        }
        __syncthreads();
        for (int c = tid; c < C; c += nth) {
            double acc = -INFINITY;
            for (int s = 0; s < S; s++)
                if (lab[s] == c) acc = lse2(acc, ab[s]);
            const double lp = LP(t, c);
            grad[((size_t)t * B + b) * C + c] = (float)(exp(lp) - exp(acc - ll - lp));
        }
        __syncthreads();
    }
#undef LP
}

static size_t wave_smem(int T, int C, int NL) {
    const int LC = 64 * NL;
    return (size_t)T * C * 4 + (size_t)LC * 4 + (size_t)(C + 1) * 4 + (size_t)LC * 4 + (size_t)8 * LC * 4;
}

int init_ctc_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)ctc_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)ctc_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)ctc_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)ctc_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return MDD_OK;
}

}  // namespace mdd

// bytes of workspace mdd_ctc_loss needs for these shapes when a gradient is requested (0 without one)
extern "C" int64_t mdd_ctc_workspace_bytes(int32_t T, int32_t B, int32_t C, int32_t Lmax, int32_t want_grad) {
    if (!want_grad || T <= 0 || B <= 0 || Lmax < 0) return 0;
    const bool wave_form = Lmax <= 255 && C <= 256 && mdd::wave_smem(T, C, Lmax <= 63 ? 1 : (Lmax <= 127 ? 2 : 4)) <= 128 * 1024 &&
                           !(getenv("MDD_CTC") && !strcmp(getenv("MDD_CTC"), "generic"));
    if (wave_form) return (int64_t)sizeof(double) * B * 2 * T * (2 * ((int64_t)Lmax + 1));
    return (int64_t)sizeof(double) * B * T * (2 * (int64_t)Lmax + 1);
}

extern "C" int mdd_ctc_loss(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int64_t *targets_dev,
                            int32_t Lmax, const int64_t *in_len_dev, const int64_t *tgt_len_dev, int32_t blank,
                            float *nll_dev, float *grad_dev, void *workspace_dev, int64_t workspace_bytes, void *stream) {
    using namespace mdd;
    if (!logp_dev || !targets_dev || !in_len_dev || !tgt_len_dev || !nll_dev || T <= 0 || B <= 0 || C <= 0 || Lmax < 0 ||
        blank < 0 || blank >= C) {
        set_error("mdd_ctc_loss: bad argument"); return MDD_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    {   // kernel attributes, once per device
        static std::mutex mu;
        static bool done[64] = {false};
        int dev = 0;
        MDD_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev & 63]) { if (int rc = init_ctc_attributes()) return rc; done[dev & 63] = true; }
    }
    const int64_t need = mdd_ctc_workspace_bytes(T, B, C, Lmax, grad_dev != nullptr);
    double *ws = reinterpret_cast<double *>(workspace_dev);
    bool own_ws = false;
    if (need > 0 && (!ws || workspace_bytes < need)) {
        if (ws) { set_error("mdd_ctc_loss: workspace of %lld bytes, need %lld (mdd_ctc_workspace_bytes)", (long long)workspace_bytes, (long long)need); return MDD_ERR_ARG; }
        MDD_HIP_CHECK(hipMallocAsync((void **)&ws, (size_t)need, st));   // no caller workspace: stream-ordered allocation
        own_ws = true;
    }
    const int NL = Lmax <= 63 ? 1 : (Lmax <= 127 ? 2 : 4);
    const bool wave_form = Lmax <= 255 && C <= 256 && wave_smem(T, C, NL) <= 128 * 1024 &&
                           !(getenv("MDD_CTC") && !strcmp(getenv("MDD_CTC"), "generic"));
    if (wave_form) {
        CtcArgs a;
        a.logp = logp_dev; a.T = T; a.B = B; a.C = C; a.targets = targets_dev; a.Lmax = Lmax; a.in_len = in_len_dev; a.tgt_len = tgt_len_dev;
        a.blank = blank; a.nll = nll_dev; a.grad = grad_dev; a.ws = grad_dev ? ws : nullptr; a.SP = 2 * (Lmax + 1);
        const size_t smem = wave_smem(T, C, NL);
        if (NL == 1) hipLaunchKernelGGL(ctc_wave_kernel<1>, dim3(B), dim3(512), smem, st, a);
        else if (NL == 2) hipLaunchKernelGGL(ctc_wave_kernel<2>, dim3(B), dim3(512), smem, st, a);
        else hipLaunchKernelGGL(ctc_wave_kernel<4>, dim3(B), dim3(512), smem, st, a);
    } else {
        const int Smax = 2 * Lmax + 1;
        const size_t smem = sizeof(double) * 3 * (size_t)Smax + sizeof(int) * (size_t)Smax;
        if (smem > 150 * 1024) { if (own_ws) (void)hipFreeAsync(ws, st); set_error("mdd_ctc_loss: Lmax=%d too long for LDS", Lmax); return MDD_ERR_ARG; }
        hipLaunchKernelGGL(ctc_generic_kernel, dim3(B), dim3(256), smem, st, logp_dev, T, B, C, targets_dev, Lmax, in_len_dev,
                           tgt_len_dev, blank, nll_dev, grad_dev, grad_dev ? ws : nullptr, Smax);
    }
    hipError_t le = hipGetLastError();
    if (own_ws) (void)hipFreeAsync(ws, st);
    if (le != hipSuccess) { set_error("ctc kernel launch failed: %s", hipGetErrorString(le)); return MDD_ERR_HIP; }
    return MDD_OK;
}
