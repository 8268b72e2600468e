// CTC forward-backward (alpha/beta lattice) for the training step.
//
// Reference: nn.CTCLoss(reduction='sum') as called at AA/steps/train_ctc.py:72,186 (blank = 0, padded 2-D
// targets, input lengths = (frac * T').long()).  The arithmetic is ATen's; restated from the published
// algorithm (Graves et al. 2006, eq. 6-8, 10-11, 16).  Outputs: per-utterance nll (the reference's loss
// is their sum) and the tensor autograd deposits on the log-probs,
//   grad[t,b,c] = exp(lp) - exp(logsumexp_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll - lp),  t < in_len[b]
// and 0 on padded frames.
//
// One workgroup per utterance, one thread per lattice state s (S = 2L+1), serial in t with the
// previous row in LDS.  The lattice is kept in fp64: the scan is latency-bound (one barrier per
// frame), so the wider type is free, and it keeps the result closer to the exact value than the
// fp32 lattice ATen uses.  alpha rows are parked in HBM for the backward sweep (T*S*8 B per
// utterance; 162 KB at T'=250, L=40).
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

// dynamic LDS: row[2][Smax] double | ab[Smax] double | lab[Smax] int
__global__ __launch_bounds__(256) void ctc_kernel(const float *__restrict__ logp, int T, int B, int C,
                                                  const int64_t *__restrict__ targets, int Lmax,
                                                  const int64_t *__restrict__ in_len, const int64_t *__restrict__ tgt_len,
                                                  int blank, float *__restrict__ nll_out, float *__restrict__ grad,
                                                  double *__restrict__ alpha_ws, int Smax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    double *row = reinterpret_cast<double *>(sm);      // [2][Smax]
    double *ab = row + 2 * Smax;                       // [Smax]
    int *lab = reinterpret_cast<int *>(ab + Smax);     // [Smax]
    __shared__ double s_ll;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    int Tb = (int)in_len[b], L = (int)tgt_len[b];
    if (Tb > T) Tb = T;
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    for (int s = tid; s < S; s += nth) lab[s] = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : blank;
    if (grad)  // padded frames carry zero gradient
        for (size_t i = (size_t)(Tb < 0 ? 0 : Tb) * C + tid; i < (size_t)T * C; i += nth) {
            const size_t t = i / C, c = i - t * C;
            grad[(t * B + b) * C + c] = 0.f;
        }
    __syncthreads();
    if (Tb <= 0) { if (tid == 0) nll_out[b] = (L == 0) ? 0.f : INFINITY; return; }
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

}  // namespace mdd

extern "C" int mdd_ctc_loss(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int64_t *targets_dev,
                            int32_t Lmax, const int64_t *in_len_dev, const int64_t *tgt_len_dev, int32_t blank,
                            float *nll_dev, float *grad_dev, void *stream) {
    using namespace mdd;
    if (!logp_dev || !targets_dev || !in_len_dev || !tgt_len_dev || !nll_dev || T <= 0 || B <= 0 || C <= 0 || Lmax < 0 ||
        blank < 0 || blank >= C) {
        set_error("mdd_ctc_loss: bad argument"); return MDD_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int Smax = 2 * Lmax + 1;
    size_t smem = sizeof(double) * 3 * (size_t)Smax + sizeof(int) * (size_t)Smax;
    if (smem > 150 * 1024) { set_error("mdd_ctc_loss: Lmax=%d too long for LDS", Lmax); return MDD_ERR_ARG; }
    double *ws = nullptr;
    if (grad_dev) MDD_HIP_CHECK(hipMallocAsync((void **)&ws, sizeof(double) * (size_t)B * T * Smax, st));
    static bool attr_set = false;
    if (!attr_set) {
        MDD_HIP_CHECK(hipFuncSetAttribute((const void *)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(256), smem, st, logp_dev, T, B, C, targets_dev, Lmax, in_len_dev,
                       tgt_len_dev, blank, nll_dev, grad_dev, ws, Smax);
    hipError_t le = hipGetLastError();
    if (ws) (void)hipFreeAsync(ws, st);
    if (le != hipSuccess) { set_error("ctc kernel launch failed: %s", hipGetErrorString(le)); return MDD_ERR_HIP; }
    return MDD_OK;
}
