// CTC decoders on the GPU: greedy and prefix beam search.
//
// Reference: GreedyDecoder.decode (AA/utils/ctcDecoder.py:188-200 + :80-92) and
// BeamDecoder.decode -> ctcBeamSearch.decode (AA/utils/ctcDecoder.py:215-226, AA/utils/BeamSearch.py:73-153).
// Both are serial scans over the posterior frames of one utterance; they are latency-bound, not
// bandwidth- or flop-bound (180 B read per frame).  One workgroup per utterance.
#include "mdd_internal.h"

namespace mdd {

// ------------------------------------------------------------------------------------------ greedy
// argmax per frame (a wave per frame, first index wins ties like torch.max on CPU), then collapse:
// drop an element equal to its immediate predecessor (blank included in the comparison), drop blanks.
__global__ __launch_bounds__(256) void greedy_kernel(const float *__restrict__ logp, int T, int B, int C,
                                                     const int32_t *__restrict__ len, int blank,
                                                     int32_t *__restrict__ ids, int32_t *__restrict__ nids) {
    extern __shared__ int am[];  // [T]
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n = len[b];
    n = n < 0 ? 0 : (n > T ? T : n);
    for (int t = wave; t < n; t += 4) {
        const float *row = logp + ((size_t)t * B + b) * C;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            float v = row[c];
            if (v > bv || bi == 0x7fffffff) { bv = v; bi = c; }  // strict >: first index wins within a lane
        }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o);
            int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) am[t] = bi;
    }
    __syncthreads();
    if (wave != 0) return;
    int count = 0;
    for (int base = 0; base < n; base += 64) {
        const int t = base + lane;
        bool keep = false;
        int a = 0;
        if (t < n) {
            a = am[t];
            keep = (a != blank) && !(t != 0 && a == am[t - 1]);
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) ids[(size_t)b * T + count + __popcll(mask & ((1ull << lane) - 1ull))] = a;
        count += __popcll(mask);
    }
    if (lane == 0) nids[b] = count;
}

// ------------------------------------------------------------------------------------------ beam
#define LOG_ZERO (-99999999.0)  // AA/utils/BeamSearch.py:6

__device__ __forceinline__ double log_add_prob(double lx, double ly) {  // BeamSearch.py:43-50
    if (lx <= LOG_ZERO) return ly;
    if (ly <= LOG_ZERO) return lx;
    if ((ly - lx) > 0.0) { double t = lx; lx = ly; ly = t; }
    return lx + log(1 + exp(ly - lx));
}

struct BeamMeta {       // one entry of `last` (BeamSearch.py:9-15); prefix bytes live in a separate LDS array
    double prTotal, prNonBlank, prBlank;
    unsigned long long hash, phash;  // rolling hash of the prefix and of the prefix without its last symbol
    int len, last;
};

constexpr int MAXBEAM = 64;

__device__ __forceinline__ void wave_argmax(double &v, int &ord) {
    // all-reduce: larger value wins, ties -> smaller insertion order (Python's stable sorted(reverse=True))
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oo = __shfl_xor(ord, o);
        if (ov > v || (ov == v && oo < ord)) { v = ov; ord = oo; }
    }
}

// One wave per utterance.  dynamic LDS layout (bytes):
//   tot[beam*C] double | lp[C] double | p[C] float | prefix[2][beam][Tcap] uint8
__global__ __launch_bounds__(64) void beam_kernel(const float *__restrict__ logp, int T, int B, int C,
                                                  const int32_t *__restrict__ len, int beam, int blank,
                                                  const double *__restrict__ lm, double alpha, int32_t *__restrict__ ids,
                                                  int32_t *__restrict__ nids, int32_t *__restrict__ status,
                                                  double *__restrict__ score, int Tcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    double *tot = reinterpret_cast<double *>(sm);
    double *lp = tot + (size_t)beam * C;
    float *p = reinterpret_cast<float *>(lp + C);
    unsigned char *pref = reinterpret_cast<unsigned char *>(p + ((C + 3) & ~3));
    __shared__ BeamMeta meta[2][MAXBEAM];
    __shared__ double cNB[MAXBEAM], cB[MAXBEAM], cT[MAXBEAM];  // copy-path contributions per beam
    __shared__ int parent[MAXBEAM], mslot[MAXBEAM];             // parent beam index / ext slot merged with this beam's copy
    __shared__ double sel_v[MAXBEAM];
    __shared__ int sel_ord[MAXBEAM];

    const int b = blockIdx.x, lane = threadIdx.x;
    const int C1 = C + 1;
    int cur = 0, nb = 1, err = 0;
    if (lane == 0) {
        BeamMeta &m = meta[0][0];
        m.prTotal = 0.0; m.prBlank = 0.0; m.prNonBlank = LOG_ZERO; m.len = 0; m.last = -1;
        m.hash = 0x243F6A8885A308D3ull; m.phash = 0;
    }
    __syncthreads();
    int tl = len[b];
    tl = tl < 0 ? 0 : (tl > T ? T : tl);

    for (int t = 0; t < tl; t++) {
        const float *row = logp + ((size_t)t * B + b) * C;
        // probabilities as the reference sees them: exp() of the fp32 log-prob, rounded to fp32
        // (ctcDecoder.py:224); computed in fp64 then rounded so the fp32 value is correctly rounded.
        const float pblank = (float)exp((double)row[blank]);
        if ((1.0f - pblank) < 0.1f) continue;  // BeamSearch.py:93-94 (float32 compare); wave-uniform
        for (int c = lane; c < C; c += 64) {
            const float pc = (float)exp((double)row[c]);
            p[c] = pc;
            lp[c] = pc > 0.0f ? log((double)pc) : 0.0;
        }
        float pprevb = 0.f;
        if (t > 0) pprevb = (float)exp((double)logp[((size_t)(t - 1) * B + b) * C + blank]);  // raw previous row (:63)
        __syncthreads();
        const BeamMeta *last = meta[cur];
        unsigned char *pcur = pref + (size_t)cur * beam * Tcap, *pnext = pref + (size_t)(cur ^ 1) * beam * Tcap;

        // ---- candidate scores.  slot idx = r*C + k ; k == blank is beam r's own ("copy") entry
        int first_err_ord = 0x7fffffff, first_err = 0;
        for (int idx = lane; idx < nb * C; idx += 64) {
            const int r = idx / C, k = idx - r * C;
            if (k == blank) continue;
            const BeamMeta &y = last[r];
            const double lmv = lm[(size_t)(y.len ? y.last : C) * C1 + k];  // consulted even when alpha == 0 (:57-60)
            const int ord = r * C1 + 1 + k;
            if (lmv != lmv) { if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_KEY_ERROR; } }
            else if (p[k] == 0.0f) { if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_VALUE_ERROR; } }
            const double bigram = lmv * alpha;
            const double base = (y.len && y.last == k && pprevb < 0.9f) ? y.prBlank : y.prTotal;  // :63-66
            tot[idx] = lp[k] + bigram + base;
        }
        if (lane < nb) {
            const BeamMeta &y = last[lane];
            double pnb = LOG_ZERO;
            bool bad = (p[blank] == 0.0f);
            if (y.len > 0) { pnb = y.prNonBlank + lp[y.last]; bad = bad || (p[y.last] == 0.0f); }  // :103
            const double pb = y.prTotal + lp[blank];                                                // :106
            if (bad) { const int ord = lane * C1; if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_VALUE_ERROR; } }
            cNB[lane] = pnb; cB[lane] = pb;
            const double tc = log_add_prob(pb, pnb);                                                // :112
            cT[lane] = tc;
            tot[lane * C + blank] = tc;
            // parent search: y_q == y[:-1] ?
            int par = -1;
            if (y.len > 0) {
                for (int q = 0; q < nb; q++) {
                    if (q == lane || last[q].len != y.len - 1 || last[q].hash != y.phash) continue;
                    bool same = true;  // verify the content (hash collisions must not merge distinct prefixes)
                    const unsigned char *pa = pcur + (size_t)lane * Tcap, *pq = pcur + (size_t)q * Tcap;
                    for (int i = 0; i < y.len - 1; i++) if (pa[i] != pq[i]) { same = false; break; }
                    if (same) { par = q; break; }
                }
            }
            parent[lane] = par;
            mslot[lane] = -1;
        }
        {   // first error in the reference's execution order wins
            int eo = first_err_ord;
            for (int o = 32; o > 0; o >>= 1) eo = min(eo, __shfl_xor(eo, o));
            if (eo != 0x7fffffff) {
                const unsigned long long who = __ballot(first_err_ord == eo);
                err = __shfl(first_err, __ffsll((long long)who) - 1);
                break;
            }
        }
        __syncthreads();
        // ---- merges: beam a's copy entry and its parent's extension by a's last symbol are one dict entry
        int nmerge = 0;
        if (lane < nb && parent[lane] >= 0) {
            const int a = lane, q = parent[a], e = q * C + last[a].last;
            const double pr = tot[e];
            if (q < a) {  // entry was created by the extension (inserted earlier), then the copy is added (:108-113)
                cNB[a] = log_add_prob(pr, cNB[a]);
                cT[a] = log_add_prob(pr, cT[a]);
                tot[e] = cT[a];
                tot[a * C + blank] = -INFINITY;
                mslot[a] = e;
            } else {      // entry was created by the copy, then the extension is added (:122-125)
                cNB[a] = log_add_prob(cNB[a], pr);
                cT[a] = log_add_prob(cT[a], pr);
                tot[a * C + blank] = cT[a];
                tot[e] = -INFINITY;
            }
            nmerge = 1;
        }
        nmerge = __popcll(__ballot(nmerge != 0));
        __syncthreads();
        // ---- `last.sort()[0:beam]`: stable, descending by prTotal; insertion order = (r, copy first, then k ascending)
        const int ncand = nb * C - nmerge;
        const int keep = ncand < beam ? ncand : beam;
        double lv = -INFINITY;
        int lord = 0x7fffffff;
        auto ord_of = [&](int idx) { const int r = idx / C, k = idx - r * C; return r * C + (k == blank ? 0 : (k < blank ? k + 1 : k)); };
        auto idx_of = [&](int ord) { const int r = ord / C, j = ord - r * C; return r * C + (j == 0 ? blank : (j <= blank ? j - 1 : j)); };
        for (int idx = lane; idx < nb * C; idx += 64) {
            const double v = tot[idx];
            const int o = ord_of(idx);
            if (v > lv || (v == lv && o < lord)) { lv = v; lord = o; }
        }
        for (int i = 0; i < keep; i++) {
            double v = lv; int o = lord;
            wave_argmax(v, o);
            if (lane == 0) { sel_v[i] = v; sel_ord[i] = o; }
            if (o == lord && lord != 0x7fffffff) {  // this lane owned the winner: retire it and rescan its slots
                tot[idx_of(o)] = -INFINITY;
                lv = -INFINITY; lord = 0x7fffffff;
                for (int idx = lane; idx < nb * C; idx += 64) {
                    const double v2 = tot[idx];
                    const int o2 = ord_of(idx);
                    if (v2 > lv || (v2 == lv && o2 < lord)) { lv = v2; lord = o2; }
                }
            }
        }
        __syncthreads();
        // ---- materialise the new beams
        BeamMeta *next = meta[cur ^ 1];
        if (lane < keep) {
            const int idx = idx_of(sel_ord[lane]);
            const int r = idx / C, k = idx - r * C;
            const BeamMeta &y = last[r];
            BeamMeta n;
            if (k == blank) {
                n.prTotal = sel_v[lane]; n.prNonBlank = cNB[r]; n.prBlank = cB[r];
                n.len = y.len; n.last = y.last; n.hash = y.hash; n.phash = y.phash;
            } else {
                n.prTotal = sel_v[lane]; n.prNonBlank = sel_v[lane]; n.prBlank = LOG_ZERO;
                for (int a = 0; a < nb; a++)
                    if (mslot[a] == idx) { n.prNonBlank = cNB[a]; n.prBlank = cB[a]; }
                n.len = y.len + 1; n.last = k; n.phash = y.hash;
                n.hash = y.hash * 0x9E3779B97F4A7C15ull + (unsigned long long)(k + 1);
            }
            next[lane] = n;
        }
        for (int i = 0; i < keep; i++) {
            const int idx = idx_of(sel_ord[i]);
            const int r = idx / C, k = idx - r * C;
            const int ln = last[r].len;
            const unsigned char *src = pcur + (size_t)r * Tcap;
            unsigned char *dst = pnext + (size_t)i * Tcap;
            for (int j = lane; j < ln; j += 64) dst[j] = src[j];
            if (k != blank && lane == 0) dst[ln] = (unsigned char)k;
        }
        __syncthreads();
        cur ^= 1;
        nb = keep;
    }
    __syncthreads();
    // ---- final: EOS LM term, length normalisation, first maximum (:130-148)
    if (lane == 0) {
        const BeamMeta *last = meta[cur];
        int best = -1;
        double bestv = 0.0;
        for (int r = 0; r < nb && !err; r++) {
            const BeamMeta &y = last[r];
            if (y.len == 0) { err = MDD_BEAM_INDEX_ERROR; break; }   // y[-1] on the empty tuple (:135)
            const double v = lm[(size_t)y.last * C1 + C];
            if (v != v) { err = MDD_BEAM_KEY_ERROR; break; }
            double pr = log_add_prob(LOG_ZERO, y.prTotal + v * alpha);
            pr = pr * (1.0 / (double)(y.len ? y.len : 1));
            if (best < 0 || pr > bestv) { best = r; bestv = pr; }
        }
        status[b] = err;
        int n = 0;
        if (!err && best >= 0) {
            n = last[best].len;
            const unsigned char *src = pref + (size_t)cur * beam * Tcap + (size_t)best * Tcap;
            for (int j = 0; j < n; j++) ids[(size_t)b * T + j] = src[j];
        }
        nids[b] = n;
        if (score) score[b] = err ? __builtin_nan("") : bestv;
    }
}

}  // namespace mdd

extern "C" int mdd_greedy(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int32_t *len_dev, int32_t blank,
                          int32_t *ids_dev, int32_t *nids_dev, void *stream) {
    using namespace mdd;
    if (!logp_dev || !len_dev || !ids_dev || !nids_dev || T <= 0 || B <= 0 || C <= 0 || blank < 0 || blank >= C) {
        set_error("mdd_greedy: bad argument"); return MDD_ERR_ARG;
    }
    if ((size_t)T * 4 > 150 * 1024) { set_error("mdd_greedy: T=%d too long", T); return MDD_ERR_ARG; }
    hipLaunchKernelGGL(greedy_kernel, dim3(B), dim3(256), (size_t)T * sizeof(int), (hipStream_t)stream, logp_dev, T, B, C,
                       len_dev, blank, ids_dev, nids_dev);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

extern "C" int mdd_beam(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int32_t *len_dev, int32_t beam,
                        int32_t blank, const double *lm_dev, double lm_alpha, int32_t *ids_dev, int32_t *nids_dev,
                        int32_t *status_dev, double *score_dev, void *stream) {
    using namespace mdd;
    if (!logp_dev || !len_dev || !lm_dev || !ids_dev || !nids_dev || !status_dev || T <= 0 || B <= 0 || C <= 1 ||
        C > 256 || beam < 1 || beam > MAXBEAM || blank < 0 || blank >= C) {
        set_error("mdd_beam: bad argument (need 1<=beam<=64, 2<=C<=256)"); return MDD_ERR_ARG;
    }
    const int Tcap = (T + 1 + 3) & ~3;
    size_t smem = sizeof(double) * ((size_t)beam * C + C) + sizeof(float) * ((C + 3) & ~3) + (size_t)2 * beam * Tcap;
    if (smem > 140 * 1024) { set_error("mdd_beam: beam*C / T too large for LDS (%zu B)", smem); return MDD_ERR_ARG; }
    static bool attr_set = false;
    if (!attr_set) {
        MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(beam_kernel, dim3(B), dim3(64), smem, (hipStream_t)stream, logp_dev, T, B, C, len_dev, beam, blank,
                       lm_dev, lm_alpha, ids_dev, nids_dev, status_dev, score_dev, Tcap);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
