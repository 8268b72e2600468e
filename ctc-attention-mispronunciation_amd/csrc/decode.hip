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
    // branch-free (every lane evaluates the sum, the early returns are selects): the ~150 fp64 instructions can then be
    // scheduled among the neighbouring loads instead of sitting behind their own exec-mask branches
    const bool swap = (ly - lx) > 0.0;
    const double hi = swap ? ly : lx, lo = swap ? lx : ly;
    const double sum = hi + log(1 + exp(lo - hi));
    return lx <= LOG_ZERO ? ly : (ly <= LOG_ZERO ? lx : sum);
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

// ---- pass 1 (parallel over frames): everything about a frame that does not depend on the beam state.
//   lp[t,b,c]   = log((double)p) with p = fp32(exp(logp))   (-inf where p == 0: the reference raises ValueError
//                 when such a value is needed, BeamSearch.py:64,66,103,106)
//   flags[t,b]  bit0: frame is live, i.e. NOT (1 - p_blank < 0.1) in float32 (BeamSearch.py:93-94)
//               bit1: p_blank of the raw previous frame < 0.9 in float32 (the repeat rule, :63)
// exp() is evaluated in fp64 and rounded once, so p is the correctly rounded fp32 value the reference's
// torch.exp (ctcDecoder.py:224) produces on ~99% of inputs (<= 1 ulp otherwise).
__global__ __launch_bounds__(256) void beam_prep_kernel(const float *__restrict__ logp, int T, int B, int C, int blank,
                                                        double *__restrict__ lp, unsigned char *__restrict__ flags) {
    const size_t n = (size_t)T * B * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = (float)exp((double)logp[i]);
        lp[i] = p > 0.0f ? log((double)p) : -INFINITY;
        const size_t tb = i / C;
        if ((int)(i - tb * C) == blank) {
            unsigned char f = ((1.0f - p) < 0.1f) ? 0 : 1;
            const size_t t = tb / B;
            if (t > 0) {
                const float pprev = (float)exp((double)logp[i - (size_t)B * C]);
                if (pprev < 0.9f) f |= 2;
            }
            flags[tb] = f;
        }
    }
}

// ---- pass 2: one wave per utterance, serial over the live frames.  dynamic LDS layout (bytes):
//   tot[beam*C] double | lmt[(C+1)*(C+1)] double (if it fits) | cl_v[beam*C] double | cl_o[beam*C] int |
//   flags[Tcap] uint8 | prefix[2][beam][Tcap] uint8
__global__ __launch_bounds__(64) void beam_kernel(const double *__restrict__ lpw, const unsigned char *__restrict__ flags,
                                                  int T, int B, int C, const int32_t *__restrict__ len, int beam, int blank,
                                                  const double *__restrict__ lm, double alpha, int32_t *__restrict__ ids,
                                                  int32_t *__restrict__ nids, int32_t *__restrict__ status,
                                                  double *__restrict__ score, int Tcap, int lm_in_lds, int skip, long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int C1 = C + 1;
    double *tot = reinterpret_cast<double *>(sm);
    double *lmt = tot + (size_t)beam * C;
    double *cl_v = lmt + (lm_in_lds ? C1 * C1 : 0);                  // compacted candidate list (values)
    int *cl_o = reinterpret_cast<int *>(cl_v + (size_t)beam * C);    //   and their insertion orders
    unsigned char *fl = reinterpret_cast<unsigned char *>(cl_o + (size_t)beam * C);  // per-frame flags of this utterance [Tcap]
    unsigned char *pref = fl + Tcap;
    __shared__ double lp[256];
    __shared__ int nlist;
    __shared__ BeamMeta meta[2][MAXBEAM];
    __shared__ double cNB[MAXBEAM], cB[MAXBEAM], cT[MAXBEAM];  // copy-path contributions per beam
    __shared__ int parent[MAXBEAM], mslot[MAXBEAM];             // parent beam index / ext slot merged with this beam's copy
    __shared__ double sel_v[MAXBEAM];
    __shared__ int sel_ord[MAXBEAM];

    const int b = blockIdx.x, lane = threadIdx.x;
    int cur = 0, nb = 1, err = 0;
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tstamp = 0;
#define STAMP(i) do { if (dbg) { long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - tstamp; tstamp = now_; } } while (0)
    if (lane == 0) {
        BeamMeta &m = meta[0][0];
        m.prTotal = 0.0; m.prBlank = 0.0; m.prNonBlank = LOG_ZERO; m.len = 0; m.last = -1;
        m.hash = 0x243F6A8885A308D3ull; m.phash = 0;
    }
    if (lm_in_lds) for (int i = lane; i < C1 * C1; i += 64) lmt[i] = lm[i];
    const double *lmp = lm_in_lds ? lmt : lm;
    __syncthreads();
    int tl = len[b];
    tl = tl < 0 ? 0 : (tl > T ? T : tl);
    for (int i = lane; i < tl; i += 64) fl[i] = flags[(size_t)i * B + b];
    __syncthreads();

    // software prefetch: the row of the next live frame is loaded while the current one is processed
    int t = 0;
    while (t < tl && !(fl[t] & 1)) t++;
    double pre[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pre[i] = (t < tl && lane + 64 * i < C) ? lpw[((size_t)t * B + b) * C + lane + 64 * i] : 0.0;
    if (dbg) tstamp = __builtin_readcyclecounter();
    while (t < tl) {
        const bool rep_ok = (fl[t] & 2) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) if (lane + 64 * i < C) lp[lane + 64 * i] = pre[i];
        int tn = t + 1;
        while (tn < tl && !(fl[tn] & 1)) tn++;
#pragma unroll
        for (int i = 0; i < 4; i++) pre[i] = (tn < tl && lane + 64 * i < C) ? lpw[((size_t)tn * B + b) * C + lane + 64 * i] : 0.0;
        __syncthreads();
        const BeamMeta *last = meta[cur];
        unsigned char *pcur = pref + (size_t)cur * beam * Tcap, *pnext = pref + (size_t)(cur ^ 1) * beam * Tcap;

        STAMP(0);
        // ---- candidate scores.  slot idx = r*C + k ; k == blank is beam r's own ("copy") entry.
        // lanes own the class index k (k = lane, lane+64, ..), the beam rank r is a uniform loop.
        int first_err_ord = 0x7fffffff, first_err = 0;
        for (int r = 0; r < nb && !(skip & 1); r++) {
            const BeamMeta &y = last[r];
            const int ylen = y.len, ylast = y.last;
            const double yB = y.prBlank, yT = y.prTotal;
            const double *lmrow = lmp + (ylen ? ylast : C) * C1;
            for (int k = lane; k < C; k += 64) {
                if (k == blank) continue;
                const double lmv = lmrow[k];  // consulted even when alpha == 0 (:57-60)
                const double lk = lp[k];
                const int ord = r * C1 + 1 + k;
                if (lmv != lmv) { if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_KEY_ERROR; } }
                else if (lk == -INFINITY) { if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_VALUE_ERROR; } }
                const double base = (ylen && ylast == k && rep_ok) ? yB : yT;  // :63-66
                tot[r * C + k] = lk + lmv * alpha + base;
            }
        }
        STAMP(1);
        if (lane < nb) {
            const BeamMeta &y = last[lane];
            double pnb = LOG_ZERO;
            bool bad = (lp[blank] == -INFINITY);
            if (y.len > 0) { pnb = y.prNonBlank + lp[y.last]; bad = bad || (lp[y.last] == -INFINITY); }  // :103
            const double pb = y.prTotal + lp[blank];                                                      // :106
            if (bad) { const int ord = lane * C1; if (ord < first_err_ord) { first_err_ord = ord; first_err = MDD_BEAM_VALUE_ERROR; } }
            cNB[lane] = pnb; cB[lane] = pb;
            const double tc = (skip & 2) ? pb : log_add_prob(pb, pnb);                                   // :112
            cT[lane] = tc;
            tot[lane * C + blank] = tc;
            // parent candidate: a beam q whose (length, hash) equal those of y without its last symbol
            int par = -1;
            if (y.len > 0)
                for (int q = 0; q < nb; q++)
                    if (q != lane && last[q].len == y.len - 1 && last[q].hash == y.phash) { par = q; break; }
            parent[lane] = par;
            mslot[lane] = -1;
        }
        STAMP(2);
        // verify the content of every hash match with the whole wave (a hash collision must never merge two
        // distinct prefixes): 64 lanes compare 64 words per pass, the tail word is masked to the prefix length
        for (int a = 0; a < nb && !(skip & 4); a++) {
            const int q = parent[a];          // uniform (LDS broadcast; written by lane a above, same wave, in order)
            if (q < 0) continue;
            const int ln = last[q].len;
            const unsigned int *wa = reinterpret_cast<const unsigned int *>(pcur + (size_t)a * Tcap);
            const unsigned int *wq = reinterpret_cast<const unsigned int *>(pcur + (size_t)q * Tcap);
            bool neq = false;
            for (int j = lane; j * 4 < ln; j += 64) {
                unsigned int x = wa[j] ^ wq[j];
                const int rem = ln - j * 4;
                if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                neq = neq || (x != 0);
            }
            if (__any(neq) && lane == 0) parent[a] = -1;
        }
        STAMP(3);
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
        if (lane < nb && parent[lane] >= 0 && !(skip & 8)) {
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
        STAMP(4);
        // ---- `last.sort()[0:beam]`: stable, descending by prTotal; insertion order = (r, copy first, then k ascending)
        const int ncand = nb * C - nmerge;
        const int keep = ncand < beam ? ncand : beam;
        // Top-`keep` of the <= nb*C candidates under (value desc, insertion order asc) without serial pops:
        //  1. theta0 = keep-th largest of the 64 per-lane maxima (rank counting over v_readlane broadcasts); at
        //     least `keep` candidates are >= theta0, so every member of the true top-`keep` is too;
        //  2. the few candidates >= theta0 are compacted into an LDS list (order irrelevant);
        //  3. each list entry counts the entries that beat it; rank < keep -> it IS output position `rank`.
        // insertion order of slot (r,k): r*C + (k == blank ? 0 : (k < blank ? k+1 : k)).
        const float invC = 1.0f / (float)C;
        auto idx_of = [&](int ord) {
            const int r = (int)(((float)ord + 0.5f) * invC), j = ord - r * C;
            return r * C + (j == 0 ? blank : (j <= blank ? j - 1 : j));
        };
        if (!(skip & 16)) {
            double lmax = -INFINITY;
            for (int r = 0; r < nb; r++)
                for (int k = lane; k < C; k += 64) lmax = fmax(lmax, tot[r * C + k]);
            STAMP(5);
            int rk = 0;
            {
                const int lo = __double2loint(lmax), hi = __double2hiint(lmax);
#pragma unroll 8
                for (int j = 0; j < 64; j++) {
                    const double sj = __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
                    rk += (sj > lmax || (sj == lmax && j < lane)) ? 1 : 0;
                }
            }
            const unsigned long long mk = __ballot(rk == keep - 1);
            const int srcl = __ffsll((long long)mk) - 1;
            const double theta0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lmax), srcl),
                                                   __builtin_amdgcn_readlane(__double2loint(lmax), srcl));
            STAMP(6);
            if (lane == 0) nlist = 0;
            __syncthreads();
            if (lmax >= theta0) {
                int cnt = 0;
                for (int r = 0; r < nb; r++)
                    for (int k = lane; k < C; k += 64) cnt += (tot[r * C + k] >= theta0) ? 1 : 0;
                int pos = atomicAdd(&nlist, cnt);
                for (int r = 0; r < nb; r++)
                    for (int k = lane; k < C; k += 64) {
                        const double v = tot[r * C + k];
                        if (v >= theta0) { cl_v[pos] = v; cl_o[pos] = r * C + (k == blank ? 0 : (k < blank ? k + 1 : k)); pos++; }
                    }
            }
            __syncthreads();
            const int nl = nlist;
            STAMP(7);
            if (dbg) ph[9] += nl;
            for (int e = lane; e < nl; e += 64) {
                const double v = cl_v[e];
                const int o = cl_o[e];
                int rank = 0;
                for (int f = 0; f < nl; f++) {
                    const double vf = cl_v[f];
                    const int of = cl_o[f];
                    rank += (vf > v || (vf == v && of < o)) ? 1 : 0;
                }
                if (rank < keep) { sel_v[rank] = v; sel_ord[rank] = idx_of(o); }
            }
        } else if (lane < keep) { sel_v[lane] = 0.0; sel_ord[lane] = lane * C + 1; }
        __syncthreads();
        STAMP(8);
        // ---- materialise the new beams
        BeamMeta *next = meta[cur ^ 1];
        if (lane < keep) {
            const int idx = sel_ord[lane];
            const int r = (int)(((float)idx + 0.5f) * invC), k = idx - r * C;
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
        for (int i = 0; i < keep && !(skip & 32); i++) {
            const int idx = sel_ord[i];
            const int r = (int)(((float)idx + 0.5f) * invC), k = idx - r * C;
            const int ln = last[r].len;
            const unsigned int *src = reinterpret_cast<const unsigned int *>(pcur + (size_t)r * Tcap);
            unsigned int *dst = reinterpret_cast<unsigned int *>(pnext + (size_t)i * Tcap);
            for (int j = lane; j * 4 < ln; j += 64) dst[j] = src[j];      // Tcap % 4 == 0: whole words
            __builtin_amdgcn_wave_barrier();
            if (k != blank && lane == 0) reinterpret_cast<unsigned char *>(dst)[ln] = (unsigned char)k;
        }
        __syncthreads();
        cur ^= 1;
        nb = keep;
        t = tn;
        if (dbg) tstamp = __builtin_readcyclecounter();
    }
    if (dbg && lane == 0) for (int i = 0; i < 10; i++) dbg[b * 10 + i] = ph[i];
    __syncthreads();
    // ---- final: EOS LM term, length normalisation, first maximum (:130-148)
    if (lane == 0) {
        const BeamMeta *last = meta[cur];
        int best = -1;
        double bestv = 0.0;
        for (int r = 0; r < nb && !err; r++) {
            const BeamMeta &y = last[r];
            if (y.len == 0) { err = MDD_BEAM_INDEX_ERROR; break; }   // y[-1] on the empty tuple (:135)
            const double v = lmp[y.last * C1 + C];
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

// ---- pass 2, fast path (beam <= 16, C <= 64): same algorithm, restructured for a single wave's latency.
// A lone wave pays ~100 cycles for every dependent LDS round trip, so the work is arranged in unrolled phases whose
// loads are all independent: slots are dealt to lanes round-robin (slot = lane + 64 i, coordinates precomputed once),
// the beam state is a structure of arrays, candidate operands are gathered before anything is stored, the top-`beam`
// selection reads each lane's slots into registers once and then uses wave ballots / v_readlane only:
//   lane maxima -> rank by v_readlane broadcast -> theta0 (keep-th largest lane maximum; because slots are dealt
//   round-robin the true winners sit in different lanes and theta0 is tight) -> ballot compaction of the ~10-20
//   candidates >= theta0 -> rank counting among those (rank < keep IS the output position).
constexpr int FB = 16;   // max beams on the fast path
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// Workgroup shape: W = blockDim.x / 64 independent waves, one utterance each, sharing only the LM table.  A beam wave
// holds ~210 VGPRs, so a CU that hosts one cannot take a workgroup of the forward's GEMM (2 x 256 VGPRs per SIMD) or
// persistent BiLSTM (414): dealt one wave per CU the search would fence the whole chip off from the next batch's
// forward for its entire duration.  Packed W = 4 to a CU (one per SIMD, each with the full 512-VGPR budget so that a
// phase's loads can all be in flight) it occupies B / 4 CUs and the forward keeps the rest.  Waves never meet at a workgroup barrier after the LM load
// (utterance lengths differ); inside a wave, LDS operations complete in order, so a wave-scope fence is the only
// synchronisation between the phases.
#define BEAM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// LDS bytes of one wave's private state (NS*64 slots, beam prefixes of up to Tcap ids)
__host__ __device__ inline size_t beam_fast_wave_bytes(int NS, int beam, int Tcap) {
    return sizeof(double) * ((size_t)2 * NS * 64 + 64 + 6 * FB + 4 * FB) + sizeof(unsigned long long) * 4 * FB +
           sizeof(int) * ((size_t)2 * NS * 64 + 4 * FB + 2 * FB) + (size_t)(2 * FB + 1) * Tcap;   // FB prefix rows whatever the beam: no row guards
}
__host__ __device__ inline size_t beam_fast_lm_bytes(int C) { return (sizeof(double) * (size_t)(C + 1) * (C + 1) + 15) & ~(size_t)15; }

// Coding rules of this kernel (a lone wave pays ~120 cycles per dependent LDS round trip and ~4 cycles per VALU
// instruction, and the compiler turns every conditional load into a branch with its own wait):
//   * loads are unconditional, from clamped (always valid) addresses, gathered at the top of a phase; conditions are
//     applied to registers afterwards; per-beam arrays of FB entries are read whole with 16-byte loads;
//   * everything wave-uniform (frame counters, beam count, keep, list lengths) is forced into SGPRs with
//     v_readfirstlane so that the control flow stays scalar;
//   * per-beam values (copy-path scores, parent, merge slot) stay in the registers of lane == beam and move between
//     lanes with v_readlane / ballots rather than through LDS.
template <int NS, bool DBG>
__global__ __launch_bounds__(256) void beam_fast_kernel(const double *__restrict__ lpw, const unsigned char *__restrict__ flags,
                                                       int T, int B, int C, const int32_t *__restrict__ len, int beam, int blank,
                                                       const double *__restrict__ lm, double alpha, int32_t *__restrict__ ids,
                                                       int32_t *__restrict__ nids, int32_t *__restrict__ status,
                                                       double *__restrict__ score, int Tcap, long long *__restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int C1 = C + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *lmt = reinterpret_cast<double *>(sm);                      // [C1*C1], shared by the waves
    unsigned char *mine = sm + beam_fast_lm_bytes(C) + (size_t)wave * ((beam_fast_wave_bytes(NS, beam, Tcap) + 15) & ~(size_t)15);
    double *tot = reinterpret_cast<double *>(mine);                    // [NS*64]
    double *cl_v = tot + NS * 64;                                      // [NS*64] compacted candidates: value,
    double *lp = cl_v + NS * 64;                                       // [64]
    double (*m_T)[FB] = reinterpret_cast<double (*)[FB]>(lp + 64);     // [2][FB] beam state, double-buffered
    double (*m_NB)[FB] = m_T + 2, (*m_B)[FB] = m_NB + 2;
    double *cNB = reinterpret_cast<double *>(m_B + 2), *cB = cNB + FB, *cT = cB + FB, *sel_v = cT + FB;
    unsigned long long (*m_hash)[FB] = reinterpret_cast<unsigned long long (*)[FB]>(sel_v + FB), (*m_phash)[FB] = m_hash + 2;
    int *cl_o = reinterpret_cast<int *>(m_phash + 2);                  //   insertion order,
    int *cl_x = cl_o + NS * 64;                                        //   slot index
    int (*m_len)[FB] = reinterpret_cast<int (*)[FB]>(cl_x + NS * 64), (*m_last)[FB] = m_len + 2;
    int *mslot = reinterpret_cast<int *>(m_last + 2), *sel_x = mslot + FB;
    unsigned char *fl = reinterpret_cast<unsigned char *>(sel_x + FB);
    unsigned char *pref = fl + Tcap;

    for (int i = threadIdx.x; i < C1 * C1; i += blockDim.x) lmt[i] = lm[i];
    __syncthreads();                                                   // the only workgroup barrier
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= B) return;
    int cur = 0, nb = 1, err = 0;
    long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tstamp = 0;   // MDD_BEAM_DBG: cycles per phase
#define FSTAMP(i) do { if (DBG) { long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - tstamp; tstamp = now_; } } while (0)
    if (lane < FB) {   // beam 0 = the empty prefix; the other entries only need to be valid numbers
        m_T[0][lane] = 0.0; m_B[0][lane] = 0.0; m_NB[0][lane] = LOG_ZERO; m_len[0][lane] = lane == 0 ? 0 : -2; m_last[0][lane] = -1;
        m_hash[0][lane] = 0x243F6A8885A308D3ull; m_phash[0][lane] = 0;
        m_T[1][lane] = 0.0; m_B[1][lane] = 0.0; m_NB[1][lane] = LOG_ZERO; m_len[1][lane] = 0; m_last[1][lane] = -1;
        m_hash[1][lane] = 0; m_phash[1][lane] = 0;
        mslot[lane] = -1; sel_x[lane] = 0; sel_v[lane] = 0.0; cNB[lane] = 0.0; cB[lane] = 0.0;
    }
    int tl = __builtin_amdgcn_readfirstlane(len[b]);
    tl = tl < 0 ? 0 : (tl > T ? T : tl);
    for (int i = lane; i < tl; i += 64) fl[i] = flags[(size_t)i * B + b];
    // slot coordinates of this lane (constant over the whole utterance): slot = lane + 64 i = r*C + k
    // (kept as data, not as hoisted lane masks: a dozen loop-invariant 64-bit masks cost more SGPRs than the wave has)
    int sr[NS], sk[NS], sx[NS];                                        // sx: the slot index, or a huge one for copy slots (k == blank)
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const int idx = lane + 64 * i;
        sr[i] = idx / C; sk[i] = idx - sr[i] * C;
        sx[i] = sk[i] == blank ? 0x7fffffff : idx;
    }
    const float invC = 1.0f / (float)C;
    const int nw = Tcap >> 2;                                          // words per prefix row
    const int li = lane & (FB - 1);                                    // the beam this lane speaks for (lanes >= FB: a valid alias)
    BEAM_WAVE_SYNC();

    int t = 0;
    while (t < tl && !(__builtin_amdgcn_readfirstlane(fl[t]) & 1)) t++;
    double pre = (t < tl && lane < C) ? lpw[((size_t)t * B + b) * C + lane] : 0.0;
    if (DBG) tstamp = __builtin_readcyclecounter();
    while (t < tl) {
        const bool rep_ok = (__builtin_amdgcn_readfirstlane(fl[t]) & 2) != 0;
        if (lane < C) lp[lane] = pre;
        int tn = t + 1;
        while (tn < tl && !(__builtin_amdgcn_readfirstlane(fl[tn]) & 1)) tn++;
        pre = (tn < tl && lane < C) ? lpw[((size_t)tn * B + b) * C + lane] : 0.0;
        BEAM_WAVE_SYNC();
        const unsigned int *srcw = reinterpret_cast<const unsigned int *>(pref + (size_t)cur * FB * Tcap);
        unsigned char *pnext = pref + (size_t)(cur ^ 1) * FB * Tcap;
        const int nslot = nb * C;

        FSTAMP(0);
        // ---- candidate scores (BeamSearch.py:53-69): gather every operand, then store.  errp = (order << 2 | kind) of
        // the first failure in the reference's execution order (kind 1: LM KeyError, 2: log(0) ValueError)
        int errp = 0x7fffffff;
        double val[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const int r = min(sr[i], nb - 1), k = sk[i];
            const int ylen = m_len[cur][r], ylast = m_last[cur][r];
            const double yB = m_B[cur][r], yT = m_T[cur][r];
            const double lmv = lmt[(ylen ? ylast : C) * C1 + k];   // consulted even when alpha == 0 (BeamSearch.py:57-60)
            const double lk = lp[k];
            const bool live = sx[i] < nslot;                             // an extension slot of a beam that exists
            const int kind = (lmv != lmv) ? 1 : ((lk == -INFINITY) ? 2 : 0);
            const int e = (live && kind) ? (((sr[i] * C1 + 1 + k) << 2) | kind) : 0x7fffffff;
            errp = min(errp, e);
            const double base = (ylen && ylast == k && rep_ok) ? yB : yT;   // :63-66
            val[i] = live ? lk + lmv * alpha + base : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) tot[lane + 64 * i] = val[i];        // copy slots (k == blank) are overwritten just below
        FSTAMP(1);
        // ---- copy path of beam `li` (:101-113), and the beam (if any) that holds its prefix minus the last id
        const int ylen = m_len[cur][li], ylast = m_last[cur][li];
        double pnb, pb, tc;
        int par = -1;
        {
            const double lpb = lp[blank], lpl = lp[max(ylast, 0)];
            const double mNB = m_NB[cur][li], mT = m_T[cur][li];
            const unsigned long long myph = m_phash[cur][li];
            i32x4 L4[FB / 4];
            u64x2 H2[FB / 2];
#pragma unroll
            for (int q = 0; q < FB / 4; q++) L4[q] = reinterpret_cast<const i32x4 *>(m_len[cur])[q];
#pragma unroll
            for (int q = 0; q < FB / 2; q++) H2[q] = reinterpret_cast<const u64x2 *>(m_hash[cur])[q];
            pnb = ylen > 0 ? mNB + lpl : LOG_ZERO;
            pb = mT + lpb;
            const bool bad = lpb == -INFINITY || (ylen > 0 && lpl == -INFINITY);
            if (lane < nb && bad) errp = min(errp, ((lane * C1) << 2) | 2);
            tc = log_add_prob(pb, pnb);
#pragma unroll
            for (int q = FB - 1; q >= 0; q--) {                          // the lowest matching q wins
                const bool hit = L4[q >> 2][q & 3] == ylen - 1 && H2[q >> 1][q & 1] == myph;   // entries >= nb hold length -2
                par = hit ? q : par;
            }
            par = (lane < nb && ylen > 0) ? par : -1;
            if (lane < nb) tot[lane * C + blank] = tc;
        }
        FSTAMP(2);
        // verify the content of every hash match with the whole wave (never merge on a hash collision): all pairs at once
        {
            const unsigned int pm = (unsigned int)__ballot(par >= 0);    // beams with a parent candidate (bits < FB)
            if (pm) {
                unsigned int neq = 0;
                for (int j0 = 0; j0 < nw; j0 += 64) {
                    const int j = min(j0 + lane, nw - 1);
                    unsigned int wa[FB], wq[FB];
#pragma unroll
                    for (int a = 0; a < FB; a++) {
                        const int qa = max(__builtin_amdgcn_readlane(par, a), 0);
                        wa[a] = srcw[a * nw + j];
                        wq[a] = srcw[qa * nw + j];
                    }
#pragma unroll
                    for (int a = 0; a < FB; a++) {
                        const int rem = __builtin_amdgcn_readlane(ylen, a) - 1 - 4 * j;      // bytes of the parent's prefix from word j on
                        const unsigned int m = rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
                        neq |= (((wa[a] ^ wq[a]) & m) != 0 ? 1u : 0u) << a;
                    }
                }
                unsigned int bad = 0;
#pragma unroll
                for (int a = 0; a < FB; a++) bad |= (__any((neq >> a) & 1) ? 1u : 0u) << a;
                if ((bad >> li) & 1) par = -1;
            }
        }
        FSTAMP(3);
        if (__any(errp != 0x7fffffff)) {   // first error in the reference's execution order wins
            int eo = errp;
            for (int o = 32; o > 0; o >>= 1) eo = min(eo, __shfl_xor(eo, o));
            err = (eo & 3) == 1 ? MDD_BEAM_KEY_ERROR : MDD_BEAM_VALUE_ERROR;
            break;
        }
        BEAM_WAVE_SYNC();
        FSTAMP(4);
        // ---- merges (see beam_kernel): a beam whose prefix is also reachable as parent + last id takes both paths
        int ms = -1;
        const bool merging = par >= 0;
        const int nmerge = __popcll(__ballot(merging));
        if (nmerge) {
            const int q = max(par, 0), e = q * C + max(ylast, 0);
            const double pr = tot[e];
            const bool fwd = q < lane;                                    // the extension was inserted before the copy
            const double nNB = log_add_prob(fwd ? pr : pnb, fwd ? pnb : pr);
            const double nT = log_add_prob(fwd ? pr : tc, fwd ? tc : pr);
            if (merging) {
                pnb = nNB; tc = nT;
                tot[fwd ? e : lane * C + blank] = nT;
                tot[fwd ? lane * C + blank : e] = -INFINITY;
                ms = fwd ? e : -1;
            }
        }
        if (lane < FB) { cNB[lane] = pnb; cB[lane] = pb; mslot[lane] = lane < nb ? ms : -1; }
        BEAM_WAVE_SYNC();
        FSTAMP(5);
        // ---- top-`keep` selection
        const int ncand = nslot - nmerge;
        const int keep = ncand < beam ? ncand : beam;
        double tv[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) tv[i] = tot[lane + 64 * i];     // slots >= nslot hold -inf
        double lmax = tv[0];
#pragma unroll
        for (int i = 1; i < NS; i++) lmax = fmax(lmax, tv[i]);
        // rank of the lane maximum = lane maxima strictly above it, counted against LDS broadcasts (two per read);
        // theta0 = the smallest lane maximum of rank < keep: at least `keep` slots are >= theta0
        cl_v[lane] = lmax;
        BEAM_WAVE_SYNC();
        int rk = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {   // two rounds of 16 reads, all of a round in flight before the first compare
            f64x2 sj[16];
#pragma unroll
            for (int j = 0; j < 16; j++) sj[j] = reinterpret_cast<const f64x2 *>(cl_v)[16 * h + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; j++) rk += (sj[j].x > lmax ? 1 : 0) + (sj[j].y > lmax ? 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned long long mk = 0;
        for (int r = keep - 1; r >= 0 && mk == 0; r--) mk = __ballot(rk == r);
        const int srcl = __ffsll((long long)mk) - 1;
        const double theta0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lmax), srcl),
                                               __builtin_amdgcn_readlane(__double2loint(lmax), srcl));
        BEAM_WAVE_SYNC();                                                // cl_v is rewritten by the compaction below
        FSTAMP(6);
        int nl = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const bool pred = tv[i] >= theta0;                           // theta0 is finite, slots >= nslot hold -inf
            const unsigned long long msk = __ballot(pred);
            if (msk) {
                if (pred) {
                    const int pos = nl + __popcll(msk & lt), k = sk[i];
                    cl_v[pos] = tv[i]; cl_x[pos] = lane + 64 * i;
                    cl_o[pos] = sr[i] * C + (k == blank ? 0 : (k < blank ? k + 1 : k));   // insertion order of the slot
                }
                nl += __popcll(msk);
            }
        }
        BEAM_WAVE_SYNC();
        FSTAMP(7);
        if (nl <= 64) {   // the usual case: one candidate per lane, ranked against v_readlane broadcasts
            const bool has = lane < nl;
            const double v_ld = cl_v[lane];                               // unconditional: three loads in flight, not three branches
            const int o_ld = cl_o[lane], xs = cl_x[lane];
            const double v = has ? v_ld : -INFINITY;
            const int o = has ? o_ld : 0x7fffffff;
            const int vlo = __double2loint(v), vhi = __double2hiint(v);
            int rank = 0;
            for (int f = 0; f < nl; f++) {
                const double vf = __hiloint2double(__builtin_amdgcn_readlane(vhi, f), __builtin_amdgcn_readlane(vlo, f));
                const int of = __builtin_amdgcn_readlane(o, f);
                rank += (vf > v || (vf == v && of < o)) ? 1 : 0;
            }
            if (has && rank < keep) { sel_v[rank] = v; sel_x[rank] = xs; }
        } else
        for (int e = lane; e < nl; e += 64) {
            const double v = cl_v[e];
            const int o = cl_o[e];
            int rank = 0;
#pragma unroll 4
            for (int f = 0; f < nl; f++) {
                const double vf = cl_v[f];
                const int of = cl_o[f];
                rank += (vf > v || (vf == v && of < o)) ? 1 : 0;
            }
            if (rank < keep) { sel_v[rank] = v; sel_x[rank] = cl_x[e]; }
        }
        BEAM_WAVE_SYNC();
        FSTAMP(8);
        if (DBG) ph[11] += nl;
        // ---- materialise new beam `li` (lanes >= keep compute on stale but valid entries and store nothing)
        const int nxt = cur ^ 1;
        int r_new, app_k, app_at;                                         // parent row; id appended (-1: none) and where
        {
            const int idx = sel_x[li];
            const double v = sel_v[li];
            i32x4 M4[FB / 4];
#pragma unroll
            for (int a = 0; a < FB / 4; a++) M4[a] = reinterpret_cast<const i32x4 *>(mslot)[a];
            const int r = min((int)(((float)idx + 0.5f) * invC), FB - 1), k = idx - r * C;
            const double cNBr = cNB[r], cBr = cB[r];
            const int len_r = m_len[cur][r], last_r = m_last[cur][r];
            const unsigned long long hash_r = m_hash[cur][r], phash_r = m_phash[cur][r];
            int fa = -1;                                                  // the beam merged into this extension, if any
#pragma unroll
            for (int a = 0; a < FB; a++) fa = M4[a >> 2][a & 3] == idx ? a : fa;   // entries >= nb hold -1
            const double cNBa = cNB[max(fa, 0)], cBa = cB[max(fa, 0)];
            const bool isb = k == blank;
            r_new = r; app_k = isb ? -1 : k; app_at = len_r;
            if (lane < FB) {   // entries keep..FB-1 get length -2: no prefix is their child (parent search above)
                const bool on = lane < keep;
                m_T[nxt][lane] = v;
                m_NB[nxt][lane] = isb ? cNBr : (fa >= 0 ? cNBa : v);
                m_B[nxt][lane] = isb ? cBr : (fa >= 0 ? cBa : LOG_ZERO);
                m_len[nxt][lane] = on ? (isb ? len_r : len_r + 1) : -2;
                m_last[nxt][lane] = on ? (isb ? last_r : k) : -1;
                m_hash[nxt][lane] = isb ? hash_r : hash_r * 0x9E3779B97F4A7C15ull + (unsigned long long)(k + 1);
                m_phash[nxt][lane] = isb ? phash_r : hash_r;
            }
        }
        FSTAMP(9);
        {   // prefixes: every new beam copies its parent's whole row (bytes past a prefix's length are never read),
            // one word per lane and row, all rows in flight together; the appended id goes in afterwards
            // (rows keep..FB-1 receive a copy of some valid row: harmless, and it keeps every load and store unconditional)
            int rr[FB];
#pragma unroll
            for (int i = 0; i < FB; i++) rr[i] = __builtin_amdgcn_readlane(r_new, i);
            unsigned int *dstw = reinterpret_cast<unsigned int *>(pnext);
            for (int j = lane; j < nw; j += 64) {
                unsigned int w[FB];
#pragma unroll
                for (int i = 0; i < FB; i++) w[i] = srcw[rr[i] * nw + j];
#pragma unroll
                for (int i = 0; i < FB; i++) dstw[i * nw + j] = w[i];
            }
            BEAM_WAVE_SYNC();
            if (lane < keep && app_k >= 0) pnext[(size_t)lane * Tcap + app_at] = (unsigned char)app_k;
        }
        BEAM_WAVE_SYNC();
        cur = nxt;
        nb = keep;
        t = tn;
        FSTAMP(10);
    }
    BEAM_WAVE_SYNC();
    if (lane == 0) {   // final: EOS LM term, length normalisation, first maximum (:130-148)
        int best = -1;
        double bestv = 0.0;
        for (int r = 0; r < nb && !err; r++) {
            const int ylen = m_len[cur][r];
            if (ylen == 0) { err = MDD_BEAM_INDEX_ERROR; break; }
            const double v = lmt[m_last[cur][r] * C1 + C];
            if (v != v) { err = MDD_BEAM_KEY_ERROR; break; }
            double pr = log_add_prob(LOG_ZERO, m_T[cur][r] + v * alpha);
            pr = pr * (1.0 / (double)(ylen ? ylen : 1));
            if (best < 0 || pr > bestv) { best = r; bestv = pr; }
        }
        status[b] = err;
        int n = 0;
        if (!err && best >= 0) {
            n = m_len[cur][best];
            const unsigned char *src = pref + (size_t)cur * FB * Tcap + (size_t)best * Tcap;
            for (int j = 0; j < n; j++) ids[(size_t)b * T + j] = src[j];
        }
        nids[b] = n;
        if (score) score[b] = err ? __builtin_nan("") : bestv;
        if (DBG) for (int i = 0; i < 12; i++) dbg[b * 12 + i] = ph[i];
    }
#undef FSTAMP
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
    hipStream_t st = (hipStream_t)stream;
    const int Tcap = (T + 1 + 3) & ~3;
    const size_t lm_bytes = sizeof(double) * (size_t)(C + 1) * (C + 1);
    const size_t base = (sizeof(double) * 2 + sizeof(int)) * (size_t)beam * C + (size_t)(2 * beam + 1) * Tcap;
    const int lm_in_lds = (base + lm_bytes <= 96 * 1024) ? 1 : 0;
    const size_t smem = base + (lm_in_lds ? lm_bytes : 0);
    if (smem > 140 * 1024) { set_error("mdd_beam: beam*C / T too large for LDS (%zu B)", smem); return MDD_ERR_ARG; }
    static bool attr_set = false;
    if (!attr_set) {
        MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    // stream-ordered scratch for the frame pre-pass: log-probabilities in fp64 + per-frame flags
    double *lpw = nullptr;
    unsigned char *flags = nullptr;
    const size_t n = (size_t)T * B * C;
    MDD_HIP_CHECK(hipMallocAsync((void **)&lpw, n * sizeof(double) + (size_t)T * B, st));
    flags = reinterpret_cast<unsigned char *>(lpw + n);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(beam_prep_kernel, dim3(grid), dim3(256), 0, st, logp_dev, T, B, C, blank, lpw, flags);
    const int nslots = beam * C;
    if (beam <= FB && C <= 64 && nslots <= 1024 && !getenv("MDD_BEAM_GENERIC")) {
        const int NS = nslots <= 512 ? 8 : 16;
        const size_t lmb = beam_fast_lm_bytes(C);
        const size_t pw = (beam_fast_wave_bytes(NS, beam, Tcap) + 15) & ~(size_t)15;
        const size_t lds_max = 160 * 1024;
        static bool fattr = false;
        if (!fattr) {
            MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_fast_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_fast_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_fast_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            MDD_HIP_CHECK(hipFuncSetAttribute((const void *)beam_fast_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            fattr = true;
        }
        if (lmb + pw > lds_max) { (void)hipFreeAsync(lpw, st); set_error("mdd_beam: T too long for LDS (%zu B)", lmb + pw); return MDD_ERR_ARG; }
        // waves per workgroup: as many as the LDS holds (<= 4); a small batch spreads out instead (one wave per CU is the
        // lowest latency when nothing else wants the CUs)
        int W = (int)((lds_max - lmb) / pw);
        W = W > 4 ? 4 : W;                                              // one wave per SIMD: the kernel is built for the 512-VGPR budget
        if (B <= 64) W = 1;
        if (const char *we = getenv("MDD_BEAM_W")) { const int w = atoi(we); if (w >= 1 && w <= W) W = w; }
        const size_t fs = lmb + (size_t)W * pw;
        const dim3 grid_f((B + W - 1) / W), block_f(64 * W);
        long long *fdbg = getenv("MDD_BEAM_DBG") ? reinterpret_cast<long long *>(score_dev) + B : nullptr;   // tools/beam_stamps.py
#define MDD_BEAM_FAST(NS_, DBG_) hipLaunchKernelGGL((beam_fast_kernel<NS_, DBG_>), grid_f, block_f, fs, st, lpw, flags, T, B, C, len_dev, beam, blank, \
                                                  lm_dev, lm_alpha, ids_dev, nids_dev, status_dev, score_dev, Tcap, fdbg)
        if (NS == 8) { if (fdbg) MDD_BEAM_FAST(8, true); else MDD_BEAM_FAST(8, false); }
        else { if (fdbg) MDD_BEAM_FAST(16, true); else MDD_BEAM_FAST(16, false); }
#undef MDD_BEAM_FAST
    } else
    hipLaunchKernelGGL(beam_kernel, dim3(B), dim3(64), smem, st, lpw, flags, T, B, C, len_dev, beam, blank, lm_dev, lm_alpha,
                       ids_dev, nids_dev, status_dev, score_dev, Tcap, lm_in_lds, getenv("MDD_BEAM_SKIP") ? atoi(getenv("MDD_BEAM_SKIP")) : 0,
                       getenv("MDD_BEAM_DBG") ? reinterpret_cast<long long *>(score_dev) + B : nullptr);
    hipError_t le = hipGetLastError();
    (void)hipFreeAsync(lpw, st);
    if (le != hipSuccess) { set_error("beam kernel launch failed: %s", hipGetErrorString(le)); return MDD_ERR_HIP; }
    return MDD_OK;
}
