/*
 * mdd_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (libmdd_hip.so) never links, loads or calls it and has no
 * CPU fallback.
 *
 * Parity is PINNED: every function below is checked in tests/test_oracle.py against
 * golden vectors produced by importing the reference's own Python in the build
 * container (oracle/gen_golden.py; fixtures in tests/golden/).
 *
 * Plain C99.  Each function cites the reference lines it follows
 * (AA = /root/reference/egs/attention_aug).  Dot products accumulate in double and
 * are rounded once to float: the reference computes them in fp32 through ATen (order
 * unspecified), so a double accumulator puts the oracle within one fp32 rounding of
 * the exact value; measured |oracle - reference| on the goldens is a few 1e-6.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LOG_ZERO (-99999999.0) /* AA/utils/BeamSearch.py:6 */

/* ------------------------------------------------------------------ A1: stack + skip + even pad
 * AA/utils/tools.py:207-227 (make_context(feat,0,right), skip_feat), AA/utils/data_loader.py:138-142.
 * raw [T,D] -> out [T_out,(right+1)*D]; frame t stacks raw[t], raw[t+1], .. with the last frame
 * replicated past the edge; keep frames 0,skip,2*skip..; zero-pad rows to a multiple of n_down. */
int orc_stack_len(int T, int skip, int n_down) {
    int kept = (skip <= 1) ? T : (T + skip - 1) / skip;
    if (n_down > 1 && kept % n_down) kept += n_down - kept % n_down;
    return kept;
}

void orc_stack_skip(const float *raw, int T, int D, int right, int skip, int n_down, float *out) {
    int W = (right + 1) * D, kept = (skip <= 1) ? T : (T + skip - 1) / skip;
    int Tout = orc_stack_len(T, skip, n_down);
    if (skip < 1) skip = 1;
    for (int i = 0; i < Tout; i++) {
        float *o = out + (size_t)i * W;
        if (i >= kept) { memset(o, 0, sizeof(float) * W); continue; }
        int t = i * skip;
        for (int r = 0; r <= right; r++) {
            int s = t + r; if (s > T - 1) s = T - 1;
            memcpy(o + r * D, raw + (size_t)s * D, sizeof(float) * D);
        }
    }
}

/* float32 length bookkeeping: frac = len/maxlen in f32 (data_loader.py:177), frames = (frac*T_out).long()
 * (AA/infer.py:296-297, AA/steps/train_ctc.py:68) -- f32 multiply, truncation toward zero. */
float orc_len_frac(int len, int maxlen) { return (float)((double)len / (double)maxlen); }
int orc_len_frames(float frac, int t_out) { volatile float p = frac * (float)t_out; return (int)p; }

/* ------------------------------------------------------------------ A2: Conv2d(k3,pad1)+bias -> BN2d(eval) -> ReLU
 * AA/models/model_ctc.py:73-81 (LayerCNN.forward), built :105-129.  NCHW. */
void orc_conv_bn_relu(const float *x, int B, int Cin, int Hin, int Win, const float *w, const float *bias,
                      const float *bn_w, const float *bn_b, const float *bn_m, const float *bn_v, float eps,
                      int Cout, int sh, int sw, float *y) {
    int Hout = (Hin + 2 - 3) / sh + 1, Wout = (Win + 2 - 3) / sw + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Cout; co++) {
            float scale = bn_w[co] / sqrtf(bn_v[co] + eps);
            float shift = bn_b[co] - bn_m[co] * scale;
            for (int ho = 0; ho < Hout; ho++)
                for (int wo = 0; wo < Wout; wo++) {
                    double acc = 0.0;
                    for (int ci = 0; ci < Cin; ci++)
                        for (int kh = 0; kh < 3; kh++) {
                            int hi = ho * sh + kh - 1;
                            if (hi < 0 || hi >= Hin) continue;
                            for (int kw = 0; kw < 3; kw++) {
                                int wi = wo * sw + kw - 1;
                                if (wi < 0 || wi >= Win) continue;
                                acc += (double)x[(((size_t)b * Cin + ci) * Hin + hi) * Win + wi] *
                                       (double)w[((co * Cin + ci) * 3 + kh) * 3 + kw];
                            }
                        }
                    float v = (float)acc + bias[co];
                    v = v * scale + shift;
                    y[(((size_t)b * Cout + co) * Hout + ho) * Wout + wo] = v > 0.f ? v : 0.f;
                }
        }
}

/* A3: [B,C,T',W] -> [T',B,C*W]  (model_ctc.py:176-181) */
void orc_cnn_to_seq(const float *x, int B, int C, int T, int W, float *y) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int t = 0; t < T; t++)
                memcpy(y + ((size_t)t * B + b) * C * W + (size_t)c * W, x + (((size_t)b * C + c) * T + t) * W,
                       sizeof(float) * W);
}

/* BatchNorm1d eval as per-feature affine over rows (model_ctc.py:41-43, :153-154) */
void orc_bn_rows(const float *x, size_t rows, int F, const float *bn_w, const float *bn_b, const float *bn_m,
                 const float *bn_v, float eps, float *y) {
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++)
        for (int f = 0; f < F; f++) {
            float scale = bn_w[f] / sqrtf(bn_v[f] + eps);
            y[r * F + f] = x[r * F + f] * scale + (bn_b[f] - bn_m[f] * scale);
        }
}

/* y[M,N] = x[M,K] . W[N,K]^T (+ b) */
void orc_linear(const float *x, size_t M, int K, const float *W, const float *b, int N, float *y) {
#pragma omp parallel for schedule(static)
    for (size_t m = 0; m < M; m++)
        for (int n = 0; n < N; n++) {
            double acc = 0.0;
            const float *xr = x + m * K, *wr = W + (size_t)n * K;
            for (int k = 0; k < K; k++) acc += (double)xr[k] * (double)wr[k];
            y[m * N + n] = (float)acc + (b ? b[n] : 0.f);
        }
}

static float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

/* ------------------------------------------------------------------ A4/A5: one LSTM direction
 * torch.nn.LSTM semantics as called at model_ctc.py:28-29,44 (bias=False) and :150,198 (bias=True):
 * gate order i,f,g,o; zero initial state; the whole padded length T is processed; the reverse
 * direction starts at t=T-1 (inside the padding).  x [T,B,I] -> out[t,b, off : off+H] with row
 * stride ldo. */
void orc_lstm_dir(const float *x, int T, int B, int I, const float *W_ih, const float *W_hh, const float *b_ih,
                  const float *b_hh, int H, int reverse, float *out, int ldo, int off) {
    int G = 4 * H;
    float *gx = (float *)malloc(sizeof(float) * (size_t)T * B * G);
    float *h = (float *)calloc((size_t)B * H, sizeof(float));
    float *c = (float *)calloc((size_t)B * H, sizeof(float));
    float *g = (float *)malloc(sizeof(float) * (size_t)B * G);
    orc_linear(x, (size_t)T * B, I, W_ih, b_ih, G, gx);
    for (int s = 0; s < T; s++) {
        int t = reverse ? T - 1 - s : s;
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++)
            for (int n = 0; n < G; n++) {
                double acc = 0.0;
                const float *hr = h + (size_t)b * H, *wr = W_hh + (size_t)n * H;
                for (int k = 0; k < H; k++) acc += (double)hr[k] * (double)wr[k];
                g[(size_t)b * G + n] = gx[((size_t)t * B + b) * G + n] + (float)acc + (b_hh ? b_hh[n] : 0.f);
            }
        for (int b = 0; b < B; b++)
            for (int u = 0; u < H; u++) {
                const float *gb = g + (size_t)b * G;
                float ig = sigmoidf_(gb[u]), fg = sigmoidf_(gb[H + u]), gg = tanhf(gb[2 * H + u]),
                      og = sigmoidf_(gb[3 * H + u]);
                float cn = fg * c[(size_t)b * H + u] + ig * gg;
                float hn = og * tanhf(cn);
                c[(size_t)b * H + u] = cn;
                h[(size_t)b * H + u] = hn;
                out[((size_t)t * B + b) * ldo + off + u] = hn;
            }
    }
    free(gx); free(h); free(c); free(g);
}

/* A5: nn.Embedding lookup (model_ctc.py:193); ids [n] -> out [n,E] */
int orc_embed(const float *table, int rows, int E, const int64_t *ids, size_t n, float *out) {
    for (size_t i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= rows) return -1; /* reference: IndexError */
        memcpy(out + i * E, table + (size_t)ids[i] * E, sizeof(float) * E);
    }
    return 0;
}

/* ------------------------------------------------------------------ A6: attention (model_ctc.py:204-211)
 * S = X.key^T (no scale, no mask) ; A = softmax_L(S) ; ctx = A.val ; out = cat(X, ctx).
 * X [B,T,D], key/val [B,L,D] -> out [B,T,2D] */
void orc_attention(const float *X, const float *key, const float *val, int B, int T, int L, int D, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            const float *q = X + ((size_t)b * T + t) * D;
            float *o = out + ((size_t)b * T + t) * 2 * D;
            float s[1024];
            float mx = -INFINITY;
            for (int l = 0; l < L; l++) {
                double acc = 0.0;
                const float *kr = key + ((size_t)b * L + l) * D;
                for (int d = 0; d < D; d++) acc += (double)q[d] * (double)kr[d];
                s[l] = (float)acc;
                if (s[l] > mx) mx = s[l];
            }
            double den = 0.0;
            for (int l = 0; l < L; l++) { s[l] = expf(s[l] - mx); den += s[l]; }
            for (int l = 0; l < L; l++) s[l] = (float)(s[l] / den);
            memcpy(o, q, sizeof(float) * D);
            for (int d = 0; d < D; d++) {
                double acc = 0.0;
                for (int l = 0; l < L; l++) acc += (double)s[l] * (double)val[((size_t)b * L + l) * D + d];
                o[D + d] = (float)acc;
            }
        }
}

/* A7: log_softmax over the last dim (model_ctc.py:218) */
void orc_log_softmax(float *x, size_t rows, int C) {
    for (size_t r = 0; r < rows; r++) {
        float *p = x + r * C, mx = p[0];
        for (int c = 1; c < C; c++) if (p[c] > mx) mx = p[c];
        double s = 0.0;
        for (int c = 0; c < C; c++) s += exp((double)p[c] - (double)mx);
        float lse = (float)log(s);
        for (int c = 0; c < C; c++) p[c] = (p[c] - mx) - lse;
    }
}

/* ------------------------------------------------------------------ A8: greedy decode
 * AA/utils/ctcDecoder.py:188-200 + :80-92: argmax per frame (first index wins ties, torch.max on CPU),
 * first len[b] frames, drop an element equal to its immediate predecessor (blank included in the
 * comparison), drop blanks.  logp [T,B,C]; ids_out [B,T]; returns 0. */
int orc_greedy(const float *logp, int T, int B, int C, const int32_t *len, int blank, int32_t *ids_out,
               int32_t *nids) {
    for (int b = 0; b < B; b++) {
        int n = 0, prev = -1;
        for (int t = 0; t < len[b] && t < T; t++) {
            const float *p = logp + ((size_t)t * B + b) * C;
            int am = 0;
            for (int c = 1; c < C; c++) if (p[c] > p[am]) am = c;
            if (am != blank && !(t != 0 && am == prev)) ids_out[(size_t)b * T + n++] = am;
            prev = am;
        }
        nids[b] = n;
    }
    return 0;
}

/* ------------------------------------------------------------------ A9: CTC prefix beam search
 * AA/utils/ctcDecoder.py:215-226 (exp in f32 first) -> AA/utils/BeamSearch.py:73-153.
 * Scores are double; probabilities are float32 exp(logp) widened to double before log().
 * lm: dense (C+1)x(C+1) table of ln-probabilities T[prev][next] (prev==C: sentence start,
 * next==C: sentence end; AA/utils/NgramLM.py:65-78), NaN where the reference raises KeyError.
 * status[b]: 0 ok, 1 IndexError (empty prefix among the final beams, BeamSearch.py:135),
 * 2 ValueError (log(0), :64,66,103,106), 3 KeyError (LM lookup, NgramLM.py:75-76).
 * The first error in the reference's execution order wins. */
static double log_add_prob(double lx, double ly) { /* BeamSearch.py:43-50 */
    if (lx <= LOG_ZERO) return ly;
    if (ly <= LOG_ZERO) return lx;
    if ((ly - lx) > 0.0) { double t = lx; lx = ly; ly = t; }
    return lx + log(1 + exp(ly - lx));
}

typedef struct { double prTotal, prNonBlank, prBlank; int len; int32_t *y; } beam_entry;
typedef struct { double prTotal, prNonBlank, prBlank; int src, sym; } cand_t;

static int prefix_is_parent(const beam_entry *p, const beam_entry *c) { /* c.y == p.y + (c.y[-1],) */
    if (c->len != p->len + 1) return 0;
    return memcmp(p->y, c->y, sizeof(int32_t) * p->len) == 0;
}

int orc_beam(const float *logp, int T, int B, int C, const int32_t *len, int beam, int blank, const double *lm,
             double alpha, int32_t *ids_out, int32_t *nids, int32_t *status, double *score_out) {
    int maxcand = beam * C;
    beam_entry *last = (beam_entry *)calloc(beam, sizeof(beam_entry));
    beam_entry *next = (beam_entry *)calloc(beam, sizeof(beam_entry));
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * maxcand);
    int *slot_of = (int *)malloc(sizeof(int) * maxcand); /* (rank, j) -> candidate index */
    int *order = (int *)malloc(sizeof(int) * maxcand);
    float *pm = (float *)malloc(sizeof(float) * C), *pprev = (float *)malloc(sizeof(float) * C);
    for (int i = 0; i < beam; i++) {
        last[i].y = (int32_t *)malloc(sizeof(int32_t) * (T + 1));
        next[i].y = (int32_t *)malloc(sizeof(int32_t) * (T + 1));
    }
    for (int b = 0; b < B; b++) {
        int nlast = 1, err = 0;
        last[0].len = 0; last[0].prBlank = 0.0; last[0].prTotal = 0.0; last[0].prNonBlank = LOG_ZERO;
        int tl = len[b] < T ? len[b] : T;
        for (int t = 0; t < tl && !err; t++) {
            const float *row = logp + ((size_t)t * B + b) * C;
            for (int c = 0; c < C; c++) pm[c] = expf(row[c]);              /* torch.exp, f32 (ctcDecoder.py:224) */
            if ((1.0f - pm[blank]) < 0.1f) continue;                        /* :93-94, float32 compare */
            if (t > 0) {
                const float *rp = logp + ((size_t)(t - 1) * B + b) * C;     /* raw previous row, even if skipped */
                pprev[blank] = expf(rp[blank]);
            }
            /* `last` already holds the top-`beam` entries in stable rank order (see selection below) */
            int ncand = 0;
            for (int r = 0; r < nlast && !err; r++) {
                beam_entry *y = &last[r];
                /* was y already inserted as an extension of a higher-ranked beam? */
                int merged = -1;
                if (y->len > 0)
                    for (int q = 0; q < r; q++)
                        if (prefix_is_parent(&last[q], y)) merged = slot_of[q * C + y->y[y->len - 1]];
                double prNonBlank = LOG_ZERO;
                if (y->len > 0) {
                    float p = pm[y->y[y->len - 1]];
                    if (p == 0.0f) { err = 2; break; }
                    prNonBlank = y->prNonBlank + log((double)p);            /* :103 */
                }
                if (pm[blank] == 0.0f) { err = 2; break; }
                double prBlank = y->prTotal + log((double)pm[blank]);       /* :106 */
                cand_t *e;
                if (merged >= 0) { e = &cand[merged]; slot_of[r * C + blank] = merged; }
                else {
                    e = &cand[ncand]; slot_of[r * C + blank] = ncand++;
                    e->prTotal = e->prNonBlank = e->prBlank = LOG_ZERO; e->src = r; e->sym = -1;
                }
                e->prNonBlank = log_add_prob(e->prNonBlank, prNonBlank);    /* :110 */
                e->prBlank = log_add_prob(e->prBlank, prBlank);             /* :111 */
                double prTotal = log_add_prob(prBlank, prNonBlank);         /* :112 */
                e->prTotal = log_add_prob(e->prTotal, prTotal);             /* :113 */
                for (int k = 0; k < C && !err; k++) {
                    if (k == blank) continue;
                    double bigram = 0.0;
                    {   /* calcExtPr, :52-66; the LM is consulted even when alpha == 0 */
                        int c1 = y->len ? y->y[y->len - 1] : C;
                        double v = lm[(size_t)c1 * (C + 1) + k];
                        if (isnan(v)) { err = 3; break; }
                        bigram = v * alpha;
                    }
                    if (pm[k] == 0.0f) { err = 2; break; }
                    double pr;
                    if (y->len && y->y[y->len - 1] == k && pprev[blank] < 0.9f)
                        pr = log((double)pm[k]) + bigram + y->prBlank;
                    else
                        pr = log((double)pm[k]) + bigram + y->prTotal;
                    /* does newY equal an already-inserted entry (a higher-ranked beam's own copy)? */
                    int tgt = -1;
                    for (int q = 0; q < r; q++)
                        if (last[q].len == y->len + 1 && last[q].y[y->len] == k && prefix_is_parent(y, &last[q]))
                            tgt = slot_of[q * C + blank];
                    cand_t *x;
                    if (tgt >= 0) x = &cand[tgt];
                    else {
                        x = &cand[ncand]; slot_of[r * C + k] = ncand++;
                        x->prTotal = x->prNonBlank = x->prBlank = LOG_ZERO; x->src = r; x->sym = k;
                    }
                    x->prNonBlank = log_add_prob(x->prNonBlank, pr);         /* :124 */
                    x->prTotal = log_add_prob(x->prTotal, pr);               /* :125 */
                }
            }
            if (err) break;
            /* last = curr, then `sort()[0:beam]` at the next use: stable descending by prTotal (:29-33,96) */
            int keep = ncand < beam ? ncand : beam;
            for (int i = 0; i < ncand; i++) order[i] = i;
            for (int i = 0; i < keep; i++) { /* stable selection: first maximal element wins */
                int best = i;
                for (int j = i + 1; j < ncand; j++)
                    if (cand[order[j]].prTotal > cand[order[best]].prTotal) best = j;
                int o = order[best];
                memmove(order + i + 1, order + i, sizeof(int) * (best - i));
                order[i] = o;
            }
            for (int i = 0; i < keep; i++) {
                cand_t *e = &cand[order[i]];
                beam_entry *src = &last[e->src];
                memcpy(next[i].y, src->y, sizeof(int32_t) * src->len);
                next[i].len = src->len;
                if (e->sym >= 0) next[i].y[next[i].len++] = e->sym;
                next[i].prTotal = e->prTotal; next[i].prNonBlank = e->prNonBlank; next[i].prBlank = e->prBlank;
            }
            beam_entry *tmp = last; last = next; next = tmp;
            nlast = keep;
        }
        /* final: EOS LM term, length normalisation, best (:130-148) */
        int best = -1; double bestv = 0.0;
        for (int r = 0; r < nlast && !err; r++) {
            beam_entry *y = &last[r];
            if (y->len == 0) { err = 1; break; }                            /* y[-1] on () -> IndexError */
            double v = lm[(size_t)y->y[y->len - 1] * (C + 1) + C];
            if (isnan(v)) { err = 3; break; }
            double pr = log_add_prob(LOG_ZERO, y->prTotal + v * alpha);     /* :137,141 */
            pr = pr * (1.0 / (double)(y->len ? y->len : 1));                /* norm(), :23-27 */
            if (best < 0 || pr > bestv) { best = r; bestv = pr; }
        }
        status[b] = err;
        nids[b] = 0;
        if (score_out) score_out[b] = err ? NAN : bestv;
        if (!err && best >= 0) {
            nids[b] = last[best].len;
            memcpy(ids_out + (size_t)b * T, last[best].y, sizeof(int32_t) * last[best].len);
        }
    }
    for (int i = 0; i < beam; i++) { free(last[i].y); free(next[i].y); }
    free(last); free(next); free(cand); free(slot_of); free(order); free(pm); free(pprev);
    return 0;
}

/* ------------------------------------------------------------------ A10: Levenshtein + backtrace
 * AA/utils/ctcDecoder.py:134-184.  a = hypothesis (s1), b = canonical (s2).
 * ops: 0 '-', 1 'S', 2 'I', 3 'D'.  Returns -1 when either side is empty (reference: TypeError). */
int orc_align(const int32_t *a, int na, const int32_t *b, int nb, int32_t *dist_out, uint8_t *ops, int32_t *nops) {
    if (na == 0 || nb == 0) return -1;
    int W = nb + 1;
    int *d = (int *)malloc(sizeof(int) * (size_t)(na + 1) * W);
    for (int j = 0; j <= nb; j++) d[j] = j;
    for (int i = 1; i <= na; i++) d[i * W] = i;
    for (int i = 1; i <= na; i++)
        for (int j = 1; j <= nb; j++) {
            int cost = a[i - 1] == b[j - 1] ? 0 : 1;
            int m = d[i * W + j - 1] + 1;
            if (d[(i - 1) * W + j] + 1 < m) m = d[(i - 1) * W + j] + 1;
            if (d[(i - 1) * W + j - 1] + cost < m) m = d[(i - 1) * W + j - 1] + cost;
            d[i * W + j] = m;
        }
    *dist_out = d[na * W + nb];
    int i = na, j = nb, n = 0;
    while (i > 0 || j > 0) {
        if (i == 0) { ops[n++] = 3; j--; }
        else if (j == 0) { ops[n++] = 2; i--; }
        else if (a[i - 1] == b[j - 1]) { ops[n++] = 0; i--; j--; }
        else if (d[i * W + j] == d[(i - 1) * W + j - 1] + 1) { ops[n++] = 1; i--; j--; }
        else if (d[i * W + j] == d[(i - 1) * W + j] + 1) { ops[n++] = 2; i--; }
        else if (d[i * W + j] == d[i * W + j - 1] + 1) { ops[n++] = 3; j--; }
    }
    for (int k = 0; k < n / 2; k++) { uint8_t t = ops[k]; ops[k] = ops[n - 1 - k]; ops[n - 1 - k] = t; }
    *nops = n;
    free(d);
    return 0;
}

/* ------------------------------------------------------------------ A12: CTC loss (alpha/beta lattice)
 * torch.nn.CTCLoss(reduction='sum'), blank=0, padded 2-D targets, as called at
 * AA/steps/train_ctc.py:72,186.  The arithmetic is ATen's (third-party, not under /root/reference);
 * restated from the published algorithm (Graves et al. 2006, eq. 6-8, 10-11, 16) in the
 * form ATen uses: nll[b] = -logsumexp(alpha_T-1(S-1), alpha_T-1(S-2)); the tensor autograd deposits on
 * log_probs is grad[t,b,c] = exp(lp) - exp(logsumexp_{s:l'_s=c}(alpha_t(s)+beta_t(s)) + nll - lp) for
 * t < in_len[b], zero for padded frames.  The lattice is kept in double here (ATen keeps it in
 * fp32, whose rounding at nll ~ 1e2..1e3 is ~1e-5..1e-4 on the gradient).  Pinned by G5 goldens. */
static double lse2d(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}
static double lse3d(double a, double b, double c) {
    double m = a > b ? a : b; if (c > m) m = c;
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

int orc_ctc_loss(const float *logp, int T, int B, int C, const int64_t *targets, int Lmax, const int64_t *in_len,
                 const int64_t *tgt_len, int blank, float *nll, float *grad) {
    int Smax = 2 * Lmax + 1;
    double *al = (double *)malloc(sizeof(double) * (size_t)T * Smax);
    double *be = (double *)malloc(sizeof(double) * (size_t)T * Smax);
    double *acc = (double *)malloc(sizeof(double) * C);
    if (grad) memset(grad, 0, sizeof(float) * (size_t)T * B * C);
    for (int b = 0; b < B; b++) {
        int Tb = (int)in_len[b], L = (int)tgt_len[b], S = 2 * L + 1;
        const int64_t *tg = targets + (size_t)b * Lmax;
#define LP(t, c) logp[((size_t)(t) * B + b) * C + (c)]
#define LAB(s) (((s) & 1) ? (int)tg[(s) >> 1] : blank)
        for (int i = 0; i < T * Smax; i++) al[i] = be[i] = -INFINITY;
        if (Tb <= 0) { nll[b] = (L == 0) ? 0.f : INFINITY; continue; }
        al[0] = LP(0, blank);
        if (S > 1) al[1] = LP(0, LAB(1));
        for (int t = 1; t < Tb; t++)
            for (int s = 0; s < S; s++) {
                double a0 = al[(t - 1) * Smax + s];
                double a1 = s >= 1 ? al[(t - 1) * Smax + s - 1] : -INFINITY;
                double a2 = (s >= 2 && LAB(s) != blank && LAB(s) != LAB(s - 2)) ? al[(t - 1) * Smax + s - 2] : -INFINITY;
                double m = lse3d(a0, a1, a2);
                al[t * Smax + s] = (m == -INFINITY) ? -INFINITY : m + LP(t, LAB(s));
            }
        double l1 = al[(Tb - 1) * Smax + S - 1], l2 = S > 1 ? al[(Tb - 1) * Smax + S - 2] : -INFINITY;
        double ll = lse2d(l1, l2);
        nll[b] = (float)-ll;
        if (!grad) continue;
        be[(Tb - 1) * Smax + S - 1] = LP(Tb - 1, blank);
        if (S > 1) be[(Tb - 1) * Smax + S - 2] = LP(Tb - 1, LAB(S - 2));
        for (int t = Tb - 2; t >= 0; t--)
            for (int s = 0; s < S; s++) {
                double b0 = be[(t + 1) * Smax + s];
                double b1 = s + 1 < S ? be[(t + 1) * Smax + s + 1] : -INFINITY;
                double b2 = (s + 2 < S && LAB(s) != blank && LAB(s) != LAB(s + 2)) ? be[(t + 1) * Smax + s + 2] : -INFINITY;
                double m = lse3d(b0, b1, b2);
                be[t * Smax + s] = (m == -INFINITY) ? -INFINITY : m + LP(t, LAB(s));
            }
        for (int t = 0; t < Tb; t++) {
            float *g = grad + ((size_t)t * B + b) * C;
            for (int c = 0; c < C; c++) acc[c] = -INFINITY;
            for (int s = 0; s < S; s++) {
                int c = LAB(s);
                acc[c] = lse2d(acc[c], al[t * Smax + s] + be[t * Smax + s]);
            }
            for (int c = 0; c < C; c++) {
                double lp = LP(t, c);
                g[c] = (float)(exp(lp) - exp(acc[c] - ll - lp));
            }
        }
#undef LP
#undef LAB
    }
    free(al); free(be); free(acc);
    return 0;
}
