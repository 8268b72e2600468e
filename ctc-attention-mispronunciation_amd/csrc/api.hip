// C-ABI of libmdd_hip.so: handle, weights, workspace, forward orchestration (include/mdd_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "mdd_internal.h"

namespace mdd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Per-device gate for forwards that contain persistent BiLSTM launches.  A persistent layer kernel needs all of its
// 256 workgroups resident (one per CU); two of them in flight on one device -- two handles, two host threads, two
// streams -- would each hold CUs the other is waiting for until the wall-clock abort.  So every such forward is
// ordered behind the previous one on the same device: the new stream waits for an event recorded at the end of the
// previous forward (a device-side dependency, the host never blocks).  Process-wide; one process per device is the
// deployment model (a second PROCESS on the same device is outside this gate: use MDD_LSTM=step there).
struct DeviceGate {
    std::mutex mu;
    hipEvent_t done = nullptr;        // recorded behind the last gated forward
    hipStream_t stream = nullptr;     // the stream it was recorded on
    bool armed = false;
};
static DeviceGate g_gate[64];

struct DevBuf {
    float *p = nullptr;
    size_t cap = 0;  // in floats
};

struct GraphKey {
    const void *x, *x1, *out, *tlen, *llen;
    int B, T, L, Traw;
    bool operator<(const GraphKey &o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};

}  // namespace mdd

struct mdd_model {
    mdd_config cfg;
    int device = 0;
    bool finalized = false, taps = false, use_graph = true;
    int precision = 2;   // 2 (default): fp32-grade, the large contractions as f32x6 on the bf16 matrix cores (falls back to 0 when the geometry does not allow);
                         // 0: exact fp32 MFMA everywhere; 1: split-bf16 x3 for every contraction (narrower than fp32: flagged variant)
    std::map<std::string, std::vector<float>> host;  // state_dict entries as loaded
    // device weights
    float *w_conv0 = nullptr, *sc0 = nullptr, *sh0 = nullptr;
    float *w_conv1t = nullptr, *sc1 = nullptr, *sh1 = nullptr;
    std::vector<float *> wih, whh, bn_scale, bn_shift;  // per rnn layer (bn_* of layer n applies to layer n's INPUT)
    float *emb = nullptr, *t_wih = nullptr, *t_whh = nullptr, *t_bias = nullptr;
    float *w_score = nullptr, *fscale = nullptr, *fshift = nullptr, *w_fc = nullptr, *w_fcp = nullptr;
    std::vector<mdd::SplitPtr> wih_s, whh_s;                // split-bf16 copies of the GEMM / recurrent weights
    std::vector<unsigned short *> wih_3;                    // three-plane (f32x6) copies of the input-projection weights, K-tile-major
    unsigned short *t_wih_3 = nullptr, *w_conv1_3 = nullptr; // (conv1 weights [co][kh][kw][ci] as three row-major planes)
    std::vector<unsigned short *> whh_3;                    // Whh' [3][2][4H][H]: three row-major planes of the recurrent weights (f32x6 layer kernel)
    unsigned short *t_whh_3 = nullptr;
    mdd::SplitPtr t_whh_s{nullptr, nullptr};
    mdd::SplitPtr t_wih_s{nullptr, nullptr}, w_score_s{nullptr, nullptr}, w_conv1_s{nullptr, nullptr};
    std::vector<void *> owned;  // every hipMalloc'd weight pointer
    // workspace
    mdd::DevBuf y0, seq0, gx, act[2], xraw, hbuf, cbuf, embo, text, key, S;
    mdd::DevBuf seq0_s, act_s[2], x_s, embo_s, text_s, key_s, hsplit, hx;   // split-bf16 activations (hi plane, then lo plane)
    mdd::DevBuf p3;             // f32x6 mode: the three bf16 planes of the projection GEMM's A operand (rewritten per GEMM)
    std::vector<mdd::DevBuf> tap_rnn;
    int *err_flag = nullptr;
    hipStream_t cap_stream = nullptr;  // graphs are captured here (the legacy default stream cannot capture)
    int lastB = 0, lastT = 0, lastL = 0;
    int raw_T = 0;              // > 0 while mdd_forward_raw runs the fused front-end straight on unstacked frames
    const int *tlen = nullptr, *llen = nullptr;   // set while mdd_forward_fused runs: per-row posterior frames / canonical length of the row's own batch
    mdd::DevBuf xstack;         // mdd_forward_raw without the fused front-end: stacked copy
    std::map<mdd::GraphKey, hipGraphExec_t> graphs;
    int W1() const { return (cfg.feat + 2 - 3) / 2 + 1; }
    int W2() const { return (W1() + 2 - 3) / 2 + 1; }
    int rnn_in() const { return cfg.channels * W2(); }
    int granule_max_b = 1024;   // batch rows the persistent kernel's teams cover (4 row tiles x 16 rows x 16 groups)
    bool lstm_persist = true;   // one persistent team-synchronised launch per BiLSTM layer (split-bf16 mode, >= 256 CUs, B <= 1024)
    int n_cu = 0;
    unsigned int *sync_words = nullptr;
    bool lstm_x3 = false;   // MDD_LSTM=x3: LDS-tiled split-bf16 step kernel (measured slower than the packed fp32 step; kept for study)
    bool conv_fused() const { return (x3() || x6()) && cfg.feat == 243 && cfg.channels == 32; }
    bool packed_h() const { return cfg.hidden == 384 || cfg.hidden == 256; }
    // one persistent launch per BiLSTM layer: the split-bf16 teams (lstm.hip) in mode 1, the exact-fp32 teams (lstm_f32.hip) in mode 0
    bool persist(int B) const { return lstm_persist && !lstm_x3 && packed_h() && B <= granule_max_b && (x3() || lstm_persist_f32); }
    bool lstm_persist_f32 = true;
    bool lstm_persist_x6 = true;     // MDD_LSTM_X6=0: mode 2 runs the exact-fp32 layer kernel instead (diagnostic); =force: lstm_x6.hip wherever it can run
    bool lstm_x6_force = false;
    // the f32x6 recurrence (lstm_x6.hip) where it is the faster of the two reference-width layer kernels (tools/lstm_kernel_choice.py,
    // profiles/round3_lstm_x6_notes.txt): at H = 384 for every batch size (0.60 - 0.93 of the exact-fp32 kernel's time), at H = 256 up to
    // 128 rows (0.83; beyond, the fp32 kernel's shorter products win: 1.07 - 1.5)
    bool lx6(int B) const {
        return x6() && lstm_persist_x6 && persist(B) && B <= mdd::lstm_x6_max_b(cfg.hidden) && (lstm_x6_force || cfg.hidden == 384 || B <= 128);
    }
    bool x6() const {   // f32x6: the time-batched input projections on the bf16 matrix cores with three planes per operand; all else as mode 0
        return precision == 2 && rnn_in() % 32 == 0 && (2 * cfg.hidden) % 32 == 0 && cfg.emb_dim % 32 == 0;
    }
    bool x3() const {   // the bf16x3 GEMM needs K % 32 == 0 for every contraction and the packed LSTM layouts
        return precision == 1 && (cfg.hidden == 384 || cfg.hidden == 256) && rnn_in() % 32 == 0 && cfg.emb_dim % 32 == 0;
    }
};

namespace mdd {

static int upload(mdd_model *m, const std::vector<float> &h, float **dev) {
    MDD_HIP_CHECK(hipMalloc((void **)dev, h.size() * sizeof(float)));
    m->owned.push_back(*dev);
    MDD_HIP_CHECK(hipMemcpy(*dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return MDD_OK;
}

static unsigned short host_bf16(float x) {   // round-to-nearest-even (weights are finite)
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float host_bf16_f32(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

static int upload_split(mdd_model *m, const std::vector<float> &h, SplitPtr *out) {
    const size_t n = h.size();
    std::vector<unsigned short> buf(2 * n);
    for (size_t i = 0; i < n; i++) {
        buf[i] = host_bf16(h[i]);
        buf[n + i] = host_bf16(h[i] - host_bf16_f32(buf[i]));
    }
    unsigned short *d = nullptr;
    MDD_HIP_CHECK(hipMalloc((void **)&d, 2 * n * sizeof(unsigned short)));
    m->owned.push_back(d);
    MDD_HIP_CHECK(hipMemcpy(d, buf.data(), 2 * n * sizeof(unsigned short), hipMemcpyHostToDevice));
    out->hi = d; out->lo = d + n;
    return MDD_OK;
}

// a DevBuf of n floats holds a split tensor of n elements: hi plane then lo plane
static SplitPtr split_view(const DevBuf &b, size_t n) {
    SplitPtr s; s.hi = reinterpret_cast<unsigned short *>(b.p); s.lo = s.hi ? s.hi + n : nullptr; return s;
}
static const SplitPtr kNoSplit = {nullptr, nullptr};

// a workspace buffer was reallocated: captured graphs of THIS thread's handle hold stale pointers (set by ensure(), consumed
// by the forward_prepare() call that ran it; thread-local because handles on different host threads prepare concurrently)
static thread_local bool g_ws_moved = false;

// Workspace growth (hipFree / hipMalloc / clearing) and stream capture exclude each other process-wide: a capture in
// progress on one host thread makes another thread's allocation-time calls fail ("operation would make the legacy
// stream depend on a capturing stream").  Both are rare (first call of a shape); graph REPLAYS never take this lock.
static std::mutex g_prep_mu;

static int ensure(DevBuf &b, size_t n, hipStream_t zero_stream) {
    if (b.cap >= n) return MDD_OK;
    g_ws_moved = true;
    if (b.p) MDD_HIP_CHECK(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    MDD_HIP_CHECK(hipMalloc((void **)&b.p, n * sizeof(float)));
    MDD_HIP_CHECK(hipMemsetAsync(b.p, 0, n * sizeof(float), zero_stream));  // padded batch rows of the packed h exchange must be finite
    MDD_HIP_CHECK(hipStreamSynchronize(zero_stream));
    b.cap = n;
    return MDD_OK;
}

static const std::vector<float> *get(mdd_model *m, const std::string &key, size_t numel) {
    auto it = m->host.find(key);
    if (it == m->host.end()) { set_error("weight '%s' was never loaded", key.c_str()); return nullptr; }
    if (it->second.size() != numel) {
        set_error("weight '%s' has %zu elements, expected %zu", key.c_str(), it->second.size(), numel);
        return nullptr;
    }
    return &it->second;
}

// eval-mode BatchNorm as y = x*scale + shift
static bool bn_fold(mdd_model *m, const std::string &prefix, int n, std::vector<float> &scale, std::vector<float> &shift) {
    const auto *w = get(m, prefix + ".weight", n), *b = get(m, prefix + ".bias", n);
    const auto *mu = get(m, prefix + ".running_mean", n), *var = get(m, prefix + ".running_var", n);
    if (!w || !b || !mu || !var) return false;
    scale.resize(n); shift.resize(n);
    for (int i = 0; i < n; i++) {
        scale[i] = (*w)[i] / sqrtf((*var)[i] + m->cfg.bn_eps);
        shift[i] = (*b)[i] - (*mu)[i] * scale[i];
    }
    return true;
}

// rows n = g*H + u  ->  n' = u*4 + g (see lstm.hip); concatenates the two directions
static bool pack_gate_rows(mdd_model *m, const std::string &base, const char *what, int H, int K, std::vector<float> &out) {
    out.assign((size_t)2 * 4 * H * K, 0.f);
    for (int d = 0; d < 2; d++) {
        const auto *w = get(m, base + what + (d ? "_reverse" : ""), (size_t)4 * H * K);
        if (!w) return false;
        for (int g = 0; g < 4; g++)
            for (int u = 0; u < H; u++)
                memcpy(&out[((size_t)d * 4 * H + u * 4 + g) * K], &(*w)[((size_t)g * H + u) * K], sizeof(float) * K);
    }
    return true;
}

static bool use_packed(const mdd_model *m) { return m->cfg.hidden == 384 || m->cfg.hidden == 256; }

// Whh' [2][4H][H] (gate-permuted rows) -> Wp[d][ut][j][lane][m] (see lstm.hip)
static void pack_whh(const std::vector<float> &w, int H, std::vector<float> &out) {
    const int NUT = H / 4, J = H / 16;
    out.resize(w.size());
    for (int d = 0; d < 2; d++)
        for (int ut = 0; ut < NUT; ut++)
            for (int j = 0; j < J; j++)
                for (int lane = 0; lane < 64; lane++)
                    for (int mm = 0; mm < 4; mm++)
                        out[((((size_t)d * NUT + ut) * J + j) * 64 + lane) * 4 + mm] =
                            w[((size_t)d * 4 * H + ut * 16 + (lane & 15)) * H + 16 * j + 4 * (lane >> 4) + mm];
}

// fp32 matrix -> three bf16 planes (hi | mid | lo, same element order), on the device
static int upload_split3(mdd_model *m, const std::vector<float> &w, unsigned short **out) {
    const size_t n = w.size();
    std::vector<unsigned short> buf(3 * n);
    for (size_t i = 0; i < n; i++) {
        buf[i] = host_bf16(w[i]);
        const float r1 = w[i] - host_bf16_f32(buf[i]);
        buf[n + i] = host_bf16(r1);
        buf[2 * n + i] = host_bf16(r1 - host_bf16_f32(buf[n + i]));
    }
    MDD_HIP_CHECK(hipMalloc((void **)out, 3 * n * sizeof(unsigned short)));
    m->owned.push_back(*out);
    MDD_HIP_CHECK(hipMemcpy(*out, buf.data(), 3 * n * sizeof(unsigned short), hipMemcpyHostToDevice));
    return MDD_OK;
}

// The forward as an ordered list of stages (one or more kernel launches each).  mdd_forward captures all
// of them into one graph; mdd_forward_profile replays them one by one between HIP events.
struct Stage { const char *name; int launches; double flops; };

static int n_stages(const mdd_model *m) { return 2 + 2 * m->cfg.layers + 6; }

static int run_stage(mdd_model *m, int si, const float *x, int B, int T, const int64_t *x1, int L, float *logp,
                     hipStream_t st, Stage *info) {
    const mdd_config &c = m->cfg;
    const int H = c.hidden, H2 = 2 * H, G2 = 8 * H, Tp = T / 2, Lp = L, nl = c.layers;
    const bool x3 = m->x3();
    const size_t rows = (size_t)Tp * B, trows = (size_t)L * B;
    static thread_local char namebuf[32];
    Stage dummy; if (!info) info = &dummy;
    info->launches = 1; info->flops = 0.0;
    if (si == 0 && m->conv_fused()) {   // conv0 recomputed per output row (x1.5) + conv1 as implicit GEMM, one kernel
        info->name = "conv_fused";
        info->flops = 2.0 * 9 * c.channels * (double)B * Tp * m->W2() * (c.channels + 6.0);
        if (m->x6())   // fp32-grade form: three K-tile-major planes straight into the projection GEMM's operand buffer
            return launch_conv_fused3(x, m->w_conv0, m->sc0, m->sh0, m->w_conv1_3, m->sc1, m->sh1, reinterpret_cast<unsigned short *>(m->p3.p),
                                      m->taps ? m->seq0.p : nullptr, B, T, m->raw_T, st);
        return launch_conv_fused(x, m->w_conv0, m->sc0, m->sh0, m->w_conv1_s, m->sc1, m->sh1, split_view(m->seq0_s, rows * m->rnn_in()),
                                 nullptr, B, T, m->raw_T, st);
    }
    if (si == 1 && m->conv_fused()) { info->name = "conv1_in_fused"; info->launches = 0; return MDD_OK; }
    if (si == 0) { info->name = "conv0"; info->flops = 2.0 * 9 * c.channels * (double)B * T * m->W1();
        return launch_conv0(x, m->w_conv0, m->sc0, m->sh0, m->y0.p, B, T, c.feat, c.channels, st); }
    if (si == 1) { info->name = "conv1"; info->flops = 2.0 * 9 * c.channels * c.channels * (double)B * Tp * m->W2();
        return launch_conv1(m->y0.p, m->w_conv1t, m->sc1, m->sh1, x3 ? nullptr : m->seq0.p,
                            x3 ? split_view(m->seq0_s, rows * m->rnn_in()) : kNoSplit, B, T, m->W1(), c.channels, st); }
    si -= 2;
    if (si < 2 * nl) {
        const int n = si / 2;
        const int K = n == 0 ? m->rnn_in() : H2;
        if (si % 2 == 0) {
            snprintf(namebuf, sizeof(namebuf), "gemm_ih%d", n); info->name = namebuf;
            info->flops = 2.0 * (double)Tp * B * G2 * K;
            if (x3) {
                const SplitPtr in = n == 0 ? split_view(m->seq0_s, rows * K) : split_view(m->act_s[(n - 1) & 1], rows * K);
                return launch_gemm_bf16x3(in, m->wih_s[n], nullptr, m->gx.p, nullptr, Tp * B, G2, K, K, K, G2, 1, 0, 0, 0, st);
            }
            const float *in = n == 0 ? m->seq0.p : m->act[(n - 1) & 1].p;
            if (m->x6()) {   // fp32-grade arithmetic at 6/16 of the fp32 MFMA's cost (gemm_bf16x6.hip)
                unsigned short *p3 = reinterpret_cast<unsigned short *>(m->p3.p);
                if (!(n == 0 && m->conv_fused()))      // (layer 0: the fused front-end has written the planes already)
                    if (int rc = launch_split3(in, Tp * B, K, K, p3, st)) return rc;
                return launch_gemm_f32x6(p3, (size_t)Tp * B * K, m->wih_3[n], (size_t)G2 * K, nullptr, m->gx.p, Tp * B, G2, K, G2, st);
            }
            return launch_gemm_nt(in, m->wih[n], nullptr, m->gx.p, Tp * B, G2, K, K, K, G2, 1, 0, 0, 0, st);
        }
        snprintf(namebuf, sizeof(namebuf), "lstm%d", n); info->name = namebuf;
        info->launches = Tp; info->flops = 2.0 * 2 * (double)B * H * 4 * H * Tp;
        LstmStepArgs a;
        a.gx = m->gx.p; a.whh = m->whh[n]; a.hbuf = m->hbuf.p; a.cbuf = m->cbuf.p;
        a.T = Tp; a.B = B; a.H = H; a.packed = use_packed(m);
        a.whh_split = m->whh_s[n]; a.hsplit = (x3 && m->lstm_x3) ? reinterpret_cast<unsigned short *>(m->hsplit.p) : nullptr;
        a.seqlen = m->tlen;
        if (n == nl - 1) {   // raw h: the attention queries X (fp32 for the tail, split for the score GEMM)
            a.out = m->xraw.p; a.out_raw = m->xraw.p; a.oscale = nullptr; a.oshift = nullptr;
            a.out_split = x3 ? split_view(m->x_s, rows * H2) : kNoSplit;
        } else {             // next layer's BatchNorm folded into the store
            a.out = x3 ? nullptr : m->act[n & 1].p; a.out_raw = m->taps ? m->tap_rnn[n].p : nullptr;
            a.out_split = x3 ? split_view(m->act_s[n & 1], rows * H2) : kNoSplit;
            a.oscale = m->bn_scale[n + 1]; a.oshift = m->bn_shift[n + 1];
        }
        if (m->persist(B)) {
            info->launches = 1;
            if (m->lx6(B)) return launch_lstm_layer_x6(a, m->whh_3[n], reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st);
            return x3 ? launch_lstm_layer_granule(a, reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st)
                      : launch_lstm_layer_f32(a, reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st);
        }
        return launch_lstm_layer(a, st);
    }
    si -= 2 * nl;
    switch (si) {
    case 0:  // text encoder (model_ctc.py:193,198) and keys (:201)
        info->name = "embed";
        return launch_embed(m->emb, c.emb_rows, c.emb_dim, x1, B, L, x3 ? nullptr : m->embo.p,
                            x3 ? split_view(m->embo_s, trows * c.emb_dim) : kNoSplit, m->err_flag, st);
    case 1:
        info->name = "gemm_text"; info->flops = 2.0 * (double)L * B * G2 * c.emb_dim;
        if (x3) return launch_gemm_bf16x3(split_view(m->embo_s, trows * c.emb_dim), m->t_wih_s, m->t_bias, m->gx.p, nullptr, L * B, G2,
                                          c.emb_dim, c.emb_dim, c.emb_dim, G2, 1, 0, 0, 0, st);
        if (m->x6()) {
            unsigned short *p3 = reinterpret_cast<unsigned short *>(m->p3.p);
            if (int rc = launch_split3(m->embo.p, L * B, c.emb_dim, c.emb_dim, p3, st)) return rc;
            return launch_gemm_f32x6(p3, (size_t)L * B * c.emb_dim, m->t_wih_3, (size_t)G2 * c.emb_dim, m->t_bias, m->gx.p, L * B, G2, c.emb_dim, G2, st);
        }
        return launch_gemm_nt(m->embo.p, m->t_wih, m->t_bias, m->gx.p, L * B, G2, c.emb_dim, c.emb_dim, c.emb_dim, G2, 1, 0, 0, 0, st);
    case 2: {
        info->name = "lstm_text"; info->launches = L; info->flops = 2.0 * 2 * (double)B * H * 4 * H * L;
        LstmStepArgs a;
        a.gx = m->gx.p; a.whh = m->t_whh; a.hbuf = m->hbuf.p; a.cbuf = m->cbuf.p;
        a.out = m->text.p; a.out_raw = m->text.p; a.oscale = nullptr; a.oshift = nullptr;
        a.out_split = x3 ? split_view(m->text_s, trows * H2) : kNoSplit;
        a.T = L; a.B = B; a.H = H; a.packed = use_packed(m);
        a.whh_split = m->t_whh_s; a.hsplit = (x3 && m->lstm_x3) ? reinterpret_cast<unsigned short *>(m->hsplit.p) : nullptr;
        a.seqlen = m->llen;
        if (m->persist(B)) {
            info->launches = 1;
            if (m->lx6(B)) return launch_lstm_layer_x6(a, m->t_whh_3, reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st);
            return x3 ? launch_lstm_layer_granule(a, reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st)
                      : launch_lstm_layer_f32(a, reinterpret_cast<unsigned short *>(m->hx.p), m->sync_words, m->err_flag, st);
        }
        return launch_lstm_layer(a, st);
    }
    case 3:
        info->name = "gemm_key"; info->flops = 2.0 * (double)L * B * H2 * H2;
        if (x3) {
            const SplitPtr ks = split_view(m->key_s, trows * H2);
            return launch_gemm_bf16x3(split_view(m->text_s, trows * H2), m->w_score_s, nullptr, nullptr, &ks, L * B, H2, H2, H2, H2, H2,
                                      1, 0, 0, 0, st);
        }
        return launch_gemm_nt(m->text.p, m->w_score, nullptr, m->key.p, L * B, H2, H2, H2, H2, H2, 1, 0, 0, 0, st);
    case 4:  // scores S[b][t][l] = X[t,b,:] . key[l,b,:]   (:204)
        info->name = "gemm_score"; info->flops = 2.0 * (double)B * Tp * L * H2;
        if (x3) return launch_gemm_bf16x3(split_view(m->x_s, rows * H2), split_view(m->key_s, trows * H2), nullptr, m->S.p, nullptr, Tp, L,
                                          H2, B * H2, B * H2, Lp, B, H2, H2, (long)Tp * Lp, st);
        return launch_gemm_nt(m->xraw.p, m->key.p, nullptr, m->S.p, Tp, L, H2, B * H2, B * H2, Lp, B, H2, H2, (long)Tp * Lp, st);
    default:
        info->name = "attn_tail"; info->flops = 2.0 * (double)B * Tp * ((double)L * H2 + 2.0 * H2 * c.num_class);
        return launch_attn_tail(m->S.p, Lp, m->xraw.p, m->text.p, m->fscale, m->fshift, m->w_fc, m->w_fcp, logp, Tp, B, L, H2, c.num_class, st, m->llen);
    }
}

// Enter / leave the device gate around the enqueue of one forward on `st` (no-op while `st` is being captured by the
// caller: the captured graph then carries the caller's own ordering).
int device_gate_enter(int device, hipStream_t st, bool *held) {
    *held = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return MDD_OK;
    DeviceGate &g = g_gate[device & 63];
    g.mu.lock();
    *held = true;
    if (g.armed && g.stream != st) {
        hipError_t e = hipStreamWaitEvent(st, g.done, 0);
        if (e != hipSuccess) { g.mu.unlock(); *held = false; set_error("device gate: %s", hipGetErrorString(e)); return MDD_ERR_HIP; }
    }
    return MDD_OK;
}
int device_gate_leave(int device, hipStream_t st, bool held, int rc) {
    if (!held) return rc;
    DeviceGate &g = g_gate[device & 63];
    hipError_t e = hipSuccess;
    if (!g.done) e = hipEventCreateWithFlags(&g.done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(g.done, st);
    if (e == hipSuccess) { g.armed = true; g.stream = st; }
    g.mu.unlock();
    if (e != hipSuccess && rc == MDD_OK) { set_error("device gate: %s", hipGetErrorString(e)); return MDD_ERR_HIP; }
    return rc;
}
static int gate_enter(mdd_model *m, hipStream_t st, bool *held) { return device_gate_enter(m->device, st, held); }
static int gate_leave(mdd_model *m, hipStream_t st, bool held, int rc) { return device_gate_leave(m->device, st, held, rc); }

static int forward_enqueue(mdd_model *m, const float *x, int B, int T, const int64_t *x1, int L, float *logp, hipStream_t st) {
    for (int si = 0; si < n_stages(m); si++)
        if (int rc = run_stage(m, si, x, B, T, x1, L, logp, st, nullptr)) return rc;
    return MDD_OK;
}

}  // namespace mdd

using namespace mdd;

extern "C" const char *mdd_last_error(void) { return g_err; }
extern "C" int mdd_version(void) { return 100; }

extern "C" int mdd_create(const mdd_config *cfg, int device, mdd_model **out) {
    if (!cfg || !out) { set_error("mdd_create: null argument"); return MDD_ERR_ARG; }
    if (cfg->hidden <= 0 || cfg->hidden % 4 || cfg->layers < 1 || cfg->num_class < 2 || cfg->feat < 3 ||
        (cfg->channels != 32 && cfg->channels != 4) || cfg->emb_rows < 1 || cfg->emb_dim < 1) {
        set_error("mdd_create: unsupported geometry (hidden %% 4 == 0, channels in {32,4})");
        return MDD_ERR_ARG;
    }
    int ndev = 0;
    MDD_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("mdd_create: device %d of %d", device, ndev); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MDD_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("libmdd_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        return MDD_ERR_ARG;
    }
    mdd_model *m = new mdd_model();
    m->cfg = *cfg;
    m->device = device;
    const char *pr = getenv("MDD_PRECISION");
    if (pr && (!strcmp(pr, "f32") || !strcmp(pr, "0"))) m->precision = 0;
    if (pr && (!strcmp(pr, "bf16x3") || !strcmp(pr, "1"))) m->precision = 1;
    if (pr && (!strcmp(pr, "f32x6") || !strcmp(pr, "2"))) m->precision = 2;
    const char *lx = getenv("MDD_LSTM");
    m->lstm_x3 = lx && !strcmp(lx, "x3");
    if (lx && !strcmp(lx, "step")) m->lstm_persist = false;
    const char *g = getenv("MDD_GRAPH");
    m->use_graph = !(g && g[0] == '0');
    if (int rc = init_kernel_attributes()) { delete m; return rc; }
    if (int rc = init_lstm_attributes()) { delete m; return rc; }
    if (int rc = init_granule_attributes()) { delete m; return rc; }
    if (int rc = init_lstm_f32_attributes()) { delete m; return rc; }
    if (int rc = init_lstm_x6_attributes()) { delete m; return rc; }
    if (int rc = init_conv_attributes()) { delete m; return rc; }
    if (int rc = init_gemm_attributes()) { delete m; return rc; }
    if (int rc = init_gemm_x6_attributes()) { delete m; return rc; }
    m->n_cu = prop.multiProcessorCount;
    if (!persistent_grid_fits(m->n_cu)) m->lstm_persist = false;   // per-step kernels instead (smaller partitions, other gfx950 SKUs)
    if (!persistent_f32_grid_fits(m->n_cu)) m->lstm_persist_f32 = false;
    if (!persistent_x6_grid_fits(m->n_cu)) m->lstm_persist_x6 = false;
    { const char *e6 = getenv("MDD_LSTM_X6"); if (e6 && e6[0] == '0') m->lstm_persist_x6 = false; if (e6 && e6[0] == 'f') m->lstm_x6_force = true; }
    hipError_t e = hipMalloc((void **)&m->err_flag, sizeof(int));
    if (e == hipSuccess) e = hipMemset(m->err_flag, 0, sizeof(int));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&m->sync_words, 32 * sizeof(unsigned int));
    if (e != hipSuccess) { set_error("mdd_create: %s", hipGetErrorString(e)); delete m; return MDD_ERR_HIP; }
    *out = m;
    return MDD_OK;
}

extern "C" void mdd_destroy(mdd_model *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void *p : m->owned) (void)hipFree(p);
    DevBuf *bufs[] = {&m->y0, &m->seq0, &m->gx, &m->act[0], &m->act[1], &m->xraw, &m->hbuf, &m->cbuf, &m->embo, &m->text, &m->key, &m->S,
                      &m->seq0_s, &m->act_s[0], &m->act_s[1], &m->x_s, &m->embo_s, &m->text_s, &m->key_s, &m->hsplit, &m->hx, &m->xstack, &m->p3};
    for (DevBuf *b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto &b : m->tap_rnn) if (b.p) (void)hipFree(b.p);
    if (m->err_flag) (void)hipFree(m->err_flag);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    if (m->sync_words) (void)hipFree(m->sync_words);
    delete m;
}

extern "C" int mdd_load_weight(mdd_model *m, const char *key, const float *data, const int64_t *shape, int32_t ndim) {
    if (!m || !key || (!data && ndim > 0) || ndim < 0 || ndim > 4) { set_error("mdd_load_weight: bad argument"); return MDD_ERR_ARG; }
    std::string k(key);
    if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return MDD_OK;  // unused in eval
    size_t n = 1;
    for (int i = 0; i < ndim; i++) { if (shape[i] < 0) { set_error("negative dim"); return MDD_ERR_ARG; } n *= (size_t)shape[i]; }
    m->host[k].assign(data, data + n);
    m->finalized = false;
    return MDD_OK;
}

extern "C" int mdd_finalize_weights(mdd_model *m) {
    if (!m) { set_error("null model"); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipSetDevice(m->device));
    const mdd_config &c = m->cfg;
    const int ch = c.channels, H = c.hidden;
    int rc;
    for (void *p : m->owned) (void)hipFree(p);
    m->owned.clear(); m->wih.clear(); m->whh.clear(); m->wih_s.clear(); m->whh_s.clear(); m->wih_3.clear(); m->t_wih_3 = nullptr;
    m->bn_scale.assign(c.layers, nullptr); m->bn_shift.assign(c.layers, nullptr);
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear();
    std::vector<float> sc, sh, tmp;
    {   // conv0 / conv1: fold bias + BN into scale/shift; conv1 weights -> [ci][kh][kw][co]
        const auto *w0 = get(m, "conv.0.conv.weight", (size_t)ch * 9), *b0 = get(m, "conv.0.conv.bias", ch);
        const auto *w1 = get(m, "conv.1.conv.weight", (size_t)ch * ch * 9), *b1 = get(m, "conv.1.conv.bias", ch);
        if (!w0 || !b0 || !w1 || !b1) return MDD_ERR_STATE;
        if (!bn_fold(m, "conv.0.batch_norm", ch, sc, sh)) return MDD_ERR_STATE;
        for (int i = 0; i < ch; i++) sh[i] += (*b0)[i] * sc[i];
        if ((rc = upload(m, *w0, &m->w_conv0)) || (rc = upload(m, sc, &m->sc0)) || (rc = upload(m, sh, &m->sh0))) return rc;
        if (!bn_fold(m, "conv.1.batch_norm", ch, sc, sh)) return MDD_ERR_STATE;
        for (int i = 0; i < ch; i++) sh[i] += (*b1)[i] * sc[i];
        tmp.assign((size_t)ch * 9 * ch, 0.f);
        for (int co = 0; co < ch; co++)
            for (int ci = 0; ci < ch; ci++)
                for (int k = 0; k < 9; k++) tmp[((size_t)ci * 9 + k) * ch + co] = (*w1)[((size_t)co * ch + ci) * 9 + k];
        if ((rc = upload(m, tmp, &m->w_conv1t)) || (rc = upload(m, sc, &m->sc1)) || (rc = upload(m, sh, &m->sh1))) return rc;
        // conv1 weights for the fused MFMA front-end: [co][kh][kw][ci] (k = (kh*3+kw)*ch + ci), split-bf16
        tmp.assign((size_t)ch * 9 * ch, 0.f);
        for (int co = 0; co < ch; co++)
            for (int ci = 0; ci < ch; ci++)
                for (int k = 0; k < 9; k++) tmp[((size_t)co * 9 + k) * ch + ci] = (*w1)[((size_t)co * ch + ci) * 9 + k];
        if ((rc = upload_split(m, tmp, &m->w_conv1_s))) return rc;
        {   // the same matrix as three planes (hi | mid | lo, each [co][288] row-major) for the fp32-grade fused front-end
            const size_t n = tmp.size();
            std::vector<unsigned short> buf(3 * n);
            for (size_t i = 0; i < n; i++) {
                buf[i] = host_bf16(tmp[i]);
                const float r1 = tmp[i] - host_bf16_f32(buf[i]);
                buf[n + i] = host_bf16(r1);
                buf[2 * n + i] = host_bf16(r1 - host_bf16_f32(buf[n + i]));
            }
            MDD_HIP_CHECK(hipMalloc((void **)&m->w_conv1_3, 3 * n * sizeof(unsigned short)));
            m->owned.push_back(m->w_conv1_3);
            MDD_HIP_CHECK(hipMemcpy(m->w_conv1_3, buf.data(), 3 * n * sizeof(unsigned short), hipMemcpyHostToDevice));
        }
    }
    for (int n = 0; n < c.layers; n++) {
        char base[64];
        snprintf(base, sizeof(base), "rnns.%d.rnn.", n);
        const int K = n == 0 ? m->rnn_in() : 2 * H;
        float *d = nullptr;
        if (!pack_gate_rows(m, base, "weight_ih_l0", H, K, tmp)) return MDD_ERR_STATE;
        if ((rc = upload(m, tmp, &d))) return rc;
        m->wih.push_back(d);
        { SplitPtr sp{nullptr, nullptr}; if ((rc = upload_split(m, tmp, &sp))) return rc; m->wih_s.push_back(sp); }
        if (K % 32 == 0) {   // f32x6: hi | mid | lo planes in the kernel's K-tile-major order, made on the device from the fp32 copy
            unsigned short *p3 = nullptr;
            MDD_HIP_CHECK(hipMalloc((void **)&p3, (size_t)3 * 8 * H * K * sizeof(unsigned short)));
            m->owned.push_back(p3);
            if ((rc = launch_split3(d, 8 * H, K, K, p3, nullptr))) return rc;
            m->wih_3.push_back(p3);
        } else m->wih_3.push_back(nullptr);
        if (!pack_gate_rows(m, base, "weight_hh_l0", H, H, tmp)) return MDD_ERR_STATE;
        { SplitPtr sp{nullptr, nullptr}; if ((rc = upload_split(m, tmp, &sp))) return rc; m->whh_s.push_back(sp); }
        { unsigned short *p3 = nullptr; if (use_packed(m) && (rc = upload_split3(m, tmp, &p3))) return rc; m->whh_3.push_back(p3); }
        if (use_packed(m)) { std::vector<float> pk; pack_whh(tmp, H, pk); tmp.swap(pk); }
        if ((rc = upload(m, tmp, &d))) return rc;
        m->whh.push_back(d);
        if (n > 0) {
            snprintf(base, sizeof(base), "rnns.%d.batch_norm", n);
            if (!bn_fold(m, base, 2 * H, sc, sh)) return MDD_ERR_STATE;
            if ((rc = upload(m, sc, &m->bn_scale[n])) || (rc = upload(m, sh, &m->bn_shift[n]))) return rc;
        }
    }
    {   // text encoder: bias_ih + bias_hh folded into the input projection's epilogue
        const auto *e = get(m, "embeds.weight", (size_t)c.emb_rows * c.emb_dim);
        if (!e) return MDD_ERR_STATE;
        if ((rc = upload(m, *e, &m->emb))) return rc;
        if (!pack_gate_rows(m, "lstm_embeds.", "weight_ih_l0", H, c.emb_dim, tmp)) return MDD_ERR_STATE;
        if ((rc = upload(m, tmp, &m->t_wih)) || (rc = upload_split(m, tmp, &m->t_wih_s))) return rc;
        if (c.emb_dim % 32 == 0) {
            MDD_HIP_CHECK(hipMalloc((void **)&m->t_wih_3, (size_t)3 * 8 * H * c.emb_dim * sizeof(unsigned short)));
            m->owned.push_back(m->t_wih_3);
            if ((rc = launch_split3(m->t_wih, 8 * H, c.emb_dim, c.emb_dim, m->t_wih_3, nullptr))) return rc;
        }
        if (!pack_gate_rows(m, "lstm_embeds.", "weight_hh_l0", H, H, tmp)) return MDD_ERR_STATE;
        if ((rc = upload_split(m, tmp, &m->t_whh_s))) return rc;
        if (use_packed(m) && (rc = upload_split3(m, tmp, &m->t_whh_3))) return rc;
        if (use_packed(m)) { std::vector<float> pk; pack_whh(tmp, H, pk); tmp.swap(pk); }
        if ((rc = upload(m, tmp, &m->t_whh))) return rc;
        std::vector<float> bi, bh;
        if (!pack_gate_rows(m, "lstm_embeds.", "bias_ih_l0", H, 1, bi) || !pack_gate_rows(m, "lstm_embeds.", "bias_hh_l0", H, 1, bh)) return MDD_ERR_STATE;
        for (size_t i = 0; i < bi.size(); i++) bi[i] += bh[i];
        if ((rc = upload(m, bi, &m->t_bias))) return rc;
    }
    {
        const auto *ws = get(m, "score.weight", (size_t)4 * H * H), *wf = get(m, "fc.1.weight", (size_t)c.num_class * 4 * H);
        if (!ws || !wf) return MDD_ERR_STATE;
        if (!bn_fold(m, "fc.0", 4 * H, sc, sh)) return MDD_ERR_STATE;
        if ((rc = upload_split(m, *ws, &m->w_score_s))) return rc;
        if ((rc = upload(m, *ws, &m->w_score)) || (rc = upload(m, *wf, &m->w_fc)) || (rc = upload(m, sc, &m->fscale)) ||
            (rc = upload(m, sh, &m->fshift))) return rc;
        m->w_fcp = nullptr;
        const int D2 = 4 * H;
        if (D2 % 64 == 0 && c.num_class <= 48) {   // consumer-order repack for attn_tail_mfma_kernel
            const int J = D2 / 64;
            std::vector<float> pk((size_t)4 * 3 * J * 64 * 4, 0.f);
            for (int w = 0; w < 4; w++)
                for (int nt = 0; nt < 3; nt++)
                    for (int j = 0; j < J; j++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int mm = 0; mm < 4; mm++) {
                                const int n = nt * 16 + (lane & 15), k = w * (D2 / 4) + 16 * j + 4 * (lane >> 4) + mm;
                                if (n < c.num_class) pk[((((size_t)w * 3 + nt) * J + j) * 64 + lane) * 4 + mm] = (*wf)[(size_t)n * D2 + k];
                            }
            if ((rc = upload(m, pk, &m->w_fcp))) return rc;
        }
    }
    MDD_HIP_CHECK(hipDeviceSynchronize());
    m->finalized = true;
    return MDD_OK;
}

extern "C" int mdd_enable_taps(mdd_model *m, int32_t on) {
    if (!m) return MDD_ERR_ARG;
    m->taps = on != 0;
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear();
    return MDD_OK;
}

extern "C" int mdd_set_precision(mdd_model *m, int32_t mode) {
    if (!m || mode < 0 || mode > 2) { set_error("mdd_set_precision: mode must be 0 (fp32 MFMA), 1 (split-bf16 x3) or 2 (f32x6 projections)"); return MDD_ERR_ARG; }
    if (m->precision != mode) {
        m->precision = mode;
        for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
        m->graphs.clear();
    }
    return MDD_OK;
}
extern "C" int32_t mdd_get_precision(mdd_model *m) { return m ? (m->x3() ? 1 : (m->x6() ? 2 : 0)) : -1; }

extern "C" int mdd_stack_skip(const float *raw_dev, int32_t B, int32_t T_raw, int32_t D, int32_t right, int32_t skip,
                              int32_t n_down, float *out_dev, void *stream) {
    if (!raw_dev || !out_dev) { set_error("mdd_stack_skip: null pointer"); return MDD_ERR_ARG; }
    return launch_stack_skip(raw_dev, B, T_raw, D, right, skip, n_down, out_dev, (hipStream_t)stream);
}

static int forward_prepare(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L, float *logp_dev) {
    if (!m || !x_dev || !x1_dev || !logp_dev) { set_error("mdd_forward: null pointer"); return MDD_ERR_ARG; }
    if (!m->finalized) { set_error("mdd_forward: call mdd_finalize_weights first"); return MDD_ERR_STATE; }
    if (B <= 0 || T < 2 || L <= 0) { set_error("mdd_forward: bad shape B=%d T=%d L=%d", B, T, L); return MDD_ERR_ARG; }
    if (T % 2) { set_error("mdd_forward: T must be even (data_loader.py:140-142 pads to n_downsample)"); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipSetDevice(m->device));
    std::lock_guard<std::mutex> prep_lock(g_prep_mu);
    hipStream_t zs = m->cap_stream;
    const mdd_config &c = m->cfg;
    const int H = c.hidden, Tp = T / 2;
    int rc;
    const size_t rows = (size_t)Tp * B, trows = (size_t)L * B, mrows = rows > trows ? rows : trows;
    if ((!m->conv_fused() && (rc = ensure(m->y0, (size_t)B * c.channels * T * m->W1(), zs))) || (rc = ensure(m->seq0, rows * m->rnn_in(), zs)) ||
        (rc = ensure(m->gx, mrows * 8 * H, zs)) || (rc = ensure(m->act[0], rows * 2 * H, zs)) || (rc = ensure(m->act[1], rows * 2 * H, zs)) ||
        (rc = ensure(m->xraw, rows * 2 * H, zs)) || (rc = ensure(m->hbuf, (size_t)4 * ((B + 15) / 16 * 16) * H, zs)) || (rc = ensure(m->cbuf, (size_t)2 * ((B + 15) / 16 * 16) * H, zs)) ||
        (rc = ensure(m->embo, trows * c.emb_dim, zs)) || (rc = ensure(m->text, trows * 2 * H, zs)) || (rc = ensure(m->key, trows * 2 * H, zs)) ||
        (rc = ensure(m->S, (size_t)B * Tp * L, zs)))
        return rc;
    if (m->x3() && ((rc = ensure(m->seq0_s, rows * m->rnn_in(), zs)) || (rc = ensure(m->act_s[0], rows * 2 * H, zs)) ||
                    (rc = ensure(m->act_s[1], rows * 2 * H, zs)) || (rc = ensure(m->x_s, rows * 2 * H, zs)) ||
                    (rc = ensure(m->embo_s, trows * c.emb_dim, zs)) || (rc = ensure(m->text_s, trows * 2 * H, zs)) ||
                    (rc = ensure(m->key_s, trows * 2 * H, zs)) || (rc = ensure(m->hsplit, (size_t)4 * B * H, zs))))
        return rc;
    {   // f32x6: three bf16 planes of the largest projection operand = 1.5 x its fp32 size (in floats: 3/2)
        const size_t kmax = (size_t)(m->rnn_in() > 2 * H ? m->rnn_in() : 2 * H), k2 = (size_t)c.emb_dim;
        const size_t need = (rows * kmax > trows * k2 ? rows * kmax : trows * k2) * 3 / 2 + 64;
        if (m->x6() && (rc = ensure(m->p3, need, zs))) return rc;
    }
    if (m->persist(B)) {   // the exchange buffer of the persistent layers (u64 granules; three bf16 planes per (parity, team, tile) in lstm_x6.hip) + stamps
        size_t need = (size_t)2 * 32 * granule_bg(B) * H * 2;
        if (m->lx6(B) && lstm_x6_hx_bytes(H, B) / 4 > need) need = lstm_x6_hx_bytes(H, B) / 4;
        if ((rc = ensure(m->hx, need + 64 + 256 * 6 * 2, zs))) return rc;
    }
    if (m->taps) {
        m->tap_rnn.resize(c.layers);
        for (int n = 0; n + 1 < c.layers; n++) if ((rc = ensure(m->tap_rnn[n], rows * 2 * H, zs))) return rc;
    }
    m->lastB = B; m->lastT = T; m->lastL = L;
    if (g_ws_moved) {   // hipFree above synchronised the device, so no replay of an old graph is still running
        for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
        m->graphs.clear();
        g_ws_moved = false;
    }
    return MDD_OK;
}

extern "C" int mdd_forward(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                           float *logp_dev, void *stream) {
    int rc = forward_prepare(m, x_dev, B, T, x1_dev, L, logp_dev);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool gated = m->persist(B);
    bool held = false;
    if (!m->use_graph) {
        if (gated && (rc = gate_enter(m, st, &held))) return rc;
        rc = forward_enqueue(m, x_dev, B, T, x1_dev, L, logp_dev, st);
        return gate_leave(m, st, held, rc);
    }
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.x = x_dev; key.x1 = x1_dev; key.out = logp_dev; key.B = B; key.T = T; key.L = L; key.Traw = m->raw_T; key.tlen = m->tlen; key.llen = m->llen;
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        if (m->graphs.size() >= 8) { for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second); m->graphs.clear(); }
        hipGraph_t graph = nullptr;
        std::lock_guard<std::mutex> prep_lock(g_prep_mu);
        MDD_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        rc = forward_enqueue(m, x_dev, B, T, x1_dev, L, logp_dev, m->cap_stream);
        hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { set_error("graph capture failed: %s", hipGetErrorString(e)); return MDD_ERR_HIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error("graph instantiate failed: %s", hipGetErrorString(e)); return MDD_ERR_HIP; }
        it = m->graphs.emplace(key, exec).first;
    }
    if (gated && (rc = gate_enter(m, st, &held))) return rc;
    hipError_t le = hipGraphLaunch(it->second, st);
    if (le != hipSuccess) { set_error("hipGraphLaunch failed: %s", hipGetErrorString(le)); return gate_leave(m, st, held, MDD_ERR_HIP); }
    return gate_leave(m, st, held, MDD_OK);
}

// Several reference batches of DIFFERENT padded lengths in one launch sequence.  The reference pads every batch to its own
// maximum and masks nothing (AA/models/model_ctc.py:186,198,204-205), so an utterance's posteriors depend on its batch's
// padded length T_g and canonical length L_g: rows of batch g carry frames_dev[b] = T_g / 2 and canon_dev[b] = L_g.  What
// depends on the batch's length -- where the reverse BiLSTM direction starts (frame T_g/2 - 1 / token L_g - 1, zero state) and
// which keys the attention softmax runs over (l < L_g) -- follows those per-row values; everything else is row-local.  x_dev
// [B, T, F] is zero beyond each batch's T_g (as the collate's zero padding leaves it), x1_dev [B, L] is zero-padded.  Rows
// t >= frames_dev[b] of logp_dev are not meaningful.  Results for every utterance are bit-identical to running its batch alone.
extern "C" int mdd_forward_fused(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                                 const int32_t *frames_dev, const int32_t *canon_dev, float *logp_dev, void *stream) {
    if (!m || !frames_dev || !canon_dev) { set_error("mdd_forward_fused: null pointer"); return MDD_ERR_ARG; }
    m->tlen = frames_dev; m->llen = canon_dev;
    const int rc = mdd_forward(m, x_dev, B, T, x1_dev, L, logp_dev, stream);
    m->tlen = nullptr; m->llen = nullptr;
    return rc;
}

// A1 + forward in one call: raw_dev = unstacked frames [B, T_raw, feat/3].  With the fused conv front-end the stack/skip
// is only an index map inside its x-tile load (no [B,T,243] copy at all); otherwise a stacked copy is made first.
extern "C" int mdd_forward_raw(mdd_model *m, const float *raw_dev, int32_t B, int32_t T_raw, const int64_t *x1_dev, int32_t L,
                               float *logp_dev, void *stream) {
    if (!m || !raw_dev || B <= 0 || T_raw < 1) { set_error("mdd_forward_raw: bad argument"); return MDD_ERR_ARG; }
    const int D = m->cfg.feat / 3, T = mdd_stack_len(T_raw, 2, 2);
    if (m->cfg.feat != 3 * D) { set_error("mdd_forward_raw: feat=%d is not 3 stacked frames", m->cfg.feat); return MDD_ERR_ARG; }
    if (m->finalized && m->conv_fused()) {
        m->raw_T = T_raw;
        const int rc = mdd_forward(m, raw_dev, B, T, x1_dev, L, logp_dev, stream);
        m->raw_T = 0;
        return rc;
    }
    MDD_HIP_CHECK(hipSetDevice(m->device));
    {
        std::lock_guard<std::mutex> prep_lock(g_prep_mu);
        if (int rc = ensure(m->xstack, (size_t)B * T * m->cfg.feat, m->cap_stream)) return rc;
    }
    if (g_ws_moved) { for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second); m->graphs.clear(); g_ws_moved = false; }
    if (int rc = mdd_stack_skip(raw_dev, B, T_raw, D, 2, 2, 2, m->xstack.p, stream)) return rc;
    return mdd_forward(m, m->xstack.p, B, T, x1_dev, L, logp_dev, stream);
}

extern "C" int32_t mdd_forward_num_stages(mdd_model *m) { return m ? n_stages(m) : 0; }

extern "C" int mdd_forward_profile(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                                   float *logp_dev, void *stream, char *names, int32_t names_cap, float *ms,
                                   int32_t *launches, double *flops, int32_t cap) {
    int rc = forward_prepare(m, x_dev, B, T, x1_dev, L, logp_dev);
    if (rc) return rc;
    const int ns = n_stages(m);
    if (cap < ns || !ms || !launches || !flops || !names) { set_error("mdd_forward_profile: need room for %d stages", ns); return MDD_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    if (m->persist(B)) {   // keep other handles' forwards off the device while the stages replay: wait for the last gated forward
        bool held = false;
        if ((rc = gate_enter(m, st, &held))) return rc;
        if ((rc = gate_leave(m, st, held, MDD_OK))) return rc;
    }
    hipEvent_t e0, e1;
    MDD_HIP_CHECK(hipEventCreate(&e0));
    MDD_HIP_CHECK(hipEventCreate(&e1));
    std::string all;
    for (int si = 0; si < ns; si++) {
        Stage info;
        hipGraph_t graph = nullptr;
        std::unique_lock<std::mutex> prep_lock(g_prep_mu);
        MDD_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        rc = run_stage(m, si, x_dev, B, T, x1_dev, L, logp_dev, m->cap_stream, &info);
        hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
        prep_lock.unlock();
        if (rc || e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); if (!rc) set_error("stage capture failed"); return rc ? rc : MDD_ERR_HIP; }
        if (info.launches == 0) {   // stage folded into a neighbour in this configuration
            (void)hipGraphDestroy(graph);
            ms[si] = 0.f; launches[si] = 0; flops[si] = 0.0;
            if (si) all += ",";
            all += info.name;
            continue;
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error("stage instantiate failed: %s", hipGetErrorString(e)); return MDD_ERR_HIP; }
        MDD_HIP_CHECK(hipGraphLaunch(exec, st));          // warm (first replay pays upload)
        MDD_HIP_CHECK(hipStreamSynchronize(st));
        MDD_HIP_CHECK(hipEventRecord(e0, st));
        MDD_HIP_CHECK(hipGraphLaunch(exec, st));
        MDD_HIP_CHECK(hipEventRecord(e1, st));
        MDD_HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        MDD_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        (void)hipGraphExecDestroy(exec);
        ms[si] = t; launches[si] = info.launches; flops[si] = info.flops;
        if (si) all += ",";
        all += info.name;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    snprintf(names, names_cap, "%s", all.c_str());
    return MDD_OK;
}

extern "C" const float *mdd_tap(mdd_model *m, const char *name, int64_t *numel) {
    if (!m || !name || !m->lastB) return nullptr;
    const int Tp = m->lastT / 2, B = m->lastB, L = m->lastL, H2 = 2 * m->cfg.hidden;
    std::string n(name);
    const float *p = nullptr;
    int64_t ne = 0;
    if (m->x3() && (n == "conv1" || n == "key")) {   // these stages exist only as split-bf16 planes: rebuild fp32 = hi + lo
        const bool cv = n == "conv1";
        ne = cv ? (int64_t)Tp * B * m->rnn_in() : (int64_t)L * B * H2;
        DevBuf &dst = cv ? m->seq0 : m->key;
        if (launch_unsplit(split_view(cv ? m->seq0_s : m->key_s, (size_t)ne), (size_t)ne, dst.p, nullptr) != MDD_OK) return nullptr;
        if (hipStreamSynchronize(nullptr) != hipSuccess) return nullptr;
        p = dst.p;
    }
    else if (n == "lstm_dbg" && m->hx.p) {   // diagnostic stamps of the last persistent layer launch (MDD_LSTM_DBG=1)
        p = m->hx.p + (m->lx6(B) ? lstm_x6_hx_bytes(m->cfg.hidden, B) / 4 : (size_t)2 * 32 * granule_bg(B) * m->cfg.hidden * 2);
        ne = 256 * 6 * 2;
    }
    else if (n == "conv1") { p = m->seq0.p; ne = (int64_t)Tp * B * m->rnn_in(); }
    else if (n == "text") { p = m->text.p; ne = (int64_t)L * B * H2; }
    else if (n == "key") { p = m->key.p; ne = (int64_t)L * B * H2; }
    else if (n.compare(0, 3, "rnn") == 0) {
        int i = atoi(n.c_str() + 3);
        if (i == m->cfg.layers - 1) { p = m->xraw.p; ne = (int64_t)Tp * B * H2; }
        else if (m->taps && i >= 0 && i < (int)m->tap_rnn.size()) { p = m->tap_rnn[i].p; ne = (int64_t)Tp * B * H2; }
    }
    if (numel) *numel = ne;
    return p;
}

extern "C" int mdd_tap_copy(mdd_model *m, const char *name, float *dst_dev, int64_t capacity, void *stream) {
    int64_t n = 0;
    const float *p = mdd_tap(m, name, &n);
    if (!p || !dst_dev) { set_error("mdd_tap_copy: no tap named '%s' (taps enabled?)", name ? name : "(null)"); return MDD_ERR_ARG; }
    if (capacity < n) { set_error("mdd_tap_copy: capacity %lld < %lld", (long long)capacity, (long long)n); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipMemcpyAsync(dst_dev, p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MDD_OK;
}

extern "C" int mdd_sync(mdd_model *m, void *stream) {
    if (!m) { set_error("null model"); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int flag = 0;
    MDD_HIP_CHECK(hipMemcpy(&flag, m->err_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag == 2) {
        MDD_HIP_CHECK(hipMemset(m->err_flag, 0, sizeof(int)));
        set_error("persistent BiLSTM kernel timed out waiting for its team (grid not fully resident?); set MDD_LSTM=step");
        return MDD_ERR_HIP;
    }
    if (flag) {
        MDD_HIP_CHECK(hipMemset(m->err_flag, 0, sizeof(int)));
        set_error("index out of range in self");  // the message of the IndexError nn.Embedding raises
        return MDD_ERR_ARG;
    }
    return MDD_OK;
}
