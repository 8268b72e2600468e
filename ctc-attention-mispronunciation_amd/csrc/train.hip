// The training step behind the C ABI: CTC_Model.forward in train mode and its backward (BASELINE config 5).
//
// Reference: run_epoch, AA/steps/train_ctc.py:28-105 -- `out = model(inputs, trans)` in train mode (AA/models/model_ctc.py:160-223:
// BatchNorm on batch statistics with running-statistics update, Dropout(p) behind each LayerCNN and BatchRNN), nn.CTCLoss(sum)/B
// (mdd_ctc_loss), loss.backward() (autograd), optimizer.step() (torch.optim.Adam lr 1e-3, weight_decay 5e-4; train_ctc.py:187).
//
// Parameters stay where the caller keeps them (the drop-in CTC_Model's torch Parameters): every call receives the device
// pointers of the 55 float tensors of the state_dict in the order mdd_train_tensor_info() reports, and mdd_train_backward
// writes one gradient tensor per parameter.  The handle owns the saved activations of the last forward.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "train.h"

struct mdd_train_ws;
namespace mdd {

struct TInfo { std::string key; int64_t numel; int is_buffer; };

struct Buf {
    float *p = nullptr; size_t cap = 0;
    int need(size_t n) {
        if (cap >= n) return MDD_OK;
        if (p) MDD_HIP_CHECK(hipFree(p));
        p = nullptr; cap = 0;
        MDD_HIP_CHECK(hipMalloc((void **)&p, n * sizeof(float)));
        cap = n;
        return MDD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace mdd

// state of the training path of one handle
struct mdd_train_ws {
    mdd_config cfg;
    int device = 0;
    std::vector<mdd::TInfo> info;
    int B = 0, T = 0, L = 0;          // shapes of the saved forward
    float p_drop = 0.f;
    // saved activations / scratch
    mdd::Buf z0, a0, col1, w1r, z1, a1, seq0, gx, dgx, hb, cb, emb, text, key, att, cat, ycat, logits, logp;
    mdd::Buf stats;                    // per-site mean / invstd
    std::vector<mdd::Buf> xin, hraw, pd, gates, cst, wihp, whhp, whht;    // per rnn layer (index layers = the text encoder)
    mdd::Buf tbias;
    mdd::Buf d_a, d_b, d_c, part, dtext, dkey;      // backward temporaries
    mdd::Buf whhs, hx;                 // flagged variant: W_hh' as hi/lo planes and the h exchange buffer of the persistent layer kernel
    unsigned int *sync_words = nullptr;
    bool persist_ok = false;           // the device can hold the persistent layer kernel's grid
    mdd::Buf xs_a, xs_b;               // split-bf16 operand planes of the flagged variant's GEMMs (hi plane, then lo plane)
    int precision = 0;                 // 0: exact fp32 MFMA everywhere (the reference trains in fp32); 1: the large contractions as split-bf16 x3
    int *err_flag = nullptr;           // set by the embedding gather on an id outside the table
    mdd::Buf masks;                    // generated dropout masks (bytes)
    std::vector<const unsigned char *> mask_ptr;
    const unsigned char *mask_rows[2] = {nullptr, nullptr};   // the two conv sites' masks in channels-last order
    mdd::Buf maskt;
    double *dacc = nullptr;            // fp64 column sums
    const int64_t *ids = nullptr;      // canonical ids of the saved forward (caller memory, must stay valid until backward)
    const float *x = nullptr;
    ~mdd_train_ws() {
        mdd::Buf *all[] = {&z0, &a0, &col1, &w1r, &z1, &a1, &seq0, &gx, &dgx, &hb, &cb, &emb, &text, &key, &att, &cat, &ycat, &logits, &logp, &stats,
                           &tbias, &d_a, &d_b, &d_c, &part, &masks, &dtext, &dkey, &xs_a, &xs_b, &whhs, &hx, &maskt};
        for (auto *b : all) b->release();
        for (auto *v : {&xin, &hraw, &pd, &gates, &cst, &wihp, &whhp, &whht}) for (auto &b : *v) b.release();
        if (dacc) (void)hipFree(dacc);
        if (err_flag) (void)hipFree(err_flag);
        if (sync_words) (void)hipFree(sync_words);
    }
};

namespace mdd {

static int W1_of(const mdd_config &c) { return (c.feat + 2 - 3) / 2 + 1; }
static int W2_of(const mdd_config &c) { return (W1_of(c) + 2 - 3) / 2 + 1; }

static void build_info(mdd_train_ws *w) {
    const mdd_config &c = w->cfg;
    const int ch = c.channels, H = c.hidden, Kin0 = ch * W2_of(c);
    auto add = [&](const std::string &k, int64_t n, int buf = 0) { w->info.push_back(TInfo{k, n, buf}); };
    auto bn = [&](const std::string &p, int n) { add(p + ".weight", n); add(p + ".bias", n); add(p + ".running_mean", n, 1); add(p + ".running_var", n, 1); };
    add("conv.0.conv.weight", (int64_t)ch * 9); add("conv.0.conv.bias", ch); bn("conv.0.batch_norm", ch);
    add("conv.1.conv.weight", (int64_t)ch * ch * 9); add("conv.1.conv.bias", ch); bn("conv.1.batch_norm", ch);
    for (int n = 0; n < c.layers; n++) {
        const std::string r = "rnns." + std::to_string(n);
        const int K = n == 0 ? Kin0 : 2 * H;
        if (n > 0) bn(r + ".batch_norm", 2 * H);
        for (const char *sfx : {"", "_reverse"}) {
            add(r + ".rnn.weight_ih_l0" + sfx, (int64_t)4 * H * K);
            add(r + ".rnn.weight_hh_l0" + sfx, (int64_t)4 * H * H);
        }
    }
    add("embeds.weight", (int64_t)c.emb_rows * c.emb_dim);
    for (const char *sfx : {"", "_reverse"}) {
        add(std::string("lstm_embeds.weight_ih_l0") + sfx, (int64_t)4 * H * c.emb_dim);
        add(std::string("lstm_embeds.weight_hh_l0") + sfx, (int64_t)4 * H * H);
        add(std::string("lstm_embeds.bias_ih_l0") + sfx, 4 * H);
        add(std::string("lstm_embeds.bias_hh_l0") + sfx, 4 * H);
    }
    add("score.weight", (int64_t)4 * H * H);
    bn("fc.0", 4 * H);
    add("fc.1.weight", (int64_t)c.num_class * 4 * H);
}

static int idx(const mdd_train_ws *w, const std::string &key) {
    for (size_t i = 0; i < w->info.size(); i++) if (w->info[i].key == key) return (int)i;
    return -1;
}

// C[M,N] = A^T . B over K rows (A stored [K, M...] with lda, B stored [K, N...] with ldb): weight gradients.  Few output tiles and a
// long contraction -> split-K into `part` + a reduction, so the whole chip works on it.
static int gemm_tn(mdd_train_ws *w, const float *A, int lda, const float *Bm, int ldb, float *C, int M, int N, int K, hipStream_t st) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int nsplit = 1;
    if (tiles < 256 && K > 1024) nsplit = std::min(64, std::max(1, std::min(K / 512, 512 / tiles)));
    if (nsplit <= 1) return launch_gemm_f32(true, true, A, Bm, nullptr, C, M, N, K, lda, ldb, N, 1, 0, 0, 0, false, st);
    const int kc = (K + nsplit - 1) / nsplit, parts = (K + kc - 1) / kc;
    if (int rc = w->part.need((size_t)parts * M * N)) return rc;
    if (int rc = launch_gemm_f32(true, true, A, Bm, nullptr, w->part.p, M, N, K, lda, ldb, N, 1, 0, 0, (long)M * N, false, st, kc)) return rc;
    return launch_reduce_parts(w->part.p, parts, (size_t)M * N, C, st);
}

// C[M,N] = opA . opB^T for the large contractions of the step; opA[m,k] = ta ? A[k*lda + m] : A[m*lda + k], opB[n,k] likewise.
// precision 0: exact fp32 (gemm_f32 / split-K for the weight gradients).  precision 1 and a problem large enough to fill 256 x 256
// tiles: both operands are written as bf16 hi/lo planes with the contraction along their rows' contiguous axis (transposed on the way
// when the operand is stored [K, *]) and the product runs on the bf16 matrix cores (3 MFMA flops per flop, fp32 accumulate).
static int gemm_big(mdd_train_ws *w, bool ta, bool tb, const float *A, int lda, const float *Bm, int ldb, const float *bias, float *C, int ldc, int M, int N,
                    int K, hipStream_t st) {
    const bool x3 = w->precision == 1 && M >= 256 && N >= 256 && K >= 256 && ldc % 4 == 0 && (size_t)M * N * K >= ((size_t)1 << 30);
    if (!x3) {
        if (ta && tb && !bias && ldc == N) return gemm_tn(w, A, lda, Bm, ldb, C, M, N, K, st);
        return launch_gemm_f32(ta, tb, A, Bm, bias, C, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, false, st);
    }
    // few output tiles and a long contraction (the weight gradients): the K axis is cut into S chunks that run as a batch of partial
    // products (the planes' K axis is contiguous, so chunk s starts s*Kc elements into every row), summed afterwards.
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int S = 1;
    if (tiles < 256 && K > 2048 && !bias && ldc == N) S = std::min(16, std::max(1, std::min(K / 1024, 512 / tiles)));
    const int Kc = ((K + S - 1) / S + 31) / 32 * 32, Kp = S * Kc;
    if (int rc = w->xs_a.need((size_t)M * Kp)) return rc;
    if (int rc = w->xs_b.need((size_t)N * Kp)) return rc;
    SplitPtr sa{reinterpret_cast<unsigned short *>(w->xs_a.p), reinterpret_cast<unsigned short *>(w->xs_a.p) + (size_t)M * Kp};
    SplitPtr sb{reinterpret_cast<unsigned short *>(w->xs_b.p), reinterpret_cast<unsigned short *>(w->xs_b.p) + (size_t)N * Kp};
    if (int rc = ta ? launch_transpose_split(A, lda, K, M, Kp, sa.hi, sa.lo, st) : launch_split_rows(A, lda, (size_t)M, K, Kp, sa.hi, sa.lo, st)) return rc;
    if (int rc = tb ? launch_transpose_split(Bm, ldb, K, N, Kp, sb.hi, sb.lo, st) : launch_split_rows(Bm, ldb, (size_t)N, K, Kp, sb.hi, sb.lo, st)) return rc;
    if (S == 1) return launch_gemm_bf16x3(sa, sb, bias, C, nullptr, M, N, Kp, Kp, Kp, ldc, 1, 0, 0, 0, st);
    if (int rc = w->part.need((size_t)S * M * N)) return rc;
    if (int rc = launch_gemm_bf16x3(sa, sb, nullptr, w->part.p, nullptr, M, N, Kc, Kp, Kp, N, S, Kc, Kc, (long)M * N, st)) return rc;
    return launch_reduce_parts(w->part.p, S, (size_t)M * N, C, st);
}


// One bidirectional layer of the training forward (gates and cell states saved for the backward pass).  Exact mode: one launch per
// step, fp32 MFMA.  Flagged split-bf16 variant on a device that holds the persistent grid: the decode path's layer kernel (one
// launch, W_hh' resident in registers as bf16 hi/lo fragments, h exchanged inside 8-workgroup teams) with the saves added.
static int lstm_forward_layer(mdd_train_ws *w, LstmStepArgs &a, hipStream_t st) {
    const int H = a.H;
    if (w->precision == 1 && w->persist_ok && (H == 384 || H == 256) && a.B <= 512) {
        const size_t nW = (size_t)8 * H * H;
        if (int rc = w->whhs.need(nW)) return rc;
        unsigned short *hi = reinterpret_cast<unsigned short *>(w->whhs.p), *lo = hi + nW;
        if (int rc = launch_split_rows(a.whh, H, (size_t)8 * H, H, H, hi, lo, st)) return rc;
        a.whh_split = SplitPtr{hi, lo};
        a.out = nullptr;
        if (int rc = w->hx.need((size_t)2 * 32 * granule_bg(a.B) * H * 2 + 64 + 256 * 6 * 2)) return rc;
        return launch_lstm_layer_granule(a, reinterpret_cast<unsigned short *>(w->hx.p), w->sync_words, w->err_flag + 1, st);
    }
    return launch_lstm_layer_train(a, st);
}

}  // namespace mdd

using namespace mdd;

extern "C" int mdd_train_set_precision(mdd_train_ws *w, int32_t mode) {
    if (!w || (mode != 0 && mode != 1)) { set_error("mdd_train_set_precision: mode must be 0 (exact fp32) or 1 (split-bf16 x3 contractions)"); return MDD_ERR_ARG; }
    w->precision = mode;
    return MDD_OK;
}

extern "C" int mdd_train_create(const mdd_config *cfg, int device, mdd_train_ws **out) {
    if (!cfg || !out) { set_error("mdd_train_create: null argument"); return MDD_ERR_ARG; }
    if (cfg->hidden <= 0 || cfg->hidden % 4 || cfg->layers < 1 || (cfg->channels != 32 && cfg->channels != 4) || cfg->emb_dim % 4) {
        set_error("mdd_train_create: unsupported geometry"); return MDD_ERR_ARG;
    }
    MDD_HIP_CHECK(hipSetDevice(device));
    mdd_train_ws *w = new mdd_train_ws();
    w->cfg = *cfg; w->device = device;
    { const char *pr = getenv("MDD_TRAIN_PRECISION"); if (pr && (!strcmp(pr, "bf16x3") || !strcmp(pr, "1"))) w->precision = 1; }
    build_info(w);
    const int nl = cfg->layers + 1;
    w->xin.resize(nl); w->hraw.resize(nl); w->pd.resize(nl); w->gates.resize(nl); w->cst.resize(nl); w->wihp.resize(nl); w->whhp.resize(nl); w->whht.resize(nl);
    if (hipMalloc((void **)&w->dacc, sizeof(double) * 2 * 8192) != hipSuccess || hipMalloc((void **)&w->err_flag, 2 * sizeof(int)) != hipSuccess ||
        hipMemset(w->err_flag, 0, 2 * sizeof(int)) != hipSuccess || hipMalloc((void **)&w->sync_words, 32 * sizeof(unsigned int)) != hipSuccess) {
        delete w; set_error("mdd_train_create: out of memory"); return MDD_ERR_NOMEM;
    }
    if (int rc = init_gemm_attributes()) { delete w; return rc; }
    if (int rc = init_granule_attributes()) { delete w; return rc; }
    if (int rc = init_conv1_attributes()) { delete w; return rc; }
    { int n_cu = 0; w->persist_ok = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && persistent_grid_fits(n_cu) &&
                                     !(getenv("MDD_LSTM") && !strcmp(getenv("MDD_LSTM"), "step")); }
    if (8 * cfg->hidden > 8192) { delete w; set_error("mdd_train_create: hidden too large for the statistics scratch"); return MDD_ERR_ARG; }
    *out = w;
    return MDD_OK;
}
extern "C" void mdd_train_destroy(mdd_train_ws *w) { if (w) { (void)hipSetDevice(w->device); (void)hipDeviceSynchronize(); delete w; } }
extern "C" int32_t mdd_train_num_tensors(mdd_train_ws *w) { return w ? (int32_t)w->info.size() : 0; }
extern "C" int mdd_train_tensor_info(mdd_train_ws *w, int32_t i, char *key, int32_t cap, int64_t *numel, int32_t *is_buffer) {
    if (!w || i < 0 || i >= (int)w->info.size() || !key || cap < 2) { set_error("mdd_train_tensor_info: bad argument"); return MDD_ERR_ARG; }
    snprintf(key, cap, "%s", w->info[i].key.c_str());
    if (numel) *numel = w->info[i].numel;
    if (is_buffer) *is_buffer = w->info[i].is_buffer;
    return MDD_OK;
}
// bytes of each dropout mask the caller may pass (sites: conv0, conv1, rnn 0..layers-1), in that order
extern "C" int32_t mdd_train_num_masks(mdd_train_ws *w) { return w ? 2 + w->cfg.layers : 0; }
extern "C" int64_t mdd_train_mask_bytes(mdd_train_ws *w, int32_t site, int32_t B, int32_t T) {
    if (!w || site < 0 || site >= 2 + w->cfg.layers) return -1;
    const mdd_config &c = w->cfg;
    if (site == 0) return (int64_t)B * c.channels * T * W1_of(c);
    if (site == 1) return (int64_t)B * c.channels * (T / 2) * W2_of(c);
    return (int64_t)(T / 2) * B * 2 * c.hidden;
}

#define P(key) (tensors[idx(w, key)])
#define TRY(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

static int train_forward_enqueue(mdd_train_ws *w, float *const *tensors, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                                 const uint8_t *const *masks, uint64_t seed, float p_drop, float *logp_dev, void *stream);
extern "C" int mdd_train_forward(mdd_train_ws *w, float *const *tensors, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                                 const uint8_t *const *masks, uint64_t seed, float p_drop, float *logp_dev, void *stream) {
    if (!w) { set_error("mdd_train_forward: null handle"); return MDD_ERR_ARG; }
    const bool gated = w->precision == 1 && w->persist_ok;      // the forward then contains persistent launches: one at a time per device
    bool held = false;
    if (gated) { MDD_HIP_CHECK(hipSetDevice(w->device)); if (int rc = device_gate_enter(w->device, (hipStream_t)stream, &held)) return rc; }
    const int rc = train_forward_enqueue(w, tensors, x_dev, B, T, x1_dev, L, masks, seed, p_drop, logp_dev, stream);
    return gated ? device_gate_leave(w->device, (hipStream_t)stream, held, rc) : rc;
}
static int train_forward_enqueue(mdd_train_ws *w, float *const *tensors, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                                 const uint8_t *const *masks, uint64_t seed, float p_drop, float *logp_dev, void *stream) {
    if (!w || !tensors || !x_dev || !x1_dev || !logp_dev || B <= 0 || T < 2 || (T & 1) || L <= 0 || p_drop < 0.f || p_drop >= 1.f) {
        set_error("mdd_train_forward: bad argument"); return MDD_ERR_ARG;
    }
    MDD_HIP_CHECK(hipSetDevice(w->device));
    hipStream_t st = (hipStream_t)stream;
    const mdd_config &c = w->cfg;
    const int ch = c.channels, H = c.hidden, H2 = 2 * H, G2 = 8 * H, Tp = T / 2, W1 = W1_of(c), W2 = W2_of(c), Kin0 = ch * W2, nl = c.layers, E = c.emb_dim, C = c.num_class;
    const size_t R0 = (size_t)B * T * W1, R1 = (size_t)B * Tp * W2, R = (size_t)Tp * B, Rt = (size_t)L * B;
    const float scale = 1.f / (1.f - p_drop), eps = c.bn_eps, mom = 0.1f;
    w->B = B; w->T = T; w->L = L; w->p_drop = p_drop; w->ids = x1_dev; w->x = x_dev;
    // ---- buffers
    TRY(w->z0.need(R0 * ch)); TRY(w->a0.need(R0 * ch)); TRY(w->w1r.need((size_t)ch * 9 * ch));
    const bool direct1 = ch == 32 && W1 <= 128 && !getenv("MDD_TRAIN_CONV1_IM2COL");   // conv1 as direct kernels (train_conv1.hip); else im2col + GEMM
    if (!direct1) TRY(w->col1.need(R1 * 9 * ch));
    TRY(w->z1.need(R1 * ch)); TRY(w->a1.need(R1 * ch)); TRY(w->seq0.need(R * Kin0));
    TRY(w->gx.need(std::max(R, Rt) * G2)); TRY(w->hb.need((size_t)4 * B * H)); TRY(w->cb.need((size_t)2 * B * H));
    TRY(w->emb.need(Rt * E)); TRY(w->key.need(Rt * H2)); TRY(w->att.need((size_t)B * Tp * L));
    TRY(w->cat.need(R * 2 * H2)); TRY(w->ycat.need(R * 2 * H2)); TRY(w->logits.need(R * C)); TRY(w->logp.need(R * C));
    TRY(w->stats.need((size_t)2 * (2 * ch + nl * H2 + 2 * H2) + 64)); TRY(w->tbias.need(G2));
    for (int n = 0; n <= nl; n++) {
        const size_t rows = n < nl ? R : Rt;
        const int K = n == nl ? E : (n == 0 ? Kin0 : H2);
        if (n > 0 && n < nl) TRY(w->xin[n].need(rows * K));
        TRY(w->hraw[n].need(rows * H2)); if (n < nl) TRY(w->pd[n].need(rows * H2));
        TRY(w->gates[n].need(rows * 2 * H * 4)); TRY(w->cst[n].need(rows * 2 * H));
        TRY(w->wihp[n].need((size_t)G2 * K)); TRY(w->whhp[n].need((size_t)G2 * H)); TRY(w->whht[n].need((size_t)G2 * H));
    }
    // ---- dropout masks: the caller's, or drawn here
    const int nm = 2 + nl;
    w->mask_ptr.assign(nm, nullptr);
    if (p_drop > 0.f) {
        size_t tot = 0;
        std::vector<size_t> off(nm);
        for (int s = 0; s < nm; s++) { off[s] = tot; tot += ((size_t)mdd_train_mask_bytes(w, s, B, T) + 15) & ~(size_t)15; }
        if (!masks) TRY(w->masks.need((tot + 3) / 4));
        for (int s = 0; s < nm; s++) {
            if (masks && masks[s]) w->mask_ptr[s] = masks[s];
            else {
                if (masks) { set_error("mdd_train_forward: mask %d missing", s); return MDD_ERR_ARG; }
                unsigned char *m = reinterpret_cast<unsigned char *>(w->masks.p) + off[s];
                TRY(launch_dropout_mask(m, (size_t)mdd_train_mask_bytes(w, s, B, T), seed, (unsigned)s, p_drop, st));
                w->mask_ptr[s] = m;
            }
        }
    }
    w->mask_rows[0] = w->mask_rows[1] = nullptr;
    if (p_drop > 0.f) {   // the conv sites' kernels walk channels-last rows: their masks once in that order
        const size_t n0 = (R0 * ch + 15) & ~(size_t)15, n1 = R1 * ch;
        TRY(w->maskt.need((n0 + n1 + 3) / 4));
        unsigned char *m0 = reinterpret_cast<unsigned char *>(w->maskt.p), *m1 = m0 + n0;
        TRY(launch_mask_rows(w->mask_ptr[0], m0, B, ch, T * W1, st));
        TRY(launch_mask_rows(w->mask_ptr[1], m1, B, ch, Tp * W2, st));
        w->mask_rows[0] = m0; w->mask_rows[1] = m1;
    }
    float *mean = w->stats.p, *invstd = w->stats.p + (2 * ch + nl * H2 + 2 * H2);
    // ---- conv0 -> BN -> ReLU -> Dropout
    TRY(launch_conv0_train_fwd(x_dev, P("conv.0.conv.weight"), P("conv.0.conv.bias"), w->z0.p, B, T, c.feat, ch, st));
    BnSite s0{w->mask_rows[0], scale};
    TRY(launch_bn_train_fwd(w->z0.p, R0, ch, P("conv.0.batch_norm.weight"), P("conv.0.batch_norm.bias"), eps, mom, P("conv.0.batch_norm.running_mean"),
                            P("conv.0.batch_norm.running_var"), w->dacc, mean, invstd, &s0, w->a0.p, st));
    // ---- conv1 (im2col + GEMM) -> BN -> ReLU -> Dropout -> [T',B,ch*W2]
    TRY(launch_pack_w1(P("conv.1.conv.weight"), w->w1r.p, ch, true, st));
    if (direct1) {
        TRY(launch_conv1_fwd_direct(w->a0.p, w->w1r.p, P("conv.1.conv.bias"), w->z1.p, B, T, W1, W2, ch, st));
    } else {
        TRY(launch_im2col1(w->a0.p, w->col1.p, B, T, W1, W2, ch, st));
        TRY(launch_gemm_f32(false, false, w->col1.p, w->w1r.p, P("conv.1.conv.bias"), w->z1.p, (int)R1, ch, 9 * ch, 9 * ch, 9 * ch, ch, 1, 0, 0, 0, false, st));
    }
    BnSite s1{w->mask_rows[1], scale};
    TRY(launch_bn_train_fwd(w->z1.p, R1, ch, P("conv.1.batch_norm.weight"), P("conv.1.batch_norm.bias"), eps, mom, P("conv.1.batch_norm.running_mean"),
                            P("conv.1.batch_norm.running_var"), w->dacc, mean + ch, invstd + ch, &s1, w->a1.p, st));
    TRY(launch_cnn_seq(w->a1.p, w->seq0.p, B, Tp, W2, ch, true, st));
    // ---- BatchRNN x layers
    for (int n = 0; n < nl; n++) {
        const std::string r = "rnns." + std::to_string(n);
        const int K = n == 0 ? Kin0 : H2;
        const float *xin = w->seq0.p;
        if (n > 0) {
            TRY(launch_bn_train_fwd(w->pd[n - 1].p, R, H2, P(r + ".batch_norm.weight"), P(r + ".batch_norm.bias"), eps, mom, P(r + ".batch_norm.running_mean"),
                                    P(r + ".batch_norm.running_var"), w->dacc, mean + 2 * ch + (n - 1) * H2, invstd + 2 * ch + (n - 1) * H2, nullptr, w->xin[n].p, st));
            xin = w->xin[n].p;
        }
        TRY(launch_pack_gates(P(r + ".rnn.weight_ih_l0"), P(r + ".rnn.weight_ih_l0_reverse"), w->wihp[n].p, H, K, st));
        TRY(launch_pack_gates(P(r + ".rnn.weight_hh_l0"), P(r + ".rnn.weight_hh_l0_reverse"), w->whhp[n].p, H, H, st));
        TRY(gemm_big(w, false, false, xin, K, w->wihp[n].p, K, nullptr, w->gx.p, G2, (int)R, G2, K, st));
        LstmStepArgs a;
        a.gx = w->gx.p; a.whh = w->whhp[n].p; a.hbuf = w->hb.p; a.cbuf = w->cb.p; a.out = w->hraw[n].p; a.out_raw = w->hraw[n].p;
        a.out_split = SplitPtr{nullptr, nullptr}; a.oscale = nullptr; a.oshift = nullptr; a.T = Tp; a.B = B; a.H = H;
        a.whh_split = SplitPtr{nullptr, nullptr}; a.hsplit = nullptr; a.packed = 0; a.gates_save = w->gates[n].p; a.c_save = w->cst[n].p;
        TRY(lstm_forward_layer(w, a, st));
        TRY(launch_dropout_rows(w->hraw[n].p, w->mask_ptr[2 + n], scale, R * H2, w->pd[n].p, st));
    }
    const float *X = w->pd[nl - 1].p;
    // ---- text encoder: Embedding -> BiLSTM (bias) ; key = score(text)
    TRY(launch_embed(P("embeds.weight"), c.emb_rows, E, x1_dev, B, L, w->emb.p, SplitPtr{nullptr, nullptr}, w->err_flag, st));
    TRY(launch_pack_gates(P("lstm_embeds.weight_ih_l0"), P("lstm_embeds.weight_ih_l0_reverse"), w->wihp[nl].p, H, E, st));
    TRY(launch_pack_gates(P("lstm_embeds.weight_hh_l0"), P("lstm_embeds.weight_hh_l0_reverse"), w->whhp[nl].p, H, H, st));
    TRY(launch_pack_gates(P("lstm_embeds.bias_ih_l0"), P("lstm_embeds.bias_ih_l0_reverse"), w->tbias.p, H, 1, st));
    TRY(w->d_a.need(G2));
    TRY(launch_pack_gates(P("lstm_embeds.bias_hh_l0"), P("lstm_embeds.bias_hh_l0_reverse"), w->d_a.p, H, 1, st));
    TRY(launch_copy_cols(w->d_a.p, G2, 0, w->tbias.p, G2, 0, 1, G2, true, st));
    TRY(gemm_big(w, false, false, w->emb.p, E, w->wihp[nl].p, E, w->tbias.p, w->gx.p, G2, (int)Rt, G2, E, st));
    {
        LstmStepArgs a;
        a.gx = w->gx.p; a.whh = w->whhp[nl].p; a.hbuf = w->hb.p; a.cbuf = w->cb.p; a.out = w->hraw[nl].p; a.out_raw = w->hraw[nl].p;
        a.out_split = SplitPtr{nullptr, nullptr}; a.oscale = nullptr; a.oshift = nullptr; a.T = L; a.B = B; a.H = H;
        a.whh_split = SplitPtr{nullptr, nullptr}; a.hsplit = nullptr; a.packed = 0; a.gates_save = w->gates[nl].p; a.c_save = w->cst[nl].p;
        TRY(lstm_forward_layer(w, a, st));
    }
    TRY(launch_gemm_f32(false, false, w->hraw[nl].p, P("score.weight"), nullptr, w->key.p, (int)Rt, H2, H2, H2, H2, H2, 1, 0, 0, 0, false, st));
    // ---- attention: S = X.key^T, softmax over L (no scale, no mask), ctx = A.text, cat(X, ctx)
    TRY(launch_gemm_f32(false, false, X, w->key.p, nullptr, w->att.p, Tp, L, H2, B * H2, B * H2, L, B, H2, H2, (long)Tp * L, false, st));
    TRY(launch_softmax_rows(w->att.p, (size_t)B * Tp, L, w->att.p, false, st));
    TRY(launch_copy_cols(X, H2, 0, w->cat.p, 2 * H2, 0, R, H2, false, st));
    TRY(launch_gemm_f32(false, true, w->att.p, w->hraw[nl].p, nullptr, w->cat.p + H2, Tp, H2, L, L, B * H2, B * 2 * H2, B, (long)Tp * L, H2, 2 * H2, false, st));
    // ---- fc: BatchNorm1d(4H) -> Linear(4H -> C, no bias) -> log-softmax
    TRY(launch_bn_train_fwd(w->cat.p, R, 2 * H2, P("fc.0.weight"), P("fc.0.bias"), eps, mom, P("fc.0.running_mean"), P("fc.0.running_var"), w->dacc,
                            mean + 2 * ch + nl * H2, invstd + 2 * ch + nl * H2, nullptr, w->ycat.p, st));
    {   // [R, 1536] x [45, 1536]^T: one column tile, 63 row tiles at R = 8000 -> the contraction is cut into partial products so the chip is busy
        const int Kfc = 2 * H2, tiles = (int)((R + 127) / 128) * ((C + 127) / 128);
        const int ks = (tiles < 128 && Kfc >= 1024) ? 256 : 0;
        if (ks) {
            const int parts = (Kfc + ks - 1) / ks;
            TRY(w->part.need((size_t)parts * R * C));
            TRY(launch_gemm_f32(false, false, w->ycat.p, P("fc.1.weight"), nullptr, w->part.p, (int)R, C, Kfc, Kfc, Kfc, C, 1, 0, 0, (long)R * C, false, st, ks));
            TRY(launch_reduce_parts(w->part.p, parts, R * C, w->logits.p, st));
        } else {
            TRY(launch_gemm_f32(false, false, w->ycat.p, P("fc.1.weight"), nullptr, w->logits.p, (int)R, C, Kfc, Kfc, Kfc, C, 1, 0, 0, 0, false, st));
        }
    }
    TRY(launch_softmax_rows(w->logits.p, R, C, w->logp.p, true, st));
    MDD_HIP_CHECK(hipMemcpyAsync(logp_dev, w->logp.p, R * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    return MDD_OK;
}

// one BiLSTM's backward: dout [rows,2H] -> DG (in w->dgx), weight gradients into the reference-layout tensors, dxin (optional)
static int lstm_backward(mdd_train_ws *w, int n, int Tn, int B, int K, const float *dout, const float *xin, float *g_ih_f, float *g_ih_r, float *g_hh_f,
                         float *g_hh_r, float *dxin, hipStream_t st) {
    const int H = w->cfg.hidden, H2 = 2 * H, G2 = 8 * H, G = 4 * H;
    const size_t rows = (size_t)Tn * B;
    TRY(launch_transpose_whh(w->whhp[n].p, w->whht[n].p, H, st));
    LstmBwdArgs a;
    a.dout = dout; a.gates = w->gates[n].p; a.cst = w->cst[n].p; a.whhT = w->whht[n].p; a.dg = w->dgx.p; a.dc = w->cb.p; a.T = Tn; a.B = B; a.H = H;
    if (w->precision == 1 && w->persist_ok && (H == 384 || H == 256) && B <= 256) {   // flagged variant: the whole recurrence in one persistent launch
        const size_t nW = (size_t)8 * H * H;
        TRY(w->whhs.need(nW));
        unsigned short *hi = reinterpret_cast<unsigned short *>(w->whhs.p), *lo = hi + nW;
        TRY(launch_split_rows(w->whht[n].p, G, (size_t)2 * H, G, G, hi, lo, st));
        TRY(w->hx.need(lstm_bwd_granule_hx_bytes(H) / 4));
        TRY(launch_lstm_bwd_granule(dout, a.gates, a.cst, SplitPtr{hi, lo}, a.dg, Tn, B, H, reinterpret_cast<unsigned short *>(w->hx.p), w->sync_words, w->err_flag + 1, st));
    } else {
        TRY(launch_lstm_bwd(a, st));
    }
    // dWih' [2*4H, K] = DG^T . xin ;  dWhh'[d] [4H, H] = DG_d^T . h_prev_d  (h_prev = the layer's raw output one step earlier in that direction)
    TRY(w->d_b.need((size_t)G2 * std::max(K, H)));
    TRY(gemm_big(w, true, true, w->dgx.p, G2, xin, K, nullptr, w->d_b.p, K, G2, K, (int)rows, st));
    TRY(launch_unpack_gates(w->d_b.p, g_ih_f, g_ih_r, H, K, st));
    if (Tn > 1) {
        const size_t rows1 = (size_t)(Tn - 1) * B;
        TRY(gemm_big(w, true, true, w->dgx.p + (size_t)B * G2, G2, w->hraw[n].p, H2, nullptr, w->d_b.p, H, G, H, (int)rows1, st));                  // forward direction: t = 1.., h_{t-1}
        TRY(gemm_big(w, true, true, w->dgx.p + G, G2, w->hraw[n].p + (size_t)B * H2 + H, H2, nullptr, w->d_b.p + (size_t)G * H, H, G, H, (int)rows1, st));   // reverse: t = 0..T-2, h_{t+1}
    } else {
        MDD_HIP_CHECK(hipMemsetAsync(w->d_b.p, 0, sizeof(float) * G2 * H, st));
    }
    TRY(launch_unpack_gates(w->d_b.p, g_hh_f, g_hh_r, H, H, st));
    if (dxin) TRY(gemm_big(w, false, true, w->dgx.p, G2, w->wihp[n].p, K, nullptr, dxin, K, (int)rows, K, G2, st));
    return MDD_OK;
}

#define GR(key) (grads[idx(w, key)])

static int train_backward_enqueue(mdd_train_ws *w, float *const *tensors, const float *dlogp_dev, float *const *grads, void *stream);
extern "C" int mdd_train_backward(mdd_train_ws *w, float *const *tensors, const float *dlogp_dev, float *const *grads, void *stream) {
    if (!w) { set_error("mdd_train_backward: null handle"); return MDD_ERR_ARG; }
    const bool gated = w->precision == 1 && w->persist_ok;      // persistent launches inside: one at a time per device
    bool held = false;
    if (gated) { MDD_HIP_CHECK(hipSetDevice(w->device)); if (int rc = device_gate_enter(w->device, (hipStream_t)stream, &held)) return rc; }
    const int rc = train_backward_enqueue(w, tensors, dlogp_dev, grads, stream);
    return gated ? device_gate_leave(w->device, (hipStream_t)stream, held, rc) : rc;
}
static int train_backward_enqueue(mdd_train_ws *w, float *const *tensors, const float *dlogp_dev, float *const *grads, void *stream) {
    if (!w || !tensors || !dlogp_dev || !grads || w->B <= 0) { set_error("mdd_train_backward: bad argument (forward first)"); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipSetDevice(w->device));
    hipStream_t st = (hipStream_t)stream;
    const mdd_config &c = w->cfg;
    const int B = w->B, T = w->T, L = w->L;
    const int ch = c.channels, H = c.hidden, H2 = 2 * H, G2 = 8 * H, Tp = T / 2, W1 = W1_of(c), W2 = W2_of(c), Kin0 = ch * W2, nl = c.layers, E = c.emb_dim, C = c.num_class;
    const size_t R0 = (size_t)B * T * W1, R1 = (size_t)B * Tp * W2, R = (size_t)Tp * B, Rt = (size_t)L * B;
    const float scale = 1.f / (1.f - w->p_drop);
    float *mean = w->stats.p, *invstd = w->stats.p + (2 * ch + nl * H2 + 2 * H2);
    const float *X = w->pd[nl - 1].p;
    TRY(w->dgx.need(std::max(R, Rt) * G2));
    const size_t big = std::max(std::max(std::max(R * 2 * H2, R0 * ch), std::max(R * (size_t)Kin0, Rt * (size_t)E)), std::max((size_t)B * Tp * L, R1 * ch));
    TRY(w->d_a.need(big)); TRY(w->d_c.need(big)); TRY(w->dtext.need(Rt * H2)); TRY(w->dkey.need(Rt * H2));
    // ---- log-softmax, Linear, BatchNorm1d of the classifier
    float *dlogits = w->logits.p;                                   // logits are not needed again
    TRY(launch_softmax_bwd_rows(w->logp.p, dlogp_dev, R, C, dlogits, true, st));
    TRY(gemm_tn(w, dlogits, C, w->ycat.p, 2 * H2, GR("fc.1.weight"), C, 2 * H2, (int)R, st));
    float *dycat = w->d_a.p, *dcat = w->d_c.p;
    TRY(launch_gemm_f32(false, true, dlogits, P("fc.1.weight"), nullptr, dycat, (int)R, 2 * H2, C, C, 2 * H2, 2 * H2, 1, 0, 0, 0, false, st));
    TRY(launch_bn_train_bwd(w->cat.p, dycat, R, 2 * H2, P("fc.0.weight"), P("fc.0.bias"), mean + 2 * ch + nl * H2, invstd + 2 * ch + nl * H2, nullptr, w->dacc,
                            dcat, GR("fc.0.weight"), GR("fc.0.bias"), st));
    // ---- attention
    float *dX = w->ycat.p;                                          // [R, 2H]  (ycat is free now)
    TRY(launch_copy_cols(dcat, 2 * H2, 0, dX, H2, 0, R, H2, false, st));
    float *dS = w->d_a.p;                                           // [B][T'][L]
    TRY(launch_gemm_f32(false, false, dcat + H2, w->hraw[nl].p, nullptr, dS, Tp, L, H2, B * 2 * H2, B * H2, L, B, 2 * H2, H2, (long)Tp * L, false, st));     // dA = dctx . text^T
    float *dtext = w->dtext.p;                                      // [Rt, 2H]
    TRY(launch_gemm_f32(true, true, w->att.p, dcat + H2, nullptr, dtext, L, H2, Tp, L, B * 2 * H2, B * H2, B, (long)Tp * L, 2 * H2, H2, false, st));    // dtext = A^T . dctx
    TRY(launch_softmax_bwd_rows(w->att.p, dS, (size_t)B * Tp, L, dS, false, st));
    TRY(launch_gemm_f32(false, true, dS, w->key.p, nullptr, dX, Tp, H2, L, L, B * H2, B * H2, B, (long)Tp * L, H2, H2, true, st));                        // dX += dS . key
    float *dkey = w->dkey.p;                                        // [Rt, 2H]
    TRY(launch_gemm_f32(true, true, dS, X, nullptr, dkey, L, H2, Tp, L, B * H2, B * H2, B, (long)Tp * L, H2, H2, false, st));                             // dkey = dS^T . X
    TRY(gemm_tn(w, dkey, H2, w->hraw[nl].p, H2, GR("score.weight"), H2, H2, (int)Rt, st));
    TRY(launch_gemm_f32(false, true, dkey, P("score.weight"), nullptr, dtext, (int)Rt, H2, H2, H2, H2, H2, 1, 0, 0, 0, true, st));                       // dtext += dkey . Ws
    // ---- text encoder
    float *demb = w->d_c.p;
    TRY(lstm_backward(w, nl, L, B, E, dtext, w->emb.p, GR("lstm_embeds.weight_ih_l0"), GR("lstm_embeds.weight_ih_l0_reverse"), GR("lstm_embeds.weight_hh_l0"),
                      GR("lstm_embeds.weight_hh_l0_reverse"), demb, st));
    TRY(launch_col_sum(w->dgx.p, Rt, G2, w->dacc, w->tbias.p, st));            // packed bias gradient (u*4+g order)
    TRY(launch_unpack_gates(w->tbias.p, GR("lstm_embeds.bias_ih_l0"), GR("lstm_embeds.bias_ih_l0_reverse"), H, 1, st));
    TRY(launch_unpack_gates(w->tbias.p, GR("lstm_embeds.bias_hh_l0"), GR("lstm_embeds.bias_hh_l0_reverse"), H, 1, st));
    TRY(launch_embed_bwd(demb, w->ids, B, L, E, c.emb_rows, GR("embeds.weight"), st));
    // ---- BatchRNN layers, last to first
    float *dpd = dX;                                                // gradient at the layer's (post-dropout) output
    for (int n = nl - 1; n >= 0; n--) {
        const std::string r = "rnns." + std::to_string(n);
        const int K = n == 0 ? Kin0 : H2;
        float *dh = w->d_a.p;                                       // [R, 2H]
        TRY(launch_dropout_rows(dpd, w->mask_ptr[2 + n], scale, R * H2, dh, st));
        float *dxin = w->d_c.p;                                     // [R, K]
        TRY(lstm_backward(w, n, Tp, B, K, dh, n == 0 ? w->seq0.p : w->xin[n].p, GR(r + ".rnn.weight_ih_l0"), GR(r + ".rnn.weight_ih_l0_reverse"),
                          GR(r + ".rnn.weight_hh_l0"), GR(r + ".rnn.weight_hh_l0_reverse"), dxin, st));
        if (n > 0) {
            TRY(launch_bn_train_bwd(w->pd[n - 1].p, dxin, R, H2, P(r + ".batch_norm.weight"), P(r + ".batch_norm.bias"), mean + 2 * ch + (n - 1) * H2,
                                    invstd + 2 * ch + (n - 1) * H2, nullptr, w->dacc, w->ycat.p, GR(r + ".batch_norm.weight"), GR(r + ".batch_norm.bias"), st));
            dpd = w->ycat.p;
        }
    }
    // ---- conv1: relayout, BN/ReLU/Dropout, weight and input gradients
    float *da1 = w->a1.p;                                           // a1 is not needed again
    TRY(launch_cnn_seq(da1, w->d_c.p, B, Tp, W2, ch, false, st));
    BnSite s1{w->mask_rows[1], scale};
    float *dz1 = w->d_a.p;
    TRY(launch_bn_train_bwd(w->z1.p, da1, R1, ch, P("conv.1.batch_norm.weight"), P("conv.1.batch_norm.bias"), mean + ch, invstd + ch, &s1, w->dacc, dz1,
                            GR("conv.1.batch_norm.weight"), GR("conv.1.batch_norm.bias"), st));
    TRY(launch_col_sum(dz1, R1, ch, w->dacc, GR("conv.1.conv.bias"), st));
    TRY(w->d_b.need((size_t)ch * 9 * ch));
    float *da0 = w->d_c.p;
    const bool direct1 = ch == 32 && W1 <= 128 && !getenv("MDD_TRAIN_CONV1_IM2COL");
    if (direct1) {
        const int parts = conv1_wgrad_parts(B, T, W2);
        TRY(w->part.need((size_t)parts * ch * 9 * ch));
        TRY(launch_conv1_wgrad_direct(dz1, w->a0.p, w->part.p, B, T, W1, W2, ch, st));
        TRY(launch_reduce_parts(w->part.p, parts, (size_t)ch * 9 * ch, w->d_b.p, st));
        TRY(launch_pack_w1(w->d_b.p, GR("conv.1.conv.weight"), ch, false, st));
        TRY(launch_conv1_dgrad_direct(dz1, w->w1r.p, da0, B, T, W1, W2, ch, st));
    } else {
        TRY(gemm_tn(w, dz1, ch, w->col1.p, 9 * ch, w->d_b.p, ch, 9 * ch, (int)R1, st));
        TRY(launch_pack_w1(w->d_b.p, GR("conv.1.conv.weight"), ch, false, st));
        TRY(launch_gemm_f32(false, true, dz1, w->w1r.p, nullptr, w->col1.p, (int)R1, 9 * ch, ch, ch, 9 * ch, 9 * ch, 1, 0, 0, 0, false, st));               // dcol
        TRY(launch_col2im1(w->col1.p, da0, B, T, W1, W2, ch, st));
    }
    // ---- conv0
    BnSite s0{w->mask_rows[0], scale};
    float *dz0 = w->d_a.p;
    TRY(launch_bn_train_bwd(w->z0.p, da0, R0, ch, P("conv.0.batch_norm.weight"), P("conv.0.batch_norm.bias"), mean, invstd, &s0, w->dacc, dz0,
                            GR("conv.0.batch_norm.weight"), GR("conv.0.batch_norm.bias"), st));
    TRY(launch_conv0_train_bwd(w->x, dz0, w->dacc, GR("conv.0.conv.weight"), GR("conv.0.conv.bias"), B, T, c.feat, ch, st));
    return MDD_OK;
}

// wait for `stream`; reports an out-of-range canonical id seen by the last forward (the reference raises IndexError)
extern "C" int mdd_train_sync(mdd_train_ws *w, void *stream) {
    if (!w) { set_error("null handle"); return MDD_ERR_ARG; }
    MDD_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int flag[2] = {0, 0};
    MDD_HIP_CHECK(hipMemcpy(flag, w->err_flag, 2 * sizeof(int), hipMemcpyDeviceToHost));
    if (flag[0] || flag[1]) MDD_HIP_CHECK(hipMemset(w->err_flag, 0, 2 * sizeof(int)));
    if (flag[1]) { set_error("persistent BiLSTM layer kernel gave up waiting for its team (was another persistent launch resident on this device?)"); return MDD_ERR_HIP; }
    if (flag[0]) { set_error("index out of range in self"); return MDD_ERR_ARG; }
    return MDD_OK;
}

// torch.optim.Adam step over n tensors (AA/steps/train_ctc.py:187: lr, weight_decay as L2 added to the gradient); step counts from 1
extern "C" int mdd_adam_step(float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel, int32_t n,
                             int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay, void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || n < 0 || step < 1) { set_error("mdd_adam_step: bad argument"); return MDD_ERR_ARG; }
    for (int i = 0; i < n; i++)
        if (params[i] && grads[i] && numel[i] > 0 && (!exp_avg[i] || !exp_avg_sq[i])) { set_error("mdd_adam_step: state of tensor %d missing", i); return MDD_ERR_ARG; }
    return launch_adam_multi(params, grads, exp_avg, exp_avg_sq, numel, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}
