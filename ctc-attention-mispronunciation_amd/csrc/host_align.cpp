// Host-only entry points of the C ABI (no HIP): the integer work on either side of the GPU path.
//   mdd_stack_len / mdd_len_frames   length bookkeeping of the collate (AA/utils/data_loader.py:138-142,177; AA/infer.py:296-297)
//   mdd_align / mdd_align_batch      Decoder.wer = _edit_distance + printChanges (AA/utils/ctcDecoder.py:118-184)
//   mdd_eval_batch                   the TA / FR / FA / TR bookkeeping of AA/steps/test_ctc_nosil.py:33-60,218-298
// Kept free of HIP headers so that the same file also builds with g++ -fsanitize=address,undefined for the CPU test-suite
// (tests/test_host.py::test_host_entry_points_under_sanitizers; -DMDD_HOST_STANDALONE supplies set_error there).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

#include "../../include/mdd_hip.h"

namespace mdd {
#ifdef MDD_HOST_STANDALONE
static thread_local char g_err_host[512] = "";
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err_host, sizeof(g_err_host), fmt, ap); va_end(ap); }
#else
void set_error(const char *fmt, ...);
#endif
}  // namespace mdd
using mdd::set_error;
#ifdef MDD_HOST_STANDALONE
extern "C" const char *mdd_last_error(void) { return mdd::g_err_host; }
#endif

extern "C" int32_t mdd_stack_len(int32_t T_raw, int32_t skip, int32_t n_down) {
    int kept = (skip <= 1) ? T_raw : (T_raw + skip - 1) / skip;
    if (n_down > 1 && kept % n_down) kept += n_down - kept % n_down;
    return kept;
}


extern "C" int32_t mdd_len_frames(int32_t len, int32_t maxlen, int32_t t_out) {
    // float32 fraction (data_loader.py:177) times T_out in float32, truncated (infer.py:296-297)
    volatile float frac = (float)((double)len / (double)maxlen);
    volatile float prod = frac * (float)t_out;
    return (int32_t)prod;
}


// A10 -- host side, pure integer work on <= ~50 tokens per utterance
extern "C" int mdd_align(const int32_t *a, int32_t na, const int32_t *b, int32_t nb, int32_t *dist, uint8_t *ops,
                         int32_t *nops) {
    if (na < 0 || nb < 0 || !dist || !ops || !nops) { set_error("mdd_align: bad argument"); return MDD_ERR_ARG; }
    if (na == 0 || nb == 0) { set_error("mdd_align: empty sequence"); return MDD_ERR_EMPTY; }
    const int W = nb + 1;
    std::vector<int> d((size_t)(na + 1) * W);
    for (int j = 0; j <= nb; j++) d[j] = j;
    for (int i = 1; i <= na; i++) d[(size_t)i * W] = i;
    for (int i = 1; i <= na; i++)
        for (int j = 1; j <= nb; j++) {
            const int sub = d[(size_t)(i - 1) * W + j - 1] + (a[i - 1] == b[j - 1] ? 0 : 1);
            const int up = d[(size_t)(i - 1) * W + j] + 1, left = d[(size_t)i * W + j - 1] + 1;
            int mn = left < up ? left : up;
            d[(size_t)i * W + j] = sub < mn ? sub : mn;
        }
    *dist = d[(size_t)na * W + nb];
    // backtrace priority: match > S (diagonal) > I (consumes hypothesis) > D (consumes canonical)
    int i = na, j = nb, n = 0;
    while (i > 0 || j > 0) {
        if (i == 0) { ops[n++] = 3; j--; }
        else if (j == 0) { ops[n++] = 2; i--; }
        else if (a[i - 1] == b[j - 1]) { ops[n++] = 0; i--; j--; }
        else if (d[(size_t)i * W + j] == d[(size_t)(i - 1) * W + j - 1] + 1) { ops[n++] = 1; i--; j--; }
        else if (d[(size_t)i * W + j] == d[(size_t)(i - 1) * W + j] + 1) { ops[n++] = 2; i--; }
        else { ops[n++] = 3; j--; }
    }
    for (int k = 0; k < n / 2; k++) { uint8_t t = ops[k]; ops[k] = ops[n - 1 - k]; ops[n - 1 - k] = t; }
    *nops = n;
    return MDD_OK;
}

extern "C" int mdd_align_batch(const int32_t *a, const int32_t *a_len, int32_t a_stride, const int32_t *b, const int32_t *b_len,
                               int32_t b_stride, int32_t n, int32_t *dist, uint8_t *ops, int32_t ops_stride, int32_t *nops) {
    if (n < 0 || a_stride < 0 || b_stride < 0 || ops_stride < 0 || (n > 0 && (!a || !a_len || !b || !b_len || !dist || !ops || !nops))) {
        set_error("mdd_align_batch: bad argument"); return MDD_ERR_ARG;
    }
    for (int x = 0; x < n; x++) {
        const int na = a_len[x], nb = b_len[x];
        if (na < 0 || nb < 0 || na > a_stride || nb > b_stride || na + nb > ops_stride) { set_error("mdd_align_batch: row %d does not fit its pitch", x); return MDD_ERR_ARG; }
        if (na == 0 || nb == 0) { dist[x] = -1; nops[x] = 0; continue; }
        if (int rc = mdd_align(a + (size_t)x * a_stride, na, b + (size_t)x * b_stride, nb, dist + x, ops + (size_t)x * ops_stride, nops + x)) return rc;
    }
    return MDD_OK;
}

// SURVEY 8(f) #2 -- the evaluation counts of steps/test_ctc_nosil.py for a whole batch, host side.
// Per utterance (ids already without 'sil'): lc = wer(labels, canonical), dc = wer(decoded, canonical),
// err = wer(decoded, labels)[0]; d1 / d2 = print_align_space_canonical_origin of the two paths (:33-60): one value
// per canonical position ('-', 'D', or 'S' + the hypothesis phoneme) plus the list 'I' of insertion gaps; then the
// TA / FR / FA / TR tallies of :249-291, including the reference's habit of removing from the list it iterates.
namespace {
struct PosVal { int type, phone; };      // 0 '-', 1 'D', 2 'S'+phone
struct CanMap { std::vector<PosVal> pos; std::vector<int> ins; };   // ins: gap index j of the key str(j-1)+str(j)

static int eval_map(const int32_t *hyp, int nh, const int32_t *can, int nc, CanMap &m, int32_t *dist_out) {
    std::vector<uint8_t> ops((size_t)nh + nc);
    int32_t dist = 0, nops = 0;
    if (int rc = mdd_align(hyp, nh, can, nc, &dist, ops.data(), &nops)) return rc;
    if (dist_out) *dist_out = dist;
    m.pos.assign(nc, PosVal{0, -1});
    m.ins.clear();
    int hi = 0, j = 0;                      // hypothesis token / canonical position consumed so far
    for (int i = 0; i < nops; i++) {
        switch (ops[i]) {
            case 0: m.pos[j++] = PosVal{0, -1}; hi++; break;
            case 1: m.pos[j++] = PosVal{2, hyp[hi]}; hi++; break;
            case 3: m.pos[j++] = PosVal{1, -1}; break;
            default: m.ins.push_back(j); hi++; break;
        }
    }
    return MDD_OK;
}
}  // namespace

extern "C" int mdd_eval_batch(const int32_t *dec, const int32_t *dec_len, const int32_t *lab, const int32_t *lab_len,
                              const int32_t *can, const int32_t *can_len, int32_t n, int32_t stride, int64_t *counts) {
    if (!dec || !dec_len || !lab || !lab_len || !can || !can_len || !counts || n < 0 || stride <= 0) {
        set_error("mdd_eval_batch: bad argument"); return MDD_ERR_ARG;
    }
    int64_t total = 0, ta = 0, fr = 0, fa = 0, trc = 0, trw = 0, err = 0, nword = 0;
    CanMap d1, d2;
    for (int x = 0; x < n; x++) {
        const int32_t *h = dec + (size_t)x * stride, *l = lab + (size_t)x * stride, *c = can + (size_t)x * stride;
        const int nh = dec_len[x], nl = lab_len[x], nc = can_len[x];
        if (nh > stride || nl > stride || nc > stride) { set_error("mdd_eval_batch: length exceeds stride"); return MDD_ERR_ARG; }
        if (nh <= 0 || nl <= 0 || nc <= 0) { set_error("mdd_eval_batch: utterance %d has an empty sequence", x); return MDD_ERR_EMPTY; }
        int32_t e = 0;
        if (int rc = eval_map(l, nl, c, nc, d1, nullptr)) return rc;      // :219
        {   // :220  decoded vs labels: only the distance is used
            std::vector<uint8_t> ops((size_t)nh + nl);
            int32_t nops = 0;
            if (int rc = mdd_align(h, nh, l, nl, &e, ops.data(), &nops)) return rc;
        }
        if (int rc = eval_map(h, nh, c, nc, d2, nullptr)) return rc;      // :221
        total += nc;                                                       // :245  len(d1.keys()) - 1
        for (int k = 0; k < nc; k++) {                                     // :249-272
            const PosVal a = d1.pos[k], b = d2.pos[k];
            if (a.type == 0 && b.type == 0) ta++;
            else if (a.type == 0) fr++;
            else if (b.type == 0) fa++;
            else if (a.type == b.type && a.phone == b.phone) trc++;
            else trw++;
        }
        // :273-291, the 'I' entry
        std::vector<int> &L1 = d1.ins, &L2 = d2.ins;
        if (L1.empty() && L2.empty()) {
        } else if (L2.empty()) fa += (int64_t)L1.size();
        else if (L1.empty()) fr += (int64_t)L2.size();
        else {
            for (size_t idx = 0; idx < L1.size(); idx++) {                 // `for e in d1['I']` while removing from d1['I']
                const int ev = L1[idx];
                auto it2 = std::find(L2.begin(), L2.end(), ev);
                if (it2 != L2.end()) {
                    L1.erase(std::find(L1.begin(), L1.end(), ev));         // list.remove: the first occurrence
                    L2.erase(it2);
                    trc++;
                }
            }
            fa += (int64_t)L1.size();
            fr += (int64_t)L2.size();
        }
        err += e;                                                          // :293
        nword += nl;                                                       // :294
    }
    counts[0] = total; counts[1] = ta; counts[2] = fr; counts[3] = fa; counts[4] = trc; counts[5] = trw; counts[6] = err; counts[7] = nword;
    return MDD_OK;
}

