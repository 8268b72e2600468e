// Internal declarations of the training step (train_kernels.hip, train.hip).
#pragma once
#include <algorithm>

#include "mdd_internal.h"

namespace mdd {

struct BnSite {            // ReLU + dropout behind a BatchNorm (the conv sites): the dropout byte of element (row, f) is mask[row * F + f]
    const unsigned char *mask; float scale;                // (channels-last copy of the [B,ch,T,W] mask: launch_mask_rows); null = no dropout
};
int launch_mask_rows(const unsigned char *src, unsigned char *dst, int B, int F, int TW, hipStream_t st);

struct LstmBwdArgs {
    const float *dout;     // [T][B][2H]   gradient arriving at the layer's output h
    const float *gates;    // [T][B][2][H][4] saved i,f,g,o (post-activation)
    const float *cst;      // [T][B][2][H]    saved cell state
    const float *whhT;     // [2][H][4H]
    float *dg;             // [T][B][2][4H]   out: pre-activation gate gradients (gate columns u*4+g)
    float *dc;             // [2][B][H]       carried cell gradient
    int T, B, H;
};

int launch_dropout_mask(unsigned char *mask, size_t n, unsigned long long seed, unsigned site, float p, hipStream_t st);
int launch_conv0_train_fwd(const float *x, const float *w, const float *bias, float *z, int B, int T, int F, int ch, hipStream_t st);
int launch_conv0_train_bwd(const float *x, const float *dz, double *acc, float *dw, float *db, int B, int T, int F, int ch, hipStream_t st);
int launch_im2col1(const float *a0, float *col, int B, int T, int W1, int W2, int ch, hipStream_t st);
int launch_col2im1(const float *dcol, float *da0, int B, int T, int W1, int W2, int ch, hipStream_t st);
// conv1 as direct kernels on the channels-last activations (train_conv1.hip); ch = 32 only (other widths: im2col + GEMM)
int launch_conv1_fwd_direct(const float *a0, const float *w1r, const float *bias, float *z1, int B, int T, int W1, int W2, int ch, hipStream_t st);
int launch_conv1_dgrad_direct(const float *dz1, const float *w1r, float *da0, int B, int T, int W1, int W2, int ch, hipStream_t st);
int conv1_wgrad_parts(int B, int T, int W2);
int launch_conv1_wgrad_direct(const float *dz1, const float *a0, float *part, int B, int T, int W1, int W2, int ch, hipStream_t st);
int init_conv1_attributes();
int launch_pack_w1(const float *src, float *dst, int ch, bool to_packed, hipStream_t st);
int launch_cnn_seq(float *a1, float *seq, int B, int Tp, int W2, int ch, bool to_seq, hipStream_t st);
int launch_bn_train_fwd(const float *x, size_t R, int F, const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                        float *running_var, double *s1s2, float *mean, float *invstd, const BnSite *site, float *y, hipStream_t st);
int launch_bn_train_bwd(const float *x, const float *g, size_t R, int F, const float *gamma, const float *beta, const float *mean, const float *invstd,
                        const BnSite *site, double *s1s2, float *dx, float *dgamma, float *dbeta, hipStream_t st);
int launch_col_sum(const float *g, size_t R, int F, double *s1s2, float *out, hipStream_t st);
int launch_dropout_rows(const float *x, const unsigned char *mask, float scale, size_t n, float *y, hipStream_t st);
int launch_copy_cols(const float *src, int ld_src, int s0, float *dst, int ld_dst, int c0, size_t R, int width, bool add, hipStream_t st);
int launch_softmax_rows(const float *x, size_t R, int n, float *y, bool log, hipStream_t st);
int launch_softmax_bwd_rows(const float *y, const float *g, size_t R, int n, float *dx, bool log, hipStream_t st);
int launch_embed_bwd(const float *g, const int64_t *ids, int B, int L, int E, int rows, float *dE, hipStream_t st);
int launch_reduce_parts(const float *part, int parts, size_t n, float *out, hipStream_t st);
int launch_pack_gates(const float *w_fwd, const float *w_rev, float *packed, int H, int K, hipStream_t st);
int launch_unpack_gates(const float *packed, float *out_fwd, float *out_rev, int H, int K, hipStream_t st);
int launch_transpose_whh(const float *w, float *wt, int H, hipStream_t st);
int launch_lstm_bwd(const LstmBwdArgs &a, hipStream_t st);
int launch_lstm_layer_train(const LstmStepArgs &a, hipStream_t st);   // lstm.hip: generic step kernels, saving gates and cell states
int launch_split_rows(const float *src, int ld, size_t rows, int cols, int cols_pad, unsigned short *hi, unsigned short *lo, hipStream_t st);
int launch_transpose_split(const float *src, int ld, int rows, int cols, int rows_pad, unsigned short *hi, unsigned short *lo, hipStream_t st);
int launch_adam_multi(float *const *p, float *const *g, float *const *m, float *const *v, const int64_t *numel, int n, float lr, float b1, float b2, float eps,
                      float wd, int step, hipStream_t st);
int launch_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st);

}  // namespace mdd
