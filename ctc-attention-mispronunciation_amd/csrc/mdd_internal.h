// Internal declarations shared by the translation units of libmdd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mdd_hip.h"

namespace mdd {

void set_error(const char *fmt, ...);

#define MDD_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            mdd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MDD_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define MDD_LAUNCH_CHECK() MDD_HIP_CHECK(hipGetLastError())

// A split-bf16 tensor: hi = bf16(x), lo = bf16(x - hi), two planes of the same [rows][ld] shape.
struct SplitPtr { unsigned short *hi, *lo; };

// ---- kernel launchers (each enqueues on `st`, returns an mdd_status) -------------------------
// C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]); fp32 MFMA (v_mfma_f32_32x32x2_f32).  Batched over
// `batch` with element strides sA/sW/sC.
int launch_gemm_nt(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int lda, int ldw,
                   int ldc, int batch, long sA, long sW, long sC, hipStream_t st);

// General fp32 GEMM (training step): C[m,n] (+)= sum_k opA[m,k] opB[n,k]; ta / tb: the operand is stored [K, rows] (see gemm.hip)
int launch_gemm_f32(bool ta, bool tb, const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda, int ldb,
                    int ldc, int batch, long sA, long sB, long sC, bool accumulate, hipStream_t st, int ksplit = 0);

// C = A . W^T on the bf16 matrix cores with split-bf16 operands (see gemm_bf16x3.hip).  Output fp32 C, or a
// split-bf16 tensor when Csplit != nullptr.  K % 32 == 0, ld* % 8 == 0.
int launch_gemm_bf16x3(const SplitPtr &A, const SplitPtr &W, const float *bias, float *C, const SplitPtr *Csplit, int M, int N,
                       int K, int lda, int ldw, int ldc, int batch, long sA, long sW, long sC, hipStream_t st);
int init_gemm_attributes();
int launch_split(const float *x, size_t n, const SplitPtr &out, hipStream_t st);
int launch_unsplit(const SplitPtr &in, size_t n, float *x, hipStream_t st);
// x [rows][ld] (K columns used) -> three bf16 planes hi | mid | lo (rows x K elements each, consecutive), hi + mid + lo == x exactly,
// each in the K-tile-major order [K / 32][rows][32] the f32x6 kernel streams (gemm_bf16x6.hip)
int launch_split3(const float *x, int rows, int K, int ld, unsigned short *planes, hipStream_t st);
// C = A . W^T with fp32-grade arithmetic on the bf16 matrix cores: operands as three K-tile-major bf16 planes each (gemm_bf16x6.hip)
int launch_gemm_f32x6(const unsigned short *A3, size_t a_plane, const unsigned short *W3, size_t w_plane, const float *bias, float *C, int M, int N, int K,
                      int ldc, hipStream_t st, long long *stamps = nullptr);
int init_gemm_x6_attributes();

int launch_stack_skip(const float *raw, int B, int T_raw, int D, int right, int skip, int n_down, float *out,
                      hipStream_t st);
// conv0: x [B,T,F] -> y0 [B,ch,T,W1]; conv1: y0 -> seq [T/2,B,ch*W2] (BN+ReLU folded; scale/shift per channel)
int launch_conv0(const float *x, const float *w, const float *scale, const float *shift, float *y0, int B, int T, int F,
                 int ch, hipStream_t st);
int launch_conv1(const float *y0, const float *w_t, const float *scale, const float *shift, float *seq, SplitPtr seq_split, int B,
                 int T, int W1, int ch, hipStream_t st);  // seq (fp32) and/or seq_split may be null

// conv0 -> conv1 fused on the bf16 matrix cores (feat 243, 32 channels): x [B,T,243] -> split-bf16 rows [T/2*B, 1952]
// (w1: conv1 weights as [co][kh][kw][ci] hi/lo planes).  out_f32 optional (taps).
int launch_conv_fused(const float *x, const float *w0, const float *sc0, const float *sh0, SplitPtr w1, const float *sc1,
                      const float *sh1, SplitPtr out, float *out_f32, int B, int T, int Traw, hipStream_t st);
int launch_conv_fused3(const float *x, const float *w0, const float *sc0, const float *sh0, const unsigned short *w1_3, const float *sc1,
                       const float *sh1, unsigned short *out3, float *out_f32, int B, int T, int Traw, hipStream_t st);   // f32x6 form: three K-tile-major planes out
int init_conv_attributes();

struct LstmStepArgs {
    const float *gx;     // [T][B][2][4H], gate columns permuted to u*4+g
    const float *whh;    // [2][4H][H], rows permuted the same way
    float *hbuf;         // [2 parity][2 dir][B][H]
    float *cbuf;         // [2 dir][B][H]
    float *out;          // [T][B][2H] layer output with oscale/oshift applied (may equal out_raw; nullable)
    SplitPtr out_split;  // same values as split-bf16 planes (nullable): the next GEMM's A operand
    float *out_raw;      // [T][B][2H] raw h (nullable)
    const float *oscale; // [2H] (nullable -> identity)
    const float *oshift;
    int T, B, H;
    SplitPtr whh_split;  // row-major Whh' [2][4H][H] as hi/lo planes (split-bf16 step only)
    unsigned short *hsplit;  // h exchange of the split-bf16 step: [2 parity][hi|lo][2 dir][B][H]; null selects the fp32 steps
    int packed;          // 1: whh / hbuf / cbuf use the packed consumer layouts of lstm_step_packed_kernel
    float *gates_save = nullptr;   // train mode (generic step kernel only): [T][B][2][H][4] post-activation i,f,g,o
    float *c_save = nullptr;       //                                          [T][B][2][H]    cell state
    const int *seqlen = nullptr;   // fused batches of different lengths: steps valid per batch row; the REVERSE direction holds h = c = 0 while t >= seqlen[b]
                                   // (it starts at seqlen[b]-1 with a zero state, as it would in the row's own batch); null = all T steps
};
// Enqueue all T steps of one bidirectional layer.
int launch_lstm_layer(const LstmStepArgs &a, hipStream_t st);

int launch_embed(const float *table, int rows, int E, const int64_t *ids, int B, int L, float *out, SplitPtr out_split,
                 int *err_flag, hipStream_t st);
// softmax over L of S[b][t][:], ctx = A.V, y = BN(cat(X, ctx)), logits = y.Wfc^T, log-softmax
int launch_attn_tail(const float *S, int Lp, const float *X, const float *V, const float *fscale, const float *fshift,
                     const float *wfc, const float *wfcp, float *logp, int Tp, int B, int L, int H2, int C, hipStream_t st,
                     const int *llen = nullptr);   // llen[b]: canonical length of b's own batch (softmax / context over l < llen[b]); null = L

int init_kernel_attributes();
int init_ctc_attributes();
int init_lstm_attributes();
// One launch for the whole layer (256 co-resident workgroups in 8-workgroup teams, data-tagged hand-off; see lstm.hip).
int granule_bg(int B);
int init_granule_attributes();
int persistent_grid_fits(int n_cu);   // 1 when all 256 workgroups of a persistent layer launch can be resident at once
// data-tagged variant (8-workgroup teams, no counter): hx = 2*32*granule_bg(B)*H u64 granules (+ stamps), sync: 32 uints
int launch_lstm_layer_granule(const LstmStepArgs &s, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st);
// Exact-fp32 persistent layer (lstm_f32.hip): same teams / exchange buffer; W_hh in the packed layout (LstmStepArgs::packed), fp32 outputs
// zero fill by a kernel (n a multiple of 16; see its definition in lstm.hip for why not hipMemsetAsync)
int launch_zero_fill(void *p, size_t n, hipStream_t st);
int launch_lstm_layer_f32(const LstmStepArgs &s, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st);
int init_lstm_f32_attributes();
int persistent_f32_grid_fits(int n_cu);
// f32x6 persistent layer (lstm_x6.hip): teams of 16, W_hh' as three row-major bf16 planes [3][2][4H][H], h exchanged as three bf16 planes; fp32 outputs
int launch_lstm_layer_x6(const LstmStepArgs &s, const unsigned short *whh3, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st);
int init_lstm_x6_attributes();
int persistent_x6_grid_fits(int n_cu);
size_t lstm_x6_hx_bytes(int H, int B);
int lstm_x6_max_b(int H);
// The backward recurrence of a layer in one launch (split-bf16 training variant, B <= 256, H in {256, 384}); hx: lstm_bwd_granule_hx_bytes(H)
size_t lstm_bwd_granule_hx_bytes(int H);
int launch_lstm_bwd_granule(const float *dout, const float *gates, const float *cst, SplitPtr whhT, float *dg, int T, int B, int H, unsigned short *hx,
                            unsigned int *sync, int *err_flag, hipStream_t st);
// Per-device ticket around work that contains persistent launches (api.hip): launches of different handles / streams of one device run
// one after another on the GPU (event dependency; the host does not block).  enter locks, leave records the event and unlocks.
int device_gate_enter(int device, hipStream_t st, bool *held);
int device_gate_leave(int device, hipStream_t st, bool held, int rc);

}  // namespace mdd
