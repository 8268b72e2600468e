// BiLSTM recurrence: one launch per time step, both directions in the same grid.
//
// Reference: torch.nn.LSTM as used by BatchRNN.forward (AA/models/model_ctc.py:36-49) and by the text
// encoder (:150,198): gate order i,f,g,o, zero initial state, every padded step is processed, the
// reverse direction starts at t = T-1.
//
// Per step and direction:  G^T[n,b] = sum_k Whh[n,k] h_prev[b,k]   (n = 4H gate rows, b = batch)
// is computed with v_mfma_f32_16x16x4_f32 with the GATE rows on the MFMA row axis and the batch on
// the column axis.  Gate rows are stored permuted (n' = unit*4 + gate) so that the four accumulator
// registers of a lane are exactly i,f,g,o of one (unit, batch) pair: the cell update needs no
// cross-lane traffic and no LDS.  A wave owns 4 hidden units x 16 batch rows; a workgroup = 4 waves
// = 4 batch tiles of the same 4 units; grid = (H/4 unit tiles, 2 directions, ceil(B/64)).
// The k axis is split over the four 16-lane groups of the wave in contiguous quarters (the MFMA
// only requires A and B to agree on k), so every lane streams H/4 contiguous floats of one Whh row
// and of one h_prev row.
//
// The launch boundary is the step-to-step dependency (all-to-all over hidden units): per the
// MI355X price list a dependent kernel boundary (~1.5 us) is cheaper than any in-launch grid sync.
#include "lstm_persist.h"

namespace mdd {

// split-bf16 kernels: hardware exp2 / rcp (<= ~2 ulp each; |error| on a gate <= 3e-7, far below the 1e-5 the bf16x3
// products already spend of the 1e-4 budget).  The exact-fp32 mode keeps the libm-grade functions above.
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }

// ---- generic variant (any H % 4 == 0; row-major Whh', h and c): used for small test geometries
// VEC: 0 scalar k loop (any H % 4 == 0), 1 float4 k loop (H % 16 == 0), N > 1: a lane's quarter of the contraction is exactly N
// float4 steps (24 at H = 384, 16 at H = 256) and all of its loads are issued before the first MFMA (one L2 round trip per step).
template <int VEC>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a, int s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = blockIdx.x, d = blockIdx.y;
    const int b0 = (blockIdx.z * 4 + wave) * 16;
    if (b0 >= a.B) return;  // wave-uniform
    const int H = a.H, B = a.B, KQ = H >> 2;
    const int t = d ? (a.T - 1 - s) : s;
    const int li = lane & 15, kq = lane >> 4;
    const int b = b0 + li, bc = b < B ? b : B - 1;
    const float *hprev = a.hbuf + ((size_t)((s & 1) ^ 1) * 2 + d) * B * H;
    float *hnext = a.hbuf + ((size_t)(s & 1) * 2 + d) * B * H;

    // the epilogue's operands do not depend on the product: request them first, so the step pays one memory round trip, not two
    const int u = ut * 4 + kq;
    const size_t ci = ((size_t)d * B + bc) * H + u;
    const float4 g4 = *reinterpret_cast<const float4 *>(a.gx + (((size_t)t * B + bc) * 2 + d) * 4 * H + u * 4);
    const float cold = s > 0 ? a.cbuf[ci] : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const float *wp = a.whh + ((size_t)d * 4 * H + ut * 16 + li) * H + kq * KQ;  // A[row li][k quarter kq]
        const float *hp = hprev + (size_t)bc * H + kq * KQ;                            // B[k quarter kq][col li]
        if (VEC > 1) {
            float4 w4[VEC > 1 ? VEC : 1], h4[VEC > 1 ? VEC : 1];
#pragma unroll
            for (int n = 0; n < VEC; n++) { w4[n] = *reinterpret_cast<const float4 *>(wp + 4 * n); h4[n] = *reinterpret_cast<const float4 *>(hp + 4 * n); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < VEC; n++) {    // same accumulation order as the float4 loop below: one chain, x y z w per step
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].x, h4[n].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].y, h4[n].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].z, h4[n].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[n].w, h4[n].w, acc, 0, 0, 0);
            }
        } else if (VEC) {
#pragma unroll 8
            for (int k = 0; k < KQ; k += 4) {
                float4 w4 = *reinterpret_cast<const float4 *>(wp + k);
                float4 h4 = *reinterpret_cast<const float4 *>(hp + k);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, h4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, h4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, h4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, h4.w, acc, 0, 0, 0);
            }
        } else {
            for (int k = 0; k < KQ; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[k], hp[k], acc, 0, 0, 0);
        }
    }
    // D layout: col = lane&15 (batch), row = 4*(lane>>4) + r  ->  unit = ut*4 + (lane>>4), gate = r
    if (b < B) {
        const float ig = sigmoid_f(acc[0] + g4.x), fg = sigmoid_f(acc[1] + g4.y);
        const float gg = tanhf(acc[2] + g4.z), og = sigmoid_f(acc[3] + g4.w);
        const bool hold = d && a.seqlen && t >= a.seqlen[b];     // reverse direction has not reached this row's last step yet
        const float cn = hold ? 0.f : fg * cold + ig * gg;
        const float hn = hold ? 0.f : og * tanhf(cn);
        a.cbuf[ci] = cn;
        hnext[(size_t)b * H + u] = hn;
        if (a.gates_save) {   // train mode: what the backward sweep needs (train_kernels.hip)
            const size_t si = (((size_t)t * B + b) * 2 + d) * H + u;
            *reinterpret_cast<float4 *>(a.gates_save + si * 4) = make_float4(ig, fg, gg, og);
            a.c_save[si] = cn;
        }
        const size_t oi = ((size_t)t * B + b) * 2 * H + d * H + u;
        if (a.out_raw) a.out_raw[oi] = hn;
        const float ov = a.oscale ? hn * a.oscale[d * H + u] + a.oshift[d * H + u] : hn;
        if (a.out && a.out != a.out_raw) a.out[oi] = ov;
        if (a.out_split.hi) {
            __bf16 hb = (__bf16)ov, lb = (__bf16)(ov - (float)hb);
            a.out_split.hi[oi] = *reinterpret_cast<unsigned short *>(&hb);
            a.out_split.lo[oi] = *reinterpret_cast<unsigned short *>(&lb);
        }
    }
}

// ---- packed variant (H % 16 == 0): every global access of the k-loop is a fully coalesced 1 KB read.
// Whh is repacked at weight-load time into the exact order the lanes consume it,
//   Wp[d][ut][j][lane][m] = Whh'[d][ut*16 + (lane&15)][16 j + 4 (lane>>4) + m]
// and h is exchanged between steps in the same consumer order,
//   hp[parity][d][bt][j][lane][m] = h[bt*16 + (lane&15)][16 j + 4 (lane>>4) + m]
// (a producer wave owns k = 4 ut .. 4 ut+3 for 16 batch rows: 64 contiguous floats), and c lives in
// producer order cp[d][bt][ut][lane].  Two independent accumulators (even / odd float4 components) keep
// the MFMA pipe at its 32-cycle issue rate instead of the 40-cycle dependent latency; lstm_layer_f32_kernel (lstm_f32.hip)
// repeats this accumulation order and the cell update bit for bit.
template <int J>
__global__ __launch_bounds__(256, 1) void lstm_step_packed_kernel(LstmStepArgs a, int s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = blockIdx.x, d = blockIdx.y;
    const int NBT = (a.B + 15) >> 4, NUT = a.H >> 2;
    const int bt = blockIdx.z * 4 + wave;
    if (bt >= NBT) return;  // wave-uniform
    const int H = a.H, B = a.B;
    const int t = d ? (a.T - 1 - s) : s;
    const int li = lane & 15, uu = lane >> 4;
    const int b = bt * 16 + li, u = ut * 4 + uu;
    const size_t hplane = (size_t)2 * NBT * J * 256;  // floats per parity
    const float4 *hp = reinterpret_cast<const float4 *>(a.hbuf + ((s & 1) ^ 1) * hplane) + ((size_t)(d * NBT + bt) * J) * 64 + lane;
    const float4 *wp = reinterpret_cast<const float4 *>(a.whh) + ((size_t)(d * NUT + ut) * J) * 64 + lane;
    const size_t ci = ((size_t)(d * NBT + bt) * NUT + ut) * 64 + lane;

    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float cold = 0.f;
    if (b < B) {
        g4 = *reinterpret_cast<const float4 *>(a.gx + (((size_t)t * B + b) * 2 + d) * 4 * H + u * 4);
        if (s > 0) cold = a.cbuf[ci];
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (s > 0) {
        float4 w4[J], h4[J];
#pragma unroll
        for (int j = 0; j < J; j++) { w4[j] = wp[j * 64]; h4[j] = hp[j * 64]; }
        // keep every load ahead of the first MFMA: the whole k-panel (2 x J KB per wave) is in flight at once,
        // so the step pays ONE L2 round trip instead of one per k-slice (hipcc otherwise sinks the loads)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < J; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].x, h4[j].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].y, h4[j].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].z, h4[j].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].w, h4[j].w, acc1, 0, 0, 0);
        }
    }
    if (b < B) {
        const float gi = (acc0[0] + acc1[0]) + g4.x;
        const float gf = (acc0[1] + acc1[1]) + g4.y;
        const float gg = (acc0[2] + acc1[2]) + g4.z;
        const float go = (acc0[3] + acc1[3]) + g4.w;
        const float ig = gate_sigmoid(gi), fg = gate_sigmoid(gf), cg = gate_tanh(gg), og = gate_sigmoid(go);
        const bool hold = d && a.seqlen && t >= a.seqlen[b];
        const float cn = hold ? 0.f : fg * cold + ig * cg;
        const float hn = hold ? 0.f : og * gate_tanh(cn);
        a.cbuf[ci] = cn;
        float *hnext = a.hbuf + (s & 1) * hplane;
        hnext[(((size_t)(d * NBT + bt) * J + (ut >> 2)) * 64 + (ut & 3) * 16 + li) * 4 + uu] = hn;
        const size_t oi = ((size_t)t * B + b) * 2 * H + d * H + u;
        if (a.out_raw) a.out_raw[oi] = hn;
        const float ov = a.oscale ? hn * a.oscale[d * H + u] + a.oshift[d * H + u] : hn;
        if (a.out && a.out != a.out_raw) a.out[oi] = ov;
        if (a.out_split.hi) {
            __bf16 hb = (__bf16)ov, lb = (__bf16)(ov - (float)hb);
            a.out_split.hi[oi] = *reinterpret_cast<unsigned short *>(&hb);
            a.out_split.lo[oi] = *reinterpret_cast<unsigned short *>(&lb);
        }
    }
}

// ---- split-bf16 tiled variant: the recurrent product on the bf16 matrix cores, operands shared through LDS.
// Per step and direction  G^T[4H x B] = Whh'[4H x H] . h^T[H x B]  is a small GEMM that every step re-reads in
// full, so the step is bound by L2->CU traffic, not by arithmetic.  A workgroup owns RT*16 gate rows (RT*4 hidden
// units) x 64 batch rows and stages its Whh' panel and its h panel ONCE in LDS (bf16 hi/lo planes: the same 4
// bytes per element as fp32), instead of every wave streaming both operands from L2 (2.6x less L2 traffic than the
// packed fp32 variant at B=256).  Products are the bf16x3 form Ah.Bh + Ah.Bl + Al.Bh on v_mfma_f32_16x16x32_bf16
// (3/16 of the fp32 MFMA time).  The D layout (col = batch, 4 accumulator registers = i,f,g,o of one unit) and the
// cell epilogue are those of the fp32 variants.  h is exchanged between steps as row-major [dir][B][H] hi/lo planes
// (the producer lane writes its bf16 pair), c stays fp32 row-major.
// LDS: Wh | Wl : RT*16 rows, hh | hl : 64 rows, each row H*2 bytes + 16 pad (conflict-free ds_read_b128 fragments).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The recurrent state as split bf16: hi = bf16(h), lo = bf16(h - hi) with its last mantissa bit cleared.  Every
// recurrent kernel (per-step and persistent) uses this same split, so they stay bit-identical; the cleared bit is where
// the granule hand-off carries its epoch tag (h keeps ~16 significant bits: 2^-17 relative).
__device__ __forceinline__ unsigned int split_h(float hn) {          // returns hi | lo << 16
    __bf16 hb = (__bf16)hn, lbf = (__bf16)(hn - (float)hb);
    return (unsigned int)*reinterpret_cast<unsigned short *>(&hb) | (((unsigned int)*reinterpret_cast<unsigned short *>(&lbf) & 0xfffeu) << 16);
}

template <int RT, int H>
__global__ __launch_bounds__(256, 1) void lstm_step_x3_kernel(LstmStepArgs a, int s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWB = H * 2 + 16, WROWS = RT * 16, CPR = H / 8;   // CPR: 16-byte chunks per row
    unsigned char *Wh = smem, *Wl = Wh + WROWS * ROWB, *Hh = Wl + WROWS * ROWB, *Hl = Hh + 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, B = a.B;
    const int r0 = blockIdx.x * WROWS;            // first gate row (permuted order) of this workgroup
    const int b0 = blockIdx.z * 64;               // first batch row
    const int t = d ? (a.T - 1 - s) : s;
    const int li = lane & 15, uu = lane >> 4;
    const int b = b0 + wave * 16 + li;
    const size_t hplane = (size_t)2 * B * H;      // elements per (parity, hi|lo) plane: [dir][B][H]
    const unsigned short *hprev_h = a.hsplit + (size_t)(((s & 1) ^ 1) * 2 + 0) * hplane + (size_t)d * B * H;
    const unsigned short *hprev_l = a.hsplit + (size_t)(((s & 1) ^ 1) * 2 + 1) * hplane + (size_t)d * B * H;

    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (s > 0) {
        // ---- stage both panels: all loads in flight at once, then one LDS write pass
        constexpr int WCH = WROWS * CPR, HCH = 64 * CPR;              // chunks per plane
        static_assert(WCH % 256 == 0 && HCH % 256 == 0, "panel sizes must be whole passes of the workgroup");
        constexpr int NW = WCH / 256, NH = HCH / 256;
        u32x4 wvh[NW], wvl[NW], hvh[NH], hvl[NH];
        const unsigned short *wsrc_h = a.whh_split.hi + ((size_t)d * 4 * H + r0) * H;
        const unsigned short *wsrc_l = a.whh_split.lo + ((size_t)d * 4 * H + r0) * H;
        // branch-free: rows past B are clamped to the last valid row (their output columns are never stored)
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int q = tid + 256 * i;                              // rows are contiguous: chunk q = row q/CPR
            wvh[i] = *reinterpret_cast<const u32x4 *>(wsrc_h + (size_t)q * 8);
            wvl[i] = *reinterpret_cast<const u32x4 *>(wsrc_l + (size_t)q * 8);
        }
#pragma unroll
        for (int i = 0; i < NH; i++) {
            const int q = tid + 256 * i, row = q / CPR, c = q - row * CPR;
            const int br = min(b0 + row, B - 1);
            hvh[i] = *reinterpret_cast<const u32x4 *>(hprev_h + (size_t)br * H + c * 8);
            hvl[i] = *reinterpret_cast<const u32x4 *>(hprev_l + (size_t)br * H + c * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int q = tid + 256 * i, row = q / CPR, c = q - row * CPR;
            *reinterpret_cast<u32x4 *>(Wh + row * ROWB + c * 16) = wvh[i];
            *reinterpret_cast<u32x4 *>(Wl + row * ROWB + c * 16) = wvl[i];
        }
#pragma unroll
        for (int i = 0; i < NH; i++) {
            const int q = tid + 256 * i, row = q / CPR, c = q - row * CPR;
            *reinterpret_cast<u32x4 *>(Hh + row * ROWB + c * 16) = hvh[i];
            *reinterpret_cast<u32x4 *>(Hl + row * ROWB + c * 16) = hvl[i];
        }
        __syncthreads();
        // ---- this wave: batch tile `wave` (16 columns) x RT row tiles
        const unsigned char *hb_h = Hh + (wave * 16 + li) * ROWB + uu * 16;
        const unsigned char *hb_l = Hl + (wave * 16 + li) * ROWB + uu * 16;
#pragma unroll 4
        for (int ks = 0; ks < H / 32; ks++) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(hb_h + ks * 64);
            const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(hb_l + ks * 64);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(Wh + (rt * 16 + li) * ROWB + uu * 16 + ks * 64);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Wl + (rt * 16 + li) * ROWB + uu * 16 + ks * 64);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[rt], 0, 0, 0);
            }
        }
    }
    if (b >= B) return;
    unsigned short *hnext_h = a.hsplit + (size_t)((s & 1) * 2 + 0) * hplane + (size_t)d * B * H;
    unsigned short *hnext_l = a.hsplit + (size_t)((s & 1) * 2 + 1) * hplane + (size_t)d * B * H;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int u = (r0 >> 2) + rt * 4 + uu;    // D: col = lane&15 (batch), row = 4*(lane>>4) + r -> unit uu of the tile, gate r
        const float4 g4 = *reinterpret_cast<const float4 *>(a.gx + (((size_t)t * B + b) * 2 + d) * 4 * H + u * 4);
        const size_t ci = ((size_t)d * B + b) * H + u;
        const float cold = s > 0 ? a.cbuf[ci] : 0.f;
        const float ig = fast_sigmoid(acc[rt][0] + g4.x), fg = fast_sigmoid(acc[rt][1] + g4.y);
        const float cg = fast_tanh(acc[rt][2] + g4.z), og = fast_sigmoid(acc[rt][3] + g4.w);
        const bool hold = d && a.seqlen && t >= a.seqlen[b];
        const float cn = hold ? 0.f : fg * cold + ig * cg;
        const float hn = hold ? 0.f : og * fast_tanh(cn);
        a.cbuf[ci] = cn;
        {
            const unsigned int pk = split_h(hn);
            hnext_h[(size_t)b * H + u] = (unsigned short)(pk & 0xffffu);
            hnext_l[(size_t)b * H + u] = (unsigned short)(pk >> 16);
        }
        const size_t oi = ((size_t)t * B + b) * 2 * H + d * H + u;
        if (a.out_raw) a.out_raw[oi] = hn;
        const float ov = a.oscale ? hn * a.oscale[d * H + u] + a.oshift[d * H + u] : hn;
        if (a.out && a.out != a.out_raw) a.out[oi] = ov;
        if (a.out_split.hi) {
            __bf16 hb = (__bf16)ov, lb = (__bf16)(ov - (float)hb);
            a.out_split.hi[oi] = *reinterpret_cast<unsigned short *>(&hb);
            a.out_split.lo[oi] = *reinterpret_cast<unsigned short *>(&lb);
        }
    }
}

template <int RT, int H>
static int launch_x3_steps(const LstmStepArgs &a, hipStream_t st) {
    constexpr int ROWB = H * 2 + 16;
    const size_t smem = (size_t)2 * (RT * 16 + 64) * ROWB;
    dim3 grid(4 * H / (RT * 16), 2, (a.B + 63) / 64), block(256);
    for (int s = 0; s < a.T; s++) hipLaunchKernelGGL((lstm_step_x3_kernel<RT, H>), grid, block, smem, st, a, s);
    return MDD_OK;
}

int init_lstm_attributes() {
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_step_x3_kernel<2, 384>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_step_x3_kernel<2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return MDD_OK;
}

// ---- persistent layer kernel: the whole recurrence of one BiLSTM layer in ONE launch (lstm_layer_granule_kernel below).
// The per-step launches above re-read W_hh (4.7 MB) from L2 every step; measured, a CU ingests only ~50-100 GB/s from L2, so at
// fused batch sizes the step is bound by that re-read.  The persistent form keeps every workgroup's W_hh rows in registers for
// the whole layer and moves only h between the workgroups of a team.  (An earlier form of it -- 16-workgroup teams around a
// per-team arrival counter -- was superseded by the data-tagged hand-off and is no longer built.)



// acc += A.B with A taken straight from an accumulation-file register (half of the resident weight fragments live
// there: the compiler would otherwise copy each one to a VGPR with four v_accvgpr_read per MFMA, on every step)
__device__ __forceinline__ void mfma_a(f32x4 &acc, const bf16x8 &a_agpr, const bf16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a_agpr), "v"(b));
}
// The other two products of a k-step in the same form.  ALL MFMAs of the persistent kernels' product loops are asm volatile, so that
// their issue order is the source order (product-major: a k-step's three products over the RTW row tiles, i.e. an accumulator is
// written again RTW >= 2 MFMAs later), whatever the scheduler does around them.  Round 2 mixed builtin MFMAs with the asm one: the
// compiler cannot see that an asm MFMA reads its accumulator, inserts no wait state in front of it, and -- once the
// __builtin_amdgcn_sched_barrier calls that happened to pin the product-major order were removed -- scheduled
//     v_mfma a[152:155], v[98:101], v[182:185], 0 ; v_mfma a[152:155], a[96:99], v[178:181], a[152:155]
// back to back: the second read a[152:155] before the first had written it, and results moved by 1.7e-4 (profiles/round3_lstm_ordering.txt).
// The sched barriers are a speed hint now (MDD_NO_SCHED_HINT builds give identical bits).  The accumulators' readers after the loop
// are vector instructions the compiler cannot order behind an asm MFMA either: mfma_drain() supplies their wait states.
__device__ __forceinline__ void mfma_v(f32x4 &acc, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v0(f32x4 &acc, const bf16x8 &a, const bf16x8 &b) {   // first product of a tile: from a literal zero
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
// The B operands of a k-step are made by vector instructions (tag masks, register moves) just before its MFMAs.  "VALU writes a VGPR ->
// MFMA reads it" needs two wait states, which the compiler supplies for builtin MFMAs and cannot for asm ones (the first all-asm build was
// wrong in the 1-, 3- and 4-tile forms, whose schedules put a v_and directly in front of the first MFMA): this statement takes both
// operands as read-write, so every instruction that produces them sits in front of it, and its s_nop 1 is the two wait states.
// The A fragments of the k-step go through it too: under register pressure (three and four tiles) the compiler parks some of the "resident"
// VGPR fragments in the accumulation file and copies them back (v_accvgpr_read = a VALU write) right in front of their MFMA.
// tools/check_mfma_hazards.py scans the built ISA for exactly these patterns (tests/test_host.py::test_asm_mfma_hazards).
template <int N>
__device__ __forceinline__ void mfma_operands_ready(bf16x8 &bh, bf16x8 &bl, bf16x8 (&a)[N]) {
    static_assert(N == 2 || N == 3, "row tiles per wave");
    if (N == 3) asm volatile("s_nop 1" : "+v"(bh), "+v"(bl), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
    else asm volatile("s_nop 1" : "+v"(bh), "+v"(bl), "+v"(a[0]), "+v"(a[1]));
}
#ifdef MDD_NO_SCHED_HINT
#define MDD_SCHED_HINT() do { } while (0)
#else
#define MDD_SCHED_HINT() __builtin_amdgcn_sched_barrier(0)
#endif

// ---- persistent layer kernel, data-tagged hand-off ("the data IS the flag", Guideline 16 form R2).
// Same team structure, but (1) a team is 8 workgroups (32 teams = 2 directions x 16 batch groups), each workgroup
// owning 4H/8 gate rows: RTW row tiles per wave, fragments resident in registers; the panel a workgroup pulls per
// step is half as tall; (2) h travels as 16-byte chunks of FOUR units of one batch row, {hi0 hi1 hi2 hi3 | lo0 lo1 lo2
// lo3}, whose epoch tag (1..3 = step % 3 + 1) rides in the spare last bits of the lo halves (split_h clears them: bit 0
// of the tag in the even units, bit 1 in the odd ones): the producer just stores them write-through -- no drain, no
// barrier, no counter -- and the consumer checks the tags of what it fetched ("the data IS the flag").  A chunk is one
// lane's single 16-byte store, so a chunk is either old or new as a whole.  One L2 round trip replaces three (drain,
// counter add, poll).  A panel of one parity only ever holds step s or step s-2 (nobody publishes step s before
// everyone has read s-2), whose tags differ; the exchange buffer is zeroed before every launch (tag 0 is never valid),
// so replays cannot see a previous launch's data.  The hand-off is bound by the bytes all 256 workgroups pull past
// their L2s and by the ~2 us a write-through store takes to become visible, hence the dense format (4 bytes per unit
// and row) and the chunk order = (unit / 4) * 16 + row: a workgroup's share is one contiguous run, the sweep a linear
// LDS-DMA copy, and the chunks are the MFMA operands as they lie (see the kernel body).
// A team's batch rows come in NBT tiles of 16 (one MFMA column tile each).  The tiles are independent recurrences and
// are advanced in turn, each with its own panel and epoch: while one tile's h_s travels to the team (the hand-off is
// ~2 us of pure latency), the workgroup runs the other tiles' MFMAs and cell updates.  With three or more tiles a
// tile's panel has been complete for a whole phase when its turn comes, so its sweep is requested one phase AHEAD
// (LDS-DMA into the other panel buffer, one piece per k-step of the MFMA loop); with two tiles it is requested
// after the MFMAs of the phase before (the panel is not complete earlier); with one there is nothing to overlap.
// The gate pre-activations never occupy registers: each tile's next [16 rows x 4*UW] slab is fetched by LDS-DMA as
// soon as the cell update has consumed the current one.
template <int H, int NBT, int RTW, bool TRAIN = false>
__global__ __launch_bounds__(256, 1) void lstm_layer_granule_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTH = 256, KS = H / 32, RM = 4 * H / 8;
    constexpr int PF = NBT == 1 ? 0 : (NBT == 2 ? 1 : 2);          // where the next phase's sweep is requested (see above)
    static_assert(RM == 4 * RTW * 16, "4 waves x RTW row tiles must cover the workgroup's gate rows");
    constexpr int UW = RM / 4;                                      // units this workgroup owns (a 2*UW-byte run per output row and plane)
    unsigned short *Oh = reinterpret_cast<unsigned short *>(smem), *Ol = Oh + 16 * UW;   // step outputs, [row][unit]
    float *Of = reinterpret_cast<float *>(Ol + 16 * UW);
    unsigned int *Og = reinterpret_cast<unsigned int *>(Of + 16 * UW);   // this step's h as tagged words, [unit chunk][row][4 units]: the publish order
    float *Gx = reinterpret_cast<float *>(Og + 16 * UW);            // [NBT][2 step parities][16 rows][UW units][4 gates] gate pre-activations (LDS-DMA)
    constexpr int GXT = 16 * UW * 4;                                // floats per tile slab
    constexpr int PANB = 16 * H * 4;                                // bytes of one tile panel
    unsigned char *Rw = reinterpret_cast<unsigned char *>(Gx + NBT * 2 * GXT);   // [2][PANB] tile panels as they travel (LDS-DMA), alternating per phase
    static_assert(UW % 8 == 0 && (16 * UW) % NTH == 0, "output rows are whole 16-byte chunks; the gx slab is whole 1 KB wave loads");
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int w = blockIdx.x, xl = w & 7, j = w >> 3;
    const int team = xl * 4 + (j >> 3), member = j & 7;            // 8 members share blockIdx%8 (one XCD under round-robin)
    const int d = team >> 4, g = team & 15;
    const int B = a.B, T = a.T;
    if (tid == 0) s_fail = 0;

    bf16x8 ah[RTW][KS], al[RTW][KS];
    float osc[RTW], osh[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; rt++) {
        const int r0 = member * RM + (wave * RTW + rt) * 16;
        const int unit = (r0 >> 2) + kq;
        osc[rt] = a.oscale ? a.oscale[d * H + unit] : 1.f;
        osh[rt] = a.oscale ? a.oshift[d * H + unit] : 0.f;
        const unsigned short *wh = a.whh.hi + ((size_t)d * 4 * H + r0 + li) * H + kq * 8;
        const unsigned short *wl = a.whh.lo + ((size_t)d * 4 * H + r0 + li) * H + kq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ah[rt][ks] = *reinterpret_cast<const bf16x8 *>(wh + ks * 32);
            al[rt][ks] = *reinterpret_cast<const bf16x8 *>(wl + ks * 32);
        }
    }
    float cst[RTW][NBT];
#pragma unroll
    for (int rt = 0; rt < RTW; rt++)
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) cst[rt][bt] = 0.f;
    int slen[NBT];                                                  // steps valid for this lane's batch row of tile bt (fused batches of different lengths)
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) {
        const int lb = bt * 16 + li, row = g * a.BGr + lb;
        slen[bt] = (a.seqlen && lb < a.BGr && row < B) ? a.seqlen[row] : T;
    }
    // A tile panel in the exchange buffer and in LDS: 16-byte chunks {4 tagged words = 4 consecutive units of one row},
    // chunk index = (unit / 4) * 16 + row.  A workgroup's share (its UW units x 16 rows) is then ONE contiguous run, the
    // sweep is a linear copy, and the MFMA operand of lane (row li, k-quarter kq) at k-step ks is the two chunks
    // (ks * 8 + kq * 2 + {0, 1}) * 16 + li: the 16 lanes of a quarter read 256 contiguous bytes (no bank conflicts).
    constexpr size_t tgran = (size_t)16 * H / 2;                   // 8-byte granules per tile panel
    const size_t pgran = NBT * tgran;                              // granules per (parity, team)
    constexpr int NLD = 16 * H / 4 / NTH;                          // 16-byte chunks per thread and tile
    static_assert(16 * H / 4 % NTH == 0 && H % 32 == 0, "panel must be whole passes of the workgroup");
    u64 *hxg = reinterpret_cast<u64 *>(a.hx);
    unsigned int *abortf = a.sync + 16;
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tst = a.dbg ? (long long)__builtin_readcyclecounter() : 0;

    // gate pre-activations of (tile bt, time tt) -> Gx[bt]: one LDS-DMA piece per row tile of the wave, and lane (row li,
    // unit kq) of piece rt fetches exactly the 16 bytes (i, f, g, o of its unit and row) that the same lane consumes in
    // the cell update of row tile rt: nothing crosses waves, so no barrier stands between the transfer and its use
    constexpr int NGX = RTW;                                        // 1 KB wave loads per slab and wave
    const unsigned gvoff = (unsigned)(d * 4 * H + (member * UW + wave * RTW * 4 + kq) * 4) * 4u;   // byte offset of row tile 0's unit in a gx row
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)wave * 1024u);   // a wave's 1 KB slot inside a 4 KB LDS-DMA pass
    const unsigned wave_gx = __builtin_amdgcn_readfirstlane((unsigned)wave * (unsigned)(RTW * 1024));   // a wave's RTW row tiles inside a slab
    const unsigned gx_lds = (unsigned)(unsigned long long)(lds_void_t *)Gx, rw_lds = (unsigned)(unsigned long long)(lds_void_t *)Rw;
    auto load_gx = [&](int bt, int par, int tt, int i0 = 0, int i1 = 99) {
        i1 = i1 > NGX ? NGX : i1;
        const float *gbase = a.gx + (size_t)tt * B * 2 * 4 * H;
        const int b = min(g * a.BGr + min(bt * 16 + li, a.BGr - 1), B - 1);           // rows past the batch read a valid row (never used)
        const unsigned rowoff = (unsigned)b * (unsigned)(2 * 4 * H * 4) + gvoff;
#pragma unroll
        for (int i = i0; i < i1; i++)
            lds_dma16_s<false>(gbase, rowoff + (unsigned)(i * 64), gx_lds + (unsigned)(((bt * 2 + par) * GXT + i * 256) * 4) + wave_gx);
    };
    // Layer outputs and the publish are read back from the LDS tiles as 16- or 8-byte pieces and leave as buffer stores.
    // Every wave stores exactly the units it produced (its RTW row tiles = 4*RTW consecutive units of all 16 rows): the
    // tiles never cross waves, so no barrier stands between the cell update and the stores either.  Every wave issues a
    // fixed number of store instructions per destination (lanes without a piece, and nothing else, are dropped by the
    // buffer range check; lanes of rows past the batch repeat the tile's last valid row: identical bytes to the same
    // address), so that the number of memory operations a wave issues after a sweep request is known exactly -- see
    // the counted wait below.  The LDS reads of a phase are issued together, ahead of the stores.
    constexpr int UWW = UW / 4, PCW = UWW / 4;                          // units per wave; 4-unit pieces per row (16 B of fp32, 8 B of a bf16 plane)
    constexpr int PWR = 16 * PCW, PWP = 16 * (UW / 4) / 4;               // pieces per wave: layer outputs (per destination), publish chunks
    static_assert(UWW % 4 == 0 && PWR <= 64 && PWP == 16 * RTW, "a wave's output pieces fit one instruction; it publishes its own row tiles");
    const int orow = lane / PCW, ocol = wave * UWW + (lane - orow * PCW) * 4;   // output role of this lane: tile row, first of its 4 units
    const int qp = min(wave * PWP + lane, 16 * (UW / 4) - 1);
    const unsigned offp = lane < PWP ? (unsigned)(member * 16 * (UW / 4) + wave * PWP + lane) * 16u : 0xffffffffu;
    const size_t slab = (size_t)B * 2 * H;                              // elements of one time step of the layer output
    auto tile_rows = [&](int bt) { const int nv = min(a.BGr, B - g * a.BGr) - bt * 16; return nv < 0 ? 0 : (nv > 16 ? 16 : nv); };
    struct OutRegs { u32x2 vh, vl; u32x4 vr; int r; };
    auto out_read = [&](int bt, OutRegs &o) {
        o.r = min(orow, max(tile_rows(bt), 1) - 1);
        if (a.out_split.hi) {
            o.vh = *reinterpret_cast<const u32x2 *>(Oh + o.r * UW + ocol);
            o.vl = *reinterpret_cast<const u32x2 *>(Ol + o.r * UW + ocol);
        }
        if (a.out_raw) o.vr = *reinterpret_cast<const u32x4 *>(Of + o.r * UW + ocol);
    };
    auto out_write = [&](int bt, int tt, const OutRegs &o) -> int {
        if (tile_rows(bt) == 0) return 0;
        int issued = 0;
        const unsigned el = lane < PWR ? (unsigned)((g * a.BGr + bt * 16 + o.r) * 2 * H + d * H + member * UW + ocol) : 0x3fffffffu;   // element offset in the step's slab (out of range: dropped)
        if (a.out_split.hi) {
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.out_split.hi + (size_t)tt * slab, 0, (int)(slab * 2), 0x00020000);
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(a.out_split.lo + (size_t)tt * slab, 0, (int)(slab * 2), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(o.vh, rh, el * 2u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(o.vl, rl, el * 2u, 0, 0);
            issued += 2;
        }
        if (a.out_raw) {
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(a.out_raw + (size_t)tt * slab, 0, (int)(slab * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(o.vr, rs_, el * 4u, 0, 0);
            issued++;
        }
        return issued;
    };
    auto store_out = [&](int bt, int tt) -> int { OutRegs o; out_read(bt, o); return out_write(bt, tt, o); };
    // Sweep request of (tile bt, step s) into panel buffer pb: a linear LDS-DMA copy of the tile's panel (parity (s-1)&1),
    // NLD pieces per wave.  With two or more tiles it is issued a phase ahead; it is waited for with vmcnt(K), K = the
    // exact number of memory instructions this wave has issued since (the wave's memory counter is in-order: vmcnt(0)
    // would also wait for the acknowledgement of every store issued after the request -- the write-through publish
    // alone takes ~1 us).
    auto request_sweep = [&](int bt, int s, int pb, int i0 = 0, int i1 = 99) {
        i1 = i1 > NLD ? NLD : i1;
        const unsigned char *srcp = reinterpret_cast<const unsigned char *>(hxg + (size_t)(((s - 1) & 1) * 32 + team) * pgran + bt * tgran);
#pragma unroll
        for (int i = i0; i < i1; i++) lds_dma16_s<true>(srcp + (size_t)i * NTH * 16, (unsigned)tid * 16u, rw_lds + (unsigned)(pb * PANB + i * NTH * 16) + wave_lds);
    };
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) load_gx(bt, 0, d ? (T - 1) : 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), compiler-visible: the weight fragments are in registers from here on (no waits for them inside the loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    int pend_bt = -1, pend_t = 0;                                   // PF == 0: the phase whose outputs still sit in the LDS tiles
    bool requested = false;                                         // the panel of the phase about to start is in flight / staged
    int younger = 0, early_gx = 0;                                  // memory instructions this wave issued after that request
    int pc = 0;                                                     // phases with a sweep so far: panel buffer = pc & 1
    for (int s = 0; s < T; s++) {
        const int t = d ? (T - 1 - s) : s;
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) {
            const int nbt = bt + 1 < NBT ? bt + 1 : 0, ns = bt + 1 < NBT ? s : s + 1;     // the next phase
            f32x4 acc[RTW];
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
                const int pb = pc & 1;
                ++pc;
                const unsigned ep = (unsigned)((s - 1) % 3 + 1), etag = (ep & 1u) | ((ep >> 1) << 16);   // tag of step s-1: bit 0 in lo0, bit 1 in lo1
                long long t0 = 0;   // wall clock at the first retry (read lazily: its scalar load would sit in front of every phase's first barrier)
                // ---- the tile's panel of step s-1 -> LDS
                if (PF == 0) {
                    // one tile: nothing to overlap.  Each wave polls its own pieces (read back through LDS) until all tags match.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own publish acknowledged: a request sent earlier only finds stale tags
                    request_sweep(bt, s, pb);
                    wait_vmcnt(pend_bt >= 0 ? store_out(pend_bt, pend_t) : 0);   // the previous phase's outputs, behind the sweep in the queue
                    int polls = 0;
                    while (true) {
                        unsigned bad = 0;
#pragma unroll
                        for (int i = 0; i < NLD; i++) {
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(Rw + (size_t)pb * PANB + (size_t)(i * NTH + tid) * 16);
                            bad |= (v[2] ^ etag) | (v[3] ^ etag);
                        }
                        ++polls;
                        if (!__any((bad & 0x00010001u) != 0)) break;
                        if ((polls & 63) == 0) {
                            int ab = 0;
                            if (t0 == 0) t0 = wall_clock64();
                            if (lane == 0) ab = (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) || (wall_clock64() - t0 > 200000000ll);
                            if (__any(ab)) {
                                if (lane == 0) { __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicExch(a.err_flag, 2); s_fail = 1; }
                                break;
                            }
                        }
                        request_sweep(bt, s, pb);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (a.dbg) ph[5] += polls;
                } else {
                    // requested a phase ago (or just now: the first step after s = 0)
                    if (!requested) { request_sweep(bt, s, pb); younger = 0; }
                    const long long tw = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
                    wait_vmcnt(younger);
                    if (a.dbg) ph[1] += (long long)__builtin_readcyclecounter() - tw;
                }
                requested = false;
                lds_barrier();                                          // every wave's pieces of the panel are in LDS
                PSTAMP(0);
                if (s_fail) return;
                // The LDS-DMA pieces of this phase -- this tile's next gx slab (other parity buffer), then, with three or more
                // tiles, the next tile's sweep (its panel was published a whole phase ago) -- are issued one per k-step inside
                // the MFMA loop: a piece costs the wave ~100 cycles of issue, which the MFMAs in flight cover.  The slab goes
                // first: nothing slow may be younger than the sweep request (see the counted wait at its consumption).
                const bool early = PF != 0 && a.early && ns < T;
                if (early) { request_sweep(nbt, ns, pb ^ 1); requested = true; }
                const bool do_rq = PF == 2 && ns < T && !early;
                const int tt_next = d ? max(T - 2 - s, 0) : min(s + 1, T - 1);   // the step whose slab this phase requests
                // The MFMA operands come straight from the travelling chunks (no unpacking pass, no second copy in LDS): per
                // k-step two 16-byte chunks {hi x4 | lo' x4} -> the hi halves of both are one operand, the lo halves (tag
                // bits cleared: 4 v_and) the other.  One tagged word per chunk is summed on the way (bit 0 and bit 16
                // fields; 2*KS words per lane cannot overflow a field).  With two or more tiles the panel was requested
                // ahead and is used unchecked: if the sum is off, some member's h had not landed when the request was
                // served; every wave sees the same panel, so all take the same decision: request it again and redo the
                // tile's products (not seen with the request placed after the MFMAs: it trails the publish by ~2 us).
                const unsigned char *fb = Rw + (size_t)pb * PANB + kq * 512 + li * 16;
                int tries = 0;
                while (true) {
                    constexpr int PD = 3;                             // chunk pairs read ahead of their use
                    u32x4 ra[PD], rb[PD];
#pragma unroll
                    for (int p = 0; p < PD; p++) {
                        ra[p] = *reinterpret_cast<const u32x4 *>(fb + p * 2048);
                        rb[p] = *reinterpret_cast<const u32x4 *>(fb + p * 2048 + 256);
                    }
                    unsigned sraw = 0, smask = 0;                     // sums of the lo words with / without their tag bits
                    MDD_SCHED_HINT();
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const u32x4 xa = ra[ks % PD], xb = rb[ks % PD];
                        if (ks + PD < KS) {
                            ra[ks % PD] = *reinterpret_cast<const u32x4 *>(fb + (ks + PD) * 2048);
                            rb[ks % PD] = *reinterpret_cast<const u32x4 *>(fb + (ks + PD) * 2048 + 256);
                        }
                        u32x4 hq, lq;
                        hq[0] = xa[0]; hq[1] = xa[1]; hq[2] = xb[0]; hq[3] = xb[1];
                        lq[0] = xa[2] & 0xfffefffeu; lq[1] = xa[3] & 0xfffefffeu; lq[2] = xb[2] & 0xfffefffeu; lq[3] = xb[3] & 0xfffefffeu;
                        if (PF != 0) { sraw += xa[2] + xb[2]; smask += lq[0] + lq[2]; }   // one tagged word per 16-byte chunk: a chunk is one lane's single store
                        bf16x8 bh = __builtin_bit_cast(bf16x8, hq), bl = __builtin_bit_cast(bf16x8, lq);
                        // The slab pieces are issued unconditionally (a branch per k-step costs the loop more than the piece: -2.6 % on the
                        // product section): at the last step they fetch a valid slab nobody reads, in a redo they fetch the same slab again.
                        // (Round 2 kept a guard around them in the three- and four-tile forms because lifting it changed last bits there: that
                        // was the hidden-hazard class described at mfma_v() -- another code shape, another schedule around the asm MFMA --
                        // not the request itself; with the hazards expressed the bits stay.)  The sweep pieces of the three- and four-tile
                        // forms stay conditional: after a redo, or with no next phase, there is nothing to request.
                        if (ks < NGX) load_gx(bt, (s + 1) & 1, tt_next, ks, ks + 1);
                        else if (PF == 2 && tries == 0 && ks < NGX + NLD) { if (do_rq) request_sweep(nbt, ns, pb ^ 1, ks - NGX, ks - NGX + 1); }
                        // product-major order: consecutive MFMAs hit different accumulators (no dependent-issue stall);
                        // each accumulator still sees ah.bl, al.bh, ah.bh in that order (bit-identical to the step kernel)
                        // (the scheduler would otherwise regroup them per accumulator into dependent back-to-back pairs)
                        bf16x8 ak[RTW];
#pragma unroll
                        for (int rt = 0; rt < RTW; rt++) ak[rt] = ah[rt][ks];
                        mfma_operands_ready(bh, bl, ak);
#pragma unroll
                        for (int rt = 0; rt < RTW; rt++) {   // the first product of a tile starts from a literal zero: no accumulator clearing per phase (or per redo)
                            if (ks == 0) mfma_v0(acc[rt], ak[rt], bl); else mfma_v(acc[rt], ak[rt], bl);
                        }
                        MDD_SCHED_HINT();
#pragma unroll
                        for (int rt = 0; rt < RTW; rt++) mfma_a(acc[rt], al[rt][ks], bh);   // lo fragments stay in AGPRs and feed the MFMA from there
                        MDD_SCHED_HINT();
#pragma unroll
                        for (int rt = 0; rt < RTW; rt++) mfma_v(acc[rt], ak[rt], bh);
                        MDD_SCHED_HINT();
                    }
                    const unsigned tsum = sraw - smask;
                    static_assert(NGX + NLD <= KS, "one LDS-DMA piece per k-step");
                    if (PF == 0 || !__any(tsum != (unsigned)(2 * KS) * etag)) break;
                    // ---- stale panel (identical verdict in every wave): fetch it again, redo the products
                    ++tries;
                    if (a.dbg) ph[5] += 1;
                    if (t0 == 0) t0 = wall_clock64();
                    if (lane == 0 && ((__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) || (wall_clock64() - t0 > 200000000ll))) {
                        __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicExch(a.err_flag, 2); s_fail = 1;
                    }
                    lds_barrier();                                      // every wave is done reading the panel; s_fail is visible
                    if (s_fail) return;
                    request_sweep(bt, s, pb);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_barrier();
                }
                if (a.dbg && PF != 0) ph[5] += 1;
                if (do_rq) requested = true;
                int in_flight = NGX + (PF != 2 ? ((early && !tries) ? NLD : 0)      // (a redo's vmcnt(0) came before its own slab pieces)
                                               : ((!tries && (do_rq || early)) ? NLD : 0));
                early_gx = (early && !tries) ? NGX : 0;                             // memory instructions younger than an early request
                if (PF == 1 && ns < T && !early) {   // two tiles: the other tile's panel has had the length of these MFMAs to arrive
                    request_sweep(nbt, ns, pb ^ 1); requested = true;
                    in_flight += NLD;
                }
                wait_vmcnt(in_flight);   // everything older than this phase's pieces has landed: in particular this wave's part of the tile's current gx slab
            }
            if (s == 0 && T > 1) load_gx(bt, 1, d ? (T - 2) : 1);
            PSTAMP(2);
            // ---- cell update; h_s goes to the LDS tiles (outputs, and tagged words for the team)
            const unsigned tg = (unsigned)(s % 3 + 1);
            const int lb = bt * 16 + li;
            const bool valid = lb < a.BGr && g * a.BGr + lb < B;
            float4 gv[RTW];
            float hn[RTW];
            mfma_drain();                        // (the asm MFMAs' results are read by vector instructions from here on)
#pragma unroll
            for (int rt = 0; rt < RTW; rt++)     // the slab reads together (one LDS round trip), each lane its own 16 bytes
                gv[rt] = *reinterpret_cast<const float4 *>(Gx + (bt * 2 + (s & 1)) * GXT + ((wave * RTW + rt) * 64 + lane) * 4);
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {   // branch-free: the row tiles' chains interleave
                const float ig = fast_sigmoid(acc[rt][0] + gv[rt].x), fg = fast_sigmoid(acc[rt][1] + gv[rt].y);
                const float cg = fast_tanh(acc[rt][2] + gv[rt].z), og = fast_sigmoid(acc[rt][3] + gv[rt].w);
                const bool live = !(d && t >= slen[bt]);              // the reverse direction starts at the row's own last step, from a zero state
                const float cn = fg * cst[rt][bt] + ig * cg;
                const float hr = og * fast_tanh(cn);
                hn[rt] = (valid && live) ? hr : 0.f;
                cst[rt][bt] = live ? cn : 0.f;
                if (TRAIN) {   // what the backward step reads: the gates after their nonlinearities and c_t, in the training layout (one 64-byte run per row and quad)
                    const unsigned el = valid ? (unsigned)(((g * a.BGr + lb) * 2 + d) * H + member * UW + (wave * RTW + rt) * 4 + kq) : 0x0fffffffu;
                    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(a.gates_save + (size_t)t * B * 2 * H * 4, 0, B * 2 * H * 16, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rc_ = __builtin_amdgcn_make_buffer_rsrc(a.c_save + (size_t)t * B * 2 * H, 0, B * 2 * H * 4, 0x00020000);
                    u32x4 gq;
                    gq[0] = __float_as_uint(ig); gq[1] = __float_as_uint(fg); gq[2] = __float_as_uint(cg); gq[3] = __float_as_uint(og);
                    __builtin_amdgcn_raw_buffer_store_b128(gq, rg, el * 16u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cn), rc_, el * 4u, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {
                const int cl = wave * RTW + rt, ul = cl * 4 + kq;        // chunk column and unit inside the workgroup's share
                const float ov = hn[rt] * osc[rt] + osh[rt];
                __bf16 ob = (__bf16)ov, ol = (__bf16)(ov - (float)ob);
                Oh[li * UW + ul] = *reinterpret_cast<unsigned short *>(&ob);
                Ol[li * UW + ul] = *reinterpret_cast<unsigned short *>(&ol);
                if (a.out_raw) Of[li * UW + ul] = hn[rt];
                const unsigned hw = split_h(hn[rt]);                   // chunk = {hi0 hi1 hi2 hi3 | lo0' lo1' lo2' lo3'}: tag bit 0 rides in the even units' lo, bit 1 in the odd units'
                unsigned short *ogp = reinterpret_cast<unsigned short *>(Og) + (cl * 16 + li) * 8 + kq;
                ogp[0] = (unsigned short)(hw & 0xffffu);
                ogp[4] = (unsigned short)((hw >> 16) | ((kq & 1) ? (tg >> 1) : (tg & 1u)));
            }
            PSTAMP(3);
            // ---- publish h_s (no drain, no signal): the workgroup's share of the tile's panel is one contiguous run of
            // 16 * UW / 4 chunks, written as 16-byte write-through stores straight from the Og tile (same order)
            younger = early_gx + (TRAIN ? 2 * RTW : 0); early_gx = 0;   // (TRAIN: the cell update's stores are younger than the sweep request too)
            {
                const bool pub = s + 1 < T, defer = PF == 0 && s > 0 && s + 1 < T;
                const u32x4 pv = *reinterpret_cast<const u32x4 *>(Og + qp * 4);
                OutRegs o;
                if (!defer) out_read(bt, o);
                if (pub) {   // all 16 rows of the tile (rows past the batch carry zeros and valid tags)
                    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(hxg + (size_t)((s & 1) * 32 + team) * pgran + bt * tgran, 0, (int)(tgran * 8), 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(pv, drs, offp, 0, 16 /* sc1 */);
                    younger++;
                }
                // layer outputs: with one tile they wait for the next sweep's request (nothing else to hide them behind);
                // otherwise the next sweep is already in flight and they go out now
                if (defer) { pend_bt = bt; pend_t = t; }
                else { younger += out_write(bt, t, o); pend_bt = -1; }
            }
            if (s == 0) {   // no MFMA section with its barriers follows before these LDS regions are reused
                if (bt + 1 == NBT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
            }
            PSTAMP(4);
        }
    }
    if (pend_bt >= 0) store_out(pend_bt, pend_t);
    if (a.dbg && tid == 0) for (int i = 0; i < 6; i++) a.dbg[blockIdx.x * 6 + i] = ph[i];
}

int granule_bg(int B) { const int r = (B + 15) / 16; return (r + 15) / 16 * 16; }   // rows per batch group (16 groups), padded to whole tiles


__global__ void zero_fill_kernel(u32x4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = (u32x4){0u, 0u, 0u, 0u};
}
// The persistent layer kernels need their exchange buffer's tags and their abort word zeroed before every launch.  Inside a captured graph a
// hipMemsetAsync becomes a memset NODE, and replays of such graphs were measured to let a layer kernel start on the previous launch's
// contents (tools/fused_repro3.py, exact-fp32 mode, small batches: the second and later mdd_forward_fused calls gave wrong layer-0 outputs,
// up to 2e-2, in 7 of 12 and 4 of 14 processes; in none of 12 with MDD_GRAPH=0; in none of 14 with this kernel in the memset's place;
// MDD_ZERO_BY_MEMSET=1 brings the memset nodes back for study).  A kernel node orders like every other kernel of the chain.
int launch_zero_fill(void *p, size_t n, hipStream_t st) {
    static const bool by_memset = [] { const char *e = getenv("MDD_ZERO_BY_MEMSET"); return e && e[0] == '1'; }();
    if (by_memset) { MDD_HIP_CHECK(hipMemsetAsync(p, 0, n, st)); return MDD_OK; }
    const size_t n16 = n / 16;
    const int blocks = (int)std::min<size_t>((n16 + 255) / 256, 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, reinterpret_cast<u32x4 *>(p), n16);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

template <int H, int NBT, int RTW, bool TRAIN = false>
static int launch_granule_t(PersistArgs a, hipStream_t st) {
    const size_t smem = (size_t)16 * (H / 8) * 12 + (size_t)NBT * 2 * 16 * (H / 8) * 16 + (size_t)2 * 16 * H * 4;   // tiles (2 bf16 planes, fp32, tagged words) + gx slabs (2 parities) + two panel buffers
    if (int rc = launch_zero_fill(a.sync, 32 * sizeof(unsigned int), st)) return rc;
    if (int rc = launch_zero_fill(a.hx, (size_t)2 * 32 * NBT * 16 * H * 4, st)) return rc;   // tags must start at 0 on every launch
    hipLaunchKernelGGL((lstm_layer_granule_kernel<H, NBT, RTW, TRAIN>), dim3(kPersistGrid), dim3(256), smem, st, a);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int launch_lstm_layer_granule(const LstmStepArgs &s, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st) {
    PersistArgs a;
    a.gx = s.gx; a.whh = s.whh_split; a.whh_f32 = nullptr; a.hx = hx; a.sync = sync; a.err_flag = err_flag;
    a.out = s.out; a.out_raw = s.out_raw; a.out_split = s.out_split; a.oscale = s.oscale; a.oshift = s.oshift;
    a.T = s.T; a.B = s.B; a.BGr = (s.B + 15) / 16; a.BG = granule_bg(s.B); a.seqlen = s.seqlen;
    a.dbg = (getenv("MDD_LSTM_DBG") && s.T > 100) ? reinterpret_cast<long long *>(reinterpret_cast<u64 *>(hx) + (size_t)2 * 32 * a.BG * s.H) : nullptr;
    a.early = getenv("MDD_LSTM_EARLY") != nullptr;
    if (a.oscale == nullptr) a.oshift = nullptr;
    if (a.out && a.out != a.out_raw) { set_error("granule lstm: a separate scaled fp32 output is not supported (split planes carry it)"); return MDD_ERR_ARG; }
    a.gates_save = s.gates_save; a.c_save = s.c_save;
    const int nbt = a.BG / 16;
    if (s.gates_save) {   // training forward: gates and cell states are saved for the backward pass (B <= 512)
        if (!s.c_save || a.seqlen || (size_t)s.B * 2 * s.H * 16 > 0x7fffffffu) { set_error("granule lstm (train): bad arguments"); return MDD_ERR_ARG; }
        if (s.H == 384 && nbt == 1) return launch_granule_t<384, 1, 3, true>(a, st);
        if (s.H == 384 && nbt == 2) return launch_granule_t<384, 2, 3, true>(a, st);
        if (s.H == 256 && nbt == 1) return launch_granule_t<256, 1, 2, true>(a, st);
        if (s.H == 256 && nbt == 2) return launch_granule_t<256, 2, 2, true>(a, st);
        set_error("granule lstm (train): built for H in {256,384}, B <= 512 (H=%d B=%d)", s.H, s.B);
        return MDD_ERR_ARG;
    }
    if (nbt < 1 || nbt > 4) { set_error("granule lstm: B=%d needs %d row tiles per team (max 4)", s.B, nbt); return MDD_ERR_ARG; }
    if (s.H == 384) return nbt == 1 ? launch_granule_t<384, 1, 3>(a, st) : nbt == 2 ? launch_granule_t<384, 2, 3>(a, st)
                         : nbt == 3 ? launch_granule_t<384, 3, 3>(a, st) : launch_granule_t<384, 4, 3>(a, st);
    if (s.H == 256) return nbt == 1 ? launch_granule_t<256, 1, 2>(a, st) : nbt == 2 ? launch_granule_t<256, 2, 2>(a, st)
                         : nbt == 3 ? launch_granule_t<256, 3, 2>(a, st) : launch_granule_t<256, 4, 2>(a, st);
    set_error("granule lstm: unsupported H=%d", s.H);
    return MDD_ERR_ARG;
}


// ------------------------------------------------------------------------------------------------ persistent BPTT layer kernel
// The backward recurrence of one bidirectional layer in ONE launch (the split-bf16 training variant; the exact mode keeps one launch
// per step, train_kernels.hip).  Same grid and teams as the forward layer kernel above (256 workgroups = 2 directions x 16 batch groups
// x 8 members, one 16-row batch tile per team: B <= 256), the same data-tagged hand-off, the roles transposed:
//   dh[b, k] = dout[t, b, k] + sum_n DG_prev[b, n] * Whh'[n][k]            (n over the 4H gate rows, k over the H units)
// A member owns UW = H/8 units k.  Its rows of Whh'^T ([UW][4H], bf16 hi/lo) stay in registers for the whole launch, the
// contraction's 4H axis split over the four waves (a quarter each, all UT = UW/16 unit tiles), partial tiles summed through LDS.
// What travels between the members is DG (the gate gradients of the step before): a panel of 4H values per batch row, four times
// the forward kernel's, as 16-byte chunks {hi x4 | lo x4} of the four gates of ONE unit and row -- exactly what one thread of the
// cell backward produces, so a chunk goes from the producing thread's registers straight to the exchange buffer (tag bits in the
// spare last bits of the lo halves, as above).  DG also leaves in fp32 (the weight-gradient and dX products read that).
struct BwdPersistArgs {
    const float *dout, *gates, *cst;     // [T][B][2H];  [T][B][2][H][4];  [T][B][2][H]
    SplitPtr whhT;                       // [2][H][4H] hi/lo: row k of direction d = column k of Whh'_d
    float *dg;                           // [T][B][2][4H]
    unsigned short *hx;                  // [2 parity][32 teams][4H/4 chunk columns][16 rows] x 16 B
    unsigned int *sync;                  // [16] unused, [16] abort flag (zeroed before every launch)
    int *err_flag;
    int T, B, BGr;
};

template <int H>
__global__ __launch_bounds__(256, 1) void lstm_bwd_granule_kernel(BwdPersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTH = 256, UW = H / 8, UT = UW / 16, KS = H / 32, G = 4 * H;
    constexpr int ROWB = G * 4;                                     // bytes of one batch row of a panel: G/4 chunks of 16 B
    constexpr int PANB = 16 * ROWB;                                 // bytes of a panel
    constexpr int NE = 16 * UW / NTH;                               // (row, unit) elements per thread in the cell backward
    static_assert(UW % 16 == 0 && (16 * UW) % NTH == 0 && PANB % (16 * NTH) == 0 && H % 32 == 0, "geometry");
    // Panel layout, in the exchange buffer and in LDS alike (the sweep is a linear copy): ROW-major, so that only the team's real rows
    // travel (2 of 16 at B = 32); inside a row the chunk of unit u sits at position u ^ row: the 16 lanes of an MFMA operand read
    // (same chunk column, rows 0..15) then hit 16 different 16-byte bank groups instead of one.
    unsigned char *Rw = smem;
    float *red = reinterpret_cast<float *>(smem + PANB);            // [4 waves][UW units][16 rows] partial dh
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int w = blockIdx.x, xl = w & 7, j = w >> 3;
    const int team = xl * 4 + (j >> 3), member = j & 7;
    const int d = team >> 4, g = team & 15;
    const int B = a.B, T = a.T;
    const int nrows = max(0, min(a.BGr, B - g * a.BGr));            // real rows of this team's tile (workgroup-uniform)
    if (tid == 0) s_fail = 0;

    bf16x8 ah[UT][KS], al[UT][KS];
#pragma unroll
    for (int ut = 0; ut < UT; ut++) {
        const size_t ro = ((size_t)d * H + member * UW + ut * 16 + li) * G + wave * H + kq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ah[ut][ks] = *reinterpret_cast<const bf16x8 *>(a.whhT.hi + ro + ks * 32);
            al[ut][ks] = *reinterpret_cast<const bf16x8 *>(a.whhT.lo + ro + ks * 32);
        }
    }
    // the cell backward's elements of this thread: e = tid + i * 256 -> (tile row e / UW, unit e % UW of the member's share): the real
    // rows come first, so with few rows most threads have nothing to do after the first element
    int ul[NE], row[NE], brow[NE];
    bool valid[NE];
    float dcar[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const int e = tid + i * NTH;
        row[i] = e / UW; ul[i] = e - row[i] * UW;
        valid[i] = row[i] < nrows;
        brow[i] = min(g * a.BGr + min(row[i], max(nrows, 1) - 1), B - 1);
        dcar[i] = 0.f;
    }
    constexpr size_t tgran = PANB / 8;                              // 8-byte granules per (parity, team)
    u64 *hxg = reinterpret_cast<u64 *>(a.hx);
    unsigned int *abortf = a.sync + 16;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)wave * 1024u);
    const unsigned rw_lds = (unsigned)(unsigned long long)(lds_void_t *)Rw;
    const int npc = (nrows * ROWB + NTH * 16 - 1) / (NTH * 16);     // 4 KB passes that cover the real rows (workgroup-uniform)
    auto request_sweep = [&](int s) {
        const unsigned char *srcp = reinterpret_cast<const unsigned char *>(hxg + (size_t)(((s - 1) & 1) * 32 + team) * tgran);
        for (int i = 0; i < npc; i++) lds_dma16_s<true>(srcp + (size_t)i * NTH * 16, (unsigned)tid * 16u, rw_lds + (unsigned)(i * NTH * 16) + wave_lds);
    };
    // operands of the cell backward at step s: they do not depend on the recurrence and are requested a step ahead
    float4 pg[NE];
    float pc[NE], pcp[NE], pdo[NE];
    auto prefetch = [&](int s) {
        const int t = d ? s : (T - 1 - s), tp = d ? t + 1 : t - 1;
#pragma unroll
        for (int i = 0; i < NE; i++) {
            if (!valid[i]) continue;
            const int unit = member * UW + ul[i];
            const size_t si = (((size_t)t * B + brow[i]) * 2 + d) * H + unit;
            pg[i] = *reinterpret_cast<const float4 *>(a.gates + si * 4);
            pc[i] = a.cst[si];
            pcp[i] = (tp >= 0 && tp < T) ? a.cst[(((size_t)tp * B + brow[i]) * 2 + d) * H + unit] : 0.f;
            pdo[i] = a.dout[((size_t)t * B + brow[i]) * 2 * H + d * H + unit];
        }
    };
#pragma unroll
    for (int i = 0; i < NE; i++) { pg[i] = make_float4(0.f, 0.f, 0.f, 0.f); pc[i] = pcp[i] = pdo[i] = 0.f; }
    prefetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    for (int s = 0; s < T; s++) {
        const int t = d ? s : (T - 1 - s);
        if (s > 0 && nrows > 0) {
            const unsigned ep = (unsigned)((s - 1) % 3 + 1), etag = (ep & 1u) | ((ep >> 1) << 16);
            long long t0 = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own publish acknowledged (a request sent earlier only finds stale tags); prefetch landed
            request_sweep(s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int polls = 0;
            while (true) {                                          // each wave polls its own pieces (read back through LDS) until all tags match
                unsigned bad = 0;
                for (int i = 0; i < npc; i++) {
                    const int off = (i * NTH + tid) * 16;
                    if (off < nrows * ROWB) {                       // (the last pass may run past the real rows)
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(Rw + off);
                        bad |= (v[2] ^ etag) | (v[3] ^ etag);
                    }
                }
                ++polls;
                if (!__any((bad & 0x00010001u) != 0)) break;
                if ((polls & 63) == 0) {
                    int ab = 0;
                    if (t0 == 0) t0 = wall_clock64();
                    if (lane == 0) ab = (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) || (wall_clock64() - t0 > 200000000ll);
                    if (__any(ab)) {
                        if (lane == 0) { __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicExch(a.err_flag, 2); s_fail = 1; }
                        break;
                    }
                }
                request_sweep(s);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();                                          // every wave's pieces of the panel are in LDS; s_fail is visible
            if (s_fail) return;
            // ---- partial dh of this wave's quarter of the gate axis: operands straight from the travelling chunks.  Lane (row li,
            // k-quarter kq) at k-step ks needs chunk columns c, c + 1 (c = wave * H/4 + ks * 8 + kq * 2, even) of row li: positions
            // c ^ li and (c ^ li) ^ 1.  Rows past the team's real ones were never fetched: whatever LDS holds there only reaches
            // the output columns of those rows, which nobody reads.
            f32x4 acc[UT];
            const unsigned char *rb_ = Rw + (size_t)li * ROWB;
            const int c0 = wave * (H / 4) + kq * 2;
            constexpr int PD = 3;
            u32x4 ra[PD], rb[PD];
#pragma unroll
            for (int p = 0; p < PD; p++) {
                const int c = (c0 + p * 8) ^ li;
                ra[p] = *reinterpret_cast<const u32x4 *>(rb_ + c * 16);
                rb[p] = *reinterpret_cast<const u32x4 *>(rb_ + (c ^ 1) * 16);
            }
            MDD_SCHED_HINT();
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const u32x4 xa = ra[ks % PD], xb = rb[ks % PD];
                if (ks + PD < KS) {
                    const int c = (c0 + (ks + PD) * 8) ^ li;
                    ra[ks % PD] = *reinterpret_cast<const u32x4 *>(rb_ + c * 16);
                    rb[ks % PD] = *reinterpret_cast<const u32x4 *>(rb_ + (c ^ 1) * 16);
                }
                u32x4 hq, lq;
                hq[0] = xa[0]; hq[1] = xa[1]; hq[2] = xb[0]; hq[3] = xb[1];
                lq[0] = xa[2] & 0xfffefffeu; lq[1] = xa[3] & 0xfffefffeu; lq[2] = xb[2] & 0xfffefffeu; lq[3] = xb[3] & 0xfffefffeu;
                bf16x8 bh = __builtin_bit_cast(bf16x8, hq), bl = __builtin_bit_cast(bf16x8, lq);
                bf16x8 ak[UT];
#pragma unroll
                for (int ut = 0; ut < UT; ut++) ak[ut] = ah[ut][ks];
                mfma_operands_ready(bh, bl, ak);
#pragma unroll
                for (int ut = 0; ut < UT; ut++) { if (ks == 0) mfma_v0(acc[ut], ak[ut], bl); else mfma_v(acc[ut], ak[ut], bl); }
                MDD_SCHED_HINT();
#pragma unroll
                for (int ut = 0; ut < UT; ut++) mfma_a(acc[ut], al[ut][ks], bh);
                MDD_SCHED_HINT();
#pragma unroll
                for (int ut = 0; ut < UT; ut++) mfma_v(acc[ut], ak[ut], bh);
                MDD_SCHED_HINT();
            }
            mfma_drain();
            // D layout: row (unit of the tile) = 4 * kq + r, column (batch row) = li
#pragma unroll
            for (int ut = 0; ut < UT; ut++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[((wave * UW) + ut * 16 + 4 * kq + r) * 16 + li] = acc[ut][r];
            lds_barrier();
        }
        // ---- cell backward of this thread's elements; DG_t to the team (tagged chunks) and to memory (fp32)
        const unsigned tg = (unsigned)(s % 3 + 1);
        const bool pub = s + 1 < T;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(hxg + (size_t)((s & 1) * 32 + team) * tgran, 0, PANB, 0x00020000);
#pragma unroll
        for (int i = 0; i < NE; i++) {
            if (!valid[i]) continue;
            const int unit = member * UW + ul[i], ri = ul[i] * 16 + row[i];
            float dh = pdo[i];
            if (s > 0) dh += (red[ri] + red[UW * 16 + ri]) + (red[2 * UW * 16 + ri] + red[3 * UW * 16 + ri]);
            const float gi = pg[i].x, gf = pg[i].y, gg = pg[i].z, go = pg[i].w;
            const float th = tanhf(pc[i]);
            const float dcell = dh * go * (1.f - th * th) + dcar[i];
            const float d_o = dh * th * go * (1.f - go);
            const float d_i = dcell * gg * gi * (1.f - gi);
            const float d_f = dcell * pcp[i] * gf * (1.f - gf);
            const float d_g = dcell * gi * (1.f - gg * gg);
            dcar[i] = dcell * gf;
            *reinterpret_cast<float4 *>(a.dg + (((size_t)t * B + brow[i]) * 2 + d) * G + unit * 4) = make_float4(d_i, d_f, d_g, d_o);
            if (pub) {   // chunk = {hi_i hi_f | hi_g hi_o | lo_i' lo_f' | lo_g' lo_o'}: tag bit 0 rides in the even values' lo, bit 1 in the odd ones'
                const unsigned wi = split_h(d_i), wf = split_h(d_f), wg = split_h(d_g), wo = split_h(d_o);
                u32x4 pv;
                pv[0] = (wi & 0xffffu) | (wf << 16);
                pv[1] = (wg & 0xffffu) | (wo << 16);
                pv[2] = ((wi >> 16) | (tg & 1u)) | (((wf >> 16) | (tg >> 1)) << 16);
                pv[3] = ((wg >> 16) | (tg & 1u)) | (((wo >> 16) | (tg >> 1)) << 16);
                __builtin_amdgcn_raw_buffer_store_b128(pv, drs, (unsigned)(row[i] * ROWB + ((unit ^ row[i]) * 16)), 0, 16 /* sc1 */);
            }
        }
        if (s + 1 < T) prefetch(s + 1);
    }
}

template <int H>
static int launch_bwd_granule_t(const BwdPersistArgs &a, hipStream_t st) {
    constexpr size_t panb = (size_t)16 * 4 * H * 4;
    const size_t smem = panb + (size_t)4 * (H / 8) * 16 * 4;
    if (int rc = launch_zero_fill(a.sync, 32 * sizeof(unsigned int), st)) return rc;
    if (int rc = launch_zero_fill(a.hx, (size_t)2 * 32 * panb, st)) return rc;   // tags must start at 0 on every launch
    hipLaunchKernelGGL((lstm_bwd_granule_kernel<H>), dim3(256), dim3(256), smem, st, a);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
size_t lstm_bwd_granule_hx_bytes(int H) { return (size_t)2 * 32 * 16 * 4 * H * 4; }
int launch_lstm_bwd_granule(const float *dout, const float *gates, const float *cst, SplitPtr whhT, float *dg, int T, int B, int H, unsigned short *hx,
                            unsigned int *sync, int *err_flag, hipStream_t st) {
    BwdPersistArgs a;
    a.dout = dout; a.gates = gates; a.cst = cst; a.whhT = whhT; a.dg = dg; a.hx = hx; a.sync = sync; a.err_flag = err_flag;
    a.T = T; a.B = B; a.BGr = (B + 15) / 16;
    if (T <= 0 || B <= 0 || a.BGr > 16) { set_error("persistent lstm backward: B=%d needs more than one tile per team (max 256)", B); return MDD_ERR_ARG; }
    if (H == 384) return launch_bwd_granule_t<384>(a, st);
    if (H == 256) return launch_bwd_granule_t<256>(a, st);
    set_error("persistent lstm backward: unsupported H=%d", H);
    return MDD_ERR_ARG;
}

int init_granule_attributes() {
#define GATTR(H, N, R) MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_granule_kernel<H, N, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
    GATTR(384, 1, 3); GATTR(384, 2, 3); GATTR(384, 3, 3); GATTR(384, 4, 3); GATTR(256, 1, 2); GATTR(256, 2, 2); GATTR(256, 3, 2); GATTR(256, 4, 2);
#undef GATTR
#define GATTR(H, N, R) MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_granule_kernel<H, N, R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
    GATTR(384, 1, 3); GATTR(384, 2, 3); GATTR(256, 1, 2); GATTR(256, 2, 2);
#undef GATTR
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_bwd_granule_kernel<384>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_bwd_granule_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    return MDD_OK;
}

// The persistent layer kernels are written for a grid of exactly kPersistGrid workgroups (2 directions x 16 batch
// groups x 8 team members), ALL resident at once, one per CU.  persistent_grid_fits() asks the runtime whether a device
// with n_cu compute units can hold that grid (>= 1 workgroup of the largest configuration per CU, and enough CUs); when it
// cannot, the library uses the per-step kernels instead of risking a grid that waits for workgroups that never start.
int persistent_grid_fits(int n_cu) {
    if (n_cu < kPersistGrid) return 0;
    int per_cu = 0;
    const size_t smem384 = (size_t)16 * (384 / 8) * 12 + (size_t)4 * 2 * 16 * (384 / 8) * 16 + (size_t)2 * 16 * 384 * 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)lstm_layer_granule_kernel<384, 4, 3>, 256, smem384) != hipSuccess) return 0;
    if (per_cu < 1) return 0;
    const size_t smem256 = (size_t)16 * (256 / 8) * 12 + (size_t)4 * 2 * 16 * (256 / 8) * 16 + (size_t)2 * 16 * 256 * 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)lstm_layer_granule_kernel<256, 4, 2>, 256, smem256) != hipSuccess) return 0;
    return per_cu >= 1 ? 1 : 0;
}

int launch_lstm_layer_train(const LstmStepArgs &a, hipStream_t st) {
    if (a.H % 4 != 0 || a.T <= 0 || a.B <= 0 || a.packed || a.hsplit) { set_error("lstm (train): bad arguments T=%d B=%d H=%d", a.T, a.B, a.H); return MDD_ERR_ARG; }
    dim3 grid(a.H / 4, 2, (a.B + 63) / 64), block(256);
    for (int s = 0; s < a.T; s++) {
        if (a.H == 384) hipLaunchKernelGGL(lstm_step_kernel<24>, grid, block, 0, st, a, s);
        else if (a.H == 256) hipLaunchKernelGGL(lstm_step_kernel<16>, grid, block, 0, st, a, s);
        else if (a.H % 16 == 0) hipLaunchKernelGGL(lstm_step_kernel<1>, grid, block, 0, st, a, s);
        else hipLaunchKernelGGL(lstm_step_kernel<0>, grid, block, 0, st, a, s);
    }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int launch_lstm_layer(const LstmStepArgs &a, hipStream_t st) {
    if (a.H % 4 != 0 || a.T <= 0 || a.B <= 0) { set_error("lstm: bad shape T=%d B=%d H=%d", a.T, a.B, a.H); return MDD_ERR_ARG; }
    if (a.hsplit) {
        if (a.H == 384) launch_x3_steps<2, 384>(a, st);
        else if (a.H == 256) launch_x3_steps<2, 256>(a, st);
        else { set_error("lstm: split-bf16 step built for H in {256,384}"); return MDD_ERR_ARG; }
        MDD_LAUNCH_CHECK();
        return MDD_OK;
    }
    dim3 grid(a.H / 4, 2, (a.B + 63) / 64), block(256);
    for (int s = 0; s < a.T; s++) {
        if (a.packed && a.H == 384) hipLaunchKernelGGL(lstm_step_packed_kernel<24>, grid, block, 0, st, a, s);
        else if (a.packed && a.H == 256) hipLaunchKernelGGL(lstm_step_packed_kernel<16>, grid, block, 0, st, a, s);
        else if (a.packed) { set_error("lstm: packed layout built for H in {256,384}"); return MDD_ERR_ARG; }
        else hipLaunchKernelGGL(lstm_step_kernel<0>, grid, block, 0, st, a, s);
    }
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

}  // namespace mdd
