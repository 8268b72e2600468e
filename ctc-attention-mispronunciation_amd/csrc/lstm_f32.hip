// Persistent BiLSTM layer kernel, reference-width arithmetic: the whole recurrence of one bidirectional layer in ONE
// launch with every product an exact fp32 MFMA (v_mfma_f32_16x16x4_f32), as torch.nn.LSTM computes it in the reference
// (AA/models/model_ctc.py:27-29 builds the nn.LSTM, :36-49 BatchRNN.forward runs it; text encoder :150,198).
//
// It repeats lstm_step_packed_kernel (lstm.hip) BIT FOR BIT -- the same two accumulators per gate tile (even / odd float4
// components of the packed W_hh'), the same k-groups per MFMA ({16j + 4q + m : q = 0..3}) in the same order (j ascending, m
// ascending), the same (a0 + a1) + gx combine, the same gate functions (lstm_persist.h) -- but the per-step launches re-read W_hh (4.7 MB) from
// L2 on every one of the 250 steps and pay a kernel boundary each; here a workgroup keeps its gate rows of W_hh' in
// registers for the whole layer (288 registers per lane at H = 384, exactly what the split-bf16 kernel spends on its hi + lo
// planes) and only h moves.
//
// Grid and teams are those of lstm_layer_granule_kernel: 256 workgroups = 2 directions x 16 batch groups x 8 members; a
// member owns 4H/8 gate rows (RTW 16-row MFMA tiles per wave); a team's batch rows are NBT tiles of 16, independent
// recurrences advanced in turn.  The hand-off is the same data-tagged exchange ("the data IS the flag"): h travels as
// 16-byte chunks of FOUR consecutive units of one batch row, chunk index (unit / 4) * 16 + row, written write-through by the
// lane-group that produced them and swept by the consumers with a linear LDS-DMA copy of the 24 KB panel.  The chunk is the
// MFMA B operand as it lies: lane (row li, k-quarter q) reads chunk 4j + q and owns k = 16j + 4q + m for its four words m.
// The fp32 payload has no spare mantissa bit, but |h| = |o * tanh(c)| <= 1 means bit 30 of every word (the top exponent
// bit, set only for |x| >= 2, Inf and NaN) is always zero: it carries the epoch tag (step % 3 + 1; bit 0 of the tag in the
// first word, bit 1 in the second; a chunk is one lane's single 16-byte store, old or new as a whole).  A NaN state (the only non-finite value h can take) is published as 1.5 -- a magnitude no
// real h has, so tags stay valid and nothing stalls; the poisoned unit's own outputs and cell state stay NaN, which is what
// reaches the next layer.
// An fp32 MFMA executes on the SIMD's fp32 lanes (that is why its rate equals the vector rate): measured, vector instructions do
// NOT hide behind it as they do behind a bf16 MFMA -- the first form of this kernel, which cleared the tag bits inside the
// product loop (4 VALU per 3 MFMAs) and let the compiler copy AGPR-resident weights to VGPRs (v_accvgpr_read), ran its
// MFMAs at 50 cycles apiece instead of 32 (profiles/round3_lstm_f32_stamps.txt).  Hence: the product loop contains nothing but
// MFMAs, the operand reads (ds_read_b128 of a chunk = four k-groups' worth of B) and the LDS-DMA requests; weights living in the
// accumulation file feed the MFMA from there (inline asm, as in lstm.hip); the tag bits are cleared once per panel, in LDS, by
// the thread that fetched the chunk, in the same pass that checks them; the gate nonlinearities are the short forms of
// lstm_persist.h.
// A tile phase is 96 * RTW MFMAs = 3.8 us at H = 384 against ~3 us of hand-off latency: with two or more tiles the next tile's
// panel is requested late in this tile's MFMA loop, waited for with a plain vmcnt(0) at the loop's end (nothing slow is older:
// the previous phase's stores were acknowledged microseconds ago), and checked + cleaned by the wave that fetched each piece
// BEFORE the cell update; a stale piece (a member's h had not landed) is simply requested again after the publish.  Nothing is
// consumed unchecked, so the product loop has no redo path.
#include "lstm_persist.h"

namespace mdd {

#ifndef MDD_F32_RQ_TAIL
#define MDD_F32_RQ_TAIL 4   // k-groups left in the product loop behind the last piece of the next tile's sweep request (H = 384, two tiles: the
#endif                      // request starts ~2.3 us after that tile's publish and has ~1.6 us to land before the loop ends)

// diagnostic phase stamps only exist in the DBG instantiations (MDD_LSTM_DBG): the production kernels carry none of their branches
#define FSTAMP(i) do { if (DBG) { long long n_ = __builtin_readcyclecounter(); ph[i] += n_ - tst; tst = n_; } } while (0)

template <int H, int NBT, bool DBG = false>
__global__ __launch_bounds__(256, 1) void lstm_layer_f32_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTH = 256, J = H / 16, RM = 4 * H / 8, RTW = RM / 64, UW = RM / 4, NUT = H / 4;
    static_assert(RM == 4 * RTW * 16 && UW % 16 == 0, "4 waves x RTW row tiles cover the workgroup's gate rows");
    float *Os = reinterpret_cast<float *>(smem);                     // [16 rows][UW] layer output (next layer's BatchNorm applied when given)
    float *Or = Os + 16 * UW;                                       // [16 rows][UW] raw h, when both leave
    unsigned int *Og = reinterpret_cast<unsigned int *>(Or + 16 * UW);   // [UW/4 chunk columns][16 rows][4] tagged words: the publish order
    float *Gx = reinterpret_cast<float *>(Og + 16 * UW);             // [NBT][2 step parities][16 rows x UW units x 4 gates]
    constexpr int GXT = 16 * UW * 4;
    constexpr int PANB = 16 * H * 4;                                 // bytes of one tile panel
    unsigned char *Rw = reinterpret_cast<unsigned char *>(Gx + NBT * 2 * GXT);   // [2][PANB] panels as they travel
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int w = blockIdx.x, xl = w & 7, jw = w >> 3;
    const int team = xl * 4 + (jw >> 3), member = jw & 7;           // 8 members share blockIdx % 8 (one XCD under round-robin; speed only)
    const int d = team >> 4, g = team & 15;
    const int B = a.B, T = a.T;
    if (tid == 0) s_fail = 0;

    // ---- resident weights: the packed consumer order of lstm_step_packed_kernel, Wp[d][row tile][j][lane][m].  The first JA k-groups
    // live in the accumulation file and feed the MFMA from there (asm); the last J - JA are ordinary registers under builtin MFMAs,
    // so the compiler sees the loop's final writers of every accumulator and places the MFMA -> VALU wait states itself.
    constexpr int JA = J * 2 / 3;
    float4 w4[RTW][J];
    float osc[RTW], osh[RTW];
    const bool scaled = a.oscale != nullptr;
#pragma unroll
    for (int rt = 0; rt < RTW; rt++) {
        const int ut = member * (RM / 16) + wave * RTW + rt;
        const int unit = ut * 4 + kq;
        osc[rt] = scaled ? a.oscale[d * H + unit] : 1.f;
        osh[rt] = scaled ? a.oshift[d * H + unit] : 0.f;
        const float4 *wp = reinterpret_cast<const float4 *>(a.whh_f32) + ((size_t)(d * NUT + ut) * J) * 64 + lane;
#pragma unroll
        for (int j = 0; j < J; j++) w4[rt][j] = wp[j * 64];
    }
    float cst[RTW][NBT];
#pragma unroll
    for (int rt = 0; rt < RTW; rt++)
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) cst[rt][bt] = 0.f;
    int slen[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) {
        const int lb = bt * 16 + li, row = g * a.BGr + lb;
        slen[bt] = (a.seqlen && lb < a.BGr && row < B) ? a.seqlen[row] : T;
    }
    constexpr size_t tgran = (size_t)16 * H / 2;                    // 8-byte granules per tile panel
    const size_t pgran = NBT * tgran;                               // granules per (parity, team)
    constexpr int NLD = 16 * H / 4 / NTH;                           // 16-byte chunks per thread and panel (= LDS-DMA pieces per wave)
    static_assert(16 * H / 4 % NTH == 0, "a panel is whole passes of the workgroup");
    constexpr int RQJ = J - NLD - MDD_F32_RQ_TAIL;                   // first k-group of the in-loop sweep request: ALL NLD pieces are issued inside the loop
    static_assert(RQJ >= RTW && RQJ + NLD <= J, "the sweep pieces follow the gx pieces and fit the loop");
    u64 *hxg = reinterpret_cast<u64 *>(a.hx);
    unsigned int *abortf = a.sync + 16;
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tst = DBG ? (long long)__builtin_readcyclecounter() : 0;

    // gate pre-activations of (tile bt, time tt): lane (row li, unit kq) of piece rt fetches the 16 bytes (i, f, g, o of its unit and
    // row) it consumes itself in the cell update of row tile rt -- nothing crosses lanes, no barrier between transfer and use
    const unsigned gvoff = (unsigned)(d * 4 * H + (member * UW + wave * RTW * 4 + kq) * 4) * 4u;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)wave * 1024u);
    const unsigned wave_gx = __builtin_amdgcn_readfirstlane((unsigned)wave * (unsigned)(RTW * 1024));
    const unsigned gx_lds = (unsigned)(unsigned long long)(lds_void_t *)Gx, rw_lds = (unsigned)(unsigned long long)(lds_void_t *)Rw;
    auto load_gx = [&](int bt, int par, int tt, int i0 = 0, int i1 = 99) {
        i1 = i1 > RTW ? RTW : i1;
        const float *gbase = a.gx + (size_t)tt * B * 2 * 4 * H;
        const int b = min(g * a.BGr + min(bt * 16 + li, a.BGr - 1), B - 1);           // rows past the batch read a valid row (never used)
        const unsigned rowoff = (unsigned)b * (unsigned)(2 * 4 * H * 4) + gvoff;
#pragma unroll
        for (int i = i0; i < i1; i++)
            lds_dma16_s<false>(gbase, rowoff + (unsigned)(i * 64), gx_lds + (unsigned)(((bt * 2 + par) * GXT + i * 256) * 4) + wave_gx);
    };
    // sweep of (tile bt, state of step sp) into panel buffer pb: NLD pieces per wave, piece i = bytes [i * 4 KB + tid * 16, +16)
    auto request_sweep = [&](int bt, int sp, int pb, int i0 = 0, int i1 = 99) {
        i1 = i1 > NLD ? NLD : i1;
        const unsigned char *srcp = reinterpret_cast<const unsigned char *>(hxg + (size_t)((sp & 1) * 32 + team) * pgran + bt * tgran);
#pragma unroll
        for (int i = i0; i < i1; i++) lds_dma16_s<true>(srcp + (size_t)i * NTH * 16, (unsigned)tid * 16u, rw_lds + (unsigned)(pb * PANB + i * NTH * 16) + wave_lds);
    };
    // This thread's own pieces of panel buffer pb against the tag of step sp (wave-uniform verdict); fresh pieces are written back
    // with the tag bits cleared, i.e. as the plain fp32 operands the product loop reads
    auto pieces_stale = [&](int pb, int sp) -> bool {
        const unsigned ep = (unsigned)(sp % 3 + 1), e0 = (ep & 1u) << 30, e1 = (ep >> 1) << 30;
        unsigned bad = 0;
        u32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            v[i] = *reinterpret_cast<const u32x4 *>(Rw + (size_t)pb * PANB + (size_t)(i * NTH + tid) * 16);
            bad |= (v[i][0] ^ e0) | (v[i][1] ^ e1);
        }
        if (__any((bad & 0x40000000u) != 0)) return true;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            v[i][0] &= 0xbfffffffu; v[i][1] &= 0xbfffffffu;
            *reinterpret_cast<u32x4 *>(Rw + (size_t)pb * PANB + (size_t)(i * NTH + tid) * 16) = v[i];
        }
        return false;
    };
    // layer outputs: every wave stores the units it produced (4 * RTW consecutive units of all 16 rows) as 16-byte pieces read back
    // from the LDS tiles; rows past the batch repeat the tile's last valid row (identical bytes to the same address)
    constexpr int UWW = UW / 4, PCW = UWW / 4, PWR = 16 * PCW, PWP = 16 * RTW;
    static_assert(UWW % 4 == 0 && PWR <= 64 && PWP <= 64, "a wave's output pieces fit one instruction");
    const int orow = lane / PCW, ocol = wave * UWW + (lane - orow * PCW) * 4;
    const int qp = min(wave * PWP + lane, 16 * (UW / 4) - 1);
    const unsigned offp = lane < PWP ? (unsigned)(member * 16 * (UW / 4) + wave * PWP + lane) * 16u : 0xffffffffu;
    const size_t slab = (size_t)B * 2 * H;
    const bool two_out = a.out_raw && a.out && a.out_raw != a.out;
    float *const out_main = a.out ? a.out : a.out_raw;             // what the Os tile goes to
    auto tile_rows = [&](int bt) { const int nv = min(a.BGr, B - g * a.BGr) - bt * 16; return nv < 0 ? 0 : (nv > 16 ? 16 : nv); };
    auto store_out = [&](int bt, int tt) {
        const int nr = tile_rows(bt);
        if (nr == 0) return;
        const int r = min(orow, nr - 1);
        const unsigned el = lane < PWR ? (unsigned)((g * a.BGr + bt * 16 + r) * 2 * H + d * H + member * UW + ocol) : 0x3fffffffu;
        if (out_main) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Os + r * UW + ocol);
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(out_main + (size_t)tt * slab, 0, (int)(slab * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_, el * 4u, 0, 0);
        }
        if (two_out) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Or + r * UW + ocol);
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(a.out_raw + (size_t)tt * slab, 0, (int)(slab * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_, el * 4u, 0, 0);
        }
    };

    // Step 0 has no state to multiply: its phases run the product loop on an all-zero panel (exact zeros out), which keeps the loop
    // free of a first-step branch -- a second definition of the 48 accumulator registers made the compiler route them through the
    // accumulation file on every phase (96 copies).  Both panel buffers start as zeros; nothing writes them before step 0 reads them.
#pragma unroll
    for (int i = 0; i < 2 * NLD; i++) *reinterpret_cast<u32x4 *>(Rw + (size_t)(i * NTH + tid) * 16) = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) load_gx(bt, 0, d ? (T - 1) : 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), compiler-visible: the weight fragments are in registers from here on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    int pc = 0;                                                     // phases so far: this phase's panel buffer = pc & 1
    for (int s = 0; s < T; s++) {
        const int t = d ? (T - 1 - s) : s;
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) {
            const int nbt = bt + 1 < NBT ? bt + 1 : 0, ns = bt + 1 < NBT ? s : s + 1;     // the next phase: tile nbt, step ns (needs the state of step ns - 1)
            const int pb = pc & 1;
            ++pc;
            const bool next_needs = ns >= 1 && ns < T;                // the next phase has products
            const int tt_next = d ? max(T - 2 - s, 0) : min(s + 1, T - 1);
            f32x4 acc[RTW][2];
            const bool in_loop = NBT >= 2 && next_needs;            // the next phase's sweep is requested inside this phase's product loop
            {
                // ---- products: the panel of (bt, s - 1) is complete in Rw[pb] (barrier at the end of the previous phase; zeros at step 0)
                const unsigned char *fb = Rw + (size_t)pb * PANB + kq * 256 + li * 16;
                constexpr int PD = 3;
                f32x4 hr[PD];
#pragma unroll
                for (int p = 0; p < PD; p++) hr[p] = *reinterpret_cast<const f32x4 *>(fb + p * 1024);
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const f32x4 hv = hr[j % PD];
                    if (j + PD < J) hr[j % PD] = *reinterpret_cast<const f32x4 *>(fb + (j + PD) * 1024);
                    // LDS-DMA pieces between the MFMAs: this tile's next gx slab first, the next tile's panel late in the loop
                    if (j < RTW) load_gx(bt, (s + 1) & 1, tt_next, j, j + 1);
                    if (NBT >= 2 && j >= RQJ && j < RQJ + NLD) { if (in_loop) request_sweep(nbt, ns - 1, pb ^ 1, j - RQJ, j - RQJ + 1); }
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int rt = 0; rt < RTW; rt++) {
                            const float wv = m == 0 ? w4[rt][j].x : m == 1 ? w4[rt][j].y : m == 2 ? w4[rt][j].z : w4[rt][j].w;
                            if (j == 0 && m < 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[rt][m & 1]) : "a"(wv), "v"(hv[m]));
                            else if (j < JA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[rt][m & 1]) : "a"(wv), "v"(hv[m]));
                            else acc[rt][m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, hv[m], acc[rt][m & 1], 0, 0, 0);
                        }
                }
            }
            FSTAMP(1);
            // everything requested in the loop has landed (older stores were acknowledged long ago): this tile's gx slab of step s
            // (requested a step ago) in particular
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool stale = in_loop && pieces_stale(pb ^ 1, ns - 1);
            if (stale) request_sweep(nbt, ns - 1, pb ^ 1);
            FSTAMP(2);
            // ---- cell update (lstm_step_packed_kernel's arithmetic); h_s goes to the LDS tiles
            const unsigned tg = (unsigned)(s % 3 + 1);
            const unsigned tagw = (kq == 0 ? (tg & 1u) : kq == 1 ? (tg >> 1) : 0u) << 30;   // this lane's word of the chunk: tag bit 0 rides in word 0, bit 1 in word 1
            const int lb = bt * 16 + li;
            const bool valid = lb < a.BGr && g * a.BGr + lb < B;
            float4 gv[RTW];
#pragma unroll
            for (int rt = 0; rt < RTW; rt++)
                gv[rt] = *reinterpret_cast<const float4 *>(Gx + (bt * 2 + (s & 1)) * GXT + ((wave * RTW + rt) * 64 + lane) * 4);
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {
                const float gi = (acc[rt][0][0] + acc[rt][1][0]) + gv[rt].x;
                const float gf = (acc[rt][0][1] + acc[rt][1][1]) + gv[rt].y;
                const float gg = (acc[rt][0][2] + acc[rt][1][2]) + gv[rt].z;
                const float go = (acc[rt][0][3] + acc[rt][1][3]) + gv[rt].w;
                const float ig = gate_sigmoid(gi), fg = gate_sigmoid(gf), cg = gate_tanh(gg), og = gate_sigmoid(go);
                const bool live = !(d && t >= slen[bt]);              // the reverse direction starts at the row's own last step, from a zero state
                const float cn = fg * cst[rt][bt] + ig * cg;
                const float hr_ = og * gate_tanh(cn);
                const float hn = (valid && live) ? hr_ : 0.f;
                cst[rt][bt] = live ? cn : 0.f;
                const int cl = wave * RTW + rt, ul = cl * 4 + kq;        // chunk column and unit inside the workgroup's share
                Os[li * UW + ul] = scaled ? hn * osc[rt] + osh[rt] : hn;
                if (two_out) Or[li * UW + ul] = hn;
                const unsigned hw = (hn != hn) ? 0x3fc00000u : __float_as_uint(hn);
                Og[(cl * 16 + li) * 4 + kq] = hw | tagw;
            }
            FSTAMP(3);
            // ---- publish h_s (write-through, no drain, no signal) and the layer outputs; a wave reads back only what it wrote itself
            {
                const u32x4 pv = *reinterpret_cast<const u32x4 *>(Og + qp * 4);
                if (s + 1 < T) {
                    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(hxg + (size_t)((s & 1) * 32 + team) * pgran + bt * tgran, 0, (int)(tgran * 8), 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(pv, drs, offp, 0, 16 /* sc1 */);
                }
                if (next_needs && !in_loop) {
                    // one tile (nothing to overlap), or the first step (no earlier phase to request from): the panel is requested
                    // here and polled below; the outputs leave behind the request.  With one tile it is the panel this very
                    // publish goes into: a request sent before the publish is acknowledged only finds stale tags
                    if (NBT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    request_sweep(nbt, ns - 1, pb ^ 1);
                    stale = true;
                }
                store_out(bt, t);
            }
            FSTAMP(4);
            // ---- the next phase's panel: every wave answers for the pieces it fetched itself.  A plain vmcnt(0) in front of every check
            // (not the split-bf16 kernel's counted vmcnt(K) past the younger stores): the store acknowledgements it also waits for
            // (~1 us) arrive before a panel requested right after the publish can be fresh (~2 us) anyway, and the check then never
            // reads LDS with a transfer of its own still in flight.
            if (stale) {
                long long t0 = 0;
                int polls = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                while (pieces_stale(pb ^ 1, ns - 1)) {
                    if ((++polls & 63) == 0) {
                        int ab = 0;
                        if (t0 == 0) t0 = wall_clock64();
                        if (lane == 0) ab = (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) || (wall_clock64() - t0 > 200000000ll);
                        if (__any(ab)) {
                            if (lane == 0) { __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicExch(a.err_flag, 2); s_fail = 1; }
                            break;
                        }
                    }
                    request_sweep(nbt, ns - 1, pb ^ 1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (DBG) ph[5] += polls;
            }
            lds_barrier();                                              // the next panel is whole; the LDS tiles and this phase's panel buffer are free again
            FSTAMP(0);
            if (s_fail) return;
        }
    }
    if (DBG && tid == 0) for (int i = 0; i < 6; i++) a.dbg[blockIdx.x * 6 + i] = ph[i];
}

template <int H, int NBT, bool DBG = false>
static int launch_f32_t(PersistArgs a, hipStream_t st) {
    const size_t smem = (size_t)16 * (H / 8) * 12 + (size_t)NBT * 2 * 16 * (H / 8) * 16 + (size_t)2 * 16 * H * 4;   // tiles + gx slabs (2 parities) + two panel buffers
    if (int rc = launch_zero_fill(a.sync, 32 * sizeof(unsigned int), st)) return rc;
    if (int rc = launch_zero_fill(a.hx, (size_t)2 * 32 * NBT * 16 * H * 4, st)) return rc;   // tags must start at 0 on every launch (by a kernel: lstm.hip)
    hipLaunchKernelGGL((lstm_layer_f32_kernel<H, NBT, DBG>), dim3(kPersistGrid), dim3(256), smem, st, a);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

int launch_lstm_layer_f32(const LstmStepArgs &s, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st) {
    PersistArgs a;
    a.gx = s.gx; a.whh = {nullptr, nullptr}; a.whh_f32 = s.whh; a.hx = hx; a.sync = sync; a.err_flag = err_flag;
    a.out = s.out; a.out_raw = s.out_raw; a.out_split = {nullptr, nullptr}; a.oscale = s.oscale; a.oshift = s.oscale ? s.oshift : nullptr;
    a.T = s.T; a.B = s.B; a.BGr = (s.B + 15) / 16; a.BG = granule_bg(s.B); a.seqlen = s.seqlen;
    a.dbg = (getenv("MDD_LSTM_DBG") && s.T > 100) ? reinterpret_cast<long long *>(reinterpret_cast<u64 *>(hx) + (size_t)2 * 32 * a.BG * s.H) : nullptr;
    a.early = 0; a.gates_save = nullptr; a.c_save = nullptr;
    if (!s.packed || s.out_split.hi || s.gates_save) { set_error("persistent fp32 lstm: packed W_hh layout, fp32 outputs, inference only"); return MDD_ERR_ARG; }
    if (!a.out && !a.out_raw) { set_error("persistent fp32 lstm: no output"); return MDD_ERR_ARG; }
    const int nbt = a.BG / 16;
    if (nbt < 1 || nbt > 4) { set_error("persistent fp32 lstm: B=%d needs %d row tiles per team (max 4)", s.B, nbt); return MDD_ERR_ARG; }
    if (a.dbg && s.H == 384) return nbt == 1 ? launch_f32_t<384, 1, true>(a, st) : nbt == 2 ? launch_f32_t<384, 2, true>(a, st) : nbt == 3 ? launch_f32_t<384, 3, true>(a, st) : launch_f32_t<384, 4, true>(a, st);
    a.dbg = nullptr;
    if (s.H == 384) return nbt == 1 ? launch_f32_t<384, 1>(a, st) : nbt == 2 ? launch_f32_t<384, 2>(a, st) : nbt == 3 ? launch_f32_t<384, 3>(a, st) : launch_f32_t<384, 4>(a, st);
    if (s.H == 256) return nbt == 1 ? launch_f32_t<256, 1>(a, st) : nbt == 2 ? launch_f32_t<256, 2>(a, st) : nbt == 3 ? launch_f32_t<256, 3>(a, st) : launch_f32_t<256, 4>(a, st);
    set_error("persistent fp32 lstm: unsupported H=%d", s.H);
    return MDD_ERR_ARG;
}

int init_lstm_f32_attributes() {
#define FATTR(H, N) MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_f32_kernel<H, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
    FATTR(384, 1); FATTR(384, 2); FATTR(384, 3); FATTR(384, 4); FATTR(256, 1); FATTR(256, 2); FATTR(256, 3); FATTR(256, 4);
#undef FATTR
#define FATTR(H, N) MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_f32_kernel<H, N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
    FATTR(384, 1); FATTR(384, 2); FATTR(384, 3); FATTR(384, 4);
#undef FATTR
    return MDD_OK;
}

int persistent_f32_grid_fits(int n_cu) {
    if (n_cu < kPersistGrid) return 0;
    int per_cu = 0;
    const size_t smem384 = (size_t)16 * (384 / 8) * 12 + (size_t)4 * 2 * 16 * (384 / 8) * 16 + (size_t)2 * 16 * 384 * 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)lstm_layer_f32_kernel<384, 4>, 256, smem384) != hipSuccess) return 0;
    return per_cu >= 1 ? 1 : 0;
}

__global__ void diag_gates_kernel(const float *x, float *sg, float *th, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { sg[i] = gate_sigmoid(x[i]); th[i] = gate_tanh(x[i]); }
}

}  // namespace mdd

extern "C" int mdd_diag_gates(const float *x_dev, float *sig_dev, float *tanh_dev, int64_t n, void *stream) {
    if (!x_dev || !sig_dev || !tanh_dev || n <= 0) { mdd::set_error("mdd_diag_gates: bad arguments"); return MDD_ERR_ARG; }
    hipLaunchKernelGGL(mdd::diag_gates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, sig_dev, tanh_dev, (long long)n);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
