// Persistent BiLSTM layer kernel, fp32-grade recurrence on the bf16 matrix cores ("f32x6", the default mode's recurrence).
//
// Reference: torch.nn.LSTM as BatchRNN.forward runs it (AA/models/model_ctc.py:27-29,36-49; text encoder :150,198), fp32.
// The exact-fp32 layer kernel (lstm_f32.hip) spends 73 % of its time in v_mfma_f32_16x16x4_f32, which executes on the SIMD's fp32
// lanes at the vector rate.  Here the recurrent product G^T = W_hh' . h^T runs as in gemm_bf16x6.hip: W_hh' and h each as THREE bf16
// planes (hi + mid + lo = the fp32 value exactly), the six cross products down to 2^-24 of a product on v_mfma_f32_16x16x32_bf16 --
// 6/16 of the fp32 MFMA's time -- with hi.hi in an accumulator of its own; cell state, gate pre-activations, the gate functions and
// the layer outputs stay fp32 (the cell update is that of lstm_f32.hip, same functions).
//
// Three planes of W_hh' are 432 registers per lane in an 8-workgroup team, so teams have SIXTEEN members: 256 workgroups = 2 directions
// x 8 batch groups x 16 members, a member owning 4H/16 gate rows = 6 MFMA row tiles at H = 384 (4 at H = 256).  Its four waves split
// the product 2 x 2: three row tiles x half of K each (216 registers of weight fragments, half of the panel to read); the two partial
// sums of a row tile meet at the wave that owns its cell update (see the kernel).  A team's batch rows are NBT tiles of 16 (B = 512:
// four), independent recurrences advanced in turn, so that one tile's h travels while the others' products run; from three tiles up
// the schedule is SKEWED: the cell update, publish and output stores of a tile run inside the next tile's product loop.
// Hand-off as in lstm_f32.hip (data-tagged, no counter, no drain): h travels as 16-byte granules = eight consecutive units of one batch
// row of ONE plane; |h| <= 1 leaves bit 14 of every bf16 element (the top exponent bit) free in all three planes, the epoch tag
// (step % 3 + 1) rides there in the granule's first two elements and is cleared on the MFMA operand registers (one v_and per fragment).
// A panel (one tile's state) is [3 planes][H/8 chunk columns][16 rows] x 16 B = 36 KB at H = 384: the MFMA B operand of lane (row li,
// k-slice q) at k-step ks is chunk column 4 ks + q of each plane, as it lies.  A NaN state travels as 1.5.
// One wave per SIMD issues one instruction every four cycles at best and stalls on every dependency, so the kernel is written for
// few instructions per phase: one assembly block per sweep (an immediate offset moves the global and the LDS address together), one
// 16-byte publish store per owner, one barrier per phase, counted waits that skip the write-through stores' acknowledgements.
#include "lstm_persist.h"
#include <type_traits>
#include <utility>

namespace mdd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define XSTAMP(i) do { if (DBG) { long long n_ = __builtin_readcyclecounter(); ph[i] += n_ - tst; tst = n_; } } while (0)

template <int N, typename F>
__device__ __forceinline__ void x6_static_for(F &&f) {
    if constexpr (N > 0) { x6_static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

struct X6Args {
    const float *gx;                 // [T][B][2][4H] permuted gate columns
    const unsigned short *whh3;      // Whh' as three row-major planes [3][2][4H][H] (hi | mid | lo)
    unsigned short *hx;              // [2 parity][16 teams][NBT tiles][3 planes][H/8][16] x 16 B
    unsigned int *sync;              // [16] unused, [16] abort flag (zeroed before every launch)
    int *err_flag;
    float *out, *out_raw;
    const float *oscale, *oshift;
    int T, B, BGr;                   // BGr: real rows per batch group (8 groups)
    int force_mask;                  // diagnostic (MDD_X6_FORCE_REDO=n, a power of two): every n-th phase is declared stale, so that the refetch-and-multiply-again branch runs; -1: off
    long long *dbg;
    const int *seqlen;
};

template <int H, int NBT, bool DBG = false>
__global__ __launch_bounds__(256, 1) void lstm_layer_x6_kernel(X6Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTH = 256, KS = H / 32, KH = KS / 2, RM = 4 * H / 16, NRT = RM / 16, UW = RM / 4;
    static_assert(NRT == 6 || NRT == 4, "H = 384 (six row tiles per member) or H = 256 (four)");
    constexpr int NT = NRT / 2;                                      // row tiles a wave multiplies (over half of K)
    constexpr int CPM = UW / 8;                                      // 16-byte chunk columns (8 units) a member produces
    constexpr bool SKEW = NBT >= 3;                                  // the cell update of a phase runs inside the next phase's product loop
    constexpr int CC = H / 8;                                        // chunk columns per plane
    constexpr int PLB = CC * 256;                                    // bytes of one plane of a panel
    constexpr int PANB = 3 * PLB;                                    // bytes of a panel
    constexpr int NLD = PANB / (NTH * 16);                           // 16-byte chunks per thread and panel (9 at H = 384, 6 at H = 256)
    static_assert(PANB % (NTH * 16) == 0 && (NLD == 9 || NLD == 6), "a panel is whole passes of the workgroup");
    constexpr int NGB = NBT == 1 ? 2 : 1;                            // gx slab buffers per tile (one tile: the next step's slab is needed right away)
    constexpr int GXT = 16 * UW * 4;                                 // floats per gx slab
    float *Os = reinterpret_cast<float *>(smem);                     // [16 rows][UW] layer output
    float *Or = Os + 16 * UW;                                       // [16 rows][UW] raw h, when both leave
    unsigned short *Og = reinterpret_cast<unsigned short *>(Or + 16 * UW);   // [3 owner waves][3 planes][16 rows][8] tagged bf16: the publish order
    float *Gx = reinterpret_cast<float *>(Og + 3 * 3 * 16 * 8);      // [NBT][NGB][GXT]
    unsigned char *Rw = reinterpret_cast<unsigned char *>(Gx + NBT * NGB * GXT);   // [2][PANB]
    float *Px = reinterpret_cast<float *>(Rw + 2 * PANB);              // [2 phases][6 tiles][2 K halves][64 lanes][4]: partial sums on their way to the tile's owner
    __shared__ int s_flag[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int w = blockIdx.x, xl = w & 7, jw = w >> 3;               // 32 workgroups share blockIdx % 8 (one XCD under round-robin; speed only)
    const int team = xl * 2 + (jw >> 4), member = jw & 15;
    const int d = team >> 3, g = team & 7;
    const int B = a.B, T = a.T;
    if (tid < 3) s_flag[tid] = 0;
    // Work of a member's four waves.  Products: wave (rh, kh) = (wave / 2, wave % 2) multiplies the row tiles {2 rh, 2 rh + 1, 4 + rh}
    // (H = 256: {2 rh, 2 rh + 1}) over the k-steps of K half kh: three tiles x half of K each, and only half of the panel to read.
    // Cell updates: a row tile's two partial sums meet at its OWNER -- wave (rh, 0) owns tiles 2 rh and 2 rh + 1 (its own partial stays in
    // registers, the other arrives through LDS), wave (0, 1) owns tiles 4 and 5 (H = 384).  Owners hold two ADJACENT tiles = eight
    // consecutive units, one 16-byte chunk of each plane per batch row: the granule h travels in.  Wave (1, 1) owns nothing and is the
    // one that compares the panels' tags.
    const int rh = wave >> 1, kh = wave & 1;

    // ---- resident weights: A fragments W'[row = member * RM + tile * 16 + li][k = (kh * KH + j) * 32 + kq * 8 .. +8], three planes
    bf16x8 wf[NT][KH][3];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int rt = t < 2 ? rh * 2 + t : 4 + rh;
        const int row = member * RM + rt * 16 + li;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const unsigned short *wp = a.whh3 + ((size_t)(p * 2 + d) * 4 * H + row) * H + kq * 8 + kh * KH * 32;
#pragma unroll
            for (int j = 0; j < KH; j++) wf[t][j][p] = *reinterpret_cast<const bf16x8 *>(wp + j * 32);
        }
    }
    const int ot0 = kh == 0 ? rh * 2 : 4;                            // first of the two tiles this wave owns (if it owns any)
    const int widx = kh == 0 ? rh : 2;                               // which of the three owners
    float osc[2], osh[2];
    const bool scaled = a.oscale != nullptr;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int unit = member * UW + min(ot0 + r, NRT - 1) * 4 + kq;
        osc[r] = scaled ? a.oscale[d * H + unit] : 1.f;
        osh[r] = scaled ? a.oshift[d * H + unit] : 0.f;
    }
    float cst[2][NBT];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) cst[r][bt] = 0.f;
    int slen[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) {
        const int lb = bt * 16 + li, row = g * a.BGr + lb;
        slen[bt] = (a.seqlen && lb < a.BGr && row < B) ? a.seqlen[row] : T;
    }
    constexpr size_t tbytes = PANB;                                  // bytes per (parity, team, tile)
    const size_t pbytes = (size_t)NBT * tbytes;                      // bytes per (parity, team)
    unsigned char *hxb = reinterpret_cast<unsigned char *>(a.hx);
    unsigned int *abortf = a.sync + 16;
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tst = DBG ? (long long)__builtin_readcyclecounter() : 0;

    // gate pre-activations of (tile bt, time tt): lane (row li, unit kq) of piece r fetches the 16 bytes (i, f, g, o of its unit and row)
    // it consumes itself in the cell update of the r-th row tile its wave owns
    const unsigned gx_lds = (unsigned)(unsigned long long)(lds_void_t *)Gx, rw_lds = (unsigned)(unsigned long long)(lds_void_t *)Rw;
    const int wv = __builtin_amdgcn_readfirstlane(wave);            // the wave's index as a scalar
    auto gx_at = [&](int tt) { return a.gx + (size_t)tt * B * 2 * 4 * H; };
    auto load_gx = [&](int bt, int buf, const float *gbase) {
        const int b = min(g * a.BGr + min(bt * 16 + li, a.BGr - 1), B - 1);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int rt = ot0 + r;
            const unsigned rowoff = (unsigned)b * (unsigned)(2 * 4 * H * 4) + (unsigned)(d * 4 * H + (member * UW + rt * 4 + kq) * 4) * 4u;
            const unsigned la = __builtin_amdgcn_readfirstlane(gx_lds + (unsigned)(((bt * NGB + buf) * GXT + rt * 256) * 4));
            lds_dma16_s<false>(gbase, rowoff, la);
        }
    };
    // sweep of one tile's state (hx_par: the team's panels of the step's parity) into panel buffer pb: a linear copy, each wave its own
    // quarter (NLD pieces of 1 KB), written as one block of assembly: the instruction's immediate offset advances the global AND the LDS
    // address, m0 and the base register pair move once per 4 KB -- little more than one instruction per piece.
    constexpr int WQ = PANB / 4;                                     // bytes of a panel a wave copies
    static_assert(WQ == NLD * 1024 && NLD <= 12, "a wave's quarter is NLD whole pieces");
    auto request_sweep = [&](const unsigned char *hx_par, int bt, int pb) {
        // (wave-uniform by construction; said so explicitly: the assembly's operands must be scalar registers whatever the compiler's divergence analysis concludes)
        const unsigned long long b0u = (unsigned long long)(hx_par + (size_t)bt * tbytes + (size_t)wv * WQ);
        const unsigned char *b0 = reinterpret_cast<const unsigned char *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b0u >> 32)) << 32) |
                                                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b0u));   // (the builtin returns int: no sign extension)
        const unsigned char *b1 = b0 + 4096, *b2 = b0 + 8192;
        const unsigned la = __builtin_amdgcn_readfirstlane(rw_lds + (unsigned)(pb * PANB) + (unsigned)wv * (unsigned)WQ), v = (unsigned)lane * 16u;
#define X6_G(B) "s_nop 0\n\tglobal_load_lds_dwordx4 %0, " B " sc1\n\tglobal_load_lds_dwordx4 %0, " B " offset:1024 sc1\n\t"
#define X6_G2(B) "global_load_lds_dwordx4 %0, " B " offset:2048 sc1\n\tglobal_load_lds_dwordx4 %0, " B " offset:3072 sc1\n\t"
        if (NLD == 9)
            asm volatile("s_mov_b32 m0, %4\n\t" X6_G("%1") X6_G2("%1") "s_add_u32 m0, m0, 0x1000\n\t" X6_G("%2") X6_G2("%2") "s_add_u32 m0, m0, 0x1000\n\t"
                         "s_nop 0\n\tglobal_load_lds_dwordx4 %0, %3 sc1"
                         :: "v"(v), "s"(b0), "s"(b1), "s"(b2), "s"(la) : "memory", "m0", "scc");
        else
            asm volatile("s_mov_b32 m0, %4\n\t" X6_G("%1") X6_G2("%1") "s_add_u32 m0, m0, 0x1000\n\t" X6_G("%2")
                         :: "v"(v), "s"(b0), "s"(b1), "s"(b2), "s"(la) : "memory", "m0", "scc");
#undef X6_G
#undef X6_G2
    };
    const unsigned char *const hx_team = hxb + (size_t)team * pbytes;    // parity 0; parity 1 is 16 teams further
    auto hx_parity = [&](int sp) { return hx_team + (size_t)(sp & 1) * 16 * pbytes; };
    auto tag_word = [](int sp) -> unsigned { const unsigned ep = (unsigned)(sp % 3 + 1); return ((ep & 1u) << 14) | ((ep >> 1) << 30); };   // elements 0, 1 of a granule = its first dword
    // does every granule of the panel in Rw[pb] carry the tag of step sp?  (all threads, NLD chunks each; nothing is written)
    auto panel_stale = [&](int pb, int sp) -> bool {
        const unsigned e = tag_word(sp);
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < NLD; i++) bad |= *reinterpret_cast<const unsigned *>(Rw + (size_t)pb * PANB + (size_t)(i * NTH + tid) * 16) ^ e;
        return (bad & 0x40004000u) != 0;
    };
    // workgroup-wide OR of a per-thread verdict: one barrier (three flag words in rotation: the word of use k + 2 is reset by thread 0
    // after barrier k, when every read of its last use (k - 1) is over and no write for its next use (after barrier k + 1) has begun)
    int fk = 0;
    auto wg_any = [&](bool v) -> bool {
        if (__any(v) && lane == 0) s_flag[fk] = 1;
        lds_barrier();
        const bool r = s_flag[fk] != 0;
        const int nx = fk == 2 ? 0 : fk + 1, n2 = nx == 2 ? 0 : nx + 1;
        if (tid == 0) s_flag[n2] = 0;
        fk = nx;
        return r;
    };
    // the same in two halves, for a verdict that is only read in the next phase: post() before the phase's barrier, posted() after it
    int vslot = 0;
    auto post = [&](bool v) { if (__any(v) && lane == 0) s_flag[fk] = 1; };
    auto posted = [&] {
        vslot = fk;
        const int nx = fk == 2 ? 0 : fk + 1, n2 = nx == 2 ? 0 : nx + 1;
        if (tid == 0) s_flag[n2] = 0;
        fk = nx;
    };
    // fetch the panel of (tile bt, step sp) into Rw[pb] until every granule carries the step's tag; false: gave up (abort flag / 200 ms)
    auto ensure_fresh = [&](int bt, int sp, int pb) -> bool {
        long long t0 = 0;
        int polls = 0;
        bool ok = true;
        for (;;) {
            request_sweep(hx_parity(sp), bt, pb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();                                          // everybody's pieces have landed
            if ((++polls & 63) == 0) {
                bool giveup = false;
                if (t0 == 0) t0 = wall_clock64();
                if (tid == 0 && ((__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) || (wall_clock64() - t0 > 200000000ll))) {
                    __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicExch(a.err_flag, 2); giveup = true;
                }
                if (wg_any(giveup)) { ok = false; break; }
            }
            if (!wg_any(panel_stale(pb, sp))) break;
        }
        if (DBG) ph[5] += polls;
        return ok;
    };
    const size_t slab = (size_t)B * 2 * H;
    const bool two_out = a.out_raw && a.out && a.out_raw != a.out;
    float *const out_main = a.out ? a.out : a.out_raw;
    auto tile_rows = [&](int bt) { const int nv = min(a.BGr, B - g * a.BGr) - bt * 16; return nv < 0 ? 0 : (nv > 16 ? 16 : nv); };

    // Products of this wave's NT row tiles against its half of the panel in Rw[pb].  The tag bits (first dword of every chunk) are
    // cleared on the fragments one k-step before their use -- a vector instruction between the MFMAs of the step before, not a stall in
    // front of its own; CHK: the tag words of the whole panel are compared with the tag `e` of the step the panel should hold (this
    // wave's half as its fragments pass, the other half's tag words read for the purpose).  Per tile two chains: hh = Wh.hh and
    // sm = the five smaller products, smallest first; on return g[t] = hh + sm, this wave's partial sum of tile t.
    auto products = [&](auto chk_, int pb, unsigned e, f32x4 *gsum, auto &&at_ks, auto &&pre) -> bool {
        constexpr bool CHK = decltype(chk_)::value;
        const unsigned char *fb = Rw + (size_t)pb * PANB + kq * 256 + li * 16 + kh * (KH * 1024);
        unsigned orall = 0u, andall = ~0u;
        if (CHK) {
            const unsigned char *ob = Rw + (size_t)pb * PANB + kq * 256 + li * 16 + (kh ^ 1) * (KH * 1024);
#pragma unroll
            for (int j = 0; j < KH; j++)
#pragma unroll
                for (int p = 0; p < 3; p++) { const unsigned v = *reinterpret_cast<const unsigned *>(ob + p * PLB + j * 1024); orall |= v; andall &= v; }
        }
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 hh[NT], sm[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) { hh[t] = z; sm[t] = z; }
        // fragments are read two k-steps ahead, and the k-steps are fenced for LDS reads and MFMAs (vector / scalar arithmetic may cross):
        // left alone, the compiler sinks each read to just before its use and waits for it there
        u32x4 fh[3][3];
#pragma unroll
        for (int p = 0; p < 3; p++) { fh[0][p] = *reinterpret_cast<const u32x4 *>(fb + p * PLB); fh[1][p] = *reinterpret_cast<const u32x4 *>(fb + p * PLB + 1024); }
        pre();   // (the skewed schedule reads the previous phase's verdict here: its LDS round trip hides behind the fragment reads just issued)
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (CHK) { orall |= fh[0][p][0]; andall &= fh[0][p][0]; }
            fh[0][p][0] &= 0xbfffbfffu;
        }
#pragma unroll
        for (int j = 0; j < KH; j++) {
            __builtin_amdgcn_sched_barrier(0x406);
            if (j + 2 < KH) {
#pragma unroll
                for (int p = 0; p < 3; p++) fh[(j + 2) % 3][p] = *reinterpret_cast<const u32x4 *>(fb + p * PLB + (j + 2) * 1024);
            }
            at_ks(j);
            if (j + 1 < KH) {
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    if (CHK) { orall |= fh[(j + 1) % 3][p][0]; andall &= fh[(j + 1) % 3][p][0]; }
                    fh[(j + 1) % 3][p][0] &= 0xbfffbfffu;
                }
            }
            const bf16x8 bh = __builtin_bit_cast(bf16x8, fh[j % 3][0]), bm = __builtin_bit_cast(bf16x8, fh[j % 3][1]), bl = __builtin_bit_cast(bf16x8, fh[j % 3][2]);
#define X6_MF(acc, wp, bp) _Pragma("unroll") for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][j][wp], bp, acc[t], 0, 0, 0)
            X6_MF(sm, 1, bm); X6_MF(sm, 0, bl); X6_MF(sm, 2, bh); X6_MF(hh, 0, bh); X6_MF(sm, 0, bm); X6_MF(sm, 1, bh);   // smallest products first
#undef X6_MF
        }
#pragma unroll
        for (int t = 0; t < NT; t++) gsum[t] = hh[t] + sm[t];
        return CHK && (((orall ^ e) | (andall ^ e)) & 0x40004000u) != 0;
    };
    auto px_at = [&](int buf, int tile, int half) -> float * { return Px + ((buf * 6 + tile) * 2 + half) * 256 + lane * 4; };
    // a wave's partial sums leave for the tiles' owners (what it owns itself stays in registers)
    auto send_partials = [&](auto role_, const f32x4 *gsum, int buf) {
        constexpr int ROLE = decltype(role_)::value;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const bool own = ROLE == 0 ? t < 2 : (ROLE == 1 && t == 2);
            if (!own) *reinterpret_cast<f32x4 *>(px_at(buf, t < 2 ? rh * 2 + t : 4 + rh, kh)) = gsum[t];
        }
    };

    // Cell update of (tile bt, step sc) for the two row tiles this wave owns: lstm_f32.hip's arithmetic.  Part 1: the partial sums
    // meet, gates, c, h into the LDS tiles (the layer output, and h as three tagged bf16 planes in publish order).  Part 2: the request
    // for the tile's next gx slab, then publish (one 16-byte granule = eight units of one row of one plane per lane, write-through) and
    // the layer outputs.  `on` false (the skewed schedule's first phase has no previous tile; a product loop run again): every store
    // lands out of bounds = is dropped.
    auto cell1 = [&](auto role_, const f32x4 *gsum, int buf, auto bt_, int sc, bool on) {
        constexpr int ROLE = decltype(role_)::value;
        constexpr int bt = decltype(bt_)::value;
        const int t = d ? (T - 1 - sc) : sc;
        const unsigned tg = (unsigned)(sc % 3 + 1);
        const int lb = bt * 16 + li;
        const bool valid = lb < a.BGr && g * a.BGr + lb < B;
        const int gb = NBT == 1 ? (sc & 1) : 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int rt = ot0 + r;
            f32x4 acc;
            if (ROLE == 0) acc = gsum[r] + *reinterpret_cast<const f32x4 *>(px_at(buf, rt, 1));
            else if (r == 0) acc = gsum[2] + *reinterpret_cast<const f32x4 *>(px_at(buf, 4, 0));
            else acc = *reinterpret_cast<const f32x4 *>(px_at(buf, 5, 0)) + *reinterpret_cast<const f32x4 *>(px_at(buf, 5, 1));
            const float4 gv = *reinterpret_cast<const float4 *>(Gx + (bt * NGB + gb) * GXT + (rt * 64 + lane) * 4);
            const float gi = acc[0] + gv.x, gf = acc[1] + gv.y, gg = acc[2] + gv.z, go = acc[3] + gv.w;
            const float ig = gate_sigmoid(gi), fg = gate_sigmoid(gf), cg = gate_tanh(gg), og = gate_sigmoid(go);
            const bool live = !(d && t >= slen[bt]);
            const float cn = fg * cst[r][bt] + ig * cg;
            const float hr_ = og * gate_tanh(cn);
            const float hn = (valid && live) ? hr_ : 0.f;
            cst[r][bt] = on ? (live ? cn : 0.f) : cst[r][bt];
            const int ul = rt * 4 + kq;                              // unit inside the workgroup's share
            Os[li * UW + ul] = scaled ? hn * osc[r] + osh[r] : hn;
            Or[li * UW + ul] = hn;
            // h -> hi + mid + lo (exact); elements 0, 1 of each plane's granule (the first tile's units 0, 1) carry the tag in bit 14;
            // a NaN state travels as 1.5 + 0 + 0
            const unsigned tagw = r == 0 ? ((kq == 0 ? (tg & 1u) : kq == 1 ? (tg >> 1) : 0u) << 14) : 0u;
            const float hs = (hn != hn) ? 1.5f : hn;
            const __bf16 b0 = (__bf16)hs;
            const float r1 = hs - (float)b0;
            const __bf16 b1 = (__bf16)r1, b2 = (__bf16)(r1 - (float)b1);
            unsigned short *og_ = Og + widx * (3 * 16 * 8) + li * 8 + r * 4 + kq;
            og_[0] = (unsigned short)(__builtin_bit_cast(unsigned short, b0) | tagw);
            og_[16 * 8] = (unsigned short)(__builtin_bit_cast(unsigned short, b1) | tagw);
            og_[2 * 16 * 8] = (unsigned short)(__builtin_bit_cast(unsigned short, b2) | tagw);
        }
    };
    // (the step's addresses come in as pointers worked out once per step: gxn = the gx slab of the tile's next step, hxpub = the team's
    // panels of the step's parity, outp / rawp = the layer outputs' slabs of the step's time index)
    auto cell2 = [&](auto bt_, const float *gxn, unsigned char *hxpub, float *outp, float *rawp, bool on) {
        constexpr int bt = decltype(bt_)::value;
        if (NBT >= 2) load_gx(bt, 0, gxn);   // this tile's next slab, into the buffer part 1 has read (needed a round from now)
        // the stores come last in the phase: the wait at its end, s_waitcnt vmcnt(3), then covers every transfer into LDS (vector memory
        // operations return in order) without waiting for the write-through store's acknowledgement (~1 us)
        {
            const int prow = lane & 15, pp = lane >> 4;              // lane -> (plane, batch row): 48 lanes
            const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(hxpub + (size_t)bt * tbytes, 0, (int)tbytes, 0x00020000);
            const u32x4 pv = *reinterpret_cast<const u32x4 *>(Og + widx * (3 * 16 * 8) + ((pp < 3 ? pp : 0) * 16 + prow) * 8);
            const unsigned off = (pp < 3 && on) ? (unsigned)(pp * PLB + ((member * CPM + (ot0 >> 1)) * 16 + prow) * 16) : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(pv, drs, off, 0, 16 /* sc1 */);
        }
        {   // layer outputs: lane -> (row, which of the two tiles): 32 pieces of 16 B read back from the LDS tiles
            const int nr = tile_rows(bt);
            const int rr = min(lane & 15, max(nr - 1, 0)), piece = lane >> 4;
            const bool mine = piece < 2 && nr > 0 && on;
            const int rt = ot0 + (piece & 1);
            const unsigned el = (unsigned)((g * a.BGr + bt * 16 + rr) * 2 * H + d * H + member * UW + rt * 4);
            const int lo = rr * UW + rt * 4;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Os + lo);
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(slab * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_, mine ? el * 4u : 0xffffffffu, 0, 0);
            const u32x4 v2 = *reinterpret_cast<const u32x4 *>(Or + lo);
            const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(rawp, 0, (int)(slab * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v2, rs2, (mine && two_out) ? el * 4u : 0xffffffffu, 0, 0);
        }
    };

    // One wave's program over all phases.  ROLE 0: waves (0, 0), (1, 0) -- K half 0, owner of its first two tiles; 1: wave (0, 1) -- K half 1,
    // owner of tiles 4, 5 at H = 384 (of nothing at H = 256); 2: wave (1, 1) -- K half 1, owner of nothing, compares the tags.
    auto run = [&](auto role_) {
        constexpr int ROLE = decltype(role_)::value;
        constexpr bool OWNER = ROLE == 0 || (ROLE == 1 && NRT == 6);
        const std::integral_constant<bool, ROLE == 2> chk_{};
        // step 0 multiplies an all-zero panel (exact zeros out): both panel buffers start as zeros
#pragma unroll
        for (int i = 0; i < 2 * NLD; i++) *reinterpret_cast<u32x4 *>(Rw + (size_t)(i * NTH + tid) * 16) = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (OWNER) x6_static_for<NBT>([&](auto bt_) { load_gx(decltype(bt_)::value, 0, gx_at(d ? (T - 1) : 0)); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        f32x4 gP[NT];                                                // SKEW: the sums of the previous phase, waiting for their cell update
#pragma unroll
        for (int t = 0; t < NT; t++) gP[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int pc = 0;
        bool dead = false, have_v = false;
        auto redo = [&](int btv, int sv, int pbv, int pxb) {   // the products of (tile btv, step sv) again, from a panel fetched until it is whole: gP and Px[pxb]
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!ensure_fresh(btv, sv - 1, pbv)) { dead = true; return; }
            products(std::false_type{}, pbv, 0u, gP, [](int) {}, [] {});
            send_partials(role_, gP, pxb);
            lds_barrier();                                              // the partial sums are in place
            if (DBG) ph[5] += 1000;
        };
        float *const raw_main = two_out ? a.out_raw : out_main;
        auto t_of = [&](int st) { return d ? (T - 1 - st) : st; };
        for (int s = 0; s < T && !dead; s++) {
            // the step's addresses (and the previous step's, which the skewed schedule's first tile still works on)
            const int sp_ = max(s - 1, 0), sn_ = min(s + 1, T - 1);
            const float *const gx_s = gx_at(t_of(s)), *const gx_n = gx_at(t_of(sn_));
            float *const out_s = out_main + (size_t)t_of(s) * slab, *const out_p = out_main + (size_t)t_of(sp_) * slab;
            float *const raw_s = raw_main + (size_t)t_of(s) * slab, *const raw_p = raw_main + (size_t)t_of(sp_) * slab;
            unsigned char *const hx_s = const_cast<unsigned char *>(hx_parity(s)), *const hx_p = const_cast<unsigned char *>(hx_parity(s - 1));
            auto phase = [&](auto bt_) {
                constexpr int bt = decltype(bt_)::value;
                constexpr int nbt = bt + 1 < NBT ? bt + 1 : 0, pbt = bt > 0 ? bt - 1 : NBT - 1;
                const int ns = bt + 1 < NBT ? s : s + 1, psv = bt > 0 ? s : s - 1;
                const int pb = pc & 1;
                ++pc;
                f32x4 gs[NT];
                if (NBT == 1) {
                    // one tile per team: the panel is the team's own state of a moment ago -- fetched (again and again) until it is whole
                    if (s >= 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own publish acknowledged: an earlier request would only find stale tags
                        if (!ensure_fresh(bt, s - 1, pb)) { dead = true; return; }
                    }
                    XSTAMP(0);
                    products(std::false_type{}, pb, 0u, gs, [&](int j) { if (OWNER && j == 0) load_gx(bt, (s + 1) & 1, gx_n); }, [] {});
                    XSTAMP(1);
                    send_partials(role_, gs, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gx slab
                    lds_barrier();                                      // the partial sums
                    XSTAMP(2);
                    if constexpr (OWNER) { cell1(role_, gs, 0, bt_, s, true); XSTAMP(3); cell2(bt_, gx_n, hx_s, out_s, raw_s, true); }
                    XSTAMP(4);
                    return;
                }
                // Two or more tiles per team.  SKEW (three or more): the next phase's panel is requested as this phase's product loop begins, and
                // the previous phase's cell update, publish and output stores run inside the loop.  Two tiles: the next panel's state was published
                // at the end of the phase before -- requested when the product loop ends, it travels during the cell update.
                constexpr int RQJ = SKEW ? 0 : KH - 1;
                const bool stale = products(chk_, pb, tag_word(s + 2), gs, [&](int j) {   // (s - 1) % 3 == (s + 2) % 3, s >= 0
                    if (j == RQJ) request_sweep(hx_parity(ns - 1), nbt, pb ^ 1);   // (step -1 = parity 1 before anything was published there: zeros, as step 0 wants them)
                    if constexpr (SKEW && OWNER) {
                        if (j == 0) cell1(role_, gP, (pc & 1), std::integral_constant<int, pbt>{}, max(psv, 0), psv >= 0);
                        if (j == 1) { if (bt > 0) cell2(std::integral_constant<int, pbt>{}, gx_n, hx_s, out_s, raw_s, true); else cell2(std::integral_constant<int, pbt>{}, gx_s, hx_p, out_p, raw_p, psv >= 0); }
                    }
                }, [&] {
                    // SKEW: was the previous phase's panel whole?  (rare: no -- a granule had not arrived: the panel is fetched until it is, that
                    // tile's products are computed again; its cell update has not begun, the request that reuses its panel buffer not been sent)
                    if constexpr (SKEW) { if (have_v && s_flag[vslot] != 0) redo(pbt, psv, pb ^ 1, pc & 1); }
                }) && s >= 1;
                const bool forced = a.force_mask >= 0 && s >= 1 && (pc & a.force_mask) == 0;
                XSTAMP(1);
                send_partials(role_, gs, (pc & 1) ^ 1);
                if constexpr (SKEW) {
                    post(stale || forced);
                    if constexpr (OWNER) wait_vmcnt(3); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next panel's pieces and the next gx slab: all but cell2()'s three stores
                    lds_barrier();                                      // the next panel is whole; this phase's panel buffer is free again; the partial sums are in place
                    posted();
                    have_v = true;
                    XSTAMP(2);
                } else {
                    const bool again = wg_any(stale || forced);         // the barrier
                    XSTAMP(2);
                    if (again) {   // (rare) a granule of this phase's panel had not arrived: fetch the panel until it is whole, multiply again
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!ensure_fresh(bt, s - 1, pb)) { dead = true; return; }
                        products(std::false_type{}, pb, 0u, gs, [](int) {}, [] {});
                        send_partials(role_, gs, (pc & 1) ^ 1);
                        lds_barrier();                                  // the partial sums are in place; the panel buffer may be the target of the next phase's request
                        if (DBG) ph[5] += 1000;
                    }
                }
                if (SKEW) {
#pragma unroll
                    for (int t = 0; t < NT; t++) gP[t] = gs[t];
                } else {
                    if constexpr (OWNER) cell1(role_, gs, (pc & 1) ^ 1, bt_, s, true);
                    XSTAMP(3);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next panel's pieces (travelling since the product loop ended) -- before the publish, whose acknowledgement takes longer
                    lds_barrier();
                    if constexpr (OWNER) cell2(bt_, gx_n, hx_s, out_s, raw_s, true);
                    XSTAMP(4);
                }
                XSTAMP(0);
            };
            x6_static_for<NBT>([&](auto bt_) { if (!dead) phase(bt_); });
        }
        if constexpr (SKEW) {   // the last phase's verdict and cell update (its partial sums were sent before the last barrier)
            if (!dead && have_v && s_flag[vslot] != 0) redo(NBT - 1, T - 1, (pc & 1) ^ 1, (pc & 1) ^ 1);
            if constexpr (OWNER) {
                if (!dead) {
                    cell1(role_, gP, (pc & 1) ^ 1, std::integral_constant<int, NBT - 1>{}, T - 1, true);
                    cell2(std::integral_constant<int, NBT - 1>{}, gx_at(t_of(T - 1)), const_cast<unsigned char *>(hx_parity(T - 1)), out_main + (size_t)t_of(T - 1) * slab, raw_main + (size_t)t_of(T - 1) * slab, true);
                }
            }
        }
    };
    if (kh == 0) run(std::integral_constant<int, 0>{});
    else if (rh == 0) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    if (DBG && tid == ((blockIdx.x >> 3) & 3) * 64)   // (one wave per workgroup reports: wave = member % 4)
        for (int i = 0; i < 6; i++) a.dbg[blockIdx.x * 6 + i] = ph[i];
}

static size_t x6_smem(int H, int NBT) {
    const int UW = H / 16;
    return (size_t)16 * UW * 4 * 2 + (size_t)3 * 3 * 16 * 8 * 2 + (size_t)NBT * (NBT == 1 ? 2 : 1) * 16 * UW * 16 + (size_t)2 * 3 * (H / 8) * 256 + (size_t)2 * 6 * 2 * 1024;
}

size_t lstm_x6_hx_bytes(int H, int B) {
    const int bgr = (B + 7) / 8, nbt = (bgr + 15) / 16;
    return (size_t)2 * 16 * nbt * 3 * (H / 8) * 256;
}

template <int H, int NBT, bool DBG = false>
static int launch_x6_t(const X6Args &a, hipStream_t st) {
    if (int rc = launch_zero_fill(a.sync, 32 * sizeof(unsigned int), st)) return rc;
    if (int rc = launch_zero_fill(a.hx, (size_t)2 * 16 * NBT * 3 * (H / 8) * 256, st)) return rc;   // tags must start at 0 on every launch (by a kernel: lstm.hip)
    hipLaunchKernelGGL((lstm_layer_x6_kernel<H, NBT, DBG>), dim3(kPersistGrid), dim3(256), x6_smem(H, NBT), st, a);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}

// MDD_X6_DEV (development builds): only the instantiations under study are compiled (a full build of this file takes minutes)
template <int H, int N>
static constexpr bool x6_built() {
#ifdef MDD_X6_DEV
    return H == 384 && N <= 4;
#else
    return true;
#endif
}
template <int H, int N>
static int launch_x6_pick(const X6Args &a, hipStream_t st) {
    if constexpr (!x6_built<H, N>()) { set_error("persistent x6 lstm: H=%d with %d tiles per team is not in this development build", H, N); return MDD_ERR_ARG; }
    else {
        if constexpr (H == 384) { if (a.dbg) return launch_x6_t<H, N, true>(a, st); }
        return launch_x6_t<H, N>(a, st);
    }
}

int lstm_x6_max_b(int H) { return 8 * 16 * 8; }   // 8 groups x 8 tiles x 16 rows

int launch_lstm_layer_x6(const LstmStepArgs &s, const unsigned short *whh3, unsigned short *hx, unsigned int *sync, int *err_flag, hipStream_t st) {
    X6Args a;
    a.gx = s.gx; a.whh3 = whh3; a.hx = hx; a.sync = sync; a.err_flag = err_flag;
    a.out = s.out; a.out_raw = s.out_raw; a.oscale = s.oscale; a.oshift = s.oscale ? s.oshift : nullptr;
    a.T = s.T; a.B = s.B; a.BGr = (s.B + 7) / 8; a.seqlen = s.seqlen;
    { const char *fr = getenv("MDD_X6_FORCE_REDO"); const int n = fr ? atoi(fr) : 0; a.force_mask = (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }
    const int nbt = (a.BGr + 15) / 16;
    a.dbg = (getenv("MDD_LSTM_DBG") && s.T > 100) ? reinterpret_cast<long long *>(reinterpret_cast<unsigned char *>(hx) + lstm_x6_hx_bytes(s.H, s.B)) : nullptr;
    if (s.out_split.hi || s.gates_save || (!a.out && !a.out_raw) || !whh3) { set_error("persistent x6 lstm: fp32 outputs, inference only"); return MDD_ERR_ARG; }
    if (nbt < 1 || nbt > 8) { set_error("persistent x6 lstm: B=%d needs %d row tiles per team (max 8)", s.B, nbt); return MDD_ERR_ARG; }
#define X6_CASE(H_, N_) case N_: return launch_x6_pick<H_, N_>(a, st)
    if (s.H == 384) switch (nbt) { X6_CASE(384, 1); X6_CASE(384, 2); X6_CASE(384, 3); X6_CASE(384, 4); X6_CASE(384, 5); X6_CASE(384, 6); X6_CASE(384, 7); X6_CASE(384, 8); }
    if (s.H == 256) switch (nbt) { X6_CASE(256, 1); X6_CASE(256, 2); X6_CASE(256, 3); X6_CASE(256, 4); X6_CASE(256, 5); X6_CASE(256, 6); X6_CASE(256, 7); X6_CASE(256, 8); }
#undef X6_CASE
    set_error("persistent x6 lstm: unsupported H=%d", s.H);
    return MDD_ERR_ARG;
}

template <int H, int N>
static int x6_attr() {
    if constexpr (x6_built<H, N>()) {
        MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_x6_kernel<H, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        if constexpr (H == 384) MDD_HIP_CHECK(hipFuncSetAttribute((const void *)lstm_layer_x6_kernel<H, N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    }
    return MDD_OK;
}

int init_lstm_x6_attributes() {
#define XATTR(H, N) if (int rc = x6_attr<H, N>()) return rc
    XATTR(384, 1); XATTR(384, 2); XATTR(384, 3); XATTR(384, 4); XATTR(384, 5); XATTR(384, 6); XATTR(384, 7); XATTR(384, 8);
    XATTR(256, 1); XATTR(256, 2); XATTR(256, 3); XATTR(256, 4); XATTR(256, 5); XATTR(256, 6); XATTR(256, 7); XATTR(256, 8);
#undef XATTR
    return MDD_OK;
}

int persistent_x6_grid_fits(int n_cu) {
    if (n_cu < kPersistGrid) return 0;
    int per_cu = 0;
    constexpr int N = x6_built<384, 8>() ? 8 : 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)lstm_layer_x6_kernel<384, N>, 256, x6_smem(384, N)) != hipSuccess) return 0;
    return per_cu >= 1 ? 1 : 0;
}

}  // namespace mdd
