// Diagnostic: the product loop of the persistent BiLSTM layer kernel (one 16-row tile: 12 k-steps x 3 row tiles x 3 products =
// 108 v_mfma_f32_16x16x32_bf16 per wave, floor 108 x 16 = 1728 cycles), rebuilt piece by piece around the same register layout
// (36 hi fragments in vector registers, 36 lo fragments fed from the accumulation file, B operands read from LDS as the travelling
// 16-byte chunks {hi x4 | lo x4}), one wave per SIMD, every CU active.  Which ingredient takes the loop from 1728 to ~3900 cycles?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lstm_kstep.hip -o lstm_kstep && ./lstm_kstep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
#define ITERS 2000
constexpr int KS = 12, RTW = 3;

__device__ __forceinline__ void mfma_a(f32x4 &acc, const bf16x8 &a_agpr, const bf16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a_agpr), "v"(b));
}

// LDSB: B operands come from LDS (else constant registers);  AND: clear the tag bits (4 v_and) and sum the tagged words;
// AGPR: lo fragments fed from the accumulation file by inline asm (else all products through the builtin with the hi fragments);
// DMA: one LDS-DMA piece per k-step for the first 9 k-steps;  SB: sched_barrier between the product groups (as the kernel has)
template <bool LDSB, bool AND, bool AGPR, bool DMA, bool SB>
__global__ __launch_bounds__(256, 1) void k(long long *out, float *sink, const unsigned short *w, const unsigned char *src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    bf16x8 ah[RTW][KS], al[RTW][KS];
#pragma unroll
    for (int rt = 0; rt < RTW; rt++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ah[rt][ks] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)((wave * RTW + rt) * 16 + li) * 384 + ks * 32 + kq * 8));
            al[rt][ks] = *reinterpret_cast<const bf16x8 *>(w + 192 * 384 + ((size_t)((wave * RTW + rt) * 16 + li) * 384 + ks * 32 + kq * 8));
        }
    for (int i = tid; i < 2 * 24576 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = 0x3c003c01u + i;
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void *)lds + 24576u + (unsigned)wave * 1024u);
    const unsigned voff = tid * 16u;
    __syncthreads();
    f32x4 tot[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; rt++) tot[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned tags = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        const unsigned char *fb = lds + kq * 512 + li * 16;
        f32x4 acc[RTW];
        constexpr int PD = 3;
        u32x4 ra[PD], rb[PD];
#pragma unroll
        for (int p = 0; p < PD; p++) {
            ra[p] = LDSB ? *reinterpret_cast<const u32x4 *>(fb + p * 2048) : (u32x4){1u + p, 2u, 3u, 4u};
            rb[p] = LDSB ? *reinterpret_cast<const u32x4 *>(fb + p * 2048 + 256) : (u32x4){5u, 6u + p, 7u, 8u};
        }
        unsigned sraw = 0, smask = 0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const u32x4 xa = ra[ks % PD], xb = rb[ks % PD];
            if (LDSB && ks + PD < KS) {
                ra[ks % PD] = *reinterpret_cast<const u32x4 *>(fb + (ks + PD) * 2048);
                rb[ks % PD] = *reinterpret_cast<const u32x4 *>(fb + (ks + PD) * 2048 + 256);
            }
            u32x4 hq, lq;
            hq[0] = xa[0]; hq[1] = xa[1]; hq[2] = xb[0]; hq[3] = xb[1];
            if (AND) {
                lq[0] = xa[2] & 0xfffefffeu; lq[1] = xa[3] & 0xfffefffeu; lq[2] = xb[2] & 0xfffefffeu; lq[3] = xb[3] & 0xfffefffeu;
                sraw += xa[2] + xb[2]; smask += lq[0] + lq[2];
            } else { lq[0] = xa[2]; lq[1] = xa[3]; lq[2] = xb[2]; lq[3] = xb[3]; }
            const bf16x8 bh = __builtin_bit_cast(bf16x8, hq), bl = __builtin_bit_cast(bf16x8, lq);
            if (DMA && ks < 9) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src + (size_t)blockIdx.x * 65536 + ((it * 9 + ks) & 15) * 4096), "s"(la) : "memory", "m0");
#pragma unroll
            for (int rt = 0; rt < RTW; rt++)
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl, ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt], 0, 0, 0);
            if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {
                if (AGPR) mfma_a(acc[rt], al[rt][ks], bh);
                else acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh, acc[rt], 0, 0, 0);
            }
            if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh, acc[rt], 0, 0, 0);
            if (SB) __builtin_amdgcn_sched_barrier(0);
        }
        tags += sraw - smask;
#pragma unroll
        for (int rt = 0; rt < RTW; rt++) { tot[rt][0] += acc[rt][0]; tot[rt][1] += acc[rt][1]; tot[rt][2] += acc[rt][2]; tot[rt][3] += acc[rt][3]; }
        if (DMA) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + tid] = tot[0][0] + tot[1][1] + tot[2][2] + (float)tags;
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <bool LDSB, bool AND, bool AGPR, bool DMA, bool SB>
void run(const char *name, long long *d_out, float *sink, unsigned short *w, unsigned char *src) {
    auto kern = k<LDSB, AND, AGPR, DMA, SB>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 56 * 1024, 0, d_out, sink, w, src);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-88s %7.0f cycles per tile (floor 1728)\n", name, (double)h / ITERS);
}
int main() {
    long long *d; float *sink; unsigned short *w; unsigned char *src;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&w, 2 * 192 * 384 * 2); hipMalloc(&src, 256 * 65536);
    hipMemset(w, 0x3c, 2 * 192 * 384 * 2); hipMemset(src, 1, 256 * 65536);
    //   LDSB   AND    AGPR   DMA    SB
    run<false, false, false, false, false>("products only, B constant, all through the builtin, scheduler free", d, sink, w, src);
    run<false, false, false, false, true >("+ sched_barriers between the product groups", d, sink, w, src);
    run<false, false, true,  false, true >("+ lo fragments from the accumulation file (inline asm)", d, sink, w, src);
    run<true,  false, true,  false, true >("+ B operands from LDS chunks (ds_read_b128 x2 per k-step, 3 k-steps ahead; 4 v_mov)", d, sink, w, src);
    run<true,  true,  true,  false, true >("+ tag bits cleared and summed (4 v_and + adds)", d, sink, w, src);
    run<true,  true,  true,  true,  true >("+ 9 LDS-DMA pieces per tile  (= the kernel's loop)", d, sink, w, src);
    run<true,  true,  true,  true,  false>("the kernel's loop without sched_barriers", d, sink, w, src);
    run<true,  false, false, false, false>("LDS chunks, builtin only, scheduler free", d, sink, w, src);
    return 0;
}
