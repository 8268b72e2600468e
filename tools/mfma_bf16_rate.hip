// Diagnostic: what hides behind v_mfma_f32_16x16x32_bf16 for a lone wave per SIMD (the f32x6 recurrence's situation; the fp32 MFMA's twin is mfma_f32_rate.hip):
// cycles per MFMA with k independent VALU / transcendental / LDS / LDS-DMA instructions interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 4000
// MODE 0: MFMAs only.  1: K v_fma_f32 after each MFMA.  2: K v_exp_f32 after each MFMA.  3: one ds_read_b128 per 12 MFMAs.
// 4: one global_load_lds_dwordx4 per 12 MFMAs (K = 1) or per 24 (K = 2).  5: K v_and_b32 after each MFMA.  6: one global_load_dwordx4 per 12.
template <int MODE, int K>
__global__ __launch_bounds__(256, 1) void k(long long *out, float *sink, const float *src, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((threadIdx.x + i + 1) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i + 1) & 3); }
    f32x4 acc[6];
    for (int i = 0; i < 6; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = 0.001f * (float)(threadIdx.x + i + seed);
    unsigned u[8];
    for (int i = 0; i < 8; i++) u[i] = threadIdx.x * 7 + i + seed;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    u32x4 lv = {0, 0, 0, 0};
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(threadIdx.x >> 6) * 1024u);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 12; r++) {
            acc[r % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[r % 6], 0, 0, 0);
            if (MODE == 1) { for (int q = 0; q < K; q++) v[(r * K + q) & 7] = __builtin_fmaf(v[(r * K + q) & 7], 1.0001f, 0.5f); }
            if (MODE == 2) { for (int q = 0; q < K; q++) v[(r * K + q) & 7] = __builtin_amdgcn_exp2f(v[(r * K + q) & 7]); }
            if (MODE == 5) { for (int q = 0; q < K; q++) u[(r * K + q) & 7] &= 0xbfffffffu + it; }
            if (MODE == 3 && r == 0) { const u32x4 t = *reinterpret_cast<const u32x4 *>(&lds[((threadIdx.x * 4 + it * 64) & 8188)]); lv[0] ^= t[0]; lv[1] ^= t[3]; }
            if (MODE == 4 && r == 0 && (it % K) == 0)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((threadIdx.x & 63) * 16u + (unsigned)(it & 63) * 4096u), "s"(src), "s"(la) : "memory", "m0");
            if (MODE == 6 && r == 0) { const f32x4 t = *reinterpret_cast<const f32x4 *>(src + (threadIdx.x & 63) * 4 + (it & 63) * 1024); gv[0] += t[0]; gv[1] += t[3]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 6; i++) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i] + (float)u[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(lv[0] + lv[1]) + gv[0] + gv[1] + (float)lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE, int K> void run(const char *name, long long *d_out, float *sink, float *src) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), 32768, 0, d_out, sink, src, 7u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), 32768, 0, d_out, sink, src, 7u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    const double mf = (double)ITERS * 12.0;
    printf("%-58s %6.1f ticks per MFMA; kernel %.3f ms -> %.0f TFLOP/s bf16 chip-wide, tick rate %.2f GHz\n", name, (double)h / mf, ms,
           1024.0 * mf * 16384.0 / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
}
int main() {
    long long *d; float *sink, *src; hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    run<0, 0>("MFMA only (6 accumulators)", d, sink, src);
    run<1, 1>("+ 1 v_fma_f32 per MFMA", d, sink, src); run<1, 2>("+ 2 v_fma_f32 per MFMA", d, sink, src); run<1, 3>("+ 3 v_fma_f32 per MFMA", d, sink, src); run<1, 4>("+ 4 v_fma_f32 per MFMA", d, sink, src); run<1, 6>("+ 6 v_fma_f32 per MFMA", d, sink, src);
    run<5, 1>("+ 1 v_and_b32 per MFMA", d, sink, src); run<5, 2>("+ 2 v_and_b32 per MFMA", d, sink, src); run<5, 4>("+ 4 v_and_b32 per MFMA", d, sink, src);
    run<2, 1>("+ 1 v_exp_f32 per MFMA", d, sink, src); run<2, 2>("+ 2 v_exp_f32 per MFMA", d, sink, src);
    run<3, 0>("+ 1 ds_read_b128 per 12 MFMAs", d, sink, src);
    run<4, 1>("+ 1 global_load_lds_dwordx4 per 12 MFMAs", d, sink, src); run<4, 2>("+ 1 global_load_lds_dwordx4 per 24 MFMAs", d, sink, src);
    run<6, 0>("+ 1 global_load_dwordx4 (to registers) per 12 MFMAs", d, sink, src);
    return 0;
}
