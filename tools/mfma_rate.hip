// Diagnostic: cycles per v_mfma_f32_16x16x32_bf16 for a lone wave (or two) per SIMD, by accumulator count / register file / interleaved VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 20000
template <int NACC, int MODE>   // MODE 4: one ds_read_b128 per 4 MFMAs; 5: same + s_barrier per 24 MFMAs; MODE 0: all VGPR; 1: acc in AGPR (asm); 2: acc+A in AGPR (asm); 3: VGPR with 2 VALU after each MFMA
__global__ void k(long long *out, float *sink, unsigned seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned x = seed + threadIdx.x, y = seed * 3 + 1, z = 0;
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 lv = {0, 0, 0, 0};
    int cnt = 0;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (MODE == 0 || MODE >= 3) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                else if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a), "v"(b));
                if (MODE >= 4 && ((r * NACC + i) & 3) == 0) { const u32x4 t = *reinterpret_cast<const u32x4 *>(&lds[((threadIdx.x * 4 + it * 64) & 8188)]); lv[0] ^= t[0]; lv[1] ^= t[3]; }
                if (MODE == 5 && (r * NACC + i) == 0 && (it & 1) == 0) asm volatile("s_barrier" ::: "memory");
                if (MODE == 3) { z = __builtin_amdgcn_perm(x, y, 0x05040100u) + z; x = __builtin_amdgcn_perm(y, z, 0x07060302u); }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(x + z + lv[0] + lv[1]);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NACC, int MODE> void run(const char *name, int threads, long long *d_out, float *sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, d_out, sink, 7u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, d_out, sink, 7u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    const double mf = (double)ITERS * 12.0, waves = 256.0 * threads / 64;
    printf("%-40s waves/SIMD %d: %.1f ticks per MFMA per wave; kernel %.3f ms -> %.0f TFLOP/s bf16 chip-wide, tick rate %.2f GHz\n", name, threads / 256,
           (double)h / mf, ms, waves * mf * 16384.0 / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
}
int main() {
    long long *d; float *sink; hipMalloc(&d, 8); hipMalloc(&sink, 256 * 512 * 4);
    run<3, 0>("3 acc, VGPR", 256, d, sink); run<6, 0>("6 acc, VGPR", 256, d, sink); run<12, 0>("12 acc, VGPR", 256, d, sink);
    run<3, 1>("3 acc in AGPR", 256, d, sink); run<12, 1>("12 acc in AGPR", 256, d, sink);
    run<3, 2>("3 acc + A in AGPR", 256, d, sink); run<12, 2>("12 acc + A in AGPR", 256, d, sink);
    run<3, 3>("3 acc, VGPR, 2 VALU after each", 256, d, sink); run<12, 3>("12 acc, VGPR, 2 VALU after each", 256, d, sink);
    run<3, 0>("3 acc, VGPR", 512, d, sink); run<12, 0>("12 acc, VGPR", 512, d, sink); run<3, 3>("3 acc, VGPR, 2 VALU after each", 512, d, sink);
    run<1, 0>("1 acc (dependent chain), VGPR", 256, d, sink); run<2, 0>("2 acc, VGPR", 256, d, sink);
    run<12, 4>("12 acc, 1 ds_read_b128 per 4 MFMA", 256, d, sink); run<12, 4>("12 acc, 1 ds_read_b128 per 4 MFMA", 512, d, sink);
    run<12, 5>("12 acc, ds_read + barrier per 24 MFMA", 256, d, sink); run<12, 5>("12 acc, ds_read + barrier per 24 MFMA", 512, d, sink);
    run<3, 0>("3 acc, VGPR", 768, d, sink); run<3, 0>("3 acc, VGPR", 1024, d, sink); run<3, 3>("3 acc, VGPR, 2 VALU after each", 1024, d, sink);
    return 0;
}
