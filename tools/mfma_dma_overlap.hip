// Diagnostic: does an LDS-DMA piece (global_load_lds_dwordx4) issued between bf16 MFMAs cost the wave its issue time (~84 cycles) or
// does the matrix pipe keep running?  Per iteration: NM x v_mfma_f32_16x16x32_bf16 (12 independent accumulators) + NP pieces.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 3000
template <int NM, int NP, int MODE>   // MODE 0: LDS-DMA; 1: plain global_load_dwordx4 to registers (never waited for inside the loop)
__global__ __launch_bounds__(512, 1) void k(long long *out, float *sink, const unsigned char *src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f32x4 acc[12];
    for (int i = 0; i < 12; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char *base = src + (size_t)(blockIdx.x & 7) * (256 << 10);     // a few hot 256 KB windows: L2 hits
    const unsigned voff = (unsigned)lane * 16u, la = (unsigned)wave * 4096u;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 12]) : "v"(a), "v"(b));
            if (NP && (m % (NM / NP)) == 0 && m / (NM / NP) < NP) {
                const unsigned char *p = base + (size_t)((((it * NP + m) * 8 + wave) * 1024) & ((256 << 10) - 1));
                if (MODE == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(p), "s"(la + (unsigned)((m & 3) * 1024)) : "memory", "m0");
                else { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(voff), "s"(p) : "memory"); g = t; }
            }
        }
        if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float s = g[0];
    for (int i = 0; i < 12; i++) s += acc[i][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NM, int NP, int MODE> void run(const char *name, int threads, long long *d, float *sink, unsigned char *src) {
    hipLaunchKernelGGL((k<NM, NP, MODE>), dim3(256), dim3(threads), 65536, 0, d, sink, src);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NM, NP, MODE>), dim3(256), dim3(threads), 65536, 0, d, sink, src);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-64s %d wave(s)/SIMD: %7.1f cycles per iteration (%d MFMAs = %d cycles of matrix pipe per wave)\n", name, threads / 256, (double)h / ITERS, NM, NM * 16);
}
int main() {
    long long *d; float *sink; unsigned char *src; hipMalloc(&d, 64); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&src, 4 << 20); hipMemset(src, 0, 4 << 20);
    for (int threads : {256, 512}) {
        run<24, 0, 0>("24 MFMAs, no loads", threads, d, sink, src);
        run<24, 1, 0>("24 MFMAs + 1 LDS-DMA piece", threads, d, sink, src);
        run<24, 2, 0>("24 MFMAs + 2 LDS-DMA pieces", threads, d, sink, src);
        run<24, 3, 0>("24 MFMAs + 3 LDS-DMA pieces", threads, d, sink, src);
        run<24, 2, 1>("24 MFMAs + 2 global_load_dwordx4 (registers)", threads, d, sink, src);
        run<24, 3, 1>("24 MFMAs + 3 global_load_dwordx4 (registers)", threads, d, sink, src);
    }
    return 0;
}
