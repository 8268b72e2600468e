// Diagnostic: how fast can a CU ingest L2-resident data through LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction)?
// waves per workgroup x pieces in flight; one workgroup per CU; each workgroup cycles over its own 512 KB window (beyond the 32 KB L1).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2000
template <int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k(long long *out, const unsigned char *src, int stride_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned char *base = src + (size_t)blockIdx.x * (512 << 10);
    const unsigned la = (unsigned)wave * 2048u;
    // stride_rows == 0: 1 KB contiguous per instruction; else 16 rows x 64 B with a row stride (row-major plane)
    const unsigned voff = stride_rows ? (unsigned)((lane >> 2) * stride_rows + (lane & 3) * 16) : (unsigned)lane * 16u;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        const unsigned char *p = base + (size_t)(((it * nw + wave) * 1024) & ((512 << 10) - 1) & ~1023) * (stride_rows ? 0 : 1) + (stride_rows ? (size_t)(((it * nw + wave) & 127) * 64) : 0);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(p), "s"(la + (unsigned)((it & 1) * 1024)) : "memory", "m0");
        if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (lds[threadIdx.x] == 77 && out[1] == 3) out[2] = 1;
}
template <int INFLIGHT> void run(const char *name, int threads, int stride, long long *d, unsigned char *src) {
    hipLaunchKernelGGL((k<INFLIGHT>), dim3(256), dim3(threads), 32768, 0, d, src, stride);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<INFLIGHT>), dim3(256), dim3(threads), 32768, 0, d, src, stride);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)ITERS * (threads / 64) * 1024.0;
    printf("%-52s %2d waves: %6.1f cycles per piece per wave, %5.1f B/clk per CU, %6.2f TB/s chip-wide\n", name, threads / 64, (double)h / ITERS, bytes / (double)h,
           bytes * 256 / (ms * 1e-3) / 1e12);
}
int main() {
    long long *d; unsigned char *src; hipMalloc(&d, 64); hipMemset(d, 0, 64); hipMalloc(&src, (size_t)256 * (512 << 10) + (1 << 20)); hipMemset(src, 1, (size_t)256 * (512 << 10) + (1 << 20));
    for (int threads : {64, 256, 512}) {
        run<1>("contiguous 1 KB, 1 in flight per wave", threads, 0, d, src);
        run<4>("contiguous 1 KB, 4 in flight per wave", threads, 0, d, src);
        run<8>("contiguous 1 KB, 8 in flight per wave", threads, 0, d, src);
        run<0>("contiguous 1 KB, unbounded in flight", threads, 0, d, src);
        run<8>("16 rows x 64 B (row stride 3904 B), 8 in flight", threads, 3904, d, src);
    }
    return 0;
}
