// Diagnostic: what one vector-memory instruction costs a lone wave per SIMD inside an MFMA stream (9 x v_mfma_f32_16x16x32_bf16 per
// memory instruction, as one k-step of the persistent BiLSTM): nothing / LDS-DMA (global_load_lds_dwordx4) / global_load_dwordx4 to
// registers / the same followed later by a ds_write_b128 of the loaded data / a 16-byte sc1 store.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
#define ITERS 4000
template <int MODE, int PER>   // PER: memory instructions per 9 MFMAs (1 or 2)
__global__ __launch_bounds__(256, 1) void k(long long *out, float *sink, const unsigned char *src, unsigned char *dst) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f32x4 acc[3];
    for (int i = 0; i < 3; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wave = threadIdx.x >> 6;
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void *)lds + (unsigned)wave * 4096u);
    const unsigned voff = threadIdx.x * 16u;
    const unsigned char *base = src + (size_t)blockIdx.x * 65536;
    u32x4 r0 = {0, 0, 0, 0}, r1 = r0;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        const unsigned char *p = base + (it & 7) * 4096;
#pragma unroll
        for (int m = 0; m < PER; m++) {
            if (MODE == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(p + m * 2048), "s"(la + m * 1024) : "memory", "m0");
            if (MODE == 2 || MODE == 3) {
                if (m == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r0) : "v"(voff), "s"(p) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r1) : "v"(voff), "s"(p + 2048) : "memory");
            }
            if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(voff), "v"(r0), "s"(dst + (size_t)blockIdx.x * 65536 + (it & 7) * 4096 + m * 2048) : "memory");
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {   // the data loaded two iterations ago is surely there: wait for all but the youngest 2*PER, then park it in LDS
            if (PER == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(r0));
            else asm volatile("s_waitcnt vmcnt(2)" : "+v"(r0), "+v"(r1));
            *reinterpret_cast<u32x4 *>(lds + wave * 4096 + (threadIdx.x & 63) * 16) = r0;
            if (PER == 2) *reinterpret_cast<u32x4 *>(lds + wave * 4096 + 1024 + (threadIdx.x & 63) * 16) = r1;
        } else if (MODE != 0) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1));
    long long t1 = __builtin_readcyclecounter();
    float s = acc[0][0] + acc[1][1] + acc[2][2] + (float)(r0[0] + r1[1] + lds[threadIdx.x]);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE, int PER> void run(const char *name, long long *d_out, float *sink, unsigned char *src, unsigned char *dst) {
    hipLaunchKernelGGL((k<MODE, PER>), dim3(256), dim3(256), 0, 0, d_out, sink, src, dst);
    hipLaunchKernelGGL((k<MODE, PER>), dim3(256), dim3(256), 0, 0, d_out, sink, src, dst);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-62s %7.1f cycles per group of 9 MFMAs (floor 147)\n", name, (double)h / ITERS);
}
int main() {
    long long *d; float *sink; unsigned char *src, *dst;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&src, 256 * 65536); hipMalloc(&dst, 256 * 65536);
    hipMemset(src, 1, 256 * 65536);
    run<0, 1>("9 MFMAs alone", d, sink, src, dst);
    run<1, 1>("+ 1 LDS-DMA piece (global_load_lds_dwordx4)", d, sink, src, dst);
    run<1, 2>("+ 2 LDS-DMA pieces", d, sink, src, dst);
    run<2, 1>("+ 1 global_load_dwordx4 to registers", d, sink, src, dst);
    run<2, 2>("+ 2 global_load_dwordx4 to registers", d, sink, src, dst);
    run<3, 1>("+ 1 global_load_dwordx4 + ds_write_b128 of earlier data", d, sink, src, dst);
    run<3, 2>("+ 2 global_load_dwordx4 + 2 ds_write_b128", d, sink, src, dst);
    run<4, 1>("+ 1 global_store_dwordx4 sc1", d, sink, src, dst);
    run<4, 2>("+ 2 global_store_dwordx4 sc1", d, sink, src, dst);
    return 0;
}
