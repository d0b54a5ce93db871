// Per-CU staging throughput global memory -> LDS of the two ways a GEMM can do it (diagnostic; build:
// hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o tools/bin/dma_probe).
//   kind 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction, no registers)
//   kind 1: global_load_dwordx4 into registers, then ds_write_b128 (two instructions and 4 registers per KiB)
// One workgroup per CU with W waves (4: one per SIMD, 8: two); every wave moves `iters` x 8 KiB, eight loads in
// flight, from a source of `span` bytes per workgroup that is walked cyclically (span 128 KiB: L2-resident after the
// first pass; span = the whole walk: every byte from HBM once).  LDS rows are written linearly; nothing reads them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int KIND>
__global__ __launch_bounds__(512, 1) void probe(const char* src, long span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const char* base = src + (long)blockIdx.x * span;
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS const char*)smem + wid * 8192;
    long pos = (long)wid * 8192;
    for (int it = 0; it < iters; ++it) {
        const char* s = base + pos + lane * 16;
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned dst = lds0 + k * 1024;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(s), "s"(dst), "n"(0) : "memory");
                s += 1024;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(s + k * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned dst = lds0 + k * 1024 + lane * 16;
                asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(v[k]) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        pos += (long)nw * 8192;
        if (pos + 8192 > span) pos = (long)wid * 8192;
    }
    if (smem[lane] == 77 && iters < 0) sink[lane] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int KIND>
void run(const char* name, int waves, long span, const char* src, unsigned* sink) {
    const int iters = 4000;
    CK(hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 65536, 0, src, span, iters, sink);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 65536, 0, src, span, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)waves * iters * 8192;      // per CU
    std::printf("%-34s %d waves, %5ld KiB per workgroup: %.3f ms, %6.1f GB/s per CU (%.1f B per 2.1 GHz cycle), %.2f TB/s over the chip\n",
                name, waves, span >> 10, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9, 256 * bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const long big = 4L << 20;                       // 4 MiB per workgroup: 1 GiB in all, streamed
    char* src;
    unsigned* sink;
    CK(hipMalloc(&src, 256 * big));
    CK(hipMemset(src, 1, 256 * big));
    CK(hipMalloc(&sink, 256));
    for (int waves : {4, 8})
        for (long span : {131072L, big}) {
            run<0>("global_load_lds_dwordx4", waves, span, src, sink);
            run<1>("global_load_dwordx4 + ds_write_b128", waves, span, src, sink);
        }
    return 0;
}
