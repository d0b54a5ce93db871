// Per-CU LDS read throughput of the fragment reads the GEMM and attention kernels use (diagnostic; build:
// hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/bin/lds_probe).
//   kind 0: ds_read_b128          16 B per lane, 1 KiB per wave instruction (the NT kernel's fragments and tile image)
//   kind 1: ds_read_b64           8 B per lane, 512 B per wave instruction
//   kind 2: ds_read_b64_tr_b16    8 B per lane through the hardware transpose (TN / attention fragments)
// One workgroup per CU with W waves (W = 4: one per SIMD, W = 8: two), every wave issues `iters` x 16 independent
// reads from a 64 KiB tile with the swizzled addresses the kernels use; the clock is s_memtime around the loop,
// reported as LDS bytes per cycle and CU (shader clock cycles of the slowest wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int KIND>
__global__ __launch_bounds__(512, 1) void probe(int iters, unsigned long long* cycles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
    __syncthreads();
    // 512-byte rows, 16-byte chunk c of row r at chunk position c ^ ((r & 7) << 1): the TN tile image
    const int l15 = lane & 15, gq = lane >> 4, q = l15 >> 2, pp = l15 & 3;
    unsigned acc = 0;
    // byte addresses of this lane's 16 reads (LDS address space: 32-bit)
    unsigned addr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int col0 = ((wid * 16 + k) & 15) * 16;
        if (KIND == 0) {
            // the NT tile image: 128-byte rows, lane = (row l15, chunk 4 s + g), chunk position c ^ (row & 7)
            const int row = ((wid * 16 + k) & 15) * 16 + l15, chunk = 4 * (k & 1) + gq;
            addr[k] = row * 128 + ((chunk ^ (row & 7)) << 4);
        } else {
            const int row = 4 * gq + q + 16 * (k & 1);
            const int chunk = ((col0 >> 3) + (pp >> 1)) ^ ((row & 7) << 1);
            addr[k] = row * 512 + chunk * 16 + (pp & 1) * 8;
        }
        addr[k] += (unsigned)(size_t)(LDS_AS char*)smem;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // sixteen independent reads in flight, then one wait: the reads are assembly so that nothing is hoisted or merged
        if (KIND == 0) {
            u32x4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(addr[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
        } else {
            u32x2 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(v[k]) : "v"(addr[k]));
                else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[k]) : "v"(addr[k]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 8 + wid] = t1 - t0;
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int KIND>
void run(const char* name, int waves, int iters, unsigned long long* d_cyc, unsigned* sink, int bytes_per_lane) {
    hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipMemset(d_cyc, 0, 256 * 8 * sizeof(unsigned long long));
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 65536, 0, iters, d_cyc, sink);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 65536, 0, iters, d_cyc, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long worst = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) worst = h[b * 8 + w] > worst ? h[b * 8 + w] : worst;
    const double bytes = (double)waves * iters * 16 * 64 * bytes_per_lane;      // per CU
    // s_memtime counts at the constant 100 MHz reference; the wall clock gives the rate, the cycle figure assumes 2.1 GHz
    std::printf("%-20s %d waves: %.3f ms, %.1f GB/s per CU, %.1f B per 2.1 GHz cycle and CU (memtime ticks of the slowest wave: %llu)\n",
                name, waves, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9, worst);
}

int main() {
    unsigned long long* d_cyc;
    unsigned* sink;
    hipMalloc(&d_cyc, 256 * 8 * sizeof(unsigned long long));
    hipMalloc(&sink, 512 * sizeof(unsigned));
    const int iters = 20000;
    for (int waves : {4, 8}) {
        run<0>("ds_read_b128", waves, iters, d_cyc, sink, 16);
        run<1>("ds_read_b64", waves, iters, d_cyc, sink, 8);
        run<2>("ds_read_b64_tr_b16", waves, iters, d_cyc, sink, 8);
    }
    return 0;
}
