// How fast 252 workgroups can add their 256 x 256 fp32 tiles into a [768, 3072] fp32 matrix (7 workgroups per tile:
// the split-K write-out of the weight-gradient GEMM), by access shape (diagnostic;
// build: hipcc -O3 --offload-arch=gfx950 tools/atomic_probe.hip -o tools/bin/atomic_probe).
//   shape 0: MFMA accumulator layout, one atomic instruction = 4 rows x 16 consecutive floats
//   shape 1: one atomic instruction = 1 row x 64 consecutive floats (256 B)
//   shape 2: shape 1 as plain stores into a per-split slab (what a separate reduction pass would read)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void probe(float* C, float* slab, int ldc, int tiles_n, int ntiles) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int split = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const int bm = (tile / tiles_n) * 256, bn = (tile % tiles_n) * 256;
    const float v = 1.0f + lane;
    if (SHAPE == 0) {
        const int l15 = lane & 15, g = lane >> 4;
        float* crow = C + (long)(bm + wm * 128 + 4 * g) * ldc + bn + wn * 64 + l15;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(crow + (long)(i * 16 + r) * ldc + j * 16, v);
    } else if (SHAPE == 1) {
        float* crow = C + (long)(bm + wm * 128) * ldc + bn + wn * 64 + lane;
#pragma unroll 16
        for (int i = 0; i < 128; ++i) atomicAdd(crow + (long)i * ldc, v);
    } else {
        float* crow = slab + ((long)split * 768 + bm + wm * 128) * ldc + bn + wn * 64 + lane;
#pragma unroll 16
        for (int i = 0; i < 128; ++i) crow[(long)i * ldc] = v;
    }
}

template <int SHAPE>
float run(float* C, float* slab, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(252), dim3(512), 0, 0, C, slab, 3072, 12, 36);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<SHAPE>), dim3(252), dim3(512), 0, 0, C, slab, 3072, 12, 36);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    float *C, *slab;
    if (hipMalloc(&C, 768L * 3072 * 4) != hipSuccess || hipMalloc(&slab, 7L * 768 * 3072 * 4) != hipSuccess) return 1;
    hipMemset(C, 0, 768L * 3072 * 4);
    printf("accumulator-layout atomics   %.1f us per launch\n", run<0>(C, slab, 20));
    printf("row-contiguous atomics       %.1f us per launch\n", run<1>(C, slab, 20));
    printf("row-contiguous slab stores   %.1f us per launch\n", run<2>(C, slab, 20));
    return 0;
}
