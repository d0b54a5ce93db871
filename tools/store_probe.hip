// Per-CU store / load throughput of the access shapes a GEMM epilogue can use for a 256 x 256 tile
// (diagnostic; build: hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/bin/store_probe).
//   shape 0: "row per lane"  - the MFMA accumulator layout: one wave instruction covers 16 rows x 64 B
//   shape 1: "full lines"    - 8 lanes x 16 B per row: one wave instruction covers 8 rows x 128 B
//   shape 2: "whole rows"    - 32 lanes x 16 B per row: one wave instruction covers 2 rows x 512 B
// 256 workgroups of 512 threads (one per CU: each asks for 128 KiB of LDS), each walks `tiles` tiles of
// 256 rows x 256 bf16 columns of a [rows, ld] matrix (ld = 3072 elements), writing or reading every byte once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LOAD>
__global__ __launch_bounds__(512, 1) void probe(char* base, long ld_bytes, int tiles, int tiles_n, u32x4* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const u32x4 val = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        char* tile = base + (long)tm * 256 * ld_bytes + (long)tn * 512;
        if (SHAPE == 0) {
            // wave sub-tile: rows wm*128 .. +127, byte columns wn*128 .. +127
            const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    char* p = tile + (long)(wm * 128 + i * 16 + l15) * ld_bytes + wn * 128 + jp * 64 + g * 16;
                    if (LOAD) { const u32x4 v = *(const u32x4*)p; acc += v; } else { *(u32x4*)p = val; }
                }
        } else if (SHAPE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* p = tile + (long)(wm * 128 + i * 8 + (lane >> 3)) * ld_bytes + wn * 128 + (lane & 7) * 16;
                if (LOAD) { const u32x4 v = *(const u32x4*)p; acc += v; } else { *(u32x4*)p = val; }
            }
        } else {
            // the whole 256 x 512 B tile by rows: wave w takes rows 32 w .. 32 w + 31, two per instruction
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* p = tile + (long)(wid * 32 + i * 2 + (lane >> 5)) * ld_bytes + (lane & 31) * 16;
                if (LOAD) { const u32x4 v = *(const u32x4*)p; acc += v; } else { *(u32x4*)p = val; }
            }
        }
    }
    if (LOAD && acc[0] == 0x12345678u) sink[threadIdx.x] = acc;
    if (smem[0] == 77 && threadIdx.x == 9999) sink[0] = acc;
}

template <int SHAPE, bool LOAD>
float run(char* buf, long ld_bytes, int tiles, int tiles_n, u32x4* sink, int reps, int grid) {
    hipFuncSetAttribute((const void*)probe<SHAPE, LOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, LOAD>), dim3(grid), dim3(512), 131072, 0, buf, ld_bytes, tiles, tiles_n, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((probe<SHAPE, LOAD>), dim3(grid), dim3(512), 131072, 0, buf, ld_bytes, tiles, tiles_n, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int tiles_n = 12, tiles_m = 394;            // a [100864, 3072] bf16 matrix: fc1's output at batch 512
    const int tiles = tiles_n * tiles_m;
    const long ld_bytes = 3072 * 2;
    const size_t bytes = (size_t)tiles_m * 256 * ld_bytes;
    char* buf; u32x4* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 512 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const double total = (double)tiles * 256 * 512;
    const char* names[3] = {"row per lane (16 rows x 64 B / instr)", "full lines  (8 rows x 128 B / instr)", "whole rows  (2 rows x 512 B / instr)"};
    // fewer workgroups than CUs: what ONE CU can store or load when the memory system is not the limit
    const int grids[4] = {256, 64, 16, 8};
    for (int gi = 0; gi < 4; ++gi) {
        const int grid = grids[gi];
        const int tl = tiles * grid / 256;             // same number of tiles per workgroup
        float ms[6];
        ms[0] = run<0, false>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        ms[1] = run<1, false>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        ms[2] = run<2, false>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        ms[3] = run<0, true>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        ms[4] = run<1, true>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        ms[5] = run<2, true>(buf, ld_bytes, tl, tiles_n, sink, 5, grid);
        for (int k = 0; k < 6; ++k)
            printf("%3d WGs %s %-40s %.3f ms  %.2f TB/s  %.2f us per tile per CU = %.1f B/ns/CU\n", grid, k < 3 ? "store" : "load ", names[k % 3],
                   ms[k], (double)tl * 131072 / ms[k] / 1e9, ms[k] * 1e3 / (tl / (double)grid), 131072.0 / (ms[k] * 1e6 / (tl / (double)grid)));
    }
    return 0;
}
