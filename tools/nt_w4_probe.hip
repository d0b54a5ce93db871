// Would the one-wave-per-SIMD structure of the weight-gradient kernel (gemm_tn_w4_kernel) also help the NT GEMM?
// A stand-alone NT product C = A . W^T (bf16, no epilogue arithmetic) built that way, timed against the
// library's eight-wave persistent kernel in the same process, alternating, results compared first.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/nt_w4_probe.hip -Lvit-pytorch_amd/vitk -lvitk
//          -Wl,-rpath,'$ORIGIN/../../vit-pytorch_amd/vitk' -o tools/bin/nt_w4_probe
// Four waves (2 x 2) of 128 x 128 per 256 x 256 tile; the reduction axis in 32-deep slabs ([256 rows][64 B] of A
// and of W, 32 KiB) in a ring of five LDS buffers, slab f + 4 in flight; accumulators in accumulation registers,
// MFMAs from inline assembly in place; one fragment read (ds_read_b128) of slab f + 1 behind every fourth MFMA, one
// LDS-DMA piece of slab f + 4 behind every eighth; one barrier per slab.  mode 1 skips the C stores (loop only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include "vitk.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
// 64-byte rows: 16-byte chunk c of row r sits at chunk position c ^ ((-(r >> 2)) & 3)
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

struct Args {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    int M, N, K;
    long lda, ldb, ldc;
    int tiles_m, tiles_n, nwg, no_store;
};

// NV: independent VALU instructions (v_fma_f32 on private registers) issued behind every MFMA: how much of an
// epilogue's arithmetic a lone wave could run under the next tile's MFMAs.
template <int NV>
__global__ __launch_bounds__(256, 1) void nt_w4_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [5][A slab 16 KiB | W slab 16 KiB]
    constexpr int SLAB = 256 * 64, STAGE = 2 * SLAB, RING = 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l15 = lane & 15, g = lane >> 4;
    // tile order: every XCD a contiguous run; bands of 8 tile rows walked column by column
    const int wg = xcd_remap(blockIdx.x, p.nwg);
    constexpr int GM = 8;
    const int band = wg / (GM * p.tiles_n), inb = wg - band * (GM * p.tiles_n);
    const int rows_in_band = (p.tiles_m - band * GM) < GM ? (p.tiles_m - band * GM) : GM;
    const int tn = inb / rows_in_band, tm = band * GM + (inb - tn * rows_in_band);
    const int bm = tm * 256, bn = tn * 256;
    const int ns = p.K / 32;

    // staging: piece q*4 + wid (q = 0..3) of an operand slab = tile rows 16 piece .. +15, lane l at row 16 piece + (l >> 2),
    // chunk position l & 3
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 16 * (q * 4 + wid) + (lane >> 2);
        const int ch = (lane & 3) ^ swz(row);
        const int am = bm + row < p.M ? bm + row : p.M - 1;
        const int wr_ = bn + row < p.N ? bn + row : p.N - 1;
        a_off[q] = (unsigned)(((long)am * p.lda + ch * 8) * 2);
        b_off[q] = (unsigned)(((long)wr_ * p.ldb + ch * 8) * 2);
    }
    const bf16_t* a_base = p.A;
    const bf16_t* b_base = p.W;
    int staged = 0;
    long step = 32;
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS const char*)smem;
    auto dma = [&](const void* sbase, unsigned voff, unsigned dst) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    auto stage_piece = [&](int buf, int k) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wid * 1024);
        if (k < 4) dma(a_base, a_off[k], dst + k * 4096);
        else dma(b_base, b_off[k - 4], dst + SLAB + (k - 4) * 4096);
    };
    auto slab_issued = [&]() __attribute__((always_inline)) {
        ++staged;
        a_base += step; b_base += step;
        if (staged >= ns) {                         // past the end: zeros (see gemm_tn_w4_kernel)
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_off[q] = 0u; b_off[q] = 0u; }
            a_base = b_base = reinterpret_cast<const bf16_t*>(g_zero16);
            step = 0;
        }
    };
    auto frag = [&](const char* tile, int row0) __attribute__((always_inline)) {
        const int row = row0 + l15;
        return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((g ^ swz(row)) << 4));
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dummy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dummy[k] = 1.0f + lane * 1e-3f + k;
    const float dmul = 0.999f;
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) stage_piece(s, k);
        slab_issued();
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa0[i] = frag(smem, wm * 128 + i * 16);
        fb0[i] = frag(smem + SLAB, wn * 128 + i * 16);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int rd = 1, wr = 4;
    auto slab = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[8], bf16x8 (&na)[8], bf16x8 (&nb)[8]) __attribute__((always_inline)) {
        const char* As = smem + rd * STAGE;
        const char* Bs = As + SLAB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // weight fragment as the MFMA A operand: a lane ends up with 4 consecutive output columns of one row
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[j]), "v"(fa[i]));
#pragma unroll
                for (int v = 0; v < NV; ++v)      // eight independent chains: no dependent-issue stalls among them
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dummy[(j * NV + v) & 7]) : "v"(dmul));
                const int n = 8 * i + j;
                if ((n & 3) == 0 && n < 64) {
                    const int m = n >> 2;           // 16 reads: A row groups 0..7, then W row groups 0..7
                    if (m < 8) na[m] = frag(As, wm * 128 + m * 16);
                    else nb[m - 8] = frag(Bs, wn * 128 + (m - 8) * 16);
                } else if ((n & 7) == 2) {
                    stage_piece(wr, n >> 3);
                }
            }
        }
        slab_issued();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd = rd + 1 == RING ? 0 : rd + 1;
        wr = wr + 1 == RING ? 0 : wr + 1;
    };
    for (int it = 0; it < ((ns + 1) >> 1); ++it) {
        slab(fa0, fb0, fa1, fb1);
        slab(fa1, fb1, fa0, fb0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    if (p.no_store) {
        float keep = dummy[0] + dummy[1] + dummy[2] + dummy[3] + dummy[4] + dummy[5] + dummy[6] + dummy[7] == 3.21e-30f ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 1.2345e-30f) p.C[0] = (bf16_t)keep;
        return;
    }
    // D[n = 4 g + r][m = l15]: lane writes 4 consecutive columns of row m (plain 8-byte stores: not the whole-line
    // write-out of the library; mode 1 leaves it out)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = bm + wm * 128 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = bn + wn * 128 + j * 16 + 4 * g;
            if (m < p.M && n < p.N) {
                const bf16x4 o = {(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
                *reinterpret_cast<bf16x4*>(p.C + (long)m * p.ldc + n) = o;
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main() {
    constexpr int LDS = 5 * 2 * 256 * 64;
    CK(hipFuncSetAttribute((const void*)nt_w4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)nt_w4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)nt_w4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)nt_w4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)nt_w4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int shapes[][3] = {{65536, 768, 3072}, {65536, 768, 768}, {65536, 3072, 768}, {65536, 2304, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 5e-4f; };
        for (auto& v : ha) v = f2bf(rnd());
        for (auto& v : hw) v = f2bf(rnd());
        void *dA, *dW, *dC, *dC2;
        CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dW, hw.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dC2, (size_t)M * N * 2));
        CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        Args a{};
        a.A = (const bf16_t*)dA; a.W = (const bf16_t*)dW; a.C = (bf16_t*)dC; a.M = M; a.N = N; a.K = K;
        a.lda = K; a.ldb = K; a.ldc = N; a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 255) / 256; a.nwg = a.tiles_m * a.tiles_n;
        vitk_gemm_epilogue_args ea{};
        auto lib = [&]() { int rc = vitk_gemm_nt(dA, dW, dC2, M, N, K, K, K, N, VITK_EPI_NONE, &ea, VITK_BF16, VITK_BF16, nullptr);
                           if (rc) { std::printf("vitk_gemm_nt: %d\n", rc); std::exit(1); } };
        auto mine = [&](int no_store) { a.no_store = no_store; hipLaunchKernelGGL(nt_w4_kernel<0>, dim3(a.nwg), dim3(256), LDS, 0, a); };
        lib(); mine(0);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> c1((size_t)M * N), c2((size_t)M * N);
        CK(hipMemcpy(c1.data(), dC, c1.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c2.data(), dC2, c2.size() * 2, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> t[3];
        for (int rnd_ = 0; rnd_ < 5; ++rnd_)
            for (int which : {0, 1, 2}) {
                for (int w = 0; w < 2; ++w) { if (which == 0) lib(); else mine(which - 1); }
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < 10; ++r) { if (which == 0) lib(); else mine(which - 1); }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t[which].push_back(ms / 10);
            }
        const double fl = 2.0 * M * N * K;
        for (auto& v : t) std::sort(v.begin(), v.end());
        std::printf("M=%d N=%d K=%d: elements that differ from the library's: %zu of %zu; library (eight waves, persistent, whole-line write-out) "
                    "%.3f ms %.0f TF/s; one wave per SIMD %.3f ms %.0f TF/s; the same without its stores %.3f ms %.0f TF/s\n",
                    M, N, K, diff, c1.size(), t[0][2], fl / t[0][2] / 1e9, t[1][2], fl / t[1][2] / 1e9, t[2][2], fl / t[2][2] / 1e9);
        // VALU instructions behind every MFMA (stores left out): what a lone wave can hide under its MFMAs
        auto with_valu = [&](int nv) {
            a.no_store = 1;
            switch (nv) {
                case 0: hipLaunchKernelGGL(nt_w4_kernel<0>, dim3(a.nwg), dim3(256), LDS, 0, a); break;
                case 1: hipLaunchKernelGGL(nt_w4_kernel<1>, dim3(a.nwg), dim3(256), LDS, 0, a); break;
                case 2: hipLaunchKernelGGL(nt_w4_kernel<2>, dim3(a.nwg), dim3(256), LDS, 0, a); break;
                case 3: hipLaunchKernelGGL(nt_w4_kernel<3>, dim3(a.nwg), dim3(256), LDS, 0, a); break;
                default: hipLaunchKernelGGL(nt_w4_kernel<4>, dim3(a.nwg), dim3(256), LDS, 0, a); break;
            }
        };
        std::printf("    VALU instructions behind every MFMA:");
        for (int nv = 0; nv <= 4; ++nv) {
            std::vector<float> tv;
            for (int rnd_ = 0; rnd_ < 3; ++rnd_) {
                with_valu(nv); with_valu(nv);
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < 10; ++r) with_valu(nv);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                tv.push_back(ms / 10);
            }
            std::sort(tv.begin(), tv.end());
            std::printf("  %d: %.3f ms", nv, tv[1]);
        }
        std::printf("\n");
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dC2));
    }
    return 0;
}
