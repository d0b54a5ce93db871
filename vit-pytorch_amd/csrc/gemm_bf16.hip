// bf16 MFMA GEMMs for the linear layers of the ViT block (gfx950).
//
//   gemm_nt:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)      forward projections and dX
//   gemm_tn:  C[M,N] += A[T,M]^T . B[T,N]  (fp32 C)     weight gradients, split over T
//
// Both use 128x128 output tiles, 4 wavefronts (2x2, 64x64 each) and
// v_mfma_f32_16x16x32_bf16 with the WEIGHT-side fragment as the MFMA "A" operand, so that
// every lane ends up holding consecutive output columns of one output row (vector stores).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;

// XCD-aware workgroup remap: blocks b and b+8 share an XCD (and its L2), so give every XCD a
// contiguous run of tile ids.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// Output column held by MFMA result row `rr` (0..15) of weight sub-tile j (0..3) of a wave:
// sub-tiles are paired so that one lane owns 8 consecutive columns across the pair.
__device__ __forceinline__ int ncol_of(int j, int rr) {
    return 32 * (j >> 1) + 8 * (rr >> 2) + (rr & 3) + 4 * (j & 1);
}

// ============================================================================ NT kernel
// LDS tile image: [128 rows][64 bf16] = 128-byte rows, 16-byte chunk c of row r stored at
// chunk position c ^ (r & 7): fragment reads (16 rows x one chunk per 16-lane group) are
// bank-conflict free, global reads are whole 128-byte lines.
constexpr int NT_BK = 64;
constexpr int NT_TILE_BYTES = 128 * 128;   // one operand, one stage

struct NtArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int tiles_n, nwg;
    GemmEpi epi;
};

__device__ __forceinline__ int nt_lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ (row & 7)) << 4);
}

template <int EPI, typename TC>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(NtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [stage][A|B][16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int wg = xcd_remap(blockIdx.x, p.nwg);
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int bm = tm * BM, bn = tn * BN;

    // ---- staging map: 4 chunks of A and 4 of B per thread and stage
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
    int lds_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 3, ch = c & 7;
        lds_dst[i] = nt_lds_off(row, ch);
        int am = bm + row; am = am < p.M ? am : p.M - 1;
        // LDS row `row` of the weight tile holds the weight row that MFMA result row
        // (row & 15) of sub-tile (row >> 4) & 3 of wave-column row >> 6 must produce.
        int bnrow = bn + (row & 64) + ncol_of((row >> 4) & 3, row & 15);
        bnrow = bnrow < p.N ? bnrow : p.N - 1;
        a_src[i] = p.A + (long)am * p.lda + ch * 8;
        b_src[i] = p.B + (long)bnrow * p.ldb + ch * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / NT_BK;
    u32x4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
        rb[i] = *reinterpret_cast<const u32x4*>(b_src[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(smem + lds_dst[i]) = ra[i];
        *reinterpret_cast<u32x4*>(smem + NT_TILE_BYTES + lds_dst[i]) = rb[i];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* As = smem + cur * (2 * NT_TILE_BYTES);
        const char* Bs = As + NT_TILE_BYTES;
        const bool more = (kt + 1) < nk;
        if (more) {
            const int ko = (kt + 1) * NT_BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + ko);
                rb[i] = *reinterpret_cast<const u32x4*>(b_src[i] + ko);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xa[i] = *reinterpret_cast<const bf16x8*>(As + nt_lds_off(wm * 64 + i * 16 + l15, 4 * s + g));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wb[j] = *reinterpret_cast<const bf16x8*>(Bs + nt_lds_off(wn * 64 + j * 16 + l15, 4 * s + g));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* dst = smem + (cur ^ 1) * (2 * NT_TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(dst + lds_dst[i]) = ra[i];
                *reinterpret_cast<u32x4*>(dst + NT_TILE_BYTES + lds_dst[i]) = rb[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane owns row m, columns n0 + 32*jp + 8*g + 0..7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = bm + wm * 64 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int n = bn + wn * 64 + 32 * jp + 8 * g;
            if (n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][2 * jp][r];
                v[4 + r] = acc[i][2 * jp + 1][r];
            }
            gemm_epilogue<EPI, TC, bf16_t, 8>(p.epi, v, m, n, reinterpret_cast<TC*>(p.C), p.ldc);
        }
    }
}

// ============================================================================ TN kernel
// A:[T,M], B:[T,N] both row-major (reduction index T is the slow one).  LDS tile image:
// [32 t][128 cols] bf16 with 288-byte rows; MFMA fragments are fetched with
// ds_read_b64_tr_b16 (hardware transpose).  Within one MFMA the 32 reduction slots are
// (lane group gq, element e) <-> t = 4*gq + (e & 3) + 16*(e >> 2), identically for both
// operands; with 288-byte rows each 32-lane half then reads 8 consecutive rows x 4 slots:
// conflict free.
constexpr int TN_BK = 32;
constexpr int TN_ROW_BYTES = 288;
constexpr int TN_TILE_BYTES = TN_BK * TN_ROW_BYTES;   // 9216

struct TnArgs {
    const bf16_t* A; const bf16_t* B; float* C;
    int M, N, T;
    long lda, ldb, ldc;
    int tiles_n, ntiles, kt_per_split;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int col0, int lane) {
    // Fragment for MFMA rows/cols col0..col0+15 over the tile's 32 t-rows.
    const int l15 = lane & 15, gq = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const char* a0 = tile + (4 * gq + q) * TN_ROW_BYTES + (col0 + 4 * pp) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0 + 16 * TN_ROW_BYTES));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(TnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int tile = blockIdx.x % p.ntiles, split = blockIdx.x / p.ntiles;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int bm = tm * BM, bn = tn * BN;
    const int kt0 = split * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    const int nkt = (p.T + TN_BK - 1) / TN_BK;
    kt1 = kt1 < nkt ? kt1 : nkt;
    if (kt0 >= kt1) return;

    // staging: tile = 32 rows x 16 chunks (8 cols each); 2 chunks per thread per operand
    int srow[2], lds_dst[2];
    long a_col[2], b_col[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        srow[i] = c >> 4;
        const int ch = c & 15;
        lds_dst[i] = srow[i] * TN_ROW_BYTES + ch * 16;
        const int am = bm + ch * 8, bnn = bn + ch * 8;
        a_ok[i] = am < p.M; b_ok[i] = bnn < p.N;
        a_col[i] = a_ok[i] ? am : 0;
        b_col[i] = b_ok[i] ? bnn : 0;
    }
    auto fetch = [&](int kt, u32x4* ra, u32x4* rb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = kt * TN_BK + srow[i];
            const bool tok = t < p.T;
            const long tt = tok ? t : 0;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const u32x4 va = *reinterpret_cast<const u32x4*>(p.A + tt * p.lda + a_col[i]);
            const u32x4 vb = *reinterpret_cast<const u32x4*>(p.B + tt * p.ldb + b_col[i]);
            ra[i] = (tok && a_ok[i]) ? va : z;
            rb[i] = (tok && b_ok[i]) ? vb : z;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[2], rb[2];
    fetch(kt0, ra, rb);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<u32x4*>(smem + lds_dst[i]) = ra[i];
        *reinterpret_cast<u32x4*>(smem + TN_TILE_BYTES + lds_dst[i]) = rb[i];
    }
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const char* As = smem + cur * (2 * TN_TILE_BYTES);
        const char* Bs = As + TN_TILE_BYTES;
        const bool more = (kt + 1) < kt1;
        if (more) fetch(kt + 1, ra, rb);
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = tr_frag(As, wm * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = tr_frag(Bs, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (more) {
            char* dst = smem + (cur ^ 1) * (2 * TN_TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<u32x4*>(dst + lds_dst[i]) = ra[i];
                *reinterpret_cast<u32x4*>(dst + TN_TILE_BYTES + lds_dst[i]) = rb[i];
            }
        }
        __syncthreads();
    }

    // D[r][c]: r = 4*g + reg <-> A column (output row), c = l15 <-> B column (output col).
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = bn + wn * 64 + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm + wm * 64 + i * 16 + 4 * g + r;
                if (m < p.M && n < p.N) atomicAdd(p.C + (long)m * p.ldc + n, acc[i][j][r]);
            }
        }
}

template <int EPI, typename TC>
int launch_nt(const NtArgs& a, hipStream_t st) {
    const size_t lds = 4 * NT_TILE_BYTES;   // 2 stages x (A, B)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<EPI, TC>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI, TC>), dim3(a.nwg), dim3(256), lds, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int vitk_gemm_nt_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda,
                      long ldb, long ldc, int epilogue, const GemmEpi& epi, int out_dtype, hipStream_t st) {
    if (K % NT_BK || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return VITK_ERR_SHAPE;
    NtArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.tiles_n = vitk_ceil_div(N, BN);
    a.nwg = vitk_ceil_div(M, BM) * a.tiles_n;
    a.epi = epi;
    if (out_dtype == VITK_F32) {
        // fp32 output = the residual stream (and its producers)
        switch (epilogue) {
            case VITK_EPI_BIAS: return launch_nt<VITK_EPI_BIAS, float>(a, st);
            case VITK_EPI_BIAS_RES: return launch_nt<VITK_EPI_BIAS_RES, float>(a, st);
            case VITK_EPI_PATCH: return launch_nt<VITK_EPI_PATCH, float>(a, st);
        }
        return VITK_ERR_UNSUPPORTED;
    }
    if (out_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    switch (epilogue) {
        case VITK_EPI_NONE: return launch_nt<VITK_EPI_NONE, bf16_t>(a, st);
        case VITK_EPI_BIAS: return launch_nt<VITK_EPI_BIAS, bf16_t>(a, st);
        case VITK_EPI_BIAS_GELU: return launch_nt<VITK_EPI_BIAS_GELU, bf16_t>(a, st);
        case VITK_EPI_BIAS_RES: return launch_nt<VITK_EPI_BIAS_RES, bf16_t>(a, st);
        case VITK_EPI_DGELU: return launch_nt<VITK_EPI_DGELU, bf16_t>(a, st);
        case VITK_EPI_PATCH: return launch_nt<VITK_EPI_PATCH, bf16_t>(a, st);
    }
    return VITK_ERR_UNSUPPORTED;
}

int vitk_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int T, long lda,
                      long ldb, long ldc, int split_k, hipStream_t st) {
    if (M % 8 || N % 8 || lda % 8 || ldb % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return VITK_ERR_SHAPE;
    TnArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
    a.M = M; a.N = N; a.T = T; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.tiles_n = vitk_ceil_div(N, BN);
    a.ntiles = vitk_ceil_div(M, BM) * a.tiles_n;
    const int nkt = vitk_ceil_div(T, TN_BK);
    if (split_k <= 0) {
        // aim for ~3 workgroups per CU, but keep at least 8 reduction tiles per slice
        split_k = vitk_ceil_div(768, a.ntiles);
        const int cap = nkt / 8 > 0 ? nkt / 8 : 1;
        if (split_k > cap) split_k = cap;
    }
    if (split_k > nkt) split_k = nkt;
    a.kt_per_split = vitk_ceil_div(nkt, split_k);
    split_k = vitk_ceil_div(nkt, a.kt_per_split);
    const size_t lds = 4 * TN_TILE_BYTES;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(a.ntiles * split_k), dim3(256), lds, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
