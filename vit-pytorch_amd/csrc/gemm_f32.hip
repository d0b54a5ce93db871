// fp32 GEMM on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32): the parity-check
// instance of the linear layers.  Operands are addressed through (row, k) strides, so the
// same kernel serves  C = X.W^T (+epilogue)  and the weight gradient  C += dY^T.X.
#include "common.h"

namespace {

constexpr int FB = 64;      // tile edge
constexpr int FBK = 16;
constexpr int FLD = 17;     // padded LDS row (floats)

struct F32Args {
    const float* X; const float* W; float* C;
    int M, N, K;            // C is [M,N]; reduction length K
    long sxm, sxk;          // X(m,k) = X[m*sxm + k*sxk]
    long swn, swk;          // W(n,k) = W[n*swn + k*swk]
    long ldc;
    int k_per_split;        // reduction elements per blockIdx.z slice (multiple of FBK)
    GemmEpi epi;
};

template <int EPI, bool ACC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32Args p) {
    __shared__ float Xs[FB * FLD];
    __shared__ float Ws[FB * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int bm = blockIdx.y * FB, bn = blockIdx.x * FB;
    const int k0 = blockIdx.z * p.k_per_split;
    int k1 = k0 + p.k_per_split;
    k1 = k1 < p.K ? k1 : p.K;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool x_kfast = (p.sxk == 1), w_kfast = (p.swk == 1);
    for (int kb = k0; kb < k1; kb += FBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            {
                const int row = x_kfast ? (e >> 4) : (e & 63);
                const int kk = x_kfast ? (e & 15) : (e >> 6);
                const int m = bm + row, k = kb + kk;
                Xs[row * FLD + kk] = (m < p.M && k < k1) ? p.X[(long)m * p.sxm + (long)k * p.sxk] : 0.f;
            }
            {
                const int row = w_kfast ? (e >> 4) : (e & 63);
                const int kk = w_kfast ? (e & 15) : (e >> 6);
                const int n = bn + row, k = kb + kk;
                Ws[row * FLD + kk] = (n < p.N && k < k1) ? p.W[(long)n * p.swn + (long)k * p.swk] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FBK / 4; ++ks) {
            float xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = Xs[(wm * 32 + i * 16 + l15) * FLD + 4 * ks + g];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = Ws[(wn * 32 + j * 16 + l15) * FLD + 4 * ks + g];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // D[r = 4g+reg <-> n][c = l15 <-> m]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = bm + wm * 32 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = bn + wn * 32 + j * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if constexpr (ACC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(p.C + (long)m * p.ldc + n + r, v[r]);
            } else {
                gemm_epilogue<EPI, float, float, 4>(p.epi, v, m, n, p.C, p.ldc);
            }
        }
    }
}

template <int EPI>
int launch_f32(const F32Args& a, hipStream_t st) {
    dim3 grid(vitk_ceil_div(a.N, FB), vitk_ceil_div(a.M, FB), 1);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, false>), grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int vitk_gemm_nt_f32(const void* A, const void* B, void* C, int M, int N, int K, long lda,
                     long ldb, long ldc, int epilogue, const GemmEpi& epi, hipStream_t st) {
    if (N % 4 || ldc % 4) return VITK_ERR_SHAPE;
    F32Args a;
    a.X = (const float*)A; a.W = (const float*)B; a.C = (float*)C;
    a.M = M; a.N = N; a.K = K;
    a.sxm = lda; a.sxk = 1; a.swn = ldb; a.swk = 1; a.ldc = ldc;
    a.k_per_split = vitk_ceil_div(K, FBK) * FBK;
    a.epi = epi;
    switch (epilogue) {
        case VITK_EPI_NONE: return launch_f32<VITK_EPI_NONE>(a, st);
        case VITK_EPI_BIAS: return launch_f32<VITK_EPI_BIAS>(a, st);
        case VITK_EPI_BIAS_GELU: return launch_f32<VITK_EPI_BIAS_GELU>(a, st);
        case VITK_EPI_BIAS_RES: return launch_f32<VITK_EPI_BIAS_RES>(a, st);
        case VITK_EPI_DGELU: return launch_f32<VITK_EPI_DGELU>(a, st);
        case VITK_EPI_PATCH: return launch_f32<VITK_EPI_PATCH>(a, st);
    }
    return VITK_ERR_UNSUPPORTED;
}

// C[M,N] += A[T,M]^T . B[T,N]:  "X side" rows are the columns of A, "W side" rows the columns of B.
int vitk_gemm_tn_f32(const void* A, const void* B, float* C, int M, int N, int T, long lda,
                     long ldb, long ldc, int split_k, hipStream_t st) {
    if (N % 4) return VITK_ERR_SHAPE;
    F32Args a;
    a.X = (const float*)A; a.W = (const float*)B; a.C = C;
    a.M = M; a.N = N; a.K = T;
    a.sxm = 1; a.sxk = lda; a.swn = 1; a.swk = ldb; a.ldc = ldc;
    const int tiles = vitk_ceil_div(N, FB) * vitk_ceil_div(M, FB);
    const int nkt = vitk_ceil_div(T, FBK);
    if (split_k <= 0) {
        split_k = vitk_ceil_div(1024, tiles);
        const int cap = nkt / 8 > 0 ? nkt / 8 : 1;
        if (split_k > cap) split_k = cap;
    }
    if (split_k > nkt) split_k = nkt;
    a.k_per_split = vitk_ceil_div(nkt, split_k) * FBK;
    split_k = vitk_ceil_div(T, a.k_per_split);
    a.epi = GemmEpi{};
    dim3 grid(vitk_ceil_div(N, FB), vitk_ceil_div(M, FB), split_k);
    hipLaunchKernelGGL((gemm_f32_kernel<VITK_EPI_NONE, true>), grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
