// LayerNorm forward / backward: one wavefront per row, the row lives in registers (single
// pass over HBM), statistics in fp32 reduced with cross-lane shuffles.  Backward fuses the
// residual-branch gradient add and the gamma/beta gradient partial reduction.
#include "common.h"

#include <mutex>

namespace {

template <typename TI, typename TO, int IT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TO* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float inv_n = 1.0f / (float)cols;
    for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
        const TI* xr = x + (long)row * cols;
        float v[IT][4];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                Vec4<TI>::load(xr + c, v[it]);
            } else {
                v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
            }
            s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[it][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        TO* yr = y + (long)row * cols;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                float ga[4], be[4], o[4];
                Vec4<float>::load(gamma + c, ga);
                Vec4<float>::load(beta + c, be);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[it][j] - mean) * rstd * ga[j] + be[j];
                Vec4<TO>::store(yr + c, o);
            }
        }
        if (lane == 0) {
            mean_o[row] = mean;
            rstd_o[row] = rstd;
        }
    }
}

template <typename T, typename TDY, int IT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ part,   // [gridDim.x][2][cols]
                                                     int rows, int cols,
                                                     bf16_t* __restrict__ dz,    // optional: bf16(mask(dx) / keep)
                                                     uint32_t thr16, float dscale, uint64_t seed) {
    __shared__ float red[3][IT * 256 * 2];   // waves 1..3 park their partial sums here
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float inv_n = 1.0f / (float)cols;
    float ga[IT][4], pg[IT][4], pb[IT][4];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < cols) {
            Vec4<float>::load(gamma + c, ga[it]);
        } else {
            ga[it][0] = ga[it][1] = ga[it][2] = ga[it][3] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pg[it][j] = pb[it][j] = 0.f;
    }
    for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        const long off = (long)row * cols;
        float xh[IT][4], gy[IT][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                float xv[4], dv[4];
                Vec4<T>::load(x + off + c, xv);
                Vec4<TDY>::load(dy + off + c, dv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[it][j] = (xv[j] - mean) * rstd;
                    gy[it][j] = dv[j] * ga[it][j];
                    s1 += gy[it][j];
                    s2 += gy[it][j] * xh[it][j];
                    pg[it][j] += dv[j] * xh[it][j];
                    pb[it][j] += dv[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xh[it][j] = gy[it][j] = 0.f;
            }
        }
        const float c1 = wave_sum(s1) * inv_n, c2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < cols) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (gy[it][j] - c1 - xh[it][j] * c2);
                if (dres) {
                    float r[4];
                    Vec4<T>::load(dres + off + c, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += r[j];
                }
                Vec4<T>::store(dx + off + c, o);
                if (dz) {
                    // the same gradient as the GEMM operand of the layer that produced x: the dropout
                    // mask of that layer's output (element index row*cols + col) and the bf16 hand-over
                    if (thr16) {
                        bool keep[4];
                        drop_keep_vec<4>(seed, (uint64_t)off + c, thr16, keep);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = keep[j] ? o[j] * dscale : 0.f;
                    }
                    Vec4<bf16_t>::store(dz + off + c, o);
                }
            }
        }
    }
    // block-level reduction of the parameter-gradient partials; one partial row per block
    if (wid > 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[wid - 1][((it * 4 + j) * 64 + lane) * 2 + 0] = pg[it][j];
                red[wid - 1][((it * 4 + j) * 64 + lane) * 2 + 1] = pb[it][j];
            }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = pg[it][j], b = pb[it][j];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    a += red[w][((it * 4 + j) * 64 + lane) * 2 + 0];
                    b += red[w][((it * 4 + j) * 64 + lane) * 2 + 1];
                }
                if (c < cols) {
                    part[((long)blockIdx.x * 2 + 0) * cols + c + j] = a;
                    part[((long)blockIdx.x * 2 + 1) * cols + c + j] = b;
                }
            }
        }
    }
}

// dgamma[c] += sum_b part[b][0][c]; dbeta[c] += sum_b part[b][1][c].
// Block = 64 columns x 4 row groups; the row groups are combined through LDS.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nblk, int cols) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + cl;            // index into the 2*cols (gamma | beta) columns
    // blockIdx.y owns partial rows [b0, b1); its four row groups stride through them.
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = (b0 + per) < nblk ? (b0 + per) : nblk;
    float s0 = 0.f, s1 = 0.f;
    if (t < 2 * cols) {
        const int which = t / cols, c = t - which * cols;
        int b = b0 + rg;
        for (; b + 4 < b1; b += 8) {
            s0 += part[((long)b * 2 + which) * cols + c];
            s1 += part[((long)(b + 4) * 2 + which) * cols + c];
        }
        for (; b < b1; b += 4) s0 += part[((long)b * 2 + which) * cols + c];
    }
    red[rg][cl] = s0 + s1;
    __syncthreads();
    if (rg == 0 && t < 2 * cols) {
        const int which = t / cols, c = t - which * cols;
        float* dst = which ? dbeta : dgamma;
        atomicAdd(dst + c, (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
    }
}

constexpr int LN_BWD_MAX_BLOCKS = 1024;
inline int ln_bwd_grid(int rows) {
    int grid = vitk_ceil_div(rows, 4);
    return grid > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : grid;
}

template <typename TI, typename TO>
int ln_fwd_dispatch(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                    float* rstd, int rows, int cols, float eps, hipStream_t st) {
    const int it = vitk_ceil_div(cols, 256);
    int grid = vitk_ceil_div(rows, 4);
    if (grid > 4096) grid = 4096;
#define LN_FWD(IT)                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, IT>), dim3(grid), dim3(256), 0, st, (const TI*)x, \
                       gamma, beta, (TO*)y, mean, rstd, rows, cols, eps)
    if (it <= 1) LN_FWD(1);
    else if (it <= 2) LN_FWD(2);
    else if (it <= 3) LN_FWD(3);
    else if (it <= 4) LN_FWD(4);
    else if (it <= 8) LN_FWD(8);
    else if (it <= 16) LN_FWD(16);
    else if (it <= 32) LN_FWD(32);
    else return VITK_ERR_SHAPE;
#undef LN_FWD
    VITK_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename TDY>
int ln_bwd_dispatch(const void* dy, const void* x, const float* gamma, const float* mean,
                    const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                    float* workspace, int rows, int cols, hipStream_t st, void* dz = nullptr,
                    float dropout_p = 0.f, uint64_t seed = 0) {
    const int it = vitk_ceil_div(cols, 256);
    const int grid = ln_bwd_grid(rows);
    const uint32_t thr = vitk_drop_thr16(dropout_p);
    const float ds = vitk_drop_scale(thr);
#define LN_BWD(IT)                                                                               \
    hipLaunchKernelGGL((ln_bwd_kernel<T, TDY, IT>), dim3(grid), dim3(256), 0, st, (const TDY*)dy, \
                       (const T*)x, gamma, mean, rstd, (const T*)dres, (T*)dx, workspace, rows, cols, \
                       (bf16_t*)dz, thr, ds, seed)
    if (it <= 1) LN_BWD(1);
    else if (it <= 2) LN_BWD(2);
    else if (it <= 3) LN_BWD(3);
    else if (it <= 4) LN_BWD(4);
    else if (it <= 8) LN_BWD(8);
    else return VITK_ERR_SHAPE;
#undef LN_BWD
    VITK_CHECK_LAUNCH();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(vitk_ceil_div(2 * cols, 64), vitk_ceil_div(grid, 32)), dim3(256), 0,
                       st, workspace, dgamma, dbeta, grid, cols);
    VITK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int vitk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int rows, int cols, float eps,
                                  int dtype, int out_dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return VITK_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16 && out_dtype == VITK_BF16)
        return ln_fwd_dispatch<bf16_t, bf16_t>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
    if (dtype == VITK_BF16 && out_dtype == VITK_F32)
        return ln_fwd_dispatch<bf16_t, float>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
    if (dtype == VITK_F32 && out_dtype == VITK_F32)
        return ln_fwd_dispatch<float, float>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
    if (dtype == VITK_F32 && out_dtype == VITK_BF16)
        return ln_fwd_dispatch<float, bf16_t>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
    return VITK_ERR_DTYPE;
}

extern "C" size_t vitk_layernorm_bwd_workspace(int rows, int cols) {
    return (size_t)ln_bwd_grid(rows) * 2 * (size_t)cols * sizeof(float);
}

extern "C" int vitk_layernorm_bwd(const void* dy, const void* x, const float* gamma,
                                  const float* mean, const float* rstd, const void* dres, void* dx,
                                  float* dgamma, float* dbeta, void* workspace, int rows, int cols,
                                  int dtype, int dy_dtype, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return VITK_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4 || cols > 2048) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
#define LN_BWD_CASE(T, TDY) \
    return ln_bwd_dispatch<T, TDY>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, ws, rows, cols, st)
    if (dtype == VITK_BF16 && dy_dtype == VITK_BF16) LN_BWD_CASE(bf16_t, bf16_t);
    if (dtype == VITK_BF16 && dy_dtype == VITK_F32) LN_BWD_CASE(bf16_t, float);
    if (dtype == VITK_F32 && dy_dtype == VITK_F32) LN_BWD_CASE(float, float);
    if (dtype == VITK_F32 && dy_dtype == VITK_BF16) LN_BWD_CASE(float, bf16_t);
#undef LN_BWD_CASE
    return VITK_ERR_DTYPE;
}

// ---------------------------------------------------------------------------- no-allocation variant
// The per-block dgamma/dbeta partials of ln_bwd_kernel live in a file-scope device array instead of
// a caller-supplied workspace: 1024 blocks x 2 x 1024 columns x 4 B = 8 MiB per GPU, held until the
// process ends.  In host code `ln_static_scratch` is only a stand-in for the symbol; its device
// address is looked up once per device and that pointer is what the kernels receive.
namespace {
constexpr int LN_STATIC_MAX_COLS = 1024;
constexpr int LN_STATIC_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) float ln_static_scratch[(size_t)LN_BWD_MAX_BLOCKS * 2 * LN_STATIC_MAX_COLS];
static float* g_ln_static_ptr[LN_STATIC_MAX_DEVICES] = {};
static std::mutex g_ln_static_mu;

// Device address of this GPU's copy of the array (every GPU holds its own at its own address).
static int ln_static_scratch_ptr(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= LN_STATIC_MAX_DEVICES) return VITK_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_ln_static_mu);
    if (!g_ln_static_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(ln_static_scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return VITK_ERR_NULL;
        g_ln_static_ptr[dev] = (float*)p;
    }
    *out = g_ln_static_ptr[dev];
    return 0;
}

extern "C" int vitk_layernorm_bwd_static(const void* dy, const void* x, const float* gamma,
                                         const float* mean, const float* rstd, const void* dres, void* dx,
                                         float* dgamma, float* dbeta, int rows, int cols,
                                         int dtype, int dy_dtype, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) return VITK_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4) return VITK_ERR_SHAPE;
    if (cols > LN_STATIC_MAX_COLS) return VITK_ERR_SCRATCH;      // beyond the built-in scratch array
    float* ws = nullptr;
    const int rc = ln_static_scratch_ptr(&ws);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    // Every partial row the reduction reads is written by ln_bwd_kernel first, so nothing in the
    // array has to start at zero; two calls in flight on one GPU would share it and race.
#define LN_BWD_CASE(T, TDY) \
    return ln_bwd_dispatch<T, TDY>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, ws, rows, cols, st)
    if (dtype == VITK_BF16 && dy_dtype == VITK_BF16) LN_BWD_CASE(bf16_t, bf16_t);
    if (dtype == VITK_BF16 && dy_dtype == VITK_F32) LN_BWD_CASE(bf16_t, float);
    if (dtype == VITK_F32 && dy_dtype == VITK_F32) LN_BWD_CASE(float, float);
    if (dtype == VITK_F32 && dy_dtype == VITK_BF16) LN_BWD_CASE(float, bf16_t);
#undef LN_BWD_CASE
    return VITK_ERR_DTYPE;
}

extern "C" int vitk_layernorm_bwd_dz(const void* dy, const void* x, const float* gamma,
                                     const float* mean, const float* rstd, const void* dres, void* dx,
                                     void* dz, float dropout_p, uint64_t seed,
                                     float* dgamma, float* dbeta, void* workspace, int rows, int cols,
                                     int dtype, int dy_dtype, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dz || !dgamma || !dbeta || !workspace) return VITK_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4 || cols > 2048) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
#define LN_BWD_CASE(T, TDY)                                                                                  \
    return ln_bwd_dispatch<T, TDY>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, ws, rows, cols, st, dz, \
                                   dropout_p, seed)
    if (dtype == VITK_F32 && dy_dtype == VITK_F32) LN_BWD_CASE(float, float);
    if (dtype == VITK_F32 && dy_dtype == VITK_BF16) LN_BWD_CASE(float, bf16_t);
#undef LN_BWD_CASE
    return VITK_ERR_DTYPE;
}
