// HBM-bound helpers around the GEMMs: im2col for the patch embedding, class-token row,
// token-assembly backward, weight cast (+transpose), dropout backward, dtype convert and the
// bias-gradient column sum.  All accesses are 8- or 16-byte vectors per lane.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------ im2col
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols,
                                                     int B, int C, int Himg, int Wimg, int P, long total4) {
    const int gh = Himg / P, gw = Wimg / P;
    const int Kc = C * P * P;
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * 256) {
        const long e = e4 * 4;
        const long rowm = e / Kc;
        const int col = (int)(e - rowm * Kc);
        const int c = col / (P * P), rem = col - c * P * P;
        const int i = rem / P, j = rem - i * P;
        const int b = (int)(rowm / (gh * gw)), pr = (int)(rowm - (long)b * gh * gw);
        const int py = pr / gw, px = pr - py * gw;
        float v[4];
        Vec4<float>::load(img + (((long)b * C + c) * Himg + py * P + i) * Wimg + px * P + j, v);
        Vec4<T>::store(cols + e, v);
    }
}

// ------------------------------------------------------------------------------ class-token row
template <typename T>
__global__ __launch_bounds__(256) void embed_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                        T* __restrict__ out, int B, int ntok, int D,
                                                        uint32_t thr16, float dscale, uint64_t seed) {
    const int d4 = D / 4;
    const long total = (long)B * d4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int b = (int)(t / d4), d = (int)(t - (long)b * d4) * 4;
        float c[4], pe[4], v[4];
        Vec4<float>::load(cls + d, c);
        Vec4<float>::load(pos + d, pe);
        const long orow = (long)b * ntok;
        bool keep[4] = {true, true, true, true};
        if (thr16) drop_keep_vec<4>(seed, (uint64_t)orow * D + d, thr16, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = c[i] + pe[i];
            if (thr16) v[i] = keep[i] ? v[i] * dscale : 0.f;
        }
        Vec4<T>::store(out + orow * D + d, v);
    }
}

// ------------------------------------------------------------------------------ token assembly backward
template <typename TE, typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const TE* __restrict__ dE, T* __restrict__ dtok,
                                                        float* __restrict__ dpos, float* __restrict__ dcls,
                                                        int B, int ntok, int D, uint32_t thr16, float dscale,
                                                        uint64_t seed) {
    const int d4 = D / 4;
    const long total = (long)ntok * d4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int tok = (int)(t / d4), d = (int)(t - (long)tok * d4) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = blockIdx.y; b < B; b += gridDim.y) {     // batch slices run in parallel, partial sums meet in atomics
        const long row = (long)b * ntok + tok;
        float v[4];
        Vec4<TE>::load(dE + row * D + d, v);
        if (thr16) {
            bool keep[4];
            drop_keep_vec<4>(seed, (uint64_t)row * D + d, thr16, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = keep[i] ? v[i] * dscale : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += v[i];
        if (tok > 0) Vec4<T>::store(dtok + ((long)b * (ntok - 1) + tok - 1) * D + d, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(dpos + (long)tok * D + d + i, acc[i]);
    if (tok == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(dcls + d + i, acc[i]);
    }
}

// ------------------------------------------------------------------------------ weight cast + transpose
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ w, T* __restrict__ out,
                                                          T* __restrict__ out_t, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = w[(long)r * C + c];
            if (out) out[(long)r * C + c] = (T)v;
        }
        tile[ty + 8 * k][tx] = v;
    }
    if (!out_t) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < R && c < C) out_t[(long)c * R + r] = (T)tile[tx][ty + 8 * k];
    }
}

// Many weights in one launch: what the optimizer step leaves behind for the next forward pass.  Entry e owns
// 64 x 64 tiles [tile0, next tile0) of its [R, C] weight, numbered row-major; the table travels as a kernel
// argument (no upload, no table in device memory).
constexpr int CAST_BATCH = 64;
struct CastTable {
    vitk_cast_entry e[CAST_BATCH];
    int n;
};
template <typename T> struct Pack4;
template <> struct Pack4<bf16_t> { typedef bf16x4 type; };
template <> struct Pack4<float> { typedef f32x4 type; };

template <typename T>
__global__ __launch_bounds__(256) void cast_weights_batched_kernel(const CastTable tab) {
    __shared__ float tile[64][65];
    int lo = 0, hi = tab.n - 1;                        // last entry with tile0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.e[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const vitk_cast_entry en = tab.e[lo];
    const int R = en.rows, C = en.cols;
    const int tiles_c = (C + 63) >> 6;
    const int t = blockIdx.x - en.tile0;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const float* __restrict__ w = (const float*)en.w;
    T* __restrict__ out = (T*)en.out;
    T* __restrict__ out_t = (T*)en.out_t;
    typedef typename Pack4<T>::type P4;
    // four elements per lane (16-byte reads, 8-byte bf16 writes) when rows, columns and the transposed row stride
    // allow it and the tile is whole; one element per lane otherwise
    const bool vec = !(R & 3) && !(C & 3) && !(en.ld_t & 3) && r0 + 64 <= R && c0 + 64 <= C &&
                     !(((uintptr_t)en.w | (uintptr_t)en.out | (uintptr_t)en.out_t) & 15);
    if (vec) {
        const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;     // 16 lanes x 4 columns per row, 16 rows per pass
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = rr + 16 * k;
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + (long)(r0 + r) * C + c0 + 4 * q);
            if (out) *reinterpret_cast<P4*>(out + (long)(r0 + r) * C + c0 + 4 * q) = P4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[r][4 * q + i] = v[i];
        }
        if (!out_t) return;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = rr + 16 * k;                             // transposed row; lane q owns source rows 4q..4q+3
            *reinterpret_cast<P4*>(out_t + (long)(c0 + c) * en.ld_t + r0 + 4 * q) =
                P4{(T)tile[4 * q][c], (T)tile[4 * q + 1][c], (T)tile[4 * q + 2][c], (T)tile[4 * q + 3][c]};
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = w[(long)r * C + c];
            if (out) out[(long)r * C + c] = (T)v;
        }
        tile[ty + 4 * k][tx] = v;
    }
    if (!out_t) return;
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < R && c < C) out_t[(long)c * en.ld_t + r] = (T)tile[tx][ty + 4 * k];
    }
}

// ------------------------------------------------------------------------------ dropout backward
template <typename TS, typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const TS* __restrict__ dy, T* __restrict__ dx,
                                                          long total4, uint32_t thr16, float dscale, uint64_t seed) {
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * 256) {
        float v[4];
        bool keep[4];
        Vec4<TS>::load(dy + e4 * 4, v);
        drop_keep_vec<4>(seed, (uint64_t)e4 * 4, thr16, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = keep[i] ? v[i] * dscale : 0.f;
        Vec4<T>::store(dx + e4 * 4, v);
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long total4) {
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * 256) {
        float v[4];
        Vec4<TS>::load(src + e4 * 4, v);
        Vec4<TD>::store(dst + e4 * 4, v);
    }
}

// ------------------------------------------------------------------------------ stochastic depth
// out = h + scale[sample] * y : a residual branch kept or dropped per sample (Swin's DropPath,
// reference models/Swin/swin_RKR.py:340-341) added to the fp32 stream in one pass; and its backward
// for the branch, dy = scale[sample] * dout (the stream's own gradient is dout itself).
template <typename T>
__global__ __launch_bounds__(256) void scale_add_kernel(const float* __restrict__ h, const T* __restrict__ y,
                                                        const float* __restrict__ scale, float* __restrict__ out,
                                                        long total4, long per_sample4) {
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * 256) {
        const float s = scale[e4 / per_sample4];
        float a[4], b[4];
        Vec4<float>::load(h + e4 * 4, a);
        Vec4<T>::load(y + e4 * 4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += s * b[i];
        Vec4<float>::store(out + e4 * 4, a);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ dout, const float* __restrict__ scale,
                                                         T* __restrict__ dy, long total4, long per_sample4) {
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * 256) {
        const float s = scale[e4 / per_sample4];
        float a[4];
        Vec4<float>::load(dout + e4 * 4, a);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] *= s;
        Vec4<T>::store(dy + e4 * 4, a);
    }
}

// ------------------------------------------------------------------------------ RKR weight fold
// W_t[o][i] = (W[o][i] + sum_k RM[k][o] LM[i][k]) * F[o],  b_t[o] = b[o] * F[o]   (reference RG_FC + SFG_FC,
// models/ViT/modeling_RKR.py:56-78, applied on the weight side), and the gradients of W, b, LM, RM, F
// from dW_t, db_t.  Weight-sized fp32 work: one launch forward, two backward, in place of a dozen
// tensor ops per layer.  K <= 8, I % 4 == 0.
constexpr int RKR_MAXK = 8;
__global__ __launch_bounds__(256) void rkr_fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                           const float* __restrict__ LM, const float* __restrict__ RM,
                                                           const float* __restrict__ F, float* __restrict__ Wt,
                                                           float* __restrict__ bt, int O, int I, int K) {
    const int o = blockIdx.x;
    float rm[RKR_MAXK];
#pragma unroll
    for (int k = 0; k < RKR_MAXK; ++k) rm[k] = k < K ? RM[(long)k * O + o] : 0.f;
    const float f = F ? F[o] : 1.0f;
    for (int i4 = threadIdx.x; i4 < I / 4; i4 += 256) {
        float w[4];
        Vec4<float>::load(W + (long)o * I + 4 * i4, w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r = 0.f;
            for (int k = 0; k < K; ++k) r += rm[k] * LM[(long)(4 * i4 + e) * K + k];
            w[e] = (w[e] + r) * f;
        }
        Vec4<float>::store(Wt + (long)o * I + 4 * i4, w);
    }
    if (threadIdx.x == 0) bt[o] = b[o] * f;
}

// One workgroup per output row: dW = dW_t * F, dRM[k][o], dF[o], db[o] (row reductions).
__global__ __launch_bounds__(256) void rkr_fold_bwd_row_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                               const float* __restrict__ LM, const float* __restrict__ RM,
                                                               const float* __restrict__ F, const float* __restrict__ dWt,
                                                               const float* __restrict__ dbt, float* __restrict__ dW,
                                                               float* __restrict__ db, float* __restrict__ dRM,
                                                               float* __restrict__ dF, int O, int I, int K) {
    __shared__ float red[4][RKR_MAXK + 1];
    const int o = blockIdx.x;
    float rm[RKR_MAXK], sum[RKR_MAXK + 1];
#pragma unroll
    for (int k = 0; k < RKR_MAXK; ++k) { rm[k] = k < K ? RM[(long)k * O + o] : 0.f; sum[k] = 0.f; }
    sum[RKR_MAXK] = 0.f;
    const float f = F ? F[o] : 1.0f;
    for (int i4 = threadIdx.x; i4 < I / 4; i4 += 256) {
        float w[4], g[4];
        Vec4<float>::load(W + (long)o * I + 4 * i4, w);
        Vec4<float>::load(dWt + (long)o * I + 4 * i4, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r = 0.f;
            const float gf = g[e] * f;
#pragma unroll
            for (int k = 0; k < RKR_MAXK; ++k)
                if (k < K) {
                    const float lm = LM[(long)(4 * i4 + e) * K + k];
                    r += rm[k] * lm;
                    sum[k] += gf * lm;
                }
            sum[RKR_MAXK] += g[e] * (w[e] + r);
            g[e] = gf;
        }
        if (dW) Vec4<float>::store(dW + (long)o * I + 4 * i4, g);
    }
#pragma unroll
    for (int k = 0; k <= RKR_MAXK; ++k) sum[k] = wave_sum(sum[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k <= RKR_MAXK; ++k) red[threadIdx.x >> 6][k] = sum[k];
    __syncthreads();
    if (threadIdx.x <= RKR_MAXK) {
        const int k = threadIdx.x;
        const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        if (k < K) dRM[(long)k * O + o] = v;
        if (k == RKR_MAXK) {
            if (dF) dF[o] = v + dbt[o] * b[o];
            db[o] = dbt[o] * f;
        }
    }
}

// dLM[i][k] += sum over a slice of rows o of dW_t[o][i] F[o] RM[k][o] (one thread per column i).
__global__ __launch_bounds__(256) void rkr_fold_bwd_col_kernel(const float* __restrict__ RM, const float* __restrict__ F,
                                                               const float* __restrict__ dWt, float* __restrict__ dLM, int O,
                                                               int I, int K, int rows_per_slice) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * rows_per_slice;
    int o1 = o0 + rows_per_slice;
    o1 = o1 < O ? o1 : O;
    if (i >= I) return;
    float acc[RKR_MAXK];
#pragma unroll
    for (int k = 0; k < RKR_MAXK; ++k) acc[k] = 0.f;
    for (int o = o0; o < o1; ++o) {
        const float g = dWt[(long)o * I + i] * (F ? F[o] : 1.0f);
#pragma unroll
        for (int k = 0; k < RKR_MAXK; ++k)
            if (k < K) acc[k] += g * RM[(long)k * O + o];
    }
#pragma unroll
    for (int k = 0; k < RKR_MAXK; ++k)
        if (k < K) atomicAdd(dLM + (long)i * K + k, acc[k]);
}

// ------------------------------------------------------------------------------ piggyback mask fold
// out = hard(m) * w with hard = binarizer (1 above the threshold, else 0) or ternarizer (-1 below zero, 1 above the
// threshold, else 0); the thresholding passes gradients through unchanged: dm = g * w, dw = g * hard(m).
__device__ __forceinline__ float mask_hard(float m, int ternary, float thr) {
    return m > thr ? 1.0f : ((ternary && m < 0.0f) ? -1.0f : 0.0f);
}
__global__ __launch_bounds__(256) void mask_fold_fwd_kernel(const float* __restrict__ m, const float* __restrict__ w,
                                                            float* __restrict__ out, long n, int ternary, float thr) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float a[4], b[4];
        Vec4<float>::load(m + 4 * i, a);
        Vec4<float>::load(w + 4 * i, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] *= mask_hard(a[e], ternary, thr);
        Vec4<float>::store(out + 4 * i, b);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        out[i] = mask_hard(m[i], ternary, thr) * w[i];
    }
}
__global__ __launch_bounds__(256) void mask_fold_bwd_kernel(const float* __restrict__ m, const float* __restrict__ w,
                                                            const float* __restrict__ g, float* __restrict__ dm,
                                                            float* __restrict__ dw, long n, int ternary, float thr) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float a[4], b[4], c[4];
        Vec4<float>::load(m + 4 * i, a);
        Vec4<float>::load(w + 4 * i, b);
        Vec4<float>::load(g + 4 * i, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b[e] *= c[e]; a[e] = c[e] * mask_hard(a[e], ternary, thr); }
        if (dm) Vec4<float>::store(dm + 4 * i, b);
        if (dw) Vec4<float>::store(dw + 4 * i, a);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        if (dm) dm[i] = g[i] * w[i];
        if (dw) dw[i] = g[i] * mask_hard(m[i], ternary, thr);
    }
}

// ------------------------------------------------------------------------------ column sum
// Block = 16 column groups (4 columns each) x 16 row lanes; rows are split over blockIdx.y.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ out, int Trows,
                                                     int N, long ld, int rows_per_block) {
    __shared__ float red[16][16][4];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = (blockIdx.x * 16 + cg) * 4;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    r1 = r1 < Trows ? r1 : Trows;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int r = r0 + rl; r < r1; r += 16) {
            float v[4];
            Vec4<T>::load(X + (long)r * ld + col, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[rl][cg][i] = acc[i];
    __syncthreads();
    if (rl == 0 && col < N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k][cg][i];
            atomicAdd(out + col + i, s);
        }
    }
}

inline int ew_grid(long total, int cap = 8192) {
    long g = (total + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int vitk_im2col(const float* img, void* cols, int B, int C, int Himg, int Wimg, int P, int dtype,
                           void* stream) {
    if (!img || !cols) return VITK_ERR_NULL;
    if (P % 4 || Wimg % 4 || Himg < P || Wimg < P) return VITK_ERR_SHAPE;
    const long total4 = (long)B * (Himg / P) * (Wimg / P) * C * P * P / 4;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(ew_grid(total4)), dim3(256), 0, st, img, (bf16_t*)cols, B, C,
                           Himg, Wimg, P, total4);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, st, img, (float*)cols, B, C,
                           Himg, Wimg, P, total4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_embed_cls(const float* cls, const float* pos, void* out, int B, int ntok, int D,
                              float dropout_p, uint64_t seed, int dtype, void* stream) {
    if (!cls || !pos || !out) return VITK_ERR_NULL;
    if (D % 4) return VITK_ERR_SHAPE;
    const uint32_t thr = vitk_drop_thr16(dropout_p);
    const float ds = vitk_drop_scale(thr);
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid((long)B * D / 4);
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(embed_cls_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, cls, pos, (bf16_t*)out, B, ntok, D,
                           thr, ds, seed);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(embed_cls_kernel<float>, dim3(grid), dim3(256), 0, st, cls, pos, (float*)out, B, ntok, D,
                           thr, ds, seed);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_embed_bwd(const void* dE, void* dtok, float* dpos, float* dcls, int B, int ntok, int D,
                              float dropout_p, uint64_t seed, int dE_dtype, int dtype, void* stream) {
    if (!dE || !dtok || !dpos || !dcls) return VITK_ERR_NULL;
    if (D % 4 || ntok < 2) return VITK_ERR_SHAPE;
    const uint32_t thr = vitk_drop_thr16(dropout_p);
    const float ds = vitk_drop_scale(thr);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(vitk_ceil_div((long)ntok * D / 4, 256), B < 16 ? B : 16);
    if (dE_dtype == VITK_BF16 && dtype == VITK_BF16)
        hipLaunchKernelGGL((embed_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dE,
                           (bf16_t*)dtok, dpos, dcls, B, ntok, D, thr, ds, seed);
    else if (dE_dtype == VITK_F32 && dtype == VITK_BF16)
        hipLaunchKernelGGL((embed_bwd_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)dE,
                           (bf16_t*)dtok, dpos, dcls, B, ntok, D, thr, ds, seed);
    else if (dE_dtype == VITK_F32 && dtype == VITK_F32)
        hipLaunchKernelGGL((embed_bwd_kernel<float, float>), grid, dim3(256), 0, st, (const float*)dE,
                           (float*)dtok, dpos, dcls, B, ntok, D, thr, ds, seed);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_cast_weight(const float* w, void* out, void* out_t, int R, int C, int dtype, void* stream) {
    if (!w || (!out && !out_t)) return VITK_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(vitk_ceil_div(C, 32), vitk_ceil_div(R, 32));
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(cast_weight_kernel<bf16_t>, grid, dim3(256), 0, st, w, (bf16_t*)out, (bf16_t*)out_t, R, C);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(cast_weight_kernel<float>, grid, dim3(256), 0, st, w, (float*)out, (float*)out_t, R, C);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_cast_weights_batched(const vitk_cast_entry* entries, int n_entries, int dtype, void* stream) {
    if (!entries) return VITK_ERR_NULL;
    if (n_entries <= 0) return VITK_ERR_SHAPE;
    if (dtype != VITK_BF16 && dtype != VITK_F32) return VITK_ERR_DTYPE;
    hipStream_t st = (hipStream_t)stream;
    for (int at = 0; at < n_entries; at += CAST_BATCH) {
        CastTable tab;
        tab.n = n_entries - at < CAST_BATCH ? n_entries - at : CAST_BATCH;
        int tiles = 0;
        for (int i = 0; i < tab.n; ++i) {
            vitk_cast_entry e = entries[at + i];
            if (!e.w || (!e.out && !e.out_t)) return VITK_ERR_NULL;
            if (e.rows <= 0 || e.cols <= 0 || (e.out_t && e.ld_t < e.rows)) return VITK_ERR_SHAPE;
            e.tile0 = tiles;
            tiles += vitk_ceil_div(e.rows, 64) * vitk_ceil_div(e.cols, 64);
            tab.e[i] = e;
        }
        if (dtype == VITK_BF16) hipLaunchKernelGGL(cast_weights_batched_kernel<bf16_t>, dim3(tiles), dim3(256), 0, st, tab);
        else hipLaunchKernelGGL(cast_weights_batched_kernel<float>, dim3(tiles), dim3(256), 0, st, tab);
        VITK_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int vitk_dropout_bwd(const void* dy, void* dx, long rows, int cols, float dropout_p, uint64_t seed,
                                int dy_dtype, int dtype, void* stream) {
    if (!dy || !dx) return VITK_ERR_NULL;
    if (cols % 4) return VITK_ERR_SHAPE;
    const uint32_t thr = vitk_drop_thr16(dropout_p);
    const float ds = vitk_drop_scale(thr);
    const long total4 = rows * cols / 4;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_grid(total4));
    if (dy_dtype == VITK_BF16 && dtype == VITK_BF16)
        hipLaunchKernelGGL((dropout_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dy,
                           (bf16_t*)dx, total4, thr, ds, seed);
    else if (dy_dtype == VITK_F32 && dtype == VITK_BF16)
        hipLaunchKernelGGL((dropout_bwd_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)dy,
                           (bf16_t*)dx, total4, thr, ds, seed);
    else if (dy_dtype == VITK_F32 && dtype == VITK_F32)
        hipLaunchKernelGGL((dropout_bwd_kernel<float, float>), grid, dim3(256), 0, st, (const float*)dy,
                           (float*)dx, total4, thr, ds, seed);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_convert(const void* src, void* dst, long n, int dtype_src, int dtype_dst, void* stream) {
    if (!src || !dst) return VITK_ERR_NULL;
    if (n % 4) return VITK_ERR_SHAPE;
    const long total4 = n / 4;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_grid(total4));
    if (dtype_src == VITK_F32 && dtype_dst == VITK_BF16)
        hipLaunchKernelGGL((convert_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, total4);
    else if (dtype_src == VITK_BF16 && dtype_dst == VITK_F32)
        hipLaunchKernelGGL((convert_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, (float*)dst, total4);
    else if (dtype_src == VITK_F32 && dtype_dst == VITK_F32)
        hipLaunchKernelGGL((convert_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, total4);
    else if (dtype_src == VITK_BF16 && dtype_dst == VITK_BF16)
        hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, total4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_rkr_fold_fwd(const float* W, const float* b, const float* LM, const float* RM, const float* F,
                                 float* Wt, float* bt, int O, int I, int K, void* stream) {
    if (!W || !b || !LM || !RM || !Wt || !bt) return VITK_ERR_NULL;
    if (O <= 0 || I <= 0 || I % 4 || K < 1 || K > RKR_MAXK) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(rkr_fold_fwd_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, W, b, LM, RM, F, Wt, bt, O, I, K);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_rkr_fold_bwd(const float* W, const float* b, const float* LM, const float* RM, const float* F,
                                 const float* dWt, const float* dbt, float* dW, float* db, float* dLM, float* dRM,
                                 float* dF, int O, int I, int K, void* stream) {
    if (!W || !b || !LM || !RM || !dWt || !dbt || !db || !dLM || !dRM) return VITK_ERR_NULL;
    if ((F == nullptr) != (dF == nullptr)) return VITK_ERR_NULL;
    if (O <= 0 || I <= 0 || I % 4 || K < 1 || K > RKR_MAXK) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rkr_fold_bwd_row_kernel, dim3(O), dim3(256), 0, st, W, b, LM, RM, F, dWt, dbt, dW, db, dRM, dF, O, I, K);
    VITK_CHECK_LAUNCH();
    // about a thousand workgroups: the column walk is latency-bound, the K atomics per column and slice are cheap
    const int colblocks = vitk_ceil_div(I, 256);
    int slices = 1024 / colblocks;
    slices = slices > O / 8 ? O / 8 : slices;
    slices = slices < 1 ? 1 : slices;
    const int rps = vitk_ceil_div(O, slices);
    hipLaunchKernelGGL(rkr_fold_bwd_col_kernel, dim3(vitk_ceil_div(I, 256), vitk_ceil_div(O, rps)), dim3(256), 0, st, RM, F, dWt,
                       dLM, O, I, K, rps);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_mask_fold_fwd(const float* mask_real, const float* w, float* out, long n, int ternary, float threshold,
                                  void* stream) {
    if (!mask_real || !w || !out) return VITK_ERR_NULL;
    if (n <= 0) return VITK_ERR_SHAPE;
    if (((uintptr_t)mask_real | (uintptr_t)w | (uintptr_t)out) % 16) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(mask_fold_fwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, mask_real, w, out, n,
                       ternary, threshold);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_mask_fold_bwd(const float* mask_real, const float* w, const float* d_out, float* d_mask, float* d_w,
                                  long n, int ternary, float threshold, void* stream) {
    if (!mask_real || !w || !d_out || (!d_mask && !d_w)) return VITK_ERR_NULL;
    if (n <= 0) return VITK_ERR_SHAPE;
    if (((uintptr_t)mask_real | (uintptr_t)w | (uintptr_t)d_out | (uintptr_t)d_mask | (uintptr_t)d_w) % 16) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(mask_fold_bwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, mask_real, w, d_out,
                       d_mask, d_w, n, ternary, threshold);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_scale_add(const float* h, const void* y, const float* scale, float* out, long samples,
                              long per_sample, int dtype, void* stream) {
    if (!h || !y || !scale || !out) return VITK_ERR_NULL;
    if (samples <= 0 || per_sample <= 0 || per_sample % 4) return VITK_ERR_SHAPE;
    const long total4 = samples * per_sample / 4;
    const dim3 grid((unsigned)(total4 / 256 + 1 < 8192 ? total4 / 256 + 1 : 8192));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(scale_add_kernel<bf16_t>, grid, dim3(256), 0, st, h, (const bf16_t*)y, scale, out, total4, per_sample / 4);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(scale_add_kernel<float>, grid, dim3(256), 0, st, h, (const float*)y, scale, out, total4, per_sample / 4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_scale_cast(const float* dout, const float* scale, void* dy, long samples, long per_sample, int dtype,
                               void* stream) {
    if (!dout || !scale || !dy) return VITK_ERR_NULL;
    if (samples <= 0 || per_sample <= 0 || per_sample % 4) return VITK_ERR_SHAPE;
    const long total4 = samples * per_sample / 4;
    const dim3 grid((unsigned)(total4 / 256 + 1 < 8192 ? total4 / 256 + 1 : 8192));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, grid, dim3(256), 0, st, dout, scale, (bf16_t*)dy, total4, per_sample / 4);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(scale_cast_kernel<float>, grid, dim3(256), 0, st, dout, scale, (float*)dy, total4, per_sample / 4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_colsum_acc(const void* X, float* out, int T, int N, long ld, int dtype, void* stream) {
    if (!X || !out) return VITK_ERR_NULL;
    if (N % 4 || ld % 4 || T <= 0) return VITK_ERR_SHAPE;
    const int gx = vitk_ceil_div(N, 64);
    int gy = vitk_ceil_div(1024, gx);
    const int max_gy = vitk_ceil_div(T, 64);
    if (gy > max_gy) gy = max_gy;
    const int rpb = vitk_ceil_div(T, gy);
    gy = vitk_ceil_div(T, rpb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(gx, gy), dim3(256), 0, st, (const bf16_t*)X, out, T, N, ld, rpb);
    else if (dtype == VITK_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, gy), dim3(256), 0, st, (const float*)X, out, T, N, ld, rpb);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}
