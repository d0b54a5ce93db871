// Kernels of the R50 hybrid stem that are not convolutions (reference models/ViT/modeling_resnet.py):
// weight standardisation of StdConv2d (:35-42), GroupNorm (+ ReLU) of the bottleneck units (:55-91) on
// channels-last activations, and the residual add + ReLU that ends a unit (:91).  All HBM-bound passes.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {    // 256 threads; every thread gets the sum
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------- weight standardisation
// w_hat[o][:] = (w[o][:] - mean_o) * rstd_o, rstd_o = 1 / sqrt(var_o + eps) (biased variance).  One block per row.
__global__ __launch_bounds__(256) void wstd_fwd_kernel(const float* __restrict__ w, float* __restrict__ w_hat,
                                                       float* __restrict__ rstd_o, int F, float eps) {
    __shared__ float red[4];
    const float* wr = w + (long)blockIdx.x * F;
    float s = 0.f;
    for (int i = threadIdx.x; i < F; i += 256) s += wr[i];
    const float mean = block_sum(s, red) / (float)F;
    float q = 0.f;
    for (int i = threadIdx.x; i < F; i += 256) {
        const float d = wr[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)F + eps);
    float* o = w_hat + (long)blockIdx.x * F;
    for (int i = threadIdx.x; i < F; i += 256) o[i] = (wr[i] - mean) * rstd;
    if (threadIdx.x == 0) rstd_o[blockIdx.x] = rstd;
}

// dw = rstd * (g - mean(g) - w_hat * mean(g * w_hat)),  g = d(loss)/d(w_hat)
__global__ __launch_bounds__(256) void wstd_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w_hat,
                                                       const float* __restrict__ rstd_i, float* __restrict__ dw, int F) {
    __shared__ float red[4];
    const float* gr = g + (long)blockIdx.x * F;
    const float* hr = w_hat + (long)blockIdx.x * F;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < F; i += 256) {
        s1 += gr[i];
        s2 += gr[i] * hr[i];
    }
    const float c1 = block_sum(s1, red) / (float)F;
    const float c2 = block_sum(s2, red) / (float)F;
    const float rstd = rstd_i[blockIdx.x];
    float* o = dw + (long)blockIdx.x * F;
    for (int i = threadIdx.x; i < F; i += 256) o[i] = rstd * (gr[i] - c1 - hr[i] * c2);
}

// ---------------------------------------------------------------------------- GroupNorm on [B, HW, C]
// A block of 256 threads is a (256 / (C/4)) x (C/4) grid: thread (r, cv) owns channels 4 cv .. 4 cv + 3 and
// walks rows r, r + R, ... of its slice of the image (float4 loads, a whole row per C/4 threads).
struct GnGeom {
    int B, HW, C, G, S;       // S slices of the HW rows per image
    int rows_per_slice;
};

template <typename T> __device__ __forceinline__ void ld4(const T* p, float* o) { Vec4<T>::load(p, o); }

// Pass 1 of forward: per (image, slice, channel): mean and centred sum of squares over the slice's rows.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, GnGeom gm) {
    __shared__ float red[2][256][4];
    const int cvn = gm.C >> 2, R = 256 / cvn;
    const int cv = threadIdx.x % cvn, r = threadIdx.x / cvn;
    const int b = blockIdx.y, s = blockIdx.x;
    const int row0 = s * gm.rows_per_slice;
    int row1 = row0 + gm.rows_per_slice;
    row1 = row1 < gm.HW ? row1 : gm.HW;
    const float* xb = x + ((long)b * gm.HW) * gm.C + cv * 4;
    float sum[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int row = row0 + r; row < row1; row += R) {
        float v[4];
        ld4(xb + (long)row * gm.C, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sum[j] += v[j]; sq[j] += v[j] * v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = sum[j]; red[1][threadIdx.x][j] = sq[j]; }
    __syncthreads();
    if (r == 0) {
        const float n = (float)(row1 - row0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, q = 0.f;
            for (int rr = 0; rr < R; ++rr) { a += red[0][rr * cvn + cv][j]; q += red[1][rr * cvn + cv][j]; }
            const float mean = n > 0.f ? a / n : 0.f;
            float* o = part + (((long)b * gm.S + s) * gm.C + cv * 4 + j) * 2;
            o[0] = mean;
            o[1] = fmaxf(q - a * mean, 0.f);          // sum (x - mean)^2 over the slice
        }
    }
}

// Sum over the `lp` adjacent lanes (a power of two up to 64) that share one (image, group).
__device__ __forceinline__ float lanes_sum(float v, int lp) {
    for (int o = lp >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Pass 2: `lp` lanes per (image, group) walk its slices x channels entries: group mean first, then the centred sum of
// squares  sum m2_e + n_e (mean_e - mean)^2  (the pairwise form of Chan's update, independent of the order).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, GnGeom gm, float eps, int lp) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int idx = t / lp, l = t - idx * lp;
    const bool live = idx < gm.B * gm.G;
    const int b = live ? idx / gm.G : 0, g = live ? idx - b * gm.G : 0;
    const int cg = gm.C / gm.G, ne = gm.S * cg;
    float a = 0.f;
    for (int e = l; e < ne && live; e += lp) {
        const int s = e / cg, c = g * cg + (e - s * cg);
        int rows = gm.HW - s * gm.rows_per_slice;
        rows = rows < gm.rows_per_slice ? rows : gm.rows_per_slice;
        a += (float)rows * part[(((long)b * gm.S + s) * gm.C + c) * 2];
    }
    const float n = (float)gm.HW * (float)cg;
    const float mean = lanes_sum(a, lp) / n;
    float m2 = 0.f;
    for (int e = l; e < ne && live; e += lp) {
        const int s = e / cg, c = g * cg + (e - s * cg);
        int rows = gm.HW - s * gm.rows_per_slice;
        rows = rows < gm.rows_per_slice ? rows : gm.rows_per_slice;
        const float2 p = *reinterpret_cast<const float2*>(part + (((long)b * gm.S + s) * gm.C + c) * 2);
        const float d = p.x - mean;
        m2 += p.y + (float)rows * d * d;
    }
    m2 = lanes_sum(m2, lp);
    if (live && l == 0) {
        mean_o[idx] = mean;
        rstd_o[idx] = rsqrtf(m2 / n + eps);
    }
}

// Pass 3: y = [relu]((x - mean) * rstd * gamma + beta)
template <typename TO>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean_i,
                                                       const float* __restrict__ rstd_i, TO* __restrict__ y, GnGeom gm,
                                                       int relu) {
    const long n4 = (long)gm.B * gm.HW * (gm.C >> 2);
    const int cvn = gm.C >> 2, cg = gm.C / gm.G;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const long row = i / cvn;
        const int b = (int)(row / gm.HW);
        float v[4], ga[4], be[4], o[4];
        ld4(x + i * 4, v);
        ld4(gamma + cv * 4, ga);
        ld4(beta + cv * 4, be);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (cv * 4 + j) / cg;
            const float t = (v[j] - mean_i[b * gm.G + g]) * rstd_i[b * gm.G + g] * ga[j] + be[j];
            o[j] = relu ? fmaxf(t, 0.f) : t;
        }
        Vec4<TO>::store(y + i * 4, o);
    }
}

// Pass 3 of the unit's last GroupNorm: out = relu(res + (x - mean) * rstd * gamma + beta) in fp32, and a second copy
// in the compute dtype (the next unit's GEMM operand) when out_lp is given.
template <typename TL>
__global__ __launch_bounds__(256) void gn_apply_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean_i,
                                                                const float* __restrict__ rstd_i, const float* __restrict__ res,
                                                                float* __restrict__ out, TL* __restrict__ out_lp, GnGeom gm) {
    const long n4 = (long)gm.B * gm.HW * (gm.C >> 2);
    const int cvn = gm.C >> 2, cg = gm.C / gm.G;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const long row = i / cvn;
        const int b = (int)(row / gm.HW);
        float v[4], r[4], ga[4], be[4], o[4];
        ld4(x + i * 4, v);
        ld4(res + i * 4, r);
        ld4(gamma + cv * 4, ga);
        ld4(beta + cv * 4, be);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (cv * 4 + j) / cg;
            const float t = (v[j] - mean_i[b * gm.G + g]) * rstd_i[b * gm.G + g] * ga[j] + be[j];
            o[j] = fmaxf(r[j] + t, 0.f);
        }
        Vec4<float>::store(out + i * 4, o);
        if (out_lp) Vec4<TL>::store(out_lp + i * 4, o);
    }
}

// Backward pass 1: per (image, slice, channel): s1 = sum dyr, s2 = sum dyr * x_hat, with dyr = dy masked by the
// ReLU (recomputed from x) when `relu`.
template <typename TDY>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                           float* __restrict__ part, GnGeom gm, int relu) {
    __shared__ float red[2][256][4];
    const int cvn = gm.C >> 2, R = 256 / cvn, cg = gm.C / gm.G;
    const int cv = threadIdx.x % cvn, r = threadIdx.x / cvn;
    const int b = blockIdx.y, s = blockIdx.x;
    const int row0 = s * gm.rows_per_slice;
    int row1 = row0 + gm.rows_per_slice;
    row1 = row1 < gm.HW ? row1 : gm.HW;
    const long base = ((long)b * gm.HW) * gm.C + cv * 4;
    float ga[4], be[4], mu[4], rs[4];
    ld4(gamma + cv * 4, ga);
    ld4(beta + cv * 4, be);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (cv * 4 + j) / cg;
        mu[j] = mean_i[b * gm.G + g];
        rs[j] = rstd_i[b * gm.G + g];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int row = row0 + r; row < row1; row += R) {
        float v[4], d[4];
        ld4(x + base + (long)row * gm.C, v);
        ld4(dy + base + (long)row * gm.C, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dd = (relu && xh * ga[j] + be[j] <= 0.f) ? 0.f : d[j];
            s1[j] += dd;
            s2[j] += dd * xh;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, q = 0.f;
            for (int rr = 0; rr < R; ++rr) { a += red[0][rr * cvn + cv][j]; q += red[1][rr * cvn + cv][j]; }
            float* o = part + (((long)b * gm.S + s) * gm.C + cv * 4 + j) * 2;
            o[0] = a;
            o[1] = q;
        }
    }
}

// Backward pass 1 behind the sum that ends a unit: d = (g1 [+ g2]) where out > 0, else 0 (what add_relu_bwd_kernel
// writes: the gradient of the residual AND of this GroupNorm's output), stored, and the same partial sums taken of it.
template <typename T2>
__global__ __launch_bounds__(256) void gn_bwd_stats_sum_kernel(const float* __restrict__ g1, const T2* __restrict__ g2,
                                                               const float* __restrict__ out, const float* __restrict__ x,
                                                               const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                               float* __restrict__ d, float* __restrict__ part, GnGeom gm) {
    __shared__ float red[2][256][4];
    const int cvn = gm.C >> 2, R = 256 / cvn, cg = gm.C / gm.G;
    const int cv = threadIdx.x % cvn, r = threadIdx.x / cvn;
    const int b = blockIdx.y, s = blockIdx.x;
    const int row0 = s * gm.rows_per_slice;
    int row1 = row0 + gm.rows_per_slice;
    row1 = row1 < gm.HW ? row1 : gm.HW;
    const long base = ((long)b * gm.HW) * gm.C + cv * 4;
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (cv * 4 + j) / cg;
        mu[j] = mean_i[b * gm.G + g];
        rs[j] = rstd_i[b * gm.G + g];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int row = row0 + r; row < row1; row += R) {
        const long at = base + (long)row * gm.C;
        float v[4], u[4], y[4], o[4];
        ld4(x + at, v);
        ld4(g1 + at, u);
        ld4(out + at, y);
        if (g2) {
            float w[4];
            ld4(g2 + at, w);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] += w[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = y[j] > 0.f ? u[j] : 0.f;
            s1[j] += o[j];
            s2[j] += o[j] * ((v[j] - mu[j]) * rs[j]);
        }
        Vec4<float>::store(d + at, o);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, q = 0.f;
            for (int rr = 0; rr < R; ++rr) { a += red[0][rr * cvn + cv][j]; q += red[1][rr * cvn + cv][j]; }
            float* o = part + (((long)b * gm.S + s) * gm.C + cv * 4 + j) * 2;
            o[0] = a;
            o[1] = q;
        }
    }
}

// Backward pass 2a: `lp` lanes per (image, group): A = mean over the group of gamma * dyr * x_hat, Bm = mean of gamma * dyr.
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                           float* __restrict__ ab, GnGeom gm, int lp) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int idx = t / lp, l = t - idx * lp;
    const bool live = idx < gm.B * gm.G;
    const int b = live ? idx / gm.G : 0, g = live ? idx - b * gm.G : 0;
    const int cg = gm.C / gm.G, ne = gm.S * cg;
    float a = 0.f, bm = 0.f;
    for (int e = l; e < ne && live; e += lp) {
        const int s = e / cg, c = g * cg + (e - s * cg);
        const float2 p = *reinterpret_cast<const float2*>(part + (((long)b * gm.S + s) * gm.C + c) * 2);
        const float ga = gamma[c];
        bm += ga * p.x;
        a += ga * p.y;
    }
    a = lanes_sum(a, lp);
    bm = lanes_sum(bm, lp);
    if (live && l == 0) {
        const float inv_n = 1.0f / ((float)gm.HW * (float)cg);
        ab[idx * 2 + 0] = a * inv_n;
        ab[idx * 2 + 1] = bm * inv_n;
    }
}

// Backward pass 2b: dgamma[c] += sum over images and slices of s2, dbeta[c] += sum of s1.  One block per 64
// channels, 16 threads per channel walking interleaved (image, slice) rows, summed in a fixed order through LDS.
__global__ __launch_bounds__(1024) void gn_bwd_param_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, GnGeom gm) {
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float t1 = 0.f, t2 = 0.f;
    if (c < gm.C) {
        const int n = gm.B * gm.S;
#pragma unroll 4
        for (int bs = q; bs < n; bs += 16) {
            const float2 p = *reinterpret_cast<const float2*>(part + ((long)bs * gm.C + c) * 2);
            t1 += p.x;
            t2 += p.y;
        }
    }
    red[0][q][cl] = t1;
    red[1][q][cl] = t2;
    __syncthreads();
    if (q == 0 && c < gm.C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        dgamma[c] += b;
        dbeta[c] += a;
    }
}

// Backward pass 3: dx = rstd * (gamma * dyr - Bm - x_hat * A)
template <typename TDY, typename TDX>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                           const float* __restrict__ ab, TDX* __restrict__ dx, GnGeom gm,
                                                           int relu) {
    const long n4 = (long)gm.B * gm.HW * (gm.C >> 2);
    const int cvn = gm.C >> 2, cg = gm.C / gm.G;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const long row = i / cvn;
        const int b = (int)(row / gm.HW);
        float v[4], d[4], ga[4], be[4], o[4];
        ld4(x + i * 4, v);
        ld4(dy + i * 4, d);
        ld4(gamma + cv * 4, ga);
        ld4(beta + cv * 4, be);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = b * gm.G + (cv * 4 + j) / cg;
            const float rs = rstd_i[g];
            const float xh = (v[j] - mean_i[g]) * rs;
            const float dd = (relu && xh * ga[j] + be[j] <= 0.f) ? 0.f : d[j];
            o[j] = rs * (ga[j] * dd - ab[g * 2 + 1] - xh * ab[g * 2 + 0]);
        }
        Vec4<TDX>::store(dx + i * 4, o);
    }
}

// ---------------------------------------------------------------------------- residual add + ReLU
template <typename TL>
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, TL* __restrict__ out_lp, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float u[4], v[4], o[4];
        ld4(a + i * 4, u);
        ld4(b + i * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(u[j] + v[j], 0.f);
        Vec4<float>::store(out + i * 4, o);
        if (out_lp) Vec4<TL>::store(out_lp + i * 4, o);
    }
}

// d = (g1 [+ g2]) where out > 0, else 0: the gradient of both addends
template <typename T2>
__global__ __launch_bounds__(256) void add_relu_bwd_kernel(const float* __restrict__ g1, const T2* __restrict__ g2,
                                                           const float* __restrict__ out, float* __restrict__ d, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float u[4], y[4], o[4];
        ld4(g1 + i * 4, u);
        ld4(out + i * 4, y);
        if (g2) {
            float w[4];
            ld4(g2 + i * 4, w);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] += w[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = y[j] > 0.f ? u[j] : 0.f;
        Vec4<float>::store(d + i * 4, o);
    }
}

// ---------------------------------------------------------------------------- im2col for the 3x3 / 7x7 convolutions
// cols[(b, oy, ox)][(kh, kw, c)] = x[b][oy*stride + kh - pad][ox*stride + kw - pad][c], zero outside the image and
// in the padding columns kh*kw*C .. Kp-1.  The input is addressed through element strides, so it may be
// channels-last (the bottleneck units) or NCHW (the image under the root convolution).
struct ColGeom {
    int B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp;
    long sb, sh, sw, sc;
};

template <typename TO>
__global__ __launch_bounds__(256) void im2col_any_kernel(const float* __restrict__ x, TO* __restrict__ cols, ColGeom gm) {
    const long total = (long)gm.B * gm.Ho * gm.Wo * gm.Kp;
    const int K = gm.KH * gm.KW * gm.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % gm.Kp);
        const long row = i / gm.Kp;
        float v = 0.f;
        if (k < K) {
            const int c = k % gm.C, kk = k / gm.C;
            const int kw = kk % gm.KW, kh = kk / gm.KW;
            const int ox = (int)(row % gm.Wo);
            const long t = row / gm.Wo;
            const int oy = (int)(t % gm.Ho), b = (int)(t / gm.Ho);
            const int iy = oy * gm.stride + kh - gm.pad, ix = ox * gm.stride + kw - gm.pad;
            if (iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W) v = x[b * gm.sb + iy * gm.sh + ix * gm.sw + c * gm.sc];
        }
        cols[i] = (TO)v;
    }
}

// Any strides again, 8 consecutive window elements of one output position per thread (Kp % 8 == 0, fewer than 2^31
// such octets): the position is decomposed once per thread and (kh, kw, c) stepped instead of divided, one 16 B (bf16)
// store per thread.  The 7x7 root convolution over the NCHW image runs here.
template <typename TO>
__global__ __launch_bounds__(256) void im2col_any8_kernel(const float* __restrict__ x, TO* __restrict__ cols, ColGeom gm) {
    const unsigned kv = (unsigned)gm.Kp >> 3, K = (unsigned)(gm.KH * gm.KW * gm.C);
    const unsigned total = (unsigned)gm.B * gm.Ho * gm.Wo * kv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / kv, k0 = (i - row * kv) * 8;
        const unsigned ox = row % (unsigned)gm.Wo, t = row / (unsigned)gm.Wo;
        const unsigned oy = t % (unsigned)gm.Ho, b = t / (unsigned)gm.Ho;
        int c = (int)(k0 % (unsigned)gm.C);
        const unsigned kk = k0 / (unsigned)gm.C;
        int kw = (int)(kk % (unsigned)gm.KW), kh = (int)(kk / (unsigned)gm.KW);
        const float* xb = x + (long)b * gm.sb;
        const int iy0 = (int)oy * gm.stride - gm.pad, ix0 = (int)ox * gm.stride - gm.pad;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = iy0 + kh, ix = ix0 + kw;
            v[j] = 0.f;
            if (k0 + j < K && iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W) v[j] = xb[iy * gm.sh + ix * gm.sw + c * gm.sc];
            if (++c == gm.C) {
                c = 0;
                if (++kw == gm.KW) { kw = 0; ++kh; }
            }
        }
        Vec4<TO>::store(cols + (long)i * 8, v);
        Vec4<TO>::store(cols + (long)i * 8 + 4, v + 4);
    }
}

// The same for a channels-last input with C % 8 == 0 (every bottleneck 3x3): one thread per 8 consecutive channels
// of one window position, 32 B read, 8 outputs written at once; Kp % 8 == 0.
template <typename TO>
__global__ __launch_bounds__(256) void im2col_cl8_kernel(const float* __restrict__ x, TO* __restrict__ cols, ColGeom gm) {
    const int kv = gm.Kp >> 3, K = gm.KH * gm.KW * gm.C;
    const long total = (long)gm.B * gm.Ho * gm.Wo * kv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % kv) * 8;
        const long row = i / kv;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const int c = k % gm.C, kk = k / gm.C;
            const int kw = kk % gm.KW, kh = kk / gm.KW;
            const int ox = (int)(row % gm.Wo);
            const long t = row / gm.Wo;
            const int oy = (int)(t % gm.Ho), b = (int)(t / gm.Ho);
            const int iy = oy * gm.stride + kh - gm.pad, ix = ox * gm.stride + kw - gm.pad;
            if (iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W) {
                const float* src = x + (((long)b * gm.H + iy) * gm.W + ix) * gm.C + c;
                ld4(src, v);
                ld4(src + 4, v + 4);
            }
        }
        Vec4<TO>::store(cols + i * 8, v);
        Vec4<TO>::store(cols + i * 8 + 4, v + 4);
    }
}

// Adjoint of im2col_any_kernel for a channels-last input: dx[b][iy][ix][c] = sum over the (kh, kw) whose window
// position covers (iy, ix) of dcols[(b, oy, ox)][(kh, kw, c)].  A gather: no atomics, every dx element written once.
template <typename TI>
__global__ __launch_bounds__(256) void col2im_cl_kernel(const TI* __restrict__ dcols, float* __restrict__ dx, ColGeom gm) {
    const long total = (long)gm.B * gm.H * gm.W * gm.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % gm.C);
        long t = i / gm.C;
        const int ix = (int)(t % gm.W);
        t /= gm.W;
        const int iy = (int)(t % gm.H), b = (int)(t / gm.H);
        float acc = 0.f;
        for (int kh = 0; kh < gm.KH; ++kh) {
            const int ny = iy + gm.pad - kh;
            if (ny < 0 || ny % gm.stride) continue;
            const int oy = ny / gm.stride;
            if (oy >= gm.Ho) continue;
            for (int kw = 0; kw < gm.KW; ++kw) {
                const int nx = ix + gm.pad - kw;
                if (nx < 0 || nx % gm.stride) continue;
                const int ox = nx / gm.stride;
                if (ox >= gm.Wo) continue;
                acc += (float)dcols[(((long)b * gm.Ho + oy) * gm.Wo + ox) * gm.Kp + (kh * gm.KW + kw) * gm.C + c];
            }
        }
        dx[i] = acc;
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void col2im_cl8_kernel(const TI* __restrict__ dcols, float* __restrict__ dx, ColGeom gm) {
    const int cv = gm.C >> 3;
    const long total = (long)gm.B * gm.H * gm.W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        long t = i / cv;
        const int ix = (int)(t % gm.W);
        t /= gm.W;
        const int iy = (int)(t % gm.H), b = (int)(t / gm.H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < gm.KH; ++kh) {
            const int ny = iy + gm.pad - kh;
            if (ny < 0 || ny % gm.stride) continue;
            const int oy = ny / gm.stride;
            if (oy >= gm.Ho) continue;
            for (int kw = 0; kw < gm.KW; ++kw) {
                const int nx = ix + gm.pad - kw;
                if (nx < 0 || nx % gm.stride) continue;
                const int ox = nx / gm.stride;
                if (ox >= gm.Wo) continue;
                const TI* src = dcols + (((long)b * gm.Ho + oy) * gm.Wo + ox) * gm.Kp + (kh * gm.KW + kw) * gm.C + c;
                float v[8];
                ld4(src, v);
                ld4(src + 4, v + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        }
        Vec4<float>::store(dx + i * 8, acc);
        Vec4<float>::store(dx + i * 8 + 4, acc + 4);
    }
}

inline int ew_grid(long n4) {
    long g = (n4 + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

inline int gn_geom(GnGeom& gm, int B, int HW, int C, int G) {
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || C % 4) return VITK_ERR_SHAPE;
    const int cvn = C / 4;
    if (cvn > 256 || 256 % cvn) return VITK_ERR_SHAPE;              // C in {4, 8, ..., 1024}, a power of two
    gm.B = B; gm.HW = HW; gm.C = C; gm.G = G;
    // enough slices to fill the chip (about 1024 blocks), at least 8 rows per block row group
    const int R = 256 / cvn;
    int S = (1024 + B - 1) / B;
    const int max_s = (HW + 8 * R - 1) / (8 * R);
    S = S > max_s ? max_s : S;
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    gm.rows_per_slice = (HW + S - 1) / S;
    gm.S = (HW + gm.rows_per_slice - 1) / gm.rows_per_slice;
    return 0;
}

// lanes per (image, group) of the two small per-group kernels: the power of two covering slices x channels, at most 64
inline int gn_lanes(const GnGeom& gm) {
    const int ne = gm.S * (gm.C / gm.G);
    int lp = 1;
    while (lp < ne && lp < 64) lp <<= 1;
    return lp;
}

}  // namespace

extern "C" int vitk_wstd_fwd(const float* w, float* w_hat, float* rstd, int O, int F, float eps, void* stream) {
    if (!w || !w_hat || !rstd) return VITK_ERR_NULL;
    if (O <= 0 || F <= 0) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(wstd_fwd_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, w_hat, rstd, F, eps);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_wstd_bwd(const float* dw_hat, const float* w_hat, const float* rstd, float* dw, int O, int F,
                             void* stream) {
    if (!dw_hat || !w_hat || !rstd || !dw) return VITK_ERR_NULL;
    if (O <= 0 || F <= 0) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(wstd_bwd_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, dw_hat, w_hat, rstd, dw, F);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t vitk_groupnorm_workspace(int B, int HW, int C, int G) {
    GnGeom gm;
    if (gn_geom(gm, B, HW, C, G)) return 0;
    return ((size_t)B * gm.S * C * 2 + (size_t)B * G * 2) * sizeof(float);
}

extern "C" int vitk_groupnorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, void* workspace, int B, int HW, int C, int G, float eps, int relu,
                                  int out_dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace) return VITK_ERR_NULL;
    GnGeom gm;
    const int rc = gn_geom(gm, B, HW, C, G);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gm.S, B), dim3(256), 0, st, x, part, gm);
    VITK_CHECK_LAUNCH();
    const int lp = gn_lanes(gm);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(vitk_ceil_div((long)B * G * lp, 256)), dim3(256), 0, st, part, mean, rstd, gm, eps, lp);
    VITK_CHECK_LAUNCH();
    const long n4 = (long)B * HW * (C / 4);
    if (out_dtype == VITK_F32)
        hipLaunchKernelGGL((gn_apply_kernel<float>), dim3(ew_grid(n4)), dim3(256), 0, st, x, gamma, beta, mean, rstd, (float*)y, gm, relu);
    else if (out_dtype == VITK_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), dim3(ew_grid(n4)), dim3(256), 0, st, x, gamma, beta, mean, rstd, (bf16_t*)y, gm, relu);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_groupnorm_add_relu_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                                           float* out, void* out_lp, float* mean, float* rstd, void* workspace, int B,
                                           int HW, int C, int G, float eps, int lp_dtype, void* stream) {
    if (!x || !gamma || !beta || !residual || !out || !mean || !rstd || !workspace) return VITK_ERR_NULL;
    if (out_lp && lp_dtype != VITK_F32 && lp_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    GnGeom gm;
    const int rc = gn_geom(gm, B, HW, C, G);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gm.S, B), dim3(256), 0, st, x, part, gm);
    VITK_CHECK_LAUNCH();
    const int lp = gn_lanes(gm);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(vitk_ceil_div((long)B * G * lp, 256)), dim3(256), 0, st, part, mean, rstd, gm, eps, lp);
    VITK_CHECK_LAUNCH();
    const long n4 = (long)B * HW * (C / 4);
    if (out_lp && lp_dtype == VITK_BF16)
        hipLaunchKernelGGL((gn_apply_add_relu_kernel<bf16_t>), dim3(ew_grid(n4)), dim3(256), 0, st, x, gamma, beta, mean, rstd, residual, out, (bf16_t*)out_lp, gm);
    else
        hipLaunchKernelGGL((gn_apply_add_relu_kernel<float>), dim3(ew_grid(n4)), dim3(256), 0, st, x, gamma, beta, mean, rstd, residual, out, (float*)out_lp, gm);
    VITK_CHECK_LAUNCH();
    return 0;
}

// passes 2a, 2b and 3 of the backward, behind either form of pass 1
static int gn_bwd_tail(const void* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, void* dx, float* dgamma, float* dbeta, float* part, const GnGeom& gm, int relu,
                       int dy_dtype, int dx_dtype, hipStream_t st) {
    float* ab = part + (size_t)gm.B * gm.S * gm.C * 2;
    const long n4 = (long)gm.B * gm.HW * (gm.C / 4);
    const int lp = gn_lanes(gm);
    hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(vitk_ceil_div((long)gm.B * gm.G * lp, 256)), dim3(256), 0, st, part, gamma, ab, gm, lp);
    VITK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(vitk_ceil_div(gm.C, 64)), dim3(1024), 0, st, part, dgamma, dbeta, gm);
    VITK_CHECK_LAUNCH();
    const dim3 grid(ew_grid(n4));
#define VITK_GN_BWD_APPLY(TDY, TDX) \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<TDY, TDX>), grid, dim3(256), 0, st, (const TDY*)dy, x, gamma, beta, mean, rstd, ab, (TDX*)dx, gm, relu)
    if (dy_dtype == VITK_F32 && dx_dtype == VITK_F32) VITK_GN_BWD_APPLY(float, float);
    else if (dy_dtype == VITK_F32) VITK_GN_BWD_APPLY(float, bf16_t);
    else if (dx_dtype == VITK_F32) VITK_GN_BWD_APPLY(bf16_t, float);
    else VITK_GN_BWD_APPLY(bf16_t, bf16_t);
#undef VITK_GN_BWD_APPLY
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_groupnorm_bwd(const void* dy, const float* x, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                  void* workspace, int B, int HW, int C, int G, int relu, int dy_dtype, int dx_dtype,
                                  void* stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return VITK_ERR_NULL;
    if (dx_dtype != VITK_F32 && dx_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    GnGeom gm;
    const int rc = gn_geom(gm, B, HW, C, G);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    if (dy_dtype == VITK_F32)
        hipLaunchKernelGGL((gn_bwd_stats_kernel<float>), dim3(gm.S, B), dim3(256), 0, st, (const float*)dy, x, gamma, beta, mean, rstd, part, gm, relu);
    else if (dy_dtype == VITK_BF16)
        hipLaunchKernelGGL((gn_bwd_stats_kernel<bf16_t>), dim3(gm.S, B), dim3(256), 0, st, (const bf16_t*)dy, x, gamma, beta, mean, rstd, part, gm, relu);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return gn_bwd_tail(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, part, gm, relu, dy_dtype, dx_dtype, st);
}

extern "C" int vitk_groupnorm_add_relu_bwd(const float* g1, const void* g2, const float* out, const float* x,
                                           const float* gamma, const float* beta, const float* mean, const float* rstd,
                                           float* d, void* dx, float* dgamma, float* dbeta, void* workspace, int B, int HW,
                                           int C, int G, int g2_dtype, int dx_dtype, void* stream) {
    if (!g1 || !out || !x || !gamma || !beta || !mean || !rstd || !d || !dx || !dgamma || !dbeta || !workspace) return VITK_ERR_NULL;
    if (dx_dtype != VITK_F32 && dx_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    if (g2 && g2_dtype != VITK_F32 && g2_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    GnGeom gm;
    const int rc = gn_geom(gm, B, HW, C, G);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    if (g2 && g2_dtype == VITK_BF16)
        hipLaunchKernelGGL((gn_bwd_stats_sum_kernel<bf16_t>), dim3(gm.S, B), dim3(256), 0, st, g1, (const bf16_t*)g2, out, x, mean, rstd, d, part, gm);
    else
        hipLaunchKernelGGL((gn_bwd_stats_sum_kernel<float>), dim3(gm.S, B), dim3(256), 0, st, g1, (const float*)g2, out, x, mean, rstd, d, part, gm);
    VITK_CHECK_LAUNCH();
    return gn_bwd_tail(d, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, part, gm, 0, VITK_F32, dx_dtype, st);
}

extern "C" int vitk_add_relu_fwd(const float* a, const float* b, float* out, void* out_lp, long n, int lp_dtype,
                                 void* stream) {
    if (!a || !b || !out) return VITK_ERR_NULL;
    if (n <= 0 || n % 4) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / 4;
    if (out_lp && lp_dtype == VITK_BF16)
        hipLaunchKernelGGL((add_relu_fwd_kernel<bf16_t>), dim3(ew_grid(n4)), dim3(256), 0, st, a, b, out, (bf16_t*)out_lp, n4);
    else if (!out_lp || lp_dtype == VITK_F32)
        hipLaunchKernelGGL((add_relu_fwd_kernel<float>), dim3(ew_grid(n4)), dim3(256), 0, st, a, b, out, (float*)out_lp, n4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_add_relu_bwd(const float* g1, const void* g2, const float* out, float* d, long n, int g2_dtype,
                                 void* stream) {
    if (!g1 || !out || !d) return VITK_ERR_NULL;
    if (n <= 0 || n % 4) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / 4;
    if (g2 && g2_dtype == VITK_BF16)
        hipLaunchKernelGGL((add_relu_bwd_kernel<bf16_t>), dim3(ew_grid(n4)), dim3(256), 0, st, g1, (const bf16_t*)g2, out, d, n4);
    else if (!g2 || g2_dtype == VITK_F32)
        hipLaunchKernelGGL((add_relu_bwd_kernel<float>), dim3(ew_grid(n4)), dim3(256), 0, st, g1, (const float*)g2, out, d, n4);
    else
        return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_im2col_conv(const float* x, void* cols, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                                long sb, long sh, long sw, long sc, int Kp, int dtype, void* stream) {
    if (!x || !cols) return VITK_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Kp < KH * KW * C) return VITK_ERR_SHAPE;
    ColGeom gm;
    gm.B = B; gm.H = H; gm.W = W; gm.C = C; gm.KH = KH; gm.KW = KW; gm.stride = stride; gm.pad = pad; gm.Kp = Kp;
    gm.Ho = (H + 2 * pad - KH) / stride + 1;
    gm.Wo = (W + 2 * pad - KW) / stride + 1;
    gm.sb = sb; gm.sh = sh; gm.sw = sw; gm.sc = sc;
    if (gm.Ho <= 0 || gm.Wo <= 0) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != VITK_BF16 && dtype != VITK_F32) return VITK_ERR_DTYPE;
    const bool cl8 = C % 8 == 0 && Kp % 8 == 0 && sc == 1 && sw == C && sh == (long)W * C && sb == (long)H * W * C &&
                     ((uintptr_t)x % 16 == 0) && ((uintptr_t)cols % 16 == 0);
    if (cl8) {
        long grid8 = ((long)B * gm.Ho * gm.Wo * (Kp / 8) + 255) / 256;
        grid8 = grid8 > 16384 ? 16384 : grid8;
        if (dtype == VITK_BF16) hipLaunchKernelGGL((im2col_cl8_kernel<bf16_t>), dim3((int)grid8), dim3(256), 0, st, x, (bf16_t*)cols, gm);
        else hipLaunchKernelGGL((im2col_cl8_kernel<float>), dim3((int)grid8), dim3(256), 0, st, x, (float*)cols, gm);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    const long total = (long)B * gm.Ho * gm.Wo * Kp;
    if (Kp % 8 == 0 && total / 8 < (1L << 31) && ((uintptr_t)cols % 16 == 0)) {
        long grid8 = (total / 8 + 255) / 256;
        grid8 = grid8 > 16384 ? 16384 : grid8;
        if (dtype == VITK_BF16) hipLaunchKernelGGL((im2col_any8_kernel<bf16_t>), dim3((int)grid8), dim3(256), 0, st, x, (bf16_t*)cols, gm);
        else hipLaunchKernelGGL((im2col_any8_kernel<float>), dim3((int)grid8), dim3(256), 0, st, x, (float*)cols, gm);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    long grid = (total + 255) / 256;
    grid = grid > 8192 ? 8192 : grid;
    if (dtype == VITK_BF16) hipLaunchKernelGGL((im2col_any_kernel<bf16_t>), dim3((int)grid), dim3(256), 0, st, x, (bf16_t*)cols, gm);
    else if (dtype == VITK_F32) hipLaunchKernelGGL((im2col_any_kernel<float>), dim3((int)grid), dim3(256), 0, st, x, (float*)cols, gm);
    else return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_col2im_conv(const void* dcols, float* dx, int B, int H, int W, int C, int KH, int KW, int stride,
                                int pad, int Kp, int dtype, void* stream) {
    if (!dcols || !dx) return VITK_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Kp < KH * KW * C) return VITK_ERR_SHAPE;
    ColGeom gm;
    gm.B = B; gm.H = H; gm.W = W; gm.C = C; gm.KH = KH; gm.KW = KW; gm.stride = stride; gm.pad = pad; gm.Kp = Kp;
    gm.Ho = (H + 2 * pad - KH) / stride + 1;
    gm.Wo = (W + 2 * pad - KW) / stride + 1;
    gm.sb = gm.sh = gm.sw = gm.sc = 0;
    if (gm.Ho <= 0 || gm.Wo <= 0) return VITK_ERR_SHAPE;
    const long total = (long)B * H * W * C;
    long grid = (total + 255) / 256;
    grid = grid > 8192 ? 8192 : grid;
    hipStream_t st = (hipStream_t)stream;
    if (C % 8 == 0 && Kp % 8 == 0 && ((uintptr_t)dcols % 16 == 0) && ((uintptr_t)dx % 16 == 0) &&
        (dtype == VITK_BF16 || dtype == VITK_F32)) {
        long grid8 = (total / 8 + 255) / 256;
        grid8 = grid8 > 16384 ? 16384 : grid8;
        if (dtype == VITK_BF16) hipLaunchKernelGGL((col2im_cl8_kernel<bf16_t>), dim3((int)grid8), dim3(256), 0, st, (const bf16_t*)dcols, dx, gm);
        else hipLaunchKernelGGL((col2im_cl8_kernel<float>), dim3((int)grid8), dim3(256), 0, st, (const float*)dcols, dx, gm);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == VITK_BF16) hipLaunchKernelGGL((col2im_cl_kernel<bf16_t>), dim3((int)grid), dim3(256), 0, st, (const bf16_t*)dcols, dx, gm);
    else if (dtype == VITK_F32) hipLaunchKernelGGL((col2im_cl_kernel<float>), dim3((int)grid), dim3(256), 0, st, (const float*)dcols, dx, gm);
    else return VITK_ERR_DTYPE;
    VITK_CHECK_LAUNCH();
    return 0;
}
