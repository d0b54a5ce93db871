// Fused softmax(Q.K^T/sqrt(d)).V for head size 64 in bf16 (gfx950), forward and backward.
//
// Heads are addressed in place inside token-major matrices (row stride ld, head h at column
// h*64), so no head split / merge copy exists.  Work decomposition: one workgroup = 64 rows
// of one (image, head); each of its 4 wavefronts owns 16 of them and streams the other
// sequence axis in tiles of 32 rows staged in LDS (160-byte rows: conflict free for both the
// row reads, ds_read_b128, and the transposed reads, ds_read_b64_tr_b16).
//
// The 16-row axis a wave owns always sits on the MFMA column (= lane & 15) index, so every
// per-row quantity (running max, sum, LSE, delta) is one scalar per lane, and the score
// accumulators are, as they stand, the B operand of the next product: within one MFMA the 32
// reduction slots are (lane group g, element e) <-> tile row 4g + (e&3) + 16(e>>2), which is
// also how the transposed LDS reads deliver the other operand.
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int AT_ROW = 160;            // bytes per LDS tile row (128 data + 32 pad)
constexpr int AT_TILE = 32 * AT_ROW;   // 32-row tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    const bf16_t* o; const bf16_t* d_o;
    bf16_t* out;            // fwd: o ; bwd: unused
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    float* lse; float* delta;
    float* o32;             // optional fp32 copy of o (fwd: written, bwd: read for delta)
    int B, H, N;
    long ld_qkv, ld_o, ld_dqkv;
    float scale;
};

__device__ __forceinline__ void load_tile32(char* lds, const bf16_t* base, long ld, int r0, int nrows, int tid) {
    const int row = tid >> 3, ch = tid & 7;
    int r = r0 + row;
    r = r < nrows ? r : nrows - 1;
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + (long)r * ld + ch * 8);
    *reinterpret_cast<u32x4*>(lds + row * AT_ROW + ch * 16) = v;
}
// 16 tile rows from rbase as MFMA rows/cols, reduction over the 32 columns [32s, 32s+32).
__device__ __forceinline__ bf16x8 row_frag(const char* lds, int rbase, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (rbase + (lane & 15)) * AT_ROW + (4 * s + (lane >> 4)) * 16);
}
// 16 tile columns from d0 as MFMA rows/cols, reduction over the tile's 32 rows.
__device__ __forceinline__ bf16x8 tr_frag32(const char* lds, int d0, int lane) {
    const int l15 = lane & 15, gq = lane >> 4;
    const char* a0 = lds + (4 * gq + (l15 >> 2)) * AT_ROW + (d0 + 4 * (l15 & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0 + 16 * AT_ROW));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                  (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}
__device__ __forceinline__ bf16x8 load_frag_global(const bf16_t* rowptr, int s, int g) {
    return *reinterpret_cast<const bf16x8*>(rowptr + 32 * s + 8 * g);
}
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v, float mul) {
    bf16x4 o = {(bf16_t)(v[0] * mul), (bf16_t)(v[1] * mul), (bf16_t)(v[2] * mul), (bf16_t)(v[3] * mul)};
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Ks[AT_TILE];
    __shared__ __attribute__((aligned(16))) char Vs[AT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const bf16_t* qb = p.q + tok0 * p.ld_qkv + h * DH;
    const bf16_t* kb = p.k + tok0 * p.ld_qkv + h * DH;
    const bf16_t* vb = p.v + tok0 * p.ld_qkv + h * DH;
    const int qi = blockIdx.x * 64 + wid * 16 + l15;
    const int qr = qi < p.N ? qi : p.N - 1;
    bf16x8 qf[2];
    qf[0] = load_frag_global(qb + (long)qr * p.ld_qkv, 0, g);
    qf[1] = load_frag_global(qb + (long)qr * p.ld_qkv, 1, g);

    const float sc = p.scale * LOG2E;
    float m = -INFINITY, lsum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntile = (p.N + 31) >> 5;
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile32(Ks, kb, p.ld_qkv, kt * 32, p.N, tid);
        load_tile32(Vs, vb, p.ld_qkv, kt * 32, p.N, tid);
        __syncthreads();
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 0, s, lane), qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 16, s, lane), qf[s], s1, 0, 0, 0);
        }
        // s0[r]: key kt*32 + 4g + r ; s1[r]: key kt*32 + 16 + 4g + r ; query = l15
        float sv[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = kt * 32 + 4 * g + r;
            sv[r] = (k0 < p.N) ? s0[r] * sc : -INFINITY;
            sv[4 + r] = (k0 + 16 < p.N) ? s1[r] * sc : -INFINITY;
            tmax = fmaxf(tmax, fmaxf(sv[r], sv[4 + r]));
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);           // finite: every tile holds >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            sv[r] = __builtin_amdgcn_exp2f(sv[r] - mn);
            ps += sv[r];
        }
        lsum = lsum * alpha + ps;
        m = mn;
        const bf16x8 pb = pack8(sv, sv + 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] *= alpha;
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Vs, 16 * dt, lane), pb, o[dt], 0, 0, 0);
        }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < p.N) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = p.out + (tok0 + qi) * p.ld_o + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt + 4 * g, o[dt], inv);
        if (p.o32) {
            float* o32row = p.o32 + (tok0 + qi) * p.ld_o + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4*>(o32row + 16 * dt + 4 * g) = o[dt] * inv;
        }
        if (g == 0) p.lse[(long)bh * p.N + qi] = (m + __builtin_amdgcn_logf(lsum)) * LN2;
    }
}

// ------------------------------------------------------------------------------ delta
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]; 8 lanes per (token, head), 8 elements per lane.
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(AttnArgs p) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long unit = gid >> 3;
    const int sub = (int)(gid & 7);
    const long total = (long)p.B * p.N * p.H;
    float acc = 0.f;
    long row = 0; int h = 0;
    if (unit < total) {
        row = unit / p.H; h = (int)(unit - row * p.H);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(p.d_o + row * p.ld_o + h * DH + sub * 8);
        if (p.o32) {
            const float* a = p.o32 + row * p.ld_o + h * DH + sub * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += a0[i] * (float)c[i] + a1[i] * (float)c[4 + i];
        } else {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.o + row * p.ld_o + h * DH + sub * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += (float)a[i] * (float)c[i];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (unit < total && sub == 0) {
        const long b = row / p.N, qi = row - b * p.N;
        p.delta[(b * p.H + h) * p.N + qi] = acc;
    }
}

// ------------------------------------------------------------------------------ dK, dV
// Wave owns 16 keys (on the lane index) and streams the queries.
__global__ __launch_bounds__(256) void attn_bwd_dkdv_bf16_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Qs[AT_TILE];
    __shared__ __attribute__((aligned(16))) char Os[AT_TILE];   // dO tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const bf16_t* qb = p.q + tok0 * p.ld_qkv + h * DH;
    const bf16_t* kb = p.k + tok0 * p.ld_qkv + h * DH;
    const bf16_t* vb = p.v + tok0 * p.ld_qkv + h * DH;
    const bf16_t* dob = p.d_o + tok0 * p.ld_o + h * DH;
    const float* lse = p.lse + (long)bh * p.N;
    const float* delta = p.delta + (long)bh * p.N;
    const int ki = blockIdx.x * 64 + wid * 16 + l15;
    const int kr = ki < p.N ? ki : p.N - 1;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kf[s] = load_frag_global(kb + (long)kr * p.ld_qkv, s, g);
        vf[s] = load_frag_global(vb + (long)kr * p.ld_qkv, s, g);
    }
    const float sc = p.scale * LOG2E;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntile = (p.N + 31) >> 5;
    for (int qt = 0; qt < ntile; ++qt) {
        __syncthreads();
        load_tile32(Qs, qb, p.ld_qkv, qt * 32, p.N, tid);
        load_tile32(Os, dob, p.ld_o, qt * 32, p.N, tid);
        __syncthreads();
        float pv[8], dsv[8];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qs, 16 * qq, s, lane), kf[s], sa, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Os, 16 * qq, s, lane), vf[s], da, 0, 0, 0);
            }
            // sa[r]: query qt*32 + 16qq + 4g + r, key = l15
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qt * 32 + 16 * qq + 4 * g + r;
                const bool ok = qi < p.N;
                const float l2 = ok ? lse[qi] * LOG2E : 0.f;
                const float dl = ok ? delta[qi] : 0.f;
                const float pr = ok ? __builtin_amdgcn_exp2f(sa[r] * sc - l2) : 0.f;
                pv[4 * qq + r] = pr;
                dsv[4 * qq + r] = pr * (da[r] - dl) * p.scale;
            }
        }
        const bf16x8 pb = pack8(pv, pv + 4);
        const bf16x8 dsb = pack8(dsv, dsv + 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Os, 16 * dt, lane), pb, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Qs, 16 * dt, lane), dsb, dk[dt], 0, 0, 0);
        }
    }
    if (ki < p.N) {
        bf16_t* dkr = p.dk + (tok0 + ki) * p.ld_dqkv + h * DH;
        bf16_t* dvr = p.dv + (tok0 + ki) * p.ld_dqkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            store4(dkr + 16 * dt + 4 * g, dk[dt], 1.0f);
            store4(dvr + 16 * dt + 4 * g, dv[dt], 1.0f);
        }
    }
}

// ------------------------------------------------------------------------------ dQ
// Wave owns 16 queries (on the lane index) and streams the keys.
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Ks[AT_TILE];
    __shared__ __attribute__((aligned(16))) char Vs[AT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const bf16_t* qb = p.q + tok0 * p.ld_qkv + h * DH;
    const bf16_t* kb = p.k + tok0 * p.ld_qkv + h * DH;
    const bf16_t* vb = p.v + tok0 * p.ld_qkv + h * DH;
    const bf16_t* dob = p.d_o + tok0 * p.ld_o + h * DH;
    const int qi = blockIdx.x * 64 + wid * 16 + l15;
    const int qr = qi < p.N ? qi : p.N - 1;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qf[s] = load_frag_global(qb + (long)qr * p.ld_qkv, s, g);
        dof[s] = load_frag_global(dob + (long)qr * p.ld_o, s, g);
    }
    const float sc = p.scale * LOG2E;
    const float l2 = p.lse[(long)bh * p.N + qr] * LOG2E;
    const float dl = p.delta[(long)bh * p.N + qr];
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntile = (p.N + 31) >> 5;
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile32(Ks, kb, p.ld_qkv, kt * 32, p.N, tid);
        load_tile32(Vs, vb, p.ld_qkv, kt * 32, p.N, tid);
        __syncthreads();
        float dsv[8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 16 * kk, s, lane), qf[s], sa, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, 16 * kk, s, lane), dof[s], da, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kt * 32 + 16 * kk + 4 * g + r;
                const float pr = (kidx < p.N) ? __builtin_amdgcn_exp2f(sa[r] * sc - l2) : 0.f;
                dsv[4 * kk + r] = pr * (da[r] - dl) * p.scale;
            }
        }
        const bf16x8 dsb = pack8(dsv, dsv + 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Ks, 16 * dt, lane), dsb, dq[dt], 0, 0, 0);
    }
    if (qi < p.N) {
        bf16_t* dqr = p.dq + (tok0 + qi) * p.ld_dqkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(dqr + 16 * dt + 4 * g, dq[dt], 1.0f);
    }
}

}  // namespace

int vitk_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* o32, float* lse, int B, int H,
                       int N, long ld_qkv, long ld_o, float scale, hipStream_t st) {
    if (ld_qkv % 8 || ld_o % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return VITK_ERR_SHAPE;
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
    a.out = (bf16_t*)o; a.lse = lse; a.o32 = o32;
    a.B = B; a.H = H; a.N = N; a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.scale = scale;
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3(vitk_ceil_div(N, 64), B * H), dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

int vitk_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const float* o32, const void* d_o,
                       const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H,
                       int N, long ld_qkv, long ld_o, long ld_dqkv, float scale, hipStream_t st) {
    if (ld_qkv % 8 || ld_o % 8 || ld_dqkv % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq |
         (uintptr_t)dk | (uintptr_t)dv) & 15)
        return VITK_ERR_SHAPE;
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
    a.o = (const bf16_t*)o; a.d_o = (const bf16_t*)d_o;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.o32 = const_cast<float*>(o32);
    a.B = B; a.H = H; a.N = N; a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.ld_dqkv = ld_dqkv; a.scale = scale;
    const long units = (long)B * N * H * 8;
    hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(vitk_ceil_div(units, 256)), dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    const dim3 grid(vitk_ceil_div(N, 64), B * H);
    hipLaunchKernelGGL(attn_bwd_dkdv_bf16_kernel, grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
