// Window attention of the Swin blocks (reference models/Swin/swin_RKR.py:178-215 WindowAttention.forward
// with the cyclic shift, window partition / reverse and shift mask of SwinTransformerBlock `:281-337`):
//
//     S = scale * Q.K^T + B[rel(i, j)] + (region(i) != region(j) ? -100 : 0);   O = softmax(S).V
//
// per (window, head); windows of ws x ws tokens (at most 64), head dimension 32.
//
// The qkv projection is per token, so roll, window_partition, window_reverse and the roll back never
// need to touch memory: the kernel reads Q, K, V of a window's tokens straight from the token-major
// [B, H*W, 3, heads, 32] projection output at their un-shifted positions and writes O to
// [B, H*W, heads*32] likewise.  The relative-position bias is gathered from its [(2ws-1)^2, heads]
// table by index arithmetic, the shift mask is computed from the token's coordinates in the shifted
// frame.  One wavefront per (window, head): lane i owns query row i (online softmax over the keys);
// K and V rows are broadcast reads from LDS.  fp32 arithmetic for both storage types.  A few percent
// of a Swin block's FLOPs live here, the rest in the GEMMs, so this first kernel stays on the VALU.
#include "common.h"

namespace {

constexpr int WD = 32;            // head dimension
constexpr int WN = 64;            // most tokens per window (one lane each)
constexpr int WROW = WD + 4;      // LDS row stride in floats: 144 bytes, 16-byte aligned, spreads banks
constexpr int WTBL = 225;         // (2*8-1)^2 entries of the bias table for one head

struct WinArgs {
    const void* qkv; const float* table; void* out; float* lse;
    const void* dout; void* dqkv; float* dtable;
    int B, H, W, nH, ws, shift, groups;     // groups: workgroups per head
    float scale;
};

struct WinTok {
    long tok;          // token index in [0, B*H*W)
    int rid;           // region of the shift mask (0 when shift == 0)
    int r, c;          // position inside the window
};

// Token `n` of window `win` (b, wh, ww in window_partition order): its coordinates in the shifted
// frame give the mask region (swin_RKR.py:284-300), rolled back by `shift` they give its address.
__device__ __forceinline__ WinTok win_token(const WinArgs& p, int win, int n) {
    const int nWw = p.W / p.ws, nW = (p.H / p.ws) * nWw;
    const int b = win / nW, wi = win - b * nW;
    const int wh = wi / nWw, ww = wi - wh * nWw;
    WinTok t;
    t.r = n / p.ws; t.c = n - t.r * p.ws;
    const int hs = wh * p.ws + t.r, wsf = ww * p.ws + t.c;
    int ho = hs + p.shift, wo = wsf + p.shift;
    ho = ho >= p.H ? ho - p.H : ho;
    wo = wo >= p.W ? wo - p.W : wo;
    t.tok = ((long)b * p.H + ho) * p.W + wo;
    t.rid = 0;
    if (p.shift > 0) {
        const int rh = hs < p.H - p.ws ? 0 : (hs < p.H - p.shift ? 1 : 2);
        const int rw = wsf < p.W - p.ws ? 0 : (wsf < p.W - p.shift ? 1 : 2);
        t.rid = 3 * rh + rw;
    }
    return t;
}

template <typename T>
__device__ __forceinline__ void load_row(const T* p, float* o) { load_vec<T, WD>(p, o); }

__device__ __forceinline__ void lds_put_row(float* row, const float* v) {
#pragma unroll
    for (int d = 0; d < WD; d += 4) *reinterpret_cast<f32x4*>(row + d) = f32x4{v[d], v[d + 1], v[d + 2], v[d + 3]};
}
__device__ __forceinline__ float dot_row(const float* row, const float* v) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < WD; d += 4) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(row + d);
        s += k[0] * v[d] + k[1] * v[d + 1] + k[2] * v[d + 2] + k[3] * v[d + 3];
    }
    return s;
}
__device__ __forceinline__ void axpy_row(const float* row, float a, float* acc) {
#pragma unroll
    for (int d = 0; d < WD; d += 4) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(row + d);
        acc[d] += a * k[0]; acc[d + 1] += a * k[1]; acc[d + 2] += a * k[2]; acc[d + 3] += a * k[3];
    }
}

template <typename T>
__global__ __launch_bounds__(64) void win_attn_fwd_kernel(WinArgs p) {
    __shared__ __attribute__((aligned(16))) float Ks[WN][WROW];
    __shared__ __attribute__((aligned(16))) float Vs[WN][WROW];
    __shared__ float tbl[WTBL];
    __shared__ int rids[WN];
    const int lane = threadIdx.x;
    const int h = blockIdx.x % p.nH, g0 = blockIdx.x / p.nH;
    const int N = p.ws * p.ws, C = p.nH * WD, span = 2 * p.ws - 1;
    const int nwin = p.B * (p.H / p.ws) * (p.W / p.ws);
    for (int e = lane; e < span * span; e += 64) tbl[e] = p.table[(long)e * p.nH + h];
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    T* out = reinterpret_cast<T*>(p.out);
    for (int win = g0; win < nwin; win += p.groups) {
        __syncthreads();                                   // previous window's K/V are no longer read
        WinTok me = win_token(p, win, lane < N ? lane : 0);
        float q[WD];
        if (lane < N) {
            const T* base = qkv + me.tok * (3L * C) + h * WD;
            float kv[WD];
            load_row(base, q);
            load_row(base + C, kv);
            lds_put_row(Ks[lane], kv);
            load_row(base + 2 * C, kv);
            lds_put_row(Vs[lane], kv);
            rids[lane] = me.rid;
#pragma unroll
            for (int d = 0; d < WD; ++d) q[d] *= p.scale;
        }
        __syncthreads();
        if (lane < N) {
            // online softmax: one pass over the keys, running maximum m and sum l
            float m = -INFINITY, l = 0.f, o[WD];
#pragma unroll
            for (int d = 0; d < WD; ++d) o[d] = 0.f;
            for (int j = 0; j < N; ++j) {
                const int rj = j / p.ws, cj = j - rj * p.ws;
                float v = dot_row(Ks[j], q) + tbl[(me.r - rj + p.ws - 1) * span + (me.c - cj + p.ws - 1)];
                if (rids[j] != me.rid) v += -100.0f;
                const float mn = fmaxf(m, v);
                const float corr = __expf(m - mn), e = __expf(v - mn);
                m = mn;
                l = l * corr + e;
#pragma unroll
                for (int d = 0; d < WD; ++d) o[d] *= corr;
                axpy_row(Vs[j], e, o);
            }
            const float inv = 1.0f / l;
#pragma unroll
            for (int d = 0; d < WD; ++d) o[d] *= inv;
            store_vec<WD>(out + me.tok * C + h * WD, o);
            p.lse[((long)win * p.nH + h) * N + lane] = m + __logf(l);
        }
    }
}

// Backward.  Two passes over the window's (query, key) pairs, both recomputing P from the saved LSE,
// run side by side by the two wavefronts of a workgroup: wave 0, lane i as QUERY row i, forms
// dS_ij = P_ij (dO_i.V_j - delta_i) and accumulates dQ_i and the bias-table gradient; wave 1, lane j as
// KEY row j, forms the same dS_ij again and accumulates dK_j, dV_j.  Recomputing the two dot products
// costs a quarter more arithmetic than keeping P and dS as [64][64] matrices in LDS, but those
// matrices would halve the number of resident wavefronts, and the kernel lives on latency hiding.
template <typename T>
__global__ __launch_bounds__(128) void win_attn_bwd_kernel(WinArgs p) {
    __shared__ __attribute__((aligned(16))) float Ks[WN][WROW];
    __shared__ __attribute__((aligned(16))) float Vs[WN][WROW];
    __shared__ __attribute__((aligned(16))) float Qs[WN][WROW];
    __shared__ __attribute__((aligned(16))) float Gs[WN][WROW];     // dO rows
    __shared__ float tbl[WTBL], dtbl[WTBL], lses[WN], deltas[WN];
    __shared__ int rids[WN];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.x % p.nH, g0 = blockIdx.x / p.nH;
    const int N = p.ws * p.ws, C = p.nH * WD, span = 2 * p.ws - 1;
    const int nwin = p.B * (p.H / p.ws) * (p.W / p.ws);
    for (int e = threadIdx.x; e < span * span; e += 128) { tbl[e] = p.table[(long)e * p.nH + h]; dtbl[e] = 0.f; }
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const T* outp = reinterpret_cast<const T*>(p.out);
    const T* dout = reinterpret_cast<const T*>(p.dout);
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    for (int win = g0; win < nwin; win += p.groups) {
        __syncthreads();
        const WinTok me = win_token(p, win, lane < N ? lane : 0);
        float a[WD], b[WD];              // wave 0: q, dO of row `lane`; wave 1: k, v of row `lane`
        float delta = 0.f, lse = 0.f;
        if (lane < N) {
            const T* base = qkv + me.tok * (3L * C) + h * WD;
            if (wid == 0) {
                float t[WD];
                load_row(base, a);
                lds_put_row(Qs[lane], a);
                load_row(dout + me.tok * C + h * WD, b);
                lds_put_row(Gs[lane], b);
                load_row(outp + me.tok * C + h * WD, t);
#pragma unroll
                for (int d = 0; d < WD; ++d) delta += b[d] * t[d];
                lse = p.lse[((long)win * p.nH + h) * N + lane];
                rids[lane] = me.rid;
                lses[lane] = lse;
                deltas[lane] = delta;
            } else {
                load_row(base + C, a);
                lds_put_row(Ks[lane], a);
                load_row(base + 2 * C, b);
                lds_put_row(Vs[lane], b);
            }
        }
        __syncthreads();
        if (lane < N) {
            T* base = dqkv + me.tok * (3L * C) + h * WD;
            float acc[WD];
#pragma unroll
            for (int d = 0; d < WD; ++d) acc[d] = 0.f;
            if (wid == 0) {
                for (int j = 0; j < N; ++j) {
                    const int rj = j / p.ws, cj = j - rj * p.ws;
                    const int idx = (me.r - rj + p.ws - 1) * span + (me.c - cj + p.ws - 1);
                    float s = p.scale * dot_row(Ks[j], a) + tbl[idx];
                    if (rids[j] != me.rid) s += -100.0f;
                    const float pr = __expf(s - lse);
                    const float ds = pr * (dot_row(Vs[j], b) - delta);
                    axpy_row(Ks[j], ds, acc);
                    atomicAdd(dtbl + idx, ds);                 // distinct idx across the lanes of one j
                }
#pragma unroll
                for (int d = 0; d < WD; ++d) acc[d] *= p.scale;
                store_vec<WD>(base, acc);
            } else {
                float dv[WD];
#pragma unroll
                for (int d = 0; d < WD; ++d) dv[d] = 0.f;
                for (int i = 0; i < N; ++i) {
                    const int ri = i / p.ws, ci = i - ri * p.ws;
                    float s = p.scale * dot_row(Qs[i], a) + tbl[(ri - me.r + p.ws - 1) * span + (ci - me.c + p.ws - 1)];
                    if (rids[i] != me.rid) s += -100.0f;
                    const float pr = __expf(s - lses[i]);
                    const float ds = pr * (dot_row(Gs[i], b) - deltas[i]);
                    axpy_row(Qs[i], ds, acc);
                    axpy_row(Gs[i], pr, dv);
                }
#pragma unroll
                for (int d = 0; d < WD; ++d) acc[d] *= p.scale;
                store_vec<WD>(base + C, acc);
                store_vec<WD>(base + 2 * C, dv);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < span * span; e += 128) atomicAdd(p.dtable + (long)e * p.nH + h, dtbl[e]);
}

int win_check(const WinArgs& a, int dh, int dtype) {
    if (dh != WD) return VITK_ERR_UNSUPPORTED;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0 || a.nH <= 0 || a.ws <= 0) return VITK_ERR_SHAPE;
    if (a.ws * a.ws > WN || a.H % a.ws || a.W % a.ws) return VITK_ERR_SHAPE;
    if (a.shift < 0 || a.shift >= a.ws) return VITK_ERR_SHAPE;
    if (dtype != VITK_BF16 && dtype != VITK_F32) return VITK_ERR_DTYPE;
    return 0;
}

int win_groups(const WinArgs& a) {
    const long nwin = (long)a.B * (a.H / a.ws) * (a.W / a.ws);
    // enough workgroups to fill the chip several times over, few enough that the bias-gradient
    // table of a workgroup is flushed once per many windows
    const long want = 4096 / a.nH > 0 ? 4096 / a.nH : 1;
    return (int)(nwin < want ? nwin : want);
}

}  // namespace

extern "C" int vitk_window_attn_fwd(const void* qkv, const float* bias_table, void* out, float* lse, int B, int H,
                                    int W, int heads, int dh, int window, int shift, float scale, int dtype,
                                    void* stream) {
    if (!qkv || !bias_table || !out || !lse) return VITK_ERR_NULL;
    WinArgs a{};
    a.qkv = qkv; a.table = bias_table; a.out = out; a.lse = lse;
    a.B = B; a.H = H; a.W = W; a.nH = heads; a.ws = window; a.shift = shift; a.scale = scale;
    const int rc = win_check(a, dh, dtype);
    if (rc) return rc;
    if (((uintptr_t)qkv | (uintptr_t)out) & 15) return VITK_ERR_SHAPE;
    a.groups = win_groups(a);
    const dim3 grid(a.groups * heads);
    if (dtype == VITK_BF16) hipLaunchKernelGGL(win_attn_fwd_kernel<bf16_t>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(win_attn_fwd_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_window_attn_bwd(const void* qkv, const float* bias_table, const void* out, const void* d_out,
                                    const float* lse, void* dqkv, float* dbias_table, int B, int H, int W, int heads,
                                    int dh, int window, int shift, float scale, int dtype, void* stream) {
    if (!qkv || !bias_table || !out || !d_out || !lse || !dqkv || !dbias_table) return VITK_ERR_NULL;
    WinArgs a{};
    a.qkv = qkv; a.table = bias_table; a.out = const_cast<void*>(out); a.lse = const_cast<float*>(lse);
    a.dout = d_out; a.dqkv = dqkv; a.dtable = dbias_table;
    a.B = B; a.H = H; a.W = W; a.nH = heads; a.ws = window; a.shift = shift; a.scale = scale;
    const int rc = win_check(a, dh, dtype);
    if (rc) return rc;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dqkv) & 15) return VITK_ERR_SHAPE;
    a.groups = win_groups(a);
    const dim3 grid(a.groups * heads);
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(win_attn_bwd_kernel<bf16_t>, grid, dim3(128), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(win_attn_bwd_kernel<float>, grid, dim3(128), 0, (hipStream_t)stream, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
