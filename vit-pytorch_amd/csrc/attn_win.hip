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
    int replicas;                           // copies of the bias-table gradient the workgroups spread their flushes over
    float scale;
#ifdef VITK_WIN_PROF
    int abl;                                // instrumented build: parts of the backward left out (tools/win_prof.py)
#endif
};

struct WinTok {
    long tok;          // token index in [0, B*H*W)
    int rid;           // region of the shift mask (0 when shift == 0)
    int r, c;          // position inside the window
};

// Token at row r, column c of window (b, wh, ww): its coordinates in the shifted frame give the mask
// region (swin_RKR.py:284-300), rolled back by `shift` they give its address.
__device__ __forceinline__ WinTok win_token_at(const WinArgs& p, int b, int wh, int ww, int r, int c) {
    WinTok t;
    t.r = r; t.c = c;
    const int hs = wh * p.ws + r, wsf = ww * p.ws + c;
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
// Window `win` in window_partition order -> (image, window row, window column); wave-uniform.
__device__ __forceinline__ void win_coords(const WinArgs& p, int win, int& b, int& wh, int& ww) {
    const int nWw = p.W / p.ws, nW = (p.H / p.ws) * nWw;
    b = win / nW;
    const int wi = win - b * nW;
    wh = wi / nWw;
    ww = wi - wh * nWw;
}
// Token `n` of window `win`.
__device__ __forceinline__ WinTok win_token(const WinArgs& p, int win, int n) {
    int b, wh, ww;
    win_coords(p, win, b, wh, ww);
    const int r = n / p.ws;
    return win_token_at(p, b, wh, ww, r, n - r * p.ws);
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
    float* dt = p.dtable + (long)(g0 % p.replicas) * span * span * p.nH;
    for (int e = threadIdx.x; e < span * span; e += 128) atomicAdd(dt + (long)e * p.nH + h, dtbl[e]);
}

// ---------------------------------------------------------------------------------------- MFMA (bf16)
// The same computation on v_mfma_f32_16x16x32_bf16, one wavefront per (window, head), the window
// padded to 64 tokens.  mfma(a, b, c): lane l = 16 g + c supplies row c of each operand, reduction
// elements 8g..8g+7 (a 16-byte read from a row with the reduction index contiguous), and receives
// out[a-row 4g+r][b-row c], r = 0..3.  Operands are staged in LDS once per window in the orientations
// the products need: rows [token][d] and the transposes [d][token].
//
// Forward.  S^T tile = mfma(K rows j, Q rows i) puts the scores of query i = c against keys
// 16 jt + 4g + r in lane (g, c): the softmax row of a query lives in the four lanes that share c
// (two xor-shuffles).  Those registers are already the b-operand of O^T = V^T . P^T, if the 32
// reduction slots of that product are numbered (g, e) <-> key 16 (2 kk + e/4) + 4g + e%4 for both
// operands; the V^T fragment is then two 8-byte reads.  P never touches LDS.
constexpr int MR = 40;            // bf16 per LDS row of a [token][d] matrix (80 bytes)
constexpr int MT = 72;            // bf16 per LDS row of a [d][token] matrix (144 bytes)

__device__ __forceinline__ bf16x8 frag_row(const bf16_t* m, int row, int g) {
    return *reinterpret_cast<const bf16x8*>(m + row * MR + 8 * g);
}
// reduction slots (g, e) = tokens 32 kk + 4g + e%4 + 16 (e/4) of row `row` of a [d][token] matrix
__device__ __forceinline__ bf16x8 frag_tok(const bf16_t* mt, int row, int kk, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(mt + row * MT + 32 * kk + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(mt + row * MT + 32 * kk + 16 + 4 * g);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// row `lane` of a token-major matrix (32 bf16 from global) into the [token][d] image, zeros beyond the window
__device__ __forceinline__ void stage_row(bf16_t* m, int lane, const bf16_t* src, bool valid, bf16x8 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = valid ? *reinterpret_cast<const bf16x8*>(src + 8 * q) : bf16x8{};
        *reinterpret_cast<bf16x8*>(m + lane * MR + 8 * q) = v[q];
    }
}
__device__ __forceinline__ void stage_col(bf16_t* mt, int lane, const bf16x8 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) mt[(8 * q + e) * MT + lane] = v[q][e];
}

__global__ __launch_bounds__(64) void win_attn_fwd_mfma_kernel(WinArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t Qr[WN * MR];
    __shared__ __attribute__((aligned(16))) bf16_t Kr[WN * MR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[WD * MT];
    __shared__ float tbl[WTBL];
    __shared__ int rids[WN];
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % p.nH, g0 = blockIdx.x / p.nH;
    const int N = p.ws * p.ws, C = p.nH * WD, span = 2 * p.ws - 1;
    const int nwin = p.B * (p.H / p.ws) * (p.W / p.ws);
    for (int e = lane; e < span * span; e += 64) tbl[e] = p.table[(long)e * p.nH + h];
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
    // this lane's 16 keys j = 16 jt + 4g + r: position code and validity do not depend on the window
    int kb[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            kb[jt][r] = j < N ? (j / p.ws) * span + j % p.ws : -1;
        }
    const int off = (p.ws - 1) * (span + 1);
    // integer divisions by the window size happen here, once per workgroup: this lane's own token
    // (staging) and its four query tokens 16 t + c
    const int lrow = (lane < N ? lane : 0) / p.ws, lcol = (lane < N ? lane : 0) % p.ws;
    int trow[4], tcol[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 16 * t + c < N ? 16 * t + c : 0;
        trow[t] = i / p.ws; tcol[t] = i % p.ws;
    }
    for (int win = g0; win < nwin; win += p.groups) {
        __syncthreads();
        int wb, wh, ww;
        win_coords(p, win, wb, wh, ww);
        {
            const bool valid = lane < N;
            const WinTok me = win_token_at(p, wb, wh, ww, lrow, lcol);
            const bf16_t* base = qkv + me.tok * (3L * C) + h * WD;
            bf16x8 v[4];
            stage_row(Qr, lane, base, valid, v);
            stage_row(Kr, lane, base + C, valid, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = valid ? *reinterpret_cast<const bf16x8*>(base + 2 * C + 8 * q) : bf16x8{};
            stage_col(Vt, lane, v);
            rids[lane] = valid ? me.rid : -1;
        }
        __syncthreads();
        bf16x8 kf[4], vf[2][2];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) kf[jt] = frag_row(Kr, 16 * jt + c, g);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) vf[dt][kk] = frag_tok(Vt, 16 * dt + c, kk, g);
        int krid[4][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) krid[jt][r] = rids[16 * jt + 4 * g + r];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = 16 * it + c;
            if (16 * it >= N) break;                               // wave-uniform: no query in this tile
            const bf16x8 qf = frag_row(Qr, i, g);
            const int qa = trow[it] * span + tcol[it] + off, qrid = rids[i];
            f32x4 s[4];
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = -INFINITY;
                    if (kb[jt][r] >= 0 && i < N) {
                        v = s[jt][r] * p.scale + tbl[qa - kb[jt][r]];
                        if (krid[jt][r] != qrid) v += -100.0f;
                    }
                    s[jt][r] = v;
                    m = fmaxf(m, v);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (i >= N) m = 0.f;
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[jt][r] - m);
                    s[jt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 pf = {(bf16_t)s[2 * kk][0], (bf16_t)s[2 * kk][1], (bf16_t)s[2 * kk][2], (bf16_t)s[2 * kk][3],
                                   (bf16_t)s[2 * kk + 1][0], (bf16_t)s[2 * kk + 1][1], (bf16_t)s[2 * kk + 1][2],
                                   (bf16_t)s[2 * kk + 1][3]};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt][kk], pf, o[dt], 0, 0, 0);
            }
            if (i < N) {
                const float inv = 1.0f / l;
                const WinTok qt = win_token_at(p, wb, wh, ww, trow[it], tcol[it]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x4 ov = {(bf16_t)(o[dt][0] * inv), (bf16_t)(o[dt][1] * inv), (bf16_t)(o[dt][2] * inv),
                                       (bf16_t)(o[dt][3] * inv)};
                    *reinterpret_cast<bf16x4*>(out + qt.tok * C + h * WD + 16 * dt + 4 * g) = ov;
                }
                if (g == 0) p.lse[((long)win * p.nH + h) * N + i] = m + __logf(l);
            }
        }
    }
}

// Backward on MFMA.  The [64][64] score-shaped matrices are produced twice, once per orientation, because in
// each orientation the accumulator registers ARE the b-operand of the products that follow (as P is in the
// forward pass), so neither P nor dS ever touches LDS:
//   wave 0, out[i][j] tiles  S = Q.K^T, dP = dO.V^T: the lane of key j = 16 jt + c holds queries 16 it + 4g + r
//           dV^T[d][j] = dO^T[d][i] . P^T[j][i],   dK^T[d][j] = Q^T[d][i] . dS^T[j][i]      (reduction over queries)
//   wave 1, out[j][i] tiles  S^T, dP^T: the lane of query i = 16 it + c holds keys 16 jt + 4g + r
//           dQ^T[d][i] = K^T[d][j] . dS[i][j]                                               (reduction over keys)
//           and the bias-table gradient, summed per lane over the workgroup's windows.
// The reduction slots (g, e) of those products stand for token 16 (2 kk + e/4) + 4g + e%4 (frag_tok).
// Each wave reads the row fragments it keeps for a whole window (Q, dO / K, V) straight from global memory
// (16 bytes per lane: row 16 t + c, elements 8g..8g+7) and leaves them in LDS for the other wave, which walks them
// one 16-row tile at a time; the transposed images Q^T, dO^T, K^T (the a-operands above) and the per-row constants
// go through LDS as well: 52 KiB per workgroup, three workgroups per CU (with the bias-gradient
// sums below), whose memory latencies overlap.
// 88 MFMAs per (window, head).  Validity of padded tokens is folded into the per-row constants (LSE = +inf gives
// P = 0) and a 0/1 factor per column, so the element loops are branch-free.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int MT2 = 68;           // bf16 per LDS row of a [d][token] image in the backward (136 bytes)

__device__ __forceinline__ bf16x8 frag_tok2(const bf16_t* mt, int row, int kk, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(mt + row * MT2 + 32 * kk + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(mt + row * MT2 + 32 * kk + 16 + 4 * g);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// fragments f[t] (row 16 t + c, elements 8g..8g+7) -> image mt[d][token]
__device__ __forceinline__ void stage_tr(bf16_t* mt, int c, int g, const bf16x8 (&f)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) mt[(8 * g + e) * MT2 + 16 * t + c] = f[t][e];
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}

#ifdef VITK_WIN_PROF
// Instrumented build (tools/win_prof.py): wave 0 / wave 1 of every workgroup add the clocks they spend per phase
// into g_win_prof[wave][phase]: 0 prologue, 1 staging, 2 scores + products + stores, 4 table flush, 5 windows.
__device__ unsigned long long g_win_prof[2][8];
#define WIN_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#define WIN_ABL(bit) (p.abl & (bit))
#else
#define WIN_T(k) do { } while (0)
#define WIN_ABL(bit) false
#endif

__global__ __launch_bounds__(128, 2) void win_attn_bwd_mfma_kernel(WinArgs p) {
#ifdef VITK_WIN_PROF
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    __shared__ __attribute__((aligned(16))) bf16_t Qt[WD * MT2], Kt[WD * MT2], Gt[WD * MT2];
    __shared__ __attribute__((aligned(16))) bf16_t Qr[WN * MR], Kr[WN * MR], Vr[WN * MR], Gr[WN * MR];   // for the other wave
    __shared__ __attribute__((aligned(16))) float lses[WN], deltas[WN];
    __shared__ __attribute__((aligned(16))) int rids[WN], toks[WN], codes[WN];
    __shared__ __attribute__((aligned(16))) f32x4 dsacc[16 * 64];       // [it][jt][lane]: sums of dS over the windows
    __shared__ float tbl[WTBL], dtbl[WTBL];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.x % p.nH, g0 = blockIdx.x / p.nH;
    const int N = p.ws * p.ws, C = p.nH * WD, span = 2 * p.ws - 1;
    const int nwin = p.B * (p.H / p.ws) * (p.W / p.ws);
    // exp(x) = exp2(x * log2 e): the factor is folded into the table, the scale, the mask constant and the LSEs
    constexpr float LOG2E = 1.4426950408889634f;
    for (int e = threadIdx.x; e < span * span; e += 128) { tbl[e] = p.table[(long)e * p.nH + h] * LOG2E; dtbl[e] = 0.f; }
    const float scale2 = p.scale * LOG2E, mask2 = -100.0f * LOG2E;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
    const bf16_t* outp = reinterpret_cast<const bf16_t*>(p.out);
    const bf16_t* dout = reinterpret_cast<const bf16_t*>(p.dout);
    bf16_t* dqkv = reinterpret_cast<bf16_t*>(p.dqkv);
    const int off = (p.ws - 1) * (span + 1);
    // integer divisions by the window size happen here, once per workgroup: this lane's tokens 16 t + c
    // (fragment rows, b-operand rows) and 16 t + 4g + r (a-operand rows); the position code of token n is
    // (n / ws) * span + n % ws, the bias index of a (query, key) pair the difference of their codes + off
    int trow[4], tcol[4], codeG[4][4];
    const int rcp = (65536 + p.ws - 1) / p.ws;            // n / ws == (n * rcp) >> 16 for n < 64, ws <= 8
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 16 * t + c < N ? 16 * t + c : 0;
        trow[t] = (i * rcp) >> 16; tcol[t] = i - trow[t] * p.ws;
        if (wid == 0 && g == 0) codes[16 * t + c] = trow[t] * span + tcol[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r < N ? 16 * t + 4 * g + r : 0;
            const int jr = (j * rcp) >> 16;
            codeG[t][r] = jr * span + (j - jr * p.ws) + (wid == 0 ? off : -off);
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the pair (i, j) of a lane's dS element has the same relative-position index in every window: wave 1 sums the
    // bias-table gradient per lane over the workgroup's windows in LDS slots of its own (plain adds, no conflicts)
    // and scatters it into the table once, after the loop
    if (wid == 1)
#pragma unroll
        for (int a = 0; a < 16; ++a) dsacc[a * 64 + lane] = zero4;
    WIN_T(0);
    for (int win = g0; win < nwin; win += p.groups) {
        __syncthreads();                                   // the previous window's images and constants are free
        int wb, wh, ww;
        win_coords(p, win, wb, wh, ww);
        // wave 0 keeps the Q and dO fragments of all four row tiles, wave 1 those of K and V
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = 16 * t + c < N;
            const WinTok me = win_token_at(p, wb, wh, ww, trow[t], tcol[t]);
            const bf16_t* base = qkv + me.tok * (3L * C) + h * WD + 8 * g;
            const bf16_t* gsrc = dout + me.tok * C + h * WD + 8 * g;
            if (wid == 0) {
                fa[t] = valid ? *reinterpret_cast<const bf16x8*>(base) : bf16x8{};
                fb[t] = valid ? *reinterpret_cast<const bf16x8*>(gsrc) : bf16x8{};
                *reinterpret_cast<bf16x8*>(Qr + (16 * t + c) * MR + 8 * g) = fa[t];
                *reinterpret_cast<bf16x8*>(Gr + (16 * t + c) * MR + 8 * g) = fb[t];
                const bf16x8 of = valid ? *reinterpret_cast<const bf16x8*>(outp + me.tok * C + h * WD + 8 * g) : bf16x8{};
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)fb[t][e] * (float)of[e];
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
                if (g == 0) {
                    deltas[16 * t + c] = d;
                    lses[16 * t + c] = valid ? p.lse[((long)win * p.nH + h) * N + 16 * t + c] * LOG2E : INFINITY;   // P = 2^(s - inf) = 0
                    rids[16 * t + c] = valid ? me.rid : -1;
                    toks[16 * t + c] = (int)me.tok;
                }
            } else {
                fa[t] = valid ? *reinterpret_cast<const bf16x8*>(base + C) : bf16x8{};
                fb[t] = valid ? *reinterpret_cast<const bf16x8*>(base + 2 * C) : bf16x8{};
                *reinterpret_cast<bf16x8*>(Kr + (16 * t + c) * MR + 8 * g) = fa[t];
                *reinterpret_cast<bf16x8*>(Vr + (16 * t + c) * MR + 8 * g) = fb[t];
                const bf16x8 gf = valid ? *reinterpret_cast<const bf16x8*>(gsrc) : bf16x8{};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!WIN_ABL(8)) Gt[(8 * g + e) * MT2 + 16 * t + c] = gf[e];
            }
        }
        if (!WIN_ABL(8)) stage_tr(wid == 0 ? Qt : Kt, c, g, fa);
        __syncthreads();
        WIN_T(1);
        if (wid == 0) {
            // ---- queries down the accumulator rows: lane holds (i = 16 it + 4g + r, j = 16 jt + c)
            bf16x8 ga[2][2], qa[2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    ga[dt][kk] = frag_tok2(Gt, 16 * dt + c, kk, g);
                    qa[dt][kk] = frag_tok2(Qt, 16 * dt + c, kk, g);
                }
#pragma unroll 1
            for (int jt = 0; jt < 4; ++jt) {
                const int j = 16 * jt + c;
                const bool validj = j < N;
                const long tokj = toks[j];
                const int codej = codes[j], ridj = rids[j];
                const float vj = validj ? 1.0f : 0.0f;
                const bf16x8 kf = frag_row(Kr, j, g), vf = frag_row(Vr, j, g);
                // every LDS read of the iteration first (one wait), then the MFMAs, then the element math
                f32x4 lse[4], del[4], tb[4];
                i32x4 rd[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    lse[it] = *reinterpret_cast<const f32x4*>(lses + 16 * it + 4 * g);
                    del[it] = *reinterpret_cast<const f32x4*>(deltas + 16 * it + 4 * g);
                    rd[it] = *reinterpret_cast<const i32x4*>(rids + 16 * it + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb[it][r] = WIN_ABL(2) ? 0.1f : tbl[codeG[it][r] - codej];
                }
                f32x4 pr[4], ds[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[it], kf, zero4, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[it], vf, zero4, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // regions differ only under a shift (all 0 otherwise; -1 marks padding, which vj / LSE zero anyway)
                        const float v = s[r] * scale2 + tb[it][r] + (rd[it][r] != ridj ? mask2 : 0.0f);
                        pr[it][r] = (WIN_ABL(16) ? v - lse[it][r] : __builtin_amdgcn_exp2f(v - lse[it][r])) * vj;
                        ds[it][r] = pr[it][r] * (dp[r] - del[it][r]);
                    }
                }
                f32x4 dv[2] = {zero4, zero4}, dk[2] = {zero4, zero4};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 pb = pack8(pr[2 * kk], pr[2 * kk + 1]), sb = pack8(ds[2 * kk], ds[2 * kk + 1]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[dt][kk], pb, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[dt][kk], sb, dk[dt], 0, 0, 0);
                    }
                }
                if (validj && !WIN_ABL(4)) {
                    bf16_t* base = dqkv + tokj * (3L * C) + h * WD;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        *reinterpret_cast<bf16x4*>(base + C + 16 * dt + 4 * g) =
                            bf16x4{(bf16_t)(dk[dt][0] * p.scale), (bf16_t)(dk[dt][1] * p.scale), (bf16_t)(dk[dt][2] * p.scale),
                                   (bf16_t)(dk[dt][3] * p.scale)};
                        *reinterpret_cast<bf16x4*>(base + 2 * C + 16 * dt + 4 * g) =
                            bf16x4{(bf16_t)dv[dt][0], (bf16_t)dv[dt][1], (bf16_t)dv[dt][2], (bf16_t)dv[dt][3]};
                    }
                }
            }
        } else {
            // ---- keys down the accumulator rows: lane holds (j = 16 jt + 4g + r, i = 16 it + c)
            bf16x8 ka[2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) ka[dt][kk] = frag_tok2(Kt, 16 * dt + c, kk, g);
#pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                const int i = 16 * it + c;
                const bool validi = i < N;
                const long toki = toks[i];
                const int codei = codes[i], ridi = rids[i];
                const float lse = lses[i], delta = deltas[i];
                const bf16x8 qf = frag_row(Qr, i, g), gf = frag_row(Gr, i, g);
                f32x4 tb[4];
                i32x4 rd[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    rd[jt] = *reinterpret_cast<const i32x4*>(rids + 16 * jt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb[jt][r] = WIN_ABL(2) ? 0.1f : tbl[codei - codeG[jt][r]];
                }
                f32x4 ds[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[jt], qf, zero4, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jt], gf, zero4, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = s[r] * scale2 + tb[jt][r] + (rd[jt][r] != ridi ? mask2 : 0.0f);
                        const float vj = 16 * jt + 4 * g + r < N ? 1.0f : 0.0f;
                        ds[jt][r] = (WIN_ABL(16) ? v - lse : __builtin_amdgcn_exp2f(v - lse)) * vj * (dp[r] - delta);
                    }
                }
                if (!WIN_ABL(1))
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) dsacc[(it * 4 + jt) * 64 + lane] += ds[jt];          // padded pairs add 0
                f32x4 dq[2] = {zero4, zero4};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 sb = pack8(ds[2 * kk], ds[2 * kk + 1]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[dt][kk], sb, dq[dt], 0, 0, 0);
                }
                if (validi && !WIN_ABL(4)) {
                    bf16_t* base = dqkv + toki * (3L * C) + h * WD;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        *reinterpret_cast<bf16x4*>(base + 16 * dt + 4 * g) =
                            bf16x4{(bf16_t)(dq[dt][0] * p.scale), (bf16_t)(dq[dt][1] * p.scale), (bf16_t)(dq[dt][2] * p.scale),
                                   (bf16_t)(dq[dt][3] * p.scale)};
                }
            }
        }
        WIN_T(2);
#ifdef VITK_WIN_PROF
        tacc[5] += 1;
#endif
    }
    if (wid == 1) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * it + c < N && 16 * jt + 4 * g + r < N)
                        atomicAdd(dtbl + codes[16 * it + c] - codeG[jt][r], dsacc[(it * 4 + jt) * 64 + lane][r]);
    }
    __syncthreads();
    float* dt = p.dtable + (long)(g0 % p.replicas) * span * span * p.nH;
    for (int e = threadIdx.x; e < span * span; e += 128) atomicAdd(dt + (long)e * p.nH + h, dtbl[e]);
#ifdef VITK_WIN_PROF
    __builtin_amdgcn_s_waitcnt(0);
    WIN_T(4);
    if (lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(&g_win_prof[wid][k], tacc[k]);
#endif
}

int win_check(const WinArgs& a, int dh, int dtype) {
    if (dh != WD) return VITK_ERR_UNSUPPORTED;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0 || a.nH <= 0 || a.ws <= 0) return VITK_ERR_SHAPE;
    if (a.ws * a.ws > WN || a.H % a.ws || a.W % a.ws) return VITK_ERR_SHAPE;
    if (a.shift < 0 || a.shift >= a.ws) return VITK_ERR_SHAPE;
    if (dtype != VITK_BF16 && dtype != VITK_F32) return VITK_ERR_DTYPE;
    return 0;
}

int win_groups(const WinArgs& a, int target = 4096) {
    const long nwin = (long)a.B * (a.H / a.ws) * (a.W / a.ws);
    // enough workgroups to fill the chip several times over, few enough that the bias-gradient
    // table of a workgroup is flushed once per many windows
    const long want = target / a.nH > 0 ? target / a.nH : 1;
    return (int)(nwin < want ? nwin : want);
}

int g_win_force_valu = 0;        // test hook: bf16 instance on the VALU kernels

}  // namespace

extern "C" void vitk_debug_force_window_attn_valu(int on) { g_win_force_valu = on; }

#ifdef VITK_WIN_PROF
// out[2][8]: the per-wave phase clocks summed over every workgroup since the last call (and clears them)
extern "C" int vitk_debug_win_prof(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_prof), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
    unsigned long long zero[16] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_win_prof), zero, sizeof(zero));
}
#endif

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
    if (dtype == VITK_BF16 && !g_win_force_valu)
        hipLaunchKernelGGL(win_attn_fwd_mfma_kernel, grid, dim3(64), 0, (hipStream_t)stream, a);
    else if (dtype == VITK_BF16) hipLaunchKernelGGL(win_attn_fwd_kernel<bf16_t>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(win_attn_fwd_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_window_attn_bwd(const void* qkv, const float* bias_table, const void* out, const void* d_out,
                                    const float* lse, void* dqkv, float* dbias_table, int replicas, int B, int H,
                                    int W, int heads, int dh, int window, int shift, float scale, int dtype,
                                    void* stream) {
    if (!qkv || !bias_table || !out || !d_out || !lse || !dqkv || !dbias_table) return VITK_ERR_NULL;
    WinArgs a{};
    a.qkv = qkv; a.table = bias_table; a.out = const_cast<void*>(out); a.lse = const_cast<float*>(lse);
    a.dout = d_out; a.dqkv = dqkv; a.dtable = dbias_table; a.replicas = replicas;
    if (replicas < 1) return VITK_ERR_SHAPE;
    a.B = B; a.H = H; a.W = W; a.nH = heads; a.ws = window; a.shift = shift; a.scale = scale;
    const int rc = win_check(a, dh, dtype);
    if (rc) return rc;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dqkv) & 15) return VITK_ERR_SHAPE;
    // the MFMA kernel's prologue (bias table, index arithmetic) and its table flush cost about as much as a
    // window: one round of workgroups (three fit a CU) walks several windows each
    a.groups = win_groups(a, (dtype == VITK_BF16 && !g_win_force_valu) ? 3 * 256 : 4096);
#ifdef VITK_WIN_PROF
    a.abl = getenv("VITK_WIN_ABL") ? atoi(getenv("VITK_WIN_ABL")) : 0;
    if (getenv("VITK_WIN_GROUPS")) a.groups = win_groups(a, atoi(getenv("VITK_WIN_GROUPS")));
#endif
    const dim3 grid(a.groups * heads);
    if (dtype == VITK_BF16 && !g_win_force_valu)
        hipLaunchKernelGGL(win_attn_bwd_mfma_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (dtype == VITK_BF16)
        hipLaunchKernelGGL(win_attn_bwd_kernel<bf16_t>, grid, dim3(128), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(win_attn_bwd_kernel<float>, grid, dim3(128), 0, (hipStream_t)stream, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
