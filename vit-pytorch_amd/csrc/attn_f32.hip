// fp32 attention (parity-check instance) and the attention-probability kernel for vis=True.
// One wavefront per (image, head, query): scores for all keys live in LDS.
#include "common.h"

namespace {

struct AttnF32Args {
    const float* q; const float* k; const float* v; const float* o; const float* d_o;
    float* out; float* dq; float* dk; float* dv;
    float* lse; float* delta;
    int B, H, N, dh;
    long ld_qkv, ld_o, ld_dqkv;
    float scale;
    uint32_t thr16;         // dropout on the probabilities: 0 = none; element ((b*H + h)*N + query)*N + key
    float dscale;
    uint64_t seed;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float dot_row(const float* a_lds, const float* b, int dh) {
    float s = 0.f;
    for (int d = 0; d < dh; d += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(b + d);
        s += a_lds[d] * kv[0] + a_lds[d + 1] * kv[1] + a_lds[d + 2] * kv[2] + a_lds[d + 3] * kv[3];
    }
    return s;
}

__global__ __launch_bounds__(64) void attn_fwd_f32_kernel(AttnF32Args p) {
    extern __shared__ float sm[];   // [128] q row, [N] probabilities
    float* qs = sm;
    float* ps = sm + 128;
    const int lane = threadIdx.x;
    const int qi = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const float* qrow = p.q + (tok0 + qi) * p.ld_qkv + h * p.dh;
    for (int d = lane; d < p.dh; d += 64) qs[d] = qrow[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < p.N; key += 64) {
        const float s = dot_row(qs, p.k + (tok0 + key) * p.ld_qkv + h * p.dh, p.dh) * p.scale;
        ps[key] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int key = lane; key < p.N; key += 64) {
        const float e = expf(ps[key] - mx);
        ps[key] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (p.thr16) {          // masked, rescaled probabilities enter P.V; the normaliser keeps the full row sum
        const uint64_t erow = ((uint64_t)bh * p.N + (uint64_t)qi) * (uint64_t)p.N;
        for (int key = lane; key < p.N; key += 64)
            ps[key] = drop_keep(p.seed, erow + (uint64_t)key, p.thr16) ? ps[key] * p.dscale : 0.f;
    }
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < p.dh; d += 64) {
        float acc = 0.f;
        for (int key = 0; key < p.N; ++key) acc += ps[key] * p.v[(tok0 + key) * p.ld_qkv + h * p.dh + d];
        p.out[(tok0 + qi) * p.ld_o + h * p.dh + d] = acc * inv;
    }
    if (lane == 0) p.lse[(long)bh * p.N + qi] = mx + logf(sum);
}

__global__ __launch_bounds__(64) void attn_bwd_f32_kernel(AttnF32Args p) {
    extern __shared__ float sm[];   // [128] q, [128] dO, [N] p, [N] ds
    float* qs = sm;
    float* dos = sm + 128;
    float* ps = sm + 256;
    float* dss = ps + p.N;
    const int lane = threadIdx.x;
    const int qi = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const long hq = h * p.dh;
    float dl = 0.f;
    for (int d = lane; d < p.dh; d += 64) {
        qs[d] = p.q[(tok0 + qi) * p.ld_qkv + hq + d];
        const float go = p.d_o[(tok0 + qi) * p.ld_o + hq + d];
        dos[d] = go;
        dl += go * p.o[(tok0 + qi) * p.ld_o + hq + d];
    }
    dl = wave_sum(dl);
    if (lane == 0) p.delta[(long)bh * p.N + qi] = dl;
    __syncthreads();
    const float l = p.lse[(long)bh * p.N + qi];
    for (int key = lane; key < p.N; key += 64) {
        const float s = dot_row(qs, p.k + (tok0 + key) * p.ld_qkv + hq, p.dh) * p.scale;
        const float pr = expf(s - l);
        float dp = dot_row(dos, p.v + (tok0 + key) * p.ld_qkv + hq, p.dh);
        float m = 1.f;
        if (p.thr16)
            m = drop_keep(p.seed, ((uint64_t)bh * p.N + (uint64_t)qi) * (uint64_t)p.N + (uint64_t)key, p.thr16) ? p.dscale : 0.f;
        ps[key] = pr * m;                       // enters dV
        dss[key] = pr * (dp * m - dl) * p.scale;
    }
    __syncthreads();
    for (int d = lane; d < p.dh; d += 64) {
        float acc = 0.f;
        const float qd = qs[d], god = dos[d];
        for (int key = 0; key < p.N; ++key) {
            const float ds = dss[key];
            acc += ds * p.k[(tok0 + key) * p.ld_qkv + hq + d];
            atomicAdd(p.dk + (tok0 + key) * p.ld_dqkv + hq + d, ds * qd);
            atomicAdd(p.dv + (tok0 + key) * p.ld_dqkv + hq + d, ps[key] * god);
        }
        p.dq[(tok0 + qi) * p.ld_dqkv + hq + d] = acc;
    }
}

// probs[b,h,q,key] = exp(scale * q.k - lse[q])
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(const T* q, const T* k, const float* lse, float* probs,
                                                        int H, int N, int dh, long ld, float scale) {
    __shared__ float qs[128];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const long tok0 = (long)b * N;
    for (int d = lane; d < dh; d += 64) qs[d] = (float)q[(tok0 + qi) * ld + h * dh + d];
    __syncthreads();
    const float l = lse[(long)bh * N + qi];
    for (int key = lane; key < N; key += 64) {
        const T* kr = k + (tok0 + key) * ld + h * dh;
        float s = 0.f;
        for (int d = 0; d < dh; d += 4) {
            float kv[4];
            Vec4<T>::load(kr + d, kv);
            s += qs[d] * kv[0] + qs[d + 1] * kv[1] + qs[d + 2] * kv[2] + qs[d + 3] * kv[3];
        }
        probs[((long)bh * N + qi) * N + key] = expf(s * scale - l);
    }
}

}  // namespace

int vitk_attn_fwd_f32(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                      int N, int dh, long ld_qkv, long ld_o, float scale, float dropout_p, uint64_t seed, hipStream_t st) {
    if (dh > 128 || dh % 4 || ld_qkv % 4) return VITK_ERR_SHAPE;
    AttnF32Args a{};
    a.q = (const float*)q; a.k = (const float*)k; a.v = (const float*)v; a.out = (float*)o; a.lse = lse;
    a.B = B; a.H = H; a.N = N; a.dh = dh; a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.scale = scale;
    a.thr16 = vitk_drop_thr16(dropout_p); a.dscale = vitk_drop_scale(a.thr16); a.seed = seed;
    const size_t lds = (128 + (size_t)N) * sizeof(float);
    if (lds > 64 * 1024) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(N, B * H), dim3(64), lds, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

int vitk_attn_bwd_f32(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int N,
                      int dh, long ld_qkv, long ld_o, long ld_dqkv, float scale, float dropout_p, uint64_t seed,
                      hipStream_t st) {
    if (dh > 128 || dh % 4 || ld_qkv % 4) return VITK_ERR_SHAPE;
    AttnF32Args a{};
    a.q = (const float*)q; a.k = (const float*)k; a.v = (const float*)v;
    a.o = (const float*)o; a.d_o = (const float*)d_o;
    a.dq = (float*)dq; a.dk = (float*)dk; a.dv = (float*)dv;
    a.lse = const_cast<float*>(lse); a.delta = delta;
    a.B = B; a.H = H; a.N = N; a.dh = dh;
    a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.ld_dqkv = ld_dqkv; a.scale = scale;
    a.thr16 = vitk_drop_thr16(dropout_p); a.dscale = vitk_drop_scale(a.thr16); a.seed = seed;
    const size_t lds = (256 + 2 * (size_t)N) * sizeof(float);
    if (lds > 64 * 1024) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(N, B * H), dim3(64), lds, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

int vitk_attn_probs_any(const void* q, const void* k, const float* lse, float* probs, int B, int H, int N,
                        int dh, long ld_qkv, float scale, int dtype, hipStream_t st) {
    if (dh > 128 || dh % 4 || ld_qkv % 4) return VITK_ERR_SHAPE;
    const dim3 grid(N, B * H);
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(attn_probs_kernel<bf16_t>, grid, dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           lse, probs, H, N, dh, ld_qkv, scale);
    else
        hipLaunchKernelGGL(attn_probs_kernel<float>, grid, dim3(64), 0, st, (const float*)q, (const float*)k,
                           lse, probs, H, N, dh, ld_qkv, scale);
    VITK_CHECK_LAUNCH();
    return 0;
}
