// Shared device helpers for the vitk kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/vitk.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define VITK_LDS __attribute__((address_space(3)))

#define VITK_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

static inline int vitk_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ----------------------------------------------------------------------------- scalar math
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 instance: gelu(x) = x * sigmoid(x * (A + B x^2 + C x^4)) with x^2 clamped at 64; the
// coefficients are a minimax fit against x * Phi(x) (|abs err| <= 2.6e-5, derivative 1.1e-4: far
// below bf16 rounding of the result).  One v_exp + one v_rcp per element; the polynomial runs
// two elements at a time on packed fp32.  The fp32 instance keeps erff().
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float GELU_A = 1.5950157685879343f, GELU_B = 0.0740112920281962f, GELU_C = -0.0007030335742745805f;
// s = sigmoid(u(x)) and t = u(x) / x (the polynomial) for two elements
__device__ __forceinline__ void gelu_sig2(f32x2 x, f32x2& s, f32x2& x2c) {
    const f32x2 x2 = x * x;
    x2c = f32x2{fminf(x2[0], 64.0f), fminf(x2[1], 64.0f)};
    constexpr float L = -1.4426950408889634f;             // exp(-u) = exp2(L * u)
    f32x2 t = (GELU_C * L) * x2c + (GELU_B * L);
    t = t * x2c + (GELU_A * L);
    const f32x2 nu = x * t;
    const f32x2 d = 1.0f + f32x2{__builtin_amdgcn_exp2f(nu[0]), __builtin_amdgcn_exp2f(nu[1])};
    s = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 s, x2c;
    gelu_sig2(x, s, x2c);
    return x * s;
}
__device__ __forceinline__ f32x2 gelu_fast_grad2(f32x2 x) {
    f32x2 s, x2c;
    gelu_sig2(x, s, x2c);
    f32x2 up = (5.0f * GELU_C) * x2c + (3.0f * GELU_B);   // u'(x)
    up = up * x2c + GELU_A;
    const f32x2 q = x * s * (1.0f - s);
    return q * up + s;
}
// v[i] = gelu(v[i]) (fwd) for NV values
template <typename T, int NV> __device__ __forceinline__ void gelu_fwd_vec(float* v) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
            const f32x2 y = gelu_fast2(f32x2{v[i], v[i + 1]});
            v[i] = y[0]; v[i + 1] = y[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = gelu_erf(v[i]);
    }
}
// v[i] = gelu(x) and d[i] = gelu'(x) - 1/2 for x = v[i], from one evaluation of the sigmoid (bf16 instance) or of
// erf (fp32): the forward epilogue of fc1 leaves the derivative behind for the backward pass instead of x itself,
// so the dX epilogue of fc2 is a load and a multiply.  It is kept as its distance from 1/2 (gelu'(0) = 1/2; the
// range is [-0.63, 0.63]): rounded to bf16, the error then vanishes with x like that of a rounded x did,
// instead of being 2^-9 of the derivative everywhere.
template <typename T, int NV> __device__ __forceinline__ void gelu_fwd_grad_vec(float* v, float* d) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
            const f32x2 x = f32x2{v[i], v[i + 1]};
            f32x2 s, x2c;
            gelu_sig2(x, s, x2c);
            const f32x2 y = x * s;
            f32x2 up = (5.0f * GELU_C) * x2c + (3.0f * GELU_B);   // u'(x)
            up = up * x2c + GELU_A;
            const f32x2 q = y * (1.0f - s);
            const f32x2 g = q * up + (s - 0.5f);
            v[i] = y[0]; v[i + 1] = y[1];
            d[i] = g[0]; d[i + 1] = g[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float x = v[i];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
            v[i] = x * cdf;
            d[i] = (cdf - 0.5f) + x * pdf;
        }
    }
}
// v[i] *= gelu'(u[i])
template <typename T, int NV, typename E> __device__ __forceinline__ void gelu_bwd_vec(float* v, const E& u) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
            const f32x2 d = gelu_fast_grad2(f32x2{u.get(i), u.get(i + 1)});
            v[i] *= d[0]; v[i + 1] *= d[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] *= gelu_erf_grad(u.get(i));
    }
}

template <typename T> __device__ __forceinline__ float gelu_fwd(float x) {
    if constexpr (sizeof(T) == 2) {
        return gelu_fast2(f32x2{x, x})[0];
    } else {
        return gelu_erf(x);
    }
}
template <typename T> __device__ __forceinline__ float gelu_bwd(float x) {
    if constexpr (sizeof(T) == 2) {
        return gelu_fast_grad2(f32x2{x, x})[0];
    } else {
        return gelu_erf_grad(x);
    }
}

// ----------------------------------------------------------------------------- dropout mask
// Counter-based mask: element e of a tensor keeps its value when the 16-bit hash lane drawn
// for it is >= thr16.  The same function is evaluated by the forward epilogue and by the
// backward pass, so no mask is ever stored.  Realised drop probability is thr16/65536.
//
// Elements are hashed in groups of 8 consecutive indices: one full 32-bit mix of (group, seed)
// gives the group's base word; the word for element pair w = (e >> 1) & 3 of the group is one
// more multiply-xorshift round of base ^ K[w], and its two 16-bit halves serve the pair.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_group_base(uint64_t seed, uint64_t group) {
    const uint32_t lo = (uint32_t)group, hi = (uint32_t)(group >> 32);
    return mix32((lo + (uint32_t)seed) ^ (hi * 0x9E3779B1U + (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ uint32_t drop_pair_key(int w) {
    return w == 0 ? 0u : (w == 1 ? 0x85ebca6bU : (w == 2 ? 0xc2b2ae35U : 0x27d4eb2fU));
}
__device__ __forceinline__ uint32_t drop_pair_word(uint32_t base, int w) {
    uint32_t h = (base ^ drop_pair_key(w)) * 0x9E3779B1U;
    return h ^ (h >> 16);
}
// keep flag of element e (any e).
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t e, uint32_t thr16) {
    const uint32_t h = drop_pair_word(drop_group_base(seed, e >> 3), (int)((e >> 1) & 3));
    const uint32_t bits = (e & 1) ? (h >> 16) : (h & 0xffffu);
    return bits >= thr16;
}
// keep flags of NV consecutive elements starting at e (NV in {2, 4, 8}, e a multiple of NV).
template <int NV>
__device__ __forceinline__ void drop_keep_vec(uint64_t seed, uint64_t e, uint32_t thr16, bool* keep) {
    static_assert(NV == 2 || NV == 4 || NV == 8, "one group of 8 covers the vector");
    const uint32_t base = drop_group_base(seed, e >> 3);
    const int w0 = NV == 8 ? 0 : (int)((e >> 1) & 3);
#pragma unroll
    for (int i = 0; i < NV; i += 2) {
        const uint32_t h = drop_pair_word(base, w0 + (i >> 1));
        keep[i] = (h & 0xffffu) >= thr16;
        keep[i + 1] = (h >> 16) >= thr16;
    }
}
static inline uint32_t vitk_drop_thr16(float p) {
    if (!(p > 0.f)) return 0;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}
static inline float vitk_drop_scale(uint32_t thr16) { return 65536.0f / (float)(65536u - thr16); }

// ----------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ----------------------------------------------------------------------------- typed vector IO
template <typename T> struct Vec4;   // 4 consecutive elements
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float* o) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static __device__ __forceinline__ void store(float* p, const float* o) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(p) = v;
    }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
        bf16x4 v = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        *reinterpret_cast<bf16x4*>(p) = v;
    }
};

template <> struct Vec4<f16_t> {
    static __device__ __forceinline__ void load(const f16_t* p, float* o) {
        const f16x4 v = *reinterpret_cast<const f16x4*>(p);
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float* o) {
        f16x4 v = {(f16_t)o[0], (f16_t)o[1], (f16_t)o[2], (f16_t)o[3]};
        *reinterpret_cast<f16x4*>(p) = v;
    }
};

template <typename T, int NV>
__device__ __forceinline__ void load_vec(const T* p, float* o) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) Vec4<T>::load(p + i, o + i);
}
template <int NV>
__device__ __forceinline__ void store_vec(bf16_t* p, const float* o) {
    if constexpr (NV == 8) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
        *reinterpret_cast<bf16x8*>(p) = v;
    } else {
#pragma unroll
        for (int i = 0; i < NV; i += 4) Vec4<bf16_t>::store(p + i, o + i);
    }
}
template <int NV>
__device__ __forceinline__ void store_vec(f16_t* p, const float* o) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) Vec4<f16_t>::store(p + i, o + i);
}
template <int NV>
__device__ __forceinline__ void store_vec(float* p, const float* o) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) Vec4<float>::store(p + i, o + i);
}

// ----------------------------------------------------------------------------- GEMM epilogue
// Shared by the bf16 and fp32 GEMM kernels.  One call handles NV consecutive output columns
// n..n+NV-1 of output row m; v[] holds the fp32 accumulators.
struct GemmEpi {
    const float* bias;      // [N] fp32 (BIAS*, PATCH)
    const void* residual;   // [M, ld_res] in the activation dtype (BIAS_RES)
    void* aux;              // BIAS_GELU: out, gelu'(pre-activation) - 1/2, or -1/2 where dropout zeroed C; DGELU: that
                            // tensor in.  Element type AuxOf<activation type>: IEEE half beside bf16, fp32 beside fp32
    const float* pos;       // PATCH: position embeddings [(npatch+1), N] fp32
    long ld_res, ld_aux;
    int npatch;             // PATCH: patches per image
    int N;                  // logical column count (dropout element index, pos row stride)
    uint32_t thr16;         // dropout threshold (0 = no dropout)
    float drop_scale;
    uint64_t seed;
};

// Raw storage of NV consecutive elements of T (kept unconverted so that many loads can be in
// flight at once without spending fp32 registers on them).
template <typename T, int NV> struct RawVec;
template <> struct RawVec<bf16_t, 8> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct RawVec<bf16_t, 4> {
    bf16x4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x4*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct RawVec<f16_t, 8> {
    f16x8 v;
    __device__ __forceinline__ void load(const f16_t* p) { v = *reinterpret_cast<const f16x8*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct RawVec<f16_t, 4> {
    f16x4 v;
    __device__ __forceinline__ void load(const f16_t* p) { v = *reinterpret_cast<const f16x4*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct RawVec<float, 8> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4*>(p);
        b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};
template <> struct RawVec<float, 4> {
    f32x4 a;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ float get(int i) const { return a[i]; }
};

// The epilogue is split in two so that a kernel can issue the loads of many output fragments
// back to back and only then consume them (a load-use-store chain per fragment serialises on
// memory latency).  TC: type of C and of the residual (the residual stream is fp32 even when
// activations are bf16); TA: activation type of `aux`.
// Element type of `aux` (the GELU derivative that BIAS_GELU writes and DGELU reads) next to activations of type
// TA: IEEE half beside bf16 activations.  Its values lie in [-0.63, 0.63], where half keeps 11 significant bits
// against the 8 of bf16 in the same two bytes; only these two epilogues ever touch the tensor.
template <typename TA> struct AuxOf { using type = TA; };
template <> struct AuxOf<bf16_t> { using type = f16_t; };

template <int EPI, typename TC, typename TA> struct EpiExt {
    static constexpr bool kHas = EPI == VITK_EPI_BIAS_RES || EPI == VITK_EPI_DGELU || EPI == VITK_EPI_PATCH;
    // element type of the extra operand: residual (TC), GELU derivative (AuxOf<TA>) or position embedding (fp32)
    using type = typename std::conditional<EPI == VITK_EPI_DGELU, typename AuxOf<TA>::type,
                 typename std::conditional<EPI == VITK_EPI_PATCH, float, TC>::type>::type;
};

// m, n must be in range (callers clamp them for the load and predicate the store).
template <int EPI, typename TC, typename TA, int NV>
__device__ __forceinline__ void epi_load_ext(const GemmEpi& e, int m, int n,
                                             RawVec<typename EpiExt<EPI, TC, TA>::type, NV>& ext) {
    if constexpr (EPI == VITK_EPI_BIAS_RES) {
        ext.load(reinterpret_cast<const TC*>(e.residual) + (long)m * e.ld_res + n);
    } else if constexpr (EPI == VITK_EPI_DGELU) {
        ext.load(reinterpret_cast<const typename AuxOf<TA>::type*>(e.aux) + (long)m * e.ld_aux + n);
    } else if constexpr (EPI == VITK_EPI_PATCH) {
        const int img = m / e.npatch, pp = m - img * e.npatch;
        ext.load(e.pos + (long)(pp + 1) * e.N + n);
    }
}

// Output row of GEMM row m: the PATCH epilogue skips one class-token row per image.
template <int EPI>
__device__ __forceinline__ long epi_out_row(const GemmEpi& e, int m) {
    if constexpr (EPI == VITK_EPI_PATCH) {
        const int img = m / e.npatch, pp = m - img * e.npatch;
        return (long)img * (e.npatch + 1) + pp + 1;
    } else {
        return m;
    }
}

// The arithmetic of the epilogue without any store.  v[]: fp32 accumulators of NV consecutive columns
// n.. of row m, replaced by the values of C; bias[]: the NV bias values; pre[] (BIAS_GELU only)
// receives what goes to `aux`: s = gelu'(pre-activation) - 1/2 for kept elements, -1/2 for dropped ones, so
// that d C / d pre-activation = (s + 1/2) / keep, which DGELU multiplies its product with.
template <int EPI, typename TC, typename TA, int NV>
__device__ __forceinline__ void epi_apply(const GemmEpi& e, float* v, const float* bias,
                                          const RawVec<typename EpiExt<EPI, TC, TA>::type, NV>& ext,
                                          int m, int n, float* pre) {
    if constexpr (EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_RES ||
                  EPI == VITK_EPI_PATCH) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += bias[i];
    }
    if constexpr (EPI == VITK_EPI_PATCH) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += ext.get(i);
    }
    if constexpr (EPI == VITK_EPI_BIAS_GELU) gelu_fwd_grad_vec<TA, NV>(v, pre);
    if constexpr (EPI == VITK_EPI_DGELU) {
        // the factor already carries the forward pass's dropout mask (-1/2 -> 0); drop_scale is 1 without dropout
        const float half = 0.5f * e.drop_scale;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] *= ext.get(i) * e.drop_scale + half;
    } else if (e.thr16) {
        bool keep[NV];
        drop_keep_vec<NV>(e.seed, (uint64_t)epi_out_row<EPI>(e, m) * (uint64_t)e.N + (uint64_t)n, e.thr16, keep);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = keep[i] ? v[i] * e.drop_scale : 0.f;
            if constexpr (EPI == VITK_EPI_BIAS_GELU) pre[i] = keep[i] ? pre[i] : -0.5f;
        }
    }
    if constexpr (EPI == VITK_EPI_BIAS_RES) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += ext.get(i);
    }
}

// v[]: fp32 accumulators of NV consecutive columns n.. of row m; bias[]: the NV bias values.
template <int EPI, typename TC, typename TA, int NV>
__device__ __forceinline__ void epi_apply_store(const GemmEpi& e, float* v, const float* bias,
                                                const RawVec<typename EpiExt<EPI, TC, TA>::type, NV>& ext,
                                                int m, int n, TC* C, long ldc) {
    float pre[NV];
    epi_apply<EPI, TC, TA, NV>(e, v, bias, ext, m, n, pre);
    if constexpr (EPI == VITK_EPI_BIAS_GELU)
        store_vec<NV>(reinterpret_cast<typename AuxOf<TA>::type*>(e.aux) + (long)m * e.ld_aux + n, pre);
    store_vec<NV>(C + epi_out_row<EPI>(e, m) * ldc + n, v);
}

// One-fragment convenience form (load, then apply).
template <int EPI, typename TC, typename TA, int NV>
__device__ __forceinline__ void gemm_epilogue(const GemmEpi& e, float* v, int m, int n, TC* C, long ldc) {
    float bias[NV];
    if constexpr (EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_RES ||
                  EPI == VITK_EPI_PATCH) {
        load_vec<float, NV>(e.bias + n, bias);
    }
    RawVec<typename EpiExt<EPI, TC, TA>::type, NV> ext;
    epi_load_ext<EPI, TC, TA, NV>(e, m, n, ext);
    epi_apply_store<EPI, TC, TA, NV>(e, v, bias, ext, m, n, C, ldc);
}
