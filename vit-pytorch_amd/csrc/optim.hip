// Optimizer step of the fine-tuning loop (reference train.py:535-538: clip_grad_norm_(1.0), SGD with
// momentum 0.9, zero_grad) as two launches over a table of tensors instead of a dozen multi-tensor
// launches of elementwise arithmetic.
//
//   pass 1  sum of squares of every gradient: up to 1024 blocks each add their chunks in a fixed order into a
//           partial of their own, one block then adds the partials in a fixed order (reads g).  No atomics:
//           the norm - and with it the clip coefficient and every updated parameter - is bit-identical
//           wherever the gradients are, which is what keeps data-parallel replicas identical
//   pass 2  coef = min(1, max_norm / (norm + 1e-6)); g' = coef g + wd p;
//           m = first ? g' : mu m + (1 - dampening) g';  p -= lr m     (reads p, g, m; writes p, m)
//
// HBM-bound: 4 + 20 bytes per parameter per step.  A table entry covers one tensor; entry e owns
// blocks [block0, next block0), each block 8192 consecutive elements.
#include "common.h"

namespace {

constexpr int SGD_CHUNK = 8192;          // elements per block
constexpr int SGD_SLOTS = 64;            // what pass 2 reads: the total in slot 0, zeros behind it
constexpr int SGD_PARTIALS = 1024;       // at most this many blocks in pass 1; their partials follow the 64 slots
static_assert(SGD_SLOTS + SGD_PARTIALS == VITK_SGD_SLOTS, "slots buffer layout");

struct SgdEntry {                        // mirrors vitk_sgd_entry
    float* p; float* g; float* m;
    long n;
    int block0;
    int flags;                           // bit 0: momentum buffer not initialised yet; bit 1: 16-byte vector access allowed
};
static_assert(sizeof(SgdEntry) == sizeof(vitk_sgd_entry), "table layout");

__device__ __forceinline__ int sgd_find(const SgdEntry* tab, int n, int block) {
    int lo = 0, hi = n - 1;              // last entry with block0 <= block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block0 <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sgd_norm_kernel(const SgdEntry* tab, int n, int n_blocks, float* partial) {
    __shared__ float part[4];
    float s = 0.f;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {     // chunks of this block, always in this order
        const int e = sgd_find(tab, n, blk);
        const SgdEntry en = tab[e];
        const long base = (long)(blk - en.block0) * SGD_CHUNK;
        const long left = en.n - base;
        const int cnt = left < SGD_CHUNK ? (int)left : SGD_CHUNK;
        const float* g = en.g + base;
        if (en.flags & 2) {
            for (int i = threadIdx.x * 4; i < cnt; i += 1024) {      // n % 4 == 0 for vector entries
                const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
                s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            }
        } else {
            for (int i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// slots[0] = sum of the nb partials (fixed order), slots[1..63] = 0.
__global__ __launch_bounds__(256) void sgd_norm_final_kernel(const float* partial, int nb, float* slots) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < SGD_SLOTS) slots[threadIdx.x] = threadIdx.x == 0 ? (part[0] + part[1]) + (part[2] + part[3]) : 0.f;
}

__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdEntry* tab, int n, const float* slots, float max_norm,
                                                       float lr, float mu, float damp, float wd) {
    float coef = 1.0f;
    if (slots) {
        float s = slots[threadIdx.x & (SGD_SLOTS - 1)];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        coef = fminf(max_norm / (sqrtf(s) + 1e-6f), 1.0f);
    }
    const int e = sgd_find(tab, n, blockIdx.x);
    const SgdEntry en = tab[e];
    const long base = (long)(blockIdx.x - en.block0) * SGD_CHUNK;
    const long left = en.n - base;
    const int cnt = left < SGD_CHUNK ? (int)left : SGD_CHUNK;
    float* p = en.p + base;
    const float* g = en.g + base;
    float* m = en.m ? en.m + base : nullptr;      // no buffer: plain SGD (momentum 0)
    const bool first = (en.flags & 1) || !m;
    const float gs = 1.0f - damp;
    if (en.flags & 2) {
        for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
            f32x4 gv = *reinterpret_cast<const f32x4*>(g + i) * coef;
            if (wd != 0.f) gv += pv * wd;
            f32x4 mv;
            if (first) mv = gv;
            else mv = *reinterpret_cast<const f32x4*>(m + i) * mu + gv * gs;
            pv -= mv * lr;
            if (m) *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(p + i) = pv;
        }
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) {
            float pv = p[i];
            float gv = g[i] * coef;
            if (wd != 0.f) gv += pv * wd;
            const float mv = first ? gv : m[i] * mu + gv * gs;
            if (m) m[i] = mv;
            p[i] = pv - mv * lr;
        }
    }
}

}  // namespace

extern "C" int vitk_sgd_norm_sq(const void* table, int n_entries, int n_blocks, float* slots, void* stream) {
    if (!table || !slots) return VITK_ERR_NULL;
    if (n_entries <= 0 || n_blocks <= 0) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = n_blocks < SGD_PARTIALS ? n_blocks : SGD_PARTIALS;
    hipLaunchKernelGGL(sgd_norm_kernel, dim3(nb), dim3(256), 0, st, (const SgdEntry*)table, n_entries, n_blocks,
                       slots + SGD_SLOTS);
    VITK_CHECK_LAUNCH();
    hipLaunchKernelGGL(sgd_norm_final_kernel, dim3(1), dim3(256), 0, st, slots + SGD_SLOTS, nb, slots);
    VITK_CHECK_LAUNCH();
    return 0;
}

extern "C" int vitk_sgd_step(const void* table, int n_entries, int n_blocks, const float* slots, float max_norm,
                             float lr, float momentum, float dampening, float weight_decay, void* stream) {
    if (!table) return VITK_ERR_NULL;
    if (n_entries <= 0 || n_blocks <= 0) return VITK_ERR_SHAPE;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, (const SgdEntry*)table,
                       n_entries, slots, max_norm, lr, momentum, dampening, weight_decay);
    VITK_CHECK_LAUNCH();
    return 0;
}
