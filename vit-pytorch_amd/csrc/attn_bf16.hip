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
    // dropout on the probabilities (streaming kernels only): keep flag of element ((b*H + h)*N + query)*N + key
    uint32_t thr16;         // 0 = none
    float dscale;           // 1 / keep probability
    uint64_t seed;
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
__device__ __forceinline__ void glds16_at(const void* gsrc, char* lds_wave_base) {     // LDS-DMA: lane l -> base + 16 l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (VITK_LDS void*)lds_wave_base, 16, 0, 0);
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
struct FwdState {
    float m, lsum;
    f32x4 o[4];
};
// One 32-key tile (Ks/Vs: LDS images, k0: index of its first key) against the wave's 16 queries.
struct Scores {
    f32x4 s0, s1;
};
// S^T for one 32-key tile: independent of the running softmax state, so the caller issues it one
// tile ahead of the state update (MFMA work between the dependent max/exp/sum chain).
__device__ __forceinline__ Scores fwd_scores(const char* Ks, const bf16x8* qf, int lane) {
    Scores r;
    r.s0 = f32x4{0.f, 0.f, 0.f, 0.f};
    r.s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        r.s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 0, s, lane), qf[s], r.s0, 0, 0, 0);
        r.s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 16, s, lane), qf[s], r.s1, 0, 0, 0);
    }
    return r;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// DROP: the probabilities that enter P.V are masked and rescaled (the normaliser keeps the full row sum);
// erow = ((b*H + h)*N + query)*N, the element index of this lane's query row.
// MASK: the tile may hold keys past the sequence (only ever the last tile of a row of tiles).
// The running maximum st.m is kept in RAW score units (before the softmax scale): the scale and the subtraction of
// the maximum are one fused multiply-add per element in front of the exponential, exp2(s * sc - m * sc), and a tile
// without padded keys spends no instruction on key indices.  The kernels that call this are bound by
// vector-instruction issue, not by the matrix pipeline.
template <bool DROP = false, bool MASK = true>
__device__ __forceinline__ void fwd_update(FwdState& st, const Scores& sc_in, const char* Vs, int k0,
                                           int N, float sc, int lane, uint32_t thr16 = 0, float dscale = 1.f,
                                           uint64_t seed = 0, uint64_t erow = 0) {
    const int g = lane >> 4;
    // s0[r]: key k0 + 4g + r ; s1[r]: key k0 + 16 + 4g + r ; query = lane & 15
    float sv[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sv[r] = sc_in.s0[r];
        sv[4 + r] = sc_in.s1[r];
    }
    if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = k0 + 4 * g + r;
            sv[r] = (kk < N) ? sv[r] : -INFINITY;
            sv[4 + r] = (kk + 16 < N) ? sv[4 + r] : -INFINITY;
        }
    }
    // (v_max3 from assembly: fmaxf() makes the compiler quiet every input first, one more instruction per value;
    // the values are MFMA results or -inf, never signalling NaNs)
    float tmax = max3_raw(max3_raw(sv[0], sv[1], sv[2]), max3_raw(sv[3], sv[4], sv[5]), max3_raw(sv[6], sv[7], sv[7]));
    tmax = max3_raw(tmax, __shfl_xor(tmax, 16, 64), tmax);
    tmax = max3_raw(tmax, __shfl_xor(tmax, 32, 64), tmax);
    const float mn = max3_raw(st.m, tmax, tmax);  // finite: the first tile of a row holds >= 1 valid key
    const float alpha = __builtin_amdgcn_exp2f((st.m - mn) * sc);
    const float nb = -mn * sc;
#pragma unroll
    for (int r = 0; r < 8; ++r) sv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], sc, nb));
    const f32x2 ps2 = (f32x2{sv[0], sv[1]} + f32x2{sv[2], sv[3]}) + (f32x2{sv[4], sv[5]} + f32x2{sv[6], sv[7]});
    st.lsum = st.lsum * alpha + (ps2[0] + ps2[1]);
    st.m = mn;
    if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t e = erow + (uint64_t)(k0 + 4 * g + r);
            sv[r] = drop_keep(seed, e, thr16) ? sv[r] * dscale : 0.f;
            sv[4 + r] = drop_keep(seed, e + 16, thr16) ? sv[4 + r] * dscale : 0.f;
        }
    }
    const bf16x8 pb = pack8(sv, sv + 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.o[dt] *= alpha;
        st.o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Vs, 16 * dt, lane), pb, st.o[dt], 0, 0, 0);
    }
}
__device__ __forceinline__ void fwd_init(FwdState& st) {
    st.m = -INFINITY; st.lsum = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void fwd_finish(FwdState& st, const AttnArgs& p, long tok0, int bh, int h, int qi, int lane, float sc) {
    const int g = lane >> 4;
    float lsum = st.lsum;
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < p.N) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = p.out + (tok0 + qi) * p.ld_o + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt + 4 * g, st.o[dt], inv);
        if (p.o32) {
            float* o32row = p.o32 + (tok0 + qi) * p.ld_o + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4*>(o32row + 16 * dt + 4 * g) = st.o[dt] * inv;
        }
        if (g == 0) p.lse[(long)bh * p.N + qi] = (st.m * sc + __builtin_amdgcn_logf(lsum)) * LN2;   // sc = scale * log2(e)
    }
}

// Streaming variant (any N): 64 queries per workgroup, K/V tiles pass through LDS one at a time.
// (The second launch bound makes the compiler keep the MFMA accumulators in ordinary registers.  Without it it parks
// them in accumulation registers and moves every one of them out and back around the online-softmax rescale: 88 of
// the ~250 vector instructions of a key tile, in a kernel that is bound by vector-instruction issue (78 % of the
// VALU cycles busy): 292 -> 265 us per layer at 577 tokens.  The two streaming backward kernels only ever add to
// their accumulators with MFMAs; the same bound made them 8 % slower and they stay as they were.)
template <bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(AttnArgs p) {
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
    FwdState st;
    fwd_init(st);
    const int ntile = (p.N + 31) >> 5;
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        load_tile32(Ks, kb, p.ld_qkv, kt * 32, p.N, tid);
        load_tile32(Vs, vb, p.ld_qkv, kt * 32, p.N, tid);
        __syncthreads();
        const Scores sco = fwd_scores(Ks, qf, lane);
        fwd_update<DROP>(st, sco, Vs, kt * 32, p.N, sc, lane, p.thr16, p.dscale, p.seed,
                         ((uint64_t)bh * p.N + (uint64_t)qr) * (uint64_t)p.N);
    }
    fwd_finish(st, p, tok0, bh, h, qi, lane, sc);
}

// All rows [0, 32*ntile) of a token-major [N, 64] head slice -> LDS tiles (rows past N repeat row N-1).
template <int NT>
__device__ __forceinline__ void load_resident(char* lds, const bf16_t* base, long ld, int ntile, int nrows, int tid) {
    for (int c = tid; c < ntile * 256; c += NT) {
        const int row = c >> 3, ch = c & 7;
        const int r = row < nrows ? row : nrows - 1;
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + (long)r * ld + ch * 8);
        *reinterpret_cast<u32x4*>(lds + (row >> 5) * AT_TILE + (row & 31) * AT_ROW + ch * 16) = v;
    }
}

// Resident variant (short sequences): one workgroup per (image, head); K and V are loaded into
// LDS once, then every wave walks its 16-query tiles with no further barrier.
template <int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void attn_fwd_res_bf16_kernel(AttnArgs p, int ntile) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kall = smem;
    char* Vall = smem + ntile * AT_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const bf16_t* qb = p.q + tok0 * p.ld_qkv + h * DH;
    if (!(ABL & 1)) {
        load_resident<NW * 64>(Kall, p.k + tok0 * p.ld_qkv + h * DH, p.ld_qkv, ntile, p.N, tid);
        load_resident<NW * 64>(Vall, p.v + tok0 * p.ld_qkv + h * DH, p.ld_qkv, ntile, p.N, tid);
    }
    __syncthreads();
    const float sc = p.scale * LOG2E;
    const int nq = (p.N + 15) >> 4;
    for (int qt = wid; qt < nq; qt += NW) {
        const int qi = qt * 16 + l15;
        const int qr = qi < p.N ? qi : p.N - 1;
        bf16x8 qf[2];
        qf[0] = load_frag_global(qb + (long)qr * p.ld_qkv, 0, g);
        qf[1] = load_frag_global(qb + (long)qr * p.ld_qkv, 1, g);
        FwdState st;
        fwd_init(st);
        Scores cur = fwd_scores(Kall, qf, lane);
        for (int kt = 0; kt < ((ABL & 2) ? 1 : ntile); ++kt) {
            Scores nxt = cur;
            if (kt + 1 < ntile) nxt = fwd_scores(Kall + (kt + 1) * AT_TILE, qf, lane);
            fwd_update(st, cur, Vall + kt * AT_TILE, kt * 32, p.N, sc, lane);
            cur = nxt;
        }
        if (!(ABL & 4) || st.lsum == 12345.f) fwd_finish(st, p, tok0, bh, h, qi, lane, sc);
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
// Wave owns 16 keys (on the lane index) and walks the queries in tiles of 32.
struct DkvState {
    f32x4 dk[4], dv[4];
};
struct SdP {
    f32x4 sa[2], da[2];
};
// S = Q.K^T and dP = dO.V^T for one 32-query tile against the wave's 16 keys.
__device__ __forceinline__ SdP dkdv_products(const char* Qs, const char* Os, const bf16x8* kf, const bf16x8* vf, int lane) {
    SdP r;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        r.sa[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
        r.da[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            r.sa[qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qs, 16 * qq, s, lane), kf[s], r.sa[qq], 0, 0, 0);
            r.da[qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Os, 16 * qq, s, lane), vf[s], r.da[qq], 0, 0, 0);
        }
    }
    return r;
}
// lse2[q] = LSE * log2(e) and delta[q] for q in [q0, q0 + 32) (any address space; 0 / 0 past N is
// the caller's job for LDS copies, and `N` masks the rest).
// DROP: P enters dV masked and rescaled, and so does dP before delta is subtracted; ebh = (b*H + h)*N, ki = the
// lane's key.
template <typename LP, bool DROP = false>
__device__ __forceinline__ void dkdv_update(DkvState& st, const SdP& pr_in, const char* Qs, const char* Os,
                                            LP lse2, LP delta, int q0, int N, float scale, float sc, int lane,
                                            uint32_t thr16 = 0, float dscale = 1.f, uint64_t seed = 0, uint64_t ebh = 0,
                                            int ki = 0) {
    const int g = lane >> 4;
    float pv[8], dsv[8];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        // sa[r]: query q0 + 16qq + 4g + r, key = lane & 15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + 16 * qq + 4 * g + r;
            const bool ok = qi < N;
            const int qc = ok ? qi : q0;
            const float l2 = lse2[qc];
            const float dl = delta[qc];
            const float pr = ok ? __builtin_amdgcn_exp2f(pr_in.sa[qq][r] * sc - l2) : 0.f;
            if constexpr (DROP) {
                const float m = drop_keep(seed, (ebh + (uint64_t)qc) * (uint64_t)N + (uint64_t)ki, thr16) ? dscale : 0.f;
                pv[4 * qq + r] = pr * m;
                dsv[4 * qq + r] = pr * (pr_in.da[qq][r] * m - dl) * scale;
            } else {
                pv[4 * qq + r] = pr;
                dsv[4 * qq + r] = pr * (pr_in.da[qq][r] - dl) * scale;
            }
        }
    }
    const bf16x8 pb = pack8(pv, pv + 4);
    const bf16x8 dsb = pack8(dsv, dsv + 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Os, 16 * dt, lane), pb, st.dv[dt], 0, 0, 0);
        st.dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Qs, 16 * dt, lane), dsb, st.dk[dt], 0, 0, 0);
    }
}
__device__ __forceinline__ void dkdv_init(DkvState& st) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void dkdv_store(const DkvState& st, const AttnArgs& p, long tok0, int h, int ki, int lane) {
    const int g = lane >> 4;
    if (ki < p.N) {
        bf16_t* dkr = p.dk + (tok0 + ki) * p.ld_dqkv + h * DH;
        bf16_t* dvr = p.dv + (tok0 + ki) * p.ld_dqkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            store4(dkr + 16 * dt + 4 * g, st.dk[dt], 1.0f);
            store4(dvr + 16 * dt + 4 * g, st.dv[dt], 1.0f);
        }
    }
}

template <bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_bf16_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Qs[AT_TILE];
    __shared__ __attribute__((aligned(16))) char Os[AT_TILE];   // dO tile
    __shared__ float Ls[32], Ds[32];                            // LSE*log2e and delta of the tile's queries
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
    DkvState st;
    dkdv_init(st);
    const int ntile = (p.N + 31) >> 5;
    for (int qt = 0; qt < ntile; ++qt) {
        __syncthreads();
        load_tile32(Qs, qb, p.ld_qkv, qt * 32, p.N, tid);
        load_tile32(Os, dob, p.ld_o, qt * 32, p.N, tid);
        if (tid < 32) {
            const int qi = qt * 32 + tid;
            Ls[tid] = qi < p.N ? lse[qi] * LOG2E : 0.f;
            Ds[tid] = qi < p.N ? delta[qi] : 0.f;
        }
        __syncthreads();
        const SdP pr = dkdv_products(Qs, Os, kf, vf, lane);
        dkdv_update<const float*, DROP>(st, pr, Qs, Os, Ls - qt * 32, Ds - qt * 32, qt * 32, p.N, p.scale, sc, lane,
                                        p.thr16, p.dscale, p.seed, (uint64_t)bh * p.N, kr);
    }
    dkdv_store(st, p, tok0, h, ki, lane);
}

// ------------------------------------------------------------------------------ dQ
// Wave owns 16 queries (on the lane index) and walks the keys in tiles of 32.
// S^T = K.Q^T and dP^T = V.dO^T for one 32-key tile against the wave's 16 queries.
__device__ __forceinline__ SdP dq_products(const char* Ks, const char* Vs, const bf16x8* qf, const bf16x8* dof, int lane) {
    SdP r;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        r.sa[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        r.da[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            r.sa[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, 16 * kk, s, lane), qf[s], r.sa[kk], 0, 0, 0);
            r.da[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, 16 * kk, s, lane), dof[s], r.da[kk], 0, 0, 0);
        }
    }
    return r;
}
// DROP: dP is masked and rescaled before delta is subtracted; erow = ((b*H + h)*N + query)*N.
template <bool DROP = false>
__device__ __forceinline__ void dq_update(f32x4* dq, const SdP& pr_in, const char* Ks, float l2, float dl, int k0,
                                          int N, float scale, float sc, int lane, uint32_t thr16 = 0, float dscale = 1.f,
                                          uint64_t seed = 0, uint64_t erow = 0) {
    const int g = lane >> 4;
    float dsv[8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kidx = k0 + 16 * kk + 4 * g + r;
            const float pr = (kidx < N) ? __builtin_amdgcn_exp2f(pr_in.sa[kk][r] * sc - l2) : 0.f;
            float dp = pr_in.da[kk][r];
            if constexpr (DROP) dp = drop_keep(seed, erow + (uint64_t)(kidx < N ? kidx : 0), thr16) ? dp * dscale : 0.f;
            dsv[4 * kk + r] = pr * (dp - dl) * scale;
        }
    }
    const bf16x8 dsb = pack8(dsv, dsv + 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag32(Ks, 16 * dt, lane), dsb, dq[dt], 0, 0, 0);
}

template <bool DROP = false>
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
        const SdP pr = dq_products(Ks, Vs, qf, dof, lane);
        dq_update<DROP>(dq, pr, Ks, l2, dl, kt * 32, p.N, p.scale, sc, lane, p.thr16, p.dscale, p.seed,
                        ((uint64_t)bh * p.N + (uint64_t)qr) * (uint64_t)p.N);
    }
    if (qi < p.N) {
        bf16_t* dqr = p.dq + (tok0 + qi) * p.ld_dqkv + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(dqr + 16 * dt + 4 * g, dq[dt], 1.0f);
    }
}

// ------------------------------------------------------------------------------ fused backward
// One workgroup per (image, head), one wave per 32-key tile (up to NW tiles).  Wave w keeps K_w, V_w
// (MFMA row fragments in registers, K_w also as an LDS tile for transposed reads) and its dK, dV
// accumulators; the queries stream through one LDS tile pair (Q_i, dO_i) that all waves share, the
// next pair being fetched into registers while the current one is multiplied.  Per step a wave
// computes S = Q_i K_w^T and dP = dO_i V_w^T once, P and dS from them, then
//     dV_w += P^T dO_i,  dK_w += dS^T Q_i           (P, dS are B operands as they sit in registers)
//     dQ_i partial = dS K_w                          (dS re-read transposed from a wave-private LDS tile)
// and parks its dQ_i partial in its own fp32 LDS slot; after a barrier 256 threads sum the slots
// and write dQ_i while the others already start on the next tile pair (double-buffered; committed
// before that barrier).  (LDS float atomics were measured several times slower than slots.)
// delta_i = rowsum(dO_i o O_i) is formed while the tile loads.
constexpr int DQ_LD = 68;                 // floats per query row of a dQ slot (64 + pad)
constexpr int DQ_SLOT = 32 * DQ_LD;       // floats per slot

template <int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_fused_bf16_kernel(AttnArgs p, int ntile) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64;
    constexpr int NCH = (512 + NT - 1) / NT;                      // 16-byte chunks of a tile pair per thread
    char* Qb = smem;                                              // [2][AT_TILE]
    char* Ob = Qb + 2 * AT_TILE;                                  // [2][AT_TILE]  dO tiles
    char* Kt = Ob + 2 * AT_TILE;                                  // [NW][AT_TILE]
    char* St = Kt + NW * AT_TILE;                                 // [NW][AT_TILE] dS as [key][query]
    float* dQs = reinterpret_cast<float*>(St + NW * AT_TILE);     // [NW][DQ_SLOT] per-wave dQ partials [query][d]
    float* Ls = dQs + NW * DQ_SLOT;                               // [2][32] LSE * log2e
    float* Ds = Ls + 64;                                          // [2][32] delta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const long tok0 = (long)b * p.N;
    const bf16_t* qb = p.q + tok0 * p.ld_qkv + h * DH;
    const bf16_t* kb = p.k + tok0 * p.ld_qkv + h * DH;
    const bf16_t* vb = p.v + tok0 * p.ld_qkv + h * DH;
    const bf16_t* dob = p.d_o + tok0 * p.ld_o + h * DH;
    const float* lse = p.lse + (long)bh * p.N;
    const bool active = wid < ntile;
    char* KtW = Kt + wid * AT_TILE;
    char* StW = St + wid * AT_TILE;
    float* dQW = dQs + wid * DQ_SLOT;

    // ---- this wave's keys
    bf16x8 kf[2][2], vf[2][2];
    float kmask[2];
    if (active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = lane + 64 * t, row = c >> 3, ch = c & 7;
            int kr = 32 * wid + row; kr = kr < p.N ? kr : p.N - 1;
            *reinterpret_cast<u32x4*>(KtW + row * AT_ROW + ch * 16) =
                *reinterpret_cast<const u32x4*>(kb + (long)kr * p.ld_qkv + ch * 8);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int key = 32 * wid + 16 * kk + l15;
            const int kr = key < p.N ? key : p.N - 1;
            kmask[kk] = key < p.N ? 1.0f : 0.0f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kf[kk][s] = load_frag_global(kb + (long)kr * p.ld_qkv, s, g);
                vf[kk][s] = load_frag_global(vb + (long)kr * p.ld_qkv, s, g);
            }
        }
    }

    // ---- tile fetch (loads only, results stay in registers) / commit (arithmetic + LDS writes)
    u32x4 pre[NCH];
    bf16x8 pdo;                       // this thread's 8 dO values of its query (delta)
    f32x4 pa0, pa1;                   // the matching O values (fp32 copy) ...
    bf16x8 pab;                       // ... or bf16 O when no fp32 copy exists
    float plse = 0.f;
    auto fetch = [&](int i) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int c = tid + n * NT;
            if (c < 512) {
                const int cc = c & 255, row = cc >> 3, ch = cc & 7;
                int r = 32 * i + row; r = r < p.N ? r : p.N - 1;
                const bf16_t* src = (c < 256) ? qb + (long)r * p.ld_qkv : dob + (long)r * p.ld_o;
                pre[n] = *reinterpret_cast<const u32x4*>(src + ch * 8);
            }
        }
        if (tid < 256 && !(ABL & 4)) {
            // 8 lanes per query: delta partial over 8 of the 64 head columns
            const int q = tid >> 3, sub = tid & 7;
            const int qi = 32 * i + q, r = qi < p.N ? qi : p.N - 1;
            pdo = *reinterpret_cast<const bf16x8*>(dob + (long)r * p.ld_o + sub * 8);
            if (p.o32) {
                const float* a = p.o32 + (tok0 + r) * p.ld_o + h * DH + sub * 8;
                pa0 = *reinterpret_cast<const f32x4*>(a);
                pa1 = *reinterpret_cast<const f32x4*>(a + 4);
            } else {
                pab = *reinterpret_cast<const bf16x8*>(p.o + (tok0 + r) * p.ld_o + h * DH + sub * 8);
            }
            plse = lse[r];
        }
    };
    auto commit = [&](int i) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int c = tid + n * NT;
            if (c < 512) {
                const int cc = c & 255, row = cc >> 3, ch = cc & 7;
                char* dst = (c < 256 ? Qb : Ob) + (i & 1) * AT_TILE + row * AT_ROW + ch * 16;
                *reinterpret_cast<u32x4*>(dst) = pre[n];
            }
        }
        if (tid < 256 && !(ABL & 4)) {
            float d = 0.f;
            if (p.o32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) d += pa0[e] * (float)pdo[e] + pa1[e] * (float)pdo[4 + e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)pab[e] * (float)pdo[e];
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            const int q = tid >> 3, sub = tid & 7;
            const bool ok = 32 * i + q < p.N;
            if (sub == 0) Ds[(i & 1) * 32 + q] = ok ? d : 0.f;
            if (sub == 1) Ls[(i & 1) * 32 + q] = ok ? plse * LOG2E : 1e30f;   // exp2(s - 1e30) = 0 for padded queries
        }
    };
    // dQ tile `ti`: sum of the active waves' slots -> global bf16
    auto flush_dq = [&](int ti) {
        if (tid < 256) {
            const int q = tid >> 3, dch = tid & 7;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < ntile; ++w) {
                const float* a = dQs + w * DQ_SLOT + q * DQ_LD + 8 * dch;
                s0 += *reinterpret_cast<const f32x4*>(a);
                s1 += *reinterpret_cast<const f32x4*>(a + 4);
            }
            const int qi = 32 * ti + q;
            if (qi < p.N) {
                const float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                *reinterpret_cast<bf16x8*>(p.dq + (tok0 + qi) * p.ld_dqkv + h * DH + 8 * dch) = pack8(v, v + 4);
            }
        }
    };

    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const float sc = p.scale * LOG2E;

    fetch(0);
    commit(0);
    __syncthreads();
    for (int i = 0; i < ntile; ++i) {
        if (i + 1 < ntile) fetch(i + 1);
        f32x4 dqp[4][2];
        if (active) {
            const char* Qs = Qb + (i & 1) * AT_TILE;
            const char* Os = Ob + (i & 1) * AT_TILE;
            // S (rows = queries, cols = keys) and dP
            f32x4 sa[2][2], da[2][2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                bf16x8 qfr[2], ofr[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    qfr[s] = row_frag(Qs, 16 * qq, s, lane);
                    ofr[s] = row_frag(Os, 16 * qq, s, lane);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    sa[kk][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    da[kk][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ABL & 16) { sa[kk][qq][0] = (float)qfr[0][0]; da[kk][qq][1] = (float)ofr[1][1]; continue; }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        sa[kk][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[s], kf[kk][s], sa[kk][qq], 0, 0, 0);
                        da[kk][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr[s], vf[kk][s], da[kk][qq], 0, 0, 0);
                    }
                }
            }
            // P = exp2(S*sc - lse2), dS = P (dP - delta) scale; rows 16qq + 4g + r, col = key l15
            bf16x8 pb[2], dsb[2];
            {
                f32x4 l2v[2], dlv[2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    l2v[qq] = *reinterpret_cast<const f32x4*>(Ls + (i & 1) * 32 + 16 * qq + 4 * g);
                    dlv[qq] = *reinterpret_cast<const f32x4*>(Ds + (i & 1) * 32 + 16 * qq + 4 * g);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    float pv[8], dsv[8];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (ABL & 2) { pv[4 * qq + r] = sa[kk][qq][r]; dsv[4 * qq + r] = da[kk][qq][r]; continue; }
                            const float pr = __builtin_amdgcn_exp2f(sa[kk][qq][r] * sc - l2v[qq][r]) * kmask[kk];
                            pv[4 * qq + r] = pr;
                            dsv[4 * qq + r] = pr * (da[kk][qq][r] - dlv[qq][r]) * p.scale;
                        }
                    pb[kk] = pack8(pv, pv + 4);
                    dsb[kk] = pack8(dsv, dsv + 4);
                    // dS^T tile: row = key, 4 consecutive queries per store
#pragma unroll
                    for (int qq = 0; qq < 2 && !(ABL & 1); ++qq) {
                        const bf16x4 w4 = {dsb[kk][4 * qq], dsb[kk][4 * qq + 1], dsb[kk][4 * qq + 2], dsb[kk][4 * qq + 3]};
                        *reinterpret_cast<bf16x4*>(StW + (16 * kk + l15) * AT_ROW + (16 * qq + 4 * g) * 2) = w4;
                    }
                }
            }
            // dV += P^T dO, dK += dS^T Q  (reduction over the tile's 32 queries)
#pragma unroll
            for (int dt = 0; dt < 4 && !(ABL & 8); ++dt) {
                const bf16x8 otr = tr_frag32(Os, 16 * dt, lane);
                const bf16x8 qtr = tr_frag32(Qs, 16 * dt, lane);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    dv[kk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otr, pb[kk], dv[kk][dt], 0, 0, 0);
                    dk[kk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsb[kk], dk[kk][dt], 0, 0, 0);
                }
            }
            // dQ_i^T partial (rows = head columns, cols = queries) = K_w^T dS^T  (reduction over this wave's keys)
            asm volatile("" ::: "memory");
            bf16x8 bS[2];
#pragma unroll
            for (int qq = 0; qq < 2 && !(ABL & 1); ++qq) bS[qq] = tr_frag32(StW, 16 * qq, lane);
#pragma unroll
            for (int dt = 0; dt < 4 && !(ABL & 1); ++dt) {
                const bf16x8 aK = tr_frag32(KtW, 16 * dt, lane);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    dqp[dt][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aK, bS[qq], z, 0, 0, 0);
                }
            }
        }
        // The next tile pair goes into the other buffer (last read one step ago, before a barrier).
        if (i + 1 < ntile) commit(i + 1);
        __syncthreads();                       // the previous step's slot readers are done
        if (ABL & 32) {
            if (ABL & 1) continue;
            if (active) { float t = 0.f;
                for (int dt = 0; dt < 4; ++dt) for (int qq = 0; qq < 2; ++qq) t += dqp[dt][qq][0] + dqp[dt][qq][3];
                dk[0][0][0] += t; }
            continue;
        }
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)   // lane: query 16qq + l15, head columns 16dt + 4g .. +3
                    *reinterpret_cast<f32x4*>(dQW + (16 * qq + l15) * DQ_LD + 16 * dt + 4 * g) = dqp[dt][qq];
        }
        __syncthreads();                       // every wave's partial is parked; next tile pair visible
        flush_dq(i);
    }

    if (active) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int key = 32 * wid + 16 * kk + l15;
            if (key < p.N) {
                bf16_t* dkr = p.dk + (tok0 + key) * p.ld_dqkv + h * DH;
                bf16_t* dvr = p.dv + (tok0 + key) * p.ld_dqkv + h * DH;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4(dkr + 16 * dt + 4 * g, dk[kk][dt], 1.0f);
                    store4(dvr + 16 * dt + 4 * g, dv[kk][dt], 1.0f);
                }
            }
        }
    }
}
constexpr size_t attn_bwd_fused_lds(int nw) {
    return (size_t)(4 + 2 * nw) * AT_TILE + ((size_t)nw * DQ_SLOT + 128) * sizeof(float);
}

// ------------------------------------------------------------- fused backward, specialised waves
// Same arithmetic as attn_bwd_fused_bf16_kernel, different division of labour: NW compute waves
// (wave w owns key tile w: K_w, V_w, dK_w, dV_w) never touch global memory inside the loop, and one
// extra wave feeds them and finishes dQ:
//   * Q_i, dO_i, the fp32 copy of O_i and LSE_i arrive by LDS-DMA in rings of four slots, three
//     tiles ahead, so the memory latency of a tile is spread over three steps instead of being paid
//     in each one; the feeder turns (dO_i, O_i, LSE_i) into delta_i = rowsum(dO_i o O_i) and
//     LSE_i log2(e) one step before the compute waves need them;
//   * the compute waves leave dS^T (bf16) in a double-buffered LDS tile each; the feeder forms
//     dQ_{i-1} = sum_w dS_{i-1,w} K_w with MFMAs during step i and stores it.  Exchanging dS (2 KiB
//     per wave and step) replaces fp32 dQ partial slots (8 KiB per wave and step, plus a sum).
// One barrier per step.  A DMA instruction writes 1 KiB linearly, so a padded 160-byte-row tile is
// filled as 320 chunks of 16 bytes of which every 9th and 10th per row are padding (their lanes
// re-load chunk 7 of the row).  The DMA is issued from inline assembly: the feeder has no other
// vector-memory loads, so its explicit vmcnt waits are exact, and the compiler does not serialise
// the wave's LDS reads behind DMA it would have to assume aliases them.
constexpr int WS_RING = 4;
constexpr int WS_DMA = AT_TILE / 1024;        // DMA instructions per bf16 tile
constexpr int WS_GROUP = 2 * WS_DMA + 8 + 1;  // DMA instructions per step: Q, dO, 8 KiB of fp32 O, LSE
constexpr int ST_ROW = 96;                    // dS^T tile: 32 keys x 32 queries, 64 data bytes + pad per row
constexpr int ST_TILE = 32 * ST_ROW;
static_assert(WS_GROUP == 19, "the vmcnt immediates below are multiples of 19");

#define VITK_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define VITK_AT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(VITK_LDS const char*)p;
}
// lane l: 16 bytes at gsrc -> LDS lds_base + 16 l   /   4 bytes -> lds_base + 4 l
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_base) : "memory");
}
template <int CTRL>
__device__ __forceinline__ float row_ror_t(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define row_ror(v, ctrl) row_ror_t<ctrl>(v)
// tr_frag32 for the 96-byte-row dS^T tiles
__device__ __forceinline__ bf16x8 tr_frag32_st(const char* lds, int d0, int lane) {
    const int l15 = lane & 15, gq = lane >> 4;
    const char* a0 = lds + (4 * gq + (l15 >> 2)) * ST_ROW + (d0 + 4 * (l15 & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0 + 16 * ST_ROW));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// lane l: 16 bytes at sbase + voff -> LDS lds_base + 16 l   /   4 bytes -> lds_base + 4 l
__device__ __forceinline__ void dma16s(const void* sbase, int voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma4s(const void* sbase, int voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

// dQ of one 32-query tile = sum_w dS_w K_w is eight 16x16 MFMA output tiles (head-column block dt,
// query block qq).  They are dealt out to the NW compute waves and the feeder (worker NW): worker x
// finishes the tiles u = x, x + NW + 1, ... < 8, with the K_w^T fragments of its head-column blocks
// in registers for the whole pair.
template <int NW>
struct DqDuty {
    static constexpr int MT = (8 + NW) / (NW + 1);
    bf16x8 kfr[MT][NW];
    __device__ __forceinline__ void load(const char* Kt, int worker, int lane) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int u = worker + m * (NW + 1);
            if (u < 8) {
#pragma unroll
                for (int w = 0; w < NW; ++w) kfr[m][w] = tr_frag32(Kt + w * AT_TILE, 16 * (u >> 1), lane);
            }
        }
    }
    // Sb: the dS^T tiles of the step being finished; dq_tile: first row of the tile's dQ rows for this head
    __device__ __forceinline__ void run(const char* Sb, int worker, int lane, bf16_t* dq_tile, long ld, int rows, float scale) const {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int u = worker + m * (NW + 1);
            if (u < 8) {
                const int dt = u >> 1, qq = u & 1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < NW; ++w)       // tiles of idle compute waves are zero
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[m][w], tr_frag32_st(Sb + w * ST_TILE, 16 * qq, lane), acc, 0, 0, 0);
                const int q = 16 * qq + (lane & 15);      // lane: query q, head columns 16dt + 4g .. +3
                if (q < rows) store4(dq_tile + (long)q * ld + 16 * dt + 4 * (lane >> 4), acc, scale);
            }
        }
    }
};

// The workgroup is persistent: it walks the (image, head) pairs blockIdx.x, blockIdx.x + gridDim.x, ...
// as ONE stream of steps t = pair ordinal * ntile + tile, so the feeder's look-ahead runs across
// pair boundaries and the compute waves fetch the next pair's K/V two steps before they need them.
// Launch, first-load latency and the drain of a pair are then paid once per workgroup, not per pair.
// Needs ntile >= 2 (with one tile the feeder would still read a pair's K tile while the next pair's
// is written).
template <int NW, int ABL = 0>
__global__ __launch_bounds__((NW + 1) * 64, 1) void attn_bwd_ws_bf16_kernel(AttnArgs p, int ntile, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qr = smem;                                              // [WS_RING][2][AT_TILE]  Q_i, dO_i
    char* Or = Qr + WS_RING * 2 * AT_TILE;                        // [WS_RING][8192]  fp32 O_i, row-major 32 x 64
    char* Lr = Or + WS_RING * 8192;                               // [WS_RING][256]   raw LSE_i (64 floats, 32 used)
    char* Kt = Lr + WS_RING * 256;                                // [NW][AT_TILE]
    char* St = Kt + NW * AT_TILE;                                 // [2][NW][ST_TILE] dS as [key][query]
    float* Ls = reinterpret_cast<float*>(St + 2 * NW * ST_TILE);  // [WS_RING][32] LSE * log2e
    float* Ds = Ls + 32 * WS_RING;                                // [WS_RING][32] delta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int cnt = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // pairs of this workgroup
    const int T = cnt * ntile;                                                          // steps of this workgroup
    const int last_row = p.N - 1 - 32 * (ntile - 1);                                    // last valid row of the last tile

    if (wid == NW) {
        // ------------------------------------------------------------ feeder / dQ wave
        // its LDS-DMA issue paces every step: ahead of the compute wave it shares a SIMD with (-3.5 % per call;
        // the forward kernel's feeder measured 2 % slower with the same priority and keeps the default)
        __builtin_amdgcn_s_setprio(3);
        // per-lane byte offsets inside a tile (second set: rows clamped for the last tile)
        int oq[2][WS_DMA], od[2][WS_DMA], oo[2][8], ol[2];
#pragma unroll
        for (int n = 0; n < WS_DMA; ++n) {
            const int c = 64 * n + lane;
            const int row = c / 10, ch = c - 10 * row;
            const int cb = (ch < 8 ? ch : 7) * 16;
            const int rc = row < last_row ? row : last_row;
            oq[0][n] = row * (int)p.ld_qkv * 2 + cb; oq[1][n] = rc * (int)p.ld_qkv * 2 + cb;
            od[0][n] = row * (int)p.ld_o * 2 + cb;   od[1][n] = rc * (int)p.ld_o * 2 + cb;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {              // chunk 64n + lane of the 32 x 64 fp32 tile: row 4n + lane/16
            const int row = 4 * n + g, rc = row < last_row ? row : last_row;
            oo[0][n] = (row * (int)p.ld_o + 4 * l15) * 4; oo[1][n] = (rc * (int)p.ld_o + 4 * l15) * 4;
        }
        {
            const int row = lane & 31, rc = row < last_row ? row : last_row;
            ol[0] = row * 4; ol[1] = rc * 4;
        }
        const unsigned qr0 = lds_addr_of(Qr), or0 = lds_addr_of(Or), lr0 = lds_addr_of(Lr);

        // stream cursors: (pair ordinal, tile) of the group to fetch, and of the dQ tile to finish
        int fk = 0, fi = 0, dk_ = 0, di = 0;
        const bf16_t *f_q = nullptr, *f_do = nullptr; const float *f_o = nullptr, *f_l = nullptr;
        bf16_t* d_dq = nullptr;
        auto fetch_bases = [&]() {
            const int bh = blockIdx.x + fk * gridDim.x, b = bh / p.H, h = bh - b * p.H;
            const long tok0 = (long)b * p.N;
            f_q = p.q + tok0 * p.ld_qkv + h * DH;
            f_do = p.d_o + tok0 * p.ld_o + h * DH;
            f_o = p.o32 + tok0 * p.ld_o + h * DH;
            f_l = p.lse + (long)bh * p.N;
        };
        auto dq_bases = [&]() {
            const int bh = blockIdx.x + dk_ * gridDim.x, b = bh / p.H, h = bh - b * p.H;
            d_dq = p.dq + (long)b * p.N * p.ld_dqkv + h * DH;
        };
        fetch_bases();
        dq_bases();
        // K tile w of pair ordinal k into its LDS tile (rows clamped like the Q tiles)
        auto dma_k = [&](int k, int w) {
            const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
            const bf16_t* tk = p.k + ((long)b * p.N + 32 * w) * p.ld_qkv + h * DH;
            const unsigned ks = lds_addr_of(Kt) + w * AT_TILE;
            const bool lt = w == ntile - 1;
#pragma unroll
            for (int n = 0; n < WS_DMA; ++n) dma16s(tk, lt ? oq[1][n] : oq[0][n], ks + n * 1024);
        };
        auto dma_next = [&](int t) {               // group of stream position t = (fk, fi); advances the cursor
            const int sl = t & (WS_RING - 1);
            const unsigned qs = qr0 + sl * 2 * AT_TILE;
            const bool lt = fi == ntile - 1;
            const bf16_t* tq = f_q + (long)(32 * fi) * p.ld_qkv;
            const bf16_t* td = f_do + (long)(32 * fi) * p.ld_o;
            const float* to = f_o + (long)(32 * fi) * p.ld_o;
#pragma unroll
            for (int n = 0; n < WS_DMA; ++n) {
                dma16s(tq, lt ? oq[1][n] : oq[0][n], qs + n * 1024);
                dma16s(td, lt ? od[1][n] : od[0][n], qs + AT_TILE + n * 1024);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) dma16s(to, lt ? oo[1][n] : oo[0][n], or0 + sl * 8192 + n * 1024);
            dma4s(f_l + 32 * fi, lt ? ol[1] : ol[0], lr0 + sl * 256);
            if (++fi == ntile) { fi = 0; ++fk; if (fk < cnt) fetch_bases(); }
        };
        // delta and LSE of stream position t (tile `ti` of its pair) out of the landed slot
        auto delta_tile = [&](int t, int ti) {
            const int sl = t & (WS_RING - 1);
            const char* Os = Or + sl * 8192;
            const char* dOs = Qr + sl * 2 * AT_TILE + AT_TILE;
            float mine = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x4 o4 = *reinterpret_cast<const f32x4*>(Os + (64 * e + lane) * 16);
                const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dOs + (4 * e + g) * AT_ROW + 8 * l15);
                float pe = o4[0] * (float)d4[0] + o4[1] * (float)d4[1] + o4[2] * (float)d4[2] + o4[3] * (float)d4[3];
                pe += row_ror(pe, 0x128);           // sum over the 16 lanes of a DPP row (rotations by 8, 4, 2, 1)
                pe += row_ror(pe, 0x124);
                pe += row_ror(pe, 0x122);
                pe += row_ror(pe, 0x121);
                mine = (l15 == e) ? pe : mine;
            }
            if (l15 < 8) {                          // row 4 l15 + g of the tile
                const int q = 4 * l15 + g;
                Ds[32 * sl + q] = (32 * ti + q < p.N) ? mine : 0.f;
            }
            if (lane < 32) {
                const float raw = *reinterpret_cast<const float*>(Lr + sl * 256 + 4 * lane);
                Ls[32 * sl + lane] = (32 * ti + lane < p.N) ? raw * LOG2E : 1e30f;   // exp2(s - 1e30) = 0 for padded queries
            }
        };
        // this wave's share of the dQ tile of stream position t; advances the cursor
        DqDuty<NW> duty;
        auto dq_next = [&](int t) {
            duty.run(St + (t & 1) * NW * ST_TILE, NW, lane, d_dq + (long)(32 * di) * p.ld_dqkv, p.ld_dqkv, p.N - 32 * di, p.scale);
            if (++di == ntile) { di = 0; ++dk_; if (dk_ < cnt) dq_bases(); }
        };
        // Wait until only the `groups` youngest DMA groups and `ktiles` K tiles issued with them may
        // still be in flight.  Loads retire in order; outstanding dQ stores only make the wait stricter.
        auto vm_allow_groups = [&](int groups, int ktiles) {
            switch (WS_GROUP * groups + WS_DMA * ktiles) {
                case 53: VITK_AT_VMCNT(53); break;
                case 43: VITK_AT_VMCNT(43); break;
                case 38: VITK_AT_VMCNT(38); break;
                case 34: VITK_AT_VMCNT(34); break;
                case 24: VITK_AT_VMCNT(24); break;
                case 19: VITK_AT_VMCNT(19); break;
                case 15: VITK_AT_VMCNT(15); break;
                case 5: VITK_AT_VMCNT(5); break;
                default: VITK_AT_VMCNT(0); break;
            }
        };

        for (int w = 0; w < ntile; ++w) dma_k(0, w);
        dma_next(0);
        if (T > 1) dma_next(1);
        if (T > 2) dma_next(2);
        vm_allow_groups((T > 1) + (T > 2), 0);
        delta_tile(0, 0);
        VITK_LDS_BARRIER();
        int ti1 = ntile > 1 ? 1 : 0;               // tile index of stream position t + 1
        int ti0 = 0;                               // tile index of stream position t
        int pk = 0;                                // pair ordinal of stream position t
        for (int t = 0; t < T; ++t) {
            // The next pair's K tiles go into the K tiles once step 0 of this pair has read them: three
            // in step 1, one per step after that, the last one a full step before the pair ends.  They
            // are issued ahead of this step's group, so the next step's wait covers them.
            int ktiles = 0;
            if (!(ABL & 64) && pk + 1 < cnt && ti0 >= 1) {     // (ABL 64: no DMA beyond the prologue: stale tiles, times only)
                if (ti0 == 1) {
                    for (int w = 0; w < 3 && w < ntile; ++w, ++ktiles) dma_k(pk + 1, w);
                } else if (ti0 + 1 < ntile) {
                    dma_k(pk + 1, ti0 + 1);
                    ktiles = 1;
                }
            }
            if (!(ABL & 64) && t + 3 < T) dma_next(t + 3);        // slots of position t-1, last read before the previous barrier
            // dQ first: it needs nothing that is still in flight, and the group below gets that long to land
            if (t >= 1 && !(ABL & 1)) dq_next(t - 1);
            if (ti0 == 0) duty.load(Kt, NW, lane); // this pair's K tiles landed before the last barrier
            if (t + 1 < T) {
                if ((ABL & 64) || (ntile == 2 && ktiles)) VITK_AT_VMCNT(0);    // two-tile pairs: the K tiles are needed after this very step
                else vm_allow_groups((t + 2 < T) + (t + 3 < T), ktiles);   // group t+1 has landed
                if (!(ABL & 2)) delta_tile(t + 1, ti1);
            }
            ti0 = ti1;
            if (ti0 == 0) ++pk;
            if (++ti1 == ntile) ti1 = 0;
            VITK_LDS_BARRIER();
        }
        if (!(ABL & 1)) dq_next(T - 1);
        return;
    }

    // ---------------------------------------------------------------- compute waves
    const bool active = wid < ntile;
    const bool last_wave = wid == ntile - 1;       // the only wave whose key tile can hold padded keys
    char* KtW = Kt + wid * AT_TILE;
    bf16x8 kf[2][2], vf[2][2], vfn[2][2];
    float kmask[2];
    kmask[0] = 32 * wid + l15 < p.N ? 1.0f : 0.0f;
    kmask[1] = 32 * wid + 16 + l15 < p.N ? 1.0f : 0.0f;
    // V row fragments of pair ordinal k (rows clamped to the sequence); K comes through the LDS tile
    auto fetch_v = [&](int k) {
        const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
        const bf16_t* vb = p.v + (long)b * p.N * p.ld_qkv + h * DH;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int key = 32 * wid + 16 * kk + l15;
            const int kr = key < p.N ? key : p.N - 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) vfn[kk][s] = load_frag_global(vb + (long)kr * p.ld_qkv, s, g);
        }
    };
    if (active) {
        fetch_v(0);
    } else {
        // an idle wave's K and dS^T tiles still enter the feeder's dQ sum: make them zero
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int c = lane; c < AT_TILE / 16; c += 64) *reinterpret_cast<u32x4*>(KtW + 16 * c) = z;
        for (int c = lane; c < ST_TILE / 16; c += 64) {
            *reinterpret_cast<u32x4*>(St + wid * ST_TILE + 16 * c) = z;
            *reinterpret_cast<u32x4*>(St + (NW + wid) * ST_TILE + 16 * c) = z;
        }
    }
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const float sc = p.scale * LOG2E;
    const int pre_i = ntile >= 3 ? ntile - 3 : 0;  // step of a pair in which the next pair's V rows are requested

    // One step on stream position t.  MASKED: multiply P by the key mask (last key tile only).
    auto compute = [&](int t, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        const int sl = t & (WS_RING - 1);
        const char* Qs = Qr + sl * 2 * AT_TILE;
        const char* Os = Qs + AT_TILE;
        char* StW = St + ((t & 1) * NW + wid) * ST_TILE;
        // S (rows = queries, cols = keys) and dP
        f32x4 sa[2][2], da[2][2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            bf16x8 qfr[2], ofr[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qfr[s] = row_frag(Qs, 16 * qq, s, lane);
                ofr[s] = row_frag(Os, 16 * qq, s, lane);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                sa[kk][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
                da[kk][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa[kk][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[s], kf[kk][s], sa[kk][qq], 0, 0, 0);
                    da[kk][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr[s], vf[kk][s], da[kk][qq], 0, 0, 0);
                }
            }
        }
        // P = exp2(S*sc - lse2), dS/scale = P (dP - delta); rows 16qq + 4g + r, col = key l15.  The
        // softmax scale of dS is applied once to dK and dQ when they are stored.
        bf16x8 pb[2], dsb[2];
        {
            f32x4 l2v[2], dlv[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                l2v[qq] = *reinterpret_cast<const f32x4*>(Ls + 32 * sl + 16 * qq + 4 * g);
                dlv[qq] = *reinterpret_cast<const f32x4*>(Ds + 32 * sl + 16 * qq + 4 * g);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float pv[8], dsv[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = __builtin_amdgcn_exp2f(sa[kk][qq][r] * sc - l2v[qq][r]);
                        if (MASKED) pr *= kmask[kk];
                        pv[4 * qq + r] = pr;
                        dsv[4 * qq + r] = pr * (da[kk][qq][r] - dlv[qq][r]);
                    }
                pb[kk] = pack8(pv, pv + 4);
                dsb[kk] = pack8(dsv, dsv + 4);
                // dS^T tile: row = key, 4 consecutive queries per store
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const bf16x4 w4 = {dsb[kk][4 * qq], dsb[kk][4 * qq + 1], dsb[kk][4 * qq + 2], dsb[kk][4 * qq + 3]};
                    *reinterpret_cast<bf16x4*>(StW + (16 * kk + l15) * ST_ROW + (16 * qq + 4 * g) * 2) = w4;
                }
            }
        }
        // dV += P^T dO, dK/scale += dS^T Q  (reduction over the tile's 32 queries)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 otr = tr_frag32(Os, 16 * dt, lane);
            const bf16x8 qtr = tr_frag32(Qs, 16 * dt, lane);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                dv[kk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otr, pb[kk], dv[kk][dt], 0, 0, 0);
                dk[kk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsb[kk], dk[kk][dt], 0, 0, 0);
            }
        }
    };

    VITK_LDS_BARRIER();
    // The whole stream, once per mask variant: a branch on the variant inside the loops makes the
    // compiler shuttle the 64 accumulator registers between two copies every step.
    auto stream = [&](auto masked) {
    DqDuty<NW> duty;
    bf16_t* dq_prev = nullptr;                     // dQ rows of the tile of stream position t - 1
    int rows_prev = 0;
    int t = 0;
    for (int k = 0; k < cnt; ++k) {
        const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
        const long tok0 = (long)b * p.N;
        bf16_t* dq_pair = p.dq + tok0 * p.ld_dqkv + h * DH;
        for (int i = 0; i < ntile; ++i, ++t) {
            // finish dQ of the previous step (idle waves take part), with the K fragments of ITS pair
            if (t >= 1 && !(ABL & 1))
                duty.run(St + ((t - 1) & 1) * NW * ST_TILE, wid, lane, dq_prev, p.ld_dqkv, rows_prev, p.scale);
            if (i == 0) {                          // this pair's K tiles landed before the last barrier
                duty.load(Kt, wid, lane);
                if (active) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            kf[kk][s2] = row_frag(KtW, 16 * kk, s2, lane);
                            vf[kk][s2] = vfn[kk][s2];
                        }
                }
            }
            if (active && !(ABL & 128)) {
                if (i == pre_i && k + 1 < cnt) fetch_v(k + 1);
                compute(t, masked);
            }
            dq_prev = dq_pair + (long)(32 * i) * p.ld_dqkv;
            rows_prev = p.N - 32 * i;
            VITK_LDS_BARRIER();
        }
        if (active) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int key = 32 * wid + 16 * kk + l15;
                if (key < p.N) {
                    bf16_t* dkr = p.dk + (tok0 + key) * p.ld_dqkv + h * DH;
                    bf16_t* dvr = p.dv + (tok0 + key) * p.ld_dqkv + h * DH;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        store4(dkr + 16 * dt + 4 * g, dk[kk][dt], p.scale);
                        store4(dvr + 16 * dt + 4 * g, dv[kk][dt], 1.0f);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dk[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dv[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    if (!(ABL & 1)) duty.run(St + ((T - 1) & 1) * NW * ST_TILE, wid, lane, dq_prev, p.ld_dqkv, rows_prev, p.scale);
    };
    if (last_wave) stream(std::true_type{});
    else stream(std::false_type{});
}
constexpr size_t attn_bwd_ws_lds(int nw) {
    return (size_t)WS_RING * (2 * AT_TILE + 8192 + 256 + 64 * sizeof(float)) + (size_t)nw * (AT_TILE + 2 * ST_TILE);
}

// ------------------------------------------------------------- forward, specialised waves
// Persistent workgroups of sixteen waves (one per CU): waves 0..13 each own one 16-query tile of
// the current (image, head) pair, wave 15 is a feeder that brings the NEXT pair's K and V tiles in
// by LDS-DMA while the current pair is worked on (two K/V buffers, one barrier per pair).  In the
// one-shot kernel above a workgroup loads, computes and stores in sequence and only its CU-mate
// overlaps with it: the three phases cost 0.07 + 0.15 + 0.085 ms per layer at batch 512 and added
// up to 0.29; here the loads of pair k+1, the arithmetic of pair k and the draining stores of pair
// k-1 run side by side.
constexpr int FWS_WAVES = 16, FWS_FEEDER = 15;

template <int ABL = 0>
__global__ __launch_bounds__(FWS_WAVES * 64, 1) void attn_fwd_ws_bf16_kernel(AttnArgs p, int ntile, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K | V][RES_TILES][AT_TILE]
    constexpr int BUF = 2 * 7 * AT_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int cnt = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    if (wid == FWS_FEEDER) {
        const int last_row = p.N - 1 - 32 * (ntile - 1);
        int ok[2][WS_DMA];
#pragma unroll
        for (int n = 0; n < WS_DMA; ++n) {
            const int c = 64 * n + lane;
            const int row = c / 10, ch = c - 10 * row;
            const int cb = (ch < 8 ? ch : 7) * 16;
            const int rc = row < last_row ? row : last_row;
            ok[0][n] = row * (int)p.ld_qkv * 2 + cb;
            ok[1][n] = rc * (int)p.ld_qkv * 2 + cb;
        }
        const unsigned s0 = lds_addr_of(smem);
        auto dma_pair = [&](int k) {
            const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
            const bf16_t* kb = p.k + (long)b * p.N * p.ld_qkv + h * DH;
            const bf16_t* vb = p.v + (long)b * p.N * p.ld_qkv + h * DH;
            const unsigned base = s0 + (k & 1) * BUF;
            for (int w = 0; w < ntile; ++w) {
                const bool lt = w == ntile - 1;
                const bf16_t* tk = kb + (long)(32 * w) * p.ld_qkv;
                const bf16_t* tv = vb + (long)(32 * w) * p.ld_qkv;
#pragma unroll
                for (int n = 0; n < WS_DMA; ++n) {
                    dma16s(tk, lt ? ok[1][n] : ok[0][n], base + w * AT_TILE + n * 1024);
                    dma16s(tv, lt ? ok[1][n] : ok[0][n], base + (7 + w) * AT_TILE + n * 1024);
                }
                VITK_AT_VMCNT(40);                 // keep the 6-bit counter far from its end
            }
        };
        dma_pair(0);
        VITK_AT_VMCNT(0);
        VITK_LDS_BARRIER();
        for (int k = 0; k < cnt; ++k) {
            if (k + 1 < cnt) dma_pair(k + 1);      // the other buffer: last read before the barrier just passed
            VITK_AT_VMCNT(0);
            VITK_LDS_BARRIER();
        }
        return;
    }

    const float sc = p.scale * LOG2E;
    const int nq = (p.N + 15) >> 4;
    const bool active = wid < nq;                  // waves nq..14 only keep the barrier count
    const int qi = wid * 16 + l15;
    const int qr = qi < p.N ? qi : p.N - 1;
    bf16x8 qn[2];
    auto fetch_q = [&](int k) {
        const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
        const bf16_t* qrow = p.q + ((long)b * p.N + qr) * p.ld_qkv + h * DH;
        qn[0] = load_frag_global(qrow, 0, g);
        qn[1] = load_frag_global(qrow, 1, g);
    };
    if (active) fetch_q(0);
    VITK_LDS_BARRIER();
    for (int k = 0; k < cnt; ++k) {
        if (active) {
            const int bh = blockIdx.x + k * gridDim.x, b = bh / p.H, h = bh - b * p.H;
            const long tok0 = (long)b * p.N;
            const char* Kall = smem + (k & 1) * BUF;
            const char* Vall = Kall + 7 * AT_TILE;
            bf16x8 qf[2] = {qn[0], qn[1]};
            if (k + 1 < cnt) fetch_q(k + 1);
            FwdState st;
            fwd_init(st);
            Scores cur = fwd_scores(Kall, qf, lane);
            // whole key tiles: no key index is looked at; two per trip, the scores in alternating registers
            int kt = 0;
            for (; kt + 2 < ntile; kt += 2) {
                const Scores odd = fwd_scores(Kall + (kt + 1) * AT_TILE, qf, lane);
                fwd_update<false, false>(st, cur, Vall + kt * AT_TILE, kt * 32, p.N, sc, lane);
                cur = fwd_scores(Kall + (kt + 2) * AT_TILE, qf, lane);
                fwd_update<false, false>(st, odd, Vall + (kt + 1) * AT_TILE, (kt + 1) * 32, p.N, sc, lane);
            }
            if (kt + 1 < ntile) {
                const Scores last = fwd_scores(Kall + (kt + 1) * AT_TILE, qf, lane);
                fwd_update<false, false>(st, cur, Vall + kt * AT_TILE, kt * 32, p.N, sc, lane);
                cur = last;
            }
            fwd_update<false, true>(st, cur, Vall + (ntile - 1) * AT_TILE, (ntile - 1) * 32, p.N, sc, lane);
            fwd_finish(st, p, tok0, bh, h, qi, lane, sc);
        }
        VITK_LDS_BARRIER();
    }
}

// Short sequences keep one operand pair resident in LDS (two workgroups per CU up to 7 tiles).
constexpr int RES_MAX_TILES = 7;
static int g_attn_force_streaming = 0;
static int g_attn_oversubscribe = 1;   // persistent attention-backward grid = this many x CUs (see vitk.h)

template <typename K>
int set_lds(K kernel, size_t lds) {
    return (int)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
// The dynamic-LDS limit of a kernel and the CU count are per DEVICE: the "done once" notes below are tables
// indexed by the current device (slot 0 for a device id beyond the table: the attribute is then set again, harmless).
constexpr int ATTN_MAX_DEV = 64;
inline int attn_dev_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ATTN_MAX_DEV) return 0;
    return dev;
}

}  // namespace

extern "C" void vitk_debug_force_attn_streaming(int on) { g_attn_force_streaming = on; }
extern "C" int vitk_attn_set_oversubscription(int factor) {
    if (factor < 1 || factor > 8) return VITK_ERR_SHAPE;
    g_attn_oversubscribe = factor;
    return 0;
}

int vitk_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* o32, float* lse, int B, int H,
                       int N, long ld_qkv, long ld_o, float scale, float dropout_p, uint64_t seed, hipStream_t st) {
    if (ld_qkv % 8 || ld_o % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return VITK_ERR_SHAPE;
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
    a.out = (bf16_t*)o; a.lse = lse; a.o32 = o32;
    a.B = B; a.H = H; a.N = N; a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.scale = scale;
    a.thr16 = vitk_drop_thr16(dropout_p); a.dscale = vitk_drop_scale(a.thr16); a.seed = seed;
    if (a.thr16) {          // dropout on the probabilities: the streaming kernel carries the mask
        hipLaunchKernelGGL(attn_fwd_bf16_kernel<true>, dim3(vitk_ceil_div(N, 64), B * H), dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    const int ntile = (N + 31) / 32;
    if (ntile <= RES_MAX_TILES && g_attn_force_streaming == 0 && 32 * ld_qkv * 2 < (1L << 30)) {
        static int num_cu_of[ATTN_MAX_DEV] = {};
        int& num_cu = num_cu_of[attn_dev_slot()];
        constexpr size_t lds_ws = 2 * 2 * 7 * (size_t)AT_TILE;
        if (!num_cu) {
            int e = set_lds(attn_fwd_ws_bf16_kernel<0>, lds_ws); if (e) return e;
            int dev = 0, n = 0;
            hipError_t he = hipGetDevice(&dev);
            if (he == hipSuccess) he = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            if (he != hipSuccess) return (int)he;
            if (n <= 0) return VITK_ERR_UNSUPPORTED;
            num_cu = n;
        }
        const int npairs = B * H;
        const int nwg = npairs < num_cu ? npairs : num_cu;
        hipLaunchKernelGGL(attn_fwd_ws_bf16_kernel<0>, dim3(nwg), dim3(FWS_WAVES * 64), lds_ws, st, a, ntile, npairs);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    if (ntile <= RES_MAX_TILES && g_attn_force_streaming != 1) {
        const size_t lds = 2 * (size_t)ntile * AT_TILE;
        const int nq = (N + 15) / 16;
        if (nq > 8) {
            static bool once_of[ATTN_MAX_DEV] = {}; bool& once = once_of[attn_dev_slot()];
            if (!once) { int e = set_lds(attn_fwd_res_bf16_kernel<7>, 2 * RES_MAX_TILES * AT_TILE); if (e) return e; once = true; }
#ifdef VITK_ATTN_ABL
            {       // ablation build only (tools): VITK_ATTN_FWD_ABL=<mask>: 1 no K/V load, 2 one key tile only, 4 no stores
                const char* ev = getenv("VITK_ATTN_FWD_ABL");
                const int abl = ev ? atoi(ev) : 0;
#define FWD_ABL_CASE(V) case V: { int e2 = set_lds(attn_fwd_res_bf16_kernel<7, V>, 2 * RES_MAX_TILES * AT_TILE); if (e2) return e2; \
                    hipLaunchKernelGGL((attn_fwd_res_bf16_kernel<7, V>), dim3(B * H), dim3(448), lds, st, a, ntile); \
                    VITK_CHECK_LAUNCH(); return 0; }
                switch (abl) { FWD_ABL_CASE(1) FWD_ABL_CASE(2) FWD_ABL_CASE(3) FWD_ABL_CASE(4) FWD_ABL_CASE(5) FWD_ABL_CASE(6) FWD_ABL_CASE(7) default: break; }
            }
#endif
            hipLaunchKernelGGL(attn_fwd_res_bf16_kernel<7>, dim3(B * H), dim3(448), lds, st, a, ntile);
        } else {
            static bool once_of[ATTN_MAX_DEV] = {}; bool& once = once_of[attn_dev_slot()];
            if (!once) { int e = set_lds(attn_fwd_res_bf16_kernel<4>, 2 * RES_MAX_TILES * AT_TILE); if (e) return e; once = true; }
            hipLaunchKernelGGL(attn_fwd_res_bf16_kernel<4>, dim3(B * H), dim3(256), lds, st, a, ntile);
        }
    } else {
        hipLaunchKernelGGL(attn_fwd_bf16_kernel<false>, dim3(vitk_ceil_div(N, 64), B * H), dim3(256), 0, st, a);
    }
    VITK_CHECK_LAUNCH();
    return 0;
}

int vitk_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const float* o32, const void* d_o,
                       const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H,
                       int N, long ld_qkv, long ld_o, long ld_dqkv, float scale, float dropout_p, uint64_t seed,
                       hipStream_t st) {
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
    a.thr16 = vitk_drop_thr16(dropout_p); a.dscale = vitk_drop_scale(a.thr16); a.seed = seed;
    if (a.thr16) {          // dropout on the probabilities: the three streaming launches carry the mask
        const long units = (long)B * N * H * 8;
        hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(vitk_ceil_div(units, 256)), dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH();
        const dim3 grid(vitk_ceil_div(N, 64), B * H);
        hipLaunchKernelGGL(attn_bwd_dkdv_bf16_kernel<true>, grid, dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<true>, grid, dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    const int ntile = (N + 31) / 32;
    if (ntile >= 2 && ntile <= RES_MAX_TILES && g_attn_force_streaming == 0 && o32 &&
        32 * ld_qkv * 2 < (1L << 30) && 32 * ld_o * 4 < (1L << 30)) {     // per-lane 32-bit tile offsets
        static int num_cu_of[ATTN_MAX_DEV] = {};
        int& num_cu = num_cu_of[attn_dev_slot()];
        if (!num_cu) {
            int e = set_lds(attn_bwd_ws_bf16_kernel<7>, attn_bwd_ws_lds(7)); if (e) return e;
            e = set_lds(attn_bwd_ws_bf16_kernel<4>, attn_bwd_ws_lds(4)); if (e) return e;
            int dev = 0, n = 0;
            hipError_t he = hipGetDevice(&dev);
            if (he == hipSuccess) he = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            if (he != hipSuccess) return (int)he;
            if (n <= 0) return VITK_ERR_UNSUPPORTED;
            num_cu = n;
        }
        const int npairs = B * H;
        const int nwg = npairs < num_cu * g_attn_oversubscribe ? npairs : num_cu * g_attn_oversubscribe;   // persistent workgroups walk the pairs
#ifdef VITK_ATTN_ABL
        if (ntile > 4) {      // ablation build only (tools): VITK_ATTN_ABL=<mask> drops parts of the kernel
            const char* ev = getenv("VITK_ATTN_ABL");
            const int abl = ev ? atoi(ev) : 0;
#define WS_ABL_CASE(V) case V: { int e2 = set_lds(attn_bwd_ws_bf16_kernel<7, V>, attn_bwd_ws_lds(7)); if (e2) return e2; \
                hipLaunchKernelGGL((attn_bwd_ws_bf16_kernel<7, V>), dim3(nwg), dim3(512), attn_bwd_ws_lds(7), st, a, ntile, npairs); \
                VITK_CHECK_LAUNCH(); return 0; }
            switch (abl) { WS_ABL_CASE(1) WS_ABL_CASE(2) WS_ABL_CASE(3) WS_ABL_CASE(64) WS_ABL_CASE(67) WS_ABL_CASE(128) WS_ABL_CASE(131) default: break; }
        }
#endif
        if (ntile > 4)
            hipLaunchKernelGGL(attn_bwd_ws_bf16_kernel<7>, dim3(nwg), dim3(512), attn_bwd_ws_lds(7), st, a, ntile, npairs);
        else
            hipLaunchKernelGGL(attn_bwd_ws_bf16_kernel<4>, dim3(nwg), dim3(320), attn_bwd_ws_lds(4), st, a, ntile, npairs);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    if (ntile <= RES_MAX_TILES && (g_attn_force_streaming == 3 || g_attn_force_streaming == 0)) {
        static bool once_of[ATTN_MAX_DEV] = {}; bool& once = once_of[attn_dev_slot()];
        if (!once) {
            int e = set_lds(attn_bwd_fused_bf16_kernel<7>, attn_bwd_fused_lds(7)); if (e) return e;
            e = set_lds(attn_bwd_fused_bf16_kernel<4>, attn_bwd_fused_lds(4)); if (e) return e;
            once = true;
        }
#ifdef VITK_ATTN_ABL
        if (ntile > 4) {      // ablation build only (tools): VITK_ATTN_ABL=<mask> drops parts of the kernel
            const char* ev = getenv("VITK_ATTN_ABL");
            const int abl = ev ? atoi(ev) : 0;
#define ABL_CASE(V) case V: { int e2 = set_lds(attn_bwd_fused_bf16_kernel<7, V>, attn_bwd_fused_lds(7)); if (e2) return e2; \
                hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<7, V>), dim3(B * H), dim3(448), attn_bwd_fused_lds(7), st, a, ntile); \
                VITK_CHECK_LAUNCH(); return 0; }
            switch (abl) { ABL_CASE(1) ABL_CASE(2) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(32) ABL_CASE(33) ABL_CASE(59) ABL_CASE(63) default: break; }
        }
#endif
        if (ntile > 4)
            hipLaunchKernelGGL(attn_bwd_fused_bf16_kernel<7>, dim3(B * H), dim3(448), attn_bwd_fused_lds(7), st, a, ntile);
        else
            hipLaunchKernelGGL(attn_bwd_fused_bf16_kernel<4>, dim3(B * H), dim3(256), attn_bwd_fused_lds(4), st, a, ntile);
        VITK_CHECK_LAUNCH();
        return 0;
    }
    const long units = (long)B * N * H * 8;
    hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(vitk_ceil_div(units, 256)), dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    const dim3 grid(vitk_ceil_div(N, 64), B * H);
    hipLaunchKernelGGL(attn_bwd_dkdv_bf16_kernel<false>, grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<false>, grid, dim3(256), 0, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}
