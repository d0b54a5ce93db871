// bf16 MFMA GEMMs for the linear layers of the ViT block (gfx950).
//
//   gemm_nt:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)      forward projections and dX
//   gemm_tn:  C[M,N] += A[T,M]^T . B[T,N]  (fp32 C)     weight gradients, split over T
//
// One kernel template per layout, instantiated for three workgroup shapes:
//   128x128 tile, 4 waves (2x2)  - small problems (2 workgroups per CU)
//   256x128 tile, 8 waves (4x2)  - N not a multiple of 256 tiles without a long tail
//   256x256 tile, 8 waves (2x4)  - large problems: 128 FLOP per staged byte keeps the L2->LDS
//                                  stream under half the L2 rate at full MFMA speed
// Every wave owns a (16*MI)x64 sub-tile computed with v_mfma_f32_16x16x32_bf16.  A 64-deep
// reduction step is staged by direct global->LDS loads (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write) into two LDS buffers; the next step's loads are in flight while the
// current one is multiplied.  An LDS-DMA wave instruction writes 1 KiB linearly (lane l ->
// base + 16 l), so the bank-conflict swizzle is applied to each lane's SOURCE address and to
// the fragment reads, never to the destination.
#include "common.h"

#include <type_traits>

#define VITK_GLB __attribute__((address_space(1)))

namespace {

constexpr int BK = 64;       // reduction depth of one staged step (TN kernel and default NT shapes)

template <int AUX = 0>      // cache-policy bits of the load: 1 = sc0, 2 = nt, 16 = sc1
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VITK_GLB void*)gsrc, (VITK_LDS void*)lds_wave_base, 16, 0, AUX);
}

// The same load issued from inline assembly: the compiler does not know of the pending LDS write.  It is needed
// where fragments are read with the transposing LDS loads: those are intrinsics without a memory operand the
// compiler could tell apart from the LDS-DMA destination, so behind a builtin LDS-DMA it drains the whole
// load queue (s_waitcnt vmcnt(0)) in front of every such read and the counted waits are void.  M0 (the LDS
// destination base of the DMA) belongs to the compiler and is restored.  `lds_dst` must be wave-uniform.
__device__ __forceinline__ void glds16_asm(const void* gsrc, const char* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(size_t)(VITK_LDS const char*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// The same with the destination as a 32-bit LDS address (made wave-uniform here).
__device__ __forceinline__ void glds16_asm_at(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// XCD-aware workgroup remap: blocks b and b+8 share an XCD (and its L2), so give every XCD a
// contiguous run of tile ids.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// Output column held by MFMA result row `rr` (0..15) of weight sub-tile j (0..3) of a wave:
// sub-tiles are paired so that one lane owns 8 consecutive columns across the pair.
__device__ __forceinline__ int ncol_of(int j, int rr) {
    return 32 * (j >> 1) + 8 * (rr >> 2) + (rr & 3) + 4 * (j & 1);
}

template <int BM_, int BN_, int WGM_, int WGN_, int STAGES_ = 2, int BK_ = 64, int WG_PER_CU_ = 0>
struct Shape {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, STAGES = STAGES_, KD = BK_;
    static constexpr int NW = WGM * WGN, NT = 64 * NW;
    static constexpr int MI = BM / WGM / 16;           // 16-row sub-tiles per wave
    static constexpr int ROWB = KD * 2;                // bytes of one LDS tile row (NT kernel)
    static constexpr int RPP = 1024 / ROWB;            // tile rows per 1 KiB staging piece
    static constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;   // one KD-deep step
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int LDS = STAGES * STAGE;
    static constexpr int WG_PER_CU = WG_PER_CU_ ? WG_PER_CU_ : (NW == 8 ? 1 : 2);
    static_assert(KD == 64 || KD == 32, "staged depth");
    static_assert(BN / WGN == 64, "a wave owns 64 output columns");
    static_assert((BM / RPP) % NW == 0 && (BN / RPP) % NW == 0, "whole 1 KiB pieces per wave");
};

// ============================================================================ NT kernel
// LDS tile image: [rows][64 bf16] = 128-byte rows; the 16-byte chunk c of row r sits at
// chunk position c ^ (r & 7).  Fragment reads (16 rows x one chunk per 16-lane group,
// ds_read_b128) are bank-conflict free; every global read is a whole 128-byte line.
struct NtArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int m0;                   // first row this launch covers (a multiple of the tile height)
    int tiles_m, tiles_n, nwg;
    GemmEpi epi;
};

// Chunk swizzle.  128-byte rows: position = chunk ^ (row & 7).  64-byte rows: position =
// chunk ^ ((-(row >> 2)) & 3); a ds_read_b128 lane group (e.g. lanes 0-3, 12-15, 20-27 = rows
// 0-3 and 12-15 of chunk g, rows 4-11 of chunk g+1) then covers each of the 16 bank quads once.
template <int ROWB>
__device__ __forceinline__ int nt_swz(int row) {
    if constexpr (ROWB == 128) return row & 7;
    else return (0 - (row >> 2)) & 3;
}
template <int ROWB>
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {
    return row * ROWB + ((chunk ^ nt_swz<ROWB>(row)) << 4);
}

// Shared C-write: lane owns row m, columns n0 + 32*jp + 8*g + 0..7 of each of its MI row groups.
template <int MI, int EPI, typename TC>
__device__ __forceinline__ void nt_epilogue(const NtArgs& p, f32x4 (&acc)[MI][4], int bm, int bn, int wm, int wn,
                                            int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    // ---- epilogue: lane owns row m, columns n0 + 32*jp + 8*g + 0..7.  Loads of the extra operand
    // (residual / pre-activation / position embedding) are issued for a whole group of fragments
    // before any is consumed; out-of-range fragments load from clamped addresses and skip the store.
    constexpr bool kBias = EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_RES ||
                           EPI == VITK_EPI_PATCH;
    using ExtT = typename EpiExt<EPI, TC, bf16_t>::type;
    constexpr bool kExt = EpiExt<EPI, TC, bf16_t>::kHas;
    constexpr int GROUP = !kExt ? 1 : (sizeof(ExtT) == 4 ? 4 : MI);   // fragments rows per load batch
    int ncol[2];
    float bias[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        ncol[jp] = bn + wn * 64 + 32 * jp + 8 * g;
        const int nc = ncol[jp] < p.N ? ncol[jp] : p.N - 8;
        if constexpr (kBias) load_vec<float, 8>(p.epi.bias + nc, bias[jp]);
    }
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += GROUP) {
        RawVec<ExtT, 8> ext[GROUP][2];
        if constexpr (kExt) {
#pragma unroll
            for (int ii = 0; ii < GROUP; ++ii) {
                int m = bm + wm * (16 * MI) + (i0 + ii) * 16 + l15;
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int nc = ncol[jp] < p.N ? ncol[jp] : p.N - 8;
                    epi_load_ext<EPI, TC, bf16_t, 8>(p.epi, m, nc, ext[ii][jp]);
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < GROUP; ++ii) {
            const int i = i0 + ii;
            const int m = bm + wm * (16 * MI) + i * 16 + l15;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                if (m < p.M && ncol[jp] < p.N) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[i][2 * jp][r];
                        v[4 + r] = acc[i][2 * jp + 1][r];
                    }
                    epi_apply_store<EPI, TC, bf16_t, 8>(p.epi, v, bias[jp], ext[ii][jp], m, ncol[jp],
                                                        reinterpret_cast<TC*>(p.C), p.ldc);
                }
            }
        }
    }
}

// C-write of the 256 x 256 kernels, staged through LDS.  In the accumulator layout a lane owns 16 bytes
// of one row, so one store instruction touches 16 rows x 64 bytes; a CU sustains only ~30 B/ns with that
// shape against ~100 B/ns when every instruction writes whole 128-byte lines (tools/store_probe.hip), and
// the epilogue is the part of a tile during which no MFMA runs.  Each wave therefore turns its 128 x 64
// sub-tile around in a 4 KiB LDS scratch of its own, 32 (bf16) or 16 (fp32) rows at a time: the values go
// in as the accumulators hold them (one row per lane, 16-byte chunk c of row r at chunk position
// c ^ (r & 7): conflict-free for the writes and for the reads below) and come back out with 8 (bf16) or 16
// (fp32) consecutive lanes per row, so a store instruction covers 8 rows x 128 B or 4 rows x 256 B.
// LDS operations of one wave execute in order: no barrier, no other wave touches the scratch.
// The extra operand (residual / pre-activation / position embedding) is still read in the accumulator
// layout, a whole group of fragments ahead of its use.
template <typename TC>
__device__ __forceinline__ void lds_put8(char* ws, int lrw, int chunk8, const float* v) {
    if constexpr (sizeof(TC) == 2) {
        typedef TC vec8 __attribute__((ext_vector_type(8)));      // bf16 or IEEE half
        vec8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (TC)v[i];
        *reinterpret_cast<vec8*>(ws + lrw * 128 + ((chunk8 ^ (lrw & 7)) << 4)) = o;
    } else {
        const int c = 2 * chunk8;
        *reinterpret_cast<f32x4*>(ws + lrw * 256 + ((c ^ (lrw & 7)) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(ws + lrw * 256 + (((c + 1) ^ (lrw & 7)) << 4)) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// Rows [mrow0, mrow0 + ROWS) of this wave's column block (first column col0) from the scratch to dst.
template <int EPI, typename T, int ROWS, bool STORE = true>
__device__ __forceinline__ void lds_flush_rows(const NtArgs& p, const char* ws, T* dst, long ld, int mrow0, int col0, int lane) {
    constexpr int ROWB = 64 * sizeof(T), LPR = ROWB / 16, RPI = 64 / LPR, EPC = 16 / sizeof(T);
    const int srow = lane / LPR, schunk = lane % LPR;
    const int n = col0 + schunk * EPC;
#pragma unroll
    for (int k = 0; k < ROWS / RPI; ++k) {
        const int lrw = k * RPI + srow;
        const u32x4 d = *reinterpret_cast<const u32x4*>(ws + lrw * ROWB + ((schunk ^ (lrw & 7)) << 4));
        const int m = mrow0 + lrw;
        if constexpr (STORE) {
            if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(dst + epi_out_row<EPI>(p.epi, m) * ld + n) = d;
        } else {
            asm volatile("" ::"v"(d), "v"(m));
        }
    }
}

// Row of the epilogue's extra operand that belongs to GEMM row m.
template <int EPI, typename ExtT>
__device__ __forceinline__ const ExtT* epi_ext_row(const GemmEpi& e, int m) {
    if constexpr (EPI == VITK_EPI_BIAS_RES) return reinterpret_cast<const ExtT*>(e.residual) + (long)m * e.ld_res;
    else if constexpr (EPI == VITK_EPI_DGELU) return reinterpret_cast<const ExtT*>(e.aux) + (long)m * e.ld_aux;
    else if constexpr (EPI == VITK_EPI_PATCH) {
        const int img = m / e.npatch, pp = m - img * e.npatch;
        return reinterpret_cast<const ExtT*>(e.pos) + (long)(pp + 1) * e.N;
    } else return nullptr;
}

// EABL (diagnostic build only; results are garbage, only the cycle stamps are read): 1 = no epilogue arithmetic,
// 2 = no global stores, 4 = no loads of the extra operand, 8 = no LDS turn-around and no stores.
template <int MI, int EPI, typename TC, int EABL = 0>
__device__ __forceinline__ void nt_epilogue_lds(const NtArgs& p, f32x4 (&acc)[MI][4], int bm, int bn, int wm, int wn,
                                                int lane, char* ws) {
    const int l15 = lane & 15, g = lane >> 4;
    constexpr bool kBias = EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_RES ||
                           EPI == VITK_EPI_PATCH;
    using ExtT = typename EpiExt<EPI, TC, bf16_t>::type;
    constexpr bool kExt = EpiExt<EPI, TC, bf16_t>::kHas;
    // The extra operand comes in the same way the result goes out (whole lines per load instruction, turned
    // around in the scratch) whenever its element size is that of C; else in the accumulator layout.
    constexpr bool kExtLds = kExt && sizeof(ExtT) == sizeof(TC);
    constexpr int FR = sizeof(TC) == 2 ? 2 : 1;                       // fragment rows (16 rows each) per 4 KiB pass
    constexpr int GROUP = !kExt ? FR : (sizeof(ExtT) == 4 ? 4 : MI);   // fragment rows whose extra operand is loaded ahead
    static_assert(GROUP % FR == 0 && MI % GROUP == 0, "passes tile the groups");
    constexpr int ROWB = 64 * sizeof(TC), LPR = ROWB / 16, RPI = 64 / LPR, EPC = 16 / sizeof(TC);
    constexpr int IPP = 16 * FR / RPI;                                 // load / store instructions per pass (4)
    const int row0 = bm + wm * (16 * MI), col0 = bn + wn * 64;
    const int srow = lane / LPR, schunk = lane % LPR;                  // this lane's place in the whole-line shape
    int ncol[2];
    float bias[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        ncol[jp] = col0 + 32 * jp + 8 * g;
        const int nc = ncol[jp] < p.N ? ncol[jp] : p.N - 8;
        if constexpr (kBias) load_vec<float, 8>(p.epi.bias + nc, bias[jp]);
    }
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += GROUP) {
        RawVec<ExtT, 8> ext[kExtLds ? 1 : GROUP][2];
        u32x4 raw[kExtLds ? (GROUP / FR) * IPP : 1];
        if constexpr (kExtLds) {
            int n = col0 + schunk * EPC;
            n = n < p.N ? n : p.N - EPC;
#pragma unroll
            for (int k = 0; k < (GROUP / FR) * IPP; ++k) {
                int m = row0 + i0 * 16 + k * RPI + srow;
                m = m < p.M ? m : p.M - 1;
                if constexpr (EABL & 4) { raw[k] = u32x4{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(raw[k])); }
                else raw[k] = *reinterpret_cast<const u32x4*>(epi_ext_row<EPI, ExtT>(p.epi, m) + n);
            }
        } else if constexpr (kExt) {
#pragma unroll
            for (int ii = 0; ii < GROUP; ++ii) {
                int m = row0 + (i0 + ii) * 16 + l15;
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int nc = ncol[jp] < p.N ? ncol[jp] : p.N - 8;
                    epi_load_ext<EPI, TC, bf16_t, 8>(p.epi, m, nc, ext[ii][jp]);
                }
            }
        }
#pragma unroll
        for (int f0 = 0; f0 < GROUP; f0 += FR) {
            if constexpr (kExtLds) {
#pragma unroll
                for (int k = 0; k < IPP; ++k) {
                    const int lrw = k * RPI + srow;
                    *reinterpret_cast<u32x4*>(ws + lrw * ROWB + ((schunk ^ (lrw & 7)) << 4)) = raw[(f0 / FR) * IPP + k];
                }
            }
            float pre[FR][2][8];
            float val[FR][2][8];
#pragma unroll
            for (int ii = 0; ii < FR; ++ii) {
                const int i = i0 + f0 + ii;
                int m = row0 + i * 16 + l15;
                m = m < p.M ? m : p.M - 1;                     // rows past the edge: computed, never stored
                const int lrw = ii * 16 + l15;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    RawVec<ExtT, 8> e8;
                    if constexpr (kExtLds) {
                        const int c8 = 4 * jp + g;
                        if constexpr (sizeof(TC) == 2) {
                            e8.v = *reinterpret_cast<const decltype(e8.v)*>(ws + lrw * 128 + ((c8 ^ (lrw & 7)) << 4));
                        } else {
                            e8.a = *reinterpret_cast<const f32x4*>(ws + lrw * 256 + (((2 * c8) ^ (lrw & 7)) << 4));
                            e8.b = *reinterpret_cast<const f32x4*>(ws + lrw * 256 + (((2 * c8 + 1) ^ (lrw & 7)) << 4));
                        }
                    } else if constexpr (kExt) {
                        e8 = ext[f0 + ii][jp];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        val[ii][jp][r] = acc[i][2 * jp][r];
                        val[ii][jp][4 + r] = acc[i][2 * jp + 1][r];
                    }
                    if constexpr (EABL & 1) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if constexpr (kExt) val[ii][jp][r] += e8.get(r);
                            pre[ii][jp][r] = val[ii][jp][r];
                        }
                    } else {
                        epi_apply<EPI, TC, bf16_t, 8>(p.epi, val[ii][jp], bias[jp], e8, m, ncol[jp], pre[ii][jp]);
                    }
                }
            }
            if constexpr (EABL & 8) {
#pragma unroll
                for (int ii = 0; ii < FR; ++ii)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                        for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(val[ii][jp][r]), "v"(pre[ii][jp][r]));
                continue;
            }
            // (the scratch still holds the extra operand until every lane has read its part: write C only now)
#pragma unroll
            for (int ii = 0; ii < FR; ++ii)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) lds_put8<TC>(ws, ii * 16 + l15, 4 * jp + g, val[ii][jp]);
            lds_flush_rows<EPI, TC, 16 * FR, !(EABL & 2)>(p, ws, reinterpret_cast<TC*>(p.C), p.ldc, row0 + (i0 + f0) * 16, col0, lane);
            if constexpr (EPI == VITK_EPI_BIAS_GELU) {
#pragma unroll
                for (int ii = 0; ii < FR; ++ii)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) lds_put8<f16_t>(ws, ii * 16 + l15, 4 * jp + g, pre[ii][jp]);
                lds_flush_rows<EPI, f16_t, 16 * FR, !(EABL & 2)>(p, ws, reinterpret_cast<f16_t*>(p.epi.aux), p.epi.ld_aux,
                                                                  row0 + (i0 + f0) * 16, col0, lane);
            }
        }
    }
}

template <typename S, int EPI, typename TC>
__global__ __launch_bounds__(S::NT, S::WG_PER_CU) void gemm_nt_bf16_kernel(NtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [stage][A | B]
    constexpr int PA = S::BM / S::RPP / S::NW, PB = S::BN / S::RPP / S::NW;   // 1 KiB pieces per wave
    constexpr int MI = S::MI, ROWB = S::ROWB, KD = S::KD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / S::WGN, wn = wid % S::WGN;
    const int l15 = lane & 15, g = lane >> 4;

    // Tile order: every XCD walks a contiguous run of ids; ids run down a band of GM tile rows
    // before moving one tile column over, so the workgroups resident on one XCD at a time cover a
    // compact block and share their operand panels in that XCD's L2.
    const int wg = xcd_remap(blockIdx.x, p.nwg);
    constexpr int GM = 8;
    const int band = wg / (GM * p.tiles_n), inb = wg - band * (GM * p.tiles_n);
    const int rows_in_band = (p.tiles_m - band * GM) < GM ? (p.tiles_m - band * GM) : GM;
    const int tn = inb / rows_in_band, tm = band * GM + (inb - tn * rows_in_band);
    const int bm = p.m0 + tm * S::BM, bn = tn * S::BN;

    // ---- staging map.  Piece (i*NW + wid) = LDS rows RPP*piece .. (1 KiB); lane l lands at
    // row RPP*piece + l / CPR, chunk position l % CPR, so it must FETCH chunk position ^ swizzle(row).
    const bf16_t* a_src[PA];
    const bf16_t* b_src[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = S::RPP * (i * S::NW + wid) + lane / S::CPR;
        const int ch = (lane % S::CPR) ^ nt_swz<ROWB>(row);
        int am = bm + row; am = am < p.M ? am : p.M - 1;
        a_src[i] = p.A + (long)am * p.lda + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = S::RPP * (i * S::NW + wid) + lane / S::CPR;
        const int ch = (lane % S::CPR) ^ nt_swz<ROWB>(row);
        // LDS row `row` of the weight tile holds the weight row that MFMA result row
        // (row & 15) of sub-tile (row >> 4) & 3 of wave-column row >> 6 must produce.
        int bnrow = bn + (row & ~63) + ncol_of((row >> 4) & 3, row & 15);
        bnrow = bnrow < p.N ? bnrow : p.N - 1;
        b_src[i] = p.B + (long)bnrow * p.ldb + ch * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * S::STAGE + wid * 1024;
        const int ko = kt * KD;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_src[i] + ko, dst + i * (S::NW * 1024));
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16(b_src[i] + ko, dst + S::A_BYTES + i * (S::NW * 1024));
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KD;
    stage(0, 0);
    if constexpr (S::STAGES == 3) {
        if (nk > 1) stage(1, 1);
    }
    for (int kt = 0; kt < nk; ++kt) {
        int cur;
        if constexpr (S::STAGES == 2) {
            // every wave's loads of step kt have landed, and nobody still reads the other buffer
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
            cur = kt & 1;
        } else {
            // three buffers: the loads of step kt+1 stay in flight across the barrier (counted
            // vmcnt + raw s_barrier: __syncthreads() would drain them), step kt+2 is issued into
            // the buffer every wave finished reading at step kt-1.
            if (kt + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(kt + 2, (kt + 2) % 3);
            cur = kt % 3;
        }
        const char* As = smem + cur * S::STAGE;
        const char* Bs = As + S::A_BYTES;
        if constexpr (KD == 64) {
            // The second 32-deep half's fragments are fetched from LDS two at a time between the
            // first half's MFMA groups, so only the first half's read latency is exposed.  (At most
            // 14 LDS reads are in flight: the lgkm counter is 4 bits wide.)
            bf16x8 xa[2][MI], wb[2][4];
            auto rd_w = [&](int s, int j) {
                wb[s][j] = *reinterpret_cast<const bf16x8*>(Bs + nt_lds_off<ROWB>(wn * 64 + j * 16 + l15, 4 * s + g));
            };
            auto rd_x = [&](int s, int i) {
                xa[s][i] = *reinterpret_cast<const bf16x8*>(As + nt_lds_off<ROWB>(wm * (16 * MI) + i * 16 + l15, 4 * s + g));
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) rd_w(0, j);
#pragma unroll
            for (int i = 0; i < MI; ++i) rd_x(0, i);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NRD = 4 + MI;                       // second-half reads to spread
            constexpr int PER = (NRD + MI - 2) / (MI - 1);    // reads issued after each MFMA group
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0][j], xa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int r = i * PER; r < (i + 1) * PER && r < NRD; ++r) {
                    if (r < 4) rd_w(1, r);
                    else rd_x(1, r - 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1][j], xa[1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // One 32-deep step: the activation fragments of the second half of the wave's rows are
            // read while the first half is multiplied; the co-resident workgroup's waves fill the
            // rest of the LDS latency.
            bf16x8 xa[MI], wb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wb[j] = *reinterpret_cast<const bf16x8*>(Bs + nt_lds_off<ROWB>(wn * 64 + j * 16 + l15, g));
#pragma unroll
            for (int i = 0; i < MI; ++i)
                xa[i] = *reinterpret_cast<const bf16x8*>(As + nt_lds_off<ROWB>(wm * (16 * MI) + i * 16 + l15, g));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    }

    nt_epilogue<MI, EPI, TC>(p, acc, bm, bn, wm, wn, lane);
}

// ============================================================================ NT kernel, ping-pong
// 256x256 tile, 8 waves (2x4), two 64-deep LDS buffers.  A K-tile is consumed in four phases, one
// 64x32 quadrant of the wave's 128x64 output (16 MFMAs) each; a phase is
//     LDS reads of the fragments it needs | 2 LDS-DMA pieces of a tile ahead | counted vmcnt
//     s_barrier | MFMA cluster | s_barrier
// and the two wave rows (wm = 0 / 1: one wave of each per SIMD) run one barrier apart, so while one
// wave of a SIMD multiplies, the other fetches.  Per K-tile t (buffer t & 1) the phases read
//     p0: A-half 0, B-half 0   p1: B-half 1   p2: A-half 1   p3: nothing (B-half 0 stays in registers)
// and stage
//     p0: B1 of t+1   p1: A1 of t+1   p2: A0 of t+2   p3: B0 of t+2
// (A-half h = rows 64h..64h+63 of every wave row, B-half h = sub-tiles 2h, 2h+1 of every wave
// column; 16 KiB = 16 pieces = 2 per wave).  Each region is restaged two phases after its last
// read and waited for (all but the 8 youngest loads, i.e. everything issued four or more phases
// ago) one phase and one barrier before its first read.
#ifdef VITK_GEMM_STAMP
static int g_gemm_abl = 0;       // which parts of the steady-state loop the next bare bf16 launches leave out
extern "C" void vitk_debug_gemm_abl(int mask) { g_gemm_abl = mask; }
// Diagnostic build only: shader-clock and 100 MHz wall stamps of wave 0 of every workgroup at kernel
// entry, after the prologue, after the main loop and after the epilogue, plus the hardware id of the CU
// (10 words per workgroup, read by no kernel).  In-kernel clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz.
constexpr int STAMP_WGS = 4096;
__device__ unsigned long long g_gemm_stamps[STAMP_WGS * 10];
#define VITK_STAMP(c, r)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    const unsigned long long c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0)
#else
#define VITK_STAMP(c, r)
#endif

// ABL (diagnostic build only): 1 = no LDS-DMA issue in the steady-state loop, 2 = no fragment reads
// there, 4 = no barriers there, 8 = no waits for landed LDS-DMA data there (results are garbage);
// 16 / 32 / 64 = LDS-DMA loads with the nt / sc0 / sc1 cache-policy bit (results valid); only the cycle stamps are read.
template <int EPI, typename TC, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_nt_pp_kernel(NtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 32 KiB | B 32 KiB] | 8 x 4 KiB epilogue scratch
    VITK_STAMP(sc0, sr0);
    constexpr int MI = 8, TILE = 256, ABYTES = 256 * 128, STAGE = 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l15 = lane & 15, g = lane >> 4;

    const int wg = xcd_remap(blockIdx.x, p.nwg);
    constexpr int GM = 8;
    const int band = wg / (GM * p.tiles_n), inb = wg - band * (GM * p.tiles_n);
    const int rows_in_band = (p.tiles_m - band * GM) < GM ? (p.tiles_m - band * GM) : GM;
    const int tn = inb / rows_in_band, tm = band * GM + (inb - tn * rows_in_band);
    const int bm = p.m0 + tm * TILE, bn = tn * TILE;

    // Staging: this wave's two pieces (8 tile rows each) of every half-tile.  LDS rows of piece
    // idx = 2*wid + q:  A-half h: 128*(idx>>3) + 64*h + 8*(idx&7);  B-half h: 64*(idx>>2) + 32*h + 8*(idx&3).
    const bf16_t* a_src[2][2];
    const bf16_t* b_src[2][2];
    int a_dst[2][2], b_dst[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = 2 * wid + q;
            const int arow0 = 128 * (idx >> 3) + 64 * h + 8 * (idx & 7);
            const int brow0 = 64 * (idx >> 2) + 32 * h + 8 * (idx & 3);
            const int ar = arow0 + (lane >> 3), br = brow0 + (lane >> 3);
            int am = bm + ar; am = am < p.M ? am : p.M - 1;
            a_src[h][q] = p.A + (long)am * p.lda + (((lane & 7) ^ (ar & 7)) << 3);
            int bnrow = bn + (br & ~63) + ncol_of((br >> 4) & 3, br & 15);
            bnrow = bnrow < p.N ? bnrow : p.N - 1;
            b_src[h][q] = p.B + (long)bnrow * p.ldb + (((lane & 7) ^ (br & 7)) << 3);
            a_dst[h][q] = arow0 * 128;
            b_dst[h][q] = ABYTES + brow0 * 128;
        }
    constexpr int AUX = ((ABL & 16) ? 2 : 0) | ((ABL & 32) ? 1 : 0) | ((ABL & 64) ? 16 : 0);
    auto stage_a = [&](int h, int t) {
        char* base = smem + (t & 1) * STAGE;
        glds16<AUX>(a_src[h][0] + t * 64, base + a_dst[h][0]);
        glds16<AUX>(a_src[h][1] + t * 64, base + a_dst[h][1]);
    };
    auto stage_b = [&](int h, int t) {
        char* base = smem + (t & 1) * STAGE;
        glds16<AUX>(b_src[h][0] + t * 64, base + b_dst[h][0]);
        glds16<AUX>(b_src[h][1] + t * 64, base + b_dst[h][1]);
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 xa[2][4] = {}, wb0[2][2] = {}, wb1[2][2] = {};
    auto read_a = [&](const char* As, int h) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xa[s][i] = *reinterpret_cast<const bf16x8*>(As + nt_lds_off<128>(wm * 128 + (4 * h + i) * 16 + l15, 4 * s + g));
    };
    auto read_b = [&](const char* Bs, int h, bf16x8 (&wb)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[s][j] = *reinterpret_cast<const bf16x8*>(Bs + nt_lds_off<128>(wn * 64 + (2 * h + j) * 16 + l15, 4 * s + g));
    };
    // barrier | MFMA cluster of quadrant (ha, hb) | barrier
    auto cluster = [&](int ha, int hb, const bf16x8 (&wb)[2][2]) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * ha + i][2 * hb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        wb[s][j], xa[s][i], acc[4 * ha + i][2 * hb + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#define VITK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    const int nk = p.K / 64;          // >= 3 (checked by the launcher)
    stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1);
    VITK_VMCNT(8);                    // A0, B0 of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // second wave row runs one barrier behind
    VITK_STAMP(sc1, sr1);

    // MODE 0: steady state (t + 2 < nk); 1: t == nk - 2; 2: t == nk - 1
    auto tile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool DMA = !(MODE == 0 && (ABL & 1)), RD = !(MODE == 0 && (ABL & 2)), WT = !(MODE == 0 && (ABL & 8));
        const char* As = smem + (t & 1) * STAGE;
        const char* Bs = As + ABYTES;
        // p0
        if constexpr (RD) { read_a(As, 0); read_b(Bs, 0, wb0); }
        if constexpr (MODE < 2) { if constexpr (DMA) stage_b(1, t + 1); if constexpr (WT) VITK_VMCNT(8); } else { VITK_VMCNT(2); }
        cluster(0, 0, wb0);
        // p1
        if constexpr (RD) read_b(Bs, 1, wb1);
        if constexpr (MODE < 2) { if constexpr (DMA) stage_a(1, t + 1); if constexpr (WT) VITK_VMCNT(8); } else { VITK_VMCNT(0); }
        cluster(0, 1, wb1);
        // p2
        if constexpr (RD) read_a(As, 1);
        if constexpr (MODE == 0) { if constexpr (DMA) stage_a(0, t + 2); if constexpr (WT) VITK_VMCNT(8); } else if constexpr (MODE == 1) { VITK_VMCNT(6); }
        cluster(1, 1, wb1);
        // p3
        if constexpr (MODE == 0) { if constexpr (DMA) stage_b(0, t + 2); if constexpr (WT) VITK_VMCNT(8); } else if constexpr (MODE == 1) { VITK_VMCNT(4); }
        cluster(1, 0, wb0);
    };
    int t = 0;
    for (; t + 2 < nk; ++t) tile(t, std::integral_constant<int, 0>{});
    tile(t, std::integral_constant<int, 1>{});
    tile(t + 1, std::integral_constant<int, 2>{});
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef VITK_VMCNT
    VITK_STAMP(sc2, sr2);

    nt_epilogue_lds<MI, EPI, TC, (ABL >> 8)>(p, acc, bm, bn, wm, wn, lane, smem + 2 * STAGE + wid * 4096);
#ifdef VITK_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores of the epilogue have left
    VITK_STAMP(sc3, sr3);
    if (tid == 0 && blockIdx.x < STAMP_WGS) {
        unsigned long long* o = g_gemm_stamps + blockIdx.x * 10;
        o[0] = sc0; o[1] = sc1; o[2] = sc2; o[3] = sc3; o[4] = sr0; o[5] = sr1; o[6] = sr2; o[7] = sr3;
        o[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: which CU ran this workgroup
        o[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    }
#endif
}


// ============================================================================ NT kernel, ping-pong, persistent
// The same tile loop as gemm_nt_pp_kernel, but a workgroup stays on its CU and walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... of the launch.  Between two one-tile workgroups a CU sits
// idle for 1.6-3.7 us (the old workgroup's stores drain, its LDS and registers are released, the new
// one is dispatched and waits a full memory latency for its first K-tile): 8-12 % of a 256 x 256 x 768
// tile.  Here the first six half-tile pieces of the NEXT tile are issued (LDS-DMA: no registers) as soon
// as the last K-tile of the current one has been read, i.e. before the epilogue, which then runs with
// the next tile's data already on its way.
// Source addresses are kept as 32-bit element offsets (the launcher checks that they fit).
template <int EPI, typename TC>
__global__ __launch_bounds__(512, 1) void gemm_nt_pps_kernel(NtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 32 KiB | B 32 KiB] | 8 x 4 KiB epilogue scratch
    constexpr int MI = 8, TILE = 256, ABYTES = 256 * 128, STAGE = 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l15 = lane & 15, g = lane >> 4;
    // staging: lane l of a piece lands at tile row (l >> 3) of the piece's 8 rows, chunk position l & 7, so it
    // fetches source chunk (l & 7) ^ (row & 7) with (row & 7) == (l >> 3) for every piece

    // Staging pieces of this wave (as in gemm_nt_pp_kernel): piece idx = 2*wid + q of half-tile h covers
    // A rows 128*(idx>>3) + 64*h + 8*(idx&7) + 0..7 and B rows 64*(idx>>2) + 32*h + 8*(idx&3) + 0..7.
    int a_dst[2][2], b_dst[2][2];                      // wave-uniform LDS offsets
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = 2 * wid + q;
            a_dst[h][q] = (128 * (idx >> 3) + 64 * h + 8 * (idx & 7)) * 128;
            b_dst[h][q] = ABYTES + (64 * (idx >> 2) + 32 * h + 8 * (idx & 3)) * 128;
        }

    int a_off[2][2], b_off[2][2];                      // element offsets of this lane's source rows (current or next tile)
    int bm = 0, bn = 0;
    // Everything per-lane is derived afresh from the lane id on every call (the copy below is opaque to the
    // compiler), so that nothing but the eight offsets stays in registers between two calls: the epilogue
    // needs every register the main loop's fragments leave free.
    auto setup = [&](int t) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lr_ = ln >> 3, colsw_ = ((ln & 7) ^ lr_) << 3;
        const int wg = xcd_remap(t, p.nwg);
        constexpr int GM = 8;
        const int band = wg / (GM * p.tiles_n), inb = wg - band * (GM * p.tiles_n);
        const int rows_in_band = (p.tiles_m - band * GM) < GM ? (p.tiles_m - band * GM) : GM;
        const int tn = inb / rows_in_band, tm = band * GM + (inb - tn * rows_in_band);
        bm = p.m0 + tm * TILE;
        bn = tn * TILE;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = 2 * wid + q;
                int am = bm + 128 * (idx >> 3) + 64 * h + 8 * (idx & 7) + lr_; am = am < p.M ? am : p.M - 1;
                const int br = 64 * (idx >> 2) + 32 * h + 8 * (idx & 3) + lr_;
                int bnrow = bn + (br & ~63) + ncol_of((br >> 4) & 3, br & 15); bnrow = bnrow < p.N ? bnrow : p.N - 1;
                a_off[h][q] = am * (int)p.lda + colsw_;
                b_off[h][q] = bnrow * (int)p.ldb + colsw_;
            }
    };
    auto stage_a = [&](int h, int t) {
        char* base = smem + (t & 1) * STAGE;
        glds16(p.A + a_off[h][0] + t * 64, base + a_dst[h][0]);
        glds16(p.A + a_off[h][1] + t * 64, base + a_dst[h][1]);
    };
    auto stage_b = [&](int h, int t) {
        char* base = smem + (t & 1) * STAGE;
        glds16(p.B + b_off[h][0] + t * 64, base + b_dst[h][0]);
        glds16(p.B + b_off[h][1] + t * 64, base + b_dst[h][1]);
    };
    auto prefetch = [&]() {                            // the first six half-tile pieces of a tile, in the order the loop waits for them
        stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1);
    };

    f32x4 acc[MI][4];
    bf16x8 xa[2][4] = {}, wb0[2][2] = {}, wb1[2][2] = {};
    auto read_a = [&](const char* As, int h) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xa[s][i] = *reinterpret_cast<const bf16x8*>(As + nt_lds_off<128>(wm * 128 + (4 * h + i) * 16 + l15, 4 * s + g));
    };
    auto read_b = [&](const char* Bs, int h, bf16x8 (&wb)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[s][j] = *reinterpret_cast<const bf16x8*>(Bs + nt_lds_off<128>(wn * 64 + (2 * h + j) * 16 + l15, 4 * s + g));
    };
    auto cluster = [&](int ha, int hb, const bf16x8 (&wb)[2][2]) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * ha + i][2 * hb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        wb[s][j], xa[s][i], acc[4 * ha + i][2 * hb + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#define VITK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    // MODE 0: steady state (t + 2 < nk); 1: t == nk - 2; 2: t == nk - 1  (see gemm_nt_pp_kernel);
    // 3: K-tile 0 of a tile that follows another one in this workgroup.  Vector-memory operations complete in
    // issue order, and the previous tile's last 16 stores (one 4 KiB pass = 4 stores, 4 passes after the
    // epilogue's last wait in every instance) were issued after the six prefetched pieces K-tile 0 waits for
    // and before the pieces it issues itself: they may stay in flight through all four of its waits, so the
    // write-out drains under this K-tile's MFMAs instead of in front of them.
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char* As = smem + (t & 1) * STAGE;
        const char* Bs = As + ABYTES;
        read_a(As, 0); read_b(Bs, 0, wb0);
        if constexpr (MODE == 3) { stage_b(1, t + 1); VITK_VMCNT(24); } else
        if constexpr (MODE < 2) { stage_b(1, t + 1); VITK_VMCNT(8); } else { VITK_VMCNT(2); }
        cluster(0, 0, wb0);
        read_b(Bs, 1, wb1);
        if constexpr (MODE == 3) { stage_a(1, t + 1); VITK_VMCNT(24); } else
        if constexpr (MODE < 2) { stage_a(1, t + 1); VITK_VMCNT(8); } else { VITK_VMCNT(0); }
        cluster(0, 1, wb1);
        read_a(As, 1);
        if constexpr (MODE == 3) { stage_a(0, t + 2); VITK_VMCNT(24); } else
        if constexpr (MODE == 0) { stage_a(0, t + 2); VITK_VMCNT(8); } else if constexpr (MODE == 1) { VITK_VMCNT(6); }
        cluster(1, 1, wb1);
        if constexpr (MODE == 3) { stage_b(0, t + 2); VITK_VMCNT(24); } else
        if constexpr (MODE == 0) { stage_b(0, t + 2); VITK_VMCNT(8); } else if constexpr (MODE == 1) { VITK_VMCNT(4); }
        cluster(1, 0, wb0);
    };

    const int nk = p.K / 64;          // >= 3 (checked by the launcher)
    int tile = blockIdx.x;            // < p.nwg: the launcher never starts more workgroups than tiles
    bool prev_partial = false;        // wave-uniform: the tile just written out crossed the edge of C
    setup(tile);
    prefetch();
    // One output tile; FIRST: no epilogue of this workgroup precedes it.  Returns whether another tile follows.
    auto run_tile = [&](auto first_tag) -> bool {
        constexpr bool FIRST = decltype(first_tag)::value;
        {   // this tile's offsets again (the epilogue of the previous tile ran without them)
            int tcur = tile;
            asm volatile("" : "+s"(tcur));
            setup(tcur);
        }
        const int cbm = __builtin_amdgcn_readfirstlane(bm), cbn = __builtin_amdgcn_readfirstlane(bn);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A0, B0 of K-tile 0 have landed: they are older than all but the 8 youngest loads of this wave (the other
        // prefetched pieces) and, from the second tile on, the previous epilogue's last 16 stores
        if constexpr (FIRST) VITK_VMCNT(8); else VITK_VMCNT(24);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();      // second wave row runs one barrier behind
        ktile(0, std::integral_constant<int, FIRST ? 0 : 3>{});
        int t = 1;
        for (; t + 2 < nk; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(t, std::integral_constant<int, 1>{});
        ktile(t + 1, std::integral_constant<int, 2>{});
        if (wm == 0) __builtin_amdgcn_s_barrier();      // every wave has read its last fragments: both buffers are free
        tile += gridDim.x;
        const bool more = tile < p.nwg;
        if (more) {
            int tnext = tile;
            asm volatile("" : "+s"(tnext));
            setup(tnext);
            prefetch();
        }
        nt_epilogue_lds<MI, EPI, TC>(p, acc, cbm, cbn, wm, wn, lane, smem + 2 * STAGE + wid * 4096);
        // A tile that sticks out over the edge of C: its stores are predicated per lane, and a wave whose whole
        // store group lies outside skips the instruction, so fewer than 16 stores may follow the prefetch.
        prev_partial = cbm + TILE > p.M || cbn + TILE > p.N;
        return more;
    };
    // The vmcnt(24) waits of a following tile count on the previous epilogue's 16 stores; after a partial
    // tile they may not all have been issued, so the next tile then waits like a first one (vmcnt(8): at
    // most the eight youngest operations pending, whatever they are).
    bool more = run_tile(std::true_type{});
    while (more) more = prev_partial ? run_tile(std::true_type{}) : run_tile(std::false_type{});
#undef VITK_VMCNT
}

// ============================================================================ TN kernel
// A:[T,M], B:[T,N] both row-major (the reduction index T is the slow one).  LDS tile image:
// [64 t][BM or BN cols] bf16; chunk c of row r sits at chunk position c ^ ((r & 7) << 1).
// MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose).  Within one MFMA the 32
// reduction slots are (lane group gq, element e) <-> t = 4*gq + (e & 3) + 16*(e >> 2),
// identically for both operands; with the swizzle above each 32-lane half reads 8 consecutive
// rows x four 8-byte slots that cover all 64 banks once.
struct TnArgs {
    const bf16_t* A; const bf16_t* B; float* C;
    float* a_colsum;          // optional: a_colsum[m] += sum_t A[t][m] (bias gradient)
    int M, N, T;
    long lda, ldb, ldc;
    int tiles_n, ntiles, kt_per_split;
};

template <int ROWB>   // bytes per LDS tile row
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int s, int col0, int lane) {
    // Fragment for MFMA rows/cols col0..col0+15 over t-rows 32s .. 32s+31 of the tile.
    const int l15 = lane & 15, gq = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const int row = 32 * s + 4 * gq + q;                       // (row + 16) & 7 == row & 7
    const int chunk = ((col0 >> 3) + (pp >> 1)) ^ ((row & 7) << 1);
    const char* a0 = tile + row * ROWB + chunk * 16 + (pp & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0 + 16 * ROWB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// One of the two transposed reads of tr_frag (half 0: reduction slots 0..15 of the slab rows, half 1: 16..31).
template <int ROWB>
__device__ __forceinline__ bf16x4 tr_half(const char* tile, int col0, int lane, int hi) {
    const int l15 = lane & 15, gq = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const int row = 4 * gq + q;
    const int chunk = ((col0 >> 3) + (pp >> 1)) ^ ((row & 7) << 1);
    const char* a0 = tile + row * ROWB + chunk * 16 + (pp & 1) * 8 + hi * 16 * ROWB;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((VITK_LDS bf16x4*)(a0));
}

template <typename S>
__global__ __launch_bounds__(S::NT, 2) void gemm_tn_bf16_kernel(TnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = S::MI;
    constexpr int RBA = S::BM * 2, RBB = S::BN * 2;            // LDS row bytes
    constexpr int PA = 64 * RBA / 1024 / S::NW, PB = 64 * RBB / 1024 / S::NW;   // pieces per wave
    constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;          // tile rows per piece
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / S::WGN, wn = wid % S::WGN;
    const int l15 = lane & 15, g = lane >> 4;

    // Split-major order with a contiguous run per XCD: the workgroups of one reduction slice land
    // on one or two XCDs, whose L2 then serves each operand panel of that slice to all its tiles.
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / p.ntiles, tile = wg - split * p.ntiles;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int bm = tm * S::BM, bn = tn * S::BN;
    const int kt0 = split * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    const int nkt = (p.T + BK - 1) / BK;
    kt1 = kt1 < nkt ? kt1 : nkt;
    if (kt0 >= kt1) return;

    // staging: piece (i*NW + wid) = RP tile rows (1 KiB); lane l lands at row
    // RP*piece + l / (lanes per row), chunk position l % (lanes per row), and fetches chunk
    // position ^ ((row & 7) << 1).  Each piece keeps a running source pointer that advances by
    // 64 rows per step; only the ragged last step re-derives clamped row addresses.
    int arow[PA], brow[PB];
    int a_col[PA], b_col[PB];
    const bf16_t* a_ptr[PA];
    const bf16_t* b_ptr[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        constexpr int LPR = RBA / 16;
        arow[i] = RPA * (i * S::NW + wid) + lane / LPR;
        const int ch = (lane % LPR) ^ ((arow[i] & 7) << 1);
        int ac = bm + ch * 8;
        ac = ac + 8 <= p.M ? ac : p.M - 8;       // columns past the edge: in-bounds garbage, never stored
        a_col[i] = ac;
        long t = (long)kt0 * BK + arow[i];
        t = t < p.T ? t : p.T - 1;
        a_ptr[i] = p.A + t * p.lda + ac;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        constexpr int LPR = RBB / 16;
        brow[i] = RPB * (i * S::NW + wid) + lane / LPR;
        const int ch = (lane % LPR) ^ ((brow[i] & 7) << 1);
        int bc = bn + ch * 8;
        bc = bc + 8 <= p.N ? bc : p.N - 8;
        b_col[i] = bc;
        long t = (long)kt0 * BK + brow[i];
        t = t < p.T ? t : p.T - 1;
        b_ptr[i] = p.B + t * p.ldb + bc;
    }
    const long a_step = (long)BK * p.lda, b_step = (long)BK * p.ldb;

    // Issues step kt (must be called for kt0, kt0+1, ... in order).  The LDS-DMA goes out from inline assembly:
    // behind the builtin the compiler drains the load queue in front of the transposed fragment reads that
    // follow, i.e. the step just requested would be waited for before the current one is multiplied.
    const unsigned lds0 = (unsigned)(size_t)(VITK_LDS const char*)smem;
    auto stage = [&](int kt, int buf) {
        const unsigned dst = lds0 + buf * S::STAGE + wid * 1024;
        if ((kt + 1) * BK <= p.T) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                glds16_asm_at(a_ptr[i], dst + i * (S::NW * 1024));
                a_ptr[i] += a_step;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                glds16_asm_at(b_ptr[i], dst + S::A_BYTES + i * (S::NW * 1024));
                b_ptr[i] += b_step;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                long t = (long)kt * BK + arow[i];
                t = t < p.T ? t : p.T - 1;           // rows past T are zeroed in LDS below
                glds16_asm_at(p.A + t * p.lda + a_col[i], dst + i * (S::NW * 1024));
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                long t = (long)kt * BK + brow[i];
                t = t < p.T ? t : p.T - 1;
                glds16_asm_at(p.B + t * p.ldb + b_col[i], dst + S::A_BYTES + i * (S::NW * 1024));
            }
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Column sums of A ride along as one extra MFMA against an all-ones fragment.  Row group i of
    // the A tile is summed by exactly one wave of one workgroup per reduction slice: wave column
    // i % WGN of tile column (i / WGN) % tiles_n.
    constexpr int NCS = (MI + S::WGN - 1) / S::WGN;
    f32x4 accb[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool cs_on = p.a_colsum != nullptr;
    const bf16_t one = (bf16_t)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
    // The ownership bits go through readfirstlane so that the tests below are scalar branches:
    // an MFMA executes even with EXEC cleared, so it must never sit under lane predication.
    int cs_sel = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
        if (cs_on && (i % S::WGN) == wn && ((i / S::WGN) % p.tiles_n) == tn) cs_sel |= 1 << i;
    const int cs_bits = __builtin_amdgcn_readfirstlane(cs_sel);
    auto cs_mine = [&](int i) { return (cs_bits >> i) & 1; };

    stage(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this step's pieces (the only loads in flight) have landed
        __syncthreads();
        char* As = smem + cur * S::STAGE;
        char* Bs = As + S::A_BYTES;
        const int valid = p.T - kt * BK;         // rows of this step that exist
        if (valid < BK) {
            // ragged last step: zero the rows past T (both operands), then re-synchronise
            for (int e = tid; e < (BK - valid) * (RBA / 16); e += S::NT)
                *reinterpret_cast<u32x4*>(As + valid * RBA + e * 16) = u32x4{0u, 0u, 0u, 0u};
            for (int e = tid; e < (BK - valid) * (RBB / 16); e += S::NT)
                *reinterpret_cast<u32x4*>(Bs + valid * RBB + e * 16) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
        if (kt + 1 < kt1) stage(kt + 1, cur ^ 1);
        // Software pipeline over the 2*MI row groups of the step (group n: half s = n / MI, rows
        // i = n % MI): the A fragment of group n+2 and, early in the first half, the B fragments
        // of the second half are requested right after the MFMAs of group n, so LDS latency hides
        // under the matrix pipe and at most ~14 reads are outstanding (4-bit lgkm counter).
        bf16x8 fa[2][MI], fb[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[0][j] = tr_frag<RBB>(Bs, 0, wn * 64 + j * 16, lane);
        fa[0][0] = tr_frag<RBA>(As, 0, wm * (16 * MI), lane);
        fa[0][1] = tr_frag<RBA>(As, 0, wm * (16 * MI) + 16, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 2 * MI; ++n) {
            const int sh = n / MI, i = n % MI;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sh][i], fb[sh][j], acc[i][j], 0, 0, 0);
            if (cs_mine(i))
                accb[i / S::WGN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sh][i], ones, accb[i / S::WGN], 0, 0, 0);
            if (n + 2 < 2 * MI) {
                const int s2 = (n + 2) / MI, i2 = (n + 2) % MI;
                fa[s2][i2] = tr_frag<RBA>(As, s2, wm * (16 * MI) + i2 * 16, lane);
            }
            if (n < 4) fb[1][n] = tr_frag<RBB>(Bs, 1, wn * 64 + n * 16, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // column sums: every result column holds the same sums; lanes of column 0 publish rows 4g+reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
        if (cs_mine(i) && l15 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm + wm * (16 * MI) + i * 16 + 4 * g + r;
                if (m < p.M) atomicAdd(p.a_colsum + m, accb[i / S::WGN][r]);
            }
        }

    // D[r][c]: r = 4*g + reg <-> A column (output row), c = l15 <-> B column (output col).
    const int m0 = bm + wm * (16 * MI) + 4 * g, n0 = bn + wn * 64 + l15;
    float* crow = p.C + (long)m0 * p.ldc + n0;
    if (bm + S::BM <= p.M && bn + S::BN <= p.N) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* c = crow + (long)(i * 16 + r) * p.ldc;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(c + j * 16, acc[i][j][r]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* c = crow + (long)(i * 16 + r) * p.ldc;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + i * 16 + r < p.M && n0 + j * 16 < p.N) atomicAdd(c + j * 16, acc[i][j][r]);
            }
    }
}

// ============================================================================ TN kernel, ping-pong
// 256x256 tile, 8 waves (2x4).  The reduction axis is consumed in 32-row slabs ([32 t][256 cols]
// of A and of B, 32 KiB) held in a ring of five LDS buffers; a phase is
//     4 LDS-DMA pieces of slab +4 | 24 transposed LDS reads (all fragments of the slab) | counted vmcnt (2nd wave row) | lgkmcnt(0)
//     s_barrier | 32 MFMAs (+ the column-sum MFMAs) | counted vmcnt (1st wave row) | s_barrier
// with the two wave rows one barrier apart (one multiplies while the other fetches).  Slab f + 5
// reuses the buffer of slab f and is issued at phase f + 1, the phase after its last read (three slabs
// = 96 KiB in flight per CU: the operands stream from HBM and two slabs did not cover its latency); slab
// f + 1 is waited for at phase f (all but the 12 youngest loads) and first read one barrier later.
// Requires T % 32 == 0 and at least five slabs per reduction slice (the launcher checks).
struct TnPpArgs {
    const bf16_t* A; const bf16_t* B; float* C;
    float* a_colsum;
    int M, N;
    long lda, ldb, ldc;
    int tiles_n, ntiles, slabs, slabs_per_split;
    int no_flush;             // diagnostic: skip the atomic write-out (results are wrong, only the time is read)
    float* slab;              // two-pass write-out: [split][tile][wave][i][j][lane][4] partial tiles, or nullptr (atomics)
};

// ABL (diagnostic build only; results are garbage, only the time is read): 1 = no LDS-DMA issue in the steady-state
// loop, 2 = no fragment reads there, 4 = no barriers there.
template <int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_tn_pp_kernel(TnPpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [5][A slab 16 KiB | B slab 16 KiB]
    constexpr int MI = 8, RB = 512, SLAB = 32 * RB, STAGE = 2 * SLAB, RING = 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l15 = lane & 15, g = lane >> 4;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / p.ntiles, tile = wg - split * p.ntiles;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int bm = tm * 256, bn = tn * 256;
    const int s0 = split * p.slabs_per_split;
    int s1 = s0 + p.slabs_per_split;
    s1 = s1 < p.slabs ? s1 : p.slabs;
    const int ns = s1 - s0;                                   // >= RING by construction

    // staging: pieces q*8 + wid (q = 0, 1) of each operand slab = tile rows 2*piece, 2*piece + 1;
    // lane l lands at row 2*piece + (l >> 5), chunk position l & 31, fetches chunk pos ^ ((row & 7) << 1)
    const bf16_t* a_ptr[2];
    const bf16_t* b_ptr[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = 2 * (q * 8 + wid) + (lane >> 5);
        const int ch = (lane & 31) ^ ((row & 7) << 1);
        int ac = bm + ch * 8; ac = ac + 8 <= p.M ? ac : p.M - 8;      // past the edge: in-bounds garbage, never stored
        int bc = bn + ch * 8; bc = bc + 8 <= p.N ? bc : p.N - 8;
        const long t = (long)s0 * 32 + row;
        a_ptr[q] = p.A + t * p.lda + ac;
        b_ptr[q] = p.B + t * p.ldb + bc;
    }
    const long a_step = 32 * p.lda, b_step = 32 * p.ldb;
    auto stage = [&](int buf) {                               // slabs are issued strictly in order
        char* dst = smem + buf * STAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_asm(a_ptr[q], dst + q * 8192);
            a_ptr[q] += a_step;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_asm(b_ptr[q], dst + SLAB + q * 8192);
            b_ptr[q] += b_step;
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of A: row group i belongs to wave column i & 3 of tile column (i >> 2) % tiles_n
    f32x4 accb[2];
    accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cs_sel = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
        if (p.a_colsum != nullptr && (i & 3) == wn && ((i >> 2) % p.tiles_n) == tn) cs_sel |= 1 << i;
    const int cs_bits = __builtin_amdgcn_readfirstlane(cs_sel);   // scalar: MFMA must not sit under EXEC predication
    const bf16_t one = (bf16_t)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};

#define VITK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    stage(0); stage(1); stage(2); stage(3);
    VITK_VMCNT(12);                                           // slab 0 has landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                // second wave row runs one barrier behind

    int rd = 0, wr = 4;                                       // ring positions of slab f and slab f + 4
    for (int f = 0; f < ns; ++f) {
        const char* As = smem + rd * STAGE;
        const char* Bs = As + SLAB;
        bf16x8 fa[MI], fb[4];
        // The pieces of slab f + 4 go out before this phase's fragment reads, into the buffer slab f - 1 was read
        // from: every wave has waited for its reads of that slab in front of a barrier it has since passed (the
        // lgkmcnt wait below sits before the barrier for that reason).  Slab f + 1 is first read one barrier
        // interval after this phase's MFMA cluster by the first wave row and two after it by the second: the
        // second row waits for its pieces in front of the barrier that closes its read interval, the first row
        // only behind its cluster (one interval more for its pieces to land).
        if constexpr (!(ABL & 1)) { if (f + 4 < ns) stage(wr); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL & 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { fb[j] = ones; asm volatile("" : "+v"(fb[j])); }
#pragma unroll
            for (int i = 0; i < MI; ++i) { fa[i] = ones; asm volatile("" : "+v"(fa[i])); }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = tr_frag<RB>(Bs, 0, wn * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = tr_frag<RB>(As, 0, wm * 128 + i * 16, lane);
        }
        auto wait_next = [&]() {
            if constexpr (ABL & 1) return;
            if (f + 4 < ns) {
                VITK_VMCNT(12);
            } else if (f + 3 < ns) {
                VITK_VMCNT(8);
            } else if (f + 2 < ns) {
                VITK_VMCNT(4);
            } else {
                VITK_VMCNT(0);
            }
        };
        if (wm == 1) wait_next();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if ((cs_bits >> i) & 1)
                accb[i >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i >> 2], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 0) wait_next();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd = rd + 1 == RING ? 0 : rd + 1;
        wr = wr + 1 == RING ? 0 : wr + 1;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef VITK_VMCNT

#pragma unroll
    for (int i = 0; i < MI; ++i)
        if (((cs_bits >> i) & 1) && l15 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm + wm * 128 + i * 16 + 4 * g + r;
                if (m < p.M) atomicAdd(p.a_colsum + m, accb[i >> 2][r]);
            }
        }
    const int m0 = bm + wm * 128 + 4 * g, n0 = bn + wn * 64 + l15;
    float* crow = p.C + (long)m0 * p.ldc + n0;
    if (p.no_flush) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 1.2345e-30f) crow[0] = keep;       // keeps the accumulators alive
        return;
    }
    if (p.slab != nullptr) {
        // the partial tile in the accumulators' own order: 1 KiB contiguous per store instruction
        f32x4* dst = reinterpret_cast<f32x4*>(p.slab) + ((long)(split * p.ntiles + tile) * 8 + wid) * (MI * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(i * 4 + j) * 64] = acc[i][j];
        return;
    }
    if (bm + 256 <= p.M && bn + 256 <= p.N) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* c = crow + (long)(i * 16 + r) * p.ldc;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(c + j * 16, acc[i][j][r]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* c = crow + (long)(i * 16 + r) * p.ldc;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + i * 16 + r < p.M && n0 + j * 16 < p.N) atomicAdd(c + j * 16, acc[i][j][r]);
            }
    }
}

// ============================================================================ TN kernel, one wave per SIMD
// Same 256x256 tile, slab ring and write-out as gemm_tn_pp_kernel, but four waves (2x2) of 128x128 each: a wave
// holds 64 accumulator tiles (256 accumulation registers: a SIMD that runs one wave gives it 512 registers)
// and two sets of sixteen fragments.  Per 32-row slab a wave issues 32 transposed LDS reads for 64 MFMAs (the
// eight-wave form: 24 for 32), i.e. the workgroup reads two thirds of the LDS bytes per FLOP.  There is no
// partner wave to multiply while this one fetches: the fragments of slab f + 1 are requested between the MFMAs
// of slab f (into the other register set), the LDS-DMA pieces of slab f + 4 likewise, and one barrier per slab
// publishes slab f + 2.  Every slab runs the same code: past the end of the reduction slice zeros are staged
// (into ring buffers whose slabs have been consumed), so the counted waits never change and a slab too many
// adds nothing.  The column sums of A (bias gradient) accumulate in ordinary registers (the 256 accumulators
// of the product take every accumulation register there is); forming them on the VALU instead (32 v_dot2c per
// slab) cost 0.05-0.07 ms per call: whatever a lone wave issues between its MFMAs is not free.
__device__ __attribute__((aligned(16))) const unsigned g_tn_zero16[4] = {0u, 0u, 0u, 0u};
template <bool CS, int NO = 8, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(TnPpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [5][A slab 16 KiB | B slab 16 KiB]
    constexpr int RB = 512, SLAB = 32 * RB, STAGE = 2 * SLAB, RING = 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / p.ntiles, tile = wg - split * p.ntiles;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int bm = tm * 256, bn = tn * 256;
    const int s0 = split * p.slabs_per_split;
    int s1 = s0 + p.slabs_per_split;
    s1 = s1 < p.slabs ? s1 : p.slabs;
    const int ns = s1 - s0;                                   // >= RING by construction

    // staging: pieces q*4 + wid (q = 0..3) of each operand slab = tile rows 2*piece, 2*piece + 1;
    // lane l lands at row 2*piece + (l >> 5), chunk position l & 31, fetches chunk pos ^ ((row & 7) << 1)
    // The source of a piece is a scalar base (the slab's first row, advanced once per slab) plus a 32-bit byte
    // offset per lane that never changes: nothing per-lane has to be updated between the MFMAs.
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 2 * (q * 4 + wid) + (lane >> 5);
        const int ch = (lane & 31) ^ ((row & 7) << 1);
        int ac = bm + ch * 8; ac = ac + 8 <= p.M ? ac : p.M - 8;      // past the edge: in-bounds garbage, never stored
        int bc = bn + ch * 8; bc = bc + 8 <= p.N ? bc : p.N - 8;
        a_off[q] = (unsigned)((row * p.lda + ac) * 2);               // < 32 rows: fits 32 bits (the launcher checks the strides)
        b_off[q] = (unsigned)((row * p.ldb + bc) * 2);
    }
    const bf16_t* a_base = p.A + (long)s0 * 32 * p.lda;
    const bf16_t* b_base = p.B + (long)s0 * 32 * p.ldb;
    int staged = 0;                                           // slabs issued so far
    long a_step = 32 * p.lda, b_step = 32 * p.ldb;
    // piece k of the next slab into ring buffer `buf`: k = 0..3 operand A, 4..7 operand B
    // (M0 is not restored behind the load: nothing the compiler generates for this kernel reads it, and the
    // restore had to wait until the load had left the issue stage)
    const unsigned lds0 = (unsigned)(size_t)(VITK_LDS const char*)smem;
    auto dma = [&](const void* sbase, unsigned voff, unsigned dst) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    auto stage_piece = [&](int buf, int k) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wid * 1024);
        if (k < 4) dma(a_base, a_off[k], dst + k * 4096);
        else dma(b_base, b_off[k - 4], dst + SLAB + (k - 4) * 4096);
    };
    // Past the end of the reduction slice every lane fetches the same sixteen zero bytes: the slabs staged from
    // then on are zeros, so the one the second half of the last loop trip multiplies when the slice has an odd
    // number of slabs adds nothing (every slab runs the same code, see below).
    auto slab_issued = [&]() __attribute__((always_inline)) {
        ++staged;
        a_base += a_step; b_base += b_step;
        if (staged >= ns) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_off[q] = 0u; b_off[q] = 0u; }
            a_base = b_base = reinterpret_cast<const bf16_t*>(g_tn_zero16);
            a_step = 0; b_step = 0;
        }
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Column sums of A (bias gradient).  The eight row groups of a wave row are dealt out to NO owner waves
    // (wave column wn of tile column tn < NO / 2 is owner tn * 2 + wn; row group k goes to owner k % NO), every
    // split adds its part.  Branches around an MFMA break up the loop (the compiler duplicates its tail per path),
    // so every wave issues the same NCS extra MFMAs per slab, on fragments 0, NO, 2 NO, ..., against ones (owned)
    // or zeros; for the owned row groups to be those, owner o holds row group (i + o) & 7 in fragment /
    // accumulator row i.
    constexpr int NCS = (8 + NO - 1) / NO;
    const int owner = (CS && tn < NO / 2) ? tn * 2 + wn : -1;
    const int rot = owner >= 0 ? owner : 0;
    f32x4 accb[NCS];
    bf16x8 wsel[NCS];
#pragma unroll
    for (int k = 0; k < NCS; ++k) {
        accb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t w = (bf16_t)((owner >= 0 && owner + k * NO < 8) ? 1.0f : 0.0f);
        wsel[k] = bf16x8{w, w, w, w, w, w, w, w};
        asm volatile("" : "+v"(wsel[k]));                 // in registers from here on: nothing writes them in front of an MFMA
    }
    auto arow = [&](int i) __attribute__((always_inline)) { return wm * 128 + ((i + rot) & 7) * 16; };   // first tile row of row group slot i

    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) stage_piece(s, k);
        slab_issued();
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");         // slab 0 has landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa0[i] = tr_frag<RB>(smem, 0, arow(i), lane);
        fb0[i] = tr_frag<RB>(smem + SLAB, 0, wn * 128 + i * 16, lane);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");         // slab 1 has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // Slab f: its fragments are in one register set; slab f + 1 is read into the other out of ring buffer
    // (f + 1) % 5; slab f + 4 goes into the buffer slab f - 1 was read from (those reads were waited for
    // before the barrier that ended slab f - 2).  The wait at the end of slab f leaves the 16 pieces of slabs
    // f + 3 and f + 4 in flight: slab f + 2 has landed when the barrier is passed.
    int rd = 1, wr = 4;                                       // ring positions of slab f + 1 and slab f + 4
    auto slab = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[8], bf16x8 (&na)[8], bf16x8 (&nb)[8]) __attribute__((always_inline)) {
        const char* As = smem + rd * STAGE;
        const char* Bs = As + SLAB;
        // One request rides behind every MFMA (a wave issues in order: whatever is not an MFMA has to fit into
        // the cycles the matrix pipeline spends on the MFMA just issued, there is no second wave to fill them).
        // The 32 transposed reads are spread evenly (four per seven MFMAs up to MFMA 55, the last eight MFMAs cover
        // the latency of the last read), the eight LDS-DMA pieces of slab f + 4 take slots in between.
        bf16x4 half[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // accumulated in place in an accumulation register (the builtin form shuffles them between registers)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
                const int n = 8 * i + j, grp = n / 7, pos = n - 7 * grp;
                if (n < 56 && (pos & 1) == 0) {
                    if (ABL & 2) continue;
                    const int m = 4 * grp + (pos >> 1), set = m >> 2, part = m & 3;
                    half[part] = tr_half<RB>(part < 2 ? As : Bs, part < 2 ? arow(set) : wn * 128 + set * 16, lane, part & 1);
                    if (part == 1) na[set] = bf16x8{half[0][0], half[0][1], half[0][2], half[0][3], half[1][0], half[1][1], half[1][2], half[1][3]};
                    if (part == 3) nb[set] = bf16x8{half[2][0], half[2][1], half[2][2], half[2][3], half[3][0], half[3][1], half[3][2], half[3][3]};
                } else if (n < 56 && pos == 1) {
                    if (!(ABL & 1)) stage_piece(wr, grp);
                }
            }
            // column sums of A: one more MFMA against ones (owned row group) or zeros, accumulated in ordinary registers
            if (CS && i % NO == 0)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb[i / NO]) : "v"(fa[i]), "v"(wsel[i / NO]));
        }
        slab_issued();
        if (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd = rd + 1 == RING ? 0 : rd + 1;
        wr = wr + 1 == RING ? 0 : wr + 1;
    };
    // One loop body for every slab and no second copy of it behind the loop: the accumulators then sit in the same
    // registers from the first MFMA to the write-out, and the compiler never has to move one (it does not know
    // when the MFMAs above, which are assembly to it, have written them).
    for (int it = 0; it < ((ns + 1) >> 1); ++it) {
        slab(fa0, fb0, fa1, fb1);
        slab(fa1, fb1, fa0, fb0);
    }
    // the re-staged pieces too: LDS is handed back after this; and the last MFMAs have left the pipeline before
    // the accumulators are read (the compiler does not know the assembly above writes them late)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    if constexpr (CS) {
        if (owner >= 0 && l15 == 0) {
#pragma unroll
            for (int k = 0; k < NCS; ++k) {
                const int grp = owner + k * NO;                   // row group of accb[k]
                if (grp < 8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = bm + wm * 128 + grp * 16 + 4 * g + r;
                        if (m < p.M) atomicAdd(p.a_colsum + m, accb[k][r]);
                    }
                }
            }
        }
    }
    if (p.no_flush) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 1.2345e-30f) p.C[0] = keep;          // keeps the accumulators alive
        return;
    }
    if (p.slab != nullptr) {
        // partial tile in the order gemm_tn_reduce_kernel reads: [wave of the eight-wave form][i][j & 3][lane]
        f32x4* base = reinterpret_cast<f32x4*>(p.slab) + (long)(split * p.ntiles + tile) * 8 * (8 * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                base[((wm * 4 + wn * 2 + (j >> 2)) * 32 + ((i + rot) & 7) * 4 + (j & 3)) * 64] = acc[i][j];
        return;
    }
    const int m0 = bm + wm * 128 + 4 * g, n0 = bn + wn * 128 + l15;
    float* crow = p.C + (long)m0 * p.ldc + n0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = ((i + rot) & 7) * 16 + r;
            float* c = crow + (long)mr * p.ldc;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (m0 + mr < p.M && n0 + j * 16 < p.N) atomicAdd(c + j * 16, acc[i][j][r]);
        }
}

// Second pass of the two-pass write-out: C[m][n] += sum over the splits of the partial tiles.  One thread per
// accumulator fragment (tile, wave, i, j, lane): the slab reads are 1 KiB contiguous per wave and split.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N,
                                                             long ldc, int tiles_n, int ntiles, int splits) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // ((tile * 8 + wave) * 32 + i * 4 + j) * 64 + lane
    if (idx >= (long)ntiles * 8 * 32 * 64) return;
    const int lane = (int)(idx & 63), frag = (int)((idx >> 6) & 31), wid = (int)((idx >> 11) & 7), tile = (int)(idx >> 14);
    const int i = frag >> 2, j = frag & 3, wm = wid >> 2, wn = wid & 3, l15 = lane & 15, g = lane >> 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(slab) + idx;
    f32x4 sum = src[0];
    for (int s = 1; s < splits; ++s) sum += src[(long)s * ntiles * (8 * 32 * 64)];
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 256 + wm * 128 + i * 16 + 4 * g, n = tn * 256 + wn * 64 + j * 16 + l15;
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (m0 + r < M) C[(long)(m0 + r) * ldc + n] += sum[r];
}

using S128 = Shape<128, 128, 2, 2>;
// Raises a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device: a process that drives several
// GPUs must set it on each before a > 64 KiB launch there).  `done` is the caller's per-kernel-instance table.
constexpr int ATTR_MAX_DEV = 64;
struct AttrOnce { bool done[ATTR_MAX_DEV] = {}; };
inline int ensure_lds_attr(const void* kernel, int lds, AttrOnce& once) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const bool tracked = dev >= 0 && dev < ATTR_MAX_DEV;
    if (tracked && once.done[dev]) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    if (tracked) once.done[dev] = true;
    return 0;
}

using S256x128 = Shape<256, 128, 4, 2, 3>;
using S256 = Shape<256, 256, 2, 4>;
using S256x128k32 = Shape<256, 128, 2, 2, 3, 32, 2>;   // 4 waves, 72 KiB LDS: two workgroups per CU

template <typename S, int EPI, typename TC>
int launch_nt(NtArgs a, hipStream_t st) {
    static AttrOnce attr_once;
    if (const int rc = ensure_lds_attr((const void*)gemm_nt_bf16_kernel<S, EPI, TC>, S::LDS, attr_once)) return rc;
    a.tiles_n = vitk_ceil_div(a.N, S::BN);
    a.tiles_m = vitk_ceil_div(a.M - a.m0, S::BM);
    a.nwg = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<S, EPI, TC>), dim3(a.nwg), dim3(S::NT), S::LDS, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

// Tile shape by problem size.  The 256x256 tile moves the fewest bytes per FLOP and measures
// fastest whenever its grid covers most of the 256 CUs (even with a partial last round); smaller
// grids fall back to shapes that produce more workgroups.
int pick_shape(int M, int N) {
    const long n256 = (long)vitk_ceil_div(M, 256) * vitk_ceil_div(N, 256);
    const long n256x128 = (long)vitk_ceil_div(M, 256) * vitk_ceil_div(N, 128);
    if (n256 >= 192) {
        // ping-pong 256x256 (falls back to the plain one for K < 192); the rows of a last round
        // that would fill under ~half of the chip are handed to small tiles instead (measured, same
        // process: fill 0.23-0.47 -> 2-8 % faster split, fill 0.62-0.93 -> 2-4 % faster whole)
        const double r = (double)n256 / 256.0, frac = r - (double)(long)r;
        return (r >= 1.0 && frac > 0.0 && frac < 0.55) ? 5 : 4;
    }
    if (n256x128 >= 192) return 1;
    return 0;
}

template <int EPI, typename TC>
int launch_nt_pp(NtArgs a, hipStream_t st) {
    static AttrOnce attr_once;
    constexpr int LDS = 2 * 2 * 256 * 128 + 8 * 4096;   // two staging buffers + the epilogue scratch
    if (const int rc = ensure_lds_attr((const void*)gemm_nt_pp_kernel<EPI, TC>, LDS, attr_once)) return rc;
    a.tiles_n = vitk_ceil_div(a.N, 256);
    if (a.tiles_m <= 0) a.tiles_m = vitk_ceil_div(a.M - a.m0, 256);   // else: the caller limits the rows
    a.nwg = a.tiles_m * a.tiles_n;
#ifdef VITK_GEMM_STAMP
    if constexpr ((EPI == VITK_EPI_BIAS_GELU && std::is_same<TC, bf16_t>::value) ||
                  (EPI == VITK_EPI_DGELU && std::is_same<TC, bf16_t>::value) ||
                  (EPI == VITK_EPI_BIAS_RES && std::is_same<TC, float>::value)) {
        if (g_gemm_abl >= 256) {           // epilogue ablations: mask << 8
            auto go = [&](auto k) -> int {
                hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(k, dim3(a.nwg), dim3(512), LDS, st, a);
                VITK_CHECK_LAUNCH();
                return 0;
            };
            switch (g_gemm_abl >> 8) {
                case 1: return go(gemm_nt_pp_kernel<EPI, TC, 1 << 8>);
                case 2: return go(gemm_nt_pp_kernel<EPI, TC, 2 << 8>);
                case 3: return go(gemm_nt_pp_kernel<EPI, TC, 3 << 8>);
                case 4: return go(gemm_nt_pp_kernel<EPI, TC, 4 << 8>);
                case 6: return go(gemm_nt_pp_kernel<EPI, TC, 6 << 8>);
                case 8: return go(gemm_nt_pp_kernel<EPI, TC, 8 << 8>);
                case 9: return go(gemm_nt_pp_kernel<EPI, TC, 9 << 8>);
            }
            return VITK_ERR_UNSUPPORTED;
        }
    }
    if constexpr (EPI == VITK_EPI_NONE && std::is_same<TC, bf16_t>::value) {
        if (g_gemm_abl) {
            auto go = [&](auto k) -> int {
                hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(k, dim3(a.nwg), dim3(512), LDS, st, a);
                VITK_CHECK_LAUNCH();
                return 0;
            };
            switch (g_gemm_abl) {
                case 1: return go(gemm_nt_pp_kernel<EPI, TC, 1>);
                case 2: return go(gemm_nt_pp_kernel<EPI, TC, 2>);
                case 3: return go(gemm_nt_pp_kernel<EPI, TC, 3>);
                case 4: return go(gemm_nt_pp_kernel<EPI, TC, 4>);
                case 7: return go(gemm_nt_pp_kernel<EPI, TC, 7>);
                case 8: return go(gemm_nt_pp_kernel<EPI, TC, 8>);
                case 10: return go(gemm_nt_pp_kernel<EPI, TC, 10>);
                case 16: return go(gemm_nt_pp_kernel<EPI, TC, 16>);
                case 32: return go(gemm_nt_pp_kernel<EPI, TC, 32>);
                case 64: return go(gemm_nt_pp_kernel<EPI, TC, 64>);
                case 96: return go(gemm_nt_pp_kernel<EPI, TC, 96>);
            }
            return VITK_ERR_UNSUPPORTED;
        }
    }
#endif
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, TC>), dim3(a.nwg), dim3(512), LDS, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}


// Workgroups of the persistent ping-pong NT kernel (one per CU); 0 = one workgroup per tile instead.
#ifdef VITK_GEMM_STAMP
int g_persist = 0;               // the cycle stamps live in the one-tile kernel
#else
int g_persist = 256;
#endif

template <int EPI, typename TC>
int launch_nt_pps(NtArgs a, int max_wgs, hipStream_t st) {
    static AttrOnce attr_once;
    constexpr int LDS = 2 * 2 * 256 * 128 + 8 * 4096;   // two staging buffers + the epilogue scratch
    if (const int rc = ensure_lds_attr((const void*)gemm_nt_pps_kernel<EPI, TC>, LDS, attr_once)) return rc;
    a.tiles_n = vitk_ceil_div(a.N, 256);
    if (a.tiles_m <= 0) a.tiles_m = vitk_ceil_div(a.M - a.m0, 256);   // else: the caller limits the rows
    a.nwg = a.tiles_m * a.tiles_n;
    const int grid = a.nwg < max_wgs ? a.nwg : max_wgs;
    hipLaunchKernelGGL((gemm_nt_pps_kernel<EPI, TC>), dim3(grid), dim3(512), LDS, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

// Ping-pong launch of a.tiles_m tile rows: the persistent form when it is enabled and the 32-bit source
// offsets it keeps cannot overflow, else one workgroup per tile.
template <int EPI, typename TC>
int launch_nt_pp_any(const NtArgs& a, hipStream_t st) {
    const long max_a = (long)a.M * a.lda + a.K, max_b = (long)a.N * a.ldb + a.K;
    if (g_persist > 0 && max_a < (1L << 31) && max_b < (1L << 31)) return launch_nt_pps<EPI, TC>(a, g_persist, st);
    return launch_nt_pp<EPI, TC>(a, st);
}

template <int EPI, typename TC>
int dispatch_nt(const NtArgs& a, int shape, hipStream_t st) {
    if ((shape == 4 || shape == 5) && a.K < 192) shape = 2;     // the ping-pong pipeline needs three K-tiles
    if (shape == 5) {
        // Whole rounds of 256x256 tiles over the 256 CUs, then the leftover rows as 128x128 tiles
        // (two workgroups per CU): a last round that would fill under ~half the chip costs a full
        // tile time otherwise.
        const int tiles_n = vitk_ceil_div(a.N, 256), tiles_m = vitk_ceil_div(a.M, 256);
        const int rounds = tiles_m * tiles_n / 256;
        const int rows_main = rounds * 256 / tiles_n;             // whole tile rows in the full rounds
        if (rounds < 1 || rows_main >= tiles_m) return launch_nt_pp_any<EPI, TC>(a, st);
        NtArgs m = a;
        m.tiles_m = rows_main;
        int rc = launch_nt_pp_any<EPI, TC>(m, st);
        if (rc) return rc;
        NtArgs t = a;
        t.m0 = rows_main * 256;
        return launch_nt<S128, EPI, TC>(t, st);
    }
    switch (shape) {
        case 4: return launch_nt_pp_any<EPI, TC>(a, st);
        case 0: return launch_nt<S128, EPI, TC>(a, st);
        case 1: return launch_nt<S256x128, EPI, TC>(a, st);
        case 3: return launch_nt<S256x128k32, EPI, TC>(a, st);
        default: return launch_nt<S256, EPI, TC>(a, st);
    }
}

template <typename S>
int launch_tn(TnArgs a, int split_k, hipStream_t st) {
    static AttrOnce attr_once;
    if (const int rc = ensure_lds_attr((const void*)gemm_tn_bf16_kernel<S>, S::LDS, attr_once)) return rc;
    a.tiles_n = vitk_ceil_div(a.N, S::BN);
    a.ntiles = vitk_ceil_div(a.M, S::BM) * a.tiles_n;
    const int nkt = vitk_ceil_div(a.T, BK);
    const int slots = S::NT == 512 ? 256 : 512;      // resident workgroups on the chip
    if (split_k <= 0) {
        // fill the chip once (each slice then runs one long reduction loop), keep >= 8 steps per slice
        split_k = slots / a.ntiles;
        if (split_k < 1) split_k = 1;
        const int cap = nkt / 8 > 0 ? nkt / 8 : 1;
        if (split_k > cap) split_k = cap;
    }
    if (split_k > nkt) split_k = nkt;
    a.kt_per_split = vitk_ceil_div(nkt, split_k);
    split_k = vitk_ceil_div(nkt, a.kt_per_split);
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<S>), dim3(a.ntiles * split_k), dim3(S::NT), S::LDS, st, a);
    VITK_CHECK_LAUNCH();
    return 0;
}

int g_tn_no_flush = 0;
int g_tn_w4 = 1;                 // weight-gradient kernel of the 256x256 path: 1 four waves (one per SIMD), 0 eight waves (ping-pong)
#ifdef VITK_GEMM_STAMP
int g_tn_abl = 0;                // which parts of the weight-gradient loop the next launches leave out (see the kernel)
#endif
// Per-device workspace for the two-pass write-out of the ping-pong weight-gradient kernel (vitk_gemm_tn_set_workspace)
constexpr int TN_MAX_DEV = 64;
float* g_tn_ws[TN_MAX_DEV] = {};
size_t g_tn_ws_bytes[TN_MAX_DEV] = {};
// The workspace holds the partial tiles of ONE weight-gradient GEMM at a time: launches on one stream follow each
// other, launches on two streams could overlap and overwrite each other's partials.  The first stream that uses
// a registration owns it; a launch on any other stream takes the atomic write-out instead.
hipStream_t g_tn_ws_stream[TN_MAX_DEV] = {};
bool g_tn_ws_claimed[TN_MAX_DEV] = {};
template <bool CS, int NO, int ABL>
int launch_tn_w4(const TnPpArgs& a, dim3 grid, int lds, hipStream_t st) {
    static AttrOnce attr_once;                     // one table per kernel instance
    if (const int rc = ensure_lds_attr((const void*)gemm_tn_w4_kernel<CS, NO, ABL>, lds, attr_once)) return rc;
    hipLaunchKernelGGL((gemm_tn_w4_kernel<CS, NO, ABL>), grid, dim3(256), lds, st, a);
    return 0;
}

// Weight-gradient kernels of the 256x256 path; returns -1 when the problem does not fit their preconditions.
int launch_tn_pp(const TnArgs& t, int split_k, hipStream_t st) {
    constexpr int LDS = 5 * 2 * 32 * 512;
    if (t.T % 32) return -1;
    TnPpArgs a;
    a.A = t.A; a.B = t.B; a.C = t.C; a.a_colsum = t.a_colsum;
    a.M = t.M; a.N = t.N; a.lda = t.lda; a.ldb = t.ldb; a.ldc = t.ldc;
    a.tiles_n = vitk_ceil_div(t.N, 256);
    a.ntiles = vitk_ceil_div(t.M, 256) * a.tiles_n;
    a.slabs = t.T / 32;
    a.no_flush = g_tn_no_flush;
    if (split_k <= 0) {
        split_k = 256 / a.ntiles;                          // one workgroup per CU
        if (split_k < 1) split_k = 1;
        const int cap = a.slabs / 16 > 0 ? a.slabs / 16 : 1;
        if (split_k > cap) split_k = cap;
    }
    if (split_k > a.slabs) split_k = a.slabs;
    a.slabs_per_split = vitk_ceil_div(a.slabs, split_k);
    split_k = vitk_ceil_div(a.slabs, a.slabs_per_split);
    if (a.slabs - (split_k - 1) * a.slabs_per_split < 5) return -1;   // last slice too short for the ring
    static AttrOnce attr_once;
    if (const int rc = ensure_lds_attr((const void*)gemm_tn_pp_kernel<0>, LDS, attr_once)) return rc;
    // Several splits and a registered workspace: every split stores its partial tile (plain 1 KiB stores) and a
    // second launch sums them into C.  Adding the tiles with atomics instead costs 36-55 us per call: the L2
    // atomic units take about 330 G adds/s whatever the access shape.
    a.slab = nullptr;
    int dev = 0;
    if (split_k > 1 && !g_tn_no_flush && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < TN_MAX_DEV &&
        g_tn_ws[dev] != nullptr && g_tn_ws_bytes[dev] >= (size_t)a.ntiles * split_k * 256 * 256 * sizeof(float)) {
        if (!g_tn_ws_claimed[dev]) { g_tn_ws_claimed[dev] = true; g_tn_ws_stream[dev] = st; }
        if (g_tn_ws_stream[dev] == st) a.slab = g_tn_ws[dev];
    }
#ifdef VITK_GEMM_STAMP
    if (g_tn_abl) {
        auto go = [&](auto k) -> int {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k, dim3(a.ntiles * split_k), dim3(512), LDS, st, a);
            VITK_CHECK_LAUNCH();
            return 0;
        };
        switch (g_tn_abl) {
            case 1: return go(gemm_tn_pp_kernel<1>);
            case 2: return go(gemm_tn_pp_kernel<2>);
            case 3: return go(gemm_tn_pp_kernel<3>);
            case 4: return go(gemm_tn_pp_kernel<4>);
            case 7: return go(gemm_tn_pp_kernel<7>);
        }
        return VITK_ERR_UNSUPPORTED;
    }
#endif
    // (the four-wave kernel addresses a slab with 32-bit byte offsets per lane)
    if (g_tn_w4 && 32 * a.lda * 2 < (1L << 31) && 32 * a.ldb * 2 < (1L << 31)) {
        const dim3 grid(a.ntiles * split_k);
        int e = 0;
        if (g_tn_w4 == 2) e = launch_tn_w4<false, 8, 1>(a, grid, LDS, st);       // loop ablations (garbage results, times only):
        else if (g_tn_w4 == 3) e = launch_tn_w4<false, 8, 2>(a, grid, LDS, st);  // 2 no DMA, 3 no reads, 4 neither, 8 neither and no barriers
        else if (g_tn_w4 == 4) e = launch_tn_w4<false, 8, 3>(a, grid, LDS, st);
        else if (g_tn_w4 == 8) e = launch_tn_w4<false, 8, 7>(a, grid, LDS, st);
        else if (a.a_colsum == nullptr) e = launch_tn_w4<false, 8, 0>(a, grid, LDS, st);
        else if (a.tiles_n >= 4) e = launch_tn_w4<true, 8, 0>(a, grid, LDS, st);
        else if (a.tiles_n == 3) e = launch_tn_w4<true, 6, 0>(a, grid, LDS, st);
        else if (a.tiles_n == 2) e = launch_tn_w4<true, 4, 0>(a, grid, LDS, st);
        else e = launch_tn_w4<true, 2, 0>(a, grid, LDS, st);
        if (e) return e;
    } else {
        hipLaunchKernelGGL(gemm_tn_pp_kernel<0>, dim3(a.ntiles * split_k), dim3(512), LDS, st, a);
    }
    VITK_CHECK_LAUNCH();
    if (a.slab != nullptr) {
        const long frags = (long)a.ntiles * 8 * 32 * 64;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, st, a.slab, a.C, a.M, a.N,
                           a.ldc, a.tiles_n, a.ntiles, split_k);
        VITK_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace

extern "C" int vitk_gemm_tn_set_workspace(void* workspace, size_t bytes) {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= TN_MAX_DEV) return VITK_ERR_UNSUPPORTED;
    if (((uintptr_t)workspace & 15) || (workspace == nullptr && bytes != 0)) return VITK_ERR_SHAPE;
    g_tn_ws[dev] = (float*)workspace;
    g_tn_ws_bytes[dev] = workspace ? bytes : 0;
    g_tn_ws_claimed[dev] = false;          // the next launch that uses it claims it for its stream
    return 0;
}

static int g_force_shape = -1;   // test hook: -1 = choose by size
extern "C" int vitk_gemm_set_persistent(int wgs) {
    if (wgs < 0 || wgs > 8192) return VITK_ERR_SHAPE;
    g_persist = wgs;
    return 0;
}

#ifdef VITK_GEMM_STAMP
// Copies the stamps of the last ping-pong NT launch to the host (n_wg x 10 words).
extern "C" int vitk_debug_gemm_stamps(unsigned long long* out, int n_wg) {
    if (!out || n_wg <= 0 || n_wg > STAMP_WGS) return VITK_ERR_SHAPE;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), (size_t)n_wg * 10 * sizeof(unsigned long long));
}
#endif
extern "C" void vitk_debug_force_gemm_shape(int shape) { g_force_shape = shape; }
// Diagnostic: 1 = the ping-pong weight-gradient kernel skips its atomic write-out (timing of the flush only).
extern "C" void vitk_debug_tn_no_flush(int on) { g_tn_no_flush = on; }
extern "C" void vitk_debug_tn_w4(int on) { g_tn_w4 = on; }
#ifdef VITK_GEMM_STAMP
extern "C" void vitk_debug_tn_abl(int mask) { g_tn_abl = mask; }
#endif

int vitk_gemm_nt_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda,
                      long ldb, long ldc, int epilogue, const GemmEpi& epi, int out_dtype, hipStream_t st) {
    if (K % BK || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return VITK_ERR_SHAPE;
    NtArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.m0 = 0; a.tiles_m = 0; a.tiles_n = 0; a.nwg = 0;
    a.epi = epi;
    const int shape = g_force_shape >= 0 ? g_force_shape : pick_shape(M, N);
    if (out_dtype == VITK_F32) {
        // fp32 output = the residual stream (and its producers)
        switch (epilogue) {
            case VITK_EPI_BIAS: return dispatch_nt<VITK_EPI_BIAS, float>(a, shape, st);
            case VITK_EPI_BIAS_RES: return dispatch_nt<VITK_EPI_BIAS_RES, float>(a, shape, st);
            case VITK_EPI_PATCH: return dispatch_nt<VITK_EPI_PATCH, float>(a, shape, st);
        }
        return VITK_ERR_UNSUPPORTED;
    }
    if (out_dtype != VITK_BF16) return VITK_ERR_DTYPE;
    switch (epilogue) {
        case VITK_EPI_NONE: return dispatch_nt<VITK_EPI_NONE, bf16_t>(a, shape, st);
        case VITK_EPI_BIAS: return dispatch_nt<VITK_EPI_BIAS, bf16_t>(a, shape, st);
        case VITK_EPI_BIAS_GELU: return dispatch_nt<VITK_EPI_BIAS_GELU, bf16_t>(a, shape, st);
        case VITK_EPI_BIAS_RES: return dispatch_nt<VITK_EPI_BIAS_RES, bf16_t>(a, shape, st);
        case VITK_EPI_DGELU: return dispatch_nt<VITK_EPI_DGELU, bf16_t>(a, shape, st);
        case VITK_EPI_PATCH: return dispatch_nt<VITK_EPI_PATCH, bf16_t>(a, shape, st);
    }
    return VITK_ERR_UNSUPPORTED;
}

int vitk_gemm_tn_bf16(const void* A, const void* B, float* C, float* a_colsum, int M, int N, int T, long lda,
                      long ldb, long ldc, int split_k, hipStream_t st) {
    if (M % 8 || N % 8 || lda % 8 || ldb % 8) return VITK_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return VITK_ERR_SHAPE;
    TnArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.a_colsum = a_colsum;
    a.M = M; a.N = N; a.T = T; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.tiles_n = a.ntiles = a.kt_per_split = 0;
    int shape = g_force_shape;
    if (shape < 0) {
        // ping-pong 256x256 when there are enough tiles, or when each of the ~256 reduction slices
        // still runs a long loop (measured: 768x768 with 112 slabs per slice favours it, 56 does not)
        const long nt = (long)vitk_ceil_div(M, 256) * vitk_ceil_div(N, 256);
        const long slabs_per_slice = nt > 0 ? (long)(T / 32) * nt / 256 : 0;
        shape = (T >= 2048 && (nt >= 16 || slabs_per_slice >= 96)) ? 4 : 0;
    }
    if (shape >= 4) {
        const int rc = launch_tn_pp(a, split_k, st);
        if (rc >= 0) return rc;
        shape = 2;
    }
    switch (shape) {
        case 0: return launch_tn<S128>(a, split_k, st);
        case 1: return launch_tn<S256x128>(a, split_k, st);
        default: return launch_tn<S256>(a, split_k, st);
    }
}
