// extern "C" entry points that dispatch on the activation dtype (see include/vitk.h).
#include "common.h"

int vitk_gemm_nt_bf16(const void*, const void*, void*, int, int, int, long, long, long, int, const GemmEpi&, int, hipStream_t);
int vitk_gemm_nt_f32(const void*, const void*, void*, int, int, int, long, long, long, int, const GemmEpi&, hipStream_t);
int vitk_gemm_tn_bf16(const void*, const void*, float*, float*, int, int, int, long, long, long, int, hipStream_t);
int vitk_gemm_tn_f32(const void*, const void*, float*, int, int, int, long, long, long, int, hipStream_t);
int vitk_attn_fwd_bf16(const void*, const void*, const void*, void*, float*, float*, int, int, int, long, long, float, float,
                       uint64_t, hipStream_t);
int vitk_attn_bwd_bf16(const void*, const void*, const void*, const void*, const float*, const void*, const float*, float*,
                       void*, void*, void*, int, int, int, long, long, long, float, float, uint64_t, hipStream_t);
int vitk_attn_fwd_f32(const void*, const void*, const void*, void*, float*, int, int, int, int, long, long, float, float,
                      uint64_t, hipStream_t);
int vitk_attn_bwd_f32(const void*, const void*, const void*, const void*, const void*, const float*, float*, void*,
                      void*, void*, int, int, int, int, long, long, long, float, float, uint64_t, hipStream_t);
int vitk_attn_probs_any(const void*, const void*, const float*, float*, int, int, int, int, long, float, int,
                        hipStream_t);

extern "C" int vitk_version(void) { return VITK_VERSION; }

extern "C" const char* vitk_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case VITK_ERR_SHAPE: return "vitk: size/stride/alignment constraint violated";
        case VITK_ERR_DTYPE: return "vitk: unsupported dtype combination";
        case VITK_ERR_NULL: return "vitk: null pointer argument";
        case VITK_ERR_UNSUPPORTED: return "vitk: unsupported option";
        case VITK_ERR_SCRATCH: return "vitk: problem exceeds the built-in scratch array of this entry point";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    if (code <= VITK_ERR_COMM) return "vitk: RCCL call failed (ncclResult_t = VITK_ERR_COMM - code)";
    return "vitk: unknown error";
}

extern "C" int vitk_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb,
                            long ldc, int epilogue, const vitk_gemm_epilogue_args* args, int dtype, int out_dtype,
                            void* stream) {
    if (!A || !B || !C) return VITK_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return VITK_ERR_SHAPE;
    GemmEpi e{};
    e.N = N;
    if (args) {
        e.bias = args->bias; e.residual = args->residual; e.aux = args->aux; e.pos = args->pos;
        e.ld_res = args->ld_res; e.ld_aux = args->ld_aux; e.npatch = args->npatch;
        e.thr16 = vitk_drop_thr16(args->dropout_p);
        e.drop_scale = vitk_drop_scale(e.thr16);
        e.seed = args->seed;
    }
    const bool need_bias = epilogue == VITK_EPI_BIAS || epilogue == VITK_EPI_BIAS_GELU ||
                           epilogue == VITK_EPI_BIAS_RES || epilogue == VITK_EPI_PATCH;
    if (need_bias && !e.bias) return VITK_ERR_NULL;
    if (epilogue == VITK_EPI_BIAS_RES && !e.residual) return VITK_ERR_NULL;
    if ((epilogue == VITK_EPI_BIAS_GELU || epilogue == VITK_EPI_DGELU) && !e.aux) return VITK_ERR_NULL;
    if (epilogue == VITK_EPI_PATCH && (!e.pos || e.npatch <= 0 || M % e.npatch)) return VITK_ERR_SHAPE;
    const int align = dtype == VITK_BF16 ? 8 : 4;
    if (e.residual && e.ld_res % align) return VITK_ERR_SHAPE;
    if (e.aux && e.ld_aux % align) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16) return vitk_gemm_nt_bf16(A, B, C, M, N, K, lda, ldb, ldc, epilogue, e, out_dtype, st);
    if (dtype == VITK_F32) {
        if (out_dtype != VITK_F32) return VITK_ERR_DTYPE;
        return vitk_gemm_nt_f32(A, B, C, M, N, K, lda, ldb, ldc, epilogue, e, st);
    }
    return VITK_ERR_DTYPE;
}

extern "C" int vitk_gemm_tn_acc(const void* A, const void* B, float* C, float* a_colsum, int M, int N, int T,
                                long lda, long ldb, long ldc, int split_k, int dtype, void* stream) {
    if (!A || !B || !C) return VITK_ERR_NULL;
    if (M <= 0 || N <= 0 || T <= 0) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16) return vitk_gemm_tn_bf16(A, B, C, a_colsum, M, N, T, lda, ldb, ldc, split_k, st);
    if (dtype == VITK_F32) {
        // the fp32 instance keeps the column sums in their own pass
        if (a_colsum) {
            const int rc = vitk_colsum_acc(A, a_colsum, T, M, lda, VITK_F32, stream);
            if (rc) return rc;
        }
        return vitk_gemm_tn_f32(A, B, C, M, N, T, lda, ldb, ldc, split_k, st);
    }
    return VITK_ERR_DTYPE;
}

extern "C" int vitk_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* o_f32, float* lse, int B,
                                     int H, int N, int dh, long ld_qkv, long ld_o, float scale, float dropout_p,
                                     uint64_t seed, int dtype, void* stream) {
    if (!q || !k || !v || !o || !lse) return VITK_ERR_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16) {
        if (dh != 64) return VITK_ERR_SHAPE;
        return vitk_attn_fwd_bf16(q, k, v, o, o_f32, lse, B, H, N, ld_qkv, ld_o, scale, dropout_p, seed, st);
    }
    if (dtype == VITK_F32) return vitk_attn_fwd_f32(q, k, v, o, lse, B, H, N, dh, ld_qkv, ld_o, scale, dropout_p, seed, st);
    return VITK_ERR_DTYPE;
}

extern "C" int vitk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* o_f32, float* lse, int B,
                             int H, int N, int dh, long ld_qkv, long ld_o, float scale, int dtype, void* stream) {
    return vitk_attn_fwd_dropout(q, k, v, o, o_f32, lse, B, H, N, dh, ld_qkv, ld_o, scale, 0.f, 0, dtype, stream);
}

extern "C" int vitk_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const float* o_f32,
                                     const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv, int B,
                                     int H, int N, int dh, long ld_qkv, long ld_o, long ld_dqkv, float scale,
                                     float dropout_p, uint64_t seed, int dtype, void* stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv) return VITK_ERR_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return VITK_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VITK_BF16) {
        if (dh != 64) return VITK_ERR_SHAPE;
        return vitk_attn_bwd_bf16(q, k, v, o, o_f32, d_o, lse, delta, dq, dk, dv, B, H, N, ld_qkv, ld_o, ld_dqkv, scale,
                                  dropout_p, seed, st);
    }
    if (dtype == VITK_F32)
        return vitk_attn_bwd_f32(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, N, dh, ld_qkv, ld_o, ld_dqkv, scale,
                                 dropout_p, seed, st);
    return VITK_ERR_DTYPE;
}

extern "C" int vitk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* o_f32,
                             const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int N,
                             int dh, long ld_qkv, long ld_o, long ld_dqkv, float scale, int dtype, void* stream) {
    return vitk_attn_bwd_dropout(q, k, v, o, o_f32, d_o, lse, delta, dq, dk, dv, B, H, N, dh, ld_qkv, ld_o, ld_dqkv, scale,
                                 0.f, 0, dtype, stream);
}

extern "C" int vitk_attn_probs(const void* q, const void* k, const float* lse, float* probs, int B, int H, int N,
                               int dh, long ld_qkv, float scale, int dtype, void* stream) {
    if (!q || !k || !lse || !probs) return VITK_ERR_NULL;
    if (dtype != VITK_BF16 && dtype != VITK_F32) return VITK_ERR_DTYPE;
    return vitk_attn_probs_any(q, k, lse, probs, B, H, N, dh, ld_qkv, scale, dtype, (hipStream_t)stream);
}
