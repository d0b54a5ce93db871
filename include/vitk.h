/* vitk: C-ABI of the MI355X (gfx950) kernels behind the ViT forward/backward hot path.
 *
 * The reference (tkd26/ViT-pytorch) has no native layer: every operation below is, there, a
 * `torch.nn` call inside models/ViT/modeling.py.  Each entry point names the reference lines
 * whose arithmetic it replaces.  All entry points
 *   - take raw device pointers + sizes (no torch types), the caller owns every buffer;
 *   - enqueue on the given hipStream_t (passed as void*) and never synchronise or allocate;
 *   - return 0 on success, a hipError_t value if a launch failed, or a VITK_ERR_* code (<0)
 *     if the arguments violate a documented constraint (nothing is launched then).
 * `dtype` selects the activation type: VITK_BF16 (production) or VITK_F32 (parity checks).
 * Parameters (gamma/beta/bias/pos/cls) and all gradients of parameters are always fp32.
 */
#ifndef VITK_H
#define VITK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_VERSION 100

enum vitk_dtype { VITK_BF16 = 0, VITK_F32 = 1 };

enum vitk_epilogue {
    VITK_EPI_NONE = 0,      /* C = A.B^T */
    VITK_EPI_BIAS = 1,      /* C = A.B^T + bias */
    VITK_EPI_BIAS_GELU = 2, /* u = A.B^T + bias ; C = dropout(gelu(u)) ; aux = gelu'(u) - 1/2 where C was kept, -1/2 where dropped */
    VITK_EPI_BIAS_RES = 3,  /* C = dropout(A.B^T + bias) + residual */
    VITK_EPI_DGELU = 4,     /* C = (A.B^T) * (aux + 1/2) / (1 - dropout_p), aux as written by VITK_EPI_BIAS_GELU: d C/d u of that call (seed unused) */
    VITK_EPI_PATCH = 5      /* C[row skipping cls] = dropout(A.B^T + bias + pos) */
};

enum vitk_error {
    VITK_ERR_SHAPE = -1,    /* size/stride/alignment constraint violated */
    VITK_ERR_DTYPE = -2,
    VITK_ERR_NULL = -3,
    VITK_ERR_UNSUPPORTED = -4,
    VITK_ERR_SCRATCH = -5,  /* problem larger than the built-in scratch array of a *_static entry point */
    VITK_ERR_COMM = -100     /* RCCL failure: code = VITK_ERR_COMM - ncclResult_t */
};

typedef struct vitk_gemm_epilogue_args {
    const float* bias;     /* [N] */
    const void* residual;  /* [M, ld_res], activation dtype */
    void* aux;             /* [M, ld_aux], see vitk_epilogue; element type: IEEE half when dtype is VITK_BF16
                              (values in [-0.63, 0.63]: 11 significant bits in the same two bytes), fp32 else */
    const float* pos;      /* [(npatch+1), N] fp32, VITK_EPI_PATCH only */
    long ld_res;
    long ld_aux;
    int npatch;            /* VITK_EPI_PATCH: patches per image */
    float dropout_p;       /* 0 disables dropout */
    uint64_t seed;         /* dropout stream; element index = out_row * N + col */
} vitk_gemm_epilogue_args;

int vitk_version(void);
const char* vitk_error_string(int code);

/* ---- LayerNorm (reference modeling.py:173-174,180,185,233,244: nn.LayerNorm(eps=1e-6)) ----
 * One wavefront per row, single pass, fp32 statistics.  cols % 4 == 0, cols <= 8192.
 * x:[rows,cols] in `dtype` (fp32 residual stream or bf16); y in `out_dtype`;
 * mean/rstd:[rows] fp32 (saved for backward). */
int vitk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int rows, int cols, float eps,
                       int dtype, int out_dtype, void* stream);
/* dx = LN'(dy) (+ dres if non-null: the residual branch's gradient, fused add);
 * dgamma/dbeta:[cols] fp32 are ACCUMULATED (+=; caller zeroes or pre-loads them) through
 * `workspace` (vitk_layernorm_bwd_workspace(rows, cols) bytes, contents undefined).
 * dy is in `dy_dtype`; x, dres, dx in `dtype` (the residual stream).  cols <= 2048. */
size_t vitk_layernorm_bwd_workspace(int rows, int cols);
int vitk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* dgamma,
                       float* dbeta, void* workspace, int rows, int cols, int dtype, int dy_dtype,
                       void* stream);
/* Same, for an fp32 residual stream whose gradient `dx` is also the output gradient of the linear
 * layer that produced x with a dropout + residual epilogue (modeling.py:92-93,117-118,182,187): also
 * writes `dz` = bf16(dropout_mask(dx) / keep), the operand that layer's dX and dW GEMMs consume, so no
 * separate mask/convert pass reads dx again.  (dropout_p, seed) are that layer's; p = 0 is a plain
 * bf16 copy. */
int vitk_layernorm_bwd_dz(const void* dy, const void* x, const float* gamma,
                          const float* mean, const float* rstd, const void* dres, void* dx,
                          void* dz, float dropout_p, uint64_t seed,
                          float* dgamma, float* dbeta, void* workspace, int rows, int cols,
                          int dtype, int dy_dtype, void* stream);

/* No counterpart in the reference.  vitk_layernorm_bwd for callers that cannot allocate device memory or
 * bring a workspace (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc):
 * the per-block dgamma/dbeta partials go to an 8 MiB array built into the library (one copy per GPU, held
 * until the process ends), so the call allocates nothing.  cols <= 1024, else VITK_ERR_SCRATCH (nothing is
 * launched).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams,
 * two threads) race; calls on one stream are ordered and safe.  Results equal vitk_layernorm_bwd's. */
int vitk_layernorm_bwd_static(const void* dy, const void* x, const float* gamma, const float* mean,
                              const float* rstd, const void* dres, void* dx, float* dgamma,
                              float* dbeta, int rows, int cols, int dtype, int dy_dtype, void* stream);

/* ---- Linear layers (modeling.py:58-62,74-76,92 q/k/v/out; :100-101,114-118 fc1/fc2) ----
 * C[M,N] = epilogue(A[M,K] . B[N,K]^T).  Both operands K-contiguous (nn.Linear layout).
 * `out_dtype` is the type of C and of the residual: with bf16 operands it may be VITK_F32 for
 * the epilogues that feed the fp32 residual stream (BIAS, BIAS_RES, PATCH).
 * bf16: K % 64 == 0, N % 8 == 0, lda/ldb/ldc % 8 == 0, 16-byte aligned pointers.
 * fp32: K % 4 == 0, N % 4 == 0. */
int vitk_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K,
                 long lda, long ldb, long ldc, int epilogue,
                 const vitk_gemm_epilogue_args* args, int dtype, int out_dtype, void* stream);
/* Weight gradient (autograd of nn.Linear: dW = dY^T.X):  C[M,N] += A[T,M]^T . B[T,N], C fp32, the token axis T
 * split into `split_k` slices (0 = choose) that run side by side.  Their partial results are summed through the
 * workspace registered with vitk_gemm_tn_set_workspace (two plain passes, a fixed order of addition) when there is
 * one and the call is on the stream that owns it, else with atomics into C.  M % 8 == 0, N % 8 == 0.
 * `a_colsum` (may be NULL): a_colsum[m] += sum_t A[t][m], the bias gradient db = sum_t dY[t],
 * produced in the same pass over A. */
int vitk_gemm_tn_acc(const void* A, const void* B, float* C, float* a_colsum, int M, int N, int T,
                     long lda, long ldb, long ldc, int split_k, int dtype, void* stream);
/* Launch shape of the large-tile bf16 GEMM (256 x 256 tiles).  Its workgroups are persistent: `wgs` of them
 * are launched and each walks every wgs-th tile, fetching the next tile's first operands while it writes
 * the current one out (a CU otherwise idles 1.6-3.7 us between two one-tile workgroups).  Default 256 =
 * one per CU.  When other kernels hold CUs for long stretches (the RCCL all-reduce on the side stream during
 * backward) a workgroup that finds no CU waits for one of the others to finish its whole share; a multiple
 * of the CU count bounds that wait (the data-parallel wrapper sets 1024).  0 = one workgroup per tile.
 * 0 <= wgs <= 8192.  No counterpart in the reference. */
int vitk_gemm_set_persistent(int wgs);
/* Workspace of the weight-gradient GEMM's two-pass write-out, for the current device (no counterpart in the
 * reference).  vitk_gemm_tn_acc splits the token axis over the chip; with a workspace of at least 64 MiB registered
 * every split stores its partial 256 x 256 tiles there and a second launch sums them into C; without one (NULL, 0:
 * the state after loading the library) the splits add into C with atomics, which costs 36-55 us more per call.
 * The memory stays the caller's.  The workspace holds the partial tiles of one call at a time, so it serves ONE
 * stream per device: the first stream that calls vitk_gemm_tn_acc after a registration owns it until the next
 * registration; calls on any other stream of that device use the atomic write-out (correct, slower). */
int vitk_gemm_tn_set_workspace(void* workspace, size_t bytes);
/* Test hook: force the workgroup tile of the bf16 GEMMs (0: 128x128, 1: 256x128, 2: 256x256,
 * -1: choose by problem size, the default).  No counterpart in the reference. */
void vitk_debug_force_gemm_shape(int shape);
/* Bias gradient: out[N] += column sums of X[T,N]. */
int vitk_colsum_acc(const void* X, float* out, int T, int N, long ld, int dtype, void* stream);

/* ---- Scaled dot-product attention (modeling.py:78-91), fused, online softmax ----
 * q/k/v element (b, tok, h, d) lives at ptr[(b*N + tok)*ld_qkv + h*dh + d]; o likewise with
 * ld_o (i.e. heads are read from / merged into token-major matrices: transpose_for_scores
 * `:68-71` and the permute/contiguous `:89-91` never materialise).  lse:[B,H,N] fp32 holds
 * log(sum(exp(scale*q.k))) per query.  bf16 requires dh == 64; fp32 requires dh <= 128.
 * o_f32 (optional, bf16 only, same addressing as o): unrounded copy of o; handing it to
 * vitk_attn_bwd keeps delta = rowsum(dO*O) free of the bf16 rounding of O, which otherwise is
 * amplified by the cancellation in P*(dP - delta). */
int vitk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* o_f32, float* lse,
                  int B, int H, int N, int dh, long ld_qkv, long ld_o, float scale,
                  int dtype, void* stream);
/* delta:[B,H,N] fp32 workspace.  dq/dk/dv use the q/k/v addressing with ld_dqkv.
 * fp32 path accumulates dk/dv with atomics: the caller must zero them. */
int vitk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* o_f32,
                  const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv,
                  int B, int H, int N, int dh, long ld_qkv, long ld_o, long ld_dqkv,
                  float scale, int dtype, void* stream);
/* The same two operators with nn.Dropout(attention_dropout_rate) on the probabilities (modeling.py:63,86):
 * element ((b*H + h)*N + query)*N + key of the [B,H,N,N] probability tensor is kept when a counter-based hash of
 * (seed, element) says so and scaled by 1/(1 - dropout_p); the normaliser and lse are those of the full row, as
 * in the reference (softmax first, dropout second).  The backward call must get the forward call's dropout_p and
 * seed.  0 <= dropout_p < 1; with 0 they are vitk_attn_fwd / vitk_attn_bwd.  With dropout_p > 0 the bf16
 * instance runs the streaming kernels for every N. */
int vitk_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* o_f32, float* lse,
                          int B, int H, int N, int dh, long ld_qkv, long ld_o, float scale,
                          float dropout_p, uint64_t seed, int dtype, void* stream);
int vitk_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const float* o_f32,
                          const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv,
                          int B, int H, int N, int dh, long ld_qkv, long ld_o, long ld_dqkv,
                          float scale, float dropout_p, uint64_t seed, int dtype, void* stream);
/* The persistent attention-backward kernel launches `factor` x (number of CUs) workgroups, each walking
 * an equal share of the (image, head) pairs; only one fits a CU at a time.  When another kernel holds
 * CUs for a long time while it runs - the RCCL all-reduce of a gradient bucket on the side stream
 * during backward - a workgroup that cannot be placed starts late and the whole launch waits for its
 * share; a larger factor bounds that delay to 1/factor of the launch.  On an otherwise idle GPU
 * factors 1..4 time the same (0.43 ms per ViT-B/16 layer at batch 512).  Default 1; the
 * data-parallel wrapper sets 4.  1 <= factor <= 8.  No counterpart in the reference. */
int vitk_attn_set_oversubscription(int factor);
/* Test hook for the bf16 kernels: 0 = default (up to 224 tokens: persistent workgroups with a feeder
 * wave), 1 = always the streaming kernels (any N), 2 = one-shot LDS-resident forward + streaming
 * backward, 3 = one-shot LDS-resident forward + one-shot fused backward.  No counterpart in the reference. */
void vitk_debug_force_attn_streaming(int on);
/* Attention probabilities for `vis=True` (modeling.py:84-85): probs:[B,H,N,N] fp32. */
int vitk_attn_probs(const void* q, const void* k, const float* lse, float* probs,
                    int B, int H, int N, int dh, long ld_qkv, float scale, int dtype, void* stream);

/* ---- Window attention of the Swin blocks (models/Swin/swin_RKR.py:178-215 WindowAttention.forward,
 * with the cyclic shift `:316-319,333-336`, window_partition / window_reverse `:90-119,322-330` and the
 * shift mask `:281-304` of SwinTransformerBlock folded into the addressing) ----
 * qkv:[B, H*W, 3, heads, dh] is the token-major output of the qkv projection on the UN-shifted,
 * un-partitioned tokens; out:[B, H*W, heads*dh] likewise (ready for the output projection).
 * S = scale*q.k + bias_table[rel(i,j), head] + (-100 where the two tokens lie in different regions of
 * the shifted frame; shift == 0: no mask).  bias_table:[(2*window-1)^2, heads] fp32 (the parameter as
 * the reference stores it), lse:[B*windows, heads, window^2] fp32.  dh == 32, window^2 <= 64,
 * H % window == 0, W % window == 0, 0 <= shift < window. */
int vitk_window_attn_fwd(const void* qkv, const float* bias_table, void* out, float* lse,
                         int B, int H, int W, int heads, int dh, int window, int shift, float scale,
                         int dtype, void* stream);
/* dqkv: same layout as qkv, every element written.  dbias_table:[replicas, (2*window-1)^2, heads] fp32
 * is accumulated into with atomics: the caller zeroes it and sums the replicas afterwards.  Every
 * workgroup flushes its partial table once, thousands of them into the same few cache lines; the
 * flushes are spread over `replicas` >= 1 copies. */
int vitk_window_attn_bwd(const void* qkv, const float* bias_table, const void* out, const void* d_out,
                         const float* lse, void* dqkv, float* dbias_table, int replicas,
                         int B, int H, int W, int heads, int dh, int window, int shift, float scale,
                         int dtype, void* stream);

/* RKR weight fold (models/ViT/modeling_RKR.py:56-78 RG_FC.forward + SFG_FC.forward moved to the weight side):
 * Wt[o][i] = (W[o][i] + sum_k RM[k][o] LM[i][k]) * F[o], bt[o] = b[o] * F[o]; W:[O,I], LM:[I,K], RM:[K,O],
 * F:[O] or NULL (no scale); all fp32; K <= 8, I % 4 == 0.  vitk_rkr_fold_bwd turns dWt, dbt into the
 * gradients of W (dW may be NULL: frozen weight), b, LM (accumulated with atomics: the caller zeroes dLM),
 * RM and F (dF NULL exactly when F is). */
int vitk_rkr_fold_fwd(const float* W, const float* b, const float* LM, const float* RM, const float* F,
                      float* Wt, float* bt, int O, int I, int K, void* stream);
int vitk_rkr_fold_bwd(const float* W, const float* b, const float* LM, const float* RM, const float* F,
                      const float* dWt, const float* dbt, float* dW, float* db, float* dLM, float* dRM,
                      float* dF, int O, int I, int K, void* stream);

/* Piggyback mask fold (models/ViT/layers_PB.py:14-54 Binarizer / Ternarizer, :105-111 ElementWiseLinear.forward's
 * `mask_thresholded * weight`, and the masked factors / scales of ElementWiseRG :253-297 and ElementWiseSFG :351-368):
 * out[i] = hard(mask_real[i]) * w[i] over n fp32 elements, hard = 1 above `threshold`, else 0 (ternary == 0), or
 * additionally -1 below zero (ternary == 1).  The thresholding passes gradients through unchanged (the reference's
 * backward returns gradOutput): d_mask = d_out * w, d_w = d_out * hard(mask_real); either may be NULL (frozen).
 * Pointers 16-byte aligned. */
int vitk_mask_fold_fwd(const float* mask_real, const float* w, float* out, long n, int ternary, float threshold,
                       void* stream);
int vitk_mask_fold_bwd(const float* mask_real, const float* w, const float* d_out, float* d_mask, float* d_w,
                       long n, int ternary, float threshold, void* stream);

/* Stochastic depth (models/Swin/swin_RKR.py:340-341, `shortcut + drop_path(branch)` with timm's DropPath):
 * out[s, e] = h[s, e] + scale[s] * y[s, e] for `samples` samples of `per_sample` elements (a multiple
 * of 4); h, out fp32 (the residual stream), y in `dtype`; scale[s] = keep-mask / keep-probability.
 * vitk_scale_cast is the branch's backward: dy[s, e] = scale[s] * dout[s, e], written in `dtype`. */
int vitk_scale_add(const float* h, const void* y, const float* scale, float* out, long samples, long per_sample,
                   int dtype, void* stream);
int vitk_scale_cast(const float* dout, const float* scale, void* dy, long samples, long per_sample, int dtype,
                    void* stream);

/* Test hook: 1 = run the bf16 window attention on the VALU kernels that serve fp32 instead of the MFMA
 * kernels.  No counterpart in the reference. */
void vitk_debug_force_window_attn_valu(int on);

/* ---- Patch embedding (modeling.py:144-147,155-165: Conv2d(k=s=P) + cls + pos + dropout) ----
 * im2col: img fp32 NCHW [B,C,Himg,Wimg] -> cols [B*gh*gw, C*P*P] (row = (b, py, px), col =
 * (c, i, j)), then vitk_gemm_nt(..., VITK_EPI_PATCH) writes rows 1.. of every image and
 * vitk_embed_cls writes row 0 (cls + pos[0], same dropout stream). */
int vitk_im2col(const float* img, void* cols, int B, int C, int Himg, int Wimg, int P,
                int dtype, void* stream);
int vitk_embed_cls(const float* cls, const float* pos, void* out, int B, int ntok, int D,
                   float dropout_p, uint64_t seed, int dtype, void* stream);
/* Backward of the token assembly: dE:[B,ntok,D] -> dtok:[B*(ntok-1), D] (dropout mask
 * re-applied, class rows dropped); dpos:[ntok,D] += sum_b; dcls:[D] += sum_b row 0. */
int vitk_embed_bwd(const void* dE, void* dtok, float* dpos, float* dcls, int B, int ntok, int D,
                   float dropout_p, uint64_t seed, int dE_dtype, int dtype, void* stream);

/* ---- R50 hybrid stem (models/ViT/modeling_resnet.py) on channels-last activations ----
 * Weight standardisation of StdConv2d (:35-42): w_hat[o][:] = (w[o][:] - mean_o) / sqrt(var_o + eps) over the
 * F = in*kh*kw weights of output channel o (biased variance, eps 1e-5 in the reference); rstd:[O] is saved for
 * the backward, which turns d(loss)/d(w_hat) into d(loss)/d(w).  All fp32. */
int vitk_wstd_fwd(const float* w, float* w_hat, float* rstd, int O, int F, float eps, void* stream);
int vitk_wstd_bwd(const float* dw_hat, const float* w_hat, const float* rstd, float* dw, int O, int F, void* stream);
/* GroupNorm (+ ReLU when `relu`) of PreActBottleneck / the root (:55-91,139-141: nn.GroupNorm(32, C, eps=1e-6)
 * followed by nn.ReLU, and nn.GroupNorm(C, C) of the projection shortcut) on x:[B, HW, C] fp32, channels last
 * (the layout in which a 1x1 convolution is a GEMM over B*HW token rows).  C a power of two in 4..1024, C % G == 0.
 * y:[B, HW, C] in `out_dtype` (fp32 for the residual sum and in front of an im2col, bf16 for
 * the next GEMM); mean, rstd:[B, G] fp32 are saved for the backward.  workspace: vitk_groupnorm_workspace bytes. */
size_t vitk_groupnorm_workspace(int B, int HW, int C, int G);
int vitk_groupnorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       void* workspace, int B, int HW, int C, int G, float eps, int relu, int out_dtype, void* stream);
/* dx:[B, HW, C] in `dx_dtype` (fp32, or the compute dtype when the only reader is the backward of the convolution
 * GEMM that produced x, which would round it anyway); dgamma, dbeta:[C] are ACCUMULATED (+=); dy in `dy_dtype`; with
 * `relu` the mask is recomputed from x. */
int vitk_groupnorm_bwd(const void* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                       int B, int HW, int C, int G, int relu, int dy_dtype, int dx_dtype, void* stream);
/* The last GroupNorm of a bottleneck unit together with the end of the unit (:88-91): out = relu(residual +
 * GroupNorm(x)) in fp32 and, when out_lp is given, a second copy in `lp_dtype` (as vitk_add_relu_fwd writes it); the
 * normalised branch itself is never stored.  mean, rstd, workspace as for vitk_groupnorm_fwd.
 * Its backward, vitk_groupnorm_add_relu_bwd: d:[B, HW, C] fp32 = (g1 [+ g2]) where out > 0 else 0 (as
 * vitk_add_relu_bwd: the gradient of the residual), written by the first GroupNorm-backward pass, which takes its sums
 * from the same values; dx, dgamma, dbeta as vitk_groupnorm_bwd(dy = d, relu = 0). */
int vitk_groupnorm_add_relu_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                                void* out_lp, float* mean, float* rstd, void* workspace, int B, int HW, int C, int G,
                                float eps, int lp_dtype, void* stream);
int vitk_groupnorm_add_relu_bwd(const float* g1, const void* g2, const float* out, const float* x, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, float* d, void* dx, float* dgamma,
                                float* dbeta, void* workspace, int B, int HW, int C, int G, int g2_dtype, int dx_dtype,
                                void* stream);
/* End of a bottleneck unit (:91): out = relu(a + b) on n fp32 elements (n % 4 == 0), optionally also written in
 * `lp_dtype` to out_lp (the operand of the next unit's 1x1 GEMMs; may be NULL).  Backward: d = (g1 [+ g2]) where
 * out > 0 else 0, the gradient of both addends; g2 (may be NULL) is a second incoming gradient in `g2_dtype`. */
int vitk_add_relu_fwd(const float* a, const float* b, float* out, void* out_lp, long n, int lp_dtype, void* stream);
int vitk_add_relu_bwd(const float* g1, const void* g2, const float* out, float* d, long n, int g2_dtype, void* stream);

/* The KH x KW convolutions (StdConv2d 3x3 :48-50, the 7x7 root :137) as im2col + GEMM, like the 1x1 ones:
 *   forward   Y[row][o]             = sum_k cols[row][k] * Wk[o][k]          (vitk_gemm_nt)
 *   weights   dWk[o][k]             = sum_rows dY[row][o] * cols[row][k]     (vitk_gemm_tn_acc)
 *   input     dcols[row][k]         = sum_o dY[row][o] * Wk[o][k]            (vitk_gemm_nt), then vitk_col2im_conv
 * with row = (b, oy, ox), k = (kh, kw, c), Wk the weight reordered to [O, (kh, kw, c)].
 * vitk_im2col_conv: cols[(b, oy, ox)][(kh, kw, c)] = x[b][oy*stride + kh - pad][ox*stride + kw - pad][c], zero outside
 * the image and in the padding columns KH*KW*C .. Kp-1; x fp32, addressed with the element strides sb, sh, sw, sc
 * (channels-last activations, or the NCHW image under the root); cols [B*Ho*Wo, Kp] in `dtype`.
 * vitk_col2im_conv: its adjoint for a channels-last input: dx:[B, H, W, C] fp32 = gather-sum of dcols (in `dtype`).
 * The activations stay channels-last between the operators, so no layout change surrounds a convolution. */
int vitk_im2col_conv(const float* x, void* cols, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                     long sb, long sh, long sw, long sc, int Kp, int dtype, void* stream);
int vitk_col2im_conv(const void* dcols, float* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                     int Kp, int dtype, void* stream);

/* ---- Element-wise helpers ----
 * Cast fp32 master weights [R,C] to the activation dtype, optionally also transposed [C,R]. */
int vitk_cast_weight(const float* w, void* out, void* out_t, int R, int C, int dtype, void* stream);
/* The same for many weights in one launch per 64 entries (host table, passed to the kernel by value): what an
 * optimizer step leaves behind for the next forward pass instead of one cast launch per layer in it.  `out`
 * [rows, cols] is contiguous (may be NULL), `out_t` [cols, ld_t] has row stride ld_t >= rows (may be NULL), so
 * several weights can share one concatenated operand (query | key | value: out pointers one block of rows
 * apart, out_t pointers `rows` columns apart with ld_t = total rows).  `tile0` is filled in by the call. */
typedef struct vitk_cast_entry {
    const void* w;         /* fp32 [rows, cols], contiguous */
    void* out;
    void* out_t;
    int rows, cols;
    int ld_t;
    int tile0;
} vitk_cast_entry;
int vitk_cast_weights_batched(const vitk_cast_entry* entries, int n_entries, int dtype, void* stream);
/* dx (dtype) = dy (dy_dtype) * mask / keep with the mask of the forward epilogue (element index
 * row*cols+col): also the fp32 -> bf16 hand-over from the residual-stream gradient to the GEMMs. */
int vitk_dropout_bwd(const void* dy, void* dx, long rows, int cols, float dropout_p,
                     uint64_t seed, int dy_dtype, int dtype, void* stream);
/* dst (dtype_dst) = src (dtype_src), n elements, n % 4 == 0. */
int vitk_convert(const void* src, void* dst, long n, int dtype_src, int dtype_dst, void* stream);

/* ---- Optimizer step (train.py:535-538: torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0),
 * SGD(momentum=0.9).step() built at train.py:475-478, optimizer.zero_grad()) ----
 * Both calls walk a device-resident table of fp32 tensors.  Entry e owns blocks
 * [block0, block0 + ceil(n / 8192)); block0 is ascending, n_blocks is the total.
 * flags: bit 0 = momentum buffer holds nothing yet (first step: buffer := gradient, as
 * torch.optim.SGD does); bit 1 = n % 4 == 0 and all three pointers are 16-byte aligned.
 * mom may be NULL (momentum 0). */
typedef struct vitk_sgd_entry {
    void* param; void* grad; void* mom;
    long n;
    int block0;
    int flags;
} vitk_sgd_entry;
/* slots (device, VITK_SGD_SLOTS floats owned by the caller): slots[0] := sum of squares of every gradient in the
 * table, slots[1..63] := 0; the rest is scratch for the partial sums.  The sum is formed without atomics, in a
 * fixed order: equal gradients give a bit-identical norm on every GPU and in every run, so replicas that apply
 * the same reduced gradients stay identical. */
#define VITK_SGD_SLOTS (64 + 1024)
int vitk_sgd_norm_sq(const void* table, int n_entries, int n_blocks, float* slots, void* stream);
/* coef = slots ? min(1, max_norm / (sqrt(sum slots) + 1e-6)) : 1;  g' = coef*grad + weight_decay*param;
 * mom = first ? g' : momentum*mom + (1 - dampening)*g';  param -= lr*mom.  Gradients are left as they are. */
int vitk_sgd_step(const void* table, int n_entries, int n_blocks, const float* slots, float max_norm,
                  float lr, float momentum, float dampening, float weight_decay, void* stream);

/* ---- Gradient exchange (DistributedDataParallel at train.py:274-278: broadcast at wrap time, mean
 * all-reduce of every gradient inside loss.backward() at train.py:528) ----
 * One RCCL communicator per process, one process per GPU.  Rank 0 obtains a unique id
 * (vitk_comm_unique_id, 128 bytes) and hands it to the other ranks through any host channel;
 * every rank then calls vitk_comm_init.  vitk_allreduce / vitk_broadcast work in place on `count`
 * elements of `dtype` and are enqueued on `stream` (nothing synchronises the host).
 * `average` != 0 divides by the world size (ncclAvg). */
int vitk_comm_unique_id(void* id_out, size_t bytes);
int vitk_comm_init(int rank, int world, const void* unique_id, void** comm_out);
int vitk_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream);
int vitk_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);
int vitk_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
