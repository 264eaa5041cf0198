/*
 * curv_hip.h -- C ABI of libcurv_hip.so, the gfx950 (MI355X) implementation of the KFAC / EFB / INF
 * curvature hot path of DLR-RM/curvature.
 *
 * Every entry point is what a binding of the reference's `Curvature` plugin API would call for the
 * arithmetic it performs today through torch (reference file:line cited per function, paths relative
 * to the reference checkout).  Conventions:
 *   - plain C types only; all tensors are dense row-major fp32 device buffers unless stated;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work and
 *     never synchronise the device, so they are ordered after whatever produced their inputs on
 *     that stream (e.g. torch's backward);
 *   - the library owns no device memory: scratch comes from the caller (`*_workspace_bytes`);
 *   - return value 0 = ok, otherwise a CURV_ERR_* code with text in curv_last_error();
 *     numerical failure ("not positive definite") is reported through caller-provided device
 *     `info` words so that a whole model can be processed without a host round trip per layer;
 *   - re-entrant; results never depend on library state.  What the library keeps per calling thread (and device)
 *     besides the error string: a set of internal HIP streams and events (curv_init_streams below) on which the
 *     whole-model sweeps of curv_chol_inv_lower* / curv_kfac_accumulate fork and join the caller's stream.  They
 *     carry no data between calls; when they are created relative to the process's other streams can change the
 *     TIME of the unchecked inversion entry points (see curv_init_streams), never a result bit.
 */
#ifndef CURV_HIP_H
#define CURV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CURV_ABI_VERSION 10

#define CURV_OK 0
#define CURV_ERR_NOT_PD 1
#define CURV_ERR_INVALID 2
#define CURV_ERR_WORKSPACE 3
#define CURV_ERR_HIP 4
#define CURV_ERR_NOT_CONVERGED 5

int curv_version(void);
const char* curv_last_error(void);

/* Create the library's internal streams (per calling thread and current device) NOW instead of at the first
 * curv_chol_inv_lower / curv_kfac_accumulate call.  Whole-model inversions run their factor groups on four internal
 * streams; the runtime deals hardware queues onto the four pipes of the command processor in creation order, and a queue
 * that is not empty - the caller's own stream, which waits for the sweep - slows every queue on its pipe.  Whether one of
 * the sweep's two chains shares that pipe depends on how many streams the process created before the set (three raw
 * streams in front of it: invert() of the ResNet-50 factors 6.9 -> 10.9 ms until round 5).  curv_chol_inv_lower_status
 * no longer depends on it (it joins the caller's stream only after the host has the verdict: 6.83 / 6.80 / 6.87 ms for
 * none / three streams after / three before the set); curv_chol_inv_lower and curv_chol_factor_inverse, which never wait
 * on the host, still do: for those call this once, early - but AFTER the caller's own stream has launched something
 * (hardware queues are given out lazily, in order of first use).  The Python estimators do it in their constructor
 * (curvature_amd._lib.init_streams runs a one-element torch kernel first).  Idempotent.  (LAB_NOTEBOOK R5.6) */
int curv_init_streams(void);

/* ------------------------------------------------------------------------------------------------
 * KFAC factor build:  dst (+)= scale * X X^T           (curvature/curvatures.py:312-352)
 *
 * One descriptor per Kronecker factor.  X is never materialised: it is the implicit im2col of
 * `src` (the reference's F.unfold at :329-330, rows ordered (c, kh, kw), columns (n, oh, ow)) plus,
 * when `has_bias`, the row of ones the reference concatenates at :333-335.
 *   A-side of Conv2d : src = layer input  (N,C,H,W), kernel/stride/padding of the layer,
 *                      scale = 1/(N*Ho*Wo)
 *   G-side of Conv2d : src = grad_output  (N,Cout,Ho,Wo) with kh=kw=1, stride 1, padding 0,
 *                      has_bias = 0, scale = N/(Ho*Wo)   (the reference scales grad_output by N
 *                      in its backward hook, :310, and divides by N*Ho*Wo at :343)
 *   Linear           : src = (N,C) viewed as (N,C,1,1); A: scale = 1/N, G: scale = N
 * `dst` is the (dim x dim) fp32 factor, dim = C*kh*kw + has_bias.  `first` != 0 overwrites dst
 * (the reference's `self.state[layer] = [...]` at :350), 0 accumulates (`+=` at :347-348).
 * The result is exactly symmetric.  Dilation and groups are not representable: the reference
 * ignores them (:329), callers must reject such layers.
 * ---------------------------------------------------------------------------------------------- */
typedef struct curv_factor_desc {
  const float* src;
  float* dst;
  int32_t N, C, H, W;
  int32_t kh, kw, sh, sw, ph, pw;
  int32_t has_bias;
  int32_t first;
  float scale;
  int32_t path_hint;   /* 0: the library picks the launch form from the launch's own size; CURV_PATH_SMALL / CURV_PATH_GROUPED:
                        * the caller says which form the UNSHARDED model's launch takes, so that a layer-sharded rank
                        * (whose share may fall under the small-launch threshold) sums every factor in the same order as
                        * the unsharded run.  All descriptors of a call carry the same value. */
} curv_factor_desc;
#define CURV_PATH_AUTO 0
#define CURV_PATH_SMALL 1
#define CURV_PATH_GROUPED 2
/* executed multiply-add flops (32 x 32 blocks on and above the diagonal, 2 * 1024 * pairs * K per factor) up to which a
 * launch takes the two-launch small form (csrc/syrk_small.hip) */
#define CURV_SMALL_MAX_FLOP 2.0e9

/* Host-only: the launch form (CURV_PATH_SMALL or CURV_PATH_GROUPED) a curv_kfac_accumulate call with exactly these
 * factors takes on its own - every gate of the small form (executed flops, number of factors, slice length, workgroup
 * count) evaluated by the library itself; `src` / `dst` are not read and any path_hint in `descs` is ignored.  A
 * layer-sharded caller passes the geometry of ALL factors of the model once and puts the answer into the path_hint
 * of its own share's descriptors (curvature/curvatures.py:20-21: layers are independent, so a rank's results must not
 * depend on what else the rank holds). */
int curv_kfac_path_for(const curv_factor_desc* descs, int n_factors);

/* Device scratch needed by curv_kfac_accumulate for this set of factors (bytes). */
size_t curv_kfac_workspace_bytes(const curv_factor_desc* descs, int n_factors);

/* Host-only introspection of the launch plan (tests, tuning): writes CURV_PLAN_INFO_FIELDS values per
 * factor: dim, Ho, Wo, chunk samples, chunk rows, chunk cols, n_chunks, LDS row stride, plane stride,
 * sample stride, channels per panel, n_tiles, chunks per item, k-slices, n_items, item_base, tile edge,
 * float4 staging flag, log2 padded patch row length, 64x64 sub-tiles of the reduce pass (0 for an unsliced factor), the number
 * of serial segments an UNSLICED 128x128-tile factor cuts its K range into (0: k-sliced through slabs; > 0: its items scale,
 * add into dst and write the mirror tile themselves, flushing their accumulators behind every segment), log2 row lanes
 * that walk patch rows while staging (the remaining row lanes split channels), 1 if the patch images are staged
 * by LDS-DMA from a pre-tiled copy of the source (full-width chunks of a kh x kw > 1 convolution), 1 if the factor is built by the LDS-DMA kernel for flattened per-pixel
 * factors (its own work list: item bases count from 0 per kernel; n_chunks = stages of <= 16 pixels), 2 if it is a
 * 3x3 / stride 1 / padding 1 factor assembled from 29 shifted correlations that run as virtual factors of the LDS-DMA
 * kernel (no items of its own; C a multiple of 128: one virtual factor per correlation, C = 64: ten packed pair tiles that
 * hold four correlations each), and last the multiply-add FLOPs (2 per multiply-add) the plan executes for the
 * factor: dim (dim + 1) K for a symmetric product over K = samples x output pixels, the sum over its correlations
 * (C (C + 1) K' for the symmetric ones, 2 C^2 K' for the others) for an assembled factor. */
#define CURV_PLAN_INFO_FIELDS 25
int curv_kfac_plan_info(const curv_factor_desc* descs, int n_factors, long long* out);

/* Grouped build of all factors of a model: a fixed, small number of launches whatever the number of factors (the
 * implicit-im2col kernel, the LDS-DMA kernel for flattened factors, for the shifted correlations of 3x3 / stride 1 /
 * padding 1 factors and for the unfolded copies of stride-2 3x3 / strided 1x1 ones, one reduce launch for each, plus a
 * padding pass and an assembly pass when such 3x3 factors are present and an unfold pass for the strided ones).  A launch that is small as a whole (LeNet scale: at most 2 GFLOP) takes a two-launch build of its own
 * instead (32 x 32 blocks x K slices gathered straight from the tensors, then a reduce pass; CURV_KFAC_SMALL=0 in the
 * environment keeps such a launch on the grouped kernels).  `descs` is a host array; it may be reused as soon as the call
 * returns. */
int curv_kfac_accumulate(void* stream, const curv_factor_desc* descs, int n_factors, void* workspace,
                         size_t workspace_bytes);

/* Same, recording HIP events (from curv_event_create) on `stream` around EVERYTHING the call enqueues behind the
 * descriptor-table uploads (padding / pre-tiling passes, the MFMA kernels, the k-slice reductions, the 3x3 assembly), so
 * that a benchmark can time the whole build without a profiler. */
int curv_kfac_accumulate_timed(void* stream, const curv_factor_desc* descs, int n_factors, void* workspace,
                               size_t workspace_bytes, void* ev_start, void* ev_stop);

/* Same with flags (ev_start / ev_stop may be NULL).  CURV_KFAC_TABLE_RESIDENT: the caller vouches that the head of
 * `workspace` (the device descriptor table) has not been written by anybody else since this thread's previous
 * curv_kfac_accumulate* call with the same workspace; argument blocks of the table that did not change (same
 * pointers, geometry, scale and `first` flags - the steady state of a training loop) are then not uploaded again. */
#define CURV_KFAC_TABLE_RESIDENT 1u
int curv_kfac_accumulate_ex(void* stream, const curv_factor_desc* descs, int n_factors, void* workspace,
                            size_t workspace_bytes, unsigned flags, void* ev_start, void* ev_stop);
void* curv_event_create(void);
void curv_event_destroy(void* event);
int curv_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* waits for ev_stop */
int curv_event_synchronize(void* event);                                /* blocks the host until the event has fired */

/* ------------------------------------------------------------------------------------------------
 * KFAC.invert:  L = lower Cholesky factor of (sqrt(multiply) * F + sqrt(add) * I)^-1
 *                                                                (curvature/curvatures.py:354-385)
 * One descriptor per Kronecker factor; all factors of a model are processed by one batched sweep.
 * F is the (n x n) fp32 factor (read only), L the (n x n) fp32 output (zeros above the diagonal, as
 * torch's cholesky returns).  The damped matrix is formed in fp32 exactly as the reference does
 * (:368-375); the factorisation itself runs in fp64.  info[i] (device int32) is 0 on success and
 * 1 + the index of the failing pivot when factor i is not positive definite, in which case L holds
 * garbage: the caller raises the reference's RuntimeError (:380, scripts/hyper.py:141).
 * ---------------------------------------------------------------------------------------------- */
typedef struct curv_inv_desc {
  const float* F;
  float* L;
  int32_t n;
  int32_t reserved;
  double add;       /* n of the reference: sqrt(add) goes on the diagonal   */
  double multiply;  /* s of the reference: the factor is scaled by sqrt(s)   */
} curv_inv_desc;

size_t curv_chol_inv_workspace_bytes(const curv_inv_desc* descs, int n_factors);
int curv_chol_inv_lower(void* stream, const curv_inv_desc* descs, int n_factors, int* info, void* workspace,
                        size_t workspace_bytes);
/* The same call with an EARLY verdict for a host that must raise on "not positive definite" before it goes on (the
 * reference's invert() raises from torch.cholesky, curvatures.py:378-380): the n status words are also copied to
 * `host_status` (PINNED host memory, n ints) by an internal stream as soon as the last factorisation step of every factor
 * has run - before the finalize passes that write L, which do not touch them - and `ev_status` (curv_event_create) is
 * recorded behind that copy.  curv_event_synchronize(ev_status) then returns while the finalize passes still run (0.15 ms
 * for a ResNet-50), and the host prepares its next launches in their shadow; work enqueued on `stream` afterwards is
 * ordered behind the whole call as always.  Not available under stream capture (CURV_ERR_INVALID).
 * The call itself WAITS (host) for that copy before it makes `stream` wait for the sweep and returns (the status words are
 * then in `host_status`; curv_event_synchronize(ev_status) returns at once): a stream that sits on an unsatisfied wait for
 * the whole sweep slows the sweep's own streams down whenever they share a pipe of the command processor with it - the
 * "streams created before the estimator" effect (10.9 -> 7.3 ms for a ResNet-50).  curv_chol_inv_lower, which never
 * blocks the host, keeps that sensitivity. */
int curv_chol_inv_lower_status(void* stream, const curv_inv_desc* descs, int n_factors, int* info, void* workspace,
                               size_t workspace_bytes, int* host_status, void* ev_status);

/* X = chol_lower(M + diag_add * I)^-1 in fp64 for a batch of symmetric fp32 matrices (same blocked sweep,
 * no index reversal; M may also be given in fp64): the A_c^-1 and B_c^-1 of INF.pre_sampler (curvatures.py:566-567).  X is (n x n)
 * fp64, lower triangular with zeros above.  info as for curv_chol_inv_lower. */
typedef struct curv_cholinv_desc {
  const void* M;     /* fp32 matrix, or fp64 when m_is_f64 != 0 (diag_add is then added in fp64) */
  double* X;
  int32_t n;
  int32_t m_is_f64;
  double diag_add;
  /* optional (NULL: X = chol(M + d I)^-1 as above).  Non-NULL: RIGHT-HAND SIDE mode - R is an (n x n) fp64 lower-triangular
   * matrix and X receives chol(M + d I)^-1 R (lower triangular): the forward substitution runs inside the sweep, in the place
   * of the inverse accumulation and at its cost, and saves the explicit inverse's product with R afterwards
   * (INF.pre_sampler's B_c^-1 A_c^-1, :566-570).  Needs a third work matrix (curv_chol_factor_inverse_workspace_bytes). */
  const double* R;
  int32_t r_minus;   /* with R: X = R - chol(M + d I)^-1 R (pre_sampler's T = (I - B_c^-1) A_c^-1 in one go) */
  int32_t reserved;
  double pivot_min;  /* a pivot <= pivot_min counts as "not positive definite" (0: the plain test).  info then reports the first
                      * column whose pivot fell to the threshold, i.e. the numerical rank of a Gram matrix whose columns are
                      * in general position: the rank detection of the low-rank eigensolver (curvature_amd/ops.py: eigh) */
} curv_cholinv_desc;

size_t curv_chol_factor_inverse_workspace_bytes(const curv_cholinv_desc* descs, int n_mats);
int curv_chol_factor_inverse(void* stream, const curv_cholinv_desc* descs, int n_mats, int* info, void* workspace,
                             size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------
 * Dense contractions of the samplers and of EFB / INF: a batch of independent strided fp32 GEMMs
 *     C = epilogue(alpha * op(A) op(B)) [+ beta * C]
 * op(A) is M x K with element (i,k) at A[i*a_rs + k*a_cs]; op(B) is K x N with (k,j) at
 * B[k*b_rs + j*b_cs]; C is M x N with (i,j) at C[i*c_rs + j*c_cs] (strides in elements, so transposes
 * and column slices need no copies).  Epilogues: NONE; SQUARE (alpha*acc^2, EFB.update's (.)**2 at
 * curvatures.py:427); MUL_E / ADD_E (multiply by / add an elementwise operand E, used for the
 * lambda scaling of EFB.sample :458 and for writing mean + sample straight into a parameter, :67-82).
 * ---------------------------------------------------------------------------------------------- */
#define CURV_EPI_NONE 0
#define CURV_EPI_SQUARE 1
#define CURV_EPI_MUL_E 2
#define CURV_EPI_ADD_E 3
#define CURV_EPI_MUL_E_ADD_F 4   /* alpha*acc*E + F: INF.sampler's Y_l - r^2 * X_p_s written onto the mean (:596-599) */
/* triangular-operand hints: the K range of every output tile is cut where the operand is known to be 0 */
#define CURV_TRI_NONE 0
#define CURV_TRI_A_LOWER 1   /* op(A)(i,k) = 0 for k > i  (e.g. A = L_G) */
#define CURV_TRI_B_UPPER 2   /* op(B)(k,j) = 0 for k > j  (e.g. B = L_A^T) */

typedef struct curv_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  const float* E;
  long long a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, e_rs, e_cs;
  int32_t M, N, K;
  int32_t epilogue;
  float alpha, beta;
  int32_t tri;
  int32_t reserved;
  const float* F;          /* second elementwise operand (CURV_EPI_MUL_E_ADD_F), else NULL */
  long long f_rs, f_cs;
} curv_gemm_desc;

size_t curv_gemm_workspace_bytes(int n_desc);
/* The same plus room for K slicing: when a call holds fewer output tiles of K-contiguous products than the chip has
 * workgroup slots (a layer-sharded rank sampling one wide layer), products with K >= 1536 are cut into K slices whose
 * partial tiles are summed in a fixed order by a second launch.  With a workspace of only
 * curv_gemm_workspace_bytes the call still works, unsliced. */
size_t curv_gemm_workspace_bytes_for(const curv_gemm_desc* descs, int n_desc);
int curv_gemm_batched(void* stream, const curv_gemm_desc* descs, int n_desc, void* workspace,
                      size_t workspace_bytes);
/* The same with flags.  CURV_GEMM_TABLE_RESIDENT: this exact descriptor array was the previous call's on this
 * workspace and nobody else has written to the workspace since (a launch plan that owns its scratch and replays a fixed
 * list of products): the device copy of the table is not uploaded again. */
#define CURV_GEMM_TABLE_RESIDENT 1u
int curv_gemm_batched_ex(void* stream, const curv_gemm_desc* descs, int n_desc, void* workspace,
                         size_t workspace_bytes, unsigned flags);

/* fp64 variant (alpha/beta only) for the ill-conditioned products of INF.pre_sampler.  `tri`: triangular operands -
 * entries outside the triangle are neither read nor multiplied (they must be zero in memory where a tile straddles the
 * diagonal): the products of the two triangular inverses of pre_sampler (curvatures.py:566-572) are 2/3 of INF.invert's
 * flops when done densely.  A lower x B lower leaves the tiles above the diagonal of C untouched for beta = 1 and
 * zero for beta = 0.  The products of one call are INDEPENDENT (no C of one may be an operand of another): they run
 * up to 24 to a launch, products with both output edges >= 1024 on 128 x 128 tiles in a launch of their own behind the
 * others - not in the caller's order. */
#define CURV_TRI64_A_LOWER 1
#define CURV_TRI64_A_UPPER 2
#define CURV_TRI64_B_LOWER 4
#define CURV_TRI64_B_UPPER 8
#define CURV_TRI64_C_LOWER 16 /* square product known to be symmetric: tiles strictly above the diagonal are not computed
                                 * (their part of C is left untouched; elements above the diagonal inside diagonal tiles are written) */
typedef struct curv_gemm64_desc {
  const double* A;
  const double* B;
  double* C;
  long long a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
  int32_t M, N, K;
  int32_t tri;      /* CURV_TRI64_* flags, 0 = dense */
  double alpha, beta;
  /* optional fused epilogues (all NULL: C = alpha * acc + beta * C as above):
   *   E          C = alpha * acc + beta * E with E != C (same strides as C): T = A^-1 - B^-1 A^-1 of INF.pre_sampler
   *              without a copy of A^-1 first; tiles whose K range is empty are still written (beta * E)
   *   C32        fp32 output instead of C (C may be NULL; strides c_rs / c_cs count floats; beta must be 0):
   *              C32[i][j] = (float)(alpha * row_scale[i] * col_scale[j] * acc), either scale vector may be NULL (= 1):
   *              P_c = diag(sigma) L_c diag(sigma) of pre_sampler (:570) straight from the product */
  const double* E;
  const float* row_scale;
  const float* col_scale;
  float* C32;
} curv_gemm64_desc;
int curv_gemm_f64_batched(void* stream, const curv_gemm64_desc* descs, int n_desc);

/* out[0..count) ~ N(0,1): Philox4x32-10 keyed by `seed`, counter starting at `offset` (in units of 4
 * values); the draw of torch.randn at curvatures.py:391, :457, :590 with a device-side generator. */
int curv_randn(void* stream, float* out, long long count, unsigned long long seed, unsigned long long offset);
/* The same stream with its position kept on the DEVICE: the draw starts at *counter (units of 4 values) and a second
 * launch advances *counter by ceil(count / 4).  A captured HIP graph that contains the call therefore draws fresh
 * noise at every replay (a host-side offset would be frozen into the kernel arguments). */
int curv_randn_counter(void* stream, float* out, long long count, unsigned long long seed, unsigned long long* counter);

/* ------------------------------------------------------------------------------------------------
 * Elementwise pieces (Diagonal / EFB / INF)
 * ---------------------------------------------------------------------------------------------- */
/* out = (s*v + n)^(-1/2)     curvatures.py:188 (Diagonal.invert), :449 (EFB.invert), :526 (INF) */
int curv_rsqrt_affine(void* stream, const float* v, double s, double n, float* out, long long count);
/* state (+)= batch_size * [grad_w | grad_b]^2   curvatures.py:160-165 (Diagonal), :431-434 (EFB diags)
 * grad_w is (rows x cols_w), grad_b (rows) or NULL; state is (rows x (cols_w + (grad_b != NULL))). */
int curv_sq_accumulate(void* stream, const float* grad_w, const float* grad_b, int rows, int cols_w,
                       double batch_size, float* state, int first);

/* The same for many layers in one launch (Diagonal.update, the `diags` of EFB.update).  first: overwrite `state`
 * instead of adding to it. */
typedef struct curv_sq_desc {
  const float* grad_w;
  const float* grad_b;     /* may be NULL */
  float* state;            /* (rows, cols_w + (grad_b != NULL)) contiguous */
  int32_t rows, cols_w;
  int32_t first, reserved;
} curv_sq_desc;
int curv_sq_accumulate_batched(void* stream, const curv_sq_desc* descs, int n, double batch_size);
/* v = max(v, 0) in place      curvatures.py:523 */
int curv_clamp_min0(void* stream, float* v, long long count);
/* out = sqrt(s*v)             curvatures.py:525 */
int curv_sqrt_scale(void* stream, const float* v, double s, float* out, long long count);
/* out = a*b */
int curv_mul(void* stream, const float* a, const float* b, float* out, long long count);

/* Batched device-to-device copy: dst[i][0 .. bytes[i]) = src[i][...] for n buffers in one or a few
 * launches (Curvature.sample_and_replace reloads the mean weights with `load_state_dict`,
 * curvatures.py:118: ~320 separate tensors for a ResNet-50, i.e. ~320 copy launches otherwise).
 * Buffers may have any size and alignment; pairs must not overlap. */
typedef struct curv_copy_desc {
  void* dst;
  const void* src;
  unsigned long long bytes;
} curv_copy_desc;
int curv_copy_batched(void* stream, const curv_copy_desc* descs, int n);

/* ------------------------------------------------------------------------------------------------
 * utils.get_eigenvectors (curvature/utils.py:45-60): symmetric eigendecomposition F = U diag(w) U^T of a
 * batch of fp32 matrices by a two-sided block-Jacobi method in fp64.  U (n x n, fp32) holds the
 * eigenvectors as columns in ascending eigenvalue order (the reference's symeig order); w (n, fp32,
 * optional) the eigenvalues of F (the reference decomposes F + F^T: same vectors, doubled values, and it
 * discards the values).  Signs / bases of degenerate clusters are arbitrary, as with LAPACK.
 * The call synchronises the stream once per sweep to test convergence (off(A) <= tol * ||A||_F);
 * max_sweeps <= 0 selects 60; tol <= 0 selects the size-dependent default: 1e-8 for n <= 1024, 5e-6 above (an fp32 phase
 * followed by one fp64 re-orthogonalisation; the reference's fp32 LAPACK reaches 1e-5 on such factors); a positive tol
 * applies to every matrix of the call.  n <= 8192.  If the iteration has not
 * converged after max_sweeps sweeps the outputs hold the last iterate and the call returns
 * CURV_ERR_NOT_CONVERGED (curv_last_error() carries the final off-norm ratio); *sweeps_done is the
 * number of sweeps executed either way.
 * ---------------------------------------------------------------------------------------------- */
typedef struct curv_eigh_desc {
  const float* F;
  float* U;
  float* w;
  int32_t n;
  int32_t reserved;
} curv_eigh_desc;

size_t curv_syevd_workspace_bytes(const curv_eigh_desc* descs, int n_mats);
int curv_syevd(void* stream, const curv_eigh_desc* descs, int n_mats, void* workspace, size_t workspace_bytes,
               int max_sweeps, double tol, int* sweeps_done);
/* With the defaults (max_sweeps <= 0 and tol <= 0) a matrix at least 2048 wide whose numerical rank is below half its width
 * - the Kronecker factor of a layer with more rows than samples went into it (a 4608-wide ResNet-50 factor at N = 32: rank
 * 1568) - is decomposed through its RANGE (csrc/eigh_lowrank.hip, ABI 10; round 5 did this in the Python mirror only): Gaussian range
 * finder, rank from a pivoted Cholesky of its Gram matrix, Cholesky-QR, the iteration on the k x k projected matrix only, an
 * orthonormal basis of the complement for the zero eigenvalues; accepted iff ||F - P F P|| <= 3e-6 ||F||, otherwise the matrix
 * takes the iteration on the whole matrix like all others.  Same outputs, same order.  The workspace size above includes
 * what the projection needs (about 1.2 GB per 4608-wide matrix).  CURV_EIGH_LOWRANK=0 in the environment, an explicit
 * max_sweeps or an explicit tol select the plain iteration.
 * curv_syevd_ex: the same call; `ranks` (optional host array, n_mats ints) receives per matrix the rank k of the projected
 * problem, or 0 for a matrix that went through the plain iteration. */
int curv_syevd_ex(void* stream, const curv_eigh_desc* descs, int n_mats, void* workspace, size_t workspace_bytes,
                  int max_sweeps, double tol, int* sweeps_done, int* ranks);

/* ------------------------------------------------------------------------------------------------
 * INF (sparse information form), curvature/curvatures.py:463-672
 * ---------------------------------------------------------------------------------------------- */
/* _dim_reduction's index work (:617-634): indices of the `rank` largest |lambda_vec| (length n*m, index
 * i*m + j) -> I = unique(idx / m), J = unique(idx % m), ascending int64, and counts = {|I|, |J|}.
 * Exact integer results; ties at the threshold magnitude are broken deterministically. n, m <= 8192. */
typedef struct curv_select_desc {
  const float* lambda_vec;
  int64_t* I;
  int64_t* J;
  int32_t* counts;
  int32_t n, m, rank, reserved;
} curv_select_desc;
int curv_inf_select(void* stream, const curv_select_desc* descs, int n_desc);

/* out[p][i*a + k] = U[p][i] * U[p][k]: rows of the Khatri-Rao square in the closed form of V_s^T V_s
 * (pre_sampler :556-564 without the (n m) x (a b) Kronecker matrix). U is n x a with row stride u_rs. */
int curv_colpairs(void* stream, const float* U, int n, int a, long long u_row_stride, float* out);
/* the same in fp64 (exact products of fp32 values), and out = (double) v^2: the closed form of V_s^T V_s is evaluated
 * in fp64 end to end (curv_gemm_f64_batched), because V_s^T V_s is ill-conditioned for strongly varying r (e.g.
 * invert(1, 1000)) and an fp32 evaluation limits P_c to ~1e-3 there */
int curv_colpairs_f64(void* stream, const float* U, int n, int a, long long u_row_stride, double* out);
int curv_square_f64(void* stream, const float* v, double* out, long long count);
/* vtv[(i,j),(k,l)] = (w + w^T)/2, w = sigma_ij sigma_kl V4[(i,k),(j,l)]   (:564-565) */
int curv_inf_vtv_assemble(void* stream, const float* V4, const float* sigma, int a, int b, float* vtv);
int curv_inf_vtv_assemble_f64(void* stream, const double* V4, const float* sigma, int a, int b, double* vtv);
/* Packed forms: the column pairs are symmetric in (i, k), so out[p][t(i,k)] = U[p][i] U[p][k] for i <= k only,
 * t(i,k) = i a - i (i - 1) / 2 + (k - i), a (a + 1) / 2 columns; V4p = PAp^T r^2 PGp is then
 * (a (a + 1) / 2) x (b (b + 1) / 2) and vtv[(i,j),(k,l)] = sigma_ij sigma_kl V4p[t_a(i,k)][t_b(j,l)] - half / a quarter
 * of the flops of the two products of the closed form, same values (the symmetrisation of :565 is exact here). */
int curv_colpairs_sym_f64(void* stream, const float* U, int n, int a, long long u_row_stride, double* out);
int curv_inf_vtv_assemble_sym_f64(void* stream, const double* V4p, const float* sigma, int a, int b, double* vtv);
/* dst[i][j] = src[i][j] * dl[i] * dr[j] (src fp32 or fp64, dst fp32): P_c = diag(s) L_c diag(s) (:570) */
int curv_diag_scale(void* stream, const void* src, int src_is_f64, float* dst, const float* dl, const float* dr,
                    int rows, int cols);
/* out[r][c] = src[(row_index ? row_index[r] : r) * src_rs + (col_index ? col_index[c] : c) * src_cs], out dense
 * (rows x cols): the `.t().flatten()` views of _dim_reduction (:617-618) and its `[:, I]` / `[I x J]` selections
 * (:636-645) with the int64 index lists curv_inf_select produced, without leaving the device. */
int curv_gather2d(void* stream, const float* src, long long src_rs, long long src_cs, const int64_t* row_index,
                  const int64_t* col_index, float* out, int rows, int cols);
/* Kronecker product with the reference's index convention (curvature/utils.py:288-310):
 * out[(i*br + k)][(j*bc + l)] = a[i][j] * b[k][l]; a is ar x ac, b is br x bc, out (ar*br) x (ac*bc), dense
 * row-major.  The estimators never form it (pre_sampler's closed form); exported for `utils.kron`. */
int curv_kron(void* stream, const float* a, int ar, int ac, const float* b, int br, int bc, float* out);
/* out[i][j] = A(i,j) * B(i,j) for strided 2-D views (EFB.sample's z * inv^T, :458) */
int curv_mul2d(void* stream, const float* A, long long a_rs, long long a_cs, const float* B, long long b_rs,
               long long b_cs, float* out, int rows, int cols);

/* ------------------------------------------------------------------------------------------------
 * The one collective of the layer-sharded estimators (reference: per-layer independence,
 * curvature/curvatures.py:20-21, sample_and_replace :117-129): every rank has written the sampled parameters of its
 * own layers into `flat` (the parameters of all layers in one vector, grouped by owning rank: rank k's segment is
 * [displs[k], displs[k] + counts[k]) in floats); after the call every rank holds every segment.  Variable counts,
 * nothing padded to the largest shard: one ncclBroadcast per rank in one RCCL group call on `stream`.
 * `comm` is an ncclComm_t (RCCL) of the caller; counts / displs have ncclCommCount(comm) entries and must agree on all
 * ranks.  RCCL is bound at run time (dlopen): the library has no link-time dependency on it.
 * curv_comm_unique_id / curv_comm_init / curv_comm_destroy: thin wrappers of ncclGetUniqueId / ncclCommInitRank /
 * ncclCommDestroy for callers without RCCL bindings (`id` is the 128-byte ncclUniqueId drawn on rank 0 and shipped to
 * the other ranks by the caller; the calling thread's current device is the rank's device).
 * curv_rccl_available: 1 when RCCL could be bound (dlopen and every symbol), 0 otherwise - local, no communicator, no
 * bootstrap socket: the probe to run on every rank before the collective set-up steps.
 * ---------------------------------------------------------------------------------------------- */
int curv_rccl_available(void);
int curv_allgather_weights(void* comm, void* stream, float* flat, const long long* counts, const long long* displs);
int curv_comm_unique_id(void* id_out_128_bytes);
int curv_comm_init(void** comm_out, int n_ranks, const void* id_128_bytes, int rank);
int curv_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* CURV_HIP_H */
