/*
 * unet_hip.h -- C ABI of libunet_hip.so: hand-written CDNA4 (gfx950) kernels for the
 * CondUNet1D hot path of AliZareh-CoE/Unet.
 *
 * The reference has no FFI: its "operator API" is torch.nn calls made from models.py.
 * Each entry point below therefore cites the reference call site(s) it replaces.  The
 * Python host (unet_amd/ops.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     the name ends in _host
 *   - activations are channels-last  [n][rows][ld]  ("NLC"), ld >= channels
 *   - dtype: UNET_BF16 = bf16 storage, bf16 MFMA, fp32 accumulate;
 *            UNET_F32  = fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)
 *   - the library never allocates or frees device memory, never synchronises the
 *     device, keeps no pointer after return, and launches only on `stream`
 *   - return 0 on success; otherwise a negative code and unet_last_error() (thread
 *     local) holds a message.  No C++ exception crosses the boundary.
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_ABI_VERSION 1
#define UNET_MAX_TAPS 32

enum { UNET_BF16 = 0, UNET_F32 = 1 };
enum { UNET_ACT_NONE = 0, UNET_ACT_RELU = 1, UNET_ACT_LEAKY = 2, UNET_ACT_GELU = 3,
       UNET_ACT_SILU = 4, UNET_ACT_MISH = 5 };  /* models.py:59-65 get_activation */

int unet_abi_version(void);
const char* unet_last_error(void);

/* ------------------------------------------------------------------------------------
 * Generic tap-list convolution (implicit GEMM on MFMA).  One descriptor covers
 *   Conv1d k=1/3 stride 1/2                     models.py:558,3032,3038,3043,492,495,605,910
 *   4-branch dilated k=7 conv, taps merged       models.py:692-699,752-759
 *   ConvTranspose1d k=4 s=2 p=1 (two phases)     models.py:592,881
 *   and the data-gradients of all of them (transposed weights, mirrored taps).
 *
 *   y[n, up*m + phase_j, co] = bias[co] + res[n, up*m+phase_j, co]
 *        + sum_{taps j of that phase} sum_ci  W_j[co][ci] * x[n, stride*m + off_j, ci]
 *   x rows outside [0, l_in) read as zero.  x is the channel-concatenation of two
 *   sources (c2 may be 0): torch.cat([x, skip], 1) of models.py:624,930 is never built.
 * ---------------------------------------------------------------------------------- */
typedef struct UnetConvDesc {
  int32_t dtype;
  int32_t n;                    /* windows */
  int32_t l_in;                 /* valid input rows per window */
  int32_t c1, c2;               /* channels of source 1 / source 2 */
  int32_t x1_ld, x2_ld;         /* row pitch (elements) */
  int64_t x1_ns, x2_ns;         /* window stride (elements) */
  int32_t l_out, c_out;
  int32_t y_ld;  int64_t y_ns;
  int32_t res_ld; int64_t res_ns;
  int32_t stride, up;
  int32_t n_taps;               /* taps must be grouped by phase, phase 0 first */
  int32_t tap_off[UNET_MAX_TAPS];
  int32_t tap_phase[UNET_MAX_TAPS];
  int32_t act;                  /* epilogue activation (UNET_ACT_*) */
} UnetConvDesc;

/* bytes of the packed weight image unet_conv1d_fwd expects for (dtype, n_taps, rows=c_out, c1, c2) */
size_t unet_conv1d_packed_bytes(int32_t dtype, int32_t n_taps, int32_t rows, int32_t c1, int32_t c2);

/* Pack fp32 weights w[tap][row][col] (arbitrary element strides) into the tile-major, LDS-image
 * layout the MFMA kernels stream with global_load_lds.  Packed tap t takes logical tap
 * tap_src_host[t]; packed rows are logical rows [row_lo, row_lo+rows); packed K is
 * cols [col_lo, col_lo+c1) then [col_lo+c1, col_lo+c1+c2), each zero-padded to the K-chunk.
 * Forward packs (row=co, col=ci); data-grad packs the transpose (row=ci, col=co) via strides. */
int unet_conv1d_pack(int32_t dtype, const float* w, int64_t tap_stride, int64_t row_stride,
                     int64_t col_stride, int32_t n_taps, const int32_t* tap_src_host,
                     int32_t row_lo, int32_t rows, int32_t col_lo, int32_t c1, int32_t c2,
                     void* packed, void* stream);

/* Up to UNET_PACK_MAX_JOBS images in one launch, each described as a unet_conv1d_pack call: a training step packs a
 * conv's forward image together with the transposed image(s) its data gradient will read (the weights are the same
 * until the optimizer runs), which halves the pack launches of a step.  has_tap_src = 0: taps in order. */
#define UNET_PACK_MAX_JOBS 4
typedef struct UnetPackJob {
  const float* w;
  int64_t tap_stride, row_stride, col_stride;
  int32_t n_taps, has_tap_src;
  int32_t tap_src[UNET_MAX_TAPS];
  int32_t row_lo, rows, col_lo, c1, c2;
  void* packed;
} UnetPackJob;
int unet_conv1d_pack_multi(int32_t dtype, const UnetPackJob* jobs, int32_t n_jobs, void* stream);

int unet_conv1d_fwd(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed,
                    const float* bias, const void* res, void* y, void* stream);

/* unet_conv1d_fwd that also emits the batch statistics of its output, so that the BatchNorm which follows
 * (models.py:801-817) needs no pass of its own over y: s1_part / s2_part [n][*parts_out][c_out] fp32 = sum and sum of
 * squares of the STORED values (after bias / residual / activation, rounded to the output dtype) over the positions
 * of one tile.  Buffers must hold unet_conv1d_stats_max_parts(d) * n * c_out floats each.  *parts_out = 0 means the
 * kernel chosen for this conv does not produce them (generic / fp32 / streaming / transposed): y is complete, the
 * caller computes the statistics with unet_rowstats. */
int32_t unet_conv1d_stats_max_parts(const UnetConvDesc* d);
int unet_conv1d_fwd_stats(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed,
                          const float* bias, const void* res, void* y, float* s1_part, float* s2_part,
                          int32_t* parts_out, void* stream);

/* FiLM's out_proj with the gate and the residual as its epilogue (models.py:515-517, `x + gate * out_proj(h)`):
 *   y_o   = conv(x1) + bias                     (stored only when y_o is non-null: training keeps o for d gate)
 *   y_out = res + gate[n][c] * y_o              (gate fp32 [n][c_out], 16-byte aligned; y_o as rounded to bf16)
 * one pass instead of unet_conv1d_fwd followed by unet_gated_add_fwd, which reads o and res back and writes out.
 * Single-tap, unit-stride bf16 convs that the streaming kernel takes only (kctot >= 3, c_out % 8 == 0, c_out <= 2048,
 * 16-byte aligned rows); y_out has y's pitches (d->y_ld / d->y_ns).  Returns 0 when launched, 1 when the shape is not
 * taken and NOTHING was launched (the caller then runs the two calls above), -1 on error. */
int unet_conv1d_fwd_gated(const UnetConvDesc* d, const void* x1, const void* w_packed, const float* bias,
                          const void* res, const float* gate, void* y_o, void* y_out, void* stream);

/* Which kernel the calling thread's most recent unet_conv1d_fwd / unet_conv1d_wgrad launch used.  No
 * counterpart in the reference (cuDNN picks its algorithm internally, models.py convs); bench.py uses it
 * to attribute live launch timings to the kernel named in its roofline object. */
enum { UNET_VARIANT_NONE = 0,
       UNET_VARIANT_FWD_GENERIC = 1,   /* conv_fwd_kernel: fp32 (exact 16x16x4 MFMA) or unaligned bf16 */
       UNET_VARIANT_FWD_TILED = 2,     /* conv_fwd2_kernel_192/_128/_64: bf16, buffer LDS-DMA, halo tile + tap loop */
       UNET_VARIANT_FWD_STREAM = 3,    /* conv_stream_kernel: bf16 single-tap convs, persistent, 256x128 tiles */
       UNET_VARIANT_WGRAD_GENERIC = 11,/* conv_wgrad_kernel */
       UNET_VARIANT_WGRAD_PAIRED = 12, /* conv_wgrad2_kernel: bf16, two taps per block, LDS-DMA ring */
       UNET_VARIANT_WGRAD_WIDE = 13    /* conv_wgrad3_kernel: bf16 single-tap convs, >= 512 channels, 256x256 tiles */ };
int unet_conv1d_last_variant(void);

/* Weight gradient:  dw[j][co][ci] (fp32, contiguous, ci over c1+c2) =
 *   sum_n sum_m dy[n, up*m+phase_j, co] * x[n, stride*m+off_j, ci]
 * d describes the FORWARD conv (dy has d->l_out rows, d->c_out channels, pitch y_ld/y_ns).
 * workspace: split-K slabs, unet_conv1d_wgrad_workspace_bytes(d) bytes. */
size_t unet_conv1d_wgrad_workspace_bytes(const UnetConvDesc* d);
int unet_conv1d_wgrad(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy,
                      float* dw, void* workspace, size_t ws_bytes, void* stream);

/* Allocation-free variant of unet_conv1d_wgrad.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory around the call (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  Same arguments and result as unet_conv1d_wgrad, but the split-K slabs live in a device
 * array built into the library instead of a caller-provided workspace: UNET_NOALLOC_SCRATCH_BYTES per GPU the
 * library runs on, held until the process ends.  Where the slabs unet_conv1d_wgrad would use fit the array, the
 * kernel and the split count are the same and dw is bit-identical.  One exception: a single-tap conv whose
 * 256 x 256-tile kernel (conv_wgrad3) needs more slabs than fit while the paired kernel's fit (e.g. C = 256, 1x1,
 * 256 position chunks: 64 MiB against 16) runs the paired kernel here -- same result to fp32 summation order, not bit-identical.
 * A conv whose slabs exceed the array under either kernel is rejected with an error.  The array is the one of the
 * CURRENT device (hipGetDevice), which must be the device that owns `stream`.  The call
 * allocates and frees nothing (no hipMalloc/hipFree, no allocating library routine).  All calls on one GPU share
 * that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls ordered on one stream
 * do not, and nothing needs clearing between them (every slab element read is written by the same call). */
#define UNET_NOALLOC_SCRATCH_BYTES (32u << 20)
size_t unet_conv1d_wgrad_noalloc_max_bytes(void);
int unet_conv1d_wgrad_noalloc(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy,
                              float* dw, void* stream);

/* ------------------------------------------------------------------------------------
 * Channels-last streaming kernels (HBM-bound).  "part" outputs are per-row-chunk partial
 * sums  [n][UNET_PARTS(rows)][c]  fp32; the caller adds them over the middle axis (tiny).
 * ---------------------------------------------------------------------------------- */
int32_t unet_parts(int32_t rows);   /* number of row chunks used for partial sums */

/* models.py:3235 z-score statistics over T of an NCL tensor x[n][c][t] (dtype in_dtype):
 * mean[n][c] and rstd[n][c] = 1/(std_unbiased + eps), fp32 (also the stats of
 * SessionAdaptiveScaling, models.py:2058-2059, and per_channel_normalize, train.py:476-495). */
int unet_ncl_moments(int32_t in_dtype, const void* x, float* mean, float* rstd, int32_t n, int32_t c,
                     int32_t t, float eps, void* stream);
/* layout change into the kernels' channels-last form, with the z-score applied on the way:
 * y[n][t][c] (dtype) = (x[n][c][t] - sub[n][c]) * mul[n][c]   (sub/mul NULL = plain transpose) */
int unet_ncl_to_nlc(int32_t in_dtype, int32_t dtype, const void* x, const float* sub, const float* mul,
                    void* y, int32_t n, int32_t c, int32_t t, void* stream);
/* and back out at the model boundary (models.py:3363-3370: residual already added by the outc
 * epilogue; ga/be carry SessionAdaptiveScaling's gamma/beta, models.py:2068):
 * y[n][c][t] (out_dtype) = ga[n][c]*x[n][t][c] + gb[n][c]*x2[n][t][c] + be[n][c]
 * ga/be NULL = identity; x2/gb NULL = absent (x2 is used by the z-score backward). */
int unet_nlc_to_ncl(int32_t dtype, int32_t out_dtype, const void* x, const void* x2, const float* ga,
                    const float* gb, const float* be, void* y, int32_t n, int32_t c, int32_t t,
                    void* stream);

/* per-(window, channel) sums for BatchNorm/InstanceNorm/GroupNorm statistics
 * (models.py:537-555, 801-817):  s1 = sum_t y, s2 = sum_t y^2 */
int unet_rowstats(int32_t dtype, const void* y, float* s1_part, float* s2_part, int32_t n,
                  int32_t rows, int32_t c, void* stream);

/* BatchNorm1d training-mode statistics (models.py:540, 804; torch defaults eps 1e-5, momentum 0.1):
 * from the unet_rowstats partials compute, per channel, mean / biased var / rstd, update the running
 * buffers in place (unbiased var; may be NULL), and emit mean/rstd/a/b as [n][c] fp32 with
 * a = gamma*rstd, b = beta - mean*a  (gamma/beta NULL = 1/0).  batches_tracked (may be NULL): BatchNorm1d's
 * num_batches_tracked buffer, incremented by one (torch does it in F.batch_norm's Python wrapper). */
int unet_bn_finalize_fwd(const float* s1_part, const float* s2_part, int32_t n, int32_t parts, int32_t c,
                         int32_t rows, float eps, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float* mean,
                         float* rstd, float* a, float* b, int64_t* batches_tracked, void* stream);
/* ... and its backward: from the unet_affine_act_bwd_reduce partials compute dgamma/dbeta [c] and the
 * coefficients ca/cb/cc [n][c] of unet_affine_act_bwd_apply. */
int unet_bn_finalize_bwd(const float* g1_part, const float* g2_part, int32_t n, int32_t parts, int32_t c,
                         int32_t rows, const float* mean, const float* rstd, const float* gamma, float* ca,
                         float* cb, float* cc, float* dgamma, float* dbeta, void* stream);
/* out[c] = sum_r in[r][c]  (fp64 accumulation; second stage of the column-sum partials) */
int unet_reduce_rows(const float* in, int32_t rows, int32_t c, float* out, void* stream);
/* out[n][c] = sum_p in[n][p][c]: adds the "part" partials of each window */
int unet_reduce_parts(const float* in, int32_t n, int32_t parts, int32_t c, float* out, void* stream);
/* Up to UNET_REDUCE_MAX_JOBS of the two reductions above in ONE launch (they are launch-bound; FiLM's backward,
 * models.py:504-517, ends in five): for every job, out[u][c] = sum_r in[u][r][c], u < units, fp64 accumulation.
 * unet_reduce_rows is a job with units = 1, unet_reduce_parts one with units = n and rows = parts. */
#define UNET_REDUCE_MAX_JOBS 8
typedef struct { const float* in; float* out; int32_t units, rows, c; } UnetReduceJob;
int unet_reduce_multi(const UnetReduceJob* jobs, int32_t n_jobs, void* stream);

/* z = act(a[n][c]*y + b[n][c]); pool_part = partial sum_t z  (norm-apply + GELU + SE squeeze,
 * models.py:559-560, 827, 649/660).  pool_part may be NULL. */
int unet_affine_act_fwd(int32_t dtype, const void* y, const float* a, const float* b, int32_t act,
                        void* z, float* pool_part, int32_t n, int32_t rows, int32_t c, void* stream);
/* g = (dz + dpool[n][c]) * act'(a*y+b);  g1 = sum_t g, g2 = sum_t g*y  (partials) */
int unet_affine_act_bwd_reduce(int32_t dtype, const void* y, const void* dz, const float* dpool,
                               const float* a, const float* b, int32_t act, float* g1_part,
                               float* g2_part, int32_t n, int32_t rows, int32_t c, void* stream);
/* dy = ca[n][c]*g + cb[n][c]*y + cc[n][c] */
int unet_affine_act_bwd_apply(int32_t dtype, const void* y, const void* dz, const float* dpool,
                              const float* a, const float* b, int32_t act, const float* ca,
                              const float* cb, const float* cc, void* dy, int32_t n, int32_t rows,
                              int32_t c, void* stream);

/* ... the same, also emitting dy_sum_part = partial column sums of dy: the bias gradient of the conv that produced y
 * (models.py:558, 692-699: nn.Conv1d bias), which otherwise costs a pass of its own over dy (unet_colsum).
 * dy_sum_part NULL = plain unet_affine_act_bwd_apply. */
int unet_affine_act_bwd_apply_sum(int32_t dtype, const void* y, const void* dz, const float* dpool,
                                  const float* a, const float* b, int32_t act, const float* ca,
                                  const float* cb, const float* cc, void* dy, float* dy_sum_part, int32_t n,
                                  int32_t rows, int32_t c, void* stream);

/* Norm + activation + squeeze-excite as one unit (models.py:801-839: the norm, nn.GELU and SqueezeExcitation1D of a
 * ModernConvBlock; SE itself models.py:637-662).  z = act(a*y+b) * s[n][c], s = SE-MLP(mean_t act(a*y+b)):
 *   unet_affine_pool            pool_part = partial sums over rows of act(a*y+b); nothing else is stored
 *   unet_affine_act_scale_fwd   z = act(a*y+b) * s
 *   unet_affine_act_se_bwd_reduce   ONE pass over (y, dz): p_part[k][n][part][c], k = 0..4:
 *                               sum dz*act, sum dz*act', sum dz*act'*y, sum act', sum act'*y   (p_part: 5*n*parts*c floats)
 *   unet_se_combine             on the part-reduced sums p[5][n][c]:  g1 = s*p1 + dpool*p3, g2 = s*p2 + dpool*p4
 *                               (= sum_t g, sum_t g*y of unet_affine_act_bwd_reduce; p0 = d s goes through the SE MLP
 *                               backward to give dpool first; dpool may be NULL)
 *   unet_affine_act_se_bwd_apply    dy = ca*g + cb*y + cc with g = (dz*s + dpool)*act'(a*y+b), + column sums of dy */
int unet_affine_pool(int32_t dtype, const void* y, const float* a, const float* b, int32_t act, float* pool_part,
                     int32_t n, int32_t rows, int32_t c, void* stream);
int unet_affine_act_scale_fwd(int32_t dtype, const void* y, const float* a, const float* b, int32_t act, const float* s,
                              void* z, int32_t n, int32_t rows, int32_t c, void* stream);
int unet_affine_act_se_bwd_reduce(int32_t dtype, const void* y, const void* dz, const float* a, const float* b,
                                  int32_t act, float* p_part, int32_t n, int32_t rows, int32_t c, void* stream);
int unet_se_combine(const float* p, const float* s, const float* dpool, float* g1, float* g2, int32_t n, int32_t c,
                    void* stream);
int unet_affine_act_se_bwd_apply(int32_t dtype, const void* y, const void* dz, const float* dpool, const float* s,
                                 const float* a, const float* b, int32_t act, const float* ca, const float* cb,
                                 const float* cc, void* dy, float* dy_sum_part, int32_t n, int32_t rows, int32_t c,
                                 void* stream);

/* out = x * s[n][c]  (SE excite, models.py:662);  bwd: dx = dout*s, ds_part = sum_t dout*x */
int unet_scale_fwd(int32_t dtype, const void* x, const float* s, void* out, int32_t n, int32_t rows,
                   int32_t c, void* stream);
int unet_scale_bwd(int32_t dtype, const void* x, const void* dout, const float* s, void* dx,
                   float* ds_part, int32_t n, int32_t rows, int32_t c, void* stream);

/* FiLM channel softmax (models.py:514-515): h[n,t,:] = softmax_c(q[n,t,:]*kk[n,:]) * v[n,:] */
int unet_film_softmax_fwd(int32_t dtype, const void* q, const float* kk, const float* v, void* h,
                          int32_t n, int32_t rows, int32_t c, void* stream);
/* dq_sum_part (nullable): partial column sums of dq = the bias gradient of q_proj, taken in the same pass for
 * rows of up to 512 channels (a second pass over dq beyond that) */
int unet_film_softmax_bwd(int32_t dtype, const void* q, const void* dh, const float* kk,
                          const float* v, void* dq, float* dkk_part, float* dv_part, float* dq_sum_part,
                          int32_t n, int32_t rows, int32_t c, void* stream);

/* out = res + g[n][c]*o  (models.py:517);  bwd: do = g*dout, dg_part = sum_t dout*o; do_sum_part (nullable):
 * partial column sums of do = the bias gradient of out_proj, taken in the same pass */
int unet_gated_add_fwd(int32_t dtype, const void* res, const void* o, const float* g, void* out,
                       int32_t n, int32_t rows, int32_t c, void* stream);
int unet_gated_add_bwd(int32_t dtype, const void* o, const void* dout, const float* g, void* d_o,
                       float* dg_part, float* do_sum_part, int32_t n, int32_t rows, int32_t c, void* stream);

/* column sums over all rows of all windows: out_part[n][parts][c] = partial sum_t x  (bias grads) */
int unet_colsum(int32_t dtype, const void* x, float* out_part, int32_t n, int32_t rows, int32_t c,
                void* stream);

/* Merged-tap weights of the 4-branch dilated conv (models.py:752-759):
 *   w_eff[t][co][ci] = sum over (b,k) with tap_index_host[b*k_size+k] == t of sw[b] * w_b[co][ci][k]
 * w_branch_host: host array of n_branch DEVICE pointers to contiguous [co][ci][k] fp32 weights; sw: device [n_branch]
 * (already softmaxed).  bwd: dw_b[co][ci][k] = sw[b]*dw_eff[t(b,k)][co][ci] and per-block partials of
 * d sw[b] = <dw_eff[t(b,k)], w_b[..,k]> in dsw_part[unet_merge_taps_parts(co,ci)][n_branch]. */
int32_t unet_merge_taps_parts(int32_t co, int32_t ci);
int unet_merge_taps_fwd(const float* const* w_branch_host, const float* sw, int32_t n_branch, int32_t k, int32_t co,
                        int32_t ci, const int32_t* tap_index_host, int32_t n_taps, float* w_eff, void* stream);
int unet_merge_taps_bwd(const float* const* w_branch_host, const float* sw, int32_t n_branch, int32_t k, int32_t co,
                        int32_t ci, const int32_t* tap_index_host, int32_t n_taps, const float* dw_eff,
                        float* const* dw_branch_host, float* dsw_part, void* stream);

/* The whole weight algebra of MultiScaleDilatedConv1d.forward (models.py:752-759) in one launch forward and two
 * backward: w_eff as above, b_eff[co] = sum_b s[b] * bias_b[co], with s = softmax(sw) when `softmax` != 0 (sw = the
 * raw scale_weights parameter) and s = sw otherwise.  sw_out (nullable): the branch weights s that were used, which
 * is what the backward call takes as `sw`.  bwd: additionally d bias_b = s[b] * db_eff, and dsw (nullable) =
 * the sum of dsw_part (incl. the bias term) taken back through the softmax when `softmax` != 0. */
int unet_merge_branches_fwd(const float* const* w_branch_host, const float* const* bias_branch_host, const float* sw,
                            int32_t softmax, int32_t n_branch, int32_t k, int32_t co, int32_t ci,
                            const int32_t* tap_index_host, int32_t n_taps, float* w_eff, float* b_eff, float* sw_out,
                            void* stream);
int unet_merge_branches_bwd(const float* const* w_branch_host, const float* const* bias_branch_host, const float* sw,
                            int32_t softmax, int32_t n_branch, int32_t k, int32_t co, int32_t ci,
                            const int32_t* tap_index_host, int32_t n_taps, const float* dw_eff, const float* db_eff,
                            float* const* dw_branch_host, float* const* dbias_branch_host, float* dsw_part, float* dsw,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Per-window vector algebra (csrc/small.hip): the Linear layers on [windows][channels] vectors -- FiLM's k_proj /
 * v_proj / gate (models.py:493-494, 497-500, 506-509) and the SqueezeExcitation MLP (models.py:650-655, 660-661).
 * A group of up to UNET_SMALL_MAX maps is one launch each way (a BLAS call + elementwise ops per map were ~45
 * launches per block and step).  All tensors fp32, row-major: x [n][n_in], w [n_out][n_in], y / pre / dy [n][n_out].
 *   fwd:  pre = scale * (x w^T + b) ; y = act(pre)          (b, pre nullable)
 *   bwd:  dpre = dy * act'(.) * scale  (GELU' from pre, sigmoid' from y) ;
 *         dw [n_out][n_in] = dpre^T x ; db [n_out] = sum_n dpre ;
 *         dx_part [slot][n][n_in], slot = dx_slot0 + (output tile index): the input gradient contributed by each
 *         16-output tile; the caller adds the slots (unet_reduce_rows) -- unet_small_linear_tiles(n_out) per problem.
 *         dw / db / dx_part nullable.
 * ---------------------------------------------------------------------------------- */
#define UNET_SMALL_MAX 32
enum { UNET_SMALL_NONE = 0, UNET_SMALL_GELU = 1, UNET_SMALL_SIGMOID = 2 };
typedef struct UnetSmallLinear {
  const float* x; const float* w; const float* b;
  float* y; float* pre;
  float* part;                  /* fwd, when unet_small_linear_splits(n_in) > 1: [splits][n][n_out] partial sums */
  int32_t n_in, n_out, act;
  float scale;
  const float* dy;
  float* dw; float* db; float* dx_part;
  int32_t dx_slot0;
} UnetSmallLinear;
int unet_small_linear_fwd(const UnetSmallLinear* problems_host, int32_t n_problems, int32_t n, void* stream);
int unet_small_linear_bwd(const UnetSmallLinear* problems_host, int32_t n_problems, int32_t n, void* stream);
int32_t unet_small_linear_tiles(int32_t n_out);
int32_t unet_small_linear_splits(int32_t n_in);

/* ------------------------------------------------------------------------------------
 * Loss / evaluation reductions (csrc/metrics.hip).  Inputs are NCL rows: row r starts at
 * base + r*ld (elements), time is the contiguous axis.  fp64 results, no host sync.
 * ---------------------------------------------------------------------------------- */

/* train.py:941-952 pooled statistics + models.py:4262-4287 / 4331-4362 per-row moments, over t in [lo,hi):
 * out_rows[r][8] = { n, S t, S t^2, S p, S p^2, S p*t, S (p-t)^2, S |p-t| }  (order of train.py:1076-1085) */
int unet_pair_moments(int32_t dtype, const void* pred, const void* target, int64_t rows, int64_t ld_pred,
                      int64_t ld_target, int32_t lo, int32_t hi, double* out_rows, void* stream);
/* out[k] = sum_r in[r][k], k < width <= 64: the cross-row totals of the above (deterministic) */
int unet_sum_rows_f64(const double* in, int64_t rows, int32_t width, double* out, void* stream);

/* train.py:1481-1486 backward of  w * F.l1_loss(crop(pred), crop(target)):
 * dpred[r][i] = gscale[0]*mul*sign(pred-target) for lo <= i < hi, else 0  (dense [rows][t], grad_dtype) */
int unet_l1_crop_bwd(int32_t dtype, int32_t grad_dtype, const void* pred, const void* target, int64_t rows,
                     int64_t ld_pred, int64_t ld_target, int32_t t, int32_t lo, int32_t hi, const float* gscale,
                     float mul, void* dpred, void* stream);

/* per-row z-score y = (x-mean)/den of dense rows [rows][t] in one HBM pass:
 * mode 0: den = std+eps (models.py:3235 unbiased=1; data.py:3121,4649 unbiased=0);
 * mode 1: den = max(std,eps) (train.py:476-495 per_channel_normalize, unbiased=1) */
int unet_row_zscore(int32_t in_dtype, int32_t out_dtype, const void* x, void* y, int64_t rows, int32_t t, float eps,
                    int32_t mode, int32_t unbiased, void* stream);

/* Welch PSD of x[r][lo:hi) (models.py:4512-4550; scipy.signal.welch of utils/dsp_metrics.py:163):
 * segments of nperseg (power of two, 128..2048), hop nperseg/2, periodic Hann, optional constant detrend;
 * psd[r][k] = scale_k * mean_seg |FFT|^2, k <= nperseg/2, scale_k = edge_scale at DC/Nyquist else scale. */
int unet_welch_psd(int32_t dtype, const void* x, int64_t rows, int64_t ld, int32_t lo, int32_t hi, int32_t nperseg,
                   int32_t detrend, float scale, float edge_scale, float* psd, void* stream);
/* models.py:4611-4627 / 4686-4695: out_rows[r] = { S_k |dB p - dB t|, S_k (dB p - dB t) } over bins [k_lo,k_hi] */
int unet_psd_db_stats(const float* psd_pred, const float* psd_target, int64_t rows, int32_t nfreq, int32_t k_lo,
                      int32_t k_hi, int32_t normalize, double* out_rows, void* stream);
/* band power per row over inclusive bin ranges (host arrays): mode 0 = sum (models.py:4808-4816),
 * mode 1 = trapezoid*df (np.trapz, utils/dsp_metrics.py:168-177).  out[r][n_bands] */
int unet_band_power(const float* psd, int64_t rows, int32_t nfreq, int32_t n_bands, const int32_t* lo_host,
                    const int32_t* hi_host, int32_t mode, float df, float* out, void* stream);

/* Log band power of a one-sided complex spectrum: out[row][k] = log(sum_f |X[row][f]|^2 * masks[k][f] + eps),
 * the einsum('bcf,kf->bck') + log of SpectroTemporalEncoder._spectral_forward (models.py:1294-1301).
 * spectrum: rows x nfreq interleaved (re, im) fp32 (torch.fft.rfft output); masks: n_bands x nfreq fp32
 * (models.py:1245-1275); n_bands <= 16. */
int unet_band_log_power(const float* spectrum, const float* masks, int64_t rows, int32_t nfreq, int32_t n_bands,
                        float eps, float* out, void* stream);

/* phase_five/sliding_window.py:238-251 overlap-average reconstruction.  pred[w][c][window_size] holds windows
 * first_window .. first_window+n_windows-1 of total_windows (window g starts at g*stride);
 * mode 0 "average": out[c][n_samples] (+)= their sum per sample (accumulate: add to what out holds, so
 * chunks / ranks contribute in turn; n_windows may be 0 for a finalize-only call); finalize: divide by
 * max(count,1), count = windows of the WHOLE recording covering the sample (written to counts if non-NULL).
 * mode 1 "last" / 2 "first": the sample is written only by the call holding its last / first covering
 * window, other samples are left untouched (zero-initialise out; chunk sums stay exact). */
int unet_overlap_add(int32_t dtype, const void* pred, int32_t n_windows, int32_t first_window, int32_t total_windows,
                     int32_t c, int32_t window_size, int32_t stride, int64_t n_samples, float* out, float* counts,
                     int32_t accumulate, int32_t finalize, int32_t mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Band-pair attention of CrossFrequencyCouplingAttentionV2 (csrc/attn.hip), bf16.
 *
 * unet_cfc_attn_fwd replaces the 25 F.scaled_dot_product_attention calls, their coupling-weighted fp32 sum and the
 * band concatenation of models.py:431-462 with one launch:
 *   out[n][l][i*c + h*hd + d] = sum_j coupling[i*n_bands + j] * softmax_keys(q_i k_j^T * scale) v_j     (hd = c / n_heads)
 * q, k, v: [n_bands][n][l][c] (band-major stack of the grouped projections, channel = head*hd + d);
 * coupling: n_bands^2 floats on the device (already softmaxed, models.py:428); out: [n][l][n_bands*c].
 * o_pairs [n_bands][n_bands][n][l][c] (the 25 attention outputs, bf16) and lse [n_bands][n_bands][n][n_heads][l]
 * (fp32, log2 domain) are what backward needs.  lse may be null (inference).  o_pairs may be null as well: the sum
 * over j is then kept in registers and no pair output is stored (slower: the second accumulator costs the registers
 * that keep LDS reads ahead of the MFMAs; and the sum is taken before instead of after the bf16 rounding of each
 * pair output, so the two modes differ in the last bf16 bit).  hd must be a multiple of 8, at most 256.
 *
 * unet_cfc_attn_bwd: gradients of the same expression.  dout: [n][l][n_bands*c]; dq, dk, dv: [n_bands][n][l][c];
 * dcoupling: n_bands^2 floats (gradient w.r.t. the softmaxed weights; overwritten); delta: scratch of lse's shape. */
int unet_cfc_attn_fwd(int32_t dtype, const void* q, const void* k, const void* v, const float* coupling, int32_t n,
                      int32_t l, int32_t c, int32_t n_heads, int32_t n_bands, float scale, void* out, void* o_pairs,
                      float* lse, void* stream);
/* dcoupling_part: unet_cfc_attn_bwd_partials(n, l, n_bands) floats laid out [n_bands^2][partials]; the caller sums the
 * last axis (no atomics: bit-reproducible).  Three launches: row dots <dO_i, O_ij>, dQ, dK/dV. */
size_t unet_cfc_attn_bwd_partials(int32_t n, int32_t l, int32_t n_bands);
int unet_cfc_attn_bwd(int32_t dtype, const void* q, const void* k, const void* v, const float* coupling,
                      const void* o_pairs, const float* lse, const void* dout, int32_t n, int32_t l, int32_t c,
                      int32_t n_heads, int32_t n_bands, float scale, void* dq, void* dk, void* dv, float* dcoupling_part,
                      float* delta, void* stream);

/* ------------------------------------------------------------------------------------
 * Data path in front of the model (csrc/data.hip).
 *
 * unet_gather_windows: out[b][ch][:] = src[ch][starts[b] : starts[b]+w], optionally z-scored per (window, channel)
 * row with numpy's biased std: (x - mean) / (std + eps) -- ECoGDataset.__getitem__ (data.py:4586-4612) and the
 * per-window normalisation of data.py:3120-3124 for a whole batch.  src: c rows of pitch src_ld elements holding
 * src_len valid samples (recordings concatenated along time); starts_dev: n_windows int64 on the device,
 * each in [0, src_len - w] (the caller builds them from the reference's window table, unet_amd/windows.py).
 *
 * unet_noise_augment: NoiseAugmentor.forward (models.py:4897-4946) after the random draws:
 *   y = (x + gauss*scales[0] + pink*scales[1]) * [channel_u > channel_p] * [time_u > time_p]
 *       + line_amp[b] * sin(2*pi*line_hz * t_i + line_phase[b]),   t_i = torch.linspace(0, t/sample_rate, t)[i] (fp32)
 * x, gauss, pink, y: [n][c][t] of `dtype`; any of gauss / pink / channel_u [n*c] / time_u [n*t] / line_amp [n]
 * may be null (that term is skipped); scales_dev: 2 floats on the device (no host read of the data's std). */
int unet_gather_windows(int32_t in_dtype, int32_t out_dtype, const void* src, int64_t src_ld, int64_t src_len,
                        const int64_t* starts_dev, int32_t n_windows, int32_t c, int32_t w, int32_t zscore,
                        float eps, void* out, void* stream);
int unet_noise_augment(int32_t dtype, const void* x, const void* gauss, const void* pink, const float* scales_dev,
                       const float* channel_u, float channel_p, const float* time_u, float time_p,
                       const float* line_amp, const float* line_phase, float line_hz, float sample_rate,
                       int32_t n, int32_t c, int32_t t, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step (csrc/optim.hip): torch.nn.utils.clip_grad_norm_(params, max_norm) followed by
 * torch.optim.AdamW(...).step() (train.py:1553,1559; AdamW arguments train.py:183-193,2214-2241), on the device end to
 * end (no host read of the norm; capturable in a HIP graph).
 *
 * The moments of a parameter group live in two flat fp32 buffers; tensor t owns elements [4*start4[t], 4*start4[t] +
 * numel[t]).  start4 (n_tensors + 1 entries, the last = end of the range), numel and p_ptrs are DEVICE arrays built once
 * by the caller; the gradient pointers are a HOST array (they change every step) of n_tensors <= UNET_OPT_MAX_TENSORS
 * entries, null = "no gradient this step" (that parameter is skipped, as torch does).  flat_elements = 4 * (start4[n] -
 * start4[0]).  Larger groups are processed as several calls over consecutive sub-tables, with their partial sums side
 * by side in one `partial` array (unet_opt_blocks(flat_elements) floats per call) and ONE prepare call over all of them.
 *
 *   unet_grad_sqnorm         partial[b] = sum of g^2 over block b's slice of the flat range; step_t[t] += 1 for every
 *                            tensor of the call that has a gradient (one fp32 counter per tensor, as torch keeps them)
 *   unet_clip_adamw_prepare  coef[0] = clip = min(1, max_norm / (||g|| + 1e-6)) (1 if max_norm <= 0);
 *                            coef[3] = ||g||;  coef[4] = lr (read from the device scalar *lr); coef holds 8 floats
 *   unet_clip_adamw_apply    g' = clip * g;  p *= 1 - lr * weight_decay;  m = beta1 m + (1 - beta1) g';
 *                            v = beta2 v + (1 - beta2) g'^2;
 *                            p -= lr / (1 - beta1^step_t) * m / (sqrt(v) / sqrt(1 - beta2^step_t) + eps)
 * The clipped gradient itself is not written back. */
#define UNET_OPT_MAX_TENSORS 256
int32_t unet_opt_blocks(int64_t flat_elements);
int unet_grad_sqnorm(const float* const* grads_host, int32_t n_tensors, const uint32_t* start4, const uint32_t* numel,
                     int64_t flat_elements, float* partial, float* step_t, void* stream);
int unet_clip_adamw_prepare(const float* partial, int32_t blocks, float max_norm, const float* lr, float* coef,
                            void* stream);
int unet_clip_adamw_apply(const float* const* grads_host, int32_t n_tensors, float* const* p_ptrs,
                          const uint32_t* start4, const uint32_t* numel, int64_t flat_elements, float* m_flat,
                          float* v_flat, const float* coef, const float* step_t, float beta1, float beta2, float eps,
                          float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
