/*
 * diffma_hip.h -- C ABI of libdiffma_hip.so (MI355X / gfx950 kernels for the DiffMa hot path).
 *
 * The reference (wongzbb/DiffMa-Diffusion-Mamba) has no C ABI of its own: its hot path sits behind
 * Python operator functions imported by name from two un-vendored CUDA wheels
 *   block/mamba.py:11      from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, mamba_inner_fn
 *   block/mamba.py:13      from causal_conv1d import causal_conv1d_fn
 *   block/mamba.py:26-82   scan_permutation / merge_permutation / CrossScan / CrossMerge (token reindex)
 * Each entry point below names the reference interface it replaces.  The Python operator mirror
 * (diffma-diffusion-mamba_amd/selective_scan_interface.py) binds these with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every buffer (inputs, outputs, workspaces) is owned by the
 *     caller; the library allocates nothing and keeps no references.
 *   - all pointers are DEVICE pointers; strides are in ELEMENTS (not bytes), 64-bit.
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous, never synchronise, and are
 *     legal inside hipGraph capture.
 *   - return 0 on success, a negative dm_status on failure; dm_last_error() returns a thread-local
 *     message for the last failure.  Nothing throws across the ABI.
 *   - "token-major" ("channel-last") layout means the channel stride is 1:  x[b][l][d].  The kernels are
 *     written for that layout (one lane per channel, coalesced 256-B wave rows); DM_ERR_LAYOUT is returned
 *     for anything else and the host shim repacks.
 */
#ifndef DIFFMA_HIP_H
#define DIFFMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_ABI_VERSION 29

typedef enum {
    DM_OK = 0,
    DM_ERR_ARG = -1,     /* null pointer / bad size                        */
    DM_ERR_LAYOUT = -2,  /* stride pattern the kernel does not implement   */
    DM_ERR_DTYPE = -3,   /* unsupported dtype enum                         */
    DM_ERR_DSTATE = -4,  /* d_state not in the instantiated set            */
    DM_ERR_LAUNCH = -5   /* hipLaunchKernel reported an error              */
} dm_status;

typedef enum { DM_F32 = 0, DM_BF16 = 1, DM_F16 = 2 } dm_dtype;

enum {
    DM_FLAG_DELTA_SOFTPLUS = 1, /* delta = softplus(delta + bias)                     */
    DM_FLAG_SILU = 2,           /* conv: apply SiLU after the bias add                */
    DM_FLAG_A_SHARED = 8,       /* scan: A[d][n] is the same for every state n of a channel (Mamba-2 / SSD: one decay per
                                   head), so the kernels evaluate ONE exp per (channel, step) instead of dstate.  A is still
                                   passed as [dim][dstate]; the caller vouches for the property.                         */
    DM_FLAG_DOUT_PER_SEQ = 4,   /* scan bwd with row indices: dout is [nseq][row][d] (one gradient per direction,
                                   Mamba-2: the gated RMSNorm sits between the scan and the merge) instead of
                                   [batch_per_dir][row][d] shared by the directions  */
    DM_FLAG_SCAN_SEQUENTIAL = 16, /* scan fwd / bwd: take the sequential-in-time kernel whatever the launch size          */
    DM_FLAG_SCAN_CHUNKED = 32,  /* scan fwd / bwd: take the chunk-parallel (two-pass) kernel where it is instantiated;
                                   by default the library chooses by launch size (small launches are latency-bound)       */
    DM_FLAG_OUT_ACCUMULATE = 64, /* scan fwd with row indices: out[s][out_row_index[l]] += y (read-add-store) instead of = y.
                                   The caller walks the directions of the CrossMerge (block/mamba.py:59-69) with one launch
                                   each into ONE token-order buffer: direction 0 stores, the others accumulate, and the
                                   separate merge pass disappears.  One direction per launch (batch_per_dir = 0 or nseq),
                                   d_state 16, z, both index tables, DM_FLAG_DELTA_SOFTPLUS, no DM_FLAG_A_SHARED; the
                                   sequential kernel is taken whatever the launch size.                                   */
    DM_FLAG_DX_MERGED = 256,    /* dm_gather_conv1d_xproj_bwd: dx is ONE token-order buffer [batch][seqlen][dim] (dx_ss = its batch stride)
                                   that receives the sum over the ndir directions -- a workgroup walks the directions of a sample one
                                   after the other, direction 0 stores, the others read-add-store -- instead of ndir slabs for
                                   dm_token_merge to add; dw_partial / db_partial are then [batch][...] (one row per sample).
                                   Width 4, SiLU and row-index tables (the mixer's call pattern).                              */
    DM_FLAG_PARTIAL_COMPACT = 512, /* dm_gather_conv1d_xproj_bwd, with DM_FLAG_DX_MERGED, when dm_gather_conv1d_xproj_bwd_slab(args)
                                   returns R > 0: the caller sums only the first R rows of dw_partial / db_partial (one row per
                                   persistent workgroup stream); without the flag the slab form zero-fills the rows of the other
                                   samples so that a sum over all `batch` rows stays right.                                   */
    DM_FLAG_DELTA_ACTIVATED = 128 /* scan fwd / bwd: `delta` already holds softplus(raw + delta_bias) -- the producer applied it
                                   once per element (dm_dtproj_softplus_fwd) instead of every scan direction evaluating it in
                                   the forward AND in the backward.  Forward: delta is used as is, delta_bias is ignored.
                                   Backward: ddelta is still the gradient of the RAW value, d raw = d delta * (1 - exp(-delta)),
                                   and dbias_partial its per-sequence sums.  Built for the DiffMa mixer's call pattern: no z
                                   (the SiLU gate is applied once per token where the directions meet, dm_token_merge /
                                   dm_gate_bwd), both row-index tables, d_state 16.                                         */
};

/* ------------------------------------------------------------------------------------------------
 * The diffusion wrapper around the denoiser call of a training step (ABI 29): GaussianDiffusion.training_losses (reference
 * diffusion/gaussian_diffusion.py:715-789) for epsilon prediction + learned-range variance + MSE loss (create_diffusion("")).
 *   dm_q_sample         x_t_out = sqrt(abar_t) x_start + sqrt(1 - abar_t) noise                                  (:215-230)
 *   dm_training_loss    per sample: mse = mean (noise - eps)^2, vb = the variational-bound term in bits/dim (KL for t > 0, discretised
 *                       decoder NLL at t = 0; the mean prediction detached, :682-713, :752-766; diffusion_utils.py:10-88), loss = mse + vb,
 *                       and grad[b] = (d mse_b / d eps | d vb_b / d v), fp32 [batch][2 channels][hw]
 *   dm_training_loss_bwd  grad_out (model output dtype) = (g_eps[b] * grad[b, :C] | g_v[b] * grad[b, C:])
 * model_out: [batch][2 channels][hw] contiguous (eps | variance logits), dtype out_dtype; x_start, x_t, noise fp32 [batch][channels][hw];
 * t int64 [batch], PRECONDITION 0 <= t[b] < T (a timestep outside the range reads no table entry: that sample's x_t / mse / vb / loss /
 * grad come back NaN, which the training loop's non-finite guard then reports); tables fp32 [nrows][T] with the row numbers of sqrt(abar), sqrt(1 - abar), log(posterior variance, clipped),
 * log(beta), sqrt(1 / abar), sqrt(1 / abar - 1), posterior mean coefficients 1 and 2.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, channels, hw, T;
    int32_t out_dtype, _pad;
    int32_t row_sqrt_ac, row_sqrt_1mac, row_post_logvar, row_log_betas, row_sqrt_recip_ac, row_sqrt_recipm1_ac, row_coef1, row_coef2;
    const void *model_out;
    const float *x_start, *x_t, *noise;
    const int64_t *t;
    const float *tables;
    float *x_t_out;
    float *mse, *vb, *loss, *grad;
    const float *g_eps, *g_v;
    void *grad_out;
} dm_training_loss_args;

int dm_q_sample(const dm_training_loss_args *args, void *stream);
int dm_training_loss(const dm_training_loss_args *args, void *stream);
int dm_training_loss_bwd(const dm_training_loss_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The optimiser step of the training loop (ABI 29): AdamW + the EMA of the weights in ONE pass over the parameters.
 * Reference: train.py:153-166 (`torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0)`), train.py:259-264 (`opt.step()`,
 * `update_ema(ema, model.module)`), train.py:36-47 (ema = decay * ema + (1 - decay) * param).  Arithmetic of torch's fused AdamW
 * (decoupled weight decay, no amsgrad, no maximize), all tensors fp32:
 *      p -= lr wd p;  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g g;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 *      (each expression in double, rounded to fp32 on assignment)
 *      ema += (1 - ema_decay)(p - ema)                    (ema == NULL: no EMA for that tensor)
 * `tensors` is a table in DEVICE memory, one entry per parameter; `step` points to the tensor's own step counter (a 0-dim fp32
 * device tensor, torch's convention for fused / capturable optimisers): the call advances it by one and uses the new value as t.
 * Work decomposition: workgroup i handles elements [block_chunk[i] * C, +C) of tensor block_tensor[i], C = dm_adamw_chunk(); the
 * caller lists every chunk of every tensor once (two int32 device arrays of nblocks entries).  found_inf (device scalar or NULL):
 * a non-zero value leaves p, m, v and the counters untouched; the EMA then still takes its step towards the unchanged weights if
 * ema_on_skip != 0 and is left alone otherwise.  Two launches on `stream` (counters, then elements); graph-capturable.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float *p, *m, *v;
    const float *g;
    float *ema;
    float *step;
    int64_t n;
} dm_adamw_tensor;

typedef struct {
    const dm_adamw_tensor *tensors;
    const int32_t *block_tensor, *block_chunk;
    int32_t ntensors, nblocks;
    double lr, beta1, beta2, eps, weight_decay, ema_decay;   /* double, as torch keeps them next to the fp32 elements */
    const float *found_inf;
    int32_t ema_on_skip, _pad;
    float *nonfinite_out;      /* dm_grads_nonfinite only */
} dm_adamw_args;

int dm_adamw_chunk(void);
int dm_adamw_ema_step(const dm_adamw_args *args, void *stream);
/* *nonfinite_out = 1 if any element of any gradient in the table is Inf / NaN (the caller zeroes it first; nothing else of `args` but the
 * table and the block arrays is read): the device-side decision of the reference's `if not torch.isfinite(...)`: continue (train.py:254-256). */
int dm_grads_nonfinite(const dm_adamw_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Selective scan, forward.   Replaces selective_scan_cuda.fwd behind
 *   selective_scan_fn(u, delta, A, B, C, D, z, delta_bias, delta_softplus)   (block/mamba.py:11)
 * and the scan stage of mamba_inner_fn (call sites block/mamba.py:346-348).
 *
 *   delta' = softplus(delta + delta_bias[d])            (if DM_FLAG_DELTA_SOFTPLUS)
 *   h[n]   = exp(delta'*A[d][n]) * h[n] + delta'*B[s][l][n]*u        h[-1] = 0
 *   y      = sum_n C[s][l][n]*h[n] + D[d]*u ;   out = y * silu(z)    (z optional)
 *
 * Sequences: nseq = ndir * batch_per_dir.  u/delta/out/B/C are indexed by the sequence index s;
 * z is indexed by (s % batch_per_dir) and, when z_row_index != NULL, read at row
 * z_row_index[dir*seqlen + l]  (dir = s / batch_per_dir): the CrossScan gather of the z half is folded
 * into the load (block/mamba.py:32-45).  When out_row_index != NULL the result of step l is stored at
 * row out_row_index[dir*seqlen + l]: the CrossMerge inverse reindex is folded into the store
 * (block/mamba.py:59-69).  With ndir = 1 and both index pointers NULL this is the plain operator.
 *
 * ckpt (optional, training): the state after every ckpt_every (= 4) steps; slot c > 0 holds the state entering
 * step 4*c, slot 0 (the state entering step 0 is zero) the state after the last step.  ckpt_dtype DM_F32: fp32, layout [s][chunk][n][d] (fp32 / fp16 I/O);
 * DM_BF16: pairs of bf16 in one 32-bit word, layout [s][chunk][n/8][d][4] (16-byte accesses per lane, dense per
 * instruction), state 2k in the low half and 2k+1 in the high half of word k (bf16 I/O: the recomputed states inherit
 * the precision the I/O tensors already have).  The buffer is private to the forward / backward pair of one build.
 * last_state (optional): final h, fp32, layout [s][n][d].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nseq, dim, seqlen, dstate;
    int32_t ngroups;         /* B/C groups over the channel axis (DiffMa: 1)           */
    int32_t batch_per_dir;   /* nseq = ndir*batch_per_dir; 0 or nseq means ndir = 1    */
    int32_t io_dtype;        /* dm_dtype of u, delta, z, out                           */
    int32_t bc_dtype;        /* dm_dtype of B, C                                       */
    int32_t flags;
    int32_t ckpt_every;      /* steps between checkpoints (only read when ckpt != 0)   */
    int32_t ckpt_dtype;      /* DM_F32 or DM_BF16 (see above)                          */
    const void *u, *delta, *z;   /* z may be NULL */
    void *out;
    const void *B, *C;
    const float *A;          /* [dim][dstate] fp32, contiguous */
    const float *D;          /* [dim] fp32 or NULL             */
    const float *delta_bias; /* [dim] fp32 or NULL             */
    const int32_t *z_row_index;   /* [ndir][seqlen] or NULL */
    const int32_t *out_row_index; /* [ndir][seqlen] or NULL */
    void *ckpt;              /* or NULL */
    float *last_state;       /* or NULL */
    /* element strides; the channel stride of u/delta/z/out and the state stride of B/C must be 1 */
    int64_t u_ss, u_sl, u_sd;
    int64_t dt_ss, dt_sl, dt_sd;
    int64_t z_ss, z_sl, z_sd;
    int64_t o_ss, o_sl, o_sd;
    int64_t B_ss, B_sl, B_sg, B_sn;
    int64_t C_ss, C_sl, C_sg, C_sn;
} dm_scan_fwd_args;

int dm_selective_scan_fwd(const dm_scan_fwd_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Selective scan, backward.  Replaces selective_scan_cuda.bwd (autograd of the above; training path
 * train.py:259).  Consumes the forward's checkpoints, recomputes the states of one chunk at a time in
 * registers and runs the adjoint recurrence in reverse time (equations: SURVEY.md A.1-bwd).
 *
 *   dout is read at row out_row_index[dir][l] (the gather that is the adjoint of the forward's scatter),
 *   dz is written at row z_row_index[dir][l] of a per-direction buffer [s][row][d].
 *   dB/dC: partial sums over groups of GC = dm_scan_bwd_group_channels(dstate) channels, fp32, layout
 *          [s][l][ceil(dim/GC)][2*dstate] (B then C);
 *   dA: [s][dim][dstate], dD: [s][dim], ddelta_bias: [s][dim]  fp32 per-sequence partials.
 *   The caller reduces the partials (deterministic; no atomics).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nseq, dim, seqlen, dstate;
    int32_t ngroups;
    int32_t batch_per_dir;
    int32_t io_dtype;
    int32_t bc_dtype;
    int32_t flags;
    int32_t ckpt_every;
    int32_t ckpt_dtype;
    const void *u, *delta, *z, *dout;
    const void *B, *C;
    const float *A, *D, *delta_bias;
    const int32_t *z_row_index;
    const int32_t *out_row_index;
    const void *ckpt;        /* required */
    void *du, *ddelta, *dz;  /* dz NULL iff z NULL; same dtype as u */
    float *dBC_partial;      /* [nseq][seqlen][ceil(dim/GC)][2*dstate] */
    float *dA_partial;       /* [nseq][dim][dstate]              */
    float *dD_partial;       /* [nseq][dim] or NULL              */
    float *dbias_partial;    /* [nseq][dim] or NULL              */
    int64_t u_ss, u_sl, u_sd;
    int64_t dt_ss, dt_sl, dt_sd;
    int64_t z_ss, z_sl, z_sd;
    int64_t do_ss, do_sl, do_sd;
    int64_t B_ss, B_sl, B_sg, B_sn;
    int64_t C_ss, C_sl, C_sg, C_sn;
    int64_t du_ss, du_sl, du_sd;
    int64_t ddt_ss, ddt_sl, ddt_sd;
    int64_t dz_ss, dz_sl, dz_sd;
    int64_t part_ss;         /* row (per-sequence) stride in floats of dA_partial / dD_partial / dbias_partial; 0 = packed
                                (dim*dstate, dim, dim).  With one stride the three can be column blocks of ONE
                                [nseq][dim*(dstate+2)] buffer that a single dm_colsum_f32 reduces.                        */
} dm_scan_bwd_args;

int dm_selective_scan_bwd(const dm_scan_bwd_args *args, void *stream);
int dm_scan_bwd_group_channels(int dstate);   /* GC of the sequential kernel; <= 0 if dstate is not instantiated */
/* GC of THIS launch: small launches (graphed small-batch training) take a chunk-parallel kernel -- NW waves per (sequence,
 * 64 channels), each owning a run of time steps, two passes joined by the linearity of the adjoint carry -- which writes
 * one dB/dC partial row per 64 channels; size dBC_partial with this value. */
int dm_scan_bwd_launch_group_channels(int nseq, int dim, int seqlen, int dstate, int flags);

/* ------------------------------------------------------------------------------------------------
 * Token gather + causal depthwise conv1d (+bias, +SiLU), forward.  Replaces
 *   CrossScan / scan_permutation     block/mamba.py:26-45   (xs[:,k] = x[:, :, order_k])
 *   causal_conv1d_cuda.causal_conv1d_fwd  (causal_conv1d_fn, block/mamba.py:13; inside mamba_inner_fn)
 * in one pass:  out[dir][b][l][d] = act(bias[d] + sum_j w[d][j] * x[b][ idx[dir][l-(W-1)+j] ][d])
 * with terms at l-(W-1)+j < 0 dropped (left zero padding).  row_index == NULL means identity, ndir = 1.
 * x: [batch][seqlen][*] token-major view (x_sd must be 1); out: [ndir*batch][seqlen][dim] token-major.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, dim, seqlen, width, ndir;
    int32_t io_dtype, w_dtype;
    int32_t flags;
    const void *x;
    const void *weight;       /* [dim][width], w_dtype, contiguous */
    const void *bias;         /* [dim] or NULL, w_dtype            */
    const int32_t *row_index; /* [ndir][seqlen] or NULL            */
    void *out;
    int64_t x_sb, x_sl, x_sd;
    int64_t o_ss, o_sl, o_sd;
} dm_conv_fwd_args;

int dm_gather_conv1d_fwd(const dm_conv_fwd_args *args, void *stream);

/* Backward of the above (CrossScan.backward block/mamba.py:47-57 + causal_conv1d_bwd).
 *   dx_dir[dir][b][ idx[dir][m] ][d] = sum_j w[d][j] * g[dir][b][m+(W-1)-j][d],   g = dout * act'(pre)
 * i.e. the gradient of every direction is written back in TOKEN order (the scatter through idx is the
 * adjoint of the gather); the caller sums the ndir slabs with dm_token_merge.
 *   dw_partial: [ndir*batch][nchunk][dim][width] fp32, db_partial: [ndir*batch][nchunk][dim] fp32: one partial row
 *   per workgroup (a run of consecutive time chunks of one sequence); nchunk = dm_conv_nchunk(seqlen).
 */
typedef struct {
    int32_t batch, dim, seqlen, width, ndir;
    int32_t io_dtype, w_dtype;
    int32_t flags;
    int32_t nchunk;           /* = dm_conv_nchunk(seqlen) */
    const void *x, *weight, *bias, *dout;
    const int32_t *row_index;
    void *dx;                 /* [ndir*batch][seqlen][dim], io_dtype, token order */
    float *dw_partial, *db_partial;
    int64_t x_sb, x_sl, x_sd;
    int64_t do_ss, do_sl, do_sd;
    int64_t dx_ss, dx_sl, dx_sd;
    int64_t part_ss;          /* ABI 29: element stride between partial rows of dw_partial AND db_partial; 0 = dense (dim * width / dim).
                                 With part_ss = dim * (width + 1) and db_partial = dw_partial + dim * width the two live in ONE
                                 [rows][dim * (width + 1)] buffer that one dm_colsum_f32 reduces. */
} dm_conv_bwd_args;

int dm_gather_conv1d_bwd(const dm_conv_bwd_args *args, void *stream);
int dm_conv_nchunk(int seqlen);

/* ------------------------------------------------------------------------------------------------
 * The same forward FUSED with x_proj (BASELINE north star: "fused causal depthwise conv1d + SiLU + x-proj kernel"):
 *   out  = as dm_gather_conv1d_fwd                                  [ndir*batch][seqlen][dim]
 *   xdbl[(s*seqlen + l)][c] = sum_d out[s][l][d] * wx[c][d]         c < nproj   (x_dbl = x~ @ x_proj.weight^T, block/mamba.py:346,
 *                                                                    SURVEY.md A.1 step 3; fp32 accumulation on the matrix pipe)
 * wx: [nproj][dim] in the I/O dtype, contiguous (x_proj.weight as it is), 16-byte aligned.  16-bit I/O only, dim in
 * {128, 256, 512, 1024}, nproj <= 64: dm_gather_conv1d_xproj_supported() tells; other shapes take the unfused pair.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, dim, seqlen, width, ndir;
    int32_t io_dtype, w_dtype;
    int32_t flags;
    int32_t nproj;            /* rows of wx = columns of xdbl (dt_rank + 2*d_state) */
    int32_t _pad;
    const void *x;
    const void *weight;       /* [dim][width], w_dtype, contiguous */
    const void *bias;         /* [dim] or NULL, w_dtype            */
    const int32_t *row_index; /* [ndir][seqlen] or NULL            */
    const void *wx;           /* [nproj][dim], io_dtype            */
    void *out;
    void *xdbl;               /* [ndir*batch*seqlen][*], io_dtype, row stride xd_sr */
    int64_t x_sb, x_sl, x_sd;
    int64_t o_ss, o_sl, o_sd;
    int64_t xd_sr;
} dm_conv_xproj_fwd_args;

int dm_gather_conv1d_xproj_fwd(const dm_conv_xproj_fwd_args *args, void *stream);
int dm_gather_conv1d_xproj_supported(int dim, int nproj, int io_dtype);
int dm_gather_conv1d_xproj_width_supported(int width);   /* conv widths the fused forward AND backward are instantiated for */

/* Backward of the conv FUSED with the x_proj input gradient: the gradient entering the conv is
 *   dxc[s][l][:] = du[s][l][:] + dxdbl[s*seqlen + l][:] @ wx            (d x~ = dL/du + d x_dbl . x_proj.weight)
 * and is never materialised (the unfused path runs an in-place addmm over [ndir*batch*seqlen][dim] first).  Otherwise as
 * dm_gather_conv1d_bwd: dx in token order per direction; dw_partial [ndir*batch][dim][width], db_partial [ndir*batch][dim]
 * fp32 -- ONE partial row per sequence.  wxt: x_proj.weight TRANSPOSED, [dim][nproj] contiguous, io dtype; nproj = 64
 * (dt_rank 32 + 2 * d_state 16, every DiffMa-* model; other widths take the unfused pair).
 */
typedef struct {
    int32_t batch, dim, seqlen, width, ndir;
    int32_t io_dtype, w_dtype;
    int32_t flags;
    int32_t nproj;
    int32_t _pad;
    const void *x, *weight, *bias;
    const int32_t *row_index;
    const void *du;           /* [ndir*batch][seqlen][dim] io dtype */
    const void *dxdbl;        /* [ndir*batch*seqlen][*] io dtype, row stride xd_sr */
    const void *wxt;          /* [dim][nproj] io dtype */
    void *dx;                 /* [ndir*batch][seqlen][dim] io dtype, token order */
    float *dw_partial, *db_partial;
    int64_t x_sb, x_sl, x_sd;
    int64_t du_ss, du_sl, du_sd;
    int64_t dx_ss, dx_sl, dx_sd;
    int64_t xd_sr;
    int64_t part_ss;          /* row stride in floats of dw_partial / db_partial; 0 = packed (dim*width, dim) */
} dm_conv_xproj_bwd_args;

int dm_gather_conv1d_xproj_bwd(const dm_conv_xproj_bwd_args *args, void *stream);
int dm_gather_conv1d_xproj_bwd_supported(int dim, int nproj, int io_dtype);
/* ABI 27.  DM_FLAG_DX_MERGED launches come in two forms: the whole-sample form (a workgroup walks the directions of a sample; the
 * running sum of dx is read back and rewritten in place per direction: 2 * (ndir-1) extra passes over [batch][seqlen][dim]) and,
 * for seqlen <= 256 with 16-byte aligned dx rows, the SLAB form (a persistent workgroup per CU owns 128 channels, the running sum
 * lives in LDS, dx is written once).  Returns 0 when `args` would take the whole-sample form, else R = the number of dw / db
 * partial rows that carry sums in the slab form (rows R .. batch-1 are zero-filled, or left alone under DM_FLAG_PARTIAL_COMPACT). */
int dm_gather_conv1d_xproj_bwd_slab(const dm_conv_xproj_bwd_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Token merge: out[b][t][c] = sum_k in[k][b][ idx[k][t] ][c]     (idx NULL = identity).
 * Replaces CrossMerge / merge_permutation  block/mamba.py:29-30,59-69 applied BEFORE out_proj
 * (out_proj is linear and bias-free, block/mamba.py:240,315, so sum-then-project == project-then-sum
 * up to rounding) and serves as the 3-slab gradient sum of the backward pass.
 * ---------------------------------------------------------------------------------------------- */
/* Gated form (gate != NULL):  pre[b][t][c] = sum_k in[k][b][idx[k][t]][c] ;  out = pre * silu(gate[b][t][c]).
 * CrossScan permutes x and z together and CrossMerge applies the inverse permutation (block/mamba.py:41-45,66-68), so the
 * SiLU(z) gate of the three per-direction operators (mamba_inner_fn, block/mamba.py:346-348) factors out of the 3-way sum:
 * it is applied ONCE per token here instead of three times inside the scans.  `pre` (optional, io_dtype) keeps the
 * ungated sum for the backward (dm_gate_bwd). */
typedef struct {
    int32_t nin, batch, seqlen, dim;
    int32_t io_dtype, out_dtype;
    const void *in;            /* [nin][batch][seqlen][dim] via strides */
    const int32_t *row_index;  /* [nin][seqlen] or NULL */
    void *out;
    int64_t in_sk, in_sb, in_sl;   /* channel stride 1 */
    int64_t o_sb, o_sl;
    const void *gate;          /* [batch][seqlen][dim] via strides (io_dtype), or NULL */
    void *pre;                 /* [batch][seqlen][dim] via strides (io_dtype), or NULL; only with gate */
    int64_t g_sb, g_sl;
    int64_t p_sb, p_sl;
} dm_merge_args;

int dm_token_merge(const dm_merge_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The operator boundary's layout change (ABI 29).  The reference calls its operators with CHANNEL-MAJOR tensors: xz is
 * (B, 2 Din, L) with L contiguous -- block/mamba.py:333-337 builds it as a permuted view of the (2 Din, B L) in_proj product,
 * CrossScan keeps the layout (block/mamba.py:31-45) and mamba_inner_fn / selective_scan_fn / causal_conv1d_fn take and return
 * (B, D, L) (block/mamba.py:346-348).  The kernels above are token-major; dm_repack is the bridge, one HBM-bound pass
 * (one read + one write of the tensor) in either direction:
 *      to_token_major = 1:   tm[b][l][d] = cm[b][d][l]        src = cm, dst = tm
 *      to_token_major = 0:   cm[b][d][l] = tm[b][l][d]        src = tm, dst = cm   (gradients back in the reference's layout)
 * cm: element strides (cm_sb, cm_sd, 1); tm: (tm_sb, tm_sl, 1).  Any strides / alignment are accepted (16-byte, 8-byte or
 * element accesses are chosen per side); elements are moved as words, so io_dtype only selects the width (DM_F32: 4 bytes,
 * DM_BF16 / DM_F16: 2).  src and dst must not overlap.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, dim, seqlen;
    int32_t io_dtype;
    int32_t to_token_major;
    const void *src;
    void *dst;
    int64_t cm_sb, cm_sd;
    int64_t tm_sb, tm_sl;
} dm_repack_args;

int dm_repack(const dm_repack_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the hoisted gate  y = pre * silu(z):   g = dy * silu(z)   (what the per-direction scan backwards read as dout)
 *                                                    dz = dy * pre * silu'(z)      -- once per token, one pass.
 * All tensors [batch][seqlen][dim] via (batch, row) strides, channel stride 1, one dtype.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, seqlen, dim;
    int32_t io_dtype;
    const void *dy, *z, *pre;
    void *g, *dz;
    int64_t dy_sb, dy_sl;
    int64_t z_sb, z_sl;
    int64_t p_sb, p_sl;
    int64_t g_sb, g_sl;
    int64_t dz_sb, dz_sl;
} dm_gate_bwd_args;

int dm_gate_bwd(const dm_gate_bwd_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * dt_proj with the softplus folded into its epilogue (the dt_proj + softplus stage of mamba_inner_fn, SURVEY.md A.1 step 3-4;
 * dt_proj.weight [dim][rank], dt_proj.bias [dim], block/mamba.py:262-287):
 *     delta[m][d] = softplus( sum_r xdbl[m][r] * w[d][r] + bias[d] )          m < rows, r < rank
 * One v_mfma_f32_16x16x32 per 16 x 16 output tile (rank <= 32; two for rank <= 64); the kernel is bound by the write of
 * delta.  The scans then run with DM_FLAG_DELTA_ACTIVATED.  16-bit I/O only (xdbl, w, delta share io_dtype; bias fp32).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t rows, dim, rank;
    int32_t io_dtype;
    const void *xdbl;          /* [rows][>= rank] , row stride xd_sr elements (a multiple of 8), 16-byte aligned */
    const void *w;             /* [dim][rank] contiguous */
    const float *bias;         /* [dim] fp32 or NULL */
    void *delta;               /* [rows][dim] contiguous */
    int64_t xd_sr;
} dm_dtproj_args;

int dm_dtproj_softplus_fwd(const dm_dtproj_args *args, void *stream);
int dm_dtproj_softplus_supported(int dim, int rank, int io_dtype);

/* Backward of the product above (the softplus' derivative is applied by dm_selective_scan_bwd, DM_FLAG_DELTA_ACTIVATED):
 *     dxdbl[m][r]  = sum_d ddelta[m][d] * w[d][r]      r < rank   (written into the first `rank` columns of the d x_dbl rows)
 *     part[blk]    = this workgroup's share of  dW[d][r] = sum_m ddelta[m][d] * xdbl[m][r]   ([nblk][dim][rank] fp32; dm_colsum_f32)
 * in ONE read of ddelta (block/mamba.py:346-348 differentiates this product inside mamba_inner_fn's backward).  Any row count,
 * nblk in 1 .. ceil(rows / 32) (the grid), dims as dm_dtproj_bwd_supported says (16-bit I/O, dim in {512, 768, 1024}, rank in {16, 32}). */
typedef struct {
    int32_t rows, dim, rank;
    int32_t io_dtype;
    int32_t nblk, _pad;
    const void *ddelta;        /* [rows][dim] contiguous, 16-byte aligned                                           */
    const void *xdbl;          /* [rows][>= rank], row stride xd_sr elements (a multiple of 8), 16-byte aligned     */
    const void *w;             /* [dim][rank] contiguous                                                            */
    void *dxdbl;               /* [rows][>= rank], row stride dxd_sr elements (a multiple of 4), 8-byte aligned     */
    float *part;               /* [nblk][dim][rank]                                                                 */
    int64_t xd_sr, dxd_sr;
} dm_dtproj_bwd_args;

int dm_dtproj_bwd(const dm_dtproj_bwd_args *args, void *stream);
int dm_dtproj_bwd_supported(int dim, int rank, int io_dtype);

/* ------------------------------------------------------------------------------------------------
 * Block elementwise ops of Spiral_MambaBlock.forward (reference block/mamba_block.py:100-115), each one
 * HBM pass instead of a chain of ATen kernels.
 *
 * dm_ln_mod_fwd :  r = [x | x2] (x2 optional: the torch.cat of :111),  n = LayerNorm(r) * gamma + beta  (:103, :90),
 *                  m = n * (1 + scale[b]) + shift[b]  (modulate, :8-9,104; skipped when scale == NULL),
 *                  y1 = m,  y2 = m * mask[row]  (soft mask, :105; skipped when mask == NULL).
 *                  stats[row] = {mean, rstd} (fp32) for the backward.  rows = batch * rows_per_batch.
 * dm_ln_mod_bwd :  given dy1 (, dy2): dx (, dx2), and fp32 partial sums over groups of rows_per_block rows (a multiple of 4, one workgroup
 *                  each; 0 = DM_LN_ROWS_PER_BLOCK.  Few samples per launch want small groups: at one sample 28-row groups are 7 workgroups
 *                  whose waves walk 7 rows one after the other, 4-row groups are 49 workgroups of one row per wave)
 *                  part[blk][4][C] = {dshift, dscale, dgamma, dbeta}  (blk = b*blocks_per_batch + i);
 *                  accumulate != 0 adds into dx/dx2 instead of overwriting (the cat branch adds to the blend's
 *                  gradients); dx_add != NULL: dx = (computed) + dx_add[row] read-only (the residual branch's gradient,
 *                  which may be shared with other autograd nodes and must not be modified), x dtype, row stride dxa_sr.
 * dm_blend_fwd  :  out = x + gate[b] * (a[row]*xs + (1-a[row])*ws)      (:113-114)
 * dm_blend_bwd  :  dxs = g*gate*a, dws = g*gate*(1-a), da[row] = sum_c g*gate*(xs-ws),
 *                  dgate_part[blk][C] = sum_rows g*(a*xs+(1-a)*ws)        (dx = g is the caller's)
 * ---------------------------------------------------------------------------------------------- */
#define DM_LN_ROWS_PER_BLOCK 28
typedef struct {
    int32_t batch, rows_per_batch, C1, C2;      /* C2 = 0 without x2; C = C1 + C2                          */
    int32_t x_dtype, y_dtype, mod_dtype;        /* x/x2/dx/dx2 ; y1/y2/dy1/dy2 ; shift/scale/mask           */
    int32_t accumulate;                         /* bwd only                                                  */
    float eps;
    int32_t rows_per_block;                     /* bwd only: rows per partial-sum group; 0 = DM_LN_ROWS_PER_BLOCK (ABI 29; this slot was padding) */
    const void *x, *x2;                         /* [rows][C1], [rows][C2], row strides below                 */
    const float *gamma, *beta;                  /* [C] or NULL (no affine)                                   */
    const void *shift, *scale;                  /* [batch][C] rows of stride mod_sb, or NULL                 */
    const void *mask;                           /* [rows] or NULL                                            */
    void *y1, *y2;                              /* fwd outputs [rows][C] (y2 NULL iff mask NULL)             */
    float *stats;                               /* [rows][2]                                                 */
    const void *dy1, *dy2;                      /* bwd inputs                                                */
    void *dx, *dx2;                             /* bwd outputs                                               */
    float *part;                                /* bwd: [batch*blocks_per_batch][4][C]                       */
    int64_t x_sr, x2_sr, y_sr, mod_sb, dx_sr, dx2_sr;
    const void *dx_add;                         /* bwd only, optional (C2 must be 0)                         */
    int64_t dxa_sr;
} dm_ln_mod_args;

int dm_ln_mod_fwd(const dm_ln_mod_args *args, void *stream);
int dm_ln_mod_bwd(const dm_ln_mod_args *args, void *stream);

typedef struct {
    int32_t batch, rows_per_batch, C;
    int32_t x_dtype, s_dtype, g_dtype;          /* x/out/g ; xs/ws/dxs/dws/a/da ; gate                       */
    int32_t rows_per_block, _pad;               /* bwd only: rows per partial-sum group; 0 = DM_LN_ROWS_PER_BLOCK (ABI 29)          */
    const void *x, *xs, *ws, *a, *gate;         /* gate [batch][C] rows of stride gate_sb; a [rows]          */
    void *out;                                  /* fwd                                                       */
    const void *g;                              /* bwd: dL/dout [rows][C]                                    */
    void *dxs, *dws, *da;                       /* bwd                                                       */
    float *dgate_part;                          /* bwd: [batch*blocks_per_batch][C]                          */
    int64_t gate_sb;
} dm_blend_args;

int dm_blend_fwd(const dm_blend_args *args, void *stream);
int dm_blend_bwd(const dm_blend_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Tail of the block's fusion MLP (block/mamba_block.py:90-91,111-112: attention_network = LayerNorm -> Linear(2C, C) -> SiLU ->
 * Linear(C, 1) -> Sigmoid).  h = the first Linear's GEMM output WITHOUT its bias, [rows][C] (row stride h_sr):
 *     dm_gate_head_fwd :  a[r] = sigmoid( sum_c silu(h[r][c] + b1[c]) * w2[c] + b2[0] )
 *     dm_gate_head_bwd :  dpre = da * a * (1 - a);  dh[r][c] = dpre * w2[c] * silu'(h + b1);  part[blk] = this workgroup's
 *                         [ sum_r dh (= d b1) | sum_r dpre * silu(h + b1) (= d w2) | sum_r dpre (= d b2), 0, 0, 0 ]
 * part: [nblk][2*C + 4] fp32, nblk chosen by the caller (it is the grid; rows are dealt round-robin to 4 * nblk waves);
 * dm_colsum_f32 reduces it.  b1 may be NULL.  C a multiple of 4 (fp32) / 8 (16-bit), C <= 1024 (fp32) / 2048; 16-byte aligned rows.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t rows;
    int32_t C, io_dtype;                        /* h, a, da, dh                                              */
    int32_t nblk, _pad;                         /* bwd only                                                  */
    const void *h;                              /* [rows][C]                                                 */
    const float *b1, *w2, *b2;                  /* [C] or NULL, [C], [1] or NULL                             */
    void *a;                                    /* fwd: output [rows]; bwd: input                            */
    const void *da;                             /* bwd: [rows]                                               */
    void *dh;                                   /* bwd: [rows][C]                                            */
    float *part;                                /* bwd: [nblk][2*C + 4]                                      */
    int64_t h_sr, dh_sr;
} dm_gate_head_args;

int dm_gate_head_fwd(const dm_gate_head_args *args, void *stream);
int dm_gate_head_bwd(const dm_gate_head_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Gated-RMSNorm epilogue of the Mamba-2 mixer fused with the 3-way CrossMerge (block/mamba2.py:349,402-403,
 * norm_before_gate = False; the gate is applied by the scan):
 *     out[r][:] = weight[:] * sum_k  y_k[r][:] * rsqrt(mean(y_k[r][:]^2) + eps)
 * y: [nslab][rows][C] slabs already in token order; rstd: [nslab][rows] fp32 (written by fwd, read by bwd).
 * bwd: dy [nslab][rows][C]; dw_part: [ceil(rows / dm_rmsnorm_merge_rows_per_block())][C] fp32 partial rows.
 * C % 4 == 0, C <= 4096; row/slab strides in elements (multiples of 4); 16-byte aligned tensors.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nslab, C;
    int64_t rows;
    int32_t io_dtype;                           /* y, out, dout, dy                                          */
    float eps;
    const void *y;
    const float *weight;                        /* [C] fp32                                                  */
    void *out;                                  /* fwd: [rows][C]                                            */
    float *rstd;
    const void *dout;                           /* bwd: [rows][C]                                            */
    void *dy;                                   /* bwd                                                       */
    float *dw_part;                             /* bwd                                                       */
    int64_t y_ss, y_sr, out_sr, dout_sr, dy_ss, dy_sr;
} dm_rmsnorm_merge_args;

int dm_rmsnorm_merge_fwd(const dm_rmsnorm_merge_args *args, void *stream);
int dm_rmsnorm_merge_bwd(const dm_rmsnorm_merge_args *args, void *stream);
int dm_rmsnorm_merge_rows_per_block(void);

/* ------------------------------------------------------------------------------------------------
 * Column sums of a tall contiguous fp32 matrix: out[c] = sum_r in[r][c].  Reduces the per-sequence partial rows the
 * scan backward writes (dA [nseq][dim*dstate], dD / ddelta_bias [nseq][dim]); deterministic.  cols % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *in;
    float *out;
    int64_t rows, cols;
} dm_colsum_args;

int dm_colsum_f32(const dm_colsum_args *args, void *stream);

/* out[m][c] = sum_w in[m][w][c] : the per-workgroup dB/dC partial rows of dm_selective_scan_bwd ([rows][nw][cols] fp32,
 * contiguous) summed and placed -- converted to out_dtype -- into their columns of the d x_dbl buffer (row stride out_sr
 * elements) in one pass (ATen: a reduction into a temporary + a converting strided copy). */
typedef struct {
    int64_t rows;
    int32_t nw, cols;
    int32_t out_dtype, _pad;
    const float *in;
    void *out;
    int64_t out_sr;
} dm_sum_partials_args;

int dm_sum_partials(const dm_sum_partials_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Mamba-2 SSD core, single chunk, on the matrix pipe (forward of --use-mamba2 with 16-bit activations).  Replaces the scan stage of
 * mamba_split_conv1d_scan_combined (block/mamba2.py:392-410) for chunk_size >= seqlen:
 *     dt = softplus(dt_raw + dt_bias[h]);  s = A[h] * cumsum(dt);  G = (C B^T) .* exp(s_l - s_i) [i <= l];
 *     y = (G diag(dt)) x + D[h] x;   out = y * silu(z)
 * x: [nseq][L][nheads*64] view (the conv output's x columns), B, C: [nseq][L][16] views (16-byte aligned rows), all after the
 * conv, per gathered sequence; dt_raw: [batch][L][nheads] and z: [batch][L][nheads*64] in TOKEN order, read through z_row_index;
 * out: [nseq][rows][nheads*64], step l stored at row out_row_index[dir][l].  16-bit I/O, headdim 64, d_state 16,
 * seqlen <= 224: dm_ssd_fwd_supported() tells; everything else takes the A-shared scan.  dm_ssd_bwd below is its backward.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nseq, batch_per_dir, seqlen, nheads, headdim, dstate;
    int32_t io_dtype, flags;
    const void *x, *B, *C, *dt, *z;
    const float *A, *D, *dt_bias;       /* [nheads] fp32 (D, dt_bias may be NULL) */
    const int32_t *z_row_index, *out_row_index;
    void *out;
    int64_t x_ss, x_sl;
    int64_t B_ss, B_sl;
    int64_t C_ss, C_sl;
    int64_t dt_sb, dt_sl;
    int64_t z_ss, z_sl;
    int64_t o_ss, o_sl;
} dm_ssd_fwd_args;

int dm_ssd_fwd(const dm_ssd_fwd_args *args, void *stream);
int dm_ssd_fwd_supported(int seqlen, int headdim, int dstate, int io_dtype);

/* ------------------------------------------------------------------------------------------------
 * Backward twin of dm_ssd_fwd (same operator, same operands; SURVEY.md A.2, reference call block/mamba2.py:392-410 under
 * autograd).  Nothing is saved by the forward: one workgroup per (sequence, head) recomputes the score tiles.
 *   dout: [nseq][rows][nheads*64] gradient of the GATED output, step l read at row out_row_index[dir][l] (token order per
 *         sequence, as dm_rmsnorm_merge_bwd leaves it);
 *   dx:   [nseq][L][..] view (scan order; the x columns of the conv output's gradient);
 *   dz:   [nseq][rows][nheads*64], step l written at row z_row_index[dir][l] (token order per direction: dm_token_merge sums them);
 *   dBC_part: fp32 [nheads][nseq][L][32]  per-head partial rows  dB (0..15) | dC (16..31) -- the caller sums over heads
 *         (head-major: one dm_colsum_f32 over nheads rows);
 *   ddt:  fp32 [nseq][rows][nheads], gradient of the RAW per-head dt (through softplus), step l at row z_row_index[dir][l];
 *   dAD_part: fp32 [nseq][3][nheads] partial sums of dA | dD | d dt_bias per (sequence, head) -- the caller sums over sequences
 *         (one dm_colsum_f32 over nseq rows when 3 * nheads is a multiple of 4).
 * 16-bit I/O, headdim 64, d_state 16, seqlen <= 196, 16-byte aligned rows: dm_ssd_bwd_supported() tells.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nseq, batch_per_dir, seqlen, nheads, headdim, dstate;
    int32_t io_dtype, flags;
    const void *x, *B, *C, *dt, *z, *dout;
    const float *A, *D, *dt_bias;       /* [nheads] fp32 (D, dt_bias may be NULL) */
    const int32_t *z_row_index, *out_row_index;
    void *dx, *dz;
    float *dBC_part, *ddt, *dAD_part;
    int64_t x_ss, x_sl;
    int64_t B_ss, B_sl;
    int64_t C_ss, C_sl;
    int64_t dt_sb, dt_sl;
    int64_t z_ss, z_sl;
    int64_t do_ss, do_sl;
    int64_t dx_ss, dx_sl;
    int64_t dz_ss, dz_sl;
} dm_ssd_bwd_args;

int dm_ssd_bwd(const dm_ssd_bwd_args *args, void *stream);
int dm_ssd_bwd_supported(int seqlen, int headdim, int dstate, int io_dtype);

/* ------------------------------------------------------------------------------------------------
 * One reverse-diffusion step after the denoiser call, fused (reference diffusion/gaussian_diffusion.py:285-323
 * p_mean_variance with learned-range variance and epsilon prediction, :232-252 q_posterior_mean_variance, :410-416 p_sample,
 * :548-598 ddim_sample).  model_out: [batch][2*channels][hw] (eps | variance logits in [-1, 1]), dtype out_dtype; x, noise,
 * sample, pred_xstart: fp32 [batch][channels][hw] contiguous (noise may be NULL: no noise term; pred_xstart may be NULL).
 * t: int64 [batch].  tables: fp32 [nrows][T], the caller's per-timestep coefficient rows; the row_* fields name the rows of
 * log(posterior variance, clipped), log(beta), sqrt(1/abar), sqrt(1/abar - 1), posterior mean coefficients 1 and 2, abar and
 * abar_prev.  mode 0 = ancestral (DDPM) step, 1 = DDIM step with `eta`.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, channels, hw, T;
    int32_t mode, clip, out_dtype;
    float eta;
    int32_t row_post_logvar, row_log_betas, row_sqrt_recip_ac, row_sqrt_recipm1_ac;
    int32_t row_coef1, row_coef2, row_ac, row_ac_prev;
    const void *model_out;
    const float *x, *noise;
    const int64_t *t;
    const float *tables;
    float *sample, *pred_xstart;
} dm_diffusion_step_args;

int dm_diffusion_step(const dm_diffusion_step_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Dense product for the mixer's projections in the small-launch regime (ABI 26):  C[P][Q] = opA(A)[P][Kc] * opB(B)[Kc][Q],
 * 16-bit operands (bf16 / fp16), fp32 accumulation on the matrix pipe, C in fp32 or in the operand dtype.
 * Replaces the library GEMMs of  in_proj / out_proj  (reference block/mamba.py:261,315; called at :333-337 and inside
 * mamba_inner_fn, :346) and of their two gradients where a training step is bound by the number of launches (one sample per GPU,
 * config/brain.yaml:11) -- dm_gemm_n below multiplies for both mixers of a block in ONE launch.
 *   a_kmajor = 1: A is stored [P][Kc] (row stride lda);  0: stored [Kc][P]
 *   b_kmajor = 1: B is stored [Q][Kc] (row stride ldb);  0: stored [Kc][Q]
 *   forward   y  = x W^T  : A = x  [M][K] (1), B = W [N][K] (1)            -> [M][N]
 *   dgrad     dx = dy W   : A = dy [M][N] (1), B = W [N][K] (0)            -> [M][K]
 *   wgrad     dW = dy^T x : A = dy [M][N] (0), B = x [M][K] (0), C fp32    -> [N][K]
 * Q % 8 == 0; the contiguous index of every operand a multiple of 8; 16-byte aligned tensors; row strides % 8 (C: % 4).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t P, Q, Kc;
    int32_t ab_dtype;        /* DM_BF16 or DM_F16 */
    int32_t c_dtype;         /* DM_F32 or ab_dtype */
    int32_t a_kmajor, b_kmajor;
    int32_t accumulate;      /* 1: C += product (C is read in c_dtype), 0: C = product */
    const void *a, *b;
    void *c;
    int64_t lda, ldb, ldc;   /* row strides of the STORED matrices, elements */
} dm_gemm_args;

int dm_gemm(const dm_gemm_args *args, void *stream);
int dm_gemm_supported(int P, int Q, int Kc, int a_kmajor, int b_kmajor, int ab_dtype, int c_dtype);

/* ------------------------------------------------------------------------------------------------
 * K12 (ABI 28)  dm_gemm_large -- the same product for the LARGE-BATCH projections (csrc/gemm_large.hip): both operands k-major,
 * 16-bit C of the operand dtype, no accumulation.  Replaces the library (hipBLASLt through F.linear / torch.mm) GEMMs of
 * in_proj / out_proj (reference block/mamba.py:261,315,333-337) and of their input gradients (through a transposed 16-bit copy
 * of the weight) at M = B L >= 2048 rows: a persistent 256 x 256 tile kernel whose operand stream and C-tile stores run across
 * tile boundaries (the contraction is only 512 .. 2048 long).  P >= 2048, Q % 256 == 0, Kc >= 512 and Kc % 512 == 0 (the unrolled body is 16 K-steps of 32); strides as dm_gemm,
 * ldc % 8 == 0; every tensor below 2 GB.  dm_gemm_large_supported answers for a shape without launching.
 * ---------------------------------------------------------------------------------------------- */
int dm_gemm_large(const dm_gemm_args *args, void *stream);
int dm_gemm_large_supported(int P, int Q, int Kc, int a_kmajor, int b_kmajor, int ab_dtype, int c_dtype);

/* ------------------------------------------------------------------------------------------------
 * Several congruent launches in one (ABI 25).
 *
 * A DiffMa block runs TWO mixers on tensors of the same shape with different weights (reference block/mamba_block.py:107-108),
 * and the reference's own configuration trains at one sample per GPU (config/brain.yaml:11), where a step is bound by the
 * NUMBER of kernel launches.  Each function below takes an ARRAY of `n` argument structs and is equivalent to calling its
 * single-launch namesake on args[0], args[1], ... in order.  Neighbouring structs that are congruent -- every size, stride,
 * dtype and flag equal, the same pointers NULL -- share ONE launch when the kernel their shape selects is built for it (all of
 * them except the large-launch sequential scans; the kernel picks its struct by blockIdx.z); anything else is launched one
 * after the other.  The results are bit-identical to the separate calls.  The launches must be independent of each other.
 * ---------------------------------------------------------------------------------------------- */
int dm_selective_scan_fwd_n(const dm_scan_fwd_args *args, int n, void *stream);
int dm_selective_scan_bwd_n(const dm_scan_bwd_args *args, int n, void *stream);
int dm_gather_conv1d_fwd_n(const dm_conv_fwd_args *args, int n, void *stream);
int dm_gather_conv1d_bwd_n(const dm_conv_bwd_args *args, int n, void *stream);
int dm_token_merge_n(const dm_merge_args *args, int n, void *stream);
int dm_gate_bwd_n(const dm_gate_bwd_args *args, int n, void *stream);
int dm_dtproj_softplus_fwd_n(const dm_dtproj_args *args, int n, void *stream);
int dm_dtproj_bwd_n(const dm_dtproj_bwd_args *args, int n, void *stream);
int dm_colsum_f32_n(const dm_colsum_args *args, int n, void *stream);
int dm_sum_partials_n(const dm_sum_partials_args *args, int n, void *stream);
int dm_gemm_n(const dm_gemm_args *args, int n, void *stream);

/* Library introspection. */
int dm_abi_version(void);
const char *dm_last_error(void);
const char *dm_build_info(void);   /* "gfx950 hipcc <ver> <date>" */

#ifdef __cplusplus
}
#endif
#endif /* DIFFMA_HIP_H */
