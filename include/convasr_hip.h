/* convasr_hip.h -- C ABI of libconvasr_hip.so: the MI355X (gfx950) implementation of convasr's one hot path.
 *
 * The reference (vadimkantorov/convasr) has no FFI on this path: its boundary is the Python nn.Module surface of
 * models.py, whose arithmetic is delegated to torch.nn.functional (ATen/cuDNN/cuFFT).  Each entry point below names
 * the reference call site(s) (file:line in /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory; nothing here allocates, frees or synchronises;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns 0, or a negative
 *     CONVASR_E* code with a message retrievable through convasr_last_error() (thread-local);
 *   - activations are "channels-last": element (b, c, t) of a logical (B, C, T) tensor lives at
 *     ((b * T + t) * C + c); dtype is CONVASR_F32, CONVASR_BF16 or CONVASR_F16 (the two 16-bit storage types run the same
 *     kernels, instantiated on v_mfma_*_bf16 / v_mfma_*_f16; every sum is accumulated in fp32).  convasr_convert_layout() moves data between
 *     this layout and arbitrary (B, C, T) strides (e.g. torch-contiguous NCW);
 *   - conv weights are consumed in a packed layout [tap][cout_pad][cin] produced by convasr_pack_conv_weight()
 *     from the reference's (Cout, Cin, K) fp32 parameters;
 *   - lengths are passed the way the reference passes them: `xlen` = fraction of the time axis per utterance
 *     (float, models.py:611-614); valid frames of a T-long axis are t < ceil(xlen[b] * T).
 */
#ifndef CONVASR_HIP_H
#define CONVASR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONVASR_ABI_VERSION 10

enum { CONVASR_F32 = 0, CONVASR_BF16 = 1, CONVASR_I16 = 2, CONVASR_F16 = 3 };
enum { CONVASR_ACT_NONE = 0, CONVASR_ACT_RELU = 1, CONVASR_ACT_HARDTANH = 2, CONVASR_ACT_LEAKY_RELU = 3 };
enum { CONVASR_OK = 0, CONVASR_EINVAL = -1, CONVASR_ELAUNCH = -2, CONVASR_EUNSUPPORTED = -3 };

int convasr_abi_version(void);
const char* convasr_last_error(void);

/* ---- layout / dtype plumbing -------------------------------------------------------------------------------- */

/* dst[b,c,t] = (dst_dtype) src[b,c,t] for arbitrary element strides on both sides. */
int convasr_convert_layout(const void* src, int src_dtype, int64_t src_sb, int64_t src_sc, int64_t src_st,
                           void* dst, int dst_dtype, int64_t dst_sb, int64_t dst_sc, int64_t dst_st,
                           int B, int C, int T, void* stream);

/* ---- frontend: models.py:565-597 (LogFilterBankFrontend.forward), models.py:684-686 (normalize_signal) -------- */

/* absmax[b] = max_t |signal[b,t]|  (models.py:685).  signal dtype F32 or I16. */
int convasr_signal_absmax(const void* signal, int signal_dtype, int B, int T, float* absmax, void* stream);

/* Fused normalise -> pre-emphasis -> mask -> reflect/zero pad -> STFT(nfft, hop, window centred in nfft) -> power ->
 * mel (nmel x (nfft/2+1), + bias) -> log.  out is channels-last (B, F, nmel) fp32 with F = 1 + T / hop.
 * absmax may be NULL (normalize_signal=False); xlen may be NULL (no mask).  nfft = 128, 256, 512 or 1024 (models.py:516: the power
 * of two above the window -- 512 at 16 kHz x 0.02 s, 256 at train.py's default 8 kHz x 0.02 s), win_length <= nfft, nmel <= 128 (one or two channels per lane);
 * anything else returns CONVASR_EUNSUPPORTED. */
int convasr_logmel_fwd(const void* signal, int signal_dtype, const float* absmax, const float* xlen,
                       const float* window, int win_length, const float* mel_weight, const float* mel_bias,
                       float* out, int B, int T, int nfft, int hop, int nmel, float preemphasis, void* stream);

/* ---- MaskedInstanceNorm1d.forward: models.py:694-719 ---------------------------------------------------------- */

/* x: (B, C, T), y: (B, C, T_out >= T) with the given element strides; frames T .. T_out - 1 of y are written as zeros (the even-length
 * input the stride-2 fold below wants).  xlen NULL -> legacy un-masked branch (models.py:704-710). */
int convasr_instnorm_fwd(const void* x, int x_dtype, int64_t x_sb, int64_t x_sc, int64_t x_st,
                         void* y, int y_dtype, int64_t y_sb, int64_t y_sc, int64_t y_st,
                         const float* xlen, int B, int C, int T, int T_out, float eps, void* stream);

/* nn.InstanceNorm1d(track_running_stats = True).forward as MaskedInstanceNorm1d reaches it (models.py:711, JasperNetSmallTrainableInstanceNorm
 * 1394-1404; un-masked, no affine).  training != 0: y = (x - mean) / sqrt(biased var + eps) per (utterance, channel), and
 * running = (1 - momentum) running + momentum * batch mean of the instance means / UNBIASED variances (num_batches_tracked += 1 when given:
 * torch's InstanceNorm never counts, so the host mirror passes NULL);
 * stats_workspace: 2 B C floats.  training == 0: y = (x - running_mean) / sqrt(running_var + eps), nothing written back. */
int convasr_instnorm_running_fwd(const void* x, int x_dtype, int64_t x_sb, int64_t x_sc, int64_t x_st,
                                 void* y, int y_dtype, int64_t y_sb, int64_t y_sc, int64_t y_st,
                                 int B, int C, int T, int T_out, float eps, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float momentum, int training, float* stats_workspace, void* stream);

/* ---- compute_output_lengths: models.py:611-614 ------------------------------------------------------------------ */

/* out[b] = ceil(xlen[b] * T) in fp32 arithmetic, as (lengths_fraction * T).ceil().long() evaluates it (xlen NULL: T). */
int convasr_output_lengths(const float* xlen, int B, int T, int64_t* out, void* stream);

/* ---- Conv1d: models.py:47-77 (ConvSamePadding -> nn.Conv1d), models.py:23-44 (Decoder 1x1) --------------------- */

/* cout_pad(cout): rows of the packed weight (multiple of the kernel's N tile). */
int convasr_conv_cout_pad(int cout);

/* Memory layout of an fp32 conv parameter / its gradient, logical shape (Cout, Cin, K):
 *   CONVASR_W_REFERENCE  element (co, ci, k) at (co*Cin + ci)*K + k  -- torch-contiguous, what the reference's state dict holds;
 *   CONVASR_W_KMAJOR     element (co, ci, k) at (k*Cout + co)*Cin + ci -- tap-major, the element order of the packed compute
 *                        copies and of the weight-gradient slabs: the layout the MI355X training arena keeps masters and
 *                        gradients in (a torch view of shape (Cout, Cin, K) with strides (Cin, 1, Cout*Cin)), so that the packed
 *                        forward weights are a cast of the master and the split-K combine is a streaming sum. */
#define CONVASR_W_REFERENCE 0
#define CONVASR_W_KMAJOR 1

/* (Cout, Cin, K) fp32 parameter (in layout w_layout) -> packed compute-dtype copies:
 *   packed_fwd  [K][cout_pad(Cout)][Cin]  : packed_fwd[k][co][ci]   = w[co][ci][k]        (required)
 *   packed_dgrad[K][cout_pad(Cin)][Cout]  : packed_dgrad[k][ci][co] = w[co][ci][K-1-k]    (optional; input gradient = conv
 *                                           with flipped taps; derived from packed_fwd by a tiled transpose)
 * Only rows < Cout (resp. < Cin) are written: the caller zero-fills the buffers once so the padded rows read as zeros.
 * w == NULL: packed_fwd is already current (e.g. written by convasr_sgd_step's bf16 mirror) and only packed_dgrad is rebuilt. */
int convasr_pack_conv_weight(const float* w, void* packed_fwd, void* packed_dgrad, int dtype, int Cout, int Cin, int K, int w_layout, void* stream);

/* y[b,t,co] = epilogue( sum_{k,ci} x[b, t*stride + k*dil - pad, ci] * wp[k][co][ci] ), zero outside [0, Tin).
 * epilogue: acc -> (+ bias[co] if bias) -> (stats of that value over the tile's valid (b,t) if stats) -> (* scale[co] + shift[co]
 * if scale) -> activation -> (zero frames t >= ceil(xlen[b]*Tout) if xlen) -> store as y_dtype.
 * stats (may be NULL): per-m-tile partial sums, [rows][2][Cout] doubles (sum then sum of squares), rows written = *stats_rows
 * (<= convasr_conv_stats_max_rows(B, Tout), the size to allocate).  No atomics: convasr_bn_finalize / convasr_reduce_rows add the
 * rows in a fixed order, so batch statistics are bit-identical from run to run.
 * Used for forward (mode FWD weights) and for dgrad (mode DGRAD weights, stride must be 1, pad' = dil*(K-1) - pad).
 * Tout is at most (Tin + 2 pad - dil (K-1) - 1) / stride + 1 (F.conv1d's output length); a smaller Tout computes the first Tout frames only. */
int convasr_conv1d_fwd(const void* x, const void* wp, void* y, int x_dtype, int y_dtype,
                       int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil, int pad,
                       const float* bias, double* stats, const float* scale, const float* shift,
                       int act, float act_lo, float act_hi, const float* xlen, int* stats_rows, void* stream);
int convasr_conv_stats_max_rows(int B, int Tout);

/* Split-K form of convasr_conv1d_fwd for launches of a few tiles -- online inference, transcribe.py:140 / benchmark_online.py:125 (B = 1 x 6 s is
 * 4-16 workgroups per layer on 256 CUs, each reducing all of Cin x K alone): the reduction is cut over the 64-channel input blocks, workgroup
 * (tile, split) stores an fp32 partial tile into `workspace`, a second streaming kernel adds the partials in split order (deterministic) and
 * runs the same epilogue (bias, scale / shift, activation, length mask, rounding to y_dtype).  stride 1, no statistics; 16-bit input (the LDS-DMA
 * kernel, 64-channel blocks) or fp32 in and out (the exact-fp32 kernel, 32-channel slabs; the partials are added in fp64).
 *   convasr_conv1d_fwd_splitk_plan: the number of splits for this geometry (1: not worth it -- from a quarter of the CUs' worth of tiles up -- or
 *     outside the envelope: call convasr_conv1d_fwd) and the workspace bytes (splits x B x Tout x Cout fp32);
 *   convasr_conv1d_fwd_splitk: `splits` must be the plan's answer (>= 2). */
int convasr_conv1d_fwd_splitk_plan(int x_dtype, int B, int Cin, int Cout, int Tout, int K, int64_t* workspace_bytes);
int convasr_conv1d_fwd_splitk(const void* x, const void* wp, void* y, int x_dtype, int y_dtype,
                              int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int pad,
                              const float* bias, const float* scale, const float* shift, int act, float act_lo, float act_hi,
                              const float* xlen, int splits, void* workspace, void* stream);

/* Stride-2 fold (the prologue conv of every model, models.py:312: ConvBn1d(kernel_size_prologue = 11, stride = 2) on the 64 mel
 * channels).  A stride-2, dilation-1 conv over an EVEN number of frames Tin equals the stride-1 conv with K' taps and padding P'
 * over the same memory read as (Tin / 2) frames of 2 Cin channels (frames 2r, 2r+1 side by side -- a view, no copy):
 *   P' = ceil(pad / 2), s0 = 2 P' - pad, K' = (K - 1 + s0) / 2 + 1,   w'[co][p Cin + ci][j] = w[co][ci][2 j + p - s0] (0 outside [0, K)),
 * called as convasr_conv1d_fwd(x, packed', y, ..., Cin = 2 Cin, Tin = Tin / 2, Tout = the stride-2 conv's own Tout, K', 1, 1, P') and
 * convasr_conv1d_wgrad the same way.  K = 11, pad = 5, Cin = 64 gives K' = 6, P' = 3, 128 channels: inside the envelope of the
 * LDS-DMA kernels, which a stride-2 / 64-channel problem is not.
 *   convasr_fold2_geometry      K, pad -> K', P'
 *   convasr_fold2_pack_weight   fp32 (Cout, Cin, K) parameter in layout w_layout -> packed forward operand [K'][cout_pad(Cout)][2 Cin]
 *   convasr_fold2_unfold_wgrad  fp32 gradient of the folded conv, tap-major [K'][Cout][2 Cin] (convasr_conv1d_wgrad with
 *                               CONVASR_W_KMAJOR) -> dw (+)= the parameter's gradient in layout dw_layout */
int convasr_fold2_geometry(int K, int pad, int* K_folded, int* pad_folded);
int convasr_fold2_pack_weight(const float* w, int w_layout, void* packed_fwd, int dtype, int Cout, int Cin, int K, int pad, void* stream);
int convasr_fold2_unfold_wgrad(const float* dw_folded, float* dw, int dw_layout, int Cout, int Cin, int K, int pad, int accumulate, void* stream);

/* A/B and test hook: bit 0 = 0 routes bf16 / fp16 launches through the general register-staged kernels instead of the LDS-DMA kernels;
 * bits 8 and up are diagnostic switches documented where they are read (e.g. 8192 << 8: no one-tap kernel, 32768 << 8: the workgroup
 * form of convasr_ctc_alignment for every target length).  Returns the previous setting of bit 0; the switches are cleared by the next call. */
int convasr_debug_set_conv_v2(int enable);

/* Bytes of fp32 workspace convasr_conv1d_wgrad needs (split-K partial slabs). */
int64_t convasr_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil);

/* dw[co][ci][k] (+)= sum_{b,t} dy[b,t,co] * x[b, t*stride + k*dil - pad, ci]; dw is the fp32 (Cout, Cin, K) gradient in layout
 * dw_layout (CONVASR_W_REFERENCE: the reference's parameter layout; CONVASR_W_KMAJOR: the training arena's).
 * accumulate != 0 adds to dw.  dbias (Cout fp32, may be NULL) = sum dy. */
int convasr_conv1d_wgrad(const void* x, const void* dy, float* dw, float* dbias, void* workspace, int dtype,
                         int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil, int pad,
                         int accumulate, int dw_layout, void* stream);

/* ---- grouped Conv1d: models.py:50-64 (ConvSamePadding(separable = True): nn.Conv1d(Cin, Cout, K, groups = G) -> ReLU -> 1x1 conv), ---------
 * ---- JasperNetSeparable (models.py:1372-1374, G = 128).  The 1x1 half is convasr_conv1d_*; this is the grouped half. ------------------ */

/* y[b,t,co] = act(bias[co] + sum_k sum_{j < Cin/G} x[b, t stride + k - pad, (co / (Cout/G)) Cin/G + j] w[co][j][k]); relu != 0: act = ReLU.
 * x, y channels-last (B, T, C) of `dtype` (F32 / BF16 / F16); w the fp32 (Cout, Cin / G, K) parameter with element strides
 * (w_sco, w_sj, w_sk) -- torch-contiguous or the training arena's tap-major view; at most 8 channels per group on either side. */
int convasr_grouped_conv1d_fwd(const void* x, const float* w, int64_t w_sco, int64_t w_sj, int64_t w_sk, const float* bias, void* y, int dtype,
                               int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int groups, int relu, void* stream);
/* dx = input gradient (stride 1 only); y_act (may be NULL): the forward's post-ReLU output -- the gradient passes where it is > 0. */
int convasr_grouped_conv1d_dgrad(const void* dy, const void* y_act, const float* w, int64_t w_sco, int64_t w_sj, int64_t w_sk, void* dx, int dtype,
                                 int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int groups, void* stream);
/* dw (+)= weight gradient (same strides as w), dbias (may be NULL) (+)= sum of the gated dy; per-utterance partial sums go through
 * `workspace` and are added in a fixed order (no atomics). */
int64_t convasr_grouped_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int K, int groups);
int convasr_grouped_conv1d_wgrad(const void* x, const void* dy, const void* y_act, float* dw, int64_t w_sco, int64_t w_sj, int64_t w_sk, float* dbias,
                                 void* workspace, int dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int groups,
                                 int accumulate, void* stream);

/* ---- BatchNorm1d + ResidualActivation + temporal mask: models.py:111-114, 127-139, 357-371, 436-443 ----------- */

/* From the conv epilogue's stats (sum, sumsq over n = B*T values per channel): batch mean / biased var ->
 * scale = gamma * invstd, shift = beta - mean * scale; mean/invstd saved for backward; running stats updated
 * with momentum (running_var with the unbiased estimate), exactly nn.BatchNorm1d training semantics.
 * stats: [stats_rows][2][C] partial sums as the conv epilogue writes them (stats_rows = 1: plain totals), added in a fixed order.
 * num_batches_tracked (may be NULL) is incremented. */
int convasr_bn_finalize(const double* stats, int stats_rows, int64_t n, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* invstd, float* scale, float* shift, int C,
                        int64_t* num_batches_tracked, void* stream);
/* out[c] = sum over rows of part[r][c] (fp64, the same fixed order): totals of a partial-sum buffer for callers that want them. */
int convasr_reduce_rows(const double* part, int rows, int width, double* out, void* stream);

/* eval: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale. */
int convasr_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                float eps, float* scale, float* shift, int C, void* stream);

/* z = mask_t( dropout( act( y * scale[c] + shift[c] + sum_r (res_r * rscale_r[c] + rshift_r[c]) ) ) ).
 * y, z, res_r channels-last (B, T, C) of `dtype`.  n_res <= 12; rscale_r NULL means the residual is added as is.
 * dropout_p == 0 disables dropout; otherwise a counter-based hash of (seed, offset, element index) decides.
 * step_key (may be NULL; here and in the three backward entry points that re-derive the mask): one device word XORed into the
 * whitened seed when the kernel runs -- the per-step key convasr_step_begin() advances -- so that a training step captured into a
 * HIP graph (whose by-value arguments are frozen) draws fresh masks at every replay, the same ones an eager step draws.
 * gate (may be NULL; B*T*C/8 bytes; activations with derivative 0 or 1 only: none / relu / hardtanh): bit (e & 7) of byte (e >> 3) for
 * element e = (b*T + t)*C + c is set iff the gradient passes the element -- inside the activation's linear range, kept by dropout,
 * frame not masked.  The backward passes below take it back in: g = dz * (1 / (1 - p)) or 0, with no re-derivation of the
 * pre-activation, no hash and no frame arithmetic (bit-identical to the re-derived g). */
int convasr_bn_act_fwd(const void* y, void* z, int dtype, const float* scale, const float* shift,
                       int n_res, const void* const* res, const float* const* rscale, const float* const* rshift,
                       int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key,
                       const float* xlen, int B, int T, int C, uint8_t* gate, void* stream);

/* Backward of the above, pass 1.  g = dz * mask * dropout * act'(pre), pre recomputed from y (and residuals).
 * Writes g (same dtype; g may be NULL when pass 2 recomputes it from dz) and reduces per channel: sums[0..C) += sum g, sums[C..2C) += sum g * xhat with
 * xhat = (y - mean) * invstd, and for each of the first two residuals r with BN: rsums_r likewise w.r.t. (res_r, rmean_r,
 * rinvstd_r).  sums / rsums_r (2*C doubles each) are WRITTEN (block partials go through `workspace`, then an fp64 sum:
 * deterministic, no contended atomics).  workspace: convasr_bn_bwd_workspace_bytes(B, T, C) bytes.
 * Optional outputs for the main BN (need mean / invstd): coef (3*C floats: dy = coef[c]*g + coef[C+c]*y + coef[2C+c], the
 * batch-norm training backward with `gamma` folded in), dgamma = sum g*xhat, dbeta = sum g (added to when accumulate).
 * gate (may be NULL): the one-bit gates convasr_bn_act_fwd stored; then g = gate ? dz / (1 - p) : 0 with no re-derivation (n_res must be
 * 0, g NULL, mean / invstd given): three streams in, two sums out. */
int64_t convasr_bn_bwd_workspace_bytes(int B, int T, int C);
int convasr_bn_act_bwd_reduce(const void* dz, const void* y, void* g, int dtype, const float* scale, const float* shift,
                              const float* mean, const float* invstd,
                              int n_res, const void* const* res, const float* const* rscale, const float* const* rshift,
                              const float* const* rmean, const float* const* rinvstd, double* const* rsums,
                              int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key,
                              const float* xlen, double* sums, void* workspace, const float* gamma, float* coef, float* dgamma, float* dbeta,
                              int accumulate, int B, int T, int C, const uint8_t* gate, void* stream);

/* Backward pass 2 in coefficient form: dy = coef[c]*g + coef[C+c]*y + coef[2C+c].  from_dz != 0: g is recomputed on the fly
 * from dz (activation derivative, dropout, temporal mask; no residuals) so pass 1 need not materialise it; gate (may be NULL): the
 * one-bit gates convasr_bn_act_fwd stored, used instead of the re-derivation (dropout_p still supplies the 1 / (1 - p)). */
int convasr_bn_act_bwd_apply(const void* dz_or_g, const void* y, void* dy, int dtype, const float* coef, int from_dz,
                             const float* scale, const float* shift, int act, float act_lo, float act_hi, float dropout_p,
                             uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C, const uint8_t* gate, void* stream);

/* Backward pass 2: dy = gamma * invstd * (g - sum_g / n - xhat * sum_gxhat / n)  (batch-norm training backward);
 * dgamma = sum_gxhat, dbeta = sum_g (written, or added when accumulate).  In place allowed (dy == g). */
int convasr_bn_bwd_apply(const void* g, const void* y, void* dy, int dtype, const float* gamma, const float* mean,
                         const float* invstd, const double* sums, float* dgamma, float* dbeta, int accumulate,
                         int B, int T, int C, void* stream);

/* ---- head: models.py:316 (log_softmax), 323 (F.ctc_loss), 645-657 (entropy), transcript_generators.py:27 (argmax) */

/* logits/log_probs channels-last (B, T, C) fp32. */
int convasr_log_softmax_fwd(const float* logits, float* log_probs, int64_t rows, int C, void* stream);
/* dlogits = g - exp(lp) * sum_c g */
int convasr_log_softmax_bwd(const float* grad_lp, const float* log_probs, float* dlogits, int64_t rows, int C, void* stream);

int64_t convasr_ctc_workspace_bytes(int B, int T, int S_max);
/* nll[b] = -log p(targets[b,:ylen[b]] | log_probs[b,:olen[b]]) with blank = `blank`, +inf when infeasible
 * (zero_infinity=False).  grad (B,T,C) = d nll / d log_probs as ATen defines it: exp(lp) - posterior for t < olen,
 * 0 for t >= olen (may be NULL: forward only).  targets: (B, S_max) int64, olen/ylen int64 (the reference's dtypes).
 * Envelope: S_max <= 1,023 labels, T <= ~13,000 frames (the per-frame hand-over slots of the two-wave sweeps live in LDS); beyond it
 * the call fails with CONVASR_EUNSUPPORTED / an invalid-argument error and launches nothing. */
int convasr_ctc_loss(const float* log_probs, const int64_t* targets, const int64_t* olen, const int64_t* ylen,
                     float* nll, float* grad, void* workspace, int B, int T, int C, int S_max, int blank, void* stream);
/* out[b,t,c] = grad[b,t,c] * gscale[b]   (chain rule for reduction='none'); with gdiv != NULL the factor is
 * gscale[b] / (float)gdiv[b * gdiv_stride]: the "/ ylen[:, 0]" of models.py:323 folded into the same pass.  gscale NULL (gdiv given):
 * the factor is 1 / gdiv -- the same division in the forward direction (per_b = 1: nll[b] / ylen[b, 0]). */
int convasr_scale_rows(const float* grad, const float* gscale, const int64_t* gdiv, int64_t gdiv_stride, float* out, int B, int64_t per_b, void* stream);

/* The scalar bookkeeping of one training iteration (train.py:754-756, 769) in one launch: loss_vec[b] = per-utterance loss
 * (models.py:323), w[b] = ylen[b * ylen_stride] (target lengths), entropy[b] (may be NULL) = models.entropy per utterance:
 *   out3[0] = mean(loss_vec * w) / accumulate_iterations, out3[1] = mean(loss_vec), out3[2] = mean(entropy);
 *   grad_loss_vec[b] (may be NULL) = d out3[0] / d loss_vec[b] = ((1 / accumulate_iterations) / B) * w[b];
 *   skipped (may be NULL, one byte) = out3[1] is inf or NaN (the gate of train.py:769).
 * loss_scaler (may be NULL): a dynamic loss scaler's state (below); grad_loss_vec is multiplied by its current scale, i.e. backward is
 * seeded with the gradient of the SCALED loss (`with apex.amp.scale_loss(loss, optimizer) as scaled_loss: scaled_loss.backward()`,
 * train.py:770-772).  metric_scale multiplies out3[1] and out3[2] (1 for one process; 1 / world size in data-parallel runs, so that the
 * SUM all-reduce of train.py:759-760 yields the mean of train.py:761-762 with no further kernel); skipped looks at the local mean. */
int convasr_loss_head(const float* loss_vec, const int64_t* ylen, int64_t ylen_stride, const float* entropy, int B, float accumulate_iterations, float* out3,
                      float* grad_loss_vec, unsigned char* skipped, const float* loss_scaler, float metric_scale, void* stream);

/* Dynamic loss scaler for fp16 training (the role of apex.amp's LossScaler, reference models.py:744-762 `apex.amp.initialize(opt_level)`
 * and train.py:770-779): CONVASR_LOSS_SCALER_FLOATS floats on the device,
 *   [0] scale  [1] clean steps since the last change  [2] did the last step overflow (0 / 1)  [3] growth window (0 = static scale)
 *   [4] min scale  [5] max scale  [6] factor  [7] number of steps skipped for overflow so far
 * read by convasr_loss_head (backward seed x scale), convasr_sumsq (the reported norm is that of the UNSCALED gradient, what
 * clip_grad_norm_ sees on apex's master gradients) and the fused optimizer steps (gradients x 1 / scale; a non-finite sum of squares =
 * overflow: parameters and optimizer state stay untouched).  The optimizer step writes the next state into a SECOND buffer
 * (scaler_in != scaler_out; the caller swaps them per step) with apex's update_scale(): overflow -> scale = max(min, scale / factor),
 * clean = 0; otherwise ++clean, and at clean == window: scale = min(max, scale * factor), clean = 0.  A step gated by a non-finite
 * LOSS (loss_gate) copies the state unchanged: the reference never enters scale_loss on such an iteration. */
#define CONVASR_LOSS_SCALER_FLOATS 8

/* ent[b] = sum_{t<olen} -sum_c p log p / (eps + olen[b])  (olen NULL: mean over T). */
int convasr_entropy(const float* log_probs, const int64_t* olen, float* ent, int B, int T, int C, float eps, void* stream);
/* weighted_mean_entropy (models.py:660-682, logged by train.py:137-139): per frame e = -sum_c p log p, w = 1 - p[eps_id]; out[b] =
 * sum_{t<olen} e w / (eps + sum_{t<olen} w)  (olen NULL: all T frames). */
int convasr_weighted_mean_entropy(const float* log_probs, const int64_t* olen, float* out, int B, int T, int C, int eps_id, float eps, void* stream);
/* idx[b,t] = argmax_c log_probs[b,t,c] (first maximum, like torch.argmax on CPU). */
int convasr_argmax(const float* log_probs, int64_t* idx, int64_t rows, int C, void* stream);

/* ---- optimizer: train.py:777-782 (clip_grad_norm_, SGD step), optimizers.py:66-90 (NovoGrad) ------------------- */

/* sumsq[0] = sum g^2 over n fp32 values, in double, added in a fixed order (partial sums in `workspace`, no atomics). */
int64_t convasr_sumsq_workspace_bytes(void);
/* norm_out (may be NULL) receives (float)(sqrt(sumsq) * norm_scale) (/ the loss scaler's scale when loss_scaler != NULL): what clip_grad_norm_ returns. */
int convasr_sumsq(const float* g, int64_t n, double* sumsq, void* workspace, float* norm_out, float norm_scale, const float* loss_scaler, void* stream);
/* torch.optim.SGD step with clip folded in: c = min(1, max_norm / (sqrt(sumsq) + 1e-6)) (c = 1 if sumsq NULL);
 * g' = c*g + wd*p; buf = first ? g' : mom*buf + g'; p -= lr * (nesterov ? g' + mom*buf : buf).
 * If grad_out != NULL the clipped gradient c*g is written back (what clip_grad_norm_ leaves in .grad).
 * grad_scale multiplies g before everything else (1 / world size when g holds rank-SUMMED gradients: the mean is never
 * materialised; sumsq is then the sum of squares of the summed gradient).
 * If loss_gate != NULL and *loss_gate (a device float, the all-reduced loss) is inf or NaN the launch changes nothing: the
 * reference's "skip the step on a non-finite loss" (train.py:769-772) without a host round trip in the middle of the step.
 * p16 (may be NULL): n values of p16_dtype (CONVASR_BF16 or CONVASR_F16), receives the updated parameters rounded to that type -- with
 * K-major masters that mirror IS the packed forward weight of every conv whose Cout is a multiple of the kernel's N tile (no
 * per-step packing launches).  scaler_in / scaler_out (both or neither): the dynamic loss scaler above; needs sumsq.
 * lr_dev (may be NULL; likewise in the two optimizers below): one device float that replaces `lr` when the kernel runs -- a captured
 * step graph follows the host's learning-rate schedule (optimizers.py:13-63, train.py:783) through it. */
int convasr_sgd_step(float* p, const float* g, float* buf, float* grad_out, int64_t n, const double* sumsq, float max_norm,
                     float lr, float momentum, float weight_decay, int nesterov, int first, const float* loss_gate, float grad_scale,
                     void* p16, int p16_dtype, const float* scaler_in, float* scaler_out, const float* lr_dev, void* stream);
/* torch.optim.AdamW step (train.py:663-668; decoupled weight decay, no amsgrad) with the same folded-in pieces as convasr_sgd_step:
 * g' = c*grad_scale*g (c from sumsq / max_norm as above); p *= 1 - lr*wd; m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t = step_in[0] + 1.  step_in / step_out: one device float each, distinct
 * buffers the caller swaps per call: the number of steps APPLIED so far (a launch gated by loss_gate or skipped by the loss scaler
 * writes step_out[0] = step_in[0] and changes nothing else, so the bias corrections follow the applied steps like torch's state['step']).
 * loss_gate, grad_scale, p16 / p16_dtype, scaler_in / scaler_out: as in convasr_sgd_step. */
int convasr_adamw_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, const double* sumsq, float max_norm, float lr,
                       float beta1, float beta2, float eps, float weight_decay, const float* step_in, float* step_out, const float* loss_gate,
                       float grad_scale, void* p16, int p16_dtype, const float* scaler_in, float* scaler_out, const float* lr_dev, void* stream);

/* Fused backward step (bf16 / fp16 storage `dtype`, stride 1): dx = dgrad(dy) of one Conv1d -- i.e. dz of the Conv+BN+activation layer that produced
 * this conv's input -- plus pass 1 of THAT layer's batch-norm backward in the epilogue, on the tile just produced:
 * g = dx * act'(bn_y*bn_scale+bn_shift) * dropout * mask(bn_xlen); bn_sums[c] += sum g, bn_sums[C+c] += sum g*(bn_y-mean)*invstd

 * (C = Cin of this conv) written as per-m-tile partial rows bn_sums[rows][2][C], *bn_rows rows (<= convasr_conv_stats_max_rows(B, T_dx)).  It replaces
 * convasr_conv1d_fwd(dy, packed_dgrad) + convasr_bn_act_bwd_reduce(write_g = 0) (models.py:111-139 backward) when dx has this
 * single consumer.  dy is (B, T_dy, Cout), dx and bn_y are (B, T_dx, Cin); pad = dil*(K-1) - padding of the forward conv.  Returns 1 (nothing launched) when the shape is outside the
 * LDS-DMA kernel's envelope (Cout % 64 != 0): run the two calls separately.  bn_gate (may be NULL): that layer's one-bit gates
 * from convasr_bn_act_fwd, used instead of re-deriving act' / dropout / mask per element in the epilogue. */
int convasr_conv1d_dgrad_bn_reduce(const void* dy, const void* packed_dgrad, void* dx, int dtype, int B, int Cout, int Cin, int T_dy, int T_dx, int K,
                                   int dil, int pad, const void* bn_y, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                   const float* bn_invstd, int bn_act, float bn_act_lo, float bn_act_hi, float dropout_p, uint64_t seed,
                                   uint64_t offset, const uint64_t* step_key, const float* bn_xlen, double* bn_sums, int* bn_rows, const uint8_t* bn_gate, void* stream);
/* coef / dgamma / dbeta from those partial rows (the second half of convasr_bn_act_bwd_reduce), added in a fixed order; n = B*T. */
int convasr_bn_bwd_finalize(const double* sums, int sums_rows, const float* gamma, const float* mean, const float* invstd, float* coef,
                            float* dgamma, float* dbeta, int accumulate, int64_t n, int C, void* stream);

/* ---- grouped one-tap launches: the residual branches of a dense block (models.py:107-110, 129-131) ------------------ */

/* n (<= 12) independent Conv1d(kernel_size = 1) problems over the SAME B * T frames in one dispatch (two when some problems have a long
 * reduction into few channels): y_i (+)= x_i * w_i^T (+ bias_i).  x_i channels-last (B, T, cin[i]) and y_i (B, T, cout[i]) of `dtype`
 * (CONVASR_BF16 / CONVASR_F16), w_i the packed forward layout [1][cout[i]][cin[i]] (convasr_pack_conv_weight; for an input gradient: the
 * packed dgrad layout with the roles of the channel counts swapped), bias (array or NULL; entries may be NULL), stats (array or NULL;
 * entries may be NULL): per-m-tile partial rows exactly as convasr_conv1d_fwd writes them, *stats_rows rows each; accumulate (array or
 * NULL): entry != 0 adds into y_i (the sum of the two stored 16-bit values, rounded once) -- the input gradients of the branches land in
 * the tapped block output's gradient without autograd's pairwise adds.  Each problem is computed exactly as convasr_conv1d_fwd computes
 * it alone (bit-identical per element).  cin[i] % 64 == 0, cout[i] % 128 == 0, else CONVASR_EUNSUPPORTED (launch them one by one). */
int convasr_conv1x1_grouped(int n, const void* const* x, const void* const* w, void* const* y, const float* const* bias, double* const* stats,
                            const int* cin, const int* cout, const int* accumulate, int dtype, int B, int T, int* stats_rows, void* stream);
/* The weight gradients of the same n branches, dw_i[co][ci] (+)= sum over (b, t) of dy_i[b, t, co] * x_i[b, t, ci] (fp32, cout[i] x cin[i],
 * the memory order of a one-tap (Cout, Cin, 1) parameter in either layout), in ONE dispatch of the LDS-DMA weight-gradient kernel over all
 * problems' (co tile, ci tile, split) units + ONE streaming combine of the split-K slabs (deterministic: fixed order, no atomics).
 * zero (array or NULL; entries may be NULL): cout[i] floats set to zero by the combine -- the branch conv's bias gradient, identically zero
 * ahead of a train-mode batch norm.  accumulate (array or NULL): add into dw_i.  workspace: convasr_wgrad1x1_grouped_workspace_bytes()
 * bytes.  cin[i] % 128 == 0 and cout[i] % 128 == 0, else CONVASR_EUNSUPPORTED. */
int64_t convasr_wgrad1x1_grouped_workspace_bytes(int n, const int* cin, const int* cout, int B, int T);
int convasr_wgrad1x1_grouped(int n, const void* const* x, const void* const* dy, float* const* dw, float* const* zero, const int* cin, const int* cout,
                             const int* accumulate, void* workspace, int dtype, int B, int T, void* stream);
/* The batch norms of those branches (nn.BatchNorm1d at models.py:110, one per branch, all of the block's channel count C), grouped the same
 * way.  convasr_bn_finalize_grouped: convasr_bn_finalize for `count` (<= 13) batch norms in one launch; out[i] receives 4 C floats (mean,
 * invstd, scale, shift), stats[i] has stats_rows partial rows.  convasr_bn_bwd_finalize_grouped: convasr_bn_bwd_finalize for `count` sets of
 * sums (sums_rows[i] partial rows each).  convasr_bn_bwd_apply_grouped: dy_i = coef_i[c] * g + coef_i[C + c] * y_i + coef_i[2 C + c] for
 * `count` (y_i, coef_i, dy_i) triples sharing ONE g, which is read once (count + 1 launches of convasr_bn_act_bwd_apply(from_dz = 0) read it
 * count + 1 times); 16-bit storage. */
int convasr_bn_finalize_grouped(int count, const double* const* stats, int stats_rows, int64_t n, const float* const* gamma, const float* const* beta,
                                float* const* running_mean, float* const* running_var, const float* momentum, const float* eps, float* const* out,
                                int64_t* const* num_batches_tracked, int C, void* stream);
int convasr_bn_bwd_finalize_grouped(int count, const double* const* sums, const int* sums_rows, const float* const* gamma, const float* const* mean,
                                    const float* const* invstd, float* const* coef, float* const* dgamma, float* const* dbeta, const int* accumulate,
                                    int64_t n, int C, void* stream);
int convasr_bn_bwd_apply_grouped(const void* g, int count, const void* const* y, const float* const* coef, void* const* dy, int dtype, int B, int T, int C, void* stream);
/* Pass 1 of a dense block's backward (models.py:127-139 backward) in ONE sweep, from the one-bit gates convasr_bn_act_fwd stored for the block's
 * output: g = gate ? dz / (1 - p) : 0 is written once, and for each of the `count` (<= 13) batch norms that feed the block's sum -- the main
 * one and the residual branches' (y_i its input, mean_i / invstd_i its batch statistics) -- sum g and sum g * (y_i - mean_i) * invstd_i
 * are formed (block partials in `workspace`, convasr_bn_bwd_reduce_many_workspace_bytes() bytes, then an fp64 sum in a fixed order: no
 * atomics) and turned into coef_i (3 C floats, as convasr_bn_act_bwd_reduce's), dgamma_i, dbeta_i (added to when accumulate[i]).  Replaces
 * convasr_bn_act_bwd_reduce (which re-derives the pre-activation from all residual inputs) + one further sweep per two branches. */
int64_t convasr_bn_bwd_reduce_many_workspace_bytes(int count, int B, int T, int C);
int convasr_bn_bwd_reduce_many(const void* dz, const uint8_t* gate, float dropout_p, void* g, int count, const void* const* y, const float* const* mean,
                               const float* const* invstd, const float* const* gamma, float* const* coef, float* const* dgamma, float* const* dbeta,
                               const int* accumulate, void* workspace, int dtype, int B, int T, int C, void* stream);
/* out = a + b over n 16-bit values (n % 8 == 0; in place allowed): the one explicit add a tapped block output's gradient needs (the main
 * path's input gradient + the branches' accumulated ones), in place of autograd's InputBuffer accumulation (models.py:129-131 backward). */
int convasr_add16(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);

/* The dgrad operands (convasr_pack_conv_weight's packed_dgrad, from packed_fwd, 16-bit) of MANY layers in one launch.  items: n_items
 * records in DEVICE memory, convasr_pack_dgrad_item_bytes() bytes each: {const void* packed_fwd; void* packed_dgrad; int Cout, Cin, K,
 * co_pad (= convasr_conv_cout_pad(Cout)), ci_pad (= convasr_conv_cout_pad(Cin)), first;} with first = the running sum of the earlier items'
 * K * ceil(Cout / 64) * ceil(Cin / 64) blocks and total_blocks = that sum over all items; Cout and Cin even. */
int convasr_pack_dgrad_item_bytes(void);
int convasr_pack_dgrad_grouped(const void* items, int n_items, int total_blocks, void* stream);

/* out[c] (+)= sum over the rows of a channels-last (rows, C) matrix of `dtype`: the bias gradient of nn.Conv1d (models.py:26, the decoder head) on
 * its own -- two launches, per-block partial rows in `workspace` (convasr_colsum_workspace_bytes) added in a fixed order: no atomics. */
int64_t convasr_colsum_workspace_bytes(int64_t rows, int C);
int convasr_colsum(const void* y, int dtype, int64_t rows, int C, float* out, void* workspace, int accumulate, void* stream);

/* dst[i] = src[i] * scale over n elements (n % 8 == 0), fp32 -> CONVASR_BF16 / CONVASR_F16 or back: the two ends of a 16-bit gradient
 * exchange (apex O2 keeps and all-reduces fp16 model gradients, models.py:744-762 / train.py:771): a bucket of the fp32 gradient arena is
 * packed into a 16-bit send buffer (scale = 1 / world size: the mean, formed before the sum so that fp16 cannot overflow in it),
 * all-reduced by RCCL at half the bytes, and unpacked into the arena. */
int convasr_cast_scale(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale, void* stream);

/* ---- split-operand ("x3") convs: fp32-class accuracy on the 16-bit matrix pipe (models.py:47-77 computed as nn.Conv1d does in fp32) ---- */

/* A value v travels as hi = rn16(v), lo = rn16(v - hi) and a product as hi*hi + hi*lo + lo*hi: three 16-bit MFMAs, each exact in fp32,
 * fp32 accumulation (bf16 planes: 16 significant bits per operand; fp16 planes: 22, but fp16's range).  The three products ride in the
 * REDUCTION axis of the ordinary conv entry points above, so no conv kernel of its own exists:
 *   convasr_split3: x fp32 channels-last [rows = B * T][C] -> out [rows][3][C] of `dtype` (CONVASR_BF16 / CONVASR_F16), planes
 *     order 0 = (hi, lo, hi) for a conv INPUT, order 1 = (hi, hi, lo) for an OUTPUT GRADIENT.  C % 8 == 0.
 *   forward:  convasr_conv1d_fwd(x3 as a (B, T, 3 Cin) input, packed_fwd of convasr_pack_conv_weight_split3, y fp32, Cin -> 3 Cin);
 *   dgrad:    convasr_conv1d_fwd(dy3 as a (B, T, 3 Cout) input, packed_dgrad of the same call, dx fp32, the usual transposed geometry);
 *   wgrad:    convasr_conv1d_wgrad(x3 read as (B, 3 Tin, Cin), dy3 read as (B, 3 Tout, Cout), dil -> 3 dil, pad -> 3 pad): frame 3 t + p is
 *             plane p of frame t, so the reduction over frames pairs plane p of x with plane p of dy.
 * convasr_pack_conv_weight_split3: (Cout, Cin, K) fp32 parameter in layout w_layout ->
 *   packed_fwd   [K][cout_pad(Cout)][3 Cin] : row (k, co)         = (w_hi[co][.][k], w_hi[co][.][k], w_lo[co][.][k])
 *   packed_dgrad [K][cout_pad(Cin)][3 Cout] : row (K - 1 - k, ci) = (w_hi[.][ci][k], w_lo[.][ci][k], w_hi[.][ci][k])      (dgrad_planes = 3)
 *             or [K][cout_pad(Cin)][Cout]   : row (K - 1 - k, ci) =  w_hi[.][ci][k], the ordinary 16-bit dgrad operand    (dgrad_planes = 1)
 * either may be NULL; rows beyond Cout / Cin are not written (zero-fill once).
 *
 * Split forward, ONE-product backward (compute types 'bf16x3f' / 'f16x3f': the loss carries the split forward's fp32-class accuracy, the
 * gradients the 16-bit path's, as under the reference's apex O2, models.py:744-762; two thirds of the backward's MFMA work are gone):
 *   dy:    convasr_bn_act_bwd_apply_to_half -- the fp32 BN backward with dy rounded once to a dense 16-bit tensor [B * T][C];
 *   dgrad: convasr_conv1d_fwd(dy16, packed_dgrad with dgrad_planes = 1, dx fp32);
 *   wgrad: convasr_conv1d_wgrad_ld -- convasr_conv1d_wgrad (stride 1) over operands whose frames are x_ld / dy_ld elements apart: plane 0 (hi)
 *          of the forward's saved planes x3 read IN PLACE (x_ld = 3 Cin) against dy16.  16-bit storage, Cin % 128 == 0, Cout % 128 == 0,
 *          ld % 8 == 0; CONVASR_EUNSUPPORTED outside -- convasr_conv1d_wgrad_ld_supported answers beforehand (the host then copies the plane out and calls
 *          convasr_conv1d_wgrad). */
int convasr_split3(const float* x, void* out, int dtype, int64_t rows, int C, int order, void* stream);
/* The two fp32 streaming passes whose result is consumed by split convs only, with the split folded in (the separate convasr_split3 pass --
 * 4 bytes read + 6 written per element -- disappears):
 *   convasr_bn_act_fwd_split3:       convasr_bn_act_fwd on fp32 y, the result z stored as its planes z3 [B * T][3][C], order 0 (scale / shift
 *                                    present and a clamp-type activation, i.e. the training launches; else CONVASR_EUNSUPPORTED);
 *   convasr_bn_act_bwd_apply_split3: convasr_bn_act_bwd_apply on fp32 inputs, dy stored as its planes dy3 [B * T][3][C], order 1.
 * Arguments as in the plain entry points, with the plane type (CONVASR_BF16 / CONVASR_F16) in place of `dtype`; the values are the
 * plain pass's values split exactly as convasr_split3 splits them (bit-identical planes). */
int convasr_bn_act_fwd_split3(const void* y, void* z3, int plane_dtype, const float* scale, const float* shift,
                              int n_res, const void* const* res, const float* const* rscale, const float* const* rshift,
                              int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key,
                              const float* xlen, int B, int T, int C, uint8_t* gate, void* stream);
int convasr_bn_act_bwd_apply_split3(const void* dz_or_g, const void* y, void* dy3, int plane_dtype, const float* coef, int from_dz,
                                    const float* scale, const float* shift, int act, float act_lo, float act_hi, float dropout_p,
                                    uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C,
                                    const uint8_t* gate, void* stream);
int convasr_pack_conv_weight_split3(const float* w, int w_layout, void* packed_fwd, void* packed_dgrad, int dgrad_planes, int dtype, int Cout, int Cin, int K, void* stream);
int convasr_bn_act_bwd_apply_to_half(const void* dz_or_g, const void* y, void* dy16, int out_dtype, const float* coef, int from_dz,
                                     const float* scale, const float* shift, int act, float act_lo, float act_hi, float dropout_p,
                                     uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C,
                                     const uint8_t* gate, void* stream);
/* 1 when convasr_conv1d_wgrad_ld takes this geometry, 0 when the caller must make the operands dense and call convasr_conv1d_wgrad (no launch). */
int convasr_conv1d_wgrad_ld_supported(int dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int x_ld, int dy_ld);
int convasr_conv1d_wgrad_ld(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, void* workspace, int dtype,
                            int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int pad, int accumulate, int dw_layout, void* stream);

/* ---- per-step device state: what lets train.py:745-783 replay from a HIP graph ------------------------------------- */

/* state: four device words {dropout seed (the caller's, models.py:365-369 draws from torch's generator), steps begun, key of the
 * current step, reserved}.  Enqueues state[1] += 1; state[2] = mix(state[0] ^ mix(state[1])) -- one thread.  The training step calls it
 * first; &state[2] is the `step_key` of the dropout entry points above.  Eager and graph-replayed steps advance the same words. */
int convasr_step_begin(uint64_t* state, void* stream);
/* dst[0..nbytes) = src[0..nbytes), device to device, on `stream` (a memcpy node under capture): hands a double-buffered device state
 * (loss scaler, NovoGrad EMAs, AdamW step count) back to the buffer a captured graph reads. */
int convasr_copy(const void* src, void* dst, int64_t nbytes, void* stream);

/* ---- SURVEY 8(f) "next" rows ------------------------------------------------------------------------------------ */

/* NovoGrad.step (optimizers.py:66-90) with torch.nn.utils.clip_grad_norm_ (train.py:777) folded in, over a flat arena of n
 * fp32 values cut into n_seg tensors by offsets[n_seg + 1] (device int64; a tensor's padding belongs to it and must hold
 * zero gradients).  Per tensor s: g2 = sum (c*g)^2 with c = min(1, max_norm / (||g||_all + 1e-6)) (c = 1 if max_norm <= 0);
 * ema_out[s] = first ? g2 : ema_in[s]*beta2 + g2*(1-beta2); d = c*g / sqrt(ema_out[s] + eps) (+ wd*p) (* (1-beta1) if
 * dampening); mom = first ? d : mom*beta1 + d; p -= lr*mom.  ema_in / ema_out must be different buffers (the caller swaps
 * them every call); g2 is n_seg doubles of scratch; the per-tensor sums of squares are formed without atomics from a static work table:
 * items[n_items][3] = {tensor, begin, end} cuts every tensor into pieces of at most convasr_novograd_item_elems() elements, in tensor
 * order, seg_first[n_seg + 1] indexes the first piece of each tensor, item_part is n_items doubles of scratch (device int64 / fp64); total_norm (may be NULL) receives ||g||_all; loss_gate and grad_scale as in
 * convasr_sgd_step (a gated call copies ema_in to ema_out and changes nothing else).  first: 1 / 0 decided by the caller, or -1 =
 * decided on the device: ema_in / ema_out then hold n_seg + 1 floats, the last one the number of steps applied so far (the call
 * writes ema_out[n_seg] = ema_in[n_seg] + 1 unless gated), and first = (ema_in[n_seg] == 0) -- a gated first iteration then leaves
 * no optimizer state behind, like the reference, which creates state only when a step runs (optimizers.py:76-80).  p16 / p16_dtype and
 * scaler_in / scaler_out as in convasr_sgd_step (the overflow check uses the per-tensor sums of squares formed here). */
int convasr_novograd_step(float* p, const float* g, float* mom, const float* ema_in, float* ema_out, double* g2, const int64_t* offsets,
                          int n_seg, int64_t n, const int64_t* items, int n_items, const int64_t* seg_first, double* item_part,
                          float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int dampening, int first,
                          const float* loss_gate, float* total_norm, float grad_scale, void* p16, int p16_dtype, const float* scaler_in,
                          float* scaler_out, const float* lr_dev, void* stream);
/* largest item the table may hold (elements) */
int64_t convasr_novograd_item_elems(void);

/* Bytes of back-pointer workspace for the call below (ABI v8: S_max added -- targets longer than 511 labels run one workgroup per
 * utterance instead of one wave and keep 1024 packed words per frame instead of 64). */
int64_t convasr_ctc_alignment_workspace_bytes(int B, int T, int S_max);
/* ctc.alignment (ctc.py:7-75): forced alignment of targets[b, :target_lengths[b]] to log_probs[b, :input_lengths[b]]
 * (log_probs batch-major (B, T, C) fp32; the reference's (T, B, C) tensor permuted).  alignment (B, S_max) int64: the last
 * frame the best path spends in each label's state, 0 for padded labels.  Forward variable = the reference's sum recursion
 * with finfo.min as log-zero over all T frames, back-pointer = first maximum of (stay, s-1, s-2), end state chosen from the
 * column at T-1, walk started at input_lengths[b]-1.  S_max <= 8191 (a recording of ten minutes aligned to its transcript in one call,
 * transcribe.py:176 without segmentation); longer targets return CONVASR_EUNSUPPORTED. */
int convasr_ctc_alignment(const float* log_probs, const int64_t* targets, const int64_t* input_lengths, const int64_t* target_lengths,
                          int64_t* alignment, void* workspace, int B, int T, int C, int S_max, int blank, void* stream);

/* The padding half of AudioTextDataset.collate_fn (datasets.py:320-330) on the device: B ragged samples of `rows` rows each,
 * delivered as one packed buffer (sample b starts at element offsets[b], its rows are lengths[b] long and contiguous), become the
 * zero-padded batch out (B, rows, Tpad).  elem_bytes: 2 (int16 / bf16), 4 (fp32 / int32) or 8 (int64 targets). */
int convasr_collate_pad(const void* packed, const int64_t* offsets, const int64_t* lengths, void* out, int elem_bytes, int B, int rows,
                        int64_t Tpad, void* stream);

#ifdef __cplusplus
}
#endif
#endif
