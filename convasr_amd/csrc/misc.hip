// HBM-bound plumbing kernels: error state, layout conversion, instance norm, log-softmax, entropy, argmax,
// gradient-norm and SGD.  gfx950 only.
#include "common.h"

thread_local char g_convasr_err[512] = {0};

int convasr_fail(int code, const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_convasr_err, sizeof(g_convasr_err), fmt, ap);
	va_end(ap);
	return code;
}

extern "C" int convasr_abi_version(void) { return CONVASR_ABI_VERSION; }
extern "C" const char* convasr_last_error(void) { return g_convasr_err; }

// ------------------------------------------------------------------------------------------------ convert_layout
// 64(c) x 64(t) tile through LDS: reads coalesced along the source's unit-stride axis, writes along the destination's.
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_layout_kernel(const S* __restrict__ src, int64_t ssb, int64_t ssc, int64_t sst,
                                                             D* __restrict__ dst, int64_t dsb, int64_t dsc, int64_t dst_st, int C, int T) {
	__shared__ float tile[64][65];
	const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const bool src_t_fast = (sst == 1) || (ssc != 1), dst_t_fast = (dst_st == 1) || (dsc != 1);
#pragma unroll 4
	for (int i = 0; i < 16; ++i) {
		int c = src_t_fast ? c0 + ty + 4 * i : c0 + tx, t = src_t_fast ? t0 + tx : t0 + ty + 4 * i;
		float v = 0.f;
		if (c < C && t < T) v = Elem<S>::load(src + b * ssb + c * ssc + t * sst);
		tile[c - c0][t - t0] = v;
	}
	__syncthreads();
#pragma unroll 4
	for (int i = 0; i < 16; ++i) {
		int c = dst_t_fast ? c0 + ty + 4 * i : c0 + tx, t = dst_t_fast ? t0 + tx : t0 + ty + 4 * i;
		if (c < C && t < T) Elem<D>::store(dst + b * dsb + c * dsc + t * dst_st, tile[c - c0][t - t0]);
	}
}

extern "C" int convasr_convert_layout(const void* src, int src_dtype, int64_t ssb, int64_t ssc, int64_t sst, void* dst, int dst_dtype,
                                      int64_t dsb, int64_t dsc, int64_t dst_st, int B, int C, int T, void* stream) {
	CONVASR_CHECK_ARG(src && dst && B > 0 && C > 0 && T > 0, "convert_layout: bad arguments");
	dim3 grid((T + 63) / 64, (C + 63) / 64, B);
	hipStream_t s = (hipStream_t)stream;
#define LAUNCH(S, D) hipLaunchKernelGGL((convert_layout_kernel<S, D>), grid, dim3(256), 0, s, (const S*)src, ssb, ssc, sst, (D*)dst, dsb, dsc, dst_st, C, T)
	if (src_dtype == CONVASR_F32 && dst_dtype == CONVASR_F32) LAUNCH(float, float);
	else if (src_dtype == CONVASR_F32 && dst_dtype == CONVASR_BF16) LAUNCH(float, bf16_t);
	else if (src_dtype == CONVASR_BF16 && dst_dtype == CONVASR_F32) LAUNCH(bf16_t, float);
	else if (src_dtype == CONVASR_BF16 && dst_dtype == CONVASR_BF16) LAUNCH(bf16_t, bf16_t);
	else if (src_dtype == CONVASR_F32 && dst_dtype == CONVASR_F16) LAUNCH(float, f16_t);
	else if (src_dtype == CONVASR_F16 && dst_dtype == CONVASR_F32) LAUNCH(f16_t, float);
	else if (src_dtype == CONVASR_F16 && dst_dtype == CONVASR_F16) LAUNCH(f16_t, f16_t);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "convert_layout: dtype %d -> %d", src_dtype, dst_dtype);
#undef LAUNCH
	CONVASR_CHECK_LAUNCH("convert_layout");
	return 0;
}

// ------------------------------------------------------------------------------------------------ instance norm
// One block per (b, 16-channel group): lane = (channel = lane & 15, time lane = lane >> 4), 16 waves stride over time.  Three passes
// over an L2-resident slab: masked mean, masked sum of squared deviations (two-pass, like the reference), normalise.  (16 channels
// per block instead of 64: four times the workgroups -- 256 for 64 mel channels x 64 utterances, one per CU -- for a latency-bound kernel.)
template <typename S, typename D>
__global__ __launch_bounds__(1024) void instnorm_kernel(const S* __restrict__ x, int64_t xsb, int64_t xsc, int64_t xst, D* __restrict__ y,
                                                        int64_t ysb, int64_t ysc, int64_t yst, const float* __restrict__ xlen, int C, int T, int T_out, float eps,
                                                        float* __restrict__ stats_out, const float* __restrict__ fixed_mean, const float* __restrict__ fixed_var) {
	__shared__ float red[64][17];
	__shared__ float bc[16];
	const int b = blockIdx.y, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, tl = threadIdx.x >> 4;  // 64 time lanes
	const int n = valid_len(xlen, b, T);
	const bool ok = c < C;
	const S* xp = x + b * xsb + (ok ? c : 0) * xsc;
	auto block_sum = [&](float v) {  // sum over the 64 time lanes of each channel, in a fixed order
		red[tl][cl] = v;
		__syncthreads();
		if (threadIdx.x < 16) { float s = 0.f; for (int i = 0; i < 64; ++i) s += red[i][threadIdx.x]; bc[threadIdx.x] = s; }
		__syncthreads();
		return bc[cl];
	};
	float mean, stdv;
	if (fixed_mean) {  // nn.InstanceNorm1d in eval mode with running statistics (F.instance_norm, use_input_stats = False)
		mean = ok ? fixed_mean[c] : 0.f;
		stdv = sqrtf((ok ? fixed_var[c] : 1.f) + eps);
	} else {
		float acc = 0.f;
		for (int t = tl; t < n; t += 64) acc += ok ? Elem<S>::load(xp + t * xst) : 0.f;
		mean = block_sum(acc) / (float)n;
		acc = 0.f;
		for (int t = tl; t < n; t += 64) { float d = ok ? Elem<S>::load(xp + t * xst) - mean : 0.f; acc += d * d; }
		const float var = block_sum(acc) / (float)n;
		stdv = sqrtf(var + eps);
		if (stats_out && ok && tl == 0) { stats_out[((int64_t)b * C + c) * 2] = mean; stats_out[((int64_t)b * C + c) * 2 + 1] = var; }
	}
	if (!ok) return;
	D* yp = y + b * ysb + c * ysc;
	for (int t = tl; t < T_out; t += 64) {  // frames T .. T_out - 1 of y: zero padding
		float v = t < n ? (Elem<S>::load(xp + t * xst) - mean) / stdv : 0.f;
		Elem<D>::store(yp + t * yst, v);
	}
}

static int instnorm_launch(const void* x, int x_dtype, int64_t xsb, int64_t xsc, int64_t xst, void* y, int y_dtype, int64_t ysb, int64_t ysc, int64_t yst, const float* xlen,
                           int B, int C, int T, int T_out, float eps, float* stats_out, const float* fixed_mean, const float* fixed_var, hipStream_t s);

extern "C" int convasr_instnorm_fwd(const void* x, int x_dtype, int64_t xsb, int64_t xsc, int64_t xst, void* y, int y_dtype, int64_t ysb,
                                    int64_t ysc, int64_t yst, const float* xlen, int B, int C, int T, int T_out, float eps, void* stream) {
	CONVASR_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && T_out >= T, "instnorm_fwd: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	return instnorm_launch(x, x_dtype, xsb, xsc, xst, y, y_dtype, ysb, ysc, yst, xlen, B, C, T, T_out, eps, nullptr, nullptr, nullptr, s);
}

static int instnorm_launch(const void* x, int x_dtype, int64_t xsb, int64_t xsc, int64_t xst, void* y, int y_dtype, int64_t ysb, int64_t ysc, int64_t yst, const float* xlen,
                           int B, int C, int T, int T_out, float eps, float* stats_out, const float* fixed_mean, const float* fixed_var, hipStream_t s) {
	dim3 grid((C + 15) / 16, B);
#define LAUNCH(S, D) hipLaunchKernelGGL((instnorm_kernel<S, D>), grid, dim3(1024), 0, s, (const S*)x, xsb, xsc, xst, (D*)y, ysb, ysc, yst, xlen, C, T, T_out, eps, stats_out, fixed_mean, fixed_var)
	if (x_dtype == CONVASR_F32 && y_dtype == CONVASR_F32) LAUNCH(float, float);
	else if (x_dtype == CONVASR_F32 && y_dtype == CONVASR_BF16) LAUNCH(float, bf16_t);
	else if (x_dtype == CONVASR_BF16 && y_dtype == CONVASR_F32) LAUNCH(bf16_t, float);
	else if (x_dtype == CONVASR_BF16 && y_dtype == CONVASR_BF16) LAUNCH(bf16_t, bf16_t);
	else if (x_dtype == CONVASR_F32 && y_dtype == CONVASR_F16) LAUNCH(float, f16_t);
	else if (x_dtype == CONVASR_F16 && y_dtype == CONVASR_F32) LAUNCH(f16_t, float);
	else if (x_dtype == CONVASR_F16 && y_dtype == CONVASR_F16) LAUNCH(f16_t, f16_t);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "instnorm_fwd: dtype %d -> %d", x_dtype, y_dtype);
#undef LAUNCH
	CONVASR_CHECK_LAUNCH("instnorm_fwd");
	return 0;
}

// running statistics of nn.InstanceNorm1d(track_running_stats = True) (F.instance_norm): the batch mean of the per-instance means and UNBIASED
// variances, blended in with `momentum`; one thread per channel, utterances in order
__global__ void instnorm_running_kernel(const float* __restrict__ stats, int B, int C, int T, float momentum, float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c == 0 && nbt) *nbt += 1;
	if (c >= C) return;
	float m = 0.f, v = 0.f;
	const float unbias = T > 1 ? (float)T / (float)(T - 1) : 1.f;
	for (int b = 0; b < B; ++b) { m += stats[((int64_t)b * C + c) * 2]; v += stats[((int64_t)b * C + c) * 2 + 1] * unbias; }
	rmean[c] = (1.f - momentum) * rmean[c] + momentum * (m / (float)B);
	rvar[c] = (1.f - momentum) * rvar[c] + momentum * (v / (float)B);
}

extern "C" int convasr_instnorm_running_fwd(const void* x, int x_dtype, int64_t xsb, int64_t xsc, int64_t xst, void* y, int y_dtype, int64_t ysb, int64_t ysc, int64_t yst,
                                            int B, int C, int T, int T_out, float eps, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                            int training, float* stats_workspace, void* stream) {
	CONVASR_CHECK_ARG(x && y && running_mean && running_var && B > 0 && C > 0 && T > 0 && T_out >= T && (!training || stats_workspace), "instnorm_running_fwd: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	if (!training) return instnorm_launch(x, x_dtype, xsb, xsc, xst, y, y_dtype, ysb, ysc, yst, nullptr, B, C, T, T_out, eps, nullptr, running_mean, running_var, s);
	const int rc = instnorm_launch(x, x_dtype, xsb, xsc, xst, y, y_dtype, ysb, ysc, yst, nullptr, B, C, T, T_out, eps, stats_workspace, nullptr, nullptr, s);
	if (rc) return rc;
	hipLaunchKernelGGL(instnorm_running_kernel, dim3((C + 63) / 64), dim3(64), 0, s, (const float*)stats_workspace, B, C, T, momentum, running_mean, running_var, (long long*)num_batches_tracked);
	CONVASR_CHECK_LAUNCH("instnorm_running_fwd");
	return 0;
}

// ------------------------------------------------------------------------------------------------ output lengths (models.py:611-614)
__global__ void output_lengths_kernel(const float* __restrict__ xlen, int B, int T, int64_t* __restrict__ out) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < B) out[b] = (int64_t)valid_len(xlen, b, T);
}

extern "C" int convasr_output_lengths(const float* xlen, int B, int T, int64_t* out, void* stream) {
	CONVASR_CHECK_ARG(out && B > 0 && T >= 0, "output_lengths: bad arguments");
	hipLaunchKernelGGL(output_lengths_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, xlen, B, T, out);
	CONVASR_CHECK_LAUNCH("output_lengths");
	return 0;
}

// ------------------------------------------------------------------------------------------------ log-softmax over C (rows of a (rows, C) matrix)
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int C) {
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (row >= rows) return;
	const float* xp = x + row * C;
	float m = -INFINITY;
	for (int c = lane; c < C; c += 64) m = fmaxf(m, xp[c]);
	m = wave_max(m);
	float s = 0.f;
	for (int c = lane; c < C; c += 64) s += expf(xp[c] - m);
	s = wave_sum(s);
	const float lse = m + logf(s);
	for (int c = lane; c < C; c += 64) y[row * C + c] = xp[c] - lse;
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ lp, float* __restrict__ dx, int64_t rows, int C) {
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (row >= rows) return;
	float s = 0.f;
	for (int c = lane; c < C; c += 64) s += g[row * C + c];
	s = wave_sum(s);
	for (int c = lane; c < C; c += 64) dx[row * C + c] = g[row * C + c] - expf(lp[row * C + c]) * s;
}

extern "C" int convasr_log_softmax_fwd(const float* logits, float* log_probs, int64_t rows, int C, void* stream) {
	CONVASR_CHECK_ARG(logits && log_probs && rows > 0 && C > 0, "log_softmax_fwd: bad arguments");
	hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, log_probs, rows, C);
	CONVASR_CHECK_LAUNCH("log_softmax_fwd");
	return 0;
}

extern "C" int convasr_log_softmax_bwd(const float* grad_lp, const float* log_probs, float* dlogits, int64_t rows, int C, void* stream) {
	CONVASR_CHECK_ARG(grad_lp && log_probs && dlogits && rows > 0 && C > 0, "log_softmax_bwd: bad arguments");
	hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, grad_lp, log_probs, dlogits, rows, C);
	CONVASR_CHECK_LAUNCH("log_softmax_bwd");
	return 0;
}

// ------------------------------------------------------------------------------------------------ entropy / argmax / row scaling
// one workgroup per utterance walks its n x C valid log-probs as one flat coalesced array (C = 38 would leave 40 % of a
// frame-per-wave mapping idle); fixed summation order: deterministic
__global__ __launch_bounds__(1024) void entropy_kernel(const float* __restrict__ lp, const int64_t* __restrict__ olen, float* __restrict__ ent, int T, int C, float eps) {
	__shared__ float red[16];
	const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int n = olen ? (int)olen[b] : T;
	const int total = min(n, T) * C;
	const float* p = lp + (int64_t)b * T * C;
	float a0 = 0.f, a1 = 0.f;
	int i = threadIdx.x;
	for (; i + 1024 < total; i += 2048) { const float v0 = p[i], v1 = p[i + 1024]; a0 -= expf(v0) * v0; a1 -= expf(v1) * v1; }
	if (i < total) { const float v = p[i]; a0 -= expf(v) * v; }
	float acc = wave_sum(a0 + a1);
	if (lane == 0) red[w] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		float s = 0.f;
		for (int k = 0; k < 16; ++k) s += red[k];
		ent[b] = olen ? s / (eps + (float)n) : s / (float)T;
	}
}

extern "C" int convasr_entropy(const float* log_probs, const int64_t* olen, float* ent, int B, int T, int C, float eps, void* stream) {
	CONVASR_CHECK_ARG(log_probs && ent && B > 0 && T > 0 && C > 0, "entropy: bad arguments");
	hipLaunchKernelGGL(entropy_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, log_probs, olen, ent, T, C, eps);
	CONVASR_CHECK_LAUNCH("entropy");
	return 0;
}

// weighted_mean_entropy (models.py:660-682): per frame e = -sum_c p log p and weight w = 1 - p[eps_id] (x mask); per utterance
// sum(e w) / (eps + sum w).  One workgroup per utterance, one wave per frame (lane = class), 16 frames in flight; the per-wave
// partial sums are added in wave order: deterministic.
__global__ __launch_bounds__(1024) void weighted_entropy_kernel(const float* __restrict__ lp, const int64_t* __restrict__ olen, float* __restrict__ out, int T, int C, int eps_id, float eps) {
	__shared__ float red[2][16];
	const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int n = olen ? min((int)olen[b], T) : T;
	const float* p = lp + (int64_t)b * T * C;
	float num = 0.f, den = 0.f;
	for (int t = w; t < n; t += 16) {
		float e = 0.f, sil = 0.f;
		for (int c = lane; c < C; c += 64) {
			const float v = p[(int64_t)t * C + c], pr = expf(v);
			e -= pr * v;
			if (c == eps_id) sil = pr;
		}
		e = wave_sum(e);
		sil = wave_sum(sil);
		const float wt = 1.f - sil;
		num += e * wt;
		den += wt;
	}
	if (lane == 0) { red[0][w] = num; red[1][w] = den; }
	__syncthreads();
	if (threadIdx.x == 0) {
		float a = 0.f, d = 0.f;
		for (int k = 0; k < 16; ++k) { a += red[0][k]; d += red[1][k]; }
		out[b] = a / (eps + d);
	}
}

extern "C" int convasr_weighted_mean_entropy(const float* log_probs, const int64_t* olen, float* out, int B, int T, int C, int eps_id, float eps, void* stream) {
	CONVASR_CHECK_ARG(log_probs && out && B > 0 && T > 0 && C > 0 && eps_id >= 0 && eps_id < C, "weighted_mean_entropy: bad arguments");
	hipLaunchKernelGGL(weighted_entropy_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, log_probs, olen, out, T, C, eps_id, eps);
	CONVASR_CHECK_LAUNCH("weighted_mean_entropy");
	return 0;
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ lp, int64_t* __restrict__ idx, int64_t rows, int C) {
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (row >= rows) return;
	float best = -INFINITY;
	int bi = 0x7fffffff;
	for (int c = lane; c < C; c += 64) { float v = lp[row * C + c]; if (v > best || (v != v && best == best)) { best = v; bi = c; } }
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		float ov = __shfl_xor(best, o, 64);
		int oi = __shfl_xor(bi, o, 64);
		if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
	}
	if (lane == 0) idx[row] = bi == 0x7fffffff ? 0 : bi;
}

extern "C" int convasr_argmax(const float* log_probs, int64_t* idx, int64_t rows, int C, void* stream) {
	CONVASR_CHECK_ARG(log_probs && idx && rows > 0 && C > 0, "argmax: bad arguments");
	hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, log_probs, idx, rows, C);
	CONVASR_CHECK_LAUNCH("argmax");
	return 0;
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ sc, const int64_t* __restrict__ div, int64_t div_stride, float* __restrict__ out, int64_t per_b) {
	const int b = blockIdx.y;
	const float s0 = sc ? sc[b] : 1.f;  // (gscale NULL: a plain division by gdiv -- the "/ ylen[:, 0]" of models.py:323 in the forward direction)
	const float s = div ? s0 / (float)div[b * div_stride] : s0;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_b; i += (int64_t)gridDim.x * 256) out[b * per_b + i] = g[b * per_b + i] * s;
}

extern "C" int convasr_scale_rows(const float* grad, const float* gscale, const int64_t* gdiv, int64_t gdiv_stride, float* out, int B, int64_t per_b, void* stream) {
	CONVASR_CHECK_ARG(grad && (gscale || gdiv) && out && B > 0 && per_b > 0, "scale_rows: bad arguments");
	unsigned gx = (unsigned)(ceil_div64(per_b, 256) > 64 ? 64 : ceil_div64(per_b, 256));
	hipLaunchKernelGGL(scale_rows_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, grad, gscale, gdiv, gdiv_stride, out, per_b);
	CONVASR_CHECK_LAUNCH("scale_rows");
	return 0;
}

// ------------------------------------------------------------------------------------------------ loss head (train.py:754-769)
// The scalar bookkeeping of one training iteration in ONE launch instead of ~40 two-element ATen kernels: from the per-utterance
// loss vector lv[b] (= CTC NLL / target length, models.py:323), the target lengths w[b] and the per-utterance entropies:
//   out[0] = mean_b(lv[b] * w[b]) / accum          the loss that is back-propagated (train.py:755)
//   out[1] = mean_b(lv[b])                          loss_cur, the logged loss and the inf/NaN gate (train.py:755, 769)
//   out[2] = mean_b(ent[b])                         the entropy metric (train.py:756)
//   gvec[b] = ((1 / accum) / B) * w[b] (* loss scale)  d out[0] / d lv[b], in autograd's own order of operations; fp16 training seeds
//                                                   backward with the SCALED loss's gradient (apex.amp.scale_loss, train.py:770-772)
//   skipped = !isfinite(out[1])
// One workgroup, sums in a fixed order: deterministic.
__global__ __launch_bounds__(256) void loss_head_kernel(const float* __restrict__ lv, const int64_t* __restrict__ ylen, int64_t ylen_stride, const float* __restrict__ ent, int B, float accum,
                                                        float* __restrict__ out, float* __restrict__ gvec, unsigned char* __restrict__ skipped, const float* __restrict__ scaler, float metric_scale) {
	__shared__ float red[3][256];
	float a = 0.f, c = 0.f, e = 0.f;
	const float gbase = (1.f / accum) / (float)B, ls = scaler ? scaler[LS_SCALE] : 1.f;
	for (int b = threadIdx.x; b < B; b += 256) {
		const float w = (float)ylen[b * ylen_stride], l = lv[b];
		a += l * w;
		c += l;
		if (ent) e += ent[b];
		if (gvec) gvec[b] = gbase * w * ls;
	}
	red[0][threadIdx.x] = a; red[1][threadIdx.x] = c; red[2][threadIdx.x] = e;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; red[2][threadIdx.x] += red[2][threadIdx.x + o]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		const float cur = red[1][0] / (float)B;
		out[0] = red[0][0] / (float)B / accum;
		out[1] = cur * metric_scale;  // (1 / world size when a SUM all-reduce over ranks follows: the result is then the mean, train.py:759-763)
		out[2] = red[2][0] / (float)B * metric_scale;
		if (skipped) *skipped = (fabsf(cur) < INFINITY) ? 0 : 1;
	}
}

// ------------------------------------------------------------------------------------------------ per-step device state
// state[0] = the caller's dropout seed, state[1] = steps begun so far, state[2] = the key of the current step (what the dropout kernels
// XOR into their seed through `step_key`), state[3] reserved.  One thread: a step that is replayed from a captured graph advances the
// same words as an eagerly launched one, so the two draw identical masks.
__global__ void step_begin_kernel(uint64_t* __restrict__ st) {
	const uint64_t step = st[1] + 1;
	st[1] = step;
	st[2] = convasr_mix_seed(st[0] ^ convasr_mix_seed(step));
}

extern "C" int convasr_step_begin(uint64_t* state, void* stream) {
	CONVASR_CHECK_ARG(state, "step_begin: state is NULL");
	hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
	CONVASR_CHECK_LAUNCH("step_begin");
	return 0;
}

// out = a + b over n 16-bit values (n % 8 == 0), 16 bytes per lane; in place allowed (out == a or out == b)
template <typename H> __global__ __launch_bounds__(256) void add16_kernel(const H* __restrict__ a, const H* __restrict__ b, H* __restrict__ out, int64_t n8) {
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
		float x[8], y[8];
		unpack16<H>(reinterpret_cast<const uint4*>(a)[i], x);
		unpack16<H>(reinterpret_cast<const uint4*>(b)[i], y);
		reinterpret_cast<uint4*>(out)[i] = make_uint4(pack16<H>(x[0] + y[0], x[1] + y[1]), pack16<H>(x[2] + y[2], x[3] + y[3]), pack16<H>(x[4] + y[4], x[5] + y[5]), pack16<H>(x[6] + y[6], x[7] + y[7]));
	}
}

extern "C" int convasr_add16(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
	CONVASR_CHECK_ARG(a && b && out && n > 0 && (n & 7) == 0 && convasr_is_half(dtype), "add16: n must be a positive multiple of 8 and dtype CONVASR_BF16 / CONVASR_F16");
	int64_t blocks = ceil_div64(n >> 3, 256);
	if (blocks > 8192) blocks = 8192;
	if (dtype == CONVASR_F16) hipLaunchKernelGGL((add16_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)a, (const f16_t*)b, (f16_t*)out, n >> 3);
	else hipLaunchKernelGGL((add16_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n >> 3);
	CONVASR_CHECK_LAUNCH("add16");
	return 0;
}

// dst[i] = (D)(src[i] * scale) over n elements (n % 8 == 0), fp32 <-> 16-bit: the two ends of a 16-bit gradient exchange
template <typename S, typename D> __global__ __launch_bounds__(256) void cast_scale_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n8, float scale) {
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
		float v[8];
		load8<S>(src + (i << 3), v);
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] *= scale;
		store8<D>(dst + (i << 3), v);
	}
}

extern "C" int convasr_cast_scale(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale, void* stream) {
	CONVASR_CHECK_ARG(src && dst && n > 0 && (n & 7) == 0 && ((src_dtype == CONVASR_F32 && convasr_is_half(dst_dtype)) || (convasr_is_half(src_dtype) && dst_dtype == CONVASR_F32)), "cast_scale: n %% 8 == 0, fp32 -> bf16 / fp16 or back");
	int64_t blocks = ceil_div64(n >> 3, 256);
	if (blocks > 8192) blocks = 8192;
	const dim3 g((unsigned)blocks), b(256);
	hipStream_t s = (hipStream_t)stream;
	if (src_dtype == CONVASR_F32) CONVASR_DISPATCH_HALF(dst_dtype, H, hipLaunchKernelGGL((cast_scale_kernel<float, H>), g, b, 0, s, (const float*)src, (H*)dst, n >> 3, scale));
	else CONVASR_DISPATCH_HALF(src_dtype, H, hipLaunchKernelGGL((cast_scale_kernel<H, float>), g, b, 0, s, (const H*)src, (float*)dst, n >> 3, scale));
	CONVASR_CHECK_LAUNCH("cast_scale");
	return 0;
}

// A device-to-device copy as a KERNEL, never hipMemcpyAsync: a stream-ordered launch like every other in the step, and a kernel node (not a
// memcpy node, which the runtime may hand to a DMA engine) when the step is captured into a HIP graph -- see convasr_copy below.
__global__ __launch_bounds__(256) void copy_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int64_t n16, int64_t nbytes, int aligned) {
	const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
	if (aligned) {
		for (int64_t i = i0; i < n16; i += step) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
		for (int64_t i = n16 * 16 + i0; i < nbytes; i += step) dst[i] = src[i];
	} else {
		for (int64_t i = i0; i < nbytes; i += step) dst[i] = src[i];
	}
}

extern "C" int convasr_copy(const void* src, void* dst, int64_t nbytes, void* stream) {
	CONVASR_CHECK_ARG(src && dst && nbytes >= 0, "copy: bad arguments");
	if (nbytes == 0) return 0;
	const int aligned = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
	const int64_t n16 = aligned ? nbytes >> 4 : 0;
	int64_t blocks = ceil_div64(aligned ? (n16 > 0 ? n16 : 1) : nbytes, 256);
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, (unsigned char*)dst, n16, nbytes, aligned);
	CONVASR_CHECK_LAUNCH("copy");
	return 0;
}

extern "C" int convasr_loss_head(const float* loss_vec, const int64_t* ylen, int64_t ylen_stride, const float* entropy, int B, float accumulate_iterations, float* out3,
                                 float* grad_loss_vec, unsigned char* skipped, const float* loss_scaler, float metric_scale, void* stream) {
	CONVASR_CHECK_ARG(loss_vec && ylen && out3 && B > 0 && accumulate_iterations > 0.f, "loss_head: bad arguments");
	hipLaunchKernelGGL(loss_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_vec, ylen, ylen_stride, entropy, B, accumulate_iterations, out3, grad_loss_vec, skipped, loss_scaler, metric_scale);
	CONVASR_CHECK_LAUNCH("loss_head");
	return 0;
}

// ------------------------------------------------------------------------------------------------ gradient norm + SGD
#define SUMSQ_BLOCKS 2048
// two launches, no atomics: SUMSQ_BLOCKS partial sums, then one workgroup adds them in index order (the clip coefficient, hence every
// parameter, is bit-identical from run to run)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
	__shared__ float red[4];
	float acc = 0.f;
	const int64_t n4 = n >> 2;
	const float4* g4 = reinterpret_cast<const float4*>(g);
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
		float4 v = g4[i];
		acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[n4 * 4 + threadIdx.x]; acc += v * v; }
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, int blocks, double* __restrict__ out, float* __restrict__ norm_out, float norm_scale, const float* __restrict__ scaler) {
	__shared__ double red[256];
	double a = 0;
	for (int i = threadIdx.x; i < blocks; i += 256) a += part[i];
	red[threadIdx.x] = a;
	__syncthreads();
	if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < 256; ++i) s += red[i]; *out = s; if (norm_out) *norm_out = (float)(sqrt(s) * (double)norm_scale) * (scaler ? 1.f / scaler[LS_SCALE] : 1.f); }
}

extern "C" int64_t convasr_sumsq_workspace_bytes(void) { return SUMSQ_BLOCKS * (int64_t)sizeof(double); }

extern "C" int convasr_sumsq(const float* g, int64_t n, double* sumsq, void* workspace, float* norm_out, float norm_scale, const float* loss_scaler, void* stream) {
	CONVASR_CHECK_ARG(g && sumsq && workspace && n > 0, "sumsq: bad arguments");
	CONVASR_CHECK_ARG(((uintptr_t)g & 15) == 0, "sumsq: g must be 16-byte aligned");
	int64_t blocks = ceil_div64(n, 1024);
	if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
	hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, (double*)workspace);
	hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, (int)blocks, sumsq, norm_out, norm_scale, loss_scaler);
	CONVASR_CHECK_LAUNCH("sumsq");
	return 0;
}

template <typename H> __global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float* __restrict__ gout,
                                                       int64_t n, const double* __restrict__ sumsq, float max_norm, float lr, float mom, float wd, int nesterov, int first,
                                                       const float* __restrict__ loss_gate, float grad_scale, H* __restrict__ p16, const float* __restrict__ scaler_in, float* __restrict__ scaler_out, const float* __restrict__ lr_dev) {
	if (lr_dev) lr = *lr_dev;  // the learning rate from device memory (a captured step graph follows the host's scheduler through it)
	const bool gated = loss_gate && !(fabsf(*loss_gate) < INFINITY);  // inf or NaN loss: the step is skipped (train.py:769-772)
	const LossScale ls = loss_scale_read(scaler_in, (scaler_in && sumsq) ? *sumsq : 0.0);
	if (scaler_in && blockIdx.x == 0 && threadIdx.x == 0) loss_scale_advance(scaler_in, scaler_out, ls.overflow, gated);
	if (gated || ls.overflow) return;  // (overflow: a non-finite gradient under the current loss scale -- apex skips the step and halves the scale)
	grad_scale *= ls.inv;
	float clip = 1.f;
	if (sumsq && max_norm > 0.f) {  // (max_norm <= 0: no clipping; sumsq may still be there for the loss scaler's overflow check)
		float total = (float)sqrt(*sumsq) * grad_scale;  // norm of the scaled gradient (grad_scale = 1 / world size on summed gradients)
		float c = max_norm / (total + 1e-6f);
		clip = c < 1.f ? c : 1.f;
	}
	clip *= grad_scale;
	auto update = [&](float gi, float pv, float& bv, float& gout_v) {
		const float gc = gi * clip;
		gout_v = gc;
		float d = gc + wd * pv;
		if (mom != 0.f) {
			bv = first ? d : mom * bv + d;
			d = nesterov ? d + mom * bv : bv;
		}
		return pv - lr * d;
	};
	// 16 bytes per lane and array; the tail (n % 4) is handled element-wise by the first lanes of block 0
	const int64_t n4 = n >> 2;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
		const float4 g4 = reinterpret_cast<const float4*>(g)[i], p4 = reinterpret_cast<const float4*>(p)[i];
		float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), o4;
		if (mom != 0.f && !first) b4 = reinterpret_cast<const float4*>(buf)[i];
		float4 r;
		r.x = update(g4.x, p4.x, b4.x, o4.x); r.y = update(g4.y, p4.y, b4.y, o4.y); r.z = update(g4.z, p4.z, b4.z, o4.z); r.w = update(g4.w, p4.w, b4.w, o4.w);
		reinterpret_cast<float4*>(p)[i] = r;
		if (mom != 0.f) reinterpret_cast<float4*>(buf)[i] = b4;
		if (gout) reinterpret_cast<float4*>(gout)[i] = o4;
		if (p16) reinterpret_cast<uint2*>(p16)[i] = make_uint2(pack16<H>(r.x, r.y), pack16<H>(r.z, r.w));
	}
	if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		float bv = (mom != 0.f && !first) ? buf[i] : 0.f, go;
		const float r = update(g[i], p[i], bv, go);
		p[i] = r;
		if (mom != 0.f) buf[i] = bv;
		if (gout) gout[i] = go;
		if (p16) Elem<H>::store(p16 + i, r);
	}
}

extern "C" int convasr_sgd_step(float* p, const float* g, float* buf, float* grad_out, int64_t n, const double* sumsq, float max_norm, float lr,
                                float momentum, float weight_decay, int nesterov, int first, const float* loss_gate, float grad_scale, void* p16, int p16_dtype,
                                const float* scaler_in, float* scaler_out, const float* lr_dev, void* stream) {
	CONVASR_CHECK_ARG(p && g && n > 0 && (momentum == 0.f || buf), "sgd_step: bad arguments");
	CONVASR_CHECK_ARG(!p16 || convasr_is_half(p16_dtype), "sgd_step: the mirror's dtype must be CONVASR_BF16 or CONVASR_F16");
	CONVASR_CHECK_ARG((scaler_in == nullptr) == (scaler_out == nullptr) && (!scaler_in || (scaler_in != scaler_out && sumsq)), "sgd_step: the loss scaler needs distinct in / out states and the gradient's sum of squares (its overflow check)");
	int64_t blocks = ceil_div64(n, 256);
	if (blocks > 4096) blocks = 4096;
	CONVASR_DISPATCH_HALF(p16_dtype, H, hipLaunchKernelGGL((sgd_step_kernel<H>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, grad_out, n, sumsq, max_norm, lr, momentum, weight_decay, nesterov, first, loss_gate, grad_scale, (H*)p16, scaler_in, scaler_out, lr_dev));
	CONVASR_CHECK_LAUNCH("sgd_step");
	return 0;
}

// torch.optim.AdamW (train.py:663-668) over the flat arena, with clip_grad_norm_ (train.py:777), the device-side loss gate, the
// gradient-mean scale, the 16-bit mirror and the dynamic loss scaler folded in exactly as in sgd_step_kernel.  The number of steps
// APPLIED so far lives on the device (step_in[0] -> step_out[0], two buffers the host swaps): a gated or overflowed launch does not
// advance the bias corrections, like a reference run that never reached optimizer.step().
template <typename H> __global__ __launch_bounds__(256) void adamw_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         int64_t n, const double* __restrict__ sumsq, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                                                         const float* __restrict__ step_in, float* __restrict__ step_out, const float* __restrict__ loss_gate, float grad_scale,
                                                         H* __restrict__ p16, const float* __restrict__ scaler_in, float* __restrict__ scaler_out, const float* __restrict__ lr_dev) {
	if (lr_dev) lr = *lr_dev;
	const bool gated = loss_gate && !(fabsf(*loss_gate) < INFINITY);
	const LossScale ls = loss_scale_read(scaler_in, (scaler_in && sumsq) ? *sumsq : 0.0);
	const float t0 = *step_in;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		if (scaler_in) loss_scale_advance(scaler_in, scaler_out, ls.overflow, gated);
		*step_out = (gated || ls.overflow) ? t0 : t0 + 1.f;
	}
	if (gated || ls.overflow) return;
	grad_scale *= ls.inv;
	float clip = 1.f;
	if (sumsq && max_norm > 0.f) {
		const float total = (float)sqrt(*sumsq) * grad_scale;
		const float c = max_norm / (total + 1e-6f);
		clip = c < 1.f ? c : 1.f;
	}
	clip *= grad_scale;
	// torch/optim/adamw.py (single-tensor path): p *= 1 - lr wd; m, v EMAs; p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
	const double t = (double)t0 + 1.0;
	const float bc1 = (float)(1.0 - pow((double)beta1, t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
	const float step_size = lr / bc1, decay = 1.f - lr * wd, omb1 = 1.f - beta1, omb2 = 1.f - beta2;
	auto update = [&](float gi, float pv, float& mv, float& vv) {
		const float gc = gi * clip;
		mv = beta1 * mv + omb1 * gc;
		vv = beta2 * vv + omb2 * gc * gc;
		return pv * decay - step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));
	};
	const int64_t n4 = n >> 2;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
		const float4 g4 = reinterpret_cast<const float4*>(g)[i], p4 = reinterpret_cast<const float4*>(p)[i];
		float4 m4 = reinterpret_cast<const float4*>(m)[i], v4 = reinterpret_cast<const float4*>(v)[i], r;
		r.x = update(g4.x, p4.x, m4.x, v4.x); r.y = update(g4.y, p4.y, m4.y, v4.y); r.z = update(g4.z, p4.z, m4.z, v4.z); r.w = update(g4.w, p4.w, m4.w, v4.w);
		reinterpret_cast<float4*>(p)[i] = r;
		reinterpret_cast<float4*>(m)[i] = m4;
		reinterpret_cast<float4*>(v)[i] = v4;
		if (p16) reinterpret_cast<uint2*>(p16)[i] = make_uint2(pack16<H>(r.x, r.y), pack16<H>(r.z, r.w));
	}
	if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		float mv = m[i], vv = v[i];
		const float r = update(g[i], p[i], mv, vv);
		p[i] = r; m[i] = mv; v[i] = vv;
		if (p16) Elem<H>::store(p16 + i, r);
	}
}

extern "C" int convasr_adamw_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, const double* sumsq, float max_norm, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, const float* step_in, float* step_out, const float* loss_gate,
                                  float grad_scale, void* p16, int p16_dtype, const float* scaler_in, float* scaler_out, const float* lr_dev, void* stream) {
	CONVASR_CHECK_ARG(p && g && exp_avg && exp_avg_sq && n > 0 && step_in && step_out && step_in != step_out, "adamw_step: bad arguments");
	CONVASR_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adamw_step: betas in [0, 1), eps >= 0");
	CONVASR_CHECK_ARG(!p16 || convasr_is_half(p16_dtype), "adamw_step: the mirror's dtype must be CONVASR_BF16 or CONVASR_F16");
	CONVASR_CHECK_ARG((scaler_in == nullptr) == (scaler_out == nullptr) && (!scaler_in || (scaler_in != scaler_out && sumsq)), "adamw_step: the loss scaler needs distinct in / out states and the gradient's sum of squares (its overflow check)");
	int64_t blocks = ceil_div64(n, 256);
	if (blocks > 4096) blocks = 4096;
	CONVASR_DISPATCH_HALF(p16_dtype, H, hipLaunchKernelGGL((adamw_step_kernel<H>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, step_in, step_out, loss_gate, grad_scale, (H*)p16, scaler_in, scaler_out, lr_dev));
	CONVASR_CHECK_LAUNCH("adamw_step");
	return 0;
}
