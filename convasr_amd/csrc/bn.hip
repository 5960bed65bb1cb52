// BatchNorm1d (training / eval) + residual sum + activation + dropout + temporal mask, forward and backward, on
// channels-last (B, T, C) activations.  Reference: nn.BatchNorm1d at models.py:111-114, ConvBn1d.forward 127-139,
// ResidualActivation.forward 357-371, relu_dropout 436-443.  All HBM-bound: 16 bytes per lane, lanes walk the channel axis.
#include "common.h"

// ------------------------------------------------------------------------------------------------ statistics -> scale / shift
// stats: [rows][2][C] fp64 partial sums, one row per m-tile of the conv launch that produced them (rows = 1: plain totals).
// Block = 16 channels x 64 row-lanes (one 128-byte segment of a partial row per 16 lanes); a lane adds rows w, w + 64, ... in order and
// the 64 lanes are combined in a fixed two-stage order: the result does not depend on which workgroup of the conv finished first (the
// conv epilogue uses no atomics).  (Round 2 used 64 channels x 16 row-lanes: 4-16 workgroups per launch, each lane walking 12+ rows one
// dependent HBM round trip after the other -- 8 us per launch, 36 launches per training step; this shape takes 3-4 loads per lane on
// C / 16 workgroups.)
#define FIN_CH 16
#define FIN_LANES 64
// Both planes of the 16 channels of this block: returns true on the one thread per channel (row-lane 0) that holds the totals.
template <typename P> __device__ __forceinline__ bool finalize_rows(const P* __restrict__ part, int rows, int C, double (&red)[2][FIN_LANES][FIN_CH], double& s1, double& s2) {
	const int cl = threadIdx.x & (FIN_CH - 1), c = blockIdx.x * FIN_CH + cl, w = threadIdx.x >> 4;
	double a = 0, q2 = 0;
	if (c < C)
		for (int r = w; r < rows; r += FIN_LANES) { a += (double)part[((int64_t)r * 2) * C + c]; q2 += (double)part[((int64_t)r * 2 + 1) * C + c]; }
	red[0][w][cl] = a;
	red[1][w][cl] = q2;
	__syncthreads();
	if (w < 8) {
		a = 0; q2 = 0;
#pragma unroll
		for (int j = 0; j < FIN_LANES / 8; ++j) { a += red[0][w + 8 * j][cl]; q2 += red[1][w + 8 * j][cl]; }
	}
	__syncthreads();
	if (w < 8) { red[0][w][cl] = a; red[1][w][cl] = q2; }
	__syncthreads();
	if (w != 0 || c >= C) return false;
	s1 = 0; s2 = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) { s1 += red[0][i][cl]; s2 += red[1][i][cl]; }
	return true;
}

__device__ __forceinline__ void bn_finalize_body(const double* __restrict__ stats, int rows, double n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps, float* __restrict__ mean,
                                                 float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int C, long long* __restrict__ nbt) {
	__shared__ double red[2][FIN_LANES][FIN_CH];
	const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
	if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
	double s1, s2;
	if (!finalize_rows(stats, rows, C, red, s1, s2)) return;
	const double m = s1 / n;
	double var = s2 / n - m * m;
	if (var < 0) var = 0;
	const float mf = (float)m, vf = (float)var;
	const float is = 1.0f / sqrtf(vf + eps);
	const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
	mean[c] = mf;
	invstd[c] = is;
	scale[c] = g * is;
	shift[c] = bt - mf * g * is;
	if (rmean) {
		const float unbiased = n > 1 ? (float)(var * n / (n - 1)) : vf;
		rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
		rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
	}
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ stats, int rows, double n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int C, long long* __restrict__ nbt) {
	bn_finalize_body(stats, rows, n, gamma, beta, rmean, rvar, momentum, eps, mean, invstd, scale, shift, C, nbt);
}

// the same for up to BN_MAX_GROUP batch norms of one channel count in one launch (blockIdx.y = which): the residual branches of a dense block
#define BN_MAX_GROUP 13
struct BnFinGroup {
	const double* stats[BN_MAX_GROUP]; const float* gamma[BN_MAX_GROUP]; const float* beta[BN_MAX_GROUP]; float* rmean[BN_MAX_GROUP]; float* rvar[BN_MAX_GROUP];
	float* out[BN_MAX_GROUP];  // 4 C floats each: mean, invstd, scale, shift
	long long* nbt[BN_MAX_GROUP];
	float momentum[BN_MAX_GROUP], eps[BN_MAX_GROUP];
};
__global__ __launch_bounds__(1024) void bn_finalize_grouped_kernel(BnFinGroup g, int rows, double n, int C) {
	const int q = blockIdx.y;
	float* const o = g.out[q];
	bn_finalize_body(g.stats[q], rows, n, g.gamma[q], g.beta[q], g.rmean[q], g.rvar[q], g.momentum[q], g.eps[q], o, o + C, o + 2 * C, o + 3 * C, C, g.nbt[q]);
}

extern "C" int convasr_bn_finalize_grouped(int count, const double* const* stats, int stats_rows, int64_t n, const float* const* gamma, const float* const* beta,
                                           float* const* running_mean, float* const* running_var, const float* momentum, const float* eps, float* const* out,
                                           int64_t* const* num_batches_tracked, int C, void* stream) {
	CONVASR_CHECK_ARG(count > 0 && count <= BN_MAX_GROUP && stats && out && momentum && eps && stats_rows > 0 && n > 0 && C > 0, "bn_finalize_grouped: bad arguments (at most %d batch norms)", BN_MAX_GROUP);
	BnFinGroup g = {};
	for (int i = 0; i < count; ++i) {
		CONVASR_CHECK_ARG(stats[i] && out[i], "bn_finalize_grouped: batch norm %d has a NULL buffer", i);
		g.stats[i] = stats[i]; g.gamma[i] = gamma ? gamma[i] : nullptr; g.beta[i] = beta ? beta[i] : nullptr; g.rmean[i] = running_mean ? running_mean[i] : nullptr;
		g.rvar[i] = running_var ? running_var[i] : nullptr; g.out[i] = out[i]; g.nbt[i] = num_batches_tracked ? (long long*)num_batches_tracked[i] : nullptr;
		g.momentum[i] = momentum[i]; g.eps[i] = eps[i];
	}
	hipLaunchKernelGGL(bn_finalize_grouped_kernel, dim3((C + FIN_CH - 1) / FIN_CH, count), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, g, stats_rows, (double)n, C);
	CONVASR_CHECK_LAUNCH("bn_finalize_grouped");
	return 0;
}

extern "C" int convasr_bn_finalize(const double* stats, int stats_rows, int64_t n, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int C, int64_t* num_batches_tracked, void* stream) {
	CONVASR_CHECK_ARG(stats && stats_rows > 0 && mean && invstd && scale && shift && n > 0 && C > 0, "bn_finalize: bad arguments");
	hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, stats, stats_rows, (double)n, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, C, (long long*)num_batches_tracked);
	CONVASR_CHECK_LAUNCH("bn_finalize");
	return 0;
}

// totals of a [rows][width] fp64 partial buffer, same fixed order (tests / tools that want the plain sums)
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const double* __restrict__ part, int rows, int width, double* __restrict__ out) {
	__shared__ double red[16][64];
	const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, w = threadIdx.x >> 6;
	double a = 0;
	if (c < width)
		for (int r = w; r < rows; r += 16) a += part[(int64_t)r * width + c];
	red[w][cl] = a;
	__syncthreads();
	if (w == 0 && c < width) { double s = 0; for (int i = 0; i < 16; ++i) s += red[i][cl]; out[c] = s; }
}

extern "C" int convasr_reduce_rows(const double* part, int rows, int width, double* out, void* stream) {
	CONVASR_CHECK_ARG(part && out && rows > 0 && width > 0, "reduce_rows: bad arguments");
	hipLaunchKernelGGL(reduce_rows_kernel, dim3((width + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, rows, width, out);
	CONVASR_CHECK_LAUNCH("reduce_rows");
	return 0;
}

__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rmean, const float* __restrict__ rvar,
                               float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	const float is = 1.0f / sqrtf(rvar[c] + eps);
	const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
	scale[c] = g * is;
	shift[c] = bt - rmean[c] * g * is;
}

extern "C" int convasr_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                           float* scale, float* shift, int C, void* stream) {
	CONVASR_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_eval_scale_shift: bad arguments");
	hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale, shift, C);
	CONVASR_CHECK_LAUNCH("bn_eval_scale_shift");
	return 0;
}

// ------------------------------------------------------------------------------------------------ elementwise forward
#define MAX_RES 12
struct ResArgs {
	const void* res[MAX_RES];
	const float* rscale[MAX_RES];
	const float* rshift[MAX_RES];
	const float* rmean[MAX_RES];
	const float* rinvstd[MAX_RES];
	double* rsums[MAX_RES];
	int n;
};

struct BnActParams {
	const void* y;
	const void* dz;
	void* out;
	const float* scale;
	const float* shift;
	const float* mean;
	const float* invstd;
	const float* xlen;
	double* sums;
	uint8_t* gate_out;       // forward, optional: bit (element index & 7) of byte (element index >> 3) = the element's gradient gate (below)
	const uint8_t* gate_in;  // backward, optional: use the stored gate instead of re-deriving act' / dropout / mask
	int act;
	float lo, hi, p_drop;
	unsigned drop_thr;  // an element is dropped when its 16 random bits are < drop_thr (= round(p * 65536))
	float keep_scale;   // 65536 / (65536 - drop_thr): E[keep] = 1 exactly
	uint64_t seed, offset;
	const uint64_t* step_key;  // optional device word XORed into the seed: the per-step dropout key (convasr_step_begin), so that a captured step graph draws new masks at every replay
	int B, T, C;
	int cgroups, rlanes, rows_per_block;  // thread tid -> channel group tid % cgroups (8 channels), row-lane tid / cgroups
};

static void set_dropout(BnActParams& p, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key) {
	p.p_drop = dropout_p; p.seed = convasr_mix_seed(seed); p.offset = offset; p.step_key = step_key;
	p.drop_thr = (unsigned)lrintf(dropout_p * 65536.f);
	if (p.drop_thr > 65535u) p.drop_thr = 65535u;
	p.keep_scale = 65536.f / (float)(65536u - p.drop_thr);
}

// All three streaming kernels walk the (B*T, C) matrix the same way: a block owns a contiguous range of rows, a thread owns 8
// consecutive channels (its per-channel constants stay in registers) and every rlanes-th row of the range; (b, t) and the
// utterance's valid length are carried along instead of divided out per row.  Launch: blockDim = cgroups * rlanes <= 256.
static void row_walk_config(BnActParams& p, int rows_per_thread, dim3& grid, dim3& block) {
	const int c8 = p.C >> 3;
	p.cgroups = c8 < 256 ? c8 : 256;
	p.rlanes = 256 / p.cgroups;
	p.rows_per_block = p.rlanes * rows_per_thread;
	const int64_t rows = (int64_t)p.B * p.T;
	int gy = (c8 + p.cgroups - 1) / p.cgroups;
	if (gy > 4) gy = 4;
	grid = dim3((unsigned)ceil_div64(rows, p.rows_per_block), gy);
	block = dim3(p.cgroups * p.rlanes);
}

struct RowWalk {
	int row, r1, b, t, nv;
	int64_t idx;
	__device__ __forceinline__ RowWalk(const BnActParams& p, int rl, int c) {
		const int rows = p.B * p.T;
		row = blockIdx.x * p.rows_per_block + rl;
		r1 = min(rows, (int)(blockIdx.x + 1) * p.rows_per_block);
		b = row / p.T; t = row - b * p.T;
		nv = b < p.B ? valid_len(p.xlen, b, p.T) : 0;
		idx = (int64_t)row * p.C + c;
	}
	__device__ __forceinline__ bool live() const { return row < r1; }
	__device__ __forceinline__ bool masked() const { return t >= nv; }
	__device__ __forceinline__ void next(const BnActParams& p) {
		row += p.rlanes; idx += (int64_t)p.rlanes * p.C; t += p.rlanes;
		while (t >= p.T) { t -= p.T; ++b; nv = b < p.B ? valid_len(p.xlen, b, p.T) : 0; }
	}
};

// Storage-type dispatch of the streaming kernels below: runs the statement with T = float / bf16_t / f16_t
#define BN_DISPATCH(name, dtype, T, ...) \
	if ((dtype) == CONVASR_F32) { typedef float T; __VA_ARGS__; } \
	else if ((dtype) == CONVASR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
	else if ((dtype) == CONVASR_F16) { typedef f16_t T; __VA_ARGS__; } \
	else return convasr_fail(CONVASR_EUNSUPPORTED, name ": dtype %d", (dtype))

// 8 consecutive elements as loaded (16 B of bf16 / fp16, 32 B of f32): kept raw so that the loads of two rows can be issued back to
// back before either is unpacked -- the kernels below are latency-bound on bytes in flight, not on arithmetic
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { uint4 v; };
template <> struct Raw8<f16_t> { uint4 v; };
template <> struct Raw8<float> { float4 a, b; };
__device__ __forceinline__ Raw8<bf16_t> raw_load8(const bf16_t* p) { Raw8<bf16_t> r; r.v = *reinterpret_cast<const uint4*>(p); return r; }
__device__ __forceinline__ Raw8<f16_t> raw_load8(const f16_t* p) { Raw8<f16_t> r; r.v = *reinterpret_cast<const uint4*>(p); return r; }
__device__ __forceinline__ Raw8<float> raw_load8(const float* p) { Raw8<float> r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r; }
__device__ __forceinline__ void unpack8(const Raw8<bf16_t>& r, float (&v)[8]) { unpack16<bf16_t>(r.v, v); }
__device__ __forceinline__ void unpack8(const Raw8<f16_t>& r, float (&v)[8]) { unpack16<f16_t>(r.v, v); }
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&v)[8]) {
	v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}

// pre-activation value of 8 consecutive channels at row (b, t): y * scale + shift + sum_r (res_r * rscale_r + rshift_r)
template <typename T> __device__ __forceinline__ void pre_act8(const BnActParams& p, const ResArgs& ra, int64_t idx, int c, const float (&sc)[8], const float (&sh)[8], const Raw8<T>& yraw, float (&yv)[8], float (&pre)[8]) {
	unpack8(yraw, yv);
#pragma unroll
	for (int i = 0; i < 8; ++i) pre[i] = p.scale ? fmaf(yv[i], sc[i], sh[i]) : yv[i];
	for (int r = 0; r < ra.n; ++r) {
		float rv[8];
		load8<T>(reinterpret_cast<const T*>(ra.res[r]) + idx, rv);
		if (ra.rscale[r]) {
			float rsc[8], rsh[8];
			load8<float>(ra.rscale[r] + c, rsc);
			load8<float>(ra.rshift[r] + c, rsh);
#pragma unroll
			for (int i = 0; i < 8; ++i) pre[i] += fmaf(rv[i], rsc[i], rsh[i]);
		} else {
#pragma unroll
			for (int i = 0; i < 8; ++i) pre[i] += rv[i];
		}
	}
}

// One row per trip (the residual variant of the reduce kernel: 183 VGPRs with two rows in flight, half the occupancy).
template <typename L, typename U> __device__ __forceinline__ void walk_rows1(const BnActParams& p, int rl, int c, L load, U use) {
	for (RowWalk w(p, rl, c); w.live(); w.next(p)) use(w, load(w));
}

// Two rows per trip: both rows' loads are issued (through `load`) before the first is consumed (by `use`).
template <typename L, typename U> __device__ __forceinline__ void walk_rows2(const BnActParams& p, int rl, int c, L load, U use) {
	for (RowWalk w(p, rl, c); w.live();) {
		RowWalk w2 = w;
		w2.next(p);
		const bool two = w2.live();
		auto r1 = load(w);
		auto r2 = two ? load(w2) : r1;
		use(w, r1);
		if (two) { use(w2, r2); w2.next(p); }
		w = w2;
	}
}

#ifdef CONVASR_BN_ROWS4  // measurement hook (python -m convasr_amd.build --variant rows4 -DCONVASR_BN_ROWS4=1): four rows in flight in the forward / apply passes too
#define WALK_ROWS_FWD walk_rows4
#else
#define WALK_ROWS_FWD walk_rows2
#endif
// Four rows per trip: all four rows' loads are issued before the first is consumed (the gated reduce pass: 33 bytes per row and lane,
// nothing but bytes in flight limits it).
template <typename L, typename U> __device__ __forceinline__ void walk_rows4(const BnActParams& p, int rl, int c, L load, U use) {
	for (RowWalk w(p, rl, c); w.live();) {
		RowWalk w1 = w; w1.next(p);
		RowWalk w2 = w1; w2.next(p);
		RowWalk w3 = w2; w3.next(p);
		const bool l1 = w1.live(), l2 = w2.live(), l3 = w3.live();
		auto r0 = load(w);
		auto r1 = l1 ? load(w1) : r0;
		auto r2 = l2 ? load(w2) : r0;
		auto r3 = l3 ? load(w3) : r0;
		use(w, r0);
		if (l1) use(w1, r1);
		if (l2) use(w2, r2);
		if (l3) use(w3, r3);
		w = w3;
		w.next(p);
	}
}

__device__ __forceinline__ void dropout_keep8(const BnActParams& p, int64_t idx, float (&keep)[8]) { dropout_mask8(p.seed, p.offset, p.drop_thr, p.keep_scale, idx, keep); }
// once per kernel (p is the kernel's own copy of the parameter block): fold the device-resident step key into the seed -- a uniform scalar load
__device__ __forceinline__ void fold_step_key(BnActParams& p) { if (p.drop_thr && p.step_key) p.seed ^= *p.step_key; }

// MODE bit 0: residual inputs present, bit 1: dropout on, bit 2: gate bits wanted, bit 3: BN scale / shift present, bit 4: the
// activation may be leaky-relu (else a clamp: one v_med3_f32 per element) -- compile-time, so
// that the training step's launches (no residuals, scale / shift, dropout, gates) run straight-line code without the other cases'
// branches, register copies and live ranges (the kernel is VALU-bound, not HBM-bound: 16 instantiations instead of one).
// SPLIT (fp32 passes of a split-operand network, csrc/split3.hip): 0 = the result is stored as T; 1 / 2 = it is stored as its three bf16 / fp16
// planes per frame, [row][3][C] (order 0: a conv input) -- the pass that would otherwise follow (read 4 + write 6 bytes per element) is gone.
template <int SPLIT> struct PlaneType { typedef bf16_t H; };
template <> struct PlaneType<2> { typedef f16_t H; };
template <> struct PlaneType<4> { typedef f16_t H; };  // (3 / 4, the backward apply only: the hi plane alone, dense [row][C] bf16 / fp16 -- the one-product backward of a split-operand forward)
template <typename T, int MODE, int SPLIT = 0> __global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnActParams p, ResArgs ra_) {
	constexpr bool RES = MODE & 1, DROP = (MODE & 2) != 0, GATE = (MODE & 4) != 0, AFFINE = (MODE & 8) != 0, LEAKY = (MODE & 16) != 0;
	ResArgs ra = ra_;
	if (!RES) ra.n = 0;
	if (DROP) fold_step_key(p);
	if (!AFFINE) p.scale = nullptr;
	else __builtin_assume(p.scale != nullptr);
	const ActConst ac = act_const(p.act, p.lo, p.hi);  // the activation kind folded into constants once: no per-element switch
	const int c8 = p.C >> 3;
	const int cg = threadIdx.x % p.cgroups, rl = threadIdx.x / p.cgroups;
	for (int cbase = blockIdx.y * p.cgroups; cbase < c8; cbase += gridDim.y * p.cgroups) {
		if (cbase + cg >= c8) continue;
		const int c = (cbase + cg) << 3;
		float sc[8], sh[8];
		if (AFFINE) { load8<float>(p.scale + c, sc); load8<float>(p.shift + c, sh); }
		const T* const py = reinterpret_cast<const T*>(p.y);
		WALK_ROWS_FWD(p, rl, c,
			[&](const RowWalk& w) { return raw_load8(py + w.idx); },  // unconditional (a masked row is still inside the tensor): no branch between the two rows' loads
			[&](const RowWalk& w, const Raw8<T>& yraw) {
				float out[8];
				unsigned gate = 0;  // bit k: the gradient passes element k (inside the activation's linear range, kept by dropout, frame not masked)
				if (w.masked()) {
#pragma unroll
					for (int k = 0; k < 8; ++k) out[k] = 0.f;
				} else {
					float yv[8], pre[8];
					pre_act8<T>(p, ra, w.idx, c, sc, sh, yraw, yv, pre);
#pragma unroll
					for (int k = 0; k < 8; ++k) out[k] = LEAKY ? apply_act(pre[k], ac) : apply_clamp(pre[k], ac);
					if (DROP) {
						float keep[8];
						dropout_keep8(p, w.idx, keep);
#pragma unroll
						for (int k = 0; k < 8; ++k) out[k] *= keep[k];
						if (GATE) {
#pragma unroll
							for (int k = 0; k < 8; ++k) gate |= (act_grad(pre[k], ac) != 0.f && keep[k] != 0.f) ? (1u << k) : 0u;
						}
					} else if (GATE) {
#pragma unroll
						for (int k = 0; k < 8; ++k) gate |= act_grad(pre[k], ac) != 0.f ? (1u << k) : 0u;
					}
				}
				if constexpr (SPLIT == 0) store8<T>(reinterpret_cast<T*>(p.out) + w.idx, out);
				else split3_store8<typename PlaneType<SPLIT>::H>(reinterpret_cast<typename PlaneType<SPLIT>::H*>(p.out) + 3 * w.idx - 2 * c, p.C, 0, out);  // (row * 3 C + c)
				if (GATE) p.gate_out[w.idx >> 3] = (uint8_t)gate;
			});
	}
}

static int fill_res(ResArgs& ra, int n_res, const void* const* res, const float* const* rscale, const float* const* rshift, const float* const* rmean,
                    const float* const* rinvstd, double* const* rsums) {
	if (n_res < 0 || n_res > MAX_RES) return convasr_fail(CONVASR_EINVAL, "bn_act: n_res %d not in [0, %d]", n_res, MAX_RES);
	ra.n = n_res;
	for (int r = 0; r < MAX_RES; ++r) {
		ra.res[r] = r < n_res ? res[r] : nullptr;
		ra.rscale[r] = (r < n_res && rscale) ? rscale[r] : nullptr;
		ra.rshift[r] = (r < n_res && rshift) ? rshift[r] : nullptr;
		ra.rmean[r] = (r < n_res && rmean) ? rmean[r] : nullptr;
		ra.rinvstd[r] = (r < n_res && rinvstd) ? rinvstd[r] : nullptr;
		ra.rsums[r] = (r < n_res && rsums) ? rsums[r] : nullptr;
	}
	return 0;
}

static unsigned ew_grid(int64_t total) {
	int64_t g = ceil_div64(total, 256);
	return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

#define BN_ROWS_PER_THREAD 8

static int bn_act_fwd_impl(const void* y, void* z, int dtype, const float* scale, const float* shift, int n_res, const void* const* res,
                           const float* const* rscale, const float* const* rshift, int act, float act_lo, float act_hi, float dropout_p,
                           uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C, uint8_t* gate, int plane_dtype, void* stream) {
	CONVASR_CHECK_ARG(y && z && B > 0 && T > 0 && C > 0 && (C & 7) == 0, "bn_act_fwd: bad arguments (C must be a multiple of 8)");
	CONVASR_CHECK_ARG(!gate || act == CONVASR_ACT_RELU || act == CONVASR_ACT_HARDTANH || act == CONVASR_ACT_NONE, "bn_act_fwd: the one-bit gate needs an activation whose derivative is 0 or 1");
	CONVASR_CHECK_ARG((scale == nullptr) == (shift == nullptr) && dropout_p >= 0.f && dropout_p < 1.f, "bn_act_fwd: bad scale/shift/dropout");
	CONVASR_CHECK_ARG((int64_t)B * T < (1ll << 31), "bn_act_fwd: B * T must fit in 31 bits");
	BnActParams p = {};
	p.y = y; p.out = z; p.scale = scale; p.shift = shift; p.xlen = xlen; p.act = act; p.lo = act_lo; p.hi = act_hi; p.gate_out = gate;
	set_dropout(p, dropout_p, seed, offset, step_key);
	p.B = B; p.T = T; p.C = C;
	ResArgs ra;
	if (int rc = fill_res(ra, n_res, res, rscale, rshift, nullptr, nullptr, nullptr)) return rc;
	dim3 grid, block;
	row_walk_config(p, BN_ROWS_PER_THREAD, grid, block);
	if (dtype != CONVASR_F32 && !convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_act_fwd: dtype %d", dtype);
	const int mode = (n_res > 0 ? 1 : 0) | (p.drop_thr ? 2 : 0) | (gate ? 4 : 0) | (scale ? 8 : 0) | (act == CONVASR_ACT_LEAKY_RELU ? 16 : 0);
	if (plane_dtype >= 0) {
		// plane output: fp32 in, the training launches only (BN scale / shift present, a clamp-type activation)
		if (dtype != CONVASR_F32 || !convasr_is_half(plane_dtype) || mode < 8 || mode > 15) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_act_fwd_split3: fp32 input with scale / shift and a clamp-type activation only (mode %d)", mode);
#define BN_FWD_SPLIT_CASE(M) case M: \
		if (plane_dtype == CONVASR_BF16) hipLaunchKernelGGL((bn_act_fwd_kernel<float, M, 1>), grid, block, 0, (hipStream_t)stream, p, ra); \
		else hipLaunchKernelGGL((bn_act_fwd_kernel<float, M, 2>), grid, block, 0, (hipStream_t)stream, p, ra); \
		break;
		switch (mode) { BN_FWD_SPLIT_CASE(8) BN_FWD_SPLIT_CASE(9) BN_FWD_SPLIT_CASE(10) BN_FWD_SPLIT_CASE(11) BN_FWD_SPLIT_CASE(12) BN_FWD_SPLIT_CASE(13) BN_FWD_SPLIT_CASE(14) BN_FWD_SPLIT_CASE(15) }
#undef BN_FWD_SPLIT_CASE
		CONVASR_CHECK_LAUNCH("bn_act_fwd_split3");
		return 0;
	}
#define BN_FWD_CASE(M) case M: \
		if (dtype == CONVASR_F32) hipLaunchKernelGGL((bn_act_fwd_kernel<float, M>), grid, block, 0, (hipStream_t)stream, p, ra); \
		else if (dtype == CONVASR_F16) hipLaunchKernelGGL((bn_act_fwd_kernel<f16_t, M>), grid, block, 0, (hipStream_t)stream, p, ra); \
		else hipLaunchKernelGGL((bn_act_fwd_kernel<bf16_t, M>), grid, block, 0, (hipStream_t)stream, p, ra); \
		break;
	switch (mode) {
		BN_FWD_CASE(0) BN_FWD_CASE(1) BN_FWD_CASE(2) BN_FWD_CASE(3) BN_FWD_CASE(4) BN_FWD_CASE(5) BN_FWD_CASE(6) BN_FWD_CASE(7)
		BN_FWD_CASE(8) BN_FWD_CASE(9) BN_FWD_CASE(10) BN_FWD_CASE(11) BN_FWD_CASE(12) BN_FWD_CASE(13) BN_FWD_CASE(14) BN_FWD_CASE(15)
		BN_FWD_CASE(16) BN_FWD_CASE(17) BN_FWD_CASE(18) BN_FWD_CASE(19) BN_FWD_CASE(24) BN_FWD_CASE(25) BN_FWD_CASE(26) BN_FWD_CASE(27)  // leaky-relu: no gate bits
	}
#undef BN_FWD_CASE
	CONVASR_CHECK_LAUNCH("bn_act_fwd");
	return 0;
}

extern "C" int convasr_bn_act_fwd(const void* y, void* z, int dtype, const float* scale, const float* shift, int n_res, const void* const* res,
                                  const float* const* rscale, const float* const* rshift, int act, float act_lo, float act_hi, float dropout_p,
                                  uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C, uint8_t* gate, void* stream) {
	return bn_act_fwd_impl(y, z, dtype, scale, shift, n_res, res, rscale, rshift, act, act_lo, act_hi, dropout_p, seed, offset, step_key, xlen, B, T, C, gate, -1, stream);
}

extern "C" int convasr_bn_act_fwd_split3(const void* y, void* z3, int plane_dtype, const float* scale, const float* shift, int n_res, const void* const* res,
                                         const float* const* rscale, const float* const* rshift, int act, float act_lo, float act_hi, float dropout_p,
                                         uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, int B, int T, int C, uint8_t* gate, void* stream) {
	return bn_act_fwd_impl(y, z3, CONVASR_F32, scale, shift, n_res, res, rscale, rshift, act, act_lo, act_hi, dropout_p, seed, offset, step_key, xlen, B, T, C, gate, plane_dtype, stream);
}

// ------------------------------------------------------------------------------------------------ backward pass 1: g and channel sums
// Per-channel partial sums are combined across the row-lanes of a block in LDS and stored (plain coalesced stores) to the
// workspace [set][block][2C]; bn_bwd_finalize_kernel sums the blocks in fp64: deterministic, and no contended fp64 atomics
// (4096 blocks x 2C atomics per call cost 4x the streaming time).
template <typename T, bool RES> __global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(BnActParams p, ResArgs ra, float* __restrict__ ws) {
	fold_step_key(p);
	const ActConst ac = act_const(p.act, p.lo, p.hi);  // the activation kind folded into constants once: no per-element switch
	__shared__ float red[256][17];
	const int c8 = p.C >> 3;
	const int cgroups = p.cgroups, rlanes = p.rlanes;
	const int cg = threadIdx.x % cgroups, rl = threadIdx.x / cgroups;
	for (int cbase = blockIdx.y * cgroups; cbase < c8; cbase += gridDim.y * cgroups) {
		const int c = (cbase + cg) << 3;
		const bool cok = cbase + cg < c8;
		float s1[8], s2[8], rs1[RES ? 2 : 1][8], rs2[RES ? 2 : 1][8];  // RES: batch-normed residual inputs whose sums are wanted too
#pragma unroll
		for (int k = 0; k < 8; ++k) { s1[k] = s2[k] = 0.f; rs1[0][k] = rs2[0][k] = 0.f; if (RES) rs1[1][k] = rs2[1][k] = 0.f; }
		float mean[8], istd[8], sc[8], sh[8];
		if (cok && p.mean) { load8<float>(p.mean + c, mean); load8<float>(p.invstd + c, istd); }
		if (cok && p.scale) { load8<float>(p.scale + c, sc); load8<float>(p.shift + c, sh); }
		struct Pair { Raw8<T> y, dz; };
		const T* const py = reinterpret_cast<const T*>(p.y);
		const T* const pdz = reinterpret_cast<const T*>(p.dz);
		{
			auto load_row = [&](const RowWalk& w) { Pair q; q.y = raw_load8(py + w.idx); q.dz = raw_load8(pdz + w.idx); return q; };
			auto use_row = [&](const RowWalk& w, const Pair& q) {
					float g[8];
					if (w.masked()) {
#pragma unroll
						for (int k = 0; k < 8; ++k) g[k] = 0.f;
					} else {
						float yv[8], pre[8], dz[8];
						pre_act8<T>(p, ra, w.idx, c, sc, sh, q.y, yv, pre);
						unpack8(q.dz, dz);
#pragma unroll
						for (int k = 0; k < 8; ++k) g[k] = dz[k] * act_grad(pre[k], ac);
						if (p.drop_thr) {
							float keep[8];
							dropout_keep8(p, w.idx, keep);
#pragma unroll
							for (int k = 0; k < 8; ++k) g[k] *= keep[k];
						}
						if (p.mean) {
#pragma unroll
							for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] += g[k] * (yv[k] - mean[k]) * istd[k]; }
						}
						for (int r = 0; RES && r < ra.n && r < 2; ++r) {
							if (ra.rsums[r]) {
								float rv[8], rm[8], ri[8];
								load8<T>(reinterpret_cast<const T*>(ra.res[r]) + w.idx, rv);
								load8<float>(ra.rmean[r] + c, rm);
								load8<float>(ra.rinvstd[r] + c, ri);
#pragma unroll
								for (int k = 0; k < 8; ++k) { rs1[RES ? r : 0][k] += g[k]; rs2[RES ? r : 0][k] += g[k] * (rv[k] - rm[k]) * ri[k]; }
							}
						}
					}
					if (p.out) store8<T>(reinterpret_cast<T*>(p.out) + w.idx, g);
				};
			if (cok) { if (RES) walk_rows1(p, rl, c, load_row, use_row); else walk_rows2(p, rl, c, load_row, use_row); }
		}
		// block reduction over the row-lanes that share a channel group
		auto reduce_to = [&](float (&a)[8], float (&bq)[8], int set) {
#pragma unroll
			for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = a[k]; red[threadIdx.x][8 + k] = bq[k]; }
			__syncthreads();
			if (rl == 0 && cok) {
				float* dst = ws + ((int64_t)set * gridDim.x + blockIdx.x) * 2 * p.C;
				float o1[8], o2[8];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					float u = 0.f, w2 = 0.f;
					for (int j = 0; j < rlanes; ++j) { u += red[j * cgroups + cg][k]; w2 += red[j * cgroups + cg][8 + k]; }
					o1[k] = u; o2[k] = w2;
				}
				store8<float>(dst + c, o1);
				store8<float>(dst + p.C + c, o2);
			}
			__syncthreads();
		};
		if (p.mean && p.sums) reduce_to(s1, s2, 0);
		for (int r = 0; RES && r < ra.n && r < 2; ++r)
			if (ra.rsums[r]) reduce_to(rs1[RES ? r : 0], rs2[RES ? r : 0], 1 + r);
	}
}

// The same pass for a residual-free layer whose forward stored its one-bit gradient gates: g = gate ? dz * keep_scale : 0, no
// pre-activation, no hash, no frame arithmetic -- three streams in (dz, y, one gate byte per 8 elements), two sums out.  Sums g and
// g * y per thread and centres / scales once per block: sum g * xhat = (sum g y - mean sum g) * invstd.
template <typename T> __global__ __launch_bounds__(256) void bn_act_bwd_reduce_gated_kernel(BnActParams p, float* __restrict__ ws) {
	__shared__ float red[256][17];
	const int c8 = p.C >> 3;
	const int cgroups = p.cgroups, rlanes = p.rlanes;
	const int cg = threadIdx.x % cgroups, rl = threadIdx.x / cgroups;
	const float gate_scale = p.drop_thr ? p.keep_scale : 1.f;
	for (int cbase = blockIdx.y * cgroups; cbase < c8; cbase += gridDim.y * cgroups) {
		const int c = (cbase + cg) << 3;
		const bool cok = cbase + cg < c8;
		float s1[8], s2[8], mean[8], istd[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
		if (cok) { load8<float>(p.mean + c, mean); load8<float>(p.invstd + c, istd); }
		struct Trip { Raw8<T> y, dz; unsigned gate; };
		const T* const py = reinterpret_cast<const T*>(p.y);
		const T* const pdz = reinterpret_cast<const T*>(p.dz);
		if (cok)
			walk_rows4(p, rl, c,
				[&](const RowWalk& w) { Trip q; q.y = raw_load8(py + w.idx); q.dz = raw_load8(pdz + w.idx); q.gate = p.gate_in[w.idx >> 3]; return q; },
				[&](const RowWalk& w, const Trip& q) {
					float dz[8], yv[8];
					unpack8(q.dz, dz);
					unpack8(q.y, yv);
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						const float g = ((q.gate >> k) & 1u) ? dz[k] * gate_scale : 0.f;
						s1[k] += g;
						s2[k] = fmaf(g, yv[k] - mean[k], s2[k]);  // centred per element like the ungated kernel: sum(g y) - mean sum(g) cancels catastrophically for |mean| >> std
					}
				});
#pragma unroll
		for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = s1[k]; red[threadIdx.x][8 + k] = cok ? s2[k] * istd[k] : 0.f; }
		__syncthreads();
		if (rl == 0 && cok) {
			float* dst = ws + (int64_t)blockIdx.x * 2 * p.C;
			float o1[8], o2[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				float u = 0.f, w2 = 0.f;
				for (int j = 0; j < rlanes; ++j) { u += red[j * cgroups + cg][k]; w2 += red[j * cgroups + cg][8 + k]; }
				o1[k] = u; o2[k] = w2;
			}
			store8<float>(dst + c, o1);
			store8<float>(dst + p.C + c, o2);
		}
		__syncthreads();
	}
}

// sums[set][.] = sum over blocks of ws[set][block][.], accumulated in fp64; for the main BN (set 0) optionally also the
// per-channel coefficients of pass 2 (dy = A*g + Bc*y + D) and the parameter gradients dgamma = sum g*xhat, dbeta = sum g.
// Block = 16 channels x 64 block-lanes (finalize_rows).
struct BnFinalizeSets {
	double* dst[3];
	const float* gamma; const float* mean; const float* invstd;
	float* coef; float* dgamma; float* dbeta;
	int accumulate; float invn;
};
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ ws, BnFinalizeSets sets, int nblocks, int C) {
	__shared__ double red[2][FIN_LANES][FIN_CH];
	const int set = blockIdx.y;
	if (sets.dst[set] == nullptr && !(set == 0 && (sets.coef || sets.dgamma || sets.dbeta))) return;
	const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
	double sg, sgx;
	if (finalize_rows(ws + (int64_t)set * nblocks * 2 * C, nblocks, C, red, sg, sgx)) {
		if (sets.dst[set]) { sets.dst[set][c] = sg; sets.dst[set][C + c] = sgx; }
		if (set == 0) {
			if (sets.coef) {
				const float gm = sets.gamma ? sets.gamma[c] : 1.f, is = sets.invstd[c], m = sets.mean[c];
				const float msg = (float)sg * sets.invn, msgx = (float)sgx * sets.invn;
				sets.coef[c] = gm * is;
				sets.coef[C + c] = -gm * is * is * msgx;
				sets.coef[2 * C + c] = gm * is * (m * is * msgx - msg);
			}
			if (sets.dgamma) sets.dgamma[c] = sets.accumulate ? sets.dgamma[c] + (float)sgx : (float)sgx;
			if (sets.dbeta) sets.dbeta[c] = sets.accumulate ? sets.dbeta[c] + (float)sg : (float)sg;
		}
	}
}

// The same finalize for the per-tile partial rows [rows][2][C] written by the fused dgrad epilogue of conv_v2s.hip: sums them in a
// fixed order (finalize_rows: 16 channels x 64 row-lanes per block) and emits coef / dgamma / dbeta.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_sums_kernel(const double* __restrict__ sums, int rows, BnFinalizeSets sets, int C) {
	__shared__ double red[2][FIN_LANES][FIN_CH];
	const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
	double sg, sgx;
	if (!finalize_rows(sums, rows, C, red, sg, sgx)) return;
	if (sets.coef) {
		const float gm = sets.gamma ? sets.gamma[c] : 1.f, is = sets.invstd[c], m = sets.mean[c];
		const float msg = (float)sg * sets.invn, msgx = (float)sgx * sets.invn;
		sets.coef[c] = gm * is;
		sets.coef[C + c] = -gm * is * is * msgx;
		sets.coef[2 * C + c] = gm * is * (m * is * msgx - msg);
	}
	if (sets.dgamma) sets.dgamma[c] = sets.accumulate ? sets.dgamma[c] + (float)sgx : (float)sgx;
	if (sets.dbeta) sets.dbeta[c] = sets.accumulate ? sets.dbeta[c] + (float)sg : (float)sg;
}

extern "C" int convasr_bn_bwd_finalize(const double* sums, int sums_rows, const float* gamma, const float* mean, const float* invstd, float* coef, float* dgamma,
                                       float* dbeta, int accumulate, int64_t n, int C, void* stream) {
	CONVASR_CHECK_ARG(sums && sums_rows > 0 && mean && invstd && n > 0 && C > 0, "bn_bwd_finalize: bad arguments");
	BnFinalizeSets sets = {};
	sets.gamma = gamma; sets.mean = mean; sets.invstd = invstd; sets.coef = coef; sets.dgamma = dgamma; sets.dbeta = dbeta;
	sets.accumulate = accumulate; sets.invn = 1.0f / (float)n;
	hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, sums, sums_rows, sets, C);
	CONVASR_CHECK_LAUNCH("bn_bwd_finalize");
	return 0;
}

// Grouped forms for a dense block's backward (main batch norm + up to twelve residual-branch batch norms of the same channel count):
// one finalize launch (blockIdx.y = which), one apply launch that reads g ONCE and writes every dy_i = A_i g + B_i y_i + D_i.
struct BnBwdFinGroup { const double* sums[BN_MAX_GROUP]; BnFinalizeSets sets[BN_MAX_GROUP]; int rows[BN_MAX_GROUP]; };
__global__ __launch_bounds__(1024) void bn_bwd_finalize_grouped_kernel(BnBwdFinGroup g, int C) {
	__shared__ double red[2][FIN_LANES][FIN_CH];
	const int q = blockIdx.y;
	const BnFinalizeSets& sets = g.sets[q];
	const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
	double sg, sgx;
	if (!finalize_rows(g.sums[q], g.rows[q], C, red, sg, sgx)) return;
	if (sets.coef) {
		const float gm = sets.gamma ? sets.gamma[c] : 1.f, is = sets.invstd[c], m = sets.mean[c];
		const float msg = (float)sg * sets.invn, msgx = (float)sgx * sets.invn;
		sets.coef[c] = gm * is;
		sets.coef[C + c] = -gm * is * is * msgx;
		sets.coef[2 * C + c] = gm * is * (m * is * msgx - msg);
	}
	if (sets.dgamma) sets.dgamma[c] = sets.accumulate ? sets.dgamma[c] + (float)sgx : (float)sgx;
	if (sets.dbeta) sets.dbeta[c] = sets.accumulate ? sets.dbeta[c] + (float)sg : (float)sg;
}

extern "C" int convasr_bn_bwd_finalize_grouped(int count, const double* const* sums, const int* sums_rows, const float* const* gamma, const float* const* mean,
                                               const float* const* invstd, float* const* coef, float* const* dgamma, float* const* dbeta, const int* accumulate,
                                               int64_t n, int C, void* stream) {
	CONVASR_CHECK_ARG(count > 0 && count <= BN_MAX_GROUP && sums && sums_rows && mean && invstd && n > 0 && C > 0, "bn_bwd_finalize_grouped: bad arguments (at most %d batch norms)", BN_MAX_GROUP);
	BnBwdFinGroup g = {};
	for (int i = 0; i < count; ++i) {
		CONVASR_CHECK_ARG(sums[i] && sums_rows[i] > 0 && mean[i] && invstd[i], "bn_bwd_finalize_grouped: batch norm %d has a NULL buffer", i);
		g.sums[i] = sums[i]; g.rows[i] = sums_rows[i];
		BnFinalizeSets& t = g.sets[i];
		t.gamma = gamma ? gamma[i] : nullptr; t.mean = mean[i]; t.invstd = invstd[i]; t.coef = coef ? coef[i] : nullptr; t.dgamma = dgamma ? dgamma[i] : nullptr; t.dbeta = dbeta ? dbeta[i] : nullptr;
		t.accumulate = accumulate ? accumulate[i] : 0; t.invn = 1.0f / (float)n;
	}
	hipLaunchKernelGGL(bn_bwd_finalize_grouped_kernel, dim3((C + FIN_CH - 1) / FIN_CH, count), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, g, C);
	CONVASR_CHECK_LAUNCH("bn_bwd_finalize_grouped");
	return 0;
}

struct BnApplyGroup { const void* y[BN_MAX_GROUP]; const float* coef[BN_MAX_GROUP]; void* dy[BN_MAX_GROUP]; int n; };
#define GA_ROWS 4  // (8 rows per thread: 150-178 VGPRs, two waves per SIMD, 0.62x the bandwidth of the one-output apply kernel; 4: see profiles/r05_ab_grouped_apply.json)
// thread = 8 channels x GA_ROWS rows (rows rl, rl + rlanes, ...): its g values stay in registers (packed) while it walks the problems; per
// problem the 24 coefficients are loaded once and the GA_ROWS y loads are issued together.
template <typename T> __global__ __launch_bounds__(256) void bn_bwd_apply_grouped_kernel(const T* __restrict__ g, BnApplyGroup grp, int64_t rows, int C, int cgroups, int rlanes) {
	const int c8 = C >> 3;
	const int cg = blockIdx.y * cgroups + threadIdx.x % cgroups, rl = threadIdx.x / cgroups;
	if (cg >= c8) return;
	const int c = cg << 3;
	const int64_t r0 = (int64_t)blockIdx.x * (rlanes * GA_ROWS) + rl;
	uint4 gr[GA_ROWS];
#pragma unroll
	for (int k = 0; k < GA_ROWS; ++k) {
		const int64_t r = r0 + (int64_t)k * rlanes;
		gr[k] = r < rows ? *reinterpret_cast<const uint4*>(g + r * C + c) : make_uint4(0u, 0u, 0u, 0u);
	}
	for (int q = 0; q < grp.n; ++q) {
		const T* const y = reinterpret_cast<const T*>(grp.y[q]);
		T* const dy = reinterpret_cast<T*>(grp.dy[q]);
		const float* const coef = grp.coef[q];
		float A[8], Bc[8], D[8];
		load8<float>(coef + c, A);
		load8<float>(coef + C + c, Bc);
		load8<float>(coef + 2 * C + c, D);
		uint4 yr[GA_ROWS];
#pragma unroll
		for (int k = 0; k < GA_ROWS; ++k) {
			const int64_t r = r0 + (int64_t)k * rlanes;
			yr[k] = r < rows ? *reinterpret_cast<const uint4*>(y + r * C + c) : make_uint4(0u, 0u, 0u, 0u);
		}
#pragma unroll
		for (int k = 0; k < GA_ROWS; ++k) {
			const int64_t r = r0 + (int64_t)k * rlanes;
			if (r >= rows) continue;
			float gv[8], yv[8], out[8];
			unpack16<T>(gr[k], gv);
			unpack16<T>(yr[k], yv);
#pragma unroll
			for (int e = 0; e < 8; ++e) out[e] = fmaf(A[e], gv[e], fmaf(Bc[e], yv[e], D[e]));
			store8<T>(dy + r * C + c, out);
		}
	}
}

extern "C" int convasr_bn_bwd_apply_grouped(const void* g, int count, const void* const* y, const float* const* coef, void* const* dy, int dtype, int B, int T, int C, void* stream) {
	CONVASR_CHECK_ARG(g && count > 0 && count <= BN_MAX_GROUP && y && coef && dy && B > 0 && T > 0 && C > 0 && (C & 7) == 0, "bn_bwd_apply_grouped: bad arguments (at most %d outputs, C a multiple of 8)", BN_MAX_GROUP);
	if (!convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_bwd_apply_grouped: dtype %d (16-bit storage only)", dtype);
	BnApplyGroup grp = {};
	grp.n = count;
	for (int i = 0; i < count; ++i) {
		CONVASR_CHECK_ARG(y[i] && coef[i] && dy[i], "bn_bwd_apply_grouped: output %d has a NULL buffer", i);
		grp.y[i] = y[i]; grp.coef[i] = coef[i]; grp.dy[i] = dy[i];
	}
	const int c8 = C >> 3, cgroups = c8 < 256 ? c8 : 256, rlanes = 256 / cgroups;
	const int64_t rows = (int64_t)B * T;
	dim3 grid((unsigned)ceil_div64(rows, (int64_t)rlanes * GA_ROWS), (unsigned)((c8 + cgroups - 1) / cgroups)), block(cgroups * rlanes);
	if (dtype == CONVASR_F16) hipLaunchKernelGGL((bn_bwd_apply_grouped_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, (const f16_t*)g, grp, rows, C, cgroups, rlanes);
	else hipLaunchKernelGGL((bn_bwd_apply_grouped_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)g, grp, rows, C, cgroups, rlanes);
	CONVASR_CHECK_LAUNCH("bn_bwd_apply_grouped");
	return 0;
}

// ---------------- a dense block's backward pass 1 in ONE sweep (16-bit storage, stored gates)
// g = gate ? dz * keep_scale : 0 (the forward pass stored, per element, whether the gradient passes: activation range, dropout, frame mask --
// so the pre-activation, i.e. the block output's TEN residual inputs, need not be re-read), written once; and for each of the M batch norms
// that feed the sum (the main one + the residual branches'): sum g (shared by all of them) and sum g * (y_q - mean_q), centred per element.
// The per-branch form (bn_act_bwd_reduce_kernel) re-derives the pre-activation from every input and takes one extra sweep over g per two
// branches: n + 3 + 3 ceil((n - 2) / 2) tensor sweeps for n branches against n + 3 here (27 -> 13 at n = 10).
// Thread = 4 channels (8-byte loads; M + 1 accumulators and M means of 4 floats each stay in registers), one row per trip with all M + 1
// loads in flight; a block owns a contiguous range of rows and every thread writes its partial sums itself (no cross-lane step):
// ws[block][M + 1][C] floats, plane 0 = sum g.
#define RM_MAX 13
struct ReduceMany { const void* y[RM_MAX]; const float* mean[RM_MAX]; };
template <typename T, int M> __global__ __launch_bounds__(256) void bn_bwd_reduce_many_kernel(const T* __restrict__ dz, const uint8_t* __restrict__ gate, float keep_scale, T* __restrict__ g_out,
                                                                                            ReduceMany rm, float* __restrict__ ws, int64_t rows, int C, int rows_per_block) {
	const int c4 = C >> 2;
	const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
	for (int cg = threadIdx.x; cg < c4; cg += blockDim.x) {
		const int c = cg << 2;
		float sg[4] = {0.f, 0.f, 0.f, 0.f}, acc[M][4], mean[M][4];
#pragma unroll
		for (int q = 0; q < M; ++q) {
			const float4 m4 = *reinterpret_cast<const float4*>(rm.mean[q] + c);
			mean[q][0] = m4.x; mean[q][1] = m4.y; mean[q][2] = m4.z; mean[q][3] = m4.w;
			acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
		}
		const int shift = c & 4;  // the 4 gate bits of this thread's channels: low or high nibble of the byte of its 8-channel group
		for (int64_t r = r0; r < r1; ++r) {
			const int64_t idx = r * C + c;
			const uint2 dzr = *reinterpret_cast<const uint2*>(dz + idx);
			const unsigned gb = ((unsigned)gate[idx >> 3] >> shift) & 15u;
			uint2 yr[M];
#pragma unroll
			for (int q = 0; q < M; ++q) yr[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(rm.y[q]) + idx);
			float dzv[8], g[4];  // (unpack16 takes 16 bytes: the upper half is padding)
			unpack16<T>(make_uint4(dzr.x, dzr.y, 0u, 0u), dzv);
#pragma unroll
			for (int k = 0; k < 4; ++k) { g[k] = ((gb >> k) & 1u) ? dzv[k] * keep_scale : 0.f; sg[k] += g[k]; }
			*reinterpret_cast<uint2*>(g_out + idx) = make_uint2(pack16<T>(g[0], g[1]), pack16<T>(g[2], g[3]));
#pragma unroll
			for (int q = 0; q < M; ++q) {
				float yv[8];
				unpack16<T>(make_uint4(yr[q].x, yr[q].y, 0u, 0u), yv);
#pragma unroll
				for (int k = 0; k < 4; ++k) acc[q][k] = fmaf(g[k], yv[k] - mean[q][k], acc[q][k]);
			}
		}
		float* const o = ws + (int64_t)blockIdx.x * (M + 1) * C + c;
		*reinterpret_cast<float4*>(o) = make_float4(sg[0], sg[1], sg[2], sg[3]);
#pragma unroll
		for (int q = 0; q < M; ++q) *reinterpret_cast<float4*>(o + (int64_t)(q + 1) * C) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
	}
}

// its finalize: set q (blockIdx.y) adds plane 0 and plane 1 + q of the blocks' partial rows in a fixed order (fp64), scales the second by
// invstd_q and emits coef / dgamma / dbeta exactly like bn_bwd_finalize_sums_kernel
struct ReduceManyFin { BnFinalizeSets sets[RM_MAX]; };
__global__ __launch_bounds__(1024) void bn_bwd_finalize_many_kernel(const float* __restrict__ ws, int nblocks, int planes, ReduceManyFin f, int C) {
	__shared__ double red[2][FIN_LANES][FIN_CH];
	const int q = blockIdx.y;
	const BnFinalizeSets& sets = f.sets[q];
	const int cl = threadIdx.x & (FIN_CH - 1), c = blockIdx.x * FIN_CH + cl, w = threadIdx.x >> 4;
	double a = 0, b2 = 0;
	if (c < C)
		for (int r = w; r < nblocks; r += FIN_LANES) { a += (double)ws[((int64_t)r * planes) * C + c]; b2 += (double)ws[((int64_t)r * planes + 1 + q) * C + c]; }
	red[0][w][cl] = a;
	red[1][w][cl] = b2;
	__syncthreads();
	if (w != 0 || c >= C) return;
	double sg = 0, sgx = 0;
	for (int i = 0; i < FIN_LANES; ++i) { sg += red[0][i][cl]; sgx += red[1][i][cl]; }
	sgx *= (double)sets.invstd[c];
	if (sets.coef) {
		const float gm = sets.gamma ? sets.gamma[c] : 1.f, is = sets.invstd[c], m = sets.mean[c];
		const float msg = (float)sg * sets.invn, msgx = (float)sgx * sets.invn;
		sets.coef[c] = gm * is;
		sets.coef[C + c] = -gm * is * is * msgx;
		sets.coef[2 * C + c] = gm * is * (m * is * msgx - msg);
	}
	if (sets.dgamma) sets.dgamma[c] = sets.accumulate ? sets.dgamma[c] + (float)sgx : (float)sgx;
	if (sets.dbeta) sets.dbeta[c] = sets.accumulate ? sets.dbeta[c] + (float)sg : (float)sg;
}

static int reduce_many_blocks(int64_t rows, int& rows_per_block) {
	const int64_t want = 768;  // three per CU, like the other backward reduce kernels
	rows_per_block = (int)ceil_div64(rows, want);
	if (rows_per_block < 1) rows_per_block = 1;
	return (int)ceil_div64(rows, rows_per_block);
}

extern "C" int64_t convasr_bn_bwd_reduce_many_workspace_bytes(int count, int B, int T, int C) {
	if (count <= 0 || count > RM_MAX || B <= 0 || T <= 0 || C <= 0) return -1;
	int rpb;
	return (int64_t)reduce_many_blocks((int64_t)B * T, rpb) * (count + 1) * C * 4;
}

template <typename T, int M> static void launch_reduce_many(const void* dz, const uint8_t* gate, float keep_scale, void* g, const ReduceMany& rm, float* ws, int64_t rows, int C, hipStream_t s) {
	int rpb;
	const int blocks = reduce_many_blocks(rows, rpb);
	int threads = ((C >> 2) + 63) / 64 * 64;
	if (threads > 256) threads = 256;
	hipLaunchKernelGGL((bn_bwd_reduce_many_kernel<T, M>), dim3(blocks), dim3(threads), 0, s, (const T*)dz, gate, keep_scale, (T*)g, rm, ws, rows, C, rpb);
}

extern "C" int convasr_bn_bwd_reduce_many(const void* dz, const uint8_t* gate, float dropout_p, void* g, int count, const void* const* y, const float* const* mean, const float* const* invstd,
                                          const float* const* gamma, float* const* coef, float* const* dgamma, float* const* dbeta, const int* accumulate, void* workspace,
                                          int dtype, int B, int T, int C, void* stream) {
	CONVASR_CHECK_ARG(dz && gate && g && count > 0 && count <= RM_MAX && y && mean && invstd && workspace && B > 0 && T > 0 && C > 0 && (C & 7) == 0 && dropout_p >= 0.f && dropout_p < 1.f, "bn_bwd_reduce_many: bad arguments (at most %d batch norms, C a multiple of 8)", RM_MAX);
	if (!convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_bwd_reduce_many: dtype %d (16-bit storage only)", dtype);
	ReduceMany rm = {};
	ReduceManyFin fin = {};
	const int64_t rows = (int64_t)B * T;
	for (int i = 0; i < count; ++i) {
		CONVASR_CHECK_ARG(y[i] && mean[i] && invstd[i], "bn_bwd_reduce_many: batch norm %d has a NULL buffer", i);
		rm.y[i] = y[i]; rm.mean[i] = mean[i];
		BnFinalizeSets& t = fin.sets[i];
		t.gamma = gamma ? gamma[i] : nullptr; t.mean = mean[i]; t.invstd = invstd[i]; t.coef = coef ? coef[i] : nullptr; t.dgamma = dgamma ? dgamma[i] : nullptr; t.dbeta = dbeta ? dbeta[i] : nullptr;
		t.accumulate = accumulate ? accumulate[i] : 0; t.invn = 1.0f / (float)rows;
	}
	unsigned thr = (unsigned)lrintf(dropout_p * 65536.f);
	if (thr > 65535u) thr = 65535u;
	const float keep_scale = thr ? 65536.f / (float)(65536u - thr) : 1.f;
	hipStream_t s = (hipStream_t)stream;
	float* const ws = (float*)workspace;
	const uint8_t* const gt = gate;
#define RM_CASE(M_) case M_: if (dtype == CONVASR_F16) launch_reduce_many<f16_t, M_>(dz, gt, keep_scale, g, rm, ws, rows, C, s); else launch_reduce_many<bf16_t, M_>(dz, gt, keep_scale, g, rm, ws, rows, C, s); break;
	switch (count) { RM_CASE(1) RM_CASE(2) RM_CASE(3) RM_CASE(4) RM_CASE(5) RM_CASE(6) RM_CASE(7) RM_CASE(8) RM_CASE(9) RM_CASE(10) RM_CASE(11) RM_CASE(12) RM_CASE(13) }
#undef RM_CASE
	CONVASR_CHECK_LAUNCH("bn_bwd_reduce_many");
	int rpb;
	const int blocks = reduce_many_blocks(rows, rpb);
	hipLaunchKernelGGL(bn_bwd_finalize_many_kernel, dim3((C + FIN_CH - 1) / FIN_CH, count), dim3(FIN_CH * FIN_LANES), 0, s, (const float*)ws, blocks, count + 1, fin, C);
	CONVASR_CHECK_LAUNCH("bn_bwd_finalize_many");
	return 0;
}

// at most BN_BWD_MAX_BLOCKS blocks (3 per CU): the finalize kernel reads one partial row per block and set (more blocks make THAT
// kernel the cost: 2,048 partial rows of 2 C floats are 12 MB for it to walk); bytes in flight come from rows per trip instead
#define BN_BWD_MAX_BLOCKS 768
static int bn_bwd_max_blocks() {
	static int n = 0;
	if (!n) { const char* e = getenv("CONVASR_BN_BWD_BLOCKS"); n = e ? atoi(e) : BN_BWD_MAX_BLOCKS; if (n <= 0) n = BN_BWD_MAX_BLOCKS; }  // (measurement hook)
	return n;
}
static void bn_bwd_config(BnActParams& p, dim3& grid, dim3& block) {
	row_walk_config(p, BN_ROWS_PER_THREAD, grid, block);
	if ((int)grid.x > bn_bwd_max_blocks()) {
		const int64_t rows = (int64_t)p.B * p.T;
		const int64_t per = ceil_div64(rows, bn_bwd_max_blocks());
		p.rows_per_block = (int)(ceil_div64(per, p.rlanes) * p.rlanes);
		grid.x = (unsigned)ceil_div64(rows, p.rows_per_block);
	}
}

extern "C" int64_t convasr_bn_bwd_workspace_bytes(int B, int T, int C) {
	BnActParams p = {};
	p.B = B; p.T = T; p.C = C;
	dim3 grid, block;
	bn_bwd_config(p, grid, block);
	return (int64_t)3 * grid.x * 2 * C * (int64_t)sizeof(float);
}

extern "C" int convasr_bn_act_bwd_reduce(const void* dz, const void* y, void* g, int dtype, const float* scale, const float* shift, const float* mean,
                                         const float* invstd, int n_res, const void* const* res, const float* const* rscale, const float* const* rshift,
                                         const float* const* rmean, const float* const* rinvstd, double* const* rsums, int act, float act_lo, float act_hi,
                                         float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key, const float* xlen, double* sums, void* workspace, const float* gamma, float* coef, float* dgamma, float* dbeta,
                                         int accumulate, int B, int T, int C, const uint8_t* gate, void* stream) {
	CONVASR_CHECK_ARG(!gate || (n_res == 0 && !g && mean && (act == CONVASR_ACT_RELU || act == CONVASR_ACT_HARDTANH || act == CONVASR_ACT_NONE)), "bn_act_bwd_reduce: the one-bit gate form takes no residuals, writes no g, needs mean / invstd and an activation whose derivative is 0 or 1");
	CONVASR_CHECK_ARG(dz && y && B > 0 && T > 0 && C > 0 && (C & 7) == 0, "bn_act_bwd_reduce: bad arguments (C must be a multiple of 8)");
	CONVASR_CHECK_ARG((mean == nullptr) == (invstd == nullptr), "bn_act_bwd_reduce: mean and invstd go together");
	CONVASR_CHECK_ARG((int64_t)B * T < (1ll << 31), "bn_act_bwd_reduce: B * T must fit in 31 bits");
	BnActParams p = {};
	p.y = y; p.dz = dz; p.out = g; p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.xlen = xlen; p.sums = sums;
	p.act = act; p.lo = act_lo; p.hi = act_hi; p.B = B; p.T = T; p.C = C;
	set_dropout(p, dropout_p, seed, offset, step_key);
	ResArgs ra;
	if (int rc = fill_res(ra, n_res, res, rscale, rshift, rmean, rinvstd, rsums)) return rc;
	for (int r = 2; r < n_res; ++r) if (ra.rsums[r]) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_act_bwd_reduce: batch-normed residuals beyond the first two must be reduced by separate calls");
	dim3 grid, block;
	bn_bwd_config(p, grid, block);
	if (mean && !sums) { p.sums = reinterpret_cast<double*>(workspace); }  // (only a non-null marker: block partials always go to the workspace)
	bool any = mean != nullptr;
	for (int r = 0; r < n_res; ++r) any = any || ra.rsums[r] != nullptr;
	CONVASR_CHECK_ARG(!(coef || dgamma || dbeta) || mean, "bn_act_bwd_reduce: coef / dgamma / dbeta need the main batch norm's mean / invstd");
	CONVASR_CHECK_ARG(!any || workspace, "bn_act_bwd_reduce: workspace required when sums are requested");
	hipStream_t st = (hipStream_t)stream;
	bool res_sums = false;
	for (int r = 0; r < n_res; ++r) res_sums = res_sums || ra.rsums[r] != nullptr;
	p.gate_in = gate;
	if (gate) { BN_DISPATCH("bn_act_bwd_reduce", dtype, T, hipLaunchKernelGGL((bn_act_bwd_reduce_gated_kernel<T>), grid, block, 0, st, p, (float*)workspace)); }
	else BN_DISPATCH("bn_act_bwd_reduce", dtype, T,
		if (res_sums) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, true>), grid, block, 0, st, p, ra, (float*)workspace);
		else hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, false>), grid, block, 0, st, p, ra, (float*)workspace));
	if (any) {
		BnFinalizeSets sets;
		sets.dst[0] = (mean && sums) ? sums : nullptr;
		sets.dst[1] = n_res > 0 ? ra.rsums[0] : nullptr;
		sets.dst[2] = n_res > 1 ? ra.rsums[1] : nullptr;
		sets.gamma = gamma; sets.mean = mean; sets.invstd = invstd; sets.coef = coef; sets.dgamma = dgamma; sets.dbeta = dbeta;
		sets.accumulate = accumulate; sets.invn = 1.0f / (float)((int64_t)B * T);
		hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH, 3), dim3(FIN_CH * FIN_LANES), 0, st, (const float*)workspace, sets, (int)grid.x, C);
	}
	CONVASR_CHECK_LAUNCH("bn_act_bwd_reduce");
	return 0;
}

// ------------------------------------------------------------------------------------------------ backward pass 2
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ dy, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd, const double* __restrict__ sums,
                                                           int64_t rows, int C) {
	const int c8 = C >> 3;
	const int64_t total = rows * c8;
	const float invn = 1.0f / (float)rows;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
		const int c = (int)(i % c8) << 3;
		const int64_t idx = (i / c8) * C + c;
		float gv[8], yv[8], m[8], is[8], gm[8], out[8];
		load8<T>(g + idx, gv);
		load8<T>(y + idx, yv);
		load8<float>(mean + c, m);
		load8<float>(invstd + c, is);
		if (gamma) load8<float>(gamma + c, gm);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const float sg = (float)sums[c + k] * invn, sgx = (float)sums[C + c + k] * invn;
			const float xhat = (yv[k] - m[k]) * is[k];
			out[k] = (gamma ? gm[k] : 1.f) * is[k] * (gv[k] - sg - xhat * sgx);
		}
		store8<T>(dy + idx, out);
	}
}

// dy = A[c] * g + Bc[c] * y + D[c] with g either given (FROM_DZ = false) or recomputed from dz: g = dz * act'(pre) * dropout * mask
// SRC 0: g given; 1: g re-derived from dz (act', dropout hash, frame mask); 2: g = dz gated by the forward pass's stored bits (compile-time:
// the gated form carries none of the re-derivation's registers -- 95 -> ~50 VGPRs -- or code)
template <typename T, int SRC, int SPLIT = 0> __global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BnActParams p, const float* __restrict__ coef, T* __restrict__ dy) {  // (SPLIT: dy as its 16-bit planes, order 1 = an output gradient: see bn_act_fwd_kernel)
	constexpr bool FROM_DZ = SRC != 0, GATED = SRC == 2;
	if (FROM_DZ && !GATED) fold_step_key(p);
	const ActConst ac = act_const(p.act, p.lo, p.hi);  // the activation kind folded into constants once: no per-element switch
	const int c8 = p.C >> 3;
	const int cg = threadIdx.x % p.cgroups, rl = threadIdx.x / p.cgroups;
	ResArgs none;
	none.n = 0;
	for (int cbase = blockIdx.y * p.cgroups; cbase < c8; cbase += gridDim.y * p.cgroups) {
		if (cbase + cg >= c8) continue;
		const int c = (cbase + cg) << 3;
		float A[8], Bc[8], D[8], sc[8], sh[8];
		load8<float>(coef + c, A);
		load8<float>(coef + p.C + c, Bc);
		load8<float>(coef + 2 * p.C + c, D);
		if (FROM_DZ && !GATED && p.scale) { load8<float>(p.scale + c, sc); load8<float>(p.shift + c, sh); }
		struct Pair { Raw8<T> y, dz; unsigned gate; };
		const T* const py = reinterpret_cast<const T*>(p.y);
		const T* const pdz = reinterpret_cast<const T*>(p.dz);
		const float gate_scale = p.drop_thr ? p.keep_scale : 1.f;
		WALK_ROWS_FWD(p, rl, c,
			[&](const RowWalk& w) { Pair q; q.y = raw_load8(py + w.idx); q.dz = raw_load8(pdz + w.idx); q.gate = GATED ? p.gate_in[w.idx >> 3] : 0u; return q; },
			[&](const RowWalk& w, const Pair& q) {
				float yv[8], pre[8], g[8], out[8];
				if (GATED) {
					// the forward pass stored, per element, whether the gradient passes (activation range, dropout, frame mask): g = dz * keep_scale or 0
					float dz[8];
					unpack8(q.dz, dz);
					unpack8(q.y, yv);
#pragma unroll
					for (int k = 0; k < 8; ++k) g[k] = ((q.gate >> k) & 1u) ? dz[k] * gate_scale : 0.f;
				} else if (FROM_DZ) {
					pre_act8<T>(p, none, w.idx, c, sc, sh, q.y, yv, pre);
					if (w.masked()) {
#pragma unroll
						for (int k = 0; k < 8; ++k) g[k] = 0.f;
					} else {
						float dz[8];
						unpack8(q.dz, dz);
#pragma unroll
						for (int k = 0; k < 8; ++k) g[k] = dz[k] * act_grad(pre[k], ac);
						if (p.drop_thr) {
							float keep[8];
							dropout_keep8(p, w.idx, keep);
#pragma unroll
							for (int k = 0; k < 8; ++k) g[k] *= keep[k];
						}
					}
				} else {
					unpack8(q.dz, g);
					unpack8(q.y, yv);
				}
#pragma unroll
				for (int k = 0; k < 8; ++k) out[k] = fmaf(A[k], g[k], fmaf(Bc[k], yv[k], D[k]));
				if constexpr (SPLIT == 0) store8<T>(dy + w.idx, out);
				else if constexpr (SPLIT >= 3) store8<typename PlaneType<SPLIT>::H>(reinterpret_cast<typename PlaneType<SPLIT>::H*>(dy) + w.idx, out);
				else split3_store8<typename PlaneType<SPLIT>::H>(reinterpret_cast<typename PlaneType<SPLIT>::H*>(dy) + 3 * w.idx - 2 * c, p.C, 1, out);
			});
	}
}

static int bn_act_bwd_apply_impl(const void* dz_or_g, const void* y, void* dy, int dtype, const float* coef, int from_dz, const float* scale,
                                 const float* shift, int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset,
                                 const uint64_t* step_key, const float* xlen, int B, int T, int C, const uint8_t* gate, int plane_dtype, void* stream, int hi_only = 0) {
	CONVASR_CHECK_ARG(dz_or_g && y && dy && coef && B > 0 && T > 0 && C > 0 && (C & 7) == 0, "bn_act_bwd_apply: bad arguments (C must be a multiple of 8)");
	CONVASR_CHECK_ARG(!gate || (from_dz && (act == CONVASR_ACT_RELU || act == CONVASR_ACT_HARDTANH || act == CONVASR_ACT_NONE)), "bn_act_bwd_apply: the one-bit gate needs from_dz and an activation whose derivative is 0 or 1");
	CONVASR_CHECK_ARG((int64_t)B * T < (1ll << 31), "bn_act_bwd_apply: B * T must fit in 31 bits");
	BnActParams p = {};
	p.y = y; p.dz = dz_or_g; p.scale = scale; p.shift = shift; p.xlen = xlen; p.act = act; p.lo = act_lo; p.hi = act_hi; p.gate_in = gate;
	set_dropout(p, dropout_p, seed, offset, step_key);
	p.B = B; p.T = T; p.C = C;
	dim3 grid, block;
	row_walk_config(p, BN_ROWS_PER_THREAD, grid, block);
	hipStream_t s = (hipStream_t)stream;
	if (plane_dtype >= 0) {
		if (dtype != CONVASR_F32 || !convasr_is_half(plane_dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "bn_act_bwd_apply_split3: fp32 input, bf16 / fp16 planes");
#define BN_APPLY_SPLIT(S) \
		if (from_dz && gate) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<float, 2, S>), grid, block, 0, s, p, coef, (float*)dy); \
		else if (from_dz) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<float, 1, S>), grid, block, 0, s, p, coef, (float*)dy); \
		else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<float, 0, S>), grid, block, 0, s, p, coef, (float*)dy);
		if (hi_only) { if (plane_dtype == CONVASR_BF16) { BN_APPLY_SPLIT(3) } else { BN_APPLY_SPLIT(4) } }
		else if (plane_dtype == CONVASR_BF16) { BN_APPLY_SPLIT(1) } else { BN_APPLY_SPLIT(2) }
#undef BN_APPLY_SPLIT
		CONVASR_CHECK_LAUNCH("bn_act_bwd_apply_split3");
		return 0;
	}
	BN_DISPATCH("bn_act_bwd_apply", dtype, T,
		if (from_dz && gate) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 2>), grid, block, 0, s, p, coef, (T*)dy);
		else if (from_dz) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 1>), grid, block, 0, s, p, coef, (T*)dy);
		else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 0>), grid, block, 0, s, p, coef, (T*)dy));
	CONVASR_CHECK_LAUNCH("bn_act_bwd_apply");
	return 0;
}

extern "C" int convasr_bn_act_bwd_apply(const void* dz_or_g, const void* y, void* dy, int dtype, const float* coef, int from_dz, const float* scale,
                                        const float* shift, int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset,
                                        const uint64_t* step_key, const float* xlen, int B, int T, int C, const uint8_t* gate, void* stream) {
	return bn_act_bwd_apply_impl(dz_or_g, y, dy, dtype, coef, from_dz, scale, shift, act, act_lo, act_hi, dropout_p, seed, offset, step_key, xlen, B, T, C, gate, -1, stream);
}

extern "C" int convasr_bn_act_bwd_apply_split3(const void* dz_or_g, const void* y, void* dy3, int plane_dtype, const float* coef, int from_dz, const float* scale,
                                               const float* shift, int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset,
                                               const uint64_t* step_key, const float* xlen, int B, int T, int C, const uint8_t* gate, void* stream) {
	return bn_act_bwd_apply_impl(dz_or_g, y, dy3, CONVASR_F32, coef, from_dz, scale, shift, act, act_lo, act_hi, dropout_p, seed, offset, step_key, xlen, B, T, C, gate, plane_dtype, stream);
}

// ... and with dy rounded once to a dense 16-bit tensor [B * T][C] (fp32 dz / y in): the output gradient of a conv whose backward runs as ONE
// 16-bit product per gradient while its forward ran split (compute types 'bf16x3f' / 'f16x3f').
extern "C" int convasr_bn_act_bwd_apply_to_half(const void* dz_or_g, const void* y, void* dy16, int out_dtype, const float* coef, int from_dz, const float* scale,
                                                const float* shift, int act, float act_lo, float act_hi, float dropout_p, uint64_t seed, uint64_t offset,
                                                const uint64_t* step_key, const float* xlen, int B, int T, int C, const uint8_t* gate, void* stream) {
	return bn_act_bwd_apply_impl(dz_or_g, y, dy16, CONVASR_F32, coef, from_dz, scale, shift, act, act_lo, act_hi, dropout_p, seed, offset, step_key, xlen, B, T, C, gate, out_dtype, stream, 1);
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int accumulate) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	const float dg = (float)sums[C + c], db = (float)sums[c];
	if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
	if (dbeta) dbeta[c] = accumulate ? dbeta[c] + db : db;
}

extern "C" int convasr_bn_bwd_apply(const void* g, const void* y, void* dy, int dtype, const float* gamma, const float* mean, const float* invstd,
                                    const double* sums, float* dgamma, float* dbeta, int accumulate, int B, int T, int C, void* stream) {
	CONVASR_CHECK_ARG(g && y && mean && invstd && sums && B > 0 && T > 0 && C > 0 && (C & 7) == 0, "bn_bwd_apply: bad arguments (C must be a multiple of 8)");
	hipStream_t s = (hipStream_t)stream;
	const int64_t rows = (int64_t)B * T;
	if (dy) {
		const int64_t total = rows * (C >> 3);
		BN_DISPATCH("bn_bwd_apply", dtype, T, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_grid(total)), dim3(256), 0, s, (const T*)g, (const T*)y, (T*)dy, gamma, mean, invstd, sums, rows, C));
		CONVASR_CHECK_LAUNCH("bn_bwd_apply");
	}
	if (dgamma || dbeta) {
		hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, dgamma, dbeta, C, accumulate);
		CONVASR_CHECK_LAUNCH("bn_param_grad");
	}
	return 0;
}
