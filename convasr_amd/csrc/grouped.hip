// Grouped Conv1d + bias (+ ReLU): the first half of the reference's separable block (models.py:50-64, JasperNetSeparable 1372-1374:
// nn.Conv1d(Cin, Cout, K, groups = G) -> ReLU -> nn.Conv1d(Cout, Cout, 1); the 1x1 half runs in the MFMA kernels).
// With G = 128 groups of 2-6 channels there is no matrix shape in it (22-150 multiply-adds per output): a vector-ALU kernel on
// channels-last (B, T, C) activations, lanes along the output channels, the input tile (+ halo) staged once in LDS.  HBM-bound for the
// shapes of JasperNetSeparable (one read of x, one write of y per pass).
#include "common.h"

#define GC_TT 16       // output frames per workgroup
#define GC_THREADS 256 // output channels per workgroup (lanes along channels)
#define GC_MAXJ 8      // channels per group this file handles on either side

struct GcParams {
	const void* x; const void* dy; const void* yact; void* out;
	const float* w; const float* bias;
	float* dw; float* dbias; float* ws;
	int64_t w_sco, w_sj, w_sk;  // element strides of the (Cout, Cin / G, K) parameter: the reference's layout or the training arena's tap-major view
	int B, Cin, Cout, Tin, Tout, K, stride, pad, G, relu;
};

// y[b,t,co] = act(bias[co] + sum_k sum_j x[b, t s + k - pad, g cgi + j] w[co][j][k]),  g = co / cgo
template <typename T> __global__ __launch_bounds__(GC_THREADS) void gc_fwd_kernel(GcParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	T* const xs = reinterpret_cast<T*>(smem);
	const int cgi = p.Cin / p.G, cgo = p.Cout / p.G;
	const int b = blockIdx.z, t0 = blockIdx.x * GC_TT, co0 = blockIdx.y * GC_THREADS;
	const int co = co0 + threadIdx.x;
	const int g_first = co0 / cgo, g_last = (min(co0 + GC_THREADS, p.Cout) - 1) / cgo;
	const int ci0 = g_first * cgi, nci = (g_last - g_first + 1) * cgi;
	const int rows = (GC_TT - 1) * p.stride + p.K;
	const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.Tin * p.Cin;
	for (int e = threadIdx.x; e < rows * nci; e += GC_THREADS) {
		const int r = e / nci, c = e - r * nci, tin = t0 * p.stride - p.pad + r;
		xs[e] = (tin >= 0 && tin < p.Tin) ? xb[(int64_t)tin * p.Cin + ci0 + c] : (T)0;
	}
	__syncthreads();
	if (co >= p.Cout) return;
	const int cl = (co / cgo - g_first) * cgi;  // this lane's group, as a column offset into the tile
	float acc[GC_TT];
	const float bs = p.bias ? p.bias[co] : 0.f;
#pragma unroll
	for (int t = 0; t < GC_TT; ++t) acc[t] = bs;
	for (int k = 0; k < p.K; ++k)
		for (int j = 0; j < cgi; ++j) {
			const float wv = p.w[co * p.w_sco + j * p.w_sj + k * p.w_sk];
#pragma unroll
			for (int t = 0; t < GC_TT; ++t) acc[t] = fmaf(Elem<T>::load(xs + (t * p.stride + k) * nci + cl + j), wv, acc[t]);
		}
	T* yb = reinterpret_cast<T*>(p.out) + (int64_t)b * p.Tout * p.Cout;
#pragma unroll
	for (int t = 0; t < GC_TT; ++t)
		if (t0 + t < p.Tout) Elem<T>::store(yb + (int64_t)(t0 + t) * p.Cout + co, p.relu ? fmaxf(acc[t], 0.f) : acc[t]);
}

// stride 1: dx[b,t,ci] = sum_k sum_m g[b, t + pad - k, g cgo + m] w[g cgo + m][j][k],  g[.] = dy (yact > 0 when relu),  j = ci - g cgi
template <typename T> __global__ __launch_bounds__(GC_THREADS) void gc_dgrad_kernel(GcParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	T* const gs = reinterpret_cast<T*>(smem);
	const int cgi = p.Cin / p.G, cgo = p.Cout / p.G;
	const int b = blockIdx.z, t0 = blockIdx.x * GC_TT, ci0 = blockIdx.y * GC_THREADS;
	const int ci = ci0 + threadIdx.x;
	const int g_first = ci0 / cgi, g_last = (min(ci0 + GC_THREADS, p.Cin) - 1) / cgi;
	const int co0 = g_first * cgo, nco = (g_last - g_first + 1) * cgo;
	const int rows = GC_TT + p.K - 1;  // dy rows t0 + pad - (K - 1) ... t0 + GC_TT - 1 + pad
	const int r0 = t0 + p.pad - (p.K - 1);
	const T* dyb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.Tout * p.Cout;
	const T* yb = p.yact ? reinterpret_cast<const T*>(p.yact) + (int64_t)b * p.Tout * p.Cout : nullptr;
	for (int e = threadIdx.x; e < rows * nco; e += GC_THREADS) {
		const int r = e / nco, c = e - r * nco, to = r0 + r;
		float v = 0.f;
		if (to >= 0 && to < p.Tout) {
			v = Elem<T>::load(dyb + (int64_t)to * p.Cout + co0 + c);
			if (yb && !(Elem<T>::load(yb + (int64_t)to * p.Cout + co0 + c) > 0.f)) v = 0.f;
		}
		Elem<T>::store(gs + e, v);
	}
	__syncthreads();
	if (ci >= p.Cin) return;
	const int g = ci / cgi, j = ci - g * cgi, cl = (g - g_first) * cgo;
	float acc[GC_TT];
#pragma unroll
	for (int t = 0; t < GC_TT; ++t) acc[t] = 0.f;
	for (int k = 0; k < p.K; ++k)
		for (int m = 0; m < cgo; ++m) {
			const float wv = p.w[(int64_t)(g * cgo + m) * p.w_sco + j * p.w_sj + k * p.w_sk];
#pragma unroll
			for (int t = 0; t < GC_TT; ++t) acc[t] = fmaf(Elem<T>::load(gs + (t + (p.K - 1) - k) * nco + cl + m), wv, acc[t]);  // row of dy frame t0 + t + pad - k
		}
	T* xb = reinterpret_cast<T*>(p.out) + (int64_t)b * p.Tin * p.Cin;
#pragma unroll
	for (int t = 0; t < GC_TT; ++t)
		if (t0 + t < p.Tin) Elem<T>::store(xb + (int64_t)(t0 + t) * p.Cin + ci, acc[t]);
}

// dw[co][j][k] = sum_(b,t) g[b,t,co] x[b, t s + k - pad, g cgi + j], dbias[co] = sum g: one workgroup per (utterance, 256 output channels)
// walks the utterance's frames and keeps its cgi x K sums per lane; partial sums [b][co][j][k] (+ [b][co] for the bias) go to the workspace
// and are added over b in a fixed order by gc_wgrad_reduce_kernel: no atomics, run-to-run identical.  KT = taps per pass (registers).
template <typename T, int KT> __global__ __launch_bounds__(GC_THREADS) void gc_wgrad_kernel(GcParams p, int k0) {
	const int cgi = p.Cin / p.G, cgo = p.Cout / p.G;
	const int b = blockIdx.z, co = blockIdx.y * GC_THREADS + threadIdx.x;
	if (co >= p.Cout) return;
	const int cbase = (co / cgo) * cgi;
	const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.Tin * p.Cin;
	const T* dyb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.Tout * p.Cout;
	const T* yb = p.yact ? reinterpret_cast<const T*>(p.yact) + (int64_t)b * p.Tout * p.Cout : nullptr;
	float acc[GC_MAXJ][KT], sb = 0.f;
#pragma unroll
	for (int j = 0; j < GC_MAXJ; ++j)
#pragma unroll
		for (int k = 0; k < KT; ++k) acc[j][k] = 0.f;
	for (int t = 0; t < p.Tout; ++t) {
		float gv = Elem<T>::load(dyb + (int64_t)t * p.Cout + co);
		if (yb && !(Elem<T>::load(yb + (int64_t)t * p.Cout + co) > 0.f)) gv = 0.f;
		sb += gv;
#pragma unroll
		for (int k = 0; k < KT; ++k) {
			const int tin = t * p.stride + k0 + k - p.pad;
			if (k0 + k < p.K && tin >= 0 && tin < p.Tin) {
#pragma unroll
				for (int j = 0; j < GC_MAXJ; ++j)
					if (j < cgi) acc[j][k] = fmaf(gv, Elem<T>::load(xb + (int64_t)tin * p.Cin + cbase + j), acc[j][k]);
			}
		}
	}
	float* part = p.ws + ((int64_t)b * p.Cout + co) * cgi * p.K;
#pragma unroll
	for (int j = 0; j < GC_MAXJ; ++j)
#pragma unroll
		for (int k = 0; k < KT; ++k)
			if (j < cgi && k0 + k < p.K) part[j * p.K + k0 + k] = acc[j][k];
	if (k0 == 0) p.ws[(int64_t)p.B * p.Cout * cgi * p.K + (int64_t)b * p.Cout + co] = sb;
}

__global__ __launch_bounds__(256) void gc_wgrad_reduce_kernel(GcParams p, int accumulate) {
	const int cgi = p.Cin / p.G;
	const int64_t per_b = (int64_t)p.Cout * cgi * p.K, n = per_b + p.Cout;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		float s = 0.f;
		if (i < per_b) {
			for (int b = 0; b < p.B; ++b) s += p.ws[(int64_t)b * per_b + i];
			const int k = (int)(i % p.K), j = (int)((i / p.K) % cgi), co = (int)(i / ((int64_t)p.K * cgi));
			float* dst = p.dw + co * p.w_sco + j * p.w_sj + k * p.w_sk;
			*dst = accumulate ? *dst + s : s;
		} else if (p.dbias) {
			const int co = (int)(i - per_b);
			for (int b = 0; b < p.B; ++b) s += p.ws[(int64_t)p.B * per_b + (int64_t)b * p.Cout + co];
			p.dbias[co] = accumulate ? p.dbias[co] + s : s;
		}
	}
}

static int gc_check(const char* what, int dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int G) {
	if (!(B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && Tout > 0 && K > 0 && stride > 0 && pad >= 0 && G > 0 && Cin % G == 0 && Cout % G == 0)) return convasr_fail(CONVASR_EINVAL, "%s: bad arguments", what);
	if (Cin / G > GC_MAXJ || Cout / G > GC_MAXJ) return convasr_fail(CONVASR_EUNSUPPORTED, "%s: at most %d channels per group on either side (got %d -> %d)", what, GC_MAXJ, Cin / G, Cout / G);
	if ((int64_t)Tout != ((int64_t)Tin + 2 * pad - (K - 1) - 1) / stride + 1) return convasr_fail(CONVASR_EINVAL, "%s: Tout %d inconsistent with Tin %d K %d stride %d pad %d", what, Tout, Tin, K, stride, pad);
	if (dtype != CONVASR_F32 && !convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "%s: dtype %d", what, dtype);
	return 0;
}

#define GC_DISPATCH(dtype, T, ...) do { if ((dtype) == CONVASR_F32) { typedef float T; __VA_ARGS__; } else if ((dtype) == CONVASR_F16) { typedef f16_t T; __VA_ARGS__; } else { typedef bf16_t T; __VA_ARGS__; } } while (0)

extern "C" int convasr_grouped_conv1d_fwd(const void* x, const float* w, int64_t w_sco, int64_t w_sj, int64_t w_sk, const float* bias, void* y, int dtype, int B, int Cin,
                                          int Cout, int Tin, int Tout, int K, int stride, int pad, int groups, int relu, void* stream) {
	if (int rc = gc_check("grouped_conv1d_fwd", dtype, B, Cin, Cout, Tin, Tout, K, stride, pad, groups)) return rc;
	CONVASR_CHECK_ARG(x && w && y, "grouped_conv1d_fwd: null pointer");
	GcParams p = {};
	p.x = x; p.w = w; p.bias = bias; p.out = y; p.w_sco = w_sco; p.w_sj = w_sj; p.w_sk = w_sk;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.pad = pad; p.G = groups; p.relu = relu;
	const int cgi = Cin / groups, cgo = Cout / groups;
	const int nci_max = ((GC_THREADS + cgo - 1) / cgo + 1) * cgi, rows = (GC_TT - 1) * stride + K;
	const size_t smem = (size_t)rows * nci_max * (dtype == CONVASR_F32 ? 4 : 2);
	CONVASR_CHECK_ARG(smem <= 64 * 1024, "grouped_conv1d_fwd: input tile needs %zu B of LDS", smem);
	dim3 grid((Tout + GC_TT - 1) / GC_TT, (Cout + GC_THREADS - 1) / GC_THREADS, B);
	GC_DISPATCH(dtype, T, hipLaunchKernelGGL((gc_fwd_kernel<T>), grid, dim3(GC_THREADS), smem, (hipStream_t)stream, p));
	CONVASR_CHECK_LAUNCH("grouped_conv1d_fwd");
	return 0;
}

extern "C" int convasr_grouped_conv1d_dgrad(const void* dy, const void* y_act, const float* w, int64_t w_sco, int64_t w_sj, int64_t w_sk, void* dx, int dtype, int B,
                                            int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int groups, void* stream) {
	if (int rc = gc_check("grouped_conv1d_dgrad", dtype, B, Cin, Cout, Tin, Tout, K, stride, pad, groups)) return rc;
	CONVASR_CHECK_ARG(dy && w && dx, "grouped_conv1d_dgrad: null pointer");
	if (stride != 1) return convasr_fail(CONVASR_EUNSUPPORTED, "grouped_conv1d_dgrad: stride %d (the separable blocks of the reference are stride 1)", stride);
	GcParams p = {};
	p.dy = dy; p.yact = y_act; p.w = w; p.out = dx; p.w_sco = w_sco; p.w_sj = w_sj; p.w_sk = w_sk;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.pad = pad; p.G = groups;
	const int cgi = Cin / groups, cgo = Cout / groups;
	const int nco_max = ((GC_THREADS + cgi - 1) / cgi + 1) * cgo, rows = GC_TT + K - 1;
	const size_t smem = (size_t)rows * nco_max * (dtype == CONVASR_F32 ? 4 : 2);
	CONVASR_CHECK_ARG(smem <= 64 * 1024, "grouped_conv1d_dgrad: gradient tile needs %zu B of LDS", smem);
	dim3 grid((Tin + GC_TT - 1) / GC_TT, (Cin + GC_THREADS - 1) / GC_THREADS, B);
	GC_DISPATCH(dtype, T, hipLaunchKernelGGL((gc_dgrad_kernel<T>), grid, dim3(GC_THREADS), smem, (hipStream_t)stream, p));
	CONVASR_CHECK_LAUNCH("grouped_conv1d_dgrad");
	return 0;
}

extern "C" int64_t convasr_grouped_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int K, int groups) {
	if (groups <= 0 || Cin % groups) return -1;
	return ((int64_t)B * Cout * (Cin / groups) * K + (int64_t)B * Cout) * 4;
}

extern "C" int convasr_grouped_conv1d_wgrad(const void* x, const void* dy, const void* y_act, float* dw, int64_t w_sco, int64_t w_sj, int64_t w_sk, float* dbias,
                                            void* workspace, int dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad, int groups,
                                            int accumulate, void* stream) {
	if (int rc = gc_check("grouped_conv1d_wgrad", dtype, B, Cin, Cout, Tin, Tout, K, stride, pad, groups)) return rc;
	CONVASR_CHECK_ARG(x && dy && dw && workspace, "grouped_conv1d_wgrad: null pointer");
	GcParams p = {};
	p.x = x; p.dy = dy; p.yact = y_act; p.dw = dw; p.dbias = dbias; p.ws = (float*)workspace; p.w_sco = w_sco; p.w_sj = w_sj; p.w_sk = w_sk;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.pad = pad; p.G = groups;
	dim3 grid(1, (Cout + GC_THREADS - 1) / GC_THREADS, B);
	constexpr int KT = 8;
	for (int k0 = 0; k0 < K; k0 += KT) GC_DISPATCH(dtype, T, hipLaunchKernelGGL((gc_wgrad_kernel<T, KT>), grid, dim3(GC_THREADS), 0, (hipStream_t)stream, p, k0));
	const int64_t n = (int64_t)Cout * (Cin / groups) * K + Cout;
	hipLaunchKernelGGL(gc_wgrad_reduce_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, p, accumulate);
	CONVASR_CHECK_LAUNCH("grouped_conv1d_wgrad");
	return 0;
}
