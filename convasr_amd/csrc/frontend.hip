// Log-mel frontend (reference: LogFilterBankFrontend.forward, models.py:565-597; normalize_signal, models.py:684-686).
//
// One pass over the waveform: every wave turns TWO frames into one 512-point complex FFT (frame A real, frame B imaginary),
// Stockham radix-8 x 3 in LDS, un-mixes the two real spectra, takes |.|^2, applies the (nmel x 257) mel matrix held
// transposed in LDS, adds the eps bias, takes the log and writes channels-last (B, F, nmel) fp32 rows.  The normalise /
// pre-emphasis / mask / reflect-left / zero-right padding of the reference are index arithmetic on the load -- the padded
// signal and the 197 MB complex spectrogram are never materialised.  HBM-bound: 4 B in per sample, 256 B out per frame.
#include "common.h"

#define FE_NFFT 512
#define FE_BINS 257
#define FE_WAVES 16
#define FE_SPAN 64  // LDS rows of the sparse mel table: melS[j][mel] = weight of bin klo[mel] + j (wider filters read the rest from global memory)

// one atomic per workgroup: thousands of wave-level atomicMax on the same B addresses serialise in L2 and cost 5x the streaming time
template <typename S> __global__ __launch_bounds__(256) void absmax_kernel(const S* __restrict__ x, int T, unsigned* __restrict__ out) {
	__shared__ float red[4];
	const int b = blockIdx.y;
	const S* xb = x + (int64_t)b * T;
	constexpr int V = 16 / sizeof(S);
	float m = 0.f;
	const bool aligned = ((reinterpret_cast<uintptr_t>(xb) & 15) == 0);
	const int nv = aligned ? T / V : 0;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
		const uint4 raw = reinterpret_cast<const uint4*>(xb)[i];
		S e[V];
		__builtin_memcpy(e, &raw, 16);
#pragma unroll
		for (int k = 0; k < V; ++k) m = fmaxf(m, fabsf((float)e[k]));
	}
	for (int i = nv * V + blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) m = fmaxf(m, fabsf((float)xb[i]));
	m = wave_max(m);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) atomicMax(out + b, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));  // non-negative floats order like their bit patterns
}

extern "C" int convasr_signal_absmax(const void* signal, int signal_dtype, int B, int T, float* absmax, void* stream) {
	CONVASR_CHECK_ARG(signal && absmax && B > 0 && T > 0, "signal_absmax: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	if (hipMemsetAsync(absmax, 0, sizeof(float) * B, s) != hipSuccess) return convasr_fail(CONVASR_ELAUNCH, "signal_absmax: memset failed");
	int gx = (T + 256 * 32 - 1) / (256 * 32);
	if (gx > 32) gx = 32;
	if (signal_dtype == CONVASR_F32) hipLaunchKernelGGL(absmax_kernel<float>, dim3(gx, B), dim3(256), 0, s, (const float*)signal, T, (unsigned*)absmax);
	else if (signal_dtype == CONVASR_I16) hipLaunchKernelGGL(absmax_kernel<short>, dim3(gx, B), dim3(256), 0, s, (const short*)signal, T, (unsigned*)absmax);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "signal_absmax: dtype %d", signal_dtype);
	CONVASR_CHECK_LAUNCH("signal_absmax");
	return 0;
}

struct cpx { float re, im; };
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.im, -a.re}; }  // * (-i)

// natural-order 4- and 8-point DFTs (decimation in time, forward sign)
__device__ __forceinline__ void dft4(cpx a0, cpx a1, cpx a2, cpx a3, cpx (&o)[4]) {
	const cpx e0 = cadd(a0, a2), e1 = csub(a0, a2), o0 = cadd(a1, a3), o1 = mul_mi(csub(a1, a3));
	o[0] = cadd(e0, o0); o[1] = cadd(e1, o1); o[2] = csub(e0, o0); o[3] = csub(e1, o1);
}
__device__ __forceinline__ void dft8(cpx (&v)[8]) {
	cpx E[4], O[4];
	dft4(v[0], v[2], v[4], v[6], E);
	dft4(v[1], v[3], v[5], v[7], O);
	const float s = 0.70710678118654752440f;
	O[1] = cpx{(O[1].re + O[1].im) * s, (O[1].im - O[1].re) * s};  // * exp(-i pi/4)
	O[2] = mul_mi(O[2]);
	O[3] = cpx{(O[3].im - O[3].re) * s, -(O[3].re + O[3].im) * s};  // * exp(-3i pi/4)
#pragma unroll
	for (int k = 0; k < 4; ++k) { v[k] = cadd(E[k], O[k]); v[k + 4] = csub(E[k], O[k]); }
}

template <typename S> __device__ __forceinline__ float sig_load(const S* p, int64_t i) { return (float)p[i]; }

// one sample of the reference's padded signal (models.py:570-582) for utterance row `xs`
template <typename S>
__device__ __forceinline__ float padded_sample(const S* xs, int i, int T, int pad, int nvalid, float inv_denom, bool normalize, float preemph) {
	int t = i - pad;
	if (i < pad) t = (pad < T) ? pad - i : -1;  // reflect (edge excluded) when T > pad, constant zeros otherwise
	if (t < 0 || t >= T || t >= nvalid) return 0.f;
	float cur = sig_load(xs, t);
	if (normalize) cur = cur * inv_denom;
	if (preemph > 0.f && t > 0) {
		float prev = sig_load(xs, t - 1);
		if (normalize) prev = prev * inv_denom;
		cur = cur - preemph * prev;
	}
	return cur;
}

template <typename S>
__global__ __launch_bounds__(64 * FE_WAVES) void logmel_kernel(const S* __restrict__ signal, const float* __restrict__ absmax, const float* __restrict__ xlen,
                                                               const float* __restrict__ window, int win_length, const float* __restrict__ melw,
                                                               const float* __restrict__ melb, float* __restrict__ out, int B, int T, int F, int hop, int nmel,
                                                               float preemph, int pairs_per_b, int total_pairs) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* const melS = reinterpret_cast<float*>(smem);                  // [FE_SPAN][64] sparse mel table
	cpx* const tw = reinterpret_cast<cpx*>(melS + FE_SPAN * 64);         // [512] exp(-2 pi i m / 512)
	float* const win = reinterpret_cast<float*>(tw + FE_NFFT);           // [512] window centred in nfft
	cpx* const work = reinterpret_cast<cpx*>(win + FE_NFFT);             // [FE_WAVES][512]
	float* const pw = reinterpret_cast<float*>(work + FE_WAVES * FE_NFFT);  // [FE_WAVES][2][FE_BINS + 7]
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

	// support of this lane's mel filter: the filterbank is ~97 % zeros (each triangle spans 3..30 of the 257 bins), so only each
	// filter's support is kept in LDS (16 KiB instead of the 67 KiB dense matrix: room for 16 waves per CU) and the 257-term dot
	// product is cut to the widest support in the wave; skipping exact zeros changes no result
	int klo = FE_BINS, khi = 0;
	if (lane < nmel)
		for (int k = 0; k < FE_BINS; ++k)
			if (melw[lane * FE_BINS + k] != 0.f) { klo = min(klo, k); khi = k + 1; }
	if (khi <= klo) { klo = 0; khi = 0; }
	if (wave == 0)
		for (int j = 0; j < FE_SPAN; ++j) melS[j * 64 + lane] = (lane < nmel && klo + j < khi) ? melw[lane * FE_BINS + klo + j] : 0.f;
	for (int i = tid; i < FE_NFFT; i += blockDim.x) {
		float s, c;
		sincospif((float)i / 256.0f, &s, &c);
		tw[i] = cpx{c, -s};
		const int left = (FE_NFFT - win_length) / 2;
		win[i] = (i >= left && i < left + win_length) ? window[i - left] : 0.f;
	}
	__syncthreads();
	int span = khi - klo;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) span = max(span, __shfl_xor(span, o, 64));

	cpx* const buf = work + wave * FE_NFFT;
	float* const p0 = pw + wave * 2 * (FE_BINS + 7);
	float* const p1 = p0 + FE_BINS + 7;
	const int pad = FE_NFFT / 2;
	const float bias = lane < nmel ? melb[lane] : 1.f;

	for (int pair = blockIdx.x * FE_WAVES + wave; pair < total_pairs; pair += gridDim.x * FE_WAVES) {
		const int b = pair / pairs_per_b, f0 = (pair % pairs_per_b) * 2;
		const bool has_f1 = f0 + 1 < F;
		const S* xs = signal + (int64_t)b * T;
		const int nvalid = valid_len(xlen, b, T);
		const float denom = absmax ? 1.f / (absmax[b] + 1e-5f) : 1.f;  // reciprocal: one division per utterance instead of two per sample
		const bool normalize = absmax != nullptr;

		// ---- stage 1 (Ns = 1): no twiddles; inputs straight from global memory
		cpx v[8];
#pragma unroll
		for (int rr = 0; rr < 8; ++rr) {
			const int n = lane + 64 * rr;
			const float w = win[n];
			float a = 0.f, c = 0.f;
			if (w != 0.f) {
				a = w * padded_sample(xs, f0 * hop + n, T, pad, nvalid, denom, normalize, preemph);
				if (has_f1) c = w * padded_sample(xs, (f0 + 1) * hop + n, T, pad, nvalid, denom, normalize, preemph);
			}
			v[rr] = cpx{a, c};
		}
		dft8(v);
#pragma unroll
		for (int rr = 0; rr < 8; ++rr) buf[lane * 8 + rr] = v[rr];
		__builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes are done before its next reads
		__builtin_amdgcn_wave_barrier();
		// ---- stage 2 (Ns = 8)
		{
			const int k = lane & 7;
#pragma unroll
			for (int rr = 0; rr < 8; ++rr) v[rr] = cmul(buf[lane + 64 * rr], tw[k * rr * 8]);
			dft8(v);
			__builtin_amdgcn_wave_barrier();
			const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
			for (int rr = 0; rr < 8; ++rr) buf[j0 + rr * 8] = v[rr];
			__builtin_amdgcn_s_waitcnt(0xc07f);
			__builtin_amdgcn_wave_barrier();
		}
		// ---- stage 3 (Ns = 64)
		{
#pragma unroll
			for (int rr = 0; rr < 8; ++rr) v[rr] = cmul(buf[lane + 64 * rr], tw[lane * rr]);
			dft8(v);
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int rr = 0; rr < 8; ++rr) buf[lane + rr * 64] = v[rr];
			__builtin_amdgcn_s_waitcnt(0xc07f);
			__builtin_amdgcn_wave_barrier();
		}
		// ---- un-mix the two real spectra and take the power: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i
		for (int k = lane; k < FE_BINS; k += 64) {
			const cpx z = buf[k], zc = buf[(FE_NFFT - k) & (FE_NFFT - 1)];
			const float are = 0.5f * (z.re + zc.re), aim = 0.5f * (z.im - zc.im);
			const float bre = 0.5f * (z.im + zc.im), bim = 0.5f * (zc.re - z.re);
			p0[k] = are * are + aim * aim;
			p1[k] = bre * bre + bim * bim;
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		// ---- mel + eps bias + log: lane = mel channel
		float m0 = 0.f, m1 = 0.f;
		for (int j = 0; j < span; ++j) {
			const int k = min(klo + j, FE_BINS - 1);
			const float w = j < FE_SPAN ? melS[j * 64 + lane] : ((klo + j < khi) ? melw[lane * FE_BINS + k] : 0.f);
			m0 = fmaf(w, p0[k], m0);
			m1 = fmaf(w, p1[k], m1);
		}
		if (lane < nmel) {
			out[((int64_t)b * F + f0) * nmel + lane] = logf(m0 + bias);
			if (has_f1) out[((int64_t)b * F + f0 + 1) * nmel + lane] = logf(m1 + bias);
		}
		__builtin_amdgcn_wave_barrier();
	}
}

extern "C" int convasr_logmel_fwd(const void* signal, int signal_dtype, const float* absmax, const float* xlen, const float* window, int win_length,
                                  const float* mel_weight, const float* mel_bias, float* out, int B, int T, int nfft, int hop, int nmel, float preemphasis,
                                  void* stream) {
	CONVASR_CHECK_ARG(signal && window && mel_weight && mel_bias && out && B > 0 && T > 0 && hop > 0, "logmel_fwd: bad arguments");
	if (nfft != FE_NFFT || nmel > 64 || nmel < 1 || win_length > nfft || win_length < 1)
		return convasr_fail(CONVASR_EUNSUPPORTED, "logmel_fwd: supports nfft == 512 (window 257..512 samples), nmel <= 64; got nfft %d nmel %d win %d", nfft, nmel, win_length);
	const int F = 1 + T / hop;  // (T + 2 * pad - nfft) / hop + 1 with pad = nfft / 2
	const int pairs_per_b = (F + 1) / 2, total_pairs = B * pairs_per_b;
	const size_t smem = sizeof(float) * (FE_SPAN * 64 + 2 * FE_NFFT + FE_NFFT + 2 * FE_WAVES * FE_NFFT + FE_WAVES * 2 * (FE_BINS + 7));
	int grid = (total_pairs + FE_WAVES - 1) / FE_WAVES;
	if (grid > 256) grid = 256;  // persistent: one workgroup per CU amortises the table set-up (mel transpose, twiddles) over ~190 frame pairs
	hipStream_t s = (hipStream_t)stream;
	if (signal_dtype == CONVASR_F32) {
		auto kern = logmel_kernel<float>;
		static unsigned long long set = 0;
		convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), set);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FE_WAVES), smem, s, (const float*)signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, F, hop, nmel, preemphasis, pairs_per_b, total_pairs);
	} else if (signal_dtype == CONVASR_I16) {
		auto kern = logmel_kernel<short>;
		static unsigned long long set = 0;
		convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), set);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FE_WAVES), smem, s, (const short*)signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, F, hop, nmel, preemphasis, pairs_per_b, total_pairs);
	} else return convasr_fail(CONVASR_EUNSUPPORTED, "logmel_fwd: signal dtype %d", signal_dtype);
	CONVASR_CHECK_LAUNCH("logmel_fwd");
	return 0;
}
