// Log-mel frontend (reference: LogFilterBankFrontend.forward, models.py:565-597; normalize_signal, models.py:684-686).
//
// One pass over the waveform: every wave turns TWO frames into one nfft-point complex FFT in LDS (frame A real, frame B imaginary; 512 points:
// Stockham radix-8 x 3, other sizes radix-2), un-mixes the two real spectra, takes |.|^2, applies the (nmel x (nfft / 2 + 1)) mel matrix whose
// non-zero spans are held transposed in LDS, adds the eps bias, takes the log and writes channels-last (B, F, nmel) fp32 rows.  The normalise /
// pre-emphasis / mask / reflect-left / zero-right padding of the reference are index arithmetic on the load -- the padded signal and the
// 197 MB complex spectrogram (64 x 15 s) are never materialised.  4 B in per sample, 4 nmel B out per frame; VALU-bound.
#include "common.h"

#define FE_SPAN 64  // LDS rows of the sparse mel table: melS[j][mel] = weight of bin klo[mel] + j (wider filters read the rest from global memory)
// nfft = 2^LOG2N for LOG2N = 7..10 (the reference takes the power of two above the window, models.py:516: 512 for 16 kHz x 0.02 s, 256 for train.py's
// default 8 kHz x 0.02 s, 1024 for 44.1 kHz).  512 has its own radix-8 x 3 schedule; the other sizes share a radix-2 Stockham loop.
template <int LOG2N, int NM = 1> struct FeCfg {  // NM: mel channels per lane (1: up to 64 channels, 2: up to 128)
	static constexpr int N = 1 << LOG2N, BINS = N / 2 + 1, RPL = N / 64;       // samples per lane
	static constexpr int WAVES = LOG2N <= 9 ? 16 : 8;                           // per workgroup: what 160 KiB of LDS holds
	static constexpr int BUF = LOG2N == 9 ? 576 : N;                            // a wave's FFT buffer, complex words (512: padded layouts, below)
	static constexpr int PQ = 2 * (BINS + 7);                                   // a wave's power buffer, floats
	static constexpr int TWX = LOG2N == 9 ? 64 + 8 * 64 : 0;                    // the radix-8 schedule's transposed twiddle tables
	static constexpr size_t smem = sizeof(float) * (NM * FE_SPAN * 64 + 2 * N + N + 2 * TWX + 2 * WAVES * BUF + WAVES * PQ + NM * 2 * 64);
};
// nfft = 512: a wave's FFT buffer is written and read in three index patterns; two padded layouts keep every ds_read / ds_write_b64 of a half-wave on
// 32 different bank pairs (the unpadded buffer had 8-way conflicts on the stage-1 and stage-2 writes and up to 8-way on the strided twiddle reads):
// generation 1 (written lane * 8 + r, read lane + 64 r) one pad element per 32, generation 2 (written g * 64 + k + 8 r, read lane + 64 r) eight per 64.
#define FE_P1(i) ((i) + ((i) >> 5))
#define FE_P2(i) ((i) + (((i) >> 6) << 3))

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

__global__ void zero_words_kernel(unsigned* __restrict__ p, int n) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) p[i] = 0u;
}

extern "C" int convasr_signal_absmax(const void* signal, int signal_dtype, int B, int T, float* absmax, void* stream) {
	CONVASR_CHECK_ARG(signal && absmax && B > 0 && T > 0, "signal_absmax: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(zero_words_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (unsigned*)absmax, B);  // (a kernel, not hipMemsetAsync: no memset node in a captured step, see convasr_copy)
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

// one sample of the reference's padded signal (models.py:570-582) for utterance row `xs`.  Branch-free: both loads are issued for every lane
// (an out-of-range sample reads xs[0] and is replaced by zero afterwards), so the 24 loads of a frame pair are in flight together instead of
// one dependent round trip per divergent branch.
template <typename S>
__device__ __forceinline__ float padded_sample(const S* xs, int i, int T, int pad, int nvalid, float inv_denom, bool normalize, float preemph) {
	int t = i - pad;
	if (i < pad) t = (pad < T) ? pad - i : -1;  // reflect (edge excluded) when T > pad, constant zeros otherwise
	const bool ok = t >= 0 && t < T && t < nvalid;
	const int tc = ok ? t : 0, tp = tc > 0 ? tc - 1 : 0;
	float cur = sig_load(xs, tc), prev = sig_load(xs, tp);
	if (normalize) { cur = cur * inv_denom; prev = prev * inv_denom; }
	if (preemph > 0.f) cur = tc > 0 ? cur - preemph * prev : cur;
	return ok ? cur : 0.f;
}

template <typename S, int LOG2N, int NM>
__global__ __launch_bounds__(64 * FeCfg<LOG2N>::WAVES) void logmel_kernel(const S* __restrict__ signal, const float* __restrict__ absmax, const float* __restrict__ xlen,
                                                                          const float* __restrict__ window, int win_length, const float* __restrict__ melw,
                                                                          const float* __restrict__ melb, float* __restrict__ out, int B, int T, int F, int hop, int nmel,
                                                                          float preemph, int pairs_per_b, int total_pairs) {
	using Cfg = FeCfg<LOG2N, NM>;
	constexpr int N = Cfg::N, BINS = Cfg::BINS, RPL = Cfg::RPL, WAVES = Cfg::WAVES;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* const melS = reinterpret_cast<float*>(smem);                  // [NM][FE_SPAN][64] sparse mel table: channel lane + 64 q in plane q
	cpx* const tw = reinterpret_cast<cpx*>(melS + NM * FE_SPAN * 64);    // [N] exp(-2 pi i m / N)
	float* const win = reinterpret_cast<float*>(tw + N);                 // [N] window centred in nfft
	cpx* const tw2 = reinterpret_cast<cpx*>(win + N);                    // 512 only: [8][8]  stage 2: exp(-2 pi i 8 k r / 512) at [r][k]
	cpx* const tw3 = tw2 + 64;                                           // 512 only: [8][64] stage 3: exp(-2 pi i lane r / 512) at [r][lane]
	cpx* const work = tw2 + Cfg::TWX;                                    // [WAVES][BUF]
	float* const pw = reinterpret_cast<float*>(work + WAVES * Cfg::BUF); // [WAVES][PQ]
	int* const supp = reinterpret_cast<int*>(pw + WAVES * Cfg::PQ);      // [NM][2][64] first / one-past-last non-zero bin of every mel filter
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

	// support of this lane's mel filter: the filterbank is ~97 % zeros (each triangle spans 3..30 of the 257 bins), so only each
	// filter's support is kept in LDS (16 KiB instead of the 67 KiB dense matrix: room for 16 waves per CU) and the 257-term dot
	// product is cut to the widest support in the wave; skipping exact zeros changes no result
	if (tid < 64 * NM) { supp[(tid >> 6) * 128 + (tid & 63)] = BINS; supp[(tid >> 6) * 128 + 64 + (tid & 63)] = 0; }
	__syncthreads();
	int klo[NM], khi[NM], span[NM];
	float bias[NM];
#pragma unroll
	for (int q = 0; q < NM; ++q) {
		const int m = lane + 64 * q;
		if (m < nmel) {  // every wave scans its slice of the bins for all filters
			int lo = BINS, hi = 0;
			for (int k = wave; k < BINS; k += WAVES)
				if (melw[m * BINS + k] != 0.f) { lo = min(lo, k); hi = k + 1; }
			if (hi > 0) { atomicMin(supp + q * 128 + lane, lo); atomicMax(supp + q * 128 + 64 + lane, hi); }
		}
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NM; ++q) {
		const int m = lane + 64 * q;
		klo[q] = supp[q * 128 + lane]; khi[q] = supp[q * 128 + 64 + lane];
		if (khi[q] <= klo[q]) { klo[q] = 0; khi[q] = 0; }
		for (int j = wave; j < FE_SPAN; j += WAVES) melS[(q * FE_SPAN + j) * 64 + lane] = (m < nmel && klo[q] + j < khi[q]) ? melw[m * BINS + klo[q] + j] : 0.f;
		bias[q] = m < nmel ? melb[m] : 1.f;
		span[q] = khi[q] - klo[q];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) span[q] = max(span[q], __shfl_xor(span[q], o, 64));
	}
	for (int i = tid; i < N; i += blockDim.x) {
		float s, c;
		sincospif((float)i / (float)(N / 2), &s, &c);
		tw[i] = cpx{c, -s};
		const int left = (N - win_length) / 2;
		win[i] = (i >= left && i < left + win_length) ? window[i - left] : 0.f;
	}
	__syncthreads();
	if constexpr (LOG2N == 9) {
		if (tid < 64) tw2[tid] = tw[(tid & 7) * (tid >> 3) * 8];
		if (tid < 512) tw3[tid] = tw[(tid & 63) * (tid >> 6)];
		__syncthreads();
	}

	cpx* const buf = work + wave * Cfg::BUF;
	float2* const pq = reinterpret_cast<float2*>(pw + wave * Cfg::PQ);  // power of bin k of the pair's two frames
	const int pad = N / 2;

	for (int pair = blockIdx.x * WAVES + wave; pair < total_pairs; pair += gridDim.x * WAVES) {
		const int b = pair / pairs_per_b, f0 = (pair % pairs_per_b) * 2;
		const bool has_f1 = f0 + 1 < F;
		const S* xs = signal + (int64_t)b * T;
		const int nvalid = valid_len(xlen, b, T);
		const float denom = absmax ? 1.f / (absmax[b] + 1e-5f) : 1.f;  // reciprocal: one division per utterance instead of two per sample
		const bool normalize = absmax != nullptr;

		const int t0 = f0 * hop - pad;  // source index of the pair's first sample
		if (t0 >= nvalid && t0 >= 0) {  // both frames lie in the masked tail: a zero spectrum, log(eps) exactly as the full computation gives it
#pragma unroll
			for (int q = 0; q < NM; ++q)
				if (lane + 64 * q < nmel) {
					const float z = logf(fmaf(0.f, 0.f, 0.f) + bias[q]);
					out[((int64_t)b * F + f0) * nmel + lane + 64 * q] = z;
					if (has_f1) out[((int64_t)b * F + f0 + 1) * nmel + lane + 64 * q] = z;
				}
			continue;
		}
		// ---- the pair's windowed samples: v[r] = (frame A, frame B) at n = lane + 64 r
		cpx v[RPL];
		if (has_f1 && preemph > 0.f && t0 >= 1 && t0 + hop + N <= min(T, nvalid)) {
			// interior pair (all but the first two and the last few of an utterance): every sample is a plain one with a predecessor, no reflection,
			// no range or mask test -- 4 RPL loads at immediate offsets from one address, no per-sample index arithmetic (the general form below spends
			// ~25 integer / select instructions per sample: the kernel is VALU-bound and that was 45 % of its instructions).  Same arithmetic.
			const S* const xp = xs + t0 + lane;
#pragma unroll
			for (int rr = 0; rr < RPL; ++rr) {
				const float w = win[lane + 64 * rr];
				float ca = sig_load(xp, 64 * rr), pa = sig_load(xp, 64 * rr - 1), cb = sig_load(xp + hop, 64 * rr), pb = sig_load(xp + hop, 64 * rr - 1);
				if (normalize) { ca = ca * denom; pa = pa * denom; cb = cb * denom; pb = pb * denom; }
				ca = ca - preemph * pa;
				cb = cb - preemph * pb;
				v[rr] = cpx{w * ca, w * cb};  // outside the window's support w is +0 and the product +-0: the sign of a zero never reaches |.|^2
			}
		} else {
#pragma unroll
			for (int rr = 0; rr < RPL; ++rr) {
				const int n = lane + 64 * rr;
				const float w = win[n];
				const float sa = padded_sample(xs, f0 * hop + n, T, pad, nvalid, denom, normalize, preemph);
				const float sc = padded_sample(xs, (has_f1 ? f0 + 1 : f0) * hop + n, T, pad, nvalid, denom, normalize, preemph);
				v[rr] = cpx{w != 0.f ? w * sa : 0.f, (w != 0.f && has_f1) ? w * sc : 0.f};
			}
		}
		if constexpr (LOG2N == 9) {
			// ---- stage 1 (Ns = 1): no twiddles; inputs straight from global memory
			dft8(v);
#pragma unroll
			for (int rr = 0; rr < 8; ++rr) buf[FE_P1(lane * 8 + rr)] = v[rr];
			__builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes are done before its next reads
			__builtin_amdgcn_wave_barrier();
			// ---- stage 2 (Ns = 8)
			{
				const int k = lane & 7;
#pragma unroll
				for (int rr = 0; rr < 8; ++rr) v[rr] = cmul(buf[FE_P1(lane + 64 * rr)], tw2[rr * 8 + k]);
				dft8(v);
				__builtin_amdgcn_wave_barrier();
				const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
				for (int rr = 0; rr < 8; ++rr) buf[FE_P2(j0 + rr * 8)] = v[rr];
				__builtin_amdgcn_s_waitcnt(0xc07f);
				__builtin_amdgcn_wave_barrier();
			}
			// ---- stage 3 (Ns = 64)
			{
#pragma unroll
				for (int rr = 0; rr < 8; ++rr) v[rr] = cmul(buf[FE_P2(lane + 64 * rr)], tw3[rr * 64 + lane]);
				dft8(v);
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int rr = 0; rr < 8; ++rr) buf[lane + rr * 64] = v[rr];
				__builtin_amdgcn_s_waitcnt(0xc07f);
				__builtin_amdgcn_wave_barrier();
			}
		} else {
			// ---- radix-2 Stockham, LOG2N stages in place: a stage's N / 2 butterflies (j, j + N / 2) -> (j0, j0 + Ns) are all read into registers
			// before any is written (the wave owns the buffer), the first stage's inputs are the registers of the load above; natural order out
			constexpr int BPL = N / 128;  // butterflies per lane and stage
			cpx lo[BPL], hi[BPL];
#pragma unroll
			for (int q = 0; q < BPL; ++q) { lo[q] = v[q]; hi[q] = v[q + BPL]; }  // j = lane + 64 q: n = j and n = j + N / 2 = lane + 64 (q + BPL)
#pragma unroll
			for (int st = 0; st < LOG2N; ++st) {
				const int Ns = 1 << st;
				if (st > 0) {
#pragma unroll
					for (int q = 0; q < BPL; ++q) { const int j = lane + 64 * q; lo[q] = buf[j]; hi[q] = buf[j + N / 2]; }
					__builtin_amdgcn_wave_barrier();
				}
#pragma unroll
				for (int q = 0; q < BPL; ++q) {
					const int j = lane + 64 * q, k = j & (Ns - 1);
					const cpx t = st > 0 ? cmul(hi[q], tw[k << (LOG2N - 1 - st)]) : hi[q];  // exp(-2 pi i k / (2 Ns))
					const int j0 = ((j - k) << 1) + k;
					buf[j0] = cadd(lo[q], t);
					buf[j0 + Ns] = csub(lo[q], t);
				}
				__builtin_amdgcn_s_waitcnt(0xc07f);
				__builtin_amdgcn_wave_barrier();
			}
		}
		// ---- un-mix the two real spectra and take the power: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i
		for (int k = lane; k < BINS; k += 64) {
			const cpx z = buf[k], zc = buf[(N - k) & (N - 1)];
			const float are = 0.5f * (z.re + zc.re), aim = 0.5f * (z.im - zc.im);
			const float bre = 0.5f * (z.im + zc.im), bim = 0.5f * (zc.re - z.re);
			pq[k] = float2{are * are + aim * aim, bre * bre + bim * bim};
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		// ---- mel + eps bias + log: lane = mel channel (+ 64 q)
#pragma unroll
		for (int q = 0; q < NM; ++q) {
			float m0 = 0.f, m1 = 0.f;
			const int s_lds = min(span[q], FE_SPAN);
			const float* const tab = melS + q * FE_SPAN * 64;
#pragma unroll 4
			for (int j = 0; j < s_lds; ++j) {  // rows past a filter's own support hold zeros
				const int k = min(klo[q] + j, BINS - 1);
				const float w = tab[j * 64 + lane];
				const float2 pp = pq[k];
				m0 = fmaf(w, pp.x, m0);
				m1 = fmaf(w, pp.y, m1);
			}
			for (int j = FE_SPAN; j < span[q]; ++j) {  // filters wider than the table (not with the reference's 64 mels over 257 bins)
				const int k = min(klo[q] + j, BINS - 1);
				const float w = (klo[q] + j < khi[q]) ? melw[(lane + 64 * q) * BINS + k] : 0.f;
				const float2 pp = pq[k];
				m0 = fmaf(w, pp.x, m0);
				m1 = fmaf(w, pp.y, m1);
			}
			if (lane + 64 * q < nmel) {
				out[((int64_t)b * F + f0) * nmel + lane + 64 * q] = logf(m0 + bias[q]);
				if (has_f1) out[((int64_t)b * F + f0 + 1) * nmel + lane + 64 * q] = logf(m1 + bias[q]);
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
}

template <typename S, int LOG2N, int NM>
static void launch_logmel(const void* signal, const float* absmax, const float* xlen, const float* window, int win_length, const float* mel_weight, const float* mel_bias,
                          float* out, int B, int T, int hop, int nmel, float preemphasis, hipStream_t s) {
	using Cfg = FeCfg<LOG2N, NM>;
	const int F = 1 + T / hop;  // (T + 2 * pad - nfft) / hop + 1 with pad = nfft / 2
	const int pairs_per_b = (F + 1) / 2, total_pairs = B * pairs_per_b;
	int grid = (total_pairs + Cfg::WAVES - 1) / Cfg::WAVES;
	if (grid > 256) grid = 256;  // persistent: one workgroup per CU amortises the table set-up (mel transpose, twiddles) over ~190 frame pairs
	auto kern = logmel_kernel<S, LOG2N, NM>;
	static unsigned long long set = 0;
	convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), set);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * Cfg::WAVES), Cfg::smem, s, (const S*)signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, F, hop, nmel, preemphasis, pairs_per_b, total_pairs);
}

extern "C" int convasr_logmel_fwd(const void* signal, int signal_dtype, const float* absmax, const float* xlen, const float* window, int win_length,
                                  const float* mel_weight, const float* mel_bias, float* out, int B, int T, int nfft, int hop, int nmel, float preemphasis,
                                  void* stream) {
	CONVASR_CHECK_ARG(signal && window && mel_weight && mel_bias && out && B > 0 && T > 0 && hop > 0, "logmel_fwd: bad arguments");
	if ((nfft != 128 && nfft != 256 && nfft != 512 && nfft != 1024) || nmel > 128 || nmel < 1 || win_length > nfft || win_length < 1)
		return convasr_fail(CONVASR_EUNSUPPORTED, "logmel_fwd: supports nfft 128 / 256 / 512 / 1024 (window <= nfft samples), nmel <= 128; got nfft %d nmel %d win %d", nfft, nmel, win_length);
	if (signal_dtype != CONVASR_F32 && signal_dtype != CONVASR_I16) return convasr_fail(CONVASR_EUNSUPPORTED, "logmel_fwd: signal dtype %d", signal_dtype);
	hipStream_t s = (hipStream_t)stream;
#define FE_LAUNCH(L) \
	do { \
		if (signal_dtype == CONVASR_F32 && nmel <= 64) launch_logmel<float, L, 1>(signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, hop, nmel, preemphasis, s); \
		else if (signal_dtype == CONVASR_F32) launch_logmel<float, L, 2>(signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, hop, nmel, preemphasis, s); \
		else if (nmel <= 64) launch_logmel<short, L, 1>(signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, hop, nmel, preemphasis, s); \
		else launch_logmel<short, L, 2>(signal, absmax, xlen, window, win_length, mel_weight, mel_bias, out, B, T, hop, nmel, preemphasis, s); \
	} while (0)
	switch (nfft) {
		case 128: FE_LAUNCH(7); break;
		case 256: FE_LAUNCH(8); break;
		case 512: FE_LAUNCH(9); break;
		default: FE_LAUNCH(10); break;
	}
#undef FE_LAUNCH
	CONVASR_CHECK_LAUNCH("logmel_fwd");
	return 0;
}
