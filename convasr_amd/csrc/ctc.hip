// CTC loss and gradient (reference call site: F.ctc_loss at models.py:323, blank = C-1, reduction none, zero_infinity False;
// arithmetic: the alpha-beta recursion of Graves et al. 2006 as ATen implements it).
//
// Kernel 1 -- one workgroup per utterance, wave 0 runs the alpha sweep forward in time while wave 1 runs the beta sweep
// backward, concurrently.  A lane owns NS consecutive states of the extended target (blank, y1, blank, y2, ...), so the
// s-1 / s-2 neighbours are registers except for two DPP wave shifts per step; log-sum-exp in fp32 in BASE 2 (v_exp_f32 and
// v_log_f32 are base-2 instructions: log-probs are scaled by log2(e) once while they are staged, the lattices hold log2
// values); the next frame's log-probs are fetched while the current one is being combined.  The lattices go to an L2/MALL-
// resident workspace laid out [t][i][lane] (state s = lane * NS + i) so that every store / load instruction is one 256-byte row.
// Every CTC_RENORM steps the sweep subtracts floor(max over states) from its column: an INTEGER in the log2 domain, so the
// subtraction and the running sum of the offsets (kept per frame beside the lattice) are exact in fp32.  Column values stay
// O(10) near the likely states instead of growing to ~3 T, and the fp32 lattice keeps ~1e-5 precision in the gradient at
// T = 753, where an unnormalised fp32 lattice (ATen's CPU/CUDA kernels) is at ~1e-3 (scratch/ctc_prec.py vs float64).
// Kernel 2 -- one wave per frame: posterior[c] = sum_{s: l'_s = c} exp2(alpha + beta - total - lp) accumulated in LDS bins,
// grad = exp(lp) - posterior for t < olen, 0 beyond.
#include "common.h"

#define CTC_NEG (-INFINITY)
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453
#define CTC_RENORM 8

// base-2 log-sum-exp of three values that may be -inf
__device__ __forceinline__ float lse3(float a, float b, float c) {
	const float m = fmaxf(a, fmaxf(b, c));
	const float r = m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
	return m == CTC_NEG ? CTC_NEG : r;
}

// lane i <- lane i - 1 (lane 0 <- fill) / lane i <- lane i + 1 (lane 63 <- fill): one DPP move, no LDS round trip
__device__ __forceinline__ float wave_shr1(float v, float fill) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// max over the 64 lanes by DPP moves (two quad permutes, the two row mirrors, the two row broadcasts) and one readlane: no LDS round
// trip, where the ds_bpermute butterfly of wave_max() costs six of them in the middle of a latency-bound recurrence
__device__ __forceinline__ float wave_max_dpp(float v) {
#define CTC_DPP_MAX(ctrl, rows) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, rows, 0xf, false)))
	CTC_DPP_MAX(0xB1, 0xf);   // quad_perm [1,0,3,2]
	CTC_DPP_MAX(0x4E, 0xf);   // quad_perm [2,3,0,1]
	CTC_DPP_MAX(0x141, 0xf);  // row_half_mirror
	CTC_DPP_MAX(0x140, 0xf);  // row_mirror: every lane holds its row's max
	CTC_DPP_MAX(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
	CTC_DPP_MAX(0x143, 0xc);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's max
#undef CTC_DPP_MAX
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// LP_LDS: the utterance's whole (T, C) log-prob slab is first copied into LDS (114 KB at T = 753, C = 38) so that the T-step
// recurrences read their per-frame class scores at LDS latency instead of L2 latency (the sweeps are latency-bound).
// offs: [2][B][T] integer-valued offsets (true log2 alpha(t, s) = lattice value + offs[0][b][t]; beta: offs[1]);
// tot: [B][2] = {integer part, remainder} of the utterance's log2 likelihood.
template <int NS, bool LP_LDS>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ olen,
                                                             const int64_t* __restrict__ ylen, float* __restrict__ nll, float* __restrict__ alpha,
                                                             float* __restrict__ beta, float* __restrict__ offs, float* __restrict__ tot, int B, int T, int C, int S_max, int blank) {
	extern __shared__ __attribute__((aligned(16))) float ctc_smem[];
	float* const fin = ctc_smem;            // [64 * NS]
	float* const lpl = ctc_smem + 64 * NS;  // [T * C] when LP_LDS
	const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr int LP = 64 * NS;
	const int Tb = (int)olen[b], S = (int)ylen[b], L = 2 * S + 1;
	const float* lpg = lp + (int64_t)b * T * C;
	if (LP_LDS) {
		const int n = (Tb > 0 && Tb <= T ? Tb : 0) * C;
		for (int i = threadIdx.x; i < n; i += blockDim.x) lpl[i] = lpg[i] * CTC_LOG2E;
		__syncthreads();
	}
	if (wave >= 2) return;
	const float* lpb = LP_LDS ? lpl : lpg;
	const float lps = LP_LDS ? 1.f : CTC_LOG2E;  // log-probs not staged in LDS are scaled as they are read
	const int64_t* tg = targets + (int64_t)b * S_max;
	float* const lat = (wave == 0 ? alpha : beta) + (int64_t)b * T * LP;
	float* const off = offs + ((int64_t)wave * B + b) * T;

	int cls[NS];
	bool skip[NS], valid[NS];
#pragma unroll
	for (int i = 0; i < NS; ++i) {
		const int s = lane * NS + i;
		valid[i] = s < L;
		const bool nb = (s & 1) && valid[i];
		cls[i] = nb ? (int)tg[s >> 1] : blank;
		if (wave == 0) skip[i] = nb && s >= 3 && tg[s >> 1] != tg[(s >> 1) - 1];          // alpha: s-2 allowed
		else skip[i] = nb && s + 2 < L && tg[s >> 1] != tg[(s >> 1) + 1];                  // beta: s+2 allowed
	}
	if (Tb <= 0 || Tb > T) {
		if (threadIdx.x == 0) nll[b] = (Tb == 0 && S == 0) ? 0.f : INFINITY;
		return;
	}

	float a[NS], cur[NS], nxt[NS];
	// The per-frame score of a state beyond the extended target is -inf, so the state update below needs no validity test: every
	// state runs the same straight-line code and the NS independent log-sum-exp chains of a lane interleave (a test per state
	// compiles to NS exec-masked blocks executed one after the other, each a serial chain of dependent transcendental ops).
	auto score = [&](float v, int i) { return valid[i] ? v * lps : CTC_NEG; };
	float cum = 0.f;  // integer-valued: sum of the offsets subtracted so far
	auto renorm = [&](float (&v)[NS]) {
		float m = v[0];
#pragma unroll
		for (int i = 1; i < NS; ++i) m = fmaxf(m, v[i]);
		m = wave_max_dpp(m);
		if (m > CTC_NEG) {
			const float k = floorf(m);
#pragma unroll
			for (int i = 0; i < NS; ++i) v[i] -= k;
			cum += k;
		}
	};
	if (wave == 0) {
#pragma unroll
		for (int i = 0; i < NS; ++i) {
			const int s = lane * NS + i;
			a[i] = (s < 2 && valid[i]) ? lpb[cls[i]] * lps : CTC_NEG;
			lat[i * 64 + lane] = a[i];
		}
		if (lane == 0) off[0] = 0.f;
		if (Tb > 1) {
#pragma unroll
			for (int i = 0; i < NS; ++i) cur[i] = score(lpb[(int64_t)1 * C + cls[i]], i);
		}
		for (int t = 1; t < Tb; ++t) {
			if (t + 1 < Tb) {
#pragma unroll
				for (int i = 0; i < NS; ++i) nxt[i] = lpb[(int64_t)(t + 1) * C + cls[i]];
			}
			const float p1 = wave_shr1(a[NS - 1], CTC_NEG);
			const float p2 = NS >= 2 ? wave_shr1(a[NS >= 2 ? NS - 2 : 0], CTC_NEG) : wave_shr1(p1, CTC_NEG);
			float n[NS];
#pragma unroll
			for (int i = NS - 1; i >= 0; --i) {
				const float m1 = i >= 1 ? a[i - 1] : p1;
				const float m2 = i >= 2 ? a[i - 2] : (i == 1 ? p1 : p2);
				n[i] = lse3(a[i], m1, skip[i] ? m2 : CTC_NEG) + cur[i];
			}
			if ((t & (CTC_RENORM - 1)) == 0) renorm(n);
#pragma unroll
			for (int i = 0; i < NS; ++i) { a[i] = n[i]; lat[(int64_t)t * LP + i * 64 + lane] = n[i]; cur[i] = score(nxt[i], i); }  // the wait for the read-ahead is here, after the step's arithmetic
			if (lane == 0) off[t] = cum;
		}
#pragma unroll
		for (int i = 0; i < NS; ++i) fin[lane * NS + i] = a[i];
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) {
			const float l1 = fin[L - 1], l2 = L >= 2 ? fin[L - 2] : CTC_NEG;
			const float m = fmaxf(l1, l2);
			const float rem = m + log2f(exp2f(l1 - m) + exp2f(l2 - m));
			nll[b] = m == CTC_NEG ? INFINITY : (float)(-CTC_LN2 * ((double)cum + (double)rem));
			tot[2 * b] = cum;
			tot[2 * b + 1] = rem;
		}
	} else {
		const float* lrow = lpb + (int64_t)(Tb - 1) * C;
#pragma unroll
		for (int i = 0; i < NS; ++i) {
			const int s = lane * NS + i;
			a[i] = (valid[i] && s >= L - 2) ? lrow[cls[i]] * lps : CTC_NEG;
			lat[(int64_t)(Tb - 1) * LP + i * 64 + lane] = a[i];
		}
		if (lane == 0) off[Tb - 1] = 0.f;
		if (Tb > 1) {
#pragma unroll
			for (int i = 0; i < NS; ++i) cur[i] = score(lpb[(int64_t)(Tb - 2) * C + cls[i]], i);
		}
		for (int t = Tb - 2; t >= 0; --t) {
			if (t > 0) {
#pragma unroll
				for (int i = 0; i < NS; ++i) nxt[i] = lpb[(int64_t)(t - 1) * C + cls[i]];
			}
			const float p1 = wave_shl1(a[0], CTC_NEG);
			const float p2 = NS >= 2 ? wave_shl1(a[NS >= 2 ? 1 : 0], CTC_NEG) : wave_shl1(p1, CTC_NEG);
			float n[NS];
#pragma unroll
			for (int i = 0; i < NS; ++i) {
				const float m1 = i + 1 < NS ? a[i + 1 < NS ? i + 1 : 0] : p1;
				const float m2 = i + 2 < NS ? a[i + 2 < NS ? i + 2 : 0] : (i + 2 == NS ? p1 : p2);
				n[i] = lse3(a[i], m1, skip[i] ? m2 : CTC_NEG) + cur[i];
			}
			if ((t & (CTC_RENORM - 1)) == 0) renorm(n);
#pragma unroll
			for (int i = 0; i < NS; ++i) { a[i] = n[i]; lat[(int64_t)t * LP + i * 64 + lane] = n[i]; cur[i] = score(nxt[i], i); }  // the wait for the read-ahead is here, after the step's arithmetic
			if (lane == 0) off[t] = cum;
		}
	}
}

template <int NS>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ olen,
                                                       const int64_t* __restrict__ ylen, const float* __restrict__ nll, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const float* __restrict__ offs, const float* __restrict__ tot, float* __restrict__ grad,
                                                       int B, int T, int C, int S_max, int blank, int t_per_block) {
	extern __shared__ float bins[];  // [4][C]
	const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr int LP = 64 * NS;
	const int Tb = (int)olen[b], S = (int)ylen[b], L = 2 * S + 1;
	const bool feasible = nll[b] < INFINITY;
	const int64_t* tg = targets + (int64_t)b * S_max;
	int cls[NS];
#pragma unroll
	for (int i = 0; i < NS; ++i) {
		const int s = lane * NS + i;
		cls[i] = ((s & 1) && s < L) ? (int)tg[s >> 1] : blank;
	}
	float* mybins = bins + wave * C;
	const int t_begin = blockIdx.x * t_per_block, t_end = min(T, t_begin + t_per_block);
	for (int t = t_begin + wave; t < t_end; t += 4) {
		const float* row = lp + ((int64_t)b * T + t) * C;
		float* grow = grad + ((int64_t)b * T + t) * C;
		if (t >= Tb || !feasible) {
			for (int c = lane; c < C; c += 64) grow[c] = 0.f;
			continue;
		}
		for (int c = lane; c < C; c += 64) mybins[c] = 0.f;
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		const float* ar = alpha + ((int64_t)b * T + t) * LP + lane;
		const float* br = beta + ((int64_t)b * T + t) * LP + lane;
		// alpha + beta - total = (lattice values - remainder of total) + (offsets - integer part of total): the second group is exact
		const float shift = (offs[(int64_t)b * T + t] + offs[((int64_t)B + b) * T + t] - tot[2 * b]) - tot[2 * b + 1];
		float blank_sum = 0.f;
#pragma unroll
		for (int i = 0; i < NS; ++i) {
			const int s = lane * NS + i;
			if (s < L) {
				const float v = __builtin_amdgcn_exp2f((ar[i * 64] + br[i * 64]) + (shift - row[cls[i]] * CTC_LOG2E));
				if (s & 1) atomicAdd(mybins + cls[i], v);
				else blank_sum += v;
			}
		}
		blank_sum = wave_sum(blank_sum);
		if (lane == 0) atomicAdd(mybins + blank, blank_sum);
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		for (int c = lane; c < C; c += 64) grow[c] = __expf(row[c]) - mybins[c];
		__builtin_amdgcn_wave_barrier();
	}
}

static int ctc_ns(int S_max) {
	const int L = 2 * S_max + 1;
	const int need = (L + 63) / 64;
	const int opts[] = {1, 2, 3, 4, 5, 6, 8, 12, 16};
	for (int o : opts) if (o >= need) return o;
	return -1;
}

extern "C" int64_t convasr_ctc_workspace_bytes(int B, int T, int S_max) {
	const int ns = ctc_ns(S_max);
	if (ns < 0) return -1;
	return (2 * (int64_t)B * T * 64 * ns + 2 * (int64_t)B * T + 2 * (int64_t)B) * (int64_t)sizeof(float);  // lattices, offsets, totals
}

extern "C" int convasr_ctc_loss(const float* log_probs, const int64_t* targets, const int64_t* olen, const int64_t* ylen, float* nll, float* grad,
                                void* workspace, int B, int T, int C, int S_max, int blank, void* stream) {
	CONVASR_CHECK_ARG(log_probs && targets && olen && ylen && nll && workspace && B > 0 && T > 0 && C > 1 && S_max >= 0 && blank >= 0 && blank < C, "ctc_loss: bad arguments");
	const int ns = ctc_ns(S_max);
	if (ns < 0) return convasr_fail(CONVASR_EUNSUPPORTED, "ctc_loss: target length %d > 511", S_max);
	CONVASR_CHECK_ARG(C <= 8192, "ctc_loss: C %d > 8192", C);
	hipStream_t s = (hipStream_t)stream;
	float* alpha = (float*)workspace;
	float* beta = alpha + (int64_t)B * T * 64 * ns;
	float* offs = beta + (int64_t)B * T * 64 * ns;
	float* tot = offs + 2 * (int64_t)B * T;
	const int t_per_block = 32;
	dim3 ggrid((T + t_per_block - 1) / t_per_block, B);
	const size_t gsmem = 4 * (size_t)C * sizeof(float);
	const size_t lds_lp = (size_t)T * C * sizeof(float);
	const bool in_lds = lds_lp + 64 * 16 * sizeof(float) <= 159 * 1024;  // T = 1001 (20 s) x C = 38 still fits the 160 KiB of a CU
#define CTC_CASE(NS) case NS: \
		if (in_lds) { \
			auto kern = ctc_alpha_beta_kernel<NS, true>; \
			static bool set = false; \
			if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
			hipLaunchKernelGGL(kern, dim3(B), dim3(256), 64 * NS * sizeof(float) + lds_lp, s, log_probs, targets, olen, ylen, nll, alpha, beta, offs, tot, B, T, C, S_max, blank); \
		} else hipLaunchKernelGGL((ctc_alpha_beta_kernel<NS, false>), dim3(B), dim3(128), 64 * NS * sizeof(float), s, log_probs, targets, olen, ylen, nll, alpha, beta, offs, tot, B, T, C, S_max, blank); \
		if (grad) hipLaunchKernelGGL((ctc_grad_kernel<NS>), ggrid, dim3(256), gsmem, s, log_probs, targets, olen, ylen, nll, alpha, beta, offs, tot, grad, B, T, C, S_max, blank, t_per_block); \
		break;
	switch (ns) {
		CTC_CASE(1) CTC_CASE(2) CTC_CASE(3) CTC_CASE(4) CTC_CASE(5) CTC_CASE(6) CTC_CASE(8) CTC_CASE(12) CTC_CASE(16)
	}
#undef CTC_CASE
	CONVASR_CHECK_LAUNCH("ctc_loss");
	return 0;
}
