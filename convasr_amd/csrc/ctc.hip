// CTC loss and gradient (reference call site: F.ctc_loss at models.py:323, blank = C-1, reduction none, zero_infinity False;
// arithmetic: the alpha-beta recursion of Graves et al. 2006 as ATen implements it).
//
// Kernel 1 -- one workgroup of four waves per utterance, one per SIMD: waves 0, 1 run the alpha sweep forward in time while waves
// 2, 3 run the beta sweep backward, concurrently.  The sweeps are bound by the ISSUE of their dependent chain (one wave per sweep with
// six states per lane: 15 quarter-rate v_exp / v_log + ~110 other instructions per frame, 0.28 ms at T = 753, two SIMDs idle), so each
// sweep is split over the two waves of a barrier-free SYSTOLIC pipeline: a lane owns NP consecutive (blank, label) pairs of the
// extended target (the wave that produces the edge NPH of them, the other one NPL: see ctc_sweep), the pair at the waves' common edge
// crosses through one LDS slot PER FRAME (pre-filled with a NaN bit pattern no published word can take; the consumer polls the slot
// of the frame it needs, the producer never waits, so nothing can deadlock), and the consumer trails its producer by about a frame.
// (Measured alternatives: six waves with a workgroup barrier per frame 0.40 ms, six polled waves 0.43 ms -- waves that share a SIMD
// share its issue slots.)  The parity of a state is the register it lives in, so a blank's update is a two-term log-sum-exp; the
// s-1 / s-2 neighbours are registers and one DPP wave shift (two for beta).  "Impossible" is the finite sentinel CTC_NEG = -1e30, not
// -inf: max + log2(sum of exp2) then needs no guard against inf - inf, the sentinel absorbs every finite score added to it and
// exp2(sentinel - anything real) is 0; a log-prob of -inf is staged as the sentinel.
// log-sum-exp in fp32 in BASE 2 (v_exp_f32 and v_log_f32 are base-2 instructions: log-probs are scaled by log2(e) once while they are
// staged, the lattices hold log2 values); the next frame's log-probs are fetched while the current one is being combined.  The
// lattices go to an L2/MALL-resident workspace laid out [t][s], states in their natural order: a lane stores its states as one 16- or
// 8-byte piece of a contiguous row.  Every CTC_RENORM frames a wave subtracts floor(max over ITS states) from its part of the column: an
// INTEGER in the log2 domain, so the subtraction, the running sum of the offsets (kept per wave and block of CTC_RENORM frames beside
// the lattice) and the conversion of the neighbour's edge states (whose owner's offsets are published the same way) are exact in
// fp32.  Values stay O(10) near the likely states instead of growing to ~3 T, and the fp32 lattice keeps ~1e-5 precision in the
// gradient at T = 753, where an unnormalised fp32 lattice (ATen's CPU/CUDA kernels) is at ~1e-3 (scratch/ctc_prec.py vs float64).
// Kernel 2 -- one wave per frame: posterior[c] = sum_{s: l'_s = c} exp2(alpha + beta - total - lp) accumulated in LDS bins,
// grad = exp(lp) - posterior for t < olen, 0 beyond.
#include "common.h"

#define CTC_NEG (-1e30f)
#define CTC_DEAD (-1e29f)        // anything below is "impossible"
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453
#define CTC_RENORM 8
#define CTC_RENORM_LOG2 3
#define CTC_EMPTY 0xFFFFFFFFu    // "not written yet": a NaN bit pattern; published words are mapped away from it (ctc_word)
#ifndef CTC_SPIN_LIMIT
#define CTC_SPIN_LIMIT (1 << 22) // a consumer never spins this long unless the producer died: give up (the loss becomes NaN) rather than hang
#endif
// Diagnostic build only (python -m convasr_amd.build --variant ctcskip -DCONVASR_CTC_SKIP_PUBLISH=100 -DCTC_SPIN_LIMIT=65536;
// tests/test_training_features_gpu.py): the producing wave of every sweep "dies" at that frame -- it never publishes the frame's edge states -- so that
// the consumer's give-up path runs: the loss of every utterance long enough must come out NaN, and the launch must end.

// base-2 log-sum-exp of three / two values >= the sentinel.  The largest term is exp2(0) = 1 exactly, so only the other terms go
// through v_exp_f32 (max3 / med3 / min3 are single instructions): 3 transcendental instructions per label state, 2 per blank state.
__device__ __forceinline__ float lse3(float a, float b, float c) {
	const float m = fmaxf(a, fmaxf(b, c));
	const float md = __builtin_amdgcn_fmed3f(a, b, c), lo = fminf(a, fminf(b, c));
	return m + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(md - m) + __builtin_amdgcn_exp2f(lo - m));
}
__device__ __forceinline__ float lse2(float a, float b) {
	// max / min as v_med3_f32 with +-inf: fmaxf / fminf first canonicalise both operands (one v_max_f32 x, x each) under IEEE rules
	const float m = __builtin_amdgcn_fmed3f(a, b, INFINITY);
	return m + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(a, b, -INFINITY) - m));
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


// -inf (and anything below the sentinel) becomes the sentinel; a NaN stays a NaN (fmaxf would drop it)
__device__ __forceinline__ float ctc_floor(float v) { return v < CTC_NEG ? CTC_NEG : v; }

__device__ __forceinline__ uint32_t ctc_word(float v) {
	const uint32_t b = __builtin_bit_cast(uint32_t, v);
	return b == CTC_EMPTY ? 0x7FC00000u : b;  // a NaN that came in through the log-probs keeps being a NaN
}

// Edge slots are read and written with relaxed workgroup-scope atomics: plain ds_read / ds_write that the compiler neither caches in
// a register nor fences (a volatile access makes it wait for vmcnt(0), i.e. for the frame's lattice stores to reach L2 -- ~0.5 us in
// every frame, measured).
__device__ __forceinline__ uint32_t slot_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void slot_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct CtcSweep {
	const float* lpb;        // the utterance's log-probs: LDS copy [Tb][C + 1] scaled by log2 e (column C = sentinel), or global [T][C]
	const int64_t* tg;       // the utterance's targets
	float* lat;              // this sweep's lattice, utterance base: rows of `cap` states in their natural order
	float* off;              // this wave's offsets [NB]
	uint32_t* edge;          // this sweep's edge slots [T] (alpha) / [T][2] (beta)
	uint32_t* ecum;          // the edge producer's offsets [NB]
	float *fin, *fcum;       // alpha only: the last column and this wave's final offset
	float* gave_up;          // set when a consumer hit CTC_SPIN_LIMIT: the utterance's loss becomes NaN instead of a wrong number
	int T, C, Tb, L, blank, cap, s_base, lane;
};

// a lane's NS consecutive lattice values as 16- and 8-byte stores (dword-aligned: NS = 6 puts odd lanes at 8 mod 16)
typedef float ctc_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float ctc_f2 __attribute__((ext_vector_type(2), aligned(4)));
template <int NS>
__device__ __forceinline__ void store_states(float* p, const float (&v)[NS]) {
#pragma unroll
	for (int i = 0; i + 4 <= NS; i += 4) *reinterpret_cast<ctc_f4*>(p + i) = ctc_f4{v[i], v[i + 1], v[i + 2], v[i + 3]};
	if (NS & 2) *reinterpret_cast<ctc_f2*>(p + (NS & ~3)) = ctc_f2{v[NS & ~3], v[(NS & ~3) + 1]};
}

// One wave's share of one sweep: the 64 NS states from s_base on, NS = 2 NP consecutive ones per lane.  The producer of the sweep's
// edge (the wave with the lower states for alpha, with the upper states for beta) never waits; it is given the larger share (NPH
// pairs per lane against the consumer's NPL) because the consumer also pays for the edge words.
template <int NP, bool FWD, bool CONSUMES, bool LP_LDS>
__device__ __forceinline__ void ctc_sweep(const CtcSweep& q) {
	constexpr int NS = 2 * NP, EW = FWD ? 1 : 2;
	const int lane = q.lane, L = q.L, C = q.C, Tb = q.Tb, blank = q.blank, LP = q.cap;
	const int s0 = q.s_base + NS * lane;   // even
	const int CS = LP_LDS ? C + 1 : C;   // row pitch of lpb
	int cls[NS];
	bool skip[NS], valid[NS];
#pragma unroll
	for (int i = 0; i < NS; ++i) {
		const int s = s0 + i;
		valid[i] = s < L;
		const bool nb = (i & 1) && valid[i];
		// the per-frame score of a state beyond the extended target is the sentinel (column C of the LDS copy): no validity test below
		cls[i] = nb ? (int)q.tg[s >> 1] : (valid[i] || !LP_LDS ? blank : C);
		// a label state may take the s-2 (alpha) / s+2 (beta) transition when its neighbour label differs
		skip[i] = nb && (FWD ? (s >= 3 && q.tg[s >> 1] != q.tg[(s >> 1) - 1]) : (s + 2 < L && q.tg[s >> 1] != q.tg[(s >> 1) + 1]));
	}
	auto score = [&](float v, int i) { return LP_LDS ? v : (valid[i] ? ctc_floor(v * CTC_LOG2E) : CTC_NEG); };
	constexpr bool consumes = CONSUMES;   // wave 1 of alpha / wave 0 of beta
	const bool edge_lane = FWD ? lane == 63 : lane == 0;
	float* lat = q.lat + s0;
	float cum = 0.f;  // integer-valued: the offsets this wave has subtracted so far
	float a[NS], cur[NS], nxt[NS];
	const int t0 = FWD ? 0 : Tb - 1, dt = FWD ? 1 : -1;
	auto publish = [&](int t) {
#ifdef CONVASR_CTC_SKIP_PUBLISH
		if (t == CONVASR_CTC_SKIP_PUBLISH) return;
#endif
		if (!consumes && edge_lane) {
			if (FWD) slot_store(q.edge + t, ctc_word(a[NS - 1]));
			else { slot_store(q.edge + 2 * t, ctc_word(a[0])); slot_store(q.edge + 2 * t + 1, ctc_word(a[1])); }
		}
	};
	{
		const float* row = q.lpb + (int64_t)t0 * CS;
#pragma unroll
		for (int i = 0; i < NS; ++i) {
			const int s = s0 + i;
			// alpha(0, s) is non-zero for s < 2, beta(Tb-1, s) for s >= L - 2
			a[i] = ((FWD ? s < 2 : s >= L - 2) && valid[i]) ? score(row[cls[i]], i) : CTC_NEG;
		}
		store_states<NS>(lat + (int64_t)t0 * LP, a);
		if (lane == 0) { q.off[0] = 0.f; if (!consumes) slot_store(q.ecum, ctc_word(0.f)); }
		publish(t0);
		if (Tb > 1) {
#pragma unroll
			for (int i = 0; i < NS; ++i) cur[i] = score(q.lpb[(int64_t)(t0 + dt) * CS + cls[i]], i);
		}
	}
	uint32_t x0 = CTC_EMPTY, x1 = FWD ? 0u : CTC_EMPTY, xc = 0;
	for (int k = 1; k < Tb; ++k) {  // k-th frame of the sweep
		const int t = t0 + k * dt;
		// the states beyond the wave's edge as their owner left them in the previous frame, converted to this wave's offset.  The slot was
		// read one frame ahead (x0, x1): once this wave trails its producer by a full frame the read-ahead always finds the slot filled
		// and the LDS round trip is off the frame's critical path; until then it re-reads here.
		float e0 = CTC_NEG, e1 = CTC_NEG;
		if (consumes) {
			const uint32_t* e = q.edge + (int64_t)(t - dt) * EW;
			if (__builtin_expect(x0 == CTC_EMPTY || x1 == CTC_EMPTY, 0)) {
				int spins = 0;
				do {
					x0 = slot_load(e);
					if (!FWD) x1 = slot_load(e + 1);
				} while ((x0 == CTC_EMPTY || x1 == CTC_EMPTY) && ++spins < CTC_SPIN_LIMIT);
				if (spins >= CTC_SPIN_LIMIT) *q.gave_up = 1.f;
				// the producer published its block's offset before that frame's states (LDS operations of a wave complete in order)
				xc = slot_load(q.ecum + ((k - 1) >> CTC_RENORM_LOG2));
			}
			const float d = __builtin_bit_cast(float, xc) - cum;   // integer-valued
			e0 = __builtin_bit_cast(float, x0) + d;
			if (!FWD) e1 = __builtin_bit_cast(float, x1) + d;
			x0 = slot_load(e + dt * EW);   // frame k's slot, for frame k + 1 (the slot array has T entries: t is in range)
			if (!FWD) x1 = slot_load(e + dt * EW + 1);
			xc = slot_load(q.ecum + (k >> CTC_RENORM_LOG2));   // only meaningful when x0 / x1 are
		}
		// (after the edge: a wait for the edge words must not also wait for these reads)
		if (k + 1 < Tb) {
#pragma unroll
			for (int i = 0; i < NS; ++i) nxt[i] = q.lpb[(int64_t)(t + dt) * CS + cls[i]];
		}
		float n[NS];
		if (FWD) {
			float p1 = wave_shr1(a[NS - 1], CTC_NEG);   // state s0 - 1
			if (lane == 0) p1 = e0;
#pragma unroll
			for (int i = NS - 1; i >= 0; --i) {
				const float m1 = i >= 1 ? a[i >= 1 ? i - 1 : 0] : p1;
				const float m2 = i >= 2 ? a[i >= 2 ? i - 2 : 0] : p1;   // i == 1: state s0 - 1 again; i == 0 is a blank
				n[i] = ((i & 1) ? lse3(a[i], m1, skip[i] ? m2 : CTC_NEG) : lse2(a[i], m1)) + cur[i];
			}
		} else {
			float q0 = wave_shl1(a[0], CTC_NEG), q1 = wave_shl1(a[1], CTC_NEG);   // states s0 + NS, s0 + NS + 1
			if (lane == 63) { q0 = e0; q1 = e1; }
#pragma unroll
			for (int i = 0; i < NS; ++i) {
				const float m1 = i + 1 < NS ? a[i + 1 < NS ? i + 1 : 0] : q0;
				const float m2 = i + 2 < NS ? a[i + 2 < NS ? i + 2 : 0] : q1;   // i == NS - 1; i == NS - 2 is a blank
				n[i] = ((i & 1) ? lse3(a[i], m1, skip[i] ? m2 : CTC_NEG) : lse2(a[i], m1)) + cur[i];
			}
		}
		if ((k & (CTC_RENORM - 1)) == 0) {
			float m = n[0];
#pragma unroll
			for (int i = 1; i < NS; ++i) m = fmaxf(m, n[i]);
			m = wave_max_dpp(m);
			if (m > CTC_DEAD && m < INFINITY) {
				const float kk = floorf(m);
#pragma unroll
				for (int i = 0; i < NS; ++i) n[i] -= kk;
				cum += kk;
			}
			if (lane == 0) { q.off[k >> CTC_RENORM_LOG2] = cum; if (!consumes) slot_store(q.ecum + (k >> CTC_RENORM_LOG2), ctc_word(cum)); }
		}
#pragma unroll
		for (int i = 0; i < NS; ++i) a[i] = n[i];
		store_states<NS>(lat + (int64_t)t * LP, a);
		publish(t);
#pragma unroll
		for (int i = 0; i < NS; ++i) cur[i] = score(nxt[i], i);  // the wait for the read-ahead is here, after the frame's arithmetic
	}
	if (FWD) {
#pragma unroll
		for (int i = 0; i < NS; ++i) q.fin[s0 + i] = a[i];
		if (lane == 0) q.fcum[0] = cum;
	}
}

// LP_LDS: the utterance's whole (T, C) log-prob slab is first copied into LDS (117 KB at T = 753, C = 38) so that the T-step
// recurrences read their per-frame class scores at LDS latency instead of L2 latency.
// Lattices: [B][T][cap] with cap = 128 (NPH + NPL) states per frame in their natural order.
// offs: [2][B][2][NB] integer-valued offsets, NB = T / CTC_RENORM + 1 (true log2 alpha(t, s) = lattice value + offs[0][b][wave of s][k / 8]
// with k = t for alpha, olen - 1 - t for beta: offs[1]; wave of s = s >= 128 NPH for alpha, s >= 128 NPL for beta);
// tot: [B][2] = {integer part, remainder} of the utterance's log2 likelihood.
// Dynamic LDS (32-bit words): fin[cap] | fcum[2], gave-up flag, pad | edge slots: alpha [T], beta [T][2], offsets [2][NB] | lp slab when LP_LDS.
template <int NPH, int NPL, bool LP_LDS>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ olen,
                                                             const int64_t* __restrict__ ylen, float* __restrict__ nll, float* __restrict__ alpha,
                                                             float* __restrict__ beta, float* __restrict__ offs, float* __restrict__ tot,
                                                             int B, int T, int C, int S_max, int blank) {
	extern __shared__ __attribute__((aligned(16))) float ctc_smem[];
	constexpr int CAP = 128 * (NPH + NPL);
	const int NB = T / CTC_RENORM + 1;
	float* const fin = ctc_smem;
	float* const fcum = fin + CAP;
	uint32_t* const edge = reinterpret_cast<uint32_t*>(fcum + 4);
	const int n_edge = 3 * T + 2 * NB;
	float* const lpl = fcum + 4 + ((n_edge + 3) & ~3);
	const int b = blockIdx.x;
	const int Tb = (int)olen[b], S = (int)ylen[b], L = 2 * S + 1;
	const float* lpg = lp + (int64_t)b * T * C;
	if (Tb <= 0 || Tb > T) {  // uniform over the workgroup
		if (threadIdx.x == 0) nll[b] = (Tb == 0 && S == 0) ? 0.f : INFINITY;
		return;
	}
	for (int i = threadIdx.x; i < n_edge; i += blockDim.x) edge[i] = CTC_EMPTY;
	if (threadIdx.x == 0) fcum[2] = 0.f;
	if (LP_LDS) {
		const int n = Tb * (C + 1);
		for (int i = threadIdx.x; i < n; i += blockDim.x) {
			const int t = i / (C + 1), c = i - t * (C + 1);
			lpl[i] = c < C ? ctc_floor(lpg[(int64_t)t * C + c] * CTC_LOG2E) : CTC_NEG;
		}
	}
	__syncthreads();
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool fwd = wave < 2;
	CtcSweep q;
	q.lpb = LP_LDS ? lpl : lpg;
	q.tg = targets + (int64_t)b * S_max;
	q.lat = (fwd ? alpha : beta) + (int64_t)b * T * CAP;
	q.off = offs + (((int64_t)(fwd ? 0 : 1) * B + b) * 2 + (wave & 1)) * NB;
	q.edge = edge + (fwd ? 0 : T);
	q.ecum = edge + 3 * T + (fwd ? 0 : NB);
	q.fin = fin; q.fcum = fcum + (wave & 1); q.gave_up = fcum + 2;
	q.T = T; q.C = C; q.Tb = Tb; q.L = L; q.blank = blank; q.cap = CAP; q.lane = threadIdx.x & 63;
	// alpha: wave 0 = states [0, 128 NPH) produces the edge, wave 1 the rest; beta: wave 2 = states [0, 128 NPL) consumes, wave 3 produces
	q.s_base = wave == 1 ? 128 * NPH : (wave == 3 ? 128 * NPL : 0);
	if (wave == 0) ctc_sweep<NPH, true, false, LP_LDS>(q);
	else if (wave == 1) ctc_sweep<NPL, true, true, LP_LDS>(q);
	else if (wave == 2) ctc_sweep<NPL, false, true, LP_LDS>(q);
	else ctc_sweep<NPH, false, false, LP_LDS>(q);
	__syncthreads();
	if (threadIdx.x == 0) {
		const float c1 = fcum[L - 1 >= 128 * NPH];
		const float l1 = fin[L - 1], l2 = L >= 2 ? fin[L - 2] + (fcum[L - 2 >= 128 * NPH] - c1) : CTC_NEG;
		const float m = fmaxf(l1, l2);
		const float rem = m + log2f(exp2f(l1 - m) + exp2f(l2 - m));
		nll[b] = fcum[2] != 0.f ? NAN : (m > CTC_DEAD ? (float)(-CTC_LN2 * ((double)c1 + (double)rem)) : INFINITY);   // NaN log-probs: NaN (as the reference)
		tot[2 * b] = c1;
		tot[2 * b + 1] = rem;
	}
}

// cap: states per lattice row; split_a / split_b: first state of the upper wave of the alpha / beta sweep
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ olen,
                                                       const int64_t* __restrict__ ylen, const float* __restrict__ nll, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const float* __restrict__ offs, const float* __restrict__ tot, float* __restrict__ grad,
                                                       int B, int T, int C, int S_max, int blank, int t_per_block, int cap, int split_a, int split_b) {
	extern __shared__ float bins[];  // [4][C]
	const int NB = T / CTC_RENORM + 1;
	const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int Tb = (int)olen[b], S = (int)ylen[b], L = 2 * S + 1;
	const bool feasible = nll[b] < INFINITY;
	const int64_t* tg = targets + (int64_t)b * S_max;
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
		const float* ar = alpha + ((int64_t)b * T + t) * cap;
		const float* br = beta + ((int64_t)b * T + t) * cap;
		const float* oa = offs + (int64_t)b * 2 * NB + (t >> CTC_RENORM_LOG2);
		const float* ob = offs + ((int64_t)B + b) * 2 * NB + ((Tb - 1 - t) >> CTC_RENORM_LOG2);
		// alpha + beta - total = (lattice values - remainder of total) + (offsets - integer part of total): the second group is exact
		const float ti = tot[2 * b], tr = tot[2 * b + 1];
		const float oa0 = oa[0] - ti, oa1 = oa[NB] - ti, ob0 = ob[0], ob1 = ob[NB];
		const float lpb = row[blank] * CTC_LOG2E;
		float blank_sum = 0.f;
		for (int s = lane; s < L; s += 64) {
			const float shift = ((s >= split_a ? oa1 : oa0) + (s >= split_b ? ob1 : ob0)) - tr;
			const float ab = ar[s] + br[s];
			if (s & 1) {
				const int c = (int)tg[s >> 1];
				atomicAdd(mybins + c, __builtin_amdgcn_exp2f(ab + (shift - row[c] * CTC_LOG2E)));
			} else blank_sum += __builtin_amdgcn_exp2f(ab + (shift - lpb));
		}
		blank_sum = wave_sum(blank_sum);
		if (lane == 0) atomicAdd(mybins + blank, blank_sum);
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		for (int c = lane; c < C; c += 64) grow[c] = __expf(row[c]) - mybins[c];
		__builtin_amdgcn_wave_barrier();
	}
}

// (blank, label) pairs per lane of the sweep's two waves {producer of the edge, consumer}: the smallest 128 (NPH + NPL) >= 2 S_max + 1
static int ctc_split(int S_max, int* nph, int* npl) {
	const int pairs = (2 * S_max + 1 + 127) / 128;  // per lane, both waves together
	if (pairs > 16) return -1;
	const int p = pairs < 2 ? 2 : pairs;
	*nph = (p + 1) / 2;
	*npl = p / 2;
	return 0;
}

extern "C" int64_t convasr_ctc_workspace_bytes(int B, int T, int S_max) {
	int nph, npl;
	if (ctc_split(S_max, &nph, &npl) < 0) return -1;
	const int NB = T / CTC_RENORM + 1;
	return (2 * (int64_t)B * T * 128 * (nph + npl) + 4 * (int64_t)B * NB + 2 * (int64_t)B) * (int64_t)sizeof(float);  // lattices, offsets, totals
}

extern "C" int convasr_ctc_loss(const float* log_probs, const int64_t* targets, const int64_t* olen, const int64_t* ylen, float* nll, float* grad,
                                void* workspace, int B, int T, int C, int S_max, int blank, void* stream) {
	CONVASR_CHECK_ARG(log_probs && targets && olen && ylen && nll && workspace && B > 0 && T > 0 && C > 1 && S_max >= 0 && blank >= 0 && blank < C, "ctc_loss: bad arguments");
	int nph, npl;
	if (ctc_split(S_max, &nph, &npl) < 0) return convasr_fail(CONVASR_EUNSUPPORTED, "ctc_loss: target length %d > 1023", S_max);
	CONVASR_CHECK_ARG(C <= 8192, "ctc_loss: C %d > 8192", C);
	hipStream_t s = (hipStream_t)stream;
	const int NB = T / CTC_RENORM + 1, cap = 128 * (nph + npl);
	float* alpha = (float*)workspace;
	float* beta = alpha + (int64_t)B * T * cap;
	float* offs = beta + (int64_t)B * T * cap;
	float* tot = offs + 4 * (int64_t)B * NB;
	const int t_per_block = 32;
	dim3 ggrid((T + t_per_block - 1) / t_per_block, B);
	const size_t gsmem = 4 * (size_t)C * sizeof(float);
	const size_t lds_fixed = (size_t)(cap + 4 + ((3 * T + 2 * NB + 3) & ~3)) * sizeof(float), lds_lp = (size_t)T * (C + 1) * sizeof(float);
	CONVASR_CHECK_ARG(lds_fixed <= 159 * 1024, "ctc_loss: T %d too long for the edge slots", T);
	const bool in_lds = lds_fixed + lds_lp <= 159 * 1024;  // T = 753, C = 38: 117 KB + 12 KB of the 160 KiB of a CU
#define CTC_LAUNCH(NPH, NPL, LDS) do { \
		auto kern = ctc_alpha_beta_kernel<NPH, NPL, LDS>; \
		static unsigned long long set = 0; \
		convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), set); \
		hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_fixed + (LDS ? lds_lp : 0), s, log_probs, targets, olen, ylen, nll, alpha, beta, offs, tot, B, T, C, S_max, blank); \
	} while (0)
#define CTC_CASE(NPH, NPL) case NPH + NPL: if (in_lds) CTC_LAUNCH(NPH, NPL, true); else CTC_LAUNCH(NPH, NPL, false); break;
	switch (nph + npl) {
		CTC_CASE(1, 1) CTC_CASE(2, 1) CTC_CASE(2, 2) CTC_CASE(3, 2) CTC_CASE(3, 3) CTC_CASE(4, 3) CTC_CASE(4, 4)
		CTC_CASE(5, 4) CTC_CASE(5, 5) CTC_CASE(6, 5) CTC_CASE(6, 6) CTC_CASE(7, 6) CTC_CASE(7, 7) CTC_CASE(8, 7) CTC_CASE(8, 8)  // 512-1,023 labels
	}
#undef CTC_CASE
#undef CTC_LAUNCH
	if (grad) hipLaunchKernelGGL(ctc_grad_kernel, ggrid, dim3(256), gsmem, s, log_probs, targets, olen, ylen, nll, alpha, beta, offs, tot, grad, B, T, C, S_max, blank, t_per_block, cap, 128 * nph, 128 * npl);
	CONVASR_CHECK_LAUNCH("ctc_loss");
	return 0;
}
