// bf16 fast path of the implicit-GEMM conv (forward and dgrad), stride 1, Cin % 64 == 0.
//
// Same decomposition as conv.hip (one workgroup = a t x co tile of one utterance; the X rows incl. halo are staged once per
// 64-channel slab and shared by all K taps) but built around LDS-DMA (`buffer_load_dwordx4 ... lds`):
//   * 256(t) x 128(co) tile, 8 waves (4 x 2, each 64 x 64): two waves per SIMD cover each other's waits;
//   * X slabs double-buffered, weight tiles in a 3-deep ring; loads are issued two K-steps ahead and stay in flight across
//     the raw s_barrier -- the only wait in the loop is a counted `s_waitcnt vmcnt(2)`;
//   * no staging registers and no ds_write traffic (the v1 kernel spends ~45 % of its LDS cycles on ds_write_b128);
//   * the conflict-free XOR swizzle is applied on the per-lane SOURCE address (the DMA destination is lane-linear);
//   * the conv's zero padding is the buffer descriptor's range check: rows before 0 / after Tin read as zeros.
#include "conv_v2_common.h"

template <typename O, int MODE> __global__ __launch_bounds__(V2_THREADS, 2) void conv1d_igemm_v2_kernel(ConvParams p) {
	constexpr bool PIPE = MODE == 1;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;

	const int v = xcd_remap(blockIdx.x, p.total_tiles);
	const int ntile = v % p.n_tiles, mtile = v / p.n_tiles;  // (walking weight tiles in L2-sized groups per XCD measured +-1 %: not the limiter)
	const int b = mtile / p.m_tiles_per_b, t0 = (mtile % p.m_tiles_per_b) * V2_BM;
	const int co0 = ntile * BN;
	const int tin0 = t0 - p.pad;  // stride 1

	const int xbytes = p.x_rows * ROW_BYTES;  // x_rows is a multiple of 8: whole 1-KiB DMA pieces
	char* const xbuf = smem;
	char* const wbuf = smem + 2 * xbytes;
	const int row_bytes = p.Cin * 2;
	const v4i32 xsrc = make_srd(reinterpret_cast<const bf16_t*>(p.x) + (int64_t)b * p.Tin * p.Cin, (unsigned)(p.Tin * row_bytes));
	const v4i32 wsrc = make_srd(p.w, (unsigned)(p.K * p.CoutPad * row_bytes));
	const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
	const int n_cib = p.Cin >> 6;
	const int x_units = p.x_rows >> 3;

	// per-lane source offsets of the pieces this wave issues (tile-relative), computed once
	const int xlane = v2_src_offset(lane, row_bytes);  // piece u adds 8 rows: (u * 8) * row_bytes, and the swizzle term is periodic in 8 pairs = 16 rows
	auto issue_x = [&](int cib) {
		const unsigned dst = lds_base + (cib & 1) * xbytes;
		const int base = tin0 * row_bytes + cib * 128;
		for (int u = wave; u < x_units; u += 8) {
			// piece u covers rows 8u .. 8u+7 = pairs 4u .. 4u+3; (pair & 7) depends on u's parity
			const int off = (u & 1) ? v2_src_offset(64 + lane, row_bytes) - 8 * row_bytes : xlane;
			dma16(xsrc, __builtin_amdgcn_readfirstlane(dst + u * 1024), base + u * 8 * row_bytes + off);
		}
	};
	const int wl0 = v2_src_offset(wave * 128 + lane, row_bytes), wl1 = v2_src_offset(wave * 128 + 64 + lane, row_bytes);
	auto issue_w = [&](int q_cib, int q_tap, int slot) {
		const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * xbytes + slot * V2_WSLOT + wave * 2048);
		const int base = (q_tap * p.CoutPad + co0) * row_bytes + q_cib * 128;
		dma16(wsrc, dst, base + wl0);
		dma16(wsrc, dst + 1024, base + wl1);
	};

	f32x16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

	const int Q = n_cib * p.K;
	const int wrow0 = wn * 64 + r;
	const int woff0 = (wrow0 >> 1) << 8, wpar0 = (wrow0 & 1) << 3, wsw0 = (wrow0 >> 1) & 7;
	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	struct Frag { u32x4 a0, a1, b0, b1; };
	typedef const __attribute__((address_space(3))) u32x4* lds_u4;
	// Fragment addressing with a handful of VALU ops per step (the matrix pipe and the address math share the issue port):
	// lds_off(row, chunk) puts the chunk index in address bits 4..6 XORed with the row-pair index, so stepping the k-substep
	// (chunk += 2) is `address ^ (kk << 5)`, and rows 32 apart (16 pairs) share the swizzle term: a1 = a0 + 4096, b1 = b0 + 4096.
	const unsigned w0 = lds_base + 2 * xbytes + woff0 + ((wpar0 | (h ^ wsw0)) << 4);
	auto load_frag = [&](unsigned xs_off, unsigned ws_off, int tap_, int kk, Frag& f) {
		const int xrow0 = wm * 64 + r + tap_ * p.dil;
		const unsigned xa = (lds_base + xs_off + ((xrow0 >> 1) << 8) + ((((xrow0 & 1) << 3) | (h ^ ((xrow0 >> 1) & 7))) << 4)) ^ (kk << 5);
		const unsigned wa = (w0 + ws_off) ^ (kk << 5);
		f.a0 = *(lds_u4)(size_t)(xa);
		f.a1 = *(lds_u4)(size_t)(xa + 4096);
		f.b0 = *(lds_u4)(size_t)(wa);
		f.b1 = *(lds_u4)(size_t)(wa + 4096);
	};
	auto mma1 = [](const u32x4& a, const u32x4& bb, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0); };
	auto mma_frag = [&](const Frag& f) {
		mma1(f.a0, f.b0, acc[0][0]);
		mma1(f.a0, f.b1, acc[0][1]);
		mma1(f.a1, f.b0, acc[1][0]);
		mma1(f.a1, f.b1, acc[1][1]);
	};

	if constexpr (MODE == 2) {
		// K >= 2, TWO taps per barrier interval: the per-step costs (DMA issue, counted wait, barrier skew, post-barrier LDS
		// round trip) are paid once per 32 MFMAs per wave instead of once per 16.  Weight ring of 4 slots = the current pair +
		// the next pair (in flight during the whole step, ~2000+ cycles of cover); fragments ping-pong through the 8 substeps.
		const int npb = (p.K + 1) >> 1, P = n_cib * npb;
		issue_x(0);
		issue_w(0, 0, 0);
		if (p.K > 1) issue_w(0, 1, 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		const bool late = wave >= 4;
		int cib = 0, pi = 0;
		Frag f0, f1;
		for (int sidx = 0; sidx < P; ++sidx) {
			const int t0_ = 2 * pi, nt = min(2, p.K - t0_);
			int cib1 = cib, pi1 = pi + 1;
			if (pi1 == npb) { pi1 = 0; ++cib1; }
			const unsigned xs = (cib & 1) * xbytes, ws0 = ((sidx & 1) * 2) * V2_WSLOT, ws1 = ws0 + V2_WSLOT;
			auto issue_step = [&]() {
				if (pi == 0 && cib + 1 < n_cib) issue_x(cib + 1);
				if (sidx + 1 < P) {
					const int sl = ((sidx + 1) & 1) * 2;
					issue_w(cib1, 2 * pi1, sl);
					if (2 * pi1 + 1 < p.K) issue_w(cib1, 2 * pi1 + 1, sl + 1);
				}
			};
			if (!late) issue_step();
			load_frag(xs, ws0, t0_, 0, f0);
			load_frag(xs, ws0, t0_, 1, f1);
			mma_frag(f0);
			load_frag(xs, ws0, t0_, 2, f0);
			mma_frag(f1);
			if (late) issue_step();
			load_frag(xs, ws0, t0_, 3, f1);
			mma_frag(f0);
			if (nt == 2) {
				load_frag(xs, ws1, t0_ + 1, 0, f0);
				mma_frag(f1);
				load_frag(xs, ws1, t0_ + 1, 1, f1);
				mma_frag(f0);
				load_frag(xs, ws1, t0_ + 1, 2, f0);
				mma_frag(f1);
				load_frag(xs, ws1, t0_ + 1, 3, f1);
				mma_frag(f0);
			}
			mma_frag(f1);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			cib = cib1; pi = pi1;
		}
	} else if constexpr (PIPE) {
		// K >= 2.  Weight ring of 4 slots, tiles issued THREE steps ahead: the tiles of step q + 1 are already published (landed +
		// barrier) while step q runs, so the fragments of k-substep kk + 1 -- including substep 0 of the NEXT step, across the
		// barrier -- are read while the MFMAs of kk issue, from two alternating register sets (no copies).  (A deeper pipeline
		// that pre-reads all 16 fragments of the next step measured 3-4 % slower at 208 VGPRs: LDS latency is not the limiter.)
		issue_x(0);
		int ci_ = 0, ti_ = 0;  // coordinates of the next weight tile to issue
		for (int i = 0; i < 3 && i < Q; ++i) { issue_w(ci_, ti_, i); if (++ti_ == p.K) { ti_ = 0; ++ci_; } }
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();

		int cib = 0, tap = 0, slot = 0;
		const bool late = wave >= 4;  // waves 4-7 issue their DMA pieces mid-step, waves 0-3 up front (no lockstep on a SIMD)
		auto issue_step = [&](int q) {
			if (tap == 0 && cib + 1 < n_cib) issue_x(cib + 1);
			if (q + 3 < Q) { issue_w(ci_, ti_, (slot + 3) & 3); if (++ti_ == p.K) { ti_ = 0; ++ci_; } }
		};
		auto end_step = [&](int q) {
			if (q + 3 < Q) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			if (++tap == p.K) { tap = 0; ++cib; }
			slot = (slot + 1) & 3;
		};
		{
			Frag f0, f1;
			load_frag(0, 0, 0, 0, f0);
			for (int q = 0; q < Q; ++q) {
				if (!late) issue_step(q);
				const unsigned xs = (cib & 1) * xbytes, ws = slot * V2_WSLOT;
				int cib1 = cib, tap1 = tap + 1;
				if (tap1 == p.K) { tap1 = 0; ++cib1; }
				load_frag(xs, ws, tap, 1, f1);
				mma_frag(f0);
				load_frag(xs, ws, tap, 2, f0);
				mma_frag(f1);
				if (late) issue_step(q);
				load_frag(xs, ws, tap, 3, f1);
				mma_frag(f0);
				if (q + 1 < Q) load_frag((cib1 & 1) * xbytes, ((slot + 1) & 3) * V2_WSLOT, tap1, 0, f0);
				mma_frag(f1);
				end_step(q);
			}
		}
	} else {
		issue_x(0);
		issue_w(0, 0, 0);
		if (Q > 1) issue_w(p.K > 1 ? 0 : 1, p.K > 1 ? 1 : 0, 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();

		int cib = 0, tap = 0;       // coordinates of step q
		int cib2 = 0, tap2 = 0;     // coordinates of step q + 2
		for (int i = 0; i < 2; ++i) { if (++tap2 == p.K) { tap2 = 0; ++cib2; } }
		int slot = 0;
		for (int q = 0; q < Q; ++q) {
			if (tap == 0 && cib + 1 < n_cib) issue_x(cib + 1);
			const bool more = q + 2 < Q;
			if (more) issue_w(cib2, tap2, slot >= 1 ? slot - 1 : 2);  // (slot + 2) % 3

			const unsigned xs = (cib & 1) * xbytes, ws = slot * V2_WSLOT;
#pragma unroll
			for (int kk = 0; kk < 4; ++kk) {
				Frag f;
				load_frag(xs, ws, tap, kk, f);
				mma_frag(f);
			}
			// everything but the two weight pieces issued in this step has landed (X of the next slab is older than them)
			if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			if (++tap == p.K) { tap = 0; ++cib; }
			if (++tap2 == p.K) { tap2 = 0; ++cib2; }
			slot = slot == 2 ? 0 : slot + 1;
		}
	}

	// ---------------- epilogue (as conv.hip): bias, BN statistics, scale/shift, activation, mask, coalesced store through LDS
	constexpr int OPITCH = BN * sizeof(O) + 16;
	char* const otile = smem;
	float* const red = reinterpret_cast<float*>(smem + V2_BM * OPITCH);  // [2][4 (wm)][BN]
	const int nvalid = valid_len(p.xlen, b, p.Tout);
	const ActConst ac = act_const(p.act, p.act_lo, p.act_hi);
#pragma unroll
	for (int ni = 0; ni < 2; ++ni) {
		const int col = wn * 64 + ni * 32 + r, co = co0 + col;
		const bool cok = co < p.Cout;
		const float bias = (p.bias && cok) ? p.bias[co] : 0.f;
		const float sc = (p.scale && cok) ? p.scale[co] : 1.f, sh = (p.scale && cok) ? p.shift[co] : 0.f;
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const int row = wm * 64 + mi * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
				const int t = t0 + row;
				float val = acc[mi][ni][g] + bias;
				if (t < p.Tout) { s1 += val; s2 += val * val; }
				val = apply_act(val * sc + sh, ac);
				if (t >= nvalid) val = 0.f;
				Elem<O>::store(reinterpret_cast<O*>(otile + row * OPITCH) + col, val);
			}
		}
		if (p.stats) {
			s1 += __shfl_xor(s1, 32, 64);
			s2 += __shfl_xor(s2, 32, 64);
			if (h == 0) { red[(0 * 4 + wm) * BN + col] = s1; red[(1 * 4 + wm) * BN + col] = s2; }
		}
	}
	__syncthreads();
	if (p.stats && tid < BN && co0 + tid < p.Cout) {
		double a = 0, q2 = 0;
#pragma unroll
		for (int m = 0; m < 4; ++m) { a += (double)red[(0 * 4 + m) * BN + tid]; q2 += (double)red[(1 * 4 + m) * BN + tid]; }
		double* const prow = p.stats + (int64_t)mtile * 2 * p.Cout;  // per-(m tile) partial row, summed in a fixed order by bn_finalize
		prow[co0 + tid] = a;
		prow[p.Cout + co0 + tid] = q2;
	}
	O* const yb = reinterpret_cast<O*>(p.y) + (int64_t)b * p.Tout * p.Cout;
	constexpr int OEPC = 16 / sizeof(O), OCHUNKS = BN / OEPC;
	const bool vec_ok = ((p.Cout * sizeof(O)) & 15) == 0;
	for (int e = tid; e < V2_BM * OCHUNKS; e += V2_THREADS) {
		const int row = e / OCHUNKS, ch = e % OCHUNKS;
		const int t = t0 + row, co = co0 + ch * OEPC;
		if (t >= p.Tout || co >= p.Cout) continue;
		const O* src = reinterpret_cast<const O*>(otile + row * OPITCH) + ch * OEPC;
		O* dst = yb + (int64_t)t * p.Cout + co;
		if (vec_ok && co + OEPC <= p.Cout) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
		else
			for (int i = 0; i < OEPC && co + i < p.Cout; ++i) dst[i] = src[i];
	}
}

const void* convasr_conv_v2s_kernel(int y_dtype, int bn_fused);  // conv_v2s.hip
#define V2_DEFAULT_SMALL_SHAPE 1

// Returns 1 if the v2 kernel took the launch, 0 if the shape is outside its envelope (caller falls back to conv.hip's kernel).
int convasr_conv1d_v2_try(ConvParams p, int y_dtype, hipStream_t s, int* m_tiles_out) {
	if (p.stride != 1 || (p.Cin & 63) != 0) return 0;
	const int xr = (V2_BM - 1) + (p.K - 1) * p.dil + 1;
	p.x_rows = (xr + 15) & ~15;  // whole 1-KiB pieces and an even number of them per 16-row swizzle period
	const size_t osz = y_dtype == CONVASR_F32 ? 4 : 2;
	int mode = (p.K < 2 && (p.debug & 16)) ? 0 : ((p.debug & 64) ? 1 : 2);  // K = 1 runs in conv_v2s.hip's loop too (three X slab buffers, read-ahead across the barrier); debug bit 16: the older K = 1 kernel
	const bool small_shape_ = mode == 2 && ((p.debug & 128) != 0) == (V2_DEFAULT_SMALL_SHAPE == 0);
	size_t smem = 2 * (size_t)p.x_rows * ROW_BYTES + (mode == 0 ? 3 : (small_shape_ ? 5 : 4)) * V2_WSLOT;  // conv_v2s.hip: 3 + 2 weight slots
	if (small_shape_ && p.K == 1) smem = 3 * (size_t)p.x_rows * ROW_BYTES + 3 * V2_WSLOT;  // ... and for K = 1 three X slab buffers + the 3-slot ring
	const size_t epi = (size_t)V2_BM * (BN * osz + 16) + 8 * BN * sizeof(float);
	if (epi > smem) smem = epi;
	if (smem > 160 * 1024) return 0;
	if ((int64_t)p.Tin * p.Cin * 2 >= (1ll << 31) || (int64_t)p.K * p.CoutPad * p.Cin * 2 >= (1ll << 31)) return 0;
	p.m_tiles_per_b = (p.Tout + V2_BM - 1) / V2_BM;
	p.total_tiles = p.B * p.m_tiles_per_b * p.n_tiles;
	const void* table[2][3] = {{(const void*)conv1d_igemm_v2_kernel<bf16_t, 0>, (const void*)conv1d_igemm_v2_kernel<bf16_t, 1>, (const void*)conv1d_igemm_v2_kernel<bf16_t, 2>},
	                           {(const void*)conv1d_igemm_v2_kernel<float, 0>, (const void*)conv1d_igemm_v2_kernel<float, 1>, (const void*)conv1d_igemm_v2_kernel<float, 2>}};
	const int oi = y_dtype == CONVASR_BF16 ? 0 : 1;
	const bool small_shape = mode == 2 && ((p.debug & 128) != 0) == (V2_DEFAULT_SMALL_SHAPE == 0);  // debug bit 128 selects the non-default MFMA shape
	if (p.bn_y && !(small_shape && y_dtype == CONVASR_BF16)) return 0;  // the fused BN-backward epilogue exists in conv_v2s.hip only
	const void* kern = small_shape ? convasr_conv_v2s_kernel(y_dtype, p.bn_y != nullptr) : table[oi][mode];
	static bool attr_set[2][5] = {{false, false, false, false, false}, {false, false, false, false, false}};
	if (small_shape) mode = p.bn_y ? 4 : 3;
	if (!attr_set[oi][mode]) { (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set[oi][mode] = true; }
	// conv_v2s: a last partial round that would occupy at most half of the CUs is cut into half-width tiles (debug bit 32: off)
	p.full_tiles = p.total_tiles;
	if (small_shape && !(p.debug & 32)) {
		static int n_cu = 0;
		if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256; }
		const int rest = p.total_tiles % n_cu;
		if (rest > 0 && 2 * rest <= n_cu && ((p.total_tiles - rest) & 7) == 0) p.full_tiles = p.total_tiles - rest;
	}
	const int grid = p.full_tiles + 2 * (p.total_tiles - p.full_tiles);
	void* args[] = {&p};
	if (hipLaunchKernel(kern, dim3(grid), dim3(small_shape ? V2S_THREADS : V2_THREADS), args, smem, s) != hipSuccess) return 0;
	if (m_tiles_out) *m_tiles_out = p.B * p.m_tiles_per_b;
	return 1;
}
