// 16-bit fast path of the implicit-GEMM conv (forward and dgrad), stride 1, Cin % 64 == 0; bf16 or fp16 storage, fp32 accumulation.
//
// Same decomposition as conv.hip (one workgroup = a t x co tile of one utterance; the X rows incl. halo are staged once per
// 64-channel slab and shared by all K taps) but built around LDS-DMA (`buffer_load_dwordx4 ... lds`):
//   * 256(t) x 128(co) tile, 8 computing waves (4 x 2, each 64 x 64) on v_mfma_f32_16x16x32_{bf16,f16} + 4 loader waves;
//   * no staging registers and no ds_write traffic; the conflict-free XOR swizzle is applied on the per-lane SOURCE address (the DMA
//     destination is lane-linear); the conv's zero padding is the buffer descriptor's range check (rows before 0 / after Tin read 0).
// (History: a 32x32x16-MFMA build of the same schedule, conv_v2.hip, was bit-identical and 3-7 % slower by wall time -- under load
// the chip holds a higher clock on the 16x16x32 shape, MI355X_MICROARCH.md "DVFS give-back" -- and was removed in round 3.)
#include "conv_v2_common.h"
#include <type_traits>

// (the LDS image of a tile, its XOR swizzle and v2s_src_offset(): conv_v2_common.h, shared with conv1x1.hip)


#ifdef CONVASR_STAMPS
// Diagnostic build only (python -m convasr_amd.build --variant stamps -DCONVASR_STAMPS=1; scratch/stamps.py): per-wave cycle sums of the
// main loop's segments, written to a buffer of their own that no kernel reads.  The stamps forbid overlaps the real kernel has:
// read the SHARES, never the run time (cdna_hip_programming.md section 7, In-kernel stamps).
__device__ unsigned long long g_v2s_stamps[256 * 8 * 8];
extern "C" int convasr_debug_read_stamps(unsigned long long* host, int count) {
	return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_v2s_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#if CONVASR_STAMPS == 1
#define STAMPI(var) STAMP(var)
#else
#define STAMPI(var)  // -DCONVASR_STAMPS=2: only the two stamps around the whole tile (tile cycles and in-kernel clock of the otherwise untouched kernel)
#endif
#else
#define STAMP(var)
#define STAMPI(var)
#endif

// NB = 16-column blocks per wave: 4 -> the 256 x 128 tile, 2 -> a 256 x 64 half tile (same X tile, half the W rows).  The last
// partial round of a launch (total_tiles mod 256 workgroups on 256 CUs) is cut into half tiles so that it occupies all CUs for
// ~0.6 of a round instead of a fraction of them for a whole one; per-element sums are unchanged (same k order).
// BM_ = rows (frames) of the tile: 256; 192 (three 16-row blocks per wave instead of four) exists for A/B runs only (see the dispatcher).
// BNF: 0 = plain epilogue; 1 = fused BN-backward sums, g re-derived per element on the vector ALU (any activation, no stored gates);
// 2 = fused BN-backward sums from the stored one-bit gates, summed on the MATRIX pipe (below, "Fused pass 1 ... form 2").
template <typename T> struct HalfOnes;
template <> struct HalfOnes<bf16_t> { static constexpr unsigned pair = 0x3F803F80u; };
template <> struct HalfOnes<f16_t> { static constexpr unsigned pair = 0x3C003C00u; };
template <> struct HalfOnes<float> { static constexpr unsigned pair = 0u; };  // (never used: the fused epilogues exist for 16-bit outputs only)

template <typename I, typename O, int NB, int BNF, int BM_, bool SK = false> __device__ __forceinline__ void v2s_tile(const ConvParams& p, char* smem, const int mtile, const int ntile, const int half, const int bm_full, const int split) {
	constexpr int BN_ = 32 * NB, MI = BM_ / 64, WROWS = 16 * MI;  // MI 16-row blocks = WROWS rows per wave
	// LDS map of the epilogue (everything the main loop used is dead by then): output tile, BN-statistics scratch, and for BNF == 2 the
	// consumer layer's y tile (row-major, BN_ * 2 bytes per row, brought in by the loader waves) and the 256-entry gate -> mask table
	constexpr int EPI_OPITCH = BN_ * (int)sizeof(O) + 16, EPI_YOFF = BM_ * EPI_OPITCH + 8 * BN * (int)sizeof(float), EPI_YBYTES = BM_ * BN_ * 2, EPI_LUT = EPI_YOFF + EPI_YBYTES;
	static_assert(BNF != 2 || (EPI_YOFF % 1024 == 0 && EPI_LUT + 4096 <= 160 * 1024), "fused epilogue (form 2): LDS map");
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int r16 = lane & 15, kb = lane >> 4, wm = wave >> 1, wn = wave & 1;

	// (bm_full: the launch's regular tile height -- m tiles are bm_full frames apart; BM_ < bm_full only for the short last tile of an utterance)
	const int b = mtile / p.m_tiles_per_b, t0 = (mtile % p.m_tiles_per_b) * bm_full;
	const int co0 = ntile * BN + half * BN_;
	const int tin0 = t0 - p.pad;

	const int x_rows = p.x_rows - (bm_full - BM_);  // (a multiple of 16 either way)
	const int xbytes = x_rows * ROW_BYTES;
	// X slab buffers: two (the slab being read + the next one landing); K = 1 has ONE interval per slab, so reading the next interval's first
	// fragments ahead of the barrier needs the next slab resident one interval earlier: three buffers (and only the 3-slot weight ring)
	const bool k1 = p.K == 1;
	auto xoff = [&](int c) { return (unsigned)((k1 ? c % 3 : (c & 1)) * xbytes); };
	const unsigned wbase = (k1 ? 3 : 2) * xbytes;
	const int row_bytes = p.Cin * 2;
	// split-K launches (ConvParams::cib_per_split): this workgroup's 64-channel input blocks start cib0 blocks into every row of x and w -- the
	// descriptors' bases move by that many bytes and their ranges shrink by as much, so the range check still ends where the tensors end
	// (SK is a template parameter: the training launches' instantiations carry none of this -- two more kernel parameters read at run time cost them 0.5 %, measured)
	const int cib0 = SK ? split * p.cib_per_split : 0;
	const int n_cib = SK ? min(p.cib_per_split, (p.Cin >> 6) - cib0) : (p.Cin >> 6);
	const v4i32 xsrc = make_srd(reinterpret_cast<const char*>(reinterpret_cast<const I*>(p.x) + (int64_t)b * p.Tin * p.Cin) + cib0 * 128, (unsigned)(p.Tin * row_bytes - cib0 * 128));
	const v4i32 wsrc = make_srd(reinterpret_cast<const char*>(p.w) + cib0 * 128, (unsigned)(p.K * p.CoutPad * row_bytes - cib0 * 128));
	const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
	const int x_units = x_rows >> 3;

	// Waves 0-7 compute; waves 8-11 (one per SIMD) only issue the LDS-DMA pieces.  A piece costs its issuing wave ~100 cycles
	// (M0 hand-over, address math, the buffer_load itself) during which an in-order wave issues no MFMA: with the 38 pieces of an
	// interval spread over the computing waves, each of them spent ~470 of its ~2700 cycles per interval issuing DMA (in-kernel
	// stamps, scratch/stamps.py).  A loader wave's SALU / VMEM issue does not take the matrix pipe from the two computing waves
	// of its SIMD.
	const bool loader = wave >= 8;
	const int lw = wave - 8;
	const int xlane = v2s_src_offset(lane, row_bytes);  // the swizzle is periodic in 4 pairs = the 8 rows of a piece: every piece has the same lane pattern
	auto issue_x = [&](int cib) {
		const unsigned dst = lds_base + xoff(cib);
		const int base = tin0 * row_bytes + cib * 128;
		for (int u = lw; u < x_units; u += 4)
			dma16(xsrc, __builtin_amdgcn_readfirstlane(dst + u * 1024), base + u * 8 * row_bytes + xlane);
	};
	constexpr int PPL = NB;  // 1-KiB pieces of a W slot (BN_ rows x 128 B = 4 NB pieces) per loader wave
	int wl[PPL];
#pragma unroll
	for (int j = 0; j < PPL; ++j) wl[j] = v2s_src_offset((lw * PPL + j) * 64 + lane, row_bytes);
	auto issue_w = [&](int q_cib, int q_tap, int slot) {
		const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + wbase + slot * V2_WSLOT + lw * (1024 * PPL));
		const int base = (q_tap * p.CoutPad + co0) * row_bytes + q_cib * 128;
#pragma unroll
		for (int j = 0; j < PPL; ++j) dma16(wsrc, dst + j * 1024, base + wl[j]);
	};

	f32x4 acc[MI][NB];
#pragma unroll
	for (int i = 0; i < MI; ++i)
#pragma unroll
		for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(3))) u32x4* lds_u4;
	struct Frag { u32x4 a[MI], b[NB]; };
	// lane (r16, kb) holds k = 8 kb .. 8 kb + 7 of row r16: 16-byte chunk (ks * 4 + kb) of the 128-byte slab row.  Rows 16 apart
	// share the swizzle term (+2048 B immediates); the second k32 substep is `address ^ 64` (kb ^ swz < 8, so the XOR stays inside the row).
	// byte offset of (row, k-block kb) in the image: row * 128 + ((kb ^ swz(row >> 1)) << 4), and (swz(row >> 1) << 4) = (row << 4) & 0x60
	const int kb4 = kb << 4;
	auto lane_off = [&](int row) { return (unsigned)((row << 7) + (kb4 ^ ((row << 4) & 0x60))); };
	unsigned w0 = lds_base + wbase + lane_off(wn * (16 * NB) + r16);
	asm volatile("" : "+v"(w0));  // ONE opaque loop invariant: left transparent, hipcc keeps three partial forms of this address live through the loop and, at the 168-register limit, reloads one of them from scratch in every barrier interval
	auto load_frag = [&](unsigned xs_off, unsigned ws_off, int tap_, int ks, Frag& f) {
		const unsigned xa = (lds_base + xs_off + lane_off(wm * WROWS + r16 + tap_ * p.dil)) ^ (ks << 6);
		const unsigned wa = (w0 + ws_off) ^ (ks << 6);
#ifdef CONVASR_AB_FRAGREADS
		// Diagnostic builds only (results are garbage, timing only; scratch/ab_fragreads.py): what would fewer LDS fragment reads per MFMA buy?
		// 1: only the first half of the X fragments is read (the others keep their values of the first load); 2: the same for the W
		// fragments; 3: both -- i.e. 6 / 6 / 4 ds_read_b128 per 16 MFMAs instead of 8.
		constexpr int RA = (CONVASR_AB_FRAGREADS & 1) ? MI / 2 : MI, RB = (CONVASR_AB_FRAGREADS & 2) ? NB / 2 : NB;
#pragma unroll
		for (int i = 0; i < (MI > NB ? MI : NB); ++i) {
			if (i < RA) f.a[i] = *(lds_u4)(size_t)(xa + i * 2048);
			if (i < RB) f.b[i] = *(lds_u4)(size_t)(wa + i * 2048);
		}
#else
#pragma unroll
		for (int i = 0; i < (MI > NB ? MI : NB); ++i) {
			if (i < MI) f.a[i] = *(lds_u4)(size_t)(xa + i * 2048);
			if (i < NB) f.b[i] = *(lds_u4)(size_t)(wa + i * 2048);
		}
#endif
	};
	// Scheduling hint placed after a (load_frag, mma_frag) pair: the MI + NB ds_read_b128 of the NEXT fragment go out one per MFMA
	// from the first MFMA of the current group on.  Left alone, hipcc sinks them behind the 12th-14th MFMA of the group, and the
	// next group's first MFMA then waits out the LDS latency four times per barrier interval (-3 % per launch; a read after every
	// second MFMA or two reads per MFMA measured no better than the unhinted schedule; MI355X_MICROARCH.md, LDS: up to two
	// ds_read_b128 per MFMA gap are hidden).
	auto interleave = [&]() {
#pragma unroll
		for (int i = 0; i < MI + NB; ++i) {
			__builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
			__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
		}
		if (MI * NB > MI + NB) __builtin_amdgcn_sched_group_barrier(0x008, MI * NB - (MI + NB), 0);
	};
	auto mma_frag = [&](const Frag& f) {
#pragma unroll
		for (int i = 0; i < MI; ++i)
#pragma unroll
			for (int j = 0; j < NB; ++j) acc[i][j] = Mma16<I>::run(f.a[i], f.b[j], acc[i][j]);
	};

#ifdef CONVASR_STAMPS
	unsigned long long ta = 0, tb = 0, tc = 0, td = 0, te = 0, acc_issue = 0, acc_work = 0, acc_vm = 0, acc_bar = 0, t_loop0 = 0, t_start = 0, t_end = 0;
	STAMP(t_start)
	const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz: in-kernel clock = (t_end - t_start) / (rt_end - rt_start) x 100 MHz
#endif
	{
		// One barrier interval = the (up to) two taps of one tap pair on one 64-channel slab.  W slots: a ring of THREE for the first
		// ("even") tap of a pair and a ring of TWO for the second: the even tap of interval s + 2 and the odd tap of interval s + 1 are
		// issued at the start of interval s and have landed by the barrier that ends it, so the first fragments of interval s + 1 are
		// read BEFORE that barrier (under the last MFMA group of interval s) and every interval opens with MFMAs instead of an LDS
		// round trip.  (In-kernel stamps, scratch/stamps.py: the DMA itself never makes a wave wait; the barrier and the restart
		// after it were ~15 % of an interval.)
		const int npb = (p.K + 1) >> 1, P = n_cib * npb;
		auto e_slot = [](int s_) { return (unsigned)((s_ % 3) * V2_WSLOT); };
		auto o_slot = [](int s_) { return (unsigned)((3 + (s_ & 1)) * V2_WSLOT); };
		if (loader) {
			issue_x(0);
			if (k1 && n_cib > 1) issue_x(1);
			issue_w(0, 0, 0);
			if (p.K > 1) issue_w(0, 1, 3);
			if (P > 1) { if (npb > 1) issue_w(0, 2, 1); else issue_w(1, 0, 1); }
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			int cib = 0, pi = 0;
			for (int sidx = 0; sidx < P; ++sidx) {
				int cib1 = cib, pi1 = pi + 1;  // interval sidx + 1
				if (pi1 == npb) { pi1 = 0; ++cib1; }
				int cib2 = cib1, pi2 = pi1 + 1;  // interval sidx + 2
				if (pi2 == npb) { pi2 = 0; ++cib2; }
				if (sidx + 2 < P) issue_w(cib2, 2 * pi2, (sidx + 2) % 3);
				if (sidx + 1 < P && 2 * pi1 + 1 < p.K) issue_w(cib1, 2 * pi1 + 1, 3 + ((sidx + 1) & 1));
				if (k1) { if (cib + 2 < n_cib) issue_x(cib + 2); }
				else if (pi == 0 && cib + 1 < n_cib) issue_x(cib + 1);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_s_barrier();
				cib = cib1; pi = pi1;
			}
			// the epilogue's workgroup barriers.  BNF == 2: the loaders first bring in the consumer layer's y tile (same (b, t, channel)
			// coordinates as the output tile; rows past the utterance read as zeros through the descriptor's range check), 1 KiB pieces of
			// whole rows, lane-linear: it lands under the compute waves' accumulator staging and store loop
			if (BNF == 2 && sizeof(O) == 2) {
				constexpr int YROW = BN_ * 2, RPP = 1024 / YROW;  // bytes per tile row, rows per piece
				const int y_row_bytes = p.Cout * 2;
				const v4i32 ysrc = make_srd(reinterpret_cast<const char*>(p.bn_y) + (int64_t)b * p.Tout * y_row_bytes, (unsigned)(p.Tout * y_row_bytes));
				const int ylane = ((lane * 16) / YROW) * y_row_bytes + (lane * 16) % YROW + co0 * 2;
				for (int u = lw; u < EPI_YBYTES / 1024; u += 4)
					dma16(ysrc, __builtin_amdgcn_readfirstlane(lds_base + EPI_YOFF + u * 1024), (t0 + u * RPP) * y_row_bytes + ylane);
			}
			__builtin_amdgcn_s_barrier();
			if (BNF == 2 && sizeof(O) == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (BNF && sizeof(O) == 2) __builtin_amdgcn_s_barrier();
			return;
		}
		__builtin_amdgcn_s_barrier();
		Frag f0, f1;
#ifdef CONVASR_AB_FRAGREADS
		{  // every fragment register gets a defined (stale) value once: the loop then re-reads only part of them
			const unsigned xa = lds_base + lane_off(wm * WROWS + r16), wa = w0 + e_slot(0);
#pragma unroll
			for (int i = 0; i < (MI > NB ? MI : NB); ++i) {
				if (i < MI) f0.a[i] = f1.a[i] = *(lds_u4)(size_t)(xa + i * 2048);
				if (i < NB) f0.b[i] = f1.b[i] = *(lds_u4)(size_t)(wa + i * 2048);
			}
		}
#endif
		load_frag(0, e_slot(0), 0, 0, f0);
		STAMPI(t_loop0)
		// The loop is instantiated per `pre` so that the read-ahead is unconditional inside it: a conditional LDS load makes hipcc
		// wait for ALL outstanding LDS reads at the branch join (ahead of the last MFMA group), which costs more than the read-ahead
		// saves.  In the last interval the read-ahead fetches a slot nobody needs (its registers are dead).
		auto main_loop = [&](auto PRE_) {
			constexpr bool PRE = decltype(PRE_)::value;
			int cib = 0, pi = 0;
			for (int sidx = 0; sidx < P; ++sidx) {
				const int t0_ = 2 * pi, nt = min(2, p.K - t0_);
				int cib1 = cib, pi1 = pi + 1;  // interval sidx + 1
				if (pi1 == npb) { pi1 = 0; ++cib1; }
				const unsigned xs = xoff(cib), xs1 = xoff(cib1);
				STAMPI(ta)
				STAMPI(tb)
				load_frag(xs, e_slot(sidx), t0_, 1, f1);
				mma_frag(f0);
				interleave();
				if (nt == 2) {
					load_frag(xs, o_slot(sidx), t0_ + 1, 0, f0);
					mma_frag(f1);
					interleave();
					load_frag(xs, o_slot(sidx), t0_ + 1, 1, f1);
					mma_frag(f0);
					interleave();
				}
				if (PRE) load_frag(xs1, e_slot(sidx + 1), 2 * pi1, 0, f0);
				mma_frag(f1);
				if (PRE) interleave();
				STAMPI(tc)
				STAMPI(td)
				__builtin_amdgcn_s_barrier();
				if (!PRE) load_frag(xs1, e_slot(sidx + 1), 2 * pi1, 0, f0);
				STAMPI(te)
#ifdef CONVASR_STAMPS
				acc_issue += tb - ta; acc_work += tc - tb; acc_vm += td - tc; acc_bar += te - td;
#endif
				cib = cib1; pi = pi1;
			}
		};
		if (npb > 1 || k1) main_loop(std::true_type()); else main_loop(std::false_type());  // K = 2: every interval opens a new slab, whose X rows land only at this interval's barrier: no read-ahead
	}

#ifdef CONVASR_STAMPS
	unsigned long long t_epi0 = 0;
	STAMPI(t_epi0)
#endif
	// ---------------- epilogue: C/D layout of 16x16 blocks: col = lane & 15, row = (lane >> 4) * 4 + reg
	constexpr int OPITCH = EPI_OPITCH;
	char* const otile = smem;
	float* const red = reinterpret_cast<float*>(smem + BM_ * OPITCH);  // [2][4 (wm)][BN_]
	const int nvalid = valid_len(p.xlen, b, p.Tout);
	const ActConst ac = act_const(p.act, p.act_lo, p.act_hi), bn_ac = act_const(p.bn_act, p.bn_lo, p.bn_hi);
	// fused BN-backward epilogue (see below): the consumer layer's y tile is fetched now, 16 B per lane and store-loop trip, so
	// that its latency hides under the accumulator staging (loading it inside the store loop cost ~8 serial L2/HBM round trips per tile)
	constexpr int OEPC_ = 16 / sizeof(O), OCH_ = BN_ / OEPC_, TRIPS = BM_ * OCH_ / V2_THREADS;
	static_assert(BM_ * OCH_ % V2_THREADS == 0, "the store loop covers the tile in whole trips");
	constexpr bool bnf = BNF == 1 && sizeof(O) == 2;  // separate instantiations: the plain launches do not carry the epilogue's registers and code
	constexpr bool bnm = BNF == 2 && sizeof(O) == 2;  // form 2: gates + matrix-pipe sums
	typedef typename std::conditional<sizeof(O) == 2, O, I>::type H;  // the 16-bit storage type of the fused epilogue's operands (= O there)
	uint4 ypre[bnf ? TRIPS : 1];
	unsigned gpre[TRIPS];  // the chunk's eight one-bit gradient gates (when the forward pass stored them: bn_gate)
	if (bnm) {
#pragma unroll
		for (int i = 0; i < TRIPS; ++i) {
			const int e = tid + i * V2_THREADS, row = e / OCH_, t = t0 + row, co = co0 + (e % OCH_) * OEPC_;
			gpre[i] = (t < p.Tout && co < p.Cout) ? p.bn_gate[(((int64_t)b * p.Tout + t) * p.Cout + co) >> 3] : 0u;  // (a frame past the utterance, a masked frame: no gradient passes)
		}
		// gate byte -> the four dword masks of its 8-element chunk (element 2 i in the low half of word i), once per tile
		if (tid < 256) {
			unsigned m[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) m[i] = (((tid >> (2 * i)) & 1) ? 0x0000FFFFu : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0xFFFF0000u : 0u);
			*reinterpret_cast<uint4*>(smem + EPI_LUT + tid * 16) = make_uint4(m[0], m[1], m[2], m[3]);
		}
	}
	if (bnf) {
#pragma unroll
		for (int i = 0; i < TRIPS; ++i) {
			const int e = tid + i * V2_THREADS, row = e / OCH_, t = t0 + row, co = co0 + (e % OCH_) * OEPC_;
			ypre[i] = make_uint4(0, 0, 0, 0);
			gpre[i] = 0;
			if (t < p.Tout && co < p.Cout) {
				const int64_t idx = ((int64_t)b * p.Tout + t) * p.Cout + co;
				ypre[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const H*>(p.bn_y) + idx);
				if (p.bn_gate) gpre[i] = p.bn_gate[idx >> 3];
			}
		}
	}
	// Accumulators -> bf16 / fp32 output tile in LDS (+ the BN statistics of a forward launch).  64 values per lane: this phase is
	// VALU-issue bound (in-kernel stamps: ~6,500 cycles per tile in its general form), so the launches of a training step -- no bias,
	// no folded scale / shift, no activation, no length mask in the conv itself -- take a specialised instantiation that does
	// nothing but (statistics,) convert and store; tiles that end inside the utterance also drop the row predicate.  Same
	// arithmetic on the same values in the same order: bit-identical to the general form.
	auto stage = [&](auto PLAIN_, auto FULL_, auto STATS_, auto LEAKY_) {
		constexpr bool PLAIN = decltype(PLAIN_)::value, FULL = decltype(FULL_)::value, STATS = decltype(STATS_)::value, LEAKY = decltype(LEAKY_)::value;
#pragma unroll
		for (int ni = 0; ni < NB; ++ni) {
			const int col = wn * (16 * NB) + ni * 16 + r16, co = co0 + col;
			const bool cok = co < p.Cout;
			const float bias = (!PLAIN && p.bias && cok) ? p.bias[co] : 0.f;
			const float sc = (!PLAIN && p.scale && cok) ? p.scale[co] : 1.f, sh = (!PLAIN && p.scale && cok) ? p.shift[co] : 0.f;
			float s1 = 0.f, s2 = 0.f;
#pragma unroll
			for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					const int row = wm * WROWS + mi * 16 + kb * 4 + g;
					const int t = t0 + row;
					float val = acc[mi][ni][g];
					if (!PLAIN) val += bias;
					if (STATS && (FULL || t < p.Tout)) { s1 += val; s2 += val * val; }
					if (!PLAIN) {
						val = LEAKY ? apply_act(val * sc + sh, ac) : apply_clamp(val * sc + sh, ac);  // (not leaky-relu: one v_med3_f32)
						if (t >= nvalid) val = 0.f;
					}
					Elem<O>::store(reinterpret_cast<O*>(otile + row * OPITCH) + col, val);
				}
			}
			if (STATS) {
				s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
				s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
				if (kb == 0) { red[(0 * 4 + wm) * BN_ + col] = s1; red[(1 * 4 + wm) * BN_ + col] = s2; }
			}
		}
	};
	{
		typedef std::true_type Y;
		typedef std::false_type N;
		const bool plain = !p.bias && !p.scale && p.act == CONVASR_ACT_NONE && !p.xlen, full = t0 + BM_ <= p.Tout;
		if (!plain) {
			if (ac.leaky) { if (p.stats) stage(N(), N(), Y(), Y()); else stage(N(), N(), N(), Y()); }
			else { if (p.stats) stage(N(), N(), Y(), N()); else stage(N(), N(), N(), N()); }
		}
		else if (p.stats) { if (full) stage(Y(), Y(), Y(), N()); else stage(Y(), N(), Y(), N()); }
		else stage(Y(), Y(), N(), N());
	}
#ifdef CONVASR_STAMPS
	unsigned long long t_e1 = 0, t_e2 = 0, t_e3 = 0;
	STAMPI(t_e1)
#endif
	__syncthreads();
	STAMPI(t_e2)
	if (p.stats && tid < BN_ && co0 + tid < p.Cout) {
		double a = 0, q2 = 0;
#pragma unroll
		for (int m = 0; m < 4; ++m) { a += (double)red[(0 * 4 + m) * BN_ + tid]; q2 += (double)red[(1 * 4 + m) * BN_ + tid]; }
		double* const prow = p.stats + (int64_t)mtile * 2 * p.Cout;  // per-(m tile) partial row, summed in a fixed order by bn_finalize
		prow[co0 + tid] = a;
		prow[p.Cout + co0 + tid] = q2;
	}
	STAMPI(t_e3)
	O* const yb = reinterpret_cast<O*>(p.y) + (SK ? (int64_t)split * p.split_stride : 0) + (int64_t)b * p.Tout * p.Cout;
	constexpr int OEPC = 16 / sizeof(O), OCHUNKS = BN_ / OEPC;
	const bool vec_ok = ((p.Cout * sizeof(O)) & 15) == 0;
	// Fused pass 1 of the consumer layer's batch-norm backward (bf16 only): this tile IS dz of that layer; with its conv output y
	// (same coordinates) g = dz * act'(y * scale + shift) * dropout * mask, and the tile's per-channel sums of g and g * xhat go to
	// per-tile fp64 partial rows -- the separate reduce pass (2 reads of B*T*C) and its launch disappear.  A thread keeps one 8-channel
	// chunk for all its rows (V2_THREADS % OCHUNKS == 0), reads dz back from the LDS tile exactly as it is stored (bf16-rounded).
	float bs1[8], bs2[8], bsc[8], bsh[8], bmean[8], bistd[8];
	const int bco = co0 + (tid % OCHUNKS) * OEPC;
	int bnv = 0;
	if (bnf) {
#pragma unroll
		for (int k = 0; k < 8; ++k) bs1[k] = bs2[k] = 0.f;
		if (bco < p.Cout) { load8<float>(p.bn_scale + bco, bsc); load8<float>(p.bn_shift + bco, bsh); load8<float>(p.bn_mean + bco, bmean); load8<float>(p.bn_invstd + bco, bistd); }
		bnv = valid_len(p.bn_xlen, b, p.Tout);
	}
	const float gate_scale = p.bn_drop_thr ? p.bn_keep_scale : 1.f;
#pragma unroll
	for (int it = 0; it < TRIPS; ++it) {
		const int e = tid + it * V2_THREADS;
		const int row = e / OCHUNKS, ch = e % OCHUNKS;
		const int t = t0 + row, co = co0 + ch * OEPC;
		if (bnm) {  // form 2: the chunk, masked by its gates, goes back into the tile in place: G = gate ? dz : 0 (exact: no arithmetic on the values)
			uint4* const cell = reinterpret_cast<uint4*>(otile + row * OPITCH + ch * 16);
			const uint4 dzv = *cell, m = *reinterpret_cast<const uint4*>(smem + EPI_LUT + gpre[it] * 16);
			if (t < p.Tout && co < p.Cout) {
				O* dst = yb + (int64_t)t * p.Cout + co;
				if (vec_ok && co + OEPC <= p.Cout) *reinterpret_cast<uint4*>(dst) = dzv;
				else
					for (int i = 0; i < OEPC && co + i < p.Cout; ++i) dst[i] = reinterpret_cast<const O*>(cell)[i];
			}
			*cell = make_uint4(dzv.x & m.x, dzv.y & m.y, dzv.z & m.z, dzv.w & m.w);
			continue;
		}
		if (t >= p.Tout || co >= p.Cout) continue;
		const O* src = reinterpret_cast<const O*>(otile + row * OPITCH) + ch * OEPC;
		O* dst = yb + (int64_t)t * p.Cout + co;
		if (vec_ok && co + OEPC <= p.Cout) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
		else
			for (int i = 0; i < OEPC && co + i < p.Cout; ++i) dst[i] = src[i];
		if (bnf && t < bnv) {
			const int64_t idx = ((int64_t)b * p.Tout + t) * p.Cout + co;
			float dz[8], yv[8], g[8];
			unpack16<H>(*reinterpret_cast<const uint4*>(src), dz);
			unpack16<H>(ypre[it], yv);
			if (p.bn_gate) {  // (workgroup-uniform) the stored gates replace act', the dropout hash and the frame mask: g = dz * keep_scale or 0
#pragma unroll
				for (int k = 0; k < 8; ++k) g[k] = ((gpre[it] >> k) & 1u) ? dz[k] * gate_scale : 0.f;
			} else {
#pragma unroll
			for (int k = 0; k < 8; ++k) g[k] = dz[k] * act_grad(fmaf(yv[k], bsc[k], bsh[k]), bn_ac);
			if (p.bn_drop_thr) {
				float keep[8];
				dropout_mask8(p.bn_seed ^ (p.bn_step_key ? *p.bn_step_key : 0ull), p.bn_offset, p.bn_drop_thr, p.bn_keep_scale, idx, keep);
#pragma unroll
				for (int k = 0; k < 8; ++k) g[k] *= keep[k];
			}
			}
#pragma unroll
			for (int k = 0; k < 8; ++k) { bs1[k] += g[k]; bs2[k] = fmaf(g[k], yv[k] - bmean[k], bs2[k]); }  // sum g * (y - mean): centred per element (no cancellation for |mean| >> std), scaled once per tile below
		}
	}
	if (bnf) {
		float* const bnred = reinterpret_cast<float*>(smem + BM_ * OPITCH + 8 * BN * sizeof(float));  // [V2_THREADS][17], past the output tile and `red`
#pragma unroll
		for (int k = 0; k < 8; ++k) { bnred[tid * 17 + k] = bs1[k]; bnred[tid * 17 + 8 + k] = bs2[k] * bistd[k]; }  // sum g * xhat of this thread's rows
		__syncthreads();
		if (tid < BN_ && co0 + tid < p.Cout) {
			const int chunk = tid >> 3, k = tid & 7;
			double a = 0, q2 = 0;
			for (int j = chunk; j < V2_THREADS; j += OCHUNKS) { a += (double)bnred[j * 17 + k]; q2 += (double)bnred[j * 17 + 8 + k]; }
			double* const prow = p.bn_sums + (int64_t)mtile * 2 * p.Cout;  // per-(m tile) partial row, summed by convasr_bn_bwd_finalize
			prow[co0 + tid] = a;
			prow[p.Cout + co0 + tid] = q2;
		}
	}
	if (bnm) {
		// Fused pass 1 of the consumer layer's batch-norm backward, form 2.  With the stored gates g = gate ? dz / (1 - p) : 0, so
		//   sum_t g = k sum_t G[t][c],   sum_t g y = k sum_t G[t][c] Y[t][c] = k diag(G^T Y)[c]       (k = 1 / (1 - p), G = gated dz)
		// are two small matrix products over the tile's rows -- G^T 1 and the diagonal blocks of G^T Y -- which the matrix pipe,
		// idle in an epilogue, does in 2 x 8 MFMAs per 16-channel block: wave w takes channels [16 w, 16 w + 16), operand fragments are
		// column reads (ds_read_b64_tr_b16) of the two row-major tiles.  Products of two 16-bit values are exact in fp32 and the sums
		// are fp32 chains over the tile's 256 rows, like the vector-ALU form's.  That form spent ~56 vector instructions per 8 elements
		// (unpack, gate, two accumulations) and ~12.4 K cycles per tile; this one masks a chunk with four v_and.
		__syncthreads();  // G complete; the loaders have waited for the y tile's DMA before arriving here
		typedef __attribute__((address_space(3))) s16x4* lp;
		typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
		auto tr8 = [](unsigned addr, unsigned row_stride) {
			const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(size_t)addr);
			const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(size_t)(addr + 4 * row_stride));
			const uint2 l = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
			u32x4_ r; r[0] = l.x; r[1] = l.y; r[2] = h2.x; r[3] = h2.y;
			return r;
		};
		if (wave < BN_ / 16) {
			// 16-lane group g4 = the MFMA operand's k block (8 rows); lane 4 q + pc of the group supplies row q, columns 4 pc .. 4 pc + 3
			const int g4 = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
			const unsigned ga = lds_base + (8 * g4 + q) * OPITCH + (16 * wave + 4 * pc) * 2;
			const unsigned ya = lds_base + EPI_YOFF + (8 * g4 + q) * (BN_ * 2) + (16 * wave + 4 * pc) * 2;
			u32x4_ ones; ones[0] = ones[1] = ones[2] = ones[3] = HalfOnes<O>::pair;
			f32x4 a1 = f32x4{0.f, 0.f, 0.f, 0.f}, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < BM_ / 32; ++ks) {
				const u32x4_ gf = tr8(ga + ks * 32 * OPITCH, OPITCH), yf = tr8(ya + ks * 32 * (BN_ * 2), BN_ * 2);
				a1 = Mma16<H>::run(gf, ones, a1);
				a2 = Mma16<H>::run(gf, yf, a2);
			}
			// D[i][j] sits in lane (j, i >> 2), register i & 3: the diagonal in the 16 lanes with (lane >> 4) == ((lane & 15) >> 2)
			const int c16 = lane & 15, co = co0 + 16 * wave + c16;
			if ((lane >> 4) == (c16 >> 2) && co < p.Cout) {
				const int r = c16 & 3;
				const float s1 = (r == 0 ? a1[0] : r == 1 ? a1[1] : r == 2 ? a1[2] : a1[3]) * gate_scale;
				const float s2 = (r == 0 ? a2[0] : r == 1 ? a2[1] : r == 2 ? a2[2] : a2[3]) * gate_scale;
				double* const prow = p.bn_sums + (int64_t)mtile * 2 * p.Cout;  // per-(m tile) partial row, summed by convasr_bn_bwd_finalize
				prow[co] = (double)s1;
				prow[p.Cout + co] = (double)((s2 - p.bn_mean[co] * s1) * p.bn_invstd[co]);  // sum g * xhat
			}
		}
	}
#ifdef CONVASR_STAMPS
	STAMP(t_end)
	if (blockIdx.x < 256 && lane == 0) {
		unsigned long long* o = g_v2s_stamps + (blockIdx.x * 8 + wave) * 8;
		o[0] = t_loop0 - t_start; o[1] = t_e1 - t_epi0; o[2] = acc_work; o[3] = __builtin_amdgcn_s_memrealtime() - rt_start; o[4] = acc_bar; o[5] = CONVASR_STAMPS == 2 ? rt_start : t_e3 - t_e2; o[6] = t_end - t_e3; o[7] = t_end - t_start;
	}
#endif
}

template <typename I, typename O, int BNF, int BM_, bool SK = false> __global__ __launch_bounds__(V2S_THREADS, 3) void conv1d_igemm_v2s_kernel(ConvParams p) {

	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int bid = blockIdx.x;
	int v, half = 0;
	const bool narrow = bid >= p.full_tiles;
	if (!narrow) v = xcd_remap(bid, p.full_tiles);
	else {  // full_tiles is a multiple of 8, so (bid - full_tiles) keeps the workgroup's XCD; the two halves of a tile share an XCD (and its X tile in L2)
		const int h = xcd_remap(bid - p.full_tiles, 2 * (p.total_tiles - p.full_tiles));
		v = p.full_tiles + (h >> 1);
		half = h & 1;
	}
	int ntile, mtile;
#ifdef CONVASR_AB_BLOCK
	{
		const int sel = (p.debug >> 10) & 7;  // m x n tiles an XCD's ~32 resident workgroups cover: 16x2 (shipped), 8x4, 32x1, 11x3, 16x3, 4x8, 16x6, 6x6
		const int MBs[8] = {16, 8, 32, 11, 16, 4, 16, 6}, NTs[8] = {2, 4, 1, 3, 3, 8, 6, 6};
		tile_coords_rt(v, p.B * p.m_tiles_per_b, p.n_tiles, MBs[sel], NTs[sel], mtile, ntile);
	}
#else
	tile_coords(v, p.B * p.m_tiles_per_b, p.n_tiles, mtile, ntile);
#endif
	// Short last tile: an utterance of 626 frames is 2 x 256 + 114 -- run as a third 256-row tile the tail costs a full tile of MFMAs for
	// 114 useful rows (mixed-length batches, BASELINE configs[4]: ~8 % of the forward / dgrad time on average); as a 128-row tile (two
	// 16-row blocks per wave instead of four, same k order per element: bit-identical values) it costs half.
	// tail128 == 2: EVERY m tile is 128 rows (m tiles 128 frames apart): launches whose 256-row tiles would occupy at most a quarter of the
	// CUs get twice the workgroups (the dispatcher says when, with the measurement behind it).
	const bool all_short = p.tail128 == 2;
	const bool tail = all_short || (p.tail128 && (mtile % p.m_tiles_per_b) == p.m_tiles_per_b - 1);
	const int bm_full = all_short ? 128 : BM_;
	const int split = SK ? blockIdx.y : 0;  // (a split-K launch: grid.y = number of splits)
	if (!tail) {
		if (!narrow) v2s_tile<I, O, 4, BNF, BM_, SK>(p, smem, mtile, ntile, 0, BM_, split);
		else v2s_tile<I, O, 2, BNF, BM_, SK>(p, smem, mtile, ntile, half, BM_, split);
	} else {
		if (!narrow) v2s_tile<I, O, 4, BNF, 128, SK>(p, smem, mtile, ntile, 0, bm_full, split);
		else v2s_tile<I, O, 2, BNF, 128, SK>(p, smem, mtile, ntile, half, bm_full, split);
	}
}

// Returns 1 if the LDS-DMA kernel took the launch, 0 if the shape is outside its envelope (the caller falls back to conv.hip's
// register-staged kernel).  x_dtype: CONVASR_BF16 or CONVASR_F16; y_dtype: the same, or CONVASR_F32 (the decoder head).
template <typename I, int BM_> static const void* v2s_kernel(int ki) {
	if (ki == 0) return (const void*)conv1d_igemm_v2s_kernel<I, I, 0, BM_>;
	if (ki == 1) return (const void*)conv1d_igemm_v2s_kernel<I, I, 1, BM_>;
	if (ki == 3) return (const void*)conv1d_igemm_v2s_kernel<I, I, 2, BM_>;
	if (ki == 4) return (const void*)conv1d_igemm_v2s_kernel<I, float, 0, BM_, true>;  // the split-K launches of small-batch inference (fp32 partial tiles)
	return (const void*)conv1d_igemm_v2s_kernel<I, float, 0, BM_>;
}

// Returns 1 if the LDS-DMA kernel took the launch, 0 if the shape is outside its envelope (the caller falls back to conv.hip's
// register-staged kernel).  x_dtype: CONVASR_BF16 or CONVASR_F16; y_dtype: the same, or CONVASR_F32 (the decoder head).
int convasr_conv1d_v2_try(ConvParams p, int x_dtype, int y_dtype, hipStream_t s, int* m_tiles_out) {
	if (p.stride != 1 || (p.Cin & 63) != 0 || !convasr_is_half(x_dtype) || (y_dtype != x_dtype && y_dtype != CONVASR_F32)) return 0;
	const int n_cu = convasr_cu_count();  // (per device ordinal: common.h)
	// Tile height: 256 rows.  A 192-row build of the same kernel (BM_ = 192: 256 x n tiles for 64 x 751 frames = whole rounds on 256
	// CUs instead of 1.5 n) was measured layer by layer in one process (scratch/ab_bm.py, profiles/r03_ab_tile_height.json): 3-5 %
	// faster only on the 384-channel layers (2.25 rounds), 5-8 % SLOWER on every layer from 512 channels up -- a partial last round costs
	// about its fraction of a tile time on this chip (640 channels: 3.75 rounds take 3.77 tile times; the CUs left idle give their power
	// to the busy ones), so there is no quantisation loss to recover, and 12 MFMAs per 7 fragment reads lose to 16 per 8.  The 192-row
	// instantiations exist only in a diagnostic build (python -m convasr_amd.build --variant tile192 -DCONVASR_AB_TILE192=1), where
	// debug bit 64 selects them.
#ifdef CONVASR_AB_TILE192
	int bm = (p.debug & 64) ? 192 : V2_BM;
#else
	int bm = V2_BM;
#endif
	// 128-row tiles for the whole launch (ConvParams::tail128 == 2, see the kernel): only when 256-row tiles would occupy at most a quarter
	// of the CUs.  Measured on JasperNetLarge's shapes at 32 x 251-876 frames (profiles/r04_ab_short_tiles.json, scratch/ab_short_tiles.py):
	// a 128-row tile costs 1.2-1.4x per FLOP (twice the weight bytes into LDS per row, 6 fragment reads per 8 MFMAs), so even launches
	// of 96-192 tiles on 256 CUs are faster with 256-row tiles; 64 tiles (32 x 251 frames, 256 channels) are not: 0.83x.
	// debug bit 1024 forces it, bit 2048 forbids it (A/B runs).
	bool all_short = false;
	if (bm == V2_BM && !(p.debug & 2048)) {
		const int tiles256 = p.B * ((p.Tout + V2_BM - 1) / V2_BM) * p.n_tiles;
		all_short = (p.debug & 1024) || tiles256 * 4 <= n_cu;
	}
	if (all_short) bm = 128;
	const int xr = (bm - 1) + (p.K - 1) * p.dil + 1;
	p.x_rows = (xr + 15) & ~15;  // whole 1-KiB pieces and an even number of them per 16-row swizzle period
	const size_t osz = y_dtype == CONVASR_F32 ? 4 : 2;
	size_t smem = 2 * (size_t)p.x_rows * ROW_BYTES + 5 * V2_WSLOT;  // two X slab buffers + the 3 + 2 weight slots
	if (p.K == 1) smem = 3 * (size_t)p.x_rows * ROW_BYTES + 3 * V2_WSLOT;  // K = 1: three X slab buffers + the 3-slot ring
	const bool mfma_sums = p.bn_y && p.bn_gate && !(p.debug & 256);  // (debug bit 256: the vector-ALU form of the fused epilogue even with gates: A/B runs)
	const size_t epi = (size_t)bm * (BN * osz + 16) + 8 * BN * sizeof(float) + (p.bn_y ? (mfma_sums ? (size_t)bm * BN * 2 + 4096 : (size_t)V2_THREADS * 17 * sizeof(float)) : 0);
	if (epi > smem) smem = epi;
	if (smem > 160 * 1024) return 0;
	if ((int64_t)p.Tin * p.Cin * 2 >= (1ll << 31) || (int64_t)p.K * p.CoutPad * p.Cin * 2 >= (1ll << 31) || (int64_t)(p.Tout + bm) * p.Cout * 2 >= (1ll << 31)) return 0;
	p.m_tiles_per_b = (p.Tout + bm - 1) / bm;
	p.total_tiles = p.B * p.m_tiles_per_b * p.n_tiles;
	const int tail_rows = p.Tout - (p.m_tiles_per_b - 1) * bm;
	p.tail128 = all_short ? 2 : ((bm == V2_BM && tail_rows <= 128 && !(p.debug & 128)) ? 1 : 0);  // (debug bit 128: short tails off, A/B runs)
	const bool f16 = x_dtype == CONVASR_F16, wide = y_dtype == CONVASR_F32, fused = p.bn_y != nullptr;
	if (fused && wide) return 0;  // the fused BN-backward epilogue reads dz back in the storage type
	if (p.cib_per_split && !wide) return 0;  // (split-K partial tiles are fp32)
	const int ki = p.cib_per_split ? 4 : (wide ? 2 : (fused ? (mfma_sums ? 3 : 1) : 0)), bi = bm == 192 ? 1 : 0;  // (bm == 128 runs the 256-row kernel's 128-row instantiation)
#ifdef CONVASR_AB_TILE192
	const void* kern = f16 ? (bi ? v2s_kernel<f16_t, 192>(ki) : v2s_kernel<f16_t, V2_BM>(ki)) : (bi ? v2s_kernel<bf16_t, 192>(ki) : v2s_kernel<bf16_t, V2_BM>(ki));
#else
	const void* kern = f16 ? v2s_kernel<f16_t, V2_BM>(ki) : v2s_kernel<bf16_t, V2_BM>(ki);
#endif
	static unsigned long long attr_set[2][2][5] = {};
	convasr_allow_160k_lds(kern, attr_set[f16][bi][ki]);
	// a last partial round that would occupy at most half of the CUs is cut into half-width tiles (debug bit 32: off)
	p.full_tiles = p.total_tiles;
	if (!(p.debug & 32)) {
		const int rest = p.total_tiles % n_cu;
		if (rest > 0 && 2 * rest <= n_cu && ((p.total_tiles - rest) & 7) == 0) p.full_tiles = p.total_tiles - rest;
	}
	const int grid = p.full_tiles + 2 * (p.total_tiles - p.full_tiles);
	const int splits = p.cib_per_split ? ((p.Cin >> 6) + p.cib_per_split - 1) / p.cib_per_split : 1;
	void* args[] = {&p};
	if (hipLaunchKernel(kern, dim3(grid, splits), dim3(V2S_THREADS), args, smem, s) != hipSuccess) { (void)hipGetLastError(); return 0; }  // (the caller's fallback starts from a clean error state)
	if (m_tiles_out) *m_tiles_out = p.B * p.m_tiles_per_b;
	return 1;
}
