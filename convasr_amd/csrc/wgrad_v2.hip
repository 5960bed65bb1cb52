// 16-bit (bf16 / fp16 storage) fast path of the weight gradient (stride 1, Cin % 128 == 0, Cout % 128 == 0).
//
//   dW[tap][co][ci] = sum_(b,t) dY[b,t,co] * X[b, t + tap*dil - pad, ci]        M = co, N = ci, reduction over (b, t)
//
// One workgroup (8 waves) = one 128(co) x 128(ci) tile x up to 4 taps over one split of the (b, t) axis.  Per 64-frame chunk
// the dY rows and the X rows (+ tap halo) are brought in by LDS-DMA into a 4-deep ring (issued three chunks ahead, counted
// vmcnt, raw s_barrier); both MFMA operands are column reads of row-major [t][c] tiles, served by ds_read_b64_tr_b16, with
// the 64-byte blocks of each 256-byte row XOR-swizzled by (row & 3) so the four rows a transposed read touches fall on
// different banks (the swizzle lives in the DMA's per-lane source address).  Each wave owns a 64 x 64 sub-tile and two
// accumulator slots (128 accumulator registers), two waves per SIMD (one of waves 0-3, one of waves 4-7).  A full group gives
// waves 0-3 taps {0,1} and waves 4-7 taps {2,3}; a ragged last group (K mod 4 taps) is still split evenly: with 3 taps each half
// takes one whole tap plus half of the third one's frames (k-substeps {0,1} / {2,3} of every chunk), with 2 taps one each, with
// 1 tap half of its frames each -- the two partial sums of a shared tap meet in LDS in the epilogue.  A group's time is then
// proportional to its tap count (K = 11: 2.75 group-times instead of 3; K = 29: 7.25 instead of 8).
// Zero padding of the conv and ragged chunk ends are the per-utterance buffer descriptors' range checks.
// (A 16x16x32-MFMA build of this kernel -- 32-byte-unit swizzle, 223 VGPRs -- is bit-identical and measured 3-5 % SLOWER in the
// same process, the opposite of the forward kernel where that shape wins 3-7 %: kept on 32x32x16.)
#include "conv_common.h"
#include <type_traits>

#define W2_THREADS 512             // computing waves 0-7
#define W2_ALL_THREADS (512 + 256)  // + loader waves 8-11 (one per SIMD), which only issue the LDS-DMA pieces: see conv_v2s.hip
#define W2_BKT 64
#define W2_YBYTES (W2_BKT * 256)

typedef int v4i32 __attribute__((ext_vector_type(4)));

#ifdef CONVASR_STAMPS
// diagnostic build only (see conv_v2s.hip): per-wave cycle sums of the chunk loop's segments
__device__ unsigned long long g_w2_stamps[256 * 8 * 8];
extern "C" int convasr_debug_read_wgrad_stamps(unsigned long long* host, int count) {
	return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w2_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#define WSTAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define WSTAMP(var)
#endif

__device__ __forceinline__ v4i32 w2_make_srd(const char* base, unsigned num_bytes) {
	const unsigned long long a = (unsigned long long)base;
	v4i32 d;
	d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
	d[2] = __builtin_amdgcn_readfirstlane((int)num_bytes);
	d[3] = 0x00020000;
	return d;
}

__device__ __forceinline__ void w2_dma16(const v4i32& srd, unsigned lds_addr, int voff) {
	unsigned keep;
	asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(voff), "s"(srd), "s"(lds_addr)
	             : "memory");
}

// two transposed 4x16 reads -> the 8 consecutive-k bf16 values of one MFMA operand fragment
__device__ __forceinline__ uint4 w2_tr_frag(const char* p0, const char* p1) {
	typedef __attribute__((address_space(3))) s16x4* lp;
	const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p0);
	const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p1);
	const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
	return make_uint4(l.x, l.y, h.x, h.y);
}

// Grouped launches: up to W2_MAX_GROUP independent weight-gradient problems (the one-tap residual branches of a dense block: same frames,
// their own dY / X / slabs) in one dispatch; the workgroup index picks the problem by a prefix sum over units x splits.
#define W2_MAX_GROUP 12
struct WgradGroup {
	WgradParams prob[W2_MAX_GROUP];
	int first[W2_MAX_GROUP + 1];
	int n;
};

template <typename H> __device__ __forceinline__ void wgrad_v2_body(const WgradParams& p, const int v, char* const smem) {
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int tp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

	const int unit = v % p.units, split = v / p.units;
#ifdef CONVASR_AB_WGRAD_ORDER
	// Diagnostic build only (scratch/ab_wgrad_order.py): which (co tile, ci tile, tap group) units run side by side on an XCD decides how
	// often a split's dY / X rows are fetched from beyond L2.  Order 0 = shipped (tap group fastest, then ci, then co: ~32 resident units =
	// 2 co x all ci x all tap groups); 1 = co fastest, then ci, then tap group; 2 = 3 x 3 (co x ci) blocks with all tap groups inside.
	int tg, ci_t, co_t;
	{
		const int order = (p.debug >> 13) & 3;
		if (order == 1) { co_t = unit % p.co_tiles; ci_t = (unit / p.co_tiles) % p.ci_tiles; tg = unit / (p.co_tiles * p.ci_tiles); }
		else if (order == 2) {
			const int BS = 3, cb = (p.co_tiles + BS - 1) / BS, ib = (p.ci_tiles + BS - 1) / BS;
			// walk blocks of up to 3 x 3 (co, ci) pairs; inside a block the tap groups are fastest.  Units beyond the ragged edge are
			// remapped by a linear search over the block list (few hundred units at most).
			int u = unit, found = 0; tg = ci_t = co_t = 0;
			for (int b0 = 0; b0 < cb * ib && !found; ++b0) {
				const int c0 = (b0 / ib) * BS, i0 = (b0 % ib) * BS;
				const int nc = min(BS, p.co_tiles - c0), ni = min(BS, p.ci_tiles - i0), n = nc * ni * p.tap_groups;
				if (u < n) { tg = u % p.tap_groups; const int r = u / p.tap_groups; ci_t = i0 + r % ni; co_t = c0 + r / ni; found = 1; }
				else u -= n;
			}
		}
		else { tg = unit % p.tap_groups; ci_t = (unit / p.tap_groups) % p.ci_tiles; co_t = unit / (p.tap_groups * p.ci_tiles); }
	}
#else
	const int tg = unit % p.tap_groups, ci_t = (unit / p.tap_groups) % p.ci_tiles, co_t = unit / (p.tap_groups * p.ci_tiles);
#endif
	// (Balanced tap groups -- K = 29 cut 5 x 4 + 3 x 3 instead of 7 x 4 + 1, K = 13 cut 4 + 3 + 3 + 3 -- were measured in round 5 and lost: a chunk's
	// DMA and barrier cost is the same whatever the group's tap count, so a three-tap group costs ~0.85 of a full one, not 0.75, and the
	// one-tap group only ~0.4: the K = 29 launch of the bench step went from 7.4 to 7.55 group-times, +38 us, profiles/r05_ab_rounds_wav2letter.json.)
	const int co0 = co_t * 128, ci0 = ci_t * 128, tap0 = tg * WG_TG;
	const int c_begin = split * p.chunks_per_split, c_end = min(p.total_chunks, c_begin + p.chunks_per_split);
	// slot A / slot B of this wave: tap index, mask of the k-substeps (of 4 per chunk) it covers, shared with the other wave half?
	const int ntaps = min(WG_TG, p.K - tap0);
	int tapA, tapB;
	unsigned mA, mB;
	bool shA = false, shB = false;
	if (ntaps == 4) { tapA = tap0 + 2 * tp; tapB = tapA + 1; mA = 15u; mB = 15u; }
	else if (ntaps == 3) { tapA = tap0 + tp; tapB = tap0 + 2; mA = 15u; mB = tp ? 12u : 3u; shB = true; }
	else if (ntaps == 2) { tapA = tap0 + tp; tapB = tapA; mA = 15u; mB = 0u; }
	else { tapA = tap0; tapB = tap0; mA = tp ? 12u : 3u; mB = 0u; shA = true; }

	const int xbytes = p.x_rows * 256;  // x_rows is a multiple of 4: whole 1-KiB pieces
	const int stage_bytes = W2_YBYTES + xbytes;
	const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
	const int y_row_bytes = (p.dy_ld ? p.dy_ld : p.Cout) * 2, x_row_bytes = (p.x_ld ? p.x_ld : p.Cin) * 2;
	const int y_pieces = W2_BKT / 4, x_pieces = p.x_rows >> 2, pieces = y_pieces + x_pieces;
	const bool loader = wave >= 8;
	const int lw = wave - 8;
	const int my_pieces = (pieces - lw + 3) >> 2;  // loader wave lw issues pieces j = lw + 4 i
	// per-lane source offset inside a 4-row piece: row (lane >> 4), 16-byte chunk (lane & 15) ^ (row << 2)
	const int prow = lane >> 4, pchunk = (lane & 15) ^ (prow << 2);
	const int ylane = prow * y_row_bytes + pchunk * 16 + co0 * 2, xlane = prow * x_row_bytes + pchunk * 16 + ci0 * 2;

	auto issue = [&](int c, int stage) {
		const int b = c / p.chunks_per_b, t0 = (c % p.chunks_per_b) * W2_BKT;
		const v4i32 ysrd = w2_make_srd(reinterpret_cast<const char*>(p.dy) + (int64_t)b * p.Tout * y_row_bytes, (unsigned)(p.Tout * y_row_bytes));
		const v4i32 xsrd = w2_make_srd(reinterpret_cast<const char*>(p.x) + (int64_t)b * p.Tin * x_row_bytes, (unsigned)(p.Tin * x_row_bytes));
		const unsigned dst = lds_base + stage * stage_bytes;
		const int tin0 = t0 - p.pad + tap0 * p.dil;
		for (int j = lw; j < pieces; j += 4) {
			if (j < y_pieces) w2_dma16(ysrd, __builtin_amdgcn_readfirstlane(dst + j * 1024), (t0 + j * 4) * y_row_bytes + ylane);
			else w2_dma16(xsrd, __builtin_amdgcn_readfirstlane(dst + W2_YBYTES + (j - y_pieces) * 1024), (tin0 + (j - y_pieces) * 4) * x_row_bytes + xlane);
		}
	};

	f32x16 acc[2][2][2];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int k = 0; k < 16; ++k) acc[a][i][j][k] = 0.f;

	// 4-stage ring, chunks issued THREE ahead: chunk c + 1 is already published while chunk c runs, so the first fragments
	// of the next chunk are read before the barrier that ends this one (same software pipeline as conv_v2.hip).
	if (loader) {
		for (int i = 0; i < 3; ++i)
			if (c_begin + i < c_end) issue(c_begin + i, i);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		int stage = 0;
		for (int c = c_begin; c < c_end; ++c) {
			const bool more = c + 3 < c_end;
			if (more) issue(c + 3, (stage + 3) & 3);
			// leave only the pieces issued in this iteration (chunk c + 3) in flight
			if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			else switch (my_pieces) {
				case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
				case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
				case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
				case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
				case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
				case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
				default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
			}
			__builtin_amdgcn_s_barrier();
			stage = (stage + 1) & 3;
		}
		if (ntaps == 3 || ntaps == 1) __builtin_amdgcn_s_barrier();  // the epilogue's barrier of a tap shared by the two wave halves
		return;
	}
	__builtin_amdgcn_s_barrier();
#ifdef CONVASR_STAMPS
	unsigned long long wa = 0, wb = 0, wc = 0, w_work = 0, w_bar = 0, w_t0 = 0, w_t1 = 0, w_t2 = 0;
	WSTAMP(w_t0)
#endif

	// fragment addressing: 16-lane group g4 -> (column block (g4 & 1) * 16, k block 8 * (g4 >> 1)); lane 4q + pc in the group
	// supplies row q, columns 4 pc .. 4 pc + 3 of the 4 x 16 block (ds_read_b64_tr_b16 contract).  Row (r & 3) is q for every
	// dY row a lane touches and (q + tap offset) & 3 for every X row, so the swizzled byte column is a per-lane constant and
	// stepping kk / the +4-row half are pure immediates: no address arithmetic inside the loop.
	const int g4 = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
	const int krow = 8 * (g4 >> 1) + q;
	const int colA = (wm * 64 + (g4 & 1) * 16 + 4 * pc) * 2, colB = (wn * 64 + (g4 & 1) * 16 + 4 * pc) * 2;
	const int offA = (tapA - tap0) * p.dil, offB = (tapB - tap0) * p.dil;
	auto swz = [](int cb, int r3) { return (((cb >> 6) ^ r3) << 6) | (cb & 63); };
	const unsigned aoff0 = krow * 256 + swz(colA, q), aoff1 = krow * 256 + swz(colA + 64, q);
	const unsigned bAoff0 = W2_YBYTES + (krow + offA) * 256 + swz(colB, (q + offA) & 3), bAoff1 = W2_YBYTES + (krow + offA) * 256 + swz(colB + 64, (q + offA) & 3);
	const unsigned bBoff0 = W2_YBYTES + (krow + offB) * 256 + swz(colB, (q + offB) & 3), bBoff1 = W2_YBYTES + (krow + offB) * 256 + swz(colB + 64, (q + offB) & 3);
	struct F2 { uint4 u0, u1; };
	typedef __attribute__((address_space(3))) s16x4* lp;
	auto tr8 = [](unsigned addr) {
		const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(size_t)addr);
		const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(size_t)(addr + 4 * 256));
		const uint2 l = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
		return make_uint4(l.x, l.y, h2.x, h2.y);
	};
	auto load_a = [&](unsigned st, int kk, F2& f) { f.u0 = tr8(st + aoff0 + kk * 4096); f.u1 = tr8(st + aoff1 + kk * 4096); };
	auto load_bA = [&](unsigned st, int kk, F2& f) { f.u0 = tr8(st + bAoff0 + kk * 4096); f.u1 = tr8(st + bAoff1 + kk * 4096); };
	auto load_bB = [&](unsigned st, int kk, F2& f) { f.u0 = tr8(st + bBoff0 + kk * 4096); f.u1 = tr8(st + bBoff1 + kk * 4096); };
	auto mma4 = [&](const F2& a, const F2& bq, f32x16 (&c)[2][2]) {
		Mma<H>::run(a.u0, bq.u0, c[0][0]);
		Mma<H>::run(a.u0, bq.u1, c[0][1]);
		Mma<H>::run(a.u1, bq.u0, c[1][0]);
		Mma<H>::run(a.u1, bq.u1, c[1][1]);
	};

	// The chunk loop is instantiated per (slot A mask, slot B mask) so that every fragment load and MFMA group is unconditional
	// inside it (runtime masks cost 5-14 %: waits at every branch join).
	auto chunk_loop = [&](auto MA_, auto MB_) {
		constexpr unsigned MA = decltype(MA_)::value, MB = decltype(MB_)::value, MAB = MA | MB;
		int stage = 0;
		F2 fa0, fa1, fbA, fbB;
		const unsigned st0 = lds_base;
		if (c_begin < c_end) { if (MAB & 1u) load_a(st0, 0, fa0); if (MA & 1u) load_bA(st0, 0, fbA); }
		for (int c = c_begin; c < c_end; ++c) {
			WSTAMP(wa)
			const unsigned st = lds_base + stage * stage_bytes, stn = lds_base + ((stage + 1) & 3) * stage_bytes;
			// unit (kk, A): MFMAs on (fa, fbA) while the slot-B fragments arrive; unit (kk, B): MFMAs on (fa, fbB) while the next
			// substep's dY and slot-A fragments arrive.  fa0 / fa1 alternate by kk parity.
			// Scheduling hint for the full (four-tap) instantiation, after each substep: the 4 slot-B fragment reads go out two per MFMA
			// from the first MFMA of the slot-A group on, the 8 reads of the next substep two per MFMA of the slot-B group.  Left alone,
			// hipcc sinks reads to one MFMA ahead of their use and the LDS latency is exposed twice per substep (+1.5-2 % per launch;
			// four reads per MFMA gap instead made the fragments spill).
#define W2_DS2_M1 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#define W2_HINT(KK) if (MA == 15u && MB == 15u) { W2_DS2_M1 W2_DS2_M1 __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); W2_DS2_M1 W2_DS2_M1 W2_DS2_M1 W2_DS2_M1 }  // (the same hints on the 3- and 2-tap instantiations: no measurable change)
#define W2_SUBSTEP(KK, FA_CUR, FA_NXT)                                                                                              \
			if ((MB >> KK) & 1u) load_bB(st, KK, fbB);                                                                                  \
			if ((MA >> KK) & 1u) mma4(FA_CUR, fbA, acc[0]);                                                                             \
			if (KK < 3) { if ((MAB >> ((KK + 1) & 3)) & 1u) load_a(st, KK + 1, FA_NXT); if ((MA >> ((KK + 1) & 3)) & 1u) load_bA(st, KK + 1, fbA); } \
			else { if (MAB & 1u) load_a(stn, 0, FA_NXT); if (MA & 1u) load_bA(stn, 0, fbA); }  /* unconditional: after the last chunk it reads a stage nobody needs (a conditional LDS load costs a full lgkmcnt drain at the join) */ \
			if ((MB >> KK) & 1u) mma4(FA_CUR, fbB, acc[1]);                                                                             \
			W2_HINT(KK)
			W2_SUBSTEP(0, fa0, fa1)
			W2_SUBSTEP(1, fa1, fa0)
			W2_SUBSTEP(2, fa0, fa1)
			W2_SUBSTEP(3, fa1, fa0)
#undef W2_SUBSTEP
#undef W2_HINT
#undef W2_DS2_M1
			WSTAMP(wb)
			__builtin_amdgcn_s_barrier();
			WSTAMP(wc)
#ifdef CONVASR_STAMPS
			w_work += wb - wa; w_bar += wc - wb;
#endif
			stage = (stage + 1) & 3;
		}
	};
	typedef std::integral_constant<unsigned, 15u> M15;
	typedef std::integral_constant<unsigned, 12u> M12;
	typedef std::integral_constant<unsigned, 3u> M3;
	typedef std::integral_constant<unsigned, 0u> M0;
	if (ntaps == 4) chunk_loop(M15(), M15());
	else if (ntaps == 3) { if (tp) chunk_loop(M15(), M12()); else chunk_loop(M15(), M3()); }
	else if (ntaps == 2) chunk_loop(M15(), M0());
	else { if (tp) chunk_loop(M12(), M0()); else chunk_loop(M3(), M0()); }

	WSTAMP(w_t1)
	// epilogue: a tap shared by the two wave halves is summed through LDS (the ring is dead after the last barrier), the lower
	// half stores it; [wave & 3][register][lane] floats = 64 KiB
	const int r = lane & 31, h = lane >> 5;
	float* const red = reinterpret_cast<float*>(smem) + (wave & 3) * 64 * 64 + lane;
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const bool shared = a == 0 ? shA : shB;   // workgroup-uniform
		const unsigned m = a == 0 ? mA : mB;
		if (shared) {
			if (tp == 1) {
#pragma unroll
				for (int mi = 0; mi < 2; ++mi)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni)
#pragma unroll
						for (int g = 0; g < 16; ++g) red[((mi * 2 + ni) * 16 + g) * 64] = acc[a][mi][ni][g];
			}
			__syncthreads();
			if (tp == 0) {
#pragma unroll
				for (int mi = 0; mi < 2; ++mi)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni)
#pragma unroll
						for (int g = 0; g < 16; ++g) acc[a][mi][ni][g] += red[((mi * 2 + ni) * 16 + g) * 64];
			}
		}
		if (m == 0u || (shared && tp == 1)) continue;
		const int tap = a == 0 ? tapA : tapB;
		float* sl = p.slab + ((int64_t)split * p.K + tap) * p.Cout * p.Cin;
#pragma unroll
		for (int mi = 0; mi < 2; ++mi)
#pragma unroll
			for (int ni = 0; ni < 2; ++ni) {
				const int ci = ci0 + wn * 64 + ni * 32 + r;
#pragma unroll
				for (int g = 0; g < 16; ++g) {
					const int co = co0 + wm * 64 + mi * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
					sl[(int64_t)co * p.Cin + ci] = acc[a][mi][ni][g];
				}
			}
	}
#ifdef CONVASR_STAMPS
	WSTAMP(w_t2)
	if (blockIdx.x < 256 && lane == 0) {
		unsigned long long* o = g_w2_stamps + (blockIdx.x * 8 + wave) * 8;
		o[0] = w_work; o[1] = w_bar; o[2] = w_t1 - w_t0; o[3] = w_t2 - w_t1; o[4] = (unsigned long long)(c_end - c_begin);
	}
#endif
}

template <typename H> __global__ __launch_bounds__(W2_ALL_THREADS, 3) void conv1d_wgrad_v2_kernel(WgradParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	wgrad_v2_body<H>(p, xcd_remap(blockIdx.x, p.units * p.splits), smem);
}

template <typename H> __global__ __launch_bounds__(W2_ALL_THREADS, 3) void conv1d_wgrad_v2_grouped_kernel(WgradGroup g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int v = xcd_remap(blockIdx.x, g.first[g.n]);
	int q = 0;
#pragma unroll
	for (int i = 1; i < W2_MAX_GROUP; ++i) q += (i < g.n && v >= g.first[i]) ? 1 : 0;
	q = __builtin_amdgcn_readfirstlane(q);
	wgrad_v2_body<H>(g.prob[q], v - g.first[q], smem);
}

// The kernel's envelope: fills the plan in `q` (a copy of `p`) and returns the dynamic LDS bytes, or 0 if the shape is outside it.
static size_t wgrad_v2_plan(const WgradParams& p, WgradParams& q) {
	if (p.stride != 1 || (p.Cin & 127) != 0 || (p.Cout & 127) != 0) return 0;
	if ((int64_t)p.Tin * (p.x_ld ? p.x_ld : p.Cin) * 2 >= (1ll << 31) || (int64_t)p.Tout * (p.dy_ld ? p.dy_ld : p.Cout) * 2 >= (1ll << 31)) return 0;
	if ((p.x_ld & 7) || (p.dy_ld & 7)) return 0;  // (16-byte DMA pieces)
	q = p;
	wgrad_plan(q, W2_BKT, 1.6);
	q.x_rows = (q.x_rows + 3) & ~3;
	const int pieces = W2_BKT / 4 + q.x_rows / 4;
	if (pieces > 40) return 0;  // at most 10 pieces per loader wave: the counted waits above
	const size_t smem = 4 * (size_t)(W2_YBYTES + q.x_rows * 256);
	return smem > 160 * 1024 ? 0 : smem;
}

int convasr_wgrad_v2_supports(const WgradParams& p) {
	WgradParams q;
	return wgrad_v2_plan(p, q) != 0;
}

// Fills the plan in `p` and launches; returns 0 (plan untouched) if the shape is outside this kernel's envelope.
int convasr_wgrad_v2_try(WgradParams& p, int dtype, hipStream_t s) {
	WgradParams q;
	const size_t smem = wgrad_v2_plan(p, q);
	if (!smem) return 0;
	const bool f16 = dtype == CONVASR_F16;
	const void* kern = f16 ? (const void*)conv1d_wgrad_v2_kernel<f16_t> : (const void*)conv1d_wgrad_v2_kernel<bf16_t>;
	static unsigned long long set[2] = {0, 0};
	convasr_allow_160k_lds(kern, set[f16]);
	void* args[] = {&q};
	if (hipLaunchKernel(kern, dim3(q.units * q.splits), dim3(W2_ALL_THREADS), args, smem, s) != hipSuccess) { (void)hipGetLastError(); return 0; }  // (the caller's fallback starts from a clean error state)
	p = q;
	return 1;
}

// combine of a grouped launch: dw_i (+)= sum over splits of slab_i, every problem's 16-byte pieces in one streaming launch; plus the
// zero fill of the branches' bias gradients (identically zero for a conv that feeds a train-mode batch norm: functional.py)
struct WgradCombine {
	const float* slab[W2_MAX_GROUP]; float* dw[W2_MAX_GROUP]; float* zero[W2_MAX_GROUP];
	long long first4[W2_MAX_GROUP + 1];  // prefix sums of the problems' float4 counts
	int zero_n[W2_MAX_GROUP];
	int accumulate[W2_MAX_GROUP];
	int n, S;
};

__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(WgradCombine c) {
	const long long total = c.first4[c.n];
	for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
		int q = 0;
#pragma unroll
		for (int k = 1; k < W2_MAX_GROUP; ++k) q += (k < c.n && i >= c.first4[k]) ? 1 : 0;
		const long long j = i - c.first4[q], n4 = c.first4[q + 1] - c.first4[q];
		const float4* const s4 = reinterpret_cast<const float4*>(c.slab[q]);
		float4 a = s4[j];
		for (int s = 1; s < c.S; ++s) { const float4 b = s4[(long long)s * n4 + j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
		float4* const d4 = reinterpret_cast<float4*>(c.dw[q]);
		if (c.accumulate[q]) { const float4 o = d4[j]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
		d4[j] = a;
	}
	if (blockIdx.x == 0)
		for (int q = 0; q < c.n; ++q)
			if (c.zero[q])
				for (int k = threadIdx.x; k < c.zero_n[q]; k += 256) c.zero[q][k] = 0.f;
}

static void wgrad_group_plan(WgradParams* probs, int n) {
	// one split count for all problems (they reduce over the same frames): rounds of workgroups over the 256 CUs x chunks per workgroup x
	// time per chunk, plus every problem's partial-slab traffic (wgrad_plan's cost model over the summed units)
	for (int i = 0; i < n; ++i) {
		WgradParams& p = probs[i];
		p.co_tiles = p.Cout / 128; p.ci_tiles = p.Cin / 128; p.tap_groups = 1; p.units = p.co_tiles * p.ci_tiles;
		p.chunks_per_b = (p.Tout + W2_BKT - 1) / W2_BKT; p.total_chunks = p.B * p.chunks_per_b;
		p.x_rows = W2_BKT;
	}
	const int total_chunks = probs[0].total_chunks;
	int units = 0; double slab_us = 0;
	for (int i = 0; i < n; ++i) { units += probs[i].units; slab_us += 2.0 * (double)probs[i].Cout * probs[i].Cin * 4.0 / 4e6; }
	double best = 1e30; int best_s = 1;
	for (int s = 1; s <= WGRAD_MAX_SPLITS && s <= total_chunks; ++s) {
		const int cps = (total_chunks + s - 1) / s, s_eff = (total_chunks + cps - 1) / cps;
		const int rounds = (units * s_eff + 255) / 256;
		const double cost = rounds * cps * 1.6 + s_eff * slab_us;
		if (cost < best) { best = cost; best_s = s_eff; }
	}
	for (int i = 0; i < n; ++i) {
		WgradParams& p = probs[i];
		p.chunks_per_split = (total_chunks + best_s - 1) / best_s;
		p.splits = (total_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
	}
}

extern "C" int64_t convasr_wgrad1x1_grouped_workspace_bytes(int n, const int* cin, const int* cout, int B, int T) {
	if (n <= 0 || n > W2_MAX_GROUP || !cin || !cout || B <= 0 || T <= 0) return -1;
	WgradParams probs[W2_MAX_GROUP];
	for (int i = 0; i < n; ++i) { probs[i] = WgradParams(); probs[i].B = B; probs[i].Cin = cin[i]; probs[i].Cout = cout[i]; probs[i].Tin = probs[i].Tout = T; probs[i].K = 1; probs[i].stride = probs[i].dil = 1; }
	wgrad_group_plan(probs, n);
	int64_t bytes = 0;
	for (int i = 0; i < n; ++i) bytes += (int64_t)probs[i].splits * cout[i] * cin[i] * 4;
	return bytes;
}

// n one-tap weight gradients dw_i[co][ci] (+)= sum_(b,t) dy_i[b,t,co] * x_i[b,t,ci] over the same B * T frames: ONE dispatch of the LDS-DMA
// wgrad kernel over all problems' (co tile, ci tile, split) units and ONE streaming combine.  zero_i (may be NULL): cout[i] floats set to
// zero by the combine (the branch's bias gradient).  cin[i] % 128 == 0, cout[i] % 128 == 0, 16-bit storage.
extern "C" int convasr_wgrad1x1_grouped(int n, const void* const* x, const void* const* dy, float* const* dw, float* const* zero, const int* cin, const int* cout,
                                        const int* accumulate, void* workspace, int dtype, int B, int T, void* stream) {
	CONVASR_CHECK_ARG(n > 0 && n <= W2_MAX_GROUP && x && dy && dw && cin && cout && workspace && B > 0 && T > 0, "wgrad1x1_grouped: bad arguments (at most %d problems)", W2_MAX_GROUP);
	if (!convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "wgrad1x1_grouped: dtype %d (16-bit storage only)", dtype);
	WgradGroup g = {};
	WgradCombine c = {};
	for (int i = 0; i < n; ++i) {
		CONVASR_CHECK_ARG(x[i] && dy[i] && dw[i], "wgrad1x1_grouped: problem %d has a NULL operand", i);
		if ((cin[i] & 127) || (cout[i] & 127) || cin[i] <= 0 || cout[i] <= 0 || (int64_t)T * cin[i] * 2 >= (1ll << 31) || (int64_t)T * cout[i] * 2 >= (1ll << 31))
			return convasr_fail(CONVASR_EUNSUPPORTED, "wgrad1x1_grouped: problem %d (%d x %d channels) is outside the kernel's envelope (channel counts %% 128)", i, cout[i], cin[i]);
		WgradParams& p = g.prob[i];
		p = WgradParams();
		p.x = x[i]; p.dy = dy[i]; p.B = B; p.Cin = cin[i]; p.Cout = cout[i]; p.Tin = p.Tout = T; p.K = 1; p.stride = p.dil = 1; p.pad = 0; p.debug = 0;
	}
	wgrad_group_plan(g.prob, n);
	g.n = c.n = n;
	c.S = g.prob[0].splits;
	float* slab = (float*)workspace;
	for (int i = 0; i < n; ++i) {
		g.prob[i].slab = slab;
		c.slab[i] = slab; c.dw[i] = dw[i]; c.zero[i] = zero ? zero[i] : nullptr; c.zero_n[i] = cout[i]; c.accumulate[i] = accumulate ? accumulate[i] : 0;
		c.first4[i + 1] = c.first4[i] + (long long)cout[i] * cin[i] / 4;
		slab += (int64_t)g.prob[i].splits * cout[i] * cin[i];
		g.first[i + 1] = g.first[i] + g.prob[i].units * g.prob[i].splits;
	}
	const size_t smem = 4 * (size_t)(W2_YBYTES + W2_BKT * 256);
	const bool f16 = dtype == CONVASR_F16;
	const void* kern = f16 ? (const void*)conv1d_wgrad_v2_grouped_kernel<f16_t> : (const void*)conv1d_wgrad_v2_grouped_kernel<bf16_t>;
	static unsigned long long set[2] = {0, 0};
	convasr_allow_160k_lds(kern, set[f16]);
	hipStream_t s = (hipStream_t)stream;
	void* args[] = {&g};
	if (hipLaunchKernel(kern, dim3(g.first[n]), dim3(W2_ALL_THREADS), args, smem, s) != hipSuccess) return convasr_fail(CONVASR_ELAUNCH, "wgrad1x1_grouped: %s", hipGetErrorString(hipGetLastError()));
	long long blocks = (c.first4[n] + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, c);
	CONVASR_CHECK_LAUNCH("wgrad1x1_grouped");
	return 0;
}
