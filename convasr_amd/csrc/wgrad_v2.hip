// bf16 fast path of the weight gradient (stride 1, Cin % 128 == 0, Cout % 128 == 0, K >= 2).
//
//   dW[tap][co][ci] = sum_(b,t) dY[b,t,co] * X[b, t + tap*dil - pad, ci]        M = co, N = ci, reduction over (b, t)
//
// One workgroup (8 waves) = one 128(co) x 128(ci) tile x up to 4 taps over one split of the (b, t) axis.  Per 64-frame chunk
// the dY rows and the X rows (+ tap halo) are brought in by LDS-DMA into a 3-deep ring (issued two chunks ahead, counted
// vmcnt, raw s_barrier); both MFMA operands are column reads of row-major [t][c] tiles, served by ds_read_b64_tr_b16, with
// the 64-byte blocks of each 256-byte row XOR-swizzled by (row & 3) so the four rows a transposed read touches fall on
// different banks (the swizzle lives in the DMA's per-lane source address).  Waves 0-3 own taps {0,1} of the group, waves 4-7
// taps {2,3}; each wave a 64 x 64 sub-tile x 2 taps = 128 accumulator registers, two waves per SIMD.
// Zero padding of the conv and ragged chunk ends are the per-utterance buffer descriptors' range checks.
#include "conv_common.h"

#define W2_THREADS 512
#define W2_BKT 64
#define W2_YBYTES (W2_BKT * 256)

typedef int v4i32 __attribute__((ext_vector_type(4)));

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

__global__ __launch_bounds__(W2_THREADS, 2) void conv1d_wgrad_v2_kernel(WgradParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int tp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

	const int v = xcd_remap(blockIdx.x, p.units * p.splits);
	const int unit = v % p.units, split = v / p.units;
	const int tg = unit % p.tap_groups, ci_t = (unit / p.tap_groups) % p.ci_tiles, co_t = unit / (p.tap_groups * p.ci_tiles);
	const int co0 = co_t * 128, ci0 = ci_t * 128, tap0 = tg * WG_TG;
	const int c_begin = split * p.chunks_per_split, c_end = min(p.total_chunks, c_begin + p.chunks_per_split);
	const int tapA = tap0 + 2 * tp, tapB = tapA + 1;
	const bool actA = tapA < p.K, actB = tapB < p.K;

	const int xbytes = p.x_rows * 256;  // x_rows is a multiple of 4: whole 1-KiB pieces
	const int stage_bytes = W2_YBYTES + xbytes;
	const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
	const int y_row_bytes = p.Cout * 2, x_row_bytes = p.Cin * 2;
	const int y_pieces = W2_BKT / 4, x_pieces = p.x_rows >> 2, pieces = y_pieces + x_pieces;
	const int my_pieces = (pieces - wave + 7) >> 3;  // pieces j = wave + 8 i
	// per-lane source offset inside a 4-row piece: row (lane >> 4), 16-byte chunk (lane & 15) ^ (row << 2)
	const int prow = lane >> 4, pchunk = (lane & 15) ^ (prow << 2);
	const int ylane = prow * y_row_bytes + pchunk * 16 + co0 * 2, xlane = prow * x_row_bytes + pchunk * 16 + ci0 * 2;

	auto issue = [&](int c, int stage) {
		const int b = c / p.chunks_per_b, t0 = (c % p.chunks_per_b) * W2_BKT;
		const v4i32 ysrd = w2_make_srd(reinterpret_cast<const char*>(p.dy) + (int64_t)b * p.Tout * y_row_bytes, (unsigned)(p.Tout * y_row_bytes));
		const v4i32 xsrd = w2_make_srd(reinterpret_cast<const char*>(p.x) + (int64_t)b * p.Tin * x_row_bytes, (unsigned)(p.Tin * x_row_bytes));
		const unsigned dst = lds_base + stage * stage_bytes;
		const int tin0 = t0 - p.pad + tap0 * p.dil;
		for (int j = wave; j < pieces; j += 8) {
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

	if (c_begin < c_end) issue(c_begin, 0);
	if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();

	// fragment addressing: 16-lane group g4 -> (column block (g4 & 1) * 16, k block 8 * (g4 >> 1)); lane 4q + pc in the group
	// supplies row q, columns 4 pc .. 4 pc + 3 of the 4 x 16 block (ds_read_b64_tr_b16 contract)
	const int g4 = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
	const int krow = 8 * (g4 >> 1) + q;
	const int colA = (wm * 64 + (g4 & 1) * 16 + 4 * pc) * 2, colB = (wn * 64 + (g4 & 1) * 16 + 4 * pc) * 2;  // byte column of mi/ni = 0 (+64 B for 1)
	const int offA = (tapA - tap0) * p.dil, offB = (tapB - tap0) * p.dil;

	int stage = 0;
	for (int c = c_begin; c < c_end; ++c) {
		const bool more = c + 2 < c_end;
		if (more) issue(c + 2, stage >= 1 ? stage - 1 : 2);
		const char* ys = smem + stage * stage_bytes;
		const char* xs = ys + W2_YBYTES;
#pragma unroll
		for (int kk = 0; kk < W2_BKT / 16; ++kk) {
			const int r0 = kk * 16 + krow;  // (r0 & 3) == q, ((r0 + 4) & 3) == q
			uint4 a[2];
#pragma unroll
			for (int mi = 0; mi < 2; ++mi) {
				const int cb = colA + mi * 64, sw = (((cb >> 6) ^ q) << 6) | (cb & 63);
				a[mi] = w2_tr_frag(ys + r0 * 256 + sw, ys + (r0 + 4) * 256 + sw);
			}
			if (actA) {
				const int rx = r0 + offA, s3 = rx & 3;
				uint4 bb[2];
#pragma unroll
				for (int ni = 0; ni < 2; ++ni) {
					const int cb = colB + ni * 64, sw = (((cb >> 6) ^ s3) << 6) | (cb & 63);
					bb[ni] = w2_tr_frag(xs + rx * 256 + sw, xs + (rx + 4) * 256 + sw);
				}
#pragma unroll
				for (int mi = 0; mi < 2; ++mi)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni) Mma<bf16_t>::run(a[mi], bb[ni], acc[0][mi][ni]);
			}
			if (actB) {
				const int rx = r0 + offB, s3 = rx & 3;
				uint4 bb[2];
#pragma unroll
				for (int ni = 0; ni < 2; ++ni) {
					const int cb = colB + ni * 64, sw = (((cb >> 6) ^ s3) << 6) | (cb & 63);
					bb[ni] = w2_tr_frag(xs + rx * 256 + sw, xs + (rx + 4) * 256 + sw);
				}
#pragma unroll
				for (int mi = 0; mi < 2; ++mi)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni) Mma<bf16_t>::run(a[mi], bb[ni], acc[1][mi][ni]);
			}
		}
		// leave only the pieces issued in this iteration (chunk c + 2) in flight
		if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		else if (my_pieces == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
		else if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		stage = stage == 2 ? 0 : stage + 1;
	}

	const int r = lane & 31, h = lane >> 5;
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const int tap = tapA + a;
		if (tap < p.K) {
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
	}
}

// Fills the plan in `p` and launches; returns 0 (plan untouched) if the shape is outside this kernel's envelope.
int convasr_wgrad_v2_try(WgradParams& p, hipStream_t s) {
	if (p.stride != 1 || (p.Cin & 127) != 0 || (p.Cout & 127) != 0 || p.K < 2) return 0;
	if ((int64_t)p.Tin * p.Cin * 2 >= (1ll << 31) || (int64_t)p.Tout * p.Cout * 2 >= (1ll << 31)) return 0;
	WgradParams q = p;
	wgrad_plan(q, W2_BKT, 1.6);
	q.x_rows = (q.x_rows + 3) & ~3;
	const int pieces = W2_BKT / 4 + q.x_rows / 4;
	if (pieces > 40) return 0;  // at most 5 pieces per wave: the counted waits above
	const size_t smem = 3 * (size_t)(W2_YBYTES + q.x_rows * 256);
	if (smem > 160 * 1024) return 0;
	static bool set = false;
	if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_wgrad_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
	hipLaunchKernelGGL(conv1d_wgrad_v2_kernel, dim3(q.units * q.splits), dim3(W2_THREADS), smem, s, q);
	p = q;
	return 1;
}
