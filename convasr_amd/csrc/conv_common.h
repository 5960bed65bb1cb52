// Shared pieces of the implicit-GEMM conv kernels (conv.hip: register-staged general kernel; conv_v2s.hip: LDS-DMA bf16 / fp16 kernel).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define BM 128
#define BN 128
#define NTHREADS 256
#define ROW_BYTES 128  // one LDS row = 128 B of the reduction axis (64 bf16 or 32 f32)

struct ConvParams {
	const void* x;
	const void* w;
	void* y;
	const float* bias;
	double* stats;
	const float* scale;
	const float* shift;
	const float* xlen;
	int B, Cin, Cout, CoutPad, Tin, Tout, K, stride, dil, pad;
	int act;
	float act_lo, act_hi;
	int m_tiles_per_b, n_tiles, total_tiles;
	int full_tiles;  // conv_v2s only: tiles [0, full_tiles) are 256 x 128; each later one is computed as two 256 x 64 halves by two workgroups
	int x_rows;  // LDS rows of one X tile (even)
	int tail128;  // conv_v2s only: the last m tile of every utterance covers at most 128 frames and runs as a 128-row tile (half the MFMA work of a padded 256-row one)
	// conv_v2s only, optional (bn_y != NULL): this launch is the dgrad that produces dz of a Conv+BN+activation layer, and its
	// epilogue also runs pass 1 of that layer's batch-norm backward on the tile it just produced (see convasr_conv1d_dgrad_bn_reduce)
	const void* bn_y; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd; const float* bn_xlen; double* bn_sums;
	int bn_act; float bn_lo, bn_hi; unsigned bn_drop_thr; float bn_keep_scale; uint64_t bn_seed, bn_offset; const uint64_t* bn_step_key;  // (bn_step_key, optional: device word XORed into bn_seed, see convasr_step_begin)
	const uint8_t* bn_gate;  // optional: the one-bit gradient gates convasr_bn_act_fwd stored (instead of re-deriving act' / dropout / mask from bn_y)
	int debug;   // experiment flags (scratch/ only): 1 = skip DMA issue in the main loop, 2 = skip MFMAs, 4 = skip epilogue stores
	// conv_v2s only, optional (cib_per_split > 0): split-K over 64-channel input blocks for launches of a few tiles (small-batch inference).
	// Workgroup (x, y = split) reduces input blocks [split * cib_per_split, ...) and stores its fp32 partial tile at y + split * split_stride
	// (elements); convasr_conv1d_fwd_splitk's second kernel adds the partials in split order and runs the epilogue.
	int cib_per_split;
	long long split_stride;
};

// Two 128-B rows share one 256-B bank row; 16-B slot = (row parity, chunk ^ row-pair index): 16 consecutive rows at the same
// chunk land on 16 distinct slots -> ds_read_b128 fragments are bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) { return ((row >> 1) << 8) | ((((row & 1) << 3) | (chunk ^ ((row >> 1) & 7))) << 4); }

// XCD-aware bijective remap: consecutive virtual ids (same X tile, neighbouring weight tiles) share one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
	const int q = n >> 3, r = n & 7, xcd = bid & 7, k = bid >> 3;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Virtual tile id -> (m tile, n tile), blocked for the per-XCD L2: consecutive ids walk a 16 (m) x 2 (n) block of tiles, so the ~32
// workgroups an XCD runs at a time share 16 X tiles and the weight tiles of 2 n tiles (~11 MB of L2 fills per round of a 768-channel
// layer) instead of ~5 X tiles and ALL n tiles (~15 MB).  m blocks outermost, then n pairs, then m, n fastest; any M, N.
__device__ __forceinline__ void tile_coords(int v, int M, int N, int& m, int& n) {
	constexpr int MB = 16, NT = 2;
	const int mb = v / (MB * N);
	const int rows = min(MB, M - mb * MB);
	const int w = v - mb * MB * N;
	const int ng = w / (rows * NT);
	const int width = min(NT, N - ng * NT);
	const int w2 = w - ng * rows * NT;
	m = mb * MB + w2 / width;
	n = ng * NT + w2 % width;
}

#ifdef CONVASR_AB_BLOCK
// Diagnostic build only (python -m convasr_amd.build --variant abblock -DCONVASR_AB_BLOCK=1; scratch/ab_block.py): the same walk with the
// block shape chosen at run time (ConvParams::debug bits 10-12), to measure what the L2-missing operand bytes of each shape cost.
__device__ __forceinline__ void tile_coords_rt(int v, int M, int N, int MB, int NT, int& m, int& n) {
	const int mb = v / (MB * N);
	const int rows = min(MB, M - mb * MB);
	const int w = v - mb * MB * N;
	const int ng = w / (rows * NT);
	const int width = min(NT, N - ng * NT);
	const int w2 = w - ng * rows * NT;
	m = mb * MB + w2 / width;
	n = ng * NT + w2 % width;
}
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
	static constexpr int EPC = 8;  // elements per 16-byte chunk
	__device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
		c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
	}
};
template <> struct Mma<f16_t> {
	static constexpr int EPC = 8;
	__device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
		c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
	}
};
template <> struct Mma<float> {
	static constexpr int EPC = 4;
	// lane half h holds k = {4(2j+h) .. +3}; the i-th of four MFMAs pairs element i of both halves: every k is summed once.
	__device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
		c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
	}
};


// v_mfma_f32_16x16x32_{bf16,f16}: the forward / dgrad kernel's shape (conv_v2s.hip); both run at the same rate on gfx950
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
	template <typename V> __device__ static __forceinline__ f32x4 run(const V& a, const V& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
};
template <> struct Mma16<f16_t> {
	template <typename V> __device__ static __forceinline__ f32x4 run(const V& a, const V& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};


// ---------------- wgrad (conv.hip: general kernel; wgrad_v2.hip: LDS-DMA bf16 kernel)
#define WG_TG 4

struct WgradParams {
	const void* x;
	const void* dy;
	float* slab;
	int B, Cin, Cout, Tin, Tout, K, stride, dil, pad;
	int co_tiles, ci_tiles, tap_groups, units, splits;
	int chunks_per_b, total_chunks, chunks_per_split;
	int x_rows;
	int debug;  // experiment flags (diagnostic builds only)
	int x_ld, dy_ld;  // wgrad_v2 only: elements between consecutive frames of x / dy when they are not Cin / Cout (0 = dense) -- one plane of a split-operand plane tensor
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

#define WGRAD_MAX_SPLITS 32

// Work decomposition of one wgrad call: units = (co tile, ci tile, tap group); the (b, t) reduction axis is cut into `splits`
// contiguous ranges of chunks.  splits is chosen by a small cost model: rounds of workgroups over the 256 CUs x chunks per
// workgroup x measured time per chunk, plus the partial-slab traffic (one fp32 slab written and read back per split).
static inline void wgrad_plan(WgradParams& p, int bkt, double chunk_us) {
	p.co_tiles = (p.Cout + 127) / 128;
	p.ci_tiles = (p.Cin + 127) / 128;
	p.tap_groups = (p.K + WG_TG - 1) / WG_TG;
	p.units = p.co_tiles * p.ci_tiles * p.tap_groups;
	p.chunks_per_b = (p.Tout + bkt - 1) / bkt;
	p.total_chunks = p.B * p.chunks_per_b;
	const double slab_us = 2.0 * p.K * (double)p.Cout * p.Cin * 4.0 / 4e6;  // bytes / (4 TB/s), write + read
	double best = 1e30;
	int best_s = 1;
	for (int s = 1; s <= WGRAD_MAX_SPLITS && s <= p.total_chunks; ++s) {
		const int cps = (p.total_chunks + s - 1) / s, s_eff = (p.total_chunks + cps - 1) / cps;
		const int rounds = (p.units * s_eff + 255) / 256;
		const double cost = rounds * cps * chunk_us + s_eff * slab_us;
		if (cost < best) { best = cost; best_s = s_eff; }
	}
	p.splits = best_s;
	p.chunks_per_split = (p.total_chunks + best_s - 1) / best_s;
	p.splits = (p.total_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
	const int taps = p.K < WG_TG ? p.K : WG_TG;
	p.x_rows = (bkt - 1) * p.stride + (taps - 1) * p.dil + 1;
}
