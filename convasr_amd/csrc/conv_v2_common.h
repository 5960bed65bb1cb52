// Tile constants and the LDS-DMA primitive of conv_v2s.hip (16x16x32 MFMA forward / dgrad kernel).
#pragma once
#include "conv_common.h"

#define V2_BM 256
#define V2_THREADS 512
#define V2_WSLOT (BN * ROW_BYTES)  // 16 KiB
#define V2S_THREADS (V2_THREADS + 256)  // conv_v2s.hip: 8 computing waves + 4 loader waves

typedef int v4i32 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0, range-checked on num_bytes) from wave-uniform pieces
__device__ __forceinline__ v4i32 make_srd(const void* base, unsigned num_bytes) {
	const unsigned long long a = (unsigned long long)base;
	v4i32 d;
	d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
	d[2] = __builtin_amdgcn_readfirstlane((int)num_bytes);
	d[3] = 0x00020000;
	return d;
}

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at LDS byte address lds_addr), issued from inline asm so that hipcc neither
// counts it nor drains it with vmcnt(0) before the next ds_read: completion is tracked by the counted waits in the loop.
__device__ __forceinline__ void dma16(const v4i32& srd, unsigned lds_addr, int voff) {
	unsigned keep;
	asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(voff), "s"(srd), "s"(lds_addr)
	             : "memory");
}

// LDS image of a tile: 256-byte lines of two 128-byte rows (row = 2 * pair + s), a row's 16-byte chunk c at position
// s * 8 + (c ^ swz(pair)) of its line.  A ds_read_b128 fragment read serves lanes {0-3, 12-15, 20-27} (and the three like groups) in
// one LDS cycle each: 16 rows base + r16, k-block kb for eight of them and kb ^ 1 for the other eight, and it is conflict-free iff
// the 16 positions are distinct.  With swz = pair & 7 (conv_v2.hip's image, made for its own read pattern) that holds only for
// base = 0 mod 16, and the X fragments of tap t start at row t * dilation: SQ_LDS_BANK_CONFLICT was 24.5 % of SQ_LDS_IDX_ACTIVE.
// swz = (pair & 3) << 1 is conflict-free for EVERY base (exhaustive check over bases, groups and k sub-steps: scratch/swizzle_search.py).
__device__ __forceinline__ int v2s_swz(int pair) { return (pair & 3) << 1; }
// LDS slot p (16-byte units) of the tile <- global (row, chunk): byte offset of that lane's 16 bytes relative to row 0 / chunk 0
__device__ __forceinline__ int v2s_src_offset(int p, int row_bytes) {
	const int pair = p >> 4, s = p & 15;
	return (2 * pair + (s >> 3)) * row_bytes + (((s & 7) ^ v2s_swz(pair)) << 4);
}
