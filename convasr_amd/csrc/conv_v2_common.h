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

