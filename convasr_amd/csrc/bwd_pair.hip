// The input gradient and the weight gradient of one Conv1d in ONE dispatch ("horizontal fusion"): workgroups [0, conv blocks) run a tile of
// conv_v2s.hip's forward / dgrad kernel (with or without the fused BN-backward epilogue), the rest one (co tile, ci tile, tap group, split) unit
// of wgrad_v2.hip's kernel.  Both read the same dY; neither reads what the other writes.  Why: at 32 utterances of 5-20 s (BASELINE
// configs[4]) a dgrad launch is 0.75-2.25 rounds of tiles on 256 CUs and the weight gradient of the same layer fills the CUs its partial
// rounds leave idle -- what the weight-gradient SIDE STREAM does in the eager step (+2-4 %), but a side stream is exactly what a replayed HIP
// graph cannot use on ROCm 7.2 (a forked capture replays slower than a linear one: DESIGN 11.3).  One launch gives the overlap to the eager
// step, the replayed step and the data-parallel step alike, and saves a launch per layer.
// Both halves are the unmodified device functions of their own files (same tile / unit code, same k order: bit-identical results); a
// workgroup is 768 threads either way (8 computing + 4 loader waves) and takes the larger of the two LDS footprints.
#undef CONVASR_STAMPS
#define CONVASR_PAIR_TU 1
#include "conv_v2s.hip"
#include "wgrad_v2.hip"

template <typename H, int BNF> __global__ __launch_bounds__(V2S_THREADS, 3) void conv1d_bwd_pair_kernel(ConvParams pc, WgradParams pw, int conv_blocks, int conv_blocks_pad) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int bid = blockIdx.x;
	if (bid < conv_blocks_pad) {
		if (bid < conv_blocks) v2s_block<H, H, BNF, V2_BM>(pc, bid, smem);  // (conv_blocks is padded to a multiple of 8 so that both halves keep their block -> XCD assumptions)
	} else {
		wgrad_v2_body<H>(pw, xcd_remap(bid - conv_blocks_pad, pw.units * pw.splits), smem);
	}
}

template <typename H> static const void* pair_kernel(int ki) {
	if (ki == 0) return (const void*)conv1d_bwd_pair_kernel<H, 0>;
	if (ki == 1) return (const void*)conv1d_bwd_pair_kernel<H, 1>;
	return (const void*)conv1d_bwd_pair_kernel<H, 2>;
}

// pc: the dgrad as conv.hip's conv1d_run filled it (bn_* set for the fused epilogue); pw: the weight gradient's problem (slab set).  Plans both,
// launches the fused kernel and returns 1 with pw's plan filled in and *m_tiles_out = the dgrad's partial-row count; returns 0 (nothing
// launched, pw untouched) when either half is outside its kernel's envelope.
int convasr_bwd_pair_try(ConvParams pc, WgradParams& pw, int dtype, hipStream_t s, int* m_tiles_out) {
	static_assert(V2S_THREADS == W2_ALL_THREADS, "both halves run 768-thread workgroups");
	if (!convasr_is_half(dtype)) return 0;
	V2sPlan pl;
	if (!v2s_plan(pc, dtype, dtype, pl) || pl.bi != 0 || pl.ki == 2) return 0;
	WgradParams q = pw;
	size_t wsmem;
	if (!w2_plan(q, wsmem)) return 0;
	const size_t smem = pl.smem > wsmem ? pl.smem : wsmem;
	const bool f16 = dtype == CONVASR_F16;
	const void* kern = f16 ? pair_kernel<f16_t>(pl.ki) : pair_kernel<bf16_t>(pl.ki);
	static unsigned long long set[2][4] = {};
	convasr_allow_160k_lds(kern, set[f16][pl.ki]);
	int conv_blocks = pl.grid, conv_pad = (pl.grid + 7) & ~7;
	void* args[] = {&pc, &q, &conv_blocks, &conv_pad};
	if (hipLaunchKernel(kern, dim3(conv_pad + q.units * q.splits), dim3(V2S_THREADS), args, smem, s) != hipSuccess) { (void)hipGetLastError(); return 0; }
	pw = q;
	if (m_tiles_out) *m_tiles_out = pc.B * pc.m_tiles_per_b;
	return 1;
}
