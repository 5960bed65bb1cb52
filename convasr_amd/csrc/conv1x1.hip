// One-tap (K = 1) Conv1d of 16-bit activations as a plain GEMM over the flattened batch:  Y[m][co] = sum_ci X[m][ci] * W[co][ci] (+ bias[co]),
// m = (b, t) over all B * T frames.  Reference arithmetic: nn.Conv1d(kernel_size = 1), the residual branches of models.py:107-110 / 129-131
// (up to ten per dense block, 55 per JasperNetLarge step) and their input gradients.
//
// Why its own kernel: conv_v2s.hip's 256 x 128 tile owns a CU (150 KB of LDS, 12 waves of 168 registers) and spends ~9.6 K cycles per tile
// in prologue + epilogue whatever the reduction length; a one-tap tile over 256-640 input channels has only 4-10 barrier intervals of
// ~870 cycles to put beside them -- the launches ran at 20-30 % of the HBM rate they are bound by (profiles/r04_config4_layers.txt).
// Here a workgroup is small (128 x 128 tile, 4 waves of 64 x 64, 37 or 64 KB of LDS, 104 registers) so that several are resident per CU:
// one workgroup's output staging and stores overlap another's DMA and MFMAs by occupancy, with nothing to schedule by hand.
//   * both operands come in by LDS-DMA (buffer_load_dwordx4 ... lds), stages of (X 128 rows + W 128 rows) x 128 B.  One stage (the default:
//     four workgroups per CU): issue the slab -> wait for the own pieces -> barrier -> 2 x (8 ds_read_b128 + 16 v_mfma_f32_16x16x32) per
//     wave -> barrier.  Two stages (two workgroups per CU; long reductions into few output channels): barrier -> issue the next slab into
//     the other stage -> the MFMAs of this one -> wait for the own pieces;
//   * the LDS image, its XOR swizzle and the fragment addressing are conv_v2s.hip's (conflict-free ds_read_b128 for every row base);
//   * rows past B * T and (for padded weights) rows past Cout read zeros through the buffer descriptors' range checks;
//   * epilogue: (+ bias) -> BN statistics of that value (per-tile fp64 partial rows, summed in a fixed order by bn_finalize) -> 16-bit
//     store through an LDS tile, 16 bytes per lane.  Same k order per element as conv_v2s.hip: bit-identical outputs.
// Envelope: K = 1, stride 1, no padding, Cin % 64 == 0, Cout % 128 == 0, output type = input type, no folded scale / activation / length
// mask (the training launches; everything else stays in conv_v2s.hip).
#include "conv_v2_common.h"

#define C11_THREADS 256
#define C11_BM 128
#define C11_STAGE (2 * C11_BM * ROW_BYTES)  // X tile + W tile of one 64-channel slab: 32 KiB

// What one tile needs to know about its problem: the single-problem launch fills it from ConvParams, the grouped launch from its table.
struct C11Args {
	const void* x; const void* w; void* y; const float* bias; double* stats;
	int Cin, Cout, CoutPad;
	int64_t M;       // rows = B * T frames
	int accumulate;  // y += (instead of y =): the grouped input-gradient launches add into the tapped block output's gradient
};

// NS = LDS stages: 2 = the next slab's DMA is in flight while this one is computed (64 KiB: two workgroups per CU); 1 = issue, wait, compute
// (36.9 KiB incl. the epilogue's tile: four workgroups per CU, twice the bytes in flight per CU, the overlap left to occupancy alone).
template <typename I, int NS> __device__ __forceinline__ void conv1x1_tile(const C11Args& p, const int mtile, const int ntile, char* const smem) {
	constexpr int MI = 4, NB = 4;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int r16 = lane & 15, kb = lane >> 4, wm = wave >> 1, wn = wave & 1;

	const int m0 = mtile * C11_BM, co0 = ntile * BN;
	const int64_t M = p.M;

	const int row_bytes = p.Cin * 2;
	const v4i32 xsrc = make_srd(p.x, (unsigned)(M * row_bytes));
	const v4i32 wsrc = make_srd(p.w, (unsigned)(p.CoutPad * row_bytes));
	const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
	const int n_cib = p.Cin >> 6;

	// 16 X pieces + 16 W pieces of 1 KiB per slab; wave w issues pieces w, w + 4, ... of each (a piece = 8 rows x 128 B, swizzled per lane)
	const int plane = v2s_src_offset(lane, row_bytes);
	auto issue = [&](int cib, int stage) {
		const unsigned dst = lds_base + stage * C11_STAGE;
		const int xbase = m0 * row_bytes + cib * 128, wbase = co0 * row_bytes + cib * 128;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int u = wave + 4 * j;
			dma16(xsrc, __builtin_amdgcn_readfirstlane(dst + u * 1024), xbase + u * 8 * row_bytes + plane);
			dma16(wsrc, __builtin_amdgcn_readfirstlane(dst + C11_BM * ROW_BYTES + u * 1024), wbase + u * 8 * row_bytes + plane);
		}
	};

	f32x4 acc[MI][NB];
#pragma unroll
	for (int i = 0; i < MI; ++i)
#pragma unroll
		for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(3))) u32x4* lds_u4;
	const int kb4 = kb << 4;
	auto lane_off = [&](int row) { return (unsigned)((row << 7) + (kb4 ^ ((row << 4) & 0x60))); };
	const unsigned xa0 = lds_base + lane_off(wm * 64 + r16), wa0 = lds_base + C11_BM * ROW_BYTES + lane_off(wn * 64 + r16);

	if (NS == 2) {
		issue(0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	for (int cib = 0; cib < n_cib; ++cib) {
		if (NS == 1) {
			if (cib) __builtin_amdgcn_s_barrier();  // everyone is done reading the stage
			issue(cib, 0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__builtin_amdgcn_s_barrier();  // slab cib has landed for everyone (NS == 2: and everyone is done reading the other stage)
		if (NS == 2 && cib + 1 < n_cib) issue(cib + 1, (cib + 1) & 1);
		const unsigned so = NS == 2 ? (cib & 1) * C11_STAGE : 0u;
#pragma unroll
		for (int ks = 0; ks < 2; ++ks) {
			u32x4 a[MI], b[NB];
			const unsigned xa = (xa0 + so) ^ (ks << 6), wa = (wa0 + so) ^ (ks << 6);
#pragma unroll
			for (int i = 0; i < MI; ++i) { a[i] = *(lds_u4)(size_t)(xa + i * 2048); b[i] = *(lds_u4)(size_t)(wa + i * 2048); }
#pragma unroll
			for (int i = 0; i < MI; ++i)
#pragma unroll
				for (int j = 0; j < NB; ++j) acc[i][j] = Mma16<I>::run(a[i], b[j], acc[i][j]);
		}
		if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the own pieces of slab cib + 1
	}
	__syncthreads();  // the stages are dead: the output tile takes their place

	// ---------------- epilogue: C/D layout of 16x16 blocks: col = lane & 15, row = (lane >> 4) * 4 + reg
	constexpr int OPITCH = BN * 2 + 16;
	char* const otile = smem;
	float* const red = reinterpret_cast<float*>(smem + C11_BM * OPITCH);  // [2][2 (wm)][BN]
	const bool full = m0 + C11_BM <= M;
#pragma unroll
	for (int ni = 0; ni < NB; ++ni) {
		const int col = wn * 64 + ni * 16 + r16, co = co0 + col;
		const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const int row = wm * 64 + mi * 16 + kb * 4 + g;
				const float val = acc[mi][ni][g] + bias;
				if (p.stats && (full || m0 + row < M)) { s1 += val; s2 += val * val; }
				Elem<I>::store(reinterpret_cast<I*>(otile + row * OPITCH) + col, val);
			}
		}
		if (p.stats) {
			s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
			s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
			if (kb == 0) { red[(0 * 2 + wm) * BN + col] = s1; red[(1 * 2 + wm) * BN + col] = s2; }
		}
	}
	__syncthreads();
	if (p.stats && tid < BN && co0 + tid < p.Cout) {
		double* const prow = p.stats + (int64_t)mtile * 2 * p.Cout;  // per-(m tile) partial row, summed in a fixed order by bn_finalize
		prow[co0 + tid] = (double)red[(0 * 2 + 0) * BN + tid] + (double)red[(0 * 2 + 1) * BN + tid];
		prow[p.Cout + co0 + tid] = (double)red[(1 * 2 + 0) * BN + tid] + (double)red[(1 * 2 + 1) * BN + tid];
	}
	I* const yb = reinterpret_cast<I*>(p.y);
	constexpr int OCH = BN / 8;  // 16-byte chunks per tile row
#pragma unroll
	for (int it = 0; it < C11_BM * OCH / C11_THREADS; ++it) {
		const int e = tid + it * C11_THREADS, row = e / OCH, ch = e % OCH;
		const int64_t m = (int64_t)m0 + row;
		if (m < M) {
			uint4 val = *reinterpret_cast<const uint4*>(otile + row * OPITCH + ch * 16);
			uint4* const dst = reinterpret_cast<uint4*>(yb + m * p.Cout + co0 + ch * 8);
			if (p.accumulate) {  // (workgroup-uniform) the sum of two stored 16-bit values, rounded once: what the pairwise add it replaces computed
				float a[8], o[8];
				unpack16<I>(val, a);
				unpack16<I>(*dst, o);
				val = make_uint4(pack16<I>(a[0] + o[0], a[1] + o[1]), pack16<I>(a[2] + o[2], a[3] + o[3]), pack16<I>(a[4] + o[4], a[5] + o[5]), pack16<I>(a[6] + o[6], a[7] + o[7]));
			}
			*dst = val;
		}
	}
}

template <typename I, int NS> __global__ __launch_bounds__(C11_THREADS, NS == 1 ? 4 : 2) void conv1x1_kernel(ConvParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// consecutive workgroups of an XCD walk the n tiles of one m tile first: its X rows are fetched from beyond L2 once
	const int v = xcd_remap(blockIdx.x, p.total_tiles);
	C11Args a;
	a.x = p.x; a.w = p.w; a.y = p.y; a.bias = p.bias; a.stats = p.stats; a.Cin = p.Cin; a.Cout = p.Cout; a.CoutPad = p.CoutPad; a.M = (int64_t)p.B * p.Tout; a.accumulate = 0;
	conv1x1_tile<I, NS>(a, v / p.n_tiles, v % p.n_tiles, smem);
}

// ---------------- grouped launches: up to C11_MAX_GROUP independent one-tap problems over the same B * T frames in ONE dispatch.
// The residual branches of a dense block (models.py:107-110, 129-131: up to ten 1x1 convs per block, each feeding its OWN batch norm, so
// they cannot share a GEMM -- but they can share a launch): every problem keeps its own operands, output, bias and statistics rows and
// is computed exactly as by a launch of its own (bit-identical per element); the tile index picks the problem by a prefix sum.  Ten
// launches of 0.2-1.5 rounds at the 10-25 us launch floor become one that fills the chip.
#define C11_MAX_GROUP 12
struct C11Group {
	C11Args prob[C11_MAX_GROUP];
	int first[C11_MAX_GROUP + 1];  // first tile of each problem; first[n] = all tiles
	int n_tiles[C11_MAX_GROUP];    // n tiles (Cout / 128) of each problem
	int n;
};

template <typename I, int NS> __global__ __launch_bounds__(C11_THREADS, NS == 1 ? 4 : 2) void conv1x1_grouped_kernel(C11Group g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int v = xcd_remap(blockIdx.x, g.first[g.n]);
	int q = 0;
#pragma unroll
	for (int i = 1; i < C11_MAX_GROUP; ++i) q += (i < g.n && v >= g.first[i]) ? 1 : 0;
	q = __builtin_amdgcn_readfirstlane(q);
	const int local = v - g.first[q], nt = g.n_tiles[q];
	conv1x1_tile<I, NS>(g.prob[q], local / nt, local % nt, smem);
}

// Returns 1 if this kernel took the launch (rows_out = partial statistics rows written), 0 if the shape is outside its envelope.
int convasr_conv1x1_try(ConvParams p, int x_dtype, int y_dtype, hipStream_t s, int* rows_out) {
	if (p.K != 1 || p.stride != 1 || p.pad != 0 || p.Tin != p.Tout || (p.Cin & 63) != 0 || (p.Cout & 127) != 0 || p.CoutPad != p.Cout) return 0;
	if (!convasr_is_half(x_dtype) || y_dtype != x_dtype || p.scale || p.act != CONVASR_ACT_NONE || p.xlen || p.bn_y) return 0;
	// Measured against conv_v2s.hip on every one-tap shape of the two bench workloads (profiles/r04_ab_conv1x1.json, scratch/ab_conv1x1.py:
	// 32 x 376-1001 frames, 256-768 channels; 64 x 753, 896 <-> 1024): bit-identical outputs, 15-25 % less time throughout (2-3.5 TB/s
	// of algorithmic bytes; the L2 -> LDS fills, ~5x those bytes at a 128 x 128 tile, are what it runs into).  debug bit 8192 forbids
	// this kernel (A/B runs)
	if (p.debug & 8192) return 0;
	const int64_t M = (int64_t)p.B * p.Tout;
	if (M * p.Cin * 2 >= (1ll << 31) || M * p.Cout * 2 >= (1ll << 31) || (int64_t)p.CoutPad * p.Cin * 2 >= (1ll << 31)) return 0;
	p.n_tiles = p.Cout / BN;
	const int m_tiles = (int)((M + C11_BM - 1) / C11_BM);
	p.total_tiles = m_tiles * p.n_tiles;
	// Stages: one (four workgroups per CU) unless the reduction is long against the output width (Cin >= 3 Cout: 768 -> 256), where the
	// two-stage form's prefetch wins by 7-10 %; everywhere else on profiles/r04_ab_conv1x1.json's shapes the single stage is 0-25 % faster
	// (640 -> 768 at 32 x 376: 27.3 -> 20.9 us; 896 -> 1024 at 64 x 753: 114.6 -> 97.5 us).  debug bit 16384 inverts the choice (A/B runs).
	const bool one_stage = (p.Cin < 3 * p.Cout) != ((p.debug & 16384) != 0);
	const size_t epi = (size_t)C11_BM * (BN * 2 + 16) + 4 * BN * sizeof(float);  // the epilogue's output tile + sums: 36.9 KB
	const size_t smem = one_stage ? (epi > C11_STAGE ? epi : C11_STAGE) : 2 * (size_t)C11_STAGE;
	const bool f16 = x_dtype == CONVASR_F16;
	const void* kern = one_stage ? (f16 ? (const void*)conv1x1_kernel<f16_t, 1> : (const void*)conv1x1_kernel<bf16_t, 1>) : (f16 ? (const void*)conv1x1_kernel<f16_t, 2> : (const void*)conv1x1_kernel<bf16_t, 2>);
	static unsigned long long set[2][2] = {};
	convasr_allow_160k_lds(kern, set[one_stage][f16]);
	void* args[] = {&p};
	if (hipLaunchKernel(kern, dim3(p.total_tiles), dim3(C11_THREADS), args, smem, s) != hipSuccess) { (void)hipGetLastError(); return 0; }
	if (rows_out) *rows_out = m_tiles;
	return 1;
}

int convasr_conv_debug_bits();
static bool c11_ok(int Cin, int Cout) { return (Cin & 63) == 0 && (Cout & 127) == 0 && Cin > 0 && Cout > 0; }

// n one-tap problems y_i (+)= x_i * w_i^T (+ bias_i) over the same B * T frames (16-bit channels-last activations, packed forward-layout
// weights [1][Cout_i][Cin_i]); stats_i (may be NULL): per-m-tile partial rows like convasr_conv1d_fwd's.  Problems whose reduction is
// long against their output width run in the two-stage instantiation, the others in the one-stage one: at most two dispatches.
extern "C" int convasr_conv1x1_grouped(int n, const void* const* x, const void* const* w, void* const* y, const float* const* bias, double* const* stats,
                                       const int* cin, const int* cout, const int* accumulate, int dtype, int B, int T, int* stats_rows, void* stream) {
	CONVASR_CHECK_ARG(n > 0 && n <= C11_MAX_GROUP && x && w && y && cin && cout && B > 0 && T > 0, "conv1x1_grouped: bad arguments (at most %d problems)", C11_MAX_GROUP);
	if (!convasr_is_half(dtype)) return convasr_fail(CONVASR_EUNSUPPORTED, "conv1x1_grouped: dtype %d (16-bit storage only)", dtype);
	const int64_t M = (int64_t)B * T;
	const int m_tiles = (int)((M + C11_BM - 1) / C11_BM);
	for (int i = 0; i < n; ++i) {
		CONVASR_CHECK_ARG(x[i] && w[i] && y[i], "conv1x1_grouped: problem %d has a NULL operand", i);
		if (!c11_ok(cin[i], cout[i]) || M * cin[i] * 2 >= (1ll << 31) || M * cout[i] * 2 >= (1ll << 31) || (int64_t)cout[i] * cin[i] * 2 >= (1ll << 31))
			return convasr_fail(CONVASR_EUNSUPPORTED, "conv1x1_grouped: problem %d (%d -> %d channels, %lld frames) is outside the one-tap kernel's envelope (Cin %% 64, Cout %% 128)", i, cin[i], cout[i], (long long)M);
	}
	const bool f16 = dtype == CONVASR_F16;
	hipStream_t s = (hipStream_t)stream;
	for (int two = 0; two < 2; ++two) {  // pass 0: the one-stage problems, pass 1: the two-stage ones
		C11Group g = {};
		for (int i = 0; i < n; ++i) {
			const bool one_stage = (cin[i] < 3 * cout[i]) != ((convasr_conv_debug_bits() & 16384) != 0);
			if (one_stage == (two != 0)) continue;
			C11Args& a = g.prob[g.n];
			a.x = x[i]; a.w = w[i]; a.y = y[i]; a.bias = bias ? bias[i] : nullptr; a.stats = stats ? stats[i] : nullptr;
			a.Cin = cin[i]; a.Cout = cout[i]; a.CoutPad = cout[i]; a.M = M; a.accumulate = accumulate ? accumulate[i] : 0;
			g.n_tiles[g.n] = cout[i] / BN;
			g.first[g.n + 1] = g.first[g.n] + m_tiles * g.n_tiles[g.n];
			++g.n;
		}
		if (g.n == 0) continue;
		const size_t epi = (size_t)C11_BM * (BN * 2 + 16) + 4 * BN * sizeof(float);
		const size_t smem = two ? 2 * (size_t)C11_STAGE : (epi > C11_STAGE ? epi : C11_STAGE);
		const void* kern = two ? (f16 ? (const void*)conv1x1_grouped_kernel<f16_t, 2> : (const void*)conv1x1_grouped_kernel<bf16_t, 2>) : (f16 ? (const void*)conv1x1_grouped_kernel<f16_t, 1> : (const void*)conv1x1_grouped_kernel<bf16_t, 1>);
		static unsigned long long set[2][2] = {};
		convasr_allow_160k_lds(kern, set[two][f16]);
		void* args[] = {&g};
		if (hipLaunchKernel(kern, dim3(g.first[g.n]), dim3(C11_THREADS), args, smem, s) != hipSuccess) return convasr_fail(CONVASR_ELAUNCH, "conv1x1_grouped: %s", hipGetErrorString(hipGetLastError()));
	}
	if (stats_rows) *stats_rows = m_tiles;
	return 0;
}
