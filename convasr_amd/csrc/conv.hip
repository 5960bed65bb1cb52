// Conv1d as an im2col-free implicit GEMM on the gfx950 matrix cores (reference arithmetic: nn.Conv1d at models.py:53-76).
//
//   forward / dgrad :  D[(b,t)][co] = sum_tap sum_ci X[b, t*stride + tap*dil - pad, ci] * Wp[tap][co][ci]
//   wgrad           :  dW[tap][co][ci] = sum_(b,t) dY[b,t,co] * X[b, t*stride + tap*dil - pad, ci]
//
// Activations are channels-last (B, T, C): the GEMM reduction axis (ci) is the unit-stride axis of both operands, so every
// MFMA fragment is one 16-byte LDS read.  One workgroup owns a 128(t) x 128(co) output tile of ONE utterance; per 128-byte
// slab of input channels it stages the X rows [t0*stride - pad, ...) ONCE (tile + halo) and re-uses them for all K taps --
// the tap only shifts the LDS row a fragment is read from.  Weight tiles stream through a 2-deep ring, one per (ci-slab, tap).
// fp32 runs the same schedule on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), bf16 / fp16 on v_mfma_f32_32x32x16_{bf16,f16}.
#include "conv_common.h"

// 16 bytes of a row, zero outside [0, n_valid_elems); scalar path when rows are not 16-byte aligned.
template <typename T, bool ALIGNED> __device__ __forceinline__ uint4 load_chunk(const T* row, int e0, int n_valid) {
	constexpr int EPC = 16 / sizeof(T);
	uint4 v = make_uint4(0, 0, 0, 0);
	if (ALIGNED) {
		if (e0 + EPC <= n_valid) v = *reinterpret_cast<const uint4*>(row + e0);
		else if (e0 < n_valid) {
			T tmp[EPC];
#pragma unroll
			for (int i = 0; i < EPC; ++i) tmp[i] = (e0 + i < n_valid) ? row[e0 + i] : (T)0;
			v = *reinterpret_cast<uint4*>(tmp);
		}
	} else if (e0 < n_valid) {
		T tmp[EPC];
#pragma unroll
		for (int i = 0; i < EPC; ++i) tmp[i] = (e0 + i < n_valid) ? row[e0 + i] : (T)0;
		v = *reinterpret_cast<uint4*>(tmp);
	}
	return v;
}

template <typename T, typename O, int XI, bool ALIGNED_X, bool SK = false>  // (SK: a split-K launch of small-batch inference -- blockIdx.y's range of input slabs, an fp32 partial tile: convasr_conv1d_fwd_splitk)
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 4 ? 1 : 2) void conv1d_igemm_kernel(ConvParams p) {  // (fp32: one workgroup per CU, its fp64 totals below take 128 registers)
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int EPC = Mma<T>::EPC;
	constexpr int BK = ROW_BYTES / sizeof(T);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;

	const int v = xcd_remap(blockIdx.x, p.total_tiles);
	const int ntile = v % p.n_tiles, mtile = v / p.n_tiles;
	const int b = mtile / p.m_tiles_per_b, t0 = (mtile % p.m_tiles_per_b) * BM;
	const int co0 = ntile * BN;
	const int tin0 = t0 * p.stride - p.pad;

	const int xbytes = p.x_rows * ROW_BYTES;
	char* const xbuf = smem;
	char* const wbuf = smem + 2 * xbytes;
	const T* const xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.Tin * p.Cin;
	const T* const wp = reinterpret_cast<const T*>(p.w);
	const int cib0 = SK ? (int)blockIdx.y * p.cib_per_split : 0;  // this workgroup's first input slab (BK channels each)
	const int n_cib = SK ? min(p.cib_per_split, (p.Cin + BK - 1) / BK - cib0) : (p.Cin + BK - 1) / BK;
	const int total_x_chunks = p.x_rows * 8;

	uint4 xreg[XI];
	uint4 wreg[4];

	auto load_x = [&](int cib) {
		const int ci0 = (cib0 + cib) * BK;
#pragma unroll
		for (int i = 0; i < XI; ++i) {
			const int e = tid + NTHREADS * i;
			const int row = e >> 3, chunk = e & 7;
			const int tin = tin0 + row;
			xreg[i] = make_uint4(0, 0, 0, 0);
			if (e < total_x_chunks && tin >= 0 && tin < p.Tin) xreg[i] = load_chunk<T, ALIGNED_X>(xb + (int64_t)tin * p.Cin, ci0 + chunk * EPC, p.Cin);
		}
	};
	auto store_x = [&](int buf) {
#pragma unroll
		for (int i = 0; i < XI; ++i) {
			const int e = tid + NTHREADS * i;
			if (e < total_x_chunks) *reinterpret_cast<uint4*>(xbuf + buf * xbytes + lds_off(e >> 3, e & 7)) = xreg[i];
		}
	};
	auto load_w = [&](int cib, int tap) {
		const int ci0 = (cib0 + cib) * BK;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int e = tid + NTHREADS * i;
			const int row = e >> 3, chunk = e & 7;
			wreg[i] = load_chunk<T, ALIGNED_X>(wp + ((int64_t)tap * p.CoutPad + co0 + row) * p.Cin, ci0 + chunk * EPC, p.Cin);
		}
	};
	auto store_w = [&](int buf) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int e = tid + NTHREADS * i;
			*reinterpret_cast<uint4*>(wbuf + buf * (BN * ROW_BYTES) + lds_off(e >> 3, e & 7)) = wreg[i];
		}
	};

	f32x16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
	// fp32 (the parity path) sums in two levels: the MFMA accumulator chain is closed after every (32-channel slab, tap) step -- 32
	// products, 16 chained MFMA adds -- and added to a running total kept in fp64.  A single fp32 chain over all Cin K terms (8448 for
	// 768 channels, K = 11) random-walks to ~sqrt(n) / 2 ulp: 768 -> 768, K = 11 measured 1.2e-6 relative L2 from the float64 result even
	// with one chain per slab, against torch-CPU's 2.1e-7 (its blocked / vectorised sum keeps dozens of short chains; this is why the
	// round-2 fp32 path sat 1.7-3x further from float64 than the CPU oracle, profiles/r02_fp64_reference.json).  The fp32 path exists
	// for parity, not speed: 64 v_cvt + 64 v_add_f64 per step beside 64 MFMAs of 64 cycles each.  16-bit builds keep the single chain:
	// their sums must stay bit-identical to the LDS-DMA kernel's (tests/test_kernels_gpu.py), and their error is the storage rounding.
	constexpr bool TWO_LEVEL = sizeof(T) == 4;
	double total[2][2][16];
	if (TWO_LEVEL) {
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int k = 0; k < 16; ++k) total[i][j][k] = 0.0;
	}

	load_x(0);
	load_w(0, 0);
	store_x(0);
	store_w(0);
	__syncthreads();

	const int Q = n_cib * p.K;
	int cib = 0, tap = 0;
	const int wrow0 = wn * 64 + r, wrow1 = wrow0 + 32;
	const int woff0 = (wrow0 >> 1) << 8, wpar0 = (wrow0 & 1) << 3, wsw0 = (wrow0 >> 1) & 7;
	const int woff1 = (wrow1 >> 1) << 8, wpar1 = (wrow1 & 1) << 3, wsw1 = (wrow1 >> 1) & 7;

	for (int q = 0; q < Q; ++q) {
		const bool has_next = q + 1 < Q;
		const bool last_tap = tap == p.K - 1;
		const bool next_x = has_next && last_tap;
		if (has_next) load_w(last_tap ? cib + 1 : cib, last_tap ? 0 : tap + 1);
		if (tap == 0 && cib + 1 < n_cib) load_x(cib + 1);

		const char* xs = xbuf + (cib & 1) * xbytes;
		const char* ws = wbuf + (q & 1) * (BN * ROW_BYTES);
		const int xrow0 = (wm * 64 + r) * p.stride + tap * p.dil, xrow1 = xrow0 + 32 * p.stride;
		const int xoff0 = (xrow0 >> 1) << 8, xpar0 = (xrow0 & 1) << 3, xsw0 = (xrow0 >> 1) & 7;
		const int xoff1 = (xrow1 >> 1) << 8, xpar1 = (xrow1 & 1) << 3, xsw1 = (xrow1 >> 1) & 7;
#pragma unroll
		for (int kk = 0; kk < 4; ++kk) {
			const int chunk = kk * 2 + h;
			const uint4 a0 = *reinterpret_cast<const uint4*>(xs + xoff0 + ((xpar0 | (chunk ^ xsw0)) << 4));
			const uint4 a1 = *reinterpret_cast<const uint4*>(xs + xoff1 + ((xpar1 | (chunk ^ xsw1)) << 4));
			const uint4 b0 = *reinterpret_cast<const uint4*>(ws + woff0 + ((wpar0 | (chunk ^ wsw0)) << 4));
			const uint4 b1 = *reinterpret_cast<const uint4*>(ws + woff1 + ((wpar1 | (chunk ^ wsw1)) << 4));
			Mma<T>::run(a0, b0, acc[0][0]);
			Mma<T>::run(a0, b1, acc[0][1]);
			Mma<T>::run(a1, b0, acc[1][0]);
			Mma<T>::run(a1, b1, acc[1][1]);
		}

		if (has_next) store_w((q + 1) & 1);
		if (next_x) store_x((cib + 1) & 1);
		if (TWO_LEVEL) {
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j)
#pragma unroll
					for (int k = 0; k < 16; ++k) { total[i][j][k] += (double)acc[i][j][k]; acc[i][j][k] = 0.f; }
		}
		__syncthreads();
		if (last_tap) { tap = 0; ++cib; } else ++tap;
	}
	if (TWO_LEVEL) {
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int k = 0; k < 16; ++k) acc[i][j][k] = (float)total[i][j][k];
	}

	// ---------------- epilogue: bias, BN statistics, scale/shift, activation, temporal mask, coalesced store through LDS
	constexpr int OPITCH = BN * sizeof(O) + 16;  // +16 B: rows 4 apart (one lane group's registers) stay off the same banks
	char* const otile = smem;
	float* const red = reinterpret_cast<float*>(smem + BM * OPITCH);  // [2 (sum, sumsq)][2 (wm)][BN]
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
			if (h == 0) { red[(0 * 2 + wm) * BN + col] = s1; red[(1 * 2 + wm) * BN + col] = s2; }
		}
	}
	__syncthreads();
	if (p.stats && tid < BN && co0 + tid < p.Cout) {
		// per-(m tile) partial row, [mtile][2][Cout]: summed in a fixed order by bn_finalize (no atomics: run-to-run identical bits)
		double* const prow = p.stats + (int64_t)mtile * 2 * p.Cout;
		prow[co0 + tid] = (double)red[(0 * 2 + 0) * BN + tid] + (double)red[(0 * 2 + 1) * BN + tid];
		prow[p.Cout + co0 + tid] = (double)red[(1 * 2 + 0) * BN + tid] + (double)red[(1 * 2 + 1) * BN + tid];
	}
	O* const yb = reinterpret_cast<O*>(p.y) + (SK ? (int64_t)blockIdx.y * p.split_stride : 0) + (int64_t)b * p.Tout * p.Cout;
	constexpr int OEPC = 16 / sizeof(O), OCHUNKS = BN / OEPC;
	const bool vec_ok = ((p.Cout * sizeof(O)) & 15) == 0;
	for (int e = tid; e < BM * OCHUNKS; e += NTHREADS) {
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

extern "C" int convasr_conv_cout_pad(int cout) { return (cout + BN - 1) / BN * BN; }

// ------------------------------------------------------------------------------------------------ weight packing
// fwd[k][co][ci] = w[co][ci][k]: a block takes 256 consecutive (co, ci) pairs = 256 * K contiguous floats, stages them in LDS
// and writes K contiguous runs of 256 elements (the (co, ci) order is the same on both sides).
template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ fwd, int64_t pairs, int K, int64_t tap_stride) {
	extern __shared__ float tile[];  // [256 * K]
	const int64_t lin0 = (int64_t)blockIdx.x * 256;
	const int n = (int)min((int64_t)256, pairs - lin0);
	for (int j = threadIdx.x; j < n * K; j += 256) tile[j] = w[lin0 * K + j];
	__syncthreads();
	if ((int)threadIdx.x < n)
		for (int k = 0; k < K; ++k) Elem<T>::store(fwd + k * tap_stride + lin0 + threadIdx.x, tile[threadIdx.x * K + k]);
}

// dgrad[K-1-k][ci][co] = fwd[k][co][ci]: per-tap 64 x 64 tiled transpose in the compute dtype
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const T* __restrict__ fwd, T* __restrict__ dgr, int Cout, int Cin, int K, int co_pad, int ci_pad) {
	__shared__ T tile[64][66];
	const int k = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const T* src = fwd + (int64_t)k * co_pad * Cin;
	T* dst = dgr + (int64_t)(K - 1 - k) * ci_pad * Cout;
#pragma unroll 4
	for (int i = 0; i < 16; ++i) {
		const int co = co0 + ty + 4 * i, ci = ci0 + tx;
		if (co < Cout && ci < Cin) tile[ty + 4 * i][tx] = src[(int64_t)co * Cin + ci];
	}
	__syncthreads();
#pragma unroll 4
	for (int i = 0; i < 16; ++i) {
		const int ci = ci0 + ty + 4 * i, co = co0 + tx;
		if (co < Cout && ci < Cin) dst[(int64_t)ci * Cout + co] = tile[tx][ty + 4 * i];
	}
}

// 16-bit builds of the two packers with 4-byte accesses on the 2-byte side (two neighbouring elements per lane): half the wave
// instructions of the scalar versions above, which remain for fp32 and for odd channel counts.
template <typename H> __global__ __launch_bounds__(256) void pack_fwd_half2_kernel(const float* __restrict__ w, H* __restrict__ fwd, int64_t pairs, int K, int64_t tap_stride) {
	extern __shared__ float tile[];  // [512 * K]
	const int64_t lin0 = (int64_t)blockIdx.x * 512;
	const int n = (int)min((int64_t)512, pairs - lin0);  // even: pairs = Cout * Cin with Cin even
	for (int j = threadIdx.x; j < n * K; j += 256) tile[j] = w[lin0 * K + j];
	__syncthreads();
	const int t2 = threadIdx.x * 2;
	if (t2 < n)
		for (int k = 0; k < K; ++k) {
			*reinterpret_cast<unsigned*>(fwd + k * tap_stride + lin0 + t2) = pack16<H>(tile[t2 * K + k], tile[(t2 + 1) * K + k]);
		}
}

// (a transpose of raw 16-bit words: one instantiation serves bf16 and fp16)
__global__ __launch_bounds__(256) void pack_dgrad_half2_kernel(const unsigned short* __restrict__ fwd, unsigned short* __restrict__ dgr, int Cout, int Cin, int K, int co_pad, int ci_pad) {
	__shared__ unsigned short tile[64][66];
	const int k = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 element pairs x 8 rows per pass
	const unsigned short* src = fwd + (int64_t)k * co_pad * Cin;
	unsigned short* dst = dgr + (int64_t)(K - 1 - k) * ci_pad * Cout;
#pragma unroll 4
	for (int i = 0; i < 8; ++i) {
		const int co = co0 + ty + 8 * i, ci = ci0 + 2 * tx;
		if (co < Cout && ci < Cin) {
			const unsigned v = *reinterpret_cast<const unsigned*>(src + (int64_t)co * Cin + ci);
			tile[ty + 8 * i][2 * tx] = (unsigned short)(v & 0xffffu);
			tile[ty + 8 * i][2 * tx + 1] = (unsigned short)(v >> 16);
		}
	}
	__syncthreads();
#pragma unroll 4
	for (int i = 0; i < 8; ++i) {
		const int ci = ci0 + ty + 8 * i, co = co0 + 2 * tx;
		if (co < Cout && ci < Cin)
			*reinterpret_cast<unsigned*>(dst + (int64_t)ci * Cout + co) = (unsigned)tile[2 * tx][ty + 8 * i] | ((unsigned)tile[2 * tx + 1][ty + 8 * i] << 16);
	}
}

// The same transpose for MANY layers in one launch: item i = {src, dst, Cout, Cin, K, co_pad, ci_pad, first block}; a block finds its item by
// a search over the (device-resident, persistent) table.  The dense-residual networks re-pack ~110 dgrad operands per step, 2.5 M weights
// on average: a launch each ran at 1.3 TB/s (7.7 us apiece, 0.83 ms per JasperNetLarge step).
struct PackItem { const unsigned short* src; unsigned short* dst; int Cout, Cin, K, co_pad, ci_pad, first; };
__global__ __launch_bounds__(256) void pack_dgrad_grouped_kernel(const PackItem* __restrict__ items, int n_items) {
	__shared__ unsigned short tile[64][66];
	int lo = 0, hi = n_items - 1;  // the last item whose first block is <= blockIdx.x
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (items[mid].first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
	}
	const PackItem it = items[lo];
	const int local = blockIdx.x - it.first;
	const int nci = (it.Cin + 63) / 64, nco = (it.Cout + 63) / 64;
	const int ci0 = (local % nci) * 64, co0 = ((local / nci) % nco) * 64, k = local / (nci * nco);
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	const unsigned short* src = it.src + (int64_t)k * it.co_pad * it.Cin;
	unsigned short* dst = it.dst + (int64_t)(it.K - 1 - k) * it.ci_pad * it.Cout;
#pragma unroll 4
	for (int i = 0; i < 8; ++i) {
		const int co = co0 + ty + 8 * i, ci = ci0 + 2 * tx;
		if (co < it.Cout && ci < it.Cin) {
			const unsigned v = *reinterpret_cast<const unsigned*>(src + (int64_t)co * it.Cin + ci);
			tile[ty + 8 * i][2 * tx] = (unsigned short)(v & 0xffffu);
			tile[ty + 8 * i][2 * tx + 1] = (unsigned short)(v >> 16);
		}
	}
	__syncthreads();
#pragma unroll 4
	for (int i = 0; i < 8; ++i) {
		const int ci = ci0 + ty + 8 * i, co = co0 + 2 * tx;
		if (co < it.Cout && ci < it.Cin)
			*reinterpret_cast<unsigned*>(dst + (int64_t)ci * it.Cout + co) = (unsigned)tile[2 * tx][ty + 8 * i] | ((unsigned)tile[2 * tx + 1][ty + 8 * i] << 16);
	}
}

extern "C" int convasr_pack_dgrad_item_bytes(void) { return (int)sizeof(PackItem); }
// items: n_items records of convasr_pack_dgrad_item_bytes() bytes in DEVICE memory, laid out as {src pointer, dst pointer, Cout, Cin, K,
// co_pad, ci_pad, first block} (two 64-bit words, six 32-bit ones); total_blocks = the sum of K * ceil(Cout / 64) * ceil(Cin / 64).
extern "C" int convasr_pack_dgrad_grouped(const void* items, int n_items, int total_blocks, void* stream) {
	CONVASR_CHECK_ARG(items && n_items > 0 && total_blocks > 0, "pack_dgrad_grouped: bad arguments");
	hipLaunchKernelGGL(pack_dgrad_grouped_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items, n_items);
	CONVASR_CHECK_LAUNCH("pack_dgrad_grouped");
	return 0;
}

// K-major master weights (CONVASR_W_KMAJOR: w[k][co][ci], the layout of the MI355X training arena) -> packed forward copy: the
// element order is already the packed one, so this is a streaming cast (8 elements per lane) into rows < Cout of each tap.
template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_kmajor_kernel(const float* __restrict__ w, T* __restrict__ fwd, int64_t per_tap, int64_t tap_stride, int K) {
	const int64_t n8 = per_tap >> 3;  // per_tap = Cout * Cin, a multiple of 8 (checked by the launcher)
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8 * K; i += (int64_t)gridDim.x * 256) {
		const int64_t k = i / n8, j = (i - k * n8) << 3;
		float v[8];
		load8<float>(w + k * per_tap + j, v);
		store8<T>(fwd + k * tap_stride + j, v);
	}
}

template <typename T> static int launch_pack(const float* w, void* fwd, void* dgr, int Cout, int Cin, int K, int w_layout, hipStream_t s) {
	const int co_pad = convasr_conv_cout_pad(Cout), ci_pad = convasr_conv_cout_pad(Cin);
	const int64_t pairs = (int64_t)Cout * Cin;
	const bool x2 = sizeof(T) == 2 && (Cin & 1) == 0 && (Cout & 1) == 0 && (size_t)512 * K * sizeof(float) <= 64 * 1024;
	if (w != nullptr) {
		if (w_layout == CONVASR_W_KMAJOR) {
			if ((pairs & 7) != 0) return convasr_fail(CONVASR_EUNSUPPORTED, "pack_conv_weight: K-major source needs Cout * Cin %% 8 == 0");
			if ((const void*)w != (const void*)fwd || sizeof(T) != 4 || co_pad != Cout) {  // (an fp32 compute copy with no row padding IS the master: nothing to do)
				int64_t blocks = ceil_div64((pairs >> 3) * K, 256);
				if (blocks > 2048) blocks = 2048;
				hipLaunchKernelGGL((pack_fwd_kmajor_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, w, (T*)fwd, pairs, (int64_t)co_pad * Cin, K);
			}
		} else if (x2) { if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((pack_fwd_half2_kernel<T>), dim3((unsigned)ceil_div64(pairs, 512)), dim3(256), (size_t)512 * K * sizeof(float), s, w, (T*)fwd, pairs, K, (int64_t)co_pad * Cin); }
		else hipLaunchKernelGGL((pack_fwd_kernel<T>), dim3((unsigned)ceil_div64(pairs, 256)), dim3(256), (size_t)256 * K * sizeof(float), s, w, (T*)fwd, pairs, K, (int64_t)co_pad * Cin);
	}
	if (dgr) {
		if (x2) hipLaunchKernelGGL(pack_dgrad_half2_kernel, dim3((Cin + 63) / 64, (Cout + 63) / 64, K), dim3(256), 0, s, (const unsigned short*)fwd, (unsigned short*)dgr, Cout, Cin, K, co_pad, ci_pad);
		else hipLaunchKernelGGL((pack_dgrad_kernel<T>), dim3((Cin + 63) / 64, (Cout + 63) / 64, K), dim3(256), 0, s, (const T*)fwd, (T*)dgr, Cout, Cin, K, co_pad, ci_pad);
	}
	return 0;
}

extern "C" int convasr_pack_conv_weight(const float* w, void* packed_fwd, void* packed_dgrad, int dtype, int Cout, int Cin, int K, int w_layout, void* stream) {
	CONVASR_CHECK_ARG(packed_fwd && (w || packed_dgrad) && Cout > 0 && Cin > 0 && K > 0 && K <= 64 && (w_layout == CONVASR_W_REFERENCE || w_layout == CONVASR_W_KMAJOR), "pack_conv_weight: bad arguments (packed_fwd is required; packed_dgrad is derived from it; w NULL = packed_fwd is current)");
	int rc;
	if (dtype == CONVASR_F32) rc = launch_pack<float>(w, packed_fwd, packed_dgrad, Cout, Cin, K, w_layout, (hipStream_t)stream);
	else if (dtype == CONVASR_BF16) rc = launch_pack<bf16_t>(w, packed_fwd, packed_dgrad, Cout, Cin, K, w_layout, (hipStream_t)stream);
	else if (dtype == CONVASR_F16) rc = launch_pack<f16_t>(w, packed_fwd, packed_dgrad, Cout, Cin, K, w_layout, (hipStream_t)stream);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "pack_conv_weight: dtype %d", dtype);
	if (rc) return rc;
	CONVASR_CHECK_LAUNCH("pack_conv_weight");
	return 0;
}

// ------------------------------------------------------------------------------------------------ forward / dgrad launcher
template <typename T, typename O, int XI, bool AL, bool SK = false> static int launch_conv(const ConvParams& p, size_t smem, hipStream_t s) {
	auto kern = conv1d_igemm_kernel<T, O, XI, AL, SK>;
	static unsigned long long attr_set = 0;
	convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), attr_set);
	constexpr int BK = ROW_BYTES / sizeof(T);
	const int splits = SK ? ((p.Cin + BK - 1) / BK + p.cib_per_split - 1) / p.cib_per_split : 1;
	hipLaunchKernelGGL(kern, dim3(p.total_tiles, splits), dim3(NTHREADS), smem, s, p);
	return 0;
}

template <typename T, typename O, bool SK = false> static int dispatch_conv(const ConvParams& p, size_t smem, hipStream_t s) {
	const int xi = (p.x_rows * 8 + NTHREADS - 1) / NTHREADS;
	const bool al = ((p.Cin * sizeof(T)) & 15) == 0;
	if (!al) {
		if (xi <= 9) return launch_conv<T, O, 9, false, SK>(p, smem, s);
		return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d: halo too large (x_rows %d)", p.x_rows);
	}
	if (xi <= 5) return launch_conv<T, O, 5, true, SK>(p, smem, s);
	if (xi <= 9) return launch_conv<T, O, 9, true, SK>(p, smem, s);
	if (xi <= 16) return launch_conv<T, O, 16, true, SK>(p, smem, s);
	return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d: halo too large (x_rows %d)", p.x_rows);
}

int convasr_conv1d_v2_try(ConvParams p, int x_dtype, int y_dtype, hipStream_t s, int* m_tiles_out);  // conv_v2s.hip
int convasr_conv1x1_try(ConvParams p, int x_dtype, int y_dtype, hipStream_t s, int* rows_out);        // conv1x1.hip
int convasr_wgrad_v2_try(WgradParams& p, int dtype, hipStream_t s);      // wgrad_v2.hip
int convasr_wgrad_v2_supports(const WgradParams& p);
static int g_conv_use_v2 = 1;
static int g_conv_debug = 0;
// test / A-B hook: bit 0 clear forces the register-staged kernels for every dtype; bits 8.. are experiment flags (ConvParams::debug)
int convasr_conv_debug_bits() { return g_conv_debug; }  // (for the other translation units' host entries)
extern "C" int convasr_debug_set_conv_v2(int enable) { const int prev = g_conv_use_v2; g_conv_use_v2 = enable & 1; g_conv_debug = enable >> 8; return prev; }

static int conv1d_run(const void* x, const void* wp, void* y, int x_dtype, int y_dtype, int B, int Cin, int Cout, int Tin, int Tout, int K,
                      int stride, int dil, int pad, const float* bias, double* stats, const float* scale, const float* shift, int act,
                      float act_lo, float act_hi, const float* xlen, const ConvParams* bn_fusion, int* rows_out, void* stream) {
	CONVASR_CHECK_ARG(x && wp && y && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && Tout > 0 && K > 0 && stride > 0 && dil > 0, "conv1d_fwd: bad arguments");
	CONVASR_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv1d_fwd: scale and shift go together");
	const int64_t expect = ((int64_t)Tin + 2 * (int64_t)pad - (int64_t)dil * (K - 1) - 1) / stride + 1;
	// a caller may ask for the first Tout < expect frames only (the stride-2 fold below needs one frame less than its even folded kernel yields)
	CONVASR_CHECK_ARG(Tout <= expect, "conv1d_fwd: Tout %d inconsistent with Tin %d K %d stride %d dil %d pad %d (at most %lld)", Tout, Tin, K, stride, dil, pad, (long long)expect);
	CONVASR_CHECK_ARG(x_dtype == CONVASR_F32 || convasr_is_half(x_dtype), "conv1d_fwd: x dtype %d", x_dtype);
	// A one-tap, stride-1, unpadded conv has no temporal coupling: without per-utterance length masks the batch is ONE sequence of
	// B * T frames, and the tiles need not stop at utterance ends (32 utterances of 626 frames: 79 tiles of 256 rows instead of 96).
	// Same k order per element; the BN partial rows are grouped by the new tiles.  (debug bit 512: off, A/B runs)
	if (K == 1 && stride == 1 && pad == 0 && Tin == Tout && B > 1 && !xlen && !(bn_fusion && bn_fusion->bn_xlen) && !(g_conv_debug & 512) &&
	    (int64_t)B * Tin * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31)) {
		Tin = Tout = B * Tin;
		B = 1;
	}
	ConvParams p = {};
	if (bn_fusion) p = *bn_fusion;  // only the bn_* fields are set in it
	p.x = x; p.w = wp; p.y = y; p.bias = bias; p.stats = stats; p.scale = scale; p.shift = shift; p.xlen = xlen;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutPad = convasr_conv_cout_pad(Cout); p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.dil = dil; p.pad = pad;
	p.act = act; p.act_lo = act_lo; p.act_hi = act_hi; p.debug = g_conv_debug;
	p.m_tiles_per_b = (Tout + BM - 1) / BM;
	p.n_tiles = p.CoutPad / BN;
	p.total_tiles = B * p.m_tiles_per_b * p.n_tiles;
	int xr = (BM - 1) * stride + (K - 1) * dil + 1;
	p.x_rows = (xr + 1) & ~1;
	const size_t osz = y_dtype == CONVASR_F32 ? 4 : 2;
	size_t smem = 2 * (size_t)p.x_rows * ROW_BYTES + 2 * BN * ROW_BYTES;
	const size_t epi = (size_t)BM * (BN * osz + 16) + 4 * BN * sizeof(float);
	if (epi > smem) smem = epi;
	CONVASR_CHECK_ARG(smem <= 160 * 1024, "conv1d_fwd: tile needs %zu B of LDS", smem);
	hipStream_t s = (hipStream_t)stream;
	int v2_rows = 0;
	if (convasr_is_half(x_dtype) && g_conv_use_v2 && convasr_conv1x1_try(p, x_dtype, y_dtype, s, &v2_rows)) {  // one-tap training launches with short reductions
		CONVASR_CHECK_LAUNCH("conv1d_fwd (1x1)");
		if (rows_out) *rows_out = v2_rows;
		return 0;
	}
	if (convasr_is_half(x_dtype) && g_conv_use_v2 && convasr_conv1d_v2_try(p, x_dtype, y_dtype, s, &v2_rows)) {
		CONVASR_CHECK_LAUNCH("conv1d_fwd (v2)");
		if (rows_out) *rows_out = v2_rows;
		return 0;
	}
	if (p.bn_y) return 1;  // the fused epilogue lives in the LDS-DMA kernel only: nothing was launched, the caller runs the two steps apart
	int rc;
	if (x_dtype == CONVASR_F32 && y_dtype == CONVASR_F32) rc = dispatch_conv<float, float>(p, smem, s);
	else if (x_dtype == CONVASR_BF16 && y_dtype == CONVASR_BF16) rc = dispatch_conv<bf16_t, bf16_t>(p, smem, s);
	else if (x_dtype == CONVASR_BF16 && y_dtype == CONVASR_F32) rc = dispatch_conv<bf16_t, float>(p, smem, s);
	else if (x_dtype == CONVASR_F16 && y_dtype == CONVASR_F16) rc = dispatch_conv<f16_t, f16_t>(p, smem, s);
	else if (x_dtype == CONVASR_F16 && y_dtype == CONVASR_F32) rc = dispatch_conv<f16_t, float>(p, smem, s);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_fwd: dtype %d -> %d", x_dtype, y_dtype);
	if (rc) return rc;
	CONVASR_CHECK_LAUNCH("conv1d_fwd");
	if (rows_out) *rows_out = B * p.m_tiles_per_b;
	return 0;
}

extern "C" int convasr_conv1d_fwd(const void* x, const void* wp, void* y, int x_dtype, int y_dtype, int B, int Cin, int Cout, int Tin, int Tout, int K,
                                  int stride, int dil, int pad, const float* bias, double* stats, const float* scale, const float* shift, int act,
                                  float act_lo, float act_hi, const float* xlen, int* stats_rows, void* stream) {
	CONVASR_CHECK_ARG(!stats || stats_rows, "conv1d_fwd: stats needs stats_rows");
	return conv1d_run(x, wp, y, x_dtype, y_dtype, B, Cin, Cout, Tin, Tout, K, stride, dil, pad, bias, stats, scale, shift, act, act_lo, act_hi, xlen, nullptr, stats_rows, stream);
}

extern "C" int convasr_conv_stats_max_rows(int B, int Tout) { return B * ((Tout + BM - 1) / BM); }

// ------------------------------------------------------------------------------------------------ split-K forward (launches of a few tiles)
// One request of online inference (transcribe.py:140, benchmark_online.py:125: B = 1 x 6 s) is 2 m tiles x (Cout / 128) n tiles per layer: 4-16
// workgroups on 256 CUs, each reducing ALL of Cin x K on its own -- the launch takes a whole tile's K loop (~60-110 us) whatever the chip could do.
// Split-K over the 64-channel input blocks: workgroup (tile, split) reduces its blocks and stores an fp32 partial tile; a second, streaming
// kernel adds the partials IN SPLIT ORDER (deterministic) and runs the epilogue the unsplit kernel runs in its own tail (bias, folded scale /
// shift, activation, length mask, rounding to the storage type) -- same formulas, the sum associated by input block instead of running through.
static int splitk_plan(int x_dtype, int B, int Cin, int Cout, int Tout, int* cib_per_split) {
	// 16-bit storage: the LDS-DMA kernel, 256-row tiles, 64-channel input blocks; fp32 (the exact-fp32 parity path): the register-staged kernel,
	// 128-row tiles, 32-channel slabs
	const bool half = convasr_is_half(x_dtype);
	const int n_cu = convasr_cu_count();
	const int n_cib = half ? Cin >> 6 : (Cin + 31) / 32;
	const int tiles = B * ((Tout + (half ? 255 : BM - 1)) / (half ? 256 : BM)) * (convasr_conv_cout_pad(Cout) / BN);
	if ((half && (Cin & 63)) || n_cib < 2 || tiles * 4 > n_cu) return 1;  // (from a quarter of the CUs up the unsplit launch is the better one: every split pays a tile's prologue and epilogue)
	int want = n_cu / tiles;  // about one workgroup per CU
	if (want > n_cib) want = n_cib;
	if (want > 32) want = 32;
	const int cps = (n_cib + want - 1) / want;
	*cib_per_split = cps;
	return (n_cib + cps - 1) / cps;
}

template <typename O> __global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int splits, int64_t split_stride, O* __restrict__ y, const float* __restrict__ bias,
                                                                                     const float* __restrict__ scale, const float* __restrict__ shift, int act, float lo, float hi, const float* __restrict__ xlen, int B, int T, int C) {
	const ActConst ac = act_const(act, lo, hi);
	const int c8 = C >> 3;
	const int64_t n = (int64_t)B * T * c8;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		const int64_t row = i / c8;
		const int c = (int)(i - row * c8) << 3, b = (int)(row / T), t = (int)(row - (int64_t)b * T);
		float v[8], a[8];
		double acc[8];  // (the partials are added in fp64: the exact-fp32 path's two-level sum keeps its last bits, and the pass is bound by its loads)
		load8<float>(part + row * C + c, a);
#pragma unroll
		for (int k = 0; k < 8; ++k) acc[k] = (double)a[k];
		for (int s = 1; s < splits; ++s) {
			load8<float>(part + s * split_stride + row * C + c, a);
#pragma unroll
			for (int k = 0; k < 8; ++k) acc[k] += (double)a[k];
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (float)acc[k];
		const bool live = t < valid_len(xlen, b, T);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			float val = v[k] + (bias ? bias[c + k] : 0.f);
			val = apply_act(scale ? val * scale[c + k] + shift[c + k] : val, ac);
			v[k] = live ? val : 0.f;
		}
		store8<O>(y + row * C + c, v);
	}
}

// Splits this launch would be cut into (1: not worth it / outside the kernel's envelope -- call convasr_conv1d_fwd) and the fp32 workspace it needs.
extern "C" int convasr_conv1d_fwd_splitk_plan(int x_dtype, int B, int Cin, int Cout, int Tout, int K, int64_t* workspace_bytes) {
	int cps = 0;
	if (!(convasr_is_half(x_dtype) || x_dtype == CONVASR_F32) || B <= 0 || Cin <= 0 || Cout <= 0 || Tout <= 0 || K <= 0 || (Cout & 7) || (convasr_is_half(x_dtype) && !g_conv_use_v2)) return 1;
	const int splits = splitk_plan(x_dtype, B, Cin, Cout, Tout, &cps);
	if (workspace_bytes) *workspace_bytes = splits > 1 ? (int64_t)splits * B * Tout * Cout * 4 : 0;
	return splits;
}

// convasr_conv1d_fwd (stride 1, no statistics) as a split-K launch + the epilogue pass; `splits` = convasr_conv1d_fwd_splitk_plan's answer (>= 2).
extern "C" int convasr_conv1d_fwd_splitk(const void* x, const void* wp, void* y, int x_dtype, int y_dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int pad,
                                         const float* bias, const float* scale, const float* shift, int act, float act_lo, float act_hi, const float* xlen, int splits, void* workspace, void* stream) {
	const bool half = convasr_is_half(x_dtype);
	CONVASR_CHECK_ARG(x && wp && y && workspace && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && Tout > 0 && K > 0 && dil > 0 && (Cout & 7) == 0 && (half ? (y_dtype == x_dtype || y_dtype == CONVASR_F32) : (x_dtype == CONVASR_F32 && y_dtype == CONVASR_F32)), "conv1d_fwd_splitk: bad arguments (16-bit or fp32 input, Cout %% 8 == 0)");
	CONVASR_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv1d_fwd_splitk: scale and shift go together");
	CONVASR_CHECK_ARG(Tout <= (int64_t)Tin + 2 * (int64_t)pad - (int64_t)dil * (K - 1), "conv1d_fwd_splitk: Tout %d inconsistent with Tin %d K %d dil %d pad %d", Tout, Tin, K, dil, pad);
	int cps = 0;
	if (splitk_plan(x_dtype, B, Cin, Cout, Tout, &cps) != splits || splits < 2) return convasr_fail(CONVASR_EINVAL, "conv1d_fwd_splitk: splits %d is not this geometry's plan", splits);
	ConvParams p = {};
	p.x = x; p.w = wp; p.y = workspace;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutPad = convasr_conv_cout_pad(Cout); p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = 1; p.dil = dil; p.pad = pad;
	p.act = CONVASR_ACT_NONE; p.debug = g_conv_debug | 2048;  // (2048: no 128-row tiles for the whole launch -- the splits fill the chip, and a 128-row tile costs 1.2-1.4x per FLOP)
	p.n_tiles = p.CoutPad / BN;
	p.cib_per_split = cps; p.split_stride = (long long)B * Tout * Cout;
	hipStream_t s = (hipStream_t)stream;
	if (half) {
		if (!convasr_conv1d_v2_try(p, x_dtype, CONVASR_F32, s, nullptr)) return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_fwd_splitk: the geometry is outside the LDS-DMA kernel's envelope");
	} else {  // the exact-fp32 kernel (two-level sums: fp64 totals per workgroup, rounded once into its fp32 partial tile)
		p.m_tiles_per_b = (Tout + BM - 1) / BM;
		p.total_tiles = B * p.m_tiles_per_b * p.n_tiles;
		p.x_rows = ((BM - 1) + (K - 1) * dil + 1 + 1) & ~1;
		size_t smem = 2 * (size_t)p.x_rows * ROW_BYTES + 2 * BN * ROW_BYTES;
		const size_t epi = (size_t)BM * (BN * 4 + 16) + 4 * BN * sizeof(float);
		if (epi > smem) smem = epi;
		if (smem > 160 * 1024) return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_fwd_splitk: tile needs %zu B of LDS", smem);
		const int rc = dispatch_conv<float, float, true>(p, smem, s);
		if (rc) return rc;
	}
	CONVASR_CHECK_LAUNCH("conv1d_fwd_splitk");
	int64_t blocks = ceil_div64((int64_t)B * Tout * (Cout >> 3), 256);
	if (blocks > 4096) blocks = 4096;
	const dim3 g((unsigned)blocks), t(256);
	const float* part = (const float*)workspace;
	if (y_dtype == CONVASR_F32) hipLaunchKernelGGL((splitk_epilogue_kernel<float>), g, t, 0, s, part, splits, p.split_stride, (float*)y, bias, scale, shift, act, act_lo, act_hi, xlen, B, Tout, Cout);
	else if (y_dtype == CONVASR_F16) hipLaunchKernelGGL((splitk_epilogue_kernel<f16_t>), g, t, 0, s, part, splits, p.split_stride, (f16_t*)y, bias, scale, shift, act, act_lo, act_hi, xlen, B, Tout, Cout);
	else hipLaunchKernelGGL((splitk_epilogue_kernel<bf16_t>), g, t, 0, s, part, splits, p.split_stride, (bf16_t*)y, bias, scale, shift, act, act_lo, act_hi, xlen, B, Tout, Cout);
	CONVASR_CHECK_LAUNCH("conv1d_fwd_splitk (epilogue)");
	return 0;
}

extern "C" int convasr_conv1d_dgrad_bn_reduce(const void* dy, const void* packed_dgrad, void* dx, int dtype, int B, int Cout, int Cin, int T_dy, int T_dx, int K, int dil, int pad,
                                              const void* bn_y, const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                                              int bn_act, float bn_act_lo, float bn_act_hi, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* step_key,
                                              const float* bn_xlen, double* bn_sums, int* bn_rows, const uint8_t* bn_gate, void* stream) {
	CONVASR_CHECK_ARG(!bn_gate || bn_act == CONVASR_ACT_RELU || bn_act == CONVASR_ACT_HARDTANH || bn_act == CONVASR_ACT_NONE, "conv1d_dgrad_bn_reduce: the one-bit gate needs an activation whose derivative is 0 or 1");
	CONVASR_CHECK_ARG(bn_y && bn_scale && bn_shift && bn_mean && bn_invstd && bn_sums && bn_rows && dropout_p >= 0.f && dropout_p < 1.f && (Cin & 7) == 0 && convasr_is_half(dtype), "conv1d_dgrad_bn_reduce: bad arguments (dtype must be CONVASR_BF16 or CONVASR_F16)");
	ConvParams f = {};
	f.bn_y = bn_y; f.bn_scale = bn_scale; f.bn_shift = bn_shift; f.bn_mean = bn_mean; f.bn_invstd = bn_invstd; f.bn_xlen = bn_xlen; f.bn_sums = bn_sums;
	f.bn_act = bn_act; f.bn_lo = bn_act_lo; f.bn_hi = bn_act_hi; f.bn_seed = convasr_mix_seed(seed); f.bn_offset = offset; f.bn_step_key = step_key; f.bn_gate = bn_gate;
	f.bn_drop_thr = (unsigned)lrintf(dropout_p * 65536.f);
	if (f.bn_drop_thr > 65535u) f.bn_drop_thr = 65535u;
	f.bn_keep_scale = 65536.f / (float)(65536u - f.bn_drop_thr);
	// dgrad = the forward kernel on (dy, flipped packed weights): channels in = Cout, channels out = Cin, stride 1
	return conv1d_run(dy, packed_dgrad, dx, dtype, dtype, B, Cout, Cin, T_dy, T_dx, K, 1, dil, pad, nullptr, nullptr, nullptr, nullptr, CONVASR_ACT_NONE, 0.f, 0.f, nullptr, &f, bn_rows, stream);
}

// ------------------------------------------------------------------------------------------------ wgrad
// M = co (A = dY^T), N = ci (B = X), reduction over (b, t).  Both operands are row-major [t][c] tiles in LDS; the MFMA wants
// 8 consecutive t per lane, i.e. a column read: bf16 uses ds_read_b64_tr_b16 (hardware 4x16 transpose), fp32 plain b32 reads.
// One workgroup = one (128 co x 128 ci) tile x up to TG taps (the taps share the staged dY rows and the X rows + halo),
// over one split of the (b, t) axis; partial tiles go to fp32 slabs [split][tap][co][ci], summed (and transposed to the
// reference's (Cout, Cin, K) parameter layout) by wgrad_reduce_kernel -- deterministic, no float atomics.
template <typename T> struct WgTile;
template <> struct WgTile<bf16_t> { static constexpr int BKT = 64, PITCH = 128 * 2 + 64, CPR = 16; };  // rows of 256 B + 64 B pad: tr reads conflict-free
template <> struct WgTile<f16_t> : WgTile<bf16_t> {};
template <> struct WgTile<float> { static constexpr int BKT = 32, PITCH = 128 * 4 + 16, CPR = 32; };

template <typename T, int XI, bool AL_X, bool AL_Y>
__global__ __launch_bounds__(NTHREADS, 1) void conv1d_wgrad_kernel(WgradParams p) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int EPC = 16 / sizeof(T);
	constexpr int BKT = WgTile<T>::BKT, PITCH = WgTile<T>::PITCH, CPR = WgTile<T>::CPR;
	constexpr int YI = BKT * CPR / NTHREADS;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;

	const int v = xcd_remap(blockIdx.x, p.units * p.splits);
	const int unit = v % p.units, split = v / p.units;
	const int tg = unit % p.tap_groups, ci_t = (unit / p.tap_groups) % p.ci_tiles, co_t = unit / (p.tap_groups * p.ci_tiles);
	const int co0 = co_t * 128, ci0 = ci_t * 128, tap0 = tg * WG_TG;
	const int ntaps = min(WG_TG, p.K - tap0);
	const int c_begin = split * p.chunks_per_split, c_end = min(p.total_chunks, c_begin + p.chunks_per_split);

	const int ybytes = BKT * PITCH, xbytes = p.x_rows * PITCH;
	char* const ybuf = smem;
	char* const xbuf = smem + 2 * ybytes;
	const int total_x_chunks = p.x_rows * CPR;

	uint4 yreg[YI];
	uint4 xreg[XI];
	auto load_tiles = [&](int c) {
		const int b = c / p.chunks_per_b, t0 = (c % p.chunks_per_b) * BKT;
		const T* yb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.Tout * p.Cout;
		const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.Tin * p.Cin;
		const int tin0 = t0 * p.stride - p.pad + tap0 * p.dil;
#pragma unroll
		for (int i = 0; i < YI; ++i) {
			const int e = tid + NTHREADS * i, row = e / CPR, ch = e % CPR, t = t0 + row;
			yreg[i] = make_uint4(0, 0, 0, 0);
			if (t < p.Tout) yreg[i] = load_chunk<T, AL_Y>(yb + (int64_t)t * p.Cout, co0 + ch * EPC, p.Cout);
		}
#pragma unroll
		for (int i = 0; i < XI; ++i) {
			const int e = tid + NTHREADS * i, row = e / CPR, ch = e % CPR, tin = tin0 + row;
			xreg[i] = make_uint4(0, 0, 0, 0);
			if (e < total_x_chunks && tin >= 0 && tin < p.Tin) xreg[i] = load_chunk<T, AL_X>(xb + (int64_t)tin * p.Cin, ci0 + ch * EPC, p.Cin);
		}
	};
	auto store_tiles = [&](int buf) {
#pragma unroll
		for (int i = 0; i < YI; ++i) {
			const int e = tid + NTHREADS * i;
			*reinterpret_cast<uint4*>(ybuf + buf * ybytes + (e / CPR) * PITCH + (e % CPR) * 16) = yreg[i];
		}
#pragma unroll
		for (int i = 0; i < XI; ++i) {
			const int e = tid + NTHREADS * i;
			if (e < total_x_chunks) *reinterpret_cast<uint4*>(xbuf + buf * xbytes + (e / CPR) * PITCH + (e % CPR) * 16) = xreg[i];
		}
	};

	f32x16 acc[WG_TG][2][2];
#pragma unroll
	for (int a = 0; a < WG_TG; ++a)
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int k = 0; k < 16; ++k) acc[a][i][j][k] = 0.f;

	if (c_begin < c_end) {
		load_tiles(c_begin);
		store_tiles(0);
	}
	__syncthreads();

	for (int c = c_begin; c < c_end; ++c) {
		const int buf = (c - c_begin) & 1;
		if (c + 1 < c_end) load_tiles(c + 1);
		const char* ys = ybuf + buf * ybytes;
		const char* xs = xbuf + buf * xbytes;
		if constexpr (sizeof(T) == 2) {
			// lane -> (16-lane group g4, q = row within the 4-row block, pc = 4-column piece)
			const int g4 = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
			const int kq = 8 * (g4 >> 1) + q, colb = (g4 & 1) * 16 + 4 * pc;
#pragma unroll
			for (int kk = 0; kk < BKT / 16; ++kk) {
				uint4 a[2];
#pragma unroll
				for (int mi = 0; mi < 2; ++mi) {
					const char* ap = ys + (kk * 16 + kq) * PITCH + (wm * 64 + mi * 32 + colb) * 2;
					s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
					s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 4 * PITCH));
					a[mi] = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
				}
#pragma unroll
				for (int tg_i = 0; tg_i < WG_TG; ++tg_i) {
					if (tg_i < ntaps) {
						uint4 bb[2];
#pragma unroll
						for (int ni = 0; ni < 2; ++ni) {
							const char* bp = xs + ((kk * 16 + kq) * p.stride + tg_i * p.dil) * PITCH + (wn * 64 + ni * 32 + colb) * 2;
							s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bp));
							s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bp + 4 * p.stride * PITCH));
							bb[ni] = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
						}
#pragma unroll
						for (int mi = 0; mi < 2; ++mi)
#pragma unroll
							for (int ni = 0; ni < 2; ++ni) Mma<T>::run(a[mi], bb[ni], acc[tg_i][mi][ni]);
					}
				}
			}
		} else {
			const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
			for (int k2 = 0; k2 < BKT / 2; ++k2) {
				float a[2];
#pragma unroll
				for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const float*>(ys + (k2 * 2 + h) * PITCH + (wm * 64 + mi * 32 + r) * 4);
#pragma unroll
				for (int tg_i = 0; tg_i < WG_TG; ++tg_i) {
					if (tg_i < ntaps) {
						float bb[2];
#pragma unroll
						for (int ni = 0; ni < 2; ++ni) bb[ni] = *reinterpret_cast<const float*>(xs + ((k2 * 2 + h) * p.stride + tg_i * p.dil) * PITCH + (wn * 64 + ni * 32 + r) * 4);
#pragma unroll
						for (int mi = 0; mi < 2; ++mi)
#pragma unroll
							for (int ni = 0; ni < 2; ++ni) acc[tg_i][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bb[ni], acc[tg_i][mi][ni], 0, 0, 0);
					}
				}
			}
		}
		if (c + 1 < c_end) store_tiles(buf ^ 1);
		__syncthreads();
	}

	const int r = lane & 31, h = lane >> 5;
#pragma unroll
	for (int tg_i = 0; tg_i < WG_TG; ++tg_i) {
		if (tg_i < ntaps) {
			float* sl = p.slab + ((int64_t)split * p.K + tap0 + tg_i) * p.Cout * p.Cin;
#pragma unroll
			for (int mi = 0; mi < 2; ++mi)
#pragma unroll
				for (int ni = 0; ni < 2; ++ni) {
					const int ci = ci0 + wn * 64 + ni * 32 + r;
#pragma unroll
					for (int g = 0; g < 16; ++g) {
						const int co = co0 + wm * 64 + mi * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
						if (co < p.Cout && ci < p.Cin) sl[(int64_t)co * p.Cin + ci] = acc[tg_i][mi][ni][g];
					}
				}
		}
	}
}

// dw[co][ci][k] (+)= sum_s slab[s][k][co][ci]: coalesced reads along ci, LDS transpose, coalesced writes along (ci, k).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int K, int Cout, int Cin, int accumulate) {
	extern __shared__ float rbuf[];  // [K][65]
	const int co = blockIdx.y, ci0 = blockIdx.x * 64;
	const int c = threadIdx.x & 63;
	const int64_t plane = (int64_t)Cout * Cin;
	for (int k = threadIdx.x >> 6; k < K; k += 4) {
		float a = 0.f;
		if (ci0 + c < Cin)
			for (int s = 0; s < S; ++s) a += slab[((int64_t)s * K + k) * plane + (int64_t)co * Cin + ci0 + c];
		rbuf[k * 65 + c] = a;
	}
	__syncthreads();
	const int ncols = min(64, Cin - ci0);
	float* out = dw + ((int64_t)co * Cin + ci0) * K;
	for (int o = threadIdx.x; o < ncols * K; o += 256) {
		const float val = rbuf[(o % K) * 65 + o / K];
		out[o] = accumulate ? out[o] + val : val;
	}
}

// K-major gradient (CONVASR_W_KMAJOR): dw[k][co][ci] has the slabs' own element order, so the combine is a streaming sum of S
// slabs, 16 bytes per lane, no transpose.
__global__ __launch_bounds__(256) void wgrad_reduce_kmajor_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int64_t n4, int accumulate) {
	const float4* const s4 = reinterpret_cast<const float4*>(slab);
	float4* const d4 = reinterpret_cast<float4*>(dw);
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
		float4 a = s4[i];
		for (int s = 1; s < S; ++s) { const float4 b = s4[(int64_t)s * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
		if (accumulate) { const float4 o = d4[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
		d4[i] = a;
	}
}

// dbias[c] (+)= sum over rows of a channels-last (rows, C) matrix: blocks own row chunks and store one partial row each; a second
// launch adds the partial rows in order (no float atomics: the bias gradient is bit-identical from run to run)
template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ y, float* __restrict__ part, int64_t rows, int C, int rows_per_block) {
	__shared__ float red[4][64];
	const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
	const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
	float a = 0.f;
	if (c < C)
		for (int64_t rr = r0 + w; rr < r1; rr += 4) a += Elem<T>::load(y + rr * C + c);
	red[w][threadIdx.x & 63] = a;
	__syncthreads();
	if (w == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out, int accumulate) {
	__shared__ double red[16][64];  // 64 channels x 16 row-lanes, combined in a fixed order
	const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, w = threadIdx.x >> 6;
	double a = 0;
	if (c < C)
		for (int i = w; i < nparts; i += 16) a += (double)part[(int64_t)i * C + c];
	red[w][cl] = a;
	__syncthreads();
	if (w != 0 || c >= C) return;
	double s = 0;
	for (int i = 0; i < 16; ++i) s += red[i][cl];
	out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// out[c] (+)= sum over rows of a channels-last (rows, C) matrix, the bias gradient of a conv on its own (the weight-gradient entry point
// forms it from the same dy it multiplies; a split-operand head multiplies dy's planes and needs the sum of dy itself)
extern "C" int64_t convasr_colsum_workspace_bytes(int64_t rows, int C) { return ceil_div64(rows, 256) * (int64_t)C * 4; }
extern "C" int convasr_colsum(const void* y, int dtype, int64_t rows, int C, float* out, void* workspace, int accumulate, void* stream) {
	CONVASR_CHECK_ARG(y && out && workspace && rows > 0 && C > 0 && (dtype == CONVASR_F32 || convasr_is_half(dtype)), "colsum: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	const int rows_per_block = 256;
	dim3 grid((C + 63) / 64, (unsigned)ceil_div64(rows, rows_per_block));
	float* part = (float*)workspace;
	if (dtype == CONVASR_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)y, part, rows, C, rows_per_block);
	else if (dtype == CONVASR_F16) hipLaunchKernelGGL((colsum_kernel<f16_t>), grid, dim3(256), 0, s, (const f16_t*)y, part, rows, C, rows_per_block);
	else hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)y, part, rows, C, rows_per_block);
	hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, (const float*)part, (int)grid.y, C, out, accumulate);
	CONVASR_CHECK_LAUNCH("colsum");
	return 0;
}

extern "C" int64_t convasr_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil) {
	WgradParams p;
	p.debug = 0;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.dil = dil;
	p.pad = 0;
	// the largest split count any of the kernels' plans would pick for this shape (bf16 LDS-DMA kernel, bf16 / fp32 general kernel)
	int splits = 1;
	const int bkt[3] = {64, WgTile<bf16_t>::BKT, WgTile<float>::BKT};
	const double us[3] = {1.6, 2.7, 11.0};
	for (int i = 0; i < 3; ++i) {
		WgradParams q = p;
		wgrad_plan(q, bkt[i], us[i]);
		if (q.splits > splits) splits = q.splits;
	}
	const int64_t slabs = (int64_t)splits * K * (int64_t)Cout * Cin * 4;
	const int64_t dbias_parts = ceil_div64((int64_t)B * Tout, 256) * Cout * 4;  // partial rows of the bias-gradient column sum reuse the buffer
	return slabs > dbias_parts ? slabs : dbias_parts;
}

template <typename T, int XI, bool AX, bool AY> static void launch_wgrad(const WgradParams& p, size_t smem, hipStream_t s) {
	auto kern = conv1d_wgrad_kernel<T, XI, AX, AY>;
	static unsigned long long attr_set = 0;
	convasr_allow_160k_lds(reinterpret_cast<const void*>(kern), attr_set);
	hipLaunchKernelGGL(kern, dim3(p.units * p.splits), dim3(NTHREADS), smem, s, p);
}

template <typename T> static int dispatch_wgrad(WgradParams& p, hipStream_t s) {
	wgrad_plan(p, WgTile<T>::BKT, sizeof(T) == 2 ? 2.7 : 11.0);
	const int xi = (p.x_rows * WgTile<T>::CPR + NTHREADS - 1) / NTHREADS;
	const size_t smem = 2 * (size_t)(WgTile<T>::BKT + p.x_rows) * WgTile<T>::PITCH;
	if (smem > 160 * 1024 || xi > 12) return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_wgrad: tile too large (x_rows %d)", p.x_rows);
	const bool ax = ((p.Cin * sizeof(T)) & 15) == 0, ay = ((p.Cout * sizeof(T)) & 15) == 0;
	if (ax && ay) { if (xi <= 5) launch_wgrad<T, 5, true, true>(p, smem, s); else launch_wgrad<T, 12, true, true>(p, smem, s); }
	else if (ax) launch_wgrad<T, 12, true, false>(p, smem, s);
	else if (ay) launch_wgrad<T, 12, false, true>(p, smem, s);
	else launch_wgrad<T, 12, false, false>(p, smem, s);
	return 0;
}

static int wgrad_impl(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, float* dbias, void* workspace, int dtype, int B, int Cin, int Cout, int Tin,
                      int Tout, int K, int stride, int dil, int pad, int accumulate, int dw_layout, void* stream) {
	CONVASR_CHECK_ARG(x && dy && dw && workspace && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && Tout > 0 && K > 0 && stride > 0 && dil > 0 && (dw_layout == CONVASR_W_REFERENCE || dw_layout == CONVASR_W_KMAJOR), "conv1d_wgrad: bad arguments");
	CONVASR_CHECK_ARG(dw_layout == CONVASR_W_REFERENCE || (((int64_t)Cout * Cin) & 3) == 0, "conv1d_wgrad: K-major dw needs Cout * Cin %% 4 == 0");
	CONVASR_CHECK_ARG(K <= 64, "conv1d_wgrad: K %d > 64", K);
	WgradParams p;
	p.debug = g_conv_debug;
	p.x = x; p.dy = dy; p.slab = (float*)workspace;
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = stride; p.dil = dil; p.pad = pad;
	p.x_ld = x_ld == Cin ? 0 : x_ld; p.dy_ld = dy_ld == Cout ? 0 : dy_ld;
	hipStream_t s = (hipStream_t)stream;
	int rc;
	if (convasr_is_half(dtype) && g_conv_use_v2 && convasr_wgrad_v2_try(p, dtype, s)) rc = 0;
	else if (p.x_ld || p.dy_ld) return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_wgrad_ld: frames %d / %d elements apart need the LDS-DMA kernel's envelope (16-bit storage, stride 1, Cin %% 128 == 0, Cout %% 128 == 0)", x_ld, dy_ld);
	else if (dtype == CONVASR_F32) rc = dispatch_wgrad<float>(p, s);
	else if (dtype == CONVASR_BF16) rc = dispatch_wgrad<bf16_t>(p, s);
	else if (dtype == CONVASR_F16) rc = dispatch_wgrad<f16_t>(p, s);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "conv1d_wgrad: dtype %d", dtype);
	if (rc) return rc;
	CONVASR_CHECK_LAUNCH("conv1d_wgrad");
	if (dw_layout == CONVASR_W_KMAJOR) {
		const int64_t n4 = (int64_t)K * Cout * Cin / 4;
		int64_t blocks = ceil_div64(n4, 256);
		if (blocks > 2048) blocks = 2048;
		hipLaunchKernelGGL(wgrad_reduce_kmajor_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)p.slab, dw, p.splits, n4, accumulate);
	} else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Cin + 63) / 64, Cout), dim3(256), (size_t)K * 65 * sizeof(float), s, p.slab, dw, p.splits, K, Cout, Cin, accumulate);
	CONVASR_CHECK_LAUNCH("conv1d_wgrad_reduce");
	if (dbias) {
		const int64_t rows = (int64_t)B * Tout;
		const int rows_per_block = 256;
		dim3 grid((Cout + 63) / 64, (unsigned)ceil_div64(rows, rows_per_block));
		float* part = p.slab;  // the split-K slabs are consumed by now (same stream): reuse the workspace for grid.y partial rows
		if (dtype == CONVASR_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)dy, part, rows, Cout, rows_per_block);
		else if (dtype == CONVASR_F16) hipLaunchKernelGGL((colsum_kernel<f16_t>), grid, dim3(256), 0, s, (const f16_t*)dy, part, rows, Cout, rows_per_block);
		else hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dy, part, rows, Cout, rows_per_block);
		hipLaunchKernelGGL(colsum_final_kernel, dim3((Cout + 63) / 64), dim3(1024), 0, s, (const float*)part, (int)grid.y, Cout, dbias, accumulate);
		CONVASR_CHECK_LAUNCH("conv1d_dbias");
	}
	return 0;
}

extern "C" int convasr_conv1d_wgrad(const void* x, const void* dy, float* dw, float* dbias, void* workspace, int dtype, int B, int Cin, int Cout, int Tin,
                                    int Tout, int K, int stride, int dil, int pad, int accumulate, int dw_layout, void* stream) {
	return wgrad_impl(x, 0, dy, 0, dw, dbias, workspace, dtype, B, Cin, Cout, Tin, Tout, K, stride, dil, pad, accumulate, dw_layout, stream);
}

// The same gradient from operands whose frames are x_ld / dy_ld elements apart (>= Cin / Cout, multiples of 8): plane 0 of a split-operand
// plane tensor [B][T][3][C] (csrc/split3.hip) read in place, ld = 3 C -- the one-product backward of a split-operand forward.
extern "C" int convasr_conv1d_wgrad_ld(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, void* workspace, int dtype, int B, int Cin, int Cout, int Tin,
                                       int Tout, int K, int dil, int pad, int accumulate, int dw_layout, void* stream) {
	CONVASR_CHECK_ARG(x_ld >= Cin && dy_ld >= Cout && (x_ld & 7) == 0 && (dy_ld & 7) == 0 && convasr_is_half(dtype), "conv1d_wgrad_ld: x_ld >= Cin, dy_ld >= Cout, multiples of 8, 16-bit storage");
	return wgrad_impl(x, x_ld, dy, dy_ld, dw, nullptr, workspace, dtype, B, Cin, Cout, Tin, Tout, K, 1, dil, pad, accumulate, dw_layout, stream);
}

// Is this geometry inside convasr_conv1d_wgrad_ld's envelope (1) or must the caller make the operands dense first (0)?  No launch, no GPU.
extern "C" int convasr_conv1d_wgrad_ld_supported(int dtype, int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int x_ld, int dy_ld) {
	if (!convasr_is_half(dtype) || B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0 || K <= 0 || K > 64 || dil <= 0 || x_ld < Cin || dy_ld < Cout) return 0;
	WgradParams p = WgradParams();
	p.B = B; p.Cin = Cin; p.Cout = Cout; p.Tin = Tin; p.Tout = Tout; p.K = K; p.stride = 1; p.dil = dil;
	p.x_ld = x_ld == Cin ? 0 : x_ld; p.dy_ld = dy_ld == Cout ? 0 : dy_ld;
	return g_conv_use_v2 && convasr_wgrad_v2_supports(p);
}

// ------------------------------------------------------------------------------------------------ stride-2 fold
// A stride-2, dilation-1 conv over an even number of frames is a stride-1 conv over the same memory read as (T / 2) rows of
// 2 Cin channels (row r = frames 2r, 2r + 1 side by side: a view, no copy): with P' = ceil(pad / 2), s0 = 2 P' - pad and
// K' = (K - 1 + s0) / 2 + 1,   y[t] = sum_{j < K'} sum_{p < 2} sum_ci  V[t + j - P'][p Cin + ci] * w[co][ci][2 j + p - s0]
// (taps outside [0, K) are zero weights).  The prologue conv of every model (models.py:312: K = 11, stride 2, 64 mel channels) becomes
// a K' = 6, 128-channel stride-1 problem, which is inside the envelope of the LDS-DMA kernels (conv_v2s.hip, wgrad_v2.hip).
static bool fold2_geometry(int K, int pad, int* Kf, int* Pf, int* s0) {
	if (K < 1 || pad < 0) return false;
	*Pf = (pad + 1) / 2;
	*s0 = 2 * *Pf - pad;
	*Kf = (K - 1 + *s0) / 2 + 1;
	return true;
}

extern "C" int convasr_fold2_geometry(int K, int pad, int* K_folded, int* pad_folded) {
	int s0;
	CONVASR_CHECK_ARG(K_folded && pad_folded && fold2_geometry(K, pad, K_folded, pad_folded, &s0), "fold2_geometry: bad arguments");
	return 0;
}

// wf[j][co][p Cin + ci] = w[co][ci][2 j + p - s0]; rows co >= Cout of the packed operand are zero
template <typename T>
__global__ __launch_bounds__(256) void fold2_pack_kernel(const float* __restrict__ w, int kmajor, T* __restrict__ wf, int Cout, int CoutPad, int Cin, int K, int Kf, int s0) {
	const int64_t n = (int64_t)Kf * CoutPad * 2 * Cin;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		const int c2 = (int)(i % (2 * Cin));
		const int64_t r = i / (2 * Cin);
		const int co = (int)(r % CoutPad), j = (int)(r / CoutPad);
		const int p = c2 >= Cin ? 1 : 0, ci = c2 - p * Cin, k = 2 * j + p - s0;
		float v = 0.f;
		if (co < Cout && k >= 0 && k < K) v = kmajor ? w[((int64_t)k * Cout + co) * Cin + ci] : w[((int64_t)co * Cin + ci) * K + k];
		Elem<T>::store(wf + i, v);
	}
}

extern "C" int convasr_fold2_pack_weight(const float* w, int w_layout, void* packed_fwd, int dtype, int Cout, int Cin, int K, int pad, void* stream) {
	int Kf, Pf, s0;
	CONVASR_CHECK_ARG(w && packed_fwd && Cout > 0 && Cin > 0 && fold2_geometry(K, pad, &Kf, &Pf, &s0) && (w_layout == CONVASR_W_REFERENCE || w_layout == CONVASR_W_KMAJOR), "fold2_pack_weight: bad arguments");
	const int co_pad = convasr_conv_cout_pad(Cout);
	int64_t blocks = ceil_div64((int64_t)Kf * co_pad * 2 * Cin, 256);
	if (blocks > 2048) blocks = 2048;
	if (dtype == CONVASR_BF16) hipLaunchKernelGGL((fold2_pack_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, w_layout == CONVASR_W_KMAJOR, (bf16_t*)packed_fwd, Cout, co_pad, Cin, K, Kf, s0);
	else if (dtype == CONVASR_F16) hipLaunchKernelGGL((fold2_pack_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, w_layout == CONVASR_W_KMAJOR, (f16_t*)packed_fwd, Cout, co_pad, Cin, K, Kf, s0);
	else if (dtype == CONVASR_F32) hipLaunchKernelGGL((fold2_pack_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, w_layout == CONVASR_W_KMAJOR, (float*)packed_fwd, Cout, co_pad, Cin, K, Kf, s0);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "fold2_pack_weight: dtype %d", dtype);
	CONVASR_CHECK_LAUNCH("fold2_pack_weight");
	return 0;
}

// dw[co][ci][k] (+)= dwf[j][co][p Cin + ci] with 2 j + p - s0 = k: the gradient of the folded conv, tap-major [K'][Cout][2 Cin], back
// in the parameter's own layout (the folded taps outside [0, K) are gradients of structural zeros and are dropped)
__global__ __launch_bounds__(256) void fold2_unfold_kernel(const float* __restrict__ dwf, float* __restrict__ dw, int kmajor, int Cout, int Cin, int K, int s0, int accumulate) {
	const int64_t n = (int64_t)K * Cout * Cin;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		int k, co, ci;
		if (kmajor) { ci = (int)(i % Cin); const int64_t r = i / Cin; co = (int)(r % Cout); k = (int)(r / Cout); }
		else { k = (int)(i % K); const int64_t r = i / K; ci = (int)(r % Cin); co = (int)(r / Cin); }
		const int j = (k + s0) >> 1, p = (k + s0) & 1;
		const float v = dwf[((int64_t)j * Cout + co) * 2 * Cin + p * Cin + ci];
		dw[i] = accumulate ? dw[i] + v : v;
	}
}

extern "C" int convasr_fold2_unfold_wgrad(const float* dw_folded, float* dw, int dw_layout, int Cout, int Cin, int K, int pad, int accumulate, void* stream) {
	int Kf, Pf, s0;
	CONVASR_CHECK_ARG(dw_folded && dw && Cout > 0 && Cin > 0 && fold2_geometry(K, pad, &Kf, &Pf, &s0) && (dw_layout == CONVASR_W_REFERENCE || dw_layout == CONVASR_W_KMAJOR), "fold2_unfold_wgrad: bad arguments");
	int64_t blocks = ceil_div64((int64_t)K * Cout * Cin, 256);
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(fold2_unfold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dw_folded, dw, dw_layout == CONVASR_W_KMAJOR, Cout, Cin, K, s0, accumulate);
	CONVASR_CHECK_LAUNCH("fold2_unfold_wgrad");
	return 0;
}
