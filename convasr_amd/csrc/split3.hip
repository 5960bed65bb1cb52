// Split-operand ("x3") conv path: fp32-class accuracy on the 16-bit matrix pipe.
//
// A value v is carried as two 16-bit numbers, hi = rn16(v) and lo = rn16(v - hi) (v - hi is exact in fp32), and a product of two such
// values as hi*hi + hi*lo + lo*hi (lo*lo, 2^-16 of the product in bf16, 2^-22 in fp16, is dropped): three MFMAs per product, each exact
// in fp32 (a product of two 16-bit values has at most 22 significant bits), accumulated in fp32 like every other conv here.  With bf16
// planes the operands keep 16 significant bits -- the CTC loss of the full Wav2Letter step then sits ~1e-5 from the fp32 reference where
// plain bf16 is at 1.3e-3 and plain fp16 at 2e-4 (north_star: 1e-4) -- at three times the 16-bit MFMA work, i.e. several times the rate
// of the exact-fp32 MFMA path (157 TF peak).
//
// No new conv kernel: the three products are folded into the REDUCTION axis of the existing LDS-DMA kernels.
//   * activations (and output gradients) are stored as three planes per frame, memory [B][T][3][C]:
//       order 0 (an input x):  (hi, lo, hi)        order 1 (an output gradient dy):  (hi, hi, lo)
//   * forward / dgrad read that memory as a conv input of 3 C channels; the packed weights carry the matching planes along their
//     channel axis -- forward (w_hi, w_hi, w_lo): x_hi w_hi + x_lo w_hi + x_hi w_lo; dgrad (w_hi, w_lo, w_hi) against dy's (hi, hi, lo);
//   * the weight gradient reduces over (b, t): the same memory read as 3 T frames of C channels, with the conv's dilation and padding
//     tripled (frame 3 t + p is plane p of frame t), pairs x's plane p with dy's plane p: x_hi dy_hi + x_lo dy_hi + x_hi dy_lo.
// Reference: models.py:47-77 (nn.Conv1d in fp32); tolerance precedent train.py:491-495.
#include "common.h"

// x fp32 [rows][C] -> out [rows][3][C]; 8 channels per thread (32 B in, 3 x 16 B out)
template <typename H> __global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, H* __restrict__ out, int64_t rows, int C, int order) {
	const int c8 = C >> 3;
	const int64_t n = rows * c8;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		const int64_t r = i / c8;
		const int c = (int)(i - r * c8) << 3;
		float v[8];
		load8<float>(x + r * C + c, v);
		split3_store8<H>(out + r * 3 * C + c, C, order, v);
	}
}

extern "C" int convasr_split3(const float* x, void* out, int dtype, int64_t rows, int C, int order, void* stream) {
	CONVASR_CHECK_ARG(x && out && rows > 0 && C > 0 && (C & 7) == 0 && (order == 0 || order == 1) && convasr_is_half(dtype), "split3: bad arguments (C %% 8 == 0, a 16-bit plane type, order 0 / 1)");
	int64_t blocks = ceil_div64(rows * (C >> 3), 256);
	if (blocks > 16384) blocks = 16384;
	CONVASR_DISPATCH_HALF(dtype, H, hipLaunchKernelGGL((split3_kernel<H>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (H*)out, rows, C, order));
	CONVASR_CHECK_LAUNCH("split3");
	return 0;
}

// Packed operands of a split conv from the fp32 master, one launch for both:
//   fwd  [K][co_pad][3 Cin]:  row (k, co)          = (w_hi[co][.][k], w_hi[co][.][k], w_lo[co][.][k])
//   dgr  [K][ci_pad][3 Cout]: row (K - 1 - k, ci)  = (w_hi[.][ci][k], w_lo[.][ci][k], w_hi[.][ci][k])      (transposed, taps flipped)
//     or, dgrad_planes = 1, [K][ci_pad][Cout]: w_hi alone -- the ordinary 16-bit dgrad operand, for a backward that runs one product per
//     gradient behind a split forward ('bf16x3f' / 'f16x3f')
// One block = one 64 (co) x 64 (ci) tile of one tap through LDS (the dgrad rows are the tile's columns).  Rows >= Cout / >= Cin of the
// padded operands are never written (zero from their allocation).
template <typename H> __global__ __launch_bounds__(256) void pack_split3_kernel(const float* __restrict__ w, int64_t s_co, int64_t s_ci, int64_t s_k, H* __restrict__ fwd, H* __restrict__ dgr, int Cout, int Cin, int K, int co_pad, int ci_pad, int dgr_planes) {
	__shared__ float tile[64][65];
	const int k = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	// the fastest source axis goes to the lanes: ci for tap-major / one-tap weights, otherwise the (co, ci) order of the reference layout
#pragma unroll 4
	for (int i = 0; i < 16; ++i) {
		const int co = co0 + ty + 4 * i, ci = ci0 + tx;
		tile[ty + 4 * i][tx] = (co < Cout && ci < Cin) ? w[co * s_co + ci * s_ci + k * s_k] : 0.f;
	}
	__syncthreads();
	if (fwd) {
		H* const dst = fwd + (int64_t)k * co_pad * 3 * Cin;
#pragma unroll 4
		for (int i = 0; i < 16; ++i) {
			const int co = co0 + ty + 4 * i, ci = ci0 + tx;
			if (co < Cout && ci < Cin) {
				const float v = tile[ty + 4 * i][tx];
				H h, l;
				Elem<H>::store(&h, v);
				Elem<H>::store(&l, v - Elem<H>::load(&h));
				H* const row = dst + (int64_t)co * 3 * Cin + ci;
				row[0] = h; row[Cin] = h; row[2 * Cin] = l;
			}
		}
	}
	if (dgr) {
		H* const dst = dgr + (int64_t)(K - 1 - k) * ci_pad * dgr_planes * Cout;
#pragma unroll 4
		for (int i = 0; i < 16; ++i) {
			const int ci = ci0 + ty + 4 * i, co = co0 + tx;
			if (co < Cout && ci < Cin) {
				const float v = tile[tx][ty + 4 * i];
				H h, l;
				Elem<H>::store(&h, v);
				Elem<H>::store(&l, v - Elem<H>::load(&h));
				H* const row = dst + (int64_t)ci * dgr_planes * Cout + co;
				row[0] = h;
				if (dgr_planes == 3) { row[Cout] = l; row[2 * Cout] = h; }
			}
		}
	}
}

extern "C" int convasr_pack_conv_weight_split3(const float* w, int w_layout, void* packed_fwd, void* packed_dgrad, int dgrad_planes, int dtype, int Cout, int Cin, int K, void* stream) {
	CONVASR_CHECK_ARG(w && (packed_fwd || packed_dgrad) && (dgrad_planes == 3 || dgrad_planes == 1) && Cout > 0 && Cin > 0 && K > 0 && K <= 64 && convasr_is_half(dtype) && (w_layout == CONVASR_W_REFERENCE || w_layout == CONVASR_W_KMAJOR), "pack_conv_weight_split3: bad arguments");
	const int co_pad = convasr_conv_cout_pad(Cout), ci_pad = convasr_conv_cout_pad(Cin);
	const int64_t s_co = w_layout == CONVASR_W_KMAJOR ? Cin : (int64_t)Cin * K, s_ci = w_layout == CONVASR_W_KMAJOR ? 1 : K, s_k = w_layout == CONVASR_W_KMAJOR ? (int64_t)Cout * Cin : 1;
	const dim3 grid((Cin + 63) / 64, (Cout + 63) / 64, K);
	CONVASR_DISPATCH_HALF(dtype, H, hipLaunchKernelGGL((pack_split3_kernel<H>), grid, dim3(256), 0, (hipStream_t)stream, w, s_co, s_ci, s_k, (H*)packed_fwd, (H*)packed_dgrad, Cout, Cin, K, co_pad, ci_pad, dgrad_planes));
	CONVASR_CHECK_LAUNCH("pack_conv_weight_split3");
	return 0;
}
