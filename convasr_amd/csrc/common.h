// Shared device/host helpers for libconvasr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/convasr_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef _Float16 f16_t;         // IEEE half (CONVASR_F16): a distinct C++ type, so the kernels' storage-type templates tell the two 16-bit formats apart

extern thread_local char g_convasr_err[512];
int convasr_fail(int code, const char* fmt, ...);

#define CONVASR_CHECK_ARG(cond, ...) do { if (!(cond)) return convasr_fail(CONVASR_EINVAL, __VA_ARGS__); } while (0)
#define CONVASR_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return convasr_fail(CONVASR_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Kernels that ask for more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER DEVICE: `mask`
// (one static per call site and kernel) keeps a bit per device ordinal, so a process that drives a second GPU sets it there too.
static inline void convasr_allow_160k_lds(const void* kern, unsigned long long& mask) {
	int dev = 0;
	(void)hipGetDevice(&dev);
	const unsigned long long bit = 1ull << (dev & 63);
	if (!(mask & bit)) { (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); mask |= bit; }
}

// Compute units of the CURRENT device, cached per device ordinal (one process may drive several GPUs; a process-wide cache would hand the
// second device the first one's count).
static inline int convasr_cu_count() {
	static int cached[64] = {};
	int dev = 0;
	(void)hipGetDevice(&dev);
	int& n = cached[dev & 63];
	if (!n && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)) n = 256;
	return n;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even; NaN stays NaN (a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
	__bf16 h = (__bf16)f;
	return __builtin_bit_cast(unsigned short, h);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
	static constexpr int dtype = CONVASR_F32;
	__device__ static __forceinline__ float load(const float* p) { return *p; }
	__device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
	static constexpr int dtype = CONVASR_BF16;
	__device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
	__device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct Elem<f16_t> {
	static constexpr int dtype = CONVASR_F16;
	__device__ static __forceinline__ float load(const f16_t* p) { return (float)*p; }
	__device__ static __forceinline__ void store(f16_t* p, float v) { *p = (f16_t)v; }  // round-to-nearest-even, +-inf beyond 65504 (what the loss scaler's overflow check keys on)
};

// 16 bytes of a 16-bit storage type <-> 8 floats (element 2i in the low half of word i)
template <typename T> __device__ __forceinline__ void unpack16(const uint4& raw, float (&v)[8]);
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& raw, float (&v)[8]) {
	const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
	for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& raw, float (&v)[8]) {
	const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
	for (int i = 0; i < 4; ++i) { const f16x2 h = __builtin_bit_cast(f16x2, w[i]); v[2 * i] = (float)h[0]; v[2 * i + 1] = (float)h[1]; }
}
template <typename T> __device__ __forceinline__ unsigned pack16(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack16<bf16_t>(float lo, float hi) { return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16); }
template <> __device__ __forceinline__ unsigned pack16<f16_t>(float lo, float hi) { f16x2 h; h[0] = (f16_t)lo; h[1] = (f16_t)hi; return __builtin_bit_cast(unsigned, h); }

// 8 consecutive elements as fp32 (16 B of bf16 / fp16 or 32 B of f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
	float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
	v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) { unpack16<bf16_t>(*reinterpret_cast<const uint4*>(p), v); }
template <> __device__ __forceinline__ void load8<f16_t>(const f16_t* p, float (&v)[8]) { unpack16<f16_t>(*reinterpret_cast<const uint4*>(p), v); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
	*reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
	*reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
	*reinterpret_cast<uint4*>(p) = make_uint4(pack16<bf16_t>(v[0], v[1]), pack16<bf16_t>(v[2], v[3]), pack16<bf16_t>(v[4], v[5]), pack16<bf16_t>(v[6], v[7]));
}
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float (&v)[8]) {
	*reinterpret_cast<uint4*>(p) = make_uint4(pack16<f16_t>(v[0], v[1]), pack16<f16_t>(v[2], v[3]), pack16<f16_t>(v[4], v[5]), pack16<f16_t>(v[6], v[7]));
}

// Split-operand planes (csrc/split3.hip): 8 consecutive fp32 values -> their hi / lo 16-bit planes at out[0 .. 8), out[C .. C + 8), out[2 C .. 2 C + 8)
// (out = &planes[row][0][c]); order 0 = (hi, lo, hi) for a conv input, 1 = (hi, hi, lo) for an output gradient.  lo = rn16(v - hi), v - hi exact.
template <typename H> __device__ __forceinline__ void split3_store8(H* out, int C, int order, const float (&v)[8]) {
	float hi[8], lo[8];
	const uint4 ph = make_uint4(pack16<H>(v[0], v[1]), pack16<H>(v[2], v[3]), pack16<H>(v[4], v[5]), pack16<H>(v[6], v[7]));
	unpack16<H>(ph, hi);
#pragma unroll
	for (int k = 0; k < 8; ++k) lo[k] = v[k] - hi[k];
	const uint4 pl = make_uint4(pack16<H>(lo[0], lo[1]), pack16<H>(lo[2], lo[3]), pack16<H>(lo[4], lo[5]), pack16<H>(lo[6], lo[7]));
	*reinterpret_cast<uint4*>(out) = ph;
	*reinterpret_cast<uint4*>(out + C) = order == 0 ? pl : ph;
	*reinterpret_cast<uint4*>(out + 2 * C) = order == 0 ? ph : pl;
}

// Run `fn` with a value of the 16-bit storage type a CONVASR_* dtype code names (callers have checked that it is one of the two).
#define CONVASR_DISPATCH_HALF(dtype, T, ...) do { if ((dtype) == CONVASR_F16) { typedef f16_t T; __VA_ARGS__; } else { typedef bf16_t T; __VA_ARGS__; } } while (0)
static inline bool convasr_is_half(int dtype) { return dtype == CONVASR_BF16 || dtype == CONVASR_F16; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
	return v;
}

// Dynamic loss scaler (fp16 training; the role of apex.amp's LossScaler behind train.py:770-772): state = CONVASR_LOSS_SCALER_FLOATS
// floats on the device, read by the loss head (backward seed x scale), the gradient-norm kernel and the fused optimizer steps
// (gradients x 1 / scale; a non-finite gradient norm = overflow: the step is skipped), and advanced by the optimizer step into a
// SECOND buffer (in != out; the host swaps the two after every step, like NovoGrad's EMAs) -- apex's update_scale():
// overflow: scale = max(min_scale, scale / factor), unskipped = 0; else ++unskipped, and at unskipped == window: scale =
// min(max_scale, scale * factor), unskipped = 0.  window == 0: static scale, no overflow check.
enum { LS_SCALE = 0, LS_UNSKIPPED = 1, LS_OVERFLOW = 2, LS_WINDOW = 3, LS_MIN = 4, LS_MAX = 5, LS_FACTOR = 6, LS_SKIPPED_STEPS = 7 };
struct LossScale { float scale, inv; bool overflow; };
// norm_sq: the squared norm of the (scaled) gradient; NULL scaler: scale 1, never an overflow
__device__ __forceinline__ LossScale loss_scale_read(const float* __restrict__ scaler, double norm_sq) {
	LossScale r; r.scale = 1.f; r.inv = 1.f; r.overflow = false;
	if (scaler) {
		r.scale = scaler[LS_SCALE];
		r.inv = 1.f / r.scale;
		r.overflow = scaler[LS_WINDOW] > 0.f && !(fabs(norm_sq) < (double)INFINITY);
	}
	return r;
}
// one thread: out = the state after this step (gated: the step was skipped for a non-finite LOSS, backward's result is not looked at)
__device__ __forceinline__ void loss_scale_advance(const float* __restrict__ in, float* __restrict__ out, bool overflow, bool gated) {
#pragma unroll
	for (int i = 0; i < CONVASR_LOSS_SCALER_FLOATS; ++i) out[i] = in[i];
	if (gated) return;
	float scale = in[LS_SCALE], unskipped = in[LS_UNSKIPPED];
	const float window = in[LS_WINDOW], factor = in[LS_FACTOR];
	out[LS_OVERFLOW] = overflow ? 1.f : 0.f;
	if (window <= 0.f) return;
	if (overflow) { scale = fmaxf(in[LS_MIN], scale / factor); unskipped = 0.f; out[LS_SKIPPED_STEPS] = in[LS_SKIPPED_STEPS] + 1.f; }
	else unskipped += 1.f;
	if (unskipped >= window) { scale = fminf(in[LS_MAX], scale * factor); unskipped = 0.f; }
	out[LS_SCALE] = scale;
	out[LS_UNSKIPPED] = unskipped;
}

// valid frames of a T-long axis: ceil(frac * T) computed in fp32 like (lengths_fraction * T).ceil().long() (models.py:614)
__device__ __forceinline__ int valid_len(const float* xlen, int b, int T) {
	if (xlen == nullptr) return T;
	float v = ceilf(xlen[b] * (float)T);
	return (int)v;
}

// Activations in a branch-free form.  `switch (act)` inside an unrolled per-element loop compiles to a chain of SCALAR branches per
// element (the activation kind is wave-uniform, so hipcc branches instead of selecting): the conv epilogue spent ~9,000 cycles per
// tile staging 64 accumulators per lane that way (in-kernel stamps).  The kind is folded ONCE into four constants:
//   value:    leaky ? (v > 0 ? v : v * slope) : min(max(v, lo), hi)        none: (-inf, +inf), relu: (0, +inf), hardtanh: (lo, hi)
//   gradient: (pre > lo && pre < hi) ? 1 : gelse                            relu / hardtanh: 0, leaky: slope, none: 1
// (strict inequalities, matching ATen's backward formulas).
struct ActConst { float lo, hi, slope, gelse; int leaky; };
__host__ __device__ __forceinline__ ActConst act_const(int act, float lo, float hi) {
	ActConst c;
	c.lo = -INFINITY; c.hi = INFINITY; c.slope = 1.f; c.gelse = 1.f; c.leaky = 0;
	if (act == CONVASR_ACT_RELU) { c.lo = 0.f; c.gelse = 0.f; }
	else if (act == CONVASR_ACT_HARDTANH) { c.lo = lo; c.hi = hi; c.gelse = 0.f; }
	else if (act == CONVASR_ACT_LEAKY_RELU) { c.lo = 0.f; c.slope = lo; c.gelse = lo; c.leaky = 1; }
	return c;
}
__device__ __forceinline__ float apply_act(float v, const ActConst& c) {
	const float clamped = fminf(fmaxf(v, c.lo), c.hi), lk = v > 0.f ? v : v * c.slope;
	return c.leaky ? lk : clamped;
}
// the same for callers that know at compile time that the kind is not leaky-relu: one v_med3_f32 instead of six instructions
__device__ __forceinline__ float apply_clamp(float v, const ActConst& c) { return __builtin_amdgcn_fmed3f(v, c.lo, c.hi); }
// derivative w.r.t. the pre-activation value
__device__ __forceinline__ float act_grad(float pre, const ActConst& c) { return (pre > c.lo && pre < c.hi) ? 1.f : c.gelse; }

// Philox4x32-10 (Salmon et al. 2011): counter = element index / 4, key = seed; returns 4 uniforms in [0,1)
template <int ROUNDS> __device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t ctr, unsigned (&r)[4]) {
	unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0, c3 = 0;
	unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i) {
		unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
		unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
		unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// 32-bit integer hash with very low bias (two multiply / xor-shift rounds; constants from C. Wellons' hash-prospector search,
// "lowbias32"): a bijection of the 32-bit counters, avalanche bias ~0.17 bits.
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
	x ^= x >> 16; x *= 0x7feb352du;
	x ^= x >> 15; x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

// Dropout mask of the 8 elements starting at element index idx (a multiple of 8): a counter-based generator, so the forward pass,
// the fused backward epilogue and the backward apply pass regenerate the same mask from (seed, offset, idx) instead of storing it.
// Block counter c = offset + idx / 8; c4 = 4 c (64 bit).  seed = the caller's seed whitened on the host (convasr_mix_seed);
// key = seed_lo ^ lowbias32(high word of c4 ^ seed_hi) (constant for 2^30 blocks); word 0 = lowbias32_keyed(low word of c4, key),
// words 1..3 = successive xorshift32 steps of it: 4 x 32 bits = 8 x 16 random bits;
// an element is dropped when its bits are < thr (= round(p * 65536)), kept ones are scaled by `scale` (= 65536 / (65536 - thr)).
// 4 integer multiplies (quarter-rate instructions) per 8 elements.  History: Philox4x32-7 (CONVASR_DROPOUT_PHILOX builds, the round-1
// generator) needs 56 and cost ~6 us per 256 x 128 tile in the dgrad epilogue; four hashed words (10 multiplies) made the BN forward
// pass VALU-bound AND gave seeds that differ in their low two bits PERMUTED copies of the same mask (the key only flipped low bits
// of the four counters of a block).  scratch/dropout_stats.py, 4 M blocks at p = 0.2: keep rate 0.79977..0.80022 per position,
// max |correlation| within a block 1.1e-3, between consecutive blocks 1.3e-3, between layers 1.1e-3, between seeds 1.5e-3
// (3 sigma = 1.5e-3), chi-square of the 256 keep patterns against the binomial law 272 (255 dof).
__device__ __forceinline__ unsigned xorshift32(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
// lowbias32 with the key injected BETWEEN its two multiply rounds: as a function of the counter it is a different bijection for every
// key.  (hash(counter ^ key), the round-2 form, only XOR-permutes the counters: seeds that agreed in their low two bits gave each other's
// mask streams with the 8-element blocks permuted -- ranks r and r + 4 of a job seeded 1 + rank.)
__device__ __forceinline__ unsigned lowbias32_keyed(unsigned x, unsigned key) {
	x ^= x >> 16; x *= 0x7feb352du;
	x ^= key;
	x ^= x >> 15; x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}
// host side of the dropout generator: the caller's seed is whitened once per launch (splitmix64), so that seeds 1, 2, 3 ... give
// unrelated keys; the kernels receive the whitened value
__host__ __device__ static inline uint64_t convasr_mix_seed(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

__device__ __forceinline__ void dropout_mask8(uint64_t seed, uint64_t offset, unsigned thr, float scale, int64_t idx, float (&keep)[8]) {
	unsigned r[4];
#ifdef CONVASR_DROPOUT_PHILOX
	philox4x32<7>(seed, offset + (uint64_t)(idx >> 3), r);
#else
	const uint64_t c4 = (offset + (uint64_t)(idx >> 3)) << 2;
	const unsigned key = (unsigned)seed ^ lowbias32((unsigned)(c4 >> 32) ^ (unsigned)(seed >> 32));
	r[0] = lowbias32_keyed((unsigned)c4, key);
#pragma unroll
	for (int i = 1; i < 4; ++i) r[i] = xorshift32(r[i - 1]);
#endif
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		keep[2 * i] = (r[i] & 0xffffu) >= thr ? scale : 0.f;
		keep[2 * i + 1] = (r[i] >> 16) >= thr ? scale : 0.f;
	}
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, float (&u)[4]) {
	unsigned r[4];
	philox4x32<10>(seed, ctr, r);
#pragma unroll
	for (int i = 0; i < 4; ++i) u[i] = (r[i] >> 8) * (1.0f / 16777216.0f);
}
