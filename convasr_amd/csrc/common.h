// Shared device/host helpers for libconvasr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/convasr_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

extern thread_local char g_convasr_err[512];
int convasr_fail(int code, const char* fmt, ...);

#define CONVASR_CHECK_ARG(cond, ...) do { if (!(cond)) return convasr_fail(CONVASR_EINVAL, __VA_ARGS__); } while (0)
#define CONVASR_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return convasr_fail(CONVASR_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

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

// 8 consecutive elements as fp32 (16 B of bf16 or 32 B of f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
	float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
	v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
	uint4 a = *reinterpret_cast<const uint4*>(p);
	unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
	for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
	*reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
	*reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
	unsigned w[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
	*reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

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
// Block counter c = offset + idx / 8; c4 = 4 c (64 bit).  key = seed_lo ^ lowbias32(high word of c4 ^ seed_hi) (constant for 2^30
// blocks); word 0 = lowbias32(low word of c4 ^ key), words 1..3 = successive xorshift32 steps of it: 4 x 32 bits = 8 x 16 random bits;
// an element is dropped when its bits are < thr (= round(p * 65536)), kept ones are scaled by `scale` (= 65536 / (65536 - thr)).
// 4 integer multiplies (quarter-rate instructions) per 8 elements.  History: Philox4x32-7 (CONVASR_DROPOUT_PHILOX builds, the round-1
// generator) needs 56 and cost ~6 us per 256 x 128 tile in the dgrad epilogue; four hashed words (10 multiplies) made the BN forward
// pass VALU-bound AND gave seeds that differ in their low two bits PERMUTED copies of the same mask (the key only flipped low bits
// of the four counters of a block).  scratch/dropout_stats.py, 4 M blocks at p = 0.2: keep rate 0.79977..0.80022 per position,
// max |correlation| within a block 1.1e-3, between consecutive blocks 1.3e-3, between layers 1.1e-3, between seeds 1.5e-3
// (3 sigma = 1.5e-3), chi-square of the 256 keep patterns against the binomial law 272 (255 dof).
__device__ __forceinline__ unsigned xorshift32(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

__device__ __forceinline__ void dropout_mask8(uint64_t seed, uint64_t offset, unsigned thr, float scale, int64_t idx, float (&keep)[8]) {
	unsigned r[4];
#ifdef CONVASR_DROPOUT_PHILOX
	philox4x32<7>(seed, offset + (uint64_t)(idx >> 3), r);
#else
	const uint64_t c4 = (offset + (uint64_t)(idx >> 3)) << 2;
	const unsigned key = (unsigned)seed ^ lowbias32((unsigned)(c4 >> 32) ^ (unsigned)(seed >> 32));
	r[0] = lowbias32((unsigned)c4 ^ key);
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
