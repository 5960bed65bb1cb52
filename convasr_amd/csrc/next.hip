// SURVEY section 8(f) "next" rows that share the hot path's data:
//   f2  NovoGrad (optimizers.py:66-90) with clip_grad_norm_ (train.py:777) folded in, over the flat parameter arena
//   f3  ctc.alignment (ctc.py:7-75): forced alignment with 2-bit back-pointers, one wave per utterance
//   f4  the padding half of AudioTextDataset.collate_fn (datasets.py:305-332) on the GPU: ragged samples -> zero-padded batch
#include "common.h"

// ------------------------------------------------------------------------------------------------ NovoGrad
// The arena is a concatenation of segments (one per parameter tensor, offsets[n_seg + 1]); NovoGrad's second moment is one
// scalar per segment: the EMA of the squared gradient norm of that tensor.  Per-segment sums of squares first (item table, no
// atomics), then the fused update: there a workgroup owns NG_CHUNK consecutive arena elements; almost every workgroup lies
// inside one segment (one table lookup); the few that straddle a boundary look the segment up per element.
#define NG_CHUNK 8192
#define NG_TABLE 2048  // segment offsets are searched in LDS (a dependent chain of global loads per lookup made the launches latency-bound)

__device__ __forceinline__ const int64_t* ng_stage_offsets(const int64_t* __restrict__ offsets, int n_seg, int64_t* tab) {
	if (n_seg + 1 > NG_TABLE) return offsets;
	for (int s = threadIdx.x; s <= n_seg; s += blockDim.x) tab[s] = offsets[s];
	__syncthreads();
	return tab;
}

__device__ __forceinline__ int ng_segment(const int64_t* offsets, int n_seg, int64_t i) {
	int lo = 0, hi = n_seg - 1;  // largest s with offsets[s] <= i
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// Per-segment sums of squares without atomics: the caller cuts every segment into items of at most NG_ITEM elements once
// (items[i] = {segment, begin, end}, seg_first[s] = first item of segment s); one workgroup reduces one item into item_part[i], then
// one thread per segment adds its items in order -- the EMAs, hence the parameters, are bit-identical from run to run.
#define NG_ITEM 65536
__global__ __launch_bounds__(256) void ng_item_sumsq_kernel(const float* __restrict__ g, const int64_t* __restrict__ items, double* __restrict__ item_part) {
	__shared__ double red[4];
	const int64_t i0 = items[3 * (int64_t)blockIdx.x + 1], i1 = items[3 * (int64_t)blockIdx.x + 2];
	float a = 0.f;
	const int64_t head = min(i1, (i0 + 3) & ~(int64_t)3);  // arena offsets are multiples of 64, so i0 is 16-byte aligned in practice
	for (int64_t i = i0 + threadIdx.x; i < head; i += 256) a += g[i] * g[i];
	const int64_t n4 = (i1 - head) >> 2;
	const float4* g4 = reinterpret_cast<const float4*>(g + head);
	for (int64_t i = threadIdx.x; i < n4; i += 256) { const float4 v = g4[i]; a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
	for (int64_t i = head + n4 * 4 + threadIdx.x; i < i1; i += 256) a += g[i] * g[i];
	double acc = (double)a;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) item_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void ng_segment_sum_kernel(const double* __restrict__ item_part, const int64_t* __restrict__ seg_first, int n_seg, double* __restrict__ g2) {
	const int s = blockIdx.x * 256 + threadIdx.x;
	if (s >= n_seg) return;
	double a = 0;
	for (int64_t i = seg_first[s]; i < seg_first[s + 1]; ++i) a += item_part[i];
	g2[s] = a;
}

struct NgParams {
	float* p; const float* g; float* mom; const float* ema_in; float* ema_out; const double* g2; const int64_t* offsets;
	int n_seg; int64_t n;
	float max_norm, lr, b1, b2, eps, wd;
	int dampening, first;
	const float* loss_gate; float* norm_out; float grad_scale;
	void* p16;
	const float* lr_dev;  // optional: the learning rate read from device memory instead of `lr`
	const float* scaler_in; float* scaler_out;  // optional dynamic loss scaler (common.h): gradients are multiplied by 1 / scale, a non-finite gradient norm skips the step
};

template <typename H> __global__ __launch_bounds__(256) void ng_step_kernel(NgParams q) {
	__shared__ double red[4];
	__shared__ float s_clip;
	__shared__ int64_t tab[NG_TABLE];
	// first < 0: the EMA buffers carry one extra element, the count of steps APPLIED so far (a gated step does not count: the
	// reference creates no optimizer state on an iteration it skips, optimizers.py:76-80 / train.py:769-772)
	const int n_carry = q.n_seg + (q.first < 0 ? 1 : 0);
	if (q.lr_dev) q.lr = *q.lr_dev;
	const bool gated = q.loss_gate && !(fabsf(*q.loss_gate) < INFINITY);
	q.offsets = ng_stage_offsets(q.offsets, q.n_seg, tab);
	double tot = 0;
	for (int s = threadIdx.x; s < q.n_seg; s += 256) tot += q.g2[s];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
	__syncthreads();
	const double total_sq = red[0] + red[1] + red[2] + red[3];
	const LossScale ls = loss_scale_read(q.scaler_in, q.scaler_in ? total_sq : 0.0);
	if (q.scaler_in && blockIdx.x == 0 && threadIdx.x == 0) loss_scale_advance(q.scaler_in, q.scaler_out, ls.overflow, gated);
	if (blockIdx.x == 0 && threadIdx.x == 0 && q.norm_out) *q.norm_out = (float)sqrt(total_sq) * q.grad_scale * ls.inv;  // (also on a skipped step: inf / NaN after an overflow, like clip_grad_norm_ would return)
	if (gated || ls.overflow) {  // skipped step: nothing changes; the caller still swaps its two EMA buffers, so carry the EMAs over
		if (blockIdx.x == 0)
			for (int s = threadIdx.x; s < n_carry; s += 256) q.ema_out[s] = q.ema_in[s];
		return;
	}
	q.grad_scale *= ls.inv;
	if (q.first < 0) {
		const float applied = q.ema_in[q.n_seg];
		if (blockIdx.x == 0 && threadIdx.x == 0) q.ema_out[q.n_seg] = fminf(applied + 1.f, 16777216.f);
		q.first = applied == 0.f;
	}
	if (threadIdx.x == 0) {
		const float total = (float)sqrt(total_sq) * q.grad_scale;  // norm of the scaled (e.g. rank-averaged, loss-unscaled) gradient
		float c = 1.f;
		if (q.max_norm > 0.f) { c = q.max_norm / (total + 1e-6f); c = c < 1.f ? c : 1.f; }
		s_clip = c * q.grad_scale;
	}
	__syncthreads();
	const float clip = s_clip;
	const int64_t i0 = (int64_t)blockIdx.x * NG_CHUNK, i1 = min(q.n, i0 + NG_CHUNK);
	const int s0 = ng_segment(q.offsets, q.n_seg, i0), s1 = ng_segment(q.offsets, q.n_seg, i1 - 1);
	auto seg_ema = [&](int s) {
		const float g2c = (float)(q.g2[s] * (double)clip * (double)clip);  // sum of squares of the CLIPPED gradient (clip_grad_norm_ scales .grad in place)
		return q.first ? g2c : q.ema_in[s] * q.b2 + g2c * (1.f - q.b2);
	};
	// the workgroup that holds a segment's first element publishes its new EMA (ema_in / ema_out are distinct buffers)
	for (int s = s0 + threadIdx.x; s <= s1; s += 256)
		if (q.offsets[s] >= i0) q.ema_out[s] = seg_ema(s);
	auto update = [&](int64_t i, float inv_denom) {
		const float pv = q.p[i];
		float d = q.g[i] * clip * inv_denom;
		if (q.wd > 0.f) d += q.wd * pv;
		if (q.dampening) d *= 1.f - q.b1;
		const float m = q.first ? d : q.mom[i] * q.b1 + d;
		q.mom[i] = m;
		const float pn = pv - q.lr * m;
		q.p[i] = pn;
		if (q.p16) Elem<H>::store(reinterpret_cast<H*>(q.p16) + i, pn);
	};
	if (s0 == s1) {
		const float inv = 1.f / sqrtf(seg_ema(s0) + q.eps);
		for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) update(i, inv);
	} else {
		for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) update(i, 1.f / sqrtf(seg_ema(ng_segment(q.offsets, q.n_seg, i)) + q.eps));
	}
}

extern "C" int convasr_novograd_step(float* p, const float* g, float* mom, const float* ema_in, float* ema_out, double* g2, const int64_t* offsets, int n_seg,
                                     int64_t n, const int64_t* items, int n_items, const int64_t* seg_first, double* item_part, float max_norm, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int dampening, int first, const float* loss_gate, float* total_norm, float grad_scale,
                                     void* p16, int p16_dtype, const float* scaler_in, float* scaler_out, const float* lr_dev, void* stream) {
	CONVASR_CHECK_ARG(!p16 || convasr_is_half(p16_dtype), "novograd_step: the mirror's dtype must be CONVASR_BF16 or CONVASR_F16");
	CONVASR_CHECK_ARG((scaler_in == nullptr) == (scaler_out == nullptr) && (!scaler_in || scaler_in != scaler_out), "novograd_step: the loss scaler needs distinct in / out states");
	CONVASR_CHECK_ARG(p && g && mom && ema_in && ema_out && ema_in != ema_out && g2 && offsets && items && seg_first && item_part && n_seg > 0 && n_items >= n_seg && n > 0, "novograd_step: bad arguments");
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(ng_item_sumsq_kernel, dim3(n_items), dim3(256), 0, s, g, items, item_part);
	hipLaunchKernelGGL(ng_segment_sum_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, s, (const double*)item_part, seg_first, n_seg, g2);
	NgParams q;
	q.p = p; q.g = g; q.mom = mom; q.ema_in = ema_in; q.ema_out = ema_out; q.g2 = g2; q.offsets = offsets; q.n_seg = n_seg; q.n = n;
	q.max_norm = max_norm; q.lr = lr; q.b1 = beta1; q.b2 = beta2; q.eps = eps; q.wd = weight_decay; q.dampening = dampening; q.first = first;
	q.loss_gate = loss_gate; q.norm_out = total_norm; q.grad_scale = grad_scale; q.p16 = p16; q.scaler_in = scaler_in; q.scaler_out = scaler_out; q.lr_dev = lr_dev;
	CONVASR_DISPATCH_HALF(p16_dtype, H, hipLaunchKernelGGL((ng_step_kernel<H>), dim3((unsigned)ceil_div64(n, NG_CHUNK)), dim3(256), 0, s, q));
	CONVASR_CHECK_LAUNCH("novograd_step");
	return 0;
}

extern "C" int64_t convasr_novograd_item_elems(void) { return NG_ITEM; }

// ------------------------------------------------------------------------------------------------ forced alignment
// One wave per utterance; lane l owns states l*NS .. l*NS+NS-1 of the extended target.  The forward variable is the reference's
// SUM recursion in natural log (expf / logf, the precision class of torch.logsumexp) with finfo.min as "log zero"; the
// back-pointer of a state is the first maximum of (stay, s-1, s-2 if allowed), 2 bits per state, one packed dword per lane and
// frame in a global workspace [B][T][64].  Lane 0 then walks the path back from frame input_length-1 (the end state is chosen
// from the column at T-1, as ctc.py:53-58 does) and records for every label the last frame spent in its state.
#define AL_ZERO (-3.4028234663852886e38f)

template <int NS>
__global__ __launch_bounds__(64) void ctc_alignment_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len,
                                                           const int64_t* __restrict__ tgt_len, int64_t* __restrict__ out, unsigned* __restrict__ bp, int T, int C,
                                                           int S_max, int blank) {
	__shared__ float fin[64 * NS];
	const int b = blockIdx.x, lane = threadIdx.x;
	const int S = (int)tgt_len[b], Tb = (int)in_len[b], L = 2 * S + 1;
	const int64_t* tg = targets + (int64_t)b * S_max;
	int64_t* ob = out + (int64_t)b * S_max;
	for (int j = lane; j < S_max; j += 64) ob[j] = 0;
	if (S <= 0 || Tb <= 0 || Tb > T) return;
	const float* lpb = lp + (int64_t)b * T * C;
	unsigned* bpb = bp + (int64_t)b * T * 64;

	int cls[NS];
	bool allow2[NS], valid[NS];
	float a[NS];
#pragma unroll
	for (int i = 0; i < NS; ++i) {
		const int s = lane * NS + i;
		valid[i] = s < L;
		const bool lab = (s & 1) && valid[i];
		cls[i] = lab ? (int)tg[s >> 1] : blank;
		allow2[i] = lab && s >= 3 && tg[s >> 1] != tg[(s >> 1) - 1];  // blanks never take the s-2 move (equal to the blank two states back)
		a[i] = (valid[i] && s < 2) ? lpb[cls[i]] : AL_ZERO;
	}
	for (int t = 1; t < T; ++t) {
		const float* row = lpb + (int64_t)t * C;
		float p1 = __shfl_up(a[NS - 1], 1, 64), p2 = NS >= 2 ? __shfl_up(a[NS >= 2 ? NS - 2 : 0], 1, 64) : __shfl_up(a[0], 2, 64);
		if (lane == 0) { p1 = AL_ZERO; p2 = AL_ZERO; }
		if (NS == 1 && lane == 1) p2 = AL_ZERO;
		float n[NS];
		unsigned word = 0;
#pragma unroll
		for (int i = NS - 1; i >= 0; --i) {
			const float stay = a[i];
			const float one = i >= 1 ? a[i - 1] : p1;
			const float two = allow2[i] ? (i >= 2 ? a[i - 2] : (i == 1 ? p1 : p2)) : AL_ZERO;
			unsigned k = 0;
			float best = stay;
			if (one > best) { k = 1; best = one; }
			if (two > best) { k = 2; best = two; }
			word |= k << (2 * i);
			n[i] = valid[i] ? row[cls[i]] + (best + logf(expf(stay - best) + expf(one - best) + expf(two - best))) : AL_ZERO;
		}
		bpb[(int64_t)t * 64 + lane] = word;
#pragma unroll
		for (int i = 0; i < NS; ++i) a[i] = n[i];
	}
#pragma unroll
	for (int i = 0; i < NS; ++i) fin[lane * NS + i] = a[i];
	__threadfence_block();
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
	if (lane == 0) {
		int s = 2 * S - 1 + (fin[2 * S] > fin[2 * S - 1] ? 1 : 0);
		int seen = -1;
		for (int t = Tb - 1; t >= 0; --t) {
			if (s != seen) { if (s & 1) ob[s >> 1] = t; seen = s; }
			if (t > 0) s -= (int)((bpb[(int64_t)t * 64 + s / NS] >> (2 * (s % NS))) & 3u);
		}
	}
}

static int al_ns(int S_max) {
	const int need = (2 * S_max + 1 + 63) / 64;
	const int opts[] = {1, 2, 3, 4, 6, 8, 12, 16};
	for (int o : opts) if (o >= need) return o;
	return -1;
}

// Targets longer than 511 labels (a whole recording aligned to its transcript, transcribe.py:176 without segmentation): one workgroup of
// 16 waves per utterance, the forward column double-buffered in LDS (up to AL_WG_STATES states), thread j owns states j, j + 1024, ...
// (neighbouring states sit in neighbouring lanes: conflict-free LDS reads), one barrier per frame.  Same arithmetic per state as the
// one-wave kernel above -- the two give the same path bit for bit on targets both can take (tests).  Back-pointers: 2 bits per state, the
// 16 states of a thread in one dword, workspace [B][T][1024].
#define AL_WG_THREADS 1024
#define AL_WG_PER 16
#define AL_WG_STATES (AL_WG_THREADS * AL_WG_PER)
__global__ __launch_bounds__(AL_WG_THREADS) void ctc_alignment_wg_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len,
                                                                         const int64_t* __restrict__ tgt_len, int64_t* __restrict__ out, unsigned* __restrict__ bp, int T, int C,
                                                                         int S_max, int blank) {
	extern __shared__ float al_col[];  // [2][per * 1024 + 2]: two "log zero" states in front of state 0
	const int b = blockIdx.x, tid = threadIdx.x;
	const int S = (int)tgt_len[b], Tb = (int)in_len[b], L = 2 * S + 1;
	const int64_t* tg = targets + (int64_t)b * S_max;
	int64_t* ob = out + (int64_t)b * S_max;
	for (int j = tid; j < S_max; j += AL_WG_THREADS) ob[j] = 0;
	if (S <= 0 || Tb <= 0 || Tb > T) return;
	const int per = (L + AL_WG_THREADS - 1) / AL_WG_THREADS, pitch = per * AL_WG_THREADS + 2;
	const float* lpb = lp + (int64_t)b * T * C;
	unsigned* bpb = bp + (int64_t)b * T * AL_WG_THREADS;
	float* cur = al_col + 2;
	float* nxt = al_col + pitch + 2;
	if (tid < 2) { cur[tid - 2] = AL_ZERO; nxt[tid - 2] = AL_ZERO; }

	int cls[AL_WG_PER];
	bool allow2[AL_WG_PER], valid[AL_WG_PER];
#pragma unroll
	for (int i = 0; i < AL_WG_PER; ++i) {
		const int s = i * AL_WG_THREADS + tid;
		valid[i] = i < per && s < L;
		const bool lab = (s & 1) && valid[i];
		cls[i] = lab ? (int)tg[s >> 1] : blank;
		allow2[i] = lab && s >= 3 && tg[s >> 1] != tg[(s >> 1) - 1];
		if (i < per) cur[s] = (valid[i] && s < 2) ? lpb[cls[i]] : AL_ZERO;
	}
	__syncthreads();
	for (int t = 1; t < T; ++t) {
		const float* row = lpb + (int64_t)t * C;
		unsigned word = 0;
#pragma unroll
		for (int i = 0; i < AL_WG_PER; ++i) {
			if (i < per) {  // uniform over the workgroup
				const int s = i * AL_WG_THREADS + tid;
				const float stay = cur[s], one = cur[s - 1], two = allow2[i] ? cur[s - 2] : AL_ZERO;
				unsigned k = 0;
				float best = stay;
				if (one > best) { k = 1; best = one; }
				if (two > best) { k = 2; best = two; }
				word |= k << (2 * i);
				nxt[s] = valid[i] ? row[cls[i]] + (best + logf(expf(stay - best) + expf(one - best) + expf(two - best))) : AL_ZERO;
			}
		}
		bpb[(int64_t)t * AL_WG_THREADS + tid] = word;
		__syncthreads();
		float* sw = cur; cur = nxt; nxt = sw;
	}
	__threadfence_block();
	__syncthreads();
	if (tid == 0) {
		int s = 2 * S - 1 + (cur[2 * S] > cur[2 * S - 1] ? 1 : 0);
		int seen = -1;
		for (int t = Tb - 1; t >= 0; --t) {
			if (s != seen) { if (s & 1) ob[s >> 1] = t; seen = s; }
			if (t > 0) s -= (int)((bpb[(int64_t)t * AL_WG_THREADS + (s & (AL_WG_THREADS - 1))] >> (2 * (s / AL_WG_THREADS))) & 3u);
		}
	}
}

int convasr_conv_debug_bits();  // conv.hip
static bool al_use_wg(int S_max) { return al_ns(S_max) < 0 || (convasr_conv_debug_bits() & 32768); }  // debug bit 32768: the workgroup kernel for every target length (tests compare the two)

extern "C" int64_t convasr_ctc_alignment_workspace_bytes(int B, int T, int S_max) {
	return (int64_t)B * T * (al_use_wg(S_max) ? AL_WG_THREADS : 64) * (int64_t)sizeof(unsigned);
}

extern "C" int convasr_ctc_alignment(const float* log_probs, const int64_t* targets, const int64_t* input_lengths, const int64_t* target_lengths, int64_t* alignment,
                                     void* workspace, int B, int T, int C, int S_max, int blank, void* stream) {
	CONVASR_CHECK_ARG(log_probs && targets && input_lengths && target_lengths && alignment && workspace && B > 0 && T > 0 && C > 1 && S_max > 0 && blank >= 0 && blank < C, "ctc_alignment: bad arguments");
	const int ns = al_ns(S_max);
	hipStream_t s = (hipStream_t)stream;
	if (al_use_wg(S_max)) {
		if (2 * S_max + 1 > AL_WG_STATES) return convasr_fail(CONVASR_EUNSUPPORTED, "ctc_alignment: target length %d > %d", S_max, (AL_WG_STATES - 1) / 2);
		const int per = (2 * S_max + 1 + AL_WG_THREADS - 1) / AL_WG_THREADS;
		const size_t smem = 2 * ((size_t)per * AL_WG_THREADS + 2) * sizeof(float);
		static unsigned long long set = 0;
		convasr_allow_160k_lds(reinterpret_cast<const void*>(ctc_alignment_wg_kernel), set);
		hipLaunchKernelGGL(ctc_alignment_wg_kernel, dim3(B), dim3(AL_WG_THREADS), smem, s, log_probs, targets, input_lengths, target_lengths, alignment, (unsigned*)workspace, T, C, S_max, blank);
		CONVASR_CHECK_LAUNCH("ctc_alignment");
		return 0;
	}
#define AL_CASE(NS) case NS: hipLaunchKernelGGL((ctc_alignment_kernel<NS>), dim3(B), dim3(64), 0, s, log_probs, targets, input_lengths, target_lengths, alignment, (unsigned*)workspace, T, C, S_max, blank); break;
	switch (ns) { AL_CASE(1) AL_CASE(2) AL_CASE(3) AL_CASE(4) AL_CASE(6) AL_CASE(8) AL_CASE(12) AL_CASE(16) }
#undef AL_CASE
	CONVASR_CHECK_LAUNCH("ctc_alignment");
	return 0;
}

// ------------------------------------------------------------------------------------------------ GPU-side collate
// out[b][c][t] = t < len[b] ? packed[off[b] + c * len[b] + t] : 0 for a batch whose samples arrived as ONE packed buffer
// (one host-to-device copy per batch instead of one per utterance); `elem` = bytes per element (2: int16 / bf16, 4: fp32, 8: int64).
template <typename E>
__global__ __launch_bounds__(256) void collate_pad_kernel(const E* __restrict__ packed, const int64_t* __restrict__ off, const int64_t* __restrict__ len, E* __restrict__ out,
                                                           int rows, int64_t Tpad) {
	const int b = blockIdx.z, c = blockIdx.y;
	const int64_t n = len[b];
	const E* src = packed + off[b] + (int64_t)c * n;
	E* dst = out + ((int64_t)b * rows + c) * Tpad;
	for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < Tpad; t += (int64_t)gridDim.x * 256) dst[t] = t < n ? src[t] : E(0);
}

extern "C" int convasr_collate_pad(const void* packed, const int64_t* offsets, const int64_t* lengths, void* out, int elem_bytes, int B, int rows, int64_t Tpad,
                                   void* stream) {
	CONVASR_CHECK_ARG(packed && offsets && lengths && out && B > 0 && rows > 0 && Tpad > 0, "collate_pad: bad arguments");
	unsigned gx = (unsigned)ceil_div64(Tpad, 256 * 8);
	if (gx < 1) gx = 1;
	if (gx > 256) gx = 256;
	dim3 grid(gx, rows, B);
	hipStream_t s = (hipStream_t)stream;
	if (elem_bytes == 2) hipLaunchKernelGGL((collate_pad_kernel<short>), grid, dim3(256), 0, s, (const short*)packed, offsets, lengths, (short*)out, rows, Tpad);
	else if (elem_bytes == 4) hipLaunchKernelGGL((collate_pad_kernel<int>), grid, dim3(256), 0, s, (const int*)packed, offsets, lengths, (int*)out, rows, Tpad);
	else if (elem_bytes == 8) hipLaunchKernelGGL((collate_pad_kernel<int64_t>), grid, dim3(256), 0, s, (const int64_t*)packed, offsets, lengths, (int64_t*)out, rows, Tpad);
	else return convasr_fail(CONVASR_EUNSUPPORTED, "collate_pad: element size %d", elem_bytes);
	CONVASR_CHECK_LAUNCH("collate_pad");
	return 0;
}
