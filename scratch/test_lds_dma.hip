// Does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for out-of-range lanes (needed for the conv halo)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out, int nbytes) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	unsigned* l = (unsigned*)smem;
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = 0xdeadbeefu;
	__syncthreads();
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	// lane reads 16 B at byte offset: wave 0 in range, wave 1 straddles the end, wave 2 negative offsets (wrap), wave 3 permuted lanes
	int off = lane * 16;
	if (wave == 1) off = nbytes - 512 + lane * 16;
	if (wave == 2) off = -256 + lane * 16;
	if (wave == 3) off = (lane ^ 5) * 16;
	__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = l[i];
}
int main() {
	const int n = 4096;  // bytes
	std::vector<unsigned> h(n / 4);
	for (int i = 0; i < n / 4; ++i) h[i] = i + 1;
	unsigned *d, *o;
	hipMalloc(&d, n + 4096); hipMalloc(&o, 4096);
	hipMemset(d, 0x77, n + 4096);
	hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, o, n);
	std::vector<unsigned> r(1024);
	hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int w = 0; w < 4; ++w) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
		long off = lane * 16; if (w == 1) off = n - 512 + lane * 16; if (w == 2) off = -256 + lane * 16; if (w == 3) off = (lane ^ 5) * 16;
		unsigned want = (off >= 0 && off + 16 <= n) ? (unsigned)(off / 4 + j + 1) : 0u;
		unsigned got = r[w * 256 + lane * 4 + j];
		if (got != want) { if (bad < 10) printf("wave %d lane %d j %d: got %08x want %08x\n", w, lane, j, got, want); ++bad; }
	}
	printf("lds-dma check: %d mismatches\n", bad);
	return bad != 0;
}
