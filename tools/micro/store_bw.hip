// Micro-benchmark: what store pattern reaches the HBM write rate on MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_bw.hip -o /tmp/store_bw && /tmp/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride over 16-byte vectors
template <bool NT>
__global__ void __launch_bounds__(256) k_stride(u32x4* out, uint64_t n) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) st<NT>(out + i, v);
}
// one workgroup per contiguous chunk of `chunk` vectors
template <bool NT, int PRO>
__global__ void __launch_bounds__(256) k_chunk(u32x4* out, uint32_t chunk, const uint32_t* src) {
	__shared__ uint32_t s[256];
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	if (PRO) {      // a dependent load chain and barriers in front, like the paint kernel's prologue
		uint32_t a = src[(blockIdx.x * 256u + threadIdx.x) & 0xFFFFF];
		if (PRO > 1) a = src[a & 0xFFFFF];
		s[threadIdx.x] = a;
		__syncthreads();
		v.x = s[(threadIdx.x + 1) & 255];
		__syncthreads();
	}
	u32x4* o = out + static_cast<uint64_t>(blockIdx.x) * chunk;
	for (uint32_t i = threadIdx.x; i < chunk; i += 256) st<NT>(o + i, v);
}
// same, every thread writes two adjacent vectors
template <bool NT>
__global__ void __launch_bounds__(256) k_chunk2(u32x4* out, uint32_t chunk) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	u32x4* o = out + static_cast<uint64_t>(blockIdx.x) * chunk;
	for (uint32_t i = threadIdx.x * 2; i < chunk; i += 512) { st<NT>(o + i, v); st<NT>(o + i + 1, v); }
}

// one workgroup per chunk, every wavefront streams its own contiguous quarter of it
template <bool NT>
__global__ void __launch_bounds__(256) k_chunk_wave(u32x4* out, uint32_t chunk) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	const uint32_t per = chunk / 4;
	u32x4* o = out + static_cast<uint64_t>(blockIdx.x) * chunk + (threadIdx.x >> 6) * per;
	for (uint32_t i = threadIdx.x & 63; i < per; i += 64) st<NT>(o + i, v);
}
// workgroups of 1024 threads per chunk
template <bool NT>
__global__ void __launch_bounds__(1024) k_chunk1024(u32x4* out, uint32_t chunk) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	u32x4* o = out + static_cast<uint64_t>(blockIdx.x) * chunk;
	for (uint32_t i = threadIdx.x; i < chunk; i += 1024) st<NT>(o + i, v);
}
// the chunk of workgroup b is chunk (b % 8) * (g / 8) + b / 8: the workgroups an XCD gets (b % 8) write one contiguous eighth of the buffer
template <bool NT>
__global__ void __launch_bounds__(256) k_chunk_xcd(u32x4* out, uint32_t chunk) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	const uint32_t b = blockIdx.x, g = gridDim.x;
	const uint32_t c = (b & 7u) * (g >> 3) + (b >> 3);
	u32x4* o = out + static_cast<uint64_t>(c) * chunk;
	for (uint32_t i = threadIdx.x; i < chunk; i += 256) st<NT>(o + i, v);
}

// general form: workgroup b writes chunk c(b) (XCD: see k_chunk_xcd); inside the chunk every wavefront streams contiguous
// pieces of `piece` vectors, the wavefronts' pieces interleaved (piece = chunk / 4: a quarter per wavefront; piece = 64:
// k_chunk's order)
template <bool NT, bool XCD>
__global__ void __launch_bounds__(256) k_piece(u32x4* out, uint32_t chunk, uint32_t piece) {
	u32x4 v = { threadIdx.x, 1, 2, 3 };
	const uint32_t b = blockIdx.x, g = gridDim.x;
	const uint32_t c = XCD ? (b & 7u) * (g >> 3) + (b >> 3) : b;
	u32x4* o = out + static_cast<uint64_t>(c) * chunk;
	const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
	for (uint32_t p0 = w * piece; p0 < chunk; p0 += 4 * piece)
		for (uint32_t i = l; i < piece; i += 64) st<NT>(o + p0 + i, v);
}

int main() {
	const uint64_t bytes = 1ull << 31, n = bytes / 16;
	u32x4* out; uint32_t* src;
	CK(hipMalloc(&out, bytes)); CK(hipMalloc(&src, 4 << 20)); CK(hipMemset(src, 0, 4 << 20));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto run = [&](const char* name, auto launch) {
		launch(); hipDeviceSynchronize();
		hipEventRecord(a);
		for (int i = 0; i < 10; i++) launch();
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
		printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
	};
	run("stride 2048 wg plain", [&] { hipLaunchKernelGGL(k_stride<false>, dim3(2048), dim3(256), 0, 0, out, n); });
	run("stride 2048 wg nt", [&] { hipLaunchKernelGGL(k_stride<true>, dim3(2048), dim3(256), 0, 0, out, n); });
	run("stride 8192 wg plain", [&] { hipLaunchKernelGGL(k_stride<false>, dim3(8192), dim3(256), 0, 0, out, n); });
	for (uint32_t kb : { 16u, 64u, 112u, 128u, 512u }) {
		const uint32_t chunk = kb * 1024 / 16, g = static_cast<uint32_t>(n / chunk);
		char nm[64];
		snprintf(nm, 64, "chunk %u KiB plain", kb); run(nm, [&] { hipLaunchKernelGGL((k_chunk<false, 0>), dim3(g), dim3(256), 0, 0, out, chunk, src); });
		snprintf(nm, 64, "chunk %u KiB nt", kb); run(nm, [&] { hipLaunchKernelGGL((k_chunk<true, 0>), dim3(g), dim3(256), 0, 0, out, chunk, src); });
		snprintf(nm, 64, "chunk %u KiB plain + 1 load prologue", kb); run(nm, [&] { hipLaunchKernelGGL((k_chunk<false, 1>), dim3(g), dim3(256), 0, 0, out, chunk, src); });
		snprintf(nm, 64, "chunk %u KiB nt + 2 load prologue", kb); run(nm, [&] { hipLaunchKernelGGL((k_chunk<true, 2>), dim3(g), dim3(256), 0, 0, out, chunk, src); });
		snprintf(nm, 64, "chunk %u KiB plain 2 vec/thread", kb); run(nm, [&] { hipLaunchKernelGGL(k_chunk2<false>, dim3(g), dim3(256), 0, 0, out, chunk); });
	}
	{
		const uint32_t chunk = 128 * 1024 / 16, g = static_cast<uint32_t>(n / chunk);
		run("chunk 128 KiB, a quarter per wavefront, plain", [&] { hipLaunchKernelGGL(k_chunk_wave<false>, dim3(g), dim3(256), 0, 0, out, chunk); });
		run("chunk 128 KiB, a quarter per wavefront, nt", [&] { hipLaunchKernelGGL(k_chunk_wave<true>, dim3(g), dim3(256), 0, 0, out, chunk); });
		run("chunk 128 KiB, 1024 threads, plain", [&] { hipLaunchKernelGGL(k_chunk1024<false>, dim3(g), dim3(1024), 0, 0, out, chunk); });
		run("chunk 128 KiB, 1024 threads, nt", [&] { hipLaunchKernelGGL(k_chunk1024<true>, dim3(g), dim3(1024), 0, 0, out, chunk); });
		run("chunk 128 KiB, XCD-contiguous, plain", [&] { hipLaunchKernelGGL(k_chunk_xcd<false>, dim3(g), dim3(256), 0, 0, out, chunk); });
		run("chunk 128 KiB, XCD-contiguous, nt", [&] { hipLaunchKernelGGL(k_chunk_xcd<true>, dim3(g), dim3(256), 0, 0, out, chunk); });
		for (uint32_t pk : { 1u, 4u, 8u, 16u, 32u }) {
			char nm[96];
			const uint32_t piece = pk * 1024 / 16;
			snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB per wavefront, plain", pk); run(nm, [&] { hipLaunchKernelGGL((k_piece<false, false>), dim3(g), dim3(256), 0, 0, out, chunk, piece); });
			snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB per wavefront, nt", pk); run(nm, [&] { hipLaunchKernelGGL((k_piece<true, false>), dim3(g), dim3(256), 0, 0, out, chunk, piece); });
			snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, XCD-contiguous, plain", pk); run(nm, [&] { hipLaunchKernelGGL((k_piece<false, true>), dim3(g), dim3(256), 0, 0, out, chunk, piece); });
			snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, XCD-contiguous, nt", pk); run(nm, [&] { hipLaunchKernelGGL((k_piece<true, true>), dim3(g), dim3(256), 0, 0, out, chunk, piece); });
		}
		const uint32_t chunk16 = 16 * 1024 / 16, g16 = static_cast<uint32_t>(n / chunk16);
		run("chunk 16 KiB, XCD-contiguous, plain", [&] { hipLaunchKernelGGL(k_chunk_xcd<false>, dim3(g16), dim3(256), 0, 0, out, chunk16); });
		run("chunk 16 KiB, XCD-contiguous, nt", [&] { hipLaunchKernelGGL(k_chunk_xcd<true>, dim3(g16), dim3(256), 0, 0, out, chunk16); });
	}
	return 0;
}
