// Micro-benchmark: what LOAD pattern reaches the HBM read rate on MI355X? (counterpart of store_bw.hip)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/load_bw.hip -o /tmp/load_bw && /tmp/load_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// workgroup b reads chunk c(b) of `chunk` vectors; inside the chunk every wavefront reads pieces of `piece` vectors, the
// wavefronts' pieces interleaved; XCD: the workgroups an XCD gets (b % 8) read one contiguous eighth of the buffer;
// AHEAD loads in flight per lane
template <bool XCD, int AHEAD, bool NT>
__global__ void __launch_bounds__(256) k_read(const u32x4* in, uint32_t chunk, uint32_t piece, uint32_t* sink) {
	const uint32_t b = blockIdx.x, g = gridDim.x;
	const uint32_t c = XCD ? (b & 7u) * (g >> 3) + (b >> 3) : b;
	const u32x4* o = in + static_cast<uint64_t>(c) * chunk;
	const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
	uint32_t acc = 0;
	// the wavefront's loads in order: piece after piece (its pieces lie 4 * piece apart), 64 vectors per load
	const uint32_t per_piece = piece / 64, loads = chunk / 4 / 64;
	for (uint32_t it = 0; it < loads; it += AHEAD) {
		u32x4 v[AHEAD];
#pragma unroll
		for (int a = 0; a < AHEAD; a++) {
			const uint32_t j = it + a < loads ? it + a : it;
			const uint32_t at = (j / per_piece) * 4 * piece + w * piece + (j % per_piece) * 64 + l;
			v[a] = NT ? __builtin_nontemporal_load(o + at) : o[at];
		}
#pragma unroll
		for (int a = 0; a < AHEAD; a++) acc += v[a].x ^ v[a].y ^ v[a].z ^ v[a].w;
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
	const uint64_t bytes = 1ull << 31, n = bytes / 16;
	u32x4* in; uint32_t* sink;
	CK(hipMalloc(&in, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(in, 1, bytes));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto run = [&](const char* name, auto launch) {
		launch(); hipDeviceSynchronize();
		hipEventRecord(a);
		for (int i = 0; i < 10; i++) launch();
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
		printf("%-64s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
	};
	const uint32_t chunk = 128 * 1024 / 16, g = static_cast<uint32_t>(n / chunk);
	for (uint32_t pk : { 1u, 4u, 32u }) {
		char nm[96];
		const uint32_t piece = pk * 1024 / 16;
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 4 ahead, nt", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<false, 4, true>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 4 ahead", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<false, 4, false>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 4 ahead, XCD-contiguous", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<true, 4, false>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 4 ahead, XCD-contiguous, nt", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<true, 4, true>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 8 ahead, XCD-contiguous", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<true, 8, false>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
		snprintf(nm, 96, "chunk 128 KiB, pieces of %u KiB, 1 ahead, XCD-contiguous", pk); run(nm, [&] { hipLaunchKernelGGL((k_read<true, 1, false>), dim3(g), dim3(256), 0, 0, in, chunk, piece, sink); });
	}
	return 0;
}
