// upload_small: small host -> device copies by a kernel that reads a pinned, device-mapped host block (see ckl_common.hpp).
// A translation unit of its own: added to ckl_decode.hip, the mere presence of this kernel in that file's code object made
// the decoder set-up of 2048 x 2048 x 256 pin streams take 10 - 20 instead of 2.5 ms in most processes (not understood;
// measured with the kernel never launched), and ckl_common.hip is also compiled as plain C++ by the sanitizer tests.
#include "ckl_common.hpp"

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

namespace ckl {

namespace {
__global__ void __launch_bounds__(256) k_upload_small(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
	const size_t i = (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) * 16;
	if (i >= bytes) return;
	if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && i + 16 <= bytes) {
		*reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
		return;
	}
	for (size_t k = i; k < bytes && k < i + 16; k++) dst[k] = src[k];
}
}

void upload_small(void* dst_device, const void* src_host, size_t bytes, hipStream_t s, const void* block) {
	if (bytes == 0) return;
	if (block && host_out_is_pinned(block) && !getenv("CKL_UPLOAD_MEMCPY")) {
		hipLaunchKernelGGL(k_upload_small, dim3(static_cast<uint32_t>((bytes + 4095) / 4096)), dim3(256), 0, s,
			static_cast<const uint8_t*>(src_host), static_cast<uint8_t*>(dst_device), bytes);
		return;
	}
	CKL_HIP(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, s));
}


}  // namespace ckl
