// Host-side common definitions for libcrackle_amd: error plumbing, the .ckl v1
// stream layout (reference: src/header.hpp, src/lib.hpp, src/crc.hpp) and small
// RAII helpers around HIP allocations.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/crackle_amd.h"

namespace ckl {

struct Error : public std::runtime_error {
	int status;
	Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

#define CKL_HIP(expr) \
	do { \
		hipError_t _e = (expr); \
		if (_e != hipSuccess) { \
			throw ::ckl::Error(CKL_ERR_RUNTIME, std::string("crackle_amd: HIP error '") + hipGetErrorString(_e) + "' at " #expr); \
		} \
	} while (0)

// ---- little-endian + widths (src/lib.hpp:11-145, 236-247) --------------------
inline int byte_width(uint64_t x) {
	if (x <= 0xFFull) return 1;
	if (x <= 0xFFFFull) return 2;
	if (x <= 0xFFFFFFFFull) return 4;
	return 8;
}
inline int ilog2w(int w) { return w == 1 ? 0 : w == 2 ? 1 : w == 4 ? 2 : 3; }

inline uint64_t rd_le(const uint8_t* p, int w) {
	uint64_t v = 0;
	for (int i = 0; i < w; i++) v |= static_cast<uint64_t>(p[i]) << (8 * i);
	return v;
}
inline void put_le(std::vector<uint8_t>& b, uint64_t v, int w) {
	for (int i = 0; i < w; i++) b.push_back(static_cast<uint8_t>((v >> (8 * i)) & 0xFF));
}

// ---- checksums (src/crc.hpp:23-57) -------------------------------------------
uint8_t crc8(const uint8_t* data, uint64_t n);
uint32_t crc32c(const uint8_t* data, uint64_t n);

// GF(2)[x] / P(x) arithmetic in the reflected CRC-32C representation
// (bit 31 <-> x^0).  Feeding n zero bytes to the table-driven CRC register is a
// multiplication by x^(8n) mod P — used to combine per-tile CRCs on device.
constexpr uint32_t kCrcPoly = 0x82F63B78u;
inline uint32_t gf_mul(uint32_t a, uint32_t b) {
	uint32_t r = 0;
	for (int i = 0; i < 32; i++) {
		if (a & (0x80000000u >> i)) r ^= b;
		b = (b >> 1) ^ ((b & 1u) ? kCrcPoly : 0u);
	}
	return r;
}
inline uint32_t gf_xpow(uint64_t nbits) {   // x^nbits mod P
	uint32_t r = 0x80000000u, base = 0x40000000u;
	while (nbits) {
		if (nbits & 1) r = gf_mul(r, base);
		base = gf_mul(base, base);
		nbits >>= 1;
	}
	return r;
}

// ---- header (src/header.hpp:35-308) -------------------------------------------
enum LabelFormat { FLAT = 0, PINS_FIXED_WIDTH = 1, PINS_VARIABLE_WIDTH = 2 };
enum CrackFormat { IMPERMISSIBLE = 0, PERMISSIBLE = 1 };

struct Header {
	static constexpr uint64_t kBytes = 29, kBytesV0 = 24;
	uint8_t format_version = 1;
	int label_format = FLAT;
	int crack_format = IMPERMISSIBLE;
	bool is_signed = false;
	int data_width = 1, stored_data_width = 1;
	uint32_t sx = 0, sy = 0, sz = 0;
	uint8_t log2_grid_size = 31;
	uint64_t num_label_bytes = 0;
	bool fortran_order = true;
	int markov_model_order = 0;
	bool is_sorted = true;

	static Header parse(const uint8_t* buf, uint64_t n);    // throws Error(CKL_ERR_FORMAT)
	void write(std::vector<uint8_t>& out) const;            // always v1, 29 bytes

	uint64_t header_bytes() const { return format_version == 0 ? kBytesV0 : kBytes; }
	uint64_t grid_index_bytes() const { return (static_cast<uint64_t>(sz) + (format_version == 0 ? 0 : 1)) * 4; }
	uint64_t markov_model_bytes() const {
		if (markov_model_order == 0) return 0;
		return ((1ull << (2 * markov_model_order)) * 5 + 4) / 8;
	}
	int pin_index_width() const {   // 32-bit product on purpose (src/header.hpp:190-192)
		uint32_t v = sx * sy * sz;
		return byte_width(v);
	}
	uint64_t voxels() const { return static_cast<uint64_t>(sx) * sy * sz; }
	// True when header, z-index, label section, markov model and the crc tail fit into n bytes.
	// The fields come from an untrusted stream (the crc8 is trivially forged): nothing is summed
	// that could wrap — the fixed parts are at most a few times 2^34, num_label_bytes is compared by
	// subtraction.  After this every offset below num_label_bytes / n is safe to add.
	bool layout_fits(uint64_t n) const {
		const uint64_t tail = format_version == 0 ? 0 : 4ull * (static_cast<uint64_t>(sz) + 1);
		const uint64_t fixed = header_bytes() + grid_index_bytes() + markov_model_bytes() + tail;
		return fixed <= n && num_label_bytes <= n - fixed;
	}
};

// ---- device buffers -------------------------------------------------------------
// Device allocations go through a small per-device pool: sessions are created and
// destroyed per call by the one-shot API, and releasing / re-acquiring gigabytes of
// scratch through hipFree / hipMalloc costs tens of milliseconds each time.
void* host_out_alloc(size_t bytes);        // pinned (cached) host buffer for results; release with ckl_free / host_out_free
void host_out_free(void* p);               // also accepts plain malloc'd pointers
bool host_out_is_pinned(const void* p);    // a block of host_out_alloc that is page-locked and mapped into the device's address space
// Small host -> device copies on the hot paths (descriptor tables, headers, crc tails: a few KiB to a few hundred): when
// the source lies in a pinned, device-mapped block of host_out_alloc (`block`: that block's base pointer) a one-workgroup-
// per-4-KiB kernel reads it over the link; else hipMemcpyAsync.  (hipMemcpyAsync of 40 KiB from pinned memory was seen to
// block its caller for 5 - 8 ms once in a few dozen calls on this stack — 9 % of a bench run's value when it happened.)
// (Defined in ckl_upload.hip, a translation unit of its own — see there.  Callers pass a null `block` for pin-label streams, whose
// set-up and assembly move tens of megabytes with the copy engines anyway: they keep hipMemcpyAsync.)
void upload_small(void* dst_device, const void* src_host, size_t bytes, hipStream_t s, const void* block);
void* pool_alloc(size_t bytes, int* device = nullptr);      // throws Error on failure; *device: the device the block lives on (the current one)
void pool_free(void* p, size_t bytes, int device = -1);     // returns the block to the pool of its device (-1: the current device)
void pool_trim();                          // hipFree everything that is pooled

template <typename T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	int device = -1;           // the device the block was allocated on: it goes back to that device's pool whatever device is current then
	bool borrowed = false;     // a view of memory somebody else owns (a part of a packed block, a caller's buffer): never freed here
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release() {
		if (p && !borrowed) pool_free(p, (n ? n : 1) * sizeof(T), device);
		p = nullptr;
		n = 0;
		borrowed = false;
	}
	// grow-only allocation (contents are not preserved)
	void ensure(size_t count) {
		if (count <= n && p && !borrowed) return;
		release();
		p = static_cast<T*>(pool_alloc((count ? count : 1) * sizeof(T), &device));
		n = count;
	}
	void borrow(T* ptr, size_t count) { release(); p = ptr; n = count; borrowed = true; }
	size_t bytes() const { return n * sizeof(T); }
};

void set_last_error(const std::string& msg);
int select_device(int device);   // throws CKL_ERR_NO_DEVICE
// Orders `s` after everything already submitted to the device's default (null) stream:
// the sessions run on their own non-blocking streams, and callers such as PyTorch fill /
// clear the buffers they hand over on the default stream.
void wait_for_default_stream(hipStream_t s, hipEvent_t scratch_event);

// markov model tables (src/markov.hpp:43-68, 222-266, 325-420)
extern const uint8_t kMarkovLUT[24];
std::vector<uint8_t> markov_model_from_stored(const uint8_t* stream, uint64_t nbytes, int order);   // rows x 4, rank -> symbol
std::vector<uint8_t> markov_stats_to_model(const uint32_t* stats, size_t rows);                     // rows x 4, symbol -> rank
std::vector<uint8_t> markov_model_to_stored(const std::vector<uint8_t>& model);

}  // namespace ckl
