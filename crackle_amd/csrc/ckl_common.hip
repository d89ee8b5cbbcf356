// Host-side pieces of libcrackle_amd that are not kernels: error plumbing, device
// selection, the 29-byte header, checksums, and the markov model tables.
#include "ckl_common.hpp"

#include <algorithm>
#include <mutex>

namespace ckl {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

int select_device(int device) {
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		(void)hipGetLastError();
		throw Error(CKL_ERR_NO_DEVICE, "crackle_amd: no usable HIP device (this library has no CPU fallback)");
	}
	if (device < 0 || device >= count) {
		throw Error(CKL_ERR_ARG, "crackle_amd: device ordinal " + std::to_string(device) + " out of range (" + std::to_string(count) + " devices)");
	}
	CKL_HIP(hipSetDevice(device));
	return count;
}

void wait_for_default_stream(hipStream_t s, hipEvent_t ev) {
	CKL_HIP(hipEventRecord(ev, nullptr));
	CKL_HIP(hipStreamWaitEvent(s, ev, 0));
}

// ---- device memory pool ------------------------------------------------------------
namespace {
struct PoolBlock { void* p; size_t bytes; int device; };
std::mutex g_pool_mutex;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = 96ull << 30;    // of 288 GB of HBM
constexpr size_t kPoolMaxBlocks = 512;
}

// Testing (CKL_POOL_POISON=1): every block handed out is filled with 0xA5 first, so that a kernel which
// reads memory nobody wrote fails the same way every time instead of only after some other volume.
static void* pool_poison(void* p, size_t bytes) {
	static const bool on = getenv("CKL_POOL_POISON") != nullptr;
	if (on && p) { (void)hipMemset(p, 0xA5, bytes); (void)hipDeviceSynchronize(); }
	return p;
}

void* pool_alloc(size_t bytes, int* device) {
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (device) *device = dev;
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		// best fit among blocks that are not wastefully large for the request
		size_t best = g_pool.size();
		for (size_t i = 0; i < g_pool.size(); i++) {
			const PoolBlock& b = g_pool[i];
			if (b.device != dev || b.bytes < bytes || b.bytes > 2 * bytes + (1u << 20)) continue;
			if (best == g_pool.size() || b.bytes < g_pool[best].bytes) best = i;
		}
		if (best != g_pool.size()) {
			void* p = g_pool[best].p;
			g_pool_bytes -= g_pool[best].bytes;
			g_pool[best] = g_pool.back();
			g_pool.pop_back();
			return pool_poison(p, bytes);
		}
	}
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		pool_trim();                       // give pooled memory back and retry once
		e = hipMalloc(&p, bytes);
	}
	if (e != hipSuccess) {
		(void)hipGetLastError();
		throw Error(CKL_ERR_RUNTIME, std::string("crackle_amd: hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
	}
	return pool_poison(p, bytes);
}

void pool_free(void* p, size_t bytes, int device) {
	if (!p) return;
	int dev = device;
	if (dev < 0) (void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		if (g_pool.size() < kPoolMaxBlocks && g_pool_bytes + bytes <= kPoolMaxBytes) {
			g_pool.push_back({ p, bytes, dev });
			g_pool_bytes += bytes;
			return;
		}
	}
	(void)hipFree(p);
}

void pool_trim() {
	std::vector<PoolBlock> blocks;
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		blocks.swap(g_pool);
		g_pool_bytes = 0;
	}
	for (const PoolBlock& b : blocks) (void)hipFree(b.p);
}

// ---- host output buffers -----------------------------------------------------------
// Streams are handed to the caller in pinned host memory (device -> host copies land in
// it directly, at full link rate) and come back through ckl_free into a small cache, so
// that a codec called in a loop does not pay for page faults / pinning on every call.
namespace {
struct HostBlock { void* p; size_t bytes; };
std::mutex g_host_mutex;
std::vector<HostBlock> g_host_live, g_host_free;
constexpr size_t kHostCacheBlocks = 12;
constexpr size_t kHostCacheMaxBytes = 1ull << 30;
}

void* host_out_alloc(size_t bytes) {
	if (bytes == 0) bytes = 1;
	{
		std::lock_guard<std::mutex> lock(g_host_mutex);
		size_t best = g_host_free.size();
		for (size_t i = 0; i < g_host_free.size(); i++) {
			if (g_host_free[i].bytes < bytes) continue;
			if (best == g_host_free.size() || g_host_free[i].bytes < g_host_free[best].bytes) best = i;
		}
		if (best != g_host_free.size()) {
			HostBlock b = g_host_free[best];
			g_host_free[best] = g_host_free.back();
			g_host_free.pop_back();
			g_host_live.push_back(b);
			return b.p;
		}
	}
	const size_t cap = bytes < (1u << 20) ? bytes : bytes + bytes / 4;
	void* p = nullptr;
	static const bool trace = getenv("CKL_POOL_TRACE") != nullptr;      // diagnostic: allocations that miss the cache, on stderr
	if (trace) fprintf(stderr, "[ckl pool] new host block of %zu bytes (%zu cached blocks)\n", bytes, g_host_free.size());
	if (bytes >= (64u << 10) && hipHostMalloc(&p, cap, hipHostMallocDefault) == hipSuccess && p) {
		std::lock_guard<std::mutex> lock(g_host_mutex);
		g_host_live.push_back({ p, cap });
		return p;
	}
	(void)hipGetLastError();
	p = malloc(bytes);
	if (!p) throw Error(CKL_ERR_RUNTIME, "crackle_amd: out of host memory");
	return p;
}

bool host_out_is_pinned(const void* p) {
	std::lock_guard<std::mutex> lock(g_host_mutex);
	for (const HostBlock& b : g_host_live) if (b.p == p) return true;
	return false;
}

void host_out_free(void* p) {
	if (!p) return;
	HostBlock b = { nullptr, 0 };
	bool drop = false;
	{
		std::lock_guard<std::mutex> lock(g_host_mutex);
		for (size_t i = 0; i < g_host_live.size(); i++) {
			if (g_host_live[i].p != p) continue;
			b = g_host_live[i];
			g_host_live[i] = g_host_live.back();
			g_host_live.pop_back();
			break;
		}
		if (b.p) {
			size_t cached = 0;
			for (const HostBlock& f : g_host_free) cached += f.bytes;
			if (g_host_free.size() < kHostCacheBlocks && cached + b.bytes <= kHostCacheMaxBytes) g_host_free.push_back(b);
			else drop = true;
		}
	}
	if (!b.p) free(p);                       // a plain malloc'd buffer
	else if (drop) {
		if (getenv("CKL_POOL_TRACE")) fprintf(stderr, "[ckl pool] host block of %zu bytes dropped from the cache\n", b.bytes);
		(void)hipHostFree(b.p);
	}
}

// ---- checksums -------------------------------------------------------------------
// crc8 (src/crc.hpp:23-37): poly 0xe7 (implicit +1, reflected), init 0xFF, no xorout
uint8_t crc8(const uint8_t* data, uint64_t n) {
	uint8_t crc = 0xFF;
	while (n--) {
		crc ^= *data++;
		for (int k = 0; k < 8; k++) crc = (crc & 1) ? static_cast<uint8_t>((crc >> 1) ^ 0xe7) : static_cast<uint8_t>(crc >> 1);
	}
	return crc;
}

static uint32_t g_tab[8][256];
static std::once_flag g_tab_once;
static void crc_tab_init() {
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ kCrcPoly : (c >> 1);
		g_tab[0][i] = c;
	}
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = g_tab[0][i];
		for (int t = 1; t < 8; t++) {
			c = g_tab[0][c & 0xFF] ^ (c >> 8);
			g_tab[t][i] = c;
		}
	}
}
// SSE4.2 crc32 instruction (the same polynomial).
#if defined(__x86_64__)
// The crc32 instruction has a latency of three cycles and a throughput of one per cycle: three
// independent chains over three thirds of the buffer run at full rate, and the thirds combine
// through  state(c, M) = c * x^bits(M) + state(0, M)  in GF(2)[x]/P.
__attribute__((target("sse4.2"))) static uint32_t crc32c_sse42(const uint8_t* data, uint64_t n) {
	uint64_t crc = 0xFFFFFFFFu;
	if (n >= 3 * 1024) {
		const uint64_t words = n / 24;             // 8-byte words per third
		const uint8_t* pa = data;
		const uint8_t* pb = data + 8 * words;
		const uint8_t* pc = data + 16 * words;
		uint64_t a = crc, b = 0, c = 0;
		for (uint64_t i = 0; i < words; i++) {
			uint64_t wa, wb, wc;
			memcpy(&wa, pa + 8 * i, 8); memcpy(&wb, pb + 8 * i, 8); memcpy(&wc, pc + 8 * i, 8);
			a = __builtin_ia32_crc32di(a, wa);
			b = __builtin_ia32_crc32di(b, wb);
			c = __builtin_ia32_crc32di(c, wc);
		}
		const uint32_t shift = gf_xpow(64ull * words);
		uint32_t st = gf_mul(static_cast<uint32_t>(a), shift) ^ static_cast<uint32_t>(b);
		st = gf_mul(st, shift) ^ static_cast<uint32_t>(c);
		crc = st;
		data += 24 * words;
		n -= 24 * words;
	}
	while (n >= 8) {
		uint64_t w;
		memcpy(&w, data, 8);
		crc = __builtin_ia32_crc32di(crc, w);
		data += 8;
		n -= 8;
	}
	uint32_t c = static_cast<uint32_t>(crc);
	while (n--) c = __builtin_ia32_crc32qi(c, *data++);
	return ~c;
}
#endif

uint32_t crc32c(const uint8_t* data, uint64_t n) {
#if defined(__x86_64__)
	static const bool have_sse42 = __builtin_cpu_supports("sse4.2");
	if (have_sse42) return crc32c_sse42(data, n);
#endif
	std::call_once(g_tab_once, crc_tab_init);
	uint32_t crc = 0xFFFFFFFFu;
	while (n >= 8) {
		uint64_t w;
		memcpy(&w, data, 8);
		w ^= crc;
		crc = g_tab[7][w & 0xFF] ^ g_tab[6][(w >> 8) & 0xFF] ^ g_tab[5][(w >> 16) & 0xFF] ^ g_tab[4][(w >> 24) & 0xFF]
			^ g_tab[3][(w >> 32) & 0xFF] ^ g_tab[2][(w >> 40) & 0xFF] ^ g_tab[1][(w >> 48) & 0xFF] ^ g_tab[0][(w >> 56) & 0xFF];
		data += 8;
		n -= 8;
	}
	while (n--) crc = g_tab[0][(crc ^ *data++) & 0xFF] ^ (crc >> 8);
	return ~crc;
}

// ---- header ------------------------------------------------------------------------
Header Header::parse(const uint8_t* buf, uint64_t n) {
	if (n < kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
	Header h;
	const bool valid_magic = (buf[0] == 'c' && buf[1] == 'r' && buf[2] == 'k' && buf[3] == 'l');
	h.format_version = buf[4];
	if (!valid_magic || h.format_version > 1) throw Error(CKL_ERR_FORMAT, "crackle: Data stream is not valid. Unable to decompress.");
	if (h.format_version == 1 && n < kBytes) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
	const uint16_t fmt = static_cast<uint16_t>(rd_le(buf + 5, 2));
	h.sx = static_cast<uint32_t>(rd_le(buf + 7, 4));
	h.sy = static_cast<uint32_t>(rd_le(buf + 11, 4));
	h.sz = static_cast<uint32_t>(rd_le(buf + 15, 4));
	h.log2_grid_size = buf[19];
	h.num_label_bytes = h.format_version == 0 ? rd_le(buf + 20, 4) : rd_le(buf + 20, 8);
	h.data_width = 1 << (fmt & 3);
	h.stored_data_width = 1 << ((fmt >> 2) & 3);
	h.crack_format = (fmt >> 4) & 1;
	h.label_format = (fmt >> 5) & 3;
	h.fortran_order = (fmt >> 7) & 1;
	h.is_signed = (fmt >> 8) & 1;
	h.markov_model_order = (fmt >> 9) & 15;
	h.is_sorted = !((fmt >> 13) & 1);
	if (h.format_version == 0) return h;
	if (crc8(buf + 5, 28 - 5) != buf[28]) {
		throw Error(CKL_ERR_FORMAT, "crackle: CRC8 check failed. Header may be corrupted. (~4.1% chance of a false positive for a single bit flip).");
	}
	return h;
}

void Header::write(std::vector<uint8_t>& out) const {
	const size_t base = out.size();
	out.push_back('c'); out.push_back('r'); out.push_back('k'); out.push_back('l');
	uint16_t fmt = 0;
	fmt |= static_cast<uint16_t>(ilog2w(data_width));
	fmt |= static_cast<uint16_t>(ilog2w(stored_data_width) << 2);
	fmt |= static_cast<uint16_t>(crack_format << 4);
	fmt |= static_cast<uint16_t>(label_format << 5);
	fmt |= static_cast<uint16_t>((fortran_order ? 1 : 0) << 7);
	fmt |= static_cast<uint16_t>((is_signed ? 1 : 0) << 8);
	fmt |= static_cast<uint16_t>((markov_model_order & 15) << 9);
	fmt |= static_cast<uint16_t>((is_sorted ? 0 : 1) << 13);
	// version 0 (src/header.hpp:213-245): 4-byte num_label_bytes, no crc8 — kept when a version 0 stream is
	// rewritten (reencode), as long as the label section's size fits the field
	const bool v0 = format_version == 0 && num_label_bytes <= 0xFFFFFFFFull;
	out.push_back(v0 ? 0 : 1);
	put_le(out, fmt, 2);
	put_le(out, sx, 4);
	put_le(out, sy, 4);
	put_le(out, sz, 4);
	out.push_back(log2_grid_size);
	if (v0) { put_le(out, num_label_bytes, 4); return; }
	put_le(out, num_label_bytes, 8);
	out.push_back(crc8(out.data() + base + 5, kBytes - 1 - 5));
}

// ---- markov model tables (src/markov.hpp) ------------------------------------------
// permutations of (0,1,2,3) in lexicographic order, element i in bits 2i..2i+1
const uint8_t kMarkovLUT[24] = {
	0xE4, 0xB4, 0xD8, 0x78, 0x9C, 0x6C, 0xE1, 0xB1, 0xC9, 0x39, 0x8D, 0x2D,
	0xD2, 0x72, 0xC6, 0x36, 0x4E, 0x1E, 0x93, 0x63, 0x87, 0x27, 0x4B, 0x1B
};

// from_stored_model (src/markov.hpp:382-420): 5-bit fields, LSB first; rows are rank -> symbol
std::vector<uint8_t> markov_model_from_stored(const uint8_t* stream, uint64_t nbytes, int order) {
	const size_t rows = static_cast<size_t>(1) << (2 * order);
	std::vector<uint8_t> model(rows * 4, 0);
	for (size_t r = 0; r < rows; r++) {
		const uint64_t bit = static_cast<uint64_t>(r) * 5;
		const uint64_t byte = bit >> 3;
		const int pos = static_cast<int>(bit & 7);
		uint32_t v = 0;
		if (byte < nbytes) v = stream[byte];
		if (byte + 1 < nbytes) v |= static_cast<uint32_t>(stream[byte + 1]) << 8;
		const uint32_t decoded = (v >> pos) & 31;
		const uint8_t row = decoded < 24 ? kMarkovLUT[decoded] : 0;
		for (int k = 0; k < 4; k++) model[r * 4 + k] = (row >> (2 * k)) & 3;
	}
	return model;
}

// stats_to_model (src/markov.hpp:222-266).  The reference sorts the four
// (symbol, count) pairs with std::sort and the comparator `a.count >= b.count`,
// which for four elements is libstdc++'s insertion sort: count descending, ties
// resolved towards the larger symbol.  Result: symbol -> rank.
std::vector<uint8_t> markov_stats_to_model(const uint32_t* stats, size_t rows) {
	std::vector<uint8_t> model(rows * 4);
	for (size_t r = 0; r < rows; r++) {
		int sym[4];
		uint32_t cnt[4];
		int m = 0;
		for (int l = 0; l < 4; l++) {
			const uint32_t c = stats[r * 4 + l];
			int p = m;
			while (p > 0 && c >= cnt[p - 1]) { sym[p] = sym[p - 1]; cnt[p] = cnt[p - 1]; p--; }
			sym[p] = l; cnt[p] = c;
			m++;
		}
		for (int j = 0; j < 4; j++) model[r * 4 + sym[j]] = static_cast<uint8_t>(j);
	}
	return model;
}

// to_stored_model (src/markov.hpp:325-380)
std::vector<uint8_t> markov_model_to_stored(const std::vector<uint8_t>& model) {
	std::vector<uint8_t> out;
	const size_t rows = model.size() / 4;
	int pos = 0;
	uint32_t acc = 0;
	for (size_t r = 0; r < rows; r++) {
		uint8_t key = 0;
		for (int s = 0; s < 4; s++) key |= static_cast<uint8_t>(s << (2 * model[r * 4 + s]));
		int idx = -1;
		for (int i = 0; i < 24; i++) if (kMarkovLUT[i] == key) { idx = i; break; }
		if (idx < 0) throw Error(CKL_ERR_RUNTIME, "Corrupted model.");
		acc |= static_cast<uint32_t>(idx) << pos;
		pos += 5;
		if (pos > 8) { out.push_back(static_cast<uint8_t>(acc)); pos -= 8; acc >>= 8; }
	}
	if (pos > 0) out.push_back(static_cast<uint8_t>(acc));
	return out;
}

}  // namespace ckl

using namespace ckl;

extern "C" {

const char* ckl_last_error(void) { return g_last_error.c_str(); }

int ckl_abi_version(void) { return 1; }

int ckl_device_count(void) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return count;
}

int ckl_header_info_from_bytes(const uint8_t* buf, uint64_t n, ckl_header_info* out) {
	try {
		if (!buf || !out) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		Header h = Header::parse(buf, n);
		out->format_version = h.format_version;
		out->label_format = static_cast<uint32_t>(h.label_format);
		out->crack_format = static_cast<uint32_t>(h.crack_format);
		out->is_signed = h.is_signed;
		out->data_width = static_cast<uint32_t>(h.data_width);
		out->stored_data_width = static_cast<uint32_t>(h.stored_data_width);
		out->sx = h.sx; out->sy = h.sy; out->sz = h.sz;
		out->fortran_order = h.fortran_order;
		out->markov_model_order = static_cast<uint32_t>(h.markov_model_order);
		out->is_sorted = h.is_sorted;
		out->num_label_bytes = h.num_label_bytes;
		out->header_bytes = h.header_bytes();
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

void ckl_free(void* p) { host_out_free(p); }

int ckl_host_register(void* p, uint64_t bytes) {
	if (!p || !bytes) { set_last_error("crackle_amd: null argument"); return CKL_ERR_ARG; }
	if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
		(void)hipGetLastError();
		set_last_error("crackle_amd: hipHostRegister failed");
		return CKL_ERR_RUNTIME;
	}
	return CKL_OK;
}
int ckl_host_unregister(void* p) {
	if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
	return CKL_OK;
}

uint32_t ckl_crc32c(const uint8_t* data, uint64_t n) { return crc32c(data, n); }

// crc32c(A || B) from crc32c(A), crc32c(B) and len(B): appending len(B) bytes multiplies the
// register by x^(8 len(B)); the init / final inversions of the two finished values cancel
uint32_t ckl_crc32c_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
	return gf_mul(crc_a, gf_xpow(8ull * len_b)) ^ crc_b;
}

}  // extern "C"
