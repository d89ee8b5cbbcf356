// TEST INFRASTRUCTURE: the host-only stages of libcrackle_amd (ckl_zstack, ckl_zsplit,
// ckl_pin_labels_host, header parsing) and the oracle's C restatement, built with
// AddressSanitizer + UndefinedBehaviorSanitizer and driven over valid, ragged and hostile inputs.
// Built and run by tests/test_sanitizers_cpu.py (CPU only; GPU sanitizers are not available).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/crackle_amd.h"
#include "../../oracle/ckl_oracle.h"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x45D9F3Bu; x ^= x >> 16; x *= 0x45D9F3Bu; x ^= x >> 16; return x; }

// blobs: label of the nearest of a few hashed seeds, x fastest
template <typename T>
static std::vector<T> blobs(int sx, int sy, int sz, uint32_t seed, uint32_t modulus) {
	const int ns = 12;
	int px[ns], py[ns], pz[ns]; T lab[ns];
	for (int i = 0; i < ns; i++) {
		px[i] = mix(seed * 31 + i * 3) % sx; py[i] = mix(seed * 31 + i * 3 + 1) % sy; pz[i] = mix(seed * 31 + i * 3 + 2) % sz;
		lab[i] = static_cast<T>(1 + mix(seed * 77 + i) % modulus);
	}
	std::vector<T> v(static_cast<size_t>(sx) * sy * sz);
	for (int z = 0; z < sz; z++) for (int y = 0; y < sy; y++) for (int x = 0; x < sx; x++) {
		long best = 1L << 60; T l = 0;
		for (int i = 0; i < ns; i++) {
			const long d = long(x - px[i]) * (x - px[i]) + long(y - py[i]) * (y - py[i]) + 16L * (z - pz[i]) * (z - pz[i]);
			if (d < best) { best = d; l = lab[i]; }
		}
		v[x + static_cast<size_t>(sx) * (y + static_cast<size_t>(sy) * z)] = l;
	}
	return v;
}

struct Stream { std::vector<uint8_t> b; };

template <typename T>
static Stream oracle_compress(const std::vector<T>& v, int sx, int sy, int sz, int pins, int order) {
	unsigned char* out = nullptr; uint64_t n = 0;
	const int rc = ckl_oracle_compress(v.data(), sizeof(T), 0, sx, sy, sz, pins, 1, order, 0, 1, 0, 2, &out, &n);
	CHECK(rc == 0);
	Stream s; if (rc == 0) { s.b.assign(out, out + n); ckl_oracle_free(out); }
	return s;
}

static uint8_t crc8(const uint8_t* p, size_t n) {
	uint8_t c = 0xFF;
	for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xE7 : (c >> 1); }
	return c;
}

template <typename T>
static void round(int sx, int sy, int sz, uint32_t seed, uint32_t modulus, int order) {
	const std::vector<T> vol = blobs<T>(sx, sy, sz, seed, modulus);
	const Stream whole = oracle_compress(vol, sx, sy, sz, 0, order);
	// decode with the restatement
	std::vector<T> back(vol.size());
	CHECK(ckl_oracle_decompress(whole.b.data(), whole.b.size(), back.data(), 0, -1, 2, 0, 0) == 0);
	CHECK(back == vol);
	// zsplit every slice, zstack them again: the whole stream, byte for byte
	std::vector<std::vector<uint8_t>> parts;
	for (int z = 0; z < sz; z++) {
		uint8_t* o = nullptr; uint64_t n = 0;
		CHECK(ckl_zsplit(whole.b.data(), whole.b.size(), z, z + 1, &o, &n) == CKL_OK);
		if (o) { parts.emplace_back(o, o + n); ckl_free(o); }
	}
	if (order == 0 && parts.size() == static_cast<size_t>(sz)) {
		std::vector<const uint8_t*> ptr; std::vector<uint64_t> len;
		for (auto& p : parts) { ptr.push_back(p.data()); len.push_back(p.size()); }
		uint8_t* o = nullptr; uint64_t n = 0;
		CHECK(ckl_zstack(ptr.data(), len.data(), ptr.size(), &o, &n) == CKL_OK);
		CHECK(o && n == whole.b.size() && memcmp(o, whole.b.data(), n) == 0);
		if (o) ckl_free(o);
	}
	// pin section: the product's host stage against the restatement's whole pin stream
	if (sz > 1) {
		const Stream pinned = oracle_compress(vol, sx, sy, sz, 1, 0);
		std::vector<uint32_t> cc(vol.size()); std::vector<uint64_t> per(sz); uint64_t N = 0;
		CHECK(ckl_oracle_connected_components(vol.data(), sizeof(T), sx, sy, sz, cc.data(), per.data(), &N) == 0);
		std::vector<uint32_t> nc(per.begin(), per.end());
		ckl_header_info hi;
		CHECK(ckl_header_info_from_bytes(pinned.b.data(), pinned.b.size(), &hi) == CKL_OK);
		if (hi.label_format == 2) {      // the volume qualified for pins (IMPERMISSIBLE)
			uint8_t* o = nullptr; uint64_t n = 0;
			CHECK(ckl_pin_labels_host(vol.data(), sizeof(T), cc.data(), sx, sy, sz, nc.data(), static_cast<int>(hi.stored_data_width), 1, 0, &o, &n) == CKL_OK);
			const uint64_t off = 29 + 4ull * (sz + 1);
			CHECK(o && n == hi.num_label_bytes && memcmp(o, pinned.b.data() + off, n) == 0);
			if (o) ckl_free(o);
		}
	}
	// hostile: truncated streams and forged section lengths must be refused, not read out of bounds
	for (size_t cut : { size_t(10), size_t(29), whole.b.size() / 2, whole.b.size() - 1 }) {
		std::vector<uint8_t> t(whole.b.begin(), whole.b.begin() + cut);      // exact-size heap block: any overread is seen
		uint8_t* o = nullptr; uint64_t n = 0;
		CHECK(ckl_zsplit(t.data(), t.size(), 0, 1, &o, &n) != CKL_OK);
		const uint8_t* one[1] = { t.data() }; const uint64_t l1[1] = { t.size() };
		CHECK(ckl_zstack(one, l1, 1, &o, &n) != CKL_OK);
		std::vector<T> sink(vol.size());
		CHECK(ckl_oracle_decompress(t.data(), t.size(), sink.data(), 0, -1, 1, 0, 0) != 0);
	}
	for (uint64_t nlb : { ~0ull, ~0ull - 28, 1ull << 63, 1ull << 40, 3ull }) {
		std::vector<uint8_t> f(whole.b);
		memcpy(f.data() + 20, &nlb, 8);
		f[28] = crc8(f.data() + 5, 23);
		uint8_t* o = nullptr; uint64_t n = 0;
		CHECK(ckl_zsplit(f.data(), f.size(), 0, 1, &o, &n) != CKL_OK);
		const uint8_t* one[1] = { f.data() }; const uint64_t l1[1] = { f.size() };
		CHECK(ckl_zstack(one, l1, 1, &o, &n) != CKL_OK);
	}
}

int main() {
	round<uint8_t>(40, 33, 6, 1, 200, 0);
	round<uint16_t>(64, 17, 3, 2, 5000, 0);
	round<uint32_t>(33, 65, 4, 3, 1u << 20, 0);
	round<uint64_t>(19, 23, 5, 4, 1u << 30, 0);
	round<uint8_t>(48, 48, 5, 5, 100, 3);
	round<uint32_t>(1, 1, 1, 6, 10, 0);
	round<uint16_t>(5, 1, 7, 7, 10, 2);
	printf("host_sanitize: %d failures\n", fails);
	return fails ? 1 : 0;
}
