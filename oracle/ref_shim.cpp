// TEST INFRASTRUCTURE ONLY — not part of the shipped product.
//
// Thin extern "C" wrapper around the reference implementation's header-only
// C++ (seung-lab/crackle, src/crackle.hpp).  It is compiled *in place* from
// /root/reference/src by oracle/Makefile (target `ref`), output goes to
// oracle/_ref/libcrackle_ref.so (git-ignored).  No reference source is copied
// into this repository: this file only #includes the headers where they lie.
//
// Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// (kind = "reference") as the strongest parity checker.
//
// Reference entry points wrapped:
//   crackle::compress<LABEL>      src/crackle.hpp:220-257
//   crackle::decompress<LABEL,OUT> src/crackle.hpp:503-663
//   crackle::cc3d::connected_components src/cc3d.hpp:371-400
//   crackle::crack_code_to_vcg    src/crackle.hpp:414-425
//   crackle::crc::crc32c          src/crc.hpp:51-57
//   crackle::reencode_with_markov_order src/crackle.hpp:858-984
//   crackle::operations::voxel_connectivity_graph src/operations.hpp:667-826
//   crackle::operations::array_equal src/operations.hpp:1039-1184
//   crackle::operations::mode_pooling_2x2x1 src/operations.hpp:1201-1340
//   crackle::operations::point_cloud src/operations.hpp:183-262 (dual_graph.hpp:133-275)
//   crackle::operations::voxel_counts / centroids / bounding_boxes src/operations.hpp:321-665

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <stdexcept>
#include <cmath>
#include <limits>
#include <algorithm>

#include "crackle.hpp"
#include "operations.hpp"

static thread_local std::string g_err;

template <typename LABEL>
static int compress_t(
	const void* labels, int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	size_t parallel, unsigned char** out, uint64_t* out_len
) {
	std::vector<unsigned char> buf = crackle::compress<LABEL>(
		reinterpret_cast<const LABEL*>(labels), sx, sy, sz,
		allow_pins != 0, fortran_order != 0, markov_order,
		optimize_pins != 0, auto_bgcolor != 0, manual_bgcolor, parallel
	);
	*out = static_cast<unsigned char*>(malloc(buf.size() ? buf.size() : 1));
	memcpy(*out, buf.data(), buf.size());
	*out_len = buf.size();
	return 0;
}

extern "C" {

__attribute__((visibility("default")))
const char* ckl_ref_last_error() { return g_err.c_str(); }

__attribute__((visibility("default")))
void ckl_ref_free(void* p) { free(p); }

__attribute__((visibility("default")))
int ckl_ref_compress(
	const void* labels, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel, unsigned char** out, uint64_t* out_len
) {
	try {
#define CALL(T) return compress_t<T>(labels, sx, sy, sz, allow_pins, fortran_order, \
	markov_order, optimize_pins, auto_bgcolor, manual_bgcolor, parallel, out, out_len)
		if (is_signed) {
			if (dtype_bytes == 1) CALL(int8_t);
			if (dtype_bytes == 2) CALL(int16_t);
			if (dtype_bytes == 4) CALL(int32_t);
			CALL(int64_t);
		}
		if (dtype_bytes == 1) CALL(uint8_t);
		if (dtype_bytes == 2) CALL(uint16_t);
		if (dtype_bytes == 4) CALL(uint32_t);
		CALL(uint64_t);
#undef CALL
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// out must hold sx*sy*(z_end-z_start) elements of data_width bytes
// (or 1 byte each when has_label).
__attribute__((visibility("default")))
int ckl_ref_decompress(
	const unsigned char* buf, uint64_t n, void* out,
	int64_t z_start, int64_t z_end, uint64_t parallel,
	int has_label, uint64_t label
) {
	try {
		crackle::CrackleHeader head(buf);
		std::optional<uint64_t> lbl = std::nullopt;
		if (has_label) lbl = label;
#define CALL(T) \
	if (has_label) { crackle::decompress<T, uint8_t>(buf, n, reinterpret_cast<uint8_t*>(out), z_start, z_end, parallel, lbl); } \
	else { crackle::decompress<T, T>(buf, n, reinterpret_cast<T*>(out), z_start, z_end, parallel, lbl); } \
	return 0
		if (head.data_width == 1) { CALL(uint8_t); }
		if (head.data_width == 2) { CALL(uint16_t); }
		if (head.data_width == 4) { CALL(uint32_t); }
		CALL(uint64_t);
#undef CALL
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// per-slice 4-connected CCL with globally increasing ids (pins path numbering)
__attribute__((visibility("default")))
int ckl_ref_connected_components(
	const void* labels, int dtype_bytes,
	int64_t sx, int64_t sy, int64_t sz,
	uint32_t* cc_out, uint64_t* per_slice, uint64_t* N
) {
	try {
		std::vector<uint64_t> ncs(sz);
		uint64_t n = 0;
		if (dtype_bytes == 1) crackle::cc3d::connected_components<uint8_t, uint32_t>(reinterpret_cast<const uint8_t*>(labels), sx, sy, sz, ncs, cc_out, n);
		else if (dtype_bytes == 2) crackle::cc3d::connected_components<uint16_t, uint32_t>(reinterpret_cast<const uint16_t*>(labels), sx, sy, sz, ncs, cc_out, n);
		else if (dtype_bytes == 4) crackle::cc3d::connected_components<uint32_t, uint32_t>(reinterpret_cast<const uint32_t*>(labels), sx, sy, sz, ncs, cc_out, n);
		else crackle::cc3d::connected_components<uint64_t, uint32_t>(reinterpret_cast<const uint64_t*>(labels), sx, sy, sz, ncs, cc_out, n);
		for (int64_t z = 0; z < sz; z++) per_slice[z] = ncs[z];
		*N = n;
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// voxel connectivity graph (4 bits / pixel) of one slice of a stream
__attribute__((visibility("default")))
int ckl_ref_slice_vcg(
	const unsigned char* buf, uint64_t n, int64_t z, uint8_t* vcg_out
) {
	try {
		crackle::CrackleHeader head(buf);
		std::span<const unsigned char> binary(buf, n);
		auto model = crackle::decode_markov_model(head, binary);
		auto codes = crackle::get_crack_codes(head, binary, z, z + 1);
		crackle::crack_code_to_vcg(
			codes[0], head.sx, head.sy,
			head.crack_format == crackle::CrackFormat::PERMISSIBLE,
			model, vcg_out
		);
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::operations::voxel_connectivity_graph  src/operations.hpp:667-826
// vcg_out: sx*sy*sz bytes (whole volume, x fastest)
__attribute__((visibility("default")))
int ckl_ref_voxel_connectivity_graph(
	const unsigned char* buf, uint64_t n, int connectivity, uint64_t parallel, uint8_t* vcg_out
) {
	try {
		crackle::CrackleHeader head(buf);
		uint8_t* vcg = crackle::operations::voxel_connectivity_graph(buf, n, 0, -1, parallel, connectivity);
		memcpy(vcg_out, vcg, static_cast<size_t>(head.sx) * head.sy * head.sz);
		delete[] vcg;
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::reencode_with_markov_order  src/crackle.hpp:858-984
__attribute__((visibility("default")))
int ckl_ref_reencode(
	const unsigned char* buf, uint64_t n, int markov_order, uint64_t parallel,
	unsigned char** out, uint64_t* out_len
) {
	try {
		std::vector<unsigned char> r = crackle::reencode_with_markov_order(buf, n, markov_order, parallel);
		*out = static_cast<unsigned char*>(malloc(r.size() ? r.size() : 1));
		memcpy(*out, r.data(), r.size());
		*out_len = r.size();
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::operations::array_equal  src/operations.hpp:1039-1184
__attribute__((visibility("default")))
int ckl_ref_array_equal(const unsigned char* buf1, uint64_t n1, const unsigned char* buf2, uint64_t n2, uint64_t parallel, int* equal) {
	try {
		*equal = crackle::operations::array_equal(buf1, n1, buf2, n2, parallel) ? 1 : 0;
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::operations::mode_pooling_2x2x1  src/operations.hpp:1201-1340
// *out: the per-slice streams one after the other, lens_out[i] their lengths (room for sz entries)
__attribute__((visibility("default")))
int ckl_ref_mode_pooling(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end, uint64_t parallel,
	unsigned char** out, uint64_t* out_len, uint64_t* lens_out, uint64_t* count
) {
	try {
		auto bins = crackle::operations::mode_pooling_2x2x1(buf, n, z_start, z_end, parallel);
		uint64_t total = 0;
		for (const auto& b : bins) total += b.size();
		*out = static_cast<unsigned char*>(malloc(total ? total : 1));
		uint64_t at = 0;
		for (size_t i = 0; i < bins.size(); i++) {
			memcpy(*out + at, bins[i].data(), bins[i].size());
			at += bins[i].size();
			lens_out[i] = bins[i].size();
		}
		*out_len = total;
		*count = bins.size();
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::operations::point_cloud  src/operations.hpp:183-262 (the binding, src/fastcrackle.cpp:315-345,
// passes z_start, z_end, labels, skip_background, parallel).  Labels ascending, offsets in points.
__attribute__((visibility("default")))
int ckl_ref_point_cloud(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end,
	const uint64_t* labels, uint64_t n_labels, int has_labels, int skip_background,
	uint64_t** labels_out, uint64_t** offsets_out, uint16_t** points_out, uint64_t* n_out
) {
	try {
		std::optional<std::vector<uint64_t>> sel = std::nullopt;
		if (has_labels) sel = std::vector<uint64_t>(labels, labels + n_labels);
		auto ptc = crackle::operations::point_cloud(buf, n, z_start, z_end, sel, skip_background != 0, 1);
		std::vector<uint64_t> keys;
		for (const auto& kv : ptc) keys.push_back(kv.first);
		std::sort(keys.begin(), keys.end());
		uint64_t total = 0;
		for (uint64_t k : keys) total += ptc[k].size();
		*labels_out = static_cast<uint64_t*>(malloc((keys.size() + 1) * 8));
		*offsets_out = static_cast<uint64_t*>(malloc((keys.size() + 2) * 8));
		*points_out = static_cast<uint16_t*>(malloc((total + 1) * 2));
		uint64_t at = 0;
		for (size_t i = 0; i < keys.size(); i++) {
			const auto& v = ptc[keys[i]];
			(*labels_out)[i] = keys[i];
			(*offsets_out)[i] = at / 3;
			memcpy(*points_out + at, v.data(), v.size() * 2);
			at += v.size();
		}
		(*offsets_out)[keys.size()] = at / 3;
		*n_out = keys.size();
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

// crackle::operations::voxel_counts / centroids / bounding_boxes  src/operations.hpp:321-665
// (bound at src/fastcrackle.cpp:346-420).  The maps come back as arrays sorted by label:
// which: 0 = counts (values: 1 x uint64 per label), 1 = centroids (3 x float64), 2 = boxes (6 x uint32).
__attribute__((visibility("default")))
int ckl_ref_label_stats(
	const unsigned char* buf, uint64_t n, int which, int64_t z_start, int64_t z_end, uint64_t parallel,
	uint64_t** labels_out, void** values_out, uint64_t* n_out
) {
	try {
		std::vector<uint64_t> keys;
		if (which == 0) {
			auto m = crackle::operations::voxel_counts(buf, n, z_start, z_end, parallel);
			for (const auto& kv : m) keys.push_back(kv.first);
			std::sort(keys.begin(), keys.end());
			uint64_t* v = static_cast<uint64_t*>(malloc((keys.size() + 1) * 8));
			for (size_t i = 0; i < keys.size(); i++) v[i] = m[keys[i]];
			*values_out = v;
		}
		else if (which == 1) {
			auto m = crackle::operations::centroids(buf, n, z_start, z_end, parallel);
			for (const auto& kv : m) keys.push_back(kv.first);
			std::sort(keys.begin(), keys.end());
			double* v = static_cast<double*>(malloc((keys.size() + 1) * 3 * 8));
			for (size_t i = 0; i < keys.size(); i++) for (int k = 0; k < 3; k++) v[3 * i + k] = m[keys[i]][k];
			*values_out = v;
		}
		else {
			auto m = crackle::operations::bounding_boxes(buf, n, z_start, z_end, parallel);
			for (const auto& kv : m) keys.push_back(kv.first);
			std::sort(keys.begin(), keys.end());
			uint32_t* v = static_cast<uint32_t*>(malloc((keys.size() + 1) * 6 * 4));
			for (size_t i = 0; i < keys.size(); i++) for (int k = 0; k < 6; k++) v[6 * i + k] = m[keys[i]][k];
			*values_out = v;
		}
		*labels_out = static_cast<uint64_t*>(malloc((keys.size() + 1) * 8));
		for (size_t i = 0; i < keys.size(); i++) (*labels_out)[i] = keys[i];
		*n_out = keys.size();
		return 0;
	}
	catch (const std::exception& e) {
		g_err = e.what();
		return 1;
	}
}

__attribute__((visibility("default")))
uint32_t ckl_ref_crc32c(const uint8_t* data, uint64_t n) {
	return crackle::crc::crc32c(data, n);
}

}
