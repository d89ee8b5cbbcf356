// `fastcrackle` for MI355X: the reference's pybind11 module (src/fastcrackle.cpp:641-669), same
// function names and positional arguments, bodies on top of the C-ABI of libcrackle_amd.so
// (include/crackle_amd.h).  crackle/codec.py:670,729 (`fastcrackle.decompress(binary, z_start,
// z_end, parallel, label)`, `fastcrackle.compress(labels, allow_pins, fortran_order,
// markov_model_order, optimize_pins, auto_bgcolor, manual_bgcolor, parallel)`) run unchanged on it.
// `parallel` is accepted and ignored: the device decides the parallelism.  The device is
// CRACKLE_AMD_DEVICE (default 0).  No CPU fallback: without a HIP device every call raises.
//
// Functions: decompress (ref :84-128), compress (:163-210), reencode_markov (:212-227),
// voxel_counts / centroids / bounding_boxes (:346-426), voxel_connectivity_graph (:538-565),
// array_equal (:594-618), mode_pooling_2x2x1 (:620-639), point_cloud (:315-345).
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/crackle_amd.h"

namespace py = pybind11;

namespace {

int device() {
	const char* env = std::getenv("CRACKLE_AMD_DEVICE");
	return env ? std::atoi(env) : 0;
}

// status -> exception, like the reference's std::runtime_error("crackle: ...") surfacing as RuntimeError
void check(int rc) {
	if (rc == CKL_OK) return;
	const std::string msg = ckl_last_error();
	if (rc == CKL_ERR_ARG) throw py::value_error(msg);
	throw std::runtime_error(msg);
}

struct Stream {
	const uint8_t* p;
	uint64_t n;
	ckl_header_info head;
	explicit Stream(const py::buffer& buffer) {
		info = buffer.request();
		if (info.ndim != 1) throw std::runtime_error("Expected a 1D buffer");
		p = static_cast<const uint8_t*>(info.ptr);
		n = static_cast<uint64_t>(info.size) * static_cast<uint64_t>(info.itemsize);
		check(ckl_header_info_from_bytes(p, n, &head));
	}
	py::buffer_info info;
};

py::dtype unsigned_dtype(int width) {
	return py::dtype(width == 1 ? "u1" : width == 2 ? "u2" : width == 4 ? "u4" : "u8");
}

// the z-range the reference's decompress_helper decodes (src/fastcrackle.cpp:49-58)
void clamp_range(const ckl_header_info& h, int64_t& z_start, int64_t& z_end) {
	z_start = std::max<int64_t>(z_start, 0);
	if (z_end == -1) z_end = h.sz;
	z_end = std::min<int64_t>(std::max<int64_t>(z_end, 0), h.sz);
}

py::array decompress(const py::buffer buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/, std::optional<uint64_t> label) {
	Stream s(buffer);
	int64_t zs = z_start, ze = z_end;
	clamp_range(s.head, zs, ze);
	const uint64_t voxels = static_cast<uint64_t>(s.head.sx) * s.head.sy * static_cast<uint64_t>(std::max<int64_t>(ze - zs, 0));
	const int width = label.has_value() ? 1 : static_cast<int>(s.head.data_width);
	const std::vector<py::ssize_t> shape = { static_cast<py::ssize_t>(voxels) };
	py::array arr(unsigned_dtype(width), shape);
	if (voxels == 0) return arr;
	void* dst = arr.mutable_data();
	int rc;
	{
		py::gil_scoped_release nogil;
		rc = ckl_decompress(s.p, s.n, dst, voxels * width, CKL_MEM_HOST, zs, ze, label.has_value() ? 1 : 0, label.value_or(0), device());
	}
	check(rc);
	return arr;
}

py::bytes compress(
	const py::array& labels, bool allow_pins, bool fortran_order, uint64_t markov_model_order,
	bool optimize_pins, bool auto_bgcolor, int64_t manual_bgcolor, size_t /*parallel*/
) {
	// missing dimensions count as 1 (src/fastcrackle.cpp:141-147)
	const int64_t sx = labels.ndim() < 1 ? 1 : labels.shape(0);
	const int64_t sy = labels.ndim() < 2 ? 1 : labels.shape(1);
	const int64_t sz = labels.ndim() < 3 ? 1 : labels.shape(2);
	uint8_t* out = nullptr;
	uint64_t n = 0;
	const void* data = labels.data();
	const int width = static_cast<int>(labels.dtype().itemsize());
	const int is_signed = labels.dtype().kind() == 'i' ? 1 : 0;
	int rc;
	{
		py::gil_scoped_release nogil;      // nothing below touches a Python object
		rc = ckl_compress(data, CKL_MEM_HOST, width, is_signed,
			sx, sy, sz, allow_pins ? 1 : 0, fortran_order ? 1 : 0, markov_model_order, optimize_pins ? 1 : 0, auto_bgcolor ? 1 : 0, manual_bgcolor,
			device(), &out, &n);
	}
	check(rc);
	py::bytes b(reinterpret_cast<const char*>(out), n);
	ckl_free(out);
	return b;
}

// every C-ABI call below runs whole-volume GPU work and stream syncs: none of them touches a Python
// object, so the GIL is released around each (buffers and output arrays are obtained before)
template <typename F>
int nogil(F&& call) {
	py::gil_scoped_release release;
	return call();
}

py::bytes reencode_markov(const py::buffer buffer, int markov_model_order, size_t /*parallel*/) {
	Stream s(buffer);
	uint8_t* out = nullptr;
	uint64_t n = 0;
	check(nogil([&] { return ckl_reencode_markov(s.p, s.n, markov_model_order, device(), &out, &n); }));
	py::bytes b(reinterpret_cast<const char*>(out), n);
	ckl_free(out);
	return b;
}

struct LabelStats {
	std::vector<uint64_t> labels, counts, sums;
	std::vector<uint32_t> boxes;
	uint64_t mask;
};

// one pass over the decoded runs on the device: per label voxel count, coordinate sums, box
LabelStats label_stats(const py::buffer& buffer, int64_t z_start, int64_t z_end) {
	Stream s(buffer);
	LabelStats st;
	st.mask = s.head.data_width >= 8 ? ~0ull : ((1ull << (8 * s.head.data_width)) - 1ull);
	{
		// operations::get_szr (src/operations.hpp:54-72) comes first in all three: an empty range raises,
		// with header.sz - 1 taken in 32-bit unsigned arithmetic as there
		int64_t zs = std::max<int64_t>(std::min<int64_t>(z_start, static_cast<int64_t>(static_cast<uint32_t>(s.head.sz - 1u))), 0);
		int64_t ze = z_end < 0 ? static_cast<int64_t>(s.head.sz) : z_end;
		ze = std::max<int64_t>(std::min<int64_t>(ze, static_cast<int64_t>(s.head.sz)), 0);
		if (zs >= ze) throw std::runtime_error("crackle: Invalid range: " + std::to_string(zs) + " - " + std::to_string(ze));
	}
	if (static_cast<uint64_t>(s.head.sx) * s.head.sy * s.head.sz == 0) return st;
	ckl_decoder* d = nullptr;
	check(nogil([&] { return ckl_decoder_create(s.p, s.n, z_start, z_end, device(), &d); }));
	uint64_t n = 0;
	int rc = nogil([&] { return ckl_decoder_label_stats(d, 0, nullptr, nullptr, nullptr, nullptr, &n); });      // size query: reports the table size
	if (rc == CKL_OK || n) {
		st.labels.resize(n); st.counts.resize(n); st.sums.resize(3 * n); st.boxes.resize(6 * n);
		rc = n ? nogil([&] { return ckl_decoder_label_stats(d, n, st.labels.data(), st.counts.data(), st.sums.data(), st.boxes.data(), &n); }) : CKL_OK;
	}
	ckl_decoder_destroy(d);
	check(rc);
	return st;
}

py::dict voxel_counts(const py::buffer& buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/) {
	const LabelStats st = label_stats(buffer, z_start, z_end);
	py::dict out;
	for (size_t i = 0; i < st.labels.size(); i++) if (st.counts[i]) out[py::int_(st.labels[i] & st.mask)] = py::int_(st.counts[i]);
	return out;
}

py::dict centroids(const py::buffer& buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/) {
	const LabelStats st = label_stats(buffer, z_start, z_end);
	py::dict out;
	for (size_t i = 0; i < st.labels.size(); i++) {
		if (!st.counts[i]) continue;
		py::array_t<double> xyz(3);
		auto v = xyz.mutable_unchecked<1>();
		for (int k = 0; k < 3; k++) v(k) = static_cast<double>(st.sums[3 * i + k]) / static_cast<double>(st.counts[i]);
		out[py::int_(st.labels[i] & st.mask)] = xyz;
	}
	return out;
}

py::dict bounding_boxes(const py::buffer& buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/) {
	const LabelStats st = label_stats(buffer, z_start, z_end);
	py::dict out;
	for (size_t i = 0; i < st.labels.size(); i++) {
		// the reference's map has every label of the unique list (absent ones with their initial box,
		// operations.hpp:561-567); a label outside that list that is absent is reported as a zero box
		if (!st.counts[i] && !st.boxes[6 * i]) continue;
		py::array_t<uint32_t> box(6);
		auto v = box.mutable_unchecked<1>();
		for (int k = 0; k < 6; k++) v(k) = st.boxes[6 * i + k];
		out[py::int_(st.labels[i] & st.mask)] = box;
	}
	return out;
}

py::array voxel_connectivity_graph(const py::buffer buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/, int connectivity) {
	Stream s(buffer);
	int64_t zs = z_start, ze = z_end;
	clamp_range(s.head, zs, ze);
	const uint64_t sx = s.head.sx, sy = s.head.sy, nz = static_cast<uint64_t>(std::max<int64_t>(ze - zs, 0));
	// x fastest, like the reference's to_numpy (src/fastcrackle.cpp:19-38)
	py::array_t<uint8_t> arr({ sx, sy, nz }, { static_cast<uint64_t>(1), sx, sx * sy });
	if (sx * sy * nz == 0) return arr;
	uint8_t* dst = arr.mutable_data();
	check(nogil([&] { return ckl_voxel_connectivity_graph_range(s.p, s.n, zs, ze, connectivity, device(), dst, sx * sy * nz); }));
	return arr;
}

bool array_equal(const py::buffer buffer1, const py::buffer buffer2, size_t /*parallel*/) {
	Stream a(buffer1), b(buffer2);
	if (a.p == b.p) return true;      // src/fastcrackle.cpp:609-611
	int eq = 0;
	check(nogil([&] { return ckl_array_equal(a.p, a.n, b.p, b.n, device(), &eq); }));
	return eq != 0;
}

py::list mode_pooling_2x2x1(const py::buffer buffer, int64_t z_start, int64_t z_end, size_t /*parallel*/) {
	Stream s(buffer);
	uint8_t* out = nullptr;
	uint64_t n = 0, count = 0;
	uint64_t* lens = nullptr;
	check(nogil([&] { return ckl_mode_pooling_2x2x1(s.p, s.n, z_start, z_end, device(), &out, &n, &lens, &count); }));
	py::list result;
	uint64_t at = 0;
	for (uint64_t i = 0; i < count; i++) {
		result.append(py::bytes(reinterpret_cast<const char*>(out + at), lens[i]));
		at += lens[i];
	}
	if (out) ckl_free(out);
	if (lens) ckl_free(lens);
	return result;
}

// src/fastcrackle.cpp:315-345: dict label -> flat uint16 array of (x, y, z) triples
py::dict point_cloud(const py::buffer buffer, int64_t z_start, int64_t z_end, const std::optional<std::vector<uint64_t>> labels, bool skip_background, size_t /*parallel*/) {
	Stream s(buffer);
	uint64_t* lab = nullptr; uint64_t* off = nullptr; uint16_t* pts = nullptr;
	uint64_t n = 0;
	check(nogil([&] { return ckl_point_cloud(s.p, s.n, z_start, z_end, labels ? labels->data() : nullptr, labels ? labels->size() : 0, labels ? 1 : 0,
		skip_background ? 1 : 0, device(), &lab, &off, &pts, &n); }));
	py::dict result;
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t count = 3 * (off[i + 1] - off[i]);
		py::array_t<uint16_t> arr(static_cast<py::ssize_t>(count));
		if (count) std::memcpy(arr.mutable_data(), pts + 3 * off[i], count * sizeof(uint16_t));
		result[py::int_(lab[i])] = arr;
	}
	if (lab) ckl_free(lab);
	if (off) ckl_free(off);
	if (pts) ckl_free(pts);
	return result;
}

}  // namespace

PYBIND11_MODULE(fastcrackle, m) {
	m.doc() = "Accelerated crackle functions (MI355X build on libcrackle_amd.so).";
	m.def("decompress", &decompress, "Decompress a crackle file into a numpy array.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1, py::arg("label") = py::none());
	m.def("compress", &compress, "Compress a numpy array into a binary crackle file returned as bytes.",
		py::arg("labels"), py::arg("allow_pins") = false, py::arg("fortran_order") = true, py::arg("markov_model_order") = 0,
		py::arg("optimize_pins") = false, py::arg("auto_bgcolor") = true, py::arg("manual_bgcolor") = 0, py::arg("parallel") = 1);
	m.def("reencode_markov", &reencode_markov, "Change the markov order of an existing crackle binary.",
		py::arg("buffer"), py::arg("markov_model_order"), py::arg("parallel") = 1);
	m.def("voxel_counts", &voxel_counts, "Compute the voxel counts for each label in the dataset.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1);
	m.def("centroids", &centroids, "Compute the centroid for each label in the dataset.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1);
	m.def("bounding_boxes", &bounding_boxes, "Compute the bounding box for each label in the dataset.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1);
	m.def("array_equal", &array_equal, "Check if two crackle arrays are equal regardless of encoding.",
		py::arg("buffer1"), py::arg("buffer2"), py::arg("parallel") = 1);
	m.def("mode_pooling_2x2x1", &mode_pooling_2x2x1, "Return an array of downsampled crackle binaries in z order.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1);
	m.def("point_cloud", &point_cloud, "Extract one or more point clouds without decompressing.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("labels") = py::none(), py::arg("skip_background") = false, py::arg("parallel") = 1);
	m.def("voxel_connectivity_graph", &voxel_connectivity_graph, "Extract the voxel connectivity graph from the image.",
		py::arg("buffer"), py::arg("z_start") = 0, py::arg("z_end") = -1, py::arg("parallel") = 1, py::arg("connectivity") = 4);
}
