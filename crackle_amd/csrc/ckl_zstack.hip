// Host-only stream surgery: concatenate FLAT-label streams of consecutive z-slabs
// into the stream of the whole volume.  Native form of crackle.operations.zstack
// (crackle/operations.py:424-548) for the case the sharded encoder needs: all slabs
// were encoded with the same crack format, stored width and markov model (the
// sharded encoder imposes them), so crack codes, per-slice CRCs
// and component counts are concatenated verbatim and only the key table is re-keyed
// against the merged, sorted unique-label list (labels.hpp:92-152).
#include "ckl_common.hpp"

#include <algorithm>

using namespace ckl;

namespace {

struct Slab {
	Header h;
	const uint8_t* buf;
	uint64_t n;
	const uint8_t* labels;      // label section
	uint64_t num_unique;
	std::vector<uint64_t> uniq;
	const uint8_t* comp;        // cc_per_slice table
	const uint8_t* keys;
	uint64_t total_comp;
	const uint8_t* model;
	const uint8_t* cracks;
	uint64_t crack_bytes;
	const uint8_t* z_index;
	const uint8_t* crcs;        // per-slice crcs
};

Slab parse_slab(const uint8_t* buf, uint64_t n) {
	Slab s;
	s.h = Header::parse(buf, n);
	s.buf = buf; s.n = n;
	const Header& h = s.h;
	if (h.format_version != 1) throw Error(CKL_ERR_ARG, "crackle_amd: zstack needs version 1 streams");
	if (h.label_format != FLAT) throw Error(CKL_ERR_ARG, "crackle_amd: zstack is implemented for FLAT label streams only");
	if (h.voxels() == 0) throw Error(CKL_ERR_ARG, "crackle_amd: zstack of an empty slab");
	const uint64_t hb = h.header_bytes(), gib = h.grid_index_bytes();
	const uint64_t tail = 4ull * (static_cast<uint64_t>(h.sz) + 1);
	if (!h.layout_fits(n)) throw Error(CKL_ERR_RUNTIME, "crackle: Unable to read past end of buffer.");      // no sum of untrusted fields that could wrap
	s.z_index = buf + hb;
	s.labels = buf + hb + gib;
	const int sw = h.stored_data_width;
	const int cw = byte_width(static_cast<uint64_t>(h.sx) * h.sy);
	if (h.num_label_bytes < 8) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	s.num_unique = rd_le(s.labels, 8);
	if (static_cast<uint64_t>(cw) * h.sz > h.num_label_bytes - 8 || s.num_unique > (h.num_label_bytes - 8 - static_cast<uint64_t>(cw) * h.sz) / sw) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	s.uniq.resize(s.num_unique);
	for (uint64_t i = 0; i < s.num_unique; i++) s.uniq[i] = rd_le(s.labels + 8 + i * sw, sw);
	s.comp = s.labels + 8 + s.num_unique * sw;
	s.total_comp = 0;
	for (uint64_t z = 0; z < h.sz; z++) s.total_comp += rd_le(s.comp + z * cw, cw);
	s.keys = s.comp + static_cast<uint64_t>(cw) * h.sz;
	const int kw = byte_width(s.num_unique);
	if (s.total_comp > (h.num_label_bytes - static_cast<uint64_t>(s.keys - s.labels)) / kw) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	s.model = s.labels + h.num_label_bytes;
	s.cracks = s.model + h.markov_model_bytes();
	uint64_t cb = 0;
	for (uint64_t z = 0; z < h.sz; z++) cb += rd_le(s.z_index + 4 * z, 4);
	s.crack_bytes = cb;
	if (static_cast<uint64_t>(s.cracks - buf) + cb + tail > n) throw Error(CKL_ERR_RUNTIME, "crackle: Unable to read past end of buffer.");
	s.crcs = buf + n - 4ull * h.sz;
	return s;
}

}  // namespace

extern "C" int ckl_zstack(const uint8_t* const* bufs, const uint64_t* lens, uint64_t count, uint8_t** out, uint64_t* out_len) {
	try {
		if (!bufs || !lens || !out || !out_len || count == 0) throw Error(CKL_ERR_ARG, "crackle_amd: zstack needs at least one stream");
		std::vector<Slab> slabs;
		slabs.reserve(count);
		for (uint64_t i = 0; i < count; i++) slabs.push_back(parse_slab(bufs[i], lens[i]));
		const Header& h0 = slabs[0].h;
		uint64_t sz = 0, total_comp = 0, crack_bytes = 0;
		for (const Slab& s : slabs) {
			const Header& h = s.h;
			if (h.sx != h0.sx || h.sy != h0.sy || h.data_width != h0.data_width
				|| h.crack_format != h0.crack_format || h.fortran_order != h0.fortran_order || h.is_signed != h0.is_signed
				|| h.markov_model_order != h0.markov_model_order) {
				throw Error(CKL_ERR_ARG, "crackle_amd: zstack slabs disagree on shape, dtype, crack format or markov order");
			}
			if (h.markov_model_order && memcmp(s.model, slabs[0].model, h.markov_model_bytes()) != 0) {
				throw Error(CKL_ERR_ARG, "crackle_amd: zstack slabs were encoded with different markov models");
			}
			sz += h.sz; total_comp += s.total_comp; crack_bytes += s.crack_bytes;
		}
		if (sz > 0xFFFFFFFFull) throw Error(CKL_ERR_ARG, "crackle_amd: zstack result has too many slices");

		std::vector<uint64_t> uniq;
		for (const Slab& s : slabs) uniq.insert(uniq.end(), s.uniq.begin(), s.uniq.end());
		std::sort(uniq.begin(), uniq.end());
		uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());

		// stored width of the whole: the slabs may have been narrowed to their own labels (zsplit)
		int sw = 1;
		for (const Slab& s : slabs) sw = std::max(sw, s.h.stored_data_width);
		if (!h0.is_signed && !uniq.empty()) sw = byte_width(uniq.back());
		const int cw = byte_width(static_cast<uint64_t>(h0.sx) * h0.sy);
		const int kw = byte_width(uniq.size());
		std::vector<uint8_t> labels_binary;
		labels_binary.reserve(8 + uniq.size() * sw + sz * cw + total_comp * kw);
		put_le(labels_binary, uniq.size(), 8);
		for (uint64_t v : uniq) put_le(labels_binary, v, sw);
		for (const Slab& s : slabs) labels_binary.insert(labels_binary.end(), s.comp, s.comp + static_cast<uint64_t>(cw) * s.h.sz);
		for (const Slab& s : slabs) {
			std::vector<uint64_t> remap(s.uniq.size());
			for (size_t i = 0; i < s.uniq.size(); i++) remap[i] = static_cast<uint64_t>(std::lower_bound(uniq.begin(), uniq.end(), s.uniq[i]) - uniq.begin());
			const int skw = byte_width(s.num_unique);
			for (uint64_t i = 0; i < s.total_comp; i++) {
				const uint64_t key = rd_le(s.keys + i * skw, skw);
				if (key >= remap.size()) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
				put_le(labels_binary, remap[key], kw);
			}
		}

		Header h = h0;
		h.stored_data_width = sw;
		h.sz = static_cast<uint32_t>(sz);
		h.num_label_bytes = labels_binary.size();
		std::vector<uint8_t> bin;
		bin.reserve(Header::kBytes + 8 * (sz + 1) + labels_binary.size() + h.markov_model_bytes() + crack_bytes);
		h.write(bin);
		const size_t zi0 = bin.size();
		for (const Slab& s : slabs) bin.insert(bin.end(), s.z_index, s.z_index + 4ull * s.h.sz);
		put_le(bin, crc32c(bin.data() + zi0, 4ull * sz), 4);
		bin.insert(bin.end(), labels_binary.begin(), labels_binary.end());
		if (h.markov_model_order) bin.insert(bin.end(), slabs[0].model, slabs[0].model + h.markov_model_bytes());
		for (const Slab& s : slabs) bin.insert(bin.end(), s.cracks, s.cracks + s.crack_bytes);
		put_le(bin, crc32c(labels_binary.data(), labels_binary.size()), 4);
		for (const Slab& s : slabs) bin.insert(bin.end(), s.crcs, s.crcs + 4ull * s.h.sz);

		uint8_t* p = static_cast<uint8_t*>(malloc(bin.size() ? bin.size() : 1));
		if (!p) throw Error(CKL_ERR_RUNTIME, "crackle_amd: out of host memory");
		memcpy(p, bin.data(), bin.size());
		*out = p;
		*out_len = bin.size();
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

// Native form of crackle.operations.zsplit's helper (crackle/operations.py:550-623): the stream
// of slices [z_start, z_end) of a FLAT stream, without decoding: crack codes, z-index entries and
// slice crcs are copied, the label table is narrowed to the labels the range uses (sorted),
// keys re-keyed, stored width = byte width of the largest label left.  Unlike the reference's
// helper the markov model is carried along, so streams with a model stay decodable.
extern "C" int ckl_zsplit(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, uint8_t** out, uint64_t* out_len) {
	try {
		if (!buf || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		const Slab s = parse_slab(buf, n);
		const Header& h0 = s.h;
		if (z_start < 0 || z_end > static_cast<int64_t>(h0.sz) || z_start >= z_end) {
			throw Error(CKL_ERR_ARG, "crackle_amd: zsplit range " + std::to_string(z_start) + " - " + std::to_string(z_end) + " is outside 0 - " + std::to_string(h0.sz));
		}
		const int cw = byte_width(static_cast<uint64_t>(h0.sx) * h0.sy);
		const int skw = byte_width(s.num_unique);
		uint64_t k0 = 0, k1 = 0, c0 = 0, c1 = 0;
		for (int64_t z = 0; z < z_end; z++) {
			const uint64_t nc = rd_le(s.comp + z * cw, cw);
			const uint64_t cb = rd_le(s.z_index + 4 * z, 4);
			if (z < z_start) { k0 += nc; c0 += cb; }
			k1 += nc; c1 += cb;
		}
		// labels the range uses, in sorted order (the table is sorted: keep the used keys)
		std::vector<uint8_t> used(s.num_unique, 0);
		for (uint64_t i = k0; i < k1; i++) {
			const uint64_t key = rd_le(s.keys + i * skw, skw);
			if (key >= s.num_unique) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
			used[key] = 1;
		}
		std::vector<uint64_t> remap(s.num_unique, 0), uniq;
		for (uint64_t k = 0; k < s.num_unique; k++) if (used[k]) { remap[k] = uniq.size(); uniq.push_back(s.uniq[k]); }
		uint64_t mx = 0;
		for (uint64_t v : uniq) mx = std::max(mx, v);
		Header h = h0;
		h.sz = static_cast<uint32_t>(z_end - z_start);
		h.stored_data_width = h0.is_signed ? h0.stored_data_width : byte_width(mx);
		const int sw = h.stored_data_width, kw = byte_width(uniq.size());
		const uint64_t nsl = h.sz;
		h.num_label_bytes = 8 + uniq.size() * sw + nsl * cw + (k1 - k0) * kw;
		const uint64_t mb = h.markov_model_bytes();
		const uint64_t total = Header::kBytes + 4 * (nsl + 1) + h.num_label_bytes + mb + (c1 - c0) + 4 * (nsl + 1);
		uint8_t* o = static_cast<uint8_t*>(malloc(total));
		if (!o) throw Error(CKL_ERR_RUNTIME, "crackle_amd: out of host memory");
		std::vector<uint8_t> hb;
		h.write(hb);
		memcpy(o, hb.data(), hb.size());
		uint64_t at = hb.size();
		auto put = [&](uint64_t v, int w) { for (int b = 0; b < w; b++) o[at++] = static_cast<uint8_t>((v >> (8 * b)) & 0xFF); };
		memcpy(o + at, s.z_index + 4 * z_start, 4 * nsl); at += 4 * nsl;
		put(crc32c(o + hb.size(), 4 * nsl), 4);
		const uint64_t lab0 = at;
		put(uniq.size(), 8);
		for (uint64_t v : uniq) put(v, sw);
		memcpy(o + at, s.comp + z_start * cw, nsl * cw); at += nsl * cw;
		for (uint64_t i = k0; i < k1; i++) put(remap[rd_le(s.keys + i * skw, skw)], kw);
		if (mb) { memcpy(o + at, s.model, mb); at += mb; }
		memcpy(o + at, s.cracks + c0, c1 - c0); at += c1 - c0;
		put(crc32c(o + lab0, h.num_label_bytes), 4);
		memcpy(o + at, s.crcs + 4 * z_start, 4 * nsl); at += 4 * nsl;
		*out = o;
		*out_len = at;
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}
