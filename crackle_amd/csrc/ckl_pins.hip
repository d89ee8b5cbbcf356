// Pin label encoding — host side.  Replaces pins::compute with the fast solver
// (src/pins.hpp:95-198, 300-403) and labels::encode_condensed_pins
// (src/labels.hpp:157-344).
//
// The greedy cover is order sensitive: it repeatedly takes `*universe.begin()` of a
// robin_hood::unordered_flat_set<uint32_t> (src/pins.hpp:325), the labels are visited in
// the slot order of a robin_hood::unordered_node_map, and ties in the background-colour
// choice follow the iteration order of a libstdc++ std::unordered_map (SURVEY.md hard
// part 2, Q6, Q7).  RhTable below is this repository's own implementation of the
// published robin-hood-hashing 3.11.5 table behaviour (hash mixing, info bytes, shift-up
// insertion, backward-shift erase, 80 % load, growth in old-slot order), written so that
// slot order is identical; the libstdc++ container is simply used as is.
//
// Per-voxel work (component labelling, crc32c) stays on the device (ckl_encode.hip); this
// file consumes the label volume and the global component-id volume on the host.
#include "ckl_common.hpp"

#include <algorithm>
#include <thread>
#include <unordered_map>

namespace ckl {

namespace {

// open-addressing table with robin-hood-hashing 3.11.5 slot order (keys: uint64, one
// uint32 payload per key)
class RhTable {
public:
	std::vector<uint64_t> keys;
	std::vector<uint32_t> vals;
	std::vector<uint8_t> info;
	size_t mask = 0, num = 0, max_allowed = 0, nbuf = 0;
	uint32_t info_inc = 32, info_shift = 0;
	uint64_t mult = 0xc4ceb9fe1a85ec53ull;

	static size_t calc_max(size_t n) { return n * 80 / 100; }
	static size_t with_buffer(size_t n) { const size_t m = calc_max(n); return n + std::min<size_t>(m, 0xFF); }

	bool allocated() const { return mask != 0; }

	void key_to_idx(uint64_t key, size_t& idx, uint32_t& inf) const {
		uint64_t h = key;
		h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
		h *= mult; h ^= h >> 33;
		inf = info_inc + static_cast<uint32_t>((h & 31u) >> info_shift);
		idx = static_cast<size_t>(h >> 5) & mask;
	}

	// returns the slot of `key`, inserting it (with payload `val`) when absent
	size_t insert(uint64_t key, uint32_t val, bool& found) {
		for (int attempt = 0; attempt < 256; attempt++) {
			size_t idx = 0; uint32_t inf = 0;
			if (mask) {
				key_to_idx(key, idx, inf);
				while (inf < info[idx]) { idx++; inf += info_inc; }
				while (inf == info[idx]) {
					if (keys[idx] == key) { found = true; return idx; }
					idx++; inf += info_inc;
				}
			}
			if (num >= max_allowed) { increase_size(); continue; }
			const size_t ins = idx;
			const uint32_t ins_info = inf;
			if (ins_info + info_inc > 0xFF) max_allowed = 0;
			while (info[idx] != 0) { idx++; inf += info_inc; }
			if (idx != ins) shift_up(idx, ins);
			keys[ins] = key; vals[ins] = val;
			info[ins] = static_cast<uint8_t>(ins_info);
			num++;
			found = false;
			return ins;
		}
		throw Error(CKL_ERR_RUNTIME, "crackle_amd: hash table overflow");
	}
	bool find(uint64_t key, size_t& slot) const {
		if (!mask) return false;
		size_t idx; uint32_t inf;
		key_to_idx(key, idx, inf);
		do {
			if (inf == info[idx] && keys[idx] == key) { slot = idx; return true; }
			idx++; inf += info_inc;
		} while (inf <= info[idx]);
		return false;
	}
	void erase(uint64_t key) {
		size_t idx;
		if (!find(key, idx)) return;
		while (info[idx + 1] >= 2 * info_inc) {
			info[idx] = static_cast<uint8_t>(info[idx + 1] - info_inc);
			keys[idx] = keys[idx + 1]; vals[idx] = vals[idx + 1];
			idx++;
		}
		info[idx] = 0;
		num--;
	}
	// first occupied slot in slot order (begin())
	bool first(size_t& slot) const {
		if (!mask || !num) return false;
		for (size_t i = 0; i < nbuf; i++) if (info[i]) { slot = i; return true; }
		return false;
	}

private:
	void init_data(size_t buckets) {
		num = 0;
		mask = buckets - 1;
		max_allowed = calc_max(buckets);
		nbuf = with_buffer(buckets);
		keys.assign(nbuf + 1, 0);
		vals.assign(nbuf + 1, 0);
		info.assign(nbuf + 16, 0);
		info[nbuf] = 1;   // sentinel
		info_inc = 32; info_shift = 0;
	}
	void shift_up(size_t start, size_t ins) {
		for (size_t i = start; i != ins; i--) { keys[i] = keys[i - 1]; vals[i] = vals[i - 1]; }
		for (size_t i = start; i != ins; i--) {
			info[i] = static_cast<uint8_t>(info[i - 1] + info_inc);
			if (static_cast<uint32_t>(info[i]) + info_inc > 0xFF) max_allowed = 0;
		}
	}
	bool try_increase_info() {
		if (info_inc <= 2) return false;
		info_inc >>= 1;
		info_shift++;
		for (size_t i = 0; i < nbuf; i++) info[i] = static_cast<uint8_t>((info[i] >> 1) & 0x7f);
		info[nbuf] = 1;
		max_allowed = calc_max(mask + 1);
		return true;
	}
	void insert_move(uint64_t key, uint32_t val) {
		if (max_allowed == 0 && !try_increase_info()) throw Error(CKL_ERR_RUNTIME, "crackle_amd: hash table overflow");
		size_t idx; uint32_t inf;
		key_to_idx(key, idx, inf);
		while (inf <= info[idx]) { idx++; inf += info_inc; }
		const size_t ins = idx;
		const uint8_t ins_info = static_cast<uint8_t>(inf);
		if (static_cast<uint32_t>(ins_info) + info_inc > 0xFF) max_allowed = 0;
		while (info[idx] != 0) { idx++; inf += info_inc; }
		if (idx != ins) shift_up(idx, ins);
		keys[ins] = key; vals[ins] = val;
		info[ins] = ins_info;
		num++;
	}
	void rehash(size_t buckets) {
		std::vector<uint64_t> ok; std::vector<uint32_t> ov; std::vector<uint8_t> oi;
		ok.swap(keys); ov.swap(vals); oi.swap(info);
		const size_t old_nbuf = mask ? with_buffer(mask + 1) : 0;
		init_data(buckets);
		for (size_t i = 0; i < old_nbuf; i++) if (oi[i]) insert_move(ok[i], ov[i]);
	}
	void increase_size() {
		if (mask == 0) { init_data(8); return; }
		const size_t maxn = calc_max(mask + 1);
		if (num < maxn && try_increase_info()) return;
		if (num * 2 < maxn) {
			mult += 0xc4ceb9fe1a85ec54ull;
			rehash(mask + 1);
		}
		else rehash((mask + 1) * 2);
	}
};

// candidate pin: a maximal z-run of one label in one (x, y) column (src/pins.hpp:51-93);
// its component ids are cc[x, y, z_s .. z_e], looked up on demand
struct CandidatePin {
	uint32_t x, y, z_s, z_e;
};

}  // namespace

template <typename LABEL>
std::vector<uint8_t> encode_pins_host(
	const LABEL* labels, const uint32_t* cc /* global component ids */,
	int64_t sx, int64_t sy, int64_t sz,
	const std::vector<uint32_t>& ncomp, uint64_t n_total,
	int index_width, int stored_width, bool auto_bgcolor, int64_t manual_bgcolor
) {
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy, voxels = sxy * sz;

	// ---- extract_columns (src/pins.hpp:126-163) with add_pin's dedup (95-124) ----
	RhTable pinsets;                                  // label -> index into pvecs, reference slot order
	std::vector<std::vector<CandidatePin>> pvecs;
	auto pinset_of = [&](uint64_t label) -> std::vector<CandidatePin>& {
		bool found;
		const size_t s = pinsets.insert(label, static_cast<uint32_t>(pvecs.size()), found);
		if (!found) pvecs.emplace_back();
		return pvecs[pinsets.vals[s]];
	};
	auto add_pin = [&](uint64_t label, uint64_t z_start, uint64_t x, uint64_t y, uint64_t z) {
		std::vector<CandidatePin>& v = pinset_of(label);
		const CandidatePin np{ static_cast<uint32_t>(x), static_cast<uint32_t>(y), static_cast<uint32_t>(z_start), static_cast<uint32_t>(z) };
		if (v.empty()) { v.push_back(np); return; }
		CandidatePin& last = v.back();
		if (static_cast<uint64_t>(last.x) == x - 1 && static_cast<uint64_t>(last.y) == y) {
			if (last.z_s <= z_start && last.z_e >= z) return;
			else if (last.z_s >= z_start && last.z_e <= z) last = np;
			else v.push_back(np);
		}
		else v.push_back(np);
	};
	for (uint64_t y = 0; y < static_cast<uint64_t>(sy); y++) {
		for (uint64_t x = 0; x < static_cast<uint64_t>(sx); x++) {
			const uint64_t loc = x + static_cast<uint64_t>(sx) * y;
			LABEL label = labels[loc];
			uint64_t z_start = 0, z = 1;
			for (; z < static_cast<uint64_t>(sz); z++) {
				const LABEL cur = labels[loc + sxy * z];
				if (label != cur) {
					add_pin(label, z_start, x, y, z - 1);
					label = cur;
					z_start = z;
				}
			}
			if (sz == 1) z = 0;
			add_pin(label, z_start, x, y, z - 1);
		}
	}

	// ---- compute_multiverse (src/pins.hpp:165-198): per label the flat set of its
	// component ids, inserted in linear voxel order at every change of component id ----
	std::vector<RhTable> universe(pvecs.size());
	{
		bool f; size_t s;
		uint32_t last = cc[0];
		if (pinsets.find(labels[0], s)) universe[pinsets.vals[s]].insert(cc[0], 0, f);
		for (uint64_t i = 1; i < voxels; i++) {
			if (cc[i] != last) {
				if (pinsets.find(labels[i], s)) universe[pinsets.vals[s]].insert(cc[i], 0, f);
				last = cc[i];
			}
		}
		if (pinsets.find(labels[voxels - 1], s)) universe[pinsets.vals[s]].insert(cc[voxels - 1], 0, f);
	}

	// ---- find_suboptimal_pins per label (src/pins.hpp:300-346).  Labels are independent:
	// solved on a thread pool, stored by label index ----
	std::vector<std::vector<CandidatePin>> chosen(pvecs.size());
	auto solve = [&](size_t li) {
		const std::vector<CandidatePin>& pv = pvecs[li];
		RhTable& uni = universe[li];
		// component -> candidate pins that contain it, in pin order (CSR over this label's components)
		RhTable c2p;
		std::vector<uint32_t> counts;
		for (size_t i = 0; i < pv.size(); i++) {
			const uint64_t base = pv[i].x + static_cast<uint64_t>(sx) * pv[i].y;
			for (uint32_t z = pv[i].z_s; z <= pv[i].z_e; z++) {
				bool f;
				const size_t s = c2p.insert(cc[base + sxy * z], static_cast<uint32_t>(counts.size()), f);
				if (!f) counts.push_back(0);
				counts[c2p.vals[s]]++;
			}
		}
		std::vector<uint32_t> start(counts.size() + 1, 0);
		for (size_t k = 0; k < counts.size(); k++) start[k + 1] = start[k] + counts[k];
		std::vector<uint32_t> fill(start.begin(), start.end() - 1), members(start.back());
		for (size_t i = 0; i < pv.size(); i++) {
			const uint64_t base = pv[i].x + static_cast<uint64_t>(sx) * pv[i].y;
			for (uint32_t z = pv[i].z_s; z <= pv[i].z_e; z++) {
				size_t s;
				c2p.find(cc[base + sxy * z], s);
				members[fill[c2p.vals[s]]++] = static_cast<uint32_t>(i);
			}
		}
		std::vector<CandidatePin>& out = chosen[li];
		while (!pv.empty() && uni.num) {
			size_t us;
			if (!uni.first(us)) break;
			const uint64_t picked = uni.keys[us];
			size_t cs;
			if (!c2p.find(picked, cs)) { uni.erase(picked); continue; }   // cannot happen: every component lies on a column run
			const uint32_t l = c2p.vals[cs];
			const CandidatePin* max_pin = &pv[members[start[l]]];
			const int max_depth = static_cast<int>(max_pin->z_e - max_pin->z_s);   // never updated
			for (uint32_t m = start[l] + 1; m < start[l + 1]; m++) {
				const CandidatePin* cur = &pv[members[m]];
				if (static_cast<int>(cur->z_e - cur->z_s) > max_depth) max_pin = cur;
			}
			const uint64_t base = max_pin->x + static_cast<uint64_t>(sx) * max_pin->y;
			for (uint32_t z = max_pin->z_s; z <= max_pin->z_e; z++) uni.erase(cc[base + sxy * z]);
			out.push_back(*max_pin);
		}
	};
	{
		const size_t nl = pvecs.size();
		size_t nthreads = std::min<size_t>(std::max<size_t>(1, std::thread::hardware_concurrency()), 64);
		nthreads = std::min(nthreads, std::max<size_t>(1, nl / 16));
		if (nthreads <= 1) {
			for (size_t li = 0; li < nl; li++) solve(li);
		}
		else {
			std::vector<std::thread> pool;
			std::vector<std::string> errors(nthreads);
			for (size_t t = 0; t < nthreads; t++) {
				pool.emplace_back([&, t]() {
					try { for (size_t li = t; li < nl; li += nthreads) solve(li); }
					catch (const std::exception& e) { errors[t] = e.what(); }
				});
			}
			for (auto& th : pool) th.join();
			for (auto& e : errors) if (!e.empty()) throw Error(CKL_ERR_RUNTIME, e);
		}
	}

	// ---- all_pins: libstdc++ unordered_map filled in pinsets slot order (src/pins.hpp:374-388);
	// its iteration order breaks ties in find_bgcolor (src/labels.hpp:157-190) ----
	std::unordered_map<uint64_t, uint32_t> all_pins;   // label -> label index
	all_pins.reserve(128);
	for (size_t slot = 0; slot < pinsets.nbuf && pinsets.allocated(); slot++) {
		if (!pinsets.info[slot]) continue;
		all_pins[pinsets.keys[slot]] = pinsets.vals[slot];
	}
	auto total_depth = [](const std::vector<CandidatePin>& v) {
		uint64_t d = 0;
		for (const CandidatePin& p : v) d += p.z_e - p.z_s;
		return d;
	};
	uint64_t bgcolor = (manual_bgcolor != 0) ? 1 : 0;   // Q1: compress_helper takes `const bool manual_bgcolor`
	if (auto_bgcolor) {
		bgcolor = 0;
		uint64_t max_pins = 0, max_pins_depth = static_cast<uint64_t>(sz);
		for (const auto& kv : all_pins) {
			const std::vector<CandidatePin>& v = chosen[kv.second];
			if (v.size() > max_pins) {
				bgcolor = kv.first;
				max_pins = v.size();
				max_pins_depth = total_depth(v);
			}
			else if (v.size() == max_pins) {
				const uint64_t d = total_depth(v);
				if (d > max_pins_depth) { bgcolor = kv.first; max_pins_depth = d; }
			}
		}
	}
	if (stored_width < 8) bgcolor &= (1ull << (8 * stored_width)) - 1;
	all_pins.erase(bgcolor);

	// ---- encode_condensed_pins (src/labels.hpp:192-344) ----
	uint64_t max_pins = 0, max_depth = 0;
	std::vector<uint64_t> all_labels;
	all_labels.reserve(all_pins.size());
	for (const auto& kv : all_pins) {
		const std::vector<CandidatePin>& v = chosen[kv.second];
		max_pins = std::max<uint64_t>(max_pins, v.size());
		for (const CandidatePin& p : v) max_depth = std::max<uint64_t>(max_depth, p.z_e - p.z_s);
		all_labels.push_back(kv.first);
	}
	std::sort(all_labels.begin(), all_labels.end());

	const int num_pins_width = byte_width(max_pins);
	const int depth_width = byte_width(max_depth);
	const int cc_label_width = byte_width(n_total);
	const int component_width = byte_width(sxy);
	const uint8_t pin_bytes = static_cast<uint8_t>(index_width + depth_width);
	const uint8_t cc_efficient_threshold = static_cast<uint8_t>(pin_bytes / cc_label_width);
	const uint8_t combined = static_cast<uint8_t>(ilog2w(num_pins_width) | (ilog2w(depth_width) << 2) | (ilog2w(cc_label_width) << 4));

	std::vector<uint8_t> bin;
	put_le(bin, bgcolor, stored_width);
	put_le(bin, all_labels.size(), 8);
	for (uint64_t l : all_labels) put_le(bin, l, stored_width);
	for (int64_t z = 0; z < sz; z++) put_le(bin, ncomp[z], component_width);
	bin.push_back(combined);

	struct Sorted { uint64_t idx, depth; const CandidatePin* pin; };
	for (uint64_t label : all_labels) {
		const std::vector<CandidatePin>& v = chosen[all_pins[label]];
		std::vector<Sorted> sp;
		sp.reserve(v.size());
		for (const CandidatePin& p : v) {
			sp.push_back({ static_cast<uint64_t>(p.x) + static_cast<uint64_t>(sx) * (static_cast<uint64_t>(p.y) + static_cast<uint64_t>(sy) * p.z_s),
				static_cast<uint64_t>(p.z_e - p.z_s), &p });
		}
		std::sort(sp.begin(), sp.end(), [](const Sorted& a, const Sorted& b) { return a.idx < b.idx; });
		uint64_t n_pin_repr = 0;
		for (const Sorted& s : sp) n_pin_repr += (s.depth >= cc_efficient_threshold);
		put_le(bin, n_pin_repr, num_pins_width);
		uint64_t prev = 0;
		bool first = true;
		for (const Sorted& s : sp) {
			if (s.depth < cc_efficient_threshold) continue;
			put_le(bin, first ? s.idx : s.idx - prev, index_width);
			prev = s.idx;
			first = false;
		}
		for (const Sorted& s : sp) if (s.depth >= cc_efficient_threshold) put_le(bin, s.depth, depth_width);
		std::vector<uint32_t> ids;
		for (const Sorted& s : sp) {
			if (s.depth >= cc_efficient_threshold) continue;
			const uint64_t base = s.pin->x + static_cast<uint64_t>(sx) * s.pin->y;
			for (uint32_t z = s.pin->z_s; z <= s.pin->z_e; z++) ids.push_back(cc[base + sxy * z]);
		}
		std::sort(ids.begin(), ids.end());
		put_le(bin, ids.size(), num_pins_width);
		for (size_t k = 0; k < ids.size(); k++) put_le(bin, k ? static_cast<uint32_t>(ids[k] - ids[k - 1]) : ids[k], cc_label_width);
	}
	return bin;
}

template std::vector<uint8_t> encode_pins_host<uint8_t>(const uint8_t*, const uint32_t*, int64_t, int64_t, int64_t, const std::vector<uint32_t>&, uint64_t, int, int, bool, int64_t);
template std::vector<uint8_t> encode_pins_host<uint16_t>(const uint16_t*, const uint32_t*, int64_t, int64_t, int64_t, const std::vector<uint32_t>&, uint64_t, int, int, bool, int64_t);
template std::vector<uint8_t> encode_pins_host<uint32_t>(const uint32_t*, const uint32_t*, int64_t, int64_t, int64_t, const std::vector<uint32_t>&, uint64_t, int, int, bool, int64_t);
template std::vector<uint8_t> encode_pins_host<uint64_t>(const uint64_t*, const uint32_t*, int64_t, int64_t, int64_t, const std::vector<uint32_t>&, uint64_t, int, int, bool, int64_t);

}  // namespace ckl

extern "C" int ckl_pin_labels_host(
	const void* labels, int dtype_bytes, const uint32_t* cc,
	int64_t sx, int64_t sy, int64_t sz, const uint32_t* ncomp,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor,
	uint8_t** out, uint64_t* out_len
) {
	using namespace ckl;
	try {
		if (!labels || !cc || !ncomp || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (sx <= 0 || sy <= 0 || sz <= 0) throw Error(CKL_ERR_ARG, "crackle_amd: empty volume");
		if (stored_width != 1 && stored_width != 2 && stored_width != 4 && stored_width != 8) throw Error(CKL_ERR_ARG, "crackle_amd: stored width must be 1, 2, 4 or 8 bytes");
		std::vector<uint32_t> nc(ncomp, ncomp + sz);
		uint64_t total = 0;
		for (uint32_t c : nc) total += c;
		Header h;
		h.sx = static_cast<uint32_t>(sx); h.sy = static_cast<uint32_t>(sy); h.sz = static_cast<uint32_t>(sz);
		std::vector<uint8_t> bin;
#define CKL_PINS(T) bin = encode_pins_host<T>(static_cast<const T*>(labels), cc, sx, sy, sz, nc, total, h.pin_index_width(), stored_width, auto_bgcolor != 0, manual_bgcolor)
		if (dtype_bytes == 1) CKL_PINS(uint8_t);
		else if (dtype_bytes == 2) CKL_PINS(uint16_t);
		else if (dtype_bytes == 4) CKL_PINS(uint32_t);
		else if (dtype_bytes == 8) CKL_PINS(uint64_t);
		else throw Error(CKL_ERR_ARG, "crackle_amd: dtype width must be 1, 2, 4 or 8 bytes");
#undef CKL_PINS
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
