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
// Per-voxel work (component labelling, crc32c, the column runs and their dedup, the
// component -> pin choice) runs on the device (ckl_encode.hip, ckl_pins_dev.hpp); this file
// consumes the per-component facts of ckl_pins.hpp.  pin_candidates_host restates the device
// passes with the reference's own loops for the CPU tests and the sharded whole-volume stage.
#include "ckl_pins.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <unistd.h>
#include <functional>
#include <thread>
#include <sched.h>
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
	// back to the state of a new table (the vectors keep their memory)
	void reset() { mask = 0; num = 0; max_allowed = 0; nbuf = 0; info_inc = 32; info_shift = 0; mult = 0xc4ceb9fe1a85ec53ull; }

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
	// first occupied slot in slot order (begin()).  `cursor` carries the search start between
	// calls of a phase that only erases: backward-shift deletion never moves an entry in front
	// of the first occupied slot.
	bool first(size_t& slot, size_t& cursor) const {
		if (!mask || !num) return false;
		for (size_t i = cursor; i < nbuf; i++) if (info[i]) { slot = i; cursor = i; return true; }
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

}  // namespace

struct PinLabelTable { RhTable pinsets; size_t n_labels = 0; };

// ---- host statement of the device passes (src/pins.hpp:95-163 with the choice rule of 325-340) ----
template <typename LABEL>
PinCandidates pin_candidates_host(const LABEL* labels, const uint32_t* cc, int64_t sx_, int64_t sy_, int64_t sz_, uint64_t N) {
	const uint64_t sx = static_cast<uint64_t>(sx_), sy = static_cast<uint64_t>(sy_), sz = static_cast<uint64_t>(sz_);
	const uint64_t sxy = sx * sy, voxels = sxy * sz;
	struct Cand { uint32_t x, y, z_s, z_e; };
	RhTable pinsets;                                  // label -> index into pvecs
	std::vector<std::vector<Cand>> pvecs;
	PinCandidates pc;
	pc.comp_label.assign(N, 0);
	pc.comp_first.assign(N, kPinNoKey);
	pc.comp_pin.assign(N, kPinNone);
	for (uint64_t i = 0; i < voxels; i++) {
		if (cc[i] >= N) throw Error(CKL_ERR_ARG, "crackle_amd: component id out of range");
		pc.comp_label[cc[i]] = static_cast<uint64_t>(labels[i]);
	}
	auto add_pin = [&](uint64_t label, uint64_t z_start, uint64_t x, uint64_t y, uint64_t z) {
		uint64_t& cf = pc.comp_first[cc[x + sx * y + sxy * z_start]];
		cf = std::min(cf, pin_key(x, y, z_start, sx, sz));
		bool found;
		const size_t s = pinsets.insert(label, static_cast<uint32_t>(pvecs.size()), found);
		if (!found) pvecs.emplace_back();
		std::vector<Cand>& v = pvecs[pinsets.vals[s]];
		const Cand np{ static_cast<uint32_t>(x), static_cast<uint32_t>(y), static_cast<uint32_t>(z_start), static_cast<uint32_t>(z) };
		if (v.empty()) { v.push_back(np); return; }
		Cand& last = v.back();
		if (static_cast<uint64_t>(last.x) == x - 1 && static_cast<uint64_t>(last.y) == y) {
			if (last.z_s <= z_start && last.z_e >= z) return;
			else if (last.z_s >= z_start && last.z_e <= z) last = np;
			else v.push_back(np);
		}
		else v.push_back(np);
	};
	for (uint64_t y = 0; y < sy; y++) {
		for (uint64_t x = 0; x < sx; x++) {
			const uint64_t loc = x + sx * y;
			LABEL label = labels[loc];
			uint64_t z_start = 0, z = 1;
			for (; z < sz; z++) {
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
	// the pin drawn for a component: first candidate holding it, then the last deeper one
	std::vector<uint32_t> first_depth(N, 0);
	std::vector<const Cand*> pick(N, nullptr);
	for (const std::vector<Cand>& pv : pvecs) {
		for (const Cand& p : pv) {
			const uint64_t base = p.x + sx * p.y;
			const uint32_t depth = p.z_e - p.z_s;
			for (uint32_t z = p.z_s; z <= p.z_e; z++) {
				const uint32_t c = cc[base + sxy * z];
				if (!pick[c]) { pick[c] = &p; first_depth[c] = depth; }
				else if (depth > first_depth[c]) pick[c] = &p;
			}
		}
	}
	std::vector<std::pair<uint64_t, uint32_t>> order;      // (key of the picked pin, component)
	order.reserve(N);
	for (uint64_t c = 0; c < N; c++) if (pick[c]) order.emplace_back(pin_key(pick[c]->x, pick[c]->y, pick[c]->z_s, sx, sz), static_cast<uint32_t>(c));
	std::sort(order.begin(), order.end());
	pc.pin_ids_off.push_back(0);
	for (size_t i = 0; i < order.size(); i++) {
		const Cand& p = *pick[order[i].second];
		if (i == 0 || order[i].first != order[i - 1].first) {
			pc.pin_x.push_back(p.x); pc.pin_y.push_back(p.y); pc.pin_zs.push_back(p.z_s); pc.pin_ze.push_back(p.z_e);
			const uint64_t base = p.x + sx * p.y;
			for (uint32_t z = p.z_s; z <= p.z_e; z++) pc.pin_ids.push_back(cc[base + sxy * z]);
			pc.pin_ids_off.push_back(pc.pin_ids.size());
		}
		pc.comp_pin[order[i].second] = static_cast<uint32_t>(pc.pin_x.size() - 1);
	}
	return pc;
}

template PinCandidates pin_candidates_host<uint8_t>(const uint8_t*, const uint32_t*, int64_t, int64_t, int64_t, uint64_t);
template PinCandidates pin_candidates_host<uint16_t>(const uint16_t*, const uint32_t*, int64_t, int64_t, int64_t, uint64_t);
template PinCandidates pin_candidates_host<uint32_t>(const uint32_t*, const uint32_t*, int64_t, int64_t, int64_t, uint64_t);
template PinCandidates pin_candidates_host<uint64_t>(const uint64_t*, const uint32_t*, int64_t, int64_t, int64_t, uint64_t);

namespace {

// Host threads this process may keep busy: the CPUs of its affinity mask (not the machine's: a rank of a node-wide
// job is usually pinned to its share), divided by the ranks of the node when the launcher says how many there are
// (LOCAL_WORLD_SIZE), at most 64; CKL_PINS_THREADS bounds it further (the pool itself, not only the regions).
static size_t host_cpu_budget() {
	size_t n = std::max<size_t>(1, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<size_t>(n, static_cast<size_t>(c)); }
	if (const char* env = getenv("LOCAL_WORLD_SIZE")) { const int w = atoi(env); if (w > 1) n = std::max<size_t>(1, n / static_cast<size_t>(w)); }
	if (const char* env = getenv("CKL_PINS_THREADS")) n = std::min<size_t>(n, static_cast<size_t>(std::max(1, atoi(env))));
	return std::min<size_t>(n, 64);
}

// Worker threads that stay: starting 32 threads costs 0.5-0.8 ms, and the pin stage runs some twenty parallel
// regions per volume.  One region at a time; a caller that finds the pool taken (another encoder's region), or
// that runs in a process forked after the pool was made, starts threads of its own as before.
class HostPool {
public:
	static HostPool& get() { static HostPool* pool = new HostPool();  return *pool; }      // never destroyed: the workers wait for ever
	size_t size() const { return n_workers; }
	// job(t) for t in [0, want) on the workers; false: not run (pool taken or not ours)
	bool run(size_t want, const std::function<void(size_t)>& job) {
		if (want > n_workers || getpid() != pid) return false;
		std::unique_lock<std::mutex> one(region, std::try_to_lock);
		if (!one.owns_lock()) return false;
		{
			std::lock_guard<std::mutex> lock(m);
			current = &job; n_want = want; remaining = want; generation++;
		}
		cv_work.notify_all();
		std::unique_lock<std::mutex> lock(m);
		cv_done.wait(lock, [&] { return remaining == 0; });
		current = nullptr;
		return true;
	}
private:
	HostPool() : pid(getpid()) {
		n_workers = host_cpu_budget();
		for (size_t t = 0; t < n_workers; t++) std::thread([this, t] { work(t); }).detach();
	}
	void work(size_t t) {
		uint64_t seen = 0;
		for (;;) {
			const std::function<void(size_t)>* job = nullptr;
			{
				std::unique_lock<std::mutex> lock(m);
				cv_work.wait(lock, [&] { return generation != seen; });
				seen = generation;
				if (t < n_want) job = current;
			}
			if (!job) continue;
			(*job)(t);
			std::lock_guard<std::mutex> lock(m);
			if (--remaining == 0) cv_done.notify_all();
		}
	}
	std::mutex m, region;
	std::condition_variable cv_work, cv_done;
	const std::function<void(size_t)>* current = nullptr;
	size_t n_workers = 0, n_want = 0, remaining = 0;
	uint64_t generation = 0;
	pid_t pid;
};

}  // namespace

void host_parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& body, size_t max_threads) {
	size_t nthreads = std::min<size_t>(host_cpu_budget(), max_threads);
	const size_t want = std::min(nthreads, std::max<size_t>(1, n / std::max<size_t>(grain, 1)));
	if (want <= 1) { body(0, n); return; }
	std::vector<std::string> errors(want);
	const std::function<void(size_t)> job = [&](size_t t) {
		try { body(n * t / want, n * (t + 1) / want); }
		catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; }
		catch (...) { errors[t] = "error"; }
	};
	if (getenv("CKL_PINS_NO_POOL") || !HostPool::get().run(want, job)) {
		std::vector<std::thread> pool;
		for (size_t t = 0; t < want; t++) pool.emplace_back([&, t]() { job(t); });
		for (auto& th : pool) th.join();
	}
	for (auto& e : errors) if (!e.empty()) throw Error(CKL_ERR_RUNTIME, e);
}

namespace {
// sort in pieces on the worker threads, then merge the pieces pairwise
template <typename It, typename Less>
void host_parallel_sort(It first, It last, Less less) {
	const size_t n = static_cast<size_t>(last - first);
	size_t nthreads = host_cpu_budget();
	if (const char* env = getenv("CKL_PINS_THREADS")) nthreads = static_cast<size_t>(std::max(1, atoi(env)));
	size_t pieces = 1;
	while (pieces < nthreads && pieces < 16 && n / (2 * pieces) >= 4096) pieces *= 2;
	if (pieces == 1) { std::sort(first, last, less); return; }
	auto bound = [&](size_t i) { return first + static_cast<std::ptrdiff_t>(n * i / pieces); };
	host_parallel_for(pieces, 1, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) std::sort(bound(i), bound(i + 1), less); });
	for (size_t w = 1; w < pieces; w *= 2) {
		const size_t pairs = pieces / (2 * w);
		host_parallel_for(pairs, 1, [&](size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; i++) std::inplace_merge(bound(2 * w * i), bound(2 * w * i + w), bound(2 * w * (i + 1)), less);
		});
	}
}

// the labels enter the table in the order of their first column run
void pinsets_insert_ordered(std::vector<std::pair<uint64_t, uint64_t>>& order /* (first key, label) */, RhTable& pinsets, size_t& n_labels) {
	host_parallel_sort(order.begin(), order.end(), [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) { return a < b; });
	for (const auto& o : order) {
		bool found;
		pinsets.insert(o.second, static_cast<uint32_t>(n_labels), found);
		if (!found) n_labels++;
	}
}
}  // namespace

std::shared_ptr<const PinLabelTable> pins_label_table_host(const std::vector<uint64_t>& label_value, const std::vector<uint64_t>& label_first) {
	if (label_first.size() != label_value.size()) throw Error(CKL_ERR_RUNTIME, "crackle_amd: inconsistent pin candidates");
	auto t = std::make_shared<PinLabelTable>();
	std::vector<std::pair<uint64_t, uint64_t>> order;
	order.reserve(label_value.size());
	for (size_t i = 0; i < label_value.size(); i++) if (label_first[i] != kPinNoKey) order.emplace_back(label_first[i], label_value[i]);
	pinsets_insert_ordered(order, t->pinsets, t->n_labels);
	return t;
}

std::vector<uint8_t> pins_cover_host(
	const PinCandidates& pc, int64_t sx, int64_t sy, int64_t sz,
	const std::vector<uint32_t>& ncomp, uint64_t n_total,
	int index_width, int stored_width, bool auto_bgcolor, int64_t manual_bgcolor,
	const std::function<void()>& components_ready, const PinLabelTable* table
) {
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	const bool viewed = pc.view_components != 0;
	const uint64_t N = viewed ? pc.view_components : pc.comp_label.size();
	const bool pins_viewed = viewed && pc.view_comp_pin;
	const size_t P = pins_viewed ? static_cast<size_t>(N) : pc.pin_x.size();
	const uint64_t* const comp_label = viewed ? pc.view_comp_label : pc.comp_label.data();
	const uint64_t* const ids_off = viewed ? pc.view_pin_ids_off : pc.pin_ids_off.data();
	const uint32_t* const ids = viewed ? pc.view_pin_ids : pc.pin_ids.data();
	const uint32_t* const comp_pin = pins_viewed ? pc.view_comp_pin : pc.comp_pin.data();
	const uint32_t* const pin_x = pins_viewed ? pc.view_pin_x : pc.pin_x.data();
	const uint32_t* const pin_y = pins_viewed ? pc.view_pin_y : pc.pin_y.data();
	const uint32_t* const pin_zs = pins_viewed ? pc.view_pin_zs : pc.pin_zs.data();
	const uint32_t* const pin_ze = pins_viewed ? pc.view_pin_ze : pc.pin_ze.data();
	const bool prof = getenv("CKL_PROFILE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	std::string marks;
	auto mark = [&](const char* name) {
		if (!prof) return;
		const auto now = std::chrono::steady_clock::now();
		char buf[64];
		snprintf(buf, sizeof buf, " %s=%.2f", name, std::chrono::duration<double, std::milli>(now - t_last).count());
		marks += buf;
		t_last = now;
	};
	if ((!pins_viewed && pc.comp_pin.size() != N) || (pins_viewed && (!pin_x || !pin_y || !pin_zs || !pin_ze)) || (!viewed && pc.pin_ids_off.size() != P + 1) || (viewed && (!comp_label || !ids_off || !ids))) throw Error(CKL_ERR_RUNTIME, "crackle_amd: inconsistent pin candidates");
	if (!table && pc.label_value.empty() && pc.comp_first.size() != N) throw Error(CKL_ERR_RUNTIME, "crackle_amd: inconsistent pin candidates");

	// worker threads for the per-label phases (labels are independent of each other)
	size_t nthreads = host_cpu_budget();
	if (const char* env = getenv("CKL_PINS_THREADS")) nthreads = static_cast<size_t>(std::max(1, atoi(env)));
	auto parallel_for = [&](size_t n, size_t grain, const std::function<void(size_t, size_t)>& body) { host_parallel_for(n, grain, body, 64); };

	auto parallel_sort = [&](auto first, auto last, auto less) { host_parallel_sort(first, last, less); };

	// ---- pinsets (src/pins.hpp:126-163): a robin-hood node map keyed by label; its slot order
	// depends on the order the labels were first seen, i.e. on their first column run ----
	PinLabelTable own;
	if (!table) {
		std::vector<std::pair<uint64_t, uint64_t>> order;      // (first key, label)
		if (!pc.label_value.empty()) {
			if (pc.label_first.size() != pc.label_value.size()) throw Error(CKL_ERR_RUNTIME, "crackle_amd: inconsistent pin candidates");
			order.reserve(pc.label_value.size());
			for (size_t i = 0; i < pc.label_value.size(); i++) if (pc.label_first[i] != kPinNoKey) order.emplace_back(pc.label_first[i], pc.label_value[i]);
		}
		else {
			order.reserve(N);
			for (uint64_t c = 0; c < N; c++) if (pc.comp_first[c] != kPinNoKey) order.emplace_back(pc.comp_first[c], comp_label[c]);
		}
		pinsets_insert_ordered(order, own.pinsets, own.n_labels);
	}
	const RhTable& pinsets = table ? table->pinsets : own.pinsets;
	const size_t n_labels = table ? table->n_labels : own.n_labels;
	mark("pinsets");
	if (components_ready) { components_ready(); mark("arrays"); }

	// ---- components by label (ascending id within a label = linear voxel order, the insertion
	// order of compute_multiverse, src/pins.hpp:165-198) ----
	std::vector<uint32_t> comp_li(N);
	parallel_for(N, 16384, [&](size_t lo, size_t hi) {
		for (size_t c = lo; c < hi; c++) {
			size_t s;
			if (!pinsets.find(comp_label[c], s)) throw Error(CKL_ERR_RUNTIME, "crackle_amd: component of a label without column runs");
			comp_li[c] = pinsets.vals[s];
		}
	});
	// (a counting sort by label, stable in the component id; in T pieces of the components, each with its own counts)
	std::vector<uint64_t> li_at(n_labels + 1, 0);
	std::vector<uint32_t> li_comp(N);
	{
		const size_t T = (N >= (1u << 14) && n_labels <= (1u << 22)) ? std::min<size_t>(nthreads, 8) : 1;
		std::vector<uint32_t> cnt(T * n_labels, 0);      // then: where piece t puts its next component of label l
		parallel_for(T, 1, [&](size_t lo, size_t hi) {
			for (size_t t = lo; t < hi; t++) {
				uint32_t* mine = cnt.data() + t * n_labels;
				for (uint64_t c = N * t / T; c < N * (t + 1) / T; c++) mine[comp_li[c]]++;
			}
		});
		for (size_t l = 0; l < n_labels; l++) {
			uint64_t tot = 0;
			for (size_t t = 0; t < T; t++) tot += cnt[t * n_labels + l];
			li_at[l + 1] = li_at[l] + tot;
		}
		if (li_at[n_labels] != N || N > 0xFFFFFFFFull) throw Error(CKL_ERR_RUNTIME, "crackle_amd: internal: component counts by label");
		parallel_for(n_labels, 4096, [&](size_t lo, size_t hi) {
			for (size_t l = lo; l < hi; l++) {
				uint32_t at = static_cast<uint32_t>(li_at[l]);
				for (size_t t = 0; t < T; t++) { const uint32_t n = cnt[t * n_labels + l]; cnt[t * n_labels + l] = at; at += n; }
			}
		});
		parallel_for(T, 1, [&](size_t lo, size_t hi) {
			for (size_t t = lo; t < hi; t++) {
				uint32_t* mine = cnt.data() + t * n_labels;
				for (uint64_t c = N * t / T; c < N * (t + 1) / T; c++) li_comp[mine[comp_li[c]]++] = static_cast<uint32_t>(c);
			}
		});
	}
	mark("universe");

	// ---- per label: the universe (a robin-hood flat set, filled in ascending id) and
	// find_suboptimal_pins (src/pins.hpp:300-346).  A label takes at most one pin per component: its
	// pins, in the order taken, go to chosen[li_at[li] ...] (no vector per label: 143 k labels at C4) ----
	auto depth_of = [&](uint32_t p) { return static_cast<uint64_t>(pin_ze[p] - pin_zs[p]); };
	std::vector<uint32_t> chosen(N);
	std::vector<uint32_t> n_chosen(n_labels, 0);
	std::vector<uint64_t> depth_sum(n_labels, 0), depth_max(n_labels, 0);
	parallel_for(n_labels, 32, [&](size_t lo, size_t hi) {
		RhTable uni;
		for (size_t li = lo; li < hi; li++) {
			uni.reset();
			bool f;
			for (uint64_t k = li_at[li]; k < li_at[li + 1]; k++) uni.insert(li_comp[k], 0, f);
			uint32_t* out = chosen.data() + li_at[li];
			const uint64_t room = li_at[li + 1] - li_at[li];
			uint32_t taken = 0;
			uint64_t dsum = 0, dmax = 0;
			size_t cursor = 0;
			while (uni.num) {
				size_t us;
				if (!uni.first(us, cursor)) break;
				const uint64_t picked = uni.keys[us];
				const uint32_t p = comp_pin[picked];
				if (p == kPinNone) { uni.erase(picked); continue; }   // cannot happen: every component lies on a kept column run
				const size_t before = uni.num;
				for (uint64_t k = ids_off[p]; k < ids_off[p + 1]; k++) uni.erase(ids[k]);
				if (uni.num == before) throw Error(CKL_ERR_RUNTIME, "crackle_amd: a pin without its own component");      // (would loop for ever)
				if (taken >= room) throw Error(CKL_ERR_RUNTIME, "crackle_amd: more pins than components");
				out[taken++] = p;
				const uint64_t d = depth_of(p);
				dsum += d; dmax = std::max(dmax, d);
			}
			n_chosen[li] = taken; depth_sum[li] = dsum; depth_max[li] = dmax;
		}
	});
	mark("cover");

	// ---- all_pins: libstdc++ unordered_map filled in pinsets slot order (src/pins.hpp:374-388);
	// its iteration order breaks ties in find_bgcolor (src/labels.hpp:157-190).  find_bgcolor keeps the
	// first label in that order with the greatest (number of pins, summed depth): when one label alone
	// holds the maximum the order does not matter and the map (143 k node allocations at C4) is not built ----
	std::vector<std::pair<uint64_t, uint32_t>> slot_labels;      // (label, label index) in slot order
	slot_labels.reserve(n_labels);
	for (size_t slot = 0; slot < pinsets.nbuf && pinsets.allocated(); slot++) {
		if (pinsets.info[slot]) slot_labels.emplace_back(pinsets.keys[slot], pinsets.vals[slot]);
	}
	uint64_t bgcolor = (manual_bgcolor != 0) ? 1 : 0;   // Q1: compress_helper takes `const bool manual_bgcolor`
	if (auto_bgcolor) {
		bgcolor = 0;
		uint64_t max_pins = 0, max_pins_depth = static_cast<uint64_t>(sz);
		size_t holders = 0;
		for (const auto& kv : slot_labels) {
			const uint64_t np = n_chosen[kv.second], d = depth_sum[kv.second];
			if (np > max_pins || (np == max_pins && d > max_pins_depth)) { bgcolor = kv.first; max_pins = np; max_pins_depth = d; holders = 1; }
			else if (np == max_pins && d == max_pins_depth) holders++;
		}
		if (holders > 1 || getenv("CKL_PINS_BGCOLOR_MAP")) {
			std::unordered_map<uint64_t, uint32_t> all_pins;   // label -> label index
			all_pins.reserve(128);
			for (const auto& kv : slot_labels) all_pins[kv.first] = kv.second;
			bgcolor = 0; max_pins = 0; max_pins_depth = static_cast<uint64_t>(sz);
			for (const auto& kv : all_pins) {
				const uint64_t np = n_chosen[kv.second], d = depth_sum[kv.second];
				if (np > max_pins) { bgcolor = kv.first; max_pins = np; max_pins_depth = d; }
				else if (np == max_pins && d > max_pins_depth) { bgcolor = kv.first; max_pins_depth = d; }
			}
		}
	}
	if (stored_width < 8) bgcolor &= (1ull << (8 * stored_width)) - 1;

	mark("bgcolor");
	// ---- encode_condensed_pins (src/labels.hpp:192-344) ----
	uint64_t max_pins = 0, max_depth = 0;
	std::vector<std::pair<uint64_t, uint32_t>>& all_labels = slot_labels;      // every label but the background colour, ascending
	{
		size_t keep = 0;
		for (size_t i = 0; i < all_labels.size(); i++) if (all_labels[i].first != bgcolor) all_labels[keep++] = all_labels[i];
		all_labels.resize(keep);
	}
	for (const auto& kv : all_labels) {
		max_pins = std::max<uint64_t>(max_pins, n_chosen[kv.second]);
		max_depth = std::max<uint64_t>(max_depth, depth_max[kv.second]);
	}
	parallel_sort(all_labels.begin(), all_labels.end(), [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; });

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
	bin.reserve(bin.size() + all_labels.size() * static_cast<size_t>(stored_width) + static_cast<size_t>(sz) * component_width + 1);
	for (const auto& kv : all_labels) put_le(bin, kv.first, stored_width);
	for (int64_t z = 0; z < sz; z++) put_le(bin, ncomp[z], component_width);
	bin.push_back(combined);

	// the labels' records are independent: sized first, then written side by side at their places
	const size_t n_rec = all_labels.size();
	std::vector<uint64_t> rec_at(n_rec + 1, 0);
	parallel_for(n_rec, 256, [&](size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; i++) {
			const uint32_t li = all_labels[i].second;
			const uint32_t* v = chosen.data() + li_at[li];
			uint64_t n_repr = 0, n_ids = 0;
			for (uint32_t k = 0; k < n_chosen[li]; k++) {
				const uint32_t p = v[k];
				if (depth_of(p) >= cc_efficient_threshold) n_repr++;
				else n_ids += ids_off[p + 1] - ids_off[p];
			}
			rec_at[i + 1] = 2ull * num_pins_width + n_repr * static_cast<uint64_t>(index_width + depth_width) + n_ids * static_cast<uint64_t>(cc_label_width);
		}
	});
	for (size_t i = 0; i < n_rec; i++) rec_at[i + 1] += rec_at[i];
	const size_t head_bytes = bin.size();
	bin.resize(head_bytes + rec_at[n_rec]);
	struct Sorted { uint64_t idx, depth; uint32_t pin; };
	auto put_at = [](uint8_t*& o, uint64_t v, int w) { for (int b = 0; b < w; b++) *o++ = static_cast<uint8_t>(v >> (8 * b)); };
	parallel_for(n_rec, 64, [&](size_t lo, size_t hi) {
		std::vector<Sorted> sp;
		std::vector<uint32_t> idv;
		for (size_t i = lo; i < hi; i++) {
			uint8_t* o = bin.data() + head_bytes + rec_at[i];
			const uint32_t li = all_labels[i].second;
			const uint32_t* v = chosen.data() + li_at[li];
			sp.clear();
			for (uint32_t k = 0; k < n_chosen[li]; k++) {
				const uint32_t p = v[k];
				sp.push_back({ static_cast<uint64_t>(pin_x[p]) + static_cast<uint64_t>(sx) * (static_cast<uint64_t>(pin_y[p]) + static_cast<uint64_t>(sy) * pin_zs[p]), depth_of(p), p });
			}
			std::sort(sp.begin(), sp.end(), [](const Sorted& a, const Sorted& b) { return a.idx < b.idx; });
			uint64_t n_pin_repr = 0;
			for (const Sorted& s : sp) n_pin_repr += (s.depth >= cc_efficient_threshold);
			put_at(o, n_pin_repr, num_pins_width);
			uint64_t prev = 0;
			bool first = true;
			for (const Sorted& s : sp) {
				if (s.depth < cc_efficient_threshold) continue;
				put_at(o, first ? s.idx : s.idx - prev, index_width);
				prev = s.idx;
				first = false;
			}
			for (const Sorted& s : sp) if (s.depth >= cc_efficient_threshold) put_at(o, s.depth, depth_width);
			idv.clear();
			for (const Sorted& s : sp) {
				if (s.depth >= cc_efficient_threshold) continue;
				for (uint64_t k = ids_off[s.pin]; k < ids_off[s.pin + 1]; k++) idv.push_back(ids[k]);
			}
			std::sort(idv.begin(), idv.end());
			put_at(o, idv.size(), num_pins_width);
			for (size_t k = 0; k < idv.size(); k++) put_at(o, k ? static_cast<uint32_t>(idv[k] - idv[k - 1]) : idv[k], cc_label_width);
			if (o != bin.data() + head_bytes + rec_at[i + 1]) throw Error(CKL_ERR_RUNTIME, "crackle_amd: internal: pin record size");
		}
	});
	mark("section");
	if (prof) fprintf(stderr, "[ckl pins cover ms]%s | components=%llu labels=%zu pins=%zu\n", marks.c_str(), static_cast<unsigned long long>(N), n_labels, P);
	return bin;
}

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
		PinCandidates pc;
		if (dtype_bytes == 1) pc = pin_candidates_host<uint8_t>(static_cast<const uint8_t*>(labels), cc, sx, sy, sz, total);
		else if (dtype_bytes == 2) pc = pin_candidates_host<uint16_t>(static_cast<const uint16_t*>(labels), cc, sx, sy, sz, total);
		else if (dtype_bytes == 4) pc = pin_candidates_host<uint32_t>(static_cast<const uint32_t*>(labels), cc, sx, sy, sz, total);
		else if (dtype_bytes == 8) pc = pin_candidates_host<uint64_t>(static_cast<const uint64_t*>(labels), cc, sx, sy, sz, total);
		else throw Error(CKL_ERR_ARG, "crackle_amd: dtype width must be 1, 2, 4 or 8 bytes");
		const std::vector<uint8_t> bin = pins_cover_host(pc, sx, sy, sz, nc, total, h.pin_index_width(), stored_width, auto_bgcolor != 0, manual_bgcolor);
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
