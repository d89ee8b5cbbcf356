// Pin label encoding — device passes (see ckl_pins.hpp for what they produce).
//
// Input: the label volume and the global component id of every voxel (k_paint_components),
// both resident.  Nothing volumetric leaves the device.
//
//   k_pin_dedup    extract_columns + add_pin (src/pins.hpp:95-163).  A column run of label L is
//                  compared with the LAST pin of L's vector, and only when that pin sits in the
//                  previous column of the same row: rows are independent, and within a row the
//                  state is "the last kept run per label of column x-1 and of column x".  One
//                  thread per row walks x with two generation-stamped open-addressing tables
//                  (labels of column x-1 / x, no clearing) and marks the kept runs in a bit
//                  volume at (x, y, z_start).
//   k_pin_first    per component: first column run (any) and first KEPT run containing it, in
//                  traversal order (y, x, z_start)            — atomicMin on 64-bit keys
//   k_pin_depth    depth of that first kept run
//   k_pin_best     last kept run deeper than the first        — atomicMax
//   k_pin_choice   the run find_suboptimal_pins takes per component (src/pins.hpp:325-340)
//   k_pin_extent / k_pin_ids   z-range and component ids of the distinct chosen runs
//
// The three per-column passes read labels + ids + kept bits coalesced along x
// (~9 B per voxel each); the dedup pass is latency bound (sx * runs-per-column dependent
// steps per row) and reads the volume once through L2.
#pragma once
#include "ckl_device.hpp"
#include "ckl_pins.hpp"

namespace ckl {

constexpr uint32_t kPinBlock = 256;
constexpr uint32_t kPinRowBlock = 64;
constexpr uint32_t kPinAhead = 8;

struct PinSlot {
	uint64_t label;
	uint32_t gen;      // column + 1 the entry belongs to (0: never used)
	uint32_t z_s, z_e;
	uint32_t pad;
};

struct PinVolume {
	uint32_t sx, sy, sz;
	uint64_t sxy;
	const uint32_t* cc;      // global component ids
	uint32_t* kept;          // one bit per voxel: a kept candidate pin starts here
};

__device__ __forceinline__ uint32_t pin_hash(uint64_t label) {
	uint64_t h = label * 0x9E3779B97F4A7C15ull;
	return static_cast<uint32_t>(h >> 32);
}

// grid = ceil(sy / 64) x 64: one thread per row
template <typename LABEL>
__global__ void __launch_bounds__(kPinRowBlock) k_pin_dedup(const LABEL* __restrict__ labels, PinVolume v, PinSlot* __restrict__ tables, uint32_t cap) {
	const uint32_t y = blockIdx.x * kPinRowBlock + threadIdx.x;
	if (y >= v.sy) return;
	const uint32_t mask = cap - 1u;
	PinSlot* tab = tables + static_cast<uint64_t>(y) * 2u * cap;
	const uint64_t row = static_cast<uint64_t>(y) * v.sx;
	for (uint32_t x = 0; x < v.sx; x++) {
		PinSlot* cur = tab + static_cast<uint64_t>(x & 1u) * cap;
		const PinSlot* prev = tab + static_cast<uint64_t>((x & 1u) ^ 1u) * cap;
		const uint32_t gen = x + 1u;
		const uint64_t col = row + x;
		LABEL label = labels[col];
		uint32_t z_s = 0;
		LABEL ahead[kPinAhead];      // the column is read kPinAhead slices at a time: independent loads
		for (uint32_t z = 1; z <= v.sz; z++) {
			const uint32_t k = (z - 1u) % kPinAhead;
			if (k == 0) {
#pragma unroll
				for (uint32_t i = 0; i < kPinAhead; i++) ahead[i] = labels[col + v.sxy * min(z + i, v.sz - 1u)];
			}
			LABEL next = label;
			bool ends = true;
			if (z < v.sz) {
#pragma unroll
				for (uint32_t i = 0; i < kPinAhead; i++) if (i == k) next = ahead[i];
				ends = next != label;
			}
			if (!ends) continue;
			const uint32_t z_e = z - 1u;
			const uint64_t L = static_cast<uint64_t>(label);
			const uint32_t h = pin_hash(L) & mask;
			// an earlier run of this column is L's last pin: plain append
			uint32_t slot = h;
			bool in_cur = false;
			while (cur[slot].gen == gen) {
				if (cur[slot].label == L) { in_cur = true; break; }
				slot = (slot + 1u) & mask;
			}
			bool keep = true;
			if (!in_cur && x > 0) {
				uint32_t p = h;
				while (prev[p].gen == x) {
					if (prev[p].label == L) {
						const uint32_t lz_s = prev[p].z_s, lz_e = prev[p].z_e;
						if (lz_s <= z_s && lz_e >= z_e) keep = false;                 // covered by the neighbour: dropped
						else if (lz_s >= z_s && lz_e <= z_e) {                          // covers the neighbour: takes its place
							const uint64_t bit = col - 1u + v.sxy * lz_s;
							atomicAnd(v.kept + (bit >> 5), ~(1u << (bit & 31u)));
						}
						break;
					}
					p = (p + 1u) & mask;
				}
			}
			if (keep) {
				const uint64_t bit = col + v.sxy * z_s;
				atomicOr(v.kept + (bit >> 5), 1u << (bit & 31u));
				cur[slot].label = L; cur[slot].gen = gen; cur[slot].z_s = z_s; cur[slot].z_e = z_e;
			}
			label = next;
			z_s = z;
		}
	}
}

struct PinComponentArrays {
	unsigned long long* first_any;      // [N] smallest key of a run starting in the component
	unsigned long long* first_kept;     // [N] smallest key of a kept run containing it
	uint32_t* first_depth;              // [N]
	unsigned long long* best;           // [N] 1 + largest key of a kept run deeper than the first (0: none)
};

// One thread per (x, y) column walks z; `pass` 0: firsts, 1: depth of the first, 2: last deeper.
// grid = ceil(sx / 256) x sy
template <typename LABEL, int PASS>
__global__ void __launch_bounds__(kPinBlock) k_pin_columns(const LABEL* __restrict__ labels, PinVolume v, PinComponentArrays a) {
	const uint32_t x = blockIdx.x * kPinBlock + threadIdx.x;
	const uint32_t y = blockIdx.y;
	if (x >= v.sx) return;
	const uint64_t col = static_cast<uint64_t>(y) * v.sx + x;
	LABEL label = labels[col];
	uint32_t z_s = 0;
	for (uint32_t z = 1; z <= v.sz; z++) {
		LABEL next = label;
		bool ends = true;
		if (z < v.sz) { next = labels[col + v.sxy * z]; ends = next != label; }
		if (!ends) continue;
		const uint32_t z_e = z - 1u;
		const unsigned long long key = (static_cast<unsigned long long>(col)) * v.sz + z_s;
		const uint64_t bit = col + v.sxy * z_s;
		const bool kept = (v.kept[bit >> 5] >> (bit & 31u)) & 1u;
		if (PASS == 0) {
			const uint32_t c0 = v.cc[bit];
			if (key < a.first_any[c0]) atomicMin(a.first_any + c0, key);
		}
		if (kept) {
			const uint32_t depth = z_e - z_s;
			for (uint32_t zz = z_s; zz <= z_e; zz++) {
				const uint32_t c = v.cc[col + v.sxy * zz];
				if (PASS == 0) { if (key < a.first_kept[c]) atomicMin(a.first_kept + c, key); }
				else if (PASS == 1) { if (a.first_kept[c] == key) a.first_depth[c] = depth; }
				else { if (depth > a.first_depth[c] && key + 1ull > a.best[c]) atomicMax(a.best + c, key + 1ull); }
			}
		}
		label = next;
		z_s = z;
	}
}

__global__ void __launch_bounds__(kPinBlock) k_pin_choice(PinComponentArrays a, uint64_t n, unsigned long long* __restrict__ choice) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long b = a.best[c];
	choice[c] = b ? b - 1ull : a.first_kept[c];
}

// z-range of the distinct chosen runs: one thread per run
template <typename LABEL>
__global__ void __launch_bounds__(kPinBlock) k_pin_extent(const LABEL* __restrict__ labels, PinVolume v, const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ z_e_out) {
	const uint32_t i = blockIdx.x * kPinBlock + threadIdx.x;
	if (i >= n) return;
	const unsigned long long key = keys[i];
	const uint32_t z_s = static_cast<uint32_t>(key % v.sz);
	const uint64_t col = key / v.sz;
	const LABEL label = labels[col + v.sxy * z_s];
	uint32_t z = z_s + 1u;
	while (z < v.sz && labels[col + v.sxy * z] == label) z++;
	z_e_out[i] = z - 1u;
}

__global__ void __launch_bounds__(kPinBlock) k_pin_ids(PinVolume v, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ z_e, const uint64_t* __restrict__ off, uint32_t n, uint32_t* __restrict__ ids) {
	const uint32_t i = blockIdx.x * kPinBlock + threadIdx.x;
	if (i >= n) return;
	const unsigned long long key = keys[i];
	const uint32_t z_s = static_cast<uint32_t>(key % v.sz);
	const uint64_t col = key / v.sz;
	uint32_t* dst = ids + off[i];
	for (uint32_t z = z_s; z <= z_e[i]; z++) dst[z - z_s] = v.cc[col + v.sxy * z];
}

}  // namespace ckl
