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
//                  (labels of column x-1 / x, no clearing) and marks the kept runs in a
//                  uint16 volume at (x, y, z_start) with their depth + 1.
//   k_pin_columns<0>  per component: first column run (any) and first KEPT run containing it,
//                  in traversal order (y, x, z_start)         — atomicMin on 64-bit keys
//   k_pin_extent<depth>  depth of that first kept run (one thread per component)
//   k_pin_columns<2>  last kept run deeper than the first     — atomicMax
//   k_pin_choice   the run find_suboptimal_pins takes per component (src/pins.hpp:325-340)
//   k_pin_extent / k_pin_ids   z-range and component ids of the distinct chosen runs
//
// The two per-column passes read labels + ids + kept marks coalesced along x
// (~10 B per voxel each); the dedup pass is latency bound (sx * runs-per-column dependent
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
	uint32_t sx, sy, sz;     // sy: the rows held here (all of them, or one rank's rows [y0, y0 + sy) of the row-sharded stage)
	uint64_t sxy;            // sx * sy: voxels between two slices of the volume as it is held
	const uint32_t* cc;      // global component ids
	uint16_t* mark;          // per voxel: depth + 1 of the kept candidate pin that starts here, else 0 (sz <= 65535)
	uint64_t key_col0 = 0;   // y0 * sx: keys name columns of the WHOLE volume ((y * sx + x) * sz + z_start), whatever rows are held
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
							v.mark[col - 1u + v.sxy * lz_s] = 0;
						}
						break;
					}
					p = (p + 1u) & mask;
				}
			}
			if (keep) {
				v.mark[col + v.sxy * z_s] = static_cast<uint16_t>(z_e - z_s + 1u);
				cur[slot].label = L; cur[slot].gen = gen; cur[slot].z_s = z_s; cur[slot].z_e = z_e;
			}
			label = next;
			z_s = z;
		}
	}
}

// ---- k_pin_dedup, one wavefront per row (sz <= 64 * K) ----
// Lane l holds the labels of slices 64 k + l of the current column (loaded one column ahead).  A run lives in the
// lane of its LAST slice: a ballot of "the label changes above me" marks the runs, the highest such lane below
// gives a run its first slice.  add_pin (src/pins.hpp:134-160) compares a run with its label's last pin in the
// previous column: the previous column's kept runs are broadcast one by one (readlane) and every lane compares
// its own label — all runs of the column are decided together, ~30 broadcasts per column instead of a dependent
// chain of table lookups per run (the first version walked the runs one at a time: 42.6 ms for 2048 x 2048 x 256,
// the wavefronts of 2048 rows being all the parallelism there is).  A label with several runs in one column —
// rare — is put right afterwards in run order: once one of its runs is kept the later ones are plain appends.
// No LDS, no hashing; the stores are the kept marks, one lane per run.
template <typename LABEL>
__device__ __forceinline__ LABEL wave_read(LABEL v, uint32_t lane) {
	if constexpr (sizeof(LABEL) == 8) {
		const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), lane);
		const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), lane);
		return static_cast<LABEL>((static_cast<uint64_t>(hi) << 32) | lo);
	}
	else return static_cast<LABEL>(__builtin_amdgcn_readlane(static_cast<uint32_t>(v), lane));
}

constexpr uint32_t kPinWaves = 4;      // rows per workgroup
constexpr uint32_t kPinDupBits = 12;   // slots of a wavefront's duplicate-label table (4096 x 2 bytes)

// grid = ceil(sy / 4) x 256
// G: columns fetched by one load (4 when the rows allow it: labels aligned to 4 of them and sx a multiple of 4).
// A lane's loads of one column lie in 64 K different slices, megabytes apart: with G = 4 a quarter of the requests.
template <typename LABEL, int K, int G>
__global__ void __launch_bounds__(64 * kPinWaves) k_pin_dedup_wave(const LABEL* __restrict__ labels, PinVolume v) {
	struct alignas(G * sizeof(LABEL)) Group { LABEL c[G]; };
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t y = blockIdx.x * kPinWaves + (threadIdx.x >> 6);
	if (y >= v.sy) return;
	const uint64_t row = static_cast<uint64_t>(y) * v.sx;
	const unsigned long long below_me = (1ull << lane) - 1ull;
	__shared__ uint16_t s_dup[kPinWaves << kPinDupBits];
	uint16_t* dup_table = s_dup + ((threadIdx.x >> 6) << kPinDupBits);
	LABEL lab[K], prev_lab[K];
	Group grp[K], ngrp[K];
	uint32_t prev_zz[K];                       // z_s | z_e << 16 of the run that ends in this lane
	unsigned long long prev_kept[K];           // lanes of the previous column's kept runs (uniform)
#pragma unroll
	for (int k = 0; k < K; k++) {
		ngrp[k] = *reinterpret_cast<const Group*>(labels + row + v.sxy * min(static_cast<uint32_t>(k) * 64u + lane, v.sz - 1u));
		prev_lab[k] = 0; prev_zz[k] = 0; prev_kept[k] = 0;
	}
	for (uint32_t x0 = 0; x0 < v.sx; x0 += G) {
#pragma unroll
		for (int k = 0; k < K; k++) grp[k] = ngrp[k];
		{
			const uint64_t ncol = row + min(x0 + G, v.sx - G);      // (the last group again: not used)
#pragma unroll
			for (int k = 0; k < K; k++) ngrp[k] = *reinterpret_cast<const Group*>(labels + ncol + v.sxy * min(static_cast<uint32_t>(k) * 64u + lane, v.sz - 1u));
		}
#pragma unroll
		for (int g = 0; g < G; g++) {
		const uint32_t x = x0 + g;
		if (x >= v.sx) break;
		const uint64_t col = row + x;
#pragma unroll
		for (int k = 0; k < K; k++) lab[k] = grp[k].c[g];
		// the column's runs
		unsigned long long ends_m[K];
		uint32_t zz[K];
		bool ends[K];
		int carry_end = -1;
#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint32_t z = static_cast<uint32_t>(k) * 64u + lane;
			// label of the slice above: the next lane, or lane 0 of the next register
			LABEL up = __shfl_down(lab[k], 1);
			if (k + 1 < K) { const LABEL first_up = wave_read(lab[k + 1 < K ? k + 1 : k], 0); if (lane == 63u) up = first_up; }
			ends[k] = z < v.sz && (z == v.sz - 1u || up != lab[k]);
			const unsigned long long m = __ballot(ends[k]);
			ends_m[k] = m;
			const unsigned long long below = m & below_me;
			const int prev_end = below ? k * 64 + 63 - __clzll(static_cast<long long>(below)) : carry_end;
			zz[k] = static_cast<uint32_t>(prev_end + 1) | (z << 16);
			if (m) carry_end = k * 64 + 63 - __clzll(static_cast<long long>(m));
		}
		// the label's last pin of the previous column, if it has one there (kPinNoRun: none; a run's word is z_s | z_e << 16
		// with z_e < 1024)
		constexpr uint32_t kPinNoRun = 0xFFFFFFFFu;
		uint32_t found[K];
#pragma unroll
		for (int k = 0; k < K; k++) found[k] = kPinNoRun;
#pragma unroll
		for (int kp = 0; kp < K; kp++) {
			unsigned long long pm = prev_kept[kp];
			while (pm) {
				const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(pm))) - 1u;
				pm &= pm - 1ull;
				const LABEL PL = wave_read(prev_lab[kp], b);
				const uint32_t pzz = __builtin_amdgcn_readlane(prev_zz[kp], b);
#pragma unroll
				for (int k = 0; k < K; k++) found[k] = lab[k] == PL ? pzz : found[k];
			}
		}
		// decisions as lane masks (uniform): kept runs, runs that take their neighbour's place
		unsigned long long keepm[K], replm[K];
		bool maybe_dup = false;
#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint32_t z_s = zz[k] & 0xFFFFu, z_e = zz[k] >> 16, lz_s = found[k] & 0xFFFFu, lz_e = found[k] >> 16;
			const bool has = found[k] != kPinNoRun;
			const bool covered = has && lz_s <= z_s && lz_e >= z_e;                   // covered by the neighbour: dropped
			keepm[k] = __ballot(ends[k] && !covered);
			replm[k] = __ballot(ends[k] && !covered && has && lz_s >= z_s && lz_e <= z_e);      // covers the neighbour: takes its place
			// does a label have several runs in this column?  Every run leaves its place in a hashed table and
			// looks at what stays there: two runs of one label meet in one slot, and one of them sees the other
			// (atomic accesses + fences: the exchange is between LANES of the wavefront; with plain accesses and a
			// wave barrier — which the compiler models as touching no memory — a lane's own store may be forwarded to
			// its own load below and `other` folds to false)
			if (ends[k]) __hip_atomic_store(dup_table + (pin_hash(static_cast<uint64_t>(lab[k])) >> (32 - kPinDupBits)), static_cast<uint16_t>(k * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
		for (int k = 0; k < K; k++) {
			const bool other = ends[k] && __hip_atomic_load(dup_table + (pin_hash(static_cast<uint64_t>(lab[k])) >> (32 - kPinDupBits)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) != static_cast<uint16_t>(k * 64 + lane);
			maybe_dup = maybe_dup || __ballot(other) != 0ull;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// a label with several runs in this column (or two labels that met in the table): once one of the label's runs
		// is kept it is the label's last pin, in THIS column, and the later ones are appended without a look at the neighbour
		if (maybe_dup) {
#pragma unroll
			for (int kq = 0; kq < K; kq++) {
				unsigned long long qm = ends_m[kq];
				while (qm) {
					const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(qm))) - 1u;
					qm &= qm - 1ull;
					if (!((keepm[kq] >> b) & 1ull)) continue;
					const LABEL QL = wave_read(lab[kq], b);
#pragma unroll
					for (int k = 0; k < K; k++) {
						if (k < kq) continue;
						const unsigned long long later = k > kq ? ~0ull : (b == 63u ? 0ull : ~0ull << (b + 1u));
						const unsigned long long same = __ballot(lab[k] == QL) & ends_m[k] & later;
						keepm[k] |= same;
						replm[k] &= ~same;
					}
				}
			}
		}
		bool keep[K], repl[K];
#pragma unroll
		for (int k = 0; k < K; k++) { keep[k] = (keepm[k] >> lane) & 1ull; repl[k] = (replm[k] >> lane) & 1ull; }
#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint32_t z_s = zz[k] & 0xFFFFu, z_e = zz[k] >> 16;
			if (repl[k]) v.mark[col - 1u + v.sxy * (found[k] & 0xFFFFu)] = 0;
			if (keep[k]) v.mark[col + v.sxy * z_s] = static_cast<uint16_t>(z_e - z_s + 1u);
			prev_kept[k] = keepm[k];
			prev_lab[k] = lab[k];
			prev_zz[k] = zz[k];
		}
		}
	}
}

struct PinComponentArrays {
	unsigned long long* first_any;      // [N] smallest key of a run starting in the component
	unsigned long long* first_kept;     // [N] smallest key of a kept run containing it
	uint32_t* first_depth;              // [N]
	unsigned long long* best;           // [N] 1 + largest key of a kept run deeper than the first (0: none)
};

// Of the lanes with `want` set, is this one the first (LAST: the last) that names component c?
// Called by all active lanes of the wavefront.
template <bool LAST>
__device__ __forceinline__ bool pin_wave_leader(bool want, uint32_t c) {
	const unsigned long long m = __ballot(want);
	if (m == 0ull) return false;      // (uniform: most voxels update nothing, and the exchange below goes through the LDS crossbar)
	const uint32_t lane = threadIdx.x & 63u;
	const unsigned long long others = LAST ? (lane == 63u ? 0ull : m & (~0ull << (lane + 1u))) : m & ((1ull << lane) - 1ull);
	// the nearest wanting lane on that side (any lane when there is none: its answer is not used)
	const uint32_t nb = others ? (LAST ? static_cast<uint32_t>(__ffsll(static_cast<long long>(others))) - 1u : 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(others)))) : lane;
	const uint32_t c_nb = static_cast<uint32_t>(__shfl(static_cast<int>(c), static_cast<int>(nb)));
	return want && !(others && c_nb == c);
}

// One thread per (x, y) column walks z once, eight slices at a time: labels, component ids and kept
// marks of the eight are loaded together, then the per-component values they will be compared with
// (independent gathers), then the comparisons; the depth of a kept run comes with its mark, so the
// run's end need not be known.  PASS 0: first run / first kept run of every component (atomicMin on
// keys); PASS 2: last kept run deeper than the first (atomicMax; the depth of the first comes from
// k_pin_extent in between).  A value gathered before another thread's atomic is only ever larger
// (PASS 0) or smaller (PASS 2) than the current one: the filter in front of the atomics stays safe.
// grid = ceil(sx / 256) x sy
constexpr uint32_t kPinChunk = 8;
template <typename LABEL, int PASS>
__global__ void __launch_bounds__(kPinBlock) k_pin_columns(const LABEL* __restrict__ labels, PinVolume v, PinComponentArrays a) {
	const uint32_t x = blockIdx.x * kPinBlock + threadIdx.x;
	const uint32_t y = blockIdx.y;
	if (x >= v.sx) return;
	const uint64_t col = static_cast<uint64_t>(y) * v.sx + x;
	LABEL prev = 0;
	bool kept = false;
	uint32_t depth = 0;
	unsigned long long key = 0;
	for (uint32_t z0 = 0; z0 < v.sz; z0 += kPinChunk) {
		LABEL lab[kPinChunk];
		uint32_t cc[kPinChunk], mk[kPinChunk];
		unsigned long long g0[kPinChunk];
		uint32_t g1[kPinChunk];
#pragma unroll
		for (uint32_t i = 0; i < kPinChunk; i++) {
			const uint64_t at = col + v.sxy * min(z0 + i, v.sz - 1u);
			lab[i] = labels[at]; cc[i] = v.cc[at]; mk[i] = v.mark[at];
		}
#pragma unroll
		for (uint32_t i = 0; i < kPinChunk; i++) {
			if (PASS == 0) { g0[i] = a.first_kept[cc[i]]; g1[i] = 0; }
			else { g0[i] = a.best[cc[i]]; g1[i] = a.first_depth[cc[i]]; }
		}
#pragma unroll
		for (uint32_t i = 0; i < kPinChunk; i++) {
			const uint32_t z = z0 + i;
			if (z >= v.sz) break;
			const bool start = z == 0 || lab[i] != prev;
			if (start) {
				key = static_cast<unsigned long long>(col + v.key_col0) * v.sz + z;
				kept = mk[i] != 0u;
				depth = mk[i] - 1u;
			}
			prev = lab[i];
			// A component spans many columns of this row: of the lanes that want to update the same
			// component only the one with the smallest (PASS 0) / largest (PASS 2) key goes to memory —
			// keys grow with x, so that is the first / last such lane of the wavefront.
			if (PASS == 0) {
				const bool want_any = start && key < a.first_any[cc[i]];
				if (pin_wave_leader<false>(want_any, cc[i])) atomicMin(a.first_any + cc[i], key);
				const bool want = kept && key < g0[i];
				if (pin_wave_leader<false>(want, cc[i])) atomicMin(a.first_kept + cc[i], key);
			}
			else {
				const bool want = kept && depth > g1[i] && key + 1ull > g0[i];
				if (pin_wave_leader<true>(want, cc[i])) atomicMax(a.best + cc[i], key + 1ull);
			}
		}
	}
}

__global__ void __launch_bounds__(kPinBlock) k_pin_choice(PinComponentArrays a, uint64_t n, unsigned long long* __restrict__ choice) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long b = a.best[c];
	choice[c] = b ? b - 1ull : a.first_kept[c];
}

// label of every component from the two volumes (the sharded encoder's whole-volume stage has no
// run tables): read where the id changes along x.  Grid-stride over the voxels.
template <typename LABEL>
__global__ void __launch_bounds__(kPinBlock) k_pin_component_labels(
	const LABEL* __restrict__ labels, const uint32_t* __restrict__ cc, uint64_t voxels, uint32_t sx, uint64_t n,
	unsigned long long* __restrict__ comp_label, uint32_t* __restrict__ err
) {
	for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x; i < voxels; i += static_cast<uint64_t>(gridDim.x) * kPinBlock) {
		const uint32_t c = cc[i];
		if (c >= n) { *err = 1u; continue; }
		if (i % sx == 0 || cc[i - 1] != c) comp_label[c] = static_cast<unsigned long long>(labels[i]);
	}
}

// z-range of a run given by its key, one thread per key: the last slice (the distinct chosen
// runs) or the depth (the first kept run of every component; a component without one keeps 0)
// PLUS1 (row-sharded stage): last slice + 1, so that 0 says "not my rows" and the ranks' arrays merge by maximum
template <typename LABEL, bool DEPTH, bool PLUS1 = false>
__global__ void __launch_bounds__(kPinBlock) k_pin_extent(const LABEL* __restrict__ labels, PinVolume v, const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * kPinBlock + threadIdx.x;
	if (i >= n) return;
	const unsigned long long key = keys[i];
	if (key == kPinNoKey) return;
	const uint32_t z_s = static_cast<uint32_t>(key % v.sz);
	const uint64_t gcol = key / v.sz;
	if (gcol < v.key_col0 || gcol - v.key_col0 >= v.sxy) return;      // a run of another rank's rows
	const uint64_t col = gcol - v.key_col0;
	const LABEL label = labels[col + v.sxy * z_s];
	uint32_t z = z_s + 1u;
	while (z < v.sz && labels[col + v.sxy * z] == label) z++;
	out[i] = DEPTH ? z - 1u - z_s : (PLUS1 ? z : z - 1u);
}

__global__ void __launch_bounds__(kPinBlock) k_pin_ids(PinVolume v, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ z_e, const uint64_t* __restrict__ off, uint32_t n, uint32_t* __restrict__ ids) {
	const uint32_t i = blockIdx.x * kPinBlock + threadIdx.x;
	if (i >= n) return;
	const unsigned long long key = keys[i];
	if (key == kPinNoKey) return;
	const uint32_t z_s = static_cast<uint32_t>(key % v.sz);
	const uint64_t gcol = key / v.sz;
	if (gcol < v.key_col0 || gcol - v.key_col0 >= v.sxy) return;      // a run of another rank's rows (its ids stay 0 here)
	const uint64_t col = gcol - v.key_col0;
	uint32_t* dst = ids + off[i];
	for (uint32_t z = z_s; z <= z_e[i]; z++) dst[z - z_s] = v.cc[col + v.sxy * z];
}

// row-sharded stage: (first kept run, its depth) of every component in one word, so that the ranks' answers merge
// by minimum — key << 16 | depth (keys < 2^47: volumes of < 2^47 voxels; depth < 2^16) — and back
__global__ void __launch_bounds__(kPinBlock) k_pin_pack_first(const unsigned long long* __restrict__ first_kept, const uint32_t* __restrict__ depth, uint64_t n, unsigned long long* __restrict__ packed) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long k = first_kept[c];
	packed[c] = k == kPinNoKey ? kPinNoKey : ((k << 16) | (depth[c] & 0xFFFFu));
}
__global__ void __launch_bounds__(kPinBlock) k_pin_unpack_first(const unsigned long long* __restrict__ packed, uint64_t n, unsigned long long* __restrict__ first_kept, uint32_t* __restrict__ depth) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long p = packed[c];
	first_kept[c] = p == kPinNoKey ? kPinNoKey : (p >> 16);
	depth[c] = p == kPinNoKey ? 0u : static_cast<uint32_t>(p & 0xFFFFu);
}
__global__ void __launch_bounds__(kPinBlock) k_pin_minus1(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c < n) out[c] = in[c] ? in[c] - 1u : 0u;
}
// offsets of the components' id lists from their chosen runs (exclusive prefix by the host is avoided: one pass with a scan per block would do,
// but the lists are only needed on the rank that writes the section: it scans there)
__global__ void __launch_bounds__(kPinBlock) k_pin_id_counts(const unsigned long long* __restrict__ choice, const uint32_t* __restrict__ ze_plus1, uint32_t sz, uint64_t n, uint32_t* __restrict__ count) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long k = choice[c];
	count[c] = (k == kPinNoKey || ze_plus1[c] == 0u) ? 0u : ze_plus1[c] - static_cast<uint32_t>(k % sz);
}


// ---- exclusive prefix sums of the id counts (uint32 counts -> uint64 offsets, n + 1 entries) in three launches:
// totals of 2048-count pieces, their scan by one workgroup, the offsets
constexpr uint32_t kPinScanItems = 8;
constexpr uint32_t kPinScanPiece = kPinBlock * kPinScanItems;
__device__ __forceinline__ unsigned long long pin_block_scan(unsigned long long mine, unsigned long long* s_wave, unsigned long long& total) {
	// inclusive scan over the workgroup's threads; `total` = the workgroup's sum
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned long long v = mine;
	for (uint32_t d = 1; d < 64u; d <<= 1) {
		const unsigned long long o = __shfl_up(v, d);
		if (lane >= d) v += o;
	}
	if (lane == 63u) s_wave[wave] = v;
	__syncthreads();
	unsigned long long base = 0, all = 0;
	for (uint32_t w = 0; w < kPinBlock / 64u; w++) { if (w < wave) base += s_wave[w]; all += s_wave[w]; }
	__syncthreads();
	total = all;
	return v + base;
}
__global__ void __launch_bounds__(kPinBlock) k_pin_scan_pieces(const uint32_t* __restrict__ count, uint64_t n, unsigned long long* __restrict__ piece_sum) {
	__shared__ unsigned long long s_wave[kPinBlock / 64u];
	const uint64_t at = static_cast<uint64_t>(blockIdx.x) * kPinScanPiece + static_cast<uint64_t>(threadIdx.x) * kPinScanItems;
	unsigned long long mine = 0;
#pragma unroll
	for (uint32_t i = 0; i < kPinScanItems; i++) if (at + i < n) mine += count[at + i];
	unsigned long long total;
	(void)pin_block_scan(mine, s_wave, total);
	if (threadIdx.x == 0) piece_sum[blockIdx.x] = total;
}
// one workgroup: piece_sum[i] -> sum of the pieces in front of i; piece_sum[pieces] = everything
__global__ void __launch_bounds__(kPinBlock) k_pin_scan_tops(unsigned long long* __restrict__ piece_sum, uint32_t pieces) {
	__shared__ unsigned long long s_wave[kPinBlock / 64u];
	unsigned long long carry = 0;
	for (uint32_t i0 = 0; i0 < pieces; i0 += kPinBlock) {
		const uint32_t i = i0 + threadIdx.x;
		const unsigned long long mine = i < pieces ? piece_sum[i] : 0ull;
		unsigned long long total;
		const unsigned long long incl = pin_block_scan(mine, s_wave, total);
		if (i < pieces) piece_sum[i] = carry + incl - mine;
		carry += total;
	}
	if (threadIdx.x == 0) piece_sum[pieces] = carry;
}
__global__ void __launch_bounds__(kPinBlock) k_pin_scan_offsets(const uint32_t* __restrict__ count, uint64_t n, const unsigned long long* __restrict__ piece_sum, uint32_t pieces, unsigned long long* __restrict__ off /* [n + 1] */) {
	__shared__ unsigned long long s_wave[kPinBlock / 64u];
	const uint64_t at = static_cast<uint64_t>(blockIdx.x) * kPinScanPiece + static_cast<uint64_t>(threadIdx.x) * kPinScanItems;
	uint32_t c[kPinScanItems];
	unsigned long long mine = 0;
#pragma unroll
	for (uint32_t i = 0; i < kPinScanItems; i++) { c[i] = at + i < n ? count[at + i] : 0u; mine += c[i]; }
	unsigned long long total;
	unsigned long long run = piece_sum[blockIdx.x] + pin_block_scan(mine, s_wave, total) - mine;
#pragma unroll
	for (uint32_t i = 0; i < kPinScanItems; i++) { if (at + i < n) off[at + i] = run; run += c[i]; }
	if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = piece_sum[pieces];
}

// The order in which the labels enter `pinsets` (src/pins.hpp:126-163) is that of their first column
// run: per label the smallest first_any of its components, in an open-addressing table keyed by
// label (the all-ones label cannot be a key: its minimum goes to max_first), then the occupied
// slots are appended to a list (any order: the host sorts the few labels by that key).
__global__ void __launch_bounds__(kPinBlock) k_pin_label_first(
	const unsigned long long* __restrict__ comp_label, const unsigned long long* __restrict__ first_any, uint64_t n,
	unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals, uint32_t mask, unsigned long long* __restrict__ max_first
) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kPinBlock + threadIdx.x;
	if (c >= n) return;
	const unsigned long long label = comp_label[c], first = first_any[c];
	if (label == kPinNoKey) { atomicMin(max_first, first); return; }
	uint32_t h = pin_hash(label) & mask;
	for (uint32_t probe = 0; probe <= mask; probe++) {
		unsigned long long seen = keys[h];
		if (seen == kPinNoKey) seen = atomicCAS(keys + h, kPinNoKey, label);
		if (seen == kPinNoKey || seen == label) { if (first < vals[h]) atomicMin(vals + h, first); return; }
		h = (h + 1u) & mask;
	}
}
__global__ void __launch_bounds__(kPinBlock) k_pin_label_list(
	const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals, uint32_t slots,
	uint32_t* __restrict__ count, unsigned long long* __restrict__ out_label, unsigned long long* __restrict__ out_first
) {
	const uint32_t i = blockIdx.x * kPinBlock + threadIdx.x;
	if (i >= slots || keys[i] == kPinNoKey) return;
	const uint32_t at = atomicAdd(count, 1u);
	out_label[at] = keys[i]; out_first[at] = vals[i];
}

}  // namespace ckl
