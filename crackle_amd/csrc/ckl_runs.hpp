// Run-based 4-connected component labelling shared by decode (connectivity from the
// rasterised crack planes) and encode (planes built from label comparisons).
// Replaces cc3d::color_connectivity_graph / connected_components2d_4 + relabel
// (src/cc3d.hpp:114-369) and the crc32c of the component image
// (src/crackle.hpp:599-611, src/labels.hpp:81).
//
// planeV bit (x,y): a crack edge lies between pixels (x-1,y) and (x,y).
// planeH bit (x,y): a crack edge lies between pixels (x,y-1) and (x,y).
// flip = 1 (IMPERMISSIBLE): crack = boundary; flip = 0 (PERMISSIBLE): crack = connection.
#pragma once

#include "ckl_device.hpp"

#include <cstdlib>

namespace ckl {
namespace dev {

enum : uint32_t {
	ERR_BOC = 1u,          // beginning-of-chain index malformed
	ERR_RANGE = 2u,        // a move left the vertex grid
	ERR_CAPACITY = 4u,     // scratch capacity exceeded
	ERR_NCOMP = 8u,        // component count differs from the label section
	ERR_CRC = 16u,         // crc32c of the component image differs from the stored one
	ERR_LIST = 32u,        // a strip's record list overflowed (k_crack_records): not an error of the stream, the rasterising path takes over
};

// ------------------------------------------------------------------------------
// horizontal runs
// ------------------------------------------------------------------------------
// A run is a maximal stretch of horizontally connected pixels of one row.  Runs are
// numbered in raster order of their first pixel; word_base[w] = number of runs that
// start before 32-pixel word w of the slice, so the run of pixel (x, y) is
//   word_base[y, x>>5] + popcount(breaks(y, x>>5) & bits <= (x & 31)) - 1.
struct RunGeom {
	const uint32_t* planeV;
	const uint32_t* planeH;
	uint32_t row_words;
	uint64_t plane_words;
	uint32_t flip;          // 1 for IMPERMISSIBLE (a crack bit is a break)
	uint32_t sx, sy;
	__device__ __forceinline__ uint32_t valid_mask(uint32_t w) const {
		const uint32_t left = sx - w * 32u;
		return left >= 32u ? 0xFFFFFFFFu : ((1u << left) - 1u);
	}
	// bit x set: a run starts at pixel x of this word
	__device__ __forceinline__ uint32_t breaks(uint32_t zi, uint32_t y, uint32_t w) const {
		const uint32_t v = planeV[zi * plane_words + static_cast<uint64_t>(y) * row_words + w];
		uint32_t b = flip ? v : ~v;
		if (w == 0) b |= 1u;
		return b & valid_mask(w);
	}
	// bit x set: pixel x is connected to the pixel above it
	__device__ __forceinline__ uint32_t ups(uint32_t zi, uint32_t y, uint32_t w) const {
		if (y == 0) return 0u;
		const uint32_t h = planeH[zi * plane_words + static_cast<uint64_t>(y) * row_words + w];
		return (flip ? ~h : h) & valid_mask(w);
	}
	// the same from a plane word that is already loaded (the strip kernels issue all their loads
	// first, unconditionally, and interpret the words afterwards: a load inside a branch is
	// waited for inside the branch)
	__device__ __forceinline__ uint32_t breaks_of(uint32_t v, uint32_t w) const {
		uint32_t b = flip ? v : ~v;
		if (w == 0) b |= 1u;
		return b & valid_mask(w);
	}
	__device__ __forceinline__ uint32_t ups_of(uint32_t h, uint32_t w) const { return (flip ? ~h : h) & valid_mask(w); }
};
__device__ __forceinline__ uint32_t mask_le(uint32_t bit) { return bit >= 31u ? 0xFFFFFFFFu : ((2u << bit) - 1u); }

struct RunArrays {
	uint32_t* word_base;       // [nslices][plane_words]
	const uint64_t* rbase;     // per slice base into the run arrays
	const uint32_t* rcap;
	uint32_t* parent;          // union-find over runs (root = smallest run index)
	uint32_t* run_start;       // first pixel of the run (slice-linear)
	uint32_t* run_cc;          // component id of the run
	uint32_t* comp_pix = nullptr;      // optional, [at rbase, one entry per component]: first pixel of the component (its root run's start)
	uint32_t* nruns;           // [nslices]
	uint32_t* ncomp;           // [nslices]
	uint32_t* slice_err;
};

constexpr int kIndexBlock = 1024;     // one workgroup per slice: as many threads as a workgroup can have

// grid = nslices, block = kIndexBlock.  The first pixels of a step's runs are collected in LDS
// and written out as one contiguous stretch (each thread owns a variable number of them: written
// straight from the threads they would land a few bytes per cache line and store instruction).
constexpr uint32_t kIndexStage = 8192;     // runs of one step (4096 plane words) staged in LDS; denser steps write directly
static __global__ void __launch_bounds__(kIndexBlock) k_run_index(RunGeom g, RunArrays r) {
	__shared__ uint32_t s_scan[kIndexBlock / kWave];
	__shared__ uint32_t s_runs[kIndexStage];
	const uint32_t zi = blockIdx.x;
	uint32_t* wb = r.word_base + zi * g.plane_words;
	uint32_t* parent = r.parent + r.rbase[zi];
	uint32_t* run_start = r.run_start + r.rbase[zi];
	const uint32_t cap = r.rcap[zi];
	const uint32_t words = static_cast<uint32_t>(g.plane_words);
	constexpr uint32_t kPer = 4;
	uint32_t carry = 0, err = 0;
	for (uint32_t w0 = 0; w0 < words; w0 += kIndexBlock * kPer) {
		uint32_t b[kPer], cnt = 0;
#pragma unroll
		for (uint32_t j = 0; j < kPer; j++) {
			const uint32_t wi = w0 + threadIdx.x * kPer + j;
			b[j] = 0;
			if (wi < words) {
				const uint32_t y = wi / g.row_words;
				b[j] = g.breaks(zi, y, wi - y * g.row_words);
			}
			cnt += __popc(b[j]);
		}
		uint32_t v[1] = { cnt }, tot[1];
		block_excl_add<1, kIndexBlock / kWave>(v, tot, s_scan);
		const bool staged = tot[0] <= kIndexStage;
		uint32_t local = v[0];
#pragma unroll
		for (uint32_t j = 0; j < kPer; j++) {
			const uint32_t wi = w0 + threadIdx.x * kPer + j;
			if (wi >= words) break;
			wb[wi] = carry + local;
			const uint32_t y = wi / g.row_words;
			const uint32_t x0 = (wi - y * g.row_words) * 32u;
			for (uint32_t m = b[j]; m; m &= m - 1u) {
				const uint32_t bit = __ffs(m) - 1;
				const uint32_t start = y * g.sx + x0 + bit;
				if (staged) s_runs[local] = start;
				else if (carry + local < cap) { run_start[carry + local] = start; parent[carry + local] = carry + local; }
				else err = ERR_CAPACITY;
				local++;
			}
		}
		if (staged) {
			__syncthreads();
			for (uint32_t i = threadIdx.x; i < tot[0]; i += kIndexBlock) {
				const uint32_t at = carry + i;
				if (at < cap) { run_start[at] = s_runs[i]; parent[at] = at; }
				else err = ERR_CAPACITY;
			}
			__syncthreads();
		}
		carry += tot[0];
	}
	if (threadIdx.x == 0) r.nruns[zi] = carry < cap ? carry : cap;
	if (err) atomicOr(r.slice_err + zi, err);
}

__device__ __forceinline__ uint32_t run_find(uint32_t* L, uint32_t a) {
	// path halving; parents only ever decrease, so racing writers stay consistent
	uint32_t p = uf_load(L, a);
	while (p != a) {
		const uint32_t gp = uf_load(L, p);
		if (gp != p) atomicMin(L + a, gp);
		a = p;
		p = gp;
	}
	return a;
}
__device__ __forceinline__ void run_unite(uint32_t* L, uint32_t a, uint32_t b) {
	for (;;) {
		a = run_find(L, a);
		b = run_find(L, b);
		if (a == b) return;
		if (a > b) { const uint32_t t = a; a = b; b = t; }
		const uint32_t old = atomicMin(L + b, a);
		if (old == b) return;
		b = old;
	}
}

// ---- strip-local union-find in LDS ------------------------------------------------
// Most unions are local: a strip of rows is united entirely in LDS (parents relative to
// the strip's first run), flattened there and written out as global parents, so that
// the global forest only ever sees trees of depth one plus the few unions across strip
// seams (k_run_union_seams).  A strip with more runs than the LDS table holds falls back
// to the global forest for its own rows.
constexpr uint32_t kStripRuns = 3072;       // LDS table: 12 KiB, so that the thread count and not the LDS bounds the workgroups per CU

// union-find in LDS with relaxed workgroup-scope atomics instead of volatile accesses: hipcc keeps
// volatile accesses on flat pointers (the address-space inference skips them), which costs a
// flat instruction per access and miscompiles on the dynamic LDS base (ROCm 7.2)
__device__ __forceinline__ uint32_t sm_load(const uint32_t* L, uint32_t i) { return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sm_store(uint32_t* L, uint32_t i, uint32_t v) { __hip_atomic_store(L + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t sm_find(uint32_t* L, uint32_t a) {
	uint32_t p = sm_load(L, a);
	while (p != a) {
		const uint32_t gp = sm_load(L, p);
		if (gp != p) sm_store(L, a, gp);     // path halving; a racing writer only ever stores an ancestor
		a = p;
		p = gp;
	}
	return a;
}
__device__ __forceinline__ void sm_unite(uint32_t* L, uint32_t a, uint32_t b) {
	for (;;) {
		a = sm_find(L, a);
		b = sm_find(L, b);
		if (a == b) return;
		if (a > b) { const uint32_t t = a; a = b; b = t; }
		const uint32_t old = atomicMin(L + b, a);
		if (old == b) return;
		b = old;
	}
}

__device__ __forceinline__ uint32_t lds_find(uint32_t* L, uint32_t a) { return sm_find(L, a); }
__device__ __forceinline__ void lds_unite(uint32_t* L, uint32_t a, uint32_t b) { sm_unite(L, a, b); }

// grid = (strips per slice, nslices); strip_rows rows per strip
static __global__ void __launch_bounds__(kBlock) k_run_union_strips(RunGeom g, RunArrays r, uint32_t strip_rows, uint32_t strip_runs) {
	extern __shared__ uint32_t s_parent[];      // strip_runs entries
	const uint32_t zi = blockIdx.y;
	const uint32_t y0 = blockIdx.x * strip_rows;
	if (y0 >= g.sy) return;
	const uint32_t y1 = min(y0 + strip_rows, g.sy);
	const uint32_t* wb = r.word_base + zi * g.plane_words;
	const uint32_t n = r.nruns[zi];
	const uint32_t base0 = min(wb[static_cast<uint64_t>(y0) * g.row_words], n);
	const uint32_t base1 = (y1 < g.sy) ? min(wb[static_cast<uint64_t>(y1) * g.row_words], n) : n;
	const uint32_t nloc = base1 - base0;
	uint32_t* parent = r.parent + r.rbase[zi];
	const bool local = nloc <= strip_runs;
	if (local) for (uint32_t j = threadIdx.x; j < nloc; j += kBlock) s_parent[j] = j;
	else for (uint32_t j = threadIdx.x; j < nloc; j += kBlock) parent[base0 + j] = base0 + j;
	__syncthreads();
	if (!local) __threadfence();
	const uint32_t w_end = (y1 - y0) * g.row_words;
	// the plane words of kUnionBatch steps are loaded before the first union: the unions are a
	// serial, divergent walk in LDS and would otherwise wait for memory once per word
	constexpr uint32_t kUnionBatch = 4;
	for (uint32_t wl0 = g.row_words + threadIdx.x; wl0 < w_end; wl0 += kBlock * kUnionBatch) {   // rows y0+1 .. y1-1
		uint32_t cand[kUnionBatch], b_here[kUnionBatch], b_up[kUnionBatch], base_here[kUnionBatch], base_up[kUnionBatch];
#pragma unroll
		for (uint32_t k = 0; k < kUnionBatch; k++) {
			const uint32_t wl = wl0 + k * kBlock;
			const bool in = wl < w_end;
			const uint32_t yl = in ? wl / g.row_words : 1u;
			const uint32_t w = in ? wl - yl * g.row_words : 0u;
			const uint32_t y = y0 + yl;
			const uint32_t up = in ? g.ups(zi, y, w) : 0u;
			const uint32_t prev_bit = (in && w) ? (g.ups(zi, y, w - 1) >> 31) : 0u;
			b_here[k] = g.breaks(zi, y, w);
			b_up[k] = g.breaks(zi, y - 1, w);
			cand[k] = up & (~((up << 1) | prev_bit) | b_here[k] | b_up[k]);
			const uint64_t wi = static_cast<uint64_t>(y) * g.row_words + w;
			base_here[k] = wb[wi]; base_up[k] = wb[wi - g.row_words];
		}
#pragma unroll
		for (uint32_t k = 0; k < kUnionBatch; k++) {
			for (uint32_t c = cand[k]; c; c &= c - 1u) {
				const uint32_t bit = __ffs(c) - 1;
				const uint32_t m = mask_le(bit);
				const uint32_t ra = base_here[k] + __popc(b_here[k] & m) - 1u;
				const uint32_t rb = base_up[k] + __popc(b_up[k] & m) - 1u;
				if (ra >= base1 || rb >= base1 || rb < base0) continue;      // capacity overflow upstream (flagged there)
				if (local) lds_unite(s_parent, ra - base0, rb - base0);
				else run_unite(parent, ra, rb);
			}
		}
	}
	if (!local) return;
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < nloc; j += kBlock) parent[base0 + j] = base0 + lds_find(s_parent, j);
}

// grid = (ceil(seams * row_words / 256), nslices): the unions across strip seams
// (rows y = k * strip_rows, k >= 1) on the global forest
static __global__ void __launch_bounds__(kBlock) k_run_union_seams(RunGeom g, RunArrays r, uint32_t strip_rows) {
	const uint32_t zi = blockIdx.y;
	const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t seam = t / g.row_words;
	const uint32_t w = t - seam * g.row_words;
	const uint32_t y = (seam + 1u) * strip_rows;
	if (y >= g.sy) return;
	const uint32_t up = g.ups(zi, y, w);
	if (!up) return;
	const uint32_t prev_bit = w ? (g.ups(zi, y, w - 1) >> 31) : 0u;
	const uint32_t b_here = g.breaks(zi, y, w);
	const uint32_t b_up = g.breaks(zi, y - 1, w);
	uint32_t cand = up & (~((up << 1) | prev_bit) | b_here | b_up);
	const uint32_t* wb = r.word_base + zi * g.plane_words;
	const uint64_t wi = static_cast<uint64_t>(y) * g.row_words + w;
	const uint32_t base_here = wb[wi], base_up = wb[wi - g.row_words];
	uint32_t* parent = r.parent + r.rbase[zi];
	const uint32_t n = r.nruns[zi];
	for (; cand; cand &= cand - 1u) {
		const uint32_t bit = __ffs(cand) - 1;
		const uint32_t m = mask_le(bit);
		const uint32_t ra = base_here + __popc(b_here & m) - 1u;
		const uint32_t rb = base_up + __popc(b_up & m) - 1u;
		if (ra < n && rb < n) run_unite(parent, ra, rb);
	}
}

// LDS table of a strip: the smaller, the more workgroups share a CU (the kernel waits on
// memory and LDS round trips, not on arithmetic); strips with more runs use the global forest
static inline uint32_t run_strip_runs() {
	if (const char* env = getenv("CKL_STRIP_RUNS")) { const int v = atoi(env); if (v >= 64 && v <= 16384) return static_cast<uint32_t>(v); }      // tuning aid
	return kStripRuns;
}
static inline uint32_t run_strip_rows(uint32_t row_words) {
	if (const char* env = getenv("CKL_STRIP_ROWS")) { const int v = atoi(env); if (v >= 2) return static_cast<uint32_t>(v); }      // tuning aid
	const uint32_t r = 1024u / (row_words ? row_words : 1u);      // 32K pixels per strip: the LDS table covers one run per ~10 pixels
	return r < 2u ? 2u : r;
}
// all unions of every slice on one stream (replaces a single k_run_union launch)
static inline void launch_run_union(hipStream_t s, uint32_t nslices, const RunGeom& g, const RunArrays& r) {
	const uint32_t rows = run_strip_rows(g.row_words);
	const uint32_t strips = (g.sy + rows - 1) / rows;
	const uint32_t sruns = run_strip_runs();
	hipLaunchKernelGGL(k_run_union_strips, dim3(strips, nslices), dim3(kBlock), sruns * sizeof(uint32_t), s, g, r, rows, sruns);
	if (strips > 1) {
		const uint32_t words = (strips - 1) * g.row_words;
		hipLaunchKernelGGL(k_run_union_seams, dim3((words + kBlock - 1) / kBlock, nslices), dim3(kBlock), 0, s, g, r, rows);
	}
}

// Component ids and the crc32c of the component image, in three fully parallel steps.
// Roots are ranked in run order (= raster order of each component's first pixel,
// cc3d.hpp:114-144).  A run of id c covering pixels [a, b) of an n-pixel slice
// contributes c * (G[n-a] ^ G[n-b]) to the raw crc, G[m] = x^32 + x^64 + ... + x^(32 m) mod P.
// The multiplication walks only the `idbits` significant bits of c; the common factor
// x^(32-idbits) is applied once per slice on the host side of the comparison.
struct ResolveScratch {
	uint16_t* run_local;       // [runs] roots: rank among the roots of their 256-run block
	uint32_t* blk_roots;       // [nslices][nblk]: roots per block, then exclusive prefix
	uint32_t nblk;             // blocks per slice = ceil(max run capacity / 256)
};

// step 1, grid = (run_count_blocks(nblk), nslices): count the roots per 256-run block (a root is
// its own parent once all unions are in: no pointer chasing here); a workgroup takes
// kCountBlocks blocks at once (their loads and scans share one trip to memory and one barrier pair)
constexpr uint32_t kCountBlocks = 4;
static inline uint32_t run_count_blocks(uint32_t nblk) { return (nblk + kCountBlocks - 1) / kCountBlocks; }
static __global__ void __launch_bounds__(kBlock) k_run_count(RunArrays r, ResolveScratch rs) {
	__shared__ uint32_t s_scan[kCountBlocks * kWaves];
	const uint32_t zi = blockIdx.y;
	const uint32_t n = r.nruns[zi];
	const uint32_t b0 = blockIdx.x * kCountBlocks;
	if (b0 * kBlock >= n) {
		if (threadIdx.x < kCountBlocks && b0 + threadIdx.x < rs.nblk) rs.blk_roots[zi * rs.nblk + b0 + threadIdx.x] = 0;
		return;
	}
	const uint32_t* parent = r.parent + r.rbase[zi];
	uint32_t v[kCountBlocks], tot[kCountBlocks], is_root[kCountBlocks];
#pragma unroll
	for (uint32_t k = 0; k < kCountBlocks; k++) {
		const uint32_t i = (b0 + k) * kBlock + threadIdx.x;
		is_root[k] = (i < n && parent[i] == i) ? 1u : 0u;
		v[k] = is_root[k];
	}
	block_excl_add<kCountBlocks>(v, tot, s_scan);
#pragma unroll
	for (uint32_t k = 0; k < kCountBlocks; k++) {
		const uint32_t i = (b0 + k) * kBlock + threadIdx.x;
		if (is_root[k]) rs.run_local[r.rbase[zi] + i] = static_cast<uint16_t>(v[k]);
		if (threadIdx.x == 0 && b0 + k < rs.nblk) rs.blk_roots[zi * rs.nblk + b0 + k] = tot[k];
	}
}

// step 2, grid = nslices: exclusive prefix of the per-block root counts -> component count
static __global__ void __launch_bounds__(kBlock) k_run_rank(RunArrays r, ResolveScratch rs, uint32_t idbits_in, uint32_t* __restrict__ crc_acc, uint32_t* __restrict__ idbits_out) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.x;
	const uint32_t n = r.nruns[zi];
	const uint32_t nb = (n + kBlock - 1) / kBlock;
	uint32_t* cnt = rs.blk_roots + zi * rs.nblk;
	uint32_t carry = 0;
	for (uint32_t b0 = 0; b0 < nb; b0 += kBlock) {
		const uint32_t b = b0 + threadIdx.x;
		const uint32_t c = b < nb ? cnt[b] : 0u;
		uint32_t v[1] = { c }, tot[1];
		block_excl_add<1>(v, tot, s_scan);
		if (b < nb) cnt[b] = carry + v[0];
		carry += tot[0];
	}
	if (threadIdx.x == 0) {
		r.ncomp[zi] = carry;
		crc_acc[zi] = 0;
		// idbits_in == 0: ids are below the run count, use its bit length (the encoder does
		// not know the component counts in advance) and report it
		if (idbits_out) idbits_out[zi] = idbits_in ? idbits_in : (n > 1 ? 32u - __clz(n - 1) : 1u);
	}
}

// component -> label applied in the same pass (flat labels on decode): run_label is typed
// like the output; has_label: 1 where the label matches
struct RunLabelArgs {
	const uint64_t* label_map;
	const uint64_t* comp_off;
	const uint32_t* ncomp_expect;
	uint32_t has_label;
	uint64_t label;
	void* run_label;
};

// step 3, grid = (run_assign_blocks(nblk), nslices): root (a short chain: strip root -> roots
// of the strips above), component id of every run + its crc contribution [+ its label].
// Every load here depends on the one before it (parent chain -> rank tables -> label), so a
// thread resolves kAssignRuns runs side by side to keep that many chains in flight.
constexpr uint32_t kAssignRuns = 4;
static inline uint32_t run_assign_blocks(uint32_t nblk) { return (nblk + kAssignRuns - 1) / kAssignRuns; }

template <typename OUT, bool LABELS>
static __global__ void __launch_bounds__(kBlock) k_run_assign(RunArrays r, ResolveScratch rs, const uint32_t* __restrict__ G, uint32_t n_pixels, uint32_t idbits_in, uint32_t* __restrict__ crc_acc, RunLabelArgs la) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.y;
	const uint32_t n = r.nruns[zi];
	const uint32_t i0 = blockIdx.x * (kBlock * kAssignRuns) + threadIdx.x;
	if (blockIdx.x * (kBlock * kAssignRuns) >= n) return;
	const uint64_t rb = r.rbase[zi];
	const uint32_t idbits = idbits_in ? idbits_in : (n > 1 ? 32u - __clz(n - 1) : 1u);
	const uint32_t* parent = r.parent + rb;
	uint32_t root[kAssignRuns], p[kAssignRuns], a[kAssignRuns], b[kAssignRuns];
	bool live[kAssignRuns];
#pragma unroll
	for (uint32_t k = 0; k < kAssignRuns; k++) {
		const uint32_t i = i0 + k * kBlock;
		live[k] = i < n;
		root[k] = live[k] ? i : 0u;
		p[k] = live[k] ? parent[i] : 0u;
		a[k] = live[k] ? r.run_start[rb + i] : 0u;
		b[k] = (live[k] && i + 1 < n) ? r.run_start[rb + i + 1] : n_pixels;
	}
	bool more;
	do {
		more = false;
#pragma unroll
		for (uint32_t k = 0; k < kAssignRuns; k++) {
			if (p[k] != root[k]) { root[k] = p[k]; p[k] = parent[p[k]]; more = true; }
		}
	} while (more);
	uint32_t cc[kAssignRuns];
#pragma unroll
	for (uint32_t k = 0; k < kAssignRuns; k++) cc[k] = rs.blk_roots[zi * rs.nblk + (root[k] >> 8)] + rs.run_local[rb + root[k]];
	uint32_t part = 0;
#pragma unroll
	for (uint32_t k = 0; k < kAssignRuns; k++) {
		if (!live[k]) continue;
		const uint32_t i = i0 + k * kBlock;
		if (LABELS) {
			uint64_t v = 0;
			if (cc[k] < la.ncomp_expect[zi]) v = la.label_map[la.comp_off[zi] + cc[k]];
			else atomicOr(r.slice_err + zi, ERR_NCOMP);
			if (la.has_label) v = (v == la.label);
			static_cast<OUT*>(la.run_label)[rb + i] = static_cast<OUT>(v);
		}
		else {
			r.run_cc[rb + i] = cc[k];
			if (r.comp_pix && root[k] == i) r.comp_pix[rb + cc[k]] = a[k];      // a component has fewer entries than the slice has runs
		}
		uint32_t wgt = G[n_pixels - a[k]] ^ G[n_pixels - b[k]];
		// sum over set bits j < idbits of c:  wgt * x^(idbits-1-j)
		for (int j = static_cast<int>(idbits) - 1; j >= 0; j--) {
			part ^= ((cc[k] >> j) & 1u) ? wgt : 0u;
			wgt = (wgt >> 1) ^ ((wgt & 1u) ? kCrcPoly : 0u);
		}
	}
	part = block_xor(part, s_scan);
	if (threadIdx.x == 0 && part) atomicXor(crc_acc + zi, part);
}

// the three steps on one stream (component ids only)
static inline void launch_run_resolve(hipStream_t s, uint32_t nslices, const RunArrays& r, const ResolveScratch& rs, const uint32_t* G, uint32_t n_pixels, uint32_t idbits_in, uint32_t* crc_acc, uint32_t* idbits_out) {
	static_assert(kBlock == 256, "run_local is indexed by root >> 8");
	hipLaunchKernelGGL(k_run_count, dim3(run_count_blocks(rs.nblk), nslices), dim3(kBlock), 0, s, r, rs);
	hipLaunchKernelGGL(k_run_rank, dim3(nslices), dim3(kBlock), 0, s, r, rs, idbits_in, crc_acc, idbits_out);
	RunLabelArgs none = {};
	hipLaunchKernelGGL((k_run_assign<uint8_t, false>), dim3(run_assign_blocks(rs.nblk), nslices), dim3(kBlock), 0, s, r, rs, G, n_pixels, idbits_in, crc_acc, none);
}

// G[k*B + i] = G[k*B] ^ x^(32 k B) * G[i]   (B = 1024; per-block constants from the host)
static __global__ void __launch_bounds__(kBlock) k_build_geom_table(const uint32_t* __restrict__ g_base, const uint32_t* __restrict__ blk_g, const uint32_t* __restrict__ blk_x, uint32_t n, uint32_t* __restrict__ G) {
	const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
	if (m > n) return;
	const uint32_t k = m >> 10, i = m & 1023u;
	G[m] = blk_g[k] ^ gf_mul(blk_x[k], g_base[i]);
}


}  // namespace dev
}  // namespace ckl
