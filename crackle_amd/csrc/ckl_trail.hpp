// The crack trail of a slice without a vertex-by-vertex serial walk.
//
// The reference's create_crack_codes (src/crackcodes.hpp:374-453) is a deterministic
// depth-first trail over the crack graph: inherently sequential, but a decision is only
// ever taken at vertices whose degree is not 2 (junctions, dead ends) and at the start
// vertex of a component.  Everything between two such vertices is a SEGMENT that the
// trail passes straight through (arrive -> exactly one edge left -> take it,
// crackcodes.hpp:419-433).  So:
//
//   k_trail_graph      crack planes -> vertex nibbles in 32 x 32 tiles (Graph::init,
//                      crackcodes.hpp:66-125) + per-block counts of the vertices of degree
//                      1, 3, 4 (NODES) and of "right+down" corners (candidate loop starts)
//   k_trail_count_scan per-block counts -> per-block bases (no atomics, tile order)
//   k_trail_nodes      node and corner lists
//   k_trail_segments   one thread per (node, direction) follows its segment to the node
//                      at the other end: end node, arrival edge, length, smallest vertex
//   k_trail_loops      closed loops without a node: found from their smallest vertex
//   k_trail_components one workgroup per slice: components of the node graph (union-find
//                      in LDS) and their smallest vertex = where next_cluster starts the
//                      chain (crackcodes.hpp:41-49, 399); a start in the middle of a
//                      segment splits it; start vertices in ascending order (bitmap scan)
//   k_trail_dfs        the exact serial trail, but over nodes only (tables in LDS): one
//                      item per traversed segment / 'b' / 't', with
//                      remove_initial_branch (185-242) and remove_spurious_branches
//                      (250-281) folded in
//   k_trail_offsets    item -> code offset (prefix sum of segment lengths)
//   k_trail_expand     one thread per item re-walks its segment and writes the code points
//
// The output is what the per-vertex walk produced: code points per chain in trail
// order, chain table (adjusted start vertex, offset, length) for k_finish.
#pragma once

#include "ckl_device.hpp"

namespace ckl {
namespace dev {

// The crack graph is stored in MICRO-TILES of 8 x 8 vertices, one nibble per vertex
// (bit0 right, bit1 left, bit2 down, bit3 up): 32 bytes, dword r = row r, nibble i = vertex i
// of the row.  Four micro-tiles (2 x 2) share one 128-byte line.  A walking lane keeps the
// micro-tile it is in in registers and only goes to memory when it leaves it: the walks are
// bound by cache-line traffic, not arithmetic, and this serves ~6 steps per fetch.
constexpr uint32_t kTrailTileShift = 5, kTrailTileDim = 32;        // thread mapping of k_trail_graph / k_trail_nodes: one thread per 32-vertex row piece
__device__ __forceinline__ uint32_t mt_index(uint32_t mx, uint32_t my, uint32_t mtx2) {
	return (((my >> 1) * mtx2 + (mx >> 1)) << 2) | ((my & 1u) << 1) | (mx & 1u);
}
struct MicroTile {
	uint4 lo, hi;          // rows 0-3, rows 4-7
	uint32_t mx, my;       // which micro-tile is cached (0xFFFFFFFF: none)
	__device__ __forceinline__ bool holds(uint32_t x, uint32_t y) const { return (x >> 3) == mx && (y >> 3) == my; }
	__device__ __forceinline__ uint32_t nib(uint32_t x, uint32_t y) const {
		const uint32_t r = y & 7u;
		const uint32_t a0 = (r & 1u) ? lo.y : lo.x, a1 = (r & 1u) ? lo.w : lo.z, a2 = (r & 1u) ? hi.y : hi.x, a3 = (r & 1u) ? hi.w : hi.z;
		const uint32_t b0 = (r & 2u) ? a1 : a0, b1 = (r & 2u) ? a3 : a2;
		return (((r & 4u) ? b1 : b0) >> ((x & 7u) * 4u)) & 15u;
	}
};

enum : uint32_t { TRAIL_ERR_CAPACITY = 1u };
constexpr uint32_t kDartNone = 0xFFFFFFFFu;
// items of the trail
constexpr uint32_t kItemSeg = 0u << 30, kItemCtl = 1u << 30, kItemDead = 2u << 30, kItemMask = 3u << 30;

struct TrailArgs {
	const uint4* adjm;           // micro-tiles (2 x uint4 each), see mt_index
	uint64_t adjm_stride;        // uint4 per slice
	uint32_t mtx2;               // micro-tile column pairs per row
	uint32_t tiles_x, tiles_y;   // 32 x 32 vertex tiles: thread mapping of k_trail_graph / k_trail_nodes
	const uint32_t* planeV;      // crack planes (k_trail_nodes recomputes the vertex nibbles from them)
	const uint32_t* planeH;
	uint32_t row_words;
	uint64_t plane_words;
	uint32_t sx, sy, inv;
	uint32_t sxe, sye;
	uint32_t nverts;
	uint32_t z0;                 // first slice of this launch (the trail runs in slice groups on several streams)
	const uint32_t* max_steps;   // [nslices] crack edges of the slice + 1
	uint32_t graph_blocks;       // workgroups per slice of k_trail_graph / k_trail_nodes
	const uint32_t* blk_special; // [nslices][graph_blocks] exclusive prefix of the node counts
	const uint32_t* blk_corner;
	// nodes
	const uint64_t* nbase;       // [nslices] base into the node arrays (darts: 4 * node)
	const uint32_t* ncap;        // [nslices]
	uint32_t* n_nodes;           // [nslices]
	uint32_t* node_vertex;
	uint8_t* node_adj;
	uint32_t* vert2node;         // [nslices][nverts], defined at node vertices only
	// candidate loop starts
	const uint64_t* cobase;
	const uint32_t* cocap;
	uint32_t* n_corners;
	uint32_t* corner_vertex;
	// darts (node * 4 + direction)
	uint32_t* dart_end;          // end node << 2 | arrival edge at the end node, kDartNone: no edge
	uint32_t* dart_len;
	uint32_t* dart_minv;         // smallest vertex on the closed segment
	uint32_t* dart_minpos;       // its distance from the node << 2 | arrival edge there
	// components
	uint32_t* parent;
	unsigned long long* compmin; // smallest vertex << 32 | dart that saw it
	uint32_t* start_bits;        // [nslices][start_words] bitmap over vertices
	uint32_t start_words;
	uint32_t* starts;            // [nslices] at nbase: start vertices ascending
	uint32_t* n_starts;
	// trail
	const uint64_t* ibase;       // [nslices] base into items / item_off
	const uint32_t* icap;
	uint32_t* items;
	uint32_t* item_off;
	uint32_t* n_items;
	const uint64_t* sbase;       // [nslices] base into the branch stack spill
	const uint32_t* scap;
	uint32_t* stack_node;
	uint32_t* stack_item;
	// chains (k_finish input)
	const uint64_t* kbase;
	const uint32_t* kcap;
	uint32_t* chain_node;
	uint32_t* chain_item0;       // first item of the chain
	uint32_t* chain_off;
	uint32_t* chain_clen;
	uint32_t* n_chains;
	uint32_t* n_raw;
	uint32_t* n_valid;
	const uint64_t* cbase;       // code points
	const uint32_t* ccap;
	uint8_t* cp;
	uint32_t* slice_err;
	unsigned long long* dbg;     // diagnostics (nullable): [0] wave iterations, [1] max per wave, [2] cycles, [3] max cycles, [4] waves, [5] lane-steps
};

__device__ __forceinline__ void mt_load(MicroTile& c, const uint4* adjm, uint32_t x, uint32_t y, uint32_t mtx2) {
	const uint4* p = adjm + static_cast<uint64_t>(mt_index(x >> 3, y >> 3, mtx2)) * 2u;
	c.lo = p[0]; c.hi = p[1];
	c.mx = x >> 3; c.my = y >> 3;
}
__device__ __forceinline__ void trail_step(uint32_t& x, uint32_t& y, uint32_t k) {
	if (k & 2u) y = (k & 1u) ? y - 1u : y + 1u;
	else x = (k & 1u) ? x - 1u : x + 1u;
}
__device__ __forceinline__ uint32_t trail_code(uint32_t k) { return (0x0231u >> (4u * k)) & 3u; }   // right->1, left->3, down->2, up->0

// ---- crack graph ------------------------------------------------------------------
// One thread per tile row (32 vertices of one row): the four edge bit rows come straight
// from plane words, are spread to one nibble per byte and stored as 32 contiguous bytes.
// planeV bit (x,y): pixels (x-1,y)|(x,y) differ; planeH bit (x,y): pixels (x,y-1)|(x,y) differ.
// An interior pixel pair carries a crack when it differs (IMPERMISSIBLE) or is equal
// (PERMISSIBLE); image-border pairs never do (crackcodes.hpp:66-125).
struct TileRowBits { uint32_t R, L, D, U; };

__device__ __forceinline__ TileRowBits trail_row_bits(
	const uint32_t* __restrict__ pv, const uint32_t* __restrict__ ph, uint32_t row_words,
	uint32_t sx, uint32_t sy, uint32_t tx, uint32_t y, uint32_t inv
) {
	// masks over the 32 vertices x = 32 tx + i
	const uint32_t x0 = tx << 5;
	const uint32_t lt_sx = x0 >= sx ? 0u : (sx - x0 >= 32u ? 0xFFFFFFFFu : ((1u << (sx - x0)) - 1u));        // x < sx
	const uint32_t le_sx = x0 > sx ? 0u : (sx - x0 >= 31u ? 0xFFFFFFFFu : ((2u << (sx - x0)) - 1u));        // x <= sx
	const uint32_t ge_1 = tx == 0 ? 0xFFFFFFFEu : 0xFFFFFFFFu;                                                // x >= 1
	TileRowBits b = { 0, 0, 0, 0 };
	const bool have_w = tx < row_words;
	if (y >= 1 && y < sy) {
		const uint32_t h = have_w ? ph[static_cast<uint64_t>(y) * row_words + tx] : 0u;
		const uint32_t hp = tx > 0 ? ph[static_cast<uint64_t>(y) * row_words + tx - 1] : 0u;
		b.R = (h ^ inv) & lt_sx;                                   // edge to the right: pixels (x,y-1)|(x,y)
		b.L = (((h << 1) | (hp >> 31)) ^ inv) & ge_1 & le_sx;       // edge to the left: pixels (x-1,y-1)|(x-1,y)
	}
	if (y < sy) b.D = ((have_w ? pv[static_cast<uint64_t>(y) * row_words + tx] : 0u) ^ inv) & ge_1 & lt_sx;       // edge down: pixels (x-1,y)|(x,y)
	if (y >= 1 && y <= sy) b.U = ((have_w ? pv[static_cast<uint64_t>(y - 1) * row_words + tx] : 0u) ^ inv) & ge_1 & lt_sx;
	return b;
}
// vertices whose degree is 1, 3 or 4 / exactly "right + down"
__device__ __forceinline__ uint32_t trail_special_mask(const TileRowBits& b) {
	const uint32_t a1 = b.R ^ b.L, a2 = b.R & b.L, c1 = b.D ^ b.U, c2 = b.D & b.U;
	const uint32_t bit0 = a1 ^ c1, carry = a1 & c1;
	const uint32_t bit1 = a2 ^ c2 ^ carry, bit2 = (a2 & c2) | (carry & (a2 ^ c2));
	const uint32_t deg2 = ~bit0 & bit1 & ~bit2;
	return (b.R | b.L | b.D | b.U) & ~deg2;
}
__device__ __forceinline__ uint32_t trail_corner_mask(const TileRowBits& b) { return b.R & b.D & ~b.L & ~b.U; }
// bit i of an 8-bit mask -> bit 4 i
__device__ __forceinline__ uint32_t spread8(uint32_t m) {
	uint32_t x = (m | (m << 12)) & 0x000F000Fu;
	x = (x | (x << 6)) & 0x03030303u;
	x = (x | (x << 3)) & 0x11111111u;
	return x;
}

constexpr uint32_t kGraphTiles = kBlock / kTrailTileDim;     // 32 x 32 vertex tiles per workgroup

// grid = (graph_blocks, nslices)
static __global__ void __launch_bounds__(kBlock) k_trail_graph(
	const uint32_t* __restrict__ planeV, const uint32_t* __restrict__ planeH, uint32_t row_words, uint64_t plane_words,
	uint32_t sx, uint32_t sy, uint32_t permissible, uint32_t* __restrict__ adjm_words, uint64_t adjm_stride_words,
	uint32_t mtx2, uint32_t tiles_x, uint32_t tiles_y, uint32_t* __restrict__ blk_special, uint32_t* __restrict__ blk_corner
) {
	__shared__ uint32_t s_red[2 * kWaves];
	const uint32_t zi = blockIdx.y;
	const uint32_t tile = blockIdx.x * kGraphTiles + (threadIdx.x >> 5);
	const uint32_t r = threadIdx.x & 31u;
	const uint32_t mtx = (sx + 1u + 7u) >> 3, mty = (sy + 1u + 7u) >> 3;
	uint32_t ns = 0, nc = 0;
	if (tile < tiles_x * tiles_y) {
		const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
		const uint32_t y = (ty << kTrailTileShift) + r;
		TileRowBits b = { 0, 0, 0, 0 };
		if (y <= sy) b = trail_row_bits(planeV + zi * plane_words, planeH + zi * plane_words, row_words, sx, sy, tx, y, permissible ? 0xFFFFFFFFu : 0u);
		uint32_t* dst = adjm_words + zi * adjm_stride_words;
		const uint32_t my = y >> 3;
		if (my < mty) {
#pragma unroll
			for (uint32_t q = 0; q < 4; q++) {
				const uint32_t mx = tx * 4u + q;
				if (mx >= mtx) break;
				const uint32_t w = spread8((b.R >> (8 * q)) & 255u) | (spread8((b.L >> (8 * q)) & 255u) << 1)
					| (spread8((b.D >> (8 * q)) & 255u) << 2) | (spread8((b.U >> (8 * q)) & 255u) << 3);
				dst[static_cast<uint64_t>(mt_index(mx, my, mtx2)) * 8u + (y & 7u)] = w;
			}
		}
		ns = __popc(trail_special_mask(b));
		nc = __popc(trail_corner_mask(b));
	}
	ns = wave_sum(ns); nc = wave_sum(nc);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) { s_red[wave] = ns; s_red[kWaves + wave] = nc; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t ts = 0, tc = 0;
		for (int w = 0; w < kWaves; w++) { ts += s_red[w]; tc += s_red[kWaves + w]; }
		blk_special[static_cast<uint64_t>(zi) * gridDim.x + blockIdx.x] = ts;
		blk_corner[static_cast<uint64_t>(zi) * gridDim.x + blockIdx.x] = tc;
	}
}

// grid = nslices: per-block counts -> exclusive prefixes (in place) and the slice totals
static __global__ void __launch_bounds__(kBlock) k_trail_count_scan(
	uint32_t* __restrict__ blk_special, uint32_t* __restrict__ blk_corner, uint32_t nblk,
	uint32_t* __restrict__ tot_special, uint32_t* __restrict__ tot_corner
) {
	__shared__ uint32_t s_scan[2 * kWaves];
	const uint32_t zi = blockIdx.x;
	uint32_t* bs = blk_special + static_cast<uint64_t>(zi) * nblk;
	uint32_t* bc = blk_corner + static_cast<uint64_t>(zi) * nblk;
	uint32_t cs = 0, cc = 0;
	for (uint32_t b0 = 0; b0 < nblk; b0 += kBlock) {
		const uint32_t b = b0 + threadIdx.x;
		uint32_t v[2] = { b < nblk ? bs[b] : 0u, b < nblk ? bc[b] : 0u }, tot[2];
		block_excl_add<2>(v, tot, s_scan);
		if (b < nblk) { bs[b] = cs + v[0]; bc[b] = cc + v[1]; }
		cs += tot[0]; cc += tot[1];
	}
	if (threadIdx.x == 0) { tot_special[zi] = cs; tot_corner[zi] = cc; }
}

// grid = (graph_blocks, nslices), same thread -> tile row mapping as k_trail_graph:
// nodes are numbered in (tile, row, x) order, no atomics
static __global__ void __launch_bounds__(kBlock) k_trail_nodes(TrailArgs a) {
	__shared__ uint32_t s_scan[2 * kWaves];
	const uint32_t zi = blockIdx.y + a.z0;
	const uint32_t tile = blockIdx.x * kGraphTiles + (threadIdx.x >> 5);
	const uint32_t r = threadIdx.x & 31u;
	TileRowBits b = { 0, 0, 0, 0 };
	uint32_t x0 = 0, y = 0;
	if (tile < a.tiles_x * a.tiles_y) {
		const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
		x0 = tx << kTrailTileShift; y = (ty << kTrailTileShift) + r;
		if (y <= a.sy) b = trail_row_bits(a.planeV + zi * a.plane_words, a.planeH + zi * a.plane_words, a.row_words, a.sx, a.sy, tx, y, a.inv);
	}
	const uint32_t ms = trail_special_mask(b), mc = trail_corner_mask(b);
	uint32_t v[2] = { static_cast<uint32_t>(__popc(ms)), static_cast<uint32_t>(__popc(mc)) }, tot[2];
	block_excl_add<2>(v, tot, s_scan);
	uint32_t j = a.blk_special[static_cast<uint64_t>(zi) * a.graph_blocks + blockIdx.x] + v[0];
	uint32_t c = a.blk_corner[static_cast<uint64_t>(zi) * a.graph_blocks + blockIdx.x] + v[1];
	const uint64_t nb = a.nbase[zi];
	const uint32_t ncap = a.ncap[zi], cocap = a.cocap[zi];
	uint32_t err = 0;
	for (uint32_t m = ms; m; m &= m - 1u) {
		const uint32_t i = __ffs(m) - 1;
		const uint32_t vtx = y * a.sxe + x0 + i;
		if (j < ncap) {
			a.node_vertex[nb + j] = vtx;
			a.node_adj[nb + j] = static_cast<uint8_t>(((b.R >> i) & 1u) | (((b.L >> i) & 1u) << 1) | (((b.D >> i) & 1u) << 2) | (((b.U >> i) & 1u) << 3));
			a.vert2node[static_cast<uint64_t>(zi) * a.nverts + vtx] = j;
		}
		else err = TRAIL_ERR_CAPACITY;
		j++;
	}
	for (uint32_t m = mc; m; m &= m - 1u) {
		const uint32_t i = __ffs(m) - 1;
		if (c < cocap) a.corner_vertex[a.cobase[zi] + c] = y * a.sxe + x0 + i;
		else err = TRAIL_ERR_CAPACITY;
		c++;
	}
	if (err) atomicOr(a.slice_err + zi, err);
	// node / corner totals of the slice (k_trail_loops and k_trail_components append to n_nodes)
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) {
		a.n_nodes[zi] = min(j, ncap);
		a.n_corners[zi] = min(c, cocap);
	}
}

// Segment walks have very uneven lengths (mean ~20 edges, long tail), so a lane that
// finishes its segment takes the next one of its wavefront's range instead of idling
// until the slowest lane is done: kWalkChunk darts (node * 4 + direction) per wavefront
// (small: the walks are latency bound, so the chip wants many wavefronts in flight).
constexpr uint32_t kWalkChunk = 192;       // darts per wavefront (k_trail_segments)
constexpr uint32_t kExpandChunk = 384;     // items per wavefront (k_trail_expand)
constexpr int kWalkRefill = 16;     // idle lanes that trigger a refill (its loads cost as much as a step)
constexpr int kWalkAhead = 3;       // steps a lane may run ahead inside its micro-tile per memory round trip

// Results are staged in LDS and written out once per wavefront: on this architecture loads
// and stores retire in order on one counter, so a store inside the loop would make every
// following step wait for it.
// grid = (ceil(4 * max nodes / (kWalkChunk * 4)), nslices)
static __global__ void __launch_bounds__(kBlock) k_trail_segments(TrailArgs a) {
	__shared__ uint4 s_res[kWaves][kWalkChunk];       // end, len, minv, minpos
	const uint32_t zi = blockIdx.y + a.z0;
	const uint32_t lane = threadIdx.x & (kWave - 1);
	const uint32_t wv = threadIdx.x >> 6;
	const uint32_t wave = blockIdx.x * kWaves + wv;
	const uint32_t n_darts = min(a.n_nodes[zi], a.ncap[zi]) * 4u;      // (k_trail_loops appends later, in its own launch)
	const uint32_t range_begin = wave * kWalkChunk;
	uint32_t next = range_begin;
	const uint32_t range_end = min(next + kWalkChunk, n_darts);
	if (next >= range_end) return;
	const uint64_t nb = a.nbase[zi];
	const uint4* adjm = a.adjm + zi * a.adjm_stride;
	const uint32_t* v2n = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
	const uint32_t cap = a.max_steps[zi];
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	uint4* res = s_res[wv];
	for (uint32_t i = lane; i < kWalkChunk; i += kWave) res[i] = make_uint4(kDartNone, 0u, 0u, 0u);
	MicroTile c;
	c.lo = c.hi = make_uint4(0, 0, 0, 0); c.mx = c.my = 0xFFFFFFFFu;
	bool active = false;
	// (x, y): the vertex to look at next, reached by a move in direction k after `steps` edges
	uint32_t d = 0, x = 0, y = 0, k = 0, steps = 0, minv = 0, minpos = 0, err = 0;
	unsigned long long dbg_iter = 0, dbg_lane = 0;
	const unsigned long long dbg_t0 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
	for (;;) {
		if (a.dbg) { dbg_iter++; dbg_lane += __popcll(__ballot(active)); }
		// One memory round trip per iteration: the loads of the lanes that take a new dart and
		// the micro-tile loads of the lanes that left theirs are issued together, then consumed.
		const unsigned long long need = __ballot(!active);
		const bool refill = next < range_end && (__popcll(need) >= kWalkRefill || need == ~0ull);
		const uint32_t cand = next + static_cast<uint32_t>(__popcll(need & lt_mask));
		const bool take = refill && !active && cand < range_end;
		uint32_t r_adj = 0, r_v0 = 0;
		if (take) { r_adj = a.node_adj[nb + (cand >> 2)]; r_v0 = a.node_vertex[nb + (cand >> 2)]; }
		if (active && !c.holds(x, y)) mt_load(c, adjm, x, y, a.mtx2);
		// every lane runs ahead for as long as it stays inside its micro-tile
		for (int it = 0; it < kWalkAhead; it++) {
			const bool go = active && c.holds(x, y);
			if (!__ballot(go)) break;
			if (go) {
				const uint32_t w = y * a.sxe + x;
				const uint32_t arr = k ^ 1u;
				if (w < minv) { minv = w; minpos = (steps << 2) | arr; }
				const uint32_t nib = c.nib(x, y);
				bool fin = false;
				if (__popc(nib) != 2) fin = true;
				else if (steps >= cap) { err = TRAIL_ERR_CAPACITY; fin = true; }
				else { k = __ffs(nib & ~(1u << arr)) - 1; trail_step(x, y, k); steps++; }
				if (fin) {
					// x: end vertex << 2 | arrival edge (the vertex becomes a node index in the flush below)
					res[d - range_begin] = make_uint4((w << 2) | arr, steps, minv, minpos);
					active = false;
				}
			}
		}
		if (take && ((r_adj >> (cand & 3u)) & 1u)) {
			d = cand; k = cand & 3u;
			y = r_v0 / a.sxe; x = r_v0 - y * a.sxe;
			minv = r_v0; minpos = 0;
			trail_step(x, y, k);
			steps = 1;
			active = true;
		}
		if (refill) next += static_cast<uint32_t>(__popcll(need));
		if (!__ballot(active) && next >= range_end) break;
	}
	if (a.dbg && lane == 0) {
		const unsigned long long cyc = __builtin_amdgcn_s_memtime() - dbg_t0;
		atomicAdd(a.dbg + 0, dbg_iter); atomicMax(a.dbg + 1, dbg_iter);
		atomicAdd(a.dbg + 2, cyc); atomicMax(a.dbg + 3, cyc);
		atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, dbg_lane);
	}
	// (LDS accesses of one wavefront are ordered: no barrier needed for its own rows)
	const uint64_t db = nb * 4u + range_begin;
	for (uint32_t i = lane; i < range_end - range_begin; i += kWave) {
		const uint4 r = res[i];
		a.dart_end[db + i] = r.x == kDartNone ? kDartNone : ((v2n[r.x >> 2] << 2) | (r.x & 3u));
		a.dart_len[db + i] = r.y;
		a.dart_minv[db + i] = r.z;
		a.dart_minpos[db + i] = r.w;
	}
	if (err) atomicOr(a.slice_err + zi, err);
}

// grid = (ceil(max corners / 256), nslices): a "right+down" corner is the start of a
// closed loop when following the loop from it never meets a node or a smaller vertex
static __global__ void __launch_bounds__(kBlock) k_trail_loops(TrailArgs a) {
	const uint32_t zi = blockIdx.y + a.z0;
	const uint32_t ci = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t nc = min(a.n_corners[zi], a.cocap[zi]);
	const uint4* adjm = a.adjm + zi * a.adjm_stride;
	const uint32_t cap = a.max_steps[zi];
	bool active = ci < nc;
	const uint32_t v0 = active ? a.corner_vertex[a.cobase[zi] + ci] : 0u;
	uint32_t y = v0 / a.sxe, x = v0 - y * a.sxe;
	uint32_t k = 0, steps = 1;
	trail_step(x, y, k);
	bool loop = false;
	MicroTile c;
	c.lo = c.hi = make_uint4(0, 0, 0, 0); c.mx = c.my = 0xFFFFFFFFu;
	while (__ballot(active)) {
		if (active && !c.holds(x, y)) mt_load(c, adjm, x, y, a.mtx2);
		for (int it = 0; it < kWalkAhead; it++) {
			const bool go = active && c.holds(x, y);
			if (!__ballot(go)) break;
			if (go) {
				const uint32_t w = y * a.sxe + x;
				const uint32_t nib = c.nib(x, y);
				if (w == v0) { loop = true; active = false; }
				else if (w < v0 || __popc(nib) != 2 || steps >= cap) active = false;
				else { k = __ffs(nib & ~(1u << (k ^ 1u))) - 1; trail_step(x, y, k); steps++; }
			}
		}
	}
	if (!loop) return;
	const uint32_t j = atomicAdd(a.n_nodes + zi, 1u);
	if (j >= a.ncap[zi]) { atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY); return; }
	const uint64_t nb = a.nbase[zi];
	a.node_vertex[nb + j] = v0;
	a.node_adj[nb + j] = 5u;
	a.vert2node[static_cast<uint64_t>(zi) * a.nverts + v0] = j;
	const uint64_t db = (nb + j) * 4u;
	for (uint32_t q = 0; q < 4; q++) { a.dart_end[db + q] = kDartNone; a.dart_len[db + q] = 0; a.dart_minv[db + q] = v0; a.dart_minpos[db + q] = 0; }
	a.dart_end[db + 0] = (j << 2) | 2u; a.dart_len[db + 0] = steps;     // leaves to the right, comes back up the down edge
	a.dart_end[db + 2] = (j << 2) | 0u; a.dart_len[db + 2] = steps;
}

// ---- components of the node graph ------------------------------------------------
// union-find with root = smallest node (parents only ever decrease); the table lives
// in LDS or, for slices with too many nodes, in global memory
template <bool LDS>
__device__ __forceinline__ uint32_t tuf_load(uint32_t* L, uint32_t i) {
	// LDS: a relaxed workgroup-scope atomic load, not a volatile one (hipcc keeps volatile accesses on
	// flat pointers: a flat instruction per access, and every wait for one waits for all memory traffic)
	if (LDS) return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS>
__device__ __forceinline__ uint32_t tuf_find(uint32_t* L, uint32_t a) {
	uint32_t p = tuf_load<LDS>(L, a);
	while (p != a) {
		const uint32_t gp = tuf_load<LDS>(L, p);
		if (gp != p) atomicMin(L + a, gp);
		a = p;
		p = gp;
	}
	return a;
}
template <bool LDS>
__device__ __forceinline__ void tuf_unite(uint32_t* L, uint32_t a, uint32_t b) {
	for (;;) {
		a = tuf_find<LDS>(L, a);
		b = tuf_find<LDS>(L, b);
		if (a == b) return;
		if (a > b) { const uint32_t t = a; a = b; b = t; }
		const uint32_t old = atomicMin(L + b, a);
		if (old == b) return;
		b = old;
	}
}

constexpr uint32_t kStartList = 2048;   // start vertices ranked directly up to this many per slice
constexpr int kCompBlock = 256;      // threads of k_trail_components (one workgroup per slice)

template <bool LDS>
__device__ __forceinline__ void trail_components_slice(const TrailArgs& a, uint32_t zi, uint32_t* parent, uint32_t nn, uint32_t* s_scan, uint32_t* s_nstart) {
	const uint64_t nb = a.nbase[zi];
	uint32_t* start_tmp = a.items + a.ibase[zi];      // the item table is not in use yet (capacity >= kStartList)
	unsigned long long dg_t = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) { if (a.dbg && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); atomicAdd(a.dbg + slot, now - dg_t); dg_t = now; } };
	const uint32_t* dart_end = a.dart_end + nb * 4u;
	unsigned long long* compmin = a.compmin + nb;
	for (uint32_t j = threadIdx.x; j < nn; j += kCompBlock) { parent[j] = j; compmin[j] = ~0ull; }
	__syncthreads();
	if (!LDS) __threadfence();
	// the dart ends of kUniteBatch steps are loaded before the first union: the unions are dependent
	// round trips to the table and would otherwise wait for memory once per step
	constexpr uint32_t kUniteBatch = 8;
	for (uint32_t d0 = threadIdx.x; d0 < nn * 4u; d0 += kCompBlock * kUniteBatch) {
		uint32_t e[kUniteBatch];
#pragma unroll
		for (uint32_t k = 0; k < kUniteBatch; k++) {
			const uint32_t d = d0 + k * kCompBlock;
			e[k] = d < nn * 4u ? dart_end[d] : kDartNone;
		}
#pragma unroll
		for (uint32_t k = 0; k < kUniteBatch; k++) {
			if (e[k] == kDartNone) continue;
			const uint32_t j = (d0 + k * kCompBlock) >> 2, j2 = e[k] >> 2;
			if (j < j2 && j2 < nn) tuf_unite<LDS>(parent, j, j2);      // the dart at the far end names the same segment
		}
	}
	__syncthreads();
	if (!LDS) __threadfence();
	stamp(12);
	// smallest vertex of every component and a dart that saw it; lanes of a wavefront that
	// share a root combine first (a slice usually has one giant component)
	for (uint32_t j0 = 0; j0 < nn; j0 += kCompBlock) {
		const uint32_t j = j0 + threadIdx.x;
		uint32_t root = 0xFFFFFFFFu;
		unsigned long long val = ~0ull;
		if (j < nn) {
			// the node's own minimum over its (at most four) segments first (one 16-byte load each)
			const uint4 e4 = *reinterpret_cast<const uint4*>(dart_end + j * 4u);
			const uint4 m4 = *reinterpret_cast<const uint4*>(a.dart_minv + nb * 4u + j * 4u);
			const uint32_t ee[4] = { e4.x, e4.y, e4.z, e4.w }, mm[4] = { m4.x, m4.y, m4.z, m4.w };
#pragma unroll
			for (uint32_t k = 0; k < 4u; k++) {
				if (ee[k] == kDartNone) continue;
				const unsigned long long v = (static_cast<unsigned long long>(mm[k]) << 32) | (j * 4u + k);
				val = v < val ? v : val;
			}
			if (val != ~0ull) root = tuf_find<LDS>(parent, j);
		}
		unsigned long long todo = __ballot(root != 0xFFFFFFFFu);
		while (todo) {
			const int first = __ffsll(static_cast<long long>(todo)) - 1;
			const uint32_t r0 = __shfl(root, first, kWave);
			const bool mine = (root == r0);
			unsigned long long v = mine ? val : ~0ull;
#pragma unroll
			for (int s = kWave / 2; s >= 1; s >>= 1) {
				const unsigned long long o = __shfl_xor(v, s, kWave);
				v = o < v ? o : v;
			}
			if ((threadIdx.x & (kWave - 1)) == first) atomicMin(compmin + r0, v);
			todo &= ~__ballot(mine);
		}
	}
	__syncthreads();
	__threadfence();
	stamp(13);
	// one thread per component root: mark the start vertex; a start inside a segment
	// becomes a node of degree 2 (right + down) that splits the segment
	uint32_t* bits = a.start_bits + static_cast<uint64_t>(zi) * a.start_words;
	for (uint32_t j = threadIdx.x; j < nn; j += kCompBlock) {
		if (tuf_load<LDS>(parent, j) != j) continue;
		const unsigned long long m = __hip_atomic_load(compmin + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (m == ~0ull) continue;
		const uint32_t minv = static_cast<uint32_t>(m >> 32);
		const uint32_t d = static_cast<uint32_t>(m);
		atomicOr(bits + (minv >> 5), 1u << (minv & 31u));
		{
			const uint32_t slot = atomicAdd(s_nstart, 1u);
			if (slot < kStartList) start_tmp[slot] = minv;
		}
		const uint64_t db = nb * 4u;
		const uint32_t len = a.dart_len[db + d];
		const uint32_t pos = a.dart_minpos[db + d] >> 2, arr = a.dart_minpos[db + d] & 3u;
		if (pos == 0 || pos == len) continue;                // the start is a node (a dead end)
		const uint32_t s = atomicAdd(a.n_nodes + zi, 1u);
		if (s >= a.ncap[zi]) { atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY); continue; }
		a.node_vertex[nb + s] = minv;
		a.node_adj[nb + s] = 5u;
		a.vert2node[static_cast<uint64_t>(zi) * a.nverts + minv] = s;
		const uint32_t d2 = a.dart_end[db + d];              // far end: node << 2 | arrival = the dart that runs back
		const uint32_t other = arr ^ 2u;
		for (uint32_t q = 0; q < 4; q++) { a.dart_end[db + s * 4u + q] = kDartNone; a.dart_len[db + s * 4u + q] = 0; }
		a.dart_end[db + d] = (s << 2) | arr;        a.dart_len[db + d] = pos;
		a.dart_end[db + d2] = (s << 2) | other;     a.dart_len[db + d2] = len - pos;
		a.dart_end[db + s * 4u + arr] = d;          a.dart_len[db + s * 4u + arr] = pos;
		a.dart_end[db + s * 4u + other] = d2;       a.dart_len[db + s * 4u + other] = len - pos;
	}
	__syncthreads();
	__threadfence();
	stamp(14);
	// start vertices in ascending order.  Usually there are a handful: rank them directly;
	// a slice with many components (noise) scans the bitmap over the vertices instead.
	uint32_t* starts = a.starts + nb;
	const uint32_t cap = a.kcap[zi];
	const uint32_t n_listed = *s_nstart;
	if (n_listed <= kStartList) {
		for (uint32_t i = threadIdx.x; i < n_listed; i += kCompBlock) {
			const uint32_t v = __hip_atomic_load(start_tmp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			uint32_t rank = 0;
			for (uint32_t q = 0; q < n_listed; q++) rank += __hip_atomic_load(start_tmp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v ? 1u : 0u;
			if (rank < cap) starts[rank] = v;
			else atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY);
		}
		if (threadIdx.x == 0) a.n_starts[zi] = n_listed < cap ? n_listed : cap;
		stamp(15);
		return;
	}
	constexpr uint32_t kPer = 8;
	uint32_t carry = 0, err = 0;
	for (uint32_t w0 = 0; w0 < a.start_words; w0 += kCompBlock * kPer) {
		uint32_t b[kPer], cnt = 0;
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t wi = w0 + threadIdx.x * kPer + q;
			b[q] = wi < a.start_words ? __hip_atomic_load(bits + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			cnt += __popc(b[q]);
		}
		uint32_t v[1] = { cnt }, tot[1];
		block_excl_add<1, kCompBlock / kWave>(v, tot, s_scan);
		uint32_t o = carry + v[0];
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t wi = w0 + threadIdx.x * kPer + q;
			for (uint32_t m = b[q]; m; m &= m - 1u) {
				if (o < cap) starts[o] = wi * 32u + (__ffs(m) - 1);
				else err = TRAIL_ERR_CAPACITY;
				o++;
			}
		}
		carry += tot[0];
	}
	if (threadIdx.x == 0) a.n_starts[zi] = carry < cap ? carry : cap;
	if (err) atomicOr(a.slice_err + zi, err);
	stamp(15);
}

// grid = nslices; dynamic LDS = lds_bytes (the union-find table of slices that fit)
static __global__ void __launch_bounds__(kCompBlock) k_trail_components(TrailArgs a, uint32_t lds_bytes) {
	extern __shared__ uint32_t s_trail[];
	__shared__ uint32_t s_scan[kCompBlock / kWave];
	__shared__ uint32_t s_nstart;
	const uint32_t zi = blockIdx.x + a.z0;
	const uint32_t nn = min(a.n_nodes[zi], a.ncap[zi]);
	if (threadIdx.x == 0) s_nstart = 0;
	__syncthreads();
	if (nn * 4u <= lds_bytes) trail_components_slice<true>(a, zi, s_trail, nn, s_scan, &s_nstart);
	else trail_components_slice<false>(a, zi, a.parent + a.nbase[zi], nn, s_scan, &s_nstart);
}

// ---- the serial trail over nodes ------------------------------------------------
// One wavefront per slice, wave-uniform scalar state (values read from memory go through
// readfirstlane); stores are issued by lane 0.  Node tables: LDS (16-bit dart ends) when
// the slice has few enough nodes, else the global arrays.  A step costs one LDS round
// trip: the remaining-edge nibble and the four dart ends of a node are fetched together,
// the edge consumed on arrival is carried in a register (pend) instead of being cleared
// at the far node first.
enum : uint32_t { TCODE_UP = 0, TCODE_RIGHT = 1, TCODE_DOWN = 2, TCODE_LEFT = 3, TCODE_NONE = 0xFE };

struct TrailTabLds {
	uint8_t* adj;
	unsigned long long* end4;      // four 16-bit dart ends (node << 2 | arrival) per node
	uint32_t dummy;                // byte offset from adj of 64 scratch dwords: where the lanes other than 0 write
	__device__ __forceinline__ void load(uint32_t j, uint32_t& av, uint32_t& e_lo, uint32_t& e_hi) const {
		const uint32_t a0 = adj[j];
		const unsigned long long e = end4[j];
		av = __builtin_amdgcn_readfirstlane(a0);
		e_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e));
		e_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e >> 32));
	}
	__device__ __forceinline__ uint32_t pick(uint32_t e_lo, uint32_t e_hi, uint32_t k) const {
		const uint32_t w = (k & 2u) ? e_hi : e_lo;
		return (w >> ((k & 1u) * 16u)) & 0xFFFFu;
	}
	// no exec-mask juggling and no 64-way same-address write: lane 0 stores to the table,
	// every other lane to its own scratch dword
	__device__ __forceinline__ void set_adj(uint32_t j, uint32_t v, bool l0) const {
		adj[l0 ? j : dummy + (threadIdx.x << 2)] = static_cast<uint8_t>(v);
	}
	__device__ __forceinline__ uint32_t get_end(uint32_t d) const {
		return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<const uint16_t*>(end4)[d]));
	}
};
struct TrailTabGlobal {
	uint8_t* adj;
	const uint32_t* end;
	__device__ __forceinline__ void load(uint32_t j, uint32_t& av, uint32_t& e_lo, uint32_t& e_hi) const {
		// word loads that bypass the vector L1: lane 0's stores must be read back as written
		const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(adj + j) & ~static_cast<uintptr_t>(3));
		const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(adj + j) & 3u) * 8u;
		av = (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> sh) & 0xFFu;
		e_lo = j; e_hi = 0;
	}
	__device__ __forceinline__ uint32_t pick(uint32_t e_lo, uint32_t e_hi, uint32_t k) const { return __builtin_amdgcn_readfirstlane(end[e_lo * 4u + k]); }
	__device__ __forceinline__ void set_adj(uint32_t j, uint32_t v, bool l0) const {
		if (l0) {
			// read-modify-write of the containing word (single writer per slice, slices are 16-byte aligned)
			uint32_t* w = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(adj + j) & ~static_cast<uintptr_t>(3));
			const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(adj + j) & 3u) * 8u;
			const uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(w, (old & ~(0xFFu << sh)) | ((v & 0xFFu) << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	__device__ __forceinline__ uint32_t get_end(uint32_t d) const { return __builtin_amdgcn_readfirstlane(end[d]); }
};

template <typename TAB>
__device__ __forceinline__ void trail_dfs_slice(
	const TrailArgs& a, uint32_t zi, const TAB& tab, uint32_t* s_stack, uint32_t lds_stack_cap
) {
	const bool l0 = (threadIdx.x == 0);
	const uint64_t nb = a.nbase[zi];
	const uint32_t* starts = a.starts + nb;
	const uint32_t n_starts = a.n_starts[zi];
	uint32_t* items = a.items + a.ibase[zi];
	const uint32_t icap = a.icap[zi];
	uint32_t* st_node = a.stack_node + a.sbase[zi];
	uint32_t* st_item = a.stack_item + a.sbase[zi];
	const uint32_t scap = a.scap[zi];
	uint32_t* ch_node = a.chain_node + a.kbase[zi];
	uint32_t* ch_item0 = a.chain_item0 + a.kbase[zi];
	const uint32_t kcap = a.kcap[zi];
	const uint32_t* vert2node = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
	uint2* s_stack2 = reinterpret_cast<uint2*>(s_stack);

	// Every lane carries the same state, so stores need no lane predicate: all lanes write
	// the same value to the same address (one LDS / memory transaction).
	uint32_t ni = 0, nch = 0, err = 0, dbg_iters = 0;
	const unsigned long long dbg_t0 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
	const unsigned long long dbg_r0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
	// last: direction of the previous symbol's last code point as an EDGE BIT number
	// (0 right, 1 left, 2 down, 3 up), 4 = nothing emitted yet.
	// 'b' is (UP,DOWN) unless the previous code is DOWN (or there is none), then (LEFT,RIGHT);
	// 't' is (DOWN,UP) unless the previous code is UP (or there is none), then (RIGHT,LEFT)   (crackcodes.hpp:155-174)
	constexpr uint32_t kNone = 4u, kLeft = 1u, kDown = 2u, kUp = 3u;
	constexpr uint32_t kB = kItemCtl | TCODE_UP | (TCODE_DOWN << 2), kBalt = kItemCtl | TCODE_LEFT | (TCODE_RIGHT << 2);
	constexpr uint32_t kT = kItemCtl | TCODE_DOWN | (TCODE_UP << 2), kTalt = kItemCtl | TCODE_RIGHT | (TCODE_LEFT << 2);
	struct __attribute__((packed, aligned(4))) Item2 { uint32_t a, b; };
	for (uint32_t si = 0; si < n_starts; si++) {
		const uint32_t sv = __builtin_amdgcn_readfirstlane(starts[si]);
		uint32_t j = __builtin_amdgcn_readfirstlane(vert2node[sv]);
		const uint32_t chain_begin = ni;
		uint32_t sp = 0, last = kNone, prev_t_b = 0, adjusted = sv;
		uint32_t prevt = 0;        // previous symbol is a live 't' that popped the 'b' item prev_t_b
		uint32_t rib = 0;          // chain began with 'b', no other 'b' and no 't' yet
		uint32_t pend = 0;         // edge of node j consumed by the move that led here
		for (;;) {
			if (ni + 4u > icap) { err |= TRAIL_ERR_CAPACITY; break; }
			dbg_iters++;
			uint32_t av_raw, e_lo, e_hi;
			tab.load(j, av_raw, e_lo, e_hi);
			const uint32_t av = av_raw & ~pend;
			if (av == 0) {
				if (pend) tab.set_adj(j, 0u, l0);
				pend = 0;
				// ---- 't': dead end, back to the most recent branch vertex
				if (sp == 0) break;
				sp--;
				uint32_t pj, pitem;
				if (sp < lds_stack_cap) {
					const uint2 pr = s_stack2[sp];
					pj = __builtin_amdgcn_readfirstlane(pr.x);
					pitem = __builtin_amdgcn_readfirstlane(pr.y);
				}
				else {
					pj = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st_node + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					pitem = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st_item + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				}
				if (rib) {
					// remove_initial_branch (crackcodes.hpp:185-242): the leading 'b' and this 't'
					// vanish, the first stretch is walked backwards from where it ended
					adjusted = __builtin_amdgcn_readfirstlane(a.node_vertex[nb + j]);
					uint32_t lc = last;
					if (l0) {
						items[chain_begin] = kItemDead;
						if (ni > chain_begin + 1u) {
							uint32_t lo = chain_begin + 1u, hi = ni - 1u;
							while (lo < hi) {
								const uint32_t x = items[lo], y = items[hi];
								items[lo] = kItemSeg | tab.get_end(y & ~kItemMask);
								items[hi] = kItemSeg | tab.get_end(x & ~kItemMask);
								lo++; hi--;
							}
							if (lo == hi) items[lo] = kItemSeg | tab.get_end(items[lo] & ~kItemMask);
							const uint32_t e = tab.get_end(items[ni - 1u] & ~kItemMask);
							lc = (e & 3u) ^ 1u;
						}
					}
					last = __builtin_amdgcn_readfirstlane(lc);
					rib = 0;
				}
				else if (prevt) {
					// remove_spurious_branches (crackcodes.hpp:250-281): the 'b' popped by the
					// previous 't' and this 't' vanish
					items[prev_t_b] = kItemDead;
					prev_t_b = pitem;
				}
				else {
					const bool alt = (last == kNone) || (last == kUp);
					items[ni] = alt ? kTalt : kT;
					ni++;
					last = alt ? kLeft : kUp;
					prevt = 1;
					prev_t_b = pitem;
				}
				j = pj;
				continue;
			}
			pend = 0;
			// ---- along the lowest-numbered remaining edge: right, left, down, up
			const uint32_t k = __ffs(av) - 1;
			tab.set_adj(j, av & ~(1u << k), l0);
			const uint32_t e = tab.pick(e_lo, e_hi, k);
			const uint32_t seg = kItemSeg | (j * 4u + k);
			if (av & (av - 1u)) {
				// ---- 'b' first: more than one edge left, remember the vertex
				if (sp < lds_stack_cap) s_stack2[l0 ? sp : lds_stack_cap + threadIdx.x] = make_uint2(j, ni);
				else if (sp < scap) { if (l0) { st_node[sp] = j; st_item[sp] = ni; } }
				else err |= TRAIL_ERR_CAPACITY;
				sp++;
				const bool alt = (last == kNone) || (last == kDown);
				Item2 two;
				two.a = alt ? kBalt : kB; two.b = seg;
				*reinterpret_cast<Item2*>(items + ni) = two;        // both items in one store
				ni += 2;
				rib = (last == kNone) ? 1u : 0u;
			}
			else { items[ni] = seg; ni++; }
			prevt = 0;
			const uint32_t k2 = e & 3u;
			last = k2 ^ 1u;
			pend = 1u << k2;
			j = e >> 2;
		}
		// the closing 't' (crackcodes.hpp:436-439)
		if (prevt) items[prev_t_b] = kItemDead;
		else {
			const bool alt = (last == kNone) || (last == kUp);
			if (ni < icap) items[ni] = alt ? kTalt : kT;
			else err |= TRAIL_ERR_CAPACITY;
			ni++;
		}
		if (nch < kcap) { ch_node[nch] = adjusted; ch_item0[nch] = chain_begin; }
		else err |= TRAIL_ERR_CAPACITY;
		nch++;
		if (err) break;
	}
	if (l0 && a.dbg) {
		atomicAdd(a.dbg + 8, static_cast<unsigned long long>(dbg_iters));
		atomicAdd(a.dbg + 9, __builtin_amdgcn_s_memtime() - dbg_t0);
		atomicAdd(a.dbg + 10, __builtin_amdgcn_s_memrealtime() - dbg_r0);
		atomicAdd(a.dbg + 11, 1ull);
	}
	if (l0) {
		a.n_items[zi] = ni < icap ? ni : icap;
		a.n_chains[zi] = nch < kcap ? nch : kcap;
		if (err) atomicOr(a.slice_err + zi, err);
	}
}

// grid = nslices, block = one wavefront; dynamic LDS = lds_bytes
static __global__ void __launch_bounds__(kWave) k_trail_dfs(TrailArgs a, uint32_t lds_bytes) {
	extern __shared__ uint32_t s_trail[];
	const uint32_t zi = blockIdx.x + a.z0;
	const uint32_t nn = min(a.n_nodes[zi], a.ncap[zi]);
	const uint64_t nb = a.nbase[zi];
	if (a.slice_err[zi]) {
		if (threadIdx.x == 0) { a.n_items[zi] = 0; a.n_chains[zi] = 0; }
		return;
	}
	// LDS: [dart ends u16 x 4 nn][adj u8 x nn][64 scratch dwords][branch stack: (node, item) pairs + 64 scratch pairs]
	const uint32_t tab_bytes = ((nn * 9u + 15u) / 16u) * 16u + 256u;
	const bool in_lds = nn < 16384u && tab_bytes + 2048u + 512u <= lds_bytes;
	if (in_lds) {
		TrailTabLds t;
		t.end4 = reinterpret_cast<unsigned long long*>(s_trail);
		t.adj = reinterpret_cast<uint8_t*>(s_trail) + nn * 8u;
		t.dummy = (tab_bytes - 256u) - nn * 8u;
		uint16_t* e16 = reinterpret_cast<uint16_t*>(s_trail);
		for (uint32_t d = threadIdx.x; d < nn * 4u; d += kWave) {
			const uint32_t e = a.dart_end[nb * 4u + d];
			e16[d] = static_cast<uint16_t>(e == kDartNone ? 0xFFFFu : e);
		}
		for (uint32_t j = threadIdx.x; j < nn; j += kWave) t.adj[j] = a.node_adj[nb + j];
		__syncthreads();
		uint32_t* stack = s_trail + tab_bytes / 4u;
		trail_dfs_slice<TrailTabLds>(a, zi, t, stack, (lds_bytes - tab_bytes) / 8u - 64u);
	}
	else {
		TrailTabGlobal t;
		t.adj = a.node_adj + nb;
		t.end = a.dart_end + nb * 4u;
		trail_dfs_slice<TrailTabGlobal>(a, zi, t, s_trail, lds_bytes / 8u - 64u);
	}
}

// grid = nslices: code offset of every item, chain offsets and lengths
static __global__ void __launch_bounds__(kBlock) k_trail_offsets(TrailArgs a) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.x + a.z0;
	const uint32_t n = a.n_items[zi];
	const uint32_t* items = a.items + a.ibase[zi];
	uint32_t* off = a.item_off + a.ibase[zi];
	const uint32_t* dlen = a.dart_len + a.nbase[zi] * 4u;
	constexpr uint32_t kPer = 4;
	uint32_t carry = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += kBlock * kPer) {
		uint32_t len[kPer], cnt = 0;
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t i = i0 + threadIdx.x * kPer + q;
			len[q] = 0;
			if (i < n) {
				const uint32_t it = items[i];
				const uint32_t kind = it & kItemMask;
				len[q] = kind == kItemSeg ? dlen[it & ~kItemMask] : (kind == kItemCtl ? 2u : 0u);
			}
			cnt += len[q];
		}
		uint32_t v[1] = { cnt }, tot[1];
		block_excl_add<1>(v, tot, s_scan);
		uint32_t o = carry + v[0];
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t i = i0 + threadIdx.x * kPer + q;
			if (i < n) off[i] = o;
			o += len[q];
		}
		carry += tot[0];
	}
	__syncthreads();
	__threadfence_block();
	const uint32_t total = carry;
	const uint32_t nch = a.n_chains[zi];
	const uint32_t* item0 = a.chain_item0 + a.kbase[zi];
	uint32_t* ch_off = a.chain_off + a.kbase[zi];
	uint32_t* ch_clen = a.chain_clen + a.kbase[zi];
	for (uint32_t c = threadIdx.x; c < nch; c += kBlock) {
		const uint32_t o0 = item0[c] < n ? off[item0[c]] : total;
		const uint32_t o1 = (c + 1 < nch && item0[c + 1] < n) ? off[item0[c + 1]] : total;
		ch_off[c] = o0;
		ch_clen[c] = o1 - o0;
	}
	if (threadIdx.x == 0) {
		a.n_raw[zi] = total;
		a.n_valid[zi] = total;
		if (total > a.ccap[zi]) atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY);
	}
}

// grid = (ceil(max items / (kExpandChunk * 4)), nslices): every item writes its code points;
// segment items are re-walked, lanes refill from the wavefront's range like k_trail_segments
static __global__ void __launch_bounds__(kBlock) k_trail_expand(TrailArgs a) {
	const uint32_t zi = blockIdx.y + a.z0;
	if (a.slice_err[zi]) return;
	const uint32_t lane = threadIdx.x & (kWave - 1);
	const uint32_t wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
	const uint32_t n_items = a.n_items[zi];
	uint32_t next = wave * kExpandChunk;
	const uint32_t range_end = min(next + kExpandChunk, n_items);
	if (next >= range_end) return;
	const uint64_t nb = a.nbase[zi];
	const uint32_t* items = a.items + a.ibase[zi];
	const uint32_t* item_off = a.item_off + a.ibase[zi];
	uint8_t* cp0 = a.cp + a.cbase[zi];
	const uint4* adjm = a.adjm + zi * a.adjm_stride;
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	MicroTile c;
	c.lo = c.hi = make_uint4(0, 0, 0, 0); c.mx = c.my = 0xFFFFFFFFu;
	bool active = false, need_nib = false;
	// at vertex (x, y); k: direction of the next move, or (need_nib) of the move that led here
	uint32_t x = 0, y = 0, k = 0, left = 0, nacc = 0;
	unsigned long long acc = 0;
	uint8_t* cp = cp0;
	for (;;) {
		const unsigned long long need = __ballot(!active);
		const bool refill = next < range_end && (__popcll(need) >= kWalkRefill || need == ~0ull);
		const uint32_t cand = next + static_cast<uint32_t>(__popcll(need & lt_mask));
		const bool take = refill && !active && cand < range_end;
		uint32_t r_it = kItemDead, r_off = 0;
		if (take) { r_it = items[cand]; r_off = item_off[cand]; }
		if (active && need_nib && !c.holds(x, y)) mt_load(c, adjm, x, y, a.mtx2);
		for (int it = 0; it < kWalkAhead; it++) {
			const bool go = active && (!need_nib || c.holds(x, y));
			if (!__ballot(go)) break;
			if (go) {
				if (need_nib) { k = __ffs(c.nib(x, y) & ~(1u << (k ^ 1u))) - 1; need_nib = false; }
				// eight code points per (unaligned) 8-byte store
				acc |= static_cast<unsigned long long>(trail_code(k)) << (8u * nacc);
				nacc++;
				trail_step(x, y, k);
				--left;
				if (nacc == 8u || left == 0) {
					if (nacc == 8u) { struct __attribute__((packed)) U64 { unsigned long long v; }; reinterpret_cast<U64*>(cp)->v = acc; }
					else for (uint32_t q = 0; q < nacc; q++) cp[q] = static_cast<uint8_t>(acc >> (8u * q));
					cp += nacc; acc = 0; nacc = 0;
				}
				if (left == 0) active = false;
				else need_nib = true;
			}
		}
		if (take) {
			const uint32_t kind = r_it & kItemMask;
			if (kind == kItemCtl) {
				uint8_t* o = cp0 + r_off;
				o[0] = static_cast<uint8_t>(r_it & 3u); o[1] = static_cast<uint8_t>((r_it >> 2) & 3u);
			}
			else if (kind == kItemSeg) {
				const uint32_t d = r_it & ~kItemMask;
				left = a.dart_len[nb * 4u + d];
				if (left) {
					const uint32_t v0 = a.node_vertex[nb + (d >> 2)];
					y = v0 / a.sxe; x = v0 - y * a.sxe;
					k = d & 3u;
					cp = cp0 + r_off;
					need_nib = false;
					active = true;
				}
			}
		}
		if (refill) next += static_cast<uint32_t>(__popcll(need));
		if (!__ballot(active) && next >= range_end) break;
	}
}

}  // namespace dev
}  // namespace ckl
