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
//   k_trail_walk       the exact serial trail, but over nodes only (tables in LDS) and reduced to
//                      WHERE it goes: one event per step (segment, 'b' + segment, dead end)
//   k_trail_items      events -> items (segment / 'b' / 't'), for all steps of a slice at once: forms
//                      of 'b' and 't', remove_initial_branch (185-242), remove_spurious_branches
//                      (250-281)
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
constexpr uint32_t kTrailTileShift = 5, kTrailTileDim = 32;        // thread mapping of k_trail_graph / k_trail_nodes: one thread per piece of 32 x 8 vertices
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
	uint32_t tiles_x, tiles_y;   // pieces of 32 x 8 vertices per row of pieces / rows of pieces: thread mapping of k_trail_graph / k_trail_nodes
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
	uint4* dart_codes;           // the segment's first kInlineCodes code points, 2 bits each (code i at bit 2 i)
	uint8_t* dart_inline;        // 1: dart_codes holds the whole segment (k_trail_expand copies instead of walking)
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
	uint32_t* stack_item;        // event of the entry's kEvBseg
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
	uint32_t* events;            // [at ibase] k_trail_walk -> k_trail_items
	uint32_t* n_events;          // [nslices]
	uint32_t* seg_len_sum;       // [nslices] sum of the darts' segment lengths (k_trail_segments): 2 x the crack edges unless some lie on node-free loops
	uint32_t* chain_ev0;         // [at kbase] first event of the chain
	uint32_t* ev_lnd;            // [at ibase] scratch of k_trail_items
	uint32_t* ev_item;
	uint32_t walk_plain;         // testing (CKL_TRAIL_WALK=plain): trail_walk_slice for every slice
	uint32_t walk_stack_cap;     // testing (CKL_TRAIL_WALK_STACK=n): branch stack of the hand-scheduled walk capped at n entries
	uint32_t walk_no_regs;       // testing (CKL_TRAIL_WALK=lds): the hand-scheduled walks on LDS tables only, not the one on register tables
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

// The edge bits of the eight vertex rows y0 .. y0 + 7 (y0 a multiple of 8) of the 32 vertices x = 32 tx + i: what
// trail_row_bits gives row by row, with all 25 plane words requested up front — from addresses clamped into the slice, the
// words masked afterwards (a load behind a test of its own is waited for there).  Rows past sy come out empty.
__device__ __forceinline__ void trail_rows8(
	const uint32_t* __restrict__ pv, const uint32_t* __restrict__ ph, uint32_t row_words,
	uint32_t sx, uint32_t sy, uint32_t tx, uint32_t y0, uint32_t inv, TileRowBits (&b)[8]
) {
#pragma unroll
	for (int k = 0; k < 8; k++) b[k] = TileRowBits{ 0, 0, 0, 0 };
	if (row_words == 0 || sy == 0) return;
	const uint32_t x0 = tx << 5;
	const uint32_t lt_sx = x0 >= sx ? 0u : (sx - x0 >= 32u ? 0xFFFFFFFFu : ((1u << (sx - x0)) - 1u));        // x < sx
	const uint32_t le_sx = x0 > sx ? 0u : (sx - x0 >= 31u ? 0xFFFFFFFFu : ((2u << (sx - x0)) - 1u));        // x <= sx
	const uint32_t ge_1 = tx == 0 ? 0xFFFFFFFEu : 0xFFFFFFFFu;                                                // x >= 1
	const bool have_w = tx < row_words;
	const uint32_t txc = have_w ? tx : row_words - 1u, txp = tx > 0 ? min(tx - 1u, row_words - 1u) : 0u;
	uint32_t hw[8], hpw[8], vw[9];
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint64_t row = static_cast<uint64_t>(min(y0 + k, sy - 1u)) * row_words;
		hw[k] = ph[row + txc];
		hpw[k] = ph[row + txp];
	}
#pragma unroll
	for (uint32_t k = 0; k < 9; k++) {      // plane V rows y0 - 1 .. y0 + 7
		const uint32_t y = y0 + k;           // (row + 1)
		vw[k] = pv[static_cast<uint64_t>(min(y > 0 ? y - 1u : 0u, sy - 1u)) * row_words + txc];
	}
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t y = y0 + k;
		const uint32_t h = have_w ? hw[k] : 0u, hp = tx > 0 ? hpw[k] : 0u;
		if (y >= 1 && y < sy) {
			b[k].R = (h ^ inv) & lt_sx;
			b[k].L = (((h << 1) | (hp >> 31)) ^ inv) & ge_1 & le_sx;
		}
		if (y < sy) b[k].D = ((have_w ? vw[k + 1] : 0u) ^ inv) & ge_1 & lt_sx;
		if (y >= 1 && y <= sy) b[k].U = ((have_w ? vw[k] : 0u) ^ inv) & ge_1 & lt_sx;
	}
}

// grid = (graph_blocks, nslices).  One thread per piece of 32 x 8 vertices = one row of four micro-tiles, the pieces of a
// slice numbered row by row (tiles_x per row of pieces, tiles_y = rows of pieces): neighbouring lanes read neighbouring
// plane words, and a thread writes whole micro-tiles (2 x 16 bytes each).  (Until round 5: one thread per 32 x 1 vertices,
// four loads, four 4-byte stores — 70 000 workgroups of C2 that lived for one trip to memory each: 0.18 ms for 0.4 GB.)
static __global__ void __launch_bounds__(kBlock) k_trail_graph(
	const uint32_t* __restrict__ planeV, const uint32_t* __restrict__ planeH, uint32_t row_words, uint64_t plane_words,
	uint32_t sx, uint32_t sy, uint32_t perm_mode, const unsigned long long* __restrict__ total_pairs, unsigned long long half_voxels,
	uint32_t* __restrict__ adjm_words, uint64_t adjm_stride_words,
	uint32_t mtx2, uint32_t tiles_x, uint32_t tiles_y, uint32_t* __restrict__ blk_special, uint32_t* __restrict__ blk_corner
) {
	__shared__ uint32_t s_red[2 * kWaves];
	// A thread's four micro-tiles are eight 16-byte pieces in two stretches of 64 bytes; stored by their owner, a store instruction
	// is 64 pieces of 16 bytes in 64 different lines (16.8 M write requests at C2: the L2s' request rate, 0.09 of the kernel's
	// 0.16 ms).  They go through LDS instead, so that four neighbouring lanes store one stretch of 64 bytes.
	__shared__ uint4 s_out[kWaves][kWave * 8];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t zi = blockIdx.y;
	// perm_mode 2: the crack format follows from the volume's equal pixel pairs (crackle.hpp:50-55), counted by the kernels in front
	const bool permissible = perm_mode == 2u ? static_cast<long long>(*total_pairs) < static_cast<long long>(half_voxels) : perm_mode != 0u;
	const uint32_t item = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t mtx = (sx + 1u + 7u) >> 3;
	uint32_t ns = 0, nc = 0;
	uint32_t my_base = 0, my_tiles = 0;      // byte offset of my first micro-tile in the slice's graph, how many of my four exist
	if (item < tiles_x * tiles_y) {
		const uint32_t ty = item / tiles_x, tx = item - ty * tiles_x;
		TileRowBits b[8];
		trail_rows8(planeV + zi * plane_words, planeH + zi * plane_words, row_words, sx, sy, tx, ty << 3, permissible ? 0xFFFFFFFFu : 0u, b);
		my_base = mt_index(tx * 4u, ty, mtx2) * 32u;
		my_tiles = min(4u, mtx - min(mtx, tx * 4u));
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			uint32_t w[8];
#pragma unroll
			for (uint32_t k = 0; k < 8; k++)
				w[k] = spread8((b[k].R >> (8 * q)) & 255u) | (spread8((b[k].L >> (8 * q)) & 255u) << 1)
					| (spread8((b[k].D >> (8 * q)) & 255u) << 2) | (spread8((b[k].U >> (8 * q)) & 255u) << 3);
			// piece p = 2 q + half of lane l sits at 8 l + (p ^ (l & 7)): eight neighbouring lanes hit eight different bank groups
			s_out[wave][lane * 8 + ((2u * q) ^ (lane & 7))] = make_uint4(w[0], w[1], w[2], w[3]);
			s_out[wave][lane * 8 + ((2u * q + 1u) ^ (lane & 7))] = make_uint4(w[4], w[5], w[6], w[7]);
		}
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) { ns += __popc(trail_special_mask(b[k])); nc += __popc(trail_corner_mask(b[k])); }
	}
	{
		// (a wavefront's own pieces: no barrier, LDS operations of a wavefront complete in order)
		uint8_t* dst = reinterpret_cast<uint8_t*>(adjm_words + zi * adjm_stride_words);
#pragma unroll
		for (uint32_t i = 0; i < 8; i++) {
			const uint32_t owner = i * 8u + (lane >> 3), p = lane & 7u;
			const uint32_t base = __shfl(my_base, owner, kWave), have = __shfl(my_tiles, owner, kWave);
			const uint4 v = s_out[wave][owner * 8u + (p ^ (owner & 7u))];
			const uint32_t q = p >> 1;
			if (q < have) *reinterpret_cast<uint4*>(dst + base + (q >> 1) * 128u + (q & 1u) * 32u + (p & 1u) * 16u) = v;
		}
	}
	ns = wave_sum(ns); nc = wave_sum(nc);
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

// grid = (graph_blocks, nslices), same thread -> piece mapping as k_trail_graph:
// nodes are numbered in (workgroup, piece, row, x) order, no atomics
static __global__ void __launch_bounds__(kBlock) k_trail_nodes(TrailArgs a) {
	__shared__ uint32_t s_scan[2 * kWaves];
	const uint32_t zi = blockIdx.y + a.z0;
	const uint32_t item = blockIdx.x * kBlock + threadIdx.x;
	TileRowBits b[8];
	uint32_t x0 = 0, y0 = 0;
	if (item < a.tiles_x * a.tiles_y) {
		const uint32_t ty = item / a.tiles_x, tx = item - ty * a.tiles_x;
		x0 = tx << kTrailTileShift; y0 = ty << 3;
		trail_rows8(a.planeV + zi * a.plane_words, a.planeH + zi * a.plane_words, a.row_words, a.sx, a.sy, tx, y0, a.inv, b);
	}
	else {
#pragma unroll
		for (int k = 0; k < 8; k++) b[k] = TileRowBits{ 0, 0, 0, 0 };
	}
	uint32_t v[2] = { 0, 0 }, tot[2];
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) { v[0] += __popc(trail_special_mask(b[k])); v[1] += __popc(trail_corner_mask(b[k])); }
	block_excl_add<2>(v, tot, s_scan);
	uint32_t j = a.blk_special[static_cast<uint64_t>(zi) * a.graph_blocks + blockIdx.x] + v[0];
	uint32_t c = a.blk_corner[static_cast<uint64_t>(zi) * a.graph_blocks + blockIdx.x] + v[1];
	const uint64_t nb = a.nbase[zi];
	const uint32_t ncap = a.ncap[zi], cocap = a.cocap[zi];
	uint32_t err = 0;
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t y = y0 + k;
		const TileRowBits& r = b[k];
		for (uint32_t m = trail_special_mask(r); m; m &= m - 1u) {
			const uint32_t i = __ffs(m) - 1;
			const uint32_t vtx = y * a.sxe + x0 + i;
			if (j < ncap) {
				a.node_vertex[nb + j] = vtx;
				a.node_adj[nb + j] = static_cast<uint8_t>(((r.R >> i) & 1u) | (((r.L >> i) & 1u) << 1) | (((r.D >> i) & 1u) << 2) | (((r.U >> i) & 1u) << 3));
				a.vert2node[static_cast<uint64_t>(zi) * a.nverts + vtx] = j;
			}
			else err = TRAIL_ERR_CAPACITY;
			j++;
		}
		for (uint32_t m = trail_corner_mask(r); m; m &= m - 1u) {
			const uint32_t i = __ffs(m) - 1;
			if (c < cocap) a.corner_vertex[a.cobase[zi] + c] = y * a.sxe + x0 + i;
			else err = TRAIL_ERR_CAPACITY;
			c++;
		}
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
constexpr int kWalkRefill = 8;      // idle lanes that trigger a refill (its loads cost as much as a step); round 3 with the code points kept: 8 and 2 below beat 16 and 3 (0.47 against 0.51 ms)
constexpr int kWalkAhead = 2;       // steps a lane may run ahead inside its micro-tile per memory round trip
constexpr uint32_t kInlineCodes = 64;      // code points of a segment that k_trail_segments keeps beside the dart (16 bytes)

// Results are staged in LDS and written out once per wavefront: on this architecture loads
// and stores retire in order on one counter, so a store inside the loop would make every
// following step wait for it.
// grid = (ceil(4 * max nodes / (kWalkChunk * 4)), nslices)
static __global__ void __launch_bounds__(kBlock) k_trail_segments(TrailArgs a) {
	__shared__ uint4 s_res[kWaves][kWalkChunk];       // end, len, minv, minpos
	__shared__ uint4 s_codes[kWaves][kWalkChunk];     // the walk's code points
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
	uint4* res_codes = s_codes[wv];
	for (uint32_t i = lane; i < kWalkChunk; i += kWave) res[i] = make_uint4(kDartNone, 0u, 0u, 0u);
	MicroTile c;
	c.lo = c.hi = make_uint4(0, 0, 0, 0); c.mx = c.my = 0xFFFFFFFFu;
	bool active = false;
	// (x, y): the vertex to look at next, reached by a move in direction k after `steps` edges
	uint32_t d = 0, x = 0, y = 0, k = 0, steps = 0, minv = 0, minpos = 0, err = 0;
	// the last kInlineCodes code points, newest in the top bits: a move shifts the 128 bits down by two
	uint4 cq = make_uint4(0, 0, 0, 0);
	auto push_code = [&](uint32_t kk) {
		cq.x = __builtin_amdgcn_alignbit(cq.y, cq.x, 2);
		cq.y = __builtin_amdgcn_alignbit(cq.z, cq.y, 2);
		cq.z = __builtin_amdgcn_alignbit(cq.w, cq.z, 2);
		cq.w = (cq.w >> 2) | (trail_code(kk) << 30);
	};
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
				else { k = __ffs(nib & ~(1u << arr)) - 1; push_code(k); trail_step(x, y, k); steps++; }
				if (fin) {
					// x: end vertex << 2 | arrival edge (the vertex becomes a node index in the flush below)
					res[d - range_begin] = make_uint4((w << 2) | arr, steps, minv, minpos);
					res_codes[d - range_begin] = cq;
					active = false;
				}
			}
		}
		if (take && ((r_adj >> (cand & 3u)) & 1u)) {
			d = cand; k = cand & 3u;
			y = r_v0 / a.sxe; x = r_v0 - y * a.sxe;
			minv = r_v0; minpos = 0;
			push_code(k);
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
	uint32_t len_sum = 0;
	for (uint32_t i = lane; i < range_end - range_begin; i += kWave) {
		const uint4 r = res[i];
		a.dart_end[db + i] = r.x == kDartNone ? kDartNone : ((v2n[r.x >> 2] << 2) | (r.x & 3u));
		a.dart_len[db + i] = r.y;
		a.dart_minv[db + i] = r.z;
		a.dart_minpos[db + i] = r.w;
		// the first code point to bit 0: the newest sits at bit 126
		const uint4 q = res_codes[i];
		const bool whole = r.x != kDartNone && r.y >= 1u && r.y <= kInlineCodes;
		uint4 o = make_uint4(0, 0, 0, 0);
		if (whole) {
			const uint32_t sh = 128u - 2u * r.y, w = sh >> 5, b = sh & 31u;
			const uint32_t v[5] = { q.x, q.y, q.z, q.w, 0u };
			auto word = [&](uint32_t t) -> uint32_t { return t == 0 ? v[0] : t == 1 ? v[1] : t == 2 ? v[2] : t == 3 ? v[3] : 0u; };
			auto pick = [&](uint32_t t) -> uint32_t { return __builtin_amdgcn_alignbit(word(t + w + 1u), word(t + w), b); };
			o = make_uint4(pick(0), pick(1), pick(2), pick(3));
			(void)v;
		}
		a.dart_codes[db + i] = o;
		a.dart_inline[db + i] = whole ? 1u : 0u;
		len_sum += r.x == kDartNone ? 0u : r.y;
	}
	len_sum = wave_sum(len_sum);
	if (lane == 0 && len_sum) atomicAdd(a.seg_len_sum + zi, len_sum);
	if (err) atomicOr(a.slice_err + zi, err);
}

// grid = (ceil(max corners / 256), nslices): a "right+down" corner is the start of a
// closed loop when following the loop from it never meets a node or a smaller vertex
static __global__ void __launch_bounds__(kBlock) k_trail_loops(TrailArgs a) {
	const uint32_t zi = blockIdx.y + a.z0;
	// every crack edge lies on a segment between nodes (each counted from both of its darts): the slice
	// has no node-free loop, there is nothing to look for
	if (a.seg_len_sum[zi] == 2u * (a.max_steps[zi] - 1u)) return;
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
	for (uint32_t q = 0; q < 4; q++) { a.dart_end[db + q] = kDartNone; a.dart_len[db + q] = 0; a.dart_minv[db + q] = v0; a.dart_minpos[db + q] = 0; a.dart_inline[db + q] = 0; }
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
		for (uint32_t q = 0; q < 4; q++) { a.dart_end[db + s * 4u + q] = kDartNone; a.dart_len[db + s * 4u + q] = 0; a.dart_inline[db + s * 4u + q] = 0; }      // (the two halves keep their codes: a prefix)
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

// ---- node tables of the walk ------------------------------------------------------
// The walk runs in lane 0 of one wavefront per slice with wave-uniform scalar state (values read
// from memory go through readfirstlane).  Node tables: LDS (16-bit dart ends) when the slice has
// few enough nodes, else the global arrays.  The remaining-edge nibble and the four dart ends of a
// node are fetched together; the edge consumed on arrival is carried in a register (pend) instead
// of being cleared at the far node first.
enum : uint32_t { TCODE_UP = 0, TCODE_RIGHT = 1, TCODE_DOWN = 2, TCODE_LEFT = 3, TCODE_NONE = 0xFE };

struct TrailTabLds {
	uint8_t* adj;
	unsigned long long* end4;      // four 16-bit dart ends (node << 2 | arrival) per node
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
	__device__ __forceinline__ void set_adj(uint32_t j, uint32_t v) const { adj[j] = static_cast<uint8_t>(v); }
};
struct TrailTabGlobal {
	uint8_t* adj;
	const uint32_t* end;
	__device__ __forceinline__ void load(uint32_t j, uint32_t& av, uint32_t& e_lo, uint32_t& e_hi) const {
		// word loads that bypass the vector L1: the lane's own stores must be read back as written
		const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(adj + j) & ~static_cast<uintptr_t>(3));
		const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(adj + j) & 3u) * 8u;
		av = (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> sh) & 0xFFu;
		e_lo = j; e_hi = 0;
	}
	__device__ __forceinline__ uint32_t pick(uint32_t e_lo, uint32_t e_hi, uint32_t k) const { return __builtin_amdgcn_readfirstlane(end[e_lo * 4u + k]); }
	__device__ __forceinline__ void set_adj(uint32_t j, uint32_t v) const {
		// read-modify-write of the containing word (single writer per slice, slices are 16-byte aligned)
		uint32_t* w = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(adj + j) & ~static_cast<uintptr_t>(3));
		const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(adj + j) & 3u) * 8u;
		const uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(w, (old & ~(0xFFu << sh)) | ((v & 0xFFu) << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
};

// ---- the walk alone -----------------------------------------------------------------
// What is serial in the trail is only WHERE it goes: which edge is taken at a node, where a dead
// end jumps back to.  That depends on the remaining-edge nibbles and the branch stack, never on what
// was emitted.  The forms of 'b' / 't', remove_initial_branch and remove_spurious_branches are
// functions of the sequence of steps, and k_trail_items computes them for all steps at once.  So the
// walk emits one EVENT per step and nothing else:
//   kEvSeg  | dart        along the only remaining edge of the node
//   kEvBseg | dart        more than one edge left: the node goes on the branch stack first ('b')
//   kEvDead | event       dead end: back to the most recent branch node; `event` = its kEvBseg
//   kEvEnd                dead end with an empty stack: the chain is complete
constexpr uint32_t kEvSeg = 0u << 30, kEvBseg = 1u << 30, kEvDead = 2u << 30, kEvEnd = 3u << 30, kEvMask = 3u << 30;
// n_events' top bit: the segment events of the slice carry 12 * node | edge instead of the dart 4 * node + edge
constexpr uint32_t kEvFormatAddr12 = 1u << 31;
constexpr uint32_t kEvWalkWide = 1u << 30;       // on a slice's event count: walked by trail_walk_chain_wide (events in the plain format; for ckl_encoder_walk_paths)

// Runs in lane 0.  Branch stack: entry q in LDS slot q + 1 (slot 0 takes the stores of steps that
// push nothing), entries beyond the LDS part in stack_node / stack_item.
template <typename TAB>
__device__ __forceinline__ void trail_walk_slice(
	const TrailArgs& a, uint32_t zi, const TAB& tab, uint2* s_stack2, uint32_t lds_slots
) {
	const uint64_t nb = a.nbase[zi];
	const uint32_t* starts = a.starts + nb;
	const uint32_t n_starts = a.n_starts[zi];
	uint32_t* ev = a.events + a.ibase[zi];
	const uint32_t ecap = a.icap[zi];
	uint32_t* st_node = a.stack_node + a.sbase[zi];
	uint32_t* st_ev = a.stack_item + a.sbase[zi];
	const uint32_t scap = a.scap[zi];
	uint32_t* ch_node = a.chain_node + a.kbase[zi];
	uint32_t* ch_ev0 = a.chain_ev0 + a.kbase[zi];
	const uint32_t kcap = a.kcap[zi];
	const uint32_t* vert2node = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
	const uint32_t cap1 = lds_slots - 1u;

	uint32_t ne = 0, nch = 0, err = 0;
	const unsigned long long dbg_t0 = (kTuning && a.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;
	const unsigned long long dbg_r0 = (kTuning && a.dbg) ? __builtin_amdgcn_s_memrealtime() : 0ull;
	for (uint32_t si = 0; si < n_starts && !err; si++) {
		const uint32_t sv = __builtin_amdgcn_readfirstlane(starts[si]);
		uint32_t j = __builtin_amdgcn_readfirstlane(vert2node[sv]);
		if (nch < kcap) { ch_node[nch] = sv; ch_ev0[nch] = ne; }
		else err |= TRAIL_ERR_CAPACITY;
		nch++;
		uint32_t sp = 0, pend = 0;        // pend: edge of node j consumed by the move that led here
		for (;;) {
			if (ne + 2u > ecap) { err |= TRAIL_ERR_CAPACITY; break; }
			uint32_t av_raw, e_lo, e_hi;
			tab.load(j, av_raw, e_lo, e_hi);
			const uint2 top = s_stack2[sp < cap1 ? sp : cap1];
			const uint32_t av = av_raw & ~pend;
			if (av == 0) {
				tab.set_adj(j, 0u);
				if (sp == 0) { ev[ne] = kEvEnd; ne++; break; }
				uint32_t pj = __builtin_amdgcn_readfirstlane(top.x), pev = __builtin_amdgcn_readfirstlane(top.y);
				if (sp > cap1) {
					pj = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st_node + sp - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					pev = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st_ev + sp - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				}
				ev[ne] = kEvDead | pev;
				ne++; sp--; pend = 0; j = pj;
				continue;
			}
			// along the lowest-numbered remaining edge: right, left, down, up
			const uint32_t k = __ffs(av) - 1;
			const uint32_t rest = av & ~(1u << k);
			tab.set_adj(j, rest);
			const uint32_t e = tab.pick(e_lo, e_hi, k);
			const bool multi = rest != 0;
			if (multi && sp >= cap1) {
				if (sp < scap) { st_node[sp] = j; st_ev[sp] = ne; }
				else err |= TRAIL_ERR_CAPACITY;
			}
			s_stack2[(multi && sp < cap1) ? sp + 1u : 0u] = make_uint2(j, ne);
			ev[ne] = (multi ? kEvBseg : kEvSeg) | (j * 4u + k);
			ne++;
			sp += multi ? 1u : 0u;
			pend = 1u << (e & 3u);
			j = e >> 2;
		}
	}
	if (kTuning && a.dbg) {
		atomicAdd(a.dbg + 8, static_cast<unsigned long long>(ne));
		atomicAdd(a.dbg + 9, __builtin_amdgcn_s_memtime() - dbg_t0);
		atomicAdd(a.dbg + 10, __builtin_amdgcn_s_memrealtime() - dbg_r0);
		atomicAdd(a.dbg + 11, 1ull);
	}
	a.n_events[zi] = ne < ecap ? ne : ecap;
	a.n_chains[zi] = nch < kcap ? nch : kcap;
	if (err) atomicOr(a.slice_err + zi, err);
}

// The walk of one chain in hand-scheduled code.  A single wavefront issues one instruction every
// ~4.5 cycles whatever its kind, pays ~22 for a taken branch, ~50 for an LDS round trip and ~25 more for
// the hop VGPR -> SGPR (tools/micro/wave_latency.hip), and hipcc's version of trail_walk_slice spends
// 70 instructions and 3-4 taken branches on a step (~550 cycles).  Here a step is 25-35 instructions:
//   node records of 12 bytes at LDS address 12 * node: { remaining edges, ends 0|1, ends 2|3 },
//   an end = 12 * node at the far end | edge it arrives by, so that the next record's address is one
//   AND away; the branch stack's top entry (4 bytes: event << 16 | node address) is fetched with the
//   record; event words go out through a VGPR byte offset that doubles as the event's index on the
//   branch stack.  A segment event carries 12 * node | edge (kEvFormatAddr12: k_trail_items divides).
// 16 bytes of LDS per node in all, so that two walks leave a third of a CU's LDS to the kernels of the
// label stream, which are scheduled beside them.
// Needs 12 * nodes < 65536, fewer than 65536 events and the dynamic LDS at address 0.  Returns 0 (chain
// complete), 1 (events full) or 2 (branch stack beyond its LDS part: the caller walks the slice again
// with the compiled walk).  Lane 0 only.
__device__ __forceinline__ uint32_t trail_walk_chain_fast(
	uint32_t j_addr, uint32_t& ev_off, uint32_t& ev_left, uint32_t stack_base, uint32_t max_depth, uint32_t* ev
) {
	uint32_t status, off = ev_off, left = ev_left;
	const uint32_t top0 = stack_base - 4u;        // "top entry" of the empty stack: never used
	const uint32_t ev_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev)));
	const uint32_t ev_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev) >> 32));
	j_addr = __builtin_amdgcn_readfirstlane(j_addr);
	max_depth = __builtin_amdgcn_readfirstlane(max_depth);
	asm volatile(
		"s_setprio 3\n"
		"s_mov_b32 s40, %[j]\n"
		"s_mov_b32 s41, 0\n"                        // pend
		"s_mov_b32 s42, 0\n"                        // stack depth
		"v_readfirstlane_b32 s43, %[left]\n"
		"s_mov_b32 s52, %[evlo]\n"
		"s_mov_b32 s53, %[evhi]\n"
		"s_mov_b32 s54, %[maxd]\n"
		"v_mov_b32 v29, %[off]\n"
		"v_mov_b32 v25, %[top]\n"
		"1:\n"                                      // ---- a step
		"v_mov_b32 v20, s40\n"
		"ds_read2_b32 v[32:33], v20 offset1:1\n"    // remaining edges, ends 0|1
		"ds_read_b32 v34, v20 offset:8\n"           // ends 2|3
		"ds_read_b32 v26, v25\n"                    // top of the branch stack
		"s_sub_u32 s43, s43, 1\n"
		"s_cbranch_scc1 8f\n"                       // no room for another event
		"s_waitcnt lgkmcnt(0)\n"
		"v_readfirstlane_b32 s46, v32\n"
		"v_readfirstlane_b32 s44, v33\n"
		"v_readfirstlane_b32 s45, v34\n"
		"s_andn2_b32 s46, s46, s41\n"               // remaining edges without the one we came by
		"s_cbranch_scc0 4f\n"
		"s_ff1_i32_b32 s47, s46\n"                  // lowest-numbered edge: right, left, down, up
		"s_bitset0_b32 s46, s47\n"
		"v_mov_b32 v21, s46\n"
		"ds_write_b32 v20, v21\n"
		"s_lshl_b32 s48, s47, 4\n"
		"s_lshr_b64 s[44:45], s[44:45], s48\n"      // the edge's end in the low 16 bits
		"s_or_b32 s49, s40, s47\n"                  // 12 * node | edge
		"s_cmp_eq_u32 s46, 0\n"
		"s_cbranch_scc1 3f\n"
		"s_cmp_ge_u32 s42, s54\n"                   // ---- more edges left: the node goes on the branch stack
		"s_cbranch_scc1 9f\n"
		"s_bitset1_b32 s49, 30\n"                   // kEvBseg
		"v_add_u32 v25, 4, v25\n"
		"v_lshl_or_b32 v28, v29, 14, s40\n"         // this event's index << 16 | node address
		"s_add_u32 s42, s42, 1\n"
		"ds_write_b32 v25, v28\n"
		"3:\n"
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_and_b32 s48, s44, 3\n"
		"s_lshl_b32 s41, 1, s48\n"                  // the edge consumed at the far end
		"s_and_b32 s40, s44, 0xfffc\n"
		"s_branch 1b\n"
		"4:\n"                                      // ---- dead end
		"v_mov_b32 v21, 0\n"
		"ds_write_b32 v20, v21\n"
		"s_cmp_eq_u32 s42, 0\n"
		"s_cbranch_scc1 7f\n"
		"v_readfirstlane_b32 s49, v26\n"            // back to the most recent branch node
		"s_and_b32 s40, s49, 0xffff\n"
		"s_lshr_b32 s49, s49, 16\n"
		"s_bitset1_b32 s49, 31\n"                   // kEvDead | its kEvBseg
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"v_add_u32 v25, -4, v25\n"
		"s_sub_u32 s42, s42, 1\n"
		"s_mov_b32 s41, 0\n"
		"s_branch 1b\n"
		"7:\n"                                      // ---- the chain is complete
		"v_mov_b32 v30, 0xc0000000\n"               // kEvEnd
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_mov_b32 s48, 0\n"
		"s_branch 6f\n"
		"8:\n"
		"s_mov_b32 s43, 0\n"
		"s_mov_b32 s48, 1\n"
		"s_branch 6f\n"
		"9:\n"
		"s_mov_b32 s48, 2\n"
		"6:\n"
		"s_setprio 0\n"
		"s_waitcnt lgkmcnt(0)\n"
		"v_mov_b32 %[off], v29\n"
		"v_mov_b32 %[left], s43\n"
		"v_mov_b32 %[st], s48\n"
		: [off] "+v"(off), [left] "+v"(left), [st] "=v"(status)
		: [j] "s"(j_addr), [top] "v"(top0), [evlo] "s"(ev_lo), [evhi] "s"(ev_hi), [maxd] "s"(max_depth)
		: "memory", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s52", "s53", "s54",
		  "v20", "v21", "v25", "v26", "v28", "v29", "v30", "v32", "v33", "v34");
	ev_off = __builtin_amdgcn_readfirstlane(off);
	ev_left = __builtin_amdgcn_readfirstlane(left);
	return __builtin_amdgcn_readfirstlane(status);
}

// The slice through trail_walk_chain_fast; false: the branch stack did not fit, nothing was kept.
__device__ __forceinline__ bool trail_walk_slice_fast(const TrailArgs& a, uint32_t zi, uint32_t stack_base, uint32_t lds_bytes) {
	const uint64_t nb = a.nbase[zi];
	const uint32_t* starts = a.starts + nb;
	const uint32_t n_starts = a.n_starts[zi];
	uint32_t* ev = a.events + a.ibase[zi];
	const uint32_t ecap = a.icap[zi];
	uint32_t* ch_node = a.chain_node + a.kbase[zi];
	uint32_t* ch_ev0 = a.chain_ev0 + a.kbase[zi];
	const uint32_t kcap = a.kcap[zi];
	const uint32_t* vert2node = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
	const uint32_t max_depth = min((lds_bytes - stack_base) / 4u, a.walk_stack_cap);
	uint32_t off = 0, left = min(ecap, 65535u), nch = 0, err = 0;
	const unsigned long long dbg_t0 = (kTuning && a.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;
	const unsigned long long dbg_r0 = (kTuning && a.dbg) ? __builtin_amdgcn_s_memrealtime() : 0ull;
	for (uint32_t si = 0; si < n_starts && !err; si++) {
		const uint32_t sv = __builtin_amdgcn_readfirstlane(starts[si]);
		const uint32_t j = __builtin_amdgcn_readfirstlane(vert2node[sv]);
		if (nch < kcap) { ch_node[nch] = sv; ch_ev0[nch] = off >> 2; }
		else err |= TRAIL_ERR_CAPACITY;
		nch++;
		const uint32_t st = trail_walk_chain_fast(j * 12u, off, left, stack_base, max_depth, ev);
		if (st == 2u || (st == 1u && ecap > 65535u)) return false;      // (more events than a stack entry can name: the compiled walk)
		if (st) err |= TRAIL_ERR_CAPACITY;
	}
	if (kTuning && a.dbg) {
		atomicAdd(a.dbg + 8, static_cast<unsigned long long>(off >> 2));
		atomicAdd(a.dbg + 9, __builtin_amdgcn_s_memtime() - dbg_t0);
		atomicAdd(a.dbg + 10, __builtin_amdgcn_s_memrealtime() - dbg_r0);
		atomicAdd(a.dbg + 11, 1ull);
	}
	a.n_events[zi] = (off >> 2) | kEvFormatAddr12;
	a.n_chains[zi] = nch < kcap ? nch : kcap;
	if (err) atomicOr(a.slice_err + zi, err);
	return true;
}


// The same walk for slices of up to 16 383 nodes (2048 x 2048 slices of C4: 13 k) on the COMPILED walk's LDS tables:
// [four 16-bit dart ends (node << 2 | arrival edge): 8 bytes per node][remaining edges: 1 byte per node][branch
// stack: 4 bytes per entry = event index << 14 | node].  12-byte records would not fit the LDS (16 k x 12 = 192 KiB);
// here a node costs 9 bytes.  The step is the one of trail_walk_chain_fast plus a shift for the record address, a
// second address register for the edge byte and a shift for the event word (node * 4 + edge, the plain format):
// about 300 cycles against the compiled walk's 550.  s40 holds node * 8.  Needs fewer than 2^18 events.
__device__ __forceinline__ uint32_t trail_walk_chain_wide(
	uint32_t j_node, uint32_t& ev_off, uint32_t& ev_left, uint32_t adj_base, uint32_t stack_base, uint32_t max_depth, uint32_t* ev
) {
	uint32_t status, off = ev_off, left = ev_left;
	const uint32_t top0 = stack_base - 4u;        // "top entry" of the empty stack: never used
	const uint32_t ev_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev)));
	const uint32_t ev_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev) >> 32));
	const uint32_t j8 = __builtin_amdgcn_readfirstlane(j_node * 8u);
	max_depth = __builtin_amdgcn_readfirstlane(max_depth);
	adj_base = __builtin_amdgcn_readfirstlane(adj_base);
	asm volatile(
		"s_setprio 3\n"
		"s_mov_b32 s40, %[j]\n"                     // node * 8
		"s_mov_b32 s41, 0\n"                        // pend
		"s_mov_b32 s42, 0\n"                        // stack depth
		"v_readfirstlane_b32 s43, %[left]\n"
		"s_mov_b32 s52, %[evlo]\n"
		"s_mov_b32 s53, %[evhi]\n"
		"s_mov_b32 s54, %[maxd]\n"
		"s_mov_b32 s55, %[adjb]\n"
		"v_mov_b32 v29, %[off]\n"
		"v_mov_b32 v25, %[top]\n"
		"1:\n"                                      // ---- a step
		"s_lshr_b32 s51, s40, 3\n"                  // node
		"v_mov_b32 v20, s40\n"
		"s_add_u32 s50, s51, s55\n"
		"ds_read_b64 v[32:33], v20\n"               // ends 0|1, 2|3
		"v_mov_b32 v22, s50\n"
		"ds_read_u8 v34, v22\n"                     // remaining edges
		"ds_read_b32 v26, v25\n"                    // top of the branch stack
		"s_sub_u32 s43, s43, 1\n"
		"s_cbranch_scc1 8f\n"                       // no room for another event
		"s_waitcnt lgkmcnt(0)\n"
		"v_readfirstlane_b32 s46, v34\n"
		"v_readfirstlane_b32 s44, v32\n"
		"v_readfirstlane_b32 s45, v33\n"
		"s_andn2_b32 s46, s46, s41\n"               // remaining edges without the one we came by
		"s_cbranch_scc0 4f\n"
		"s_ff1_i32_b32 s47, s46\n"                  // lowest-numbered edge: right, left, down, up
		"s_bitset0_b32 s46, s47\n"
		"v_mov_b32 v21, s46\n"
		"ds_write_b8 v22, v21\n"
		"s_lshl_b32 s48, s47, 4\n"
		"s_lshr_b64 s[44:45], s[44:45], s48\n"      // the edge's end in the low 16 bits
		"s_lshr_b32 s49, s40, 1\n"
		"s_or_b32 s49, s49, s47\n"                  // node * 4 + edge
		"s_cmp_eq_u32 s46, 0\n"
		"s_cbranch_scc1 3f\n"
		"s_cmp_ge_u32 s42, s54\n"                   // ---- more edges left: the node goes on the branch stack
		"s_cbranch_scc1 9f\n"
		"s_bitset1_b32 s49, 30\n"                   // kEvBseg
		"v_add_u32 v25, 4, v25\n"
		"v_lshl_or_b32 v28, v29, 12, s51\n"         // this event's index << 14 | node
		"s_add_u32 s42, s42, 1\n"
		"ds_write_b32 v25, v28\n"
		"3:\n"
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_and_b32 s48, s44, 3\n"
		"s_lshl_b32 s41, 1, s48\n"                  // the edge consumed at the far end
		"s_and_b32 s40, s44, 0xfffc\n"
		"s_lshl_b32 s40, s40, 1\n"                  // its node * 8
		"s_branch 1b\n"
		"4:\n"                                      // ---- dead end
		"v_mov_b32 v21, 0\n"
		"ds_write_b8 v22, v21\n"
		"s_cmp_eq_u32 s42, 0\n"
		"s_cbranch_scc1 7f\n"
		"v_readfirstlane_b32 s49, v26\n"            // back to the most recent branch node
		"s_and_b32 s40, s49, 0x3fff\n"
		"s_lshl_b32 s40, s40, 3\n"
		"s_lshr_b32 s49, s49, 14\n"
		"s_bitset1_b32 s49, 31\n"                   // kEvDead | its kEvBseg
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"v_add_u32 v25, -4, v25\n"
		"s_sub_u32 s42, s42, 1\n"
		"s_mov_b32 s41, 0\n"
		"s_branch 1b\n"
		"7:\n"                                      // ---- the chain is complete
		"v_mov_b32 v30, 0xc0000000\n"               // kEvEnd
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_mov_b32 s48, 0\n"
		"s_branch 6f\n"
		"8:\n"
		"s_mov_b32 s43, 0\n"
		"s_mov_b32 s48, 1\n"
		"s_branch 6f\n"
		"9:\n"
		"s_mov_b32 s48, 2\n"
		"6:\n"
		"s_setprio 0\n"
		"s_waitcnt lgkmcnt(0)\n"
		"v_mov_b32 %[off], v29\n"
		"v_mov_b32 %[left], s43\n"
		"v_mov_b32 %[st], s48\n"
		: [off] "+v"(off), [left] "+v"(left), [st] "=v"(status)
		: [j] "s"(j8), [top] "v"(top0), [evlo] "s"(ev_lo), [evhi] "s"(ev_hi), [maxd] "s"(max_depth), [adjb] "s"(adj_base)
		: "memory", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
		  "v20", "v21", "v22", "v25", "v26", "v28", "v29", "v30", "v32", "v33", "v34");
	ev_off = __builtin_amdgcn_readfirstlane(off);
	ev_left = __builtin_amdgcn_readfirstlane(left);
	return __builtin_amdgcn_readfirstlane(status);
}

// The slice through trail_walk_chain_wide; false: the branch stack did not fit or the events outgrew 18 bits, nothing was kept.
__device__ __forceinline__ bool trail_walk_slice_wide(const TrailArgs& a, uint32_t zi, uint32_t adj_base, uint32_t stack_base, uint32_t lds_bytes) {
	const uint64_t nb = a.nbase[zi];
	const uint32_t* starts = a.starts + nb;
	const uint32_t n_starts = a.n_starts[zi];
	uint32_t* ev = a.events + a.ibase[zi];
	const uint32_t ecap = a.icap[zi];
	uint32_t* ch_node = a.chain_node + a.kbase[zi];
	uint32_t* ch_ev0 = a.chain_ev0 + a.kbase[zi];
	const uint32_t kcap = a.kcap[zi];
	const uint32_t* vert2node = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
	const uint32_t max_depth = min((lds_bytes - stack_base) / 4u, a.walk_stack_cap);
	uint32_t off = 0, left = min(ecap, 262143u), nch = 0, err = 0;
	for (uint32_t si = 0; si < n_starts && !err; si++) {
		const uint32_t sv = __builtin_amdgcn_readfirstlane(starts[si]);
		const uint32_t j = __builtin_amdgcn_readfirstlane(vert2node[sv]);
		if (nch < kcap) { ch_node[nch] = sv; ch_ev0[nch] = off >> 2; }
		else err |= TRAIL_ERR_CAPACITY;
		nch++;
		const uint32_t st = trail_walk_chain_wide(j, off, left, adj_base, stack_base, max_depth, ev);
		if (st == 2u || (st == 1u && ecap > 262143u)) return false;
		if (st) err |= TRAIL_ERR_CAPACITY;
	}
	a.n_events[zi] = (off >> 2) | kEvWalkWide;
	a.n_chains[zi] = nch < kcap ? nch : kcap;
	if (err) atomicOr(a.slice_err + zi, err);
	return true;
}

// The walk with the node tables in the wavefront's own REGISTERS: node j's record {remaining edges, ends 0|1, ends 2|3}
// lives in lane j & 63 of registers v[64 + b], v[120 + b], v[176 + b], b = j >> 6 (56 blocks: 3584 nodes), fetched with
// v_readlane under VGPR indexing (s_set_gpr_idx_on: M0 = b) and updated by a one-lane v_mov into the indexed register.
// An LDS round trip (~64 cycles of a step's ~330, tools/micro/wave_latency.hip) leaves the step; the branch stack stays
// in LDS, its top is requested at the start of every step and waited for at dead ends only.  The whole slice — tables
// from LDS into the registers, every chain from its start node — is ONE asm block: between two blocks the compiler
// would be free to use the table registers.  Ends are in the plain format (node << 2 | arrival edge), events too
// (kEvWalkWide); stack entries event index << 14 | node.
//   LDS at address 0: [records of 12 bytes, 64 * nblk of them][start node of every chain][branch stack]
// Called by the whole wavefront.  Returns 0 (all chains walked), 1 (events full), 2 (branch stack full).
constexpr uint32_t kWalkRegBlocks = 56, kWalkRegNodes = 64u * kWalkRegBlocks;
__device__ __forceinline__ uint32_t trail_walk_slice_regs_asm(
	uint32_t nblk, uint32_t list_base, uint32_t n_starts, uint32_t stack_base, uint32_t max_depth, uint32_t* ev, uint32_t* chain_ev0, uint32_t& ev_off, uint32_t& ev_left
) {
	uint32_t status, off = 0, left = ev_left;
	const uint32_t top0 = stack_base - 4u;        // "top entry" of the empty stack: never used
	const uint32_t ev_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev)));
	const uint32_t ev_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ev) >> 32));
	const uint32_t c0_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(chain_ev0)));
	const uint32_t c0_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(chain_ev0) >> 32));
	nblk = __builtin_amdgcn_readfirstlane(nblk);
	list_base = __builtin_amdgcn_readfirstlane(list_base);
	n_starts = __builtin_amdgcn_readfirstlane(n_starts);
	max_depth = __builtin_amdgcn_readfirstlane(max_depth);
	asm volatile(
		"s_mov_b32 s61, m0\n"
		"s_mov_b64 s[62:63], exec\n"
		// ---- the tables: LDS -> registers, lane l takes node 64 b + l
		"s_mov_b64 exec, -1\n"
		"v_mbcnt_lo_u32_b32 v20, -1, 0\n"
		"v_mbcnt_hi_u32_b32 v20, -1, v20\n"
		"v_mul_u32_u24 v20, 12, v20\n"
		"s_mov_b32 s50, 0\n"
		"10:\n"
		"ds_read2_b32 v[32:33], v20 offset1:1\n"
		"ds_read_b32 v34, v20 offset:8\n"
		"s_waitcnt lgkmcnt(0)\n"
		"s_set_gpr_idx_on s50, gpr_idx(DST)\n"
		"s_nop 0\n"
		"v_mov_b32 v64, v32\n"
		"v_mov_b32 v120, v33\n"
		"v_mov_b32 v176, v34\n"
		"s_set_gpr_idx_off\n"
		"v_add_u32 v20, 0x300, v20\n"
		"s_add_u32 s50, s50, 1\n"
		"s_cmp_lt_u32 s50, %[nblk]\n"
		"s_cbranch_scc1 10b\n"
		"s_mov_b64 exec, 1\n"                       // lane 0 from here on (the table registers are read with v_readlane)
		"s_setprio 3\n"
		"v_readfirstlane_b32 s43, %[left]\n"
		"s_mov_b32 s52, %[evlo]\n"
		"s_mov_b32 s53, %[evhi]\n"
		"s_mov_b32 s54, %[maxd]\n"
		"s_mov_b32 s58, %[c0lo]\n"
		"s_mov_b32 s59, %[c0hi]\n"
		"s_mov_b32 s56, %[nst]\n"
		"s_mov_b32 s57, %[list]\n"
		"s_mov_b32 s55, 0\n"                        // chain
		"v_mov_b32 v29, 0\n"                        // byte offset of the next event
		"20:\n"                                     // ---- next chain
		"s_cmp_ge_u32 s55, s56\n"
		"s_cbranch_scc1 30f\n"
		"s_lshl_b32 s48, s55, 2\n"
		"s_add_u32 s49, s48, s57\n"
		"v_mov_b32 v21, s49\n"
		"ds_read_b32 v21, v21\n"                    // its start node
		"v_lshrrev_b32 v22, 2, v29\n"
		"v_mov_b32 v24, s48\n"
		"global_store_dword v24, v22, s[58:59]\n"   // chain_ev0[chain] = index of its first event
		"s_mov_b32 s41, 0\n"                        // pend
		"s_mov_b32 s42, 0\n"                        // stack depth
		"v_mov_b32 v25, %[top]\n"
		"s_waitcnt lgkmcnt(0)\n"
		"v_readfirstlane_b32 s40, v21\n"
		"1:\n"                                      // ---- a step
		"s_lshr_b32 s50, s40, 6\n"
		"s_and_b32 s51, s40, 63\n"
		"ds_read_b32 v26, v25\n"                    // top of the branch stack: wanted at a dead end only
		"s_set_gpr_idx_on s50, gpr_idx(SRC0)\n"
		"s_nop 0\n"
		"v_readlane_b32 s46, v64, s51\n"            // remaining edges
		"v_readlane_b32 s44, v120, s51\n"           // ends 0|1
		"v_readlane_b32 s45, v176, s51\n"           // ends 2|3
		"s_set_gpr_idx_off\n"
		"s_sub_u32 s43, s43, 1\n"
		"s_cbranch_scc1 8f\n"                       // no room for another event
		"s_andn2_b32 s46, s46, s41\n"               // remaining edges without the one we came by
		"s_cbranch_scc0 4f\n"
		"s_ff1_i32_b32 s47, s46\n"                  // lowest-numbered edge: right, left, down, up
		"s_bitset0_b32 s46, s47\n"
		"s_lshl_b64 exec, 1, s51\n"                 // the node's lane
		"s_set_gpr_idx_on s50, gpr_idx(DST)\n"
		"s_nop 0\n"
		"v_mov_b32 v64, s46\n"
		"s_set_gpr_idx_off\n"
		"s_mov_b64 exec, 1\n"
		"s_lshl_b32 s48, s47, 4\n"
		"s_lshr_b64 s[44:45], s[44:45], s48\n"      // the edge's end in the low 16 bits
		"s_lshl_b32 s49, s40, 2\n"
		"s_or_b32 s49, s49, s47\n"                  // node * 4 + edge
		"s_cmp_eq_u32 s46, 0\n"
		"s_cbranch_scc1 3f\n"
		"s_cmp_ge_u32 s42, s54\n"                   // ---- more edges left: the node goes on the branch stack
		"s_cbranch_scc1 9f\n"
		"s_bitset1_b32 s49, 30\n"                   // kEvBseg
		"v_add_u32 v25, 4, v25\n"
		"v_lshl_or_b32 v28, v29, 12, s40\n"         // this event's index << 14 | node
		"s_add_u32 s42, s42, 1\n"
		"ds_write_b32 v25, v28\n"
		"3:\n"
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_and_b32 s48, s44, 3\n"
		"s_lshl_b32 s41, 1, s48\n"                  // the edge consumed at the far end
		"s_bfe_u32 s40, s44, 0xe0002\n"             // its node: bits 2 .. 15
		"s_branch 1b\n"
		"4:\n"                                      // ---- dead end
		"s_lshl_b64 exec, 1, s51\n"
		"s_set_gpr_idx_on s50, gpr_idx(DST)\n"
		"s_nop 0\n"
		"v_mov_b32 v64, 0\n"
		"s_set_gpr_idx_off\n"
		"s_mov_b64 exec, 1\n"
		"s_cmp_eq_u32 s42, 0\n"
		"s_cbranch_scc1 7f\n"
		"s_waitcnt lgkmcnt(0)\n"
		"v_readfirstlane_b32 s49, v26\n"            // back to the most recent branch node
		"s_and_b32 s40, s49, 0x3fff\n"
		"s_lshr_b32 s49, s49, 14\n"
		"s_bitset1_b32 s49, 31\n"                   // kEvDead | its kEvBseg
		"v_mov_b32 v30, s49\n"
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"v_add_u32 v25, -4, v25\n"
		"s_sub_u32 s42, s42, 1\n"
		"s_mov_b32 s41, 0\n"
		"s_branch 1b\n"
		"7:\n"                                      // ---- the chain is complete
		"v_mov_b32 v30, 0xc0000000\n"               // kEvEnd
		"global_store_dword v29, v30, s[52:53]\n"
		"v_add_u32 v29, 4, v29\n"
		"s_add_u32 s55, s55, 1\n"
		"s_branch 20b\n"
		"30:\n"
		"s_mov_b32 s48, 0\n"
		"s_branch 6f\n"
		"8:\n"
		"s_mov_b32 s43, 0\n"
		"s_mov_b32 s48, 1\n"
		"s_branch 6f\n"
		"9:\n"
		"s_mov_b32 s48, 2\n"
		"6:\n"
		"s_setprio 0\n"
		"s_waitcnt lgkmcnt(0)\n"
		"s_mov_b64 exec, s[62:63]\n"
		"s_mov_b32 m0, s61\n"
		"v_readfirstlane_b32 s49, v29\n"
		"v_mov_b32 %[off], s49\n"
		"v_mov_b32 %[left], s43\n"
		"v_mov_b32 %[st], s48\n"
		: [off] "+v"(off), [left] "+v"(left), [st] "=v"(status)
		: [nblk] "s"(nblk), [list] "s"(list_base), [nst] "s"(n_starts), [top] "v"(top0), [evlo] "s"(ev_lo), [evhi] "s"(ev_hi), [c0lo] "s"(c0_lo), [c0hi] "s"(c0_hi), [maxd] "s"(max_depth)
		: "memory", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s61", "s62", "s63",
		  "v20", "v21", "v22", "v24", "v25", "v26", "v28", "v29", "v30", "v32", "v33", "v34",
		  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
	ev_off = __builtin_amdgcn_readfirstlane(off);
	ev_left = __builtin_amdgcn_readfirstlane(left);
	return __builtin_amdgcn_readfirstlane(status);
}

// grid = nslices, block = one wavefront; dynamic LDS = lds_bytes
static __global__ void __launch_bounds__(kWave) k_trail_walk(TrailArgs a, uint32_t lds_bytes) {
	extern __shared__ uint32_t s_trail[];
	const uint32_t zi = blockIdx.x + a.z0;
	const uint32_t nn = min(a.n_nodes[zi], a.ncap[zi]);
	const uint64_t nb = a.nbase[zi];
	if (a.slice_err[zi]) {
		if (threadIdx.x == 0) { a.n_events[zi] = 0; a.n_chains[zi] = 0; }
		return;
	}
	const unsigned long long dbg_k0 = (kTuning && a.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;
	// hand-scheduled walk with the node tables in registers (slices of up to 3584 nodes)
	if (!a.walk_plain && !a.walk_no_regs && nn <= kWalkRegNodes && __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s_trail))) == 0u) {
		const uint32_t nblk = (nn + 63u) / 64u;
		const uint32_t recs = nblk * 64u * 12u;
		const uint32_t n_starts = a.n_starts[zi];
		const uint32_t list_bytes = ((n_starts * 4u + 15u) / 16u) * 16u;
		const uint32_t kcap = a.kcap[zi];
		const uint32_t ecap = a.icap[zi];
		if (nblk > 0u && n_starts <= kcap && recs + list_bytes + 2048u <= lds_bytes) {
			for (uint32_t j = threadIdx.x; j < nblk * 64u; j += kWave) {
				uint32_t e[4] = { 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu };
				if (j < nn) {
#pragma unroll
					for (uint32_t k = 0; k < 4; k++) {
						const uint32_t d = a.dart_end[(nb + j) * 4u + k];
						e[k] = d == kDartNone ? 0xFFFFu : d;
					}
				}
				s_trail[j * 3u] = j < nn ? a.node_adj[nb + j] : 0u;
				s_trail[j * 3u + 1u] = e[0] | (e[1] << 16);
				s_trail[j * 3u + 2u] = e[2] | (e[3] << 16);
			}
			const uint32_t* starts = a.starts + nb;
			const uint32_t* vert2node = a.vert2node + static_cast<uint64_t>(zi) * a.nverts;
			uint32_t* ch_node = a.chain_node + a.kbase[zi];
			for (uint32_t si = threadIdx.x; si < n_starts; si += kWave) {
				const uint32_t sv = starts[si];
				s_trail[recs / 4u + si] = vert2node[sv];
				ch_node[si] = sv;
			}
			__syncthreads();
			const uint32_t stack_base = recs + list_bytes;
			const uint32_t max_depth = min((lds_bytes - stack_base) / 4u, a.walk_stack_cap);
			uint32_t off = 0, left = min(ecap, 262143u);
			const uint32_t st = trail_walk_slice_regs_asm(nblk, recs, n_starts, stack_base, max_depth, a.events + a.ibase[zi], a.chain_ev0 + a.kbase[zi], off, left);
			if (st == 0u || (st == 1u && ecap <= 262143u)) {
				if (threadIdx.x == 0) {
					a.n_events[zi] = (off >> 2) | kEvWalkWide;
					a.n_chains[zi] = n_starts;
					if (st) atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY);
				}
				return;
			}
			__syncthreads();      // (branch stack beyond its LDS part, or more events than a stack entry can name: the walks below)
		}
	}
	// hand-scheduled walk: [node records of 12 bytes][branch stack: 4 bytes per entry]
	const uint32_t rec_bytes = nn * 12u + 16u;
	if (!a.walk_plain && nn < 5461u && rec_bytes + 2048u <= lds_bytes && __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s_trail))) == 0u) {
		for (uint32_t j = threadIdx.x; j < nn; j += kWave) {
			uint32_t e[4];
#pragma unroll
			for (uint32_t k = 0; k < 4; k++) {
				const uint32_t d = a.dart_end[(nb + j) * 4u + k];
				e[k] = d == kDartNone ? 0xFFFFu : (((d >> 2) * 12u) | (d & 3u));
			}
			s_trail[j * 3u] = a.node_adj[nb + j];
			s_trail[j * 3u + 1u] = e[0] | (e[1] << 16);
			s_trail[j * 3u + 2u] = e[2] | (e[3] << 16);
		}
		__syncthreads();
		if (kTuning && a.dbg && threadIdx.x == 0) atomicAdd(a.dbg + 6, __builtin_amdgcn_s_memtime() - dbg_k0);
		bool done = false;
		if (threadIdx.x == 0) done = trail_walk_slice_fast(a, zi, rec_bytes, lds_bytes);
		if (kTuning && a.dbg && threadIdx.x == 0) atomicAdd(a.dbg + 7, __builtin_amdgcn_s_memtime() - dbg_k0);
		if (__builtin_amdgcn_readfirstlane(done ? 1u : 0u)) return;
		__syncthreads();
	}
	// LDS: [dart ends u16 x 4 nn][adj u8 x nn][branch stack: (node, event) pairs]
	const uint32_t tab_bytes = ((nn * 9u + 15u) / 16u) * 16u;
	const bool in_lds = nn < 16384u && tab_bytes + 2048u <= lds_bytes;
	if (in_lds) {
		TrailTabLds t;
		t.end4 = reinterpret_cast<unsigned long long*>(s_trail);
		t.adj = reinterpret_cast<uint8_t*>(s_trail) + nn * 8u;
		uint16_t* e16 = reinterpret_cast<uint16_t*>(s_trail);
		auto fill = [&]() {
			for (uint32_t d = threadIdx.x; d < nn * 4u; d += kWave) {
				const uint32_t e = a.dart_end[nb * 4u + d];
				e16[d] = static_cast<uint16_t>(e == kDartNone ? 0xFFFFu : e);
			}
			for (uint32_t j = threadIdx.x; j < nn; j += kWave) t.adj[j] = a.node_adj[nb + j];
			__syncthreads();
		};
		fill();
		// the hand-scheduled loop on these tables (slices too large for the 12-byte records); it starts over with the
		// compiled walk below when its branch stack (4 bytes per entry behind the tables) or its event index runs out
		if (!a.walk_plain && tab_bytes + 2048u <= lds_bytes && __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s_trail))) == 0u) {
			bool done = false;
			if (threadIdx.x == 0) done = trail_walk_slice_wide(a, zi, nn * 8u, tab_bytes, lds_bytes);
			if (__builtin_amdgcn_readfirstlane(done ? 1u : 0u)) return;
			__syncthreads();
			fill();      // the walk consumed the edge bytes
		}
		if (threadIdx.x == 0) trail_walk_slice<TrailTabLds>(a, zi, t, reinterpret_cast<uint2*>(s_trail + tab_bytes / 4u), (lds_bytes - tab_bytes) / 8u);
	}
	else if (threadIdx.x == 0) {
		TrailTabGlobal t;
		t.adj = a.node_adj + nb;
		t.end = a.dart_end + nb * 4u;
		trail_walk_slice<TrailTabGlobal>(a, zi, t, reinterpret_cast<uint2*>(s_trail), lds_bytes / 8u);
	}
}

// ---- events -> items ------------------------------------------------------------------
// What the reference's walk decides step by step as it emits, for all steps of a slice at once (one
// workgroup per slice).
// Per event i of a chain (a chain's events end with its kEvEnd):
//   rib(i)      kEvDead that remove_initial_branch swallows (crackcodes.hpp:185-242): the chain began
//               with a kEvBseg and nothing but plain segments lies between the two.  The leading 'b'
//               and this 't' vanish, the first stretch is emitted backwards (its darts reversed), the
//               chain starts where that stretch ended
//   dead(i)     kEvDead / kEvEnd that is not rib(i): a 't'
//   items       kEvSeg 1, kEvBseg 2 ('b' + segment), dead(i) 1 -- unless dead(i - 1): then the 't' and
//               the 'b' that event i - 1 had popped vanish instead (remove_spurious_branches,
//               crackcodes.hpp:250-281; the chain's end counts as a 't', crackcodes.hpp:436-439)
//   last(i)     direction of the code in front of event i's items, which picks the form of 'b' and 't'
//               (crackcodes.hpp:155-174): arrival direction of a segment; after a run of dead events
//               only its first one emitted, so the direction in front of the run decides
// Scratch: lnp[i] = latest event <= i that is not a plain segment, lnd[i] = latest event <= i that is
// not dead, xi[i] = first item of event i | flags.
constexpr uint32_t kXiRib = 1u << 31, kXiDead = 1u << 30, kXiPrevDead = 1u << 29, kXiMask = (1u << 29) - 1u;

struct TrailEvents {
	const uint32_t* ev;
	const int32_t* lnp;
	uint32_t n;
	__device__ __forceinline__ bool first(uint32_t i) const { return i == 0 || (ev[i - 1] & kEvMask) == kEvEnd; }
	// chain-first kEvBseg that event i (a kEvDead) closes as an initial branch, or -1
	__device__ __forceinline__ int32_t rib_of(uint32_t i) const {
		if ((ev[i] & kEvMask) != kEvDead || i == 0) return -1;
		const int32_t q = lnp[i - 1];
		if (q < 0 || (ev[q] & kEvMask) != kEvBseg || !first(static_cast<uint32_t>(q))) return -1;
		return q;
	}
	__device__ __forceinline__ bool dead(uint32_t i) const {
		const uint32_t k = ev[i] & kEvMask;
		return k == kEvEnd || (k == kEvDead && rib_of(i) < 0);
	}
};

constexpr int kItemsBlock = 1024;      // the passes are chains of dependent loads: resident wavefronts are what hides them

// grid = nslices
static __global__ void __launch_bounds__(kItemsBlock) k_trail_items(TrailArgs a) {
	constexpr int kNW = kItemsBlock / kWave;
	__shared__ uint32_t s_scan[kNW];
	__shared__ int32_t s_scan_max[kNW];
	const uint32_t zi = blockIdx.x + a.z0;
	if (a.slice_err[zi]) {
		if (threadIdx.x == 0) a.n_items[zi] = 0;
		return;
	}
	const uint32_t n = a.n_events[zi] & ~(kEvFormatAddr12 | kEvWalkWide);
	const bool addr12 = (a.n_events[zi] & kEvFormatAddr12) != 0;
	// the dart (4 * node + edge) of a segment event
	auto dart = [&](uint32_t e) -> uint32_t {
		const uint32_t p = e & ~kEvMask;
		return addr12 ? (((((p >> 2) * 0xAAABu) >> 17) << 2) | (p & 3u)) : p;      // p = 12 * node | edge
	};
	const uint64_t ib = a.ibase[zi], nb = a.nbase[zi];
	const uint32_t* ev = a.events + ib;
	int32_t* lnp = reinterpret_cast<int32_t*>(a.item_off + ib);      // free until k_trail_offsets
	int32_t* lnd = reinterpret_cast<int32_t*>(a.ev_lnd + ib);
	uint32_t* xi = a.ev_item + ib;
	uint32_t* items = a.items + ib;
	const uint32_t icap = a.icap[zi];
	const uint32_t* dend = a.dart_end + nb * 4u;
	constexpr uint32_t kPer = 4;
	constexpr uint32_t kNone = 4u, kLeft = 1u, kUp = 3u;
	constexpr uint32_t kB = kItemCtl | TCODE_UP | (TCODE_DOWN << 2), kBalt = kItemCtl | TCODE_LEFT | (TCODE_RIGHT << 2);
	constexpr uint32_t kT = kItemCtl | TCODE_DOWN | (TCODE_UP << 2), kTalt = kItemCtl | TCODE_RIGHT | (TCODE_LEFT << 2);

	// pass 1: lnp
	{
		int32_t carry = -1;
		for (uint32_t i0 = 0; i0 < n; i0 += kItemsBlock * kPer) {
			int32_t v[kPer], run = -1;
#pragma unroll
			for (uint32_t q = 0; q < kPer; q++) {
				const uint32_t i = i0 + threadIdx.x * kPer + q;
				if (i < n && (ev[i] & kEvMask) != kEvSeg) run = static_cast<int32_t>(i);
				v[q] = run;
			}
			int32_t total;
			const int32_t excl = block_excl_max<kNW>(run, total, s_scan_max);
			const int32_t pre = carry > excl ? carry : excl;
#pragma unroll
			for (uint32_t q = 0; q < kPer; q++) {
				const uint32_t i = i0 + threadIdx.x * kPer + q;
				if (i < n) lnp[i] = v[q] > pre ? v[q] : pre;
			}
			carry = carry > total ? carry : total;
		}
	}
	__syncthreads();
	__threadfence_block();
	TrailEvents te = { ev, lnp, n };

	// pass 2: flags, first item of every event, lnd
	uint32_t total_items = 0;
	{
		int32_t carry_max = -1;
		for (uint32_t i0 = 0; i0 < n; i0 += kItemsBlock * kPer) {
			uint32_t cnt[kPer], fl[kPer], sum = 0;
			int32_t v[kPer], run = -1;
#pragma unroll
			for (uint32_t q = 0; q < kPer; q++) {
				const uint32_t i = i0 + threadIdx.x * kPer + q;
				cnt[q] = 0; fl[q] = 0;
				if (i < n) {
					const uint32_t k = ev[i] & kEvMask;
					if (k == kEvSeg) cnt[q] = 1;
					else if (k == kEvBseg) cnt[q] = 2;
					else {
						const bool rib = k == kEvDead && te.rib_of(i) >= 0;
						const bool prev_dead = !te.first(i) && te.dead(i - 1);
						if (rib) fl[q] = kXiRib;
						else { fl[q] = kXiDead | (prev_dead ? kXiPrevDead : 0u); cnt[q] = prev_dead ? 0u : 1u; }
					}
					if (!(fl[q] & kXiDead)) run = static_cast<int32_t>(i);
				}
				v[q] = run;
				sum += cnt[q];
			}
			uint32_t sv[1] = { sum }, tot[1];
			block_excl_add<1, kNW>(sv, tot, s_scan);
			int32_t total_max;
			const int32_t excl = block_excl_max<kNW>(run, total_max, s_scan_max);
			const int32_t pre = carry_max > excl ? carry_max : excl;
			uint32_t x = total_items + sv[0];
#pragma unroll
			for (uint32_t q = 0; q < kPer; q++) {
				const uint32_t i = i0 + threadIdx.x * kPer + q;
				if (i < n) {
					xi[i] = x | fl[q];
					lnd[i] = v[q] > pre ? v[q] : pre;
				}
				x += cnt[q];
			}
			total_items += tot[0];
			carry_max = carry_max > total_max ? carry_max : total_max;
		}
	}
	__syncthreads();
	__threadfence_block();
	if (total_items > icap || total_items > kXiMask) {
		if (threadIdx.x == 0) { atomicOr(a.slice_err + zi, TRAIL_ERR_CAPACITY); a.n_items[zi] = 0; }
		return;
	}

	// direction of the code in front of event i when event i - 1 is not dead (4: none)
	auto last_simple = [&](uint32_t i) -> uint32_t {
		if (te.first(i)) return kNone;
		const uint32_t p = ev[i - 1];
		if ((p & kEvMask) == kEvDead) {      // rib: the reversed stretch ends with the first segment walked backwards
			const int32_t q = lnp[i - 2];
			return (ev[q] & 3u) ^ 1u;
		}
		return (dend[dart(p)] & 3u) ^ 1u;
	};
	auto last_before = [&](uint32_t i) -> uint32_t {
		if (te.first(i) || !(xi[i - 1] & kXiDead)) return last_simple(i);
		// a run of dead events in front: its first one emitted the 't'
		const uint32_t r = static_cast<uint32_t>(lnd[i - 1] + 1);
		const uint32_t lb = last_simple(r);
		return ((0x18u >> lb) & 1u) ? kLeft : kUp;
	};

	// pass 3: the items
	for (uint32_t i = threadIdx.x; i < n; i += kItemsBlock) {
		const uint32_t e = ev[i], k = e & kEvMask, xf = xi[i], x = xf & kXiMask;
		if (k == kEvSeg) items[x] = kItemSeg | dart(e);
		else if (k == kEvBseg) {
			const uint32_t lb = last_before(i);
			items[x] = ((0x14u >> lb) & 1u) ? kBalt : kB;        // no code in front, or DOWN
			items[x + 1] = kItemSeg | dart(e);
		}
		else if ((xf & (kXiDead | kXiPrevDead)) == kXiDead) {
			const uint32_t lb = last_simple(i);
			items[x] = ((0x18u >> lb) & 1u) ? kTalt : kT;        // no code in front, or UP
		}
	}
	__syncthreads();
	__threadfence_block();

	// pass 4: what vanishes, and the initial branches
	for (uint32_t i = threadIdx.x; i < n; i += kItemsBlock) {
		const uint32_t xf = xi[i];
		if ((xf & (kXiDead | kXiPrevDead)) == (kXiDead | kXiPrevDead)) {
			const uint32_t b = ev[i - 1] & ~kEvMask;        // event i - 1 is a kEvDead: the 'b' it popped
			items[xi[b] & kXiMask] = kItemDead;
		}
		if (xf & kXiRib) {
			const uint32_t q = static_cast<uint32_t>(lnp[i - 1]);
			const uint32_t x0 = xi[q] & kXiMask;
			items[x0] = kItemDead;
			// chain of q: the start moves to where the stretch ended
			const uint32_t* ev0 = a.chain_ev0 + a.kbase[zi];
			uint32_t lo = 0, hi = a.n_chains[zi];
			while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (ev0[mid] <= q) lo = mid; else hi = mid; }
			a.chain_node[a.kbase[zi] + lo] = a.node_vertex[nb + (dend[dart(ev[i - 1])] >> 2)];
			// (the stretch is one segment, or two when the first one came back to the start node: a node
			// reached for the first time either has more than one edge left, which ends the stretch, or none)
			for (uint32_t m = q; m < i; m++) items[x0 + 1u + (i - 1u - m)] = kItemSeg | dend[dart(ev[m])];
		}
	}
	__syncthreads();
	// chains
	{
		const uint32_t nch = a.n_chains[zi];
		const uint32_t* ev0 = a.chain_ev0 + a.kbase[zi];
		uint32_t* item0 = a.chain_item0 + a.kbase[zi];
		for (uint32_t c = threadIdx.x; c < nch; c += kItemsBlock) item0[c] = ev0[c] < n ? (xi[ev0[c]] & kXiMask) : total_items;
	}
	if (threadIdx.x == 0) a.n_items[zi] = total_items;
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

// grid = (ceil(max items / (kExpandChunk * 4)), nslices): every item writes its code points.  A segment
// whose code points k_trail_segments kept beside its dart (nearly all: kInlineCodes of them) is a copy;
// the others are re-walked, lanes refilling from the wavefront's list like k_trail_segments
static __global__ void __launch_bounds__(kBlock) k_trail_expand(TrailArgs a) {
	__shared__ uint32_t s_long[kWaves][kExpandChunk];      // items to walk
	const uint32_t zi = blockIdx.y + a.z0;
	if (a.slice_err[zi]) return;
	const uint32_t lane = threadIdx.x & (kWave - 1);
	const uint32_t wv = threadIdx.x >> 6;
	const uint32_t wave = blockIdx.x * kWaves + wv;
	const uint32_t n_items = a.n_items[zi];
	const uint32_t range_begin = wave * kExpandChunk;
	const uint32_t range_end = min(range_begin + kExpandChunk, n_items);
	if (range_begin >= range_end) return;
	const uint64_t nb = a.nbase[zi];
	const uint32_t* items = a.items + a.ibase[zi];
	const uint32_t* item_off = a.item_off + a.ibase[zi];
	uint8_t* cp0 = a.cp + a.cbase[zi];
	const uint4* adjm = a.adjm + zi * a.adjm_stride;
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	uint32_t* longs = s_long[wv];
	struct __attribute__((packed)) U64 { unsigned long long v; };
	struct __attribute__((packed)) U32 { uint32_t v; };
	struct __attribute__((packed)) U16 { uint16_t v; };
	// the low n (< 8) bytes of v: at most three stores (4 + 2 + 1 bytes) instead of one per byte
	auto store_tail = [](uint8_t* o, unsigned long long v, uint32_t n) {
		if (n & 4u) { reinterpret_cast<U32*>(o)->v = static_cast<uint32_t>(v); o += 4; v >>= 32; }
		if (n & 2u) { reinterpret_cast<U16*>(o)->v = static_cast<uint16_t>(v); o += 2; v >>= 16; }
		if (n & 1u) o[0] = static_cast<uint8_t>(v);
	};

	// ---- the copies
	uint32_t n_long = 0;
	for (uint32_t i0 = range_begin; i0 < range_end; i0 += kWave) {
		const uint32_t i = i0 + lane;
		bool walk = false;
		if (i < range_end) {
			const uint32_t it = items[i], kind = it & kItemMask;
			uint8_t* o = cp0 + item_off[i];
			if (kind == kItemCtl) reinterpret_cast<U16*>(o)->v = static_cast<uint16_t>((it & 3u) | (((it >> 2) & 3u) << 8));
			else if (kind == kItemSeg) {
				const uint32_t d = it & ~kItemMask;
				const uint32_t len = a.dart_len[nb * 4u + d];
				if (len && a.dart_inline[nb * 4u + d] && len <= kInlineCodes) {
					const uint4 q = a.dart_codes[nb * 4u + d];
					const uint32_t w[4] = { q.x, q.y, q.z, q.w };
					// eight code points (16 bits) -> eight bytes per store
#pragma unroll
					for (uint32_t g = 0; g < kInlineCodes / 8u; g++) {
						if (g * 8u < len) {
							const uint32_t h = (w[g >> 1] >> ((g & 1u) * 16u)) & 0xFFFFu;
							uint32_t lo = h & 0xFFu, hi = h >> 8;
							lo = (lo | (lo << 12)) & 0x000F000Fu; lo = (lo | (lo << 6)) & 0x03030303u;
							hi = (hi | (hi << 12)) & 0x000F000Fu; hi = (hi | (hi << 6)) & 0x03030303u;
							const unsigned long long v = (static_cast<unsigned long long>(hi) << 32) | lo;
							if (g * 8u + 8u <= len) reinterpret_cast<U64*>(o + g * 8u)->v = v;
							else store_tail(o + g * 8u, v, len - g * 8u);
						}
					}
				}
				else walk = len != 0;
			}
		}
		const unsigned long long m = __ballot(walk);
		if (walk) longs[n_long + static_cast<uint32_t>(__popcll(m & lt_mask))] = i;
		n_long += static_cast<uint32_t>(__popcll(m));
	}
	if (n_long == 0) return;

	// ---- the walks
	MicroTile c;
	c.lo = c.hi = make_uint4(0, 0, 0, 0); c.mx = c.my = 0xFFFFFFFFu;
	bool active = false, need_nib = false;
	// at vertex (x, y); k: direction of the next move, or (need_nib) of the move that led here
	uint32_t x = 0, y = 0, k = 0, left = 0, nacc = 0, next = 0;
	unsigned long long acc = 0;
	uint8_t* cp = cp0;
	for (;;) {
		const unsigned long long need = __ballot(!active);
		const bool refill = next < n_long && (__popcll(need) >= kWalkRefill || need == ~0ull);
		const uint32_t cand = next + static_cast<uint32_t>(__popcll(need & lt_mask));
		const bool take = refill && !active && cand < n_long;
		uint32_t r_it = kItemDead, r_off = 0;
		if (take) { const uint32_t i = longs[cand]; r_it = items[i]; r_off = item_off[i]; }
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
					if (nacc == 8u) reinterpret_cast<U64*>(cp)->v = acc;
					else store_tail(cp, acc, nacc);
					cp += nacc; acc = 0; nacc = 0;
				}
				if (left == 0) active = false;
				else need_nib = true;
			}
		}
		if (take) {
			const uint32_t d = r_it & ~kItemMask;
			left = a.dart_len[nb * 4u + d];
			const uint32_t v0 = a.node_vertex[nb + (d >> 2)];
			y = v0 / a.sxe; x = v0 - y * a.sxe;
			k = d & 3u;
			cp = cp0 + r_off;
			need_nib = false;
			active = true;
		}
		if (refill) next += static_cast<uint32_t>(__popcll(need));
		if (!__ballot(active) && next >= n_long) break;
	}
}

}  // namespace dev
}  // namespace ckl
