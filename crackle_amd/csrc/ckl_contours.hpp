// Point clouds: the contour tracer of crackle::dual_graph (src/dual_graph.hpp:133-275) behind
// operations::point_cloud (src/operations.hpp:183-262), on the crack planes the decoder leaves.
//
// The reference traces one slice on one core: a raster scan finds the next start pixel
// (VCGGraph::next_contour), a wall follower walks the boundary loop through it and sets a
// visited bit on every pixel it passes; what a later start sees depends on those bits, so the
// order of discovery is part of the result.  Slices are independent, the walk inside a slice is
// not: here a slice is one wavefront.  The scan for the next start is 64 pixels wide (ballot);
// the walk itself is wave-uniform code — the compiler keeps it on the scalar unit, one
// scalar-cache load of the pixel's direction mask per step — with the visited bits of the
// whole slice in LDS (one bit per pixel; slices above ~1.2 M pixels keep them in HBM).
//
//   k_contour_dirs     passable directions of every pixel from the crack planes, border cleared
//                      (dual_graph.hpp:139-150), as a mask in rotational order R, D, L, U
//   k_trace_contours   the scan + walk of extract_contours_helper; per kept contour: where its
//                      nodes lie, how many, where the smallest node sits (the rotation of
//                      dual_graph.hpp:203-211) and that node
//   k_contour_components   component of every contour (merge_contours_via_vcg_coloring looks up
//                      cc_labels[contour[0]], dual_graph.hpp:229) through the run tables
//   k_contour_emit     (x, y, z) uint16 triples of the contours in the order the host planned
//                      (operations.hpp:229-257)
#pragma once
#include "ckl_device.hpp"
#include "ckl_runs.hpp"

namespace ckl {
namespace dev {

constexpr uint32_t kDirR = 1u, kDirD = 2u, kDirL = 4u, kDirU = 8u;

// grid = (ceil(sxy / 256), nslices).  Also the two candidate bitmaps of the start scan (one bit
// per pixel, cand_words per slice, zeroed beforehand): A = a wall on the left or right side,
// B = a wall on the right side and a pixel beyond it.
__global__ void __launch_bounds__(256) k_contour_dirs(RunGeom g, uint64_t sxy, uint64_t stride, uint8_t* __restrict__ out, uint32_t cand_words, uint32_t* __restrict__ cand_a, uint32_t* __restrict__ cand_b) {
	const uint32_t zi = blockIdx.y;
	const uint64_t p = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	uint32_t m = kDirR | kDirL, x = 0;
	if (p < sxy) {
		const uint32_t y = static_cast<uint32_t>(p / g.sx);
		x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
		const uint32_t* pv = g.planeV + zi * g.plane_words;
		const uint32_t* ph = g.planeH + zi * g.plane_words;
		auto joined = [&](const uint32_t* plane, uint32_t px, uint32_t py) -> uint32_t {
			const uint32_t bit = (plane[static_cast<uint64_t>(py) * g.row_words + (px >> 5)] >> (px & 31u)) & 1u;
			return g.flip ? (bit ^ 1u) : bit;
		};
		m = 0;
		if (x + 1 < g.sx && joined(pv, x + 1, y)) m |= kDirR;
		if (y + 1 < g.sy && joined(ph, x, y + 1)) m |= kDirD;
		if (x >= 1 && joined(pv, x, y)) m |= kDirL;
		if (y >= 1 && joined(ph, x, y)) m |= kDirU;
		out[static_cast<uint64_t>(zi) * stride + p] = static_cast<uint8_t>(m);
	}
	const unsigned long long ba = __ballot(p < sxy && (m & (kDirR | kDirL)) != (kDirR | kDirL));
	const unsigned long long bb = __ballot(p < sxy && (m & kDirR) == 0u && x + 1 < g.sx);
	if ((threadIdx.x & 63u) == 0 && p < sxy) {
		const uint64_t w = static_cast<uint64_t>(zi) * cand_words + (p >> 5);
		cand_a[w] = static_cast<uint32_t>(ba); cand_a[w + 1] = static_cast<uint32_t>(ba >> 32);
		cand_b[w] = static_cast<uint32_t>(bb); cand_b[w + 1] = static_cast<uint32_t>(bb >> 32);
	}
}

struct ContourArgs {
	const uint8_t* dirs;       // [nslices][dirs_stride], dirs_stride a multiple of 4 (the walk reads aligned words)
	const uint32_t* cand_a;    // [nslices][cand_words] start candidates (k_contour_dirs); cand_words covers a scan window past the end
	const uint32_t* cand_b;
	uint32_t* visited;         // [nslices][vis_words] (only when the bits do not fit the LDS; zeroed)
	uint32_t* walked_r;        // [nslices][vis_words] zeroed: the wall at this pixel's right side lies on a loop already walked (null: no memo)
	uint32_t* raw;             // [nslices][raw_cap] contour nodes as walked
	uint4* table;              // [nslices][tab_cap]: offset into raw, length, position of the smallest node, that node
	uint32_t* counts;          // [nslices][4]: contours, nodes, flags (1 raw overflow, 2 table overflow, 4 walk did not close), steps
	uint32_t sx, sy;
	uint32_t sxy;
	uint32_t raw_cap, tab_cap, vis_words, cand_words;
	uint64_t dirs_stride;
};

constexpr uint32_t kContourRawOverflow = 1u, kContourTableOverflow = 2u, kContourOpenWalk = 4u;
constexpr uint32_t kContourWindow = 64u;      // words of the candidate bitmaps one scan step looks at (2048 pixels)

// compute_next_move (dual_graph.hpp:66-131) on rotational direction indices (0 R, 1 D, 2 L, 3 U):
// first the turn towards the followed wall (clockwise: +1), then straight on, the other turn, back.
// wall_r: the pixel's right side was tried and found closed, i.e. that wall lies on the loop walked.
__device__ __forceinline__ uint32_t contour_next_move(uint32_t turn, uint32_t last, uint32_t allowed, bool& wall_r) {
	const uint32_t a = (last + turn) & 3u, c = (last - turn) & 3u, d = (last + 2u) & 3u;
	wall_r = false;
	if ((allowed >> a) & 1u) return a;
	wall_r = a == 0u;
	if ((allowed >> last) & 1u) return last;
	wall_r = wall_r || last == 0u;
	if ((allowed >> c) & 1u) return c;
	wall_r = wall_r || c == 0u;
	if ((allowed >> d) & 1u) return d;
	return 4u;
}

// The same as a table per sense of rotation, for the walk's inner loop: entry (last << 4 | allowed)
// holds the next move (2 bits) resp. whether the right side was tried and found closed (1 bit).
// A pixel reached by a move always allows the move back, so `allowed` is never 0 there.
struct ContourMoveTable { unsigned long long next_lo, next_hi, wall_r; };
constexpr ContourMoveTable contour_move_table(uint32_t turn) {
	ContourMoveTable t = { 0ull, 0ull, 0ull };
	for (uint32_t last = 0; last < 4; last++) {
		for (uint32_t allowed = 0; allowed < 16; allowed++) {
			const uint32_t order[4] = { (last + turn) & 3u, last, (last - turn) & 3u, (last + 2u) & 3u };
			uint32_t next = last, wall = 0;
			for (uint32_t k = 0; k < 4; k++) {
				if ((allowed >> order[k]) & 1u) { next = order[k]; break; }
				if (order[k] == 0u) wall = 1;
			}
			const uint32_t e = last * 16u + allowed;
			if (e < 32u) t.next_lo |= static_cast<unsigned long long>(next) << (2u * e);
			else t.next_hi |= static_cast<unsigned long long>(next) << (2u * (e - 32u));
			t.wall_r |= static_cast<unsigned long long>(wall) << e;
		}
	}
	return t;
}
constexpr ContourMoveTable kContourClockwise = contour_move_table(1u), kContourCounter = contour_move_table(3u);

// the direction masks are written by an earlier launch: read through the constant address space,
// a wave-uniform index becomes a scalar load (its own counter, no wait behind the node stores)
typedef const __attribute__((address_space(4))) uint32_t* contour_const_words;

// One wavefront per slice.  grid = nslices, block = 64, dynamic LDS = vis_words * 4 when LDSVIS.
template <bool LDSVIS>
__global__ void __launch_bounds__(64) k_trace_contours(ContourArgs a) {
	extern __shared__ uint32_t s_vis[];
	const uint32_t zi = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const contour_const_words dirs4 = (contour_const_words)(reinterpret_cast<uintptr_t>(a.dirs + static_cast<uint64_t>(zi) * a.dirs_stride));
	auto dir_of = [&](uint32_t v) -> uint32_t { return (dirs4[v >> 2] >> ((v & 3u) * 8u)) & 15u; };
	const uint32_t* __restrict__ cand_a = a.cand_a + static_cast<uint64_t>(zi) * a.cand_words;
	const uint32_t* __restrict__ cand_b = a.cand_b + static_cast<uint64_t>(zi) * a.cand_words;
	uint32_t* vis = LDSVIS ? s_vis : a.visited + static_cast<uint64_t>(zi) * a.vis_words;
	uint32_t* walked_r = a.walked_r ? a.walked_r + static_cast<uint64_t>(zi) * a.vis_words : nullptr;
	uint32_t* __restrict__ raw = a.raw + static_cast<uint64_t>(zi) * a.raw_cap;
	uint4* __restrict__ table = a.table + static_cast<uint64_t>(zi) * a.tab_cap;
	if (LDSVIS) {
		for (uint32_t i = lane; i < a.vis_words; i += 64u) s_vis[i] = 0u;
		__syncthreads();
	}
	const uint32_t sxy = a.sxy, sx = a.sx, vis_words = a.vis_words;
	// step of a move in the row-major pixel index: R +1, D +sx, L -1, U -sx
	auto delta = [&](uint32_t move) -> uint32_t { const uint32_t d = (move & 1u) ? sx : 1u; return (move & 2u) ? 0u - d : d; };
	uint32_t n_contours = 0, tail = 0, flags = 0, total_steps = 0;

	uint32_t pos = 0;
	uint32_t win = 0xFFFFFFFFu, a_w = 0, b_w = 0;      // this lane's words of the current candidate window
	while (pos < sxy && !flags) {
		// ---- VCGGraph::next_contour (dual_graph.hpp:40-61): the first pixel at or after pos that is
		// unvisited with a wall at its left or right, or whose right neighbour is unvisited behind a
		// wall — 64 words of the candidate bitmaps against the visited bits per step ----
		const uint32_t wbase = (pos >> 5) & ~(kContourWindow - 1u);
		if (wbase != win) { win = wbase; a_w = cand_a[wbase + lane]; b_w = cand_b[wbase + lane]; }
		const uint32_t w = wbase + lane;
		const uint32_t v0 = w < vis_words ? vis[w] : 0xFFFFFFFFu;
		const uint32_t v1 = w + 1u < vis_words ? vis[w + 1u] : 0xFFFFFFFFu;
		uint32_t fire = (a_w & ~v0) | (b_w & ~((v0 >> 1) | (v1 << 31)));
		const uint32_t pw = pos >> 5;
		if (w < pw) fire = 0u;
		else if (w == pw) fire &= 0xFFFFFFFFu << (pos & 31u);
		const unsigned long long found = __ballot(fire != 0u);
		if (!found) { pos = (wbase + kContourWindow) << 5; continue; }
		const uint32_t fl = static_cast<uint32_t>(__ffsll(static_cast<long long>(found))) - 1u;
		const uint32_t fw = __builtin_amdgcn_readlane(fire, fl);
		const uint32_t start = ((wbase + fl) << 5) + static_cast<uint32_t>(__ffs(static_cast<int>(fw))) - 1u;

		// ---- the walk (dual_graph.hpp:161-199): wave-uniform ----
		uint32_t node = start;
		const uint32_t m_start = dir_of(node);
		const bool start_visited = (vis[node >> 5] >> (node & 31u)) & 1u;
		if (start_visited && walked_r) {
			// A visited start fires only for its unvisited right neighbour behind a wall; the walk then
			// follows the loop that wall lies on, clockwise, and is dropped unless it meets a pixel no
			// walk has passed.  A walk tries a closed side only on the loop it follows, a walk always
			// closes, and the walk in the other sense is the same loop backwards: once any walk has
			// tried this wall, every pixel of the loop has been passed and this one changes nothing.
			uint32_t seen_wall = 0;
			if (lane == 0) seen_wall = (__hip_atomic_load(walked_r + (start >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (start & 31u)) & 1u;
			if (__builtin_amdgcn_readfirstlane(seen_wall)) { pos = start + 1u; continue; }
		}
		uint32_t already = start_visited ? 1u : 0u;
		uint32_t n = 0, mn = start, mn_pos = 0;
		const uint32_t room = a.raw_cap - tail;      // nodes this walk may still store
		uint32_t* __restrict__ out = raw + tail;
		if (m_start == 0u) {
			// an isolated pixel (dual_graph.hpp:168-171)
			if (lane == 0) { vis[node >> 5] |= 1u << (node & 31u); if (room) out[0] = node; }
			if (!room) flags |= kContourRawOverflow;
			n = 1;
		}
		else {
			// counterclockwise for |x, clockwise for x| (dual_graph.hpp:177)
			const bool clockwise = (m_start & kDirR) == 0u || (start_visited && m_start == (kDirU | kDirD));
			const ContourMoveTable tab = clockwise ? kContourClockwise : kContourCounter;
			bool wall_r = false;
			const uint32_t ending = contour_next_move(clockwise ? 1u : 3u, 3u /* UP */, m_start, wall_r);
			// Lane 0 walks alone (the other lanes wait at the end of the branch): its stores and LDS
			// updates need no per-instruction lane mask, and every value in the loop is still uniform,
			// so the loop stays on the scalar unit.
			uint32_t seen = 0, bad = 0;
			if (lane == 0) {
				if (room) out[0] = start; else bad = kContourRawOverflow;
				if (wall_r && walked_r) __hip_atomic_fetch_or(walked_r + (start >> 5), 1u << (start & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				n = 1;
				uint32_t next = ending;
				do {
					node += delta(next);
					if (node >= sxy || n >= room) { bad = node >= sxy ? kContourOpenWalk : kContourRawOverflow; break; }
					const uint32_t m = dir_of(node);      // issued before the visited bit is touched: the two latencies overlap
					out[n] = node;
					if (node < mn) { mn = node; mn_pos = n; }
					n++;
					// nothing on the walk depends on the old bit: swapped in and counted
					if (LDSVIS) seen += (atomicOr(&vis[node >> 5], 1u << (node & 31u)) >> (node & 31u)) & 1u;
					else {
						const uint32_t w = __hip_atomic_load(vis + (node >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						__hip_atomic_store(vis + (node >> 5), w | (1u << (node & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						seen += (w >> (node & 31u)) & 1u;
					}
					const uint32_t e = next * 16u + m;
					if (walked_r && ((tab.wall_r >> e) & 1ull)) __hip_atomic_fetch_or(walked_r + (node >> 5), 1u << (node & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					next = static_cast<uint32_t>(((e & 32u) ? tab.next_hi : tab.next_lo) >> (2u * (e & 31u))) & 3u;
				} while (!(node == start && next == ending));
			}
			n = __builtin_amdgcn_readfirstlane(n);
			mn = __builtin_amdgcn_readfirstlane(mn);
			mn_pos = __builtin_amdgcn_readfirstlane(mn_pos);
			flags |= __builtin_amdgcn_readfirstlane(bad);
			already += __builtin_amdgcn_readfirstlane(seen);
			total_steps += n - 1u;
		}
		pos = start + 1u;
		if (flags) break;
		if (n == 0u || n == already) continue;      // nothing new on this loop (dual_graph.hpp:199-201)
		if (n_contours < a.tab_cap) { if (lane == 0) table[n_contours] = make_uint4(tail, n, mn_pos, mn); }
		else { flags |= kContourTableOverflow; break; }
		n_contours++;
		tail += n;
	}
	if (lane == 0) {
		uint32_t* c = a.counts + 4u * zi;
		c[0] = n_contours; c[1] = tail; c[2] = flags; c[3] = total_steps;
	}
}

// component (within its slice) of every contour: the run of its smallest node.
// grid = (ceil(tab_cap / 256), nslices)
__global__ void __launch_bounds__(256) k_contour_components(
	RunGeom g, RunArrays r, const uint4* __restrict__ table, const uint32_t* __restrict__ counts, uint32_t tab_cap, uint32_t* __restrict__ comp
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= counts[4u * zi]) return;
	const uint32_t node = table[static_cast<uint64_t>(zi) * tab_cap + i].w;
	const uint32_t y = node / g.sx, x = node - y * g.sx;
	const uint32_t w = x >> 5;
	const uint32_t b = g.breaks(zi, y, w);
	const uint32_t run = r.word_base[zi * g.plane_words + static_cast<uint64_t>(y) * g.row_words + w] - 1u + __popc(b & mask_le(x & 31u));
	comp[static_cast<uint64_t>(zi) * tab_cap + i] = r.run_cc[r.rbase[zi] + run];
}

// the kept contours of all slices of a chunk packed one after the other (slice order, discovery
// order within a slice): one copy to the host instead of one per slice.  grid = (ceil(tab_cap / 256), nslices)
__global__ void __launch_bounds__(256) k_contour_pack(
	const uint4* __restrict__ table, const uint32_t* __restrict__ comp, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ base,
	uint32_t tab_cap, uint4* __restrict__ out_table, uint32_t* __restrict__ out_comp
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= counts[4u * zi]) return;
	out_table[base[zi] + i] = table[static_cast<uint64_t>(zi) * tab_cap + i];
	out_comp[base[zi] + i] = comp[static_cast<uint64_t>(zi) * tab_cap + i];
}

// index of every component's label in the sorted label table (the labels as the label map holds them)
__global__ void __launch_bounds__(256) k_component_label_index(
	const uint64_t* __restrict__ label_map, uint64_t n, const uint64_t* __restrict__ table, uint32_t n_table, uint32_t* __restrict__ out
) {
	const uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (c >= n) return;
	const uint64_t v = label_map[c];
	uint32_t lo = 0, hi = n_table;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (table[mid] < v) lo = mid + 1; else hi = mid;
	}
	out[c] = (lo < n_table && table[lo] == v) ? lo : 0xFFFFFFFFu;
}

struct ContourJob {
	uint64_t src;       // first node of the contour in the raw array (all slices)
	uint64_t dst;       // first point of the contour in the output
	uint32_t len, rot;  // nodes, position of the node that comes first
	uint32_t z, pad;
};

// grid = ceil(jobs / 4), block = 256: one wavefront per contour
__global__ void __launch_bounds__(256) k_contour_emit(const ContourJob* __restrict__ jobs, uint64_t n_jobs, const uint32_t* __restrict__ raw, uint32_t sx, uint16_t* __restrict__ out) {
	const uint64_t j = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
	if (j >= n_jobs) return;
	const ContourJob job = jobs[j];
	for (uint32_t i = threadIdx.x & 63u; i < job.len; i += 64u) {
		uint32_t k = i + job.rot;
		if (k >= job.len) k -= job.len;
		const uint32_t loc = raw[job.src + k];
		const uint16_t y = static_cast<uint16_t>(loc / sx);                       // 16-bit truncation as operations.hpp:246-247
		const uint16_t x = static_cast<uint16_t>(loc - sx * y);
		uint16_t* q = out + 3u * (job.dst + i);
		q[0] = x; q[1] = y; q[2] = static_cast<uint16_t>(job.z);
	}
}

}  // namespace dev
}  // namespace ckl
