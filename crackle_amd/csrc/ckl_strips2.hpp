// k_strip_ccl2: the strip kernel of ckl_strips.hpp (crack records -> plane pieces -> runs -> strip components)
// rebuilt for fewer vector instructions per wavefront (round 5).  Same inputs, same outputs (planes, run_lid, sc_w,
// strip_nruns / _nsc, row_run, seam_first / _last) as k_strip_ccl<*, true>; taken when a row is a power of two of
// plane words (>= 4: a thread's four words share a row) — every BASELINE.json shape —, k_strip_ccl serves the rest.
// Replaces the raster half of decode_crack_code (src/crackcodes.hpp:706-862), color_connectivity_graph inside a
// strip (src/cc3d.hpp:146-254) and the run part of the crc32c of the component image (src/crackle.hpp:599-611).
//
// What differs from k_strip_ccl:
//  * raster: a vertex is ONE linear integer relative to the strip, p = (y - y0) * S + x with S = 32 * row_words, so
//    that the plane word of a crack is p >> 5 and its bit p & 31 (no row / column split, no multiply); a move's
//    delta (-S, +1, +S, -1) comes out of a 64-bit constant shifted by 16 * kind; words outside the strip are
//    clamped to a dummy word instead of tested; moves along the image border are rasterised like any other and
//    masked once per plane word when the pieces are read back (column 0 and columns >= sx of plane V, row 0 of
//    plane H).  Range errors (a move that leaves the vertex grid) are found where the records are made
//    (k_crack_match: record_in_grid), not per move here.
//  * rows are powers of two of words: shifts instead of divisions, 16-byte LDS accesses for a thread's four words.
//  * the strip-local union-find takes the FIRST contact of a run with the row above as its parent with one
//    atomicMin per contact (a run's parent is a smaller index by construction); only contacts that find a link
//    already there go through sm_unite.  No edge list, no second block scan.
//  * roots are marked by wave ballots (64 consecutive runs = two bitmap words), not by LDS atomics.
#pragma once

#include "ckl_strips.hpp"

namespace ckl {
namespace dev {

struct Strip2Args {
	uint32_t rsh;       // log2(row_words)
	uint32_t variant;   // tuning builds: bit 0 = edge-list unions as in k_strip_ccl (A/B)
};

// Half a record (positions 8 h .. 8 h + 7) into the strip's plane pieces: `lds` = piece of plane V (words 0 .. nw,
// word nw is the dummy), piece of plane H kStripCap words behind it.  table: the four deltas, 16 bits each.
__device__ __forceinline__ void raster2_half(const uint4 rec, uint32_t h, uint32_t y0, uint32_t sh, uint32_t nw, unsigned long long table, uint32_t* lds) {
	constexpr uint32_t kLo = 0x55555555u;
	uint32_t kinds = rec.y, flags = rec.z;
	uint32_t p = (((rec.x >> 16) - y0) << sh) + (rec.x & 0xFFFFu);
	// the jump between the record's two stretches: a difference of packed vertices (y << 16 | x) -> linear
	const int32_t jdx = static_cast<int16_t>(rec.w & 0xFFFFu);
	const int32_t jdy = (static_cast<int32_t>(rec.w) - jdx) >> 16;
	const uint32_t jw = (static_cast<uint32_t>(jdy) << sh) + static_cast<uint32_t>(jdx);
	if (h) {
		// displacement of the emitting positions 0 .. 7, and the jump when the record's 't' lies among them
		const uint32_t ms = flags & kLo & 0xFFFFu, pv = kinds;
		const uint32_t mR = ms & ~(pv >> 1) & pv, mL = ms & (pv >> 1) & pv, mD = ms & (pv >> 1) & ~pv, mU = ms & ~(pv >> 1) & ~pv;
		p += __popc(mR) - __popc(mL) + ((__popc(mD) - __popc(mU)) << sh);
		if (flags & 0xAAAAu) p += jw;
		kinds >>= 16; flags >>= 16;
	}
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t idx = k < 2 ? (kinds << (4u - 2u * k)) & 0x30u : (kinds >> (2u * k - 4u)) & 0x30u;      // 16 * kind
		const uint32_t tv = static_cast<uint32_t>(table >> idx);
		const uint32_t e1 = (flags >> (2u * k)) & 1u;
		const int32_t jm = static_cast<int32_t>(flags << (30u - 2u * k)) >> 31;      // all ones at a 't'
		p += jw & static_cast<uint32_t>(jm);
		const int32_t d = static_cast<int32_t>(static_cast<int16_t>(tv)) * static_cast<int32_t>(e1);
		const uint32_t q = p + static_cast<uint32_t>(d);
		const uint32_t c = static_cast<int32_t>(p) < static_cast<int32_t>(q) ? p : q;      // the crack sits at the smaller vertex (signed: rows above the strip are negative)
		uint32_t w = c >> 5;
		w = w < nw ? w : nw;                   // outside the strip (above: wrapped to huge): the dummy word
		// horizontal moves (kind bit 0 = idx bit 4) cross plane H
		atomicOr(lds + w + (idx & 16u) * (kStripCap / 16u), e1 << (c & 31u));
		p = q;
	}
}

constexpr uint32_t kStrip2Words = kStripCclWords;

// per-workgroup cycle stamps of the tuning build: [workgroup][kStrip2Stamps] raw s_memtime values
constexpr uint32_t kStrip2Stamps = 8;

template <bool DIAG, bool EDGELIST>
__device__ __forceinline__ void strip_ccl2_body(
	const RunGeom& g, const StripArrays& sa, const RecordLists& rl, const uint32_t* __restrict__ G, uint32_t n_pixels, uint32_t rsh,
	unsigned long long* __restrict__ diag, uint32_t zi, uint32_t k, uint32_t* lds
) {
	uint32_t* s_par = lds;                           // piece of plane V, then the union-find, then the crc weights per strip component
	uint32_t* s_mem = s_par + kStripCap;             // piece of plane H, then s_b | s_pool
	uint32_t* s_b = s_mem;                           // break words of the strip
	uint16_t* s_pool = reinterpret_cast<uint16_t*>(s_mem + kStripWords);      // first pixel of each run (relative to the strip), later its strip component
	uint16_t* s_wb = reinterpret_cast<uint16_t*>(s_mem + kStripEdgeCap);      // runs before each word
	uint32_t* s_bm = s_mem + kStripEdgeCap + kStripWords / 2;
	uint32_t* s_bmbase = s_bm + kStripBitmapWords;
	uint32_t* s_scan = s_bmbase + kStripBitmapWords;
	const uint32_t t = threadIdx.x;
	unsigned long long* my_diag = DIAG ? diag + (static_cast<uint64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * kStrip2Stamps : nullptr;
	auto stamp = [&](int slot) { if (DIAG && t == 0) my_diag[slot] = __builtin_amdgcn_s_memtime(); };
	stamp(0);
	const uint32_t rw = 1u << rsh, sh = rsh + 5u;
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t nw = (y1 - y0) << rsh;
	const uint64_t slot = static_cast<uint64_t>(si) * sa.cap;
	// ---- the strip's records -> its pieces of the two planes
	{
		const uint32_t n_rec = min(rl.count[si], rl.cap);
		const uint4* list = rl.rec + static_cast<uint64_t>(si) * rl.cap;
		const uint4 first = list[(t >> 1) < rl.cap ? (t >> 1) : 0u];      // requested with the count, not behind it
		const uint4 second = list[(t >> 1) + kBlock / 2u < rl.cap ? (t >> 1) + kBlock / 2u : 0u];      // (most strips of a 1024-wide slice hold 130 - 300 records)
		*reinterpret_cast<uint4*>(s_par + t * 4u) = make_uint4(0u, 0u, 0u, 0u);
		*reinterpret_cast<uint4*>(s_mem + t * 4u) = make_uint4(0u, 0u, 0u, 0u);
		__syncthreads();
		const uint32_t S = 1u << sh;
		const unsigned long long table = static_cast<unsigned long long>((0u - S) & 0xFFFFu) | (1ull << 16) | (static_cast<unsigned long long>(S) << 32) | (0xFFFFull << 48);
		const uint32_t half = t & 1u;
		if ((t >> 1) < n_rec) raster2_half(first, half, y0, sh, nw, table, s_par);
		if ((t >> 1) + kBlock / 2u < n_rec) raster2_half(second, half, y0, sh, nw, table, s_par);
		for (uint32_t r = (t >> 1) + kBlock; r < n_rec; r += kBlock / 2u) raster2_half(list[r], half, y0, sh, nw, table, s_par);
		__syncthreads();
	}
	stamp(1);
	// ---- my four plane words (one row), masked, out to HBM, and as breaks / connections
	const uint32_t row = (t * 4u) >> rsh, w0 = (t * 4u) & (rw - 1u);
	const bool in_strip = t * 4u < nw;
	uint32_t b[4], up[4], upl0;
	{
		uint4 v4 = *reinterpret_cast<const uint4*>(s_par + t * 4u);
		uint4 h4 = *reinterpret_cast<const uint4*>(s_mem + t * 4u);
		const uint32_t hl = s_mem[t ? t * 4u - 1u : 0u];
		const uint32_t tail = g.sx & 31u;
		const uint32_t lastmask = (w0 + 4u == rw && tail) ? (1u << tail) - 1u : 0xFFFFFFFFu;
		if (w0 == 0u) v4.x &= ~1u;                 // moves along the left border cross nothing
		v4.w &= lastmask; h4.w &= lastmask;        // nor do moves along the right border
		if (y0 + row == 0u) h4 = make_uint4(0u, 0u, 0u, 0u);      // nor moves along the top border
		if (in_strip) {
			uint32_t* pv = const_cast<uint32_t*>(g.planeV) + zi * g.plane_words + (static_cast<uint64_t>(y0) << rsh);
			uint32_t* ph = const_cast<uint32_t*>(g.planeH) + zi * g.plane_words + (static_cast<uint64_t>(y0) << rsh);
			*reinterpret_cast<uint4*>(pv + t * 4u) = v4;
			*reinterpret_cast<uint4*>(ph + t * 4u) = h4;
		}
		const uint32_t fm = g.flip ? 0u : 0xFFFFFFFFu;
		const uint32_t on = in_strip ? 0xFFFFFFFFu : 0u;
		const uint32_t onu = (in_strip && row) ? 0xFFFFFFFFu : 0u;      // rows 1.. of the strip: connections to the row above
		b[0] = ((v4.x ^ fm) | (w0 == 0u ? 1u : 0u)) & on;
		b[1] = (v4.y ^ fm) & on;
		b[2] = (v4.z ^ fm) & on;
		b[3] = (v4.w ^ fm) & lastmask & on;
		const uint32_t fu = ~fm;                   // (a crack of an IMPERMISSIBLE stream is a boundary: connected where plane H has none)
		up[0] = (h4.x ^ fu) & onu;
		up[1] = (h4.y ^ fu) & onu;
		up[2] = (h4.z ^ fu) & onu;
		up[3] = (h4.w ^ fu) & lastmask & onu;
		upl0 = w0 ? (hl ^ fu) & onu : 0u;          // the word before mine in my row (never the row's last)
	}
	uint32_t l[4];      // runs before each of my words
	uint32_t nloc;
	{
		const uint32_t c0 = __popc(b[0]), c1 = __popc(b[1]), c2 = __popc(b[2]), c3 = __popc(b[3]);
		uint32_t v[1] = { c0 + c1 + c2 + c3 }, tot[1];
		block_excl_add<1>(v, tot, s_scan);      // its barriers: every read of the pieces is done, s_par / s_mem take their tables
		nloc = tot[0];
		l[0] = v[0]; l[1] = l[0] + c0; l[2] = l[1] + c1; l[3] = l[2] + c2;
	}
	if (nloc > sa.cap) {      // uniform: the general pipeline takes over (host)
		if (t == 0) { sa.strip_nruns[si] = kStripOverflow; sa.strip_nsc[si] = 0u; atomicOr(sa.overflow, 1u); }
		return;
	}
	stamp(2);
	// ---- break words, run prefixes, first pixel of every run; per-row and seam tables
	if (in_strip) {
		*reinterpret_cast<uint4*>(s_b + t * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
		const uint2 wb2 = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
		*reinterpret_cast<uint2*>(s_wb + t * 4u) = wb2;
		if (w0 == 0u) sa.row_run[static_cast<uint64_t>(zi) * g.sy + y0 + row] = static_cast<uint16_t>(l[0]);      // (pins look pixels up)
		if (row == 0u) *reinterpret_cast<uint2*>(sa.seam_first + (static_cast<uint64_t>(si) << rsh) + w0) = wb2;
		if (t * 4u + rw >= nw) *reinterpret_cast<uint2*>(sa.seam_last + (static_cast<uint64_t>(si) << rsh) + w0) = wb2;
		const uint32_t px = row * g.sx + w0 * 32u;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			uint32_t at = l[j];
			for (uint32_t m = b[j]; m; m &= m - 1u) s_pool[at++] = static_cast<uint16_t>(px + j * 32u + (__ffs(m) - 1u));
		}
	}
	for (uint32_t j = t; j < nloc; j += kBlock) s_par[j] = j;
	__syncthreads();
	stamp(3);
	// the crc weights of my runs are requested now and collected after the unions
	uint32_t gv[kStripRunsPerThread];
	{
		const uint32_t p0 = y0 * g.sx;
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			if (i * kBlock < nloc) gv[i] = G[n_pixels - (p0 + s_pool[j < nloc ? j : 0u])];      // uniform condition
		}
	}
	// ---- unions between vertically adjacent runs of the strip (first contact of each pair)
	{
		const uint32_t at_up = (in_strip && row) ? t * 4u - rw : 0u;
		const uint4 bu4 = *reinterpret_cast<const uint4*>(s_b + at_up);
		const uint2 lu2 = *reinterpret_cast<const uint2*>(s_wb + at_up);
		const uint32_t b_up[4] = { bu4.x, bu4.y, bu4.z, bu4.w };
		const uint32_t l_up[4] = { lu2.x & 0xFFFFu, lu2.x >> 16, lu2.y & 0xFFFFu, lu2.y >> 16 };
		uint32_t c[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t prev = j ? up[j - 1] : upl0;
			c[j] = up[j] & (~((up[j] << 1) | (prev >> 31)) | b[j] | b_up[j]);
		}
		if constexpr (EDGELIST) {
			const uint32_t n_e = __popc(c[0]) + __popc(c[1]) + __popc(c[2]) + __popc(c[3]);
			uint32_t ve[1] = { n_e }, te[1];
			block_excl_add<1>(ve, te, s_scan);      // its barriers: every read of s_b / s_pool is done
			const uint32_t n_edges = te[0];
			if (n_edges <= kStripEdgeCap) {      // uniform
				uint32_t at = ve[0];
#pragma unroll
				for (uint32_t j = 0; j < 4; j++) {
					for (uint32_t cc = c[j]; cc; cc &= cc - 1u) {
						const uint32_t m = cc ^ (cc - 1u);
						s_mem[at++] = (l[j] + __popc(b[j] & m) - 1u) | ((l_up[j] + __popc(b_up[j] & m) - 1u) << 16);
					}
				}
				__syncthreads();
				for (uint32_t e = t; e < n_edges; e += kBlock) { const uint32_t pr = s_mem[e]; sm_unite(s_par, pr & 0xFFFFu, pr >> 16); }
			}
			else {
#pragma unroll
				for (uint32_t j = 0; j < 4; j++) {
					for (uint32_t cc = c[j]; cc; cc &= cc - 1u) {
						const uint32_t m = cc ^ (cc - 1u);
						sm_unite(s_par, l[j] + __popc(b[j] & m) - 1u, l_up[j] + __popc(b_up[j] & m) - 1u);
					}
				}
			}
		}
		else {
			// a run's first contact with the row above becomes its parent: one atomicMin (the run above has the
			// smaller index); a contact that finds another link in place unites the two runs above
#pragma unroll
			for (uint32_t j = 0; j < 4; j++) {
				for (uint32_t cc = c[j]; cc; cc &= cc - 1u) {
					const uint32_t m = cc ^ (cc - 1u);
					const uint32_t jh = l[j] + __popc(b[j] & m) - 1u, ju = l_up[j] + __popc(b_up[j] & m) - 1u;
					const uint32_t old = atomicMin(s_par + jh, ju);
					if (old != jh && old != ju) sm_unite(s_par, ju, old);
				}
			}
		}
	}
	__syncthreads();      // the pool now takes the strip components
	stamp(4);
	// ---- roots -> strip-local component ids in run order; the roots of 64 consecutive runs are one ballot
	// (all finds of a thread advancing together, one hop per round, were slower: 0.169 against 0.148 ms — every slot
	// reads the table every round until the slowest lane of the wavefront is through, and LDS traffic is what this
	// kernel is short of)
	uint32_t root[kStripRunsPerThread];
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		if (i * kBlock >= nloc) break;      // uniform
		const uint32_t j = t + i * kBlock;
		root[i] = j < nloc ? sm_find(s_par, j) : 0xFFFFFFFFu;
		const unsigned long long roots = __ballot(root[i] == j);
		if ((t & 31u) == 0u) s_bm[(j >> 5) * 2u] = (t & 32u) ? static_cast<uint32_t>(roots >> 32) : static_cast<uint32_t>(roots);      // (bits, base) pairs: s_bm2 below
	}
	__syncthreads();
	// root bitmap -> (bits, roots before the word) pairs: a run's strip component is one 8-byte LDS read away
	uint2* s_bm2 = reinterpret_cast<uint2*>(s_bm);      // [kStripBitmapWords] over s_bm | s_bmbase
	// every wavefront scans the whole bitmap for itself and writes all the bases (the same values four times over:
	// a wavefront then reads what it wrote itself, and the barrier behind a scan by one wavefront goes)
	uint32_t nsc;
	{
		static_assert(kStripBitmapWords <= 2 * kWave, "two bitmap words per lane");
		const uint32_t ln = t & 63u;
		const uint32_t nbw = (nloc + 31u) >> 5;
		const uint32_t c0 = ln < nbw ? __popc(s_bm2[ln].x) : 0u;
		const uint32_t c1 = ln + kWave < nbw ? __popc(s_bm2[ln + kWave < kStripBitmapWords ? ln + kWave : 0u].x) : 0u;
		const uint32_t i0 = wave_incl_add(c0);
		const uint32_t tot0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i0), kWave - 1));
		const uint32_t i1 = wave_incl_add(c1);
		if (ln < nbw) s_bm2[ln].y = i0 - c0;
		if (ln + kWave < nbw) s_bm2[ln + kWave].y = tot0 + i1 - c1;
		nsc = tot0 + static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i1), kWave - 1));
	}
	uint32_t lid[kStripRunsPerThread];
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		if (i * kBlock >= nloc) break;      // uniform
		const uint32_t j = t + i * kBlock;
		const uint32_t r = j < nloc ? root[i] : 0u;
		const uint2 e = s_bm2[r >> 5];
		lid[i] = e.y + __popc(e.x & ((1u << (r & 31u)) - 1u));
		if (j < nloc) s_pool[j] = static_cast<uint16_t>(lid[i]);
	}
	for (uint32_t j = t; j < nsc; j += kBlock) s_par[j] = 0u;      // every find is done (the barriers of the bitmap scan): the table becomes the weights
	__syncthreads();
	stamp(5);
	// the runs' strip components to HBM, four per thread and store: one byte per run while the strip has at most 256
	// components (strip_lid: the paint kernel reads them beside its stores, where every byte read costs several
	// bytes' worth of store time)
	if (nsc <= 256u) {
		uint32_t* out4 = reinterpret_cast<uint32_t*>(sa.run_lid + slot);
		for (uint32_t j4 = t * 4u; j4 < nloc; j4 += kBlock * 4u) {
			const uint2 q = *reinterpret_cast<const uint2*>(s_pool + j4);
			out4[j4 >> 2] = (q.x & 0xFFu) | ((q.x >> 8) & 0xFF00u) | ((q.y & 0xFFu) << 16) | ((q.y >> 16) << 24);
		}
	}
	else {
		uint2* out8 = reinterpret_cast<uint2*>(sa.run_lid + slot);
		for (uint32_t j4 = t * 4u; j4 < nloc; j4 += kBlock * 4u) out8[j4 >> 2] = *reinterpret_cast<const uint2*>(s_pool + j4);
	}
	// ---- crc weights: run j covering [a_j, a_j+1) adds G[n - a_j] ^ G[n - a_j+1] to its component; the start
	// weight of run j + 1 comes from the lane above (the wavefront's last lane leaves it to that run's own thread)
	{
		const uint32_t g_end = G[n_pixels - y1 * g.sx];
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			if (i * kBlock >= nloc) break;      // uniform
			const uint32_t j = t + i * kBlock;
			const uint32_t mine = j < nloc ? gv[i] : 0u;
			uint32_t x = mine ^ dpp_u32<kDppWaveShl1, 0xF>(0u, mine);
			if (j + 1u == nloc) x ^= g_end;
			if (j < nloc) atomicXor(s_par + lid[i], x);
			if ((t & 63u) == 0u && j && j < nloc) atomicXor(s_par + s_pool[j - 1u], mine);
		}
	}
	__syncthreads();
	uint32_t* w_out = sa.sc_w + slot;
	for (uint32_t j = t; j < nsc; j += kBlock) w_out[j] = s_par[j];
	if (t == 0) { sa.strip_nruns[si] = nloc; sa.strip_nsc[si] = nsc; }
	stamp(6);
}

template <bool DIAG, bool EDGELIST>
static __global__ void __launch_bounds__(kBlock, 7) k_strip_ccl2(RunGeom g, StripArrays sa, RecordLists rl, const uint32_t* __restrict__ G, uint32_t n_pixels, Strip2Args a2, unsigned long long* __restrict__ diag) {
	__shared__ __attribute__((aligned(16))) uint32_t s_lds[kStrip2Words];
	uint32_t zl, k;
	if (sa.layout & 4u) strip_of_block(sa, zl, k);
	else { zl = blockIdx.y; k = blockIdx.x; }
	strip_ccl2_body<DIAG, EDGELIST>(g, sa, rl, G, n_pixels, a2.rsh, diag, zl + sa.zbase, k, s_lds);
}

}  // namespace dev
}  // namespace ckl
