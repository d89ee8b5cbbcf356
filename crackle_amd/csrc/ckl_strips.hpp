// Strip-component labelling: the decoder's fast path from crack planes to painted labels.
// Replaces cc3d::color_connectivity_graph + relabel (src/cc3d.hpp:114-254), the crc32c of the
// component image (src/crackle.hpp:599-611), label_map[cc] (src/labels.hpp:453-506) and the
// paint loop (src/crackle.hpp:617-656) with three kernels whose intermediates are per RUN
// (2 + sizeof(label) bytes) and per STRIP COMPONENT instead of per voxel:
//
//   k_strip_ccl      one workgroup per strip of rows (<= 1024 plane words): runs of the strip from
//                    the V plane, union-find over them in LDS through the H plane, strip-local
//                    component of every run (16 bits), and per strip component the XOR of its runs'
//                    crc weights G[n-a] ^ G[n-b] (the crc32c of the component image is linear in
//                    them: crc = XOR over components of id (x) W, see ckl_runs.hpp).
//   k_slice_resolve  one workgroup per slice: union-find over the strip components (a few thousand
//                    per slice instead of ~50 k runs) across the strip seams, ranks the roots in
//                    raster order of their first pixel = the reference's component ids, checks count
//                    and crc32c, maps ids to labels: one label per strip component.
//   k_paint_strips   one workgroup per strip: the strip's plane words and their run prefix are built
//                    in LDS (no per-word table in HBM), the label of every run is staged there
//                    (run -> strip component -> label), then 16-byte streaming stores.
//
// A strip with more runs than the LDS tables hold, or a slice with more strip components than
// k_slice_resolve's table, raises `overflow`: the host then runs the general run pipeline of
// ckl_runs.hpp on the same planes (dense / noisy volumes).
#pragma once

#include "ckl_runs.hpp"
#include <type_traits>

namespace ckl {
namespace dev {

constexpr uint32_t kStripWords = 1024;      // plane words per strip: 4 per thread
constexpr uint32_t kStripCap = 2560;        // runs per strip held in LDS (the host sizes the strips for ~0.7 of it)
constexpr uint32_t kStripRunsPerThread = kStripCap / kBlock;
constexpr uint32_t kStripBitmapWords = kStripCap / 32;
constexpr uint32_t kStripEdgeCap = kStripWords + kStripCap / 2;      // 32-bit words of the break words + the 16-bit run pool
constexpr uint32_t kStripOverflow = 0xFFFFFFFFu;
constexpr int kResolveBlock = 1024;
constexpr uint32_t kResolveCap = 0xFFFFu;   // strip components per slice: index and rank share a table entry; the table is dynamic LDS, sized by the host (ResolveArgs::cap)
constexpr uint32_t kMaxStrips = 1024;

struct StripArrays {
	// per strip (index si = slice * nstrips + strip); the per-run and per-strip-component arrays
	// give every strip a fixed slot of `cap` entries, so that no kernel waits for another's counts
	uint16_t* run_lid;         // [strips][cap] strip-local component of each run
	uint32_t* sc_w;            // [strips][cap] per strip component: XOR of its runs' crc weights
	uint32_t* sc_cc;           // [strips][cap] per strip component: component id of the slice (pins)
	void* sc_label;            // [strips][cap] per strip component: label, typed like the output
	uint32_t* strip_nruns;     // [strips] (kStripOverflow: the strip did not fit)
	uint32_t* strip_nsc;       // [strips] strip components
	uint16_t* row_run;         // [nslices][sy] runs of the strip before each row (pins look pixels up)
	uint16_t* seam_first;      // [strips][row_words] runs of the strip before each word of its first row
	uint16_t* seam_last;       // ... of its last row
	// k_strip_fused hands these three over inside the launch, as 32-bit words (the in-launch hand-off is kept to
	// 4-byte accesses): the seam prefixes, and the strip component of every run of a strip's first and last row
	uint32_t* seam32_first;    // [strips][row_words]
	uint32_t* seam32_last;
	uint32_t* lid32;           // [strips][cap]
	uint32_t* slice_err;
	uint32_t* overflow;        // one word
	uint32_t nstrips, strip_rows;
	uint32_t cap;              // slot size: min(kStripCap, pixels of a strip)
	uint32_t zbase;            // first slice of this launch (z-chunked launches)
	uint32_t ablate;           // tuning aid (CKL_ABLATE): skips parts of the strip kernels, results are wrong
	uint32_t layout;           // bit 0: workgroup -> strip mapping that gives each XCD a contiguous eighth of the launch's strips (strip_of_block);
	                           // bit 1: the paint's wavefronts stream a contiguous quarter of the strip each
};

// Which strip does workgroup (blockIdx.x, blockIdx.y) of a grid (nstrips, slices) serve?  Workgroups go to the eight
// XCDs round robin in launch order, so with strip = launch index every XCD touches every 128 KiB stretch of the
// volume in turn; with the launch index' low three bits as the HIGH part of the strip number each XCD streams
// through one contiguous eighth of it (its own L2 and TLB see one range).  A pure fill of 2 GiB: 5.9 -> 6.4 TB/s with
// a quarter of each 128 KiB chunk per wavefront, 6.6 TB/s with this mapping on top (tools/micro/store_bw.hip).
__device__ __forceinline__ void strip_of_block(const StripArrays& sa, uint32_t& zi_local, uint32_t& k) {
	if ((sa.layout & 1u) == 0u) { zi_local = blockIdx.y; k = blockIdx.x; return; }
	const uint32_t total = gridDim.x * gridDim.y, i = blockIdx.y * gridDim.x + blockIdx.x;
	const uint32_t body = total & ~7u;      // (a grid that is no multiple of eight: its last workgroups keep their own number)
	const uint32_t c = i < body ? (i & 7u) * (body >> 3) + (i >> 3) : i;
	zi_local = c / gridDim.x;
	k = c - zi_local * gridDim.x;
}

// tuning builds only: is part `mask` of a strip kernel switched off (CKL_ABLATE)
__device__ __forceinline__ bool ablated(const StripArrays& sa, uint32_t mask) { return kTuning && (sa.ablate & mask) != 0u; }

// strip component of run j of a strip with nsc components: stored in one byte while nsc <= 256
__device__ __forceinline__ uint32_t strip_lid(const uint16_t* slot_lids, uint32_t nsc, uint32_t j) {
	return nsc <= 256u ? reinterpret_cast<const uint8_t*>(slot_lids)[j] : slot_lids[j];
}

// Loads / stores of bytes that another workgroup of the SAME launch wrote / will read (k_strip_fused): agent-scope
// relaxed atomics = `global_load/store ... sc1` — the store writes through the XCD's L2, the load bypasses this CU's
// L1 (MI355X: per-XCD L2s are not coherent, L1 is never refreshed by other CUs' stores).  H = false: plain accesses
// (the data crossed a kernel boundary).
template <bool H, typename T>
__device__ __forceinline__ T hand_ld(const T* p) {
	if constexpr (H) return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else return *p;
}
#ifdef CKL_TUNING
__device__ uint32_t g_exp_flags;      // CKL_EXP (tuning builds): timing experiments on k_strip_fused, results may be wrong
#endif
__device__ __forceinline__ bool exp_on(uint32_t bit) {
#ifdef CKL_TUNING
	return (g_exp_flags & bit) != 0u;
#else
	return false;
#endif
}
template <bool H, typename T>
__device__ __forceinline__ void hand_st(T* p, T v) {
	if constexpr (H) {
		if (exp_on(1u)) *p = v;
		else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	else *p = v;
}

// ---- crack records (k_crack_records, ckl_crack_records.hpp) -----------------------------------------
// the lists the records go to: one slot of `cap` records per strip
struct RecordLists {
	uint4* rec;                  // [strips of all slices][cap]
	uint32_t* count;             // [strips of all slices]
	uint32_t cap;                // records per list
	uint32_t nstrips, strip_rows;
	uint32_t strip_shift;        // log2(strip_rows) when it is a power of two, else 0xFFFFFFFF
	uint32_t strip_magic;        // floor(2^32 / strip_rows) + 1: y / strip_rows = high word of y * magic for y < 65536 (the error term y * (magic * rows - 2^32) stays below 2^32)
	// (two of these per record: as a division by a run-time number it was ~20 instructions each, a tenth of k_crack_match's vector work)
	__device__ __forceinline__ uint32_t strip_of(uint32_t y) const { return strip_shift != 0xFFFFFFFFu ? y >> strip_shift : __umulhi(y, strip_magic); }
};

// Walks one record — x: the vertex before its first move, packed y << 16 | x; y: the 16 moves of its
// word, 2 bits each (0 up, 1 right, 2 down, 3 left); z: bit 2k set when position k emits its move, bit
// 2k + 1 when position k is a 't' (at most one per record): the walk jumps by w there — and sets the
// cracks its moves cross in rows [y0, y0 + rows) of the strip's plane pieces in LDS
// (crackcodes.hpp:706-862: a vertical move crosses plane V, a horizontal one plane H, at the smaller of
// its two vertices; moves along the outer border cross nothing).  `bad` is raised when a move leaves the
// vertex grid.
__device__ __forceinline__ void raster_record(
	const uint4 rec, uint32_t y0, uint32_t rows, uint32_t sx, uint32_t sy, uint32_t rw, uint32_t* sV, uint32_t* sH, uint32_t& bad
) {
	uint32_t p = rec.x;
	const uint32_t jump = rec.z & 0xAAAAAAAAu;
#pragma unroll
	for (uint32_t k = 0; k < 16; k++) {
		if ((jump >> (2u * k + 1u)) & 1u) p += rec.w;
		if (((rec.z >> (2u * k)) & 1u) == 0u) continue;
		const uint32_t kind = (rec.y >> (2u * k)) & 3u;
		const uint32_t horiz = kind & 1u;
		const uint32_t unit = horiz ? 1u : 0x10000u;
		const uint32_t neg = ((kind ^ (kind >> 1)) & 1u) ^ 1u;      // up (0) and left (3)
		const uint32_t q = neg ? p - unit : p + unit;
		const uint32_t c = p < q ? p : q;
		const uint32_t col = c & 0xFFFFu, row = c >> 16;
		const uint32_t qx = q & 0xFFFFu, qy = q >> 16;
		if (qx > sx || qy > sy) bad |= 1u;
		else {
			const bool ok = horiz ? (row - 1u < sy - 1u && col < sx) : (col - 1u < sx - 1u && row < sy);
			const uint32_t rel = row - y0;
			if (ok && rel < rows) atomicOr((horiz ? sH : sV) + rel * rw + (col >> 5), 1u << (col & 31u));
		}
		p = q;
	}
}

// The same for HALF a record — positions 8 h .. 8 h + 7 — without a branch per move: two lanes share a record, so
// that a strip's ~250 records keep all 256 lanes busy for 8 steps instead of some of them for 16 (a strip with
// more than 128 records paid a second round of 16 for a handful of them).  The half's start is the record's start
// plus the displacement of the moves in front (four popcounts); a move that crosses no crack of this strip ORs a
// zero into word 0 instead of being skipped (no exec-mask juggling in the unrolled body: 813 SALU instructions per
// wavefront went with it).  `lds_h` / `lds_v`: the strip's plane pieces; rlo_h: first row whose upper crack lies in
// the strip and inside the image (max(y0, 1)).
__device__ __forceinline__ void raster_half_record(
	const uint4 rec, uint32_t h, uint32_t y0, uint32_t y1, uint32_t rlo_h, uint32_t sx, uint32_t sy, uint32_t rw, uint32_t* lds_v, uint32_t* lds_h, uint32_t& bad
) {
	constexpr uint32_t kLo = 0x55555555u;
	uint32_t p = rec.x;
	uint32_t kinds = rec.y, flags = rec.z;
	if (h) {
		// displacement of positions 0 .. 7 (the emitting ones), and the jump when the record's 't' lies among them
		const uint32_t ms = flags & kLo & 0xFFFFu, pv = kinds;
		const uint32_t mR = ms & ~(pv >> 1) & pv, mL = ms & (pv >> 1) & pv, mD = ms & (pv >> 1) & ~pv, mU = ms & ~(pv >> 1) & ~pv;
		p += __popc(mR) - __popc(mL) + ((__popc(mD) - __popc(mU)) << 16);
		if (flags & 0xAAAAu) p += rec.w;
		kinds >>= 16; flags >>= 16;
	}
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t j = (flags >> (2u * k + 1u)) & 1u;
		p += rec.w & (0u - j);
		const uint32_t e = (flags >> (2u * k)) & 1u;
		const uint32_t kind = (kinds >> (2u * k)) & 3u;
		const uint32_t horiz = kind & 1u;
		const uint32_t neg = ((kind ^ (kind >> 1)) & 1u) ^ 1u;      // up (0) and left (3)
		const uint32_t unit = horiz ? 1u : 0x10000u;
		const uint32_t d = ((unit ^ (0u - neg)) + neg) & (0u - e);
		const uint32_t q = p + d;
		const uint32_t c = p < q ? p : q;
		const uint32_t col = c & 0xFFFFu, row = c >> 16;
		bad |= ((q & 0xFFFFu) > sx || (q >> 16) > sy) ? 1u : 0u;
		const uint32_t rlo = horiz ? rlo_h : y0, clo = horiz ? 0u : 1u;
		const bool ok = e != 0u && (row - rlo) <= (y1 - 1u - rlo) && row >= rlo && (col - clo) <= (sx - 1u - clo) && col >= clo;
		uint32_t* base = horiz ? lds_h : lds_v;
		const uint32_t at = ok ? (row - y0) * rw + (col >> 5) : 0u;
		atomicOr(base + at, ok ? (1u << (col & 31u)) : 0u);
		p = q;
	}
}

// grid = (nstrips, slices of the launch), block = kBlock.  The kernel waits on LDS round trips
// (union-find), so what counts is the number of resident wavefronts: <= 72 registers and 22 KiB of
// LDS keep seven workgroups on a CU.  (A persistent variant that fetched the next strip's words
// while working on the current one needed 113 registers and was slower: 0.36 against 0.29 ms.)
// RECORDS: the strip's plane pieces are rasterised here, in LDS, from the strip's list of crack records
// (k_crack_records) and written to the planes in HBM on the way (the paint kernel and the seams of
// k_slice_resolve read them there); otherwise they are read from the planes k_decode_cracks left.
// The body works in `lds` (kStripCclWords 32-bit words, 16-byte aligned) on strip k of slice zi, so that
// the caller owns the block (a launch mixing stages of different slice groups was tried: DESIGN.md section 10).
constexpr uint32_t kStripCclWords = (kStripCap + kStripEdgeCap + kStripWords / 2 + 2 * kStripBitmapWords + kWaves + 2 + 7) & ~7u;
// FUSED (k_strip_fused: the same workgroup paints the strip once its slice is resolved): the break words, the run
// prefixes and the runs' strip components stay in LDS (s_b, s_wb, s_pool) and only what the slice's resolver reads
// leaves the workgroup, with write-through stores (hand_st): counts, weights, the seam rows of the planes and the
// strip components of the runs of the first and the last row.  Returns the strip's runs (kStripOverflow: too many).
template <bool DIAG, bool RECORDS, bool FUSED = false>
__device__ __forceinline__ uint32_t strip_ccl_body(
	const RunGeom& g, const StripArrays& sa, const RecordLists& rl, const uint32_t* __restrict__ G, uint32_t n_pixels, unsigned long long* __restrict__ diag,
	uint32_t zi, uint32_t k, uint32_t* lds, uint32_t* nsc_ret = nullptr
) {
	uint32_t* s_parent = lds;                         // union-find, then the crc weights per strip component (RECORDS: first the piece of plane H)
	uint32_t* s_mem = s_parent + kStripCap;           // s_b | s_pool, and over both of them the edge list of the unions
	uint16_t* s_wb = reinterpret_cast<uint16_t*>(s_mem + kStripEdgeCap);          // runs before each word
	uint32_t* s_b = s_mem;                                                        // break words of the strip
	uint16_t* s_pool = reinterpret_cast<uint16_t*>(s_mem + kStripWords);          // first pixel of each run (relative to the strip), later its strip component
	uint32_t* s_bm = s_mem + kStripEdgeCap + kStripWords / 2;
	uint32_t* s_bmbase = s_bm + kStripBitmapWords;
	uint32_t* s_scan = s_bmbase + kStripBitmapWords;
	uint32_t* s_misc = s_scan + kWaves;
	static_assert((kStripCap * 4) % 16 == 0 && (kStripEdgeCap * 4) % 16 == 0, "tables stay 16-byte aligned");
	unsigned long long d_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) {
		if (DIAG && threadIdx.x == 0) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			atomicAdd(diag + slot, now - d_t);
			d_t = now;
		}
	};
	const uint32_t rw = g.row_words;
	const uint32_t t = threadIdx.x;
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t nw = (y1 - y0) * rw;
	const uint64_t slot = static_cast<uint64_t>(si) * sa.cap;
	// ---- plane words: every load is issued at once, none sits in a branch (a load inside a branch
	// is waited for there); words past the strip read word 0
	uint32_t b[4], up[4], upl[4], cnt = 0;
	if (RECORDS) {
		// the strip's records -> its pieces of the two planes, in LDS (s_mem: plane V, later the break
		// words; s_parent: plane H, dead once the words are in registers)
		static_assert(kStripCap >= kStripWords, "plane H piece is built in the union-find table");
		uint32_t* sV = s_mem;
		uint32_t* sH = s_parent;
		const uint32_t n_rec = min(rl.count[si], rl.cap);
		const uint4* list = rl.rec + static_cast<uint64_t>(si) * rl.cap;
		uint4 first = list[(t >> 1) < rl.cap ? (t >> 1) : 0u];      // requested with the count, not behind it
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) { sV[t * 4u + j] = 0u; sH[t * 4u + j] = 0u; }
		__syncthreads();
		uint32_t bad = 0;
		{
			// two lanes per record, eight positions each
			const uint32_t half = t & 1u, rlo_h = y0 ? y0 : 1u;
			if ((t >> 1) < n_rec) raster_half_record(first, half, y0, y1, rlo_h, g.sx, g.sy, rw, sV, sH, bad);
			for (uint32_t r = (t >> 1) + kBlock / 2u; r < n_rec; r += kBlock / 2u) raster_half_record(list[r], half, y0, y1, rlo_h, g.sx, g.sy, rw, sV, sH, bad);
		}
		if (bad) atomicOr(sa.slice_err + zi, ERR_RANGE);
		__syncthreads();
		stamp(5);
		const uint4 v4 = *reinterpret_cast<const uint4*>(sV + t * 4u);
		const uint4 h4 = *reinterpret_cast<const uint4*>(sH + t * 4u);
		b[0] = v4.x; b[1] = v4.y; b[2] = v4.z; b[3] = v4.w;
		up[0] = h4.x; up[1] = h4.y; up[2] = h4.z; up[3] = h4.w;
		upl[0] = sH[t ? t * 4u - 1u : 0u];
		if constexpr (FUSED) {
			// only the rows the resolver unites across: V and H of the first row, V of the last
			uint32_t* pv = const_cast<uint32_t*>(g.planeV) + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
			uint32_t* ph = const_cast<uint32_t*>(g.planeH) + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
#pragma unroll
			for (uint32_t j = 0; j < 4; j++) {
				const uint32_t wl = t * 4u + j;
				if (wl < rw) { hand_st<true>(pv + wl, b[j]); hand_st<true>(ph + wl, up[j]); }
				else if (wl < nw && wl + rw >= nw) hand_st<true>(pv + wl, b[j]);
			}
		}
		// planes to HBM: 16 bytes per thread, the strip's rows are contiguous
		else if (t * 4u < nw) {
			uint32_t* pv = const_cast<uint32_t*>(g.planeV) + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
			uint32_t* ph = const_cast<uint32_t*>(g.planeH) + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
			if (t * 4u + 4u <= nw) {
				*reinterpret_cast<uint4*>(pv + t * 4u) = v4;
				*reinterpret_cast<uint4*>(ph + t * 4u) = h4;
			}
			else {
				for (uint32_t j = 0; t * 4u + j < nw; j++) { pv[t * 4u + j] = b[j]; ph[t * 4u + j] = up[j]; }
			}
		}
		__syncthreads();      // every read of the pieces is done: s_mem / s_parent take their tables
	}
	else {
		const uint32_t* pv = g.planeV + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
		const uint32_t* ph = g.planeH + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			const uint32_t at = wl < nw ? wl : 0u;
			b[j] = pv[at];
			up[j] = ph[at];
		}
		upl[0] = ph[(t * 4u < nw && t) ? t * 4u - 1u : 0u];
	}
	// word t*4+j of the strip sits at word w[j] of its row
	uint32_t w[4];
	{
		const uint32_t yy = (t * 4u) / rw;
		uint32_t ww = t * 4u - yy * rw;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) { w[j] = ww; if (++ww == rw) ww = 0; }
	}
	upl[0] = (w[0] && t * 4u >= rw && t * 4u < nw) ? g.ups_of(upl[0], w[0] - 1u) : 0u;
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t wl = t * 4u + j;
		b[j] = wl < nw ? g.breaks_of(b[j], w[j]) : 0u;
		up[j] = (wl >= rw && wl < nw) ? g.ups_of(up[j], w[j]) : 0u;      // rows 1.. of the strip: connections to the row above
		cnt += __popc(b[j]);
	}
#pragma unroll
	for (uint32_t j = 1; j < 4; j++) upl[j] = w[j] ? up[j - 1] : 0u;      // same row: the word before is my own
	uint32_t v[1] = { cnt }, tot[1];
	block_excl_add<1>(v, tot, s_scan);
	const uint32_t nloc = tot[0];
	if (nloc > sa.cap) {      // uniform: the general pipeline takes over (host)
		if (t == 0) { hand_st<FUSED>(sa.strip_nruns + si, kStripOverflow); hand_st<FUSED>(sa.strip_nsc + si, 0u); atomicOr(sa.overflow, 1u); }
		return kStripOverflow;
	}
	stamp(0);
	{
		uint32_t local = v[0];
		uint32_t px = ((t * 4u) / rw) * g.sx;      // first pixel of my first word's row, relative to the strip
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			if (j && w[j] == 0) px += g.sx;
			if (wl < nw) {
				s_b[wl] = b[j];
				s_wb[wl] = static_cast<uint16_t>(local);
				for (uint32_t m = b[j]; m; m &= m - 1u) s_pool[local++] = static_cast<uint16_t>(px + w[j] * 32u + (__ffs(m) - 1u));
			}
		}
	}
	for (uint32_t j = t; j < nloc; j += kBlock) s_parent[j] = j;
	if (t < kStripBitmapWords) s_bm[t] = 0u;
	__syncthreads();
	stamp(1);
	// the crc weights of my runs are requested now and collected after the unions
	uint32_t gv[kStripRunsPerThread];
	{
		const uint32_t p0 = y0 * g.sx;
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			gv[i] = G[ablated(sa, 4u) ? j : n_pixels - (p0 + s_pool[j < nloc ? j : 0u])];
		}
	}
	// per-row and seam tables
	{
		uint32_t yy = (t * 4u) / rw;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			if (j && w[j] == 0) yy++;
			if (wl >= nw) break;
			const uint16_t wbv = s_wb[wl];
			if (!FUSED && w[j] == 0) sa.row_run[static_cast<uint64_t>(zi) * g.sy + y0 + yy] = wbv;      // (pins look pixels up: never on the fused path)
			if constexpr (FUSED) {
				if (wl < rw) hand_st<true>(sa.seam32_first + static_cast<uint64_t>(si) * rw + wl, static_cast<uint32_t>(wbv));
				if (wl + rw >= nw) hand_st<true>(sa.seam32_last + static_cast<uint64_t>(si) * rw + (wl + rw - nw), static_cast<uint32_t>(wbv));
			}
			else {
				if (wl < rw) sa.seam_first[static_cast<uint64_t>(si) * rw + wl] = wbv;
				if (wl + rw >= nw) sa.seam_last[static_cast<uint64_t>(si) * rw + (wl + rw - nw)] = wbv;
			}
		}
	}
	// FUSED: runs [0, n_top) are the first row's, [bot0, nloc) the last row's (read before the edge list takes the tables)
	const uint32_t n_top = FUSED ? (nw > rw ? s_wb[rw] : nloc) : 0u;
	const uint32_t bot0 = FUSED ? s_wb[nw - rw] : 0u;
	// ---- unions between vertically adjacent runs of the strip (first contact of each pair).
	// The contacts are spread unevenly over the words (0 .. 6 each): they are first written out as a
	// list of (run, run above) pairs, over the break words and the run pool, which nobody needs any
	// more, and then dealt out evenly, so that a wavefront does not wait for its busiest lane.
	{
		uint32_t c[4], b_up[4], base_up[4], base_here[4], n_e = 0;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			const uint32_t above = up[j] ? wl - rw : 0u;
			b_up[j] = s_b[above]; base_up[j] = s_wb[above]; base_here[j] = s_wb[wl < nw ? wl : 0u];
			c[j] = ablated(sa, 1u) ? 0u : up[j] & (~((up[j] << 1) | (upl[j] >> 31)) | b[j] | b_up[j]);
			n_e += __popc(c[j]);
		}
		uint32_t ve[1] = { n_e }, te[1];
		block_excl_add<1>(ve, te, s_scan);      // its barriers: every read of s_b / s_pool is done
		const uint32_t n_edges = te[0];
		if (n_edges <= kStripEdgeCap) {      // uniform
			uint32_t at = ve[0];
#pragma unroll
			for (uint32_t j = 0; j < 4; j++) {
				for (uint32_t cc = c[j]; cc; cc &= cc - 1u) {
					const uint32_t m = mask_le(__ffs(cc) - 1u);
					s_mem[at++] = (base_here[j] + __popc(b[j] & m) - 1u) | ((base_up[j] + __popc(b_up[j] & m) - 1u) << 16);
				}
			}
			__syncthreads();
			for (uint32_t e = t; e < n_edges; e += kBlock) { const uint32_t pr = s_mem[e]; sm_unite(s_parent, pr & 0xFFFFu, pr >> 16); }
		}
		else {
#pragma unroll
			for (uint32_t j = 0; j < 4; j++) {
				for (uint32_t cc = c[j]; cc; cc &= cc - 1u) {
					const uint32_t m = mask_le(__ffs(cc) - 1u);
					sm_unite(s_parent, base_here[j] + __popc(b[j] & m) - 1u, base_up[j] + __popc(b_up[j] & m) - 1u);
				}
			}
		}
	}
	__syncthreads();      // the pool now takes the strip components
	if constexpr (FUSED) {      // and the break words come back for the paint
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) { const uint32_t wl = t * 4u + j; if (wl < nw) s_b[wl] = b[j]; }
	}
	stamp(2);
	// ---- roots -> strip-local component ids in run order
	uint32_t root[kStripRunsPerThread];
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		const uint32_t j = t + i * kBlock;
		root[i] = j < nloc ? sm_find(s_parent, j) : 0u;
		if (j < nloc && root[i] == j) atomicOr(s_bm + (j >> 5), 1u << (j & 31u));
	}
	__syncthreads();
	if (t < kWave) {
		static_assert(kStripBitmapWords <= 2 * kWave, "two bitmap words per lane");
		const uint32_t c0 = t < kStripBitmapWords ? __popc(s_bm[t]) : 0u;
		const uint32_t c1 = t + kWave < kStripBitmapWords ? __popc(s_bm[t + kWave]) : 0u;
		const uint32_t i0 = wave_incl_add(c0);
		const uint32_t tot0 = __shfl(i0, kWave - 1, kWave);
		const uint32_t i1 = wave_incl_add(c1);
		if (t < kStripBitmapWords) s_bmbase[t] = i0 - c0;
		if (t + kWave < kStripBitmapWords) s_bmbase[t + kWave] = tot0 + i1 - c1;
		if (t == kWave - 1) s_misc[1] = tot0 + i1;
	}
	__syncthreads();
	const uint32_t nsc = s_misc[1];
	// one byte per run while the strip has at most 256 components (strip_lid): the paint kernel reads
	// them beside its stores, where every byte read costs several bytes' worth of store time
	uint16_t* lid_out = sa.run_lid + slot;
	const bool narrow = nsc <= 256u;
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		const uint32_t j = t + i * kBlock;
		if (j >= nloc) break;
		const uint32_t r = root[i];
		const uint32_t lid = s_bmbase[r >> 5] + __popc(s_bm[r >> 5] & ((1u << (r & 31u)) - 1u));
		s_pool[j] = static_cast<uint16_t>(lid);
		if constexpr (FUSED) {      // only the seam rows' runs are looked up by the resolver
			if (j < n_top || j >= bot0) hand_st<true>(sa.lid32 + slot + j, lid);
			continue;
		}
		if (narrow) reinterpret_cast<uint8_t*>(lid_out)[j] = static_cast<uint8_t>(lid);
		else lid_out[j] = static_cast<uint16_t>(lid);
	}
	for (uint32_t j = t; j < nsc; j += kBlock) s_parent[j] = 0u;      // every find is done: the table becomes the weights
	__syncthreads();
	stamp(3);
	// ---- crc weights: run j covering [a_j, a_j+1) adds G[n - a_j] ^ G[n - a_j+1] to its component
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		const uint32_t j = t + i * kBlock;
		if (j >= nloc || ablated(sa, 2u)) break;
		atomicXor(s_parent + s_pool[j], gv[i]);
		if (j) atomicXor(s_parent + s_pool[j - 1], gv[i]);
	}
	if (t == 0 && nloc) atomicXor(s_parent + s_pool[nloc - 1], G[n_pixels - y1 * g.sx]);
	__syncthreads();
	uint32_t* w_out = sa.sc_w + slot;
	for (uint32_t j = t; j < nsc; j += kBlock) hand_st<FUSED>(w_out + j, s_parent[j]);
	if (t == 0) { hand_st<FUSED>(sa.strip_nruns + si, nloc); hand_st<FUSED>(sa.strip_nsc + si, nsc); }
	stamp(4);
	if (nsc_ret) *nsc_ret = nsc;
	return nloc;
}

template <bool DIAG, bool RECORDS>
static __global__ void __launch_bounds__(kBlock, 7) k_strip_ccl(RunGeom g, StripArrays sa, RecordLists rl, const uint32_t* __restrict__ G, uint32_t n_pixels, unsigned long long* __restrict__ diag) {
	__shared__ __attribute__((aligned(16))) uint32_t s_lds[kStripCclWords];
	strip_ccl_body<DIAG, RECORDS>(g, sa, rl, G, n_pixels, diag, blockIdx.y + sa.zbase, blockIdx.x, s_lds);
}

// what k_slice_resolve needs besides the strips
struct ResolveArgs {
	uint32_t idbits, crc_fix, check_crc;
	const uint32_t* crc_expect;      // [nslices] raw
	const uint32_t* ncomp_expect;    // [nslices]
	const uint64_t* comp_off;        // [nslices] first entry of the slice in label_map
	const uint64_t* label_map;       // component -> label (pins: built from the component ids)
	// flat labels (labels.hpp:453-506): label = uniq[key[component]], read straight from the stream
	const uint8_t* keys;             // key of the first component of the decoded range
	const uint8_t* uniq;
	uint32_t key_width, stored_width, is_signed;
	uint64_t num_unique;
	uint32_t has_label;
	uint64_t label;
	uint32_t cap;                    // strip components the LDS table holds (<= kResolveCap)
	uint32_t* host_flags;            // mapped pinned host memory or null: [slices of the session] final error words, then the overflow word —
	                                 // the run's verdicts reach the host with the kernel that settles them, no launch of their own behind the paint
	uint32_t host_flags_n;           // slices of the session (index of the overflow word)
};

// little-endian integer of W bytes at any address (global memory takes unaligned accesses: one load, not W)
template <int W>
__device__ __forceinline__ uint64_t ld_le(const uint8_t* p) {
	if constexpr (W == 1) return *p;
	else if constexpr (W == 2) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
	else if constexpr (W == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
	else { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
}
// f(integral_constant<int, width>) for the widths the format stores (1, 2, 4, 8: lib.hpp compute_byte_width);
// the switch sits OUTSIDE the loops over entries, so that the loads of all entries are issued side by side — a
// byte loop with a run-time bound made every byte a trip to memory of its own (k_slice_resolve: 28 of its 48 us)
template <typename F>
__device__ __forceinline__ void with_width(uint32_t w, F&& f) {
	switch (w) {
		case 1: f(std::integral_constant<int, 1>{}); break;
		case 2: f(std::integral_constant<int, 2>{}); break;
		case 4: f(std::integral_constant<int, 4>{}); break;
		default: f(std::integral_constant<int, 8>{}); break;
	}
}

// grid = slices of the launch, block = kResolveBlock
// LABELS: flat labels (label_map is ready): the label of every strip component is written here.
// Otherwise its component id goes to sc_cc (pins: the label table needs the ids first).
// Registers: nothing per strip component survives a barrier — roots keep their index in the low 16 bits
// of their table entry and get their rank in the high 16 (kResolveCap <= 65535) — so that two
// workgroups of 1024 share a CU (<= 64 registers) and all slices of a 512-slice volume run at once.
// The body works for a workgroup of BLOCK threads on slice zi with its tables in the caller's LDS (s_tab: ra.cap
// entries, s_scbase: nstrips + 1, s_scan: BLOCK / 64, s_flag: 1) — k_slice_resolve (1024 threads, a launch of its
// own) and the last strip workgroup of a slice in k_strip_fused (256 threads, HANDOFF: what the strips of the slice
// left is read with hand_ld and the labels are written with hand_st).  Returns false when the slice does not fit.
template <typename OUT, bool LABELS, bool DIAG, int BLOCK, bool HANDOFF>
__device__ __forceinline__ bool slice_resolve_body(
	const RunGeom& g, const StripArrays& sa, const ResolveArgs& ra, uint32_t* __restrict__ ncomp_out, unsigned long long* __restrict__ diag,
	uint32_t zi, uint32_t* s_tab, uint32_t* s_scbase, uint32_t* s_scan, uint32_t* s_flag_p
) {
	constexpr int kResolveBlock = BLOCK;      // (shadows the launch constant: the body is written in its terms)
	uint32_t& s_flag = *s_flag_p;
	static_assert(kResolveCap <= 0xFFFFu, "index and rank share a table entry");
	constexpr int NW = kResolveBlock / kWave;
	unsigned long long d_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) {
		if (DIAG && threadIdx.x == 0) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			atomicAdd(diag + slot, now - d_t);
			d_t = now;
		}
	};
	const uint32_t t = threadIdx.x;
	const uint32_t ns = sa.nstrips;
	const uint32_t rw = g.row_words;
	const uint32_t si0 = zi * ns;
	// ---- the seam words of this thread are requested first: they depend on nothing
	const uint32_t items = (ns - 1u) * rw;
	constexpr uint32_t kSeamPer = 2;      // seam words per thread and round
	uint32_t q_up[kSeamPer], q_prev[kSeamPer], q_bh[kSeamPer], q_bu[kSeamPer], q_wh[kSeamPer], q_wu[kSeamPer];
	auto seam_load = [&](uint32_t it0) {
		// all loads issued, none in a branch (seam words past the end read the first seam instead)
#pragma unroll
		for (uint32_t u = 0; u < kSeamPer; u++) {
			const uint32_t it_raw = it0 + u * kResolveBlock + t;
			const uint32_t it = it_raw < items ? it_raw : 0u;
			const uint32_t seam = it / rw, w = it - seam * rw;
			const uint32_t k = seam + 1u, y = k * sa.strip_rows;
			const uint64_t at = zi * g.plane_words + static_cast<uint64_t>(y) * rw + w;
			const uint32_t h = hand_ld<HANDOFF>(g.planeH + at), hp = hand_ld<HANDOFF>(g.planeH + (w ? at - 1u : at));
			const uint32_t vh = hand_ld<HANDOFF>(g.planeV + at), vu = hand_ld<HANDOFF>(g.planeV + (at - rw));
			if constexpr (HANDOFF) {
				q_wh[u] = hand_ld<true>(sa.seam32_first + static_cast<uint64_t>(si0 + k) * rw + w);
				q_wu[u] = hand_ld<true>(sa.seam32_last + static_cast<uint64_t>(si0 + k - 1u) * rw + w);
			}
			else {
				q_wh[u] = sa.seam_first[static_cast<uint64_t>(si0 + k) * rw + w];
				q_wu[u] = sa.seam_last[static_cast<uint64_t>(si0 + k - 1u) * rw + w];
			}
			q_up[u] = it_raw < items ? g.ups_of(h, w) : 0u;
			q_prev[u] = w ? (g.ups_of(hp, w - 1u) >> 31) : 0u;
			q_bh[u] = g.breaks_of(vh, w); q_bu[u] = g.breaks_of(vu, w);
		}
	};
	if (items) seam_load(0);
	// ---- strip tables
	uint32_t my_nsc = 0;
	if (t == 0) s_flag = 0;
	__syncthreads();
	if (t < ns) {      // ns <= BLOCK (the host checks it for either caller): a strip per thread
		const uint32_t nr = hand_ld<HANDOFF>(sa.strip_nruns + si0 + t);
		if (nr == kStripOverflow) s_flag = 1;
		my_nsc = hand_ld<HANDOFF>(sa.strip_nsc + si0 + t);
	}
	uint32_t v[1] = { my_nsc }, tot[1];
	block_excl_add<1, NW>(v, tot, s_scan);
	if (t < ns) s_scbase[t] = v[0];
	if (t == 0) s_scbase[ns] = tot[0];
	const uint32_t total = tot[0];
	__syncthreads();
	if (s_flag || total > ra.cap) {      // uniform
		if (t == 0) {
			// bit 0: a strip had more runs than its tables (the strip kernel said so already); bit 1: this table is too small
			atomicOr(sa.overflow, s_flag ? 1u : 2u);
			if (ra.host_flags) { ra.host_flags[ra.host_flags_n + (s_flag ? 0u : 2u)] = 1u; ra.host_flags[zi] = sa.slice_err[zi]; }
		}
		return false;
	}
	for (uint32_t i = t; i < total; i += kResolveBlock) s_tab[i] = i;
	__syncthreads();
	stamp(0);
	// ---- unions across the strip seams
	for (uint32_t it0 = 0; it0 < items; it0 += kSeamPer * kResolveBlock) {
		if (it0) seam_load(it0);
#pragma unroll
		for (uint32_t u = 0; u < kSeamPer; u++) {
			if (!q_up[u]) continue;
			const uint32_t it = it0 + u * kResolveBlock + t;
			const uint32_t k = it / rw + 1u;
			const uint16_t* lid_h = sa.run_lid + static_cast<uint64_t>(si0 + k) * sa.cap;
			const uint16_t* lid_u = sa.run_lid + static_cast<uint64_t>(si0 + k - 1u) * sa.cap;
			const uint32_t nsc_h = s_scbase[k + 1u] - s_scbase[k], nsc_u = s_scbase[k] - s_scbase[k - 1u];
			for (uint32_t c = q_up[u] & (~((q_up[u] << 1) | q_prev[u]) | q_bh[u] | q_bu[u]); c; c &= c - 1u) {
				const uint32_t m = mask_le(__ffs(c) - 1u);
				const uint32_t jh = q_wh[u] + __popc(q_bh[u] & m) - 1u, ju = q_wu[u] + __popc(q_bu[u] & m) - 1u;
				if (jh >= sa.cap || ju >= sa.cap) continue;
				uint32_t lh, lu;
				if constexpr (HANDOFF) {
					lh = hand_ld<true>(sa.lid32 + static_cast<uint64_t>(si0 + k) * sa.cap + jh);
					lu = hand_ld<true>(sa.lid32 + static_cast<uint64_t>(si0 + k - 1u) * sa.cap + ju);
				}
				else { lh = strip_lid(lid_h, nsc_h, jh); lu = strip_lid(lid_u, nsc_u, ju); }
				if (lh < nsc_h && lu < nsc_u) sm_unite(s_tab, s_scbase[k] + lh, s_scbase[k - 1u] + lu);
			}
		}
	}
	__syncthreads();
	stamp(1);
	// ---- every entry -> its root (a racing find only ever meets ancestors); the roots ranked in index order =
	// raster order of the components' first pixels
	const uint32_t per = (total + kResolveBlock - 1) / kResolveBlock;      // <= kResolvePer
	const uint32_t i0 = min(total, t * per), i1 = min(total, i0 + per);
	for (uint32_t i = i0; i < i1; i++) sm_store(s_tab, i, sm_find(s_tab, i));
	__syncthreads();
	uint32_t nroot = 0;
	for (uint32_t i = i0; i < i1; i++) nroot += s_tab[i] == i ? 1u : 0u;
	// the crc weights of the first entries are requested before the scan's barriers
	constexpr uint32_t kPer = 4;
	uint32_t strip0 = 0;
	if (i0 < i1) {      // strip of my first entry
		uint32_t lo = 0, hi = ns;
		while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_scbase[mid] <= i0) lo = mid; else hi = mid; }
		strip0 = lo;
	}
	uint32_t v2[1] = { nroot }, tot2[1];
	block_excl_add<1, NW>(v2, tot2, s_scan);
	{
		uint32_t rk = v2[0];
		for (uint32_t i = i0; i < i1; i++) if (s_tab[i] == i) s_tab[i] = i | (rk++ << 16);
	}
	__syncthreads();
	stamp(2);
	const uint32_t ncomp = tot2[0];
	const uint32_t nexp = ra.ncomp_expect[zi];
	const uint64_t coff = ra.comp_off[zi];
	// ---- ids, labels, crc32c of the component image: four entries per round, their loads side by side
	uint32_t part = 0;
	uint32_t sidx = strip0;
	for (uint32_t r0 = i0; r0 < i1; r0 += kPer) {
		uint32_t cc[kPer], wgt[kPer];
		uint64_t gi[kPer], key[kPer];
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t i = r0 + q;
			const bool on = i < i1;
			const uint32_t ii = on ? i : i0;
			if (on) while (sidx + 1 < ns && s_scbase[sidx + 1] <= ii) sidx++;
			const uint32_t sq = on ? sidx : strip0;
			gi[q] = static_cast<uint64_t>(si0 + sq) * sa.cap + (ii - s_scbase[sq]);
			cc[q] = s_tab[s_tab[ii] & 0xFFFFu] >> 16;
			wgt[q] = hand_ld<HANDOFF>(sa.sc_w + gi[q]);
			key[q] = 0;
		}
		uint64_t lab[kPer];
		if (LABELS) {
			// keys of the four entries in one trip to memory, their labels in a second one (entries past the
			// end and ids outside the label section read entry 0 and are set right afterwards)
			with_width(ra.key_width, [&](auto W) {
#pragma unroll
				for (uint32_t q = 0; q < kPer; q++) key[q] = ld_le<decltype(W)::value>(ra.keys + (coff + (cc[q] < nexp ? cc[q] : 0u)) * static_cast<uint64_t>(decltype(W)::value));
			});
#pragma unroll
			for (uint32_t q = 0; q < kPer; q++) lab[q] = 0;
			if (ra.num_unique) {
				with_width(ra.stored_width, [&](auto W) {
#pragma unroll
					for (uint32_t q = 0; q < kPer; q++) lab[q] = ld_le<decltype(W)::value>(ra.uniq + (key[q] < ra.num_unique ? key[q] : 0ull) * static_cast<uint64_t>(decltype(W)::value));
				});
			}
		}
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			if (r0 + q >= i1) continue;
			if (LABELS) {
				uint64_t val = 0;
				if (cc[q] < nexp && key[q] < ra.num_unique) {
					val = lab[q];
					if (ra.is_signed && ra.stored_width < 8u && (val >> (8u * ra.stored_width - 1u))) val |= ~0ull << (8u * ra.stored_width);
				}
				if (ra.has_label) val = (val == ra.label);
				if constexpr (HANDOFF) {      // (narrow labels travel as 32-bit words inside the launch)
					typedef typename std::conditional<sizeof(OUT) < 4, uint32_t, OUT>::type LAB;
					hand_st<true>(static_cast<LAB*>(sa.sc_label) + gi[q], static_cast<LAB>(static_cast<OUT>(val)));
				}
				else static_cast<OUT*>(sa.sc_label)[gi[q]] = static_cast<OUT>(val);
			}
			else sa.sc_cc[gi[q]] = cc[q];
			// sum over set bits j < idbits of the id:  wgt * x^(idbits-1-j)
			uint32_t wg = wgt[q];
			for (int j = static_cast<int>(ra.idbits) - 1; j >= 0; j--) {
				part ^= ((cc[q] >> j) & 1u) ? wg : 0u;
				wg = (wg >> 1) ^ ((wg & 1u) ? kCrcPoly : 0u);
			}
		}
	}
	part = wave_xor(part);
	if ((t & (kWave - 1)) == 0) s_scan[t >> 6] = part;
	__syncthreads();
	if (t == 0) {
		uint32_t x = 0;
		for (int wv = 0; wv < NW; wv++) x ^= s_scan[wv];
		uint32_t e = 0;
		if (ncomp != nexp) e |= ERR_NCOMP;
		else if (ra.check_crc && gf_mul(x, ra.crc_fix) != ra.crc_expect[zi]) e |= ERR_CRC;
		if (e) atomicOr(sa.slice_err + zi, e);
		ncomp_out[zi] = ncomp;
		if (ra.host_flags) ra.host_flags[zi] = sa.slice_err[zi] | e;      // (the bits of the kernels in front crossed a kernel boundary)
	}
	stamp(3);
	return true;
}

template <typename OUT, bool LABELS, bool DIAG>
static __global__ void __launch_bounds__(kResolveBlock, 8) k_slice_resolve(RunGeom g, StripArrays sa, ResolveArgs ra, uint32_t* __restrict__ ncomp_out, unsigned long long* __restrict__ diag) {
	extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];      // ra.cap entries
	__shared__ uint32_t s_scbase[kMaxStrips + 1];
	__shared__ uint32_t s_scan[kResolveBlock / kWave];
	__shared__ uint32_t s_flag;
	slice_resolve_body<OUT, LABELS, DIAG, kResolveBlock, false>(g, sa, ra, ncomp_out, diag, blockIdx.x + sa.zbase, s_tab, s_scbase, s_scan, &s_flag);
}

// pins: labels of the strip components once label_map has been filled from their component ids
// grid = (nstrips, slices of the launch), block = kBlock
template <typename OUT>
static __global__ void __launch_bounds__(kBlock) k_strip_labels(StripArrays sa, ResolveArgs ra) {
	const uint32_t zi = blockIdx.y + sa.zbase;
	const uint32_t si = zi * sa.nstrips + blockIdx.x;
	if (sa.strip_nruns[si] == kStripOverflow) return;
	const uint32_t n = min(sa.strip_nsc[si], sa.cap);
	const uint32_t nexp = ra.ncomp_expect[zi];
	const uint64_t coff = ra.comp_off[zi];
	const uint64_t slot = static_cast<uint64_t>(si) * sa.cap;
	for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
		const uint32_t c = sa.sc_cc[slot + j];
		uint64_t val = c < nexp ? ra.label_map[coff + c] : 0ull;
		if (ra.has_label) val = (val == ra.label);
		static_cast<OUT*>(sa.sc_label)[slot + j] = static_cast<OUT>(val);
	}
}

// component id of the run that holds pixel (x, y) — pin decoding (labels.hpp:600-614)
__device__ __forceinline__ uint32_t strip_component_of_pixel(const RunGeom& g, const StripArrays& sa, uint32_t zi, uint32_t x, uint32_t y) {
	const uint32_t k = y / sa.strip_rows;
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t nr = sa.strip_nruns[si];
	if (nr == kStripOverflow) return 0xFFFFFFFFu;
	uint32_t run = sa.row_run[static_cast<uint64_t>(zi) * g.sy + y];
	const uint32_t wx = x >> 5;
	for (uint32_t w = 0; w < wx; w++) run += __popc(g.breaks(zi, y, w));
	run += __popc(g.breaks(zi, y, wx) & mask_le(x & 31u)) - 1u;
	if (run >= nr || run >= sa.cap) return 0xFFFFFFFFu;
	const uint64_t slot = static_cast<uint64_t>(si) * sa.cap;
	const uint32_t lid = strip_lid(sa.run_lid + slot, sa.strip_nsc[si], run);
	return lid < sa.cap ? sa.sc_cc[slot + lid] : 0xFFFFFFFFu;
}

}  // namespace dev
}  // namespace ckl
