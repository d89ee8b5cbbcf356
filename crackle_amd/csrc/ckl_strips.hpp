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
//                    and crc32c, maps ids to labels and writes the label of every run.
//   k_paint_strips   4096-pixel tiles of whole rows: the tile's plane words and their run prefix
//                    are built in LDS (no per-word table in HBM), the run labels are staged, 16-byte
//                    streaming stores.
//
// A strip with more runs than the LDS tables hold, or a slice with more strip components than
// k_slice_resolve's table, raises `overflow`: the host then runs the general run pipeline of
// ckl_runs.hpp on the same planes (dense / noisy volumes).
#pragma once

#include "ckl_runs.hpp"

namespace ckl {
namespace dev {

constexpr uint32_t kStripWords = 1024;      // plane words per strip: 4 per thread
constexpr uint32_t kStripCap = 3072;        // runs per strip held in LDS
constexpr uint32_t kStripRunsPerThread = kStripCap / kBlock;
constexpr uint32_t kStripBitmapWords = kStripCap / 32;
constexpr uint32_t kStripOverflow = 0xFFFFFFFFu;
constexpr int kResolveBlock = 1024;
constexpr uint32_t kResolveCap = 12288;     // strip components per slice held in LDS
constexpr uint32_t kResolvePer = kResolveCap / kResolveBlock;
constexpr uint32_t kMaxStrips = 1024;

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

struct StripArrays {
	const uint64_t* rbase;     // [nslices] base of the slice in the per-run arrays
	const uint32_t* rcap;      // [nslices] capacity
	uint32_t* cursor;          // [nslices] runs handed out to strips so far (zeroed per decode)
	uint16_t* run_lid;         // [runs] strip-local component of the run
	uint32_t* row_run;         // [nslices][sy] slice-relative index of the first run of each row
	uint32_t* strip_base;      // [nslices][nstrips] slice-relative index of the strip's first run
	uint32_t* strip_nruns;     // [nslices][nstrips] (kStripOverflow: the strip did not fit)
	uint32_t* strip_nsc;       // [nslices][nstrips] strip components
	uint16_t* seam_first;      // [nslices][nstrips][row_words] runs of the strip before each word of its first row
	uint16_t* seam_last;       // ... of its last row
	uint32_t* sc_w;            // [runs] per strip component (at strip_base + local id): XOR of crc weights
	uint32_t* sc_cc;           // [runs] per strip component: component id of the slice
	uint32_t* slice_err;
	uint32_t* overflow;        // one word
	uint32_t nstrips, strip_rows;
	uint32_t zbase;            // first slice of this launch (z-chunked launches)
};

// grid = (nstrips, slices of the launch), block = kBlock
static __global__ void __launch_bounds__(kBlock) k_strip_ccl(RunGeom g, StripArrays sa, const uint32_t* __restrict__ G, uint32_t n_pixels) {
	__shared__ uint32_t s_parent[kStripCap];          // union-find, then the crc weights per strip component
	__shared__ uint32_t s_b[kStripWords];             // break words of the strip
	__shared__ uint16_t s_wb[kStripWords];            // runs before each word
	__shared__ uint16_t s_start[kStripCap];           // first pixel of each run, relative to the strip
	__shared__ uint16_t s_lid[kStripCap];
	__shared__ uint32_t s_bm[kStripBitmapWords], s_bmbase[kStripBitmapWords];
	__shared__ uint32_t s_scan[kWaves];
	__shared__ uint32_t s_misc[2];
	const uint32_t zi = blockIdx.y + sa.zbase;
	const uint32_t k = blockIdx.x;
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t rw = g.row_words;
	const uint32_t nw = (y1 - y0) * rw;
	const uint32_t t = threadIdx.x;
	const uint64_t rb = sa.rbase[zi];

	// ---- runs of the strip
	uint32_t b[4], yl[4], w[4], cnt = 0;
	{
		uint32_t yy = (t * 4u) / rw, ww = t * 4u - yy * rw;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			yl[j] = yy; w[j] = ww;
			b[j] = wl < nw ? g.breaks(zi, y0 + yy, ww) : 0u;
			cnt += __popc(b[j]);
			if (++ww == rw) { ww = 0; yy++; }
		}
	}
	// the H words of rows 1.. are requested before the scan's barriers
	uint32_t up[4], upl[4];
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t wl = t * 4u + j;
		const bool in = wl >= rw && wl < nw;
		up[j] = in ? g.ups(zi, y0 + yl[j], w[j]) : 0u;
		upl[j] = (in && w[j]) ? (j ? 0u : g.ups(zi, y0 + yl[j], w[j] - 1u)) : 0u;
	}
#pragma unroll
	for (uint32_t j = 1; j < 4; j++) if (w[j]) upl[j] = up[j - 1];      // same row: the word before is my own
	uint32_t v[1] = { cnt }, tot[1];
	block_excl_add<1>(v, tot, s_scan);
	const uint32_t nloc = tot[0];
	if (nloc > kStripCap) {      // uniform: the general pipeline takes over (host)
		if (t == 0) { sa.strip_nruns[si] = kStripOverflow; sa.strip_nsc[si] = 0; sa.strip_base[si] = 0; atomicOr(sa.overflow, 1u); }
		return;
	}
	if (t == 0) {
		uint32_t base = atomicAdd(sa.cursor + zi, nloc);
		if (base + nloc > sa.rcap[zi]) { atomicOr(sa.slice_err + zi, ERR_CAPACITY); base = kStripOverflow; }
		s_misc[0] = base;
	}
	{
		uint32_t local = v[0];
		const uint32_t p_strip = 0;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			if (wl < nw) {
				s_b[wl] = b[j];
				s_wb[wl] = static_cast<uint16_t>(local);
				const uint32_t px = p_strip + yl[j] * g.sx + w[j] * 32u;
				for (uint32_t m = b[j]; m; m &= m - 1u) s_start[local++] = static_cast<uint16_t>(px + (__ffs(m) - 1u));
			}
		}
	}
	for (uint32_t j = t; j < nloc; j += kBlock) s_parent[j] = j;
	for (uint32_t j = t; j < kStripBitmapWords; j += kBlock) s_bm[j] = 0u;
	__syncthreads();
	const uint32_t base = s_misc[0];
	if (base == kStripOverflow) {      // capacity of the slice exceeded (malformed stream): flagged, nothing written
		if (t == 0) { sa.strip_nruns[si] = 0; sa.strip_nsc[si] = 0; sa.strip_base[si] = 0; }
		return;
	}
	// per-row and seam tables
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t wl = t * 4u + j;
		if (wl >= nw) break;
		const uint32_t wbv = s_wb[wl];
		if (w[j] == 0) sa.row_run[static_cast<uint64_t>(zi) * g.sy + y0 + yl[j]] = base + wbv;
		if (wl < rw) sa.seam_first[static_cast<uint64_t>(si) * rw + wl] = static_cast<uint16_t>(wbv);
		if (wl + rw >= nw) sa.seam_last[static_cast<uint64_t>(si) * rw + (wl + rw - nw)] = static_cast<uint16_t>(wbv);
	}
	// ---- unions between vertically adjacent runs of the strip (first contact of each pair)
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t wl = t * 4u + j;
		if (!up[j]) continue;
		const uint32_t b_here = b[j], b_up = s_b[wl - rw];
		const uint32_t base_here = s_wb[wl], base_up = s_wb[wl - rw];
		for (uint32_t c = up[j] & (~((up[j] << 1) | (upl[j] >> 31)) | b_here | b_up); c; c &= c - 1u) {
			const uint32_t m = mask_le(__ffs(c) - 1u);
			sm_unite(s_parent, base_here + __popc(b_here & m) - 1u, base_up + __popc(b_up & m) - 1u);
		}
	}
	__syncthreads();
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
		static_assert(kStripBitmapWords <= 2 * kWave, "two words per lane");
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
	uint16_t* lid_out = sa.run_lid + rb + base;
#pragma unroll
	for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
		const uint32_t j = t + i * kBlock;
		if (j >= nloc) break;
		const uint32_t r = root[i];
		const uint32_t lid = s_bmbase[r >> 5] + __popc(s_bm[r >> 5] & ((1u << (r & 31u)) - 1u));
		s_lid[j] = static_cast<uint16_t>(lid);
		lid_out[j] = static_cast<uint16_t>(lid);
	}
	for (uint32_t j = t; j < nsc; j += kBlock) s_parent[j] = 0u;      // every find is done: the table becomes the weights
	__syncthreads();
	// ---- crc weights: run j covering [a_j, a_j+1) adds G[n - a_j] ^ G[n - a_j+1] to its component
	{
		const uint32_t p0 = y0 * g.sx;
		uint32_t gv[kStripRunsPerThread];
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			gv[i] = j < nloc ? G[n_pixels - (p0 + s_start[j])] : 0u;
		}
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			if (j >= nloc) break;
			atomicXor(s_parent + s_lid[j], gv[i]);
			if (j) atomicXor(s_parent + s_lid[j - 1], gv[i]);
		}
		if (t == 0 && nloc) atomicXor(s_parent + s_lid[nloc - 1], G[n_pixels - y1 * g.sx]);
	}
	__syncthreads();
	uint32_t* w_out = sa.sc_w + rb + base;
	for (uint32_t j = t; j < nsc; j += kBlock) w_out[j] = s_parent[j];
	if (t == 0) { sa.strip_base[si] = base; sa.strip_nruns[si] = nloc; sa.strip_nsc[si] = nsc; }
}

// The dynamic LDS of the resolve kernels seen as ids (uint32) and as labels (OUT): one extern
// array per element type, all at the same address (a pointer cast from the uint32 view loses the
// LDS address space in hipcc 7.2 and ends in an illegal instruction).
template <typename T> struct DynLds;
template <> struct DynLds<uint8_t> { static __device__ __forceinline__ uint8_t* get() { extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn_u8[]; return s_dyn_u8; } };
template <> struct DynLds<uint16_t> { static __device__ __forceinline__ uint16_t* get() { extern __shared__ __attribute__((aligned(16))) uint16_t s_dyn_u16[]; return s_dyn_u16; } };
template <> struct DynLds<uint32_t> { static __device__ __forceinline__ uint32_t* get() { extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn_u32[]; return s_dyn_u32; } };
template <> struct DynLds<uint64_t> { static __device__ __forceinline__ uint64_t* get() { extern __shared__ __attribute__((aligned(16))) uint64_t s_dyn_u64[]; return s_dyn_u64; } };

// what k_slice_resolve needs besides the strips
struct ResolveArgs {
	uint32_t idbits, crc_fix, check_crc;
	const uint32_t* crc_expect;      // [nslices] raw
	const uint32_t* ncomp_expect;    // [nslices]
	const uint64_t* comp_off;        // [nslices] first entry of the slice in label_map
	const uint64_t* label_map;       // component -> label (LABELS)
	uint32_t has_label;
	uint64_t label;
	void* run_label;                 // [runs] typed like the output (LABELS)
	uint32_t cap;                    // strip components the dynamic LDS holds
};

// component -> label -> every run of the slice (shared by the flat path, inside k_slice_resolve,
// and the pin path, where the label table only exists after the component ids)
template <typename OUT>
__device__ __forceinline__ void strip_run_labels(const StripArrays& sa, uint64_t rb, const uint32_t* s_sbase, const uint32_t* s_nruns, const uint32_t* s_scbase, const OUT* s_lab, OUT* __restrict__ run_label) {
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
	for (uint32_t s = wave; s < sa.nstrips; s += kResolveBlock / kWave) {
		const uint32_t nr = s_nruns[s];
		const uint16_t* lid = sa.run_lid + rb + s_sbase[s];
		OUT* dst = run_label + rb + s_sbase[s];
		const uint32_t scb = s_scbase[s];
		for (uint32_t j0 = 0; j0 < nr; j0 += 4 * kWave) {
			uint32_t l[4];
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) { const uint32_t j = j0 + u * kWave + lane; l[u] = j < nr ? lid[j] : 0u; }
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) { const uint32_t j = j0 + u * kWave + lane; if (j < nr) dst[j] = s_lab[scb + l[u]]; }
		}
	}
}

// grid = slices of the launch, block = kResolveBlock, dynamic LDS = cap * max(4, sizeof(OUT))
// LABELS: flat labels (label_map is ready): run labels are written here.  Otherwise the component
// id of every strip component goes to sc_cc (pins: the label table needs them first).
template <typename OUT, bool LABELS>
static __global__ void __launch_bounds__(kResolveBlock) k_slice_resolve(RunGeom g, StripArrays sa, ResolveArgs ra, uint32_t* __restrict__ ncomp_out) {
	uint32_t* s_tab = DynLds<uint32_t>::get();
	__shared__ uint32_t s_sbase[kMaxStrips], s_nruns[kMaxStrips], s_scbase[kMaxStrips + 1];
	__shared__ uint32_t s_scan[kResolveBlock / kWave];
	__shared__ uint32_t s_flag;
	constexpr int NW = kResolveBlock / kWave;
	const uint32_t zi = blockIdx.x + sa.zbase;
	const uint32_t t = threadIdx.x;
	const uint32_t ns = sa.nstrips;
	const uint32_t rw = g.row_words;
	const uint64_t rb = sa.rbase[zi];
	// ---- strip tables
	uint32_t my_nsc = 0;
	if (t == 0) s_flag = 0;
	__syncthreads();
	if (t < ns) {
		const uint32_t si = zi * ns + t;
		const uint32_t nr = sa.strip_nruns[si];
		if (nr == kStripOverflow) s_flag = 1;
		s_nruns[t] = nr == kStripOverflow ? 0u : nr;
		s_sbase[t] = sa.strip_base[si];
		my_nsc = sa.strip_nsc[si];
	}
	uint32_t v[1] = { my_nsc }, tot[1];
	block_excl_add<1, NW>(v, tot, s_scan);
	if (t < ns) s_scbase[t] = v[0];
	if (t == 0) s_scbase[ns] = tot[0];
	const uint32_t total = tot[0];
	__syncthreads();
	if (s_flag || total > ra.cap) {      // uniform
		if (t == 0) atomicOr(sa.overflow, 1u);
		return;
	}
	const uint32_t per = (total + kResolveBlock - 1) / kResolveBlock;      // <= kResolvePer
	const uint32_t i0 = min(total, t * per), i1 = min(total, i0 + per);
	for (uint32_t i = t; i < total; i += kResolveBlock) s_tab[i] = i;
	__syncthreads();
	// ---- unions across the strip seams
	const uint32_t items = (ns - 1u) * rw;
	for (uint32_t it = t; it < items; it += kResolveBlock) {
		const uint32_t seam = it / rw, w = it - seam * rw;
		const uint32_t k = seam + 1u, y = k * sa.strip_rows;
		const uint32_t up = g.ups(zi, y, w);
		if (!up) continue;
		const uint32_t prev_bit = w ? (g.ups(zi, y, w - 1u) >> 31) : 0u;
		const uint32_t b_here = g.breaks(zi, y, w), b_up = g.breaks(zi, y - 1u, w);
		const uint32_t wbh = sa.seam_first[(static_cast<uint64_t>(zi) * ns + k) * rw + w];
		const uint32_t wbu = sa.seam_last[(static_cast<uint64_t>(zi) * ns + k - 1u) * rw + w];
		const uint16_t* lid_h = sa.run_lid + rb + s_sbase[k];
		const uint16_t* lid_u = sa.run_lid + rb + s_sbase[k - 1u];
		for (uint32_t c = up & (~((up << 1) | prev_bit) | b_here | b_up); c; c &= c - 1u) {
			const uint32_t m = mask_le(__ffs(c) - 1u);
			const uint32_t jh = wbh + __popc(b_here & m) - 1u, ju = wbu + __popc(b_up & m) - 1u;
			if (jh >= s_nruns[k] || ju >= s_nruns[k - 1u]) continue;      // a strip that hit the slice capacity (flagged there)
			const uint32_t a = s_scbase[k] + lid_h[jh], bb = s_scbase[k - 1u] + lid_u[ju];
			if (a < total && bb < total) sm_unite(s_tab, a, bb);
		}
	}
	__syncthreads();
	// ---- roots ranked in index order = raster order of the components' first pixels
	uint32_t root[kResolvePer], nroot = 0;
#pragma unroll
	for (uint32_t q = 0; q < kResolvePer; q++) {
		const uint32_t i = i0 + q;
		root[q] = i < i1 ? sm_find(s_tab, i) : 0u;
		nroot += (i < i1 && root[q] == i) ? 1u : 0u;
	}
	uint32_t v2[1] = { nroot }, tot2[1];
	block_excl_add<1, NW>(v2, tot2, s_scan);      // its barriers also end every find
	{
		uint32_t rk = v2[0];
#pragma unroll
		for (uint32_t q = 0; q < kResolvePer; q++) {
			const uint32_t i = i0 + q;
			if (i < i1 && root[q] == i) s_tab[i] = rk++;
		}
	}
	__syncthreads();
	const uint32_t ncomp = tot2[0];
	const uint32_t nexp = ra.ncomp_expect[zi];
	// ---- ids, crc32c of the component image, labels
	uint32_t cc[kResolvePer], part = 0;
	{
		uint32_t s = 0;
		if (i0 < i1) {      // strip of my first entry
			uint32_t lo = 0, hi = ns;
			while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_scbase[mid] <= i0) lo = mid; else hi = mid; }
			s = lo;
		}
#pragma unroll
		for (uint32_t q = 0; q < kResolvePer; q++) {
			const uint32_t i = i0 + q;
			cc[q] = 0;
			if (i >= i1) continue;
			cc[q] = s_tab[root[q]];
			while (s + 1 < ns && s_scbase[s + 1] <= i) s++;
			const uint64_t gi = rb + s_sbase[s] + (i - s_scbase[s]);
			uint32_t wgt = sa.sc_w[gi];
			if (!LABELS) sa.sc_cc[gi] = cc[q];
			// sum over set bits j < idbits of the id:  wgt * x^(idbits-1-j)
			for (int j = static_cast<int>(ra.idbits) - 1; j >= 0; j--) {
				part ^= ((cc[q] >> j) & 1u) ? wgt : 0u;
				wgt = (wgt >> 1) ^ ((wgt & 1u) ? kCrcPoly : 0u);
			}
		}
	}
	// block xor (NW wavefronts)
	part = wave_xor(part);
	__syncthreads();      // every s_tab[root] has been read
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
	}
	if (!LABELS) return;
	OUT* s_lab = DynLds<OUT>::get();
	OUT mine[kResolvePer];
#pragma unroll
	for (uint32_t q = 0; q < kResolvePer; q++) {
		uint64_t val = 0;
		if (i0 + q < i1 && cc[q] < nexp) val = ra.label_map[ra.comp_off[zi] + cc[q]];
		if (ra.has_label) val = (val == ra.label);
		mine[q] = static_cast<OUT>(val);
	}
	if (sizeof(OUT) > 4) __syncthreads();      // wider than the ids they replace: all ids are in registers by now (barrier above), kept for symmetry
#pragma unroll
	for (uint32_t q = 0; q < kResolvePer; q++) if (i0 + q < i1) s_lab[i0 + q] = mine[q];
	__syncthreads();
	strip_run_labels<OUT>(sa, rb, s_sbase, s_nruns, s_scbase, s_lab, static_cast<OUT*>(ra.run_label));
}

// pins: labels of the runs once label_map has been filled from the component ids
// grid = slices of the launch, block = kResolveBlock, dynamic LDS = cap * sizeof(OUT)
template <typename OUT>
static __global__ void __launch_bounds__(kResolveBlock) k_strip_labels(StripArrays sa, ResolveArgs ra) {
	__shared__ uint32_t s_sbase[kMaxStrips], s_nruns[kMaxStrips], s_scbase[kMaxStrips + 1];
	__shared__ uint32_t s_scan[kResolveBlock / kWave];
	constexpr int NW = kResolveBlock / kWave;
	const uint32_t zi = blockIdx.x + sa.zbase;
	const uint32_t t = threadIdx.x;
	const uint32_t ns = sa.nstrips;
	const uint64_t rb = sa.rbase[zi];
	uint32_t my_nsc = 0;
	if (t < ns) {
		const uint32_t si = zi * ns + t;
		const uint32_t nr = sa.strip_nruns[si];
		s_nruns[t] = nr == kStripOverflow ? 0u : nr;
		s_sbase[t] = sa.strip_base[si];
		my_nsc = nr == kStripOverflow ? 0u : sa.strip_nsc[si];
	}
	uint32_t v[1] = { my_nsc }, tot[1];
	block_excl_add<1, NW>(v, tot, s_scan);
	if (t < ns) s_scbase[t] = v[0];
	if (t == 0) s_scbase[ns] = tot[0];
	__syncthreads();
	if (tot[0] > ra.cap) return;      // flagged by k_slice_resolve
	OUT* s_lab = DynLds<OUT>::get();
	const uint32_t nexp = ra.ncomp_expect[zi];
	for (uint32_t s = t >> 6; s < ns; s += NW) {
		const uint32_t n = s_scbase[s + 1] - s_scbase[s];
		for (uint32_t j = t & (kWave - 1); j < n; j += kWave) {
			const uint32_t c = sa.sc_cc[rb + s_sbase[s] + j];
			uint64_t val = 0;
			if (c < nexp) val = ra.label_map[ra.comp_off[zi] + c];
			if (ra.has_label) val = (val == ra.label);
			s_lab[s_scbase[s] + j] = static_cast<OUT>(val);
		}
	}
	__syncthreads();
	strip_run_labels<OUT>(sa, rb, s_sbase, s_nruns, s_scbase, s_lab, static_cast<OUT*>(ra.run_label));
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
	const uint32_t sb = sa.strip_base[si];
	if (run < sb || run - sb >= nr) return 0xFFFFFFFFu;
	const uint64_t rb = sa.rbase[zi];
	return sa.sc_cc[rb + sb + sa.run_lid[rb + run]];
}

}  // namespace dev
}  // namespace ckl
