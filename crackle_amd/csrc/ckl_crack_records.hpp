// Crack codes -> crack RECORDS, binned by strip: the front half of the decoder's strip path.
// (Included by ckl_decode.hip inside namespace ckl, after the symbol machinery it shares with
// k_decode_cracks: tile_symbols, prev_smaller, markov_expand_parallel.)
//
// Replaces, for streams on the strip path, the rasteriser of k_decode_cracks
// (decode_(im)permissible_crack_code, src/crackcodes.hpp:706-876): that kernel builds whole planes
// in a workgroup's LDS (one 1024-thread workgroup per CU, 149 KiB) and spends most of its time
// there.  Here a slice's workgroup only resolves WHERE every stretch of moves starts:
//
//   k_crack_match     one workgroup of 512 per slice, two per CU (76 KiB of LDS): BOC index,
//                     symbols 16 codes per word (tile_symbols), control symbols recorded with
//                     the displacement before them, branch matching (match_controls_packed) ->
//                     the offset of every segment between 't' jumps.  On the way every word
//                     of 16 code positions is parked as it stands in the registers — its moves,
//                     which positions emit one, which are 't's, the displacement and the number
//                     of 't's before it (WordRec, 16 bytes) — and read back once the offsets
//                     are known: the word's segment offsets turn its displacement into the
//                     absolute start vertex (y << 16 | x) and the word into one 16-byte RECORD
//                     per pair of stretches between 't's (the record carries the jump between
//                     the two).  A record goes to the list of every strip of rows its moves can
//                     touch (one, seldom two), through one cursor per strip in LDS.
//   raster_record     (ckl_strips.hpp) the strip kernel walks the records of its list into the
//                     strip's two plane pieces in LDS — 8 KiB instead of 256 KiB — and goes on
//                     to label them without the planes making a round trip through HBM.
//
// Vertices are packed as y * 65536 + x in one 32-bit integer (sx, sy <= 65534): a move adds
// +-1 or +-65536, the crack a move crosses sits at the smaller of its two vertices — min() of
// the two integers.
#pragma once

constexpr int kRecBlock = 512;                    // threads per slice (two workgroups per CU); kRecBlockWide when a decode has no more slices than the chip has CUs
constexpr int kRecBlockWide = 1024;               // a slice alone on its CU: twice the threads, half the tiles and half the work per thread in every phase
// words of 16 code positions per thread and tile.  (Round 5: with the block scans reading all wavefront totals into registers the
// 512-thread kernel spilled at eight words — threadIdx.x and the wavefront's scan slot among others, fetched back from scratch memory
// in front of most barriers: 0.120 ms at C2; six words without spills 0.102; eight words with the DPP block scans, no spills: 0.098.)
constexpr uint32_t rec_words(int /*block*/) { return 8u; }
constexpr uint32_t rec_tile(int block) { return static_cast<uint32_t>(block) * rec_words(block) * 16u; }      // code positions per tile
constexpr uint32_t kRecMaxStrips = 512;           // strips per slice the LDS cursors cover
constexpr uint32_t kRecMaxDim = 65534;            // packed vertices: 16 bits per coordinate
constexpr uint32_t kEmitSegWindow = 2048;         // segment offsets k_crack_emit stages in LDS per workgroup

// control tables, positions packed (LDS: 16-bit indices / depths, global: 32-bit)
template <typename IDX, typename DEP>
struct CtlTablesP {
	unsigned long long* link;    // low: value (packed vertex / packed difference), high: index of the 't' it is relative to (or NONE)
	uint32_t* pos;               // packed displacement of the stream before the symbol
	DEP* depth;
	IDX* lastT;
	DEP* gmin;
	uint8_t* kind;
	uint32_t* seg;               // per segment: packed offset to add to the displacement (LDS: over depth | lastT, which are dead by then)
};

__device__ __forceinline__ uint32_t pack_vertex(uint32_t v, uint32_t sxe) { const uint32_t y = v / sxe; return (y << 16) | (v - y * sxe); }

// LDS bytes of the tables for n control symbols: link | pos | depth | lastT | gmin | kind
static inline size_t rec_lds_bytes(uint32_t n) {
	return static_cast<size_t>(n) * 8 + static_cast<size_t>(n) * 4 + static_cast<size_t>(n) * 2 * 2 + (static_cast<size_t>(n) / 7 + 48) * 2 + n + 16;
}

// Branch matching over the N control symbols of a slice (see match_controls): the same three steps —
// clamped depth, previous smaller depth through the tree of minima, pointer jumping — on packed
// positions, with every popping 't' searched by the thread that owns it (no worklist: the tree keeps
// a search at a handful of steps) and the segment offsets written over the depth tables.
// Output: seg[k] for the valid segments, their number in *s_valid_segs.
template <typename IDX, typename DEP, int BLOCK>
__device__ __forceinline__ void match_controls_packed(
	const CtlTablesP<IDX, DEP>& t, uint32_t N, const uint32_t* nodes, uint32_t n_nodes, uint32_t sxe, uint32_t sx, uint32_t sy,
	uint32_t* s_scan, int32_t* s_scanmax, uint32_t* s_first_dead, uint32_t* s_valid_segs, uint32_t* s_loff, uint32_t* s_lcnt, uint32_t& rerr,
	unsigned long long* dg = nullptr, uint32_t* wl = nullptr, uint32_t wl_cap = 0, uint32_t* s_wn = nullptr      // work list of the searches (LDS tables only) and its counter
) {
	unsigned long long dg_t = (kTuning && dg) ? __builtin_amdgcn_s_memtime() : 0ull;
	auto sub = [&](int slot) { if (kTuning && dg && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); dg[slot] += now - dg_t; dg_t = now; } };
	constexpr int NW = BLOCK / kWave;
	constexpr IDX NONE = static_cast<IDX>(~static_cast<IDX>(0));
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (N + BLOCK - 1) / BLOCK;
	const uint32_t i0 = min(N, tid * per), i1 = min(N, i0 + per);

	int32_t s = 0, mn = INT32_MAX, lt = -1;
	for (uint32_t i = i0; i < i1; i++) {
		const bool isT = t.kind[i] == SYM_T;
		s += isT ? -1 : 1;
		mn = s < mn ? s : mn;
		if (isT) lt = static_cast<int32_t>(i);
	}
	uint32_t v1[1] = { static_cast<uint32_t>(s) }, t1[1];
	block_excl_add<1, NW>(v1, t1, s_scan);
	const int32_t S0 = static_cast<int32_t>(v1[0]);
	int32_t neg_tot, lt_tot;
	const int32_t neg_ex = block_excl_max<NW>((i0 < i1) ? -(S0 + mn) : INT32_MIN, neg_tot, s_scanmax);
	const int32_t M0 = -(neg_ex > 0 ? neg_ex : 0);     // min(0, running minimum before my symbols)
	const int32_t LT0 = block_excl_max<NW>(lt, lt_tot, s_scanmax);

	uint32_t nT = 0, nCE = 0;
	{
		int32_t cur = LT0 < 0 ? -1 : LT0;
		int32_t sr = S0, m = M0;
		for (uint32_t i = i0; i < i1; i++) {
			const bool isT = t.kind[i] == SYM_T;
			const int32_t before = sr - m;
			sr += isT ? -1 : 1;
			m = sr < m ? sr : m;
			t.depth[i] = static_cast<DEP>(sr - m);
			if (isT) { nT++; nCE += (before == 0); cur = static_cast<int32_t>(i); }
			t.lastT[i] = cur < 0 ? NONE : static_cast<IDX>(cur);
		}
	}
	uint32_t v2[2] = { nT, nCE }, t2[2];
	block_excl_add<2, NW>(v2, t2, s_scan);      // its barriers publish depth[] and lastT[]
	const uint32_t T0 = v2[0], C0 = v2[1], totalT = t2[0];
	sub(4);
	// the tree of minima over depth[] (see prev_smaller)
	if (tid == 0) {
		uint32_t c = N, off = 0, l = 0;
		s_lcnt[0] = N; s_loff[0] = 0;
		while (c > 8u && l < 11u) {
			c = (c + 7u) / 8u;
			l++;
			s_lcnt[l] = c; s_loff[l] = off;
			off += (c + 7u) & ~7u;
		}
		s_lcnt[l + 1u] = 0;      // end marker
		if (s_wn) *s_wn = 0u;
	}
	__syncthreads();
	for (uint32_t l = 1; s_lcnt[l] != 0; l++) {
		const uint32_t cnt = s_lcnt[l], below = s_lcnt[l - 1];
		const DEP* src = l == 1 ? t.depth : t.gmin + s_loff[l - 1];
		for (uint32_t gi = tid; gi < cnt; gi += BLOCK) {
			int32_t mv = INT32_MAX;
			const uint32_t e = min(below, gi * 8u + 8u);
			for (uint32_t i = gi * 8u; i < e; i++) { const int32_t d = static_cast<int32_t>(src[i]); mv = d < mv ? d : mv; }
			t.gmin[s_loff[l] + gi] = static_cast<DEP>(mv);
		}
		__syncthreads();
	}
	sub(5);
	// ---- links.  A 't' on the empty stack ends its chain: the next chain starts at the next node of the
	// BOC index; past the last node the trailing pad codes begin (first_dead).  A 't' that pops returns
	// to its 'b': the symbol after the previous symbol of smaller depth.
	// Most 't's return to a 'b' a few symbols back, inside the thread's own stretch: `open` keeps the thread's last 64
	// symbols, bit h set = symbol i - 1 - h is a 'b' nobody has returned to yet, so the stack's top is its lowest bit.  Only a
	// 't' that finds the window empty searches the tree, and from the window's far end on (all symbols inside lie deeper).
	// Those searches (C2: 860 of a slice's 1 570 returns, four steps on average, nine at worst) are not made where they
	// turn up — a wavefront would walk the longest search of its lanes in every one of its rounds, nine times the steps its
	// lanes need — but listed and dealt out evenly afterwards.
	auto link_to = [&](uint32_t i, uint32_t j) {      // the 't' at i returns to the 'b' at j
		uint32_t val, ptr = kLinkNone;
		const uint32_t pos_j = t.pos[j];
		const IDX tp = t.lastT[j];
		if (tp == NONE) val = pack_vertex(nodes[0], sxe) + pos_j;
		else { val = pos_j - t.pos[tp]; ptr = static_cast<uint32_t>(tp); }
		t.link[i] = (static_cast<unsigned long long>(ptr) << 32) | val;
	};
	auto search = [&](uint32_t i, uint32_t own0, int32_t before) -> uint32_t {      // own0: where the stretch of i's thread begins
		const uint32_t known = min(64u, i - own0);
		uint32_t steps = 0;
		const uint32_t j = static_cast<uint32_t>(prev_smaller<DEP>(t.depth, t.gmin, s_loff, s_lcnt, static_cast<int32_t>(i - known), before, (kTuning && dg) ? &steps : nullptr) + 1);
		if (kTuning && dg) { atomicAdd(dg + 17, 1ull); atomicAdd(dg + 18, static_cast<unsigned long long>(steps)); }
		return j;
	};
	{
		int32_t sr = S0, m = M0;
		uint32_t c = C0;
		unsigned long long open = 0;
		for (uint32_t i = i0; i < i1; i++) {
			const bool isT = t.kind[i] == SYM_T;
			const int32_t before = sr - m;
			sr += isT ? -1 : 1;
			m = sr < m ? sr : m;
			if (!isT) { open = (open << 1) | 1ull; continue; }
			if (before == 0) {
				uint32_t val = 0;
				c++;
				if (c >= n_nodes) atomicMin(s_first_dead, i);
				else val = pack_vertex(nodes[c], sxe);
				t.link[i] = (static_cast<unsigned long long>(kLinkNone) << 32) | val;
				if (kTuning && dg) atomicAdd(dg + 20, 1ull);
			}
			else if (open) {
				link_to(i, i - 1u - static_cast<uint32_t>(__builtin_ctzll(open)));
				open &= open - 1ull;
				if (kTuning && dg) atomicAdd(dg + 19, 1ull);
			}
			else {
				uint32_t at = wl_cap;
				if (wl) at = atomicAdd(s_wn, 1u);
				if (at < wl_cap) wl[at] = i | (static_cast<uint32_t>(before) << 16);
				else link_to(i, search(i, i0, before));
			}
			open <<= 1;
		}
	}
	if (wl) {
		__syncthreads();
		const uint32_t n_wl = uni(min(*s_wn, wl_cap));
		for (uint32_t k = tid; k < n_wl; k += BLOCK) {
			const uint32_t item = wl[k];
			const uint32_t i = item & 0xFFFFu;
			link_to(i, search(i, i / per * per, static_cast<int32_t>(item >> 16)));
		}
	}
	__syncthreads();
	sub(6);
	const uint32_t first_dead = uni(*s_first_dead);
	const uint32_t n_eff = min(N, first_dead);
	// ---- pointer jumping: (value, parent) pairs updated in single 8-byte accesses, consistent under races
	for (uint32_t i = i0; i < i1; i++) {
		if (t.kind[i] != SYM_T || i >= n_eff) continue;
		unsigned long long* lk = t.link;
		unsigned long long me = __hip_atomic_load(lk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		for (uint32_t guard = 0; (me >> 32) != kLinkNone && guard <= N; guard++) {
			const unsigned long long other = __hip_atomic_load(lk + static_cast<uint32_t>(me >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			me = (other & 0xFFFFFFFF00000000ull) | static_cast<uint32_t>(static_cast<uint32_t>(me) + static_cast<uint32_t>(other));
			__hip_atomic_store(lk + i, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	__syncthreads();
	sub(7);
	// ---- segment offsets
	{
		uint32_t tc = T0;
		for (uint32_t i = i0; i < i1; i++) {
			if (t.kind[i] != SYM_T) continue;
			if (i < n_eff) {
				uint32_t A = static_cast<uint32_t>(t.link[i]);
				if ((A & 0xFFFFu) > sx || (A >> 16) > sy) { rerr |= ERR_RANGE; A = 0; }
				t.seg[tc + 1] = A - t.pos[i];
			}
			if (i == first_dead) *s_valid_segs = tc + 1u;
			tc++;
		}
	}
	if (tid == 0) {
		t.seg[0] = pack_vertex(nodes[0], sxe);
		if (first_dead >= N) *s_valid_segs = totalT + 1u;
	}
	__syncthreads();
	sub(8);
}

// a word of 16 code positions as k_crack_match parks it between its two passes
struct WordRec {               // per word of 16 code positions (16 bytes)
	uint32_t prevs;            // the move position k would emit, 2 bits each
	uint32_t flags;            // bit 2k: position k emits its move; bit 2k + 1: position k is a 't'
	uint32_t o_p;              // packed displacement of the stream before the word
	uint32_t o_t;              // 't's before the word
};
struct RecArgs {
	CrackArgs c;                 // stream, descriptors, markov scratch, global control tables (g_dx: positions, g_seg_x: segment offsets)
	RecordLists lists;
	uint32_t lds_controls;       // capacity of the LDS tables
	uint32_t lds_bytes;
	WordRec* words;              // [word_base[zi] + word]
	const uint64_t* word_base;   // [nslices]
	uint32_t* fused_ctl;         // k_strip_fused's words (fused_ctl_words of ticket counters and timeout, arrive[fused_n], ready[fused_n]): zeroed here, or null
	uint32_t fused_n, fused_ctl_words;
	unsigned long long* diag;    // tuning builds: cycle stamps, summed over the slices
	uint32_t ablate;             // tuning builds (CKL_ABLATE, results wrong): 0x400000 no parked-word stores, 0x800000 the parked words are loaded and dropped, 0x1000000 they are not loaded
};

// the seldom-taken parts of k_crack_match (as functions of their own, not inlined, they made the kernel
// slower: 0.131 against 0.101 ms at C2)
template <bool GLOBAL, int BLOCK>
__device__ __forceinline__ void rec_markov_expand(
	const uint8_t* s, uint32_t nbytes, int order, const uint8_t* model_g, uint32_t cap, uint32_t* upacked, uint32_t* lds, uint32_t* gscratch,
	bool model_in_lds, uint32_t* s_scan, uint32_t* s_total, uint32_t* out2 /* codes, error bits */, uint32_t lds_words
) {
	uint32_t nc = 0, er = 0;
	markov_expand_parallel<GLOBAL, BLOCK>(s, nbytes, order, model_g, cap, upacked, lds, gscratch, model_in_lds, s_scan, s_total, nc, er, lds_words);
	if (threadIdx.x == 0) { out2[0] = nc; out2[1] = er; }
}
// one thread: the markov bitstream -> difference codes, 16 per word (markov.hpp:268-313); returns the codes, *err the error bits
__device__ __forceinline__ uint32_t rec_markov_serial(const uint8_t* s, uint32_t nbytes, int order, const uint8_t* model, uint32_t cap, uint32_t* upacked, uint32_t* err) {
	const int shift = 2 * (order - 1);
	const uint32_t start = s[0] & 3u;
	uint32_t m = 1, word = start;
	uint32_t ctx = start << shift;
	int pos = 2;
	for (uint32_t i = 0; i < nbytes; i++) {
		uint32_t byte = s[i];
		if (i + 1 < nbytes) byte |= static_cast<uint32_t>(s[i + 1]) << 8;
		while (pos < 8) {
			const uint32_t cp = (byte >> pos) & 7u;
			uint32_t rank;
			if ((cp & 1u) == 0) { rank = 0; pos += 1; }
			else if ((cp & 2u) == 0) { rank = 1; pos += 2; }
			else if ((cp & 4u) == 0) { rank = 2; pos += 3; }
			else { rank = 3; pos += 3; }
			const uint32_t v = model[ctx * 4u + rank];
			if (m < cap) {
				word |= v << (2u * (m & 15u));
				m++;
				if ((m & 15u) == 0) { upacked[(m >> 4) - 1u] = word; word = 0; }
			}
			else *err |= ERR_CAPACITY;
			ctx = (ctx >> 2) + (v << shift);
		}
		pos -= 8;
	}
	upacked[m >> 4] = word;
	return m;
}
// branch matching with the tables in global memory (a slice with more control symbols than the LDS tables hold)
template <int BLOCK>
__device__ __forceinline__ void rec_match_global(
	const CtlTablesP<uint32_t, int32_t>* gt, uint32_t n, const uint32_t* nodes, uint32_t n_nodes, uint32_t sxe, uint32_t sx, uint32_t sy,
	uint32_t* s_scan, int32_t* s_scanmax, uint32_t* s_first_dead, uint32_t* s_valid_segs, uint32_t* s_loff, uint32_t* s_lcnt, uint32_t* rerr_out
) {
	uint32_t rerr = 0;
	match_controls_packed<uint32_t, int32_t, BLOCK>(*gt, n, nodes, n_nodes, sxe, sx, sy, s_scan, s_scanmax, s_first_dead, s_valid_segs, s_loff, s_lcnt, rerr);
	if (rerr) atomicOr(rerr_out, rerr);
}

// Do the moves of a stretch — the positions of `part` (spread mask, bit 2k = position k emits) of a word whose
// moves are `prevs` — from vertex `start` (y << 16 | x, inside the grid) stay inside the vertex grid?  The strip
// kernel rasterises without looking (k_strip_ccl2 clamps what leaves its strip), so a stream whose trail walks
// off the image is caught here: the counts of the four directions settle it for every stretch that is not next
// to a border, the others are stepped through (crackcodes.hpp:706-862 indexes its edge vector with such a vertex).
__device__ __forceinline__ bool stretch_in_grid(uint32_t start, uint32_t prevs, uint32_t part, uint32_t sx, uint32_t sy) {
	const uint32_t x = start & 0xFFFFu, y = start >> 16;
	const uint32_t hz = part & prevs, vt = part & ~prevs;
	const uint32_t nr = __popc(hz & ~(prevs >> 1)), nl = __popc(hz & (prevs >> 1)), nd = __popc(vt & (prevs >> 1)), nu = __popc(vt & ~(prevs >> 1));
	if (nl <= x && x + nr <= sx && nu <= y && y + nd <= sy) return true;
	uint32_t p = start;
	for (uint32_t m = part; m; m &= m - 1u) {
		const uint32_t kind = (prevs >> (__ffs(m) - 1u)) & 3u;
		const uint32_t unit = (kind & 1u) ? 1u : 0x10000u;
		p = ((kind ^ (kind >> 1)) & 1u) ? p + unit : p - unit;      // right (1) and down (2) add
		if ((p & 0xFFFFu) > sx || (p >> 16) > sy) return false;
	}
	return true;
}

// The records of one parked word: the stretches between its 't's, two per record ([A] t [B] | t [A] t [B]
// | ...; the record carries the jump between its two), each record into the list of every strip its
// moves can touch.  seg: the segments' offsets (LDS or global), cursor: the slice's list lengths in LDS.
__device__ __forceinline__ void word_to_records(
	const uint4& word, const uint32_t* seg, uint32_t valid_segs, const RecordLists& L, uint4* lists, uint32_t* cursor,
	uint32_t sx, uint32_t sy, uint32_t& rerr
) {
	const uint32_t prevs = word.x, ms = word.y & kLo, isT = (word.y >> 1) & kLo, o_p = word.z, o_t0 = word.w;
	if ((ms | isT) == 0u) return;
	const uint32_t nstrips = L.nstrips;
	const uint32_t mR = ms & ~(prevs >> 1) & prevs, mL = ms & (prevs >> 1) & prevs, mD = ms & (prevs >> 1) & ~prevs, mU = ms & ~(prevs >> 1) & ~prevs;
	// packed displacement of the word's moves at the positions of `mask`
	auto disp = [&](uint32_t mask) -> uint32_t { return __popc(mR & mask) - __popc(mL & mask) + ((__popc(mD & mask) - __popc(mU & mask)) << 16); };
	// the record goes to the strips k0 .. k1 and, when it jumps, to those of k2 .. k3 not among them
	auto put_to = [&](const uint4& rec, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
		for (uint32_t k = k0; k <= k1; k++) {
			const uint32_t at = atomicAdd(&cursor[k], 1u);
			if (at < L.cap) lists[static_cast<uint64_t>(k) * L.cap + at] = rec;
		}
		for (uint32_t k = k2; k <= k3; k++) {
			if (k >= k0 && k <= k1) continue;
			const uint32_t at = atomicAdd(&cursor[k], 1u);
			if (at < L.cap) lists[static_cast<uint64_t>(k) * L.cap + at] = rec;
		}
	};
	// strips of the rows a stretch of moves from vertex `start` can touch: the vertex rows
	// [y - ups, y + downs] (vertical moves cross the plane row of their smaller vertex, horizontal ones
	// that of their own); false when the vertex is outside the grid
	auto strips_of = [&](uint32_t start, uint32_t part, uint32_t& k0, uint32_t& k1) -> bool {
		const uint32_t y = start >> 16, x = start & 0xFFFFu;
		if (x > sx || y > sy || !stretch_in_grid(start, prevs, part, sx, sy)) { rerr |= ERR_RANGE; return false; }
		const uint32_t nu = __popc(mU & part), nd = __popc(mD & part);
		const uint32_t ya = y > nu ? y - nu : 0u, yb = min(y + nd, sy);
		k0 = L.strip_of(ya); k1 = min(L.strip_of(yb), nstrips - 1u);
		return true;
	};
	uint32_t ot = o_t0, done = 0;      // done: spread mask of the positions handed out (and of the 't's passed)
	for (uint32_t tm = isT; ; ) {
		const uint32_t bA = tm ? __ffs(tm) - 1u : 32u;
		const uint32_t uptoA = bA >= 32u ? kLo : ((1u << bA) - 1u) & kLo;
		const uint32_t partA = ms & uptoA & ~done;
		const bool actA = ot < valid_segs;
		const uint32_t offA = actA ? seg[ot] : 0u;
		const uint32_t startA = offA + o_p + disp(done);
		uint32_t k0 = 1, k1 = 0, k2 = 1, k3 = 0;
		if (!tm) {
			if (partA && actA && strips_of(startA, partA, k0, k1)) put_to(make_uint4(startA, prevs, partA, 0u), k0, k1, 1u, 0u);
			break;
		}
		const uint32_t tm2 = tm & (tm - 1u);
		const uint32_t bB = tm2 ? __ffs(tm2) - 1u : 32u;
		const uint32_t uptoB = bB >= 32u ? kLo : ((1u << bB) - 1u) & kLo;
		const uint32_t behindA = uptoA | (1u << bA);
		const uint32_t partB = ms & uptoB & ~behindA;
		const bool actB = ot + 1u < valid_segs;
		const uint32_t offB = actB ? seg[ot + 1u] : 0u;
		const uint32_t startB = offB + o_p + disp(behindA);
		const bool hasA = partA && actA && strips_of(startA, partA, k0, k1);
		const bool hasB = partB && actB && strips_of(startB, partB, k2, k3);
		if (hasA && hasB) put_to(make_uint4(startA, prevs, partA | partB | (1u << (bA + 1u)), offB - offA), k0, k1, k2, k3);
		else if (hasA) put_to(make_uint4(startA, prevs, partA, 0u), k0, k1, 1u, 0u);
		else if (hasB) put_to(make_uint4(startB, prevs, partB, 0u), k2, k3, 1u, 0u);
		if (!tm2) break;
		done = uptoB | (1u << bB);
		ot += 2u;
		tm = tm2 & (tm2 - 1u);
	}
}

// grid = slices of the launch, block = BLOCK (BLOCK or BLOCKWide), dynamic LDS = lds_bytes
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 4) k_crack_match(RecArgs ra) {
	constexpr int kRecWaves = BLOCK / kWave;
	constexpr uint32_t kRecTile = rec_tile(BLOCK);
	constexpr uint32_t kRecWords = rec_words(BLOCK);
	extern __shared__ __attribute__((aligned(16))) unsigned long long s_dyn[];
	__shared__ uint32_t s_scan[4 * kRecWaves];
	__shared__ int32_t s_scanmax[kRecWaves];
	__shared__ uint8_t s_last_move[BLOCK];
	__shared__ uint8_t s_last_ctrl[BLOCK];
	__shared__ uint32_t s_nnodes, s_ncodes, s_valid_segs, s_err, s_first_dead;
	__shared__ uint32_t s_loff[14], s_lcnt[14];
	__shared__ uint32_t s_mk_parallel, s_mk_total, s_index_end;
	__shared__ uint32_t s_cursor[kRecMaxStrips];      // lengths of the slice's record lists

	// tuning builds: cycle stamps of thread 0, summed per workgroup in LDS and added to the launch's counters at the end
	// (512 workgroups adding to the same words after every phase waited for each other's atomics and distorted what they measured)
	__shared__ unsigned long long s_dg[kTuning ? 32 : 1];
	unsigned long long* const dgp = (kTuning && ra.diag) ? s_dg : nullptr;
	if (kTuning && dgp) { if (threadIdx.x < 32) s_dg[threadIdx.x] = 0ull; __syncthreads(); }
	unsigned long long d_t = (kTuning && dgp) ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) { if (kTuning && dgp && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); s_dg[slot] += now - d_t; d_t = now; } };
	const CrackArgs& a = ra.c;
	const uint32_t zi = blockIdx.x + a.zbase;
	const uint32_t tid = threadIdx.x;
	const uint8_t* code = a.stream + a.code_off[zi];
	const uint32_t code_len = a.code_len[zi];
	const uint64_t cb = a.cbase[zi];
	const uint32_t cap = a.ccap[zi];
	uint32_t* nodes = a.nodes + a.nbase[zi];
	const uint32_t ncap = a.ncap[zi];
	const uint32_t sxe = a.sx + 1, sye = a.sy + 1;
	const uint32_t sx = a.sx, sy = a.sy;
	uint32_t* upacked = a.upacked ? a.upacked + cb / 16u + 2ull * zi : nullptr;
	const uint32_t nstrips = ra.lists.nstrips;
	for (uint32_t k = tid; k < nstrips; k += BLOCK) s_cursor[k] = 0u;

	// the head of the slice's code (its beginning-of-chain index, usually under a hundred bytes) is staged in LDS by
	// all threads: the one thread that parses it would otherwise make a trip to memory per field
	constexpr uint32_t kIndexStage = 2048;
	uint8_t* s_idx = reinterpret_cast<uint8_t*>(s_dyn);
	const uint32_t stage_n = min(min(code_len, kIndexStage), ra.lds_bytes);
	for (uint32_t i = tid; i < stage_n; i += BLOCK) s_idx[i] = code[i];
	__syncthreads();
	auto rd_idx = [&](const uint8_t* p, int w) -> uint32_t {      // (the staged bytes are only valid until the markov expansion takes the LDS)
		const uint32_t at = static_cast<uint32_t>(p - code);
		uint32_t v = 0;
		if (at + static_cast<uint32_t>(w) <= stage_n) for (int i = 0; i < w; i++) v |= static_cast<uint32_t>(s_idx[at + i]) << (8 * i);
		else for (int i = 0; i < w; i++) v |= static_cast<uint32_t>(p[i]) << (8 * i);
		return v;
	};

	// ---- beginning-of-chain index (crackcodes.hpp:283-316), one thread ----
	if (tid == 0) {
		s_mk_parallel = 0;
		uint32_t err = 0, nn = 0, ncodes = 0;
		uint32_t index_end = 0;
		if (code_len < 4u + a.yw) {
			err |= ERR_BOC;
		}
		else {
			const uint32_t index_size = rd_idx(code, 4);
			index_end = 4u + index_size;
			if (index_size < static_cast<uint32_t>(a.yw) || index_end > code_len || index_end < 4u) {
				err |= ERR_BOC;
				index_end = code_len;
			}
			else {
				uint32_t idx = 4;
				const uint32_t num_y = rd_idx(code + idx, a.yw);
				idx += a.yw;
				uint32_t y = 0;
				for (uint32_t yi = 0; yi < num_y && !err; yi++) {
					if (idx + a.yw + a.xw > index_end) { err |= ERR_BOC; break; }
					y += rd_idx(code + idx, a.yw); idx += a.yw;
					const uint32_t num_x = rd_idx(code + idx, a.xw); idx += a.xw;
					uint32_t x = 0;
					for (uint32_t xi = 0; xi < num_x; xi++) {
						if (idx + a.xw > index_end) { err |= ERR_BOC; break; }
						x += rd_idx(code + idx, a.xw); idx += a.xw;
						if (x >= sxe || y >= sye || nn >= ncap) { err |= ERR_BOC; break; }
						nodes[nn++] = x + sxe * y;
					}
				}
			}
			// markov bitstream -> difference codes (markov.hpp:268-313): by the whole workgroup when the
			// slice's tables fit the LDS or the global scratch, else serially here
			const uint32_t nbytes = code_len - index_end;
			bool mdl_lds = false;
			if (a.markov_order == 0) {
				ncodes = nbytes * 4u;
			}
			else if (nbytes > 0 && !a.markov_serial && markov_lds_need(nbytes, cap, a.markov_order, ra.lds_bytes, mdl_lds, BLOCK) <= ra.lds_bytes) {
				s_mk_parallel = 1u + (mdl_lds ? 1u : 0u);
			}
			else if (nbytes > 0 && !a.markov_serial && a.mkscratch && ra.lds_bytes >= 2u * BLOCK * 4u + 768u * 4u) {      // (context words + the byte table of markov_expand_parallel)
				s_mk_parallel = 3u + (markov_model_fits_alone(a.markov_order, ra.lds_bytes, BLOCK) ? 1u : 0u);
			}
			else if (nbytes > 0) ncodes = rec_markov_serial(code + index_end, nbytes, a.markov_order, a.model, cap, upacked, &err);
		}
		if (ncodes > cap) { ncodes = cap; err |= ERR_CAPACITY; }
		s_nnodes = nn;
		s_ncodes = ncodes;
		s_err = err;
		s_valid_segs = 1;
		s_first_dead = 0xFFFFFFFFu;
		s_index_end = index_end;
	}
	__syncthreads();
	if (s_mk_parallel) {
		const uint32_t ie = s_index_end;
		const uint32_t mode = s_mk_parallel;
		uint32_t* gsc = a.mkscratch ? a.mkscratch + a.mkbase[zi] : nullptr;
		__shared__ uint32_t s_mk_out[2];
		if (mode <= 2u) rec_markov_expand<false, BLOCK>(code + ie, code_len - ie, a.markov_order, a.model, cap, upacked, reinterpret_cast<uint32_t*>(s_dyn), gsc, mode == 2u, s_scan, &s_mk_total, s_mk_out, ra.lds_bytes / 4u);
		else rec_markov_expand<true, BLOCK>(code + ie, code_len - ie, a.markov_order, a.model, cap, upacked, reinterpret_cast<uint32_t*>(s_dyn), gsc, mode == 4u, s_scan, &s_mk_total, s_mk_out, ra.lds_bytes / 4u);
		if (tid == 0) { s_ncodes = s_mk_out[0]; if (s_mk_out[1]) s_err |= s_mk_out[1]; }
		__syncthreads();
	}
	if (a.markov_order) __threadfence_block();
	stamp(0);
	// (values every lane holds alike, read from LDS or memory, are moved to scalar registers: the kernel is short of vector registers)
	const uint32_t n_codes = uni(s_ncodes);
	const uint32_t n_nodes = uni(s_nnodes);
	// the code positions 0 .. n_codes are dealt out in tiles of BLOCK x span, span the same for every
	// tile of the slice (a multiple of 16, at most 128): all threads get an even share
	const uint32_t n_tiles = n_codes / kRecTile + 1u;
	const uint32_t span = min(kRecWords * 16u, ((n_codes / n_tiles + BLOCK) / BLOCK + 15u) / 16u * 16u);
	const uint32_t tile_step = span * BLOCK;
	const uint32_t index_end = uni(s_index_end);      // (what the index parse found: 4 + its size, or the end of the code when it is broken — no second trip to memory for it)
	const uint8_t* packed = code + index_end;
	const uint32_t* words;
	uint32_t wshift;
	if (a.markov_order) { words = upacked; wshift = 0; }
	else {
		const uintptr_t pa = reinterpret_cast<uintptr_t>(packed);
		words = reinterpret_cast<const uint32_t*>(pa & ~static_cast<uintptr_t>(3));
		wshift = static_cast<uint32_t>(pa & 3u) * 8u;
	}

	// control tables: LDS when the slice's control symbols fit, else global; the segment offsets go to
	// global memory either way (k_crack_emit reads them)
	const uint32_t lcap = ra.lds_controls;
	const uint64_t kb = cb / 2u + 4ull * zi;
	const uint32_t kcap = cap / 2u + 4u;
	CtlTablesP<uint16_t, int16_t> lt;
	{
		unsigned long long* p8 = s_dyn;
		lt.link = p8; p8 += lcap;
		uint32_t* p4 = reinterpret_cast<uint32_t*>(p8);
		lt.pos = p4; p4 += lcap;
		uint16_t* p2 = reinterpret_cast<uint16_t*>(p4);
		lt.depth = reinterpret_cast<int16_t*>(p2); p2 += lcap;
		lt.lastT = p2; p2 += lcap;
		lt.gmin = reinterpret_cast<int16_t*>(p2); p2 += lcap / 7 + 48;
		lt.kind = reinterpret_cast<uint8_t*>(p2);
		lt.seg = reinterpret_cast<uint32_t*>(lt.depth);      // over depth | lastT, dead by the time the offsets are written
	}
	uint32_t rerr = 0;
	const bool have_cracks = n_nodes > 0 && n_codes > 0;
	uint32_t n_words = 0;

	if (have_cracks) {
		// ---- the control symbols ('b' / 't', ~3 % of the stream) with the displacement before them, and
		// what every thread's stretch inherits
		WordRec* wout = ra.words + ra.word_base[zi];
		const uint32_t words_per = span / 16u;
		TileCarry c;
		uint32_t t_before = 0;      // 't's of the tiles before
		for (uint32_t tile = 0; tile <= n_codes; tile += tile_step) {
			WordSyms ws[kRecWords];
			uint32_t o_a, o_dx, o_dy;
			tile_symbols<false, BLOCK, kRecWords>(words, wshift, n_codes, span, tile, c, ws, o_a, o_dx, o_dy, s_scan, s_scanmax, s_last_move, s_last_ctrl, dgp);
			stamp(1);
			uint32_t o_p = (o_dy << 16) + o_dx;
			uint32_t nt = 0;
#pragma unroll
			for (uint32_t j = 0; j < kRecWords; j++) nt += __popc(ws[j].isT);
			uint32_t vt[1] = { nt }, tt[1];
			block_excl_add<1, kRecWaves>(vt, tt, s_scan);
			uint32_t o_t = t_before + vt[0];
			t_before = uni(t_before + tt[0]);
			// word j of all threads side by side (one contiguous KiB per wavefront and store): the array's
			// order is [tile][j][thread], not stream order — k_crack_bin does not care
			const uint32_t w0 = (tile / 16u) + tid;
			n_words = (tile + tile_step) / 16u;
#pragma unroll
			for (uint32_t j = 0; j < kRecWords; j++) {
				const WordSyms& w = ws[j];
				const uint32_t mR = w.right(), mL = w.left(), mD = w.down(), mU = w.up();
				for (uint32_t m = w.ctl; m; m &= m - 1u) {
					const uint32_t b = __ffs(m) - 1u;
					const uint32_t below = (1u << b) - 1u;
					const uint32_t kind = ((w.isT >> b) & 1u) ? SYM_T : SYM_B;
					const uint32_t cp = o_p + __popc(mR & below) - __popc(mL & below) + ((__popc(mD & below) - __popc(mU & below)) << 16);
					if (o_a + 2u < lcap) { lt.kind[o_a] = static_cast<uint8_t>(kind); lt.pos[o_a] = cp; }
					else if (o_a + 2u < kcap) { a.g_kind[kb + o_a] = static_cast<uint8_t>(kind); a.g_dx[kb + o_a] = cp; }
					o_a++;
				}
				if (j < words_per && !(kTuning && (ra.ablate & 0x400000u))) *reinterpret_cast<uint4*>(wout + w0 + j * BLOCK) = make_uint4(w.prevs, w.ms | (w.isT << 1), o_p, o_t);
				o_t += __popc(w.isT);
				o_p += __popc(mR) - __popc(mL) + ((__popc(mD) - __popc(mU)) << 16);
			}
			__syncthreads();
			stamp(2);
		}
		// ---- branch matching
		const uint32_t n_ctl = uni(c.a);
		const uint32_t* seg = lt.seg;
		if (n_ctl + 2u <= lcap) {
			match_controls_packed<uint16_t, int16_t, BLOCK>(lt, n_ctl, nodes, n_nodes, sxe, sx, sy, s_scan, s_scanmax, &s_first_dead, &s_valid_segs, s_loff, s_lcnt, rerr, dgp,
				reinterpret_cast<uint32_t*>(lt.link + n_ctl), (lcap - n_ctl) * 2u, &s_mk_total);      // (the link table's unused end holds the list of searches)
		}
		else {
			// more control symbols than the LDS tables hold: the first of them were recorded in LDS
			CtlTablesP<uint32_t, int32_t> gt;
			gt.kind = a.g_kind + kb; gt.pos = a.g_dx + kb; gt.depth = a.g_depth + kb; gt.lastT = a.g_lastT + kb;
			gt.link = a.g_link + kb; gt.seg = a.g_seg_x + kb; gt.gmin = a.g_gmin + kb;
			uint32_t n = n_ctl;
			if (n + 2u >= kcap) { n = kcap - 3u; rerr |= ERR_CAPACITY; }
			for (uint32_t i = tid; i + 2u < lcap && i < n; i += BLOCK) { gt.kind[i] = lt.kind[i]; gt.pos[i] = lt.pos[i]; }
			__syncthreads();
			__threadfence_block();
			rec_match_global<BLOCK>(&gt, n, nodes, n_nodes, sxe, sx, sy, s_scan, s_scanmax, &s_first_dead, &s_valid_segs, s_loff, s_lcnt, &s_err);
			__threadfence_block();
			seg = gt.seg;
		}
		stamp(3);
		// ---- the parked words once more (through L2): records into the strips' lists.  Four words out of five
		// hold no 't': they become their one record on the spot; the others are queued (in the link table,
		// which is free now) and dealt out again, so that the loop over stretches runs with all lanes busy
		// instead of once per word for the sake of a few lanes.
		{
			const uint32_t valid_segs = uni(s_valid_segs);
			const RecordLists& L = ra.lists;
			uint4* lists = L.rec + static_cast<uint64_t>(zi) * nstrips * L.cap;
			const uint4* wsrc = reinterpret_cast<const uint4*>(wout);
			uint4* queue = reinterpret_cast<uint4*>(lt.link);      // the words themselves (16 bytes each: the link table holds 8 per control symbol)
			const uint32_t queue_cap = lcap / 2u;
			uint32_t* s_qn = &s_mk_total;      // (the markov expansion is long done)
			if (tid == 0) *s_qn = 0u;
			__syncthreads();
			typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
			auto load_word = [&](uint32_t w) -> uint4 {
				const u32x4_t raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wsrc + w));      // written by other threads of the workgroup: not through L1
				return make_uint4(raw.x, raw.y, raw.z, raw.w);
			};
			constexpr uint32_t kBatch = 8;      // words of a thread in flight (a slice of C2 gives a thread 14: two trips to memory)
			for (uint32_t w0 = 0; w0 < n_words; w0 += kBatch * BLOCK) {
				uint4 wr[kBatch];
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) {
					const uint32_t w = w0 + q * BLOCK + tid;
					if (kTuning && (ra.ablate & 0x1000000u)) { wr[q] = make_uint4(0u, 0u, 0u, 0u); continue; }
					wr[q] = load_word(w < n_words ? w : 0u);
					if (w >= n_words) wr[q].y = 0u;
					if (kTuning && (ra.ablate & 0x800000u)) wr[q].y = 0u;
				}
				// The batch's words side by side, step by step — segment offset (LDS), start vertex and strips, a slot from the
				// strip's cursor (LDS atomic), the store: done word by word, each behind its own branches, a thread went
				// through these latencies once per word, ~1 000 cycles each.
				uint32_t off[kBatch];
				bool dir[kBatch];
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) {
					const uint32_t flags = wr[q].y;
					dir[q] = flags != 0u && (flags & 0xAAAAAAAAu) == 0u && wr[q].w < valid_segs;      // one stretch, one record
					off[q] = dir[q] ? seg[wr[q].w] : 0u;
				}
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) {
					if ((wr[q].y & 0xAAAAAAAAu) == 0u) continue;      // a 't' in the word: later, with its like
					const uint32_t at = atomicAdd(s_qn, 1u);
					if (at < queue_cap) queue[at] = wr[q];
					else word_to_records(wr[q], seg, valid_segs, L, lists, s_cursor, sx, sy, rerr);
				}
				uint32_t k0[kBatch], k1[kBatch], at0[kBatch];
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) {
					const uint32_t prevs = wr[q].x, ms = wr[q].y;
					const uint32_t start = off[q] + wr[q].z;
					wr[q].z = start;
					const uint32_t y = start >> 16, x = start & 0xFFFFu;
					if (dir[q] && (x > sx || y > sy || !stretch_in_grid(start, prevs, ms, sx, sy))) { rerr |= ERR_RANGE; dir[q] = false; }
					const uint32_t nu = __popc(ms & ~(prevs >> 1) & ~prevs), nd = __popc(ms & (prevs >> 1) & ~prevs);
					k0[q] = L.strip_of(min(y > nu ? y - nu : 0u, sy));
					k1[q] = min(L.strip_of(min(y + nd, sy)), nstrips - 1u);
				}
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) at0[q] = dir[q] ? atomicAdd(&s_cursor[k0[q]], 1u) : L.cap;
#pragma unroll
				for (uint32_t q = 0; q < kBatch; q++) {
					if (!dir[q]) continue;
					const uint4 rec = make_uint4(wr[q].z, wr[q].x, wr[q].y, 0u);
					if (at0[q] < L.cap) lists[static_cast<uint64_t>(k0[q]) * L.cap + at0[q]] = rec;
					for (uint32_t k = k0[q] + 1u; k <= k1[q]; k++) {
						const uint32_t at = atomicAdd(&s_cursor[k], 1u);
						if (at < L.cap) lists[static_cast<uint64_t>(k) * L.cap + at] = rec;
					}
				}
			}
			__syncthreads();
			stamp(15);
			const uint32_t qn = uni(min(*s_qn, queue_cap));
			for (uint32_t i = tid; i < qn; i += BLOCK) word_to_records(queue[i], seg, valid_segs, L, lists, s_cursor, sx, sy, rerr);
		}
	}
	__syncthreads();
	stamp(16);
	{
		const RecordLists& L = ra.lists;
		for (uint32_t k = tid; k < nstrips; k += BLOCK) {
			const uint32_t n = s_cursor[k];
			L.count[static_cast<uint64_t>(zi) * nstrips + k] = min(n, L.cap);
			if (n > L.cap) rerr |= ERR_LIST;
		}
	}
	stamp(9);
	if (rerr) atomicOr(&s_err, rerr);
	__syncthreads();
	if (tid == 0) {
		a.slice_err[zi] = s_err;      // later kernels of the decode OR their bits in
		if (a.overflow && blockIdx.x == 0 && a.zbase == 0) *a.overflow = 0u;      // the strip kernels' overflow word (this is the first kernel of the decode)
		if (ra.fused_ctl) {      // the next launch's ticket / arrival / ready words (a kernel boundary lies between)
			ra.fused_ctl[ra.fused_ctl_words + zi] = 0u;
			ra.fused_ctl[ra.fused_ctl_words + ra.fused_n + zi] = 0u;
		}
	}
	if (ra.fused_ctl && blockIdx.x == 0) for (uint32_t w = tid; w < ra.fused_ctl_words; w += BLOCK) ra.fused_ctl[w] = 0u;      // ticket counters, timeout
	if (kTuning && dgp && tid < 32 && s_dg[tid]) atomicAdd(ra.diag + tid, s_dg[tid]);
}

