// Decode path: .ckl bytes resident in HBM -> label volume in HBM.
// Replaces crackle::decompress<LABEL,OUT> (src/crackle.hpp:503-663) and the
// per-slice functions it calls:
//   read_boc_index / packed_codepoints_to_symbols   src/crackcodes.hpp:283-316, 523-603
//   markov::decode_codepoints / codepoints_to_symbols src/markov.hpp:268-323, crackcodes.hpp:606-676
//   decode_(im)permissible_crack_code (rasteriser)   src/crackcodes.hpp:706-876
//   cc3d::color_connectivity_graph + relabel         src/cc3d.hpp:114-254
//   crc32c of the component image                    src/crackle.hpp:599-611
//   labels::decode_flat / decode_condensed_pins      src/labels.hpp:453-617
//   the paint loop                                   src/crackle.hpp:617-656
//
// Kernels:
//   k_decode_cracks   one workgroup of 1024 per slice: BOC index, 2-bit unpack + mod-4 prefix
//                     sum (undo the difference code; markov streams are expanded by the whole
//                     workgroup first), 16 codes per word bit-parallel symbol derivation,
//                     branch matching of the control symbols (clamped depth scan, previous
//                     smaller value through a min tree, pointer jumping), then the two crack
//                     planes rasterised in LDS, one plane per pass, and stored.
//   k_run_index       horizontal runs of each slice from the vertical-crack plane:
//                     exclusive prefix sum of break counts per 32-pixel word.
//   k_run_union_*     union-find over RUNS (not pixels): vertically adjacent runs that are
//                     connected through the horizontal-crack plane are united, strip by
//                     strip in LDS, the seams between strips on the global forest.
//   k_run_count/rank/assign  roots ranked in raster order = the reference's component ids
//                     (cc3d.hpp:114-144); crc32c of the (never materialised) component
//                     image accumulated per run from a geometric-sum table; flat labels
//                     are applied to the runs in the same pass.
//   k_label_map_*     component -> label tables (flat keys / pins)
//   k_run_labels      run -> label (pins)
//   k_paint_runs      streams the output: per 4 pixels one plane word, a popcount and a
//                     look-up in an LDS-staged run->label table; 16-byte streaming stores.
//   k_run_stats       per-label voxel counts, coordinate sums and boxes from the runs
//   k_vcg             voxel connectivity graph from the planes (+ labels for the z bits)
#include "ckl_common.hpp"
#include "ckl_runs.hpp"
#include "ckl_strips.hpp"
#include "ckl_strips2.hpp"
#include "ckl_contours.hpp"
#include "ckl_pins.hpp"

#include <algorithm>
#include <deque>
#include <chrono>
#include <memory>
#include <mutex>
#include <type_traits>
#include <tuple>

namespace ckl {

using namespace dev;

// ------------------------------------------------------------------------------
// crack code -> crack planes
// ------------------------------------------------------------------------------
enum : uint8_t { SYM_U = 0, SYM_R = 1, SYM_D = 2, SYM_L = 3, SYM_B = 4, SYM_T = 5 };

struct CrackArgs {
	const uint8_t* stream;
	const uint64_t* code_off;    // [nslices] byte offset of each slice's crack code
	const uint32_t* code_len;    // [nslices]
	const uint64_t* cbase;       // [nslices] running total of the code capacities
	const uint32_t* ccap;        // [nslices] capacity (codes) of this slice
	const uint64_t* nbase;       // [nslices] base index into `nodes`
	const uint32_t* ncap;        // [nslices]
	int sx, sy;
	int xw, yw;
	int markov_order;
	const uint8_t* model;        // [4^order][4] rank -> symbol
	uint32_t* symbuf;            // slices whose codes span several tiles: symbols of every tile parked between the band passes
	const uint64_t* symbase;     // [nslices] word offsets into symbuf
	uint32_t* mkscratch;         // markov only: per-slice scratch (payload copy + ranks) for slices whose tables exceed the LDS
	const uint64_t* mkbase;      // [nslices] word offsets into mkscratch
	uint32_t* upacked;           // markov only: decoded difference codes, 16 per word (slice base cbase/16 + 2 zi)
	// control symbol tables in global memory, used when a slice has more control symbols
	// than the LDS tables hold (slice base cbase/2 + 4 zi, capacity ccap/2 + 4)
	uint8_t* g_kind;
	uint32_t* g_dx;
	uint32_t* g_dy;
	int32_t* g_depth;
	uint32_t* g_lastT;
	unsigned long long* g_link;
	uint32_t* g_seg_x;
	uint32_t* g_seg_y;
	int32_t* g_gmin;
	uint32_t* nodes;
	uint32_t* planeV;
	uint32_t* planeH;
	uint32_t row_words;
	uint64_t plane_words;
	uint32_t lds_controls;       // capacity of the LDS control tables (dynamic LDS is sized for it)
	uint32_t lds_words;          // dynamic LDS size in 4-byte words
	uint32_t lds_raster;         // 1: planes are built in LDS bands and stored; 0: zeroed by the host, atomics on HBM
	uint32_t markov_serial;      // testing: expand markov streams with one thread
	uint32_t zbase;              // first slice of this launch (z-chunked launches)
	uint32_t* slice_err;         // [nslices] sticky error bits, cleared by this kernel (it is the first of every decode)
	uint32_t* overflow;          // strip path: its overflow word, cleared here as well (or null)
};

// a value all lanes of the wavefront hold alike, as a scalar
__device__ __forceinline__ uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ uint32_t rd_le_dev(const uint8_t* p, int w) {
	uint32_t v = 0;
	for (int i = 0; i < w; i++) v |= static_cast<uint32_t>(p[i]) << (8 * i);
	return v;
}

constexpr int kCrackBlock = 1024;                 // threads per slice
constexpr int kCrackWaves = kCrackBlock / kWave;
constexpr uint32_t kCrackWords = 8;               // packed words (16 codes each) per thread and tile
constexpr uint32_t kCrackTile = kCrackBlock * kCrackWords * 16u;

// ---- sixteen 2-bit fields per word ("spread" masks carry one flag per field at bit 2k) ----
constexpr uint32_t kLo = 0x55555555u;     // low bit of every field
constexpr uint32_t kEvenF = 0x11111111u;  // low bit of the even fields
constexpr uint32_t kOddF = 0x44444444u;   // low bit of the odd fields
__device__ __forceinline__ uint32_t add_fields(uint32_t a, uint32_t b) {   // per-field sum mod 4
	return (a ^ b) ^ (((a & b) & kLo) << 1);
}
__device__ __forceinline__ uint32_t prefix_fields(uint32_t c) {            // inclusive running sum mod 4
	c = add_fields(c, c << 2);
	c = add_fields(c, c << 4);
	c = add_fields(c, c << 8);
	c = add_fields(c, c << 16);
	return c;
}
__device__ __forceinline__ uint32_t fields_below(int64_t n) {              // spread mask of fields k < n
	return n <= 0 ? 0u : (n >= 16 ? kLo : (((1u << (2 * n)) - 1u) & kLo));
}

// Symbols of one packed word (code positions gw .. gw+15), crackcodes.hpp:547-598 /
// SURVEY.md Appendix D6.  Position g finalises the symbol of code g-1: a move becomes
// 'b'/'t' when code g is its exact reverse and g sits at an odd distance from the last
// position that was not a reverse (runs of reverses alternate control / move).
struct WordSyms {
	uint32_t prevs;     // move of code g-1 per field
	uint32_t ms;        // spread: position emits a move (of kind prevs)
	uint32_t ctl;       // spread: position emits a control symbol
	uint32_t isT;       // spread: ... and it is a 't'
	// spread masks of the emitted moves by direction
	__device__ __forceinline__ uint32_t right() const { return ms & ~(prevs >> 1) & prevs; }
	__device__ __forceinline__ uint32_t left() const { return ms & (prevs >> 1) & prevs; }
	__device__ __forceinline__ uint32_t down() const { return ms & (prevs >> 1) & ~prevs; }
	__device__ __forceinline__ uint32_t up() const { return ms & ~(prevs >> 1) & ~prevs; }
};

// control tables (LDS: 16-bit indices / depths, global: 32-bit)
template <typename IDX, typename DEP>
struct CtlTables {
	uint8_t* kind;
	uint32_t* dx;
	uint32_t* dy;
	DEP* depth;
	IDX* lastT;
	unsigned long long* link;    // low: value, high: index of the 't' it is relative to (or NONE)
	uint32_t* seg_x;
	uint32_t* seg_y;
	DEP* gmin;
};
constexpr uint32_t kLinkNone = 0xFFFFFFFFu;

// Which of the 8 consecutive values at p (16-byte aligned for the 16-bit tables) are < L.
template <typename DEP> __device__ __forceinline__ uint32_t below_mask8(const DEP* p, int32_t L);
template <> __device__ __forceinline__ uint32_t below_mask8<int16_t>(const int16_t* p, int32_t L) {
	const uint4 q = *reinterpret_cast<const uint4*>(p);
	const uint32_t w[4] = { q.x, q.y, q.z, q.w };
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		m |= ((static_cast<int32_t>(w[k] << 16) >> 16) < L ? 1u : 0u) << (2 * k);
		m |= ((static_cast<int32_t>(w[k]) >> 16) < L ? 1u : 0u) << (2 * k + 1);
	}
	return m;
}
template <> __device__ __forceinline__ uint32_t below_mask8<int32_t>(const int32_t* p, int32_t L) {
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) m |= (p[k] < L ? 1u : 0u) << k;
	return m;
}

// largest p < i with depth[p] < L, or -1: a tree of minima with fan-out 8 over depth[]
// (level l entry g = min of the 8 entries 8g..8g+7 of level l-1; level 0 = depth).  One
// 8-wide read per step; up to the first level whose block holds a smaller value, then down
// again: at most 2 log8(N) steps for every lane (the lanes of a wavefront search very
// different distances; a linear scan makes all of them wait for the farthest).
// loff / lcnt: offset into gmin / number of entries of each level.
template <typename DEP>
__device__ __forceinline__ int32_t prev_smaller(const DEP* depth, const DEP* gmin, const uint32_t* loff, const uint32_t* lcnt, int32_t i, int32_t L, uint32_t* iters = nullptr) {
	int32_t level = 0, idx = i - 1;
	for (;;) {
		if (idx < 0) return -1;
		if (iters) (*iters)++;
		const DEP* base = level == 0 ? depth : gmin + loff[level];
		const int32_t blk = idx & ~7;
		const uint32_t m = below_mask8<DEP>(base + blk, L) & ((2u << (idx & 7)) - 1u);
		if (m) {
			const int32_t e = blk + (31 - __clz(m));
			if (level == 0) return e;
			level--;
			const int32_t last = static_cast<int32_t>(lcnt[level]) - 1;
			idx = e * 8 + 7 < last ? e * 8 + 7 : last;
		}
		else { idx = (idx >> 3) - 1; level++; }
	}
}

// Branch matching over the N control symbols of a slice, all threads of the workgroup
// (crackcodes.hpp:771-781, 849-859: the rasteriser's revisit stack; chain segmentation
// by branches_taken, crackcodes.hpp:549-598).
//   depth      stack depth after each symbol ('b' pushes, 't' pops, a 't' on the empty
//              stack ends the chain): a clamped running sum = sum - running minimum.
//   match      the 'b' a 't' returns to is the first symbol after the previous symbol of
//              smaller depth.
//   positions  the vertex after a 't' = vertex of its 'b' = vertex after the last 't'
//              before that 'b' + displacement between them: a forest resolved by pointer
//              jumping on (value, parent) pairs updated in single 8-byte accesses, so no
//              barrier is needed between rounds.
// Output: per segment (stretch between 't's) the vertex offset (seg_x, seg_y) to add to
// the 2-D displacement prefix sums, and the number of valid segments.
template <typename IDX, typename DEP>
__device__ __forceinline__ void match_controls(
	const CtlTables<IDX, DEP>& t, uint32_t N, const uint32_t* nodes, uint32_t n_nodes, uint32_t sxe, uint32_t nverts,
	uint32_t* s_scan, int32_t* s_scanmax, uint32_t* s_first_dead, uint32_t* s_valid_segs, uint32_t* s_loff, uint32_t* s_lcnt, uint32_t& rerr,
	unsigned long long* dg = nullptr
) {
	unsigned long long dg_t = dg ? __builtin_amdgcn_s_memtime() : 0ull;
	auto sub = [&](int slot) { if (dg && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); dg[slot] = now - dg_t; dg_t = now; } };
	constexpr IDX NONE = static_cast<IDX>(~static_cast<IDX>(0));
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (N + kCrackBlock - 1) / kCrackBlock;
	const uint32_t i0 = min(N, tid * per), i1 = min(N, i0 + per);

	int32_t s = 0, mn = INT32_MAX;
	for (uint32_t i = i0; i < i1; i++) {
		s += (t.kind[i] == SYM_T) ? -1 : 1;
		mn = s < mn ? s : mn;
	}
	uint32_t v1[1] = { static_cast<uint32_t>(s) }, t1[1];
	block_excl_add<1, kCrackWaves>(v1, t1, s_scan);
	const int32_t S0 = static_cast<int32_t>(v1[0]);
	int32_t neg_tot;
	const int32_t neg_ex = block_excl_max<kCrackWaves>((i0 < i1) ? -(S0 + mn) : INT32_MIN, neg_tot, s_scanmax);
	const int32_t M0 = -(neg_ex > 0 ? neg_ex : 0);     // min(0, running minimum before my symbols)

	uint32_t nT = 0, nCE = 0;
	int32_t lt = -1;
	{
		int32_t sr = S0, m = M0;
		for (uint32_t i = i0; i < i1; i++) {
			const bool isT = t.kind[i] == SYM_T;
			const int32_t before = sr - m;
			sr += isT ? -1 : 1;
			m = sr < m ? sr : m;
			t.depth[i] = static_cast<DEP>(sr - m);
			if (isT) { nT++; nCE += (before == 0); lt = static_cast<int32_t>(i); }
		}
	}
	uint32_t v2[2] = { nT, nCE }, t2[2];
	block_excl_add<2, kCrackWaves>(v2, t2, s_scan);
	const uint32_t T0 = v2[0], C0 = v2[1], totalT = t2[0];
	int32_t lt_tot;
	const int32_t LT0 = block_excl_max<kCrackWaves>(lt, lt_tot, s_scanmax);
	{
		int32_t cur = LT0 < 0 ? -1 : LT0;
		int32_t sr = S0, m = M0;
		uint32_t c = C0;
		for (uint32_t i = i0; i < i1; i++) {
			const bool isT = t.kind[i] == SYM_T;
			const int32_t before = sr - m;
			sr += isT ? -1 : 1;
			m = sr < m ? sr : m;
			if (isT) {
				cur = static_cast<int32_t>(i);
				if (before == 0) {
					c++;
					if (c >= n_nodes) atomicMin(s_first_dead, i);   // trailing pad codes start here
				}
			}
			t.lastT[i] = cur < 0 ? NONE : static_cast<IDX>(cur);
		}
	}
	__syncthreads();
	sub(8);
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
	}
	__syncthreads();
	for (uint32_t l = 1; s_lcnt[l] != 0; l++) {
		const uint32_t cnt = s_lcnt[l], below = s_lcnt[l - 1];
		const DEP* src = l == 1 ? t.depth : t.gmin + s_loff[l - 1];
		for (uint32_t gi = tid; gi < cnt; gi += kCrackBlock) {
			int32_t mv = INT32_MAX;
			const uint32_t e = min(below, gi * 8u + 8u);
			for (uint32_t i = gi * 8u; i < e; i++) { const int32_t d = static_cast<int32_t>(src[i]); mv = d < mv ? d : mv; }
			t.gmin[s_loff[l] + gi] = static_cast<DEP>(mv);
		}
		__syncthreads();
	}
	__syncthreads();
	sub(9);
	const uint32_t first_dead = *s_first_dead;
	const uint32_t n_eff = min(N, first_dead);

	uint32_t dg_iters = 0;
	// 't's that return to a branch need a search whose length varies a lot and clusters (runs
	// of pops): they go to a worklist (in the not yet used seg_x table) that is then dealt out
	// evenly over the workgroup
	uint32_t* worklist = t.seg_x;
	{
		int32_t sr = S0, m = M0;
		uint32_t c = C0;
		for (uint32_t i = i0; i < i1; i++) {
			const bool isT = t.kind[i] == SYM_T;
			const int32_t before = sr - m;
			sr += isT ? -1 : 1;
			m = sr < m ? sr : m;
			uint32_t val = 0;
			if (isT && before == 0) c++;
			if (isT && i < n_eff) {
				if (before == 0) val = nodes[c];                    // next chain (c < n_nodes: i is before the pad)
				else { worklist[atomicAdd(s_valid_segs, 1u)] = i; continue; }
			}
			t.link[i] = (static_cast<unsigned long long>(kLinkNone) << 32) | val;
		}
	}
	__syncthreads();
	const uint32_t n_work = *s_valid_segs - 1u;      // the counter starts at 1 (its later meaning: valid segments)
	for (uint32_t e = tid; e < n_work; e += kCrackBlock) {
		const uint32_t i = worklist[1u + e];
		const int32_t before = static_cast<int32_t>(t.depth[i]) + 1;      // a 't' that pops: depth after = depth before - 1
		uint32_t val, ptr = kLinkNone;
		const uint32_t j = static_cast<uint32_t>(prev_smaller<DEP>(t.depth, t.gmin, s_loff, s_lcnt, static_cast<int32_t>(i), before, dg ? &dg_iters : nullptr) + 1);
		const uint32_t pos_j = t.dx[j] + sxe * t.dy[j];
		const IDX tp = t.lastT[j];
		if (tp == NONE) val = nodes[0] + pos_j;
		else { val = pos_j - (t.dx[tp] + sxe * t.dy[tp]); ptr = static_cast<uint32_t>(tp); }
		t.link[i] = (static_cast<unsigned long long>(ptr) << 32) | val;
	}
	__syncthreads();
	if (tid == 0) *s_valid_segs = 1u;
	if (dg) { atomicAdd(dg + 12, static_cast<unsigned long long>(dg_iters)); atomicMax(dg + 13, static_cast<unsigned long long>(dg_iters)); }
	__syncthreads();
	sub(10);
	for (uint32_t i = i0; i < i1; i++) {
		if (t.kind[i] != SYM_T || i >= n_eff) continue;
		// relaxed atomics, not volatile: hipcc keeps volatile accesses on flat pointers (a flat
		// instruction and a wait for all memory traffic per access)
		unsigned long long* lk = t.link;
		unsigned long long me = __hip_atomic_load(lk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		for (uint32_t guard = 0; (me >> 32) != kLinkNone && guard <= N; guard++) {
			const unsigned long long other = __hip_atomic_load(lk + static_cast<uint32_t>(me >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			me = (other & 0xFFFFFFFF00000000ull) | static_cast<uint32_t>(static_cast<uint32_t>(me) + static_cast<uint32_t>(other));
			__hip_atomic_store(lk + i, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	__syncthreads();
	sub(11);
	{
		uint32_t tc = T0;
		for (uint32_t i = i0; i < i1; i++) {
			if (t.kind[i] != SYM_T) continue;
			if (i < n_eff) {
				uint32_t A = static_cast<uint32_t>(t.link[i]);
				if (A >= nverts) { rerr |= ERR_RANGE; A = 0; }
				const uint32_t ay = A / sxe;
				t.seg_x[tc + 1] = (A - ay * sxe) - t.dx[i];
				t.seg_y[tc + 1] = ay - t.dy[i];
			}
			if (i == first_dead) *s_valid_segs = tc + 1u;
			tc++;
		}
	}
	if (tid == 0) {
		const uint32_t n0 = nodes[0];
		t.seg_x[0] = n0 % sxe;
		t.seg_y[0] = n0 / sxe;
		if (first_dead >= N) *s_valid_segs = totalT + 1u;
	}
	__syncthreads();
}

// LDS carving for a capacity of n control symbols: seg_x | seg_y | link | dx | dy | depth | lastT | gmin | kind.
// In pass 1 only the segment offsets are still needed; everything behind them is reused as
// the raster band buffer.
static inline size_t crack_lds_seg_bytes(uint32_t n) { return (static_cast<size_t>(n) + 2) * 4 * 2; }
static inline size_t crack_lds_bytes(uint32_t n) {
	return crack_lds_seg_bytes(n) + static_cast<size_t>(n) * 8 + static_cast<size_t>(n) * 4 * 2
		+ static_cast<size_t>(n) * 2 * 2 + (static_cast<size_t>(n) / 7 + 48) * 2 + n + 16;
}

struct TileCarry {
	uint32_t sum = 0;        // running mod-4 sum of difference codes
	uint32_t move = 0;       // move of the last code of the previous tile
	uint32_t ctrl = 0;       // was that code the second half of a control pair
	int32_t lf = -1;         // last position whose `reverse-of-previous` test was false
	uint32_t a = 0, dx = 0, dy = 0;   // a: controls (pass 0) / 't's (pass 1); displacement
};

// Symbols of one tile (kCrackTile code positions from `tile`), all threads of the
// workgroup: thread t derives the symbols of its 128 positions into ws[] and gets the
// exclusive counts before its first position (o_a, o_dx, o_dy); the carries advance
// to the next tile.  The last barrier inside is behind every use of the LDS scratch.
template <bool COUNT_T, int BLOCK = kCrackBlock, uint32_t WORDS = kCrackWords>
__device__ __forceinline__ void tile_symbols(
	const uint32_t* __restrict__ words, uint32_t wshift, uint32_t n_codes, uint32_t span, uint32_t tile, TileCarry& c,
	WordSyms (&ws)[WORDS], uint32_t& o_a, uint32_t& o_dx, uint32_t& o_dy,
	uint32_t* s_scan, int32_t* s_scanmax, uint8_t* s_last_move, uint8_t* s_last_ctrl, unsigned long long* dg = nullptr
) {
	constexpr int NW = BLOCK / kWave;
	const uint32_t tid = threadIdx.x;
	unsigned long long dg_t = (kTuning && dg) ? __builtin_amdgcn_s_memtime() : 0ull;
	auto sub = [&](int slot) { if (kTuning && dg && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); dg[slot] += now - dg_t; dg_t = now; } };
	// `span` positions per thread (a multiple of 16, at most WORDS * 16): a slice with fewer codes
	// than a full tile spreads them over all threads instead of filling the first ones with eight
	// words each; a thread's words past its span are empty
	const uint32_t g0 = tile + tid * span;
	const int64_t lim = min(static_cast<int64_t>(n_codes), static_cast<int64_t>(g0) + span);            // codes of this thread end here
	const int64_t lim_ev = min(static_cast<int64_t>(n_codes) + 1, static_cast<int64_t>(g0) + span);      // positions
	const uint32_t last_word = span / 16u - 1u;
	// -- load, running sums mod 4
	uint32_t mv[WORDS];
	uint32_t tsum = 0;
	{
		uint32_t q[WORDS + 1];
		const uint32_t w0 = g0 / 16u;
#pragma unroll
		for (uint32_t j = 0; j <= WORDS; j++) {
			// word j is needed when any of its codes exists; the shifted read also takes the low bytes of word j+1
			// (issuing all nine loads unconditionally, with clamped indices, was measured slower: 0.322 against 0.313 ms)
			const bool need = (static_cast<int64_t>(g0) + 16u * j < lim) || (j > 0 && wshift && static_cast<int64_t>(g0) + 16u * (j - 1u) < lim);
			q[j] = need ? words[w0 + j] : 0u;
		}
#pragma unroll
		for (uint32_t j = 0; j < WORDS; j++) {
			uint32_t cw = wshift ? __funnelshift_r(q[j], q[j + 1], wshift) : q[j];
			const uint32_t fm = fields_below(lim - static_cast<int64_t>(g0 + 16u * j));
			cw &= fm | (fm << 1);
			cw = prefix_fields(cw);
			mv[j] = add_fields(cw, tsum * kLo);
			tsum = mv[j] >> 30;
		}
	}
	sub(10);
	uint32_t v1[1] = { tsum }, t1[1];
	block_excl_add<1, NW>(v1, t1, s_scan);
	sub(11);
	const uint32_t base_sum = ((c.sum + v1[0]) & 3u) * kLo;
#pragma unroll
	for (uint32_t j = 0; j < WORDS; j++) mv[j] = add_fields(mv[j], base_sum);
	s_last_move[tid] = static_cast<uint8_t>(mv[WORDS - 1] >> 30);
	__syncthreads();
	const uint32_t prev_move = tid ? s_last_move[tid - 1] : c.move;
	const uint32_t tile_last_move = s_last_move[BLOCK - 1];

	// -- r: code g is the exact reverse of code g-1
	uint32_t r[WORDS];
	int32_t lf = INT32_MIN;
#pragma unroll
	for (uint32_t j = 0; j < WORDS; j++) {
		const uint32_t gw = g0 + 16u * j;
		const uint32_t prevs = (mv[j] << 2) | (j ? (mv[j - 1] >> 30) : prev_move);
		const uint32_t x = mv[j] ^ prevs;
		const uint32_t in_span = fields_below(lim - static_cast<int64_t>(gw));
		uint32_t rv = in_span;
		if (gw == 0) rv &= ~1u;
		r[j] = (x >> 1) & ~x & rv;
		const uint32_t nr = ~r[j] & in_span;
		if (nr) lf = static_cast<int32_t>(gw + ((31u - __clz(nr)) >> 1));
	}
	int32_t lf_tot;
	int32_t lf_in = block_excl_max<NW>(lf, lf_tot, s_scanmax);
	sub(12);
	if (lf_in < c.lf) lf_in = c.lf;
	// -- ctrl: within a run of reverses, the positions at an odd distance from the last non-reverse
	uint32_t ctrl[WORDS];
#pragma unroll
	for (uint32_t j = 0; j < WORDS; j++) {
		const uint32_t gw = g0 + 16u * j;
		const uint32_t rj = r[j];
		uint32_t es = rj & ~(rj << 2) & kEvenF;           // runs starting on an even field
		if ((rj & 1u) && !((gw - static_cast<uint32_t>(lf_in)) & 1u)) es &= ~1u;   // run continues from before and field 0 is not a control
		const uint32_t r2 = rj | (rj << 1);
		const uint32_t inA = rj & ~(r2 + es);             // fields of the runs counted from an even field
		ctrl[j] = (inA & kEvenF) | (rj & ~inA & kOddF);
		const uint32_t nr = ~rj & fields_below(lim - static_cast<int64_t>(gw));
		if (nr) lf_in = static_cast<int32_t>(gw + ((31u - __clz(nr)) >> 1));
	}
	{
		uint32_t lc = ctrl[WORDS - 1];
#pragma unroll
		for (uint32_t j = 0; j + 1 < WORDS; j++) if (j == last_word) lc = ctrl[j];
		s_last_ctrl[tid] = static_cast<uint8_t>((lc >> 30) & 1u);
	}
	__syncthreads();
	const uint32_t prev_ctrl = tid ? s_last_ctrl[tid - 1] : c.ctrl;
	const uint32_t tile_last_ctrl = s_last_ctrl[BLOCK - 1];
	sub(13);

	// -- events: position g emits the symbol of code g-1 unless g-1 was a control half
	uint32_t n_a = 0, ddx = 0, ddy = 0;
#pragma unroll
	for (uint32_t j = 0; j < WORDS; j++) {
		const uint32_t gw = g0 + 16u * j;
		WordSyms& w = ws[j];
		w.prevs = (mv[j] << 2) | (j ? (mv[j - 1] >> 30) : prev_move);
		const uint32_t pc = ((ctrl[j] << 2) | (j ? ((ctrl[j - 1] >> 30) & 1u) : prev_ctrl)) & kLo;
		uint32_t ev = fields_below(lim_ev - static_cast<int64_t>(gw));
		if (gw == 0) ev &= ~1u;
		const uint32_t emit = ev & ~pc;
		w.ctl = emit & ctrl[j];
		w.isT = w.ctl & ~(mv[j] ^ (mv[j] >> 1));
		w.ms = emit & ~ctrl[j];
		n_a += __popc(COUNT_T ? w.isT : w.ctl);
		ddx += __popc(w.right()) - __popc(w.left());
		ddy += __popc(w.down()) - __popc(w.up());
	}
	uint32_t v3[3] = { n_a, ddx, ddy }, t3[3];
	block_excl_add<3, NW>(v3, t3, s_scan);
	sub(14);
	o_a = c.a + v3[0]; o_dx = c.dx + v3[1]; o_dy = c.dy + v3[2];

	// (the carries are the same in every lane: scalar registers)
	c.sum = uni((c.sum + t1[0]) & 3u);
	c.move = uni(tile_last_move);
	c.ctrl = uni(tile_last_ctrl);
	c.lf = static_cast<int32_t>(uni(static_cast<uint32_t>(lf_tot > c.lf ? lf_tot : c.lf)));
	c.a = uni(c.a + t3[0]); c.dx = uni(c.dx + t3[1]); c.dy = uni(c.dy + t3[2]);
}

// Rasterises the moves of one thread's 128 positions (crackcodes.hpp:706-862) straight into the
// planes in HBM (zeroed by the host): the fallback when not even one plane row fits the LDS band
// buffer.  Vertical moves cross planeV, horizontal moves cross planeH; consecutive moves that
// land in one plane word are OR-ed into it once (an atomic on HBM is expensive).
__device__ __forceinline__ void raster_moves_hbm(
	const WordSyms (&ws)[kCrackWords], uint32_t o_t, uint32_t o_dx, uint32_t o_dy, uint32_t valid_segs,
	const uint32_t* seg_x, const uint32_t* seg_y, uint32_t sx, uint32_t sy, uint32_t row_words,
	uint32_t* pv, uint32_t* ph, uint32_t& rerr
) {
	uint32_t bx = 0, by = 0;
	bool act = o_t < valid_segs;
	if (act) { bx = seg_x[o_t]; by = seg_y[o_t]; }
	uint32_t x = bx + o_dx, y = by + o_dy;
	uint32_t cur_bits = 0;
	uint32_t* cur_word = nullptr;
#pragma unroll
	for (uint32_t j = 0; j < kCrackWords; j++) {
		const WordSyms& w = ws[j];
		for (uint32_t m = w.ms | w.isT; m; m &= m - 1u) {
			const uint32_t b = __ffs(m) - 1u;
			if ((w.isT >> b) & 1u) {
				o_t++;
				act = o_t < valid_segs;
				if (act) {
					const uint32_t nbx = seg_x[o_t], nby = seg_y[o_t];
					x += nbx - bx; y += nby - by;
					bx = nbx; by = nby;
				}
				continue;
			}
			const uint32_t k = (w.prevs >> b) & 3u;
			const uint32_t isU = (k == SYM_U), isL = (k == SYM_L);
			const uint32_t row = y - isU, col = x - isL;
			const uint32_t nx = x + (k == SYM_R) - isL, ny = y + (k == SYM_D) - isU;
			if (act) {
				if (x > sx || y > sy || nx > sx || ny > sy) rerr |= ERR_RANGE;
				else {
					const bool horiz = k & 1u;
					// a move along the outer border crosses no crack of the planes
					const bool ok = horiz ? (row - 1u < sy - 1u && col < sx) : (col - 1u < sx - 1u && row < sy);
					if (ok) {
						uint32_t* word = (horiz ? ph : pv) + static_cast<uint64_t>(row) * row_words + (col >> 5);
						if (word != cur_word) {
							if (cur_word) atomicOr(cur_word, cur_bits);
							cur_word = word;
							cur_bits = 0;
						}
						cur_bits |= 1u << (col & 31u);
					}
				}
			}
			x = nx; y = ny;
		}
	}
	if (cur_word) atomicOr(cur_word, cur_bits);
}

// The LDS band buffer variant, one plane per pass: HORIZ rasterises the horizontal moves
// into rows [band_y0, band_y0 + band_rows) of plane H, otherwise the vertical moves into plane V.
// Written for the instruction count (the loop body is what k_decode_cracks spends most of its
// time in): a pass only visits the moves of its orientation; the vertex of a move is the
// segment's offset plus the displacement before it, which inside a word is four popcounts, so
// the moves of the other orientation never have to be stepped through.  One ds_or per move:
// vertical moves change the row every time and hardly ever share a plane word.
template <bool HORIZ, bool SKIP>
__device__ __forceinline__ void raster_plane(
	const WordSyms (&ws)[kCrackWords], uint32_t o_t, uint32_t o_dx, uint32_t o_dy, uint32_t valid_segs,
	const uint32_t* seg_x, const uint32_t* seg_y, uint32_t sx, uint32_t sy, uint32_t row_words,
	uint32_t* band, uint32_t band_y0, uint32_t band_rows, uint32_t& rerr
) {
	uint32_t act = o_t < valid_segs ? 1u : 0u;
	uint32_t dx = o_dx, dy = o_dy;        // displacement of the whole stream before the current word
	// vertex before the current word = segment offset + that displacement.  The offsets are added in
	// where they are loaded (start and jumps), so that the wait for those loads sits there and not in
	// front of every move, where it would also wait for the previous move's ds_or.
	uint32_t xb = dx, yb = dy;
	if (act) { xb += seg_x[o_t]; yb += seg_y[o_t]; }
	uint32_t bad = 0;
#pragma unroll
	for (uint32_t j = 0; j < kCrackWords; j++) {
		const WordSyms& w = ws[j];
		const uint32_t mR = w.right(), mL = w.left(), mD = w.down(), mU = w.up(), isT = w.isT;
		const uint32_t nr = __popc(mR), nl = __popc(mL), nd = __popc(mD), nu = __popc(mU);
		bool skip = isT == 0u && !act;        // a cut-off segment stays silent until the next jump
		if (SKIP && isT == 0u && act) {
			// (many-band slices) no jump inside this word: its moves stay within a box known from
			// the four popcounts.  A box inside the grid (no range error possible) that misses the
			// band, with the one-row reach of vertical moves, is skipped whole.
			const bool inside = xb >= nl && xb + nr <= sx && yb >= nu && yb + nd <= sy;
			skip = inside && (yb + nd + 1u < band_y0 || yb - nu > band_y0 + band_rows);
		}
		if (!skip) {
			for (uint32_t m = (HORIZ ? (mR | mL) : (mD | mU)) | isT; m; m &= m - 1u) {
				const uint32_t b = __ffs(m) - 1u;
				if ((isT >> b) & 1u) {
					o_t++;
					act = o_t < valid_segs ? 1u : 0u;
					if (act) { xb = seg_x[o_t] + dx; yb = seg_y[o_t] + dy; }
					continue;
				}
				const uint32_t below = (1u << b) - 1u;
				const uint32_t x = xb + __popc(mR & below) - __popc(mL & below);
				const uint32_t y = yb + __popc(mD & below) - __popc(mU & below);
				// the crossed crack sits at the smaller vertex; a move along the outer border crosses no
				// crack of the planes (never in a stream of the reference's encoder), a move that leaves
				// the vertex grid is an error: both fail `ok` and are told apart on the side
				uint32_t ok, row, col, neg;
				if (HORIZ) {
					neg = (mL >> b) & 1u;                 // left
					col = x - neg; row = y;
					ok = static_cast<uint32_t>(row - 1u < sy - 1u) & static_cast<uint32_t>(col < sx);
				}
				else {
					neg = (mU >> b) & 1u;                 // up
					col = x; row = y - neg;
					ok = static_cast<uint32_t>(col - 1u < sx - 1u) & static_cast<uint32_t>(row < sy);
				}
				if (act & ok) {
					const uint32_t rel = row - band_y0;
					if (rel < band_rows) atomicOr(band + rel * row_words + (col >> 5), 1u << (col & 31u));
				}
				else if (act) {
					const uint32_t far = HORIZ ? x + 1u - 2u * neg : y + 1u - 2u * neg;      // the vertex moved to
					const uint32_t in_range = HORIZ
						? static_cast<uint32_t>(max(x, far) <= sx) & static_cast<uint32_t>(y <= sy)
						: static_cast<uint32_t>(x <= sx) & static_cast<uint32_t>(max(y, far) <= sy);
					bad |= in_range ^ 1u;
				}
			}
		}
		dx += nr - nl; dy += nd - nu;
		xb += nr - nl; yb += nd - nu;
	}
	if (bad) rerr |= ERR_RANGE;
}

// ------------------------------------------------------------------------------
// markov bitstream -> difference codes (markov.hpp:268-313), all threads of the workgroup
// ------------------------------------------------------------------------------
// The stream is a raw 2-bit start code followed by codes `0`, `10`, `110`, `111` (LSB first)
// for the rank of each difference code under the order-N model; rank -> code needs the
// previous N codes.  Two serial dependencies, both broken here:
//   1. where codes start: a 3-state automaton over the bits (at a code start / after `1` /
//      after `11`); every thread simulates its bit range from each state, a block scan
//      composes the maps, then the ranks are written out by code index;
//   2. the context recurrence: every thread decodes its chunk of codes speculatively,
//      warming up over the kMarkovWarm codes before it from an arbitrary context (the
//      context is the last N codes, so once N decoded codes are right the state is right);
//      chunks whose assumed start context differs from their predecessor's end context
//      are re-decoded, round by round, until all agree (at worst one chunk per round:
//      the serial order; typically one or two rounds).
struct BitMap3 { uint32_t e, c0, c1, c2; };      // end state per start state (2 bits each), code starts per start state
__device__ __forceinline__ BitMap3 bitmap3_compose(const BitMap3& f, const BitMap3& g) {      // f first, then g
	const uint32_t f0 = f.e & 3u, f1 = (f.e >> 2) & 3u, f2 = (f.e >> 4) & 3u;
	BitMap3 r;
	r.e = ((g.e >> (2u * f0)) & 3u) | (((g.e >> (2u * f1)) & 3u) << 2) | (((g.e >> (2u * f2)) & 3u) << 4);
	// (picked with masks: written as `f0 == 0 ? g.c0 : ...` hipcc turns the choice between fields into an index into a copy of g
	// in scratch memory — a store and a dependent load, a trip to memory, per composition of the scans below)
	auto pick = [&](uint32_t st) -> uint32_t {
		const uint32_t m1 = 0u - static_cast<uint32_t>(st == 1u), m2 = 0u - static_cast<uint32_t>(st >= 2u);
		return (g.c0 & ~(m1 | m2)) | (g.c1 & m1) | (g.c2 & m2);
	};
	r.c0 = f.c0 + pick(f0);
	r.c1 = f.c1 + pick(f1);
	r.c2 = f.c2 + pick(f2);
	return r;
}
constexpr uint32_t kMarkovWarm = 192;      // (64 before round 5: a quarter of C2's chunks then started from a wrong context and 2.1 repair rounds followed; 192: 3 % and 1.4)

// lds: [payload words][rank words][model rows or nothing][2 x kCrackBlock context words]
__device__ __forceinline__ uint32_t markov_lds_need(uint32_t nbytes, uint32_t cap, int order, uint32_t budget, bool& model_in_lds, uint32_t block = kCrackBlock) {
	const uint32_t pay = ((nbytes + 3u) / 4u + 2u) * 4u;
	const uint32_t rnk = (cap / 16u + 2u) * 4u;
	const uint32_t fix = 2u * block * 4u + 768u * 4u;      // context words, byte table
	const uint64_t mdl = 4ull << (2 * order);
	model_in_lds = order <= 8 && pay + rnk + fix + mdl <= budget;
	return pay + rnk + fix + (model_in_lds ? static_cast<uint32_t>(mdl) : 0u);
}
// the same with payload and ranks in the global scratch: only the model and the context words
__device__ __forceinline__ bool markov_model_fits_alone(int order, uint32_t budget, uint32_t block = kCrackBlock) {
	return order <= 8 && (4ull << (2 * order)) + 2ull * block * 4ull + 768ull * 4ull <= budget;
}

// GLOBAL: the payload copy and the ranks live in a global scratch area (slices too big for the
// LDS); those words are written by some threads and read by others of the workgroup, so they are
// read with L1-bypassing loads.
template <bool GLOBAL, int BLOCK = kCrackBlock>
__device__ __forceinline__ void markov_expand_parallel(
	const uint8_t* __restrict__ s, uint32_t nbytes, int order, const uint8_t* __restrict__ model_g, uint32_t cap,
	uint32_t* __restrict__ upacked, uint32_t* lds, uint32_t* gscratch, bool model_in_lds, uint32_t* s_scan, uint32_t* s_total,
	uint32_t& ncodes_out, uint32_t& err_out, uint32_t lds_words = 0
) {
	const uint32_t tid = threadIdx.x;
	const uint32_t pay_words = (nbytes + 3u) / 4u + 2u;
	const uint32_t rank_words = cap / 16u + 2u;
	uint32_t* pay = GLOBAL ? gscratch : lds;
	uint32_t* ranks = pay + pay_words;
	uint32_t* rows_lds = GLOBAL ? lds : ranks + rank_words;
	const uint32_t n_rows = 1u << (2 * order);
	uint32_t* ctx_in = rows_lds + (model_in_lds ? n_rows : 0u);
	uint32_t* ctx_end = ctx_in + BLOCK;
	const uint32_t* rows = model_in_lds ? rows_lds : reinterpret_cast<const uint32_t*>(model_g);

	// ---- stage the payload (zero padded), clear the ranks, stage the model
	// (whole words with one unaligned load each, eight of a thread's words requested side by side: byte loads behind a
	// bounds test of their own made every byte a trip to memory — a third of this function's time at C2)
	{
		const uint32_t full = nbytes / 4u;      // words that lie wholly inside the payload
		constexpr uint32_t kStage = 8;
		for (uint32_t w0 = tid; w0 < pay_words; w0 += BLOCK * kStage) {
			uint32_t v[kStage];
#pragma unroll
			for (uint32_t q = 0; q < kStage; q++) {
				const uint32_t w = w0 + q * BLOCK;
				uint32_t x = 0;
				if (full) __builtin_memcpy(&x, s + 4ull * (w < full ? w : 0u), 4);      // (uniform: a payload of fewer than four bytes is read byte by byte below)
				v[q] = w < full ? x : 0u;
			}
#pragma unroll
			for (uint32_t q = 0; q < kStage; q++) {
				const uint32_t w = w0 + q * BLOCK;
				if (w >= pay_words) continue;
				uint32_t x = v[q];
				if (w == full) for (uint32_t b = 0; b < 4u; b++) { const uint32_t i = w * 4u + b; if (i < nbytes) x |= static_cast<uint32_t>(s[i]) << (8u * b); }      // the tail bytes
				pay[w] = x;
			}
		}
	}
	for (uint32_t w = tid; w < rank_words; w += BLOCK) ranks[w] = 0;
	if (model_in_lds) for (uint32_t r = tid; r < n_rows; r += BLOCK) rows_lds[r] = reinterpret_cast<const uint32_t*>(model_g)[r];
	if (GLOBAL) __threadfence();      // the zeroes must be in L2 before another wavefront's atomicOr lands there
	__syncthreads();
	auto ldw = [&](const uint32_t* p) -> uint32_t { return GLOBAL ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };

	// ---- 1. + 2. code boundaries and ranks, a BYTE of the bit stream per step (round 5; a bit per step before: 0.33 of
	// C2's 0.42 ms with a markov model).  The codes are a prefix code over the bits behind the two-bit start code
	// (rank 0 = "0", 1 = "10", 2 = "110", 3 = "111", low bit first, markov.hpp:268-313): with the state = ones seen of
	// the code in progress (0, 1, 2), a byte maps every state to an end state, the number of codes it COMPLETES
	// (<= 8) and their ranks — one 32-bit table entry (s_tab, built here: 3 x 256).  The stream behind the start code
	// is read through a two-bit funnel shift, so that bytes are bytes; its last byte holds six bits and is walked bit
	// by bit by the last thread, which also closes a code left open at the end the way the reference's look-ahead does
	// (it reads zeros: the open code's rank is its state).
	uint32_t* s_tab = ctx_end + BLOCK;      // [768]: (state << 8 | byte) -> end state | completed << 2 | ranks << 6
	for (uint32_t e = tid; e < 768u; e += BLOCK) {
		uint32_t st = e >> 8, cnt = 0, rk = 0;
		for (uint32_t i = 0; i < 8u; i++) {
			const uint32_t b = (e >> i) & 1u;
			uint32_t rank = 4u;
			if (st == 2u) { rank = 2u + b; st = 0u; }
			else if (b) st++;
			else { rank = st; st = 0u; }
			if (rank != 4u) { rk |= rank << (2u * cnt); cnt++; }
		}
		s_tab[e] = st | (cnt << 2) | (rk << 6);
	}
	__syncthreads();
	const uint32_t nb_full = nbytes - 1u;      // whole bytes of the shifted stream (8 nbytes - 2 bits: the last byte has six)
	const uint32_t QB = (nb_full + BLOCK - 1u) / BLOCK;
	const uint32_t b_lo = min(nb_full, tid * QB), b_hi = min(nb_full, b_lo + QB);
	auto shifted_word = [&](uint32_t w) -> uint32_t { return __funnelshift_r(ldw(pay + w), ldw(pay + w + 1u), 2u); };      // shifted bytes 4 w .. 4 w + 3
	BitMap3 m;
	{
		uint32_t st0 = 0, st1 = 1, st2 = 2, n0 = 0, n1 = 0, n2 = 0;
		uint32_t wc = b_lo < b_hi ? shifted_word(b_lo >> 2) : 0u;
		for (uint32_t j = b_lo; j < b_hi; j++) {
			if ((j & 3u) == 0u && j != b_lo) wc = shifted_word(j >> 2);
			const uint32_t by = (wc >> (8u * (j & 3u))) & 255u;
			const uint32_t e0 = s_tab[(st0 << 8) | by], e1 = s_tab[(st1 << 8) | by], e2 = s_tab[(st2 << 8) | by];
			st0 = e0 & 3u; n0 += (e0 >> 2) & 15u;
			st1 = e1 & 3u; n1 += (e1 >> 2) & 15u;
			st2 = e2 & 3u; n2 += (e2 >> 2) & 15u;
		}
		if (tid == BLOCK - 1) {      // the six bits of the last byte
			const uint32_t by = (shifted_word(nb_full >> 2) >> (8u * (nb_full & 3u))) & 63u;
			uint32_t st[3] = { st0, st1, st2 }, nn[3] = { n0, n1, n2 };
#pragma unroll
			for (uint32_t q = 0; q < 3u; q++) {      // (unrolled: st[q] under a run-time q is an array in scratch memory)
#pragma unroll
				for (uint32_t i = 0; i < 6u; i++) {
					const uint32_t b = (by >> i) & 1u;
					if (st[q] == 2u) { nn[q]++; st[q] = 0u; }
					else if (b) st[q]++;
					else { nn[q]++; st[q] = 0u; }
				}
			}
			st0 = st[0]; st1 = st[1]; st2 = st[2]; n0 = nn[0]; n1 = nn[1]; n2 = nn[2];
		}
		m.e = st0 | (st1 << 2) | (st2 << 4); m.c0 = n0; m.c1 = n1; m.c2 = n2;
	}
	// inclusive scan of the maps inside the wavefront, wave totals through LDS
	const uint32_t lane = tid & (kWave - 1), wave = tid >> 6;
	BitMap3 inc = m;
	for (uint32_t d = 1; d < static_cast<uint32_t>(kWave); d <<= 1) {
		BitMap3 o;
		o.e = __shfl_up(inc.e, d, kWave); o.c0 = __shfl_up(inc.c0, d, kWave); o.c1 = __shfl_up(inc.c1, d, kWave); o.c2 = __shfl_up(inc.c2, d, kWave);
		if (lane >= d) inc = bitmap3_compose(o, inc);
	}
	if (lane == kWave - 1) { s_scan[wave * 4 + 0] = inc.e; s_scan[wave * 4 + 1] = inc.c0; s_scan[wave * 4 + 2] = inc.c1; s_scan[wave * 4 + 3] = inc.c2; }
	BitMap3 exc;
	exc.e = __shfl_up(inc.e, 1, kWave); exc.c0 = __shfl_up(inc.c0, 1, kWave); exc.c1 = __shfl_up(inc.c1, 1, kWave); exc.c2 = __shfl_up(inc.c2, 1, kWave);
	if (lane == 0) { exc.e = 36u; exc.c0 = exc.c1 = exc.c2 = 0; }      // identity
	__syncthreads();
	BitMap3 pre = { 36u, 0, 0, 0 };
	for (uint32_t w = 0; w < wave; w++) {
		BitMap3 t = { s_scan[w * 4 + 0], s_scan[w * 4 + 1], s_scan[w * 4 + 2], s_scan[w * 4 + 3] };
		pre = bitmap3_compose(pre, t);
	}
	const BitMap3 before = bitmap3_compose(pre, exc);
	if (tid == BLOCK - 1) {
		const BitMap3 all = bitmap3_compose(before, m);
		*s_total = 1u + all.c0 + ((all.e & 3u) ? 1u : 0u);      // the start code, the completed codes, a code left open at the end
	}
	// ---- ranks by code index: the codes a thread's bytes complete are consecutive; only the first and the last rank word
	// it touches can be shared with a neighbour (atomicOr), the ones between are its own
	{
		uint32_t st = before.e & 3u;
		uint32_t k = 1u + before.c0;          // index of the next code to complete
		unsigned long long acc = 0;           // rank bits from word (k >> 4) on
		bool first_word = true;
		auto put = [&](uint32_t wi, uint32_t bits, bool shared) {
			if (!bits || wi >= rank_words) return;
			if (shared) atomicOr(ranks + wi, bits);
			else if (GLOBAL) __hip_atomic_store(ranks + wi, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else ranks[wi] = bits;
		};
		auto add = [&](uint32_t cnt, uint32_t rk) {      // cnt codes (ranks rk, two bits each) complete
			const uint32_t kc = k < cap ? k : cap;        // (codes past the capacity are dropped: the caller reports ERR_CAPACITY)
			acc |= static_cast<unsigned long long>(k < cap ? rk : 0u) << (2u * (kc & 15u));
			const uint32_t k2 = k + cnt;
			if ((k2 >> 4) != (k >> 4)) {
				put(k >> 4, static_cast<uint32_t>(acc), first_word);
				first_word = false;
				acc >>= 32;
			}
			k = k2;
		};
		uint32_t wc = b_lo < b_hi ? shifted_word(b_lo >> 2) : 0u;
		for (uint32_t j = b_lo; j < b_hi; j++) {
			if ((j & 3u) == 0u && j != b_lo) wc = shifted_word(j >> 2);
			const uint32_t e = s_tab[(st << 8) | ((wc >> (8u * (j & 3u))) & 255u)];
			st = e & 3u;
			add((e >> 2) & 15u, (e >> 6) & 0xFFFFu);
		}
		if (tid == BLOCK - 1) {
			const uint32_t by = (shifted_word(nb_full >> 2) >> (8u * (nb_full & 3u))) & 63u;
			for (uint32_t i = 0; i < 6u; i++) {
				const uint32_t b = (by >> i) & 1u;
				if (st == 2u) { add(1u, 2u + b); st = 0u; }
				else if (b) st++;
				else { add(1u, st); st = 0u; }
			}
			if (st) add(1u, st);      // the code left open reads zeros behind the payload
		}
		put(k >> 4, static_cast<uint32_t>(acc), true);
	}
	if (GLOBAL) __threadfence();
	__syncthreads();
	uint32_t n = *s_total;
	if (n > cap) { n = cap; err_out |= ERR_CAPACITY; }
	ncodes_out = n;

	// GLOBAL: the ranks as they stand (2 bits per code that exists, not per code the slice's bytes could hold) often fit
	// the LDS behind the model: the chunk decoders below then read them there instead of through a chain of ~60
	// dependent L1-bypassing loads per pass (2048 x 2048 slices: 27 k words)
	const uint32_t* ranks_src = ranks;
	bool ranks_in_lds = false;
	if (GLOBAL) {
		uint32_t* stage = s_tab + 768u;
		const uint32_t used = static_cast<uint32_t>(stage - lds), need = (n >> 4) + 2u;
		if (lds_words > used && need <= lds_words - used) {      // uniform
			for (uint32_t w = tid; w < need; w += BLOCK) stage[w] = ldw(ranks + w);
			__syncthreads();
			ranks_src = stage;
			ranks_in_lds = true;
		}
	}
	auto ldr = [&](uint32_t w) -> uint32_t { return (GLOBAL && !ranks_in_lds) ? __hip_atomic_load(ranks_src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ranks_src[w]; };
	// ---- 3. the context recurrence, speculatively per chunk
	const int shift = 2 * (order - 1);
	const uint32_t start = ldw(pay) & 3u;
	uint32_t C = (n + BLOCK - 1u) / BLOCK;
	C = max(16u, (C + 15u) & ~15u);
	const uint32_t nchunks = (n + C - 1u) / C;
	const uint32_t k0 = tid * C, k1 = min(n, k0 + C);
	const bool have = tid < nchunks;
	// codes max(k0, 1) .. k1 - 1 from context ctx, a WORD of 16 at a time (chunks start on words; only the slice's first
	// and last word can be partial): writes the words, returns the end context.  REPAIR (a chunk whose start context
	// turned out different): the context is a shift register of the last `order` (< 16) decoded codes, so once a whole
	// word comes out as it did before, everything behind it does too — the chunk is left there, kMarkovSame returned
	// (its end context stands).  A repaired chunk typically rejoins its old trajectory inside one or two words.
	constexpr uint32_t kMarkovSame = 0xFFFFFFFFu;
	auto decode_chunk = [&](uint32_t ctx, bool repair) -> uint32_t {
		const uint32_t w0 = k0 >> 4, w1 = (k1 + 15u) >> 4;
		for (uint32_t w = w0; w < w1; w++) {
			const uint32_t rw = ldr(w);
			const uint32_t lo = w == 0u ? 1u : 0u, hi = min(16u, k1 - 16u * w);      // (code 0 of the slice is the raw start code)
			uint32_t word = w == 0u ? start : 0u;
			if (lo == 0u && hi == 16u) {
#pragma unroll
				for (uint32_t i = 0; i < 16u; i++) {
					const uint32_t v = (rows[ctx] >> (8u * ((rw >> (2u * i)) & 3u))) & 3u;
					ctx = (ctx >> 2) + (v << shift);
					word |= v << (2u * i);
				}
			}
			else {
				for (uint32_t i = lo; i < hi; i++) {
					const uint32_t v = (rows[ctx] >> (8u * ((rw >> (2u * i)) & 3u))) & 3u;
					ctx = (ctx >> 2) + (v << shift);
					word |= v << (2u * i);
				}
			}
			if (repair && lo == 0u && hi == 16u && __hip_atomic_load(upacked + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == word) return kMarkovSame;
			upacked[w] = word;
		}
		if ((k1 & 15u) == 0u && k1 == n) upacked[k1 >> 4] = 0u;
		return ctx;
	};
	uint32_t my_in = start << shift;
	if (have) {
		if (k0 > 1u) {
			const uint32_t kb = k0 > kMarkovWarm ? k0 - kMarkovWarm : 1u;
			uint32_t ctx = kb == 1u ? (start << shift) : 0u;
			uint32_t rw = 0;
			for (uint32_t k = kb; k < k0; k++) {
				if ((k & 15u) == 0u || k == kb) rw = ldr(k >> 4);
				const uint32_t r = (rw >> (2u * (k & 15u))) & 3u;
				const uint32_t v = (rows[ctx] >> (8u * r)) & 3u;
				ctx = (ctx >> 2) + (v << shift);
			}
			my_in = ctx;
		}
		ctx_in[tid] = my_in;
		ctx_end[tid] = decode_chunk(my_in, false);
	}
	for (uint32_t round = 0; round <= nchunks; round++) {
		__syncthreads();
		const uint32_t want = (have && tid > 0u) ? ctx_end[tid - 1] : my_in;
		const bool bad = have && tid > 0u && want != my_in;
		if (!__syncthreads_or(bad ? 1 : 0)) break;
		if (bad) {
			my_in = want;
			const uint32_t e = decode_chunk(my_in, true);
			if (e != kMarkovSame) ctx_end[tid] = e;
		}
	}
	__syncthreads();
}

// DIAG builds stamp the phase boundaries (diagnostic only): diag[zi*8 + {A, B, C, D}] cycles
template <bool DIAG>
__global__ void __launch_bounds__(kCrackBlock) k_decode_cracks(CrackArgs a, unsigned long long* __restrict__ diag) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long s_dyn[];
	__shared__ uint32_t s_scan[4 * kCrackWaves];
	__shared__ int32_t s_scanmax[kCrackWaves];
	__shared__ uint8_t s_last_move[kCrackBlock];
	__shared__ uint8_t s_last_ctrl[kCrackBlock];
	__shared__ uint32_t s_nnodes, s_ncodes, s_valid_segs, s_err, s_first_dead;
	__shared__ uint32_t s_loff[14], s_lcnt[14];
	__shared__ uint32_t s_mk_parallel, s_mk_total, s_index_end;

	unsigned long long d_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) {
		if (DIAG && threadIdx.x == 0 && diag) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			diag[static_cast<uint64_t>(blockIdx.x) * 16 + slot] = now - d_t;
			d_t = now;
		}
	};

	auto stamp_add = [&](int slot) {      // like stamp, accumulating (phases that repeat per band)
		if (DIAG && threadIdx.x == 0 && diag) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			diag[static_cast<uint64_t>(blockIdx.x) * 16 + slot] += now - d_t;
			d_t = now;
		}
	};

	const uint32_t zi = blockIdx.x + a.zbase;
	const uint32_t tid = threadIdx.x;
	const uint8_t* code = a.stream + a.code_off[zi];
	const uint32_t code_len = a.code_len[zi];
	const uint64_t cb = a.cbase[zi];
	const uint32_t cap = a.ccap[zi];
	uint32_t* nodes = a.nodes + a.nbase[zi];
	const uint32_t ncap = a.ncap[zi];
	const uint32_t sxe = a.sx + 1, sye = a.sy + 1;
	const uint32_t nverts = sxe * sye;
	const uint32_t sx = a.sx, sy = a.sy;
	uint32_t* upacked = a.upacked ? a.upacked + cb / 16u + 2ull * zi : nullptr;

	// ---- phase A: beginning-of-chain index (crackcodes.hpp:283-316), serial ----
	if (tid == 0) {
		s_mk_parallel = 0;
		uint32_t err = 0, nn = 0, ncodes = 0;
		uint32_t index_end = 0;
		if (code_len < 4u + a.yw) {
			err |= ERR_BOC;
		}
		else {
			const uint32_t index_size = rd_le_dev(code, 4);
			index_end = 4u + index_size;
			if (index_size < static_cast<uint32_t>(a.yw) || index_end > code_len || index_end < 4u) {
				err |= ERR_BOC;
				index_end = code_len;
			}
			else {
				uint32_t idx = 4;
				const uint32_t num_y = rd_le_dev(code + idx, a.yw);
				idx += a.yw;
				uint32_t y = 0;
				for (uint32_t yi = 0; yi < num_y && !err; yi++) {
					if (idx + a.yw + a.xw > index_end) { err |= ERR_BOC; break; }
					y += rd_le_dev(code + idx, a.yw); idx += a.yw;
					const uint32_t num_x = rd_le_dev(code + idx, a.xw); idx += a.xw;
					uint32_t x = 0;
					for (uint32_t xi = 0; xi < num_x; xi++) {
						if (idx + a.xw > index_end) { err |= ERR_BOC; break; }
						x += rd_le_dev(code + idx, a.xw); idx += a.xw;
						if (x >= sxe || y >= sye || nn >= ncap) { err |= ERR_BOC; break; }
						nodes[nn++] = x + sxe * y;
					}
				}
			}
			// ---- markov bitstream -> difference codes, 16 per word (markov.hpp:268-313) ----
			// all threads (markov_expand_parallel) when the slice's tables fit the LDS, else serially here
			const uint32_t nbytes = code_len - index_end;
			bool mdl_lds = false;
			if (a.markov_order == 0) {
				ncodes = nbytes * 4u;
			}
			else if (nbytes > 0 && !a.markov_serial && markov_lds_need(nbytes, cap, a.markov_order, a.lds_words * 4u, mdl_lds) <= a.lds_words * 4u) {
				s_mk_parallel = 1u + (mdl_lds ? 1u : 0u);
			}
			else if (nbytes > 0 && !a.markov_serial && a.mkscratch && a.lds_words * 4u >= 2u * kCrackBlock * 4u + 768u * 4u) {
				s_mk_parallel = 3u + (markov_model_fits_alone(a.markov_order, a.lds_words * 4u) ? 1u : 0u);      // payload + ranks in the global scratch
			}
			else if (nbytes > 0) {
				const uint8_t* s = code + index_end;
				const int shift = 2 * (a.markov_order - 1);
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
						const uint32_t v = a.model[ctx * 4u + rank];
						if (m < cap) {
							word |= v << (2u * (m & 15u));
							m++;
							if ((m & 15u) == 0) { upacked[(m >> 4) - 1u] = word; word = 0; }
						}
						else err |= ERR_CAPACITY;
						ctx = (ctx >> 2) + (v << shift);
					}
					pos -= 8;
				}
				upacked[m >> 4] = word;
				ncodes = m;
			}
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
		uint32_t nc = 0, er = 0;
		const uint32_t ie = s_index_end;
		const uint32_t mode = s_mk_parallel;
		uint32_t* gsc = a.mkscratch ? a.mkscratch + a.mkbase[zi] : nullptr;
		if (mode <= 2u) markov_expand_parallel<false>(code + ie, code_len - ie, a.markov_order, a.model, cap, upacked, reinterpret_cast<uint32_t*>(s_dyn), gsc, mode == 2u, s_scan, &s_mk_total, nc, er);
		else markov_expand_parallel<true>(code + ie, code_len - ie, a.markov_order, a.model, cap, upacked, reinterpret_cast<uint32_t*>(s_dyn), gsc, mode == 4u, s_scan, &s_mk_total, nc, er, a.lds_words);
		if (tid == 0) { s_ncodes = nc; if (er) s_err |= er; }
		__syncthreads();
	}
	if (a.markov_order) __threadfence_block();
	stamp(0);
	const uint32_t n_codes = s_ncodes;
	// positions per thread: a slice that fits one tile spreads its codes over all threads
	const uint32_t span = n_codes < kCrackTile ? max(1u, (n_codes + 16u * kCrackBlock) / (16u * kCrackBlock)) * 16u : kCrackWords * 16u;
	const uint32_t n_nodes = s_nnodes;
	const uint32_t index_end = 4u + (code_len >= 4u ? rd_le_dev(code, 4) : 0u);
	// packed difference codes as aligned words + a byte shift (only dereferenced when n_codes > 0)
	const uint8_t* packed = code + index_end;
	const uint32_t* words;
	uint32_t wshift;
	if (a.markov_order) { words = upacked; wshift = 0; }
	else {
		const uintptr_t pa = reinterpret_cast<uintptr_t>(packed);
		words = reinterpret_cast<const uint32_t*>(pa & ~static_cast<uintptr_t>(3));
		wshift = static_cast<uint32_t>(pa & 3u) * 8u;
	}

	// control tables: LDS when the slice's control symbols fit, else global
	const uint32_t lcap = a.lds_controls;
	CtlTables<uint16_t, int16_t> lt;
	{
		uint32_t* p4 = reinterpret_cast<uint32_t*>(s_dyn);
		lt.seg_x = p4; p4 += lcap + 2;
		lt.seg_y = p4; p4 += lcap + 2;
		unsigned long long* p8 = reinterpret_cast<unsigned long long*>(p4);   // (lcap + 2) * 8 bytes in: still 8-byte aligned
		lt.link = p8; p8 += lcap;
		p4 = reinterpret_cast<uint32_t*>(p8);
		lt.dx = p4; p4 += lcap;
		lt.dy = p4; p4 += lcap;
		uint16_t* p2 = reinterpret_cast<uint16_t*>(p4);
		lt.depth = reinterpret_cast<int16_t*>(p2); p2 += lcap;
		lt.lastT = p2; p2 += lcap;
		lt.gmin = reinterpret_cast<int16_t*>(p2); p2 += lcap / 7 + 48;
		lt.kind = reinterpret_cast<uint8_t*>(p2);
	}
	const uint64_t kb = cb / 2u + 4ull * zi;
	const uint32_t kcap = cap / 2u + 4u;

	uint32_t* pv = a.planeV + zi * a.plane_words;
	uint32_t* ph = a.planeH + zi * a.plane_words;
	const uint32_t row_words = a.row_words;
	uint32_t rerr = 0;
	bool segs_in_lds = true;                       // segment offsets: LDS (seg_x at word 0, seg_y at word seg_y_words) or the HBM tables
	uint32_t seg_y_words = lcap + 2u;
	uint32_t band_off_words = 2u * (lcap + 2u);   // LDS words in front of the raster band buffer
	const bool have_cracks = n_nodes > 0 && n_codes > 0;

	// The symbol stream is derived twice from the packed codes (~25 KiB per slice) instead
	// of being stored: pass 0 records only the control symbols ('b'/'t', ~3 % of the
	// stream) for the branch matcher; pass 1 recomputes every symbol and rasterises the
	// moves straight from registers once the offset of every segment is known.
	if (have_cracks) {
		TileCarry c;
		for (uint32_t tile = 0; tile <= n_codes; tile += kCrackTile) {
			WordSyms ws[kCrackWords];
			uint32_t o_a, o_dx, o_dy;
			tile_symbols<false>(words, wshift, n_codes, span, tile, c, ws, o_a, o_dx, o_dy, s_scan, s_scanmax, s_last_move, s_last_ctrl);
			if (DIAG && tid == 0 && diag) { const unsigned long long now = __builtin_amdgcn_s_memtime(); diag[static_cast<uint64_t>(blockIdx.x) * 16 + 3] += now - d_t; }      // B up to here: the symbols
			// ---- record the control symbols with the displacement before them
#pragma unroll
			for (uint32_t j = 0; j < kCrackWords; j++) {
				const WordSyms& w = ws[j];
				const uint32_t mR = w.right(), mL = w.left(), mD = w.down(), mU = w.up();
				for (uint32_t m = w.ctl; m; m &= m - 1u) {
					const uint32_t b = __ffs(m) - 1u;
					const uint32_t below = (1u << b) - 1u;
					const uint32_t kind = ((w.isT >> b) & 1u) ? SYM_T : SYM_B;
					const uint32_t cx = o_dx + __popc(mR & below) - __popc(mL & below);
					const uint32_t cy = o_dy + __popc(mD & below) - __popc(mU & below);
					if (o_a < lcap) { lt.kind[o_a] = static_cast<uint8_t>(kind); lt.dx[o_a] = cx; lt.dy[o_a] = cy; }
					else if (o_a + 2u < kcap) { a.g_kind[kb + o_a] = static_cast<uint8_t>(kind); a.g_dx[kb + o_a] = cx; a.g_dy[kb + o_a] = cy; }
					o_a++;
				}
				o_dx += __popc(mR) - __popc(mL);
				o_dy += __popc(mD) - __popc(mU);
			}
			__syncthreads();
		}
		stamp(1);

		// ---- branch matching
		const uint32_t n_ctl = c.a;
		if (n_ctl <= lcap) {
			match_controls<uint16_t, int16_t>(lt, n_ctl, nodes, n_nodes, sxe, nverts, s_scan, s_scanmax, &s_first_dead, &s_valid_segs, s_loff, s_lcnt, rerr, (DIAG && diag) ? diag + static_cast<uint64_t>(zi) * 16 : nullptr);
			// pack seg_y right behind the used part of seg_x: the rest of the LDS becomes the band buffer
			const uint32_t vs = s_valid_segs;
			constexpr uint32_t kMoves = 8;
			uint32_t tmp[kMoves];
#pragma unroll
			for (uint32_t k = 0; k < kMoves; k++) { const uint32_t i = tid + k * kCrackBlock; tmp[k] = i < vs ? lt.seg_y[i] : 0u; }
			__syncthreads();
			if (vs <= kMoves * kCrackBlock) {
#pragma unroll
				for (uint32_t k = 0; k < kMoves; k++) { const uint32_t i = tid + k * kCrackBlock; if (i < vs) lt.seg_x[vs + i] = tmp[k]; }
				seg_y_words = vs;
				band_off_words = 2u * vs;
			}
			__syncthreads();
		}
		else {
			// more control symbols than the LDS tables hold: the first lcap were recorded in LDS
			CtlTables<uint32_t, int32_t> gt;
			gt.kind = a.g_kind + kb; gt.dx = a.g_dx + kb; gt.dy = a.g_dy + kb; gt.depth = a.g_depth + kb;
			gt.lastT = a.g_lastT + kb; gt.link = a.g_link + kb; gt.seg_x = a.g_seg_x + kb; gt.seg_y = a.g_seg_y + kb;
			gt.gmin = a.g_gmin + kb;
			segs_in_lds = false;
			band_off_words = 0;
			uint32_t n = n_ctl;
			if (n + 2u >= kcap) { n = kcap - 3u; rerr |= ERR_CAPACITY; }
			for (uint32_t i = tid; i < lcap && i < n; i += kCrackBlock) { gt.kind[i] = lt.kind[i]; gt.dx[i] = lt.dx[i]; gt.dy[i] = lt.dy[i]; }
			__syncthreads();
			__threadfence_block();
			match_controls<uint32_t, int32_t>(gt, n, nodes, n_nodes, sxe, nverts, s_scan, s_scanmax, &s_first_dead, &s_valid_segs, s_loff, s_lcnt, rerr);
			__threadfence_block();
		}
		if (DIAG && tid == 0 && diag) diag[static_cast<uint64_t>(zi) * 16 + 5] = n_ctl;
		stamp(2);
	}

	// ---- pass 1: rasterise.  The segment offsets sit in LDS or, for slices with more control symbols
	// than its tables hold, in HBM: one instantiation per address space (through one pointer of
	// either kind they would be flat loads, and every wait for one also waits for all LDS traffic).
	const uint32_t valid_segs = s_valid_segs;
	auto rasterise = [&](const uint32_t* seg_x, const uint32_t* seg_y) {
		if (a.lds_raster) {
			// Both planes are built band by band in LDS (ds_or) and streamed out with plain
			// stores: no memset of the planes, no atomics on HBM.
			uint32_t* band = reinterpret_cast<uint32_t*>(s_dyn) + band_off_words;
			const uint32_t band_words = a.lds_words - band_off_words;
			uint32_t band_rows = band_words / row_words;              // one plane per pass; >= 1 (checked by the host)
			if (band_rows > sy) band_rows = sy;
			const bool single_tile = n_codes < kCrackTile;
			WordSyms ws[kCrackWords];
			uint32_t o_a = 0, o_dx = 0, o_dy = 0;
			// A slice with more codes than one tile needs several bands as well (it is big): its
			// symbols are derived once per tile and parked in HBM instead of once per tile AND band.
			uint32_t* sym = a.symbuf + a.symbase[zi];
			constexpr uint32_t kSymWords = 4u * kCrackWords + 3u;
			if (have_cracks && !single_tile) {
				TileCarry c;
				uint32_t ti = 0;
				for (uint32_t tile = 0; tile <= n_codes; tile += kCrackTile, ti++) {
					tile_symbols<true>(words, wshift, n_codes, span, tile, c, ws, o_a, o_dx, o_dy, s_scan, s_scanmax, s_last_move, s_last_ctrl);
					uint32_t* sb = sym + static_cast<uint64_t>(ti) * kSymWords * kCrackBlock + tid;
	#pragma unroll
					for (uint32_t j = 0; j < kCrackWords; j++) {
						sb[(4u * j + 0u) * kCrackBlock] = ws[j].prevs; sb[(4u * j + 1u) * kCrackBlock] = ws[j].ms;
						sb[(4u * j + 2u) * kCrackBlock] = ws[j].ctl; sb[(4u * j + 3u) * kCrackBlock] = ws[j].isT;
					}
					sb[(4u * kCrackWords + 0u) * kCrackBlock] = o_a; sb[(4u * kCrackWords + 1u) * kCrackBlock] = o_dx; sb[(4u * kCrackWords + 2u) * kCrackBlock] = o_dy;
				}
			}
			bool have_syms = false;
			for (uint32_t plane = 0; plane < 2u; plane++) {      // 0: plane V (vertical moves), 1: plane H
				for (uint32_t y0 = 0; y0 < sy; y0 += band_rows) {
					const uint32_t rows = min(band_rows, sy - y0);
					const uint32_t nw = rows * row_words;
					for (uint32_t i = tid; i < band_rows * row_words; i += kCrackBlock) band[i] = 0u;
					__syncthreads();
					stamp_add(14);
					if (have_cracks && single_tile) {
						TileCarry c;
						if (!have_syms) {
							tile_symbols<true>(words, wshift, n_codes, span, 0u, c, ws, o_a, o_dx, o_dy, s_scan, s_scanmax, s_last_move, s_last_ctrl);
							have_syms = true;
							stamp_add(6);
						}
						if (plane == 0u) raster_plane<false, false>(ws, o_a, o_dx, o_dy, valid_segs, seg_x, seg_y, sx, sy, row_words, band, y0, band_rows, rerr);
						else raster_plane<true, false>(ws, o_a, o_dx, o_dy, valid_segs, seg_x, seg_y, sx, sy, row_words, band, y0, band_rows, rerr);
					}
					else if (have_cracks) {
						uint32_t ti = 0;
						for (uint32_t tile = 0; tile <= n_codes; tile += kCrackTile, ti++) {
							const uint32_t* sb = sym + static_cast<uint64_t>(ti) * kSymWords * kCrackBlock + tid;
	#pragma unroll
							for (uint32_t j = 0; j < kCrackWords; j++) {
								ws[j].prevs = sb[(4u * j + 0u) * kCrackBlock]; ws[j].ms = sb[(4u * j + 1u) * kCrackBlock];
								ws[j].ctl = sb[(4u * j + 2u) * kCrackBlock]; ws[j].isT = sb[(4u * j + 3u) * kCrackBlock];
							}
							o_a = sb[(4u * kCrackWords + 0u) * kCrackBlock]; o_dx = sb[(4u * kCrackWords + 1u) * kCrackBlock]; o_dy = sb[(4u * kCrackWords + 2u) * kCrackBlock];
							if (plane == 0u) raster_plane<false, true>(ws, o_a, o_dx, o_dy, valid_segs, seg_x, seg_y, sx, sy, row_words, band, y0, band_rows, rerr);
							else raster_plane<true, true>(ws, o_a, o_dx, o_dy, valid_segs, seg_x, seg_y, sx, sy, row_words, band, y0, band_rows, rerr);
						}
					}
					__syncthreads();
					stamp_add(7);
					uint32_t* dst = (plane == 0u ? pv : ph) + static_cast<uint64_t>(y0) * row_words;
					for (uint32_t i = tid; i < nw; i += kCrackBlock) dst[i] = band[i];
					__syncthreads();
					stamp_add(15);
				}
			}
		}
		else if (have_cracks) {
			// planes were zeroed by the host; bits go straight to HBM
			TileCarry c;
			for (uint32_t tile = 0; tile <= n_codes; tile += kCrackTile) {
				WordSyms ws[kCrackWords];
				uint32_t o_a, o_dx, o_dy;
				tile_symbols<true>(words, wshift, n_codes, span, tile, c, ws, o_a, o_dx, o_dy, s_scan, s_scanmax, s_last_move, s_last_ctrl);
				raster_moves_hbm(ws, o_a, o_dx, o_dy, valid_segs, seg_x, seg_y, sx, sy, row_words, pv, ph, rerr);
				__syncthreads();
			}
		}

	};
	if (segs_in_lds) rasterise(reinterpret_cast<const uint32_t*>(s_dyn), reinterpret_cast<const uint32_t*>(s_dyn) + seg_y_words);
	else rasterise(a.g_seg_x + kb, a.g_seg_y + kb);

	if (rerr) atomicOr(&s_err, rerr);
	__syncthreads();
	if (tid == 0) {
		a.slice_err[zi] = s_err;      // later kernels of the decode OR their bits in
		if (a.overflow && blockIdx.x == 0 && a.zbase == 0) *a.overflow = 0u;
	}
	if (DIAG && tid == 0 && diag) diag[static_cast<uint64_t>(zi) * 16 + 4] = n_codes;
}

#include "ckl_crack_records.hpp"

// ------------------------------------------------------------------------------
// component -> label tables
// ------------------------------------------------------------------------------
// flat (labels.hpp:453-506): label_map[i] = uniq[key[i]] for the components of the
// decoded slices; i counts from the first component of slice z_start.
__global__ void __launch_bounds__(kBlock) k_label_map_flat(
	const uint8_t* __restrict__ keys, int key_width, const uint8_t* __restrict__ uniq, int stored_width,
	uint64_t num_unique, uint32_t is_signed, uint64_t n, uint64_t* __restrict__ label_map
) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (i >= n) return;
	uint64_t key = 0;
	for (int b = 0; b < key_width; b++) key |= static_cast<uint64_t>(keys[i * key_width + b]) << (8 * b);
	uint64_t v = 0;
	if (key < num_unique) {
		for (int b = 0; b < stored_width; b++) v |= static_cast<uint64_t>(uniq[key * stored_width + b]) << (8 * b);
		if (is_signed && stored_width < 8 && (v >> (8 * stored_width - 1))) v |= ~0ull << (8 * stored_width);
	}
	label_map[i] = v;
}
__global__ void __launch_bounds__(kBlock) k_fill_u64(uint64_t* p, uint64_t v, uint64_t n) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (i < n) p[i] = v;
}
// condensed pins, single-component lists (labels.hpp:578-593): ids are global component ids
__global__ void __launch_bounds__(kBlock) k_label_map_ccids(
	const uint64_t* __restrict__ ids, const uint64_t* __restrict__ labels, uint64_t n,
	uint64_t left, uint64_t right, uint64_t* __restrict__ label_map
) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (i >= n) return;
	const uint64_t id = ids[i];
	if (id >= left && id < right) label_map[id - left] = labels[i];
}
// condensed pins proper (labels.hpp:600-614): a pin at (loc, z0..z0+depth) labels the
// component of pixel loc in every slice it pierces.  One thread per (pin, slice) pair.
__global__ void __launch_bounds__(kBlock) k_label_map_pins(
	const uint64_t* __restrict__ pin_index, const uint64_t* __restrict__ pin_depth, const uint64_t* __restrict__ pin_label,
	const uint64_t* __restrict__ pin_work_off, uint64_t n_pins, uint64_t total_work,
	RunGeom g, RunArrays r, uint64_t sxy,
	int64_t z_start, int64_t z_end, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ ncomp_expect,
	uint64_t* __restrict__ label_map
) {
	const uint64_t w = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (w >= total_work) return;
	uint64_t lo = 0, hi = n_pins;
	while (lo + 1 < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (pin_work_off[mid] <= w) lo = mid; else hi = mid;
	}
	const uint64_t j = lo;
	const int64_t pin_z = static_cast<int64_t>(pin_index[j] / sxy);
	const uint64_t loc = pin_index[j] - static_cast<uint64_t>(pin_z) * sxy;
	const int64_t zs = pin_z > z_start ? pin_z : z_start;
	const int64_t z = zs + static_cast<int64_t>(w - pin_work_off[j]);
	int64_t ze = pin_z + static_cast<int64_t>(pin_depth[j]) + 1;
	if (ze > z_end) ze = z_end;
	if (z >= ze) return;
	const uint32_t zi = static_cast<uint32_t>(z - z_start);
	const uint32_t y = static_cast<uint32_t>(loc / g.sx);
	const uint32_t x = static_cast<uint32_t>(loc - static_cast<uint64_t>(y) * g.sx);
	const uint32_t wi = y * g.row_words + (x >> 5);
	const uint32_t run = r.word_base[zi * g.plane_words + wi] + __popc(g.breaks(zi, y, x >> 5) & mask_le(x & 31u)) - 1u;
	if (run >= r.nruns[zi]) return;
	const uint32_t cc = r.run_cc[r.rbase[zi] + run];
	if (cc < ncomp_expect[zi]) label_map[comp_off[zi] + cc] = pin_label[j];
}

// run -> label, typed like the output (has_label: 1 where the label matches).
// grid = (ceil(max runs / 256), nslices)
template <typename OUT>
__global__ void __launch_bounds__(kBlock) k_run_labels(
	RunArrays r, const uint64_t* __restrict__ label_map, const uint64_t* __restrict__ comp_off,
	const uint32_t* __restrict__ ncomp_expect, uint32_t has_label, uint64_t label, OUT* __restrict__ run_label
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= r.nruns[zi]) return;
	const uint32_t cc = r.run_cc[r.rbase[zi] + i];
	uint64_t v = 0;
	if (cc < ncomp_expect[zi]) v = label_map[comp_off[zi] + cc];
	else atomicOr(r.slice_err + zi, ERR_NCOMP);
	if (has_label) v = (v == label);
	run_label[r.rbase[zi] + i] = static_cast<OUT>(v);
}

// ------------------------------------------------------------------------------
// per-label statistics over the runs (operations.hpp:321-618: voxel_counts, centroids,
// bounding_boxes).  The reference walks every pixel of the slice's component image into
// per-component accumulators and merges those into the per-label maps through label_map.
// A run is a stretch of one row, so its voxel count, coordinate sums and x extent are closed
// forms of its end points: one workgroup per slice adds its runs into per-component
// accumulators in LDS, then merges the components into the label table (a binary search per
// component) with global atomics.  A slice with more components than the LDS holds merges
// run by run instead.
// ------------------------------------------------------------------------------
struct StatsArgs {
	const uint64_t* table;     // label values as label_map holds them (sign-extended), ascending as unsigned
	uint32_t n_table;
	unsigned long long* acc;   // [n_table][4]: N, sum x, sum y, sum z
	uint32_t* box;             // [n_table][6]: xmin ymin zmin xmax ymax zmax
	uint32_t sx, n_pixels;
	uint32_t z_start;
	uint32_t lds_comps;        // per-component accumulators the workgroup's LDS holds
};

constexpr int kStatsBlock = 1024;
constexpr uint32_t kStatsBytesPerComp = 8 + 8 + 4 * 5;   // sum x, sum y, N, xmin, xmax, ymin, ymax

__device__ __forceinline__ uint32_t stats_find(const StatsArgs& sa, uint64_t v) {
	uint32_t lo = 0, hi = sa.n_table;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sa.table[mid] < v) lo = mid + 1; else hi = mid;
	}
	return (lo < sa.n_table && sa.table[lo] == v) ? lo : 0xFFFFFFFFu;
}

__device__ __forceinline__ void stats_merge(
	const StatsArgs& sa, uint32_t idx, uint32_t z, unsigned long long n, unsigned long long sumx, unsigned long long sumy,
	uint32_t xmin, uint32_t xmax, uint32_t ymin, uint32_t ymax
) {
	unsigned long long* acc = sa.acc + 4ull * idx;
	atomicAdd(acc + 0, n); atomicAdd(acc + 1, sumx); atomicAdd(acc + 2, sumy); atomicAdd(acc + 3, n * z);
	uint32_t* box = sa.box + 6ull * idx;
	atomicMin(box + 0, xmin); atomicMin(box + 1, ymin); atomicMin(box + 2, z);
	atomicMax(box + 3, xmax); atomicMax(box + 4, ymax); atomicMax(box + 5, z);
}

// grid = nslices, block = kStatsBlock, dynamic LDS = lds_comps * kStatsBytesPerComp
static __global__ void __launch_bounds__(kStatsBlock) k_run_stats(
	RunArrays r, const uint64_t* __restrict__ label_map, const uint64_t* __restrict__ comp_off,
	const uint32_t* __restrict__ ncomp_expect, StatsArgs sa
) {
	extern __shared__ __attribute__((aligned(16))) unsigned char s_stats[];
	const uint32_t zi = blockIdx.x;
	const uint32_t z = sa.z_start + zi;
	const uint32_t n = r.nruns[zi];
	const uint32_t nc = ncomp_expect[zi];
	const uint64_t rb = r.rbase[zi];
	const uint64_t* lmap = label_map + comp_off[zi];
	const bool in_lds = nc <= sa.lds_comps;
	const uint32_t cap = sa.lds_comps;
	unsigned long long* s_sumx = reinterpret_cast<unsigned long long*>(s_stats);
	unsigned long long* s_sumy = s_sumx + cap;
	uint32_t* s_n = reinterpret_cast<uint32_t*>(s_sumy + cap);
	uint32_t* s_xmin = s_n + cap;
	uint32_t* s_xmax = s_xmin + cap;
	uint32_t* s_ymin = s_xmax + cap;
	uint32_t* s_ymax = s_ymin + cap;
	if (in_lds) {
		for (uint32_t c = threadIdx.x; c < nc; c += kStatsBlock) {
			s_sumx[c] = 0; s_sumy[c] = 0; s_n[c] = 0;
			s_xmin[c] = 0xFFFFFFFFu; s_xmax[c] = 0; s_ymin[c] = 0xFFFFFFFFu; s_ymax[c] = 0;
		}
		__syncthreads();
	}
	for (uint32_t i = threadIdx.x; i < n; i += kStatsBlock) {
		const uint32_t a = r.run_start[rb + i];
		const uint32_t b = (i + 1 < n) ? r.run_start[rb + i + 1] : sa.n_pixels;
		const uint32_t cc = r.run_cc[rb + i];
		if (b <= a) continue;
		if (cc >= nc) { atomicOr(r.slice_err + zi, ERR_NCOMP); continue; }
		const uint32_t y = a / sa.sx;
		const uint32_t x0 = a - y * sa.sx;
		const uint32_t len = b - a;
		const uint32_t x1 = x0 + len - 1;
		const unsigned long long sumx = (static_cast<unsigned long long>(x0) + x1) * len / 2;
		const unsigned long long sumy = static_cast<unsigned long long>(y) * len;
		if (in_lds) {
			atomicAdd(s_sumx + cc, sumx); atomicAdd(s_sumy + cc, sumy); atomicAdd(s_n + cc, len);
			atomicMin(s_xmin + cc, x0); atomicMax(s_xmax + cc, x1);
			atomicMin(s_ymin + cc, y); atomicMax(s_ymax + cc, y);
		}
		else {
			const uint32_t idx = stats_find(sa, lmap[cc]);
			if (idx == 0xFFFFFFFFu) { atomicOr(r.slice_err + zi, ERR_NCOMP); continue; }   // a label outside the table: inconsistent label section
			stats_merge(sa, idx, z, len, sumx, sumy, x0, x1, y, y);
		}
	}
	if (!in_lds) return;
	__syncthreads();
	for (uint32_t c = threadIdx.x; c < nc; c += kStatsBlock) {
		if (!s_n[c]) continue;
		const uint32_t idx = stats_find(sa, lmap[c]);
		if (idx == 0xFFFFFFFFu) { atomicOr(r.slice_err + zi, ERR_NCOMP); continue; }
		stats_merge(sa, idx, z, s_n[c], s_sumx[c], s_sumy[c], s_xmin[c], s_xmax[c], s_ymin[c], s_ymax[c]);
	}
}

static __global__ void k_stats_init(uint32_t* box, uint32_t n_table) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_table) return;
	box[6ull * i + 0] = box[6ull * i + 1] = box[6ull * i + 2] = 0xFFFFFFFFu;
	box[6ull * i + 3] = box[6ull * i + 4] = box[6ull * i + 5] = 0;
}

// ------------------------------------------------------------------------------
// paint (crackle.hpp:617-656): out[p] = label of p's run
// ------------------------------------------------------------------------------
constexpr uint32_t kPaintTile = 4096;          // pixels per workgroup
constexpr uint32_t kPaintStage = 3072;         // run labels staged in LDS per workgroup
constexpr uint32_t kPaintTable = 512;          // strip path: labels of a strip's components staged in LDS

template <typename OUT> struct Vec4;
template <> struct Vec4<uint8_t> { typedef uchar4 type; };
template <> struct Vec4<uint16_t> { typedef ushort4 type; };
template <> struct Vec4<uint32_t> { typedef uint4 type; };
template <> struct Vec4<uint64_t> { typedef ulonglong4 type; };

// the output is written once and not read again by the pipeline: non-temporal stores keep it
// from displacing the run tables in L2
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename OUT, typename V4>
__device__ __forceinline__ void store_stream(OUT* dst, const V4& v) {
	if constexpr (sizeof(V4) == 4) __builtin_nontemporal_store(*reinterpret_cast<const uint32_t*>(&v), reinterpret_cast<uint32_t*>(dst));
	else if constexpr (sizeof(V4) == 8) __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&v), reinterpret_cast<u32x2*>(dst));
	else if constexpr (sizeof(V4) == 16) __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(dst));
	else {
		__builtin_nontemporal_store(reinterpret_cast<const u32x4*>(&v)[0], reinterpret_cast<u32x4*>(dst));
		__builtin_nontemporal_store(reinterpret_cast<const u32x4*>(&v)[1], reinterpret_cast<u32x4*>(dst) + 1);
	}
}

// x-fastest output with sx % 4 == 0: a thread paints groups of 4 consecutive pixels (one plane
// word each) and stores each group as one vector.  All plane loads of the thread are issued
// first; the first and the last run of the tile fall out of them, the run labels in between are
// staged in LDS, then the look-ups and the stores follow: two dependent trips to memory per
// workgroup.  32-bit pixel arithmetic: a slice has fewer than 2^32 pixels.
template <typename OUT>
__device__ __forceinline__ void paint_fast(
	const RunGeom& g, const uint32_t* __restrict__ wb, OUT* s_lab, uint32_t* s_bounds, const OUT* __restrict__ lab,
	uint32_t nruns, uint32_t zi, uint32_t p_lo, uint32_t p_hi, uint32_t sxy, OUT* __restrict__ oz, bool no_stage
) {
	typedef typename Vec4<OUT>::type V4;
	constexpr uint32_t kIters = kPaintTile / (kBlock * 4);
	const uint32_t sx = g.sx;
	const uint32_t* pv = g.planeV + zi * g.plane_words;
	const uint32_t inv = g.flip ? 0u : 0xFFFFFFFFu;
	uint32_t p = p_lo + threadIdx.x * 4u;
	uint32_t y = p / sx;
	uint32_t x = p - y * sx;
	uint32_t pp[kIters], bws[kIters], base[kIters], shs[kIters];
#pragma unroll
	for (uint32_t it = 0; it < kIters; it++) {
		const bool ok = p < sxy;
		const uint32_t w = x >> 5;
		const uint32_t at = ok ? y * g.row_words + w : 0u;
		uint32_t bw = pv[at] ^ inv;          // bit set: a run starts at that pixel (bits past the row end are not looked at: sx % 4 == 0)
		if (w == 0) bw |= 1u;
		bws[it] = bw; base[it] = wb[at]; shs[it] = x & 31u;
		pp[it] = ok ? p : 0xFFFFFFFFu;
		p += kBlock * 4; x += kBlock * 4;
		if (x >= sx) {
			x -= sx; y++;
			if (x >= sx) { const uint32_t q = x / sx; y += q; x -= q * sx; }
		}
	}
	uint32_t run0[kIters], nibs[kIters];
#pragma unroll
	for (uint32_t it = 0; it < kIters; it++) {
		run0[it] = base[it] + __popc(bws[it] & mask_le(shs[it])) - 1u;
		nibs[it] = ((bws[it] >> shs[it]) >> 1) & 7u;      // break flags of pixels x+1 .. x+3
		if (pp[it] == p_lo) s_bounds[0] = run0[it];
		if (pp[it] + 3u == p_hi) s_bounds[1] = run0[it] + __popc(nibs[it]);
	}
	__syncthreads();
	const uint32_t lo = s_bounds[0], hi = s_bounds[1];
	const bool staged = (hi - lo) < kPaintStage && hi < nruns && !no_stage;
	if (staged) {
		for (uint32_t i = threadIdx.x; i <= hi - lo; i += kBlock) s_lab[i] = lab[lo + i];
	}
	__syncthreads();
	const uint32_t last = nruns ? nruns - 1u : 0u;
	V4 v[kIters];
	if (staged) {
#pragma unroll
		for (uint32_t it = 0; it < kIters; it++) {
			uint32_t run = pp[it] != 0xFFFFFFFFu ? run0[it] - lo : 0u;
			const uint32_t nib = pp[it] != 0xFFFFFFFFu ? nibs[it] : 0u;
			v[it].x = s_lab[run];
			run += nib & 1u;        v[it].y = s_lab[run];
			run += (nib >> 1) & 1u; v[it].z = s_lab[run];
			run += (nib >> 2) & 1u; v[it].w = s_lab[run];
		}
	}
	else {
#pragma unroll
		for (uint32_t it = 0; it < kIters; it++) {
			uint32_t run = run0[it];
			const uint32_t nib = nibs[it];
			v[it].x = lab[run < last ? run : last];
			run += nib & 1u;        v[it].y = lab[run < last ? run : last];
			run += (nib >> 1) & 1u; v[it].z = lab[run < last ? run : last];
			run += (nib >> 2) & 1u; v[it].w = lab[run < last ? run : last];
		}
	}
#pragma unroll
	for (uint32_t it = 0; it < kIters; it++) {
		if (pp[it] != 0xFFFFFFFFu) store_stream(oz + pp[it], v[it]);
	}
}

// FAST: sx % 4 == 0 and x-fastest output: a thread paints 4 consecutive pixels of one
// plane word per step and stores them as one vector.  grid = (ceil(sxy / 4096), nslices)
template <typename OUT, bool FAST>
__global__ void __launch_bounds__(kBlock) k_paint_runs(
	RunGeom g, RunArrays r, const OUT* __restrict__ run_label, OUT* __restrict__ out,
	uint64_t sxy, uint32_t nslices, uint32_t fortran_order
) {
	__shared__ OUT s_lab[kPaintStage];
	__shared__ uint32_t s_bounds[2];
	const uint32_t zi = blockIdx.y;
	const uint64_t p_lo = static_cast<uint64_t>(blockIdx.x) * kPaintTile;
	const uint64_t p_hi = (p_lo + kPaintTile < sxy ? p_lo + kPaintTile : sxy) - 1;   // last pixel of the tile
	const uint32_t* wb = r.word_base + zi * g.plane_words;
	const OUT* lab = run_label + r.rbase[zi];
	const uint32_t nruns = r.nruns[zi];
	if (FAST) {
		paint_fast<OUT>(g, wb, s_lab, s_bounds, lab, nruns, zi, static_cast<uint32_t>(p_lo), static_cast<uint32_t>(p_hi), static_cast<uint32_t>(sxy),
			out + static_cast<uint64_t>(zi) * sxy, (fortran_order & 2u) != 0);
		return;
	}
	auto run_of = [&](uint64_t p) {
		const uint32_t y = static_cast<uint32_t>(p / g.sx);
		const uint32_t x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
		return wb[y * g.row_words + (x >> 5)] + __popc(g.breaks(zi, y, x >> 5) & mask_le(x & 31u)) - 1u;
	};
	if (threadIdx.x == 0) s_bounds[0] = run_of(p_lo);
	if (threadIdx.x == 1) s_bounds[1] = run_of(p_hi);
	__syncthreads();
	const uint32_t lo = s_bounds[0], hi = s_bounds[1];
	const bool staged = (hi - lo) < kPaintStage && hi < nruns;
	if (staged) {
		for (uint32_t i = threadIdx.x; i <= hi - lo; i += kBlock) s_lab[i] = lab[lo + i];
	}
	__syncthreads();
	auto label_of = [&](uint32_t run) -> OUT {
		if (staged) return s_lab[run - lo];
		return run < nruns ? lab[run] : static_cast<OUT>(0);
	};
	{
		for (uint32_t i = threadIdx.x; i < kPaintTile; i += kBlock) {
			const uint64_t p = p_lo + i;
			if (p >= sxy) break;
			const uint32_t y = static_cast<uint32_t>(p / g.sx);
			const uint32_t x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
			const OUT v = label_of(run_of(p));
			if (fortran_order) out[static_cast<uint64_t>(zi) * sxy + p] = v;
			else out[zi + static_cast<uint64_t>(nslices) * (y + static_cast<uint64_t>(g.sy) * x)] = v;
		}
	}
}

// the groups of 4 pixels of one tile: labels from the staged table (STAGED) or straight from HBM
template <typename OUT, bool STAGED, uint32_t U = 4>
__device__ __forceinline__ void paint_tile_groups(
	const OUT* s_lab, const OUT* __restrict__ lab, const uint32_t* s_b, const uint16_t* s_wb, OUT* __restrict__ oz,
	uint32_t ngroups, uint32_t gpr, uint32_t gpr_shift, uint32_t rw, uint32_t sx, uint32_t lo, uint32_t cap
) {
	typedef typename Vec4<OUT>::type V4;
	const uint32_t t = threadIdx.x;
	const uint32_t last = cap - 1u;
	for (uint32_t g0 = 0; g0 < ngroups; g0 += kBlock * U) {
		V4 val[U];
		uint32_t at[U];
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const uint32_t gi = g0 + u * kBlock + t;
			at[u] = 0xFFFFFFFFu;
			if (gi >= ngroups) continue;
			const uint32_t row = gpr_shift != 0xFFFFFFFFu ? gi >> gpr_shift : gi / gpr;
			const uint32_t x = (gi - row * gpr) << 2;
			const uint32_t wl = row * rw + (x >> 5);
			const uint32_t bw = s_b[wl], sh = x & 31u;
			uint32_t run = s_wb[wl] + __popc(bw & mask_le(sh)) - 1u;
			const uint32_t nib = ((bw >> sh) >> 1) & 7u;
			at[u] = row * sx + x;
			if (STAGED) {
				val[u].x = s_lab[run];
				run += nib & 1u;        val[u].y = s_lab[run];
				run += (nib >> 1) & 1u; val[u].z = s_lab[run];
				run += (nib >> 2) & 1u; val[u].w = s_lab[run];
			}
			else {
				run += lo;
				val[u].x = lab[run < last ? run : last];
				run += nib & 1u;        val[u].y = lab[run < last ? run : last];
				run += (nib >> 1) & 1u; val[u].z = lab[run < last ? run : last];
				run += (nib >> 2) & 1u; val[u].w = lab[run < last ? run : last];
			}
		}
#pragma unroll
		for (uint32_t u = 0; u < U; u++) if (at[u] != 0xFFFFFFFFu) store_stream(oz + at[u], val[u]);
	}
}

// The strip path's paint (ckl_strips.hpp): one workgroup per strip.  The strip's plane words and
// the run prefix over them are built in LDS, the label of every run of the strip is staged next
// to them (run -> strip component -> label: two small dependent loads per run, issued for all
// runs at once and never inside a branch), then every thread paints pairs of adjacent 4-pixel
// groups with plain 16-byte stores.  sx % 4 == 0, x fastest.  grid = (nstrips, slices of the launch)
// What was measured at C2 (0.41 ms for the general pipeline's k_paint_runs):
//  * store pattern (tools/micro/store_bw.hip, 2 GiB): one 128 KiB chunk per workgroup 5.9 TB/s,
//    112 KiB 5.5, 16 KiB chunks 6.0 - 6.5, a grid-stride loop 4.5; non-temporal stores cost 3 - 10 %;
//  * one workgroup per 16 KiB tile of a strip (three dependent loads in front of two store sweeps):
//    0.64 ms, the prologue is not amortised;
//  * a persistent workgroup that loads two strips ahead of its stores: 0.50 ms (hipcc can only wait
//    for the prefetched words with vmcnt(0), i.e. for all of the workgroup's stores).
// The body works in `lds` (paint_strips_words<OUT>() 32-bit words, 16-byte aligned) on strip k of slice
// zi (k_paint_strips; a body so that the LDS is one block the caller owns).
// `units` groups of PX pixels of a strip from its tables in LDS; each wave-store is one contiguous stretch
// (16 bytes per lane for 4- and 8-byte labels).  SX32: rows are whole plane words (pixel p = bit p & 31 of word p >> 5).
template <typename OUT, uint32_t PX, bool SX32, typename F>
__device__ __forceinline__ void paint_units_from_lds(const uint32_t* s_b, const uint16_t* s_wb, F label_of, OUT* __restrict__ oz, uint32_t units, uint32_t sx, uint32_t rw, bool waveq = false) {
	constexpr uint32_t U = 4;
	struct alignas(PX * sizeof(OUT)) VX { OUT v[PX]; };
	const uint32_t t = threadIdx.x;
	const uint32_t inv_sx = SX32 ? 0u : 0xFFFFFFFFu / sx + 1u;      // p / sx = umulhi(p, inv) for p * sx < 2^32 (a strip has at most 2^15 pixels)
	// WAVEQ: every wavefront streams its own contiguous quarter of the strip (a multiple of 64 groups, so that
	// every wave-store stays one aligned stretch) instead of the four taking turns KiB by KiB
	const uint32_t lanes = waveq ? static_cast<uint32_t>(kWave) : static_cast<uint32_t>(kBlock);
	const uint32_t quarter = ((units + 3u) / 4u + 63u) & ~63u;
	const uint32_t first = waveq ? (t >> 6) * quarter : 0u;
	const uint32_t end = waveq ? min(units, first + quarter) : units;
	const uint32_t tl = waveq ? (t & 63u) : t;
	for (uint32_t g0 = first; g0 < end; g0 += lanes * U) {
		VX val[U];
		uint32_t at[U];
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const uint32_t gi = g0 + u * lanes + tl;
			at[u] = 0xFFFFFFFFu;
			if (gi >= end) continue;
			const uint32_t p = gi * PX;
			uint32_t wl, sh;
			if constexpr (SX32) { wl = p >> 5; sh = p & 31u; }
			else {
				const uint32_t row = __umulhi(p, inv_sx);
				const uint32_t x = p - row * sx;
				wl = row * rw + (x >> 5); sh = x & 31u;
			}
			const uint32_t bw = s_b[wl];
			uint32_t run = s_wb[wl] + __popc(bw & mask_le(sh)) - 1u;
			const uint32_t nib = (bw >> sh) >> 1;
			at[u] = p;
			val[u].v[0] = label_of(run);
#pragma unroll
			for (uint32_t q = 1; q < PX; q++) { run += (nib >> (q - 1u)) & 1u; val[u].v[q] = label_of(run); }
		}
#pragma unroll
		for (uint32_t u = 0; u < U; u++) if (at[u] != 0xFFFFFFFFu && !(exp_on(2u) && val[u].v[0] != static_cast<OUT>(0x5A))) store_stream(oz + at[u], val[u]);
	}
}

template <typename OUT>
constexpr uint32_t paint_strips_words() {
	return (((kStripCap + 8u + kPaintTable) * static_cast<uint32_t>(sizeof(OUT)) + 3u) / 4u + kStripWords + kStripWords / 2u + kWaves + 7u) & ~7u;      // (8 entries of slack behind s_lab: its last vector)
}
// VEC (rows of a multiple of four plane words: a thread's four words and every strip start are 16-byte aligned):
// the thread's plane words are ONE 16-byte load and the runs' strip components come eight (four: 16-bit) to a
// thread in 8-byte loads — 5 vector-memory loads per thread in front of its 32 stores instead of 16: a third of
// the kernel's memory instructions were loads of 1 - 4 bytes per lane.
template <typename OUT, bool DIAG, bool VEC>
__device__ __forceinline__ void paint_strips_body(
	const RunGeom& g, const StripArrays& sa, OUT* __restrict__ out, uint32_t sxy, unsigned long long* __restrict__ diag,
	uint32_t zi, uint32_t k, uint32_t* lds
) {
	OUT* s_lab = reinterpret_cast<OUT*>(lds);                // [kStripCap + 8] label of every run
	OUT* s_tab = s_lab + kStripCap + 8u;                     // [kPaintTable] labels of the strip's components (a strip with more reads them from memory)
	uint32_t* s_b = lds + ((kStripCap + 8u + kPaintTable) * static_cast<uint32_t>(sizeof(OUT)) + 3u) / 4u;
	uint16_t* s_wb = reinterpret_cast<uint16_t*>(s_b + kStripWords);
	uint32_t* s_scan = s_b + kStripWords + kStripWords / 2u;
	unsigned long long d_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) {
		if (DIAG && threadIdx.x == 0) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			atomicAdd(diag + slot, now - d_t);
			d_t = now;
		}
	};
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t rw = g.row_words, sx = g.sx;
	const uint32_t nw = (y1 - y0) * rw;
	const uint32_t t = threadIdx.x;
	if (ablated(sa, 0x10000u)) {      // tuning: the stores alone, same pattern (what the box's write path gives this grid)
		OUT* oz = out + static_cast<uint64_t>(zi) * sxy + static_cast<uint64_t>(y0) * sx;
		constexpr uint32_t PX = sizeof(OUT) == 8 ? 2u : 4u;
		auto label_of = [&](uint32_t) -> OUT { return static_cast<OUT>(k); };
		const uint32_t units = (y1 - y0) * sx / PX;
		struct alignas(PX * sizeof(OUT)) VX { OUT v[PX]; };
		if (sa.layout & 2u) {
			const uint32_t quarter = ((units + 3u) / 4u + 63u) & ~63u, first = (t >> 6) * quarter, end = min(units, first + quarter);
			for (uint32_t gi = first + (t & 63u); gi < end; gi += kWave) { VX val; for (uint32_t q = 0; q < PX; q++) val.v[q] = label_of(gi); store_stream(oz + gi * PX, val); }
		}
		else for (uint32_t gi = t; gi < units; gi += kBlock) { VX val; for (uint32_t q = 0; q < PX; q++) val.v[q] = label_of(gi); store_stream(oz + gi * PX, val); }
		return;
	}
	const uint64_t slot = ablated(sa, 0x2000u) ? 0ull : static_cast<uint64_t>(si) * sa.cap;
	// Two trips to memory in front of the stores, and as few bytes as possible: beside the stores
	// every byte read costs several bytes' worth of store time (C2: 0.41 ms with every load of the
	// strip, 0.34 ms with the loads pointed at one cached line; one trip or two made no difference).
	// First the plane words and the strip's counts, then exactly the strip components of its runs
	// (one byte each while there are at most 256) and exactly the labels of its components.  No
	// load sits in a branch of its own (it would be waited for there): lanes past the end read entry 0.
	uint32_t b[4];
	{
		const uint32_t* pv = ablated(sa, 0x6000u) ? g.planeV : g.planeV + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
		if constexpr (VEC) {
			const uint4 v4 = *reinterpret_cast<const uint4*>(pv + (t * 4u < nw ? t * 4u : 0u));
			b[0] = v4.x; b[1] = v4.y; b[2] = v4.z; b[3] = v4.w;
		}
		else {
#pragma unroll
			for (uint32_t j = 0; j < 4; j++) { const uint32_t wl = t * 4u + j; b[j] = pv[wl < nw ? wl : 0u]; }
		}
	}
	const uint32_t nr = sa.strip_nruns[si];
	const uint32_t nsc = min(sa.strip_nsc[si], sa.cap);
	if (nr > sa.cap) return;      // uniform (kStripOverflow): flagged by k_strip_ccl, the general pipeline repaints
	constexpr uint32_t kLidRounds = (kStripCap + 1023u) / 1024u;      // VEC: 8-byte loads of 8 (one byte each) or 4 strip components
	uint32_t lid[VEC ? 1 : kStripRunsPerThread];
	uint2 lid8[VEC ? kLidRounds : 1];
	const bool narrow = nsc <= 256u;
	const OUT* lab = static_cast<const OUT*>(sa.sc_label) + slot;
	{
		const uint16_t* lp = sa.run_lid + (ablated(sa, 0x8000u) ? 0ull : slot);
		if constexpr (VEC) {
			const uint32_t per = narrow ? 2048u : 1024u;      // runs of a round
#pragma unroll
			for (uint32_t r = 0; r < kLidRounds; r++) {
				const uint32_t first = r * per + t * (per / kBlock);      // my first run of the round
				if (r * per < nr) lid8[r] = reinterpret_cast<const uint2*>(lp)[first < nr ? r * kBlock + t : 0u];      // uniform condition
			}
		}
		else {
#pragma unroll
			for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
				const uint32_t j = t + i * kBlock;
				if (i * kBlock < nr) lid[i] = strip_lid(lp, nsc, j < nr ? j : 0u);      // uniform condition
			}
		}
		if (nsc <= kPaintTable) for (uint32_t j = t; j < nsc; j += kBlock) s_tab[j] = lab[j];
	}
	uint32_t cnt = 0;
	{
		const uint32_t yy = (t * 4u) / rw;
		uint32_t ww = t * 4u - yy * rw;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			b[j] = t * 4u + j < nw ? g.breaks_of(b[j], ww) : 0u;
			cnt += __popc(b[j]);
			if (++ww == rw) ww = 0;
		}
	}
	uint32_t v[1] = { cnt }, tot[1];
	block_excl_add<1>(v, tot, s_scan);      // its barriers also publish the component labels
	if (tot[0] != nr && !(kTuning && sa.ablate)) return;      // uniform; cannot happen: the same plane words gave k_strip_ccl its count
	{
		uint32_t local = v[0];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t wl = t * 4u + j;
			if (wl < nw) { s_b[wl] = b[j]; s_wb[wl] = static_cast<uint16_t>(local); }
			local += __popc(b[j]);
		}
	}
	stamp(0);
	// the labels of my runs through their strip components
	if constexpr (VEC) {
		// my runs of a round are consecutive: their labels leave in vectors
		const bool tab = nsc <= kPaintTable;
		auto label_of_lid = [&](uint32_t l) -> OUT { l = l < nsc ? l : 0u; return tab ? s_tab[l] : lab[l]; };
#pragma unroll
		for (uint32_t r = 0; r < kLidRounds; r++) {
			if (narrow) {
				const uint32_t first = r * 2048u + t * 8u;
				if (r * 2048u >= nr) break;      // uniform
				if (first >= nr) continue;
				struct alignas(8 * sizeof(OUT) > 16 ? 16 : 8 * sizeof(OUT)) V8 { OUT v[8]; } v8;
#pragma unroll
				for (uint32_t q = 0; q < 8; q++) v8.v[q] = label_of_lid(((q < 4 ? lid8[r].x : lid8[r].y) >> (8u * (q & 3u))) & 0xFFu);
				*reinterpret_cast<V8*>(s_lab + first) = v8;
			}
			else {
				const uint32_t first = r * 1024u + t * 4u;
				if (r * 1024u >= nr) break;      // uniform
				if (first >= nr) continue;
				struct alignas(4 * sizeof(OUT) > 16 ? 16 : 4 * sizeof(OUT)) V4 { OUT v[4]; } v4;
#pragma unroll
				for (uint32_t q = 0; q < 4; q++) v4.v[q] = label_of_lid(((q < 2 ? lid8[r].x : lid8[r].y) >> (16u * (q & 1u))) & 0xFFFFu);
				*reinterpret_cast<V4*>(s_lab + first) = v4;
			}
		}
	}
	else if (nsc <= kPaintTable) {      // uniform
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			if (j < nr) s_lab[j] = s_tab[lid[i] < nsc ? lid[i] : 0u];
		}
	}
	else {
		OUT lv[kStripRunsPerThread];
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) lv[i] = lab[(i * kBlock < nr && lid[i] < nsc) ? lid[i] : 0u];
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			if (j < nr) s_lab[j] = lv[i];
		}
	}
	__syncthreads();
	stamp(1);
	// groups of 4 pixels (2 for 8-byte labels: 16 bytes per lane, so that every wave-store is one contiguous KiB —
	// 32 bytes per lane left C3's paint at 2.0 TB/s); non-temporal stores: 0.39 against 0.41 ms at C2
	OUT* oz = out + static_cast<uint64_t>(zi) * sxy + static_cast<uint64_t>(y0) * sx;
	const uint32_t npix = (y1 - y0) * sx;
	auto label_of = [&](uint32_t run) -> OUT { return ablated(sa, 0x20000u) ? static_cast<OUT>(run) : s_lab[run]; };      // (tuning: without the label look-ups)
	constexpr uint32_t PX = sizeof(OUT) == 8 ? 2u : 4u;
	const bool waveq = (sa.layout & 2u) != 0u;
	if (sx == rw * 32u) paint_units_from_lds<OUT, PX, true>(s_b, s_wb, label_of, oz, npix / PX, sx, rw, waveq);
	else paint_units_from_lds<OUT, PX, false>(s_b, s_wb, label_of, oz, npix / PX, sx, rw, waveq);
	stamp(2);
}

// The same strip painted by four INDEPENDENT wavefronts, a quarter of its rows each (rows of whole plane words, a
// multiple of four of them per row, a multiple of four rows): every wavefront loads its own plane words (16 bytes
// per lane), takes the run number of its first row from row_run (k_strip_ccl / k_strip_ccl2), scans inside the
// wavefront (DPP), stages the labels of its own runs and streams its own contiguous quarter of the strip.  One
// barrier (the strip's label table) instead of five, and none between a wavefront's loads and its stores: the
// 2048 workgroups of a round no longer load, scan and store in step with each other (0.376 -> see DESIGN.md).
template <typename OUT>
__device__ __forceinline__ void paint_strips_waves_body(
	const RunGeom& g, const StripArrays& sa, OUT* __restrict__ out, uint32_t sxy, uint32_t zi, uint32_t k, uint32_t* lds
) {
	OUT* s_lab = reinterpret_cast<OUT*>(lds);                // [kStripCap + 8] label of every run
	OUT* s_tab = s_lab + kStripCap + 8u;                     // [kPaintTable] labels of the strip's components
	uint32_t* s_b = lds + ((kStripCap + 8u + kPaintTable) * static_cast<uint32_t>(sizeof(OUT)) + 3u) / 4u;
	uint16_t* s_wb = reinterpret_cast<uint16_t*>(s_b + kStripWords);
	const uint32_t t = threadIdx.x, wv = t >> 6, ln = t & 63u;
	const uint32_t si = zi * sa.nstrips + k;
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t rw = g.row_words, sx = g.sx;
	const uint32_t rq = (y1 - y0) >> 2;             // rows of a wavefront
	const uint32_t wwords = rq * rw;                // its plane words (<= 256: a strip has <= 1024)
	const uint32_t wbase = wv * wwords;             // its first word, relative to the strip
	const uint64_t slot = static_cast<uint64_t>(si) * sa.cap;
	// ---- first trip: my plane words, the strip's counts, the run numbers of my first row and of the next wavefront's
	const uint32_t* pv = g.planeV + zi * g.plane_words + static_cast<uint64_t>(y0) * rw;
	const bool have = ln * 4u < wwords;
	const uint4 v4 = *reinterpret_cast<const uint4*>(pv + wbase + (have ? ln * 4u : 0u));
	const uint32_t nr = sa.strip_nruns[si];
	const uint32_t nsc = min(sa.strip_nsc[si], sa.cap);
	const uint16_t* rr = sa.row_run + static_cast<uint64_t>(zi) * g.sy + y0;
	const uint32_t r0 = rr[wv * rq];
	const uint32_t r1_raw = rr[wv < 3u ? (wv + 1u) * rq : 0u];
	if (nr > sa.cap) return;      // uniform (kStripOverflow): flagged by the strip kernel, the general pipeline repaints
	const uint32_t r1 = wv < 3u ? r1_raw : nr;
	const OUT* lab = static_cast<const OUT*>(sa.sc_label) + slot;
	const bool tab = nsc <= kPaintTable;
	if (tab) for (uint32_t j = t; j < nsc; j += kBlock) s_tab[j] = lab[j];
	// ---- second trip: the strip components of my runs [r0, r1), 8 bytes per lane from an aligned start
	const bool narrow = nsc <= 256u;
	const uint32_t per8 = narrow ? 8u : 4u;              // runs per 8-byte piece
	const uint32_t a0 = r0 & ~(per8 - 1u);
	const uint2* lp8 = reinterpret_cast<const uint2*>(sa.run_lid + slot);
	uint32_t first = a0 + ln * per8;
	uint2 piece = lp8[first < r1 ? first / per8 : 0u];
	// ---- breaks of my words, run numbers inside the wavefront
	{
		const uint32_t w0 = (wbase + ln * 4u) & (rw - 1u);      // (rw is a multiple of 4, not necessarily a power of two)
		const uint32_t wrow = (wbase + ln * 4u) % rw;
		(void)w0;
		const uint32_t fm = g.flip ? 0u : 0xFFFFFFFFu;
		const uint32_t on = have ? 0xFFFFFFFFu : 0u;
		uint32_t b[4];
		b[0] = ((v4.x ^ fm) | (wrow == 0u ? 1u : 0u)) & on;
		b[1] = (v4.y ^ fm) & on;
		b[2] = (v4.z ^ fm) & on;
		b[3] = (v4.w ^ fm) & on;
		const uint32_t c0 = __popc(b[0]), c1 = __popc(b[1]), c2 = __popc(b[2]), c3 = __popc(b[3]);
		const uint32_t tot = c0 + c1 + c2 + c3;
		const uint32_t l0 = r0 + wave_incl_add(tot) - tot, l1 = l0 + c0, l2 = l1 + c1, l3 = l2 + c2;
		if (have) {
			*reinterpret_cast<uint4*>(s_b + wbase + ln * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
			*reinterpret_cast<uint2*>(s_wb + wbase + ln * 4u) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
		}
	}
	__syncthreads();      // the label table (the only thing the wavefronts share)
	// ---- labels of my runs, in vectors (a piece that starts below r0 repeats what the wavefront before me writes)
	auto label_of_lid = [&](uint32_t l) -> OUT { l = l < nsc ? l : 0u; return tab ? s_tab[l] : lab[l]; };
	for (;;) {
		if (first < r1) {
			if (narrow) {
				struct alignas(8 * sizeof(OUT) > 16 ? 16 : 8 * sizeof(OUT)) V8 { OUT v[8]; } v8;
#pragma unroll
				for (uint32_t q = 0; q < 8; q++) v8.v[q] = label_of_lid(((q < 4 ? piece.x : piece.y) >> (8u * (q & 3u))) & 0xFFu);
				*reinterpret_cast<V8*>(s_lab + first) = v8;
			}
			else {
				struct alignas(4 * sizeof(OUT) > 16 ? 16 : 4 * sizeof(OUT)) V4 { OUT v[4]; } v4l;
#pragma unroll
				for (uint32_t q = 0; q < 4; q++) v4l.v[q] = label_of_lid(((q < 2 ? piece.x : piece.y) >> (16u * (q & 1u))) & 0xFFFFu);
				*reinterpret_cast<V4*>(s_lab + first) = v4l;
			}
		}
		first += kWave * per8;
		if (__ballot(first < r1) == 0ull) break;      // wave-uniform (a0 + 64 * per8 covers the runs of 8 rows of every BASELINE shape: one round)
		piece = lp8[first < r1 ? first / per8 : 0u];
	}
	// ---- my quarter of the strip: groups of 4 pixels (2 for 8-byte labels), 16 bytes per lane and store
	constexpr uint32_t PX = sizeof(OUT) == 8 ? 2u : 4u;
	constexpr uint32_t U = 4;
	struct alignas(PX * sizeof(OUT)) VX { OUT v[PX]; };
	OUT* oz = out + static_cast<uint64_t>(zi) * sxy + static_cast<uint64_t>(y0) * sx;
	const uint32_t q_units = rq * sx / PX;
	const uint32_t u_first = wv * q_units, u_end = u_first + q_units;
	for (uint32_t g0 = u_first; g0 < u_end; g0 += kWave * U) {
		VX val[U];
		uint32_t at[U];
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const uint32_t gi = g0 + u * kWave + ln;
			at[u] = 0xFFFFFFFFu;
			if (gi >= u_end) continue;
			const uint32_t p = gi * PX;
			const uint32_t wl = p >> 5, sh = p & 31u;
			const uint32_t bw = s_b[wl];
			uint32_t run = s_wb[wl] + __popc(bw & mask_le(sh)) - 1u;
			const uint32_t nib = (bw >> sh) >> 1;
			at[u] = p;
			val[u].v[0] = s_lab[run];
#pragma unroll
			for (uint32_t q = 1; q < PX; q++) { run += (nib >> (q - 1u)) & 1u; val[u].v[q] = s_lab[run]; }
		}
#pragma unroll
		for (uint32_t u = 0; u < U; u++) if (at[u] != 0xFFFFFFFFu) store_stream(oz + at[u], val[u]);
	}
}

template <typename OUT, bool DIAG>
__global__ void __launch_bounds__(kBlock) k_paint_strips(
	RunGeom g, StripArrays sa, OUT* __restrict__ out, uint32_t sxy, unsigned long long* __restrict__ diag
) {
	__shared__ __attribute__((aligned(16))) uint32_t s_lds[paint_strips_words<OUT>()];
	uint32_t zl, k;
	strip_of_block(sa, zl, k);
	if (!DIAG && (sa.layout & 2u) && (g.row_words & 3u) == 0u && g.sx == g.row_words * 32u) {
		const uint32_t y0 = k * sa.strip_rows, rows = min(y0 + sa.strip_rows, g.sy) - y0;
		if ((rows & 3u) == 0u && !(kTuning && sa.ablate)) { paint_strips_waves_body<OUT>(g, sa, out, sxy, zl + sa.zbase, k, s_lds); return; }
	}
	if ((g.row_words & 3u) == 0u) paint_strips_body<OUT, DIAG, true>(g, sa, out, sxy, diag, zl + sa.zbase, k, s_lds);
	else paint_strips_body<OUT, DIAG, false>(g, sa, out, sxy, diag, zl + sa.zbase, k, s_lds);
}

// ---- k_strip_fused: strips -> labels in ONE launch ------------------------------------------------------
// Replaces, for flat labels on the record path, the three launches k_strip_ccl / k_slice_resolve /
// k_paint_strips (color_connectivity_graph + relabel, src/cc3d.hpp:114-254; crc32c of the component image,
// src/crackle.hpp:599-611; decode_flat, src/labels.hpp:453-506; the paint loop, src/crackle.hpp:617-656) with
// one launch in which the VALU-bound labelling of later slices runs beside the store-bound paint of earlier ones.
// Every workgroup takes ONE work item from a ticket counter; the items of a counter are, in order: the strips of
// the first `lag` slices to LABEL, then alternately one strip of slice i + lag to label and one strip of slice i
// to PAINT, then the strips of the last `lag` slices to paint.
//   label  strip_ccl_body<FUSED>: records -> plane pieces -> runs -> strip components, all in LDS; the strip's
//          paint tables (break words, run prefixes, strip component of every run: one block of 11 KiB) and what
//          the resolver reads leave with write-through stores; the workgroup counts itself in on its slice (one
//          agent-scope atomic add), and the one whose add came last resolves the slice (slice_resolve_body: seams,
//          ranks = the reference's ids, crc32c, the label of every strip component) and raises the slice's flag;
//   paint  polls that flag (one lane, relaxed, s_sleep; raised long before when `lag` covers label + resolve),
//          fetches the strip's tables and labels in ONE trip to memory (no scan, nothing derived again) and paints.
// Forward progress: a paint item only ever waits for label items with SMALLER tickets of the same counter, which
// running workgroups hold (tickets are taken by workgroups that already run) and which wait for nothing.  Waits are
// bounded all the same (`timeout` -> the host falls back to the three launches).
// Tickets come from `nheads` counters (one word serves ~90 tickets per microsecond, the launch takes 70): a
// workgroup starts at counter blockIdx % nheads — workgroups b and b + 8 share an XCD, so a slice's strips and
// their tables mostly stay in one L2 — and moves on to the next counter when its own is used up.
// Hand-offs follow the write-through form (every handed-off byte stored sc1 in 4- or 16-byte pieces and drained
// by its wave, a barrier, then one lane's agent-scope atomic; every load of those bytes sc1): no dependence on
// dispatch order or XCD placement, no L2 write-back / invalidate while the paint's stores are in flight.
constexpr uint32_t kFusedStateWords = kStripEdgeCap + kStripWords / 2u;      // s_mem (s_b | s_pool) and s_wb of strip_ccl_body, as they lie in LDS
constexpr uint32_t kFusedScbaseWords = 264;                                      // nstrips + 1 <= 257
constexpr uint32_t kFusedResolveCap = kStripCap + kFusedStateWords - kFusedScbaseWords;      // strip components of a slice the resolver's table holds
constexpr uint32_t kFusedMaxStrips = 256;
constexpr uint32_t kFusedMaxHeads = 8;
constexpr uint32_t kFusedHeadStride = 64;          // words between two ticket counters: a 256-byte stretch each (one line serves ~90 atomics per microsecond whatever the word)
constexpr uint32_t kFusedCtlWords = (kFusedMaxHeads + 1u) * kFusedHeadStride;      // heads, timeout; then arrive[nslices], ready[nslices]
constexpr uint32_t kFusedSpinLimit = 1u << 21;
static_assert(kFusedStateWords % 4 == 0 && kStripCap % 4 == 0 && kStripWords % 8 == 0, "16-byte pieces");
static_assert(kFusedResolveCap <= 0xFFFFu, "index and rank share a table entry");

struct FusedCtl {
	uint32_t* heads;       // [nheads] items handed out so far (zeroed by k_crack_match, like arrive / ready / timeout)
	uint32_t* arrive;      // [nslices] strips of the slice that are labelled
	uint32_t* ready;       // [nslices] 0: not yet, 1: sc_label holds the labels of the slice's strip components, 2: the slice does not fit the strip tables
	uint32_t* timeout;     // a wait gave up
	uint32_t* state;       // [strips][kFusedStateWords] paint tables of every strip
	uint32_t nslices, nheads, lag;      // lag: slices (of one counter) between label and paint
};

template <typename OUT>
__global__ void __launch_bounds__(kBlock, 7) k_strip_fused(
	RunGeom g, StripArrays sa, RecordLists rl, ResolveArgs ra, FusedCtl fc, const uint32_t* __restrict__ G, uint32_t n_pixels,
	uint32_t* __restrict__ ncomp_out, OUT* __restrict__ out, uint32_t sxy
) {
	typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) uint32_t s_lds[kStripCclWords];
	__shared__ uint32_t s_ctl[6];
	const uint32_t t = threadIdx.x;
	const uint32_t N = sa.nstrips;
	// ---- one work item
	if (t == 0) {
		uint32_t found = 0xFFFFFFFFu, head = 0;
		for (uint32_t a = 0; a < fc.nheads; a++) {
			const uint32_t h = (blockIdx.x + a) % fc.nheads;
			const uint32_t n_h = h < fc.nslices ? (fc.nslices - h + fc.nheads - 1u) / fc.nheads : 0u;
			const uint32_t items = 2u * n_h * N;
			if (items == 0u) continue;
			const uint32_t tk = __hip_atomic_fetch_add(fc.heads + h * kFusedHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (tk < items) { found = tk; head = h; break; }
		}
		s_ctl[0] = found; s_ctl[1] = head;
	}
	__syncthreads();
	if (s_ctl[0] == 0xFFFFFFFFu) return;      // (the grid is exactly the items: cannot happen)
	uint32_t zi, k;
	bool paint;
	{
		const uint32_t tk = s_ctl[0], h = s_ctl[1];
		const uint32_t n_h = (fc.nslices - h + fc.nheads - 1u) / fc.nheads;
		const uint32_t lag = min(fc.lag, n_h);
		const uint32_t head_items = lag * N, mixed = 2u * (n_h - lag) * N;
		uint32_t strip;      // strip index within the counter's slices
		if (tk < head_items) { paint = false; strip = tk; }
		else if (tk < head_items + mixed) { const uint32_t m = tk - head_items; paint = (m & 1u) != 0u; strip = paint ? (m >> 1) : head_items + (m >> 1); }
		else { paint = true; strip = (n_h - lag) * N + (tk - head_items - mixed); }
		const uint32_t zl = strip / N;
		k = strip - zl * N;
		zi = h + zl * fc.nheads;
	}
	const uint32_t si = zi * N + k;
	uint32_t* state = fc.state + static_cast<uint64_t>(si) * kFusedStateWords;
	const __amdgpu_buffer_rsrc_t state_rsrc = __builtin_amdgcn_make_buffer_rsrc(state, 0, kFusedStateWords * 4u, 0x00020000);
	const uint32_t y0 = k * sa.strip_rows;
	const uint32_t y1 = min(y0 + sa.strip_rows, g.sy);
	const uint32_t nw = (y1 - y0) * g.row_words;
	u32x4_t* keep = reinterpret_cast<u32x4_t*>(s_lds + kStripCap);      // s_b [kStripWords] | s_pool [kStripCap / 2] | s_wb [kStripWords / 2]
	constexpr uint32_t kPoolAt = kStripWords / 4u, kWbAt = kStripEdgeCap / 4u;      // in 16-byte pieces

	if (!paint) {
		// ================= label =================
		uint32_t nsc = 0;
		const uint32_t nloc = exp_on(32u) ? 1u : strip_ccl_body<false, true, true>(g, sa, rl, G, n_pixels, nullptr, zi, k, s_lds, &nsc);
		if (nloc != kStripOverflow) {      // uniform: the strip's paint tables (write-through, 16 bytes per lane)
			const uint32_t nb = (nw + 3u) / 4u, np = (nloc + 7u) / 8u, nwb = (nw + 7u) / 8u;
			for (uint32_t i = t; i < kFusedStateWords / 4u; i += kBlock) {
				const bool on = i < kPoolAt ? i < nb : (i < kWbAt ? i - kPoolAt < np : i - kWbAt < nwb);
				if (on && !exp_on(4u)) {
					if (exp_on(1u)) __builtin_amdgcn_raw_buffer_store_b128(keep[i], state_rsrc, i * 16u, 0, 0);
					else __builtin_amdgcn_raw_buffer_store_b128(keep[i], state_rsrc, i * 16u, 0, 16);
				}
			}
		}
		// count in: every wave has drained its write-through stores, then one lane adds
		if (!exp_on(8u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (t == 0) s_ctl[2] = __hip_atomic_fetch_add(fc.arrive + zi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		if (s_ctl[2] + 1u != N) return;      // uniform
		// the slice's last strip resolves it, its LDS is free
		ResolveArgs rb = ra;
		rb.cap = min(ra.cap, kFusedResolveCap);
		uint32_t* s_scan = s_lds + kStripCap + kFusedStateWords;
		const bool ok = slice_resolve_body<OUT, true, false, kBlock, true>(g, sa, rb, ncomp_out, nullptr, zi, s_lds + kFusedScbaseWords, s_lds, s_scan, &s_ctl[3]);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its labels are out
		__syncthreads();
		if (t == 0) __hip_atomic_store(fc.ready + zi, ok ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	// ================= paint =================
	if (t == 0) {
		uint32_t v, spins = 0;
		while ((v = __hip_atomic_load(fc.ready + zi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
			__builtin_amdgcn_s_sleep(16);
			if (++spins > kFusedSpinLimit) { atomicOr(fc.timeout, 1u); v = 3u; break; }
		}
		s_ctl[4] = v;
	}
	__syncthreads();
	if (s_ctl[4] != 1u || exp_on(16u)) return;      // uniform: the host repaints through the general pipeline
	// one trip: counts, the strip's tables, the first labels
	typedef typename std::conditional<sizeof(OUT) < 4, uint32_t, OUT>::type LAB;      // narrow labels travel as 32-bit words
	const LAB* lab = static_cast<const LAB*>(sa.sc_label) + static_cast<uint64_t>(si) * sa.cap;
	const uint32_t nloc = hand_ld<true>(sa.strip_nruns + si);
	const uint32_t nsc = min(hand_ld<true>(sa.strip_nsc + si), sa.cap);
	LAB first_lab = hand_ld<true>(lab + (t < sa.cap ? t : 0u));
	{
		const uint32_t nb = (nw + 3u) / 4u, np = (sa.cap + 7u) / 8u, nwb = (nw + 7u) / 8u;      // (the runs are not known yet: the whole pool)
		constexpr uint32_t kPer = (kFusedStateWords / 4u + kBlock - 1u) / kBlock;
		u32x4_t piece[kPer];
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			const uint32_t i = t + q * kBlock;
			const bool on = i < kPoolAt ? i < nb : (i < kWbAt ? i - kPoolAt < np : (i < kFusedStateWords / 4u && i - kWbAt < nwb));
			piece[q] = __builtin_amdgcn_raw_buffer_load_b128(state_rsrc, ((on && !exp_on(4u)) ? i : 0u) * 16u, 0, 16);
		}
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) { const uint32_t i = t + q * kBlock; if (i < kFusedStateWords / 4u) keep[i] = piece[q]; }
	}
	if (nloc > sa.cap) return;      // uniform (kStripOverflow; cannot be: the slice would not be ready)
	const uint32_t* s_b = s_lds + kStripCap;
	const uint16_t* s_pool = reinterpret_cast<const uint16_t*>(s_lds + kStripCap + kStripWords);
	const uint16_t* s_wb = reinterpret_cast<const uint16_t*>(s_lds + kStripCap + kStripEdgeCap);
	OUT* oz = out + static_cast<uint64_t>(zi) * sxy + static_cast<uint64_t>(y0) * g.sx;
	const uint32_t npix = (y1 - y0) * g.sx;
	const bool sx32 = g.sx == g.row_words * 32u;
	// the strip components' labels staged over the union-find table's place: kStripCap words
	OUT* s_tab = reinterpret_cast<OUT*>(s_lds);
	constexpr uint32_t kTab = kStripCap * 4u / static_cast<uint32_t>(sizeof(OUT)) >= kStripCap ? kStripCap : kStripCap * 4u / static_cast<uint32_t>(sizeof(OUT));
	const bool staged = nsc <= kTab;      // uniform
	if (staged) {
		if (t < nsc) s_tab[t] = static_cast<OUT>(first_lab);
		for (uint32_t j = t + kBlock; j < nsc; j += kBlock) s_tab[j] = static_cast<OUT>(hand_ld<true>(lab + j));      // a second trip: strips with more than 256 components
	}
	__syncthreads();
	if (!staged) {      // more strip components than the table holds: every pixel looks its label up in memory
		auto label_of = [&](uint32_t run) -> OUT { return static_cast<OUT>(hand_ld<true>(lab + s_pool[run])); };
		if constexpr (sizeof(OUT) == 8) paint_units_from_lds<OUT, 2, false>(s_b, s_wb, label_of, oz, npix >> 1, g.sx, g.row_words);
		else paint_units_from_lds<OUT, 4, false>(s_b, s_wb, label_of, oz, npix >> 2, g.sx, g.row_words);
		return;
	}
	if constexpr (sizeof(OUT) == 8) {
		// 8-byte labels: a pixel looks up run -> strip component -> label; a lane paints 2 pixels = 16 bytes per
		// store, so that a wave-store is one contiguous KiB
		auto label_of = [&](uint32_t run) -> OUT { return s_tab[s_pool[run]]; };
		if (sx32) paint_units_from_lds<OUT, 2, true>(s_b, s_wb, label_of, oz, npix >> 1, g.sx, g.row_words);
		else paint_units_from_lds<OUT, 2, false>(s_b, s_wb, label_of, oz, npix >> 1, g.sx, g.row_words);
	}
	else {
		// the label of every run, once, in the table's place: a pixel then costs one look-up
		OUT lv[kStripRunsPerThread];
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) {
			const uint32_t j = t + i * kBlock;
			const uint32_t lid = s_pool[j < nloc ? j : 0u];
			lv[i] = s_tab[lid < nsc ? lid : 0u];
		}
		__syncthreads();
		OUT* s_lab = reinterpret_cast<OUT*>(s_lds);
#pragma unroll
		for (uint32_t i = 0; i < kStripRunsPerThread; i++) { const uint32_t j = t + i * kBlock; if (j < nloc) s_lab[j] = lv[i]; }
		__syncthreads();
		auto label_of = [&](uint32_t run) -> OUT { return s_lab[run]; };
		if (sx32) paint_units_from_lds<OUT, 4, true>(s_b, s_wb, label_of, oz, npix >> 2, g.sx, g.row_words);
		else paint_units_from_lds<OUT, 4, false>(s_b, s_wb, label_of, oz, npix >> 2, g.sx, g.row_words);
	}
}

// condensed pins on the strip path (labels.hpp:600-614): one thread per (pin, slice) pair
__global__ void __launch_bounds__(kBlock) k_label_map_pins_strips(
	const uint64_t* __restrict__ pin_index, const uint64_t* __restrict__ pin_depth, const uint64_t* __restrict__ pin_label,
	const uint64_t* __restrict__ pin_work_off, uint64_t n_pins, uint64_t total_work,
	RunGeom g, StripArrays sa, uint64_t sxy,
	int64_t z_start, int64_t z_end, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ ncomp_expect,
	uint64_t* __restrict__ label_map
) {
	const uint64_t w = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (w >= total_work) return;
	uint64_t lo = 0, hi = n_pins;
	while (lo + 1 < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (pin_work_off[mid] <= w) lo = mid; else hi = mid;
	}
	const uint64_t j = lo;
	const int64_t pin_z = static_cast<int64_t>(pin_index[j] / sxy);
	const uint64_t loc = pin_index[j] - static_cast<uint64_t>(pin_z) * sxy;
	const int64_t zs = pin_z > z_start ? pin_z : z_start;
	const int64_t z = zs + static_cast<int64_t>(w - pin_work_off[j]);
	int64_t ze = pin_z + static_cast<int64_t>(pin_depth[j]) + 1;
	if (ze > z_end) ze = z_end;
	if (z >= ze) return;
	const uint32_t zi = static_cast<uint32_t>(z - z_start);
	const uint32_t y = static_cast<uint32_t>(loc / g.sx);
	const uint32_t x = static_cast<uint32_t>(loc - static_cast<uint64_t>(y) * g.sx);
	const uint32_t cc = strip_component_of_pixel(g, sa, zi, x, y);
	if (cc < ncomp_expect[zi]) label_map[comp_off[zi] + cc] = pin_label[j];
}

// ------------------------------------------------------------------------------
// voxel connectivity graph (operations.hpp:667-826): bit0 +x, bit1 -x, bit2 +y, bit3 -y from the
// crack planes (a pair across the image border is passable for IMPERMISSIBLE streams and not for
// PERMISSIBLE ones: the reference starts from all-ones / all-zeros and only touches interior
// pairs); connectivity 6 adds bit4 +z / bit5 -z where the decoded labels of neighbouring slices
// agree, and marks the first slice's -z and the last slice's +z.
// grid = (ceil(sxy / 256), nslices); out: x fastest
// ------------------------------------------------------------------------------
template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_vcg(
	RunGeom g, const LABEL* __restrict__ labels, uint32_t six, uint32_t fortran_order, uint32_t nslices, uint64_t sxy, uint8_t* __restrict__ out
) {
	const uint32_t zi = blockIdx.y;
	const uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (p >= sxy) return;
	const uint32_t y = static_cast<uint32_t>(p / g.sx);
	const uint32_t x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
	const uint32_t* pv = g.planeV + zi * g.plane_words;
	const uint32_t* ph = g.planeH + zi * g.plane_words;
	// plane bit: crack (IMPERMISSIBLE, flip) or connection (PERMISSIBLE)
	auto joined = [&](const uint32_t* plane, uint32_t px, uint32_t py) -> uint32_t {
		const uint32_t bit = (plane[static_cast<uint64_t>(py) * g.row_words + (px >> 5)] >> (px & 31u)) & 1u;
		return g.flip ? (bit ^ 1u) : bit;
	};
	const uint32_t border = g.flip ? 1u : 0u;
	uint32_t v = 0;
	v |= (x + 1 < g.sx ? joined(pv, x + 1, y) : border) << 0;
	v |= (x >= 1 ? joined(pv, x, y) : border) << 1;
	v |= (y + 1 < g.sy ? joined(ph, x, y + 1) : border) << 2;
	v |= (y >= 1 ? joined(ph, x, y) : border) << 3;
	if (six && nslices > 1) {
		auto at = [&](uint32_t z) -> LABEL {
			return fortran_order ? labels[static_cast<uint64_t>(z) * sxy + p] : labels[z + static_cast<uint64_t>(nslices) * (y + static_cast<uint64_t>(g.sy) * x)];
		};
		const LABEL me = at(zi);
		if (zi + 1 < nslices ? at(zi + 1) == me : true) v |= 0x10u;
		if (zi >= 1 ? at(zi - 1) == me : true) v |= 0x20u;
	}
	out[static_cast<uint64_t>(zi) * sxy + p] = static_cast<uint8_t>(v);
}

// compares the accumulated raw crc with the stored per-slice crc32c and the computed
// component counts with the label section; grid = ceil(nslices / 256)
__global__ void __launch_bounds__(kBlock) k_check(
	const uint32_t* __restrict__ crc_acc, const uint32_t* __restrict__ crc_expect_raw,
	const uint32_t* __restrict__ ncomp, const uint32_t* __restrict__ ncomp_expect,
	uint32_t check_crc, uint32_t crc_fix, uint32_t nslices, uint32_t* __restrict__ slice_err
) {
	const uint32_t zi = blockIdx.x * kBlock + threadIdx.x;
	if (zi >= nslices) return;
	uint32_t e = 0;
	if (ncomp[zi] != ncomp_expect[zi]) e |= ERR_NCOMP;
	// crc_acc holds the raw crc divided by x^(32 - idbits): multiply it back
	else if (check_crc && gf_mul(crc_acc[zi], crc_fix) != crc_expect_raw[zi]) e |= ERR_CRC;
	if (e) atomicOr(slice_err + zi, e);
}

// ------------------------------------------------------------------------------
// array_equal (operations.hpp:1039-1184) and mode_pooling_2x2x1 (operations.hpp:1201-1304)
// ------------------------------------------------------------------------------
// one flag: do two device buffers differ anywhere (16 bytes per thread and step, tail by bytes)
__global__ void __launch_bounds__(kBlock) k_buffers_differ(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n, uint32_t* __restrict__ differ) {
	const uint64_t nv = n / 16;
	uint32_t bad = 0;
	for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < nv; i += static_cast<uint64_t>(gridDim.x) * kBlock) {
		const uint4 x = reinterpret_cast<const uint4*>(a)[i], y = reinterpret_cast<const uint4*>(b)[i];
		bad |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) bad |= a[nv * 16 + threadIdx.x] ^ b[nv * 16 + threadIdx.x];
	if (bad) atomicOr(differ, 1u);
}

// the reference's 2 x 2 pooling rule (operations.hpp:1254-1290): a == b -> a, a == c -> a, b == c -> b,
// else d; the last column / row of an odd-sized slice is copied.  in: x fastest, one slice after the other
template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_mode_pool_2x2(const LABEL* __restrict__ in, LABEL* __restrict__ out, uint32_t sx, uint32_t sy, uint32_t nslices) {
	const uint32_t osx = (sx + 1u) >> 1, osy = (sy + 1u) >> 1;
	const uint64_t osxy = static_cast<uint64_t>(osx) * osy, sxy = static_cast<uint64_t>(sx) * sy;
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (i >= osxy * nslices) return;
	const uint32_t z = static_cast<uint32_t>(i / osxy);
	const uint32_t r = static_cast<uint32_t>(i - z * osxy);
	const uint32_t oy = r / osx, ox = r - oy * osx;
	const LABEL* src = in + z * sxy;
	const uint32_t x = 2u * ox, y = 2u * oy;
	const bool has_x = x + 1u < sx, has_y = y + 1u < sy;
	const LABEL a = src[x + static_cast<uint64_t>(sx) * y];
	LABEL v = a;
	if (has_x && has_y) {
		const LABEL b = src[x + 1u + static_cast<uint64_t>(sx) * y];
		const LABEL c = src[x + static_cast<uint64_t>(sx) * (y + 1u)];
		const LABEL dd = src[x + 1u + static_cast<uint64_t>(sx) * (y + 1u)];
		v = (a == b) ? a : (a == c) ? a : (b == c) ? b : dd;
	}
	out[i] = v;
}

// the per-slice error words and the strip path's overflow word, stored straight into the host's pinned memory
__global__ void __launch_bounds__(kBlock) k_flags_to_host(const uint32_t* __restrict__ slice_err, uint32_t n, const uint32_t* __restrict__ overflow, const uint32_t* __restrict__ timeout, uint32_t* __restrict__ dst_host) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i < n) dst_host[i] = slice_err[i];
	if (i == 0) { dst_host[n] = overflow ? *overflow : 0u; dst_host[n + 1] = timeout ? *timeout : 0u; }
}

// a few KiB from HBM into the host's pinned (device-mapped) memory: what ckl_decoder_create_device reads
// back.  A kernel's stores arrive sooner than a copy-engine transfer is even scheduled.
__global__ void __launch_bounds__(kBlock) k_fetch_to_host(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst_host, uint64_t off0, uint64_t len0, uint64_t off1, uint64_t len1) {
	const uint64_t i = (static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x) * 16u;
	const uint64_t off = blockIdx.y ? off1 : off0, len = blockIdx.y ? len1 : len0;
	if (i >= len) return;
	if (i + 16u <= len && ((reinterpret_cast<uintptr_t>(src + off + i) | reinterpret_cast<uintptr_t>(dst_host + off + i)) & 15u) == 0) {
		*reinterpret_cast<uint4*>(dst_host + off + i) = *reinterpret_cast<const uint4*>(src + off + i);
	}
	else for (uint64_t b = i; b < len && b < i + 16u; b++) dst_host[off + b] = src[off + b];
}

}  // namespace ckl

// ------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------
using namespace ckl;

namespace {
constexpr int kMaxStages = 20;
}

struct ckl_decoder {
	int device = 0;
	int n_cus = 256;
	int max_lds = 0;
	hipStream_t stream = nullptr;
	// stage boundaries: ev[i] .. ev[i+1] brackets stage i of the last run
	hipEvent_t ev[kMaxStages + 2] = {};      // [kMaxStages + 1]: end of the pipeline
	hipEvent_t ev_in = nullptr;
	const char* stage_name[kMaxStages] = {};
	float stage_ms[kMaxStages] = {};
	int n_stages = 0;
	float pipeline_ms = 0.f;

	Header head;
	uint64_t n_bytes = 0;
	int64_t z_start = 0, z_end = 0;
	uint32_t nslices = 0;
	uint64_t sxy = 0;

	// device residents
	DevBuf<uint8_t> d_stream;            // the whole stream (a view of the caller's buffer for ckl_decoder_create_device)
	DevBuf<uint8_t> d_desc;              // the per-slice descriptor tables, one block, one upload
	void* desc_staging = nullptr;        // pinned host image of d_desc (host_out_alloc), kept until the session dies
	uint32_t* host_flags = nullptr;      // pinned: the runs' per-slice error words + overflow word land here
	bool host_flags_pinned = false;
	bool flags_by_resolve = false;       // this run: k_slice_resolve wrote them there itself (ResolveArgs::host_flags)
	bool stage_events = true;            // HIP events between the kernels (ckl_decoder_stage_timing); off: only around the pipeline
	bool pending_upload = false;         // decoder_build left copies in flight that no run has waited for yet (ckl_decoder_destroy waits)
	bool stream_resident = false;        // the stream was in HBM already: capacities come from the z-index alone
	DevBuf<uint64_t> d_code_off, d_cbase, d_nbase, d_comp_off, d_rbase;
	DevBuf<uint32_t> d_code_len, d_ccap, d_ncap, d_rcap;
	DevBuf<uint8_t> d_model, d_ctl_kind;
	DevBuf<uint32_t> d_symbuf;
	DevBuf<uint64_t> d_symbase;
	DevBuf<uint32_t> d_mkscratch;
	DevBuf<uint64_t> d_mkbase;
	DevBuf<uint32_t> d_upacked, d_ctl_dx, d_ctl_dy, d_ctl_lastT, d_seg_x, d_seg_y, d_nodes;
	DevBuf<int32_t> d_ctl_depth, d_ctl_gmin;
	DevBuf<unsigned long long> d_ctl_link;
	uint32_t lds_controls = 0;          // capacity of k_decode_cracks' LDS control tables
	size_t lds_bytes = 0;               // dynamic LDS of k_decode_cracks: the tables, or everything the workgroup may have (raster bands)
	DevBuf<uint32_t> d_planes;          // V then H
	DevBuf<uint32_t> d_word_base, d_parent, d_run_start, d_run_cc, d_nruns, d_ncomp, d_ncomp_expect, d_blk_roots;
	DevBuf<uint16_t> d_run_local;
	DevBuf<uint64_t> d_run_label;       // typed on use (1..8 bytes per run)
	DevBuf<uint64_t> d_stats_table;     // ckl_decoder_label_stats: sorted label values
	DevBuf<unsigned long long> d_stats_acc;
	DevBuf<uint32_t> d_stats_box;
	std::vector<uint64_t> stats_table;
	bool bg_unlisted = false;           // pin stream whose background colour is not in its unique list
	std::shared_ptr<DevBuf<uint32_t>> G;      // geometric-sum table of the slice size, shared by the sessions of a device (geom_table)
	DevBuf<uint32_t> d_crc_acc, d_crc_expect, d_slice_err;
	DevBuf<uint64_t> d_label_map;
	DevBuf<uint64_t> d_pin_index, d_pin_depth, d_pin_label, d_pin_work_off, d_ccl_id, d_ccl_label;
	// strip path (ckl_strips.hpp): one slot of strip_cap entries per strip
	DevBuf<uint32_t> d_strip_nruns, d_strip_nsc, d_overflow, d_sc_w, d_sc_cc;
	DevBuf<uint16_t> d_seam_first, d_seam_last, d_row_run, d_run_lid;
	DevBuf<uint64_t> d_sc_label;        // typed on use
	DevBuf<unsigned long long> d_diag;
	bool strip_ok = false;              // shape / layout qualify for the strip path
	// crack records (ckl_crack_records.hpp): the strip path's front end
	bool use_records = false;           // k_crack_records + rasterising strip kernel instead of k_decode_cracks
	uint32_t resolve_cap = 12288;       // strip components of a slice k_slice_resolve's table holds (dynamic LDS)
	uint32_t resolve_cap_max = 12288;   // with a CU's LDS to itself: a run whose slices overflow resolve_cap is repeated with this one before the general pipeline is asked
	bool ran_fused = false;             // the last run went through k_strip_fused
	bool use_fused = false;             // k_strip_fused (flat labels on the record path): strips, resolve and paint in one launch
	DevBuf<uint32_t> d_fused_ctl;       // heads[8], timeout, pad to kFusedCtlWords, arrive[nslices], ready[nslices]
	DevBuf<uint32_t> d_fused_state;     // [strips][kFusedStateWords]: the paint tables of every strip
	DevBuf<uint32_t> d_seam32;          // [2][strips][row_words]
	DevBuf<uint4> d_rec;
	DevBuf<uint32_t> d_rec_count;
	DevBuf<uint4> d_words;             // WordRec per word of 16 code positions, parked between the two passes of k_crack_match
	DevBuf<uint64_t> d_word_off;
	uint32_t max_words = 0;             // most words of one slice
	uint32_t rec_cap = 0, rec_lds_controls = 0;
	int rec_block = kRecBlock;          // threads of k_crack_match's workgroups
	size_t rec_lds = 0;
	const uint64_t* foreign_label_map = nullptr;   // array_equal: component -> label table of ANOTHER stream (same component counts)
	int paint_width = 0;                // array_equal: bytes per painted voxel when it is not this stream's data width
	std::vector<uint32_t> ncomp_expect_host;       // components per slice of the range, as the label section states them
	bool use_general = false;           // a run overflowed the strip path's LDS tables: stay on the general pipeline
	uint32_t strip_rows = 0, nstrips = 0, strip_cap = 0;
	uint64_t rtot = 0;                  // entries of the general pipeline's per-run arrays (allocated when it runs)
	static constexpr int kMaxChunks = 8;
	hipStream_t chunk_stream[kMaxChunks] = {};
	hipEvent_t chunk_done[kMaxChunks] = {};
	hipEvent_t ev_fork = nullptr;

	// label section layout
	uint64_t total_comp = 0;            // components in [z_start, z_end)
	uint64_t comp_left = 0;             // global id of the first component of z_start
	uint64_t keys_offset = 0, uniq_offset = 0, num_unique = 0;
	int key_width = 1;
	uint64_t bgcolor = 0;
	uint64_t n_pins = 0, pin_total_work = 0, n_ccl = 0;

	uint32_t row_words = 0;
	uint64_t plane_words = 0;
	uint32_t max_rcap = 0;
	uint32_t max_comp = 1;              // most components of one slice in the range
	uint32_t idbits = 1, crc_fix = 0;
	bool check_crc = true;

	~ckl_decoder() {
		for (auto& e : ev) if (e) (void)hipEventDestroy(e);
		if (ev_in) (void)hipEventDestroy(ev_in);
		if (ev_fork) (void)hipEventDestroy(ev_fork);
		for (auto& e : chunk_done) if (e) (void)hipEventDestroy(e);
		for (auto& cs : chunk_stream) if (cs) (void)hipStreamDestroy(cs);
		if (stream) (void)hipStreamDestroy(stream);
		if (desc_staging) host_out_free(desc_staging);
		if (host_flags) host_out_free(host_flags);
	}
};

namespace {

template <typename T>
void upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t s) {
	d.ensure(h.size());
	if (!h.empty()) CKL_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

// The per-slice tables of a session go to the device as ONE block: every table is staged at its offset
// of a pinned host image and the DevBufs become views of the block (a dozen separate copies from
// pageable vectors cost more than the rest of ckl_decoder_create together).
struct DescPacker {
	struct Item { void* buf; size_t off, bytes, count; void (*bind)(void*, uint8_t*, size_t); };
	std::vector<Item> items;
	std::vector<uint8_t> image;
	template <typename T>
	void add(DevBuf<T>& dst, const std::vector<T>& src) {
		const size_t off = (image.size() + 255) & ~static_cast<size_t>(255);
		image.resize(off + std::max<size_t>(src.size(), 1) * sizeof(T));
		if (!src.empty()) memcpy(image.data() + off, src.data(), src.size() * sizeof(T));
		items.push_back({ &dst, off, src.size() * sizeof(T), src.size(),
			[](void* b, uint8_t* base, size_t n) { static_cast<DevBuf<T>*>(b)->borrow(reinterpret_cast<T*>(base), n); } });
	}
	// by_kernel: see upload_small (flat label streams; pin streams upload their tables with the copy engines anyway)
	void commit(ckl_decoder& d, hipStream_t s, bool by_kernel) {
		if (image.empty()) return;
		d.d_desc.ensure(image.size());
		if (d.desc_staging) host_out_free(d.desc_staging);
		d.desc_staging = host_out_alloc(std::max<size_t>(image.size(), 64u << 10));      // >= 64 KiB: pinned
		memcpy(d.desc_staging, image.data(), image.size());
		upload_small(d.d_desc.p, d.desc_staging, image.size(), s, by_kernel ? d.desc_staging : nullptr);
		for (const Item& it : items) it.bind(it.buf, d.d_desc.p + it.off, it.count);
	}
};

// G[m] = x^32 + ... + x^(32 m) for m <= pixels of a slice (ckl_runs.hpp): 4 bytes per pixel, built by
// one kernel launch and a stream sync.  Sessions are created per call by the one-shot API, so the
// tables are kept per (device, slice size) for the life of the process, the four most recent ones.
std::shared_ptr<DevBuf<uint32_t>> geom_table(int device, uint64_t sxy, hipStream_t s) {
	static std::mutex mu;
	typedef std::vector<std::pair<std::pair<int, uint64_t>, std::shared_ptr<DevBuf<uint32_t>>>> Cache;
	static Cache& cache = *new Cache();      // never destroyed: no HIP calls from static destructors at exit
	std::lock_guard<std::mutex> lock(mu);
	for (size_t i = 0; i < cache.size(); i++) {
		if (cache[i].first == std::make_pair(device, sxy)) {
			auto hit = cache[i];
			cache.erase(cache.begin() + i);
			cache.push_back(hit);      // most recently used last
			return hit.second;
		}
	}
	const uint32_t npx = static_cast<uint32_t>(sxy);
	const uint32_t B = 1024, nblk = npx / B + 1;
	std::vector<uint32_t> g_base(B), blk_g(nblk), blk_x(nblk);
	const uint32_t X = gf_xpow(32);
	g_base[0] = 0;
	for (uint32_t i = 1; i < B; i++) g_base[i] = gf_mul(X, g_base[i - 1] ^ 0x80000000u);
	const uint32_t gB = gf_mul(X, g_base[B - 1] ^ 0x80000000u);   // G[B]
	const uint32_t XB = gf_xpow(32ull * B);
	blk_g[0] = 0; blk_x[0] = 0x80000000u;
	for (uint32_t k = 1; k < nblk; k++) {
		blk_g[k] = blk_g[k - 1] ^ gf_mul(blk_x[k - 1], gB);
		blk_x[k] = gf_mul(blk_x[k - 1], XB);
	}
	DevBuf<uint32_t> t_base, t_g, t_x;
	upload(t_base, g_base, s); upload(t_g, blk_g, s); upload(t_x, blk_x, s);
	auto tab = std::make_shared<DevBuf<uint32_t>>();
	tab->ensure(static_cast<size_t>(npx) + 1);
	hipLaunchKernelGGL(k_build_geom_table, dim3(npx / kBlock + 1), dim3(kBlock), 0, s, t_base.p, t_g.p, t_x.p, npx, tab->p);
	CKL_HIP(hipStreamSynchronize(s));
	if (cache.size() >= 4) cache.erase(cache.begin());
	cache.push_back({ { device, sxy }, tab });
	return tab;
}

uint64_t read_stored(const Header& h, const uint8_t* lb, uint64_t offset) {
	const int w = h.stored_data_width;
	uint64_t v = rd_le(lb + offset, w);
	if (h.is_signed && w < 8 && (v >> (8 * w - 1))) v |= ~0ull << (8 * w);
	return v;
}

// buf: the stream on the host — all of it, or (stream_device given: the stream is resident in HBM already
// and stays the caller's) an image that holds header, z-index, label section head, markov model and crc
// tail at their offsets and nothing of the crack codes.
// a kernel's dynamic LDS beyond the default 64 KiB has to be allowed once per device and function
void allow_dynamic_lds(const void* fn, int device, size_t bytes) {
	if (bytes <= 48u * 1024u) return;
	static std::mutex mu;
	static std::vector<std::tuple<const void*, int, size_t>>& done = *new std::vector<std::tuple<const void*, int, size_t>>();
	std::lock_guard<std::mutex> lock(mu);
	for (auto& e : done) if (std::get<0>(e) == fn && std::get<1>(e) == device && std::get<2>(e) >= bytes) return;
	CKL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
	done.emplace_back(fn, device, bytes);
}

void decoder_build(ckl_decoder& d, const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, const uint8_t* stream_device = nullptr) {
	if (n < Header::kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
	d.head = Header::parse(buf, n);
	const Header& h = d.head;
	// range clamp (crackle.hpp:527-537)
	int64_t zs = z_start, ze = z_end;
	zs = std::max<int64_t>(std::min<int64_t>(zs, static_cast<int64_t>(h.sz) - 1), 0);
	ze = ze < 0 ? static_cast<int64_t>(h.sz) : ze;
	ze = std::max<int64_t>(std::min<int64_t>(ze, static_cast<int64_t>(h.sz)), 0);
	if (zs >= ze) throw Error(CKL_ERR_RUNTIME, "crackle: Invalid range: " + std::to_string(zs) + " - " + std::to_string(ze));
	d.z_start = zs; d.z_end = ze;
	d.nslices = static_cast<uint32_t>(ze - zs);
	d.sxy = static_cast<uint64_t>(h.sx) * h.sy;
	d.n_bytes = n;
	if (d.sxy == 0) return;
	if (d.sxy >= (1ull << 31)) throw Error(CKL_ERR_ARG, "crackle_amd: slices of 2^31 or more pixels are not supported");

	const uint64_t hb = h.header_bytes(), gib = h.grid_index_bytes();
	const uint64_t tail = h.format_version == 0 ? 0 : 4ull * (static_cast<uint64_t>(h.sz) + 1);
	if (!h.layout_fits(n)) {      // no sum of untrusted fields that could wrap
		throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_code_offsets: Unable to read past end of buffer.");
	}
	// z-index (crackle.hpp:262-313)
	if (h.format_version > 0) {
		const uint32_t stored = static_cast<uint32_t>(rd_le(buf + hb + 4ull * h.sz, 4));
		const uint32_t computed = crc32c(buf + hb, 4ull * h.sz);
		if (stored != computed) {
			throw Error(CKL_ERR_CRC, "crackle: grid index crc32c did not match. stored: " + std::to_string(stored) + " computed: " + std::to_string(computed));
		}
	}
	std::vector<uint64_t> z_index(static_cast<size_t>(h.sz) + 1);
	z_index[0] = hb + gib + h.num_label_bytes + h.markov_model_bytes();
	for (uint64_t z = 0; z < h.sz; z++) z_index[z + 1] = z_index[z] + rd_le(buf + hb + 4 * z, 4);
	if (z_index[h.sz] > n || tail > n - z_index[h.sz]) throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_codes: Unable to read past end of buffer.");

	hipStream_t s = d.stream;
	DescPacker pack;
	d.stream_resident = stream_device != nullptr;
	if (stream_device) d.d_stream.borrow(const_cast<uint8_t*>(stream_device), n);
	else {
		// the whole stream goes to HBM once
		d.d_stream.ensure(n + 16);
		CKL_HIP(hipMemcpyAsync(d.d_stream.p, buf, n, hipMemcpyHostToDevice, s));
	}

	// per-slice descriptors and scratch layout
	const int xw = byte_width(static_cast<uint64_t>(h.sx) + 1);
	const bool permissible = h.crack_format == PERMISSIBLE;
	std::vector<uint64_t> code_off(d.nslices), cbase(d.nslices), nbase(d.nslices), rbase(d.nslices);
	std::vector<uint32_t> code_len(d.nslices), ccap(d.nslices), ncap(d.nslices), rcap(d.nslices);
	uint64_t ctot = 0, ntot = 0, rtot = 0;
	double est_codes = 0;      // markov: ~1.4 bits per code on typical streams
	d.max_rcap = 0;
	for (uint32_t zi = 0; zi < d.nslices; zi++) {
		const uint64_t z = static_cast<uint64_t>(zs) + zi;
		const uint64_t len = z_index[z + 1] - z_index[z];
		if (len > 0xFFFFFFF0ull / 8) throw Error(CKL_ERR_RUNTIME, "crackle_amd: crack code of a slice is too large");
		code_off[zi] = z_index[z];
		code_len[zi] = static_cast<uint32_t>(len);
		// (a resident stream's BOC indices are not looked at here: capacities as if the code were all payload)
		const uint64_t index_size = (len >= 4 && !stream_device) ? rd_le(buf + z_index[z], 4) : 0;
		const uint64_t payload = (len >= 4 + index_size) ? len - 4 - index_size : 0;
		// codes: 4 per byte (plain) or at most 8 per byte (+1 raw) for the markov bitstream
		const uint64_t cap = (h.markov_model_order ? payload * 8 + 1 : payload * 4) + 2;
		const uint64_t nodes_cap = std::min<uint64_t>(stream_device ? len : index_size, len) / xw + 1;
		// runs: one per row plus one per vertical crack move (IMPERMISSIBLE), else up to one per pixel
		const uint64_t runs_cap = permissible ? d.sxy : std::min<uint64_t>(d.sxy, static_cast<uint64_t>(h.sy) + cap);
		cbase[zi] = ctot; ccap[zi] = static_cast<uint32_t>(cap); ctot += cap;
		est_codes += h.markov_model_order ? payload * 8 / 1.4 : payload * 4.0;
		nbase[zi] = ntot; ncap[zi] = static_cast<uint32_t>(nodes_cap); ntot += nodes_cap;
		rbase[zi] = rtot; rcap[zi] = static_cast<uint32_t>(runs_cap); rtot += runs_cap;
		d.max_rcap = std::max<uint32_t>(d.max_rcap, static_cast<uint32_t>(runs_cap));
	}
	pack.add(d.d_code_off, code_off);
	pack.add(d.d_code_len, code_len);
	pack.add(d.d_cbase, cbase);
	pack.add(d.d_ccap, ccap);
	pack.add(d.d_nbase, nbase);
	pack.add(d.d_ncap, ncap);
	pack.add(d.d_rbase, rbase);
	pack.add(d.d_rcap, rcap);
	std::vector<uint64_t> symbase(d.nslices, 0);
	{
		// symbols of multi-tile slices: (4 words per packed word + 3 counts) per thread and tile
		uint64_t stot = 0;
		for (uint32_t zi = 0; zi < d.nslices; zi++) {
			symbase[zi] = stot;
			const uint64_t tiles = static_cast<uint64_t>(ccap[zi]) / kCrackTile + 1;
			if (tiles > 1) stot += tiles * (4 * kCrackWords + 3) * kCrackBlock;
		}
		pack.add(d.d_symbase, symbase);
		d.d_symbuf.ensure(stot + 4);
	}
	std::vector<uint64_t> mkbase;
	if (h.markov_model_order) {
		d.d_upacked.ensure(ctot / 16 + 2ull * d.nslices + 4);
		mkbase.assign(d.nslices, 0);
		uint64_t mtot = 0;
		for (uint32_t zi = 0; zi < d.nslices; zi++) { mkbase[zi] = mtot; mtot += (code_len[zi] + 3ull) / 4 + 2 + ccap[zi] / 16 + 2; }
		pack.add(d.d_mkbase, mkbase);
		d.d_mkscratch.ensure(mtot + 4);
	}
	{
		// control symbol tables (global fallback of the LDS tables): a control symbol takes two codes
		const size_t ktot = ctot / 2 + 4ull * d.nslices + 8;
		d.d_ctl_kind.ensure(ktot); d.d_ctl_dx.ensure(ktot); d.d_ctl_dy.ensure(ktot);
		d.d_ctl_depth.ensure(ktot); d.d_ctl_lastT.ensure(ktot); d.d_ctl_link.ensure(ktot);
		d.d_seg_x.ensure(ktot); d.d_seg_y.ensure(ktot); d.d_ctl_gmin.ensure(ktot);
	}
	d.d_nodes.ensure(ntot);
	d.rtot = rtot;      // the general pipeline's per-run arrays are allocated when it runs

	if (h.markov_model_order) {
		std::vector<uint8_t> model = markov_model_from_stored(buf + hb + gib + h.num_label_bytes, h.markov_model_bytes(), h.markov_model_order);
		pack.add(d.d_model, model);
	}

	d.row_words = (h.sx + 31) / 32;
	d.plane_words = static_cast<uint64_t>(d.row_words) * h.sy;
	d.d_planes.ensure(2 * d.plane_words * d.nslices);
	{
		// strip path: x-fastest output in groups of 4 pixels, strips of <= 1024 plane words
		// strips of <= 1024 plane words, and few enough rows that the runs expected from the length of
		// the crack codes (one per row plus one per vertical move, about half of the moves) fill ~0.7
		// of the strip kernels' LDS tables; a strip that still overflows falls back (ckl_strips.hpp)
		d.strip_rows = std::max<uint32_t>(1u, kStripWords / d.row_words);
		if (!permissible && d.nslices) {
			const double runs_per_row = 1.0 + 0.5 * est_codes / (static_cast<double>(d.nslices) * h.sy);
			const uint32_t fit = static_cast<uint32_t>(0.7 * kStripCap / runs_per_row);
			d.strip_rows = std::max<uint32_t>(1u, std::min(d.strip_rows, fit));
		}
		if (const char* env = getenv("CKL_CCL_ROWS")) d.strip_rows = std::max<uint32_t>(1u, std::min<uint32_t>(kStripWords / d.row_words, static_cast<uint32_t>(std::max(1, atoi(env)))));
		d.nstrips = (h.sy + d.strip_rows - 1) / d.strip_rows;
		d.strip_ok = h.fortran_order && (h.sx % 4 == 0) && d.row_words <= kStripWords && d.nstrips <= kMaxStrips &&
			d.sxy < 0xFFFF0000ull && !getenv("CKL_DECODE_GENERAL");
		if (d.strip_ok) {
			// per-slice kernels (k_slice_resolve, k_crack_match): two workgroups share a CU's LDS while a decode has more
			// slices than the chip has CUs; with fewer (C4: 32 slices of 2048 x 2048 per GPU) one takes nearly all of it
			const bool few = d.nslices <= static_cast<uint32_t>(std::max(1, d.n_cus));
			const size_t lds_share = few ? static_cast<size_t>(d.max_lds) - 8192u : static_cast<size_t>(d.max_lds) / 2u - 6144u;
			d.resolve_cap = static_cast<uint32_t>(std::min<size_t>(kResolveCap, (lds_share - (kMaxStrips + 64u) * 4u) / 4u));
			// (a workgroup alone on its CU: what an over-segmented slice gets, whatever the slice count — see below the label section)
			d.resolve_cap_max = static_cast<uint32_t>(std::min<size_t>(kResolveCap, (static_cast<size_t>(d.max_lds) - 8192u - (kMaxStrips + 64u) * 4u) / 4u));
			if (few && !getenv("CKL_LDS_CONTROLS")) {
				uint32_t nctl = 16384;
				while (nctl > 64 && rec_lds_bytes(nctl) > lds_share) nctl -= 64;
				if (rec_lds_bytes(nctl) <= lds_share && nctl > d.rec_lds_controls) {
					d.rec_lds_controls = nctl;
					d.rec_lds = lds_share & ~static_cast<size_t>(15);
					d.rec_block = getenv("CKL_REC_NARROW") ? kRecBlock : kRecBlockWide;      // a slice has its CU to itself: 16 wavefronts instead of 8
					if (d.rec_block == kRecBlockWide) allow_dynamic_lds(reinterpret_cast<const void*>(&k_crack_match<kRecBlockWide>), d.device, d.rec_lds);
					else allow_dynamic_lds(reinterpret_cast<const void*>(&k_crack_match<kRecBlock>), d.device, d.rec_lds);
				}
			}
			const size_t nst = static_cast<size_t>(d.nstrips) * d.nslices;
			d.strip_cap = static_cast<uint32_t>(std::min<uint64_t>(kStripCap, static_cast<uint64_t>(d.strip_rows) * h.sx));
			d.d_row_run.ensure(static_cast<size_t>(h.sy) * d.nslices);
			d.d_strip_nruns.ensure(nst); d.d_strip_nsc.ensure(nst);
			d.d_seam_first.ensure(nst * d.row_words); d.d_seam_last.ensure(nst * d.row_words);
			d.d_run_lid.ensure(nst * d.strip_cap + 16); d.d_sc_w.ensure(nst * d.strip_cap);      // (+16: the paint's 8-byte loads of the last strip's last components)
			d.d_sc_label.ensure(nst * d.strip_cap);
			if (h.label_format != FLAT) d.d_sc_cc.ensure(nst * d.strip_cap);
			d.d_overflow.ensure(1);
			// crack records: vertices packed 16 + 16 bits, one LDS cursor per strip
			d.use_records = h.sx <= kRecMaxDim && h.sy <= kRecMaxDim && d.nstrips <= kRecMaxStrips && d.rec_lds_controls > 0 && !getenv("CKL_DECODE_RASTER");
			if (d.use_records) {
				// a strip's list: one record per 16 codes that touch it, twice that for the records that
				// straddle two strips and the stretches cut by 't' jumps; a list that still overflows
				// sends the session to the rasterising kernel
				const double per_strip = est_codes / 16.0 / (static_cast<double>(d.nslices) * d.nstrips);
				uint32_t cap = static_cast<uint32_t>(std::min<double>(per_strip * 3.0 + 256.0, 65536.0));
				if (const char* env = getenv("CKL_REC_CAP")) cap = static_cast<uint32_t>(std::max(1, atoi(env)));      // testing: forces the overflow hand-over
				d.rec_cap = cap;
				d.d_rec.ensure(nst * cap);
				d.d_rec_count.ensure(nst);
				// words of 16 code positions: k_crack_match deals a slice's codes out in tiles of rec_block x <= rec_words(rec_block) words
				std::vector<uint64_t> word_off(d.nslices);
				uint64_t wtot = 0;
				d.max_words = 0;
				for (uint32_t zi = 0; zi < d.nslices; zi++) {
					const uint32_t nw = (ccap[zi] / rec_tile(d.rec_block) + 1u) * static_cast<uint32_t>(d.rec_block) * rec_words(d.rec_block);
					word_off[zi] = wtot; wtot += nw;
					d.max_words = std::max(d.max_words, nw);
				}
				pack.add(d.d_word_off, word_off);
				d.d_words.ensure(wtot);
				// one launch for strips + resolve + paint (k_strip_fused): flat labels, a slice's strips resident together
				// (opt-in: measured slower than the three launches at C2)
				d.use_fused = h.label_format == FLAT && d.nstrips <= kFusedMaxStrips && getenv("CKL_DECODE_FUSED") != nullptr;
				if (d.use_fused) {
					d.d_fused_ctl.ensure(kFusedCtlWords + 2 * static_cast<size_t>(d.nslices));
					d.d_fused_state.ensure(nst * kFusedStateWords);
					d.d_seam32.ensure(2 * nst * d.row_words);
					d.d_sc_cc.ensure(nst * d.strip_cap);      // lid32 (the pins' table is free on the flat path)
				}
			}
		}
	}
	d.d_ncomp.ensure(d.nslices);
	d.d_slice_err.ensure(d.nslices);
	d.d_crc_acc.ensure(d.nslices);

	// label section layout (labels.hpp:424-451, 453-617), parsed once
	struct PinBlock { uint8_t* p = nullptr; hipStream_t s = nullptr; ~PinBlock() { if (p) { (void)hipStreamSynchronize(s); host_out_free(p); } } } pin_block;      // the pin tables on their way to the device
	pin_block.s = s;
	const uint8_t* lb = buf + hb + gib;
	const uint64_t nlb = h.num_label_bytes;
	const int sw = h.stored_data_width;
	const int component_width = byte_width(d.sxy);
	uint64_t offset;
	if (h.label_format == FLAT) {
		if (nlb < 8) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
		d.num_unique = rd_le(lb, 8);
		d.uniq_offset = 8;
		if (d.num_unique > (nlb - 8) / sw) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
		offset = 8 + static_cast<uint64_t>(sw) * d.num_unique;
	}
	else if (h.label_format == PINS_VARIABLE_WIDTH) {
		if (nlb < static_cast<uint64_t>(sw) + 8) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		d.bgcolor = read_stored(h, lb, 0);
		d.num_unique = rd_le(lb + sw, 8);
		d.uniq_offset = static_cast<uint64_t>(sw) + 8;
		if (d.num_unique > (nlb - 8 - static_cast<uint64_t>(sw)) / sw) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		offset = 8 + static_cast<uint64_t>(sw) * (d.num_unique + 1);
	}
	else {
		throw Error(CKL_ERR_RUNTIME, "crackle: Unsupported label format. Got: " + std::to_string(h.label_format));
	}
	if (offset > nlb || static_cast<uint64_t>(component_width) * h.sz > nlb - offset) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	std::vector<uint64_t> comp_prefix(static_cast<size_t>(h.sz) + 1, 0);
	for (uint64_t z = 0; z < h.sz; z++) {
		comp_prefix[z + 1] = comp_prefix[z] + rd_le(lb + offset + z * component_width, component_width);
	}
	offset += static_cast<uint64_t>(component_width) * h.sz;
	d.comp_left = comp_prefix[zs];
	d.total_comp = comp_prefix[ze] - comp_prefix[zs];
	std::vector<uint64_t> comp_off(d.nslices);
	std::vector<uint32_t> ncomp_expect(d.nslices);
	uint32_t max_comp = 1;
	for (uint32_t zi = 0; zi < d.nslices; zi++) {
		comp_off[zi] = comp_prefix[zs + zi] - comp_prefix[zs];
		const uint64_t c = comp_prefix[zs + zi + 1] - comp_prefix[zs + zi];
		if (c > d.sxy) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
		ncomp_expect[zi] = static_cast<uint32_t>(c);
		max_comp = std::max<uint32_t>(max_comp, static_cast<uint32_t>(c));
	}
	pack.add(d.d_comp_off, comp_off);
	pack.add(d.d_ncomp_expect, ncomp_expect);
	d.ncomp_expect_host = ncomp_expect;
	if (d.strip_ok && d.resolve_cap < d.resolve_cap_max) {
		// Over-segmented slices (the reference's watershed benchmark: 16 k segments per 1024 x 1024 slice): a component
		// meets one to two strips, so a slice has up to ~2 strip components per component.  Where that passes what two
		// workgroups per CU can hold, k_slice_resolve gets a CU's LDS to itself (two rounds of workgroups for more
		// slices than CUs) instead of overflowing into the general run pipeline at a tenth of the speed.
		uint32_t most = 0;
		for (uint32_t c : ncomp_expect) most = std::max(most, c);
		if (2ull * most > d.resolve_cap) d.resolve_cap = d.resolve_cap_max;
	}
	d.d_label_map.ensure(d.total_comp + 1);

	// crc machinery: geometric-sum table G[m] = x^32 + ... + x^(32 m), component ids are
	// multiplied in over their `idbits` significant bits only (see k_run_resolve)
	d.check_crc = h.format_version > 0;
	d.idbits = 1;
	d.max_comp = max_comp;
	while (d.idbits < 32 && (1ull << d.idbits) < max_comp) d.idbits++;
	d.crc_fix = gf_xpow(32 - d.idbits);
	d.G = geom_table(d.device, d.sxy, s);
	if (d.check_crc) {
		// stored = ~(x^(32 n) * 0xFFFFFFFF ^ raw)  =>  raw = ~stored ^ init_term
		const uint32_t init_term = gf_mul(0xFFFFFFFFu, gf_xpow(32ull * d.sxy));
		std::vector<uint32_t> expect(d.nslices);
		const uint8_t* crcs = buf + n - 4ull * h.sz;
		for (uint32_t zi = 0; zi < d.nslices; zi++) {
			const uint32_t stored = static_cast<uint32_t>(rd_le(crcs + 4ull * (zs + zi), 4));
			expect[zi] = (~stored) ^ init_term;
		}
		pack.add(d.d_crc_expect, expect);
	}
	else {
		d.d_crc_expect.ensure(d.nslices);
	}

	if (h.label_format == FLAT) {
		d.key_width = byte_width(d.num_unique);
		d.keys_offset = hb + gib + offset;
		if (offset > nlb || comp_prefix[h.sz] > (nlb - offset) / static_cast<uint64_t>(d.key_width)) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	}
	else {
		if (offset + 1 > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		const uint8_t combined = lb[offset++];
		const int npw = 1 << (combined & 3), dw = 1 << ((combined >> 2) & 3), ccw = 1 << ((combined >> 4) & 3);
		const int iw = h.pin_index_width();
		// The records are of variable length: one sequential pass finds where each label's record starts (two reads
		// per label), then the labels are decoded side by side on the worker threads into one pinned block
		// (host_out_alloc: cached) that is uploaded as it stands.  Nothing is filtered out on the host: a pin that
		// does not touch the decoded slices gets no work items (k_label_map_pins' search for the pin of a work item
		// takes the LAST pin that starts at or before it, so pins without work are never found), and
		// k_label_map_ccids checks the ids' range itself.  (C4, 1.6 M pins + ids: 42 ms -> a few, serial with six
		// growing vectors and pageable uploads before.)
		struct PinRec { uint64_t at, num_pins, num_cc; };
		const uint64_t nu = d.num_unique;
		std::vector<PinRec> recs(nu);
		std::vector<uint64_t> pin_base(nu + 1, 0), cc_base(nu + 1, 0), work_base(nu + 1, 0);
		{
			uint64_t i = offset;
			for (uint64_t label = 0; label < nu; label++) {
				if (i + npw > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
				const uint64_t num_pins = rd_le(lb + i, npw);
				if (num_pins > nlb || i + npw + num_pins * static_cast<uint64_t>(iw + dw) + npw > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
				const uint64_t j = i + npw + num_pins * static_cast<uint64_t>(iw + dw);
				const uint64_t num_cc = rd_le(lb + j, npw);
				if (num_cc > nlb || j + npw + num_cc * static_cast<uint64_t>(ccw) > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
				recs[label] = { i, num_pins, num_cc };
				pin_base[label + 1] = pin_base[label] + num_pins;
				cc_base[label + 1] = cc_base[label] + num_cc;
				i = j + npw + num_cc * static_cast<uint64_t>(ccw);
			}
			// The records must end where the section does.  They do not for the sections the reference's encoder writes when a
			// label has more than 255 single-component ids while no label has 256 pins (the count shares the pins' field width
			// and overflows it: labels.hpp:209-229, tools/repro_pins_u8.py): the reference then returns labels read out of
			// step; this decoder refuses.
			if (i != nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted (its records do not end with the section: a count field overflowed?).");
		}
		const uint64_t P = pin_base[nu], C = cc_base[nu];
		pin_block.p = static_cast<uint8_t*>(host_out_alloc((4 * std::max<uint64_t>(P, 1) + 2 * std::max<uint64_t>(C, 1)) * 8));
		uint64_t* const pin_index = reinterpret_cast<uint64_t*>(pin_block.p);
		uint64_t* const pin_depth = pin_index + std::max<uint64_t>(P, 1);
		uint64_t* const pin_label = pin_depth + std::max<uint64_t>(P, 1);
		uint64_t* const pin_work_off = pin_label + std::max<uint64_t>(P, 1);
		uint64_t* const ccl_id = pin_work_off + std::max<uint64_t>(P, 1);
		uint64_t* const ccl_label = ccl_id + std::max<uint64_t>(C, 1);
		const uint64_t volume = d.sxy * h.sz;
		host_parallel_for(nu, 512, [&](size_t lo, size_t hi) {
			for (size_t label = lo; label < hi; label++) {
				const PinRec& r = recs[label];
				const uint64_t lv = read_stored(h, lb, d.uniq_offset + label * sw);
				const uint8_t* at = lb + r.at + npw;
				uint64_t idx = 0, work = 0;
				for (uint64_t j = 0; j < r.num_pins; j++) {
					idx += rd_le(at + j * iw, iw);
					const uint64_t depth = rd_le(at + r.num_pins * iw + j * dw, dw);
					const int64_t pin_z = static_cast<int64_t>(idx / d.sxy);
					const int64_t a = std::max<int64_t>(pin_z, zs);
					const int64_t b = std::min<int64_t>(pin_z + static_cast<int64_t>(depth) + 1, ze);
					const bool touches = idx < volume && b > a;      // else: the pin does not touch the decoded range
					const uint64_t o = pin_base[label] + j;
					pin_index[o] = touches ? idx : 0; pin_depth[o] = touches ? depth : 0; pin_label[o] = lv;
					pin_work_off[o] = touches ? static_cast<uint64_t>(b - a) : 0;      // its work items for now, their offset below
					work += pin_work_off[o];
				}
				work_base[label + 1] = work;
				const uint8_t* ids = at + r.num_pins * static_cast<uint64_t>(iw + dw) + npw;
				uint64_t id = 0;
				for (uint64_t j = 0; j < r.num_cc; j++) {
					id = (id + rd_le(ids + j * ccw, ccw)) & 0xFFFFFFFFull;
					ccl_id[cc_base[label] + j] = id; ccl_label[cc_base[label] + j] = lv;
				}
			}
		});
		for (uint64_t label = 0; label < nu; label++) work_base[label + 1] += work_base[label];
		host_parallel_for(nu, 2048, [&](size_t lo, size_t hi) {
			for (size_t label = lo; label < hi; label++) {
				uint64_t run = work_base[label];
				for (uint64_t o = pin_base[label]; o < pin_base[label + 1]; o++) { const uint64_t w = pin_work_off[o]; pin_work_off[o] = run; run += w; }
			}
		});
		d.n_pins = P;
		d.pin_total_work = work_base[nu];
		d.n_ccl = C;
		d.d_pin_index.ensure(std::max<uint64_t>(P, 1)); d.d_pin_depth.ensure(std::max<uint64_t>(P, 1)); d.d_pin_label.ensure(std::max<uint64_t>(P, 1));
		d.d_pin_work_off.ensure(std::max<uint64_t>(P, 1)); d.d_ccl_id.ensure(std::max<uint64_t>(C, 1)); d.d_ccl_label.ensure(std::max<uint64_t>(C, 1));
		if (P) {
			CKL_HIP(hipMemcpyAsync(d.d_pin_index.p, pin_index, P * 8, hipMemcpyHostToDevice, s));
			CKL_HIP(hipMemcpyAsync(d.d_pin_depth.p, pin_depth, P * 8, hipMemcpyHostToDevice, s));
			CKL_HIP(hipMemcpyAsync(d.d_pin_label.p, pin_label, P * 8, hipMemcpyHostToDevice, s));
			CKL_HIP(hipMemcpyAsync(d.d_pin_work_off.p, pin_work_off, P * 8, hipMemcpyHostToDevice, s));
		}
		if (C) {
			CKL_HIP(hipMemcpyAsync(d.d_ccl_id.p, ccl_id, C * 8, hipMemcpyHostToDevice, s));
			CKL_HIP(hipMemcpyAsync(d.d_ccl_label.p, ccl_label, C * 8, hipMemcpyHostToDevice, s));
		}
	}
	pack.commit(d, s, h.label_format == FLAT);
	// the caller's host stream and the pin tables above are copied from memory that is not ours to keep
	if (!stream_device || h.label_format != FLAT) CKL_HIP(hipStreamSynchronize(s));
	else d.pending_upload = true;      // (the descriptors' packed copy is still on its way: the first run orders itself behind it, ckl_decoder_destroy waits for it)
}

struct StageTimer {
	ckl_decoder& d;
	hipStream_t s;
	int i = 0;
	bool on = true;      // off: z-chunks overlap on several streams, only the whole pipeline is timed
	StageTimer(ckl_decoder& dec, hipStream_t st) : d(dec), s(st) { on = dec.stage_events; CKL_HIP(hipEventRecord(d.ev[0], s)); }
	void done(const char* name) {
		if (!on || i >= kMaxStages) return;
		d.stage_name[i] = name;
		CKL_HIP(hipEventRecord(d.ev[i + 1], s));
		i++;
	}
};

// component ids (ckl_runs.hpp), component -> label, run -> label, paint.  Flat labels: the
// label table is ready before the components are, so k_run_assign writes the run labels
// directly; pins need the component ids first (k_label_map_pins looks pixels up).
// component -> label table of a pin stream (labels.hpp:540-650); needs run_cc
void launch_pin_label_map(ckl_decoder& d, const RunGeom& g, const RunArrays& ra, const StripArrays* sa = nullptr) {
	hipStream_t s = d.stream;
	const uint64_t nlm = d.total_comp;
	if (nlm) hipLaunchKernelGGL(k_fill_u64, dim3(static_cast<uint32_t>((nlm + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d.d_label_map.p, d.bgcolor, nlm);
	if (d.n_ccl) hipLaunchKernelGGL(k_label_map_ccids, dim3(static_cast<uint32_t>((d.n_ccl + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
		d.d_ccl_id.p, d.d_ccl_label.p, d.n_ccl, d.comp_left, d.comp_left + nlm, d.d_label_map.p);
	if (!d.pin_total_work) return;
	const dim3 grid(static_cast<uint32_t>((d.pin_total_work + kBlock - 1) / kBlock));
	if (sa) hipLaunchKernelGGL(k_label_map_pins_strips, grid, dim3(kBlock), 0, s,
		d.d_pin_index.p, d.d_pin_depth.p, d.d_pin_label.p, d.d_pin_work_off.p, d.n_pins, d.pin_total_work,
		g, *sa, d.sxy, d.z_start, d.z_end, d.d_comp_off.p, d.d_ncomp_expect.p, d.d_label_map.p);
	else hipLaunchKernelGGL(k_label_map_pins, grid, dim3(kBlock), 0, s,
		d.d_pin_index.p, d.d_pin_depth.p, d.d_pin_label.p, d.d_pin_work_off.p, d.n_pins, d.pin_total_work,
		g, ra, d.sxy, d.z_start, d.z_end, d.d_comp_off.p, d.d_ncomp_expect.p, d.d_label_map.p);
}

void launch_flat_label_map(ckl_decoder& d) {
	const Header& h = d.head;
	const uint64_t nlm = d.total_comp;
	if (!nlm) return;
	const uint8_t* keys = d.d_stream.p + d.keys_offset + d.comp_left * static_cast<uint64_t>(d.key_width);
	const uint8_t* uniq = d.d_stream.p + h.header_bytes() + h.grid_index_bytes() + d.uniq_offset;
	hipLaunchKernelGGL(k_label_map_flat, dim3(static_cast<uint32_t>((nlm + kBlock - 1) / kBlock)), dim3(kBlock), 0, d.stream,
		keys, d.key_width, uniq, h.stored_data_width, d.num_unique, h.is_signed ? 1u : 0u, nlm, d.d_label_map.p);
}

// component ids, component -> label, then the statistics kernel instead of the paint
void launch_resolve_and_stats(ckl_decoder& d, const RunGeom& g, const RunArrays& ra, StageTimer& st, const StatsArgs* stats) {
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	ResolveScratch rs;
	rs.run_local = d.d_run_local.p; rs.blk_roots = d.d_blk_roots.p; rs.nblk = (d.max_rcap + kBlock - 1) / kBlock;
	hipLaunchKernelGGL(k_run_count, dim3(run_count_blocks(rs.nblk), ns), dim3(kBlock), 0, s, ra, rs);
	st.done("k_run_count");
	hipLaunchKernelGGL(k_run_rank, dim3(ns), dim3(kBlock), 0, s, ra, rs, d.idbits, d.d_crc_acc.p, static_cast<uint32_t*>(nullptr));
	st.done("k_run_rank");
	RunLabelArgs none = {};
	hipLaunchKernelGGL((k_run_assign<uint8_t, false>), dim3(run_assign_blocks(rs.nblk), ns), dim3(kBlock), 0, s, ra, rs, d.G->p, static_cast<uint32_t>(d.sxy), d.idbits, d.d_crc_acc.p, none);
	st.done("k_run_assign");
	if (d.head.label_format == FLAT) launch_flat_label_map(d);
	else launch_pin_label_map(d, g, ra);
	st.done("k_label_map");
	if (!stats) return;      // integrity check only: component counts and crcs are in, k_check follows
	hipLaunchKernelGGL(k_run_stats, dim3(ns), dim3(kStatsBlock), static_cast<size_t>(stats->lds_comps) * kStatsBytesPerComp, s,
		ra, d.d_label_map.p, d.d_comp_off.p, d.d_ncomp_expect.p, *stats);
	st.done("k_run_stats");
}

template <typename OUT>
void launch_resolve_and_paint(ckl_decoder& d, const RunGeom& g, const RunArrays& ra, void* out_device, int has_label, uint64_t label, StageTimer& st) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	OUT* run_label = reinterpret_cast<OUT*>(d.d_run_label.p);
	const bool foreign = d.foreign_label_map != nullptr;      // the table of another stream: nothing to build
	const bool flat = h.label_format == FLAT || foreign;
	if (flat && !foreign) launch_flat_label_map(d);
	if (flat) st.done("k_label_map");
	ResolveScratch rs;
	rs.run_local = d.d_run_local.p; rs.blk_roots = d.d_blk_roots.p; rs.nblk = (d.max_rcap + kBlock - 1) / kBlock;
	hipLaunchKernelGGL(k_run_count, dim3(run_count_blocks(rs.nblk), ns), dim3(kBlock), 0, s, ra, rs);
	st.done("k_run_count");
	hipLaunchKernelGGL(k_run_rank, dim3(ns), dim3(kBlock), 0, s, ra, rs, d.idbits, d.d_crc_acc.p, static_cast<uint32_t*>(nullptr));
	st.done("k_run_rank");
	RunLabelArgs la;
	la.label_map = foreign ? d.foreign_label_map : d.d_label_map.p; la.comp_off = d.d_comp_off.p; la.ncomp_expect = d.d_ncomp_expect.p;
	la.has_label = has_label ? 1u : 0u; la.label = label; la.run_label = run_label;
	if (flat) {
		hipLaunchKernelGGL((k_run_assign<OUT, true>), dim3(run_assign_blocks(rs.nblk), ns), dim3(kBlock), 0, s, ra, rs, d.G->p, static_cast<uint32_t>(d.sxy), d.idbits, d.d_crc_acc.p, la);
		st.done("k_run_assign");
	}
	else {
		hipLaunchKernelGGL((k_run_assign<OUT, false>), dim3(run_assign_blocks(rs.nblk), ns), dim3(kBlock), 0, s, ra, rs, d.G->p, static_cast<uint32_t>(d.sxy), d.idbits, d.d_crc_acc.p, la);
		st.done("k_run_assign");
		launch_pin_label_map(d, g, ra);
		st.done("k_label_map");
		hipLaunchKernelGGL(k_run_labels<OUT>, dim3((d.max_rcap + kBlock - 1) / kBlock, ns), dim3(kBlock), 0, s,
			ra, d.d_label_map.p, d.d_comp_off.p, d.d_ncomp_expect.p, has_label ? 1u : 0u, label, run_label);
		st.done("k_run_labels");
	}
	const uint32_t tiles = static_cast<uint32_t>((d.sxy + kPaintTile - 1) / kPaintTile);
	const bool fast = h.fortran_order && (h.sx % 4 == 0) && d.sxy < 0xFFFF0000ull;   // 32-bit pixel arithmetic in the fast path
	if (fast) hipLaunchKernelGGL((k_paint_runs<OUT, true>), dim3(tiles, ns), dim3(kBlock), 0, s, g, ra, run_label, reinterpret_cast<OUT*>(out_device), d.sxy, ns, getenv("CKL_PAINT_NOSTAGE") ? 3u : 1u);
	else hipLaunchKernelGGL((k_paint_runs<OUT, false>), dim3(tiles, ns), dim3(kBlock), 0, s, g, ra, run_label, reinterpret_cast<OUT*>(out_device), d.sxy, ns, h.fortran_order ? 1u : 0u);
	st.done("k_paint_runs");
}

// ---- the strip path (ckl_strips.hpp): planes -> labels in three kernels per z-chunk ----
struct StripPlan {
	StripArrays sa;
	ResolveArgs ra;
};

StripPlan strip_plan(ckl_decoder& d, int has_label, uint64_t label) {
	StripPlan p;
	StripArrays& sa = p.sa;
	sa.run_lid = d.d_run_lid.p; sa.sc_w = d.d_sc_w.p; sa.sc_cc = d.d_sc_cc.p; sa.sc_label = d.d_sc_label.p;
	sa.strip_nruns = d.d_strip_nruns.p; sa.strip_nsc = d.d_strip_nsc.p; sa.row_run = d.d_row_run.p;
	sa.seam_first = d.d_seam_first.p; sa.seam_last = d.d_seam_last.p;
	sa.seam32_first = d.d_seam32.p; sa.seam32_last = d.d_seam32.p ? d.d_seam32.p + static_cast<size_t>(d.nstrips) * d.nslices * d.row_words : nullptr;
	sa.lid32 = d.d_sc_cc.p;
	sa.slice_err = d.d_slice_err.p; sa.overflow = d.d_overflow.p;
	sa.nstrips = d.nstrips; sa.strip_rows = d.strip_rows; sa.cap = d.strip_cap; sa.zbase = 0;
	sa.ablate = 0;
	if (kTuning) if (const char* env = getenv("CKL_ABLATE")) sa.ablate = static_cast<uint32_t>(strtoul(env, nullptr, 0));
	sa.layout = 3u;      // XCD-contiguous strips, a quarter of the strip per wavefront in the paint (CKL_STRIP_LAYOUT=0: launch order, KiB by KiB)
	if (const char* env = getenv("CKL_STRIP_LAYOUT")) sa.layout = static_cast<uint32_t>(strtoul(env, nullptr, 0));
	ResolveArgs& ra = p.ra;
	ra.idbits = d.idbits; ra.crc_fix = d.crc_fix; ra.check_crc = d.check_crc ? 1u : 0u;
	ra.crc_expect = d.d_crc_expect.p; ra.ncomp_expect = d.d_ncomp_expect.p; ra.comp_off = d.d_comp_off.p;
	ra.label_map = d.d_label_map.p; ra.has_label = has_label ? 1u : 0u; ra.label = label;
	{
		const Header& h = d.head;
		ra.keys = d.d_stream.p + d.keys_offset + d.comp_left * static_cast<uint64_t>(d.key_width);
		ra.uniq = d.d_stream.p + h.header_bytes() + h.grid_index_bytes() + d.uniq_offset;
		ra.key_width = static_cast<uint32_t>(d.key_width); ra.stored_width = static_cast<uint32_t>(h.stored_data_width);
		ra.is_signed = h.is_signed ? 1u : 0u; ra.num_unique = d.num_unique;
	}
	ra.host_flags = nullptr; ra.host_flags_n = d.nslices;
	ra.cap = d.resolve_cap;
	if (const char* env = getenv("CKL_RESOLVE_CAP")) ra.cap = std::min<uint32_t>(d.resolve_cap, static_cast<uint32_t>(std::max(1, atoi(env))));   // testing: forces the overflow path
	return p;
}

void launch_cracks(ckl_decoder& d, hipStream_t s, CrackArgs ca, uint32_t z0, uint32_t n, size_t crack_lds) {
	ca.zbase = z0;
	hipLaunchKernelGGL(k_decode_cracks<false>, dim3(n), dim3(kCrackBlock), crack_lds, s, ca, static_cast<unsigned long long*>(nullptr));
}

template <typename OUT>
void launch_paint_strips(ckl_decoder& d, hipStream_t s, const RunGeom& g, const StripPlan& p, uint32_t n, void* out_device, unsigned long long* diag) {
	const dim3 grid(d.nstrips, n);
	if constexpr (kTuning) {
		if (diag) { hipLaunchKernelGGL((k_paint_strips<OUT, true>), grid, dim3(kBlock), 0, s, g, p.sa, reinterpret_cast<OUT*>(out_device), static_cast<uint32_t>(d.sxy), diag + 16); return; }
	}
	// Two workgroups per CU, not eight: the write path is saturated by 8 wavefronts per CU (a pure fill takes its 0.334 ms
	// at C2 with 12 or with 32), and every further workgroup in flight only lengthens the queues its siblings' loads wait
	// in (0.370 ms at 8 per CU, 0.346 at 2, 0.352 at 1).  The limit is dynamic LDS nobody uses: just over a third of a
	// CU's LDS per workgroup.  Grids too small to fill the chip that way keep the full occupancy.  CKL_PAINT_WGS=n (0: no limit).
	size_t pad = 0;
	uint32_t wgs = 2;
	if (const char* env = getenv("CKL_PAINT_WGS")) wgs = static_cast<uint32_t>(std::max(0, atoi(env)));
	// (only where the strips are painted by four independent wavefronts — paint_strips_waves_body: rows of whole plane
	// words, four of them per lane load, strips of a multiple of four rows —: the one-barrier-per-phase body of other
	// shapes wants its eight workgroups: 0.42 against 0.67 ms on 13-row strips of an over-segmented volume)
	const bool waves_body = (p.sa.layout & 2u) && (d.row_words & 3u) == 0u && d.head.sx == d.row_words * 32u && (d.strip_rows & 3u) == 0u;
	if (wgs && waves_body && static_cast<uint64_t>(d.nstrips) * n >= 4096u) {
		const size_t own = paint_strips_words<OUT>() * sizeof(uint32_t);
		const size_t want = static_cast<size_t>(d.max_lds) / (wgs + 1u) + 1024u;
		if (want > own && want <= static_cast<size_t>(d.max_lds)) {
			pad = (want - own + 15u) & ~static_cast<size_t>(15);
			allow_dynamic_lds(reinterpret_cast<const void*>(&k_paint_strips<OUT, false>), d.device, pad);
		}
	}
	hipLaunchKernelGGL((k_paint_strips<OUT, false>), grid, dim3(kBlock), pad, s, g, p.sa, reinterpret_cast<OUT*>(out_device), static_cast<uint32_t>(d.sxy), static_cast<unsigned long long*>(nullptr));
}

// strips + resolve of slices [z0, z0 + n); flat labels also paint
RecordLists record_lists(ckl_decoder& d) {
	RecordLists rl;
	rl.rec = d.d_rec.p; rl.count = d.d_rec_count.p; rl.cap = d.rec_cap;
	rl.nstrips = d.nstrips; rl.strip_rows = d.strip_rows;
	rl.strip_shift = 0xFFFFFFFFu;
	for (uint32_t b = 0; b < 32; b++) if ((1u << b) == d.strip_rows) rl.strip_shift = b;
	rl.strip_magic = static_cast<uint32_t>((1ull << 32) / std::max<uint32_t>(2u, d.strip_rows)) + 1u;      // (rows = 1 is a power of two: the shift)
	return rl;
}

// crack codes -> the strips' record lists (ckl_crack_records.hpp); also clears the session's error words
void launch_crack_records(ckl_decoder& d, hipStream_t s, CrackArgs ca, uint32_t z0, uint32_t n, StageTimer* st) {
	RecArgs ra;
	ca.zbase = z0;
	ra.c = ca;
	ra.lists = record_lists(d);
	ra.lds_controls = d.rec_lds_controls;
	ra.lds_bytes = static_cast<uint32_t>(d.rec_lds);
	ra.words = reinterpret_cast<WordRec*>(d.d_words.p);
	ra.word_base = d.d_word_off.p;
	ra.fused_ctl = (d.use_fused && z0 == 0 && n == d.nslices) ? d.d_fused_ctl.p : nullptr;      // k_strip_fused's counters start from zero
	ra.fused_n = d.nslices;
	ra.fused_ctl_words = kFusedCtlWords;
	ra.diag = nullptr;
	ra.ablate = 0;
	if (kTuning) if (const char* env = getenv("CKL_ABLATE")) ra.ablate = static_cast<uint32_t>(strtoul(env, nullptr, 0));
	if (kTuning && getenv("CKL_CRACK_DIAG")) {
		d.d_diag.ensure(64);
		CKL_HIP(hipMemsetAsync(d.d_diag.p + 32, 0, 32 * sizeof(unsigned long long), s));
		ra.diag = d.d_diag.p + 32;
	}
	if (d.rec_block == kRecBlockWide) hipLaunchKernelGGL(k_crack_match<kRecBlockWide>, dim3(n), dim3(kRecBlockWide), d.rec_lds, s, ra);
	else hipLaunchKernelGGL(k_crack_match<kRecBlock>, dim3(n), dim3(kRecBlock), d.rec_lds, s, ra);
	if (st) st->done("k_crack_match");
	if (ra.diag) {
		unsigned long long hd[32];
		CKL_HIP(hipMemcpyAsync(hd, ra.diag, sizeof(hd), hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
		fprintf(stderr, "[ckl crack diag, mean cycles per slice] k_crack_match: boc=%.0f symbols=%.0f record=%.0f match=%.0f (depth=%.0f tree=%.0f links=%.0f jump=%.0f seg=%.0f) records=%.0f\n",
			hd[0] / double(n), hd[1] / double(n), hd[2] / double(n), hd[3] / double(n), hd[4] / double(n), hd[5] / double(n), hd[6] / double(n), hd[7] / double(n), hd[8] / double(n), hd[9] / double(n));
		fprintf(stderr, "[ckl crack diag] tile_symbols: load+sums=%.0f scan1=%.0f r+lfscan=%.0f ctrl=%.0f events+scan3=%.0f | records: direct=%.0f queued=%.0f counts=%.0f\n",
			hd[10] / double(n), hd[11] / double(n), hd[12] / double(n), hd[13] / double(n), hd[14] / double(n), hd[15] / double(n), hd[16] / double(n), hd[9] / double(n));
		fprintf(stderr, "[ckl crack diag] links per slice: local pops=%.0f searches=%.0f (steps %.0f) chain ends=%.0f\n", hd[19] / double(n), hd[17] / double(n), hd[18] / double(n), hd[20] / double(n));
	}
}

template <typename OUT>
void launch_strips(ckl_decoder& d, hipStream_t s, const RunGeom& g, StripPlan p, uint32_t z0, uint32_t n, void* out_device, bool flat, StageTimer* st, unsigned long long* diag, bool records) {
	p.sa.zbase = z0;
	const uint32_t npx = static_cast<uint32_t>(d.sxy);
	const RecordLists rl = record_lists(d);
	bool launched = false;
	// k_strip_ccl2 (ckl_strips2.hpp): rows of 4 .. 512 plane words, a power of two; CKL_STRIP_V1 keeps k_strip_ccl
	uint32_t rsh = 0;
	while ((1u << rsh) < d.row_words) rsh++;
	const bool v2 = records && (1u << rsh) == d.row_words && rsh >= 2 && rsh <= 9 && !getenv("CKL_STRIP_V1");
	if (v2) {
		Strip2Args a2;
		a2.rsh = rsh; a2.variant = 0;
		if (const char* env = getenv("CKL_STRIP_VARIANT")) a2.variant = static_cast<uint32_t>(atoi(env));
		const bool edgelist = (a2.variant & 1u) != 0u;
		if constexpr (kTuning) {
			if (diag) {
				// per-workgroup stamps: raw s_memtime at the phase boundaries
				const size_t nwg = static_cast<size_t>(d.nstrips) * n;
				DevBuf<unsigned long long> stamps;
				stamps.ensure(nwg * kStrip2Stamps);
				CKL_HIP(hipMemsetAsync(stamps.p, 0, nwg * kStrip2Stamps * sizeof(unsigned long long), s));
				if (edgelist) hipLaunchKernelGGL((k_strip_ccl2<true, true>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, a2, stamps.p);
				else hipLaunchKernelGGL((k_strip_ccl2<true, false>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, a2, stamps.p);
				std::vector<unsigned long long> hs(nwg * kStrip2Stamps);
				CKL_HIP(hipMemcpyAsync(hs.data(), stamps.p, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
				CKL_HIP(hipStreamSynchronize(s));
				double ph[kStrip2Stamps] = {};
				size_t cnt = 0;
				unsigned long long t_first = ~0ull, t_last = 0;
				for (size_t w = 0; w < nwg; w++) {
					const unsigned long long* q = hs.data() + w * kStrip2Stamps;
					if (!q[0] || !q[6]) continue;      // overflowed strip
					for (uint32_t i = 1; i <= 6; i++) ph[i] += static_cast<double>(q[i] - q[i - 1]);
					t_first = std::min(t_first, q[0]); t_last = std::max(t_last, q[6]);
					cnt++;
				}
				const double c = cnt ? static_cast<double>(cnt) : 1.0;
				fprintf(stderr, "[ckl strip2 diag, mean cycles per workgroup, %zu workgroups] raster=%.0f read+scan=%.0f starts=%.0f unions=%.0f roots+ids=%.0f weights=%.0f | lifetime=%.0f span(raw)=%.0f\n",
					cnt, ph[1] / c, ph[2] / c, ph[3] / c, ph[4] / c, ph[5] / c, ph[6] / c, (ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6]) / c, static_cast<double>(t_last - t_first));
				launched = true;
			}
		}
		if (!launched) {
			size_t pad = 0;      // tuning: dynamic LDS nobody uses, to see what fewer workgroups per CU cost (CKL_STRIP_PAD bytes)
			if (kTuning) if (const char* env = getenv("CKL_STRIP_PAD")) pad = static_cast<size_t>(atoi(env));
			if (kTuning && edgelist) hipLaunchKernelGGL((k_strip_ccl2<false, true>), dim3(d.nstrips, n), dim3(kBlock), pad, s, g, p.sa, rl, d.G->p, npx, a2, static_cast<unsigned long long*>(nullptr));
			else hipLaunchKernelGGL((k_strip_ccl2<false, false>), dim3(d.nstrips, n), dim3(kBlock), pad, s, g, p.sa, rl, d.G->p, npx, a2, static_cast<unsigned long long*>(nullptr));
			launched = true;
		}
	}
	if constexpr (kTuning) {
		if (diag && !launched) {
			if (records) hipLaunchKernelGGL((k_strip_ccl<true, true>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, diag);
			else hipLaunchKernelGGL((k_strip_ccl<true, false>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, diag);
			launched = true;
		}
	}
	if (!launched) {
		if (records) hipLaunchKernelGGL((k_strip_ccl<false, true>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, static_cast<unsigned long long*>(nullptr));
		else hipLaunchKernelGGL((k_strip_ccl<false, false>), dim3(d.nstrips, n), dim3(kBlock), 0, s, g, p.sa, rl, d.G->p, npx, static_cast<unsigned long long*>(nullptr));
	}
	if (st) st->done(v2 ? "k_strip_ccl2" : "k_strip_ccl");
	const size_t tab_bytes = static_cast<size_t>(p.ra.cap) * sizeof(uint32_t);      // k_slice_resolve's table
	if (flat) {
		bool done = false;
		if constexpr (kTuning) {
			if (diag) {
				allow_dynamic_lds(reinterpret_cast<const void*>(&k_slice_resolve<OUT, true, true>), d.device, tab_bytes);
				hipLaunchKernelGGL((k_slice_resolve<OUT, true, true>), dim3(n), dim3(kResolveBlock), tab_bytes, s, g, p.sa, p.ra, d.d_ncomp.p, diag + 8); done = true;
			}
		}
		if (!done) {
			allow_dynamic_lds(reinterpret_cast<const void*>(&k_slice_resolve<OUT, true, false>), d.device, tab_bytes);
			hipLaunchKernelGGL((k_slice_resolve<OUT, true, false>), dim3(n), dim3(kResolveBlock), tab_bytes, s, g, p.sa, p.ra, d.d_ncomp.p, static_cast<unsigned long long*>(nullptr));
		}
	}
	else {
		allow_dynamic_lds(reinterpret_cast<const void*>(&k_slice_resolve<OUT, false, false>), d.device, tab_bytes);
		hipLaunchKernelGGL((k_slice_resolve<OUT, false, false>), dim3(n), dim3(kResolveBlock), tab_bytes, s, g, p.sa, p.ra, d.d_ncomp.p, static_cast<unsigned long long*>(nullptr));
	}
	if (st) st->done("k_slice_resolve");
	if (!flat) return;
	launch_paint_strips<OUT>(d, s, g, p, n, out_device, diag);
	if (st) st->done("k_paint_strips");
}

// number of z-chunks the strip path pipelines over its streams
uint32_t decode_chunks(const ckl_decoder& d) {
	uint32_t want = 1;      // measured at C2: 4 chunks on 4 streams 1.39 ms against 1.22 ms on one stream
	if (const char* env = getenv("CKL_DECODE_CHUNKS")) want = static_cast<uint32_t>(std::max(1, atoi(env)));
	want = std::min<uint32_t>(want, ckl_decoder::kMaxChunks);
	// a chunk should still fill the chip: at least 64 Mi voxels and 64 slices each
	const uint64_t by_size = std::max<uint64_t>(1, d.sxy * d.nslices >> 26);
	const uint64_t by_slices = std::max<uint32_t>(1u, d.nslices / 64u);
	if (!getenv("CKL_DECODE_CHUNKS")) want = static_cast<uint32_t>(std::min<uint64_t>(want, std::min(by_size, by_slices)));
	return std::max<uint32_t>(1u, std::min(want, d.nslices));
}

template <typename OUT>
void strip_pipeline(ckl_decoder& d, const CrackArgs& ca, size_t crack_lds, const RunGeom& g, const RunArrays& ra_legacy, void* out_device, int has_label, uint64_t label, StageTimer& st) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	const bool flat = h.label_format == FLAT;
	StripPlan p = strip_plan(d, has_label, label);
	const uint32_t chunks = decode_chunks(d);
	// flat labels in one chunk: k_slice_resolve settles every slice's error word and the overflow word, and writes them
	// to the host's mapped memory itself (no k_flags_to_host launch behind the paint)
	d.flags_by_resolve = flat && chunks <= 1 && d.host_flags_pinned && !(d.use_records && d.use_fused) && !(kTuning && getenv("CKL_STRIP_DIAG")) && !getenv("CKL_FLAGS_KERNEL");
	if (d.flags_by_resolve) {
		d.host_flags[ns] = 0u; d.host_flags[ns + 1] = 0u; d.host_flags[ns + 2] = 0u;
		p.ra.host_flags = d.host_flags;
	}
	if (chunks > 1) CKL_HIP(hipMemsetAsync(d.d_overflow.p, 0, sizeof(uint32_t), s));      // one chunk: the crack kernel clears it
	unsigned long long* diag = nullptr;
	if (kTuning && getenv("CKL_STRIP_DIAG")) {      // tuning builds: cycle stamps of the strip kernels (adds a sync and a print)
		d.d_diag.ensure(64);
		CKL_HIP(hipMemsetAsync(d.d_diag.p, 0, 32 * sizeof(unsigned long long), s));
		diag = d.d_diag.p;
	}
	const bool records = d.use_records;
	if (chunks <= 1 || diag) {
		CrackArgs ca1 = ca;
		if (chunks <= 1) ca1.overflow = d.d_overflow.p;
		if (records) launch_crack_records(d, s, ca1, 0, ns, &st);
		else { launch_cracks(d, s, ca1, 0, ns, crack_lds); st.done("k_decode_cracks"); }
		if (records && flat && d.use_fused && chunks <= 1 && !diag) {
			// strips, resolve and paint in one launch (k_strip_fused); its counters were zeroed by k_crack_match
			FusedCtl fc;
			uint32_t* ctl = d.d_fused_ctl.p;
			fc.heads = ctl; fc.timeout = ctl + kFusedMaxHeads * kFusedHeadStride; fc.arrive = ctl + kFusedCtlWords; fc.ready = ctl + kFusedCtlWords + ns;
			fc.state = d.d_fused_state.p; fc.nslices = ns;
			fc.nheads = std::min<uint32_t>(kFusedMaxHeads, ns);
			fc.lag = 12;      // slices of one counter between a strip's labelling and its paint: covers label + resolve at C2 (tools: CKL_FUSED_LAG)
			if (const char* env = getenv("CKL_FUSED_LAG")) fc.lag = static_cast<uint32_t>(std::max(1, atoi(env)));
			if (const char* env = getenv("CKL_FUSED_HEADS")) fc.nheads = std::max<uint32_t>(1u, std::min<uint32_t>(fc.nheads, static_cast<uint32_t>(atoi(env))));
			p.sa.zbase = 0;
#ifdef CKL_TUNING
			{
				const uint32_t ex = getenv("CKL_EXP") ? static_cast<uint32_t>(strtoul(getenv("CKL_EXP"), nullptr, 0)) : 0u;
				CKL_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(ckl::dev::g_exp_flags), &ex, sizeof(ex), 0, hipMemcpyHostToDevice, s));
			}
#endif
			hipLaunchKernelGGL(k_strip_fused<OUT>, dim3(2u * d.nstrips * ns), dim3(kBlock), 0, s, g, p.sa, record_lists(d), p.ra, fc, d.G->p,
				static_cast<uint32_t>(d.sxy), d.d_ncomp.p, reinterpret_cast<OUT*>(out_device), static_cast<uint32_t>(d.sxy));
			st.done("k_strip_fused");
			d.ran_fused = true;
		}
		else launch_strips<OUT>(d, s, g, p, 0, ns, out_device, flat, &st, diag, records);
		if (diag) {
			unsigned long long hd[32];
			CKL_HIP(hipMemcpyAsync(hd, diag, sizeof(hd), hipMemcpyDeviceToHost, s));
			CKL_HIP(hipStreamSynchronize(s));
			const double nst = static_cast<double>(d.nstrips) * ns;
			fprintf(stderr, "[ckl strip diag, mean cycles per workgroup] k_strip_ccl: raster=%.0f load+scan=%.0f starts=%.0f unions=%.0f roots+ids=%.0f weights=%.0f | k_slice_resolve: tables=%.0f seams=%.0f rank=%.0f labels+crc=%.0f | k_paint_strips: words+scan=%.0f labels=%.0f paint=%.0f\n",
				hd[5] / nst, hd[0] / nst, hd[1] / nst, hd[2] / nst, hd[3] / nst, hd[4] / nst, hd[8] / double(ns), hd[9] / double(ns), hd[10] / double(ns), hd[11] / double(ns), hd[16] / nst, hd[17] / nst, hd[18] / nst);
		}
	}
	else {
		// z-chunks on their own streams: the instruction-bound crack and strip kernels of one chunk
		// run beside the bandwidth-bound paint of another
		st.on = false;
		if (!d.ev_fork) CKL_HIP(hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming));
		CKL_HIP(hipEventRecord(d.ev_fork, s));
		const uint32_t per = (ns + chunks - 1) / chunks;
		for (uint32_t c = 0; c < chunks; c++) {
			const uint32_t z0 = c * per;
			if (z0 >= ns) break;
			const uint32_t n = std::min(per, ns - z0);
			if (!d.chunk_stream[c]) CKL_HIP(hipStreamCreateWithFlags(&d.chunk_stream[c], hipStreamNonBlocking));
			if (!d.chunk_done[c]) CKL_HIP(hipEventCreateWithFlags(&d.chunk_done[c], hipEventDisableTiming));
			hipStream_t cs = d.chunk_stream[c];
			const bool same_stream = getenv("CKL_DECODE_CHUNKS_SERIAL") != nullptr;      // experiment: the chunks one after the other on the session's stream
			if (same_stream) cs = s;
			else CKL_HIP(hipStreamWaitEvent(cs, d.ev_fork, 0));
			if (records) launch_crack_records(d, cs, ca, z0, n, nullptr);
			else launch_cracks(d, cs, ca, z0, n, crack_lds);
			launch_strips<OUT>(d, cs, g, p, z0, n, out_device, flat, nullptr, nullptr, records);
			if (!same_stream) {
				CKL_HIP(hipEventRecord(d.chunk_done[c], cs));
				CKL_HIP(hipStreamWaitEvent(s, d.chunk_done[c], 0));
			}
		}
	}
	if (!flat) {
		// pins: the label table needs the component ids of every slice a pin pierces
		launch_pin_label_map(d, g, ra_legacy, &p.sa);
		st.done("k_label_map_pins");
		p.sa.zbase = 0;
		hipLaunchKernelGGL(k_strip_labels<OUT>, dim3(d.nstrips, ns), dim3(kBlock), 0, s, p.sa, p.ra);
		st.done("k_strip_labels");
		launch_paint_strips<OUT>(d, s, g, p, ns, out_device, nullptr);
		st.done("k_paint_strips");
	}
}

// the general run pipeline (ckl_runs.hpp) from planes that are already in HBM
void general_pipeline(ckl_decoder& d, const RunGeom& g, RunArrays& ra, void* out_device, int has_label, uint64_t label, const StatsArgs* stats, bool check_only, StageTimer& st) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	// per-run arrays of the general pipeline (a session on the strip path never needs them)
	d.d_word_base.ensure(d.plane_words * ns);
	d.d_parent.ensure(d.rtot); d.d_run_start.ensure(d.rtot); d.d_run_cc.ensure(d.rtot); d.d_run_local.ensure(d.rtot);
	d.d_run_label.ensure(d.rtot); d.d_nruns.ensure(ns);
	d.d_blk_roots.ensure(static_cast<size_t>((d.max_rcap + kBlock - 1) / kBlock) * ns);
	ra.word_base = d.d_word_base.p; ra.parent = d.d_parent.p; ra.run_start = d.d_run_start.p; ra.run_cc = d.d_run_cc.p; ra.nruns = d.d_nruns.p;
	hipLaunchKernelGGL(k_run_index, dim3(ns), dim3(kIndexBlock), 0, s, g, ra);
	st.done("k_run_index");
	{
		// launch_run_union / launch_run_resolve of ckl_runs.hpp, kernel by kernel for the stage timers
		const uint32_t rows = run_strip_rows(g.row_words);
		const uint32_t strips = (g.sy + rows - 1) / rows;
		const uint32_t sruns = run_strip_runs();
		hipLaunchKernelGGL(k_run_union_strips, dim3(strips, ns), dim3(kBlock), sruns * sizeof(uint32_t), s, g, ra, rows, sruns);
		st.done("k_run_union_strips");
		if (strips > 1) {
			const uint32_t words = (strips - 1) * g.row_words;
			hipLaunchKernelGGL(k_run_union_seams, dim3((words + kBlock - 1) / kBlock, ns), dim3(kBlock), 0, s, g, ra, rows);
		}
		st.done("k_run_union_seams");
	}
	if (stats || check_only) launch_resolve_and_stats(d, g, ra, st, stats);
	else {
		const int dw = d.paint_width ? d.paint_width : h.data_width;
		if (has_label || dw == 1) launch_resolve_and_paint<uint8_t>(d, g, ra, out_device, has_label, label, st);
		else if (dw == 2) launch_resolve_and_paint<uint16_t>(d, g, ra, out_device, has_label, label, st);
		else if (dw == 4) launch_resolve_and_paint<uint32_t>(d, g, ra, out_device, has_label, label, st);
		else launch_resolve_and_paint<uint64_t>(d, g, ra, out_device, has_label, label, st);
	}

	hipLaunchKernelGGL(k_check, dim3((ns + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
		d.d_crc_acc.p, d.d_crc_expect.p, d.d_ncomp.p, d.d_ncomp_expect.p,
		d.check_crc ? 1u : 0u, d.crc_fix, ns, d.d_slice_err.p);
	st.done("k_check");
}

void decoder_run(ckl_decoder& d, void* out_device, uint64_t out_capacity_bytes, int has_label, uint64_t label, const StatsArgs* stats = nullptr, bool planes_only = false, uint32_t* errs_out = nullptr) {
	const Header& h = d.head;
	if (d.sxy == 0 || d.nslices == 0) return;
	const int ow = has_label ? 1 : (d.paint_width ? d.paint_width : h.data_width);
	const uint64_t need = d.sxy * d.nslices * static_cast<uint64_t>(ow);
	if (!stats && !planes_only && !errs_out && out_capacity_bytes < need) throw Error(CKL_ERR_ARG, "crackle_amd: output buffer too small: need " + std::to_string(need) + " bytes");
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	const bool prof = getenv("CKL_PROFILE") != nullptr;
	auto h0 = std::chrono::steady_clock::now();

	// the verdicts come back through the host's pinned memory (a kernel's stores: a copy-engine transfer of
	// two KiB costs tens of microseconds before it starts)
	if (!d.host_flags) {
		d.host_flags = static_cast<uint32_t*>(host_out_alloc(std::max<size_t>(64u << 10, (static_cast<size_t>(ns) + 4) * sizeof(uint32_t))));
		d.host_flags_pinned = host_out_is_pinned(d.host_flags);
	}
	d.flags_by_resolve = false;
	StageTimer st(d, s);
	// k_decode_cracks builds the planes band by band in LDS when at least one row of both
	// planes fits behind the segment tables; otherwise it ORs bits into zeroed planes in HBM
	const size_t crack_lds = d.lds_bytes;
	const bool lds_raster = !getenv("CKL_NO_LDS_RASTER") &&
		crack_lds >= crack_lds_seg_bytes(d.lds_controls) + 2ull * d.row_words * sizeof(uint32_t);
	if (!lds_raster) CKL_HIP(hipMemsetAsync(d.d_planes.p, 0, 2 * d.plane_words * ns * sizeof(uint32_t), s));
	if (!lds_raster) st.done("memset");

	CrackArgs ca;
	ca.stream = d.d_stream.p;
	ca.code_off = d.d_code_off.p; ca.code_len = d.d_code_len.p;
	ca.cbase = d.d_cbase.p; ca.ccap = d.d_ccap.p; ca.nbase = d.d_nbase.p; ca.ncap = d.d_ncap.p;
	ca.sx = static_cast<int>(h.sx); ca.sy = static_cast<int>(h.sy);
	ca.xw = byte_width(static_cast<uint64_t>(h.sx) + 1); ca.yw = byte_width(static_cast<uint64_t>(h.sy) + 1);
	ca.markov_order = h.markov_model_order;
	ca.symbuf = d.d_symbuf.p; ca.symbase = d.d_symbase.p;
	ca.mkscratch = h.markov_model_order ? d.d_mkscratch.p : nullptr; ca.mkbase = d.d_mkbase.p;
	ca.model = d.d_model.p; ca.upacked = h.markov_model_order ? d.d_upacked.p : nullptr;
	ca.g_kind = d.d_ctl_kind.p; ca.g_dx = d.d_ctl_dx.p; ca.g_dy = d.d_ctl_dy.p;
	ca.g_depth = d.d_ctl_depth.p; ca.g_lastT = d.d_ctl_lastT.p; ca.g_link = d.d_ctl_link.p;
	ca.g_seg_x = d.d_seg_x.p; ca.g_seg_y = d.d_seg_y.p; ca.g_gmin = d.d_ctl_gmin.p;
	ca.nodes = d.d_nodes.p;
	ca.lds_controls = d.lds_controls;
	ca.lds_words = static_cast<uint32_t>(crack_lds / 4);
	ca.lds_raster = lds_raster ? 1u : 0u;
	ca.markov_serial = getenv("CKL_MARKOV_SERIAL") ? 1u : 0u;
	ca.zbase = 0;
	ca.planeV = d.d_planes.p; ca.planeH = d.d_planes.p + d.plane_words * ns;
	ca.row_words = d.row_words; ca.plane_words = d.plane_words;
	ca.slice_err = d.d_slice_err.p;
	ca.overflow = nullptr;

	RunGeom g;
	g.planeV = ca.planeV; g.planeH = ca.planeH; g.row_words = d.row_words; g.plane_words = d.plane_words;
	g.flip = (h.crack_format == IMPERMISSIBLE) ? 1u : 0u;
	g.sx = h.sx; g.sy = h.sy;
	RunArrays ra;
	ra.word_base = d.d_word_base.p; ra.rbase = d.d_rbase.p; ra.rcap = d.d_rcap.p;
	ra.parent = d.d_parent.p; ra.run_start = d.d_run_start.p; ra.run_cc = d.d_run_cc.p;
	ra.nruns = d.d_nruns.p; ra.ncomp = d.d_ncomp.p; ra.slice_err = d.d_slice_err.p;

	const bool paint = !stats && !planes_only && !errs_out;
	const bool strips = paint && d.strip_ok && !d.use_general && !d.foreign_label_map && !d.paint_width && !(kTuning && getenv("CKL_DECODE_DIAG"));
	if (strips) {
		// planes -> strips -> labels, z-chunk by z-chunk (ckl_strips.hpp)
		if (has_label || h.data_width == 1) strip_pipeline<uint8_t>(d, ca, crack_lds, g, ra, out_device, has_label, label, st);
		else if (h.data_width == 2) strip_pipeline<uint16_t>(d, ca, crack_lds, g, ra, out_device, has_label, label, st);
		else if (h.data_width == 4) strip_pipeline<uint32_t>(d, ca, crack_lds, g, ra, out_device, has_label, label, st);
		else strip_pipeline<uint64_t>(d, ca, crack_lds, g, ra, out_device, has_label, label, st);
	}
	else {
		bool diag_launched = false;
		if constexpr (kTuning) if (getenv("CKL_DECODE_DIAG")) {
			diag_launched = true;
			DevBuf<unsigned long long> d_diag;
			d_diag.ensure(static_cast<size_t>(ns) * 16);
			CKL_HIP(hipMemsetAsync(d_diag.p, 0, static_cast<size_t>(ns) * 128, s));
			hipLaunchKernelGGL(k_decode_cracks<true>, dim3(ns), dim3(kCrackBlock), crack_lds, s, ca, d_diag.p);
			std::vector<unsigned long long> dg(static_cast<size_t>(ns) * 16);
			CKL_HIP(hipMemcpyAsync(dg.data(), d_diag.p, dg.size() * 8, hipMemcpyDeviceToHost, s));
			CKL_HIP(hipStreamSynchronize(s));
			double m[16] = { 0 };
			for (uint32_t zi = 0; zi < ns; zi++) for (int k = 0; k < 16; k++) m[k] += static_cast<double>(dg[zi * 16 + k]) / ns;
			fprintf(stderr, "[ckl decode_cracks diag, mean cycles per slice] A(boc)=%.0f B(symbols+record)=%.0f C(match)=%.0f B.symbols=%.0f  codes=%.0f controls=%.0f | match: depth/lastT=%.0f gmin=%.0f links=%.0f jump=%.0f search steps total=%.0f max/thread=%.0f | raster: zero=%.0f symbols=%.0f moves=%.0f store=%.0f\n", m[0], m[1], m[2], m[3], m[4], m[5], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[6], m[7], m[15]);
		}
		if (!diag_launched) launch_cracks(d, s, ca, 0, ns, crack_lds);
		st.done("k_decode_cracks");

		if (planes_only) {
			// the crack planes are all the caller wants (ckl_reencode_markov): check what the parser flagged
			d.n_stages = st.i;
			std::vector<uint32_t> errs(ns);
			CKL_HIP(hipMemcpyAsync(errs.data(), d.d_slice_err.p, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
			CKL_HIP(hipStreamSynchronize(s));
			CKL_HIP(hipGetLastError());
			for (uint32_t zi = 0; zi < ns; zi++) {
				if (errs[zi]) throw Error(CKL_ERR_RUNTIME, "crackle: crack code is malformed or corrupted on z=" + std::to_string(d.z_start + zi));
			}
			return;
		}
		general_pipeline(d, g, ra, out_device, has_label, label, stats, errs_out != nullptr, st);
	}
	d.n_stages = st.i;
	CKL_HIP(hipEventRecord(d.ev[kMaxStages + 1], s));      // end of the pipeline (all chunk streams joined)

	auto h1 = std::chrono::steady_clock::now();
	std::vector<uint32_t> errs(ns);
	uint32_t overflow = 0, timeout = 0;
	const bool fused_ran = d.ran_fused;
	d.ran_fused = false;
	if (d.flags_by_resolve) {
		CKL_HIP(hipStreamSynchronize(s));
		memcpy(errs.data(), d.host_flags, ns * sizeof(uint32_t));
		overflow = (d.host_flags[ns] ? 1u : 0u) | (d.host_flags[ns + 2] ? 2u : 0u);      // (two words: each is only ever written with one constant)
	}
	else if (d.host_flags_pinned) {
		hipLaunchKernelGGL(k_flags_to_host, dim3((ns + kBlock) / kBlock), dim3(kBlock), 0, s, d.d_slice_err.p, ns, strips ? d.d_overflow.p : nullptr, fused_ran ? d.d_fused_ctl.p + kFusedMaxHeads * kFusedHeadStride : nullptr, d.host_flags);
		CKL_HIP(hipStreamSynchronize(s));
		memcpy(errs.data(), d.host_flags, ns * sizeof(uint32_t));
		overflow = d.host_flags[ns];
		timeout = d.host_flags[ns + 1];
	}
	else {
		CKL_HIP(hipMemcpyAsync(errs.data(), d.d_slice_err.p, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		if (strips) CKL_HIP(hipMemcpyAsync(&overflow, d.d_overflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		if (fused_ran) CKL_HIP(hipMemcpyAsync(&timeout, d.d_fused_ctl.p + kFusedMaxHeads * kFusedHeadStride, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
	}
	CKL_HIP(hipGetLastError());
	d.pending_upload = false;
	if (fused_ran && (timeout || overflow)) {
		// a wait inside k_strip_fused gave up (it never should), or a strip / slice did not fit its tables (the fused
		// launch leaves no planes in HBM for the general pipeline): this and all later runs of the session take the three launches
		d.use_fused = false;
		return decoder_run(d, out_device, out_capacity_bytes, has_label, label, stats, planes_only, errs_out);
	}
	if (strips && d.use_records) {
		// a record list that did not fit (dense slices): this and all later runs of the session take the
		// rasterising kernel; nothing of the run's output is kept
		bool list_over = false;
		for (uint32_t zi = 0; zi < ns; zi++) list_over = list_over || (errs[zi] & ERR_LIST);
		if (list_over) {
			d.use_records = false;
			return decoder_run(d, out_device, out_capacity_bytes, has_label, label, stats, planes_only, errs_out);
		}
	}
	if (strips && overflow == 2u && d.resolve_cap < d.resolve_cap_max && !getenv("CKL_RESOLVE_CAP")) {
		// only k_slice_resolve's table was too small (bit 1), and it can be larger: once more with a CU's LDS per slice
		d.resolve_cap = d.resolve_cap_max;
		return decoder_run(d, out_device, out_capacity_bytes, has_label, label, stats, planes_only, errs_out);
	}
	if (strips && overflow) {
		// a strip or a slice has more runs / strip components than the LDS tables of the strip path
		// hold (dense, noisy labels): the general pipeline takes over from the planes, for this and
		// all later runs of the session
		d.use_general = true;
		StageTimer st2(d, s);
		st2.on = st.on;
		general_pipeline(d, g, ra, out_device, has_label, label, nullptr, false, st2);
		d.n_stages = st2.i;
		CKL_HIP(hipEventRecord(d.ev[kMaxStages + 1], s));
		CKL_HIP(hipMemcpyAsync(errs.data(), d.d_slice_err.p, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
		CKL_HIP(hipGetLastError());
	}
	auto h2 = std::chrono::steady_clock::now();
	if (prof) fprintf(stderr, "[ckl decode host ms] enqueue=%.2f wait=%.2f\n",
		std::chrono::duration<double, std::milli>(h1 - h0).count(), std::chrono::duration<double, std::milli>(h2 - h1).count());
	CKL_HIP(hipEventElapsedTime(&d.pipeline_ms, d.ev[0], d.ev[kMaxStages + 1]));
	for (int i = 0; i < d.n_stages; i++) CKL_HIP(hipEventElapsedTime(&d.stage_ms[i], d.ev[i], d.ev[i + 1]));
	if (errs_out) {      // the caller wants the per-slice verdicts, not an exception
		for (uint32_t zi = 0; zi < ns; zi++) errs_out[zi] = errs[zi];
		return;
	}
	for (uint32_t zi = 0; zi < ns; zi++) {
		const uint32_t e = errs[zi];
		if (!e) continue;
		const std::string z = std::to_string(d.z_start + zi);
		if (e & (ERR_BOC | ERR_RANGE | ERR_CAPACITY)) throw Error(CKL_ERR_RUNTIME, "crackle: crack code is malformed or corrupted on z=" + z);
		if (e & ERR_NCOMP) throw Error(CKL_ERR_RUNTIME, "crackle: component count does not match the label section on z=" + z);
		throw Error(CKL_ERR_CRC, "crackle: crack code crc mismatch on z=" + z);
	}
}

// the stream's labels (the unique list of the label section, plus the background colour of a pin
// stream), as the label map holds them (sign-extended), ascending as unsigned: host and device copy
void ensure_label_table(ckl_decoder& d) {
	if (!d.stats_table.empty()) return;
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const int sw = h.stored_data_width;
	std::vector<uint8_t> raw(static_cast<size_t>(d.num_unique) * sw);
	const uint64_t at = h.header_bytes() + h.grid_index_bytes() + d.uniq_offset;
	if (!raw.empty()) CKL_HIP(hipMemcpyAsync(raw.data(), d.d_stream.p + at, raw.size(), hipMemcpyDeviceToHost, s));
	CKL_HIP(hipStreamSynchronize(s));
	std::vector<uint64_t> t(d.num_unique);
	for (uint64_t i = 0; i < d.num_unique; i++) t[i] = read_stored(h, raw.data(), i * sw);
	d.bg_unlisted = h.label_format != FLAT && std::find(t.begin(), t.end(), d.bgcolor) == t.end();
	if (h.label_format != FLAT) t.push_back(d.bgcolor);
	std::sort(t.begin(), t.end());
	t.erase(std::unique(t.begin(), t.end()), t.end());
	d.stats_table.swap(t);
	upload(d.d_stats_table, d.stats_table, s);
	CKL_HIP(hipStreamSynchronize(s));
}

// voxel_counts / centroids / bounding_boxes of the decoded z-range (operations.hpp:321-618):
// one pipeline run up to the run labels, then k_run_stats instead of the paint.
void decoder_label_stats(ckl_decoder& d, uint64_t capacity, uint64_t* labels, uint64_t* counts, uint64_t* sums, uint32_t* boxes, uint64_t* n_out) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	if (d.sxy == 0 || d.nslices == 0) { *n_out = 0; return; }
	ensure_label_table(d);
	const uint64_t nt = d.stats_table.size();
	*n_out = nt;
	if (capacity < nt) throw Error(CKL_ERR_ARG, "crackle_amd: label statistics need room for " + std::to_string(nt) + " labels");
	d.d_stats_acc.ensure(nt * 4);
	d.d_stats_box.ensure(nt * 6);
	CKL_HIP(hipMemsetAsync(d.d_stats_acc.p, 0, nt * 4 * sizeof(unsigned long long), s));
	hipLaunchKernelGGL(k_stats_init, dim3(static_cast<uint32_t>((nt + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d.d_stats_box.p, static_cast<uint32_t>(nt));
	StatsArgs sa;
	sa.table = d.d_stats_table.p; sa.n_table = static_cast<uint32_t>(nt);
	sa.acc = d.d_stats_acc.p; sa.box = d.d_stats_box.p;
	sa.sx = h.sx; sa.n_pixels = static_cast<uint32_t>(d.sxy); sa.z_start = static_cast<uint32_t>(d.z_start);
	{
		// per-component accumulators in LDS: all of the fullest slice if the workgroup's LDS allows
		int max_lds = 0;
		CKL_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, d.device));
		uint32_t fit = static_cast<uint32_t>(std::max(0, max_lds - 1024)) / kStatsBytesPerComp;
		if (const char* env = getenv("CKL_STATS_LDS_COMPS")) fit = std::min<uint32_t>(fit, static_cast<uint32_t>(std::max(0, atoi(env))));   // testing: forces the run-by-run merge
		sa.lds_comps = std::min<uint32_t>((d.max_comp + 1) & ~1u, fit & ~1u);
		CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_run_stats), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sa.lds_comps * kStatsBytesPerComp)));
	}
	decoder_run(d, nullptr, 0, 0, 0, &sa);
	std::vector<unsigned long long> acc(nt * 4);
	CKL_HIP(hipMemcpyAsync(acc.data(), d.d_stats_acc.p, acc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	if (boxes) CKL_HIP(hipMemcpyAsync(boxes, d.d_stats_box.p, nt * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	CKL_HIP(hipStreamSynchronize(s));
	for (uint64_t i = 0; i < nt; i++) {
		if (labels) labels[i] = d.stats_table[i];
		if (counts) counts[i] = acc[4 * i];
		if (sums) { sums[3 * i] = acc[4 * i + 1]; sums[3 * i + 1] = acc[4 * i + 2]; sums[3 * i + 2] = acc[4 * i + 3]; }
	}
	// The reference seeds its box map from the unique list only (operations.hpp:561-567); a label outside
	// that list — the background colour of a pin stream — is default-constructed by bbxes[label] when
	// its first component is merged (:596), so its minima start at 0; absent, it has no entry at all
	// (reported here as a zero box with count 0).
	if (boxes && d.bg_unlisted) {
		const uint64_t i = static_cast<uint64_t>(std::lower_bound(d.stats_table.begin(), d.stats_table.end(), d.bgcolor) - d.stats_table.begin());
		if (i < nt && d.stats_table[i] == d.bgcolor) {
			boxes[6 * i] = boxes[6 * i + 1] = boxes[6 * i + 2] = 0;
			if (acc[4 * i] == 0) boxes[6 * i + 3] = boxes[6 * i + 4] = boxes[6 * i + 5] = 0;
		}
	}
}

// voxel_connectivity_graph of the decoder's range into a device buffer of sx*sy*slices bytes
void decoder_vcg(ckl_decoder& d, uint8_t* out_device, uint64_t capacity, int connectivity) {
	const Header& h = d.head;
	if (connectivity != 4 && connectivity != 6) throw Error(CKL_ERR_ARG, "crackle: voxel_connectivity_graph: only connectivity 4 and 6 are currently supported.");
	if (d.sxy == 0 || d.nslices == 0) return;
	const uint64_t need = d.sxy * d.nslices;
	if (!out_device || capacity < need) throw Error(CKL_ERR_ARG, "crackle_amd: output buffer too small: need " + std::to_string(need) + " bytes");
	hipStream_t s = d.stream;
	const bool six = connectivity == 6 && d.nslices > 1;
	DevBuf<uint8_t> labels;
	if (six) {
		// the z bits compare decoded labels: the whole pipeline runs into a scratch volume first
		d.use_fused = false;      // (k_vcg reads the crack planes afterwards: the one-launch strip kernel leaves only their seam rows in HBM)
		labels.ensure(need * h.data_width);
		decoder_run(d, labels.p, need * h.data_width, 0, 0);
	}
	else decoder_run(d, nullptr, 0, 0, 0, nullptr, true);
	RunGeom g;
	g.planeV = d.d_planes.p; g.planeH = d.d_planes.p + d.plane_words * d.nslices;
	g.row_words = d.row_words; g.plane_words = d.plane_words;
	g.flip = (h.crack_format == IMPERMISSIBLE) ? 1u : 0u;
	g.sx = h.sx; g.sy = h.sy;
	const dim3 grid(static_cast<uint32_t>((d.sxy + kBlock - 1) / kBlock), d.nslices);
	const uint32_t f = h.fortran_order ? 1u : 0u;
	if (h.data_width == 1) hipLaunchKernelGGL(k_vcg<uint8_t>, grid, dim3(kBlock), 0, s, g, reinterpret_cast<const uint8_t*>(labels.p), six ? 1u : 0u, f, d.nslices, d.sxy, out_device);
	else if (h.data_width == 2) hipLaunchKernelGGL(k_vcg<uint16_t>, grid, dim3(kBlock), 0, s, g, reinterpret_cast<const uint16_t*>(labels.p), six ? 1u : 0u, f, d.nslices, d.sxy, out_device);
	else if (h.data_width == 4) hipLaunchKernelGGL(k_vcg<uint32_t>, grid, dim3(kBlock), 0, s, g, reinterpret_cast<const uint32_t*>(labels.p), six ? 1u : 0u, f, d.nslices, d.sxy, out_device);
	else hipLaunchKernelGGL(k_vcg<uint64_t>, grid, dim3(kBlock), 0, s, g, reinterpret_cast<const uint64_t*>(labels.p), six ? 1u : 0u, f, d.nslices, d.sxy, out_device);
	CKL_HIP(hipStreamSynchronize(s));
	CKL_HIP(hipGetLastError());
}

// operations::point_cloud (src/operations.hpp:183-262): contours of every component of the decoder's
// range (ckl_contours.hpp), grouped by label.  The pipeline runs up to the run tables and the
// component -> label map (the integrity check's mode), then per z-chunk: direction masks, the
// per-slice tracer, the contours' components; the host orders the contours of each component
// (dual_graph.hpp:223-241), groups components by label in (z, component) order and plans the
// output, which k_contour_emit writes on the device.
struct PointCloud {
	std::vector<uint64_t> labels;      // ascending
	std::vector<uint64_t> offsets;     // [labels + 1], in points
	uint16_t* points = nullptr;        // host_out_alloc, 3 uint16 per point
};

void decoder_point_cloud(ckl_decoder& d, const uint64_t* sel, uint64_t n_sel, bool has_sel, bool skip_background, PointCloud& out) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	out.offsets.assign(1, 0);
	if (d.sxy == 0 || ns == 0) return;
	if (d.sxy >= 0xFFFFFFF0ull / 4) throw Error(CKL_ERR_ARG, "crackle_amd: point_cloud: slices of this size are not supported");
	{
		std::vector<uint32_t> errs(ns);
		decoder_run(d, nullptr, 0, 0, 0, nullptr, false, errs.data());
		for (uint32_t zi = 0; zi < ns; zi++) {
			if (!errs[zi]) continue;
			const std::string z = std::to_string(d.z_start + zi);
			if (errs[zi] & (ERR_BOC | ERR_RANGE | ERR_CAPACITY)) throw Error(CKL_ERR_RUNTIME, "crackle: crack code is malformed or corrupted on z=" + z);
			if (errs[zi] & ERR_NCOMP) throw Error(CKL_ERR_RUNTIME, "crackle: component count does not match the label section on z=" + z);
			throw Error(CKL_ERR_CRC, "crackle: crack code crc mismatch on z=" + z);
		}
	}
	RunGeom g;
	g.planeV = d.d_planes.p; g.planeH = d.d_planes.p + d.plane_words * ns;
	g.row_words = d.row_words; g.plane_words = d.plane_words;
	g.flip = (h.crack_format == IMPERMISSIBLE) ? 1u : 0u;
	g.sx = h.sx; g.sy = h.sy;
	RunArrays ra;
	ra.word_base = d.d_word_base.p; ra.rbase = d.d_rbase.p; ra.rcap = d.d_rcap.p;
	ra.parent = d.d_parent.p; ra.run_start = d.d_run_start.p; ra.run_cc = d.d_run_cc.p;
	ra.nruns = d.d_nruns.p; ra.ncomp = d.d_ncomp.p; ra.slice_err = d.d_slice_err.p;

	// component -> label as an index into the stream's sorted label table (binary search on the
	// device); point_cloud<LABEL> keys its map by the unsigned type of the data width, which keeps
	// the table's order (sign-extended labels, ascending as unsigned)
	ensure_label_table(d);
	const uint32_t n_table = static_cast<uint32_t>(d.stats_table.size());
	std::vector<uint32_t> comp_key(d.total_comp);
	if (d.total_comp) {
		DevBuf<uint32_t> d_comp_key;
		d_comp_key.ensure(d.total_comp);
		hipLaunchKernelGGL(k_component_label_index, dim3(static_cast<uint32_t>((d.total_comp + 255) / 256)), dim3(256), 0, s,
			d.d_label_map.p, d.total_comp, d.d_stats_table.p, n_table, d_comp_key.p);
		CKL_HIP(hipMemcpyAsync(comp_key.data(), d_comp_key.p, d.total_comp * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
	}
	const uint64_t lmask = h.data_width >= 8 ? ~0ull : ((1ull << (8 * h.data_width)) - 1);
	std::vector<uint64_t> comp_off(ns + 1, 0);
	for (uint32_t zi = 0; zi < ns; zi++) comp_off[zi + 1] = comp_off[zi] + d.ncomp_expect_host[zi];
	std::vector<uint64_t> selected;
	if (has_sel) { selected.assign(sel, sel + n_sel); std::sort(selected.begin(), selected.end()); }
	auto takes = [&](uint64_t label) {
		if (skip_background && label == 0) return false;
		return !has_sel || std::binary_search(selected.begin(), selected.end(), label);
	};

	const bool prof = getenv("CKL_PROFILE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	std::string marks;
	auto mark = [&](const char* name) {
		if (!prof) return;
		CKL_HIP(hipStreamSynchronize(s));
		const auto now = std::chrono::steady_clock::now();
		char buf[64];
		snprintf(buf, sizeof buf, " %s=%.2f", name, std::chrono::duration<double, std::milli>(now - t_last).count());
		marks += buf;
		t_last = now;
	};
	mark("tables");
	uint64_t walk_steps = 0;
	const uint32_t sxy = static_cast<uint32_t>(d.sxy);
	const uint64_t dirs_stride = (d.sxy + 3) & ~3ull;
	const uint32_t vis_words = (sxy + 31) / 32;
	const uint32_t cand_words = ((vis_words + 1) & ~1u) + 2 * kContourWindow;      // a scan window may start at the last word
	int max_lds = 0;
	CKL_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, d.device));
	const bool lds_vis = !getenv("CKL_CONTOUR_HBM_VISITED") && static_cast<uint64_t>(vis_words) * 4 + 256 <= static_cast<uint64_t>(std::max(0, max_lds));
	if (lds_vis) CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_contours<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(vis_words * 4)));

	// contours of all slices: per slice the kept ones in discovery order
	struct Kept { uint32_t zi, offset, len, rot, first, comp; };
	std::vector<Kept> kept;
	// per z-chunk device buffers, kept until the emit; first try with room for the usual volume,
	// a chunk that overflows is traced again with the bounds of the worst case
	struct Chunk { uint32_t z0, n; uint32_t raw_cap; std::unique_ptr<DevBuf<uint32_t>> raw; };
	std::vector<Chunk> chunks;
	const uint64_t budget = 3ull << 30;
	uint32_t raw_cap0 = static_cast<uint32_t>(std::min<uint64_t>(d.sxy / 2 + 4096, 8ull * d.sxy + 16));
	uint32_t tab_cap0 = static_cast<uint32_t>(std::min<uint64_t>(d.sxy / 32 + 1024, d.sxy + 1));
	if (const char* env = getenv("CKL_CONTOUR_SMALL")) { raw_cap0 = std::max(16, atoi(env)); tab_cap0 = std::max(2, atoi(env) / 8); }      // testing: forces the second pass
	DevBuf<uint8_t> d_dirs;
	DevBuf<uint32_t> d_vis, d_counts, d_comp, d_cand, d_walked;
	DevBuf<uint4> d_table;
	uint32_t z0 = 0;
	while (z0 < ns) {
		uint32_t raw_cap = raw_cap0, tab_cap = tab_cap0;
		for (int attempt = 0; ; attempt++) {
			const uint64_t per_slice = dirs_stride + 4ull * vis_words + 8ull * cand_words + 4ull * raw_cap + 16ull * tab_cap + 4ull * tab_cap + (lds_vis ? 0 : 4ull * vis_words) + 16;
			const uint32_t nz = static_cast<uint32_t>(std::min<uint64_t>(ns - z0, std::max<uint64_t>(1, budget / per_slice)));
			RunGeom gz = g;
			gz.planeV = g.planeV + static_cast<uint64_t>(z0) * d.plane_words;
			gz.planeH = g.planeH + static_cast<uint64_t>(z0) * d.plane_words;
			RunArrays rz = ra;
			rz.word_base = ra.word_base + static_cast<uint64_t>(z0) * d.plane_words;
			rz.rbase = ra.rbase + z0;
			d_dirs.ensure(dirs_stride * nz);
			std::unique_ptr<DevBuf<uint32_t>> raw(new DevBuf<uint32_t>());
			raw->ensure(static_cast<uint64_t>(raw_cap) * nz);
			d_table.ensure(static_cast<uint64_t>(tab_cap) * nz);
			d_comp.ensure(static_cast<uint64_t>(tab_cap) * nz);
			d_counts.ensure(4ull * nz);
			if (!lds_vis) {
				d_vis.ensure(static_cast<uint64_t>(vis_words) * nz);
				CKL_HIP(hipMemsetAsync(d_vis.p, 0, static_cast<uint64_t>(vis_words) * nz * sizeof(uint32_t), s));
			}
			d_cand.ensure(2ull * cand_words * nz);
			CKL_HIP(hipMemsetAsync(d_cand.p, 0, 2ull * cand_words * nz * sizeof(uint32_t), s));
			hipLaunchKernelGGL(k_contour_dirs, dim3(static_cast<uint32_t>((d.sxy + 255) / 256), nz), dim3(256), 0, s, gz, d.sxy, dirs_stride, d_dirs.p,
				cand_words, d_cand.p, d_cand.p + static_cast<uint64_t>(cand_words) * nz);
			mark("dirs");
			const bool memo = !getenv("CKL_CONTOUR_NO_MEMO");      // testing: every start is walked, like the reference does
			if (memo) {
				d_walked.ensure(static_cast<uint64_t>(vis_words) * nz);
				CKL_HIP(hipMemsetAsync(d_walked.p, 0, static_cast<uint64_t>(vis_words) * nz * sizeof(uint32_t), s));
			}
			ContourArgs ca;
			ca.walked_r = memo ? d_walked.p : nullptr;
			ca.cand_a = d_cand.p; ca.cand_b = d_cand.p + static_cast<uint64_t>(cand_words) * nz; ca.cand_words = cand_words;
			ca.dirs = d_dirs.p; ca.visited = d_vis.p; ca.raw = raw->p; ca.table = d_table.p; ca.counts = d_counts.p;
			ca.sx = h.sx; ca.sy = h.sy; ca.sxy = sxy; ca.raw_cap = raw_cap; ca.tab_cap = tab_cap; ca.vis_words = vis_words; ca.dirs_stride = dirs_stride;
			if (lds_vis) hipLaunchKernelGGL(k_trace_contours<true>, dim3(nz), dim3(64), vis_words * 4, s, ca);
			else hipLaunchKernelGGL(k_trace_contours<false>, dim3(nz), dim3(64), 0, s, ca);
			mark("trace");
			hipLaunchKernelGGL(k_contour_components, dim3((tab_cap + 255) / 256, nz), dim3(256), 0, s, gz, rz, d_table.p, d_counts.p, tab_cap, d_comp.p);
			std::vector<uint32_t> counts(4ull * nz);
			CKL_HIP(hipMemcpyAsync(counts.data(), d_counts.p, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
			CKL_HIP(hipStreamSynchronize(s));
			CKL_HIP(hipGetLastError());
			uint32_t flags = 0;
			for (uint32_t i = 0; i < nz; i++) { flags |= counts[4 * i + 2]; walk_steps += counts[4 * i + 3]; }
			if (flags & kContourOpenWalk) throw Error(CKL_ERR_RUNTIME, "crackle_amd: point_cloud: a contour walk did not close");
			if (flags) {
				if (attempt) throw Error(CKL_ERR_RUNTIME, "crackle_amd: point_cloud: contour buffers overflow");
				raw_cap = static_cast<uint32_t>(8ull * d.sxy + 16);
				tab_cap = sxy + 1;
				continue;
			}
			// the kept contours of the chunk, packed on the device: one copy instead of one per slice
			std::vector<uint32_t> base(nz + 1, 0);
			for (uint32_t i = 0; i < nz; i++) base[i + 1] = base[i] + counts[4 * i];
			const uint32_t n_chunk = base[nz];
			if (n_chunk) {
				DevBuf<uint32_t> d_base, d_pcomp;
				DevBuf<uint4> d_ptable;
				upload(d_base, base, s);
				d_ptable.ensure(n_chunk); d_pcomp.ensure(n_chunk);
				hipLaunchKernelGGL(k_contour_pack, dim3((tab_cap + 255) / 256, nz), dim3(256), 0, s, d_table.p, d_comp.p, d_counts.p, d_base.p, tab_cap, d_ptable.p, d_pcomp.p);
				std::vector<uint4> table(n_chunk);
				std::vector<uint32_t> comp(n_chunk);
				CKL_HIP(hipMemcpyAsync(table.data(), d_ptable.p, n_chunk * sizeof(uint4), hipMemcpyDeviceToHost, s));
				CKL_HIP(hipMemcpyAsync(comp.data(), d_pcomp.p, n_chunk * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
				CKL_HIP(hipStreamSynchronize(s));
				const size_t at0 = kept.size();
				kept.resize(at0 + n_chunk);
				for (uint32_t i = 0; i < nz; i++) {
					for (uint32_t k = base[i]; k < base[i + 1]; k++) {
						const uint4 t = table[k];
						kept[at0 + k] = { z0 + i, t.x, t.y, t.z, t.w, comp[k] };
					}
				}
			}
			chunks.push_back({ z0, nz, raw_cap, std::move(raw) });
			z0 += nz;
			break;
		}
	}

	mark("collect");
	// merge_contours_via_vcg_coloring (dual_graph.hpp:213-243): a contour whose first node lies before
	// the component's current first node goes to the front, any other to the back
	std::vector<uint64_t> comp_n(d.total_comp, 0);                // contours per component
	for (const Kept& k : kept) {
		if (k.comp >= d.ncomp_expect_host[k.zi]) throw Error(CKL_ERR_RUNTIME, "crackle_amd: point_cloud: component out of range");
		comp_n[comp_off[k.zi] + k.comp]++;
	}
	std::vector<uint64_t> comp_at(d.total_comp + 1, 0);
	for (uint64_t c = 0; c < d.total_comp; c++) comp_at[c + 1] = comp_at[c] + comp_n[c];
	std::vector<uint32_t> order(kept.size());                    // per component: its contours in merged order
	{
		std::vector<std::deque<uint32_t>> lists;                    // only for components with more than one contour
		std::vector<int64_t> list_of(d.total_comp, -1);
		for (uint32_t i = 0; i < kept.size(); i++) {
			const uint64_t c = comp_off[kept[i].zi] + kept[i].comp;
			if (comp_n[c] == 1) { order[comp_at[c]] = i; continue; }
			if (list_of[c] < 0) { list_of[c] = static_cast<int64_t>(lists.size()); lists.emplace_back(); }
			std::deque<uint32_t>& l = lists[list_of[c]];
			if (!l.empty() && kept[l.front()].first > kept[i].first) l.push_front(i);
			else l.push_back(i);
		}
		for (uint64_t c = 0; c < d.total_comp; c++) {
			if (list_of[c] < 0) continue;
			uint64_t at = comp_at[c];
			for (uint32_t i : lists[list_of[c]]) order[at++] = i;
		}
	}

	// operations.hpp:229-257: components in (z, index) order append to their label's points.  The
	// labels of the output are the table's entries that own a component of the range and pass the
	// filters, in table order
	std::vector<uint8_t> present(n_table, 0);
	for (uint64_t c = 0; c < d.total_comp; c++) {
		if (comp_key[c] >= n_table) throw Error(CKL_ERR_RUNTIME, "crackle_amd: point_cloud: a component's label is not in the stream's label table");
		present[comp_key[c]] = 1;
	}
	std::vector<int64_t> table_out(n_table, -1);
	std::vector<uint64_t> keys;
	for (uint32_t i = 0; i < n_table; i++) {
		const uint64_t label = d.stats_table[i] & lmask;
		if (!present[i] || !takes(label)) continue;
		table_out[i] = static_cast<int64_t>(keys.size());
		keys.push_back(label);
	}
	std::vector<uint64_t> off(keys.size() + 1, 0);
	std::vector<int64_t> key_of(d.total_comp, -1);
	for (uint64_t c = 0; c < d.total_comp; c++) {
		const int64_t k = table_out[comp_key[c]];
		if (k < 0) continue;
		key_of[c] = k;
		for (uint64_t j = comp_at[c]; j < comp_at[c + 1]; j++) off[k + 1] += kept[order[j]].len;
	}
	for (size_t k = 0; k < keys.size(); k++) off[k + 1] += off[k];
	const uint64_t total_points = off[keys.size()];
	out.labels = keys;
	out.offsets = off;
	out.points = static_cast<uint16_t*>(host_out_alloc(std::max<uint64_t>(total_points * 6, 2)));
	if (!total_points) return;

	mark("order");
	DevBuf<uint16_t> d_points;
	d_points.ensure(total_points * 3);
	std::vector<uint64_t> fill(off.begin(), off.end() - 1);
	std::vector<std::vector<ContourJob>> jobs(chunks.size());
	{
		std::vector<uint32_t> chunk_of(ns);
		for (size_t ci = 0; ci < chunks.size(); ci++) for (uint32_t i = 0; i < chunks[ci].n; i++) chunk_of[chunks[ci].z0 + i] = static_cast<uint32_t>(ci);
		for (uint32_t zi = 0; zi < ns; zi++) {
			const Chunk& ch = chunks[chunk_of[zi]];
			for (uint64_t c = comp_off[zi]; c < comp_off[zi + 1]; c++) {
				if (key_of[c] < 0) continue;
				for (uint64_t j = comp_at[c]; j < comp_at[c + 1]; j++) {
					const Kept& k = kept[order[j]];
					ContourJob job;
					job.src = static_cast<uint64_t>(zi - ch.z0) * ch.raw_cap + k.offset;
					job.dst = fill[key_of[c]];
					job.len = k.len; job.rot = k.rot; job.z = static_cast<uint32_t>(d.z_start + zi); job.pad = 0;
					fill[key_of[c]] += k.len;
					jobs[chunk_of[zi]].push_back(job);
				}
			}
		}
	}
	mark("plan");
	DevBuf<ContourJob> d_jobs;
	for (size_t ci = 0; ci < chunks.size(); ci++) {
		if (jobs[ci].empty()) continue;
		upload(d_jobs, jobs[ci], s);
		const uint64_t nj = jobs[ci].size();
		hipLaunchKernelGGL(k_contour_emit, dim3(static_cast<uint32_t>((nj + 3) / 4)), dim3(256), 0, s, d_jobs.p, nj, chunks[ci].raw->p, h.sx, d_points.p);
		CKL_HIP(hipStreamSynchronize(s));      // the job list is reused by the next chunk
	}
	mark("emit");
	CKL_HIP(hipMemcpyAsync(out.points, d_points.p, total_points * 6, hipMemcpyDeviceToHost, s));
	CKL_HIP(hipStreamSynchronize(s));
	CKL_HIP(hipGetLastError());
	mark("d2h");
	if (prof) fprintf(stderr, "[ckl point_cloud ms]%s | contours=%zu points=%llu walk_steps=%llu\n", marks.c_str(), kept.size(), static_cast<unsigned long long>(total_points), static_cast<unsigned long long>(walk_steps));
}

}  // namespace

extern "C" {

// A session's stream and its two dozen events cost more to create than the rest of its set-up: sessions
// that end hand them to a small per-device cache (ckl_decoder_destroy), new ones take them from it.
struct SessionResources {
	int device = -1;
	hipStream_t stream = nullptr;
	hipEvent_t ev[kMaxStages + 2] = {};
	hipEvent_t ev_in = nullptr;
	int n_cus = 0, max_lds = 0;
};
static std::mutex g_session_mutex;
static std::vector<SessionResources>& session_cache() { static std::vector<SessionResources>& c = *new std::vector<SessionResources>(); return c; }      // never destroyed: no HIP calls at exit

// stream, events and kernel attributes of a new session; the caller builds it
static std::unique_ptr<ckl_decoder> decoder_new(int device) {
	select_device(device);
	std::unique_ptr<ckl_decoder> d(new ckl_decoder());
	d->device = device;
	int max_lds = 0;
	bool cached = false;
	{
		std::lock_guard<std::mutex> lock(g_session_mutex);
		auto& c = session_cache();
		for (size_t i = 0; i < c.size(); i++) {
			if (c[i].device != device) continue;
			d->stream = c[i].stream; d->ev_in = c[i].ev_in; d->n_cus = c[i].n_cus; max_lds = c[i].max_lds;
			for (int k = 0; k < kMaxStages + 2; k++) d->ev[k] = c[i].ev[k];
			c[i] = c.back(); c.pop_back();
			cached = true;
			break;
		}
	}
	if (!cached) {
		CKL_HIP(hipDeviceGetAttribute(&d->n_cus, hipDeviceAttributeMultiprocessorCount, device));
		CKL_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
		for (auto& e : d->ev) CKL_HIP(hipEventCreate(&e));
		CKL_HIP(hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming));
		CKL_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
	}
	d->max_lds = max_lds;
	// the kernels' dynamic LDS limits are set once per device and size
	static std::mutex mu;
	static std::vector<std::pair<int, size_t>>& done = *new std::vector<std::pair<int, size_t>>();
	auto once = [&](int key, size_t bytes) {
		std::lock_guard<std::mutex> lock(mu);
		const std::pair<int, size_t> k(device * 8 + key, bytes);
		if (std::find(done.begin(), done.end(), k) != done.end()) return false;
		done.push_back(k);
		return true;
	};
	{
		// LDS control tables of k_decode_cracks: as many symbols as the workgroup's LDS allows
		const size_t budget = static_cast<size_t>(max_lds > 4096 ? max_lds - 4096 : 0);   // static LDS of the kernel: ~2.4 KiB
		uint32_t nctl = 5120;
		if (const char* env = getenv("CKL_LDS_CONTROLS")) nctl = static_cast<uint32_t>(std::max(0, atoi(env))) & ~63u;   // testing: forces the global tables (multiples of 64: 16-byte aligned tables)
		while (nctl > 64 && crack_lds_bytes(nctl) > budget) nctl -= 64;
		if (crack_lds_bytes(nctl) > budget) nctl = 0;
		if (nctl > 32000) nctl = 32000;
		d->lds_controls = nctl;
		// one workgroup per CU either way: the raster phase takes whatever LDS is left for
		// its bands (1024x1024 slices: 2 passes instead of 3)
		d->lds_bytes = getenv("CKL_LDS_CONTROLS") ? crack_lds_bytes(nctl) : std::max(crack_lds_bytes(nctl), budget & ~static_cast<size_t>(15));
		const int bytes = static_cast<int>(d->lds_bytes);
		if (once(0, d->lds_bytes)) {
			CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_cracks<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
			if constexpr (kTuning) CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_cracks<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
		}
	}
	{
		// k_crack_match: two workgroups per CU, each with half of the CU's LDS for its control tables
		const size_t budget = static_cast<size_t>(max_lds / 2 > 4096 ? max_lds / 2 - 4096 : 0);      // static LDS of the kernel: ~1.5 KiB
		uint32_t nctl = 8192;
		if (const char* env = getenv("CKL_LDS_CONTROLS")) nctl = static_cast<uint32_t>(std::max(0, atoi(env))) & ~63u;   // testing: forces the global tables
		while (nctl > 64 && rec_lds_bytes(nctl) > budget) nctl -= 64;
		if (rec_lds_bytes(nctl) > budget) nctl = 0;
		d->rec_lds_controls = nctl;
		d->rec_lds = (rec_lds_bytes(nctl) + 15) & ~static_cast<size_t>(15);
		// markov streams are expanded in the same LDS: give them all of the half
		if (!getenv("CKL_LDS_CONTROLS")) d->rec_lds = std::max(d->rec_lds, budget & ~static_cast<size_t>(15));
		if (nctl && once(1, d->rec_lds)) CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_crack_match<kRecBlock>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(d->rec_lds)));
	}
	return d;
}

int ckl_decoder_create(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int device, ckl_decoder** out) {
	try {
		if (!buf || !out) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		// header problems are format errors even when no device is present
		if (n < Header::kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
		(void)Header::parse(buf, n);
		std::unique_ptr<ckl_decoder> d = decoder_new(device);
		decoder_build(*d, buf, n, z_start, z_end);
		*out = d.release();
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

// The stream is resident in HBM already (ckl_encoder_device_stream, or a caller that keeps streams on
// the device): nothing is uploaded.  What the host has to see — header, z-index, the head of the label
// section (the whole section of a pin stream), markov model, crc tail — comes back in three small
// copies, each sized by the one before; the crack codes are never touched by the host.
// The per-slice component counts of a FLAT label section (labels.hpp:424-451) lie behind its unique labels, i.e.
// at an offset the host only knows once it has read the header and the section's first eight bytes: a second round
// trip of ckl_decoder_create_device.  This kernel reads those fields on the device (the host validates its own copy
// of them afterwards) and sends the counts along with the first round.  grid = ceil(max_len / 4096), block = kBlock
__global__ void __launch_bounds__(kBlock) k_fetch_flat_counts(const uint8_t* __restrict__ src, uint64_t n, uint8_t* __restrict__ dst_host, uint64_t max_len) {
	__shared__ uint64_t s_off, s_len;
	if (threadIdx.x == 0) {
		uint64_t off = 0, len = 0;
		auto rd = [&](uint64_t at, int w) -> uint64_t { uint64_t v = 0; for (int i = 0; i < w; i++) v |= static_cast<uint64_t>(src[at + i]) << (8 * i); return v; };
		if (n >= Header::kBytesV0 && src[0] == 'c' && src[1] == 'r' && src[2] == 'k' && src[3] == 'l' && src[4] <= 1 && (src[4] == 0 || n >= Header::kBytes)) {
			const uint32_t ver = src[4];
			const uint32_t fmt = static_cast<uint32_t>(rd(5, 2));
			const uint64_t sx = rd(7, 4), sy = rd(11, 4), sz = rd(15, 4);
			const uint64_t nlb = ver == 0 ? rd(20, 4) : rd(20, 8);
			const uint64_t sw = 1ull << ((fmt >> 2) & 3u);
			const uint64_t hb = ver == 0 ? Header::kBytesV0 : Header::kBytes, gib = (sz + (ver ? 1u : 0u)) * 4u;
			if (((fmt >> 5) & 3u) == FLAT && nlb >= 8 && hb + gib + 8 <= n && sx * sy) {
				const uint64_t nu = rd(hb + gib, 8);
				const uint64_t px = sx * sy, cw = px <= 0xFFull ? 1 : px <= 0xFFFFull ? 2 : px <= 0xFFFFFFFFull ? 4 : 8;
				if (nu <= (nlb - 8) / sw) {
					const uint64_t o = 8 + sw * nu;
					off = hb + gib + o;
					len = cw * sz < nlb - o ? cw * sz : nlb - o;
				}
			}
		}
		if (off >= n) len = 0;
		if (len > n - off) len = n - off;
		if (len > max_len) len = max_len;
		s_off = off; s_len = len;
	}
	__syncthreads();
	const uint64_t off = s_off, len = s_len;
	const uint64_t i = (static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x) * 16u;
	if (i >= len) return;
	for (uint64_t b = i; b < len && b < i + 16u; b++) dst_host[off + b] = src[off + b];
}

int ckl_decoder_create_device(const uint8_t* stream_device, uint64_t n, int64_t z_start, int64_t z_end, int device, ckl_decoder** out) {
	try {
		if (!stream_device || !out) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (n < Header::kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
		const bool prof = getenv("CKL_PROFILE") != nullptr;
		auto c0 = std::chrono::steady_clock::now();
		std::vector<std::pair<const char*, double>> marks;
		auto mark = [&](const char* name) { if (prof) { auto c1 = std::chrono::steady_clock::now(); marks.emplace_back(name, std::chrono::duration<double, std::milli>(c1 - c0).count()); c0 = c1; } };
		std::unique_ptr<ckl_decoder> d = decoder_new(device);
		mark("session");
		hipStream_t s = d->stream;
		wait_for_default_stream(s, d->ev_in);      // the caller may just have written the stream on the default stream
		// sparse host image of the stream: only the ranges fetched below are ever read (decoder_build)
		struct Image {
			uint8_t* p = nullptr; uint64_t n = 0;
			~Image() { if (p) host_out_free(p); }
		} img;
		img.n = n;
		img.p = static_cast<uint8_t*>(host_out_alloc(n));
		const bool mapped = host_out_is_pinned(img.p);      // host_out_alloc pins (and maps) blocks of 64 KiB and more
		auto fetch = [&](uint64_t off, uint64_t len) {
			if (off >= n || len == 0) return;
			len = std::min<uint64_t>(len, n - off);
			if (mapped && len <= (1u << 20)) hipLaunchKernelGGL(k_fetch_to_host, dim3(static_cast<uint32_t>((len + 16u * kBlock - 1) / (16u * kBlock)), 1), dim3(kBlock), 0, s, stream_device, img.p, off, len, 0ull, 0ull);
			else CKL_HIP(hipMemcpyAsync(img.p + off, stream_device + off, len, hipMemcpyDeviceToHost, s));
		};
		mark("image");
		if (prof) { CKL_HIP(hipStreamSynchronize(s)); mark("presync"); }
		// first round, sized for the common case: the front of the stream (header, z-index and label section
		// head of up to ~16 K slices) and its end (the crc tail); what a stream needs beyond that follows
		const uint64_t kFront = 64u << 10;
		const uint64_t front = std::min<uint64_t>(n, kFront), back0 = n > 2 * kFront ? n - kFront : front;
		if (mapped) {      // both ranges in one launch
			const uint64_t longest = std::max(front, n - back0);
			hipLaunchKernelGGL(k_fetch_to_host, dim3(static_cast<uint32_t>((longest + 16u * kBlock - 1) / (16u * kBlock)), 2), dim3(kBlock), 0, s, stream_device, img.p, 0ull, front, back0, n - back0);
		}
		else { fetch(0, front); fetch(back0, n - back0); }
		const uint64_t kCountsAhead = 256u << 10;      // bytes of component counts the first round brings along (k_fetch_flat_counts)
		if (mapped) hipLaunchKernelGGL(k_fetch_flat_counts, dim3(static_cast<uint32_t>(kCountsAhead / (16u * kBlock))), dim3(kBlock), 0, s, stream_device, n, img.p, kCountsAhead);
		CKL_HIP(hipStreamSynchronize(s));
		mark("header");
		const Header h = Header::parse(img.p, n);
		if (!h.layout_fits(n)) throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_code_offsets: Unable to read past end of buffer.");
		const uint64_t hb = h.header_bytes(), gib = h.grid_index_bytes();
		const uint64_t tail = h.format_version == 0 ? 0 : 4ull * (static_cast<uint64_t>(h.sz) + 1);
		const int sw = h.stored_data_width;
		const bool flat = h.label_format == FLAT;
		// z-index, label section head (all of a pin section), model, crc tail: whatever the first round missed
		auto need = [&](uint64_t off, uint64_t len) {
			if (len == 0 || off >= n) return false;
			len = std::min<uint64_t>(len, n - off);
			if (off + len <= front || off >= back0) return false;
			fetch(off, len);
			return true;
		};
		bool more = need(hb, gib + (flat ? 16 : h.num_label_bytes));
		more = need(hb + gib + h.num_label_bytes, h.markov_model_bytes()) || more;
		more = need(n - tail, tail) || more;
		if (more) CKL_HIP(hipStreamSynchronize(s));
		if (flat && h.num_label_bytes >= 8 && static_cast<uint64_t>(h.sx) * h.sy) {
			// component counts: behind the unique labels (labels.hpp:424-451)
			const uint64_t nu = rd_le(img.p + hb + gib, 8);
			const uint64_t cw = static_cast<uint64_t>(byte_width(static_cast<uint64_t>(h.sx) * h.sy));
			if (nu <= (h.num_label_bytes - 8) / static_cast<uint64_t>(sw)) {
				const uint64_t off = 8 + static_cast<uint64_t>(sw) * nu;
				const uint64_t len = std::min<uint64_t>(cw * h.sz, h.num_label_bytes - std::min(off, h.num_label_bytes));
				if (!(mapped && len <= kCountsAhead)) {      // (else k_fetch_flat_counts sent exactly this range with the first round)
					fetch(hb + gib + off, len);
					CKL_HIP(hipStreamSynchronize(s));
				}
			}
		}
		mark("tables");
		decoder_build(*d, img.p, n, z_start, z_end, stream_device);
		mark("build");
		if (prof) {
			fprintf(stderr, "[ckl decoder_create_device host ms]");
			for (auto& m : marks) fprintf(stderr, " %s=%.3f", m.first, m.second);
			fprintf(stderr, "\n");
		}
		*out = d.release();
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_run(ckl_decoder* d, void* out_device, uint64_t out_capacity_bytes, int has_label, uint64_t label) {
	try {
		if (!d) throw Error(CKL_ERR_ARG, "crackle_amd: null decoder");
		select_device(d->device);
		wait_for_default_stream(d->stream, d->ev_in);
		decoder_run(*d, out_device, out_capacity_bytes, has_label, label);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_label_stats(ckl_decoder* d, uint64_t capacity, uint64_t* labels, uint64_t* counts, uint64_t* sums, uint32_t* boxes, uint64_t* n_labels) {
	try {
		if (!d || !n_labels) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(d->device);
		wait_for_default_stream(d->stream, d->ev_in);
		decoder_label_stats(*d, capacity, labels, counts, sums, boxes, n_labels);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_crack_planes(ckl_decoder* d, const uint32_t** plane_v, const uint32_t** plane_h, uint32_t* row_words, uint64_t* plane_words) {
	try {
		if (!d || !plane_v || !plane_h || !row_words || !plane_words) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(d->device);
		wait_for_default_stream(d->stream, d->ev_in);
		*plane_v = nullptr; *plane_h = nullptr; *row_words = d->row_words; *plane_words = d->plane_words;
		if (d->sxy == 0 || d->nslices == 0) return CKL_OK;
		decoder_run(*d, nullptr, 0, 0, 0, nullptr, true);
		*plane_v = d->d_planes.p;
		*plane_h = d->d_planes.p + d->plane_words * d->nslices;
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_check(ckl_decoder* d, uint32_t* slice_errors, uint64_t capacity) {
	try {
		if (!d || !slice_errors) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (capacity < d->nslices) throw Error(CKL_ERR_ARG, "crackle_amd: room for " + std::to_string(d->nslices) + " slice verdicts needed");
		select_device(d->device);
		wait_for_default_stream(d->stream, d->ev_in);
		if (d->sxy == 0 || d->nslices == 0) return CKL_OK;
		decoder_run(*d, nullptr, 0, 0, 0, nullptr, false, slice_errors);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_vcg(ckl_decoder* d, uint8_t* out_device, uint64_t out_capacity_bytes, int connectivity) {
	try {
		if (!d) throw Error(CKL_ERR_ARG, "crackle_amd: null decoder");
		select_device(d->device);
		wait_for_default_stream(d->stream, d->ev_in);
		decoder_vcg(*d, out_device, out_capacity_bytes, connectivity);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_voxel_connectivity_graph(const uint8_t* buf, uint64_t n, int connectivity, int device, uint8_t* out_host, uint64_t out_capacity_bytes) {
	return ckl_voxel_connectivity_graph_range(buf, n, 0, -1, connectivity, device, out_host, out_capacity_bytes);
}

int ckl_voxel_connectivity_graph_range(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int connectivity, int device, uint8_t* out_host, uint64_t out_capacity_bytes) {
	ckl_decoder* d = nullptr;
	int rc = ckl_decoder_create(buf, n, z_start, z_end, device, &d);
	if (rc != CKL_OK) return rc;
	try {
		const uint64_t need = d->sxy * d->nslices;
		if (need) {
			if (!out_host || out_capacity_bytes < need) throw Error(CKL_ERR_ARG, "crackle_amd: output buffer too small: need " + std::to_string(need) + " bytes");
			DevBuf<uint8_t> tmp;
			tmp.ensure(need);
			decoder_vcg(*d, tmp.p, need, connectivity);
			CKL_HIP(hipMemcpy(out_host, tmp.p, need, hipMemcpyDeviceToHost));
		}
		else if (connectivity != 4 && connectivity != 6) throw Error(CKL_ERR_ARG, "crackle: voxel_connectivity_graph: only connectivity 4 and 6 are currently supported.");
		ckl_decoder_destroy(d);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); ckl_decoder_destroy(d); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); ckl_decoder_destroy(d); return CKL_ERR_RUNTIME; }
}

int ckl_point_cloud(
	const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end,
	const uint64_t* labels, uint64_t n_labels, int has_labels, int skip_background, int device,
	uint64_t** labels_out, uint64_t** offsets_out, uint16_t** points_out, uint64_t* n_out
) {
	ckl_decoder* d = nullptr;
	uint64_t* lo = nullptr; uint64_t* oo = nullptr;
	PointCloud pc;
	try {
		if (!buf || !labels_out || !offsets_out || !points_out || !n_out || (has_labels && n_labels && !labels)) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		*labels_out = nullptr; *offsets_out = nullptr; *points_out = nullptr; *n_out = 0;
		Header h = Header::parse(buf, n);
		// operations::get_szr (src/operations.hpp:54-72): the clamped range must hold a slice
		int64_t zs = z_start, ze = z_end;
		const int64_t last = static_cast<int64_t>(static_cast<uint32_t>(h.sz - 1u));
		zs = std::max<int64_t>(std::min(zs, last), 0);
		ze = ze < 0 ? static_cast<int64_t>(h.sz) : ze;
		ze = std::max<int64_t>(std::min<int64_t>(ze, h.sz), 0);
		if (zs >= ze) throw Error(CKL_ERR_RUNTIME, "crackle: Invalid range: " + std::to_string(zs) + " - " + std::to_string(ze));
		const int rc = ckl_decoder_create(buf, n, zs, ze, device, &d);
		if (rc != CKL_OK) return rc;
		decoder_point_cloud(*d, labels, n_labels, has_labels != 0, skip_background != 0, pc);
		const uint64_t k = pc.labels.size();
		lo = static_cast<uint64_t*>(host_out_alloc(std::max<uint64_t>(k, 1) * 8));
		oo = static_cast<uint64_t*>(host_out_alloc((k + 1) * 8));
		if (k) memcpy(lo, pc.labels.data(), k * 8);
		memcpy(oo, pc.offsets.data(), (k + 1) * 8);
		if (!pc.points) pc.points = static_cast<uint16_t*>(host_out_alloc(2));
		*labels_out = lo; *offsets_out = oo; *points_out = pc.points; *n_out = k;
		ckl_decoder_destroy(d);
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); if (pc.points) host_out_free(pc.points); if (lo) host_out_free(lo); if (oo) host_out_free(oo); if (d) ckl_decoder_destroy(d); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); if (pc.points) host_out_free(pc.points); if (lo) host_out_free(lo); if (oo) host_out_free(oo); if (d) ckl_decoder_destroy(d); return CKL_ERR_RUNTIME; }
}

int ckl_decoder_last_timing(const ckl_decoder* d, float* pipeline_ms, float* dominant_kernel_ms) {
	if (!d) { set_last_error("crackle_amd: null decoder"); return CKL_ERR_ARG; }
	float mx = 0.f;
	for (int i = 0; i < d->n_stages; i++) mx = d->stage_ms[i] > mx ? d->stage_ms[i] : mx;
	if (pipeline_ms) *pipeline_ms = d->pipeline_ms;
	if (dominant_kernel_ms) *dominant_kernel_ms = mx;
	return CKL_OK;
}

int ckl_decoder_stage_timing(const ckl_decoder* d, int index, const char** name, float* ms) {
	if (!d || index < 0 || index >= d->n_stages) { set_last_error("crackle_amd: stage index out of range"); return CKL_ERR_ARG; }
	if (name) *name = d->stage_name[index];
	if (ms) *ms = d->stage_ms[index];
	return CKL_OK;
}

int ckl_decoder_set_stage_events(ckl_decoder* d, int on) {
	if (!d) { set_last_error("crackle_amd: null decoder"); return CKL_ERR_ARG; }
	d->stage_events = on != 0;
	return CKL_OK;
}

void ckl_decoder_destroy(ckl_decoder* d) {
	if (!d) return;
	// a session that was built and never run still has its descriptor upload in flight: the staging block and the
	// stream must not go back to their caches under it
	if (d->pending_upload && d->stream) (void)hipStreamSynchronize(d->stream);
	// stream and events go to the cache (at most four sets are kept), idle: every run waits for its own work
	if (d->stream) {
		std::lock_guard<std::mutex> lock(g_session_mutex);
		auto& c = session_cache();
		if (c.size() < 4) {
			SessionResources r;
			r.device = d->device; r.stream = d->stream; r.ev_in = d->ev_in; r.n_cus = d->n_cus; r.max_lds = d->max_lds;
			for (int k = 0; k < kMaxStages + 2; k++) { r.ev[k] = d->ev[k]; d->ev[k] = nullptr; }
			d->stream = nullptr; d->ev_in = nullptr;
			c.push_back(r);
		}
	}
	delete d;
}

int ckl_array_equal(const uint8_t* buf1, uint64_t n1, const uint8_t* buf2, uint64_t n2, int device, int* equal) {
	ckl_decoder *d1 = nullptr, *d2 = nullptr;
	try {
		if (!buf1 || !buf2 || !equal) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		*equal = 0;
		const Header h1 = Header::parse(buf1, n1), h2 = Header::parse(buf2, n2);
		// operations.hpp:1049-1062; get_voxels -> get_szr (:54-87) throws for a stream without slices
		if (h1.sz == 0 || h2.sz == 0) throw Error(CKL_ERR_RUNTIME, "crackle: Invalid range: 0 - 0");
		if (h1.voxels() == 0 || h2.voxels() == 0) { *equal = h1.voxels() == h2.voxels(); return CKL_OK; }
		if (h1.sx != h2.sx || h1.sy != h2.sy || h1.sz != h2.sz) return CKL_OK;
		int rc = ckl_decoder_create(buf1, n1, 0, -1, device, &d1);
		if (rc == CKL_OK) rc = ckl_decoder_create(buf2, n2, 0, -1, device, &d2);
		if (rc != CKL_OK) { ckl_decoder_destroy(d1); ckl_decoder_destroy(d2); return rc; }
		// the reference compares the component counts its CCL finds (:1146-1149); those of a valid
		// stream are the counts its label section states
		if (d1->ncomp_expect_host != d2->ncomp_expect_host) { ckl_decoder_destroy(d1); ckl_decoder_destroy(d2); return CKL_OK; }
		// label_map1[ccl1] against label_map1[ccl2] (:1160-1171; yes, label_map1 on both sides): the
		// first stream is decoded as it is, the second one's components are painted through the FIRST
		// stream's component -> label table.  Both x fastest, whatever the headers say.
		d1->use_general = true; d2->use_general = true;      // the general pipeline keeps the component -> label table
		d1->head.fortran_order = true; d2->head.fortran_order = true;
		const uint64_t bytes = d1->sxy * d1->nslices * static_cast<uint64_t>(h1.data_width);
		DevBuf<uint8_t> a, b;
		DevBuf<uint32_t> flag;
		a.ensure(bytes); b.ensure(bytes); flag.ensure(1);
		decoder_run(*d1, a.p, bytes, 0, 0);
		d2->foreign_label_map = d1->d_label_map.p;
		d2->paint_width = h1.data_width;
		CKL_HIP(hipStreamSynchronize(d1->stream));
		decoder_run(*d2, b.p, bytes, 0, 0);
		hipStream_t s = d2->stream;
		CKL_HIP(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
		hipLaunchKernelGGL(k_buffers_differ, dim3(2048), dim3(kBlock), 0, s, a.p, b.p, bytes, flag.p);
		uint32_t differ = 0;
		CKL_HIP(hipMemcpyAsync(&differ, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
		CKL_HIP(hipGetLastError());
		*equal = differ ? 0 : 1;
		ckl_decoder_destroy(d1); ckl_decoder_destroy(d2);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); ckl_decoder_destroy(d1); ckl_decoder_destroy(d2); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); ckl_decoder_destroy(d1); ckl_decoder_destroy(d2); return CKL_ERR_RUNTIME; }
}

int ckl_mode_pooling_2x2x1(
	const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int device,
	uint8_t** out, uint64_t* out_len, uint64_t** lengths, uint64_t* count
) {
	ckl_decoder* d = nullptr;
	ckl_encoder* e = nullptr;
	uint64_t* lens = nullptr;
	try {
		if (!buf || !out || !out_len || !lengths || !count) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		*out = nullptr; *out_len = 0; *lengths = nullptr; *count = 0;
		const Header h = Header::parse(buf, n);
		{
			// get_szr (operations.hpp:54-72): an empty range is an error, an empty slice is not (:1213-1215)
			int64_t zs = std::max<int64_t>(std::min<int64_t>(z_start, static_cast<int64_t>(static_cast<uint32_t>(h.sz - 1u))), 0);
			int64_t ze = z_end < 0 ? static_cast<int64_t>(h.sz) : z_end;
			ze = std::max<int64_t>(std::min<int64_t>(ze, static_cast<int64_t>(h.sz)), 0);
			if (zs >= ze) throw Error(CKL_ERR_RUNTIME, "crackle: Invalid range: " + std::to_string(zs) + " - " + std::to_string(ze));
		}
		if (static_cast<uint64_t>(h.sx) * h.sy == 0) return CKL_OK;
		int rc = ckl_decoder_create(buf, n, z_start < 0 ? 0 : z_start, z_end, device, &d);
		if (rc != CKL_OK) return rc;
		d->head.fortran_order = true;      // the pooling walks x fastest (operations.hpp:1241-1249)
		d->use_general = d->use_general || !(h.fortran_order);      // a C-order stream was laid out for the general pipeline
		const uint32_t sx = h.sx, sy = h.sy, nz = d->nslices;
		const uint32_t osx = (sx + 1u) >> 1, osy = (sy + 1u) >> 1;
		const int w = h.data_width;
		const uint64_t vox = d->sxy * nz, ovox = static_cast<uint64_t>(osx) * osy * nz;
		DevBuf<uint8_t> full, pooled;
		full.ensure(vox * w); pooled.ensure(ovox * w);
		decoder_run(*d, full.p, vox * w, 0, 0);
		hipStream_t s = d->stream;
		const dim3 grid(static_cast<uint32_t>((ovox + kBlock - 1) / kBlock));
		if (w == 1) hipLaunchKernelGGL(k_mode_pool_2x2<uint8_t>, grid, dim3(kBlock), 0, s, full.p, pooled.p, sx, sy, nz);
		else if (w == 2) hipLaunchKernelGGL(k_mode_pool_2x2<uint16_t>, grid, dim3(kBlock), 0, s, reinterpret_cast<const uint16_t*>(full.p), reinterpret_cast<uint16_t*>(pooled.p), sx, sy, nz);
		else if (w == 4) hipLaunchKernelGGL(k_mode_pool_2x2<uint32_t>, grid, dim3(kBlock), 0, s, reinterpret_cast<const uint32_t*>(full.p), reinterpret_cast<uint32_t*>(pooled.p), sx, sy, nz);
		else hipLaunchKernelGGL(k_mode_pool_2x2<uint64_t>, grid, dim3(kBlock), 0, s, reinterpret_cast<const uint64_t*>(full.p), reinterpret_cast<uint64_t*>(pooled.p), sx, sy, nz);
		CKL_HIP(hipStreamSynchronize(s));
		CKL_HIP(hipGetLastError());
		// every pooled slice becomes a stream of its own: crackle::compress<LABEL>(oimg, osx, osy, 1) with
		// its defaults (operations.hpp:1294-1297; LABEL is the unsigned type of the data width)
		rc = ckl_encoder_create(osx, osy, 1, w, device, &e);
		if (rc != CKL_OK) { ckl_decoder_destroy(d); return rc; }
		std::vector<uint8_t> all;
		lens = static_cast<uint64_t*>(host_out_alloc(sizeof(uint64_t) * (nz ? nz : 1)));
		for (uint32_t z = 0; z < nz; z++) {
			uint8_t* one = nullptr; uint64_t len = 0;
			rc = ckl_encoder_run(e, pooled.p + static_cast<uint64_t>(z) * osx * osy * w, osx, osy, 1, 0, 1, 0, 0, 1, 0, nullptr, &one, &len);
			if (rc != CKL_OK) { host_out_free(lens); ckl_encoder_destroy(e); ckl_decoder_destroy(d); return rc; }
			all.insert(all.end(), one, one + len);
			lens[z] = len;
			ckl_free(one);
		}
		ckl_encoder_destroy(e); e = nullptr;
		ckl_decoder_destroy(d); d = nullptr;
		uint8_t* o = static_cast<uint8_t*>(host_out_alloc(all.size() ? all.size() : 1));
		memcpy(o, all.data(), all.size());
		*out = o; *out_len = all.size(); *lengths = lens; *count = nz;
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); if (lens) host_out_free(lens); if (e) ckl_encoder_destroy(e); ckl_decoder_destroy(d); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); if (lens) host_out_free(lens); if (e) ckl_encoder_destroy(e); ckl_decoder_destroy(d); return CKL_ERR_RUNTIME; }
}


int ckl_decompress(
	const uint8_t* buf, uint64_t n, void* out, uint64_t out_capacity_bytes, int out_mem,
	int64_t z_start, int64_t z_end, int has_label, uint64_t label, int device
) {
	ckl_decoder* d = nullptr;
	int rc = ckl_decoder_create(buf, n, z_start, z_end, device, &d);
	if (rc != CKL_OK) return rc;
	try {
		const uint64_t need = d->sxy * d->nslices * static_cast<uint64_t>(has_label ? 1 : d->head.data_width);
		if (need == 0) { ckl_decoder_destroy(d); return CKL_OK; }
		if (out_capacity_bytes < need || !out) throw Error(CKL_ERR_ARG, "crackle_amd: output buffer too small: need " + std::to_string(need) + " bytes");
		if (out_mem == CKL_MEM_DEVICE) {
			decoder_run(*d, out, out_capacity_bytes, has_label, label);
		}
		else {
			DevBuf<uint8_t> tmp;
			tmp.ensure(need);
			decoder_run(*d, tmp.p, need, has_label, label);
			CKL_HIP(hipMemcpy(out, tmp.p, need, hipMemcpyDeviceToHost));
		}
		ckl_decoder_destroy(d);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); ckl_decoder_destroy(d); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); ckl_decoder_destroy(d); return CKL_ERR_RUNTIME; }
}

}  // extern "C"
