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
// Kernels (block = 256 threads):
//   k_decode_cracks   one workgroup per slice: BOC index, 2-bit unpack + mod-4 prefix
//                     sum (undo the difference code), control-pair detection, symbol
//                     compaction and displacement prefix sums (block scans), branch
//                     matching of the control symbols 64 at a time by one wavefront
//                     (ballot / shuffle pointer jumping, stack in LDS), then parallel
//                     rasterisation of every move into two bit planes.
//   k_run_index       horizontal runs of each slice from the vertical-crack plane:
//                     exclusive prefix sum of break counts per 32-pixel word.
//   k_run_union       union-find over RUNS (not pixels): vertically adjacent runs that
//                     are connected through the horizontal-crack plane are united.
//   k_run_resolve     roots ranked in raster order = the reference's component ids
//                     (cc3d.hpp:114-144); crc32c of the (never materialised) component
//                     image accumulated per run from a geometric-sum table.
//   k_label_map_*     component -> label tables (flat keys / pins)
//   k_run_labels      run -> label
//   k_paint_runs      streams the output: per 4 pixels one plane word, a popcount and a
//                     look-up in an LDS-staged run->label table; 16-byte stores.
#include "ckl_common.hpp"
#include "ckl_runs.hpp"

#include <algorithm>
#include <chrono>
#include <memory>

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
	const uint64_t* cbase;       // [nslices] base index into the per-code scratch arrays
	const uint32_t* ccap;        // [nslices] capacity (codes) of this slice's scratch
	const uint64_t* nbase;       // [nslices] base index into `nodes`
	const uint32_t* ncap;        // [nslices]
	int sx, sy;
	int xw, yw;
	int markov_order;
	const uint8_t* model;        // [4^order][4] rank -> symbol
	uint8_t* ucode;              // unpacked difference codes (markov only)
	uint8_t* ctl_kind;           // control symbols ('b'/'t') in stream order
	uint32_t* ctl_pos;           // displacement prefix sum (mod 2^32) before each control symbol
	uint32_t* ctl_seg;           // number of 't' symbols before each control symbol
	uint32_t* seg_off;           // per segment: vertex offset to add to the displacement
	uint32_t* stack;
	uint32_t* nodes;
	uint32_t* planeV;
	uint32_t* planeH;
	uint32_t row_words;
	uint64_t plane_words;
	uint32_t* slice_err;         // [nslices] sticky error bits
};

__device__ __forceinline__ uint32_t rd_le_dev(const uint8_t* p, int w) {
	uint32_t v = 0;
	for (int i = 0; i < w; i++) v |= static_cast<uint32_t>(p[i]) << (8 * i);
	return v;
}

constexpr int kCrackBlock = 1024;                 // threads per slice
constexpr int kCrackWaves = kCrackBlock / kWave;

// DIAG builds stamp the phase boundaries (diagnostic only): diag[zi*8 + {A, B, C, D}] cycles
template <bool DIAG>
__global__ void __launch_bounds__(kCrackBlock) k_decode_cracks(CrackArgs a, unsigned long long* __restrict__ diag) {
	__shared__ uint32_t s_scan[4 * kCrackWaves];
	__shared__ int32_t s_scanmax[kCrackWaves];
	__shared__ uint32_t s_last_move[kCrackBlock];
	__shared__ uint32_t s_last_ctrl[kCrackBlock];
	constexpr uint32_t kStackLds = 2048;
	__shared__ uint32_t s_stack[kStackLds];
	__shared__ uint32_t s_nnodes, s_ncodes, s_nctl, s_valid_segs, s_err;

	unsigned long long d_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
	auto stamp = [&](int slot) {
		if (DIAG && threadIdx.x == 0 && diag) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			diag[static_cast<uint64_t>(blockIdx.x) * 8 + slot] = now - d_t;
			d_t = now;
		}
	};

	const uint32_t zi = blockIdx.x;
	const int tid = threadIdx.x;
	const uint8_t* code = a.stream + a.code_off[zi];
	const uint32_t code_len = a.code_len[zi];
	const uint64_t cb = a.cbase[zi];
	const uint32_t cap = a.ccap[zi];
	uint32_t* nodes = a.nodes + a.nbase[zi];
	const uint32_t ncap = a.ncap[zi];
	const uint32_t sxe = a.sx + 1, sye = a.sy + 1;
	const uint32_t nverts = sxe * sye;
	const uint32_t sx = a.sx, sy = a.sy;

	// ---- phase A: beginning-of-chain index (crackcodes.hpp:283-316), serial ----
	if (tid == 0) {
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
			// ---- markov bitstream -> unpacked difference codes (markov.hpp:268-313), serial ----
			const uint32_t nbytes = code_len - index_end;
			if (a.markov_order == 0) {
				ncodes = nbytes * 4u;
			}
			else if (nbytes > 0) {
				const uint8_t* s = code + index_end;
				uint8_t* uc = a.ucode + cb;
				const int shift = 2 * (a.markov_order - 1);
				uint32_t m = 0;
				const uint32_t start = s[0] & 3u;
				uc[m++] = static_cast<uint8_t>(start);
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
						if (m < cap) uc[m++] = static_cast<uint8_t>(v);
						else err |= ERR_CAPACITY;
						ctx = (ctx >> 2) + (v << shift);
					}
					pos -= 8;
				}
				ncodes = m;
			}
		}
		if (ncodes > cap) { ncodes = cap; err |= ERR_CAPACITY; }
		s_nnodes = nn;
		s_ncodes = ncodes;
		s_err = err;
		s_valid_segs = 0;
		s_nctl = 0;
	}
	__syncthreads();
	stamp(0);
	const uint32_t n_codes = s_ncodes;
	const uint32_t n_nodes = s_nnodes;
	const uint32_t index_end = 4u + (code_len >= 4u ? rd_le_dev(code, 4) : 0u);
	const uint8_t* packed = code + index_end;          // only dereferenced when n_codes > 0 (then index_end <= code_len)
	const uint8_t* ucode = a.ucode + cb;

	uint8_t* ctl_kind = a.ctl_kind + cb;
	uint32_t* ctl_pos = a.ctl_pos + cb;
	uint32_t* ctl_seg = a.ctl_seg + cb;
	uint32_t* seg_off = a.seg_off + cb;
	uint32_t* stack = a.stack + cb;
	uint32_t* pv = a.planeV + zi * a.plane_words;
	uint32_t* ph = a.planeH + zi * a.plane_words;
	uint32_t rerr = 0;

	// The symbol stream is derived twice from the packed codes (~25 KiB per slice) instead
	// of being stored: pass 0 records only the control symbols ('b'/'t', ~3 % of the
	// stream) for the branch matcher (phase C); pass 1 recomputes every symbol and
	// rasterises the moves straight from registers (phase D) once the offset of every
	// segment is known.
	for (int pass = 0; pass < 2 && n_nodes > 0 && n_codes > 0; pass++) {
		const uint32_t valid_segs = s_valid_segs;   // pass 1: set by phase C

		// ---- phase B: codes -> symbols, tiled block scans with carries -------------------
		// (crackcodes.hpp:547-598 / SURVEY.md Appendix D6)
		// Every code position g in [0, n_codes] is visited; position g finalises the
		// symbol of code g-1 (a move becomes 'b'/'t' when code g is its exact reverse).
		uint32_t carry_sum = 0;          // running mod-4 sum of difference codes
		uint32_t carry_move = 0xFF;      // move of code g0-1 (0xFF: none)
		uint32_t carry_ctrl = 0;         // was code g0-1 the second half of a control pair
		int32_t carry_lf = -1;           // last position whose `reverse-of-previous` test was false
		uint32_t carry_nt = 0, carry_nctl = 0, carry_pos = 0;
		constexpr uint32_t kPer = 16;
		constexpr uint32_t kTile = kCrackBlock * kPer;

		for (uint32_t tile = 0; tile <= n_codes; tile += kTile) {
			const uint32_t g0 = tile + tid * kPer;
			// -- load 16 difference codes
			uint32_t dc[kPer];
			uint32_t tsum = 0;
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) {
				const uint32_t g = g0 + k;
				uint32_t c = 0;
				if (g < n_codes) {
					c = a.markov_order ? ucode[g] : ((packed[g >> 2] >> (2 * (g & 3))) & 3u);
				}
				tsum = (tsum + c) & 3u;
				dc[k] = tsum;   // inclusive local sum mod 4
			}
			uint32_t v1[1] = { tsum }, t1[1];
			block_excl_add<1, kCrackWaves>(v1, t1, s_scan);
			const uint32_t base_sum = (carry_sum + v1[0]) & 3u;
			uint32_t mv[kPer];
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) mv[k] = (dc[k] + base_sum) & 3u;
			s_last_move[tid] = mv[kPer - 1];
			__syncthreads();
			const uint32_t prev_move = tid ? s_last_move[tid - 1] : carry_move;
			const uint32_t tile_last_move = s_last_move[kCrackBlock - 1];

			// -- r[g]: code g is the exact reverse of code g-1; runs of r alternate ctrl/move
			uint32_t rmask = 0;
			int32_t lf = INT32_MIN;
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) {
				const uint32_t g = g0 + k;
				const uint32_t pm = k ? mv[k - 1] : prev_move;
				const bool r = (g > 0) && (g < n_codes) && (pm != 0xFF) && ((mv[k] ^ pm) == 2u);
				if (r) rmask |= (1u << k);
				else lf = static_cast<int32_t>(g);
			}
			int32_t lf_tot;
			int32_t lf_in = block_excl_max<kCrackWaves>(lf, lf_tot, s_scanmax);
			if (lf_in < carry_lf) lf_in = carry_lf;
			uint32_t cmask = 0;   // ctrl flags of my 16 codes
			{
				int32_t cur = lf_in;
#pragma unroll
				for (uint32_t k = 0; k < kPer; k++) {
					const int32_t g = static_cast<int32_t>(g0 + k);
					if (rmask & (1u << k)) { if ((g - cur) & 1) cmask |= (1u << k); }
					else cur = g;
				}
			}
			s_last_ctrl[tid] = (cmask >> (kPer - 1)) & 1u;
			__syncthreads();
			const uint32_t prev_ctrl = tid ? s_last_ctrl[tid - 1] : carry_ctrl;
			const uint32_t tile_last_ctrl = s_last_ctrl[kCrackBlock - 1];

			// -- events: position g emits the symbol of code g-1 unless g-1 was a control half
			uint32_t emask = 0;
			uint32_t n_t = 0, n_ctl = 0, dpos = 0;
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) {
				const uint32_t g = g0 + k;
				const uint32_t pm = k ? mv[k - 1] : prev_move;
				const uint32_t pc = k ? ((cmask >> (k - 1)) & 1u) : prev_ctrl;
				if (g >= 1 && g <= n_codes && !pc && pm != 0xFF) {
					uint32_t kind;
					if (g < n_codes && (cmask & (1u << k))) kind = (mv[k] == 0u || mv[k] == 3u) ? SYM_T : SYM_B;
					else kind = pm;
					emask |= (1u << k);
					if (kind == SYM_T) n_t++;
					if (kind >= SYM_B) n_ctl++;
					else dpos += (kind == SYM_R) ? 1u : (kind == SYM_L) ? 0xFFFFFFFFu : (kind == SYM_D) ? sxe : (0u - sxe);
				}
			}
			uint32_t v3[3] = { n_t, n_ctl, dpos }, t3[3];
			block_excl_add<3, kCrackWaves>(v3, t3, s_scan);
			uint32_t o_t = carry_nt + v3[0], o_ctl = carry_nctl + v3[1], o_pos = carry_pos + v3[2];

			uint32_t* cur_word = nullptr;     // pass 1: bits of consecutive moves in one plane word
			uint32_t cur_bits = 0;
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) {
				if (!(emask & (1u << k))) continue;
				const uint32_t g = g0 + k;
				const uint32_t pm = k ? mv[k - 1] : prev_move;
				uint32_t kind;
				if (g < n_codes && (cmask & (1u << k))) kind = (mv[k] == 0u || mv[k] == 3u) ? SYM_T : SYM_B;
				else kind = pm;
				if (pass == 0) {
					if (kind >= SYM_B && o_ctl < cap) {
						ctl_kind[o_ctl] = static_cast<uint8_t>(kind);
						ctl_pos[o_ctl] = o_pos;
						ctl_seg[o_ctl] = o_t;
					}
				}
				else if (kind < SYM_B && o_t < valid_segs) {
					// ---- phase D: rasterise the move (crackcodes.hpp:706-862).  Vertical moves
					// cross planeV, horizontal moves cross planeH; consecutive moves of a straight
					// horizontal stretch share a plane word and are OR-ed into memory once.
					const uint32_t t = seg_off[o_t] + o_pos;
					if (t >= nverts) rerr |= ERR_RANGE;
					else {
						const uint32_t y = t / sxe;
						const uint32_t x = t - y * sxe;
						uint32_t* word = nullptr;
						uint32_t bx = 0;
						if (kind == SYM_D) {          // edge (x,y)-(x,y+1): between pixels (x-1,y) | (x,y)
							if (x >= 1 && x < sx && y < sy) { word = pv + static_cast<uint64_t>(y) * a.row_words + (x >> 5); bx = x; }
							else if (y >= sy) rerr |= ERR_RANGE;
						}
						else if (kind == SYM_U) {     // edge (x,y-1)-(x,y)
							if (x >= 1 && x < sx && y >= 1) { word = pv + static_cast<uint64_t>(y - 1) * a.row_words + (x >> 5); bx = x; }
							else if (y < 1) rerr |= ERR_RANGE;
						}
						else if (kind == SYM_R) {     // edge (x,y)-(x+1,y): between pixels (x,y-1) | (x,y)
							if (y >= 1 && y < sy && x < sx) { word = ph + static_cast<uint64_t>(y) * a.row_words + (x >> 5); bx = x; }
							else if (x >= sx) rerr |= ERR_RANGE;
						}
						else {                        // SYM_L: edge (x-1,y)-(x,y)
							if (y >= 1 && y < sy && x >= 1) { word = ph + static_cast<uint64_t>(y) * a.row_words + ((x - 1) >> 5); bx = x - 1; }
							else if (x < 1) rerr |= ERR_RANGE;
						}
						if (word) {
							if (word != cur_word) {
								if (cur_word) atomicOr(cur_word, cur_bits);
								cur_word = word;
								cur_bits = 0;
							}
							cur_bits |= 1u << (bx & 31);
						}
					}
				}
				if (kind == SYM_T) o_t++;
				if (kind >= SYM_B) o_ctl++;
				else o_pos += (kind == SYM_R) ? 1u : (kind == SYM_L) ? 0xFFFFFFFFu : (kind == SYM_D) ? sxe : (0u - sxe);
			}
			if (cur_word) atomicOr(cur_word, cur_bits);

			// -- carries to the next tile (uniform across the block)
			carry_sum = (carry_sum + t1[0]) & 3u;
			carry_move = (tile + kTile <= n_codes) ? tile_last_move : 0xFF;
			carry_ctrl = tile_last_ctrl;
			if (lf_tot > carry_lf) carry_lf = lf_tot;
			carry_nt += t3[0]; carry_nctl += t3[1]; carry_pos += t3[2];
			__syncthreads();
		}
		if (pass == 1) break;
		if (tid == 0) {
			s_nctl = carry_nctl < cap ? carry_nctl : cap;
			if (carry_nctl > cap) s_err |= ERR_CAPACITY;
		}
		__syncthreads();
		stamp(1);
		const uint32_t n_ctl = s_nctl;

		// ---- phase C: branch matching over the control symbols, 64 at a time by wave 0 ----
		// (crackcodes.hpp:771-781, 849-859: the rasteriser's revisit stack; chain
		// segmentation by branches_taken, crackcodes.hpp:549-598).
		// A 't' returns the cursor to where its matching 'b' was pushed.  Inside a chunk a
		// 't' matches the nearest earlier control of the same nesting level when that is a
		// 'b'; otherwise it pops the stack carried between chunks (or ends the chain when
		// the stack is empty).  Segment offsets chain through earlier 't's of the chunk and
		// are resolved by pointer jumping with shuffles.
		if (tid < kWave) {
			const int lane = tid;
			uint32_t valid = 1;
			uint32_t off = nodes[0];          // offset of the current segment (wave uniform)
			uint32_t chain = 0, sp = 0;
			bool done = false;
			if (lane == 0) seg_off[0] = off;
			const unsigned long long below = (1ull << lane) - 1ull;
			constexpr int PTR_NONE = -1, PTR_CARRY = -2;
			for (uint32_t base = 0; base < n_ctl && !done; base += kWave) {
				const uint32_t k = base + lane;
				const bool live = k < n_ctl;
				uint32_t pos = 0, seg = 0, kind = SYM_U;
				if (live) { kind = ctl_kind[k]; pos = ctl_pos[k]; seg = ctl_seg[k]; }
				const bool isT = live && kind == SYM_T;
				const bool isB = live && kind == SYM_B;
				const unsigned long long tmask = __ballot(isT), bmask = __ballot(isB);
				const uint32_t d = wave_incl_add(isB ? 1u : (isT ? 0xFFFFFFFFu : 0u));
				const uint32_t level = isT ? d + 1u : d;     // 'b': depth after, 't': depth before
				int match = -1;           // 't': lane of the matching 'b' in this chunk
				bool matched_b = false;   // 'b': closed inside this chunk
				for (unsigned long long todo = tmask | bmask; todo;) {
					const int lead = __ffsll(static_cast<long long>(todo)) - 1;
					const uint32_t lv = __shfl(level, lead, kWave);
					const unsigned long long same = __ballot((isT || isB) && level == lv);
					if ((isT || isB) && level == lv) {
						if (isT) {
							const unsigned long long prev = same & below;
							if (prev) {
								const int j = 63 - __clzll(static_cast<long long>(prev));
								if ((bmask >> j) & 1ull) match = j;
							}
						}
						else {
							const unsigned long long next = same & ~below & ~(1ull << lane);
							matched_b = next != 0;   // levels alternate b,t,b,t: the next one is its 't'
						}
					}
					todo &= ~same;
				}
				const bool um_t = isT && match < 0;
				const unsigned long long um_t_mask = __ballot(um_t);
				const uint32_t q = __popcll(um_t_mask & below);      // rank among the chunk's unmatched 't'
				const bool chain_end = um_t && q >= sp;
				const uint32_t new_chain = chain + (q - sp) + 1u;     // meaningful when chain_end
				const bool dead = chain_end && new_chain >= n_nodes;   // trailing pad codes start here
				const unsigned long long dead_mask = __ballot(dead);
				const int first_dead = dead_mask ? (__ffsll(static_cast<long long>(dead_mask)) - 1) : 64;
				const bool ok = lane < first_dead;                    // lanes before the pad codes

				// stack pops are read before anything is pushed back
				uint32_t popped = 0;
				if (um_t && !chain_end) {
					const uint32_t idx = sp - 1u - q;
					popped = idx < kStackLds ? s_stack[idx] : __hip_atomic_load(stack + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				const uint32_t pos_match = __shfl(pos, match >= 0 ? match : 0, kWave);
				uint32_t val = 0;
				int ptr = PTR_NONE;
				if (isT && ok) {
					if (match >= 0) {
						const unsigned long long pt = tmask & ((1ull << match) - 1ull);
						val = pos_match - pos;
						ptr = pt ? (63 - __clzll(static_cast<long long>(pt))) : PTR_CARRY;
					}
					else if (!chain_end) val = popped - pos;
					else val = nodes[new_chain] - pos;
				}
#pragma unroll
				for (int r = 0; r < 6; r++) {
					const int pl = ptr >= 0 ? ptr : lane;
					const uint32_t pvv = __shfl(val, pl, kWave);
					const int pp = __shfl(ptr, pl, kWave);
					if (ptr >= 0) { val += pvv; ptr = pp; }
				}
				const uint32_t my_off = val + (ptr == PTR_CARRY ? off : 0u);   // 't' lanes: offset of the segment they open

				// position of every 'b' = offset of the segment it sits in + its displacement
				const unsigned long long pt_b = tmask & below;
				const int src = pt_b ? (63 - __clzll(static_cast<long long>(pt_b))) : -1;
				const uint32_t src_off = __shfl(my_off, src >= 0 ? src : 0, kWave);
				const uint32_t t_b = (src >= 0 ? src_off : off) + pos;

				const unsigned long long ok_um_t = um_t_mask & (first_dead >= 64 ? ~0ull : ((1ull << first_dead) - 1ull));
				const uint32_t n_um_t = __popcll(ok_um_t);
				const uint32_t n_pop = n_um_t < sp ? n_um_t : sp;
				const uint32_t sp_base = sp - n_pop;
				const bool um_b = isB && ok && !matched_b;
				const unsigned long long um_b_mask = __ballot(um_b);
				if (um_b) {
					const uint32_t idx = sp_base + __popcll(um_b_mask & below);
					if (idx < kStackLds) s_stack[idx] = t_b; else stack[idx] = t_b;   // idx < #controls <= cap
				}
				if (isT && ok && seg + 1u < cap) seg_off[seg + 1u] = my_off;

				// carries
				const unsigned long long ok_t = tmask & (first_dead >= 64 ? ~0ull : ((1ull << first_dead) - 1ull));
				const int last_t = ok_t ? (63 - __clzll(static_cast<long long>(ok_t))) : -1;
				const uint32_t last_off = __shfl(my_off, last_t >= 0 ? last_t : 0, kWave);
				const uint32_t last_seg = __shfl(seg, last_t >= 0 ? last_t : 0, kWave);
				if (last_t >= 0) { off = last_off; valid = last_seg + 2u; }
				chain += n_um_t - n_pop;
				sp = sp_base + __popcll(um_b_mask);
				if (dead_mask) {
					valid = __shfl(seg, first_dead, kWave) + 1u;
					done = true;
				}
				__threadfence_block();
			}
			if (tid == 0) s_valid_segs = valid;
		}
		__syncthreads();
		__threadfence_block();
		stamp(2);
	}

	if (rerr) atomicOr(&s_err, rerr);
	__syncthreads();
	stamp(3);
	if (tid == 0 && s_err) atomicOr(a.slice_err + zi, s_err);
	if (DIAG && tid == 0 && diag) { diag[static_cast<uint64_t>(zi) * 8 + 4] = n_codes; diag[static_cast<uint64_t>(zi) * 8 + 5] = s_nctl; }
}

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
// paint (crackle.hpp:617-656): out[p] = label of p's run
// ------------------------------------------------------------------------------
constexpr uint32_t kPaintTile = 4096;          // pixels per workgroup
constexpr uint32_t kPaintStage = 3072;         // run labels staged in LDS per workgroup

template <typename OUT> struct Vec4;
template <> struct Vec4<uint8_t> { typedef uchar4 type; };
template <> struct Vec4<uint16_t> { typedef ushort4 type; };
template <> struct Vec4<uint32_t> { typedef uint4 type; };
template <> struct Vec4<uint64_t> { typedef ulonglong4 type; };

// FAST: sx % 4 == 0 and x-fastest output: a thread paints 4 consecutive pixels of one
// plane word per step and stores them as one vector.  grid = (ceil(sxy / 4096), nslices)
template <typename OUT, bool FAST>
__global__ void __launch_bounds__(kBlock) k_paint_runs(
	RunGeom g, RunArrays r, const OUT* __restrict__ run_label, OUT* __restrict__ out,
	uint64_t sxy, uint32_t nslices, uint32_t fortran_order
) {
	__shared__ OUT s_lab[kPaintStage];
	__shared__ uint32_t s_lo, s_hi;
	const uint32_t zi = blockIdx.y;
	const uint64_t p_lo = static_cast<uint64_t>(blockIdx.x) * kPaintTile;
	const uint64_t p_hi = (p_lo + kPaintTile < sxy ? p_lo + kPaintTile : sxy) - 1;   // last pixel of the tile
	const uint32_t* wb = r.word_base + zi * g.plane_words;
	const OUT* lab = run_label + r.rbase[zi];
	const uint32_t nruns = r.nruns[zi];
	auto run_of = [&](uint64_t p) {
		const uint32_t y = static_cast<uint32_t>(p / g.sx);
		const uint32_t x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
		return wb[y * g.row_words + (x >> 5)] + __popc(g.breaks(zi, y, x >> 5) & mask_le(x & 31u)) - 1u;
	};
	if (threadIdx.x == 0) s_lo = run_of(p_lo);
	if (threadIdx.x == 1) s_hi = run_of(p_hi);
	__syncthreads();
	const uint32_t lo = s_lo, hi = s_hi;
	const bool staged = (hi - lo) < kPaintStage && hi < nruns;
	if (staged) {
		for (uint32_t i = threadIdx.x; i <= hi - lo; i += kBlock) s_lab[i] = lab[lo + i];
	}
	__syncthreads();
	auto label_of = [&](uint32_t run) -> OUT {
		if (staged) return s_lab[run - lo];
		return run < nruns ? lab[run] : static_cast<OUT>(0);
	};
	if (FAST) {
		typedef typename Vec4<OUT>::type V4;
		OUT* oz = out + static_cast<uint64_t>(zi) * sxy;
#pragma unroll
		for (uint32_t it = 0; it < kPaintTile / (kBlock * 4); it++) {
			const uint64_t p = p_lo + it * (kBlock * 4) + threadIdx.x * 4u;
			if (p >= sxy) break;
			const uint32_t y = static_cast<uint32_t>(p / g.sx);
			const uint32_t x = static_cast<uint32_t>(p - static_cast<uint64_t>(y) * g.sx);
			const uint32_t bw = g.breaks(zi, y, x >> 5);
			const uint32_t sh = x & 31u;
			uint32_t run = wb[y * g.row_words + (x >> 5)] + __popc(bw & mask_le(sh)) - 1u;
			const uint32_t nib = (bw >> sh) >> 1;      // break flags of pixels x+1 .. x+3
			V4 v;
			v.x = label_of(run);
			run += nib & 1u;        v.y = label_of(run);
			run += (nib >> 1) & 1u; v.z = label_of(run);
			run += (nib >> 2) & 1u; v.w = label_of(run);
			*reinterpret_cast<V4*>(oz + p) = v;
		}
	}
	else {
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

}  // namespace ckl

// ------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------
using namespace ckl;

namespace {
constexpr int kMaxStages = 12;
}

struct ckl_decoder {
	int device = 0;
	hipStream_t stream = nullptr;
	// stage boundaries: ev[i] .. ev[i+1] brackets stage i of the last run
	hipEvent_t ev[kMaxStages + 1] = {};
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
	DevBuf<uint8_t> d_stream;
	DevBuf<uint64_t> d_code_off, d_cbase, d_nbase, d_comp_off, d_rbase;
	DevBuf<uint32_t> d_code_len, d_ccap, d_ncap, d_rcap;
	DevBuf<uint8_t> d_model, d_ucode, d_ctl_kind;
	DevBuf<uint32_t> d_ctl_pos, d_ctl_seg, d_seg_off, d_stack, d_nodes;
	DevBuf<uint32_t> d_planes;          // V then H
	DevBuf<uint32_t> d_word_base, d_parent, d_run_start, d_run_cc, d_nruns, d_ncomp, d_ncomp_expect, d_blk_roots;
	DevBuf<uint16_t> d_run_local;
	DevBuf<uint64_t> d_run_label;       // typed on use (1..8 bytes per run)
	DevBuf<uint32_t> d_G, d_crc_acc, d_crc_expect, d_slice_err;
	DevBuf<uint64_t> d_label_map;
	DevBuf<uint64_t> d_pin_index, d_pin_depth, d_pin_label, d_pin_work_off, d_ccl_id, d_ccl_label;

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
	uint32_t idbits = 1, crc_fix = 0;
	bool check_crc = true;

	~ckl_decoder() {
		for (auto& e : ev) if (e) (void)hipEventDestroy(e);
		if (stream) (void)hipStreamDestroy(stream);
	}
};

namespace {

template <typename T>
void upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t s) {
	d.ensure(h.size());
	if (!h.empty()) CKL_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

uint64_t read_stored(const Header& h, const uint8_t* lb, uint64_t offset) {
	const int w = h.stored_data_width;
	uint64_t v = rd_le(lb + offset, w);
	if (h.is_signed && w < 8 && (v >> (8 * w - 1))) v |= ~0ull << (8 * w);
	return v;
}

void decoder_build(ckl_decoder& d, const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end) {
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
	if (hb + gib + h.num_label_bytes + h.markov_model_bytes() + tail > n) {
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
	if (z_index[h.sz] + tail > n) throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_codes: Unable to read past end of buffer.");

	hipStream_t s = d.stream;
	// the whole stream goes to HBM once
	d.d_stream.ensure(n + 16);
	CKL_HIP(hipMemcpyAsync(d.d_stream.p, buf, n, hipMemcpyHostToDevice, s));

	// per-slice descriptors and scratch layout
	const int xw = byte_width(static_cast<uint64_t>(h.sx) + 1);
	const bool permissible = h.crack_format == PERMISSIBLE;
	std::vector<uint64_t> code_off(d.nslices), cbase(d.nslices), nbase(d.nslices), rbase(d.nslices);
	std::vector<uint32_t> code_len(d.nslices), ccap(d.nslices), ncap(d.nslices), rcap(d.nslices);
	uint64_t ctot = 0, ntot = 0, rtot = 0;
	d.max_rcap = 0;
	for (uint32_t zi = 0; zi < d.nslices; zi++) {
		const uint64_t z = static_cast<uint64_t>(zs) + zi;
		const uint64_t len = z_index[z + 1] - z_index[z];
		if (len > 0xFFFFFFF0ull / 8) throw Error(CKL_ERR_RUNTIME, "crackle_amd: crack code of a slice is too large");
		code_off[zi] = z_index[z];
		code_len[zi] = static_cast<uint32_t>(len);
		const uint64_t index_size = len >= 4 ? rd_le(buf + z_index[z], 4) : 0;
		const uint64_t payload = (len >= 4 + index_size) ? len - 4 - index_size : 0;
		// codes: 4 per byte (plain) or at most 8 per byte (+1 raw) for the markov bitstream
		const uint64_t cap = (h.markov_model_order ? payload * 8 + 1 : payload * 4) + 2;
		const uint64_t nodes_cap = std::min<uint64_t>(index_size, len) / xw + 1;
		// runs: one per row plus one per vertical crack move (IMPERMISSIBLE), else up to one per pixel
		const uint64_t runs_cap = permissible ? d.sxy : std::min<uint64_t>(d.sxy, static_cast<uint64_t>(h.sy) + cap);
		cbase[zi] = ctot; ccap[zi] = static_cast<uint32_t>(cap); ctot += cap;
		nbase[zi] = ntot; ncap[zi] = static_cast<uint32_t>(nodes_cap); ntot += nodes_cap;
		rbase[zi] = rtot; rcap[zi] = static_cast<uint32_t>(runs_cap); rtot += runs_cap;
		d.max_rcap = std::max<uint32_t>(d.max_rcap, static_cast<uint32_t>(runs_cap));
	}
	upload(d.d_code_off, code_off, s);
	upload(d.d_code_len, code_len, s);
	upload(d.d_cbase, cbase, s);
	upload(d.d_ccap, ccap, s);
	upload(d.d_nbase, nbase, s);
	upload(d.d_ncap, ncap, s);
	upload(d.d_rbase, rbase, s);
	upload(d.d_rcap, rcap, s);
	if (h.markov_model_order) d.d_ucode.ensure(ctot);
	d.d_ctl_kind.ensure(ctot);
	d.d_ctl_pos.ensure(ctot);
	d.d_ctl_seg.ensure(ctot);
	d.d_seg_off.ensure(ctot);
	d.d_stack.ensure(ctot);
	d.d_nodes.ensure(ntot);
	d.d_parent.ensure(rtot);
	d.d_run_start.ensure(rtot);
	d.d_run_cc.ensure(rtot);
	d.d_run_local.ensure(rtot);
	d.d_blk_roots.ensure(static_cast<size_t>((d.max_rcap + kBlock - 1) / kBlock) * d.nslices);
	d.d_run_label.ensure(rtot);

	if (h.markov_model_order) {
		std::vector<uint8_t> model = markov_model_from_stored(buf + hb + gib + h.num_label_bytes, h.markov_model_bytes(), h.markov_model_order);
		upload(d.d_model, model, s);
	}

	d.row_words = (h.sx + 31) / 32;
	d.plane_words = static_cast<uint64_t>(d.row_words) * h.sy;
	d.d_planes.ensure(2 * d.plane_words * d.nslices);
	d.d_word_base.ensure(d.plane_words * d.nslices);
	d.d_nruns.ensure(d.nslices);
	d.d_ncomp.ensure(d.nslices);
	d.d_slice_err.ensure(d.nslices);
	d.d_crc_acc.ensure(d.nslices);

	// label section layout (labels.hpp:424-451, 453-617), parsed once
	const uint8_t* lb = buf + hb + gib;
	const uint64_t nlb = h.num_label_bytes;
	const int sw = h.stored_data_width;
	const int component_width = byte_width(d.sxy);
	uint64_t offset;
	if (h.label_format == FLAT) {
		if (nlb < 8) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
		d.num_unique = rd_le(lb, 8);
		d.uniq_offset = 8;
		if (d.num_unique > nlb / sw) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
		offset = 8 + static_cast<uint64_t>(sw) * d.num_unique;
	}
	else if (h.label_format == PINS_VARIABLE_WIDTH) {
		if (nlb < static_cast<uint64_t>(sw) + 8) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		d.bgcolor = read_stored(h, lb, 0);
		d.num_unique = rd_le(lb + sw, 8);
		d.uniq_offset = static_cast<uint64_t>(sw) + 8;
		if (d.num_unique > nlb / sw) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		offset = 8 + static_cast<uint64_t>(sw) * (d.num_unique + 1);
	}
	else {
		throw Error(CKL_ERR_RUNTIME, "crackle: Unsupported label format. Got: " + std::to_string(h.label_format));
	}
	if (offset + static_cast<uint64_t>(component_width) * h.sz > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
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
	upload(d.d_comp_off, comp_off, s);
	upload(d.d_ncomp_expect, ncomp_expect, s);
	d.d_label_map.ensure(d.total_comp + 1);

	// crc machinery: geometric-sum table G[m] = x^32 + ... + x^(32 m), component ids are
	// multiplied in over their `idbits` significant bits only (see k_run_resolve)
	d.check_crc = h.format_version > 0;
	d.idbits = 1;
	while (d.idbits < 32 && (1ull << d.idbits) < max_comp) d.idbits++;
	d.crc_fix = gf_xpow(32 - d.idbits);
	{
		const uint32_t npx = static_cast<uint32_t>(d.sxy);
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
		d.d_G.ensure(static_cast<size_t>(npx) + 1);
		hipLaunchKernelGGL(k_build_geom_table, dim3(npx / kBlock + 1), dim3(kBlock), 0, s, t_base.p, t_g.p, t_x.p, npx, d.d_G.p);
		CKL_HIP(hipStreamSynchronize(s));
	}
	if (d.check_crc) {
		// stored = ~(x^(32 n) * 0xFFFFFFFF ^ raw)  =>  raw = ~stored ^ init_term
		const uint32_t init_term = gf_mul(0xFFFFFFFFu, gf_xpow(32ull * d.sxy));
		std::vector<uint32_t> expect(d.nslices);
		const uint8_t* crcs = buf + n - 4ull * h.sz;
		for (uint32_t zi = 0; zi < d.nslices; zi++) {
			const uint32_t stored = static_cast<uint32_t>(rd_le(crcs + 4ull * (zs + zi), 4));
			expect[zi] = (~stored) ^ init_term;
		}
		upload(d.d_crc_expect, expect, s);
	}
	else {
		d.d_crc_expect.ensure(d.nslices);
	}

	if (h.label_format == FLAT) {
		d.key_width = byte_width(d.num_unique);
		d.keys_offset = hb + gib + offset;
		if (offset + comp_prefix[h.sz] * static_cast<uint64_t>(d.key_width) > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: label section is malformed or corrupted.");
	}
	else {
		if (offset + 1 > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
		const uint8_t combined = lb[offset++];
		const int npw = 1 << (combined & 3), dw = 1 << ((combined >> 2) & 3), ccw = 1 << ((combined >> 4) & 3);
		const int iw = h.pin_index_width();
		std::vector<uint64_t> pin_index, pin_depth, pin_label, pin_work_off, ccl_id, ccl_label;
		const uint64_t comp_right = comp_prefix[ze];
		uint64_t i = offset, work = 0;
		for (uint64_t label = 0; label < d.num_unique; label++) {
			if (i + npw > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
			const uint64_t lv = read_stored(h, lb, d.uniq_offset + label * sw);
			const uint64_t num_pins = rd_le(lb + i, npw); i += npw;
			if (num_pins > nlb || i + num_pins * static_cast<uint64_t>(iw + dw) + npw > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
			uint64_t idx = 0;
			for (uint64_t j = 0; j < num_pins; j++) {
				idx += rd_le(lb + i + j * iw, iw);
				const uint64_t depth = rd_le(lb + i + num_pins * iw + j * dw, dw);
				const int64_t pin_z = static_cast<int64_t>(idx / d.sxy);
				const int64_t a = std::max<int64_t>(pin_z, zs);
				const int64_t b = std::min<int64_t>(pin_z + static_cast<int64_t>(depth) + 1, ze);
				if (idx >= d.sxy * h.sz || b <= a) continue;   // pin does not touch the decoded range
				pin_index.push_back(idx); pin_depth.push_back(depth); pin_label.push_back(lv);
				pin_work_off.push_back(work);
				work += static_cast<uint64_t>(b - a);
			}
			i += num_pins * static_cast<uint64_t>(iw + dw);
			const uint64_t num_cc = rd_le(lb + i, npw); i += npw;
			if (num_cc > nlb || i + num_cc * static_cast<uint64_t>(ccw) > nlb) throw Error(CKL_ERR_RUNTIME, "crackle: pin section is malformed or corrupted.");
			uint64_t id = 0;
			for (uint64_t j = 0; j < num_cc; j++) {
				id = (id + rd_le(lb + i, ccw)) & 0xFFFFFFFFull; i += ccw;
				if (id >= d.comp_left && id < comp_right) { ccl_id.push_back(id); ccl_label.push_back(lv); }
			}
		}
		d.n_pins = pin_index.size();
		d.pin_total_work = work;
		d.n_ccl = ccl_id.size();
		upload(d.d_pin_index, pin_index, s);
		upload(d.d_pin_depth, pin_depth, s);
		upload(d.d_pin_label, pin_label, s);
		upload(d.d_pin_work_off, pin_work_off, s);
		upload(d.d_ccl_id, ccl_id, s);
		upload(d.d_ccl_label, ccl_label, s);
	}
	CKL_HIP(hipStreamSynchronize(s));   // host vectors above go out of scope
}

struct StageTimer {
	ckl_decoder& d;
	hipStream_t s;
	int i = 0;
	StageTimer(ckl_decoder& dec, hipStream_t st) : d(dec), s(st) { CKL_HIP(hipEventRecord(d.ev[0], s)); }
	void done(const char* name) {
		if (i >= kMaxStages) return;
		d.stage_name[i] = name;
		CKL_HIP(hipEventRecord(d.ev[i + 1], s));
		i++;
	}
};

template <typename OUT>
void launch_labels_and_paint(ckl_decoder& d, const RunGeom& g, const RunArrays& ra, void* out_device, int has_label, uint64_t label, StageTimer& st) {
	const Header& h = d.head;
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	OUT* run_label = reinterpret_cast<OUT*>(d.d_run_label.p);
	hipLaunchKernelGGL(k_run_labels<OUT>, dim3((d.max_rcap + kBlock - 1) / kBlock, ns), dim3(kBlock), 0, s,
		ra, d.d_label_map.p, d.d_comp_off.p, d.d_ncomp_expect.p, has_label ? 1u : 0u, label, run_label);
	st.done("k_run_labels");
	const uint32_t tiles = static_cast<uint32_t>((d.sxy + kPaintTile - 1) / kPaintTile);
	const bool fast = h.fortran_order && (h.sx % 4 == 0);
	if (fast) hipLaunchKernelGGL((k_paint_runs<OUT, true>), dim3(tiles, ns), dim3(kBlock), 0, s, g, ra, run_label, reinterpret_cast<OUT*>(out_device), d.sxy, ns, 1u);
	else hipLaunchKernelGGL((k_paint_runs<OUT, false>), dim3(tiles, ns), dim3(kBlock), 0, s, g, ra, run_label, reinterpret_cast<OUT*>(out_device), d.sxy, ns, h.fortran_order ? 1u : 0u);
	st.done("k_paint_runs");
}

void decoder_run(ckl_decoder& d, void* out_device, uint64_t out_capacity_bytes, int has_label, uint64_t label) {
	const Header& h = d.head;
	if (d.sxy == 0 || d.nslices == 0) return;
	const int ow = has_label ? 1 : h.data_width;
	const uint64_t need = d.sxy * d.nslices * static_cast<uint64_t>(ow);
	if (out_capacity_bytes < need) throw Error(CKL_ERR_ARG, "crackle_amd: output buffer too small: need " + std::to_string(need) + " bytes");
	hipStream_t s = d.stream;
	const uint32_t ns = d.nslices;
	const bool prof = getenv("CKL_PROFILE") != nullptr;
	auto h0 = std::chrono::steady_clock::now();

	StageTimer st(d, s);
	CKL_HIP(hipMemsetAsync(d.d_planes.p, 0, 2 * d.plane_words * ns * sizeof(uint32_t), s));
	CKL_HIP(hipMemsetAsync(d.d_slice_err.p, 0, ns * sizeof(uint32_t), s));
	st.done("memset planes");

	CrackArgs ca;
	ca.stream = d.d_stream.p;
	ca.code_off = d.d_code_off.p; ca.code_len = d.d_code_len.p;
	ca.cbase = d.d_cbase.p; ca.ccap = d.d_ccap.p; ca.nbase = d.d_nbase.p; ca.ncap = d.d_ncap.p;
	ca.sx = static_cast<int>(h.sx); ca.sy = static_cast<int>(h.sy);
	ca.xw = byte_width(static_cast<uint64_t>(h.sx) + 1); ca.yw = byte_width(static_cast<uint64_t>(h.sy) + 1);
	ca.markov_order = h.markov_model_order;
	ca.model = d.d_model.p; ca.ucode = d.d_ucode.p;
	ca.ctl_kind = d.d_ctl_kind.p; ca.ctl_pos = d.d_ctl_pos.p; ca.ctl_seg = d.d_ctl_seg.p;
	ca.seg_off = d.d_seg_off.p; ca.stack = d.d_stack.p; ca.nodes = d.d_nodes.p;
	ca.planeV = d.d_planes.p; ca.planeH = d.d_planes.p + d.plane_words * ns;
	ca.row_words = d.row_words; ca.plane_words = d.plane_words;
	ca.slice_err = d.d_slice_err.p;
	if (getenv("CKL_DECODE_DIAG")) {
		DevBuf<unsigned long long> d_diag;
		d_diag.ensure(static_cast<size_t>(ns) * 8);
		CKL_HIP(hipMemsetAsync(d_diag.p, 0, static_cast<size_t>(ns) * 64, s));
		hipLaunchKernelGGL(k_decode_cracks<true>, dim3(ns), dim3(kCrackBlock), 0, s, ca, d_diag.p);
		std::vector<unsigned long long> dg(static_cast<size_t>(ns) * 8);
		CKL_HIP(hipMemcpyAsync(dg.data(), d_diag.p, dg.size() * 8, hipMemcpyDeviceToHost, s));
		CKL_HIP(hipStreamSynchronize(s));
		double m[8] = { 0 };
		for (uint32_t zi = 0; zi < ns; zi++) for (int k = 0; k < 8; k++) m[k] += static_cast<double>(dg[zi * 8 + k]) / ns;
		fprintf(stderr, "[ckl decode_cracks diag, mean cycles per slice] A(boc)=%.0f B(symbols)=%.0f C(match)=%.0f D(raster)=%.0f  codes=%.0f controls=%.0f\n", m[0], m[1], m[2], m[3], m[4], m[5]);
	}
	else hipLaunchKernelGGL(k_decode_cracks<false>, dim3(ns), dim3(kCrackBlock), 0, s, ca, static_cast<unsigned long long*>(nullptr));
	st.done("k_decode_cracks");

	RunGeom g;
	g.planeV = ca.planeV; g.planeH = ca.planeH; g.row_words = d.row_words; g.plane_words = d.plane_words;
	g.flip = (h.crack_format == IMPERMISSIBLE) ? 1u : 0u;
	g.sx = h.sx; g.sy = h.sy;
	RunArrays ra;
	ra.word_base = d.d_word_base.p; ra.rbase = d.d_rbase.p; ra.rcap = d.d_rcap.p;
	ra.parent = d.d_parent.p; ra.run_start = d.d_run_start.p; ra.run_cc = d.d_run_cc.p;
	ra.nruns = d.d_nruns.p; ra.ncomp = d.d_ncomp.p; ra.slice_err = d.d_slice_err.p;

	hipLaunchKernelGGL(k_run_index, dim3(ns), dim3(kBlock), 0, s, g, ra);
	st.done("k_run_index");
	launch_run_union(s, ns, g, ra);
	st.done("k_run_union");
	ResolveScratch rs;
	rs.run_local = d.d_run_local.p; rs.blk_roots = d.d_blk_roots.p; rs.nblk = (d.max_rcap + kBlock - 1) / kBlock;
	launch_run_resolve(s, ns, ra, rs, d.d_G.p, static_cast<uint32_t>(d.sxy), d.idbits, d.d_crc_acc.p, nullptr);
	st.done("k_run_resolve");

	// component -> label
	const uint64_t nlm = d.total_comp;
	if (h.label_format == FLAT) {
		if (nlm) {
			const uint8_t* keys = d.d_stream.p + d.keys_offset + d.comp_left * static_cast<uint64_t>(d.key_width);
			const uint8_t* uniq = d.d_stream.p + h.header_bytes() + h.grid_index_bytes() + d.uniq_offset;
			hipLaunchKernelGGL(k_label_map_flat, dim3(static_cast<uint32_t>((nlm + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
				keys, d.key_width, uniq, h.stored_data_width, d.num_unique, h.is_signed ? 1u : 0u, nlm, d.d_label_map.p);
		}
	}
	else {
		if (nlm) hipLaunchKernelGGL(k_fill_u64, dim3(static_cast<uint32_t>((nlm + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d.d_label_map.p, d.bgcolor, nlm);
		if (d.n_ccl) hipLaunchKernelGGL(k_label_map_ccids, dim3(static_cast<uint32_t>((d.n_ccl + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
			d.d_ccl_id.p, d.d_ccl_label.p, d.n_ccl, d.comp_left, d.comp_left + nlm, d.d_label_map.p);
		if (d.pin_total_work) hipLaunchKernelGGL(k_label_map_pins, dim3(static_cast<uint32_t>((d.pin_total_work + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
			d.d_pin_index.p, d.d_pin_depth.p, d.d_pin_label.p, d.d_pin_work_off.p, d.n_pins, d.pin_total_work,
			g, ra, d.sxy, d.z_start, d.z_end, d.d_comp_off.p, d.d_ncomp_expect.p, d.d_label_map.p);
	}
	st.done("k_label_map");

	if (has_label || h.data_width == 1) launch_labels_and_paint<uint8_t>(d, g, ra, out_device, has_label, label, st);
	else if (h.data_width == 2) launch_labels_and_paint<uint16_t>(d, g, ra, out_device, has_label, label, st);
	else if (h.data_width == 4) launch_labels_and_paint<uint32_t>(d, g, ra, out_device, has_label, label, st);
	else launch_labels_and_paint<uint64_t>(d, g, ra, out_device, has_label, label, st);

	hipLaunchKernelGGL(k_check, dim3((ns + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
		d.d_crc_acc.p, d.d_crc_expect.p, d.d_ncomp.p, d.d_ncomp_expect.p,
		d.check_crc ? 1u : 0u, d.crc_fix, ns, d.d_slice_err.p);
	st.done("k_check");
	d.n_stages = st.i;

	auto h1 = std::chrono::steady_clock::now();
	std::vector<uint32_t> errs(ns);
	CKL_HIP(hipMemcpyAsync(errs.data(), d.d_slice_err.p, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	CKL_HIP(hipStreamSynchronize(s));
	auto h2 = std::chrono::steady_clock::now();
	if (prof) fprintf(stderr, "[ckl decode host ms] enqueue=%.2f wait=%.2f\n",
		std::chrono::duration<double, std::milli>(h1 - h0).count(), std::chrono::duration<double, std::milli>(h2 - h1).count());
	CKL_HIP(hipGetLastError());
	CKL_HIP(hipEventElapsedTime(&d.pipeline_ms, d.ev[0], d.ev[d.n_stages]));
	for (int i = 0; i < d.n_stages; i++) CKL_HIP(hipEventElapsedTime(&d.stage_ms[i], d.ev[i], d.ev[i + 1]));
	for (uint32_t zi = 0; zi < ns; zi++) {
		const uint32_t e = errs[zi];
		if (!e) continue;
		const std::string z = std::to_string(d.z_start + zi);
		if (e & (ERR_BOC | ERR_RANGE | ERR_CAPACITY)) throw Error(CKL_ERR_RUNTIME, "crackle: crack code is malformed or corrupted on z=" + z);
		if (e & ERR_NCOMP) throw Error(CKL_ERR_RUNTIME, "crackle: component count does not match the label section on z=" + z);
		throw Error(CKL_ERR_CRC, "crackle: crack code crc mismatch on z=" + z);
	}
}

}  // namespace

extern "C" {

int ckl_decoder_create(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int device, ckl_decoder** out) {
	try {
		if (!buf || !out) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		// header problems are format errors even when no device is present
		if (n < Header::kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
		(void)Header::parse(buf, n);
		select_device(device);
		std::unique_ptr<ckl_decoder> d(new ckl_decoder());
		d->device = device;
		CKL_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
		for (auto& e : d->ev) CKL_HIP(hipEventCreate(&e));
		decoder_build(*d, buf, n, z_start, z_end);
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
		decoder_run(*d, out_device, out_capacity_bytes, has_label, label);
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
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

void ckl_decoder_destroy(ckl_decoder* d) { delete d; }

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
