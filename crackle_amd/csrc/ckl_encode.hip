// Encode path: label volume resident in HBM -> .ckl bytes.
// Replaces crackle::compress<LABEL> (src/crackle.hpp:34-257) and what it calls:
//   lib::max_label / pixel_pairs                    src/lib.hpp:224-256        k_stats
//   crackcodes::Graph::init                         src/crackcodes.hpp:66-125  k_label_planes + k_trail_graph (ckl_trail.hpp)
//   create_crack_codes walk + remove_initial_branch +
//     remove_spurious_branches + symbols_to_codepoints
//                                                   src/crackcodes.hpp:128-281, 374-453  k_trail_* (ckl_trail.hpp)
//   pack_codepoints / write_boc_index               src/crackcodes.hpp:318-372, 455-496  k_finish
//   markov::gather_statistics / encode_markov       src/markov.hpp:193-220, 422-473      k_markov_hist / k_markov_pack
//   cc3d::connected_components2d_4 + relabel        src/cc3d.hpp:114-144, 257-369        ckl_runs.hpp (runs of the label planes)
//   labels::encode_flat                             src/labels.hpp:30-155      k_run_resolve (crc), k_mapping_comps + sort/unique
//   stream assembly                                 src/crackle.hpp:171-216    host
#include "ckl_common.hpp"
#include "ckl_runs.hpp"
#include "ckl_trail.hpp"
#include "ckl_pins_dev.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>

namespace ckl {

using namespace dev;

// ------------------------------------------------------------------------------
// whole-volume reductions (lib.hpp:224-256)
// ------------------------------------------------------------------------------
template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_stats(const LABEL* __restrict__ labels, uint64_t voxels, unsigned long long* __restrict__ out /* [0]=max [1]=pairs */) {
	__shared__ unsigned long long s_max[kWaves], s_pairs[kWaves];
	unsigned long long mx = 0, pairs = 0;
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
	for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < voxels; i += stride) {
		const LABEL v = labels[i];
		if (static_cast<unsigned long long>(v) > mx) mx = v;
		if (i > 0) pairs += (labels[i - 1] == v);
	}
	for (int d = kWave / 2; d >= 1; d >>= 1) {
		const unsigned long long om = __shfl_xor(mx, d, kWave);
		const unsigned long long op = __shfl_xor(pairs, d, kWave);
		mx = om > mx ? om : mx;
		pairs += op;
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) { s_max[wave] = mx; s_pairs[wave] = pairs; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < kWaves; w++) { mx = s_max[w] > mx ? s_max[w] : mx; pairs += s_pairs[w]; }
		atomicMax(out, mx);
		atomicAdd(out + 1, pairs);
	}
}

// ------------------------------------------------------------------------------
// label planes: one pass over the labels produces, per slice, two row-aligned bit
// planes of "differs from the left neighbour" (V) and "differs from the upper
// neighbour" (H).  Both the crack graph (crackcodes.hpp:66-125) and the 4-connected
// components (cc3d.hpp:257-369) are derived from these 0.25 B/voxel instead of
// re-reading the labels.  One wavefront per 64 pixels of a row: coalesced loads, the
// left neighbour comes from a lane shuffle, the plane words from a ballot.
// grid = (ceil(chunks_per_row * sy / 4), nslices), chunks_per_row = ceil(sx / 64)
// ------------------------------------------------------------------------------
constexpr uint32_t kPlaneUnroll = 4;   // 64-pixel chunks per wavefront (loads issued together)

template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_label_planes(
	const LABEL* __restrict__ labels, uint32_t sx, uint32_t sy, uint32_t chunks_per_row,
	uint32_t* __restrict__ planeV, uint32_t* __restrict__ planeH, uint32_t row_words, uint64_t plane_words,
	uint32_t* __restrict__ count_v, uint32_t* __restrict__ count_h
) {
	__shared__ uint32_t s_red[2 * kWaves];
	const uint32_t zi = blockIdx.y;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t units = chunks_per_row * sy;
	const uint32_t unit0 = (blockIdx.x * kWaves + wave) * kPlaneUnroll;
	const LABEL* slice = labels + static_cast<uint64_t>(zi) * sy * sx;
	LABEL v[kPlaneUnroll], up[kPlaneUnroll], lf[kPlaneUnroll];
	uint32_t xs[kPlaneUnroll], ys[kPlaneUnroll];
#pragma unroll
	for (uint32_t k = 0; k < kPlaneUnroll; k++) {
		const uint32_t unit = unit0 + k;
		const uint32_t y = unit / chunks_per_row;
		const uint32_t x = (unit - y * chunks_per_row) * 64u + lane;
		xs[k] = x; ys[k] = y;
		const bool valid = unit < units && x < sx;
		const LABEL* row = slice + static_cast<uint64_t>(y) * sx;
		v[k] = valid ? row[x] : LABEL(0);
		up[k] = (valid && y > 0) ? row[static_cast<int64_t>(x) - static_cast<int64_t>(sx)] : v[k];
		lf[k] = (valid && lane == 0 && x > 0) ? row[x - 1] : v[k];
	}
	uint32_t nv = 0, nh = 0;
#pragma unroll
	for (uint32_t k = 0; k < kPlaneUnroll; k++) {
		const uint32_t unit = unit0 + k;
		const bool valid = unit < units && xs[k] < sx;
		LABEL left = __shfl_up(v[k], 1, kWave);
		if (lane == 0) left = lf[k];
		const bool dv = valid && xs[k] > 0 && v[k] != left;
		const bool dh = valid && v[k] != up[k];
		const unsigned long long mv = __ballot(dv), mh = __ballot(dh);
		if (lane == 0 && unit < units) {
			const uint32_t c = xs[k] >> 6;
			const uint64_t wbase = zi * plane_words + static_cast<uint64_t>(ys[k]) * row_words + c * 2u;
			planeV[wbase] = static_cast<uint32_t>(mv);
			planeH[wbase] = static_cast<uint32_t>(mh);
			if (c * 2u + 1u < row_words) {
				planeV[wbase + 1] = static_cast<uint32_t>(mv >> 32);
				planeH[wbase + 1] = static_cast<uint32_t>(mh >> 32);
			}
			nv += __popcll(mv);
			nh += __popcll(mh);
		}
	}
	if (lane == 0) { s_red[wave] = nv; s_red[kWaves + wave] = nh; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tv = 0, th = 0;
		for (int w = 0; w < kWaves; w++) { tv += s_red[w]; th += s_red[kWaves + w]; }
		if (tv) atomicAdd(count_v + zi, tv);
		if (th) atomicAdd(count_h + zi, th);
	}
}

// Fast path (sx a multiple of the 16-byte vector, 16-byte aligned volume): one wavefront
// owns a strip of 64 vectors x kBandRows rows and walks down it, so every label is loaded
// once with 16-byte loads (the row above stays in registers) and the same pass yields
// lib::max_label and lib::pixel_pairs (lib.hpp:224-256).  Plane words are assembled from
// the per-lane bit groups with an OR butterfly.  No atomics: per-workgroup partial counts
// go to `partial` and are folded per slice by k_planes_reduce.
// grid = (ceil(strips * bands / 4), nslices)
constexpr uint32_t kBandRows = 32;

// 16 bytes that nobody reads again
template <typename V, typename T>
__device__ __forceinline__ V nt_load_vec(const T* p) {
	typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
	static_assert(sizeof(V) == 16, "one 16-byte vector");
	const u32x4_t raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
	V v;
	__builtin_memcpy(&v, &raw, 16);
	return v;
}

template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_label_planes_fast(
	const LABEL* __restrict__ labels, uint32_t sx, uint32_t sy, uint32_t strips, uint32_t bands,
	uint32_t* __restrict__ planeV, uint32_t* __restrict__ planeH, uint32_t row_words, uint64_t plane_words,
	uint32_t* __restrict__ partial /* [nslices][gridDim.x][4]: nv, nh, pairs, - */, unsigned long long* __restrict__ partial_max,
	unsigned long long* __restrict__ total_pairs /* zeroed here, summed by k_planes_reduce behind this kernel */
) {
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *total_pairs = 0;
	constexpr uint32_t P = 16 / sizeof(LABEL);        // pixels per lane
	constexpr uint32_t G = 32 / P;                    // lanes per plane word
	struct alignas(16) Vec { LABEL v[P]; };
	__shared__ uint32_t s_red[3 * kWaves];
	__shared__ unsigned long long s_max[kWaves];
	const uint32_t zi = blockIdx.y;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t task = blockIdx.x * kWaves + wave;
	uint32_t nv = 0, nh = 0, pairs = 0;
	LABEL mxl = 0;      // (in the label's own width: a 64-bit compare + select per pixel was a tenth of the kernel's VALU work)
	if (task < strips * bands) {
		const uint32_t band = task / strips, strip = task - band * strips;
		const uint32_t x = strip * (64u * P) + lane * P;
		const uint32_t y0 = band * kBandRows, y1 = min(y0 + kBandRows, sy);
		const bool active = x < sx;
		const uint64_t slice_off = static_cast<uint64_t>(zi) * sy * sx;
		const LABEL* col = labels + slice_off + x;
		Vec prev;
#pragma unroll
		for (uint32_t i = 0; i < P; i++) prev.v[i] = 0;
		if (active && y0 > 0) prev = *reinterpret_cast<const Vec*>(col + static_cast<uint64_t>(y0 - 1) * sx);
		uint32_t* pv = planeV + zi * plane_words;
		uint32_t* ph = planeH + zi * plane_words;
		const uint32_t word = (x >> 5);
		// kRowsAhead rows are loaded before the first of them is compared: one 16-byte load per
		// lane in flight does not cover the memory latency at the occupancy the CU allows
		constexpr uint32_t kRowsAhead = 4;
		for (uint32_t yb = y0; yb < y1; yb += kRowsAhead) {
			Vec rows[kRowsAhead];
			LABEL edges[kRowsAhead];
			bool have_edges[kRowsAhead];
#pragma unroll
			for (uint32_t r = 0; r < kRowsAhead; r++) {
				const uint32_t y = yb + r;
				edges[r] = 0; have_edges[r] = false;
				if (active && y < y1) {
					rows[r] = nt_load_vec<Vec>(col + static_cast<uint64_t>(y) * sx);      // read once: non-temporal (7.0 against 6.2 TB/s in a pure read, tools/micro/load_bw.hip)
					if (lane == 0) {
						// linear predecessor of the strip's first pixel (previous row / slice when x == 0)
						const uint64_t lin = slice_off + static_cast<uint64_t>(y) * sx + x;
						if (lin > 0) { edges[r] = labels[lin - 1]; have_edges[r] = true; }
					}
				}
				else rows[r] = prev;
			}
#pragma unroll
			for (uint32_t r = 0; r < kRowsAhead; r++) {
				const uint32_t y = yb + r;
				if (y >= y1) break;
				const Vec cur = active ? rows[r] : prev;
				LABEL left = __shfl_up(cur.v[P - 1], 1, kWave);
				bool have_left = true;
				if (lane == 0) { left = edges[r]; have_left = have_edges[r]; }
				uint32_t bv = 0, bh = 0;
				if (active) {
					// bit i: pixel i differs from its left / upper neighbour; the border cases (no left neighbour at
					// the volume's first voxel, x == 0, y == 0) are masked once per row instead of tested per pixel
#pragma unroll
					for (uint32_t i = 0; i < P; i++) {
						const LABEL l = i ? cur.v[i - 1] : left;
						bv |= (cur.v[i] != l ? 1u : 0u) << i;
						bh |= (cur.v[i] != prev.v[i] ? 1u : 0u) << i;
						mxl = cur.v[i] > mxl ? cur.v[i] : mxl;
					}
					// lib::pixel_pairs counts equal LINEAR neighbours (lib.hpp:249-256): every pixel but the volume's first has one
					pairs += P - __popc(bv) - ((have_left || (bv & 1u)) ? 0u : 1u);
					if (x == 0) bv &= ~1u;      // no crack along the image's left border
					if (y == 0) bh = 0u;
				}
				nv += __popc(bv); nh += __popc(bh);
				bv <<= (lane % G) * P; bh <<= (lane % G) * P;
#pragma unroll
				for (uint32_t sft = 1; sft < G; sft <<= 1) {
					bv |= __shfl_xor(bv, sft, kWave);
					bh |= __shfl_xor(bh, sft, kWave);
				}
				if ((lane % G) == 0 && word < row_words && active) {
					pv[static_cast<uint64_t>(y) * row_words + word] = bv;
					ph[static_cast<uint64_t>(y) * row_words + word] = bh;
				}
				prev = cur;
			}
		}
	}
	nv = wave_sum(nv); nh = wave_sum(nh); pairs = wave_sum(pairs);
	unsigned long long mx = static_cast<unsigned long long>(mxl);
	for (int d = kWave / 2; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d, kWave); mx = o > mx ? o : mx; }
	if (lane == 0) { s_red[wave] = nv; s_red[kWaves + wave] = nh; s_red[2 * kWaves + wave] = pairs; s_max[wave] = mx; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tv = 0, th = 0, tp = 0;
		unsigned long long tm = 0;
		for (int w = 0; w < kWaves; w++) { tv += s_red[w]; th += s_red[kWaves + w]; tp += s_red[2 * kWaves + w]; tm = s_max[w] > tm ? s_max[w] : tm; }
		const uint64_t o = static_cast<uint64_t>(zi) * gridDim.x + blockIdx.x;
		partial[o * 4 + 0] = tv; partial[o * 4 + 1] = th; partial[o * 4 + 2] = tp; partial[o * 4 + 3] = 0;
		partial_max[o] = tm;
	}
}

// The same pass with nothing but label loads in the wavefront's memory queue (round 5).  k_label_planes_fast stores two
// plane words per row from inside its row loop; loads and stores retire on ONE counter on this architecture and
// "mixed" means "wait for all": every group of four rows drained its stores before the next loads went out, and the
// kernel read 2.15 GB in 0.42 - 0.46 ms where a plain read takes 0.31.  Here a row's plane words go to LDS (2 KiB per
// wavefront for a strip of 32 rows) and leave in two 16-byte stores per lane after the loop; the label of the pixel left
// of the strip is a SCALAR load (its own counter); and the rows sit in a window of kRowsAhead registers that is
// refilled as it is consumed, so that kRowsAhead - 1 row loads are always in flight.
// (or_over_lane_group / wave_shift_up1_label: DPP instead of ds_bpermute.)  Same grid, same outputs.
template <typename T>
__device__ __forceinline__ T wave_shift_up1_label(T v) {
	if constexpr (sizeof(T) == 8) {
		const uint32_t lo = dev::wave_shift_up1(static_cast<uint32_t>(v), 0u), hi = dev::wave_shift_up1(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), 0u);
		return static_cast<T>((static_cast<uint64_t>(hi) << 32) | lo);
	}
	else return static_cast<T>(dev::wave_shift_up1(static_cast<uint32_t>(v), 0u));
}
// OR of a value over every aligned group of G lanes (2, 4, 8, 16), the result in all lanes of the group: quad permutes,
// then mirrors of half a row / a row of 16 lanes (a mirror pairs every lane with one of the other half: all an OR needs)
template <uint32_t G>
__device__ __forceinline__ uint32_t or_over_lane_group(uint32_t v) {
	static_assert(G == 2 || G == 4 || G == 8 || G == 16, "lane groups inside a row of 16");
	if constexpr (G >= 2) v |= dev::dpp_u32<0xB1, 0xF>(0u, v);      // quad_perm [1, 0, 3, 2]
	if constexpr (G >= 4) v |= dev::dpp_u32<0x4E, 0xF>(0u, v);      // quad_perm [2, 3, 0, 1]
	if constexpr (G >= 8) v |= dev::dpp_u32<0x141, 0xF>(0u, v);     // row_half_mirror
	if constexpr (G >= 16) v |= dev::dpp_u32<0x140, 0xF>(0u, v);    // row_mirror
	return v;
}

template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_label_planes_stream(
	const LABEL* __restrict__ labels, uint32_t sx, uint32_t sy, uint32_t strips, uint32_t bands,
	uint32_t* __restrict__ planeV, uint32_t* __restrict__ planeH, uint32_t row_words, uint64_t plane_words,
	uint32_t* __restrict__ partial, unsigned long long* __restrict__ partial_max, unsigned long long* __restrict__ total_pairs
) {
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *total_pairs = 0;
	constexpr uint32_t P = 16 / sizeof(LABEL);        // pixels per lane
	constexpr uint32_t G = 32 / P;                    // lanes per plane word
	constexpr uint32_t WPR = 64 / G;                  // plane words of a strip's row
	struct alignas(16) Vec { LABEL v[P]; };
	__shared__ uint32_t s_red[3 * kWaves];
	__shared__ unsigned long long s_max[kWaves];
	__shared__ __attribute__((aligned(16))) uint32_t s_words[kWaves][2][kBandRows * WPR];      // plane words of the wavefront's strip: V, H
	const uint32_t zi = blockIdx.y;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t task = blockIdx.x * kWaves + wave;
	uint32_t nv = 0, nh = 0, pairs = 0;
	LABEL mxl = 0;
	if (task < strips * bands) {
		const uint32_t band = __builtin_amdgcn_readfirstlane(task / strips), strip = __builtin_amdgcn_readfirstlane(task - (task / strips) * strips);
		const uint32_t x0 = strip * (64u * P);            // the strip's first pixel column (wave-uniform)
		const uint32_t x = x0 + lane * P;
		const uint32_t y0 = band * kBandRows, y1 = min(y0 + kBandRows, sy);
		const bool active = x < sx;
		const uint64_t slice_off = static_cast<uint64_t>(zi) * sy * sx;
		const LABEL* col = labels + slice_off + (active ? x : 0u);      // (lanes past the row's end load the row's first vector: every load is unconditional)
		Vec prev;
#pragma unroll
		for (uint32_t i = 0; i < P; i++) prev.v[i] = 0;
		if (y0 > 0) prev = nt_load_vec<Vec>(col + static_cast<uint64_t>(y0 - 1) * sx);
		uint32_t* sv = s_words[wave][0];
		uint32_t* sh = s_words[wave][1];
		// the pixel before the strip's first one in LINEAR order (the previous row / slice when the strip starts a row):
		// one label per row, the same for the whole wavefront -> a scalar load of the 4 bytes that hold it
		auto edge_of = [&](uint32_t y, bool& have) -> LABEL {
			const uint64_t lin = slice_off + static_cast<uint64_t>(y) * sx + x0;      // wave-uniform
			have = lin > 0;
			const uint64_t at = have ? lin - 1 : 0;
			if constexpr (sizeof(LABEL) >= 4) return labels[at];
			else {
				const uint64_t byte = at * sizeof(LABEL);
				const uint32_t w = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(labels) & ~static_cast<uintptr_t>(3))[(byte + (reinterpret_cast<uintptr_t>(labels) & 3u)) >> 2];
				return static_cast<LABEL>(w >> (8u * static_cast<uint32_t>((byte + (reinterpret_cast<uintptr_t>(labels) & 3u)) & 3u)));
			}
		};
		constexpr uint32_t kRowsAhead = 4;
		Vec rows[kRowsAhead];
		LABEL edges[kRowsAhead];
		bool have_edges[kRowsAhead];
		auto request = [&](uint32_t r, uint32_t y) {
			const uint32_t yl = y < y1 ? y : y1 - 1u;      // (rows past the band re-read its last row: no load in a branch of its own)
			rows[r] = nt_load_vec<Vec>(col + static_cast<uint64_t>(yl) * sx);
			edges[r] = edge_of(yl, have_edges[r]);
		};
#pragma unroll
		for (uint32_t r = 0; r < kRowsAhead; r++) request(r, y0 + r);
		// (a fixed trip count, fully unrolled: straight-line code, so that the compiler waits for exactly the load a row
		// needs — across a loop's back edge it waits for all of them)
#pragma unroll
		for (uint32_t it = 0; it < kBandRows / kRowsAhead; it++) {
			const uint32_t yb = y0 + it * kRowsAhead;
#pragma unroll
			for (uint32_t r = 0; r < kRowsAhead; r++) {
				const uint32_t y = yb + r;
				const Vec cur = rows[r];
				const LABEL edge = edges[r];
				const bool have_edge = have_edges[r];
				request(r, y + kRowsAhead);      // the slot is free: its next row goes out before this one is looked at
				if (y >= y1) continue;
				LABEL left = wave_shift_up1_label(cur.v[P - 1]);
				bool have_left = true;
				if (lane == 0) { left = edge; have_left = have_edge; }
				uint32_t bv = 0, bh = 0;
				if (active) {
#pragma unroll
					for (uint32_t i = 0; i < P; i++) {
						const LABEL l = i ? cur.v[i - 1] : left;
						bv |= (cur.v[i] != l ? 1u : 0u) << i;
						bh |= (cur.v[i] != prev.v[i] ? 1u : 0u) << i;
						mxl = cur.v[i] > mxl ? cur.v[i] : mxl;
					}
					// lib::pixel_pairs counts equal LINEAR neighbours (lib.hpp:249-256): every pixel but the volume's first has one
					pairs += P - __popc(bv) - ((have_left || (bv & 1u)) ? 0u : 1u);
					if (x == 0) bv &= ~1u;      // no crack along the image's left border
					if (y == 0) bh = 0u;
				}
				nv += __popc(bv); nh += __popc(bh);
				bv <<= (lane % G) * P; bh <<= (lane % G) * P;
				bv = or_over_lane_group<G>(bv); bh = or_over_lane_group<G>(bh);
				if ((lane % G) == 0) { sv[(y - y0) * WPR + lane / G] = bv; sh[(y - y0) * WPR + lane / G] = bh; }
				prev = cur;
			}
		}
		// the strip's plane words out: row r of the band, words word0 .. word0 + WPR - 1 of the row (LDS accesses of one
		// wavefront are ordered: no barrier)
		const uint32_t word0 = x0 >> 5;
		uint32_t* pv = planeV + zi * plane_words;
		uint32_t* ph = planeH + zi * plane_words;
		constexpr uint32_t kVecWords = WPR >= 4 ? 4u : WPR;      // words per lane and store
		constexpr uint32_t kLanesPerRow = WPR / kVecWords;
		for (uint32_t r0 = 0; r0 < y1 - y0; r0 += 64u / kLanesPerRow) {
			const uint32_t r = r0 + lane / kLanesPerRow, w = (lane % kLanesPerRow) * kVecWords;
			if (r >= y1 - y0) continue;
			const uint64_t at = static_cast<uint64_t>(y0 + r) * row_words + word0 + w;
			if (word0 + w + kVecWords <= row_words && ((at & (kVecWords - 1u)) == 0u) && kVecWords == 4u) {
				*reinterpret_cast<uint4*>(pv + at) = *reinterpret_cast<const uint4*>(sv + r * WPR + w);
				*reinterpret_cast<uint4*>(ph + at) = *reinterpret_cast<const uint4*>(sh + r * WPR + w);
			}
			else {
				for (uint32_t q = 0; q < kVecWords; q++) if (word0 + w + q < row_words) { pv[at + q] = sv[r * WPR + w + q]; ph[at + q] = sh[r * WPR + w + q]; }
			}
		}
	}
	nv = wave_sum(nv); nh = wave_sum(nh); pairs = wave_sum(pairs);
	unsigned long long mx = static_cast<unsigned long long>(mxl);
	for (int d = kWave / 2; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d, kWave); mx = o > mx ? o : mx; }
	if (lane == 0) { s_red[wave] = nv; s_red[kWaves + wave] = nh; s_red[2 * kWaves + wave] = pairs; s_max[wave] = mx; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tv = 0, th = 0, tp = 0;
		unsigned long long tm = 0;
		for (int w = 0; w < kWaves; w++) { tv += s_red[w]; th += s_red[kWaves + w]; tp += s_red[2 * kWaves + w]; tm = s_max[w] > tm ? s_max[w] : tm; }
		const uint64_t o = static_cast<uint64_t>(zi) * gridDim.x + blockIdx.x;
		partial[o * 4 + 0] = tv; partial[o * 4 + 1] = th; partial[o * 4 + 2] = tp; partial[o * 4 + 3] = 0;
		partial_max[o] = tm;
	}
}

// ckl_reencode_markov: the decoder's crack planes become the encoder's "differs from the
// neighbour" planes (a crack is a differing pair for IMPERMISSIBLE streams and an equal pair for
// PERMISSIBLE ones; image-border pairs carry neither) with their per-slice population counts.
// grid = (ceil(plane_words / 256), nslices); counts: [nslices] v, then [nslices] h, zeroed by the host
__global__ void __launch_bounds__(kBlock) k_planes_from_cracks(
	const uint32_t* __restrict__ crackV, const uint32_t* __restrict__ crackH, uint32_t sx, uint32_t sy,
	uint32_t row_words, uint64_t plane_words, uint32_t permissible,
	uint32_t* __restrict__ planeV, uint32_t* __restrict__ planeH, uint32_t* __restrict__ counts, uint32_t nslices
) {
	__shared__ uint32_t s_red[kWaves];
	const uint32_t zi = blockIdx.y;
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	uint32_t nv = 0, nh = 0;
	if (i < plane_words) {
		const uint32_t y = static_cast<uint32_t>(i / row_words);
		const uint32_t w = static_cast<uint32_t>(i - static_cast<uint64_t>(y) * row_words);
		const uint32_t left = sx - w * 32u;
		const uint32_t in_row = left >= 32u ? 0xFFFFFFFFu : ((1u << left) - 1u);           // x < sx
		const uint32_t mv = in_row & (w == 0 ? 0xFFFFFFFEu : 0xFFFFFFFFu);                   // 1 <= x < sx
		const uint32_t mh = y >= 1 ? in_row : 0u;                                            // y >= 1
		const uint64_t at = zi * plane_words + i;
		const uint32_t inv = permissible ? 0xFFFFFFFFu : 0u;
		const uint32_t v = (crackV[at] ^ inv) & mv, h = (crackH[at] ^ inv) & mh;
		planeV[at] = v; planeH[at] = h;
		nv = __popc(v); nh = __popc(h);
	}
	nv = block_sum(nv, s_red); nh = block_sum(nh, s_red);
	if (threadIdx.x == 0) {
		if (nv) atomicAdd(counts + zi, nv);
		if (nh) atomicAdd(counts + nslices + zi, nh);
	}
}

// grid = nslices: folds the per-workgroup partials of one slice
__global__ void __launch_bounds__(kBlock) k_planes_reduce(
	const uint32_t* __restrict__ partial, const unsigned long long* __restrict__ partial_max, uint32_t nblk,
	unsigned long long* __restrict__ out /* [nslices][4]: count_v, count_h, pairs, max */,
	unsigned long long* __restrict__ total_pairs /* the volume's equal pixel pairs: k_trail_graph decides the crack format from it */
) {
	__shared__ uint32_t s_red[kWaves];
	__shared__ unsigned long long s_max[kWaves];
	const uint32_t zi = blockIdx.x;
	uint32_t tv = 0, th = 0, tp = 0;
	unsigned long long tm = 0;
	for (uint32_t b = threadIdx.x; b < nblk; b += kBlock) {
		const uint64_t o = static_cast<uint64_t>(zi) * nblk + b;
		tv += partial[o * 4 + 0]; th += partial[o * 4 + 1]; tp += partial[o * 4 + 2];
		const unsigned long long m = partial_max[o];
		tm = m > tm ? m : tm;
	}
	tv = block_sum(tv, s_red); th = block_sum(th, s_red); tp = block_sum(tp, s_red);
	for (int d = kWave / 2; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(tm, d, kWave); tm = o > tm ? o : tm; }
	if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = tm;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 0; w < kWaves; w++) tm = s_max[w] > tm ? s_max[w] : tm;
		out[4ull * zi + 0] = tv; out[4ull * zi + 1] = th; out[4ull * zi + 2] = tp; out[4ull * zi + 3] = tm;
		atomicAdd(total_pairs, static_cast<unsigned long long>(tp));
	}
}

enum : uint8_t { CODE_UP = 0, CODE_RIGHT = 1, CODE_DOWN = 2, CODE_LEFT = 3, CODE_NONE = 0xFE, CODE_TOMB = 0xFF };

// ------------------------------------------------------------------------------
// k_finish: chain order, compaction, difference coding, 2-bit packing, BOC index
// (crackcodes.hpp:318-372, 455-496); one workgroup per slice.
// ------------------------------------------------------------------------------
struct FinishArgs {
	int sx, sy, xw, yw;
	uint32_t z0;               // first slice of this launch
	uint32_t markov;           // 1: write difference codes to dcode instead of packing
	const uint64_t* cbase;
	const uint64_t* kbase;
	const uint32_t* n_chains;
	const uint32_t* n_raw;
	const uint32_t* n_valid;
	const uint8_t* cp;
	const uint32_t* chain_node;
	const uint32_t* chain_off;
	const uint32_t* chain_clen;
	uint32_t* chain_order;     // scratch [kcap]
	uint32_t* chain_dst;       // scratch [kcap]
	uint32_t* chain_vstart;    // scratch [kcap]
	uint8_t* fcode;            // final-order code points   (at cbase)
	uint8_t* dcode;            // final-order difference codes (at cbase; markov only)
	const uint64_t* pbase;     // per slice: base into payload (4-byte aligned)
	uint8_t* payload;
	const uint64_t* bbase;     // per slice: base into boc
	uint8_t* boc;
	uint32_t* payload_len;     // [nslices]
	uint32_t* boc_len;         // [nslices]
};

__device__ __forceinline__ void put_le_dev(uint8_t* p, uint32_t v, int w) {
	for (int i = 0; i < w; i++) p[i] = static_cast<uint8_t>((v >> (8 * i)) & 0xFF);
}

// 1024 threads per slice: two workgroups share a CU, and the phases below are rounds of
// latency-bound byte accesses, so the time goes with the rounds per thread
constexpr int kFinishBlock = 1024;
constexpr int kFinishWaves = kFinishBlock / kWave;

// chains of a slice whose tables k_finish stages in LDS (a slice of the bench volume has ~100; noise has thousands)
constexpr uint32_t kFinishChains = 2048;

__global__ void __launch_bounds__(kFinishBlock) k_finish(FinishArgs a) {
	__shared__ uint32_t s_scan[kFinishWaves];
	// phase 1 is one thread's serial walk over the chains — an insertion sort and the BOC index, every step a
	// dependent access — and in global memory each of those was a trip of a microsecond: a third of the kernel.  The
	// chain tables are staged here (5 x 8 KiB; two workgroups share a CU) and phase 2 reads them back from here.
	__shared__ uint32_t s_node[kFinishChains], s_clen[kFinishChains], s_off[kFinishChains], s_ord[kFinishChains], s_dst[kFinishChains], s_vst[kFinishChains];
	const uint32_t zi = blockIdx.x + a.z0;
	const int tid = threadIdx.x;
	const uint32_t nch = a.n_chains[zi], nraw = a.n_raw[zi], nvalid = a.n_valid[zi];
	const uint8_t* cp = a.cp + a.cbase[zi];
	uint8_t* fcode = a.fcode + a.cbase[zi];
	const bool staged = nch <= kFinishChains;      // uniform
	const uint32_t* ch_node = a.chain_node + a.kbase[zi];
	const uint32_t* ch_off = a.chain_off + a.kbase[zi];
	const uint32_t* ch_clen = a.chain_clen + a.kbase[zi];
	uint32_t* order = a.chain_order + a.kbase[zi];
	uint32_t* dst = a.chain_dst + a.kbase[zi];
	uint32_t* vstart = a.chain_vstart + a.kbase[zi];
	if (staged) {
		for (uint32_t c = tid; c < nch; c += kFinishBlock) { s_node[c] = ch_node[c]; s_clen[c] = ch_clen[c]; s_off[c] = ch_off[c]; }
		__syncthreads();
		ch_node = s_node; ch_off = s_off; ch_clen = s_clen; order = s_ord; dst = s_dst; vstart = s_vst;
	}
	const uint32_t sxe = a.sx + 1;

	// ---- phase 1 (serial over chains): order by start vertex, output offsets, BOC index ----
	if (tid == 0) {
		uint32_t v = 0;
		for (uint32_t c = 0; c < nch; c++) { vstart[c] = v; v += ch_clen[c]; }
		// chains come out in ascending raw start order and remove_initial_branch only
		// moves a start inside its own component: insertion sort on a nearly sorted list
		for (uint32_t c = 0; c < nch; c++) {
			const uint32_t key = ch_node[c];
			uint32_t p = c;
			while (p > 0 && ch_node[order[p - 1]] > key) { order[p] = order[p - 1]; p--; }
			order[p] = c;
		}
		uint32_t d = 0;
		for (uint32_t i = 0; i < nch; i++) { dst[order[i]] = d; d += ch_clen[order[i]]; }

		// write_boc_index (crackcodes.hpp:318-372)
		uint8_t* b = a.boc + a.bbase[zi];
		uint32_t num_y = 0, index_size = a.yw;
		for (uint32_t i = 0; i < nch;) {
			const uint32_t y = ch_node[order[i]] / sxe;
			uint32_t j = i;
			while (j < nch && ch_node[order[j]] / sxe == y) j++;
			index_size += a.yw + (j - i + 1) * a.xw;
			num_y++;
			i = j;
		}
		uint32_t idx = 0;
		put_le_dev(b + idx, index_size, 4); idx += 4;
		put_le_dev(b + idx, num_y, a.yw); idx += a.yw;
		uint32_t last_y = 0;
		for (uint32_t i = 0; i < nch;) {
			const uint32_t y = ch_node[order[i]] / sxe;
			uint32_t j = i;
			while (j < nch && ch_node[order[j]] / sxe == y) j++;
			put_le_dev(b + idx, y - last_y, a.yw); idx += a.yw;
			last_y = y;
			put_le_dev(b + idx, j - i, a.xw); idx += a.xw;
			uint32_t last_x = 0;
			for (uint32_t k = i; k < j; k++) {
				const uint32_t x = ch_node[order[k]] - sxe * y;
				put_le_dev(b + idx, x - last_x, a.xw); idx += a.xw;
				last_x = x;
			}
			i = j;
		}
		a.boc_len[zi] = idx;
	}
	__syncthreads();
	__threadfence_block();

	// ---- phase 2: compaction of tombstones + scatter of each chain to its sorted place ----
	constexpr uint32_t kPer = 16;
	uint32_t carry = 0;
	for (uint32_t tile = 0; tile < nraw; tile += kFinishBlock * kPer) {
		const uint32_t i0 = tile + tid * kPer;
		uint32_t cnt = 0;
		uint8_t c[kPer];
		static_assert(kPer == 16, "one 16-byte load per thread");
		if (i0 + kPer <= nraw) {      // (the slice's codes start 16-byte aligned: crack_pass rounds the capacities)
			const uint4 v = *reinterpret_cast<const uint4*>(cp + i0);
			const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) c[k] = static_cast<uint8_t>(w4[k >> 2] >> (8u * (k & 3u)));
		}
		else {
#pragma unroll
			for (uint32_t k = 0; k < kPer; k++) { const uint32_t i = i0 + k; c[k] = i < nraw ? cp[i] : CODE_TOMB; }
		}
#pragma unroll
		for (uint32_t k = 0; k < kPer; k++) cnt += (c[k] != CODE_TOMB);
		uint32_t v[1] = { cnt }, tot[1];
		block_excl_add<1, kFinishWaves>(v, tot, s_scan);
		uint32_t g = carry + v[0];
		if (cnt) {
			// chain of raw index i0: last chain with ch_off <= i0
			uint32_t lo = 0, hi = nch;
			while (lo + 1 < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (ch_off[mid] <= i0) lo = mid; else hi = mid;
			}
			uint32_t chain = lo;
			uint32_t next_off = (chain + 1 < nch) ? ch_off[chain + 1] : 0xFFFFFFFFu;
			if (i0 + kPer <= next_off) {
				// all sixteen in one chain (a slice has ~100 chains in 128 k codes): the surviving codes go out together — one
				// unaligned 16-byte store when none is a tombstone, else 8 + 4 + 2 + 1 — instead of a store per byte (67 M one-byte
				// requests at C2 ran at the L2s' request rate)
				struct __attribute__((packed)) U128 { uint4 v; };
				struct __attribute__((packed)) U64 { unsigned long long v; };
				struct __attribute__((packed)) U32 { uint32_t v; };
				struct __attribute__((packed)) U16 { uint16_t v; };
				uint8_t* d = fcode + dst[chain] + (g - vstart[chain]);
				unsigned long long w0 = 0, w1 = 0;
				uint32_t m = 0;
#pragma unroll
				for (uint32_t k = 0; k < kPer; k++) {
					if (c[k] != CODE_TOMB) {
						if (m < 8u) w0 |= static_cast<unsigned long long>(c[k]) << (8u * m);
						else w1 |= static_cast<unsigned long long>(c[k]) << (8u * (m - 8u));
						m++;
					}
				}
				if (m == 16u) reinterpret_cast<U128*>(d)->v = make_uint4(static_cast<uint32_t>(w0), static_cast<uint32_t>(w0 >> 32), static_cast<uint32_t>(w1), static_cast<uint32_t>(w1 >> 32));
				else {
					if (m & 8u) { reinterpret_cast<U64*>(d)->v = w0; d += 8; w0 = w1; }
					if (m & 4u) { reinterpret_cast<U32*>(d)->v = static_cast<uint32_t>(w0); d += 4; w0 >>= 32; }
					if (m & 2u) { reinterpret_cast<U16*>(d)->v = static_cast<uint16_t>(w0); d += 2; w0 >>= 16; }
					if (m & 1u) d[0] = static_cast<uint8_t>(w0);
				}
			}
			else {
#pragma unroll
				for (uint32_t k = 0; k < kPer; k++) {
					const uint32_t i = i0 + k;
					while (i >= next_off) { chain++; next_off = (chain + 1 < nch) ? ch_off[chain + 1] : 0xFFFFFFFFu; }
					if (c[k] != CODE_TOMB) {
						fcode[dst[chain] + (g - vstart[chain])] = c[k];
						g++;
					}
				}
			}
		}
		carry += tot[0];
	}
	__syncthreads();
	__threadfence_block();

	// ---- phase 3: whole-slice mod-4 difference code, then 2-bit packing ----
	if (a.markov) {
		uint8_t* dcode = a.dcode + a.cbase[zi];
		for (uint32_t g = tid; g < nvalid; g += kFinishBlock) {
			const uint32_t prev = g ? fcode[g - 1] : 0u;
			dcode[g] = static_cast<uint8_t>((fcode[g] - prev) & 3u);
		}
		if (tid == 0) a.payload_len[zi] = 0;
	}
	else {
		uint8_t* out = a.payload + a.pbase[zi];        // 4-byte aligned
		const uint32_t nbytes = (nvalid + 3) / 4;
		const uint32_t nwords = (nvalid + 15) / 16;
		// 16 codes -> one 32-bit word per thread and step (one 16-byte load, one 4-byte store)
		for (uint32_t w = tid; w < nwords; w += kFinishBlock) {
			const uint32_t g0 = w * 16u;
			uint8_t c[16];
			if (g0 + 16u <= nvalid) {
				const uint4 v = *reinterpret_cast<const uint4*>(fcode + g0);
				const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (uint32_t j = 0; j < 16; j++) c[j] = static_cast<uint8_t>(w4[j >> 2] >> (8u * (j & 3u)));
			}
			else {
#pragma unroll
				for (uint32_t j = 0; j < 16; j++) c[j] = g0 + j < nvalid ? fcode[g0 + j] : 0;
			}
			uint32_t prev = g0 ? fcode[g0 - 1] : 0u;
			uint32_t enc = 0;
#pragma unroll
			for (uint32_t j = 0; j < 16; j++) {
				if (g0 + j < nvalid) enc |= ((c[j] - prev) & 3u) << (2 * j);
				prev = c[j];
			}
			if (4u * w + 4u <= nbytes) reinterpret_cast<uint32_t*>(out)[w] = enc;
			else for (uint32_t k = 4u * w; k < nbytes; k++) out[k] = static_cast<uint8_t>(enc >> (8u * (k - 4u * w)));
		}
		if (tid == 0) a.payload_len[zi] = nbytes;
	}
}

// ------------------------------------------------------------------------------
// markov (src/markov.hpp): context = previous N difference codes, oldest in the
// least-significant base-4 digit; codes before the slice start count as 0.
// ------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t markov_ctx(const uint8_t* d, uint32_t g, int order) {
	uint32_t ctx = 0;
	for (int j = 1; j <= order; j++) {
		const uint32_t v = (g >= static_cast<uint32_t>(j)) ? d[g - j] : 0u;
		ctx |= v << (2 * (order - j));
	}
	return ctx;
}

// gather_statistics (markov.hpp:193-220): stats[ctx][code]++ for every code of every
// slice (the first code of a slice is counted in row 0).  grid = nslices.
// LDS: the workgroup's own histogram (4^order x 4 counters) when it fits (order <= 6),
// flushed once: the global counters see one atomic per non-zero row entry and workgroup
// instead of one per run of equal codes.
__global__ void __launch_bounds__(kBlock) k_markov_hist(
	const uint8_t* __restrict__ dcode, const uint64_t* __restrict__ cbase, const uint32_t* __restrict__ n_valid,
	int order, uint32_t* __restrict__ hist, uint32_t lds_entries
) {
	extern __shared__ uint32_t s_hist[];
	const uint32_t zi = blockIdx.x;
	const uint8_t* d = dcode + cbase[zi];
	const uint32_t n = n_valid[zi];
	const uint32_t entries = 4u << (2 * order);
	const bool local = entries <= lds_entries;
	if (local) {
		for (uint32_t i = threadIdx.x; i < entries; i += kBlock) s_hist[i] = 0;
		__syncthreads();
	}
	uint32_t* target = local ? s_hist : hist;
	constexpr uint32_t kPer = 32;
	// each thread owns 32 consecutive codes and merges equal (row, code) neighbours
	// before touching memory: straight crack runs collapse to one atomic
	for (uint32_t g0 = threadIdx.x * kPer; g0 < n; g0 += kBlock * kPer) {
		uint32_t key = 0xFFFFFFFFu, cnt = 0;
		const uint32_t g1 = g0 + kPer < n ? g0 + kPer : n;
		for (uint32_t g = g0; g < g1; g++) {
			const uint32_t k = markov_ctx(d, g, order) * 4u + d[g];
			if (k == key) cnt++;
			else {
				if (cnt) atomicAdd(target + key, cnt);
				key = k; cnt = 1;
			}
		}
		if (cnt) atomicAdd(target + key, cnt);
	}
	if (local) {
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < entries; i += kBlock) { const uint32_t v = s_hist[i]; if (v) atomicAdd(hist + i, v); }
	}
}

// encode_markov (markov.hpp:422-473): first code raw in 2 bits, then rank codes
// 0 -> `0`, 1 -> `10`, 2 -> `110`, 3 -> `111` (LSB first).  Bit offsets come from
// a block prefix sum of the code lengths; set bits are OR-ed into 32-bit words.
__global__ void __launch_bounds__(kBlock) k_markov_pack(
	const uint8_t* __restrict__ dcode, const uint64_t* __restrict__ cbase, const uint32_t* __restrict__ n_valid,
	int order, const uint8_t* __restrict__ model /* symbol -> rank */,
	const uint64_t* __restrict__ pbase, uint8_t* __restrict__ payload, uint32_t* __restrict__ payload_len
) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.x;
	const uint8_t* d = dcode + cbase[zi];
	const uint32_t n = n_valid[zi];
	uint32_t* words = reinterpret_cast<uint32_t*>(payload + pbase[zi]);
	constexpr uint32_t kPer = 8;
	uint32_t carry = 0;
	for (uint32_t tile = 0; tile < n; tile += kBlock * kPer) {
		const uint32_t g0 = tile + threadIdx.x * kPer;
		uint32_t bits[kPer], lens[kPer], tot_len = 0;
#pragma unroll
		for (uint32_t k = 0; k < kPer; k++) {
			const uint32_t g = g0 + k;
			bits[k] = 0; lens[k] = 0;
			if (g < n) {
				if (g == 0) { bits[k] = d[0]; lens[k] = 2; }
				else {
					const uint32_t r = model[markov_ctx(d, g, order) * 4u + d[g]];
					bits[k] = (r == 0) ? 0u : (r == 1) ? 1u : (r == 2) ? 3u : 7u;
					lens[k] = (r == 0) ? 1u : (r == 1) ? 2u : 3u;
				}
			}
			tot_len += lens[k];
		}
		uint32_t v[1] = { tot_len }, tot[1];
		block_excl_add<1>(v, tot, s_scan);
		uint32_t off = carry + v[0];
#pragma unroll
		for (uint32_t k = 0; k < kPer; k++) {
			if (lens[k] && bits[k]) {
				const uint32_t w = off >> 5, sh = off & 31;
				atomicOr(words + w, bits[k] << sh);
				if (sh + lens[k] > 32) atomicOr(words + w + 1, bits[k] >> (32 - sh));
			}
			off += lens[k];
		}
		carry += tot[0];
	}
	if (threadIdx.x == 0) payload_len[zi] = (carry + 7) / 8;
}

// ------------------------------------------------------------------------------
// flat labels (labels.hpp:56-88): component -> label, read at the first pixel of every
// component.  grid = (ceil(max components of a slice / 256), nslices)
// ------------------------------------------------------------------------------
// component -> label from the component's first pixel (k_run_assign leaves it in comp_pix): one thread per
// component instead of one per run (C2: 1.6 M instead of 26 M)
template <typename LABEL>
__global__ void __launch_bounds__(kBlock) k_mapping_comps(
	const LABEL* __restrict__ labels, RunArrays r, uint64_t sxy,
	const uint64_t* __restrict__ comp_off, uint64_t* __restrict__ mapping
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
	if (c >= r.ncomp[zi]) return;
	mapping[comp_off[zi] + c] = static_cast<uint64_t>(labels[zi * sxy + r.comp_pix[r.rbase[zi] + c]]);
}


// Global component id of every voxel (cc3d.hpp:371-400 numbers components continuously
// across slices) for the pin encoder: one thread per pixel, its run found by a popcount in
// the row's break word, ids read from the run tables.  grid = (ceil(sxy / 256), nslices)
__global__ void __launch_bounds__(kBlock) k_paint_components(
	const uint32_t* __restrict__ planeV, uint32_t row_words, uint64_t plane_words, uint32_t sx, uint64_t sxy,
	const uint32_t* __restrict__ word_base, const uint64_t* __restrict__ rbase, const uint32_t* __restrict__ run_cc,
	const uint64_t* __restrict__ comp_off, uint32_t id_base, uint32_t* __restrict__ out
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t px = blockIdx.x * kBlock + threadIdx.x;
	if (px >= sxy) return;
	const uint32_t y = px / sx, x = px - y * sx;
	const uint32_t w = x >> 5, bit = x & 31u;
	const uint64_t wi = zi * plane_words + static_cast<uint64_t>(y) * row_words + w;
	uint32_t b = planeV[wi];
	if (w == 0) b |= 1u;
	// word_base counts the runs that start before this word; bit i set = a run starts at pixel i
	const uint32_t run = word_base[wi] - 1u + __popc(b & (0xFFFFFFFFu >> (31u - bit)));
	out[zi * sxy + px] = run_cc[rbase[zi] + run] + static_cast<uint32_t>(comp_off[zi]) + id_base;
}

// The same with four pixels (16 bytes of ids) per thread: rows whose length is a multiple of four, where the four
// share a break word.  (4.1 ms -> for 2048 x 2048 x 256: one 4-byte store per lane moved 1 TB/s.)
__global__ void __launch_bounds__(kBlock) k_paint_components4(
	const uint32_t* __restrict__ planeV, uint32_t row_words, uint64_t plane_words, uint32_t sx, uint64_t sxy,
	const uint32_t* __restrict__ word_base, const uint64_t* __restrict__ rbase, const uint32_t* __restrict__ run_cc,
	const uint64_t* __restrict__ comp_off, uint32_t id_base, uint32_t* __restrict__ out
) {
	const uint32_t zi = blockIdx.y;
	const uint32_t px = (blockIdx.x * kBlock + threadIdx.x) * 4u;
	if (px >= sxy) return;
	const uint32_t y = px / sx, x = px - y * sx;
	const uint32_t w = x >> 5, bit = x & 31u;
	const uint64_t wi = zi * plane_words + static_cast<uint64_t>(y) * row_words + w;
	uint32_t b = planeV[wi];
	if (w == 0) b |= 1u;
	const uint32_t* cc = run_cc + rbase[zi];
	const uint32_t add = static_cast<uint32_t>(comp_off[zi]) + id_base;
	uint32_t run = word_base[wi] - 1u + __popc(b & (0xFFFFFFFFu >> (31u - bit)));
	uint4 o;
	o.x = cc[run] + add;
	run += (b >> (bit + 1u)) & 1u; o.y = cc[run] + add;
	run += (b >> (bit + 2u)) & 1u; o.z = cc[run] + add;
	run += (b >> (bit + 3u)) & 1u; o.w = cc[run] + add;
	*reinterpret_cast<uint4*>(out + zi * sxy + px) = o;
}

inline void launch_paint_components(hipStream_t s, const uint32_t* planeV, uint32_t row_words, uint64_t plane_words, int64_t sx, int64_t sy, int64_t sz,
	const uint32_t* word_base, const uint64_t* rbase, const uint32_t* run_cc, const uint64_t* comp_off, uint32_t id_base, uint32_t* out) {
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	const bool four = sx % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && !getenv("CKL_PAINT_COMPONENTS_1");
	const uint64_t threads = four ? sxy / 4 : sxy;
	const dim3 grid(static_cast<uint32_t>((threads + kBlock - 1) / kBlock), static_cast<uint32_t>(sz));
	if (four) hipLaunchKernelGGL(k_paint_components4, grid, dim3(kBlock), 0, s, planeV, row_words, plane_words, static_cast<uint32_t>(sx), sxy, word_base, rbase, run_cc, comp_off, id_base, out);
	else hipLaunchKernelGGL(k_paint_components, grid, dim3(kBlock), 0, s, planeV, row_words, plane_words, static_cast<uint32_t>(sx), sxy, word_base, rbase, run_cc, comp_off, id_base, out);
}

// ------------------------------------------------------------------------------
// label table (labels.hpp:92-152): sorted unique labels and the key of every component.
// The component -> label list is sorted on device (bitonic network over a power-of-two
// padded copy), uniqued on the host in one linear pass, and the keys are found by binary
// search on device and packed at their byte width.
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_bitonic_step(uint64_t* __restrict__ a, uint32_t j, uint32_t k, uint32_t n) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const uint32_t l = i ^ j;
	if (l > i) {
		const uint64_t x = a[i], y = a[l];
		const bool up = (i & k) == 0;
		if ((x > y) == up) { a[i] = y; a[l] = x; }
	}
}
// all steps with j < 2048 of one k-stage inside LDS (2048 elements per workgroup)
// every stage k = 2 .. 2048 of the network inside blocks of 2048 keys: one launch instead of eleven
// two steps of the network (distances j and j / 2) in one launch: a thread takes the four elements that the two
// steps exchange among themselves (the launches of the sort are what it costs: ~6 us each on the label stream)
__global__ void __launch_bounds__(kBlock) k_bitonic_step2(uint64_t* __restrict__ a, uint32_t j, uint32_t k, uint32_t n) {
	const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
	if (t >= n / 4u) return;
	const uint32_t h = j >> 1;
	const uint32_t lb = static_cast<uint32_t>(__ffs(h)) - 1u;
	const uint32_t i = (t & (h - 1u)) | ((t >> lb) << (lb + 2u));      // bits h and j clear
	uint64_t v0 = a[i], v1 = a[i | h], v2 = a[i | j], v3 = a[i | j | h];
	const bool up = (i & k) == 0;
	auto cx = [&](uint64_t& x, uint64_t& y) { if ((x > y) == up) { const uint64_t z = x; x = y; y = z; } };
	cx(v0, v2); cx(v1, v3);
	cx(v0, v1); cx(v2, v3);
	a[i] = v0; a[i | h] = v1; a[i | j] = v2; a[i | j | h] = v3;
}

__global__ void __launch_bounds__(kBlock) k_bitonic_first(uint64_t* __restrict__ a, uint32_t n) {
	__shared__ uint64_t s[2048];
	const uint32_t base = blockIdx.x * 2048u;
	for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) s[t] = base + t < n ? a[base + t] : ~0ull;
	__syncthreads();
	for (uint32_t k = 2; k <= 2048u; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) {
				const uint32_t l = t ^ j;
				if (l > t) {
					const uint64_t x = s[t], y = s[l];
					const bool up = ((base + t) & k) == 0;
					if ((x > y) == up) { s[t] = y; s[l] = x; }
				}
			}
			__syncthreads();
		}
	}
	for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) if (base + t < n) a[base + t] = s[t];
}
__global__ void __launch_bounds__(kBlock) k_bitonic_local(uint64_t* __restrict__ a, uint32_t j_start, uint32_t k, uint32_t n) {
	__shared__ uint64_t s[2048];
	const uint32_t base = blockIdx.x * 2048u;
	for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) s[t] = base + t < n ? a[base + t] : ~0ull;
	__syncthreads();
	for (uint32_t j = j_start; j > 0; j >>= 1) {
		for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) {
			const uint32_t l = t ^ j;
			if (l > t) {
				const uint64_t x = s[t], y = s[l];
				const bool up = ((base + t) & k) == 0;
				if ((x > y) == up) { s[t] = y; s[l] = x; }
			}
		}
		__syncthreads();
	}
	for (uint32_t t = threadIdx.x; t < 2048u; t += kBlock) if (base + t < n) a[base + t] = s[t];
}
__global__ void __launch_bounds__(kBlock) k_pad_copy_u64(const uint64_t* __restrict__ src, uint64_t n, uint64_t* __restrict__ dst, uint64_t n_pad) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	if (i < n_pad) dst[i] = i < n ? src[i] : ~0ull;
}
// sorted[N] -> uniq[U] (first of every run of equal values), U: heads counted per 2048-item
// block, block counts scanned by one workgroup, heads scattered.
constexpr uint32_t kUniqPer = 8, kUniqItems = kBlock * kUniqPer;
__device__ __forceinline__ uint32_t unique_heads(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t i0, uint64_t (&v)[kUniqPer]) {
	uint32_t flag = 0;
#pragma unroll
	for (uint32_t q = 0; q < kUniqPer; q++) {
		const uint32_t i = i0 + q;
		v[q] = i < n ? sorted[i] : 0;
		const uint64_t prev = q ? v[q - 1] : ((i > 0 && i < n) ? sorted[i - 1] : 0);
		flag |= ((i < n && (i == 0 || v[q] != prev)) ? 1u : 0u) << q;
	}
	return flag;
}
// grid = ceil(n / 2048)
__global__ void __launch_bounds__(kBlock) k_unique_count(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ blk_count) {
	__shared__ uint32_t s_red[kWaves];
	uint64_t v[kUniqPer];
	const uint32_t flag = unique_heads(sorted, n, blockIdx.x * kUniqItems + threadIdx.x * kUniqPer, v);
	const uint32_t tot = block_sum(static_cast<uint32_t>(__popc(flag)), s_red);
	if (threadIdx.x == 0) blk_count[blockIdx.x] = tot;
}
// one workgroup: exclusive prefix of the block counts (in place) and the total
__global__ void __launch_bounds__(kBlock) k_unique_scan(uint32_t* __restrict__ blk_count, uint32_t nblk, uint32_t* __restrict__ n_uniq) {
	__shared__ uint32_t s_scan[kWaves];
	uint32_t carry = 0;
	for (uint32_t b0 = 0; b0 < nblk; b0 += kBlock) {
		const uint32_t b = b0 + threadIdx.x;
		uint32_t c[1] = { b < nblk ? blk_count[b] : 0u }, tot[1];
		block_excl_add<1>(c, tot, s_scan);
		if (b < nblk) blk_count[b] = carry + c[0];
		carry += tot[0];
	}
	if (threadIdx.x == 0) *n_uniq = carry;
}
// grid = ceil(n / 2048)
__global__ void __launch_bounds__(kBlock) k_unique_scatter(const uint64_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ blk_base, uint64_t* __restrict__ uniq) {
	__shared__ uint32_t s_scan[kWaves];
	uint64_t v[kUniqPer];
	const uint32_t flag = unique_heads(sorted, n, blockIdx.x * kUniqItems + threadIdx.x * kUniqPer, v);
	uint32_t c[1] = { static_cast<uint32_t>(__popc(flag)) }, tot[1];
	block_excl_add<1>(c, tot, s_scan);
	uint32_t o = blk_base[blockIdx.x] + c[0];
#pragma unroll
	for (uint32_t q = 0; q < kUniqPer; q++) if ((flag >> q) & 1u) uniq[o++] = v[q];
}

// The component labels are first reduced to their distinct values (a volume has far fewer labels
// than components: C2 has 1.6 M components and 65 k labels), so that only those are sorted: an
// open-addressing table keyed by label, then the occupied slots compacted like the unique heads
// above.  The all-ones label doubles as the empty marker and is noted in a flag of its own.
constexpr uint64_t kLabelHashEmpty = ~0ull;
__device__ __forceinline__ uint32_t label_hash(uint64_t v) {
	v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33;
	return static_cast<uint32_t>(v);
}
// grid = ceil(n / 256); table: mask + 1 slots, preset to kLabelHashEmpty
__global__ void __launch_bounds__(kBlock) k_label_hash_insert(const uint64_t* __restrict__ mapping, uint32_t n, unsigned long long* __restrict__ table, uint32_t mask, uint32_t* __restrict__ has_max) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const unsigned long long v = mapping[i];
	if (v == kLabelHashEmpty) { *has_max = 1u; return; }
	uint32_t h = label_hash(v) & mask;
	for (uint32_t probe = 0; probe <= mask; probe++) {
		const unsigned long long seen = table[h];      // most components find their label already there
		if (seen == v) return;
		if (seen == kLabelHashEmpty) {
			const unsigned long long old = atomicCAS(table + h, kLabelHashEmpty, v);
			if (old == kLabelHashEmpty || old == v) return;
		}
		h = (h + 1u) & mask;
	}
}
__device__ __forceinline__ uint32_t hash_occupied(const unsigned long long* __restrict__ table, uint32_t slots, uint32_t i0, uint64_t (&v)[kUniqPer]) {
	uint32_t flag = 0;
#pragma unroll
	for (uint32_t q = 0; q < kUniqPer; q++) {
		v[q] = i0 + q < slots ? table[i0 + q] : kLabelHashEmpty;
		flag |= (v[q] != kLabelHashEmpty ? 1u : 0u) << q;
	}
	return flag;
}
// grid = ceil(slots / 2048)
__global__ void __launch_bounds__(kBlock) k_label_hash_count(const unsigned long long* __restrict__ table, uint32_t slots, uint32_t* __restrict__ blk_count) {
	__shared__ uint32_t s_red[kWaves];
	uint64_t v[kUniqPer];
	const uint32_t flag = hash_occupied(table, slots, blockIdx.x * kUniqItems + threadIdx.x * kUniqPer, v);
	const uint32_t tot = block_sum(static_cast<uint32_t>(__popc(flag)), s_red);
	if (threadIdx.x == 0) blk_count[blockIdx.x] = tot;
}
// grid = ceil(slots / 2048); the all-ones label, if seen, goes behind the others
__global__ void __launch_bounds__(kBlock) k_label_hash_scatter(
	const unsigned long long* __restrict__ table, uint32_t slots, const uint32_t* __restrict__ blk_base, const uint32_t* __restrict__ n_found,
	const uint32_t* __restrict__ has_max, uint64_t* __restrict__ out
) {
	__shared__ uint32_t s_scan[kWaves];
	uint64_t v[kUniqPer];
	const uint32_t flag = hash_occupied(table, slots, blockIdx.x * kUniqItems + threadIdx.x * kUniqPer, v);
	uint32_t c[1] = { static_cast<uint32_t>(__popc(flag)) }, tot[1];
	block_excl_add<1>(c, tot, s_scan);
	uint32_t o = blk_base[blockIdx.x] + c[0];
#pragma unroll
	for (uint32_t q = 0; q < kUniqPer; q++) if ((flag >> q) & 1u) out[o++] = v[q];
	if (blockIdx.x == 0 && threadIdx.x == 0 && *has_max) out[*n_found] = kLabelHashEmpty;
}

// The flat label section (labels.hpp:123-152) assembled on device:
//   u64 num_unique | uniq[num_unique] : stored_width | cc_per_slice[sz] : component_width | key[N] : byte_width(num_unique)
// grid = ceil(max(U, sz, N) / 256) over three index spaces handled by one kernel
__global__ void __launch_bounds__(kBlock) k_flat_section(
	const uint64_t* __restrict__ uniq, const uint32_t* __restrict__ n_uniq, int stored_width,
	const uint32_t* __restrict__ ncomp, uint32_t nslices, int component_width,
	const uint64_t* __restrict__ mapping, uint32_t n, uint8_t* __restrict__ out
) {
	const uint32_t nu = *n_uniq;
	const int key_width = nu <= 0xFFu ? 1 : (nu <= 0xFFFFu ? 2 : 4);
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i == 0) for (int b = 0; b < 8; b++) out[b] = static_cast<uint8_t>((static_cast<uint64_t>(nu) >> (8 * b)) & 0xFF);
	uint8_t* o_uniq = out + 8;
	uint8_t* o_cc = o_uniq + static_cast<uint64_t>(nu) * stored_width;
	uint8_t* o_keys = o_cc + static_cast<uint64_t>(nslices) * component_width;
	if (i < nu) {
		const uint64_t v = uniq[i];
		for (int b = 0; b < stored_width; b++) o_uniq[static_cast<uint64_t>(i) * stored_width + b] = static_cast<uint8_t>((v >> (8 * b)) & 0xFF);
	}
	if (i < nslices) {
		const uint32_t v = ncomp[i];
		for (int b = 0; b < component_width; b++) o_cc[static_cast<uint64_t>(i) * component_width + b] = b < 4 ? static_cast<uint8_t>((v >> (8 * b)) & 0xFF) : 0;
	}
	if (i < n) {
		const uint64_t v = mapping[i];
		uint32_t lo = 0, hi = nu;          // last index with uniq[idx] <= v
		while (lo + 1 < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (uniq[mid] <= v) lo = mid; else hi = mid;
		}
		for (int b = 0; b < key_width; b++) o_keys[static_cast<uint64_t>(i) * key_width + b] = static_cast<uint8_t>((lo >> (8 * b)) & 0xFF);
	}
}

// one workgroup: final offset of every slice's codes (exclusive prefix of BOC + payload bytes) and
// what the host needs of them in one buffer: report = [code_len[nslices] | total lo | total hi | errors]
__global__ void __launch_bounds__(kBlock) k_code_offsets(
	const uint32_t* __restrict__ boc_len, const uint32_t* __restrict__ payload_len, const uint32_t* __restrict__ slice_err,
	uint32_t nslices, uint64_t* __restrict__ out_off, uint32_t* __restrict__ report
) {
	__shared__ unsigned long long s_part[kBlock];
	__shared__ uint32_t s_err;
	if (threadIdx.x == 0) s_err = 0;
	const uint32_t per = (nslices + kBlock - 1) / kBlock;
	const uint32_t z0 = threadIdx.x * per, z1 = min(z0 + per, nslices);
	unsigned long long sum = 0;
	uint32_t err = 0;
	for (uint32_t z = z0; z < z1; z++) { sum += static_cast<unsigned long long>(boc_len[z]) + payload_len[z]; err |= slice_err[z]; }
	s_part[threadIdx.x] = sum;
	__syncthreads();
	if (err) atomicOr(&s_err, err);
	if (threadIdx.x == 0) {
		unsigned long long run = 0;
		for (uint32_t i = 0; i < kBlock; i++) { const unsigned long long v = s_part[i]; s_part[i] = run; run += v; }
		report[nslices] = static_cast<uint32_t>(run); report[nslices + 1] = static_cast<uint32_t>(run >> 32);
	}
	__syncthreads();
	unsigned long long off = s_part[threadIdx.x];
	for (uint32_t z = z0; z < z1; z++) {
		const uint32_t len = boc_len[z] + payload_len[z];
		out_off[z] = off;
		report[z] = len;
		off += len;
	}
	if (threadIdx.x == 0) report[nslices + 2] = s_err;
}

// grid = nslices: copy each slice's BOC index and payload to their final offsets
__global__ void __launch_bounds__(kBlock) k_gather_codes(
	const uint8_t* __restrict__ boc, const uint64_t* __restrict__ bbase, const uint32_t* __restrict__ boc_len,
	const uint8_t* __restrict__ payload, const uint64_t* __restrict__ pbase, const uint32_t* __restrict__ payload_len,
	const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out
) {
	const uint32_t zi = blockIdx.x;
	uint8_t* o = out + out_off[zi];
	const uint32_t nb = boc_len[zi], np = payload_len[zi];
	const uint8_t* b = boc + bbase[zi];
	const uint8_t* p = payload + pbase[zi];
	for (uint32_t i = threadIdx.x; i < nb; i += kBlock) o[i] = b[i];
	for (uint32_t i = threadIdx.x; i < np; i += kBlock) o[nb + i] = p[i];
}

// n bytes from src to dst, any alignment on either side: the device-resident copy of the assembled stream
// (ckl_encoder_keep_device_stream) takes its two bulky sections from buffers of this session.  A device-to-device
// ---- crc32c of a device buffer (the flat label section: header.hpp / crackle.hpp:171-216 store it behind the crack codes) ----
// The register's journey is linear over GF(2): state(c, M) = c * x^bits(M) + state(0, M).  The n bytes are laid right-aligned
// into a frame of G * 256 * P bytes (G a power of two; zeros in front leave a zero register where it is), every thread
// runs its P bytes from a zero register through the byte table, and the pieces are folded pairwise with x^(bits of the
// right half) — inside the workgroups here, over the workgroups in k_crc32c_fold.
struct CrcShifts { uint32_t x[8]; };      // x[k] = x^(8 * piece * 2^k) mod P, reflected representation (ckl_common.hpp: gf_xpow)
__global__ void __launch_bounds__(kBlock) k_crc32c_pieces(const uint8_t* __restrict__ data, uint64_t n, uint64_t pad, uint32_t P, CrcShifts sh, uint32_t* __restrict__ parts) {
	__shared__ uint32_t s_tab[256];
	__shared__ uint32_t s_st[kBlock];
	{
		uint32_t c = threadIdx.x;
		for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ dev::kCrcPoly : (c >> 1);
		s_tab[threadIdx.x] = c;
	}
	__syncthreads();
	const uint64_t f0 = (static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x) * P;      // my piece's place in the frame
	uint32_t c = 0;
	for (uint32_t g = 0; g < P; g += 16) {
		const int64_t m0 = static_cast<int64_t>(f0 + g) - static_cast<int64_t>(pad);           // ... and in the message
		uint32_t w[4] = { 0, 0, 0, 0 };
		if (m0 >= 0 && static_cast<uint64_t>(m0) + 16u <= n) __builtin_memcpy(w, data + m0, 16);      // (unaligned 16-byte load)
		else if (m0 + 16 > 0) for (int b = 0; b < 16; b++) { const int64_t i = m0 + b; if (i >= 0 && static_cast<uint64_t>(i) < n) w[b >> 2] |= static_cast<uint32_t>(data[i]) << (8 * (b & 3)); }
#pragma unroll
		for (int b = 0; b < 16; b++) c = s_tab[(c ^ (w[b >> 2] >> (8 * (b & 3)))) & 0xFFu] ^ (c >> 8);
	}
	s_st[threadIdx.x] = c;
	__syncthreads();
	for (int k = 0; k < 8; k++) {
		const uint32_t stride = 1u << k;
		if ((threadIdx.x & (2u * stride - 1u)) == 0u) s_st[threadIdx.x] = gf_mul(s_st[threadIdx.x], sh.x[k]) ^ s_st[threadIdx.x + stride];
		__syncthreads();
	}
	if (threadIdx.x == 0) parts[blockIdx.x] = s_st[0];
}
// one workgroup: the G (a power of two, <= kBlock) workgroup states -> the finished crc32c (register preset to all ones, inverted at the end)
__global__ void __launch_bounds__(kBlock) k_crc32c_fold(const uint32_t* __restrict__ parts, uint32_t G, CrcShifts sh, uint32_t init_term, uint32_t* __restrict__ out) {
	__shared__ uint32_t s_st[kBlock];
	s_st[threadIdx.x] = threadIdx.x < G ? parts[threadIdx.x] : 0u;
	__syncthreads();
	for (int k = 0; (1u << k) < G; k++) {
		const uint32_t stride = 1u << k;
		if ((threadIdx.x & (2u * stride - 1u)) == 0u && threadIdx.x + stride < G) s_st[threadIdx.x] = gf_mul(s_st[threadIdx.x], sh.x[k]) ^ s_st[threadIdx.x + stride];
		__syncthreads();
	}
	if (threadIdx.x == 0) out[0] = ~(s_st[0] ^ init_term);
}

// hipMemcpyAsync moved the 13 MB of C2's crack codes at 115 GB/s (113 us on the encode's tail); this is a plain
// streaming kernel.  Destination words are written aligned; a source word is two aligned loads and a funnel shift.
// grid-stride, block = kBlock.
__global__ void __launch_bounds__(kBlock) k_copy_bytes(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n) {
	const uint64_t head = min(n, static_cast<uint64_t>((4u - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));      // bytes in front of dst's first aligned word
	const uint64_t words = (n - head) / 4u;
	const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
	if (tid < head) dst[tid] = src[tid];
	const uint64_t tail0 = head + words * 4u;
	if (tid < n - tail0) dst[tail0 + tid] = src[tail0 + tid];
	const uint8_t* s0 = src + head;
	const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s0) & 3u) * 8u;
	const uint32_t* sw = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(s0) & ~static_cast<uintptr_t>(3));
	uint32_t* dw = reinterpret_cast<uint32_t*>(dst + head);
	if (sh == 0u) {
		for (uint64_t i = tid; i < words; i += stride) dw[i] = sw[i];
	}
	else {
		// (the last word's second load reads the aligned word that holds the source's last bytes: inside the buffer)
		for (uint64_t i = tid; i < words; i += stride) dw[i] = __funnelshift_r(sw[i], sw[i + 1], sh);
	}
}

}  // namespace ckl

// ------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------
using namespace ckl;

// crc32c of n > 0 device bytes into out[0], on stream s (parts: scratch of kBlock words)
static void crc32c_device(const uint8_t* data, uint64_t n, uint32_t* parts, uint32_t* out, hipStream_t s) {
	uint32_t P = 128, G = 1;
	while (G < static_cast<uint32_t>(kBlock) && static_cast<uint64_t>(G) * kBlock * P < n) G <<= 1;
	if (static_cast<uint64_t>(G) * kBlock * P < n) P = static_cast<uint32_t>(((n + static_cast<uint64_t>(G) * kBlock - 1) / (static_cast<uint64_t>(G) * kBlock) + 15) / 16 * 16);
	const uint64_t frame = static_cast<uint64_t>(G) * kBlock * P;
	// (the sixteen powers depend on the piece length only — 128 bytes for every section up to 8 MB — and cost the host ~40 us,
	// on the encode's tail: kept from call to call)
	static std::mutex shifts_mutex;
	static uint32_t shifts_for = 0;
	static CrcShifts shifts_in, shifts_over;
	CrcShifts in_wg, over_wg;
	{
		std::lock_guard<std::mutex> lock(shifts_mutex);
		if (shifts_for != P) {
			for (int k = 0; k < 8; k++) {
				shifts_in.x[k] = gf_xpow(8ull * P << k);
				shifts_over.x[k] = gf_xpow(8ull * P * kBlock << k);
			}
			shifts_for = P;
		}
		in_wg = shifts_in; over_wg = shifts_over;
	}
	hipLaunchKernelGGL(k_crc32c_pieces, dim3(G), dim3(kBlock), 0, s, data, n, frame - n, P, in_wg, parts);
	hipLaunchKernelGGL(k_crc32c_fold, dim3(1), dim3(kBlock), 0, s, parts, G, over_wg, gf_mul(0xFFFFFFFFu, gf_xpow(8ull * n)), out);
}

// k_copy_bytes on stream s
static void copy_bytes_device(const uint8_t* src, uint8_t* dst, uint64_t n, hipStream_t s) {
	if (!n) return;
	const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n / 4u + kBlock - 1) / kBlock + 1u, 8192u));
	hipLaunchKernelGGL(k_copy_bytes, dim3(blocks), dim3(kBlock), 0, s, src, dst, n);
}

constexpr uint32_t kTrailStreams = 8;

struct VolumeStats { uint64_t max_label = 0, pairs = 0, first = 0, last = 0; };

struct ckl_encoder {
	int device = 0;
	hipStream_t stream = nullptr;      // crack codes
	hipStream_t stream2 = nullptr;     // labels (components, crcs, label table), concurrent with the crack trail
	hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr, ev_in = nullptr;
	hipEvent_t evd0 = nullptr, evd1 = nullptr;      // around k_trail_walk (first slice group)
	float trail_ms = 0.f;
	hipStream_t trail_stream[kTrailStreams] = {};     // slice groups of the crack trail
	hipEvent_t ev_prezero = nullptr;   // trail_prezero()'s fills on stream2 are done
	hipEvent_t ev_fork = nullptr, ev_join[kTrailStreams] = {}, ev_pre[kTrailStreams] = {};      // ev_pre[g]: group g's kernels in front of its walk are done
	float pipeline_ms = 0.f, dominant_ms = 0.f;
	int64_t max_sx = 0, max_sy = 0, max_sz = 0;
	int dtype_bytes = 0;

	DevBuf<unsigned long long> d_stats;
	DevBuf<uint4> d_adjm;                        // crack graph in micro-tiles (ckl_trail.hpp)
	DevBuf<uint32_t> d_slice_err;
	DevBuf<uint64_t> d_cbase, d_sbase, d_kbase, d_pbase, d_bbase, d_out_off, d_comp_off;
	DevBuf<uint32_t> d_ccap, d_scap, d_kcap;
	DevBuf<uint8_t> d_crack_tables;               // packed block behind the per-slice base / capacity tables of crack_pass
	DevBuf<uint8_t> d_cp, d_fcode, d_dcode, d_payload, d_boc, d_codes_out, d_model;
	DevBuf<uint32_t> d_code_report;
	uint64_t codes_capacity = 0;        // bound of all slices' BOC + payload bytes
	bool defer_codes = false;           // ckl_encoder_defer_codes: the codes stay in d_codes_out for ckl_encoder_codes_to_host
	bool keep_device_stream = false;    // ckl_encoder_keep_device_stream: every run also assembles the whole stream in HBM
	bool async_host_copy = false;       // ckl_encoder_async_host_copy: a run returns when the stream is complete in HBM; its crack codes reach the host buffer on stream_copy
	bool host_copy_pending = false;     // ... until ckl_encoder_host_wait
	hipStream_t stream_copy = nullptr;
	hipStream_t stream_tab = nullptr;   // the pin stage's label lists come to the host beside the passes of the label stream
	hipEvent_t ev_codes = nullptr;
	void* tables_staging = nullptr;              // pinned host image of d_crack_tables (crack_pass: UploadPacker::commit)
	bool uploads_by_kernel = false;              // this run's small uploads go by upload_small's kernel (flat label streams)
	hipEvent_t ev_labels_crc = nullptr;          // the label section's crc32c stands in d_labels_crc (device-resident streams: ckl_encoder_run)
	DevBuf<uint32_t> d_labels_crc;               // [0]: the crc, [1 ..]: the workgroups' states
	DevBuf<uint8_t> d_stream_out;       // ... here (valid until the next run)
	uint64_t device_stream_bytes = 0;
	uint64_t last_codes_total = 0;      // bytes of the last run's crack codes
	DevBuf<uint32_t> d_stack_node, d_stack_code;
	DevBuf<uint32_t> d_chain_node, d_chain_off, d_chain_clen, d_chain_order, d_chain_dst, d_chain_vstart;
	DevBuf<uint32_t> d_n_chains, d_n_raw, d_n_valid, d_payload_len, d_boc_len;
	DevBuf<uint32_t> d_hist;
	// label planes + run-based CCL (ckl_runs.hpp)
	DevBuf<uint32_t> d_planes, d_count_vh;
	uint32_t row_words = 0;
	uint64_t plane_words = 0;
	std::vector<uint32_t> count_v, count_h;     // differing neighbour pairs per slice (host copy)
	DevBuf<uint64_t> d_rbase;
	DevBuf<uint32_t> d_rcap, d_word_base, d_parent, d_run_start, d_run_cc, d_comp_pix, d_nruns, d_ncomp, d_idbits, d_blk_roots;
	DevBuf<uint16_t> d_run_local;
	DevBuf<uint32_t> d_G, d_crc_acc;
	DevBuf<uint32_t> d_flat_report;     // [4][nslices]: ncomp | crc_acc | idbits | slice_err2 (views above)
	uint64_t g_table_pixels = 0;                // slice size the G table was built for
	DevBuf<uint64_t> d_mapping, d_sorted, d_uniq, d_label_hash, d_label_list;
	DevBuf<uint8_t> d_keys;
	DevBuf<uint32_t> d_cc_volume;                // global component id of every voxel (pin encoding only)
	DevBuf<uint32_t> d_pin_kept, d_pin_u32;      // pin passes (ckl_pins_dev.hpp): kept-run marks (uint16 per voxel), per-component depths
	DevBuf<uint8_t> d_pin_tables;                // per-row label tables of k_pin_dedup
	DevBuf<uint64_t> d_pin_u64;                  // per-component keys
	DevBuf<uint32_t> d_slice_err2, d_n_uniq, d_uniq_blk;
	DevBuf<uint8_t> d_labels_bin;                // the flat label section, assembled on device
	uint32_t flat_max_rcap = 0;
	// label planes left by ckl_encoder_stats for the ckl_encoder_run that follows on the same
	// volume (the sharded encoder: stats -> all-gather -> run with the agreed formats)
	const void* planes_for = nullptr;
	// ckl_encoder_markov_stats leaves the whole trail behind (difference codes, chains, BOC index): the run that
	// follows for the same volume and crack format only packs them under the agreed model (single use, like the planes)
	const void* trail_for = nullptr;
	bool trail_perm = false;
	int trail_order = 0;
	VolumeStats planes_stats;          // max / pairs of the volume the cached planes were built from
	int64_t planes_dims[3] = { 0, 0, 0 };
	std::vector<uint64_t> h_rbase;
	std::vector<uint32_t> h_rcap;
	// trail graph (ckl_trail.hpp)
	std::vector<uint32_t> count_special, count_corner;
	DevBuf<uint32_t> d_plane_partial, t_blk_special, t_blk_corner;
	DevBuf<unsigned long long> d_plane_partial_max, d_plane_out;
	uint32_t graph_blocks = 0;
	uint32_t tiles_x = 0, tiles_y = 0, mtx2 = 0;
	uint64_t adjm_stride = 0;
	bool graph_permissible = false;
	bool planes_deferred = false;      // planes_pass left its counts on the device: graph_pass (or planes_collect) fetches them
	bool trail_zeroed = false;         // trail_prezero() ran on the stream since the last trail: crack_pass skips its fills
	DevBuf<uint64_t> t_nbase, t_cobase, t_ibase;
	DevBuf<uint32_t> t_ncap, t_cocap, t_icap, t_max_steps;
	size_t last_trail_slices = 0;                // slices of the last crack pass (the layout of t_counters)
	DevBuf<uint32_t> t_counters;                 // n_nodes | n_snap | n_corners | n_starts | n_items | n_events | seg_len_sum, [nslices] each
	DevBuf<uint32_t> t_node_vertex, t_vert2node, t_corner_vertex;
	DevBuf<uint8_t> t_node_adj;
	DevBuf<uint32_t> t_dart_end, t_dart_len, t_dart_minv, t_dart_minpos, t_parent, t_start_bits, t_starts;
	DevBuf<uint4> t_dart_codes;
	DevBuf<uint8_t> t_dart_inline;
	DevBuf<unsigned long long> t_compmin;
	DevBuf<uint32_t> t_items, t_item_off, t_chain_item0, t_events, t_chain_ev0, t_ev_lnd, t_ev_item;

	~ckl_encoder() {
		if (ev0) (void)hipEventDestroy(ev0);
		if (ev1) (void)hipEventDestroy(ev1);
		if (evk0) (void)hipEventDestroy(evk0);
		if (evk1) (void)hipEventDestroy(evk1);
		if (evd0) (void)hipEventDestroy(evd0);
		if (evd1) (void)hipEventDestroy(evd1);
		if (ev_in) (void)hipEventDestroy(ev_in);
		if (ev_fork) (void)hipEventDestroy(ev_fork);
		if (ev_prezero) (void)hipEventDestroy(ev_prezero);
		for (auto& ev : ev_join) if (ev) (void)hipEventDestroy(ev);
		for (auto& ev : ev_pre) if (ev) (void)hipEventDestroy(ev);
		for (auto& st : trail_stream) if (st) (void)hipStreamDestroy(st);
		if (stream) (void)hipStreamDestroy(stream);
		if (stream2) (void)hipStreamDestroy(stream2);
		if (stream_copy) { (void)hipStreamSynchronize(stream_copy); (void)hipStreamDestroy(stream_copy); }
		if (stream_tab) { (void)hipStreamSynchronize(stream_tab); (void)hipStreamDestroy(stream_tab); }
		if (ev_codes) (void)hipEventDestroy(ev_codes);
		if (tables_staging) host_out_free(tables_staging);
		if (ev_labels_crc) (void)hipEventDestroy(ev_labels_crc);
	}
};

namespace {

template <typename T>
void upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t s) {
	d.ensure(h.size());
	if (!h.empty()) CKL_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}
// Many small tables in ONE copy: the tables become views of a packed block (a copy of a few bytes costs
// ~8 us of stream time, and crack_pass needs fifteen of them between two kernels of the trail).
struct UploadPacker {
	struct Item { void* buf; size_t off, count; void (*bind)(void*, uint8_t*, size_t); };
	std::vector<Item> items;
	std::vector<uint8_t> image;
	template <typename T>
	void add(DevBuf<T>& dst, const std::vector<T>& src) {
		const size_t off = (image.size() + 255) & ~static_cast<size_t>(255);
		image.resize(off + std::max<size_t>(src.size(), 1) * sizeof(T));
		if (!src.empty()) memcpy(image.data() + off, src.data(), src.size() * sizeof(T));
		items.push_back({ &dst, off, src.size(), [](void* b, uint8_t* base, size_t n) { static_cast<DevBuf<T>*>(b)->borrow(reinterpret_cast<T*>(base), n); } });
	}
	// `block` must not be in use by kernels of another stream; the copy is ordered on `s`
	// staging: the session's pinned host block for this image (kept until its next commit: the copy kernel reads it), or null
	void commit(DevBuf<uint8_t>& block, hipStream_t s, void** staging = nullptr) {
		if (image.empty()) return;
		for (const Item& it : items) it.bind(it.buf, nullptr, 0);        // views of the old block go first: ensure() may free it
		block.ensure(image.size());
		if (staging) {
			if (*staging) host_out_free(*staging);
			*staging = host_out_alloc(std::max<size_t>(image.size(), 64u << 10));      // >= 64 KiB: pinned
			memcpy(*staging, image.data(), image.size());
			upload_small(block.p, *staging, image.size(), s, *staging);
		}
		else CKL_HIP(hipMemcpyAsync(block.p, image.data(), image.size(), hipMemcpyHostToDevice, s));      // pageable source: staged before the call returns
		for (const Item& it : items) it.bind(it.buf, block.p + it.off, it.count);
	}
};

template <typename T>
std::vector<T> download(const T* p, size_t n, hipStream_t s) {
	std::vector<T> h(n);
	if (n) CKL_HIP(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
	CKL_HIP(hipStreamSynchronize(s));
	return h;
}


template <typename LABEL>
VolumeStats volume_stats(ckl_encoder& e, const LABEL* labels, uint64_t voxels) {
	VolumeStats st;
	if (voxels == 0) return st;
	hipStream_t s = e.stream;
	e.d_stats.ensure(2);
	CKL_HIP(hipMemsetAsync(e.d_stats.p, 0, 2 * sizeof(unsigned long long), s));
	const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((voxels + kBlock - 1) / kBlock, 256ull * 8));
	hipLaunchKernelGGL(k_stats<LABEL>, dim3(blocks), dim3(kBlock), 0, s, labels, voxels, e.d_stats.p);
	auto r = download(e.d_stats.p, 2, s);
	st.max_label = r[0]; st.pairs = r[1];
	LABEL f, l;
	CKL_HIP(hipMemcpy(&f, labels, sizeof(LABEL), hipMemcpyDeviceToHost));
	CKL_HIP(hipMemcpy(&l, labels + (voxels - 1), sizeof(LABEL), hipMemcpyDeviceToHost));
	st.first = f; st.last = l;
	return st;
}

// wall-clock breakdown of the host side, printed when CKL_PROFILE is set
struct HostTimer {
	bool on;
	std::chrono::steady_clock::time_point t0;
	std::vector<std::pair<const char*, double>> marks;
	HostTimer() : on(getenv("CKL_PROFILE") != nullptr), t0(std::chrono::steady_clock::now()) {}
	void mark(const char* name) {
		if (!on) return;
		auto t1 = std::chrono::steady_clock::now();
		marks.emplace_back(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
		t0 = t1;
	}
	~HostTimer();
};
thread_local HostTimer* g_ht = nullptr;
#define HT_MARK(name) do { if (g_ht) g_ht->mark(name); } while (0)
inline HostTimer::~HostTimer() {
		g_ht = nullptr;
		if (!on) return;
		fprintf(stderr, "[ckl encode host ms]");
		for (auto& m : marks) fprintf(stderr, " %s=%.2f", m.first, m.second);
		fprintf(stderr, "\n");
}

// One pass over the labels -> the two "differs from neighbour" bit planes and their
// per-slice population counts (exact crack edge and run counts follow from these).
template <typename LABEL>
VolumeStats volume_stats(ckl_encoder& e, const LABEL* labels, uint64_t voxels);

// One pass over the labels -> the two "differs from neighbour" bit planes and their
// per-slice population counts (exact crack edge and run counts follow from these) and,
// on the fast path, the whole-volume reductions of lib.hpp:224-256 from the same read.
// the fast planes kernel's per-slice counts -> e.count_v / count_h and the volume's statistics
void planes_collect(ckl_encoder& e, uint32_t ns, VolumeStats* st, const std::vector<unsigned long long>* fetched = nullptr) {
	std::vector<unsigned long long> mine;
	if (!fetched) { mine = download(e.d_plane_out.p, 4ull * ns, e.stream); fetched = &mine; }
	const std::vector<unsigned long long>& o = *fetched;
	e.count_v.resize(ns); e.count_h.resize(ns);
	VolumeStats v;
	for (uint32_t zi = 0; zi < ns; zi++) {
		e.count_v[zi] = static_cast<uint32_t>(o[4ull * zi]);
		e.count_h[zi] = static_cast<uint32_t>(o[4ull * zi + 1]);
		v.pairs += o[4ull * zi + 2];
		v.max_label = std::max<uint64_t>(v.max_label, o[4ull * zi + 3]);
	}
	e.planes_deferred = false;
	if (st) *st = v;
}

// The trail's zero fills (slice errors, counters, chain start bits) depend on the dimensions only.  On the
// label stream, which is idle until the walk, they run beside the planes kernel instead of between the node
// counts and k_trail_nodes (3 fills, 25 us + their launch gaps at C2).  crack_pass waits for ev_prezero.
void trail_prezero(ckl_encoder& e, int64_t sx, int64_t sy, int64_t sz) {
	hipStream_t s = e.stream2;
	const uint32_t ns = static_cast<uint32_t>(sz);
	const uint64_t nverts = static_cast<uint64_t>(sx + 1) * (sy + 1);
	const uint32_t start_words = static_cast<uint32_t>((nverts + 31) / 32);
	e.d_slice_err.ensure(ns);
	e.t_counters.ensure(7 * static_cast<size_t>(ns));
	e.t_start_bits.ensure(static_cast<size_t>(start_words) * ns);
	CKL_HIP(hipMemsetAsync(e.d_slice_err.p, 0, ns * sizeof(uint32_t), s));
	CKL_HIP(hipMemsetAsync(e.t_counters.p, 0, 7 * static_cast<size_t>(ns) * sizeof(uint32_t), s));
	CKL_HIP(hipMemsetAsync(e.t_start_bits.p, 0, static_cast<size_t>(start_words) * ns * sizeof(uint32_t), s));
	CKL_HIP(hipEventRecord(e.ev_prezero, s));
	e.trail_zeroed = true;
}

template <typename LABEL>
void planes_pass(ckl_encoder& e, const LABEL* labels, int64_t sx, int64_t sy, int64_t sz, VolumeStats* st, bool defer = false) {
	hipStream_t s = e.stream;
	const uint32_t ns = static_cast<uint32_t>(sz);
	e.trail_zeroed = false;
	e.planes_deferred = false;
	e.row_words = static_cast<uint32_t>((sx + 31) / 32);
	e.plane_words = static_cast<uint64_t>(e.row_words) * sy;
	e.d_planes.ensure(2 * e.plane_words * ns);
	constexpr uint32_t P = 16 / sizeof(LABEL);
	const bool fast = !getenv("CKL_PLANES_GENERIC") && sx >= static_cast<int64_t>(P) && sx % P == 0 && (reinterpret_cast<uintptr_t>(labels) & 15u) == 0;
	if (fast) {
		const uint32_t strips = static_cast<uint32_t>((sx + 64 * P - 1) / (64 * P));
		const uint32_t bands = static_cast<uint32_t>((sy + kBandRows - 1) / kBandRows);
		const uint32_t nblk = (strips * bands + kWaves - 1) / kWaves;
		e.d_plane_partial.ensure(4ull * nblk * ns);
		e.d_plane_partial_max.ensure(static_cast<size_t>(nblk) * ns);
		e.d_plane_out.ensure(4ull * ns + 1);
		if (!getenv("CKL_NO_PREZERO")) trail_prezero(e, sx, sy, sz);
		if (!getenv("CKL_PLANES_V1")) hipLaunchKernelGGL(k_label_planes_stream<LABEL>, dim3(nblk, ns), dim3(kBlock), 0, s,
			labels, static_cast<uint32_t>(sx), static_cast<uint32_t>(sy), strips, bands,
			e.d_planes.p, e.d_planes.p + e.plane_words * ns, e.row_words, e.plane_words,
			e.d_plane_partial.p, e.d_plane_partial_max.p, e.d_plane_out.p + 4ull * ns);
		else hipLaunchKernelGGL(k_label_planes_fast<LABEL>, dim3(nblk, ns), dim3(kBlock), 0, s,
			labels, static_cast<uint32_t>(sx), static_cast<uint32_t>(sy), strips, bands,
			e.d_planes.p, e.d_planes.p + e.plane_words * ns, e.row_words, e.plane_words,
			e.d_plane_partial.p, e.d_plane_partial_max.p, e.d_plane_out.p + 4ull * ns);
		hipLaunchKernelGGL(k_planes_reduce, dim3(ns), dim3(kBlock), 0, s, e.d_plane_partial.p, e.d_plane_partial_max.p, nblk, e.d_plane_out.p, e.d_plane_out.p + 4ull * ns);
		// deferred: graph_pass fetches the counts together with its own (k_trail_graph decides the crack
		// format from the device's pair count: one host round trip less in front of it)
		e.planes_deferred = defer && !getenv("CKL_NO_DEFER");
		if (!e.planes_deferred) planes_collect(e, ns, st);
		return;
	}
	if (st) *st = volume_stats<LABEL>(e, labels, static_cast<uint64_t>(sx) * sy * sz);
	e.d_count_vh.ensure(2 * static_cast<size_t>(ns));
	CKL_HIP(hipMemsetAsync(e.d_count_vh.p, 0, 2 * static_cast<size_t>(ns) * sizeof(uint32_t), s));
	const uint32_t chunks = static_cast<uint32_t>((sx + 63) / 64);
	const uint64_t units = static_cast<uint64_t>(chunks) * sy;
	hipLaunchKernelGGL(k_label_planes<LABEL>, dim3(static_cast<uint32_t>((units + kWaves * kPlaneUnroll - 1) / (kWaves * kPlaneUnroll)), ns), dim3(kBlock), 0, s,
		labels, static_cast<uint32_t>(sx), static_cast<uint32_t>(sy), chunks,
		e.d_planes.p, e.d_planes.p + e.plane_words * ns, e.row_words, e.plane_words,
		e.d_count_vh.p, e.d_count_vh.p + ns);
	std::vector<uint32_t> c = download(e.d_count_vh.p, 2 * static_cast<size_t>(ns), s);
	e.count_v.assign(c.begin(), c.begin() + ns);
	e.count_h.assign(c.begin() + ns, c.end());
}

// crack graph (vertex nibbles in tiles, crackcodes.hpp:66-125) + the node / corner counts
// of the trail graph (ckl_trail.hpp)
// perm_mode: 0 impermissible, 1 permissible, 2 decided on the device from the planes kernel's pair count
// (crackle.hpp:50-55; needs planes_deferred); `st` receives the deferred statistics
void graph_pass(ckl_encoder& e, int64_t sx, int64_t sy, int64_t sz, int perm_mode, VolumeStats* st = nullptr) {
	hipStream_t s = e.stream;
	const uint32_t ns = static_cast<uint32_t>(sz);
	e.tiles_x = static_cast<uint32_t>((sx + 1 + kTrailTileDim - 1) / kTrailTileDim);
	e.tiles_y = static_cast<uint32_t>((sy + 1 + 7) / 8);      // rows of 32 x 8 pieces
	// micro-tiles of 8 x 8 vertices, 32 bytes (2 uint4) each, 2 x 2 of them per 128-byte line
	const uint32_t mtx = static_cast<uint32_t>((sx + 1 + 7) / 8), mty = static_cast<uint32_t>((sy + 1 + 7) / 8);
	e.mtx2 = (mtx + 1) / 2;
	e.adjm_stride = static_cast<uint64_t>(e.mtx2) * ((mty + 1) / 2) * 4 * 2;      // uint4 per slice
	e.d_adjm.ensure(e.adjm_stride * ns);
	e.graph_blocks = (e.tiles_x * e.tiles_y + kBlock - 1) / kBlock;
	e.t_blk_special.ensure(static_cast<size_t>(e.graph_blocks) * ns);
	e.t_blk_corner.ensure(static_cast<size_t>(e.graph_blocks) * ns);
	e.d_count_vh.ensure(4 * static_cast<size_t>(ns));
	if (perm_mode == 2 && !e.planes_deferred) throw Error(CKL_ERR_RUNTIME, "crackle_amd: internal: crack format left to the device without deferred counts");
	const unsigned long long half_voxels = static_cast<unsigned long long>(static_cast<int64_t>(static_cast<uint64_t>(sx) * sy * sz) / 2);
	hipLaunchKernelGGL(k_trail_graph, dim3(e.graph_blocks, ns), dim3(kBlock), 0, s,
		e.d_planes.p, e.d_planes.p + e.plane_words * ns, e.row_words, e.plane_words,
		static_cast<uint32_t>(sx), static_cast<uint32_t>(sy), static_cast<uint32_t>(perm_mode),
		perm_mode == 2 ? e.d_plane_out.p + 4ull * ns : nullptr, half_voxels,
		reinterpret_cast<uint32_t*>(e.d_adjm.p), e.adjm_stride * 4,
		e.mtx2, e.tiles_x, e.tiles_y, e.t_blk_special.p, e.t_blk_corner.p);
	hipLaunchKernelGGL(k_trail_count_scan, dim3(ns), dim3(kBlock), 0, s, e.t_blk_special.p, e.t_blk_corner.p, e.graph_blocks,
		e.d_count_vh.p + 2 * static_cast<size_t>(ns), e.d_count_vh.p + 3 * static_cast<size_t>(ns));
	std::vector<unsigned long long> po;
	if (e.planes_deferred) {
		po.resize(4ull * ns);
		CKL_HIP(hipMemcpyAsync(po.data(), e.d_plane_out.p, po.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	}
	std::vector<uint32_t> c = download(e.d_count_vh.p + 2 * static_cast<size_t>(ns), 2 * static_cast<size_t>(ns), s);
	e.count_special.assign(c.begin(), c.begin() + ns);
	e.count_corner.assign(c.begin() + ns, c.end());
	VolumeStats v;
	if (!po.empty()) planes_collect(e, ns, &v, &po);
	if (st && !po.empty()) *st = v;
	e.graph_permissible = perm_mode == 2 ? static_cast<int64_t>(v.pairs) < static_cast<int64_t>(static_cast<uint64_t>(sx) * sy * sz) / 2 : perm_mode == 1;
}

struct CrackResult {
	std::vector<uint32_t> code_len;     // per slice: boc + payload bytes
	uint64_t total = 0;                 // bytes of all crack codes; they stay in e.d_codes_out
	bool any_chain = false;
};

// Dynamic LDS of k_trail_walk for slices of up to `max_special` nodes (+ what k_trail_loops / _components add)
inline size_t trail_walk_lds(uint32_t max_special) { return (static_cast<size_t>(max_special) + 256) * 16 + 1024; }

// Runs graph + walk + finish.  When `hist_only` is set, stops after the markov
// histogram (returned in hist).  `model` (symbol -> rank) is required for markov packing.
void crack_pass(
	ckl_encoder& e, int64_t sx, int64_t sy, int64_t sz, bool permissible,
	int markov_order, bool hist_only, const std::vector<uint8_t>* model_in,
	std::vector<uint32_t>* hist_out, std::vector<uint8_t>* model_out, CrackResult* result,
	const std::function<void()>& overlap = std::function<void()>(), bool reuse_trail = false
) {
	hipStream_t s = e.stream;
	const uint32_t ns = static_cast<uint32_t>(sz);
	const uint64_t nverts = static_cast<uint64_t>(sx + 1) * (sy + 1);
	e.d_slice_err.ensure(ns);
	const bool prezeroed = e.trail_zeroed && !reuse_trail;
	e.trail_zeroed = false;
	if (prezeroed) CKL_HIP(hipStreamWaitEvent(s, e.ev_prezero, 0));
	if (!reuse_trail && !prezeroed) CKL_HIP(hipMemsetAsync(e.d_slice_err.p, 0, ns * sizeof(uint32_t), s));
	// exact crack edge count per slice: interior pixel pairs that differ (or are equal)
	const uint64_t interior = static_cast<uint64_t>(sx > 0 ? sx - 1 : 0) * sy + static_cast<uint64_t>(sx) * (sy > 0 ? sy - 1 : 0);

	// capacities from the exact edge counts (codes <= 7 E, chains <= E, stack <= E)
	std::vector<uint64_t> cbase(ns), sbase(ns), kbase(ns);
	std::vector<uint32_t> ccap(ns), scap(ns), kcap(ns), max_steps(ns);
	std::vector<uint64_t> nbase(ns), cobase(ns), ibase(ns);
	std::vector<uint32_t> ncap(ns), cocap(ns), icap(ns);
	uint64_t ctot = 0, stot = 0, ktot = 0, ntot = 0, cotot = 0, itot = 0;
	uint32_t max_ncap = 0, max_cocap = 0, max_icap = 0, max_special = 0;
	bool any = false;
	for (uint32_t zi = 0; zi < ns; zi++) {
		const uint64_t differ = static_cast<uint64_t>(e.count_v[zi]) + e.count_h[zi];
		const uint64_t E = permissible ? interior - differ : differ;
		any = any || E > 0;
		const uint64_t cc = ((7 * E + 16 + 15) / 16) * 16;      // a multiple of 16: every slice's code arrays start 16-byte aligned (k_finish loads 16 codes at once)
		if (cc > 0xFFFFFFF0ull) throw Error(CKL_ERR_RUNTIME, "crackle_amd: slice has too many crack edges");
		cbase[zi] = ctot; ccap[zi] = static_cast<uint32_t>(cc); ctot += cc;
		max_steps[zi] = static_cast<uint32_t>(E + 1);
		// trail graph: nodes = vertices of degree 1, 3, 4 plus at most one per "right+down" corner
		// (loop starts, chain starts inside a segment); segments <= 2 nodes; items <= 3 segments + 1 per chain
		const uint64_t nc = ((static_cast<uint64_t>(e.count_special[zi]) + e.count_corner[zi] + 16) / 16) * 16;
		if (nc >= (1ull << 28)) throw Error(CKL_ERR_RUNTIME, "crackle_amd: slice has too many crack vertices");
		nbase[zi] = ntot; ncap[zi] = static_cast<uint32_t>(nc); ntot += nc;
		cobase[zi] = cotot; cocap[zi] = e.count_corner[zi]; cotot += e.count_corner[zi];
		const uint64_t ic = 7 * nc + 16;
		ibase[zi] = itot; icap[zi] = static_cast<uint32_t>(ic); itot += ic;
		sbase[zi] = stot; scap[zi] = static_cast<uint32_t>(2 * nc + 4); stot += 2 * nc + 4;
		kbase[zi] = ktot; kcap[zi] = static_cast<uint32_t>(nc); ktot += nc;
		max_ncap = std::max<uint32_t>(max_ncap, ncap[zi]);
		max_cocap = std::max<uint32_t>(max_cocap, cocap[zi]);
		max_icap = std::max<uint32_t>(max_icap, icap[zi]);
		max_special = std::max<uint32_t>(max_special, e.count_special[zi]);
	}
	if (result) result->any_chain = any;
	if (g_ht && g_ht->on) {
		uint32_t over = 0;
		for (uint32_t zi = 0; zi < ns; zi++) over += e.count_special[zi] > 3584u;
		fprintf(stderr, "[ckl trail nodes] max special %u, slices above 3584: %u of %u\n", max_special, over, ns);
	}
	HT_MARK("c:caps");
	UploadPacker tables;
	tables.add(e.d_cbase, cbase); tables.add(e.d_ccap, ccap);
	tables.add(e.d_sbase, sbase); tables.add(e.d_scap, scap);
	tables.add(e.d_kbase, kbase); tables.add(e.d_kcap, kcap);
	e.d_cp.ensure(ctot); e.d_fcode.ensure(ctot);
	if (markov_order) e.d_dcode.ensure(ctot);
	e.d_stack_node.ensure(stot); e.d_stack_code.ensure(stot);
	e.d_chain_node.ensure(ktot); e.d_chain_off.ensure(ktot); e.d_chain_clen.ensure(ktot);
	e.d_chain_order.ensure(ktot); e.d_chain_dst.ensure(ktot); e.d_chain_vstart.ensure(ktot);
	e.d_n_chains.ensure(ns); e.d_n_raw.ensure(ns); e.d_n_valid.ensure(ns);
	e.d_payload_len.ensure(ns); e.d_boc_len.ensure(ns);

	// output buffers sized from the capacities (no round trip to the host before k_finish)
	const int xw = byte_width(static_cast<uint64_t>(sx) + 1), yw = byte_width(static_cast<uint64_t>(sy) + 1);
	std::vector<uint64_t> pbase(ns), bbase(ns);
	uint64_t ptot = 0, btot = 0;
	for (uint32_t zi = 0; zi < ns; zi++) {
		// codes <= E + 2 (b + t) <= 5 E + 2; plain: 2 bits / code; markov: at most 3 bits / code (+2)
		const uint64_t ncodes = ccap[zi];
		const uint64_t pbytes = markov_order ? (3ull * ncodes + 2 + 7) / 8 : (ncodes + 3) / 4;
		pbase[zi] = ptot; ptot += ((pbytes + 8 + 3) / 4) * 4;
		const uint64_t bbytes = 4 + yw + static_cast<uint64_t>(kcap[zi]) * (yw + 2 * xw);
		bbase[zi] = btot; btot += bbytes;
	}
	tables.add(e.d_pbase, pbase); tables.add(e.d_bbase, bbase);
	tables.add(e.t_nbase, nbase); tables.add(e.t_ncap, ncap);
	tables.add(e.t_cobase, cobase); tables.add(e.t_cocap, cocap);
	tables.add(e.t_ibase, ibase); tables.add(e.t_icap, icap);
	tables.add(e.t_max_steps, max_steps);
	tables.commit(e.d_crack_tables, s, e.uploads_by_kernel ? &e.tables_staging : nullptr);
	HT_MARK("c:tables");
	e.d_payload.ensure(ptot + 8); e.d_boc.ensure(btot + 8);
	e.codes_capacity = ptot + btot;
	if (markov_order) CKL_HIP(hipMemsetAsync(e.d_payload.p, 0, ptot + 8, s));

	FinishArgs fa;
	fa.sx = static_cast<int>(sx); fa.sy = static_cast<int>(sy); fa.xw = xw; fa.yw = yw;
	fa.markov = markov_order ? 1u : 0u;
	fa.cbase = e.d_cbase.p; fa.kbase = e.d_kbase.p;
	fa.n_chains = e.d_n_chains.p; fa.n_raw = e.d_n_raw.p; fa.n_valid = e.d_n_valid.p;
	fa.cp = e.d_cp.p; fa.chain_node = e.d_chain_node.p; fa.chain_off = e.d_chain_off.p; fa.chain_clen = e.d_chain_clen.p;
	fa.chain_order = e.d_chain_order.p; fa.chain_dst = e.d_chain_dst.p; fa.chain_vstart = e.d_chain_vstart.p;
	fa.fcode = e.d_fcode.p; fa.dcode = e.d_dcode.p;
	fa.pbase = e.d_pbase.p; fa.payload = e.d_payload.p; fa.bbase = e.d_bbase.p; fa.boc = e.d_boc.p;
	fa.payload_len = e.d_payload_len.p; fa.boc_len = e.d_boc_len.p;

	unsigned long long* ta_dbg = nullptr;
	size_t trail_lds_used = 0;
	DevBuf<unsigned long long> d_tdbg;
	CKL_HIP(hipEventRecord(e.evk0, s));
	if (!reuse_trail) {
		// ---- the trail over the node graph (ckl_trail.hpp)
		e.t_counters.ensure(7 * static_cast<size_t>(ns));
		if (!prezeroed) CKL_HIP(hipMemsetAsync(e.t_counters.p, 0, 7 * static_cast<size_t>(ns) * sizeof(uint32_t), s));
		e.t_node_vertex.ensure(ntot); e.t_node_adj.ensure(ntot + 16); e.t_vert2node.ensure(nverts * ns);
		e.t_corner_vertex.ensure(cotot);
		e.t_dart_end.ensure(4 * ntot); e.t_dart_len.ensure(4 * ntot); e.t_dart_minv.ensure(4 * ntot); e.t_dart_minpos.ensure(4 * ntot);
		e.t_dart_codes.ensure(4 * ntot); e.t_dart_inline.ensure(4 * ntot);
		e.t_parent.ensure(ntot); e.t_compmin.ensure(ntot); e.t_starts.ensure(ntot);
		const uint32_t start_words = static_cast<uint32_t>((nverts + 31) / 32);
		e.t_start_bits.ensure(static_cast<size_t>(start_words) * ns);
		if (!prezeroed) CKL_HIP(hipMemsetAsync(e.t_start_bits.p, 0, static_cast<size_t>(start_words) * ns * sizeof(uint32_t), s));
		e.t_items.ensure(itot); e.t_item_off.ensure(itot); e.t_chain_item0.ensure(ktot);
		e.t_events.ensure(itot); e.t_chain_ev0.ensure(ktot); e.t_ev_lnd.ensure(itot); e.t_ev_item.ensure(itot);

		TrailArgs ta;
		ta.adjm = e.d_adjm.p; ta.adjm_stride = e.adjm_stride; ta.mtx2 = e.mtx2; ta.tiles_x = e.tiles_x; ta.tiles_y = e.tiles_y;
		ta.planeV = e.d_planes.p; ta.planeH = e.d_planes.p + e.plane_words * ns; ta.row_words = e.row_words; ta.plane_words = e.plane_words;
		ta.sx = static_cast<uint32_t>(sx); ta.sy = static_cast<uint32_t>(sy); ta.inv = e.graph_permissible ? 0xFFFFFFFFu : 0u;
		ta.sxe = static_cast<uint32_t>(sx + 1); ta.sye = static_cast<uint32_t>(sy + 1); ta.nverts = static_cast<uint32_t>(nverts);
		ta.max_steps = e.t_max_steps.p;
		ta.nbase = e.t_nbase.p; ta.ncap = e.t_ncap.p;
		ta.n_nodes = e.t_counters.p; ta.n_corners = e.t_counters.p + 2 * ns;
		ta.n_starts = e.t_counters.p + 3 * ns; ta.n_items = e.t_counters.p + 4 * ns;
		ta.node_vertex = e.t_node_vertex.p; ta.node_adj = e.t_node_adj.p; ta.vert2node = e.t_vert2node.p;
		ta.cobase = e.t_cobase.p; ta.cocap = e.t_cocap.p; ta.corner_vertex = e.t_corner_vertex.p;
		ta.dart_end = e.t_dart_end.p; ta.dart_len = e.t_dart_len.p; ta.dart_minv = e.t_dart_minv.p; ta.dart_minpos = e.t_dart_minpos.p;
		ta.dart_codes = e.t_dart_codes.p; ta.dart_inline = e.t_dart_inline.p;
		ta.parent = e.t_parent.p; ta.compmin = e.t_compmin.p;
		ta.start_bits = e.t_start_bits.p; ta.start_words = start_words; ta.starts = e.t_starts.p;
		ta.ibase = e.t_ibase.p; ta.icap = e.t_icap.p; ta.items = e.t_items.p; ta.item_off = e.t_item_off.p;
		ta.sbase = e.d_sbase.p; ta.scap = e.d_scap.p; ta.stack_node = e.d_stack_node.p; ta.stack_item = e.d_stack_code.p;
		ta.kbase = e.d_kbase.p; ta.kcap = e.d_kcap.p;
		ta.chain_node = e.d_chain_node.p; ta.chain_item0 = e.t_chain_item0.p; ta.chain_off = e.d_chain_off.p; ta.chain_clen = e.d_chain_clen.p;
		ta.n_chains = e.d_n_chains.p; ta.n_raw = e.d_n_raw.p; ta.n_valid = e.d_n_valid.p;
		ta.cbase = e.d_cbase.p; ta.ccap = e.d_ccap.p; ta.cp = e.d_cp.p; ta.slice_err = e.d_slice_err.p;

		e.last_trail_slices = ns;
		ta.events = e.t_events.p; ta.n_events = e.t_counters.p + 5 * ns; ta.seg_len_sum = e.t_counters.p + 6 * ns; ta.chain_ev0 = e.t_chain_ev0.p; ta.ev_lnd = e.t_ev_lnd.p; ta.ev_item = e.t_ev_item.p;
		ta.dbg = nullptr;
		{ const char* env = getenv("CKL_TRAIL_WALK"); ta.walk_plain = (env && !strcmp(env, "plain")) ? 1u : 0u; ta.walk_no_regs = (env && !strcmp(env, "lds")) ? 1u : 0u; }
		{ const char* env = getenv("CKL_TRAIL_WALK_STACK"); ta.walk_stack_cap = env ? static_cast<uint32_t>(std::max(1, atoi(env))) : 0xFFFFFFFFu; }
		if (kTuning && getenv("CKL_TRAIL_DIAG")) { d_tdbg.ensure(16); CKL_HIP(hipMemsetAsync(d_tdbg.p, 0, 128, s)); ta.dbg = d_tdbg.p; ta_dbg = d_tdbg.p; }
		ta.graph_blocks = e.graph_blocks; ta.blk_special = e.t_blk_special.p; ta.blk_corner = e.t_blk_corner.p;
		int max_lds = 0;
		CKL_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, e.device));
		const size_t budget = static_cast<size_t>(max_lds > 1024 ? max_lds - 1024 : 0);
		const uint32_t dart_blocks = (4 * max_ncap + kWalkChunk * kWaves - 1) / (kWalkChunk * kWaves);
		// union-find table of k_trail_components in LDS: 4 bytes per node
		size_t clds = (static_cast<size_t>(max_special) + 1024) * 4;
		if (const char* env = getenv("CKL_TRAIL_LDS")) clds = static_cast<size_t>(std::max(0, atoi(env)));
		clds = (std::min(budget, std::max<size_t>(clds, 1024)) / 16) * 16;
		CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trail_components), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(clds)));
		// node tables of k_trail_walk in LDS + 16 KiB of branch stack when that fits
		size_t lds = trail_walk_lds(max_special);      // k_trail_walk: 12 bytes of record + 4 of branch stack per node
		if (const char* env = getenv("CKL_TRAIL_LDS")) lds = static_cast<size_t>(std::max(0, atoi(env)));   // testing: small values force the global tables
		lds = std::min(budget, std::max<size_t>(lds, 4096));
		lds = (lds / 16) * 16;
		CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trail_walk), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
		trail_lds_used = lds;
		HT_MARK("c:setup");

		// The serial k_trail_walk keeps 1 wavefront per slice busy for ~1 ms while the chip idles.
		// Slices can be processed in groups on their own streams (CKL_TRAIL_GROUPS): while one
		// group is in its DFS the others run their parallel stages.
		uint32_t groups = 1u;      // measured at C2: 2 groups between -0.17 and +0.1 ms from run to run, 4 and 8 slower (the DFS wavefronts want their SIMDs to themselves)
		if (const char* env = getenv("CKL_TRAIL_GROUPS")) groups = static_cast<uint32_t>(std::max(1, atoi(env)));
		groups = std::min<uint32_t>(std::min<uint32_t>(groups, ns), kTrailStreams);
		const bool stagger = getenv("CKL_TRAIL_STAGGER") != nullptr;
		if (groups > 1) CKL_HIP(hipEventRecord(e.ev_fork, s));
		for (uint32_t g = 0; g < groups; g++) {
			const uint32_t z0 = static_cast<uint32_t>(static_cast<uint64_t>(ns) * g / groups);
			const uint32_t z1 = static_cast<uint32_t>(static_cast<uint64_t>(ns) * (g + 1) / groups);
			const uint32_t gn = z1 - z0;
			if (gn == 0) continue;
			// group 0 stays on the session's own stream: the runtime multiplexes streams onto few hardware
			// queues (4 by default), streams created later share one and serialise
			if (g > 0 && !e.trail_stream[g - 1]) CKL_HIP(hipStreamCreateWithFlags(&e.trail_stream[g - 1], hipStreamNonBlocking));      // only when slice groups are asked for
			hipStream_t gs = g == 0 ? s : e.trail_stream[g - 1];
			// staggered: a group's chip-filling kernels start when the previous group's are through, i.e. beside
			// that group's serial walk (one wavefront per slice), instead of all groups at once
			if (g > 0) CKL_HIP(hipStreamWaitEvent(gs, stagger ? e.ev_pre[g - 1] : e.ev_fork, 0));
			ta.z0 = z0;
			fa.z0 = z0;
			if (any) {
				hipLaunchKernelGGL(k_trail_nodes, dim3(e.graph_blocks, gn), dim3(kBlock), 0, gs, ta);
				hipLaunchKernelGGL(k_trail_segments, dim3(dart_blocks, gn), dim3(kBlock), 0, gs, ta);
				if (max_cocap) hipLaunchKernelGGL(k_trail_loops, dim3((max_cocap + kBlock - 1) / kBlock, gn), dim3(kBlock), 0, gs, ta);
				hipLaunchKernelGGL(k_trail_components, dim3(gn), dim3(kCompBlock), clds, gs, ta, static_cast<uint32_t>(clds));
			}
			if (g == 0) CKL_HIP(hipEventRecord(e.evd0, gs));
			if (groups > 1) CKL_HIP(hipEventRecord(e.ev_pre[g], gs));
			hipLaunchKernelGGL(k_trail_walk, dim3(gn), dim3(kWave), lds, gs, ta, static_cast<uint32_t>(lds));
			if (g == 0) CKL_HIP(hipEventRecord(e.evd1, gs));
			hipLaunchKernelGGL(k_trail_items, dim3(gn), dim3(kItemsBlock), 0, gs, ta);
			hipLaunchKernelGGL(k_trail_offsets, dim3(gn), dim3(kBlock), 0, gs, ta);
			hipLaunchKernelGGL(k_trail_expand, dim3((max_icap + kExpandChunk * kWaves - 1) / (kExpandChunk * kWaves), gn), dim3(kBlock), 0, gs, ta);
			hipLaunchKernelGGL(k_finish, dim3(gn), dim3(kFinishBlock), 0, gs, fa);
			if (g > 0) {
				CKL_HIP(hipEventRecord(e.ev_join[g], gs));
				CKL_HIP(hipStreamWaitEvent(s, e.ev_join[g], 0));
			}
		}
	}
	CKL_HIP(hipEventRecord(e.evk1, s));
	HT_MARK("c:enqueue");
	if (kTuning && getenv("CKL_TRAIL_DIAG")) {
		std::vector<uint32_t> c = download(e.t_counters.p, 5 * static_cast<size_t>(ns), s);
		double m[5] = { 0 };
		for (int k = 0; k < 5; k++) for (uint32_t zi = 0; zi < ns; zi++) m[k] += static_cast<double>(c[static_cast<size_t>(k) * ns + zi]) / ns;
		double sp = 0, co = 0;
		for (uint32_t zi = 0; zi < ns; zi++) { sp += static_cast<double>(e.count_special[zi]) / ns; co += static_cast<double>(e.count_corner[zi]) / ns; }
		if (ta_dbg) {
			std::vector<unsigned long long> g = download(ta_dbg, 16, s);
			if (g[11]) fprintf(stderr, "[ckl trail diag, k_trail_walk] mean cycles per slice: table fill=%.0f fill + walk=%.0f\n", static_cast<double>(g[6]) / g[11], static_cast<double>(g[7]) / g[11]);
			if (g[11]) fprintf(stderr, "[ckl trail diag, k_trail_walk] slices=%llu iterations/slice=%.0f cycles/iteration=%.0f clock=%.2f GHz (cycles / 100 MHz ticks)\n",
				g[11], static_cast<double>(g[8]) / g[11], g[8] ? static_cast<double>(g[9]) / g[8] : 0.0, g[10] ? static_cast<double>(g[9]) / (g[10] * 10.0) : 0.0);
			fprintf(stderr, "[ckl trail diag, k_trail_components, mean cycles per slice] union=%.0f component minima=%.0f starts/splits=%.0f bitmap scan=%.0f\n",
				static_cast<double>(g[12]) / ns, static_cast<double>(g[13]) / ns, static_cast<double>(g[14]) / ns, static_cast<double>(g[15]) / ns);
			fprintf(stderr, "[ckl trail diag, k_trail_segments] waves=%llu iterations/wave mean=%.1f max=%llu cycles/wave mean=%.0f max=%llu cycles/iteration=%.0f active lanes/iteration=%.1f\n",
				g[4], g[4] ? static_cast<double>(g[0]) / g[4] : 0.0, g[1], g[4] ? static_cast<double>(g[2]) / g[4] : 0.0, g[3],
				g[0] ? static_cast<double>(g[2]) / g[0] : 0.0, g[0] ? static_cast<double>(g[5]) / g[0] : 0.0);
		}
		fprintf(stderr, "[ckl trail diag] max degree-1/3/4 vertices of a slice=%u, k_trail_walk LDS=%zu bytes\n", max_special, trail_lds_used);
		fprintf(stderr, "[ckl trail diag, mean per slice] degree-1/3/4 vertices=%.0f corners=%.0f nodes=%.0f starts=%.0f items=%.0f\n", sp, co, m[0], m[3], m[4]);
	}

	// without a markov model nothing stands between k_finish and the final offsets + gather: they are enqueued before the
	// label side takes the host thread (it is host-synchronous and ends after the trail: the trail's queue used to idle
	// 0.2 ms until the host came back to launch them)
	const bool early_tail = !markov_order && result != nullptr;
	auto enqueue_tail = [&]() {
		e.d_out_off.ensure(ns);
		e.d_code_report.ensure(static_cast<size_t>(ns) + 3);
		e.d_codes_out.ensure(e.codes_capacity + 8);
		hipLaunchKernelGGL(k_code_offsets, dim3(1), dim3(kBlock), 0, s, e.d_boc_len.p, e.d_payload_len.p, e.d_slice_err.p, ns, e.d_out_off.p, e.d_code_report.p);
		hipLaunchKernelGGL(k_gather_codes, dim3(ns), dim3(kBlock), 0, s, e.d_boc.p, e.d_bbase.p, e.d_boc_len.p,
			e.d_payload.p, e.d_pbase.p, e.d_payload_len.p, e.d_out_off.p, e.d_codes_out.p);
	};
	if (early_tail) enqueue_tail();
	// the label side (other stream, host-synchronous) runs while the trail kernels execute
	if (overlap) overlap();
	HT_MARK("c:overlap");


	if (markov_order) {
		std::vector<uint8_t> model;
		if (model_in) {
			model = *model_in;
		}
		else {
			const size_t rows = static_cast<size_t>(1) << (2 * markov_order);
			e.d_hist.ensure(rows * 4);
			CKL_HIP(hipMemsetAsync(e.d_hist.p, 0, rows * 4 * sizeof(uint32_t), s));
			const uint32_t hist_lds = (rows * 4 <= 16384) ? static_cast<uint32_t>(rows * 4) : 0u;    // order <= 6: 64 KiB
			if (hist_lds * 4 > 48 * 1024) CKL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_markov_hist), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(hist_lds * 4)));
			hipLaunchKernelGGL(k_markov_hist, dim3(ns), dim3(kBlock), hist_lds * sizeof(uint32_t), s, e.d_dcode.p, e.d_cbase.p, e.d_n_valid.p, markov_order, e.d_hist.p, hist_lds);
			std::vector<uint32_t> hist = download(e.d_hist.p, rows * 4, s);
			if (hist_out) *hist_out = hist;
			if (hist_only) {
				std::vector<uint32_t> errs = download(e.d_slice_err.p, ns, s);
				for (uint32_t zi = 0; zi < ns; zi++) if (errs[zi]) throw Error(CKL_ERR_RUNTIME, "crackle_amd: crack walk scratch overflow on z=" + std::to_string(zi));
				return;
			}
			model = markov_stats_to_model(hist.data(), rows);
		}
		if (model_out) *model_out = model;
		upload(e.d_model, model, s);
		hipLaunchKernelGGL(k_markov_pack, dim3(ns), dim3(kBlock), 0, s, e.d_dcode.p, e.d_cbase.p, e.d_n_valid.p, markov_order,
			e.d_model.p, e.d_pbase.p, e.d_payload.p, e.d_payload_len.p);
	}
	if (!result) { CKL_HIP(hipStreamSynchronize(s)); return; }

	// final offsets on the device, the gather behind them, one small report back (lengths for the
	// z-index, total, error bits): a single wait before the codes are copied out
	if (!early_tail) enqueue_tail();
	HT_MARK("c:finish_enq");
	std::vector<uint32_t> report = download(e.d_code_report.p, static_cast<size_t>(ns) + 3, s);
	HT_MARK("c:finish_wait");
	if (report[ns + 2]) {
		std::vector<uint32_t> errs = download(e.d_slice_err.p, ns, s);
		for (uint32_t zi = 0; zi < ns; zi++) {
			if (errs[zi]) throw Error(CKL_ERR_RUNTIME, "crackle_amd: crack walk scratch overflow on z=" + std::to_string(zi));
		}
	}
	result->code_len.assign(report.begin(), report.begin() + ns);
	const uint64_t otot = static_cast<uint64_t>(report[ns]) | (static_cast<uint64_t>(report[ns + 1]) << 32);
	result->total = otot;
}

struct FlatResult {
	std::vector<uint32_t> ncomp;      // per slice
	std::vector<uint32_t> crcs;       // per slice crc32c of the component image
	uint64_t total = 0;               // components over all slices; mapping stays in e.d_mapping
};

// geometric-sum table of ckl_runs.hpp (k_run_resolve), cached per slice size
void ensure_geom_table(ckl_encoder& e, uint64_t sxy) {
	if (e.g_table_pixels == sxy && e.d_G.p) return;
	hipStream_t s = e.stream2;
	const uint32_t npx = static_cast<uint32_t>(sxy);
	const uint32_t B = 1024, nblk = npx / B + 1;
	std::vector<uint32_t> g_base(B), blk_g(nblk), blk_x(nblk);
	const uint32_t X = gf_xpow(32);
	g_base[0] = 0;
	for (uint32_t i = 1; i < B; i++) g_base[i] = gf_mul(X, g_base[i - 1] ^ 0x80000000u);
	const uint32_t gB = gf_mul(X, g_base[B - 1] ^ 0x80000000u);
	const uint32_t XB = gf_xpow(32ull * B);
	blk_g[0] = 0; blk_x[0] = 0x80000000u;
	for (uint32_t k = 1; k < nblk; k++) {
		blk_g[k] = blk_g[k - 1] ^ gf_mul(blk_x[k - 1], gB);
		blk_x[k] = gf_mul(blk_x[k - 1], XB);
	}
	DevBuf<uint32_t> t_base, t_g, t_x;
	upload(t_base, g_base, s); upload(t_g, blk_g, s); upload(t_x, blk_x, s);
	e.d_G.ensure(static_cast<size_t>(npx) + 1);
	hipLaunchKernelGGL(k_build_geom_table, dim3(npx / kBlock + 1), dim3(kBlock), 0, s, t_base.p, t_g.p, t_x.p, npx, e.d_G.p);
	CKL_HIP(hipStreamSynchronize(s));
	e.g_table_pixels = sxy;
}

// encode_flat per-slice part (labels.hpp:56-88) on runs of the label planes: components
// and their crc32c.  Enqueued on the label stream right after the planes exist; the
// results are collected (flat_collect) while the crack trail runs on the other stream.
void flat_enqueue(ckl_encoder& e, int64_t sx, int64_t sy, int64_t sz) {
	hipStream_t s = e.stream2;
	const uint32_t ns = static_cast<uint32_t>(sz);
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	ensure_geom_table(e, sxy);

	// a run starts at x = 0 of every row and wherever the left neighbour differs
	// (the host tables stay alive in the session: nothing here waits for the uploads)
	std::vector<uint64_t>& rbase = e.h_rbase;
	std::vector<uint32_t>& rcap = e.h_rcap;
	rbase.assign(ns, 0); rcap.assign(ns, 0);
	uint64_t rtot = 0;
	uint32_t max_rcap = 0;
	for (uint32_t zi = 0; zi < ns; zi++) {
		const uint64_t runs = static_cast<uint64_t>(sy) + e.count_v[zi];
		rbase[zi] = rtot; rcap[zi] = static_cast<uint32_t>(runs); rtot += runs;
		max_rcap = std::max<uint32_t>(max_rcap, static_cast<uint32_t>(runs));
	}
	e.flat_max_rcap = max_rcap;
	upload(e.d_rbase, rbase, s); upload(e.d_rcap, rcap, s);
	e.d_word_base.ensure(e.plane_words * ns);
	e.d_parent.ensure(rtot); e.d_run_start.ensure(rtot); e.d_run_cc.ensure(rtot); e.d_comp_pix.ensure(rtot);
	// component counts, crc accumulators, id widths and error words of the slices lie side by side: flat_collect brings
	// them back in ONE transfer (four separate round trips cost the label path 0.2 ms, and it is what the encode ends with)
	e.d_nruns.ensure(ns);
	e.d_flat_report.ensure(4 * static_cast<size_t>(ns));
	e.d_ncomp.borrow(e.d_flat_report.p, ns); e.d_crc_acc.borrow(e.d_flat_report.p + ns, ns);
	e.d_idbits.borrow(e.d_flat_report.p + 2 * static_cast<size_t>(ns), ns); e.d_slice_err2.borrow(e.d_flat_report.p + 3 * static_cast<size_t>(ns), ns);
	CKL_HIP(hipMemsetAsync(e.d_slice_err2.p, 0, ns * sizeof(uint32_t), s));

	RunGeom g;
	g.planeV = e.d_planes.p; g.planeH = e.d_planes.p + e.plane_words * ns;
	g.row_words = e.row_words; g.plane_words = e.plane_words;
	g.flip = 1u;   // a set bit (labels differ) is a break, whatever the stream's crack format
	g.sx = static_cast<uint32_t>(sx); g.sy = static_cast<uint32_t>(sy);
	RunArrays ra;
	ra.word_base = e.d_word_base.p; ra.rbase = e.d_rbase.p; ra.rcap = e.d_rcap.p;
	ra.parent = e.d_parent.p; ra.run_start = e.d_run_start.p; ra.run_cc = e.d_run_cc.p;
	ra.nruns = e.d_nruns.p; ra.ncomp = e.d_ncomp.p; ra.slice_err = e.d_slice_err2.p;
	ra.comp_pix = e.d_comp_pix.p;
	hipLaunchKernelGGL(k_run_index, dim3(ns), dim3(kIndexBlock), 0, s, g, ra);
	launch_run_union(s, ns, g, ra);
	ResolveScratch rs;
	rs.nblk = (max_rcap + kBlock - 1) / kBlock;
	e.d_run_local.ensure(rtot);
	e.d_blk_roots.ensure(static_cast<size_t>(rs.nblk) * ns);
	rs.run_local = e.d_run_local.p; rs.blk_roots = e.d_blk_roots.p;
	launch_run_resolve(s, ns, ra, rs, e.d_G.p, static_cast<uint32_t>(sxy), 0u, e.d_crc_acc.p, e.d_idbits.p);
	HT_MARK("f:enqueue");
}

// component counts, crcs, component -> label (labels.hpp:71-88)
template <typename LABEL>
void flat_collect(ckl_encoder& e, const LABEL* labels, int64_t sx, int64_t sy, int64_t sz, FlatResult& out) {
	hipStream_t s = e.stream2;
	const uint32_t ns = static_cast<uint32_t>(sz);
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	std::vector<uint32_t> acc, idbits, errs;
	if (e.d_ncomp.p == e.d_flat_report.p && e.d_flat_report.p) {      // laid out by flat_enqueue: one transfer
		const std::vector<uint32_t> rep = download(e.d_flat_report.p, 4 * static_cast<size_t>(ns), s);
		out.ncomp.assign(rep.begin(), rep.begin() + ns);
		acc.assign(rep.begin() + ns, rep.begin() + 2 * static_cast<size_t>(ns));
		idbits.assign(rep.begin() + 2 * static_cast<size_t>(ns), rep.begin() + 3 * static_cast<size_t>(ns));
		errs.assign(rep.begin() + 3 * static_cast<size_t>(ns), rep.end());
		HT_MARK("f:wait");
	}
	else {
		out.ncomp = download(e.d_ncomp.p, ns, s);
		HT_MARK("f:wait");
		acc = download(e.d_crc_acc.p, ns, s);
		idbits = download(e.d_idbits.p, ns, s);
		errs = download(e.d_slice_err2.p, ns, s);
	}
	for (uint32_t zi = 0; zi < ns; zi++) if (errs[zi]) throw Error(CKL_ERR_RUNTIME, "crackle_amd: run table overflow on z=" + std::to_string(zi));
	const uint32_t init_term = gf_mul(0xFFFFFFFFu, gf_xpow(32ull * sxy));
	out.crcs.resize(ns);
	uint32_t fix_bits = 0xFFFFFFFFu, fix = 0;
	for (uint32_t zi = 0; zi < ns; zi++) {
		if (idbits[zi] != fix_bits) { fix_bits = idbits[zi]; fix = gf_xpow(32 - fix_bits); }
		out.crcs[zi] = ~(gf_mul(acc[zi], fix) ^ init_term);
	}

	std::vector<uint64_t> comp_off(ns);
	uint64_t total = 0;
	for (uint32_t zi = 0; zi < ns; zi++) { comp_off[zi] = total; total += out.ncomp[zi]; }
	upload(e.d_comp_off, comp_off, s);
	e.d_mapping.ensure(total + 1);
	RunArrays ra;
	ra.word_base = e.d_word_base.p; ra.rbase = e.d_rbase.p; ra.rcap = e.d_rcap.p;
	ra.parent = e.d_parent.p; ra.run_start = e.d_run_start.p; ra.run_cc = e.d_run_cc.p;
	ra.nruns = e.d_nruns.p; ra.ncomp = e.d_ncomp.p; ra.slice_err = e.d_slice_err2.p;
	ra.comp_pix = e.d_comp_pix.p;
	uint32_t max_ncomp = 1;
	for (uint32_t zi = 0; zi < ns; zi++) max_ncomp = std::max(max_ncomp, out.ncomp[zi]);
	hipLaunchKernelGGL(k_mapping_comps<LABEL>, dim3((max_ncomp + kBlock - 1) / kBlock, ns), dim3(kBlock), 0, s,
		labels, ra, sxy, e.d_comp_off.p, e.d_mapping.p);
	CKL_HIP(hipStreamSynchronize(s));   // comp_off (pageable) must be consumed before it goes out of scope
	out.total = total;
}

// the labels with the key of their first column run (the order they enter `pinsets`, src/pins.hpp:126-163), from the
// components' labels and first runs on the device
// The labels with their first column runs (k_pin_label_first / _list) in two steps: the kernels, enqueued as soon as
// first_any is final (behind the first column pass), and the collection of the two short lists — on a stream of its
// own, so that it does not queue behind the passes that follow on the label stream.
struct PinLabelLists {
	DevBuf<uint64_t> d_tab;
	DevBuf<uint32_t> d_count;
	uint32_t slots = 0;
	hipEvent_t ready = nullptr;
	~PinLabelLists() { if (ready) (void)hipEventDestroy(ready); }
};
void pin_label_table_enqueue(ckl_encoder& e, const uint64_t* comp_label, const uint64_t* first_any_dev, uint64_t N, PinLabelLists& t) {
	hipStream_t s = e.stream2;
	if (N > (1ull << 30)) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many components for pin labels");
	uint32_t slots = 1024;
	while (slots < 2 * N) slots <<= 1;
	t.slots = slots;
	t.d_tab.ensure(4ull * slots + 1);      // keys | values | label list | first list, + the all-ones label's minimum
	t.d_count.ensure(1);
	CKL_HIP(hipMemsetAsync(t.d_tab.p, 0xFF, (2ull * slots) * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(t.d_tab.p + 4ull * slots, 0xFF, sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(t.d_count.p, 0, sizeof(uint32_t), s));
	unsigned long long* tab = reinterpret_cast<unsigned long long*>(t.d_tab.p);
	hipLaunchKernelGGL(k_pin_label_first, dim3(static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock)), dim3(kPinBlock), 0, s,
		reinterpret_cast<const unsigned long long*>(comp_label), reinterpret_cast<const unsigned long long*>(first_any_dev), N, tab, tab + slots, slots - 1u, tab + 4ull * slots);
	hipLaunchKernelGGL(k_pin_label_list, dim3((slots + kPinBlock - 1) / kPinBlock), dim3(kPinBlock), 0, s, tab, tab + slots, slots, t.d_count.p, tab + 2ull * slots, tab + 3ull * slots);
	if (!t.ready) CKL_HIP(hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
	CKL_HIP(hipEventRecord(t.ready, s));
}
void pin_label_table_collect(ckl_encoder& e, PinLabelLists& t, PinCandidates& pc) {
	if (!e.stream_tab) CKL_HIP(hipStreamCreateWithFlags(&e.stream_tab, hipStreamNonBlocking));
	hipStream_t s = e.stream_tab;
	CKL_HIP(hipStreamWaitEvent(s, t.ready, 0));
	const uint32_t nl = download(t.d_count.p, 1, s)[0];
	pc.label_value = download(t.d_tab.p + 2ull * t.slots, nl, s);
	pc.label_first = download(t.d_tab.p + 3ull * t.slots, nl, s);
	const uint64_t max_first = download(t.d_tab.p + 4ull * t.slots, 1, s)[0];
	if (max_first != kPinNoKey) { pc.label_value.push_back(kPinNoKey); pc.label_first.push_back(max_first); }
}
void pin_label_table(ckl_encoder& e, const uint64_t* comp_label, const uint64_t* first_any_dev, uint64_t N, PinCandidates& pc) {
	PinLabelLists t;
	pin_label_table_enqueue(e, comp_label, first_any_dev, N, t);
	pin_label_table_collect(e, t, pc);
}

// extract_columns + add_pin (src/pins.hpp:95-163) over the rows of `v`: the kept runs marked in v.mark
template <typename LABEL>
void pin_dedup_pass(ckl_encoder& e, const LABEL* labels, const PinVolume& v) {
	hipStream_t s = e.stream2;
	const bool by_thread = getenv("CKL_PINS_ROW_THREADS") != nullptr;      // testing: the general kernel on small volumes
	if (v.sz <= 1024u && !by_thread) {
		// a wavefront per row, label tables in registers
		const dim3 wgrid((v.sy + kPinWaves - 1) / kPinWaves), wblock(64 * kPinWaves);
		// four columns per load where the rows allow it
		// (2048 x 2048 x 256 uint32, the kernel alone: 19.9 ms with one column per load, 9.2 with four, 9.3 / 9.6 with 8 / 16)
		const bool groups = v.sx % 4u == 0 && (reinterpret_cast<uintptr_t>(labels) % (4 * sizeof(LABEL))) == 0 && !getenv("CKL_PINS_COLUMN_LOADS");
#define CKL_DEDUP(K) do { \
			if (groups) hipLaunchKernelGGL((k_pin_dedup_wave<LABEL, K, 4>), wgrid, wblock, 0, s, labels, v); \
			else hipLaunchKernelGGL((k_pin_dedup_wave<LABEL, K, 1>), wgrid, wblock, 0, s, labels, v); \
		} while (0)
		if (v.sz <= 64u) CKL_DEDUP(1);
		else if (v.sz <= 128u) CKL_DEDUP(2);
		else if (v.sz <= 256u) CKL_DEDUP(4);
		else if (v.sz <= 512u) CKL_DEDUP(8);
		else CKL_DEDUP(16);
#undef CKL_DEDUP
	}
	else {
		// taller volumes: a thread per row with its label tables in global memory
		uint32_t cap = 16;
		while (cap < 2u * v.sz) cap <<= 1;
		const uint64_t slots = 2ull * cap * v.sy;
		e.d_pin_tables.ensure(slots * sizeof(PinSlot));
		CKL_HIP(hipMemsetAsync(e.d_pin_tables.p, 0, slots * sizeof(PinSlot), s));
		hipLaunchKernelGGL(k_pin_dedup<LABEL>, dim3((v.sy + kPinRowBlock - 1) / kPinRowBlock), dim3(kPinRowBlock), 0, s,
			labels, v, reinterpret_cast<PinSlot*>(e.d_pin_tables.p), cap);
	}

}

// extract_columns / compute_multiverse / the component -> pin choice of find_suboptimal_pins
// (src/pins.hpp:95-198, 300-346) as device passes over the resident label volume and
// the component id volume (ckl_pins_dev.hpp); only per-component facts and the chosen pins are copied out.
template <typename LABEL>
void pin_passes_device(
	ckl_encoder& e, const LABEL* labels, const uint32_t* cc /* device: component id of every voxel */,
	int64_t sx_, int64_t sy_, int64_t sz_, uint64_t N, PinVolume& v, unsigned long long*& choice, const uint64_t*& first_any,
	const std::function<void(const uint64_t*)>& first_any_final = std::function<void(const uint64_t*)>()      // called (with first_any) once the pass that completes it is enqueued
) {
	hipStream_t s = e.stream2;
	v = PinVolume();
	v.sx = static_cast<uint32_t>(sx_); v.sy = static_cast<uint32_t>(sy_); v.sz = static_cast<uint32_t>(sz_);
	v.sxy = static_cast<uint64_t>(v.sx) * v.sy;
	const uint64_t voxels = v.sxy * v.sz;
	if (v.sz > 65535u) throw Error(CKL_ERR_ARG, "crackle_amd: pin labels need at most 65535 slices");      // the kept marks hold depth + 1 in 16 bits
	const uint64_t mark_words = (voxels + 1) / 2;
	e.d_pin_kept.ensure(mark_words);
	CKL_HIP(hipMemsetAsync(e.d_pin_kept.p, 0, mark_words * sizeof(uint32_t), s));
	v.cc = cc; v.mark = reinterpret_cast<uint16_t*>(e.d_pin_kept.p);

	pin_dedup_pass<LABEL>(e, labels, v);

	e.d_pin_u64.ensure(4 * N + 1);
	e.d_pin_u32.ensure(N + 1);
	PinComponentArrays a;
	a.first_any = reinterpret_cast<unsigned long long*>(e.d_pin_u64.p);
	a.first_kept = a.first_any + N;
	a.best = a.first_kept + N;
	choice = a.best + N;
	first_any = reinterpret_cast<const uint64_t*>(a.first_any);
	a.first_depth = e.d_pin_u32.p;
	CKL_HIP(hipMemsetAsync(a.first_any, 0xFF, 2 * N * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(a.best, 0, N * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(a.first_depth, 0, N * sizeof(uint32_t), s));
	const dim3 cgrid((v.sx + kPinBlock - 1) / kPinBlock, v.sy);
	hipLaunchKernelGGL((k_pin_columns<LABEL, 0>), cgrid, dim3(kPinBlock), 0, s, labels, v, a);
	if (first_any_final) first_any_final(first_any);
	hipLaunchKernelGGL((k_pin_extent<LABEL, true>), dim3(static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock)), dim3(kPinBlock), 0, s,
		labels, v, a.first_kept, static_cast<uint32_t>(N), a.first_depth);
	hipLaunchKernelGGL((k_pin_columns<LABEL, 2>), cgrid, dim3(kPinBlock), 0, s, labels, v, a);
	hipLaunchKernelGGL(k_pin_choice, dim3(static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock)), dim3(kPinBlock), 0, s, a, N, choice);
}

// `passes`: the PinVolume / choice / first_any of a pin_passes_device call that has run already (else it runs here)
template <typename LABEL>
PinCandidates pin_candidates_device(
	ckl_encoder& e, const LABEL* labels, const uint32_t* cc /* device: component id of every voxel */,
	const uint64_t* comp_label /* device: label of every component */, int64_t sx_, int64_t sy_, int64_t sz_, uint64_t N,
	const PinVolume* passes = nullptr, unsigned long long* choice = nullptr, const uint64_t* first_any_dev = nullptr
) {
	hipStream_t s = e.stream2;
	PinVolume v;
	if (passes) v = *passes;
	else pin_passes_device<LABEL>(e, labels, cc, sx_, sy_, sz_, N, v, choice, first_any_dev);
	struct { const unsigned long long* first_any; } a = { reinterpret_cast<const unsigned long long*>(first_any_dev) };

	CKL_HIP(hipStreamSynchronize(s));
	HT_MARK("p:passes");
	PinCandidates pc;
	pc.comp_label = download(comp_label, N, s);
	pc.comp_first = download(reinterpret_cast<const uint64_t*>(a.first_any), N, s);
	std::vector<uint64_t> chosen = download(reinterpret_cast<const uint64_t*>(choice), N, s);

	HT_MARK("p:d2h");
	if (N >= kPinNone) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many pins");
	// The pins: by default every component's pin is an entry of its own (the same run may appear several
	// times: a pin is taken at most once, since taking it removes every component that maps to it) and no
	// sorting is needed.  The ids of a run are then stored once per component that chose it — volumes
	// with long z-runs could reach N x sz ids — so when the entries' ids pass a budget the chosen runs
	// are reduced to the distinct ones first (a sort of the keys on the host).
	uint32_t P = static_cast<uint32_t>(N);
	std::vector<uint64_t> pin_key(chosen);      // key of pin p (identity mapping: the component's choice)
	pc.comp_pin.assign(N, kPinNone);
	for (uint32_t c = 0; c < P; c++) if (chosen[c] != kPinNoKey) pc.comp_pin[c] = c;
	{
		DevBuf<uint32_t> d_ze0;
		d_ze0.ensure(P);
		CKL_HIP(hipMemsetAsync(d_ze0.p, 0, static_cast<size_t>(P) * sizeof(uint32_t), s));
		hipLaunchKernelGGL((k_pin_extent<LABEL, false>), dim3((P + kPinBlock - 1) / kPinBlock), dim3(kPinBlock), 0, s, labels, v, choice, P, d_ze0.p);
		pc.pin_ze = download(d_ze0.p, P, s);
	}
	uint64_t id_total = 0;
	for (uint32_t c = 0; c < P; c++) if (chosen[c] != kPinNoKey) id_total += pc.pin_ze[c] - static_cast<uint32_t>(chosen[c] % v.sz) + 1u;
	uint64_t id_budget = 1ull << 26;      // 256 MiB of ids
	if (const char* env = getenv("CKL_PIN_IDS_BUDGET")) id_budget = static_cast<uint64_t>(std::max(0, atoi(env)));      // testing: forces the distinct-pin path
	DevBuf<unsigned long long> d_key2;
	const unsigned long long* key_dev = choice;
	if (id_total > id_budget) {
		std::vector<uint32_t> order;
		order.reserve(P);
		for (uint32_t c = 0; c < P; c++) if (chosen[c] != kPinNoKey) order.push_back(c);
		std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return chosen[a] != chosen[b] ? chosen[a] < chosen[b] : a < b; });
		std::vector<uint64_t> keys2;
		std::vector<uint32_t> ze2;
		for (size_t i = 0; i < order.size(); i++) {
			const uint32_t c = order[i];
			if (i == 0 || chosen[c] != chosen[order[i - 1]]) { keys2.push_back(chosen[c]); ze2.push_back(pc.pin_ze[c]); }
			pc.comp_pin[c] = static_cast<uint32_t>(keys2.size() - 1);
		}
		pin_key.swap(keys2);
		pc.pin_ze.swap(ze2);
		P = static_cast<uint32_t>(pin_key.size());
		d_key2.ensure(std::max<size_t>(P, 1));
		if (P) CKL_HIP(hipMemcpyAsync(d_key2.p, pin_key.data(), static_cast<size_t>(P) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
		key_dev = d_key2.p;
	}
	pc.pin_x.assign(P, 0); pc.pin_y.assign(P, 0); pc.pin_zs.assign(P, 0);
	for (uint32_t p = 0; p < P; p++) {
		if (pin_key[p] == kPinNoKey) continue;
		const uint64_t col = pin_key[p] / v.sz;
		pc.pin_zs[p] = static_cast<uint32_t>(pin_key[p] % v.sz);
		pc.pin_x[p] = static_cast<uint32_t>(col % v.sx);
		pc.pin_y[p] = static_cast<uint32_t>(col / v.sx);
	}
	HT_MARK("p:keys");
	pc.pin_ids_off.assign(static_cast<size_t>(P) + 1, 0);
	{
		DevBuf<uint64_t> d_off;
		DevBuf<uint32_t> d_ze, d_ids;
		for (uint32_t p = 0; p < P; p++) {
			if (pin_key[p] == kPinNoKey) { pc.pin_ze[p] = pc.pin_zs[p]; pc.pin_ids_off[p + 1] = pc.pin_ids_off[p]; }
			else pc.pin_ids_off[p + 1] = pc.pin_ids_off[p] + (pc.pin_ze[p] - pc.pin_zs[p] + 1u);
		}
		upload(d_off, pc.pin_ids_off, s);
		upload(d_ze, pc.pin_ze, s);
		d_ids.ensure(pc.pin_ids_off[P] + 1);
		if (P) hipLaunchKernelGGL(k_pin_ids, dim3((P + kPinBlock - 1) / kPinBlock), dim3(kPinBlock), 0, s, v, key_dev, d_ze.p, d_off.p, P, d_ids.p);
		pc.pin_ids = download(d_ids.p, pc.pin_ids_off[P], s);
	}
	pin_label_table(e, comp_label, reinterpret_cast<const uint64_t*>(a.first_any), N, pc);
	HT_MARK("p:ids");
	return pc;
}


// ---- the pin stage sharded by ROWS (config C4 on several GPUs) -----------------------------------------------------
// Candidate pins are z-runs per (x, y) column over the WHOLE volume, so a z-slab cannot find them; a slab of rows
// [y0, y0 + rows) of every slice can: extract_columns / add_pin compare a run only with the label's last pin in the
// previous column of the same row (src/pins.hpp:134-160), so rows are independent, and what is kept per COMPONENT
// (a component lies in one slice but spans rows) is an extremum over its voxels:
//   first_any   smallest key of a run starting in the component                          -> minimum over the ranks
//   first_kept  smallest key of a kept run containing it, with that run's depth          -> minimum of key << 16 | depth
//   best        1 + largest key of a kept run deeper than the first kept one (0: none)   -> maximum, once every rank knows
//                                                                                           the first run's depth
//   z_e + 1     last slice of the chosen run, known to the rank that holds its row        -> maximum (0 elsewhere)
//   ids         component ids along the chosen run, likewise                              -> maximum (0 elsewhere)
// The caller (crackle_amd/distributed.py) holds the arrays in its device memory, reduces them over its process group
// between the calls (RCCL all_reduce of N-entry arrays: 13 MB each for C4's 1.6 M components) and hands the reduced
// arrays back; rank 0 finally runs the ordered cover (ckl_pins_rows_section).  Keys name columns of the whole volume.
template <typename LABEL>
PinVolume pin_rows_volume(ckl_encoder& e, const uint32_t* cc, int64_t sx, int64_t rows, int64_t sz, int64_t y0, bool fresh_marks) {
	PinVolume v;
	v.sx = static_cast<uint32_t>(sx); v.sy = static_cast<uint32_t>(rows); v.sz = static_cast<uint32_t>(sz);
	v.sxy = static_cast<uint64_t>(v.sx) * v.sy;
	v.key_col0 = static_cast<uint64_t>(y0) * v.sx;
	if (v.sz > 65535u) throw Error(CKL_ERR_ARG, "crackle_amd: pin labels need at most 65535 slices");
	const uint64_t mark_words = (v.sxy * v.sz + 1) / 2;
	if (fresh_marks) {
		e.d_pin_kept.ensure(mark_words);
		CKL_HIP(hipMemsetAsync(e.d_pin_kept.p, 0, mark_words * sizeof(uint32_t), e.stream2));
	}
	else if (!e.d_pin_kept.p || e.d_pin_kept.n < mark_words) throw Error(CKL_ERR_ARG, "crackle_amd: ckl_pins_rows_first has to run first");
	v.cc = cc; v.mark = reinterpret_cast<uint16_t*>(e.d_pin_kept.p);
	return v;
}

template <typename LABEL>
void pins_rows_first(ckl_encoder& e, const LABEL* labels, const uint32_t* cc, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t N,
	uint64_t* first_any, uint64_t* first_kept_packed, uint64_t* comp_label) {
	hipStream_t s = e.stream2;
	const PinVolume v = pin_rows_volume<LABEL>(e, cc, sx, rows, sz, y0, true);
	const uint32_t nb = static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock);
	// the label of every component that shows in these rows (0 elsewhere: the ranks' arrays merge by maximum)
	e.d_slice_err2.ensure(1);
	CKL_HIP(hipMemsetAsync(comp_label, 0, N * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(e.d_slice_err2.p, 0, sizeof(uint32_t), s));
	const uint64_t voxels = v.sxy * v.sz;
	const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((voxels + kPinBlock - 1) / kPinBlock, 0x7FFFFFFFull));
	hipLaunchKernelGGL(k_pin_component_labels<LABEL>, dim3(blocks), dim3(kPinBlock), 0, s, labels, cc, voxels, v.sx, N, reinterpret_cast<unsigned long long*>(comp_label), e.d_slice_err2.p);
	pin_dedup_pass<LABEL>(e, labels, v);
	e.d_pin_u64.ensure(N + 1);
	e.d_pin_u32.ensure(N + 1);
	PinComponentArrays a;
	a.first_any = reinterpret_cast<unsigned long long*>(first_any);
	a.first_kept = reinterpret_cast<unsigned long long*>(e.d_pin_u64.p);
	a.best = nullptr;
	a.first_depth = e.d_pin_u32.p;
	CKL_HIP(hipMemsetAsync(a.first_any, 0xFF, N * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(a.first_kept, 0xFF, N * sizeof(uint64_t), s));
	CKL_HIP(hipMemsetAsync(a.first_depth, 0, N * sizeof(uint32_t), s));
	const dim3 cgrid((v.sx + kPinBlock - 1) / kPinBlock, v.sy);
	hipLaunchKernelGGL((k_pin_columns<LABEL, 0>), cgrid, dim3(kPinBlock), 0, s, labels, v, a);
	hipLaunchKernelGGL((k_pin_extent<LABEL, true>), dim3(nb), dim3(kPinBlock), 0, s, labels, v, a.first_kept, static_cast<uint32_t>(N), a.first_depth);
	hipLaunchKernelGGL(k_pin_pack_first, dim3(nb), dim3(kPinBlock), 0, s, a.first_kept, a.first_depth, N, reinterpret_cast<unsigned long long*>(first_kept_packed));
	if (download(e.d_slice_err2.p, 1, s)[0]) throw Error(CKL_ERR_ARG, "crackle_amd: component id out of range");
}

template <typename LABEL>
void pins_rows_best(ckl_encoder& e, const LABEL* labels, const uint32_t* cc, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t N,
	const uint64_t* first_kept_packed, uint64_t* best) {
	hipStream_t s = e.stream2;
	const PinVolume v = pin_rows_volume<LABEL>(e, cc, sx, rows, sz, y0, false);
	const uint32_t nb = static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock);
	e.d_pin_u64.ensure(N + 1);
	e.d_pin_u32.ensure(N + 1);
	PinComponentArrays a;
	a.first_any = nullptr;
	a.first_kept = reinterpret_cast<unsigned long long*>(e.d_pin_u64.p);
	a.best = reinterpret_cast<unsigned long long*>(best);
	a.first_depth = e.d_pin_u32.p;
	hipLaunchKernelGGL(k_pin_unpack_first, dim3(nb), dim3(kPinBlock), 0, s, reinterpret_cast<const unsigned long long*>(first_kept_packed), N, a.first_kept, a.first_depth);
	CKL_HIP(hipMemsetAsync(a.best, 0, N * sizeof(uint64_t), s));
	const dim3 cgrid((v.sx + kPinBlock - 1) / kPinBlock, v.sy);
	hipLaunchKernelGGL((k_pin_columns<LABEL, 2>), cgrid, dim3(kPinBlock), 0, s, labels, v, a);
	CKL_HIP(hipStreamSynchronize(s));
}

template <typename LABEL>
void pins_rows_extent(ckl_encoder& e, const LABEL* labels, const uint32_t* cc, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t N,
	const uint64_t* first_kept_packed, const uint64_t* best, uint64_t* choice, uint32_t* ze_plus1) {
	hipStream_t s = e.stream2;
	const PinVolume v = pin_rows_volume<LABEL>(e, cc, sx, rows, sz, y0, false);
	const uint32_t nb = static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock);
	e.d_pin_u64.ensure(N + 1);
	e.d_pin_u32.ensure(N + 1);
	PinComponentArrays a;
	a.first_any = nullptr;
	a.first_kept = reinterpret_cast<unsigned long long*>(e.d_pin_u64.p);
	a.best = reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(best));
	a.first_depth = e.d_pin_u32.p;
	hipLaunchKernelGGL(k_pin_unpack_first, dim3(nb), dim3(kPinBlock), 0, s, reinterpret_cast<const unsigned long long*>(first_kept_packed), N, a.first_kept, a.first_depth);
	hipLaunchKernelGGL(k_pin_choice, dim3(nb), dim3(kPinBlock), 0, s, a, N, reinterpret_cast<unsigned long long*>(choice));
	CKL_HIP(hipMemsetAsync(ze_plus1, 0, N * sizeof(uint32_t), s));
	hipLaunchKernelGGL((k_pin_extent<LABEL, false, true>), dim3(nb), dim3(kPinBlock), 0, s, labels, v, reinterpret_cast<const unsigned long long*>(choice), static_cast<uint32_t>(N), ze_plus1);
	CKL_HIP(hipStreamSynchronize(s));
}

// The pin label section from the per-component arrays in device memory (every chosen pin an entry of its own:
// pin c is component c's choice): the arrays come to the host in ONE pinned block (host_out_alloc: cached between
// calls), the per-pin bookkeeping runs on the worker threads, pins_cover_host reads the block in place.
// choice[c]: key of the chosen run or kPinNoKey; ze_plus1[c]: its last slice + 1; offsets: N + 1 prefix sums of the
// runs' lengths; ids: the component ids along the runs.
std::vector<uint8_t> pins_section_from_device(
	ckl_encoder& e, int64_t sx, int64_t sy, int64_t sz, uint64_t N, const std::vector<uint32_t>& nc,
	const uint64_t* d_comp_label, const uint64_t* d_first_any, const uint64_t* d_choice, const uint32_t* d_ze_plus1, const uint64_t* d_offsets, const uint32_t* d_ids,
	int stored_width, bool auto_bgcolor, int64_t manual_bgcolor, const PinLabelTable* table = nullptr
) {
	hipStream_t s = e.stream2;
	PinCandidates pc;
	if (!table) pin_label_table(e, d_comp_label, d_first_any, N, pc);      // the labels with their first runs: a few downloads of its own
	const uint64_t total = download(d_offsets + N, 1, s)[0];
	auto up64 = [](uint64_t b) { return (b + 63) & ~static_cast<uint64_t>(63); };
	const uint64_t o_label = 0, o_choice = o_label + up64(N * 8), o_off = o_choice + up64(N * 8), o_ze = o_off + up64((N + 1) * 8), o_ids = o_ze + up64(N * 4);
	const uint64_t o_pins = o_ids + up64(std::max<uint64_t>(total, 1) * 4);      // comp_pin, pin_x, pin_y, pin_zs, pin_ze: written by the host, in the same cached block (no page faults, no zero fill)
	const uint64_t bytes = o_pins + 5 * up64(N * 4);
	struct Block { uint8_t* p = nullptr; hipStream_t s = nullptr; ~Block() { if (p) { (void)hipStreamSynchronize(s); host_out_free(p); } } } blk;      // (copies may still be on their way when an error unwinds)
	blk.s = s;
	blk.p = static_cast<uint8_t*>(host_out_alloc(bytes));
	CKL_HIP(hipMemcpyAsync(blk.p + o_label, d_comp_label, N * 8, hipMemcpyDeviceToHost, s));
	CKL_HIP(hipMemcpyAsync(blk.p + o_choice, d_choice, N * 8, hipMemcpyDeviceToHost, s));
	CKL_HIP(hipMemcpyAsync(blk.p + o_off, d_offsets, (N + 1) * 8, hipMemcpyDeviceToHost, s));
	CKL_HIP(hipMemcpyAsync(blk.p + o_ze, d_ze_plus1, N * 4, hipMemcpyDeviceToHost, s));
	if (total) CKL_HIP(hipMemcpyAsync(blk.p + o_ids, d_ids, total * 4, hipMemcpyDeviceToHost, s));
	uint32_t* const h_comp_pin = reinterpret_cast<uint32_t*>(blk.p + o_pins);
	uint32_t* const h_x = reinterpret_cast<uint32_t*>(blk.p + o_pins + up64(N * 4));
	uint32_t* const h_y = reinterpret_cast<uint32_t*>(blk.p + o_pins + 2 * up64(N * 4));
	uint32_t* const h_zs = reinterpret_cast<uint32_t*>(blk.p + o_pins + 3 * up64(N * 4));
	uint32_t* const h_ze = reinterpret_cast<uint32_t*>(blk.p + o_pins + 4 * up64(N * 4));
	pc.view_comp_pin = h_comp_pin; pc.view_pin_x = h_x; pc.view_pin_y = h_y; pc.view_pin_zs = h_zs; pc.view_pin_ze = h_ze;
	pc.view_components = N;
	pc.view_comp_label = reinterpret_cast<const uint64_t*>(blk.p + o_label);
	pc.view_pin_ids_off = reinterpret_cast<const uint64_t*>(blk.p + o_off);
	pc.view_pin_ids = reinterpret_cast<const uint32_t*>(blk.p + o_ids);
	HT_MARK("p:enqueue");
	const uint64_t* chosen = reinterpret_cast<const uint64_t*>(blk.p + o_choice);
	const uint32_t* ze_plus1 = reinterpret_cast<const uint32_t*>(blk.p + o_ze);
	const uint64_t usx = static_cast<uint64_t>(sx), usz = static_cast<uint64_t>(sz);
	// the copies travel while the cover builds its table of the labels; it calls back when it needs the arrays
	auto arrays_ready = [&]() {
	CKL_HIP(hipStreamSynchronize(s));
	host_parallel_for(N, 65536, [&](size_t lo, size_t hi) {
		for (size_t c = lo; c < hi; c++) {
			if (chosen[c] == kPinNoKey) { h_comp_pin[c] = kPinNone; h_x[c] = h_y[c] = h_zs[c] = h_ze[c] = 0; continue; }
			if (ze_plus1[c] == 0) throw Error(CKL_ERR_RUNTIME, "crackle_amd: a chosen pin lies in no rank's rows");
			h_comp_pin[c] = static_cast<uint32_t>(c);
			const uint64_t col = chosen[c] / usz;
			h_zs[c] = static_cast<uint32_t>(chosen[c] % usz);
			h_ze[c] = ze_plus1[c] - 1u;
			h_x[c] = static_cast<uint32_t>(col % usx);
			h_y[c] = static_cast<uint32_t>(col / usx);
		}
	});
	};
	Header h;
	h.sx = static_cast<uint32_t>(sx); h.sy = static_cast<uint32_t>(sy); h.sz = static_cast<uint32_t>(sz);
	return pins_cover_host(pc, sx, sy, sz, nc, N, h.pin_index_width(), stored_width, auto_bgcolor, manual_bgcolor, arrays_ready, table);
}

// The whole pin stage of a volume that one device holds: the passes of pin_candidates_device, then the chosen
// runs' ends, lengths, offsets (a device scan) and ids without a visit to the host, then pins_section_from_device.
// Volumes whose id lists pass the budget (long z-runs chosen by many components) take pin_candidates_device's
// route, which reduces the chosen runs to the distinct ones first.
template <typename LABEL>
std::vector<uint8_t> pins_section_plain(
	ckl_encoder& e, const LABEL* labels, const uint32_t* cc, const uint64_t* comp_label, int64_t sx, int64_t sy, int64_t sz, uint64_t N,
	const std::vector<uint32_t>& nc, int index_width, int stored_width, bool auto_bgcolor, int64_t manual_bgcolor
) {
	hipStream_t s = e.stream2;
	if (N >= kPinNone) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many pins");
	unsigned long long* choice = nullptr;
	const uint64_t* first_any = nullptr;
	PinVolume v;
	if (getenv("CKL_PINS_ALONE")) CKL_HIP(hipStreamSynchronize(e.stream));      // measuring: the pin kernels without the trail's beside them
	// The labels' first runs are final after the first column pass: their lists are made there and come to the host
	// on a stream of their own, and the host builds the label table (7 ms at C4) while the later passes still run.
	PinLabelLists lists;
	const bool early_table = !getenv("CKL_PINS_LATE_TABLE");
	pin_passes_device<LABEL>(e, labels, cc, sx, sy, sz, N, v, choice, first_any,
		early_table ? std::function<void(const uint64_t*)>([&](const uint64_t* fa) { pin_label_table_enqueue(e, comp_label, fa, N, lists); }) : std::function<void(const uint64_t*)>());
	const uint32_t nb = static_cast<uint32_t>((N + kPinBlock - 1) / kPinBlock);
	const uint32_t pieces = static_cast<uint32_t>((N + kPinScanPiece - 1) / kPinScanPiece);
	DevBuf<uint32_t> d_zep, d_count, d_ze, d_ids;
	DevBuf<unsigned long long> d_piece, d_off;
	d_zep.ensure(N); d_count.ensure(N); d_piece.ensure(static_cast<size_t>(pieces) + 1); d_off.ensure(N + 1);
	CKL_HIP(hipMemsetAsync(d_zep.p, 0, N * sizeof(uint32_t), s));
	hipLaunchKernelGGL((k_pin_extent<LABEL, false, true>), dim3(nb), dim3(kPinBlock), 0, s, labels, v, choice, static_cast<uint32_t>(N), d_zep.p);
	hipLaunchKernelGGL(k_pin_id_counts, dim3(nb), dim3(kPinBlock), 0, s, choice, d_zep.p, v.sz, N, d_count.p);
	hipLaunchKernelGGL(k_pin_scan_pieces, dim3(pieces), dim3(kPinBlock), 0, s, d_count.p, N, d_piece.p);
	hipLaunchKernelGGL(k_pin_scan_tops, dim3(1), dim3(kPinBlock), 0, s, d_piece.p, pieces);
	hipLaunchKernelGGL(k_pin_scan_offsets, dim3(pieces), dim3(kPinBlock), 0, s, d_count.p, N, d_piece.p, pieces, d_off.p);
	std::shared_ptr<const PinLabelTable> table;
	if (early_table) {
		PinCandidates lc;
		pin_label_table_collect(e, lists, lc);
		HT_MARK("p:lists");
		table = pins_label_table_host(lc.label_value, lc.label_first);
		HT_MARK("p:table");
	}
	const uint64_t total = download(d_off.p + N, 1, s)[0];
	HT_MARK("p:passes");
	uint64_t id_budget = 1ull << 26;      // 256 MiB of ids
	if (const char* env = getenv("CKL_PIN_IDS_BUDGET")) id_budget = static_cast<uint64_t>(std::max(0, atoi(env)));      // testing: forces the distinct-pin path
	if (total > id_budget || getenv("CKL_PINS_HOST_BOOKKEEPING")) {
		const PinCandidates pc = pin_candidates_device<LABEL>(e, labels, cc, comp_label, sx, sy, sz, N, &v, choice, first_any);
		HT_MARK("pins_device");
		return pins_cover_host(pc, sx, sy, sz, nc, N, index_width, stored_width, auto_bgcolor, manual_bgcolor);
	}
	d_ze.ensure(N); d_ids.ensure(total + 1);
	hipLaunchKernelGGL(k_pin_minus1, dim3(nb), dim3(kPinBlock), 0, s, d_zep.p, N, d_ze.p);
	hipLaunchKernelGGL(k_pin_ids, dim3(nb), dim3(kPinBlock), 0, s, v, choice, d_ze.p, reinterpret_cast<const uint64_t*>(d_off.p), static_cast<uint32_t>(N), d_ids.p);
	std::vector<uint8_t> bin = pins_section_from_device(e, sx, sy, sz, N, nc, comp_label, first_any, reinterpret_cast<const uint64_t*>(choice), d_zep.p,
		reinterpret_cast<const uint64_t*>(d_off.p), d_ids.p, stored_width, auto_bgcolor, manual_bgcolor, table.get());
	HT_MARK("pins_host");
	return bin;
}

// The flat label section (labels.hpp:92-152) on device: sort + unique of the component
// labels, keys by binary search, everything packed at its byte width into e.d_labels_bin.
// Returns the section size; num_unique comes back for the header arithmetic only.
uint64_t flat_section(ckl_encoder& e, uint64_t N, int stored_width, int component_width, uint32_t ns, const ckl_encode_overrides* ov = nullptr) {
	hipStream_t s = e.stream2;
	if (N > 0x7FFFFFFFull) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many components");
	e.d_n_uniq.ensure(2);
	e.d_uniq.ensure(N + 1);
	// the values to sort: the distinct labels when the components are many (a hash pass, then the
	// sort network runs over tens of thousands instead of millions of keys), else all of them
	const uint64_t* sort_src = e.d_mapping.p;
	uint64_t n_sort = N;
	if (N > 8192 && N <= (1ull << 26) && !getenv("CKL_LABEL_SORT_ALL")) {
		uint32_t slots = 16384;
		while (slots < 2 * N) slots <<= 1;
		e.d_label_hash.ensure(slots);
		CKL_HIP(hipMemsetAsync(e.d_label_hash.p, 0xFF, static_cast<size_t>(slots) * sizeof(uint64_t), s));
		CKL_HIP(hipMemsetAsync(e.d_n_uniq.p, 0, 2 * sizeof(uint32_t), s));
		unsigned long long* table = reinterpret_cast<unsigned long long*>(e.d_label_hash.p);
		hipLaunchKernelGGL(k_label_hash_insert, dim3(static_cast<uint32_t>((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, e.d_mapping.p, static_cast<uint32_t>(N), table, slots - 1u, e.d_n_uniq.p + 1);
		const uint32_t hb = (slots + kUniqItems - 1) / kUniqItems;
		e.d_uniq_blk.ensure(hb + 1);
		hipLaunchKernelGGL(k_label_hash_count, dim3(hb), dim3(kBlock), 0, s, table, slots, e.d_uniq_blk.p);
		hipLaunchKernelGGL(k_unique_scan, dim3(1), dim3(kBlock), 0, s, e.d_uniq_blk.p, hb, e.d_n_uniq.p);
		hipLaunchKernelGGL(k_label_hash_scatter, dim3(hb), dim3(kBlock), 0, s, table, slots, e.d_uniq_blk.p, e.d_n_uniq.p, e.d_n_uniq.p + 1, e.d_uniq.p);
		const std::vector<uint32_t> found = download(e.d_n_uniq.p, 2, s);
		n_sort = static_cast<uint64_t>(found[0]) + (found[1] ? 1u : 0u);
		if (n_sort == 0 || n_sort > N) throw Error(CKL_ERR_RUNTIME, "crackle_amd: label table pass failed");
		e.d_label_list.ensure(n_sort);
		CKL_HIP(hipMemcpyAsync(e.d_label_list.p, e.d_uniq.p, n_sort * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
		sort_src = e.d_label_list.p;
	}
	// sharded encode: the ranks' lists are merged (and sorted) by the caller anyway, so the slab's distinct
	// labels go out as the hash pass left them and the local sort is skipped
	const bool merge_unsorted = ov && ov->merge_unique && sort_src == e.d_label_list.p;
	uint32_t n_pad = 2048;
	while (n_pad < n_sort) n_pad <<= 1;
	if (!merge_unsorted) {
	e.d_sorted.ensure(n_pad);
	const uint32_t blocks = (n_pad + kBlock - 1) / kBlock;
	hipLaunchKernelGGL(k_pad_copy_u64, dim3(blocks), dim3(kBlock), 0, s, sort_src, n_sort, e.d_sorted.p, static_cast<uint64_t>(n_pad));
	hipLaunchKernelGGL(k_bitonic_first, dim3(n_pad / 2048), dim3(kBlock), 0, s, e.d_sorted.p, n_pad);
	for (uint32_t k = 4096; k <= n_pad; k <<= 1) {
		uint32_t j = k >> 1;
		for (; j >= 4096 && !getenv("CKL_BITONIC_SINGLE_STEPS"); j >>= 2) hipLaunchKernelGGL(k_bitonic_step2, dim3((n_pad / 4 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, e.d_sorted.p, j, k, n_pad);
		for (; j >= 2048; j >>= 1) hipLaunchKernelGGL(k_bitonic_step, dim3(blocks), dim3(kBlock), 0, s, e.d_sorted.p, j, k, n_pad);
		hipLaunchKernelGGL(k_bitonic_local, dim3(n_pad / 2048), dim3(kBlock), 0, s, e.d_sorted.p, j, k, n_pad);
	}
	{
		const uint32_t ub = static_cast<uint32_t>((n_sort + kUniqItems - 1) / kUniqItems);
		e.d_uniq_blk.ensure(ub + 1);
		hipLaunchKernelGGL(k_unique_count, dim3(ub), dim3(kBlock), 0, s, e.d_sorted.p, static_cast<uint32_t>(n_sort), e.d_uniq_blk.p);
		hipLaunchKernelGGL(k_unique_scan, dim3(1), dim3(kBlock), 0, s, e.d_uniq_blk.p, ub, e.d_n_uniq.p);
		hipLaunchKernelGGL(k_unique_scatter, dim3(ub), dim3(kBlock), 0, s, e.d_sorted.p, static_cast<uint32_t>(n_sort), e.d_uniq_blk.p, e.d_uniq.p);
	}
	}
	uint64_t uniq_bound = N;      // entries of the unique list the section kernel may have to write
	if (ov && ov->merge_unique) {
		// sharded encode: the keys are written against the unique labels of all slabs.  The caller
		// exchanges the lists now, under the crack trail that is still running on the other stream.
		HT_MARK("l:enqueue");
		const uint32_t n_local = merge_unsorted ? static_cast<uint32_t>(n_sort) : download(e.d_n_uniq.p, 1, s)[0];
		std::vector<uint64_t> local = download(merge_unsorted ? e.d_label_list.p : e.d_uniq.p, n_local, s);      // distinct; sorted unless merge_unsorted
		HT_MARK("l:local");
		const uint64_t* merged = nullptr;
		uint64_t n_merged = 0;
		if (ov->merge_unique(ov->merge_ctx, local.data(), n_local, &merged, &n_merged) != 0 || (!merged && n_merged))
			throw Error(CKL_ERR_RUNTIME, "crackle_amd: the merge_unique callback failed");
		HT_MARK("l:merge");
		if (n_merged < n_local || n_merged > 0xFFFFFFFFull) throw Error(CKL_ERR_ARG, "crackle_amd: merge_unique returned a list that cannot contain the slab's labels");
		e.d_uniq.ensure(n_merged + 1);
		if (n_merged) CKL_HIP(hipMemcpyAsync(e.d_uniq.p, merged, n_merged * sizeof(uint64_t), hipMemcpyHostToDevice, s));
		const uint32_t nm = static_cast<uint32_t>(n_merged);
		CKL_HIP(hipMemcpyAsync(e.d_n_uniq.p, &nm, sizeof(uint32_t), hipMemcpyHostToDevice, s));
		CKL_HIP(hipStreamSynchronize(s));      // `nm` and the caller's list may go away
		e.d_labels_bin.ensure(8 + n_merged * static_cast<uint64_t>(stored_width) + static_cast<uint64_t>(ns) * component_width + N * 4 + 16);
		uniq_bound = std::max<uint64_t>(uniq_bound, n_merged);
	}
	// worst case: every component has its own label and 4-byte keys
	e.d_labels_bin.ensure(8 + N * static_cast<uint64_t>(stored_width) + static_cast<uint64_t>(ns) * component_width + N * 4 + 16);
	const uint64_t work = std::max<uint64_t>(std::max<uint64_t>(uniq_bound, ns), 1);
	hipLaunchKernelGGL(k_flat_section, dim3(static_cast<uint32_t>((work + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
		e.d_uniq.p, e.d_n_uniq.p, stored_width, e.d_ncomp.p, ns, component_width, e.d_mapping.p, static_cast<uint32_t>(N), e.d_labels_bin.p);
	HT_MARK("l:enqueue");
	const uint32_t nu = download(e.d_n_uniq.p, 1, s)[0];
	HT_MARK("l:wait");
	return 8 + static_cast<uint64_t>(nu) * stored_width + static_cast<uint64_t>(ns) * component_width + N * static_cast<uint64_t>(byte_width(nu));
}

template <typename LABEL>
void encode_typed(
	ckl_encoder& e, const LABEL* labels, int64_t sx, int64_t sy, int64_t sz,
	bool allow_pins, bool fortran_order, uint64_t markov_model_order,
	bool optimize_pins, bool auto_bgcolor, int64_t manual_bgcolor,
	const ckl_encode_overrides* ov, uint8_t** out, uint64_t* out_len
) {
	const uint64_t voxels = static_cast<uint64_t>(sx) * sy * sz;
	hipStream_t s = e.stream;
	CKL_HIP(hipEventRecord(e.ev0, s));

	HostTimer ht;
	g_ht = &ht;
	VolumeStats st;
	e.trail_zeroed = false;
	const bool planes_cached = ov && voxels > 0 && e.planes_for == static_cast<const void*>(labels)
		&& e.planes_dims[0] == sx && e.planes_dims[1] == sy && e.planes_dims[2] == sz;
	e.planes_for = nullptr;       // single use: the caller may change the volume afterwards
	const bool labels_first = getenv("CKL_NO_OVERLAP") || static_cast<uint64_t>(sx) * sy > (1536ull * 1536ull)
		|| (getenv("CKL_LABELS_AT_WALK") && atoi(getenv("CKL_LABELS_AT_WALK")) == 0);
	// (a label stream that starts in front of the graph kernel sizes its run arrays from the planes' counts: no deferral then)
	if (voxels > 0 && !planes_cached) planes_pass<LABEL>(e, labels, sx, sy, sz, &st, !labels_first);
	else if (planes_cached) st = e.planes_stats;      // overrides that force nothing still decide from the volume
	ht.mark("planes");
	bool graph_done = false;
	if (e.planes_deferred) {
		// the planes kernel's counts are still on the device: the graph kernel goes behind it without a round
		// trip (it reads the crack format's deciding pair count there), both kernels' counts come in one
		int perm_mode = 2;
		if (ov && ov->force_crack_format >= 0) perm_mode = ov->force_crack_format == PERMISSIBLE ? 1 : 0;
		e.trail_for = nullptr;
		graph_pass(e, sx, sy, sz, perm_mode, &st);
		graph_done = true;
		ht.mark("graph");
	}
	int stored_width = byte_width(st.max_label);                     // crackle.hpp:233-235
	if (ov && ov->force_stored_width) stored_width = ov->force_stored_width;

	Header head;
	head.crack_format = IMPERMISSIBLE;
	head.label_format = PINS_VARIABLE_WIDTH;
	if (static_cast<int64_t>(st.pairs) < static_cast<int64_t>(voxels) / 2) {   // crackle.hpp:50-55
		head.crack_format = PERMISSIBLE;
		head.label_format = FLAT;
	}
	if (ov && ov->force_crack_format >= 0) {
		head.crack_format = ov->force_crack_format;
		head.label_format = ov->force_crack_format == PERMISSIBLE ? FLAT : PINS_VARIABLE_WIDTH;
	}
	if (sz == 1 || !allow_pins) head.label_format = FLAT;           // crackle.hpp:62-64
	if (ov && ov->force_label_format >= 0) head.label_format = ov->force_label_format;
	e.uploads_by_kernel = head.label_format == FLAT;
	head.is_signed = false;
	head.data_width = static_cast<int>(sizeof(LABEL));
	head.stored_data_width = stored_width;
	head.sx = static_cast<uint32_t>(sx); head.sy = static_cast<uint32_t>(sy); head.sz = static_cast<uint32_t>(sz);
	head.log2_grid_size = 31;
	head.fortran_order = fortran_order;
	head.markov_model_order = static_cast<int>(markov_model_order & 0xFF);
	head.is_sorted = true;

	if (voxels == 0) {   // crackle.hpp:96-98
		std::vector<uint8_t> hb;
		head.write(hb);
		uint8_t* o = static_cast<uint8_t*>(malloc(hb.size()));
		if (!o) throw Error(CKL_ERR_RUNTIME, "crackle_amd: out of host memory");
		memcpy(o, hb.data(), hb.size());
		*out = o; *out_len = hb.size();
		return;
	}
	if (optimize_pins) throw Error(CKL_ERR_ARG, "crackle_amd: allow_pins=2 (find_optimal_pins) is out of scope");
	if (head.label_format != FLAT && head.label_format != PINS_VARIABLE_WIDTH) throw Error(CKL_ERR_ARG, "crackle_amd: unsupported label format");
	if (head.markov_model_order > 13) throw Error(CKL_ERR_ARG, "crackle_amd: markov_model_order > 13 is not supported on device");

	// labels (labels.hpp:30-155): components + crcs start on the label stream now and are
	// collected while the crack trail runs
	// When does the label stream's component labelling run?  Beside the trail's chip-filling kernels the two
	// take turns (graph 0.2 -> 0.65 ms at C2); the serial walk leaves the chip idle for a millisecond, and
	// two walks per CU leave room for other workgroups when they keep to a third of the LDS each.  Then the
	// label stream starts with the walk (it waits for the event in front of it).  Otherwise it starts now,
	// in front of the graph kernel: slices too large for that (decided from their size, before the node
	// counts are known).  (The sharded encode's label stream also carries the ranks' exchange of unique
	// labels: 0.5 ms on the host.  It fits since the slab's labels are exchanged unsorted.)
	bool labels_at_walk = false;
	if (labels_first && !graph_done) flat_enqueue(e, sx, sy, sz);
	const bool trail_cached = !graph_done && planes_cached && ov && ov->has_model && head.markov_model_order > 0 && e.trail_for == static_cast<const void*>(labels)
		&& e.trail_perm == (head.crack_format == PERMISSIBLE) && e.trail_order == head.markov_model_order && !getenv("CKL_NO_TRAIL_REUSE");
	e.trail_for = nullptr;
	if (graph_done && e.graph_permissible != (head.crack_format == PERMISSIBLE)) throw Error(CKL_ERR_RUNTIME, "crackle_amd: internal: device and host disagree on the crack format");
	if (!trail_cached && !graph_done) graph_pass(e, sx, sy, sz, head.crack_format == PERMISSIBLE ? 1 : 0);
	ht.mark("graph");
	if (!labels_first) {
		uint32_t max_special = 0;
		for (uint32_t v : e.count_special) max_special = std::max(max_special, v);
		int max_lds = 0;
		CKL_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, e.device));
		labels_at_walk = max_special < 5000 && 2 * trail_walk_lds(max_special) + 40960 <= static_cast<size_t>(max_lds);
		if (!labels_at_walk) flat_enqueue(e, sx, sy, sz);
	}

	const bool permissible = head.crack_format == PERMISSIBLE;
	// if no slice has a crack edge the reference resets the markov order to 0 (crackle.hpp:107-118)
	{
		const uint64_t interior = static_cast<uint64_t>(sx > 0 ? sx - 1 : 0) * sy + static_cast<uint64_t>(sx) * (sy > 0 ? sy - 1 : 0);
		bool any = false;
		for (int64_t z = 0; z < sz && !any; z++) {
			const uint64_t differ = static_cast<uint64_t>(e.count_v[z]) + e.count_h[z];
			any = (permissible ? interior - differ : differ) > 0;
		}
		if (!any && !(ov && ov->has_model)) head.markov_model_order = 0;
	}
	std::vector<uint8_t> forced_model;
	const std::vector<uint8_t>* model_in = nullptr;
	if (ov && ov->has_model && head.markov_model_order > 0) {
		const size_t rows = static_cast<size_t>(1) << (2 * head.markov_model_order);
		forced_model.assign(ov->model, ov->model + rows * 4);
		model_in = &forced_model;
	}

	FlatResult fr;
	std::vector<uint8_t> pins_binary;      // pin label section (host built)
	uint64_t label_bytes = 0;
	const uint64_t off_index = Header::kBytes;
	const uint64_t off_labels = off_index + 4ull * (sz + 1);
	// crack code bytes, estimated from the edge and node counts: every edge is one code, a branch
	// costs a few more; 2 bits per code (markov: at most 3), BOC index of a few chains per slice
	uint64_t est_code_bytes = 0;
	{
		const uint64_t interior = static_cast<uint64_t>(sx > 0 ? sx - 1 : 0) * sy + static_cast<uint64_t>(sx) * (sy > 0 ? sy - 1 : 0);
		const int xw = byte_width(static_cast<uint64_t>(sx) + 1), yw = byte_width(static_cast<uint64_t>(sy) + 1);
		for (int64_t z = 0; z < sz; z++) {
			const uint64_t differ = static_cast<uint64_t>(e.count_v[z]) + e.count_h[z];
			const uint64_t E = permissible ? interior - differ : differ;
			const uint64_t codes = E + 4ull * (static_cast<uint64_t>(e.count_special[z]) + e.count_corner[z]) + 16;
			est_code_bytes += (head.markov_model_order ? (3 * codes + 7) / 8 : (codes + 3) / 4) + 8;
			est_code_bytes += 4 + yw + 64ull * (yw + 2 * xw);
		}
	}
	if (getenv("CKL_SMALL_ESTIMATE")) est_code_bytes = 0;      // testing: the stream outgrows the early buffer
	const uint64_t model_bytes_est = head.markov_model_order ? (1ull << (2 * head.markov_model_order)) : 0;
	struct HostOut { void* p = nullptr; ~HostOut() { if (p) host_out_free(p); } } early;
	uint64_t early_cap = 0;
	uint32_t labels_crc = 0;
	const int component_width = byte_width(static_cast<uint64_t>(sx) * sy);
	hipStream_t s2 = e.stream2;
	bool labels_stay = false;      // set by label_side: the label section and its crc stay on the device until the background copy
	auto label_side = [&]() {
		if (labels_at_walk) {
			CKL_HIP(hipStreamWaitEvent(s2, e.evd0, 0));      // (starting one kernel earlier, beside k_trail_components, cost 0.2 ms)
			flat_enqueue(e, sx, sy, sz);
		}
		flat_collect<LABEL>(e, labels, sx, sy, sz, fr);
		HT_MARK("flat");
		const uint64_t N = fr.total;
		if (head.label_format == PINS_VARIABLE_WIDTH) {
			// pins (pins.hpp:348-403, labels.hpp:157-344): components and crcs come from the
			// device passes above; the order-sensitive cover runs on the host (ckl_pins.hip)
			if (N > 0xFFFFFFFFull) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many components");
			e.d_cc_volume.ensure(voxels);
			launch_paint_components(s2, e.d_planes.p, e.row_words, e.plane_words, sx, sy, sz, e.d_word_base.p, e.d_rbase.p, e.d_run_cc.p, e.d_comp_off.p, 0u, e.d_cc_volume.p);
			pins_binary = pins_section_plain<LABEL>(e, labels, e.d_cc_volume.p, e.d_mapping.p, sx, sy, sz, N, fr.ncomp, head.pin_index_width(), stored_width, auto_bgcolor, manual_bgcolor);
			label_bytes = pins_binary.size();
		}
		else {
			label_bytes = flat_section(e, N, stored_width, component_width, static_cast<uint32_t>(sz), ov);
			HT_MARK("label_table");
			labels_stay = e.async_host_copy && e.keep_device_stream && !e.defer_codes && label_bytes > 0 && !getenv("CKL_LABELS_CRC_HOST");
			// The output buffer is taken now, sized with an estimate of the crack code bytes, so that
			// the label section is copied out and checksummed while the trail still runs; a stream
			// that outgrows the estimate is moved to a larger buffer at assembly.
			early_cap = off_labels + label_bytes + model_bytes_est + est_code_bytes + 4ull * (sz + 1) + 64;
			early.p = host_out_alloc(early_cap);
			HT_MARK("l:buffer");
			uint8_t* eo = static_cast<uint8_t*>(early.p);
			// (one copy: four pieces with an event each, checksummed while the next was on the link, saved 0.04 ms
			// and now and then stalled the enqueueing thread for milliseconds)
			if (labels_stay) {
				// the caller goes on from the stream in HBM: the section's crc32c is taken on the device and the section itself travels
				// to the host with the crack codes, in the background (until round 5 both sat on the encode's tail: 8 MB over the link,
				// then the host's crc over them, 0.25 ms at C2)
				e.d_labels_crc.ensure(1 + kBlock);
				if (!e.ev_labels_crc) CKL_HIP(hipEventCreateWithFlags(&e.ev_labels_crc, hipEventDisableTiming));
				crc32c_device(e.d_labels_bin.p, label_bytes, e.d_labels_crc.p + 1, e.d_labels_crc.p, s2);
				CKL_HIP(hipEventRecord(e.ev_labels_crc, s2));
			}
			else {
				if (label_bytes) CKL_HIP(hipMemcpyAsync(eo + off_labels, e.d_labels_bin.p, label_bytes, hipMemcpyDeviceToHost, s2));
				CKL_HIP(hipStreamSynchronize(s2));
				labels_crc = crc32c(eo + off_labels, label_bytes);
			}
			HT_MARK("labels_d2h");
		}
	};

	CrackResult cr;
	std::vector<uint8_t> model, stored_model;
	if (getenv("CKL_NO_OVERLAP")) {   // diagnostic: kernel timings without the two streams competing
		crack_pass(e, sx, sy, sz, permissible, head.markov_model_order, false, model_in, nullptr, &model, &cr, std::function<void()>(), trail_cached);
		label_side();
	}
	else crack_pass(e, sx, sy, sz, permissible, head.markov_model_order, false, model_in, nullptr, &model, &cr, label_side, trail_cached);
	ht.mark("cracks");
	if (head.markov_model_order > 0) stored_model = markov_model_to_stored(model);

	// assembly (crackle.hpp:171-216), straight into the caller's buffer:
	// header | z-index + crc | labels | model | crack codes | labels crc | slice crcs
	head.num_label_bytes = label_bytes;
	const uint64_t off_model = off_labels + label_bytes;
	const uint64_t off_codes = off_model + stored_model.size();
	const uint64_t off_tail = off_codes + cr.total;
	const uint64_t total = off_tail + 4ull * (sz + 1);
	uint8_t* o;
	if (early.p && total <= early_cap) { o = static_cast<uint8_t*>(early.p); early.p = nullptr; }
	else {
		o = static_cast<uint8_t*>(host_out_alloc(total));
		if (early.p && label_bytes) memcpy(o + off_labels, static_cast<uint8_t*>(early.p) + off_labels, label_bytes);
	}
	try {
		e.last_codes_total = cr.total;
		if (head.label_format == PINS_VARIABLE_WIDTH) {
			if (label_bytes) memcpy(o + off_labels, pins_binary.data(), label_bytes);
			labels_crc = crc32c(o + off_labels, label_bytes);
		}
		std::vector<uint8_t> hb;
		head.write(hb);
		memcpy(o, hb.data(), hb.size());
		auto put4 = [&](uint64_t at, uint32_t v) { for (int b = 0; b < 4; b++) o[at + b] = static_cast<uint8_t>((v >> (8 * b)) & 0xFF); };
		for (int64_t z = 0; z < sz; z++) put4(off_index + 4ull * z, cr.code_len[z]);
		put4(off_index + 4ull * sz, crc32c(o + off_index, 4ull * sz));
		if (!stored_model.empty()) memcpy(o + off_model, stored_model.data(), stored_model.size());
		for (int64_t z = 0; z < sz; z++) put4(off_tail + 4 + 4ull * z, fr.crcs[z]);
		put4(off_tail, labels_crc);
		e.device_stream_bytes = 0;
		if (e.keep_device_stream) {
			// the same bytes once more in HBM, for a decoder that takes its stream from the device
			// (ckl_decoder_create_device): the two bulky sections are device-to-device copies, the small
			// ones go up from the host buffer
			e.d_stream_out.ensure(total + 16);
			uint8_t* ds = e.d_stream_out.p;
			const void* up = head.label_format == FLAT ? o : nullptr;      // (by kernel for flat label streams only, see upload_small; o: a pinned, device-mapped block when the stream is 64 KiB or more)
			upload_small(ds, o, off_labels, s, up);
			if (label_bytes) {
				if (head.label_format == FLAT) copy_bytes_device(e.d_labels_bin.p, ds + off_labels, label_bytes, s);
				else CKL_HIP(hipMemcpyAsync(ds + off_labels, o + off_labels, label_bytes, hipMemcpyHostToDevice, s));
			}
			if (!stored_model.empty()) upload_small(ds + off_model, o + off_model, stored_model.size(), s, up);
			if (cr.total) copy_bytes_device(e.d_codes_out.p, ds + off_codes, cr.total, s);
			upload_small(ds + off_tail, o + off_tail, 4ull * (sz + 1), s, up);
			if (labels_stay) {
				CKL_HIP(hipStreamWaitEvent(s, e.ev_labels_crc, 0));
				CKL_HIP(hipMemcpyAsync(ds + off_tail, e.d_labels_crc.p, 4, hipMemcpyDeviceToDevice, s));
			}
			e.device_stream_bytes = total;
		}
		// the crack codes' copy to the host goes LAST: started first, its 13 MB kept the link busy and the 2 KB header
		// upload above waited 0.19 ms behind it (rocprofv3 timeline, round 4)
		if (cr.total && !e.defer_codes) {
			if (e.async_host_copy && e.keep_device_stream) {
				// the caller goes on with the stream in HBM (a decoder, the next volume's planes) while the codes
				// cross PCIe: ckl_encoder_host_wait before the host bytes are read
				if (!e.stream_copy) CKL_HIP(hipStreamCreateWithFlags(&e.stream_copy, hipStreamNonBlocking));
				if (!e.ev_codes) CKL_HIP(hipEventCreateWithFlags(&e.ev_codes, hipEventDisableTiming));
				CKL_HIP(hipEventRecord(e.ev_codes, s));
				CKL_HIP(hipStreamWaitEvent(e.stream_copy, e.ev_codes, 0));
				CKL_HIP(hipMemcpyAsync(o + off_codes, e.d_codes_out.p, cr.total, hipMemcpyDeviceToHost, e.stream_copy));
				if (labels_stay) {
					CKL_HIP(hipMemcpyAsync(o + off_labels, e.d_labels_bin.p, label_bytes, hipMemcpyDeviceToHost, e.stream_copy));
					CKL_HIP(hipMemcpyAsync(o + off_tail, e.d_labels_crc.p, 4, hipMemcpyDeviceToHost, e.stream_copy));
				}
				e.host_copy_pending = true;
			}
			else CKL_HIP(hipMemcpyAsync(o + off_codes, e.d_codes_out.p, cr.total, hipMemcpyDeviceToHost, s));
		}
		ht.mark("assembly");
		CKL_HIP(hipEventRecord(e.ev1, s));
		CKL_HIP(hipStreamSynchronize(s));
		ht.mark("codes_d2h");
		CKL_HIP(hipGetLastError());
		CKL_HIP(hipEventElapsedTime(&e.pipeline_ms, e.ev0, e.ev1));
		CKL_HIP(hipEventElapsedTime(&e.trail_ms, e.evk0, e.evk1));
		CKL_HIP(hipEventElapsedTime(&e.dominant_ms, e.evd0, e.evd1));      // the encoder's longest kernel: k_trail_walk
	}
	catch (...) {
		// the codes' background copy may already be queued into `o`: it must be over before the block goes back to the cache
		if (e.host_copy_pending) { (void)hipStreamSynchronize(e.stream_copy); e.host_copy_pending = false; }
		host_out_free(o);
		throw;
	}
	*out = o;
	*out_len = total;
}

// reencode_with_markov_order (src/crackle.hpp:858-984).  The reference takes every slice's
// code apart into symbols and packs them again; here the decoder rasterises the codes into the
// crack planes and the encoder's trail writes them out again under the new order.  The trail is a
// function of the crack set alone, so the code points are the ones the reference's own encoder
// gives for these slices — which is what its reencode reproduces for any stream that encoder
// (or this one) wrote.  Header, z-index and model are rewritten, label section and crcs copied.
void reencode_markov(const uint8_t* buf, uint64_t n, int markov_order, int device, uint8_t** out, uint64_t* out_len) {
	if (n < Header::kBytesV0) throw Error(CKL_ERR_FORMAT, "crackle: Input too small to be a valid stream. Bytes: " + std::to_string(n));
	Header head = Header::parse(buf, n);
	if (markov_order < 0 || markov_order > 13) throw Error(CKL_ERR_ARG, "crackle_amd: markov_model_order must be in [0, 13] on device");
	auto copy_out = [&](const uint8_t* p, uint64_t len) {
		uint8_t* o = static_cast<uint8_t*>(host_out_alloc(len));
		memcpy(o, p, len);
		*out = o; *out_len = len;
	};
	if (head.markov_model_order == markov_order) { copy_out(buf, n); return; }      // crackle.hpp:887-889
	// Version 0 streams (24-byte header, no crcs) stay version 0, as src/crackle.hpp:947-984 means them to: the
	// reference's own output for them is broken — its header vector has 29 bytes of which 24 are written
	// (src/header.hpp:277-281), so five stray zero bytes follow the header (tests/golden/v0.npz keeps a sample) —
	// what is written here is the stream that code intends, which the reference reads back.
	const bool v0 = head.format_version == 0;
	const uint64_t sz = head.sz;
	const uint64_t tail_bytes = v0 ? 0 : 4 * (sz + 1);
	const uint64_t off_index = head.header_bytes();
	const uint64_t off_labels = off_index + head.grid_index_bytes();
	if (!head.layout_fits(n)) throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_code_offsets: Unable to read past end of buffer.");      // no sum of untrusted fields that could wrap
	const uint64_t old_codes = off_labels + head.num_label_bytes + head.markov_model_bytes();
	uint64_t old_tail = old_codes;
	for (uint64_t z = 0; z < sz; z++) old_tail += rd_le(buf + off_index + 4 * z, 4);
	if (old_tail > n || tail_bytes > n - old_tail) throw Error(CKL_ERR_RUNTIME, "crackle: get_crack_codes: Unable to read past end of buffer.");

	const bool prof = getenv("CKL_PROFILE") != nullptr;
	auto t_prev = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!prof) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[ckl reencode host ms] %s=%.2f\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
		t_prev = now;
	};
	struct DecoderGuard { ckl_decoder* d = nullptr; ~DecoderGuard() { if (d) ckl_decoder_destroy(d); } } dec;
	struct EncoderGuard { ckl_encoder* e = nullptr; ~EncoderGuard() { if (e) ckl_encoder_destroy(e); } } enc;
	if (ckl_decoder_create(buf, n, 0, -1, device, &dec.d) != CKL_OK) throw Error(CKL_ERR_RUNTIME, ckl_last_error());
	lap("decoder_create");
	CrackResult cr;
	std::vector<uint8_t> model, stored_model;
	head.markov_model_order = markov_order;
	if (head.voxels() > 0) {
		const uint32_t *cv = nullptr, *ch = nullptr;
		uint32_t row_words = 0;
		uint64_t plane_words = 0;
		if (ckl_decoder_crack_planes(dec.d, &cv, &ch, &row_words, &plane_words) != CKL_OK) throw Error(CKL_ERR_RUNTIME, ckl_last_error());
		lap("crack_planes");
		if (ckl_encoder_create(head.sx, head.sy, head.sz, 1, device, &enc.e) != CKL_OK) throw Error(CKL_ERR_RUNTIME, ckl_last_error());
		lap("encoder_create");
		ckl_encoder& e = *enc.e;
		hipStream_t s = e.stream;
		const uint32_t ns = head.sz;
		const bool permissible = head.crack_format == PERMISSIBLE;
		e.row_words = row_words;
		e.plane_words = plane_words;
		e.d_planes.ensure(2 * plane_words * ns);
		e.d_count_vh.ensure(4 * static_cast<size_t>(ns));
		CKL_HIP(hipMemsetAsync(e.d_count_vh.p, 0, 2 * static_cast<size_t>(ns) * sizeof(uint32_t), s));
		hipLaunchKernelGGL(k_planes_from_cracks, dim3(static_cast<uint32_t>((plane_words + kBlock - 1) / kBlock), ns), dim3(kBlock), 0, s,
			cv, ch, head.sx, head.sy, row_words, plane_words, permissible ? 1u : 0u,
			e.d_planes.p, e.d_planes.p + plane_words * ns, e.d_count_vh.p, ns);
		std::vector<uint32_t> c = download(e.d_count_vh.p, 2 * static_cast<size_t>(ns), s);
		e.count_v.assign(c.begin(), c.begin() + ns);
		e.count_h.assign(c.begin() + ns, c.end());
		lap("planes");
		graph_pass(e, head.sx, head.sy, head.sz, permissible);
		lap("graph");
		crack_pass(e, head.sx, head.sy, head.sz, permissible, markov_order, false, nullptr, nullptr, &model, &cr);
		lap("cracks");
		if (markov_order > 0) stored_model = markov_model_to_stored(model);
	}
	else cr.code_len.assign(sz, 0);

	// assembly (crackle.hpp:935-981)
	const uint64_t off_model = off_labels + head.num_label_bytes;
	const uint64_t off_codes = off_model + stored_model.size();
	const uint64_t off_tail = off_codes + cr.total;
	const uint64_t total = off_tail + tail_bytes;
	uint8_t* o = static_cast<uint8_t*>(host_out_alloc(total));
	try {
		if (cr.total) CKL_HIP(hipMemcpyAsync(o + off_codes, enc.e->d_codes_out.p, cr.total, hipMemcpyDeviceToHost, enc.e->stream));
		std::vector<uint8_t> hb;
		head.write(hb);
		memcpy(o, hb.data(), hb.size());
		auto put4 = [&](uint64_t at, uint32_t v) { for (int b = 0; b < 4; b++) o[at + b] = static_cast<uint8_t>((v >> (8 * b)) & 0xFF); };
		for (uint64_t z = 0; z < sz; z++) put4(off_index + 4 * z, cr.code_len[z]);
		if (!v0) put4(off_index + 4 * sz, crc32c(o + off_index, 4 * sz));
		memcpy(o + off_labels, buf + off_labels, head.num_label_bytes);
		if (!stored_model.empty()) memcpy(o + off_model, stored_model.data(), stored_model.size());
		if (tail_bytes) memcpy(o + off_tail, buf + old_tail, tail_bytes);
		if (cr.total) CKL_HIP(hipStreamSynchronize(enc.e->stream));
		lap("assembly");
	}
	catch (...) { host_out_free(o); throw; }
	*out = o;
	*out_len = total;
}

void check_dims(int64_t sx, int64_t sy, int64_t sz, int dtype_bytes, int is_signed) {
	if (is_signed) throw Error(CKL_ERR_ARG, "Signed integer data types are not currently supported.");
	if (dtype_bytes != 1 && dtype_bytes != 2 && dtype_bytes != 4 && dtype_bytes != 8) throw Error(CKL_ERR_ARG, "crackle_amd: dtype width must be 1, 2, 4 or 8 bytes");
	if (sx < 0 || sy < 0 || sz < 0) throw Error(CKL_ERR_ARG, "crackle_amd: negative dimension");
	if (sx > 0x7FFFFFF0ll || sy > 0x7FFFFFF0ll || sz > 0x7FFFFFF0ll) throw Error(CKL_ERR_ARG, "crackle_amd: dimension too large");
	if (static_cast<uint64_t>(sx + 1) * static_cast<uint64_t>(sy + 1) >= (1ull << 31)) throw Error(CKL_ERR_ARG, "crackle_amd: slices of 2^31 or more crack vertices are not supported");
}

}  // namespace

extern "C" {

int ckl_encoder_create(int64_t sx, int64_t sy, int64_t sz, int dtype_bytes, int device, ckl_encoder** out) {
	try {
		if (!out) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		check_dims(sx, sy, sz, dtype_bytes, 0);
		select_device(device);
		std::unique_ptr<ckl_encoder> e(new ckl_encoder());
		e->device = device;
		e->max_sx = sx; e->max_sy = sy; e->max_sz = sz; e->dtype_bytes = dtype_bytes;
		CKL_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
		{
			// the label stream ahead of the trail stream: its many small kernels should be through before
			// the serial walk (k_trail_walk, one wavefront per slice) starts, whose step time suffers beside them
			int lo = 0, hi = 0;
			CKL_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
			const char* env = getenv("CKL_LABEL_STREAM_PRIORITY");
			const int prio = env ? atoi(env) : hi;
			CKL_HIP(hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, prio));
		}
		CKL_HIP(hipEventCreate(&e->ev0));
		CKL_HIP(hipEventCreate(&e->ev1));
		CKL_HIP(hipEventCreate(&e->evk0));
		CKL_HIP(hipEventCreate(&e->evk1));
		CKL_HIP(hipEventCreate(&e->evd0));
		CKL_HIP(hipEventCreate(&e->evd1));
		CKL_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
		CKL_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
		CKL_HIP(hipEventCreateWithFlags(&e->ev_prezero, hipEventDisableTiming));
		for (auto& ev : e->ev_join) CKL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		for (auto& ev : e->ev_pre) CKL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		*out = e.release();
		return CKL_OK;
	}
	catch (const Error& e) { set_last_error(e.what()); return e.status; }
	catch (const std::exception& e) { set_last_error(e.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_run(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_model_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	const ckl_encode_overrides* overrides, uint8_t** out, uint64_t* out_len
) {
	try {
		if (!e || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		check_dims(sx, sy, sz, e->dtype_bytes, 0);
		if (markov_model_order > 15) throw Error(CKL_ERR_ARG, "crackle_amd: markov_model_order must be in [0, 15]");
		select_device(e->device);
		if (e->host_copy_pending) {      // the previous run's codes are still on their way out of d_codes_out
			CKL_HIP(hipStreamSynchronize(e->stream_copy));
			e->host_copy_pending = false;
		}
		wait_for_default_stream(e->stream, e->ev_in);
		wait_for_default_stream(e->stream2, e->ev_in);
		const auto t_run0 = std::chrono::steady_clock::now();
#define CKL_ENC(T) encode_typed<T>(*e, reinterpret_cast<const T*>(labels_device), sx, sy, sz, allow_pins != 0, fortran_order != 0, \
	markov_model_order, optimize_pins != 0, auto_bgcolor != 0, manual_bgcolor, overrides, out, out_len)
		if (e->dtype_bytes == 1) CKL_ENC(uint8_t);
		else if (e->dtype_bytes == 2) CKL_ENC(uint16_t);
		else if (e->dtype_bytes == 4) CKL_ENC(uint32_t);
		else CKL_ENC(uint64_t);
#undef CKL_ENC
		if (getenv("CKL_PROFILE")) {
			fprintf(stderr, "[ckl encoder_run ms] encode=%.2f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count());
		}
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_reencode_markov(const uint8_t* buf, uint64_t n, int markov_model_order, int device, uint8_t** out, uint64_t* out_len) {
	try {
		if (!buf || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(device);
		reencode_markov(buf, n, markov_model_order, device, out, out_len);
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_defer_codes(ckl_encoder* e, int defer) {
	if (!e) { set_last_error("crackle_amd: null encoder"); return CKL_ERR_ARG; }
	e->defer_codes = defer != 0;
	return CKL_OK;
}

int ckl_encoder_async_host_copy(ckl_encoder* e, int on) {
	if (!e) { set_last_error("crackle_amd: null encoder"); return CKL_ERR_ARG; }
	e->async_host_copy = on != 0;
	return CKL_OK;
}

int ckl_encoder_host_wait(ckl_encoder* e) {
	try {
		if (!e) throw Error(CKL_ERR_ARG, "crackle_amd: null encoder");
		if (e->host_copy_pending) {
			select_device(e->device);
			CKL_HIP(hipStreamSynchronize(e->stream_copy));
			e->host_copy_pending = false;
		}
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_keep_device_stream(ckl_encoder* e, int keep) {
	if (!e) { set_last_error("crackle_amd: null encoder"); return CKL_ERR_ARG; }
	e->keep_device_stream = keep != 0;
	return CKL_OK;
}

int ckl_encoder_device_stream(const ckl_encoder* e, const uint8_t** stream_device, uint64_t* n_bytes) {
	if (!e || !stream_device || !n_bytes) { set_last_error("crackle_amd: null argument"); return CKL_ERR_ARG; }
	if (!e->device_stream_bytes) { set_last_error("crackle_amd: no device-resident stream (ckl_encoder_keep_device_stream before the run)"); return CKL_ERR_ARG; }
	*stream_device = e->d_stream_out.p;
	*n_bytes = e->device_stream_bytes;
	return CKL_OK;
}

int ckl_encoder_codes_to_host(ckl_encoder* e, uint8_t* dst_host, uint64_t capacity, uint64_t* n_bytes) {
	try {
		if (!e || !n_bytes) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		*n_bytes = e->last_codes_total;
		if (e->last_codes_total == 0) return CKL_OK;
		if (!dst_host || capacity < e->last_codes_total) throw Error(CKL_ERR_ARG, "crackle_amd: code buffer too small: need " + std::to_string(e->last_codes_total) + " bytes");
		select_device(e->device);
		if (e->async_host_copy) {
			// (the run that left the codes has drained e->stream: nothing to order after)
			if (!e->stream_copy) CKL_HIP(hipStreamCreateWithFlags(&e->stream_copy, hipStreamNonBlocking));
			CKL_HIP(hipMemcpyAsync(dst_host, e->d_codes_out.p, e->last_codes_total, hipMemcpyDeviceToHost, e->stream_copy));
			e->host_copy_pending = true;
			return CKL_OK;
		}
		CKL_HIP(hipMemcpyAsync(dst_host, e->d_codes_out.p, e->last_codes_total, hipMemcpyDeviceToHost, e->stream));
		CKL_HIP(hipStreamSynchronize(e->stream));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_stats(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint64_t* max_label, uint64_t* pixel_pairs, uint64_t* first_voxel, uint64_t* last_voxel
) {
	try {
		if (!e) throw Error(CKL_ERR_ARG, "crackle_amd: null encoder");
		check_dims(sx, sy, sz, e->dtype_bytes, 0);
		select_device(e->device);
		wait_for_default_stream(e->stream, e->ev_in);
		wait_for_default_stream(e->stream2, e->ev_in);
		const uint64_t voxels = static_cast<uint64_t>(sx) * sy * sz;
		VolumeStats st;
		e->planes_for = nullptr;
		if (voxels > 0) {
			// the label-plane pass yields max / pairs from the same read and leaves the planes for
			// the ckl_encoder_run that follows; the slab's first and last voxel come back with its counts
			// (they were two blocking copies behind it)
			uint64_t f = 0, l = 0;
			const uint8_t* base = static_cast<const uint8_t*>(labels_device);
			// the two copies land in this frame: whatever leaves it early (planes_pass throws on an allocation or a HIP
			// error) waits for them first
			struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{ e->stream };
			CKL_HIP(hipMemcpyAsync(&f, base, e->dtype_bytes, hipMemcpyDeviceToHost, e->stream));
			CKL_HIP(hipMemcpyAsync(&l, base + (voxels - 1) * e->dtype_bytes, e->dtype_bytes, hipMemcpyDeviceToHost, e->stream));
			if (e->dtype_bytes == 1) planes_pass<uint8_t>(*e, reinterpret_cast<const uint8_t*>(labels_device), sx, sy, sz, &st);
			else if (e->dtype_bytes == 2) planes_pass<uint16_t>(*e, reinterpret_cast<const uint16_t*>(labels_device), sx, sy, sz, &st);
			else if (e->dtype_bytes == 4) planes_pass<uint32_t>(*e, reinterpret_cast<const uint32_t*>(labels_device), sx, sy, sz, &st);
			else planes_pass<uint64_t>(*e, reinterpret_cast<const uint64_t*>(labels_device), sx, sy, sz, &st);
			CKL_HIP(hipStreamSynchronize(e->stream));      // (planes_pass has waited for the stream already: this returns at once)
			st.first = f; st.last = l;
			e->planes_for = labels_device;
			e->planes_stats = st;
			e->planes_dims[0] = sx; e->planes_dims[1] = sy; e->planes_dims[2] = sz;
		}
		if (max_label) *max_label = st.max_label;
		if (pixel_pairs) *pixel_pairs = st.pairs;
		if (first_voxel) *first_voxel = st.first;
		if (last_voxel) *last_voxel = st.last;
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_markov_stats(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	int crack_format, uint64_t markov_model_order, uint32_t* hist
) {
	try {
		if (!e || !hist) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		check_dims(sx, sy, sz, e->dtype_bytes, 0);
		if (markov_model_order == 0 || markov_model_order > 13) throw Error(CKL_ERR_ARG, "crackle_amd: markov_model_order must be in [1, 13]");
		select_device(e->device);
		wait_for_default_stream(e->stream, e->ev_in);
		wait_for_default_stream(e->stream2, e->ev_in);
		const size_t rows = static_cast<size_t>(1) << (2 * markov_model_order);
		std::vector<uint32_t> h(rows * 4, 0);
		if (static_cast<uint64_t>(sx) * sy * sz > 0) {
			const bool perm = crack_format == PERMISSIBLE;
			const int order = static_cast<int>(markov_model_order);
			const bool cached = e->planes_for == labels_device && e->planes_dims[0] == sx && e->planes_dims[1] == sy && e->planes_dims[2] == sz;
			if (cached) {}
			else if (e->dtype_bytes == 1) planes_pass<uint8_t>(*e, reinterpret_cast<const uint8_t*>(labels_device), sx, sy, sz, nullptr);
			else if (e->dtype_bytes == 2) planes_pass<uint16_t>(*e, reinterpret_cast<const uint16_t*>(labels_device), sx, sy, sz, nullptr);
			else if (e->dtype_bytes == 4) planes_pass<uint32_t>(*e, reinterpret_cast<const uint32_t*>(labels_device), sx, sy, sz, nullptr);
			else planes_pass<uint64_t>(*e, reinterpret_cast<const uint64_t*>(labels_device), sx, sy, sz, nullptr);
			e->trail_for = nullptr;
			graph_pass(*e, sx, sy, sz, perm);
			crack_pass(*e, sx, sy, sz, perm, order, true, nullptr, &h, nullptr, nullptr);
			if (cached) { e->trail_for = labels_device; e->trail_perm = perm; e->trail_order = order; }      // (only beside cached planes: the run checks both)
		}
		memcpy(hist, h.data(), h.size() * sizeof(uint32_t));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

// shared by ckl_encoder_components / ckl_encoder_components_device: components of the slab, ids
// painted into cc_device (null: the session's own volume); returns where they are
static uint32_t* encoder_components(ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz, uint32_t id_base, uint32_t* cc_device, uint32_t* ncomp_host) {
	check_dims(sx, sy, sz, e->dtype_bytes, 0);
	select_device(e->device);
	wait_for_default_stream(e->stream, e->ev_in);
	wait_for_default_stream(e->stream2, e->ev_in);
	const uint64_t voxels = static_cast<uint64_t>(sx) * sy * sz;
	if (voxels == 0) return cc_device;
	const bool cached = e->planes_for == labels_device && e->planes_dims[0] == sx && e->planes_dims[1] == sy && e->planes_dims[2] == sz;
	FlatResult fr;
#define CKL_COMP(T) do { \
		if (!cached) planes_pass<T>(*e, reinterpret_cast<const T*>(labels_device), sx, sy, sz, nullptr); \
		flat_enqueue(*e, sx, sy, sz); \
		flat_collect<T>(*e, reinterpret_cast<const T*>(labels_device), sx, sy, sz, fr); \
	} while (0)
	if (e->dtype_bytes == 1) CKL_COMP(uint8_t);
	else if (e->dtype_bytes == 2) CKL_COMP(uint16_t);
	else if (e->dtype_bytes == 4) CKL_COMP(uint32_t);
	else CKL_COMP(uint64_t);
#undef CKL_COMP
	if (fr.total + id_base > 0xFFFFFFFFull) throw Error(CKL_ERR_RUNTIME, "crackle_amd: too many components");
	hipStream_t s2 = e->stream2;
	if (!cc_device) { e->d_cc_volume.ensure(voxels); cc_device = e->d_cc_volume.p; }
	launch_paint_components(s2, e->d_planes.p, e->row_words, e->plane_words, sx, sy, sz, e->d_word_base.p, e->d_rbase.p, e->d_run_cc.p, e->d_comp_off.p, id_base, cc_device);
	for (int64_t z = 0; z < sz; z++) ncomp_host[z] = fr.ncomp[z];
	return cc_device;
}

int ckl_encoder_components(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint32_t id_base, uint32_t* cc_host, uint32_t* ncomp_host
) {
	try {
		if (!e || !cc_host || !ncomp_host) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		const uint32_t* cc = encoder_components(e, labels_device, sx, sy, sz, id_base, nullptr, ncomp_host);
		const uint64_t voxels = static_cast<uint64_t>(sx) * sy * sz;
		if (voxels) CKL_HIP(hipMemcpyAsync(cc_host, cc, voxels * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream2));
		CKL_HIP(hipStreamSynchronize(e->stream2));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_components_device(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint32_t id_base, uint32_t* cc_device, uint32_t* ncomp_host
) {
	try {
		if (!e || !cc_device || !ncomp_host) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		encoder_components(e, labels_device, sx, sy, sz, id_base, cc_device, ncomp_host);
		CKL_HIP(hipStreamSynchronize(e->stream2));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_pin_labels(
	ckl_encoder* e, const void* labels_device, const uint32_t* cc_device,
	int64_t sx, int64_t sy, int64_t sz, const uint32_t* ncomp_host,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor,
	uint8_t** out, uint64_t* out_len
) {
	try {
		if (!e || !labels_device || !cc_device || !ncomp_host || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (sx <= 0 || sy <= 0 || sz <= 0) throw Error(CKL_ERR_ARG, "crackle_amd: empty volume");
		if (sx > 0xFFFFFFFFll || sy > 0xFFFFFFFFll || sz > 0xFFFFFFFFll) throw Error(CKL_ERR_ARG, "crackle_amd: dimensions must fit 32 bits");
		if (stored_width != 1 && stored_width != 2 && stored_width != 4 && stored_width != 8) throw Error(CKL_ERR_ARG, "crackle_amd: stored width must be 1, 2, 4 or 8 bytes");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		std::vector<uint32_t> nc(ncomp_host, ncomp_host + sz);
		uint64_t N = 0;
		for (uint32_t c : nc) N += c;
		if (N == 0 || N > 0xFFFFFFFFull) throw Error(CKL_ERR_ARG, "crackle_amd: component counts out of range");
		Header h;
		h.sx = static_cast<uint32_t>(sx); h.sy = static_cast<uint32_t>(sy); h.sz = static_cast<uint32_t>(sz);
		hipStream_t s = e->stream2;
		const uint64_t voxels = static_cast<uint64_t>(sx) * sy * sz;
		// label of every component, read where the id changes along x (every component has such a voxel)
		e->d_mapping.ensure(N + 1);
		e->d_slice_err2.ensure(1);
		CKL_HIP(hipMemsetAsync(e->d_mapping.p, 0, N * sizeof(uint64_t), s));
		CKL_HIP(hipMemsetAsync(e->d_slice_err2.p, 0, sizeof(uint32_t), s));
		std::vector<uint8_t> bin;
		const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((voxels + kPinBlock - 1) / kPinBlock, 0x7FFFFFFFull));
#define CKL_PINS(T) do { \
			hipLaunchKernelGGL(k_pin_component_labels<T>, dim3(blocks), dim3(kPinBlock), 0, s, reinterpret_cast<const T*>(labels_device), cc_device, voxels, static_cast<uint32_t>(sx), N, \
				reinterpret_cast<unsigned long long*>(e->d_mapping.p), e->d_slice_err2.p); \
			if (download(e->d_slice_err2.p, 1, s)[0]) throw Error(CKL_ERR_ARG, "crackle_amd: component id out of range"); \
			bin = pins_section_plain<T>(*e, reinterpret_cast<const T*>(labels_device), cc_device, e->d_mapping.p, sx, sy, sz, N, nc, h.pin_index_width(), stored_width, auto_bgcolor != 0, manual_bgcolor); \
		} while (0)
		if (e->dtype_bytes == 1) CKL_PINS(uint8_t);
		else if (e->dtype_bytes == 2) CKL_PINS(uint16_t);
		else if (e->dtype_bytes == 4) CKL_PINS(uint32_t);
		else CKL_PINS(uint64_t);
#undef CKL_PINS
		uint8_t* p = static_cast<uint8_t*>(host_out_alloc(bin.size() ? bin.size() : 1));
		memcpy(p, bin.data(), bin.size());
		*out = p;
		*out_len = bin.size();
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}


#define CKL_ROWS_DISPATCH(CALL) do { \
		if (e->dtype_bytes == 1) { typedef uint8_t T; CALL; } \
		else if (e->dtype_bytes == 2) { typedef uint16_t T; CALL; } \
		else if (e->dtype_bytes == 4) { typedef uint32_t T; CALL; } \
		else { typedef uint64_t T; CALL; } \
	} while (0)

static void pins_rows_check(const ckl_encoder* e, const void* labels, const uint32_t* cc, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t N) {
	if (!e || !labels || !cc) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
	if (sx <= 0 || rows <= 0 || sz <= 0 || y0 < 0) throw Error(CKL_ERR_ARG, "crackle_amd: empty row slab");
	if (sx > 0x7FFFFFF0ll || rows > 0x7FFFFFF0ll || y0 > 0x7FFFFFF0ll || sz > 65535) throw Error(CKL_ERR_ARG, "crackle_amd: row slab dimensions out of range");
	if (N == 0 || N >= kPinNone) throw Error(CKL_ERR_ARG, "crackle_amd: component count out of range");
	if (static_cast<unsigned __int128>(y0 + rows) * static_cast<uint64_t>(sx) * static_cast<uint64_t>(sz) >= (static_cast<unsigned __int128>(1) << 47)) throw Error(CKL_ERR_ARG, "crackle_amd: volume too large for the row-sharded pin stage");
}

int ckl_pins_rows_first(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t n_components,
	uint64_t* first_any, uint64_t* first_kept, uint64_t* comp_label) {
	try {
		pins_rows_check(e, labels_rows, cc_rows, sx, rows, sz, y0, n_components);
		if (!first_any || !first_kept || !comp_label) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		CKL_ROWS_DISPATCH(pins_rows_first<T>(*e, static_cast<const T*>(labels_rows), cc_rows, sx, rows, sz, y0, n_components, first_any, first_kept, comp_label));
		CKL_HIP(hipStreamSynchronize(e->stream2));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_pins_rows_best(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t n_components,
	const uint64_t* first_kept, uint64_t* best) {
	try {
		pins_rows_check(e, labels_rows, cc_rows, sx, rows, sz, y0, n_components);
		if (!first_kept || !best) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		CKL_ROWS_DISPATCH(pins_rows_best<T>(*e, static_cast<const T*>(labels_rows), cc_rows, sx, rows, sz, y0, n_components, first_kept, best));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_pins_rows_extent(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t n_components,
	const uint64_t* first_kept, const uint64_t* best, uint64_t* choice, uint32_t* ze_plus1) {
	try {
		pins_rows_check(e, labels_rows, cc_rows, sx, rows, sz, y0, n_components);
		if (!first_kept || !best || !choice || !ze_plus1) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		CKL_ROWS_DISPATCH(pins_rows_extent<T>(*e, static_cast<const T*>(labels_rows), cc_rows, sx, rows, sz, y0, n_components, first_kept, best, choice, ze_plus1));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_pins_rows_ids(ckl_encoder* e, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0, uint64_t n_components,
	const uint64_t* choice, const uint32_t* ze_plus1, const uint64_t* offsets, uint32_t* ids) {
	try {
		if (!e || !cc_rows || !choice || !ze_plus1 || !offsets || !ids) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (sx <= 0 || rows <= 0 || sz <= 0 || y0 < 0 || n_components == 0 || n_components >= kPinNone) throw Error(CKL_ERR_ARG, "crackle_amd: row slab out of range");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		hipStream_t s = e->stream2;
		PinVolume v;
		v.sx = static_cast<uint32_t>(sx); v.sy = static_cast<uint32_t>(rows); v.sz = static_cast<uint32_t>(sz);
		v.sxy = static_cast<uint64_t>(v.sx) * v.sy; v.key_col0 = static_cast<uint64_t>(y0) * v.sx; v.cc = cc_rows; v.mark = nullptr;
		// k_pin_ids wants the last slice itself: the entries that are 0 ("not mine") are skipped by the row check, so ze_plus1 - 1 of the others
		DevBuf<uint32_t> d_ze;
		d_ze.ensure(n_components);
		const uint32_t nb = static_cast<uint32_t>((n_components + kPinBlock - 1) / kPinBlock);
		hipLaunchKernelGGL(k_pin_minus1, dim3(nb), dim3(kPinBlock), 0, s, ze_plus1, n_components, d_ze.p);
		hipLaunchKernelGGL(k_pin_ids, dim3(nb), dim3(kPinBlock), 0, s, v, reinterpret_cast<const unsigned long long*>(choice), d_ze.p, offsets, static_cast<uint32_t>(n_components), ids);
		CKL_HIP(hipStreamSynchronize(s));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_pins_rows_section(ckl_encoder* e, int64_t sx, int64_t sy, int64_t sz, uint64_t n_components, const uint32_t* ncomp_host,
	const uint64_t* comp_label, const uint64_t* first_any, const uint64_t* choice, const uint32_t* ze_plus1, const uint64_t* offsets, const uint32_t* ids,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor, uint8_t** out, uint64_t* out_len) {
	try {
		if (!e || !ncomp_host || !comp_label || !first_any || !choice || !ze_plus1 || !offsets || !ids || !out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (sx <= 0 || sy <= 0 || sz <= 0) throw Error(CKL_ERR_ARG, "crackle_amd: empty volume");
		if (stored_width != 1 && stored_width != 2 && stored_width != 4 && stored_width != 8) throw Error(CKL_ERR_ARG, "crackle_amd: stored width must be 1, 2, 4 or 8 bytes");
		const uint64_t N = n_components;
		if (N == 0 || N >= kPinNone) throw Error(CKL_ERR_ARG, "crackle_amd: component count out of range");
		select_device(e->device);
		wait_for_default_stream(e->stream2, e->ev_in);
		std::vector<uint32_t> nc(ncomp_host, ncomp_host + sz);
		HostTimer ht;
		g_ht = &ht;
		const std::vector<uint8_t> bin = pins_section_from_device(*e, sx, sy, sz, N, nc, comp_label, first_any, choice, ze_plus1, offsets, ids, stored_width, auto_bgcolor != 0, manual_bgcolor);
		ht.mark("p:cover");
		uint8_t* p = static_cast<uint8_t*>(host_out_alloc(bin.size() ? bin.size() : 1));
		memcpy(p, bin.data(), bin.size());
		*out = p;
		*out_len = bin.size();
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

int ckl_encoder_last_timing(const ckl_encoder* e, float* pipeline_ms, float* dominant_kernel_ms) {
	if (!e) { set_last_error("crackle_amd: null encoder"); return CKL_ERR_ARG; }
	if (pipeline_ms) *pipeline_ms = e->pipeline_ms;
	if (dominant_kernel_ms) *dominant_kernel_ms = e->dominant_ms;
	return CKL_OK;
}

int ckl_encoder_walk_paths(ckl_encoder* e, uint32_t* fast_slices, uint32_t* compiled_slices) {
	try {
		if (!e) throw Error(CKL_ERR_ARG, "crackle_amd: null encoder");
		uint32_t fast = 0, plain = 0;
		const size_t ns = e->last_trail_slices;
		if (ns && e->t_counters.p) {
			select_device(e->device);
			std::vector<uint32_t> nev(ns);
			CKL_HIP(hipMemcpy(nev.data(), e->t_counters.p + 5 * ns, ns * sizeof(uint32_t), hipMemcpyDeviceToHost));      // n_events, flagged with the event format
			for (uint32_t v : nev) { if (v & (kEvFormatAddr12 | kEvWalkWide)) fast++; else if (v) plain++; }
		}
		if (fast_slices) *fast_slices = fast;
		if (compiled_slices) *compiled_slices = plain;
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

// grid = slices, block = kBlock: the walk's events of a slice by kind, and the longest run of events that take no decision
// (a node with one remaining edge is left by it, a dead end returns: kEvSeg / kEvDead / kEvEnd; only kEvBseg picks an edge
// AND leaves one behind).  counts: [slices][5] = seg, bseg, dead, end, longest run without a kEvBseg
__global__ void __launch_bounds__(kBlock) k_trail_step_kinds(const uint32_t* __restrict__ events, const uint64_t* __restrict__ ibase, const uint32_t* __restrict__ n_events, uint32_t* __restrict__ counts) {
	__shared__ uint32_t s_c[5];
	const uint32_t zi = blockIdx.x;
	if (threadIdx.x < 5) s_c[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t n = n_events[zi] & ~(dev::kEvFormatAddr12 | dev::kEvWalkWide);
	const uint32_t* ev = events + ibase[zi];
	uint32_t c[4] = { 0, 0, 0, 0 };
	for (uint32_t i = threadIdx.x; i < n; i += kBlock) c[ev[i] >> 30]++;
	for (int k = 0; k < 4; k++) if (c[k]) atomicAdd(&s_c[k], c[k]);
	if (threadIdx.x == 0) {      // (one thread: a diagnostic, not a product path)
		uint32_t run = 0, longest = 0;
		for (uint32_t i = 0; i < n; i++) { if ((ev[i] >> 30) == 1u) run = 0; else { run++; longest = run > longest ? run : longest; } }
		s_c[4] = longest;
	}
	__syncthreads();
	if (threadIdx.x < 5) counts[zi * 5u + threadIdx.x] = s_c[threadIdx.x];
}

int ckl_encoder_walk_step_kinds(ckl_encoder* e, uint32_t* counts, uint32_t max_slices, uint32_t* n_slices) {
	try {
		if (!e || !counts || !n_slices) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		const size_t ns = e->last_trail_slices;
		*n_slices = static_cast<uint32_t>(ns);
		if (!ns || !e->t_counters.p || !e->t_events.p) { *n_slices = 0; return CKL_OK; }
		if (ns > max_slices) throw Error(CKL_ERR_ARG, "crackle_amd: counts holds fewer slices than the last run had");
		select_device(e->device);
		DevBuf<uint32_t> d_counts;
		d_counts.ensure(ns * 5);
		hipLaunchKernelGGL(k_trail_step_kinds, dim3(static_cast<uint32_t>(ns)), dim3(kBlock), 0, e->stream, e->t_events.p, e->t_ibase.p, e->t_counters.p + 5 * ns, d_counts.p);
		CKL_HIP(hipMemcpyAsync(counts, d_counts.p, ns * 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
		CKL_HIP(hipStreamSynchronize(e->stream));
		return CKL_OK;
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); return CKL_ERR_RUNTIME; }
}

void ckl_encoder_destroy(ckl_encoder* e) { delete e; }

int ckl_compress(
	const void* labels, int labels_mem, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_model_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	int device, uint8_t** out, uint64_t* out_len
) {
	ckl_encoder* e = nullptr;
	try {
		check_dims(sx, sy, sz, dtype_bytes, is_signed);
		if (!out || !out_len) throw Error(CKL_ERR_ARG, "crackle_amd: null argument");
		if (static_cast<uint64_t>(sx) * sy * sz == 0) {
			// crackle.hpp:96-98: an empty volume is just the 29-byte header; nothing to compute
			Header head;
			head.crack_format = IMPERMISSIBLE;              // pixel_pairs 0 < 0 / 2 is false (crackle.hpp:50-55)
			head.label_format = (sz == 1 || !allow_pins) ? FLAT : PINS_VARIABLE_WIDTH;
			head.data_width = dtype_bytes;
			head.stored_data_width = 1;                     // max_label of nothing is 0
			head.sx = static_cast<uint32_t>(sx); head.sy = static_cast<uint32_t>(sy); head.sz = static_cast<uint32_t>(sz);
			head.fortran_order = fortran_order != 0;
			head.markov_model_order = static_cast<int>(markov_model_order & 0xFF);
			std::vector<uint8_t> bin;
			head.write(bin);
			uint8_t* p = static_cast<uint8_t*>(malloc(bin.size()));
			if (!p) throw Error(CKL_ERR_RUNTIME, "crackle_amd: out of host memory");
			memcpy(p, bin.data(), bin.size());
			*out = p;
			*out_len = bin.size();
			return CKL_OK;
		}
	}
	catch (const Error& err) { set_last_error(err.what()); return err.status; }
	int rc = ckl_encoder_create(sx, sy, sz, dtype_bytes, device, &e);
	if (rc != CKL_OK) return rc;
	try {
		const uint64_t bytes = static_cast<uint64_t>(sx) * sy * sz * dtype_bytes;
		const void* dev_labels = labels;
		DevBuf<uint8_t> tmp;
		if (labels_mem == CKL_MEM_HOST && bytes) {
			if (!labels) throw Error(CKL_ERR_ARG, "crackle_amd: null labels");
			tmp.ensure(bytes);
			CKL_HIP(hipMemcpy(tmp.p, labels, bytes, hipMemcpyHostToDevice));
			dev_labels = tmp.p;
		}
		rc = ckl_encoder_run(e, dev_labels, sx, sy, sz, allow_pins, fortran_order, markov_model_order,
			optimize_pins, auto_bgcolor, manual_bgcolor, nullptr, out, out_len);
		ckl_encoder_destroy(e);
		return rc;
	}
	catch (const Error& err) { set_last_error(err.what()); ckl_encoder_destroy(e); return err.status; }
	catch (const std::exception& err) { set_last_error(err.what()); ckl_encoder_destroy(e); return CKL_ERR_RUNTIME; }
}

}  // extern "C"
