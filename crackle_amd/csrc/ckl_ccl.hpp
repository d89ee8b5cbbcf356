// 4-connected component labelling of z-slices on device, shared by decode (the
// connectivity comes from the rasterised crack planes) and encode (connectivity
// = label equality).  Replaces cc3d::color_connectivity_graph (src/cc3d.hpp:146-254)
// and cc3d::connected_components2d_4 (src/cc3d.hpp:257-369).
//
// Canonical numbering (src/cc3d.hpp:114-144): component ids
// are 0..N-1 per slice in the raster order of each component's first pixel.  We get
// it with a union-find whose roots are the smallest pixel index of each component,
// then rank the roots in raster order with a prefix sum.
//
// Pipeline per batch of slices (all kernels: block = 256 threads):
//   k_ccl_rows    L[p] = first pixel of p's horizontal run          (ballot + carry)
//   k_ccl_merge   unite vertically adjacent runs                    (atomicMin links)
//   k_ccl_flatten L[p] = root(p); count roots per 1024-pixel tile
//   k_ccl_scan    exclusive scan of the tile counts of each slice -> tile offsets, N
//   k_ccl_rank    R[root] = rank of the root within its slice
// The consumer then reads cc(p) = R[L[p]].
#pragma once

#include "ckl_device.hpp"

namespace ckl {
namespace dev {

constexpr int kCclTile = 1024;   // pixels per block in the flat (non-row) kernels

// Connectivity from the crack planes written by the crack-code rasteriser.
// planeV bit (x,y): a crack edge lies between pixels (x-1,y) and (x,y).
// planeH bit (x,y): a crack edge lies between pixels (x,y-1) and (x,y).
// IMPERMISSIBLE streams: crack = boundary; PERMISSIBLE: crack = connection.
struct PlaneConn {
	const uint32_t* planeV;
	const uint32_t* planeH;
	uint32_t row_words;     // words per image row (rows are word aligned)
	uint64_t plane_words;   // words per slice
	uint32_t flip;          // 1 for IMPERMISSIBLE (connected = !bit)
	__device__ __forceinline__ bool left(uint32_t zi, int x, int y) const {
		uint32_t w = planeV[zi * plane_words + static_cast<uint64_t>(y) * row_words + (x >> 5)];
		return (((w >> (x & 31)) & 1u) ^ flip) != 0;
	}
	__device__ __forceinline__ bool up(uint32_t zi, int x, int y) const {
		uint32_t w = planeH[zi * plane_words + static_cast<uint64_t>(y) * row_words + (x >> 5)];
		return (((w >> (x & 31)) & 1u) ^ flip) != 0;
	}
};

// Connectivity = equal labels (encode side).
template <typename LABEL>
struct LabelConn {
	const LABEL* labels;    // x-fastest volume
	uint64_t sxy;
	int sx;
	__device__ __forceinline__ bool left(uint32_t zi, int x, int y) const {
		const LABEL* s = labels + zi * sxy + static_cast<uint64_t>(y) * sx + x;
		return s[0] == s[-1];
	}
	__device__ __forceinline__ bool up(uint32_t zi, int x, int y) const {
		const LABEL* s = labels + zi * sxy + static_cast<uint64_t>(y) * sx + x;
		return s[0] == s[-sx];
	}
};

// grid = (sy, nslices)
template <typename Conn>
__global__ void __launch_bounds__(kBlock) k_ccl_rows(Conn conn, uint32_t* __restrict__ L, int sx, int sy) {
	__shared__ int s_wave_last[kWaves];
	__shared__ int s_carry;
	const int y = blockIdx.x;
	const uint32_t zi = blockIdx.y;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	uint32_t* Lz = L + zi * sxy;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (int x0 = 0; x0 < sx; x0 += kBlock) {
		const int x = x0 + threadIdx.x;
		const bool valid = x < sx;
		const bool brk = valid && (x == 0 || !conn.left(zi, x, y));
		const unsigned long long m = __ballot(brk);
		const unsigned long long below = m & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
		int start = below ? (x0 + wave * 64 + (63 - __clzll(below))) : -1;
		if (lane == 63) s_wave_last[wave] = m ? (x0 + wave * 64 + (63 - __clzll(m))) : -1;
		__syncthreads();
		if (start < 0) {
			int c = s_carry;
			for (int w = 0; w < wave; w++) if (s_wave_last[w] >= 0) c = s_wave_last[w];
			start = c;
		}
		if (valid) Lz[static_cast<uint64_t>(y) * sx + x] = static_cast<uint32_t>(y) * sx + start;
		__syncthreads();
		if (threadIdx.x == 0) {
			int c = s_carry;
			for (int w = 0; w < kWaves; w++) if (s_wave_last[w] >= 0) c = s_wave_last[w];
			s_carry = c;
		}
		__syncthreads();
	}
}

// grid = (tiles, nslices): unite p with p - sx at the first pixel of every stretch
// along which both the horizontal run and the upward connection continue.
template <typename Conn>
__global__ void __launch_bounds__(kBlock) k_ccl_merge(Conn conn, uint32_t* __restrict__ L, int sx, int sy) {
	const uint32_t zi = blockIdx.y;
	const uint64_t sxy = static_cast<uint64_t>(sx) * sy;
	uint32_t* Lz = L + zi * sxy;
#pragma unroll
	for (int i = 0; i < kCclTile / kBlock; i++) {
		const uint64_t p = static_cast<uint64_t>(blockIdx.x) * kCclTile + i * kBlock + threadIdx.x;
		if (p >= sxy) continue;
		const int y = static_cast<int>(p / sx);
		const int x = static_cast<int>(p - static_cast<uint64_t>(y) * sx);
		if (y == 0) continue;
		if (!conn.up(zi, x, y)) continue;
		if (x > 0 && conn.left(zi, x, y) && conn.up(zi, x - 1, y)) continue;
		uf_unite(Lz, static_cast<uint32_t>(p), static_cast<uint32_t>(p - sx));
	}
}

// grid = (tiles, nslices)
static __global__ void __launch_bounds__(kBlock) k_ccl_flatten(uint32_t* __restrict__ L, uint32_t* __restrict__ tile_count, uint64_t sxy, uint32_t tiles) {
	__shared__ uint32_t s_red[kWaves];
	const uint32_t zi = blockIdx.y;
	uint32_t* Lz = L + zi * sxy;
	uint32_t cnt = 0;
#pragma unroll
	for (int i = 0; i < kCclTile / kBlock; i++) {
		const uint64_t p = static_cast<uint64_t>(blockIdx.x) * kCclTile + i * kBlock + threadIdx.x;
		if (p < sxy) {
			const uint32_t r = uf_find(Lz, static_cast<uint32_t>(p));
			Lz[p] = r;
			cnt += (r == static_cast<uint32_t>(p));
		}
	}
	const uint32_t tot = block_sum(cnt, s_red);
	if (threadIdx.x == 0) tile_count[static_cast<uint64_t>(zi) * tiles + blockIdx.x] = tot;
}

// grid = (nslices): tile_count -> exclusive offsets in place, total -> ncomp[zi]
static __global__ void __launch_bounds__(kBlock) k_ccl_scan(uint32_t* __restrict__ tile_count, uint32_t* __restrict__ ncomp, uint32_t tiles) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.x;
	uint32_t* tc = tile_count + static_cast<uint64_t>(zi) * tiles;
	uint32_t carry = 0;
	for (uint32_t t0 = 0; t0 < tiles; t0 += kBlock) {
		const uint32_t t = t0 + threadIdx.x;
		uint32_t v[1] = { t < tiles ? tc[t] : 0u };
		uint32_t tot[1];
		block_excl_add<1>(v, tot, s_scan);
		if (t < tiles) tc[t] = carry + v[0];
		carry += tot[0];
	}
	if (threadIdx.x == 0) ncomp[zi] = carry;
}

// grid = (tiles, nslices): R[root] = tile offset + rank of the root inside its tile.
// Tile pixels are visited in raster order: thread j owns pixels 4j .. 4j+3.
static __global__ void __launch_bounds__(kBlock) k_ccl_rank(const uint32_t* __restrict__ L, uint32_t* __restrict__ R, const uint32_t* __restrict__ tile_off, uint64_t sxy, uint32_t tiles) {
	__shared__ uint32_t s_scan[kWaves];
	const uint32_t zi = blockIdx.y;
	const uint32_t* Lz = L + zi * sxy;
	uint32_t* Rz = R + zi * sxy;
	constexpr int kPer = kCclTile / kBlock;
	const uint64_t p0 = static_cast<uint64_t>(blockIdx.x) * kCclTile + static_cast<uint64_t>(threadIdx.x) * kPer;
	bool root[kPer];
	uint32_t c = 0;
#pragma unroll
	for (int i = 0; i < kPer; i++) {
		const uint64_t p = p0 + i;
		root[i] = (p < sxy) && (Lz[p] == static_cast<uint32_t>(p));
		c += root[i];
	}
	uint32_t v[1] = { c }, tot[1];
	block_excl_add<1>(v, tot, s_scan);
	uint32_t r = tile_off[static_cast<uint64_t>(zi) * tiles + blockIdx.x] + v[0];
#pragma unroll
	for (int i = 0; i < kPer; i++) {
		if (root[i]) Rz[p0 + i] = r++;
	}
}

}  // namespace dev
}  // namespace ckl
