// Device-side building blocks shared by the decode and encode kernels (gfx950,
// wave64).  Block size is fixed at 256 threads = 4 wavefronts everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace ckl {
namespace dev {

// Tuning builds (-DCKL_TUNING, `python -m crackle_amd.build --tuning` -> libcrackle_amd_tuning.so) keep the
// in-kernel cycle stamps (CKL_*_DIAG) and the ablation switches (CKL_ABLATE: parts of kernels skipped,
// results wrong); the shipped library compiles both out.
#ifdef CKL_TUNING
constexpr bool kTuning = true;
#else
constexpr bool kTuning = false;
#endif

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kWaves = kBlock / kWave;

// ---- wavefront / block scans ---------------------------------------------------
// Scans inside a wavefront run on the DPP data path (row_shr 1, 2, 4, 8 inside the rows of 16 lanes,
// then row_bcast:15 / row_bcast:31 across them: the sequence LLVM's own atomic optimizer emits for
// gfx9 wave64): seven VALU instructions.  hipcc lowers __shfl_up to ds_bpermute_b32, a round trip
// through the LDS crossbar per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
	return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v), CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppRowShr = 0x110, kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;      // wave_shl:1 = every lane takes the value of the lane above it
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
	v += dpp_u32<kDppRowShr + 1, 0xF>(0u, v);
	v += dpp_u32<kDppRowShr + 2, 0xF>(0u, v);
	v += dpp_u32<kDppRowShr + 4, 0xF>(0u, v);
	v += dpp_u32<kDppRowShr + 8, 0xF>(0u, v);
	v += dpp_u32<kDppBcast15, 0xA>(0u, v);
	v += dpp_u32<kDppBcast31, 0xC>(0u, v);
	return v;
}
__device__ __forceinline__ int32_t wave_incl_max(int32_t v) {
	auto mx = [](int32_t a, uint32_t b) { const int32_t c = static_cast<int32_t>(b); return c > a ? c : a; };
	constexpr uint32_t id = 0x80000000u;      // INT32_MIN
	v = mx(v, dpp_u32<kDppRowShr + 1, 0xF>(id, static_cast<uint32_t>(v)));
	v = mx(v, dpp_u32<kDppRowShr + 2, 0xF>(id, static_cast<uint32_t>(v)));
	v = mx(v, dpp_u32<kDppRowShr + 4, 0xF>(id, static_cast<uint32_t>(v)));
	v = mx(v, dpp_u32<kDppRowShr + 8, 0xF>(id, static_cast<uint32_t>(v)));
	v = mx(v, dpp_u32<kDppBcast15, 0xA>(id, static_cast<uint32_t>(v)));
	v = mx(v, dpp_u32<kDppBcast31, 0xC>(id, static_cast<uint32_t>(v)));
	return v;
}
// value of the lane below (lane 0 gets `first`)
__device__ __forceinline__ uint32_t wave_shift_up1(uint32_t v, uint32_t first) { return dpp_u32<kDppWaveShr1, 0xF>(first, v); }
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
	for (int d = kWave / 2; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, kWave);
	return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
	for (int d = kWave / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
	return v;
}

// Exclusive block scan (sum) of K values per thread at once.  `lds` must hold
// K * NW uint32 (NW = wavefronts per workgroup).  Returns exclusive prefixes in v[], block totals in total[].
// Ends with a barrier so `lds` can be reused immediately.
// Workgroups of eight and sixteen wavefronts combine the wavefronts' totals with a second DPP scan over lanes 0 .. NW - 1
// (one LDS read per value and thread; reading all NW totals into registers was 48 live registers for K = 3, NW = 16: the
// 1 024-thread k_crack_match spilled over it).
template <int K, int NW = kWaves>
__device__ __forceinline__ void block_excl_add(uint32_t (&v)[K], uint32_t (&total)[K], uint32_t* lds) {
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = threadIdx.x >> 6;
	uint32_t incl[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		incl[k] = wave_incl_add(v[k]);
		if (lane == kWave - 1) lds[k * NW + wave] = incl[k];
	}
	__syncthreads();
	if constexpr (NW >= 8 && NW <= 16) {
		const int wsel = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
		for (int k = 0; k < K; k++) {
			uint32_t t = lds[k * NW + (lane & (NW - 1))];      // (lanes NW .. 63 repeat the pattern: rows of 16 lanes scan on their own)
			t += dpp_u32<kDppRowShr + 1, 0xF>(0u, t);
			t += dpp_u32<kDppRowShr + 2, 0xF>(0u, t);
			t += dpp_u32<kDppRowShr + 4, 0xF>(0u, t);
			if constexpr (NW == 16) t += dpp_u32<kDppRowShr + 8, 0xF>(0u, t);
			const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(t), NW - 1));
			const uint32_t base = wsel ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(t), wsel - 1)) : 0u;
			v[k] = base + incl[k] - v[k];
			total[k] = tot;
		}
	}
	else {
#pragma unroll
		for (int k = 0; k < K; k++) {
			uint32_t base = 0, tot = 0;
#pragma unroll
			for (int w = 0; w < NW; w++) {
				uint32_t s = lds[k * NW + w];
				if (w < wave) base += s;
				tot += s;
			}
			v[k] = base + incl[k] - v[k];
			total[k] = tot;
		}
	}
	__syncthreads();
}

// Exclusive block max-scan of one int32 (identity INT32_MIN).  lds: NW int32.
template <int NW = kWaves>
__device__ __forceinline__ int32_t block_excl_max(int32_t v, int32_t& total, int32_t* lds) {
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = threadIdx.x >> 6;
	const int32_t incl = wave_incl_max(v);
	if (lane == kWave - 1) lds[wave] = incl;
	const int32_t excl = static_cast<int32_t>(wave_shift_up1(static_cast<uint32_t>(incl), 0x80000000u));      // lane 0: INT32_MIN
	__syncthreads();
	int32_t base = INT32_MIN, tot = INT32_MIN;
	if constexpr (NW >= 8 && NW <= 16) {      // (as in block_excl_add)
		auto mx = [](int32_t a, uint32_t b) { const int32_t c = static_cast<int32_t>(b); return c > a ? c : a; };
		constexpr uint32_t id = 0x80000000u;
		const int wsel = __builtin_amdgcn_readfirstlane(wave);
		int32_t t = lds[lane & (NW - 1)];
		t = mx(t, dpp_u32<kDppRowShr + 1, 0xF>(id, static_cast<uint32_t>(t)));
		t = mx(t, dpp_u32<kDppRowShr + 2, 0xF>(id, static_cast<uint32_t>(t)));
		t = mx(t, dpp_u32<kDppRowShr + 4, 0xF>(id, static_cast<uint32_t>(t)));
		if constexpr (NW == 16) t = mx(t, dpp_u32<kDppRowShr + 8, 0xF>(id, static_cast<uint32_t>(t)));
		tot = __builtin_amdgcn_readlane(t, NW - 1);
		if (wsel) base = __builtin_amdgcn_readlane(t, wsel - 1);
	}
	else {
#pragma unroll
		for (int w = 0; w < NW; w++) {
			int32_t s = lds[w];
			if (w < wave) base = s > base ? s : base;
			tot = s > tot ? s : tot;
		}
	}
	total = tot;
	__syncthreads();
	return excl > base ? excl : base;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* lds) {
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = threadIdx.x >> 6;
	v = wave_sum(v);
	if (lane == 0) lds[wave] = v;
	__syncthreads();
	uint32_t tot = 0;
#pragma unroll
	for (int w = 0; w < kWaves; w++) tot += lds[w];
	__syncthreads();
	return tot;
}
__device__ __forceinline__ uint32_t block_xor(uint32_t v, uint32_t* lds) {
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = threadIdx.x >> 6;
	v = wave_xor(v);
	if (lane == 0) lds[wave] = v;
	__syncthreads();
	uint32_t tot = 0;
#pragma unroll
	for (int w = 0; w < kWaves; w++) tot ^= lds[w];
	__syncthreads();
	return tot;
}

// ---- CRC-32C in GF(2)[x]/P, reflected (bit 31 <-> x^0) -------------------------
constexpr uint32_t kCrcPoly = 0x82F63B78u;
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
	uint32_t r = 0;
#pragma unroll 8
	for (int i = 0; i < 32; i++) {
		r ^= (a & (0x80000000u >> i)) ? b : 0u;
		b = (b >> 1) ^ ((b & 1u) ? kCrcPoly : 0u);
	}
	return r;
}

// Per-tile CRC machinery.  A tile is kCrcTile consecutive u32 words of one
// slice's component image; thread j folds words j, j+256, ... with the Horner
// step acc = acc * x^(32*256) ^ word, the multiplication done with four
// 256-entry tables staged in LDS (crc_stride_tab, 4 KiB, computed on the host).
constexpr int kCrcTile = 1024;                 // words per tile
constexpr int kCrcRows = kCrcTile / kBlock;    // words per thread
__device__ __forceinline__ uint32_t crc_stride_step(const uint32_t* tab /* LDS [4][256] */, uint32_t acc) {
	return tab[acc & 0xFF] ^ tab[256 + ((acc >> 8) & 0xFF)] ^ tab[512 + ((acc >> 16) & 0xFF)] ^ tab[768 + (acc >> 24)];
}

// ---- union-find on a per-slice u32 parent array (root = smallest index) ---------
__device__ __forceinline__ uint32_t uf_load(const uint32_t* L, uint32_t i) {
	return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t uf_find(const uint32_t* L, uint32_t a) {
	for (;;) {
		uint32_t p = uf_load(L, a);
		if (p == a) return a;
		a = p;
	}
}
// Link the trees of a and b.  Parents only ever decrease, and the link is an
// atomicMin on a node believed to be a root: if it no longer is, the returned
// value is its newer (smaller) parent and the merge continues from there.
__device__ __forceinline__ void uf_unite(uint32_t* L, uint32_t a, uint32_t b) {
	for (;;) {
		a = uf_find(L, a);
		b = uf_find(L, b);
		if (a == b) return;
		if (a > b) { uint32_t t = a; a = b; b = t; }
		uint32_t old = atomicMin(L + b, a);
		if (old == b) return;
		b = old;
	}
}

}  // namespace dev
}  // namespace ckl
