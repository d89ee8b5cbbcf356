// Micro-benchmark: what does ONE wavefront pay per dependent step on MI355X?  (the encoder's serial
// trail walk is one wavefront per slice: its step time is a sum of these)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wave_latency.hip -o /tmp/wave_latency && /tmp/wave_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 4096;

// out[0] = cycles (s_memtime), out[1] = 100 MHz ticks, out[2] = sink
#define T0 const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define T1(sink) if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime() - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0; out[2] = (sink); }

__global__ void __launch_bounds__(64) k_salu_chain(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed);
	T0
	for (int i = 0; i < N / 16; i++) {
		asm volatile(
			"s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
			"s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
			"s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
			"s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
			: "+s"(x) : : "scc");
	}
	T1(x)
}
__global__ void __launch_bounds__(64) k_valu_chain(unsigned long long* out, uint32_t seed) {
	uint32_t x = seed + threadIdx.x;
	T0
	for (int i = 0; i < N / 16; i++) {
		asm volatile(
			"v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
			"v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
			"v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
			"v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
			: "+v"(x));
	}
	T1(x)
}
// SALU -> VALU -> readfirstlane -> SALU round trip without memory
__global__ void __launch_bounds__(64) k_sv_roundtrip(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed);
	T0
	for (int i = 0; i < N / 4; i++) {
		uint32_t v;
		asm volatile(
			"v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n"
			: "+s"(x), "=&v"(v) : : "scc");
	}
	T1(x)
}
// pointer chase through LDS: address from an SGPR, value back to an SGPR
__global__ void __launch_bounds__(64) k_lds_chase(unsigned long long* out, uint32_t seed, int lane0_only) {
	__shared__ uint32_t s[4096];
	for (int i = threadIdx.x; i < 4096; i += 64) s[i] = ((i * 1237u + 101u) & 4095u) * 4u;
	__syncthreads();
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 4095u) * 4u;
	if (lane0_only && threadIdx.x) return;
	T0
	for (int i = 0; i < N / 4; i++) {
		uint32_t v;
		asm volatile(
			"v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n"
			"v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n"
			"v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n"
			"v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n"
			: "+s"(x), "=&v"(v) : : "memory", "scc");
	}
	T1(x + s[0])
}
// the same chase kept in VGPRs (per-lane addresses, no SGPR round trip)
__global__ void __launch_bounds__(64) k_lds_chase_v(unsigned long long* out, uint32_t seed) {
	__shared__ uint32_t s[4096];
	for (int i = threadIdx.x; i < 4096; i += 64) s[i] = ((i * 1237u + 101u) & 4095u) * 4u;
	__syncthreads();
	uint32_t x = ((seed + threadIdx.x) & 4095u) * 4u;
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"
			"ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"
			: "+v"(x) : : "memory");
	}
	T1(x + s[0])
}
// write then dependent read (what a step of the walk does: clear an edge bit, then load the next node)
__global__ void __launch_bounds__(64) k_lds_write_read(unsigned long long* out, uint32_t seed) {
	__shared__ uint32_t s[4096];
	for (int i = threadIdx.x; i < 4096; i += 64) s[i] = ((i * 1237u + 101u) & 4095u) * 4u;
	__syncthreads();
	uint32_t x = ((seed + threadIdx.x) & 4095u) * 4u;
	T0
	for (int i = 0; i < N / 2; i++) {
		uint32_t v;
		asm volatile(
			"ds_read_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %0, %1\n v_mov_b32 %0, %1\n"
			"ds_read_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %0, %1\n v_mov_b32 %0, %1\n"
			: "+v"(x), "=&v"(v) : : "memory");
	}
	T1(x + s[0])
}
// pointer chase through a VGPR with v_readlane (64-entry table in one register)
__global__ void __launch_bounds__(64) k_readlane_chase(unsigned long long* out, uint32_t seed) {
	uint32_t tab = (threadIdx.x * 37u + 11u) & 63u;
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 63u);
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"s_nop 3\n v_readlane_b32 %0, %1, %0\n s_nop 3\n v_readlane_b32 %0, %1, %0\n"
			"s_nop 3\n v_readlane_b32 %0, %1, %0\n s_nop 3\n v_readlane_b32 %0, %1, %0\n"
			: "+s"(x) : "v"(tab) : "scc");
	}
	T1(x)
}
// taken branches
__global__ void __launch_bounds__(64) k_branches(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed);
	T0
	for (int i = 0; i < N / 8; i++) {
		asm volatile(
			"s_add_u32 %0, %0, 1\n s_branch 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"1: s_add_u32 %0, %0, 1\n s_branch 2f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"2: s_add_u32 %0, %0, 1\n s_branch 3f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"3: s_add_u32 %0, %0, 1\n s_branch 4f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"4: s_add_u32 %0, %0, 1\n s_branch 5f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"5: s_add_u32 %0, %0, 1\n s_branch 6f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"6: s_add_u32 %0, %0, 1\n s_branch 7f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
			"7: s_add_u32 %0, %0, 1\n"
			: "+s"(x) : : "scc");
	}
	T1(x)
}
// pointer chase through the scalar cache
__global__ void __launch_bounds__(64) k_smem_chase(unsigned long long* out, const uint32_t* tab, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 4095u) * 4u;
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			"s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			: "+s"(x) : "s"(tab) : "memory", "scc");
	}
	T1(x)
}
// pointer chase through the vector L1 / L2 (address from an SGPR base + VGPR offset)
template <bool SC1>
__global__ void __launch_bounds__(64) k_vmem_chase(unsigned long long* out, const uint32_t* tab, uint32_t seed) {
	uint32_t x = ((seed + threadIdx.x) & 4095u) * 4u;
	T0
	for (int i = 0; i < N / 4; i++) {
		if (SC1) asm volatile(
			"global_load_dword %0, %0, %1 sc1\n s_waitcnt vmcnt(0)\n global_load_dword %0, %0, %1 sc1\n s_waitcnt vmcnt(0)\n"
			"global_load_dword %0, %0, %1 sc1\n s_waitcnt vmcnt(0)\n global_load_dword %0, %0, %1 sc1\n s_waitcnt vmcnt(0)\n"
			: "+v"(x) : "s"(tab) : "memory");
		else asm volatile(
			"global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n"
			"global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n"
			: "+v"(x) : "s"(tab) : "memory");
	}
	T1(x)
}

// ---- pieces of the register-table walk (trail_walk_slice_regs_asm)
// a chase through a table held in registers: index -> s_set_gpr_idx_on, v_readlane, off -> next index
__global__ void __launch_bounds__(64) k_gpridx_chase(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 1023u);
	const uint32_t lane = threadIdx.x;
	T0
	asm volatile(
		// 16 table registers v64..v79: entry of node n (lane n & 63 of register n >> 6) = (n * 37 + 11) & 1023
		"s_mov_b32 s50, 0\n"
		"10:\n"
		"s_lshl_b32 s51, s50, 6\n"
		"v_add_u32 v20, s51, %[lane]\n"
		"v_mul_u32_u24 v20, 37, v20\n"
		"v_add_u32 v20, 11, v20\n"
		"v_and_b32 v20, 1023, v20\n"
		"s_set_gpr_idx_on s50, gpr_idx(DST)\n"
		"v_mov_b32 v64, v20\n"
		"s_set_gpr_idx_off\n"
		"s_add_u32 s50, s50, 1\n"
		"s_cmp_lt_u32 s50, 16\n"
		"s_cbranch_scc1 10b\n"
		"s_mov_b32 s52, 0\n"
		"1:\n"
		"s_lshr_b32 s50, %[x], 6\n"
		"s_and_b32 s51, %[x], 63\n"
		"s_set_gpr_idx_on s50, gpr_idx(SRC0)\n"
		"v_readlane_b32 %[x], v64, s51\n"
		"s_set_gpr_idx_off\n"
		"s_add_u32 s52, s52, 1\n"
		"s_cmp_lt_u32 s52, 4096\n"
		"s_cbranch_scc1 1b\n"
		: [x] "+s"(x) : [lane] "v"(lane)
		: "scc", "s50", "s51", "s52", "v20", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
	T1(x)
}
// the mode switch alone: on / off pairs in a dependent SALU chain
__global__ void __launch_bounds__(64) k_gpridx_onoff(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 7u);
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"s_set_gpr_idx_on %0, gpr_idx(SRC0)\n s_set_gpr_idx_off\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 7\n"
			"s_set_gpr_idx_on %0, gpr_idx(SRC0)\n s_set_gpr_idx_off\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 7\n"
			"s_set_gpr_idx_on %0, gpr_idx(SRC0)\n s_set_gpr_idx_off\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 7\n"
			"s_set_gpr_idx_on %0, gpr_idx(SRC0)\n s_set_gpr_idx_off\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 7\n"
			: "+s"(x) : : "scc", "m0");
	}
	T1(x)
}
// EXEC narrowed to one lane for one VALU write, then back
__global__ void __launch_bounds__(64) k_exec_one_lane(unsigned long long* out, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed & 63u);
	uint32_t v = threadIdx.x;
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"s_mov_b64 s[62:63], exec\n"
			"s_lshl_b64 exec, 1, %0\n v_mov_b32 %1, %0\n s_mov_b64 exec, s[62:63]\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 63\n"
			"s_lshl_b64 exec, 1, %0\n v_mov_b32 %1, %0\n s_mov_b64 exec, s[62:63]\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 63\n"
			"s_lshl_b64 exec, 1, %0\n v_mov_b32 %1, %0\n s_mov_b64 exec, s[62:63]\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 63\n"
			"s_lshl_b64 exec, 1, %0\n v_mov_b32 %1, %0\n s_mov_b64 exec, s[62:63]\n s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 63\n"
			: "+s"(x), "+v"(v) : : "scc", "s62", "s63");
	}
	T1(x + v)
}
// a fire-and-forget store per step beside a dependent SALU chain
__global__ void __launch_bounds__(64) k_store_per_step(unsigned long long* out, uint32_t* sink, uint32_t seed) {
	uint32_t x = __builtin_amdgcn_readfirstlane(seed);
	uint32_t off = 0;
	T0
	for (int i = 0; i < N / 4; i++) {
		asm volatile(
			"v_mov_b32 v30, %0\n global_store_dword %1, v30, %2\n v_add_u32 %1, 4, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 v30, %0\n global_store_dword %1, v30, %2\n v_add_u32 %1, 4, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 v30, %0\n global_store_dword %1, v30, %2\n v_add_u32 %1, 4, %1\n s_add_u32 %0, %0, 1\n"
			"v_mov_b32 v30, %0\n global_store_dword %1, v30, %2\n v_add_u32 %1, 4, %1\n s_add_u32 %0, %0, 1\n"
			: "+s"(x), "+v"(off) : "s"(sink) : "scc", "v30", "memory");
	}
	T1(x)
}

int main() {
	unsigned long long* out; uint32_t* tab;
	CK(hipMalloc(&out, 64)); CK(hipMalloc(&tab, 16384));
	uint32_t h[4096];
	for (int i = 0; i < 4096; i++) h[i] = ((i * 1237u + 101u) & 4095u) * 4u;
	CK(hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice));
	auto report = [&](const char* name, int steps) {
		hipDeviceSynchronize();
		unsigned long long r[3];
		hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
		printf("%-58s %7.1f cycles/step  (%.2f GHz, %.1f ns/step)\n", name, double(r[0]) / steps, double(r[0]) / (r[1] * 10.0), r[1] * 10.0 / steps);
		fflush(stdout);
	};
	for (int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(k_salu_chain, dim3(1), dim3(64), 0, 0, out, 1u); report("dependent SALU op", N);
		hipLaunchKernelGGL(k_valu_chain, dim3(1), dim3(64), 0, 0, out, 1u); report("dependent VALU op", N);
		hipLaunchKernelGGL(k_sv_roundtrip, dim3(1), dim3(64), 0, 0, out, 1u); report("v_mov <- s; v_readfirstlane; s_add", N);
		hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, out, 1u, 0); report("LDS chase via SGPR (v_mov, ds_read, readfirstlane)", N);
		hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, out, 1u, 1); report("LDS chase via SGPR, lane 0 only", N);
		hipLaunchKernelGGL(k_lds_chase_v, dim3(1), dim3(64), 0, 0, out, 1u); report("LDS chase in a VGPR (ds_read, waitcnt)", N);
		hipLaunchKernelGGL(k_lds_write_read, dim3(1), dim3(64), 0, 0, out, 1u); report("LDS read, write, dependent read", N);
		hipLaunchKernelGGL(k_readlane_chase, dim3(1), dim3(64), 0, 0, out, 1u); report("v_readlane chase (s_nop 3 + v_readlane)", N);
		hipLaunchKernelGGL(k_branches, dim3(1), dim3(64), 0, 0, out, 1u); report("s_add + taken s_branch (over 18 instructions)", N - N / 8);
		hipLaunchKernelGGL(k_smem_chase, dim3(1), dim3(64), 0, 0, out, tab, 1u); report("scalar-cache chase (s_load_dword)", N);
		hipLaunchKernelGGL(k_vmem_chase<false>, dim3(1), dim3(64), 0, 0, out, tab, 1u); report("vector L1 chase (global_load_dword)", N);
		hipLaunchKernelGGL(k_vmem_chase<true>, dim3(1), dim3(64), 0, 0, out, tab, 1u); report("L2 chase (global_load_dword sc1)", N);
	}
	{
		uint32_t* sink; CK(hipMalloc(&sink, 65536 * 4));
		for (int rep = 0; rep < 2; rep++) {
			hipLaunchKernelGGL(k_gpridx_chase, dim3(1), dim3(64), 0, 0, out, 1u); report("register-table chase (idx on, v_readlane, idx off)", 4096);
			hipLaunchKernelGGL(k_gpridx_onoff, dim3(1), dim3(64), 0, 0, out, 1u); report("s_set_gpr_idx_on + off + 2 SALU", N);
			hipLaunchKernelGGL(k_exec_one_lane, dim3(1), dim3(64), 0, 0, out, 1u); report("exec = 1 << lane, v_mov, exec back, 2 SALU", N);
			hipLaunchKernelGGL(k_store_per_step, dim3(1), dim3(64), 0, 0, out, sink, 1u); report("v_mov, global_store_dword, v_add, s_add", N);
		}
	}
	// the same with 512 wavefronts in flight, one per workgroup (how the walk runs)
	hipLaunchKernelGGL(k_lds_chase, dim3(512), dim3(64), 0, 0, out, 1u, 1); report("LDS chase via SGPR, lane 0 only, 512 workgroups", N);
	hipLaunchKernelGGL(k_salu_chain, dim3(512), dim3(64), 0, 0, out, 1u); report("dependent SALU op, 512 workgroups", N);
	return 0;
}
