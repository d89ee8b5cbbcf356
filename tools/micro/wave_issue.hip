// Micro-benchmark 2: issue cost of instruction mixes for ONE wavefront on MI355X (see wave_latency.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wave_issue.hip -o /tmp/wave_issue && /tmp/wave_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int N = 4096;
#define T0 const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define T1(sink) if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime() - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0; out[2] = (sink); }
#define REP4(x) x x x x
#define KERNEL(name, body, ...) \
__global__ void __launch_bounds__(64) name(unsigned long long* out, uint32_t* mem, uint32_t seed) { \
	__shared__ uint32_t lds[4096]; \
	for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 1237u + 101u) & 1023u) * 16u; \
	__syncthreads(); \
	uint32_t x = __builtin_amdgcn_readfirstlane(seed), y = x + 1, v = seed + threadIdx.x, w = v; \
	if (threadIdx.x) return; \
	T0 \
	for (int i = 0; i < N / 4; i++) asm volatile(REP4(body) : "+s"(x), "+s"(y), "+v"(v), "+v"(w) : "s"(mem) : "memory", "scc", "s40", "s41", "s42", "s43", "v20", "v21", "v22", "v23"); \
	T1(x + y + lds[1]) \
}
// independent SALU ops (two chains)
KERNEL(k_salu_indep, "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n")
// SALU and VALU alternating, independent
KERNEL(k_salu_valu, "s_add_u32 %0, %0, 1\n v_add_u32 %2, 1, %2\n")
// compare + not-taken branch
KERNEL(k_not_taken, "s_add_u32 %0, %0, 1\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n 1:\n")
// compare + select
KERNEL(k_cselect, "s_cmp_lg_u32 %0, 0\n s_cselect_b32 %0, %0, 7\n")
// readfirstlane -> dependent SALU
KERNEL(k_rfl_salu, "v_readfirstlane_b32 %0, %2\n s_add_u32 %1, %0, %1\n")
// SALU -> v_mov -> VALU
KERNEL(k_salu_vmov, "s_add_u32 %0, %0, 1\n v_mov_b32 %2, %0\n")
// a store per step
KERNEL(k_store, "v_mov_b32 v20, %0\n global_store_dword %3, v20, %4\n v_add_u32 %3, 4, %3\n s_add_u32 %0, %0, 1\n")
// LDS: read b128 + b64, write b32 + b64 (the walk's LDS traffic), address through an SGPR
KERNEL(k_lds_mix, "v_mov_b32 v20, %0\n ds_read_b128 v[20:23], v20\n ds_read_b64 v[22:23], %2 offset:8192\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, v20\n v_mov_b32 v21, %0\n ds_write_b32 v21, v21 offset:4\n ds_write_b64 v21, v[22:23] offset:8200\n")
// the same without the writes
KERNEL(k_lds_reads, "v_mov_b32 v20, %0\n ds_read_b128 v[20:23], v20\n ds_read_b64 v[22:23], %2 offset:8192\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, v20\n")
// 10 dependent SALU + taken loop branch
KERNEL(k_salu10_branch, "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_branch 1f\n s_nop 0\n 1:\n")

int main() {
	unsigned long long* out; uint32_t* mem;
	CK(hipMalloc(&out, 64)); CK(hipMalloc(&mem, 1 << 20)); CK(hipMemset(mem, 0, 1 << 20));
	auto report = [&](const char* name, int steps, int instr) {
		(void)hipDeviceSynchronize();
		unsigned long long r[3];
		(void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
		printf("%-60s %7.1f cycles/step = %5.1f per instruction  (%.2f GHz)\n", name, double(r[0]) / steps, double(r[0]) / steps / instr, double(r[0]) / (r[1] * 10.0));
		fflush(stdout);
	};
	for (int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(k_salu_indep, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("2 independent SALU", N, 2);
		hipLaunchKernelGGL(k_salu_valu, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("SALU + VALU, independent", N, 2);
		hipLaunchKernelGGL(k_not_taken, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("s_add, s_cmp, branch not taken", N, 3);
		hipLaunchKernelGGL(k_cselect, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("s_cmp, s_cselect (dependent)", N, 2);
		hipLaunchKernelGGL(k_rfl_salu, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("v_readfirstlane, dependent s_add", N, 2);
		hipLaunchKernelGGL(k_salu_vmov, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("s_add, dependent v_mov", N, 2);
		hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("v_mov, global_store, v_add, s_add", N, 4);
		hipLaunchKernelGGL(k_lds_reads, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("LDS: read b128 + b64 via SGPR", N, 5);
		hipLaunchKernelGGL(k_lds_mix, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("LDS: the same + write b32 + b64", N, 8);
		hipLaunchKernelGGL(k_salu10_branch, dim3(1), dim3(64), 0, 0, out, mem, 16u); report("9 dependent SALU + taken branch", N, 10);
	}
	return 0;
}
