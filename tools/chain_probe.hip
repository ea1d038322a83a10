// chain_probe.hip -- the chain walk of oklab2_kernel<64> alone: one wave, 12 live lanes (or 64), 512 dependent adds fed by
// ds_read_b128, in the kernel's own loop form.  hipcc --offload-arch=gfx950 -O3 -o chain_probe.bin chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld(const float4 *p)
{
	const f32x4_t t = *(const volatile __attribute__((address_space(3))) f32x4_t *)p;
	return make_float4(t.x, t.y, t.z, t.w);
}
template <int LIVE, bool WAITS_ONCE>
__global__ void __launch_bounds__(1024) k_walk(uint64_t *out, float seed, int reps)
{
	__shared__ __attribute__((aligned(16))) float buf[12 * 516];
	for (int i = threadIdx.x; i < 12 * 516; i += blockDim.x) buf[i] = seed + i;
	__syncthreads();
	if (threadIdx.x >= 64) return;
	const uint32_t lane = threadIdx.x;
	float sum = 0.f;
	uint64_t t0 = clock64();
	if (lane < LIVE) {
		for (int rep = 0; rep < reps; ++rep) {
			const float4 *x = reinterpret_cast<const float4 *>(buf + (lane % 12) * 516);
			float4 va[8], vb[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) va[q] = ld(x + q);
#pragma unroll 1
			for (int r = 0; r < 8; ++r) {
#pragma unroll
				for (int q = 0; q < 8; ++q) vb[q] = ld(x + 8 + q);
				__builtin_amdgcn_sched_barrier(0);
				if (WAITS_ONCE) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#pragma unroll
				for (int q = 0; q < 8; ++q) { sum += va[q].x; sum += va[q].y; sum += va[q].z; sum += va[q].w; }
				asm volatile("" : "+v"(sum));
				__builtin_amdgcn_sched_barrier(0);
				const float4 *xn = r + 1 < 8 ? x + 16 : x - 16 * r;
#pragma unroll
				for (int q = 0; q < 8; ++q) va[q] = ld(xn + q);
				__builtin_amdgcn_sched_barrier(0);
				if (WAITS_ONCE) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#pragma unroll
				for (int q = 0; q < 8; ++q) { sum += vb[q].x; sum += vb[q].y; sum += vb[q].z; sum += vb[q].w; }
				asm volatile("" : "+v"(sum));
				__builtin_amdgcn_sched_barrier(0);
				x = xn;
			}
		}
	}
	uint64_t t1 = clock64();
	if (threadIdx.x == 0) out[0] = t1 - t0;
	if (sum == 12345.f) out[1] = 1;
}

// adds only: 64 registers loaded once, 512 dependent adds per walk
__global__ void __launch_bounds__(1024) k_adds_only(uint64_t *out, float seed, int reps)
{
	if (threadIdx.x >= 64) return;
	float v[64];
#pragma unroll
	for (int i = 0; i < 64; ++i) v[i] = seed * (i + 1);
	float sum = 0.f;
	uint64_t t0 = clock64();
	for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
		for (int r = 0; r < 8; ++r) {
#pragma unroll
			for (int i = 0; i < 64; ++i) sum += v[i];
			asm volatile("" : "+v"(sum));
		}
	}
	uint64_t t1 = clock64();
	if (threadIdx.x == 0) out[0] = t1 - t0;
	if (sum == 12345.f) out[1] = 1;
}
// hand-placed loads: 8 ds_read_b128 per asm statement, ONE counted wait per 32 adds
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD8(A, base, off) asm volatile( \
	"ds_read_b128 %0, %8 offset:" #off "+0\n\tds_read_b128 %1, %8 offset:" #off "+16\n\tds_read_b128 %2, %8 offset:" #off "+32\n\tds_read_b128 %3, %8 offset:" #off "+48\n\t" \
	"ds_read_b128 %4, %8 offset:" #off "+64\n\tds_read_b128 %5, %8 offset:" #off "+80\n\tds_read_b128 %6, %8 offset:" #off "+96\n\tds_read_b128 %7, %8 offset:" #off "+112" \
	: "=v"(A[0]), "=v"(A[1]), "=v"(A[2]), "=v"(A[3]), "=v"(A[4]), "=v"(A[5]), "=v"(A[6]), "=v"(A[7]) : "v"(base))
#define WAIT8(A, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]))
template <int LIVE>
__global__ void __launch_bounds__(1024) k_walk_asm(uint64_t *out, float seed, int reps)
{
	__shared__ __attribute__((aligned(16))) float buf[12 * 516];
	for (int i = threadIdx.x; i < 12 * 516; i += blockDim.x) buf[i] = seed + i;
	__syncthreads();
	if (threadIdx.x >= 64) return;
	const uint32_t lane = threadIdx.x;
	float sum = 0.f;
	uint64_t t0 = clock64();
	if (lane < LIVE) {
		for (int rep = 0; rep < reps; ++rep) {
			uint32_t base = (uint32_t)(uintptr_t)(buf + (lane % 12) * 516);  // LDS byte address (low 32 bits of the generic pointer)
			f32x4 va[8], vb[8];
			LD8(va, base, 0);
#pragma unroll 1
			for (int r = 0; r < 8; ++r) {
				LD8(vb, base, 128);
				WAIT8(va, 8);
#pragma unroll
				for (int q = 0; q < 8; ++q) { sum += va[q].x; sum += va[q].y; sum += va[q].z; sum += va[q].w; }
				asm volatile("" : "+v"(sum));
				base += r + 1 < 8 ? 256u : (uint32_t)(-256 * 7);
				LD8(va, base, 0);
				WAIT8(vb, 8);
#pragma unroll
				for (int q = 0; q < 8; ++q) { sum += vb[q].x; sum += vb[q].y; sum += vb[q].z; sum += vb[q].w; }
				asm volatile("" : "+v"(sum));
			}
			WAIT8(va, 0);
		}
	}
	uint64_t t1 = clock64();
	if (threadIdx.x == 0) out[0] = t1 - t0;
	if (sum == 12345.f) out[1] = 1;
}
template <class K>
static void run(const char *name, K k, int waves)
{
	uint64_t *d;
	(void)hipMalloc(&d, 16);
	(void)hipMemset(d, 0, 16);
	const int reps = 200;
	for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.5f, reps);
	(void)hipDeviceSynchronize();
	uint64_t h[2];
	(void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("%-28s waves %2d: %8.2f ticks per add (512 adds + 128 ds_read_b128 per walk: %.0f ticks per walk)\n", name, waves, (double)h[0] / (reps * 512.0), (double)h[0] / reps);
	(void)hipFree(d);
}
int main()
{
	run("walk 12 lanes", k_walk<12, false>, 1);
	run("walk 64 lanes", k_walk<64, false>, 1);
	run("walk 12 lanes, one wait", k_walk<12, true>, 1);
	run("walk 64 lanes, one wait", k_walk<64, true>, 1);
	run("adds only", k_adds_only, 1);
	run("asm loads 12 lanes", k_walk_asm<12>, 1);
	run("asm loads 64 lanes", k_walk_asm<64>, 1);
	run("asm loads 12 lanes, 16 waves", k_walk_asm<12>, 16);
	return 0;
}
