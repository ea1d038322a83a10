// issue_probe.hip -- issue cost (cycles per instruction, one wave alone on a SIMD) of the vector instructions the
// Oklab conversion is made of, independent and dependent streams.  hipcc --offload-arch=gfx950 -O2 -o issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// independent stream: 8 destination registers round-robin
#define PROBE(NAME, BODY, CLOBBER_INIT)                                                                       \
	__global__ void NAME(uint64_t *out, float seed)                                                           \
	{                                                                                                         \
		double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7; \
		float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7;   \
		int i0 = 1, i1 = 2, i2 = 3, i3 = 4;                                                                   \
		CLOBBER_INIT;                                                                                         \
		uint64_t t0 = clock64();                                                                              \
		for (int it = 0; it < 256; ++it) {                                                                    \
			asm volatile(REP64(BODY) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
			             "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)); \
		}                                                                                                     \
		uint64_t t1 = clock64();                                                                              \
		if (threadIdx.x == 0) out[0] = t1 - t0;                                                               \
		if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + i0 + i1 + i2 + i3 == 12345.f) out[1] = 1; \
	}

// operands: %0-%7 doubles, %8-%15 floats, %16-%19 ints
PROBE(k_fma_f64_ind, "v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %4, %5, %6, %7\n v_fma_f64 %1, %2, %3, %0\n v_fma_f64 %5, %6, %7, %4\n", )
PROBE(k_fma_f64_dep, "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n", )
PROBE(k_mul_f64_ind, "v_mul_f64 %0, %1, %2\n v_mul_f64 %4, %5, %6\n v_mul_f64 %3, %2, %1\n v_mul_f64 %7, %6, %5\n", )
PROBE(k_add_f64_ind, "v_add_f64 %0, %1, %2\n v_add_f64 %4, %5, %6\n v_add_f64 %3, %2, %1\n v_add_f64 %7, %6, %5\n", )
PROBE(k_rcp_f64_ind, "v_rcp_f64 %0, %1\n v_rcp_f64 %2, %3\n v_rcp_f64 %4, %5\n v_rcp_f64 %6, %7\n", )
PROBE(k_rcp_f64_dep, "v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n", )
PROBE(k_rcp_f32_ind, "v_rcp_f32 %8, %9\n v_rcp_f32 %10, %11\n v_rcp_f32 %12, %13\n v_rcp_f32 %14, %15\n", )
PROBE(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n", )
PROBE(k_cvt_f32_f64, "v_cvt_f32_f64 %8, %0\n v_cvt_f32_f64 %9, %1\n v_cvt_f32_f64 %10, %2\n v_cvt_f32_f64 %11, %3\n", )
PROBE(k_frexp_mant, "v_frexp_mant_f32 %8, %9\n v_frexp_mant_f32 %10, %11\n v_frexp_mant_f32 %12, %13\n v_frexp_mant_f32 %14, %15\n", )
PROBE(k_frexp_exp, "v_frexp_exp_i32_f32 %16, %9\n v_frexp_exp_i32_f32 %17, %11\n v_frexp_exp_i32_f32 %18, %13\n v_frexp_exp_i32_f32 %19, %15\n", )
PROBE(k_pk_mul_f32, "v_pk_mul_f32 %0, %1, %2\n v_pk_mul_f32 %4, %5, %6\n v_pk_mul_f32 %3, %2, %1\n v_pk_mul_f32 %7, %6, %5\n", )
PROBE(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %3\n v_pk_fma_f32 %4, %5, %6, %7\n v_pk_fma_f32 %1, %2, %3, %0\n v_pk_fma_f32 %5, %6, %7, %4\n", )
PROBE(k_add_f32_ind, "v_add_f32 %8, %9, %10\n v_add_f32 %11, %12, %13\n v_add_f32 %14, %15, %9\n v_add_f32 %10, %12, %13\n", )
PROBE(k_add_f32_dep, "v_add_f32 %8, %8, %9\n v_add_f32 %8, %8, %10\n v_add_f32 %8, %8, %11\n v_add_f32 %8, %8, %12\n", )
PROBE(k_add_f32_dep_abs, "v_add_f32 %8, %8, |%9|\n v_add_f32 %8, %8, |%10|\n v_add_f32 %8, %8, |%11|\n v_add_f32 %8, %8, |%12|\n", )
PROBE(k_add_f32_dep2, "v_add_f32 %8, %8, %9\n v_add_f32 %10, %10, %11\n v_add_f32 %8, %8, %12\n v_add_f32 %10, %10, %13\n", )
PROBE(k_add_f32_dpp_dep, "v_add_f32_dpp %8, %9, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, %10, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, %11, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, %12, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n", )
PROBE(k_add_f32_dpp_dep_abs, "v_add_f32_dpp %8, |%9|, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, |%10|, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, |%11|, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %8, |%12|, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n", )
PROBE(k_add_f32_dpp_ind, "v_add_f32_dpp %8, %9, %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %11, %12, %13 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %14, %15, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %10, %12, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n", )
PROBE(k_add_f32_dep_two_chains, "v_add_f32 %8, %8, %9\n v_add_f32 %10, %10, %11\n v_add_f32 %8, %8, %12\n v_add_f32 %10, %10, %13\n", )
// integer forms the directional detector and the resample are made of (%16-%19 ints)
PROBE(k_perm_b32, "v_perm_b32 %16, %17, %18, %19\n v_perm_b32 %17, %18, %19, %16\n v_perm_b32 %18, %19, %16, %17\n v_perm_b32 %19, %16, %17, %18\n", )
PROBE(k_sad_u16, "v_sad_u16 %16, %17, %18, %16\n v_sad_u16 %19, %17, %18, %19\n v_sad_u16 %16, %18, %17, %16\n v_sad_u16 %19, %18, %17, %19\n", )
PROBE(k_mad_u32_u24, "v_mad_u32_u24 %16, %17, %18, %19\n v_mad_u32_u24 %17, %18, %19, %16\n v_mad_u32_u24 %18, %19, %16, %17\n v_mad_u32_u24 %19, %16, %17, %18\n", )
PROBE(k_pk_add_u16, "v_pk_add_u16 %16, %17, %18\n v_pk_add_u16 %17, %18, %19\n v_pk_add_u16 %18, %19, %16\n v_pk_add_u16 %19, %16, %17\n", )
PROBE(k_dot2_i32_i16, "v_dot2_i32_i16 %16, %17, %18, %16\n v_dot2_i32_i16 %19, %17, %18, %19\n v_dot2_i32_i16 %16, %18, %17, %16\n v_dot2_i32_i16 %19, %18, %17, %19\n", )
PROBE(k_add_u32, "v_add_u32 %16, %17, %18\n v_add_u32 %17, %18, %19\n v_add_u32 %18, %19, %16\n v_add_u32 %19, %16, %17\n", )
PROBE(k_add3_u32, "v_add3_u32 %16, %17, %18, %19\n v_add3_u32 %17, %18, %19, %16\n v_add3_u32 %18, %19, %16, %17\n v_add3_u32 %19, %16, %17, %18\n", )
PROBE(k_lshl_add_u32, "v_lshl_add_u32 %16, %17, 2, %19\n v_lshl_add_u32 %17, %18, 2, %16\n v_lshl_add_u32 %18, %19, 2, %17\n v_lshl_add_u32 %19, %16, 2, %18\n", )
PROBE(k_bfe_u32, "v_bfe_u32 %16, %17, 8, 8\n v_bfe_u32 %17, %18, 8, 8\n v_bfe_u32 %18, %19, 8, 8\n v_bfe_u32 %19, %16, 8, 8\n", )
PROBE(k_med3_i32, "v_med3_i32 %16, %17, 0, %19\n v_med3_i32 %17, %18, 0, %16\n v_med3_i32 %18, %19, 0, %17\n v_med3_i32 %19, %16, 0, %18\n", )
PROBE(k_alignbit, "v_alignbit_b32 %16, %17, %18, 8\n v_alignbit_b32 %17, %18, %19, 8\n v_alignbit_b32 %18, %19, %16, 8\n v_alignbit_b32 %19, %16, %17, 8\n", )
PROBE(k_and_b32, "v_and_b32 %16, %17, %18\n v_and_b32 %17, %18, %19\n v_and_b32 %18, %19, %16\n v_and_b32 %19, %16, %17\n", )
PROBE(k_lshrrev_b32, "v_lshrrev_b32 %16, 8, %18\n v_lshrrev_b32 %17, 8, %19\n v_lshrrev_b32 %18, 8, %16\n v_lshrrev_b32 %19, 8, %17\n", )
PROBE(k_lshlrev_b64, "v_lshlrev_b64 %0, 8, %1\n v_lshlrev_b64 %2, 8, %3\n v_lshlrev_b64 %4, 8, %5\n v_lshlrev_b64 %6, 8, %7\n", )
PROBE(k_dot4_u32_u8, "v_dot4_u32_u8 %16, %17, %18, %16\n v_dot4_u32_u8 %19, %17, %18, %19\n v_dot4_u32_u8 %16, %18, %17, %16\n v_dot4_u32_u8 %19, %18, %17, %19\n", )
PROBE(k_cmp_cndmask_vcc, "v_cmp_lt_u32 vcc, %16, %17\n v_cndmask_b32 %18, %19, %16, vcc\n v_cmp_lt_u32 vcc, %17, %18\n v_cndmask_b32 %19, %16, %17, vcc\n", )
PROBE(k_cmp_e64_cndmask_e64, "v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cndmask_b32_e64 %18, %19, %16, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %17, %18\n v_cndmask_b32_e64 %19, %16, %17, s[22:23]\n", )
PROBE(k_fma_f32_dep, "v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %8, %8, %9, %10\n", )
PROBE(k_mul_f32_ind, "v_mul_f32 %8, %9, %10\n v_mul_f32 %11, %12, %13\n v_mul_f32 %14, %15, %9\n v_mul_f32 %10, %12, %13\n", )
PROBE(k_ldexp_f64, "v_ldexp_f64 %0, %1, %16\n v_ldexp_f64 %2, %3, %17\n v_ldexp_f64 %4, %5, %18\n v_ldexp_f64 %6, %7, %19\n", )
PROBE(k_and_or, "v_and_or_b32 %16, %17, %18, %19\n v_and_or_b32 %17, %18, %19, %16\n v_and_or_b32 %18, %19, %16, %17\n v_and_or_b32 %19, %16, %17, %18\n", )
PROBE(k_cndmask, "v_cndmask_b32 %8, %9, %10, vcc\n v_cndmask_b32 %11, %12, %13, vcc\n v_cndmask_b32 %14, %15, %9, vcc\n v_cndmask_b32 %10, %12, %13, vcc\n", )
PROBE(k_sdwa_shift, "v_lshlrev_b32_sdwa %16, %17, %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %17, %18, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %18, %19, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %19, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n", )
PROBE(k_mixed_f64_f32, "v_fma_f64 %0, %1, %2, %3\n v_add_f32 %8, %9, %10\n v_fma_f64 %4, %5, %6, %7\n v_add_f32 %11, %12, %13\n", )
PROBE(k_cvt_dep_chain, "v_cvt_f64_f32 %0, %8\n v_cvt_f32_f64 %8, %0\n v_cvt_f64_f32 %0, %8\n v_cvt_f32_f64 %8, %0\n", )

// LDS: dependent-free reads
__global__ void k_ds_read_b128(uint64_t *out, float seed)
{
	__shared__ float4 buf[1024];
	for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = make_float4(seed, seed, seed, seed);
	__syncthreads();
	float acc = 0;
	uint64_t t0 = clock64();
	for (int it = 0; it < 1024; ++it) {
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			float4 v = buf[(threadIdx.x + j * 64) & 1023];
			acc += v.x;
		}
	}
	uint64_t t1 = clock64();
	if (threadIdx.x == 0) out[0] = (t1 - t0);
	if (acc == 12345.f) out[1] = 1;
}

// one chain lane pattern: ds_read_b128 of a lane-private sequence + 4 dependent adds
__global__ void k_chain_walk(uint64_t *out, float seed)
{
	__shared__ float buf[64 * 260];
	for (int i = threadIdx.x; i < 64 * 260; i += 64) buf[i] = seed;
	__syncthreads();
	const float4 *x = reinterpret_cast<const float4 *>(buf + (threadIdx.x & 63) * 260);
	float acc = 0;
	uint64_t t0 = clock64();
	for (int it = 0; it < 256; ++it) {
#pragma unroll
		for (int j = 0; j < 64; ++j) {
			float4 v = x[j];
			acc += v.x;
			acc += v.y;
			acc += v.z;
			acc += v.w;
		}
	}
	uint64_t t1 = clock64();
	if (threadIdx.x == 0) out[0] = (t1 - t0);
	if (acc == 12345.f) out[1] = 1;
}

template <class K>
static void run(const char *name, K k, int n_instr, int waves = 1)
{
	uint64_t *d;
	hipMalloc(&d, 16);
	hipMemset(d, 0, 16);
	for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.5f);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.5f);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	uint64_t h[2];
	hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	printf("%-22s waves/block %d: %8.2f clock64-ticks per instruction %7.2f ns per instruction by events (launch %.3f ms)\n", name, waves,
	       (double)h[0] / n_instr, ms * 1e6 / n_instr, ms);
	hipFree(d);
}

int main()
{
	const int N = 256 * 64 * 4;
#define R(k) run(#k, k, N); run(#k, k, N, 8);
	R(k_perm_b32) R(k_sad_u16) R(k_mad_u32_u24) R(k_pk_add_u16) R(k_dot2_i32_i16) R(k_add_u32) R(k_add3_u32) R(k_lshl_add_u32) R(k_bfe_u32) R(k_med3_i32) R(k_alignbit) R(k_and_b32) R(k_lshrrev_b32) R(k_lshlrev_b64) R(k_dot4_u32_u8) R(k_cmp_cndmask_vcc) R(k_cmp_e64_cndmask_e64) R(k_add_f32_dpp_dep) R(k_add_f32_dpp_dep_abs) R(k_add_f32_dpp_ind) R(k_add_f32_dep_two_chains) R(k_add_f32_ind) R(k_add_f32_dep) R(k_add_f32_dep_abs) R(k_add_f32_dep2) R(k_fma_f32_dep) R(k_mul_f32_ind)
	R(k_fma_f64_ind) R(k_fma_f64_dep) R(k_mul_f64_ind) R(k_add_f64_ind) R(k_rcp_f64_ind) R(k_rcp_f64_dep) R(k_rcp_f32_ind)
	R(k_cvt_f64_f32) R(k_cvt_f32_f64) R(k_cvt_dep_chain) R(k_frexp_mant) R(k_frexp_exp) R(k_pk_mul_f32) R(k_pk_fma_f32) R(k_ldexp_f64)
	R(k_and_or) R(k_cndmask) R(k_sdwa_shift) R(k_mixed_f64_f32)
	run("k_ds_read_b128", k_ds_read_b128, 1024 * 16);
	run("k_ds_read_b128", k_ds_read_b128, 1024 * 16, 4);
	run("k_chain_walk(per add)", k_chain_walk, 256 * 64 * 4);
	run("k_chain_walk(per add)", k_chain_walk, 256 * 64 * 4, 4);
	return 0;
}
