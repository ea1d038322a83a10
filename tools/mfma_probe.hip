#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int8_t *A, const int8_t *B, int *D)
{
	// A: 16x32 row-major, B: 32x16 row-major (k rows), D: 16x16
	const uint32_t l = threadIdx.x, o = l & 15, g = l >> 4;
	long a = 0, b = 0;
	for (int j = 0; j < 8; ++j) {
		a |= (long)(uint8_t)A[o * 32 + 8 * g + j] << (8 * j);
		b |= (long)(uint8_t)B[(8 * g + j) * 16 + o] << (8 * j);
	}
	v4i c = {0, 0, 0, 0};
	v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + o] = d[r];
	// SDWA check: place (x >> s) low byte into byte 2
	uint32_t dst = 0x11223344u, val = 0x0001f300u + l, sh = 8;
	asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(sh), "v"(val));
	D[256 + l] = (int)dst;
}
int main()
{
	int8_t hA[512], hB[512];
	for (int i = 0; i < 512; ++i) { hA[i] = (int8_t)((i * 37 + 11) % 251 - 125); hB[i] = (int8_t)((i * 53 + 7) % 241 - 120); }
	int8_t *dA, *dB; int *dD; int hD[320];
	hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, sizeof hD);
	hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
		int s = 0; for (int k = 0; k < 32; ++k) s += (int)hA[i * 32 + k] * (int)hB[k * 16 + j];
		if (s != hD[i * 16 + j]) ++bad;
	}
	printf("mfma_i32_16x16x32_i8 layout mismatches: %d\n", bad);
	int sbad = 0;
	for (int l = 0; l < 64; ++l) { uint32_t e = (0x11223344u & 0xff00ffffu) | ((((0x0001f300u + l) >> 8) & 0xffu) << 16); if ((uint32_t)hD[256 + l] != e) ++sbad; }
	printf("sdwa mismatches: %d (lane0 %08x)\n", sbad, hD[256]);
	return bad || sbad;
}
