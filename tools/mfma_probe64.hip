// Layout probes for the 64x64 fast path: v_mfma_i32_16x16x64_i8 operand lanes and the wave-wide DPP shift.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int8_t *A, const int8_t *B, int *D)
{
	// A: 16x64 row-major, B: 64x16 row-major (k rows), D: 16x16.  Guess: lane (o = l & 15, g = l >> 4) holds k = 16g .. 16g+15
	const uint32_t l = threadIdx.x, o = l & 15, g = l >> 4;
	v4i a, b;
	for (int w = 0; w < 4; ++w) {
		uint32_t av = 0, bv = 0;
		for (int j = 0; j < 4; ++j) {
			av |= (uint32_t)(uint8_t)A[o * 64 + 16 * g + 4 * w + j] << (8 * j);
			bv |= (uint32_t)(uint8_t)B[(16 * g + 4 * w + j) * 16 + o] << (8 * j);
		}
		a[w] = (int)av;
		b[w] = (int)bv;
	}
	v4i c = {0, 0, 0, 0};
	v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + o] = d[r];
	// wave shift left by one lane (DPP wave_shl:1 = 0x130): lane i reads lane i+1, lane 63 keeps bound value 0
	D[256 + l] = __builtin_amdgcn_update_dpp(0, (int)(1000 + l), 0x130, 0xf, 0xf, true);
}
int main()
{
	int8_t hA[1024], hB[1024];
	for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)((i * 37 + 11) % 251 - 125); hB[i] = (int8_t)((i * 53 + 7) % 241 - 120); }
	int8_t *dA, *dB; int *dD; int hD[320];
	hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, sizeof hD);
	hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
		int s = 0; for (int k = 0; k < 64; ++k) s += (int)hA[i * 64 + k] * (int)hB[k * 16 + j];
		if (s != hD[i * 16 + j]) ++bad;
	}
	printf("mfma_i32_16x16x64_i8 layout (k = 16g + j) mismatches: %d\n", bad);
	int sbad = 0;
	for (int l = 0; l < 63; ++l) if (hD[256 + l] != 1000 + l + 1) ++sbad;
	printf("wave_shl:1 mismatches: %d (lane 15 -> %d, lane 63 -> %d)\n", sbad, hD[256 + 15], hD[256 + 63]);
	return 0;
}
