/* tools/ubench_mfma_fir.hip -- stand-alone check of the int8-MFMA form of the 51-tap int16 FIR
 * (the layout assumptions of hvk_k_filter's MFMA path), against a scalar loop on the host.
 *   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_fir.hip -o /tmp/ubench_mfma_fir && /tmp/ubench_mfma_fir
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));

#define NT 51
#define H 25
#define LEAD 26
#define TILE 1024

/* y[n] = sum_k h[k] x[n - H + k]; x given from sample -LEAD */
__global__ __launch_bounds__(128) void k_fir_packed(const int16_t *x, const int4v *atab, int cI, int cQ, int *iq)
{
	__shared__ __attribute__((aligned(16))) unsigned char xh[TILE + 64], xl[TILE + 64];
	__shared__ __attribute__((aligned(16))) int outl[TILE];
	const int t = threadIdx.x;
	const int16_t *src = x + (size_t) blockIdx.x * TILE;

	{
		const int4v *s4 = (const int4v *) src;          /* 8 samples per lane */
		for(int q = t; q < (TILE + 64) / 8; q += 128)
		{
			const int4v d = s4[q];
			int2v hh2, ll2;
			hh2.x = __builtin_amdgcn_perm(d.y, d.x, 0x07050301); hh2.y = __builtin_amdgcn_perm(d.w, d.z, 0x07050301);
			ll2.x = __builtin_amdgcn_perm(d.y, d.x, 0x06040200) ^ 0x80808080; ll2.y = __builtin_amdgcn_perm(d.w, d.z, 0x06040200) ^ 0x80808080;
			((int2v *) xh)[q] = hh2;
			((int2v *) xl)[q] = ll2;
		}
	}
	__syncthreads();

	const int lane = t & 63, wave = t >> 6, g = lane >> 4, c = lane & 15;
	const int4v a_hh = atab[lane], a_hl = atab[64 + lane];

#pragma unroll
	for(int j = 0; j < 4; j++)
	{
		const int seg = wave * 64 + j * 16 + c;
		const int off = seg * 8 + g * 16;
		int4v bh, bl;
		bh.xy = *(const int2v *) (xh + off); bh.zw = *(const int2v *) (xh + off + 8);
		bl.xy = *(const int2v *) (xl + off); bl.zw = *(const int2v *) (xl + off + 8);
		int4v p_hh = { 0, 0, 0, 0 }, p_m = { 0, 0, 0, 0 }, p_ll = { cI, cQ, cI, cQ };
		p_hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bh, p_hh, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bh, p_m, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bl, p_m, 0, 0, 0);
		p_ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bl, p_ll, 0, 0, 0);
		int y[4];
#pragma unroll
		for(int i = 0; i < 4; i++) y[i] = (int) ((((unsigned) p_hh[i] << 8) + (unsigned) p_m[i]) << 8) + p_ll[i];
		int2v o;
		o.x = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(y[0] >> 15, y[1] >> 15));
		o.y = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(y[2] >> 15, y[3] >> 15));
		*(int2v *) (outl + seg * 8 + 2 * g) = o;
	}
	__syncthreads();
	int4v *dst = (int4v *) (iq + (size_t) blockIdx.x * TILE + t * 8);
	dst[0] = ((const int4v *) (outl + t * 8))[0];
	dst[1] = ((const int4v *) (outl + t * 8))[1];
}

__global__ __launch_bounds__(128) void k_fir(const int16_t *x, const int4v *atab, int cI, int cQ, int *yi, int *yq)
{
	__shared__ __attribute__((aligned(16))) unsigned char xh[TILE + 64], xl[TILE + 64];
	const int t = threadIdx.x;
	const int16_t *src = x + (size_t) blockIdx.x * TILE;      /* element 0 = sample n0 - LEAD */

	for(int q = t; q < (TILE + 64) / 4; q += 128)
	{
		const int d0 = ((const int *) src)[q * 2], d1 = ((const int *) src)[q * 2 + 1];
		((int *) xh)[q] = __builtin_amdgcn_perm(d1, d0, 0x07050301);
		((int *) xl)[q] = __builtin_amdgcn_perm(d1, d0, 0x06040200) ^ 0x80808080;
	}
	__syncthreads();

	const int lane = t & 63, wave = t >> 6, g = lane >> 4, c = lane & 15;
	const int4v a_hh = atab[lane], a_hl = atab[64 + lane];

	for(int j = 0; j < 4; j++)
	{
		const int seg = wave * 64 + j * 16 + c;
		const int off = seg * 8 + g * 16;
		int4v bh, bl;
		bh.xy = *(const int2v *) (xh + off); bh.zw = *(const int2v *) (xh + off + 8);
		bl.xy = *(const int2v *) (xl + off); bl.zw = *(const int2v *) (xl + off + 8);
		int4v p_hh = { 0, 0, 0, 0 }, p_m = { 0, 0, 0, 0 }, p_ll = { cI, cQ, cI, cQ };
		p_hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bh, p_hh, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bh, p_m, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bl, p_m, 0, 0, 0);
		p_ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bl, p_ll, 0, 0, 0);
		int y[4];
		for(int i = 0; i < 4; i++) y[i] = (int) ((((unsigned) p_hh[i] << 8) + (unsigned) p_m[i]) << 8) + p_ll[i];
		/* rows 4g + i: b = 2g + (i >> 1), iq = i & 1 */
		const size_t n = (size_t) blockIdx.x * TILE + seg * 8 + 2 * g;
		yi[n] = y[0]; yq[n] = y[1]; yi[n + 1] = y[2]; yq[n + 1] = y[3];
	}
}

int main(void)
{
	const int tiles = 8192, N = tiles * TILE;
	std::vector<int16_t> x(N + 128), hi(NT), hq(NT);
	srand(1);
	for(auto &v : x) v = (int16_t) (rand() & 0xFFFF);
	for(int k = 0; k < NT; k++) { hi[k] = (int16_t) ((rand() % 65000) - 32500); hq[k] = (int16_t) ((rand() % 65000) - 32500); }
	hi[3] = 32639; hq[7] = -32768; x[100] = 32767; x[101] = -32768;

	/* A table: [part hh/hl][lane][16 bytes]; lane (g, m): t' = 16 g + j, row m: b = 2 (m >> 2) + ((m & 3) >> 1), iq = m & 1;
	 * A[m][t'] = h_iq[t' - 1 - b] */
	std::vector<signed char> A(2 * 64 * 16);
	long sumI = 0, sumQ = 0;
	for(int k = 0; k < NT; k++) { sumI += hi[k]; sumQ += hq[k]; }
	for(int lane = 0; lane < 64; lane++)
	{
		const int g = lane >> 4, m = lane & 15, b = 2 * (m >> 2) + ((m & 3) >> 1), iq = m & 1;
		for(int j = 0; j < 16; j++)
		{
			const int k = 16 * g + j - 1 - b;
			const int h = (k >= 0 && k < NT) ? (iq ? hq[k] : hi[k]) : 0;
			const int lo = (int) (signed char) (h & 0xFF), hh = (h - lo) >> 8;
			if(hh < -128 || hh > 127) { printf("tap out of range\n"); return(1); }
			A[(0 * 64 + lane) * 16 + j] = (signed char) hh;
			A[(1 * 64 + lane) * 16 + j] = (signed char) lo;
		}
	}

	int16_t *dx; int4v *da; int *dyi, *dyq;
	hipMalloc(&dx, x.size() * 2); hipMalloc(&da, A.size()); hipMalloc(&dyi, N * 4); hipMalloc(&dyq, N * 4);
	hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
	hipMemcpy(da, A.data(), A.size(), hipMemcpyHostToDevice);
	const int cI = (int) (128 * sumI), cQ = (int) (128 * sumQ);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for(int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fir, dim3(tiles), dim3(128), 0, 0, dx, da, cI, cQ, dyi, dyq);
	hipEventRecord(e0);
	for(int r = 0; r < 10; r++) hipLaunchKernelGGL(k_fir, dim3(tiles), dim3(128), 0, 0, dx, da, cI, cQ, dyi, dyq);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<int> yi(N), yq(N);
	hipMemcpy(yi.data(), dyi, N * 4, hipMemcpyDeviceToHost); hipMemcpy(yq.data(), dyq, N * 4, hipMemcpyDeviceToHost);

	long bad = 0;
	for(int n = 0; n < N; n += (n < 4096 ? 1 : 97))
	{
		unsigned ai = 0, aq = 0;
		for(int k = 0; k < NT; k++)
		{
			const int xv = x[n + LEAD - H + k];     /* x[] element 0 = sample -LEAD */
			ai += (unsigned) (xv * hi[k]); aq += (unsigned) (xv * hq[k]);
		}
		if((int) ai != yi[n] || (int) aq != yq[n]) { if(bad < 5) printf("n=%d want %d %d got %d %d\n", n, (int) ai, (int) aq, yi[n], yq[n]); bad++; }
	}
	{
		int *diq; hipMalloc(&diq, (size_t) N * 4);
		for(int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fir_packed, dim3(tiles), dim3(128), 0, 0, dx, da, cI, cQ, diq);
		hipEventRecord(e0);
		for(int r = 0; r < 10; r++) hipLaunchKernelGGL(k_fir_packed, dim3(tiles), dim3(128), 0, 0, dx, da, cI, cQ, diq);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms2; hipEventElapsedTime(&ms2, e0, e1);
		std::vector<int> o(N);
		hipMemcpy(o.data(), diq, (size_t) N * 4, hipMemcpyDeviceToHost);
		long bad2 = 0;
		for(int n = 0; n < N; n++)
		{
			int a = yi[n] >> 15, b = yq[n] >> 15;
			a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a); b = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
			if(o[n] != (int) ((a & 0xFFFF) | ((unsigned) b << 16))) bad2++;
		}
		printf("packed kernel: %ld mismatches against the unpacked one; %.4f ms = %.1f Gsamples/s, %.2f TB/s of 6 B/sample\n", bad2, ms2 / 10, N / (ms2 / 10) / 1e6, N * 6.0 / (ms2 / 10) / 1e9);
	}
	printf("%s: %ld mismatches; %.3f ms per %d samples = %.1f Gsamples/s (with uncoalesced int32 stores)\n", bad ? "DIFFERENT" : "EQUAL", bad, ms / 10, N, N / (ms / 10) / 1e6);
	return(bad ? 1 : 0);
}
