// tools/ubench_valu2.hip -- issue rates of the forms the SECAM walk and the select-heavy parts of hvk_k_direct use
// (FP64, conversions, v_cndmask with either mask register, population count, v_med3), gfx950, cycles per wave64
// instruction with 4 or 8 waves per SIMD. hipcc --offload-arch=gfx950 -O3 tools/ubench_valu2.hip -o tools/_ubench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2048
#define UNROLL 16

template<int OP>
__global__ void k(const int *in, int *out, int s0, unsigned long long smask)
{
	int a[UNROLL];
	double d[UNROLL / 2];
	int x = in[threadIdx.x], y = in[threadIdx.x + 64];
	const double dy = (double) y * 1e-9 + 1.0, dx = (double) x * 1e-9;
#pragma unroll
	for(int i = 0; i < UNROLL; i++) a[i] = x + i;
#pragma unroll
	for(int i = 0; i < UNROLL / 2; i++) d[i] = (double) (x + i);
	for(int it = 0; it < ITER; it++)
	{
#pragma unroll
		for(int i = 0; i < UNROLL; i++)
		{
			if(OP == 0) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "s"(smask));
			if(OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y));
			if(OP == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 4) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 5) asm volatile("v_bfm_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 6) asm volatile("v_readlane_b32 %0, %1, 3\n\tv_add_u32 %2, %0, %2" : "=s"(s0), "+v"(y), "+v"(a[i]));
			if(OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i / 2]) : "v"(dy));
			if(OP == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i / 2]) : "v"(dx));
			if(OP == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i / 2]) : "v"(dy), "v"(dx));
			if(OP == 10) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i / 2]) : "v"(a[i]));
			if(OP == 11) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i / 2]));
			if(OP == 12) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i / 2]));
			if(OP == 13) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
			if(OP == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
			if(OP == 15) asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 16) asm volatile("v_ashrrev_i64 %0, 31, %0" : "+v"(*(long long *) &a[i & ~1]));
			if(OP == 17) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(*(long long *) &a[i & ~1]) : "v"(*(long long *) &d[0]));
		}
	}
	int r = 0;
#pragma unroll
	for(int i = 0; i < UNROLL; i++) r += a[i];
#pragma unroll
	for(int i = 0; i < UNROLL / 2; i++) r += (int) d[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r + s0;
}

template<int OP>
static void run(const char *name, int *din, int *dout, int per)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for(int wps = 4; wps <= 8; wps *= 2)
	{
		int blocks = 256 * wps;
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, din, dout, 5, 0x5555AAAA5555AAAAull);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, din, dout, 5, 0x5555AAAA5555AAAAull);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		double inst_per_simd = (double) ITER * UNROLL * wps * per;
		printf("%-44s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
	}
}

int main()
{
	int *din, *dout;
	hipMalloc(&din, 4096); hipMemset(din, 1, 4096);
	hipMalloc(&dout, 256 * 8 * 256 * 4);
	run<0>("v_cndmask_b32 (mask in an SGPR pair)", din, dout, 1);
	run<1>("v_cndmask_b32 (mask in vcc)", din, dout, 1);
	run<13>("v_cmp_lt_i32 + v_cndmask_b32 (per pair)", din, dout, 2);
	run<2>("v_bcnt_u32_b32", din, dout, 1);
	run<3>("v_med3_i32", din, dout, 1);
	run<14>("v_min_i32", din, dout, 1);
	run<4>("v_mul_i32_i24", din, dout, 1);
	run<15>("v_mul_hi_i32", din, dout, 1);
	run<5>("v_bfm_b32", din, dout, 1);
	run<6>("v_readlane_b32 + v_add_u32 (per pair)", din, dout, 2);
	run<16>("v_ashrrev_i64", din, dout, 1);
	run<17>("v_lshl_add_u64", din, dout, 1);
	run<7>("v_mul_f64", din, dout, 1);
	run<8>("v_add_f64", din, dout, 1);
	run<9>("v_fma_f64", din, dout, 1);
	run<10>("v_cvt_f64_i32", din, dout, 1);
	run<11>("v_cvt_i32_f64", din, dout, 1);
	run<12>("v_rndne_f64", din, dout, 1);
	return 0;
}
