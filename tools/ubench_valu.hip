// tools/ubench_valu.hip -- throughput of the integer VALU forms the filter kernel could use,
// measured on gfx950: cycles per wave64 instruction with 4 or 8 waves per SIMD resident.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu && ./tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef short short2v __attribute__((ext_vector_type(2)));

#define ITER 4096
#define UNROLL 16

template<int OP>
__global__ void k(const int *in, int *out, int s0, int s1)
{
	int a[UNROLL];
	int x = in[threadIdx.x], y = in[threadIdx.x + 64];
#pragma unroll
	for(int i = 0; i < UNROLL; i++) a[i] = x + i;
	for(int it = 0; it < ITER; it++)
	{
#pragma unroll
		for(int i = 0; i < UNROLL; i++)
		{
			if(OP == 0) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, y), __builtin_bit_cast(short2v, s0), a[i], false);       // v_dot2c_i32_i16 sgpr
			if(OP == 1) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, y), __builtin_bit_cast(short2v, x), a[i], false);        // vgpr x vgpr
			if(OP == 2) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "s"(s0));
			if(OP == 3) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "s"(s0));
			if(OP == 4) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "s"(s0));
			if(OP == 5) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a[i]) : "v"(y));
			if(OP == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 7) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 8) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 9) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 11) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 12) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(*(long long *) &a[i & ~1]) : "v"(y), "v"(x) : "vcc");
			if(OP == 13) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 14) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(y));
			if(OP == 15) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 16) asm volatile("v_ashrrev_i32 %0, 15, %0" : "+v"(a[i]));
			if(OP == 17) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 18) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(y));
			if(OP == 19) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 20) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
			if(OP == 21) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[i]));
			if(OP == 22) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
			if(OP == 23) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(y));
			if(OP == 24) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 25) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 26) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 27) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(y), "v"(x));
			if(OP == 28) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % UNROLL]));
			if(OP == 29) asm volatile("v_pk_add_i16 %0, %1, %0" : "+v"(a[i]) : "v"(y));
			if(OP == 30) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(a[i]));
			if(OP == 31) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
		}
	}
	int r = 0;
#pragma unroll
	for(int i = 0; i < UNROLL; i++) r += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template<int OP>
static void run(const char *name, int *din, int *dout)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for(int wps = 4; wps <= 8; wps *= 2)
	{
		// 256 CUs, 4 SIMDs each: blocks of 256 threads = 1 wave per SIMD; wps blocks per CU
		int blocks = 256 * wps;
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, din, dout, 0x00030002, 5);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, din, dout, 0x00030002, 5);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		double inst_per_simd = (double) ITER * UNROLL * wps;      // wave-instructions issued on each SIMD
		double cyc = ms * 1e-3 * 2.4e9 / inst_per_simd;
		printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, wps, ms, cyc);
	}
}

int main()
{
	int *din, *dout;
	hipMalloc(&din, 4096); hipMemset(din, 1, 4096);
	hipMalloc(&dout, 256 * 8 * 256 * 4);
	run<0>("v_dot2c_i32_i16 (sgpr tap)", din, dout);
	run<1>("v_dot2c_i32_i16 (vgpr)", din, dout);
	run<11>("v_dot2_i32_i16 (vop3p)", din, dout);
	run<2>("v_mad_i32_i16", din, dout);
	run<3>("v_mad_i32_i24", din, dout);
	run<4>("v_mad_u32_u24", din, dout);
	run<5>("v_alignbit_b32", din, dout);
	run<6>("v_add_u32", din, dout);
	run<7>("v_pk_mad_i16", din, dout);
	run<8>("v_mul_lo_u32", din, dout);
	run<9>("v_dot4_i32_i8", din, dout);
	run<10>("v_fma_f32", din, dout);
	run<12>("v_mad_i64_i32", din, dout);
	run<13>("v_pk_add_u16", din, dout);
	run<14>("v_pk_sub_u16", din, dout);
	run<29>("v_pk_add_i16", din, dout);
	run<25>("v_pk_mad_u16", din, dout);
	run<26>("v_pk_mul_lo_u16", din, dout);
	run<15>("v_perm_b32", din, dout);
	run<16>("v_ashrrev_i32", din, dout);
	run<30>("v_lshlrev_b32", din, dout);
	run<17>("v_xor_b32", din, dout);
	run<18>("v_lshl_or_b32", din, dout);
	run<19>("v_and_or_b32", din, dout);
	run<20>("v_cvt_pk_i16_i32", din, dout);
	run<21>("v_bfe_i32", din, dout);
	run<22>("v_cndmask_b32", din, dout);
	run<23>("v_lshl_add_u32", din, dout);
	run<24>("v_add3_u32", din, dout);
	run<27>("v_bfi_b32", din, dout);
	run<28>("v_mov_b32", din, dout);
	run<31>("v_sub_u32", din, dout);
	return 0;
}
