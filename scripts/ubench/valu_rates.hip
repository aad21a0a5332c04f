// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// path tracer leans on, measured with all 4 SIMDs of every CU busy (4 waves/SIMD).  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int WHICH>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed, int lanes = 64)
{
	float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
	float b = seed * 0.999f, c = seed * 0.001f;
	double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = b, dc = c;
	unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x1234567, u2 = u0 + 77, u3 = u1 + 99;
	unsigned long long l0 = u0, l1 = u1, l2 = u2, l3 = u3;
	typedef float float2v __attribute__((ext_vector_type(2)));
	float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
	if ((int) (threadIdx.x & 63) < lanes)      /* EXEC mask: does a half-empty wave issue faster? */
	for (int i = 0; i < iters; i++) {
		if (WHICH == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
		if (WHICH == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
		if (WHICH == 2) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
		if (WHICH == 3) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
		if (WHICH == 4) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
		if (WHICH == 5) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %2, %5\n v_add_f64 %3, %3, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));) }
		if (WHICH == 6) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) : "v"(u0), "v"(u1) : "vcc");) }
		if (WHICH == 7) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779B9u));) }
		if (WHICH == 8) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
		if (WHICH == 9) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
		if (WHICH == 10) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_lshlrev_b32 %2, 3, %2\n v_and_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779B9u));) }
		if (WHICH == 11) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
		if (WHICH == 12) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
		if (WHICH == 13) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f32_f64 %2, %1\n v_cvt_f64_f32 %1, %5\n v_cvt_f32_f64 %3, %0" : "+v"(d0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
		if (WHICH == 14) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mad_u32_u24 %2, %2, %4, %3\n v_mul_u32_u24 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779u));) }
		if (WHICH == 15) { REP8(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) : "v"(u0));) }
		if (WHICH == 16) { REP8(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
		if (WHICH == 17) { REP8(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0));) }
		if (WHICH == 18) { REP8(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_u32 %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
		if (WHICH == 19) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));) }
		if (WHICH == 20) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, %4\n v_alignbit_b32 %1, %1, %2, %4\n v_min_u32 %2, %2, %4\n v_min3_f32 %3, %3, %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(a3) : "v"(u3));) }
		if (WHICH == 21) { REP8(asm volatile("v_lshrrev_b64 %0, 30, %0\n v_lshrrev_b64 %1, 27, %1\n v_lshrrev_b64 %2, 31, %2\n v_lshrrev_b64 %3, 5, %3" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));) }
		if (WHICH == 22) { REP8(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float) (d0 + d1 + d2 + d3) + (float) (u0 ^ u1 ^ u2 ^ u3) + (float) (l0 ^ l1 ^ l2 ^ l3) + p0.x + p1.y + p2.x + p3.y;
}

template <int W> double run(const char *name, float *d_out, int cus, int lanes = 64)
{
	const int iters = 20000, blocks = cus * 4;    // 4 blocks of 256 = 16 waves/CU = 4 waves/SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f, lanes);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f, lanes);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// instructions per SIMD: 4 waves * iters * 32 instrs
	double instr_per_simd = 4.0 * iters * 32.0;
	double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
	printf("%-40s %8.3f ms  -> %.2f cycles/instr/SIMD (at 2.4 GHz)\n", name, ms, cyc);
	return cyc;
}

int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	int cus = p.multiProcessorCount;
	float *d; hipMalloc(&d, sizeof(float) * cus * 4 * 256);
	printf("%s, %d CUs\n", p.name, cus);
	run<0>("v_fma_f32", d, cus);
	run<1>("v_mul_f32 / v_add_f32", d, cus);
	run<2>("v_pk_fma_f32", d, cus);
	run<3>("v_pk_mul_f32 / v_pk_add_f32", d, cus);
	run<4>("v_fma_f64", d, cus);
	run<5>("v_mul_f64 / v_add_f64", d, cus);
	run<6>("v_mad_u64_u32", d, cus);
	run<7>("v_mul_lo_u32 / v_mul_hi_u32", d, cus);
	run<8>("v_rcp_f32 / v_sqrt_f32", d, cus);
	run<9>("v_cndmask / v_cmp_f32", d, cus);
	run<10>("v_add_u32/xor/lshl/and", d, cus);
	run<11>("v_rcp_f64 / v_rsq_f64", d, cus);
	run<12>("div_scale/fixup/fmas/max3", d, cus);
	run<13>("cvt f64<->f32", d, cus);
	run<14>("u24 mul family", d, cus);
	run<15>("v_lshlrev_b64 (variable)", d, cus);
	run<16>("v_ffbh_u32", d, cus);
	run<17>("v_ldexp_f32", d, cus);
	run<18>("v_cvt_f32_u32", d, cus);
	run<19>("v_lshl_add_u64", d, cus);
	run<20>("alignbit/alignbit/min_u32/min3_f32", d, cus);
	run<21>("v_lshrrev_b64 (constant)", d, cus);
	run<22>("v_cvt_f64_u32", d, cus);
	run<0>("v_fma_f32, lanes 0-31 only", d, cus, 32);
	run<0>("v_fma_f32, lanes 0-15 only", d, cus, 16);
	run<4>("v_fma_f64, lanes 0-31 only", d, cus, 32);
	run<6>("v_mad_u64_u32, lanes 0-31 only", d, cus, 32);
	run<8>("v_rcp/sqrt_f32, lanes 0-31 only", d, cus, 32);
	return 0;
}
