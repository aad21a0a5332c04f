// Micro-benchmark: issue cost of the VALU instructions the path tracer leans on, in REAL shader cycles.
//
// Every wave stamps s_memtime (shader-clock ticks) and s_memrealtime (100 MHz) around its loop, so the
// result does not depend on an assumed clock: cycles per wave64 instruction per SIMD =
//   median over waves of  delta(s_memtime) / (waves per SIMD x instructions per wave),
// and the clock the chip actually held = delta(s_memtime) / delta(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  Each kind runs at 1, 2, 4 and 8 waves per SIMD with
// eight independent accumulators per wave (no dependent-issue stalls), all CUs busy, after a warm-up of
// back-to-back launches.  The wall-clock figure (HIP events, nominal 2.4 GHz) is printed beside it to show
// what an un-clocked measurement would have claimed.  Development aid; output kept under profiles/.
//
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

struct Stamp { unsigned long long ticks, real; };

template <int WHICH>
__global__ void __launch_bounds__(256) k(float *out, Stamp *stamps, int iters, float seed, int lanes)
{
	float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
	float b = seed * 0.999f, c = seed * 0.001f;
	double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = b, dc = c;
	unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x1234567, u2 = u0 + 77, u3 = u1 + 99, u4 = u0 * 3, u5 = u1 * 5, u6 = u2 * 7, u7 = u3 * 9;
	unsigned ub = u0 ^ 0x04050607u, uc = 16u & u1;
	unsigned long long l0 = u0, l1 = u1, l2 = u2, l3 = u3, l4 = u4, l5 = u5, l6 = u6, l7 = u7;
	typedef float float2v __attribute__((ext_vector_type(2)));
	float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, pb = {b, b}, pc = {c, c};
	unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
	__syncthreads();
	t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
	if ((int) (threadIdx.x & 63) < lanes)      /* EXEC mask: does a half-empty wave issue faster? */
	for (int i = 0; i < iters; i++) {
#define F8(op, A, B) REP8(asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" \
		: "+v"(A##0), "+v"(A##1), "+v"(A##2), "+v"(A##3), "+v"(A##4), "+v"(A##5), "+v"(A##6), "+v"(A##7) : "v"(B##b), "v"(B##c));)
#define G8(op, A, B) REP8(asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
		: "+v"(A##0), "+v"(A##1), "+v"(A##2), "+v"(A##3), "+v"(A##4), "+v"(A##5), "+v"(A##6), "+v"(A##7) : "v"(B));)
#define H8(op, A) REP8(asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
		: "+v"(A##0), "+v"(A##1), "+v"(A##2), "+v"(A##3), "+v"(A##4), "+v"(A##5), "+v"(A##6), "+v"(A##7));)
		if (WHICH == 0) { F8("v_fma_f32", a, ) }
		if (WHICH == 1) { G8("v_mul_f32", a, b) }
		if (WHICH == 2) { G8("v_add_f32", a, c) }
		if (WHICH == 3) { F8("v_pk_fma_f32", p, p) }
		if (WHICH == 4) { G8("v_pk_mul_f32", p, pb) }
		if (WHICH == 5) { F8("v_fma_f64", d, d) }
		if (WHICH == 6) { G8("v_mul_f64", d, db) }
		if (WHICH == 7) { G8("v_add_f64", d, dc) }
		if (WHICH == 8) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
		                                    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
		                                    : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4), "+v"(l5), "+v"(l6), "+v"(l7) : "v"(u0), "v"(u1) : "vcc");) }
		if (WHICH == 9) { H8("v_rcp_f32", a) }
		if (WHICH == 10) { H8("v_sqrt_f32", a) }
		if (WHICH == 11) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
		                                     "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
		if (WHICH == 12) { G8("v_add_u32", u, u0) }
		if (WHICH == 13) { G8("v_max_f32", a, b) }
		if (WHICH == 14) { G8("v_mul_lo_u32", u, u1) }
		if (WHICH == 15) { REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f32_f64 %4, %1\n v_cvt_f64_f32 %1, %9\n v_cvt_f32_f64 %5, %0\n v_cvt_f64_f32 %2, %8\n v_cvt_f32_f64 %6, %3\n v_cvt_f64_f32 %3, %9\n v_cvt_f32_f64 %7, %2"
		                                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
		if (WHICH == 16) { H8("v_rcp_f64", d) }
		if (WHICH == 17) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
		                                     "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
		if (WHICH == 19) { F8("v_max3_f32", a, ) }
		if (WHICH == 20) { F8("v_med3_f32", a, ) }
		if (WHICH == 21) { G8("v_min_f32", a, b) }
		if (WHICH == 22) { REP8(asm volatile("v_cmp_gt_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32 %3, %3, %8, s[22:23]\n"
		                                     "v_cmp_gt_f32 s[24:25], %4, %8\n v_cndmask_b32 %5, %5, %8, s[24:25]\n v_cmp_lt_f32 s[26:27], %6, %8\n v_cndmask_b32 %7, %7, %8, s[26:27]"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
		if (WHICH == 23) { G8("v_xor_b32", u, u0) }
		if (WHICH == 24) { G8("v_lshlrev_b32", u, u1) }
		if (WHICH == 25) { G8("v_sub_f32", a, c) }
		if (WHICH == 26) { G8("v_mul_hi_u32", u, u1) }
		if (WHICH == 27) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 32) { REP8(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[1,0,0]\n"
		                                     "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(c));) }
		if (WHICH == 33) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %9\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %9\n v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %9"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1));) }
		if (WHICH == 34) { F8("v_perm_b32", u, u) }
		if (WHICH == 35) { F8("v_alignbit_b32", u, u) }
		if (WHICH == 36) { F8("v_min3_f32", a, ) }
		if (WHICH == 28) { REP8(asm volatile("v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %9\n v_cvt_f32_u32 %2, %8\n v_cvt_f32_u32 %3, %9\n v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %8\n v_cvt_f32_u32 %7, %9"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1));) }
		if (WHICH == 29) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n"
		                                     "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc"
		                                     : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779B9u), "v"(0x7F4A7C15u) : "vcc");) }
		if (WHICH == 30) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
		                                     "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8"
		                                     : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4), "+v"(l5), "+v"(l6), "+v"(l7) : "v"(l0));) }
		if (WHICH == 31) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_bfe_u32 %4, %4, 3, 9\n v_lshrrev_b32 %5, 3, %5\n v_bfe_u32 %6, %6, 3, 9\n v_lshrrev_b32 %7, 3, %7"
		                                     : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779B9u));) }
		if (WHICH == 18) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
		                                     "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
		                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
	}
	t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
	if ((threadIdx.x & 63) == 0) { Stamp s; s.ticks = t1 - t0; s.real = r1 - r0; stamps[(blockIdx.x * 256 + threadIdx.x) >> 6] = s; }
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float) (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float) (u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7)
	                                        + (float) (l0 ^ l1 ^ l2 ^ l3 ^ l4 ^ l5 ^ l6 ^ l7) + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int W> void run(const char *name, float *d_out, Stamp *d_stamps, int cus, int lanes = 64)
{
	printf("%-28s", name);
	for (int wps : {1, 2, 4, 8}) {
		const int blocks = cus * wps;                 // a 256-thread workgroup puts one wave on each SIMD of its CU
		const int iters = 40000 / wps;                // 64 instructions per iteration
		for (int warm = 0; warm < 3; warm++)
			hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d_out, d_stamps, iters, 1.0f, lanes);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d_out, d_stamps, iters, 1.0f, lanes);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<Stamp> st((size_t) blocks * 4);
		hipMemcpy(st.data(), d_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
		std::vector<double> ticks, ghz;
		for (auto &s : st) { ticks.push_back((double) s.ticks); ghz.push_back((double) s.ticks / (double) s.real * 0.1); }
		std::sort(ticks.begin(), ticks.end()); std::sort(ghz.begin(), ghz.end());
		/* cycles the SIMD spent per instruction = wall time x the clock the waves measured / instructions per SIMD
		 * (a wave's own tick count only covers the time it was resident: with more waves than fit at once, or a
		 * dispatcher that fills CUs unevenly, it under-counts -- the wall time does not) */
		const double instr_per_simd = (double) wps * iters * 64.0;
		const double clock_ghz = ghz[ghz.size() / 2];
		const double real_cyc = ms * 1e-3 * clock_ghz * 1e9 / instr_per_simd;
		const double wave_cyc = ticks[ticks.size() / 2] / ((double) iters * 64.0);
		printf("  | %dw: %5.2f clk @%.2f GHz (wave: %5.1f)", wps, real_cyc, clock_ghz, wave_cyc);
		hipEventDestroy(e0); hipEventDestroy(e1);
	}
	printf("\n");
}

int main(int argc, char **argv)
{
	const bool only_new = argc > 1;      /* any argument: only the kinds added in round 6 */
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	float *d; hipMalloc(&d, sizeof(float) * cus * 8 * 256);
	Stamp *s; hipMalloc(&s, sizeof(Stamp) * cus * 8 * 4);
	printf("%s, %d CUs; cycles per wave64 instruction per SIMD (s_memtime), clock = d(s_memtime)/d(s_memrealtime) x 100 MHz\n", p.name, cus);
	run<0>("v_fma_f32", d, s, cus);
	run<32>("v_fma_mix_f32 (f16 x f32 + f32)", d, s, cus);
	run<33>("v_cvt_f32_ubyteN", d, s, cus);
	run<34>("v_perm_b32", d, s, cus);
	run<35>("v_alignbit_b32", d, s, cus);
	run<36>("v_min3_f32", d, s, cus);
	if (only_new) return 0;
	run<1>("v_mul_f32", d, s, cus);
	run<2>("v_add_f32", d, s, cus);
	run<13>("v_max_f32", d, s, cus);
	run<12>("v_add_u32", d, s, cus);
	run<25>("v_sub_f32", d, s, cus);
	run<27>("v_mov_b32", d, s, cus);
	run<21>("v_min_f32", d, s, cus);
	run<19>("v_max3_f32", d, s, cus);
	run<20>("v_med3_f32", d, s, cus);
	run<23>("v_xor_b32", d, s, cus);
	run<24>("v_lshlrev_b32", d, s, cus);
	run<31>("v_and/or/bfe/lshr", d, s, cus);
	run<29>("v_add_co/addc_co_u32", d, s, cus);
	run<30>("v_lshl_add_u64", d, s, cus);
	run<26>("v_mul_hi_u32", d, s, cus);
	run<28>("v_cvt_f32_u32", d, s, cus);
	run<22>("v_cmp->sgpr + v_cndmask", d, s, cus);
	run<18>("v_cmp_f32 (vcc)", d, s, cus);
	run<17>("v_cndmask_b32 (vcc)", d, s, cus);
	run<11>("v_cmp + v_cndmask pairs", d, s, cus);
	run<3>("v_pk_fma_f32", d, s, cus);
	run<4>("v_pk_mul_f32", d, s, cus);
	run<5>("v_fma_f64", d, s, cus);
	run<6>("v_mul_f64", d, s, cus);
	run<7>("v_add_f64", d, s, cus);
	run<15>("v_cvt f64<->f32", d, s, cus);
	run<8>("v_mad_u64_u32", d, s, cus);
	run<14>("v_mul_lo_u32", d, s, cus);
	run<9>("v_rcp_f32", d, s, cus);
	run<10>("v_sqrt_f32", d, s, cus);
	run<16>("v_rcp_f64", d, s, cus);
	run<0>("v_fma_f32, lanes 0-31 only", d, s, cus, 32);
	run<5>("v_fma_f64, lanes 0-31 only", d, s, cus, 32);
	return 0;
}
