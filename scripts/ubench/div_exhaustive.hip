// Exhaustive search: for which reciprocal refinements is   q = n*r; q = fma(fma(-d,q,n), r, q)
// (ONE residual correction) the correctly rounded n/d for EVERY pair of float significands?
// All operations are scale-invariant inside the exponent window the kernels use, so testing
// n, d in [1, 2) x [1, 2) (2^46 pairs) covers every in-window operand pair.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)

__device__ __forceinline__ float rcp1(float d) { float r = __builtin_amdgcn_rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r); }
__device__ __forceinline__ float rcp2(float d) { float r = rcp1(d); return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r); }
__device__ __forceinline__ float one_fix(float n, float d, float r) { float q = n * r; return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q); }

// counts[v] = mismatching pairs of variant v; bad_d[v][md] = 1 if denominator md has any mismatch
__global__ void __launch_bounds__(256) sweep(unsigned long long *counts, unsigned char *bad_d, unsigned d_begin, unsigned d_count, unsigned n_step)
{
	const unsigned di = blockIdx.x * 256 + threadIdx.x;
	if (di >= d_count) return;
	const unsigned md = d_begin + di;
	const float d = __uint_as_float(0x3f800000u | md);
	const float r1 = rcp1(d), r2 = rcp2(d), r3 = 1.0f / d;
	unsigned long long c1 = 0, c2 = 0, c3 = 0;
	for (unsigned mn = blockIdx.y; mn < (1u << 23); mn += n_step) {
		const float n = __uint_as_float(0x3f800000u | mn);
		const float ref = n / d;
		c1 += one_fix(n, d, r1) != ref;
		c2 += one_fix(n, d, r2) != ref;
		c3 += one_fix(n, d, r3) != ref;
	}
	if (c1) { atomicAdd(&counts[0], c1); bad_d[md] = 1; }
	if (c2) { atomicAdd(&counts[1], c2); bad_d[(1u << 23) + md] = 1; }
	if (c3) { atomicAdd(&counts[2], c3); bad_d[(2u << 23) + md] = 1; }
	if (blockIdx.y == 0 && __uint_as_float(0x3f800000u | 5u) * r1 != __uint_as_float(0x3f800000u | 5u) / d) atomicAdd(&counts[5], 1ull);   /* control: no correction */
	if (r1 != r3) atomicAdd(&counts[3], 1ull);
	if (r2 != r3) atomicAdd(&counts[4], 1ull);
}

int main(int argc, char **argv)
{
	// argv[1] = log2 of the fraction of numerators to test (0 = all 2^23 per denominator)
	const unsigned n_step = argc > 1 ? 1u << atoi(argv[1]) : 1u;
	unsigned long long *counts; unsigned char *bad;
	hipMalloc(&counts, 6 * sizeof(unsigned long long)); hipMemset(counts, 0, 6 * sizeof(unsigned long long));
	hipMalloc(&bad, 3u << 23); hipMemset(bad, 0, 3u << 23);
	const unsigned chunk = 1u << 18;        // denominators per launch (keeps launches ~1 s)
	for (unsigned d0 = 0; d0 < (1u << 23); d0 += chunk) {
		hipLaunchKernelGGL(sweep, dim3(chunk / 256, n_step > 1 ? 1 : 1), dim3(256), 0, 0, counts, bad, d0, chunk, n_step);
		if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
	}
	unsigned long long h[6]; hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
	std::vector<unsigned char> hb(3u << 23); hipMemcpy(hb.data(), bad, hb.size(), hipMemcpyDeviceToHost);
	const char *names[3] = { "rcp + 1 Newton step", "rcp + 2 Newton steps", "1.0f / d (correctly rounded)" };
	printf("numerators tested per denominator: 2^23 / %u\n", n_step);
	printf("denominators whose reciprocal differs from RN(1/d): 1 step %llu, 2 steps %llu\n", h[3], h[4]);
	printf("control (n*r without correction, one numerator per denominator): %llu mismatches\n", h[5]);
	for (int v = 0; v < 3; v++) {
		unsigned nbad = 0, first = 0, last = 0;
		for (unsigned m = 0; m < (1u << 23); m++) if (hb[((unsigned) v << 23) + m]) { if (!nbad) first = m; last = m; nbad++; }
		printf("%-32s mismatching pairs %llu, denominators involved %u (first 0x%06x last 0x%06x)\n", names[v], h[v], nbad, first, last);
		unsigned shown = 0;
		for (unsigned m = 0; m < (1u << 23) && shown < 12; m++) if (hb[((unsigned) v << 23) + m]) { printf("   d significand 0x%06x\n", m); shown++; }
	}
	return 0;
}
