// Exhaustive check of lean sequences for sqrt((double) x), x a positive float (the sphere discriminant,
// scene.c:117), against the compiler's IEEE fp64 sqrt.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

template <int CORR, bool SEED32, bool HUPD = true>
__device__ __forceinline__ double root(double x, float xf)
{
	const double y = SEED32 ? (double) __builtin_amdgcn_rsqf(xf) : __builtin_amdgcn_rsq(x);
	double g = x * y, h = 0.5 * y;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	if (HUPD) h = __builtin_fma(h, r, h);
	for (int k = 0; k < CORR; k++) g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	return g;
}

__global__ void __launch_bounds__(256) sweep(unsigned long long *c, unsigned first, unsigned long long count)
{
	for (unsigned long long k = (unsigned long long) blockIdx.x * 256 + threadIdx.x; k < count; k += (unsigned long long) gridDim.x * 256) {
		const float xf = __uint_as_float(first + (unsigned) k);
		const double x = (double) xf, ref = __builtin_sqrt(x);
		if (root<2, false>(x, xf) != ref) atomicAdd(&c[0], 1ull);
		if (root<1, false>(x, xf) != ref) atomicAdd(&c[1], 1ull);
		if (root<0, false>(x, xf) != ref) atomicAdd(&c[2], 1ull);
		if (root<2, true>(x, xf) != ref) atomicAdd(&c[3], 1ull);
		if (root<1, true>(x, xf) != ref) atomicAdd(&c[4], 1ull);
		if (root<1, true, false>(x, xf) != ref) atomicAdd(&c[5], 1ull);
		if (root<1, false, false>(x, xf) != ref) atomicAdd(&c[6], 1ull);
	}
}

int main()
{
	unsigned long long *c; hipMalloc(&c, 7 * 8); hipMemset(c, 0, 7 * 8);
	const unsigned lo = 0x00800000u, hi = 0x3f800000u + (120u << 23);     // normal floats up to 2^120
	hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, c, lo, (unsigned long long) (hi - lo) + 1ull);
	if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
	unsigned long long h[7]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
	printf("inputs: %llu normal floats in [2^-126, 2^120]\n", (unsigned long long) (hi - lo) + 1ull);
	printf("rsq_f64 seed, step, 2 corrections (current) : %llu mismatches\n", h[0]);
	printf("rsq_f64 seed, step, 1 correction            : %llu\n", h[1]);
	printf("rsq_f64 seed, step, 0 corrections           : %llu\n", h[2]);
	printf("rsq_f32 seed, step, 2 corrections           : %llu\n", h[3]);
	printf("rsq_f32 seed, step, 1 correction            : %llu\n", h[4]);
	printf("rsq_f32 seed, step without h update, 1 corr.: %llu\n", h[5]);
	printf("rsq_f64 seed, step without h update, 1 corr.: %llu\n", h[6]);
	return 0;
}
