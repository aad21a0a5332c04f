// Exhaustive check: for dd = dot(d,d) of a normalised direction (|dd - 1| <= 2^-18), is
// 0.5 * (1 - e + e*e), e = dd - 1, the correctly rounded double reciprocal of (double)(2*dd)?  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
__global__ void sweep(unsigned long long *c)
{
	const unsigned k = blockIdx.x * 256 + threadIdx.x;          // 2^22 floats around 1.0
	const float dd = __uint_as_float(0x3f800000u - (1u << 21) + k);
	if (!(__builtin_fabsf(dd - 1.0f) <= 0x1p-18f)) return;
	const double den = (double) (2.0f * dd);
	const double want = 1.0 / den;
	const double e = (double) (dd - 1.0f);
	const double got = 0.5 * __builtin_fma(e, e, 1.0 - e);
	if (want != got) atomicAdd(&c[0], 1ull);
	atomicAdd(&c[1], 1ull);
}
int main()
{
	unsigned long long *c; hipMalloc(&c, 16); hipMemset(c, 0, 16);
	hipLaunchKernelGGL(sweep, dim3((1u << 22) / 256), dim3(256), 0, 0, c);
	hipDeviceSynchronize();
	unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
	printf("dd values tested %llu, mismatches of the series reciprocal: %llu\n", h[1], h[0]);
	return 0;
}
