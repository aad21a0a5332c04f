// Exhaustive check of lean correctly-rounded sqrt / reciprocal-of-sqrt sequences against the compiler's
// IEEE sqrtf and 1.0f / x, for every float in [2^-30, 2^60].  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

// A: v_sqrt_f32 + the +-1 ulp residual selection hipcc emits, without its input scaling / class fix-up
__device__ __forceinline__ float sqrt_a(float x)
{
	const float s = __builtin_amdgcn_sqrtf(x);
	const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
	float r = __builtin_fmaf(-sm, s, x) <= 0.0f ? sm : s;
	return __builtin_fmaf(-sp, s, x) > 0.0f ? sp : r;
}
// B: rsq-seeded Goldschmidt, one coupled step + one residual correction; h ~ 1/(2 sqrt x) comes for free
__device__ __forceinline__ float sqrt_b(float x, float &h_out)
{
	const float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	const float e = __builtin_fmaf(-h, g, 0.5f);
	g = __builtin_fmaf(g, e, g);
	h = __builtin_fmaf(h, e, h);
	g = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
	h_out = h;
	return g;
}
__device__ __forceinline__ float newton(float d, float r) { return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r); }

__global__ void __launch_bounds__(256) sweep(unsigned long long *c, unsigned lo_bits, unsigned long long count)
{
	for (unsigned long long k = (unsigned long long) blockIdx.x * 256 + threadIdx.x; k < count; k += (unsigned long long) gridDim.x * 256) {
		const float x = __uint_as_float(lo_bits + (unsigned) k);
		const float ref = __builtin_sqrtf(x);
		const float rref = 1.0f / ref;
		float h;
		const float a = sqrt_a(x), b = sqrt_b(x, h);
		unsigned long long m = 0;
		if (a != ref) atomicAdd(&c[0], 1ull);
		if (b != ref) atomicAdd(&c[1], 1ull);
		if (newton(ref, h + h) != rref) atomicAdd(&c[2], 1ull);                         // reciprocal of len from h: one Newton step
		if (newton(ref, newton(ref, h + h)) != rref) atomicAdd(&c[3], 1ull);            // two steps
		if (newton(ref, __builtin_amdgcn_rcpf(ref)) != rref) atomicAdd(&c[4], 1ull);    // rcp + one step (current code)
		if (newton(ref, __builtin_amdgcn_rsqf(x)) != rref) atomicAdd(&c[5], 1ull);      // rsq(x) + one step
		(void) m;
	}
}

int main()
{
	unsigned long long *c; hipMalloc(&c, 6 * 8); hipMemset(c, 0, 6 * 8);
	const unsigned lo = 0x3f800000u - (30u << 23), hi = 0x3f800000u + (60u << 23);
	hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, c, lo, (unsigned long long) (hi - lo) + 1ull);
	if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
	unsigned long long h[6]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
	printf("inputs: %llu floats in [2^-30, 2^60]\n", (unsigned long long) (hi - lo) + 1ull);
	printf("A  v_sqrt + +-1ulp selection        != sqrtf : %llu\n", h[0]);
	printf("B  rsq Goldschmidt + 1 correction   != sqrtf : %llu\n", h[1]);
	printf("1/len: 2h + 1 Newton step           != 1/len : %llu\n", h[2]);
	printf("1/len: 2h + 2 Newton steps          != 1/len : %llu\n", h[3]);
	printf("1/len: v_rcp + 1 Newton step        != 1/len : %llu\n", h[4]);
	printf("1/len: v_rsq(x) + 1 Newton step     != 1/len : %llu\n", h[5]);
	return 0;
}
