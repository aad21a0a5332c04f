/*
 * rt_math.hip.h -- device building blocks for the path-tracing kernels (gfx950 only).
 *
 * Parity rules: every float product/sum rounded separately (no FMA
 * contraction -- enforced here by the pragma and by -ffp-contract=off on the command line),
 * IEEE-correct `/` and sqrt (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt), and the reference's
 * double-precision islands evaluated in fp64.  Each function names the reference code it follows.
 */
#ifndef RT_MATH_HIP_H
#define RT_MATH_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define RT_DEV __device__ __forceinline__

struct V3 { float x, y, z; };

RT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

/* vector.c:148-155 combine(u, v, a, b) */
RT_DEV V3 lin2(V3 u, V3 v, float a, float b)
{
	return mk3(u.x * a + v.x * b, u.y * a + v.y * b, u.z * a + v.z * b);
}
/* combine(u, v, 1, 1) and combine(u, v, 1, -1): products by +-1 are exact */
RT_DEV V3 add3(V3 u, V3 v) { return mk3(u.x + v.x, u.y + v.y, u.z + v.z); }
RT_DEV V3 sub3(V3 u, V3 v) { return mk3(u.x - v.x, u.y - v.y, u.z - v.z); }
/* combine(u, v, 1, b) */
RT_DEV V3 madd3(V3 u, V3 v, float b) { return mk3(u.x + v.x * b, u.y + v.y * b, u.z + v.z * b); }
RT_DEV V3 scale3(V3 v, float f) { return mk3(v.x * f, v.y * f, v.z * f); }          /* vector.c:140 */
RT_DEV V3 neg3(V3 v) { return mk3(-v.x, -v.y, -v.z); }                              /* scalev(v,-1) */
RT_DEV V3 had3(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }         /* vector.c:366 */
RT_DEV float dot3(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }         /* vector.c:361 */

/* vector.c:129-138.  (float)sqrt((double)f) == correctly rounded sqrtf(f).  The epsilon test is the
 * reference's double comparison `norm < 0.00001` (norm >= 0 or NaN, so the lower bound is moot). */
RT_DEV V3 unit3(V3 v)
{
	float len = __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
	if ((double) len < 0.00001)
		return v;
	return mk3(v.x / len, v.y / len, v.z / len);
}

RT_DEV float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }   /* vector.c:52 */
RT_DEV float clamp11(float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); }
RT_DEV bool  tiny_f(float f) { return (double) f < 0.0001 && (double) f > -0.0001; } /* vector.c:79 */

/* ---- RNG: utils.c:60-75 ---------------------------------------------------------------- */

RT_DEV uint64_t fold_mul(uint64_t a, uint64_t b) { return __umul64hi(a, b) ^ (a * b); }

RT_DEV float rng_draw(uint64_t &state)
{
	state += 0x60bee2bee120fc15ull;
	uint64_t bits = fold_mul(fold_mul(state, 0xa3b195354a39b70dull), 0x1b03738712fad5c9ull);
	return (float) bits * 0x1p-64f;       /* == (float)bits / (float)UINT64_MAX, the divisor is 2^64 */
}

/* `counter` mode path seed -- must match oracle/rt_oracle.c orc_path_seed() */
RT_DEV uint64_t path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index)
{
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t) sample_index << 32) | (uint64_t) pixel_index);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

/* vector.c:99-111: x, y, z drawn in that order */
RT_DEV V3 rng_direction(uint64_t &state)
{
	float x = rng_draw(state) * 2.0f - 1.0f;
	float y = rng_draw(state) * 2.0f - 1.0f;
	float z = rng_draw(state) * 2.0f - 1.0f;
	return unit3(mk3(x, y, z));
}

#endif
