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

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#ifndef __HIPCC_RTC__
#include <stdint.h>
#else   /* hiprtc (rt_compile_scene): no host headers */
#ifndef RT_RTC_STDINT
#define RT_RTC_STDINT
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
#endif
#endif

#pragma clang fp contract(off)

#define RT_DEV __device__ __forceinline__
#ifndef RT_STAT_UNIT_SLOW
#define RT_STAT_UNIT_SLOW do {} while (0)
#endif

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
/* the same predicate without fp64: the largest float below the double 0.0001 is 0x38D1B717 (NaN -> false) */
RT_DEV bool  tiny_f_fast(float f) { return __builtin_fabsf(f) <= __uint_as_float(0x38D1B717u); }


/* ---- exact division with a shared reciprocal ---------------------------------------------------
 * hipcc lowers an IEEE `n / d` to  v_div_scale x2, v_rcp, a Newton step on the reciprocal, a product,
 * two residual corrections (the last inside v_div_fmas) and v_div_fixup.  When neither operand needs
 * rescaling -- both inside the exponent window below, so that every intermediate is a normal number and
 * the sequence is invariant under scaling by powers of two -- much less is needed on gfx950:
 *     r = rcp(d); r += r*(1 - d*r);          is RN(1/d) for EVERY significand of d, and
 *     q = n*r;    q += r*(n - d*q)           is RN(n/d) for EVERY pair of significands (n, d)
 * (fused multiply-adds).  Both statements are verified exhaustively on the device -- all 2^23
 * reciprocals, all 2^46 significand pairs, against `/` -- by rt_selftest(5) (tests/test_gpu_selftest.py
 * runs a stratified subset on every run, the full sweep takes 52 s; scripts/ubench/div_exhaustive.hip).
 * The reciprocal depends on d only, so a ray that divides many numerators by the same three direction
 * components (the slab test, scene.c:31-59), or a vector divided by its length (vector.c:134-136),
 * pays for it once and each quotient costs three instructions.  Operands outside the window (zeros,
 * denormals, infinities, NaN included) take the ordinary `/`.
 */
RT_DEV float rcp_refined(float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e  = __builtin_fmaf(-d, r0, 1.0f);
	return __builtin_fmaf(e, r0, r0);
}

RT_DEV float div_by_refined(float n, float d, float r)
{
	const float q = n * r;
	return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}

/* |x| in [2^-30, 2^20]: a denominator for which rcp_refined/div_by_refined are exact (with a numerator
 * accepted by num_in_window: quotient >= 2^-120 and residual granularity >= 2^-147, all normal/exact) */
RT_DEV bool den_in_window(float x)
{
	const float a = __builtin_fabsf(x);
	return a >= 0x1p-30f && a <= 0x1p+20f;
}
/* |x| in [2^-60, 2^30]; zero, NaN and infinities are outside */
RT_DEV bool num_in_window(float lo_abs, float hi_abs) { return lo_abs >= 0x1p-60f && hi_abs <= 0x1p+30f; }

/* sqrtf for x in [2^-30, 2^60]: rsq-seeded coupled (Goldschmidt) step + one residual correction.  Equal
 * to the correctly rounded sqrtf for every float in that range (exhaustive: rt_selftest(6)); hipcc's own
 * expansion spends twice the instructions on input scaling, +-1 ulp selection and class fix-ups. */
RT_DEV float sqrt_in_window(float x)
{
	const float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	const float e = __builtin_fmaf(-h, g, 0.5f);
	g = __builtin_fmaf(g, e, g);
	h = __builtin_fmaf(h, e, h);
	return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}

/* 1 / (double)(2*a) for a = dot(d, d) of a normalised direction (scene.c:93,117-118): with e = a - 1 (exact, and
 * |e| <= 2^-18) the correctly rounded reciprocal is 0.5*(1 - e + e*e) -- the next term of the series is below
 * 2^-55 -- checked against `/` for all 97 floats in that range by rt_selftest(1).  Replaces v_rcp_f64 + two
 * Newton steps per ray. */
RT_DEV bool near_one(float a) { return __builtin_fabsf(a - 1.0f) <= 0x1p-18f; }
RT_DEV double rcp_twice_near_one(float a)
{
	const double e = (double) (a - 1.0f);
	return 0.5 * __builtin_fma(e, e, 1.0 - e);
}

RT_DEV double div_by_refined64(double n, double d, double r)
{
	const double q = n * r;
	return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

/* avgv(): x / 3.0f (vector.c:89-92) as div_by_refined with the reciprocal RN(1/3) as a literal; +0 is exact
 * too (see rt_kernels.hip prepare_ray); anything else outside the numerator window takes the wave through `/` */
RT_DEV float third_of(float x)
{
	if (__ballot(!(x == 0.0f ? !__builtin_signbit(x) : (__builtin_fabsf(x) >= 0x1p-100f && __builtin_fabsf(x) <= 0x1p+30f))) == 0ull)
		return div_by_refined(x, 3.0f, __uint_as_float(0x3eaaaaabu));
	return x / 3.0f;
}

/* vector.c:129-138 with the three divisions sharing one reciprocal.  The tuned form needs the squared
 * length in [2^-30, 2^60] (so the length is >= 2^-15 > 0.00001: the reference's epsilon branch is not
 * taken) and every component at least 2^-60 in magnitude (a +-0 / denormal numerator keeps its sign and
 * rounding only through `/`); one lane outside sends its wave through the reference-order form. */
RT_DEV V3 unit3_fast(V3 v)
{
	const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
	const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z));
	if (__ballot(!(s2 >= 0x1p-30f && s2 <= 0x1p+60f && lo >= 0x1p-60f)) == 0ull) {
		const float len = sqrt_in_window(s2);
		const float r = rcp_refined(len);
		return mk3(div_by_refined(v.x, len, r), div_by_refined(v.y, len, r), div_by_refined(v.z, len, r));
	}
	RT_STAT_UNIT_SLOW;
	const float len = __builtin_sqrtf(s2);
	if (len <= __uint_as_float(0x3727C5ACu))      /* `(double) len < 0.00001`: 0x3727C5AC is the largest float below it */
		return v;
	return mk3(v.x / len, v.y / len, v.z / len);
}

/* ---- RNG: utils.c:60-75 ---------------------------------------------------------------- */

/* hi64(a*B) ^ lo64(a*B) with the four 32x32 partial products formed once (each line is one
 * v_mad_u64_u32); written out because `__umul64hi(a,b) ^ (a*b)` makes the compiler form them twice.
 * The multiplier is a template argument: the two the generator uses (utils.c:64,67) are literals, and the inline asm
 * below takes the multiplier's low word in a SCALAR register -- a lane-varying value there would silently be replaced
 * by its first lane's. */
template <uint64_t B>
RT_DEV uint64_t fold_mul(uint64_t a)
{
	constexpr uint32_t b0 = (uint32_t) B, b1 = (uint32_t) (B >> 32);
	const uint32_t a0 = (uint32_t) a, a1 = (uint32_t) (a >> 32);
	const uint64_t p0 = (uint64_t) a0 * b0;
	const uint64_t x  = (uint64_t) a0 * b1 + (p0 >> 32);          /* fits 64 bits */
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__)
	/* a1*b0 + x can carry out of 64 bits: v_mad_u64_u32 reports it in its wave64 carry mask (the compiler has no
	 * pattern for that carry, so it would split x into halves and add them separately: two moves and a 64-bit add
	 * more per product) */
	uint64_t y, carry, unused;
	asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(y), "=&s"(carry) : "v"(a1), "s"(b0), "v"(x));
	const uint64_t h = (uint64_t) a1 * b1 + (y >> 32);            /* high 64 bits of a*B, but for that carry (weight 2^96) */
	uint32_t h1;
	asm("v_addc_co_u32 %0, %1, 0, %2, %3" : "=v"(h1), "=&s"(unused) : "v"((uint32_t) (h >> 32)), "s"(carry));
	return ((uint64_t) (h1 ^ (uint32_t) y) << 32) | (uint64_t) ((uint32_t) h ^ (uint32_t) p0);
#else
	const uint64_t m = (uint64_t) a1 * b0;
	const uint64_t y = m + x;                                     /* may wrap: the carry has weight 2^96 */
	const uint64_t h = (uint64_t) a1 * b1 + (y >> 32) + ((uint64_t) (y < m) << 32);
	return ((uint64_t) ((uint32_t) (h >> 32) ^ (uint32_t) y) << 32) | (uint64_t) ((uint32_t) h ^ (uint32_t) p0);
#endif
}

/* (float) bits * 2^-64  [== (float) bits / (float) UINT64_MAX (utils.c:74): the divisor is 2^64] */
RT_DEV float unit_float_of_bits(uint64_t bits)
{
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__)
	/* As the compiler converts a 64-bit integer -- normalise, keep the top 32 bits with everything below them OR-ed into the
	 * last one, round those to 24 with v_cvt_f32_u32, scale with v_ldexp_f32 -- except that the 2^-64 goes into that same ldexp
	 * instead of a multiply behind it (both scalings are exact: a non-zero result is at least 2^-64).  One instruction less per
	 * draw, fourteen draws per bounce: C1 -1.1 %, C3 -2.7 % (profiles/r04/ab_draw_ldexp.txt); rt_selftest(7) compares the two.
	 * (Written as instructions: from C the compiler turns both minima into a compare and a select.) */
	const uint32_t hi = (uint32_t) (bits >> 32);
	uint32_t lead, sticky;
	asm("v_ffbh_u32 %0, %1\n\tv_min_u32 %0, 32, %0" : "=&v"(lead) : "v"(hi));      /* leading zeros of the high word; 32 when it is zero */
	const uint64_t norm = bits << lead;
	asm("v_min_u32 %0, 1, %1" : "=v"(sticky) : "v"((uint32_t) norm));
	const uint32_t top = (uint32_t) (norm >> 32) | sticky;
	return __builtin_ldexpf((float) top, -32 - (int) lead);
#else
	return (float) bits * 0x1p-64f;
#endif
}

RT_DEV float rng_draw(uint64_t &state)
{
	state += 0x60bee2bee120fc15ull;
	return unit_float_of_bits(fold_mul<0x1b03738712fad5c9ull>(fold_mul<0xa3b195354a39b70dull>(state)));
}

/* `counter` mode path seed -- must match oracle/rt_oracle.c orc_path_seed() */
RT_DEV uint64_t path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index)
{
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t) sample_index << 32) | (uint64_t) pixel_index);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

/* unit3_fast for a vector of `draw * 2 - 1` components (draw in [0, 1]): each is +0 -- never -0: RN(1 - 1) -- or at
 * least 2^-24 in magnitude, and the squared length is at most 3, so of unit3_fast's window only the lower bound of
 * the squared length is left to test.  A +0 numerator is exact in div_by_refined: q = +0, fma(-len, +0, +0) = +0,
 * fma(+0, r, +0) = +0 = 0 / len.  (rt_selftest(7) compares with unit3 on such vectors.) */
RT_DEV V3 unit3_of_draws(V3 v)
{
	const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
	if (__ballot(!(s2 >= 0x1p-30f)) == 0ull) {
		const float len = sqrt_in_window(s2);
		const float r = rcp_refined(len);
		return mk3(div_by_refined(v.x, len, r), div_by_refined(v.y, len, r), div_by_refined(v.z, len, r));
	}
	return unit3(v);
}

/* vector.c:99-111: x, y, z drawn in that order; rng_vector is random_vector(), before normalize() */
RT_DEV V3 rng_vector(uint64_t &state)
{
	/* draw * 2 - 1 (vector.c:101-103) as one fused multiply-add: doubling is exact, so the one rounding of fma(d, 2, -1) is the
	 * rounding of the subtraction -- the same bits, one instruction instead of two, twelve times per bounce */
	float x = __builtin_fmaf(rng_draw(state), 2.0f, -1.0f);
	float y = __builtin_fmaf(rng_draw(state), 2.0f, -1.0f);
	float z = __builtin_fmaf(rng_draw(state), 2.0f, -1.0f);
	return mk3(x, y, z);
}
template <bool FAST = false>
RT_DEV V3 unit3_of_vector(V3 v) { return FAST ? unit3_of_draws(v) : unit3(v); }
template <bool FAST = false>
RT_DEV V3 rng_direction(uint64_t &state) { return unit3_of_vector<FAST>(rng_vector(state)); }

/* The sign of dot(normalize(v), n) without normalising: `dot(rand_dir, hit.normal) <= 0` (main.c:194) only needs it.
 * With u = normalize(v) as the reference computes it (vector.c:129-138: three correctly rounded quotients by the rounded
 * length, or v itself below the epsilon), float dot(u, n) is within 1e-6 of (v.n)/|v| for a unit normal, and float
 * dot(v, n) within 4e-7 |v| of v.n: where |dot(v, n)| > 1e-5 |v| both have the sign of v.n.  v = random_vector() has
 * components in [-1, 1], so |v| <= sqrt(3) and |dot(v, n)| > 1.75e-5 is enough.  Answers whether that holds; the
 * caller falls back to the real thing for the whole wave otherwise (one tap in 3 x 10^4). */
RT_DEV bool side_is_certain(float dot_vn) { return dot_vn * dot_vn > 3.1e-10f; }

#endif
