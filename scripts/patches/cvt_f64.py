m = open("rt_math.hip.h").read()
old = "	return (float) bits * 0x1p-64f;       /* == (float)bits / (float)UINT64_MAX, the divisor is 2^64 */"
new = """#ifdef RT_CVT_F64
	/* RN24(bits) through fp64: hi * 2^32 + lo rounded to 53 bits, then to 24.  The two roundings give RN24(bits) unless the
	 * 53-bit value is exactly the midpoint of two floats (low 29 bits of its significand = 1 << 28): 2^-29 of the draws, which
	 * take the exact integer route with their whole wave */
	const double dd_ = __builtin_fma((double) (uint32_t) (bits >> 32), 0x1p+32, (double) (uint32_t) bits);
	float f_ = (float) dd_;
	if (__builtin_expect(__ballot(((uint32_t) __double_as_longlong(dd_) & 0x1fffffffu) == 0x10000000u) != 0ull, 0)) {
		asm volatile("" ::: "memory");
		f_ = (float) bits;
	}
	return f_ * 0x1p-64f;
#else
	return (float) bits * 0x1p-64f;       /* == (float)bits / (float)UINT64_MAX, the divisor is 2^64 */
#endif"""
assert old in m
m = m.replace(old, new)
open("rt_math.hip.h", "w").write(m)
