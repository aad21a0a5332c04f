# timing-only ablations (frames are WRONG with any of them): what would a section cost nothing be worth?
s = open("rt_kernels.hip").read()
m = open("rt_math.hip.h").read()
# A1: cheap (inexact) u64 -> f32 conversion
m = m.replace("	return (float) bits * 0x1p-64f;       /* == (float)bits / (float)UINT64_MAX, the divisor is 2^64 */",
"""#ifdef ABL_CVT
	return (float) (uint32_t) (bits >> 32) * 0x1p-32f;
#else
	return (float) bits * 0x1p-64f;
#endif""")
assert "ABL_CVT" in m
# A1b: whole RNG = one LCG step
m = m.replace("""	state += 0x60bee2bee120fc15ull;
	uint64_t bits = fold_mul<0x1b03738712fad5c9ull>(fold_mul<0xa3b195354a39b70dull>(state));""",
"""	state += 0x60bee2bee120fc15ull;
#ifdef ABL_RNG
	uint64_t bits = state * 0x9E3779B97F4A7C15ull;
#else
	uint64_t bits = fold_mul<0x1b03738712fad5c9ull>(fold_mul<0xa3b195354a39b70dull>(state));
#endif""")
assert "ABL_RNG" in m
# A2: taps: no object loop
s = s.replace("""				const Hit hit = FAST ? NEAREST_HIT_TUNED(sc, n, o, dn, false) : nearest_hit(sc, n, o, dn);
				tap_answer(meta, hit.obj);""",
"""#ifdef ABL_TAPTRACE
				tap_answer(meta, dn.x > 0.3f ? light_obj : -1);
#else
				const Hit hit = FAST ? NEAREST_HIT_TUNED(sc, n, o, dn, false) : nearest_hit(sc, n, o, dn);
				tap_answer(meta, hit.obj);
#endif""")
assert "ABL_TAPTRACE" in s
# A3: sphere roots in fp32
s = s.replace("""	if (wave_all(rp.den_ok && discr >= 0x1p-126f && discr <= 0x1p+120f && __builtin_fabsf(b) >= 0x1p-90f)) {""",
"""#ifdef ABL_F32ROOTS
	{
		const float root = __builtin_sqrtf(discr), nbf = -b, inv = 1.0f / (2.0f * rp.dd);
		float t = (nbf - root) * inv;
		if (t < 0) t = (nbf + root) * inv;
		if (t < 0) return false;
		t_entry = t;
		return true;
	}
#endif
	if (wave_all(rp.den_ok && discr >= 0x1p-126f && discr <= 0x1p+120f && __builtin_fabsf(b) >= 0x1p-90f)) {""")
assert "ABL_F32ROOTS" in s
# A4: sky texel: constant address
s = s.replace("""					sky1 = sky_texel<FAST>(L, hp); rec1 |= REC_LAST | REC_SKY;                  /* main.c:163-172 */""",
"""#ifdef ABL_SKY
					sky1 = L.sky[lane]; rec1 |= REC_LAST | REC_SKY;
#else
					sky1 = sky_texel<FAST>(L, hp); rec1 |= REC_LAST | REC_SKY;                  /* main.c:163-172 */
#endif""")
assert "ABL_SKY" in s
# A5: bounce-ray trace: no object loop at all (everything misses after bounce 0)
s = s.replace("""				const Hit hit = FAST ? NEAREST_HIT_TUNED(sc, n, ray_o, dn, true) : nearest_hit(sc, n, ray_o, dn);
				hobj = hit.obj; hn = hit.n;""",
"""#ifdef ABL_BOUNCETRACE
				Hit hit; hit.obj = dn.y > 0.2f ? -1 : 3; hit.t = 1.0f; hit.n = mk3(0, 1, 0);
#else
				const Hit hit = FAST ? NEAREST_HIT_TUNED(sc, n, ray_o, dn, true) : nearest_hit(sc, n, ray_o, dn);
#endif
				hobj = hit.obj; hn = hit.n;""")
assert "ABL_BOUNCETRACE" in s
# A6: no in-order sum work (slots are released, nothing is added)
s = s.replace("""			const bool active = (int) j < k;
			const bool offset_slot = boundary && (int) j == lastpos + 1 && active;""",
"""			const bool active = (int) j < k;
			const bool offset_slot = boundary && (int) j == lastpos + 1 && active;
#ifdef ABL_SUM
			if (offset_slot) { float *dst = L.frame + (size_t) xb * 3; dst[0] = 0.5f; }
			if (active) W.win[0][e] = __uint_as_float(WF_EMPTY);
			if (active && (int) j == k - 1) W.s_drained[g] = d + (unsigned int) k;
			again = false; wave_fence(); break;
#endif""")
assert "ABL_SUM" in s
# A7: specular branch: no normalisation
s = s.replace("""				out_dir = unit3_sel<FAST>(lin2(scatter, madd3(hdir, nneg, f), m0.w, 1.0f));""",
"""#ifdef ABL_SPEC
				out_dir = lin2(scatter, madd3(hdir, nneg, f), m0.w, 1.0f);
#else
				out_dir = unit3_sel<FAST>(lin2(scatter, madd3(hdir, nneg, f), m0.w, 1.0f));
#endif""")
assert "ABL_SPEC" in s
open("rt_kernels.hip", "w").write(s)
open("rt_math.hip.h", "w").write(m)
