# experiment: the sky texel gathers as non-temporal loads (they miss L2 anyway -- a 100 MB skybox -- and turn each XCD's L2 over
# every 45 us, so a frame line never lives long enough to collect its neighbours' pixel writes)
s = open("rt_kernels.hip").read()
old = "	return L.sky[(uint32_t) ((face * L.sky_h + y) * L.sky_w + x)];"
assert s.count(old) == 1
s = s.replace(old, "	return __builtin_nontemporal_load(L.sky + (uint32_t) ((face * L.sky_h + y) * L.sky_w + x));")
open("rt_kernels.hip", "w").write(s)
