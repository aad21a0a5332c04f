s = open("rt_kernels.hip").read()
old1 = """				const V3 res = scale3(sum, inv_spp);
				float *dst = L.frame + (size_t) xb * 3;
				dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;"""
new1 = """				const V3 res = scale3(sum, inv_spp);
				float *dst = L.frame + (size_t) xb * 3;
				__builtin_nontemporal_store(res.x, dst); __builtin_nontemporal_store(res.y, dst + 1); __builtin_nontemporal_store(res.z, dst + 2);"""
assert old1 in s
s = s.replace(old1, new1)
open("rt_kernels.hip", "w").write(s)
