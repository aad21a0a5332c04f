# experiment: workgroups of two waves instead of four (a workgroup's LDS and registers are free for the next launch as soon as its
# own two waves are done, not when the slowest of four is)
s = open("rt_kernels.hip").read()
assert s.count("#define RT_BLOCK    256") == 1
s = s.replace("#define RT_BLOCK    256", "#define RT_BLOCK    128")
assert s.count("	if (per_cu > 4) per_cu = 4;") == 1
s = s.replace("	if (per_cu > 4) per_cu = 4;", "	if (per_cu > 8) per_cu = 8;")
old = 'static_assert(4 * sizeof(WaveLDS) + 96 * 17 <= 160 * 1024 / 4,'
assert s.count(old) == 1
s = s.replace(old, 'static_assert(2 * sizeof(WaveLDS) + 96 * 12 <= 160 * 1024 / 8,')
open("rt_kernels.hip", "w").write(s)
a = open("rt_api.cpp").read()
assert a.count("return q == hipErrorNotReady ? 2 : 0;") == 1
a = a.replace("return q == hipErrorNotReady ? 2 : 0;", "return q == hipErrorNotReady ? 4 : 0;")
open("rt_api.cpp", "w").write(a)
