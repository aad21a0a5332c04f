# experiment: the cluster cull for scenes of 16 objects and more (shipped: 65 and more, below that scenes are compiled)
s = open("rt_cull.h").read()
assert s.count("#define RT_CULL_MIN_OBJECTS 65") == 1
s = s.replace("#define RT_CULL_MIN_OBJECTS 65", "#define RT_CULL_MIN_OBJECTS 16")
open("rt_cull.h", "w").write(s)
