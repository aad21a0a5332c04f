/* rt_pack.h -- host-side packing of a Scene for the kernels (rt_pack.cpp); not installed. */
#ifndef RT_PACK_H
#define RT_PACK_H

#include <string>
#include <vector>
#include "../../include/rt_types.h"
#include "rt_device.h"

struct rt_packed_scene_info {
	int   light_index;          /* first object with emission_power > 0 (main.c:140-146), -1: none */
	float light_pos[3];         /* its origin_of() (scene.c:10-15); zeros without one */
	bool  only_light_emits;     /* no other object's emission has a non-zero component, and the emitter's is finite */
	bool  fast_ok;              /* every coordinate inside the windows of the tuned intersection (rt_kernels.hip prepare_ray) */
};
/* geom / shade get max(num_objects, 1) records */
void rt_pack_scene(const Scene *scene, std::vector<rt_geom> &geom, std::vector<rt_shade> &shade, rt_packed_scene_info *info);
std::string rt_jit_scene_header(const rt_geom *geom, int n, int light_index, const float light_pos[3], int only_light_emits);

#endif
