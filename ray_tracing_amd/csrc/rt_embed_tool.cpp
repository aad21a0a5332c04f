/* rt_embed_tool -- build step: prints the header a scene-specialised kernel is compiled with (rt_pack.cpp), for the scenes
 * whose kernels are compiled when the library is built (Makefile: the three scene files the reference ships).
 * usage: rt_embed_tool <scene.txt>     Host only: parses with the library's loader, packs as rt_set_scene() does. */
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/rt_hip.h"
#include "rt_pack.h"

int main(int argc, char **argv)
{
	if (argc != 2) { fprintf(stderr, "usage: %s <scene.txt>\n", argv[0]); return 2; }
	Scene *scene = (Scene *) calloc(1, sizeof(Scene));
	if (!scene || rt_parse_scene_file(argv[1], scene) != RT_OK) { fprintf(stderr, "%s: cannot parse %s\n", argv[0], argv[1]); return 1; }
	std::vector<rt_geom> geom; std::vector<rt_shade> shade; rt_packed_scene_info info;
	rt_pack_scene(scene, geom, shade, &info);
	if (scene->num_objects < 1 || scene->num_objects > 64 || !info.fast_ok) { fprintf(stderr, "%s: %s cannot be specialised\n", argv[0], argv[1]); return 1; }
	fputs(rt_jit_scene_header(geom.data(), scene->num_objects, info.light_index, info.light_pos, info.only_light_emits ? 1 : 0).c_str(), stdout);
	free(scene);
	return 0;
}
