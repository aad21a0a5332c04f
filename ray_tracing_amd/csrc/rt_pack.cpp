/*
 * rt_pack.cpp -- Scene (the reference's 68-byte AoS records, scene.h:24-36) -> the packed records the kernels read
 * (rt_device.h), and the header a scene-specialised kernel is compiled with (rt_jit.cpp).  Host only, no HIP: the
 * library calls it from rt_set_scene(), and the build-time tool csrc/rt_embed_tool.cpp calls it to generate the headers of
 * the shipped scenes, whose kernels are compiled when the library is built (Makefile) and embedded in it.
 *
 * Packing folds every ray-independent term of the reference's inner loops, with the reference's own float roundings:
 *   cube far corner  origin*1 + size*1                         scene.c:27
 *   sphere r*r                                                  scene.c:112
 *   f0, 1-f0, albedo*(1-metallic), emission_color*power, metallic > 0.001
 *                                                               main.c:219-221,128,248,232,241
 *   first emitter and origin_of() of it                         main.c:140-146, scene.c:10-15
 * Compiled with -ffp-contract=off like everything else.
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_types.h"
#include "rt_device.h"
#include "rt_pack.h"

#pragma clang fp contract(off)

void rt_pack_scene(const Scene *scene, std::vector<rt_geom> &geom, std::vector<rt_shade> &shade, rt_packed_scene_info *info)
{
	const int n = scene->num_objects;
	geom.assign((size_t) (n > 0 ? n : 1), rt_geom());
	shade.assign((size_t) (n > 0 ? n : 1), rt_shade());
	memset(geom.data(), 0, geom.size() * sizeof(rt_geom));
	memset(shade.data(), 0, shade.size() * sizeof(rt_shade));
	int light = -1;
	bool fast_ok = true;
	auto bounded = [](float x) { return x >= -0x1p+29f && x <= 0x1p+29f; };   /* false for NaN */
	/* a slab plane coordinate the tuned slab test accepts: +0, or 2^-76 <= |x| <= 2^29 (then `plane - origin`
	 * is exactly +0 or at least 2^-100 in magnitude unless the origin itself is tiny; see prepare_ray) */
	auto plane_ok = [](float x) { return (x == 0.0f && !std::signbit(x)) || (std::fabs(x) >= 0x1p-76f && std::fabs(x) <= 0x1p+29f); };
	for (int i = 0; i < n; i++) {
		const Object &o = scene->objects[i];
		const Material &m = o.material;
		rt_geom &g = geom[(size_t) i];
		if (o.type == OBJECT_CUBE) {
			g.type = RT_GEOM_CUBE;
			g.a[0] = o.cube.origin.x; g.a[1] = o.cube.origin.y; g.a[2] = o.cube.origin.z;
			g.b0 = o.cube.origin.x * 1.0f + o.cube.size.x * 1.0f;
			g.b1 = o.cube.origin.y * 1.0f + o.cube.size.y * 1.0f;
			g.b2 = o.cube.origin.z * 1.0f + o.cube.size.z * 1.0f;
			/* the tuned slab test assumes lo <= hi (the loader enforces size >= 0, scene.c:593) */
			fast_ok = fast_ok && g.a[0] <= g.b0 && g.a[1] <= g.b1 && g.a[2] <= g.b2 &&
			          plane_ok(g.a[0]) && plane_ok(g.a[1]) && plane_ok(g.a[2]) && plane_ok(g.b0) && plane_ok(g.b1) && plane_ok(g.b2);
		} else if (o.type == OBJECT_SPHERE) {
			g.type = RT_GEOM_SPHERE;
			g.a[0] = o.sphere.center.x; g.a[1] = o.sphere.center.y; g.a[2] = o.sphere.center.z;
			g.b0 = o.sphere.radius * o.sphere.radius;
			fast_ok = fast_ok && bounded(g.a[0]) && bounded(g.a[1]) && bounded(g.a[2]) && bounded(g.b0);
		} else {
			g.type = -1;   /* intersect_object() returns false for unknown types (scene.c:153) */
		}

		rt_shade &s = shade[(size_t) i];
		const float f0d = (float) (0.16 * (double) m.reflectance * (double) m.reflectance);
		const float om  = 1 - m.metallic;
		const float alb[3] = { m.albedo.x, m.albedo.y, m.albedo.z };
		const float ecol[3] = { m.emission_color.x, m.emission_color.y, m.emission_color.z };
		for (int k = 0; k < 3; k++) {
			s.f0[k]           = f0d * om + alb[k] * m.metallic;
			s.one_minus_f0[k] = 1.0f * 1.0f + s.f0[k] * -1.0f;
			s.tint[k]         = alb[k] * om;
			s.emission[k]     = ecol[k] * m.emission_power;
		}
		s.roughness = m.roughness;
		s.is_metal  = ((double) m.metallic > 0.001) ? 1 : 0;
		if (light < 0 && m.emission_power > 0) light = i;
	}
	/* emission = emission_color * emission_power, bit pattern by bit pattern: +-0 adds nothing to a tap sum, anything else
	 * (negative, NaN) does and makes the object one whose index a tap must report */
	bool only = light >= 0;
	for (int i = 0; i < n && only; i++)
		for (int k = 0; k < 3; k++) {
			uint32_t bits_;
			memcpy(&bits_, &shade[(size_t) i].emission[k], 4);
			if (i != light && (bits_ & 0x7fffffffu) != 0u) only = false;
			if (i == light && (bits_ & 0x7f800000u) == 0x7f800000u) only = false;     /* (an infinite or NaN emission: n x e is not the n-fold sum for n = 0) */
		}
	info->light_index = light;
	info->fast_ok = fast_ok;
	info->only_light_emits = only;
	info->light_pos[0] = info->light_pos[1] = info->light_pos[2] = 0.0f;
	if (light >= 0) {             /* origin_of(), scene.c:10-15 */
		const Object &o = scene->objects[light];
		if (o.type == OBJECT_SPHERE) {
			info->light_pos[0] = o.sphere.center.x; info->light_pos[1] = o.sphere.center.y; info->light_pos[2] = o.sphere.center.z;
		} else {
			info->light_pos[0] = o.cube.origin.x * 1.0f + o.cube.size.x * 0.5f;
			info->light_pos[1] = o.cube.origin.y * 1.0f + o.cube.size.y * 0.5f;
			info->light_pos[2] = o.cube.origin.z * 1.0f + o.cube.size.z * 0.5f;
		}
	}
}

static void append_float(std::string &s, float f)
{
	char buf[64];
	snprintf(buf, sizeof(buf), "%af", (double) f);      /* hex float: exact */
	s += buf;
}

/* The header the specialised translation unit includes (nearest_hit_spec in rt_kernels.hip).  Its text is also the key
 * under which compiled scenes are cached and embedded: same text, same kernel.  (The text prints geom.b1 of a sphere as it finds
 * it: 0 as packed -- what the build's tool embeds the shipped scenes under --, and the sphere's half extent for the cull once
 * rt_cull_build() has run, i.e. in scenes of 32 objects and more, whose key therefore also depends on the scene's extent and
 * margin.  Harmless: such scenes are never the embedded ones, and equal scenes still get equal keys.) */
std::string rt_jit_scene_header(const rt_geom *geom, int n, int light_index, const float light_pos[3], int only_light_emits)
{
	std::string h = "/* generated by rt_compile_scene */\n#define SPEC_N " + std::to_string(n) + "\n";
	h += "#define SPEC_LIGHT " + std::to_string(light_index) + "   /* first emitter (main.c:140-146), -1 = none */\n";
	h += "static constexpr float SPEC_LIGHT_POS[3] = {";   /* its origin_of() (scene.c:10-15) */
	for (int k = 0; k < 3; k++) { append_float(h, light_pos[k]); h += k < 2 ? ", " : ""; }
	h += "};\n";
	h += "#define SPEC_ONLY_LIGHT_EMITS " + std::to_string(only_light_emits ? 1 : 0) + "   /* no other object's emission has a non-zero component (rt_device.h) */\n";
	h += "static constexpr int SPEC_T[SPEC_N] = {";
	for (int i = 0; i < n; i++) { h += std::to_string(geom[i].type); h += i + 1 < n ? ", " : ""; }
	h += "};\nstatic constexpr float SPEC_G[SPEC_N][6] = {\n";
	for (int i = 0; i < n; i++) {
		const float v[6] = { geom[i].a[0], geom[i].a[1], geom[i].a[2], geom[i].b0, geom[i].b1, geom[i].b2 };
		h += "\t{";
		for (int k = 0; k < 6; k++) { append_float(h, v[k]); h += k < 5 ? ", " : ""; }
		h += i + 1 < n ? "},\n" : "}\n";
	}
	h += "};\n";
	return h;
}
