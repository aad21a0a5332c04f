/* Development probe (CPU, test infrastructure): checks ray_tracing_amd/csrc/rt_lit.h against the oracle.
 *
 * rt_taps_certainly_lit() claims, for a hit point, that every soft-shadow tap from it (main.c:191-206) has the emitter
 * as its nearest hit, so that trace_ray() need not run for those taps.  The probe renders a frame with the CPU oracle,
 * calls the SHIPPED function at every shading point (all bounces, although rt_primary_pass only uses it for the camera
 * ray's hit), counts the taps it would answer and compares each answer with the real trace: violations must be 0.
 *
 * build: gcc -O2 -std=c11 -o /tmp/lit_probe tests/lit_probe.c -lm -lpthread
 * usage: /tmp/lit_probe scene.txt W H spp bounces [pos.x pos.y pos.z yaw pitch fov]      exit status 1 on a violation */
#include <stdio.h>
#include <stdint.h>
static void lit_probe_tap(const void *hit, int light, int blocker, int bounce, const void *ray);
#define ORC_TAP_HOOK(hit, light, blocker, bounce, ray) lit_probe_tap(hit, light, blocker, bounce, ray)
#include "../oracle/rt_oracle.c"
static _Atomic uint64_t n_why[1024 + 8];
static int why_bounce;
#define RT_LIT_REFUSE(why) do { if (why_bounce == 0) n_why[(why) + 8]++; } while (0)
#include "../ray_tracing_amd/csrc/rt_lit.h"

static _Atomic uint64_t n_taps[16], n_known[16], n_lit[16], n_viol, n_table[16], n_table_viol;
static _Atomic uint64_t n_dark[16], n_dark_table[16], n_dark_viol;      /* "certainly NOT the emitter" (rt_region_certainly_dark) */
static unsigned int *dark_table;
static int only_light_emits;
static rt_lit_grid grids[1024];
static unsigned int *table;
static long long table_bits;
static float probe_cell = 0.125f;
static float packed[8 * 1024];

static void pack_scene(void)          /* as rt_set_scene packs rt_geom (rt_api.cpp) */
{
	const Scene *sc = &G.scene;
	for (int i = 0; i < sc->num_objects; i++) {
		const Object *o = &sc->objects[i];
		float *g = packed + 8 * i;
		if (o->type == OBJECT_CUBE) {
			g[0] = o->cube.origin.x; g[1] = o->cube.origin.y; g[2] = o->cube.origin.z;
			g[3] = o->cube.origin.x * 1.0f + o->cube.size.x * 1.0f;
			g[4] = o->cube.origin.y * 1.0f + o->cube.size.y * 1.0f;
			g[5] = o->cube.origin.z * 1.0f + o->cube.size.z * 1.0f;
			((int *) g)[6] = 0;
		} else {
			g[0] = o->sphere.center.x; g[1] = o->sphere.center.y; g[2] = o->sphere.center.z;
			g[3] = o->sphere.radius * o->sphere.radius; g[4] = g[5] = 0;
			((int *) g)[6] = 1;
		}
	}
}

/* ---- a look ahead (not in the product): taps that can only hit the EMITTER first, or nothing that matters ------------
 * With one emissive object in the scene a tap contributes nothing unless the emitter is its nearest hit.  If the cone of
 * all taps from P is clear of every other object -- in front of the emitter; an object wholly beyond the emitter's far
 * plane along an axis every tap direction moves along cannot be reached before the emitter -- then "is the emitter the
 * nearest hit" is the reference's test of the emitter ALONE: one box or sphere instead of the whole scene, whatever the
 * emitter's shape or size (scene_1's thin panel, which a third of the taps miss).  Counted here to size the idea. */
static _Atomic uint64_t n_only[16], n_only_viol;
static int emitter_only_would_do(const Hit *h, int light)
{
	const Scene *sc = &G.scene;
	int emitters = 0;
	for (int i = 0; i < sc->num_objects; i++) emitters += sc->objects[i].material.emission_power > 0;
	if (emitters != 1 || h->object == light) return 0;
	const float *ge = packed + 8 * light;
	float elo[3], ehi[3];
	rt_lit_object_box(ge, elo, ehi);
	const V3 c = centre_of(&sc->objects[light]);
	const float Rb = ((const int *) ge)[6] == 1 ? sqrtf(ge[3]) : 0.5f * sqrtf((ehi[0] - elo[0]) * (ehi[0] - elo[0]) + (ehi[1] - elo[1]) * (ehi[1] - elo[1]) + (ehi[2] - elo[2]) * (ehi[2] - elo[2]));
	const float p[3] = { h->point.x, h->point.y, h->point.z }, n[3] = { h->normal.x, h->normal.y, h->normal.z };
	if (!rt_lit_point_on_surface(packed + 8 * h->object, p[0], p[1], p[2], n[0], n[1], n[2])) return 0;
	const float l[3] = { c.x - p[0], c.y - p[1], c.z - p[2] };
	const float D = sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
	if (!(D >= Rb + 0.75f) || fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2])) > 32.0f) return 0;
	const float a[3] = { l[0] / D, l[1] / D, l[2] / D };
	const float s = 0.505f / (D - 0.5f);
	if (!(s <= 0.7f)) return 0;
	const float cs = sqrtf(1.0f - s * s), tau = 1.01f * s / cs, icos = 1.01f / cs, lean = 1.1f * s + 0.1f;
	if (!(a[0] * n[0] + a[1] * n[1] + a[2] * n[2] >= lean)) return 0;
	const float T = 1.01f * (D + Rb), m = RT_LIT_MARGIN;
	float lo[3], hi[3];
	for (int k = 0; k < 3; k++) {
		const float w = sqrtf(fmaxf(0.0f, 1.0f - a[k] * a[k])), r = T * tau * w + m, q = p[k] + T * a[k];
		lo[k] = fminf(a[k] >= lean ? p[k] + 2e-5f : p[k] - m, q - r);
		hi[k] = fmaxf(-a[k] >= lean ? p[k] - 2e-5f : p[k] + m, q + r);
	}
	for (int i = 0; i < sc->num_objects; i++) {
		if (i == light || i == h->object) continue;
		const float *g = packed + 8 * i;
		float blo[3], bhi[3], q[3], rho;
		rt_lit_object_box(g, blo, bhi);
		int behind = 0;                       /* wholly beyond the emitter along an axis all taps move along */
		for (int k = 0; k < 3; k++) behind |= (a[k] >= lean && blo[k] >= ehi[k]) || (-a[k] >= lean && bhi[k] <= elo[k]);
		if (behind) continue;
		if (((const int *) g)[6] == 1) {
			const float wx = g[0] - p[0], wy = g[1] - p[1], wz = g[2] - p[2], far = sqrtf(wx * wx + wy * wy + wz * wz);
			rho = sqrtf(g[3] + 2e-4f * far * far + 1e-4f);
			for (int k = 0; k < 3; k++) { q[k] = g[k]; blo[k] = g[k] - 1.001f * rho; bhi[k] = g[k] + 1.001f * rho; }
		} else {
			for (int k = 0; k < 3; k++) q[k] = 0.5f * (blo[k] + bhi[k]);
			rho = 0.5f * sqrtf((bhi[0] - blo[0]) * (bhi[0] - blo[0]) + (bhi[1] - blo[1]) * (bhi[1] - blo[1]) + (bhi[2] - blo[2]) * (bhi[2] - blo[2]));
		}
		if (lo[0] > bhi[0] || hi[0] < blo[0] || lo[1] > bhi[1] || hi[1] < blo[1] || lo[2] > bhi[2] || hi[2] < blo[2]) continue;
		const float v[3] = { q[0] - p[0], q[1] - p[1], q[2] - p[2] };
		const float along = v[0] * a[0] + v[1] * a[1] + v[2] * a[2], vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], rr = 1.001f * rho + m;
		if (along < -rr || along - rr > T) continue;
		const float lim = fmaxf(along, 0.0f) * tau + rr * icos;
		if (vv - along * along > lim * lim + 1e-4f * vv + 1e-4f) continue;
		return 0;
	}
	return 1;
}

static void lit_probe_tap(const void *hit, int light, int blocker, int bounce, const void *ray)
{
	const Hit *h = (const Hit *) hit;
	const V3 c = centre_of(&G.scene.objects[light]);
	n_taps[bounce]++;
	why_bounce = bounce;
	if (blocker == light) n_lit[bounce]++;
	const int cls = rt_taps_class(packed, G.scene.num_objects, light, c.x, c.y, c.z, only_light_emits, h->object,
	                              h->point.x, h->point.y, h->point.z, h->normal.x, h->normal.y, h->normal.z);
	if (cls == 1) {
		n_known[bounce]++;
		if (blocker != light && n_viol++ < 10)
			fprintf(stderr, "VIOLATION: point %.9g %.9g %.9g object %d: tap hit %d, not the emitter\n", h->point.x, h->point.y, h->point.z, h->object, blocker);
	} else if (cls == 2) {
		n_dark[bounce]++;
		if (blocker == light && n_dark_viol++ < 10)
			fprintf(stderr, "DARK VIOLATION: point %.9g %.9g %.9g object %d: the tap reaches the emitter\n", h->point.x, h->point.y, h->point.z, h->object);
	}
	if (emitter_only_would_do(h, light)) {
		const Ray *r = (const Ray *) ray;
		const Object *em = &G.scene.objects[light];
		const V3 d = unit(r->direction);
		float t = 0; V3 nn;
		const int hits = em->type == OBJECT_CUBE ? box_entry(r->origin, d, &em->cube, &t, &nn) : ball_entry(r->origin, d, &em->sphere, &t);
		n_only[bounce]++;
		if ((hits && t >= 0) != (blocker == light) && n_only_viol++ < 10)
			fprintf(stderr, "EMITTER-ONLY VIOLATION: point %.9g %.9g %.9g object %d: the emitter alone says %d, the scene says object %d\n", h->point.x, h->point.y, h->point.z, h->object, hits && t >= 0, blocker);
	}
	if (table_bits && h->object != light) {              /* the table of rt_lit_build, read as the trace kernel reads it */
		const int b = rt_lit_bit_of(&grids[h->object], h->point.x, h->point.y, h->point.z);
		const int on = rt_lit_point_on_surface(packed + 8 * h->object, h->point.x, h->point.y, h->point.z, h->normal.x, h->normal.y, h->normal.z);
		if (((table[b >> 5] >> (b & 31)) & 1u) && on) {
			n_table[bounce]++;
			if (blocker != light && n_table_viol++ < 10)
				fprintf(stderr, "TABLE VIOLATION: point %.9g %.9g %.9g object %d: tap hit %d, not the emitter\n", h->point.x, h->point.y, h->point.z, h->object, blocker);
		} else if (dark_table && ((dark_table[b >> 5] >> (b & 31)) & 1u) && on) {
			n_dark_table[bounce]++;
			if (blocker == light && n_dark_viol++ < 10)
				fprintf(stderr, "DARK TABLE VIOLATION: point %.9g %.9g %.9g object %d: the tap reaches the emitter\n", h->point.x, h->point.y, h->point.z, h->object);
		}
	}
}

int main(int argc, char **argv)
{
	static Scene sc;
	if (argc < 6 || orc_parse_scene_file(argv[1], &sc)) { fprintf(stderr, "usage: lit_probe scene W H spp bounces [camera: 6 floats]\n"); return 2; }
	orc_set_scene(&sc);
	pack_scene();
	static uint8_t texel[4] = { 128, 128, 128, 255 };
	uint8_t *faces[6] = { texel, texel, texel, texel, texel, texel };
	orc_set_skybox(faces, 1, 1, 3);
	rt_camera cam; orc_default_camera(&cam);
	if (argc >= 12) for (int k = 0; k < 6; k++) ((float *) &cam)[k] = (float) atof(argv[6 + k]);
	orc_set_camera(&cam);
	if (getenv("LIT_CELL")) probe_cell = (float) atof(getenv("LIT_CELL"));
	{
		int light = -1;
		for (int i = 0; i < sc.num_objects; i++) if (sc.objects[i].material.emission_power > 0) { light = i; break; }
		table_bits = rt_lit_layout(packed, sc.num_objects, light, probe_cell, grids);
		if (table_bits) {
			const V3 c = centre_of(&G.scene.objects[light]);
			table = malloc(sizeof(unsigned int) * (size_t) ((table_bits + 31) / 32));
			/* as rt_set_scene decides: "certainly dark" is an answer only when no other object's emission is non-zero */
			only_light_emits = 1;
			for (int i = 0; i < sc.num_objects; i++) {
				const Material *mt = &sc.objects[i].material;
				const float e[3] = { mt->emission_color.x * mt->emission_power, mt->emission_color.y * mt->emission_power, mt->emission_color.z * mt->emission_power };
				if (i != light && (e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f || e[0] != e[0] || e[1] != e[1] || e[2] != e[2])) only_light_emits = 0;
			}
			if (only_light_emits) dark_table = malloc(sizeof(unsigned int) * (size_t) ((table_bits + 31) / 32));
			rt_lit_build(packed, sc.num_objects, light, c.x, c.y, c.z, grids, table, dark_table, table_bits);
			long long set = 0, dset = 0;
			for (long long w = 0; w < (table_bits + 31) / 32; w++) { set += __builtin_popcount(table[w]); if (dark_table) dset += __builtin_popcount(dark_table[w]); }
			printf("table: %lld cells of %g, %lld lit, %lld dark\n", table_bits, probe_cell, set, dset);
		}
	}
	const int W = atoi(argv[2]), H = atoi(argv[3]), spp = atoi(argv[4]), nb = atoi(argv[5]);
	float *frame = malloc(sizeof(float) * 3 * W * H);
	orc_render_counter(W, H, spp, nb, 0, 0, H, 8, frame);
	uint64_t t = 0, k = 0, l = 0, kt = 0;
	for (int b = 0; b < 16; b++) if (n_taps[b]) {
		printf("bounce %d: taps %10llu  lit %5.1f %%  answered without tracing %5.1f %%  by the table %5.1f %%\n", b, (unsigned long long) n_taps[b],
		       100.0 * n_lit[b] / n_taps[b], 100.0 * n_known[b] / n_taps[b], 100.0 * n_table[b] / n_taps[b]);
		kt += n_table[b];
		t += n_taps[b]; k += n_known[b]; l += n_lit[b];
	}
	{
		uint64_t o = 0;
		for (int b = 0; b < 16; b++) o += n_only[b];
		if (t) printf("look ahead: %.1f %% of the taps (bounce 0: %.1f %%) would need the emitter tested alone; disagreements with the full trace: %llu\n",
		              100.0 * o / t, n_taps[0] ? 100.0 * n_only[0] / n_taps[0] : 0.0, (unsigned long long) n_only_viol);
	}
	if (t) printf("all: taps %llu  lit %.1f %%  answered %.1f %%  violations %llu  table %.1f %%  table violations %llu\n", (unsigned long long) t, 100.0 * l / t, 100.0 * k / t,
	              (unsigned long long) (n_viol + n_table_viol), 100.0 * kt / t, (unsigned long long) n_table_viol);
	{
		uint64_t d = 0, dt = 0;
		for (int b = 0; b < 16; b++) { d += n_dark[b]; dt += n_dark_table[b]; }
		if (t) printf("dark: %.1f %% of the taps certainly miss the emitter by the point classifier (bounce 0: %.1f %%), %.1f %% by the table; violations %llu\n",
		              100.0 * d / t, n_taps[0] ? 100.0 * n_dark[0] / n_taps[0] : 0.0, 100.0 * dt / t, (unsigned long long) n_dark_viol);
	}
	printf("bounce-0 taps refused: coordinates %llu, emitter too near %llu, cone wider than the emitter %llu, own surface %llu", (unsigned long long) n_why[7], (unsigned long long) n_why[6], (unsigned long long) n_why[5], (unsigned long long) n_why[4]);
	for (int i = 0; i < sc.num_objects; i++) if (n_why[8 + i]) printf(", object %d: %llu", i, (unsigned long long) n_why[8 + i]);
	printf("\n");
	return n_viol != 0 || n_table_viol != 0 || n_only_viol != 0 || n_dark_viol != 0;
}
