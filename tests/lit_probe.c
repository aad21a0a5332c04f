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

/* ---- a look ahead (not in the product; the kernel side was built and measured in round 3 -- scripts/patches/emitter_alone_two_bags.diff,
 * DESIGN.md section 9 -- and did not pay) ---------------------------------------------------------------------------------------- */
/* The third certainty, again for scenes in which the emitter alone emits, and for an emitter of either shape (scene_1's
 * is a thin panel that a third of the taps miss): no object BUT the emitter can be an accepted tap's nearest hit in front
 * of the emitter.  Then the tap adds the emitter's emission exactly when the reference's intersection test of the emitter
 * ALONE reports a hit with t >= 0 (scene.c:17-150 on one object instead of trace_ray() on all of them): whatever else the
 * tap may hit when it misses the emitter adds nothing.  The tap still has to be traced -- against one object.
 *   - every accepted tap leaves P's own object (the condition of rt_region_certainly_lit);
 *   - every other object is clear of the cone of all taps, cut off behind the emitter's bounding sphere, by RT_LIT_MARGIN
 *     (the same test), or lies wholly beyond the emitter -- by the margin -- along an axis that every tap direction moves
 *     along by >= 0.1: a tap that reaches such an object has passed the emitter's far plane before, so if it hits the
 *     emitter it hits it first;
 *   - EXCEPT the emitter's companions (`with`, a bit per object, the emitter's own among them): objects that touch the emitter
 *     or come within twice the margin of it (scene_1's panel hangs from the ceiling: their planes coincide, and which of
 *     the two a grazing tap reaches first is a matter of the last bit).  No clearance is asked of them; the tap is traced
 *     against the emitter AND its companions, each with the reference's own test, in index order: if the emitter is the
 *     nearest of those it is the nearest of all. */
RT_LIT_FN unsigned long long rt_lit_companions(const float *geom, int num_objects, int light);
RT_LIT_FN int rt_region_emitter_alone(const float *geom, int num_objects, int light, float cx, float cy, float cz, int hobj,
                                      float px, float py, float pz, float hx, float hy, float hz, float nx, float ny, float nz, float nslack,
                                      unsigned long long with)
{
	if (light < 0 || hobj == light || hobj < 0 || hobj >= num_objects || num_objects > 64 || !((with >> light) & 1ull)) return 0;
	const float *ge = geom + 8 * light;
	float elo[3], ehi[3];
	if (((const int *) ge)[6] == 0) { elo[0] = ge[0]; elo[1] = ge[1]; elo[2] = ge[2]; ehi[0] = ge[3]; ehi[1] = ge[4]; ehi[2] = ge[5]; }
	else if (((const int *) ge)[6] == 1) { const float rho = 1.001f * RT_LIT_SQRT(ge[3]); for (int k = 0; k < 3; k++) { elo[k] = ge[k] - rho; ehi[k] = ge[k] + rho; } }
	else return 0;
	const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px) + hx, __builtin_fabsf(py) + hy), __builtin_fabsf(pz) + hz),
	                                  __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)), __builtin_fabsf(cz)));
	if (!(big <= 32.0f)) return 0;
	/* no point of the emitter is farther from (cx, cy, cz) than Rb (for a cube that is its centre, scene.c:10-15) */
	float Rb = 0.0f;
	for (int k = 0; k < 3; k++) { const float c = k == 0 ? cx : (k == 1 ? cy : cz); const float e = __builtin_fmaxf(ehi[k] - c, c - elo[k]); Rb += e * e; }
	Rb = 1.001f * RT_LIT_SQRT(Rb);
	if (!(Rb <= 32.0f)) return 0;
	const float sl = 1.001f * RT_LIT_SQRT(hx * hx + hy * hy + hz * hz);
	const float lx = cx - px, ly = cy - py, lz = cz - pz;
	const float D = RT_LIT_SQRT(lx * lx + ly * ly + lz * lz);
	const float Dmin = D - sl, Dmax = D + sl;
	if (!(Dmin >= 0.75f)) return 0;                      /* (the emitter's own test is the reference's: the point may be anywhere outside 0.75 of its centre) */
	const float inv = 1.0f / D;
	const float ax = lx * inv, ay = ly * inv, az = lz * inv;
	const float s = 0.505f / (Dmin - 0.5f) + 1.05f * sl / Dmin;
	if (!(s <= 0.7f)) return 0;
	const float cs = RT_LIT_SQRT(1.0f - s * s);
	const float tau = 1.01f * s / cs, icos = 1.01f / cs;
	const float lean = 1.1f * s + 0.1f;
	const float leave = 0.102f * (1.0f + 0.5f / Dmin) + 1.05f * sl / Dmin + 1e-4f;
	if (!(ax * nx + ay * ny + az * nz - nslack >= leave)) return 0;
	const float T = 1.01f * (Dmax + Rb);
	const float m = RT_LIT_MARGIN;
	const float p[3] = { px, py, pz }, a[3] = { ax, ay, az }, h[3] = { hx, hy, hz };
	float lo[3], hi[3];
	for (int k = 0; k < 3; k++) {
		const float w = RT_LIT_SQRT(__builtin_fmaxf(0.0f, 1.0f - a[k] * a[k]));
		const float r = T * tau * w + m + h[k];
		const float q = p[k] + T * a[k];
		lo[k] = __builtin_fminf(a[k] >= lean ? p[k] - h[k] + 2e-5f : p[k] - h[k] - m, q - r);
		hi[k] = __builtin_fmaxf(-a[k] >= lean ? p[k] + h[k] - 2e-5f : p[k] + h[k] + m, q + r);
	}
	for (int i = 0; i < num_objects; i++) {
		if (i == hobj || ((with >> i) & 1ull)) continue;
		const float *g = geom + 8 * i;
		const int type = ((const int *) g)[6];
		float blo[3], bhi[3], q[3], rho;
		if (type == 0) {
			blo[0] = g[0]; blo[1] = g[1]; blo[2] = g[2]; bhi[0] = g[3]; bhi[1] = g[4]; bhi[2] = g[5];
			const float ex = bhi[0] - blo[0], ey = bhi[1] - blo[1], ez = bhi[2] - blo[2];
			q[0] = blo[0] + 0.5f * ex; q[1] = blo[1] + 0.5f * ey; q[2] = blo[2] + 0.5f * ez;
			rho = 0.5f * RT_LIT_SQRT(ex * ex + ey * ey + ez * ez);
		} else if (type == 1) {
			const float wx = g[0] - px, wy = g[1] - py, wz = g[2] - pz;
			const float far = RT_LIT_SQRT(wx * wx + wy * wy + wz * wz) + sl;
			rho = RT_LIT_SQRT(g[3] + 2e-4f * far * far + 1e-4f);
			for (int k = 0; k < 3; k++) { q[k] = g[k]; blo[k] = g[k] - 1.001f * rho; bhi[k] = g[k] + 1.001f * rho; }
		} else
			continue;
		int behind = 0;                                  /* wholly beyond the emitter along an axis all taps move along */
		for (int k = 0; k < 3; k++) behind |= (a[k] >= lean && blo[k] >= ehi[k] + m) || (-a[k] >= lean && bhi[k] <= elo[k] - m);
		if (behind) continue;
		if (lo[0] > bhi[0] || hi[0] < blo[0] || lo[1] > bhi[1] || hi[1] < blo[1] || lo[2] > bhi[2] || hi[2] < blo[2])
			continue;
		const float vx = q[0] - px, vy = q[1] - py, vz = q[2] - pz;
		const float along = vx * ax + vy * ay + vz * az;
		const float vv = vx * vx + vy * vy + vz * vz;
		const float rr = 1.001f * rho + m + sl;
		if (along < -rr || along - rr > T) continue;
		const float lim = __builtin_fmaxf(along, 0.0f) * tau + rr * icos;
		if (vv - along * along > lim * lim + 1e-4f * vv + 1e-4f) continue;
		return 0;
	}
	return 1;
}

/* the emitter and its companions (see above), a bit per object; 0: the scene has more than 64 objects, the emitter more than
 * three companions (then the shortcut is not worth having), or there is no emitter */
RT_LIT_FN unsigned long long rt_lit_companions(const float *geom, int num_objects, int light)
{
	if (light < 0 || light >= num_objects || num_objects > 64) return 0ull;
	float elo[3], ehi[3];
	const float *ge = geom + 8 * light;
	if (((const int *) ge)[6] == 0) { elo[0] = ge[0]; elo[1] = ge[1]; elo[2] = ge[2]; ehi[0] = ge[3]; ehi[1] = ge[4]; ehi[2] = ge[5]; }
	else if (((const int *) ge)[6] == 1) { const float rho = 1.001f * RT_LIT_SQRT(ge[3]); for (int k = 0; k < 3; k++) { elo[k] = ge[k] - rho; ehi[k] = ge[k] + rho; } }
	else return 0ull;
	unsigned long long with = 1ull << light;
	int count = 0;
	for (int i = 0; i < num_objects; i++) {
		if (i == light) continue;
		const float *g = geom + 8 * i;
		float blo[3], bhi[3];
		if (((const int *) g)[6] == 0) { blo[0] = g[0]; blo[1] = g[1]; blo[2] = g[2]; bhi[0] = g[3]; bhi[1] = g[4]; bhi[2] = g[5]; }
		else if (((const int *) g)[6] == 1) { const float rho = 1.001f * RT_LIT_SQRT(g[3] + 1e-4f); for (int k = 0; k < 3; k++) { blo[k] = g[k] - rho; bhi[k] = g[k] + rho; } }
		else continue;
		const float m2 = 2.0f * RT_LIT_MARGIN;
		/* (written so that a NaN anywhere makes the object a companion) */
		if (!(blo[0] > ehi[0] + m2 || bhi[0] < elo[0] - m2 || blo[1] > ehi[1] + m2 || bhi[1] < elo[1] - m2 || blo[2] > ehi[2] + m2 || bhi[2] < elo[2] - m2)) {
			with |= 1ull << i;
			if (++count > 3) return 0ull;
		}
	}
	return with;
}

static _Atomic uint64_t n_only[16], n_only_viol;
static unsigned long long with_set;     /* rt_lit_companions(): the emitter and the objects that touch it */
static int emitter_only_would_do(const Hit *h, int light)
{
	if (!only_light_emits || with_set == 0ull || h->object == light) return 0;
	const V3 c = centre_of(&G.scene.objects[light]);
	if (!rt_lit_point_on_surface(packed + 8 * h->object, h->point.x, h->point.y, h->point.z, h->normal.x, h->normal.y, h->normal.z)) return 0;
	return rt_region_emitter_alone(packed, G.scene.num_objects, light, c.x, c.y, c.z, h->object, h->point.x, h->point.y, h->point.z, 0.0f, 0.0f, 0.0f,
	                               h->normal.x, h->normal.y, h->normal.z, 0.0f, with_set);
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
	if (cls == 0 && emitter_only_would_do(h, light)) {   /* the reference's tests of the emitter and its companions alone, in index order, on this tap's ray */
		const Ray *r = (const Ray *) ray;
		const V3 d = unit(r->direction);
		float best = 3.402823466e+38f; int who = -1;
		for (int i = 0; i < G.scene.num_objects && i < 64; i++) {
			if (!((with_set >> i) & 1ull)) continue;
			const Object *em = &G.scene.objects[i];
			float t = 0; V3 nn;
			const int hits = em->type == OBJECT_CUBE ? box_entry(r->origin, d, &em->cube, &t, &nn) : ball_entry(r->origin, d, &em->sphere, &t);
			if (hits && t >= 0 && t < best) { best = t; who = i; }
		}
		n_only[bounce]++;
		if ((who == light) != (blocker == light) && n_only_viol++ < 10)
			fprintf(stderr, "EMITTER-ALONE VIOLATION: point %.9g %.9g %.9g object %d: the emitter and its companions alone say %d, the scene says object %d\n", h->point.x, h->point.y, h->point.z, h->object, who, blocker);
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
		/* as rt_set_scene decides: "certainly dark" is an answer only when no other object's emission is non-zero */
		only_light_emits = light >= 0;
		for (int i = 0; i < sc.num_objects; i++) {
			const Material *mt = &sc.objects[i].material;
			const float e[3] = { mt->emission_color.x * mt->emission_power, mt->emission_color.y * mt->emission_power, mt->emission_color.z * mt->emission_power };
			if (i != light && (e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f || e[0] != e[0] || e[1] != e[1] || e[2] != e[2])) only_light_emits = 0;
		}
		with_set = only_light_emits ? rt_lit_companions(packed, sc.num_objects, light) : 0ull;
		table_bits = rt_lit_layout(packed, sc.num_objects, light, probe_cell, grids);
		if (table_bits) {
			const V3 c = centre_of(&G.scene.objects[light]);
			table = malloc(sizeof(unsigned int) * (size_t) ((table_bits + 31) / 32));
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
		if (t) printf("alone: %.1f %% of the taps (bounce 0: %.1f %%), none of them answered by the shipped classes, would need only the emitter and what touches it (set 0x%llx) intersected; violations %llu\n",
		              100.0 * o / t, n_taps[0] ? 100.0 * n_only[0] / n_taps[0] : 0.0, with_set, (unsigned long long) n_only_viol);
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
