/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's hot path.
 *
 * This is the parity oracle for the HIP kernels and the "port" CPU baseline of bench.py.  It is
 * NOT part of the product and nothing under ray_tracing_amd/ may call it.
 *
 * It restates, operation by operation and rounding by rounding, what the reference computes
 * (citations are into /root/reference/src).  Arithmetic rules that matter for bit-equality with
 * the reference binary (x86-64 SSE2, -std=c11 => no FMA contraction, FLT_EVAL_METHOD 0):
 *   - every float product and sum is rounded separately (vector.c:148-155);
 *   - double-precision islands are kept in double (scene.c:117-118, main.c:128, main.c:219,
 *     main.c:241, vector.c:81, vector.c:132);
 *   - normalisation divides three times (vector.c:134-136), it does not multiply by a reciprocal.
 * Build with -ffp-contract=off semantics (gcc -std=c11 on x86-64 gives exactly that).
 *
 * Pinning: tests/test_oracle_vs_golden.py (fixtures captured from the compiled reference) and
 * tests/test_oracle_vs_ref.py (live, when oracle/_ref/ exists).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

/* ------------------------------------------------------------------------------------------ */
/* counters                                                                                    */
/* ------------------------------------------------------------------------------------------ */

#ifdef ORC_COUNTERS
static _Thread_local orc_counters tl_cnt;
static orc_counters g_cnt;
static pthread_mutex_t g_cnt_lock = PTHREAD_MUTEX_INITIALIZER;
#define COUNT(field, n) (tl_cnt.field += (n))
static void flush_counters(void)
{
	pthread_mutex_lock(&g_cnt_lock);
	g_cnt.samples += tl_cnt.samples;           g_cnt.rays += tl_cnt.rays;
	g_cnt.object_tests += tl_cnt.object_tests; g_cnt.rng_draws += tl_cnt.rng_draws;
	g_cnt.sky_fetches += tl_cnt.sky_fetches;   g_cnt.flops += tl_cnt.flops;
	g_cnt.box_tests += tl_cnt.box_tests;       g_cnt.box_flops += tl_cnt.box_flops;
	g_cnt.sphere_tests += tl_cnt.sphere_tests; g_cnt.sphere_flops += tl_cnt.sphere_flops;
	g_cnt.sky_samples += tl_cnt.sky_samples;   g_cnt.sky_sample_flops += tl_cnt.sky_sample_flops;
	g_cnt.first_ray_flops += tl_cnt.first_ray_flops;
	pthread_mutex_unlock(&g_cnt_lock);
	memset(&tl_cnt, 0, sizeof(tl_cnt));
}
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof(g_cnt)); memset(&tl_cnt, 0, sizeof(tl_cnt)); }
void orc_counters_get(orc_counters *out) { flush_counters(); *out = g_cnt; }
int  orc_has_counters(void) { return 1; }
#define FLOPS_NOW() (tl_cnt.flops)
#define COUNT_SPAN(tests, flops_field, since) (tl_cnt.tests += 1, tl_cnt.flops_field += tl_cnt.flops - (since))
#else
#define COUNT(field, n) ((void) 0)
#define FLOPS_NOW() ((uint64_t) 0)
#define COUNT_SPAN(tests, flops_field, since) ((void) (since))
static void flush_counters(void) {}
void orc_counters_reset(void) {}
void orc_counters_get(orc_counters *out) { memset(out, 0, sizeof(*out)); }
int  orc_has_counters(void) { return 0; }
#endif

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static struct {
	Scene     scene;
	Cubemap   sky;
	rt_camera cam;
	int       cam_set;
} G;

void orc_default_camera(rt_camera *cam)
{
	/* camera.c:28,33-35 */
	cam->pos   = (Vector3) {5, 5, 5};
	cam->front = (Vector3) {-1, -1, -1};
	cam->up    = (Vector3) {0, 1, 0};
	cam->fov   = 30.0f;
}

static const rt_camera *camera(void)
{
	if (!G.cam_set) { orc_default_camera(&G.cam); G.cam_set = 1; }
	return &G.cam;
}

void orc_set_scene(const Scene *scene) { memcpy(&G.scene, scene, sizeof(Scene)); }
const Scene *orc_scene(void) { return &G.scene; }
void orc_set_camera(const rt_camera *cam) { G.cam = *cam; G.cam_set = 1; }
void orc_set_skybox(uint8_t *const faces[6], int w, int h, int chan)
{
	for (int i = 0; i < 6; i++) G.sky.data[i] = faces[i];
	G.sky.w = w; G.sky.h = h; G.sky.chan = chan;
}

/* ------------------------------------------------------------------------------------------ */
/* RNG -- utils.c:60-75                                                                        */
/* ------------------------------------------------------------------------------------------ */

static uint64_t fold_mul(uint64_t a, uint64_t b)
{
	__uint128_t wide = (__uint128_t) a * b;
	return (uint64_t) (wide >> 64) ^ (uint64_t) wide;
}

static float draw(uint64_t *state)
{
	COUNT(rng_draws, 1); COUNT(flops, 1);
	*state += 0x60bee2bee120fc15ull;                                       /* utils.c:63 */
	uint64_t bits = fold_mul(fold_mul(*state, 0xa3b195354a39b70dull), 0x1b03738712fad5c9ull);
	return (float) bits / (float) UINT64_MAX;                              /* utils.c:74: divisor == 2^64 */
}

uint64_t orc_path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index)
{
	/* Not in the reference (it has no seed, utils.c:60): the `counter` mode's definition of the
	 * RNG state a path starts from.  splitmix64 finaliser over a Weyl sequence of the path id. */
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t) sample_index << 32) | (uint64_t) pixel_index);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

/* ------------------------------------------------------------------------------------------ */
/* vector algebra -- vector.c                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef Vector3 V3;

static V3 v3(float x, float y, float z) { return (V3) {x, y, z}; }

/* vector.c:148-155: u*a + v*b, two rounded products then one rounded sum per component */
static V3 lin2(V3 u, V3 v, float a, float b)
{
	COUNT(flops, 9);
	return v3(u.x * a + v.x * b, u.y * a + v.y * b, u.z * a + v.z * b);
}

/* vector.c:157-164, summed left to right */
static V3 lin4(V3 u, V3 v, V3 g, V3 t, float a, float b, float c, float d)
{
	COUNT(flops, 21);
	return v3(u.x * a + v.x * b + g.x * c + t.x * d,
	          u.y * a + v.y * b + g.y * c + t.y * d,
	          u.z * a + v.z * b + g.z * c + t.z * d);
}

static V3 scale(V3 v, float f) { COUNT(flops, 3); return v3(v.x * f, v.y * f, v.z * f); }   /* vector.c:140 */
static V3 hadamard(V3 a, V3 b) { COUNT(flops, 3); return v3(a.x * b.x, a.y * b.y, a.z * b.z); } /* :366 */
static float dot(V3 u, V3 v)   { COUNT(flops, 5); return u.x * v.x + u.y * v.y + u.z * v.z; }  /* :361 */

static V3 cross3(V3 u, V3 v)                                                                   /* :166 */
{
	COUNT(flops, 9);
	return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}

/* vector.c:119-138: norm via double sqrt of the float sum of squares; the epsilon test is a
 * double comparison against 0.00001; three divisions. */
static V3 unit(V3 v)
{
	COUNT(flops, 9);
	float len = (float) sqrt((double) (v.x * v.x + v.y * v.y + v.z * v.z));
	if ((double) len < 0.00001 && (double) len > -0.00001)
		return v;
	return v3(v.x / len, v.y / len, v.z / len);
}

static float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }  /* :52 */
static int tiny(float f) { return (double) f < 0.0001 && (double) f > -0.0001; }              /* :79 */

/* vector.c:99-111 -- components drawn in x, y, z order, each r*2-1, then normalised */
static V3 random_unit(uint64_t *state)
{
	COUNT(flops, 6);
	float x = draw(state) * 2 - 1;
	float y = draw(state) * 2 - 1;
	float z = draw(state) * 2 - 1;
	return unit(v3(x, y, z));
}

/* vector.c:113-117 */
static V3 mirror(V3 dir, V3 normal)
{
	COUNT(flops, 1);
	float f = -2 * dot(normal, dir);
	return lin2(dir, normal, 1, f);
}

/* ------------------------------------------------------------------------------------------ */
/* camera -- camera.c:95-125                                                                   */
/* ------------------------------------------------------------------------------------------ */

void orc_camera_basis(float aspect, rt_camera_basis *out)
{
	const rt_camera *c = camera();
	V3 w = unit(scale(c->front, -1));                               /* camera.c:99  */
	V3 u = unit(cross3(c->up, w));                                  /* camera.c:100 */
	V3 v = cross3(w, u);                                            /* camera.c:101 */
	float screen_h = (float) (2 * tan((double) (c->fov / 2)));      /* camera.c:107: float/2, tan in double */
	float screen_w = aspect * screen_h;                             /* camera.c:108 */
	out->pos        = c->pos;
	out->horizontal = scale(u, screen_w);                           /* camera.c:112 */
	out->vertical   = scale(v, screen_h);                           /* camera.c:113 */
	out->lower_left_corner = lin4(c->pos, out->horizontal, out->vertical, w, 1, -0.5f, -0.5f, -1); /* :118 */
}

static Ray primary_ray(float px, float py, float aspect)
{
	rt_camera_basis b;
	orc_camera_basis(aspect, &b);
	Ray r;
	r.origin    = b.pos;
	r.direction = lin4(b.lower_left_corner, b.horizontal, b.vertical, b.pos, 1, px, py, -1);   /* :121 */
	return r;
}

void orc_camera_ray(float px, float py, float aspect, float out[6])
{
	Ray r = primary_ray(px, py, aspect);
	out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
	out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
}

/* ------------------------------------------------------------------------------------------ */
/* intersection -- scene.c:10-190                                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float t; V3 point, normal; int object; } Hit;

/* scene.c:17-77.  Slab test.  Per axis the entry/exit parameters are ordered by the SIGN TEST
 * `d >= 0` (so -0.0 and +0.0 both take the first branch); comparisons are plain IEEE (NaN from
 * 0/0 compares false everywhere).  A hit is reported even when the entry parameter is negative. */
static int box_entry(V3 o, V3 d, const Cube *box, float *t_entry, V3 *normal)
{
	COUNT(flops, 12);
	V3 lo = box->origin;
	V3 hi = lin2(box->origin, box->size, 1, 1);                      /* scene.c:27 */

	float nx, fx, ny, fy, nz, fz;
	if (d.x >= 0) { nx = (lo.x - o.x) / d.x; fx = (hi.x - o.x) / d.x; }
	else          { fx = (lo.x - o.x) / d.x; nx = (hi.x - o.x) / d.x; }
	if (d.y >= 0) { ny = (lo.y - o.y) / d.y; fy = (hi.y - o.y) / d.y; }
	else          { fy = (lo.y - o.y) / d.y; ny = (hi.y - o.y) / d.y; }

	if (nx > fy || ny > fx) return 0;                                /* scene.c:47 */

	int axis = 0;
	float tn = nx, tf = fx;
	if (ny > tn) { tn = ny; axis = 1; }                              /* scene.c:50 (strict) */
	if (fy < tf) tf = fy;

	if (d.z >= 0) { nz = (lo.z - o.z) / d.z; fz = (hi.z - o.z) / d.z; }
	else          { fz = (lo.z - o.z) / d.z; nz = (hi.z - o.z) / d.z; }

	if (tn > fz || nz > tf) return 0;                                /* scene.c:61 */
	if (nz > tn) { tn = nz; axis = 2; }                              /* scene.c:64 */

	*t_entry = tn;
	float dc = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
	float s  = dc > 0 ? -1.0f : 1.0f;                                /* scene.c:71-73 */
	*normal = v3(axis == 0 ? s : 0.0f, axis == 1 ? s : 0.0f, axis == 2 ? s : 0.0f);
	return 1;
}

/* scene.c:79-134.  Discriminant in float without FMA; the two roots in double
 * ((-b +- sqrt(discr)) / (2a) with -b and 2*a formed in float first), rounded to float. */
static int ball_entry(V3 o, V3 d, const Sphere *ball, float *t_entry)
{
	COUNT(flops, 7);
	V3 oc = lin2(ball->center, o, 1, -1);
	float a = dot(d, d);
	float b = -2 * dot(oc, d);
	float c = dot(oc, oc) - ball->radius * ball->radius;
	float discr = b * b - 4 * a * c;
	if (!(discr > 0)) return 0;                                      /* tangent rays miss */
	COUNT(flops, 8);
	float r0 = (float) (((double) -b + sqrt((double) discr)) / (double) (2 * a));
	float r1 = (float) (((double) -b - sqrt((double) discr)) / (double) (2 * a));
	if (r0 > r1) { float tmp = r0; r0 = r1; r1 = tmp; }
	if (r0 < 0) { r0 = r1; if (r0 < 0) return 0; }
	*t_entry = r0;
	return 1;
}

/* scene.c:156-190 (+ intersect_object scene.c:136-154).  Linear scan, strict `<` so the lowest
 * index wins ties; the direction is normalised here, not by the caller. */
static Hit nearest_hit(Ray ray)
{
	COUNT(rays, 1);
	V3 o = ray.origin;
	V3 d = unit(ray.direction);
	Hit best = { FLT_MAX, {0, 0, 0}, {0, 0, 0}, -1 };
	for (int i = 0; i < G.scene.num_objects; i++) {
		const Object *obj = &G.scene.objects[i];
		COUNT(object_tests, 1);
		float t; V3 n;
		const uint64_t before = FLOPS_NOW();
		if (obj->type == OBJECT_CUBE) {
			const int hit = box_entry(o, d, &obj->cube, &t, &n);
			COUNT_SPAN(box_tests, box_flops, before);
			if (!hit) continue;
		} else if (obj->type == OBJECT_SPHERE) {
			const int hit = ball_entry(o, d, &obj->sphere, &t);
			if (hit) {
				V3 p = lin2(o, d, 1, t);                             /* scene.c:146 */
				n = unit(lin2(p, obj->sphere.center, 1, -1));        /* scene.c:147: outward, never flipped */
			}
			COUNT_SPAN(sphere_tests, sphere_flops, before);
			if (!hit) continue;
		} else
			continue;
		if (t >= 0 && t < best.t) { best.t = t; best.normal = n; best.object = i; }
	}
	if (best.object < 0)
		return (Hit) { -1, {0, 0, 0}, {0, 0, 0}, -1 };
	best.point = lin2(o, d, 1, best.t);                              /* scene.c:186 */
	return best;
}

int orc_trace_ray(const float o[3], const float d[3], float out[7])
{
	Hit h = nearest_hit((Ray) { {o[0], o[1], o[2]}, {d[0], d[1], d[2]} });
	out[0] = h.t;
	out[1] = h.point.x;  out[2] = h.point.y;  out[3] = h.point.z;
	out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z;
	return h.object;
}

/* scene.c:10-15 */
static V3 centre_of(const Object *obj)
{
	if (obj->type == OBJECT_SPHERE) return obj->sphere.center;
	return lin2(obj->cube.origin, obj->cube.size, 1, 0.5f);
}

/* ------------------------------------------------------------------------------------------ */
/* skybox -- gpu_and_windowing.c:42-112                                                        */
/* ------------------------------------------------------------------------------------------ */

static V3 sky_lookup(V3 dir)
{
	COUNT(sky_fetches, 1); COUNT(flops, 11);
	float ax = dir.x < 0 ? -dir.x : dir.x;
	float ay = dir.y < 0 ? -dir.y : dir.y;
	float az = dir.z < 0 ? -dir.z : dir.z;
	int face; float u, v;
	if (ax > ay && ax > az) {                    /* strict: ties fall through to Z */
		if (dir.x > 0) { face = CF_RIGHT; u = -dir.z / (ax + 0.0f); v = -dir.y / (ax + 0.0f); }
		else           { face = CF_LEFT;  u =  dir.z / (ax + 0.0f); v = -dir.y / (ax + 0.0f); }
	} else if (ay > ax && ay > az) {
		if (dir.y > 0) { face = CF_TOP;    u = dir.x / (ay + 0.0f); v =  dir.z / (ay + 0.0f); }
		else           { face = CF_BOTTOM; u = dir.x / (ay + 0.0f); v = -dir.z / (ay + 0.0f); }
	} else {
		if (dir.z > 0) { face = CF_FRONT; u =  dir.x / (az + 0.0f); v = -dir.y / (az + 0.0f); }
		else           { face = CF_BACK;  u = -dir.x / (az + 0.0f); v = -dir.y / (az + 0.0f); }
	}
	u = clampf(u, -1, 1);
	v = clampf(v, -1, 1);
	u = 0.5f * (u + 1.0f);
	v = 0.5f * (v + 1.0f);
	int x = (int) (u * (float) (G.sky.w - 1));                       /* truncation, nearest texel */
	int y = (int) (v * (float) (G.sky.h - 1));
	const uint8_t *texel = &G.sky.data[face][(y * G.sky.w + x) * G.sky.chan];
	return v3((float) texel[0] / 255, (float) texel[1] / 255, (float) texel[2] / 255);
}

void orc_sample_cubemap(const float d[3], float out[3])
{
	V3 c = sky_lookup(v3(d[0], d[1], d[2]));
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

#ifndef ORC_TAP_HOOK
#define ORC_TAP_HOOK(hit, light, blocker, bounce, ray) ((void) 0)
#endif

/* ------------------------------------------------------------------------------------------ */
/* one camera path -- main.c:126-272                                                           */
/* ------------------------------------------------------------------------------------------ */

static V3 shade_path(float px, float py, float aspect, int max_bounces, uint64_t *state)
{
	COUNT(samples, 1);
	const uint64_t sample_began = FLOPS_NOW();
	int left_at_once = 0;             /* (counters) the camera ray left the scene: a sky sample */
	uint64_t first_ray = 0;           /* (counters) flops up to the return of the first trace_ray */
	const Scene *sc = &G.scene;
	Ray ray = primary_ray(px, py, aspect);                           /* main.c:135 */

	int light = -1;                                                  /* main.c:140-146 */
	for (int i = 0; i < sc->num_objects; i++)
		if (sc->objects[i].material.emission_power > 0) { light = i; break; }

	V3 carry  = v3(1, 1, 1);      /* "contrib" */
	V3 radiance = v3(0, 0, 0);    /* "result"  */

	for (int bounce = 0; bounce < max_bounces; bounce++) {
		Hit hit = nearest_hit(ray);                                  /* main.c:161 */
		if (bounce == 0) first_ray = FLOPS_NOW() - sample_began;
		if (hit.object < 0) {
			left_at_once = bounce == 0;
			V3 sky = sky_lookup(unit(ray.direction));                /* main.c:170 */
			radiance = lin2(radiance, hadamard(sky, carry), 1, 1);   /* main.c:171 */
			break;
		}

		/* soft-shadow taps towards the first emitter -- main.c:180-210 */
		V3 lit = v3(0, 0, 0);
		if (light >= 0) {
			V3 to_light = lin2(centre_of(&sc->objects[light]), hit.point, 1, -1);
			int taps = 0;
			for (int k = 0; k < 3; k++) {
				V3 jitter = random_unit(state);                      /* always consumes 3 draws */
				if (dot(jitter, hit.normal) <= 0) continue;
				V3 sd = unit(lin2(jitter, to_light, 0.5f, 1));
				Ray shadow = { lin2(hit.point, sd, 1, 0.001f), sd };
				Hit blocker = nearest_hit(shadow);
				ORC_TAP_HOOK(&hit, light, blocker.object, bounce, &shadow);      /* development probes (tests/lit_probe.c); nothing by default */
				if (blocker.object >= 0) {
					const Material *bm = &sc->objects[blocker.object].material;
					lit = lin2(lit, bm->emission_color, 1, bm->emission_power);
				}
				taps++;                                              /* counted on a miss too */
			}
			if (taps > 0) { COUNT(flops, 1); lit = scale(lit, 1.0f / taps); }
		}

		const Material *m = &sc->objects[hit.object].material;       /* main.c:212 */
		V3 view = scale(ray.direction, -1);                          /* un-normalised on bounce 0 */
		V3 n = hit.normal;
		float n_dot_v = clampf(dot(n, view), 0, 1);

		/* Schlick -- main.c:219-222, 126-129; 0.16*r*r and pow() are double */
		COUNT(flops, 6);
		float f0_dielectric = (float) (0.16 * (double) m->reflectance * (double) m->reflectance);
		V3 f0 = lin2(v3(f0_dielectric, f0_dielectric, f0_dielectric), m->albedo, (1 - m->metallic), m->metallic);
		float grazing = (float) pow(1.0 - (double) n_dot_v, 5.0);
		V3 fresnel = lin2(f0, lin2(v3(1.0f, 1.0f, 1.0f), f0, 1, -1), 1, grazing);

		V3 scatter = random_unit(state);                             /* main.c:226-228 */
		if (dot(scatter, n) < 0) scatter = scale(scatter, -1);

		radiance = lin2(radiance, hadamard(scale(m->emission_color, m->emission_power), carry), 1, 1); /* :232 */

		V3 out_dir;
		COUNT(flops, 3);
		if ((double) m->metallic > 0.001                             /* main.c:241: the draw is short-circuited */
		    || draw(state) <= (fresnel.x + fresnel.y + fresnel.z) / 3) {
			V3 r = mirror(ray.direction, scale(n, -1));              /* un-normalised dir on bounce 0 */
			out_dir = unit(lin2(scatter, r, m->roughness, 1));
		} else {
			out_dir = scatter;
			COUNT(flops, 1);
			carry = hadamard(carry, scale(m->albedo, (1 - m->metallic)));
		}
		Ray next = { lin2(hit.point, out_dir, 1, 0.001f), out_dir }; /* main.c:250 */

		if (!(tiny(lit.x) && tiny(lit.y) && tiny(lit.z))) {          /* main.c:257-261 */
			COUNT(flops, 1);
			float w = 0.05f;
			radiance = lin2(radiance, hadamard(lit, carry), 1, w);
			carry = scale(carry, 1 - w);
		}
		ray = next;
	}
	if (left_at_once) COUNT_SPAN(sky_samples, sky_sample_flops, sample_began);
	else COUNT(first_ray_flops, first_ray);
	(void) sample_began; (void) left_at_once; (void) first_ray;
	return v3(clampf(radiance.x, 0, 1), clampf(radiance.y, 0, 1), clampf(radiance.z, 0, 1));
}

void orc_pixel(float u, float v, float aspect, int max_bounces, uint64_t *state, float out[3])
{
	V3 c = shade_path(u, v, aspect, max_bounces, state);
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

float orc_random_float(uint64_t *state) { return draw(state); }
void  orc_random_direction(uint64_t *state, float out[3])
{
	V3 d = random_unit(state);
	out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* ------------------------------------------------------------------------------------------ */
/* frame drivers                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* main.c:274-322 at a given scale for one column; returns the pass weight 1/scale^2.
 * `state` != NULL: the reference's sequential stream.  `state` == NULL: counter mode, the path of
 * low-resolution pixel (i, j) of pass `pass` starts from path_seed(seed, index of the tile's first
 * full-resolution pixel, pass). */
static float column_pass(V3 *data, int scale_, int column_w, int column_i, int W, int H,
                         int max_bounces, uint64_t *state, uint64_t seed, uint32_t pass)
{
	float weight = 1.0f / (scale_ * scale_);
	int column_x = column_w * column_i;
	float aspect = (float) W / H;
	int lw = W / scale_, lh = H / scale_;
	int lcw = column_w / scale_ + 1;       /* the "+1": one extra path per row, usually discarded */
	int lcx = column_x / scale_;
	for (int j = 0; j < lh; j++)
		for (int i = 0; i < lcw; i++) {
			COUNT(flops, 4);
			float u = (float) (lcx + i) / (lw - 1);
			float v = (float) j / (lh - 1);
			u = 1 - u;
			v = 1 - v;
			int tw = scale_, th = scale_;
			if (tw > column_w - i * scale_) tw = column_w - i * scale_;
			uint64_t local;
			uint64_t *st = state;
			if (!st) {
				local = orc_path_seed(seed, (uint32_t) ((j * scale_) * W + (lcx + i) * scale_), pass);
				st = &local;
			}
			V3 c = shade_path(u, v, aspect, max_bounces, st);
			for (int g = 0; g < th; g++)
				for (int t = 0; t < tw; t++)
					data[(j * scale_ + g) * column_w + (i * scale_ + t)] = c;
		}
	return weight;
}

/* Progressive accumulation in counter mode: worker()'s publish step (main.c:387-408) for one pass.
 * accum += column * (1/scale^2); *count += weight; returns the scale of the NEXT pass. */
int orc_progressive_pass(int W, int H, int scale_, int pass, int max_bounces, uint64_t seed,
                         float *accum, float *count)
{
	size_t n = (size_t) W * H;
	V3 *col = calloc(n, sizeof(V3));
	float w = column_pass(col, scale_, W, 0, W, H, max_bounces, NULL, seed, (uint32_t) pass);
	float k = 1.0f / (scale_ * scale_);
	V3 *acc = (V3*) accum;
	for (size_t q = 0; q < n; q++)
		acc[q] = lin2(acc[q], col[q], 1, k);                             /* main.c:394 */
	*count += w;                                                         /* main.c:396 */
	free(col);
	flush_counters();
	return scale_ > 1 ? scale_ >> 1 : scale_;                            /* main.c:402-403 */
}

/* update_frame()'s resolve, main.c:467-477 */
void orc_resolve(int W, int H, const float *accum, float count, float *frame_out)
{
	const V3 *acc = (const V3*) accum;
	for (size_t q = 0; q < (size_t) W * H; q++) {
		V3 f = scale(acc[q], 1.0f / count);
		frame_out[3*q] = f.x; frame_out[3*q+1] = f.y; frame_out[3*q+2] = f.z;
	}
}

void orc_render_stream(int W, int H, int passes, int init_scale, int max_bounces,
                       uint64_t *state, float *frame_out, float *accum_out)
{
	/* worker() with num_columns = 1: main.c:354-408; resolve main.c:467-477 */
	size_t n = (size_t) W * H;
	V3 *col = calloc(n, sizeof(V3));
	V3 *acc = calloc(n, sizeof(V3));
	float count = 0;
	int s = init_scale;
	for (int p = 0; p < passes; p++) {
		float w = column_pass(col, s, W, 0, W, H, max_bounces, state, 0, 0);
		float k = 1.0f / (s * s);
		for (size_t q = 0; q < n; q++)
			acc[q] = lin2(acc[q], col[q], 1, k);                     /* main.c:394 */
		count += w;                                                  /* main.c:396 */
		if (s > 1) s >>= 1;                                          /* main.c:402-403 */
	}
	if (accum_out) memcpy(accum_out, acc, n * sizeof(V3));
	if (frame_out)
		for (size_t q = 0; q < n; q++) {
			V3 f = scale(acc[q], 1.0f / count);                      /* main.c:476 */
			frame_out[3*q] = f.x; frame_out[3*q+1] = f.y; frame_out[3*q+2] = f.z;
		}
	free(col); free(acc);
	flush_counters();
}

typedef struct {
	int W, H, spp, max_bounces, row1;
	uint64_t seed;
	float *frame;
	atomic_int *next_row;
	const int *row_list;             /* NULL: rows next_row .. row1-1; else row_list[next_row .. row1-1] */
} CounterJob;

static void *counter_worker(void *arg)
{
	CounterJob *job = arg;
	int W = job->W, H = job->H;
	float aspect = (float) W / H;
	float inv = 1.0f / (float) job->spp;
	for (;;) {
		int j = atomic_fetch_add(job->next_row, 1);
		if (j >= job->row1) break;
		if (job->row_list) j = job->row_list[j];
		for (int i = 0; i < W; i++) {
			COUNT(flops, 4);
			float u = (float) i / (W - 1);                           /* main.c:293-296 at scale 1 */
			float v = (float) j / (H - 1);
			u = 1 - u;
			v = 1 - v;
			uint32_t p = (uint32_t) (j * W + i);
			V3 sum = v3(0, 0, 0);
			for (int s = 0; s < job->spp; s++) {
				uint64_t state = orc_path_seed(job->seed, p, (uint32_t) s);
				V3 c = shade_path(u, v, aspect, job->max_bounces, &state);
				sum = lin2(sum, c, 1, 1.0f);                         /* main.c:394, in sample order */
			}
			V3 f = scale(sum, inv);                                  /* main.c:476 */
			size_t q = (size_t) j * W + i;
			job->frame[3*q] = f.x; job->frame[3*q+1] = f.y; job->frame[3*q+2] = f.z;
		}
	}
	flush_counters();
	return NULL;
}

/* rows are dealt through job->next_row, so any number of workers finishes the job: threads that cannot be created (a box's
 * thread limit) are simply not joined, and the calling thread always works too */
static void run_counter_workers(CounterJob *job, int threads)
{
	pthread_t tid[256];
	int started = 0;
	for (int t = 1; t < threads && t < 256; t++)
		if (pthread_create(&tid[started], NULL, counter_worker, job) == 0) started++;
	counter_worker(job);
	for (int t = 0; t < started; t++) pthread_join(tid[t], NULL);
}

void orc_render_counter(int W, int H, int spp, int max_bounces, uint64_t seed,
                        int row0, int row1, int threads, float *frame_out)
{
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	atomic_int next_row = row0;
	CounterJob job = { W, H, spp, max_bounces, row1, seed, frame_out, &next_row, NULL };
	run_counter_workers(&job, threads);
}

/* the same for an arbitrary list of frame rows (every row in [0, H)), dealt to the threads one at a time: what the
 * full-size GPU parity tests use to check rows spread over a whole frame */
void orc_render_counter_rows(int W, int H, int spp, int max_bounces, uint64_t seed,
                             const int *rows, int num_rows, int threads, float *frame_out)
{
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	if (threads > num_rows) threads = num_rows > 0 ? num_rows : 1;
	atomic_int next_row = 0;
	CounterJob job = { W, H, spp, max_bounces, num_rows, seed, frame_out, &next_row, rows };
	run_counter_workers(&job, threads);
}

typedef struct { int column_i, column_w, W, H, passes, max_bounces; } ColumnJob;

static void *column_worker(void *arg)
{
	ColumnJob *job = arg;
	V3 *col = malloc(sizeof(V3) * (size_t) job->column_w * job->H);
	uint64_t state = 0;                                              /* every reference thread starts at 0 */
	for (int p = 0; p < job->passes; p++)
		column_pass(col, 1, job->column_w, job->column_i, job->W, job->H, job->max_bounces, &state, 0, 0);
	free(col);
	flush_counters();
	return NULL;
}

void orc_time_columns(int W, int H, int passes, int max_bounces, int threads)
{
	if (threads < 1) threads = 1;
	if (threads > 32) threads = 32;                                  /* MAX_COLUMNS, main.c:46 */
	pthread_t tid[32];
	ColumnJob jobs[32];
	int created[32];
	for (int t = 0; t < threads; t++) {
		jobs[t] = (ColumnJob) { t, W / threads, W, H, passes, max_bounces };
		created[t] = pthread_create(&tid[t], NULL, column_worker, &jobs[t]) == 0;
		if (!created[t]) column_worker(&jobs[t]);                /* (thread limit of the box: the column is rendered here) */
	}
	for (int t = 0; t < threads; t++) if (created[t]) pthread_join(tid[t], NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* scene text loader -- scene.c:206-624                                                        */
/* ------------------------------------------------------------------------------------------ */

static int blank(char c) { return c == ' ' || c == '\r' || c == '\t' || c == '\n'; }   /* utils.h:34 */
static int digit(char c) { return c >= '0' && c <= '9'; }                              /* utils.h:35 */

typedef struct { const char *s; size_t n, i; int line; } Cursor;

static void skip_blank(Cursor *c)
{
	while (c->i < c->n && blank(c->s[c->i])) { if (c->s[c->i] == '\n') c->line++; c->i++; }
}

/* keyword match with the reference's guard: `guard < len - i`, where guard is the keyword length
 * minus one (`5 < len - i` for "sphere", scene.c:224) -- except "albedo", guarded by 6 (scene.c:271) */
static int at_word(const Cursor *c, const char *word, size_t guard)
{
	if (c->i > c->n || !(guard < c->n - c->i)) return 0;
	return memcmp(c->s + c->i, word, strlen(word)) == 0;
}

/* scene.c:429-461: digits folded as v = v*10 + d in float, fraction as v += q*d with q /= 10 */
static int read_number(Cursor *c, float *out, int in_vector, int j)
{
	int sign = 1;
	char ch = c->i < c->n ? c->s[c->i] : '\0';
	if (ch == '-') {
		sign = -1;
		c->i++;
		if (c->i == c->n || !digit(c->s[c->i])) {
			fprintf(stderr, "Error: Missing number after minus sign (line %d)\n", c->line);
			return 0;
		}
	} else if (!digit(ch)) {
		if (in_vector) fprintf(stderr, "Error: Missing number %d in vector value (line %d)\n", j, c->line);
		else           fprintf(stderr, "Error: Missing number after property name (line %d)\n", c->line);
		return 0;
	}
	float val = 0;
	do { val = val * 10 + (c->s[c->i] - '0'); c->i++; } while (c->i < c->n && digit(c->s[c->i]));
	if (c->i < c->n && c->s[c->i] == '.') {
		c->i++;
		if (c->i == c->n || !digit(c->s[c->i])) {
			fprintf(stderr, "Error: Missing decimal part after dot (line %d)\n", c->line);
			return 0;
		}
		float q = 1.0f / 10;
		do { val += q * (c->s[c->i] - '0'); q /= 10; c->i++; } while (c->i < c->n && digit(c->s[c->i]));
	}
	*out = val * sign;
	return 1;
}

enum { P_ALBEDO, P_ROUGH, P_REFL, P_METAL, P_EPOW, P_ECOL, P_RADIUS, P_CENTER, P_ORIGIN, P_SIZE };

static const struct { const char *word; int guard; int advance; int is_vector; int only_for; const char *label; } PROPS[] = {
	/* `advance` reproduces the reference's cursor bumps, including albedo -> 9 (scene.c:280)
	 * and metallic -> 11 (scene.c:320) */
	{ "albedo",         6, 9,  1, -1,            NULL     },
	{ "roughness",      8, 9,  0, -1,            NULL     },
	{ "reflectance",    10, 11, 0, -1,            NULL     },
	{ "metallic",       7, 11, 0, -1,            NULL     },
	{ "emission_power", 13, 14, 0, -1,            NULL     },
	{ "emission_color", 13, 14, 1, -1,            NULL     },
	{ "radius",         5, 6,  0, OBJECT_SPHERE, "spheres" },
	{ "center",         5, 6,  1, OBJECT_SPHERE, "spheres" },
	{ "origin",         5, 6,  1, OBJECT_CUBE,   "cubes"   },
	{ "size",           3, 4,  1, OBJECT_CUBE,   "cubes"   },
};

static int unit_range(V3 v) { return !(v.x < 0 || v.x > 1 || v.y < 0 || v.y > 1 || v.z < 0 || v.z > 1); }

int orc_parse_scene_string(const char *src, size_t len, Scene *scene)
{
	Cursor c = { src, len, 0, 1 };
	scene->num_objects = 0;
	Object obj;
	memset(&obj, 0, sizeof(obj));
	for (;;) {
		skip_blank(&c);
		if (c.i == c.n) break;

		if (at_word(&c, "sphere", 5)) {
			obj.type = OBJECT_SPHERE;
			obj.sphere.center = v3(0, 0, 0);
			obj.sphere.radius = 1;
			c.i += 6;
		} else if (at_word(&c, "cube", 3)) {
			obj.type = OBJECT_CUBE;
			obj.cube.origin = v3(0, 0, 0);
			obj.cube.size   = v3(1, 1, 1);
			c.i += 4;
		} else {
			fprintf(stderr, "Error: Invalid character (line %d)\n", c.line);
			return -1;
		}
		obj.material.albedo         = v3(0.44f, 0.68f, 0.84f);       /* scene.c:234 (double literals -> float) */
		obj.material.roughness      = 0;
		obj.material.reflectance    = 0.2f;
		obj.material.metallic       = 0;
		obj.material.emission_power = 0;
		obj.material.emission_color = v3(1, 1, 1);

		for (;;) {
			skip_blank(&c);
			int p = -1;
			for (int k = 0; k < (int) (sizeof(PROPS) / sizeof(PROPS[0])); k++)
				if (at_word(&c, PROPS[k].word, (size_t) PROPS[k].guard)) { p = k; break; }
			if (p < 0) break;
			if (PROPS[p].only_for >= 0 && (int) obj.type != PROPS[p].only_for) {
				fprintf(stderr, "Poperty '%s' only allowed on %s (line %d)\n", PROPS[p].word, PROPS[p].label, c.line);
				return -1;
			}
			c.i += PROPS[p].advance;

			skip_blank(&c);
			if (c.i >= c.n) {
				fprintf(stderr, "Error: Property value is missing (line %d)\n", c.line);
				return -1;
			}

			float f = 0; V3 vec = {0, 0, 0};
			if (!PROPS[p].is_vector) {
				if (!read_number(&c, &f, 0, 0)) return -1;
			} else {
				if (c.s[c.i] != '{') {
					fprintf(stderr, "Error: Missing '{' after property name (line %d)\n", c.line);
					return -1;
				}
				c.i++;
				float tmp[3];
				for (int j = 0; j < 3; j++) {
					skip_blank(&c);
					if (!read_number(&c, &tmp[j], 1, j)) return -1;
				}
				skip_blank(&c);
				if (c.i >= c.n || c.s[c.i] != '}') {
					fprintf(stderr, "Error: Missing '}' after property value (line %d)\n", c.line);
					return -1;
				}
				c.i++;
				vec = v3(tmp[0], tmp[1], tmp[2]);
			}

			switch (p) {
			case P_ALBEDO:
				if (!unit_range(vec)) { fprintf(stderr, "Error: albedo values must be between 0 and 1 (line %d)\n", c.line); return -1; }
				obj.material.albedo = vec; break;
			case P_ROUGH:
				if (f < 0 || f > 1) { fprintf(stderr, "Error: Roughness must be between 0 and 1 (line %d)\n", c.line); return -1; }
				obj.material.roughness = f; break;
			case P_REFL:
				if (f < 0 || f > 1) { fprintf(stderr, "Error: Reflectance must be between 0 and 1 (line %d)\n", c.line); return -1; }
				obj.material.reflectance = f; break;
			case P_METAL:
				if (f < 0 || f > 1) { fprintf(stderr, "Error: Metallic must be between 0 and 1 (line %d)\n", c.line); return -1; }
				obj.material.metallic = f; break;
			case P_EPOW:   obj.material.emission_power = f; break;
			case P_ECOL:
				if (!unit_range(vec)) { fprintf(stderr, "Error: Emission color values must be between 0 and 1 (line %d)\n", c.line); return -1; }
				obj.material.emission_color = vec; break;
			case P_RADIUS: obj.sphere.radius = f; break;
			case P_CENTER: obj.sphere.center = vec; break;
			case P_ORIGIN: obj.cube.origin = vec; break;
			case P_SIZE:
				if (vec.x < 0 || vec.y < 0 || vec.z < 0) { fprintf(stderr, "Error: Size values must be positive (line %d)\n", c.line); return -1; }
				obj.cube.size = vec; break;
			}
		}

		if (scene->num_objects == MAX_OBJECTS)
			fprintf(stderr, "Warning: Ignoring object because the scene is too big (line %d)\n", c.line);
		else
			scene->objects[scene->num_objects++] = obj;
	}
	return 0;
}

int orc_parse_scene_file(const char *path, Scene *scene)
{
	FILE *f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "Error: Couldn't open scene file\n"); return -1; }
	fseek(f, 0, SEEK_END);
	long size = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = malloc((size_t) size + 1);
	if (!buf) { fclose(f); return -1; }
	size_t got = fread(buf, 1, (size_t) size, f);
	buf[got] = '\0';
	fclose(f);
	int rc = orc_parse_scene_string(buf, (size_t) size, scene);
	free(buf);
	return rc;
}
