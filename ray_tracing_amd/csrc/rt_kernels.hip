/*
 * rt_kernels.hip -- hand-written HIP kernels for the path-tracing hot path (gfx950 / CDNA4).
 *
 * Replaces, on the GPU, what the reference's worker threads do per pass:
 *   render_column() -> pixel() -> trace_ray()/sample_cubemap() -> accumulate -> resolve
 *   (main.c:274-322, 131-272, 387-396, 467-477; scene.c:10-190; gpu_and_windowing.c:42-112).
 *
 * Kernels:
 *   rt_trace_simple      the path loop in the reference's own order, one lane per pixel (cross-check)
 *   rt_primary_pass      camera rays of all pixels, once per launch: sky-only pixels are finished, the rest is handed on
 *   rt_trace_wavefront   the tuned schedule: persistent waves that deal samples to lanes, per-wave LDS ray queue,
 *                        in-order sample sum through a per-wave LDS window, exact shortcuts
 *   rt_trace_spec        the same, recompiled by hiprtc with the scene as constants (rt_compile_scene)
 *   rt_accumulate / rt_resolve (progressive passes); rt_deinterleave (multi-GPU root); rt_selftest_kernel
 * One lane owns one path at a time and the samples of a pixel are always added in sample order
 * (main.c:394), whichever kernel runs.  Scene geometry and shading records are staged once per
 * workgroup into LDS and read with wave-uniform (broadcast) ds_read_b128; the skybox stays in HBM /
 * Infinity Cache as RGBA8 (one dword per fetch).  No MFMA: there is no contraction in this workload.
 */
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#endif
#include "rt_device.h"
#include "rt_math.hip.h"
#define RT_LIT_FN __host__ __device__ static inline
#include "rt_lit.h"

#pragma clang fp contract(off)

#define RT_BLOCK    256
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4     /* register budget of the wavefront kernels: 512 / 4 = 128 VGPRs */
#endif
#define RT_TILE_W   32     /* workgroup tile: 32 x 8 pixels = four 8x8 wave tiles side by side */
#define RT_TILE_H   8

struct Hit { float t; V3 n; int obj; };
#define RT_PIX_TAPS_LIT  0x10000     /* flags beside the object index of a pixel record (objects: < 1024): every accepted tap from */
#define RT_PIX_TAPS_DARK 0x20000     /* the hit point certainly reaches the emitter first / certainly does not (rt_lit.h) */
#define RT_PIX_TAPS_AUDIT 0x40000    /* ... and the launch audits this pixel's answer: its taps are traced all the same and compared (rt_launch.audit_taps) */

/* Development instrumentation hooks (rt_stats.hip.h: `make stats`, rt_tuning.jit_flags "-DRT_STATS"): nothing in the product build */
#ifdef RT_STATS
#include "rt_stats.hip.h"
#else
#define STAT(site) do {} while (0)
#define STAMP_LOCAL do {} while (0)
#define STAMP(k) do {} while (0)
#define STAMP_DRY do {} while (0)
#define STAMP_ROUND do {} while (0)
#define STAMP_FLUSH(waves_per_block) do {} while (0)
#endif

/* LDS written by some lanes of a wave is read by others of the same wave */
RT_DEV int lanes_below(unsigned long long m)      /* number of set bits of m below this lane */
{
	return (int) __builtin_amdgcn_mbcnt_hi((unsigned int) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m, 0u));
}
RT_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

/* ---- LDS-resident scene ------------------------------------------------------------------- */

struct SceneLDS {
	const float4 *geom;    /* 2 x float4 per object */
	const float4 *shade;   /* 4 x float4 per object */
};

/* Large scenes (the culled kernels): neither geometry nor shading records go to LDS -- the lanes test a cluster's members against
 * the cluster record's quantised boxes (stage_clusters), the few exact tests that follow read the 32-byte geometry records from
 * memory (L2), as the shading records, touched once per bounce, already were: a scene of 1024 objects takes 14 KB of LDS per
 * workgroup instead of 34 (round 4: 32 KB of geometry, read 8 x 32 bytes per ray and cluster at four lanes to a bank). */
RT_DEV SceneLDS stage_geometry(const rt_launch &L)
{
	SceneLDS sc; sc.geom = reinterpret_cast<const float4*>(L.geom); sc.shade = reinterpret_cast<const float4*>(L.shade);
	return sc;
}

RT_DEV SceneLDS stage_scene(const rt_launch &L, float4 *lds, int n)
{
	const float4 *g = reinterpret_cast<const float4*>(L.geom);
	const float4 *s = reinterpret_cast<const float4*>(L.shade);
	for (int i = threadIdx.x; i < 2 * n; i += RT_BLOCK) lds[i] = g[i];
	for (int i = threadIdx.x; i < 4 * n; i += RT_BLOCK) lds[2 * n + i] = s[i];
	__syncthreads();
	SceneLDS sc; sc.geom = lds; sc.shade = lds + 2 * n;
	return sc;
}

/* ---- intersection: scene.c:17-190 ----------------------------------------------------------- */

/* scene.c:17-77.  `lo`,`hi` are wave-uniform; o,d per lane.  Entry/exit ordered by the sign test
 * d >= 0 (so +-0 take the first branch), plain IEEE compares, hit reported even for t < 0. */
RT_DEV bool box_entry(V3 o, V3 d, V3 lo, V3 hi, float &t_entry, V3 &normal)
{
	float ax = (lo.x - o.x) / d.x, bx = (hi.x - o.x) / d.x;
	float ay = (lo.y - o.y) / d.y, by = (hi.y - o.y) / d.y;
	float nx = d.x >= 0 ? ax : bx, fx = d.x >= 0 ? bx : ax;
	float ny = d.y >= 0 ? ay : by, fy = d.y >= 0 ? by : ay;
	if (nx > fy || ny > fx) return false;

	int axis = 0;
	float tn = nx, tf = fx;
	if (ny > tn) { tn = ny; axis = 1; }
	if (fy < tf) tf = fy;

	float az = (lo.z - o.z) / d.z, bz = (hi.z - o.z) / d.z;
	float nz = d.z >= 0 ? az : bz, fz = d.z >= 0 ? bz : az;
	if (tn > fz || nz > tf) return false;
	if (nz > tn) { tn = nz; axis = 2; }

	t_entry = tn;
	float dc = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
	float s  = dc > 0 ? -1.0f : 1.0f;
	normal = mk3(axis == 0 ? s : 0.0f, axis == 1 ? s : 0.0f, axis == 2 ? s : 0.0f);
	return true;
}

/* scene.c:79-134.  `dd` = dot(d,d) is the same for every sphere of one ray and is hoisted by the
 * caller.  Discriminant in float (no FMA); roots in fp64 from float -b and float 2*a. */
RT_DEV bool ball_entry(V3 o, V3 d, float dd, V3 center, float r2, float &t_entry)
{
	V3 oc = sub3(center, o);
	float b = -2.0f * dot3(oc, d);
	float c = dot3(oc, oc) - r2;
	float discr = b * b - 4.0f * dd * c;
	if (!(discr > 0)) return false;
	double root = __builtin_sqrt((double) discr);
	double den  = (double) (2.0f * dd);
	float r0 = (float) (((double) -b + root) / den);
	float r1 = (float) (((double) -b - root) / den);
	if (r0 > r1) { float tmp = r0; r0 = r1; r1 = tmp; }
	if (r0 < 0) { r0 = r1; if (r0 < 0) return false; }
	t_entry = r0;
	return true;
}

/* scene.c:156-190.  `d` must already be normalised (trace_ray normalises a local copy, :158).
 * Linear scan in object order, strict `<`: lowest index wins ties. */
RT_DEV Hit nearest_hit(const SceneLDS &sc, int n, V3 o, V3 d)
{
	Hit best; best.t = 3.402823466e+38f; best.obj = -1; best.n = mk3(0, 0, 0);
	const float dd = dot3(d, d);
	for (int i = 0; i < n; i++) {
		const float4 g0 = sc.geom[2 * i], g1 = sc.geom[2 * i + 1];
		const int type = __float_as_int(g1.z);
		float t; V3 nn;
		if (type == RT_GEOM_CUBE) {
			if (!box_entry(o, d, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), t, nn)) continue;
			if (t >= 0 && t < best.t) { best.t = t; best.n = nn; best.obj = i; }
		} else if (type == RT_GEOM_SPHERE) {
			V3 c = mk3(g0.x, g0.y, g0.z);
			if (!ball_entry(o, d, dd, c, g0.w, t)) continue;
			if (t >= 0 && t < best.t) {
				best.t = t; best.obj = i;
				best.n = unit3(sub3(madd3(o, d, t), c));    /* scene.c:146-147 */
			}
		}
	}
	return best;
}

/* ---- tuned intersection: same results as box_entry/ball_entry/nearest_hit, fewer instructions ---
 *  - the six slab quotients of a box share the ray's three refined reciprocals (rt_math.hip.h);
 *  - the slab test is branch-free (a 64-ray batch is incoherent: an early-out would only idle lanes);
 *  - the sphere roots share one refined fp64 reciprocal of 2a per ray, and the larger root is only
 *    formed when the smaller one is negative (2a > 0, so (-b - sqrt(D))/2a <= (-b + sqrt(D))/2a after
 *    every rounding: the reference's sort, scene.c:119-123, always ends with that order);
 *  - the hit normal is formed once, for the winning object, instead of for every candidate.
 * Every quotient is still the correctly rounded IEEE one; operands outside the window in which the
 * shared-reciprocal form is exact (including zero numerators) fall back to `/` for the whole wave. */

struct RayPrep {
	V3     inv;        /* refined reciprocals of the direction components */
	bool   inv_ok;
	float  dd;         /* dot(d, d): `a` of every sphere test                      */
	double den, rden;  /* (double)(2*a) and its refined reciprocal                 */
	bool   den_ok;
};

RT_DEV uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }

template <bool ORIGIN_CHECK>
RT_DEV RayPrep prepare_ray(V3 o, V3 d)
{
	RayPrep p;
	const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
	/* slab numerators `plane - o`: with the plane coordinates rt_set_scene accepts they are exactly +0 (fine:
	 * div_by_refined(+0, d, r) is the IEEE +-0) or at least 2^-100 in magnitude, unless o itself is a non-zero
	 * number below 2^-100 facing a plane at 0.  (bits - 1 as unsigned: 0 -> 0xffffffff, so zero passes.) */
	const uint32_t otiny = umin3((__float_as_uint(o.x) & 0x7fffffffu) - 1u, (__float_as_uint(o.y) & 0x7fffffffu) - 1u, (__float_as_uint(o.z) & 0x7fffffffu) - 1u);
	p.inv_ok = den_in_window(d.x) && den_in_window(d.y) && den_in_window(d.z) && omax <= 0x1p+29f && (!ORIGIN_CHECK || otiny >= 0x0d800000u - 1u);
	p.inv = mk3(rcp_refined(d.x), rcp_refined(d.y), rcp_refined(d.z));
	p.dd = dot3(d, d);
	p.den = (double) (2.0f * p.dd);
	p.den_ok = near_one(p.dd);
	p.rden = rcp_twice_near_one(p.dd);
	return p;
}

RT_DEV bool wave_all(bool ok) { return __ballot(!ok) == 0ull; }

RT_DEV bool box_entry_fast(V3 o, V3 d, const RayPrep &rp, bool ray_ok, V3 lo, V3 hi, float &t_entry, int &axis_out)
{
	const float n0 = lo.x - o.x, n1 = hi.x - o.x, n2 = lo.y - o.y, n3 = hi.y - o.y, n4 = lo.z - o.z, n5 = hi.z - o.z;
	STAT(1);
	/* ray_ok (wave-uniform, prepare_ray): every lane's direction is inside the window of the shared-reciprocal
	 * division and every numerator is +0 or in [2^-100, 2^30] in magnitude */
	if (ray_ok) {
		const float ax = div_by_refined(n0, d.x, rp.inv.x), bx = div_by_refined(n1, d.x, rp.inv.x);
		const float ay = div_by_refined(n2, d.y, rp.inv.y), by = div_by_refined(n3, d.y, rp.inv.y);
		const float az = div_by_refined(n4, d.z, rp.inv.z), bz = div_by_refined(n5, d.z, rp.inv.z);
		/* All six quotients are finite here, and lo <= hi (checked by rt_set_scene), so the reference's
		 * sign-ordered pairs (scene.c:31-59) are (min, max) and its two overlap tests (scene.c:47,61) are
		 * "the three parameter intervals intersect": max of the entries <= min of the exits.  The entry is
		 * selected with the reference's own strict compares (scene.c:50,64: ties -- two zero quotients of
		 * opposite sign included -- keep the earlier axis), not with v_max. */
		const float nx = __builtin_fminf(ax, bx), fx = __builtin_fmaxf(ax, bx);
		const float ny = __builtin_fminf(ay, by), fy = __builtin_fmaxf(ay, by);
		const float nz = __builtin_fminf(az, bz), fz = __builtin_fmaxf(az, bz);
		const bool y_in = ny > nx;
		const float nxy = y_in ? ny : nx;
		const bool z_in = nz > nxy;
		const float tn = z_in ? nz : nxy;
		const float tf = __builtin_fminf(__builtin_fminf(fx, fy), fz);
		axis_out = z_in ? 2 : (y_in ? 1 : 0);
		t_entry = tn;
		return tn <= tf;
	}
	STAT(2);
	const float ax = n0 / d.x, bx = n1 / d.x, ay = n2 / d.y, by = n3 / d.y, az = n4 / d.z, bz = n5 / d.z;
	const float nx = d.x >= 0 ? ax : bx, fx = d.x >= 0 ? bx : ax;
	const float ny = d.y >= 0 ? ay : by, fy = d.y >= 0 ? by : ay;
	const float nz = d.z >= 0 ? az : bz, fz = d.z >= 0 ? bz : az;
	const bool miss_xy = nx > fy || ny > fx;                 /* scene.c:47 */
	const bool y_in = ny > nx;                               /* scene.c:50 */
	const float tn1 = y_in ? ny : nx;
	const float tf1 = fy < fx ? fy : fx;                     /* scene.c:51 */
	const bool miss_z = tn1 > fz || nz > tf1;                /* scene.c:61 */
	const bool z_in = nz > tn1;                              /* scene.c:64 */
	t_entry = z_in ? nz : tn1;
	axis_out = z_in ? 2 : (y_in ? 1 : 0);
	return !(miss_xy || miss_z);
}

/* sqrt((double) x) for a normal float x <= 2^120: the rsq-seeded coupled (Goldschmidt) step hipcc emits for
 * an IEEE fp64 sqrt, seeded with the cheaper fp32 v_rsq and with ONE residual correction, without the input
 * rescaling and class checks a general double argument needs.  Equal to the correctly rounded fp64 sqrt for
 * every one of the 2.06e9 floats in that range: rt_selftest(3) sweeps them all. */
RT_DEV double sqrt_of_float64(float xf)
{
	const double x = (double) xf;
	const double y = (double) __builtin_amdgcn_rsqf(xf);
	double g = x * y;
	double h = 0.5 * y;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	h = __builtin_fma(h, r, h);
	return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}

RT_DEV bool ball_entry_fast(V3 o, V3 d, const RayPrep &rp, V3 center, float r2, float &t_entry)
{
	const V3 oc = sub3(center, o);
	const float b = -2.0f * dot3(oc, d);
	const float c = dot3(oc, oc) - r2;
	const float discr = b * b - 4.0f * rp.dd * c;
	STAT(3);
	if (!(discr > 0)) return false;
	STAT(4);
	const double nb = (double) -b;
	/* 2a > 0, so the smaller root is the one with `- root` and its sign is its numerator's: the reference's
	 * sort-then-pick (scene.c:119-127) reduces to "the small root if it is >= 0, else the large one if that is
	 * >= 0", and only one quotient is formed.  The one way the float results could disagree with those signs
	 * is a negative quotient so small that it rounds to -0.0f (which `r0 < 0` lets through).  It cannot happen
	 * when |b| >= 2^-90: -b and D = discr are floats, so (-b)^2 and D are equal or differ by at least
	 * 2^-47 (-b)^2, hence |-b - sqrt(D)| is 0 or at least 2^-49 |b| >= 2^-139, and so is the numerator after
	 * rounding the root.  (b == 0, tiny b and an out-of-window 2a or D take the reference-order path.) */
	if (wave_all(rp.den_ok && discr >= 0x1p-126f && discr <= 0x1p+120f && __builtin_fabsf(b) >= 0x1p-90f)) {
		const double root = sqrt_of_float64(discr);
		const double num_lo = nb - root;
		const bool small_ok = num_lo >= 0.0;
		const double num = small_ok ? num_lo : nb + root;
		const float t = (float) div_by_refined64(num, rp.den, rp.rden);
		if (!small_ok) STAT(5);
		if (!small_ok && t < 0) return false;
		t_entry = t;
		return true;
	}
	/* reference order, scene.c:117-127 */
	STAT(6);
	const double root = __builtin_sqrt((double) discr);
	float r0 = (float) ((nb + root) / rp.den);
	float r1 = (float) ((nb - root) / rp.den);
	if (r0 > r1) { const float tmp = r0; r0 = r1; r1 = tmp; }
	if (r0 < 0) { r0 = r1; if (r0 < 0) return false; }
	t_entry = r0;
	return true;
}

RT_DEV Hit nearest_hit_fast(const SceneLDS &sc, int n, V3 o, V3 d, bool want_normal = true)
{
	const RayPrep rp = prepare_ray<true>(o, d);
	const bool ray_ok = wave_all(rp.inv_ok);
	float best_t = 3.402823466e+38f;
	int best_obj = -1, best_axis = 0;
	for (int i = 0; i < n; i++) {
		const float4 g0 = sc.geom[2 * i], g1 = sc.geom[2 * i + 1];
		const int type = __float_as_int(g1.z);
		float t = 0.0f; int axis = 0; bool hit = false;
		if (type == RT_GEOM_CUBE)
			hit = box_entry_fast(o, d, rp, ray_ok, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), t, axis);
		else if (type == RT_GEOM_SPHERE)
			hit = ball_entry_fast(o, d, rp, mk3(g0.x, g0.y, g0.z), g0.w, t);
		if (hit && t >= 0 && t < best_t) { best_t = t; best_obj = i; best_axis = axis; }
	}
	Hit best; best.t = best_t; best.obj = best_obj; best.n = mk3(0, 0, 0);
	if (best_obj >= 0 && want_normal) {   /* shadow taps only need the object (main.c:201-204) */
		const float4 g0 = sc.geom[2 * best_obj], g1 = sc.geom[2 * best_obj + 1];
		if (__float_as_int(g1.z) == RT_GEOM_CUBE) {
			const float dc = best_axis == 0 ? d.x : (best_axis == 1 ? d.y : d.z);
			const float s = dc > 0 ? -1.0f : 1.0f;                               /* scene.c:71-73 */
			best.n = mk3(best_axis == 0 ? s : 0.0f, best_axis == 1 ? s : 0.0f, best_axis == 2 ? s : 0.0f);
		} else
			best.n = unit3_fast(sub3(madd3(o, d, best_t), mk3(g0.x, g0.y, g0.z)));   /* scene.c:146-147 */
	}
	return best;
}

/* ---- large scenes: the same nearest hit, most objects never tested (rt_cull.h has the margins and their proof) -------------
 * Objects are grouped into clusters of up to RT_CLUSTER_SIZE spatially close ones (rt_set_scene).  A ray first tests the
 * clusters' conservative boxes -- all lanes the same cluster, box read once for the wave -- and remembers the ones it may
 * touch in a bit mask; then every lane walks ITS clusters, tests the members' conservative boxes (a slab test on the box
 * inflated by the margin, products by the ray's refined reciprocals) and runs the reference's exact test only on the members
 * that pass.  The exact tests are the very functions of the linear scan, so every distance is the same number; the scan's
 * "lowest index among equal distances" (strict `<` in index order, scene.c:168) is kept explicitly, because the clusters
 * are visited out of index order. */
typedef const __attribute__((address_space(4))) float *rt_const_f;        /* memory read with scalar loads when the address is wave-uniform */
/* recs: the rt_cluster records (RT_CLUSTER_F4 x float4 each) in LDS; mem: the same in memory, for scalar loads; groups / gmem: the rt_group records behind them */
struct ClusterLDS { const float4 *recs; rt_const_f mem; int count; float margin, origin_max; const float4 *groups; rt_const_f gmem; int ngroups; };
/* a wave's scratch for the culled trace: the rays' best hits and the queue of (ray, object) candidates waiting for their exact test --
 * and, before those are in use, the rays' cluster masks as the dealt (ray, group) pairs fill them in, a byte per group (16 bytes per ray) */
#define CULL_QUEUE 128
struct CullWave {
	union {
		struct { unsigned long long best[64]; unsigned short queue[CULL_QUEUE]; };
		uint32_t cmask[64][RT_MAX_CLUSTERS / 32];
	};
};
static_assert(RT_MAX_CLUSTERS / RT_GROUP_SIZE == 16 && RT_GROUP_SIZE == 8, "a ray's cluster mask is sixteen bytes, one per group");

/* the conservative slab test: parameters plane * (1/d) - o * (1/d), one fused multiply-add each (oi = o * inv is formed once per
 * ray).  This is the cull's own arithmetic, not the reference's: its error -- 2^-23 |t| + 2^-24 |o| / |d| <= 3.1e-5 / |d| with
 * coordinates within 64 and origins within 128 -- is what the margin of rt_cull.h (1.95e-3 / |d|) has to cover, 63 times over */
RT_DEV bool slab_may_touch(V3 oi, V3 inv, V3 lo, V3 hi)
{
	const float ax = __builtin_fmaf(lo.x, inv.x, -oi.x), bx = __builtin_fmaf(hi.x, inv.x, -oi.x);
	const float ay = __builtin_fmaf(lo.y, inv.y, -oi.y), by = __builtin_fmaf(hi.y, inv.y, -oi.y);
	const float az = __builtin_fmaf(lo.z, inv.z, -oi.z), bz = __builtin_fmaf(hi.z, inv.z, -oi.z);
	const float enter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
	const float leave = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
	return enter <= leave && leave >= 0.0f;
}

/* Two boxes of a record's box pairs (rt_device.h RT_QPLANE: a cluster's members on the cluster's grid, a group's clusters on the group's)
 * against a ray.  The parameter of grid plane q along an axis is (lo + q step - o) / d = q (step r) + (lo r - o r), one fused multiply-add
 * with the per-(ray, record) constants gs = step r and gb = fma(lo, r, -o r); v_cvt_f32_ubyteN turns byte N of a word into the float q.
 * Which of an axis's two planes the ray meets first is the sign of r: ONE byte permute per word (selectors formed once per ray,
 * grid_selectors) puts it in front, and the slab test is max3 / min3 / one compare -- fma is monotonic in q, so the plane picked by the
 * sign IS the min of the two parameters, bit for bit (tests/csrc/cull_check.cpp compares the two forms).  No parameter can be a NaN
 * (gs, gb finite: the direction window of the cull), so "enter <= leave and leave >= 0" is max(enter, 0) <= leave.  The boxes were rounded
 * outwards on the host (rt_cull.h), so this never misses what the boxes' own conservative tests would catch.  Returns bit 0 / bit 1. */
struct GridSel { uint32_t xy, z; };
RT_DEV GridSel grid_selectors(V3 inv)
{
	GridSel g;      /* word (lo.x hi.x lo.y hi.y): swap the pair of an axis the ray runs down; word (lo.z hi.z | lo.z hi.z of the second box): both or neither */
	g.xy = (inv.x < 0.0f ? 0x0001u : 0x0100u) | (inv.y < 0.0f ? 0x02030000u : 0x03020000u);
	g.z = inv.z < 0.0f ? 0x02030001u : 0x03020100u;
	return g;
}
RT_DEV uint32_t grid_pair_may_touch(uint32_t w0, uint32_t w1, uint32_t w2, GridSel sel, V3 gs, V3 gb)
{
	const uint32_t a = __builtin_amdgcn_perm(w0, w0, sel.xy), b = __builtin_amdgcn_perm(w1, w1, sel.xy), z = __builtin_amdgcn_perm(w2, w2, sel.z);
	uint32_t bits = 0u;
	{
		const float nx = __builtin_fmaf((float) (a & 255u), gs.x, gb.x), fx = __builtin_fmaf((float) ((a >> 8) & 255u), gs.x, gb.x);
		const float ny = __builtin_fmaf((float) ((a >> 16) & 255u), gs.y, gb.y), fy = __builtin_fmaf((float) (a >> 24), gs.y, gb.y);
		const float nz = __builtin_fmaf((float) (z & 255u), gs.z, gb.z), fz = __builtin_fmaf((float) ((z >> 8) & 255u), gs.z, gb.z);
		const float enter = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, 0.0f));
		const float leave = __builtin_fminf(__builtin_fminf(fx, fy), fz);
		if (enter <= leave) bits |= 1u;
	}
	{
		const float nx = __builtin_fmaf((float) (b & 255u), gs.x, gb.x), fx = __builtin_fmaf((float) ((b >> 8) & 255u), gs.x, gb.x);
		const float ny = __builtin_fmaf((float) ((b >> 16) & 255u), gs.y, gb.y), fy = __builtin_fmaf((float) (b >> 24), gs.y, gb.y);
		const float nz = __builtin_fmaf((float) ((z >> 16) & 255u), gs.z, gb.z), fz = __builtin_fmaf((float) (z >> 24), gs.z, gb.z);
		const float enter = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, 0.0f));
		const float leave = __builtin_fminf(__builtin_fminf(fx, fy), fz);
		if (enter <= leave) bits |= 2u;
	}
	return bits;
}
/* all four pairs of a record whose three words of pairs are p0, p1, p2: one bit per box */
RT_DEV uint32_t grid_boxes_may_touch(float4 p0, float4 p1, float4 p2, GridSel sel, V3 gs, V3 gb)
{
	const uint32_t w[12] = { __float_as_uint(p0.x), __float_as_uint(p0.y), __float_as_uint(p0.z), __float_as_uint(p0.w), __float_as_uint(p1.x), __float_as_uint(p1.y),
	                         __float_as_uint(p1.z), __float_as_uint(p1.w), __float_as_uint(p2.x), __float_as_uint(p2.y), __float_as_uint(p2.z), __float_as_uint(p2.w) };
	uint32_t bits = 0u;
#pragma unroll
	for (int p = 0; p < 4; p++) bits |= grid_pair_may_touch(w[3 * p], w[3 * p + 1], w[3 * p + 2], sel, gs, gb) << (2 * p);
	return bits;
}

/* the value lane `src` holds (every lane of the wave must be active: ds_bpermute reads registers of executing lanes) */
RT_DEV float from_lane(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
RT_DEV uint32_t from_lane(uint32_t v, int src) { return (uint32_t) __builtin_amdgcn_ds_bpermute(src << 2, (int) v); }

/* ALL 64 lanes must call this (on: the lane has a ray); best = 64 words of 8 bytes of LDS of the calling wave.
 *
 * Step 2 is shared out evenly: with the first, Morton-ordered clusters a ray touched 11 clusters on average but the busiest of 64
 * rays 31 (1024 random objects, profiles/r04/stats_large_1024.txt; median-split clusters: 6.6 on average) -- walking its own
 * clusters, a wave took 31 steps at a quarter of its lanes.  Instead the
 * (ray, cluster) pairs of the whole wave are numbered by a prefix sum over the lanes' counts and dealt 64 at a time: lane i
 * takes pair 64 k + i, finds the lane whose ray it is by a binary search over the prefix sums, picks that lane's r-th cluster
 * out of its mask, fetches the ray with ds_bpermute (no LDS memory: registers of another lane), tests the members' conservative
 * boxes and queues what is left for the exact tests, which run on full batches (step 3); a hit goes to its ray with one 64-bit LDS minimum on the packed
 * (distance, object index, -0 flag, entry axis) -- distances are >= 0, so their bit patterns order like the numbers, and among
 * equal distances the lower index wins, as in the reference's scan (scene.c:168). */
RT_DEV Hit nearest_hit_culled(const SceneLDS &sc, int n, const ClusterLDS &cl, CullWave *cw, bool on, V3 o, V3 d, bool want_normal = true)
{
	const int lane = threadIdx.x & 63;
	unsigned long long *best = cw->best;
	const RayPrep rp = prepare_ray<true>(o, d);
	const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
	/* the margins are proved for directions inside the window of the shared-reciprocal division and origins within twice the
	 * scene's extent; a wave with any other ray tests every object */
	if (!wave_all(!on || (rp.inv_ok && omax <= cl.origin_max))) {
		Hit h; h.t = 0.0f; h.obj = -1; h.n = mk3(0, 0, 0);
		if (on) h = nearest_hit_fast(sc, n, o, d, want_normal);
		return h;
	}
	if (on) STAT(9);
	/* 1. which clusters may this ray touch? */
	uint32_t mask[RT_MAX_CLUSTERS / 32];
	const V3 oi = mk3(o.x * rp.inv.x, o.y * rp.inv.y, o.z * rp.inv.z);
	/* the first lane whose inclusive prefix sum exceeds q: whose pair number q is (steps 1b and 2 deal pairs to the lanes) */
	auto owner_of = [&](uint32_t sums, uint32_t q) {
		int src = 0;
#pragma unroll
		for (int bit = 32; bit >= 1; bit >>= 1) {
			const uint32_t v = from_lane(sums, src + bit - 1);
			if (v <= q) src += bit;
		}
		return src > 63 ? 63 : src;
	};
	/* every cluster box for every ray, wave-uniformly -- one box per step, read once for all lanes: two scalar loads, planes as scalar
	 * operands of the FMAs (the LDS pipe is what the culled trace is short of).  Scenes of few clusters; and a wave whose rays pass
	 * most of the groups (below) */
	auto ask_every_cluster = [&]() {
#pragma unroll
		for (int w = 0; w < RT_MAX_CLUSTERS / 32; w++) {
			uint32_t bits = 0u;
			const int first = 32 * w, last = cl.count < first + 32 ? cl.count : first + 32;
			for (int c = first; c < last; c++) {
				STAT(32);
				const rt_const_f km = cl.mem + 4 * RT_CLUSTER_F4 * c;
				float4 k0, k1;
				k0.x = km[0]; k0.y = km[1]; k0.z = km[2]; k0.w = km[3]; k1.x = km[4]; k1.y = km[5];
				if (slab_may_touch(oi, rp.inv, mk3(k0.x, k0.y, k0.z), mk3(k0.w, k1.x, k1.y))) bits |= 1u << (c - first);
			}
			mask[w] = on ? bits : 0u;
		}
	};
	bool every_cluster = cl.count < RT_GROUPS_FROM_CLUSTERS;       /* (wave-uniform) */
	if (!every_cluster) {
		/* 1a. the GROUPS of eight clusters, wave-uniformly: one box per step, read once for all lanes (scalar loads, planes as scalar
		 * operands of the FMAs) -- sixteen steps for 1024 objects where asking every cluster box took 128 */
		uint32_t gbits = 0u;
		for (int k = 0; k < cl.ngroups; k++) {
			STAT(43);
			const rt_const_f km = cl.gmem + 4 * RT_GROUP_F4 * k;
			float4 k0, k1;
			k0.x = km[0]; k0.y = km[1]; k0.z = km[2]; k0.w = km[3]; k1.x = km[4]; k1.y = km[5];
			if (slab_may_touch(oi, rp.inv, mk3(k0.x, k0.y, k0.z), mk3(k0.w, k1.x, k1.y))) gbits |= 1u << k;
		}
		if (!on) gbits = 0u;
		/* 1b. the wave's (ray, group) pairs, numbered by a prefix sum over the lanes' counts and dealt 64 at a time, as the (ray, cluster)
		 * pairs of step 2 are: lane i takes pair 64 k + i, fetches the ray from its lane's registers, tests the group's eight cluster
		 * boxes on the group's grid (the member-box test) and writes the eight answers as ONE BYTE of that ray's mask in LDS */
		*reinterpret_cast<uint4*>(cw->cmask[lane]) = make_uint4(0u, 0u, 0u, 0u);
		const uint32_t gcount = (uint32_t) __popc(gbits);
		uint32_t gupto = gcount;
#pragma unroll
		for (int step = 1; step < 64; step <<= 1) {
			const uint32_t below = from_lane(gupto, lane >= step ? lane - step : lane);
			if (lane >= step) gupto += below;
		}
		const uint32_t gtotal = (uint32_t) __builtin_amdgcn_readlane((int) gupto, 63);
		wave_fence();
		/* A dealt step costs about as much as eighteen uniform box tests: a wave whose rays pass MOST groups -- objects that span the
		 * scene, groups that overlap -- would pay more for its pairs than for asking every cluster box, which costs the same whatever the
		 * rays do: such a wave does that.  (Either way the mask holds every cluster a ray may touch: the two differ in which boxes that
		 * the ray misses they let through.) */
		if (gtotal > RT_GROUP_PAIRS_MAX) every_cluster = true;
		else {
		for (uint32_t base = 0; base < gtotal; base += 64u) {
			STAT(44);
			const uint32_t q = base + (uint32_t) lane;
			const bool mine = q < gtotal;
			const int src = owner_of(gupto, q);
			uint32_t r = q - (from_lane(gupto, src) - from_lane(gcount, src));        /* its r-th group */
			uint32_t gm = from_lane(gbits, src);
			const V3 sinv = mk3(from_lane(rp.inv.x, src), from_lane(rp.inv.y, src), from_lane(rp.inv.z, src));
			const V3 soi = mk3(from_lane(oi.x, src), from_lane(oi.y, src), from_lane(oi.z, src));
			if (mine) {
				int pos = 0;
				uint32_t t_;
				t_ = (uint32_t) __popc(gm & 0xffu); if (r >= t_) { pos += 8; r -= t_; gm >>= 8; }
				t_ = (uint32_t) __popc(gm & 0xfu);  if (r >= t_) { pos += 4; r -= t_; gm >>= 4; }
				t_ = (uint32_t) __popc(gm & 0x3u);  if (r >= t_) { pos += 2; r -= t_; gm >>= 2; }
				t_ = gm & 1u;                       if (r >= t_) { pos += 1; }
				const float4 *rec = cl.groups + RT_GROUP_F4 * pos;
				const float4 k0 = rec[0], k1 = rec[1];
				const V3 gs = mk3(RT_CLUSTER_STEP(k0.x, k0.w) * sinv.x, RT_CLUSTER_STEP(k0.y, k1.x) * sinv.y, RT_CLUSTER_STEP(k0.z, k1.y) * sinv.z);
				const V3 gb = mk3(__builtin_fmaf(k0.x, sinv.x, -soi.x), __builtin_fmaf(k0.y, sinv.y, -soi.y), __builtin_fmaf(k0.z, sinv.z, -soi.z));
				STAT(45);
				uint32_t touched = grid_boxes_may_touch(rec[2], rec[3], rec[4], grid_selectors(sinv), gs, gb);      /* the group's eight clusters */
				touched &= (1u << __float_as_int(k1.z)) - 1u;      /* (the last group may have fewer clusters: their slots hold zeros) */
				reinterpret_cast<unsigned char*>(cw->cmask[src])[pos] = (unsigned char) touched;      /* clusters 8 pos ... 8 pos + 7: byte pos of the ray's sixteen */
			}
		}
		wave_fence();
		const uint4 mine4 = *reinterpret_cast<const uint4*>(cw->cmask[lane]);
		mask[0] = mine4.x; mask[1] = mine4.y; mask[2] = mine4.z; mask[3] = mine4.w;
		}
		wave_fence();                   /* (the masks share their LDS with `best` and the queue, which are written from here on) */
	}
	if (every_cluster) ask_every_cluster();
	static_assert(RT_MAX_CLUSTERS == 128, "the pair numbering below reads a lane's clusters as four 32-bit words");
	/* 2. the wave's (ray, cluster) pairs, numbered: inclusive prefix sum of the lanes' counts */
	const uint32_t count = (uint32_t) (__popc(mask[0]) + __popc(mask[1]) + __popc(mask[2]) + __popc(mask[3]));
	uint32_t upto = count;
#pragma unroll
	for (int step = 1; step < 64; step <<= 1) {
		const uint32_t below = from_lane(upto, lane >= step ? lane - step : lane);
		if (lane >= step) upto += below;
	}
	const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) upto, 63);
	best[lane] = ~0ull;
	wave_fence();
	/* 3 (declared first). the exact tests -- the reference's own, on full batches: a (ray, cluster) pair leaves 0.4 members on
	 * average and one lane in five with any, so the members that pass their conservative box go into a queue of (ray, object) and
	 * are tested 64 at a time by whichever lane gets them (the ray fetched from its lane's registers again) */
	uint32_t cq_head = 0u, cq_tail = 0u;
	auto test_queued = [&](uint32_t batch) {               /* the `batch` <= 64 oldest candidates */
		STAT(35);
		const bool mine = (uint32_t) lane < batch;
		const uint32_t e = mine ? (uint32_t) cw->queue[(cq_head + (uint32_t) lane) & (CULL_QUEUE - 1)] : (uint32_t) lane;
		const int src = (int) (e & 63u);
		const uint32_t idx = e >> 6;
		const V3 so = mk3(from_lane(o.x, src), from_lane(o.y, src), from_lane(o.z, src));
		const V3 sd = mk3(from_lane(d.x, src), from_lane(d.y, src), from_lane(d.z, src));
		RayPrep sp;
		sp.inv = mk3(from_lane(rp.inv.x, src), from_lane(rp.inv.y, src), from_lane(rp.inv.z, src));
		sp.dd = from_lane(rp.dd, src);
		if (mine) {
			sp.inv_ok = true;
			sp.den = (double) (2.0f * sp.dd); sp.den_ok = near_one(sp.dd); sp.rden = rcp_twice_near_one(sp.dd);     /* as prepare_ray() forms them */
			const float4 g0 = sc.geom[2 * idx], g1 = sc.geom[2 * idx + 1];
			float t = 0.0f; int axis = 0; bool hit = false;
			if (__float_as_int(g1.z) == RT_GEOM_CUBE)
				hit = box_entry_fast(so, sd, sp, true, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), t, axis);
			else
				hit = ball_entry_fast(so, sd, sp, mk3(g0.x, g0.y, g0.z), g0.w, t);
			if (hit && t >= 0) {
				const uint32_t tb = __float_as_uint(t);                          /* t >= 0: +0 ... +inf, or -0 (0x80000000), which orders as 0 */
				const unsigned long long key = ((unsigned long long) (tb & 0x7fffffffu) << 32) | ((unsigned long long) idx << 3) | (unsigned long long) ((tb >> 31) << 2) | (unsigned long long) axis;
				__hip_atomic_fetch_min(best + src, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			}
		}
		cq_head += batch;
		wave_fence();
	};
	for (uint32_t base = 0; base < total; base += 64u) {
		STAT(33);
		const uint32_t q = base + (uint32_t) lane;
		const bool mine = q < total;
		const int src = owner_of(upto, q);                                       /* whose ray */
		uint32_t r = q - (from_lane(upto, src) - from_lane(count, src));        /* its r-th cluster */
		const uint32_t m0 = from_lane(mask[0], src), m1 = from_lane(mask[1], src), m2 = from_lane(mask[2], src), m3 = from_lane(mask[3], src);
		const V3 sinv = mk3(from_lane(rp.inv.x, src), from_lane(rp.inv.y, src), from_lane(rp.inv.z, src));
		const V3 soi = mk3(from_lane(oi.x, src), from_lane(oi.y, src), from_lane(oi.z, src));
		const unsigned short *member = reinterpret_cast<const unsigned short*>(cl.recs + 2);
		uint32_t cand = 0u;
		if (mine) {
			int word = 0;
			uint32_t mm = m0;
			const uint32_t c0 = (uint32_t) __popc(m0), c1 = (uint32_t) __popc(m1), c2 = (uint32_t) __popc(m2);
			if (r >= c0) { r -= c0; word = 1; mm = m1; if (r >= c1) { r -= c1; word = 2; mm = m2; if (r >= c2) { r -= c2; word = 3; mm = m3; } } }
			int pos = 0;
			uint32_t t_;
			t_ = (uint32_t) __popc(mm & 0xffffu); if (r >= t_) { pos += 16; r -= t_; mm >>= 16; }
			t_ = (uint32_t) __popc(mm & 0xffu);   if (r >= t_) { pos += 8;  r -= t_; mm >>= 8; }
			t_ = (uint32_t) __popc(mm & 0xfu);    if (r >= t_) { pos += 4;  r -= t_; mm >>= 4; }
			t_ = (uint32_t) __popc(mm & 0x3u);    if (r >= t_) { pos += 2;  r -= t_; mm >>= 2; }
			t_ = mm & 1u;                         if (r >= t_) { pos += 1; }
			const int c = 32 * word + pos;
			/* the cluster's record: five reads of 16 bytes (box and count, three words of member boxes; the member indices only if a member
			 * is touched) where the members' geometry records took sixteen -- and seven slots apart, the lanes' reads spread over the
			 * whole bank row */
			const float4 *rec = cl.recs + RT_CLUSTER_F4 * c;
			member = reinterpret_cast<const unsigned short*>(rec + 2);
			const float4 k0 = rec[0], k1 = rec[1];
			const V3 gs = mk3(RT_CLUSTER_STEP(k0.x, k0.w) * sinv.x, RT_CLUSTER_STEP(k0.y, k1.x) * sinv.y, RT_CLUSTER_STEP(k0.z, k1.y) * sinv.z);
			const V3 gb = mk3(__builtin_fmaf(k0.x, sinv.x, -soi.x), __builtin_fmaf(k0.y, sinv.y, -soi.y), __builtin_fmaf(k0.z, sinv.z, -soi.z));
			STAT(34);
			cand = grid_boxes_may_touch(rec[3], rec[4], rec[5], grid_selectors(sinv), gs, gb);      /* the cluster's eight members */
			cand &= (1u << __float_as_int(k1.z)) - 1u;          /* (the last cluster may have fewer members: their slots hold zeros) */
		}
		/* the members that passed: into the queue, one per lane at a time (at most 63 wait there, so 64 more always fit) */
		for (;;) {
			const bool has = cand != 0u;
			const unsigned long long pm = __ballot(has);
			if (pm == 0ull) break;
			if (has) {
				const uint32_t idx = member[__builtin_ctz(cand)];
				cand &= cand - 1u;
				cw->queue[(cq_tail + (uint32_t) lanes_below(pm)) & (CULL_QUEUE - 1)] = (unsigned short) ((uint32_t) src | (idx << 6));
			}
			cq_tail += (uint32_t) __popcll(pm);
			wave_fence();
			if (cq_tail - cq_head >= 64u) test_queued(64u);
		}
	}
	if (cq_tail != cq_head) test_queued(cq_tail - cq_head);
	wave_fence();
	const unsigned long long won = best[lane];
	Hit hit; hit.t = 3.402823466e+38f; hit.obj = -1; hit.n = mk3(0, 0, 0);
	if (on && won != ~0ull) {
		const int best_axis = (int) (won & 3ull), best_obj = (int) ((won >> 3) & 0xffffull);
		hit.t = __uint_as_float((uint32_t) (won >> 32) | ((uint32_t) ((won >> 2) & 1ull) << 31));
		hit.obj = best_obj;
		if (want_normal) {
			const float4 g0 = sc.geom[2 * best_obj], g1 = sc.geom[2 * best_obj + 1];
			if (__float_as_int(g1.z) == RT_GEOM_CUBE) {
				const float dc = best_axis == 0 ? d.x : (best_axis == 1 ? d.y : d.z);
				const float s = dc > 0 ? -1.0f : 1.0f;                               /* scene.c:71-73 */
				hit.n = mk3(best_axis == 0 ? s : 0.0f, best_axis == 1 ? s : 0.0f, best_axis == 2 ? s : 0.0f);
			} else
				hit.n = unit3_fast(sub3(madd3(o, d, hit.t), mk3(g0.x, g0.y, g0.z)));   /* scene.c:146-147 */
		}
	}
	wave_fence();                   /* (the next call of this wave clears `best` again) */
	return hit;
}

/* the cluster records in LDS (for the lanes' reads; the wave-uniform pass over the cluster boxes uses scalar loads from memory) */
RT_DEV ClusterLDS stage_clusters(const rt_launch &L, float4 *dst)
{
	const float4 *src = reinterpret_cast<const float4*>(L.clusters);
	for (int i = threadIdx.x; i < RT_CULL_F4(L.num_clusters); i += (int) blockDim.x) dst[i] = src[i];      /* the cluster records and, behind them, the group records */
	__syncthreads();
	ClusterLDS cl; cl.recs = dst; cl.mem = (rt_const_f) (unsigned long long) L.clusters; cl.count = L.num_clusters; cl.margin = L.cull_margin; cl.origin_max = L.cull_origin_max;
	cl.groups = dst + RT_CLUSTER_F4 * L.num_clusters; cl.gmem = cl.mem + 4 * RT_CLUSTER_F4 * L.num_clusters; cl.ngroups = RT_NUM_GROUPS(L.num_clusters);
	return cl;
}

/* ---- scene-specialised trace (rt_compile_scene, rt_jit.cpp) ----------------------------------------
 * With the geometry known at compile time the object loop unrolls, boxes that share slab planes share
 * their quotients (common-subexpression elimination of identical exact chains), geometry needs no LDS
 * reads. */
#ifdef RT_SPEC_HEADER
#include RT_SPEC_HEADER
RT_DEV Hit nearest_hit_spec(const SceneLDS &sc, int n, V3 o, V3 d, bool want_normal)
{
	const RayPrep rp = prepare_ray<true>(o, d);
	if (!wave_all(rp.inv_ok))
		return nearest_hit_fast(sc, n, o, d, want_normal);
	float best_t = 3.402823466e+38f;
	int best_obj = -1, best_axis = 0;
#pragma unroll
	for (int i = 0; i < SPEC_N; i++) {
		float t = 0.0f; int axis = 0; bool hit = false;
		if (SPEC_T[i] == RT_GEOM_CUBE) {
			const float ax = div_by_refined(SPEC_G[i][0] - o.x, d.x, rp.inv.x), bx = div_by_refined(SPEC_G[i][3] - o.x, d.x, rp.inv.x);
			const float ay = div_by_refined(SPEC_G[i][1] - o.y, d.y, rp.inv.y), by = div_by_refined(SPEC_G[i][4] - o.y, d.y, rp.inv.y);
			const float az = div_by_refined(SPEC_G[i][2] - o.z, d.z, rp.inv.z), bz = div_by_refined(SPEC_G[i][5] - o.z, d.z, rp.inv.z);
			const float nx = __builtin_fminf(ax, bx), fx = __builtin_fmaxf(ax, bx);
			const float ny = __builtin_fminf(ay, by), fy = __builtin_fmaxf(ay, by);
			const float nz = __builtin_fminf(az, bz), fz = __builtin_fmaxf(az, bz);
			const bool y_in = ny > nx;                      /* scene.c:50,64: strict, ties (and zeros of either sign) keep the earlier axis */
			const float nxy = y_in ? ny : nx;
			const bool z_in = nz > nxy;
			t = z_in ? nz : nxy;
			const float tf = __builtin_fminf(__builtin_fminf(fx, fy), fz);
			axis = z_in ? 2 : (y_in ? 1 : 0);
			hit = t <= tf;
		} else if (SPEC_T[i] == RT_GEOM_SPHERE)
			hit = ball_entry_fast(o, d, rp, mk3(SPEC_G[i][0], SPEC_G[i][1], SPEC_G[i][2]), SPEC_G[i][3], t);
		if (hit && t >= 0 && t < best_t) { best_t = t; best_obj = i; best_axis = axis; }
	}
	Hit best; best.t = best_t; best.obj = best_obj; best.n = mk3(0, 0, 0);
	if (best_obj >= 0 && want_normal) {
		const float4 g0 = sc.geom[2 * best_obj], g1 = sc.geom[2 * best_obj + 1];
		if (__float_as_int(g1.z) == RT_GEOM_CUBE) {
			const float dc = best_axis == 0 ? d.x : (best_axis == 1 ? d.y : d.z);
			const float s = dc > 0 ? -1.0f : 1.0f;
			best.n = mk3(best_axis == 0 ? s : 0.0f, best_axis == 1 ? s : 0.0f, best_axis == 2 ? s : 0.0f);
		} else
			best.n = unit3_fast(sub3(madd3(o, d, best_t), mk3(g0.x, g0.y, g0.z)));
	}
	return best;
}
#define NEAREST_HIT_TUNED nearest_hit_spec
#else
#define NEAREST_HIT_TUNED nearest_hit_fast
#endif

/* ---- skybox: gpu_and_windowing.c:42-112 ---------------------------------------------------- */

template <bool FAST = false>
RT_DEV uint32_t sky_texel(const rt_launch &L, V3 dir)
{
	/* absf() (utils.c): x < 0 ? -x : x keeps -0, and `abs + eps` with eps = 0 (gpu_and_windowing.c:51-58) turns
	 * it into +0; |x| gives the same compares and the same sums */
	const float ax = FAST ? __builtin_fabsf(dir.x) : (dir.x < 0 ? -dir.x : dir.x);
	const float ay = FAST ? __builtin_fabsf(dir.y) : (dir.y < 0 ? -dir.y : dir.y);
	const float az = FAST ? __builtin_fabsf(dir.z) : (dir.z < 0 ? -dir.z : dir.z);
	int face; float nu, nv, m;                     /* u = nu / m, v = nv / m */
	if (ax > ay && ax > az) {
		m = ax + 0.0f;
		if (dir.x > 0) { face = 3; nu = -dir.z; nv = -dir.y; }          /* CF_RIGHT  */
		else           { face = 2; nu =  dir.z; nv = -dir.y; }          /* CF_LEFT   */
	} else if (ay > ax && ay > az) {
		m = ay + 0.0f;
		if (dir.y > 0) { face = 4; nu = dir.x; nv =  dir.z; }           /* CF_TOP    */
		else           { face = 5; nu = dir.x; nv = -dir.z; }           /* CF_BOTTOM */
	} else {
		m = az + 0.0f;
		if (dir.z > 0) { face = 0; nu =  dir.x; nv = -dir.y; }          /* CF_FRONT  */
		else           { face = 1; nu = -dir.x; nv = -dir.y; }          /* CF_BACK   */
	}
	float u, v;
	/* Shared-reciprocal quotients (rt_math.hip.h).  A numerator below the window (zero, -0, tiny) needs no care
	 * here: its quotient is below 2^-25 in magnitude either way and u + 1.0f below rounds it away. */
	if (FAST && wave_all(m >= 0x1p-30f && __builtin_fmaxf(__builtin_fmaxf(ax, ay), az) <= 0x1p+20f)) {
		const float r = rcp_refined(m);             /* (finite quotients: clamp(x, -1, 1) is the median of the three) */
		u = __builtin_amdgcn_fmed3f(div_by_refined(nu, m, r), -1.0f, 1.0f);
		v = __builtin_amdgcn_fmed3f(div_by_refined(nv, m, r), -1.0f, 1.0f);
	} else {
		u = clamp11(nu / m);
		v = clamp11(nv / m);
	}
	u = 0.5f * (u + 1.0f);
	v = 0.5f * (v + 1.0f);
	const int x = (int) (u * L.sky_wm1);           /* int x = u * (c->w - 1)  (gpu_and_windowing.c:103-104) */
	const int y = (int) (v * L.sky_hm1);
	/* texel index in the reference's own int arithmetic (:106); rt_set_skybox bounds 6*w*h below 2^31 */
	return L.sky[(uint32_t) ((face * L.sky_h + y) * L.sky_w + x)];
}

/* the three colour channels of an RGBA8 texel: (float) color[k] / 255 (gpu_and_windowing.c:107-111) */
template <bool FAST = false>
RT_DEV V3 sky_colour(uint32_t texel)
{
	if (FAST) {                                     /* (float) b / 255 (gpu_and_windowing.c:108-110) with the literal RN(1/255):
	                                                 * exact for the 256 possible numerators, rt_selftest(4) */
		const float r255 = __uint_as_float(0x3b808081u);
		return mk3(div_by_refined((float) (texel & 255u), 255.0f, r255), div_by_refined((float) ((texel >> 8) & 255u), 255.0f, r255),
		           div_by_refined((float) ((texel >> 16) & 255u), 255.0f, r255));
	}
	return mk3((float) (texel & 255u) / 255.0f,
	           (float) ((texel >> 8) & 255u) / 255.0f,
	           (float) ((texel >> 16) & 255u) / 255.0f);
}

template <bool FAST = false>
RT_DEV V3 sky_lookup(const rt_launch &L, V3 dir) { return sky_colour<FAST>(sky_texel<FAST>(L, dir)); }

/* ---- pixel mapping --------------------------------------------------------------------------- */


RT_DEV int global_row(int row_block, int rank, int world, int local_row)
{
	return ((local_row / row_block) * world + rank) * row_block + local_row % row_block;
}
RT_DEV int global_row(const rt_launch &L, int local_row) { return global_row(L.row_block, L.rank, L.world, local_row); }

/* camera.c:121 with the frame constants of camera.c:99-118 hoisted to the host */
RT_DEV V3 primary_dir(const rt_launch &L, float px, float py)
{
	return mk3(L.llc[0] + L.horiz[0] * px + L.vert[0] * py - L.pos[0],
	           L.llc[1] + L.horiz[1] * px + L.vert[1] * py - L.pos[1],
	           L.llc[2] + L.horiz[2] * px + L.vert[2] * py - L.pos[2]);
}

#ifndef RT_SPEC_ONLY
/* =============================================================================================
 * rt_trace_simple: the path loop in the reference's own order (main.c:131-272), one lane per pixel.
 * Kept as the in-GPU cross-check for the tuned kernel (tests compare the two at full frame sizes).
 * ============================================================================================= */
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_trace_simple(const rt_launch L)
{
	extern __shared__ float4 lds[];
	const int n = L.num_objects;
	const SceneLDS sc = stage_scene(L, lds, n);

	const int tiles_x = (L.width + RT_TILE_W - 1) / RT_TILE_W;
	const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int i  = tile_x * RT_TILE_W + wave * 8 + (lane & 7);
	const int lr = tile_y * RT_TILE_H + (lane >> 3);
	if (i >= L.width || lr >= L.local_rows) return;
	const int j = global_row(L, lr);
	if (j >= L.height) return;

	/* main.c:293-296 at scale 1 */
	float u = (float) i / (float) L.u_den;
	float v = (float) j / (float) L.v_den;
	u = 1.0f - u;
	v = 1.0f - v;
	const V3 cam = ld3(L.pos);
	const V3 dir0 = primary_dir(L, u, v);
	const uint32_t pixel_index = (uint32_t) ((j * L.pix_scale) * L.pix_width + i * L.pix_scale);
	const V3 light_pos = ld3(L.light_pos);

	V3 sum = mk3(0, 0, 0);
	for (int s = 0; s < L.spp; s++) {
		uint64_t rng = path_seed(L.seed, pixel_index, (uint32_t) (L.sample_base + s));
		V3 ro = cam, rd = dir0;
		V3 carry = mk3(1, 1, 1), radiance = mk3(0, 0, 0);

		for (int bounce = 0; bounce < L.max_bounces; bounce++) {
			const V3 dn = unit3(rd);
			const Hit hit = nearest_hit(sc, n, ro, dn);
			if (hit.obj < 0) {
				radiance = add3(radiance, had3(sky_lookup<false>(L, dn), carry));   /* main.c:170-171 */
				break;
			}
			const V3 point = madd3(ro, dn, hit.t);                           /* scene.c:186 */

			/* main.c:180-210 */
			V3 lit = mk3(0, 0, 0);
			if (L.light_index >= 0) {
				const V3 to_light = sub3(light_pos, point);
				int taps = 0;
				for (int k = 0; k < 3; k++) {
					const V3 jitter = rng_direction(rng);
					if (dot3(jitter, hit.n) <= 0) continue;
					const V3 sd = unit3(lin2(jitter, to_light, 0.5f, 1.0f));
					const V3 so = madd3(point, sd, 0.001f);
					const Hit blocker = nearest_hit(sc, n, so, unit3(sd));
					if (blocker.obj >= 0) {
						const float4 e = sc.shade[4 * blocker.obj + 3];
						lit = add3(lit, mk3(e.x, e.y, e.z));
					}
					taps++;
				}
				if (taps > 0) lit = scale3(lit, 1.0f / (float) taps);
			}

			const float4 m0 = sc.shade[4 * hit.obj], m1 = sc.shade[4 * hit.obj + 1];
			const float4 m2 = sc.shade[4 * hit.obj + 2], m3 = sc.shade[4 * hit.obj + 3];
			const V3 f0 = mk3(m0.x, m0.y, m0.z), omf0 = mk3(m1.x, m1.y, m1.z);
			const float rough = m0.w;
			const bool is_metal = __float_as_int(m1.w) != 0;

			const float n_dot_v = clamp01(dot3(hit.n, neg3(rd)));            /* main.c:214-216 */
			/* main.c:128: (float)pow(1.0 - (double)u, 5.0) == x2*x2*x in fp64 for u in [0,1]
			 * (pinned exhaustively by tests/test_pow5.py) */
			const double xg = 1.0 - (double) n_dot_v;
			const double xg2 = xg * xg;
			const float grazing = (float) (xg2 * xg2 * xg);
			const V3 fresnel = madd3(f0, omf0, grazing);

			V3 scatter = rng_direction(rng);                                 /* main.c:226-228 */
			if (dot3(scatter, hit.n) < 0) scatter = neg3(scatter);

			radiance = add3(radiance, had3(mk3(m3.x, m3.y, m3.z), carry));   /* main.c:232 */

			V3 out_dir;
			bool specular = is_metal;
			if (!specular)
				specular = rng_draw(rng) <= (fresnel.x + fresnel.y + fresnel.z) / 3.0f;
			if (specular) {
				const V3 nneg = neg3(hit.n);
				const float f = -2.0f * dot3(nneg, rd);                      /* vector.c:113-117 */
				const V3 refl = madd3(rd, nneg, f);
				out_dir = unit3(lin2(scatter, refl, rough, 1.0f));
			} else {
				out_dir = scatter;
				carry = had3(carry, mk3(m2.x, m2.y, m2.z));
			}
			const V3 next_o = madd3(point, out_dir, 0.001f);                 /* main.c:250 */

			if (!(tiny_f(lit.x) && tiny_f(lit.y) && tiny_f(lit.z))) {        /* main.c:257-261 */
				const float w = 0.05f;
				radiance = madd3(radiance, had3(lit, carry), w);
				carry = scale3(carry, 1.0f - w);
			}
			ro = next_o; rd = out_dir;
		}
		sum = add3(sum, mk3(clamp01(radiance.x), clamp01(radiance.y), clamp01(radiance.z)));
	}

	const V3 res = scale3(sum, 1.0f / (float) L.spp);                        /* main.c:476 */
	float *dst = L.frame + ((size_t) lr * L.width + i) * 3;
	dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
}

#endif /* RT_SPEC_ONLY */

#ifndef RT_SPEC_ONLY
/* =============================================================================================
 * rt_primary_pass: the camera ray of every pixel, once per launch.  The reference has no sub-pixel jitter
 * (main.c:293-296), so bounce 0 of every sample of a pixel is the same ray: one wave per 8x8 pixel block traces
 * the block's 64 camera rays, finishes its sky-only pixels on the spot -- every sample is clamp(0 + sky * 1)
 * (main.c:171,267-269), summed in sample order and resolved (main.c:394,476) -- and appends a 12-word record
 * per OBJECT pixel (hit point, normal, object, camera ray, RNG pixel index, frame offset) to one of
 * L.num_shards pixel lists, from which the trace kernel's waves deal pixels to their lanes.
 * ============================================================================================= */
#define RT_PIX_WORDS 12
template <bool FAST, bool CULL = false>
__global__ void __launch_bounds__(RT_BLOCK)
rt_primary_pass(const rt_launch L, int blocks_per_group)
{
	extern __shared__ float4 lds[];
	const int n = L.num_objects;
	const SceneLDS sc = CULL ? stage_geometry(L) : stage_scene(L, lds, n);
	ClusterLDS cl; cl.recs = nullptr; cl.mem = nullptr; cl.count = 0; cl.margin = 0.0f; cl.origin_max = 0.0f;
	if (CULL) cl = stage_clusters(L, lds);
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	CullWave *cull_wave = reinterpret_cast<CullWave*>(lds + RT_CULL_F4(L.num_clusters)) + wave;     /* (CULL) */
	const int tiles_x = (L.width + 7) >> 3, tiles_y = (L.local_rows + 7) >> 3;
	const unsigned int total = (unsigned int) (tiles_x * tiles_y);
	/* the workgroup's share of the blocks: blocks_per_group consecutive ones.  Its object pixels go to list
	 * blockIdx.x % num_shards: one atomic per block with object pixels, spread over the lists' counters (a single
	 * address takes ~88 atomics per microsecond: one counter for all blocks would cost more than the camera rays) */
	const unsigned int shard = blockIdx.x % (unsigned int) L.num_shards;
	const V3 cam = ld3(L.pos);
	const float inv_spp = L.sum_onto ? 1.0f : 1.0f / (float) L.spp;
	const unsigned int first = blockIdx.x * (unsigned int) blocks_per_group;
	unsigned int blocks_done = 0u;
	for (unsigned int k = (unsigned int) wave; k < (unsigned int) blocks_per_group; k += RT_BLOCK / 64, blocks_done++) {
		const unsigned int blk = first + k;
		if (blk >= total) break;
		const int i = (int) (blk % (unsigned int) tiles_x) * 8 + (lane & 7), lr = (int) (blk / (unsigned int) tiles_x) * 8 + (lane >> 3);
		const int j = global_row(L, lr);
		int obj = -2, known = 0;                        /* outside the frame */
		V3 a = mk3(0, 0, 0), nn = mk3(0, 0, 0), pd = mk3(0, 0, 0);
		const bool inside = i < L.width && lr < L.local_rows && j < L.height;
		Hit culled; culled.t = 0.0f; culled.obj = -1; culled.n = mk3(0, 0, 0);
		if (CULL) {         /* the culled trace shares its work out over the wave: every lane takes part, with or without a ray of its own */
			float u = 1.0f - (float) (inside ? i : 0) / (float) L.u_den, v = 1.0f - (float) (inside ? j : 0) / (float) L.v_den;
			const V3 cd = unit3_fast(primary_dir(L, u, v));
			culled = nearest_hit_culled(sc, n, cl, cull_wave, inside, cam, cd);
		}
		if (inside) {
			float u = (float) i / (float) L.u_den;      /* main.c:293-296 */
			float v = (float) j / (float) L.v_den;
			u = 1.0f - u;
			v = 1.0f - v;
			pd = primary_dir(L, u, v);
			const V3 dn = FAST ? unit3_fast(pd) : unit3(pd);                           /* scene.c:158 */
			const Hit hit = CULL ? culled : (FAST ? nearest_hit_fast(sc, n, cam, dn) : nearest_hit(sc, n, cam, dn));
			obj = hit.obj;
			if (obj >= 0) {
				a = madd3(cam, dn, hit.t);                                               /* scene.c:186 */
				nn = hit.n;
				/* every soft-shadow tap from this point certainly hits the emitter first (rt_lit.h): bounce 0 of all the
				 * pixel's samples needs no tap traced */
				if (FAST && L.skip_known_taps) {
					const int cls = rt_taps_class(reinterpret_cast<const float*>(sc.geom), n, L.light_index, L.light_pos[0], L.light_pos[1], L.light_pos[2],
					                              L.only_light_emits, obj, a.x, a.y, a.z, nn.x, nn.y, nn.z);
					known = cls == 1 ? RT_PIX_TAPS_LIT : (cls == 2 ? RT_PIX_TAPS_DARK : 0);
					/* audited in production: one in audit_taps (a power of two) of the classified pixels, picked by a hash of the pixel and the seed */
					if (known && L.audit_taps != 0u &&
					    ((uint32_t) (path_seed(L.seed, (uint32_t) (lr * L.width + i), 0x7a9u) >> 24) & (L.audit_taps - 1u)) == 0u) known |= RT_PIX_TAPS_AUDIT;
				}
			} else {
				const V3 sky = sky_lookup<FAST>(L, dn);                                  /* main.c:170 */
				const V3 c = mk3(clamp01(sky.x), clamp01(sky.y), clamp01(sky.z));
				V3 acc = mk3(0, 0, 0);
				if (L.sum_onto) { const float *b = L.sum_onto + ((size_t) lr * L.width + i) * 3; acc = mk3(b[0], b[1], b[2]); }
				for (int s = 0; s < L.spp; s++) acc = add3(acc, c);
				const V3 res = scale3(acc, inv_spp);
				float *dst = L.frame + ((size_t) lr * L.width + i) * 3;
				dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
			}
		}
		else if (i < L.width && lr < L.local_rows) {    /* padding row of a strip (multi_gpu.py): inside the buffer, outside the frame */
			float *dst = L.frame + ((size_t) lr * L.width + i) * 3;
			dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f;
		}
		const unsigned long long om = __ballot(obj >= 0);
		if (om) {                                       /* sky-only blocks leave nothing behind */
			unsigned int base = 0;
			if (lane == 0) base = atomicAdd(L.pix_count + shard * 32u, (unsigned int) __popcll(om));
			base = (unsigned int) __builtin_amdgcn_readfirstlane((int) base);
			if (obj >= 0) {
				/* the record as three 16-byte words, 48 bytes apart from its neighbours': a wave's stores fill whole cache lines, and
				 * the lane that takes the pixel later reads three words of one or two lines instead of twelve words of twelve */
				float4 *dst = reinterpret_cast<float4*>(L.pix) + 3 * ((size_t) shard * L.pix_shard_cap + base + (unsigned int) lanes_below(om));
				dst[0] = make_float4(a.x, a.y, a.z, nn.x);
				dst[1] = make_float4(nn.y, nn.z, __int_as_float(obj | known), pd.x);
				dst[2] = make_float4(pd.y, pd.z, __uint_as_float((uint32_t) ((j * L.pix_scale) * L.pix_width + i * L.pix_scale)) /* main.c:286 order */,
				                     __int_as_float(lr * L.width + i));
			}
		}
	}
	/* the blocks this wave finished, on the line of the list it appends to: the trace kernel's last wave adds the lines up and
	 * the host compares the sum with the number of blocks the frame has (RT_CTL_PRIMARY) */
	if (lane == 0 && blocks_done) atomicAdd(L.pix_count + shard * 32u + 1u, blocks_done);
}
#endif /* RT_SPEC_ONLY */

/* =============================================================================================
 * rt_trace_wavefront: the tuned kernel.  Same arithmetic as rt_trace_simple, different schedule.
 *
 * Profile of the simple kernel on C1 (profiles/r01): VALU-issue bound with only ~34 % of lanes
 * active -- paths of one wave end at different bounces and half of the soft-shadow taps are skipped
 * per lane (main.c:194), so most trace_ray() instructions run on half-empty waves.  Here every
 * wave is a small wavefront path tracer of its own:
 *
 *   - persistent waves pull object pixels, a few at a time, from the lists rt_primary_pass filled (global
 *     counters); inside a wave every free lane takes the next SAMPLE of a pixel the wave is working on
 *     (ballot + mbcnt prefix), so a lane never waits for its neighbours and a pixel's samples run on several
 *     lanes at once (a 1080p frame has about as many object pixels as the chip has lanes);
 *   - a path lives in its lane's registers, and so does its next bounce ray: every lane traces its own, once per
 *     round.  The up to three shadow taps of a bounce (known right after shading: they only feed the light term that is
 *     added afterwards, main.c:257-261) are compacted into a per-wave LDS queue with ballot/mbcnt prefix sums and
 *     traced in full batches of 64 by whichever lane gets them -- unless their answer is known without tracing
 *     (rt_lit.h: the camera ray's hit point was flagged by rt_primary_pass, or the hit point of a later bounce lies in a
 *     flagged cell of the scene's table): 69 % of the taps of the shipped scene_0;
 *   - the radiance arithmetic of a bounce (emission, albedo, light term: main.c:232,248,257-261) needs
 *     the bounce's tap results but nothing else does, so it runs two rounds behind the ray generation
 *     (the "back" and the "front" of a lane): taps that do not fill a batch simply wait in the queue, up to two
 *     rounds, and tap batches are full;
 *   - the primary hit is traced once per pixel, by rt_primary_pass, and re-used by all samples (the
 *     reference has no sub-pixel jitter, main.c:293-296, so bounce 0 of every sample is the same ray):
 *     bit-exact; sky-only pixels never reach this kernel;
 *   - the samples of a pixel are ADDED IN SAMPLE ORDER (main.c:394) although they finish in any order on
 *     any lane: a finished sample's clamped colour goes into a slot of a per-wave LDS window (its slot was
 *     reserved, in sample order, when the sample was handed out), and at the end of every round one lane per
 *     pixel stream adds the slots that have become contiguous, resolves a pixel whose last sample went in
 *     (main.c:476) and writes it -- 12 bytes per pixel -- to the frame.  No sample ever leaves the chip.
 * Results are bit-identical to rt_trace_simple and to the CPU oracle.
 * ============================================================================================= */

#define WF_SUM_EVERY 2u                /* rounds between two passes of the in-order sum (section 6) when a pixel has >= 32 samples */
#define WF_SHARDS  64                  /* pixel lists (at most); each has a fill counter and a dequeue counter, 128 B apart */
#define RT_COUNTER_BYTES ((2 * WF_SHARDS + 1) * 128)   /* + one line of launch control words (rt_launch.control) */
#define WF_QUEUE   128                 /* tap ring: at most 63 waiting + 64 pushed at a time */
#define WF_STREAMS 8                   /* pixels a wave adds up concurrently (at most) */
#define WF_WINDOW  384                 /* sample slots per wave, shared equally by its streams */
#define WF_CLAIM   8                   /* pixels a wave claims from its list at a time ... */
/* ... until the list is down to this many such claims for every wave that shares it; from then on a wave claims what its streams ask for:
 * at the end of a launch no wave may sit on a reserve while others have run dry, and a pixel of 256 samples is 0.4 ms of a wave --
 * measured against "always what is asked for" (profiles/r05/ab_claim_generations.txt): 3 claims: C1 -2.6 %, C2 +6 %; 6: -2.8 / +1 %;
 * 12: -2.1 / +0.2 %, C4 strip -0.3 % -- hence more of them the more samples a pixel has */
#define WF_CLAIM_GENERATIONS(spp) (3u + (unsigned int) (spp) / 32u)
/* ... and only for pixels of fewer samples than this: a wave looks for rt_cancel() when it CLAIMS (the hand-outs from its reserve read
 * nothing), so a reserve of eight pixels of 256 ... 1024 samples is 0.4 ... 1.6 ms more before a cancelled launch lets go -- and buys
 * nothing there: claims of eight were worth -2 ... 3 % at 64 samples per pixel (neighbouring pixel writes, an eighth of the atomics) and
 * +0.4 ... 0.6 % at 256 and 1024, where a pixel is written once per hundreds of rounds (profiles/r05/ab_claim_generations.txt) */
#define WF_CLAIM_BELOW_SPP 128u
#define WF_EMPTY   0xffffffffu         /* window slot not written yet (a colour channel is in [0,1]: never this pattern) */
#define WF_LAST    0x8000              /* slot word: this sample is the last one of its pixel */
#define REC_VALID    1                /* per-bounce record handed from the front to the back (wavefront_body) */
#define REC_SPECULAR 2
#define REC_LAST     4                /* the path ends after this bounce ...               */
#define REC_SKY      8                /* ... because its bounce ray left the scene (sky1 / sky2) */
#define REC_TAPS_LIT 128              /* the bounce's accepted taps are known to hit the emitter (rt_lit.h): none was queued */
#define REC_AUDIT     0x80000         /* (rt_launch.audit_taps) the bounce's taps are known AND were queued: the back compares the two */
#define REC_TAPS_DARK 0x40000         /* ... known NOT to have the emitter as their nearest hit: they add nothing (above the object index, < 1024 << 8) */

struct WaveLDS {
	float q[6][WF_QUEUE];              /* tap queue SoA: hit point xyz, random_vector() xyz (kinds 2..4 = tap 0..2 of lane `owner`) */
	unsigned short qmeta[WF_QUEUE];    /* owner lane | kind << 8 | round mod 3 << 12                 */
	short tap[3][3][64];               /* shadow tap results per owner lane (object index < 1024 or -1), by the round (mod 3) that queued them */
	/* Sample window.  Stream g (one pixel at a time, pixels one after another) owns slots [g*WF_WINDOW/P, (g+1)*WF_WINDOW/P)
	 * as a ring indexed by the sequence number of the sample within the stream (modulo its 48 slots).  win[0] holds the red channel
	 * with bit 31 = "last sample of its pixel" (red is >= +0, and adding +0 for a -0 changes no sum), or WF_EMPTY;
	 * the slot after a pixel's last sample holds the pixel's frame offset instead of a colour. */
	float win[3][WF_WINDOW];
	float rec[12][WF_STREAMS];         /* the stream's current pixel: primary hit xyz, normal xyz, object, camera ray xyz, RNG pixel index, frame offset */
	unsigned int s_nxt[WF_STREAMS];    /* next sample of the current pixel to hand out (== spp: the stream needs a pixel) */
	unsigned int s_seq[WF_STREAMS];    /* slots reserved so far (sequence number of the next one) */
	unsigned int s_drained[WF_STREAMS];/* slots added up and released so far */
	float s_sum[3][WF_STREAMS];        /* running sum of the pixel being added up */
	/* what the wave reports when it leaves (RT_CTL_*): object pixels it wrote; audited taps and those whose trace contradicts rt_lit.h */
	unsigned int n_written, n_audited, n_disagree;
	/* ... added up per workgroup in the first wave's copy of this struct (g_left: waves of the workgroup that have left): 4 096
	 * waves reporting to one address each would queue up for 50 us at the end of every launch (an address takes ~88 atomics per
	 * microsecond); the last wave of a workgroup reports for all of them */
	unsigned int g_written, g_audited, g_disagree, g_left;
	/* the wave's reserve: list entries [res_next, res_end) (absolute record numbers) it has claimed and not yet given to a stream;
	 * res_seen: entries its list had left after that claim */
	unsigned int res_next, res_end, res_seen, pad_[2];
};

/* The launch record as the code that takes a new pixel block reads it.  Kernel arguments are invariant, so the
 * compiler loads every field once at kernel entry and keeps it in a scalar register for the whole kernel --
 * far more than the 102 there are, and the spill/reload traffic lands in the per-round code.  The fields only
 * a block hand-out needs are read through a pointer whose origin is hidden, i.e. from memory when used. */
typedef const __attribute__((address_space(1))) rt_launch *rt_launch_cold;
RT_DEV rt_launch_cold cold_view()
{
	/* the launch record is the first kernel argument of every trace kernel: offset 0 of the kernarg segment */
	unsigned long long a = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(a));
	return (rt_launch_cold) a;
}

/* four workgroups (16 waves) per CU need 4 x (96 B x objects + 4 x sizeof(WaveLDS)) <= 160 KiB: keep room for 17
 * objects -- one workgroup less per CU costs 15 %.  (The sample window is what a wave's LDS is spent on: half the
 * slots cost 58 % on C1, 48 instead of 32 per stream buy 2 % there and 9 % at 1024 samples per pixel.) */
static_assert(4 * sizeof(WaveLDS) + 96 * 17 <= 160 * 1024 / 4, "WaveLDS grew: scenes of up to 17 objects no longer fit four workgroups per CU");
static_assert(sizeof(WaveLDS) % 16 == 0 && sizeof(CullWave) % 16 == 0 && sizeof(CullWave) == 1024, "the waves' LDS records follow each other: the culled trace reads and writes a ray's cluster mask as one 16-byte word");


template <bool FAST> RT_DEV V3 unit3_sel(V3 v) { return FAST ? unit3_fast(v) : unit3(v); }

/* the value lane - 2 of the same 16-lane row holds (the row's first two lanes: zero); every lane must be active */
RT_DEV float from_lane_two_below(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112 /* row_shr:2 */, 0xf, 0xf, true));
}

/* What a path needs to know about its pixel; filled by the lane that hands the pixel out. */
struct PixelRec { V3 a, n; int obj; V3 dir; uint32_t index; int off; };

/* record c of the pixel lists rt_primary_pass filled */
RT_DEV PixelRec load_pixel(rt_launch_cold C, size_t c)
{
	/* (the pointer comes out of the launch record in memory: said to be a global one, the three loads are global loads off
	 * one base instead of flat loads with a 64-bit address each) */
	typedef float word4 __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(1))) word4 *gword4;
	const gword4 src = (gword4) C->pix + 3 * c;
	const word4 w0 = src[0], w1 = src[1], w2 = src[2];
	PixelRec p;
	p.a = mk3(w0.x, w0.y, w0.z);
	p.n = mk3(w0.w, w1.x, w1.y);
	p.obj = __float_as_int(w1.z);
	p.dir = mk3(w1.w, w2.x, w2.y);
	p.index = __float_as_uint(w2.z);
	p.off = __float_as_int(w2.w);
	return p;
}

/* The dequeue counters (the trace kernels' second argument) are read from the kernarg segment where they are used -- a pixel
 * fetch, the end of the launch -- instead of living in two scalar registers through every round: the compiled kernel runs at its
 * register limit, and a value that the cold ends of the kernel keep alive costs the rounds a spill. */
typedef __attribute__((address_space(1))) unsigned int *gcounters;
RT_DEV gcounters counters_of(rt_launch_cold C)
{
	typedef const __attribute__((address_space(1))) unsigned long long *gaddr;
	return (gcounters) *(gaddr) ((const __attribute__((address_space(1))) char *) C + ((sizeof(rt_launch) + 7) & ~(size_t) 7));
}

/* A wave leaves the launch: it adds what it wrote to the launch's control words, and the LAST wave to leave adds up the
 * pixel lists -- listed by the camera-ray pass, fetched by the waves -- and stamps the launch with its number.  The host
 * reads the words behind the launch and delivers the frame only if the stamp is there and the sums agree (rt_api.cpp
 * judge_launch): every wave left, every listed pixel was fetched, every fetched pixel was written.  The reference publishes a
 * column whole or not at all (main.c:377-396). */
template <int BLOCK>
RT_DEV void leave_launch(WaveLDS *Wp, int wave)
{
	WaveLDS &Wv = *Wp;
	typedef __attribute__((address_space(1))) unsigned int *gwuint;
	typedef __attribute__((address_space(1))) unsigned long long *gwulong;
	/* (the lane number is formed here, not handed in: whatever this cold end of the kernel derives from a value that lives through
	 * the rounds is kept in a register through the rounds -- and the compiled kernel runs at its register limit) */
	const int lane = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	const rt_launch_cold C = cold_view();
	wave_fence();
	/* The pixels this wave wrote.  In direct mode (one sample per pixel) the lanes counted them; otherwise nothing was counted in the
	 * rounds: a stream drains spp slots per pixel, one more for the frame offset that releases the write (section 6), and one more
	 * for the sums held so far when the launch adds onto them -- so a stream's drained-slot count says how many pixels it completed. */
	unsigned int wrote = Wv.n_written;
	if (C->spp != 1) {
		const unsigned int per_pixel = (unsigned int) C->spp + 1u + (C->sum_onto != nullptr ? 1u : 0u);
		unsigned int mine = lane < WF_STREAMS ? Wv.s_drained[lane & (WF_STREAMS - 1)] / per_pixel : 0u;
#pragma unroll
		for (int m = 1; m < WF_STREAMS; m <<= 1) mine += from_lane(mine, lane ^ m);
		wrote = mine;
	}
	/* the workgroup's waves add up in its first wave's LDS; the last of them to leave reports for the workgroup: 4 096 waves
	 * reporting to one address each would queue up for 50 us at the end of every launch (an address takes ~88 atomics per us) */
	WaveLDS &G = Wp[-wave];
	unsigned int before = 0u;
	if (lane == 0) {
		if (wrote) __hip_atomic_fetch_add(&G.g_written, wrote, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (Wv.n_audited) __hip_atomic_fetch_add(&G.g_audited, Wv.n_audited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (Wv.n_disagree) __hip_atomic_fetch_add(&G.g_disagree, Wv.n_disagree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		before = __hip_atomic_fetch_add(&G.g_left, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	before = (unsigned int) __builtin_amdgcn_readfirstlane((int) before);
	if (before + 1u != (unsigned int) (BLOCK / 64)) return;
	/* ... on the dequeue line of the workgroup's first list (word 1: pixels written, word 2: workgroups that have left): 64 lines share
	 * the launch's ~1 000 reports; the workgroup that completes a line reports the line, and the one that completes the last line is
	 * the launch's last */
	const gcounters block_counter = counters_of(C);
	const gwuint ctl = (gwuint) C->control;
	const unsigned int lists = (unsigned int) C->num_shards, groups = (unsigned int) C->trace_workgroups;
	const unsigned int line = blockIdx.x % lists;
	const unsigned int on_line = (groups - line + lists - 1u) / lists;             /* workgroups w of the grid with w % lists == line */
	bool last = false;
	if (lane == 0) {
		const unsigned int nw = G.g_written, na = G.g_audited, nd = G.g_disagree;
		const gwuint mine = (gwuint) block_counter + line * 32u;
		if (nw) __hip_atomic_fetch_add(mine + 1, nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (na) __hip_atomic_fetch_add((gwulong) (mine + 4), (unsigned long long) na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* (words 4-5, 6-7: audited taps, and those that disagree) */
		if (nd) __hip_atomic_fetch_add((gwulong) (mine + 6), (unsigned long long) nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (__hip_atomic_fetch_add(mine + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == on_line)
			last = __hip_atomic_fetch_add(ctl + RT_CTL_LINES_DONE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == (groups < lists ? groups : lists);
	}
	if (__ballot(last) == 0ull) return;
	/* the launch's last workgroup: lane s adds up list s */
	unsigned int listed = 0u, fetched = 0u, blocks = 0u, written = 0u, left = 0u;
	unsigned long long audited = 0ull, disagree = 0ull;
	if (lane < C->num_shards) {
		const gwuint fill = (gwuint) C->pix_count + (unsigned int) lane * 32u, taken_at = (gwuint) block_counter + (unsigned int) lane * 32u;
		listed = __hip_atomic_load(fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		blocks = __hip_atomic_load(fill + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		/* (the dequeue counter counts what was ASKED for, so it passes the fill count at the end of every list: a list whose
		 * counter stayed below it was not dealt out to the end) */
		const unsigned int taken = __hip_atomic_load(taken_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		fetched = taken < listed ? taken : listed;
		written = __hip_atomic_load(taken_at + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		left = __hip_atomic_load(taken_at + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		audited = __hip_atomic_load((gwulong) (taken_at + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		disagree = __hip_atomic_load((gwulong) (taken_at + 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		/* ... and leaves the line as a launch must find it: the scratch set's next launch -- which starts when this one and whoever
		 * reads its control words are done -- has nothing to clear when it keeps the lists (an interactive pass that differs from
		 * the set's last in its sample number only, rt_api.cpp: two memsets per pass were a fifth of it) */
		__hip_atomic_store(taken_at, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(taken_at + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(taken_at + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store((gwulong) (taken_at + 4), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store((gwulong) (taken_at + 6), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) {
		listed += from_lane(listed, lane ^ m); fetched += from_lane(fetched, lane ^ m);
		blocks += from_lane(blocks, lane ^ m); written += from_lane(written, lane ^ m); left += from_lane(left, lane ^ m);
		audited += (unsigned long long) from_lane((uint32_t) audited, lane ^ m) | ((unsigned long long) from_lane((uint32_t) (audited >> 32), lane ^ m) << 32);
		disagree += (unsigned long long) from_lane((uint32_t) disagree, lane ^ m) | ((unsigned long long) from_lane((uint32_t) (disagree >> 32), lane ^ m) << 32);
	}
	if (lane == 0) {
		/* every word a reader looks at is WRITTEN here, none is counted up in place: the line needs no clearing either.  "Cut short by
		 * rt_cancel()" is the relay word, which a wave sets when it gives up; it and the count of finished lines go back to zero. */
		ctl[RT_CTL_CANCELLED] = __hip_atomic_load(ctl + RT_CTL_STOP_RELAY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
		__hip_atomic_store(ctl + RT_CTL_STOP_RELAY, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(ctl + RT_CTL_LINES_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		ctl[RT_CTL_AUDITED] = (unsigned int) audited; ctl[RT_CTL_AUDITED + 1] = (unsigned int) (audited >> 32);
		ctl[RT_CTL_DISAGREE] = (unsigned int) disagree; ctl[RT_CTL_DISAGREE + 1] = (unsigned int) (disagree >> 32);
		ctl[RT_CTL_LISTED] = listed; ctl[RT_CTL_FETCHED] = fetched; ctl[RT_CTL_PRIMARY] = blocks; ctl[RT_CTL_WRITTEN] = written;
		ctl[RT_CTL_WAVES_LEFT] = left * (unsigned int) (BLOCK / 64);
		__hip_atomic_store(ctl + RT_CTL_STAMP, C->launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}


/* BLOCK: threads per workgroup (the culled variant also comes with RT_BLOCK_WIDE); AUDIT: the variant that compares the answers
 * of rt_lit.h it is asked to audit with a trace of the taps (rt_tuning.audit_known_taps) */
template <bool FAST, bool CULL = false, int BLOCK = RT_BLOCK, bool AUDIT = false>
RT_DEV void wavefront_body(const rt_launch &L, unsigned int *)
{
	extern __shared__ float4 lds[];
#ifdef RT_SPEC_HEADER
	constexpr int n = SPEC_N;              /* the compiled scene's object count: every LDS offset below is a literal */
#else
	const int n = L.num_objects;
#endif
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	WaveLDS &W = reinterpret_cast<WaveLDS*>(lds + (CULL ? RT_CULL_F4(L.num_clusters) : (L.lit_grids_in_lds ? 9 : 6) * n))[wave];
	if (threadIdx.x == 0) { W.g_written = 0u; W.g_audited = 0u; W.g_disagree = 0u; W.g_left = 0u; }     /* (ordered before any wave's report at the end by the staging barrier) */
	const SceneLDS sc = CULL ? stage_geometry(L) : stage_scene(L, lds, n);
	/* the grids of the "taps certainly lit" table (rt_lit.h), when the launcher found room for them: 3 x float4 per object */
	if (L.lit_grids_in_lds) {
		const float4 *gsrc = reinterpret_cast<const float4*>(L.lit_grids);
		for (int i = threadIdx.x; i < 3 * n; i += BLOCK) lds[6 * n + i] = gsrc[i];
		__syncthreads();
	}
	/* (two pointers, not one chosen at run time: a pointer that may be LDS or memory makes every access a flat load with a
	 * 64-bit address -- ten per bounce ray) */
	const rt_lit_grid *lit_grids_lds = reinterpret_cast<const rt_lit_grid*>(lds + 6 * n);
	const rt_lit_grid *lit_grids_mem = reinterpret_cast<const rt_lit_grid*>(L.lit_grids);
	const bool grids_in_lds = L.lit_grids_in_lds != 0;
	/* large scenes: the clusters of rt_cull.h behind the scene records (such scenes have no lit-taps table: it needs <= 64 objects) */
	ClusterLDS cl; cl.recs = nullptr; cl.mem = nullptr; cl.count = 0; cl.margin = 0.0f; cl.origin_max = 0.0f;
	if (CULL) cl = stage_clusters(L, lds);
	CullWave *cull_wave = reinterpret_cast<CullWave*>(reinterpret_cast<WaveLDS*>(lds + RT_CULL_F4(L.num_clusters)) + BLOCK / 64) + wave;   /* (CULL) */

	/* rt_launch.sum_onto (several interactive passes in one launch): a pixel's samples are added to what the frame holds so far.
	 * That value takes the window slot BEFORE the pixel's first sample, as if it were a sample: 0 + value is the value (it is
	 * >= +0), and section 6 then adds the samples onto it in order; the sum is written as it is. */
	const bool onto = L.sum_onto != nullptr;
	const float inv_spp = onto ? 1.0f : 1.0f / (float) L.spp;
	const unsigned int spp = (unsigned int) L.spp;
	/* spp == 1 (progressive passes): a pixel is its one sample, nothing has to be ordered -- every lane takes a
	 * pixel for itself and writes it when the path retires.  Otherwise the wave runs P pixel streams. */
	const bool direct = L.spp == 1;
	/* Eight streams of eight lanes.  The lanes of a stream are every second lane of one 16-lane DPP row (streams 2r
	 * and 2r+1 share row r), so that "the value of the stream's previous lane" is one DPP row_shr:2 -- which hands
	 * the stream's first lane a zero -- in the in-order sum of section 6. */
	constexpr int P = WF_STREAMS, G = 64 / P;
	constexpr unsigned int wn = WF_WINDOW / P;              /* slots per stream */
	static_assert(P == 8 && G == 8, "the lane layout below is written for 8 streams of 8 lanes");
	const int g = 2 * (lane >> 4) + (lane & 1);             /* this lane's home stream */
	const unsigned int j = (unsigned int) (lane & 15) >> 1; /* its place among the stream's lanes */
	const bool leader = j == 0u;                            /* does the bookkeeping of a stream */
	const int gshift = (lane & 48) + (lane & 1);            /* position of the stream's first lane */
	const unsigned long long gmask = 0x5555ull << gshift;
#ifdef RT_SPEC_HEADER
	constexpr bool only_light = FAST && SPEC_ONLY_LIGHT_EMITS != 0 && SPEC_LIGHT >= 0;
	const bool have_light = SPEC_LIGHT >= 0;               /* the compiled scene's emitter: literals, no scalar registers */
	const V3 light_pos = mk3(SPEC_LIGHT_POS[0], SPEC_LIGHT_POS[1], SPEC_LIGHT_POS[2]);
	const int light_obj = SPEC_LIGHT;
#else
	const bool only_light = FAST && L.only_light_emits != 0 && L.light_index >= 0;
	const V3 light_pos = ld3(L.light_pos);
	const bool have_light = L.light_index >= 0;
	const int light_obj = L.light_index;
#endif

	for (int k = lane; k < WF_WINDOW; k += 64) W.win[0][k] = __uint_as_float(WF_EMPTY);
	if (lane < WF_STREAMS) {
		W.s_nxt[lane] = spp; W.s_seq[lane] = 0u; W.s_drained[lane] = 0u;
		W.s_sum[0][lane] = 0.0f; W.s_sum[1][lane] = 0.0f; W.s_sum[2][lane] = 0.0f;
	}
	if (lane == 0) { W.n_written = 0u; W.n_audited = 0u; W.n_disagree = 0u; W.res_next = 0u; W.res_end = 0u; W.res_seen = ~0u; }
	wave_fence();

	/* wave-uniform pixel supply: object pixels are dealt from the lists rt_primary_pass filled, each with its own
	 * dequeue counter on its own 128-byte line (one counter saturates at ~88 dequeues/us, MI355X_MICROARCH.md).  A
	 * wave starts at the list of its workgroup and moves on to the next one when a list has run out. */
	unsigned int shard = blockIdx.x % (unsigned int) L.num_shards;
	bool exhausted = false;                 /* no pixels left to fetch */
	bool cancelled = false;                 /* rt_cancel(): nothing more is handed out, the paths in flight finish, the wave leaves */

	/* per-lane path state.  A path is worked on at two places two rounds apart.  The FRONT (section 2) turns the
	 * pending hit into the next rays: it owns rng, bounce, the hit and f_slot.  The BACK (section 5) does the
	 * radiance arithmetic of a bounce (main.c:232,248,257-261) two rounds later, when its shadow taps are certainly
	 * traced: it owns rad and carry.  rec1 / rec2 are the records of the bounces the front shaded one and two rounds
	 * ago, with their slot words and sky texels.  A slot word is the window slot the sample's colour goes to
	 * (| WF_LAST); in direct mode, the pixel's frame offset. */
	bool  f_live = false;                   /* the front is on a sample */
	int   f_slot = 0, slot1 = 0, slot2 = 0;
	int   bounce = 0;
	bool  has_hit = false;
	V3    carry = mk3(1, 1, 1), rad = mk3(0, 0, 0);
	V3    hp = mk3(0, 0, 0), hn = mk3(0, 0, 0), hdir = mk3(0, 0, 0);
	uint32_t sky1 = 0, sky2 = 0;            /* sky texel that ends the sample of rec1 / rec2 (REC_SKY): fetched when the bounce ray
	                                         * is found to have left the scene, converted two rounds later when it is used */
	int   hobj = -1;
	uint32_t lit_next = 0;                  /* != 0: the taps from the pending hit point certainly reach the emitter, none is traced (rt_lit.h) */
	int   rec1 = 0, rec2 = 0;               /* REC_* | tapmask << 4 | object << 8 of the bounces shaded one and two rounds ago */
	uint64_t rng = 0;
	/* the tap queue persists across rounds: taps that do not fill a batch wait, at most two rounds */
	unsigned int q_head = 0, q_tail = 0, phase = 0;     /* phase = round number mod 3 */
	unsigned int sum_tick = 0;
	const unsigned int sum_every = L.spp >= 32 ? WF_SUM_EVERY : 1u;

	/* Section 6 of a round: adding the finished samples in sample order (main.c:394).  Lane j of a stream looks at
	 * the j-th slot after the stream's last added one.  The slots that are filled without a gap from the first form
	 * the run (it ends behind the first pixel that is completed in it); their colours are added to the stream's
	 * running sum one after the other -- lane j's partial sum is lane j-1's plus its own colour, handed on with DPP
	 * row shifts, all eight streams in lockstep -- and a pixel whose last sample is in the run is resolved
	 * (main.c:476) and written to the frame.  Every lane of the wave must be active. */
	auto add_finished_samples = [&]() {
		bool again;
		do {
			STAT(23);
			const unsigned int d = W.s_drained[g], seq = W.s_seq[g];
			const unsigned int slot = d + j;
			const unsigned int e = (unsigned int) g * wn + (slot % wn);
			uint32_t xb = WF_EMPTY;
			if ((int) (seq - slot) > 0) xb = __float_as_uint(W.win[0][e]);
			const bool filled = xb != WF_EMPTY;
			const unsigned long long fm = __ballot(filled);
			const unsigned long long lm = __ballot(filled && (xb >> 31) != 0u);      /* last samples of their pixels */
			const uint32_t fbits = (uint32_t) (fm >> gshift) & 0x5555u, lbits = (uint32_t) (lm >> gshift) & 0x5555u;
			int k = (int) __builtin_ctz((~fbits & 0x5555u) | 0x10000u) >> 1;        /* length of the run, 0..8 */
			/* the run ends with the first pixel that is completed in it: its last sample and the frame-offset slot
			 * behind it (filled since the sample was handed out); a last sample in the stream's last lane waits */
			const int lastpos = (int) __builtin_ctz(lbits | 0x10000u) >> 1;          /* 8: none */
			const bool boundary = lastpos < k;
			if (boundary) k = lastpos + 2 <= G ? lastpos + 2 : lastpos;
			const bool active = (int) j < k;
			const bool offset_slot = boundary && (int) j == lastpos + 1 && active;
			V3 c = mk3(0, 0, 0);
			if (active && !offset_slot) c = mk3(__uint_as_float(xb & 0x7fffffffu), W.win[1][e], W.win[2][e]);
			if (leader) c = add3(mk3(W.s_sum[0][g], W.s_sum[1][g], W.s_sum[2][g]), c);   /* the running sum enters at the first lane */
			V3 sum = c;
#pragma unroll
			for (int t = 1; t < G; t++)             /* after step t lanes j <= t hold their final partial sums */
				sum = mk3(from_lane_two_below(sum.x) + c.x, from_lane_two_below(sum.y) + c.y, from_lane_two_below(sum.z) + c.z);
			if (offset_slot) {                  /* the pixel is complete: resolve and write it (main.c:476); this lane's own colour was zero */
				const V3 res = scale3(sum, inv_spp);
				float *dst = L.frame + (size_t) xb * 3;
				dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;      /* (counted when the wave leaves: a pixel is spp + 1 drained slots of its stream) */
			}
			if (active) W.win[0][e] = __uint_as_float(WF_EMPTY);
			if (active && (int) j == k - 1) {
				W.s_drained[g] = d + (unsigned int) k;
				W.s_sum[0][g] = offset_slot ? 0.0f : sum.x; W.s_sum[1][g] = offset_slot ? 0.0f : sum.y; W.s_sum[2][g] = offset_slot ? 0.0f : sum.z;
			}
			/* more may be waiting behind the run; it can wait for the next round unless the window is filling up */
			again = __ballot(k > 0 && (k == G || boundary) && seq - (d + (unsigned int) k) > 3u * wn / 4u) != 0ull;
			wave_fence();
		} while (again);
	};

	/* `taps` taps whose answer rt_lit.h gave were traced all the same; `wrong` of them contradict it */
	auto audit_taps = [&](int taps, int wrong) {
		__hip_atomic_fetch_add(&W.n_audited, (unsigned int) taps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		if (wrong) __hip_atomic_fetch_add(&W.n_disagree, (unsigned int) wrong, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	};

	STAMP_LOCAL;
	for (;; phase = phase == 2u ? 0u : phase + 1u) {
		STAMP(7);
		STAMP_ROUND;
		/* ---- 1. sample supply ---------------------------------------------------------------
		 * Lanes whose front is free take new samples.  A lane asks its home stream first and, in the following
		 * attempts, the streams next to it (stream (g + attempt) mod P is asked by exactly one group of lanes per
		 * attempt, so the group's leader lane does that stream's bookkeeping).  A stream hands out the samples of
		 * its pixel in order, each with the next slot of the stream's window; when the pixel has none left the
		 * stream takes the next object pixel of the wave's work item. */
#pragma unroll 1
		for (int attempt = 0; attempt < P; attempt++) {
			const bool want = !f_live;
			const unsigned long long wmask = __ballot(want);
			if (wmask == 0ull || cancelled) break;
			/* the lists have run out: go on only while some stream still has samples (and slots) to give */
			if (exhausted && (direct || (attempt > 0 &&
			    __ballot(leader && W.s_nxt[g] < spp && W.s_seq[g] - W.s_drained[g] < wn - 1u) == 0ull))) break;
			const int sg = (g + attempt) & (P - 1);
			const unsigned long long gm = wmask & gmask;              /* wanting lanes of my group */
			STAT(20);
			unsigned int nxt = 0;
			bool need_pixel = want;
			if (!direct) {
				nxt = W.s_nxt[sg];
				need_pixel = leader && gm != 0ull && nxt >= spp;
				if (onto) need_pixel = need_pixel && W.s_seq[sg] - W.s_drained[sg] < wn - 1u;      /* (the slot for what the frame holds so far) */
			}
			const unsigned long long nmask = __ballot(need_pixel);
			STAMP(0);
			if (nmask != 0ull) {
				STAT(21);
				const rt_launch_cold C = cold_view();
				const gcounters block_counter = counters_of(C);
				const int asked = __popcll(nmask);
				/* Pixels are CLAIMED from the list WF_CLAIM at a time and handed to the streams from the wave's reserve: consecutive
				 * entries are neighbours in a row of an 8x8 block (rt_primary_pass lists a block row by row), so a wave's streams work on
				 * a run of a frame row and their 12-byte pixel writes land in the same 32-byte sectors, from the same compute unit,
				 * within a few rounds of each other -- L2 merges them -- instead of one pixel per wave all over the chip (six times the
				 * frame's bytes written).  Also an eighth of the atomics. */
				unsigned int res_next = (unsigned int) __builtin_amdgcn_readfirstlane((int) W.res_next), res_end = (unsigned int) __builtin_amdgcn_readfirstlane((int) W.res_end);
				if (res_next == res_end && !exhausted) {
					typedef const __attribute__((address_space(1))) unsigned int *guint;
					typedef __attribute__((address_space(1))) unsigned int *gwuint;
					/* rt_cancel() (main.c:316-317: the frame has been invalidated).  The host stores its request in a word of host
					 * memory: "the launches up to this number are to stop".  Reading that is slow (a read over the link takes the
					 * place of 75 ns of everybody else's: 4 096 waves asking at once cost a strip a third of its time), so only eight
					 * waves of the launch do, when they fetch pixels, and pass the news on through control[2] in device memory, which
					 * every other wave reads when it fetches pixels.  Told to stop, a wave hands out nothing more, lets the paths in
					 * flight finish and leaves; the launch is marked incomplete.  (A wave that has no pixels left to fetch does not
					 * ask any more: it is about to leave anyway.) */
					const unsigned int seen = (unsigned int) __builtin_amdgcn_readfirstlane((int) W.res_seen);
					const unsigned int until = (unsigned int) C->trace_workgroups * (unsigned int) (BLOCK / 64) / (unsigned int) C->num_shards * (unsigned int) WF_CLAIM * WF_CLAIM_GENERATIONS(spp);
					const unsigned int want = (WF_CLAIM > 0 && !direct && spp < WF_CLAIM_BELOW_SPP && seen > until && asked < WF_CLAIM) ? (unsigned int) WF_CLAIM : (unsigned int) asked;
					unsigned int word = 0u, k = 0u;
					if (lane == 0) {
						word = (blockIdx.x < 8u && wave == 0) ? __hip_atomic_load((guint) C->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - C->launch_id
						                                      : __hip_atomic_load((guint) C->control + RT_CTL_STOP_RELAY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u;
						k = __hip_atomic_fetch_add(block_counter + shard * 32u, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					k = (unsigned int) __builtin_amdgcn_readfirstlane((int) k);
					if ((int) __builtin_amdgcn_readfirstlane((int) word) >= 0) {      /* the request covers this launch / another wave has seen it */
						cancelled = true;
						if (lane == 0) __hip_atomic_store((gwuint) C->control + RT_CTL_STOP_RELAY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     /* (becomes control[RT_CTL_CANCELLED] when the launch's last workgroup leaves) */
					}
					const guint fill_counts = (guint) C->pix_count;
					const unsigned int drop = C->test_drop_pixels;          /* (testing aid: 0 in production) */
					unsigned int filled = (unsigned int) __builtin_amdgcn_readfirstlane((int) fill_counts[shard * 32u]);
					filled -= filled < drop ? filled : drop;
					const unsigned int take = k < filled && !cancelled ? (filled - k < want ? filled - k : want) : 0u;
					res_next = shard * (unsigned int) C->pix_shard_cap + k;
					res_end = res_next + take;
					if (lane == 0) W.res_seen = filled > k + want ? filled - (k + want) : 0u;
					if (take < want) {
						/* This list has run out: look at all of them at once (lane s reads list s's two counters; a stale
						 * dequeue count can only show more pixels left than there are, never fewer) and move to the next
						 * one that still has pixels.  None: the launch has no pixels left to hand out. */
						unsigned int left = 0, taken = 0;
						if (lane < C->num_shards) {
							taken = __hip_atomic_load(block_counter + (unsigned int) lane * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							unsigned int have = fill_counts[(unsigned int) lane * 32u];
							have -= have < drop ? have : drop;
							left = have > taken ? have - taken : 0u;
						}
						const unsigned long long some = __ballot(left != 0u);
						/* (nothing left anywhere, and nothing in the reserve: the wave has no pixels to hand out any more) */
						if ((some == 0ull && take == 0u) || cancelled) { exhausted = true; STAMP_DRY; }
						else if (some == 0ull) { }
						else {      /* the waves spread over the lists that are left (one address takes ~88 atomics per microsecond) */
							const int pick = (int) ((blockIdx.x * (BLOCK / 64) + (unsigned int) wave) % (unsigned int) __popcll(some));
							shard = (unsigned int) __builtin_ctzll(__ballot(left != 0u && lanes_below(some) == pick));
						}
					}
				}
				const int got = (int) (res_end - res_next) < asked ? (int) (res_end - res_next) : asked;
				const size_t first = (size_t) res_next;
				if (lane == 0) { W.res_next = res_next + (unsigned int) got; W.res_end = res_end; }
				const int rr = lanes_below(nmask);
				if (need_pixel && rr < got) {
					const PixelRec px = load_pixel(C, first + (size_t) rr);
					if (direct) {
						f_slot = px.off; f_live = true;
						rng = (uint64_t) px.index;                /* (pixel index, sample 0): seeded where the sample is first shaded (section 2) */
						bounce = 0;
						hp = px.a; hn = px.n; hobj = px.obj & (RT_PIX_TAPS_LIT - 1); hdir = px.dir;
						lit_next = ((uint32_t) px.obj >> 16) & 7u;
						has_hit = true;
					} else {
						W.rec[0][sg] = px.a.x;   W.rec[1][sg] = px.a.y;   W.rec[2][sg] = px.a.z;
						W.rec[3][sg] = px.n.x;   W.rec[4][sg] = px.n.y;   W.rec[5][sg] = px.n.z;
						W.rec[6][sg] = __int_as_float(px.obj);
						W.rec[7][sg] = px.dir.x; W.rec[8][sg] = px.dir.y; W.rec[9][sg] = px.dir.z;
						W.rec[10][sg] = __uint_as_float(px.index);
						W.rec[11][sg] = __int_as_float(px.off);
						W.s_nxt[sg] = 0u;
						if (onto) {
							typedef const __attribute__((address_space(1))) float *gfloat;
							const gfloat held = (gfloat) C->sum_onto + (size_t) px.off * 3;
							const unsigned int sq = W.s_seq[sg], e = (unsigned int) sg * wn + sq % wn;
							W.win[1][e] = held[1]; W.win[2][e] = held[2];
							W.win[0][e] = held[0];
							W.s_seq[sg] = sq + 1u;
						}
					}
				}
				wave_fence();
				STAMP(5);
			}
			if (!direct) {
				nxt = W.s_nxt[sg];
				const unsigned int seq = W.s_seq[sg], drained = W.s_drained[sg];
				/* samples the pixel still has, and slots free in the stream's window (one is kept back for the
				 * frame-offset slot that follows a pixel's last sample) */
				const int left = (int) spp - (int) nxt, room = (int) wn - 1 - (int) (seq - drained);
				const int avail = left < room ? left : room;
				const int r = lanes_below(gm);
				if (want && r < avail) {
					STAT(22);
					const unsigned int s = nxt + (unsigned int) r, slot = seq + (unsigned int) r;
					const bool last = s + 1u == spp;
					f_slot = (int) ((unsigned int) sg * wn + (slot % wn)) | (last ? WF_LAST : 0);
					if (last) W.win[0][(unsigned int) sg * wn + ((slot + 1u) % wn)] = W.rec[11][sg];
					rng = ((uint64_t) s << 32) | (uint64_t) __float_as_uint(W.rec[10][sg]);   /* (pixel index, sample): seeded where the sample is first shaded (section 2) */
					bounce = 0;
					hp = mk3(W.rec[0][sg], W.rec[1][sg], W.rec[2][sg]);
					hn = mk3(W.rec[3][sg], W.rec[4][sg], W.rec[5][sg]);
					hobj = __float_as_int(W.rec[6][sg]) & (RT_PIX_TAPS_LIT - 1);
					lit_next = (__float_as_uint(W.rec[6][sg]) >> 16) & 7u;
					hdir = mk3(W.rec[7][sg], W.rec[8][sg], W.rec[9][sg]);
					has_hit = true; f_live = true;
				}
				if (leader && gm != 0ull && avail > 0) {
					const int asked = __popcll(gm);
					const unsigned int handed = (unsigned int) (asked < avail ? asked : avail);
					W.s_nxt[sg] = nxt + handed;
					W.s_seq[sg] = seq + handed + (nxt + handed == spp ? 1u : 0u);
				}
				wave_fence();
				STAMP(6);
			}
		}
#ifdef RT_STATS
		if (!f_live) STAT(24);                  /* lanes that start the round without a sample */
#endif
		if (__ballot(f_live || ((rec1 | rec2) & REC_VALID) != 0) == 0ull) {
			/* nothing in flight: every reserved slot is filled, so whatever is still waiting can be added now */
			if (!direct && __ballot(W.s_drained[g] != W.s_seq[g]) != 0ull) { add_finished_samples(); continue; }
			if (exhausted) break;
			continue;
		}

		STAMP(0);
		/* ---- 2. shade the pending hit of every live path (main.c:180-261) ------------------- */
		int  tapmask = 0, cur = 0;
		bool emit_main = false;
		V3   ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 0);
		V3   tap_j0 = mk3(0, 0, 0), tap_j1 = mk3(0, 0, 0), tap_j2 = mk3(0, 0, 0);   /* accepted rand_dir of each tap (main.c:193) */
		STAT(7);
		if (has_hit) {
			STAT(8);
			/* a camera-ray hit point from which every tap certainly reaches the emitter (flagged by rt_primary_pass): the taps
			 * are drawn and accepted as always (main.c:193-195), but not traced */
			/* a sample's first bounce: its path is seeded here, once per round for all the lanes that start one, not once per hand-out attempt */
			if (bounce == 0) rng = path_seed(L.seed, (uint32_t) rng, (uint32_t) L.sample_base + (uint32_t) (rng >> 32));
			const bool taps_lit = AUDIT ? (lit_next & 3u) != 0u : lit_next != 0u;      /* ... or a hit point of a later bounce in such a cell of the scene's table (section 4);
			                                            * 1: they reach the emitter, 2: they certainly do not (and nothing else emits) */
#ifdef RT_STATS
			{	/* what kind of shading event this is (scripts/stats_c1.py): a sample's first or a later one, on a box or a sphere, taps known or not */
				const bool on_box = __float_as_int(sc.geom[2 * hobj + 1].z) == RT_GEOM_CUBE;
				if (bounce == 0) STAT(0);            /* (sites 25-28 are the section stamps' words) */
				if (bounce == 0 && taps_lit) STAT(10);
				if (on_box) STAT(11);
				if (bounce == 0 && on_box) STAT(18);
				if (bounce == 0 && on_box && taps_lit) STAT(19);
				if (taps_lit) STAT(29);
				if (bounce == 1) STAT(30);
			}
#endif
			if (have_light) {
				/* main.c:191-195: three rand_dir draws, a tap is skipped when it points into the surface.  Only the sign of
				 * dot(rand_dir, normal) is needed here, and random_vector() has it before normalize() does (rt_math.hip.h:
				 * side_is_certain): the tap is queued as drawn, and normalised -- with its direction and origin, main.c:197-198 --
				 * where it is traced, on a full batch, if it is traced at all. */
				tap_j0 = rng_vector(rng); tap_j1 = rng_vector(rng); tap_j2 = rng_vector(rng);
				const float side0 = dot3(tap_j0, hn), side1 = dot3(tap_j1, hn), side2 = dot3(tap_j2, hn);
				if (FAST && wave_all(side_is_certain(side0) && side_is_certain(side1) && side_is_certain(side2)))
					tapmask = (side0 > 0 ? 1 : 0) | (side1 > 0 ? 2 : 0) | (side2 > 0 ? 4 : 0);
				else
					tapmask = (dot3(unit3_of_vector<FAST>(tap_j0), hn) > 0 ? 1 : 0) | (dot3(unit3_of_vector<FAST>(tap_j1), hn) > 0 ? 2 : 0) |
					          (dot3(unit3_of_vector<FAST>(tap_j2), hn) > 0 ? 4 : 0);
			}
			const float4 m0 = sc.shade[4 * hobj], m1 = sc.shade[4 * hobj + 1];
			const V3 f0 = mk3(m0.x, m0.y, m0.z), omf0 = mk3(m1.x, m1.y, m1.z);

			const float n_dot_v = clamp01(dot3(hn, neg3(hdir)));
			/* main.c:128: (float) pow(1.0 - (double) u, 5.0) == x2*x2*x in fp64 for every float u in [0,1] (tests/test_pow5.py) */
			const double xg = 1.0 - (double) n_dot_v;
			const double xg2 = xg * xg;
			const float grazing = (float) (xg2 * xg2 * xg);
			const V3 fresnel = madd3(f0, omf0, grazing);

			V3 scatter = rng_direction<FAST>(rng);
			if (dot3(scatter, hn) < 0) scatter = neg3(scatter);

			bool specular = __float_as_int(m1.w) != 0;
			if (!specular)
				{ STAT(31); specular = rng_draw(rng) <= (FAST ? third_of(fresnel.x + fresnel.y + fresnel.z) : (fresnel.x + fresnel.y + fresnel.z) / 3.0f); }
			V3 out_dir;
			if (specular) {
				STAT(15);
				const V3 nneg = neg3(hn);
				const float f = -2.0f * dot3(nneg, hdir);
				out_dir = unit3_sel<FAST>(lin2(scatter, madd3(hdir, nneg, f), m0.w, 1.0f));
			} else
				out_dir = scatter;
			bounce++;
			emit_main = bounce < L.max_bounces;
			ray_o = madd3(hp, out_dir, 0.001f);
			ray_d = out_dir;
			hdir = out_dir;
			has_hit = false;
			cur = REC_VALID | (specular ? REC_SPECULAR : 0) | (emit_main ? 0 : REC_LAST) | (tapmask << 4) | (hobj << 8);
			if (taps_lit) {             /* the record keeps the accepted taps, the queue gets none ... */
				cur |= (AUDIT ? (lit_next & 1u) != 0u : lit_next == 1u) ? REC_TAPS_LIT : REC_TAPS_DARK;
				/* ... unless (AUDIT kernels: rt_tuning.audit_known_taps) the answer came with the audit bit -- the camera-ray pass marks one in
				 * 2^k of the pixels it classifies, the host one in 2^k cells of the scene's table --: the taps are traced like unknown ones
				 * and the back compares.  The comparison is a kernel VARIANT, not a launch constant: a round of the compiled kernel is
				 * ~500 instructions, and the eleven this costs are 2 % of the frame (profiles/r05/ab_audit_variant.txt) */
				if (AUDIT && (lit_next & 4u) != 0u && tapmask != 0) cur |= REC_AUDIT; else tapmask = 0;
			}
		}

		STAMP(1);
		/* ---- 3. the shadow taps go into the wave's ring (ballot + mbcnt prefix), one kind at a time; whenever 64 are
		 * queued, any lane traces any tap (scene.c:156-190 on full waves).  A tap is queued as (hit point, random_vector());
		 * its normalisation, direction and origin (main.c:193-198) are formed where it is traced.  Taps that do not fill a
		 * batch wait: the bounce they belong to is retired two rounds from now ------------------------------------------------------- */
		auto push = [&](bool on, V3 qo, V3 qd, int k) {
			const unsigned long long m = __ballot(on);
			if (on) {
				const unsigned int slot = (q_tail + (unsigned int) lanes_below(m)) & (WF_QUEUE - 1);
				W.q[0][slot] = qo.x; W.q[1][slot] = qo.y; W.q[2][slot] = qo.z;
				W.q[3][slot] = qd.x; W.q[4][slot] = qd.y; W.q[5][slot] = qd.z;
				W.qmeta[slot] = (unsigned short) (lane | (k << 8) | (int) (phase << 12));
			}
			q_tail += (unsigned int) __popcll(m);
			wave_fence();
		};
		/* the ray of the queued tap in ring slot `slot`, and where its answer goes */
		auto tap_ray = [&](unsigned int slot, V3 &o, V3 &d, int &meta) {
			o = mk3(W.q[0][slot], W.q[1][slot], W.q[2][slot]);
			d = mk3(W.q[3][slot], W.q[4][slot], W.q[5][slot]);
			meta = W.qmeta[slot];
			d = unit3_of_vector<FAST>(d);                                                /* main.c:193: normalize(random_vector()) */
			d = unit3_sel<FAST>(lin2(d, sub3(light_pos, o), 0.5f, 1.0f));               /* main.c:186,197 */
			o = madd3(o, d, 0.001f);                                                     /* main.c:198 */
		};
		auto tap_answer = [&](int meta, int obj) { W.tap[(meta >> 12) & 3][((meta >> 8) & 15) - 2][meta & 255] = (short) obj; };
		auto trace_taps = [&](int count) {         /* the `count` <= 64 oldest taps */
			STAT(36);
			if (CULL) {            /* every lane takes part in the culled trace (its work is shared out over the wave); lanes without a tap pass `on` = false */
				const bool on = lane < count;
				V3 o = mk3(0, 0, 0), d = mk3(1, 0, 0); int meta = 0;
				if (on) tap_ray((q_head + (unsigned int) lane) & (WF_QUEUE - 1), o, d, meta);
				const V3 dn = unit3_sel<FAST>(d);
				const Hit hit = nearest_hit_culled(sc, n, cl, cull_wave, on, o, dn, false);
				if (on) tap_answer(meta, hit.obj);
			} else
			if (lane < count) {
				STAT(37);
				V3 o, d; int meta;
				tap_ray((q_head + (unsigned int) lane) & (WF_QUEUE - 1), o, d, meta);
				const V3 dn = unit3_sel<FAST>(d);                                         /* scene.c:158 */
				const Hit hit = FAST ? NEAREST_HIT_TUNED(sc, n, o, dn, false) : nearest_hit(sc, n, o, dn);
				tap_answer(meta, hit.obj);
			}
			q_head += (unsigned int) count;
			wave_fence();
		};
		/* all three kinds in one go when the ring has room for them (it has, unless most taps of most lanes are traced): one
		 * prefix sum over the lanes' tap counts instead of three ballots, one fence instead of three (C1 -2.1 %, strip -4.6 %, C2 +0.4 %:
		 * profiles/r03/ab_push_together.txt) */
		const unsigned int mine = (unsigned int) __popc((unsigned int) tapmask);
		const unsigned long long c0 = __ballot((mine & 1u) != 0u), c1 = __ballot((mine & 2u) != 0u);
		const unsigned int all = (unsigned int) __popcll(c0) + 2u * (unsigned int) __popcll(c1);
		if (q_tail - q_head + all <= (unsigned int) WF_QUEUE) {
			unsigned int slot = q_tail + (unsigned int) lanes_below(c0) + 2u * (unsigned int) lanes_below(c1);
#pragma unroll
			for (int t = 0; t < 3; t++)
				if ((tapmask >> t) & 1) {
					const unsigned int e = slot & (WF_QUEUE - 1);
					const V3 qd = t == 0 ? tap_j0 : (t == 1 ? tap_j1 : tap_j2);
					W.q[0][e] = hp.x; W.q[1][e] = hp.y; W.q[2][e] = hp.z;
					W.q[3][e] = qd.x; W.q[4][e] = qd.y; W.q[5][e] = qd.z;
					W.qmeta[e] = (unsigned short) (lane | ((t + 2) << 8) | (int) (phase << 12));
					slot++;
				}
			q_tail += all;
			wave_fence();
			while (q_tail - q_head >= 64u) { STAT(40); trace_taps(64); }
		} else
#pragma unroll 1
		for (int kind = 2; kind < 5; kind++) {
			switch (kind) {
			case 2:  push((tapmask & 1) != 0, hp, tap_j0, 2); break;
			case 3:  push((tapmask & 2) != 0, hp, tap_j1, 3); break;
			default: push((tapmask & 4) != 0, hp, tap_j2, 4); break;
			}
			while (q_tail - q_head >= 64u) { STAT(41); trace_taps(64); }
		}

		/* ---- 4. the bounce rays: every lane traces its own, straight from its registers -- no queue, no LDS ---------- */
		if (__ballot(emit_main) != 0ull) {
			STAT(12);
			Hit culled; culled.t = 0.0f; culled.obj = -1; culled.n = mk3(0, 0, 0);
			if (CULL) {            /* (all lanes: see trace_taps) */
				/* A culled trace costs the same for one ray as for 64 (its first step asks every cluster box for every lane), and the
				 * lanes without a bounce ray -- paths at their last bounce, lanes between samples: a third of them -- have nothing to
				 * do in it: the oldest taps waiting in the ring ride along in those lanes instead of waiting for a batch of their own. */
				const unsigned long long idle = ~__ballot(emit_main);
				const unsigned int waiting = q_tail - q_head, room = (unsigned int) __popcll(idle);
				const unsigned int take = waiting < room ? waiting : room;
				const unsigned int place = (unsigned int) lanes_below(idle);
				const bool rider = !emit_main && place < take;
				V3 to = ray_o, td = emit_main ? ray_d : mk3(1, 0, 0);
				int tmeta = 0;
				if (rider) tap_ray((q_head + place) & (WF_QUEUE - 1), to, td, tmeta);
				culled = nearest_hit_culled(sc, n, cl, cull_wave, emit_main || rider, to, unit3_sel<FAST>(td), true);
				if (rider) tap_answer(tmeta, culled.obj);
				if (take) { q_head += take; wave_fence(); }
			}
			if (emit_main) {
				STAT(13);
				const V3 dn = unit3_sel<FAST>(ray_d);                                     /* scene.c:158 */
				const Hit hit = CULL ? culled : (FAST ? NEAREST_HIT_TUNED(sc, n, ray_o, dn, true) : nearest_hit(sc, n, ray_o, dn));
				hobj = hit.obj; hn = hit.n;
				hp = hit.obj >= 0 ? madd3(ray_o, dn, hit.t)                               /* scene.c:186 */
				                  : dn;                 /* left the scene: the sky is looked up in that direction (main.c:170) */
				/* will the taps from this hit point need tracing?  The scene's table (rt_lit.h) has one entry per cell of a grid
				 * over every object: 1 = every surface point in the cell certainly sees the emitter.  The load is in flight
				 * until the next round's front asks */
				lit_next = 0u;
				if (FAST && L.lit_cells != nullptr && hit.obj >= 0 && hit.obj != light_obj &&
				    rt_lit_point_on_surface(reinterpret_cast<const float*>(sc.geom) + 8 * hit.obj, hp.x, hp.y, hp.z, hn.x, hn.y, hn.z))
					lit_next = L.lit_cells[grids_in_lds ? rt_lit_bit_of(lit_grids_lds + hit.obj, hp.x, hp.y, hp.z)
					                                    : rt_lit_bit_of(lit_grids_mem + hit.obj, hp.x, hp.y, hp.z)];
			}
		}
		/* the bounces retired below were shaded two rounds ago: whatever is left of their taps (the oldest in the queue) is
		 * traced now, in a batch that need not be full */
		const unsigned int due = phase == 2u ? 0u : phase + 1u;
		while (q_tail != q_head && ((unsigned int) __builtin_amdgcn_readfirstlane((int) W.qmeta[q_head & (WF_QUEUE - 1)]) >> 12 & 3u) == due)
			{ STAT(42); trace_taps(q_tail - q_head < 64u ? (int) (q_tail - q_head) : 64); }

		STAMP(2);
		/* ---- 5. back: retire the bounce shaded two rounds ago (its taps are traced by now), take this round's
		 * bounce-ray result, and free the front when the path has ended --------------------------------------- */
		STAT(16);
		if (rec2 & REC_VALID) {
			STAT(17);
			const int pobj = (rec2 >> 8) & 1023, ptaps = (rec2 >> 4) & 7;
			const float4 m2 = sc.shade[4 * pobj + 2], m3 = sc.shade[4 * pobj + 3];
			rad = add3(rad, had3(mk3(m3.x, m3.y, m3.z), carry));                    /* main.c:232 */
			if (!(rec2 & REC_SPECULAR)) carry = had3(carry, mk3(m2.x, m2.y, m2.z));  /* main.c:248 */
			if (ptaps) {
				STAT(38);
				V3 lit = mk3(0, 0, 0);
				int taps = 0;
				if (only_light) {
					/* The emitter is the only object whose emission is not all (signed) zeros, so main.c:200-204 adds its emission once
					 * per tap that reaches it and a zero otherwise -- which changes nothing: the sum starts as +0 and is never -0.
					 * n equal terms e: e, 2e (exact), RN(2e + e) = RN(3e): the product n x e, and "+ 0" makes a -0 product the +0 the
					 * sum would be.  No emission is looked up per tap, nothing branches per tap. */
					const int t0 = W.tap[due][0][lane], t1 = W.tap[due][1][lane], t2 = W.tap[due][2][lane];
					taps = __popc((unsigned int) ptaps);
					int n_hit = ((ptaps & 1) && t0 == light_obj ? 1 : 0) + ((ptaps & 2) && t1 == light_obj ? 1 : 0) + ((ptaps & 4) && t2 == light_obj ? 1 : 0);
					if (AUDIT && (rec2 & REC_AUDIT)) audit_taps(taps, (rec2 & REC_TAPS_LIT) ? taps - n_hit : n_hit);
					if (rec2 & REC_TAPS_LIT) n_hit = taps;
					if (rec2 & REC_TAPS_DARK) n_hit = 0;
					const float4 e = sc.shade[4 * light_obj + 3];
					const float nf = (float) n_hit;
					lit = mk3(e.x * nf + 0.0f, e.y * nf + 0.0f, e.z * nf + 0.0f);
				} else
#pragma unroll
				for (int k = 0; k < 3; k++)
					if ((ptaps >> k) & 1) {
						const int obj = (rec2 & REC_TAPS_LIT) ? light_obj : ((rec2 & REC_TAPS_DARK) ? -1 : W.tap[due][k][lane]);
						if (AUDIT && (rec2 & REC_AUDIT)) audit_taps(1, ((rec2 & REC_TAPS_LIT) != 0) != (W.tap[due][k][lane] == light_obj) ? 1 : 0);
						if (obj >= 0) { const float4 e = sc.shade[4 * obj + 3]; lit = add3(lit, mk3(e.x, e.y, e.z)); }
						taps++;
					}
				/* main.c:208-209: 1.0f / num_samples, num_samples in 1..3 (the quotients as literals; RN(1/3) = 0x3eaaaaab) */
				lit = scale3(lit, FAST ? (taps == 1 ? 1.0f : (taps == 2 ? 0.5f : __uint_as_float(0x3eaaaaabu))) : 1.0f / (float) taps);
				const bool dark = FAST ? (tiny_f_fast(lit.x) && tiny_f_fast(lit.y) && tiny_f_fast(lit.z))
				                       : (tiny_f(lit.x) && tiny_f(lit.y) && tiny_f(lit.z));
				if (!dark) {                                                      /* main.c:257-261 */
					const float w = 0.05f;
					rad = madd3(rad, had3(lit, carry), w);
					carry = scale3(carry, 1.0f - w);
				}
			}
			if (rec2 & REC_LAST) { STAT(39);    /* the path ended with that bounce: sky (main.c:171) or bounce limit (main.c:158) */
				if (rec2 & REC_SKY) rad = add3(rad, had3(sky_colour<FAST>(sky2), carry));
				const V3 col = mk3(clamp01(rad.x), clamp01(rad.y), clamp01(rad.z));     /* main.c:267-269 */
				if (direct) {                   /* the pixel's only sample: 0 + colour (main.c:394), resolved (main.c:476) */
					const V3 res = scale3(add3(mk3(0, 0, 0), col), inv_spp);
					float *dst = L.frame + (size_t) slot2 * 3;
					dst[0] = res.x; dst[1] = res.y; dst[2] = res.z;
					__hip_atomic_fetch_add(&W.n_written, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				} else {                        /* into its slot of the window; section 6 adds it when its turn comes */
					const unsigned int e = (unsigned int) slot2 & (WF_LAST - 1);
					W.win[1][e] = col.y; W.win[2][e] = col.z;
					W.win[0][e] = __uint_as_float((__float_as_uint(col.x) & 0x7fffffffu) | ((slot2 & WF_LAST) ? 0x80000000u : 0u));
				}
				carry = mk3(1, 1, 1); rad = mk3(0, 0, 0);
			}
		}
		rec2 = rec1; sky2 = sky1; slot2 = slot1;
		rec1 = cur;
		if (cur & REC_VALID) {
			slot1 = f_slot;
			bool path_ended = true;                                  /* bounce limit (main.c:158) */
			if (emit_main) {
				if (hobj < 0) {
					STAT(14);
					sky1 = sky_texel<FAST>(L, hp); rec1 |= REC_LAST | REC_SKY;                  /* main.c:163-172 */
				} else {
					has_hit = true; path_ended = false;
				}
			}
			if (path_ended) f_live = false;      /* the front takes its next sample at the top of the next round */
		}
		wave_fence();

		STAMP(3);
		/* ---- 6. add the finished samples in sample order (main.c:394) ---------------------------------------- */
		/* Every second round when a pixel has many samples: a stream's eight lanes add up to eight slots per pass and a round finishes
		 * three or four samples per stream (C1), so one pass has room for two rounds' worth, and the pass costs the same 140
		 * instructions whether it finds one slot or eight.  C1 -0.5 %, C2 -3.4 %, strips -0.3 % (profiles/r03/ab_sum_every.txt;
		 * every third round: the window fills and lanes wait for slots, +1.5 ... +7 %).  Pixels of few samples complete too fast for
		 * that -- a pass resolves at most one pixel per stream. */
		if (!direct && (++sum_tick >= sum_every)) { sum_tick = 0u; add_finished_samples(); }
		STAMP(4);
	}
	STAMP_FLUSH(BLOCK / 64);
	leave_launch<BLOCK>(&W, wave);
}

#ifndef RT_SPEC_ONLY
/* (a large scene's records leave room for one or two workgroups per CU anyway: the culled variant may use 256 registers) */
template <bool FAST, bool CULL = false, bool AUDIT = false>
__global__ void __launch_bounds__(RT_BLOCK, CULL ? 2 : 4)
rt_trace_wavefront(const rt_launch L, unsigned int *block_counter)
{
	wavefront_body<FAST, CULL, RT_BLOCK, AUDIT>(L, block_counter);
}
/* The culled trace is bound by the latency of its LDS reads and cross-lane fetches, i.e. by the waves a SIMD has to switch
 * between, and the scene's records are per workgroup while a wave's own LDS is 10 KB: one workgroup of twelve waves shares
 * one copy of 1024 objects' records (36 + 12 x 10 = 156 KB: three waves per SIMD) where two workgroups of four hold two
 * copies (2 x 77 KB: two waves per SIMD). */
#define RT_BLOCK_WIDE 768
template <bool AUDIT = false>
__global__ void __launch_bounds__(RT_BLOCK_WIDE, 1)
rt_trace_wavefront_wide(const rt_launch L, unsigned int *block_counter)
{
	wavefront_body<true, true, RT_BLOCK_WIDE, AUDIT>(L, block_counter);
}
#endif

#ifdef RT_SPEC_HEADER
/* the same kernel with the trace loop specialised for one scene (rt_compile_scene, JIT) */
extern "C" __global__ void __launch_bounds__(RT_BLOCK, RT_WAVES_PER_SIMD)
rt_trace_spec(const rt_launch L, unsigned int *block_counter)
{
#ifdef RT_SPEC_AUDIT        /* (rt_compile_scene's second build of a scene, made when the audit is first asked for) */
	wavefront_body<true, false, RT_BLOCK, true>(L, block_counter);
#else
	wavefront_body<true>(L, block_counter);
#endif
}
#endif


#ifndef RT_SPEC_ONLY
/* ---- progressive accumulation: worker()'s publish step (main.c:387-396) and update_frame()'s
 * resolve (main.c:467-477) ------------------------------------------------------------------ */
/* May the launch whose control words are `control` be published?  Not when rt_cancel() cut it short (main.c:382) -- and not when it
 * is incomplete (rt_device.h RT_CTL_*: no stamp of its last wave, listed pixels not fetched or not written, camera-ray blocks
 * missing, audited taps that contradict rt_lit.h): a column is published whole or not at all (main.c:377-396).  The second case
 * is an error, counted in count[RT_COUNT_INCOMPLETE] for the host's next look (rt_progressive_count). */
RT_DEV bool launch_may_be_published(const unsigned int *control, unsigned int launch_id, int stamped, unsigned int primary_blocks, float *count)
{
	if (!control) return true;            /* nothing was launched (a rank without rows at this scale) */
	/* The stamp FIRST: a launch that keeps its scratch set's pixel lists clears nothing (rt_api.cpp), so until its own last wave has
	 * written the line, CANCELLED is what the set's previous launch left there -- a launch that died without a last wave behind a
	 * cancelled one must count as incomplete, not as cancelled.  (A launch cut short by rt_cancel() still ends with its stamp.) */
	const bool stamp_ok = !stamped || control[RT_CTL_STAMP] == launch_id;
	if (stamp_ok && control[RT_CTL_CANCELLED]) return false;
	if (!stamped) return true;
	const bool complete = stamp_ok && control[RT_CTL_FETCHED] == control[RT_CTL_LISTED] &&
	                      control[RT_CTL_WRITTEN] == control[RT_CTL_LISTED] && control[RT_CTL_PRIMARY] == primary_blocks &&
	                      control[RT_CTL_DISAGREE] == 0u && control[RT_CTL_DISAGREE + 1] == 0u;
	if (!complete && blockIdx.x == 0 && threadIdx.x == 0) {
		unsigned int *errors = reinterpret_cast<unsigned int*>(count) + RT_COUNT_INCOMPLETE;
		*errors = *errors + 1u;
	}
	return complete;
}

extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_accumulate(float *accum, const float *lowres, int width, int height, int scale, int low_w, int low_h, float k,
              const unsigned int *control, unsigned int launch_id, int stamped, unsigned int primary_blocks, float *count,
              int row_block, int rank, int world, int local_rows)
{
	if (!launch_may_be_published(control, launch_id, stamped, primary_blocks, count)) return;
	/* accum_counts[] += weight (main.c:396) lives beside the buffer it describes: a pass that rt_cancel() cut short
	 * leaves both untouched, whatever the host believed when it enqueued the pass */
	if (blockIdx.x == 0 && threadIdx.x == 0) *count = *count + k;
	/* `accum` holds the frame rows of this rank's row blocks (blocks of `row_block` frame rows dealt round-robin: one rank,
	 * the whole frame), `lowres` the low-resolution rows that cover them: blocks of row_block / scale rows, same deal */
	const size_t total = (size_t) width * local_rows;
	const int low_block = row_block / scale;
	for (size_t p = (size_t) blockIdx.x * RT_BLOCK + threadIdx.x; p < total; p += (size_t) gridDim.x * RT_BLOCK) {
		const int ly = (int) (p / (size_t) width), x = (int) (p % (size_t) width);
		const int y = world == 1 ? ly : global_row(row_block, rank, world, ly);
		if (y >= height) continue;            /* padding rows of the last block */
		const int j = y / scale, i = x / scale;
		if (j >= low_h) continue;             /* rows the low-resolution pass did not paint */
		const int lj = world == 1 ? j : (j / low_block / world) * low_block + j % low_block;
		const float *c = lowres + ((size_t) lj * low_w + i) * 3;
		float *a = accum + p * 3;
		a[0] = a[0] + c[0] * k; a[1] = a[1] + c[1] * k; a[2] = a[2] + c[2] * k;
	}
}

/* the publish step of `passes` full-resolution passes that one launch rendered onto the sums so far (rt_launch.sum_onto):
 * the new sums take the old ones' place, and the passes count (weight 1 each, main.c:396) -- unless the launch was cut short */
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_commit_sums(float *accum, const float *sums, size_t floats, int passes,
               const unsigned int *control, unsigned int launch_id, int stamped, unsigned int primary_blocks, float *count)
{
	if (!launch_may_be_published(control, launch_id, stamped, primary_blocks, count)) return;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		float c = *count;
		for (int k = 0; k < passes; k++) c = c + 1.0f;
		*count = c;
	}
	for (size_t p = (size_t) blockIdx.x * RT_BLOCK + threadIdx.x; p < floats; p += (size_t) gridDim.x * RT_BLOCK)
		accum[p] = sums[p];
}

extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_resolve(const float *accum, float *frame, size_t floats, const float *count)
{
	const float inv_count = 1.0f / *count;                                       /* main.c:476 */
	for (size_t p = (size_t) blockIdx.x * RT_BLOCK + threadIdx.x; p < floats; p += (size_t) gridDim.x * RT_BLOCK)
		frame[p] = accum[p] * inv_count;
}

/* ---- de-interleave: gathered per-rank strips -> full frame (multi-GPU root) ------------------- */
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_deinterleave(const float *strips, float *frame, int width, int height, int row_block, int world, int rows_per_rank, int first)
{
	const size_t row_floats = (size_t) width * 3;
	const size_t total = (size_t) height * row_floats;
	for (size_t k = (size_t) blockIdx.x * RT_BLOCK + threadIdx.x; k < total; k += (size_t) gridDim.x * RT_BLOCK) {
		const int j = (int) (k / row_floats);
		const size_t c = k % row_floats;
		/* strip s = blk % world sits at position (s + first) % world of the gathered buffer: the host may hand the strips out
		 * rotated, so that the root renders the shortest one (first = 1: rank r renders strip r - 1, rank 0 the last) */
		const int blk = j / row_block, rank = (blk % world + first) % world, lblk = blk / world;
		const int lr = lblk * row_block + j % row_block;
		frame[k] = strips[((size_t) rank * rows_per_rank + lr) * row_floats + c];
	}
}

/* ---- self-test of the exact-arithmetic shortcuts (tests/test_gpu_selftest.py) -------------------
 * which = 0: div_by_refined   vs `/`   on floats,  numerator in [2^-100, 2^30], denominator in [2^-30, 2^30]
 *         1: div_by_refined64 vs `/`   on doubles, numerator in [2^-300, 2^300], denominator (double) 2a for every float a with
 *            |a - 1| <= 2^-18 (dot(d,d) of a normalised direction), and its series reciprocal vs 1.0 / den
 *         2: unit3_fast       vs unit3 on vectors of every magnitude (incl. zero / tiny / huge components)
 *         4: tiny_f_fast      vs tiny_f (the |x| < 0.0001 test of vector.c:79 without fp64)
 *         3: EXHAUSTIVE: sqrt_of_float64 vs the IEEE fp64 sqrt for every normal float up to 2^120 (`iters` ignored)
 *         5: EXHAUSTIVE: rcp_refined vs 1/d for all 2^23 significands; div_by_refined vs `/` for every d significand x
 *            `iters` n significands (iters = 2^23 = all 2^46 pairs, 52 s; `seed` picks the first numerator)
 *         6: EXHAUSTIVE: sqrt_in_window vs sqrtf (+ the refined reciprocal of the root) for every float in [2^-30, 2^60]
 *         7: unit3_of_draws   vs unit3 on vectors with `draw * 2 - 1` components (rng_direction)
 *         8: side_is_certain: the sign of dot(v, n) vs the sign of dot(normalize(v), n) wherever it answers yes
 * out[0] = number of mismatching results, out[1..] = operands of one mismatch. */
RT_DEV uint64_t st_next(uint64_t &s)
{
	s += 0x9E3779B97F4A7C15ull;
	uint64_t z = s;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

RT_DEV float st_float(uint64_t r, int emin, int emax)    /* random sign/mantissa, exponent uniform in [emin, emax) */
{
	const uint32_t mant = (uint32_t) r & 0x7fffffu;
	const uint32_t sign = (uint32_t) (r >> 23) & 1u;
	const int e = emin + (int) ((r >> 32) % (uint64_t) (emax - emin));
	return __uint_as_float((sign << 31) | ((uint32_t) (e + 127) << 23) | mant);
}

extern "C" __global__ void __launch_bounds__(RT_BLOCK)
rt_selftest_kernel(int which, uint64_t seed, int iters, unsigned long long *out)
{
	uint64_t s = seed + 0x1000003ull * (uint64_t) (blockIdx.x * RT_BLOCK + threadIdx.x);
	unsigned long long bad = 0;
	if (which == 5) {
		/* every denominator significand x `iters` numerator significands (iters = 2^23: all pairs).  The
		 * sequences are invariant under power-of-two scaling inside the window, so [1,2) x [1,2) is enough. */
		const uint32_t step = 0x9E3779B1u & 0x7fffffu, offset = (uint32_t) seed & 0x7fffffu;      /* odd: a bijection mod 2^23 */
		for (uint32_t md = blockIdx.x * RT_BLOCK + threadIdx.x; md < (1u << 23); md += gridDim.x * RT_BLOCK) {
			const float d = __uint_as_float(0x3f800000u | md);
			const float r = rcp_refined(d);
			if (__float_as_uint(r) != __float_as_uint(1.0f / d)) { bad++; out[1] = __float_as_uint(d); out[3] = __float_as_uint(1.0f / d); out[4] = __float_as_uint(r); }
			uint32_t mn = offset;
			for (int it = 0; it < iters; it++, mn = (mn + step) & 0x7fffffu) {
				const float n = __uint_as_float(0x3f800000u | mn);
				const float want = n / d, got = div_by_refined(n, d, r);
				if (__float_as_uint(want) != __float_as_uint(got)) {
					bad++;
					out[1] = __float_as_uint(n); out[2] = __float_as_uint(d); out[3] = __float_as_uint(want); out[4] = __float_as_uint(got);
				}
			}
		}
		if (bad) atomicAdd(&out[0], bad);
		return;
	}
	if (which == 3) {
		/* every normal float up to 2^120: sqrt_of_float64 vs the IEEE fp64 sqrt */
		const uint32_t first = 0x00800000u, count = 0x3f800000u + (120u << 23) - first + 1u;
		for (uint32_t k = blockIdx.x * RT_BLOCK + threadIdx.x; k < count; k += gridDim.x * RT_BLOCK) {
			const float f = __uint_as_float(first + k);
			const double want = __builtin_sqrt((double) f), got = sqrt_of_float64(f);
			if (__double_as_longlong(want) != __double_as_longlong(got)) {
				bad++;
				out[1] = __float_as_uint(f); out[3] = (unsigned long long) __double_as_longlong(want); out[4] = (unsigned long long) __double_as_longlong(got);
			}
		}
		if (bad) atomicAdd(&out[0], bad);
		return;
	}
	if (which == 6) {
		/* every float in [2^-30, 2^60]: sqrt_in_window vs sqrtf, and the refined reciprocal of the root */
		const uint32_t first = 0x3f800000u - (30u << 23), count = (90u << 23) + 1u;
		for (uint32_t k = blockIdx.x * RT_BLOCK + threadIdx.x; k < count; k += gridDim.x * RT_BLOCK) {
			const float x = __uint_as_float(first + k);
			const float want = __builtin_sqrtf(x), got = sqrt_in_window(x);
			if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(rcp_refined(want)) != __float_as_uint(1.0f / want)) {
				bad++;
				out[1] = __float_as_uint(x); out[3] = __float_as_uint(want); out[4] = __float_as_uint(got);
			}
		}
		if (bad) atomicAdd(&out[0], bad);
		return;
	}
	for (int it = 0; it < iters; it++) {
		if (which == 0) {
			const uint64_t r0 = st_next(s), r1 = st_next(s);
			float n = st_float(r0, -100, 30), d = st_float(r1, -30, 20);
			if ((r0 >> 60) == 0) d = __uint_as_float((__float_as_uint(d) & 0xff800000u) | ((uint32_t) (r1 >> 40) & 0x3u));   /* near powers of two */
			if ((r1 >> 60) == 1) n = d * st_float(r0, -2, 2);                        /* correlated operands */
			if ((r1 >> 56) == 0x2f) n = 0.0f;                                      /* +0 numerator: a ray origin on a slab plane */
			const float want = n / d;
			const float got = div_by_refined(n, d, rcp_refined(d));
			if (__float_as_uint(want) != __float_as_uint(got)) {
				bad++;
				out[1] = __float_as_uint(n); out[2] = __float_as_uint(d); out[3] = __float_as_uint(want); out[4] = __float_as_uint(got);
			}
			/* avgv()'s division by 3 (mixed with -0, denormal and huge numerators, which take the wave's `/` path) */
			float m = n;
			if ((r0 >> 56) == 0x11) m = -0.0f;
			if ((r0 >> 56) == 0x12) m = __uint_as_float((uint32_t) r1 & 0x807fffffu);
			if ((r0 >> 56) == 0x13) m = st_float(r1, 30, 127);
			const float want3 = m / 3.0f, got3 = third_of(m);
			if (__float_as_uint(want3) != __float_as_uint(got3)) {
				bad++;
				out[1] = __float_as_uint(m); out[2] = __float_as_uint(3.0f); out[3] = __float_as_uint(want3); out[4] = __float_as_uint(got3);
			}
		} else if (which == 1) {
			const uint64_t r0 = st_next(s), r1 = st_next(s), r2 = st_next(s);
			const float a = __uint_as_float(0x3f800000u - 64u + (uint32_t) (r0 % 97ull));      /* every float with |a - 1| <= 2^-18 */
			const double den = (double) (2.0f * a);
			const int e = -300 + (int) (r1 % 600ull);
			double num = __builtin_ldexp(1.0 + (double) (r2 >> 12) * 0x1p-52, e);
			if (r1 >> 63) num = -num;
			if ((r1 >> 58 & 15) == 0) num = (double) st_float(r2, -60, 30) - __builtin_sqrt((double) __builtin_fabsf(st_float(r1, -60, 30)));
			if (!(__builtin_fabs(num) >= 0x1p-300 && __builtin_fabs(num) <= 0x1p+300)) continue;
			const double want = num / den;
			const double got = div_by_refined64(num, den, rcp_twice_near_one(a));
			if (!near_one(a) || __double_as_longlong(rcp_twice_near_one(a)) != __double_as_longlong(1.0 / den)) { bad++; out[1] = __float_as_uint(a); }
			if (__double_as_longlong(want) != __double_as_longlong(got)) {
				bad++;
				out[1] = (unsigned long long) __double_as_longlong(num); out[2] = (unsigned long long) __double_as_longlong(den);
				out[3] = (unsigned long long) __double_as_longlong(want); out[4] = (unsigned long long) __double_as_longlong(got);
			}
		} else if (which == 7) {
			/* unit3_of_draws on vectors of `draw * 2 - 1` components: draws anywhere in [0, 1], with extra weight on
			 * 0, 1, the neighbourhood of 0.5 (components of 0 and +-2^-24) and all three components tiny at once */
			const uint64_t r0 = st_next(s), r1 = st_next(s), r2 = st_next(s), r3 = st_next(s);
			float dr[3] = { (float) r0 * 0x1p-64f, (float) r1 * 0x1p-64f, (float) r2 * 0x1p-64f };
			const bool all_near_half = (it & 7) == 7 && ((r3 >> 40) & 3) == 0;
			for (int c = 0; c < 3; c++) {
				const uint32_t pick = (uint32_t) (r3 >> (8 * c)) & 255u;
				if (pick == 0) dr[c] = 0.0f;
				if (pick == 1) dr[c] = 1.0f;
				if (pick == 2) dr[c] = 0.5f;
				if ((pick < 16 && pick > 2) || all_near_half)
					dr[c] = __uint_as_float(0x3f000000u + (int) ((r3 >> (32 + 2 * c)) % (pick < 8 || all_near_half ? 9u : 4097u)) - (int) (pick < 8 || all_near_half ? 4 : 2048));
			}
			const V3 v = mk3(dr[0] * 2.0f - 1.0f, dr[1] * 2.0f - 1.0f, dr[2] * 2.0f - 1.0f);
			const V3 want = unit3(v), got = unit3_of_draws(v);
			/* the draw itself: the conversion of rng_draw() against the plain one, on words with any number of leading zeros, on
			 * rounding boundaries (a run of ones / a one and zeros below the 24th bit) and on 0, 1 and 2^64 - 1; and the fused
			 * draw * 2 - 1 of rng_vector() against the two operations */
			const uint64_t word = r3 >> (r1 & 63);
			const uint64_t cases[6] = { r0, word, word | (word >> 25), (word >> 40 << 40) | (1ull << 39) >> (r2 & 1), (r2 & 3) == 0 ? 0ull : 1ull, ~0ull >> (r0 & 63) };
			bool draw_ok = true;
			for (int c = 0; c < 6; c++)
				draw_ok = draw_ok && __float_as_uint(unit_float_of_bits(cases[c])) == __float_as_uint((float) cases[c] * 0x1p-64f);
			for (int c = 0; c < 3; c++)
				draw_ok = draw_ok && __float_as_uint(__builtin_fmaf(dr[c], 2.0f, -1.0f)) == __float_as_uint(dr[c] * 2.0f - 1.0f);
			if (!draw_ok) { bad++; out[6] = (uint32_t) word; out[7] = (uint32_t) (word >> 32); }
			if (__float_as_uint(want.x) != __float_as_uint(got.x) || __float_as_uint(want.y) != __float_as_uint(got.y) ||
			    __float_as_uint(want.z) != __float_as_uint(got.z)) {
				bad++;
				out[1] = __float_as_uint(v.x); out[2] = __float_as_uint(v.y); out[3] = __float_as_uint(v.z);
				out[4] = __float_as_uint(want.x); out[5] = __float_as_uint(got.x);
			}
		} else if (which == 8) {
			/* side_is_certain: where it says yes, dot(v, n) > 0 must equal dot(normalize(v), n) > 0 (main.c:194).  v as in
			 * case 7; n an axis (cube faces), a random unit vector, or a unit vector all but perpendicular to v */
			const uint64_t r0 = st_next(s), r1 = st_next(s), r2 = st_next(s), r3 = st_next(s), r4 = st_next(s), r5 = st_next(s);
			float dr[3] = { (float) r0 * 0x1p-64f, (float) r1 * 0x1p-64f, (float) r2 * 0x1p-64f };
			for (int c = 0; c < 3; c++) {
				const uint32_t pick = (uint32_t) (r3 >> (8 * c)) & 255u;
				if (pick == 0) dr[c] = 0.0f;
				if (pick == 1) dr[c] = 1.0f;
				if (pick == 2) dr[c] = 0.5f;
				if (pick < 16 && pick > 2) dr[c] = __uint_as_float(0x3f000000u + (int) ((r3 >> (32 + 2 * c)) % (pick < 8 ? 9u : 4097u)) - (int) (pick < 8 ? 4 : 2048));
			}
			const V3 v = mk3(dr[0] * 2.0f - 1.0f, dr[1] * 2.0f - 1.0f, dr[2] * 2.0f - 1.0f);
			V3 nn;
			const uint32_t kind = (uint32_t) (r3 >> 56) & 3u;
			const V3 w = mk3((float) r4 * 0x1p-63f - 1.0f, (float) (r4 >> 20) * 0x1p-43f - 1.0f, (float) r5 * 0x1p-63f - 1.0f);
			if (kind == 0) { const int ax = (int) ((r5 >> 3) % 3u); const float sg = (r5 & 1) ? 1.0f : -1.0f; nn = mk3(ax == 0 ? sg : 0.0f, ax == 1 ? sg : 0.0f, ax == 2 ? sg : 0.0f); }
			else if (kind == 1) nn = unit3(w);
			else {          /* cross(v, w) is perpendicular to v; a pinch of v decides the side */
				const V3 c = mk3(v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x);
				const float eps = ((r5 >> 8) & 1 ? 1.0f : -1.0f) * __uint_as_float(0x2f800000u + ((uint32_t) (r5 >> 12) & 0x0fffffffu));   /* 2^-32 .. 2^0 */
				nn = unit3(madd3(unit3(c), v, eps));
			}
			const float side = dot3(v, nn);
			if (side_is_certain(side) && (side > 0) != (dot3(unit3(v), nn) > 0)) {
				bad++;
				out[1] = __float_as_uint(v.x); out[2] = __float_as_uint(v.y); out[3] = __float_as_uint(v.z);
				out[4] = __float_as_uint(nn.x); out[5] = __float_as_uint(nn.y);
			}
		} else if (which == 4) {
			const uint64_t r0 = st_next(s);
			float f = st_float(r0, -16, -10);                                  /* around 1e-4 = 2^-13.3 */
			if ((r0 >> 60) == 0) f = __uint_as_float(0x38D1B717u + (uint32_t) ((r0 >> 8) & 7) - 3u);
			if ((r0 >> 60) == 1) f = -__uint_as_float(0x38D1B717u + (uint32_t) ((r0 >> 8) & 7) - 3u);
			if ((r0 >> 60) == 2) f = st_float(r0, -126, 127);
			if ((r0 >> 56) == 0x30) f = __uint_as_float(0x7fc00000u);
			if (tiny_f(f) != tiny_f_fast(f)) { bad++; out[1] = __float_as_uint(f); }
			const float byte = (float) (uint32_t) (r0 & 255u);                 /* and the sky texel channels: b / 255 */
			if (__float_as_uint(byte / 255.0f) != __float_as_uint(div_by_refined(byte, 255.0f, __uint_as_float(0x3b808081u)))) { bad++; out[1] = __float_as_uint(byte); }
		} else {
			const uint64_t r0 = st_next(s), r1 = st_next(s), r2 = st_next(s), r3 = st_next(s);
			/* unit3_fast decides per WAVE: even iterations keep a whole wave inside the tuned form's window,
			 * odd ones mix magnitudes and special values so that single lanes push their wave out of it */
			int lo = -14, hi = 29;
			const bool mixed = (it & 1) != 0;
			if (mixed) switch (r3 & 7) { case 0: lo = -126; hi = 127; break; case 1: lo = -30; hi = -10; break; case 2: lo = 20; hi = 64; break; default: lo = -20; hi = 20; break; }
			V3 v = mk3(st_float(r0, lo, hi), st_float(r1, lo, hi), st_float(r2, lo, hi));
			if (mixed && ((r3 >> 8) & 15) == 0) v.x = 0.0f;
			if (mixed && ((r3 >> 12) & 15) == 0) v.y = -0.0f;
			if (mixed && ((r3 >> 16) & 31) == 0) v.z = __uint_as_float((uint32_t) r2 & 0x807fffffu);   /* denormal */
			if (mixed && ((r3 >> 24) & 63) == 0) { const float k = 0.00001f / __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); v = scale3(v, k); }
			const V3 want = unit3(v), got = unit3_fast(v);
			if (__float_as_uint(want.x) != __float_as_uint(got.x) || __float_as_uint(want.y) != __float_as_uint(got.y) ||
			    __float_as_uint(want.z) != __float_as_uint(got.z)) {
				bad++;
				out[1] = __float_as_uint(v.x); out[2] = __float_as_uint(v.y); out[3] = __float_as_uint(v.z);
				out[4] = __float_as_uint(want.x); out[5] = __float_as_uint(got.x);
			}
		}
	}
	if (bad) atomicAdd(&out[0], bad);
}

hipError_t rt_launch_accumulate(float *accum, const float *lowres, int width, int height, int scale,
                                int low_w, int low_h, float k, const unsigned int *control, const rt_launch_expect &expect, float *count,
                                int row_block, int rank, int world, int local_rows, hipStream_t stream)
{
	hipLaunchKernelGGL(rt_accumulate, dim3(2048), dim3(RT_BLOCK), 0, stream, accum, lowres, width, height, scale, low_w, low_h, k,
	                   control, expect.launch_id, expect.stamped, expect.primary_blocks, count, row_block, rank, world, local_rows);
	return hipGetLastError();
}

hipError_t rt_launch_commit_sums(float *accum, const float *sums, size_t floats, int passes, const unsigned int *control, const rt_launch_expect &expect, float *count, hipStream_t stream)
{
	hipLaunchKernelGGL(rt_commit_sums, dim3(2048), dim3(RT_BLOCK), 0, stream, accum, sums, floats, passes,
	                   control, expect.launch_id, expect.stamped, expect.primary_blocks, count);
	return hipGetLastError();
}

hipError_t rt_launch_resolve(const float *accum, float *frame, size_t floats, const float *count, hipStream_t stream)
{
	hipLaunchKernelGGL(rt_resolve, dim3(2048), dim3(RT_BLOCK), 0, stream, accum, frame, floats, count);
	return hipGetLastError();
}

hipError_t rt_launch_selftest(int which, uint64_t seed, int blocks, int iters, unsigned long long *d_out, hipStream_t stream)
{
	hipLaunchKernelGGL(rt_selftest_kernel, dim3(blocks), dim3(RT_BLOCK), 0, stream, which, seed, iters, d_out);
	return hipGetLastError();
}


/* ---- host-callable launchers (C++ linkage inside the library; the C ABI lives in rt_api.cpp) -- */

size_t rt_counter_bytes() { return RT_COUNTER_BYTES; }

size_t rt_scene_lds_bytes(int num_objects) { return (size_t) num_objects * (sizeof(rt_geom) + sizeof(rt_shade)); }

size_t rt_wavefront_lds_bytes(int num_objects) { return rt_scene_lds_bytes(num_objects) + (RT_BLOCK / 64) * sizeof(WaveLDS); }
static_assert(sizeof(rt_cluster) == 16 * RT_CLUSTER_F4 && sizeof(rt_group) == 16 * RT_GROUP_F4 && RT_CLUSTER_SIZE == 8 && RT_GROUP_SIZE == 8 && RT_CLUSTER_F4 % 2 == 1 && RT_GROUP_F4 % 2 == 1, "the kernels read a cluster / a group as float4 words: box, count, (eight members,) eight boxes as four pairs of three words");

/* rt_primary_pass: a few workgroups per CU, each with a run of consecutive 8x8 pixel blocks (at least one per wave) */
void rt_primary_geometry(int width, int local_rows, int num_cus, unsigned int *groups_out, int *per_group_out)
{
	const unsigned int pixel_blocks = (unsigned int) (((width + 7) / 8) * ((local_rows + 7) / 8));
	unsigned int groups = (unsigned int) num_cus * 8u;
	if (groups * (RT_BLOCK / 64) > pixel_blocks) groups = (pixel_blocks + RT_BLOCK / 64 - 1) / (RT_BLOCK / 64);
	if (groups < 1) groups = 1;
	const int per_group = (int) ((pixel_blocks + groups - 1) / groups) > 0 ? (int) ((pixel_blocks + groups - 1) / groups) : 1;
	groups = (pixel_blocks + (unsigned int) per_group - 1) / (unsigned int) per_group;
	*groups_out = groups; *per_group_out = per_group;
}

/* records one pixel list must be able to hold: workgroup w of rt_primary_pass appends to list w % num_shards */
size_t rt_pixel_list_capacity(int width, int local_rows, int num_cus, int num_shards)
{
	unsigned int groups; int per_group;
	rt_primary_geometry(width, local_rows, num_cus, &groups, &per_group);
	const size_t groups_per_list = (groups + (unsigned int) num_shards - 1) / (unsigned int) num_shards;
	return groups_per_list * (size_t) per_group * 64;
}

hipError_t rt_launch_trace(const rt_launch &L, int variant, bool scene_fast_ok, hipFunction_t spec_fn,
                           unsigned int *block_counter, hipEvent_t cleared, hipEvent_t primary_done, int num_cus, int workgroups_per_cu, hipStream_t stream,
                           bool reuse_pixel_lists, rt_launch_expect *expect, hipEvent_t primary_timed, bool audit)
{
	rt_launch_expect unused;
	if (!expect) expect = &unused;
	expect->launch_id = L.launch_id; expect->stamped = 0; expect->primary_blocks = 0u;
	if (L.local_rows <= 0 || L.width <= 0) {
		hipError_t e = cleared ? hipEventRecord(cleared, stream) : hipSuccess;
		return e != hipSuccess ? e : hipEventRecord(primary_done, stream);
	}
	const bool simple = variant == 1 /* RT_KERNEL_SIMPLE */ || L.max_bounces < 1;
	if (simple) {
		/* it has no use for the counters, but every launch leaves its scratch set's control words describing itself */
		hipError_t e = hipMemsetAsync(block_counter, 0, RT_COUNTER_BYTES, stream);
		if (e == hipSuccess && cleared) e = hipEventRecord(cleared, stream);
		if (e != hipSuccess) return e;
		const int tiles_x = (L.width + RT_TILE_W - 1) / RT_TILE_W;
		const int tiles_y = (L.local_rows + RT_TILE_H - 1) / RT_TILE_H;
		hipLaunchKernelGGL(rt_trace_simple, dim3(tiles_x * tiles_y), dim3(RT_BLOCK), rt_scene_lds_bytes(L.num_objects), stream, L);
		e = hipGetLastError();
		return e != hipSuccess ? e : hipEventRecord(primary_done, stream);
	}
	/* persistent waves: as many workgroups as fit on the chip at once, capped by the work */
	/* (a scene the host has had compiled keeps its compiled kernel: clusters are built from RT_CULL_MIN_OBJECTS objects, scenes
	 * of up to 64 can be compiled) */
	const bool cull = L.num_clusters > 0 && L.clusters != nullptr && scene_fast_ok && variant == 0 && !spec_fn;
	size_t lds = cull ? (size_t) RT_CULL_F4(L.num_clusters) * 16 + (RT_BLOCK / 64) * (sizeof(WaveLDS) + sizeof(CullWave))
	                  : rt_wavefront_lds_bytes(L.num_objects);
	int per_cu = (int) ((160u * 1024u) / lds);
	if (per_cu < 1) per_cu = 1;
	if (per_cu > 4) per_cu = 4;
	/* the grids of the lit-taps table go into LDS as well if that costs no workgroup per CU (else they are read from memory) */
	rt_launch Lq = L;
	Lq.lit_grids_in_lds = 0;
	/* (the culled kernel's LDS layout has no place for the lit-taps grids: a culled scene that has the table -- 40 to 64 objects --
	 * reads them from memory) */
	if (Lq.lit_cells != nullptr && !cull) {
		const size_t with = lds + (size_t) L.num_objects * 48;
		if (with <= 160u * 1024u && (int) ((160u * 1024u) / with) >= per_cu) { Lq.lit_grids_in_lds = 1; lds = with; }
	}
	if (cull && per_cu > 3) per_cu = 3;                                                        /* (the culled variant's 136 registers: three waves per SIMD) */
	/* a large scene whose records leave room for fewer than three workgroups of four waves: one workgroup of twelve (rt_trace_wavefront_wide) */
	int block = RT_BLOCK;
	if (cull && per_cu < 3 && workgroups_per_cu < 1) {
		const size_t wide = (size_t) RT_CULL_F4(L.num_clusters) * 16 + (RT_BLOCK_WIDE / 64) * (sizeof(WaveLDS) + sizeof(CullWave));
		if (wide <= 160u * 1024u) { block = RT_BLOCK_WIDE; lds = wide; per_cu = 1; }
	}
	if (workgroups_per_cu >= 1 && workgroups_per_cu < per_cu) per_cu = workgroups_per_cu;     /* rt_tuning */
	const long long blocks = (long long) ((L.width + 7) / 8) * ((L.local_rows + 7) / 8);
	/* what the launch must leave in its control words (rt_device.h RT_CTL_*): the stamp of its last wave, and -- the lists'
	 * lines keep the count when the lists are kept -- every 8x8 block finished by the camera-ray pass */
	expect->stamped = 1; expect->primary_blocks = (unsigned int) blocks;
	long long grid = (long long) num_cus * per_cu;
	const long long useful = (blocks + (block / 64) - 1) / (block / 64);
	if (grid > useful) grid = useful;
	if (grid < 1) grid = 1;
	Lq.trace_workgroups = (int) grid;
	/* counter block: WF_SHARDS dequeue counters, WF_SHARDS fill counters, one line of control words.  When the lists of the
	 * previous launch of this scratch set are this launch's lists (an interactive pass with nothing changed but the
	 * sample number: rt_api.cpp), the fill counters and the records stay and rt_primary_pass is not run
	 * -- and nothing is cleared: the last workgroup of a launch leaves the dequeue lines and the control line as the next launch
	 * must find them (leave_launch) */
	hipError_t e = reuse_pixel_lists ? hipSuccess : hipMemsetAsync(block_counter, 0, (size_t) RT_COUNTER_BYTES, stream);
	if (e == hipSuccess && cleared) e = hipEventRecord(cleared, stream);
	if (e != hipSuccess) return e;
	if (reuse_pixel_lists) {
		e = hipEventRecord(primary_done, stream);
		if (e != hipSuccess) return e;
	} else {
		unsigned int groups; int per_group;
		rt_primary_geometry(L.width, L.local_rows, num_cus, &groups, &per_group);
		const size_t plds = cull ? (size_t) RT_CULL_F4(L.num_clusters) * 16 + (RT_BLOCK / 64) * sizeof(CullWave)
		                         : rt_scene_lds_bytes(L.num_objects);
		if (variant == 2 || !scene_fast_ok)
			hipLaunchKernelGGL(rt_primary_pass<false>, dim3(groups), dim3(RT_BLOCK), plds, stream, L, per_group);
		else if (cull)
			hipLaunchKernelGGL((rt_primary_pass<true, true>), dim3(groups), dim3(RT_BLOCK), plds, stream, L, per_group);
		else
			hipLaunchKernelGGL(rt_primary_pass<true>, dim3(groups), dim3(RT_BLOCK), plds, stream, L, per_group);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipEventRecord(primary_done, stream);       /* the next launch of the context may start (rt_api.cpp) */
		if (e != hipSuccess) return e;
	}
	if (primary_timed) { e = hipEventRecord(primary_timed, stream); if (e != hipSuccess) return e; }      /* (rt_profile_*: between the two kernels) */
	if (variant == 0 /* RT_KERNEL_AUTO */ && scene_fast_ok && spec_fn) {
		/* same kernel, trace loop specialised for this scene by rt_compile_scene() */
		rt_launch Lc = Lq;
		unsigned int *counter = block_counter;
		void *args[] = { &Lc, &counter };
		return hipModuleLaunchKernel(spec_fn, (unsigned int) grid, 1, 1, RT_BLOCK, 1, 1, (unsigned int) lds, stream, args, nullptr);
	}
	if (variant == 2 /* RT_KERNEL_WAVEFRONT: same schedule, plain IEEE operations */ || !scene_fast_ok)
		hipLaunchKernelGGL(rt_trace_wavefront<false>, dim3((unsigned int) grid), dim3(RT_BLOCK), lds, stream, Lq, block_counter);
	else if (cull && block == RT_BLOCK_WIDE) {
		if (audit) hipLaunchKernelGGL(rt_trace_wavefront_wide<true>, dim3((unsigned int) grid), dim3(RT_BLOCK_WIDE), lds, stream, Lq, block_counter);
		else hipLaunchKernelGGL(rt_trace_wavefront_wide<false>, dim3((unsigned int) grid), dim3(RT_BLOCK_WIDE), lds, stream, Lq, block_counter);
	} else if (cull) {
		if (audit) hipLaunchKernelGGL((rt_trace_wavefront<true, true, true>), dim3((unsigned int) grid), dim3(RT_BLOCK), lds, stream, Lq, block_counter);
		else hipLaunchKernelGGL((rt_trace_wavefront<true, true>), dim3((unsigned int) grid), dim3(RT_BLOCK), lds, stream, Lq, block_counter);
	} else {
		if (audit) hipLaunchKernelGGL((rt_trace_wavefront<true, false, true>), dim3((unsigned int) grid), dim3(RT_BLOCK), lds, stream, Lq, block_counter);
		else hipLaunchKernelGGL(rt_trace_wavefront<true>, dim3((unsigned int) grid), dim3(RT_BLOCK), lds, stream, Lq, block_counter);
	}
	return hipGetLastError();
}

hipError_t rt_launch_deinterleave(const float *strips, float *frame, int width, int height,
                                  int row_block, int world, int rows_per_rank, int first, hipStream_t stream)
{
	hipLaunchKernelGGL(rt_deinterleave, dim3(2048), dim3(RT_BLOCK), 0, stream,
	                   strips, frame, width, height, row_block, world, rows_per_rank, first);
	return hipGetLastError();
}
#endif /* RT_SPEC_ONLY */
