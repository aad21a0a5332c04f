/* CPU check of csrc/rt_cull.h (tests/test_cull_margins.py): the cluster structure, and the soundness of the conservative tests --
 * for random scenes and rays, many of them aimed at box edges, box corners and sphere tangents, WHENEVER the reference's own
 * float test of an object reports a hit with t >= 0 (scene.c:17-77 slab test with correctly rounded quotients, scene.c:79-134 float
 * discriminant; the acceptance of scene.c:168), the cull's test of that object's conservative box and of its cluster's box --
 * products by the correctly rounded reciprocal of the direction, as nearest_hit_culled forms them -- must pass.
 * usage: cull_check <scenes> <rays per scene>   -> prints "<pairs tested> <reference hits> <violations> <structure errors> <scenes refused>" */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../ray_tracing_amd/csrc/rt_cull.h"

struct V { float x, y, z; };

static bool ref_box(V o, V d, const rt_geom &g, float &t)          /* scene.c:17-77 */
{
	const float lo[3] = { g.a[0], g.a[1], g.a[2] }, hi[3] = { g.b0, g.b1, g.b2 };
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
	float n[3], f[3];
	for (int k = 0; k < 3; k++) {
		const float a = (lo[k] - oo[k]) / dd[k], b = (hi[k] - oo[k]) / dd[k];
		n[k] = dd[k] >= 0 ? a : b; f[k] = dd[k] >= 0 ? b : a;
	}
	if (n[0] > f[1] || n[1] > f[0]) return false;
	float tn = n[0], tf = f[0];
	if (n[1] > tn) tn = n[1];
	if (f[1] < tf) tf = f[1];
	if (tn > f[2] || n[2] > tf) return false;
	if (n[2] > tn) tn = n[2];
	t = tn;
	return true;
}

static bool ref_sphere(V o, V d, const rt_geom &g, float &t)       /* scene.c:79-134 */
{
	const V oc = { g.a[0] - o.x, g.a[1] - o.y, g.a[2] - o.z };
	const float a = d.x * d.x + d.y * d.y + d.z * d.z;
	const float b = -2.0f * (oc.x * d.x + oc.y * d.y + oc.z * d.z);
	const float c = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - g.b0;
	const float discr = b * b - 4.0f * a * c;
	if (!(discr > 0)) return false;
	const double root = sqrt((double) discr);
	float r0 = (float) (((double) -b + root) / (double) (2.0f * a)), r1 = (float) (((double) -b - root) / (double) (2.0f * a));
	if (r0 > r1) { const float s = r0; r0 = r1; r1 = s; }
	if (r0 < 0) { r0 = r1; if (r0 < 0) return false; }
	t = r0;
	return true;
}

static bool may_touch(V o, V inv, const float lo[3], const float hi[3])     /* slab_may_touch of rt_kernels.hip: plane * inv - o * inv, fused */
{
	const V oi = { o.x * inv.x, o.y * inv.y, o.z * inv.z };
	const float ax = fmaf(lo[0], inv.x, -oi.x), bx = fmaf(hi[0], inv.x, -oi.x);
	const float ay = fmaf(lo[1], inv.y, -oi.y), by = fmaf(hi[1], inv.y, -oi.y);
	const float az = fmaf(lo[2], inv.z, -oi.z), bz = fmaf(hi[2], inv.z, -oi.z);
	const float enter = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
	const float leave = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
	return enter <= leave && leave >= 0.0f;
}

static unsigned long long form_mismatches = 0;       /* the sign form of the grid test against its min / max form (must stay 0) */

/* grid_pair_may_touch of rt_kernels.hip: member j of cluster K on the cluster's grid -- fma(q, step * inv, fma(lo, inv, -o * inv)) */
static bool grid_may_touch(V o, V inv, const rt_cluster &K, int j)
{
	const float clo[3] = { K.lo[0], K.lo[1], K.lo[2] }, chi[3] = { K.hi0, K.hi1, K.hi2 };
	const float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
	float a[3], b[3];
	for (int k = 0; k < 3; k++) {
		const float gs = RT_CLUSTER_STEP(clo[k], chi[k]) * ii[k];
		const float gb = fmaf(clo[k], ii[k], -(oo[k] * ii[k]));
		a[k] = fmaf((float) RT_QPLANE(K.qpair, j, k, 0), gs, gb);
		b[k] = fmaf((float) RT_QPLANE(K.qpair, j, k, 1), gs, gb);
	}
	/* as the kernels decide it (grid_pair_may_touch): the plane the ray meets first by the SIGN of the reciprocal -- not by a min / max --
	 * and max(enter, 0) <= leave; it must be the same answer as the min / max form, for every ray */
	float nr[3], fr[3];
	for (int k = 0; k < 3; k++) { const bool neg = std::signbit(ii[k]); nr[k] = neg ? b[k] : a[k]; fr[k] = neg ? a[k] : b[k]; }
	const float enter = fmaxf(fmaxf(nr[0], nr[1]), nr[2]), leave = fminf(fminf(fr[0], fr[1]), fr[2]);
	const bool by_sign = fmaxf(enter, 0.0f) <= leave;
	const float enter2 = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
	const float leave2 = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
	if (by_sign != (enter2 <= leave2 && leave2 >= 0.0f)) form_mismatches++;
	return by_sign;
}

/* the same for cluster j of group G on the group's grid (round 6: the dealt (ray, group) pairs of nearest_hit_culled, step 1b) */
static bool group_grid_may_touch(V o, V inv, const rt_group &G, int j)
{
	const float clo[3] = { G.lo[0], G.lo[1], G.lo[2] }, chi[3] = { G.hi0, G.hi1, G.hi2 };
	const float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
	float a[3], b[3];
	for (int k = 0; k < 3; k++) {
		const float gs = RT_CLUSTER_STEP(clo[k], chi[k]) * ii[k];
		const float gb = fmaf(clo[k], ii[k], -(oo[k] * ii[k]));
		a[k] = fmaf((float) RT_QPLANE(G.qpair, j, k, 0), gs, gb);
		b[k] = fmaf((float) RT_QPLANE(G.qpair, j, k, 1), gs, gb);
	}
	float nr[3], fr[3];
	for (int k = 0; k < 3; k++) { const bool neg = std::signbit(ii[k]); nr[k] = neg ? b[k] : a[k]; fr[k] = neg ? a[k] : b[k]; }
	const float enter = fmaxf(fmaxf(nr[0], nr[1]), nr[2]), leave = fminf(fminf(fr[0], fr[1]), fr[2]);
	const bool by_sign = fmaxf(enter, 0.0f) <= leave;
	const float enter2 = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
	const float leave2 = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
	if (by_sign != (enter2 <= leave2 && leave2 >= 0.0f)) form_mismatches++;
	return by_sign;
}

int main(int argc, char **argv)
{
	const int scenes = argc > 1 ? atoi(argv[1]) : 20, rays = argc > 2 ? atoi(argv[2]) : 20000;
	std::mt19937_64 rng(12345);
	auto uni = [&](float a, float b) { return a + (b - a) * (float) ((rng() >> 11) * (1.0 / 9007199254740992.0)); };
	unsigned long long pairs = 0, hits = 0, bad = 0, structure_bad = 0, refused = 0;
	for (int sc = 0; sc < scenes; sc++) {
		const int n = 32 + (int) (rng() % 993);
		const float extent = (const float[]) { 0.5f, 3.0f, 10.0f, 30.0f, 60.0f, 250.0f, 3000.0f, 100000.0f }[rng() % 8];      /* (beyond 64: the margin grows with the scene) */
		std::vector<rt_geom> geom((size_t) n);
		for (int i = 0; i < n; i++) {
			rt_geom &g = geom[(size_t) i];
			memset(&g, 0, sizeof(g));
			const float s = extent * (i % 17 == 0 ? 0.5f : 0.03f);           /* a few large objects among many small ones */
			if (i & 1) { g.type = RT_GEOM_SPHERE; for (int k = 0; k < 3; k++) g.a[k] = uni(-extent * 0.9f, extent * 0.9f); const float r = uni(0.02f, 1.0f) * s; g.b0 = r * r; }
			else { g.type = RT_GEOM_CUBE; for (int k = 0; k < 3; k++) g.a[k] = uni(-extent * 0.9f, extent * 0.8f);
			       g.b0 = g.a[0] * 1.0f + uni(0.01f, 1.0f) * s * 1.0f; g.b1 = g.a[1] * 1.0f + uni(0.01f, 1.0f) * s * 1.0f; g.b2 = g.a[2] * 1.0f + uni(0.01f, 1.0f) * s * 1.0f; }
		}
		std::vector<rt_cluster> cl;
		std::vector<rt_group> gr;
		const rt_cull_info info = rt_cull_build(geom, n, cl, gr);
		if (info.num_clusters <= 0) { refused++; continue; }      /* (coordinates beyond RT_CULL_MAX_COORD: none here) */
		/* the groups: every cluster in exactly one, its box inside the group's box and inside its own box on the group's grid */
		if ((int) gr.size() != RT_NUM_GROUPS(info.num_clusters)) structure_bad++;
		for (int c = 0; c < info.num_clusters; c++) {
			const rt_group &G = gr[(size_t) (c / RT_GROUP_SIZE)];
			const int j = c % RT_GROUP_SIZE;
			if (j >= G.count) structure_bad++;
			const float klo[3] = { cl[(size_t) c].lo[0], cl[(size_t) c].lo[1], cl[(size_t) c].lo[2] }, khi[3] = { cl[(size_t) c].hi0, cl[(size_t) c].hi1, cl[(size_t) c].hi2 };
			const float glo[3] = { G.lo[0], G.lo[1], G.lo[2] }, ghi[3] = { G.hi0, G.hi1, G.hi2 };
			for (int k = 0; k < 3; k++) {
				if (klo[k] < glo[k] || khi[k] > ghi[k]) structure_bad++;
				const double step = (double) RT_CLUSTER_STEP(glo[k], ghi[k]);
				if ((double) glo[k] + RT_QPLANE(G.qpair, j, k, 0) * step > (double) klo[k] || (double) glo[k] + RT_QPLANE(G.qpair, j, k, 1) * step < (double) khi[k]) structure_bad++;
			}
		}
		{ int total = 0; for (const rt_group &G : gr) total += G.count; if (total != info.num_clusters) structure_bad++; }
		std::vector<int> owner((size_t) n, -1), place((size_t) n, -1);
		for (int c = 0; c < info.num_clusters; c++)
			for (int j = 0; j < RT_CLUSTER_SIZE; j++) {
				const int i = cl[(size_t) c].member[j];
				if (i == 0xffff) continue;
				if (i >= n || owner[(size_t) i] >= 0) structure_bad++; else { owner[(size_t) i] = c; place[(size_t) i] = j; }
				if (j >= cl[(size_t) c].count) structure_bad++;         /* members fill the first `count` places: the kernel masks the others */
				float lo[3], hi[3];
				rt_cull_object_box(geom[(size_t) i], info.margin, lo, hi);
				const float clo[3] = { cl[(size_t) c].lo[0], cl[(size_t) c].lo[1], cl[(size_t) c].lo[2] }, chi[3] = { cl[(size_t) c].hi0, cl[(size_t) c].hi1, cl[(size_t) c].hi2 };
				for (int k = 0; k < 3; k++) if (lo[k] < clo[k] || hi[k] > chi[k]) structure_bad++;
				/* ... and inside its own box on the cluster's grid (the kernels' float step, the comparison in double) */
				for (int k = 0; k < 3; k++) {
					const double step = (double) RT_CLUSTER_STEP(clo[k], chi[k]);
					if ((double) clo[k] + RT_QPLANE(cl[(size_t) c].qpair, j, k, 0) * step > (double) lo[k] || (double) clo[k] + RT_QPLANE(cl[(size_t) c].qpair, j, k, 1) * step < (double) hi[k]) structure_bad++;
				}
			}
		for (int i = 0; i < n; i++) if (owner[(size_t) i] < 0) structure_bad++;
		for (int r = 0; r < rays; r++) {
			/* origin within the bound the kernel checks; direction: random, or aimed at a feature of a random object (edge, corner,
			 * tangent), optionally nearly axis-aligned */
			V o = { uni(-1, 1) * info.origin_max, uni(-1, 1) * info.origin_max, uni(-1, 1) * info.origin_max };
			if (r % 3 == 0) { const rt_geom &g = geom[rng() % (size_t) n]; o = { g.a[0] + uni(-0.01f, 0.01f), g.a[1] + uni(-0.01f, 0.01f), g.a[2] + uni(-2, 2) }; }
			V d = { uni(-1, 1), uni(-1, 1), uni(-1, 1) };
			if (r % 2 == 0) {
				const rt_geom &g = geom[rng() % (size_t) n];
				V p;
				if (g.type == RT_GEOM_CUBE) p = { (rng() & 1) ? g.a[0] : g.b0, (rng() & 1) ? g.a[1] : g.b1, (rng() & 2) ? uni(g.a[2], g.b2) : ((rng() & 1) ? g.a[2] : g.b2) };
				else { const float rr = sqrtf(g.b0); V u = { uni(-1, 1), uni(-1, 1), uni(-1, 1) }; const float l = sqrtf(u.x * u.x + u.y * u.y + u.z * u.z) + 1e-9f;
				       p = { g.a[0] + u.x / l * rr, g.a[1] + u.y / l * rr, g.a[2] + u.z / l * rr };
				       /* tangent: move the target perpendicular to the line of sight is approximated by a tiny random offset */ }
				const float e = (r % 4 == 0) ? 0.0f : uni(-1, 1) * 1e-4f * (1.0f + fabsf(p.x) + fabsf(p.y) + fabsf(p.z));
				d = { p.x + e - o.x, p.y - e - o.y, p.z + e - o.z };
			}
			if (r % 7 == 0) d.y *= 1e-6f;
			const float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
			if (!(len > 1e-12f)) continue;
			d = { d.x / len, d.y / len, d.z / len };
			const float w[3] = { fabsf(d.x), fabsf(d.y), fabsf(d.z) };
			if (!(w[0] >= 0x1p-30f && w[1] >= 0x1p-30f && w[2] >= 0x1p-30f)) continue;       /* (outside the window the kernel tests every object) */
			const V inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };                               /* rcp_refined == RN(1/d) inside the window */
			{	/* the two forms of the grid test on boxes the ray mostly MISSES as well (form_mismatches): one cluster's members, one group's clusters */
				const int c = (int) ((unsigned) r % (unsigned) info.num_clusters);
				for (int j = 0; j < cl[(size_t) c].count; j++) (void) grid_may_touch(o, inv, cl[(size_t) c], j);
				const rt_group &G = gr[(size_t) (c / RT_GROUP_SIZE)];
				for (int j = 0; j < G.count; j++) (void) group_grid_may_touch(o, inv, G, j);
			}
			for (int i = 0; i < n; i++) {
				float t = 0;
				const bool hit = geom[(size_t) i].type == RT_GEOM_CUBE ? ref_box(o, d, geom[(size_t) i], t) : ref_sphere(o, d, geom[(size_t) i], t);
				pairs++;
				if (!(hit && t >= 0)) continue;
				hits++;
				float lo[3], hi[3];
				rt_cull_object_box(geom[(size_t) i], info.margin, lo, hi);
				const rt_cluster &K = cl[(size_t) owner[(size_t) i]];
				const float clo[3] = { K.lo[0], K.lo[1], K.lo[2] }, chi[3] = { K.hi0, K.hi1, K.hi2 };
				/* the cluster's box, and the member's box as the kernel tests it: quantised on the cluster's grid */
				if (!may_touch(o, inv, clo, chi) || !grid_may_touch(o, inv, K, place[(size_t) i])) bad++;
				/* ... and one level up, as scenes of RT_GROUPS_FROM_CLUSTERS clusters and more are tested: the group's box, then the
				 * cluster's box on the group's grid (which replaces the float test of the cluster's own box there) */
				const int c = owner[(size_t) i];
				const rt_group &G = gr[(size_t) (c / RT_GROUP_SIZE)];
				const float glo[3] = { G.lo[0], G.lo[1], G.lo[2] }, ghi[3] = { G.hi0, G.hi1, G.hi2 };
				if (!may_touch(o, inv, glo, ghi) || !group_grid_may_touch(o, inv, G, c % RT_GROUP_SIZE)) bad++;
				(void) lo; (void) hi;
			}
		}
	}
	printf("%llu %llu %llu %llu %llu\n", pairs, hits, bad, structure_bad + form_mismatches, refused);
	return bad || structure_bad || form_mismatches ? 1 : 0;
}
