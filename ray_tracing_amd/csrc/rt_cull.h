/*
 * rt_cull.h -- object culling for scenes of many objects ("where a BVH or wave-cooperative culling would start
 * to pay").  Host side: clusters of spatially close objects with conservative bounding boxes, built by rt_set_scene().
 *
 * The reference tests every object for every ray (scene.c:163-173) and keeps the nearest hit, the lowest index among equal
 * distances (strict `<`, scene.c:168).  The kernels keep that answer bit for bit and skip the tests that cannot produce it:
 * an object -- or a whole cluster -- is skipped only when its CONSERVATIVE box is missed by the ray, or lies entirely behind
 * it, in a slab test whose margins exceed every rounding error of the reference's own tests:
 *
 *   - a box is tested as [lo - m, hi + m] with m = RT_CULL_MARGIN.  The reference's quotients (plane - o) / d are correctly
 *     rounded (relative error 2^-24).  The cull forms plane' * r - o * r with r = RN(1/d), one fused multiply-add per plane:
 *     its error is at most 2^-23 |t| + 2^-24 |o| / |d|.  With every coordinate of the scene within S <= 64 and ray origins
 *     within 2 S of the origin, |t| <= 3 S / |d|, so the error stays below 4.8e-7 S / |d| <= 3.1e-5 / |d|, while the inflation
 *     moves an entry / exit parameter by m / |d| = 1.95e-3 / |d|: a margin of 63, on every axis, for every |d| (the ray
 *     directions the cull accepts are those of the shared-reciprocal division: 2^-30 <= |d| <= 2^20 on every axis).
 *     "Entirely behind" is exit' < 0 on the inflated box: then the true exit is negative and the reference rejects the hit
 *     itself (t >= 0, scene.c:168).
 *   - a sphere is tested as the box centre +- h with h = sqrt(r^2 + E) + m.  The reference reports a sphere hit when its FLOAT
 *     discriminant b*b - 4*a*c is positive (scene.c:93-101), which a ray can reach although it passes outside the sphere:
 *     summing the roundings of oc, the two dot products, the squares and the final difference bounds the error of the
 *     discriminant by 108 * 2^-24 * |oc|^2, i.e. r^2 - dist^2 > -1.6e-6 |oc|^2 for every hit the reference reports.  With
 *     |oc| <= 3 sqrt(3) S: E = 2 * 1.6e-6 * 27 S^2 (twice the bound).
 *   Both bounds are proportional to S: a scene that reaches beyond 64 gets the margin m = 2^-9 * P / 64, P the power of two at
 *   or above its S -- the same factor of 63 (and the spheres' E is already in terms of S).  Scenes beyond RT_CULL_MAX_COORD
 *   (2^20: squares stay far from the top of the float range) get no clusters: every object is tested, as before.
 * A wave one of whose rays starts farther out than 2 S -- a camera far outside the scene -- tests every object as well.
 *
 *   - inside a cluster the members are tested as their conservative boxes QUANTISED outwards on the cluster's grid (rt_device.h): the
 *     box only grows (checked here in double: lo + q_lo * step <= member lo, lo + q_hi * step >= member hi with the float step the
 *     kernels form), and the kernels' parameter of a grid plane, fma(q, step * r, fma(lo, r, -o * r)), has an error of at most
 *     2^-24 (|q step r| + |o r| + |b| + |t|) <= 6.0e-7 S / |d| = 3.8e-5 / |d| at S = 64: the margin keeps a factor of 51.
 *
 *   - one level up (round 6, rt_device.h rt_group): groups of eight consecutive clusters, tested as the union of their boxes with the
 *     cluster-box arithmetic, and a group's clusters as their boxes QUANTISED outwards on the group's grid with the member-box
 *     arithmetic: every box of the chain member (inflated) < cluster < quantised cluster < group contains the one before, so a ray
 *     that the reference lets hit a member passes each of them with at least the slack the margin gives the innermost, against the
 *     same error bounds as above (tests/csrc/cull_check.cpp holds every reference hit against the whole chain).
 *
 * Clusters: the leaves of a median-split tree over the objects' centres, RT_CLUSTER_SIZE objects each (members keep their object
 * indices: ties between equal distances go to the lowest INDEX whatever the order of the tests).
 */
#ifndef RT_CULL_H
#define RT_CULL_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "rt_device.h"

#define RT_CULL_MARGIN     0.001953125f     /* 2^-9 */
#define RT_CULL_UNIT_EXTENT 64.0f           /* scenes within this get the margin as it stands; larger ones a proportionally larger one */
#define RT_CULL_MAX_COORD  1048576.0f
#define RT_CULL_MIN_OBJECTS 32              /* below this the every-object loop is faster (profiles/r04/cull_vs_compiled_probe.txt: 24 objects +5 %, 32 -6 %, 64 -22 %);
                                             * a scene of up to 64 objects that the host has compiled (rt_compile_scene) keeps its compiled kernel */

struct rt_cull_info { int num_clusters; float margin; float origin_max; };

/* the conservative box of object i as the kernels form it from the packed record (geom.b1 of a sphere = its h) */
static inline void rt_cull_object_box(const rt_geom &g, float m, float lo[3], float hi[3])
{
	if (g.type == RT_GEOM_CUBE) {
		lo[0] = g.a[0] - m; lo[1] = g.a[1] - m; lo[2] = g.a[2] - m;
		hi[0] = g.b0 + m;   hi[1] = g.b1 + m;   hi[2] = g.b2 + m;
	} else {
		for (int k = 0; k < 3; k++) { lo[k] = g.a[k] - g.b1; hi[k] = g.a[k] + g.b1; }
	}
}


/* box j of an array of box pairs (rt_device.h RT_QPLANE): [blo, bhi] on the grid of [lo, hi] (RT_CLUSTER_GRID steps per axis, the float
 * step the kernels form), rounded OUTWARDS.  Checked in double: lo + q_lo * step <= blo and lo + q_hi * step >= bhi. */
static inline void rt_cull_quantise(const float lo[3], const float hi[3], const float blo[3], const float bhi[3], unsigned char (*qpair)[12], int j)
{
	for (int k = 0; k < 3; k++) {
		const double origin = (double) lo[k], step = (double) RT_CLUSTER_STEP(lo[k], hi[k]);
		int qa = 0, qb = 255;
		if (step > 0.0) {
			qa = (int) floor(((double) blo[k] - origin) / step); qb = (int) ceil(((double) bhi[k] - origin) / step);
			qa = qa < 0 ? 0 : (qa > 255 ? 255 : qa); qb = qb < 0 ? 0 : (qb > 255 ? 255 : qb);
			while (qa > 0 && origin + qa * step > (double) blo[k]) qa--;
			while (qb < 255 && origin + qb * step < (double) bhi[k]) qb++;
		}
		RT_QPLANE(qpair, j, k, 0) = (unsigned char) qa;
		RT_QPLANE(qpair, j, k, 1) = (unsigned char) qb;
	}
}

/* Fills `clusters` and `groups` (and the spheres' geom.b1) when the scene qualifies; returns num_clusters = 0 otherwise. */
static inline rt_cull_info rt_cull_build(std::vector<rt_geom> &geom, int n, std::vector<rt_cluster> &clusters, std::vector<rt_group> &groups)
{
	rt_cull_info info = { 0, RT_CULL_MARGIN, 0.0f };
	clusters.clear();
	groups.clear();
	if (n < RT_CULL_MIN_OBJECTS || n > RT_CLUSTER_SIZE * RT_MAX_CLUSTERS) return info;
	float S = 0.0f;
	for (int i = 0; i < n; i++) {
		const rt_geom &g = geom[(size_t) i];
		if (g.type == RT_GEOM_CUBE) {
			const float v[6] = { g.a[0], g.a[1], g.a[2], g.b0, g.b1, g.b2 };
			for (float x : v) { if (!(fabsf(x) <= RT_CULL_MAX_COORD)) return info; S = std::max(S, fabsf(x)); }
		} else if (g.type == RT_GEOM_SPHERE) {
			const float r = sqrtf(g.b0);
			for (int k = 0; k < 3; k++) { const float x = fabsf(g.a[k]) + r; if (!(x <= RT_CULL_MAX_COORD)) return info; S = std::max(S, x); }
		} else return info;          /* an object of unknown type is never hit (scene.c:153); keep such scenes on the plain path */
	}
	if (!(S > 0.0f)) return info;
	float margin = RT_CULL_MARGIN;
	for (float p = RT_CULL_UNIT_EXTENT; p < S; p *= 2.0f) margin *= 2.0f;      /* (exact: powers of two) */
	info.margin = margin;
	const float E = 2.0f * 1.6e-6f * 27.0f * S * S;
	for (int i = 0; i < n; i++) {
		rt_geom &g = geom[(size_t) i];
		if (g.type == RT_GEOM_SPHERE) g.b1 = sqrtf(g.b0 + E) * 1.0001f + margin;      /* h: see above (b1 of a sphere is otherwise unused) */
	}
	/* Clusters = the leaves of a median-split tree over the objects' centres: a range of objects is cut in two at a multiple of
	 * RT_CLUSTER_SIZE nearest its middle, along the axis on which its centres spread most, until it fits one cluster.  (A Morton
	 * curve of the centres -- round 4's first version -- gives longer, overlapping boxes: a ray of the 1024-object test scene then
	 * touches 12.2 clusters instead of 6.6; profiles/r04/stats_large_1024.txt and stats_large_1024_split.txt.) */
	std::vector<std::pair<uint32_t, int>> order((size_t) n);
	{
		std::vector<float> centre((size_t) n * 3);
		for (int i = 0; i < n; i++) {
			float lo[3], hi[3];
			rt_cull_object_box(geom[(size_t) i], margin, lo, hi);
			for (int k = 0; k < 3; k++) centre[(size_t) i * 3 + k] = 0.5f * (lo[k] + hi[k]);
		}
		std::vector<int> ids((size_t) n);
		for (int i = 0; i < n; i++) ids[(size_t) i] = i;
		struct Range { int first, last; };
		std::vector<Range> todo{ { 0, n } };
		while (!todo.empty()) {
			const Range r = todo.back(); todo.pop_back();
			if (r.last - r.first <= RT_CLUSTER_SIZE) continue;
			float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
			for (int i = r.first; i < r.last; i++)
				for (int k = 0; k < 3; k++) { const float c = centre[(size_t) ids[(size_t) i] * 3 + k]; lo[k] = std::min(lo[k], c); hi[k] = std::max(hi[k], c); }
			int axis = 0;
			for (int k = 1; k < 3; k++) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
			const int clusters = (r.last - r.first + RT_CLUSTER_SIZE - 1) / RT_CLUSTER_SIZE;
			const int mid = r.first + (clusters / 2) * RT_CLUSTER_SIZE;          /* whole clusters on the left: only the very last cluster may be short */
			std::nth_element(ids.begin() + r.first, ids.begin() + mid, ids.begin() + r.last,
			                 [&](int a, int b) { const float ca = centre[(size_t) a * 3 + axis], cb = centre[(size_t) b * 3 + axis]; return ca < cb || (ca == cb && a < b); });
			todo.push_back({ r.first, mid }); todo.push_back({ mid, r.last });
		}
		for (int i = 0; i < n; i++) order[(size_t) i] = { (uint32_t) i, ids[(size_t) i] };
	}
	const int C = (n + RT_CLUSTER_SIZE - 1) / RT_CLUSTER_SIZE;
	clusters.assign((size_t) C, rt_cluster());
	for (int c = 0; c < C; c++) {
		rt_cluster &K = clusters[(size_t) c];
		memset(&K, 0, sizeof(K));
		float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
		int members[RT_CLUSTER_SIZE], cnt = 0;
		for (int j = 0; j < RT_CLUSTER_SIZE && c * RT_CLUSTER_SIZE + j < n; j++) members[cnt++] = order[(size_t) (c * RT_CLUSTER_SIZE + j)].second;
		std::sort(members, members + cnt);          /* (index order inside a cluster: not needed for the answer, tidy for reading) */
		for (int j = 0; j < RT_CLUSTER_SIZE; j++) K.member[j] = 0xffff;
		for (int j = 0; j < cnt; j++) {
			float blo[3], bhi[3];
			rt_cull_object_box(geom[(size_t) members[j]], margin, blo, bhi);
			for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[k]); hi[k] = std::max(hi[k], bhi[k]); }
			K.member[j] = (unsigned short) members[j];
		}
		K.lo[0] = lo[0]; K.lo[1] = lo[1]; K.lo[2] = lo[2]; K.hi0 = hi[0]; K.hi1 = hi[1]; K.hi2 = hi[2];
		K.count = cnt;
		/* the members' boxes on the cluster's grid, rounded outwards */
		for (int j = 0; j < cnt; j++) {
			float blo[3], bhi[3];
			rt_cull_object_box(geom[(size_t) members[j]], margin, blo, bhi);
			rt_cull_quantise(lo, hi, blo, bhi, K.qpair, j);
		}
	}
	/* groups of RT_GROUP_SIZE consecutive clusters (the clusters are in the order of the split tree: neighbours in it are neighbours
	 * in space): the union of their boxes, and their boxes on the group's grid, rounded outwards */
	const int Gn = RT_NUM_GROUPS(C);
	groups.assign((size_t) Gn, rt_group());
	for (int gi = 0; gi < Gn; gi++) {
		rt_group &Gr = groups[(size_t) gi];
		memset(&Gr, 0, sizeof(Gr));
		const int first = gi * RT_GROUP_SIZE, cnt = std::min(RT_GROUP_SIZE, C - first);
		float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
		for (int j = 0; j < cnt; j++) {
			const rt_cluster &K = clusters[(size_t) (first + j)];
			const float klo[3] = { K.lo[0], K.lo[1], K.lo[2] }, khi[3] = { K.hi0, K.hi1, K.hi2 };
			for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], klo[k]); hi[k] = std::max(hi[k], khi[k]); }
		}
		Gr.lo[0] = lo[0]; Gr.lo[1] = lo[1]; Gr.lo[2] = lo[2]; Gr.hi0 = hi[0]; Gr.hi1 = hi[1]; Gr.hi2 = hi[2];
		Gr.count = cnt;
		for (int j = 0; j < cnt; j++) {
			const rt_cluster &K = clusters[(size_t) (first + j)];
			const float klo[3] = { K.lo[0], K.lo[1], K.lo[2] }, khi[3] = { K.hi0, K.hi1, K.hi2 };
			rt_cull_quantise(lo, hi, klo, khi, Gr.qpair, j);
		}
	}
	info.num_clusters = C;
	info.origin_max = 2.0f * S;
	return info;
}

#endif
