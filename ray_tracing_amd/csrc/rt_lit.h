/*
 * rt_lit.h -- which soft-shadow taps need no trace_ray() at all.
 *
 * main.c:186-206: from a hit point P the reference traces up to three taps along normalize(0.5 r + L), r a unit
 * vector, L = origin_of(first emitter) - P, starting at P + 0.001 * direction, and only asks which object each
 * tap hits first.  Every tap of P therefore lies in a cone around L whose half-angle has
 * sin <= 0.5 / (|L| - 0.5).  rt_taps_certainly_lit() decides, conservatively and from P alone, whether EVERY ray of
 * that cone is certain to have the emitter as its nearest hit in the reference's own floating-point tests:
 *   - the emitter is a sphere and every line of the cone passes well inside it (5 % of its radius to spare);
 *   - every ACCEPTED tap (main.c:194: the random unit vector has a positive component along the normal) leaves P's own
 *     object (cubes and spheres are convex: a ray that starts 0.001 * cos >= 1e-4 above the tangent plane and moves away
 *     cannot come back): the direction to the emitter leans >= 0.1 (1 + 0.5 / |L|) beyond the tangent plane;
 *   - every other object is missed by the cone -- cut off behind the emitter -- by at least 0.01 scene units:
 *     either its bounding box is clear of the cone's bounding box, or its bounding sphere is clear of the cone.
 * The clearances are two or more orders of magnitude above the rounding errors of the reference's slab and
 * discriminant tests at the coordinate sizes accepted here (|coordinates| <= 32, emitter within 50 radii); the one
 * assumption that is not a clearance -- that the hit point lies on its object's surface -- is measured for every point
 * (rt_lit_point_on_surface).  "Certainly lit" then means the reference's trace_ray() returns the emitter for every
 * such tap; anything doubtful answers 0 and is traced.
 * When it answers 1 for the camera ray's hit point, the taps of bounce 0 of all the pixel's samples are known
 * without being traced (rt_primary_pass sets the flag, the trace kernel honours it): same object index, same
 * emission added in the same order -- bit-identical frames, fewer rays.
 *
 * Plain C99 / HIP, float only, no FMA dependence (every comparison has margins).  tests/lit_probe.c runs this
 * very function on the CPU at every shading point of a frame and checks each answer against the oracle's trace;
 * tests/test_gpu_parity.py compares frames with the flag honoured and ignored (rt_tuning.trace_known_taps).
 */
#ifndef RT_LIT_H
#define RT_LIT_H

#ifndef RT_LIT_FN
#define RT_LIT_FN static inline
#endif
#ifndef RT_LIT_SQRT
#define RT_LIT_SQRT(x) __builtin_sqrtf(x)
#endif

#ifndef RT_LIT_REFUSE
#define RT_LIT_REFUSE(why) ((void) 0)      /* tests/lit_probe.c counts the reasons */
#endif
#define RT_LIT_MARGIN 0.01f          /* clearance demanded of every other object, scene units */

/* geom: the packed geometry records of rt_device.h read as floats, 8 words per object
 * (cube: lo.xyz, hi.x | hi.y, hi.z, type, -;  sphere: centre.xyz, r*r | -, -, type, -).
 *
 * rt_region_certainly_lit: the statement for EVERY point P of the box centre +- half (on object `hobj`, with an outward
 * normal within `nslack` of n there), so that one answer can serve a whole cell of a table.  A tap from P' = P + d
 * is the tap from P shifted by d: the region's extent is added to every clearance, the direction to the emitter
 * wobbles by at most |d| / distance, which widens the cone. */
RT_LIT_FN int rt_region_certainly_lit(const float *geom, int num_objects, int light, float cx, float cy, float cz, int hobj,
                                      float px, float py, float pz, float hx, float hy, float hz, float nx, float ny, float nz, float nslack)
{
	if (light < 0 || hobj == light) return 0;
	const float *ge = geom + 8 * light;
	if (((const int *) ge)[6] != 1 /* RT_GEOM_SPHERE */) return 0;
	const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px) + hx, __builtin_fabsf(py) + hy), __builtin_fabsf(pz) + hz),
	                                  __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)), __builtin_fabsf(cz)));
	if (!(big <= 32.0f)) { RT_LIT_REFUSE(-1); return 0; }
	const float sl = 1.001f * RT_LIT_SQRT(hx * hx + hy * hy + hz * hz);     /* no point of the region is farther from its centre */
	const float R = RT_LIT_SQRT(ge[3]);
	const float lx = cx - px, ly = cy - py, lz = cz - pz;
	const float D = RT_LIT_SQRT(lx * lx + ly * ly + lz * lz);
	const float Dmin = D - sl, Dmax = D + sl;
	/* (the emitter's own discriminant -- b*b - 4ac with |c - o| = D -- carries an error of a few 1e-6 D^2 against the
	 * >= 0.39 R^2 the 5 % margin below leaves it: D <= 50 R keeps two orders of magnitude between them) */
	if (!(Dmin >= R + 0.75f) || !(R >= 0.05f) || !(Dmax <= 50.0f * R)) { RT_LIT_REFUSE(-2); return 0; }
	const float inv = 1.0f / D;
	const float ax = lx * inv, ay = ly * inv, az = lz * inv;
	const float s1 = 0.505f / (Dmin - 0.5f);             /* sin of the half-angle of one point's cone, 1 % over */
	if (!(Dmax * s1 <= 0.95f * R)) { RT_LIT_REFUSE(-3); return 0; }
	const float s = s1 + 1.05f * sl / Dmin;              /* ... of the cone around the centre's axis that holds every point's cone */
	if (!(s <= 0.7f)) { RT_LIT_REFUSE(-3); return 0; }
	const float cs = RT_LIT_SQRT(1.0f - s * s);
	const float tau = 1.01f * s / cs, icos = 1.01f / cs; /* tan, 1 / cos: 1 % over */
	const float lean = 1.1f * s + 0.1f;                  /* a component of the axis above this: every cone direction has it >= 0.1 */
	/* The taps must leave P's own (convex) object: their origin is moved 0.001 along the tap direction (main.c:198), which
	 * has to amount to >= 1e-4 along the normal.  Only ACCEPTED taps are ever asked about -- dot(r, n) > 0 for the unit
	 * vector r (main.c:194) -- and their direction is (0.5 r + L) / |0.5 r + L|: its normal component is above
	 * (L.n) / (|L| + 0.5), so L^.n >= 0.1 (1 + 0.5 / |L|) is enough, whatever the cone's width (2 % and the region's wobble
	 * of the direction to the emitter on top) */
	const float leave = 0.102f * (1.0f + 0.5f / Dmin) + 1.05f * sl / Dmin + 1e-4f;
	if (!(ax * nx + ay * ny + az * nz - nslack >= leave)) { RT_LIT_REFUSE(-4); return 0; }
	const float T = 1.01f * (Dmax + R);                  /* axial length of the cone that matters: the emitter ends before it */
	const float m = RT_LIT_MARGIN;
	const float p[3] = { px, py, pz }, a[3] = { ax, ay, az }, h[3] = { hx, hy, hz };
	float lo[3], hi[3];
	for (int k = 0; k < 3; k++) {
		const float w = RT_LIT_SQRT(__builtin_fmaxf(0.0f, 1.0f - a[k] * a[k]));
		const float r = T * tau * w + m + h[k];
		const float q = p[k] + T * a[k];
		/* the rays start 0.001 * direction away from P: where every direction moves away along this axis by >= 0.1,
		 * they stay >= 1e-4 (minus rounding, < 1e-5 at these coordinate sizes) beyond P */
		lo[k] = __builtin_fminf(a[k] >= lean ? p[k] - h[k] + 2e-5f : p[k] - h[k] - m, q - r);
		hi[k] = __builtin_fmaxf(-a[k] >= lean ? p[k] + h[k] - 2e-5f : p[k] + h[k] + m, q + r);
	}
	for (int i = 0; i < num_objects; i++) {
		if (i == light || i == hobj) continue;
		const float *g = geom + 8 * i;
		const int type = ((const int *) g)[6];
		float blo[3], bhi[3], q[3], rho;
		if (type == 0 /* RT_GEOM_CUBE */) {
			blo[0] = g[0]; blo[1] = g[1]; blo[2] = g[2]; bhi[0] = g[3]; bhi[1] = g[4]; bhi[2] = g[5];
			const float ex = bhi[0] - blo[0], ey = bhi[1] - blo[1], ez = bhi[2] - blo[2];
			q[0] = blo[0] + 0.5f * ex; q[1] = blo[1] + 0.5f * ey; q[2] = blo[2] + 0.5f * ez;
			rho = 0.5f * RT_LIT_SQRT(ex * ex + ey * ey + ez * ez);
		} else if (type == 1) {
			/* a sphere is "missed" by the reference when its float discriminant b*b - 4ac is not positive, and that
			 * carries an error of about 2.4e-6 |centre - origin|^2: the sphere is taken larger by enough that a line
			 * clear of it has a discriminant below -8e-4 of that square */
			const float wx = g[0] - px, wy = g[1] - py, wz = g[2] - pz;
			const float far = RT_LIT_SQRT(wx * wx + wy * wy + wz * wz) + sl;
			rho = RT_LIT_SQRT(g[3] + 2e-4f * far * far + 1e-4f);
			for (int k = 0; k < 3; k++) { q[k] = g[k]; blo[k] = g[k] - 1.001f * rho; bhi[k] = g[k] + 1.001f * rho; }
		} else
			continue;
		if (lo[0] > bhi[0] || hi[0] < blo[0] || lo[1] > bhi[1] || hi[1] < blo[1] || lo[2] > bhi[2] || hi[2] < blo[2])
			continue;                                    /* bounding boxes apart (or the object lies behind the emitter) */
		const float vx = q[0] - px, vy = q[1] - py, vz = q[2] - pz;
		const float along = vx * ax + vy * ay + vz * az;
		const float vv = vx * vx + vy * vy + vz * vz;
		const float rr = 1.001f * rho + m + sl;
		if (along < -rr || along - rr > T) continue;     /* behind P (the cone opens by less than 45 degrees) or behind the emitter */
		const float lim = __builtin_fmaxf(along, 0.0f) * tau + rr * icos;
		if (vv - along * along > lim * lim + 1e-4f * vv + 1e-4f) continue;   /* bounding sphere clear of the cone */
		RT_LIT_REFUSE(i);
		return 0;
	}
	return 1;
}

/* The opposite certainty, for scenes in which the emitter is the ONLY object whose emission is not all zeros (the caller
 * checks: then a tap adds something only if the emitter is its nearest hit, main.c:200-204): every accepted tap from every
 * point of the region certainly has some OTHER object as its nearest hit, so it adds nothing and need not be traced.
 * Two cases, both about spheres:
 *   - the hit point lies on the far side of its own sphere as seen from the emitter: the direction of every accepted tap
 *     (0.5 r + L, dot(r, n) > 0) has a normal component of at most (0.5 + L.n) / (|L| + 0.5) <= -0.1, the tap starts
 *     >= 1e-4 INSIDE the sphere and the reference finds the sphere's far side (one root negative, one positive, the
 *     constant term of its quadratic a few 1e-4 r below zero against a rounding error of 2.4e-6 r^2) -- or something nearer,
 *     which is not the emitter either: that lies outside the sphere, with a gap;
 *   - another sphere stands in the cone's way: every line of the cone passes within 93 % of its radius of its centre
 *     (a discriminant of >= 0.5 r^2 against an error of 2.4e-6 (distance + r)^2 at up to 50 radii), the sphere is in front
 *     of every tap's origin and ends before the emitter begins. */
RT_LIT_FN int rt_region_certainly_dark(const float *geom, int num_objects, int light, float cx, float cy, float cz, int hobj,
                                       float px, float py, float pz, float hx, float hy, float hz, float nx, float ny, float nz, float nslack)
{
	if (light < 0 || hobj == light || hobj < 0 || hobj >= num_objects) return 0;
	const float *ge = geom + 8 * light;
	if (((const int *) ge)[6] != 1 /* RT_GEOM_SPHERE */) return 0;
	const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px) + hx, __builtin_fabsf(py) + hy), __builtin_fabsf(pz) + hz),
	                                  __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)), __builtin_fabsf(cz)));
	if (!(big <= 32.0f)) return 0;
	const float sl = 1.001f * RT_LIT_SQRT(hx * hx + hy * hy + hz * hz) + 0.001f;     /* the region's radius and the 0.001 a tap's origin is moved */
	const float R = RT_LIT_SQRT(ge[3]);
	const float lx = cx - px, ly = cy - py, lz = cz - pz;
	const float D = RT_LIT_SQRT(lx * lx + ly * ly + lz * lz);
	const float Dmin = D - sl, Dmax = D + sl;
	if (!(Dmin >= R + 0.75f) || !(R >= 0.05f) || !(Dmax <= 50.0f * R)) return 0;
	const float inv = 1.0f / D;
	const float ax = lx * inv, ay = ly * inv, az = lz * inv;
	const float s = 0.505f / (Dmin - 0.5f) + 1.05f * sl / Dmin;      /* sin of the half-angle of the cone that holds every point's taps */
	if (!(s <= 0.7f)) return 0;
	const float cs = RT_LIT_SQRT(1.0f - s * s);
	const float *gh = geom + 8 * hobj;
	if (((const int *) gh)[6] == 1) {                    /* the own sphere's far side */
		const float rho = RT_LIT_SQRT(gh[3]);
		const float ex = cx - gh[0], ey = cy - gh[1], ez = cz - gh[2];
		if (rho >= 0.05f && rho <= 16.0f && RT_LIT_SQRT(ex * ex + ey * ey + ez * ez) >= rho + R + 0.1f &&
		    ax * nx + ay * ny + az * nz + 1.05f * sl / Dmin + nslack <= -(0.56f / Dmin + 0.102f))
			return 1;
	}
	for (int i = 0; i < num_objects; i++) {              /* another sphere in the way of the whole cone */
		if (i == light || i == hobj) continue;
		const float *g = geom + 8 * i;
		if (((const int *) g)[6] != 1) continue;
		const float rho = RT_LIT_SQRT(g[3]);
		const float wx = g[0] - px, wy = g[1] - py, wz = g[2] - pz;
		const float dist = RT_LIT_SQRT(wx * wx + wy * wy + wz * wz);
		if (!(rho >= 0.05f) || !(dist - sl >= rho + 0.01f) || !(dist + sl <= 50.0f * rho)) continue;   /* origins outside it, not too far */
		if (!(dist + sl + rho <= Dmin - R - 0.1f)) continue;                                          /* it ends before the emitter begins */
		const float cphi = (wx * ax + wy * ay + wz * az) / dist;
		if (!(cphi > 0.0f)) continue;
		const float sphi = RT_LIT_SQRT(__builtin_fmaxf(0.0f, 1.0f - cphi * cphi));
		const float spsi = 1.05f * sl / (dist - sl);                 /* the direction to its centre wobbles by this much over the region */
		if (!(cphi * cs - sphi * s >= 0.2f) || !(spsi <= 0.1f)) continue;   /* the angles below add up to less than a right angle */
		if ((sphi * cs + cphi * s + spsi) * 1.01f * (dist + sl) <= 0.93f * rho) return 1;
	}
	return 0;
}

/* Is the hit point a ray reported really on the object's surface, to within the 1e-4 by which a tap's origin is moved
 * off it (main.c:198 with a direction that leans >= 0.1 away)?  trace_ray()'s hit point is origin + direction * t in
 * float: on a cube face it is off the plane by a few ulps of the ray's extent, but a sphere's t comes out of a
 * discriminant whose error grows with the SQUARE of the distance to the ray's origin -- 2e-4 inside a sphere of radius
 * 2.25 hit from 40 units away (found by tests/lit_fuzz.py at scale 4.5): a tap from there starts INSIDE and hits the
 * sphere itself.  So the deviation is measured, not assumed: g = the 8 words of the hit object, n its normal at P. */
RT_LIT_FN int rt_lit_point_on_surface(const float *g, float px, float py, float pz, float nx, float ny, float nz)
{
	if (((const int *) g)[6] == 1) {                     /* sphere: |P - centre|^2 - r^2 = 2 r x (radial offset), at most 5e-5 either way */
		const float vx = px - g[0], vy = py - g[1], vz = pz - g[2];
		const float dev2 = vx * vx + vy * vy + vz * vz - g[3];
		return dev2 * dev2 <= 1e-8f * g[3];              /* (outside matters too: a tap that is to start INSIDE the sphere, rt_region_certainly_dark) */
	}
	/* cube: the face is the one the normal names; P within 2e-5 of its plane */
	const float off = nx != 0.0f ? px - (nx > 0.0f ? g[3] : g[0]) : (ny != 0.0f ? py - (ny > 0.0f ? g[4] : g[1]) : pz - (nz > 0.0f ? g[5] : g[2]));
	return __builtin_fabsf(off) <= 2e-5f;
}

/* one point, both questions: 1 = every accepted tap certainly reaches the emitter first, 2 = certainly none does (asked only
 * when the caller knows that the emitter alone emits), 0 = trace them */
RT_LIT_FN int rt_taps_certainly_lit(const float *geom, int num_objects, int light, float cx, float cy, float cz,
                                    int hobj, float px, float py, float pz, float nx, float ny, float nz);
RT_LIT_FN int rt_taps_class(const float *geom, int num_objects, int light, float cx, float cy, float cz, int only_light_emits,
                            int hobj, float px, float py, float pz, float nx, float ny, float nz)
{
	if (rt_taps_certainly_lit(geom, num_objects, light, cx, cy, cz, hobj, px, py, pz, nx, ny, nz)) return 1;
	if (!only_light_emits || hobj < 0 || hobj >= num_objects || !rt_lit_point_on_surface(geom + 8 * hobj, px, py, pz, nx, ny, nz)) return 0;
	return rt_region_certainly_dark(geom, num_objects, light, cx, cy, cz, hobj, px, py, pz, 0.0f, 0.0f, 0.0f, nx, ny, nz, 0.0f) ? 2 : 0;
}

/* one point: the hit point of a ray, with the normal trace_ray() reports there */
RT_LIT_FN int rt_taps_certainly_lit(const float *geom, int num_objects, int light, float cx, float cy, float cz,
                                    int hobj, float px, float py, float pz, float nx, float ny, float nz)
{
	if (hobj < 0 || hobj >= num_objects || !rt_lit_point_on_surface(geom + 8 * hobj, px, py, pz, nx, ny, nz)) return 0;
	return rt_region_certainly_lit(geom, num_objects, light, cx, cy, cz, hobj, px, py, pz, 0.0f, 0.0f, 0.0f, nx, ny, nz, 0.0f);
}

/* ---- a table of answers for hit points of any bounce --------------------------------------------------------
 * Every object gets a grid over its bounding box (cells of about `cell` scene units, at least two per axis so that
 * opposite faces of a thin slab never share a cell); bit base + (iz * res[1] + iy) * res[0] + ix of the table says:
 * every point of the object's SURFACE inside that cell, with the normal the surface has there, is a point whose taps
 * certainly reach the emitter (rt_region_certainly_lit over the cell, per cube face that touches it).  Built once per
 * scene on the host (rt_set_scene); the trace kernel turns a hit point into its cell with three multiplies and reads
 * one entry.  Cells are inflated by a thousandth of their size before they are classified, so that a hit point whose
 * index rounds into the neighbouring cell is still covered.  An entry speaks for points ON the surface: whoever reads
 * it for a hit point checks rt_lit_point_on_surface() for that point first. */
typedef struct { float lo[3]; float scale[3]; int res[3]; int base; int pad[2]; } rt_lit_grid;   /* 48 bytes */

#define RT_LIT_MAX_RES 256

RT_LIT_FN void rt_lit_object_box(const float *g, float blo[3], float bhi[3])
{
	if (((const int *) g)[6] == 0) { blo[0] = g[0]; blo[1] = g[1]; blo[2] = g[2]; bhi[0] = g[3]; bhi[1] = g[4]; bhi[2] = g[5]; }
	else { const float rho = 1.001f * RT_LIT_SQRT(g[3]); for (int k = 0; k < 3; k++) { blo[k] = g[k] - rho; bhi[k] = g[k] + rho; } }
}

/* grids for all objects; returns the number of table bits (0: nothing to look up -- no sphere emitter) */
RT_LIT_FN long long rt_lit_layout(const float *geom, int num_objects, int light, float cell, rt_lit_grid *grids)
{
	if (light < 0 || ((const int *) (geom + 8 * light))[6] != 1) return 0;
	long long bits = 0;
	for (int i = 0; i < num_objects; i++) {
		float blo[3], bhi[3];
		rt_lit_object_box(geom + 8 * i, blo, bhi);
		rt_lit_grid *G = &grids[i];
		long long cells = 1;
		for (int k = 0; k < 3; k++) {
			const float size = bhi[k] - blo[k];
			int r = size > 0.0f && size < 1e6f ? (int) (size / cell) + 1 : 2;
			r = r < 2 ? 2 : (r > RT_LIT_MAX_RES ? RT_LIT_MAX_RES : r);
			G->lo[k] = blo[k]; G->res[k] = r;
			G->scale[k] = size > 0.0f ? (float) r / size : 0.0f;
			cells *= r;
		}
		G->base = (int) bits; G->pad[0] = G->pad[1] = 0;
		bits += i == light ? 0 : cells;
		if (bits > 0x7fffff00ll) return 0;
	}
	return bits;
}

RT_LIT_FN int rt_lit_clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* the table bit of a hit point on object G (float -> index exactly as the kernel computes it) */
RT_LIT_FN int rt_lit_bit_of(const rt_lit_grid *G, float px, float py, float pz)
{
	const int ix = rt_lit_clampi((int) __builtin_floorf((px - G->lo[0]) * G->scale[0]), G->res[0] - 1);
	const int iy = rt_lit_clampi((int) __builtin_floorf((py - G->lo[1]) * G->scale[1]), G->res[1] - 1);
	const int iz = rt_lit_clampi((int) __builtin_floorf((pz - G->lo[2]) * G->scale[2]), G->res[2] - 1);
	return G->base + (iz * G->res[1] + iy) * G->res[0] + ix;
}

RT_LIT_FN int rt_lit_cell_certainly(int dark, const float *geom, int num_objects, int light, float cx, float cy, float cz,
                                    int i, const rt_lit_grid *G, int ix, int iy, int iz)
{
	const float *g = geom + 8 * i;
	const int idx[3] = { ix, iy, iz };
	float c[3], h[3];
	for (int k = 0; k < 3; k++) {
		if (!(G->scale[k] > 0.0f)) return 0;
		const float w = 1.0f / G->scale[k];
		c[k] = G->lo[k] + ((float) idx[k] + 0.5f) * w;
		h[k] = 0.501f * w + 1e-5f;                   /* the cell, a little larger (index rounding at its borders) */
	}
	if (((const int *) g)[6] == 0) {                     /* cube: the faces that touch the cell, each with its own normal */
		int faces = 0;
		for (int k = 0; k < 3; k++)
			for (int side = 0; side < 2; side++) {
				if (idx[k] != (side ? G->res[k] - 1 : 0)) continue;
				float p[3] = { c[0], c[1], c[2] }, e[3] = { h[0], h[1], h[2] }, n[3] = { 0.0f, 0.0f, 0.0f };
				p[k] = side ? g[3 + k] : g[k];           /* the face's plane; hit points lie within rounding of it */
				e[k] = 1e-5f;
				n[k] = side ? 1.0f : -1.0f;
				if (!(dark ? rt_region_certainly_dark(geom, num_objects, light, cx, cy, cz, i, p[0], p[1], p[2], e[0], e[1], e[2], n[0], n[1], n[2], 0.0f)
				           : rt_region_certainly_lit(geom, num_objects, light, cx, cy, cz, i, p[0], p[1], p[2], e[0], e[1], e[2], n[0], n[1], n[2], 0.0f))) return 0;
				faces++;
			}
		return faces > 0;
	}
	/* sphere: cells the surface passes through; the normal at the cell's centre, off by at most 2 |cell| / radius elsewhere */
	const float rho = RT_LIT_SQRT(g[3]);
	const float vx = c[0] - g[0], vy = c[1] - g[1], vz = c[2] - g[2];
	const float d = RT_LIT_SQRT(vx * vx + vy * vy + vz * vz);
	const float sl = 1.001f * RT_LIT_SQRT(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
	if (!(d > 1e-3f * rho) || d - sl > rho || d + sl < rho) return 0;
	const float inv = 1.0f / d;
	return dark ? rt_region_certainly_dark(geom, num_objects, light, cx, cy, cz, i, c[0], c[1], c[2], h[0], h[1], h[2], vx * inv, vy * inv, vz * inv, 2.1f * sl / rho)
	            : rt_region_certainly_lit(geom, num_objects, light, cx, cy, cz, i, c[0], c[1], c[2], h[0], h[1], h[2], vx * inv, vy * inv, vz * inv, 2.1f * sl / rho);
}
RT_LIT_FN int rt_lit_cell_certainly_lit(const float *geom, int num_objects, int light, float cx, float cy, float cz,
                                        int i, const rt_lit_grid *G, int ix, int iy, int iz)
{
	return rt_lit_cell_certainly(0, geom, num_objects, light, cx, cy, cz, i, G, ix, iy, iz);
}

/* the whole table (host side, once per scene); `words` must hold (bits + 31) / 32 words; `dark_words` likewise, or NULL when
 * other objects emit too (then only "certainly lit" is a usable answer) */
RT_LIT_FN void rt_lit_build(const float *geom, int num_objects, int light, float cx, float cy, float cz,
                            const rt_lit_grid *grids, unsigned int *words, unsigned int *dark_words, long long bits)
{
	for (long long w = 0; w < (bits + 31) / 32; w++) { words[w] = 0u; if (dark_words) dark_words[w] = 0u; }
	for (int i = 0; i < num_objects; i++) {
		if (i == light) continue;
		const rt_lit_grid *G = &grids[i];
		for (int iz = 0; iz < G->res[2]; iz++)
			for (int iy = 0; iy < G->res[1]; iy++)
				for (int ix = 0; ix < G->res[0]; ix++)
					{
						const long long b = G->base + ((long long) iz * G->res[1] + iy) * G->res[0] + ix;
						if (rt_lit_cell_certainly(0, geom, num_objects, light, cx, cy, cz, i, G, ix, iy, iz)) words[b >> 5] |= 1u << (b & 31);
						else if (dark_words && rt_lit_cell_certainly(1, geom, num_objects, light, cx, cy, cz, i, G, ix, iy, iz)) dark_words[b >> 5] |= 1u << (b & 31);
					}
	}
}

#endif
