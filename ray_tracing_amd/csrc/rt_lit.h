/*
 * rt_lit.h -- which soft-shadow taps need no trace_ray() at all.
 *
 * main.c:186-206: from a hit point P the reference traces up to three taps along normalize(0.5 r + L), r a unit
 * vector, L = origin_of(first emitter) - P, starting at P + 0.001 * direction, and only asks which object each
 * tap hits first.  Every tap of P therefore lies in a cone around L whose half-angle has
 * sin <= 0.5 / (|L| - 0.5).  rt_taps_certainly_lit() decides, conservatively and from P alone, whether EVERY ray of
 * that cone is certain to have the emitter as its nearest hit in the reference's own floating-point tests:
 *   - the emitter is a sphere and every line of the cone passes well inside it (5 % of its radius to spare);
 *   - the whole cone leaves P's own object (cubes and spheres are convex: a ray that starts 0.001 * cos >= 1e-4
 *     above the tangent plane and moves away cannot come back);
 *   - every other object is missed by the cone -- cut off behind the emitter -- by at least 0.01 scene units:
 *     either its bounding box is clear of the cone's bounding box, or its bounding sphere is clear of the cone.
 * The margins are four or more orders of magnitude above the rounding errors of the reference's slab and
 * discriminant tests at the coordinate sizes accepted here (|coordinates| <= 32), so "certainly lit" means the
 * reference's trace_ray() returns the emitter for every such tap; anything doubtful answers 0 and is traced.
 * When it answers 1 for the camera ray's hit point, the taps of bounce 0 of all the pixel's samples are known
 * without being traced (rt_primary_pass sets the flag, the trace kernel honours it): same object index, same
 * emission added in the same order -- bit-identical frames, fewer rays.
 *
 * Plain C99 / HIP, float only, no FMA dependence (every comparison has margins).  scripts/lit_probe.c runs this
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
#define RT_LIT_REFUSE(why) ((void) 0)      /* scripts/lit_probe.c counts the reasons */
#endif
#define RT_LIT_MARGIN 0.01f          /* clearance demanded of every other object, scene units */

/* geom: the packed geometry records of rt_device.h read as floats, 8 words per object
 * (cube: lo.xyz, hi.x | hi.y, hi.z, type, -;  sphere: centre.xyz, r*r | -, -, type, -). */
RT_LIT_FN int rt_taps_certainly_lit(const float *geom, int num_objects, int light, float cx, float cy, float cz,
                                    int hobj, float px, float py, float pz, float nx, float ny, float nz)
{
	if (light < 0 || hobj == light) return 0;
	const float *ge = geom + 8 * light;
	if (((const int *) ge)[6] != 1 /* RT_GEOM_SPHERE */) return 0;
	const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px), __builtin_fabsf(py)), __builtin_fabsf(pz)),
	                                  __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)), __builtin_fabsf(cz)));
	if (!(big <= 32.0f)) { RT_LIT_REFUSE(-1); return 0; }
	const float R = RT_LIT_SQRT(ge[3]);
	const float lx = cx - px, ly = cy - py, lz = cz - pz;
	const float D = RT_LIT_SQRT(lx * lx + ly * ly + lz * lz);
	if (!(D >= R + 0.75f) || !(R >= 0.05f)) { RT_LIT_REFUSE(-2); return 0; }
	const float inv = 1.0f / D;
	const float ax = lx * inv, ay = ly * inv, az = lz * inv;
	const float s = 0.505f / (D - 0.5f);                 /* sin of the cone's half-angle, 1 % over */
	if (!(D * s <= 0.95f * R) || !(s <= 0.7f)) { RT_LIT_REFUSE(-3); return 0; }
	const float cs = RT_LIT_SQRT(1.0f - s * s);
	const float tau = 1.01f * s / cs, icos = 1.01f / cs; /* tan, 1 / cos: 1 % over */
	const float lean = 1.1f * s + 0.1f;                  /* a component of the axis above this: every cone direction has it >= 0.1 */
	if (!(ax * nx + ay * ny + az * nz >= lean)) { RT_LIT_REFUSE(-4); return 0; }
	const float T = 1.01f * (D + R);                     /* axial length of the cone that matters: the emitter ends before it */
	const float m = RT_LIT_MARGIN;
	const float p[3] = { px, py, pz }, a[3] = { ax, ay, az };
	float lo[3], hi[3];
	for (int k = 0; k < 3; k++) {
		const float w = RT_LIT_SQRT(__builtin_fmaxf(0.0f, 1.0f - a[k] * a[k]));
		const float r = T * tau * w + m;
		const float q = p[k] + T * a[k];
		/* the rays start 0.001 * direction away from P: where every direction moves away along this axis by >= 0.1,
		 * they stay >= 1e-4 (minus rounding, < 1e-5 at these coordinate sizes) beyond P */
		lo[k] = __builtin_fminf(a[k] >= lean ? p[k] + 2e-5f : p[k] - m, q - r);
		hi[k] = __builtin_fmaxf(-a[k] >= lean ? p[k] - 2e-5f : p[k] + m, q + r);
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
			rho = RT_LIT_SQRT(g[3]);
			for (int k = 0; k < 3; k++) { q[k] = g[k]; blo[k] = g[k] - 1.001f * rho; bhi[k] = g[k] + 1.001f * rho; }
		} else
			continue;
		if (lo[0] > bhi[0] || hi[0] < blo[0] || lo[1] > bhi[1] || hi[1] < blo[1] || lo[2] > bhi[2] || hi[2] < blo[2])
			continue;                                    /* bounding boxes apart (or the object lies behind the emitter) */
		const float vx = q[0] - px, vy = q[1] - py, vz = q[2] - pz;
		const float along = vx * ax + vy * ay + vz * az;
		const float vv = vx * vx + vy * vy + vz * vz;
		const float rr = 1.001f * rho + m;
		if (along < -rr || along - rr > T) continue;     /* behind P (the cone opens by less than 45 degrees) or behind the emitter */
		const float lim = __builtin_fmaxf(along, 0.0f) * tau + rr * icos;
		if (vv - along * along > lim * lim + 1e-4f * vv + 1e-4f) continue;   /* bounding sphere clear of the cone */
		RT_LIT_REFUSE(i);
		return 0;
	}
	return 1;
}

#endif
