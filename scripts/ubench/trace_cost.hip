// Micro-benchmark: cost of the building blocks of the scene-specialised trace kernel, in shader cycles per call
// per wave, under the kernel's own conditions (4 waves per SIMD, all CUs busy, scene_0 compiled in, random rays).
// The functions are the kernel's own (rt_kernels.hip is included).  Development aid; output kept under profiles/.
//
// build (from the repo root; rt_scene_spec.h = scripts/spec_asm.py's header for data/scene_0.txt):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -DRT_SPEC_ONLY \
//         -DRT_SPEC_HEADER='"rt_scene_spec.h"' -I$TMPDIR/spec -Iray_tracing_amd/csrc -o $TMPDIR/trace_cost scripts/ubench/trace_cost.hip
#include "rt_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

struct Stamp { unsigned long long ticks, real; };

RT_DEV uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
RT_DEV float frand(uint32_t &s) { return (float) (lcg(s) >> 8) * 0x1p-24f; }

template <int WHICH>
__global__ void __launch_bounds__(256, 4) k(float *out, Stamp *stamps, int iters, rt_launch L)
{
	extern __shared__ float4 lds[];
	const SceneLDS sc = stage_scene(L, lds);
	uint32_t s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
	float acc = 0.0f;
	uint64_t rng = s;
	__syncthreads();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int i = 0; i < iters; i++) {
		// a ray inside scene_0's room, any direction
		const V3 o = mk3(0.5f + 8.0f * frand(s), 0.2f + 4.0f * frand(s), 0.3f + 8.0f * frand(s));
		const V3 d = mk3(frand(s) * 2.0f - 1.0f, frand(s) * 2.0f - 1.0f, frand(s) * 2.0f - 1.0f);
		if (WHICH == 0) { acc += o.x + d.y; }                                              // the ray generation alone
		if (WHICH == 1) { const V3 dn = unit3_fast(d); acc += dn.x + o.x; }
		if (WHICH == 2) { const V3 dn = unit3_fast(d); const Hit h = nearest_hit_spec(sc, L.num_objects, o, dn, true);  acc += h.t + h.n.x + (float) h.obj; }
		if (WHICH == 3) { const V3 dn = unit3_fast(d); const Hit h = nearest_hit_spec(sc, L.num_objects, o, dn, false); acc += h.t + (float) h.obj; }
		if (WHICH == 4) { const V3 dn = unit3_fast(d); const Hit h = nearest_hit_fast(sc, L.num_objects, o, dn, true);  acc += h.t + h.n.x + (float) h.obj; }
		if (WHICH == 5) { acc += rng_draw(rng) + o.x; }
		if (WHICH == 6) { const V3 v = rng_direction<true>(rng); acc += v.x + o.x; }
		if (WHICH == 7) { const V3 dn = unit3_fast(d); acc += __uint_as_float(sky_texel<true>(L, dn) & 0x3f000000u) + o.x; }
		if (WHICH == 8) { const V3 dn = unit3_fast(d); const RayPrep rp = prepare_ray<true>(o, dn); acc += rp.inv.x + rp.inv.y + rp.inv.z + (float) rp.rden + (rp.inv_ok ? 1.0f : 0.0f); }
		if (WHICH == 9) { const V3 dn = unit3_fast(d); const RayPrep rp = prepare_ray<true>(o, dn); float t = 0; bool h = false;
		                  for (int sp = 0; sp < SPEC_N; sp++) if (SPEC_T[sp] == RT_GEOM_SPHERE) { float ts; if (ball_entry_fast(o, dn, rp, mk3(SPEC_G[sp][0], SPEC_G[sp][1], SPEC_G[sp][2]), SPEC_G[sp][3], ts)) { h = true; t += ts; } }
		                  acc += t + (h ? 1.0f : 0.0f); }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	if ((threadIdx.x & 63) == 0) { Stamp st; st.ticks = t1 - t0; st.real = r1 - r0; stamps[(blockIdx.x * 256 + threadIdx.x) >> 6] = st; }
	out[blockIdx.x * 256 + threadIdx.x] = acc + (float) rng;
}

static rt_launch make_launch(uint32_t *d_sky, rt_geom *d_geom, rt_shade *d_shade)
{
	rt_launch L; memset(&L, 0, sizeof(L));
	L.num_objects = SPEC_N; L.sky = d_sky; L.sky_w = 64; L.sky_h = 64; L.sky_wm1 = 63.0f; L.sky_hm1 = 63.0f;
	L.geom = d_geom; L.shade = d_shade;
	return L;
}

template <int W> double run(const char *name, float *d_out, Stamp *d_stamps, int cus, const rt_launch &L, double base)
{
	const int blocks = cus * 4, iters = 4000;
	const size_t lds = (size_t) SPEC_N * (sizeof(rt_geom) + sizeof(rt_shade));
	for (int warm = 0; warm < 2; warm++) hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), lds, 0, d_out, d_stamps, iters, L);
	(void) hipDeviceSynchronize();
	hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	(void) hipEventRecord(e0);
	hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), lds, 0, d_out, d_stamps, iters, L);
	(void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
	float ms; (void) hipEventElapsedTime(&ms, e0, e1);
	std::vector<Stamp> st((size_t) blocks * 4);
	(void) hipMemcpy(st.data(), d_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
	std::vector<double> ghz;
	for (auto &s : st) ghz.push_back((double) s.ticks / (double) s.real * 0.1);
	std::sort(ghz.begin(), ghz.end());
	const double clock = ghz[ghz.size() / 2];
	// SIMD cycles per call of one wave (4 waves share the SIMD): wall x clock / (4 waves x iters)
	const double cyc = ms * 1e-3 * clock * 1e9 / (4.0 * iters);
	printf("%-46s %8.1f SIMD-cycles per wave-call (%.2f GHz)   net of ray generation: %8.1f\n", name, cyc, clock, cyc - base);
	return cyc;
}

int main()
{
	hipDeviceProp_t p; (void) hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	float *d; (void) hipMalloc(&d, sizeof(float) * cus * 4 * 256);
	Stamp *s; (void) hipMalloc(&s, sizeof(Stamp) * cus * 4 * 4);
	uint32_t *sky; (void) hipMalloc(&sky, 6 * 64 * 64 * 4); (void) hipMemset(sky, 0x40, 6 * 64 * 64 * 4);
	std::vector<rt_geom> g(SPEC_N); std::vector<rt_shade> sh(SPEC_N);
	for (int i = 0; i < SPEC_N; i++) { memset(&g[i], 0, sizeof(rt_geom)); memset(&sh[i], 0, sizeof(rt_shade)); g[i].type = SPEC_T[i];
		g[i].a[0] = SPEC_G[i][0]; g[i].a[1] = SPEC_G[i][1]; g[i].a[2] = SPEC_G[i][2]; g[i].b0 = SPEC_G[i][3]; g[i].b1 = SPEC_G[i][4]; g[i].b2 = SPEC_G[i][5]; }
	rt_geom *dg; rt_shade *ds; (void) hipMalloc(&dg, sizeof(rt_geom) * SPEC_N); (void) hipMalloc(&ds, sizeof(rt_shade) * SPEC_N);
	(void) hipMemcpy(dg, g.data(), sizeof(rt_geom) * SPEC_N, hipMemcpyHostToDevice); (void) hipMemcpy(ds, sh.data(), sizeof(rt_shade) * SPEC_N, hipMemcpyHostToDevice);
	const rt_launch L = make_launch(sky, dg, ds);
	printf("%s, %d CUs, scene_0 compiled in (%d objects), 4 waves per SIMD\n", p.name, cus, SPEC_N);
	const double base = run<0>("ray generation (6 LCG draws)", d, s, cus, L, 0.0);
	const double u = run<1>("+ unit3_fast(d)", d, s, cus, L, base);
	run<8>("+ unit3_fast + prepare_ray", d, s, cus, L, base);
	run<9>("+ unit3_fast + prepare_ray + the 3 spheres", d, s, cus, L, base);
	run<2>("+ unit3_fast + nearest_hit_spec (bounce ray)", d, s, cus, L, base);
	run<3>("+ unit3_fast + nearest_hit_spec (shadow tap)", d, s, cus, L, base);
	run<4>("+ unit3_fast + nearest_hit_fast (generic, LDS scene)", d, s, cus, L, base);
	run<5>("+ rng_draw", d, s, cus, L, base);
	run<6>("+ rng_direction (3 draws + unit3_fast)", d, s, cus, L, base);
	run<7>("+ unit3_fast + sky_texel", d, s, cus, L, base);
	(void) u;
	return 0;
}
