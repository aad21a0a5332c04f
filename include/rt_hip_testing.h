/*
 * rt_hip_testing.h -- what librt_hip.so exports for its own test suite, measurement scripts and development tools.
 *
 * Nothing here is part of the drop-in boundary (rt_hip.h): a host of the reference needs none of it, and none of it may be
 * relied upon to stay.  It is a separate header so that the boundary header carries only what a maintainer binds
 * (round 5's review: rt_tuning had grown three fault-injection fields, and rt_hip.h declared self-tests and instrumentation
 * read-outs beside rt_render()).  The symbols are exported from the same library; tests/, bench.py and scripts/ use them.
 */
#ifndef RT_HIP_TESTING_H
#define RT_HIP_TESTING_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Knobs that exist so that tests can reach a code path, cross-check a shortcut or inject a fault.  Like rt_tuning: set per
 * context, read at every launch, `size` = sizeof(rt_test_knobs) checked by the setters; 0 = off.  Only the two fault
 * injections change a frame -- by making it wrong on purpose. */
typedef struct {
	size_t size;                /* = sizeof(rt_test_knobs); set by rt_default_test_knobs() */
	int    force_collective;    /* rt_multi_render() runs its ncclGather + de-interleave path even for a single device (a one-rank
	                             * communicator), so that path can be checked on a 1-GPU box */
	int    poison_frame;        /* fill the destination with NaNs before every launch, so that a pixel the launch fails to write
	                             * cannot pass for correct because an earlier frame left it there */
	int    trace_known_taps;    /* trace every soft-shadow tap, also those of hit points from which every tap provably reaches the
	                             * emitter first (csrc/rt_lit.h; normally answered untraced): the cross-check of that proof */
	int    test_every_object;   /* scenes of 32 objects and more are rendered without the cluster cull (csrc/rt_cull.h) -- every ray
	                             * tests every object, as the reference does; same frames, slower: the cross-check of the cull's margins */
	int    test_drop_pixels;    /* FAULT INJECTION: the trace kernel's waves see every pixel list this many entries shorter -- a launch
	                             * that loses its tail, which every delivering call must then refuse (RT_ERR_DEVICE) */
	int    test_corrupt_lit_table; /* FAULT INJECTION, read by rt_set_scene(): every cell of the scene's lit-taps table says "certainly
	                             * lit" -- a wrong csrc/rt_lit.h, which rt_tuning.audit_known_taps must then catch */
} rt_test_knobs;
RT_API void rt_default_test_knobs(rt_test_knobs *k);
RT_API int  rt_set_test_knobs(rt_context *ctx, const rt_test_knobs *knobs);
RT_API int  rt_multi_set_test_knobs(rt_multi *m, const rt_test_knobs *knobs);

/* A group of n contexts that all live on ONE device, for 1-GPU boxes.  Everything the n-device path does runs -- n strips from n
 * contexts on their own streams, the strip buffers in rotation, the rotated hand-out, the de-interleave, the frame queue, the
 * ladder -- except RCCL, which refuses two ranks on one device: the gather is the n device-to-device copies it amounts to
 * there.  Frames are bit-identical to rt_render()'s.  Not a performance configuration. */
RT_API int rt_multi_create_on_one_device(rt_multi **out, int device_id, int n);

/* rt_compiled_scene_cache_cap(cap >= 1) changes the cap of the per-process cache of compiled scenes (rt_hip.h:
 * rt_compiled_scene_counts); returns the old one. */
RT_API int rt_compiled_scene_cache_cap(int cap);

/* Instrumentation counters of a compiled kernel built with rt_tuning.jit_flags "-DRT_STATS" (scripts/stats_c1.py).  They are
 * variables of the compiled MODULE, i.e. per (device, scene, options), not per context: contexts that share a compiled scene
 * read and reset the same counters (the same holds for rt_spec_symbol_read). */
RT_API int rt_spec_stats_read(rt_context *ctx, unsigned long long out[64], int reset);
/* Copy the named device variable of the compiled kernel's module (e.g. "rt_wave_log" of a build with "-DRT_STATS
 * -DRT_STATS_LIFETIMES_ONLY", scripts/probes/tail_probe.py) to dst, at most `bytes` bytes; *copied = bytes copied.  The empty
 * name "" stands for the compiled kernel's code object itself (*copied = its full size), for disassembly. */
RT_API int rt_spec_symbol_read(rt_context *ctx, const char *name, void *dst, size_t bytes, size_t *copied);

/* How many of this context's launches ran rt_primary_pass (camera rays) -- an interactive pass that differs from the pass
 * before last in its sample number only keeps that pass's camera rays instead. */
RT_API long long rt_primary_passes_run(rt_context *ctx);

/* On-GPU self-test of the exact-arithmetic shortcuts the tuned kernel uses (shared-reciprocal
 * division, vector normalisation): compares them bit-for-bit with the plain IEEE forms on
 * blocks*256*iters random operand sets.  which = 0 (f32 divide), 1 (f64 divide), 2 (normalize), 3 (f64 sqrt of a float), 4 (|x| < 0.0001 threshold), 7 (normalize of a `draw * 2 - 1` vector).
 * Three sweeps are exhaustive instead of random: which = 3 checks the fp64 square root of every normal
 * float up to 2^120 (`iters` ignored); which = 5 checks the refined reciprocal for all 2^23
 * significands and the 3-instruction quotient for every denominator significand x `iters` numerator
 * significands (iters = 8388608 covers all 2^46 pairs, ~1 min; `seed` picks the first numerator);
 * which = 6 checks the tuned sqrtf on every float in [2^-30, 2^60] (`iters` ignored).
 * out[0] = mismatches (must be 0); out[1..7] = operands of one mismatch, for diagnosis. */
RT_API int rt_selftest(rt_context *ctx, int which, uint64_t seed, int blocks, int iters, unsigned long long out[8]);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_TESTING_H */
