#!/usr/bin/env python3
"""The patch of INTEGRATION.md section 2, applied: turns the reference's src/main.c (or src/camera.c) into the host that
drives librt_hip.so.  A text filter -- stdin -> stdout -- so that nothing of the reference is ever written into this
repository; tests/test_c_abi_compile.py pipes its output into `gcc -x c -c -` and links the result against the library.

    reference_main_rt.py main [--ladder | --blocking]  < src/main.c    > patched main.c
    reference_main_rt.py camera                        < src/camera.c  > camera.c + the two getters it lacks
    reference_main_rt.py binding [--ladder | --blocking]               > the inserted text alone (no input)

Two variants of the main.c patch:

  --ladder   (the default, the documented one) keeps the reference's defining behaviour: passes ACCUMULATE until the camera
             moves, and after every invalidation the image refines from 1/init_scale resolution up to full resolution
             (worker() main.c:354-408, update_frame() :450-482, invalidate_accumulation() :115-124, --init-scale :585-634).
               realloc_frame_buffer()      -> + rt_progressive_begin(rt, frame_w, frame_h, init_scale, 10, seed)
               update_frame()              -> rt_progressive_passes(rt, n) + rt_progressive_resolve(rt, frame) + move_frame_to_the_gpu
               invalidate_accumulation()   -> rt_set_camera + rt_progressive_invalidate
  --blocking one independent rt_render() of 16 samples per pixel per shown frame (a new seed each): the simplest binding, for
             hosts that want whole frames rather than the interactive protocol; init_scale is parsed and unused.

Every edit is anchored on a statement of the reference and must match exactly once (the script fails otherwise: a reference
that has moved on needs a new patch, not a silent no-op).  What it does to main.c (line numbers: the reference's):
  * after the includes (:32-36): rt_hip.h with RT_HAVE_REFERENCE_TYPES, the context, the getters' prototypes;
  * invalidate_accumulation() (:115-124) and update_frame() (:450-482): replaced as above -- the workers' generation
    counter, mutex and condition variables have no work left;
  * main(): start_workers() (:516) becomes rt_create / rt_set_scene / rt_set_skybox (/ rt_compile_scene), stop_workers()
    (:577) becomes rt_destroy().

The `binding` mode prints exactly the text the patch inserts (header + invalidate_accumulation() + update_frame()), so that a
test can compile THAT text against stand-ins for the handful of reference globals it touches and run it on a GPU, where the
reference itself cannot travel (tests/test_gpu_ladder_binding.py, tests/c/ladder_host.c)."""
import re
import sys

HEADER_COMMON = r'''
/* ---- librt_hip.so binding (INTEGRATION.md section 2) ---- */
#define RT_HAVE_REFERENCE_TYPES
#include <rt_hip.h>
Vector3 get_camera_front(void);   /* the two getters camera.c lacks */
Vector3 get_camera_up(void);
static rt_context *rt;            /* replaces the worker threads, accum_conds[], accum_counts[] */
'''

# ---------------------------------------------------------------------------------------------------------------------------
# --ladder: accumulate until the camera moves, refine from 1/init_scale (the reference's interactive protocol)
# ---------------------------------------------------------------------------------------------------------------------------
LADDER_HEADER = HEADER_COMMON + r'''
/* worker() iterations (main.c:354-408) asked of the GPU between two shown frames.  Right after an invalidation one pass per
 * shown frame -- the low-resolution steps of the ladder, so that a camera move is answered at once (a pass is 0.05-0.2 ms) --
 * then sixteen to a call: at full resolution the library renders them in one launch, bit-identical to sixteen single passes. */
#ifndef RT_PASSES_PER_FRAME
#define RT_PASSES_PER_FRAME 16
#endif
static int rt_ladder_running;     /* rt_progressive_begin() has been called for frame_w x frame_h */

static void rt_die(void)
{
	fprintf(stderr, "rt: %s\n", rt_last_error());
	abort();                      /* the reference's own policy on failure (main.c:373,425-434) */
}

static void rt_bind_camera(void)
{
	rt_camera cam = { get_camera_pos(), get_camera_front(), get_camera_up(), 30.0f };   /* camera.c:28,33-35 */
	if (rt_set_camera(rt, &cam) != RT_OK) rt_die();
}
'''

LADDER_INVALIDATE = r'''void invalidate_accumulation(void)
{
	/* main.c:115-124: counts to zero, generation + 1, buffers cleared -- and the workers restart from init_scale.  Here the
	 * new pose goes to the device and the ladder restarts; a pass in flight is cancelled and never published (main.c:382). */
	if (rt && rt_ladder_running) {
		rt_bind_camera();
		if (rt_progressive_invalidate(rt) != RT_OK) rt_die();
	}
	if (frame) memset(frame, 0, sizeof(Vector3) * frame_w * frame_h);
}
'''

LADDER_UPDATE_FRAME = r'''void update_frame(void)
{
	if (frame_buffer_size_doesnt_match_window()) {
		realloc_frame_buffer();                               /* unchanged: allocates `frame` (main.c:416-443) */
		rt_bind_camera();
		/* the accumulation buffer, its weights and the generation counter live on the device from here on;
		 * init_scale is the global --init-scale sets (main.c:50, :585-634); 10 is the bounce limit of main.c:156 */
		if (rt_progressive_begin(rt, frame_w, frame_h, init_scale, 10, 0) != RT_OK) rt_die();
		rt_ladder_running = 1;
	}

	/* the workers' iterations since the last shown frame (main.c:354-408): scale ladder, weights 1/scale^2, in pass order */
	int next_scale = 0, passes = 0;
	if (rt_progressive_state(rt, &next_scale, NULL, NULL, &passes) != RT_OK) rt_die();
	int n = next_scale > 1 || passes == 0 ? 1 : RT_PASSES_PER_FRAME;
	if (rt_progressive_passes(rt, n) != RT_OK) rt_die();

	/* frame = accum * (1 / sum of weights), the order of main.c:467-477; every column has a pass by now (main.c:461-464) */
	if (rt_progressive_resolve(rt, frame) != RT_OK) rt_die();
	move_frame_to_the_gpu(frame_w, frame_h, frame);           /* unchanged (main.c:479) */
}
'''

# ---------------------------------------------------------------------------------------------------------------------------
# --blocking: one independent frame of 16 samples per pixel per update_frame()
# ---------------------------------------------------------------------------------------------------------------------------
BLOCKING_HEADER = HEADER_COMMON + r'''static int rt_passes;             /* frames shown since the last invalidation */
'''

BLOCKING_INVALIDATE = r'''void invalidate_accumulation(void)
{
	rt_passes = 0;
	if (rt) rt_cancel(rt);        /* one atomic max on a word the kernels poll: callable from here, returns at once */
	if (frame) memset(frame, 0, sizeof(Vector3) * frame_w * frame_h);
}
'''

BLOCKING_UPDATE_FRAME = r'''void update_frame(void)
{
	if (frame_buffer_size_doesnt_match_window()) {
		realloc_frame_buffer();
		rt_reserve(rt, frame_w, frame_h);
	}
	rt_camera cam = { get_camera_pos(), get_camera_front(), get_camera_up(), 30.0f };   /* camera.c:28,33-35 */
	rt_set_camera(rt, &cam);

	rt_render_params p;
	rt_default_params(&p, frame_w, frame_h, 16, 10);          /* 16 passes per shown frame; the bounce limit of main.c:156 */
	p.seed = (uint64_t) rt_passes++;
	int rc = rt_render(rt, &p, frame);                        /* fills frame[j * frame_w + i], the order of main.c:467-477 */
	if (rc == RT_CANCELLED) return;                           /* the camera moved meanwhile: this frame is skipped */
	if (rc != RT_OK) {
		fprintf(stderr, "rt: %s\n", rt_last_error());
		abort();                                              /* the reference's own policy on failure (main.c:373,425-434) */
	}
	move_frame_to_the_gpu(frame_w, frame_h, frame);           /* unchanged (main.c:479) */
}
'''

START = r'''if (rt_create(&rt, 0) || rt_set_scene(rt, &scene) || rt_set_skybox(rt, &skybox)) {
		fprintf(stderr, "rt: %s\n", rt_last_error());
		return -1;
	}
	if (rt_compile_scene(rt) != RT_OK)
		fprintf(stderr, "rt: %s (continuing with the generic kernel)\n", rt_last_error());
'''

GETTERS = r'''
/* ---- added for the librt_hip.so binding (INTEGRATION.md section 2): the pose is file-static here ---- */
Vector3 get_camera_front(void) { return camera_front; }
Vector3 get_camera_up(void)    { return camera_up; }
'''

VARIANTS = {
    "--ladder":   (LADDER_HEADER, LADDER_INVALIDATE, LADDER_UPDATE_FRAME),
    "--blocking": (BLOCKING_HEADER, BLOCKING_INVALIDATE, BLOCKING_UPDATE_FRAME),
}


def once(pattern, repl, text, flags=0):
    out, n = re.subn(pattern, lambda m: repl, text, count=0, flags=flags)
    if n != 1:
        sys.exit(f"reference_main_rt.py: anchor {pattern!r} matched {n} times, expected 1")
    return out


def function_body(name):
    """`<type> name(void) { ... }` up to the closing brace in column 0 (the reference's style)."""
    return r"^[A-Za-z_][\w \*]*\b" + name + r"\(void\)\n\{\n.*?^\}\n"


def patch_main(src, variant):
    header, invalidate, update_frame = VARIANTS[variant]
    # the ladder binding reads the reference's own --init-scale global: it must still be there (main.c:50)
    once(r"^int init_scale;\n", "", src, re.M)
    src = once(r'^#include "gpu_and_windowing\.h"\n', '#include "gpu_and_windowing.h"\n' + header, src, re.M)
    src = once(function_body("invalidate_accumulation"), invalidate, src, re.M | re.S)
    src = once(function_body("update_frame"), update_frame, src, re.M | re.S)
    # inside main(): the calls, not the definitions (which are `void start_workers(void)` at column 0)
    src = once(r"^\tstart_workers\(\);\n", "\t" + START, src, re.M)
    src = once(r"^\tstop_workers\(\);\n", "\trt_destroy(rt);\n", src, re.M)
    return src


def patch_camera(src):
    once(r"^static Vector3 camera_front\b", "", src, re.M)      # (the statics the getters return must exist)
    once(r"^static Vector3 camera_up\b", "", src, re.M)
    return src + GETTERS


def binding_text(variant):
    return "".join(VARIANTS[variant])


if __name__ == "__main__":
    args = sys.argv[1:]
    variant = "--ladder"
    for a in list(args):
        if a in VARIANTS:
            variant = a
            args.remove(a)
    if len(args) != 1 or args[0] not in ("main", "camera", "binding"):
        sys.exit(__doc__)
    if args[0] == "binding":
        sys.stdout.write(binding_text(variant))
    else:
        text = sys.stdin.read()
        sys.stdout.write(patch_main(text, variant) if args[0] == "main" else patch_camera(text))
