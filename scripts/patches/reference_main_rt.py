#!/usr/bin/env python3
"""The patch of INTEGRATION.md section 2, applied: turns the reference's src/main.c (or src/camera.c) into the host that
drives librt_hip.so.  A text filter -- stdin -> stdout -- so that nothing of the reference is ever written into this
repository; tests/test_c_abi_compile.py pipes its output into `gcc -x c -c -` and links the result against the library.

    reference_main_rt.py main   < src/main.c    > patched main.c
    reference_main_rt.py camera < src/camera.c  > camera.c + the two getters it lacks

Every edit is anchored on a statement of the reference and must match exactly once (the script fails otherwise: a reference
that has moved on needs a new patch, not a silent no-op).  What it does to main.c (line numbers: the reference's):
  * after the includes (:32-36): rt_hip.h with RT_HAVE_REFERENCE_TYPES, the context, the pass counter, the getters' prototypes;
  * invalidate_accumulation() (:115-124): the workers' generation counter and mutex go; rt_cancel() gives up what is in flight;
  * update_frame() (:450-482): camera -> rt_set_camera, one rt_render() into `frame`, move_frame_to_the_gpu() as before;
  * main(): start_workers() (:516) becomes rt_create / rt_set_scene / rt_set_skybox (/ rt_compile_scene), stop_workers()
    (:577) becomes rt_destroy()."""
import re
import sys

HEADER = r'''
/* ---- librt_hip.so binding (INTEGRATION.md section 2) ---- */
#define RT_HAVE_REFERENCE_TYPES
#include <rt_hip.h>
Vector3 get_camera_front(void);   /* the two getters camera.c lacks */
Vector3 get_camera_up(void);
static rt_context *rt;            /* replaces the worker threads, accum_conds[], accum_counts[] */
static int rt_passes;             /* frames shown since the last invalidation */
'''

INVALIDATE = r'''void invalidate_accumulation(void)
{
	rt_passes = 0;
	if (rt) rt_cancel(rt);        /* one atomic max on a word the kernels poll: callable from here, returns at once */
	if (frame) memset(frame, 0, sizeof(Vector3) * frame_w * frame_h);
}
'''

UPDATE_FRAME = r'''void update_frame(void)
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


def once(pattern, repl, text, flags=0):
    out, n = re.subn(pattern, lambda m: repl, text, count=0, flags=flags)
    if n != 1:
        sys.exit(f"reference_main_rt.py: anchor {pattern!r} matched {n} times, expected 1")
    return out


def function_body(name):
    """`<type> name(void) { ... }` up to the closing brace in column 0 (the reference's style)."""
    return r"^[A-Za-z_][\w \*]*\b" + name + r"\(void\)\n\{\n.*?^\}\n"


def patch_main(src):
    src = once(r'^#include "gpu_and_windowing\.h"\n', '#include "gpu_and_windowing.h"\n' + HEADER, src, re.M)
    src = once(function_body("invalidate_accumulation"), INVALIDATE, src, re.M | re.S)
    src = once(function_body("update_frame"), UPDATE_FRAME, src, re.M | re.S)
    # inside main(): the calls, not the definitions (which are `void start_workers(void)` at column 0)
    src = once(r"^\tstart_workers\(\);\n", "\t" + START, src, re.M)
    src = once(r"^\tstop_workers\(\);\n", "\trt_destroy(rt);\n", src, re.M)
    return src


def patch_camera(src):
    once(r"^static Vector3 camera_front\b", "", src, re.M)      # (the statics the getters return must exist)
    once(r"^static Vector3 camera_up\b", "", src, re.M)
    return src + GETTERS


if __name__ == "__main__":
    if len(sys.argv) != 2 or sys.argv[1] not in ("main", "camera"):
        sys.exit(__doc__)
    text = sys.stdin.read()
    sys.stdout.write(patch_main(text) if sys.argv[1] == "main" else patch_camera(text))
