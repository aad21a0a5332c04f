/* rt_internal.h -- declarations shared between rt_api.cpp and rt_kernels.hip (not installed). */
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H

#include <hip/hip_runtime_api.h>
#include <string>
#include <vector>
#include "rt_device.h"
#include "../../include/rt_hip.h"
#include "../../include/rt_hip_testing.h"

struct rt_context;
/* everything a context can be told about how to run its launches: rt_tuning (rt_hip.h) and rt_test_knobs (rt_hip_testing.h) land here */
struct rt_knobs {
	int dequeue_shards = 0, workgroups_per_cu = 0, jit_waves_per_simd = 0, audit_known_taps = 0;
	int force_collective = 0, poison_frame = 0, trace_known_taps = 0, test_every_object = 0, test_drop_pixels = 0, test_corrupt_lit_table = 0;
};
/* rt_api.cpp: sets the thread's error text (rt_last_error()) and returns `code`; the context's own stream */
int        rt_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void      *rt_context_stream(rt_context *ctx);
/* A launch's control words travel to the host in a copy of RT_CTL_COPY_BYTES, of which the first RT_CTL_WORDS words matter.  The runtime
 * does device-to-host copies of up to 16 KB with a KERNEL, and a kernel needs a workgroup slot: behind a persistent launch that holds
 * every slot a 64-byte copy took 7.5 ms -- until that launch drained -- and every frame of a frame loop was delivered one launch late
 * (the copy sits between the frame and the event the host waits for).  From 64 KB on a DMA engine does the copy: 0.05 ms beside
 * the same launch (profiles/r05/copy_beside_kernel.txt).  Destinations are pinned blocks of RT_CTL_COPY_BYTES; the scratch sets'
 * counter blocks are allocated that much longer. */
#define RT_CTL_COPY_BYTES 65536
#define RT_CTL_COPY_WORDS (RT_CTL_COPY_BYTES / 4)
/* copies the control words of the context's most recent launch to pinned h_dst[0 ... RT_CTL_COPY_WORDS) on `stream`, which the caller has
 * ordered behind that launch (not the launch's own stream); *behind (optional) = an event behind the copy; *expect = what
 * rt_judge_launch() is to hold the words against */
int        rt_context_read_control(rt_context *ctx, unsigned int *h_dst, hipStream_t stream, hipEvent_t *behind, rt_launch_expect *expect);
/* the verdict on a launch from its control words: RT_OK, RT_CANCELLED (rt_cancel() cut it short), or RT_ERR_DEVICE with the counts
 * in rt_last_error() -- no stamp of the last wave, listed pixels not fetched or not written, camera-ray blocks missing, audited
 * taps that contradict rt_lit.h: the frame is not to be delivered.  report (optional) receives the numbers. */
int        rt_judge_launch(const unsigned int *words, const rt_launch_expect &expect, const char *who, rt_launch_report *report);
void      *rt_context_launch_done(rt_context *ctx);                   /* hipEvent_t recorded behind the context's most recent launch, on that launch's stream */
#define RT_PROGRESSIVE_ROW_BLOCK 16      /* == the header's: a multiple of every scale of the ladder */
int        rt_progressive_count(rt_context *ctx, float *count);      /* the ladder's sum of published weights, read from the device (waits for the passes enqueued so far) */

size_t     rt_counter_bytes();
size_t     rt_scene_lds_bytes(int num_objects);
size_t     rt_wavefront_lds_bytes(int num_objects);
void       rt_primary_geometry(int width, int local_rows, int num_cus, unsigned int *groups_out, int *per_group_out);
size_t     rt_pixel_list_capacity(int width, int local_rows, int num_cus, int num_shards);
/* spec_fn: kernel compiled by rt_compile_scene for the current scene, or nullptr.  Events recorded on `stream`: `cleared`
 * (may be nullptr) behind the clearing of the counters, i.e. when the launch has got its first compute unit;
 * `primary_done` in front of the trace kernel (behind the whole launch when there is no separate primary pass) */
hipError_t rt_launch_trace(const rt_launch &L, int variant, bool scene_fast_ok, hipFunction_t spec_fn,
                           unsigned int *block_counter, hipEvent_t cleared, hipEvent_t primary_done, int num_cus, int workgroups_per_cu, hipStream_t stream,
                           bool reuse_pixel_lists = false,    /* the scratch set still holds rt_primary_pass's output for this very launch */
                           rt_launch_expect *expect = nullptr,
                           hipEvent_t primary_timed = nullptr,    /* recorded between the camera-ray pass and the trace kernel (profiling) */
                           bool audit = false);                   /* the kernel variant that compares audited answers of rt_lit.h with a trace (spec_fn, if given, must be that variant) */
int        rt_jit_build(const rt_geom *geom, int n, int light_index, const float light_pos[3], int only_light_emits, int waves_per_simd, const char *extra_flags, hipModule_t *module, hipFunction_t *function, std::string &message,
                        std::vector<char> *code_out = nullptr,    /* the code object (development aid) */
                        std::string *compiler = nullptr,          /* where it came from: "embedded, compiled with the library by ..." / "hiprtc x.y at run time" */
                        bool embedded_only = false);              /* only what was compiled with the library: RT_ERR_STATE, and nothing compiled, if the scene (with these flags) is not among it */
hipError_t rt_launch_deinterleave(const float *strips, float *frame, int width, int height,
                                  int row_block, int world, int rows_per_rank, int first, hipStream_t stream);

/* the publish steps look at the launch's control words ON THE DEVICE (`control`, `expect`; control = nullptr: nothing was
 * launched, the step always publishes): a launch cut short by rt_cancel() is not published (main.c:382), an incomplete one is not
 * published either and counted in count[RT_COUNT_INCOMPLETE] */
hipError_t rt_launch_accumulate(float *accum, const float *lowres, int width, int height, int scale,
                                int low_w, int low_h, float k, const unsigned int *control, const rt_launch_expect &expect, float *count,
                                int row_block, int rank, int world, int local_rows, hipStream_t stream);
hipError_t rt_launch_commit_sums(float *accum, const float *sums, size_t floats, int passes, const unsigned int *control, const rt_launch_expect &expect, float *count, hipStream_t stream);
hipError_t rt_launch_resolve(const float *accum, float *frame, size_t floats, const float *count, hipStream_t stream);
hipError_t rt_launch_selftest(int which, uint64_t seed, int blocks, int iters, unsigned long long *d_out, hipStream_t stream);

#endif
