/*
 * rt_hip.h -- C ABI of librt_hip.so: the MI355X path-tracing back end.
 *
 * The reference (cozis/ray_tracing) has no FFI seam for this path: the work is done by N worker
 * pthreads inside main.c.  This header is the seam a maintainer would bind instead.  Each entry
 * point names the reference code it replaces (paths relative to the reference repo).
 *
 *   reference today                                    this library
 *   -----------------------------------------------    ------------------------------------------
 *   Scene scene; parse_scene_file()   main.c:54,493    rt_parse_scene_file() + rt_set_scene()
 *   Cubemap skybox; load_cubemap()    main.c:55,500    rt_load_cubemap()     + rt_set_skybox()
 *   camera.c statics                  camera.c:28-35   rt_camera + rt_set_camera()
 *   start_workers()/worker()/render_column()/pixel()
 *     + accum[] += ... + update_frame() resolve
 *                          main.c:695,324,274,131,467  rt_render()  (one synchronous call)
 *   move_frame_to_the_gpu(w,h,frame)  main.c:479       caller passes the filled Vector3[w*h] on
 *
 * Conventions: every function returns 0 on success and a negative rt_status on failure and never
 * aborts (the reference abort()s / exit()s: main.c:373,425-434); rt_last_error() gives the text.
 * All host pointers are caller-owned and not retained after the call returns.  A context is bound
 * to one GPU and must be used from one host thread at a time.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API __attribute__((visibility("default")))

typedef enum {
	RT_PENDING       =  2,   /* not an error: rt_frame_poll() -- the frame is not in host memory yet */
	RT_CANCELLED     =  1,   /* not an error: the launch was cut short by rt_cancel(); the frame is incomplete */
	RT_OK            =  0,
	RT_ERR_ARGUMENT  = -1,   /* NULL / out-of-range argument                     */
	RT_ERR_DEVICE    = -2,   /* HIP runtime error (text in rt_last_error())      */
	RT_ERR_STATE     = -3,   /* scene / skybox / camera not set before rendering */
	RT_ERR_IO        = -4,   /* file could not be read                           */
	RT_ERR_FORMAT    = -5,   /* scene text or JPEG stream rejected               */
	RT_ERR_MEMORY    = -6
} rt_status;

typedef struct rt_context rt_context;

/* Kernel selection.  RT_KERNEL_AUTO picks the tuned kernel; RT_KERNEL_SIMPLE is the path loop in
 * the reference's own order, kept as an on-GPU cross-check. */
enum { RT_KERNEL_AUTO = 0, RT_KERNEL_SIMPLE = 1, RT_KERNEL_WAVEFRONT = 2 };

/*
 * What the reference hard-codes or lacks, made explicit:
 *   spp          number of 1-sample passes accumulated (reference: until the camera moves, main.c:357)
 *   max_bounces  main.c:156's literal 10
 *   seed         `counter` RNG mode: path (pixel p, sample s) starts from rt_path_seed(seed, p, s);
 *                the reference's thread-local stream (utils.c:60) cannot be evaluated in parallel
 *   row_block/rank/world   interleaved row-block partition for multi-GPU: this call renders the row
 *                blocks b with b % world == rank into a compact strip (see rt_strip_rows()).
 *                Single GPU: row_block = any (>0), rank = 0, world = 1 -> the strip is the frame.
 */
typedef struct {
	int      width, height;
	int      spp;
	int      max_bounces;
	uint64_t seed;
	int      row_block, rank, world;
	int      kernel;            /* RT_KERNEL_* */
} rt_render_params;

RT_API void rt_default_params(rt_render_params *p, int width, int height, int spp, int max_bounces);

/* Which revision of this header a library was built from: changes whenever a struct layout or a function signature here
 * changes.  A host compares rt_abi_version() with the RT_ABI_VERSION it was compiled against before anything else. */
#define RT_ABI_VERSION 6
RT_API int rt_abi_version(void);
#define RT_AUDIT_OFF (-2)           /* rt_tuning.audit_known_taps */

/* Scheduling knobs of the trace kernels.  None of them changes a single bit of any frame (tests render with
 * several settings and compare); they exist for hosts with a loop of their own and for measurement scripts.  0 / NULL = let the
 * library decide.  Set per context, read at every launch; the library never reads environment variables.
 * `size` is sizeof(rt_tuning) as the HOST was compiled: rt_default_tuning() sets it, rt_set_tuning() returns RT_ERR_ARGUMENT when
 * it is not the library's own -- a host built against another revision of this header is told so instead of having a struct of
 * another layout read.  (The knobs that exist for the test suite only -- poisoned frames, injected faults -- are in
 * rt_hip_testing.h: rt_test_knobs.) */
typedef struct {
	size_t size;                /* = sizeof(rt_tuning); set by rt_default_tuning() */
	int    dequeue_shards;      /* lists the object pixels are dealt from: 1 or 64 */
	int    workgroups_per_cu;   /* resident workgroups per CU, 1..8, capped by what fits (0: all that fit -- half of them for a launch
	                             * enqueued before the previous one, on the context's other stream, has started: the two are resident
	                             * side by side; any setting also keeps a large scene's culled kernel on workgroups of four waves) */
	int    jit_waves_per_simd;  /* rt_compile_scene: register budget = 512 / this many VGPRs (default 4) */
	int    audit_known_taps;    /* csrc/rt_lit.h audited in production.  Of the answers "these soft-shadow taps need no tracing" (per camera-ray
	                             * hit point, per cell of the scene's table) some are marked: bounces that use a marked answer have their taps
	                             * traced all the same and compared; the frame is unchanged, a disagreement fails the launch (RT_ERR_DEVICE,
	                             * rt_launch_report.taps_disagreeing).  0, the default: the BACKGROUND audit -- one launch in 61 of a context
	                             * is rendered that way with one answer in 8 marked (that launch costs 2 % more: 0.03 % of a frame loop;
	                             * nothing is ever compiled for it); k = 1 ... 24: every launch, one answer in 2^k; -1: every launch, every
	                             * answer; RT_AUDIT_OFF: never (measurement scripts that compare kernels) */
	const char *jit_flags;      /* rt_compile_scene: extra hiprtc options, space separated (copied) */
} rt_tuning;
RT_API void rt_default_tuning(rt_tuning *t);

/* ---- context ------------------------------------------------------------------------------ */
RT_API int  rt_create(rt_context **out, int device_id);
RT_API void rt_destroy(rt_context *ctx);
RT_API const char *rt_last_error(void);

/* ---- inputs: replaces the globals of main.c:54-55 and the statics of camera.c:28,33-35 ------ */
RT_API int rt_set_scene(rt_context *ctx, const Scene *scene);
/* Optional: compile a trace kernel specialised for the scene that is set (in-process, hiprtc; about
 * a second).  Frames are bit-identical with and without it; it only removes work (shared slab planes are
 * divided once, no geometry loads).  The compiled kernel is dropped by the next rt_set_scene().  Scenes of
 * 1..64 objects; returns an error -- and leaves the generic kernels in use -- if hiprtc is unavailable.
 * The kernels of the three scene files the reference ships are compiled when the library is BUILT and embedded in it: for
 * those (recognised by their packed contents, not by a file name) the call is a module load -- no hiprtc, a few milliseconds,
 * and the same code object in every host process.  rt_compiled_scene_info(): where the kernel in use came from, e.g.
 * "embedded, compiled with the library by hipcc 7.2..." or "hiprtc 7.2 at run time" ("" when no scene is compiled; the string
 * is the context's, valid until the scene changes). */
RT_API int rt_compile_scene(rt_context *ctx);
RT_API int rt_scene_is_compiled(rt_context *ctx);
RT_API const char *rt_compiled_scene_info(rt_context *ctx);
/* Compiled scenes are cached per process -- key: device, scene geometry and emitter, options -- and shared by the contexts
 * that compile the same scene.  The cache holds at most 32 entries, least recently used out first; an evicted entry's module
 * stays loaded (contexts that still use it are unaffected) but is never looked up again.  *cached = entries in the cache,
 * *parked = modules evicted so far. */
RT_API void rt_compiled_scene_counts(int *cached, int *parked);
/* chan must be 3 or 4 (what stb_image returns for the shipped JPEGs is 3); all faces w x h */
RT_API int rt_set_skybox(rt_context *ctx, const Cubemap *skybox);
RT_API int rt_set_camera(rt_context *ctx, const rt_camera *camera);
RT_API int rt_set_tuning(rt_context *ctx, const rt_tuning *tuning);

/* ---- the hot path: replaces start_workers()+worker()+update_frame() ------------------------ */
/* Renders the whole frame (world must be 1) into caller-allocated host memory: width*height
 * Vector3, row-major, frame[j*width+i], row 0 = bottom of the displayed image, values in [0,1] --
 * exactly what update_frame() hands to move_frame_to_the_gpu() (main.c:467-479).  Returns RT_CANCELLED when rt_cancel() cut the
 * launch short, and RT_ERR_DEVICE when the launch did not account for every pixel ("every delivered frame is a complete frame"
 * below): an incomplete frame is an error, never a frame with a hole in it; frame_out is then undefined. */
RT_API int rt_render(rt_context *ctx, const rt_render_params *params, Vector3 *frame_out);

/* Same, but the destination is DEVICE memory (rt_strip_rows()*width*12 bytes) and the call only
 * enqueues work on `hip_stream` (a hipStream_t; NULL = the context's own non-blocking stream;
 * RT_STREAM_LEGACY = the device's legacy null stream, which a literal 0 cannot name here).  No sync.
 * A context owns RT_LAUNCH_SETS (five) sets of launch scratch (pixel lists, counters) and uses them in rotation: a launch is
 * ordered (through an event, when the streams differ) behind the launch RT_LAUNCH_SETS before it, never behind the ones in
 * between -- consecutive launches enqueued on different streams may be on the GPU together: one draining, the others running, the
 * last starting in the workgroup slots the first leaves (strips of a millisecond: a seventh more frames per second than one
 * launch after the other, profiles/r05/strip_loop_probe_five_sets.txt; they must of course write different destinations).  A host
 * that keeps five small launches in flight (the strips of an N-GPU frame) is given ONE workgroup slot per CU and launch: a wave
 * then has four times the pixels, and the rounds it spends on the last of them weigh a quarter.  Launches on
 * one stream run in stream order as always.  rt_set_scene / rt_set_skybox wait for the context's own launches only, not for
 * the whole device. */
#define RT_STREAM_LEGACY ((void *) (intptr_t) -1)
#define RT_LAUNCH_SETS 5        /* scratch sets, and streams (rt_stream), a context rotates its launches through */
RT_API int rt_render_device(rt_context *ctx, const rt_render_params *params, void *d_strip, void *hip_stream);

/* The context's streams as hipStream_t: which = 0 the stream NULL stands for above; which = 1 ... RT_LAUNCH_SETS - 1 more
 * (all made by rt_create(), in this order -- the hardware queue a stream is given depends on what exists when it is made --, with
 * the device's lowest stream priority so that they get hardware queues of their own) for hosts that rotate consecutive frames
 * through the streams to let them overlap.  NULL on error. */
RT_API void *rt_stream(rt_context *ctx, int which);

/* Optional: allocate the launch scratch (2 x 48 bytes per pixel of pixel records) and rt_render()'s device frame for
 * frames of up to width x height now.  Without it they are allocated -- a blocking hipMalloc -- inside the first
 * render call of a size and kept; nothing is ever allocated per launch. */
RT_API int rt_reserve(rt_context *ctx, int width, int height);

/* Rows held by one rank's strip, padded so every rank has the same count (gather-friendly). */
RT_API int rt_strip_rows(int height, int row_block, int world);

/* Root side of the multi-GPU exchange: `d_strips` = world strips back to back (what one RCCL
 * gather / all-gather delivers) -> d_frame = height*width Vector3 in frame order. */
RT_API int rt_deinterleave_device(rt_context *ctx, const void *d_strips, void *d_frame,
                                  int width, int height, int row_block, int world, void *hip_stream);

/* Which strip each rank renders is the host's choice (rt_render_params.rank names the STRIP: row blocks b with
 * b % world == rank).  Strips differ by one row block when the blocks do not divide evenly -- the last strip is never
 * longer than any other -- and the root has work the others do not (gather, de-interleave, the copy to the host), so
 * both hosts of this library hand the strips out rotated by one: rank r renders strip rt_strip_of_rank(r, world) =
 * (r + world - 1) % world, the root the last one (1080 rows in blocks of 8 over 8 ranks: 16 blocks instead of 17).
 * The gathered buffer then holds strip s at position (s + first) % world with first = 1:
 * rt_deinterleave_rotated_device() takes that `first` (0: rt_deinterleave_device()). */
RT_API int rt_strip_of_rank(int rank, int world);
RT_API int rt_deinterleave_rotated_device(rt_context *ctx, const void *d_strips, void *d_frame,
                                          int width, int height, int row_block, int world, int first, void *hip_stream);

RT_API int rt_synchronize(rt_context *ctx);

/* ---- frames in flight: the reference's workers keep rendering while its main thread presents -------------------
 * (main.c:354-408 vs main.c:450-482).  rt_render() above is one blocking call per frame; here a frame is SUBMITTED --
 * the call only enqueues: render into the slot's device buffer, copy to `frame_out` on a copy stream of its own -- and
 * WAITED for later, so that the copy of frame k to the host overlaps the render of frame k+1, and consecutive renders
 * overlap each other on the context's streams (the waves of frame k+1 fill the compute units as the waves of frame
 * k run out of pixels).  A slot holds one frame at a time; RT_FRAME_SLOTS of them can be in flight:
 *
 *     rt_frame_submit(ctx, &p0, 0, frame[0]);
 *     for (k = 0; ; k++) {
 *         rt_frame_submit(ctx, &p_next, (k + 1) % 2, frame[(k + 1) % 2]);     // frame k+1 starts behind frame k
 *         rt_frame_wait(ctx, k % 2);                                          // frame k is in frame[k % 2]
 *         move_frame_to_the_gpu(w, h, frame[k % 2]);                          // main.c:479
 *     }
 *
 * frame_out is caller-owned and must stay valid until the slot has been waited for; memory from rt_host_alloc()
 * (page-locked) lets the copy run beside the next render -- with ordinary malloc()ed memory the runtime stages the copy
 * and the submit call may block for it.  Frames are bit-identical to rt_render()'s.  rt_frame_wait() returns RT_OK, or
 * RT_CANCELLED when rt_cancel() cut the frame short; rt_frame_poll() never blocks: RT_PENDING while the frame is not
 * there yet.  Submitting into a slot that holds a frame nobody waited for is an error (RT_ERR_STATE); params->world
 * must be 1.  rt_set_scene / rt_set_skybox / rt_destroy wait for the frames in flight. */
#define RT_FRAME_SLOTS 8
RT_API int rt_frame_submit(rt_context *ctx, const rt_render_params *params, int slot, Vector3 *frame_out);
RT_API int rt_frame_wait(rt_context *ctx, int slot);
RT_API int rt_frame_poll(rt_context *ctx, int slot);
/* The same queue for a consumer that is ON THE GPU.  The reference's only device crossing is host -> GPU: update_frame() hands
 * the finished frame to glTexImage2D (main.c:479, gpu_and_windowing.c:371-376) -- a presenter that takes the frame from device
 * memory (GL / Vulkan interop, a tone-mapping or encoding kernel) needs no copy at all.  rt_frame_submit_device() renders
 * into the slot's device buffer and returns it: *d_frame = height * width Vector3 in frame order on the context's device,
 * *hip_event (optional) = a hipEvent_t recorded behind the render on its stream -- the consumer orders its own stream behind it
 * (hipStreamWaitEvent) or waits on it; nothing is copied to the host but the launch's 4-byte control word.  Both stay valid
 * until the slot is submitted again; rt_frame_wait() / rt_frame_poll() release the slot as usual (and report RT_CANCELLED):
 * wait for the slot before submitting into it again, i.e. when the consumer is done with the buffer.  C3 (3840 x 2160): the
 * delivered-to-host step is the 99.5 MB copy (1.77 ms at 56 GB/s); the device-resident step is the kernels' 0.63 ms. */
RT_API int rt_frame_submit_device(rt_context *ctx, const rt_render_params *params, int slot, void **d_frame, void **hip_event);
/* page-locked host memory for frame_out (hipHostMalloc): any thread, no context needed */
RT_API int  rt_host_alloc(void **out, size_t bytes);
RT_API void rt_host_free(void *p);

/* Giving up what has been asked for, as the reference's workers do when the camera moves mid-pass (main.c:316-317, where
 * the generation counter invalidates whatever every worker is doing): rt_cancel() asks EVERY launch enqueued on this
 * context so far -- the one that is running and those still queued behind it (the second of two frames in flight) -- to
 * stop: its waves hand out no more samples, finish the paths in flight and leave.  A wave looks for the request when it
 * FETCHES PIXELS (eight waves of a launch read the host word and relay it through a word in device memory that every other
 * wave reads at its fetches), so the latency grows with the samples per pixel: 0.4 ms at 64 spp, 0.76 ms at 256, 3.4 ms at 1024
 * (profiles/r06/cancel_latency_claim_rule.txt; a wave claims several pixels at a time only below 128 samples per pixel); a queued
 * launch stops at its first fetch.  The request is one atomic max on a word of
 * host memory that the kernels read: the call returns at once, enqueues nothing, and may come from ANY host thread while
 * another one is inside rt_render*() for the same context -- a launch is covered from the moment the render call has
 * announced it, which is before its first kernel is enqueued.  rt_render() then returns RT_CANCELLED and its frame is incomplete; after
 * rt_render_device(), rt_was_cancelled() (which waits for the launch) tells; rt_frame_wait() reports it for a submitted
 * frame.  Launches enqueued after the call are not affected.  rt_progressive_invalidate() does this by itself for a pass
 * in flight, and that pass is not accumulated. */
RT_API int rt_cancel(rt_context *ctx);
RT_API int rt_was_cancelled(rt_context *ctx);

/* ---- every delivered frame is a complete frame ---------------------------------------------------------------------------
 * The reference publishes a column when render_column() has returned for all of it, under the mutex, or not at all
 * (main.c:377-396).  A GPU launch can fail in ways a function call cannot -- waves that never ran or never finished, a list
 * entry nobody fetched -- so every launch of the trace kernels accounts for itself: the camera-ray pass counts the 8x8 blocks
 * it finished and the object pixels it listed, every wave counts the pixels it wrote, and the LAST wave to leave adds the
 * lists up and stamps the launch with its number.  Those sixteen words travel to the host behind the frame, and every call
 * that delivers a frame -- rt_render(), rt_frame_wait() / rt_frame_poll(), rt_multi_frame_wait() / _poll() / rt_multi_render(),
 * rt_launch_check_wait(), and through the device-side publish step rt_progressive_resolve() / rt_progressive_state() and
 * their rt_multi_ forms -- returns RT_ERR_DEVICE, with the numbers in rt_last_error(), unless
 *     the stamp is the launch's  &&  fetched == listed  &&  written == listed  &&  blocks done == blocks of the frame
 *     &&  no audited tap (rt_tuning.audit_known_taps) contradicts csrc/rt_lit.h.
 * RT_CANCELLED (rt_cancel() cut the launch short) takes precedence: that frame is incomplete on request.  The contents of
 * frame_out are undefined after RT_ERR_DEVICE.  An interactive pass that is incomplete is not published (its weight is not
 * counted either) and the error is reported by the next call that looks at the count.  Cost: nothing in the rounds (a wave's
 * pixels are its streams' drained-slot counters at exit), one report per workgroup and per dequeue line when waves leave, and
 * the copy that already fetched the cancel word: 64 KB per launch through a DMA engine (the runtime does copies of up to 16 KB with
 * a kernel, which would wait for a workgroup slot behind the next persistent launch), into pinned blocks -- 64 KB x (1 +
 * RT_FRAME_SLOTS + RT_CHECK_TICKETS) per context, RT_FRAME_SLOTS x n x 64 KB more per rt_multi group.  Not measurable in a
 * frame's time (profiles/r05/ab_r04_vs_verified_launches.txt, profiles/r05/copy_beside_kernel.txt). */
typedef struct {
	int                launch_checked;      /* 0: the kernel does not account for itself (RT_KERNEL_SIMPLE, the cross-check kernel) */
	unsigned int       launch_id, stamp;    /* the launch's number; the stamp its last wave left (0: none) */
	unsigned int       cancelled;
	unsigned int       waves_left;          /* waves of the trace kernel that left */
	unsigned int       primary_blocks_expected, primary_blocks_done;
	unsigned long long pixels_listed, pixels_fetched, pixels_written;
	unsigned long long taps_audited, taps_disagreeing;
} rt_launch_report;
/* the context's most recent launch, judged: waits for it, fills *report (optional) and returns RT_OK / RT_CANCELLED / RT_ERR_DEVICE */
RT_API int rt_last_launch_report(rt_context *ctx, rt_launch_report *report);
/* For hosts that enqueue launches themselves (rt_render_device() + their own collective: one process per GPU): ticket t of
 * RT_CHECK_TICKETS takes the control words of the context's most recent launch -- rt_launch_check_submit() enqueues their
 * copy on `hip_stream`, which the caller has ordered behind that launch (not the launch's own stream: a copy between two
 * kernels there costs the overlap of consecutive launches) -- and rt_launch_check_wait() waits for the copy and judges.  A
 * ticket holds one launch at a time; at most one ticket per launch. */
#define RT_CHECK_TICKETS 8
RT_API int rt_launch_check_submit(rt_context *ctx, int ticket, void *hip_stream);
RT_API int rt_launch_check_wait(rt_context *ctx, int ticket, rt_launch_report *report);

/* ---- several GPUs of one node, one host process: replaces start_workers()'s fan-out (main.c:695-718) ----
 * rt_multi_create() makes one context per listed device and, for n > 1, the RCCL communicators of the group
 * (librccl.so is loaded on first use; a single-device handle never touches it).  The setters broadcast to
 * every device.  rt_multi_render() renders the frame with the interleaved row-block partition of
 * rt_render_device() -- device i takes strip rt_strip_of_rank(i, n), i.e. the row blocks b with b % n == (i + n - 1) % n: the
 * first device, which also gathers and de-interleaves, has the last strip, never the longest; all devices at once -- gathers the
 * strips on the first device with ONE ncclGather over xGMI, de-interleaves them there and returns the frame
 * in host memory exactly as rt_render() does; params->rank / world are ignored.  Frames are bit-identical
 * to rt_render()'s for every n. */
typedef struct rt_multi rt_multi;
RT_API int  rt_multi_create(rt_multi **out, const int *device_ids, int n);
RT_API void rt_multi_destroy(rt_multi *m);
RT_API int  rt_multi_size(const rt_multi *m);
RT_API rt_context *rt_multi_context(rt_multi *m, int i);       /* device i's context (owned by the handle) */
RT_API int  rt_multi_set_scene(rt_multi *m, const Scene *scene);
RT_API int  rt_multi_set_skybox(rt_multi *m, const Cubemap *skybox);
RT_API int  rt_multi_set_camera(rt_multi *m, const rt_camera *camera);
RT_API int  rt_multi_set_tuning(rt_multi *m, const rt_tuning *tuning);
RT_API int  rt_multi_compile_scene(rt_multi *m);
RT_API int  rt_multi_render(rt_multi *m, const rt_render_params *params, Vector3 *frame_out);
/* The same with frames in flight (rt_frame_submit / rt_frame_wait above, same rules): every device renders frame k's
 * strip on its context's streams in rotation, into one of RT_LAUNCH_SETS + 1 strip buffers; the grouped ncclGather of frame k, the
 * de-interleave on the first device and the copy to frame_out run on streams of their own behind events, beside the
 * renders of the frames after it -- a render stream only ever waits for the gather RT_LAUNCH_SETS + 1 frames back, whose strip buffer
 * it reuses.  (The persistent trace kernel of the next frame holds every compute unit until it drains, so a render
 * stream that waited for the previous frame's collective would lose the overlap of consecutive strips.)  With one
 * device and no rt_test_knobs.force_collective (rt_hip_testing.h) this is rt_frame_submit() on that device's context. */
RT_API int  rt_multi_frame_submit(rt_multi *m, const rt_render_params *params, int slot, Vector3 *frame_out);
RT_API int  rt_multi_frame_wait(rt_multi *m, int slot);
RT_API int  rt_multi_frame_poll(rt_multi *m, int slot);
/* rt_frame_submit_device() for the group: the assembled frame stays on the FIRST device (*d_frame: height * width Vector3 in
 * frame order; *hip_event: a hipEvent_t of that device recorded behind the de-interleave); nothing but the launches' control
 * words goes to the host. */
RT_API int  rt_multi_frame_submit_device(rt_multi *m, const rt_render_params *params, int slot, void **d_frame, void **hip_event);
/* What the group's RCCL communicator itself says (not what the host asked for): *ranks = ncclCommCount, devices[i] =
 * ncclCommCuDevice of rank i's communicator (room for 64), *version = ncclGetVersion (e.g. 22707).  A group that has no
 * communicator -- one device without rt_test_knobs.force_collective (rt_hip_testing.h) -- reports *ranks = 0. */
RT_API int  rt_multi_collective_info(rt_multi *m, int *ranks, int devices[64], int *version);

/* ---- progressive accumulation: the reference's interactive protocol ----------------------------
 * worker() renders passes of 1 sample per (low-resolution) pixel, starting at 1/init_scale resolution
 * and doubling it after every published pass; each pass is added into `accum` with weight 1/scale^2
 * and update_frame() shows accum / sum-of-weights (main.c:354-408, 450-482).  A camera move calls
 * invalidate_accumulation() (main.c:115-124).  Here the same protocol, one pass per call:
 *
 *   rt_progressive_begin(ctx, w, h, init_scale, max_bounces, seed);   // realloc_frame_buffer()
 *   for (;;) { rt_progressive_pass(ctx, &weight);                     // one worker iteration
 *              rt_progressive_resolve(ctx, frame);                    // update_frame()
 *              if (camera moved) { rt_set_camera(...); rt_progressive_invalidate(ctx); } }
 *
 * Pass number p (since the last invalidation) seeds the path of low-resolution pixel (i, j) at scale s
 * with rt_path_seed(seed, (j*s)*w + i*s, p).  init_scale must be 1, 2, 4, 8 or 16 (main.c:611-621). */
RT_API int rt_progressive_begin(rt_context *ctx, int width, int height, int init_scale, int max_bounces, uint64_t seed);
RT_API int rt_progressive_pass(rt_context *ctx, float *weight_out);
/* `count` worker iterations in as few launches as the ladder allows: below full resolution one launch per pass, as above; at
 * full resolution up to RT_PROGRESSIVE_BATCH passes per launch -- the trace kernel adds a pixel's samples, in pass order, onto
 * the sums so far, which is what that many publish steps (main.c:394-396) do one after the other.  Sums, count and sample
 * numbers are those of `count` calls of rt_progressive_pass(): the resolved frame is bit-identical.  A GPU renders a 1080p
 * pass of one sample per pixel in 0.18 ms (consecutive passes overlap on the context's streams; their publish steps run in pass
 * order), most of it waves running out their last few paths; a host that shows a frame every 16 ms gets twice the samples out of
 * rt_progressive_passes(ctx, n) between two of them.
 * (A launch cut short by rt_cancel() publishes none of its passes; their sample numbers stay unused.) */
#define RT_PROGRESSIVE_BATCH 256
#define RT_PROGRESSIVE_BATCH_MIN 8        /* fewer passes than this are launched one by one (faster: profiles/r04/progressive_rate.txt) */
RT_API int rt_progressive_passes(rt_context *ctx, int count);
/* (RT_ERR_STATE when nothing has been published yet -- every pass so far was cut short --: frame_out is then untouched) */
RT_API int rt_progressive_resolve(rt_context *ctx, Vector3 *frame_out);
RT_API int rt_progressive_invalidate(rt_context *ctx);
RT_API int rt_progressive_state(rt_context *ctx, int *next_scale, float *count, uint32_t *generation, int *passes);
/* (`count` is read from the device, where rt_accumulate keeps it beside the accumulation buffer: a pass that a bare
 * rt_cancel() cut short -- without rt_progressive_invalidate() -- is neither added nor counted, main.c:382.  The ladder
 * and the sample numbering go on: that pass's sample number stays unused.)
 *
 * One rank of several (hosts with one process per GPU and a collective of their own): rt_progressive_begin_rank() makes
 * the context accumulate only the frame rows of the row blocks b with b % world == rank -- blocks of
 * RT_PROGRESSIVE_ROW_BLOCK frame rows, a multiple of every scale of the ladder, so that a low-resolution row never
 * straddles two ranks; rt_progressive_pass() then renders the low-resolution rows that cover them.  rt_progressive_resolve()
 * returns those rows, resolved: rt_strip_rows(height, RT_PROGRESSIVE_ROW_BLOCK, world) x width Vector3 (padding rows are
 * zero); rt_progressive_resolve_device() leaves them in device memory instead (*d_rows, owned by the context, valid until
 * the next resolve; enqueued on the context's stream, no sync) for the host's gather + rt_deinterleave_device().
 *
 * The same protocol on a device group (all workers run the ladder, main.c:354-408): every device accumulates the frame
 * rows of its own row blocks (16 frame rows each, dealt round-robin) pass after pass with nothing exchanged; a displayed
 * frame costs one resolve per device, ONE gather to the first device, a de-interleave and the copy to the host.  Frames
 * are bit-identical to the single-device ladder's. */
#define RT_PROGRESSIVE_ROW_BLOCK 16
RT_API int rt_progressive_begin_rank(rt_context *ctx, int width, int height, int init_scale, int max_bounces, uint64_t seed, int rank, int world);
RT_API int rt_progressive_resolve_device(rt_context *ctx, void **d_rows);
RT_API int rt_multi_progressive_begin(rt_multi *m, int width, int height, int init_scale, int max_bounces, uint64_t seed);
RT_API int rt_multi_progressive_pass(rt_multi *m, float *weight_out);
RT_API int rt_multi_progressive_passes(rt_multi *m, int count);       /* rt_progressive_passes() on every device */
RT_API int rt_multi_progressive_resolve(rt_multi *m, Vector3 *frame_out);
RT_API int rt_multi_progressive_invalidate(rt_multi *m);
RT_API int rt_multi_progressive_state(rt_multi *m, int *next_scale, float *count, uint32_t *generation, int *passes);

/* ---- measurement --------------------------------------------------------------------------- */
/* When enabled, every rt_render_device()/rt_render() brackets its kernels (rt_primary_pass and the trace kernel) with
 * hipEvents on the launch stream, the first one behind the clearing of the launch counters -- that is when a launch
 * that overlaps its predecessor (rt_stream) has got its first compute unit; rt_profile_collect() synchronises and
 * returns the summed time and the number of launches since the last collect.  (The time two overlapping launches are
 * on the GPU together is counted in both.) */
RT_API int rt_profile_enable(rt_context *ctx, int on);
RT_API int rt_profile_collect(rt_context *ctx, double *kernel_ms_total, int *launches);
/* the same, plus the time from the first of those launches getting its first compute unit to the end of the last one's
 * trace kernel (no double counting of the time overlapping launches share, but idle time between launches is in it) */
RT_API int rt_profile_collect_span(rt_context *ctx, double *kernel_ms_total, int *launches, double *span_ms);
/* the same, plus the part of kernel_ms_total that was the camera-ray pass (rt_primary_pass): first compute unit -> an event
 * between the two kernels; the rest is the trace kernel.  The two are bound differently (bench.py reports them apart).  That
 * event is only recorded after rt_profile_enable(ctx, 2) -- it costs a launch about 20 us, so the plain mode (1) leaves it out and
 * reports 0 here -- and only means something for launches that have the GPU to themselves: a launch that overlaps its
 * predecessor waits for workgroup slots between its two kernels. */
RT_API int rt_profile_collect_split(rt_context *ctx, double *kernel_ms_total, int *launches, double *span_ms, double *primary_ms_total);


/* Where a step of the N-GPU frame loop goes, per device -- so that a scaling run that falls short says WHY.  After
 * rt_multi_profile_enable(m, 1) every rt_multi_frame_submit() brackets each phase of the frame on each device with timed events;
 * rt_multi_profile_collect() waits for the frames in flight and fills per_device[0 ... rt_multi_size(m)).  Several frames are in
 * flight, so every instant between the device's first and last frame END (first device: frame in host memory; the others: their
 * part of the gather done) is given to what the device was doing then, whichever frame it was for, by priority: a strip render
 * in progress (incl. its wait for workgroup slots) > the de-interleave > the copy to the host (these two: first device only) > a
 * rendered strip waiting for its gather > idle (no launch was there to run).  The five shares are disjoint, sum to step_ms --
 * the mean interval between two frame ends -- and are ms per step.  Reading: a device that renders all of its step bounds the
 * frame rate; one that mostly waits for the gather has slack; idle time is the host's.  Costs two to five event records per
 * device and frame; collect resets the log (which stops growing by itself after 4 096 frames).  (One device without the collective path records nothing: frames == 0.)
 * main.c:695-718 is the fan-out this instruments. */
typedef struct {
	int    frames;                      /* intervals the means are over */
	double step_ms;
	double idle_ms, render_ms, gather_ms, deinterleave_ms, copy_ms;
} rt_multi_phases;
RT_API int rt_multi_profile_enable(rt_multi *m, int on);
RT_API int rt_multi_profile_collect(rt_multi *m, rt_multi_phases *per_device, int capacity);

/* ---- host-side mirror of the reference's loaders / camera (plain C, no GPU needed) ---------- */
/* scene.c:611 parse_scene_file(): same grammar, defaults, range checks, float accumulation and
 * stderr diagnostics.  Returns RT_OK / RT_ERR_IO / RT_ERR_FORMAT. */
RT_API int rt_parse_scene_file(const char *file, Scene *scene);
RT_API int rt_parse_scene_string(const char *src, size_t len, Scene *scene);

/* gpu_and_windowing.c:24-40 load_cubemap()/free_cubemap(): decodes six baseline JPEGs with the
 * arithmetic of stb_image v2.29 (the decoder the reference vendors), so texel bytes are identical.
 * `files` is indexed by CubeFace.  Unlike the reference it returns an error instead of abort(). */
RT_API int  rt_load_cubemap(Cubemap *c, const char *files[6]);
RT_API void rt_free_cubemap(Cubemap *c);
/* one image: *out is malloc()ed w*h*chan bytes */
RT_API int  rt_decode_jpeg_file(const char *file, uint8_t **out, int *w, int *h, int *chan);

/* camera.c:28,33-35 defaults; camera.c:99-118 frame constants (tan() in double on the host) */
RT_API void rt_camera_default(rt_camera *cam);
RT_API void rt_camera_basis_for(const rt_camera *cam, float aspect_ratio, rt_camera_basis *out);
/* camera.c:80-88 move_camera(), camera.c:42-78 rotate_camera() on explicit state */
typedef enum { RT_DIR_UP, RT_DIR_DOWN, RT_DIR_LEFT, RT_DIR_RIGHT } rt_direction;
typedef struct { int first_mouse; float yaw, pitch, last_x, last_y; } rt_mouse_state;
RT_API void rt_mouse_state_default(rt_mouse_state *m);
RT_API void rt_move_camera(rt_camera *cam, rt_direction dir, float speed);
RT_API void rt_rotate_camera(rt_camera *cam, rt_mouse_state *m, double mouse_x, double mouse_y);

/* `counter`-mode path seed (shared definition with the kernels and the oracle) */
RT_API uint64_t rt_path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index);

/* Headless stand-in for the reference's presenter hand-off, same signature as
 * move_frame_to_the_gpu() (gpu_and_windowing.h:44): a registered sink receives the frame. */
typedef void (*rt_frame_sink)(int w, int h, Vector3 *data, void *user);
RT_API void rt_set_frame_sink(rt_frame_sink sink, void *user);
RT_API void rt_move_frame_to_the_gpu(int w, int h, Vector3 *data);
/* screenshot() main.c:637-681: float -> u8 by truncating *255, vertical flip; written as binary PPM */
RT_API int  rt_write_ppm(const char *file, int w, int h, const Vector3 *data);
/* the same conversion into an RGB8 PNG (what screenshot() produces through stb_image_write), and
 * screenshot() itself: first free "screenshot_<n>.png" in the working directory (main.c:642-659) */
RT_API int  rt_write_png(const char *file, int w, int h, const Vector3 *data);
RT_API int  rt_screenshot(int w, int h, const Vector3 *data, char *name_out, size_t name_cap);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */
