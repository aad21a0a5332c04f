/*
 * rt_api.cpp -- the C ABI of librt_hip.so (see include/rt_hip.h): context, input packing,
 * launches.  Host-only code; the kernels are in rt_kernels.hip.
 *
 * Packing folds every ray-independent term of the reference's inner loops on the host, with the
 * reference's own float roundings, so the kernels start from the same bits the CPU would compute:
 *   cube far corner  origin*1 + size*1                         scene.c:27
 *   sphere r*r                                                  scene.c:112
 *   f0, 1-f0, albedo*(1-metallic), emission_color*power, metallic > 0.001
 *                                                               main.c:219-221,128,248,232,241
 *   first emitter and origin_of() of it                         main.c:140-146, scene.c:10-15
 *   camera basis                                                camera.c:99-118
 * This translation unit is compiled with -ffp-contract=off like everything else.
 */
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_internal.h"
#include "rt_pack.h"
#include "rt_cull.h"
#define RT_LIT_FN static inline
#include "rt_lit.h"

#pragma clang fp contract(off)

static thread_local char g_error[512] = "";

static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	return code;
}

int rt_fail(int code, const char *fmt, ...)      /* for the library's other translation units (rt_internal.h) */
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                       \
	do {                                                                                    \
		hipError_t e_ = (expr);                                                             \
		if (e_ != hipSuccess)                                                               \
			return fail(RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

/* the context's pinned host words (h_words), in blocks of RT_CTL_COPY_WORDS (rt_internal.h): block 0 takes the control words of rt_render()'s launch,
 * blocks 1 ... RT_FRAME_SLOTS those of the frames in flight, the next RT_CHECK_TICKETS the tickets of rt_launch_check_*(); behind
 * them rt_cancel()'s request (read by the trace kernels through rt_launch.stop) and the ladder's count words */
#define RT_WORDS_FRAME(slot)   ((1 + (slot)) * RT_CTL_COPY_WORDS)
#define RT_WORDS_TICKET(t)     ((1 + RT_FRAME_SLOTS + (t)) * RT_CTL_COPY_WORDS)
#define RT_STOP_WORD           ((1 + RT_FRAME_SLOTS + RT_CHECK_TICKETS) * RT_CTL_COPY_WORDS)
#define RT_COUNT_WORD          (RT_STOP_WORD + 16)
#define RT_HOST_WORDS          (RT_COUNT_WORD + 16)

struct rt_context {
	int          device = 0;
	hipStream_t  stream = nullptr;

	rt_geom     *d_geom = nullptr;
	rt_shade    *d_shade = nullptr;
	int          num_objects = 0;
	int          capacity = 0;
	bool         have_scene = false;
	std::vector<rt_geom> h_geom;         /* host copy of the packed geometry (rt_compile_scene) */
	hipModule_t  spec_module = nullptr;  /* scene-specialised kernel, valid until the scene changes */
	std::vector<char> spec_code;         /* its code object (rt_spec_symbol_read("") hands it out) */
	std::string  spec_compiler;          /* where it came from (rt_compiled_scene_info) */
	hipFunction_t spec_fn = nullptr;
	hipFunction_t spec_fn_audit = nullptr;  /* the same scene's kernel in its audit variant (rt_tuning.audit_known_taps), built when first asked for; valid like spec_fn */
	bool         spec_audit_failed = false; /* ... could not be built (no hiprtc): audited launches use the generic kernel's audit variant */
	bool         spec_audit_not_embedded = false;   /* ... does not come with the library: the background audit, which compiles nothing, uses the generic kernel's */
	bool         scene_fast_ok = false;  /* every cube has 0 <= size, plane coordinates +0 or 2^-76 <= |x| <= 2^29, sphere data |x| <= 2^29 */
	int          light_index = -1;
	float        light_pos[3] = {0, 0, 0};
	bool         only_light_emits = false;   /* no other object has a non-zero emission component (rt_device.h) */
	/* rt_lit.h: which hit points need no soft-shadow tap traced -- one bit per cell of a grid over every object, built
	 * by rt_set_scene on the host (a few milliseconds) */
	unsigned char *d_lit_cells = nullptr;     /* the lit-taps table as built, and behind it (at + lit_cells_capacity) the copy that carries the audit's marks */
	std::vector<unsigned char> h_lit_cells;   /* the table as built (values 1 / 2 / 0), for re-marking when the audit setting changes */
	int          lit_audit_marked = 0;        /* the rt_tuning.audit_known_taps the device's copy is marked for (bit 2 of one cell in 2^k) */
	void        *d_lit_grids = nullptr;
	size_t       lit_cells_capacity = 0;
	int          lit_grids_capacity = 0;
	bool         have_lit = false;
	long long    primary_passes = 0;     /* launches that ran rt_primary_pass (rt_primary_passes_run) */
	uint64_t     input_version = 1;      /* bumped by rt_set_scene / rt_set_skybox: part of launch_slot::lists_key */
	/* large scenes (rt_cull.h): clusters of close objects with conservative boxes; num_clusters = 0: every ray tests every object */
	rt_cluster  *d_clusters = nullptr;
	int          clusters_capacity = 0;
	rt_cull_info cull = { 0, 0.0f, 0.0f };

	uint32_t    *d_sky = nullptr;
	size_t       sky_bytes = 0;
	int          sky_w = 0, sky_h = 0;
	bool         have_sky = false;

	rt_camera    camera;
	bool         have_camera = false;

	/* Scratch of a launch.  There are RT_LAUNCH_SETS sets, used in rotation, so that consecutive launches enqueued on
	 * different streams can be on the GPU together: the waves of the next fill the compute units as the waves of the
	 * first run out of pixels (the last wave of a launch leaves 100 us after the average one).  With two
	 * sets, launch n + 2 had to wait for the END of launch n, whose scratch it took over -- and for all of launch n's tail the
	 * workgroup slots its waves left stayed empty beside launch n + 1 (a strip of an eighth of a frame: 0.748 ms per step against
	 * 0.672 ideal); with three, launch n + 2 is resident while launch n drains (0.720); with five, small launches take ONE workgroup
	 * slot per CU each and five are resident or queued for the four slots (workgroups_per_cu_for(): 0.690). */
	struct launch_slot {
		unsigned int *d_counter = nullptr;   /* dequeue + fill counters of the pixel lists, launch control words */
		float       *d_pix = nullptr;        /* rt_primary_pass output: pixel records (rt_device.h) */
		size_t       pix_capacity = 0;       /* records it holds */
		hipEvent_t   done = nullptr;         /* recorded behind the last launch that used the set, on that launch's stream */
		hipEvent_t   started = nullptr;      /* recorded behind that launch's primary pass, i.e. in front of its trace kernel */
		hipStream_t  stream = nullptr;
		bool         used = false;
		uint64_t     lists_key = 0;          /* what rt_primary_pass's output in this set belongs to (0: nothing reusable) */
		hipEvent_t   readback = nullptr;     /* behind the copy of this set's control word to the host (rt_context_read_control) */
		hipStream_t  readback_stream = nullptr;
		bool         readback_pending = false; /* ... which the set's next launch, which clears the word, has to wait for */
		rt_launch_expect expect = { 0u, 0, 0u };   /* what the set's most recent launch must leave in its control words (rt_judge_launch) */
	} slot[RT_LAUNCH_SETS];
	unsigned     launches = 0;           /* launch n uses slot[n % RT_LAUNCH_SETS] */
	unsigned int last_id = 0;            /* launch numbers handed out so far (prepare_launch): a number is never used twice */
	int          cur = 0;                /* set of the most recent launch */
	hipStream_t  more_streams[RT_LAUNCH_SETS - 1] = {};   /* rt_stream(ctx, 1 ...): made on first request */
	std::once_flag more_streams_once[RT_LAUNCH_SETS - 1];
	std::atomic<unsigned int> enqueued{0}; /* for rt_cancel() on another thread: the launches announced so far are numbers 1 ... enqueued */
	unsigned int *h_words = nullptr;     /* pinned, RT_HOST_WORDS of them (layout above); [RT_STOP_WORD] = rt_cancel()'s request: "launches up to this number stop" */
	unsigned int *d_stop = nullptr;      /* the device's address of that word */
	int          num_cus = 256;

	float       *d_frame = nullptr;      /* scratch for rt_render() */
	size_t       frame_bytes = 0;

	/* frames in flight (rt_frame_submit / rt_frame_wait): a device frame per slot, one copy stream for all */
	struct frame_slot {
		float     *d_buf = nullptr;
		size_t     bytes = 0;
		hipEvent_t copied = nullptr;     /* behind the copy of the frame (and of the launch's control word) to the host */
		hipEvent_t rendered = nullptr;   /* behind the render, on its stream: what rt_frame_submit_device() hands to the caller */
		bool       busy = false;         /* submitted and not waited for yet */
		rt_launch_expect expect = { 0u, 0, 0u };
	} fq[RT_FRAME_SLOTS];
	struct check_ticket {                /* rt_launch_check_submit / _wait */
		hipEvent_t copied = nullptr;
		bool       busy = false;
		rt_launch_expect expect = { 0u, 0, 0u };
	} tickets[RT_CHECK_TICKETS];
	unsigned long long frames_submitted = 0;   /* frame n renders on the context's stream n % RT_LAUNCH_SETS */
	hipStream_t  copy_stream = nullptr;  /* made by the first rt_frame_submit() */

	/* progressive accumulation (rt_progressive_*) */
	struct {
		bool     active = false;
		int      width = 0, height = 0, init_scale = 1, scale = 1, max_bounces = 10, passes = 0;
		int      rank = 0, world = 1, rows = 0;   /* this context accumulates the frame rows of the row blocks b % world == rank: `rows` of them */
		uint64_t seed = 0;
		uint32_t generation = 0;
		float   *d_accum = nullptr, *d_out = nullptr;
		float   *d_low[RT_LAUNCH_SETS] = {};  /* a pass's low-resolution frame, one per scratch set: pass n + 1 renders while pass n is being added to the sums */
		/* the sums and their count are read and written by the publish steps of passes that render on different streams (and by
		 * invalidate / resolve on the context's own): whoever touches them waits for the last one that did, and says so */
		hipEvent_t  sums_touched = nullptr;
		hipStream_t sums_stream = nullptr;
		bool        sums_used = false;
		float   *d_count = nullptr;          /* RT_COUNT_WORDS words (rt_device.h): the sum of the published passes' weights (accum_counts[], main.c:396), written by
		                                      * rt_accumulate; a word that stays zero; the launches not published because they were incomplete */
		size_t   accum_bytes = 0, low_bytes = 0;
	} prog;

	rt_knobs     tuning;                 /* rt_set_tuning() + rt_set_test_knobs() */
	std::string  jit_flags;              /* rt_tuning.jit_flags, copied */

	bool         profiling = false;
	bool         profiling_split = false;   /* rt_profile_enable(ctx, 2): also an event BETWEEN the camera-ray pass and the trace kernel (it costs a launch ~20 us) */
	struct launch_events { hipEvent_t first, mid, second; };   /* first compute unit; between the camera-ray pass and the trace kernel; end of the trace kernel */
	std::vector<launch_events> events;
	std::vector<hipEvent_t> event_pool;
};

/* Everything this context has enqueued -- on its own streams or on a caller's -- has finished. */
static int wait_for_launches(rt_context *ctx)
{
	for (auto &sl : ctx->slot)
		if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if (ctx->copy_stream) HIP_TRY(hipStreamSynchronize(ctx->copy_stream));   /* frames on their way to the host */
	return RT_OK;
}

/* Is the previous launch the only unfinished one ahead of the launch about to be enqueued? */
static bool lone_launch_ahead(rt_context *ctx)
{
	if (ctx->launches < 2) return true;
	const rt_context::launch_slot &before = ctx->slot[(ctx->launches - 2) % RT_LAUNCH_SETS];
	if (!before.used) return true;
	const hipError_t d = hipEventQuery(before.done);
	(void) hipGetLastError();           /* "not ready" is an answer, not an error the launch below should find */
	return d != hipErrorNotReady;
}

/* Called before launch number ctx->launches is enqueued on `stream`: launch n - RT_LAUNCH_SETS used the same scratch set,
 * so this one is ordered behind it when that ran on a different stream.  (The launches in between have the other sets:
 * they may overlap with this one.) */
static int order_behind_previous(rt_context *ctx, hipStream_t stream)
{
	rt_context::launch_slot &sl = ctx->slot[ctx->launches % RT_LAUNCH_SETS];
	if (sl.used && sl.stream != stream) HIP_TRY(hipStreamWaitEvent(stream, sl.done, 0));
	/* ... and it starts when the previous launch's trace kernel is next in line on its stream: that kernel's workgroups
	 * take the whole chip first, and this launch gets the compute units as they leave.  (Two launches that become ready
	 * together would share the chip half and half from start to end, and finish together: nothing gained, and the late
	 * half of either grid finds no pixels left.) */
	rt_context::launch_slot &prev = ctx->slot[ctx->cur];
	if (ctx->launches && prev.used && prev.stream != stream) HIP_TRY(hipStreamWaitEvent(stream, prev.started, 0));
	/* ... and a frame in flight must have read them (rt_frame_submit) */
	if (sl.readback_pending) { sl.readback_pending = false; if (sl.readback_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, sl.readback, 0)); }
	return RT_OK;
}

/* Resident workgroups per CU for the launch about to be enqueued on `stream`.  rt_tuning.workgroups_per_cu, if set.  Else: a
 * launch enqueued while the previous one, on the context's OTHER stream, has not even started takes half the chip's
 * workgroup slots (2 of 4 per CU).  In a run of such launches two are resident side by side, half a launch apart, and the
 * compute units a launch's last waves leave idle belong to waves of the next one that are already there, not to workgroups
 * that have yet to start (C1: strips of 8 ranks -4.6 %, of 4 -4.4 %, of 2 -3 %, whole frames the same; scripts/probes/wg_probe.py).
 * It only pays when the host keeps two launches resident at all times, i.e. enqueues more than a launch ahead -- the N-GPU
 * loops do (three frames in flight) -- and costs when it does not: a loop with two frames in flight submits frame k+2 when
 * frame k has been delivered, and until then frame k+1 would have half a chip to itself (C1 whole frames +2.7 %).  That is
 * what the test tells apart: such a host finds the previous launch running.  A launch with nothing beside it (rt_render(),
 * the first frame of a run, every launch of a host that uses one stream) takes all the slots. */
/* Round 5: ... and ONE slot per CU when the launch is small and the host keeps at least four more in flight (the launch four
 * before this one has not finished: the N-GPU loops rotate five strips through the five scratch sets).  A strip of one of eight
 * ranks is five object pixels per stream of a full grid: a wave spends a fifth of its life on the last of them, at thinning lanes,
 * and with a quarter of the waves per launch -- five launches resident or queued for the four slots -- that weighs a quarter.  C1,
 * strips of 8 / 4 / 2 ranks: 0.716 / 1.381 / 2.630 -> 0.682 / 1.344 / 2.620 ms per step; a whole frame loses 1.5 % that way and
 * keeps two slots (profiles/r05/strip_loop_probe_five_sets.txt).  `pixels`: those of the launch, sky included. */
#define RT_SMALL_LAUNCH_PIXELS_PER_STREAM 64
static int workgroups_per_cu_for(rt_context *ctx, hipStream_t stream, long long pixels)
{
	if (ctx->tuning.workgroups_per_cu > 0) return ctx->tuning.workgroups_per_cu;
	/* A culled scene's launches never share the chip: a large one runs as ONE workgroup of twelve waves per CU (one copy of the cluster
	 * records, three waves per SIMD), and any other number of slots puts it back on workgroups of four -- a third of the waves.  With
	 * three L1024 frames in flight the launches given half the slots took 24 ms each once nothing came behind them, where a frame takes
	 * 6.4 (profiles/r06/L1024_depth.txt). */
	if (ctx->cull.num_clusters > 0 && !ctx->tuning.test_every_object && !ctx->spec_fn && ctx->scene_fast_ok) return 0;
	const rt_context::launch_slot &prev = ctx->slot[ctx->cur];
	if (!ctx->launches || !prev.used || prev.stream == stream) return 0;
	/* Two launches unfinished ahead of this one -- the previous one and the one before it -- : the host keeps three in flight, two are
	 * (or will be) resident side by side, and this one joins them at half the slots.  Only ONE ahead (a host with two frames in
	 * flight; the second launch of any run): all the slots -- it gets them as its predecessor drains, and nothing else is there
	 * to take the rest.  (Until round 5 the test was "the previous launch has not started": the second frame of every run then
	 * ran at half the chip by itself until the third was submitted.) */
	if (lone_launch_ahead(ctx)) return 0;
	const long long streams_at_two = (long long) ctx->num_cus * 2 * 4 * 8;      /* two workgroups of four waves per CU, eight streams per wave (rt_kernels.hip) */
	if (ctx->launches >= 4 && pixels < streams_at_two * RT_SMALL_LAUNCH_PIXELS_PER_STREAM) {
		/* five launches on five streams (a host that is far ahead on fewer streams has fewer launches on the GPU: they keep two slots) */
		hipStream_t seen[5] = { stream, nullptr, nullptr, nullptr, nullptr };
		bool distinct = true;
		for (unsigned back = 1; back <= 4 && distinct; back++) {
			const rt_context::launch_slot &b = ctx->slot[(ctx->launches - back) % RT_LAUNCH_SETS];
			distinct = b.used;
			for (unsigned k = 0; k < back && distinct; k++) distinct = seen[k] != b.stream;
			seen[back] = b.stream;
		}
		if (distinct) {
			const hipError_t d = hipEventQuery(ctx->slot[(ctx->launches - 4) % RT_LAUNCH_SETS].done);
			(void) hipGetLastError();
			if (d == hipErrorNotReady) return 1;
		}
	}
	return 2;
}

/* Does rt_primary_pass classify the object pixels' taps (rt_lit.h: rt_taps_class)?  It looks at every object for every object
 * pixel -- once per launch --, and what it saves is taps not traced -- per sample.  Where every ray tests every object anyway a
 * tap costs as much as the classification, and it pays from the second sample on.  A culled scene traces a tap in far fewer
 * steps than it has objects: there the camera-ray pass took 0.86 / 1.57 / 2.72 ms with the classification against 0.14 / 0.23 /
 * 0.35 ms without (256 / 512 / 1024 objects, 1080p), for 0.025 ... 0.05 ms less tracing per sample per pixel: it pays from
 * objects / 9 ... objects / 22 samples on (profiles/r04/classify_probe.txt); the library asks for objects / 8. */
#define RT_CLASSIFY_OBJECTS_PER_SAMPLE 8
static int classify_pixels(const rt_context *ctx, int samples)
{
	if (ctx->tuning.trace_known_taps || samples < 2) return 0;
	const bool culled = ctx->cull.num_clusters > 0 && !ctx->tuning.test_every_object && !ctx->spec_fn && ctx->scene_fast_ok;
	return !culled || (long long) samples * RT_CLASSIFY_OBJECTS_PER_SAMPLE >= (long long) ctx->num_objects;
}

static int mark_launch(rt_context *ctx, hipStream_t stream)
{
	rt_context::launch_slot &sl = ctx->slot[ctx->launches % RT_LAUNCH_SETS];
	HIP_TRY(hipEventRecord(sl.done, stream));
	sl.stream = stream; sl.used = true;
	ctx->cur = (int) (ctx->launches % RT_LAUNCH_SETS);
	ctx->launches++;               /* (its number, ctx->last_id, was published by prepare_launch() before the launch's first kernel) */
	return RT_OK;
}

static hipStream_t pick_stream(rt_context *ctx, void *hip_stream)
{
	if (hip_stream == RT_STREAM_LEGACY) return nullptr;             /* the device's legacy null stream */
	return hip_stream ? (hipStream_t) hip_stream : ctx->stream;
}

void *rt_context_stream(rt_context *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

/* The control words of the context's most recent launch (rt_device.h RT_CTL_*) are copied to h_dst[0 ... RT_CTL_COPY_WORDS) (pinned; the first RT_CTL_WORDS matter)
 * on `stream`, which the caller has already ordered behind that launch -- NOT the launch's own stream: a copy between
 * two kernels of a render stream costs the overlap of consecutive launches (measured: +0.15 ms per C1 frame).  The
 * scratch set's next launch clears the word: it is ordered behind this copy.  *behind (optional) = an event recorded
 * behind the copy. */
int rt_context_read_control(rt_context *ctx, unsigned int *h_dst, hipStream_t stream, hipEvent_t *behind, rt_launch_expect *expect)
{
	rt_context::launch_slot &sl = ctx->slot[ctx->cur];
	if (!sl.readback) HIP_TRY(hipEventCreateWithFlags(&sl.readback, hipEventDisableTiming));
	if (expect) *expect = sl.expect;
	HIP_TRY(hipMemcpyAsync(h_dst, sl.d_counter + 128 * 32, RT_CTL_COPY_BYTES, hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipEventRecord(sl.readback, stream));
	sl.readback_pending = true; sl.readback_stream = stream;
	if (behind) *behind = sl.readback;
	return RT_OK;
}

/* The verdict on a launch from its control words (rt_internal.h). */
int rt_judge_launch(const unsigned int *w, const rt_launch_expect &x, const char *who, rt_launch_report *report)
{
	rt_launch_report r;
	memset(&r, 0, sizeof(r));
	r.launch_checked = x.stamped; r.launch_id = x.launch_id; r.stamp = w[RT_CTL_STAMP]; r.cancelled = w[RT_CTL_CANCELLED];
	r.waves_left = w[RT_CTL_WAVES_LEFT];
	r.primary_blocks_expected = x.primary_blocks; r.primary_blocks_done = w[RT_CTL_PRIMARY];
	r.pixels_listed = w[RT_CTL_LISTED]; r.pixels_fetched = w[RT_CTL_FETCHED]; r.pixels_written = w[RT_CTL_WRITTEN];
	r.taps_audited = (unsigned long long) w[RT_CTL_AUDITED] | ((unsigned long long) w[RT_CTL_AUDITED + 1] << 32);
	r.taps_disagreeing = (unsigned long long) w[RT_CTL_DISAGREE] | ((unsigned long long) w[RT_CTL_DISAGREE + 1] << 32);
	if (report) *report = r;
	/* the stamp first: until a launch's own last wave has written the line, `cancelled` may be what the scratch set's PREVIOUS launch
	 * left there (launches that keep the set's pixel lists clear nothing); a launch cut short by rt_cancel() still ends with its stamp */
	if (x.stamped && r.stamp != x.launch_id)
		return fail(RT_ERR_DEVICE, "%s: launch %u ended without the stamp of its last wave (stamp %u, %u waves left it, %llu pixels written): the frame is incomplete",
		            who, x.launch_id, r.stamp, r.waves_left, r.pixels_written);
	if (r.cancelled) return RT_CANCELLED;
	if (!x.stamped) return RT_OK;
	if (r.pixels_fetched != r.pixels_listed || r.pixels_written != r.pixels_listed || r.primary_blocks_done != r.primary_blocks_expected)
		return fail(RT_ERR_DEVICE, "%s: launch %u is incomplete: object pixels listed %llu, fetched %llu, written %llu; camera-ray blocks %u of %u",
		            who, x.launch_id, r.pixels_listed, r.pixels_fetched, r.pixels_written, r.primary_blocks_done, r.primary_blocks_expected);
	if (r.taps_disagreeing)
		return fail(RT_ERR_DEVICE, "%s: launch %u: %llu of %llu audited soft-shadow taps contradict the answer rt_lit.h gave without tracing them: the frame is wrong",
		            who, x.launch_id, r.taps_disagreeing, r.taps_audited);
	return RT_OK;
}

/* an event recorded behind the context's most recent launch, on that launch's stream */
void *rt_context_launch_done(rt_context *ctx) { return ctx ? (void *) ctx->slot[ctx->cur].done : nullptr; }

extern "C" void *rt_stream(rt_context *ctx, int which)
{
	if (!ctx || which < 0 || which >= RT_LAUNCH_SETS) return nullptr;
	if (which == 0) return (void *) ctx->stream;
	/* (made by rt_create(), see there) with the lowest priority
	 * the device offers: streams of different priority never share a queue, so their launches overlap those of stream 0
	 * (and each other's: the runtime deals the streams of one priority over several hardware queues). */
	hipStream_t &st = ctx->more_streams[which - 1];
	std::call_once(ctx->more_streams_once[which - 1], [ctx, &st]() {
		if (hipSetDevice(ctx->device) != hipSuccess) return;
		int least = 0, greatest = 0;
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
		if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess &&
		    hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
	});
	if (!st) { fail(RT_ERR_DEVICE, "rt_stream: could not create stream %d", which); return nullptr; }
	return (void *) st;
}

extern "C" {

const char *rt_last_error(void) { return g_error; }

int rt_abi_version(void) { return RT_ABI_VERSION; }

void rt_default_tuning(rt_tuning *t) { if (t) { memset(t, 0, sizeof(*t)); t->size = sizeof(*t); } }

int rt_set_tuning(rt_context *ctx, const rt_tuning *t)
{
	if (!t) return fail(RT_ERR_ARGUMENT, "rt_set_tuning: NULL argument");
	/* (only `size` is looked at until it has been found to be this library's: a struct of another revision of the header is not
	 * read; checked before anything else, so that a host can find out with no context at all) */
	if (t->size != sizeof(rt_tuning))
		return fail(RT_ERR_ARGUMENT, "rt_set_tuning: rt_tuning.size is %zu, this library's rt_tuning has %zu bytes (ABI version %d): was the host compiled against "
		            "another revision of rt_hip.h, or the struct not initialised with rt_default_tuning()?", t->size, sizeof(rt_tuning), RT_ABI_VERSION);
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_set_tuning: NULL context");
	if ((t->dequeue_shards != 0 && t->dequeue_shards != 1 && t->dequeue_shards != 64) ||
	    t->workgroups_per_cu < 0 || t->workgroups_per_cu > 8 || t->jit_waves_per_simd < 0 || t->jit_waves_per_simd > 8 ||
	    t->audit_known_taps < RT_AUDIT_OFF || t->audit_known_taps > RT_AUDIT_MAX_LOG2)
		return fail(RT_ERR_ARGUMENT, "rt_set_tuning: value out of range");
	ctx->tuning.dequeue_shards = t->dequeue_shards; ctx->tuning.workgroups_per_cu = t->workgroups_per_cu;
	ctx->tuning.jit_waves_per_simd = t->jit_waves_per_simd; ctx->tuning.audit_known_taps = t->audit_known_taps;
	ctx->jit_flags = t->jit_flags ? t->jit_flags : "";
	return RT_OK;
}

void rt_default_test_knobs(rt_test_knobs *k) { if (k) { memset(k, 0, sizeof(*k)); k->size = sizeof(*k); } }

int rt_set_test_knobs(rt_context *ctx, const rt_test_knobs *k)
{
	if (!ctx || !k) return fail(RT_ERR_ARGUMENT, "rt_set_test_knobs: NULL argument");
	if (k->size != sizeof(rt_test_knobs))
		return fail(RT_ERR_ARGUMENT, "rt_set_test_knobs: rt_test_knobs.size is %zu, this library's has %zu bytes", k->size, sizeof(rt_test_knobs));
	if (k->test_drop_pixels < 0) return fail(RT_ERR_ARGUMENT, "rt_set_test_knobs: value out of range");
	ctx->tuning.force_collective = k->force_collective; ctx->tuning.poison_frame = k->poison_frame; ctx->tuning.trace_known_taps = k->trace_known_taps;
	ctx->tuning.test_every_object = k->test_every_object; ctx->tuning.test_drop_pixels = k->test_drop_pixels;
	ctx->tuning.test_corrupt_lit_table = k->test_corrupt_lit_table;
	return RT_OK;
}

void rt_default_params(rt_render_params *p, int width, int height, int spp, int max_bounces)
{
	if (!p) return;
	memset(p, 0, sizeof(*p));
	p->width = width; p->height = height; p->spp = spp; p->max_bounces = max_bounces;
	p->seed = 0; p->row_block = 8; p->rank = 0; p->world = 1; p->kernel = RT_KERNEL_AUTO;
}

int rt_create(rt_context **out, int device_id)
{
	if (!out) return fail(RT_ERR_ARGUMENT, "rt_create: out is NULL");
	*out = nullptr;
	int count = 0;
	HIP_TRY(hipGetDeviceCount(&count));
	if (device_id < 0 || device_id >= count)
		return fail(RT_ERR_ARGUMENT, "rt_create: device %d out of range (%d visible)", device_id, count);
	HIP_TRY(hipSetDevice(device_id));
	rt_context *ctx = new (std::nothrow) rt_context();
	if (!ctx) return fail(RT_ERR_MEMORY, "rt_create: out of host memory");
	ctx->device = device_id;
	hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete ctx; return fail(RT_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
			ctx->num_cus = prop.multiProcessorCount;
		for (auto &sl : ctx->slot) {
			if (e == hipSuccess) e = hipMalloc((void**) &sl.d_counter, rt_counter_bytes() + RT_CTL_COPY_BYTES);      /* (the copy of the control words reads that far behind them) */
			if (e == hipSuccess) e = hipMemsetAsync(sl.d_counter, 0, rt_counter_bytes() + RT_CTL_COPY_BYTES, ctx->stream);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.started, hipEventDisableTiming);
		}
		if (e == hipSuccess) e = hipHostMalloc((void**) &ctx->h_words, (size_t) RT_HOST_WORDS * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent);
		if (e == hipSuccess) { for (int k = 0; k < RT_HOST_WORDS; k++) ctx->h_words[k] = 0u; }
		if (e == hipSuccess) e = hipHostGetDevicePointer((void**) &ctx->d_stop, &ctx->h_words[RT_STOP_WORD], 0);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) {
			for (auto &sl : ctx->slot) { (void) hipFree(sl.d_counter); if (sl.done) (void) hipEventDestroy(sl.done); if (sl.started) (void) hipEventDestroy(sl.started); }
			if (ctx->h_words) (void) hipHostFree(ctx->h_words);
			(void) hipStreamDestroy(ctx->stream); delete ctx;
			return fail(RT_ERR_DEVICE, "rt_create: %s", hipGetErrorString(e));
		}
	}
	rt_camera_default(&ctx->camera);
	ctx->have_camera = true;     /* the reference starts from its default pose too (camera.c:33-35) */
	/* The render streams are made now, in order, before any other stream of the context (the copy stream of the frame queue): made
	 * on first request -- as they were until round 5 -- a host that ran the frame queue first got interactive passes that did not
	 * overlap (0.230 instead of 0.178 ms per pass; scripts/probes/progressive_reps.py first-frames): which hardware queue a
	 * stream gets depends on what exists when it is made. */
	for (int which = 1; which < RT_LAUNCH_SETS; which++) (void) rt_stream(ctx, which);
	*out = ctx;
	return RT_OK;
}

void rt_destroy(rt_context *ctx)
{
	if (!ctx) return;
	(void) hipSetDevice(ctx->device);
	(void) hipStreamSynchronize(ctx->stream);
	for (auto &p : ctx->events) { (void) hipEventDestroy(p.first); if (p.mid) (void) hipEventDestroy(p.mid); (void) hipEventDestroy(p.second); }
	for (auto &e : ctx->event_pool) (void) hipEventDestroy(e);
	for (auto &sl : ctx->slot) {
		if (sl.used) (void) hipEventSynchronize(sl.done);
		if (sl.done) (void) hipEventDestroy(sl.done);
		if (sl.started) (void) hipEventDestroy(sl.started);
		if (sl.readback) (void) hipEventDestroy(sl.readback);
		(void) hipFree(sl.d_counter); (void) hipFree(sl.d_pix);
	}
	if (ctx->copy_stream) { (void) hipStreamSynchronize(ctx->copy_stream); (void) hipStreamDestroy(ctx->copy_stream); }
	for (auto &f : ctx->fq) { if (f.copied) (void) hipEventDestroy(f.copied); if (f.rendered) (void) hipEventDestroy(f.rendered); (void) hipFree(f.d_buf); }
	for (auto &t : ctx->tickets) if (t.copied) (void) hipEventDestroy(t.copied);
	for (hipStream_t st : ctx->more_streams) if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
	if (ctx->h_words) (void) hipHostFree(ctx->h_words);
	/* (the compiled scene's module belongs to the process-wide cache of rt_jit.cpp: never unloaded) */
	(void) hipFree(ctx->d_geom); (void) hipFree(ctx->d_shade);
	(void) hipFree(ctx->d_lit_cells); (void) hipFree(ctx->d_lit_grids); (void) hipFree(ctx->d_clusters);
	(void) hipFree(ctx->d_sky);  (void) hipFree(ctx->d_frame);
	(void) hipFree(ctx->prog.d_accum); for (float *low : ctx->prog.d_low) (void) hipFree(low); (void) hipFree(ctx->prog.d_out); (void) hipFree(ctx->prog.d_count);
	if (ctx->prog.sums_touched) (void) hipEventDestroy(ctx->prog.sums_touched);
	(void) hipStreamDestroy(ctx->stream);
	delete ctx;
}

static int mark_audited_cells(rt_context *ctx);

int rt_set_scene(rt_context *ctx, const Scene *scene)
{
	if (!ctx || !scene) return fail(RT_ERR_ARGUMENT, "rt_set_scene: NULL argument");
	const int n = scene->num_objects;
	if (n < 0 || n > MAX_OBJECTS) return fail(RT_ERR_ARGUMENT, "rt_set_scene: num_objects %d not in [0,%d]", n, MAX_OBJECTS);
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }   /* frames still in flight read the old scene / compiled kernel */

	/* every ray-independent term folded with the reference's roundings: rt_pack.cpp (host only; also behind the build-time
	 * tool that generates the shipped scenes' headers) */
	std::vector<rt_geom>  geom;
	std::vector<rt_shade> shade;
	rt_packed_scene_info info;
	rt_pack_scene(scene, geom, shade, &info);
	const int light = info.light_index;
	const bool fast_ok = info.fast_ok;
	ctx->light_index = light;
	ctx->scene_fast_ok = fast_ok;
	ctx->only_light_emits = info.only_light_emits;
	memcpy(ctx->light_pos, info.light_pos, sizeof(ctx->light_pos));
	/* scenes of more than 64 objects: clusters for the culled trace (also sets the half extents in the spheres' records) */
	std::vector<rt_cluster> clusters;
	std::vector<rt_group> groups;
	ctx->cull = fast_ok ? rt_cull_build(geom, n, clusters, groups) : rt_cull_info{ 0, 0.0f, 0.0f };
	if (ctx->cull.num_clusters > 0) {
		if (ctx->cull.num_clusters > ctx->clusters_capacity) {
			(void) hipFree(ctx->d_clusters); ctx->d_clusters = nullptr; ctx->clusters_capacity = 0;
			HIP_TRY(hipMalloc((void**) &ctx->d_clusters, (size_t) RT_CULL_F4(RT_MAX_CLUSTERS) * 16));      /* the cluster records, the group records behind them */
			ctx->clusters_capacity = RT_MAX_CLUSTERS;
		}
		static_assert(sizeof(rt_cluster) == 16 * RT_CLUSTER_F4 && sizeof(rt_group) == 16 * RT_GROUP_F4, "records as float4 words");
		HIP_TRY(hipMemcpy(ctx->d_clusters, clusters.data(), clusters.size() * sizeof(rt_cluster), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(ctx->d_clusters + clusters.size(), groups.data(), groups.size() * sizeof(rt_group), hipMemcpyHostToDevice));
	}
	ctx->h_geom.assign(geom.begin(), geom.begin() + n);
	ctx->spec_module = nullptr; ctx->spec_fn = nullptr;      /* (not unloaded: rt_jit.cpp keeps compiled scenes for the life of the process) */
	ctx->spec_fn_audit = nullptr; ctx->spec_audit_failed = false; ctx->spec_audit_not_embedded = false;
	if (n > ctx->capacity || !ctx->d_geom) {
		(void) hipFree(ctx->d_geom); (void) hipFree(ctx->d_shade);
		ctx->d_geom = nullptr; ctx->d_shade = nullptr; ctx->capacity = 0;
		const int cap = n > 16 ? n : 16;
		HIP_TRY(hipMalloc((void**) &ctx->d_geom, (size_t) cap * sizeof(rt_geom)));
		HIP_TRY(hipMalloc((void**) &ctx->d_shade, (size_t) cap * sizeof(rt_shade)));
		ctx->capacity = cap;
	}
	if (n > 0) {
		HIP_TRY(hipMemcpy(ctx->d_geom, geom.data(), (size_t) n * sizeof(rt_geom), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(ctx->d_shade, shade.data(), (size_t) n * sizeof(rt_shade), hipMemcpyHostToDevice));
	}
	/* the table of hit points whose soft-shadow taps certainly reach the emitter (rt_lit.h): cells of 1/32 scene unit,
	 * coarser if that takes more than 2^21 cells (a byte each); scenes of up to 64 objects with a sphere as the first emitter */
	ctx->have_lit = false;
	if (light >= 0 && n <= 64 && fast_ok) {
		static_assert(sizeof(rt_geom) == 8 * sizeof(float) && sizeof(rt_lit_grid) == 48, "rt_lit.h reads rt_geom as 8 floats");
		std::vector<rt_lit_grid> grids((size_t) n);
		const float *words8 = reinterpret_cast<const float*>(geom.data());
		long long bits = 0;
		for (float cell = 0.03125f; cell <= 64.0f; cell *= 2.0f) {
			bits = rt_lit_layout(words8, n, light, cell, grids.data());
			if (bits <= (1ll << 21)) break;
			bits = 0;
		}
		if (bits > 0) {
			std::vector<uint32_t> words((size_t) ((bits + 31) / 32)), dark(ctx->only_light_emits ? (size_t) ((bits + 31) / 32) : 0);
			rt_lit_build(words8, n, light, ctx->light_pos[0], ctx->light_pos[1], ctx->light_pos[2], grids.data(), words.data(),
			             dark.empty() ? nullptr : dark.data(), bits);
			std::vector<unsigned char> cells((size_t) bits);          /* a byte per cell on the device: one load, no shift; 1 lit, 2 dark */
			for (long long b = 0; b < bits; b++)
				cells[(size_t) b] = (unsigned char) (ctx->tuning.test_corrupt_lit_table ? 1u :       /* (testing aid: a table that is wrong) */
				                                     ((words[(size_t) (b >> 5)] >> (b & 31)) & 1u) ? 1u :
				                                     (!dark.empty() && ((dark[(size_t) (b >> 5)] >> (b & 31)) & 1u)) ? 2u : 0u);
			if (cells.size() > ctx->lit_cells_capacity) {
				(void) hipFree(ctx->d_lit_cells); ctx->d_lit_cells = nullptr; ctx->lit_cells_capacity = 0;
				HIP_TRY(hipMalloc((void**) &ctx->d_lit_cells, 2 * cells.size()));      /* (+ the marked copy of audited launches: mark_audited_cells) */
				ctx->lit_cells_capacity = cells.size();
			}
			if (n > ctx->lit_grids_capacity) {
				(void) hipFree(ctx->d_lit_grids); ctx->d_lit_grids = nullptr; ctx->lit_grids_capacity = 0;
				HIP_TRY(hipMalloc(&ctx->d_lit_grids, (size_t) n * sizeof(rt_lit_grid)));
				ctx->lit_grids_capacity = n;
			}
			ctx->h_lit_cells = cells;
			ctx->lit_audit_marked = 0;                        /* (mark_audited_cells() before the next launch, if the audit is on) */
			HIP_TRY(hipMemcpy(ctx->d_lit_cells, cells.data(), cells.size(), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(ctx->d_lit_grids, grids.data(), (size_t) n * sizeof(rt_lit_grid), hipMemcpyHostToDevice));
			ctx->have_lit = true;
			{ const int mrc = mark_audited_cells(ctx); if (mrc != RT_OK) return mrc; }      /* (now, not in front of the first launch that is audited) */
		}
	}
	ctx->num_objects = n;
	ctx->have_scene = true;
	ctx->input_version++;
	return RT_OK;
}

/* Scene "compilation": see rt_jit.cpp.  Optional; everything works without it. */
static hipFunction_t trace_kernel_for(rt_context *ctx, bool audit);

int rt_compile_scene(rt_context *ctx)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_compile_scene: NULL context");
	if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_compile_scene: no scene set (rt_set_scene)");
	if (ctx->spec_fn) return RT_OK;
	const int n = ctx->num_objects;
	if (n < 1 || n > 64) return fail(RT_ERR_ARGUMENT, "rt_compile_scene: %d objects; only scenes of 1..64 objects are specialised", n);
	if (!ctx->scene_fast_ok) return fail(RT_ERR_ARGUMENT, "rt_compile_scene: scene has a box with negative size or out-of-range coordinates");
	HIP_TRY(hipSetDevice(ctx->device));
	std::string message;
	const int rc = rt_jit_build(ctx->h_geom.data(), n, ctx->light_index, ctx->light_pos, ctx->only_light_emits ? 1 : 0, ctx->tuning.jit_waves_per_simd, ctx->jit_flags.c_str(),
	                            &ctx->spec_module, &ctx->spec_fn, message, &ctx->spec_code, &ctx->spec_compiler);
	if (rc != RT_OK) { ctx->spec_module = nullptr; ctx->spec_fn = nullptr; return fail(rc, "rt_compile_scene: %s", message.c_str()); }
	/* the audit variant, if it comes with the library (the shipped scenes' do): a module load now instead of in front of the first
	 * launch the background audit picks */
	if (ctx->tuning.audit_known_taps == 0) (void) trace_kernel_for(ctx, true);
	return RT_OK;
}

int rt_scene_is_compiled(rt_context *ctx) { return ctx && ctx->spec_fn ? 1 : 0; }

const char *rt_compiled_scene_info(rt_context *ctx) { return ctx && ctx->spec_fn ? ctx->spec_compiler.c_str() : ""; }

/* Development aid (scripts/stats_c1.py): the instrumentation counters of a scene-specialised kernel that was
 * compiled with rt_tuning.jit_flags = "-DRT_STATS" (the kernel then carries its own `rt_stats` array). */
int rt_spec_stats_read(rt_context *ctx, unsigned long long out[64], int reset)
{
	if (!ctx || !out) return fail(RT_ERR_ARGUMENT, "rt_spec_stats_read: NULL argument");
	if (!ctx->spec_module) return fail(RT_ERR_STATE, "rt_spec_stats_read: no compiled scene");
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }
	hipDeviceptr_t p = nullptr; size_t bytes = 0;
	HIP_TRY(hipModuleGetGlobal(&p, &bytes, ctx->spec_module, "rt_stats"));
	if (bytes < 64 * sizeof(unsigned long long)) return fail(RT_ERR_STATE, "rt_spec_stats_read: unexpected symbol size");
	HIP_TRY(hipMemcpy(out, (void *) p, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	if (reset) HIP_TRY(hipMemset((void *) p, 0, 64 * sizeof(unsigned long long)));
	return RT_OK;
}

/* Development aid: copy a named device variable of the compiled scene's module (instrumented builds carry more than
 * rt_stats, e.g. rt_wave_log) to the host; at most `bytes` bytes, returns the number copied through *copied. */
int rt_spec_symbol_read(rt_context *ctx, const char *name, void *dst, size_t bytes, size_t *copied)
{
	if (!ctx || !name || !dst) return fail(RT_ERR_ARGUMENT, "rt_spec_symbol_read: NULL argument");
	if (!ctx->spec_module) return fail(RT_ERR_STATE, "rt_spec_symbol_read: no compiled scene");
	if (name[0] == '\0') {                 /* the empty name: the compiled kernel's code object itself (scripts/probes/runtime_probe.py) */
		const size_t n = ctx->spec_code.size() < bytes ? ctx->spec_code.size() : bytes;
		memcpy(dst, ctx->spec_code.data(), n);
		if (copied) *copied = ctx->spec_code.size();
		return RT_OK;
	}
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }
	hipDeviceptr_t p = nullptr; size_t have = 0;
	HIP_TRY(hipModuleGetGlobal(&p, &have, ctx->spec_module, name));
	const size_t n = have < bytes ? have : bytes;
	HIP_TRY(hipMemcpy(dst, (void *) p, n, hipMemcpyDeviceToHost));
	if (copied) *copied = n;
	return RT_OK;
}

int rt_set_skybox(rt_context *ctx, const Cubemap *sky)
{
	if (!ctx || !sky) return fail(RT_ERR_ARGUMENT, "rt_set_skybox: NULL argument");
	if (sky->w <= 0 || sky->h <= 0) return fail(RT_ERR_ARGUMENT, "rt_set_skybox: bad size %dx%d", sky->w, sky->h);
	if ((long long) sky->w * sky->h * 6 > 0x7fffffffLL)     /* the kernels index texels with the reference's int arithmetic */
		return fail(RT_ERR_ARGUMENT, "rt_set_skybox: %dx%d faces are too large", sky->w, sky->h);
	if (sky->chan != 3 && sky->chan != 4)
		return fail(RT_ERR_ARGUMENT, "rt_set_skybox: %d channels unsupported (need 3 or 4)", sky->chan);
	for (int f = 0; f < 6; f++)
		if (!sky->data[f]) return fail(RT_ERR_ARGUMENT, "rt_set_skybox: face %d is NULL", f);
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }   /* frames still in flight read the old texels */

	/* RGBA8 repack: one aligned dword per texel for the kernel's gather (sample_cubemap reads
	 * bytes [0..2] of a `chan`-strided texel, gpu_and_windowing.c:106-111) */
	const size_t texels = (size_t) sky->w * sky->h;
	const size_t bytes = texels * 6 * sizeof(uint32_t);
	uint32_t *staging = nullptr;
	HIP_TRY(hipHostMalloc((void**) &staging, bytes, hipHostMallocDefault));
	for (int f = 0; f < 6; f++) {
		const uint8_t *src = sky->data[f];
		uint32_t *dst = staging + (size_t) f * texels;
		const int ch = sky->chan;
		for (size_t t = 0; t < texels; t++, src += ch)
			dst[t] = (uint32_t) src[0] | ((uint32_t) src[1] << 8) | ((uint32_t) src[2] << 16) | 0xff000000u;
	}
	if (bytes != ctx->sky_bytes) {
		(void) hipFree(ctx->d_sky); ctx->d_sky = nullptr; ctx->sky_bytes = 0;
		hipError_t e = hipMalloc((void**) &ctx->d_sky, bytes);
		if (e != hipSuccess) { (void) hipHostFree(staging); return fail(RT_ERR_DEVICE, "hipMalloc(skybox): %s", hipGetErrorString(e)); }
		ctx->sky_bytes = bytes;
	}
	hipError_t e = hipMemcpy(ctx->d_sky, staging, bytes, hipMemcpyHostToDevice);
	(void) hipHostFree(staging);
	if (e != hipSuccess) return fail(RT_ERR_DEVICE, "hipMemcpy(skybox): %s", hipGetErrorString(e));
	ctx->sky_w = sky->w; ctx->sky_h = sky->h;
	ctx->have_sky = true;
	ctx->input_version++;
	return RT_OK;
}

int rt_set_camera(rt_context *ctx, const rt_camera *camera)
{
	if (!ctx || !camera) return fail(RT_ERR_ARGUMENT, "rt_set_camera: NULL argument");
	ctx->camera = *camera;
	ctx->have_camera = true;
	return RT_OK;
}

int rt_strip_rows(int height, int row_block, int world)
{
	if (height <= 0 || row_block <= 0 || world <= 0) return 0;
	const int blocks = (height + row_block - 1) / row_block;
	const int per_rank = (blocks + world - 1) / world;
	return per_rank * row_block;
}

static int check_params(const rt_context *ctx, const rt_render_params *p)
{
	if (!ctx || !p) return fail(RT_ERR_ARGUMENT, "render: NULL argument");
	if (!ctx->have_scene)  return fail(RT_ERR_STATE, "render: no scene set (rt_set_scene)");
	if (!ctx->have_sky)    return fail(RT_ERR_STATE, "render: no skybox set (rt_set_skybox)");
	if (p->width < 2 || p->height < 2 || (int64_t) p->width * p->height > (int64_t) 1 << 30)
		return fail(RT_ERR_ARGUMENT, "render: frame %dx%d unsupported (need >= 2x2, <= 2^30 pixels)", p->width, p->height);
	if (p->spp < 1)         return fail(RT_ERR_ARGUMENT, "render: spp %d < 1", p->spp);
	if (p->max_bounces < 0) return fail(RT_ERR_ARGUMENT, "render: max_bounces %d < 0", p->max_bounces);
	if (p->row_block < 1 || p->world < 1 || p->rank < 0 || p->rank >= p->world)
		return fail(RT_ERR_ARGUMENT, "render: bad partition row_block=%d rank=%d world=%d", p->row_block, p->rank, p->world);
	if (p->kernel < RT_KERNEL_AUTO || p->kernel > RT_KERNEL_WAVEFRONT)
		return fail(RT_ERR_ARGUMENT, "render: unknown kernel %d", p->kernel);
	return RT_OK;
}

static hipEvent_t take_event(rt_context *ctx)
{
	if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}

static void give_event(rt_context *ctx, hipEvent_t e) { if (e) ctx->event_pool.push_back(e); }

/* Scheduling parameters of the wavefront kernels for one launch (rt_device.h) and the pixel lists rt_primary_pass
 * fills for the trace kernel (grown on demand).  Any schedule renders the same frame. */
static int prepare_scratch(rt_context *ctx, rt_launch &L, unsigned which)
{
	rt_context::launch_slot &sl = ctx->slot[which % RT_LAUNCH_SETS];
	const long long pixel_blocks = (long long) ((L.width + 7) / 8) * ((L.local_rows + 7) / 8);
	/* 64 lists (a single dequeue counter takes ~88 atomics per microsecond: 4 K waves asking for their first pixels at
	 * once would already queue up) unless the launch is small */
	L.num_shards = (pixel_blocks >= 64 * 16 && ctx->num_cus >= 64) ? 64 : 1;
	if (ctx->tuning.dequeue_shards) L.num_shards = ctx->tuning.dequeue_shards;

	const size_t cap = rt_pixel_list_capacity(L.width, L.local_rows, ctx->num_cus, L.num_shards);
	const size_t records = cap * (size_t) L.num_shards;
	if (cap > (size_t) 0x7fffffff) return fail(RT_ERR_ARGUMENT, "render: frame too large");
	if (records > sl.pix_capacity) {
		(void) hipFree(sl.d_pix); sl.d_pix = nullptr; sl.pix_capacity = 0;    /* (hipFree waits for the device: nothing still reads it) */
		sl.lists_key = 0;
		HIP_TRY(hipMalloc((void**) &sl.d_pix, records * 12 * sizeof(float)));
		sl.pix_capacity = records;
	}
	L.pix = sl.d_pix;
	L.pix_shard_cap = (int) cap;
	L.pix_count = sl.d_counter + 64 * 32;      /* counter block: 64 dequeue counters, 64 fill counters, one control line */
	L.control = sl.d_counter + 128 * 32;
	L.stop = ctx->d_stop;
	return RT_OK;
}

/* rt_lit.h audited in production (include/rt_hip.h rt_tuning.audit_known_taps).  k != 0: what the host asked for -- every launch,
 * one answer in 2^k (-1: every one) -- or RT_AUDIT_OFF.  k == 0, the default, is the BACKGROUND audit (round 6; until then the
 * default was "never", and a production host ran the taps of 80 % of C1's bounces on a table nobody looked at): every
 * RT_AUDIT_PERIOD-th launch of a context is rendered by the audit variant with one answer in 2^RT_AUDIT_BACKGROUND_LOG2 re-traced
 * and compared.  An audited launch costs 2 % more, i.e. 0.03 % of a frame loop; a wrong table is caught within a second of
 * frames (RT_ERR_DEVICE, taps_disagreeing).  The background audit never compiles anything: a compiled scene's audit variant is
 * used when it comes embedded with the library (loaded by rt_compile_scene), otherwise the generic kernel's. */
#define RT_AUDIT_PERIOD 61u
#define RT_AUDIT_BACKGROUND_LOG2 3
static int audit_log2_of_launch(const rt_context *ctx, unsigned int launch)
{
	const int k = ctx->tuning.audit_known_taps;
	if (k == RT_AUDIT_OFF) return 0;
	if (k != 0) return k;
	return launch % RT_AUDIT_PERIOD == RT_AUDIT_PERIOD - 1u ? RT_AUDIT_BACKGROUND_LOG2 : 0;
}
/* ... and the marks the scene's table carries for it (they mean nothing to a launch that is not audited) */
static int audit_log2_of_table(const rt_context *ctx)
{
	const int k = ctx->tuning.audit_known_taps;
	return k == RT_AUDIT_OFF ? 0 : (k != 0 ? k : RT_AUDIT_BACKGROUND_LOG2);
}

/* ... and the launch gets its number.  rt_cancel() on another thread must cover this launch from the moment its first kernel
 * can be on the GPU: the number is published BEFORE anything of the launch is enqueued (a request that arrives in between
 * stops a launch that has not started yet, at its first pixel fetch).  A launch that then fails to enqueue keeps its number:
 * numbers are never handed out twice, so a request that named it can only stop launches that were announced when it was made
 * (such a number is simply never seen on the device; unpublish_launch() below is what the failure paths call, and does
 * nothing more than say so). */
static int prepare_launch(rt_context *ctx, rt_launch &L, unsigned which)
{
	const int rc = prepare_scratch(ctx, L, which);
	if (rc != RT_OK) return rc;
	L.launch_id = ++ctx->last_id;
	if (L.launch_id == 0u) L.launch_id = ++ctx->last_id;      /* (0 is "no stamp") */
	const int k = audit_log2_of_launch(ctx, which);
	L.audit_taps = k == 0 ? 0u : (k < 0 ? 1u : 1u << k);
	L.test_drop_pixels = (unsigned int) ctx->tuning.test_drop_pixels;
	ctx->enqueued.store(L.launch_id, std::memory_order_release);
	return RT_OK;
}

static void unpublish_launch(rt_context *) { }

/* The launch scratch (pixel lists) and rt_render()'s device frame for frames up to width x height, allocated now
 * instead of inside the first render call of that size. */
int rt_reserve(rt_context *ctx, int width, int height)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_reserve: NULL context");
	if (width < 2 || height < 2 || (int64_t) width * height > (int64_t) 1 << 30)
		return fail(RT_ERR_ARGUMENT, "rt_reserve: frame %dx%d unsupported (need >= 2x2, <= 2^30 pixels)", width, height);
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }
	rt_launch L;
	memset(&L, 0, sizeof(L));
	L.width = width; L.local_rows = rt_strip_rows(height, 8, 1);
	for (unsigned which = 0; which < RT_LAUNCH_SETS; which++) { const int rc = prepare_scratch(ctx, L, which); if (rc != RT_OK) return rc; }     /* (nothing is announced: nothing is launched) */
	const size_t need = (size_t) L.local_rows * width * 3 * sizeof(float);
	if (need > ctx->frame_bytes) {
		(void) hipFree(ctx->d_frame); ctx->d_frame = nullptr; ctx->frame_bytes = 0;
		HIP_TRY(hipMalloc((void**) &ctx->d_frame, need));
		ctx->frame_bytes = need;
	}
	return RT_OK;
}

/* The trace kernel of an audited launch (rt_tuning.audit_known_taps != 0) is a VARIANT that carries the comparison; a compiled
 * scene gets its variant from a second build (-DRT_SPEC_AUDIT: hiprtc at run time, about half a second, once per scene and
 * process).  Where that is not possible the generic kernel's audit variant renders the audited launches: same frames. */
static hipFunction_t trace_kernel_for(rt_context *ctx, bool audit)
{
	if (!audit || !ctx->spec_fn) return ctx->spec_fn;
	/* (the background audit compiles nothing: what rt_compile_scene found embedded, or the generic kernel) */
	const bool may_compile = ctx->tuning.audit_known_taps != 0;
	if (!ctx->spec_fn_audit && !ctx->spec_audit_failed && (may_compile || !ctx->spec_audit_not_embedded)) {
		hipModule_t module = nullptr;
		std::string message, flags = ctx->jit_flags + (ctx->jit_flags.empty() ? "" : " ") + "-DRT_SPEC_AUDIT";
		const int rc = rt_jit_build(ctx->h_geom.data(), ctx->num_objects, ctx->light_index, ctx->light_pos, ctx->only_light_emits ? 1 : 0, ctx->tuning.jit_waves_per_simd,
		                            flags.c_str(), &module, &ctx->spec_fn_audit, message, nullptr, nullptr, /* embedded_only = */ !may_compile);
		if (rc != RT_OK) { ctx->spec_fn_audit = nullptr; if (may_compile) ctx->spec_audit_failed = true; else ctx->spec_audit_not_embedded = true; }
	}
	return ctx->spec_fn_audit;        /* (nullptr: the generic kernel) */
}

/* rt_tuning.audit_known_taps: the scene's lit-taps table carries the audit mark itself -- bit 2 of one answered cell in 2^k,
 * picked by a hash of the cell's index -- so that the trace kernel needs no launch constant for it (a scalar register through
 * every round of a kernel at its register limit).  The device's copy is re-marked, between launches, when the setting changed. */
static int mark_audited_cells(rt_context *ctx)
{
	const int k = audit_log2_of_table(ctx);
	if (!ctx->have_lit || ctx->lit_audit_marked == k || k == 0) return RT_OK;
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }       /* launches in flight read the table */
	std::vector<unsigned char> cells = ctx->h_lit_cells;
	if (k != 0) {
		const uint32_t mask = k < 0 ? 0u : (1u << k) - 1u;
		for (size_t b = 0; b < cells.size(); b++)
			if (cells[b] && ((uint32_t) ((b * 0x9E3779B97F4A7C15ull) >> (64 - RT_AUDIT_MAX_LOG2)) & mask) == 0u) cells[b] |= 4u;
	}
	/* (a copy of its own: launches that are not audited -- every one but the background audit's -- read the table without marks,
	 * and their kernels need not mask them out) */
	HIP_TRY(hipMemcpy(ctx->d_lit_cells + ctx->lit_cells_capacity, cells.data(), cells.size(), hipMemcpyHostToDevice));
	ctx->lit_audit_marked = k;
	return RT_OK;
}

int rt_render_device(rt_context *ctx, const rt_render_params *p, void *d_strip, void *hip_stream)
{
	int rc = check_params(ctx, p);
	if (rc != RT_OK) return rc;
	if (!d_strip) return fail(RT_ERR_ARGUMENT, "rt_render_device: d_strip is NULL");
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t stream = pick_stream(ctx, hip_stream);
	{ const int mrc = mark_audited_cells(ctx); if (mrc != RT_OK) return mrc; }

	rt_launch L;
	memset(&L, 0, sizeof(L));
	rt_camera_basis basis;
	rt_camera_basis_for(&ctx->camera, (float) p->width / p->height, &basis);   /* main.c:281 */
	const Vector3 *src[4] = { &basis.pos, &basis.lower_left_corner, &basis.horizontal, &basis.vertical };
	float *dst[4] = { L.pos, L.llc, L.horiz, L.vert };
	for (int k = 0; k < 4; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
	L.light_index = ctx->light_index;
	memcpy(L.light_pos, ctx->light_pos, sizeof(L.light_pos));
	L.only_light_emits = ctx->only_light_emits ? 1 : 0;
	L.num_objects = ctx->num_objects;
	L.width = p->width; L.height = p->height;
	L.spp = p->spp; L.max_bounces = p->max_bounces; L.seed = p->seed;
	L.u_den = p->width - 1; L.v_den = p->height - 1; L.pix_scale = 1; L.pix_width = p->width; L.sample_base = 0;
	L.row_block = p->row_block; L.rank = p->rank; L.world = p->world;
	/* rows this rank owns: blocks rank, rank+world, ... ; the last one may be partial or absent */
	{
		const int blocks = (p->height + p->row_block - 1) / p->row_block;
		const int mine = blocks > p->rank ? (blocks - p->rank + p->world - 1) / p->world : 0;
		L.local_rows = mine * p->row_block;
	}
	L.sky = ctx->d_sky; L.sky_w = ctx->sky_w; L.sky_h = ctx->sky_h; L.sky_wm1 = (float) (ctx->sky_w - 1); L.sky_hm1 = (float) (ctx->sky_h - 1);
	L.frame = (float*) d_strip;
	L.skip_known_taps = classify_pixels(ctx, p->spp);
	L.lit_cells = (ctx->have_lit && !ctx->tuning.trace_known_taps) ? ctx->d_lit_cells + (audit_log2_of_launch(ctx, ctx->launches) != 0 ? ctx->lit_cells_capacity : 0) : nullptr;
	L.lit_grids = ctx->d_lit_grids;
	L.geom = ctx->d_geom; L.shade = ctx->d_shade;
	if (ctx->cull.num_clusters > 0 && !ctx->tuning.test_every_object) {
		L.clusters = ctx->d_clusters; L.num_clusters = ctx->cull.num_clusters; L.cull_margin = ctx->cull.margin; L.cull_origin_max = ctx->cull.origin_max;
	}
	{ const int rc = prepare_launch(ctx, L, ctx->launches); if (rc != RT_OK) { unpublish_launch(ctx); return rc; } }
	{ const int rc = order_behind_previous(ctx, stream); if (rc != RT_OK) { unpublish_launch(ctx); return rc; } }
	if (ctx->tuning.poison_frame) {
		const hipError_t pe = hipMemsetAsync(d_strip, 0xff, (size_t) rt_strip_rows(p->height, p->row_block, p->world) * p->width * 3 * sizeof(float), stream);
		if (pe != hipSuccess) { unpublish_launch(ctx); return fail(RT_ERR_DEVICE, "hipMemsetAsync(poison): %s", hipGetErrorString(pe)); }
	}
	hipEvent_t e0 = nullptr, em = nullptr, e1 = nullptr;
	if (ctx->profiling) {
		e0 = take_event(ctx); em = ctx->profiling_split ? take_event(ctx) : nullptr; e1 = take_event(ctx);
		if (!e0 || (!em && ctx->profiling_split) || !e1) { give_event(ctx, e0); give_event(ctx, em); give_event(ctx, e1); unpublish_launch(ctx); return fail(RT_ERR_DEVICE, "rt_render_device: hipEventCreate failed"); }
	}
	ctx->slot[ctx->launches % RT_LAUNCH_SETS].lists_key = 0;
	ctx->primary_passes++;
	const bool audit = L.audit_taps != 0u;           /* (prepare_launch: asked for, or this launch's turn of the background audit) */
	hipError_t le = rt_launch_trace(L, p->kernel, ctx->scene_fast_ok, trace_kernel_for(ctx, audit), ctx->slot[ctx->launches % RT_LAUNCH_SETS].d_counter, e0, ctx->slot[ctx->launches % RT_LAUNCH_SETS].started, ctx->num_cus, workgroups_per_cu_for(ctx, stream, (long long) L.width * L.local_rows), stream,
	                                false, &ctx->slot[ctx->launches % RT_LAUNCH_SETS].expect, em, audit);
	if (le == hipSuccess && ctx->profiling) le = hipEventRecord(e1, stream);
	if (ctx->profiling) {
		if (le == hipSuccess) ctx->events.push_back({ e0, em, e1 });
		else { give_event(ctx, e0); give_event(ctx, em); give_event(ctx, e1); }                 /* a failed launch keeps no events */
	}
	if (le != hipSuccess) { unpublish_launch(ctx); return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(le)); }
	return mark_launch(ctx, stream);
}

int rt_render(rt_context *ctx, const rt_render_params *p, Vector3 *frame_out)
{
	int rc = check_params(ctx, p);
	if (rc != RT_OK) return rc;
	if (!frame_out) return fail(RT_ERR_ARGUMENT, "rt_render: frame_out is NULL");
	if (p->world != 1) return fail(RT_ERR_ARGUMENT, "rt_render: world must be 1 (use rt_render_device for strips)");
	HIP_TRY(hipSetDevice(ctx->device));
	const int rows = rt_strip_rows(p->height, p->row_block, 1);
	const size_t need = (size_t) rows * p->width * 3 * sizeof(float);
	if (need > ctx->frame_bytes) {
		(void) hipFree(ctx->d_frame); ctx->d_frame = nullptr; ctx->frame_bytes = 0;
		HIP_TRY(hipMalloc((void**) &ctx->d_frame, need));
		ctx->frame_bytes = need;
	}
	rc = rt_render_device(ctx, p, ctx->d_frame, ctx->stream);
	if (rc != RT_OK) return rc;
	HIP_TRY(hipMemcpyAsync(frame_out, ctx->d_frame, (size_t) p->height * p->width * 3 * sizeof(float),
	                       hipMemcpyDeviceToHost, ctx->stream));
	/* the launch's control words follow the frame in the same stream: one synchronisation, then the verdict (rt_judge_launch: a
	 * launch that did not account for every pixel gives an error, not a frame with a hole in it -- round 3 saw a blocking render
	 * lose the pixels a launch deals last, twice, cause unknown: docs/lab/r04.md) */
	HIP_TRY(hipMemcpyAsync(&ctx->h_words[0], ctx->slot[ctx->cur].d_counter + 128 * 32, RT_CTL_COPY_BYTES, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return rt_judge_launch(&ctx->h_words[0], ctx->slot[ctx->cur].expect, "rt_render", nullptr);
}

/* ---- frames in flight (include/rt_hip.h): the reference's workers keep accumulating while its main thread presents
 * (main.c:354-408 vs 450-482).  Frame n is rendered on the context's stream n & 1 -- consecutive launches sit in two
 * hardware queues and overlap on the GPU, see order_behind_previous() -- into its slot's device frame; the copy to the
 * caller's memory runs on a third, high-priority stream (a queue of its own: streams of equal priority share a handful
 * of queues in creation order, and a copy queued behind the next render would not overlap it). */
static int frame_copy_stream(rt_context *ctx)
{
	if (ctx->copy_stream) return RT_OK;
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
	if (hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess)
		HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
	return RT_OK;
}

/* frame_out != NULL: the frame is copied to the caller's host memory; NULL (rt_frame_submit_device): it stays in the slot's
 * device buffer and the caller gets that buffer and an event recorded behind the render */
static int frame_submit(rt_context *ctx, const rt_render_params *p, int slot, Vector3 *frame_out, void **d_frame, void **hip_event)
{
	int rc = check_params(ctx, p);
	if (rc != RT_OK) return rc;
	if (slot < 0 || slot >= RT_FRAME_SLOTS) return fail(RT_ERR_ARGUMENT, "rt_frame_submit: slot %d not in [0,%d)", slot, RT_FRAME_SLOTS);
	if (p->world != 1) return fail(RT_ERR_ARGUMENT, "rt_frame_submit: world must be 1 (rt_multi_frame_submit renders on several GPUs)");
	rt_context::frame_slot &f = ctx->fq[slot];
	if (f.busy) return fail(RT_ERR_STATE, "rt_frame_submit: slot %d holds a frame that has not been waited for", slot);
	HIP_TRY(hipSetDevice(ctx->device));
	rc = frame_copy_stream(ctx);
	if (rc != RT_OK) return rc;
	if (!f.copied) HIP_TRY(hipEventCreateWithFlags(&f.copied, hipEventDisableTiming));
	if (!f.rendered) HIP_TRY(hipEventCreateWithFlags(&f.rendered, hipEventDisableTiming));
	const size_t need = (size_t) rt_strip_rows(p->height, p->row_block, 1) * p->width * 3 * sizeof(float);
	if (need > f.bytes) {
		(void) hipFree(f.d_buf); f.d_buf = nullptr; f.bytes = 0;       /* (the slot is idle: its last copy was waited for) */
		HIP_TRY(hipMalloc((void**) &f.d_buf, need));
		f.bytes = need;
	}
	hipStream_t stream = (hipStream_t) rt_stream(ctx, (int) (ctx->frames_submitted % RT_LAUNCH_SETS));
	if (!stream) return RT_ERR_DEVICE;                                  /* rt_stream() left the text */
	rc = rt_render_device(ctx, p, f.d_buf, stream);
	if (rc != RT_OK) return rc;
	ctx->frames_submitted++;
	HIP_TRY(hipEventRecord(f.rendered, stream));
	/* the copy waits for the render (an event recorded behind it on `stream`); the launch's control word (rt_cancel) follows
	 * the frame on the copy stream, and nothing waits for either but the slot -- and the scratch set's launch after next,
	 * which clears that word */
	HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, f.rendered, 0));
	if (frame_out)
		HIP_TRY(hipMemcpyAsync(frame_out, f.d_buf, (size_t) p->height * p->width * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
	rc = rt_context_read_control(ctx, &ctx->h_words[RT_WORDS_FRAME(slot)], ctx->copy_stream, nullptr, &f.expect);
	if (rc != RT_OK) return rc;
	HIP_TRY(hipEventRecord(f.copied, ctx->copy_stream));
	f.busy = true;
	if (d_frame) *d_frame = f.d_buf;
	if (hip_event) *hip_event = (void *) f.rendered;
	return RT_OK;
}

int rt_frame_submit(rt_context *ctx, const rt_render_params *p, int slot, Vector3 *frame_out)
{
	if (!frame_out) return fail(RT_ERR_ARGUMENT, "rt_frame_submit: frame_out is NULL");
	return frame_submit(ctx, p, slot, frame_out, nullptr, nullptr);
}

int rt_frame_submit_device(rt_context *ctx, const rt_render_params *p, int slot, void **d_frame, void **hip_event)
{
	if (!d_frame) return fail(RT_ERR_ARGUMENT, "rt_frame_submit_device: d_frame is NULL");
	return frame_submit(ctx, p, slot, nullptr, d_frame, hip_event);
}

int rt_frame_wait(rt_context *ctx, int slot)
{
	if (!ctx || slot < 0 || slot >= RT_FRAME_SLOTS) return fail(RT_ERR_ARGUMENT, "rt_frame_wait: bad argument");
	rt_context::frame_slot &f = ctx->fq[slot];
	if (!f.busy) return fail(RT_ERR_STATE, "rt_frame_wait: nothing was submitted into slot %d", slot);
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipEventSynchronize(f.copied));
	f.busy = false;
	return rt_judge_launch(&ctx->h_words[RT_WORDS_FRAME(slot)], f.expect, "rt_frame_wait", nullptr);
}

/* ---- launches a host enqueued itself (rt_render_device), judged like the library's own (include/rt_hip.h) ---- */
int rt_launch_check_submit(rt_context *ctx, int ticket, void *hip_stream)
{
	if (!ctx || ticket < 0 || ticket >= RT_CHECK_TICKETS) return fail(RT_ERR_ARGUMENT, "rt_launch_check_submit: bad argument");
	rt_context::check_ticket &t = ctx->tickets[ticket];
	if (t.busy) return fail(RT_ERR_STATE, "rt_launch_check_submit: ticket %d holds a launch that has not been waited for", ticket);
	if (!ctx->launches) return fail(RT_ERR_STATE, "rt_launch_check_submit: nothing has been launched");
	HIP_TRY(hipSetDevice(ctx->device));
	if (!t.copied) HIP_TRY(hipEventCreateWithFlags(&t.copied, hipEventDisableTiming));
	hipStream_t stream = pick_stream(ctx, hip_stream);
	const int rc = rt_context_read_control(ctx, &ctx->h_words[RT_WORDS_TICKET(ticket)], stream, nullptr, &t.expect);
	if (rc != RT_OK) return rc;
	HIP_TRY(hipEventRecord(t.copied, stream));
	t.busy = true;
	return RT_OK;
}

int rt_launch_check_wait(rt_context *ctx, int ticket, rt_launch_report *report)
{
	if (!ctx || ticket < 0 || ticket >= RT_CHECK_TICKETS) return fail(RT_ERR_ARGUMENT, "rt_launch_check_wait: bad argument");
	rt_context::check_ticket &t = ctx->tickets[ticket];
	if (!t.busy) return fail(RT_ERR_STATE, "rt_launch_check_wait: ticket %d holds nothing", ticket);
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipEventSynchronize(t.copied));
	t.busy = false;
	return rt_judge_launch(&ctx->h_words[RT_WORDS_TICKET(ticket)], t.expect, "rt_launch_check_wait", report);
}

int rt_last_launch_report(rt_context *ctx, rt_launch_report *report)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_last_launch_report: NULL context");
	if (!ctx->launches) return fail(RT_ERR_STATE, "rt_last_launch_report: nothing has been launched");
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }
	unsigned int words[RT_CTL_WORDS];
	HIP_TRY(hipMemcpy(words, ctx->slot[ctx->cur].d_counter + 128 * 32, sizeof(words), hipMemcpyDeviceToHost));
	return rt_judge_launch(words, ctx->slot[ctx->cur].expect, "rt_last_launch_report", report);
}

int rt_frame_poll(rt_context *ctx, int slot)
{
	if (!ctx || slot < 0 || slot >= RT_FRAME_SLOTS) return fail(RT_ERR_ARGUMENT, "rt_frame_poll: bad argument");
	rt_context::frame_slot &f = ctx->fq[slot];
	if (!f.busy) return fail(RT_ERR_STATE, "rt_frame_poll: nothing was submitted into slot %d", slot);
	HIP_TRY(hipSetDevice(ctx->device));
	const hipError_t e = hipEventQuery(f.copied);
	if (e == hipErrorNotReady) return RT_PENDING;
	if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_frame_poll: %s", hipGetErrorString(e));
	return rt_frame_wait(ctx, slot);
}

int rt_host_alloc(void **out, size_t bytes)
{
	if (!out || bytes == 0) return fail(RT_ERR_ARGUMENT, "rt_host_alloc: bad argument");
	*out = nullptr;
	const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);     /* page-locked for every device of the process */
	if (e != hipSuccess) { *out = nullptr; return fail(RT_ERR_MEMORY, "rt_host_alloc(%zu): %s", bytes, hipGetErrorString(e)); }
	return RT_OK;
}

void rt_host_free(void *p) { if (p) (void) hipHostFree(p); }

/* Every launch enqueued so far -- running or still queued -- is asked to stop; launches enqueued later have higher numbers and
 * are not affected.  Uses nothing of the context that a render call on another thread changes. */
int rt_cancel(rt_context *ctx)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_cancel: NULL context");
	/* one store into host memory the trace kernels poll (rt_kernels.hip, top of a round): nothing is enqueued, nothing has to
	 * find room on a GPU whose every wave slot the persistent kernel holds.  (Rounds 1-2 sent a copy down a stream of its own:
	 * the runtime does small copies with a kernel, and beside the compiled trace kernel -- 4 x 128 registers per SIMD -- that
	 * kernel only got to run when the launch it was meant to stop had finished.) */
	/* fetch-max, not a store: of two concurrent calls the one that read the smaller number must not move the word backwards */
	const unsigned int upto = ctx->enqueued.load(std::memory_order_acquire);
	unsigned int seen = __atomic_load_n(&ctx->h_words[RT_STOP_WORD], __ATOMIC_RELAXED);
	while ((int) (upto - seen) > 0 &&
	       !__atomic_compare_exchange_n(&ctx->h_words[RT_STOP_WORD], &seen, upto, true, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) { }
	return RT_OK;
}

long long rt_primary_passes_run(rt_context *ctx) { return ctx ? ctx->primary_passes : -1; }

int rt_was_cancelled(rt_context *ctx)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_was_cancelled: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }
	unsigned int w = 0;
	HIP_TRY(hipMemcpy(&w, ctx->slot[ctx->cur].d_counter + 128 * 32 + RT_CTL_CANCELLED, sizeof(w), hipMemcpyDeviceToHost));
	return w ? RT_CANCELLED : RT_OK;
}

int rt_deinterleave_rotated_device(rt_context *ctx, const void *d_strips, void *d_frame,
                                   int width, int height, int row_block, int world, int first, void *hip_stream)
{
	if (!ctx || !d_strips || !d_frame) return fail(RT_ERR_ARGUMENT, "rt_deinterleave_device: NULL argument");
	if (width < 1 || height < 1 || row_block < 1 || world < 1 || first < 0 || first >= world)
		return fail(RT_ERR_ARGUMENT, "rt_deinterleave_device: bad geometry");
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t stream = pick_stream(ctx, hip_stream);
	HIP_TRY(rt_launch_deinterleave((const float*) d_strips, (float*) d_frame, width, height, row_block, world,
	                               rt_strip_rows(height, row_block, world), first, stream));
	return RT_OK;
}

int rt_deinterleave_device(rt_context *ctx, const void *d_strips, void *d_frame,
                           int width, int height, int row_block, int world, void *hip_stream)
{
	return rt_deinterleave_rotated_device(ctx, d_strips, d_frame, width, height, row_block, world, 0, hip_stream);
}

int rt_strip_of_rank(int rank, int world)
{
	return world > 1 ? (rank + world - 1) % world : 0;
}

/* ---- progressive accumulation: worker() scale ladder + update_frame() (main.c:354-408, 450-482) ---- */

} /* extern "C" */

/* `stream` is about to read or write the ladder's sums or their count ... */
static int sums_wait(rt_context *ctx, hipStream_t stream)
{
	auto &g = ctx->prog;
	if (g.sums_used && g.sums_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, g.sums_touched, 0));
	return RT_OK;
}
/* ... and has enqueued what it does with them */
static int sums_mark(rt_context *ctx, hipStream_t stream)
{
	auto &g = ctx->prog;
	HIP_TRY(hipEventRecord(g.sums_touched, stream));
	g.sums_stream = stream; g.sums_used = true;
	return RT_OK;
}

extern "C" {

int rt_progressive_begin(rt_context *ctx, int width, int height, int init_scale, int max_bounces, uint64_t seed)
{
	return rt_progressive_begin_rank(ctx, width, height, init_scale, max_bounces, seed, 0, 1);
}

int rt_progressive_begin_rank(rt_context *ctx, int width, int height, int init_scale, int max_bounces, uint64_t seed, int rank, int world)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_progressive_begin: NULL context");
	if (init_scale != 1 && init_scale != 2 && init_scale != 4 && init_scale != 8 && init_scale != 16)
		return fail(RT_ERR_ARGUMENT, "rt_progressive_begin: init_scale %d not in {1,2,4,8,16}", init_scale);   /* main.c:611-621 */
	if (width / init_scale < 2 || height / init_scale < 2 || (int64_t) width * height > (int64_t) 1 << 30)
		return fail(RT_ERR_ARGUMENT, "rt_progressive_begin: frame %dx%d too small for scale %d (or too large)", width, height, init_scale);
	if (max_bounces < 1) return fail(RT_ERR_ARGUMENT, "rt_progressive_begin: max_bounces %d < 1", max_bounces);
	if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_ARGUMENT, "rt_progressive_begin: bad partition rank=%d world=%d", rank, world);
	HIP_TRY(hipSetDevice(ctx->device));
	auto &g = ctx->prog;
	/* one rank: the whole frame; several: this rank's row blocks, padded like a strip (rt_strip_rows) */
	const int rows = world == 1 ? height : rt_strip_rows(height, RT_PROGRESSIVE_ROW_BLOCK, world);
	const size_t accum_bytes = (size_t) width * rows * 3 * sizeof(float);
	const size_t low_bytes = (size_t) (width + 1) * rows * 3 * sizeof(float);
	if (accum_bytes != g.accum_bytes) {
		(void) hipFree(g.d_accum); (void) hipFree(g.d_out); g.d_accum = g.d_out = nullptr; g.accum_bytes = 0;
		HIP_TRY(hipMalloc((void**) &g.d_accum, accum_bytes));
		HIP_TRY(hipMalloc((void**) &g.d_out, accum_bytes));
		g.accum_bytes = accum_bytes;
	}
	if (!g.d_count) HIP_TRY(hipMalloc((void**) &g.d_count, RT_COUNT_WORDS * sizeof(float)));
	if (low_bytes != g.low_bytes) {
		{ const int rc = wait_for_launches(ctx); if (rc != RT_OK) return rc; }     /* (passes of the frame before may still be writing theirs) */
		for (float *&low : g.d_low) { (void) hipFree(low); low = nullptr; }
		g.low_bytes = 0;
		for (float *&low : g.d_low) HIP_TRY(hipMalloc((void**) &low, low_bytes));
		g.low_bytes = low_bytes;
	}
	if (!g.sums_touched) HIP_TRY(hipEventCreateWithFlags(&g.sums_touched, hipEventDisableTiming));
	g.width = width; g.height = height; g.init_scale = init_scale; g.max_bounces = max_bounces; g.seed = seed;
	g.rank = rank; g.world = world; g.rows = rows;
	g.active = true;
	/* whatever rt_primary_pass left in the scratch sets belongs to an earlier low-resolution buffer (which may even have had
	 * this one's address: the key hashes the pointer, not the contents) */
	for (auto &sl : ctx->slot) sl.lists_key = 0;
	return rt_progressive_invalidate(ctx);
}

/* invalidate_accumulation() (main.c:115-124): counts to zero, generation + 1, buffers cleared; the next
 * pass starts again at init_scale (main.c:405-408). */
int rt_progressive_invalidate(rt_context *ctx)
{
	if (!ctx || !ctx->prog.active) return fail(RT_ERR_STATE, "rt_progressive_invalidate: call rt_progressive_begin first");
	HIP_TRY(hipSetDevice(ctx->device));
	auto &g = ctx->prog;
	/* a pass still in flight is given up as soon as its waves notice (main.c:316-317) and is not published
	 * (main.c:382: rt_accumulate looks at control[1]); the clear below is ordered behind it */
	if (ctx->launches) { const int rc = rt_cancel(ctx); if (rc != RT_OK) return rc; }
	{ const int rc = sums_wait(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	HIP_TRY(hipMemsetAsync(g.d_accum, 0, g.accum_bytes, ctx->stream));
	HIP_TRY(hipMemsetAsync(g.d_count, 0, RT_COUNT_WORDS * sizeof(float), ctx->stream));
	{ const int rc = sums_mark(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	g.passes = 0; g.scale = g.init_scale; g.generation++;
	return RT_OK;
}

#define RT_PASSES_IN_FLIGHT 3
/* One launch of the ladder: `samples` = 1 is one worker iteration (main.c:354-408) at the ladder's current scale; more than one
 * (full resolution only) is that many iterations in ONE launch -- the trace kernel adds the pixel's samples, in pass order, onto
 * the sums so far (rt_launch.sum_onto), which is what `samples` publish steps (main.c:394) would have done one after the other. */
static int progressive_launch(rt_context *ctx, int samples, float *weight_out)
{
	if (!ctx || !ctx->prog.active) return fail(RT_ERR_STATE, "rt_progressive_pass: call rt_progressive_begin first");
	if (!ctx->have_scene) return fail(RT_ERR_STATE, "render: no scene set (rt_set_scene)");
	if (!ctx->have_sky)   return fail(RT_ERR_STATE, "render: no skybox set (rt_set_skybox)");
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int mrc = mark_audited_cells(ctx); if (mrc != RT_OK) return mrc; }
	auto &g = ctx->prog;
	const int s = g.scale;
	const int lw = g.width / s, lh = g.height / s, lcw = g.width / s + 1;     /* main.c:284-286 */
	/* Passes rotate through the context's scratch sets and low-resolution frames, and through THREE of its render streams (three
	 * passes in flight: more measure worse -- each gets fewer slots, none gets done; profiles/r05/interactive_passes_scratch_sets_probe.txt), so that the next pass's waves
	 * take the compute units this one's leave (a 1080p pass of one sample per pixel is ramp-up and tail from end to end: one
	 * stream 0.254 ms per pass, docs/lab/r05.md section 3); only the publish steps -- sums += pass, in pass order -- wait for
	 * each other. */
	const unsigned which = ctx->launches % RT_LAUNCH_SETS;
	hipStream_t stream = (hipStream_t) rt_stream(ctx, (int) (ctx->launches % RT_PASSES_IN_FLIGHT));
	if (!stream) return RT_ERR_DEVICE;
	float *const d_low = g.d_low[which];

	rt_launch L;
	memset(&L, 0, sizeof(L));
	rt_camera_basis basis;
	rt_camera_basis_for(&ctx->camera, (float) g.width / g.height, &basis);      /* main.c:281: full-size aspect */
	const Vector3 *src[4] = { &basis.pos, &basis.lower_left_corner, &basis.horizontal, &basis.vertical };
	float *dst[4] = { L.pos, L.llc, L.horiz, L.vert };
	for (int k = 0; k < 4; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
	L.light_index = ctx->light_index;
	memcpy(L.light_pos, ctx->light_pos, sizeof(L.light_pos));
	L.only_light_emits = ctx->only_light_emits ? 1 : 0;
	L.num_objects = ctx->num_objects;
	const bool batch = samples > 1;             /* (s == 1: the caller's business) */
	/* (the reference's low-resolution frame has one column more than it shows, main.c:286; a batch renders onto the sums, which
	 * have the frame's own width, and leaves that column out -- nothing ever reads it) */
	L.width = batch ? lw : lcw; L.height = lh; L.local_rows = lh;
	L.spp = samples; L.max_bounces = g.max_bounces; L.seed = g.seed;
	L.u_den = lw - 1; L.v_den = lh - 1; L.pix_scale = s; L.pix_width = g.width; L.sample_base = g.passes;
	L.row_block = 8; L.rank = 0; L.world = 1;
	if (g.world > 1) {
		/* this rank's share of the low-resolution rows: the rows that cover its blocks of 16 frame rows */
		L.row_block = RT_PROGRESSIVE_ROW_BLOCK / s; L.rank = g.rank; L.world = g.world;
		const int blocks = (lh + L.row_block - 1) / L.row_block;
		const int mine = blocks > g.rank ? (blocks - g.rank + g.world - 1) / g.world : 0;
		L.local_rows = mine * L.row_block;
	}
	L.sky = ctx->d_sky; L.sky_w = ctx->sky_w; L.sky_h = ctx->sky_h; L.sky_wm1 = (float) (ctx->sky_w - 1); L.sky_hm1 = (float) (ctx->sky_h - 1);
	L.frame = d_low;
	L.sum_onto = batch ? g.d_accum : nullptr;
	/* (the flags are kept with the lists: a camera position pays once for all its passes -- reckoned as sixteen) */
	L.skip_known_taps = classify_pixels(ctx, samples > 16 ? samples : 16);
	L.lit_cells = (ctx->have_lit && !ctx->tuning.trace_known_taps) ? ctx->d_lit_cells + (audit_log2_of_launch(ctx, ctx->launches) != 0 ? ctx->lit_cells_capacity : 0) : nullptr;
	L.lit_grids = ctx->d_lit_grids;
	L.geom = ctx->d_geom; L.shade = ctx->d_shade;
	if (ctx->cull.num_clusters > 0 && !ctx->tuning.test_every_object) {
		L.clusters = ctx->d_clusters; L.num_clusters = ctx->cull.num_clusters; L.cull_margin = ctx->cull.margin; L.cull_origin_max = ctx->cull.origin_max;
	}
	{ const int rc = prepare_launch(ctx, L, ctx->launches); if (rc != RT_OK) { unpublish_launch(ctx); return rc; } }
	{ const int rc = order_behind_previous(ctx, stream); if (rc != RT_OK) { unpublish_launch(ctx); return rc; } }
	if (batch) { const int rc = sums_wait(ctx, stream); if (rc != RT_OK) { unpublish_launch(ctx); return rc; } }     /* (its trace kernel reads the sums so far) */
	if (ctx->tuning.poison_frame) {
		const hipError_t pe = hipMemsetAsync(d_low, 0xff, g.low_bytes, stream);
		if (pe != hipSuccess) { unpublish_launch(ctx); return fail(RT_ERR_DEVICE, "hipMemsetAsync(poison): %s", hipGetErrorString(pe)); }
	}
	if (L.local_rows <= 0) {
		unpublish_launch(ctx);              /* nothing is launched for a rank without rows at this scale */
		/* a rank without rows at this scale -- its few frame rows lie below the last whole low-resolution row -- renders
		 * nothing and adds nothing, but the pass counts (main.c:396): those rows are divided by the same count as all others */
		{ const int rc = sums_wait(ctx, stream); if (rc != RT_OK) return rc; }
		for (int k = 0; k < samples; k++)
			HIP_TRY(rt_launch_accumulate(g.d_accum, d_low, g.width, g.height, s, lcw, lh, 1.0f / (s * s), nullptr, rt_launch_expect{ 0u, 0, 0u }, g.d_count,
			                             RT_PROGRESSIVE_ROW_BLOCK, g.rank, g.world, 0, stream));
		{ const int rc = sums_mark(ctx, stream); if (rc != RT_OK) return rc; }
		g.passes += samples;
		if (g.scale > 1) g.scale >>= 1;
		if (weight_out) *weight_out = 1.0f / (s * s);
		return RT_OK;
	}
	/* A pass differs from the last pass of its scratch set (RT_LAUNCH_SETS passes ago) in its sample number only, once the scale
	 * ladder has reached full resolution and until the camera moves: the camera rays, their hits and the sky pixels in the set's
	 * low-resolution frame are the same, so rt_primary_pass's output is kept (a sixth of a 1080p pass).  The key is
	 * everything that output depends on. */
	uint64_t key = 0;
	{
		rt_launch K = L;
		K.seed = 0; K.sample_base = 0; K.max_bounces = 0; K.lit_cells = nullptr; K.lit_grids = nullptr; K.lit_grids_in_lds = 0;
		K.launch_id = 0;
		/* (K.audit_taps stays in the key: the records of an audited launch carry the camera-ray hit points' audit marks, which a kernel
		 * that is not the audit variant does not mask out -- the pass the background audit picks, one in 61, traces its camera rays
		 * again, and so does the next pass of its scratch set) */
		K.pix = nullptr; K.pix_count = nullptr; K.control = nullptr; K.frame = nullptr;     /* the scratch set's own addresses: the same output in any set has the same key */
		key = 0xcbf29ce484222325ull ^ ctx->input_version;
		const unsigned char *b = reinterpret_cast<const unsigned char*>(&K);
		for (size_t i = 0; i < sizeof(K); i++) key = (key ^ b[i]) * 0x100000001b3ull;
		if (key == 0) key = 1;
	}
	rt_context::launch_slot &sl = ctx->slot[which];
	const bool reuse = !ctx->tuning.poison_frame && !batch && sl.lists_key == key;      /* (a batch's sky pixels are sums: never the same twice) */
	sl.lists_key = 0;
	{
		const bool audit = L.audit_taps != 0u;
		/* Resident workgroups per CU.  Batches run one after the other (each needs the sums of the one before): all the slots.  Single
		 * passes enqueued ahead of the GPU: ONE of four -- three passes are resident at a time; a 1080p pass is five pixels per lane
		 * with all the slots, and a wave then spends more rounds on the last of its paths (ten bounces, a few lanes) than on all the
		 * others: 15.3 rounds per wave and pass, 7.6 of them nearly empty, against 44.4 / 8.8 for a quarter of the waves, i.e. 62 700
		 * against 45 500 rounds per pass (profiles/r05/progressive_stats.txt); 0.191 -> 0.174 ms per pass. */
		int wg = ctx->tuning.workgroups_per_cu;
		if (!batch && wg < 1) { wg = workgroups_per_cu_for(ctx, stream, (long long) L.width * L.local_rows); if (wg == 2) wg = 1; }
		const hipError_t le = rt_launch_trace(L, RT_KERNEL_AUTO, ctx->scene_fast_ok, trace_kernel_for(ctx, audit), sl.d_counter, nullptr, sl.started, ctx->num_cus,
		                                      wg, stream, reuse, &sl.expect, nullptr, audit);
		if (le != hipSuccess) { unpublish_launch(ctx); return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(le)); }
	}
	if (!reuse) ctx->primary_passes++;
	sl.lists_key = ctx->tuning.poison_frame || batch ? 0 : key;
	const float weight = 1.0f / (s * s);                                         /* main.c:278 */
	/* accum += pass * weight and accum_counts += weight (main.c:394-396), both on the device and both skipped for a pass
	 * that rt_cancel() cut short (main.c:382): the count can never include a pass the buffer does not */
	/* (on the pass's own stream: a stream for the publish steps alone, of the highest priority, was tried -- a hop between queues
	 * costs 50 ... 80 us here, and priority does not make a workgroup slot where persistent kernels hold them all) */
	hipStream_t pub = stream;
	{ const int rc = sums_wait(ctx, pub); if (rc != RT_OK) return rc; }
	if (batch)        /* the launch wrote sums-so-far + its samples: they become the sums, and `samples` passes count */
		HIP_TRY(rt_launch_commit_sums(g.d_accum, d_low, (size_t) g.width * L.local_rows * 3, samples, L.control, sl.expect, g.d_count, pub));
	else
	HIP_TRY(rt_launch_accumulate(g.d_accum, d_low, g.width, g.height, s, lcw, lh, 1.0f / (s * s), L.control, sl.expect, g.d_count,
	                             RT_PROGRESSIVE_ROW_BLOCK, g.rank, g.world, g.rows, pub));
	{ const int rc = sums_mark(ctx, pub); if (rc != RT_OK) return rc; }
	/* (the launch is done when it is published: the scratch set's next launch, RT_LAUNCH_SETS passes on, clears the control words
	 * the publish step reads and writes the low-resolution frame it adds) */
	{ const int rc = mark_launch(ctx, pub); if (rc != RT_OK) return rc; }
	g.passes += samples;
	if (g.scale > 1) g.scale >>= 1;                                              /* main.c:402-403 */
	if (weight_out) *weight_out = weight;
	return RT_OK;
}

int rt_progressive_pass(rt_context *ctx, float *weight_out) { return progressive_launch(ctx, 1, weight_out); }

/* `count` worker iterations.  While the ladder is below full resolution they are what `count` calls of rt_progressive_pass()
 * are; at full resolution the rest go into launches of up to RT_PROGRESSIVE_BATCH samples per pixel.  Same sums, same count,
 * same sample numbers: the frame is bit-identical -- a 1080p pass of one sample per pixel is 0.18 ms, most of it waves running out
 * their last few paths, sixty of them in one launch take half the time of sixty launches. */
int rt_progressive_passes(rt_context *ctx, int count)
{
	if (!ctx || !ctx->prog.active) return fail(RT_ERR_STATE, "rt_progressive_passes: call rt_progressive_begin first");
	if (count < 1) return fail(RT_ERR_ARGUMENT, "rt_progressive_passes: count must be at least 1");
	while (count > 0) {
		/* (fewer than RT_PROGRESSIVE_BATCH_MIN passes are faster one by one: a batch renders its camera rays again -- single passes
		 * keep them from the pass before last -- and copies the sums once more; profiles/r04/progressive_rate.txt) */
		const int n = ctx->prog.scale > 1 || count < RT_PROGRESSIVE_BATCH_MIN ? 1 : (count < RT_PROGRESSIVE_BATCH ? count : RT_PROGRESSIVE_BATCH);
		const int rc = progressive_launch(ctx, n, nullptr);
		if (rc != RT_OK) return rc;
		count -= n;
	}
	return RT_OK;
}

int rt_progressive_resolve(rt_context *ctx, Vector3 *frame_out)
{
	if (!ctx || !ctx->prog.active) return fail(RT_ERR_STATE, "rt_progressive_resolve: call rt_progressive_begin first");
	if (!frame_out) return fail(RT_ERR_ARGUMENT, "rt_progressive_resolve: frame_out is NULL");
	auto &g = ctx->prog;
	if (g.passes == 0) return fail(RT_ERR_STATE, "rt_progressive_resolve: nothing accumulated yet");
	HIP_TRY(hipSetDevice(ctx->device));
	/* the count the device holds first (update_frame() waits for a column's count before it divides by it, main.c:461-464): when
	 * every pass so far was cut short there is nothing to divide, and frame_out is left as it is */
	float count = 0.0f;
	{ const int rc = rt_progressive_count(ctx, &count); if (rc != RT_OK) return rc; }
	if ((double) count < 0.0001)
		return fail(RT_ERR_STATE, "rt_progressive_resolve: nothing accumulated yet (every pass so far was cancelled)");
	/* frame = accum * (1 / count) (main.c:467-477) */
	{ const int rc = sums_wait(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	HIP_TRY(rt_launch_resolve(g.d_accum, g.d_out, (size_t) g.width * g.rows * 3, g.d_count, ctx->stream));
	{ const int rc = sums_mark(ctx, ctx->stream); if (rc != RT_OK) return rc; }      /* (the next publish step must not overtake the read) */
	HIP_TRY(hipMemcpyAsync(frame_out, g.d_out, g.accum_bytes, hipMemcpyDeviceToHost, ctx->stream));     /* one rank of several: its rows */
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return RT_OK;
}

/* one rank's rows, resolved, left on the device (rt_multi_progressive_resolve, or the host's own collective, gathers them) */
int rt_progressive_resolve_device(rt_context *ctx, void **d_strip)
{
	if (!ctx || !ctx->prog.active || !d_strip) return fail(RT_ERR_STATE, "rt_progressive_resolve: call rt_progressive_begin first");
	auto &g = ctx->prog;
	HIP_TRY(hipSetDevice(ctx->device));
	{ const int rc = sums_wait(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	HIP_TRY(rt_launch_resolve(g.d_accum, g.d_out, (size_t) g.width * g.rows * 3, g.d_count, ctx->stream));
	{ const int rc = sums_mark(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	*d_strip = g.d_out;
	return RT_OK;
}

} /* extern "C" */

int rt_progressive_count(rt_context *ctx, float *count)
{
	if (!ctx || !ctx->prog.active || !count) return fail(RT_ERR_STATE, "rt_progressive_state: call rt_progressive_begin first");
	HIP_TRY(hipSetDevice(ctx->device));
	float *h_count = reinterpret_cast<float*>(&ctx->h_words[RT_COUNT_WORD]);
	{ const int rc = sums_wait(ctx, ctx->stream); if (rc != RT_OK) return rc; }
	HIP_TRY(hipMemcpyAsync(h_count, ctx->prog.d_count, RT_COUNT_WORDS * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	*count = *h_count;
	/* passes whose launch was incomplete were not published (rt_accumulate / rt_commit_sums looked at the launch's control words
	 * on the device): the sums are those of the other passes, and the caller is told */
	const unsigned int incomplete = ctx->h_words[RT_COUNT_WORD + RT_COUNT_INCOMPLETE];
	if (incomplete) {
		/* (a launch that did not end as launches end may not have left its scratch set's counters as the next one expects them:
		 * no set's lists are kept, so every set is cleared by its next launch) */
		for (auto &sl : ctx->slot) sl.lists_key = 0;
		return fail(RT_ERR_DEVICE, "rt_progressive: %u launch(es) since the last rt_progressive_invalidate() were incomplete and were not published (rt_last_launch_report() has the most recent launch's numbers)", incomplete);
	}
	return RT_OK;
}

extern "C" {

int rt_progressive_state(rt_context *ctx, int *next_scale, float *count, uint32_t *generation, int *passes)
{
	if (!ctx || !ctx->prog.active) return fail(RT_ERR_STATE, "rt_progressive_state: call rt_progressive_begin first");
	if (next_scale) *next_scale = ctx->prog.scale;
	if (count) {          /* the passes enqueued so far, as far as they were published: waits for them */
		const int rc = rt_progressive_count(ctx, count);
		if (rc != RT_OK) return rc;
	}
	if (generation) *generation = ctx->prog.generation;
	if (passes) *passes = ctx->prog.passes;
	return RT_OK;
}

int rt_selftest(rt_context *ctx, int which, uint64_t seed, int blocks, int iters, unsigned long long out[8])
{
	if (!ctx || !out || which < 0 || which > 8 || blocks < 1 || iters < 1)
		return fail(RT_ERR_ARGUMENT, "rt_selftest: bad argument");
	HIP_TRY(hipSetDevice(ctx->device));
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void**) &d, 8 * sizeof(unsigned long long)));
	hipError_t e = hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), ctx->stream);
	if (e == hipSuccess) e = rt_launch_selftest(which, seed, blocks, iters, d, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(out, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void) hipFree(d);
	if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_selftest: %s", hipGetErrorString(e));
	return RT_OK;
}

int rt_synchronize(rt_context *ctx)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_synchronize: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	return wait_for_launches(ctx);
}

int rt_profile_enable(rt_context *ctx, int on)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_profile_enable: NULL context");
	ctx->profiling = on != 0;
	ctx->profiling_split = on == 2;
	return RT_OK;
}

int rt_profile_collect(rt_context *ctx, double *kernel_ms_total, int *launches)
{
	return rt_profile_collect_span(ctx, kernel_ms_total, launches, nullptr);
}

int rt_profile_collect_span(rt_context *ctx, double *kernel_ms_total, int *launches, double *span_ms)
{
	return rt_profile_collect_split(ctx, kernel_ms_total, launches, span_ms, nullptr);
}

int rt_profile_collect_split(rt_context *ctx, double *kernel_ms_total, int *launches, double *span_ms, double *primary_ms_total)
{
	if (!ctx) return fail(RT_ERR_ARGUMENT, "rt_profile_collect: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	double total = 0;
	int n = 0;
	if (span_ms) {
		*span_ms = 0;
		if (!ctx->events.empty()) {
			float ms = 0;
			HIP_TRY(hipEventSynchronize(ctx->events.back().second));
			HIP_TRY(hipEventElapsedTime(&ms, ctx->events.front().first, ctx->events.back().second));
			*span_ms = ms;
		}
	}
	double primary = 0;
	for (auto &p : ctx->events) {
		HIP_TRY(hipEventSynchronize(p.second));
		float ms = 0, pms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
		if (p.mid) HIP_TRY(hipEventElapsedTime(&pms, p.first, p.mid));
		total += ms; primary += pms; n++;
		ctx->event_pool.push_back(p.first); if (p.mid) ctx->event_pool.push_back(p.mid); ctx->event_pool.push_back(p.second);
	}
	ctx->events.clear();
	if (primary_ms_total) *primary_ms_total = primary;
	if (kernel_ms_total) *kernel_ms_total = total;
	if (launches) *launches = n;
	return RT_OK;
}

} /* extern "C" */
