/*
 * rt_multi.cpp -- one frame on several GPUs of one node from ONE host process, behind the C ABI
 * (include/rt_hip.h, rt_multi_*).
 *
 * The reference fans its work out inside the host binary: start_workers() creates one pthread per image
 * column and every worker adds its passes into the shared accumulation buffer (main.c:695-718, 324-414),
 * while the main thread presents what has been accumulated so far (main.c:450-482).
 * Here the fan-out is over GPUs: strip s is the row blocks b with b % n == s (the interleaved partition
 * rt_render_device() implements for one rank) and device i renders strip rt_strip_of_rank(i, n) -- device 0, which also
 * gathers, de-interleaves and copies out, the last one, which is never the longest; every device renders its strip concurrently, ONE grouped
 * ncclGather over xGMI brings the strips to device 0, a de-interleave kernel there puts the rows in frame order
 * and the frame is copied to the caller's host buffer -- what update_frame() hands to move_frame_to_the_gpu()
 * (main.c:467-479).  Frames are SUBMITTED and WAITED for (rt_multi_frame_submit / rt_multi_frame_wait), so that
 * the gather, the de-interleave and the copy of frame k run beside the renders of frames k+1 and k+2;
 * rt_multi_render() is submit + wait.
 *
 * Streams per device: the context's render streams (frame k on stream k % RT_LAUNCH_SETS: consecutive strips overlap on
 * the GPU, three of them resident -- one draining, one running, one starting --, rt_api.cpp) and one high-priority stream for
 * the collective (and, on device 0, the de-interleave); device 0 has one more for the copy to the host.  Strips rotate
 * through four buffers per device: the persistent trace kernels of the following frames hold every compute unit until they
 * drain, so the collective of frame k only gets to run then, and a render stream must not wait for it -- it waits (through
 * an event) for the gather four frames back, whose strip buffer it reuses.
 *
 * RCCL is loaded with dlopen() on the first multi-device create (as rt_jit.cpp does for hiprtc) and the handful of
 * entry points used are declared here, so the library needs neither RCCL's headers nor its .so at build or load time,
 * and single-GPU hosts never touch it.  There is no fallback: if RCCL cannot be loaded or initialised,
 * rt_multi_create() with n > 1 fails with RT_ERR_DEVICE.
 */
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <new>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_internal.h"

namespace {

/* the part of RCCL's C API (rccl.h, NCCL 2.x ABI) this file calls */
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                    /* ncclSuccess = 0 */
enum { ncclSuccess_ = 0, ncclFloat_ = 7 };   /* ncclDataType_t: ncclFloat32 = 7 */

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*comm_init_all)(ncclComm_t *comms, int ndev, const int *devlist) = nullptr;
	ncclResult_t (*comm_destroy)(ncclComm_t comm) = nullptr;
	ncclResult_t (*gather)(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, int root,
	                       ncclComm_t comm, hipStream_t stream) = nullptr;
	ncclResult_t (*group_start)() = nullptr;
	ncclResult_t (*group_end)() = nullptr;
	const char  *(*error_string)(ncclResult_t) = nullptr;
	ncclResult_t (*comm_count)(const ncclComm_t comm, int *count) = nullptr;       /* optional: rt_multi_collective_info() */
	ncclResult_t (*comm_device)(const ncclComm_t comm, int *device) = nullptr;
	ncclResult_t (*comm_rank)(const ncclComm_t comm, int *rank) = nullptr;
	ncclResult_t (*get_version)(int *version) = nullptr;
	bool ok = false;
};

Rccl &rccl()
{
	static Rccl r;
	if (r.lib) return r;
	const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" };
	for (const char *n : names) {
		r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (r.lib) break;
	}
	if (!r.lib) return r;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
	SYM(comm_init_all, "ncclCommInitAll"); SYM(comm_destroy, "ncclCommDestroy"); SYM(gather, "ncclGather");
	SYM(group_start, "ncclGroupStart");    SYM(group_end, "ncclGroupEnd");       SYM(error_string, "ncclGetErrorString");
	SYM(comm_count, "ncclCommCount");      SYM(comm_device, "ncclCommCuDevice"); SYM(comm_rank, "ncclCommUserRank"); SYM(get_version, "ncclGetVersion");
#undef SYM
	r.ok = r.comm_init_all && r.comm_destroy && r.gather && r.group_start && r.group_end && r.error_string;
	return r;
}

constexpr int STRIP_BUFFERS = RT_LAUNCH_SETS + 1;     /* strips in flight: RT_LAUNCH_SETS renders and the gather of the one before them */

} // namespace

struct rt_multi {
	int n = 0;
	bool force_collective = false;              /* rt_tuning.force_collective: gather + de-interleave even with one device */
	bool one_device = false;                    /* rt_multi_create_on_one_device(): n contexts on ONE device, the gather is n device copies (testing aid) */
	std::vector<int>          devices;
	std::vector<rt_context *> ctx;
	std::vector<ncclComm_t>   comms;            /* made on first use of the collective */

	struct per_device {
		float      *d_strip[STRIP_BUFFERS] = {};
		hipEvent_t  gathered[STRIP_BUFFERS] = {};   /* behind the gather that read d_strip[j] */
		bool        gathered_set[STRIP_BUFFERS] = {};
		hipStream_t gather_stream = nullptr;    /* the collective (device 0: and the de-interleave) */
	};
	std::vector<per_device> dev;
	size_t strip_floats = 0;                    /* capacity of every strip buffer */

	/* device 0 */
	float *d_strips[STRIP_BUFFERS] = {};   /* n strips back to back: the gather's destination */
	size_t strips_floats = 0;
	hipStream_t copy_stream = nullptr;
	unsigned int *h_control = nullptr;          /* pinned: the control words of the launch of (slot, device) at (slot * n + device) * RT_CTL_COPY_WORDS (rt_internal.h) */
	struct frame_slot {
		float     *d_frame = nullptr;           /* the frame in row order */
		size_t     floats = 0;
		hipEvent_t assembled = nullptr;         /* behind the de-interleave into d_frame */
		hipEvent_t copied = nullptr;            /* behind the copy to the caller's memory */
		bool       busy = false;
		bool       plain = false;               /* the frame went through rt_frame_submit() of the only context */
		std::vector<rt_launch_expect> expect;   /* per device: what its launch must leave in its control words (rt_judge_launch) */
	} fq[RT_FRAME_SLOTS];
	unsigned long long frames = 0;              /* frame k: render stream k % RT_LAUNCH_SETS, strip buffers k % STRIP_BUFFERS */
	int prog_w = 0, prog_h = 0;                 /* rt_multi_progressive_begin's frame */

	/* rt_multi_profile_enable(): timed events around every phase of every frame, per device (rt_multi_profile_collect) */
	struct phase_marks { hipEvent_t render_begins = nullptr, rendered = nullptr, gathered = nullptr, assembled = nullptr, copied = nullptr; };
	bool profiling = false;
	std::vector<std::vector<phase_marks>> marks;    /* [device][frame since rt_multi_profile_enable] */
};

static void free_marks(rt_multi *m);
#define RT_PROFILE_FRAMES_MAX 4096u      /* frames rt_multi_profile_enable() records before it turns itself off (five events per device and frame) */
/* a timed event recorded on `stream` now (profiling only) */
static hipError_t mark_now(hipEvent_t *e, hipStream_t stream)
{
	hipError_t rc = hipEventCreate(e);
	if (rc == hipSuccess) rc = hipEventRecord(*e, stream);
	return rc;
}

#define MULTI_HIP(expr)                                                                      \
	do {                                                                                    \
		hipError_t e_ = (expr);                                                             \
		if (e_ != hipSuccess)                                                               \
			return rt_fail(RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

/* the group's RCCL communicators, one per device, made by one ncclCommInitAll (single process) */
static int init_communicators(rt_multi *m)
{
	if (!m->comms.empty()) return RT_OK;
	Rccl &r = rccl();
	if (!r.ok) return rt_fail(RT_ERR_DEVICE, "rt_multi: RCCL is not available (dlopen librccl.so failed)");
	m->comms.assign((size_t) m->n, nullptr);
	const ncclResult_t rc = r.comm_init_all(m->comms.data(), m->n, m->devices.data());
	if (rc != ncclSuccess_) {
		m->comms.clear();
		return rt_fail(RT_ERR_DEVICE, "ncclCommInitAll: %s", r.error_string(rc));
	}
	return RT_OK;
}

/* Everything the handle has enqueued on any device has finished (error paths: nothing may still read or write the
 * buffers the handle owns when a call returns a failure). */
static void drain(rt_multi *m)
{
	for (int i = 0; i < m->n; i++) {
		if (!m->ctx[(size_t) i]) continue;
		(void) rt_synchronize(m->ctx[(size_t) i]);
		(void) hipSetDevice(m->devices[(size_t) i]);
		if (m->dev[(size_t) i].gather_stream) (void) hipStreamSynchronize(m->dev[(size_t) i].gather_stream);
	}
	if (m->copy_stream) { (void) hipSetDevice(m->devices[0]); (void) hipStreamSynchronize(m->copy_stream); }
}

/* streams, events and buffers of the pipelined frame loop for frames of W x H, made or grown on demand */
static int prepare(rt_multi *m, int W, int H, int rb, int slot)
{
	const int n = m->n;
	const size_t strip_floats = (size_t) rt_strip_rows(H, rb, n) * W * 3, frame_floats = (size_t) H * W * 3;
	int least = 0, greatest = 0;
	for (int i = 0; i < n; i++) {
		rt_multi::per_device &d = m->dev[(size_t) i];
		MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
		if (!d.gather_stream) {
			if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
			if (hipStreamCreateWithPriority(&d.gather_stream, hipStreamNonBlocking, greatest) != hipSuccess)
				MULTI_HIP(hipStreamCreateWithFlags(&d.gather_stream, hipStreamNonBlocking));
			for (int j = 0; j < STRIP_BUFFERS; j++) {
				MULTI_HIP(hipEventCreateWithFlags(&d.gathered[j], hipEventDisableTiming));
			}
		}
	}
	if (strip_floats > m->strip_floats) {
		drain(m);                                   /* nothing in flight reads the buffers that are replaced */
		for (int i = 0; i < n; i++) {
			rt_multi::per_device &d = m->dev[(size_t) i];
			MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
			for (int j = 0; j < STRIP_BUFFERS; j++) { (void) hipFree(d.d_strip[j]); d.d_strip[j] = nullptr; d.gathered_set[j] = false; }
		}
		m->strip_floats = 0;
		for (int i = 0; i < n; i++)
			for (int j = 0; j < STRIP_BUFFERS; j++) {
				MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
				MULTI_HIP(hipMalloc((void **) &m->dev[(size_t) i].d_strip[j], strip_floats * sizeof(float)));
			}
		m->strip_floats = strip_floats;
	}
	MULTI_HIP(hipSetDevice(m->devices[0]));
	if (strip_floats * (size_t) n > m->strips_floats) {
		drain(m);
		MULTI_HIP(hipSetDevice(m->devices[0]));
		for (int j = 0; j < STRIP_BUFFERS; j++) { (void) hipFree(m->d_strips[j]); m->d_strips[j] = nullptr; }
		m->strips_floats = 0;
		for (int j = 0; j < STRIP_BUFFERS; j++)
			MULTI_HIP(hipMalloc((void **) &m->d_strips[j], strip_floats * (size_t) n * sizeof(float)));
		m->strips_floats = strip_floats * (size_t) n;
	}
	if (!m->copy_stream) {
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
		if (hipStreamCreateWithPriority(&m->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess)
			MULTI_HIP(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
	}
	if (!m->h_control) {
		MULTI_HIP(hipHostMalloc((void **) &m->h_control, (size_t) RT_FRAME_SLOTS * (size_t) n * RT_CTL_COPY_BYTES, hipHostMallocPortable));   /* every device writes its words */
		memset(m->h_control, 0, (size_t) RT_FRAME_SLOTS * (size_t) n * RT_CTL_COPY_BYTES);
	}
	rt_multi::frame_slot &f = m->fq[slot];
	f.expect.assign((size_t) n, rt_launch_expect{ 0u, 0, 0u });
	if (!f.copied) MULTI_HIP(hipEventCreateWithFlags(&f.copied, hipEventDisableTiming));
	if (!f.assembled) MULTI_HIP(hipEventCreateWithFlags(&f.assembled, hipEventDisableTiming));
	if (frame_floats > f.floats) {
		(void) hipFree(f.d_frame); f.d_frame = nullptr; f.floats = 0;   /* (the slot is idle: its last copy was waited for) */
		MULTI_HIP(hipMalloc((void **) &f.d_frame, frame_floats * sizeof(float)));
		f.floats = frame_floats;
	}
	return RT_OK;
}

extern "C" {

static int multi_create(rt_multi **out, const int *device_ids, int n, bool one_device)
{
	if (!out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: out is NULL");
	*out = nullptr;
	if (!device_ids || n < 1 || n > 64) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: need 1..64 device ids");
	for (int i = 0; i < n && !one_device; i++)
		for (int k = 0; k < i; k++)
			if (device_ids[i] == device_ids[k])
				return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: device %d listed twice", device_ids[i]);
	rt_multi *m = new (std::nothrow) rt_multi();
	if (!m) return rt_fail(RT_ERR_MEMORY, "rt_multi_create: out of host memory");
	m->n = n;
	m->one_device = one_device;
	m->devices.assign(device_ids, device_ids + n);
	m->ctx.assign((size_t) n, nullptr);
	m->dev.resize((size_t) n);
	m->marks.assign((size_t) n, {});
	for (int i = 0; i < n; i++) {
		const int rc = rt_create(&m->ctx[(size_t) i], device_ids[i]);
		if (rc != RT_OK) { rt_multi_destroy(m); return rc; }          /* rt_last_error() holds rt_create's text */
	}
	if (n > 1 && !one_device) {
		const int rc = init_communicators(m);
		if (rc != RT_OK) { rt_multi_destroy(m); return rc; }
	}
	*out = m;
	return RT_OK;
}

int rt_multi_create(rt_multi **out, const int *device_ids, int n) { return multi_create(out, device_ids, n, false); }

/* TESTING AID for boxes with one GPU: a group of n contexts that all live on `device_id`.  Everything the n-device path does
 * runs -- n strips rendered by n contexts on their own streams, three strip buffers each, the rotated hand-out, the
 * de-interleave, the frame queue, the ladder -- except RCCL, which refuses two ranks on one device: the gather is the n
 * device-to-device copies it amounts to there.  Frames are bit-identical to rt_render()'s; not a performance configuration. */
int rt_multi_create_on_one_device(rt_multi **out, int device_id, int n)
{
	if (n < 1 || n > 64) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create_on_one_device: need 1..64 contexts");
	int ids[64];
	for (int i = 0; i < n; i++) ids[i] = device_id;
	return multi_create(out, ids, n, true);
}

void rt_multi_destroy(rt_multi *m)
{
	if (!m) return;
	drain(m);
	free_marks(m);
	for (int i = 0; i < m->n; i++) {
		if (!m->ctx[(size_t) i]) continue;
		rt_multi::per_device &d = m->dev[(size_t) i];
		(void) hipSetDevice(m->devices[(size_t) i]);
		for (int j = 0; j < STRIP_BUFFERS; j++) {
			(void) hipFree(d.d_strip[j]);
			if (d.gathered[j]) (void) hipEventDestroy(d.gathered[j]);
		}
		if (d.gather_stream) (void) hipStreamDestroy(d.gather_stream);
		if (i == 0) {
			for (int j = 0; j < STRIP_BUFFERS; j++) (void) hipFree(m->d_strips[j]);
			for (auto &f : m->fq) { (void) hipFree(f.d_frame); if (f.copied) (void) hipEventDestroy(f.copied); if (f.assembled) (void) hipEventDestroy(f.assembled); }
			if (m->copy_stream) (void) hipStreamDestroy(m->copy_stream);
			if (m->h_control) (void) hipHostFree(m->h_control);
		}
	}
	for (ncclComm_t c : m->comms) if (c) (void) rccl().comm_destroy(c);
	for (rt_context *c : m->ctx) rt_destroy(c);
	delete m;
}

int rt_multi_size(const rt_multi *m) { return m ? m->n : 0; }

rt_context *rt_multi_context(rt_multi *m, int i) { return m && i >= 0 && i < m->n ? m->ctx[(size_t) i] : nullptr; }

#define FOR_ALL(call)                                                            \
	do {                                                                         \
		if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi: NULL handle");        \
		for (int i = 0; i < m->n; i++) {                                         \
			rt_context *ctx = m->ctx[(size_t) i];                                \
			const int rc = (call);                                               \
			if (rc != RT_OK) return rc;                                          \
		}                                                                        \
		return RT_OK;                                                            \
	} while (0)

/* every device holds the whole scene and skybox (a 69 KB scene and a 100 MB skybox against 288 GB) */
int rt_multi_set_scene(rt_multi *m, const Scene *scene)       { FOR_ALL(rt_set_scene(ctx, scene)); }
int rt_multi_set_skybox(rt_multi *m, const Cubemap *skybox)   { FOR_ALL(rt_set_skybox(ctx, skybox)); }
int rt_multi_set_camera(rt_multi *m, const rt_camera *camera) { FOR_ALL(rt_set_camera(ctx, camera)); }
int rt_multi_set_tuning(rt_multi *m, const rt_tuning *tuning) { FOR_ALL(rt_set_tuning(ctx, tuning)); }
int rt_multi_set_test_knobs(rt_multi *m, const rt_test_knobs *knobs)
{
	if (m && knobs && knobs->size == sizeof(rt_test_knobs)) m->force_collective = knobs->force_collective != 0;
	FOR_ALL(rt_set_test_knobs(ctx, knobs));
}
int rt_multi_compile_scene(rt_multi *m)                       { FOR_ALL(rt_compile_scene(ctx)); }

/* the strips of frame buffer j to device 0: ONE grouped ncclGather, each rank's part on its own collective stream -- or, for a
 * group whose contexts share one device (testing aid), the copies that gather amounts to there */
static int gather_strips(rt_multi *m, int j, size_t strip_floats, const std::vector<hipEvent_t> &reported)
{
	const int n = m->n;
	if (m->one_device) {
		/* on the ROOT's collective stream, where the collective's receives run too: behind the root's own earlier work on the
		 * destination (the de-interleave that read d_strips[j] four frames back) and behind every context's render AND the copy of
		 * its launch's control words (`reported`: on the context's own collective stream, behind the render) -- a real collective
		 * orders the root behind those by itself: a rank's send sits behind them on its stream, and the root's receive completes
		 * after every send */
		for (int i = 0; i < n; i++) {
			hipError_t e = i == 0 ? hipSuccess : hipStreamWaitEvent(m->dev[0].gather_stream, reported[(size_t) i], 0);
			if (e == hipSuccess)
				e = hipMemcpyAsync(m->d_strips[j] + (size_t) i * strip_floats, m->dev[(size_t) i].d_strip[j], strip_floats * sizeof(float),
				                   hipMemcpyDeviceToDevice, m->dev[0].gather_stream);
			if (e != hipSuccess) return rt_fail(RT_ERR_DEVICE, "rt_multi (one device): %s", hipGetErrorString(e));
		}
		return RT_OK;
	}
	Rccl &r = rccl();
	ncclResult_t nrc = r.group_start();
	for (int i = 0; i < n && nrc == ncclSuccess_; i++)
		nrc = r.gather(m->dev[(size_t) i].d_strip[j], i == 0 ? m->d_strips[j] : nullptr, strip_floats, ncclFloat_, 0,
		               m->comms[(size_t) i], m->dev[(size_t) i].gather_stream);
	{ const ncclResult_t end = r.group_end(); if (nrc == ncclSuccess_) nrc = end; }
	if (nrc != ncclSuccess_) return rt_fail(RT_ERR_DEVICE, "ncclGather: %s", r.error_string(nrc));
	return RT_OK;
}

static int multi_frame_submit(rt_multi *m, const rt_render_params *params, int slot, Vector3 *frame_out, void **d_frame, void **hip_event)
{
	if (!m || !params) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_submit: NULL argument");
	if (slot < 0 || slot >= RT_FRAME_SLOTS) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_submit: slot %d not in [0,%d)", slot, RT_FRAME_SLOTS);
	rt_multi::frame_slot &f = m->fq[slot];
	if (f.busy) return rt_fail(RT_ERR_STATE, "rt_multi_frame_submit: slot %d holds a frame that has not been waited for", slot);
	if (m->n == 1 && !m->force_collective) {    /* one device: the strip is the frame */
		rt_render_params p = *params;
		p.rank = 0; p.world = 1;
		const int rc = frame_out ? rt_frame_submit(m->ctx[0], &p, slot, frame_out) : rt_frame_submit_device(m->ctx[0], &p, slot, d_frame, hip_event);
		if (rc == RT_OK) { f.busy = true; f.plain = true; }
		return rc;
	}
	if (params->width < 2 || params->height < 2 || params->row_block < 1)
		return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_submit: bad frame %dx%d / row_block %d", params->width, params->height, params->row_block);
	const int n = m->n, W = params->width, H = params->height, rb = params->row_block;
	if (!m->one_device) { const int rc = init_communicators(m); if (rc != RT_OK) return rc; }
	{ const int rc = prepare(m, W, H, rb, slot); if (rc != RT_OK) return rc; }
	const size_t strip_floats = (size_t) rt_strip_rows(H, rb, n) * W * 3, frame_floats = (size_t) H * W * 3;
	const int j = (int) (m->frames % STRIP_BUFFERS), which = (int) (m->frames % RT_LAUNCH_SETS);

	/* every device renders its interleaved row blocks, concurrently (the calls only enqueue); its collective stream
	 * takes over behind the render */
	int rc = RT_OK;
	int enqueued = 0;
	std::vector<hipEvent_t> reported((size_t) n, nullptr);      /* per device: behind the copy of its launch's control words */
	for (int i = 0; i < n && rc == RT_OK; i++) {
		rt_multi::per_device &d = m->dev[(size_t) i];
		rt_context *ctx = m->ctx[(size_t) i];
		hipStream_t rs = (hipStream_t) rt_stream(ctx, which);
		hipError_t e = rs ? hipSetDevice(m->devices[(size_t) i]) : hipErrorInvalidValue;
		/* the gather three frames back still reads d_strip[j] */
		if (e == hipSuccess && d.gathered_set[j]) e = hipStreamWaitEvent(rs, d.gathered[j], 0);
		if (e != hipSuccess) { rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: device %d: %s", m->devices[(size_t) i], hipGetErrorString(e)); break; }
		rt_render_params p = *params;
		p.rank = rt_strip_of_rank(i, n); p.world = n;     /* rotated by one: device 0, the root, renders the last strip -- never the longest (rt_hip.h) */
		if (m->profiling && m->marks[(size_t) i].size() >= RT_PROFILE_FRAMES_MAX) m->profiling = false;      /* (a host that never collects: the log stops growing) */
		if (m->profiling) {
			m->marks[(size_t) i].emplace_back();
			if (mark_now(&m->marks[(size_t) i].back().render_begins, rs) != hipSuccess) { rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: profiling event"); break; }
		}
		rc = rt_render_device(ctx, &p, d.d_strip[j], rs);
		if (rc != RT_OK) break;
		enqueued++;
		if (m->profiling && mark_now(&m->marks[(size_t) i].back().rendered, rs) != hipSuccess) { rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: profiling event"); break; }
		/* the collective stream takes over behind the render; first this launch's control words (rt_device.h RT_CTL_*: did a wave
		 * give up after rt_cancel(), did the launch account for every pixel), for rt_multi_frame_wait() -- not on the render
		 * stream: a copy between two kernels there costs the overlap of consecutive launches */
		e = hipStreamWaitEvent(d.gather_stream, (hipEvent_t) rt_context_launch_done(ctx), 0);
		if (e == hipSuccess) { rc = rt_context_read_control(ctx, &m->h_control[((size_t) slot * (size_t) m->n + (size_t) i) * RT_CTL_COPY_WORDS], d.gather_stream, &reported[(size_t) i], &f.expect[(size_t) i]); if (rc != RT_OK) break; }
		if (e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: device %d: %s", m->devices[(size_t) i], hipGetErrorString(e));
	}
	/* ONE gather of the finished strips to device 0, each rank's part on its own collective stream */
	if (rc == RT_OK) rc = gather_strips(m, j, strip_floats, reported);
	for (int i = 0; i < n && rc == RT_OK; i++) {
		rt_multi::per_device &d = m->dev[(size_t) i];
		hipError_t e = hipSetDevice(m->devices[(size_t) i]);
		/* (contexts sharing one device: all the copies ran on the root's stream) */
		if (e == hipSuccess) e = hipEventRecord(d.gathered[j], m->one_device ? m->dev[0].gather_stream : d.gather_stream);
		if (e == hipSuccess && m->profiling) e = mark_now(&m->marks[(size_t) i].back().gathered, m->one_device ? m->dev[0].gather_stream : d.gather_stream);
		if (e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: device %d: %s", m->devices[(size_t) i], hipGetErrorString(e));
		else d.gathered_set[j] = true;
	}
	/* device 0: rows back into frame order (behind the gather, on its stream), then the frame to the caller on the
	 * copy stream (main.c:467-479) */
	if (rc == RT_OK) {
		hipError_t e = hipSetDevice(m->devices[0]);
		if (e == hipSuccess)
			rc = rt_deinterleave_rotated_device(m->ctx[0], m->d_strips[j], f.d_frame, W, H, rb, n, n > 1 ? 1 : 0, m->dev[0].gather_stream);
		if (rc == RT_OK && e == hipSuccess) e = hipEventRecord(f.assembled, m->dev[0].gather_stream);
		if (rc == RT_OK && e == hipSuccess && m->profiling) e = mark_now(&m->marks[0].back().assembled, m->dev[0].gather_stream);
		if (rc == RT_OK && e == hipSuccess) e = hipStreamWaitEvent(m->copy_stream, f.assembled, 0);
		if (rc == RT_OK && e == hipSuccess && frame_out) e = hipMemcpyAsync(frame_out, f.d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, m->copy_stream);
		if (rc == RT_OK && e == hipSuccess) e = hipEventRecord(f.copied, m->copy_stream);      /* (device-resident frame: behind the de-interleave only) */
		if (rc == RT_OK && e == hipSuccess && m->profiling) e = mark_now(&m->marks[0].back().copied, m->copy_stream);
		if (rc == RT_OK && e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_multi_frame_submit: %s", hipGetErrorString(e));
	}
	if (rc != RT_OK) {
		/* some devices may have renders or a partial collective enqueued on buffers the handle owns: nothing of it is
		 * left running when the failure is reported (the error text survives: drain() sets none on success) */
		if (enqueued) drain(m);
		return rc;
	}
	m->frames++;
	f.busy = true; f.plain = false;
	if (d_frame) *d_frame = f.d_frame;
	if (hip_event) *hip_event = (void *) f.assembled;
	return RT_OK;
}

int rt_multi_frame_submit(rt_multi *m, const rt_render_params *params, int slot, Vector3 *frame_out)
{
	if (!frame_out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_submit: frame_out is NULL");
	return multi_frame_submit(m, params, slot, frame_out, nullptr, nullptr);
}

/* the frame stays on the first device: *d_frame = height * width Vector3 in frame order, *hip_event = behind the de-interleave */
int rt_multi_frame_submit_device(rt_multi *m, const rt_render_params *params, int slot, void **d_frame, void **hip_event)
{
	if (!d_frame) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_submit_device: d_frame is NULL");
	return multi_frame_submit(m, params, slot, nullptr, d_frame, hip_event);
}

/* ---- where a step of the N-GPU frame loop goes, per device (include/rt_hip.h: rt_multi_phases) ----
 * Several frames are in flight, so a frame's own phases say little about the step (frame f was rendered while frame f - 3 was
 * gathered).  Every instant between the device's first and last frame END -- the first device: frame in host memory; the others:
 * their part of the gather done -- is given to what the device was doing then, whichever frame it was for, by priority: a strip
 * render in progress (from the render stream reaching the launch to the end of its trace kernel: waiting for workgroup slots
 * behind the previous launches is in it) > a de-interleave > a copy to the host > a rendered strip waiting for its gather >
 * nothing (idle: no launch was there to run).  The shares are disjoint and sum to the window; divided by the intervals in it
 * they are ms per step.  (ray_tracing_amd.attribute_phases is the same in Python, for hosts that record their own events.) */
static void free_marks(rt_multi *m)
{
	for (auto &per_device : m->marks)
		for (auto &k : per_device)
			for (hipEvent_t e : { k.render_begins, k.rendered, k.gathered, k.assembled, k.copied })
				if (e) (void) hipEventDestroy(e);
	m->marks.assign((size_t) m->n, {});
}

int rt_multi_profile_enable(rt_multi *m, int on)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_profile_enable: NULL handle");
	drain(m);
	free_marks(m);
	m->profiling = on != 0;
	return RT_OK;
}

int rt_multi_profile_collect(rt_multi *m, rt_multi_phases *per_device, int capacity)
{
	if (!m || !per_device || capacity < m->n) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_profile_collect: need room for rt_multi_size() entries");
	drain(m);
	for (int i = 0; i < m->n; i++) {
		rt_multi_phases &out = per_device[i];
		memset(&out, 0, sizeof(out));
		const auto &marks = m->marks[(size_t) i];
		if (marks.size() < 2) continue;
		MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
		const hipEvent_t base = marks[0].render_begins;
		auto at = [&](hipEvent_t e, double *ms) -> hipError_t { float f = 0; const hipError_t rc = hipEventElapsedTime(&f, base, e); *ms = f; return rc; };
		struct span { int cls; double from, to; };       /* cls in priority order: 0 render, 1 de-interleave, 2 copy, 3 waiting for the gather */
		std::vector<span> spans;
		std::vector<double> cuts;
		double first_end = 0, last_end = 0;
		int frames = 0;
		for (const rt_multi::phase_marks &f : marks) {
			if (!f.render_begins || !f.rendered || !f.gathered) break;              /* (a submit that failed half way) */
			double b = 0, r = 0, g = 0, a = 0, c = 0;
			MULTI_HIP(at(f.render_begins, &b)); MULTI_HIP(at(f.rendered, &r)); MULTI_HIP(at(f.gathered, &g));
			a = c = g;                                                               /* a device other than the first ends with its gather */
			if (i == 0 && f.assembled && f.copied) { MULTI_HIP(at(f.assembled, &a)); MULTI_HIP(at(f.copied, &c)); }
			spans.push_back({ 0, b, r }); spans.push_back({ 3, r, g }); spans.push_back({ 1, g, a }); spans.push_back({ 2, a, c });
			if (frames == 0) first_end = c;
			if (c > last_end) last_end = c;
			frames++;
		}
		if (frames < 2 || !(last_end > first_end)) continue;
		cuts.push_back(first_end); cuts.push_back(last_end);
		for (const span &sp : spans)
			for (double t : { sp.from, sp.to })
				if (t > first_end && t < last_end) cuts.push_back(t);
		std::sort(cuts.begin(), cuts.end());
		double share[5] = { 0, 0, 0, 0, 0 };             /* render, de-interleave, copy, gather, idle */
		for (size_t k = 0; k + 1 < cuts.size(); k++) {
			const double lo = cuts[k], hi = cuts[k + 1];
			if (!(hi > lo)) continue;
			const double mid = 0.5 * (lo + hi);
			int cls = 4;
			for (const span &sp : spans)
				if (sp.from <= mid && mid < sp.to && sp.cls < cls) cls = sp.cls;
			share[cls] += hi - lo;
		}
		const double inv = 1.0 / (frames - 1);
		out.frames = frames - 1;
		out.step_ms = (last_end - first_end) * inv;
		out.render_ms = share[0] * inv; out.deinterleave_ms = share[1] * inv; out.copy_ms = share[2] * inv; out.gather_ms = share[3] * inv; out.idle_ms = share[4] * inv;
	}
	free_marks(m);
	return RT_OK;
}

/* What the group's communicator itself reports (bench.py puts it into its line): ranks = ncclCommCount of the first
 * communicator, devices[i] = ncclCommCuDevice of communicator i, *version = ncclGetVersion.  A group without a communicator
 * (one device without force_collective, or contexts sharing one device) reports ranks = 0. */
int rt_multi_collective_info(rt_multi *m, int *ranks, int devices[64], int *version)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_collective_info: NULL handle");
	if (ranks) *ranks = 0;
	if (version) *version = 0;
	if (m->one_device || (m->n == 1 && !m->force_collective)) return RT_OK;
	{ const int rc = init_communicators(m); if (rc != RT_OK) return rc; }
	Rccl &r = rccl();
	if (!r.comm_count || !r.comm_device || !r.get_version) return rt_fail(RT_ERR_DEVICE, "rt_multi_collective_info: this RCCL lacks ncclCommCount / ncclCommCuDevice");
	int count = 0, v = 0;
	if (r.comm_count(m->comms[0], &count) != ncclSuccess_ || r.get_version(&v) != ncclSuccess_) return rt_fail(RT_ERR_DEVICE, "ncclCommCount failed");
	for (int i = 0; i < m->n && devices; i++) {
		int dev = -1;
		if (r.comm_device(m->comms[(size_t) i], &dev) != ncclSuccess_) return rt_fail(RT_ERR_DEVICE, "ncclCommCuDevice failed");
		devices[i] = dev;
	}
	if (ranks) *ranks = count;
	if (version) *version = v;
	return RT_OK;
}

static int finish_slot(rt_multi *m, int slot)
{
	rt_multi::frame_slot &f = m->fq[slot];
	f.busy = false;
	if (f.plain) return rt_frame_wait(m->ctx[0], slot);
	MULTI_HIP(hipSetDevice(m->devices[0]));
	MULTI_HIP(hipEventSynchronize(f.copied));      /* behind the gather, hence behind every device's render and control-word copy ... */
	/* every device's launch is judged (rt_judge_launch): the frame is delivered if all of them accounted for every pixel; one
	 * incomplete strip is an error (its text names the launch), one cancelled strip a cancelled frame */
	int verdict = RT_OK;
	for (int i = 0; i < m->n; i++) {
		const int rc = rt_judge_launch(&m->h_control[((size_t) slot * (size_t) m->n + (size_t) i) * RT_CTL_COPY_WORDS], f.expect[(size_t) i], "rt_multi_frame_wait", nullptr);
		if (rc < 0) return rc;
		if (rc == RT_CANCELLED) verdict = RT_CANCELLED;
	}
	return verdict;
}

int rt_multi_frame_wait(rt_multi *m, int slot)
{
	if (!m || slot < 0 || slot >= RT_FRAME_SLOTS) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_wait: bad argument");
	if (!m->fq[slot].busy) return rt_fail(RT_ERR_STATE, "rt_multi_frame_wait: nothing was submitted into slot %d", slot);
	return finish_slot(m, slot);
}

int rt_multi_frame_poll(rt_multi *m, int slot)
{
	if (!m || slot < 0 || slot >= RT_FRAME_SLOTS) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_frame_poll: bad argument");
	rt_multi::frame_slot &f = m->fq[slot];
	if (!f.busy) return rt_fail(RT_ERR_STATE, "rt_multi_frame_poll: nothing was submitted into slot %d", slot);
	if (f.plain) {
		const int rc = rt_frame_poll(m->ctx[0], slot);
		if (rc != RT_PENDING) f.busy = false;
		return rc;
	}
	MULTI_HIP(hipSetDevice(m->devices[0]));
	const hipError_t e = hipEventQuery(f.copied);
	if (e == hipErrorNotReady) return RT_PENDING;
	if (e != hipSuccess) return rt_fail(RT_ERR_DEVICE, "rt_multi_frame_poll: %s", hipGetErrorString(e));
	return finish_slot(m, slot);
}

/* ---- the reference's interactive protocol on the device group (main.c:354-408, 450-482, 115-124) ------------------------
 * Every device accumulates the frame rows of ITS row blocks (blocks of 16 frame rows dealt round-robin: a multiple of every
 * scale of the ladder, so a low-resolution row never straddles two devices) through the whole ladder: a pass renders the
 * low-resolution rows that cover them and adds them in place, nothing is exchanged.  Only a displayed frame costs a
 * collective: every device resolves its rows, ONE gather brings them to the first device, de-interleave, copy to the host.
 * Pass p of every device uses the same seeds as the single-device ladder: frames are bit-identical to rt_progressive_*'s. */
int rt_multi_progressive_begin(rt_multi *m, int width, int height, int init_scale, int max_bounces, uint64_t seed)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_begin: NULL handle");
	for (int i = 0; i < m->n; i++) {
		const int rc = rt_progressive_begin_rank(m->ctx[(size_t) i], width, height, init_scale, max_bounces, seed, i, m->n);
		if (rc != RT_OK) return rc;
	}
	m->prog_w = width; m->prog_h = height;
	return RT_OK;
}

int rt_multi_progressive_pass(rt_multi *m, float *weight_out)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_pass: NULL handle");
	for (int i = 0; i < m->n; i++) {       /* the calls only enqueue: the devices render side by side */
		const int rc = rt_progressive_pass(m->ctx[(size_t) i], weight_out);
		if (rc != RT_OK) return rc;
	}
	return RT_OK;
}

int rt_multi_progressive_passes(rt_multi *m, int count)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_passes: NULL handle");
	for (int i = 0; i < m->n; i++) {       /* every device goes through the same ladder: the same launches on each */
		const int rc = rt_progressive_passes(m->ctx[(size_t) i], count);
		if (rc != RT_OK) return rc;
	}
	return RT_OK;
}

int rt_multi_progressive_invalidate(rt_multi *m)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_invalidate: NULL handle");
	for (int i = 0; i < m->n; i++) {
		const int rc = rt_progressive_invalidate(m->ctx[(size_t) i]);
		if (rc != RT_OK) return rc;
	}
	return RT_OK;
}

/* The ladder's sum of published weights, as EVERY device holds it.  A pass whose launch was incomplete on one device is not
 * published there (and that device says so: RT_ERR_DEVICE); all devices publish the same passes or the gathered frame would mix
 * rows of different sample counts -- so every context is asked, the first error is returned, and counts that differ are an
 * error of their own. */
static int group_count(rt_multi *m, float *count, const char *who)
{
	float first = 0;
	for (int i = 0; i < m->n; i++) {
		float c = 0;
		const int rc = rt_progressive_count(m->ctx[(size_t) i], &c);
		if (rc != RT_OK) return rc;
		if (i == 0) first = c;
		else if (c != first)
			return rt_fail(RT_ERR_DEVICE, "%s: device %d of the group has published a weight sum of %.9g, the first device %.9g: their rows would not be of one frame", who, i, (double) c, (double) first);
	}
	*count = first;
	return RT_OK;
}

int rt_multi_progressive_state(rt_multi *m, int *next_scale, float *count, uint32_t *generation, int *passes)
{
	if (!m) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_state: NULL handle");
	if (count) { const int rc = group_count(m, count, "rt_multi_progressive_state"); if (rc != RT_OK) return rc; }
	return rt_progressive_state(m->ctx[0], next_scale, nullptr, generation, passes);    /* (host-side state: the same on every device) */
}

int rt_multi_progressive_resolve(rt_multi *m, Vector3 *frame_out)
{
	if (!m || !frame_out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_progressive_resolve: NULL argument");
	if (m->prog_w == 0) return rt_fail(RT_ERR_STATE, "rt_multi_progressive_resolve: call rt_multi_progressive_begin first");
	if (m->n == 1 && !m->force_collective) return rt_progressive_resolve(m->ctx[0], frame_out);
	const int n = m->n, W = m->prog_w, H = m->prog_h, rb = RT_PROGRESSIVE_ROW_BLOCK;
	/* the gather lands in the frame queue's first strips buffer: not while frames are in flight */
	for (int s = 0; s < RT_FRAME_SLOTS; s++)
		if (m->fq[s].busy) return rt_fail(RT_ERR_STATE, "rt_multi_progressive_resolve: slot %d holds a frame that has not been waited for", s);
	const int slot = 0;
	if (!m->one_device) { const int rc = init_communicators(m); if (rc != RT_OK) return rc; }
	{ const int rc = prepare(m, W, H, rb, slot); if (rc != RT_OK) return rc; }
	rt_multi::frame_slot &f = m->fq[slot];
	{	/* the count first (update_frame() waits for it, main.c:461-464): with nothing published there is nothing to divide by,
		 * and frame_out is left as it is */
		float count = 0;
		const int crc = group_count(m, &count, "rt_multi_progressive_resolve");
		if (crc != RT_OK) return crc;
		if ((double) count < 0.0001) return rt_fail(RT_ERR_STATE, "rt_multi_progressive_resolve: nothing accumulated yet (every pass so far was cancelled)");
	}
	/* what one rank sends: its rows, padded like a strip -- one rank alone holds exactly the frame */
	const size_t strip_floats = (size_t) (n == 1 ? H : rt_strip_rows(H, rb, n)) * W * 3, frame_floats = (size_t) H * W * 3;
	std::vector<void *> d_rows((size_t) n, nullptr);
	int rc = RT_OK;
	for (int i = 0; i < n && rc == RT_OK; i++)          /* frame = accum * (1 / count) (main.c:467-477), every device its own rows */
		rc = rt_progressive_resolve_device(m->ctx[(size_t) i], &d_rows[(size_t) i]);
	if (rc == RT_OK && m->one_device) {                 /* (testing aid: the copies the gather amounts to on one device, then the root waits for them) */
		for (int i = 0; i < n && rc == RT_OK; i++) {
			hipStream_t cs = (hipStream_t) rt_context_stream(m->ctx[(size_t) i]);
			hipError_t e = hipMemcpyAsync(m->d_strips[0] + (size_t) i * strip_floats, d_rows[(size_t) i], strip_floats * sizeof(float), hipMemcpyDeviceToDevice, cs);
			if (e == hipSuccess) e = hipEventRecord(m->dev[(size_t) i].gathered[0], cs);
			if (e == hipSuccess && i > 0) e = hipStreamWaitEvent((hipStream_t) rt_context_stream(m->ctx[0]), m->dev[(size_t) i].gathered[0], 0);
			if (e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_multi_progressive_resolve: %s", hipGetErrorString(e));
		}
	}
	Rccl &r = rccl();
	if (rc == RT_OK && !m->one_device) {                /* ONE gather per displayed frame, behind the resolves on the contexts' streams */
		ncclResult_t nrc = r.group_start();
		for (int i = 0; i < n && nrc == ncclSuccess_; i++)
			nrc = r.gather(d_rows[(size_t) i], i == 0 ? m->d_strips[0] : nullptr, strip_floats, ncclFloat_, 0,
			               m->comms[(size_t) i], (hipStream_t) rt_context_stream(m->ctx[(size_t) i]));
		{ const ncclResult_t end = r.group_end(); if (nrc == ncclSuccess_) nrc = end; }
		if (nrc != ncclSuccess_) rc = rt_fail(RT_ERR_DEVICE, "ncclGather: %s", r.error_string(nrc));
	}
	if (rc == RT_OK) rc = rt_deinterleave_device(m->ctx[0], m->d_strips[0], f.d_frame, W, H, rb, n, nullptr);
	if (rc == RT_OK) {
		hipError_t e = hipSetDevice(m->devices[0]);
		if (e == hipSuccess) e = hipMemcpyAsync(frame_out, f.d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t) rt_context_stream(m->ctx[0]));
		if (e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_multi_progressive_resolve: %s", hipGetErrorString(e));
	}
	if (rc != RT_OK) { drain(m); return rc; }
	for (int i = 0; i < n; i++) {
		const int src = rt_synchronize(m->ctx[(size_t) i]);
		if (src != RT_OK) return src;
	}
	return RT_OK;
}

/* one frame, start to finish: submit + wait (what rt_render() is for one device) */
int rt_multi_render(rt_multi *m, const rt_render_params *params, Vector3 *frame_out)
{
	if (!m || !params || !frame_out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_render: NULL argument");
	if (m->n == 1 && !m->force_collective) {
		rt_render_params p = *params;
		p.rank = 0; p.world = 1;
		return rt_render(m->ctx[0], &p, frame_out);
	}
	int slot = -1;
	for (int s = 0; s < RT_FRAME_SLOTS; s++) if (!m->fq[s].busy) { slot = s; break; }
	if (slot < 0) return rt_fail(RT_ERR_STATE, "rt_multi_render: every frame slot holds a frame that has not been waited for");
	const int rc = rt_multi_frame_submit(m, params, slot, frame_out);
	if (rc != RT_OK) return rc;
	return rt_multi_frame_wait(m, slot);
}

} /* extern "C" */
