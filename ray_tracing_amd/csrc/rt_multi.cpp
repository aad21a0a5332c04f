/*
 * rt_multi.cpp -- one frame on several GPUs of one node from ONE host process, behind the C ABI
 * (include/rt_hip.h, rt_multi_*).
 *
 * The reference fans its work out inside the host binary: start_workers() creates one pthread per image
 * column and every worker adds its passes into the shared accumulation buffer (main.c:695-718, 324-414).
 * Here the fan-out is over GPUs: rt_multi_render() gives device i the row blocks b with b % n == i (the same
 * interleaved partition rt_render_device() implements for one rank), every device renders its strip
 * concurrently on its own stream, ONE grouped ncclGather over xGMI brings the strips to device 0, a
 * de-interleave kernel there puts the rows in frame order and the frame is copied to the caller's host
 * buffer -- what update_frame() hands to move_frame_to_the_gpu() (main.c:467-479).
 *
 * RCCL is loaded with dlopen() on the first multi-device create (as rt_jit.cpp does for hiprtc), so the library
 * has no link-time dependency on it and single-GPU hosts never touch it.  There is no fallback: if RCCL cannot
 * be loaded or initialised, rt_multi_create() with n > 1 fails with RT_ERR_DEVICE.
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_internal.h"

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclCommInitAll)    comm_init_all = nullptr;
	decltype(&ncclCommDestroy)    comm_destroy = nullptr;
	decltype(&ncclGather)         gather = nullptr;
	decltype(&ncclGroupStart)     group_start = nullptr;
	decltype(&ncclGroupEnd)       group_end = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
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
#undef SYM
	r.ok = r.comm_init_all && r.comm_destroy && r.gather && r.group_start && r.group_end && r.error_string;
	return r;
}

} // namespace

struct rt_multi {
	int n = 0;
	bool force_collective = false;              /* rt_tuning.force_collective: gather + de-interleave even with one device */
	std::vector<int>          devices;
	std::vector<rt_context *> ctx;
	std::vector<ncclComm_t>   comms;            /* n > 1 only */
	std::vector<float *>      d_strip;          /* one strip per device */
	size_t strip_floats = 0;                    /* capacity of each */
	float *d_strips = nullptr;                  /* device 0: n strips back to back (the gather's destination) */
	float *d_frame = nullptr;                   /* device 0: the frame in row order */
	size_t strips_floats = 0, frame_floats = 0;
};

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
	if (rc != ncclSuccess) {
		m->comms.clear();
		return rt_fail(RT_ERR_DEVICE, "ncclCommInitAll: %s", r.error_string(rc));
	}
	return RT_OK;
}

extern "C" {

int rt_multi_create(rt_multi **out, const int *device_ids, int n)
{
	if (!out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: out is NULL");
	*out = nullptr;
	if (!device_ids || n < 1 || n > 64) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: need 1..64 device ids");
	for (int i = 0; i < n; i++)
		for (int k = 0; k < i; k++)
			if (device_ids[i] == device_ids[k])
				return rt_fail(RT_ERR_ARGUMENT, "rt_multi_create: device %d listed twice", device_ids[i]);
	rt_multi *m = new (std::nothrow) rt_multi();
	if (!m) return rt_fail(RT_ERR_MEMORY, "rt_multi_create: out of host memory");
	m->n = n;
	m->devices.assign(device_ids, device_ids + n);
	m->ctx.assign((size_t) n, nullptr);
	m->d_strip.assign((size_t) n, nullptr);
	for (int i = 0; i < n; i++) {
		const int rc = rt_create(&m->ctx[(size_t) i], device_ids[i]);
		if (rc != RT_OK) { rt_multi_destroy(m); return rc; }          /* rt_last_error() holds rt_create's text */
	}
	if (n > 1) {
		const int rc = init_communicators(m);
		if (rc != RT_OK) { rt_multi_destroy(m); return rc; }
	}
	*out = m;
	return RT_OK;
}

void rt_multi_destroy(rt_multi *m)
{
	if (!m) return;
	for (int i = 0; i < m->n; i++) {
		if (!m->ctx[(size_t) i]) continue;
		(void) rt_synchronize(m->ctx[(size_t) i]);
		(void) hipSetDevice(m->devices[(size_t) i]);
		(void) hipFree(m->d_strip[(size_t) i]);
		if (i == 0) { (void) hipFree(m->d_strips); (void) hipFree(m->d_frame); }
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
int rt_multi_set_tuning(rt_multi *m, const rt_tuning *tuning)
{
	if (m && tuning) m->force_collective = tuning->force_collective != 0;
	FOR_ALL(rt_set_tuning(ctx, tuning));
}
int rt_multi_compile_scene(rt_multi *m)                       { FOR_ALL(rt_compile_scene(ctx)); }

int rt_multi_render(rt_multi *m, const rt_render_params *params, Vector3 *frame_out)
{
	if (!m || !params || !frame_out) return rt_fail(RT_ERR_ARGUMENT, "rt_multi_render: NULL argument");
	if (m->n == 1 && !m->force_collective) {    /* one device: the strip is the frame */
		rt_render_params p = *params;
		p.rank = 0; p.world = 1;
		return rt_render(m->ctx[0], &p, frame_out);
	}
	if (params->width < 2 || params->height < 2 || params->row_block < 1)
		return rt_fail(RT_ERR_ARGUMENT, "rt_multi_render: bad frame %dx%d / row_block %d", params->width, params->height, params->row_block);
	const int n = m->n, W = params->width, H = params->height, rb = params->row_block;
	const int rows = rt_strip_rows(H, rb, n);
	const size_t strip_floats = (size_t) rows * W * 3, frame_floats = (size_t) H * W * 3;

	/* buffers, grown on demand */
	if (strip_floats > m->strip_floats) {
		for (int i = 0; i < n; i++) {
			MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
			{ const int rc = rt_synchronize(m->ctx[(size_t) i]); if (rc != RT_OK) return rc; }
			(void) hipFree(m->d_strip[(size_t) i]); m->d_strip[(size_t) i] = nullptr;
		}
		m->strip_floats = 0;
		for (int i = 0; i < n; i++) {
			MULTI_HIP(hipSetDevice(m->devices[(size_t) i]));
			MULTI_HIP(hipMalloc((void **) &m->d_strip[(size_t) i], strip_floats * sizeof(float)));
		}
		m->strip_floats = strip_floats;
	}
	MULTI_HIP(hipSetDevice(m->devices[0]));
	if (strip_floats * (size_t) n > m->strips_floats) {
		(void) hipFree(m->d_strips); m->d_strips = nullptr; m->strips_floats = 0;
		MULTI_HIP(hipMalloc((void **) &m->d_strips, strip_floats * (size_t) n * sizeof(float)));
		m->strips_floats = strip_floats * (size_t) n;
	}
	if (frame_floats > m->frame_floats) {
		(void) hipFree(m->d_frame); m->d_frame = nullptr; m->frame_floats = 0;
		MULTI_HIP(hipMalloc((void **) &m->d_frame, frame_floats * sizeof(float)));
		m->frame_floats = frame_floats;
	}

	{ const int rc = init_communicators(m); if (rc != RT_OK) return rc; }
	/* every device renders its interleaved row blocks, concurrently (the calls only enqueue) */
	for (int i = 0; i < n; i++) {
		rt_render_params p = *params;
		p.rank = i; p.world = n;
		const int rc = rt_render_device(m->ctx[(size_t) i], &p, m->d_strip[(size_t) i], nullptr);
		if (rc != RT_OK) return rc;
	}
	/* ONE gather of the finished strips to device 0, each rank's part on its own stream behind its render */
	Rccl &r = rccl();
	ncclResult_t nrc = r.group_start();
	for (int i = 0; i < n && nrc == ncclSuccess; i++)
		nrc = r.gather(m->d_strip[(size_t) i], i == 0 ? m->d_strips : nullptr, strip_floats, ncclFloat, 0,
		               m->comms[(size_t) i], (hipStream_t) rt_context_stream(m->ctx[(size_t) i]));
	{ const ncclResult_t end = r.group_end(); if (nrc == ncclSuccess) nrc = end; }
	if (nrc != ncclSuccess) return rt_fail(RT_ERR_DEVICE, "ncclGather: %s", r.error_string(nrc));
	/* device 0: rows back into frame order, then the frame to the caller (main.c:467-479) */
	{
		const int rc = rt_deinterleave_device(m->ctx[0], m->d_strips, m->d_frame, W, H, rb, n, nullptr);
		if (rc != RT_OK) return rc;
	}
	MULTI_HIP(hipSetDevice(m->devices[0]));
	MULTI_HIP(hipMemcpyAsync(frame_out, m->d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost,
	                         (hipStream_t) rt_context_stream(m->ctx[0])));
	for (int i = 0; i < n; i++) {
		const int rc = rt_synchronize(m->ctx[(size_t) i]);
		if (rc != RT_OK) return rc;
	}
	return RT_OK;
}

} /* extern "C" */
