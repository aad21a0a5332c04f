/*
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Body of the headless harness around the UNMODIFIED reference sources.  This file is
 * appended (by oracle/Makefile) to a translation unit that has already textually
 * included /root/reference/src/main.c (with `main` renamed), so the reference's globals
 * (`scene`, `skybox`, `num_columns`, `init_scale`, `frame_w`, `frame_h`) and functions
 * (`pixel`, `render_column`, `trace_ray`, `sample_cubemap`, ...) are in scope.  Nothing of the
 * reference is copied here: the harness only *calls* it.
 *
 * It drives the reference exactly the way worker() does (main.c:363-396) and the way
 * update_frame() resolves (main.c:467-477), minus threads / window.
 *
 * The built library lands in oracle/_ref/ (git-ignored).  It exists to
 *   (1) pin the C restatement in oracle/rt_oracle.c (tests/, fixtures in tests/golden/),
 *   (2) serve as the "reference" CPU baseline timed by bench.py.
 */
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define REF_API __attribute__((visibility("default")))

/* lives in the TU that includes utils.c (the RNG state there is `static _Thread_local`) */
void     ref_set_rng(uint64_t s);
uint64_t ref_get_rng(void);
/* lives in the TU that includes camera.c (camera state there is `static`) */
void     ref_set_camera(const float pos[3], const float front[3], const float up[3], float fov_value);

#ifdef REF_PATCHED_BOUNCES
extern int oracle_bounce_limit;
REF_API void ref_set_bounce_limit(int n) { oracle_bounce_limit = n; }
REF_API int  ref_has_bounce_limit(void)  { return 1; }
#else
REF_API void ref_set_bounce_limit(int n) { (void) n; }
REF_API int  ref_has_bounce_limit(void)  { return 0; }
#endif

/* The path-seed used by the `counter` RNG mode.  This is OUR definition (the reference has no
 * seed, utils.c:60); it must be kept identical in oracle/rt_oracle.c and in the HIP kernels. */
static uint64_t harness_path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index)
{
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t) sample_index << 32) | (uint64_t) pixel_index);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

REF_API int ref_load_scene(const char *file)
{
	return parse_scene_file((char*) file, &scene) ? 0 : -1;
}

REF_API int ref_parse_scene_into(const char *file, void *dst, size_t dst_size)
{
	if (dst_size < sizeof(Scene)) return -2;
	memset(dst, 0, sizeof(Scene));
	return parse_scene_file((char*) file, (Scene*) dst) ? 0 : -1;
}

REF_API void ref_set_scene(const void *src) { memcpy(&scene, src, sizeof(Scene)); }
REF_API const void *ref_scene_ptr(void)     { return &scene; }
REF_API size_t ref_sizeof_scene(void)       { return sizeof(Scene); }
REF_API size_t ref_sizeof_object(void)      { return sizeof(Object); }
REF_API int ref_num_objects(void)           { return scene.num_objects; }

static int skybox_owned = 0;

REF_API void ref_load_skybox(const char *dir)
{
	static char paths[6][1024];
	static const char *names[6];
	names[CF_RIGHT] = "right.jpg"; names[CF_LEFT]  = "left.jpg";  names[CF_TOP]  = "top.jpg";
	names[CF_BOTTOM]= "bottom.jpg"; names[CF_FRONT] = "front.jpg"; names[CF_BACK] = "back.jpg";
	const char *files[6];
	for (int i = 0; i < 6; i++) {
		snprintf(paths[i], sizeof(paths[i]), "%s/%s", dir, names[i]);
		files[i] = paths[i];
	}
	if (skybox_owned) free_cubemap(&skybox);
	load_cubemap(&skybox, files);  /* aborts on failure, as the reference does */
	skybox_owned = 1;
}

/* Synthetic skybox: point the reference's Cubemap at caller-owned faces (order = CubeFace enum). */
REF_API void ref_set_skybox_raw(uint8_t *const faces[6], int w, int h, int chan)
{
	if (skybox_owned) { free_cubemap(&skybox); skybox_owned = 0; }
	for (int i = 0; i < 6; i++) skybox.data[i] = faces[i];
	skybox.w = w; skybox.h = h; skybox.chan = chan;
}

REF_API const uint8_t *ref_skybox_face(int i) { return skybox.data[i]; }
REF_API void ref_skybox_dims(int *w, int *h, int *chan) { *w = skybox.w; *h = skybox.h; *chan = skybox.chan; }

REF_API float ref_random_float(void) { return random_float(); }
REF_API void  ref_random_direction(float out[3])
{
	Vector3 v = random_direction();
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

REF_API void ref_camera_ray(float px, float py, float aspect, float out[6])
{
	Ray r = ray_through_screen_at(px, py, aspect);
	out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
	out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
}

/* out = {distance, point[3], normal[3]}, returns object index (or -1) */
REF_API int ref_trace_ray(const float o[3], const float d[3], float out[7])
{
	Ray r = { {o[0], o[1], o[2]}, {d[0], d[1], d[2]} };
	HitInfo h = trace_ray(r, &scene);
	out[0] = h.distance;
	out[1] = h.point.x;  out[2] = h.point.y;  out[3] = h.point.z;
	out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z;
	return h.object;
}

REF_API void ref_sample_cubemap(const float d[3], float out[3])
{
	Vector3 c = sample_cubemap(&skybox, (Vector3) {d[0], d[1], d[2]});
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

REF_API void ref_normalize(const float d[3], float out[3])
{
	Vector3 c = normalize((Vector3) {d[0], d[1], d[2]});
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

REF_API void ref_pixel(float u, float v, float aspect, float out[3])
{
	Vector3 c = pixel(u, v, aspect);
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/*
 * `stream` mode: what one reference worker does with num_columns=1, init_scale=1
 * (main.c:363,377,387-396) for `passes` passes, then update_frame()'s resolve (main.c:476).
 * The RNG is whatever the calling thread's state is (the reference never seeds it: 0).
 */
REF_API void ref_render_stream_scaled(int W, int H, int passes, int first_scale, float *frame_out, float *accum_out);

REF_API void ref_render_stream(int W, int H, int passes, float *frame_out, float *accum_out)
{
	ref_render_stream_scaled(W, H, passes, 1, frame_out, accum_out);
}

/* worker() with num_columns = 1 and the scale ladder: init_scale, then halved after every publish
 * (main.c:354,377,387-403) */
REF_API void ref_render_stream_scaled(int W, int H, int passes, int first_scale, float *frame_out, float *accum_out)
{
	num_columns = 1; init_scale = first_scale; frame_w = W; frame_h = H;
	size_t n = (size_t) W * H;
	Vector3 *col = calloc(n, sizeof(Vector3));
	Vector3 *acc = calloc(n, sizeof(Vector3));
	float count = 0;
	int scale = first_scale;
	for (int p = 0; p < passes; p++) {
		float w = render_column(col, scale, W, 0, W, H, atomic_load(&accum_generation));
		for (size_t k = 0; k < n; k++)
			acc[k] = combine(acc[k], col[k], 1, 1.0f / (scale * scale));
		count += w;
		if (scale > 1) scale >>= 1;
	}
	if (accum_out) memcpy(accum_out, acc, sizeof(Vector3) * n);
	if (frame_out)
		for (size_t k = 0; k < n; k++) {
			Vector3 f = scalev(acc[k], 1.0f / count);
			frame_out[3*k+0] = f.x; frame_out[3*k+1] = f.y; frame_out[3*k+2] = f.z;
		}
	free(col); free(acc);
}

/*
 * `counter` mode: the reference's pixel() (main.c:131) called once per (pixel, sample) with
 * the thread-local RNG state preset to path_seed(seed, pixel_index, sample) -- rows [row0,row1).
 * Pixel coordinates as render_column() computes them at scale 1 (main.c:293-296); sample sum and
 * resolve as worker()/update_frame() (main.c:394,476).
 */
REF_API void ref_render_counter_rows(int W, int H, int spp, uint64_t seed, int row0, int row1, float *frame_out)
{
	float aspect = (float) W / H;
	float inv = 1.0f / (float) spp;
	for (int j = row0; j < row1; j++)
		for (int i = 0; i < W; i++) {
			float u = (float) i / (W - 1);
			float v = (float) j / (H - 1);
			u = 1 - u;
			v = 1 - v;
			uint32_t p = (uint32_t) (j * W + i);
			Vector3 acc = {0, 0, 0};
			for (int s = 0; s < spp; s++) {
				ref_set_rng(harness_path_seed(seed, p, (uint32_t) s));
				Vector3 c = pixel(u, v, aspect);
				acc = combine(acc, c, 1, 1.0f);
			}
			Vector3 f = scalev(acc, inv);
			size_t k = (size_t) j * W + i;
			frame_out[3*k+0] = f.x; frame_out[3*k+1] = f.y; frame_out[3*k+2] = f.z;
		}
}

REF_API void ref_render_counter(int W, int H, int spp, uint64_t seed, float *frame_out)
{
	ref_render_counter_rows(W, H, spp, seed, 0, H, frame_out);
}

REF_API uint64_t ref_path_seed(uint64_t seed, uint32_t p, uint32_t s) { return harness_path_seed(seed, p, s); }

/*
 * CPU baseline: the reference's own parallelisation -- one thread per image column, each thread
 * calling render_column() on its column (main.c:333,363,377), `passes` passes each, RNG from 0 in
 * every thread as in the reference.  Pixels are discarded; only the wall time matters.
 */
typedef struct { int column_i, column_w, W, H, passes; } ColumnJob;

static void *column_thread(void *arg)
{
	ColumnJob *job = arg;
	Vector3 *col = malloc(sizeof(Vector3) * (size_t) job->column_w * job->H);
	for (int p = 0; p < job->passes; p++)
		render_column(col, 1, job->column_w, job->column_i, job->W, job->H, atomic_load(&accum_generation));
	free(col);
	return NULL;
}

REF_API void ref_time_columns(int W, int H, int passes, int threads)
{
	if (threads < 1) threads = 1;
	if (threads > MAX_COLUMNS) threads = MAX_COLUMNS;
	num_columns = threads; init_scale = 1; frame_w = W; frame_h = H;
	pthread_t tid[MAX_COLUMNS];
	ColumnJob jobs[MAX_COLUMNS];
	int created[MAX_COLUMNS];
	for (int t = 0; t < threads; t++) {
		jobs[t] = (ColumnJob) { t, W / threads, W, H, passes };
		created[t] = pthread_create(&tid[t], NULL, column_thread, &jobs[t]) == 0;
		if (!created[t]) column_thread(&jobs[t]);      /* (a box's thread limit: the column is rendered by the caller) */
	}
	for (int t = 0; t < threads; t++)
		if (created[t]) pthread_join(tid[t], NULL);
}
