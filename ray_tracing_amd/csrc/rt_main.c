/*
 * rt_main.c -- headless host for librt_hip.so, shaped like the reference's main() (main.c:484-518):
 * parse flags, load the scene and the six skybox faces, render, hand the frame to the presenter
 * hook.  It keeps the reference's three flags (main.c:585-634) and adds the parameters the
 * reference hard-codes:
 *
 *   --scene <file>       required, as in the reference
 *   --threads <n>        accepted for command-line compatibility; the GPU replaces the worker threads
 *   --init-scale <n>     with --interactive: the scale the ladder starts at after an invalidation (reference: 8, main.c:611-621);
 *                        otherwise accepted and ignored (the image accumulates full-resolution passes)
 *   --interactive <n>    the reference's interactive protocol instead of one frame (main.c:354-408, 450-482): the scale ladder
 *                        from --init-scale, n passes in all, a frame resolved (update_frame()) every --present-every passes
 *                        (default 16: the passes between two frames go to the GPU in one call, rt_multi_progressive_passes);
 *                        the last resolved frame goes to --out
 *   --width/--height     frame size         (reference: window size, 1280x960, main.c:512)
 *   --spp <n>            passes accumulated (reference: until the camera moves)
 *   --bounces <n>        path length limit  (reference: 10, main.c:156)
 *   --seed <n>           counter-mode seed
 *   --skybox <dir>       directory with {right,left,top,bottom,front,back}.jpg (default assets/skybox)
 *   --device <n>         GPU index
 *   --gpus <n>           render on GPUs 0..n-1 of this node at once (row blocks interleaved over them, one RCCL
 *                        gather of the strips): the counterpart of the reference's --threads
 *   --out <file>         where the presenter hook writes the frame: .png or .ppm (default frame.ppm)
 *   --frames <k>         render k frames (seeds seed, seed+1, ...) with two in flight -- rt_multi_frame_submit /
 *                        rt_multi_frame_wait: the copy of a frame to the host runs beside the render of the next, as the
 *                        reference's workers keep rendering while its main thread presents (main.c:354-408 vs 450-482) --
 *                        and print the rate; the last frame goes to the presenter hook
 *   --compile            specialise the trace kernel for the scene first (rt_compile_scene; same pixels); the shipped scenes'
 *                        kernels are embedded in the library, any other scene is compiled by hiprtc
 *   --warmup <n>         with --frames: untimed frames before the timed ones (default 3, as bench.py)
 *   --force-collective   testing aid for 1-GPU boxes: one device runs the N-GPU path all the same (RCCL gather on a one-rank
 *                        communicator, de-interleave, three strip buffers)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/rt_hip.h"
#include "../../include/rt_hip_testing.h"     /* --force-collective only */

static const char *out_file = "frame.ppm";

static void write_frame(int w, int h, Vector3 *data, void *user)
{
	(void) user;
	size_t n = strlen(out_file);
	int png = n > 4 && strcmp(out_file + n - 4, ".png") == 0;
	int rc = png ? rt_write_png(out_file, w, h, data) : rt_write_ppm(out_file, w, h, data);
	if (rc != RT_OK) fprintf(stderr, "Could not write %s (%d)\n", out_file, rc);
	else             fprintf(stderr, "Wrote %s (%dx%d)\n", out_file, w, h);
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

int main(int argc, char **argv)
{
	const char *scene_file = NULL, *sky_dir = "assets/skybox";
	int width = 1280, height = 960, spp = 16, bounces = 10, device = 0, gpus = 0, frames = 0, compile = 0, force_collective = 0, warmup = 3;
	int init_scale = 8, interactive = 0, present_every = 16;
	double compile_s = 0;
	unsigned long long seed = 0;

	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		const char *v = i + 1 < argc ? argv[i + 1] : NULL;
#define NEED_VALUE() do { if (!v) { fprintf(stderr, "Error: Missing value after %s\n", a); return -1; } i++; } while (0)
		if      (!strcmp(a, "--scene"))      { NEED_VALUE(); scene_file = v; }
		else if (!strcmp(a, "--threads"))    { NEED_VALUE(); }
		else if (!strcmp(a, "--init-scale")) { NEED_VALUE(); init_scale = atoi(v); }
		else if (!strcmp(a, "--interactive")) { NEED_VALUE(); interactive = atoi(v); }
		else if (!strcmp(a, "--present-every")) { NEED_VALUE(); present_every = atoi(v); }
		else if (!strcmp(a, "--width"))      { NEED_VALUE(); width = atoi(v); }
		else if (!strcmp(a, "--height"))     { NEED_VALUE(); height = atoi(v); }
		else if (!strcmp(a, "--spp"))        { NEED_VALUE(); spp = atoi(v); }
		else if (!strcmp(a, "--bounces"))    { NEED_VALUE(); bounces = atoi(v); }
		else if (!strcmp(a, "--seed"))       { NEED_VALUE(); seed = strtoull(v, NULL, 0); }
		else if (!strcmp(a, "--skybox"))     { NEED_VALUE(); sky_dir = v; }
		else if (!strcmp(a, "--device"))     { NEED_VALUE(); device = atoi(v); }
		else if (!strcmp(a, "--gpus"))       { NEED_VALUE(); gpus = atoi(v); }
		else if (!strcmp(a, "--out"))        { NEED_VALUE(); out_file = v; }
		else if (!strcmp(a, "--frames"))     { NEED_VALUE(); frames = atoi(v); }
		else if (!strcmp(a, "--warmup"))     { NEED_VALUE(); warmup = atoi(v); }
		else if (!strcmp(a, "--compile"))    { compile = 1; }
		else if (!strcmp(a, "--force-collective")) { force_collective = 1; }
		else fprintf(stderr, "Warning: Ignoring option %s\n", a);
#undef NEED_VALUE
	}
	if (!scene_file) {
		fprintf(stderr, "Error: Missing --scene <file>\n");
		return -1;
	}

	static Scene scene;
	if (rt_parse_scene_file(scene_file, &scene) != RT_OK) {
		fprintf(stderr, "Couldn't parse scene\n");
		return -1;
	}
	fprintf(stderr, "Scene parsed (%d objects)\n", scene.num_objects);

	char paths[6][1024];
	const char *files[6];
	static const char *names[6] = { "front.jpg", "back.jpg", "left.jpg", "right.jpg", "top.jpg", "bottom.jpg" };
	for (int f = 0; f < 6; f++) {
		snprintf(paths[f], sizeof(paths[f]), "%s/%s", sky_dir, names[f]);
		files[f] = paths[f];
	}
	Cubemap skybox;
	if (rt_load_cubemap(&skybox, files) != RT_OK)
		return -1;
	fprintf(stderr, "Cubemap loaded (%dx%dx%d)\n", skybox.w, skybox.h, skybox.chan);

	/* one device: a context; several: a group of contexts with their RCCL communicators (rt_multi_*) */
	int ids[64];
	if (gpus > 64) gpus = 64;
	for (int i = 0; i < gpus; i++) ids[i] = i;
	if (gpus < 1) { gpus = 1; ids[0] = device; }
	rt_multi *group = NULL;
	if (rt_abi_version() != RT_ABI_VERSION) {
		fprintf(stderr, "Error: librt_hip.so has ABI version %d, this host was compiled against %d\n", rt_abi_version(), RT_ABI_VERSION);
		return -1;
	}
	if (rt_multi_create(&group, ids, gpus) != RT_OK || rt_multi_set_scene(group, &scene) != RT_OK || rt_multi_set_skybox(group, &skybox) != RT_OK) {
		fprintf(stderr, "Error: %s\n", rt_last_error());
		return -1;
	}
	rt_camera cam;
	rt_camera_default(&cam);
	rt_multi_set_camera(group, &cam);

	if (force_collective) {
		rt_test_knobs k;
		rt_default_test_knobs(&k);
		k.force_collective = 1;
		rt_multi_set_test_knobs(group, &k);
	}
	if (compile) {
		const double tc = now_s();
		if (rt_multi_compile_scene(group) != RT_OK)
			fprintf(stderr, "Warning: %s (continuing with the generic kernel)\n", rt_last_error());
		else {
			compile_s = now_s() - tc;
			fprintf(stderr, "Scene kernel: %s (%.3f s)\n", rt_compiled_scene_info(rt_multi_context(group, 0)), compile_s);
		}
	}

	rt_render_params p;
	rt_default_params(&p, width, height, spp, bounces);
	p.seed = seed;
	rt_set_frame_sink(write_frame, NULL);

	if (frames > 0) {
		/* the presenter's loop with two frames in flight: submit k+1, wait for k, hand k on */
		Vector3 *buf[2] = { NULL, NULL };
		for (int s = 0; s < 2; s++)
			if (rt_host_alloc((void **) &buf[s], sizeof(Vector3) * (size_t) width * height) != RT_OK) {
				fprintf(stderr, "Error: %s\n", rt_last_error());
				return -1;
			}
		for (int k = 0; k < (warmup > 1 ? warmup : 1); k++)     /* warm-up: allocations, first launches on both streams */
			if (rt_multi_frame_submit(group, &p, k & 1, buf[k & 1]) != RT_OK || rt_multi_frame_wait(group, k & 1) < 0) {
				fprintf(stderr, "Error: %s\n", rt_last_error());
				return -1;
			}
		rt_profile_enable(rt_multi_context(group, 0), 1);      /* the library's own per-launch events on the first device */
		double t0 = now_s();
		int rc = rt_multi_frame_submit(group, &p, 0, buf[0]);
		for (int k = 0; k < frames && rc >= 0; k++) {
			if (k + 1 < frames) {
				rt_render_params q = p;
				q.seed = seed + (unsigned long long) (k + 1);
				rc = rt_multi_frame_submit(group, &q, (k + 1) & 1, buf[(k + 1) & 1]);
				if (rc < 0) break;
			}
			rc = rt_multi_frame_wait(group, k & 1);       /* frame k is in buf[k & 1]: update_frame() would present it now */
		}
		if (rc < 0) {
			fprintf(stderr, "Error: %s\n", rt_last_error());
			return -1;
		}
		double dt = now_s() - t0;
		double kernel_ms = 0, span_ms = 0;
		int launches = 0;
		rt_profile_collect_span(rt_multi_context(group, 0), &kernel_ms, &launches, &span_ms);
		if (launches < 1) launches = 1;
		fprintf(stderr, "Rendered %d frames of %dx%d, %d spp, %d bounces on %d GPU(s), two in flight: %.3f ms per frame, %.1f Msamples/s (frames in host memory); "
		        "kernels of the first device: %.3f ms per launch, %.3f ms of span per launch\n",
		        frames, width, height, spp, bounces, rt_multi_size(group), dt / frames * 1e3, (double) width * height * spp * frames / dt / 1e6,
		        kernel_ms / launches, span_ms / launches);
		printf("{\"frames\": %d, \"ms_per_frame\": %.4f, \"msamples_per_s\": %.2f, \"gpus\": %d, \"kernel_ms_per_launch\": %.4f, \"span_ms_per_launch\": %.4f, "
		       "\"scene_kernel\": \"%s\", \"compile_s\": %.4f}\n",
		       frames, dt / frames * 1e3, (double) width * height * spp * frames / dt / 1e6, rt_multi_size(group), kernel_ms / launches, span_ms / launches,
		       rt_compiled_scene_info(rt_multi_context(group, 0)), compile_s);
		rt_move_frame_to_the_gpu(width, height, buf[(frames - 1) & 1]);   /* where update_frame() hands off, main.c:479 */
		rt_host_free(buf[0]); rt_host_free(buf[1]);
		rt_free_cubemap(&skybox);
		rt_multi_destroy(group);
		return 0;
	}

	Vector3 *frame = malloc(sizeof(Vector3) * (size_t) width * height);
	if (!frame) { printf("OUT OF MEMORY\n"); return -1; }

	if (interactive > 0) {
		/* worker() + update_frame() with the GPU in the workers' place: the ladder runs on the device(s), the host asks for the
		 * passes between two displayed frames in one call and resolves a frame after each (main.c:354-408, 450-482) */
		if (present_every < 1) present_every = 1;
		if (rt_multi_progressive_begin(group, width, height, init_scale, bounces, seed) != RT_OK) {
			fprintf(stderr, "Error: %s\n", rt_last_error());
			return -1;
		}
		double t0 = now_s();
		int done = 0, shown = 0;
		while (done < interactive) {
			const int n = interactive - done < present_every ? interactive - done : present_every;
			if (rt_multi_progressive_passes(group, n) != RT_OK || rt_multi_progressive_resolve(group, frame) != RT_OK) {
				fprintf(stderr, "Error: %s\n", rt_last_error());
				return -1;
			}
			done += n; shown++;
		}
		double dt = now_s() - t0;
		int next_scale = 0, passes = 0; float count = 0; uint32_t generation = 0;
		rt_multi_progressive_state(group, &next_scale, &count, &generation, &passes);
		fprintf(stderr, "Interactive: %d passes from scale 1/%d at %dx%d, %d bounces on %d GPU(s), %d frames resolved: %.3f s, %.3f ms per pass\n",
		        passes, init_scale, width, height, bounces, rt_multi_size(group), shown, dt, dt / interactive * 1e3);
		printf("{\"passes\": %d, \"frames_resolved\": %d, \"ms_per_pass\": %.4f, \"weight_sum\": %.6f, \"next_scale\": %d, \"gpus\": %d}\n",
		       passes, shown, dt / interactive * 1e3, (double) count, next_scale, rt_multi_size(group));
		rt_move_frame_to_the_gpu(width, height, frame);
		free(frame);
		rt_free_cubemap(&skybox);
		rt_multi_destroy(group);
		return 0;
	}

	double t0 = now_s();
	if (rt_multi_render(group, &p, frame) != RT_OK) {
		fprintf(stderr, "Error: %s\n", rt_last_error());
		return -1;
	}
	double dt = now_s() - t0;
	fprintf(stderr, "Rendered %dx%d, %d spp, %d bounces on %d GPU(s) in %.3f s (%.1f Msamples/s incl. copy-back)\n",
	        width, height, spp, bounces, rt_multi_size(group), dt, (double) width * height * spp / dt / 1e6);

	rt_move_frame_to_the_gpu(width, height, frame);   /* where update_frame() hands off, main.c:479 */

	free(frame);
	rt_free_cubemap(&skybox);
	rt_multi_destroy(group);
	return 0;
}
