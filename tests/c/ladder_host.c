/*
 * ladder_host.c -- runs the TEXT of the integration patch on a GPU.
 *
 * scripts/patches/reference_main_rt.py inserts three pieces of C into the reference's main.c: a header (the context, two
 * helpers), invalidate_accumulation() and update_frame().  The reference cannot travel to the GPU box, so this file supplies
 * stand-ins for the handful of reference globals and functions that text touches -- `frame`, `frame_w`, `frame_h`, `init_scale`,
 * the window size, realloc_frame_buffer(), the camera getters, move_frame_to_the_gpu() -- with the meaning they have in
 * main.c (:50, :71-75, :416-448) / camera.c (:33-37) / gpu_and_windowing.c (:361-376), and #includes the patch's text
 * unchanged (BINDING_TEXT: a file written by `reference_main_rt.py binding --ladder`; or `binding --blocking` with
 * -DBINDING_IS_BLOCKING, the variant that renders one independent frame per update_frame()).  main() then plays the part of the
 * reference's event loop (main.c:520-574):
 *
 *     ladder_host <scene> <skybox dir> <w> <h> <init_scale> <frames before> <frames after> <out.raw>
 *
 * `frames before` calls of update_frame(), a camera move + invalidate_accumulation() (main.c:540-563), `frames after` more
 * calls; the last frame handed to move_frame_to_the_gpu() is written to out.raw as w*h*3 floats, and a JSON line says how
 * many passes the library counts.  tests/test_gpu_ladder_binding.py compares that frame with the oracle's ladder.
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_types.h"          /* Vector3, Scene, Cubemap with the reference's layouts (the patch text then sees them as "reference types") */

/* ---- stand-ins for what the patch text uses of the reference ---- */
int      init_scale;           /* main.c:50 */
Vector3 *frame;                /* main.c:71 */
int      frame_w, frame_h;     /* main.c:74-75 */
static int screen_w, screen_h; /* gpu_and_windowing.c:361-369 get_screen_w/h */
static Vector3 camera_pos = {5, 5, 5}, camera_front = {-1, -1, -1}, camera_up = {0, 1, 0};   /* camera.c:33-35 */
static Vector3 *shown; static int shown_count;

Vector3 get_camera_pos(void)   { return camera_pos; }      /* camera.c:37 */
Vector3 get_camera_front(void) { return camera_front; }    /* the two getters the camera.c patch adds */
Vector3 get_camera_up(void)    { return camera_up; }
bool frame_buffer_size_doesnt_match_window(void) { return frame_w != screen_w || frame_h != screen_h; }   /* main.c:445-448 */
void realloc_frame_buffer(void)                            /* main.c:416-443, the part that concerns `frame` */
{
	frame_w = screen_w; frame_h = screen_h;
	free(frame);
	frame = malloc(sizeof(Vector3) * frame_w * frame_h);
	if (!frame) { printf("OUT OF MEMORY\n"); abort(); }
	memset(frame, 0, sizeof(Vector3) * frame_w * frame_h);
}
void move_frame_to_the_gpu(int w, int h, Vector3 *data)    /* gpu_and_windowing.h:44: the presenter */
{
	memcpy(shown, data, sizeof(Vector3) * (size_t) w * h);
	shown_count++;
}

#include BINDING_TEXT          /* the patch's header + invalidate_accumulation() + update_frame(), verbatim */

int main(int argc, char **argv)
{
	if (argc != 9) { fprintf(stderr, "usage: ladder_host scene skybox_dir w h init_scale frames_before frames_after out.raw\n"); return 2; }
	static Scene scene;
	static Cubemap skybox;
	if (rt_parse_scene_file(argv[1], &scene) != RT_OK) return 1;
	char paths[6][1024]; const char *files[6];
	static const char *names[6] = { "front.jpg", "back.jpg", "left.jpg", "right.jpg", "top.jpg", "bottom.jpg" };   /* CubeFace order */
	for (int f = 0; f < 6; f++) { snprintf(paths[f], sizeof paths[f], "%s/%s", argv[2], names[f]); files[f] = paths[f]; }
	if (rt_load_cubemap(&skybox, files) != RT_OK) return 1;
	screen_w = atoi(argv[3]); screen_h = atoi(argv[4]); init_scale = atoi(argv[5]);
	const int before = atoi(argv[6]), after = atoi(argv[7]);
	shown = malloc(sizeof(Vector3) * (size_t) screen_w * screen_h);

	/* what the patch puts in place of start_workers() (main.c:516) */
	if (rt_create(&rt, 0) || rt_set_scene(rt, &scene) || rt_set_skybox(rt, &skybox)) {
		fprintf(stderr, "rt: %s\n", rt_last_error());
		return -1;
	}

	invalidate_accumulation();                 /* an input event before the first frame (main.c:540-563) must be harmless */
	for (int k = 0; k < before; k++) update_frame();
	/* EVENT_PRESS_W ... (main.c:540-563): the camera moves, the accumulation is invalidated */
	camera_pos = (Vector3) {2, 3, 9}; camera_front = (Vector3) {0.1f, -0.3f, -1};
	invalidate_accumulation();
	for (int k = 0; k < after; k++) update_frame();

#ifdef BINDING_IS_BLOCKING          /* the --blocking variant keeps no ladder: frames shown since the last invalidation = the seed of the next one */
	printf("{\"frames_since_invalidation\": %d, \"frames_shown\": %d}\n", rt_passes, shown_count);
#else
	int next_scale = 0, passes = 0; float count = 0; uint32_t generation = 0;
	rt_progressive_state(rt, &next_scale, &count, &generation, &passes);
	printf("{\"passes\": %d, \"next_scale\": %d, \"weight_sum\": %.9g, \"generation\": %u, \"frames_shown\": %d}\n", passes, next_scale, (double) count, generation, shown_count);
#endif
	FILE *f = fopen(argv[8], "wb");
	if (!f || fwrite(shown, sizeof(Vector3), (size_t) screen_w * screen_h, f) != (size_t) screen_w * screen_h) return 1;
	fclose(f);
	invalidate_accumulation();                 /* main.c:575: "tell workers to stop" */
	rt_destroy(rt);                            /* the patch's replacement of stop_workers() */
	rt_free_cubemap(&skybox);
	return 0;
}
