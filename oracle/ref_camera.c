/*
 * TEST INFRASTRUCTURE ONLY.  Separate TU because the reference's camera pose is file-static
 * (camera.c:28,33-35); including camera.c here lets tests move the camera without editing it.
 */
#include "camera.c"

__attribute__((visibility("default")))
void ref_set_camera(const float pos[3], const float front[3], const float up[3], float fov_value)
{
	camera_pos   = (Vector3) {pos[0], pos[1], pos[2]};
	camera_front = (Vector3) {front[0], front[1], front[2]};
	camera_up    = (Vector3) {up[0], up[1], up[2]};
	fov          = fov_value;
}

/* interaction (camera.c:42-88) and a read-back of the statics, for tests of the product's mirror */
__attribute__((visibility("default"))) void ref_move_camera(int dir, float speed) { move_camera((Direction) dir, speed); }
__attribute__((visibility("default"))) void ref_rotate_camera(double x, double y) { rotate_camera(x, y); }
__attribute__((visibility("default"))) void ref_reset_mouse(void)
{
	first_mouse = true; yaw = -90.0f; pitch = 0.0f; last_x = 800.0f / 2.0; last_y = 600.0f / 2.0;
}
__attribute__((visibility("default"))) void ref_get_camera(float out[9])
{
	out[0] = camera_pos.x;   out[1] = camera_pos.y;   out[2] = camera_pos.z;
	out[3] = camera_front.x; out[4] = camera_front.y; out[5] = camera_front.z;
	out[6] = camera_up.x;    out[7] = camera_up.y;    out[8] = camera_up.z;
}
