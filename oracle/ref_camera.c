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
