/*
 * rt_types.h -- the plain-data structures that cross the drop-in boundary.
 *
 * These are declared with the SAME names, field order and layout as the reference host uses, so a
 * `Scene` / `Cubemap` filled in by the reference's own loader can be handed to this library by
 * pointer and a frame produced here can be handed to the reference's presenter unchanged:
 *
 *   Vector3, Ray, Sphere        reference src/vector.h:32-36, 53-56, 58-61
 *   Material, Cube, Object,
 *   Scene, MAX_OBJECTS          reference src/scene.h:3-36   (sizeof(Object)=68, sizeof(Scene)=69636)
 *   Cubemap, CubeFace           reference src/gpu_and_windowing.h:4-16
 *
 * A host that already includes the reference's scene.h / gpu_and_windowing.h defines
 * RT_HAVE_REFERENCE_TYPES before including this header and gets only the
 * library-specific types below.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef RT_HAVE_REFERENCE_TYPES

typedef struct { float x, y, z; } Vector3;

typedef struct { Vector3 origin, direction; } Ray;

typedef struct { Vector3 center; float radius; } Sphere;

typedef struct {
	Vector3 albedo;
	float   roughness;
	float   reflectance;
	float   metallic;
	float   emission_power;
	Vector3 emission_color;
} Material;

typedef struct { Vector3 origin, size; } Cube;

typedef enum { OBJECT_CUBE, OBJECT_SPHERE } ObjectType;

typedef struct {
	ObjectType type;
	union { Sphere sphere; Cube cube; };
	Material material;
} Object;

#define MAX_OBJECTS 1024

typedef struct {
	Object objects[MAX_OBJECTS];
	int    num_objects;
} Scene;

/* Six decoded faces, indexed by CubeFace; rows top-first, `chan` interleaved bytes per texel. */
typedef struct {
	uint8_t *data[6];
	int w, h, chan;
} Cubemap;

typedef enum { CF_FRONT, CF_BACK, CF_LEFT, CF_RIGHT, CF_TOP, CF_BOTTOM } CubeFace;

#endif /* RT_HAVE_REFERENCE_TYPES */

/*
 * Camera pose.  The reference keeps these as file-statics (src/camera.c:28,33-35) reachable only
 * through ray_through_screen_at(); a library boundary needs them as data.  `fov` is passed to
 * tan(fov/2) as-is, exactly like camera.c:107 (the reference's default 30.0f is therefore radians).
 */
typedef struct {
	Vector3 pos;
	Vector3 front;
	Vector3 up;
	float   fov;
} rt_camera;

/* Frame-constant camera terms, computed once on the host with the reference's roundings
 * (camera.c:99-118): dir(px,py) = (llc + horizontal*px + vertical*py) - pos, see rt_camera_basis(). */
typedef struct {
	Vector3 pos;
	Vector3 lower_left_corner;
	Vector3 horizontal;
	Vector3 vertical;
} rt_camera_basis;

#ifdef __cplusplus
}
#endif

#endif /* RT_TYPES_H */
