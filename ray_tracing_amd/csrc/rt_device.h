/*
 * rt_device.h -- device-side layouts shared by the host packer (rt_api.cpp) and the HIP kernels.
 *
 * The reference walks `Scene.objects[]` as 68-byte AoS records passed BY VALUE per test
 * (scene.c:136,166).  Here every frame-constant term is folded on the host, once, with the
 * reference's own roundings, and the scene is split into
 *   - a 32-byte geometry record per object  -> staged in LDS, read as two wave-uniform b128 loads;
 *   - a 64-byte shading record per object   -> staged in LDS, touched once per bounce.
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#else   /* hiprtc (rt_compile_scene): no host headers */
#ifndef RT_RTC_STDINT
#define RT_RTC_STDINT
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
#endif
#endif

#define RT_GEOM_CUBE   0
#define RT_GEOM_SPHERE 1

/* 32 B.  cube: a = origin, b = origin*1 + size*1 (scene.c:27);  sphere: a = center, b0 = r*r (scene.c:112) */
typedef struct {
	float a[3];
	float b0;
	float b1, b2;
	int   type;
	int   pad;
} rt_geom;

/* 64 B.  All of main.c:212-249's per-material terms that do not depend on the ray. */
typedef struct {
	float f0[3];      /* combine(vec(0.16*r*r), albedo, 1-metallic, metallic)   main.c:219-221 */
	float roughness;
	float one_minus_f0[3]; /* combine(vec(1), f0, 1, -1)                        main.c:128     */
	int   is_metal;   /* (double) metallic > 0.001                              main.c:241     */
	float tint[3];    /* scalev(albedo, 1 - metallic)                           main.c:248     */
	float pad0;
	float emission[3];/* scalev(emission_color, emission_power)                 main.c:232,203 */
	float pad1;
} rt_shade;

/* Large scenes (more objects than a scene-specialised kernel takes): objects grouped into clusters of up to RT_CLUSTER_SIZE
 * spatially close ones, so that a ray first asks RT_CLUSTER_SIZE times fewer boxes "could anything in here be hit at all"
 * (rt_kernels.hip nearest_hit_culled, rt_cull.h).  112 bytes = seven 16-byte LDS slots (an odd number: a wave's lanes reading one
 * field of different clusters spread over all sixteen slots of a bank row): a conservative bounding box of the members' own
 * conservative boxes; the members' object indices (0xffff: none); and the members' conservative boxes once more, QUANTISED on the
 * cluster's own grid -- RT_CLUSTER_GRID steps from lo to hi on every axis, a byte per plane, rounded outwards (rt_cull.h): what the
 * lanes test a cluster's members against, 64 bytes per cluster instead of 8 x 32 of geometry records. */
#define RT_CLUSTER_SIZE 8
#define RT_MAX_CLUSTERS 128
#define RT_CLUSTER_F4 7              /* a cluster as float4 words: 2 of box and count, 1 of members, 3 of quantised member boxes, 1 unused */
#define RT_CLUSTER_GRID 254
typedef struct {
	float          lo[3], hi0;
	float          hi1, hi2;
	int            count, pad;
	unsigned short member[RT_CLUSTER_SIZE];
	unsigned char  qpair[RT_CLUSTER_SIZE / 2][12];   /* the members' boxes in steps of (hi - lo) * RN(1 / RT_CLUSTER_GRID) from lo, two members to twelve bytes (RT_QPLANE below) */
	unsigned char  pad2[16];                   /* (seven slots: an odd number, see above) */
} rt_cluster;
/* Plane `upper` (0 / 1) of axis k of box j in an array of box PAIRS: three words per pair -- word 0: box 2p (lo.x hi.x lo.y hi.y), word 1:
 * box 2p + 1 likewise, word 2: lo.z hi.z of box 2p, lo.z hi.z of box 2p + 1.  The two planes of an axis are neighbours so that ONE byte
 * permute per word (v_perm_b32 with a selector formed once per ray from the signs of its direction) puts the plane the ray meets first in
 * front: the slab test then needs no min / max per axis (round 6: they were a third of its cycles -- v_min / v_max issue at half the rate
 * of v_fma on gfx950, profiles/r02/valu_rates.txt). */
#define RT_QPLANE(qpair, j, k, upper) ((qpair)[(j) >> 1][(k) < 2 ? 4 * ((j) & 1) + 2 * (k) + (upper) : 8 + 2 * ((j) & 1) + (upper)])
/* the grid's step along one axis, formed the same way -- one subtraction, one product by the literal -- on the host and in the kernels */
#define RT_CLUSTER_STEP(lo, hi) (((hi) - (lo)) * (1.0f / (float) RT_CLUSTER_GRID))

/* One level up (round 6): GROUPS of RT_GROUP_SIZE consecutive clusters -- consecutive in the order of the median-split tree, i.e.
 * neighbours in space.  Asking every cluster box for every ray, wave-uniformly, was half the instructions of a culled trace at 1024
 * objects (128 steps, of which a ray passes 4 ... 7); now the wave asks the <= 16 group boxes uniformly, and the (ray, group) pairs
 * that pass are DEALT to the lanes 64 at a time -- as the (ray, cluster) pairs always were -- each lane testing its pair's eight
 * cluster boxes: 16 + 3 ... 6 x 8 box tests per trace instead of 128.  A per-lane walk of a tree over the clusters would cost a wave
 * the MAXIMUM over its lanes -- 75 ... 85 node visits, each with its own LDS reads (scripts/sim/bvh_walk_sim.py) --: dealing keeps
 * the lanes full.  A group record: its box (the union of its clusters' boxes), the count of its clusters, and its clusters' boxes
 * quantised outwards on the group's grid exactly as a cluster's members are on the cluster's (80 bytes = five 16-byte slots).  The
 * records follow the cluster records in the same buffer (rt_launch.clusters + num_clusters): RT_CULL_F4 float4 words in all. */
#define RT_GROUP_SIZE 8
#define RT_GROUP_F4 5
#define RT_NUM_GROUPS(clusters) (((clusters) + RT_GROUP_SIZE - 1) / RT_GROUP_SIZE)
#define RT_CULL_F4(clusters) (RT_CLUSTER_F4 * (clusters) + RT_GROUP_F4 * RT_NUM_GROUPS(clusters))
/* scenes of fewer clusters ask every cluster box, as before: dealing costs more than it saves there -- groups from 17 clusters: +5 ... 8 % at
 * 136 ... 224 objects; from 33: +2 ... 4 % at 272 ... 448; at 512 objects (64 clusters) -6 %, 768 -11 %, 1024 -22 % (profiles/r06/ab_groups_threshold.txt) */
#define RT_GROUPS_FROM_CLUSTERS 60
#define RT_GROUP_PAIRS_MAX 448u       /* (ray, group) pairs of a wave beyond which it asks every cluster box instead: seven dealt steps (rt_kernels.hip) */
typedef struct {
	float          lo[3], hi0;
	float          hi1, hi2;
	int            count, pad;                 /* clusters in the group (the last group may have fewer than RT_GROUP_SIZE) */
	unsigned char  qpair[RT_GROUP_SIZE / 2][12];   /* the group's clusters' boxes on the group's grid (RT_CLUSTER_STEP of the group's box), in pairs (RT_QPLANE) */
} rt_group;

/* The control words of a launch: one 128-byte line behind the pixel lists' counters (rt_launch.control), cleared when the
 * launch is enqueued, copied to the host behind it (the first RT_CTL_WORDS words) and judged there (rt_api.cpp judge_launch):
 * a frame is delivered whole or not at all -- the reference publishes a column only when render_column() has returned for all
 * of it (main.c:377-396). */
#define RT_CTL_STAMP       0    /* the launch's number, written by the LAST wave of the trace kernel to leave, behind the sums below:
                                 * a launch whose waves did not all leave has no stamp */
#define RT_CTL_CANCELLED   1    /* by the last wave: a wave gave up because of rt_cancel() (the relay word below was set): the frame is incomplete, and meant to be */
#define RT_CTL_STOP_RELAY  2    /* the request, relayed from the waves that read the host's word to all others; back to zero when the last wave leaves */
#define RT_CTL_WAVES_LEFT  3    /* by the last wave: waves of the trace kernel that have left (the workgroups count themselves on the lists' dequeue lines) */
#define RT_CTL_WRITTEN     4    /* by the last wave: object pixels resolved and written to the frame (the workgroups report them on the lists' dequeue lines as they leave) */
#define RT_CTL_PRIMARY     5    /* by the last wave: 8x8 pixel blocks the camera-ray pass finished (sum over the lists' lines) */
#define RT_CTL_LISTED      6    /* by the last wave: object pixels the camera-ray pass listed */
#define RT_CTL_FETCHED     7    /* by the last wave: ... of which the trace kernel's waves fetched */
#define RT_CTL_AUDITED     8    /* (64 bits) by the last wave (sum over the dequeue lines' words 4-5 / 6-7): soft-shadow taps answered by rt_lit.h that were traced all the same (rt_launch.audit_taps) */
#define RT_CTL_DISAGREE    10   /* (64 bits) ... whose trace contradicts the answer */
#define RT_CTL_LINES_DONE  12   /* dequeue lines whose workgroups have all left: the workgroup that completes the last one is the launch's last -- and sets this, and the
                                 * dequeue lines, back to zero: a launch that keeps its scratch set's pixel lists has nothing to clear (rt_launch_trace) */
#define RT_CTL_WORDS       16
#define RT_AUDIT_MAX_LOG2  24   /* rt_tuning.audit_known_taps = k marks one answer in 2^k by a 24-bit hash of its cell / pixel: k up to this */
/* What a launch is expected to leave in those words, kept by the host until they have been copied back and judged. */
typedef struct {
	unsigned int launch_id;
	int          stamped;          /* the kernel stamps the launch (the persistent trace kernels do; rt_trace_simple, the cross-check kernel, does not) */
	unsigned int primary_blocks;   /* 8x8 pixel blocks of the camera-ray pass */
} rt_launch_expect;
/* the ladder's count words (rt_progressive_*): [0] the sum of the published passes' weights (float), [1] a word that stays
 * zero, [RT_COUNT_INCOMPLETE] launches that were not published because they were incomplete (unsigned) */
#define RT_COUNT_INCOMPLETE 2
#define RT_COUNT_WORDS      4

typedef struct {
	/* camera.c:99-118, frame constants */
	float pos[3], llc[3], horiz[3], vert[3];
	/* first emitter (main.c:140-146) and its origin_of() (scene.c:10-15) */
	int   light_index;
	float light_pos[3];
	int   only_light_emits;    /* that emitter is the only object whose emission is not all zeros: a tap then adds something only if
	                            * the emitter is its nearest hit, and "certainly not" is as good an answer as "certainly" (rt_lit.h) */

	int   num_objects;
	int   width, height;       /* full frame                                   */
	int   spp, max_bounces;
	uint64_t seed;
	/* pixel (i, j) of this launch: u = i / u_den, v = j / v_den (main.c:293-294: lowres_frame_w - 1,
	 * lowres_frame_h - 1); its paths are seeded with pixel index (j*pix_scale)*pix_width + i*pix_scale
	 * and sample index sample_base + s.  Full-resolution frame: u_den = width-1, v_den = height-1,
	 * pix_scale = 1, pix_width = width, sample_base = 0. */
	int   u_den, v_den, pix_scale, pix_width, sample_base;

	/* interleaved row-block partition: local row r -> global row
	 * ((r / row_block) * world + rank) * row_block + r % row_block          */
	int   row_block, rank, world;
	int   local_rows;          /* rows this launch renders (incl. none past `height`)        */

	/* skybox: 6 faces of RGBA8 texels, face-major                            */
	const uint32_t *sky;
	int   sky_w, sky_h;
	float sky_wm1, sky_hm1;    /* (float)(w - 1), (float)(h - 1): the texel scale of gpu_and_windowing.c:103-104 */

	float *frame;              /* local_rows x width x 3 floats, resolved      */
	/* several interactive passes at full resolution in one launch (rt_progressive_passes): the sums the frame rows hold so far,
	 * laid out like `frame`.  A pixel's samples are then added, in order, to sum_onto[...] instead of to zero and the result is
	 * written to `frame` as it is, not divided by the sample count -- worker()'s publish step (main.c:394) `spp` times over.
	 * NULL: a frame of its own. */
	const float *sum_onto;
	/* rt_lit.h audited in production (rt_tuning.audit_known_taps): 0 = no; a power of two 2^k = the taps of one in 2^k bounces
	 * whose taps are answered without tracing are traced all the same and compared (the frame still uses the answer; a
	 * disagreement is counted into control[RT_CTL_DISAGREE] and fails the launch on the host) */
	unsigned int audit_taps;
	int    trace_workgroups;   /* workgroups of the trace kernel (set by the launcher): the last one to leave stamps the launch */
	unsigned int test_drop_pixels;   /* TESTING AID (rt_tuning.test_drop_pixels): the trace kernel's waves see every pixel list this many entries shorter */
	int    skip_known_taps;    /* rt_primary_pass flags the pixels whose bounce-0 taps need no tracing (rt_lit.h); 0: every tap is traced */
	/* the same answer for hit points of any bounce, from a table built once per scene (rt_lit.h: one entry per cell of a grid
	 * over every object's bounding box, rt_lit_grid per object); NULL: no table (no sphere emitter, or every tap is traced) */
	const unsigned char *lit_cells;     /* one byte per cell: 1 = every surface point in it is such a point, 2 = from every one the accepted taps certainly do NOT reach the emitter first;
	                                     * | 4: the launch audits this cell's answer (the host marks one cell in audit_taps when the audit is on) */
	const void     *lit_grids;
	int             lit_grids_in_lds;   /* the trace kernel's workgroups keep a copy of the grids behind the scene records */
	/* scheduling of the wavefront kernels (any values give the same frame):
	 *   num_shards   pixel lists in use (1 or 64), each with its own fill and dequeue counter */
	int    num_shards;
	/* written by rt_primary_pass, read by the trace kernels: one 12-word record per object pixel (camera-ray hit
	 * point xyz, normal xyz, object, camera ray xyz, RNG pixel index, offset in the strip), record c = the 48 bytes at
	 * pix + 12 c (three 16-byte words); list s holds records s * pix_shard_cap ... + pix_count[32 * s] */
	float *pix;
	unsigned int *pix_count;   /* one fill counter per list, 128 bytes apart; the word behind it: 8x8 blocks finished by the waves that append to the list */
	int    pix_shard_cap;
	/* the launch's control words (RT_CTL_* above).  [RT_CTL_CANCELLED]: set by a wave that gave up because of rt_cancel() -- the frame is incomplete.  The request itself is a word
	 * in host memory the device can read (`stop`, written by a plain store of the calling thread: nothing has to get past the
	 * kernel that fills the chip): "the launches up to this number are to stop"; this launch is number launch_id. */
	unsigned int *control;
	const unsigned int *stop;
	unsigned int launch_id;
	const rt_geom  *geom;      /* num_objects records (global; staged to LDS)  */
	const rt_shade *shade;
	/* object culling for large scenes (rt_cull.h): num_clusters = 0 -> every ray tests every object, as the reference does */
	const rt_cluster *clusters;
	int   num_clusters;
	float cull_margin;         /* every box is tested as if it were this much larger on every side ...                         */
	float cull_origin_max;     /* ... which covers the rounding of the tests for rays that start within this of the origin      */
} rt_launch;

#endif
