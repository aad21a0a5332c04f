/*
 * rt_host.c -- host-side mirror of the reference interface around the hot path (plain C11).
 *
 *   rt_parse_scene_file / _string   scene.c:206-624  (grammar, defaults, range checks, messages)
 *   rt_camera_default / _basis_for  camera.c:28,33-35 / camera.c:99-118
 *   rt_move_camera / rt_rotate_camera   camera.c:80-88 / camera.c:42-78
 *   rt_move_frame_to_the_gpu        gpu_and_windowing.h:44 hand-off signature, headless sink
 *   rt_write_ppm                    screenshot() main.c:637-681 (u8 conversion + flip)
 *
 * Built with -std=c11 -ffp-contract=off: the float recurrences of the number reader and the
 * camera basis must round exactly like the reference binary.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rt_hip.h"

/* ------------------------------------------------------------------------------------------ */
/* scene text                                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
	const char *text;
	size_t      len;
	size_t      at;
	int         line;
} Reader;

static int is_blank(char c) { return c == ' ' || c == '\r' || c == '\t' || c == '\n'; }
static int is_num(char c)   { return c >= '0' && c <= '9'; }

static void eat_blanks(Reader *r)
{
	while (r->at < r->len && is_blank(r->text[r->at])) {
		if (r->text[r->at] == '\n') r->line++;
		r->at++;
	}
}

static char peek(const Reader *r) { return r->at < r->len ? r->text[r->at] : '\0'; }

/* The reference matches a keyword only if strictly more than `need` characters remain: `need` is
 * the keyword length minus one (`5 < len - i` for "sphere", scene.c:224) except for "albedo", whose
 * guard is its full length (`6 < len - i`, scene.c:271), so "albedo" cannot be the file's last bytes. */
static int keyword(const Reader *r, const char *kw, size_t need)
{
	if (r->at > r->len) return 0;
	if (!(need < r->len - r->at)) return 0;
	return strncmp(r->text + r->at, kw, strlen(kw)) == 0;
}

/* scene.c:441-461 / 491-511.  Integer part v = v*10 + d, fraction v += q*d with q = 0.1f, q /= 10,
 * all in float.  No exponent, no leading '+' or '.'.  `slot` < 0: scalar property. */
static int number(Reader *r, int slot, float *out)
{
	int negative = 0;
	if (peek(r) == '-') {
		negative = 1;
		r->at++;
		if (r->at >= r->len || !is_num(r->text[r->at])) {
			fprintf(stderr, "Error: Missing number after minus sign (line %d)\n", r->line);
			return 0;
		}
	} else if (!is_num(peek(r))) {
		if (slot < 0) fprintf(stderr, "Error: Missing number after property name (line %d)\n", r->line);
		else          fprintf(stderr, "Error: Missing number %d in vector value (line %d)\n", slot, r->line);
		return 0;
	}
	float acc = 0;
	while (r->at < r->len && is_num(r->text[r->at])) {
		int d = r->text[r->at++] - '0';
		acc = acc * 10 + d;
	}
	if (r->at < r->len && r->text[r->at] == '.') {
		r->at++;
		if (r->at >= r->len || !is_num(r->text[r->at])) {
			fprintf(stderr, "Error: Missing decimal part after dot (line %d)\n", r->line);
			return 0;
		}
		float place = 1.0f / 10;
		while (r->at < r->len && is_num(r->text[r->at])) {
			int d = r->text[r->at++] - '0';
			acc += place * d;
			place /= 10;
		}
	}
	*out = negative ? acc * -1 : acc * 1;
	return 1;
}

static int triple(Reader *r, Vector3 *out)
{
	if (peek(r) != '{') {
		fprintf(stderr, "Error: Missing '{' after property name (line %d)\n", r->line);
		return 0;
	}
	r->at++;
	float c[3];
	for (int k = 0; k < 3; k++) {
		eat_blanks(r);
		if (!number(r, k, &c[k])) return 0;
	}
	eat_blanks(r);
	if (peek(r) != '}' || r->at >= r->len) {
		fprintf(stderr, "Error: Missing '}' after property value (line %d)\n", r->line);
		return 0;
	}
	r->at++;
	out->x = c[0]; out->y = c[1]; out->z = c[2];
	return 1;
}

static int outside01(float f) { return f < 0 || f > 1; }
static int any_outside01(Vector3 v) { return outside01(v.x) || outside01(v.y) || outside01(v.z); }

static void reset_material(Material *m)
{
	m->albedo = (Vector3) {0.44, 0.68, 0.84};      /* scene.c:234 */
	m->roughness = 0;
	m->reflectance = 0.2;
	m->metallic = 0;
	m->emission_power = 0;
	m->emission_color = (Vector3) {1, 1, 1};
}

/* returns 1: property consumed, 0: no property here (object ends), -1: error */
static int property(Reader *r, Object *obj)
{
	float f; Vector3 v;
	int sphere = obj->type == OBJECT_SPHERE;

#define SKIP_TO_VALUE(n)                                                                   \
	do {                                                                                   \
		r->at += (n);                                                                      \
		eat_blanks(r);                                                                     \
		if (r->at >= r->len) {                                                             \
			fprintf(stderr, "Error: Property value is missing (line %d)\n", r->line);      \
			return -1;                                                                     \
		}                                                                                  \
	} while (0)

	if (keyword(r, "albedo", 6)) {
		SKIP_TO_VALUE(9);                                  /* sic: 9, scene.c:280 */
		if (!triple(r, &v)) return -1;
		if (any_outside01(v)) { fprintf(stderr, "Error: albedo values must be between 0 and 1 (line %d)\n", r->line); return -1; }
		obj->material.albedo = v;
	} else if (keyword(r, "roughness", 8)) {
		SKIP_TO_VALUE(9);
		if (!number(r, -1, &f)) return -1;
		if (outside01(f)) { fprintf(stderr, "Error: Roughness must be between 0 and 1 (line %d)\n", r->line); return -1; }
		obj->material.roughness = f;
	} else if (keyword(r, "reflectance", 10)) {
		SKIP_TO_VALUE(11);
		if (!number(r, -1, &f)) return -1;
		if (outside01(f)) { fprintf(stderr, "Error: Reflectance must be between 0 and 1 (line %d)\n", r->line); return -1; }
		obj->material.reflectance = f;
	} else if (keyword(r, "metallic", 7)) {
		SKIP_TO_VALUE(11);                                 /* sic: 11, scene.c:320 */
		if (!number(r, -1, &f)) return -1;
		if (outside01(f)) { fprintf(stderr, "Error: Metallic must be between 0 and 1 (line %d)\n", r->line); return -1; }
		obj->material.metallic = f;
	} else if (keyword(r, "emission_power", 13)) {
		SKIP_TO_VALUE(14);
		if (!number(r, -1, &f)) return -1;
		obj->material.emission_power = f;
	} else if (keyword(r, "emission_color", 13)) {
		SKIP_TO_VALUE(14);
		if (!triple(r, &v)) return -1;
		if (any_outside01(v)) { fprintf(stderr, "Error: Emission color values must be between 0 and 1 (line %d)\n", r->line); return -1; }
		obj->material.emission_color = v;
	} else if (keyword(r, "radius", 5)) {
		if (!sphere) { fprintf(stderr, "Poperty 'radius' only allowed on spheres (line %d)\n", r->line); return -1; }
		SKIP_TO_VALUE(6);
		if (!number(r, -1, &f)) return -1;
		obj->sphere.radius = f;
	} else if (keyword(r, "center", 5)) {
		if (!sphere) { fprintf(stderr, "Poperty 'center' only allowed on spheres (line %d)\n", r->line); return -1; }
		SKIP_TO_VALUE(6);
		if (!triple(r, &v)) return -1;
		obj->sphere.center = v;
	} else if (keyword(r, "origin", 5)) {
		if (sphere) { fprintf(stderr, "Poperty 'origin' only allowed on cubes (line %d)\n", r->line); return -1; }
		SKIP_TO_VALUE(6);
		if (!triple(r, &v)) return -1;
		obj->cube.origin = v;
	} else if (keyword(r, "size", 3)) {
		if (sphere) { fprintf(stderr, "Poperty 'size' only allowed on cubes (line %d)\n", r->line); return -1; }
		SKIP_TO_VALUE(4);
		if (!triple(r, &v)) return -1;
		if (v.x < 0 || v.y < 0 || v.z < 0) { fprintf(stderr, "Error: Size values must be positive (line %d)\n", r->line); return -1; }
		obj->cube.size = v;
	} else
		return 0;
#undef SKIP_TO_VALUE
	return 1;
}

int rt_parse_scene_string(const char *src, size_t len, Scene *scene)
{
	if (!src || !scene) return RT_ERR_ARGUMENT;
	Reader r = { src, len, 0, 1 };
	Object obj;
	memset(&obj, 0, sizeof(obj));
	scene->num_objects = 0;

	for (;;) {
		eat_blanks(&r);
		if (r.at >= r.len) break;

		if (keyword(&r, "sphere", 5)) {
			r.at += 6;
			obj.type = OBJECT_SPHERE;
			obj.sphere.center = (Vector3) {0, 0, 0};
			obj.sphere.radius = 1;
		} else if (keyword(&r, "cube", 3)) {
			r.at += 4;
			obj.type = OBJECT_CUBE;
			obj.cube.origin = (Vector3) {0, 0, 0};
			obj.cube.size   = (Vector3) {1, 1, 1};
		} else {
			fprintf(stderr, "Error: Invalid character (line %d)\n", r.line);
			return RT_ERR_FORMAT;
		}
		reset_material(&obj.material);

		for (;;) {
			eat_blanks(&r);
			int got = property(&r, &obj);
			if (got < 0) return RT_ERR_FORMAT;
			if (got == 0) break;
		}

		if (scene->num_objects == MAX_OBJECTS)   /* scene.c:602-605 */
			fprintf(stderr, "Warning: Ignoring object because the scene is too big (line %d)\n", r.line);
		else
			scene->objects[scene->num_objects++] = obj;
	}
	return RT_OK;
}

int rt_parse_scene_file(const char *file, Scene *scene)
{
	if (!file || !scene) return RT_ERR_ARGUMENT;
	FILE *fp = fopen(file, "rb");
	if (!fp) {
		fprintf(stderr, "Error: Couldn't open scene file\n");   /* scene.c:616 */
		return RT_ERR_IO;
	}
	fseek(fp, 0, SEEK_END);
	long size = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	if (size < 0) { fclose(fp); return RT_ERR_IO; }
	char *text = malloc((size_t) size + 1);
	if (!text) { fclose(fp); return RT_ERR_MEMORY; }
	size_t got = fread(text, 1, (size_t) size, fp);
	int bad = ferror(fp);
	fclose(fp);
	if (bad) { free(text); fprintf(stderr, "Error: Couldn't open scene file\n"); return RT_ERR_IO; }
	text[got] = '\0';
	int rc = rt_parse_scene_string(text, got, scene);     /* what was read, not what ftell promised */
	free(text);
	return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* camera                                                                                      */
/* ------------------------------------------------------------------------------------------ */

static Vector3 vmul(Vector3 v, float f) { return (Vector3) { v.x * f, v.y * f, v.z * f }; }

static Vector3 vcross(Vector3 u, Vector3 v)
{
	return (Vector3) { u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x };
}

static Vector3 vunit(Vector3 v)
{
	float len = (float) sqrt((double) (v.x * v.x + v.y * v.y + v.z * v.z));   /* vector.c:119-127 */
	if ((double) len < 0.00001 && (double) len > -0.00001) return v;
	return (Vector3) { v.x / len, v.y / len, v.z / len };
}

static Vector3 vmadd(Vector3 u, Vector3 v, float b)   /* combine(u, v, 1, b) */
{
	return (Vector3) { u.x * 1 + v.x * b, u.y * 1 + v.y * b, u.z * 1 + v.z * b };
}

void rt_camera_default(rt_camera *cam)
{
	if (!cam) return;
	cam->pos   = (Vector3) {5, 5, 5};
	cam->front = (Vector3) {-1, -1, -1};
	cam->up    = (Vector3) {0, 1, 0};
	cam->fov   = 30.0f;
}

void rt_camera_basis_for(const rt_camera *cam, float aspect_ratio, rt_camera_basis *out)
{
	Vector3 w = vunit(vmul(cam->front, -1));
	Vector3 u = vunit(vcross(cam->up, w));
	Vector3 v = vcross(w, u);
	float screen_h = 2 * tan(cam->fov / 2);           /* float/2 -> double tan -> float, camera.c:107 */
	float screen_w = aspect_ratio * screen_h;
	Vector3 H = vmul(u, screen_w);
	Vector3 V = vmul(v, screen_h);
	out->pos = cam->pos;
	out->horizontal = H;
	out->vertical = V;
	/* combine4(pos, H, V, w, 1, -0.5, -0.5, -1), summed left to right (camera.c:118) */
	out->lower_left_corner = (Vector3) {
		cam->pos.x * 1 + H.x * -0.5f + V.x * -0.5f + w.x * -1,
		cam->pos.y * 1 + H.y * -0.5f + V.y * -0.5f + w.y * -1,
		cam->pos.z * 1 + H.z * -0.5f + V.z * -0.5f + w.z * -1,
	};
}

void rt_mouse_state_default(rt_mouse_state *m)
{
	/* camera.c:23-27 */
	m->first_mouse = 1;
	m->yaw = -90.0f;
	m->pitch = 0.0f;
	m->last_x = 800.0f / 2.0;
	m->last_y = 600.0f / 2.0;
}

void rt_move_camera(rt_camera *cam, rt_direction dir, float speed)
{
	switch (dir) {
	case RT_DIR_UP:    cam->pos = vmadd(cam->pos, cam->front, +speed); break;
	case RT_DIR_DOWN:  cam->pos = vmadd(cam->pos, cam->front, -speed); break;
	case RT_DIR_LEFT:  cam->pos = vmadd(cam->pos, vunit(vcross(cam->front, cam->up)), -speed); break;
	case RT_DIR_RIGHT: cam->pos = vmadd(cam->pos, vunit(vcross(cam->front, cam->up)), +speed); break;
	}
}

static float to_radians(float deg) { return 3.14159265358979323846 * deg / 180; }   /* vector.c:94-97 */

void rt_rotate_camera(rt_camera *cam, rt_mouse_state *m, double mouse_x, double mouse_y)
{
	float x = mouse_x, y = mouse_y;
	if (m->first_mouse) { m->last_x = x; m->last_y = y; m->first_mouse = 0; }
	float dx = x - m->last_x;
	float dy = m->last_y - y;
	m->last_x = x; m->last_y = y;
	float sensitivity = 0.1f;
	dx *= sensitivity; dy *= sensitivity;
	m->yaw += dx; m->pitch += dy;
	if (m->pitch >  89.0f) m->pitch =  89.0f;
	if (m->pitch < -89.0f) m->pitch = -89.0f;
	float yr = to_radians(m->yaw), pr = to_radians(m->pitch);
	Vector3 f;
	f.x = cos(yr) * cos(pr);
	f.y = sin(pr);
	f.z = sin(yr) * cos(pr);
	cam->front = vunit(f);
}

uint64_t rt_path_seed(uint64_t seed, uint32_t pixel_index, uint32_t sample_index)
{
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t) sample_index << 32) | (uint64_t) pixel_index);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

/* ------------------------------------------------------------------------------------------ */
/* frame hand-off                                                                              */
/* ------------------------------------------------------------------------------------------ */

static rt_frame_sink g_sink = NULL;
static void *g_sink_user = NULL;

void rt_set_frame_sink(rt_frame_sink sink, void *user) { g_sink = sink; g_sink_user = user; }

void rt_move_frame_to_the_gpu(int w, int h, Vector3 *data)
{
	if (g_sink) g_sink(w, h, data, g_sink_user);
}

int rt_write_ppm(const char *file, int w, int h, const Vector3 *data)
{
	if (!file || !data || w < 1 || h < 1) return RT_ERR_ARGUMENT;
	FILE *fp = fopen(file, "wb");
	if (!fp) return RT_ERR_IO;
	fprintf(fp, "P6\n%d %d\n255\n", w, h);
	unsigned char *row = malloc((size_t) w * 3);
	if (!row) { fclose(fp); return RT_ERR_MEMORY; }
	for (int j = h - 1; j >= 0; j--) {                 /* stbi_flip_vertically_on_write(1), main.c:672 */
		for (int i = 0; i < w; i++) {
			const Vector3 *p = &data[(size_t) j * w + i];
			row[3*i+0] = (unsigned char) (p->x * 255);    /* truncation, main.c:662-664 */
			row[3*i+1] = (unsigned char) (p->y * 255);
			row[3*i+2] = (unsigned char) (p->z * 255);
		}
		fwrite(row, 1, (size_t) w * 3, fp);
	}
	free(row);
	fclose(fp);
	return RT_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* PNG sink -- screenshot() main.c:637-681                                                     */
/* ------------------------------------------------------------------------------------------ */
/* The reference hands an RGB8 buffer to stbi_write_png with a vertical flip.  The container here
 * is a plain PNG with stored (uncompressed) deflate blocks: every decoder yields the same pixels
 * stb's compressed stream would. */

static uint32_t crc_table[256];

static void crc_init(void)
{
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
		crc_table[n] = c;
	}
}

static uint32_t crc_update(uint32_t c, const unsigned char *p, size_t n)
{
	for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
	return c;
}

static void put32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

static int chunk(FILE *fp, const char *tag, const unsigned char *data, size_t n)
{
	unsigned char head[8];
	put32(head, (uint32_t) n);
	memcpy(head + 4, tag, 4);
	uint32_t c = crc_update(0xffffffffu, head + 4, 4);
	c = crc_update(c, data, n) ^ 0xffffffffu;
	unsigned char tail[4];
	put32(tail, c);
	return fwrite(head, 1, 8, fp) == 8 && fwrite(data, 1, n, fp) == n && fwrite(tail, 1, 4, fp) == 4;
}

/* ---- deflate (RFC 1951) with the fixed Huffman code and a hash-chain LZ77 matcher: what a screenshot needs
 * (stb_image_write, which the reference uses, compresses the same way; file BYTES differ, decoded pixels do not) */

typedef struct { unsigned char *p; size_t n, cap; uint32_t bits; int nbits; int failed; } BitSink;

static void sink_byte(BitSink *s, unsigned v)
{
	if (s->n == s->cap) {
		size_t cap = s->cap ? s->cap * 2 : 1 << 16;
		unsigned char *q = realloc(s->p, cap);
		if (!q) { s->failed = 1; return; }
		s->p = q; s->cap = cap;
	}
	s->p[s->n++] = (unsigned char) v;
}

static void sink_bits(BitSink *s, uint32_t v, int n)        /* LSB first */
{
	s->bits |= v << s->nbits; s->nbits += n;
	while (s->nbits >= 8) { sink_byte(s, s->bits & 0xff); s->bits >>= 8; s->nbits -= 8; }
}

static uint32_t reverse_bits(uint32_t v, int n) { uint32_t r = 0; while (n--) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

static void sink_huff(BitSink *s, uint32_t code, int n) { sink_bits(s, reverse_bits(code, n), n); }   /* Huffman codes go MSB first */

static void sink_litlen(BitSink *s, int sym)               /* fixed code, RFC 1951 3.2.6 */
{
	if      (sym < 144) sink_huff(s, 0x30 + sym, 8);
	else if (sym < 256) sink_huff(s, 0x190 + (sym - 144), 9);
	else if (sym < 280) sink_huff(s, sym - 256, 7);
	else                sink_huff(s, 0xc0 + (sym - 280), 8);
}

static unsigned char *deflate_fixed(const unsigned char *in, size_t n, size_t *out_n)
{
	static const unsigned short len_base[] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258, 259 };
	static const unsigned char  len_extra[] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
	static const unsigned short dist_base[] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577, 32769 };
	static const unsigned char  dist_extra[] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };
	enum { HASH_BITS = 15, HASH_SIZE = 1 << HASH_BITS, WINDOW = 32768, CHAIN = 24 };
	BitSink s = { 0 };
	int32_t *head = malloc(sizeof(int32_t) * HASH_SIZE), *prev = malloc(sizeof(int32_t) * WINDOW);
	if (!head || !prev) { free(head); free(prev); return NULL; }
	for (int i = 0; i < HASH_SIZE; i++) head[i] = -1;
	sink_byte(&s, 0x78); sink_byte(&s, 0x5e);              /* zlib header: deflate, 32 KiB window */
	sink_bits(&s, 1, 1); sink_bits(&s, 1, 2);               /* one final block, fixed Huffman */
	uint32_t a = 1, b = 0;
	size_t i = 0;
	while (i < n) {
		int best_len = 0, best_dist = 0;
		if (i + 3 <= n) {
			const uint32_t h = ((uint32_t) in[i] << 10 ^ (uint32_t) in[i + 1] << 5 ^ in[i + 2]) & (HASH_SIZE - 1);
			int32_t cand = head[h];
			const size_t max_len = n - i < 258 ? n - i : 258;
			for (int tries = 0; cand >= 0 && (size_t) cand + WINDOW > i && tries < CHAIN; tries++) {
				size_t l = 0;
				while (l < max_len && in[(size_t) cand + l] == in[i + l]) l++;
				if ((int) l > best_len) { best_len = (int) l; best_dist = (int) (i - (size_t) cand); if (l == max_len) break; }
				cand = prev[(size_t) cand & (WINDOW - 1)];
			}
			prev[i & (WINDOW - 1)] = head[h]; head[h] = (int32_t) i;
		}
		if (best_len >= 3) {
			int lc = 0; while (best_len >= len_base[lc + 1]) lc++;
			sink_litlen(&s, 257 + lc);
			if (len_extra[lc]) sink_bits(&s, (uint32_t) (best_len - len_base[lc]), len_extra[lc]);
			int dc = 0; while (best_dist >= dist_base[dc + 1]) dc++;
			sink_huff(&s, (uint32_t) dc, 5);
			if (dist_extra[dc]) sink_bits(&s, (uint32_t) (best_dist - dist_base[dc]), dist_extra[dc]);
			for (int k = 1; k < best_len; k++) {               /* the skipped positions still enter the dictionary */
				const size_t q = i + (size_t) k;
				if (q + 3 <= n) {
					const uint32_t h = ((uint32_t) in[q] << 10 ^ (uint32_t) in[q + 1] << 5 ^ in[q + 2]) & (HASH_SIZE - 1);
					prev[q & (WINDOW - 1)] = head[h]; head[h] = (int32_t) q;
				}
			}
			i += (size_t) best_len;
		} else
			sink_litlen(&s, in[i++]);
	}
	sink_litlen(&s, 256);                                   /* end of block */
	if (s.nbits) sink_bits(&s, 0, 8 - s.nbits);
	for (size_t k = 0; k < n; k++) { a += in[k]; b += a; if ((k & 4095) == 4095) { a %= 65521u; b %= 65521u; } }
	a %= 65521u; b %= 65521u;                               /* adler-32 of the uncompressed bytes */
	sink_byte(&s, b >> 8); sink_byte(&s, b & 0xff); sink_byte(&s, a >> 8); sink_byte(&s, a & 0xff);
	free(head); free(prev);
	if (s.failed) { free(s.p); return NULL; }
	*out_n = s.n;
	return s.p;
}

static int paeth(int a, int b, int c)
{
	const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	return pa <= pb && pa <= pc ? a : (pb <= pc ? b : c);
}

int rt_write_png(const char *file, int w, int h, const Vector3 *data)
{
	if (!file || !data || w < 1 || h < 1) return RT_ERR_ARGUMENT;
	if (!crc_table[1]) crc_init();
	const size_t row_bytes = (size_t) w * 3, stride = row_bytes + 1, raw_n = stride * (size_t) h;
	unsigned char *raw = malloc(raw_n), *pix = malloc(row_bytes * 2);
	if (!raw || !pix) { free(raw); free(pix); return RT_ERR_MEMORY; }
	unsigned char *cur = pix, *up = pix + row_bytes;
	memset(up, 0, row_bytes);
	for (int row = 0; row < h; row++) {
		const Vector3 *src = data + (size_t) (h - 1 - row) * w;        /* stbi_flip_vertically_on_write(1) */
		for (int i = 0; i < w; i++) {
			cur[3 * i + 0] = (unsigned char) (src[i].x * 255);             /* main.c:662-664: truncation */
			cur[3 * i + 1] = (unsigned char) (src[i].y * 255);
			cur[3 * i + 2] = (unsigned char) (src[i].z * 255);
		}
		unsigned char *dst = raw + stride * (size_t) row;
		*dst++ = 4;                                                    /* filter: Paeth */
		for (size_t k = 0; k < row_bytes; k++) {
			const int left = k >= 3 ? cur[k - 3] : 0, upleft = k >= 3 ? up[k - 3] : 0;
			dst[k] = (unsigned char) (cur[k] - paeth(left, up[k], upleft));
		}
		unsigned char *t = cur; cur = up; up = t;
	}
	size_t o = 0;
	unsigned char *z = deflate_fixed(raw, raw_n, &o);
	free(raw); free(pix);
	if (!z) return RT_ERR_MEMORY;

	FILE *fp = fopen(file, "wb");
	int ok = fp != NULL;
	if (ok) {
		static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
		unsigned char ihdr[13];
		put32(ihdr, (uint32_t) w); put32(ihdr + 4, (uint32_t) h);
		ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
		ok = fwrite(sig, 1, 8, fp) == 8 && chunk(fp, "IHDR", ihdr, 13) && chunk(fp, "IDAT", z, o) && chunk(fp, "IEND", NULL, 0);
		ok = (fclose(fp) == 0) && ok;
	}
	free(z);
	return ok ? RT_OK : RT_ERR_IO;
}

/* main.c:642-659: first free "screenshot_<n>.png", n in [0, 1000) */
int rt_screenshot(int w, int h, const Vector3 *data, char *name_out, size_t name_cap)
{
	char file[64];
	int i = 0;
	for (; i < 1000; i++) {
		snprintf(file, sizeof(file), "screenshot_%d.png", i);
		FILE *probe = fopen(file, "rb");
		if (!probe) break;
		fclose(probe);
	}
	if (i == 1000) return RT_ERR_IO;
	int rc = rt_write_png(file, w, h, data);
	if (rc == RT_OK) {
		fprintf(stderr, "Took screenshot! (%s)\n", file);
		if (name_out && name_cap) snprintf(name_out, name_cap, "%s", file);
	} else
		fprintf(stderr, "Could not take screenshot (write error)\n");
	return rc;
}
