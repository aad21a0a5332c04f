/*
 * rt_jpeg.c -- baseline JPEG decoder whose output bytes equal stb_image v2.29's.
 *
 * Why it exists: the reference decodes its skybox with the stb_image.h v2.29 it vendors
 * (gpu_and_windowing.c:5-6,27) and sample_cubemap() reads those bytes (gpu_and_windowing.c:106).
 * JPEG decoders differ in their inverse DCT, chroma upsampling and colour conversion, so "the
 * skybox" is defined by that decoder's integer arithmetic (libjpeg differs in 1-3 % of the bytes).
 * This file restates stb_image v2.29's published (public-domain) algorithm for the baseline, Huffman,
 * 8-bit path.
 *
 * PROVENANCE.  stb_image.h is third-party code (Sean Barrett et al., public domain / MIT, v2.29), not the
 * reference's own; the reference vendors it unmodified under 3p/stb.  The fixed-point IDCT butterfly below keeps
 * stb's temporaries and statement order (STBI__IDCT_1D, stb_image.h:2429-2464), and the resampler and YCbCr rows follow
 * stb__resample_row_* / stbi__YCbCr_to_RGB_row -- a deliberate restatement of that public-domain arithmetic, kept
 * recognisable so that it can be checked against the original line by line.  What byte-identical texels require is
 * the arithmetic -- the constants, the rounding offsets, the shifts and where values are truncated to 16 or 8 bits --
 * not the order of the (commutative, overflow-free) integer additions.  A host that prefers to may link stb_image.h
 * itself and hand its bytes to rt_set_skybox(): the Cubemap struct is the seam.
 *
 * What is restated:
 *   - entropy decoding per ITU T.81 annex F (any conforming decoder yields the same coefficients),
 *     each coefficient multiplied by its quantiser and kept as int16;
 *   - 2-D IDCT: column pass then row pass of the 12-bit fixed-point LLM butterfly
 *     (constants = round(x * 4096)), column results >> 10 after +512, row results >> 17 after
 *     +65536 + (128 << 17), clamped to 0..255; an all-zero-AC column short-circuits to dc*4;
 *   - chroma upsampling with the 3:1 / 9:3:3:1 triangle filters ("fancy upsampling");
 *   - YCbCr -> RGB in 20-bit fixed point with constants round(x * 4096) << 8, the Cb->G term masked
 *     with 0xffff0000.
 * Pinned by tests/test_host_mirror.py (test_jpeg_decoder_matches_stb_decode_of_reference) against the SHA-256 of the reference's decoded faces
 * (tests/golden/reference_meta.json) and texel probes.
 *
 * Not supported (RT_ERR_FORMAT): progressive / arithmetic / 12-bit / CMYK streams -- the reference's
 * assets are baseline 4:2:0 JFIF.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rt_hip.h"

typedef struct {
	uint8_t  size[257];
	uint16_t code[256];
	uint8_t  value[256];
	int      maxcode[18];     /* left-justified to 16 bits, exclusive */
	int      delta[17];
	int      count;
	/* 9-bit lookahead: entry = symbol index, 255 = longer code */
	uint8_t  fast[512];
	int      present;
} Huff;

typedef struct {
	int id, h, v, tq;
	int td, ta;
	int dc_pred;
	int x, y, w2, h2;
	uint8_t *plane;
	uint8_t *linebuf;
} Comp;

typedef struct {
	const uint8_t *p, *end;
	uint32_t bitbuf;
	int      bitcnt;
	int      marker;          /* pending marker found in the entropy stream, or 0 */
	int      nomore;

	int W, H, ncomp;
	Comp comp[4];
	int hmax, vmax, mcu_w, mcu_h, mcu_x, mcu_y;
	uint16_t quant[4][64];
	Huff dc[4], ac[4];
	int restart_interval, todo;
	int jfif, adobe_transform;
	int scan_n, order[4];
} Jpeg;

static const uint8_t ZIGZAG[64 + 15] = {
	0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
	28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
	54, 47, 55, 62, 63,
	63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63
};

/* ---- Huffman tables (T.81 annex C) ------------------------------------------------------------ */

static int huff_build(Huff *h, const uint8_t counts[16])
{
	int k = 0;
	for (int len = 1; len <= 16; len++)
		for (int i = 0; i < counts[len - 1]; i++) {
			if (k >= 256) return 0;
			h->size[k++] = (uint8_t) len;
		}
	h->size[k] = 0;
	h->count = k;

	int code = 0;
	k = 0;
	for (int len = 1; len <= 16; len++) {
		h->delta[len] = k - code;
		if (h->size[k] == len) {
			while (h->size[k] == len) h->code[k++] = (uint16_t) code++;
			if (code - 1 >= (1 << len)) return 0;
		}
		h->maxcode[len] = code << (16 - len);
		code <<= 1;
	}
	h->maxcode[17] = 0x7fffffff;

	memset(h->fast, 255, sizeof(h->fast));
	for (int i = 0; i < k; i++) {
		int len = h->size[i];
		if (len <= 9) {
			int first = h->code[i] << (9 - len);
			int n = 1 << (9 - len);
			for (int j = 0; j < n; j++) h->fast[first + j] = (uint8_t) i;
		}
	}
	h->present = 1;
	return 1;
}

/* ---- bit reader with 0xFF00 unstuffing --------------------------------------------------------- */

static void refill(Jpeg *j)
{
	while (j->bitcnt <= 24) {
		int b = 0;
		if (!j->nomore) {
			if (j->p < j->end) {
				b = *j->p++;
				if (b == 0xff) {
					int c = j->p < j->end ? *j->p++ : 0;
					while (c == 0xff) c = j->p < j->end ? *j->p++ : 0;
					if (c != 0) { j->marker = c; j->nomore = 1; b = 0; }
				}
			} else
				j->nomore = 1;
		}
		j->bitbuf |= (uint32_t) b << (24 - j->bitcnt);
		j->bitcnt += 8;
	}
}

static int huff_symbol(Jpeg *j, const Huff *h)
{
	if (j->bitcnt < 16) refill(j);
	int look = (int) (j->bitbuf >> 23) & 511;
	int idx = h->fast[look];
	if (idx < 255) {
		int len = h->size[idx];
		if (len > j->bitcnt) return -1;
		j->bitbuf <<= len; j->bitcnt -= len;
		return h->value[idx];
	}
	int top = (int) (j->bitbuf >> 16);
	int len = 10;
	while (top >= h->maxcode[len]) len++;
	if (len >= 17 || len > j->bitcnt) return -1;
	idx = (int) ((j->bitbuf >> (32 - len)) & ((1u << len) - 1)) + h->delta[len];
	if (idx < 0 || idx >= h->count) return -1;
	j->bitbuf <<= len; j->bitcnt -= len;
	return h->value[idx];
}

/* T.81 F.2.2.1 RECEIVE + EXTEND */
static int receive_extend(Jpeg *j, int nbits)
{
	if (nbits == 0) return 0;
	if (j->bitcnt < nbits) refill(j);
	if (j->bitcnt < nbits) return 0;
	int v = (int) (j->bitbuf >> (32 - nbits));
	j->bitbuf <<= nbits; j->bitcnt -= nbits;
	if (v < (1 << (nbits - 1))) v += (int) ((~0u) << nbits) + 1;
	return v;
}

/* one 8x8 block of quantised coefficients -> dequantised int16 in natural order */
static int decode_block(Jpeg *j, Comp *c, short data[64])
{
	const Huff *hdc = &j->dc[c->td], *hac = &j->ac[c->ta];
	const uint16_t *q = j->quant[c->tq];
	memset(data, 0, 64 * sizeof(short));

	int t = huff_symbol(j, hdc);
	if (t < 0 || t > 15) return 0;
	int diff = t ? receive_extend(j, t) : 0;
	c->dc_pred += diff;
	data[0] = (short) (c->dc_pred * q[0]);

	int k = 1;
	while (k < 64) {
		int rs = huff_symbol(j, hac);
		if (rs < 0) return 0;
		int s = rs & 15, r = rs >> 4;
		if (s == 0) {
			if (rs != 0xf0) break;
			k += 16;
		} else {
			k += r;
			/* a run that overshoots the block (corrupt stream): stb_image's dezigzag table is padded with 63s,
			 * so the coefficient lands on position 63; do the same instead of indexing past the tables */
			const int kk = k > 63 ? 63 : k;
			data[ZIGZAG[kk]] = (short) (receive_extend(j, s) * q[kk]);
			k++;
		}
	}
	return 1;
}

/* ---- inverse DCT ----------------------------------------------------------------------------- */

#define FIX(x)  ((int) ((x) * 4096 + 0.5))

/* one 8-point butterfly; outputs are the even part sums x0..x3 and the odd part t0..t3 */
#define BUTTERFLY(s0, s1, s2, s3, s4, s5, s6, s7)                       \
	int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;             \
	p2 = s2; p3 = s6;                                                   \
	p1 = (p2 + p3) * FIX(0.5411961f);                                   \
	t2 = p1 + p3 * FIX(-1.847759065f);                                  \
	t3 = p1 + p2 * FIX(0.765366865f);                                   \
	p2 = s0; p3 = s4;                                                   \
	t0 = (p2 + p3) * 4096;                                              \
	t1 = (p2 - p3) * 4096;                                              \
	x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;             \
	t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                 \
	p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;             \
	p5 = (p3 + p4) * FIX(1.175875602f);                                 \
	t0 = t0 * FIX(0.298631336f);                                        \
	t1 = t1 * FIX(2.053119869f);                                        \
	t2 = t2 * FIX(3.072711026f);                                        \
	t3 = t3 * FIX(1.501321110f);                                        \
	p1 = p5 + p1 * FIX(-0.899976223f);                                  \
	p2 = p5 + p2 * FIX(-2.562915447f);                                  \
	p3 = p3 * FIX(-1.961570560f);                                       \
	p4 = p4 * FIX(-0.390180644f);                                       \
	t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

static uint8_t clamp255(int x)
{
	if ((unsigned) x > 255) return x < 0 ? 0 : 255;
	return (uint8_t) x;
}

static void idct_8x8(uint8_t *out, int stride, const short in[64])
{
	int tmp[64];
	for (int c = 0; c < 8; c++) {
		const short *d = in + c;
		int *v = tmp + c;
		if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
			int dc = d[0] * 4;
			v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
		} else {
			BUTTERFLY(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
			x0 += 512; x1 += 512; x2 += 512; x3 += 512;
			v[0]  = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
			v[8]  = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
			v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
			v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
		}
	}
	for (int r = 0; r < 8; r++) {
		const int *v = tmp + 8 * r;
		uint8_t *o = out + (size_t) r * stride;
		BUTTERFLY(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
		const int bias = 65536 + (128 << 17);
		x0 += bias; x1 += bias; x2 += bias; x3 += bias;
		o[0] = clamp255((x0 + t3) >> 17); o[7] = clamp255((x0 - t3) >> 17);
		o[1] = clamp255((x1 + t2) >> 17); o[6] = clamp255((x1 - t2) >> 17);
		o[2] = clamp255((x2 + t1) >> 17); o[5] = clamp255((x2 - t1) >> 17);
		o[3] = clamp255((x3 + t0) >> 17); o[4] = clamp255((x3 - t0) >> 17);
	}
}

/* ---- marker segments --------------------------------------------------------------------------- */

static int get8(Jpeg *j)  { return j->p < j->end ? *j->p++ : 0; }
static int get16(Jpeg *j) { int a = get8(j); return (a << 8) | get8(j); }

static int next_marker(Jpeg *j)
{
	if (j->marker) { int m = j->marker; j->marker = 0; return m; }
	int c = get8(j);
	if (c != 0xff) return -1;
	while (c == 0xff) c = get8(j);
	return c;
}

static int read_segment(Jpeg *j, int m)
{
	switch (m) {
	case 0xdd: /* DRI */
		if (get16(j) != 4) return 0;
		j->restart_interval = get16(j);
		return 1;
	case 0xdb: { /* DQT */
		int len = get16(j) - 2;
		while (len > 0) {
			int pq = get8(j);
			int prec = pq >> 4, t = pq & 15;
			if ((prec != 0 && prec != 1) || t > 3) return 0;
			for (int i = 0; i < 64; i++)
				j->quant[t][i] = (uint16_t) (prec ? get16(j) : get8(j));   /* kept in zig-zag order */
			len -= prec ? 129 : 65;
		}
		return len == 0;
	}
	case 0xc4: { /* DHT */
		int len = get16(j) - 2;
		while (len > 0) {
			int tc_th = get8(j);
			int tc = tc_th >> 4, th = tc_th & 15;
			if (tc > 1 || th > 3) return 0;
			uint8_t counts[16];
			int n = 0;
			for (int i = 0; i < 16; i++) { counts[i] = (uint8_t) get8(j); n += counts[i]; }
			if (n > 256) return 0;
			Huff *h = tc ? &j->ac[th] : &j->dc[th];
			if (!huff_build(h, counts)) return 0;
			for (int i = 0; i < n; i++) h->value[i] = (uint8_t) get8(j);
			len -= 17 + n;
		}
		return len == 0;
	}
	}
	if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
		int len = get16(j);
		if (len < 2) return 0;
		len -= 2;
		if (m == 0xe0 && len >= 5) {
			static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
			int ok = 1;
			for (int i = 0; i < 5; i++) if (get8(j) != tag[i]) ok = 0;
			len -= 5;
			if (ok) j->jfif = 1;
		} else if (m == 0xee && len >= 12) {
			static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
			int ok = 1;
			for (int i = 0; i < 6; i++) if (get8(j) != tag[i]) ok = 0;
			len -= 6;
			if (ok) { get8(j); get16(j); get16(j); j->adobe_transform = get8(j); len -= 6; }
		}
		if (j->p + len > j->end) return 0;
		j->p += len;
		return 1;
	}
	return 0;
}

static int read_frame_header(Jpeg *j)
{
	int len = get16(j);
	if (len < 11) return 0;
	if (get8(j) != 8) return 0;                        /* 8-bit samples only */
	j->H = get16(j); j->W = get16(j);
	if (j->H <= 0 || j->W <= 0) return 0;
	j->ncomp = get8(j);
	if (j->ncomp != 1 && j->ncomp != 3) return 0;
	if (len != 8 + 3 * j->ncomp) return 0;
	j->hmax = j->vmax = 1;
	for (int i = 0; i < j->ncomp; i++) {
		Comp *c = &j->comp[i];
		c->id = get8(j);
		int hv = get8(j);
		c->h = hv >> 4; c->v = hv & 15;
		if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4) return 0;
		c->tq = get8(j);
		if (c->tq > 3) return 0;
		if (c->h > j->hmax) j->hmax = c->h;
		if (c->v > j->vmax) j->vmax = c->v;
	}
	for (int i = 0; i < j->ncomp; i++)
		if (j->hmax % j->comp[i].h || j->vmax % j->comp[i].v) return 0;
	j->mcu_w = j->hmax * 8; j->mcu_h = j->vmax * 8;
	j->mcu_x = (j->W + j->mcu_w - 1) / j->mcu_w;
	j->mcu_y = (j->H + j->mcu_h - 1) / j->mcu_h;
	for (int i = 0; i < j->ncomp; i++) {
		Comp *c = &j->comp[i];
		c->x = (j->W * c->h + j->hmax - 1) / j->hmax;
		c->y = (j->H * c->v + j->vmax - 1) / j->vmax;
		c->w2 = j->mcu_x * c->h * 8;
		c->h2 = j->mcu_y * c->v * 8;
		c->plane = malloc((size_t) c->w2 * c->h2 + 15);
		c->linebuf = malloc((size_t) j->W + 3);
		if (!c->plane || !c->linebuf) return 0;
	}
	return 1;
}

static int read_scan_header(Jpeg *j)
{
	int len = get16(j);
	j->scan_n = get8(j);
	if (j->scan_n < 1 || j->scan_n > 4 || j->scan_n > j->ncomp) return 0;
	if (len != 6 + 2 * j->scan_n) return 0;
	for (int i = 0; i < j->scan_n; i++) {
		int id = get8(j), tables = get8(j), which;
		for (which = 0; which < j->ncomp; which++)
			if (j->comp[which].id == id) break;
		if (which == j->ncomp) return 0;
		j->comp[which].td = tables >> 4;
		j->comp[which].ta = tables & 15;
		if (j->comp[which].td > 3 || j->comp[which].ta > 3) return 0;
		if (!j->dc[j->comp[which].td].present || !j->ac[j->comp[which].ta].present) return 0;
		j->order[i] = which;
	}
	int ss = get8(j), se = get8(j), ahl = get8(j);
	if (ss != 0 || se != 63 || ahl != 0) return 0;     /* baseline only */
	return 1;
}

static void entropy_reset(Jpeg *j)
{
	j->bitbuf = 0; j->bitcnt = 0; j->nomore = 0; j->marker = 0;
	for (int i = 0; i < 4; i++) j->comp[i].dc_pred = 0;
	j->todo = j->restart_interval ? j->restart_interval : 0x7fffffff;
}

static int restart_if_due(Jpeg *j)
{
	if (--j->todo > 0) return 1;
	if (j->bitcnt < 24) refill(j);
	if (!(j->marker >= 0xd0 && j->marker <= 0xd7)) return 2;   /* no RST: scan ends here */
	entropy_reset(j);
	return 1;
}

static int decode_scan(Jpeg *j)
{
	short block[64];
	entropy_reset(j);
	if (j->scan_n == 1) {
		Comp *c = &j->comp[j->order[0]];
		int bw = (c->x + 7) >> 3, bh = (c->y + 7) >> 3;
		for (int by = 0; by < bh; by++)
			for (int bx = 0; bx < bw; bx++) {
				if (!decode_block(j, c, block)) return 0;
				idct_8x8(c->plane + (size_t) c->w2 * by * 8 + bx * 8, c->w2, block);
				int r = restart_if_due(j);
				if (r == 2) return 1;
			}
		return 1;
	}
	for (int my = 0; my < j->mcu_y; my++)
		for (int mx = 0; mx < j->mcu_x; mx++) {
			for (int k = 0; k < j->scan_n; k++) {
				Comp *c = &j->comp[j->order[k]];
				for (int y = 0; y < c->v; y++)
					for (int x = 0; x < c->h; x++) {
						int x2 = (mx * c->h + x) * 8, y2 = (my * c->v + y) * 8;
						if (!decode_block(j, c, block)) return 0;
						idct_8x8(c->plane + (size_t) c->w2 * y2 + x2, c->w2, block);
					}
			}
			int r = restart_if_due(j);
			if (r == 2) return 1;
		}
	return 1;
}

/* ---- chroma upsampling (one output row) --------------------------------------------------------- */

typedef uint8_t *(*RowFilter)(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs);

static uint8_t *row_copy(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs)
{
	(void) out; (void) far; (void) w; (void) hs;
	return near;
}

static uint8_t *row_v2(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs)
{
	(void) hs;
	for (int i = 0; i < w; i++) out[i] = (uint8_t) ((3 * near[i] + far[i] + 2) >> 2);
	return out;
}

static uint8_t *row_h2(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs)
{
	(void) far; (void) hs;
	const uint8_t *in = near;
	if (w == 1) { out[0] = out[1] = in[0]; return out; }
	out[0] = in[0];
	out[1] = (uint8_t) ((in[0] * 3 + in[1] + 2) >> 2);
	int i;
	for (i = 1; i < w - 1; i++) {
		int n = 3 * in[i] + 2;
		out[2*i]   = (uint8_t) ((n + in[i-1]) >> 2);
		out[2*i+1] = (uint8_t) ((n + in[i+1]) >> 2);
	}
	out[2*i]   = (uint8_t) ((in[w-2] * 3 + in[w-1] + 2) >> 2);
	out[2*i+1] = in[w-1];
	return out;
}

static uint8_t *row_hv2(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs)
{
	(void) hs;
	if (w == 1) { out[0] = out[1] = (uint8_t) ((3 * near[0] + far[0] + 2) >> 2); return out; }
	int prev, cur = 3 * near[0] + far[0];
	out[0] = (uint8_t) ((cur + 2) >> 2);
	for (int i = 1; i < w; i++) {
		prev = cur;
		cur = 3 * near[i] + far[i];
		out[2*i-1] = (uint8_t) ((3 * prev + cur + 8) >> 4);
		out[2*i]   = (uint8_t) ((3 * cur + prev + 8) >> 4);
	}
	out[2*w-1] = (uint8_t) ((cur + 2) >> 2);
	return out;
}

static uint8_t *row_repeat(uint8_t *out, uint8_t *near, uint8_t *far, int w, int hs)
{
	(void) far;
	for (int i = 0; i < w; i++)
		for (int k = 0; k < hs; k++) out[i * hs + k] = near[i];
	return out;
}

/* ---- colour conversion ------------------------------------------------------------------------- */

#define CFIX(x)  (((int) ((x) * 4096.0f + 0.5f)) << 8)

static void ycc_row(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int n)
{
	for (int i = 0; i < n; i++) {
		int yf = (y[i] << 20) + (1 << 19);
		int cr = pcr[i] - 128, cb = pcb[i] - 128;
		int r = yf + cr * CFIX(1.40200f);
		int g = yf + (cr * -CFIX(0.71414f)) + ((cb * -CFIX(0.34414f)) & 0xffff0000);
		int b = yf + cb * CFIX(1.77200f);
		out[3*i]   = clamp255(r >> 20);
		out[3*i+1] = clamp255(g >> 20);
		out[3*i+2] = clamp255(b >> 20);
	}
}

/* ---- driver ------------------------------------------------------------------------------------ */

static void jpeg_release(Jpeg *j)
{
	for (int i = 0; i < 4; i++) { free(j->comp[i].plane); free(j->comp[i].linebuf); }
}

static int decode_memory(const uint8_t *buf, size_t len, uint8_t **out, int *w, int *h, int *chan)
{
	Jpeg *j = calloc(1, sizeof(Jpeg));
	if (!j) return RT_ERR_MEMORY;
	j->p = buf; j->end = buf + len;
	j->adobe_transform = -1;
	int rc = RT_ERR_FORMAT, have_frame = 0, have_scan = 0;

	if (get8(j) != 0xff || get8(j) != 0xd8) goto done;          /* SOI */
	for (;;) {
		int m = next_marker(j);
		if (m < 0) goto done;
		if (m == 0xd9) break;                                    /* EOI */
		if (m == 0xc0 || m == 0xc1) {                            /* SOF0 / SOF1 (Huffman, sequential) */
			if (have_frame || !read_frame_header(j)) goto done;
			have_frame = 1;
		} else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
			fprintf(stderr, "rt_jpeg: only baseline Huffman JPEG is supported (SOF marker 0x%02x)\n", m);
			goto done;
		} else if (m == 0xda) {                                  /* SOS */
			if (!have_frame || !read_scan_header(j) || !decode_scan(j)) goto done;
			have_scan = 1;
			if (j->marker == 0) {
				/* skip to the next marker, as a decoder must after a scan */
				while (j->p < j->end) {
					if (*j->p == 0xff && j->p + 1 < j->end && j->p[1] != 0 && j->p[1] != 0xff) break;
					j->p++;
				}
			}
		} else if (m >= 0xd0 && m <= 0xd7) {
			/* stray RST */
		} else if (!read_segment(j, m))
			goto done;
		if (j->p >= j->end && !j->marker) break;
	}
	if (!have_scan) goto done;

	{
		int n = j->ncomp;
		int rgb_ids = n == 3 && j->comp[0].id == 'R' && j->comp[1].id == 'G' && j->comp[2].id == 'B';
		int is_rgb = n == 3 && (rgb_ids || (j->adobe_transform == 0 && !j->jfif));
		if (is_rgb) { fprintf(stderr, "rt_jpeg: RGB-coded JPEG unsupported\n"); goto done; }
		uint8_t *img = malloc((size_t) j->W * j->H * n + 1);
		if (!img) { rc = RT_ERR_MEMORY; goto done; }

		struct { RowFilter f; uint8_t *line0, *line1; int hs, vs, w_lores, ystep, ypos; } rs[3];
		for (int k = 0; k < n; k++) {
			Comp *c = &j->comp[k];
			rs[k].hs = j->hmax / c->h; rs[k].vs = j->vmax / c->v;
			rs[k].ystep = rs[k].vs >> 1;
			rs[k].w_lores = (j->W + rs[k].hs - 1) / rs[k].hs;
			rs[k].ypos = 0;
			rs[k].line0 = rs[k].line1 = c->plane;
			if      (rs[k].hs == 1 && rs[k].vs == 1) rs[k].f = row_copy;
			else if (rs[k].hs == 1 && rs[k].vs == 2) rs[k].f = row_v2;
			else if (rs[k].hs == 2 && rs[k].vs == 1) rs[k].f = row_h2;
			else if (rs[k].hs == 2 && rs[k].vs == 2) rs[k].f = row_hv2;
			else                                     rs[k].f = row_repeat;
		}
		for (int row = 0; row < j->H; row++) {
			uint8_t *line[3];
			for (int k = 0; k < n; k++) {
				Comp *c = &j->comp[k];
				int bottom = rs[k].ystep >= (rs[k].vs >> 1);
				line[k] = rs[k].f(c->linebuf, bottom ? rs[k].line1 : rs[k].line0,
				                  bottom ? rs[k].line0 : rs[k].line1, rs[k].w_lores, rs[k].hs);
				if (++rs[k].ystep >= rs[k].vs) {
					rs[k].ystep = 0;
					rs[k].line0 = rs[k].line1;
					if (++rs[k].ypos < c->y) rs[k].line1 += c->w2;
				}
			}
			uint8_t *dst = img + (size_t) row * j->W * n;
			if (n == 3) ycc_row(dst, line[0], line[1], line[2], j->W);
			else        memcpy(dst, line[0], (size_t) j->W);
		}
		*out = img; *w = j->W; *h = j->H; *chan = n;
		rc = RT_OK;
	}
done:
	jpeg_release(j);
	free(j);
	return rc;
}

int rt_decode_jpeg_file(const char *file, uint8_t **out, int *w, int *h, int *chan)
{
	if (!file || !out || !w || !h || !chan) return RT_ERR_ARGUMENT;
	*out = NULL;
	FILE *fp = fopen(file, "rb");
	if (!fp) return RT_ERR_IO;
	fseek(fp, 0, SEEK_END);
	long size = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	if (size <= 0) { fclose(fp); return RT_ERR_IO; }
	uint8_t *buf = malloc((size_t) size);
	if (!buf) { fclose(fp); return RT_ERR_MEMORY; }
	size_t got = fread(buf, 1, (size_t) size, fp);
	fclose(fp);
	int rc = decode_memory(buf, got, out, w, h, chan);
	free(buf);
	return rc;
}

/* gpu_and_windowing.c:24-33: the w/h/chan of the LAST face win, as in the reference */
int rt_load_cubemap(Cubemap *c, const char *files[6])
{
	if (!c || !files) return RT_ERR_ARGUMENT;
	memset(c, 0, sizeof(*c));
	for (int i = 0; i < 6; i++) {
		int rc = rt_decode_jpeg_file(files[i], &c->data[i], &c->w, &c->h, &c->chan);
		if (rc != RT_OK) {
			fprintf(stderr, "Couldn't load image '%s'\n", files[i] ? files[i] : "(null)");
			rt_free_cubemap(c);
			return rc;
		}
	}
	return RT_OK;
}

void rt_free_cubemap(Cubemap *c)
{
	if (!c) return;
	for (int i = 0; i < 6; i++) { free(c->data[i]); c->data[i] = NULL; }
}
