/* hvk_secam.c -- host pre-pass for the SECAM colour sub-carrier.
 *
 * SECAM chroma (the reference's _vid_render_secam, src/video.c:3068-3233) is a
 * frequency-modulated sub-carrier. Three couplings make it one serial chain
 * over the whole stream when the output has to be bit-exact (SURVEY.md H6,
 * DESIGN.md section 5):
 *
 *   1. the pre-emphasis IIR runs in double precision and its state is never
 *      reset (src/fir.c:721-735 called at src/video.c:3208), so every line
 *      starts from the previous processed line's end state;
 *   2. the FM loop runs to burst_left + burst_width, two samples past the line
 *      (src/video.c:3220-3229 with :4140-4147), and leaves its last two outputs
 *      where the NEXT line's 15-tap filter over-read picks them up
 *      (src/video.c:3207, src/fir.c:365-372) -- which moves that line's last
 *      filter outputs, hence its IIR end state, hence (through 1.) the rounding
 *      of the line after;
 *   3. the FM phasor itself is the floor-after-every-step recurrence of
 *      src/common.h:80-89 (944 steps per line, restarted every line).
 *
 * Like the FM/AM sound carriers this is therefore computed once, in stream
 * order, on the host, and handed to the device as a side input: the int16
 * value the process adds to each sample of the line's I channel (2 bytes per
 * sample, zero outside [burst_left, burst_left + burst_width)). The parallel
 * parts of the SECAM process -- the luma notch and the add -- are done by the
 * raster kernel.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hvk_internal.h"

#define FM_DEV   1000e3
#define FM_FREQ  4328125
#define CB_FREQ  4250000
#define CR_FREQ  4406250

struct hvk_secam {
	const hvk_tables_t *t;
	int W, lines, hline;
	int16_t *uv;            /* 2^24 x {u, v}: the chroma half of the reference's level table */
	int16_t *line;          /* [0, W): the line being built */
	int16_t *held;          /* [0, W): the other component of the previous line; [0], [1] also take the FM tail */
	int16_t *padded;        /* filter input with 7 zeros in front and the over-read behind */
	double ix, iy;          /* IIR state */
	int64_t next_frame;     /* frames must come in order */
};

/* The chroma columns of the level table (src/video.c:3912-3958, SECAM branch) */
static int16_t *_build_uv(const hvk_tables_t *t)
{
	const hvk_yuvparams_t *p = &t->yuv;
	int16_t *uv = malloc(0x1000000UL * 2 * sizeof(int16_t));
	long c;

	if(!uv) return(NULL);

	for(c = 0; c <= 0xFFFFFF; c++)
	{
		double r = p->glut[(c & 0xFF0000) >> 16];
		double g = p->glut[(c & 0x00FF00) >> 8];
		double b = p->glut[(c & 0x0000FF) >> 0];
		double y = r * p->rw + g * p->gw + b * p->bw;
		double u = (b - y) * p->eu;
		double v = (r - y) * p->ev;

		u = (u + CB_FREQ - FM_FREQ) / FM_DEV;
		v = (v + CR_FREQ - FM_FREQ) / FM_DEV;
		u = u < -1 ? -1 : (u > 1 ? 1 : u);
		v = v < -1 ? -1 : (v > 1 ? 1 : v);

		uv[c * 2 + 0] = round(u * INT16_MAX);
		uv[c * 2 + 1] = round(v * INT16_MAX);
	}

	return(uv);
}

hvk_secam_t *hvk_secam_new(const hvk_tables_t *t)
{
	hvk_secam_t *s = calloc(1, sizeof(hvk_secam_t));
	if(!s) return(NULL);

	s->t = t;
	s->W = t->k.width;
	s->lines = t->k.lines;
	s->hline = t->conf.hline;
	s->uv = _build_uv(t);
	s->line = calloc(s->W + 8, sizeof(int16_t));
	s->held = calloc(s->W + 8, sizeof(int16_t));
	s->padded = calloc(s->W + 32, sizeof(int16_t));

	if(!s->uv || !s->line || !s->held || !s->padded)
	{
		hvk_secam_free(s);
		return(NULL);
	}

	return(s);
}

void hvk_secam_free(hvk_secam_t *s)
{
	if(!s) return;
	free(s->uv);
	free(s->line);
	free(s->held);
	free(s->padded);
	free(s);
}

/* lround() without the library call: truncate, then look at the (exactly representable) rest.
 * |x| < 2^31 here. Halves go away from zero, like lround. */
static inline int32_t _round_away(double x)
{
	int32_t i = (int32_t) x;
	double f = x - (double) i;
	if(f >= 0.5) i++;
	else if(f <= -0.5) i--;
	return(i);
}

/* One line of the process. `row` points at the source pixels shown on this
 * line (NULL: none); out receives the W values to add to the line (NULL: a
 * pipeline-fill slot whose result is never emitted). */
static void _line(hvk_secam_t *s, int frame, int line, int picture, int right_half, int field_id,
                  const uint32_t *row, int row_width, int vframe_x, int16_t *out)
{
	const hvk_tables_t *t = s->t;
	const int W = s->W;
	const int dr = ((frame * s->lines) + line) & 1;    /* D'r line, else D'b */
	const int sl = t->k.burst_left;
	const int sr = (right_half || field_id) ? sl + t->k.burst_width : t->k.half_width;
	int x;

	if(out) memset(out, 0, sizeof(int16_t) * W);

	/* top of a field: both halves of the reference's buffer are cleared (src/video.c:3095-3099) */
	if(line == 1 || line == s->hline)
	{
		memset(s->line, 0, sizeof(int16_t) * W);
		memset(s->held, 0, sizeof(int16_t) * W);
	}

	if((!picture && !field_id) || sr <= sl) return;

	if(field_id)
	{
		/* field identification line (src/video.c:3101-3133): the sub-carrier ramps from the line's
		 * rest frequency by 350 kHz over 15 us (D'r) / 18 us (D'b); the held component is left alone */
		const int16_t level = s->uv[dr ? 1 : 0];            /* of RGB 000000 */
		const int16_t dev = dr ? t->secam_fsync_level : -t->secam_fsync_level;
		const double rw = dr ? 15e-6 : 18e-6;

		for(x = 0; x < W; x++)
		{
			double tt = (double) (x - t->k.active_left) / t->pixel_rate / rw;
			if(tt < 0) tt = 0;
			else if(tt > 1) tt = 1;
			s->line[x] = level + dev * tt;
		}
	}
	else
	/* colour difference of this line, averaged with the previous line's
	 * (src/video.c:3149-3196): D'r lines carry v and keep u for the next line */
	{
		const int mine = dr ? 1 : 0 /* index into {u, v} */, other = dr ? 0 : 1;
		const int16_t rest = s->uv[mine];                   /* of RGB 000000 */
		const int p0 = t->k.active_left + vframe_x;

		for(x = 0; x < p0; x++) s->line[x] = rest;
		for(; x < p0 + row_width; x++)
		{
			const uint32_t rgb = row ? (row[x - p0] & 0xFFFFFF) : 0;
			s->line[x] = (s->uv[rgb * 2 + mine] + s->held[x]) / 2;
			s->held[x] = s->uv[rgb * 2 + other];
		}
		for(; x < W; x++) s->line[x] = rest;
	}

	/* 15-tap low pass, zero history, reading 7 samples past the line: the
	 * start of the held component (src/video.c:3207) */
	{
		const int16_t *taps = t->secam_fir;
		int16_t *in = s->padded;
		int k;

		memset(in, 0, 7 * sizeof(int16_t));
		memcpy(in + 7, s->line, W * sizeof(int16_t));
		memcpy(in + 7 + W, s->held, 7 * sizeof(int16_t));

		for(x = 0; x < W; x++)
		{
			int32_t a = 0;
			for(k = 0; k < 15; k++) a += (int32_t) in[x + k] * taps[k];
			a >>= 15;
			s->line[x] = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
		}
	}

	/* pre-emphasis (src/fir.c:721-735), state carried for ever */
	{
		double ix = s->ix, iy = s->iy;
		for(x = 0; x < W; x++)
		{
			const double in = (double) s->line[x];
			iy = in * 2.90456054 + ix * -2.80912108 - iy * -0.90456054;
			ix = in;
			s->line[x] = _round_away(iy < INT16_MIN ? INT16_MIN : (iy > INT16_MAX ? INT16_MAX : iy));
		}
		s->ix = ix;
		s->iy = iy;
	}

	/* limit, bell gain, FM, envelope (src/video.c:3210-3229, :2278-2297) */
	{
		const int16_t dmin = t->secam_dmin[dr], dmax = t->secam_dmax[dr];
		const int32_t level = t->secam_level;
		int32_t pi = ((frame * s->lines) + line) % 3 == 0 ? INT32_MAX : -INT32_MAX;
		int32_t pq = 0;

		for(x = sl; x < sr; x++)
		{
			/* past the line the loop works on the first entries of the held component */
			int16_t *cell = x < W ? &s->line[x] : &s->held[x - W];
			int16_t v = *cell < dmin ? dmin : (*cell > dmax ? dmax : *cell);
			const hvk_c16_t g = t->secam_bell[(uint16_t) v];
			const hvk_c32_t st = t->secam_lut[v - INT16_MIN];
			int64_t ni = (int64_t) pi * st.i - (int64_t) pq * st.q;
			int64_t nq = (int64_t) pi * st.q + (int64_t) pq * st.i;
			int32_t vi, vq;

			pi = (int32_t) (ni >> 31);
			pq = (int32_t) (nq >> 31);

			vi = ((pi >> 16) * level) >> 15;
			vq = ((pq >> 16) * level) >> 15;
			v = (int16_t) (((vi * g.i) >> 15) - ((vq * g.q) >> 15));
			*cell = v;

			if(x < W && out) out[x] = (int16_t) ((v * t->burst_win[x - sl]) >> 15);
		}
	}
}

/* Chroma contribution of one whole frame (frame_samples int16). fb is the
 * cropped, dense frame shown on it (NULL: none). Frames must be presented in
 * stream order. */
int hvk_secam_frame(hvk_secam_t *s, int64_t frame_index, const uint32_t *fb1, int fb1_width, int fb1_height,
                    int fb1_interlaced, const uint32_t *fb2, int fb2_width, int fb2_height, int fb2_interlaced, int16_t *out)
{
	const hvk_tables_t *t = s->t;
	const hvk_kconst_t *k = &t->k;
	const int frame = (int) (frame_index + 1);
	int line;

	if(frame_index != s->next_frame) return(HVK_ERROR);

	/* The line pipeline hands the process two never-emitted slots (frame 1,
	 * line 0) before the first real line; it treats them as picture lines
	 * without a picture and they advance the IIR (src/video.c:4676-4688 with
	 * :4665-4667) */
	if(frame_index == 0)
	{
		_line(s, 1, 0, 1, 1, 0, NULL, k->active_width, 0, NULL);
		_line(s, 1, 0, 1, 1, 0, NULL, k->active_width, 0, NULL);
	}

	for(line = 1; line <= k->lines; line++)
	{
		const hvk_linedesc_t *d = &t->desc[(frame & 1) * k->lines + line - 1];
		const int picture = d->ar > d->al;
		const int right_half = picture && d->ar > k->half_width;
		/* the frame the line's field shows */
		const int second = k->fields == 2 && line >= k->hline;
		const uint32_t *fb = second ? fb2 : fb1;
		const int fb_width = second ? fb2_width : fb1_width, fb_height = second ? fb2_height : fb1_height;
		const int fb_interlaced = second ? fb2_interlaced : fb1_interlaced;
		const int vframe_x = (k->active_width - fb_width) / 2;
		const int vframe_y = (k->active_lines - fb_height) / 2;
		const uint32_t *row = NULL;
		int vy = d->src_row;

		if(vy >= 0 && k->interlaced != 0 && fb_interlaced != k->interlaced) vy += 1;
		vy -= vframe_y;
		if(fb && vy >= 0 && vy < fb_height) row = fb + (size_t) vy * fb_width;

		/* an empty frame (0 x 0, what a source past its end hands out) shows no pixels at all */
		_line(s, frame, line, picture, right_half, d->secam_fid & 1, row, fb_width, vframe_x, out + (size_t) (line - 1) * k->width);
	}

	s->next_frame++;
	return(HVK_OK);
}
