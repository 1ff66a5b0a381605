/* oracle/oracle_secam.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of hacktv's SECAM colour process, _vid_render_secam
 * (src/video.c:3068-3233), with the tables vid_init() builds for it
 * (src/video.c:4075-4162) and the pieces it calls: the zero-history FIR
 * (src/fir.c:357-375), the double-precision IIR (src/fir.c:710-735), the
 * complex-gain FM modulator (src/video.c:2278-2297) and the bell curve
 * (src/video.c:2172-2185).
 *
 * The process is serial in three ways that matter for bit-exactness
 * (SURVEY.md H6), all kept as the reference has them:
 *   - the vertical average uses the other colour component of the previous
 *     processed line, held in the upper half of the chrominance buffer;
 *   - the IIR state is never reset: it carries from line to line, field to
 *     field, and is even advanced by the two never-emitted "line 0" slots the
 *     line pipeline hands to the process before the first real line;
 *   - the FM loop runs to burst_left + burst_width > width, so its last two
 *     outputs land in the upper half of the buffer, where the NEXT line's
 *     15-tap FIR over-read picks them up.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

#define SECAM_FM_DEV  1000e3
#define SECAM_FM_FREQ 4328125
#define SECAM_CB_FREQ 4250000
#define SECAM_CR_FREQ 4406250
#define IRT1090 2.0738786

/* Kaiser-windowed designs shared with oracle_tables.c */
double orc_i_zero(double x);
void orc_kaiser(double *taps, int ntaps, double beta);
void orc_low_pass(double *taps, int ntaps, double sample_rate, double cutoff, double gain);

/* src/fir.c:179-228 */
static void _band_reject(double *taps, int ntaps, double sample_rate, double low_cutoff, double high_cutoff, double gain)
{
	int n, M;
	double fmax, fwT0, fwT1;

	orc_kaiser(taps, ntaps, 7.0);

	M = (ntaps - 1) / 2;
	fwT0 = 2.0 * M_PI * low_cutoff / sample_rate;
	fwT1 = 2.0 * M_PI * high_cutoff / sample_rate;

	for(n = -M; n <= M; n++)
	{
		if(n == 0) taps[n + M] *= 1.0 + (fwT0 - fwT1) / M_PI;
		else taps[n + M] *= (sin(n * fwT0) - sin(n * fwT1)) / (n * M_PI);
	}

	fmax = taps[0 + M];
	for(n = 1; n <= M; n++) fmax += 2 * taps[n + M];

	gain /= fmax;
	for(n = 0; n < ntaps; n++) taps[n] *= gain;
}

static int16_t *_q15_reversed(const double *taps, int ntaps)
{
	int16_t *q = calloc(ntaps, sizeof(int16_t));
	int j;
	for(j = 0; j < ntaps; j++) q[j] = lround(taps[ntaps - 1 - j] * 32767.0);
	return(q);
}

int orc_secam_init(orc_t *s)
{
	const hvk_config_t *c = &s->conf;
	double level = c->video_level * (c->modulation == HVK_FM ? 1.0 : c->level);
	double secam_level = (c->white_level - c->blanking_level) * level;
	double taps[51], a;
	int r, i;

	/* FM modulator at the pixel rate (src/video.c:4080, :2218-2243) */
	s->sc_level = round(INT16_MAX * secam_level);
	s->sc_lut = malloc(sizeof(c32_t) * 65536);
	for(r = INT16_MIN; r <= INT16_MAX; r++)
	{
		double d = 2.0 * M_PI / s->pixel_rate * (SECAM_FM_FREQ + (double) r / INT16_MAX * SECAM_FM_DEV);
		s->sc_lut[r - INT16_MIN].i = lround(cos(d) * INT32_MAX);
		s->sc_lut[r - INT16_MIN].q = lround(sin(d) * INT32_MAX);
	}

	/* pre-emphasis IIR (src/video.c:4087-4090) */
	s->sc_a1 = -0.90456054;
	s->sc_b0 = 2.90456054;
	s->sc_b1 = -2.80912108;
	s->sc_ix = s->sc_iy = 0;

	/* chroma low pass, 15 taps (src/video.c:4097-4098) */
	orc_low_pass(taps, 15, s->pixel_rate, 1.70e6, 1.0);
	s->sc_fir = _q15_reversed(taps, 15);

	/* luma notch around the sub-carrier, weakened (src/video.c:4100-4107) */
	_band_reject(taps, 51, s->pixel_rate, SECAM_FM_FREQ - 1e6, SECAM_FM_FREQ + 1e6, 1.0);
	taps[51 / 2] += 0.5;
	for(a = i = 0; i < 51; i++) a += taps[i];
	a = a / 1.0;
	for(i = 0; i < 51; i++) taps[i] /= a;
	s->sc_notch = _q15_reversed(taps, 51);

	/* field identification lines (src/video.c:4130-4137) */
	s->sc_fsync_level = round(350e3 / SECAM_FM_DEV * INT16_MAX);
	s->sc_fid_lines = c->secam_field_id_lines;
	if(s->sc_fid_lines < 1 || s->sc_fid_lines > 9) s->sc_fid_lines = 9;

	/* deviation limits (src/video.c:4110-4113): [0] D'b, [1] D'r */
	s->sc_dmin[0] = lround((SECAM_CB_FREQ - SECAM_FM_FREQ - 350e3) / SECAM_FM_DEV * INT16_MAX);
	s->sc_dmax[0] = lround((SECAM_CB_FREQ - SECAM_FM_FREQ + 506e3) / SECAM_FM_DEV * INT16_MAX);
	s->sc_dmin[1] = lround((SECAM_CR_FREQ - SECAM_FM_FREQ - 506e3) / SECAM_FM_DEV * INT16_MAX);
	s->sc_dmax[1] = lround((SECAM_CR_FREQ - SECAM_FM_FREQ + 350e3) / SECAM_FM_DEV * INT16_MAX);

	/* bell curve, indexed by the sample as uint16 (src/video.c:4115-4128, :2172-2185).
	 * The reference allocates 65535 entries and writes index 65535 too. */
	s->sc_bell = malloc(sizeof(c16_t) * 65536);
	for(r = INT16_MIN; r <= INT16_MAX; r++)
	{
		const double f0 = 4.286e6;
		double f = SECAM_FM_FREQ + (double) r * SECAM_FM_DEV / INT16_MAX;
		double lq, rq, d, g0, g1;

		f = f / f0 - f0 / f;
		lq = 16.0 * f;
		rq = 1.26 * f;
		d = 1.0 + rq * rq;
		g0 = 0.115 * (1.0 + lq * rq) / d;
		g1 = 0.115 * (lq - rq) / d;

		s->sc_bell[(uint16_t) r].i = lround(g0 * INT16_MAX);
		s->sc_bell[(uint16_t) r].q = lround(g1 * INT16_MAX);
	}

	/* sub-carrier envelope (src/video.c:4140-4147, :2194-2214) */
	{
		double rise = c->burst_rise * IRT1090;
		s->burst_left = round(s->pixel_rate * (c->burst_left - c->burst_rise / 2));
		s->burst_width = ceil(s->pixel_rate * (c->burst_width + rise));
		s->burst_win = malloc(s->burst_width * sizeof(int16_t));
		for(i = 0; i < s->burst_width; i++)
		{
			double t = 1.0 / s->pixel_rate * i;
			s->burst_win[i] = round(orc_rc_window(t, rise / 2, c->burst_width, rise) * 1.0 * INT16_MAX);
		}
	}

	/* lower half: the line being built; upper half: the other component of the
	 * previous line (src/video.c:4154-4156) */
	s->chroma = calloc(2 * s->width + 64, sizeof(int16_t));

	return(0);
}

void orc_secam_free(orc_t *s)
{
	free(s->sc_lut);
	free(s->sc_fir);
	free(s->sc_notch);
	free(s->sc_bell);
}

/* zero-history centred FIR in place over n samples of stride `step`; reads
 * ntaps / 2 samples past the end (src/fir.c:357-375) */
static void _fir_block(int16_t *buf, int n, int step, const int16_t *taps, int ntaps)
{
	int h = ntaps / 2, j, k;
	int16_t *in = malloc((n + 2 * h) * sizeof(int16_t));

	for(j = 0; j < h; j++) in[j] = 0;
	for(j = 0; j < n + h; j++) in[h + j] = buf[j * step];

	for(j = 0; j < n; j++)
	{
		int32_t a = 0;
		for(k = 0; k < ntaps; k++) a += (int32_t) in[j + k] * taps[k];
		a >>= 15;
		buf[j * step] = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
	}

	free(in);
}

/* The process on one line. `o` is the line's I channel (width samples, final
 * raster). frame / line are 1-based; line 0 is a pipeline-fill slot. */
void orc_secam_line(orc_t *s, int16_t *o, int16_t *oq, int frame, int line, int active_l, int active_r, int vy)
{
	const hvk_config_t *c = &s->conf;
	int W = s->width, x;
	int16_t *cb = s->chroma;
	int sl = 0, sr = 0;
	int dr = ((frame * c->lines) + line) & 1;
	int vframe_x, fbw;

	/* (the reference's process looks at the frame the RASTER is on, two lines ahead; the lines
	 * around a field change show no picture, so the line's own field is the same thing) */
	if(line >= 1) orc_select_frame(s, line);
	fbw = s->fb_width;
	/* The FIRST of the two fill slots (line 0) is processed before the source has been read at all -- the colour process's
	 * thread takes it while vid_init()'s frame is still in force: the full active width at offset 0, no pixels
	 * (src/video.c:4169-4177); the second one finds the stream's first picture, its place and width. Found with a first
	 * picture narrower than the raster at 13.5 and 14 MHz, where the sub-carrier's last two samples run past the line into
	 * the buffer's upper half and the next line's low pass reads them (tools/fuzz_oracle_ref.py, FUZZ_LONG; the filter state
	 * behind the two slots now equals the reference's to the last bit: ref_table("secam_iir")). */
	if(line == 0 && s->sc_fill_slots++ == 0) fbw = s->active_width;
	vframe_x = (s->active_width - fbw) / 2;

	if(line == 1 || line == c->hline) memset(cb, 0, sizeof(int16_t) * 2 * W);

	if(c->secam_field_id && ((line >= 7 && line < 7 + s->sc_fid_lines) || (line >= 320 && line < 320 + s->sc_fid_lines)))
	{
		/* field identification ("bottle") lines, src/video.c:3101-3133: the sub-carrier ramps from
		 * the line's rest frequency by 350 kHz over 15 us (D'r) / 18 us (D'b) */
		int16_t level = dr ? s->yuv[2] : s->yuv[1];
		int16_t dev = dr ? s->sc_fsync_level : -s->sc_fsync_level;
		double rw = dr ? 15e-6 : 18e-6;

		for(x = 0; x < W; x++)
		{
			double t = (double) (x - s->active_left) / s->pixel_rate / rw;
			if(t < 0) t = 0;
			else if(t > 1) t = 1;
			cb[x] = level + dev * t;
		}

		sl = s->burst_left;
		sr = sl + s->burst_width;
	}
	else if(active_l || active_r)
	{
		const uint32_t *prgb = NULL;
		int stride = 0;
		int comp = dr ? 2 : 1;          /* this line carries V (D'r) or U (D'b) ... */
		int other = dr ? 1 : 2;         /* ... and stores the other one for the next line */
		int16_t base = s->yuv[comp];    /* of RGB 000000 */

		if(s->fb && vy >= 0)
		{
			prgb = &s->fb[vy * s->fb_line_stride];
			stride = s->fb_pixel_stride;
		}

		for(x = 0; x < s->active_left + vframe_x; x++) cb[x] = base;

		for(; x < s->active_left + vframe_x + fbw; x++)
		{
			uint32_t rgb = prgb ? (*prgb & 0xFFFFFF) : 0;
			cb[x] = (s->yuv[rgb * 3 + comp] + cb[W + x]) / 2;
			cb[W + x] = s->yuv[rgb * 3 + other];
			if(prgb) prgb += stride;
		}

		for(; x < W; x++) cb[x] = base;

		sl = s->burst_left;
		sr = active_r ? sl + s->burst_width : s->half_width;
	}

	if(sr > sl)
	{
		c32_t phase;
		int16_t dmin = s->sc_dmin[dr], dmax = s->sc_dmax[dr];

		/* luma notch over the active picture, zero history; not with S-Video (src/video.c:3206) */
		if(!c->s_video) _fir_block(o + s->active_left, s->active_width, 1, s->sc_notch, 51);

		/* chroma low pass over the whole line; the over-read reaches the first
		 * entries of the upper half (src/video.c:3207) */
		_fir_block(cb, W, 1, s->sc_fir, 15);

		/* pre-emphasis, state carried for ever (src/video.c:3208, src/fir.c:721-735) */
		for(x = 0; x < W; x++)
		{
			double in = (double) cb[x];
			s->sc_iy = in * s->sc_b0 + s->sc_ix * s->sc_b1 - s->sc_iy * s->sc_a1;
			s->sc_ix = in;
			cb[x] = lround(s->sc_iy < INT16_MIN ? INT16_MIN : (s->sc_iy > INT16_MAX ? INT16_MAX : s->sc_iy));
		}

		/* phase reset every line, inverted two lines out of three (src/video.c:3211-3213) */
		phase.i = ((frame * c->lines) + line) % 3 == 0 ? INT32_MAX : -INT32_MAX;
		phase.q = 0;

		for(x = sl; x < sr; x++)
		{
			const c16_t *g;
			const c32_t *st;
			int64_t pi, pq;
			int32_t vi, vq;

			if(cb[x] < dmin) cb[x] = dmin;
			else if(cb[x] > dmax) cb[x] = dmax;

			g = &s->sc_bell[(uint16_t) cb[x]];
			st = &s->sc_lut[cb[x] - INT16_MIN];

			pi = (int64_t) phase.i * st->i - (int64_t) phase.q * st->q;
			pq = (int64_t) phase.i * st->q + (int64_t) phase.q * st->i;
			phase.i = pi >> 31;
			phase.q = pq >> 31;

			vi = ((phase.i >> 16) * s->sc_level) >> 15;
			vq = ((phase.q >> 16) * s->sc_level) >> 15;
			cb[x] = ((vi * g->i) >> 15) - ((vq * g->q) >> 15);

			/* x can run past the line: the result stays in the buffer's upper
			 * half, the add lands outside the line and is never emitted */
			if(x < W) (c->s_video && oq ? oq : o)[x] += (cb[x] * s->burst_win[x - sl]) >> 15;
		}
	}
}
