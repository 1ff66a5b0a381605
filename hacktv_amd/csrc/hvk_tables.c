/* hvk_tables.c -- host-side table builder of the MI355X engine.
 *
 * Everything vid_init() precomputes for the hot path (src/video.c:3812-4162,
 * :4375-4558 and the designers in src/fir.c:32-255, src/nicam728.c:257-331) is
 * built here, once, in double precision with the host libm, and then uploaded
 * to HBM by hvk_open(). The tables have to be produced on the host: entries
 * such as round(cos(d * c) * 32767) over 2.56 M phases are only bit-identical
 * with the reference when the same libm evaluates them.
 *
 * The layout is the device's, not the reference's: pulses are a flat value
 * array with an index, the per-line code strings become a [2][lines] array of
 * descriptors, filter taps are stored in the order they meet the samples.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <gnu/libc-version.h>
#include "hvk_internal.h"
#include "hvk_fm_taps.h"

/* a configuration the engine does not render says why on stderr, as hvk_open()'s own checks do (the reference prints its
 * refusals the same way, src/video.c, src/hacktv.c), and hvk_open() returns HVK_UNSUPPORTED */
#define REFUSE(...) do { fprintf(stderr, "libhvk: refused: "); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); return(HVK_UNSUPPORTED); } while(0)

#define EDGE_0_100 2.0738786   /* 10-90 % rise time -> full width of an integrated raised-cosine edge (src/common.h:28) */

/* ------------------------------------------------------------------ */
/* shaped edges                                                        */

/* Integrated raised-cosine window: 1 inside, 0 outside, `rise` wide edges
 * centred on left and left + width (src/common.c:231-257) */
static double _window(double t, double left, double width, double rise)
{
	t -= left + width / 2;
	t = fabs(t) - (width - rise) / 2;

	if(t <= 0) return(1.0);
	if(t >= rise) return(0.0);

	t = 1.0 - t / rise * 2;
	return(0.5 * (1.0 + t + sin(M_PI * t) / M_PI));
}

/* One sync pulse as the reference's vbidata step renderer quantises it
 * (src/vbidata.c:36-81): samples floor(offset - rise/2) .. ceil(offset +
 * width + rise/2), leading and trailing zeros trimmed. Appends the values to
 * `out` (if not NULL) and returns their count; *first receives the sample
 * index of the first value. */
static int _quantise_pulse(int16_t *out, int *first, double offset, double width, double rise, int level)
{
	int x, x1 = floor(offset - rise / 2), x2 = ceil(offset + width + rise / 2);
	int start = 0, len = 0;

	for(x = x1; x <= x2; x++)
	{
		int v = round(_window(x, offset, width, rise) * level);
		if(v == 0) continue;
		if(len == 0) start = x;
		if(out)
		{
			int i;
			for(i = len; i < x - start; i++) out[i] = 0;
			out[x - start] = v;
		}
		len = x - start + 1;
	}

	*first = start;
	return(len);
}

/* ------------------------------------------------------------------ */
/* line descriptors                                                    */

/* Field structure of the two interlaced rasters as runs of lines. Pulse ids:
 * 0 line sync, 1 equalising, 2 broad, 3 equalising at mid-line, 4 broad at
 * mid-line, -1 none. burst: 'A' every frame, 'E' frames with (frame & 1) == 0,
 * 'O' frames with (frame & 1) == 1, '-' none (the frame parity tests are
 * src/video.c:2901-2903). act: bit 0 picture in the left half, bit 1 in the
 * right half. Source: src/video.c:2479-2591. */
typedef struct { short first, last; signed char left, mid; char burst; char act; } _run_t;

static const _run_t _runs_625[] = {
	{   1,   2, 2,  4, '-', 0 }, {   3,   3, 2,  3, '-', 0 }, {   4,   5, 1,  3, '-', 0 },
	{   6,   6, 0, -1, 'E', 0 }, {   7,  22, 0, -1, 'A', 0 }, {  23,  23, 0, -1, 'A', 2 },
	{  24, 309, 0, -1, 'A', 3 }, { 310, 310, 0, -1, 'E', 3 }, { 311, 312, 1,  3, '-', 0 },
	{ 313, 313, 1,  4, '-', 0 }, { 314, 315, 2,  4, '-', 0 }, { 316, 317, 1,  3, '-', 0 },
	{ 318, 318, 1, -1, '-', 0 }, { 319, 319, 0, -1, 'O', 0 }, { 320, 335, 0, -1, 'A', 0 },
	{ 336, 621, 0, -1, 'A', 3 }, { 622, 622, 0, -1, 'E', 3 }, { 623, 623, 0,  3, '-', 1 },
	{ 624, 625, 1,  3, '-', 0 }, { 0, 0, 0, 0, 0, 0 },
};

static const _run_t _runs_525[] = {
	{   1,   3, 1,  3, '-', 0 }, {   4,   6, 2,  4, '-', 0 }, {   7,   9, 1,  3, '-', 0 },
	{  10,  20, 0, -1, 'A', 0 }, {  21, 262, 0, -1, 'A', 3 }, { 263, 263, 0,  3, 'A', 1 },
	{ 264, 265, 1,  3, '-', 0 }, { 266, 266, 1,  4, '-', 0 }, { 267, 268, 2,  4, '-', 0 },
	{ 269, 269, 2,  3, '-', 0 }, { 270, 271, 1,  3, '-', 0 }, { 272, 272, 1, -1, '-', 0 },
	{ 273, 282, 0, -1, 'A', 0 }, { 283, 283, 0, -1, 'A', 2 }, { 284, 525, 0, -1, 'A', 3 },
	{ 0, 0, 0, 0, 0, 0 },
};

/* The other rasters of src/video.c:2592-2810. None of them but the 405-line ones ('A' on lines that exist) carries a
 * colour burst; the mechanical ones have no vertical interval to speak of. */
static const _run_t _runs_819[] = {
	{   1,   1, 2, -1, '-', 0 }, {   2,  38, 0, -1, '-', 0 }, {  39, 405, 0, -1, '-', 3 }, { 406, 406, 0, -1, '-', 1 },
	{ 407, 408, 0, -1, '-', 0 }, { 409, 409, 0,  4, '-', 0 }, { 410, 446, 0, -1, '-', 0 }, { 447, 447, 0, -1, '-', 2 },
	{ 448, 816, 0, -1, '-', 3 }, { 817, 819, 0, -1, '-', 0 }, { 0, 0, 0, 0, 0, 0 },
};

static const _run_t _runs_405[] = {
	{   1,   4, 2,  4, '-', 0 }, {   5,  15, 0, -1, 'A', 0 }, {  16, 202, 0, -1, 'A', 3 }, { 203, 203, 0,  4, 'A', 1 },
	{ 204, 206, 2,  4, '-', 0 }, { 207, 207, 2, -1, '-', 0 }, { 208, 217, 0, -1, 'A', 0 }, { 218, 218, 0, -1, 'A', 2 },
	{ 219, 405, 0, -1, 'A', 3 }, { 0, 0, 0, 0, 0, 0 },
};

static const _run_t _runs_cbs405[] = {
	{   1,   3, 1,  3, '-', 0 }, {   4,   6, 2,  4, '-', 0 }, {   7,   9, 1,  3, '-', 0 }, {  10,  14, 0, -1, '-', 0 },
	{  15, 202, 0, -1, '-', 3 }, { 203, 203, 0,  3, '-', 1 }, { 204, 205, 1,  3, '-', 0 }, { 206, 206, 1,  4, '-', 0 },
	{ 207, 208, 2,  4, '-', 0 }, { 209, 209, 2,  3, '-', 0 }, { 210, 211, 1,  3, '-', 0 }, { 212, 212, 1, -1, '-', 0 },
	{ 213, 216, 0, -1, '-', 0 }, { 217, 217, 0, -1, '-', 2 }, { 218, 405, 0, -1, '-', 3 }, { 0, 0, 0, 0, 0, 0 },
};

static const _run_t _runs_apollo320[] = { { 1, 8, 2, 3, '-', 0 }, { 9, 320, 0, -1, '-', 3 }, { 0, 0, 0, 0, 0, 0 } };
static const _run_t _runs_baird240[] = { { 1, 12, 2, 4, '-', 0 }, { 13, 20, 0, -1, '-', 0 }, { 21, 240, 0, -1, '-', 3 }, { 0, 0, 0, 0, 0, 0 } };
static const _run_t _runs_baird30[] = { { 1, 30, -1, -1, '-', 3 }, { 0, 0, 0, 0, 0, 0 } };     /* no sync pulses at all */
static const _run_t _runs_nbtv32[] = { { 1, 1, -1, -1, '-', 3 }, { 2, 32, 0, -1, '-', 3 }, { 0, 0, 0, 0, 0, 0 } };

/* a raster type's runs and its number of lines (0: not a raster this engine renders) */
static const _run_t *_runs_of(int type, int *lines)
{
	switch(type)
	{
	case HVK_RASTER_625: *lines = 625; return(_runs_625);
	case HVK_RASTER_525: *lines = 525; return(_runs_525);
	case HVK_RASTER_819: *lines = 819; return(_runs_819);
	case HVK_RASTER_405: *lines = 405; return(_runs_405);
	case HVK_CBS_405:    *lines = 405; return(_runs_cbs405);
	case HVK_APOLLO_320: *lines = 320; return(_runs_apollo320);
	case HVK_BAIRD_240:  *lines = 240; return(_runs_baird240);
	case HVK_BAIRD_30:   *lines = 30;  return(_runs_baird30);
	case HVK_NBTV_32:    *lines = 32;  return(_runs_nbtv32);
	}
	*lines = 0;
	return(NULL);
}

static const _run_t *_find_run(const _run_t *r, int line)
{
	for(; r->first; r++) if(line >= r->first && line <= r->last) return(r);
	return(NULL);
}

/* The source row a line shows, before centring (src/video.c:2812-2862) */
static int _row_of_line(int type, int line)
{
	switch(type)
	{
	case HVK_RASTER_625: return(line < 313 ? (line - 23) * 2 : (line - 336) * 2 + 1);
	case HVK_RASTER_525: return(line < 265 ? (line - 23) * 2 : (line - 286) * 2 + 1);
	case HVK_RASTER_819: return(line < 406 ? (line - 48) * 2 : (line - 457) * 2 + 1);
	case HVK_RASTER_405: return(line < 210 ? (line - 16) * 2 : (line - 218) * 2 + 1);
	case HVK_CBS_405:    return(line < 210 ? (line - 16) * 2 : (line - 219) * 2 + 1);
	case HVK_APOLLO_320: return(line - 9);
	case HVK_BAIRD_240:  return(line - 20);
	case HVK_BAIRD_30:   return(line - 1);
	case HVK_NBTV_32:    return(line - 1);
	}
	return(-1);
}

/* The part of a line that only depends on which sync pulses it carries: blanking level plus its own
 * left and mid pulse plus the part of the NEXT line's left pulse that starts before that line's
 * sample 0 (src/video.c:2944-2958 with src/vbidata.c:186-239, :211-216), modulo 2^16. A frame has a
 * handful of such combinations; the kernels start a line from the row of its combination -- one
 * aligned 16-byte load per lane -- instead of adding up to three pulses under range tests. The
 * row's index goes into the high byte of the descriptor's secam_fid. */
static int _build_linebase(hvk_tables_t *t)
{
	const int W = t->k.width, L = t->conf.lines, n = 2 * L;
	const int stride = ((W + 7) & ~7) + 8;
	int key[126][5], nbase = 0, i, b, p, j, pass;

	/* a line's base: its own left and mid pulse, the part of the next line's left pulse in front of that line's sample 0,
	 * and what the pulses of the line BEFORE leave behind their line's end (key[3], key[4]; -1: nothing) */
	for(pass = 0; pass < 2; pass++)
	for(i = 0; i < n; i++)
	{
		hvk_linedesc_t *d = &t->desc[i];
		const hvk_linedesc_t *pd = &t->desc[(i / L) * L + (i % L + L - 1) % L];
		int k5[5] = { d->pulse_left, d->pulse_mid, d->pulse_next, -1, -1 };
		if(pass == 0)
		{
			if(pd->pulse_left >= 0 && t->k.pulse_offset[pd->pulse_left] + t->k.pulse_length[pd->pulse_left] > W) k5[3] = pd->pulse_left;
			if(pd->pulse_mid >= 0 && t->k.pulse_offset[pd->pulse_mid] + t->k.pulse_length[pd->pulse_mid] > W) k5[4] = pd->pulse_mid;
			if(k5[3] >= 0 || k5[4] >= 0) t->k.spill_lines = 1;       /* (the count follows below) */
		}
		for(b = 0; b < nbase; b++) if(!memcmp(key[b], k5, sizeof(k5))) break;
		if(b == nbase)
		{
			if(nbase == 126) REFUSE("more than 126 kinds of line start in this raster");
			memcpy(key[b], k5, sizeof(k5));
			nbase++;
		}
		/* pass 0: the row with what the line before left here (bits 8 ..); pass 1: the row without (bits 1 .. 7) */
		if(pass == 0) d->secam_fid = (int16_t) ((d->secam_fid & 1) | (b << 8));
		else d->secam_fid = (int16_t) (d->secam_fid | (b << 1));
	}

	free(t->linebase);
	t->linebase = calloc((size_t) nbase * stride, sizeof(int16_t));
	if(!t->linebase) return(HVK_OUT_OF_MEMORY);
	t->nbase = nbase;
	t->k.base_stride = stride;

	for(b = 0; b < nbase; b++)
	{
		int16_t *row = t->linebase + (size_t) b * stride;
		for(j = 0; j < stride; j++) row[j] = (int16_t) t->k.blanking;
		for(p = 0; p < 5; p++)
		{
			const int id = key[b][p];
			if(id < 0) continue;
			const int off = t->k.pulse_offset[id] + (p == 2 ? W : (p >= 3 ? -W : 0));
			const int16_t *v = t->pulse_values + t->k.pulse_start[id];
			for(j = 0; j < t->k.pulse_length[id]; j++)
			{
				if(off + j >= 0 && off + j < W) row[off + j] = (int16_t) (row[off + j] + v[j]);
			}
		}
	}
	return(HVK_OK);
}

/* The reference's line buffers form a ring of `olines` (src/video.c:3578: every process adds its window, two neighbours
 * share a buffer unless one of them runs on a thread of its own): its length for this configuration */
static int _ring_lines(const hvk_tables_t *t)
{
	const hvk_config_t *c = &t->conf;
	int olines = c->raw_bb ? 1 : 3, prev_thread = 0;
#define PROCESS(nl, th) do { olines += (nl) - ((th) || prev_thread ? 0 : 1); prev_thread = (th); } while(0)
	if(!c->raw_bb && c->colour_mode == HVK_SECAM) PROCESS(1, 1);
	if(c->vits) PROCESS(1, 0);
	if(c->wss) PROCESS(1, 0);
	if(c->acp) PROCESS(1, 0);
	if(c->vitc) PROCESS(1, 0);
	if(c->cc608) PROCESS(1, 0);
	if(c->sis) PROCESS(1, 0);
	if(c->teletext) PROCESS(1, 0);
	if(t->pixel_rate != t->sample_rate) PROCESS(2, 1);
	if(c->vfilter) PROCESS(2, 1);                   /* (1 + the filter's delay of one line) */
	PROCESS(1, 1);                                  /* audio, always */
	if(c->modulation == HVK_FM) PROCESS(1, 1);
	if(c->swap_iq) PROCESS(1, 0);
	if(c->offset) PROCESS(1, 1);
	if(c->passthru) PROCESS(1, 0);
	PROCESS(1, 0);                                  /* output */
#undef PROCESS
	return(olines);
}

static int _build_linedesc(hvk_tables_t *t)
{
	const hvk_config_t *c = &t->conf;
	int nl;
	const _run_t *runs = _runs_of(c->type, &nl);
	int colour = c->colour_mode == HVK_PAL || c->colour_mode == HVK_NTSC;
	int p, line;

	if(!runs || nl != c->lines) REFUSE("no line sequence for a raster of %d lines of this type", c->lines);
	t->desc = calloc(2 * c->lines, sizeof(hvk_linedesc_t));
	if(!t->desc) return(HVK_OUT_OF_MEMORY);

	for(p = 0; p < 2; p++)
	for(line = 1; line <= c->lines; line++)
	{
		hvk_linedesc_t *d = &t->desc[p * c->lines + line - 1];
		const _run_t *r = _find_run(runs, line);
		const _run_t *rn = _find_run(runs, line == c->lines ? 1 : line + 1);

		if(!r || !rn) return(HVK_ERROR);

		d->pulse_left = r->left;
		d->pulse_mid = r->mid;
		d->pulse_next = (rn->left >= 0 && t->k.pulse_offset[rn->left] < 0) ? rn->left : -1;

		d->al = d->ar = 0;
		if(r->act)
		{
			d->al = (r->act & 1) ? t->k.active_left : t->k.half_width;
			d->ar = (r->act & 2) ? t->k.active_left + t->k.active_width : t->k.half_width;
		}
		d->src_row = _row_of_line(c->type, line);
		if(d->src_row < 0) d->src_row = -1;     /* (a line in front of the picture -- 819 lines: 39 .. 47 -- shows black) */

		/* SECAM field identification lines at the top of each field (src/video.c:3101-3103) */
		d->secam_fid = c->colour_mode == HVK_SECAM && c->secam_field_id && c->lines == 625 &&
		               ((line >= 7 && line < 7 + t->secam_fid_lines) || (line >= 320 && line < 320 + t->secam_fid_lines));

		d->pal = 0;
		if(colour)
		{
			/* p is (frame & 1) */
			if(r->burst == 'A' || (r->burst == 'E' && p == 0) || (r->burst == 'O' && p == 1)) d->pal = 1;
			if(c->colour_mode == HVK_PAL && d->pal && ((p + line) & 1)) d->pal = -1;
		}
	}

	/* the pulses the lines use end inside the line behind their own at the latest */
	for(p = 0; p < 2 * c->lines; p++)
	{
		const int ids[2] = { t->desc[p].pulse_left, t->desc[p].pulse_mid };
		for(line = 0; line < 2; line++)
		{
			if(ids[line] >= 0 && t->k.pulse_offset[ids[line]] + t->k.pulse_length[ids[line]] > 2 * t->k.width) REFUSE("a sync pulse of this raster runs on past the line after its own at %u Hz", t->pixel_rate);
		}
	}
	{
		int r = _build_linebase(t);
		if(r != HVK_OK) return(r);
	}
	if(t->k.spill_lines)
	{
		/* How many lines the stream is old before a pulse can run on into the line behind its own: every buffer of the ring
		 * (_ring_lines()) starts out with width 0, and the renderer stops at such a buffer (src/vbidata.c:219-236) -- the
		 * buffer behind the line being drawn has been used once the raster has gone round the ring: from stream line
		 * olines - 1 on. (The pulses of stream lines 0 .. olines - 2 lose what runs over: lines 1 .. olines - 1 receive
		 * nothing, nor does line 0.) */
		t->k.spill_lines = _ring_lines(t);
	}
	return(HVK_OK);
}

static hvk_c32_t _unit_phasor(double radians);

/* ------------------------------------------------------------------ */
/* FIR designers                                                       */

/* modified Bessel I0 by its power series (src/fir.c:32-51) */
static double _bessel_i0(double x)
{
	double sum = 1, u = 1, halfx = x / 2.0, temp;
	int n = 1;

	do
	{
		temp = halfx / (double) n;
		n += 1;
		temp *= temp;
		u *= temp;
		sum += u;
	}
	while(u >= 1e-21 * sum);

	return(sum);
}

/* Kaiser-windowed sinc low pass with unity DC gain (src/fir.c:53-69, :89-137) */
static void _design_low_pass_gain(double *taps, int ntaps, double sample_rate, double cutoff, double gain);

static void _design_low_pass(double *taps, int ntaps, double sample_rate, double cutoff)
{
	_design_low_pass_gain(taps, ntaps, sample_rate, cutoff, 1);
}

static void _design_low_pass_gain(double *taps, int ntaps, double sample_rate, double cutoff, double gain)
{
	const double beta = 7.0;
	double inv_i0 = 1.0 / _bessel_i0(beta);
	double inm1 = 1.0 / ((double) (ntaps - 1));
	double w = 2.0 * M_PI * cutoff / sample_rate;
	double dc;
	int M = (ntaps - 1) / 2, i, n;

	taps[0] = taps[ntaps - 1] = inv_i0;
	for(i = 1; i < ntaps - 1; i++)
	{
		double temp = 2 * i * inm1 - 1;
		taps[i] = _bessel_i0(beta * sqrt(1.0 - temp * temp)) * inv_i0;
	}

	for(n = -M; n <= M; n++)
	{
		if(n == 0) taps[n + M] *= w / M_PI;
		else taps[n + M] *= sin(n * w) / (n * M_PI);
	}

	dc = taps[M];
	for(n = 1; n <= M; n++) dc += 2 * taps[n + M];

	gain /= dc;
	for(n = 0; n < ntaps; n++) taps[n] *= gain;
}

/* Gaussian low pass for the chroma baseband (src/fir.c:139-177) */
static int _design_gaussian(double **ptaps, double sample_rate, double cutoff)
{
	int ntaps = ((int) ceil(sample_rate / 1.35e6 / (cutoff / 1.4e6))) | 1;
	double *taps = calloc(ntaps, sizeof(double));
	double f = 13.5e6 / sample_rate;
	double s = 354372.0 / cutoff;
	double sum = 0, gain = 1;
	int h = ntaps / 2, x;

	for(x = 0; x <= h; x++)
	{
		double t = (double) x / 5 * f;
		double r = 1.0 / s * pow(2.0 * M_PI, 0.5) * pow(M_E, -pow(t, 2.0) / (2.0 * pow(s, 2)));
		sum += r * (x > 0 ? 2 : 1);
		taps[h + x] = taps[h - x] = r;
	}

	gain /= sum;
	for(x = 0; x < ntaps; x++) taps[x] *= gain;

	*ptaps = taps;
	return(ntaps);
}

/* Quantise to Q15 in the order the taps meet the samples: the reference
 * stores the design reversed (src/fir.c:279-286), so tap[k] multiplies the
 * sample k places after the oldest one in the window. */
static int16_t *_q15_applied(const double *taps, int ntaps, int stride)
{
	int16_t *q = calloc(ntaps, sizeof(int16_t));
	int k;
	if(q) for(k = 0; k < ntaps; k++) q[k] = lround(taps[(ntaps - 1 - k) * stride] * 32767.0);
	return(q);
}

/* ------------------------------------------------------------------ */
/* the ghost samples                                    */

static long _malloc_chunk(long request)
{
	long c = (request + 8 + 15) & ~15L;
	return(c < 32 ? 32 : c);
}

/* Default for what follows the reference's chrominance buffer in memory.
 * The reference's zero-history chroma FIR is fed width + ntaps/2 samples per
 * channel (src/fir.c:365-372 after the pre-load, with samples = width from
 * src/video.c:3019-3020), i.e. it reads ntaps/2 interleaved samples past the
 * 2*width int16 allocation (src/video.c:3990). In the reference CLI on glibc
 * 2.35 those bytes are: the rest of the buffer's own chunk (never written:
 * zero), the size word of the next chunk, and that chunk's contents. The next
 * chunk is the temporary array of design taps (src/video.c:4004), freed at
 * :4013 and immediately handed out again for the burst window (:4021 ->
 * :2201) because both requests round to the same chunk size. */
void hvk_tables_default_ghost(hvk_tables_t *t)
{
	long req = (long) sizeof(int16_t) * 2 * t->k.width;
	long slack = (_malloc_chunk(req) - 8 - req) / (long) sizeof(int16_t);
	long c_taps = _malloc_chunk((long) t->k.chroma_ntaps * (long) sizeof(double));
	long c_bwin = _malloc_chunk((long) t->k.burst_width * (long) sizeof(int16_t));
	int i, o;

	memset(t->ghost, 0, sizeof(t->ghost));
	if(t->k.chroma_ntaps == 0 || t->chroma_unfiltered || t->burst_win == NULL) return;
	{
		/* the model is glibc 2.35's malloc (chunk sizes, the size word's flag bits, which freed chunk the next request gets):
		 * said once where the process runs on another, instead of silently */
		static int said;
		const char *v = gnu_get_libc_version();
		if(!said && v && strncmp(v, "2.35", 4) != 0 && !getenv("HVK_QUIET"))
		{
			said = 1;
			fprintf(stderr, "libhvk: note: the samples the reference's chroma low pass reads past its buffer (src/fir.c:365-372) are modelled "
			                "for glibc 2.35; this is glibc %s -- a reference built here may end its colour lines differently: hvk_set_chroma_ghost()\n", v);
		}
	}

	o = slack;
	if(o + 4 > HVK_GHOST_LEN) return;
	t->ghost[o] = (int16_t) ((c_taps | 1) & 0xFFFF); /* low 16 bits of size | PREV_INUSE */
	o += 4;

	if(c_taps != c_bwin) return;
	for(i = 0; o < HVK_GHOST_LEN && i < t->k.burst_width; i++, o++) t->ghost[o] = t->burst_win[i];
}

/* ------------------------------------------------------------------ */
/* audio sub-carrier tables                                            */

/* lround(x * 32767) of the 65-tap symmetric 32 kHz audio filters of
 * src/video.c:2118-2168 (first 33 taps; tap[64 - k] == tap[k]) */
static const int32_t _audio_flat[33] = {
	0, -26, 10, -42, 25, -68, 44, -101, 63, -133, 71, -149, 52, -130, -13, -60,
	-138, 77, -333, 283, -593, 550, -904, 856, -1235, 1169, -1552, 1450, -1814,
	1663, -1987, 1777, 30719
};
static const int32_t _audio_50us[33] = {
	40, -86, 95, -158, 177, -265, 290, -399, 409, -518, 478, -552, 418, -414,
	138, -17, -437, 699, -1345, 1748, -2566, 3076, -4015, 4560, -5532, 5997,
	-6890, 7033, -7753, 6441, -7411, -19876, 81829
};
static const int32_t _audio_75us[33] = {
	65, -123, 147, -227, 270, -385, 440, -580, 619, -752, 726, -799, 641, -588,
	231, 6, -616, 1073, -1956, 2632, -3763, 4603, -5910, 6798, -8169, 8898,
	-10227, 10324, -11683, 9020, -11904, -32509, 116205
};

static void _mirror65(int32_t *dst, const int32_t *half)
{
	int k;
	for(k = 0; k <= 32; k++) dst[k] = dst[64 - k] = half[k];
}

static hvk_c32_t _unit_phasor(double radians)
{
	hvk_c32_t r;
	r.i = lround(cos(radians) * INT32_MAX);
	r.q = lround(sin(radians) * INT32_MAX);
	return(r);
}

static double _root_raised_cosine(double x, double b, double t)
{
	/* src/common.c:259-283 */
	if(x == 0) return((1.0 / t) * (1.0 + b * (4.0 / M_PI - 1)));

	if(fabs(x) == t / (4.0 * b))
	{
		return(b / (t * sqrt(2.0)) * ((1.0 + 2.0 / M_PI) * sin(M_PI / (4.0 * b)) + (1.0 - 2.0 / M_PI) * cos(M_PI / (4.0 * b))));
	}

	{
		double t1 = (4.0 * b * (x / t));
		double t2 = (sin(M_PI * (x / t) * (1.0 - b)) + 4.0 * b * (x / t) * cos(M_PI * (x / t) * (1.0 + b)));
		double t3 = (M_PI * (x / t) * (1.0 - t1 * t1));
		return((1.0 / t) * (t2 / t3));
	}
}

static unsigned int _gcd_u(unsigned int a, unsigned int b)
{
	unsigned int c;
	while((c = a % b)) { a = b; b = c; }
	return(b);
}

static int _build_audio(hvk_tables_t *t, double slevel)
{
	const hvk_config_t *c = &t->conf;
	int i;

	t->k.has_carriers = 0;
	t->k.has_nicam = 0;

	/* FM sound carrier (src/video.c:4404-4441, :2218-2243) */
	if(c->fm_mono_level > 0 && c->fm_mono_carrier != 0)
	{
		int r;

		t->fm_level = (int16_t) round(INT16_MAX * (c->fm_mono_level * slevel));
		t->fm_lut = malloc(sizeof(hvk_c32_t) * 65536);
		if(!t->fm_lut) return(HVK_OUT_OF_MEMORY);

		for(r = INT16_MIN; r <= INT16_MAX; r++)
		{
			double d = 2.0 * M_PI / t->sample_rate * (c->fm_mono_carrier + (double) r / INT16_MAX * c->fm_mono_deviation);
			t->fm_lut[r - INT16_MIN] = _unit_phasor(d);
		}

		t->has_limiter = 0;
		if(c->fm_mono_preemph == HVK_50US || c->fm_mono_preemph == HVK_75US)
		{
			_mirror65(t->limiter_vtaps, c->fm_mono_preemph == HVK_50US ? _audio_50us : _audio_75us);
			_mirror65(t->limiter_ftaps, _audio_flat);
			for(i = 0; i < 21; i++)
			{
				/* src/fir.c:800-803 with width 21 */
				t->limiter_shape[i] = lround((1.0 - cos(2.0 * M_PI / (21 + 1) * (i + 1))) * 0.5 * INT16_MAX);
			}
			t->has_limiter = 1;
		}
		else if(c->fm_mono_preemph != 0) REFUSE("FM sound pre-emphasis %d (the reference has 50 us, 75 us and J.17)", c->fm_mono_preemph);

		t->k.has_carriers = 1;
	}

	/* AM sound carrier (src/video.c:4550-4558, :2343-2357) */
	if(c->am_audio_level > 0 && c->am_mono_carrier != 0)
	{
		t->am_level = (int16_t) round(INT16_MAX * (c->am_audio_level * slevel));
		t->am_delta = _unit_phasor(2.0 * M_PI / t->sample_rate * c->am_mono_carrier);
		t->k.has_carriers = 1;
	}

	/* Zweikanalton (src/video.c:4375-4400): a second FM carrier derived from the first (-7 dB,
	 * 242.1875 kHz above it; 224.213 kHz on system M), a 54.6875 kHz pilot amplitude modulated
	 * with the 117.5 Hz "stereo" identification tone. NICAM is switched off. */
	if(c->a2stereo && t->fm_lut)
	{
		int r;
		double carrier;

		t->a2_system_m = c->fm_mono_carrier == 4500000;
		carrier = c->fm_mono_carrier + (t->a2_system_m ? 224213 : 242187.5);

		t->a2_level = (int16_t) round(INT16_MAX * ((c->fm_mono_level * 0.446684) * slevel));
		t->a2_lut = malloc(sizeof(hvk_c32_t) * 65536);
		if(!t->a2_lut) return(HVK_OUT_OF_MEMORY);
		for(r = INT16_MIN; r <= INT16_MAX; r++)
		{
			double d = 2.0 * M_PI / t->sample_rate * (carrier + (double) r / INT16_MAX * c->fm_mono_deviation);
			t->a2_lut[r - INT16_MIN] = _unit_phasor(d);
		}

		t->a2_pilot_level = (int16_t) round(INT16_MAX * 0.05);
		t->a2_pilot_delta = _unit_phasor(2.0 * M_PI / t->sample_rate * (t->a2_system_m ? 55.06993e3 : 54.6875e3));
		t->a2_signal_level = (int16_t) round(INT16_MAX * 1.0);
		t->a2_signal_delta = _unit_phasor(2.0 * M_PI / t->sample_rate * (t->a2_system_m ? 149.9 : 117.5));
	}

	/* NICAM-728 (src/video.c:4522-4533, src/nicam728.c:257-331) */
	if(c->nicam_level > 0 && c->nicam_carrier != 0 && !c->a2stereo)
	{
		unsigned int sr = t->sample_rate, freq = c->nicam_carrier, g;
		double sps = (double) sr / 364000.0;
		double level = c->nicam_level * slevel;
		double d;
		int h, x;

		t->k.nicam_ntaps = ((unsigned int) (sps * 5) + 1) | 1;
		t->nicam_taps = malloc(sizeof(int16_t) * t->k.nicam_ntaps);
		if(!t->nicam_taps) return(HVK_OUT_OF_MEMORY);

		h = t->k.nicam_ntaps / 2;
		for(x = -h; x <= h; x++)
		{
			double tt = ((double) x) / sps;
			double hx = (double) x / h;
			double win = (hx < -1 || hx > 1) ? 0 : 0.54 - 0.46 * cos((M_PI * (1.0 + hx)));
			double r = _root_raised_cosine(tt, c->nicam_beta, 1.0) * win;
			r *= M_SQRT1_2 * INT16_MAX * level;
			t->nicam_taps[x + h] = lround(r);
		}

		g = _gcd_u(sr, 364000);
		t->k.nicam_decimation = 364000 / g;
		t->k.nicam_sps = (sr + 364000 - 1) / 364000;
		t->k.nicam_inv20 = (uint32_t) (((1u << 20) + t->k.nicam_sps - 1) / t->k.nicam_sps);      /* (hvk_device.h:nicam_add()) */
		t->k.nicam_dsl = (t->k.nicam_sps * t->k.nicam_decimation) % (sr / g);

		g = _gcd_u(sr, freq);
		t->k.nicam_cc_len = sr / g;
		t->nicam_cc = malloc(sizeof(hvk_c16_t) * t->k.nicam_cc_len);
		if(!t->nicam_cc) return(HVK_OUT_OF_MEMORY);

		d = 2.0 * M_PI / t->k.nicam_cc_len * (freq / g);
		for(x = 0; x < t->k.nicam_cc_len; x++)
		{
			t->nicam_cc[x].i = round(cos(d * x) * 1.0 * INT16_MAX);
			t->nicam_cc[x].q = round(sin(d * x) * 1.0 * INT16_MAX);
		}

		t->k.has_nicam = 1;
	}

	return(HVK_OK);
}

/* ------------------------------------------------------------------ */
/* teletext                                                            */

/* 360 raised-cosine data symbols (beta 0.7) at 444 x line rate, 66 % of
 * white - black, first symbol 12 us less 12 bit periods after 0H: the table
 * tt_init() asks vbidata_init() for (src/teletext.c:1057-1074,
 * src/vbidata.c:26-35, :83-121). Each symbol keeps the samples from its first
 * to its last non-zero value. */
static int _build_teletext(hvk_tables_t *t)
{
	const int W = t->k.width;
	int level = round((t->white_level - t->black_level) * 0.66);
	double bw = (double) W / 444;
	double offset = t->pixel_rate * (12e-6 - (64e-6 / 444 * 12));
	int16_t *row = malloc(W * sizeof(int16_t));
	int b, x, total = 0, pass;

	t->tt_symbols = calloc(360 * 3, sizeof(int32_t));
	if(!row || !t->tt_symbols) { free(row); return(HVK_OUT_OF_MEMORY); }

	for(pass = 0; pass < 2; pass++)
	{
		if(pass == 1)
		{
			t->tt_total = total;
			t->tt_values = calloc(total + 8, sizeof(int16_t));
			if(!t->tt_values) { free(row); return(HVK_OUT_OF_MEMORY); }
			total = 0;
		}

		for(b = 0; b < 360; b++)
		{
			double t0 = -bw * b - offset;
			int first = 0, len = 0;

			for(x = 0; x < W; x++)
			{
				double u = (t0 + x) / bw, h;
				if(u == 0) h = 1.0;
				else h = (sin(M_PI * (u / 1)) / (M_PI * (u / 1))) * (cos(M_PI * 0.7 * u / 1) / (1.0 - (4.0 * 0.7 * 0.7 * u * u / (1 * 1))));
				row[x] = round(h * level);
				if(row[x] == 0) continue;
				if(len == 0) first = x;
				len = x - first + 1;
			}

			if(pass == 1)
			{
				t->tt_symbols[b * 3 + 0] = first;
				t->tt_symbols[b * 3 + 1] = len;
				t->tt_symbols[b * 3 + 2] = total;
				memcpy(t->tt_values + total, row + first, len * sizeof(int16_t));
			}
			total += len;
		}
	}

	free(row);
	t->k.teletext = 1;
	return(HVK_OK);
}

/* ------------------------------------------------------------------ */
/* SECAM                                                               */

/* Kaiser-windowed band stop with unity DC gain (src/fir.c:179-228) */
static void _design_band_reject(double *taps, int ntaps, double sample_rate, double low, double high)
{
	const double beta = 7.0;
	double inv_i0 = 1.0 / _bessel_i0(beta);
	double inm1 = 1.0 / ((double) (ntaps - 1));
	double w0 = 2.0 * M_PI * low / sample_rate, w1 = 2.0 * M_PI * high / sample_rate;
	double dc, gain = 1.0;
	int M = (ntaps - 1) / 2, i, n;

	taps[0] = taps[ntaps - 1] = inv_i0;
	for(i = 1; i < ntaps - 1; i++)
	{
		double temp = 2 * i * inm1 - 1;
		taps[i] = _bessel_i0(beta * sqrt(1.0 - temp * temp)) * inv_i0;
	}

	for(n = -M; n <= M; n++)
	{
		if(n == 0) taps[n + M] *= 1.0 + (w0 - w1) / M_PI;
		else taps[n + M] *= (sin(n * w0) - sin(n * w1)) / (n * M_PI);
	}

	dc = taps[M];
	for(n = 1; n <= M; n++) dc += 2 * taps[n + M];

	gain /= dc;
	for(n = 0; n < ntaps; n++) taps[n] *= gain;
}

static int _build_secam(hvk_tables_t *t, double level)
{
	const hvk_config_t *c = &t->conf;
	const double fm_dev = 1000e3, fm_freq = 4328125, cb = 4250000, cr = 4406250;   /* src/video.c:45-48 */
	double amp = (c->white_level - c->blanking_level) * level;
	double taps[51], sum, rise;
	int r, i;

	t->k.secam = 1;

	/* FM steps at the pixel rate, +/- 1 MHz full scale (src/video.c:4080, :2218-2243) */
	t->secam_level = (int16_t) round(INT16_MAX * amp);
	t->secam_lut = malloc(sizeof(hvk_c32_t) * 65536);
	t->secam_bell = malloc(sizeof(hvk_c16_t) * 65536);
	if(!t->secam_lut || !t->secam_bell) return(HVK_OUT_OF_MEMORY);

	for(r = INT16_MIN; r <= INT16_MAX; r++)
	{
		double d = 2.0 * M_PI / t->pixel_rate * (fm_freq + (double) r / INT16_MAX * fm_dev);
		double f, lq, rq, den;

		t->secam_lut[r - INT16_MIN] = _unit_phasor(d);

		/* "bell" filter: complex gain at the instantaneous frequency
		 * (src/video.c:2172-2185, :4122-4128) */
		f = fm_freq + (double) r * fm_dev / INT16_MAX;
		f = f / 4.286e6 - 4.286e6 / f;
		lq = 16.0 * f;
		rq = 1.26 * f;
		den = 1.0 + rq * rq;
		t->secam_bell[(uint16_t) r].i = lround(0.115 * (1.0 + lq * rq) / den * INT16_MAX);
		t->secam_bell[(uint16_t) r].q = lround(0.115 * (lq - rq) / den * INT16_MAX);
	}

	/* colour-difference low pass (src/video.c:4097-4098) */
	_design_low_pass(taps, 15, t->pixel_rate, 1.70e6);
	t->secam_fir = _q15_applied(taps, 15, 1);

	/* luma notch at the sub-carrier, deliberately weakened (src/video.c:4100-4107) */
	_design_band_reject(taps, 51, t->pixel_rate, fm_freq - 1e6, fm_freq + 1e6);
	taps[51 / 2] += 0.5;
	for(sum = i = 0; i < 51; i++) sum += taps[i];
	sum = sum / 1.0;
	for(i = 0; i < 51; i++) taps[i] /= sum;
	t->secam_notch = _q15_applied(taps, 51, 1);
	if(!t->secam_fir || !t->secam_notch) return(HVK_OUT_OF_MEMORY);

	/* deviation limits: [0] D'b lines, [1] D'r lines (src/video.c:4110-4113) */
	/* field identification lines (src/video.c:4130-4137) */
	t->secam_fsync_level = (int16_t) round(350e3 / fm_dev * INT16_MAX);
	t->secam_fid_lines = c->secam_field_id_lines;
	if(t->secam_fid_lines < 1 || t->secam_fid_lines > 9) t->secam_fid_lines = 9;

	t->secam_dmin[0] = lround((cb - fm_freq - 350e3) / fm_dev * INT16_MAX);
	t->secam_dmax[0] = lround((cb - fm_freq + 506e3) / fm_dev * INT16_MAX);
	t->secam_dmin[1] = lround((cr - fm_freq - 506e3) / fm_dev * INT16_MAX);
	t->secam_dmax[1] = lround((cr - fm_freq + 350e3) / fm_dev * INT16_MAX);

	/* sub-carrier envelope over the line (src/video.c:4140-4147) */
	rise = c->burst_rise * EDGE_0_100;
	t->k.burst_left = round(t->pixel_rate * (c->burst_left - c->burst_rise / 2));
	t->k.burst_width = ceil(t->pixel_rate * (c->burst_width + rise));
	t->burst_win = malloc(t->k.burst_width * sizeof(int16_t));
	if(!t->burst_win) return(HVK_OUT_OF_MEMORY);
	for(i = 0; i < t->k.burst_width; i++)
	{
		double tt = 1.0 / t->pixel_rate * i;
		t->burst_win[i] = round(_window(tt, rise / 2, c->burst_width, rise) * 1.0 * INT16_MAX);
	}

	return(HVK_OK);
}

/* ---- step-shaped vbidata tables (src/vbidata.c:58-81, :145-184): symbol b is a pulse of
 * `width` samples at offset + width * b with integrated raised-cosine edges, exactly the
 * quantiser the sync pulses use ---- */
static int _append_step_lut(hvk_tables_t *t, int lut, int nsymbols, int level, double width, double rise, double offset)
{
	int b, first, total = 0;
	int32_t *sym;
	int16_t *val;

	for(b = 0; b < nsymbols; b++) total += _quantise_pulse(NULL, &first, offset + width * b, width, rise, level);

	sym = realloc(t->vbi_sym, (size_t) (t->vbi_nsym + nsymbols) * 3 * sizeof(int32_t));
	if(!sym) return(HVK_OUT_OF_MEMORY);
	t->vbi_sym = sym;
	val = realloc(t->vbi_val, (size_t) (t->vbi_total + total + 8) * sizeof(int16_t));
	if(!val) return(HVK_OUT_OF_MEMORY);
	t->vbi_val = val;

	t->lut_base[lut] = t->vbi_nsym;
	t->lut_nsym[lut] = nsymbols;

	for(b = 0; b < nsymbols; b++)
	{
		int len = _quantise_pulse(t->vbi_val + t->vbi_total, &first, offset + width * b, width, rise, level);
		sym[(t->vbi_nsym + b) * 3 + 0] = len ? first : 0;
		sym[(t->vbi_nsym + b) * 3 + 1] = len;
		sym[(t->vbi_nsym + b) * 3 + 2] = t->vbi_total;
		t->vbi_total += len;
	}
	memset(t->vbi_val + t->vbi_total, 0, 8 * sizeof(int16_t));
	t->vbi_nsym += nsymbols;
	return(HVK_OK);
}

/* teletext's raised-cosine table joins the same store */
static int _append_teletext_lut(hvk_tables_t *t)
{
	int32_t *sym = realloc(t->vbi_sym, (size_t) (t->vbi_nsym + 360) * 3 * sizeof(int32_t));
	int16_t *val;
	int b;

	if(!sym) return(HVK_OUT_OF_MEMORY);
	t->vbi_sym = sym;
	val = realloc(t->vbi_val, (size_t) (t->vbi_total + t->tt_total + 8) * sizeof(int16_t));
	if(!val) return(HVK_OUT_OF_MEMORY);
	t->vbi_val = val;

	t->lut_base[0] = t->vbi_nsym;
	t->lut_nsym[0] = 360;
	for(b = 0; b < 360; b++)
	{
		sym[(t->vbi_nsym + b) * 3 + 0] = t->tt_symbols[b * 3 + 0];
		sym[(t->vbi_nsym + b) * 3 + 1] = t->tt_symbols[b * 3 + 1];
		sym[(t->vbi_nsym + b) * 3 + 2] = t->vbi_total + t->tt_symbols[b * 3 + 2];
	}
	memcpy(t->vbi_val + t->vbi_total, t->tt_values, t->tt_total * sizeof(int16_t));
	t->vbi_total += t->tt_total;
	memset(t->vbi_val + t->vbi_total, 0, 8 * sizeof(int16_t));
	t->vbi_nsym += 360;
	return(HVK_OK);
}

/* Widescreen signalling (src/wss.c:46-137): run-in and start code, then four groups of
 * bi-phase coded bits; 137 step symbols of 200 ns starting 11 us after 0H, 5/7 of white */
static void _wss_group(uint8_t *vbi, uint8_t code, int *offset, int length)
{
	int i, o = *offset;

	while(length--)
	{
		for(i = 0; i < 6; i++, o++)
		{
			const int b = 7 - (o % 8);
			if(i == 3) code ^= 1;
			vbi[o / 8] &= ~(1 << b);
			vbi[o / 8] |= (code & 1) << b;
		}
		code >>= 1;
	}

	*offset = o;
}

static int _build_wss(hvk_tables_t *t)
{
	static const uint8_t lead[7] = { 0xF8, 0xE3, 0x8E, 0x38, 0xF1, 0xE0, 0xF8 };
	const hvk_config_t *c = &t->conf;
	int level = round((t->white_level - t->black_level) * (5.0 / 7.0));
	int o = 29 + 24, r;

	if(c->lines != 625) REFUSE("widescreen signalling needs a 625-line mode (src/hacktv.c:1350-1356)");            /* src/hacktv.c: 625-line modes only */
	if(c->wss != 0xFF && (c->wss < 0 || c->wss > 0x0F)) REFUSE("WSS mode byte 0x%X (src/wss.c:33-44 has 0x01 .. 0x0E and auto)", c->wss);

	r = _append_step_lut(t, 1, 137, level, (double) t->pixel_rate * 200e-9, (double) t->pixel_rate * 200e-9, (double) t->pixel_rate * 11e-6);
	if(r != HVK_OK) return(r);

	memset(t->wss_bits, 0, sizeof(t->wss_bits));
	memcpy(t->wss_bits, lead, sizeof(lead));
	_wss_group(t->wss_bits, c->wss, &o, 4);     /* aspect ratio ("auto": rewritten per frame, hvk_wss_bits) */
	_wss_group(t->wss_bits, 0x00, &o, 4);       /* enhanced services */
	_wss_group(t->wss_bits, 0x00, &o, 3);       /* subtitles */
	_wss_group(t->wss_bits, 0x00, &o, 3);       /* reserved */

	/* 42.5 us of the line are blanked first, from mid-line (sic, src/wss.c:176-182) */
	t->wss_blank_lo = t->k.half_width;
	t->wss_blank_hi = round(t->pixel_rate * 42.5e-6);
	return(HVK_OK);
}

/* Line 23's bits for a frame whose source has pixel aspect par_num / par_den. "auto" (0xFF,
 * src/wss.c:166-179) signals 4:3 up to the pixel aspect at which the active area is 14:9 wide,
 * 16:9 beyond; every other mode ignores the source. */
void hvk_wss_bits(const hvk_tables_t *t, int64_t par_num, int64_t par_den, uint8_t bits[18])
{
	memcpy(bits, t->wss_bits, 18);
	if(t->conf.wss == 0xFF)
	{
		/* threshold = (14 / 9) / (active_width / active_lines), compared as fractions (src/common.c:76-80) */
		const int64_t tn = (int64_t) 14 * t->k.active_lines, td = (int64_t) 9 * t->k.active_width;
		const int64_t c = par_num * td - par_den * tn;
		int o = 29 + 24;
		_wss_group(bits, c <= 0 ? 0x08 : 0x07, &o, 4);
	}
}

/* Vertical interval time code (src/vitc.c:42-112): 116 (625) / 115 (525) step symbols per line */
static int _build_vitc(hvk_tables_t *t)
{
	const hvk_config_t *c = &t->conf;
	int hr, level = round((t->white_level - t->black_level) * 0.785);

	if(c->type == HVK_RASTER_625) { t->vitc_lines[0] = 19; t->vitc_lines[1] = 332; hr = 116; }
	else { t->vitc_lines[0] = 14; t->vitc_lines[1] = 277; hr = 115; }

	if(c->frame_rate.num <= 30 && c->frame_rate.den == 1) { t->vitc_fps = c->frame_rate.num; t->vitc_drop = 0; }
	else if(c->frame_rate.num == 30000 && c->frame_rate.den == 1001) { t->vitc_fps = 30; t->vitc_drop = 1; }
	else REFUSE("VITC time code at a frame rate of %d/%d (src/vitc.c:135-152: whole rates up to 30, and 30000/1001)", (int) c->frame_rate.num, (int) c->frame_rate.den);

	return(_append_step_lut(t, 2, hr, level, (double) t->k.width / hr, t->pixel_rate * 200e-9, 0));
}

static int _put_bits(uint8_t *data, int offset, uint64_t bits, int nbits)
{
	for(; nbits; nbits--, offset++, bits >>= 1)
	{
		if(bits & 1) data[offset >> 3] |= 1 << (offset & 7);
		else data[offset >> 3] &= ~(1 << (offset & 7));
	}
	return(offset);
}

int hvk_vitc_bits(const hvk_tables_t *t, int frame, int line, uint8_t data[12])
{
	const int fps = t->vitc_fps, second_field = line >= t->vitc_lines[1];
	uint32_t tc;
	uint8_t crc = 0;
	int fn = frame, x = 0, i;

	if(t->vitc_drop)
	{
		/* drop-frame numbering for 29.97 fps */
		fn += (fn / 17982) * 18;
		fn += (fn % 18000 - 2) / 1798 * 2;
	}

	tc  = (fn % fps % 10) << 0;
	tc |= (fn % fps / 10) << 4;
	tc |= (t->vitc_drop ? 1 : 0) << 6;
	tc |= 1 << 7;                               /* colour framing */
	fn /= fps;
	tc |= (fn % 10) << 8;
	tc |= (fn / 10 % 6) << 12;
	if(t->conf.type != HVK_RASTER_625) tc |= (uint32_t) second_field << 15;
	fn /= 60;
	tc |= (fn % 10) << 16;
	tc |= (fn / 10 % 6) << 20;
	fn /= 60;
	tc |= (fn % 24 % 10) << 24;
	tc |= (uint32_t) (fn % 24 / 10) << 28;
	if(t->conf.type == HVK_RASTER_625) tc |= (uint32_t) second_field << 31;

	memset(data, 0, 12);
	for(i = 0; i < 8; i++)
	{
		x = _put_bits(data, x, 0x01, 2);        /* sync */
		x = _put_bits(data, x, tc >> (i * 4), 4);
		x = _put_bits(data, x, 0, 4);           /* user bits */
	}
	x = _put_bits(data, x, 0x01, 2);
	_put_bits(data, x, 0, 8);
	for(i = 0; i < 11; i++) crc ^= data[i];
	crc = ((crc << 6) | (crc >> 2)) & 0xFF;
	x = _put_bits(data, x, crc, 8);
	return(x);
}

/* CEA/EIA-608 captions (src/cc608.c:97-160): 32 bit cells of width / 32 starting 27.5 us
 * (27.382 us on 525 lines) after 0H at half of white - black, and in front of them seven
 * cycles of clock run-in -- a fixed waveform, kept here as a 33rd symbol whose bit is always set */
static int _build_cc608(hvk_tables_t *t)
{
	const hvk_config_t *c = &t->conf;
	const int W = t->k.width;
	const double offset = c->type == HVK_RASTER_525 ? 27.382e-6 : 27.5e-6;
	const double level = round((t->white_level - t->black_level) * 0.5);
	const double w = (double) W * 7 / 32;
	const double x = (double) t->pixel_rate * offset - (W * 8.75 / 32);
	const int cri_x = x, cri_len = ceil(w);
	int32_t *sym;
	int16_t *val;
	int i, r;

	t->cc608_line = c->type == HVK_RASTER_525 ? 21 : 22;

	r = _append_step_lut(t, 3, 32, (int) level, (double) W / 32, t->pixel_rate * 240e-9 * EDGE_0_100, t->pixel_rate * offset);
	if(r != HVK_OK) return(r);

	sym = realloc(t->vbi_sym, (size_t) (t->vbi_nsym + 1) * 3 * sizeof(int32_t));
	if(!sym) return(HVK_OUT_OF_MEMORY);
	t->vbi_sym = sym;
	val = realloc(t->vbi_val, (size_t) (t->vbi_total + cri_len + 8) * sizeof(int16_t));
	if(!val) return(HVK_OUT_OF_MEMORY);
	t->vbi_val = val;

	sym[t->vbi_nsym * 3 + 0] = cri_x;
	sym[t->vbi_nsym * 3 + 1] = cri_len;
	sym[t->vbi_nsym * 3 + 2] = t->vbi_total;
	for(i = 0; i < cri_len; i++)
	{
		/* truncated, not rounded (src/cc608.c:150) */
		val[t->vbi_total + i] = (0.5 - cos(((double) i - (x - cri_x)) * (2 * M_PI / w * 7)) * 0.5) * level;
	}
	t->vbi_total += cri_len;
	memset(val + t->vbi_total, 0, 8 * sizeof(int16_t));
	t->vbi_nsym += 1;
	t->lut_nsym[3] = 33;
	return(HVK_OK);
}

void hvk_cc608_bits(uint8_t c1, uint8_t c2, uint8_t data[3])
{
	int i;

	/* odd parity in bit 7 */
	c1 = (c1 & 0x7F) | 0x80;
	c2 = (c2 & 0x7F) | 0x80;
	for(i = 1; i < 8; i++)
	{
		c1 ^= (c1 << i) & 0x80;
		c2 ^= (c2 << i) & 0x80;
	}

	/* a start bit, then the two characters */
	data[0] = (c1 << 1) | 0x01;
	data[1] = (c2 << 1) | (c1 >> 7);
	data[2] = (c2 >> 7);
}

/* Anti-copy pulses (src/acp.c:26-64): six P-sync / AGC pulse pairs per line; levels are assigned */
static void _build_acp(hvk_tables_t *t)
{
	const int is625 = t->conf.lines == 625;
	const double left = is625 ? 8.88e-6 : 8.288e-6;
	const double spacing = is625 ? 5.92e-6 : 8.288e-6;
	const double psync_width = is625 ? 2.368e-6 : 2.222e-6;
	const hvk_yuvparams_t *p = &t->yuv;
	int i;

	t->acp_psync_level = (int16_t) (t->sync_level + round((t->white_level - t->sync_level) * 0.06));
	t->acp_psync_width = round(t->pixel_rate * psync_width);
	t->acp_pagc_width = round(t->pixel_rate * 2.7e-6);
	for(i = 0; i < 6; i++) t->acp_left[i] = round(t->pixel_rate * (left + spacing * i));

	/* luma of the greys, as the level table has it (src/video.c:3917-3958) */
	for(i = 0; i < 256; i++)
	{
		double g = p->glut[i];
		double y = g * p->rw + g * p->gw + g * p->bw;
		y = (p->black + (y * p->range)) * p->level;
		y = y < -1 ? -1 : (y > 1 ? 1 : y);
		t->grey_y[i] = (int16_t) round(y * INT16_MAX);
	}
}

int hvk_acp_agc_level(const hvk_tables_t *t, int frame)
{
	/* a clipped sawtooth over 428 frames (src/acp.c:80-89) */
	int i = abs(frame * 4 % 1712 - 856) - 150;
	if(i < 0) i = 0;
	else if(i > 255) i = 255;
	return((int16_t) (t->sync_level + round((t->grey_y[i] - t->sync_level) * 1.10)));
}

/* Insertion test signals (src/vits.c:53-296): per line a luma waveform that is added and a
 * chroma amplitude that rides on the line's sub-carrier */
static double _sin2_pulse(double t, double position, double width, double amplitude)
{
	double a;
	t -= position - width;
	if(t <= 0 || t >= width * 2) return(0);
	a = t / (width * 2) * M_PI;
	return(pow(sin(a), 2) * amplitude);
}

static int _build_vits(hvk_tables_t *t)
{
	static const double b625[6] = { 0.5e6, 1.0e6, 2.0e6, 4.0e6, 4.8e6, 5.8e6 };
	static const double b525[6] = { 0.50e6, 1.00e6, 2.00e6, 3.00e6, 3.58e6, 4.20e6 };
	const int W = t->k.width, is625 = t->conf.lines == 625;
	const int n = is625 ? 4 : 2;
	const int level = t->white_level - t->blanking_level;   /* src/video.c:4217-4221 */
	const double unit = is625 ? 0.7 : 100;
	double ts, h, bs[6];
	int i, x, b;

	t->vits_l = calloc((size_t) n * W, sizeof(int16_t));
	t->vits_c = calloc((size_t) n * W, sizeof(int16_t));
	if(!t->vits_l || !t->vits_c) return(HVK_OUT_OF_MEMORY);

	ts = is625 ? 1.0 / 25 / 625 : 1001.0 / 30000 / 525;
	h = is625 ? ts / 32 : ts / 128;
	ts = ts / W;
	for(b = 0; b < 6; b++) bs[b] = 2.0 * M_PI * (is625 ? b625[b] : b525[b]);

	for(i = 0; i < n; i++)
	{
		for(x = 0; x < W; x++)
		{
			double tt = ts * x, r = 0.0, c = 0.0;

			if(is625 && i == 0)                 /* line 17 */
			{
				r += _window(tt, 6 * h, 5 * h, 200e-9) * 0.70;
				r += _sin2_pulse(tt, 13 * h, 200e-9, 0.70);
				r += _sin2_pulse(tt, 16 * h, 2000e-9, 0.70 / 2);
				c += _sin2_pulse(tt, 16 * h, 2000e-9, 0.70 / 2);
				r += _window(tt, 20 * h, 2 * h, 200e-9) * 0.14;
				r += _window(tt, 22 * h, 2 * h, 200e-9) * 0.28;
				r += _window(tt, 24 * h, 2 * h, 200e-9) * 0.42;
				r += _window(tt, 26 * h, 2 * h, 200e-9) * 0.56;
				r += _window(tt, 28 * h, 3 * h, 200e-9) * 0.70;
			}
			else if(is625 && i == 1)            /* line 18 */
			{
				r += _window(tt, 6 * h, 25 * h, 200e-9) *  0.35;
				r += _window(tt, 6 * h,  2 * h, 200e-9) *  0.21;
				r += _window(tt, 8 * h,  2 * h, 200e-9) * -0.21;
				for(b = 0; b < 6; b++)
				{
					r += _window(tt, (12 + 3 * b) * h, 2 * h, 200e-9) * 0.21
					   * sin((tt - (12 + 3 * b) * h) * bs[b]);
				}
			}
			else if(is625 && i == 2)            /* line 330 */
			{
				r += _window(tt, 6 * h, 5 * h, 200e-9) * 0.70;
				r += _sin2_pulse(tt, 13 * h, 200e-9, 0.70);
				c += _window(tt, 15 * h, 15 * h, 1e-6) * 0.28 / 2;
				r += _window(tt, 20 * h, 2 * h, 200e-9) * 0.14;
				r += _window(tt, 22 * h, 2 * h, 200e-9) * 0.28;
				r += _window(tt, 24 * h, 2 * h, 200e-9) * 0.42;
				r += _window(tt, 26 * h, 2 * h, 200e-9) * 0.56;
				r += _window(tt, 28 * h, 3 * h, 200e-9) * 0.70;
			}
			else if(is625)                      /* line 331 */
			{
				r += _window(tt, 6 * h, 25 * h, 200e-9) * 0.35;
				c += _window(tt, 7 * h, 7 * h, 1e-6) * 0.70 / 2;
				c += _window(tt, 17 * h, 13 * h, 1e-6) * 0.42 / 2;
			}
			else if(i == 0)                     /* 525: line 17 */
			{
				r += _window(tt, 24 * h, 36 * h, 125e-9) * 100;
				r += _sin2_pulse(tt, 68 * h, 250e-9, 100);
				r += _sin2_pulse(tt, 75 * h, 1570e-9, 100 / 2);
				c += _sin2_pulse(tt, 75 * h, 1570e-9, 100 / 2);
				r += _window(tt,  92 * h,  6 * h, 250e-9) * 18;
				r += _window(tt,  98 * h,  6 * h, 250e-9) * 36;
				r += _window(tt, 104 * h,  6 * h, 250e-9) * 54;
				r += _window(tt, 110 * h,  6 * h, 250e-9) * 72;
				r += _window(tt, 116 * h,  8 * h, 250e-9) * 90;
				c += _window(tt,  84 * h, 38 * h, 400e-9) * 40 / 2;
			}
			else                                /* 525: line 280 */
			{
				r += _window(tt, 24 * h, 8 * h, 125e-9) * 100;
				r += _window(tt, 32 * h, 92 * h, 125e-9) * 50;
				r += _window(tt, 36 * h, 12 * h, 250e-9) * 50 / 2 * sin((tt - 36 * h) * bs[0]);
				for(b = 1; b < 6; b++)
				{
					r += _window(tt, (40 + 8 * b) * h, 8 * h, 250e-9) * 50 / 2
					   * sin((tt - (40 + 8 * b) * h) * bs[b]);
				}
				c += _window(tt,  92 * h,  8 * h, 400e-9) * 20 / 2;
				c += _window(tt, 100 * h,  8 * h, 400e-9) * 40 / 2;
				c += _window(tt, 108 * h, 12 * h, 400e-9) * 80 / 2;
			}

			t->vits_l[(size_t) i * W + x] = lround(r / unit * level);
			t->vits_c[(size_t) i * W + x] = lround(c / unit * level);
		}
	}

	t->k.vits = n;
	if(is625) { t->k.vits_line[0] = 16; t->k.vits_line[1] = 17; t->k.vits_line[2] = 329; t->k.vits_line[3] = 330; }
	else { t->k.vits_line[0] = 16; t->k.vits_line[1] = 279; }

	if(t->conf.colour_mode == HVK_PAL)
	{
		/* 60 degrees from the +(B-Y) axis */
		const double p = 60.0 * (M_PI / 180.0);
		t->k.vits_pi = (int16_t) round(cos(p) * INT16_MAX);
		t->k.vits_pq = (int16_t) round(sin(p) * INT16_MAX);
	}
	else
	{
		t->k.vits_pi = 0;
		t->k.vits_pq = -INT16_MAX;
	}

	return(HVK_OK);
}

/* ------------------------------------------------------------------ */

/* ---- sound-in-syncs (src/sis.c:36-153) ---- */

static double _sis_rc(double x)
{
	if(x <= -1 || x >= 1) return(0);
	return((1.0 + cos(M_PI * x)) / 2);
}

static int _build_sis(hvk_tables_t *t)
{
	const int W = t->k.width;
	const double bwidth = (double) W / 382, offset = (double) W / 382 * 3.32;      /* (the offset: "measured", src/sis.c:113) */
	const double left = 0.2e-6, rise = 80e-9, width = 4.56e-6;
	int levels[2], b, x, off[50], len[50];
	int16_t *packed;
	long pos[50], n = 0;

	{
		const int level = (int) round((double) (t->white_level - t->black_level));
		levels[0] = level / 2 / 0.75;
		levels[1] = level / 4 / 0.75;
	}

	t->sis_dense = calloc((size_t) 50 * HVK_SIS_SPAN, sizeof(int16_t));
	packed = calloc((size_t) 50 * (W + 2) + 1, sizeof(int16_t));
	if(!t->sis_dense || !packed) { free(packed); return(HVK_OUT_OF_MEMORY); }

	/* entry b shapes bit b of the burst: bits 2 n and 2 n + 1 share a place and weigh 2 : 1. Kept twice: as dense rows
	 * for the kernel, and packed the reference's way ([length][offset][values], from the first non-zero value to
	 * the last: vbidata_update(), src/vbidata.c:36-60) for what follows */
	for(b = 0; b < 50; b++)
	{
		const double tt = -bwidth * (b / 2) - offset;
		off[b] = len[b] = 0;
		pos[b] = n + 2;
		for(x = 0; x < W; x++)
		{
			const int v = (int) round(_sis_rc((tt + x) / bwidth) * levels[b & 1]);
			if(v == 0) continue;
			if(len[b] == 0) off[b] = x;
			while(len[b] < x - off[b]) packed[pos[b] + len[b]++] = 0;
			packed[pos[b] + len[b]++] = (int16_t) v;
			if(x >= HVK_SIS_SPAN) { free(packed); REFUSE("the sound-in-syncs burst at %u Hz is longer than the %d samples the kernel looks at", t->pixel_rate, HVK_SIS_SPAN); }       /* (a burst longer than the kernel looks at: not at any rate hvk_open takes) */
			t->sis_dense[(size_t) b * HVK_SIS_SPAN + x] = (int16_t) v;
		}
		packed[n] = (int16_t) len[b];
		packed[n + 1] = (int16_t) off[b];
		n = pos[b] + len[b];
	}
	packed[n++] = -1;

	/* the blanking window */
	t->k.sis_left = (int) floor(t->pixel_rate * (left - rise / 2));
	t->k.sis_width = (int) ceil(t->pixel_rate * (width + rise));
	if(t->k.sis_left < 0 || t->k.sis_left + t->k.sis_width > HVK_SIS_SPAN) { free(packed); REFUSE("the sound-in-syncs blanking window at %u Hz does not fit the %d samples the kernel looks at", t->pixel_rate, HVK_SIS_SPAN); }
	t->sis_win = calloc(t->k.sis_width, sizeof(int16_t));
	t->sis_first = calloc(HVK_SIS_SPAN, sizeof(int16_t));
	if(!t->sis_win || !t->sis_first) { free(packed); return(HVK_OUT_OF_MEMORY); }
	for(x = t->k.sis_left; x < t->k.sis_left + t->k.sis_width; x++)
	{
		t->sis_win[x - t->k.sis_left] = (int16_t) round(_window(1.0 / t->pixel_rate * x, left, width, rise) * INT16_MAX);
	}
	t->k.sis_sync = t->sync_level;

	/* The process runs on the never-emitted slots in front of line 1 as well -- one, or three where the colour process
	 * is a thread between the raster and it (SECAM, src/video.c:4211 with :3543-3583). The last of them has no width
	 * and line 1's slot behind it: vbidata_render() then draws every set symbol into LINE 1, from its sample 0 on, with a
	 * negative index into the symbol's values (src/vbidata.c:211-217) -- the symbol lands in its place and the samples
	 * in front of it get what the packed table holds in front of the values: earlier entries, the entry's own header,
	 * and for the first entries the 16 bytes in front of the table on the reference's heap (glibc: the chunk's size
	 * word, zeros before it). Line 1's own invocation blanks most of it away; the stream's first samples keep it.
	 * The burst of those invocations is known: the frame store is still all zeros. */
	t->k.sis_dummies = t->conf.colour_mode == HVK_SECAM ? 3 : 1;
	/* (--raw-bb-file: the process that reads the lines in works on ONE line, not on the raster's three, and this one
	 * shares its slot: src/video.c:4190 with :4676-4688 -- no slot in front of line 1, nothing left on it) */
	if(t->conf.raw_bb) t->k.sis_dummies = 0;
	if(t->k.sis_dummies > 0)
	{
		static const uint8_t gc[2][4] = { { 3, 0, 2, 1 }, { 0, 3, 1, 2 } };
		int re = 0, nb = 50, call;
		uint8_t vbi[7];
		int16_t heap[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		const unsigned long chunk = (((unsigned long) n * 2 + 8 + 15) & ~15UL) | 1;
		heap[4] = (int16_t) (chunk & 0xFFFF);
		heap[5] = (int16_t) ((chunk >> 16) & 0xFFFF);

		for(call = 1; call <= t->k.sis_dummies; call++)
		{
			nb = 50;
			if((re += 44) >= 125) { nb -= 4; re -= 125; }
		}
		memset(vbi, 0, sizeof(vbi));
		vbi[0] = 0xC0;
		for(x = 2; x < nb; x += 2) vbi[x >> 3] |= gc[(x & 4) ? 1 : 0][0] << (6 - (x & 7));

		for(b = 0; b < nb; b++)
		{
			const int e = 50 - nb + b;
			int i, at;
			if(!((vbi[b >> 3] >> (7 - (b & 7))) & 1)) continue;
			for(i = -off[e], at = 0; i < len[e] && at < HVK_SIS_SPAN; i++, at++)
			{
				const long q = pos[e] + i;
				const int16_t v = q >= 0 ? packed[q] : (q >= -8 ? heap[q + 8] : 0);
				t->sis_first[at] = (int16_t) (t->sis_first[at] + v);
			}
		}
	}

	free(packed);
	t->k.sis = 1;
	return(HVK_OK);
}

int hvk_tables_build(hvk_tables_t *t, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate)
{
	hvk_config_t *c;
	double line_s, level, slevel, sync_amp;
	int i, r;

	memset(t, 0, sizeof(*t));
	t->conf = *conf;
	t->sample_rate = sample_rate;
	if(pixel_rate == 0) pixel_rate = sample_rate;    /* src/video.c:3839 */
	t->pixel_rate = pixel_rate;
	c = &t->conf;

	/* what the engine renders */
	{
		int nl;
		if(!_runs_of(c->type, &nl) || nl != c->lines) REFUSE("video type %d with %d lines: not one of the rasters of src/video.c:2447-2862 (D/D2-MAC has a line process of its own)", c->type, c->lines);
	}
	if(c->modulation == HVK_FM && (c->fm_level <= 0 || c->fm_deviation <= 0)) return(HVK_ERROR);
	if(c->frame_rate.num <= 0 || c->frame_rate.den <= 0) return(HVK_ERROR);

	/* defaults (src/video.c:3832-3836) */
	if(c->hline <= 0 && c->interlaced != 0) c->hline = (c->lines + 1) / 2;
	if(c->gamma <= 0) c->gamma = 1.0;
	if(c->rw_co <= 0) c->rw_co = 0.299;
	if(c->gw_co <= 0) c->gw_co = 0.587;
	if(c->bw_co <= 0) c->bw_co = 0.114;

	/* geometry (src/video.c:3844-3853); pixel rate == sample rate */
	line_s = (double) c->frame_rate.den / c->frame_rate.num / c->lines;
	t->k.width = round((double) pixel_rate * line_s);
	t->k.half_width = round((double) pixel_rate * line_s / 2);
	t->k.active_left = round(pixel_rate * c->active_left);
	t->k.active_width = ceil(pixel_rate * c->active_width);
	if(t->k.active_width > t->k.width) t->k.active_width = t->k.width;
	t->k.lines = c->lines;
	t->k.active_lines = c->active_lines;
	t->k.interlaced = c->interlaced;
	t->k.fields = (c->interlace && c->interlaced != 0) ? 2 : 1;   /* src/video.c:4873: needs a second field to load for */
	t->k.hline = c->hline;
	t->k.frame_samples = t->k.width * t->k.lines;
	t->k.raster_samples = t->k.frame_samples;
	t->k.slab_lines = t->k.lines + 2;
	t->max_width = t->k.width;

	if(t->k.width < 64 || t->k.width > 8192) REFUSE("lines of %d samples (%u Hz): the kernels take 64 .. 8192", t->k.width, t->pixel_rate);

	/* levels (src/video.c:3858-3881) */
	/* sub-carriers ride on the FM baseband at unit level; the overall level then scales the FM phasor */
	slevel = c->modulation == HVK_FM ? 1.0 : c->level;
	level = c->video_level * slevel;

	if(c->invert_video)
	{
		double w = c->white_level;
		c->white_level = c->sync_level;
		c->sync_level = w;
		c->blanking_level = c->sync_level - (c->blanking_level - c->white_level);
		c->black_level = c->sync_level - (c->black_level - c->white_level);
	}

	t->white_level    = (int16_t) round(c->white_level    * level * INT16_MAX);
	t->black_level    = (int16_t) round(c->black_level    * level * INT16_MAX);
	t->blanking_level = (int16_t) round(c->blanking_level * level * INT16_MAX);
	t->sync_level     = (int16_t) round(c->sync_level     * level * INT16_MAX);
	t->k.blanking = t->blanking_level;

	/* sync pulses (src/video.c:3884-3891): two passes, size then fill.
	 * The amplitude reaches the renderer as an int (truncated). */
	sync_amp = (c->sync_level - c->blanking_level) * level * INT16_MAX;
	{
		const double at[5]  = { 0, 0, 0, line_s / 2, line_s / 2 };
		const double len[5] = { c->hsync_width, c->vsync_short_width, c->vsync_long_width,
		                        c->vsync_short_width, c->vsync_long_width };
		double rise = c->sync_rise * EDGE_0_100 * pixel_rate;
		int total = HVK_PULSE_PAD, o = 0, first;      /* HVK_PULSE_PAD zeros in front of and behind every pulse: the kernel reads 8 values wherever a lane stands */

		t->k.npulses = 5;
		for(i = 0; i < 5; i++)
		{
			t->k.pulse_length[i] = _quantise_pulse(NULL, &first, at[i] * pixel_rate, len[i] * pixel_rate, rise, (int) sync_amp);
			t->k.pulse_offset[i] = first;
			t->k.pulse_start[i] = total;
			total += t->k.pulse_length[i] + HVK_PULSE_PAD;

			/* (a pulse may run on into the next line -- _build_linebase -- but no further) */
			if(first < -t->k.width) REFUSE("a sync pulse of this raster begins more than a line before its own line at %u Hz", t->pixel_rate);
		}

		t->pulse_total = total;
		t->pulse_values = calloc(total + 8, sizeof(int16_t));
		t->sync_packed = calloc(total + 2 * 5 + 1, sizeof(int16_t));
		if(!t->pulse_values || !t->sync_packed) return(HVK_OUT_OF_MEMORY);

		for(i = 0; i < 5; i++)
		{
			_quantise_pulse(t->pulse_values + t->k.pulse_start[i], &first, at[i] * pixel_rate, len[i] * pixel_rate, rise, (int) sync_amp);

			/* the same data in the reference's packed form, for table parity tests */
			t->sync_packed[o++] = t->k.pulse_length[i];
			t->sync_packed[o++] = t->k.pulse_offset[i];
			memcpy(t->sync_packed + o, t->pulse_values + t->k.pulse_start[i], t->k.pulse_length[i] * sizeof(int16_t));
			o += t->k.pulse_length[i];
		}
		t->sync_packed[o++] = -1;
		t->sync_packed_len = o;
	}

	/* RGB -> level conversion parameters (src/video.c:3905-3958); the 2^24
	 * entry table itself is expanded on the device */
	for(i = 0; i < 256; i++) t->yuv.glut[i] = pow((double) i / 255, 1 / c->gamma);
	t->yuv.rw = c->rw_co;
	t->yuv.gw = c->gw_co;
	t->yuv.bw = c->bw_co;
	t->yuv.eu = c->eu_co;
	t->yuv.ev = c->ev_co;
	t->yuv.black = c->black_level;
	t->yuv.range = c->white_level - c->black_level;
	t->yuv.level = level;
	t->yuv.chroma_scale = (c->white_level - c->black_level) * level;
	t->yuv.secam = c->colour_mode == HVK_SECAM;
	t->yuv.fast = 0;        /* (the engine's to set, once it has checked every colour: hvk_yuvparams_t) */
	t->yuv.f_y0 = c->black_level * level * 32767.0;
	t->yuv.f_y1 = (c->white_level - c->black_level) * level * 32767.0;
	if(t->yuv.secam)
	{
		t->yuv.f_u0 = (4250000.0 - 4328125.0) / 1000000.0 * 32767.0; t->yuv.f_u1 = c->eu_co / 1000000.0 * 32767.0;
		t->yuv.f_v0 = (4406250.0 - 4328125.0) / 1000000.0 * 32767.0; t->yuv.f_v1 = c->ev_co / 1000000.0 * 32767.0;
	}
	else
	{
		t->yuv.f_u0 = 0; t->yuv.f_u1 = c->eu_co * t->yuv.chroma_scale * 32767.0;
		t->yuv.f_v0 = 0; t->yuv.f_v1 = c->ev_co * t->yuv.chroma_scale * 32767.0;
	}
	{
		/* luma of black, used wherever the picture does not cover the active area */
		double y = (c->black_level + (0.0 * (c->white_level - c->black_level))) * level;
		y = y < -1 ? -1 : (y > 1 ? 1 : y);
		t->k.black_y = (int16_t) round(y * INT16_MAX);
	}

	/* colour sub-carrier (src/video.c:3961-4014) */
	t->k.colour = c->colour_mode == HVK_PAL || c->colour_mode == HVK_NTSC;
	if(t->k.colour)
	{
		/* samples per sub-carrier cycle as a reduced fraction num/den:
		 * the phase pattern repeats every `num` samples */
		int64_t num = (int64_t) pixel_rate * c->colour_carrier.den;
		int64_t den = c->colour_carrier.num;
		int64_t a = num, b = den, e, n;
		double step;

		if(den <= 0) return(HVK_ERROR);
		while((e = a % b)) { a = b; b = e; }
		num /= b;
		den /= b;
		if(num > 0x7FFFFFFF) REFUSE("the colour sub-carrier's table at %u Hz would hold more than 2^31 entries", t->pixel_rate);

		t->k.clw = num;
		step = 2.0 * M_PI * ((double) den / num);

		t->colour_lookup_len = num + t->k.width;
		t->colour_lookup = malloc(t->colour_lookup_len * sizeof(hvk_c16_t));
		if(!t->colour_lookup) return(HVK_OUT_OF_MEMORY);

		for(n = 0; n < t->colour_lookup_len; n++)
		{
			t->colour_lookup[n].i = round(cos(step * n) * INT16_MAX);
			t->colour_lookup[n].q = round(sin(step * n) * INT16_MAX);
		}

		if(c->colour_bw > 0)
		{
			double *taps;
			t->k.chroma_ntaps = _design_gaussian(&taps, pixel_rate, c->colour_bw);
			t->chroma_taps = _q15_applied(taps, t->k.chroma_ntaps, 1);
			free(taps);
			if(!t->chroma_taps) return(HVK_OUT_OF_MEMORY);
			if(t->k.chroma_ntaps / 2 * 2 > HVK_GHOST_LEN) REFUSE("a chroma filter of %d taps (pixel rate %u Hz) reads further past the line than the %d samples modelled", t->k.chroma_ntaps, t->pixel_rate, HVK_GHOST_LEN);
		}
		else
		{
			/* no chroma low pass (`ntsc-a`): the kernels' filter stage with "three taps", which stands for handing the
			 * window's middle element on as it is (fir8<3>, hvk_device.h); nothing is read past the line's end */
			t->k.chroma_ntaps = 3;
			t->chroma_taps = calloc(4, sizeof(int16_t));
			if(!t->chroma_taps) return(HVK_OUT_OF_MEMORY);
			t->chroma_taps[1] = INT16_MAX;
			t->chroma_unfiltered = 1;
		}
	}

	/* colour burst envelope (src/video.c:4017-4048, :2194-2214) */
	if(c->burst_level > 0)
	{
		double rise = c->burst_rise * EDGE_0_100;
		double amp = c->burst_level * (c->white_level - c->blanking_level) / 2 * level;

		t->k.burst_left = round(pixel_rate * (c->burst_left - c->burst_rise / 2));
		t->k.burst_width = ceil(pixel_rate * (c->burst_width + rise));
		t->burst_win = malloc(t->k.burst_width * sizeof(int16_t));
		if(!t->burst_win) return(HVK_OUT_OF_MEMORY);

		for(i = 0; i < t->k.burst_width; i++)
		{
			double tt = 1.0 / pixel_rate * i;
			t->burst_win[i] = round(_window(tt, rise / 2, c->burst_width, rise) * amp * INT16_MAX);
		}

		if(c->colour_mode == HVK_PAL)
		{
			double p = 135.0 * (M_PI / 180.0);
			t->k.burst_i = (int16_t) round(cos(p) * INT16_MAX);
			t->k.burst_q = (int16_t) round(sin(p) * INT16_MAX);
		}
		else if(c->colour_mode == HVK_NTSC)
		{
			t->k.burst_i = -INT16_MAX;
			t->k.burst_q = 0;
		}

		if(t->k.colour && t->k.burst_left + t->k.burst_width > t->k.width) REFUSE("the colour burst ends behind the line at %u Hz", t->pixel_rate);
	}
	else if(t->k.colour) REFUSE("colour without a sub-carrier frequency");

	if(c->colour_mode == HVK_APOLLO_FSC || c->colour_mode == HVK_CBS_FSC)
	{
		/* The flag pulse that marks one field of the colour sequence (src/video.c:4050-4073): rendered like the sync pulses,
		 * kept as dense rows the raster kernel adds on the flag lines -- Apollo: one pulse, CBS: one at the line's start and
		 * one half a line on */
		const double amp = (c->fsc_flag_level - c->blanking_level) * level * INT16_MAX;
		const double rise = c->sync_rise * EDGE_0_100 * pixel_rate;
		const int np = c->colour_mode == HVK_CBS_FSC ? 2 : 1;
		int16_t tmp[8192 + 64];
		int p, first, len, j;

		t->k.fsc_mode = c->colour_mode == HVK_APOLLO_FSC ? 1 : 2;
		t->k.fsc_split = c->colour_mode == HVK_APOLLO_FSC ? 264 : 202;
		t->fsc_rows = calloc((size_t) 2 * t->k.width + 32, sizeof(int16_t));
		if(!t->fsc_rows) return(HVK_OUT_OF_MEMORY);
		for(p = 0; p < np; p++)
		{
			const double at = (p ? line_s / 2 : 0) + c->fsc_flag_left;
			len = _quantise_pulse(NULL, &first, at * pixel_rate, c->fsc_flag_width * pixel_rate, rise, (int) amp);
			if(len > 8192 || first < 0 || first + len > t->k.width) REFUSE("the field-sequential colour flag does not lie inside its line at %u Hz", t->pixel_rate);   /* (the flag lies inside its line at every rate there is) */
			_quantise_pulse(tmp, &first, at * pixel_rate, c->fsc_flag_width * pixel_rate, rise, (int) amp);
			for(j = 0; j < len; j++) t->fsc_rows[(size_t) p * t->k.width + first + j] = tmp[j];
		}
	}

	if(c->colour_mode == HVK_SECAM && (r = _build_secam(t, level)) != HVK_OK) return(r);

	/* SECAM's luma notch runs over the active picture and looks 25 samples past its right edge
	 * (src/video.c:3206, src/fir.c:365-372). At pixel rates where that reaches beyond the line
	 * (13.5 MHz: 140 + 702 + 25 > 864) the reference reads behind its line buffer -- whatever the
	 * heap holds there. There is nothing to be exact to: refused. (S-Video has no notch.) */
	if(c->colour_mode == HVK_SECAM && !c->s_video && !c->raw_bb &&
	   t->k.active_left + t->k.active_width + 25 > t->k.width)
		REFUSE("SECAM at %u Hz: the reference's luma notch reads %d samples past its line buffer there (src/video.c:3206) -- its output depends on its heap", t->pixel_rate, t->k.active_left + t->k.active_width + 25 - t->k.width);

	hvk_tables_default_ghost(t);

	/* video filter (src/video.c:3653-3764) */
	t->k.vf_type = 0;
	t->k.delay_lines = 0;
	if(c->vfilter)
	{
		int ntaps = 51;
		const int fw = round((double) sample_rate * line_s);   /* the line width at the sample rate, src/video.c:3660 */

		if(c->modulation == HVK_FM)
		{
			/* FM video: a fixed pre-emphasis filter picked by line count and sample rate
			 * (src/video.c:3690-3730; hvk_fm_taps.h has the tables), real, in front of the modulator */
			const hvk_fm_taps_t *f = hvk_fm_taps;
			int k;
			/* (the reference tests for 525 lines and takes the 625-line tables for every other count -- Apollo's 320 lines
			 * among them, src/video.c:3693, :3711) */
			const int ll = c->lines == 525 ? 525 : 625;
			while(f->lines && !(f->lines == ll && (f->sample_rate == (int) sample_rate || f->sample_rate == 0))) f++;
			if(!f->lines || f->ntaps > HVK_MAX_VF_TAPS) REFUSE("no FM video pre-emphasis taps for %d lines at %u Hz (src/video.c:3452-3564)", c->lines, sample_rate);
			ntaps = f->ntaps;
			t->k.vf_type = 1;
			t->k.vf_ntaps = ntaps;
			t->vf_itaps = calloc(ntaps, sizeof(int16_t));
			if(!t->vf_itaps) return(HVK_OUT_OF_MEMORY);
			for(k = 0; k < ntaps; k++) t->vf_itaps[k] = f->q15[ntaps - 1 - k];      /* applied order: the design reversed (src/fir.c:279-286) */
		}
		else if(c->modulation == HVK_VSB)
		{
			/* complex band pass = low pass of half the pass band rotated to the
			 * band centre; the rotation phase is ACCUMULATED tap by tap
			 * (src/fir.c:230-255) */
			double lp[51], rot[51 * 2];
			double freq = M_PI * (c->vsb_upper_bw + -c->vsb_lower_bw) / (double) sample_rate;
			double phase = -freq * (ntaps >> 1);

			_design_low_pass(lp, ntaps, sample_rate, (c->vsb_upper_bw - -c->vsb_lower_bw) / 2);
			for(i = 0; i < ntaps; i++, phase += freq)
			{
				rot[i * 2 + 0] = lp[i] * cos(phase);
				rot[i * 2 + 1] = lp[i] * sin(phase);
			}

			t->k.vf_type = 3;
			t->k.vf_ntaps = ntaps;
			t->vf_itaps = _q15_applied(rot + 0, ntaps, 2);
			t->vf_qtaps = _q15_applied(rot + 1, ntaps, 2);
			if(!t->vf_itaps || !t->vf_qtaps) return(HVK_OUT_OF_MEMORY);
		}
		else
		{
			double lp[51];
			_design_low_pass(lp, ntaps, sample_rate, c->video_bw);
			t->k.vf_type = 1;
			t->k.vf_ntaps = ntaps;
			t->vf_itaps = _q15_applied(lp, ntaps, 1);
			if(!t->vf_itaps) return(HVK_OUT_OF_MEMORY);
		}

		/* whole lines of latency the reference's pipeline drops at start-up
		 * (src/video.c:3620-3625, :3759) */
		t->k.delay_lines = (ntaps / 2 + fw - 1) / fw;
	}

	/* --pixelrate (src/video.c:3627-3651, src/fir.c:393-428, :260-296): the raster is built at the
	 * pixel rate and a rational poly-phase FIR takes it to the sample rate. Output sample r of the
	 * resampled stream is made from input n = floor(r * D / L) and the ataps - 1 inputs before it
	 * with the taps of phase (r * D) mod L. */
	if(pixel_rate != sample_rate)
	{
		int64_t a = sample_rate, b = pixel_rate, g;
		int L, D, ntaps, total, j;
		double *taps, cutoff;
		int64_t w0, w1;

		while(b) { g = a % b; a = b; b = g; }
		L = sample_rate / a;
		D = pixel_rate / a;

		/* what the device kernel is sized for: a window of raster samples per tile (decimation of up to four times the
		 * interpolation), a tap table that fits a 32-bit index. Up to 256 phases the table lives in LDS; beyond that
		 * (27 MHz <-> 4 x f_sc: 709379 : 1080000, fifteen million taps, src/fir.c:404) a sample reads its phase's row from HBM
		 * (hvk_k_resample<true>) */
		if(D > 4 * (int64_t) L || L > 20000000) REFUSE("resampling %u -> %u Hz is %d : %d in lowest terms: the kernel takes a decimation of up to four times the interpolation and up to 20 000 000 phases", pixel_rate, sample_rate, L, D);
		/* frames of constant length, or (525 lines at 13.5 -> 16 MHz: 450450 * 32 / 27) of two lengths one sample apart */
		t->k.rs_irr = ((int64_t) t->k.raster_samples * L) % D != 0;

		ntaps = (21 * L) | 1;
		taps = calloc(ntaps, sizeof(double));
		if(!taps) return(HVK_OUT_OF_MEMORY);

		cutoff = L > D ? 0.45 : 0.45 * L / D;    /* up / down */
		_design_low_pass_gain(taps, ntaps, L, cutoff, L);

		t->k.rs_L = L;
		t->k.rs_D = D;
		t->k.rs_ataps = (ntaps + L - 1) / L;
		total = t->k.rs_ataps * L;
		if(L <= 256 && total > 8192) { free(taps); REFUSE("the resampler %u -> %u Hz has %d taps: the kernel's table holds 8192", pixel_rate, sample_rate, total); }
		t->rs_taps = calloc(total, sizeof(int16_t));
		if(!t->rs_taps) { free(taps); return(HVK_OUT_OF_MEMORY); }

		/* phase p's taps at [p * ataps, (p + 1) * ataps), oldest sample first (src/fir.c:277-284) */
		j = total - t->k.rs_ataps;
		for(i = ntaps - 1; i >= 0; i--)
		{
			t->rs_taps[j] = lround(taps[i] * 32767.0);
			j -= t->k.rs_ataps;
			if(j < 0) j += total + 1;
		}
		free(taps);

		/* the kernel's position arithmetic is 32-bit: (frame-local resampled index) * D */
		if(L <= 256 && ((uint64_t) t->k.raster_samples * L / D + 4 * (uint64_t) t->k.width * L / D + 4096) * D >= 0xFFFFFFFFull) { free(t->rs_taps); t->rs_taps = NULL; REFUSE("resampling %u -> %u Hz: a frame's positions times %d leave the kernel's 32-bit arithmetic", pixel_rate, sample_rate, D); }
		t->k.frame_samples = (int32_t) ((int64_t) t->k.raster_samples * L / D) + (t->k.rs_irr ? 1 : 0);
		t->k.slab_lines = t->k.lines + 3;
		t->max_width = (int32_t) (((int64_t) t->k.width * L + D - 1) / D);   /* fir_int16_output_size, src/fir.c:376-381 */

		/* Start-up: the resampler's output for raster line N lands in the slot
		 * of line N - 1 and it has no latency to make up for that, so its first chunk -- the resampled
		 * raster line 1, w0 samples -- never leaves the pipeline; with the filter on the next slot
		 * (w1 samples) is the filter's start-up line. The filter's own latency is `fw` samples. */
		w0 = ((int64_t) t->k.width * L + D - 1) / D;
		w1 = ((int64_t) 2 * t->k.width * L + D - 1) / D - w0;
		t->k.rs_shift = (int32_t) (w0 + (t->k.vf_type ? w1 - round((double) sample_rate * line_s) : 0));
		t->k.out_prime = (int32_t) (w0 + (t->k.vf_type ? w1 : 0));
		if(t->k.rs_shift < 64 + (t->k.rs_irr ? 2 : 0)) REFUSE("resampling %u -> %u Hz: lines of %d samples are shorter than the kernel's lead", pixel_rate, sample_rate, (int) w0);
	}
	else
	{
		t->k.out_prime = t->k.delay_lines * t->k.width;
	}

	/* the filter kernel's input slab: the raster itself (one halo line either side), or the
	 * resampled stream with 64 samples either side */
	if(t->k.rs_L)
	{
		t->k.s_lead = 64;
		t->k.s_stride = (t->k.frame_samples + 2 * 64 + 7) & ~7;
	}
	else
	{
		t->k.s_lead = t->k.width;
		t->k.s_stride = (t->k.lines + 2) * t->k.width;
	}

	if((r = _build_audio(t, slevel)) != HVK_OK) return(r);

	/* FM video: carrier at 0 Hz, deviation per unit of signal (src/video.c:4563-4585, :2218-2243) */
	if(c->modulation == HVK_FM)
	{
		t->fmv_level = (int16_t) round(INT16_MAX * (c->fm_level * c->level));
		t->fmv_lut = malloc(sizeof(hvk_c32_t) * 65536);
		if(!t->fmv_lut) return(HVK_OUT_OF_MEMORY);
		for(i = INT16_MIN; i <= INT16_MAX; i++)
		{
			t->fmv_lut[i - INT16_MIN] = _unit_phasor(2.0 * M_PI / t->sample_rate * (0 + (double) i / INT16_MAX * c->fm_deviation));
		}
		t->k.fm_video = 1;
		/* (with the resampler the modulator's start-up samples come from the RESAMPLED stream -- the pipeline's chunks
		 * before the first emitted line, src/video.c:4936-4952 with :3627-3651 --: hvk_engine_launch.cpp primes the phasor with
		 * them) */
	}

	/* raw baseband input (src/video.c:2406-2446, :4180-4191): no raster, no colour process, no
	 * sub-carrier table on the lines (VITS then adds luma only) */
	if(c->raw_bb)
	{
		if(c->raw_bb_white_level == c->raw_bb_blanking_level) return(HVK_ERROR);
		if(c->s_video) REFUSE("--s-video beside --raw-bb-file: raw baseband has no sub-carrier to put on a second channel");
		t->k.rawbb = 1;
		t->k.rawbb_blank = c->raw_bb_blanking_level;
		t->k.rawbb_range = c->raw_bb_white_level - c->raw_bb_blanking_level;
		t->k.white = t->white_level;
		t->k.colour = 0;
		t->k.secam = 0;
	}

	/* S-Video: baseband colour modes only (src/hacktv.c:1136-1148); behind the resampler the sub-carrier has a channel of its own (src/video.c:4361-4367) */
	if(c->s_video)
	{
		if(c->output_type != HVK_INT16_REAL || c->colour_mode == HVK_MONOCHROME) REFUSE("--s-video needs a baseband colour mode (src/hacktv.c:1136-1148)");
		/* With the video filter behind a resampler whose lines are not all of one width (525 lines at 16 MHz: 1017, 1017,
		 * ..., 1016) the reference pairs a line's luma -- as many samples as the chunk the filter was last fed, dst->width =
		 * fir_int16_process(), src/video.c:3243 -- with the sub-carrier its line buffer holds, which is the chunk of the line
		 * before's width: where that differs from the width of the last chunk dropped at start-up the sub-carrier stands a
		 * sample off in the line, and a line a sample longer than it ends on what the buffer held before -- the raster's
		 * sub-carrier of the line before it at that place when resampling downwards, the raster's blanking (it clears
		 * max_width samples, src/video.c:2934-2939) when upwards. hvk_k_svq makes the Q channel line by line that way (hvk_engine_launch.cpp has the
		 * per-line records); the oracle keeps the ring itself (oracle_video.c). */
		if(t->k.rs_L && t->k.vf_type && ((int64_t) t->k.width * t->k.rs_L) % t->k.rs_D != 0)
		{
			const int olines = _ring_lines(t);
			t->k.sv_ring = olines > t->k.delay_lines + 2 ? olines : t->k.delay_lines + 2;
			if(getenv("HVK_SV_EXPERIMENT")) t->k.sv_ring = 0;       /* (tools/sv_probe.py: the sub-carrier at the luma's own position, as before) */
			/* hvk_k_svq walks the batch's sub-carrier as ONE run of samples, the end of the batch before in front of it: the layout of
			 * frames of two lengths, also where only the LINES have two widths and a frame is a whole number of samples (4 x the PAL
			 * sub-carrier from 18 MHz pixels: 1135.0064 samples a line, 709379 a frame -- found by tools/fuzz_parity.py, round 6: the
			 * records were made for that layout, the frames lay s_stride apart, and the per-frame record the lines' places are
			 * worked out from was not there). frame_samples stays the exact length. */
			if(t->k.sv_ring) t->k.rs_irr = 1;
		}
		t->k.s_video = 1;
	}

	/* complex tail (src/video.c:4587-4645) */
	t->k.swap_iq = c->swap_iq != 0;
	t->k.has_offset = c->offset != 0;
	t->k.has_passthru = c->passthru != 0;
	if(c->offset != 0) t->offset_delta = _unit_phasor(2.0 * M_PI / t->sample_rate * c->offset);

	t->k.black = t->black_level;

	if(c->teletext)
	{
		/* 625-line systems only (src/hacktv.c:1182-1186) */
		if(c->lines != 625) REFUSE("teletext needs a 625-line mode (src/hacktv.c:1182-1186)");
		if((r = _build_teletext(t)) != HVK_OK) return(r);
		if((r = _append_teletext_lut(t)) != HVK_OK) return(r);
	}

	if(c->wss && (r = _build_wss(t)) != HVK_OK) return(r);
	if(c->vitc && (r = _build_vitc(t)) != HVK_OK) return(r);
	if(c->cc608 && (r = _build_cc608(t)) != HVK_OK) return(r);
	if(c->vits && (r = _build_vits(t)) != HVK_OK) return(r);
	if(c->acp) _build_acp(t);
	t->k.vbi = t->vbi_nsym > 0 || c->acp;
	if(c->sis)
	{
		/* (the burst is laid out in pixels of the raster and drawn there -- in front of the resampler, beside --s-video's second
		 * channel, over a line that came from --raw-bb-file all the same; the hand-over of the sound blocks goes by the
		 * pipeline's steps, whatever the width of the audio process's lines) */
		if(c->sis != 1) REFUSE("sound-in-syncs mode %d (the reference has dcsis)", c->sis);
		if((r = _build_sis(t)) != HVK_OK) return(r);
	}

	return(_build_linedesc(t));
}

void hvk_tables_free(hvk_tables_t *t)
{
	free(t->sis_dense);
	free(t->sis_win);
	free(t->sis_first);
	free(t->desc);
	free(t->linebase);
	free(t->pulse_values);
	free(t->sync_packed);
	free(t->colour_lookup);
	free(t->burst_win);
	free(t->chroma_taps);
	free(t->vf_itaps);
	free(t->vf_qtaps);
	free(t->fm_lut);
	free(t->a2_lut);
	free(t->nicam_taps);
	free(t->nicam_cc);
	free(t->tt_symbols);
	free(t->tt_values);
	free(t->secam_lut);
	free(t->secam_bell);
	free(t->secam_fir);
	free(t->secam_notch);
	free(t->fmv_lut);
	free(t->rs_taps);
	free(t->vbi_sym);
	free(t->vbi_val);
	free(t->fsc_rows);
	free(t->vits_l);
	free(t->vits_c);
	memset(t, 0, sizeof(*t));
}

int64_t hvk_tables_frame_start(const hvk_tables_t *t, int64_t frame)
{
	const hvk_kconst_t *k = &t->k;
	if(!k->rs_irr) return(frame * (int64_t) k->frame_samples);
	/* where the frame's first EMITTED line begins (hvk_tables_line_widths(): emitted line j is the resampler's chunk
	 * j + s, chunk g begins at ceil(g W L / D)) */
	{
		const int64_t s = 1 + (k->vf_type ? k->delay_lines : 0);
		const int64_t g = frame * k->lines + s;
		return((g * k->width * k->rs_L + k->rs_D - 1) / k->rs_D - (s * k->width * k->rs_L + k->rs_D - 1) / k->rs_D);
	}
}

void hvk_tables_line_widths(const hvk_tables_t *t, int64_t first, int n, int32_t *widths)
{
	const hvk_kconst_t *k = &t->k;
	int i;

	for(i = 0; i < n; i++)
	{
		if(k->rs_L == 0) widths[i] = k->width;
		else
		{
			/* emitted line j is resampler chunk j + (chunks dropped at start-up); chunk g holds the
			 * outputs made from raster line g: [ceil(g W L / D), ceil((g + 1) W L / D)) */
			const int64_t g = first + i + 1 + (k->vf_type ? k->delay_lines : 0);
			const int64_t lo = (g * k->width * k->rs_L + k->rs_D - 1) / k->rs_D;
			const int64_t hi = ((g + 1) * k->width * k->rs_L + k->rs_D - 1) / k->rs_D;
			widths[i] = (int32_t) (hi - lo);
		}
	}
}

static long _give(void *dst, long max_bytes, const void *src, long bytes)
{
	if(src == NULL || bytes <= 0) return(0);
	if(dst == NULL) return(bytes);
	if(bytes > max_bytes) bytes = max_bytes;
	memcpy(dst, src, bytes);
	return(bytes);
}

/* Table access by the names the oracle and the reference probe use */
long hvk_tables_get(const hvk_tables_t *t, const char *name, void *dst, long max_bytes)
{
	if(!strcmp(name, "syncs"))         return(_give(dst, max_bytes, t->sync_packed, (long) t->sync_packed_len * 2));
	if(!strcmp(name, "colour_lookup")) return(_give(dst, max_bytes, t->colour_lookup, (long) t->colour_lookup_len * 4));
	if(!strcmp(name, "burst_win"))     return(_give(dst, max_bytes, t->burst_win, (long) t->k.burst_width * 2));
	if(!strcmp(name, "chroma_taps"))   return(_give(dst, max_bytes, t->chroma_taps, (long) t->k.chroma_ntaps * 2));
	if(!strcmp(name, "chroma_ghost"))  return(_give(dst, max_bytes, t->ghost, sizeof(t->ghost)));
	if(!strcmp(name, "vfilter_itaps")) return(_give(dst, max_bytes, t->vf_itaps, (long) t->k.vf_ntaps * 2));
	if(!strcmp(name, "vfilter_qtaps")) return(_give(dst, max_bytes, t->vf_qtaps, t->vf_qtaps ? (long) t->k.vf_ntaps * 2 : 0));
	if(!strcmp(name, "fm_mono_lut"))   return(_give(dst, max_bytes, t->fm_lut, 65536L * 8));
	if(!strcmp(name, "fm_video_lut"))  return(_give(dst, max_bytes, t->fmv_lut, t->fmv_lut ? 65536L * 8 : 0));
	if(!strcmp(name, "resampler_taps")) return(_give(dst, max_bytes, t->rs_taps, t->rs_taps ? (long) t->k.rs_L * t->k.rs_ataps * 2 : 0));
	if(!strcmp(name, "nicam_taps"))    return(_give(dst, max_bytes, t->nicam_taps, (long) t->k.nicam_ntaps * 2));
	if(!strcmp(name, "nicam_cc"))      return(_give(dst, max_bytes, t->nicam_cc, (long) t->k.nicam_cc_len * 4));
	if(!strcmp(name, "limiter_shape")) return(_give(dst, max_bytes, t->limiter_shape, t->has_limiter ? 21L * 2 : 0));
	if(!strcmp(name, "limiter_vtaps")) return(_give(dst, max_bytes, t->limiter_vtaps, t->has_limiter ? 65L * 4 : 0));
	if(!strcmp(name, "limiter_ftaps")) return(_give(dst, max_bytes, t->limiter_ftaps, t->has_limiter ? 65L * 4 : 0));
	if(!strcmp(name, "fm_secam_lut"))  return(_give(dst, max_bytes, t->secam_lut, t->secam_lut ? 65536L * 8 : 0));
	if(!strcmp(name, "fm_secam_bell")) return(_give(dst, max_bytes, t->secam_bell, t->secam_bell ? 65535L * 4 : 0));
	if(!strcmp(name, "fm_secam_fir"))  return(_give(dst, max_bytes, t->secam_fir, t->secam_fir ? 15L * 2 : 0));
	if(!strcmp(name, "secam_l_fir"))   return(_give(dst, max_bytes, t->secam_notch, t->secam_notch ? 51L * 2 : 0));
	if(!strcmp(name, "teletext_lut"))
	{
		/* the reference's packed layout [length][offset][values...]...[-1] */
		long n = 1, o = 0;
		int b;
		int16_t *q;
		if(!t->tt_symbols) return(0);
		for(b = 0; b < 360; b++) n += 2 + t->tt_symbols[b * 3 + 1];
		if(dst == NULL) return(n * 2);
		q = malloc(n * 2);
		if(!q) return(-1);
		for(b = 0; b < 360; b++)
		{
			q[o++] = t->tt_symbols[b * 3 + 1];
			q[o++] = t->tt_symbols[b * 3 + 0];
			memcpy(q + o, t->tt_values + t->tt_symbols[b * 3 + 2], t->tt_symbols[b * 3 + 1] * 2);
			o += t->tt_symbols[b * 3 + 1];
		}
		q[o++] = -1;
		n = _give(dst, max_bytes, q, n * 2);
		free(q);
		return(n);
	}
	if(!strcmp(name, "linedesc"))      return(_give(dst, max_bytes, t->desc, (long) 2 * t->k.lines * sizeof(hvk_linedesc_t)));
	return(-1);
}
