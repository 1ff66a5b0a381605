/* hvk_secam.c -- the SECAM colour sub-carrier on the host: the serial chain in stream order.
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
 * The line's arithmetic is hvk_secam_chain.h, shared with the device kernels (hvk_secam.hip); what one line hands
 * to the next is the small hvk_secam_state_t. This file walks the lines in order with the true state -- the
 * reference's own order of work, the pin for the shared code (tests/test_oracle_*.py against the reference's
 * lines) and the engine's fall-back where the device's speculation fails -- and lists, per frame, the lines the
 * process touches at all ("tasks": lines with a picture or a field identification ramp).
 *
 * HVK_SECAM_SPEC=K (a test switch): the frame is computed the way the device does it -- every task on its own,
 * its entry state taken from K warm-up lines that start from nothing, the entry states then checked against
 * the exit states of the tasks before -- and the failures counted (hvk_secam_counters()).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hvk_internal.h"
#include "hvk_secam_chain.h"

#define FM_DEV   1000e3
#define FM_FREQ  4328125
#define CB_FREQ  4250000
#define CR_FREQ  4406250

struct hvk_secam {
	const hvk_tables_t *t;
	int W, lines, hline;
	int16_t *uv;            /* 2^24 x {u, v}: the chroma half of the reference's level table */
	hvk_secam_consts_t C;
	hvk_secam_task_t *tasks[2];     /* per frame parity */
	int ntasks[2];
	int16_t *fid_row[2];    /* the cells of a field identification line, D'b / D'r */
	hvk_secam_state_t state;        /* after the last line walked */
	int64_t next_frame;     /* frames must come in order */
	/* scratch: one frame's tasks */
	int16_t *cells;         /* [W + 16] */
	int16_t *F;             /* [max tasks + 2][W] */
	int32_t *acc;           /* [max tasks + 2][8] */
	int max_tasks;
	int spec_k;             /* HVK_SECAM_SPEC */
	hvk_secam_state_t *entry, *exit;
	int64_t n_tasks, n_mismatch, n_repaired;
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

/* The lines of a frame of this parity the process works on, in order, with what each needs to know of the lines
 * before it (hvk_secam_task_t, hvk_internal.h). Shared with the engine, which uploads the lists. */
int hvk_secam_tasks(const hvk_tables_t *t, int parity, hvk_secam_task_t *out, int max)
{
	const hvk_kconst_t *k = &t->k;
	const int sl = k->burst_left;
	int line, n = 0, pending_clear = 0, prev = 0;

	for(line = 1; line <= k->lines; line++)
	{
		const hvk_linedesc_t *d = &t->desc[parity * k->lines + line - 1];
		const int picture = d->ar > d->al;
		const int right_half = picture && d->ar > k->half_width;
		const int fid = d->secam_fid & 1;
		const int sr = (right_half || fid) ? sl + k->burst_width : k->half_width;

		/* top of a field: the reference clears its buffer, the line before's component and what lies behind the
		 * line with it (src/video.c:3095-3099) */
		if(line == 1 || line == t->conf.hline) { pending_clear = 1; prev = 0; }

		if((!picture && !fid) || sr <= sl) continue;

		if(out)
		{
			if(n >= max) return(-1);
			memset(&out[n], 0, sizeof(out[n]));
			out[n].line = (int16_t) line;
			out[n].prev_line = fid ? 0 : (int16_t) prev;
			out[n].sr = (int16_t) sr;
			out[n].flags = (fid ? HVK_SECAM_TASK_FID : 0) | (pending_clear ? HVK_SECAM_TASK_CLEAR : 0) | HVK_SECAM_TASK_VALID;
		}
		n++;
		pending_clear = 0;
		if(!fid) prev = line;     /* a field identification line leaves the kept component alone */
	}

	return(n);
}

hvk_secam_t *hvk_secam_new(const hvk_tables_t *t)
{
	hvk_secam_t *s = calloc(1, sizeof(hvk_secam_t));
	int p, i;

	if(!s) return(NULL);

	s->t = t;
	s->W = t->k.width;
	s->lines = t->k.lines;
	s->hline = t->conf.hline;
	s->uv = _build_uv(t);

	s->C.W = s->W;
	s->C.sl = t->k.burst_left;
	s->C.level = t->secam_level;
	for(i = 0; i < 2; i++) { s->C.dmin[i] = t->secam_dmin[i]; s->C.dmax[i] = t->secam_dmax[i]; }
	for(i = 0; i < 15; i++) s->C.fir[i] = t->secam_fir[i];

	for(p = 0; p < 2; p++)
	{
		s->ntasks[p] = hvk_secam_tasks(t, p, NULL, 0);
		s->tasks[p] = calloc(s->ntasks[p] + 1, sizeof(hvk_secam_task_t));
		if(!s->tasks[p] || hvk_secam_tasks(t, p, s->tasks[p], s->ntasks[p]) != s->ntasks[p]) { hvk_secam_free(s); return(NULL); }
		if(s->ntasks[p] > s->max_tasks) s->max_tasks = s->ntasks[p];

		/* field identification line (src/video.c:3101-3133): the sub-carrier ramps from the line's rest frequency
		 * by 350 kHz over 15 us (D'r) / 18 us (D'b) */
		s->fid_row[p] = calloc(s->W, sizeof(int16_t));
		if(!s->fid_row[p] || !s->uv) { hvk_secam_free(s); return(NULL); }
		hvk_secam_fid_row(t, p, s->uv[p ? 1 : 0], s->fid_row[p]);
	}

	s->cells = calloc(s->W + 16, sizeof(int16_t));
	s->F = calloc((size_t) (s->max_tasks + 2) * s->W, sizeof(int16_t));
	s->acc = calloc((size_t) (s->max_tasks + 2) * 8, sizeof(int32_t));
	s->entry = calloc(s->max_tasks + 2, sizeof(hvk_secam_state_t));
	s->exit = calloc(s->max_tasks + 2, sizeof(hvk_secam_state_t));
	if(getenv("HVK_SECAM_SPEC")) s->spec_k = atoi(getenv("HVK_SECAM_SPEC"));

	if(!s->cells || !s->F || !s->acc || !s->entry || !s->exit)
	{
		hvk_secam_free(s);
		return(NULL);
	}

	return(s);
}

/* cells of a field identification line of a D'b (dr = 0) / D'r line; `level`: the line's rest value */
void hvk_secam_fid_row(const hvk_tables_t *t, int dr, int16_t level, int16_t *row)
{
	const int16_t dev = dr ? t->secam_fsync_level : -t->secam_fsync_level;
	const double rw = dr ? 15e-6 : 18e-6;
	int x;

	for(x = 0; x < t->k.width; x++)
	{
		double tt = (double) (x - t->k.active_left) / t->pixel_rate / rw;
		if(tt < 0) tt = 0;
		else if(tt > 1) tt = 1;
		row[x] = level + dev * tt;
	}
}

void hvk_secam_free(hvk_secam_t *s)
{
	if(!s) return;
	free(s->uv);
	free(s->tasks[0]); free(s->tasks[1]);
	free(s->fid_row[0]); free(s->fid_row[1]);
	free(s->cells);
	free(s->F);
	free(s->acc);
	free(s->entry);
	free(s->exit);
	free(s);
}

void hvk_secam_counters(const hvk_secam_t *s, int64_t *tasks, int64_t *mismatches, int64_t *repaired)
{
	if(tasks) *tasks = s->n_tasks;
	if(mismatches) *mismatches = s->n_mismatch;
	if(repaired) *repaired = s->n_repaired;
}

void hvk_secam_get_state(const hvk_secam_t *s, hvk_secam_state_t *st, int64_t *next_frame) { *st = s->state; if(next_frame) *next_frame = s->next_frame; }
void hvk_secam_set_state(hvk_secam_t *s, const hvk_secam_state_t *st, int64_t next_frame) { s->state = *st; s->next_frame = next_frame; }

/* The sample-parallel part of a line: cells (the colour difference of this line averaged with the line before's,
 * src/video.c:3149-3196) and the 15-tap low pass without the share of what lies behind the line.
 * row / prow: source pixels shown on this line / on the line before in the field (NULL: none, shows as RGB 0);
 * have_prev: a line of the field has left its other component behind; comp / pcomp: index into {u, v} of this
 * line's component and of the one the line before kept. */
static void _cells_fir(hvk_secam_t *s, const int16_t *fid_row, int comp, int have_prev, int pcomp,
                       const uint32_t *row, const uint32_t *prow, int row_width, int vframe_x, int16_t *F, int32_t *acc)
{
	const int W = s->W;
	int16_t *c = s->cells + 7;      /* 7 zeros in front (the filter's zero history), 7 behind (the tail's places) */
	int x, k;

	if(fid_row) memcpy(c, fid_row, sizeof(int16_t) * W);
	else
	{
		const int16_t rest = s->uv[comp];                   /* of RGB 000000 */
		const int p0 = s->t->k.active_left + vframe_x;

		for(x = 0; x < p0; x++) c[x] = rest;
		for(; x < p0 + row_width; x++)
		{
			const uint32_t rgb = row ? (row[x - p0] & 0xFFFFFF) : 0;
			const uint32_t prgb = prow ? (prow[x - p0] & 0xFFFFFF) : 0;
			const int16_t held = have_prev ? s->uv[prgb * 2 + pcomp] : 0;
			c[x] = (s->uv[rgb * 2 + comp] + held) / 2;
		}
		for(; x < W; x++) c[x] = rest;
	}

	for(x = 0; x < W; x++)
	{
		int32_t a = 0;
		for(k = 0; k < 15; k++) a += (int32_t) c[x - 7 + k] * s->C.fir[k];
		if(x >= W - HVK_SECAM_TAIL) acc[x - (W - HVK_SECAM_TAIL)] = a;
		else
		{
			a >>= 15;
			F[x] = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
		}
	}
}

static int _same_state(const hvk_secam_state_t *a, const hvk_secam_state_t *b)
{
	return(memcmp(&a->ix, &b->ix, sizeof(double)) == 0 && memcmp(&a->iy, &b->iy, sizeof(double)) == 0 &&
	       memcmp(a->tail, b->tail, sizeof(int16_t) * HVK_SECAM_TAIL) == 0);
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
	const int parity = frame & 1;
	const int W = s->W;
	const hvk_secam_task_t *T = s->tasks[parity];
	const int nprime = frame_index == 0 ? 2 : 0;
	const int n = s->ntasks[parity] + nprime;
	int i;

	if(frame_index != s->next_frame) return(HVK_ERROR);

	memset(out, 0, sizeof(int16_t) * (size_t) k->lines * W);

	/* ---- the sample-parallel part of every task ---- */
	for(i = 0; i < n; i++)
	{
		if(i < nprime)
		{
			/* The line pipeline hands the process two never-emitted slots (frame 1, line 0) before the first real
			 * line; it treats them as picture lines without a picture and they advance the IIR (src/video.c:4676-4688
			 * with :4665-4667). Both are D'r or D'b alike: the second finds the OTHER component
			 * of the first kept */
			const int dr = (frame * s->lines) & 1;
			/* (the second with the place and width of the picture in force, the stream's first, and no row of it: none for
			 * an empty frame; the FIRST is taken by the colour process's thread before the source has been read at all,
			 * while vid_init()'s frame is in force -- the full active width, no pixels, src/video.c:4169-4177: with a first
			 * picture narrower than the raster the filter state behind the two slots differs, and at 13.5 / 14 MHz the first
			 * field identification line shows it; the reference on it: tests/ref_random_check.py secam_sv_narrow) */
			const int fw = i == 0 ? k->active_width : (fb1 ? fb1_width : 0);
			_cells_fir(s, NULL, dr ? 1 : 0, i == 1, dr ? 0 : 1, NULL, NULL, fw, (k->active_width - fw) / 2, s->F + (size_t) i * W, s->acc + (size_t) i * 8);
			continue;
		}
		{
			const hvk_secam_task_t *q = &T[i - nprime];
			const int line = q->line;
			const int dr = ((frame * s->lines) + line) & 1;
			const int pdr = ((frame * s->lines) + q->prev_line) & 1;
			/* the frame the line's field shows */
			const int second = k->fields == 2 && line >= k->hline;
			const uint32_t *fb = second ? fb2 : fb1;
			const int fb_width = second ? fb2_width : fb1_width, fb_height = second ? fb2_height : fb1_height;
			const int fb_interlaced = second ? fb2_interlaced : fb1_interlaced;
			const int vframe_x = (k->active_width - fb_width) / 2;
			const int vframe_y = (k->active_lines - fb_height) / 2;
			const uint32_t *row = NULL, *prow = NULL;
			int vy = t->desc[parity * k->lines + line - 1].src_row;

			if(vy >= 0 && k->interlaced != 0 && fb_interlaced != k->interlaced) vy += 1;
			vy -= vframe_y;
			/* an empty frame (0 x 0, what a source past its end hands out) shows no pixels at all */
			if(fb && vy >= 0 && vy < fb_height) row = fb + (size_t) vy * fb_width;
			if(q->prev_line)
			{
				int py = t->desc[parity * k->lines + q->prev_line - 1].src_row;
				if(py >= 0 && k->interlaced != 0 && fb_interlaced != k->interlaced) py += 1;
				py -= vframe_y;
				if(fb && py >= 0 && py < fb_height) prow = fb + (size_t) py * fb_width;
			}

			_cells_fir(s, (q->flags & HVK_SECAM_TASK_FID) ? s->fid_row[dr] : NULL, dr ? 1 : 0, q->prev_line != 0, pdr ? 0 : 1,
			           row, prow, fb_width, vframe_x, s->F + (size_t) i * W, s->acc + (size_t) i * 8);
		}
	}

	/* ---- the serial part ---- */
#define TASK_ARGS(i) \
	const int line_ = (i) < nprime ? 0 : T[(i) - nprime].line; \
	const int dr_ = ((frame * s->lines) + line_) & 1; \
	const int sr_ = (i) < nprime ? k->burst_left + k->burst_width : T[(i) - nprime].sr; \
	const int pos_ = ((frame * s->lines) + line_) % 3 == 0; \
	const int clear_ = (i) >= nprime && (T[(i) - nprime].flags & HVK_SECAM_TASK_CLEAR)
#define TASK_RUN(i, st, o) \
	do { \
		TASK_ARGS(i); \
		if(clear_) memset((st)->tail, 0, sizeof((st)->tail)); \
		hvk_secam_chain_line(&s->C, (const hvk_secam_c32_t *) t->secam_lut, (const hvk_secam_c16_t *) t->secam_bell, t->burst_win, \
		                     (st), s->F + (size_t) (i) * W, 1, s->acc + (size_t) (i) * 8, 1, dr_, sr_, pos_, (o), 1); \
	} while(0)
#define TASK_OUT(i) ((i) < nprime ? NULL : out + (size_t) (T[(i) - nprime].line - 1) * W)

	if(s->spec_k <= 0)
	{
		hvk_secam_state_t st = s->state;
		for(i = 0; i < n; i++) TASK_RUN(i, &st, TASK_OUT(i));
		s->state = st;
	}
	else
	{
		/* the device's way: every task by itself, K warm-up lines from nothing (or from the frame's true entry state
		 * when they reach back that far), then the check */
		const int K = s->spec_k;

		for(i = 0; i < n; i++)
		{
			int j0 = i < K ? i : K, m;
			hvk_secam_state_t st;
			if(i - j0 == 0) st = s->state;
			else memset(&st, 0, sizeof(st));
			for(m = i - j0; m < i; m++) TASK_RUN(m, &st, NULL);
			s->entry[i] = st;
			TASK_RUN(i, &st, TASK_OUT(i));
			s->exit[i] = st;
		}
		s->n_tasks += n;
		{
			/* the check: walk the chain of (entry, exit) pairs. A task whose derived entry state is not the state the
			 * tasks before it really left is redone from that state -- and so are the tasks after it for as long as
			 * their own derived entry states differ from what the redone task before them leaves */
			hvk_secam_state_t st = s->state;
			for(i = 0; i < n; i++)
			{
				if(_same_state(&s->entry[i], &st)) { st = s->exit[i]; continue; }
				s->n_mismatch++;
				s->n_repaired++;
				TASK_RUN(i, &st, TASK_OUT(i));
			}
			s->exit[n ? n - 1 : 0] = st;
		}
		if(n > 0) s->state = s->exit[n - 1];
	}

	s->next_frame++;
	return(HVK_OK);
}
