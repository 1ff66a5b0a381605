/* hvk_tail.c -- the serial part of the output tail of the line pipeline.
 *
 * After the audio process the reference runs up to four more processes on the
 * finished I/Q line, in this order (src/video.c:4563-4645):
 *
 *   fmmod     FM video: the I sample steers a Q31 phasor, which replaces the
 *             sample (src/video.c:3452-3464, :2299-2335)
 *   swap_iq   exchanges I and Q (src/video.c:3466-3480)
 *   offset    multiplies by a free-running Q31 phasor (src/video.c:3482-3515)
 *   passthru  adds an external int16 I/Q stream, whole lines at a time
 *             (src/video.c:3517-3541)
 *
 * Both phasors are the floor-after-every-step recurrence of src/common.h:80-89
 * with an atan2/cos/sin re-normalisation every 32767 samples: sample n needs
 * sample n - 1 bit for bit, so they run here, once, in stream
 * order on the host.
 *
 *  - The offset phasor does not depend on the signal: it is generated as a
 *    side stream (int16 pairs, phase >> 16) and the multiply is device work
 *    (hvk_k_tail), as are the swap and the passthru add.
 *  - The FM video phasor depends on every sample of the composite, which the
 *    device renders: hvk_tail_fm_apply() runs the whole tail over the fetched
 *    baseband on the host. It is the engine's only sample-rate host pass.
 *
 * Pipeline fill: with the video filter on, the line pipeline hands these
 * processes delay_lines never-emitted lines of full width before the first real
 * one (src/video.c:3235-3248 sets their width); the offset phasor advances over
 * them and the passthru source loses its first delay_lines * width samples.
 * The FM video phasor runs over them too -- over the video filter's output while its history fills, plus
 * the sound carriers of those samples: hvk_tail_fm_prime().
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hvk_internal.h"

struct hvk_tail {
	const hvk_tables_t *t;
	int W;
	int64_t prime;              /* samples of the start-up lines (delay_lines * width; see hvk_tables.c for --pixelrate) */

	/* offset phasor; steps counts the multiplications done so far */
	hvk_c32_t off_phase, off_delta;
	int32_t off_counter;
	int64_t off_steps;

	/* FM video phasor */
	hvk_c32_t fm_phase;
	int32_t fm_counter;
	int64_t fm_pos;             /* next output position to modulate */

	/* passthru queue: q[0] is source sample q_base */
	int16_t *q;
	size_t q_len, q_cap;
	size_t q_head;              /* pairs at the front that have been consumed already (dropped lazily) */
	int64_t q_base;
	int ended;                  /* a line found the source short: nothing is added from then on */
};

hvk_tail_t *hvk_tail_new(const hvk_tables_t *t)
{
	hvk_tail_t *s = calloc(1, sizeof(hvk_tail_t));
	if(!s) return(NULL);

	s->t = t;
	s->W = t->k.width;
	s->prime = t->k.out_prime;

	/* src/video.c:4596-4602: the offset phasor starts at INT16_MAX (sic), so
	 * its >> 16 is zero until the first re-normalisation */
	s->off_phase.i = INT16_MAX;
	s->off_phase.q = 0;
	s->off_counter = INT16_MAX;
	s->off_delta = t->offset_delta;

	/* src/video.c:2225-2227 */
	s->fm_phase.i = INT32_MAX;
	s->fm_phase.q = 0;
	s->fm_counter = INT16_MAX;

	return(s);
}

void hvk_tail_free(hvk_tail_t *s)
{
	if(!s) return;
	free(s->q);
	free(s);
}

/* src/common.h:80-89 */
static inline hvk_c32_t _mul(hvk_c32_t a, hvk_c32_t b)
{
	hvk_c32_t r;
	r.i = (int32_t) (((int64_t) a.i * b.i - (int64_t) a.q * b.q) >> 31);
	r.q = (int32_t) (((int64_t) a.i * b.q + (int64_t) a.q * b.i) >> 31);
	return(r);
}

static inline hvk_c32_t _renormalise(hvk_c32_t p)
{
	const double ra = atan2(p.q, p.i);
	hvk_c32_t r;
	r.i = lround(cos(ra) * INT32_MAX);
	r.q = lround(sin(ra) * INT32_MAX);
	return(r);
}

/* one step of the offset process: the value the sample is multiplied by */
static inline hvk_c16_t _offset_step(hvk_tail_t *s)
{
	hvk_c16_t b;

	s->off_phase = _mul(s->off_phase, s->off_delta);
	b.i = s->off_phase.i >> 16;
	b.q = s->off_phase.q >> 16;

	if(--s->off_counter == 0)
	{
		s->off_phase = _renormalise(s->off_phase);
		s->off_counter = INT16_MAX;
	}

	s->off_steps++;
	return(b);
}

int hvk_tail_offset_stream(hvk_tail_t *s, int64_t first, int64_t count, int16_t *out)
{
	const int64_t want = s->prime + first;
	int64_t n;

	if(first < 0 || count < 0 || want < s->off_steps) return(HVK_ERROR);

	while(s->off_steps < want) (void) _offset_step(s);

	for(n = 0; n < count; n++)
	{
		const hvk_c16_t b = _offset_step(s);
		out[n * 2 + 0] = b.i;
		out[n * 2 + 1] = b.q;
	}

	return(HVK_OK);
}

int hvk_tail_passthru_push(hvk_tail_t *s, const int16_t *iq, size_t nsamples)
{
	if(nsamples == 0) return(HVK_OK);
	if(!iq) return(HVK_ERROR);

	if(s->q_len + nsamples > s->q_cap)
	{
		size_t cap = s->q_cap ? s->q_cap : 1 << 16;
		int16_t *q;
		while(cap < s->q_len + nsamples) cap *= 2;
		q = realloc(s->q, cap * 2 * sizeof(int16_t));
		if(!q) return(HVK_OUT_OF_MEMORY);
		s->q = q;
		s->q_cap = cap;
	}

	memcpy(s->q + s->q_len * 2, iq, nsamples * 2 * sizeof(int16_t));
	s->q_len += nsamples;
	return(HVK_OK);
}

/* Drop queued source samples below source index `upto` */
static void _passthru_discard(hvk_tail_t *s, int64_t upto)
{
	int64_t drop = upto - s->q_base - (int64_t) s->q_head;
	if(drop <= 0) return;
	if((size_t) drop > s->q_len - s->q_head) drop = s->q_len - s->q_head;
	/* a read offset moves on; the queue is compacted once more than half of it has been consumed (FM video asks line
	 * by line: moving the rest of the queue for every line would be quadratic) */
	s->q_head += (size_t) drop;
	if(s->q_head * 2 > s->q_len)
	{
		memmove(s->q, s->q + s->q_head * 2, (s->q_len - s->q_head) * 2 * sizeof(int16_t));
		s->q_len -= s->q_head;
		s->q_base += (int64_t) s->q_head;
		s->q_head = 0;
	}
}

/* one line of `w` samples at output position p (src/video.c:3522-3533): all of its source samples [p + prime,
 * p + prime + w) or -- once the source is short -- none, now and ever after */
static int _passthru_line(hvk_tail_t *s, int64_t p, int w, int16_t *dst)
{
	const int64_t src = p + s->prime;

	if(src < s->q_base + (int64_t) s->q_head) return(HVK_ERROR);   /* forward only: what has been consumed is gone, dropped from the queue or not */
	if(src + w > s->q_base + (int64_t) s->q_len)
	{
		s->ended = 1;
		return(HVK_OK);
	}
	memcpy(dst, s->q + (src - s->q_base) * 2, (size_t) w * 2 * sizeof(int16_t));
	return(HVK_OK);
}

int hvk_tail_passthru_stream(hvk_tail_t *s, int64_t first, int64_t count, int16_t *out)
{
	const int W = s->W;
	int64_t p;
	int r;

	if(first < 0 || count < 0) return(HVK_ERROR);
	memset(out, 0, count * 2 * sizeof(int16_t));

	if(s->t->k.rs_L)
	{
		/* behind the resampler the lines the process sees vary in width (hvk_tables_line_widths()): whole frames only,
		 * whose lines are walked with their own widths -- the source is read on without gaps either way, the widths
		 * only decide where a short source stops */
		const int64_t FS = s->t->k.frame_samples;
		const int lines = s->t->k.lines;
		int32_t *w;
		/* which frame begins at `first`: frame_samples apart -- or, where a raster frame does not resample to a whole number
		 * of samples (frames of two lengths, FS the longer), where hvk_tables_frame_start() says */
		int64_t f = first / FS;
		while(hvk_tables_frame_start(s->t, f) < first) f++;
		if(hvk_tables_frame_start(s->t, f) != first) return(HVK_ERROR);
		{
			int64_t f2 = f;
			while(hvk_tables_frame_start(s->t, f2) < first + count) f2++;
			if(hvk_tables_frame_start(s->t, f2) != first + count) return(HVK_ERROR);
		}
		w = malloc(sizeof(int32_t) * lines);
		if(!w) return(HVK_OUT_OF_MEMORY);
		for(; hvk_tables_frame_start(s->t, f) < first + count && !s->ended; f++)
		{
			int64_t at = hvk_tables_frame_start(s->t, f);
			hvk_tables_line_widths(s->t, f * lines, lines, w);
			for(int l = 0; l < lines && !s->ended; l++)
			{
				if((r = _passthru_line(s, at, w[l], out + (at - first) * 2)) != HVK_OK) { free(w); return(r); }
				at += w[l];
			}
		}
		free(w);
		_passthru_discard(s, first + count + s->prime);
		return(HVK_OK);
	}

	if(first % W || count % W) return(HVK_ERROR);

	for(p = first; p < first + count && !s->ended; p += W)
	{
		if((r = _passthru_line(s, p, W, out + (p - first) * 2)) != HVK_OK) return(r);
	}

	_passthru_discard(s, first + count + s->prime);
	return(HVK_OK);
}

int64_t hvk_tail_fm_position(const hvk_tail_t *s)
{
	return(s->fm_pos);
}

/* The never-emitted start-up samples of the line pipeline (src/video.c:3235-3248, :4936-4952: slots whose line
 * number is 0 are dropped at the output, after every process has run over them): the FM phasor advances over
 * their modulator input -- count values, before the first call of hvk_tail_fm_apply(). */
int hvk_tail_fm_prime(hvk_tail_t *s, const int16_t *input, int64_t count)
{
	const hvk_c32_t *lut = s->t->fmv_lut;
	int64_t n;

	if(!lut || s->fm_pos != 0 || count < 0) return(HVK_ERROR);
	for(n = 0; n < count; n++)
	{
		s->fm_phase = _mul(s->fm_phase, lut[input[n] - INT16_MIN]);
		if(--s->fm_counter == 0)
		{
			s->fm_phase = _renormalise(s->fm_phase);
			s->fm_counter = INT16_MAX;
		}
	}
	return(HVK_OK);
}

int hvk_tail_fm_apply(hvk_tail_t *s, int64_t first, int64_t count, int16_t *iq)
{
	const hvk_tables_t *t = s->t;
	const hvk_c32_t *lut = t->fmv_lut;
	const int32_t level = t->fmv_level;
	const int swap = t->k.swap_iq, offset = t->k.has_offset, pass = t->k.has_passthru;
	int16_t *line = NULL;
	int64_t n;

	if(!lut || first != s->fm_pos || count < 0) return(HVK_ERROR);
	/* --passthru is added line by line as the lines end; behind the resampler, where the lines' widths vary, frame by
	 * frame behind the loop (hvk_tail_passthru_stream() walks whole frames there; the sum does not feed the modulator) */
	const int pass_frames = pass && t->k.rs_L;
	if(pass && !pass_frames && (first % s->W || count % s->W)) return(HVK_ERROR);
	if(pass_frames)
	{
		int64_t f = first / t->k.frame_samples, f2;
		while(hvk_tables_frame_start(t, f) < first) f++;
		for(f2 = f; hvk_tables_frame_start(t, f2) < first + count; f2++);
		if(hvk_tables_frame_start(t, f) != first || hvk_tables_frame_start(t, f2) != first + count) return(HVK_ERROR);
	}

	if(pass && !pass_frames)
	{
		line = malloc((size_t) s->W * 2 * sizeof(int16_t));
		if(!line) return(HVK_OUT_OF_MEMORY);
	}

	/* the offset phasor has run over the pipeline's start-up samples as well (they pass through every process) */
	if(offset) while(s->off_steps < s->prime + first) (void) _offset_step(s);

	for(n = 0; n < count; n++)
	{
		int16_t i, q;

		/* src/video.c:2321-2335 */
		s->fm_phase = _mul(s->fm_phase, lut[iq[n * 2] - INT16_MIN]);
		i = (int16_t) (((s->fm_phase.i >> 16) * level) >> 15);
		q = (int16_t) (((s->fm_phase.q >> 16) * level) >> 15);

		if(--s->fm_counter == 0)
		{
			s->fm_phase = _renormalise(s->fm_phase);
			s->fm_counter = INT16_MAX;
		}

		if(swap)
		{
			const int16_t x = i;
			i = q;
			q = x;
		}

		if(offset)
		{
			/* src/common.h:58-67 */
			const hvk_c16_t b = _offset_step(s);
			const int32_t ri = (int32_t) i * b.i - (int32_t) q * b.q;
			const int32_t rq = (int32_t) i * b.q + (int32_t) q * b.i;
			i = (int16_t) (ri >> 15);
			q = (int16_t) (rq >> 15);
		}

		iq[n * 2 + 0] = i;
		iq[n * 2 + 1] = q;

		if(pass && !pass_frames && (n + 1) % s->W == 0)
		{
			/* the line that just ended */
			const int64_t p = first + n + 1 - s->W;
			int x;
			int r = hvk_tail_passthru_stream(s, p, s->W, line);
			if(r != HVK_OK) { free(line); return(r); }
			for(x = 0; x < s->W * 2; x++) iq[(p - first) * 2 + x] = (int16_t) (iq[(p - first) * 2 + x] + line[x]);
		}
	}

	free(line);
	if(pass_frames && count > 0)
	{
		int16_t *add = malloc((size_t) count * 2 * sizeof(int16_t));
		int r;
		if(!add) return(HVK_OUT_OF_MEMORY);
		r = hvk_tail_passthru_stream(s, first, count, add);
		if(r != HVK_OK) { free(add); return(r); }
		for(n = 0; n < count * 2; n++) iq[n] = (int16_t) (iq[n] + add[n]);
		free(add);
	}
	s->fm_pos += count;
	return(HVK_OK);
}
