/* oracle/oracle_tail.c -- TEST INFRASTRUCTURE (not product code).
 *
 * The line processes that follow the audio process in the reference's chain
 * (src/video.c:4563-4645), restated one line at a time exactly as the
 * reference runs them: FM video, swap_iq, frequency offset, passthru.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

/* src/common.h:80-89 */
static void _rotate(c32_t *p, const c32_t *d)
{
	int64_t i = (int64_t) p->i * (int64_t) d->i - (int64_t) p->q * (int64_t) d->q;
	int64_t q = (int64_t) p->i * (int64_t) d->q + (int64_t) p->q * (int64_t) d->i;
	p->i = i >> 31;
	p->q = q >> 31;
}

static void _correct(c32_t *p)
{
	double ra = atan2(p->q, p->i);
	p->i = lround(cos(ra) * INT32_MAX);
	p->q = lround(sin(ra) * INT32_MAX);
}

int orc_tail_init(orc_t *s)
{
	const hvk_config_t *c = &s->conf;
	int r;

	if(c->modulation == HVK_FM)
	{
		/* src/video.c:4566, :2218-2243: frequency 0 */
		s->fm_video.on = 1;
		s->fm_video.level = round(INT16_MAX * (c->fm_level * c->level));
		s->fm_video.counter = INT16_MAX;
		s->fm_video.phase.i = INT32_MAX;
		s->fm_video.phase.q = 0;
		s->fm_video.lut = malloc(sizeof(c32_t) * 65536);
		if(!s->fm_video.lut) return(-1);

		for(r = INT16_MIN; r <= INT16_MAX; r++)
		{
			double d = 2.0 * M_PI / s->sample_rate * (0 + (double) r / INT16_MAX * c->fm_deviation);
			s->fm_video.lut[r - INT16_MIN].i = lround(cos(d) * INT32_MAX);
			s->fm_video.lut[r - INT16_MIN].q = lround(sin(d) * INT32_MAX);
		}
	}

	if(c->offset != 0)
	{
		/* src/video.c:4592-4604 */
		double d = 2.0 * M_PI / s->sample_rate * c->offset;
		s->offset_counter = INT16_MAX;
		s->offset_phase.i = INT16_MAX;
		s->offset_phase.q = 0;
		s->offset_delta.i = lround(cos(d) * INT32_MAX);
		s->offset_delta.q = lround(sin(d) * INT32_MAX);
	}

	return(0);
}

void orc_tail_free(orc_t *s)
{
	free(s->fm_video.lut);
	free(s->passline);
}

void orc_set_passthru(orc_t *s, const int16_t *iq, long nsamples)
{
	s->pass_src = iq;
	s->pass_len = nsamples;
	s->pass_pos = 0;
	s->pass_eof = 0;
}

/* fread() of up to n samples from the passthru source */
static long _pass_read(orc_t *s, int16_t *dst, long n)
{
	long left = s->pass_len - s->pass_pos;
	if(n > left) { n = left; s->pass_eof = 1; }   /* a short read sets the end-of-file flag */
	if(n > 0) memcpy(dst, s->pass_src + s->pass_pos * 2, n * 2 * sizeof(int16_t));
	s->pass_pos += n;
	return(n);
}

void orc_tail_line(orc_t *s, int16_t *iq, int width)
{
	const hvk_config_t *c = &s->conf;
	int x;

	/* src/video.c:3452-3464 with :2321-2335 */
	if(s->fm_video.on)
	{
		orc_mod_t *m = &s->fm_video;
		for(x = 0; x < width; x++)
		{
			int16_t sample = iq[x * 2];
			_rotate(&m->phase, &m->lut[sample - INT16_MIN]);
			iq[x * 2 + 0] = ((m->phase.i >> 16) * m->level) >> 15;
			iq[x * 2 + 1] = ((m->phase.q >> 16) * m->level) >> 15;
			if(--m->counter == 0)
			{
				_correct(&m->phase);
				m->counter = INT16_MAX;
			}
		}
	}

	/* src/video.c:3466-3480 */
	if(c->swap_iq)
	{
		for(x = 0; x < width; x++)
		{
			int16_t t = iq[x * 2 + 0];
			iq[x * 2 + 0] = iq[x * 2 + 1];
			iq[x * 2 + 1] = t;
		}
	}

	/* src/video.c:3482-3515 */
	if(c->offset != 0)
	{
		for(x = 0; x < width; x++)
		{
			int32_t ai = iq[x * 2 + 0], aq = iq[x * 2 + 1], bi, bq;

			_rotate(&s->offset_phase, &s->offset_delta);
			bi = s->offset_phase.i >> 16;
			bq = s->offset_phase.q >> 16;

			/* cint16_mul, src/common.h:58-67 */
			iq[x * 2 + 0] = (int16_t) ((ai * (int16_t) bi - aq * (int16_t) bq) >> 15);
			iq[x * 2 + 1] = (int16_t) ((ai * (int16_t) bq + aq * (int16_t) bi) >> 15);

			if(--s->offset_counter == 0)
			{
				_correct(&s->offset_phase);
				s->offset_counter = INT16_MAX;
			}
		}
	}

	/* src/video.c:3517-3541 */
	if(c->passthru && s->pass_src)
	{
		long got;

		if(s->pass_eof) return;

		if(!s->passline) s->passline = calloc(width * 2, sizeof(int16_t));

		for(x = 0; x < width;)
		{
			got = _pass_read(s, s->passline + x * 2, width - x);
			if(got == 0) return;
			x += got;
		}

		for(x = 0; x < width * 2; x++) iq[x] += s->passline[x];
	}
}
