/* oracle/oracle_audio.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of hacktv's audio sub-carrier stage: the per-line audio
 * process (src/video.c:3261-3450), the FM / AM phasor modulators
 * (src/video.c:2216-2276, :2343-2378), the 32 kHz soft limiter with its two
 * int32 FIRs (src/fir.c:620-694, :758-870) and the NICAM-728 encoder and
 * DQPSK modulator (src/nicam728.c).
 *
 * Everything here is evaluated one sample at a time in stream order, exactly
 * as the reference does; it is the serial ground truth the device path's
 * split (host control path + data-parallel kernels) is checked against.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

#define AUDIO_RATE 32000 /* HACKTV_AUDIO_SAMPLE_RATE, src/hacktv.h:31 */

/* Quantised (x 32767, lround) halves of the symmetric 65-tap 32 kHz audio
 * filters of src/video.c:2118-2168: index 0..32, tap[64 - k] == tap[k]. */
static const int32_t _flat_half[33] = {
	0, -26, 10, -42, 25, -68, 44, -101, 63, -133, 71, -149, 52, -130, -13, -60,
	-138, 77, -333, 283, -593, 550, -904, 856, -1235, 1169, -1552, 1450, -1814,
	1663, -1987, 1777, 30719
};
static const int32_t _50us_half[33] = {
	40, -86, 95, -158, 177, -265, 290, -399, 409, -518, 478, -552, 418, -414,
	138, -17, -437, 699, -1345, 1748, -2566, 3076, -4015, 4560, -5532, 5997,
	-6890, 7033, -7753, 6441, -7411, -19876, 81829
};
static const int32_t _75us_half[33] = {
	65, -123, 147, -227, 270, -385, 440, -580, 619, -752, 726, -799, 641, -588,
	231, 6, -616, 1073, -1956, 2632, -3763, 4603, -5910, 6798, -8169, 8898,
	-10227, 10324, -11683, 9020, -11904, -32509, 116205
};

static int32_t *_expand_half(const int32_t *half)
{
	int32_t *t = malloc(65 * sizeof(int32_t));
	int k;
	for(k = 0; k <= 32; k++) t[k] = t[64 - k] = half[k];
	return(t);
}

/* ---- limiter ---- */

static int _limiter_init(orc_limiter_t *l, int16_t level, int width, const int32_t *vhalf, const int32_t *fhalf)
{
	int i;

	memset(l, 0, sizeof(*l));
	l->ntaps = 65;
	l->vtaps = _expand_half(vhalf);
	l->ftaps = _expand_half(fhalf);
	l->vwin = calloc(l->ntaps, sizeof(int32_t));
	l->fwin = calloc(l->ntaps, sizeof(int32_t));

	l->width = width | 1;
	l->shape = malloc(sizeof(int16_t) * l->width);
	for(i = 0; i < l->width; i++)
	{
		l->shape[i] = lround((1.0 - cos(2.0 * M_PI / (l->width + 1) * (i + 1))) * 0.5 * INT16_MAX);
	}

	l->level = level;
	l->att = calloc(sizeof(int16_t), l->width);
	l->fix = calloc(sizeof(int32_t), l->width);
	l->var = calloc(sizeof(int32_t), l->width);
	l->p = 0;
	l->h = l->width / 2;

	return(0);
}

static void _limiter_free(orc_limiter_t *l)
{
	free(l->vtaps); free(l->ftaps); free(l->vwin); free(l->fwin);
	free(l->shape); free(l->att); free(l->fix); free(l->var);
}

/* One step of a 65-tap int32 FIR: newest sample in, one sample out.
 * tap[y] meets the sample that is (64 - y) steps old (src/fir.c:655-694). */
static int32_t _fir32_step(int32_t *win, int *pos, const int32_t *taps, int ntaps, int32_t in)
{
	int64_t a = 0;
	int y, p;

	win[*pos] = in;
	if(++(*pos) == ntaps) *pos = 0;

	/* *pos now indexes the oldest sample */
	for(p = *pos, y = 0; y < ntaps; y++)
	{
		a += (int64_t) win[p] * (int64_t) taps[y];
		if(++p == ntaps) p = 0;
	}

	a >>= 15;
	return(a < INT32_MIN ? INT32_MIN : (a > INT32_MAX ? INT32_MAX : a));
}

static int16_t _limiter_step(orc_limiter_t *l, int16_t vin, int16_t fin)
{
	int32_t a, b;
	int j;

	l->var[l->p] = _fir32_step(l->vwin, &l->vpos, l->vtaps, l->ntaps, vin);
	l->fix[l->p] = _fir32_step(l->fwin, &l->fpos, l->ftaps, l->ntaps, fin);
	l->att[l->p] = 0;

	if(l->fix[l->p] < -l->level) l->fix[l->p] = -l->level;
	else if(l->fix[l->p] > l->level) l->fix[l->p] = l->level;

	l->var[l->p] -= l->fix[l->p];

	if(++l->p == l->width) l->p = 0;
	if(++l->h == l->width) l->h = 0;

	a = abs(l->var[l->h] + l->fix[l->h]);
	if(a > l->level)
	{
		a = INT16_MAX - (l->level + abs(l->var[l->h]) - a) * INT16_MAX / abs(l->var[l->h]);

		for(j = 0; j < l->width; j++)
		{
			b = (a * l->shape[j]) >> 15;
			if(b > l->att[l->p]) l->att[l->p] = b;
			if(++l->p == l->width) l->p = 0;
		}
	}

	a  = l->fix[l->p];
	a += ((int64_t) l->var[l->p] * (INT16_MAX - l->att[l->p])) >> 15;

	if(a < -l->level) a = -l->level;
	else if(a > l->level) a = l->level;

	return(a);
}

/* ---- phasor modulators ---- */

static void _c32_rotate(c32_t *p, const c32_t *b)
{
	/* src/common.h:80-89: 64-bit products, floor shift by 31 */
	int64_t i = (int64_t) p->i * (int64_t) b->i - (int64_t) p->q * (int64_t) b->q;
	int64_t q = (int64_t) p->i * (int64_t) b->q + (int64_t) p->q * (int64_t) b->i;
	p->i = i >> 31;
	p->q = q >> 31;
}

static void _renormalise(orc_mod_t *m)
{
	/* src/video.c:2266-2275 */
	if(--m->counter == 0)
	{
		double ra = atan2(m->phase.q, m->phase.i);
		m->phase.i = lround(cos(ra) * INT32_MAX);
		m->phase.q = lround(sin(ra) * INT32_MAX);
		m->counter = INT16_MAX;
	}
}

static void _fm_add(orc_mod_t *m, int16_t *dst, int16_t sample)
{
	_c32_rotate(&m->phase, &m->lut[sample - INT16_MIN]);
	dst[0] += ((m->phase.i >> 16) * m->level) >> 15;
	dst[1] += ((m->phase.q >> 16) * m->level) >> 15;
	_renormalise(m);
}

static void _am_add(orc_mod_t *m, int16_t *dst, int16_t sample)
{
	_c32_rotate(&m->phase, &m->delta);
	sample = ((int32_t) sample - INT16_MIN) / 2;
	dst[0] += ((((m->phase.i >> 16) * sample) >> 15) * m->level) >> 15;
	dst[1] += ((((m->phase.q >> 16) * sample) >> 15) * m->level) >> 15;
	_renormalise(m);
}

/* ---- NICAM-728 ---- */

static const int32_t _j17_half[42] = {
	/* src/nicam728.c:37-44, first 42 of 83 (symmetric about index 41) */
	-1, 0, -1, -1, -1, -1, -1, -1, -1, -1, -2, -2, -3, -3, -3, -3, -5, -5,
	-6, -7, -9, -10, -13, -14, -18, -21, -27, -32, -42, -51, -69, -86, -120,
	-159, -233, -332, -524, -814, -1402, -2372, -4502, 25590
};

static int _j17(int k) { return(_j17_half[k <= 41 ? k : 82 - k]); }

/* scale factor code and down-shift for coding range b (src/nicam728.c:59-68) */
static const int _sf_factor[8] = { 0, 1, 2, 4, 3, 5, 6, 7 };
static const int _sf_shift[8]  = { 2, 2, 2, 2, 3, 4, 5, 6 };

static int _coding_range(const int16_t *pcm)
{
	int i, b = 1;
	for(i = 0; b < 7 && i < 32; i++, pcm += 2)
	{
		int16_t v = (*pcm < 0) ? ~*pcm : *pcm;
		while(b < 7 && v >> (b + 8)) b++;
	}
	return(b);
}

static uint8_t _parity(unsigned int v)
{
	uint8_t p = 0;
	while(v) { p ^= v & 1; v >>= 1; }
	return(p);
}

static void _nicam_encode_frame(orc_nicam_t *n)
{
	int16_t w[64];
	int rng[2];
	int x, xi, k;

	/* J.17 pre-emphasis, oldest sample meets tap 0 (src/nicam728.c:147-162) */
	for(x = 0; x < 32; x++)
	{
		int32_t l = 0, r = 0;

		n->fir_l[n->fir_p] = n->audio[x * 2 + 0];
		n->fir_r[n->fir_p] = n->audio[x * 2 + 1];
		if(++n->fir_p == 83) n->fir_p = 0;

		for(k = 0; k < 83; k++)
		{
			l += (int32_t) n->fir_l[n->fir_p] * _j17(k);
			r += (int32_t) n->fir_r[n->fir_p] * _j17(k);
			if(++n->fir_p == 83) n->fir_p = 0;
		}

		w[x * 2 + 0] = l >> 15;
		w[x * 2 + 1] = r >> 15;
	}

	/* companding to 10 bits + parity + scale-factor signalling (:165-182) */
	rng[0] = _coding_range(w + 0);
	rng[1] = _coding_range(w + 1);

	for(x = 0; x < 64; x++)
	{
		w[x] = (w[x] >> _sf_shift[rng[x & 1]]) & 0x3FF;
		w[x] |= _parity(w[x] >> 4) << 10;
		if(x < 54) w[x] ^= ((_sf_factor[rng[x & 1]] >> (2 - (x / 2 % 3))) & 1) << 10;
	}

	/* header (:204-218) */
	memset(n->frame, 0, 91);
	n->frame[0] = 0x4E;
	n->frame[1]  = (((~n->frame_no) >> 3) & 1) << 7;
	n->frame[1] |= ((n->mode >> 2) & 1) << 6;
	n->frame[1] |= ((n->mode >> 1) & 1) << 5;
	n->frame[1] |= ((n->mode >> 0) & 1) << 4;
	n->frame[1] |= (n->reserve & 1) << 3;

	/* bit interleave, LSB first, 16 bits apart (:221-240) */
	for(xi = x = 0; x < 64; x++)
	{
		int b;
		for(b = 0; b < 11; b++, w[x] >>= 1)
		{
			if(w[x] & 1) n->frame[3 + (xi / 8)] |= 1 << (7 - (xi % 8));
			xi += 16;
			if(xi >= 728 - 24) xi -= 728 - 24 - 1;
		}
	}

	/* scrambling (:243-246) */
	for(x = 0; x < 90; x++) n->frame[x + 1] ^= n->prn[x];

	n->frame_no++;
}

/* the encoder without a modulator: what sound-in-syncs has (src/sis.c:141, src/nicam728.c:96-126) */
void orc_nicam_encoder_init(orc_nicam_t *n, uint8_t mode, uint8_t reserve)
{
	int x, poly = 0x1FF;
	memset(n, 0, sizeof(*n));
	n->mode = mode;
	n->reserve = reserve;
	for(x = 0; x < 90; x++)
	{
		int i;
		n->prn[x] = 0;
		for(i = 0; i < 8; i++)
		{
			uint8_t b = (poly & 1) ^ ((poly >> 4) & 1);
			poly >>= 1;
			poly |= b << 8;
			n->prn[x] = (n->prn[x] << 1) | b;
		}
	}
}

void orc_nicam_encode(orc_nicam_t *n) { _nicam_encode_frame(n); }

static double _rrc(double x, double b, double t)
{
	/* src/common.c:259-283 */
	double r;
	if(x == 0) r = (1.0 / t) * (1.0 + b * (4.0 / M_PI - 1));
	else if(fabs(x) == t / (4.0 * b))
	{
		r = b / (t * sqrt(2.0)) * ((1.0 + 2.0 / M_PI) * sin(M_PI / (4.0 * b)) + (1.0 - 2.0 / M_PI) * cos(M_PI / (4.0 * b)));
	}
	else
	{
		double t1 = (4.0 * b * (x / t));
		double t2 = (sin(M_PI * (x / t) * (1.0 - b)) + 4.0 * b * (x / t) * cos(M_PI * (x / t) * (1.0 + b)));
		double t3 = (M_PI * (x / t) * (1.0 - t1 * t1));
		r = (1.0 / t) * (t2 / t3);
	}
	return(r);
}

static double _hamming(double x)
{
	if(x < -1 || x > 1) return(0);
	return(0.54 - 0.46 * cos((M_PI * (1.0 + x))));
}

static unsigned int _ugcd(unsigned int a, unsigned int b)
{
	unsigned int c;
	while((c = a % b)) { a = b; b = c; }
	return(b);
}

static void _nicam_init(orc_nicam_t *n, unsigned int sample_rate, unsigned int frequency, double beta, double level)
{
	double sps, d;
	int x, h, g, poly;

	memset(n, 0, sizeof(*n));
	n->on = 1;
	n->mode = 0x00;   /* NICAM_MODE_STEREO */
	n->reserve = 1;

	/* pulse shape (src/nicam728.c:265-292) */
	sps = (double) sample_rate / 364000.0;
	n->ntaps = ((unsigned int) (sps * 5) + 1) | 1;
	n->taps = malloc(sizeof(int16_t) * n->ntaps);
	h = n->ntaps / 2;
	for(x = -h; x <= h; x++)
	{
		double t = ((double) x) / sps;
		double r = _rrc(t, beta, 1.0) * _hamming((double) x / h);
		r *= M_SQRT1_2 * INT16_MAX * level;
		n->taps[x + h] = lround(r);
	}

	n->bb = calloc(n->ntaps, sizeof(c16_t));
	n->bb_pos = 0;
	n->bb_len = 0;

	/* symbol timing (:301-307) */
	g = _ugcd(sample_rate, 364000);
	n->decimation = 364000 / g;
	n->sps = (sample_rate + 364000 - 1) / 364000;
	n->dsl = (n->sps * n->decimation) % (sample_rate / g);
	n->ds = 0;

	/* mixer (:309-314, src/common.c:209-229) */
	g = _ugcd(sample_rate, frequency);
	n->cc_len = sample_rate / g;
	n->cc = malloc(n->cc_len * sizeof(c16_t));
	d = 2.0 * M_PI / n->cc_len * (frequency / g);
	for(x = 0; x < n->cc_len; x++)
	{
		n->cc[x].i = round(cos(d * x) * 1.0 * INT16_MAX);
		n->cc[x].q = round(sin(d * x) * 1.0 * INT16_MAX);
	}
	n->cc_pos = 0;

	/* scrambler sequence (src/nicam728.c:96-126) */
	poly = 0x1FF;
	for(x = 0; x < 90; x++)
	{
		int i;
		n->prn[x] = 0;
		for(i = 0; i < 8; i++)
		{
			uint8_t b = (poly & 1) ^ ((poly >> 4) & 1);
			poly >>= 1;
			poly |= b << 8;
			n->prn[x] = (n->prn[x] << 1) | b;
		}
	}

	n->frame_bit = 728;
}

/* src/nicam728.c:342-411 */
static void _nicam_output(orc_nicam_t *n, int16_t *iq, int samples)
{
	static const int step[4] = { 0, 3, 1, 2 };
	static const int syms[4] = { 0, 1, 3, 2 };
	int x, i;

	for(x = 0; x < samples;)
	{
		for(; x < samples && n->bb_len; x++, n->bb_len--)
		{
			c16_t *bb = &n->bb[n->bb_pos];
			const c16_t *cc = &n->cc[n->cc_pos];
			int32_t mi = (int32_t) bb->i * cc->i - (int32_t) bb->q * cc->q;
			int32_t mq = (int32_t) bb->i * cc->q + (int32_t) bb->q * cc->i;

			iq[x * 2 + 0] += mi >> 15;
			iq[x * 2 + 1] += mq >> 15;

			bb->i = bb->q = 0;
			if(++n->bb_pos == n->ntaps) n->bb_pos = 0;
			if(++n->cc_pos == n->cc_len) n->cc_pos = 0;
		}

		if(n->bb_len > 0) break;

		if(n->frame_bit == 728)
		{
			_nicam_encode_frame(n);
			n->frame_bit = 0;
		}

		n->dsym += step[(n->frame[n->frame_bit >> 3] >> (6 - (n->frame_bit & 0x07))) & 0x03];
		n->dsym &= 0x03;
		n->frame_bit += 2;

		for(i = 0; i < n->ntaps; i++)
		{
			int p = n->bb_pos + i;
			int16_t r = n->taps[i];
			if(p >= n->ntaps) p -= n->ntaps;
			n->bb[p].i += (syms[n->dsym] & 1 ? r : -r);
			n->bb[p].q += (syms[n->dsym] & 2 ? r : -r);
		}

		n->bb_len = n->sps;
		n->ds += n->dsl;
		if(n->ds >= n->decimation)
		{
			n->bb_len--;
			n->ds -= n->decimation;
		}
	}
}

/* ---- set-up ---- */

static void _mod_lut(orc_mod_t *m, int sample_rate, double frequency, double deviation)
{
	int r;
	m->lut = malloc(sizeof(c32_t) * 65536);
	for(r = INT16_MIN; r <= INT16_MAX; r++)
	{
		double d = 2.0 * M_PI / sample_rate * (frequency + (double) r / INT16_MAX * deviation);
		m->lut[r - INT16_MIN].i = lround(cos(d) * INT32_MAX);
		m->lut[r - INT16_MIN].q = lround(sin(d) * INT32_MAX);
	}
}

int orc_audio_init(orc_t *s)
{
	const hvk_config_t *c = &s->conf;
	double slevel = c->modulation == HVK_FM ? 1.0 : c->level;

	s->interp = 0;

	/* src/video.c:4404-4441 */
	if(c->fm_mono_level > 0 && c->fm_mono_carrier != 0)
	{
		orc_mod_t *m = &s->fm_mono;
		m->on = 1;
		m->level = round(INT16_MAX * (c->fm_mono_level * slevel));
		m->counter = INT16_MAX;
		m->phase.i = INT32_MAX;
		m->phase.q = 0;
		_mod_lut(m, s->sample_rate, c->fm_mono_carrier, c->fm_mono_deviation);

		if(c->fm_mono_preemph == HVK_50US || c->fm_mono_preemph == HVK_75US)
		{
			_limiter_init(&m->lim, INT16_MAX, 21, c->fm_mono_preemph == HVK_50US ? _50us_half : _75us_half, _flat_half);
			m->has_lim = 1;
		}
		else if(c->fm_mono_preemph != 0) return(-1); /* J.17 FM pre-emphasis: not restated */
	}

	/* Zweikanalton (src/video.c:4375-4400): the second carrier is derived from the first, the pilot
	 * is a 54.6875 kHz tone amplitude modulated with the 117.5 Hz "stereo" identification */
	if(c->a2stereo && s->fm_mono.on)
	{
		orc_mod_t *m = &s->fm_right;
		double carrier, d;

		s->a2_system_m = c->fm_mono_carrier == 4500000;
		carrier = c->fm_mono_carrier + (s->a2_system_m ? 224213 : 242187.5);

		m->on = 1;
		m->level = round(INT16_MAX * ((c->fm_mono_level * 0.446684) * slevel));
		m->counter = INT16_MAX;
		m->phase.i = INT32_MAX;
		m->phase.q = 0;
		_mod_lut(m, s->sample_rate, carrier, c->fm_mono_deviation);
		if(c->fm_mono_preemph == HVK_50US || c->fm_mono_preemph == HVK_75US)
		{
			_limiter_init(&m->lim, INT16_MAX, 21, c->fm_mono_preemph == HVK_50US ? _50us_half : _75us_half, _flat_half);
			m->has_lim = 1;
		}

		d = 2.0 * M_PI / s->sample_rate * (s->a2_system_m ? 55.06993e3 : 54.6875e3);
		s->a2_pilot.level = round(INT16_MAX * 0.05);
		s->a2_pilot.counter = INT16_MAX;
		s->a2_pilot.phase.i = INT32_MAX;
		s->a2_pilot.delta.i = lround(cos(d) * INT32_MAX);
		s->a2_pilot.delta.q = lround(sin(d) * INT32_MAX);

		d = 2.0 * M_PI / s->sample_rate * (s->a2_system_m ? 149.9 : 117.5);
		s->a2_signal.level = round(INT16_MAX * 1.0);
		s->a2_signal.counter = INT16_MAX;
		s->a2_signal.phase.i = INT32_MAX;
		s->a2_signal.delta.i = lround(cos(d) * INT32_MAX);
		s->a2_signal.delta.q = lround(sin(d) * INT32_MAX);
	}

	/* src/video.c:4522-4533; A2 stereo switches NICAM off (:4397-4399) */
	if(c->nicam_level > 0 && c->nicam_carrier != 0 && !c->a2stereo)
	{
		_nicam_init(&s->nicam, s->sample_rate, c->nicam_carrier, c->nicam_beta, c->nicam_level * slevel);
		s->nicam_buf_len = 0;
	}

	/* src/video.c:4550-4558, :2343-2357 */
	if(c->am_audio_level > 0 && c->am_mono_carrier != 0)
	{
		orc_mod_t *m = &s->am_mono;
		double d = 2.0 * M_PI / s->sample_rate * c->am_mono_carrier;
		m->on = 1;
		m->level = round(INT16_MAX * (c->am_audio_level * slevel));
		m->counter = INT16_MAX;
		m->phase.i = INT32_MAX;
		m->phase.q = 0;
		m->delta.i = lround(cos(d) * INT32_MAX);
		m->delta.q = lround(sin(d) * INT32_MAX);
	}

	return(0);
}

void orc_audio_free(orc_t *s)
{
	free(s->fm_mono.lut);
	if(s->fm_mono.has_lim) _limiter_free(&s->fm_mono.lim);
	free(s->fm_right.lut);
	if(s->fm_right.has_lim) _limiter_free(&s->fm_right.lim);
	free(s->nicam.taps);
	free(s->nicam.bb);
	free(s->nicam.cc);
}

/* One emitted line of the audio process (src/video.c:3261-3450): the tick
 * loop over every sample first, then the NICAM modulator over the line.
 * carrier_tap, if not NULL, receives only the serial-carrier part. */
void orc_audio_line(orc_t *s, int16_t *iq, int width, int16_t *carrier_tap)
{
	const hvk_config_t *c = &s->conf;
	int16_t audio[2] = { 0, 0 };
	int x, i;

	for(x = 0; x < width; x++)
	{
		int16_t add[2] = { 0, 0 };

		/* 32 kHz tick by accumulation (:3273-3276) */
		s->interp += AUDIO_RATE;
		if(s->interp >= s->sample_rate)
		{
			s->interp -= s->sample_rate;

			/* next source sample; the test source hands the same loop back
			 * each time it runs dry (:3278-3304, src/av_test.c:54-60) */
			if(s->audio_src && s->audio_pos >= s->audio_len && s->audio_loop) s->audio_pos = 0;
			if(s->audio_src && s->audio_pos < s->audio_len)
			{
				for(i = 0; i < 2; i++)
				{
					int32_t v = ((int32_t) s->audio_src[s->audio_pos * 2 + i] * c->volume + 128) >> 8;
					audio[i] = (v < INT16_MIN ? INT16_MIN : (v > INT16_MAX ? INT16_MAX : v));
				}
				s->audio_pos++;
			}
			else
			{
				audio[0] = audio[1] = 0;
			}

			if(s->am_mono.on) s->am_mono.sample = (audio[0] + audio[1]) / 2;

			if(s->fm_mono.on)
			{
				s->fm_mono.sample = (audio[0] + audio[1]) / 2;
				if(s->fm_mono.has_lim)
				{
					s->fm_mono.sample = _limiter_step(&s->fm_mono.lim, s->fm_mono.sample, s->fm_mono.sample);
				}
				/* room for the pilot (:3325-3327) */
				if(c->a2stereo) s->fm_mono.sample *= 0.95;
			}

			if(s->fm_right.on)
			{
				s->fm_right.sample = audio[1];
				if(s->fm_right.has_lim)
				{
					s->fm_right.sample = _limiter_step(&s->fm_right.lim, s->fm_right.sample, s->fm_right.sample);
				}
				s->fm_right.sample *= 0.95;
			}

			if(s->nicam.on)
			{
				s->nicam_buf[s->nicam_buf_len++] = audio[0];
				s->nicam_buf[s->nicam_buf_len++] = audio[1];
				if(s->nicam_buf_len == 64)
				{
					memcpy(s->nicam.audio, s->nicam_buf, sizeof(int16_t) * 64);
					s->nicam_buf_len = 0;
				}
			}
		}

		if(s->fm_mono.on) _fm_add(&s->fm_mono, add, s->fm_mono.sample);
		if(s->fm_right.on)
		{
			/* src/video.c:3402-3424 */
			int16_t a2 = s->fm_right.sample;
			int16_t s1[2] = { 0, 0 }, s2[2] = { 0, 0 };

			if(s->a2_system_m) a2 = s->fm_mono.sample - s->fm_right.sample;    /* L - R on system M */
			_am_add(&s->a2_signal, s1, 0);
			_am_add(&s->a2_pilot, s2, s1[0]);
			a2 += s2[0];

			_fm_add(&s->fm_right, add, a2);
		}
		if(s->am_mono.on) _am_add(&s->am_mono, add, s->am_mono.sample);

		if(iq)
		{
			iq[x * 2 + 0] += add[0];
			iq[x * 2 + 1] += add[1];
		}
		if(carrier_tap)
		{
			carrier_tap[x * 2 + 0] = add[0];
			carrier_tap[x * 2 + 1] = add[1];
		}
	}

	if(s->nicam.on)
	{
		if(iq) _nicam_output(&s->nicam, iq, width);
		else
		{
			int16_t *tmp = calloc(width * 2, sizeof(int16_t));
			_nicam_output(&s->nicam, tmp, width);
			free(tmp);
		}
	}
}
