/* hvk_audio.c -- host audio-rate control path of the MI355X engine.
 *
 * The reference adds its audio sub-carriers inside a per-sample loop on a
 * pipeline thread (_vid_audio_process, src/video.c:3261-3450). Two different
 * kinds of work hide in that loop:
 *
 *  (1) serial, non-associative recurrences: the FM / AM carrier phasors are
 *      advanced by a complex multiply that is floored after every step
 *      (cint32_mul, src/common.h:80-89; _fm_modulator_add, src/video.c:
 *      2259-2276; _am_modulator_add, :2359-2378), and re-normalised through
 *      libm every 32767 samples. Sample n cannot be had without sample n-1,
 *      bit for bit. This stays on the host, on one core, and
 *      its result -- the summed int16 I/Q contribution of all such carriers,
 *      4 bytes per sample -- is a side INPUT of the device path;
 *
 *  (2) everything else: the 32 kHz control logic (volume, mono mix, soft
 *      limiter with its two 65-tap FIRs, NICAM-728 companding / interleaving
 *      / scrambling, the differential QPSK state) is cheap and sequential and
 *      is done here too, but the per-sample NICAM work -- pulse shaping by
 *      overlap-add and the mix onto the 6.552 MHz carrier, src/nicam728.c:
 *      342-411 -- is data parallel and is done in the filter kernel from the
 *      symbol values this file emits (1 byte per 44 samples).
 *
 * Ordering follows the reference exactly, including its line granularity:
 * for every line of `width` samples the tick loop runs first and the NICAM
 * modulator afterwards, which decides which 32-sample audio block a NICAM
 * frame carries.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "hvk_internal.h"

#define AUDIO_RATE      32000   /* src/hacktv.h:31 */
#define NICAM_SYMRATE   364000
#define NICAM_FRAME_BITS 728
#define J17_TAPS        83
#define SYM_HISTORY     64      /* symbols kept behind the newest request */

/* ---- 65-tap int32 FIR, one sample at a time (src/fir.c:655-694) ---- */
typedef struct {
	const int32_t *taps;    /* taps[y] meets the sample 64 - y steps old */
	int32_t hist[65];
	int pos;                /* where the next sample goes == oldest sample */
} _fir65_t;

static int32_t _fir65(_fir65_t *f, int32_t in)
{
	int64_t acc = 0;
	int y, p;

	f->hist[f->pos] = in;
	f->pos = f->pos == 64 ? 0 : f->pos + 1;

	for(y = 0, p = f->pos; y < 65; y++)
	{
		acc += (int64_t) f->hist[p] * f->taps[y];
		p = p == 64 ? 0 : p + 1;
	}

	acc >>= 15;
	if(acc < INT32_MIN) return(INT32_MIN);
	if(acc > INT32_MAX) return(INT32_MAX);
	return((int32_t) acc);
}

/* ---- look-ahead soft limiter, 21 samples wide ----
 * What the reference's limiter_process() computes (src/fir.c:748-870, called with the same sample on all three
 * inputs, src/video.c:3322): the sample goes through two 65-tap FIRs -- "flat" (clipped to the ceiling on the spot)
 * and pre-emphasised, of which only the excess over the flat branch is kept. Ten samples later the pair is looked at
 * again: if flat + excess would overshoot the ceiling, the share of the excess that has to go is worked out and,
 * shaped by a 21-point window centred there, raises the cut of the 21 samples around it. What leaves is the oldest
 * sample: flat + excess x (1 - cut). All integer, every rounding as in the reference; laid out here as one ring
 * of (flat, excess, cut) slots with the three steps as functions of their own. */
#define LIM_SPAN 21
typedef struct {
	int32_t flat, excess;
	int32_t cut;                /* Q15 share of the excess to take away, 0 .. 32767 */
} _lim_slot_t;

typedef struct {
	_fir65_t pre, flat;         /* the two branches' filters */
	const int16_t *shape;       /* the window a cut is spread with */
	int32_t ceiling;
	_lim_slot_t ring[LIM_SPAN];
	int oldest;                 /* slot that leaves next (and is overwritten by the newcomer) */
	int centre;                 /* slot LIM_SPAN / 2 samples behind the newcomer */
} _limiter_t;

static inline int _lim_next(int i) { return(i + 1 == LIM_SPAN ? 0 : i + 1); }

/* the newcomer takes the place of the sample that left last time */
static void _lim_take(_limiter_t *l, int16_t in)
{
	_lim_slot_t *s = &l->ring[l->oldest];
	const int32_t pre = _fir65(&l->pre, in);
	int32_t flat = _fir65(&l->flat, in);

	if(flat < -l->ceiling) flat = -l->ceiling;
	else if(flat > l->ceiling) flat = l->ceiling;

	s->flat = flat;
	s->excess = pre - flat;
	s->cut = 0;

	l->oldest = _lim_next(l->oldest);
	l->centre = _lim_next(l->centre);
}

/* does the sample in the middle of the ring overshoot? Then every slot's cut is raised to what the window, centred
 * there, asks of it (slot `oldest` is the window's first point) */
static void _lim_look_ahead(_limiter_t *l)
{
	const _lim_slot_t *c = &l->ring[l->centre];
	const int32_t mag = abs(c->excess + c->flat);
	int32_t need;
	int i, at;

	if(mag <= l->ceiling) return;

	need = INT16_MAX - (l->ceiling + abs(c->excess) - mag) * INT16_MAX / abs(c->excess);
	for(i = 0, at = l->oldest; i < LIM_SPAN; i++, at = _lim_next(at))
	{
		const int32_t want = (need * l->shape[i]) >> 15;
		if(want > l->ring[at].cut) l->ring[at].cut = want;
	}
}

static int16_t _lim_give(const _limiter_t *l)
{
	const _lim_slot_t *s = &l->ring[l->oldest];
	/* the cut is held as the reference's int16 */
	int32_t v = s->flat + (int32_t) (((int64_t) s->excess * (INT16_MAX - (int16_t) s->cut)) >> 15);
	if(v < -l->ceiling) v = -l->ceiling;
	else if(v > l->ceiling) v = l->ceiling;
	return((int16_t) v);
}

static int16_t _limit(_limiter_t *l, int16_t in)
{
	_lim_take(l, in);
	_lim_look_ahead(l);
	return(_lim_give(l));
}

/* ---- carrier phasor ---- */
typedef struct {
	int on;
	int32_t pi, pq;         /* phase, Q31 */
	int32_t counter;        /* steps until the amplitude correction */
	int32_t level;
	int16_t sample;         /* modulating sample in force */
} _phasor_t;

static inline void _step(_phasor_t *p, int32_t ci, int32_t cq)
{
	int64_t i = (int64_t) p->pi * ci - (int64_t) p->pq * cq;
	int64_t q = (int64_t) p->pi * cq + (int64_t) p->pq * ci;
	p->pi = (int32_t) (i >> 31);
	p->pq = (int32_t) (q >> 31);
}

static inline void _correct(_phasor_t *p)
{
	/* amplitude drift correction every INT16_MAX steps (src/video.c:2266-2275) */
	if(--p->counter == 0)
	{
		double ra = atan2(p->pq, p->pi);
		p->pi = lround(cos(ra) * INT32_MAX);
		p->pq = lround(sin(ra) * INT32_MAX);
		p->counter = INT16_MAX;
	}
}

/* ---- NICAM-728 framing ---- */
typedef struct {
	unsigned int frame_no;
	uint8_t prn[90];
	int16_t l[J17_TAPS], r[J17_TAPS];
	int pos;
	int16_t block[64];      /* audio block the next frame will carry */
	int16_t fill[64];
	int fill_len;
	uint8_t bits[91];       /* current frame */
	int bit;                /* next bit pair to send */
	int dsym;               /* differential phase state */
	/* symbol schedule */
	int64_t k;              /* index of the next symbol to be created */
	int64_t next_start;     /* its first sample */
	int sps, dsl, decimation, ds;
	int reserve;            /* control bit C4: 1 on the NICAM carrier (src/video.c:4524), 0 in sound-in-syncs (src/video.c:4332) */
} _nicam_t;

/* J.17 pre-emphasis taps at 32 kHz (src/nicam728.c:37-44), first half */
static const int16_t _j17[42] = {
	-1, 0, -1, -1, -1, -1, -1, -1, -1, -1, -2, -2, -3, -3, -3, -3, -5, -5,
	-6, -7, -9, -10, -13, -14, -18, -21, -27, -32, -42, -51, -69, -86, -120,
	-159, -233, -332, -524, -814, -1402, -2372, -4502, 25590
};

static void _nicam_reset(_nicam_t *n, const hvk_tables_t *t)
{
	int x, i, lfsr = 0x1FF;

	memset(n, 0, sizeof(*n));
	n->reserve = 1;

	/* 9-bit LFSR x^9 + x^4 + 1, all ones start (src/nicam728.c:96-126) */
	for(x = 0; x < 90; x++)
	{
		for(i = 0; i < 8; i++)
		{
			int b = (lfsr ^ (lfsr >> 4)) & 1;
			lfsr = (lfsr >> 1) | (b << 8);
			n->prn[x] = (n->prn[x] << 1) | b;
		}
	}

	n->bit = NICAM_FRAME_BITS; /* a frame is built before the first symbol */
	n->sps = t->k.nicam_sps;
	n->dsl = t->k.nicam_dsl;
	n->decimation = t->k.nicam_decimation;
}

static int _range_of(const int16_t *pcm)
{
	/* smallest coding range that holds every sample of the block
	 * (src/nicam728.c:70-94) */
	int i, b = 1;
	for(i = 0; i < 32 && b < 7; i++)
	{
		int16_t m = pcm[i * 2] < 0 ? ~pcm[i * 2] : pcm[i * 2];
		while(b < 7 && (m >> (b + 8))) b++;
	}
	return(b);
}

static int _even_parity6(int v)
{
	v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
	return(v & 1);
}

static void _nicam_build_frame(_nicam_t *n)
{
	/* range -> (3-bit scale factor code, down shift), src/nicam728.c:59-68 */
	static const uint8_t code[8]  = { 0, 1, 2, 4, 3, 5, 6, 7 };
	static const uint8_t shift[8] = { 2, 2, 2, 2, 3, 4, 5, 6 };
	int16_t w[64];
	int rng[2], x, k, at;

	/* pre-emphasis: the newest sample is written, then the 83 taps sweep the ring from the oldest sample
	 * (src/nicam728.c:147-162). Here the history is kept in order, oldest first (n->l / n->r: the last 83
	 * samples), so output x is the plain dot product of samples x .. x + 82 of history + block with the taps */
	{
		static int16_t taps[88];        /* the 83 taps, zero padded to whole vectors */
		int16_t hl[120], hr[120];

		if(taps[41] == 0) for(k = 0; k < J17_TAPS; k++) taps[k] = _j17[k <= 41 ? k : 82 - k];

		memcpy(hl, n->l + 1, sizeof(int16_t) * (J17_TAPS - 1));
		memcpy(hr, n->r + 1, sizeof(int16_t) * (J17_TAPS - 1));
		for(x = 0; x < 32; x++)
		{
			hl[J17_TAPS - 1 + x] = n->block[x * 2 + 0];
			hr[J17_TAPS - 1 + x] = n->block[x * 2 + 1];
		}
		memset(hl + J17_TAPS - 1 + 32, 0, sizeof(int16_t) * (120 - (J17_TAPS - 1 + 32)));
		memset(hr + J17_TAPS - 1 + 32, 0, sizeof(int16_t) * (120 - (J17_TAPS - 1 + 32)));

		for(x = 0; x < 32; x++)
		{
			int32_t al = 0, ar = 0;
#if defined(__SSE2__)
			__m128i sl = _mm_setzero_si128(), sr = _mm_setzero_si128();
			int32_t q[4];
			for(k = 0; k < 88; k += 8)
			{
				const __m128i t = _mm_loadu_si128((const __m128i *) (taps + k));
				sl = _mm_add_epi32(sl, _mm_madd_epi16(_mm_loadu_si128((const __m128i *) (hl + x + k)), t));
				sr = _mm_add_epi32(sr, _mm_madd_epi16(_mm_loadu_si128((const __m128i *) (hr + x + k)), t));
			}
			/* (sums of int32 wrap the same way in any order) */
			_mm_storeu_si128((__m128i *) q, sl); al = (int32_t) ((uint32_t) q[0] + (uint32_t) q[1] + (uint32_t) q[2] + (uint32_t) q[3]);
			_mm_storeu_si128((__m128i *) q, sr); ar = (int32_t) ((uint32_t) q[0] + (uint32_t) q[1] + (uint32_t) q[2] + (uint32_t) q[3]);
#else
			for(k = 0; k < J17_TAPS; k++)
			{
				al += (int32_t) hl[x + k] * taps[k];
				ar += (int32_t) hr[x + k] * taps[k];
			}
#endif
			w[x * 2 + 0] = (int16_t) (al >> 15);
			w[x * 2 + 1] = (int16_t) (ar >> 15);
		}

		memcpy(n->l, hl + 32 - 1, sizeof(int16_t) * J17_TAPS);
		memcpy(n->r, hr + 32 - 1, sizeof(int16_t) * J17_TAPS);
	}

	rng[0] = _range_of(w + 0);
	rng[1] = _range_of(w + 1);

	/* 10-bit companded sample + parity over its 6 MSBs; the first 54 samples
	 * also signal the scale factor through the parity bit (:165-182) */
	for(x = 0; x < 64; x++)
	{
		int ch = x & 1;
		int v = (w[x] >> shift[rng[ch]]) & 0x3FF;
		v |= _even_parity6(v >> 4) << 10;
		if(x < 54) v ^= ((code[rng[ch]] >> (2 - (x / 2 % 3))) & 1) << 10;
		w[x] = v;
	}

	/* frame alignment word, control bits C0-C4 (C0 flips every 8 frames,
	 * mode = stereo, reserve flag = 1: src/video.c:4524), AD bits zero */
	memset(n->bits, 0, sizeof(n->bits));
	n->bits[0] = 0x4E;
	n->bits[1] = ((((~n->frame_no) >> 3) & 1) << 7) | ((n->reserve & 1) << 3);

	/* 704 sound bits, 11 per sample LSB first, interleaved 16 apart (:221-240) */
	for(x = 0, at = 0; x < 64; x++)
	{
		for(k = 0; k < 11; k++)
		{
			if((w[x] >> k) & 1) n->bits[3 + (at >> 3)] |= 0x80 >> (at & 7);
			at += 16;
			if(at >= 704) at -= 703;
		}
	}

	for(x = 0; x < 90; x++) n->bits[x + 1] ^= n->prn[x];

	n->frame_no++;
}

/* ---- sound-in-syncs (src/sis.c:155-221) ----
 *
 * Every line carries 23 or 25 four-level symbols of a NICAM-728 stream of its own inside its sync pulse. The framing
 * runs here, in step with the sound chains: one invocation per line of the line pipeline, the never-emitted slots in
 * front of line 1 included (they move the rate counter and the bit position on). A frame is encoded when the bit
 * position runs out, from the newest 32-sample block the audio process has handed over -- in the reference an unlocked
 * hand-over between two threads (src/video.c:3370-3373): the engine's reading is "the newest block completed in an
 * EARLIER step of the pipeline", i.e. before the audio line that runs beside this invocation. */
#define KEPT_LINES 8            /* `ahead` is at most 4 (SECAM's three slots + the resampler's line); counted in lines of the WIDEST width, that is up to five
                                * lines past a request's end where most lines are a sample narrower, and the line the request ended in */
typedef struct {
	int on;
	int dummies;            /* invocations in front of line 1: 1, or 3 behind a threaded colour process (hvk_tables.c) */
	int re, frame_bit;
	int64_t calls;          /* invocations so far */
	_nicam_t enc;           /* the encoder: J.17 history, frame counter, scrambler; enc.bits is the frame being sent */
	int16_t fill[64];       /* the block being filled by the 32 kHz ticks */
	int fill_len;
	int16_t newest[64];     /* the last block handed over */
	int have_block;
	/* the lines' bursts made so far: rec[(g - rec_g0) * 8] = the 7 burst bytes (MSB first) and the number of bits */
	uint8_t *rec;
	int64_t rec_g0;
	size_t rec_len, rec_cap;
} _sis_t;

/* ---- the path ---- */

struct hvk_audio {
	const hvk_tables_t *t;
	int oom;                /* an allocation failed: the symbol store is incomplete, every later request fails */
	int width;
	int sample_rate;

	/* 32 kHz stereo source queue */
	int16_t *src;
	size_t src_len, src_cap, src_pos;
	int64_t src_base;       /* position of src[0] in the 32 kHz source stream */
	int64_t generated;      /* stream samples the chains have worked through on THIS engine (not counting an imported start) */

	int interp;             /* 32 kHz tick accumulator (src/video.c:3273-3276) */
	int64_t pos;            /* stream samples generated so far (always a line boundary) */
	int64_t line_no;        /* lines generated so far */

	_phasor_t fm, am;
	_limiter_t lim;
	int has_lim;
	_phasor_t a2, a2_pilot, a2_signal;  /* A2 stereo: second FM carrier, pilot and identification tones */
	struct _pilot_ring *pil;            /* ... the last two on a thread of their own (then a2_pilot / a2_signal stand still until _pilot_settle()) */
	int pil_off;
	int thr_lines, thr_slow;            /* lines since the last look; waits for the tone thread in which this one had to step aside */
	int64_t pil_pos;                    /* pilot values taken from the ring so far */
	_limiter_t a2_lim;

	int nicam_on;
	_nicam_t nicam;

	/* symbols created so far: sym[i] belongs to symbol sym_k0 + i */
	uint8_t *sym;
	int64_t sym_k0;
	size_t sym_len, sym_cap;

	/* the carriers of the last lines generated, newest in slot `kept_at`: a request may start inside them, and with
	 * sound-in-syncs behind a threaded colour process (SECAM) the chains run three lines ahead of the requests, four behind
	 * the resampler. (Four kept lines were one too few exactly there -- SECAM, sound-in-syncs, a rate pair with lines of two
	 * widths: the first request was refused; found by tools/fuzz_parity.py, round 6) */
	int16_t *kept[KEPT_LINES];
	int64_t kept_pos[KEPT_LINES];
	int kept_w[KEPT_LINES];
	int kept_at;
	int ahead;              /* lines */
	int ahead_w;            /*   ... of this many samples each (behind the resampler the lines' widths vary: the widest) */

	_sis_t sis;
};

static void _sis_invocation(hvk_audio_t *a);

hvk_audio_t *hvk_audio_new(const hvk_tables_t *t)
{
	hvk_audio_t *a = calloc(1, sizeof(hvk_audio_t));
	if(!a) return(NULL);

	a->t = t;
	a->width = t->k.width;
	a->ahead_w = t->k.rs_L ? t->max_width : t->k.width;
	a->sample_rate = t->sample_rate;

	if(t->fm_lut)
	{
		a->fm.on = 1;
		a->fm.pi = INT32_MAX;
		a->fm.counter = INT16_MAX;
		a->fm.level = t->fm_level;

		if(t->has_limiter)
		{
			a->has_lim = 1;
			a->lim.pre.taps = t->limiter_vtaps;
			a->lim.flat.taps = t->limiter_ftaps;
			a->lim.shape = t->limiter_shape;
			a->lim.ceiling = INT16_MAX;
			a->lim.centre = LIM_SPAN / 2;
		}
	}

	if(t->a2_lut)
	{
		a->a2.on = 1;
		a->a2.pi = INT32_MAX;
		a->a2.counter = INT16_MAX;
		a->a2.level = t->a2_level;
		if(t->has_limiter)
		{
			a->a2_lim.pre.taps = t->limiter_vtaps;
			a->a2_lim.flat.taps = t->limiter_ftaps;
			a->a2_lim.shape = t->limiter_shape;
			a->a2_lim.ceiling = INT16_MAX;
			a->a2_lim.centre = LIM_SPAN / 2;
		}
		a->a2_pilot.pi = a->a2_signal.pi = INT32_MAX;
		a->a2_pilot.counter = a->a2_signal.counter = INT16_MAX;
		a->a2_pilot.level = t->a2_pilot_level;
		a->a2_signal.level = t->a2_signal_level;
	}

	if(t->am_level)
	{
		a->am.on = 1;
		a->am.pi = INT32_MAX;
		a->am.counter = INT16_MAX;
		a->am.level = t->am_level;
	}

	if(t->k.has_nicam)
	{
		a->nicam_on = 1;
		_nicam_reset(&a->nicam, t);
	}

	for(int i = 0; i < KEPT_LINES; i++)
	{
		a->kept[i] = malloc(sizeof(int16_t) * 2 * t->max_width);
		if(!a->kept[i]) { hvk_audio_free(a); return(NULL); }
	}

	if(t->k.sis)
	{
		a->sis.on = 1;
		a->sis.dummies = t->k.sis_dummies;
		_nicam_reset(&a->sis.enc, t);
		a->sis.enc.reserve = 0;
		/* invocation t needs the audio lines before t - 1 complete, line g is invocation g + 1 + dummies: its burst is known
		 * once audio line g + dummies - 1 is through. A frame's render also wants the burst of the line BEHIND it (the video
		 * filter looks into that line's first samples): the chains stay `dummies` lines ahead of the requests -- one more
		 * behind the resampler, whose output stands a raster line back: there the filter looks into the line after that */
		a->ahead = a->sis.dummies + (t->k.rs_L ? 1 : 0);
		_sis_invocation(a);                 /* the first one runs before any audio line has */
	}

	return(a);
}

static void _pilot_stop(hvk_audio_t *a);
static void _pilot_settle(hvk_audio_t *a);

void hvk_audio_free(hvk_audio_t *a)
{
	if(!a) return;
	_pilot_stop(a);
	free(a->src);
	free(a->sym);
	for(int i = 0; i < KEPT_LINES; i++) free(a->kept[i]);
	free(a->sis.rec);
	free(a);
}

int64_t hvk_audio_position(const hvk_audio_t *a) { return(a->pos); }

int hvk_audio_push(hvk_audio_t *a, const int16_t *stereo, size_t nsamples)
{
	if(a->src_pos > 0 && a->src_pos == a->src_len) { a->src_base += (int64_t) a->src_pos; a->src_pos = a->src_len = 0; }

	if(a->src_len + nsamples > a->src_cap)
	{
		/* compact, then grow */
		if(a->src_pos > 0)
		{
			memmove(a->src, a->src + a->src_pos * 2, (a->src_len - a->src_pos) * 2 * sizeof(int16_t));
			a->src_len -= a->src_pos;
			a->src_base += (int64_t) a->src_pos;
			a->src_pos = 0;
		}
		if(a->src_len + nsamples > a->src_cap)
		{
			size_t cap = (a->src_len + nsamples) * 2;
			int16_t *n = realloc(a->src, cap * 2 * sizeof(int16_t));
			if(!n) return(HVK_OUT_OF_MEMORY);
			a->src = n;
			a->src_cap = cap;
		}
	}

	memcpy(a->src + a->src_len * 2, stereo, nsamples * 2 * sizeof(int16_t));
	a->src_len += nsamples;
	return(HVK_OK);
}

/* 32 kHz source samples (beyond those queued) needed to reach stream
 * position upto_pos: ticks fire when the accumulator crosses sample_rate */
size_t hvk_audio_source_needed(const hvk_audio_t *a, int64_t upto_pos)
{
	int64_t n = upto_pos + (int64_t) a->ahead * a->ahead_w - a->pos;
	int64_t ticks, have;
	if(n <= 0) return(0);
	ticks = ((int64_t) a->interp + n * AUDIO_RATE) / a->sample_rate;
	have = (int64_t) (a->src_len - a->src_pos);
	return(ticks > have ? (size_t) (ticks - have) : 0);
}

/* One 32 kHz tick: fetch, scale, distribute (src/video.c:3278-3377) */
static void _tick(hvk_audio_t *a)
{
	int16_t s[2] = { 0, 0 };
	int i;

	if(a->src_pos < a->src_len)
	{
		for(i = 0; i < 2; i++)
		{
			int32_t v = ((int32_t) a->src[a->src_pos * 2 + i] * a->t->conf.volume + 128) >> 8;
			s[i] = v < INT16_MIN ? INT16_MIN : (v > INT16_MAX ? INT16_MAX : v);
		}
		a->src_pos++;
	}

	if(a->am.on) a->am.sample = (s[0] + s[1]) / 2;

	if(a->fm.on)
	{
		a->fm.sample = (s[0] + s[1]) / 2;
		if(a->has_lim) a->fm.sample = _limit(&a->lim, a->fm.sample);
		/* room for the pilot in A2 stereo mode (src/video.c:3325-3327): int16 *= double truncates */
		if(a->a2.on) a->fm.sample *= 0.95;
	}

	if(a->a2.on)
	{
		/* the right channel (src/video.c:3340-3350) */
		a->a2.sample = s[1];
		if(a->has_lim) a->a2.sample = _limit(&a->a2_lim, a->a2.sample);
		a->a2.sample *= 0.95;
	}

	if(a->nicam_on)
	{
		_nicam_t *n = &a->nicam;
		n->fill[n->fill_len++] = s[0];
		n->fill[n->fill_len++] = s[1];
		if(n->fill_len == 64)
		{
			memcpy(n->block, n->fill, sizeof(n->block));
			n->fill_len = 0;
		}
	}

	if(a->sis.on)
	{
		/* the same blocks go to the sound-in-syncs encoder (src/video.c:3353-3373) */
		_sis_t *q = &a->sis;
		q->fill[q->fill_len++] = s[0];
		q->fill[q->fill_len++] = s[1];
		if(q->fill_len == 64)
		{
			memcpy(q->newest, q->fill, sizeof(q->newest));
			q->have_block = 1;
			q->fill_len = 0;
		}
	}
}

/* One invocation of the sound-in-syncs process (src/sis.c:155-201): the burst's bits. Called once when the engine is
 * made and then after every audio line: invocation t sees the blocks handed over in audio lines 0 .. t - 2. */
static void _sis_invocation(hvk_audio_t *a)
{
	static const uint8_t gc[2][4] = { { 3, 0, 2, 1 }, { 0, 3, 1, 2 } };
	_sis_t *q = &a->sis;
	uint8_t vbi[8];
	int x, nb = 50;
	int64_t g;

	q->calls++;
	/* rate: 48 bits on most lines, 44 on 44 of every 125 (728 bits a millisecond) */
	if((q->re += 44) >= 125)
	{
		nb -= 4;
		q->re -= 125;
	}

	memset(vbi, 0, sizeof(vbi));
	vbi[0] = 0xC0;
	for(x = 2; x < nb; x += 2, q->frame_bit += 2)
	{
		uint8_t sym;
		if(q->frame_bit >= NICAM_FRAME_BITS)
		{
			if(q->have_block) memcpy(q->enc.block, q->newest, sizeof(q->enc.block));
			else memset(q->enc.block, 0, sizeof(q->enc.block));
			_nicam_build_frame(&q->enc);
			q->frame_bit = 0;
		}
		/* (before the first frame the store is all zeros, src/sis.c:88) */
		sym = (q->enc.bits[q->frame_bit >> 3] >> (6 - (q->frame_bit & 7))) & 3;
		sym = gc[(x & 4) ? 1 : 0][sym];
		vbi[x >> 3] |= sym << (6 - (x & 7));
	}
	vbi[7] = (uint8_t) nb;

	g = q->calls - 1 - q->dummies;
	if(g < 0) return;
	if(q->rec_len == 0) q->rec_g0 = g;
	if(q->rec_len + 8 > q->rec_cap)
	{
		size_t cap = q->rec_cap ? q->rec_cap * 2 : 65536;
		uint8_t *p = realloc(q->rec, cap);
		if(!p) { a->oom = 1; return; }
		q->rec = p;
		q->rec_cap = cap;
	}
	memcpy(q->rec + q->rec_len, vbi, 8);
	q->rec_len += 8;
}


static void _sym_append(hvk_audio_t *a, uint8_t v)
{
	if(a->sym_len == a->sym_cap)
	{
		size_t cap = a->sym_cap ? a->sym_cap * 2 : 65536;
		uint8_t *p = realloc(a->sym, cap);
		if(!p) { a->oom = 1; return; }       /* hvk_audio_generate() reports it */
		a->sym = p;
		a->sym_cap = cap;
	}
	a->sym[a->sym_len++] = v;
}

/* v[i] = (v[i] * level) >> 15 (src/video.c:2263-2264 after the phase's top half has been taken) */
static void _scale16(int16_t *v, int n, int32_t level)
{
	int i = 0;
#if defined(__SSE2__)
	if(level >= 0 && level <= INT16_MAX)
	{
		const __m128i lv = _mm_set1_epi16((short) level);
		for(; i + 8 <= n; i += 8)
		{
			const __m128i a = _mm_loadu_si128((const __m128i *) (v + i));
			const __m128i lo = _mm_mullo_epi16(a, lv), hi = _mm_mulhi_epi16(a, lv);
			const __m128i p0 = _mm_srai_epi32(_mm_unpacklo_epi16(lo, hi), 15);
			const __m128i p1 = _mm_srai_epi32(_mm_unpackhi_epi16(lo, hi), 15);
			_mm_storeu_si128((__m128i *) (v + i), _mm_packs_epi32(p0, p1));   /* |product >> 15| < 2^15: nothing saturates */
		}
	}
#endif
	for(; i < n; i++) v[i] = (int16_t) (((int32_t) v[i] * level) >> 15);
}

/* v[i] = (int16_t) ((v[i] * w[i]) >> 15), w[i] >= 0 */
static void _scale16v(int16_t *v, const int16_t *w, int n)
{
	int i = 0;
#if defined(__SSE2__)
	for(; i + 8 <= n; i += 8)
	{
		const __m128i a = _mm_loadu_si128((const __m128i *) (v + i)), b = _mm_loadu_si128((const __m128i *) (w + i));
		const __m128i lo = _mm_mullo_epi16(a, b), hi = _mm_mulhi_epi16(a, b);
		const __m128i p0 = _mm_srai_epi32(_mm_unpacklo_epi16(lo, hi), 15);
		const __m128i p1 = _mm_srai_epi32(_mm_unpackhi_epi16(lo, hi), 15);
		_mm_storeu_si128((__m128i *) (v + i), _mm_packs_epi32(p0, p1));   /* |product >> 15| <= 2^15 - 1: nothing saturates */
	}
#endif
	for(; i < n; i++) v[i] = (int16_t) (((int32_t) v[i] * w[i]) >> 15);
}

/* ---- Zweikanalton (A2 stereo) ---- */

#define A2_MUL(pi_, pq_, st_) \
	do { \
		const int64_t ni_ = (int64_t) (pi_) * (st_).i - (int64_t) (pq_) * (st_).q; \
		const int64_t nq_ = (int64_t) (pi_) * (st_).q + (int64_t) (pq_) * (st_).i; \
		(pi_) = (int32_t) (ni_ >> 31); \
		(pq_) = (int32_t) (nq_ >> 31); \
	} while(0)
/* amplitude drift correction every INT16_MAX steps of a phasor (src/video.c:2266-2275) */
#define A2_FIX(pi_, pq_, cnt_) \
	do { \
		if((cnt_) == 0) \
		{ \
			const double ra_ = atan2((pq_), (pi_)); \
			(pi_) = lround(cos(ra_) * INT32_MAX); \
			(pq_) = lround(sin(ra_) * INT32_MAX); \
			(cnt_) = INT16_MAX; \
		} \
	} while(0)

/* v[i] = ((uint16) v[i] ^ 0x8000) >> 1 = (v[i] - INT16_MIN) / 2 */
static void _half_offset16(int16_t *v, int n)
{
	int i = 0;
#if defined(__SSE2__)
	const __m128i top = _mm_set1_epi16((short) 0x8000);
	for(; i + 8 <= n; i += 8)
	{
		const __m128i a = _mm_loadu_si128((const __m128i *) (v + i));
		_mm_storeu_si128((__m128i *) (v + i), _mm_srli_epi16(_mm_xor_si128(a, top), 1));
	}
#endif
	for(; i < n; i++) v[i] = (int16_t) (((int32_t) v[i] - INT16_MIN) / 2);
}

/* v[i] += w[i] (+ c), int16 wrap-around */
static void _add16(int16_t *v, const int16_t *w, int16_t c, int n)
{
	int i = 0;
#if defined(__SSE2__)
	const __m128i cc = _mm_set1_epi16(c);
	for(; i + 8 <= n; i += 8)
	{
		__m128i a = _mm_add_epi16(_mm_loadu_si128((const __m128i *) (v + i)), cc);
		if(w) a = _mm_add_epi16(a, _mm_loadu_si128((const __m128i *) (w + i)));
		_mm_storeu_si128((__m128i *) (v + i), a);
	}
#endif
	for(; i < n; i++) v[i] = (int16_t) (v[i] + (w ? w[i] : 0) + c);
}

/* The identification tone and the pilot it modulates (src/video.c:3408-3416): two recurrences with CONSTANT steps that
 * nothing feeds -- their values are a function of the sample's number in the stream alone. n samples from the state
 * (si .. lc), pilot values (before the right channel is added) to out; tmp: n + 8 int16 of scratch. */
typedef struct { int32_t si, sq, li, lq, sc, lc; } _a2_tone_state_t;

static void _a2_pilot_run(const hvk_tables_t *t, int32_t sg_level, int32_t pl_level, _a2_tone_state_t *st, int16_t *out, int16_t *tmp, int n)
{
	const hvk_c32_t sg_step = t->a2_signal_delta, pl_step = t->a2_pilot_delta;
	int32_t si = st->si, sq = st->sq, li = st->li, lq = st->lq, sc = st->sc, lc = st->lc;
	int done, i;

	for(done = 0; done < n; )
	{
		int run = n - done;
		if(run > sc) run = sc;
		if(run > lc) run = lc;
		for(i = done; i < done + run; i++)
		{
			A2_MUL(si, sq, sg_step);
			A2_MUL(li, lq, pl_step);
			if(out)
			{
				tmp[i] = (int16_t) (si >> 16);
				out[i] = (int16_t) (li >> 16);
			}
		}
		done += run;
		sc -= run; lc -= run;
		A2_FIX(si, sq, sc);
		A2_FIX(li, lq, lc);
	}
	st->si = si; st->sq = sq; st->li = li; st->lq = lq; st->sc = sc; st->lc = lc;
	if(!out) return;

	/* tone = (((si >> 16) * 16384 >> 15) * level) >> 15; pilot = (((li >> 16) * ((tone - INT16_MIN) / 2) >> 15) * level) >> 15 */
	_scale16(tmp, n, 16384);
	_scale16(tmp, n, sg_level);
	_half_offset16(tmp, n);
	_scale16v(out, tmp, n);
	_scale16(out, n, pl_level);
}

/* A thread of their own for those two: it runs ahead of the sound chains through a ring of blocks and leaves the main
 * thread the two recurrences that do follow the sound (independent carrier chains side by side). Every
 * block keeps the state it began with, so the state at any sample the consumer stands at -- what
 * hvk_audio_state_export() hands on -- is a replay of less than a block away. */
#define PIL_BS 4096
#define PIL_NB 32
struct _pilot_ring {
	pthread_t th;
	pthread_mutex_t mx;
	pthread_cond_t cv;
	int stop;
	_Atomic int64_t produced, consumed;     /* samples since the thread's start: published / no longer needed (whole blocks) */
	int16_t ring[PIL_NB * PIL_BS];
	_a2_tone_state_t start[PIL_NB];         /* a block's state at its first sample */
	_a2_tone_state_t run;                   /* the producer's */
	const hvk_tables_t *t;
	int32_t sg_level, pl_level;
};

static void *_pilot_thread(void *arg)
{
	struct _pilot_ring *r = arg;
	int16_t tmp[PIL_BS + 8];

	for(;;)
	{
		const int64_t p = atomic_load_explicit(&r->produced, memory_order_relaxed);
		pthread_mutex_lock(&r->mx);
		while(!r->stop && p + PIL_BS - atomic_load_explicit(&r->consumed, memory_order_acquire) > (int64_t) PIL_NB * PIL_BS)
			pthread_cond_wait(&r->cv, &r->mx);
		if(r->stop) { pthread_mutex_unlock(&r->mx); break; }
		pthread_mutex_unlock(&r->mx);

		const int b = (int) ((p / PIL_BS) % PIL_NB);
		r->start[b] = r->run;
		_a2_pilot_run(r->t, r->sg_level, r->pl_level, &r->run, r->ring + (size_t) b * PIL_BS, tmp, PIL_BS);
		atomic_store_explicit(&r->produced, p + PIL_BS, memory_order_release);
	}
	return(NULL);
}

static void _pilot_stop(hvk_audio_t *a)
{
	struct _pilot_ring *r = a->pil;
	if(!r) return;
	pthread_mutex_lock(&r->mx);
	r->stop = 1;
	pthread_cond_broadcast(&r->cv);
	pthread_mutex_unlock(&r->mx);
	pthread_join(r->th, NULL);
	pthread_cond_destroy(&r->cv);
	pthread_mutex_destroy(&r->mx);
	free(r);
	a->pil = NULL;
}

/* Starts the thread at the state a->a2_signal / a->a2_pilot hold; false: no thread (HVK_AUDIO_THREADS=0, or none to be had) */
static int _pilot_start(hvk_audio_t *a)
{
	const char *ev = getenv("HVK_AUDIO_THREADS");
	struct _pilot_ring *r;

	if(a->pil) return(1);
	if(a->pil_off || (ev && atoi(ev) == 0)) { a->pil_off = 1; return(0); }
	r = calloc(1, sizeof(*r));
	if(!r) { a->pil_off = 1; return(0); }
	r->t = a->t;
	r->sg_level = a->a2_signal.level; r->pl_level = a->a2_pilot.level;
	r->run = (_a2_tone_state_t) { a->a2_signal.pi, a->a2_signal.pq, a->a2_pilot.pi, a->a2_pilot.pq, a->a2_signal.counter, a->a2_pilot.counter };
	pthread_mutex_init(&r->mx, NULL);
	pthread_cond_init(&r->cv, NULL);
	if(pthread_create(&r->th, NULL, _pilot_thread, r) != 0)
	{
		pthread_cond_destroy(&r->cv);
		pthread_mutex_destroy(&r->mx);
		free(r);
		a->pil_off = 1;
		return(0);
	}
	a->pil = r;
	a->pil_pos = 0;
	return(1);
}

/* (returns whether the wait was long enough for this thread to step aside) */
static int _pilot_wait(struct _pilot_ring *r, int64_t upto)
{
	int spins = 0, yielded = 0;
	while(atomic_load_explicit(&r->produced, memory_order_acquire) < upto)
	{
#if defined(__x86_64__) || defined(__i386__)
		if(++spins < 100) { __builtin_ia32_pause(); continue; }
#endif
		yielded = 1;
		/* (the producer may be asleep on a full ring only if this thread is behind its bookkeeping: wake it all the same) */
		pthread_mutex_lock(&r->mx);
		pthread_cond_signal(&r->cv);
		pthread_mutex_unlock(&r->mx);
		sched_yield();
	}
	return(yielded);
}

/* the next n (<= PIL_BS) pilot values */
static void _pilot_fetch(hvk_audio_t *a, int16_t *dst, int n)
{
	struct _pilot_ring *r = a->pil;
	const int64_t pos = a->pil_pos;
	const size_t at = (size_t) (pos % ((int64_t) PIL_NB * PIL_BS));
	const size_t first = (size_t) n < (size_t) PIL_NB * PIL_BS - at ? (size_t) n : (size_t) PIL_NB * PIL_BS - at;

	a->thr_slow += _pilot_wait(r, pos + n);
	memcpy(dst, r->ring + at, first * sizeof(int16_t));
	if(first < (size_t) n) memcpy(dst + first, r->ring, ((size_t) n - first) * sizeof(int16_t));
	a->pil_pos = pos + n;
	if(a->pil_pos / PIL_BS != pos / PIL_BS)
	{
		/* the blocks behind the one this thread stands in are free */
		atomic_store_explicit(&r->consumed, a->pil_pos / PIL_BS * PIL_BS, memory_order_release);
		pthread_mutex_lock(&r->mx);
		pthread_cond_signal(&r->cv);
		pthread_mutex_unlock(&r->mx);
	}
}

/* a->a2_signal / a->a2_pilot brought to where the consumer stands, the thread ended: before the state is read (export),
 * replaced (import) or the object goes */
static void _pilot_settle(hvk_audio_t *a)
{
	struct _pilot_ring *r = a->pil;
	_a2_tone_state_t st;
	if(!r) return;
	_pilot_wait(r, a->pil_pos / PIL_BS * PIL_BS + PIL_BS);       /* the block the consumer stands in has been begun and finished */
	st = r->start[(a->pil_pos / PIL_BS) % PIL_NB];
	_a2_pilot_run(r->t, r->sg_level, r->pl_level, &st, NULL, NULL, (int) (a->pil_pos % PIL_BS));
	a->a2_signal.pi = st.si; a->a2_signal.pq = st.sq; a->a2_signal.counter = st.sc;
	a->a2_pilot.pi = st.li; a->a2_pilot.pq = st.lq; a->a2_pilot.counter = st.lc;
	_pilot_stop(a);
}

/* Samples [x0, x1) of the line: four recurrences per sample -- the first carrier, the identification tone, the pilot it
 * modulates, and the second carrier whose step follows right channel + pilot (src/video.c:3402-3424). They depend on one
 * another only through VALUES, never through state, and everything that is not a recurrence -- the level scalings, the
 * pilot's modulation, the sums: eight of the reference's 24 multiplies per sample -- works on eight samples at a time.
 * Tone and pilot come from their own thread (or, without it, from the same function called here); the two carriers'
 * recurrences run side by side in this one (the second one's step is a table look-up on right channel + pilot). Same
 * operations on the same values in the same order per phasor as the general loop in _carriers(); the amplitude
 * corrections fall on the same samples (every phasor counts its own steps). */
#define A2_SPAN 512
static void _carriers_a2(hvk_audio_t *a, int16_t *carriers, int x0, int x1)
{
	const hvk_tables_t *t = a->t;
	const hvk_c32_t fm_step = t->fm_lut[a->fm.sample - INT16_MIN];
	const int16_t m0 = t->a2_system_m ? (int16_t) (a->fm.sample - a->a2.sample) : a->a2.sample;
	const int32_t fm_level = a->fm.level, a2_level = a->a2.level;
	int32_t fi = a->fm.pi, fq = a->fm.pq, ci = a->a2.pi, cq = a->a2.pq;
	int32_t fc = a->fm.counter, cc = a->a2.counter;
	int16_t tmp[A2_SPAN + 8], pil[A2_SPAN + 8], car2[2 * A2_SPAN + 16];
	const int threaded = _pilot_start(a);
	int x;

	for(x = x0; x < x1; )
	{
		const int n = x1 - x < A2_SPAN ? x1 - x : A2_SPAN;
		int16_t *o = carriers + (size_t) x * 2;
		int i, done;

		if(threaded) _pilot_fetch(a, pil, n);
		else
		{
			_a2_tone_state_t st = { a->a2_signal.pi, a->a2_signal.pq, a->a2_pilot.pi, a->a2_pilot.pq, a->a2_signal.counter, a->a2_pilot.counter };
			_a2_pilot_run(t, a->a2_signal.level, a->a2_pilot.level, &st, pil, tmp, n);
			a->a2_signal.pi = st.si; a->a2_signal.pq = st.sq; a->a2_signal.counter = st.sc;
			a->a2_pilot.pi = st.li; a->a2_pilot.pq = st.lq; a->a2_pilot.counter = st.lc;
		}
		/* the second carrier's modulating sample: right channel (L - R on system M) + pilot, int16 wrap-around (src/video.c:3417-3419) */
		_add16(pil, NULL, m0, n);

		for(done = 0; done < n; )
		{
			int run = n - done;
			if(run > fc) run = fc;
			if(run > cc) run = cc;
			for(i = done; i < done + run; i++)
			{
				const hvk_c32_t st = t->a2_lut[pil[i] - INT16_MIN];
				A2_MUL(fi, fq, fm_step);
				A2_MUL(ci, cq, st);
				o[i * 2 + 0] = (int16_t) (fi >> 16);
				o[i * 2 + 1] = (int16_t) (fq >> 16);
				car2[i * 2 + 0] = (int16_t) (ci >> 16);
				car2[i * 2 + 1] = (int16_t) (cq >> 16);
			}
			done += run;
			fc -= run; cc -= run;
			A2_FIX(fi, fq, fc);
			A2_FIX(ci, cq, cc);
		}

		_scale16(o, n * 2, fm_level);
		_scale16(car2, n * 2, a2_level);
		_add16(o, car2, 0, n * 2);
		x += n;
	}

	a->fm.pi = fi; a->fm.pq = fq; a->fm.counter = fc;
	a->a2.pi = ci; a->a2.pq = cq; a->a2.counter = cc;
}

/* Samples [x0, x1) of the current line with the modulating samples in force */
static void _carriers(hvk_audio_t *a, int16_t *carriers, int x0, int x1)
{
	int x;

	if(!a->fm.on && !a->am.on)
	{
		for(x = x0; x < x1; x++) carriers[x * 2 + 0] = carriers[x * 2 + 1] = 0;
		return;
	}

	if((a->fm.on && !a->a2.on && !a->am.on) || (a->am.on && !a->fm.on && !a->a2.on))
	{
		/* One carrier and nothing else (the common cases: every mono FM system; system L's AM sound): the recurrence
		 * with its state in registers, in runs that end where the amplitude correction is due. The chain is bound
		 * by the latency of its dependent multiply -> subtract -> shift; nothing else sits on that path. */
		_phasor_t *const ph = a->fm.on ? &a->fm : &a->am;
		const hvk_c32_t st = a->fm.on ? a->t->fm_lut[a->fm.sample - INT16_MIN] : a->t->am_delta;
		const int64_t ci = st.i, cq = st.q;
		const int32_t level = ph->level;
		/* AM: the carrier is scaled by the sound first (src/video.c:3386-3392) */
		const int32_t am_s = ((int32_t) a->am.sample - INT16_MIN) / 2;
		/* the phase as sign-extended 64-bit values: the reference's (int32_t) cast of the shifted product changes
		 * nothing while |phase| stays below 2^31 -- always, in practice: a step scales the amplitude by < 1 -- so the
		 * sign extension comes off the dependent chain (multiply -> subtract -> shift: 5 cycles instead of 6 - 7) and
		 * a never-taken branch keeps the cast for the case it would matter */
		int64_t pi = ph->pi, pq = ph->pq;
		int32_t counter = ph->counter;

		x = x0;
		while(x < x1)
		{
			int run = x1 - x, i;
			int16_t *o = carriers + (size_t) x * 2;
			if(run > counter) run = counter;
			for(i = 0; i < run; i++)
			{
				int64_t ni = (pi * ci - pq * cq) >> 31;
				int64_t nq = (pi * cq + pq * ci) >> 31;
				if(__builtin_expect((((uint64_t) ni + 0x80000000ULL) | ((uint64_t) nq + 0x80000000ULL)) >> 32 != 0, 0))
				{
					__asm__ volatile("" : "+r" (ni), "+r" (nq));      /* (keeps the compiler from folding the test into the cast) */
					ni = (int32_t) ni;
					nq = (int32_t) nq;
				}
				pi = ni;
				pq = nq;
				/* the top halves now, their scaling below, eight at a time: the recurrence's four multiplies are
				 * all the one multiplier of a core should see per sample */
				o[i * 2 + 0] = (int16_t) ((int32_t) pi >> 16);
				o[i * 2 + 1] = (int16_t) ((int32_t) pq >> 16);
			}
			if(a->am.on) _scale16(o, run * 2, am_s);
			_scale16(o, run * 2, level);
			x += run;
			counter -= run;
			if(counter == 0)
			{
				/* amplitude drift correction every INT16_MAX steps (src/video.c:2266-2275) */
				const double ra = atan2((int32_t) pq, (int32_t) pi);
				pi = lround(cos(ra) * INT32_MAX);
				pq = lround(sin(ra) * INT32_MAX);
				counter = INT16_MAX;
			}
		}
		ph->pi = (int32_t) pi;
		ph->pq = (int32_t) pq;
		ph->counter = counter;
		return;
	}

	if(a->fm.on && a->a2.on && !a->am.on)
	{
		_carriers_a2(a, carriers, x0, x1);
		return;
	}

	{
		/* the step is constant between two 32 kHz ticks */
		const hvk_c32_t fm_step = a->fm.on ? a->t->fm_lut[a->fm.sample - INT16_MIN] : (hvk_c32_t) { 0, 0 };
		const hvk_c32_t am_step = a->t->am_delta;
		const int32_t am_s = ((int32_t) a->am.sample - INT16_MIN) / 2;

		for(x = x0; x < x1; x++)
		{
			int16_t ai = 0, aq = 0;

			if(a->fm.on)
			{
				_step(&a->fm, fm_step.i, fm_step.q);
				ai += (int16_t) (((a->fm.pi >> 16) * a->fm.level) >> 15);
				aq += (int16_t) (((a->fm.pq >> 16) * a->fm.level) >> 15);
				_correct(&a->fm);
			}

			if(a->a2.on)
			{
				/* second carrier (src/video.c:3402-3424): right channel (L - R on system M) plus the
				 * pilot, itself amplitude modulated by the identification tone -- its step
				 * changes with every sample */
				const hvk_tables_t *t = a->t;
				int16_t m = t->a2_system_m ? (int16_t) (a->fm.sample - a->a2.sample) : a->a2.sample;
				int32_t tone, pilot;
				hvk_c32_t st;

				_step(&a->a2_signal, t->a2_signal_delta.i, t->a2_signal_delta.q);
				tone = (int16_t) (((((a->a2_signal.pi >> 16) * 16384) >> 15) * a->a2_signal.level) >> 15);
				_correct(&a->a2_signal);

				_step(&a->a2_pilot, t->a2_pilot_delta.i, t->a2_pilot_delta.q);
				pilot = (int16_t) (((((a->a2_pilot.pi >> 16) * ((tone - INT16_MIN) / 2)) >> 15) * a->a2_pilot.level) >> 15);
				_correct(&a->a2_pilot);

				m = (int16_t) (m + pilot);
				st = t->a2_lut[m - INT16_MIN];
				_step(&a->a2, st.i, st.q);
				ai += (int16_t) (((a->a2.pi >> 16) * a->a2.level) >> 15);
				aq += (int16_t) (((a->a2.pq >> 16) * a->a2.level) >> 15);
				_correct(&a->a2);
			}

			if(a->am.on)
			{
				_step(&a->am, am_step.i, am_step.q);
				ai += (int16_t) (((((a->am.pi >> 16) * am_s) >> 15) * a->am.level) >> 15);
				aq += (int16_t) (((((a->am.pq >> 16) * am_s) >> 15) * a->am.level) >> 15);
				_correct(&a->am);
			}

			carriers[x * 2 + 0] = ai;
			carriers[x * 2 + 1] = aq;
		}
	}
}

/* Width of line g of the audio process's stream. The process runs line by line on what
 * the stage before it hands over: raster lines, or -- with the resampler -- its chunks,
 * the outputs made from raster line g: [ceil(g W L / D), ceil((g + 1) W L / D)). */
static int _line_width(const hvk_audio_t *a, int64_t g)
{
	const hvk_kconst_t *k = &a->t->k;
	if(k->rs_L == 0) return(k->width);
	return((int) ((((g + 1) * k->width * k->rs_L + k->rs_D - 1) / k->rs_D) - ((g * k->width * k->rs_L + k->rs_D - 1) / k->rs_D)));
}

/* Advance the stream by one line of W samples; carriers (W int16 pairs) receives the
 * summed contribution of the serial carriers. */
static void _line(hvk_audio_t *a, int16_t *carriers, const int W)
{
	const int sr = a->sample_rate;
	int x = 0;

	while(x < W)
	{
		/* the accumulator next crosses sample_rate on the to_tick-th sample from here */
		int to_tick = (sr - a->interp + AUDIO_RATE - 1) / AUDIO_RATE;
		int quiet = to_tick - 1;

		if(quiet > W - x) quiet = W - x;

		_carriers(a, carriers, x, x + quiet);
		a->interp += quiet * AUDIO_RATE;
		x += quiet;

		if(quiet == to_tick - 1 && x < W)
		{
			/* new audio first, then this sample is modulated with it */
			a->interp += AUDIO_RATE - sr;
			_tick(a);
			_carriers(a, carriers, x, x + 1);
			x++;
		}
	}
	if(a->pil && ++a->thr_lines == 512)
	{
		/* A2: where the tone thread keeps this one waiting again and again it does not run BESIDE it -- too few cores to
		 * go round -- and costs more than it gives: back to one thread (HVK_AUDIO_THREADS=2: never) */
		if(a->thr_slow > 128 && !(getenv("HVK_AUDIO_THREADS") && atoi(getenv("HVK_AUDIO_THREADS")) > 1))
		{
			_pilot_settle(a);
			a->pil_off = 1;
		}
		a->thr_lines = a->thr_slow = 0;
	}

	/* NICAM: every symbol that starts inside this line is created now, after
	 * the whole line's ticks (src/video.c:3435-3438, src/nicam728.c:368-408) */
	if(a->nicam_on)
	{
		static const uint8_t advance[4] = { 0, 3, 1, 2 };
		_nicam_t *n = &a->nicam;
		int64_t line_end = a->pos + W;

		while(n->next_start < line_end)
		{
			int len;

			if(n->bit == NICAM_FRAME_BITS)
			{
				_nicam_build_frame(n);
				n->bit = 0;
			}

			n->dsym = (n->dsym + advance[(n->bits[n->bit >> 3] >> (6 - (n->bit & 7))) & 3]) & 3;
			n->bit += 2;
			_sym_append(a, n->dsym);

			len = n->sps;
			n->ds += n->dsl;
			if(n->ds >= n->decimation) { len--; n->ds -= n->decimation; }

			n->next_start += len;
			n->k++;
		}
	}

	a->pos += W;
	a->generated += W;
	a->line_no++;
	if(a->sis.on) _sis_invocation(a);
}

/* Index of the symbol whose pulse starts at or before sample m (>= 0) */
static int64_t _symbol_at(const hvk_audio_t *a, int64_t m)
{
	const _nicam_t *n = &a->nicam;
	int64_t period = (int64_t) n->sps * n->decimation - n->dsl; /* samples per `decimation` symbols */
	int64_t k = m * n->decimation / period;

	/* start of symbol k: sps * k - floor(k * dsl / decimation) */
	while(n->sps * (k + 1) - ((k + 1) * n->dsl) / n->decimation <= m) k++;
	while(k > 0 && n->sps * k - (k * n->dsl) / n->decimation > m) k--;
	return(k);
}

int hvk_audio_generate(hvk_audio_t *a, int64_t first, int64_t count,
                       int16_t *carriers, uint8_t *symbols, int max_symbols, int64_t *k0)
{
	int64_t end = first + count;
	int nsym = 0;

	/* the chains cannot be rewound: a request may start inside the lines kept, not before them */
	{
		int64_t oldest = a->pos;
		for(int i = 0; i < KEPT_LINES; i++) if(a->kept_w[i] && a->kept_pos[i] < oldest) oldest = a->kept_pos[i];
		if(first < oldest) return(HVK_ERROR);
	}
	if(a->oom) return(HVK_OUT_OF_MEMORY);

	/* the part of the request the kept lines cover */
	if(carriers)
	{
		for(int i = 0; i < KEPT_LINES; i++)
		{
			const int64_t lo = first > a->kept_pos[i] ? first : a->kept_pos[i];
			const int64_t hi = end < a->kept_pos[i] + a->kept_w[i] ? end : a->kept_pos[i] + a->kept_w[i];
			if(a->kept_w[i] && hi > lo) memcpy(carriers + (lo - first) * 2, a->kept[i] + (lo - a->kept_pos[i]) * 2, (hi - lo) * 2 * sizeof(int16_t));
		}
	}

	/* ... and new lines up to its end (and `ahead` lines past it) */
	while(a->pos < end + (int64_t) a->ahead * a->ahead_w)
	{
		const int i = a->kept_at = (a->kept_at + 1) % KEPT_LINES;
		a->kept_pos[i] = a->pos;
		a->kept_w[i] = _line_width(a, a->line_no);
		_line(a, a->kept[i], a->kept_w[i]);
		if(carriers)
		{
			const int64_t lo = first > a->kept_pos[i] ? first : a->kept_pos[i];
			const int64_t hi = end < a->kept_pos[i] + a->kept_w[i] ? end : a->kept_pos[i] + a->kept_w[i];
			if(hi > lo) memcpy(carriers + (lo - first) * 2, a->kept[i] + (lo - a->kept_pos[i]) * 2, (hi - lo) * 2 * sizeof(int16_t));
		}
	}

	if(a->nicam_on && symbols)
	{
		int64_t klo = _symbol_at(a, first) - 7;
		int64_t khi = _symbol_at(a, end - 1);
		int64_t k;

		if(khi - klo + 1 > max_symbols) return(HVK_ERROR);
		if(klo >= 0 && klo < a->sym_k0) return(HVK_ERROR); /* history already dropped */

		for(k = klo; k <= khi; k++)
		{
			symbols[nsym++] = (k < 0) ? 0xFF : a->sym[k - a->sym_k0];
		}
		if(k0) *k0 = klo;

		/* drop symbols no later request can need */
		if(khi - SYM_HISTORY > a->sym_k0)
		{
			size_t drop = (size_t) (khi - SYM_HISTORY - a->sym_k0);
			memmove(a->sym, a->sym + drop, a->sym_len - drop);
			a->sym_len -= drop;
			a->sym_k0 += drop;
		}
	}
	else if(k0) *k0 = 0;

	if(a->oom) return(HVK_OUT_OF_MEMORY);
	return(nsym);
}

/* Run the chains on to stream position `end` (and the lines they keep ahead of it) without handing anything out */
int hvk_audio_advance(hvk_audio_t *a, int64_t end)
{
	while(a->pos < end + (int64_t) a->ahead * a->ahead_w)
	{
		const int i = a->kept_at = (a->kept_at + 1) % KEPT_LINES;
		a->kept_pos[i] = a->pos;
		a->kept_w[i] = _line_width(a, a->line_no);
		_line(a, a->kept[i], a->kept_w[i]);
	}
	return(a->oom ? HVK_OUT_OF_MEMORY : HVK_OK);
}

/* The sound-in-syncs bursts of lines [g_first, g_first + count) of the stream (8 bytes a line: 7 bytes of bits, MSB first,
 * then their number); the sound chains have to have been run over the lines' frames. Earlier lines are dropped. */
int hvk_audio_sis_fetch(hvk_audio_t *a, int64_t g_first, int count, uint8_t *out)
{
	_sis_t *q = &a->sis;
	if(!q->on || count < 0) return(HVK_ERROR);
	if(a->oom) return(HVK_OUT_OF_MEMORY);
	if(g_first < q->rec_g0 || (size_t) (g_first - q->rec_g0 + count) * 8 > q->rec_len) return(HVK_ERROR);
	memcpy(out, q->rec + (size_t) (g_first - q->rec_g0) * 8, (size_t) count * 8);
	if(g_first > q->rec_g0)
	{
		const size_t drop = (size_t) (g_first - q->rec_g0) * 8;
		memmove(q->rec, q->rec + drop, q->rec_len - drop);
		q->rec_len -= drop;
		q->rec_g0 = g_first;
	}
	return(HVK_OK);
}

/* ---- the chains' state, to be carried to another engine ----
 *
 * The sound chains are one recurrence over the whole stream: an engine that renders frames f .. can only
 * start where the engine that rendered up to f - 1 stopped. This is everything it stopped with -- the phasors (phase,
 * steps to the next amplitude correction, sample in force), the 32 kHz tick accumulator, both limiters (filter
 * histories, look-ahead ring), the NICAM framer (J.17 history, the block being filled, the frame being sent and the bit
 * it stands at, the differential phase, the symbol schedule) and the last symbols, whose pulses reach into the next
 * samples -- as one flat block; pointers inside it are set again by the importer. Both ends are the same build. */
#define STATE_MAGIC  0x48564B41u    /* "HVKA" */
#define STATE_SYMS   256
typedef struct {
	uint32_t magic, bytes;
	int32_t width, sample_rate;
	int32_t interp;
	int32_t has_lim, nicam_on;
	int64_t pos, line_no;
	int64_t source_pos;         /* 32 kHz source samples consumed so far */
	_phasor_t fm, am, a2, a2_pilot, a2_signal;
	_limiter_t lim, a2_lim;
	_nicam_t nicam;
	_sis_t sis;                 /* (its record store stays behind: the importer makes its own lines' bursts) */
	int64_t sym_first;          /* symbol index of sym[0] */
	int32_t nsym;
	uint8_t sym[STATE_SYMS];
} _audio_state_t;

size_t hvk_audio_state_bytes(void) { return(sizeof(_audio_state_t)); }

int64_t hvk_audio_generated(const hvk_audio_t *a) { return(a ? a->generated : 0); }

/* the position in the 32 kHz source behind the last pair the queue holds: the next hvk_audio_push() continues there */
int64_t hvk_audio_source_end(const hvk_audio_t *a) { return(a ? a->src_base + (int64_t) a->src_len : 0); }

int hvk_audio_state_export(hvk_audio_t *a, void *buf, size_t bytes)
{
	_audio_state_t *st = buf;
	size_t n;

	if(!a || !buf || bytes < sizeof(*st)) return(HVK_ERROR);
	if(a->ahead) return(HVK_UNSUPPORTED);    /* the chains stand past the last request (SECAM with sound-in-syncs: one engine renders such a stream anyway) */
	_pilot_settle(a);       /* (A2: the tone thread's state brought to where the chains stand; it starts again with the next line) */
	memset(st, 0, sizeof(*st));
	st->magic = STATE_MAGIC;
	st->bytes = (uint32_t) sizeof(*st);
	st->width = a->width;
	st->sample_rate = a->sample_rate;
	st->interp = a->interp;
	st->has_lim = a->has_lim;
	st->nicam_on = a->nicam_on;
	st->pos = a->pos;
	st->line_no = a->line_no;
	st->source_pos = a->src_base + (int64_t) a->src_pos;
	st->fm = a->fm; st->am = a->am; st->a2 = a->a2; st->a2_pilot = a->a2_pilot; st->a2_signal = a->a2_signal;
	st->lim = a->lim;
	st->a2_lim = a->a2_lim;
	st->nicam = a->nicam;
	st->sis = a->sis;
	st->sis.rec = NULL; st->sis.rec_len = st->sis.rec_cap = 0;
	n = a->sym_len < STATE_SYMS ? a->sym_len : STATE_SYMS;
	st->nsym = (int32_t) n;
	st->sym_first = a->sym_k0 + (int64_t) (a->sym_len - n);
	if(n) memcpy(st->sym, a->sym + (a->sym_len - n), n);
	return(HVK_OK);
}

/* *source_pos receives the position in the 32 kHz source stream the chains go on from. A queue that holds it keeps its
 * later samples; otherwise the queue is emptied and the NEXT hvk_audio_push() is taken to start there. */
int hvk_audio_state_import(hvk_audio_t *a, const void *buf, size_t bytes, int64_t *source_pos)
{
	const _audio_state_t *st = buf;

	if(!a || !buf || bytes < sizeof(*st) || st->magic != STATE_MAGIC || st->bytes != sizeof(*st)) return(HVK_ERROR);
	if(st->width != a->width || st->sample_rate != a->sample_rate || st->has_lim != a->has_lim || st->nicam_on != a->nicam_on) return(HVK_ERROR);
	if(st->fm.on != a->fm.on || st->am.on != a->am.on || st->a2.on != a->a2.on || st->sis.on != a->sis.on) return(HVK_ERROR);   /* another configuration's state */
	if(a->ahead) return(HVK_UNSUPPORTED);
	if(st->nsym < 0 || st->nsym > STATE_SYMS) return(HVK_ERROR);
	_pilot_settle(a);       /* (A2: the tone thread ends; the next line starts it again from the imported state) */

	a->interp = st->interp;
	a->pos = st->pos;
	a->line_no = st->line_no;
	for(int i = 0; i < KEPT_LINES; i++) a->kept_w[i] = 0;
	a->fm = st->fm; a->am = st->am; a->a2 = st->a2; a->a2_pilot = st->a2_pilot; a->a2_signal = st->a2_signal;
	{
		/* the tables the filters and the look-ahead window point at are this engine's own */
		const int32_t *vt = a->lim.pre.taps, *ft = a->lim.flat.taps, *vt2 = a->a2_lim.pre.taps, *ft2 = a->a2_lim.flat.taps;
		const int16_t *sh = a->lim.shape, *sh2 = a->a2_lim.shape;
		a->lim = st->lim;
		a->a2_lim = st->a2_lim;
		a->lim.pre.taps = vt; a->lim.flat.taps = ft; a->lim.shape = sh;
		a->a2_lim.pre.taps = vt2; a->a2_lim.flat.taps = ft2; a->a2_lim.shape = sh2;
	}
	a->nicam = st->nicam;
	{
		uint8_t *rec = a->sis.rec;
		const size_t cap = a->sis.rec_cap;
		a->sis = st->sis;
		a->sis.rec = rec; a->sis.rec_cap = cap; a->sis.rec_len = 0;
	}

	a->sym_len = 0;
	a->sym_k0 = st->sym_first;
	for(int i = 0; i < st->nsym; i++) _sym_append(a, st->sym[i]);
	if(a->oom) return(HVK_OUT_OF_MEMORY);

	if(st->source_pos >= a->src_base + (int64_t) a->src_pos && st->source_pos <= a->src_base + (int64_t) a->src_len)
	{
		a->src_pos = (size_t) (st->source_pos - a->src_base);
	}
	else
	{
		a->src_pos = a->src_len = 0;
		a->src_base = st->source_pos;
	}
	if(source_pos) *source_pos = st->source_pos;
	return(HVK_OK);
}

/* The newest NICAM symbol whose pulse has started by stream sample m, and
 * its first sample (closed form of the schedule in src/nicam728.c:398-407) */
int hvk_audio_symbol_info(const hvk_audio_t *a, int64_t m, int64_t *k, int64_t *start)
{
	int64_t kk;
	if(!a->nicam_on) return(HVK_ERROR);
	kk = _symbol_at(a, m);
	if(k) *k = kk;
	if(start) *start = (int64_t) a->nicam.sps * kk - (kk * a->nicam.dsl) / a->nicam.decimation;
	return(HVK_OK);
}
