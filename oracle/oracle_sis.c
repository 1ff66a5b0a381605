/* oracle/oracle_sis.c -- TEST INFRASTRUCTURE (not product code).
 *
 * Sound-in-syncs, --sis dcsis (src/sis.c:36-221; registered at src/video.c:4330-4338 behind CC608 and in front of
 * teletext; its audio hook src/video.c:3353-3373): on EVERY line the sync pulse's place is blanked to the sync level
 * through a window and carries 23 or 25 four-level symbols ("quits") of a NICAM-728 stream of its own.
 *
 * One invocation per step of the line pipeline, like every non-threaded process, on the slot the raster has just
 * finished (raster's lines[0]: the line before the one being built): invocation t works on line t - 1, the first one
 * on the never-emitted slot with line 0 -- which still moves the rate counter and the bit position on. (SECAM: two
 * slots further back, three never-emitted slots.)
 *
 * The block hand-over. The audio thread writes sis->audio (32 stereo samples after the volume control) whenever its
 * 32 kHz tick has filled a block (src/video.c:3353-3373); the SiS process, on the main thread, reads it when its bit
 * position runs into the next NICAM frame (src/sis.c:190-195) -- no lock, no barrier between the two inside a step.
 * Both things happen 1000 times a second exactly, so they meet in the SAME step for ever, and which block the frame
 * gets depends on whether the audio thread has passed its tick when the main thread arrives. `visible`: how many
 * samples of the step's audio line the audio thread has behind it at that moment (orc_set_sis_visible(); 0, the
 * default and the product's reading: none -- hand-overs of earlier steps only). The hand-overs fall on a few fixed
 * positions of a line, so this one number orders all of them. On its test source the reference CLI does what 0 .. 639
 * say (tests/test_oracle_vs_ref.py); the tone's blocks are all alike, which is why its output does not change from
 * run to run.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

#define AUDIO_RATE 32000

static double _rc(double x)
{
	if(x <= -1 || x >= 1) return(0);
	return((1.0 + cos(M_PI * x)) / 2);
}

int orc_sis_init(orc_t *s)
{
	const int W = s->width;
	const double bwidth = (double) W / 382, offset = (double) W / 382 * 3.32;
	int levels[2], b, x;
	double left = 0.2e-6, rise = 80e-9, width = 4.56e-6;

	if(!s->conf.sis) return(0);

	/* the 50 half symbols: entry b shapes bit b of the burst, bits 2 n and 2 n + 1 share a place and weigh 2 : 1
	 * (src/sis.c:36-75; the table's bookkeeping is vbidata_update(), src/vbidata.c:36-60: from the first
	 * non-zero value to the last) */
	{
		const int level = (int) round((double) (s->white_level - s->black_level));
		levels[0] = level / 2 / 0.75;
		levels[1] = level / 4 / 0.75;
	}
	s->sis_lut = calloc(50, sizeof(orc_pulse_t));
	for(b = 0; b < 50; b++)
	{
		orc_pulse_t *p = &s->sis_lut[b];
		const double t = -bwidth * (b / 2) - offset;
		p->value = calloc(W, sizeof(int16_t));
		p->offset = p->length = 0;
		for(x = 0; x < W; x++)
		{
			const int v = (int) round(_rc((t + x) / bwidth) * levels[b & 1]);
			if(v == 0) continue;
			if(p->length == 0) p->offset = x;
			while(p->length < x - p->offset) p->value[p->length++] = 0;
			p->value[p->length++] = (int16_t) v;
		}
	}

	/* the table as the reference packs it: [length][offset][values ...] per entry, -1 at the end (src/vbidata.h) */
	{
		long n = 0;
		for(b = 0; b < 50; b++) n += 2 + s->sis_lut[b].length;
		s->sis_packed = calloc(n + 1, sizeof(int16_t));
		n = 0;
		for(b = 0; b < 50; b++)
		{
			s->sis_packed[n++] = (int16_t) s->sis_lut[b].length;
			s->sis_packed[n++] = (int16_t) s->sis_lut[b].offset;
			s->sis_pos[b] = n;
			memcpy(s->sis_packed + n, s->sis_lut[b].value, s->sis_lut[b].length * sizeof(int16_t));
			n += s->sis_lut[b].length;
		}
		s->sis_packed[n] = -1;
		/* what glibc 2.35 has in front of the allocation in the reference CLI: the previous chunk's last 8 bytes and
		 * the chunk's size word ((bytes + 8 rounded up to 16) | PREV_INUSE) */
		memset(s->sis_heap, 0, sizeof(s->sis_heap));
		{
			const unsigned long bytes = (unsigned long) (n + 1) * 2, chunk = ((bytes + 8 + 15) & ~15UL) | 1;
			s->sis_heap[4] = (int16_t) (chunk & 0xFFFF);
			s->sis_heap[5] = (int16_t) ((chunk >> 16) & 0xFFFF);
		}
	}

	/* the blanking window (src/sis.c:118-138) */
	s->sis_blank_left = (int) floor(s->pixel_rate * (left - rise / 2));
	s->sis_blank_width = (int) ceil(s->pixel_rate * (width + rise));
	s->sis_blank_win = malloc(s->sis_blank_width * sizeof(int16_t));
	for(x = s->sis_blank_left; x < s->sis_blank_left + s->sis_blank_width; x++)
	{
		const double t = 1.0 / s->pixel_rate * x;
		s->sis_blank_win[x - s->sis_blank_left] = (int16_t) round(orc_rc_window(t, left, width, rise) * INT16_MAX);
	}

	/* its own NICAM encoder: stereo, reserve bit 0 (src/video.c:4332) */
	orc_nicam_encoder_init(&s->sis_nicam, 0x00, 0);
	s->sis_frame_bit = 0;
	s->sis_re = 0;
	s->sis_calls = 0;
	memset(s->sis_frame, 0, sizeof(s->sis_frame));
	return(0);
}

long orc_sis_bursts(orc_t *s, long first_line, long nlines, uint8_t *out)
{
	if(first_line < 0 || first_line + nlines > s->sis_rec_n) return(-1);
	memcpy(out, s->sis_rec + first_line * 8, (size_t) nlines * 8);
	return(nlines);
}

void orc_set_sis_visible(orc_t *s, int samples) { s->sis_visible = samples < 0 ? 0 : samples; }

/* the 8 samples in front of the reference's symbol table as THIS process's heap has them (oracle/ref_probe.c, table
 * "sis_heap"): the default above is what the reference CLI's heap holds there */
void orc_set_sis_heap(orc_t *s, const int16_t *h8) { memcpy(s->sis_heap, h8, sizeof(s->sis_heap)); }

void orc_sis_free(orc_t *s)
{
	int b;
	if(s->sis_lut) for(b = 0; b < 50; b++) free(s->sis_lut[b].value);
	free(s->sis_lut);
	free(s->sis_blank_win);
	free(s->sis_packed);
	free(s->sis_rec);
	s->sis_rec = NULL;
	s->sis_packed = NULL;
	s->sis_lut = NULL;
	s->sis_blank_win = NULL;
}

/* The block the audio thread has handed over by the time invocation `call` (counted from 1) encodes a frame: block j
 * is complete with tick 32 (j + 1), which fires on stream sample ceil(32 (j + 1) * rate / 32000) - 1
 * (src/video.c:3273-3276); the audio thread has `call - 1` whole lines behind it and `visible` samples of this step's.
 * Returns 0 and zeros when nothing has been handed over yet. */
static int _block_seen(orc_t *s, long call, int16_t out[64])
{
	const int visible = s->sis_visible;
	/* samples [0, reach) are behind the audio thread: call - 1 of its lines -- behind the resampler those are the resampler's
	 * chunks, chunk g beginning at ceil(g width L / D) (src/video.c:3627-3651) */
	const long long reach = (s->rs_taps ? ((long long) (call - 1) * s->width * s->rs_L + s->rs_D - 1) / s->rs_D : (long long) (call - 1) * s->width) + visible;
	long long j, n;
	int i;

	memset(out, 0, 64 * sizeof(int16_t));
	/* newest j with ceil(32 (j + 1) SR / 32000) - 1 < reach */
	j = (reach * AUDIO_RATE / s->sample_rate) / 32 - 1;
	while(j >= 0 && (32 * (j + 1) * (long long) s->sample_rate + AUDIO_RATE - 1) / AUDIO_RATE - 1 >= reach) j--;
	while((32 * (j + 2) * (long long) s->sample_rate + AUDIO_RATE - 1) / AUDIO_RATE - 1 < reach) j++;
	if(j < 0) return(0);

	/* its samples: source pairs 32 j .. 32 j + 31 through the volume control (src/video.c:3290-3297); the test source
	 * starts over when it has run dry (src/av_test.c:54-60) */
	for(i = 0; i < 32; i++)
	{
		n = 32 * j + i;
		if(!s->audio_src || s->audio_len <= 0) continue;
		if(n >= s->audio_len)
		{
			if(!s->audio_loop) continue;
			n %= s->audio_len;
		}
		for(int ch = 0; ch < 2; ch++)
		{
			int32_t v = ((int32_t) s->audio_src[n * 2 + ch] * s->conf.volume + 128) >> 8;
			out[i * 2 + ch] = (int16_t) (v < INT16_MIN ? INT16_MIN : (v > INT16_MAX ? INT16_MAX : v));
		}
	}
	return(1);
}

/* One invocation (src/sis.c:155-215). g: the line it works on, -1 for a never-emitted slot in front of line 1 -- one
 * such slot, or three where the colour process is a thread of its own between the raster and this process (SECAM:
 * src/video.c:4211, :3543-3583: a process behind a threaded one gets a slot of its own). first_line: the slot behind
 * this one is line 1's. */
void orc_sis_line(orc_t *s, long g, int first_line)
{
	static const uint8_t gc[2][4] = { { 3, 0, 2, 1 }, { 0, 3, 1, 2 } };
	uint8_t vbi[7];
	int x, nb = 50, b, i;
	int16_t *o;

	if(!s->conf.sis) return;
	s->sis_calls++;

	/* rate: 48 bits on most lines, 44 on 44 of every 125 */
	if((s->sis_re += 44) >= 125)
	{
		nb -= 4;
		s->sis_re -= 125;
	}

	memset(vbi, 0, sizeof(vbi));
	vbi[0] = 0xC0;
	for(x = 2; x < nb; x += 2, s->sis_frame_bit += 2)
	{
		uint8_t sym;
		if(s->sis_frame_bit >= 728)
		{
			_block_seen(s, s->sis_calls, s->sis_nicam.audio);
			orc_nicam_encode(&s->sis_nicam);
			memcpy(s->sis_frame, s->sis_nicam.frame, 91);
			s->sis_frame_bit = 0;
		}
		sym = (s->sis_frame[s->sis_frame_bit >> 3] >> (6 - (s->sis_frame_bit & 7))) & 3;
		sym = gc[(x & 4) ? 1 : 0][sym];
		vbi[x >> 3] |= sym << (6 - (x & 7));
	}

	if(g >= 0)
	{
		/* kept for tests: what the host half of the product has to come up with (orc_sis_bursts()) */
		if(g >= s->sis_rec_cap)
		{
			s->sis_rec_cap = s->sis_rec_cap ? s->sis_rec_cap * 2 : 4096;
			while(g >= s->sis_rec_cap) s->sis_rec_cap *= 2;
			s->sis_rec = realloc(s->sis_rec, (size_t) s->sis_rec_cap * 8);
		}
		memcpy(s->sis_rec + g * 8, vbi, 7);
		s->sis_rec[g * 8 + 7] = (uint8_t) nb;
		if(g + 1 > s->sis_rec_n) s->sis_rec_n = g + 1;
	}

	if(g < 0 && !first_line) return;        /* a slot without width whose successor has none either: nothing is drawn (src/vbidata.c:219-225) */
	if(g < 0)
	{
		/* This slot has no width, the next one is line 1's: vbidata_render() moves on to it and
		 * starts every set symbol at that line's sample 0 with a NEGATIVE index into the symbol's values
		 * (src/vbidata.c:211-217: x = -lx): the symbol lands where it belongs, and the samples in front of it get what lies
		 * in front of its values in the packed table -- the entries before it, its own header, and for the first entries
		 * what the heap holds in front of the table (sis_heap[]: the allocation's chunk header). The next invocation blanks
		 * line 1's sync area, so all that stays of this are the stream's first samples in front of the window. */
		int16_t *o1 = orc_line_ptr(s, 0);
		if(!o1) return;
		for(b = 0; b < nb; b++)
		{
			const int e = 50 - nb + b;
			const orc_pulse_t *p = &s->sis_lut[e];
			long at;
			if(!((vbi[b >> 3] >> (7 - (b & 7))) & 1)) continue;
			for(i = -p->offset, at = 0; i < p->length && at < s->width; i++, at++)
			{
				const long q = s->sis_pos[e] + i;           /* index into the packed table */
				int16_t v;
				if(q >= 0) v = s->sis_packed[q];
				else v = q >= -8 ? s->sis_heap[q + 8] : 0;
				o1[at] += v;
			}
		}
		return;
	}
	o = orc_line_ptr(s, g);
	if(!o) return;

	/* blank the data area to the sync level through the window */
	for(x = s->sis_blank_left; x < s->sis_blank_left + s->sis_blank_width; x++)
	{
		const int w = s->sis_blank_win[x - s->sis_blank_left];
		if(x < 0 || x >= s->width) continue;
		o[x] = (int16_t) ((o[x] * (INT16_MAX - w) + s->sync_level * w) >> 15);
	}

	/* vbidata_render(lut, vbi, 50 - nb, nb, MSB first): the first 50 - nb entries are passed over, bit b then meets
	 * entry 50 - nb + b (src/vbidata.c:186-239) */
	for(b = 0; b < nb; b++)
	{
		const orc_pulse_t *p = &s->sis_lut[50 - nb + b];
		if(!((vbi[b >> 3] >> (7 - (b & 7))) & 1)) continue;
		for(i = 0; i < p->length; i++)
		{
			const int at = p->offset + i;
			if(at >= 0 && at < s->width) o[at] += p->value[i];
		}
	}
}
