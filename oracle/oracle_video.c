/* oracle/oracle_video.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of hacktv's composite-video -> int16 IQ path
 * (vid_init / vid_next_line, src/video.c:3812-4952) for the raster modes the
 * MI355X engine covers. It is the checker the HIP path is compared with:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * it, never the product.
 *
 * Pinning: the reference ships no tests or golden vectors,
 * so this restatement is pinned against the reference ITSELF, compiled
 * unmodified from /root/reference by oracle/Makefile (oracle/_ref):
 *   - every table against the reference's vid_init() output (tests/test_oracle_tables.py),
 *   - the IQ stream against the reference CLI's output, both live
 *     (tests/test_oracle_vs_ref.py, where oracle/_ref exists) and through
 *     committed digests and line excerpts (tests/golden/, made by
 *     oracle/make_golden.py).
 *
 * Structure. The reference is a ring of line buffers walked by a chain of
 * line processes on threads (src/video.c:3543-3618, :4867-4934). The
 * restatement keeps the same arithmetic but states each stage over the
 * continuous sample stream:
 *   raster   S[n]              oracle_raster.c   (I channel; sync edges spill
 *                                                 into the previous line)
 *   filter   F[n] = sum_k tap[k] * S[n - 25 + k] >> 15, clamped; S[n<0] = 0
 *                                                 (src/fir.c:304-355, :564-615;
 *                                                 delay src/video.c:3620-3625)
 *   audio    out[n] = F[n] + carriers[n + delay_lines * width]
 *                                                 oracle_audio.c (src/video.c:3261-3450;
 *                                                 SURVEY.md H3 for the offset)
 */
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

orc_t *orc_open(const hvk_config_t *conf, unsigned int sample_rate)
{
	return(orc_open_rates(conf, sample_rate, 0));
}

orc_t *orc_open_rates(const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate)
{
	orc_t *s = calloc(1, sizeof(orc_t));
	if(!s) return(NULL);

	s->conf = *conf;
	s->sample_rate = sample_rate;
	s->pixel_rate = pixel_rate ? pixel_rate : sample_rate;   /* src/video.c:3839 */

	/* S-Video: baseband colour modes only (src/hacktv.c:1136-1148) */
	if(conf->s_video && (conf->output_type != HVK_INT16_REAL || conf->colour_mode == HVK_MONOCHROME)) { free(s); return(NULL); }

	if(orc_build_tables(s) != 0 || orc_audio_init(s) != 0 || orc_tail_init(s) != 0 || orc_vbi_init(s) != 0 || orc_sis_init(s) != 0)
	{
		orc_close(s);
		return(NULL);
	}

	/* no frame yet: an empty frame the size of the active area (src/video.c:4169-4177) */
	s->fb = NULL;
	s->fb_width = s->active_width;
	s->fb_height = s->conf.active_lines;
	s->field_fb[0].width = s->field_fb[1].width = s->active_width;
	s->field_fb[0].height = s->field_fb[1].height = s->conf.active_lines;

	return(s);
}

void orc_close(orc_t *s)
{
	if(!s) return;
	orc_free_tables(s);
	orc_audio_free(s);
	orc_tail_free(s);
	orc_vbi_free(s);
	orc_sis_free(s);
	orc_teletext_free(s);
	free(s->S);
	free(s->C);
	free(s->last_raster);
	free(s->last_carrier);
	free(s->last_widths);
	free(s->cbuf);
	free(s->ciq);
	free(s->ccar);
	free(s->prev_r);
	free(s->prev_q);
	free(s->rs_win2);
	free(s->prev_w);
	free(s);
}

int orc_info(orc_t *s, int32_t *out, int n)
{
	int32_t v[] = {
		s->width, s->half_width, s->active_width, s->active_left,
		s->conf.lines, s->conf.active_lines,
		s->white_level, s->black_level, s->blanking_level, s->sync_level,
		(int32_t) s->colour_lookup_width, s->burst_left, s->burst_width,
		s->burst_phase.i, s->burst_phase.q,
		s->chroma_ntaps, 0, s->max_width,
		s->fm_mono.level, s->nicam.ntaps, s->nicam.sps, s->nicam.dsl, s->nicam.decimation,
		s->nicam.cc_len,
		s->am_mono.level, s->am_mono.delta.i, s->am_mono.delta.q,
	};
	int c = sizeof(v) / sizeof(v[0]);
	if(n < c) c = n;
	memcpy(out, v, c * sizeof(int32_t));
	return(sizeof(v) / sizeof(v[0]));
}

static long _copy(void *dst, long max_bytes, const void *src, long bytes)
{
	if(src == NULL) return(0);
	if(dst == NULL) return(bytes);
	if(bytes > max_bytes) bytes = max_bytes;
	memcpy(dst, src, bytes);
	return(bytes);
}

long orc_table(orc_t *s, const char *name, void *dst, long max_bytes)
{
	if(strcmp(name, "secam_iir") == 0)
	{
		/* as oracle/ref_probe.c's: the pre-emphasis filter's state and the chrominance buffer */
		static unsigned char tmp[16 + 2 * 8192 * 2];
		if(s->conf.colour_mode != HVK_SECAM || !s->chroma || s->width > 8192) return(0);
		memcpy(tmp, &s->sc_ix, 8);
		memcpy(tmp + 8, &s->sc_iy, 8);
		memcpy(tmp + 16, s->chroma, (size_t) 2 * s->width * 2);
		return(_copy(dst, max_bytes, tmp, 16 + (long) 2 * s->width * 2));
	}
	if(strcmp(name, "syncs") == 0) return(_copy(dst, max_bytes, s->sync_packed, s->sync_packed_len * sizeof(int16_t)));
	if(strcmp(name, "yuv") == 0) return(_copy(dst, max_bytes, s->yuv, 0x1000000L * 3 * sizeof(int16_t)));
	if(strcmp(name, "colour_lookup") == 0) return(_copy(dst, max_bytes, s->colour_lookup, s->colour_lookup ? (long) (s->colour_lookup_width + s->width) * sizeof(c16_t) : 0));
	if(strcmp(name, "burst_win") == 0) return(_copy(dst, max_bytes, s->burst_win, (long) s->burst_width * sizeof(int16_t)));
	if(strcmp(name, "chroma_taps") == 0) return(_copy(dst, max_bytes, s->chroma_taps, (long) s->chroma_ntaps * sizeof(int16_t)));
	if(strcmp(name, "chroma_ghost") == 0) return(_copy(dst, max_bytes, s->ghost, sizeof(s->ghost)));
	if(strcmp(name, "vfilter_itaps") == 0) return(_copy(dst, max_bytes, s->vf_itaps, (long) s->vf_ntaps * sizeof(int16_t)));
	if(strcmp(name, "vfilter_qtaps") == 0) return(_copy(dst, max_bytes, s->vf_qtaps, (long) s->vf_ntaps * sizeof(int16_t)));
	if(strcmp(name, "fm_mono_lut") == 0) return(_copy(dst, max_bytes, s->fm_mono.lut, 65536L * sizeof(c32_t)));
	if(strcmp(name, "resampler_taps") == 0) return(_copy(dst, max_bytes, s->rs_taps, s->rs_taps ? (long) s->rs_L * s->rs_ataps * sizeof(int16_t) : 0));
	if(strcmp(name, "fm_video_lut") == 0) return(_copy(dst, max_bytes, s->fm_video.lut, s->fm_video.lut ? 65536L * sizeof(c32_t) : 0));
	if(strcmp(name, "nicam_taps") == 0) return(_copy(dst, max_bytes, s->nicam.taps, (long) s->nicam.ntaps * sizeof(int16_t)));
	if(strcmp(name, "nicam_cc") == 0) return(_copy(dst, max_bytes, s->nicam.cc, (long) s->nicam.cc_len * sizeof(c16_t)));
	if(strcmp(name, "fm_secam_lut") == 0) return(_copy(dst, max_bytes, s->sc_lut, s->sc_lut ? 65536L * sizeof(c32_t) : 0));
	if(strcmp(name, "fm_secam_bell") == 0) return(_copy(dst, max_bytes, s->sc_bell, s->sc_bell ? 65535L * sizeof(c16_t) : 0));
	if(strcmp(name, "fm_secam_fir") == 0) return(_copy(dst, max_bytes, s->sc_fir, s->sc_fir ? 15L * 2 : 0));
	if(strcmp(name, "secam_l_fir") == 0) return(_copy(dst, max_bytes, s->sc_notch, s->sc_notch ? 51L * 2 : 0));
	if(strcmp(name, "teletext_lut") == 0)
	{
		/* the reference's packed layout [length][offset][values...]...[-1] (src/vbidata.c:196-202) */
		long n = 0, o = 0;
		int b;
		int16_t *t;
		if(!s->tt_sym) return(0);
		for(b = 0; b < 360; b++) n += 2 + s->tt_sym[b].length;
		n += 1;
		if(dst == NULL) return(n * 2);
		t = malloc(n * 2);
		for(b = 0; b < 360; b++)
		{
			t[o++] = s->tt_sym[b].length;
			t[o++] = s->tt_sym[b].offset;
			memcpy(t + o, s->tt_sym[b].value, s->tt_sym[b].length * 2);
			o += s->tt_sym[b].length;
		}
		t[o++] = -1;
		n = _copy(dst, max_bytes, t, n * 2);
		free(t);
		return(n);
	}
	if(strcmp(name, "limiter_shape") == 0) return(_copy(dst, max_bytes, s->fm_mono.lim.shape, (long) s->fm_mono.lim.width * sizeof(int16_t)));
	if(strcmp(name, "limiter_vtaps") == 0) return(_copy(dst, max_bytes, s->fm_mono.lim.vtaps, (long) s->fm_mono.lim.ntaps * sizeof(int32_t)));
	if(strcmp(name, "limiter_ftaps") == 0) return(_copy(dst, max_bytes, s->fm_mono.lim.ftaps, (long) s->fm_mono.lim.ntaps * sizeof(int32_t)));
	return(-1);
}

int orc_teletext_packets(orc_t *s, long frame_index, const uint8_t *packets, uint32_t mask)
{
	int q = frame_index % 16;
	if(s->conf.lines != 625) return(-1);
	if(!s->tt_sym && orc_teletext_init(s) != 0) return(-1);
	s->tt_queue[q].frame = frame_index + 1;     /* 0 marks an empty entry */
	s->tt_queue[q].mask = mask;
	memcpy(s->tt_queue[q].packets, packets, 32 * 45);
	return(0);
}

void orc_set_ghost(orc_t *s, const int16_t *ghost, int n)
{
	memset(s->ghost, 0, sizeof(s->ghost));
	if(n > 32) n = 32;
	memcpy(s->ghost, ghost, n * sizeof(int16_t));
}

void orc_set_frame(orc_t *s, const uint32_t *fb, int width, int height, int pixel_stride, int line_stride, int interlaced)
{
	/* centre-crop to the active area (src/video.c:4887-4893, src/av.c:293-303) */
	int x = (width - s->active_width) / 2;
	int y = (height - s->conf.active_lines) / 2;
	int w = s->active_width, h = s->conf.active_lines;

	if(x < 0) { w += x; x = 0; }
	if(y < 0) { h += y; y = 0; }
	if(x + w > width) w = width - x;
	if(y + h > height) h = height - y;

	s->fb = fb ? fb + y * line_stride + x * pixel_stride : NULL;
	s->fb_width = w;
	s->fb_height = h;
	s->fb_pixel_stride = pixel_stride;
	s->fb_line_stride = line_stride;
	s->fb_interlaced = interlaced;

	/* both fields show it until orc_set_frame2() says otherwise */
	s->field_fb[0].fb = s->fb; s->field_fb[0].width = w; s->field_fb[0].height = h;
	s->field_fb[0].pixel_stride = pixel_stride; s->field_fb[0].line_stride = line_stride; s->field_fb[0].interlaced = interlaced;
	s->field_fb[1] = s->field_fb[0];
}

void orc_set_frame2(orc_t *s, const uint32_t *fb, int width, int height, int pixel_stride, int line_stride, int interlaced)
{
	int x = (width - s->active_width) / 2;
	int y = (height - s->conf.active_lines) / 2;
	int w = s->active_width, h = s->conf.active_lines;

	if(x < 0) { w += x; x = 0; }
	if(y < 0) { h += y; y = 0; }
	if(x + w > width) w = width - x;
	if(y + h > height) h = height - y;

	s->field_fb[1].fb = fb ? fb + y * line_stride + x * pixel_stride : NULL;
	s->field_fb[1].width = w; s->field_fb[1].height = h;
	s->field_fb[1].pixel_stride = pixel_stride; s->field_fb[1].line_stride = line_stride; s->field_fb[1].interlaced = interlaced;
}

void orc_select_frame(orc_t *s, int line)
{
	const int f = (s->conf.interlace && s->conf.hline > 0 && line >= s->conf.hline) ? 1 : 0;
	s->fb = s->field_fb[f].fb;
	s->fb_width = s->field_fb[f].width;
	s->fb_height = s->field_fb[f].height;
	s->fb_pixel_stride = s->field_fb[f].pixel_stride;
	s->fb_line_stride = s->field_fb[f].line_stride;
	s->fb_interlaced = s->field_fb[f].interlaced;
}

void orc_set_rawbb(orc_t *s, const int16_t *samples, long nsamples)
{
	s->rawbb = samples;
	s->rawbb_len = nsamples;
}

void orc_set_audio(orc_t *s, const int16_t *stereo, long nsamples, int loop)
{
	s->audio_src = stereo;
	s->audio_len = nsamples;
	s->audio_pos = 0;
	s->audio_loop = loop;
}

/* Make sure raster lines up to and including `last` exist. Lines older than
 * `keep_from` may be dropped. */
static void _raster_until(orc_t *s, long last, long keep_from)
{
	long need_first = keep_from < 0 ? 0 : keep_from;
	long need_count = last + 2 - need_first; /* + the line blanked ahead */
	int W = s->width;

	if(need_first > s->s_first && s->s_count > 0)
	{
		long drop = need_first - s->s_first;
		if(drop > s->s_count) drop = s->s_count;
		memmove(s->S, s->S + drop * W, (s->s_count - drop) * W * sizeof(int16_t));
		if(s->C) memmove(s->C, s->C + drop * W, (s->s_count - drop) * W * sizeof(int16_t));
		s->s_first += drop;
		s->s_count -= drop;
	}
	if(s->s_count == 0) s->s_first = need_first;

	if(need_count > s->s_cap)
	{
		s->S = realloc(s->S, need_count * W * sizeof(int16_t));
		if(s->conf.s_video) s->C = realloc(s->C, need_count * W * sizeof(int16_t));
		s->s_cap = need_count;
	}
	if(need_count > s->s_count)
	{
		/* new lines enter zeroed; the raster blanks each before use */
		memset(s->S + s->s_count * W, 0, (need_count - s->s_count) * W * sizeof(int16_t));
		if(s->C) memset(s->C + s->s_count * W, 0, (need_count - s->s_count) * W * sizeof(int16_t));
		s->s_count = need_count;
	}

	while(s->rastered <= last)
	{
		orc_raster_line(s, s->rastered);
		s->rastered++;

		/* SECAM colour is a separate line process two slots behind the raster
		 * (src/video.c:4206-4212, :4676-4688): line r - 1 is processed once line
		 * r has been built (and has put its sync edge into r - 1's tail). Before
		 * the first real line the process is handed two never-emitted slots
		 * with frame 1, line 0, which it treats as picture lines without a
		 * picture; they advance its IIR state. */
		/* (not with --raw-bb-file: the line reader stands where the raster AND the colour process would, src/video.c:4190) */
		if(s->conf.colour_mode == HVK_SECAM && !s->conf.raw_bb)
		{
			int frame, line, la, ra, vy;

			if(s->sc_done == 0 && s->rastered == 1)
			{
				int16_t *scratch = malloc(W * sizeof(int16_t));
				int k, x;
				for(k = 0; k < 2; k++)
				{
					for(x = 0; x < W; x++) scratch[x] = s->blanking_level;
					orc_secam_line(s, scratch, NULL, 1, 0, 1, 1, -1);
				}
				free(scratch);
			}

			if(s->rastered >= 2)
			{
				long g = s->rastered - 2;
				orc_line_info(s, g, &frame, &line, &la, &ra, &vy);
				orc_secam_line(s, orc_line_ptr(s, g), orc_cline_ptr(s, g), frame, line, la, ra, vy);
				s->sc_done++;
			}
		}

		/* VITS, WSS, VITC: line processes between the colour process and teletext
		 * (src/video.c:4214-4316) */
		if((s->conf.vits || s->conf.wss || s->conf.vitc || s->conf.acp || s->conf.cc608) && s->rastered >= 2)
		{
			long g = s->rastered - 2;
			const c16_t *lut = NULL;
			if(!s->conf.raw_bb && s->colour_lookup && (s->conf.colour_mode == HVK_PAL || s->conf.colour_mode == HVK_NTSC))   /* rawbb lines have no sub-carrier table (src/video.c:2414) */
			{
				/* the table position advances by one line per line (src/video.c:2906-2910) */
				lut = &s->colour_lookup[(unsigned long) (((unsigned long long) g * s->width) % s->colour_lookup_width)];
			}
			orc_vbi_line(s, g, (int) (g / s->conf.lines) + 1, (int) (g % s->conf.lines) + 1, lut);
		}

		/* sound-in-syncs: behind CC608, in front of teletext (src/video.c:4330-4338); its first invocation works on the
		 * never-emitted slot in front of line 1 */
		if(s->conf.sis && s->conf.raw_bb)
		{
			/* the process that reads the lines in works on one line and this one shares its slot (src/video.c:4190 with
			 * :4676-4688): the line just read, no slot in front of line 1 */
			orc_sis_line(s, s->rastered - 1, 0);
		}
		else if(s->conf.sis)
		{
			if(s->rastered == 1)
			{
				const int dummies = s->conf.colour_mode == HVK_SECAM ? 3 : 1;
				int d;
				for(d = 1; d <= dummies; d++) orc_sis_line(s, -1, d == dummies);
			}
			else orc_sis_line(s, s->rastered - 2, 0);
		}

		/* teletext comes after the colour process and before the filter
		 * (src/video.c:4346-4359): 16 lines per field (src/teletext.c:1222-1224) */
		if(s->tt_sym && s->rastered >= 2)
		{
			long g = s->rastered - 2;
			long fi = g / s->conf.lines;
			int line = g % s->conf.lines + 1;
			int slot = (line >= 7 && line <= 22) ? line - 7 : ((line >= 320 && line <= 335) ? 16 + line - 320 : -1);
			int q = fi % 16;
			if(slot >= 0 && s->tt_queue[q].frame == fi + 1 && ((s->tt_queue[q].mask >> slot) & 1))
			{
				orc_teletext_render(s, orc_line_ptr(s, g), s->tt_queue[q].packets[slot]);
			}
		}
	}
}

/* ---- the line pipeline behind the raster, one chunk (= one raster line's worth of samples) at a time ----
 *
 * The reference's processes pass line slots down a ring (src/video.c:4676-4688): a process with
 * a two-slot window (resampler, filter) reads the newer slot and writes the older one, so its
 * output for raster line N lands in the slot of line N - 1, and at start-up in a slot whose line
 * number is still 0. Slots with line < 1 are dropped at the output (src/video.c:4936-4952) but
 * they do carry samples, and every later process runs over them. Hence:
 *   - the video filter carries `width` samples of latency (ntaps / 2 + its configured delay,
 *     src/video.c:3620-3625), exactly the one slot its output is shifted by: with the filter on,
 *     chunk c of the output holds the filtered raster line c - 1 and the first chunk is dropped;
 *   - the resampler (src/video.c:3627-3651) has NO latency to make up for its slot shift: its first
 *     chunk -- the resampled raster line 1 -- is dropped as well, and the stream starts with line 2;
 *   - the audio process and the tail run over the dropped chunks too: their state
 *     is ahead by the dropped chunks' widths when the first emitted sample is made. */

/* the poly-phase resampler of src/fir.c:304-355 (fir_int16_process with interpolation L,
 * decimation D): returns the number of samples made from `n` inputs */
static int _resample_ch(orc_t *s, int *pd, int16_t *win, const int16_t *in, int n, int16_t *out)
{
	int x = 0, i, y, d = *pd;

	for(i = 0; i < n;)
	{
		if(d >= s->rs_L)
		{
			d -= s->rs_L;
			memmove(win, win + 1, (s->rs_ataps - 1) * sizeof(int16_t));
			win[s->rs_ataps - 1] = in[i++];
		}

		for(; d < s->rs_L; d += s->rs_D)
		{
			const int16_t *taps = &s->rs_taps[d * s->rs_ataps];
			int32_t a = 0;
			for(y = 0; y < s->rs_ataps; y++) a += (int32_t) win[y] * taps[y];
			a >>= 15;
			out[x++] = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
		}
	}

	*pd = d;
	return(x);
}

static int _resample(orc_t *s, const int16_t *in, int n, int16_t *out)
{
	return(_resample_ch(s, &s->rs_d, s->rs_win, in, n, out));
}

/* the video filter as the reference runs it (src/fir.c:304-355 real, :564-615 real -> complex):
 * a window of ataps + delay samples; each output is made from the OLDEST ataps of them */
static void _filter(orc_t *s, const int16_t *in, int n, int16_t *iq)
{
	const int lwin = s->vf_ntaps + s->vf_delay;
	int x, k;

	for(x = 0; x < n; x++)
	{
		int32_t ai = 0, aq = 0;

		memmove(s->vf_win, s->vf_win + 1, (lwin - 1) * sizeof(int16_t));
		s->vf_win[lwin - 1] = in[x];

		for(k = 0; k < s->vf_ntaps; k++) ai += (int32_t) s->vf_win[k] * s->vf_itaps[k];
		if(s->vf_type == 3)
		{
			for(k = 0; k < s->vf_ntaps; k++) aq += (int32_t) s->vf_win[k] * s->vf_qtaps[k];
		}
		ai >>= 15;
		aq >>= 15;
		iq[x * 2 + 0] = ai < INT16_MIN ? INT16_MIN : (ai > INT16_MAX ? INT16_MAX : ai);
		iq[x * 2 + 1] = aq < INT16_MIN ? INT16_MIN : (aq > INT16_MAX ? INT16_MAX : aq);
	}
}

long orc_render_lines(orc_t *s, int16_t *iq, long nlines)
{
	const int W = s->width;
	const int ndrop = (s->rs_taps ? 1 : 0) + s->delay_lines;
	long o = 0, emitted = 0;
	int x;

	if(nlines <= 0) return(0);

	if(!s->cbuf)
	{
		s->cbuf = malloc(s->max_width * sizeof(int16_t));
		s->ciq = malloc(s->max_width * 2 * sizeof(int16_t));
		s->ccar = malloc(s->max_width * 2 * sizeof(int16_t));
		s->prev_r = calloc((size_t) (s->delay_lines + 1) * s->max_width, sizeof(int16_t));
		s->prev_w = calloc(s->delay_lines + 1, sizeof(int));
	}

	free(s->last_raster);
	free(s->last_carrier);
	free(s->last_widths);
	s->last_raster = malloc(nlines * s->max_width * sizeof(int16_t));
	s->last_carrier = calloc(nlines * s->max_width * 2, sizeof(int16_t));
	s->last_widths = calloc(nlines, sizeof(int));
	s->last_raster_len = 0;
	s->last_carrier_len = 0;
	s->last_nwidths = 0;

	while(emitted < nlines)
	{
		const long c = s->chunks_done;
		const int16_t *line;
		int w, slot;

		/* every line also takes the leading sync edge of its successor: raster one line ahead */
		_raster_until(s, c + 1, c - 1);
		line = orc_line_ptr(s, c);

		if(s->rs_taps) w = _resample(s, line, W, s->cbuf);
		else
		{
			memcpy(s->cbuf, line, W * sizeof(int16_t));
			w = W;
		}

		/* the chunk the filter's output is centred on: delay_lines chunks back */
		slot = c % (s->delay_lines + 1);
		if(emitted < nlines && c >= ndrop)
		{
			const int back = (int) ((c - s->delay_lines) % (s->delay_lines + 1));
			const int bw = s->delay_lines ? s->prev_w[back] : w;
			const int16_t *br = s->delay_lines ? s->prev_r + (size_t) back * s->max_width : s->cbuf;
			memcpy(s->last_raster + s->last_raster_len, br, bw * sizeof(int16_t));
			s->last_raster_len += bw;
		}
		if(s->delay_lines)
		{
			memcpy(s->prev_r + (size_t) slot * s->max_width, s->cbuf, w * sizeof(int16_t));
			s->prev_w[slot] = w;
		}

		if(s->vf_type == 0)
		{
			for(x = 0; x < w; x++)
			{
				s->ciq[x * 2 + 0] = s->cbuf[x];
				s->ciq[x * 2 + 1] = 0;
			}
		}
		else _filter(s, s->cbuf, w, s->ciq);

		/* S-Video: the filter leaves the Q channel alone, and the slot it writes the luma of
		 * line c - delay_lines into is that line's own: Q is its sub-carrier (src/video.c:3235-3248) */
		if(s->conf.s_video && s->rs_taps)
		{
			/* ... behind the resampler, which has a second channel for it (src/video.c:4361-4367: an instance of the same
			 * filter, fed the Q channel of the same slots): the sub-carrier of chunk c - delay_lines, resampled.
			 * The reference keeps it in the Q channel of the lines' own buffers, a ring of `olines` of them (src/video.c:3578):
			 * the raster writes line c's sub-carrier into buffer c at the raster's width, the resampler reads it there and
			 * writes chunk c over the front of buffer c - 1 (its two line pointers: dst = lines[0], src = lines[1]), and the
			 * filter process then gives the line it finishes the width of the chunk it has just been fed (dst->width =
			 * fir_int16_process(), src/video.c:3243). Where the widths differ from line to line (525 lines at 16 MHz:
			 * 1017, 1017, ..., 1016) a line one sample shorter than its new width ends on what its buffer held before:
			 * the raster's sub-carrier of the line before it at that place when resampling downwards, the raster's blanking
			 * (zero) when upwards. Hence the same ring of buffers here, written in the same order. */
			const int ring = s->olines > s->delay_lines + 2 ? s->olines : s->delay_lines + 2;
			const size_t qw = (size_t) (s->max_width > W ? s->max_width : W);
			const int16_t *cl = orc_cline_ptr(s, c);
			int16_t *own;
			int wq, back;
			if(!s->prev_q)
			{
				s->prev_q = calloc((size_t) ring * qw, sizeof(int16_t));
				s->rs_win2 = calloc(s->rs_ataps, sizeof(int16_t));
				s->rs_d2 = s->rs_L;
			}
			own = s->prev_q + (size_t) (c % ring) * qw;
			/* (the raster blanks the whole buffer -- max_width samples, the resampled chunks' width where that is the larger,
			 * src/video.c:2934-2939 with :3645-3646 -- two lines before it writes its line there) */
			memset(own, 0, qw * sizeof(int16_t));
			if(cl) memcpy(own, cl, W * sizeof(int16_t));
			wq = _resample_ch(s, &s->rs_d2, s->rs_win2, own, W, s->prev_q + (size_t) ((c - 1 + ring) % ring) * qw);
			(void) wq;      /* == w: both channels consume the same inputs from the same phase */
			back = (int) ((c - s->delay_lines - 1 + 2 * ring) % ring);
			for(x = 0; x < w; x++) s->ciq[x * 2 + 1] = c >= s->delay_lines ? s->prev_q[(size_t) back * qw + x] : 0;
		}
		else if(s->conf.s_video)
		{
			const int16_t *cq = orc_cline_ptr(s, c - s->delay_lines);
			for(x = 0; x < w; x++) s->ciq[x * 2 + 1] = cq ? cq[x] : 0;
		}

		memset(s->ccar, 0, w * 2 * sizeof(int16_t));
		orc_audio_line(s, s->ciq, w, s->ccar);
		orc_tail_line(s, s->ciq, w);

		if(c >= ndrop)
		{
			memcpy(iq + o * 2, s->ciq, w * 2 * sizeof(int16_t));
			memcpy(s->last_carrier + s->last_carrier_len * 2, s->ccar, w * 2 * sizeof(int16_t));
			s->last_carrier_len += w;
			s->last_widths[s->last_nwidths++] = w;
			o += w;
			emitted++;
		}

		s->chunks_done++;
	}

	s->emitted += nlines;
	return(o);
}

long orc_last_widths(orc_t *s, int32_t *dst, long max)
{
	long n = s->last_nwidths < max ? s->last_nwidths : max, i;
	for(i = 0; dst && i < n; i++) dst[i] = s->last_widths[i];
	return(s->last_nwidths);
}

long orc_last_raster(orc_t *s, int16_t *dst, long max_samples)
{
	long n = s->last_raster_len < max_samples ? s->last_raster_len : max_samples;
	if(dst && n > 0) memcpy(dst, s->last_raster, n * sizeof(int16_t));
	return(s->last_raster_len);
}

long orc_last_carrier(orc_t *s, int16_t *dst, long max_samples)
{
	long n = s->last_carrier_len < max_samples ? s->last_carrier_len : max_samples;
	if(dst && n > 0) memcpy(dst, s->last_carrier, n * 2 * sizeof(int16_t));
	return(s->last_carrier_len);
}
