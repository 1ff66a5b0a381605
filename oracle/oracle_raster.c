/* oracle/oracle_raster.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of one scanline of hacktv's raster build for PAL / NTSC /
 * monochrome modes: _vid_next_line_raster (src/video.c:2864-3066), the line
 * sequence tables (src/video.c:2447-2862), the pulse renderer
 * (src/vbidata.c:186-239) and the zero-history chroma FIR
 * (src/fir.c:357-375 -> :304-355).
 *
 * The raster stream is kept as the I channel only (the reference's Q channel
 * is zero until the filter / audio stages).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

int16_t *orc_line_ptr(orc_t *s, long g)
{
	if(g < s->s_first || g >= s->s_first + s->s_count) return(NULL);
	return(s->S + (g - s->s_first) * s->width);
}

int16_t *orc_cline_ptr(orc_t *s, long g)
{
	if(!s->C || g < s->s_first || g >= s->s_first + s->s_count) return(NULL);
	return(s->C + (g - s->s_first) * s->width);
}

/* Per-line content code. The reference keeps these as four-character strings
 * per line number (src/video.c:2447-2810); restated here as field-position
 * rules for the 625- and 525-line interlaced rasters.
 *
 * left:  0 none, 'h' line sync, 'v' short (equalising) pulse, 'V' long (broad) pulse
 * mid:   0 none, 'v' or 'V' half-line pulse
 * burst: 0 never, 1 always, 2 even frames only, 3 odd frames only (the
 *        reference's '1' is tested as (frame & 1) == 0 and its '2' as
 *        (frame & 1) == 1, src/video.c:2901-2903 -- the code, not the comment
 *        at :2462-2463, is what is restated)
 * la/ra: active video in the left / right half of the line */
typedef struct {
	char left, mid;
	int burst;
	int la, ra;
} _linecode_t;

static _linecode_t _line_code(int type, int line)
{
	_linecode_t c = { 'h', 0, 1, 1, 1 }; /* an ordinary picture line: "h0aa" */

	if(type == HVK_RASTER_625)
	{
		if(line <= 2 || line == 314 || line == 315)       c = (_linecode_t) { 'V', 'V', 0, 0, 0 };
		else if(line == 3)                                c = (_linecode_t) { 'V', 'v', 0, 0, 0 };
		else if(line == 4 || line == 5 || line == 311 || line == 312 ||
		        line == 316 || line == 317 || line >= 624) c = (_linecode_t) { 'v', 'v', 0, 0, 0 };
		else if(line == 313)                              c = (_linecode_t) { 'v', 'V', 0, 0, 0 };
		else if(line == 318)                              c = (_linecode_t) { 'v', 0, 0, 0, 0 };
		else if(line == 6)                                c = (_linecode_t) { 'h', 0, 2, 0, 0 };
		else if(line == 319)                              c = (_linecode_t) { 'h', 0, 3, 0, 0 };
		else if(line <= 22 || (line >= 320 && line <= 335)) c = (_linecode_t) { 'h', 0, 1, 0, 0 };
		else if(line == 23)                               c = (_linecode_t) { 'h', 0, 1, 0, 1 };
		else if(line == 310 || line == 622)               c = (_linecode_t) { 'h', 0, 2, 1, 1 };
		else if(line == 623)                              c = (_linecode_t) { 'h', 'v', 0, 1, 0 };
	}
	else if(type == HVK_RASTER_525)
	{
		if(line <= 3 || (line >= 7 && line <= 9) || line == 264 || line == 265 ||
		   line == 270 || line == 271)                    c = (_linecode_t) { 'v', 'v', 0, 0, 0 };
		else if(line <= 6 || line == 267 || line == 268)  c = (_linecode_t) { 'V', 'V', 0, 0, 0 };
		else if(line == 266)                              c = (_linecode_t) { 'v', 'V', 0, 0, 0 };
		else if(line == 269)                              c = (_linecode_t) { 'V', 'v', 0, 0, 0 };
		else if(line == 272)                              c = (_linecode_t) { 'v', 0, 0, 0, 0 };
		else if(line <= 20 || (line >= 273 && line <= 282)) c = (_linecode_t) { 'h', 0, 1, 0, 0 };
		else if(line == 263)                              c = (_linecode_t) { 'h', 'v', 1, 1, 0 };
		else if(line == 283)                              c = (_linecode_t) { 'h', 0, 1, 0, 1 };
	}
	else if(type == HVK_RASTER_819)
	{
		/* src/video.c:2592-2696: no colour, no equalising pulses, one long pulse per field */
		c = (_linecode_t) { 'h', 0, 0, 1, 1 };                                       /* "h_aa" */
		if(line == 1)                                     c = (_linecode_t) { 'V', 0, 0, 0, 0 };
		else if(line <= 38 || line >= 817 || (line >= 407 && line <= 446 && line != 409)) c = (_linecode_t) { 'h', 0, 0, 0, 0 };
		else if(line == 406)                              c = (_linecode_t) { 'h', 0, 0, 1, 0 };
		else if(line == 409)                              c = (_linecode_t) { 'h', 'V', 0, 0, 0 };
		else if(line == 447)                              c = (_linecode_t) { 'h', 0, 0, 0, 1 };
	}
	else if(type == HVK_RASTER_405)
	{
		/* src/video.c:2697-2738 */
		if(line <= 4 || (line >= 204 && line <= 206))     c = (_linecode_t) { 'V', 'V', 0, 0, 0 };
		else if(line == 207)                              c = (_linecode_t) { 'V', 0, 0, 0, 0 };
		else if(line <= 15 || (line >= 208 && line <= 217)) c = (_linecode_t) { 'h', 0, 1, 0, 0 };
		else if(line == 203)                              c = (_linecode_t) { 'h', 'V', 1, 1, 0 };
		else if(line == 218)                              c = (_linecode_t) { 'h', 0, 1, 0, 1 };
	}
	else if(type == HVK_CBS_405)
	{
		/* src/video.c:2739-2779 */
		c = (_linecode_t) { 'h', 0, 0, 1, 1 };
		if(line <= 3 || (line >= 7 && line <= 9) || line == 204 || line == 205 ||
		   line == 210 || line == 211)                    c = (_linecode_t) { 'v', 'v', 0, 0, 0 };
		else if(line <= 6 || line == 207 || line == 208)  c = (_linecode_t) { 'V', 'V', 0, 0, 0 };
		else if(line == 206)                              c = (_linecode_t) { 'v', 'V', 0, 0, 0 };
		else if(line == 209)                              c = (_linecode_t) { 'V', 'v', 0, 0, 0 };
		else if(line == 212)                              c = (_linecode_t) { 'v', 0, 0, 0, 0 };
		else if(line <= 14 || (line >= 213 && line <= 216)) c = (_linecode_t) { 'h', 0, 0, 0, 0 };
		else if(line == 203)                              c = (_linecode_t) { 'h', 'v', 0, 1, 0 };
		else if(line == 217)                              c = (_linecode_t) { 'h', 0, 0, 0, 1 };
	}
	else if(type == HVK_APOLLO_320)
	{
		/* src/video.c:2780-2784 */
		c = line <= 8 ? (_linecode_t) { 'V', 'v', 0, 0, 0 } : (_linecode_t) { 'h', 0, 0, 1, 1 };
	}
	else if(type == HVK_BAIRD_240)
	{
		/* src/video.c:2785-2812 */
		c = (_linecode_t) { 'h', 0, 0, 1, 1 };
		if(line <= 12)                                    c = (_linecode_t) { 'V', 'V', 0, 0, 0 };
		else if(line <= 20)                               c = (_linecode_t) { 'h', 0, 0, 0, 0 };
	}
	else if(type == HVK_BAIRD_30)
	{
		c = (_linecode_t) { 0, 0, 0, 1, 1 };             /* no sync pulses at all, src/video.c:2813-2817 */
	}
	else if(type == HVK_NBTV_32)
	{
		c = (_linecode_t) { (char) (line == 1 ? 0 : 'h'), 0, 0, 1, 1 };            /* src/video.c:2818-2825 */
	}

	return(c);
}

/* Source row shown on a line, before centring (src/video.c:2812-2862) */
static int _source_row(int type, int line)
{
	if(type == HVK_RASTER_625) return(line < 313 ? (line - 23) * 2 : (line - 336) * 2 + 1);
	if(type == HVK_RASTER_525) return(line < 265 ? (line - 23) * 2 : (line - 286) * 2 + 1);
	if(type == HVK_RASTER_819) return(line < 406 ? (line - 48) * 2 : (line - 457) * 2 + 1);
	if(type == HVK_RASTER_405) return(line < 210 ? (line - 16) * 2 : (line - 218) * 2 + 1);
	if(type == HVK_CBS_405)    return(line < 210 ? (line - 16) * 2 : (line - 219) * 2 + 1);
	if(type == HVK_APOLLO_320) return(line - 9);
	if(type == HVK_BAIRD_240)  return(line - 20);
	if(type == HVK_BAIRD_30 || type == HVK_NBTV_32) return(line - 1);
	return(-1);
}

/* Add one pulse to line g, continuing into the neighbouring lines when it
 * starts before sample 0 or runs past the end. A line that is not part of
 * the stream (before the first line) is a boundary: the part of the pulse
 * that would land there is dropped (src/vbidata.c:211-236). */
static void _add_pulse(orc_t *s, long g, const orc_pulse_t *p)
{
	long n = g * s->width + p->offset;
	int x;

	for(x = 0; x < p->length; x++, n++)
	{
		int16_t *l;
		if(n < 0) continue;
		/* running on into the line behind: that line's buffer has width 0 -- a boundary the renderer stops at
		 * (src/vbidata.c:219-236) -- until the raster has been round the ring once (src/video.c:4665) */
		if(n / s->width > g && g < s->olines - 1) break;
		l = orc_line_ptr(s, n / s->width);
		if(l == NULL) continue;
		l[n % s->width] += p->value[x];
	}
}

/* Zero-history centred FIR over one interleaved channel of the chroma
 * buffer, in place. Samples past the end of the channel come from
 * s->chroma[2*width ...], where the caller has placed the ghost values. */
static void _chroma_fir(orc_t *s, int16_t *ch)
{
	int h = s->chroma_ntaps / 2;
	int W = s->width;
	int16_t *in = malloc((W + 2 * h) * sizeof(int16_t));
	int j, k;

	for(j = 0; j < h; j++) in[j] = 0;
	for(j = 0; j < W + h; j++) in[h + j] = ch[j * 2];

	for(j = 0; j < W; j++)
	{
		int32_t a = 0;
		for(k = 0; k < s->chroma_ntaps; k++) a += (int32_t) in[j + k] * s->chroma_taps[k];
		a >>= 15;
		ch[j * 2] = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
	}

	free(in);
}

void orc_raster_line(orc_t *s, long g)
{
	const hvk_config_t *c = &s->conf;
	int frame = g / c->lines + 1;
	int line = g % c->lines + 1;
	_linecode_t code = _line_code(c->type, line);
	int16_t *o = orc_line_ptr(s, g);
	int W = s->width;
	int vy, pal = 0, x, fsc = 0;
	const c16_t *lut = NULL;
	int vframe_x, vframe_y;

	orc_select_frame(s, line);
	vframe_x = (s->active_width - s->fb_width) / 2;
	vframe_y = (c->active_lines - s->fb_height) / 2;

	/* Building a line first blanks the one after it (src/video.c:2935-2939);
	 * the very first line was blanked at start-up (:4659-4663) */
	{
		int16_t *nl = orc_line_ptr(s, g + 1);
		int16_t *cl = orc_cline_ptr(s, g), *ncl = orc_cline_ptr(s, g + 1);
		if(g == 0) for(x = 0; x < W; x++) o[x] = s->blanking_level;
		if(nl) for(x = 0; x < W; x++) nl[x] = s->blanking_level;
		/* ... and its Q channel is cleared (src/video.c:2938) */
		if(g == 0 && cl) memset(cl, 0, W * sizeof(int16_t));
		if(ncl) memset(ncl, 0, W * sizeof(int16_t));
	}

	if(c->raw_bb)
	{
		/* _vid_next_line_rawbb, src/video.c:2406-2446: the line is the next `width` samples of the
		 * external stream (which starts over at its end), mapped from its levels onto the mode's */
		for(x = 0; x < W; x++)
		{
			int in = (s->rawbb && s->rawbb_len > 0) ? s->rawbb[(g * W + x) % s->rawbb_len] : 0;
			o[x] = s->blanking_level +
				((in - c->raw_bb_blanking_level) * (s->white_level - s->blanking_level) / (c->raw_bb_white_level - c->raw_bb_blanking_level));
		}
		return;
	}

	/* src/video.c:2884-2895 */
	vy = _source_row(c->type, line);
	if(vy >= 0 && c->interlaced != 0 && s->fb_interlaced != c->interlaced) vy += 1;
	vy -= vframe_y;
	if(vy < 0 || vy >= s->fb_height) vy = -1;

	if(c->colour_mode == HVK_PAL || c->colour_mode == HVK_NTSC)
	{
		/* src/video.c:2900-2916 */
		pal  = code.burst == 1;
		pal |= code.burst == 2 && (frame & 1) == 0;
		pal |= code.burst == 3 && (frame & 1) == 1;

		lut = &s->colour_lookup[s->colour_lookup_offset];
		s->colour_lookup_offset += W;
		s->colour_lookup_offset %= s->colour_lookup_width;

		if(c->colour_mode == HVK_PAL && pal && ((frame + line) & 1)) pal = -1;

		if(pal)
		{
			memset(s->chroma, 0, sizeof(int16_t) * 2 * W);
			memcpy(s->chroma + 2 * W, s->ghost, sizeof(s->ghost));
		}
	}

	/* field-sequential colour: which channel this field shows (src/video.c:2919-2930) */
	if(c->colour_mode == HVK_APOLLO_FSC) fsc = (frame * 2 + (line < 264 ? 0 : 1)) % 3;
	if(c->colour_mode == HVK_CBS_FSC)    fsc = (frame * 2 + (line < 202 ? 0 : 1)) % 3;

	/* sync pulses, bit order h, v, V, mid-v, mid-V (src/video.c:2944-2958) */
	if(code.left == 'h') _add_pulse(s, g, &s->sync[0]);
	if(code.left == 'v') _add_pulse(s, g, &s->sync[1]);
	if(code.left == 'V') _add_pulse(s, g, &s->sync[2]);
	if(code.mid == 'v')  _add_pulse(s, g, &s->sync[3]);
	if(code.mid == 'V')  _add_pulse(s, g, &s->sync[4]);

	/* active video (src/video.c:2961-3009): luma is ASSIGNED, not added */
	if(code.la || code.ra)
	{
		int al = code.la ? s->active_left : s->half_width;
		int ar = code.ra ? s->active_left + s->active_width : s->half_width;
		int16_t black = s->yuv[0];
		const uint32_t *prgb = NULL;
		int stride = 0;

		/* (NBTV at 800 kHz: active_left + active_width = W + 1 -- the reference writes one sample past its line buffer,
		 * into its heap; here the stream is contiguous and that sample would be the next line's first: not written) */
		for(x = al; x < s->active_left + vframe_x; x++) if(x < W) o[x] = black;

		if(s->fb && vy >= 0)
		{
			prgb = &s->fb[vy * s->fb_line_stride];
			prgb += (x - s->active_left - vframe_x) * s->fb_pixel_stride;
			stride = s->fb_pixel_stride;
		}

		for(; x < s->active_left + vframe_x + s->fb_width && x < ar; x++)
		{
			uint32_t rgb = prgb ? (*prgb & 0xFFFFFF) : 0;
			if(c->colour_mode == HVK_APOLLO_FSC || c->colour_mode == HVK_CBS_FSC)
			{
				/* one channel as a grey, src/video.c:2995-3000 */
				rgb = (rgb >> (8 * fsc)) & 0xFF;
				rgb |= (rgb << 8) | (rgb << 16);
			}
			if(x < W) o[x] = s->yuv[rgb * 3 + 0];
			if(pal && x < W)
			{
				s->chroma[x * 2 + 0] = s->yuv[rgb * 3 + 1];
				s->chroma[x * 2 + 1] = s->yuv[rgb * 3 + 2];
			}
			if(prgb) prgb += stride;
		}

		for(; x < ar; x++) if(x < W) o[x] = black;
	}

	if(pal)
	{
		/* src/video.c:3015-3040 */
		if(s->chroma_ntaps > 0)
		{
			_chroma_fir(s, s->chroma + 0);
			_chroma_fir(s, s->chroma + 1);
		}

		for(x = 0; x < s->burst_width; x++)
		{
			s->chroma[(s->burst_left + x) * 2 + 0] = (s->burst_phase.i * s->burst_win[x]) >> 15;
			s->chroma[(s->burst_left + x) * 2 + 1] = (s->burst_phase.q * s->burst_win[x]) >> 15;
		}

		{
			/* onto the luma, or -- S-Video -- into the Q channel (src/video.c:3032) */
			int16_t *dst = c->s_video ? orc_cline_ptr(s, g) : o;
			for(x = 0; x < W; x++)
			{
				dst[x] += (lut[x].i * s->chroma[x * 2 + 1] * pal +
				           lut[x].q * s->chroma[x * 2 + 0]) >> 15;
			}
		}
	}

	/* the field-sequential colour flags (src/video.c:3043-3063) */
	if(c->colour_mode == HVK_APOLLO_FSC && fsc == 1 && (line == 18 || line == 281)) _add_pulse(s, g, &s->fsc[0]);
	if(c->colour_mode == HVK_CBS_FSC && fsc == 2 && (line == 1 || line == 203)) _add_pulse(s, g, &s->fsc[line == 1 ? 0 : 1]);
}

/* What the SECAM process needs to know about a line (src/video.c:3078-3090) */
void orc_line_info(orc_t *s, long g, int *frame, int *line, int *la, int *ra, int *vy)
{
	const hvk_config_t *c = &s->conf;
	int vframe_y;
	_linecode_t code;

	*frame = g / c->lines + 1;
	*line = g % c->lines + 1;
	orc_select_frame(s, *line);
	vframe_y = (c->active_lines - s->fb_height) / 2;
	code = _line_code(c->type, *line);
	*la = code.la;
	*ra = code.ra;

	*vy = _source_row(c->type, *line);
	if(*vy >= 0 && c->interlaced != 0 && s->fb_interlaced != c->interlaced) *vy += 1;
	*vy -= vframe_y;
	if(*vy < 0 || *vy >= s->fb_height) *vy = -1;
}
