/* oracle/oracle_vbi.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of three of the reference's VBI inserters, each a line process
 * between the colour process and teletext (src/video.c:4214-4316):
 *   vits  insertion test signals        src/vits.c
 *   wss   widescreen signalling, line 23 src/wss.c
 *   vitc  vertical interval time code    src/vitc.c
 * The data-line inserters shape their bits through vbidata step tables
 * (src/vbidata.c:58-81, :145-184) and vbidata_render (src/vbidata.c:186-239).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

/* one stepped symbol, as vbidata_update_step leaves it */
static void _step(orc_pulse_t *p, double offset, double width, double rise, int level)
{
	int x1 = floor(offset - rise / 2);
	int x2 = ceil(offset + width + rise / 2);
	int x;

	p->value = calloc(x2 - x1 + 2 > 0 ? x2 - x1 + 2 : 1, sizeof(int16_t));
	p->offset = p->length = 0;

	for(x = x1; x <= x2; x++)
	{
		int v = round(orc_rc_window(x, offset, width, rise) * level);
		if(v == 0) continue;
		if(p->length == 0) p->offset = x;
		p->value[x - p->offset] = v;
		p->length = x - p->offset + 1;
	}
}

static orc_pulse_t *_step_table(int nsymbols, int level, double width, double rise, double offset)
{
	orc_pulse_t *t = calloc(nsymbols, sizeof(orc_pulse_t));
	int b;
	for(b = 0; b < nsymbols; b++) _step(&t[b], offset + width * b, width, rise, level);
	return(t);
}

static void _free_table(orc_pulse_t *t, int n)
{
	int b;
	if(!t) return;
	for(b = 0; b < n; b++) free(t[b].value);
	free(t);
}

/* vbidata_render on the I samples of line g of the raster stream; symbols that begin before
 * sample 0 reach into the previous line (none of these tables does at the offsets used) */
static void _render(orc_t *s, long g, const orc_pulse_t *lut, int nsym, const uint8_t *src, int offset, int length, int msb_first)
{
	int16_t *o = orc_line_ptr(s, g);
	int b, i, x;

	for(b = -offset, i = 0; b < length && i < nsym; b++, i++)
	{
		const orc_pulse_t *p = &lut[i];
		int bit = b < 0 ? 0 : (src[b >> 3] >> (msb_first ? 7 - (b & 7) : (b & 7))) & 1;
		if(!bit) continue;
		for(x = 0; x < p->length; x++)
		{
			long at = p->offset + x;
			if(at >= 0 && at < s->width) o[at] += p->value[x];
			else if(at < 0 && g > 0)
			{
				int16_t *prev = orc_line_ptr(s, g - 1);
				if(prev) prev[s->width + at] += p->value[x];
			}
		}
	}
}

/* ---- WSS ---- */

static int _group_bits(uint8_t *vbi, uint8_t code, int offset, int length)
{
	int i, b;
	while(length--)
	{
		for(i = 0; i < 6; i++, offset++)
		{
			if(i == 3) code ^= 1;
			b = 7 - (offset % 8);
			vbi[offset / 8] &= ~(1 << b);
			vbi[offset / 8] |= (code & 1) << b;
		}
		code >>= 1;
	}
	return(offset);
}

/* ---- VITS ---- */

static double _sin2(double t, double position, double width, double amplitude)
{
	double a;
	t -= position - width;
	if(t <= 0 || t >= width * 2) return(0);
	a = t / (width * 2) * M_PI;
	return(pow(sin(a), 2) * amplitude);
}

static void _vits_625(orc_t *s, int level)
{
	static const double bursts[6] = { 0.5e6, 1.0e6, 2.0e6, 4.0e6, 4.8e6, 5.8e6 };
	double ts = 1.0 / 25 / 625, h = ts / 32, bs[6];
	int i, x, b, W = s->width;

	ts = ts / W;
	for(b = 0; b < 6; b++) bs[b] = 2.0 * M_PI * bursts[b];

	for(i = 0; i < 4; i++)
	{
		s->vits_line[i] = calloc(W * 2, sizeof(int16_t));
		for(x = 0; x < W; x++)
		{
			double t = ts * x, r = 0.0, c = 0.0;
			switch(i)
			{
			case 0:
				r += orc_rc_window(t, 6 * h, 5 * h, 200e-9) * 0.70;
				r += _sin2(t, 13 * h, 200e-9, 0.70);
				r += _sin2(t, 16 * h, 2000e-9, 0.70 / 2);
				c += _sin2(t, 16 * h, 2000e-9, 0.70 / 2);
				r += orc_rc_window(t, 20 * h, 2 * h, 200e-9) * 0.14;
				r += orc_rc_window(t, 22 * h, 2 * h, 200e-9) * 0.28;
				r += orc_rc_window(t, 24 * h, 2 * h, 200e-9) * 0.42;
				r += orc_rc_window(t, 26 * h, 2 * h, 200e-9) * 0.56;
				r += orc_rc_window(t, 28 * h, 3 * h, 200e-9) * 0.70;
				break;
			case 1:
				r += orc_rc_window(t, 6 * h, 25 * h, 200e-9) *  0.35;
				r += orc_rc_window(t, 6 * h,  2 * h, 200e-9) *  0.21;
				r += orc_rc_window(t, 8 * h,  2 * h, 200e-9) * -0.21;
				for(b = 0; b < 6; b++)
				{
					r += orc_rc_window(t, (12 + 3 * b) * h, 2 * h, 200e-9) * 0.21
					   * sin((t - (12 + 3 * b) * h) * bs[b]);
				}
				break;
			case 2:
				r += orc_rc_window(t, 6 * h, 5 * h, 200e-9) * 0.70;
				r += _sin2(t, 13 * h, 200e-9, 0.70);
				c += orc_rc_window(t, 15 * h, 15 * h, 1e-6) * 0.28 / 2;
				r += orc_rc_window(t, 20 * h, 2 * h, 200e-9) * 0.14;
				r += orc_rc_window(t, 22 * h, 2 * h, 200e-9) * 0.28;
				r += orc_rc_window(t, 24 * h, 2 * h, 200e-9) * 0.42;
				r += orc_rc_window(t, 26 * h, 2 * h, 200e-9) * 0.56;
				r += orc_rc_window(t, 28 * h, 3 * h, 200e-9) * 0.70;
				break;
			case 3:
				r += orc_rc_window(t, 6 * h, 25 * h, 200e-9) * 0.35;
				c += orc_rc_window(t, 7 * h, 7 * h, 1e-6) * 0.70 / 2;
				c += orc_rc_window(t, 17 * h, 13 * h, 1e-6) * 0.42 / 2;
				break;
			}
			s->vits_line[i][x * 2 + 0] = lround(r / 0.7 * level);
			s->vits_line[i][x * 2 + 1] = lround(c / 0.7 * level);
		}
	}
}

static void _vits_525(orc_t *s, int level)
{
	static const double bursts[6] = { 0.50e6, 1.00e6, 2.00e6, 3.00e6, 3.58e6, 4.20e6 };
	double ts = 1001.0 / 30000 / 525, h = ts / 128, bs[6];
	int i, x, b, W = s->width;

	ts = ts / W;
	for(b = 0; b < 6; b++) bs[b] = 2.0 * M_PI * bursts[b];

	for(i = 0; i < 2; i++)
	{
		s->vits_line[i] = calloc(W * 2, sizeof(int16_t));
		for(x = 0; x < W; x++)
		{
			double t = ts * x, r = 0.0, c = 0.0;
			if(i == 0)
			{
				r += orc_rc_window(t, 24 * h, 36 * h, 125e-9) * 100;
				r += _sin2(t, 68 * h, 250e-9, 100);
				r += _sin2(t, 75 * h, 1570e-9, 100 / 2);
				c += _sin2(t, 75 * h, 1570e-9, 100 / 2);
				r += orc_rc_window(t,  92 * h,  6 * h, 250e-9) * 18;
				r += orc_rc_window(t,  98 * h,  6 * h, 250e-9) * 36;
				r += orc_rc_window(t, 104 * h,  6 * h, 250e-9) * 54;
				r += orc_rc_window(t, 110 * h,  6 * h, 250e-9) * 72;
				r += orc_rc_window(t, 116 * h,  8 * h, 250e-9) * 90;
				c += orc_rc_window(t,  84 * h, 38 * h, 400e-9) * 40 / 2;
			}
			else
			{
				r += orc_rc_window(t, 24 * h, 8 * h, 125e-9) * 100;
				r += orc_rc_window(t, 32 * h, 92 * h, 125e-9) * 50;
				r += orc_rc_window(t, 36 * h, 12 * h, 250e-9) * 50 / 2 * sin((t - 36 * h) * bs[0]);
				for(b = 1; b < 6; b++)
				{
					r += orc_rc_window(t, (40 + 8 * b) * h, 8 * h, 250e-9) * 50 / 2
					   * sin((t - (40 + 8 * b) * h) * bs[b]);
				}
				c += orc_rc_window(t,  92 * h,  8 * h, 400e-9) * 20 / 2;
				c += orc_rc_window(t, 100 * h,  8 * h, 400e-9) * 40 / 2;
				c += orc_rc_window(t, 108 * h, 12 * h, 400e-9) * 80 / 2;
			}
			s->vits_line[i][x * 2 + 0] = lround(r / 100 * level);
			s->vits_line[i][x * 2 + 1] = lround(c / 100 * level);
		}
	}
}

int orc_vbi_init(orc_t *s)
{
	const hvk_config_t *c = &s->conf;

	if(c->vits)
	{
		if(c->colour_mode == HVK_PAL)
		{
			double p = 60.0 * (M_PI / 180.0);
			s->vits_phase.i = round(cos(p) * INT16_MAX);
			s->vits_phase.q = round(sin(p) * INT16_MAX);
		}
		else
		{
			s->vits_phase.i = 0;
			s->vits_phase.q = -INT16_MAX;
		}
		if(c->lines == 625) _vits_625(s, s->white_level - s->blanking_level);
		else if(c->lines == 525) _vits_525(s, s->white_level - s->blanking_level);
		else return(-1);
	}

	if(c->wss)
	{
		static const uint8_t lead[7] = { 0xF8, 0xE3, 0x8E, 0x38, 0xF1, 0xE0, 0xF8 };
		int level = round((s->white_level - s->black_level) * (5.0 / 7.0)), o;

		if(c->lines != 625 || (c->wss > 0x0F && c->wss != 0xFF)) return(-1);
		s->wss_lut = _step_table(137, level, (double) s->pixel_rate * 200e-9, (double) s->pixel_rate * 200e-9, (double) s->pixel_rate * 11e-6);
		memset(s->wss_vbi, 0, sizeof(s->wss_vbi));
		memcpy(s->wss_vbi, lead, sizeof(lead));
		o = _group_bits(s->wss_vbi, c->wss, 29 + 24, 4);
		o = _group_bits(s->wss_vbi, 0x00, o, 4);
		o = _group_bits(s->wss_vbi, 0x00, o, 3);
		o = _group_bits(s->wss_vbi, 0x00, o, 3);
		s->wss_blank_width = round(s->pixel_rate * 42.5e-6);
	}

	if(c->vitc)
	{
		int level = round((s->white_level - s->black_level) * 0.785);
		if(c->type == HVK_RASTER_625) { s->vitc_lines[0] = 19; s->vitc_lines[1] = 332; s->vitc_hr = 116; }
		else { s->vitc_lines[0] = 14; s->vitc_lines[1] = 277; s->vitc_hr = 115; }
		if(c->frame_rate.num <= 30 && c->frame_rate.den == 1) { s->vitc_fps = c->frame_rate.num; s->vitc_drop = 0; }
		else if(c->frame_rate.num == 30000 && c->frame_rate.den == 1001) { s->vitc_fps = 30; s->vitc_drop = 1; }
		else return(-1);
		s->vitc_lut = _step_table(s->vitc_hr, level, (double) s->width / s->vitc_hr, s->pixel_rate * 200e-9, 0);
	}

	if(c->acp)
	{
		/* src/acp.c:26-64 */
		double left = c->lines == 625 ? 8.88e-6 : 8.288e-6;
		double spacing = c->lines == 625 ? 5.92e-6 : 8.288e-6;
		double psync_width = c->lines == 625 ? 2.368e-6 : 2.222e-6;
		int i;
		s->acp_psync_level = s->sync_level + round((s->white_level - s->sync_level) * 0.06);
		s->acp_pagc_level  = s->sync_level + round((s->white_level - s->sync_level) * 1.10);
		s->acp_psync_width = round(s->pixel_rate * psync_width);
		s->acp_pagc_width  = round(s->pixel_rate * 2.7e-6);
		for(i = 0; i < 6; i++) s->acp_left[i] = round(s->pixel_rate * (left + spacing * i));
	}

	if(c->cc608)
	{
		/* src/cc608.c:97-160 */
		double offset, x, w, level;
		int i;

		if(c->type == HVK_RASTER_525) { s->cc_line = 21; offset = 27.382e-6; }
		else { s->cc_line = 22; offset = 27.5e-6; }

		s->cc_lut = _step_table(32, level = round((s->white_level - s->black_level) * 0.5), (double) s->width / 32,
		                        s->pixel_rate * 240e-9 * 2.0738786, s->pixel_rate * offset);
		w = (double) s->width * 7 / 32;
		x = (double) s->pixel_rate * offset - (s->width * 8.75 / 32);
		s->cc_cri_x = x;
		s->cc_cri_len = ceil(w);
		s->cc_cri = malloc(sizeof(int16_t) * s->cc_cri_len);
		for(i = 0; i < s->cc_cri_len; i++)
		{
			s->cc_cri[i] = (0.5 - cos(((double) i - (x - s->cc_cri_x)) * (2 * M_PI / w * 7)) * 0.5) * level;
		}
	}

	return(0);
}

void orc_set_frame_aspect(orc_t *s, long long par_num, long long par_den)
{
	s->fb_par_num = par_num;
	s->fb_par_den = par_den;
}

void orc_set_cc608(orc_t *s, long frame_index, uint8_t c1, uint8_t c2)
{
	s->cc_frame = frame_index + 1;
	s->cc_pair[0] = c1;
	s->cc_pair[1] = c2;
}

void orc_vbi_free(orc_t *s)
{
	int i;
	_free_table(s->cc_lut, 32);
	free(s->cc_cri);
	for(i = 0; i < 4; i++) free(s->vits_line[i]);
	_free_table(s->wss_lut, 137);
	_free_table(s->vitc_lut, s->vitc_hr);
}

static int _bits(uint8_t *data, int offset, uint64_t bits, int nbits)
{
	for(; nbits; nbits--, offset++, bits >>= 1)
	{
		uint8_t b = 1 << (offset & 7);
		if(bits & 1) data[offset >> 3] |= b;
		else data[offset >> 3] &= ~b;
	}
	return(offset);
}

/* The three processes on raster line g (frame and line count from 1), in the reference's order.
 * lut: the line's colour sub-carrier table (NULL without PAL / NTSC colour). */
void orc_vbi_line(orc_t *s, long g, int frame, int line, const c16_t *lut)
{
	const hvk_config_t *c = &s->conf;
	int16_t *o = orc_line_ptr(s, g);
	int x, i;

	if(c->vits)
	{
		i = -1;
		if(c->lines == 625)
		{
			if(line == 17 || line == 18) i = line - 17;
			else if(line == 330 || line == 331) i = line - 330 + 2;
		}
		else
		{
			if(line == 17) i = 0;
			else if(line == 280) i = 1;
		}

		if(i >= 0)
		{
			const int16_t *v = s->vits_line[i];
			for(x = 0; x < s->width; x++)
			{
				o[x] += v[x * 2 + 0];
				if(lut) o[x] += (((s->vits_phase.i * lut[x].q + s->vits_phase.q * lut[x].i) >> 15) * v[x * 2 + 1]) >> 15;
			}
		}
	}

	if(c->wss && line == 23)
	{
		if(c->wss == 0xFF)
		{
			/* auto (src/wss.c:166-179): 4:3 or 16:9 from the source's pixel aspect against the
			 * aspect at which the active area is 14:9 wide; r64_cmp, src/common.c:76-80 */
			long long pn = s->fb_par_den ? s->fb_par_num : 1, pd = s->fb_par_den ? s->fb_par_den : 1;
			long long tn = 14LL * c->active_lines, td = 9LL * s->active_width;
			_group_bits(s->wss_vbi, (pn * td - pd * tn) <= 0 ? 0x08 : 0x07, 29 + 24, 4);
		}
		for(x = s->half_width; x < s->wss_blank_width; x++) o[x] = s->black_level;
		_render(s, g, s->wss_lut, 137, s->wss_vbi, 0, 137, 1);
	}

	if(c->acp)
	{
		/* src/acp.c:73-128 */
		int on = 0;

		if(line == 1)
		{
			/* the AGC pulse level follows a clipped sawtooth */
			i = abs(frame * 4 % 1712 - 856) - 150;
			if(i < 0) i = 0;
			else if(i > 255) i = 255;
			i = s->yuv[(long) (i << 16 | i << 8 | i) * 3 + 0];
			s->acp_pagc_level = s->sync_level + round((i - s->sync_level) * 1.10);
		}

		if(c->lines == 625) on = (line >= 9 && line <= 18) || (line >= 321 && line <= 330);
		else on = (line >= 12 && line <= 19) || (line >= 275 && line <= 282);

		/* lines already held are left alone (src/acp.c:108): of the inserters only VITS comes earlier, and SECAM's
		 * field identification lines are marked by the colour process before any of them (src/video.c:3135) */
		if(on && !(c->vits && orc_vbi_allocated_by_vits(s, line)) && !orc_vbi_allocated_by_secam(s, line))
		{
			for(i = 0; i < 6; i++)
			{
				for(x = s->acp_left[i]; x < s->acp_left[i] + s->acp_psync_width; x++) o[x] = s->acp_psync_level;
				for(; x < s->acp_left[i] + s->acp_psync_width + s->acp_pagc_width; x++) o[x] = s->acp_pagc_level;
			}
		}
	}

	if(c->vitc && (line == s->vitc_lines[0] || line == s->vitc_lines[0] + 2 || line == s->vitc_lines[1] || line == s->vitc_lines[1] + 2))
	{
		uint32_t timecode;
		uint8_t data[12], crc;
		int fn = frame, n;

		if(s->vitc_drop)
		{
			fn += (fn / 17982) * 18;
			fn += (fn % 18000 - 2) / 1798 * 2;
		}

		timecode  = (fn % s->vitc_fps % 10) << 0;
		timecode |= (fn % s->vitc_fps / 10) << 4;
		timecode |= (s->vitc_drop ? 1 : 0) << 6;
		timecode |= 1 << 7;
		fn /= s->vitc_fps;
		timecode |= (fn % 10) << 8;
		timecode |= (fn / 10 % 6) << 12;
		if(c->type != HVK_RASTER_625) timecode |= (line >= s->vitc_lines[1] ? 1 : 0) << 15;
		fn /= 60;
		timecode |= (fn % 10) << 16;
		timecode |= (fn / 10 % 6) << 20;
		fn /= 60;
		timecode |= (fn % 24 % 10) << 24;
		timecode |= (fn % 24 / 10) << 28;
		if(c->type == HVK_RASTER_625) timecode |= (uint32_t) (line >= s->vitc_lines[1] ? 1 : 0) << 31;

		for(n = i = 0; i < 8; i++)
		{
			n = _bits(data, n, 0x01, 2);
			n = _bits(data, n, timecode >> (i * 4), 4);
			n = _bits(data, n, 0, 4);
		}
		n = _bits(data, n, 0x01, 2);
		_bits(data, n, 0, 8);
		for(crc = i = 0; i < 11; i++) crc ^= data[i];
		crc = ((crc << 6) | (crc >> 2)) & 0xFF;
		n = _bits(data, n, crc, 8);

		_render(s, g, s->vitc_lut, s->vitc_hr, data, 21, n, 0);
	}

	if(c->cc608 && line == s->cc_line)
	{
		/* src/cc608.c:188-221: the frame's byte pair, zeros when there is none */
		uint8_t c1 = 0, c2 = 0, data[3];

		if(s->cc_frame == frame) { c1 = s->cc_pair[0]; c2 = s->cc_pair[1]; }
		c1 = (c1 & 0x7F) | 0x80;
		c2 = (c2 & 0x7F) | 0x80;
		for(i = 1; i < 8; i++)
		{
			c1 ^= (c1 << i) & 0x80;
			c2 ^= (c2 << i) & 0x80;
		}
		data[0] = (c1 << 1) | 0x01;
		data[1] = (c2 << 1) | (c1 >> 7);
		data[2] = (c2 >> 7);

		for(i = 0; i < s->cc_cri_len; i++) o[s->cc_cri_x + i] += s->cc_cri[i];
		_render(s, g, s->cc_lut, 32, data, 0, 17, 0);
	}
}

/* SECAM's colour process marks its field identification lines as held (src/video.c:3101-3103, :3135); it runs where the
 * mode's colour is SECAM and the video is not raw baseband (src/video.c:4200-4215) */
int orc_vbi_allocated_by_secam(orc_t *s, int line)
{
	const hvk_config_t *c = &s->conf;
	if(c->colour_mode != HVK_SECAM || c->raw_bb || !c->secam_field_id) return(0);
	return((line >= 7 && line < 7 + s->sc_fid_lines) || (line >= 320 && line < 320 + s->sc_fid_lines));
}

int orc_vbi_allocated_by_vits(orc_t *s, int line)
{
	if(s->conf.lines == 625) return(line == 17 || line == 18 || line == 330 || line == 331);
	return(line == 17 || line == 280);
}

/* does one of the inserters above occupy this line? (teletext leaves such lines alone,
 * src/teletext.c:1219, and keeps the packet for the next free one) */
int orc_vbi_allocated(orc_t *s, int line)
{
	const hvk_config_t *c = &s->conf;
	if(orc_vbi_allocated_by_secam(s, line)) return(1);
	if(c->vits)
	{
		if(c->lines == 625 && (line == 17 || line == 18 || line == 330 || line == 331)) return(1);
		if(c->lines == 525 && (line == 17 || line == 280)) return(1);
	}
	if(c->wss && line == 23) return(1);
	if(c->acp && c->lines == 625 && ((line >= 9 && line <= 18) || (line >= 321 && line <= 330))) return(1);
	if(c->acp && c->lines == 525 && ((line >= 12 && line <= 19) || (line >= 275 && line <= 282))) return(1);
	if(c->cc608 && line == s->cc_line) return(1);
	if(c->vitc && (line == s->vitc_lines[0] || line == s->vitc_lines[0] + 2 || line == s->vitc_lines[1] || line == s->vitc_lines[1] + 2)) return(1);
	return(0);
}
