/* hvk_presets.c -- TV system presets for the modes the engine renders.
 *
 * Mirrors the reference's `vid_configs[]` lookup (src/video.c:1956-2008,
 * used by `-m <id>` at src/hacktv.c:1078-1107): same mode ids, same numbers
 * (the numbers are the broadcast standards' and the reference's level
 * choices, src/video.c:50-1009). Presets are composed from a raster timing,
 * a colour system and an RF/audio plan instead of one flat struct per mode.
 */
#include <string.h>
#include "hacktv_amd.h"

/* ---- raster timings ---- */

static void _raster_625(hvk_config_t *c, double sync_rise)
{
	c->type = HVK_RASTER_625;
	c->frame_rate = (hvk_rational_t) { 25, 1 };
	c->lines = 625;
	c->interlaced = 1;
	c->active_lines = 576;
	c->active_width = 0.00005195;        /* 51.95 us */
	c->active_left = 0.00001040;         /* 10.40 us */
	c->hsync_width = 0.00000470;         /*  4.70 us */
	c->vsync_short_width = 0.00000235;   /*  2.35 us */
	c->vsync_long_width = 0.00002730;    /* 27.30 us */
	c->sync_rise = sync_rise;
}

static void _raster_525w(hvk_config_t *c, double active_width, double sync_rise);

static void _raster_525(hvk_config_t *c)
{
	_raster_525w(c, 0.00005290, 0.00000025);     /* 52.90 us */
}

static void _raster_525w(hvk_config_t *c, double active_width, double sync_rise)
{
	c->type = HVK_RASTER_525;
	c->frame_rate = (hvk_rational_t) { 30000, 1001 };
	c->lines = 525;
	c->interlaced = 1;
	c->active_lines = 480;
	c->active_width = active_width;
	c->active_left = 0.00000920;         /*  9.20 us */
	c->hsync_width = 0.00000470;
	c->vsync_short_width = 0.00000230;
	c->vsync_long_width = 0.00002710;
	c->sync_rise = sync_rise;
}

/* the rasters of src/video.c:1303-1960 other than 625 / 525 lines: their numbers */
static void _raster(hvk_config_t *c, int type, int64_t fnum, int64_t fden, int lines, int interlaced, int active_lines,
                    double active_width, double active_left, double hsync, double vshort, double vlong, double sync_rise)
{
	c->type = type;
	c->frame_rate = (hvk_rational_t) { fnum, fden };
	c->lines = lines;
	c->interlaced = interlaced;
	c->active_lines = active_lines;
	c->active_width = active_width;
	c->active_left = active_left;
	c->hsync_width = hsync;
	c->vsync_short_width = vshort;
	c->vsync_long_width = vlong;
	c->sync_rise = sync_rise;
}
static void _raster_819(hvk_config_t *c) { _raster(c, HVK_RASTER_819, 25, 1, 819, 1, 720, 0.00003944, 0.00000890, 0.00000250, 0, 0.00002000, 0); }
static void _raster_405(hvk_config_t *c) { _raster(c, HVK_RASTER_405, 25, 1, 405, 2, 378, 0.00008030, 0.00001680, 0.00000900, 0, 0.00004000, 0.00000025); }
static void _raster_240(hvk_config_t *c) { _raster(c, HVK_BAIRD_240, 25, 1, 240, 0, 220, 0.00015, 0.000016667, 0.000013333, 0, 0.000166667, 0); }
static void _raster_30(hvk_config_t *c)  { _raster(c, HVK_BAIRD_30, 25, 2, 30, 0, 30, 0.002666667, 0, 0, 0, 0, 0); c->frame_orientation = HVK_ROTATE_270 | HVK_HFLIP; }
static void _raster_nbtv(hvk_config_t *c) { _raster(c, HVK_NBTV_32, 25, 2, 32, 0, 32, 2.5e-3 - 0.1e-3, 0.1e-3, 0.1e-3, 0, 0, 0); c->frame_orientation = HVK_ROTATE_270 | HVK_HFLIP; }
/* (the long pulse is wider than half a line: the short one completes it, src/video.c:1821-1823) */
static void _raster_apollo(hvk_config_t *c) { _raster(c, HVK_APOLLO_320, 10, 1, 320, 0, 312, 0.00028250, 0.00002500, 0.00002000, 1.0 / 10.0 / 320.0 / 2.0 - 45e-6, 0.00026750, 0); }
static void _raster_cbs(hvk_config_t *c) { _raster(c, HVK_CBS_405, 72, 1, 405, 1, 376, 0.00002812, 0.00000480, 0.000002743, 0.000001372, 0.000014746, 0); }

/* NTSC on 405 lines (BBC Engineering Division Monograph No. 32, Appendix A; src/video.c:1422-1433, :1538-1550) */
static void _colour_ntsc405(hvk_config_t *c, double colour_bw)
{
	c->colour_mode = HVK_NTSC;
	c->burst_width = 0.00000339;
	c->burst_rise = 0.00000030;
	c->burst_left = 0.00001050;
	c->burst_level = 3.0 / 7.0;
	c->colour_carrier = (hvk_rational_t) { 5315625, 2 };   /* 2657812.5 Hz */
	c->colour_bw = colour_bw;
	c->ev_co = 0.877;
	c->eu_co = 0.493;
}

static void _fsc(hvk_config_t *c, int mode, double width, double left, double level)
{
	c->colour_mode = mode;
	c->fsc_flag_width = width;
	c->fsc_flag_left = left;
	c->fsc_flag_level = level;
}

/* plain AM: the real signal on I (src/video.h:71) */
static void _am(hvk_config_t *c, double white, double black, double blank, double sync)
{
	c->output_type = HVK_INT16_COMPLEX;
	c->modulation = HVK_AM;
	c->level = 1.0;
	c->video_level = 1.0;
	c->white_level = white;
	c->black_level = black;
	c->blanking_level = blank;
	c->sync_level = sync;
}

/* ---- colour systems ---- */

static void _colour_pal(hvk_config_t *c)
{
	c->colour_mode = HVK_PAL;
	c->burst_width = 0.00000225;         /* 2.25 us */
	c->burst_rise = 0.00000030;
	c->burst_left = 0.00000560;          /* 5.6 us after 0H */
	c->burst_level = 3.0 / 7.0;          /* of white - blanking */
	c->colour_carrier = (hvk_rational_t) { 17734475, 4 }; /* 4433618.75 Hz */
	c->colour_bw = 1.4e6;
	c->ev_co = 0.877;
	c->eu_co = 0.493;
}

/* PAL on the American rasters' sub-carriers (src/video.c:316-455): burst 2.52 us at 5.3 us, 33 / 73 */
static void _colour_pal_mn(hvk_config_t *c, int64_t num, int64_t den)
{
	_colour_pal(c);
	c->burst_width = 0.00000252;
	c->burst_left = 0.00000530;
	c->burst_level = 33.0 / 73.0;
	c->colour_carrier = (hvk_rational_t) { num, den };
}

static void _colour_ntsc(hvk_config_t *c)
{
	c->colour_mode = HVK_NTSC;
	c->burst_width = 0.00000250;
	c->burst_rise = 0.00000030;
	c->burst_left = 0.00000530;
	c->burst_level = 4.0 / 10.0;
	c->colour_carrier = (hvk_rational_t) { 39375000, 11 }; /* 3579545.45 Hz */
	c->colour_bw = 1.4e6;
	c->ev_co = 0.877;
	c->eu_co = 0.493;
}

static void _colour_secam(hvk_config_t *c)
{
	c->colour_mode = HVK_SECAM;
	c->burst_width = 0.00005690;         /* sub-carrier envelope, 56.9 us */
	c->burst_rise = 0.00000100;
	c->burst_left = 0.00000560;
	c->ev_co = -1.902 * 280e3;           /* D'R */
	c->eu_co =  1.505 * 230e3;           /* D'B */
}

/* ---- signal plans ---- */

static void _baseband(hvk_config_t *c, double white, double black, double blank, double sync)
{
	c->output_type = HVK_INT16_REAL;
	c->modulation = HVK_NONE;
	c->level = 1.0;
	c->video_level = 1.0;
	c->video_bw = 6.0e6;
	c->white_level = white;
	c->black_level = black;
	c->blanking_level = blank;
	c->sync_level = sync;
}

static void _vsb(hvk_config_t *c, double upper, double lower, double video_level,
                 double white, double black, double blank, double sync)
{
	c->output_type = HVK_INT16_COMPLEX;
	c->modulation = HVK_VSB;
	c->vsb_upper_bw = upper;
	c->vsb_lower_bw = lower;
	c->level = 1.0;
	c->video_level = video_level;
	c->white_level = white;
	c->black_level = black;
	c->blanking_level = blank;
	c->sync_level = sync;
}

static void _fm_video(hvk_config_t *c, double deviation, double white, double black, double blank, double sync)
{
	c->output_type = HVK_INT16_COMPLEX;
	c->modulation = HVK_FM;
	c->fm_level = 1.0;
	c->fm_deviation = deviation;         /* Hz per unit of signal */
	c->level = 1.0;
	c->video_level = 1.0;
	c->white_level = white;
	c->black_level = black;
	c->blanking_level = blank;
	c->sync_level = sync;
}

static void _fm_sound(hvk_config_t *c, double level, double carrier, double deviation, int preemph)
{
	c->fm_mono_level = level;
	c->fm_mono_carrier = carrier;
	c->fm_mono_deviation = deviation;
	c->fm_mono_preemph = preemph;
}

static void _nicam(hvk_config_t *c, double level, double carrier, double beta)
{
	c->nicam_level = level;
	c->nicam_carrier = carrier;
	c->nicam_beta = beta;
}

static const struct {
	const char *id;
	const char *desc;
} _modes[] = {
	{ "i",     "PAL colour, 25 fps, 625 lines, AM (complex), 6.0 MHz FM audio" },
	{ "b",     "PAL colour, 25 fps, 625 lines, AM (complex), 5.5 MHz FM audio" },
	{ "g",     "PAL colour, 25 fps, 625 lines, AM (complex), 5.5 MHz FM audio" },
	{ "pal",   "PAL colour, 25 fps, 625 lines, unmodulated (real)" },
	{ "l",     "SECAM colour, 25 fps, 625 lines, AM (complex), 6.5 MHz AM audio" },
	{ "secam", "SECAM colour, 25 fps, 625 lines, unmodulated (real)" },
	{ "m",     "NTSC colour, 30/1.001 fps, 525 lines, AM (complex), 4.5 MHz FM audio" },
	{ "ntsc",  "NTSC colour, 30/1.001 fps, 525 lines, unmodulated (real)" },
	{ "pal-fm",   "PAL colour, 25 fps, 625 lines, FM (complex), 6.5 MHz FM audio" },
	{ "secam-fm", "SECAM colour, 25 fps, 625 lines, FM (complex), 6.5 MHz FM audio" },
	{ "ntsc-fm",  "NTSC colour, 30/1.001 fps, 525 lines, FM (complex), 6.5 MHz FM audio" },
	{ "pal-d",    "PAL colour, 25 fps, 625 lines, AM (complex), 6.5 MHz FM audio" },
	{ "pal-k",    "PAL colour, 25 fps, 625 lines, AM (complex), 6.5 MHz FM audio" },
	{ "pal-m",    "PAL colour, 30/1.001 fps, 525 lines, AM (complex), 4.5 MHz FM audio" },
	{ "pal-n",    "PAL colour, 25 fps, 625 lines, AM (complex), 4.5 MHz FM audio" },
	{ "525pal",   "PAL colour, 30/1.001 fps, 525 lines, unmodulated (real)" },
	{ "d",        "SECAM colour, 25 fps, 625 lines, AM (complex), 6.5 MHz FM audio" },
	{ "k",        "SECAM colour, 25 fps, 625 lines, AM (complex), 6.5 MHz FM audio" },
	{ "secam-i",  "SECAM colour, 25 fps, 625 lines, AM (complex), 6.0 MHz FM audio" },
	{ "secam-b",  "SECAM colour, 25 fps, 625 lines, AM (complex), 5.5 MHz FM audio" },
	{ "secam-g",  "SECAM colour, 25 fps, 625 lines, AM (complex), 5.5 MHz FM audio" },
	{ "ntsc-i",   "NTSC colour, 30/1.001 fps, 525 lines, AM (complex), 6.0 MHz FM audio" },
	{ "pal60-i",  "PAL colour, 30/1.001 fps, 525 lines, AM (complex), 6.0 MHz FM audio" },
	{ "pal60",    "PAL colour, 30/1.001 fps, 525 lines, unmodulated (real)" },
	{ "e",             "No colour, 25 fps, 819 lines, AM (complex), 11.15 MHz AM audio" },
	{ "819",           "No colour, 25 fps, 819 lines, unmodulated (real)" },
	{ "a",             "No colour, 25 fps, 405 lines, AM (complex), -3.5 MHz AM audio" },
	{ "ntsc-a",        "NTSC colour, 25 fps, 405 lines, AM (complex), -3.5 MHz AM audio" },
	{ "405-i",         "No colour, 25 fps, 405 lines, AM (complex), 6.0 MHz FM audio" },
	{ "405",           "No colour, 25 fps, 405 lines, unmodulated (real)" },
	{ "ntsc-405",      "NTSC colour, 25 fps, 405 lines, unmodulated (real)" },
	{ "240-am",        "No colour, 25 fps, 240 lines, AM (complex)" },
	{ "240",           "No colour, 25 fps, 240 lines, unmodulated (real)" },
	{ "30-am",         "No colour, 12.5 fps, 30 lines, AM (complex)" },
	{ "30",            "No colour, 12.5 fps, 30 lines, unmodulated (real)" },
	{ "nbtv-am",       "No colour, 12.5 fps, 32 lines, AM (complex)" },
	{ "nbtv",          "No colour, 12.5 fps, 32 lines, unmodulated (real)" },
	{ "apollo-fsc-fm", "Field sequential colour, 30/1.001 fps, 525 lines, FM (complex), 1.25 MHz FM audio" },
	{ "apollo-fsc",    "Field sequential colour, 30/1.001 fps, 525 lines, unmodulated (real)" },
	{ "apollo-fm",     "No colour, 10 fps, 320 lines, FM (complex), 1.25 MHz FM audio" },
	{ "apollo",        "No colour, 10 fps, 320 lines, unmodulated (real)" },
	{ "m-cbs405",      "Field sequential colour, 72 fps, 405 lines, VSB (complex), 4.5MHz FM audio" },
	{ "cbs405",        "Field sequential colour, 72 fps, 405 lines, unmodulated (real)" },
	{ NULL, NULL },
};

const char *hvk_preset_id(int index)
{
	if(index < 0 || index >= (int) (sizeof(_modes) / sizeof(_modes[0])) - 1) return(NULL);
	return(_modes[index].id);
}

const char *hvk_preset_desc(int index)
{
	if(index < 0 || index >= (int) (sizeof(_modes) / sizeof(_modes[0])) - 1) return(NULL);
	return(_modes[index].desc);
}

int hvk_config_preset(hvk_config_t *c, const char *id)
{
	if(c == NULL || id == NULL) return(HVK_ERROR);

	memset(c, 0, sizeof(*c));
	c->struct_size = (uint32_t) sizeof(*c);
	c->volume = 256; /* src/hacktv.c:1431 with the default --volume 1.0 */

	if(strcmp(id, "i") == 0)
	{
		/* src/video.c:50-102 */
		_vsb(c, 5500000, 1250000, 0.71, 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000025);
		_colour_pal(c);
		_fm_sound(c, 0.22, 6000000 - 400, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 6552000, 1.0);
	}
	else if(strcmp(id, "b") == 0 || strcmp(id, "g") == 0)
	{
		/* src/video.c:104-156 */
		_vsb(c, 5000000, 750000, 0.71, 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000020);
		_colour_pal(c);
		_fm_sound(c, 0.15, 5500000, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 5850000, 0.4);
	}
	else if(strcmp(id, "pal") == 0)
	{
		/* src/video.c:274-314 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		_raster_625(c, 0.00000020);
		_colour_pal(c);
	}
	else if(strcmp(id, "l") == 0)
	{
		/* src/video.c:457-504 */
		_vsb(c, 6000000, 1250000, 0.80 * (100.0 / 124.0), 1.00, 0.30, 0.30, 0.05);
		_raster_625(c, 0.00000020);
		_colour_secam(c);
		c->am_audio_level = 0.10;
		c->am_mono_carrier = 6500000;
		_nicam(c, 0.04, 5850000, 0.4);
	}
	else if(strcmp(id, "secam") == 0)
	{
		/* src/video.c:716-753 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		_raster_625(c, 0.00000020);
		_colour_secam(c);
	}
	else if(strcmp(id, "m") == 0)
	{
		/* src/video.c:755-803 */
		_vsb(c, 4200000, 750000, 0.77, 0.125000, 0.703125, 0.750000, 1.000000);
		_raster_525(c);
		_colour_ntsc(c);
		_fm_sound(c, 0.15, 4500000, 25000, HVK_75US);
	}
	else if(strcmp(id, "ntsc") == 0)
	{
		/* src/video.c:968-1008 */
		_baseband(c, 100.0 / 140, 7.5 / 140, 0.0 / 140, -40.0 / 140);
		_raster_525(c);
		_colour_ntsc(c);
	}
	else if(strcmp(id, "pal-fm") == 0)
	{
		/* src/video.c:213-272 (satellite FM, 16 MHz/V) */
		_fm_video(c, 16e6, 0.50, -0.20, -0.20, -0.50);
		_raster_625(c, 0.00000020);
		_colour_pal(c);
		_fm_sound(c, 0.06, 6500000, 85000, HVK_50US);
	}
	else if(strcmp(id, "secam-fm") == 0)
	{
		/* src/video.c:659-714 */
		_fm_video(c, 16e6, 0.50, -0.20, -0.20, -0.50);
		_raster_625(c, 0.00000020);
		_colour_secam(c);
		_fm_sound(c, 0.05, 6500000, 85000, HVK_50US);
	}
	else if(strcmp(id, "ntsc-fm") == 0)
	{
		/* src/video.c:859-918 */
		_fm_video(c, 16e6, 0.5000, -0.1607, -0.2143, -0.5000);
		_raster_525(c);
		_colour_ntsc(c);
		_fm_sound(c, 0.05, 6500000, 85000, HVK_50US);
	}
	else if(strcmp(id, "pal-d") == 0 || strcmp(id, "pal-k") == 0)
	{
		/* src/video.c:158-211 */
		_vsb(c, 5500000, 750000, 0.70, 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000020);
		_colour_pal(c);
		_fm_sound(c, 0.20, 6500000, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 5850000, 0.4);
	}
	else if(strcmp(id, "pal-m") == 0)
	{
		/* src/video.c:316-364 */
		_vsb(c, 4200000, 750000, 0.77, 0.2000, 0.7280, 0.7712, 1.0000);
		_raster_525w(c, 0.00005280, 0.00000020);
		_colour_pal_mn(c, 511312500, 143);        /* 3575611.888... Hz */
		_fm_sound(c, 0.15, 4500000, 25000, HVK_75US);
	}
	else if(strcmp(id, "pal-n") == 0)
	{
		/* src/video.c:366-413 (no sync rise time in the preset) */
		_vsb(c, 4200000, 750000, 0.77, 0.2000, 0.7280, 0.7712, 1.0000);
		_raster_625(c, 0);
		_colour_pal_mn(c, 14328225, 4);           /* 3582056.25 Hz */
		_fm_sound(c, 0.15, 4500000, 25000, HVK_75US);
	}
	else if(strcmp(id, "525pal") == 0)
	{
		/* src/video.c:415-455 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		_raster_525w(c, 0.00005280, 0.00000020);
		_colour_pal_mn(c, 511312500, 143);
	}
	else if(strcmp(id, "d") == 0 || strcmp(id, "k") == 0)
	{
		/* src/video.c:506-555 */
		_vsb(c, 5500000, 750000, 0.70, 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000020);
		_colour_secam(c);
		_fm_sound(c, 0.20, 6500000, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 5850000, 0.4);
	}
	else if(strcmp(id, "secam-i") == 0)
	{
		/* src/video.c:557-606 */
		_vsb(c, 5500000, 1250000, 0.71, 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000025);
		_colour_secam(c);
		_fm_sound(c, 0.15, 6000000 - 400, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 6552000, 1.0);
	}
	else if(strcmp(id, "secam-b") == 0 || strcmp(id, "secam-g") == 0)
	{
		/* src/video.c:608-657 */
		_vsb(c, 5000000, 750000, 0.80 * (100.0 / 124.0), 0.20, 0.76, 0.76, 1.00);
		_raster_625(c, 0.00000020);
		_colour_secam(c);
		_fm_sound(c, 0.15, 5500000, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 5850000, 0.4);
	}
	else if(strcmp(id, "ntsc-i") == 0)
	{
		/* src/video.c:805-857 */
		_vsb(c, 5500000, 1250000, 0.71, 0.200000, 0.728571, 0.771428, 1.000000);
		_raster_525(c);
		_colour_ntsc(c);
		_fm_sound(c, 0.22, 6000000 - 400, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 6552000, 1.0);
	}
	else if(strcmp(id, "pal60-i") == 0)
	{
		/* src/video.c:1010-1062 */
		_vsb(c, 5500000, 1250000, 0.71, 0.20, 0.76, 0.76, 1.00);
		_raster_525(c);
		_colour_pal(c);
		_fm_sound(c, 0.22, 6000000 - 400, 50000, HVK_50US);
		_nicam(c, 0.07 / 2, 6552000, 1.0);
	}
	else if(strcmp(id, "pal60") == 0)
	{
		/* src/video.c:1064-1103 (no sync rise time in the preset) */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		_raster_525w(c, 0.00005290, 0);
		_colour_pal(c);
	}
	else if(strcmp(id, "e") == 0)
	{
		/* src/video.c:1303-1337: System E, 819 lines; the sound is AM, 11.15 MHz above the vision carrier */
		_vsb(c, 2000000, 10400000, 0.8, 1.00, 0.35, 0.30, 0.00);
		_raster_819(c);
		c->am_audio_level = 0.2;
		c->am_mono_carrier = 11.15e6;
	}
	else if(strcmp(id, "819") == 0)
	{
		/* src/video.c:1339-1366 */
		_baseband(c, 0.70, 0.05, 0.00, -0.30);
		c->video_bw = 10.4e6;
		_raster_819(c);
	}
	else if(strcmp(id, "a") == 0)
	{
		/* src/video.c:1368-1403: System A, 405 lines; AM sound 3.5 MHz BELOW the vision carrier */
		_vsb(c, 750000, 3000000, 0.8, 1.00, 0.30, 0.30, 0.00);
		_raster_405(c);
		c->am_audio_level = 0.2;
		c->am_mono_carrier = -3500000;
	}
	else if(strcmp(id, "ntsc-a") == 0)
	{
		/* src/video.c:1405-1451 (video level reduced for NTSC's 122 % overshoot; no chroma low pass in the preset) */
		_vsb(c, 750000, 3000000, 0.80 / 1.22, 1.00, 0.35, 0.30, 0.00);
		_raster_405(c);
		_colour_ntsc405(c, 0);
		c->am_audio_level = 0.20;
		c->am_mono_carrier = -3500000;
	}
	else if(strcmp(id, "405-i") == 0)
	{
		/* src/video.c:1453-1489 */
		_vsb(c, 5500000, 1250000, 0.80, 0.20, 0.76, 0.76, 1.00);
		_raster_405(c);
		_fm_sound(c, 0.19, 6000000 - 400, 50000, HVK_50US);
	}
	else if(strcmp(id, "405") == 0)
	{
		/* src/video.c:1491-1519 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		c->video_bw = 3.0e6;
		_raster_405(c);
	}
	else if(strcmp(id, "ntsc-405") == 0)
	{
		/* src/video.c:1521-1561 */
		_baseband(c, 0.70, 0.05, 0.00, -0.30);
		c->video_bw = 3.0e6;
		_raster_405(c);
		_colour_ntsc405(c, 1.1e6);
	}
	else if(strcmp(id, "240-am") == 0)
	{
		/* src/video.c:1563-1589 */
		_am(c, 1.00, 0.40, 0.40, 0.00);
		_raster_240(c);
	}
	else if(strcmp(id, "240") == 0)
	{
		/* src/video.c:1591-1615 (no video bandwidth in the preset) */
		_baseband(c, 1.00, 0.40, 0.40, 0.00);
		c->video_bw = 0;
		_raster_240(c);
	}
	else if(strcmp(id, "30-am") == 0)
	{
		/* src/video.c:1617-1641: Baird 30 lines, scanned vertically, no sync pulses */
		_am(c, 1.00, 0.00, 0.00, 0.00);
		_raster_30(c);
	}
	else if(strcmp(id, "30") == 0)
	{
		/* src/video.c:1643-1665 */
		_baseband(c, 1.00, -1.00, -1.00, -1.00);
		c->video_bw = 0;
		_raster_30(c);
	}
	else if(strcmp(id, "nbtv-am") == 0)
	{
		/* src/video.c:1667-1693: NBTV Club standard, negative modulation */
		_am(c, 0.10, 0.73, 0.73, 1.00);
		_raster_nbtv(c);
	}
	else if(strcmp(id, "nbtv") == 0)
	{
		/* src/video.c:1695-1719 */
		_baseband(c, 1.00, 0.30, 0.30, 0.00);
		c->video_bw = 0;
		_raster_nbtv(c);
	}
	else if(strcmp(id, "apollo-fsc-fm") == 0)
	{
		/* src/video.c:1721-1767: Unified S-Band, Apollo colour: 525 lines, one colour channel per field */
		_fm_video(c, 2e6, 0.5000, -0.1475, -0.2000, -0.5000);
		_raster_525(c);
		_fsc(c, HVK_APOLLO_FSC, 0.00002000, 0.00001470, 0.5000);
		_fm_sound(c, 0.150, 1250000, 25000, 0);
	}
	else if(strcmp(id, "apollo-fsc") == 0)
	{
		/* src/video.c:1769-1801 */
		_baseband(c, 0.70, 0.0525, 0.00, -0.30);
		c->video_bw = 0;
		_raster_525(c);
		_fsc(c, HVK_APOLLO_FSC, 0.00002000, 0.00001470, 0.70);
	}
	else if(strcmp(id, "apollo-fm") == 0)
	{
		/* src/video.c:1803-1845: Apollo 10 fps, 320 lines */
		_fm_video(c, 2e6, 0.50, -0.20, -0.20, -0.50);
		_raster_apollo(c);
		_fm_sound(c, 0.150, 1250000, 25000, 0);
	}
	else if(strcmp(id, "apollo") == 0)
	{
		/* src/video.c:1847-1880 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		c->video_bw = 0;
		_raster_apollo(c);
	}
	else if(strcmp(id, "m-cbs405") == 0)
	{
		/* src/video.c:1882-1923: CBS field-sequential colour, 405 lines at 72 frames a second */
		_vsb(c, 4200000, 750000, 0.77, 0.159, 0.595, 0.595, 1.000);
		_raster_cbs(c);
		_fsc(c, HVK_CBS_FSC, 0.000001372, 0.000008573, 1.000);
		_fm_sound(c, 0.15, 4500000, 25000, HVK_75US);
	}
	else if(strcmp(id, "cbs405") == 0)
	{
		/* src/video.c:1925-1954 */
		_baseband(c, 0.70, 0.00, 0.00, -0.30);
		c->video_bw = 0;
		_raster_cbs(c);
		_fsc(c, HVK_CBS_FSC, 0.000001372, 0.000008573, -0.30);
	}
	else
	{
		return(HVK_ERROR);
	}

	return(HVK_OK);
}

/* The command-line switches that edit a preset before vid_init()
 * (src/hacktv.c:1126-1171, :1412-1415) */
void hvk_config_apply_flags(hvk_config_t *c, int flags)
{
	if(flags & HVK_FLAG_NOCOLOUR)
	{
		if(c->colour_mode == HVK_PAL || c->colour_mode == HVK_SECAM || c->colour_mode == HVK_NTSC)
		{
			c->colour_mode = HVK_MONOCHROME;
		}
	}

	if(flags & HVK_FLAG_NOAUDIO)
	{
		c->fm_mono_level = c->am_audio_level = c->nicam_level = 0;
		c->fm_mono_carrier = c->nicam_carrier = c->am_mono_carrier = 0;
	}

	if(flags & HVK_FLAG_NONICAM)
	{
		c->nicam_level = 0;
		c->nicam_carrier = 0;
	}

	if(flags & HVK_FLAG_FILTER) c->vfilter = 1;
}
