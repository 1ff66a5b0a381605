/* hvk_config.h -- mode configuration for the composite-video -> IQ engine.
 *
 * Plain data, no behaviour. `hvk_config_t` carries the subset of the
 * reference's `vid_config_t` (src/video.h:125-292) that the hot path reads:
 * raster geometry, levels, colour system, VSB/low-pass filter and audio
 * sub-carrier parameters. Field names and units are the reference's, so a
 * caller that holds a `vid_config_t` can fill this member by member (the
 * video.h-compatible shim does exactly that).
 *
 * Everything the reference's engine supports but this engine does not render
 * (scramblers, MAC, sound-in-syncs, DANCE, separate L/R FM sub-carriers,
 * FM energy dispersal, raw baseband input) has no field here; the shim
 * refuses such configurations rather than silently dropping them.
 */
#ifndef HVK_CONFIG_H
#define HVK_CONFIG_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Output sample type: src/rf.h:27-28 (RF_INT16_COMPLEX / RF_INT16_REAL).
 * The engine always produces interleaved I/Q pairs (src/video.h:306-311);
 * REAL only tells the sink that Q carries nothing. */
#define HVK_INT16_COMPLEX 0
#define HVK_INT16_REAL    1

/* Raster type: src/video.h:50-59 */
#define HVK_RASTER_625 0
#define HVK_RASTER_525 1
#define HVK_RASTER_405 2
#define HVK_RASTER_819 3
#define HVK_BAIRD_240  4
#define HVK_BAIRD_30   5
#define HVK_NBTV_32    6
#define HVK_APOLLO_320 7
#define HVK_CBS_405    9

/* How the caller turns a source picture before it hands it over (src/video.h:62-67, src/video.c:4883-4885): the mechanical
 * systems scan vertically. The engine shows what it is given; the video.h shim applies this to every frame it pulls. */
#define HVK_ROTATE_0   0
#define HVK_ROTATE_90  1
#define HVK_ROTATE_180 2
#define HVK_ROTATE_270 3
#define HVK_HFLIP      4
#define HVK_VFLIP      8

/* Output modulation: src/video.h:70-73 */
#define HVK_NONE 0
#define HVK_AM   1
#define HVK_VSB  2
#define HVK_FM   3 /* the composite is rendered on the device; the FM phasor itself is a serial host pass */

/* Colour modes: src/video.h:76-81 */
#define HVK_MONOCHROME 0
#define HVK_PAL        1
#define HVK_NTSC       2
#define HVK_SECAM      3
#define HVK_APOLLO_FSC 4 /* field-sequential colour: one of the picture's colour channels per field, as luma */
#define HVK_CBS_FSC    5

/* Audio pre-emphasis: src/video.h:85-87 */
#define HVK_50US 1
#define HVK_75US 2
#define HVK_J17  3

typedef struct {
	int64_t num;
	int64_t den;
} hvk_rational_t; /* r64_t, src/common.h:31-34 */

typedef struct hvk_config_t {

	uint32_t struct_size;       /* sizeof(hvk_config_t) of the header the CALLER was built with: hvk_config_preset() fills it
	                             * in, a caller that fills the struct member by member sets it itself (HVK_CONFIG_INIT).
	                             * hvk_open() / hvk_open_rates() / hvk_group_open() return HVK_ERROR on any other value -- an
	                             * embedder built against another release's layout is told so instead of being misread */

	int output_type;            /* HVK_INT16_COMPLEX | HVK_INT16_REAL */

	int modulation;             /* HVK_NONE | HVK_AM | HVK_VSB | HVK_FM */
	double video_bw;            /* Hz, low-pass cut-off for AM / baseband */
	double vsb_upper_bw;        /* Hz */
	double vsb_lower_bw;        /* Hz */

	double level;               /* overall signal level */
	double video_level;
	double fm_mono_level;
	double am_audio_level;
	double nicam_level;

	int type;                   /* HVK_RASTER_625 | _525 | _405 | _819, HVK_BAIRD_240 | _30, HVK_NBTV_32, HVK_APOLLO_320, HVK_CBS_405 */
	hvk_rational_t frame_rate;
	int lines;
	int hline;                  /* 0 = derive, src/video.c:3832 */
	int interlaced;             /* 0 none, 1 TFF, 2 BFF */
	int interlace;              /* --interlace (src/video.c:4873): a new source frame is taken at the start of EACH
	                             * field; render calls then name two frame slots per frame (first field, second field) */
	int active_lines;

	double hsync_width;         /* seconds */
	double vsync_short_width;
	double vsync_long_width;
	double sync_rise;           /* 10%-90% */

	int invert_video;
	double white_level;
	double black_level;
	double blanking_level;
	double sync_level;

	double active_width;        /* seconds */
	double active_left;

	double gamma;               /* <= 0 means 1.0 */
	double rw_co, gw_co, bw_co; /* <= 0 means 0.299 / 0.587 / 0.114 */

	int colour_mode;
	hvk_rational_t colour_carrier; /* Hz */
	double colour_bw;           /* Hz, gaussian chroma low-pass, 0 = none */

	double burst_width;         /* seconds */
	double burst_left;
	double burst_level;
	double burst_rise;

	double ev_co;
	double eu_co;

	int secam_field_id;
	int secam_field_id_lines;

	int volume;                 /* 256 = unity, src/hacktv.c:1431 */

	double fm_mono_carrier;     /* Hz */
	double fm_mono_deviation;   /* +/- Hz */
	int fm_mono_preemph;        /* HVK_50US | HVK_75US | HVK_J17 | 0 */

	double nicam_carrier;       /* Hz */
	double nicam_beta;

	double am_mono_carrier;     /* Hz */

	int a2stereo;               /* --a2stereo (Zweikanalton): a second FM carrier 242.1875 kHz (224.213 kHz on
	                             * system M) above the first with the right channel (L - R on M) and the
	                             * 54.6875 kHz pilot; replaces NICAM (src/video.c:4375-4400, :3404-3424) */

	int vfilter;                /* --filter, src/hacktv.c:1412 */

	int raw_bb;                 /* --raw-bb-file: the raster is not built from frames but taken from an external int16
	                             * baseband stream, hvk_rawbb_write() (src/video.c:2406-2446); the inserters, filter,
	                             * sound and modulators still apply */
	int raw_bb_blanking_level;  /* the stream's blanking and white levels: mapped onto the mode's */
	int raw_bb_white_level;

	int s_video;                /* --s-video (baseband PAL / NTSC / SECAM only, src/hacktv.c:1136-1148): the colour
	                             * sub-carrier goes to the Q channel instead of onto the luma (src/video.c:3032, :3219) */

	int teletext;               /* != 0: teletext packets will be supplied for the VBI lines
	                             * (625-line modes; the reference's conf.teletext names the page
	                             * source, which stays with the caller: hvk_teletext_packets()) */

	/* VBI inserters whose data are a function of the configuration and the frame number */
	int wss;                    /* widescreen signalling on line 23 (625 lines): 0 none, else the reference's
	                             * mode byte, src/wss.c:33-44 -- 0x08 4:3, 0x01 14:9-letterbox, 0x02 14:9-top,
	                             * 0x0B 16:9-letterbox, 0x04 16:9-top, 0x0D 16:9+-letterbox, 0x0E 14:9-window,
	                             * 0x07 16:9; 0xFF "auto": 4:3 or 16:9 from the pixel aspect of the frame shown
	                             * (hvk_frame_aspect()) */
	int vits;                   /* --vits: insertion test signals, lines 17/18/330/331 (625) or 17/280 (525) */
	int vitc;                   /* --vitc: vertical interval time code, lines 19/21/332/334 (625) or 14/16/277/279 (525) */
	int acp;                    /* --acp: P-sync / AGC pulse pairs on lines 9-18, 321-330 (625) or 12-19, 275-282 (525) */
	int cc608;                  /* --cc608: CEA/EIA-608 caption line 22 (625) / 21 (525); the byte pairs are supplied
	                             * per frame with hvk_cc608_write(), zeros otherwise */
	int sis;                    /* --sis dcsis: sound-in-syncs -- a NICAM-728 stream of its own as 4-level symbols inside every
	                             * line's sync pulse (src/sis.c). 0 none, 1 "dcsis". The reference hands the 32-sample audio
	                             * blocks from its audio thread to this process without a lock (src/sis.c:217-221,
	                             * src/video.c:3370-3373); the engine's reading: a NICAM frame carries the newest block handed
	                             * over in an EARLIER step of the line pipeline -- what the reference CLI does on its test
	                             * source */

	/* FM video (modulation == HVK_FM), src/video.h:141-142 */
	double fm_level;
	double fm_deviation;        /* Hz per unit of signal */

	/* The complex-output tail of the line pipeline (src/video.c:4589-4645), applied in this order */
	int swap_iq;                /* --swap-iq */
	int64_t offset;             /* --offset: frequency shift in Hz, 0 = none */
	int passthru;               /* != 0: an int16 I/Q stream is added to the output; the samples are
	                             * supplied with hvk_passthru_write() (the reference's conf.passthru
	                             * names the file, which stays with the caller) */

	/* (members added after the first release stand at the END: an embedder built against an earlier header hands over a
	 * struct whose members up to here lie where they lay) */

	/* Field-sequential colour (colour_mode HVK_APOLLO_FSC / HVK_CBS_FSC, src/video.c:2919-2930, :3043-3063): the flag
	 * pulse that marks one of the three fields of the colour sequence, src/video.h:236-238 */
	double fsc_flag_width;      /* seconds */
	double fsc_flag_left;
	double fsc_flag_level;

	int frame_orientation;      /* HVK_ROTATE_* | HVK_HFLIP | HVK_VFLIP, src/video.h:62-67 */

} hvk_config_t;

/* all members zero, struct_size set: the starting point of a configuration filled member by member */
#define HVK_CONFIG_INIT(c) do { memset((c), 0, sizeof(hvk_config_t)); (c)->struct_size = (uint32_t) sizeof(hvk_config_t); } while(0)

#ifdef __cplusplus
}
#endif

#endif
