/* oracle/oracle_internal.h -- TEST INFRASTRUCTURE (not product code). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include "oracle_video.h"

typedef struct { int16_t i, q; } c16_t;
typedef struct { int32_t i, q; } c32_t;

/* One pre-shaped pulse: `length` values that are ADDED to the line starting
 * at sample `offset` (may be negative: spills into the previous line) */
typedef struct {
	int offset;
	int length;
	int16_t *value;
} orc_pulse_t;

/* 32 kHz soft limiter state (src/fir.c:758-870) */
typedef struct {
	int width;
	int16_t level;
	int16_t *shape;
	int16_t *att;
	int32_t *fix;
	int32_t *var;
	int p, h;
	/* the two 65-tap int32 FIRs in front of it (src/fir.c:620-694) */
	int ntaps;
	int32_t *vtaps, *ftaps;
	int32_t *vwin, *fwin;
	int vpos, fpos;
} orc_limiter_t;

/* Phasor modulators (src/video.h:91-115) */
typedef struct {
	int on;
	int16_t level;
	int32_t counter;
	c32_t phase;
	c32_t *lut;       /* FM: 65536 steps indexed by sample + 32768 */
	c32_t delta;      /* AM: constant step */
	int16_t sample;
	orc_limiter_t lim;
	int has_lim;
} orc_mod_t;

/* NICAM-728 encoder + DQPSK modulator (src/nicam728.h:52-102) */
typedef struct {
	int on;
	uint8_t mode, reserve;
	unsigned int frame_no;
	uint8_t prn[90];
	int fir_p;
	int16_t fir_l[83], fir_r[83];
	int16_t audio[64];
	int ntaps;
	int16_t *taps;
	int dsym;
	c16_t *bb;
	int bb_pos, bb_len;
	int sps, ds, dsl, decimation;
	c16_t *cc;
	int cc_len, cc_pos;
	uint8_t frame[91];
	int frame_bit;
} orc_nicam_t;

struct orc_t {
	hvk_config_t conf;
	int sample_rate;
	int pixel_rate;

	/* geometry (src/video.c:3844-3853) */
	int width, half_width, active_width, active_left;

	/* levels (src/video.c:3878-3881) */
	int16_t white_level, black_level, blanking_level, sync_level;

	/* the five sync pulses: h, v, V, mid-v, mid-V (src/video.c:3885-3891) */
	orc_pulse_t sync[5];
	orc_pulse_t fsc[2];         /* field-sequential colour flag pulses (src/video.c:4050-4073) */
	int olines;                 /* line buffers in the reference's ring (src/video.c:3578): a buffer has width 0 until the raster has used it once */
	int16_t *sync_packed;       /* the reference's packed vbidata table, for comparison */
	long sync_packed_len;

	int16_t *yuv;               /* 0x1000000 x {y,u,v} */

	unsigned int colour_lookup_width;
	unsigned int colour_lookup_offset;
	c16_t *colour_lookup;

	c16_t burst_phase;
	int burst_left, burst_width;
	int16_t *burst_win;

	int chroma_ntaps;
	int16_t *chroma_taps;
	int16_t ghost[32];
	int16_t *chroma;            /* 2*width + slack */

	/* video filter (src/video.c:3653-3764) */
	int vf_type;                /* 0 none, 1 real->real, 3 real->complex */
	int vf_ntaps;
	int16_t *vf_itaps, *vf_qtaps;
	int delay_lines;
	int vf_delay;               /* extra samples of window: _calc_filter_delay (src/video.c:3620-3625) */
	int16_t *vf_win;            /* the last vf_ntaps + vf_delay input samples, oldest first */

	/* --pixelrate resampler (src/video.c:3627-3651, src/fir.c:393-428) */
	int rs_L, rs_D, rs_ataps, rs_d;
	int16_t *rs_taps;           /* [L][ataps], the order they are applied in; NULL: no resampler */
	int16_t *rs_win;
	int16_t *rs_win2; int rs_d2;    /* the resampler's second channel (--s-video) */
	int16_t *prev_q;            /* its last chunks */
	int max_width;              /* widest chunk the pipeline can emit */

	/* pipeline state (oracle_video.c) */
	long chunks_done;
	int16_t *cbuf, *ciq, *ccar, *prev_r;
	int *prev_w;
	int *last_widths; long last_nwidths;

	/* current source frame */
	const uint32_t *fb;
	int fb_width, fb_height, fb_pixel_stride, fb_line_stride, fb_interlaced;
	long long fb_par_num, fb_par_den;   /* pixel aspect of the source frame, 0: 1:1 */
	/* --raw-bb-file: the external baseband stream (looped like the reference's rewind at end of file) */
	const int16_t *rawbb; long rawbb_len;

	/* --interlace: the frame shown by each field; the fields above are set from these per line */
	struct { const uint32_t *fb; int width, height, pixel_stride, line_stride, interlaced; } field_fb[2];

	/* raster stream window: lines [s_first, s_first + s_count) */
	int16_t *S;
	int16_t *C;                 /* --s-video: the Q channel of the same lines (the colour sub-carrier) */
	long s_first, s_count, s_cap;
	long rastered;              /* number of lines rastered so far (next g) */
	long emitted;               /* number of lines emitted so far */

	/* audio-rate state */
	int interp;
	const int16_t *audio_src;
	long audio_len, audio_pos;
	int audio_loop;
	orc_mod_t fm_mono;
	orc_mod_t fm_right;         /* A2 stereo: the second sound carrier */
	orc_mod_t a2_pilot, a2_signal;
	int a2_system_m;
	orc_mod_t am_mono;
	orc_nicam_t nicam;
	int16_t nicam_buf[64];
	int nicam_buf_len;
	int audio_primed;

	/* FM video, offset, passthru (oracle_tail.c) */
	orc_mod_t fm_video;
	c32_t offset_phase, offset_delta;
	int32_t offset_counter;
	const int16_t *pass_src;
	long pass_len, pass_pos;
	int pass_eof;
	int16_t *passline;

	/* sound-in-syncs (oracle_sis.c) */
	orc_pulse_t *sis_lut;       /* 50 half symbols */
	int16_t *sis_packed;        /* ... as the reference packs them */
	long sis_pos[50];           /* where an entry's values start in it */
	int16_t sis_heap[8];        /* the 16 bytes in front of the reference's table on its heap */
	int sis_blank_left, sis_blank_width;
	int16_t *sis_blank_win;
	orc_nicam_t sis_nicam;
	uint8_t sis_frame[91];
	int sis_frame_bit, sis_re;
	long sis_calls;
	uint8_t *sis_rec; long sis_rec_n, sis_rec_cap;     /* the bursts of the lines made so far, 8 bytes each */
	int sis_visible;            /* samples of the step's audio line the audio thread is taken to have behind it (0: none) */

	/* SECAM colour process (oracle_secam.c) */
	int16_t sc_level;
	c32_t *sc_lut;
	double sc_a1, sc_b0, sc_b1, sc_ix, sc_iy;
	int16_t *sc_fir, *sc_notch;
	int16_t sc_dmin[2], sc_dmax[2];
	c16_t *sc_bell;
	long sc_done;               /* lines the process has been applied to */
	int16_t sc_fsync_level;
	int sc_fid_lines;
	int sc_fill_slots;          /* fill slots (line 0) the colour process has seen */

	/* teletext render (oracle_teletext.c): symbol table and the packets queued per frame */
	orc_pulse_t *tt_sym;
	struct { long frame; uint32_t mask; uint8_t packets[32][45]; } tt_queue[16];

	/* VBI inserters (oracle_vbi.c) */
	int16_t *vits_line[4];
	c16_t vits_phase;
	orc_pulse_t *wss_lut, *vitc_lut;
	uint8_t wss_vbi[18];
	int wss_blank_width;
	int vitc_lines[2], vitc_hr, vitc_fps, vitc_drop;
	int acp_left[6], acp_psync_width, acp_pagc_width;
	int16_t acp_psync_level, acp_pagc_level;
	orc_pulse_t *cc_lut;
	int16_t *cc_cri;
	int cc_cri_x, cc_cri_len, cc_line;
	long cc_frame; uint8_t cc_pair[2];

	/* stage taps of the last render call */
	int16_t *last_raster; long last_raster_len;
	int16_t *last_carrier; long last_carrier_len;
};

/* oracle_tables.c */
int orc_build_tables(orc_t *s);
void orc_free_tables(orc_t *s);
double orc_rc_window(double t, double left, double width, double rise);

/* oracle_raster.c */
void orc_raster_line(orc_t *s, long g);
int16_t *orc_line_ptr(orc_t *s, long g);
int16_t *orc_cline_ptr(orc_t *s, long g);    /* the line's Q channel (--s-video) */

void orc_line_info(orc_t *s, long g, int *frame, int *line, int *la, int *ra, int *vy);

/* the source frame in force on a line: with --interlace the second field has its own (src/video.c:4873) */
void orc_select_frame(orc_t *s, int line);

/* oracle_secam.c */
int orc_secam_init(orc_t *s);
void orc_secam_free(orc_t *s);
/* oq: the line's Q channel, where the sub-carrier goes with --s-video (may be NULL) */
void orc_secam_line(orc_t *s, int16_t *o, int16_t *oq, int frame, int line, int active_l, int active_r, int vy);

/* oracle_teletext.c */
int orc_teletext_init(orc_t *s);
void orc_teletext_free(orc_t *s);
void orc_teletext_render(orc_t *s, int16_t *o, const uint8_t packet[45]);

/* oracle_vbi.c */
int orc_vbi_init(orc_t *s);
void orc_vbi_free(orc_t *s);
void orc_vbi_line(orc_t *s, long g, int frame, int line, const c16_t *lut);
int orc_vbi_allocated(orc_t *s, int line);
int orc_vbi_allocated_by_vits(orc_t *s, int line);
int orc_vbi_allocated_by_secam(orc_t *s, int line);

/* oracle_tail.c */
int orc_tail_init(orc_t *s);
void orc_tail_free(orc_t *s);
void orc_tail_line(orc_t *s, int16_t *iq, int width);

/* oracle_sis.c */
int orc_sis_init(orc_t *s);
void orc_sis_free(orc_t *s);
void orc_sis_line(orc_t *s, long g, int first_line);

/* oracle_audio.c */
/* a NICAM-728 frame encoder on its own (src/nicam728.c:96-126, :195-249): n->audio in, n->frame out */
void orc_nicam_encoder_init(orc_nicam_t *n, uint8_t mode, uint8_t reserve);
void orc_nicam_encode(orc_nicam_t *n);
int orc_audio_init(orc_t *s);
void orc_audio_free(orc_t *s);
void orc_audio_line(orc_t *s, int16_t *iq, int width, int16_t *carrier_tap);

#endif
