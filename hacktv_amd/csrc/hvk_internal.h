/* hvk_internal.h -- internal types shared by the host side (C) and the HIP
 * side (C++) of libhvk. Everything the kernels read is plain data laid out
 * for the device; the host builds it once in hvk_open(). */
#ifndef HVK_INTERNAL_H
#define HVK_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include "hacktv_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HVK_MAX_PULSES   8
#define HVK_GHOST_LEN    32
#define HVK_SPL          8      /* samples per lane in both kernels */
#define HVK_TILE         1024   /* samples per filter workgroup */
#define HVK_MAX_VF_TAPS  72     /* (51 for the designed filters; the FM pre-emphasis tables have 67 and 71) */
#define HVK_PULSE_PAD    8      /* zero int16 either side of every sync pulse in the flat value table */
#define HVK_NICAM_LEAD   8      /* zero dwords in front of the duplicated NICAM pulse table */
#ifndef HVK_NICAM_BACK
/* Symbols that can lie over a lane's 8 samples. A symbol started at st reaches sample n while 0 <= n - st < ntaps, so a lane's samples
 * [x0, x0 + 7] see the starts in (x0 - ntaps, x0 + 7]: ntaps + 7 positions. Starts are sps or sps - 1 apart (sps = ceil(rate / 364000),
 * src/nicam728.c:302-304, :398-407), ntaps <= 5 rate / 364000 + 2 (:268): at most floor((ntaps + 6) / (sps - 1)) + 1 = 6 of them for
 * every rate above 5.1 MHz (16 MHz: 227 / 43 -> 5 + 1; 10 MHz: 145 / 27; 36 MHz: 501 / 98) -- NICAM is taken from 10 MHz up. Rounds 1-5
 * walk 7: the seventh always reads the pulse table's zero tail. 6 is exact (every parity gate) and was measured in round 6 -- 14 vector
 * instructions and two LDS reads fewer a lane of 401 and 33 -- at 0.2027 / 0.2039 / 0.2020 ms against 0.2013 / 0.2023 / 0.2017 with 7 on
 * one box (profiles/r06_direct_persistent_pipelined_experiment.txt): the launch does not follow its instruction count. 7 stays. */
#define HVK_NICAM_BACK   7
#endif
#define HVK_VBI_OPS      64     /* VBI lines per frame (32 teletext + WSS + 4 VITC + CC608 + 20 ACP + spare) */
#define HVK_VBI_OPWORDS  16     /* dwords per op: sym_base, nbits, blank range, spare, 12 data words */
#define HVK_SIS_SPAN     256    /* samples at a line's start the sound-in-syncs burst and its window lie in */
#define HVK_VBI_COVER    30     /* symbols that may lie over one sample in a table's cover list (hvk_engine.cpp: the data lines as a gather); a sample's entry: 16 dwords */
#define HVK_VBI_LUTS     4      /* 0 teletext, 1 WSS, 2 VITC, 3 CC608 (32 bit cells + the clock run-in as a 33rd symbol) */

typedef struct { int16_t i, q; } hvk_c16_t;
typedef struct { int32_t i, q; } hvk_c32_t;

/* One scanline's content, indexed [frame & 1][line - 1]. Replaces the
 * reference's per-line code strings (src/video.c:2447-2862). */
typedef struct {
	int16_t pulse_left;     /* pulse index or -1 */
	int16_t pulse_mid;      /* pulse index or -1 */
	int16_t pulse_next;     /* left pulse of the FOLLOWING line if it starts before sample 0, else -1 */
	int16_t al, ar;         /* luma is assigned on [al, ar); al == ar: none */
	int16_t src_row;        /* source row before centring / field shift, -1: none */
	int16_t pal;            /* 0 no chroma, +1, -1 (PAL V switch) */
	int16_t secam_fid;      /* bit 0: SECAM field identification line -- sub-carrier (and luma notch) without a picture;
	                         * bits 8..: the line's row of the base-line table (blanking + sync pulses);
	                         * bits 1..7: its row among the stream's first lines (hvk_kconst_t.spill_lines), where a pulse of the
	                         * line before that runs past its end has NOT reached this one */
} __attribute__((aligned(16))) hvk_linedesc_t;      /* 16 bytes, aligned: one scalar load on the device */

/* RGB -> (Y,U,V) level conversion, evaluated in double on the device with
 * contraction off, operation for operation as src/video.c:3912-3958 */
typedef struct {
	double glut[256];
	double rw, gw, bw;
	double eu, ev;
	double black, range;    /* black_level, white_level - black_level */
	double level;
	double chroma_scale;    /* (white - black) * level */
	int32_t secam;
	/* fast != 0: the short form of the arithmetic below -- Y = f_y0 + y f_y1, U = f_u0 + (b - y) f_u1, V = f_v0 + (r - y) f_v1,
	 * fused multiply-adds, the scale by 32767 folded in, rounding by the add of 1.5 * 2^52, limits applied to the integers --
	 * which differs from the reference's sequence of operations only in the last bits of the doubles. Whether that ever
	 * moves a LEVEL is not argued but tried: the engine sets the flag only after a kernel has computed all 2^24 colours this
	 * way and found every one equal to the table made the reference's way (hvk_k_check_levels, hvk_engine.cpp) */
	int32_t fast;
	double f_y0, f_y1, f_u0, f_u1, f_v0, f_v1;
} hvk_yuvparams_t;

/* Kernel-visible engine constants */
typedef struct {
	int32_t width;          /* samples per line */
	int32_t lines;
	int32_t half_width;
	int32_t active_left;
	int32_t active_width;
	int32_t active_lines;
	int32_t interlaced;
	int32_t fields;         /* frame descriptors per frame: 2 with --interlace (one source frame per field), else 1 */
	int32_t hline;          /* first line (1-based) of the second field */
	int32_t blanking;
	int32_t colour;         /* PAL / NTSC sub-carrier present */
	uint32_t clw;           /* colour lookup period in samples */
	int32_t burst_left, burst_width;
	int32_t burst_i, burst_q;
	int32_t chroma_ntaps;
	int32_t npulses;
	int32_t pulse_offset[HVK_MAX_PULSES];
	int32_t pulse_length[HVK_MAX_PULSES];
	int32_t pulse_start[HVK_MAX_PULSES];   /* index into the flat value array */
	int32_t black_y;        /* luma of RGB 000000 */
	int32_t base_stride;    /* int16 entries per row of the base-line table */
	/* filter / audio stage */
	int32_t vf_type;        /* 0 none, 1 real, 3 real -> complex */
	int32_t vf_ntaps;
	int32_t delay_lines;
	int32_t has_carriers;
	int32_t has_nicam;
	int32_t nicam_ntaps, nicam_sps, nicam_dsl, nicam_decimation, nicam_cc_len;
	uint32_t nicam_inv20;   /* ceil(2^20 / nicam_sps): a / sps == (a * inv20) >> 20 for a < 2^20 / sps */
	int32_t frame_samples;  /* OUTPUT samples per frame (sample rate) */
	int32_t raster_samples; /* width * lines (pixel rate); == frame_samples without the resampler */
	int32_t slab_lines;     /* raster lines kept per frame: lines + 2, + 1 with the resampler */
	int32_t s_lead, s_stride;   /* the filter kernel's input: samples in front of a frame's first one, samples between frames */
	int32_t out_prime;      /* samples of the never-emitted start-up lines the audio / tail processes run over */
	int32_t rs_L, rs_D, rs_ataps;   /* --pixelrate poly-phase resampler: interpolation, decimation, taps per phase; rs_L == 0: none */
	int32_t rs_shift;       /* resampled-stream index (frame local) of output sample 0's filter centre */
	int32_t rs_irr;         /* a raster frame does not resample to a whole number of samples (858 x 525 at 13.5 -> 16 MHz): frames
	                         * of floor / ceil length, frame f's first output sample ceil(f * raster_samples * L / D); frame_samples is the
	                         * longer of the two lengths. A staged batch is then ONE run of samples (hvk_engine_stage.cpp) */
	int32_t secam;          /* SECAM: luma notch + the FM sub-carrier stream (hvk_secam.hip; hvk_secam.c where the device does not take it) */
	int32_t teletext;       /* teletext symbol table present */
	int32_t vbi;            /* VBI data lines (teletext / WSS / VITC ops) may be present */
	int32_t vits;           /* insertion test signals: 0 none, else the number of VITS lines (2 or 4) */
	int32_t vits_line[4];   /* their 0-based line numbers */
	int32_t vits_pi, vits_pq;   /* chroma phase of the insertion signal, Q15 */
	int32_t black;          /* black level (WSS blanks part of line 23 to it) */
	int32_t rawbb;          /* the raster comes from an external baseband stream (--raw-bb-file) */
	int32_t rawbb_blank, rawbb_range, white;    /* its blanking level and white - blanking; the mode's white level */
	int32_t s_video;        /* the colour sub-carrier goes to the Q channel (a second raster slab) */
	int32_t fm_video;       /* the engine's device output is the FM modulator's input (hvk_tail.c does the rest) */
	int32_t swap_iq, has_offset, has_passthru;   /* complex tail done by hvk_k_tail (not FM video) */
	int32_t sis;            /* sound-in-syncs: every line's sync area blanked through a window and carrying 4-level symbols */
	int32_t sis_left, sis_width, sis_sync;  /* the window's first sample, its length, the level it blanks to */
	int32_t sis_dummies;    /* never-emitted invocations before line 1: 1, or 3 behind a threaded colour process (SECAM) */
	int32_t spill_lines;    /* a sync pulse of some line runs on into the next one (Baird's 240 lines: the broad pulse at mid-line is a line long).
	                         * The reference's renderer stops at a line buffer that has not been used yet (src/vbidata.c:219-236 with
	                         * src/video.c:4665): for the stream's first `spill_lines` lines the part that runs over is lost. 0: no such pulse */
	int32_t fsc_mode;       /* field-sequential colour: 0 none, 1 Apollo (525 lines), 2 CBS (405 lines): a line shows ONE colour channel of the
	                         * picture as grey -- channel (frame * 2 + field) mod 3, frames counted from 1 (src/video.c:2919-2930, :2995-3000) */
	int32_t fsc_split;      /* first line (1-based) of the second field for that count: 264, 202 */
	int32_t sv_ring;        /* S-Video behind resampler + video filter where the lines have two widths: the reference pairs a line's luma
	                         * with what ITS RING of line buffers holds in the Q channel (src/video.c:3243, :3578; hvk_k_svq, hvk_engine_launch.cpp).
	                         * The ring's length in lines; 0: every line has one width (or no such combination) */
	int32_t ablate;         /* profiling only (HVK_ABLATE): bit mask of stages to skip; 0 in production */
} hvk_kconst_t;

/* Per rendered frame */
typedef struct {
	int64_t frame_index;    /* 0-based frame number in the stream */
	int64_t fb_offset;      /* pixel offset of the cropped frame's first pixel in the slot pool */
	int32_t fb_width, fb_height;
	int32_t pixel_stride, line_stride;
	int32_t vframe_x, vframe_y;
	int32_t fb_interlaced;
	int32_t fb_valid;       /* 0: no pixels (black) */
	uint32_t clut_off0;     /* colour table position of the frame's first line: (frame_index * lines * width) mod clw */
	int32_t parity;         /* (frame number) & 1 with frames counted from 1: (frame_index + 1) & 1 */
	int32_t plane_row0;     /* picture planes (hvk_direct.hip): the row of this picture's line 0; line l is row plane_row0 + l */
	int32_t chroma_row;     /* SECAM: the row of the sub-carrier store ([rows][raster_samples]) this frame's colour chain output lies in -- its place
	                         * in the batch, or the row kept for its picture and frame number modulo 6 (hvk_engine_stage.cpp: kept sub-carrier) */
} hvk_framedesc_t;

/* Host-built tables (hvk_tables.c) */
typedef struct {
	hvk_config_t conf;      /* with defaults applied */
	int32_t sample_rate, pixel_rate;
	int32_t max_width;      /* widest output line */
	int16_t *rs_taps;       /* [rs_L][rs_ataps] in the order they are applied */
	int32_t white_level, black_level, blanking_level, sync_level;
	hvk_kconst_t k;
	hvk_yuvparams_t yuv;
	hvk_linedesc_t *desc;   /* [2][lines] */
	int16_t *linebase; int32_t nbase;   /* [nbase][k.base_stride]: blanking + sync pulses of every kind of line */
	int16_t *pulse_values; int32_t pulse_total;
	int16_t *sync_packed; int32_t sync_packed_len;  /* reference layout, for tests */
	hvk_c16_t *colour_lookup; int64_t colour_lookup_len;
	int16_t *burst_win;
	int16_t *chroma_taps;
	int chroma_unfiltered;      /* a colour mode without a chroma low pass: chroma_ntaps = 3 stands for "no filter" (fir8<3>) */
	int16_t ghost[HVK_GHOST_LEN];
	int16_t *vf_itaps, *vf_qtaps;
	/* audio */
	int32_t fm_level; hvk_c32_t *fm_lut;
	int32_t am_level; hvk_c32_t am_delta;
	/* A2 stereo (src/video.c:4375-4400): second FM carrier, pilot and identification tones */
	int32_t a2_level; hvk_c32_t *a2_lut;
	int32_t a2_system_m;
	int32_t a2_pilot_level, a2_signal_level;
	hvk_c32_t a2_pilot_delta, a2_signal_delta;
	int16_t *nicam_taps; hvk_c16_t *nicam_cc;
	int16_t limiter_shape[21];
	int32_t limiter_vtaps[65], limiter_ftaps[65];
	int32_t has_limiter;
	/* teletext symbols (src/teletext.c:1057-1074): [360] { offset, length, start in tt_values } */
	int32_t *tt_symbols;
	int16_t *tt_values; int32_t tt_total;
	/* all vbidata look-up tables in one store (hvk_tables.c:_vbi_store): symbol i of LUT u is
	 * vbi_sym[(lut_base[u] + i) * 3 ...] = { first sample, length, start in vbi_val } */
	int32_t *vbi_sym; int32_t vbi_nsym;
	int16_t *vbi_val; int32_t vbi_total;
	int32_t lut_base[HVK_VBI_LUTS], lut_nsym[HVK_VBI_LUTS];
	uint8_t wss_bits[18];       /* line 23's 137 bits, MSB first (src/wss.c:118-136) */
	int32_t wss_blank_lo, wss_blank_hi;
	int32_t vitc_lines[2], vitc_fps, vitc_drop;
	int32_t cc608_line;         /* 1-based */
	int32_t acp_left[6], acp_psync_width, acp_pagc_width, acp_psync_level;
	int32_t grey_y[256];        /* luma level of RGB (i, i, i): what ACP's AGC pulse follows */
	int16_t *vits_l, *vits_c;   /* [vits][width]: luma added, chroma amplitude */
	int16_t *fsc_rows;          /* field-sequential colour: [2][width] the flag pulse(s) as dense rows (src/video.c:4050-4073) */
	/* sound-in-syncs (src/sis.c): the 50 half symbols as dense rows over the line's first HVK_SIS_SPAN samples, the
	 * blanking window, and what the last never-emitted invocation leaves on the stream's first line */
	int16_t *sis_dense;         /* [50][HVK_SIS_SPAN] */
	int16_t *sis_win;           /* [sis_width] */
	int16_t *sis_first;         /* [HVK_SIS_SPAN] */
	/* SECAM (src/video.c:4075-4162) */
	int32_t secam_level;
	hvk_c32_t *secam_lut;       /* 65536 FM steps at the pixel rate */
	hvk_c16_t *secam_bell;      /* 65536 complex gains, indexed by the sample as uint16 */
	int16_t *secam_fir;         /* 15 taps, applied order */
	int16_t *secam_notch;       /* 51 taps, applied order */
	int16_t secam_dmin[2], secam_dmax[2];
	int16_t secam_fsync_level; int32_t secam_fid_lines;   /* field identification lines, src/video.c:4130-4137 */
	/* FM video and frequency offset (src/video.c:4563-4607) */
	int32_t fmv_level; hvk_c32_t *fmv_lut;
	hvk_c32_t offset_delta;
} hvk_tables_t;

int hvk_tables_build(hvk_tables_t *t, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate);
/* widths of output lines [first, first + n) of the stream (they vary with the resampler) */
void hvk_tables_line_widths(const hvk_tables_t *t, int64_t first, int n, int32_t *widths);
int64_t hvk_tables_frame_start(const hvk_tables_t *t, int64_t frame);     /* first output sample of a frame, counted from the stream's first */
void hvk_tables_free(hvk_tables_t *t);
void hvk_tables_default_ghost(hvk_tables_t *t);
long hvk_tables_get(const hvk_tables_t *t, const char *name, void *dst, long max_bytes);

/* The 90 bits of a VITC line (src/vitc.c:120-196) as 12 bytes, least significant bit first;
 * frame counts from 1, line is the 1-based line number. Returns the number of bits. */
int hvk_vitc_bits(const hvk_tables_t *t, int frame, int line, uint8_t data[12]);

/* WSS: line 23's 137 bits (MSB first) for a frame whose source has the given pixel aspect */
void hvk_wss_bits(const hvk_tables_t *t, int64_t par_num, int64_t par_den, uint8_t bits[18]);
/* ACP: the AGC pulse level of a frame (counted from 1), src/acp.c:78-90 */
int hvk_acp_agc_level(const hvk_tables_t *t, int frame);
/* CC608: the 17 bits of a caption byte pair, LSB first (src/cc608.c:170-186) */
void hvk_cc608_bits(uint8_t c1, uint8_t c2, uint8_t data[3]);

/* SECAM colour sub-carrier on the host: the serial chain (hvk_secam.c; the device has its own way, hvk_secam.hip) */
#include "hvk_secam_chain.h"
typedef struct hvk_secam hvk_secam_t;
hvk_secam_t *hvk_secam_new(const hvk_tables_t *t);
void hvk_secam_free(hvk_secam_t *s);
/* the lines of a frame of this parity the process works on (out NULL: just the count) */
int hvk_secam_tasks(const hvk_tables_t *t, int parity, hvk_secam_task_t *out, int max);
void hvk_secam_fid_row(const hvk_tables_t *t, int dr, int16_t level, int16_t *row);
void hvk_secam_counters(const hvk_secam_t *s, int64_t *tasks, int64_t *mismatches, int64_t *repaired);
void hvk_secam_get_state(const hvk_secam_t *s, hvk_secam_state_t *st, int64_t *next_frame);
void hvk_secam_set_state(hvk_secam_t *s, const hvk_secam_state_t *st, int64_t next_frame);
/* fb2 .. : the frame the second field shows (--interlace); pass the first field's again otherwise */
int hvk_secam_frame(hvk_secam_t *s, int64_t frame_index, const uint32_t *fb, int fb_width, int fb_height,
                    int fb_interlaced, const uint32_t *fb2, int fb2_width, int fb2_height, int fb2_interlaced, int16_t *out);

/* The serial part of the output tail (hvk_tail.c): FM video phasor, offset phasor, passthru queue */
typedef struct hvk_tail hvk_tail_t;
hvk_tail_t *hvk_tail_new(const hvk_tables_t *t);
void hvk_tail_free(hvk_tail_t *s);
int hvk_tail_passthru_push(hvk_tail_t *s, const int16_t *iq, size_t nsamples);
/* offset phasor (int16 pairs, phase >> 16) for output positions [first, first + count); forward only */
int hvk_tail_offset_stream(hvk_tail_t *s, int64_t first, int64_t count, int16_t *out);
/* passthru samples added to output positions [first, first + count) (whole lines; zeros where the
 * source has ended); forward only */
int hvk_tail_passthru_stream(hvk_tail_t *s, int64_t first, int64_t count, int16_t *out);
/* FM video: the whole tail on the host, in place, for output positions [first, first + count);
 * strictly sequential */
int hvk_tail_fm_apply(hvk_tail_t *s, int64_t first, int64_t count, int16_t *iq);
/* ... before that, once: the modulator's input over the never-emitted start-up samples (with the video filter on) */
int hvk_tail_fm_prime(hvk_tail_t *s, const int16_t *input, int64_t count);
int64_t hvk_tail_fm_position(const hvk_tail_t *s);

/* Host audio-rate control path (hvk_audio.c) */
typedef struct hvk_audio hvk_audio_t;

hvk_audio_t *hvk_audio_new(const hvk_tables_t *t);
void hvk_audio_free(hvk_audio_t *a);
int hvk_audio_push(hvk_audio_t *a, const int16_t *stereo, size_t nsamples);
size_t hvk_audio_source_needed(const hvk_audio_t *a, int64_t upto_pos);

/* Generate the side streams for audio-stream positions [first, first + count)
 * (count a multiple of width; first must be >= every earlier request's end or
 * inside the retained window). carriers: count int16 pairs. Symbols: every
 * NICAM symbol that can touch the range; *k0 receives the stream index of
 * symbols[0]. Returns the number of symbols written, or < 0. */
int hvk_audio_generate(hvk_audio_t *a, int64_t first, int64_t count,
                       int16_t *carriers, uint8_t *symbols, int max_symbols, int64_t *k0);
int64_t hvk_audio_position(const hvk_audio_t *a);
/* the chains' state as a flat block (hvk_audio.c), for an engine that goes on where another one stopped */
size_t hvk_audio_state_bytes(void);
int hvk_audio_state_export(hvk_audio_t *a, void *buf, size_t bytes);
int hvk_audio_state_import(hvk_audio_t *a, const void *buf, size_t bytes, int64_t *source_pos);
int64_t hvk_audio_generated(const hvk_audio_t *a);
int64_t hvk_audio_source_end(const hvk_audio_t *a);
/* sound-in-syncs: the bursts of stream lines [g_first, g_first + count), 8 bytes a line (7 bytes of bits MSB first, their number) */
int hvk_audio_sis_fetch(hvk_audio_t *a, int64_t g_first, int count, uint8_t *out);
int hvk_audio_advance(hvk_audio_t *a, int64_t end);

#ifdef __cplusplus
}
#endif

#endif
