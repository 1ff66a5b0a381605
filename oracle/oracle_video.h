/* oracle/oracle_video.h -- TEST INFRASTRUCTURE (not product code).
 *
 * C ABI of the CPU restatement of hacktv's composite-video -> IQ path.
 * Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg only. The product (hacktv_amd/) never links or calls it.
 */
#ifndef ORACLE_VIDEO_H
#define ORACLE_VIDEO_H

#include <stdint.h>
#include "../include/hvk_config.h"

typedef struct orc_t orc_t;

orc_t *orc_open(const hvk_config_t *conf, unsigned int sample_rate);
/* with --pixelrate: the raster is built at pixel_rate and resampled to sample_rate (0: same rate) */
orc_t *orc_open_rates(const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate);
void orc_close(orc_t *s);

/* Geometry / levels in the order tests/refprobe.py INFO_NAMES lists */
int orc_info(orc_t *s, int32_t *out, int n);

/* Table dump, same names as oracle/ref_probe.c:ref_table */
long orc_table(orc_t *s, const char *name, void *dst, long max_bytes);

/* The int16 values the reference's chroma FIR over-reads past the end of
 * its chrominance buffer. Default: the glibc-2.35 fresh-heap
 * model (slack, chunk size word, burst window). */
void orc_set_ghost(orc_t *s, const int16_t *ghost, int n);

/* Current source frame (kept by reference, like av_read_video's contract) */
void orc_set_frame(orc_t *s, const uint32_t *fb, int width, int height, int pixel_stride, int line_stride, int interlaced);

/* --interlace (conf.interlace): the frame the SECOND field of the frames rendered from now on shows
 * (the reference pulls a new frame at line hline, src/video.c:4873). Call after orc_set_frame(). */
void orc_set_frame2(orc_t *s, const uint32_t *fb, int width, int height, int pixel_stride, int line_stride, int interlaced);

/* 32 kHz interleaved stereo source; loop != 0 repeats it forever (av_test) */
void orc_set_audio(orc_t *s, const int16_t *stereo, long nsamples, int loop);

/* Teletext packets for one frame (0-based stream frame index): slot 0..15 are
 * lines 7..22, slot 16..31 lines 320..335; bit i of mask says slot i carries
 * packets[i] (45 bytes: clock run-in, framing code, 42 data bytes). 625-line
 * modes only. Up to 16 frames may be queued ahead. */
int orc_teletext_packets(orc_t *s, long frame_index, const uint8_t *packets, uint32_t mask);

/* the pixel aspect ratio of the current source frame (only --wss auto looks at it) */
void orc_set_frame_aspect(orc_t *s, long long par_num, long long par_den);

/* --raw-bb-file (conf.raw_bb): the external int16 baseband stream, kept by reference; read line by
 * line, starting over at its end (src/video.c:2419-2429) */
void orc_set_rawbb(orc_t *s, const int16_t *samples, long nsamples);

/* --cc608: the caption byte pair of a frame (0-based stream frame index); frames without a call send zeros */
void orc_set_cc608(orc_t *s, long frame_index, uint8_t c1, uint8_t c2);

/* --sis: how many samples of a step's audio line the reference's audio thread is taken to have behind it when its SiS
 * process looks for the newest audio block (oracle_sis.c; 0: none -- the default) */
void orc_set_sis_visible(orc_t *s, int samples);
void orc_set_sis_heap(orc_t *s, const int16_t *h8);
/* ... the bursts of the lines rendered so far (8 bytes a line: 7 bytes of bits, MSB first, and their number); -1: not made yet */
long orc_sis_bursts(orc_t *s, long first_line, long nlines, uint8_t *out);

/* --passthru: the external int16 I/Q signal (kept by reference), from its first sample */
void orc_set_passthru(orc_t *s, const int16_t *iq, long nsamples);

/* Render the next nlines emitted lines (interleaved I/Q int16). Returns
 * the number of samples (pairs) written. */
long orc_render_lines(orc_t *s, int16_t *iq, long nlines);

/* Widths of the lines the last orc_render_lines call emitted (they vary with --pixelrate) */
long orc_last_widths(orc_t *s, int32_t *dst, long max);

/* Stage taps for tests: the final raster (I channel, before filter/audio) of
 * the lines produced by the last orc_render_lines call */
long orc_last_raster(orc_t *s, int16_t *dst, long max_samples);

/* The serial-carrier contribution (FM/AM audio, I/Q int16 pairs) and NICAM
 * symbol values added to the lines of the last call */
long orc_last_carrier(orc_t *s, int16_t *dst, long max_samples);

#endif
