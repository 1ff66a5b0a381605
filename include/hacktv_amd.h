/* hacktv_amd.h -- C ABI of the MI355X composite-video -> IQ engine (libhvk.so).
 *
 * This is the drop-in boundary for the one hot path of fsphil/hacktv that is
 * rebuilt here: everything between the av_* source API and the rf_* sink API,
 * i.e. what the reference implements in vid_init() / vid_next_line() /
 * vid_free() (src/video.h:510-516, src/video.c:3812-4952) on top of fir.c,
 * vbidata.c and nicam728.c.
 *
 * Plain C: opaque handle, plain pointers and sizes, no C++/torch types. The
 * reference's host program stays C and calls this through the video.h-shaped
 * shim in hacktv_amd/csrc/shim/ (INTEGRATION.md shows the wiring). Python
 * (tests/, bench.py) binds the same symbols with ctypes.
 *
 * Model. The reference renders one scanline per vid_next_line() call on the
 * host. The engine renders WHOLE FRAMES on the GPU, many per launch:
 *
 *   hvk_open()            vid_init(): builds every table on the host (double +
 *                         libm, bit-identical with the reference's) and
 *                         uploads them once.
 *   hvk_frame_upload()    what av_read_video() returned for a frame, copied
 *                         into one of the engine's frame slots in HBM.
 *   hvk_audio_write()     what av_read_audio() returned: 32 kHz stereo int16.
 *                         Runs the audio-rate control path on the host
 *                         (volume, limiter, NICAM framing, and the serial
 *                         FM/AM phasor chains) and queues the
 *                         per-sample side streams for upload.
 *   hvk_render()          renders the next `nframes` frames of the stream
 *                         into device memory: raster kernel, filter/audio
 *                         kernel. Output: int16 I/Q pairs, frame after frame,
 *                         line after line -- exactly the concatenation of the
 *                         buffers the reference hands to rf_write()
 *                         (src/hacktv.c:1579-1587).
 *   hvk_fetch()           copies rendered samples to a host buffer for a
 *                         host-side rf_* sink; hvk_fetch_async() /
 *                         hvk_fetch_wait() queue that copy behind the render
 *                         (page-locked buffers: hvk_host_alloc()) so that it
 *                         runs beside the next batch's host work.
 *
 * SECAM's colour sub-carrier -- in the reference one chain over every line of
 * the stream -- is computed on the device as well, every line from a derived
 * entry state that is then checked bit for bit (hvk_secam_stats()).
 *
 * All entry points return HVK_OK (0) or a negative HVK_* code, like the
 * reference's VID_OK / VID_ERROR / VID_OUT_OF_MEMORY (src/video.h:45-47).
 * There is no CPU fallback: if the HIP device is missing every call that
 * would touch it fails with HVK_NO_DEVICE.
 */
#ifndef HACKTV_AMD_H
#define HACKTV_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "hvk_config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HVK_OK              0
#define HVK_ERROR          -1   /* VID_ERROR */
#define HVK_OUT_OF_MEMORY  -2   /* VID_OUT_OF_MEMORY */
#define HVK_NO_DEVICE      -3
#define HVK_UNSUPPORTED    -4   /* configuration outside the engine's scope */

/* hvk_config_apply_flags(): the reference CLI's preset edits */
#define HVK_FLAG_FILTER    (1 << 0)  /* --filter   src/hacktv.c:1412 */
#define HVK_FLAG_NOAUDIO   (1 << 1)  /* --noaudio  src/hacktv.c:1150 */
#define HVK_FLAG_NONICAM   (1 << 2)  /* --nonicam  src/hacktv.c:1167 */
#define HVK_FLAG_NOCOLOUR  (1 << 3)  /* --nocolour src/hacktv.c:1126 */

/* ---- presets: vid_configs[] (src/video.c:1956-2008) ---- */
int hvk_config_preset(hvk_config_t *conf, const char *mode_id);
void hvk_config_apply_flags(hvk_config_t *conf, int flags);
const char *hvk_preset_id(int index);
const char *hvk_preset_desc(int index);

/* ---- engine ---- */
typedef struct hvk_engine hvk_engine_t;

/* Geometry and levels vid_init() derives (src/video.c:3844-3881); the
 * members hacktv.c reads from vid_t after vid_init() (src/hacktv.c:1503-1518)
 * are all here. */
typedef struct hvk_info_t {
	uint32_t struct_size;       /* set by the CALLER to sizeof(hvk_info_t) before hvk_get_info(): any other value is
	                             * HVK_ERROR and nothing is written (a caller built against another release's layout) */
	int32_t sample_rate;
	int32_t width;              /* samples per line */
	int32_t half_width;
	int32_t active_width;       /* source frame width the raster shows */
	int32_t active_left;
	int32_t lines;              /* lines per frame */
	int32_t active_lines;       /* source frame height */
	int32_t white_level, black_level, blanking_level, sync_level;
	int32_t delay_lines;        /* filter latency in lines (dropped at start) */
	int32_t frame_samples;      /* width * lines */
	int32_t max_frames;         /* frames one hvk_render() call may take */
	int32_t frame_slots;        /* source frame slots in HBM */
	int32_t colour_lookup_width;
	int32_t burst_left, burst_width;
	int32_t has_carriers;       /* serial FM/AM audio carriers present */
	int32_t has_nicam;
	int32_t pixel_rate;         /* rate the raster is built at; width, half_width, active_* are in pixels of it */
	int32_t max_width;          /* widest line handed out (== width unless the resampler is on) */
	int32_t startup_samples;    /* samples of the never-emitted start-up lines the audio and tail processes
	                             * run over before the first output sample (delay_lines * width without the
	                             * resampler) */
} hvk_info_t;

/* vid_init(): src/video.c:3812. `device` is the HIP device ordinal.
 * `max_frames` bounds one render call (device buffers are sized for it). */
int hvk_open(hvk_engine_t **e, const hvk_config_t *conf, unsigned int sample_rate, int device, int max_frames);

/* vid_init() with --pixelrate (src/video.c:3812, :3839, :4361-4367): the raster is built at
 * pixel_rate and a rational poly-phase FIR resamples it to sample_rate before the video
 * filter (src/video.c:3627-3651, src/fir.c:393-428). 0 or == sample_rate: no resampler.
 * Frame-regular rates only: raster samples per frame * L / D must be whole (L / D the
 * reduced sample_rate / pixel_rate), L <= 256, D <= 4 L. info.frame_samples is the OUTPUT
 * frame length; lines have the widths hvk_line_widths() reports. As in the reference the
 * resampled first raster line never leaves the pipeline: the stream starts with line 2. */
int hvk_open_rates(hvk_engine_t **e, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate,
                   int device, int max_frames);

/* Widths of output lines [first_line, first_line + nlines) of the stream, counted from 0 (what
 * vid_next_line() puts in vid_line_t.width): always info.width without the resampler. */
int hvk_line_widths(const hvk_engine_t *e, int64_t first_line, int nlines, int32_t *widths);

/* The first output sample of stream frame `frame`, counted from the stream's first: frame * frame_samples -- but for
 * --pixelrate pairs at which a raster frame does not resample to a whole number of samples (525 lines, 13.5 -> 16 MHz:
 * 450450 * 32 / 27). Frames then have two lengths one sample apart, hvk_info_t.frame_samples is the longer, a render of n
 * frames from frame f yields hvk_frame_start(f + n) - hvk_frame_start(f) samples without gaps (what hvk_fetch() counts
 * in), and such a stream renders in order on one engine: no strides, no interleaved output slots. The lines keep the
 * widths hvk_line_widths() gives (src/video.c:3246): they never depended on where a frame is cut. */
int64_t hvk_frame_start(const hvk_engine_t *e, int64_t frame);

/* vid_free(): src/video.c:4706 */
void hvk_close(hvk_engine_t *e);

int hvk_get_info(const hvk_engine_t *e, hvk_info_t *info);

/* vid_get_framebuffer_length(): src/video.c:4862 */
size_t hvk_get_framebuffer_length(const hvk_engine_t *e);

/* The int16 values the reference's chroma filter reads past the end of its
 * chrominance buffer (src/fir.c:365-372 called with samples = width,
 * src/video.c:3019-3020). The default reproduces the reference
 * CLI on glibc 2.35; a caller that embeds the reference differently can
 * supply what its own heap holds. n <= 32. Call before the first render. */
int hvk_set_chroma_ghost(hvk_engine_t *e, const int16_t *ghost, int n);
int hvk_get_chroma_ghost(const hvk_engine_t *e, int16_t *ghost, int n);

/* av_read_video() result -> frame slot. Strides are in pixels and may be
 * negative (src/av.h:31-54). The frame is centre-cropped to the active area
 * like src/video.c:4887-4897. fb == NULL stores an empty (black) frame. */
int hvk_frame_upload(hvk_engine_t *e, int slot, const uint32_t *fb, int width, int height,
                     int pixel_stride, int line_stride, int interlaced);

/* hvk_frame_upload() for a picture in page-locked memory (hvk_host_alloc()), pixels adjacent, rows `width` pixels
 * apart: nothing is copied on the host, the centre crop goes from where it lies into the slot by one strided DMA queued
 * on the engine's stream. The picture must stay as it is until that copy is through -- hvk_sync(), or the return of
 * the hvk_fetch() / hvk_fetch_wait() of a batch staged after this call. (SECAM engines keep a host copy of every
 * picture and take the ordinary way.) A source that decodes into such memory uploads at PCIe speed. */
int hvk_frame_upload_pinned(hvk_engine_t *e, int slot, const uint32_t *fb, int width, int height, int interlaced);

/* The pixel aspect ratio of the frame in `slot` (av_frame_t.pixel_aspect_ratio, src/av.h:43):
 * only `--wss auto` (conf.wss == 0xFF) looks at it (src/wss.c:166-179). 1:1 unless set, and it stays with the slot: a
 * caller that mirrors a source says it again with every upload -- the reference's frames without a picture have square
 * pixels (av_frame_init(), src/av.c:21-33; the shim passes what the source's frame says). */
int hvk_frame_aspect(hvk_engine_t *e, int slot, int64_t par_num, int64_t par_den);

/* Teletext (conf.teletext != 0): the packets for the VBI lines of frame
 * `frame_in_batch` (0-based) of the NEXT hvk_render() / hvk_stage_strided()
 * call. packets is [32][45] bytes -- row 0..15 for lines 7..22, row 16..31 for
 * lines 320..335 (src/teletext.c:1222-1224), each row the 45 bytes the
 * reference's tt_next_packet() fills (src/teletext.c:1178-1209): clock run-in
 * 55 55, framing code 27, 42 data bytes. Bit i of mask set: row i is sent
 * (TT_OK); clear: the line stays empty (TT_NO_PACKET). Which packet goes where
 * is the caller's page store / scheduler; the engine shapes the 360 symbols
 * and adds them to the line (vbidata_render, src/vbidata.c:186-239). Frames
 * without a call carry no teletext. */
int hvk_teletext_packets(hvk_engine_t *e, int frame_in_batch, const uint8_t *packets, uint32_t mask);
/* The same for frames [first_frame_in_batch, first_frame_in_batch + nframes) of the next batch in one call: packets
 * [nframes][32][45], masks [nframes] (a caller in an interpreted language pays its call overhead once per batch). */
int hvk_teletext_packets_block(hvk_engine_t *e, int first_frame_in_batch, int nframes, const uint8_t *packets, const uint32_t *masks);

/* Which lines (held[line - 1] != 0) the configuration's other inserters -- VITS, WSS, ACP, VITC, CC608, SECAM field
 * identification -- write to: the lines on which the reference's vid_line_t.vbialloc is already set when its
 * teletext process looks (src/teletext.c:1219), where a packet must not go and is kept for the next free line.
 * nlines >= the mode's line count. Needs no device. */
int hvk_vbi_lines_held(const hvk_engine_t *e, uint8_t *held, int nlines);

/* --raw-bb-file (conf.raw_bb != 0): the next nsamples int16 samples of the external baseband
 * stream that takes the raster's place (src/video.c:2406-2446), in stream order; a line is `width`
 * samples (of the pixel rate). Rendering frame f needs the stream up to sample ((f + 1) * lines + 1) * width (the
 * filter looks into the line after the frame; with --pixelrate one line more: the resampler's chunks lag the raster
 * by a slot); what is not queued reads as zeros. Starting the
 * file over at its end, as the reference does, is the caller's job. */
int hvk_rawbb_write(hvk_engine_t *e, const int16_t *samples, size_t nsamples);

/* --cc608 (conf.cc608 != 0): the caption byte pair av_read_video() delivered with frame
 * `frame_in_batch` of the NEXT render (av_frame_t.cc608, src/av.h:52; queued by
 * src/video.c:4901-4904, sent on line 22 / 21 by src/cc608.c:188-221). Frames without a call,
 * or with an empty pair, send the parity-only null code. */
int hvk_cc608_write(hvk_engine_t *e, int frame_in_batch, uint8_t c1, uint8_t c2);

/* av_read_audio() result: nsamples interleaved stereo pairs at 32 kHz. */
int hvk_audio_write(hvk_engine_t *e, const int16_t *stereo, size_t nsamples);

/* 32 kHz stereo pairs still needed before `nframes` more frames can be
 * rendered with audio (0 when the mode has no audio sub-carriers). */
size_t hvk_audio_needed(const hvk_engine_t *e, int nframes);

/* Sharded streams with sound. The FM / AM carrier phasors, the limiter and the NICAM framer are ONE chain over the whole
 * stream (src/video.c:2259-2276, :3261-3450; src/fir.c:758-870; src/nicam728.c:140-249): an engine that renders frames
 * f, f + 1, ... has to start where the engine that rendered up to frame f - 1 stopped. hvk_sound_state_export() writes
 * that state -- everything the chains carry from one line to the next, hvk_sound_state_size() bytes, valid between
 * engines of the same build and configuration -- as it stands after the last frame staged;
 * hvk_sound_state_import() puts it into an engine that has staged nothing of those frames yet. *source_position
 * (may be NULL) receives the position in the 32 kHz source stream the chains go on from: an engine whose audio queue
 * does not hold it drops what it holds and takes the next hvk_audio_write() to start there. With it a rank of a
 * sharded render runs the chains over its own frames only (hvk_sound_samples_generated() says over how many samples
 * it did) instead of over the whole stream up to them. An engine whose queue DOES hold the position keeps the pairs
 * behind it: hvk_sound_source_end() is the source position the next hvk_audio_write() continues at (the caller must not
 * write pairs in front of it a second time). */
size_t hvk_sound_state_size(const hvk_engine_t *e);
int hvk_sound_state_export(hvk_engine_t *e, void *buf, size_t bytes);
int hvk_sound_state_import(hvk_engine_t *e, const void *buf, size_t bytes, int64_t *source_position);
int64_t hvk_sound_samples_generated(const hvk_engine_t *e);
int64_t hvk_sound_source_end(const hvk_engine_t *e);

/* --passthru (conf.passthru != 0): the next nsamples int16 I/Q pairs of the
 * external signal that is added to the output (src/video.c:3517-3541), in
 * stream order from the source's first sample. Like the reference the engine
 * adds whole lines only: a line for which the source is short gets nothing,
 * and neither does any later line (the reference's end-of-file behaviour), so
 * queue at least the samples of the frames about to be rendered -- plus, once,
 * delay_lines * width more when the video filter is on: the line pipeline
 * feeds its never-emitted start-up line through the process as well. */
int hvk_passthru_write(hvk_engine_t *e, const int16_t *iq, size_t nsamples);

/* Render the next nframes frames of the stream. slots[i] names the frame
 * slot shown by frame i -- with conf.interlace slots[2 i] and slots[2 i + 1]
 * name the slots the first and the second field of frame i show (the
 * reference takes a new source frame at the start of each field,
 * src/video.c:4873). d_iq, if not NULL, is a DEVICE pointer to
 * nframes * frame_samples * 2 int16 that receives the samples; if NULL the
 * engine's own output buffer is used (read it back with hvk_fetch()).
 * Missing audio is rendered as silence (src/video.c:3299-3304). */
int hvk_render(hvk_engine_t *e, int nframes, const int32_t *slots, void *d_iq);

/* Same, for a caller that shards the stream over several engines/GPUs: render
 * frames first_frame, first_frame + stride, ... (nframes of them) of the
 * stream this engine's audio queue describes. The audio side streams for
 * every rendered frame must have been queued (hvk_audio_write() is stream
 * ordered and must be fed the whole stream on every rank). */
int hvk_render_strided(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes,
                       const int32_t *slots, void *d_iq);

/* The two halves of hvk_render_strided(), for callers that want the side
 * inputs resident in HBM before they start a clock: hvk_stage_strided() runs
 * the host audio control path for the named frames and uploads frame
 * descriptors, carrier side stream and NICAM symbols; hvk_launch() enqueues
 * the raster and filter kernels for the staged batch (it may be called
 * repeatedly for the same staged batch). */
/* (On 525 lines the last line of a frame shows picture and lies within the filter's reach of the next
 * frame's first samples. Consecutive frames -- within a batch or from one call to the next -- are
 * handled exactly: the engine keeps the source row. A strided call, or one that does not continue where
 * the last one ended, does not have the frame before: on those modes it is refused with HVK_UNSUPPORTED --
 * hvk_stage_strided_prev() takes the slot that holds the frame before, the frame's own slot where the
 * picture does not change.) */
int hvk_stage_strided(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots);
/* As hvk_stage_strided(), for sharded streams whose pictures change: prev_slots[i] names the frame slot that holds the
 * picture of the frame BEFORE staged frame i (with conf.interlace: the one its second field shows), -1 if the caller
 * does not have it. It is looked at where the engine does not have that frame itself -- a stride other than 1, or a
 * call that does not continue the last one -- and makes such renders exact on 525 lines, where the last line of a
 * frame shows picture within the video filter's reach of the next frame's first samples (src/video.c:4873-4897 takes a
 * new frame at line 1; the line pipeline still holds the old frame's last line). */
int hvk_stage_strided_prev(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots,
                           const int32_t *prev_slots);
int hvk_launch(hvk_engine_t *e, void *d_iq);

/* As hvk_launch(), but frame i of the batch is written at frame position
 * i * out_stride of d_iq (out_stride >= 1): a rank that renders every N-th
 * frame writes straight into its slots of the interleaved stream buffer. */
int hvk_launch_strided_out(hvk_engine_t *e, void *d_iq, int64_t out_stride);

/* Run the engine on a caller's HIP stream (hipStream_t passed as void *),
 * e.g. torch.cuda.current_stream().cuda_stream, so that its kernels order
 * with the caller's work. NULL restores the engine's own stream. */
int hvk_set_stream(hvk_engine_t *e, void *hip_stream);

/* The host half on its own: run the audio-rate control path up to stream
 * position first + count and hand back the side streams for
 * [first, first + count) -- count int16 I/Q pairs of serial-carrier samples
 * and the NICAM symbols that touch the range (*k0 = stream index of
 * symbols[0]; 0xFF marks "before the first symbol"). Positions are audio-stream
 * positions (output position + info.startup_samples); requests go forward
 * (a request may start inside the line the previous one ended in). Needs no device. Returns the number of
 * symbols written or a negative HVK_* code. */
int hvk_host_side_streams(hvk_engine_t *e, int64_t first, int64_t count,
                          int16_t *carriers, uint8_t *symbols, int max_symbols, int64_t *k0);

/* --sis (conf.sis), host half on its own: the sound-in-syncs bursts of stream lines [first_line, first_line + nlines),
 * lines counted from 0 over the whole stream -- 8 bytes a line: the burst's bits, most significant first (the two start
 * bits, then 23 or 25 grey-coded bit pairs of the NICAM-728 stream, src/sis.c:155-201), and in the eighth byte their
 * number (46 or 50). Runs the sound chains up to the lines' end; forward only; use it instead of, not next to,
 * hvk_render(). Needs no device. */
int hvk_host_sis_bursts(hvk_engine_t *e, int64_t first_line, int nlines, uint8_t *out);

/* SECAM only, host half on its own: the value the colour process adds to every
 * sample of the NEXT frame of the stream (frame_samples int16), given the
 * picture shown on it (fb == NULL: an empty frame). Frames are taken in stream
 * order; the call advances the pre-pass, so use it instead of, not next to,
 * hvk_render(). Needs no device. */
int hvk_host_secam_stream(hvk_engine_t *e, const uint32_t *fb, int width, int height, int interlaced, int16_t *out);

/* SECAM: how the colour sub-carrier's line-to-line chain went so far. Lines are worked on independently, each from
 * an entry state derived by running a few lines before it from nothing; the derived states are then checked against
 * the true ones (the exit states of the lines before) and lines that started wrong are redone in order.
 * counts[0] lines worked on speculatively, [1] lines whose derived entry state was wrong, [2] lines redone in order
 * because of them, [3] frames that went through the host's serial chain instead. */
int hvk_secam_stats(hvk_engine_t *e, int64_t counts[4]);
/* ... and how many stages took the entry states of NEW pictures' lines from the estimate kernel (the values behind a line
 * from the summed angle of the FM loop's steps, the IIR's state from a short walk of the IIR alone) instead of from
 * warm-up walks over the twelve lines before; HVK_SECAM_EST=0 or a pinned HVK_SECAM_WARMUP keep the walks. The check
 * is the same either way. */
int64_t hvk_secam_estimated_stages(const hvk_engine_t *e);
/* Which kernel walked the stages' lines: counts[0] hvk_k_secam_chain (warm-up lines, runs of several lines per lane),
 * [1] hvk_k_secam_walk<0> (one line per lane from an estimated or kept entry state; FM step and bell-filter gain of a
 * sample from the 16-byte table entry), [2] hvk_k_secam_walk<1> (the step COMPUTED -- coarse phasor from LDS times a
 * polynomial fine one, rounded -- and the gain decoded from 32-index blocks in LDS: no table read from HBM per sample;
 * taken for blocks that show pictures of many colours). Returns 2 where hvk_open() tried the computed steps and decoded
 * gains on every index of the deviation range and found them equal to the tables' (src/video.c:2218-2243, :2172-2185),
 * 1 where only the table form may be taken, 0 without the device's chain. HVK_SECAM_WALK=0 / 1 / 2 forces a kernel. */
int hvk_secam_walk_stages(const hvk_engine_t *e, int64_t counts[3]);
/* The kept sub-carrier. What the colour chain makes of a frame (src/video.c:3068-3233) is a function of the picture's cells,
 * the frame's number modulo 6 (D'r / D'b: modulo 2; the sub-carrier's start phase, :3211-3212: modulo 3) and the state its first
 * line starts from. A picture that stays meets all three again, so the rows of a walk whose every line passed the check are
 * kept per picture slot and number, with every line's entry state and the state behind the last line; a later frame of that
 * picture and number TAKES the set instead of being walked, and the same check that every line gets -- does it start where
 * the line before ended, bit for bit -- decides whether it may. A set is tried only behind the picture its frame stood behind
 * when it was made (what a frame starts from is what the frame before it leaves: its picture and number); the frame behind a
 * change of picture is walked from estimated states. counts[0]: frames that took a set; [1]: stages done again
 * without kept sets because a frame did not start where its set's walk had; [2]: picture slots sets are kept for (0: none --
 * HVK_SECAM_KEEP=0, --interlace, the host's chain). SECAM-L test card, 128-frame blocks: 170 -> see DESIGN.md section 5. */
int hvk_secam_kept(const hvk_engine_t *e, int64_t counts[3]);

/* Levels computed per pixel (hvk_set_levels(): pictures with many colours) take the short form of the arithmetic -- fused
 * multiply-adds, the scale folded into the constants, rounding by a magic addend -- where hvk_open() has TRIED it on every one
 * of the 2^24 colours of the mode and found the table's levels (src/video.c:3912-3958) each time: 1. Otherwise, and with
 * HVK_EXACT_LEVELS=1, the reference's sequence of operations: 0. */
int hvk_levels_short_form(const hvk_engine_t *e);

/* SECAM: the number of lines a lane walks in front of a line to derive its entry state, as it stands. It follows the
 * pictures (one less after a block without a wrong start, two more after one with; HVK_SECAM_WARMUP=n in the
 * environment pins it) and decides only how much is redone, never what comes out. < 0: an HVK_* code. */
int hvk_secam_warmup_lines(hvk_engine_t *e);

/* --offset, host half on its own: the values the offset process multiplies
 * output samples [first, first + count) by (src/video.c:3482-3515): count int16
 * pairs, the free-running Q31 phasor >> 16, advanced over the pipeline's
 * start-up line first when the video filter is on. Forward only. Needs no
 * device. */
int hvk_host_offset_stream(hvk_engine_t *e, int64_t first, int64_t count, int16_t *out);

/* FM video, host half on its own: run the serial tail (FM phasor, then swap,
 * offset and passthru as configured) over the next `count` samples of the
 * stream, in place: iq holds the modulator's input in its I values on entry
 * and the modulated I/Q pairs on return. Use it instead of, not next to,
 * hvk_fetch(). Needs no device. */
int hvk_host_fm_video(hvk_engine_t *e, int16_t *iq, int64_t count);

/* Wait for the engine's stream; returns HVK_OK or the HIP failure. */
int hvk_sync(hvk_engine_t *e);

/* Copy rendered samples [first, first + count) of the last render (engine
 * buffer) to host memory: what the shim hands to rf_write().
 * FM video modes (conf.modulation == HVK_FM): the device renders the
 * modulator's input (composite + sound carriers); the FM phasor -- a serial
 * recurrence over every sample, src/video.c:2299-2335 -- and whatever follows
 * it (swap, offset, passthru) run on the host inside this call, once per
 * sample and in stream order: frames must be rendered consecutively, d_iq must
 * be NULL, hvk_fetch_as() is not available, and samples of a batch that were
 * never fetched are modulated when the next batch is staged. */
int hvk_fetch(hvk_engine_t *e, int16_t *iq, size_t first, size_t count);

/* hvk_fetch() without the wait: the copy is queued behind the render and the call returns a ticket (0 .. 3, reused in
 * turn; negative: an HVK_* code); iq may be read once hvk_fetch_wait() has returned for that ticket. With iq in page-locked
 * memory (hvk_host_alloc()) the copy runs at PCIe speed beside the host pre-passes of the next hvk_stage() -- how the
 * shim keeps the reference's serial sound chain (src/video.c:2249-2289, the slowest stage of the drop-in) busy all the
 * time. hvk_fetch_wait() may be called from another thread than the one that renders.
 * FM video: the ticket stands for a job of the engine's own FM thread -- the modulator's input is copied into iq and the
 * serial phasor pass runs over it there, job after job in stream order, beside the caller's next hvk_stage(); a request
 * that does not continue where the last one ended, a configuration with --passthru (whose queue the caller's thread
 * fills) and HVK_FM_SYNC=1 keep the pass inside this call. What went out this way is not handed out again by hvk_fetch(). */
int hvk_fetch_async(hvk_engine_t *e, int16_t *iq, size_t first, size_t count);
int hvk_fetch_wait(hvk_engine_t *e, int ticket);

/* Page-locked host memory for hvk_fetch() / hvk_fetch_async() targets (a pageable target is copied through the runtime's
 * bounce buffers at a fraction of the PCIe rate). NULL: no device or out of memory. */
void *hvk_host_alloc(hvk_engine_t *e, size_t bytes);
void hvk_host_free(hvk_engine_t *e, void *p);

/* Sample formats of the reference's file sink (src/rf.h:31-36) */
#define HVK_UINT8  0
#define HVK_INT8   1
#define HVK_UINT16 2
#define HVK_INT16  3
#define HVK_INT32  4
#define HVK_FLOAT  5

/* hvk_fetch() with the file sink's sample-format conversion done on the device
 * (src/rf_file.c:34-277; int8 complex is also what the HackRF sink sends,
 * src/rf_hackrf.c:246-276): samples [first, first + count) of the last render
 * are converted to `type`, I only (complex == 0) or I/Q pairs, and only the
 * converted bytes cross PCIe -- 2 B/sample for int8 complex instead of 4.
 * dst receives count * (complex ? 2 : 1) values. Returns the bytes written or
 * a negative HVK_* code. */
long hvk_fetch_as(hvk_engine_t *e, void *dst, size_t first, size_t count, int type, int complex_out);

/* Device pointer of the engine's own output buffer (for HIP/RCCL callers) */
void *hvk_output_device_ptr(hvk_engine_t *e);

/* Average duration in milliseconds of each kernel over the launches since
 * the last reset, measured with HIP events on the engine's stream.
 * which: 0 raster kernel, 1 filter/audio kernel. */
/* RGB -> level conversion of the raster (src/video.c:3912-3958 builds a 2^24-entry table, :2981-2995
 * looks every pixel up). The engine has the same table in HBM and the arithmetic that fills it as a
 * device function; results are identical. Looking up is cheaper while the pictures' colours are few
 * and recur (test cards, graphics: the entries stay in cache); computing is cheaper for camera
 * pictures, where most look-ups would be an HBM round trip (measured: DESIGN.md section 6).
 * AUTO (default; environment HVK_LEVELS=table|compute|auto overrides the default) decides per staged
 * block from a sample of each uploaded picture's pixels. */
#define HVK_LEVELS_AUTO    0
#define HVK_LEVELS_TABLE   1
#define HVK_LEVELS_COMPUTE 2
int hvk_set_levels(hvk_engine_t *e, int mode);

/* The plain configurations (PAL / NTSC / monochrome at the sample rate, one picture per frame, no inserters) render
 * from PICTURE PLANES: what src/video.c:2864-3030 computes of a scanline before the sub-carrier is modulated -- sync
 * pulses, the levels of the pixels, the low-passed chroma, the burst -- depends on the picture alone and is made once
 * per uploaded picture, by the first hvk_stage_strided() / hvk_render() that shows it (a picture that stays is not
 * worked on again) -- when that block is LAUNCHED, on the engine's stream in front of the render
 * (HVK_PREP_CHUNK / HVK_PREP_STREAMS=2 keep the measured-and-lost experiment of making them a chunk of frames at a time
 * on a second stream beside the render of the chunk before). hvk_planes_refresh() has the planes of the named slots made again by the next launch that shows them: for a
 * caller that wants that work inside a clock of its own. SECAM has a
 * per-picture share of the same kind: the low-passed colour-difference cells of a picture, kept per slot and frame
 * parity (hvk_secam.hip); for the named slots they are dropped and made again by the next stage that shows them.
 * HVK_OK and nothing done where the configuration renders straight from the pictures. */
int hvk_planes_refresh(hvk_engine_t *e, const int32_t *slots, int n);

/* Pictures that change: where a block shows mostly NEW pictures (at least half of its frames) and the configuration allows
 * it -- PAL colour at 1024 samples per line with the video filter: the metric configuration's geometry -- the block is
 * rendered from the pixels in one kernel (hvk_fused.hip) and the planes are not made at all: they would be written once and
 * read once. (Blocks whose levels are computed -- pictures of many colours, hvk_set_levels() -- keep the planes: faster, measured.) HVK_FUSED=0 in the environment keeps the planes, =1 takes the one kernel for every block with a new picture.
 * hvk_fused_launches(): how many launches went that way. */
int64_t hvk_fused_launches(const hvk_engine_t *e);

int hvk_timing_enable(hvk_engine_t *e, int on);
/* The names of the kernels a launch of this configuration enqueues, as a profiler prints them,
 * separated by ';' -- one name where the per-sample path runs as one kernel from picture planes (the
 * plain configurations; HVK_DIRECT=0 in the environment keeps the kernel pair), else raster
 * [; resampler] ; filter. With one kernel hvk_timing_read() reports its time as kernel 1 and nothing
 * for kernel 0. */
int hvk_kernel_names(const hvk_engine_t *e, char *buf, int n);
/* ... and the whole plan: what runs per uploaded picture, per staged block, per launch and behind it, one line each. */
int hvk_kernel_plan(const hvk_engine_t *e, char *buf, int n);
int hvk_timing_read(hvk_engine_t *e, int which, double *avg_ms, int64_t *launches);

/* ---- host tables (for parity tests): same names as the oracle's ---- */
long hvk_table(const hvk_engine_t *e, const char *name, void *dst, long max_bytes);

/* ---- stage taps (parity tests): the raster stream (int16 I) of the last
 * render, as produced by the raster kernel, copied from device ---- */
int hvk_fetch_raster(hvk_engine_t *e, int16_t *dst, size_t first, size_t count);

const char *hvk_version(void);

/* ---- several devices: one stream rendered by N engines (BASELINE config 5) --------------------------------------
 *
 * The reference renders on one CPU and hands every line to one sink (src/hacktv.c:1579-1587 -> rf_write,
 * src/rf.c:23-31). A group cuts the stream into blocks of `block_frames` frames; block b is rendered by engine
 * b mod N, each engine on the device named for it (a device may be named more than once: N engines on one GPU). The
 * serial sound chains are handed from engine to engine in process, the 32 kHz source is kept by the group and dealt to
 * the engine whose block draws it, and on 525 lines the picture of the frame before a block reaches the block's engine
 * too (hvk_group.cpp), and the SECAM colour chain's state (the IIR pair and the values behind the last line: 40 bytes,
 * hvk_secam_state_export / _import) travels with the blocks like the sound chains' does. Configurations that are one chain
 * over every SAMPLE of the stream (FM video, --pixelrate pairs with frames of two lengths, passthru, raw baseband,
 * sound-in-syncs) are refused for N > 1.
 *
 * A block: hvk_group_frame_upload() for its pictures (frame i of the block -> slot i of the block's engine),
 * hvk_group_audio_write() while hvk_group_audio_needed() > 0, anything per frame (teletext packets, caption pairs)
 * on hvk_group_block_engine(), then hvk_group_stage() and hvk_group_launch(). The stream's samples come back either
 *   (i)  host-direct: hvk_fetch_async() on the block's engine, straight into the block's place in the caller's
 *        page-locked stream buffer -- N devices, N PCIe links, the shape a host rf_* sink wants; or
 *   (ii) gathered on one device: hvk_group_gather() after a round of N blocks -- grouped ncclSend / ncclRecv from C
 *        (librccl is loaded on first use) between distinct devices, device-to-device copies between engines that share
 *        one; HVK_GATHER=peer takes hipMemcpyPeerAsync instead (every sender pushing its block on a stream of its own),
 *        which is also what a machine without a usable librccl falls back to (hvk_group_gather_backend() says which was
 *        taken; hvk_rccl_probe() whether librccl.so.1 loads and has the seven entry points -- no device needed).
 * Refused for N > 1: --interlace (a picture per field; a group of one engine takes it), the configurations above and
 * sound-in-syncs. */
typedef struct hvk_group hvk_group_t;
int hvk_group_open(hvk_group_t **g, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate,
                   const int *devices, int ndevices, int block_frames);
void hvk_group_close(hvk_group_t *g);
int hvk_group_size(const hvk_group_t *g);
int hvk_group_block_frames(const hvk_group_t *g);
hvk_engine_t *hvk_group_engine(hvk_group_t *g, int i);
hvk_engine_t *hvk_group_block_engine(hvk_group_t *g);      /* the engine the next block goes to */
int hvk_group_block_index(const hvk_group_t *g);           /* ... and its index */
int64_t hvk_group_next_frame(const hvk_group_t *g);        /* the stream's next frame (frames launched so far) */
int hvk_group_frame_upload(hvk_group_t *g, int frame_in_block, const uint32_t *fb, int width, int height,
                           int pixel_stride, int line_stride, int interlaced);
/* The picture of slot from_slot of engine `from` into slot `slot` of engine e, device to device (any two devices;
 * engines of one configuration). hvk_group_stage() uses it on 525 lines: the picture the block's last frame shows
 * goes to the next block's engine, which needs it within its first samples' filter reach. */
int hvk_frame_copy(hvk_engine_t *e, int slot, hvk_engine_t *from, int from_slot);
int hvk_group_audio_write(hvk_group_t *g, const int16_t *stereo, size_t nsamples);
size_t hvk_group_audio_needed(hvk_group_t *g, int nframes);
int hvk_group_stage(hvk_group_t *g, int nframes, const int32_t *slots);
int hvk_group_launch(hvk_group_t *g, void *d_iq);          /* returns the index of the engine that renders the block */
int hvk_group_gather(hvk_group_t *g, int root, void *d_root, size_t samples);
const char *hvk_group_gather_backend(const hvk_group_t *g);
int hvk_rccl_probe(char *msg, size_t len);

/* What a group asks of an engine: the HIP stream it launches on; whether the last line of a frame shows picture (525
 * lines: a block's first frame needs the picture of the frame before); whether the stream is one serial chain that
 * cannot be cut into blocks for several engines. */
void *hvk_engine_stream(hvk_engine_t *e);
int hvk_last_line_shows_picture(const hvk_engine_t *e);
int hvk_stream_is_one_chain(const hvk_engine_t *e);
/* SECAM: the colour chain's state between two frames -- the pre-emphasis IIR's two doubles, which the reference never resets
 * (src/fir.c:721-735), and the values the FM loop's last steps leave behind a line (src/video.c:3202-3229) -- with the number
 * of the frame that comes next: what the engine of block b hands to the engine of block b + 1 (hvk_group_stage does; 40
 * bytes). Export waits for the engine's stage to be through; import makes the engine's next stage start there, at that frame. */
size_t hvk_secam_state_size(const hvk_engine_t *e);
int hvk_secam_state_export(hvk_engine_t *e, void *buf, size_t bytes);
int hvk_secam_state_import(hvk_engine_t *e, const void *buf, size_t bytes);

/* Two 64-bit sums over samples [first, first + count) of the last render, read as little-endian uint32 words w[i]
 * (an I/Q pair each): sums[0] = sum w[i], sums[1] = sum (i + 1) w[i], modulo 2^64 -- computed on the device, 16 bytes
 * cross PCIe. A position-sensitive check of a whole block where fetching it to hash it would take longer than
 * rendering it (the one-hour run of BASELINE config 5: tests/golden/ref_hour.json holds the reference's sums). */
int hvk_block_sums(hvk_engine_t *e, size_t first, size_t count, uint64_t sums[2]);

#ifdef __cplusplus
}
#endif

#endif
