/* oracle/ref_probe.c -- TEST INFRASTRUCTURE (not product code).
 *
 * A thin probe linked with the UNMODIFIED reference objects (oracle/Makefile,
 * target _ref/libhacktv_ref.so). It drives the reference engine in-process the
 * way the reference's own main() does (src/hacktv.c:1077-1587: preset lookup,
 * flag overrides, vid_init, s.vid.av set-up, av_test_open, vid_next_line loop)
 * and exposes, through a C ABI that tests/ load with ctypes:
 *
 *   - the lines the reference emits (int16 I/Q pairs), and
 *   - the tables vid_init() built (src/video.c:3812-4162), so the product's
 *     host table builder and the oracle restatement can be compared entry for
 *     entry with the reference's own.
 *
 * Only tests/ and the golden-vector generator use this. It needs
 * /root/reference at BUILD time only; the built .so travels to the GPU box.
 */
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <unistd.h>
#include "hacktv.h"
#include "../hacktv_amd/csrc/shim/hvk_shim_depth.h"

#define REF_FLAG_FILTER   (1 << 0)
#define REF_FLAG_NOAUDIO  (1 << 1)
#define REF_FLAG_NONICAM  (1 << 2)
#define REF_FLAG_NOCOLOUR (1 << 3)
#define REF_FLAG_INTERLACE (1 << 4)
#define REF_FLAG_A2STEREO (1 << 5)
#define REF_FLAG_CC608    (1 << 6)
#define REF_FLAG_WSS_AUTO (1 << 7)
#define REF_FLAG_ACP      (1 << 8)
#define REF_FLAG_VITS     (1 << 9)
#define REF_FLAG_VITC     (1 << 10)
#define REF_FLAG_SVIDEO   (1 << 11)
#define REF_FLAG_SECAM_FID (1 << 12)
#define REF_FLAG_SIS      (1 << 13)

typedef struct {
	vid_t vid;
	int open;
	long rendered;
	int16_t *own_cb, *pinned_cb;    /* ref_pin_ghost() */
} ref_probe_t;

/* Settings main() takes from --gamma / --level / --invert-video / --volume (src/hacktv.c:1179-1184,
 * :1431-1432), applied by the NEXT ref_open(); 0 leaves the preset's value */
static struct { double gamma, level; int invert, volume; } _override;

void ref_override(double gamma, double level, int invert, int volume)
{
	_override.gamma = gamma;
	_override.level = level;
	_override.invert = invert;
	_override.volume = volume;
}

/* --offset, --swap-iq, --wss <mode>, --secam-field-id-lines (src/hacktv.c:1428-1437, :1341-1356), applied by the NEXT ref_open() */
static struct { long long offset; int swap_iq, fid_lines; char wss[32]; } _override2;

void ref_override2(long long offset, int swap_iq, const char *wss, int fid_lines)
{
	_override2.offset = offset;
	_override2.swap_iq = swap_iq;
	_override2.fid_lines = fid_lines;
	_override2.wss[0] = 0;
	if(wss) strncpy(_override2.wss, wss, sizeof(_override2.wss) - 1);
}

/* --raw-bb-file <path> --raw-bb-blanking <n> --raw-bb-white <n>, --passthru <path> (src/hacktv.c:1158-1171, :1428-1430),
 * applied by the NEXT ref_open(); the paths are kept by the reference */
static struct { char raw_bb[256], passthru[256]; int blanking, white; } _override3;

void ref_override_files(const char *raw_bb, int blanking, int white, const char *passthru)
{
	memset(&_override3, 0, sizeof(_override3));
	if(raw_bb) strncpy(_override3.raw_bb, raw_bb, sizeof(_override3.raw_bb) - 1);
	if(passthru) strncpy(_override3.passthru, passthru, sizeof(_override3.passthru) - 1);
	_override3.blanking = blanking;
	_override3.white = white;
}

ref_probe_t *ref_open(const char *mode, unsigned int sample_rate, unsigned int pixel_rate, int flags, const char *teletext)
{
	const vid_configs_t *vc;
	vid_config_t conf;
	ref_probe_t *p;

	for(vc = vid_configs; vc->id != NULL; vc++)
	{
		if(strcmp(mode, vc->id) == 0) break;
	}
	if(vc->id == NULL) return(NULL);

	memcpy(&conf, vc->conf, sizeof(vid_config_t));

	/* The same overrides main() applies for these flags
	 * (src/hacktv.c:1126-1171, :1412-1415, :1431) */
	if(flags & REF_FLAG_NOCOLOUR)
	{
		if(conf.colour_mode == VID_PAL || conf.colour_mode == VID_SECAM || conf.colour_mode == VID_NTSC)
		{
			conf.colour_mode = VID_NONE;
		}
	}
	if(flags & REF_FLAG_NOAUDIO)
	{
		conf.fm_mono_level = conf.fm_left_level = conf.fm_right_level = 0;
		conf.am_audio_level = conf.nicam_level = conf.dance_level = 0;
		conf.fm_mono_carrier = conf.fm_left_carrier = conf.fm_right_carrier = 0;
		conf.nicam_carrier = conf.dance_carrier = conf.am_mono_carrier = 0;
	}
	if(flags & REF_FLAG_NONICAM)
	{
		conf.nicam_level = 0;
		conf.nicam_carrier = 0;
	}
	if(flags & REF_FLAG_FILTER) conf.vfilter = 1;
	/* src/hacktv.c:1121-1124, :1174-1177, :1341-1397, :1401-1410 */
	if(flags & REF_FLAG_INTERLACE) conf.interlace = 1;
	if(flags & REF_FLAG_A2STEREO) conf.a2stereo = 1;
	if(flags & REF_FLAG_CC608) conf.cc608 = 1;
	if(flags & REF_FLAG_WSS_AUTO) conf.wss = "auto";
	if(flags & REF_FLAG_ACP) conf.acp = 1;
	if(flags & REF_FLAG_VITS) conf.vits = 1;
	if(flags & REF_FLAG_VITC) conf.vitc = 1;
	/* src/hacktv.c:1136-1147, :1417-1425, :1436 */
	if(flags & REF_FLAG_SVIDEO) conf.s_video = 1;
	if(flags & REF_FLAG_SECAM_FID) conf.secam_field_id = 1;
	if(flags & REF_FLAG_SIS) conf.sis = "dcsis";
	if(teletext && teletext[0]) conf.teletext = (char *) teletext;
	conf.volume = 1.0 * 256 + 0.5;
	if(_override.gamma > 0) conf.gamma = _override.gamma;
	if(_override.level > 0) conf.level *= _override.level;
	if(_override.invert) conf.invert_video = 1;
	if(_override.volume > 0) conf.volume = _override.volume;
	memset(&_override, 0, sizeof(_override));
	if(_override2.offset) conf.offset = _override2.offset;
	if(_override2.swap_iq) conf.swap_iq = 1;
	if(_override2.fid_lines) conf.secam_field_id_lines = _override2.fid_lines;
	if(_override3.raw_bb[0])
	{
		conf.raw_bb_file = strdup(_override3.raw_bb);
		conf.raw_bb_blanking_level = _override3.blanking;
		conf.raw_bb_white_level = _override3.white;
	}
	if(_override3.passthru[0]) conf.passthru = strdup(_override3.passthru);
	memset(&_override3, 0, sizeof(_override3));
	if(_override2.wss[0]) conf.wss = strdup(_override2.wss);     /* (the reference keeps the pointer; a few bytes per open, never freed) */

	p = calloc(1, sizeof(ref_probe_t));
	if(!p) return(NULL);

	if(vid_init(&p->vid, sample_rate, pixel_rate, &conf) != VID_OK)
	{
		free(p);
		return(NULL);
	}

	/* src/hacktv.c:1503-1518 */
	p->vid.av = (av_t) {
		.frame_rate = (r64_t) {
			.num = p->vid.conf.frame_rate.num * (p->vid.conf.interlace ? 2 : 1),
			.den = p->vid.conf.frame_rate.den,
		},
		.display_aspect_ratios = { p->vid.conf.frame_aspects[0], p->vid.conf.frame_aspects[1] },
		.fit_mode = AV_FIT_STRETCH,
		.width = p->vid.active_width,
		.height = p->vid.conf.active_lines,
		.sample_rate = (r64_t) { HACKTV_AUDIO_SAMPLE_RATE, 1 },
	};

	/* src/hacktv.c:1520-1526: the dimensions change places where the lines are scanned vertically */
	if((p->vid.conf.frame_orientation & 3) == VID_ROTATE_90 || (p->vid.conf.frame_orientation & 3) == VID_ROTATE_270)
	{
		p->vid.av.width = p->vid.conf.active_lines;
		p->vid.av.height = p->vid.active_width;
	}

	if(av_test_open(&p->vid.av) != AV_OK)
	{
		vid_free(&p->vid);
		free(p);
		return(NULL);
	}

	p->open = 1;
	memset(&_override2, 0, sizeof(_override2));
	return(p);
}

/* ---- a source of the caller's own: frames and audio handed to the reference through its
 * av_* callbacks (src/av.h:57-75), so that it can be run on pictures and sound the built-in test
 * source does not have (moving pictures, saturated colours, loud audio, caption pairs) ---- */
typedef struct {
	const uint32_t *frames;
	int nframes, width, height, interlaced, pos;
	const uint8_t *cc;          /* nframes x 2 caption bytes or NULL */
	int par_num, par_den;
	const int16_t *audio;
	long nsamples, apos;
} _src_t;

static _src_t _src;
static unsigned _blank_mask;

/* frames (by their place in the stream, the first 32) the source has no picture for; applies to the next ref_set_source() */
void ref_blank_frames(unsigned mask) { _blank_mask = mask; }

static int _src_read_video(void *ctx, av_frame_t *frame)
{
	_src_t *c = ctx;
	const int i = c->pos % c->nframes;
	if(c->pos < 32 && ((_blank_mask >> c->pos) & 1))
	{
		/* no picture for this frame: what av_read_video() hands out without a source (src/av.c:50-53) */
		av_frame_init(frame, 0, 0, NULL, 0, 0);
		c->pos++;
		return(AV_OK);
	}
	av_frame_init(frame, c->width, c->height, (uint32_t *) c->frames + (size_t) i * c->width * c->height, 1, c->width);
	frame->interlaced = c->interlaced;
	frame->pixel_aspect_ratio = (r64_t) { c->par_num, c->par_den };
	if(c->cc) { frame->cc608[0] = c->cc[i * 2]; frame->cc608[1] = c->cc[i * 2 + 1]; }
	c->pos++;
	return(AV_OK);
}

static int _src_read_audio(void *ctx, int16_t **samples, size_t *nsamples)
{
	/* the whole loop at a time, like src/av_test.c:54-60 */
	_src_t *c = ctx;
	if(!c->audio || c->nsamples <= 0) return(AV_EOF);
	*samples = (int16_t *) c->audio;
	*nsamples = c->nsamples;
	return(AV_OK);
}

static int _src_close(void *ctx) { return(AV_OK); }

/* Replace the test source. frames: nframes x height x width RGBx, shown in turn (one per frame, one
 * per field with --interlace); audio: nsamples stereo pairs at 32 kHz, looped. Call before the first line. */
void ref_set_source(ref_probe_t *p, const uint32_t *frames, int nframes, int width, int height, int interlaced,
                    int par_num, int par_den, const uint8_t *cc, const int16_t *audio, long nsamples)
{
	av_close(&p->vid.av);
	_src = (_src_t) { .frames = frames, .nframes = nframes, .width = width, .height = height, .interlaced = interlaced,
	                  .cc = cc, .par_num = par_num, .par_den = par_den, .audio = audio, .nsamples = nsamples };
	p->vid.av.av_source_ctx = &_src;
	p->vid.av.read_video = _src_read_video;
	p->vid.av.read_audio = _src_read_audio;
	p->vid.av.close = _src_close;
}

/* The reference's malloc() in this build (oracle/Makefile: -Dmalloc=ref_zmalloc for its translation units): zeroed memory. The
 * reference reads some of what it allocates before it has written it -- SECAM's chrominance buffer is malloc'd (src/video.c:4156),
 * and the colour process's first two invocations, on the never-emitted slots in front of line 1, average their cells with the
 * buffer's upper half (:3160) before line 1 clears it (:3095): what those two lines leave in the pre-emphasis IIR and behind the
 * line, and with it the stream's first line with a sub-carrier, is whatever the allocator handed out. In the CLI's fresh heap that
 * is zeros (what the oracle and the engine assume, SURVEY H2's kind); in a test process that has freed megabytes before it is not
 * (tools/fuzz_oracle_ref.py seed 60221 found it; clang's MemorySanitizer named the read: tools/ref_msan.sh). */
void *ref_zmalloc(size_t n)
{
	return(calloc(1, n ? n : 1));
}

/* The chroma filter reads up to ataps / 2 samples per channel past the 2 * width chrominance buffer: bytes
 * that belong to whatever the allocator put behind it -- in some heap layouts an object of the reference's own that
 * changes while it runs, and then no read-out before or after the run describes what the filter saw. For comparisons
 * that are not about the heap: move the buffer into one of the probe's with `n` (<= 64) samples of the caller's choosing
 * behind it -- normally the ones just read out with ref_table("chroma_ghost") -- so that they stay what they were. Only
 * the pointer changes hands (every use goes through s->chrominance_buffer); ref_close() gives the reference its own back. */
void ref_pin_ghost(ref_probe_t *p, const int16_t *ghost, int n)
{
	vid_t *s = &p->vid;
	int16_t *nb;
	if(!s->chrominance_buffer || p->pinned_cb || n < 0 || n > 64) return;
	nb = calloc((size_t) 2 * s->width + 64, sizeof(int16_t));
	if(!nb) return;
	memcpy(nb, s->chrominance_buffer, sizeof(int16_t) * 2 * s->width);
	memcpy(nb + 2 * s->width, ghost, sizeof(int16_t) * n);
	p->own_cb = s->chrominance_buffer;
	p->pinned_cb = nb;
	s->chrominance_buffer = nb;
}

void ref_close(ref_probe_t *p)
{
	if(p && p->pinned_cb)
	{
		p->vid.chrominance_buffer = p->own_cb;
		free(p->pinned_cb);
		p->pinned_cb = NULL;
	}
	if(!p) return;
	/* vid_free() is not usable from a long-lived test process: its worker
	 * shutdown (src/video.c:4714-4721 against :3590-3613) races twice -- the
	 * main thread stops taking part in the barrier as soon as it reads
	 * nthreads == 0, and the workers decrement nthreads without a lock
	 * (:3607), so with two or more workers the count can stick at 1 and a
	 * worker waits at the barrier for ever (observed here). The reference CLI
	 * survives because the process exits right after. The probe therefore
	 * leaves the workers parked at their barrier (they hold no lock and never
	 * run again), closes the source and releases the large tables by hand. */
	if(p->open)
	{
		vid_t *s = &p->vid;
		if(s->nthreads == 0)
		{
			vid_free(s);
			free(p);
		}
		else
		{
			/* let workers still finishing the line after the last barrier park */
			usleep(50000);
			av_close(&s->av);
			free(s->yuv_level_lookup);
			free(s->colour_lookup);
			free(s->fm_mono.lut);
			free(s->fm_secam.lut);
			free(s->fm_secam_bell);
			/* p itself is NOT freed: the parked workers wait on the barrier
			 * that lives inside it */
		}
	}
	else free(p);
}

/* The lines the reference's pipeline holds back -- the distance in its ring of output lines between the
 * buffer the raster (or the raw baseband reader) writes and the one handed out (src/video.c:3578,
 * :4675-4688, :2873, :2408) -- and what the shim's own count says for the same vid_t
 * (hacktv_amd/csrc/shim/hvk_shim_depth.h). The video filter's delay is one line at every rate in use. */
int ref_pipeline_depths(ref_probe_t *p, int32_t *reference, int32_t *shim)
{
	const vid_t *s = &p->vid;
	*reference = s->olines - (s->raw_bb_file ? 1 : 2);
	*shim = hvk_shim_pipeline_depth(s, 1);
	return(0);
}

/* Geometry and levels, in a fixed order the python side names */
int ref_info(ref_probe_t *p, int32_t *out, int n)
{
	const vid_t *s = &p->vid;
	int32_t v[] = {
		s->width, s->half_width, s->active_width, s->active_left,
		s->conf.lines, s->conf.active_lines,
		s->white_level, s->black_level, s->blanking_level, s->sync_level,
		(int32_t) s->colour_lookup_width, s->burst_left, s->burst_width,
		s->burst_phase.i, s->burst_phase.q,
		s->chrominance_fir.ataps, s->olines, s->max_width,
		s->fm_mono.level, s->nicam.ntaps, s->nicam.sps, s->nicam.dsl, s->nicam.decimation,
		(int32_t) (s->nicam.cc_end - s->nicam.cc_start),
		s->am_mono.level, s->am_mono.delta.i, s->am_mono.delta.q,
		s->fm_secam.level, s->fm_secam_dmin[0], s->fm_secam_dmax[0], s->fm_secam_dmin[1], s->fm_secam_dmax[1],
		s->secam_fsync_level, s->secam_field_id_lines,
	};
	int c = sizeof(v) / sizeof(v[0]);
	if(n < c) c = n;
	memcpy(out, v, c * sizeof(int32_t));
	return(sizeof(v) / sizeof(v[0]));
}

/* Copy the next emitted line. Returns its width in samples (pairs), or -1 */
int ref_next_line(ref_probe_t *p, int16_t *iq, int max_samples, int32_t *frame, int32_t *line)
{
	vid_line_t *l = vid_next_line(&p->vid);
	if(l == NULL) return(-1);
	p->rendered++;
	if(l->width > max_samples) return(-2);
	memcpy(iq, l->output, sizeof(int16_t) * 2 * l->width);
	if(frame) *frame = l->frame;
	if(line) *line = l->line;
	return(l->width);
}

/* Render n whole lines back to back into iq. Returns samples written */
long ref_render_lines(ref_probe_t *p, int16_t *iq, long nlines)
{
	long i, o = 0;
	for(i = 0; i < nlines; i++)
	{
		vid_line_t *l = vid_next_line(&p->vid);
		if(l == NULL) break;
		p->rendered++;
		memcpy(iq + o * 2, l->output, sizeof(int16_t) * 2 * l->width);
		o += l->width;
	}
	return(o);
}

static long _copy(void *dst, long max_bytes, const void *src, long bytes)
{
	if(src == NULL) return(0);
	if(dst == NULL) return(bytes);
	if(bytes > max_bytes) bytes = max_bytes;
	memcpy(dst, src, bytes);
	return(bytes);
}

static long _vbilut_bytes(const vbidata_lut_t *lut)
{
	const int16_t *p = (const int16_t *) lut;
	long n = 0;
	if(!lut) return(0);
	while(p[n] != -1) n += 2 + p[n];
	return((n + 1) * sizeof(int16_t));
}

/* Table dump: returns the table's size in bytes (copying up to max_bytes if
 * dst != NULL), 0 if the table does not exist in this mode, -1 if unknown. */
long ref_table(ref_probe_t *p, const char *name, void *dst, long max_bytes)
{
	vid_t *s = &p->vid;

	if(strcmp(name, "syncs") == 0)
		return(_copy(dst, max_bytes, s->syncs, _vbilut_bytes(s->syncs)));
	if(strcmp(name, "yuv") == 0)
		return(_copy(dst, max_bytes, s->yuv_level_lookup, 0x1000000L * sizeof(_yuv16_t)));
	if(strcmp(name, "colour_lookup") == 0)
		return(_copy(dst, max_bytes, s->colour_lookup, s->colour_lookup ? (long) (s->colour_lookup_width + s->width) * sizeof(cint16_t) : 0));
	if(strcmp(name, "burst_win") == 0)
		return(_copy(dst, max_bytes, s->burst_win, s->burst_win ? (long) s->burst_width * sizeof(int16_t) : 0));
	if(strcmp(name, "chroma_taps") == 0)
		return(_copy(dst, max_bytes, s->chrominance_fir.itaps, s->chrominance_fir.itaps ? (long) s->chrominance_fir.ntaps * sizeof(int16_t) : 0));
	if(strcmp(name, "secam_iir") == 0)
	{
		/* the pre-emphasis filter's state (ix, iy) as two doubles, and behind them the chrominance buffer (2 * width int16) */
		static unsigned char tmp[16 + 2 * 8192 * 2];
		if(s->conf.colour_mode != VID_SECAM || !s->chrominance_buffer || s->width > 8192) return(0);
		memcpy(tmp, &s->fm_secam_iir.ix, 8);
		memcpy(tmp + 8, &s->fm_secam_iir.iy, 8);
		memcpy(tmp + 16, s->chrominance_buffer, (size_t) 2 * s->width * 2);
		return(_copy(dst, max_bytes, tmp, 16 + (long) 2 * s->width * 2));
	}
	if(strcmp(name, "sis_heap") == 0)
	{
		/* The 8 int16s in front of the sound-in-syncs symbol table on the heap: the burst encoder's first invocation
		 * reads them (src/vbidata.c:211-217 with a slot of no width, oracle_sis.c) */
		return(_copy(dst, max_bytes, s->conf.sis && s->sis.lut ? (const int16_t *) s->sis.lut - 8 : NULL, 8 * sizeof(int16_t)));
	}
	if(strcmp(name, "chroma_buffer") == 0)
	{
		/* the 2 * width chrominance buffer as it stands (SECAM: the line's cells in the lower half, the line before's other
		 * colour difference in the upper) */
		return(_copy(dst, max_bytes, s->chrominance_buffer, s->chrominance_buffer ? (long) 2 * s->width * sizeof(int16_t) : 0));
	}
	if(strcmp(name, "chroma_ghost") == 0)
	{
		/* The int16s that follow the 2*width chrominance buffer on the heap:
		 * the reference's chroma FIR reads ataps/2 of them per channel
		 * (src/fir.c:365-372 with samples = width) */
		return(_copy(dst, max_bytes, s->chrominance_buffer ? s->chrominance_buffer + 2 * s->width : NULL, 32 * sizeof(int16_t)));
	}
	if(strcmp(name, "vfilter_itaps") == 0 || strcmp(name, "vfilter_qtaps") == 0)
	{
		int i;
		for(i = 0; i < s->nprocesses; i++)
		{
			if(strcmp(s->processes[i].name, "vfilter") == 0)
			{
				/* _vid_filter_process_t is private to video.c: { int channels; fir_int16_t fir[2]; } */
				struct { int channels; fir_int16_t fir[2]; } *fp = s->processes[i].arg;
				const int16_t *t = name[8] == 'i' ? fp->fir[0].itaps : fp->fir[0].qtaps;
				return(_copy(dst, max_bytes, t, t ? (long) fp->fir[0].ntaps * sizeof(int16_t) : 0));
			}
		}
		return(0);
	}
	if(strcmp(name, "resampler_taps") == 0)
	{
		int i;
		for(i = 0; i < s->nprocesses; i++)
		{
			if(strcmp(s->processes[i].name, "vresampler") == 0)
			{
				struct { int channels; fir_int16_t fir[2]; } *fp = s->processes[i].arg;
				return(_copy(dst, max_bytes, fp->fir[0].itaps, (long) fp->fir[0].ntaps * sizeof(int16_t)));
			}
		}
		return(0);
	}
	if(strcmp(name, "fm_mono_lut") == 0)
		return(_copy(dst, max_bytes, s->fm_mono.lut, s->fm_mono.lut ? 65536L * sizeof(cint32_t) : 0));
	if(strcmp(name, "fm_video_lut") == 0)
		return(_copy(dst, max_bytes, s->fm_video.lut, s->fm_video.lut ? 65536L * sizeof(cint32_t) : 0));
	if(strcmp(name, "fm_secam_lut") == 0)
		return(_copy(dst, max_bytes, s->fm_secam.lut, s->fm_secam.lut ? 65536L * sizeof(cint32_t) : 0));
	if(strcmp(name, "fm_secam_bell") == 0)
		return(_copy(dst, max_bytes, s->fm_secam_bell, s->fm_secam_bell ? 65535L * sizeof(cint16_t) : 0));
	if(strcmp(name, "fm_secam_fir") == 0)
		return(_copy(dst, max_bytes, s->fm_secam_fir.itaps, s->fm_secam_fir.itaps ? (long) s->fm_secam_fir.ntaps * sizeof(int16_t) : 0));
	if(strcmp(name, "secam_l_fir") == 0)
		return(_copy(dst, max_bytes, s->secam_l_fir.itaps, s->secam_l_fir.itaps ? (long) s->secam_l_fir.ntaps * sizeof(int16_t) : 0));
	if(strcmp(name, "nicam_taps") == 0)
		return(_copy(dst, max_bytes, s->nicam.taps, s->nicam.taps ? (long) s->nicam.ntaps * sizeof(int16_t) : 0));
	if(strcmp(name, "nicam_cc") == 0)
		return(_copy(dst, max_bytes, s->nicam.cc_start, s->nicam.cc_start ? (long) (s->nicam.cc_end - s->nicam.cc_start) * sizeof(cint16_t) : 0));
	if(strcmp(name, "limiter_shape") == 0)
		return(_copy(dst, max_bytes, s->fm_mono.limiter.shape, s->fm_mono.limiter.shape ? (long) s->fm_mono.limiter.width * sizeof(int16_t) : 0));
	if(strcmp(name, "limiter_vtaps") == 0)
		return(_copy(dst, max_bytes, s->fm_mono.limiter.vfir.itaps, s->fm_mono.limiter.vfir.itaps ? (long) s->fm_mono.limiter.vfir.ntaps * sizeof(int32_t) : 0));
	if(strcmp(name, "limiter_ftaps") == 0)
		return(_copy(dst, max_bytes, s->fm_mono.limiter.ffir.itaps, s->fm_mono.limiter.ffir.itaps ? (long) s->fm_mono.limiter.ffir.ntaps * sizeof(int32_t) : 0));
	if(strcmp(name, "teletext_lut") == 0)
		return(_copy(dst, max_bytes, s->tt.lut, s->conf.teletext ? _vbilut_bytes(s->tt.lut) : 0));

	return(-1);
}

/* The test source's frame and audio loop, for building golden input fixtures.
 * Must be called before any line is rendered. */
long ref_test_frame(ref_probe_t *p, uint32_t *dst, long max_pixels)
{
	av_frame_t f;
	long n;
	if(p->vid.av.read_video == NULL) return(-1);
	if(p->vid.av.read_video(p->vid.av.av_source_ctx, &f) != AV_OK) return(-1);
	n = (long) f.width * f.height;
	if(dst)
	{
		if(n > max_pixels) n = max_pixels;
		memcpy(dst, f.framebuffer, n * sizeof(uint32_t));
	}
	return((long) f.width * f.height);
}

long ref_test_audio(ref_probe_t *p, int16_t *dst, long max_samples)
{
	int16_t *a;
	size_t n;
	if(p->vid.av.read_audio == NULL) return(-1);
	if(p->vid.av.read_audio(p->vid.av.av_source_ctx, &a, &n) != AV_OK) return(-1);
	if(dst)
	{
		if((long) n > max_samples) n = max_samples;
		memcpy(dst, a, n * 2 * sizeof(int16_t));
	}
	return((long) n);
}
