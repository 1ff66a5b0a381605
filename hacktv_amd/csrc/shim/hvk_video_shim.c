/* hvk_video_shim.c -- the reference's video.h interface on top of libhvk.
 *
 * This file is what makes the MI355X engine a drop-in for the one hot path:
 * it defines exactly the entry points the reference's main() calls
 * (src/video.h:512-516; call sites src/hacktv.c:1440, :1446, :1581, :1599)
 *
 *     int vid_init(vid_t *s, unsigned int sample_rate, unsigned int pixel_rate, const vid_config_t *conf);
 *     vid_line_t *vid_next_line(vid_t *s);
 *     void vid_free(vid_t *s);
 *     void vid_info(vid_t *s);
 *     size_t vid_get_framebuffer_length(vid_t *s);
 *
 * with the reference's own types, so hacktv.c, av*.c and rf*.c compile and run
 * unchanged. It is built AGAINST THE REFERENCE'S HEADERS (-I<reference>/src):
 * nothing of the reference is copied here. INTEGRATION.md shows the two build
 * recipes (replace video.c's engine entry points / link libhvk).
 *
 * Behaviour mirrored from the reference:
 *  - vid_init() returns VID_OK / VID_ERROR / VID_OUT_OF_MEMORY and fills the
 *    vid_t members main() reads afterwards: sample_rate, conf, active_width
 *    (src/hacktv.c:1452, :1493, :1503-1518), plus width / levels for vid_info.
 *  - video is pulled with av_eof / av_read_video at the start of every frame
 *    (src/video.c:4873-4897), audio with av_read_audio as the 32 kHz tick needs
 *    it (src/video.c:3278-3286). The reference pulls video on the main thread and audio on
 *    its audio thread; here one worker thread pulls both, in stream order.
 *  - vid_next_line() hands out one scanline of interleaved int16 I/Q, valid
 *    until the next call, with frame / line numbers starting at 1, and returns
 *    NULL at the end of the source (src/video.c:4936-4952).
 *  - configurations this engine does not render are REFUSED with VID_ERROR and
 *    a message; nothing is silently dropped.
 *
 * Difference: frames are rendered HVK_BATCH at a time on the GPU (environment
 * variable, default 8) and read back into a host buffer the lines point into;
 * a worker thread prepares the next batch (source pulls, host pre-passes, render,
 * read-back) while the lines of the current one are handed out.
 * line->audio / audio_len (src/video.c:3445-3447; what a sink with rf_write_audio() is handed, src/hacktv.c:1586): the
 * reference's audio process writes every 32 kHz stereo sample it has drawn, after the volume control, into a FIFO of
 * 320-sample blocks and reads that FIFO once per line -- a block can be read once the writer has left it, i.e. from the
 * line on which the 321st, 641st, ... sample is drawn (src/fifo.c:230-290). The shim keeps the drawn samples and hands
 * out the same blocks on the same lines.
 *
 * Teletext: which packet goes on which line -- the TTI page store, the magazine
 * scheduler, the wall clock (src/teletext.c:489-990) -- is host control logic and
 * stays the reference's own code: the shim calls tt_init() and, for the 32 VBI
 * lines of every frame in line order (src/teletext.c:1222-1224), tt_next_packet();
 * the engine shapes and adds the symbols on the GPU (hvk_teletext_packets()).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <pthread.h>
#include <time.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include "video.h"          /* the reference's */
#include "hacktv_amd.h"
#include "hvk_shim_depth.h"

#define SHIM_AUDIO_RATE 32000    /* HACKTV_AUDIO_SAMPLE_RATE, src/hacktv.h:31 */
#define SHIM_AUDIO_BLOCK 320     /* stereo samples per block of the reference's audio FIFO, src/video.c:4342 */

typedef struct {
	hvk_engine_t *e;
	/* HVK_DEVICES=0,1,...: the stream's batches go round a group of engines, one per device named (hvk_group_*): batch b
	 * on engine b mod N, the sound chains handed on in process, every engine's read-back straight into the batch's
	 * page-locked buffer -- N devices, N PCIe links into the one host stream the rf_* sink reads. `e` is then engine 0
	 * (tables, line widths) */
	hvk_group_t *g;
	hvk_engine_t *fe[2];    /* the engine that renders (and reads back) the batch in buf[i] */
	hvk_info_t info;
	int batch;              /* frames per GPU launch */
	int16_t *iq;            /* the batch being handed out: batch * frame_samples pairs */
	int have;               /* frames rendered in iq */
	/* the next batch is pulled, rendered and fetched by a worker thread while this one goes out */
	int16_t *buf[2];
	int source_closed;      /* main() has closed a source: the stream ends there (see _hooked_close) */
	int pinned;             /* buf[] came from hvk_host_alloc() */
	int ticket[2];          /* hvk_fetch_async() ticket of the buffer's read-back; the consumer waits for it */
	/* HVK_SHIM_STATS=1: where the time of both threads went, printed by vid_free() */
	int stats;
	double t_first;
	double t_pull, t_audio, t_render, t_fetch, t_worker_idle, t_consumer_wait;
	int ready[2];           /* 0: free for the worker, 1: filled */
	int count[2];           /* frames in the buffer; 0: the source has ended, < 0: failure */
	int cur;                /* buffer being handed out, -1: none yet */
	int stop;
	pthread_t worker;
	int worker_on;
	pthread_mutex_t lock;
	pthread_cond_t cond;
	int64_t frames_pulled;  /* frames taken from the source so far (worker side) */
	int frame_in_batch;
	int line;               /* next line of the current frame, 0-based */
	int64_t frames_done;    /* frames handed out completely */
	int ended;              /* source has ended: no more batches */
	int frame_in_batch_pull;
	int passthru_primed;    /* the start-up line's share of the passthru source has been queued */
	uint8_t held[1250];     /* lines the inserters other than teletext write to (hvk_vbi_lines_held) */
	int32_t *widths;        /* widths of the lines of the frame being handed out (they vary with --pixelrate) */
	size_t line_at;         /* sample offset of the next line in iq */
	int16_t *passbuf;
	int64_t raw_fed;        /* samples of the raw baseband file queued so far */
	/* caption pairs waiting for a line 21, like src/cc608.c:26-96: 128 pairs, empty ones never queued */
	uint8_t cc_fifo[256];
	int cc_in, cc_out, cc_len;
	int64_t audio_drawn;    /* 32 kHz samples taken from the source when no carrier needs them */
	int last[2];            /* buffer holds the last frames of the source */
	int end_drop;           /* lines of the last frame the reference never hands out (see vid_next_line) */
	/* line->audio: the 32 kHz samples drawn so far after the volume control (pairs; aud[0] is pair aud_base), the stream
	 * position the audio process has reached when a line goes out, the blocks handed out */
	int16_t *aud;
	size_t aud_len, aud_cap;
	int64_t aud_base;
	int volume;
	int64_t out_pos;
	int64_t ticks, tick_rem;    /* out_pos * 32000 = ticks * sample_rate + tick_rem */
	int64_t aud_blocks_read;
	int16_t aud_out[SHIM_AUDIO_BLOCK * 2];
	vid_line_t out;
} shim_t;

/* HVK_SHIM_STATS: SIGUSR1 shows where the worker thread is (a diagnostic for a drop-in that does not leave) */
static pthread_t _dbg_worker;
static volatile int _dbg_have_worker;
static void _dbg_usr1(int sig)
{
	void *bt[48];
	(void) sig;
	if(_dbg_have_worker && !pthread_equal(pthread_self(), _dbg_worker)) { pthread_kill(_dbg_worker, SIGUSR1); return; }
	backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
}

/* The reference's main() closes the source -- av_close(), src/hacktv.c:1596 -- as soon as its line loop ends (end of
 * the source, or a signal), BEFORE vid_free(); its own vid_next_line() reads the source only inside the call, so that
 * is safe there. Here a worker thread reads ahead: it has to be off the source before the source's close callback
 * frees what it reads (a test card freed under the worker's copy is a segmentation fault -- which hacktv's handler
 * turns into an endless loop). The shim therefore puts its own close callback in front of the source's: it stops the
 * worker, then lets the source close.
 * hacktv plays its arguments one after the other (and over again with --repeat), and the reference's line pipeline
 * carries the last lines of one source over into the next (the lines vid_next_line() withholds at the end, src/video.c
 * :4876). That hand-over is not reproduced: the stream ends with the first source, and vid_next_line() says so once if
 * it is asked for more. */
#define SHIM_HOOKS 8
static struct { void *ctx; vid_t *s; av_close_t close; } _hooks[SHIM_HOOKS];
static pthread_mutex_t _hooks_lock = PTHREAD_MUTEX_INITIALIZER;

static void _worker_stop(vid_t *s);
static void _source_closed(vid_t *s);

static int _hooked_close(void *ctx)
{
	av_close_t orig = NULL;
	vid_t *s = NULL;
	int i;

	pthread_mutex_lock(&_hooks_lock);
	for(i = 0; i < SHIM_HOOKS; i++)
	{
		if(_hooks[i].s && _hooks[i].ctx == ctx)
		{
			orig = _hooks[i].close;
			s = _hooks[i].s;
			_hooks[i].s = NULL;
			break;
		}
	}
	pthread_mutex_unlock(&_hooks_lock);

	if(s)
	{
		_worker_stop(s);
		_source_closed(s);
	}
	return(orig ? orig(ctx) : AV_OK);
}

static int _hook_source(vid_t *s)
{
	int i, r = -1;
	if(s->av.close == _hooked_close) return(0);
	pthread_mutex_lock(&_hooks_lock);
	for(i = 0; i < SHIM_HOOKS; i++)
	{
		if(!_hooks[i].s)
		{
			_hooks[i].s = s;
			_hooks[i].ctx = s->av.av_source_ctx;
			_hooks[i].close = s->av.close;
			s->av.close = _hooked_close;
			r = 0;
			break;
		}
	}
	pthread_mutex_unlock(&_hooks_lock);
	return(r);
}

static void _unhook_source(vid_t *s)
{
	int i;
	pthread_mutex_lock(&_hooks_lock);
	for(i = 0; i < SHIM_HOOKS; i++)
	{
		if(_hooks[i].s == s)
		{
			if(s->av.close == _hooked_close) s->av.close = _hooks[i].close;
			_hooks[i].s = NULL;
		}
	}
	pthread_mutex_unlock(&_hooks_lock);
}

static double _now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return(ts.tv_sec + ts.tv_nsec * 1e-9);
}

/* keep what av_read_audio() delivered, scaled like src/video.c:3293-3298 (worker side) */
static int _aud_append(shim_t *m, const int16_t *a, size_t pairs)
{
	size_t i;
	pthread_mutex_lock(&m->lock);
	if(m->aud_len + pairs > m->aud_cap)
	{
		size_t cap = (m->aud_len + pairs) * 2 + 4096;
		int16_t *p = realloc(m->aud, cap * 2 * sizeof(int16_t));
		if(!p) { pthread_mutex_unlock(&m->lock); return(-1); }
		m->aud = p;
		m->aud_cap = cap;
	}
	for(i = 0; i < pairs * 2; i++)
	{
		const int32_t v = ((int32_t) a[i] * m->volume + 128) >> 8;
		m->aud[m->aud_len * 2 + i] = (int16_t) (v < INT16_MIN ? INT16_MIN : (v > INT16_MAX ? INT16_MAX : v));
	}
	m->aud_len += pairs;
	pthread_mutex_unlock(&m->lock);
	return(0);
}

static void _cc_push(shim_t *m, const uint8_t *pair)
{
	if(m->cc_len >= 256 || ((pair[0] | pair[1]) & 0x7F) == 0) return;
	m->cc_fifo[m->cc_in++] = pair[0];
	m->cc_fifo[m->cc_in++] = pair[1];
	if(m->cc_in >= 256) m->cc_in = 0;
	m->cc_len += 2;
}

static int _cc_pop(shim_t *m, uint8_t *pair)
{
	if(m->cc_len < 2) return(0);
	pair[0] = m->cc_fifo[m->cc_out++];
	pair[1] = m->cc_fifo[m->cc_out++];
	if(m->cc_out >= 256) m->cc_out = 0;
	m->cc_len -= 2;
	return(1);
}

/* vid_t has no member for an engine handle (src/video.h:358-508), and its private ones are the reference engine's to
 * use: the shim keeps its state beside the vid_t, in a small table of (vid_t, state) pairs. */
#define SHIM_INSTANCES 8
static struct { vid_t *s; shim_t *m; } _instances[SHIM_INSTANCES];
static pthread_mutex_t _instances_lock = PTHREAD_MUTEX_INITIALIZER;

static shim_t *_shim(vid_t *s)
{
	int i;
	for(i = 0; i < SHIM_INSTANCES; i++) if(_instances[i].s == s) return(_instances[i].m);
	return(NULL);
}

static int _shim_register(vid_t *s, shim_t *m)
{
	int i, r = -1;
	pthread_mutex_lock(&_instances_lock);
	for(i = 0; i < SHIM_INSTANCES; i++) if(_instances[i].s == s) { _instances[i].m = m; r = 0; break; }
	for(i = 0; r != 0 && i < SHIM_INSTANCES; i++) if(!_instances[i].s) { _instances[i].m = m; _instances[i].s = s; r = 0; }
	pthread_mutex_unlock(&_instances_lock);
	return(r);
}

static void _shim_unregister(vid_t *s)
{
	int i;
	pthread_mutex_lock(&_instances_lock);
	for(i = 0; i < SHIM_INSTANCES; i++) if(_instances[i].s == s) { _instances[i].s = NULL; _instances[i].m = NULL; }
	pthread_mutex_unlock(&_instances_lock);
}

/* Stop the read-ahead: the worker leaves the source and the engine alone from here on. What it had rendered and not
 * handed out is dropped (the caller is leaving, or -- at the end of a source -- there is nothing left) */
static void _worker_stop(vid_t *s)
{
	shim_t *m = _shim(s);
	if(!m || !m->worker_on) return;
	pthread_mutex_lock(&m->lock);
	m->stop = 1;
	pthread_cond_broadcast(&m->cond);
	pthread_mutex_unlock(&m->lock);
	pthread_join(m->worker, NULL);
	m->worker_on = 0;
	m->stop = 0;
	m->ready[0] = m->ready[1] = 0;
	m->cur = -1;
	m->have = 0;
	m->frame_in_batch = 0;
	m->ended = 0;
}

static void _source_closed(vid_t *s)
{
	shim_t *m = _shim(s);
	if(m) m->source_closed = 1;
}

static void _engine_close(shim_t *m)
{
	if(m->g) hvk_group_close(m->g);
	else hvk_close(m->e);
	m->g = NULL;
	m->e = NULL;
}

static int _refuse(const char *what)
{
	fprintf(stderr, "hacktv-amd: %s is not rendered by the MI355X engine (see DESIGN.md, scope)\n", what);
	return(VID_ERROR);
}

static int _translate(hvk_config_t *h, const vid_config_t *c, unsigned int sample_rate, unsigned int pixel_rate)
{
	HVK_CONFIG_INIT(h);

	/* (src/video.h:50-59: the raster types' numbers are the engine's; MAC is a packet multiplex, not a raster) */
	if(c->type != VID_RASTER_625 && c->type != VID_RASTER_525 && c->type != VID_RASTER_405 && c->type != VID_RASTER_819 && c->type != VID_BAIRD_240 &&
	   c->type != VID_BAIRD_30 && c->type != VID_NBTV_32 && c->type != VID_APOLLO_320 && c->type != VID_CBS_405) return(_refuse("this raster type"));
	if(c->modulation == VID_FM && c->fm_energy_dispersal) return(_refuse("FM energy dispersal"));
	if(c->colour_mode != VID_NONE && c->colour_mode != VID_PAL && c->colour_mode != VID_NTSC && c->colour_mode != VID_SECAM &&
	   c->colour_mode != VID_APOLLO_FSC && c->colour_mode != VID_CBS_FSC) return(_refuse("this colour mode"));
	if(c->teletext && c->lines != 625) return(_refuse("teletext on a raster other than 625 lines"));
	if(c->videocrypt || c->videocrypt2 || c->videocrypts || c->syster || c->d11 ||
	   c->systercnr || c->eurocrypt) return(_refuse("a scrambler"));
	if(c->sis && strcmp(c->sis, "dcsis") != 0) return(_refuse("this sound-in-syncs mode"));      /* (so does the reference, src/sis.c:95-103) */
	if(c->fm_left_level > 0 || c->fm_right_level > 0 || c->dance_level > 0) return(_refuse("this audio mode"));
	if(c->raw_bb_file && c->s_video) return(_refuse("raw baseband input with --s-video"));

	h->output_type = c->output_type;
	h->modulation = c->modulation;
	h->video_bw = c->video_bw;
	h->vsb_upper_bw = c->vsb_upper_bw;
	h->vsb_lower_bw = c->vsb_lower_bw;
	h->level = c->level;
	h->video_level = c->video_level;
	h->fm_mono_level = c->fm_mono_level;
	h->am_audio_level = c->am_audio_level;
	h->nicam_level = c->nicam_level;
	h->type = c->type;
	h->frame_rate.num = c->frame_rate.num;
	h->frame_rate.den = c->frame_rate.den;
	h->lines = c->lines;
	h->hline = c->hline;
	h->interlaced = c->interlaced;
	h->interlace = c->interlace;
	h->active_lines = c->active_lines;
	h->hsync_width = c->hsync_width;
	h->vsync_short_width = c->vsync_short_width;
	h->vsync_long_width = c->vsync_long_width;
	h->sync_rise = c->sync_rise;
	h->invert_video = c->invert_video;
	h->white_level = c->white_level;
	h->black_level = c->black_level;
	h->blanking_level = c->blanking_level;
	h->sync_level = c->sync_level;
	h->active_width = c->active_width;
	h->active_left = c->active_left;
	h->gamma = c->gamma;
	h->rw_co = c->rw_co;
	h->gw_co = c->gw_co;
	h->bw_co = c->bw_co;
	h->colour_mode = c->colour_mode == VID_PAL ? HVK_PAL : (c->colour_mode == VID_NTSC ? HVK_NTSC : (c->colour_mode == VID_SECAM ? HVK_SECAM :
	                 (c->colour_mode == VID_APOLLO_FSC ? HVK_APOLLO_FSC : (c->colour_mode == VID_CBS_FSC ? HVK_CBS_FSC : HVK_MONOCHROME))));
	h->fsc_flag_width = c->fsc_flag_width;
	h->fsc_flag_left = c->fsc_flag_left;
	h->fsc_flag_level = c->fsc_flag_level;
	h->frame_orientation = c->frame_orientation;
	h->colour_carrier.num = c->colour_carrier.num;
	h->colour_carrier.den = c->colour_carrier.den;
	h->colour_bw = c->colour_bw;
	h->burst_width = c->burst_width;
	h->burst_left = c->burst_left;
	h->burst_level = c->burst_level;
	h->burst_rise = c->burst_rise;
	h->ev_co = c->ev_co;
	h->eu_co = c->eu_co;
	h->secam_field_id = c->secam_field_id;
	h->secam_field_id_lines = c->secam_field_id_lines;
	h->volume = c->volume;
	h->fm_mono_carrier = c->fm_mono_carrier;
	h->fm_mono_deviation = c->fm_mono_deviation;
	h->fm_mono_preemph = c->fm_mono_preemph;
	h->nicam_carrier = c->nicam_carrier;
	h->nicam_beta = c->nicam_beta;
	h->am_mono_carrier = c->am_mono_carrier;
	h->a2stereo = c->a2stereo;
	h->vfilter = c->vfilter;
	h->s_video = c->s_video;
	h->raw_bb = c->raw_bb_file != NULL;
	h->raw_bb_blanking_level = c->raw_bb_blanking_level;
	h->raw_bb_white_level = c->raw_bb_white_level;
	h->teletext = c->teletext != NULL;
	h->vits = c->vits;
	h->vitc = c->vitc;
	h->acp = c->acp;
	h->cc608 = c->cc608;
	h->sis = c->sis ? 1 : 0;
	if(c->wss)
	{
		/* mode name -> the aspect ratio group of ETSI EN 300 294 with its odd parity bit, as src/wss.c:33-44 */
		static const struct { const char *id; int code; } modes[] = {
			{ "4:3", 0x08 }, { "14:9-letterbox", 0x01 }, { "14:9-top", 0x02 }, { "16:9-letterbox", 0x0B },
			{ "16:9-top", 0x04 }, { "16:9+-letterbox", 0x0D }, { "14:9-window", 0x0E }, { "16:9", 0x07 }, { NULL, 0 },
		};
		int i;
		for(i = 0; modes[i].id && strcasecmp(c->wss, modes[i].id) != 0; i++);
		if(modes[i].id) h->wss = modes[i].code;
		else if(strcasecmp(c->wss, "auto") == 0) h->wss = 0xFF;    /* per frame, from its pixel aspect */
		else return(_refuse("this WSS mode"));
	}
	h->fm_level = c->fm_level;
	h->fm_deviation = c->fm_deviation;
	h->swap_iq = c->swap_iq;
	h->offset = c->offset;
	h->passthru = c->passthru != NULL;

	return(VID_OK);
}

int vid_init(vid_t *s, unsigned int sample_rate, unsigned int pixel_rate, const vid_config_t * const conf)
{
	hvk_config_t hc;
	shim_t *m;
	const char *env;
	int r, device = 0;

	memset(s, 0, sizeof(vid_t));
	s->conf = *conf;

	if((r = _translate(&hc, conf, sample_rate, pixel_rate)) != VID_OK) return(r);

	m = calloc(1, sizeof(shim_t));
	if(!m) return(VID_OUT_OF_MEMORY);

	m->batch = 8;
	if((env = getenv("HVK_BATCH")) && atoi(env) > 0) m->batch = atoi(env);
	if((env = getenv("HVK_DEVICE"))) device = atoi(env);
	{
		/* HVK_DEVICES: a list of device ordinals, one engine each (the same device may be named more than once) */
		int devs[64], nd = 0;
		if((env = getenv("HVK_DEVICES")))
		{
			const char *p = env;
			while(*p && nd < 64)
			{
				char *end;
				long v = strtol(p, &end, 10);
				if(end == p) break;
				devs[nd++] = (int) v;
				p = *end == ',' ? end + 1 : end;
			}
		}
		if(nd == 1) device = devs[0];
		if(nd > 1)
		{
			if((conf->interlace && conf->interlaced) || conf->raw_bb_file || conf->passthru)
			{
				free(m);
				return(_refuse("--interlace, --raw-bb-file and --passthru on more than one device (HVK_DEVICES)"));
			}
			r = hvk_group_open(&m->g, &hc, sample_rate, pixel_rate, devs, nd, m->batch);
			if(r == HVK_OK) m->e = hvk_group_engine(m->g, 0);
		}
		else r = hvk_open_rates(&m->e, &hc, sample_rate, pixel_rate, device, m->batch);
	}
	if(r != HVK_OK)
	{
		free(m);
		if(r == HVK_UNSUPPORTED) return(_refuse("this configuration"));
		if(r == HVK_NO_DEVICE) fprintf(stderr, "hacktv-amd: no MI355X / HIP device; there is no CPU path\n");
		return(r == HVK_OUT_OF_MEMORY ? VID_OUT_OF_MEMORY : VID_ERROR);
	}

	m->info.struct_size = (uint32_t) sizeof(m->info);
	hvk_get_info(m->e, &m->info);
	if(m->info.lines > (int) sizeof(m->held) || hvk_vbi_lines_held(m->e, m->held, (int) sizeof(m->held)) != HVK_OK)
	{
		_engine_close(m);
		free(m);
		return(VID_ERROR);
	}
	m->out_pos = m->info.startup_samples;
	m->ticks = m->out_pos / m->info.sample_rate * SHIM_AUDIO_RATE + m->out_pos % m->info.sample_rate * SHIM_AUDIO_RATE / m->info.sample_rate;
	m->tick_rem = m->out_pos % m->info.sample_rate * SHIM_AUDIO_RATE % m->info.sample_rate;
	m->volume = conf->volume;

	/* the read-back buffers: page-locked, so that the copy runs at PCIe speed and beside the next batch's host work
	 * (HVK_SHIM_PAGEABLE=1: plain malloc, for comparison) */
	m->stats = getenv("HVK_SHIM_STATS") != NULL;
	if(!getenv("HVK_SHIM_PAGEABLE"))
	{
		m->buf[0] = hvk_host_alloc(m->e, sizeof(int16_t) * 2 * (size_t) m->info.frame_samples * m->batch);
		m->buf[1] = hvk_host_alloc(m->e, sizeof(int16_t) * 2 * (size_t) m->info.frame_samples * m->batch);
		m->pinned = m->buf[0] && m->buf[1];
		if(!m->pinned)
		{
			hvk_host_free(m->e, m->buf[0]);
			hvk_host_free(m->e, m->buf[1]);
		}
	}
	if(!m->pinned)
	{
		m->buf[0] = malloc(sizeof(int16_t) * 2 * (size_t) m->info.frame_samples * m->batch);
		m->buf[1] = malloc(sizeof(int16_t) * 2 * (size_t) m->info.frame_samples * m->batch);
	}
	m->widths = malloc(sizeof(int32_t) * m->info.lines);
	m->cur = -1;
	pthread_mutex_init(&m->lock, NULL);
	pthread_cond_init(&m->cond, NULL);
	if(!m->buf[0] || !m->buf[1] || !m->widths)
	{
		if(m->pinned) { hvk_host_free(m->e, m->buf[0]); hvk_host_free(m->e, m->buf[1]); }
		else { free(m->buf[0]); free(m->buf[1]); }
		free(m->widths);
		_engine_close(m);
		free(m);
		return(VID_OUT_OF_MEMORY);
	}

	/* what main() and vid_info() read back (src/hacktv.c:1452-1518, src/video.c:4846-4860) */
	if(s->conf.hline <= 0 && s->conf.interlaced != 0) s->conf.hline = (s->conf.lines + 1) / 2;
	s->sample_rate = sample_rate;
	s->pixel_rate = m->info.pixel_rate;
	s->width = m->info.width;
	s->half_width = m->info.half_width;
	s->max_width = m->info.max_width;
	s->active_width = m->info.active_width;
	s->active_left = m->info.active_left;
	s->white_level = m->info.white_level;
	s->black_level = m->info.black_level;
	s->blanking_level = m->info.blanking_level;
	s->sync_level = m->info.sync_level;
	s->bframe = 1;
	s->bline = 1;
	if(_shim_register(s, m) != 0)
	{
		fprintf(stderr, "hacktv-amd: more than %d engines in one process\n", SHIM_INSTANCES);
		if(m->pinned) { hvk_host_free(m->e, m->buf[0]); hvk_host_free(m->e, m->buf[1]); }
		else { free(m->buf[0]); free(m->buf[1]); }
		free(m->widths);
		_engine_close(m);
		free(m);
		return(VID_ERROR);
	}

	if(s->conf.raw_bb_file)
	{
		/* src/video.c:4180-4188 */
		s->raw_bb_file = fopen(s->conf.raw_bb_file, "rb");
		if(!s->raw_bb_file)
		{
			perror("fopen");
			vid_free(s);
			return(VID_ERROR);
		}
	}

	if(s->conf.passthru)
	{
		/* src/video.c:4609-4622 */
		s->passthru = strcmp(s->conf.passthru, "-") == 0 ? stdin : fopen(s->conf.passthru, "rb");
		if(!s->passthru)
		{
			perror(s->conf.passthru);
			vid_free(s);
			return(VID_ERROR);
		}
	}

	if(s->conf.teletext)
	{
		/* tt_init() reads width, pixel_rate and the levels from vid_t (src/teletext.c:1057-1074) */
		if((r = tt_init(&s->tt, s, s->conf.teletext)) != VID_OK)
		{
			vid_free(s);
			return(r);
		}
	}

	m->end_drop = hvk_shim_pipeline_depth(s, m->info.delay_lines);

	return(VID_OK);
}

void vid_free(vid_t *s)
{
	shim_t *m = _shim(s);

	if(s->conf.teletext && s->tt.vid) tt_free(&s->tt);
	if(s->passthru && s->passthru != stdin) fclose(s->passthru);   /* src/video.c:4783-4786 */
	if(s->raw_bb_file) fclose(s->raw_bb_file);

	/* the worker is the only caller of the engine and the source: stop it first */
	if(m) { _unhook_source(s); _worker_stop(s); }

	av_close(&s->av);       /* src/video.c:4711 */

	if(m)
	{
		if(m->stats)
		{
			const double el = _now() - m->t_first;
			fprintf(stderr, "hacktv-amd: %lld frames in %.3f s from the first line on = %.1f Msamples/s\n", (long long) m->frames_done, el,
				el > 0 ? m->frames_done * (double) m->info.frame_samples / el * 1e-6 : 0.0);
			fprintf(stderr, "hacktv-amd: worker: source pulls + picture uploads %.3f s, audio pulls %.3f s, stage + launch %.3f s, read-back queueing %.3f s, waiting for a free buffer %.3f s; consumer: waiting for a batch %.3f s (%lld frames)\n",
				m->t_pull, m->t_audio, m->t_render, m->t_fetch, m->t_worker_idle, m->t_consumer_wait, (long long) m->frames_done);
		}
		if(m->g) { for(int i = 0; i < hvk_group_size(m->g); i++) hvk_sync(hvk_group_engine(m->g, i)); }
		else hvk_sync(m->e);    /* a read-back the consumer never waited for */
		if(m->pinned) { hvk_host_free(m->e, m->buf[0]); hvk_host_free(m->e, m->buf[1]); }
		else { free(m->buf[0]); free(m->buf[1]); }
		_engine_close(m);
		pthread_mutex_destroy(&m->lock);
		pthread_cond_destroy(&m->cond);
		free(m->widths);
		free(m->passbuf);
		free(m->aud);
		free(m);
	}
	_shim_unregister(s);

	memset(s, 0, sizeof(vid_t));
}

void vid_info(vid_t *s)
{
	/* same three lines as src/video.c:4846-4860 */
	fprintf(stderr, "Video: %dx%d %.2f fps (full frame %dx%d)\n",
		s->active_width, s->conf.active_lines,
		(double) s->conf.frame_rate.num / s->conf.frame_rate.den,
		s->width, s->conf.lines);
	fprintf(stderr, "Sample rate: %d\n", s->sample_rate);
	fprintf(stderr, "Engine: %s, %d frame(s) per launch\n", hvk_version(), _shim(s) ? _shim(s)->batch : 0);
}

/* The engine behind a vid_t, for an embedder that needs an engine-level call the video.h interface has no
 * place for -- hvk_set_chroma_ghost() with what ITS heap holds behind the reference's chrominance buffer
 * (SURVEY.md H2), before the first vid_next_line(). NULL: not one of the shim's. */
hvk_engine_t *hvk_shim_engine(vid_t *s)
{
	shim_t *m = _shim(s);
	return(m ? m->e : NULL);
}

size_t vid_get_framebuffer_length(vid_t *s)
{
	return(sizeof(uint32_t) * s->active_width * s->conf.active_lines);
}

/* Pull up to `batch` frames and the audio they need from the source, render them into iq */
static int _next_batch(vid_t *s, shim_t *m, int16_t *iq, int *ticket, hvk_engine_t **eng)
{
	double t0 = m->stats ? _now() : 0, t1;
	hvk_engine_t *const E = m->g ? hvk_group_block_engine(m->g) : m->e;     /* the engine this batch goes to */
	int32_t slots[512];
	const int fields = (s->conf.interlace && s->conf.interlaced) ? 2 : 1;
	int n = 0;

	m->frame_in_batch_pull = 0;

	while(n < m->batch && n < 256)
	{
		av_frame_t *f = &s->vframe;
		const int slot = n * fields;
		uint8_t cc[2];
		r64_t par;

		/* src/video.c:4873-4897: end of source is tested at the start of each frame -- and, with
		 * --interlace, of each field, which shows a frame of its own. (A source that ends between the
		 * two fields ends the stream with the frame before; the reference would still emit that
		 * frame's first field.) */
		if(av_eof(&s->av)) { m->ended = 1; break; }

		/* like src/video.c:4881: into the vid_t's own frame, so that a source whose video ends before
		 * its sound shows its last picture once more (the failed read leaves the frame as it was) and
		 * blank frames from then on (src/av.c:55-59) */
		av_read_video(&s->av, f);
		/* src/video.c:4883-4885: the systems that scan vertically turn every picture (pointer and stride arithmetic, src/av.c:242-290;
		 * the upload gathers along whatever strides it is given) */
		if(s->conf.frame_orientation)
		{
			av_rotate_frame(f, s->conf.frame_orientation & 3);
			if(s->conf.frame_orientation & VID_HFLIP) av_hflip_frame(f);
			if(s->conf.frame_orientation & VID_VFLIP) av_vflip_frame(f);
		}

		if((m->g ? hvk_group_frame_upload(m->g, slot, f->framebuffer, f->width, f->height, f->pixel_stride, f->line_stride, f->interlaced)
		        : hvk_frame_upload(m->e, slot, f->framebuffer, f->width, f->height, f->pixel_stride, f->line_stride, f->interlaced)) != HVK_OK) return(-1);
		slots[slot] = slot;
		par = f->pixel_aspect_ratio;

		/* src/video.c:4899-4903 queues the pair of every picture read; line 21 takes one per frame
		 * (src/cc608.c:203), before the second field's picture is read */
		if(s->conf.cc608)
		{
			_cc_push(m, f->cc608);
			if(_cc_pop(m, cc) && hvk_cc608_write(E, n, cc[0], cc[1]) != HVK_OK) return(-1);
		}

		if(fields == 2)
		{
			if(av_eof(&s->av)) { m->ended = 1; break; }
			av_read_video(&s->av, f);
			if(s->conf.frame_orientation)
			{
				av_rotate_frame(f, s->conf.frame_orientation & 3);
				if(s->conf.frame_orientation & VID_HFLIP) av_hflip_frame(f);
				if(s->conf.frame_orientation & VID_VFLIP) av_vflip_frame(f);
			}
			if(hvk_frame_upload(m->e, slot + 1, f->framebuffer, f->width, f->height, f->pixel_stride, f->line_stride, f->interlaced) != HVK_OK) return(-1);
			slots[slot + 1] = slot + 1;
			if(s->conf.cc608) _cc_push(m, f->cc608);
		}

		if(s->conf.wss && hvk_frame_aspect(E, slot, par.num, par.den) != HVK_OK) return(-1);

		if(s->conf.teletext)
		{
			/* the packets of this frame, asked for in the order the lines go out */
			uint8_t rows[32][45];
			uint32_t mask = 0;
			int frame = (int) (m->frames_pulled + 1), row;

			for(row = 0; row < 32; row++)
			{
				int line = row < 16 ? 7 + row : 320 + row - 16;
				/* a line another inserter holds is left alone and the packet kept for the next one (vbialloc,
				 * src/teletext.c:1219): the engine says which (hvk_vbi_lines_held(): from the tables it
				 * renders those inserters with) */
				if(m->held[line - 1]) continue;
				if(tt_next_packet(&s->tt, rows[row], frame, line) == TT_OK) mask |= 1u << row;
			}
			if(hvk_teletext_packets(E, n, &rows[0][0], mask) != HVK_OK) return(-1);
		}

		m->frame_in_batch_pull++;
		m->frames_pulled++;
		n++;

		if(m->stats) { t1 = _now(); m->t_pull += t1 - t0; t0 = t1; }

		/* 32 kHz audio for this frame (src/video.c:3278-3286), pulled frame by frame so that the end
		 * of the sound is seen by the same av_eof() as in the reference; a source that runs dry
		 * leaves silence (src/video.c:3299-3304) */
		if(m->info.has_carriers || m->info.has_nicam || s->conf.sis)      /* (sound-in-syncs takes the sound of a mode without a sound carrier too) */
		{
			while((m->g ? hvk_group_audio_needed(m->g, n) : hvk_audio_needed(m->e, n)) > 0)
			{
				int16_t *a = NULL;
				size_t an = 0;
				av_read_audio(&s->av, &a, &an);
				if(a == NULL || an == 0) break;
				if((m->g ? hvk_group_audio_write(m->g, a, an) : hvk_audio_write(m->e, a, an)) != HVK_OK) return(-1);
				if(_aud_append(m, a, an) != 0) return(-1);
			}
		}
		else
		{
			/* no sound carrier: the reference's audio process still draws the source at 32 kHz
			 * (src/video.c:3272-3286), and a source only ends once its sound has */
			const int64_t pos = hvk_frame_start(m->e, m->frames_pulled) + m->info.startup_samples;
			const int64_t ticks = pos / m->info.sample_rate * SHIM_AUDIO_RATE
			                    + pos % m->info.sample_rate * SHIM_AUDIO_RATE / m->info.sample_rate;
			while(m->audio_drawn < ticks)
			{
				int16_t *a = NULL;
				size_t an = 0;
				av_read_audio(&s->av, &a, &an);
				if(a == NULL || an == 0) break;
				m->audio_drawn += an;
				if(_aud_append(m, a, an) != 0) return(-1);
			}
		}
		if(m->stats) { t1 = _now(); m->t_audio += t1 - t0; t0 = t1; }
	}

	/* a full batch that used up the source is the last one too */
	if(!m->ended && av_eof(&s->av)) m->ended = 1;

	if(n == 0) return(0);

	/* --raw-bb-file: the lines of these frames and the one after them (the filter looks into it), read
	 * like src/video.c:2419-2429 -- at the end of the file, start over */
	if(s->raw_bb_file)
	{
		/* (behind the resampler the engine looks one raster line further: its chunks lag the raster by a slot) */
		const int64_t want = ((m->frames_pulled * (int64_t) m->info.lines) + 1 + (m->info.pixel_rate != m->info.sample_rate ? 1 : 0)) * m->info.width;
		int16_t chunk[4096];
		int empty = 0;

		while(m->raw_fed < want && empty < 2)
		{
			size_t ask = (size_t) (want - m->raw_fed) < 4096 ? (size_t) (want - m->raw_fed) : 4096;
			size_t r = fread(chunk, sizeof(int16_t), ask, s->raw_bb_file);
			if(r == 0)
			{
				if(!feof(s->raw_bb_file)) break;
				rewind(s->raw_bb_file);
				empty++;            /* an empty file would loop for ever */
				continue;
			}
			empty = 0;
			if(hvk_rawbb_write(m->e, chunk, r) != HVK_OK) return(-1);
			m->raw_fed += r;
		}
	}

	/* --passthru: the lines of these frames (and, once, of the filter's start-up lines) from the
	 * external signal (src/video.c:3517-3541); a short source simply ends */
	if(s->passthru)
	{
		/* (the samples of these n frames: n * frame_samples but for rate pairs with frames of two lengths) */
		size_t want = (size_t) (hvk_frame_start(m->e, m->frames_pulled) - hvk_frame_start(m->e, m->frames_pulled - n)), got;

		if(!m->passthru_primed)
		{
			want += (size_t) m->info.startup_samples;      /* (delay_lines * width; with --pixelrate the dropped chunks' widths) */
			m->passthru_primed = 1;
		}

		if(!m->passbuf) m->passbuf = malloc(sizeof(int16_t) * 2 * ((size_t) m->batch * m->info.frame_samples + (size_t) m->info.startup_samples));
		if(!m->passbuf) return(-1);

		got = 0;
		while(got < want && !feof(s->passthru))
		{
			size_t r = fread(m->passbuf + got * 2, sizeof(int16_t) * 2, want - got, s->passthru);
			if(r == 0) break;
			got += r;
		}
		if(hvk_passthru_write(m->e, m->passbuf, got) != HVK_OK) return(-1);
	}

	if(m->stats) t0 = _now();
	if(m->g)
	{
		if(hvk_group_stage(m->g, n, slots) != HVK_OK || hvk_group_launch(m->g, NULL) < 0) return(-1);
	}
	else if(hvk_render(m->e, n, slots, NULL) != HVK_OK) return(-1);
	if(m->stats) { t1 = _now(); m->t_render += t1 - t0; t0 = t1; }
	/* the read-back is queued behind the render; the consumer waits for it (vid_next_line) while this thread goes
	 * on with the next batch's pulls and host pre-passes */
	*eng = E;
	*ticket = hvk_fetch_async(E, iq, 0, (size_t) (hvk_frame_start(m->e, m->frames_pulled) - hvk_frame_start(m->e, m->frames_pulled - n)));
	if(*ticket < 0) return(-1);
	if(m->stats) { t1 = _now(); m->t_fetch += t1 - t0; }

	return(n);
}

/* The worker: keeps the buffer that is not being handed out filled. It alone talks to the source
 * (av_read_video / av_read_audio / tt_next_packet / the passthru file) and to the engine. */
static void *_worker(void *arg)
{
	vid_t *s = arg;
	shim_t *m = _shim(s);
	int b = 0;

	for(;;)
	{
		int n;

		int ticket = -1;
		double t0 = m->stats ? _now() : 0;

		pthread_mutex_lock(&m->lock);
		while(m->ready[b] && !m->stop) pthread_cond_wait(&m->cond, &m->lock);
		if(m->stop) { pthread_mutex_unlock(&m->lock); break; }
		pthread_mutex_unlock(&m->lock);
		if(m->stats) m->t_worker_idle += _now() - t0;

		hvk_engine_t *eng = m->e;
		n = m->ended ? 0 : _next_batch(s, m, m->buf[b], &ticket, &eng);

		pthread_mutex_lock(&m->lock);
		m->fe[b] = eng;
		m->last[b] = m->ended;
		m->count[b] = n;
		m->ticket[b] = ticket;
		m->ready[b] = 1;
		pthread_cond_broadcast(&m->cond);
		pthread_mutex_unlock(&m->lock);

		if(n <= 0) break;       /* end of the source (or a failure): nothing more to render */
		b ^= 1;
	}

	return(NULL);
}

vid_line_t *vid_next_line(vid_t *s)
{
	shim_t *m = _shim(s);
	vid_line_t *l;

	if(!m) return(NULL);

	if(m->source_closed)
	{
		if(m->source_closed == 1) fprintf(stderr, "hacktv-amd: one source per run (the next one would start in the middle of the reference's line pipeline); stopping here\n");
		m->source_closed = 2;
		return(NULL);
	}

	if(m->frame_in_batch >= m->have)
	{
		int nb = m->cur < 0 ? 0 : m->cur ^ 1, n;

		if(!m->worker_on)
		{
			/* started on the first line, once main() has filled in s->av (src/hacktv.c:1493) */
			if(_hook_source(s) != 0) return(NULL);
			if(pthread_create(&m->worker, NULL, _worker, s) != 0) return(NULL);
			m->worker_on = 1;
			m->t_first = _now();
			if(m->stats) { _dbg_worker = m->worker; _dbg_have_worker = 1; signal(SIGUSR1, _dbg_usr1); }
		}

		/* hand the finished buffer back, wait for the next one and for its read-back */
		double t0 = m->stats ? _now() : 0;
		int ticket;
		pthread_mutex_lock(&m->lock);
		if(m->cur >= 0) { m->ready[m->cur] = 0; pthread_cond_broadcast(&m->cond); }
		while(!m->ready[nb]) pthread_cond_wait(&m->cond, &m->lock);
		n = m->count[nb];
		ticket = m->ticket[nb];
		pthread_mutex_unlock(&m->lock);

		if(n <= 0) { m->have = 0; m->frame_in_batch = 0; return(NULL); }
		if(hvk_fetch_wait(m->fe[nb] ? m->fe[nb] : m->e, ticket) != HVK_OK) { m->have = 0; m->frame_in_batch = 0; return(NULL); }
		if(m->stats) m->t_consumer_wait += _now() - t0;
		m->cur = nb;
		m->iq = m->buf[nb];
		m->have = n;
		m->frame_in_batch = 0;
		m->line = 0;
		m->line_at = 0;
	}

	/* The reference tests for the end of the source when it starts a frame (src/video.c:4876), and
	 * returns NULL there and then: the lines still in its pipeline -- one per two-slot process, the
	 * resampler and the video filter -- never come out. Nor do they here. */
	if(m->last[m->cur] && m->frame_in_batch == m->have - 1 && m->line >= m->info.lines - m->end_drop)
	{
		m->frame_in_batch = m->have;    /* a further call finds the worker's empty batch */
		return(NULL);
	}

	if(m->line == 0)
	{
		/* the widths of this frame's lines: constant without the resampler (src/video.c:3246) */
		if(hvk_line_widths(m->e, m->frames_done * m->info.lines, m->info.lines, m->widths) != HVK_OK) return(NULL);
	}

	l = &m->out;
	l->output = m->iq + 2 * m->line_at;
	l->width = m->widths[m->line];
	m->line_at += l->width;
	l->frame = (int) (m->frames_done + 1);
	l->line = m->line + 1;
	l->lut = NULL;
	l->vbialloc = 0;
	l->audio = NULL;
	l->audio_len = 0;
	{
		/* 32 kHz samples the audio process has drawn once it is through with this line (one every sample_rate / 32000
		 * output samples, src/video.c:3272-3274; it is startup_samples ahead of the output); a block goes out on the
		 * first line that finds it complete, one block per line at most */
		int64_t ticks, avail;
		/* (floor(out_pos * 32000 / sample_rate), carried along from line to line: no division on the way) */
		m->out_pos += l->width;
		m->tick_rem += (int64_t) l->width * SHIM_AUDIO_RATE;
		while(m->tick_rem >= m->info.sample_rate) { m->tick_rem -= m->info.sample_rate; m->ticks++; }
		ticks = m->ticks;
		avail = ticks >= 1 ? (ticks - 1) / SHIM_AUDIO_BLOCK : 0;
		if(m->aud_blocks_read < avail)
		{
			const int64_t first = m->aud_blocks_read * SHIM_AUDIO_BLOCK;
			int i;
			pthread_mutex_lock(&m->lock);
			for(i = 0; i < SHIM_AUDIO_BLOCK; i++)
			{
				/* a source that has run dry leaves silence (src/video.c:3299-3304) */
				const int64_t at = first + i - m->aud_base;
				const int have = at >= 0 && at < (int64_t) m->aud_len;
				m->aud_out[i * 2 + 0] = have ? m->aud[at * 2 + 0] : 0;
				m->aud_out[i * 2 + 1] = have ? m->aud[at * 2 + 1] : 0;
			}
			/* what has gone out is dropped now and then */
			if(first + SHIM_AUDIO_BLOCK - m->aud_base >= 65536 && first + SHIM_AUDIO_BLOCK - m->aud_base <= (int64_t) m->aud_len)
			{
				const size_t drop = (size_t) (first + SHIM_AUDIO_BLOCK - m->aud_base);
				memmove(m->aud, m->aud + drop * 2, (m->aud_len - drop) * 2 * sizeof(int16_t));
				m->aud_len -= drop;
				m->aud_base += drop;
			}
			pthread_mutex_unlock(&m->lock);
			m->aud_blocks_read++;
			l->audio = m->aud_out;
			l->audio_len = SHIM_AUDIO_BLOCK * 2;
		}
	}
	l->previous = l->next = l;

	s->frame = l->frame;
	s->line = l->line;

	if(++m->line == m->info.lines)
	{
		m->line = 0;
		m->frame_in_batch++;
		m->frames_done++;
	}

	return(l);
}
