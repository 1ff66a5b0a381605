/* oracle/shim_check.c -- TEST INFRASTRUCTURE (not product code).
 *
 * The video.h shim (hacktv_amd/csrc/shim/hvk_video_shim.c + libhvk, the GPU engine) against the
 * reference's own engine IN THE SAME PROCESS, line by line: both are driven through the reference's
 * vid_init / vid_next_line interface from two identical instances of a source of our own -- random
 * pictures that change every frame, caption pairs, noise as audio, and an END (video after a given
 * number of frames, audio a little later), which the built-in test source never reaches. Compared per
 * line: width, frame and line numbers, every sample; and both must return NULL on the same call.
 *
 * Built by `make -C oracle shimcheck` from the reference objects where they lie: video.c is compiled
 * with its engine entry points renamed to ref_vid_* (as for the drop-in binary), so both engines link
 * into one program. Usage: shim_check <mode> <sample rate> <frames> <flags> [pixel rate]
 *   flags: 1 --filter, 2 --noaudio, 4 --vits, 8 --vitc, 16 --acp, 32 --cc608, 64 --interlace, 128 --a2stereo
 *
 * The reference's chroma filter over-reads its heap: the shim's engine is given the bytes
 * that follow the reference's chrominance buffer in THIS process (hvk_set_chroma_ghost(); the shim's default
 * models the CLI's heap) -- with FM video a difference there would stay in the modulator's phase for good.
 * Should those bytes change while the run lasts, only the last samples of colour lines and the filter's
 * reach around them could differ: they stay out of the comparison on PAL / NTSC.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hacktv.h"

extern int ref_vid_init(vid_t *s, unsigned int sample_rate, unsigned int pixel_rate, const vid_config_t * const conf);
extern vid_line_t *ref_vid_next_line(vid_t *s);
/* the shim's engine (hvk_video_shim.c) and the one engine-level call made here (include/hacktv_amd.h) */
extern void *hvk_shim_engine(vid_t *s);
extern int hvk_set_chroma_ghost(void *e, const int16_t *ghost, int n);

typedef struct {
	uint32_t *fb;
	int width, height, frames_left, nframe;
	uint32_t seed;
	int16_t audio[2048 * 2];
	int audio_chunks_left;
	uint32_t aseed;
} src_t;

static uint32_t _lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return(*s >> 8); }

static int _read_video(void *ctx, av_frame_t *frame)
{
	src_t *c = ctx;
	int i;
	if(c->frames_left <= 0) return(AV_EOF);
	c->frames_left--;
	for(i = 0; i < c->width * c->height; i++) c->fb[i] = _lcg(&c->seed) & 0xFFFFFF;
	av_frame_init(frame, c->width, c->height, c->fb, 1, c->width);
	frame->interlaced = (c->nframe & 1) ? 1 : 0;    /* alternately field ordered and progressive */
	frame->cc608[0] = 0x40 + (c->nframe % 20);
	frame->cc608[1] = (c->nframe % 3) ? 0x61 : 0x00;
	frame->pixel_aspect_ratio = (r64_t) { 12, 13 };
	c->nframe++;
	return(AV_OK);
}

static int _read_audio(void *ctx, int16_t **samples, size_t *nsamples)
{
	src_t *c = ctx;
	int i;
	if(c->audio_chunks_left <= 0) return(AV_EOF);
	c->audio_chunks_left--;
	for(i = 0; i < 2048 * 2; i++) c->audio[i] = (int16_t) (_lcg(&c->aseed) & 0xFFFF);
	*samples = c->audio;
	*nsamples = 2048;
	return(AV_OK);
}

static int _close(void *ctx) { return(AV_OK); }

static void _source(vid_t *v, src_t *c, int frames, int audio_chunks)
{
	memset(c, 0, sizeof(*c));
	c->width = v->active_width;
	c->height = v->conf.active_lines;
	c->fb = malloc(sizeof(uint32_t) * c->width * c->height);
	c->frames_left = frames;
	c->seed = 12345;
	c->aseed = 777;
	c->audio_chunks_left = audio_chunks;
	v->av = (av_t) {
		.frame_rate = (r64_t) { v->conf.frame_rate.num * (v->conf.interlace ? 2 : 1), v->conf.frame_rate.den },
		.width = v->active_width, .height = v->conf.active_lines,
		.sample_rate = (r64_t) { HACKTV_AUDIO_SAMPLE_RATE, 1 },
		.av_source_ctx = c, .read_video = _read_video, .read_audio = _read_audio, .close = _close,
	};
}

int main(int argc, char *argv[])
{
	const vid_configs_t *vc;
	vid_config_t conf;
	static vid_t a, b;
	static src_t sa, sb;
	unsigned int sr, pr = 0;
	int frames, flags, colour_tail;
	long lines = 0, bad_lines = 0, compared = 0, audio_blocks = 0;

	if(argc < 5) { fprintf(stderr, "usage: shim_check <mode> <sample rate> <frames> <flags> [pixel rate]\n"); return(2); }
	sr = atoi(argv[2]);
	frames = atoi(argv[3]);
	flags = atoi(argv[4]);
	if(argc > 5) pr = atoi(argv[5]);

	for(vc = vid_configs; vc->id != NULL && strcmp(argv[1], vc->id) != 0; vc++);
	if(vc->id == NULL) { fprintf(stderr, "no such mode\n"); return(2); }
	memcpy(&conf, vc->conf, sizeof(conf));
	conf.volume = 256;
	if(flags & 1) conf.vfilter = 1;
	if(flags & 2)
	{
		conf.fm_mono_level = conf.am_audio_level = conf.nicam_level = 0;
		conf.fm_mono_carrier = conf.nicam_carrier = conf.am_mono_carrier = 0;
	}
	if(flags & 4) conf.vits = 1;
	if(flags & 8) conf.vitc = 1;
	if(flags & 16) conf.acp = 1;
	if(flags & 32) conf.cc608 = 1;
	if(flags & 64) conf.interlace = 1;
	if(flags & 128) conf.a2stereo = 1;

	if(ref_vid_init(&a, sr, pr, &conf) != VID_OK) { printf("REFUSED by the reference\n"); return(1); }
	if(vid_init(&b, sr, pr, &conf) != VID_OK) { printf("REFUSED by the shim\n"); return(1); }

	/* video for `frames` frames (twice as many pictures with --interlace), audio for a frame longer */
	{
		const int pictures = frames * (conf.interlace ? 2 : 1);
		const int chunks = (int) ((double) (frames + 1) * conf.frame_rate.den / conf.frame_rate.num * 32000 / 2048) + 1;
		_source(&a, &sa, pictures, chunks);
		_source(&b, &sb, pictures, chunks);
	}

	colour_tail = (conf.colour_mode == VID_PAL || conf.colour_mode == VID_NTSC);
	if(colour_tail && a.chrominance_buffer)
	{
		if(hvk_set_chroma_ghost(hvk_shim_engine(&b), a.chrominance_buffer + 2 * a.width, 32) != 0) { printf("hvk_set_chroma_ghost failed\n"); return(1); }
	}

	for(;;)
	{
		vid_line_t *la = ref_vid_next_line(&a);
		vid_line_t *lb = vid_next_line(&b);
		int x, bad = 0, x0 = 0, x1;

		if(!la || !lb)
		{
			if(la || lb) { printf("DIFFERENT: %s ended first, after %ld lines\n", la ? "the shim" : "the reference", lines); fflush(stdout); _exit(1); }
			break;
		}

		if(la->width != lb->width || la->frame != lb->frame || la->line != lb->line)
		{
			printf("DIFFERENT: line %ld is (frame %d line %d width %d) in the reference, (%d %d %d) in the shim\n",
			       lines, la->frame, la->line, la->width, lb->frame, lb->line, lb->width);
			fflush(stdout);
			_exit(1);
		}

		/* the line's share of the 32 kHz sound (src/video.c:3445-3447: what rf_write_audio() sinks are handed) */
		if(la->audio_len != lb->audio_len || (la->audio == NULL) != (lb->audio == NULL) ||
		   (la->audio && memcmp(la->audio, lb->audio, la->audio_len * sizeof(int16_t)) != 0))
		{
			printf("DIFFERENT: frame %d line %d: line->audio (%zu samples in the reference, %zu in the shim)\n", la->frame, la->line, la->audio_len, lb->audio_len);
			fflush(stdout);
			_exit(1);
		}
		if(la->audio) audio_blocks++;

		x1 = la->width;
		if(colour_tail) { x0 = 32; x1 = la->width - 40; }   /* H2: see the header */
		for(x = x0; x < x1; x++)
		{
			if(la->output[x * 2] != lb->output[x * 2] || la->output[x * 2 + 1] != lb->output[x * 2 + 1]) { bad++; }
		}
		compared += x1 - x0;

		if(bad && bad_lines < 5)
		{
			for(x = x0; x < x1 && la->output[x * 2] == lb->output[x * 2] && la->output[x * 2 + 1] == lb->output[x * 2 + 1]; x++);
			printf("frame %d line %d: %d samples differ, first x = %d: reference (%d, %d) shim (%d, %d)\n", la->frame, la->line, bad, x,
			       la->output[x * 2], la->output[x * 2 + 1], lb->output[x * 2], lb->output[x * 2 + 1]);
		}
		if(bad) bad_lines++;
		lines++;
	}

	printf("%s: %ld lines, %ld samples compared, %ld lines differ; %ld blocks of line->audio equal; both ended on the same call\n", bad_lines ? "DIFFERENT" : "EQUAL", lines, compared, bad_lines, audio_blocks);
	fflush(stdout);
	_exit(bad_lines ? 1 : 0);       /* the reference's vid_free() can hang on its thread shutdown (see ref_probe.c) */
}
