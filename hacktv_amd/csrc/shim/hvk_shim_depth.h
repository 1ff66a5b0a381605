/* hvk_shim_depth.h -- how many lines the reference's line pipeline holds back (part of the video.h shim,
 * hvk_video_shim.c; in a header of its own so that the test probe, oracle/ref_probe.c, can put the same
 * function next to the reference's own vid_t and compare: tests/test_oracle_vs_ref.py). Needs the
 * reference's video.h. */
#ifndef HVK_SHIM_DEPTH_H
#define HVK_SHIM_DEPTH_H

/* The lines in flight in the reference's line pipeline: vid_init() gives each process a window of
 * `nlines` output-line buffers in a ring (src/video.c:3578, :4675-4688); two neighbours share one
 * buffer unless either runs on a thread of its own. The raster writes the middle one of its three
 * (src/video.c:2873), "output" is the last window, and a buffer takes as many calls of
 * _vid_next_line() from the one to the other as their positions in the ring differ. That many lines
 * are handed out late -- and, at the end of the source, not at all (see vid_next_line below). */
static inline int hvk_shim_pipeline_depth(const vid_t *s, int filter_delay)
{
	int depth = s->raw_bb_file ? 0 : 1;     /* rawbb has one buffer, the raster one ahead of the line it writes */
	int prev_thread = 0;

#define PROCESS(nlines, thread) do { depth += (nlines) - ((thread) || prev_thread ? 0 : 1); prev_thread = (thread); } while(0)
	if(!s->raw_bb_file && s->conf.colour_mode == VID_SECAM) PROCESS(1, 1);
	if(s->conf.vits) PROCESS(1, 0);
	if(s->conf.wss) PROCESS(1, 0);
	if(s->conf.acp) PROCESS(1, 0);
	if(s->conf.vitc) PROCESS(1, 0);
	if(s->conf.cc608) PROCESS(1, 0);
	if(s->conf.teletext) PROCESS(1, 0);
	if(s->pixel_rate != s->sample_rate) PROCESS(2, 1);             /* src/video.c:3648 */
	if(s->conf.vfilter) PROCESS(1 + filter_delay, 1);              /* src/video.c:3761 */
	PROCESS(1, 1);                                                 /* audio, always: src/video.c:4561 */
	if(s->conf.modulation == VID_FM) PROCESS(1, 1);
	if(s->conf.swap_iq) PROCESS(1, 0);
	if(s->conf.offset) PROCESS(1, 1);
	if(s->conf.passthru) PROCESS(1, 0);
	PROCESS(1, 0);                                                 /* output */
#undef PROCESS

	return(depth);
}

#endif
