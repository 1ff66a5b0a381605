/* oracle/ref_stubs.c -- TEST INFRASTRUCTURE (not product code).
 *
 * The three ffmpeg entry points declared by the reference in
 * src/av_ffmpeg.h:21-23. libav* is not installed in this image, so the
 * reference's av_ffmpeg.c is left out of the oracle build and these stubs
 * stand in for it: opening an ffmpeg source always fails, init/deinit do
 * nothing. The built-in "test" source (src/av_test.c) is unaffected.
 */
#include <stdlib.h>
#include <stdint.h>
#include "hacktv.h"

int av_ffmpeg_open(av_t *av, char *input_url, char *format, char *options)
{
	(void) av; (void) input_url; (void) format; (void) options;
	return(AV_ERROR);
}

void av_ffmpeg_init(void) { }
void av_ffmpeg_deinit(void) { }
