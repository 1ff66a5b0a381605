/* oracle/pin_time.c -- TEST INFRASTRUCTURE (not product code).
 *
 * LD_PRELOAD shim that pins the wall clock: the reference's teletext service puts the time of day
 * into its packets (src/teletext.c:605 calls time(NULL) for every packet and sends an 8/30 packet
 * whenever the second changes; the page header's clock comes from the same value, :443-445), so two
 * runs of `--teletext demo.tti` never agree. With the clock pinned -- for the
 * unmodified reference and for the drop-in binary alike, whose teletext scheduler is the reference's
 * own unchanged code -- BASELINE config 4 is reproducible. HVK_PIN_TIME overrides the instant;
 * run with TZ=UTC so that localtime() does not depend on the box either. */
#include <stdlib.h>
#include <time.h>

time_t time(time_t *t)
{
	const char *e = getenv("HVK_PIN_TIME");
	time_t v = e ? (time_t) atoll(e) : (time_t) 1700000000;
	if(t) *t = v;
	return(v);
}
