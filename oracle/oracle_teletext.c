/* oracle/oracle_teletext.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of the RENDER step of hacktv's teletext inserter:
 * tt_render_line (src/teletext.c:1211-1236) with the symbol table tt_init()
 * builds (src/teletext.c:1057-1074 -> vbidata_init, src/vbidata.c:83-141) and
 * vbidata_render (src/vbidata.c:186-239).
 *
 * Which 45-byte packet goes on which line is host control logic (page store,
 * magazine scheduler, wall clock: src/teletext.c:489-990) and stays with the
 * caller: the packets are an INPUT here, exactly as they are an input of the
 * device path.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

static double _sinc(double x) { return(sin(M_PI * x) / (M_PI * x)); }

/* src/vbidata.c:31-35 */
static double _raised_cosine(double x, double b, double t)
{
	if(x == 0) return(1.0);
	return(_sinc(x / t) * (cos(M_PI * b * x / t) / (1.0 - (4.0 * b * b * x * x / (t * t)))));
}

int orc_teletext_init(orc_t *s)
{
	/* 66 % of white - black; 360 symbols at 444 x line rate, raised cosine
	 * beta 0.7, first symbol 12 us less 12 bit periods after 0H */
	int level = round((s->white_level - s->black_level) * 0.66);
	double bwidth = (double) s->width / 444;
	double offset = s->pixel_rate * (12e-6 - (64e-6 / 444 * 12));
	int b, x;

	s->tt_sym = calloc(360, sizeof(orc_pulse_t));

	for(b = 0; b < 360; b++)
	{
		double t = -bwidth * b - offset;
		orc_pulse_t *p = &s->tt_sym[b];

		p->value = calloc(s->width, sizeof(int16_t));
		p->offset = p->length = 0;

		for(x = 0; x < s->width; x++)
		{
			int v = round(_raised_cosine((t + x) / bwidth, 0.7, 1) * level);
			if(v == 0) continue;
			if(p->length == 0) p->offset = x;
			p->value[x - p->offset] = v;
			p->length = x - p->offset + 1;
		}
	}

	return(0);
}

void orc_teletext_free(orc_t *s)
{
	int b;
	if(!s->tt_sym) return;
	for(b = 0; b < 360; b++) free(s->tt_sym[b].value);
	free(s->tt_sym);
	s->tt_sym = NULL;
}

/* Add one packet, least significant bit of each byte first, to a line */
void orc_teletext_render(orc_t *s, int16_t *o, const uint8_t packet[45])
{
	int b, x;
	for(b = 0; b < 360; b++)
	{
		const orc_pulse_t *p = &s->tt_sym[b];
		if(!((packet[b >> 3] >> (b & 7)) & 1)) continue;
		for(x = 0; x < p->length && p->offset + x < s->width; x++) o[p->offset + x] += p->value[x];
	}
}
