/* hvk_secam_chain.h -- the SECAM colour sub-carrier of ONE line, written once for the device kernels (hvk_secam.hip)
 * and for the host (hvk_secam.c: the serial chain that pins it, and the fall-back).
 *
 * The reference's _vid_render_secam (src/video.c:3068-3233) per line with a picture or a field identification ramp:
 *
 *   cells    the line's colour difference as a frequency deviation: D'r lines carry v, D'b lines u, each averaged
 *            with the same component of the line before it in the field (src/video.c:3149-3196)
 *   low pass 15 taps, zero history, reading 7 values PAST the line (src/video.c:3207, src/fir.c:365-372) -- the
 *            memory behind the line is where the line before left the last values of its FM loop
 *   IIR      pre-emphasis in double precision, state never reset (src/fir.c:721-735)
 *   FM       limit, bell-filter gain, the floor-after-every-step phasor restarted on every line, burst envelope
 *            (src/video.c:3210-3229, :2278-2297, src/common.h:80-89); the loop runs to burst_left + burst_width,
 *            which can lie past the line's end: those steps work on the values behind the line
 *
 * What one line hands to the next is therefore small: the IIR's two doubles and the (up to 7) values behind the
 * line -- hvk_secam_state_t. Everything else of a line depends on its own pixels and those of the line above.
 * cells + low pass without the tail's share are sample-parallel (hvk_secam_cells_fir); the rest is a serial walk
 * over the line given the state (hvk_secam_chain_line).
 */
#ifndef HVK_SECAM_CHAIN_H
#define HVK_SECAM_CHAIN_H

#include <stdint.h>

#ifdef __HIPCC__
#define HVK_HD __host__ __device__ __forceinline__
#else
#define HVK_HD static inline
#endif

#define HVK_SECAM_TAIL 7        /* values behind the line the low pass can reach */

typedef struct {
	double ix, iy;              /* IIR: last input, last output */
	int16_t tail[8];            /* the values behind the line ([7] unused) */
} hvk_secam_state_t;

typedef struct {
	int32_t W, sl;              /* line width; first sample of the sub-carrier window (burst_left) */
	int32_t level;
	int16_t dmin[2], dmax[2];   /* deviation limits of D'b / D'r lines */
	int16_t fir[16];            /* the 15 low-pass taps, applied order */
} hvk_secam_consts_t;

/* A line of a frame the process works on (a picture line or a field identification line), listed per frame parity
 * in line order (hvk_secam_tasks()) */
#define HVK_SECAM_TASK_VALID 1
#define HVK_SECAM_TASK_FID   2     /* field identification ramp instead of a picture */
#define HVK_SECAM_TASK_CLEAR 4     /* a field has begun since the task before: nothing lies behind the line, no line above */
typedef struct {
	int16_t line;               /* 1-based */
	int16_t prev_line;          /* the picture line before it in the field whose other component it averages with; 0: none */
	int16_t sr;                 /* end of the FM loop */
	int16_t flags;
} hvk_secam_task_t;

typedef struct { int32_t i, q; } hvk_secam_c32_t;
typedef struct { int16_t i, q; } hvk_secam_c16_t;

/* lround() for |x| < 2^31 without the library call: truncate, then look at the (exactly representable) rest;
 * halves go away from zero */
HVK_HD int32_t hvk_secam_round_away(double x)
{
	int32_t i = (int32_t) x;
	double f = x - (double) i;
	if(f >= 0.5) i++;
	else if(f <= -0.5) i--;
	return(i);
}

/* One FM step on `cell` (src/video.c:3214-3227): returns the sub-carrier value that replaces it */
HVK_HD int16_t hvk_secam_fm_step(const hvk_secam_c32_t *lut, const hvk_secam_c16_t *bell, int16_t cell, int16_t dmin, int16_t dmax,
                                 int32_t level, int32_t *ppi, int32_t *ppq)
{
	const int16_t v = cell < dmin ? dmin : (cell > dmax ? dmax : cell);
	const hvk_secam_c16_t g = bell[(uint16_t) v];
	const hvk_secam_c32_t st = lut[(int32_t) v + 32768];
	const int64_t ni = (int64_t) *ppi * st.i - (int64_t) *ppq * st.q;
	const int64_t nq = (int64_t) *ppi * st.q + (int64_t) *ppq * st.i;
	int32_t vi, vq;

	*ppi = (int32_t) (ni >> 31);
	*ppq = (int32_t) (nq >> 31);
	vi = ((*ppi >> 16) * level) >> 15;
	vq = ((*ppq >> 16) * level) >> 15;
	return((int16_t) (((vi * g.i) >> 15) - ((vq * g.q) >> 15)));
}

/* The serial part of one line. F: the line's low-pass outputs [0, W - 7) (stride fs elements); acc: the 7 outputs
 * after them as 32-bit sums without the tail's share (stride as). S: state on entry, on exit. dr: D'r line;
 * sr: end of the FM loop (may exceed W by up to 7); phase_pos: the phasor starts at +1 (every third line,
 * src/video.c:3211-3212). out (NULL: a warm-up pass): receives the W values the line adds to the video, stride os. */
HVK_HD void hvk_secam_chain_line(const hvk_secam_consts_t *C, const hvk_secam_c32_t *lut, const hvk_secam_c16_t *bell, const int16_t *burst_win,
                                 hvk_secam_state_t *S, const int16_t *F, long fs, const int32_t *acc, long as,
                                 int dr, int sr, int phase_pos, int16_t *out, long os)
{
	const int W = C->W, sl = C->sl;
	const int16_t dmin = C->dmin[dr], dmax = C->dmax[dr];
	const int32_t level = C->level;
	const int fm_end = sr < W ? sr : W;
	double ix = S->ix, iy = S->iy;
	int32_t pi = phase_pos ? INT32_MAX : -INT32_MAX, pq = 0;
	int x;

	for(x = 0; x < W; x++)
	{
		int32_t f;
		int16_t y, v = 0;

		if(x < W - HVK_SECAM_TAIL) f = F[(long) x * fs];
		else
		{
			/* output x reads line[x - 7 + k]; entries W + i are tail[i]: k = W + 7 + i - x */
			int32_t a = acc[(long) (x - (W - HVK_SECAM_TAIL)) * as];
			int i;
			for(i = 0; i < HVK_SECAM_TAIL; i++)
			{
				const int k = W + 7 + i - x;
				if(k <= 14) a += (int32_t) S->tail[i] * C->fir[k];
			}
			a >>= 15;
			f = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
		}

		{
			const double in = (double) f;
			/* (the reference's expression, term by term; no contraction) */
			double t0 = in * 2.90456054;
			double t1 = ix * -2.80912108;
			double t2 = iy * -0.90456054;
			iy = (t0 + t1) - t2;
			ix = in;
			y = (int16_t) hvk_secam_round_away(iy < INT16_MIN ? INT16_MIN : (iy > INT16_MAX ? INT16_MAX : iy));
		}

		if(x >= sl && x < fm_end)
		{
			v = hvk_secam_fm_step(lut, bell, y, dmin, dmax, level, &pi, &pq);
			v = (int16_t) ((v * burst_win[x - sl]) >> 15);
		}
		if(out) out[(long) x * os] = v;
	}

	S->ix = ix;
	S->iy = iy;

	/* past the line the loop works on what lies behind it (src/video.c:3220-3229 with :4140-4147) */
	for(x = W; x < sr; x++)
	{
		S->tail[x - W] = hvk_secam_fm_step(lut, bell, S->tail[x - W], dmin, dmax, level, &pi, &pq);
	}
}

#endif
