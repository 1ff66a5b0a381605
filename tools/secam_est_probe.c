/* tools/secam_est_probe.c -- how often the estimate of hvk_k_secam_est (hvk_secam.hip) is right, on the CPU with the host chain as the truth:
 * the values behind a line from the summed angle of the FM steps after K lines from nothing, and the IIR state from a walk of the IIR alone over the last P samples.
 * gcc -O2 -Iinclude -Ihacktv_amd/csrc tools/secam_est_probe.c -Lhacktv_amd -lhvk -lm -Wl,-rpath,$PWD/hacktv_amd -o /tmp/est && /tmp/est l 16000000 1
 * arguments: mode, rate, pictures (0 bars, 1 noise, 2 gradients), x1 */
#include "../hacktv_amd/csrc/hvk_secam.c"
#include <stdio.h>
#include "hacktv_amd.h"

typedef struct { long S; double iyA; } a0_t;

static double kap0, kap1;  /* angle = kap0 + kap1 * v */

static int32_t ra(double x){ return hvk_secam_round_away(x); }

/* A0: zero-entry IIR over the line, sum of clamped u over [x1, min(fm_end, W-7)) ; iy at W-8 */
static void a0_line(hvk_secam_t *s, const int16_t *F, int dr, int sr, int x1, a0_t *o)
{
	const int W = s->W; int fm_end = sr < W ? sr : W; int hi = fm_end < W - 7 ? fm_end : W - 7;
	double ix = 0, iy = 0; long S = 0; int x;
	for(x = 0; x < W - 7; x++)
	{
		double in = F[x];
		iy = (in * 2.90456054 + ix * -2.80912108) - iy * -0.90456054; ix = in;
		if(x >= x1 && x < hi) { int32_t r = ra(iy); int32_t c = r < s->C.dmin[dr] ? s->C.dmin[dr] : (r > s->C.dmax[dr] ? s->C.dmax[dr] : r); S += c; }
	}
	o->S = S; o->iyA = iy;
}

typedef struct { double ix, iy; int16_t tail[8]; } est_t;

/* B step: line with entry estimate E (tail = entry tail of this line, ix/iy = entry) -> exit estimate */
static void b_line(hvk_secam_t *s, const int16_t *F, const int32_t *acc, const a0_t *A, int dr, int sr, int pos, int x1, est_t *E)
{
	const int W = s->W, sl = s->C.sl; int fm_end = sr < W ? sr : W;
	const int16_t dmin = s->C.dmin[dr], dmax = s->C.dmax[dr];
	double ix = E->ix, iy = E->iy; long S = 0; int x, n = 0;
	/* head */
	for(x = 0; x < x1; x++)
	{
		double in = F[x];
		iy = (in * 2.90456054 + ix * -2.80912108) - iy * -0.90456054; ix = in;
		if(x >= sl && x < fm_end) { int32_t r = ra(iy); S += r < dmin ? dmin : (r > dmax ? dmax : r); n++; }
	}
	{ int hi = fm_end < W - 7 ? fm_end : W - 7; if(hi > x1) { S += A->S; n += hi - x1; } }
	/* last 7 */
	ix = F[W - 8]; iy = A->iyA;
	for(x = W - 7; x < W; x++)
	{
		int32_t a = acc[x - (W - 7)]; int i;
		for(i = 0; i < 7; i++) { int k = W + 7 + i - x; if(k <= 14) a += (int32_t) E->tail[i] * s->C.fir[k]; }
		a >>= 15; a = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
		double in = a;
		iy = (in * 2.90456054 + ix * -2.80912108) - iy * -0.90456054; ix = in;
		if(x >= sl && x < fm_end) { int32_t r = ra(iy); S += r < dmin ? dmin : (r > dmax ? dmax : r); n++; }
	}
	E->ix = ix; E->iy = iy;
	/* tail steps */
	if(sr > W)
	{
		double th = (pos ? 0.0 : M_PI) + kap0 * n + kap1 * (double) S;
		double amp = (double) INT32_MAX - (double) n;
		for(x = W; x < sr; x++)
		{
			int16_t v = E->tail[x - W]; v = v < dmin ? dmin : (v > dmax ? dmax : v);
			th += kap0 + kap1 * v; amp -= 1.0; n++;
			int32_t pi = (int32_t) floor(amp * cos(th)), pq = (int32_t) floor(amp * sin(th));
			const hvk_secam_c16_t g = ((const hvk_secam_c16_t *) s->t->secam_bell)[(uint16_t) v];
			int32_t vi = ((pi >> 16) * s->C.level) >> 15, vq = ((pq >> 16) * s->C.level) >> 15;
			E->tail[x - W] = (int16_t) (((vi * g.i) >> 15) - ((vq * g.q) >> 15));
		}
	}
}

int main(int argc, char **argv)
{
	const char *mode = argc > 1 ? argv[1] : "l"; unsigned rate = argc > 2 ? atoi(argv[2]) : 16000000; int noisy = argc > 3 ? atoi(argv[3]) : 1;
	int x1 = argc > 4 ? atoi(argv[4]) : 256;
	hvk_config_t c; static hvk_tables_t t; hvk_config_preset(&c, mode);
	if(hvk_tables_build(&t, &c, rate, rate) != 0) { printf("tables failed\n"); return 1; }
	hvk_secam_t *s = hvk_secam_new(&t);
	const int W = s->W, fw = t.k.active_width, fh = t.k.active_lines;
	printf("W=%d sl=%d bw=%d active %dx%d\n", W, t.k.burst_left, t.k.burst_width, fw, fh);
	kap0 = 2.0 * M_PI / rate * FM_FREQ; kap1 = 2.0 * M_PI / rate * FM_DEV / INT16_MAX;
	uint32_t *fb = malloc((size_t) fw * fh * 4); int16_t *out = malloc((size_t) t.k.lines * W * 2);
	srand(5);
	int Ks[] = {2, 4, 6, 8, 12, 16}; long wrongT[6] = {0}, wrongI[6] = {0}, total = 0;
	long pre_bad[3] = {0}; int pres[3] = {384, 448, 512};
	for(int f = 0; f < 6; f++)
	{
		for(long i = 0; i < (long) fw * fh; i++)
		{
			if(noisy == 1) fb[i] = ((rand() & 0xFFF) << 12 | (rand() & 0xFFF)) & 0xFFFFFF;
			else if(noisy == 2) { int x = i % fw, y = i / fw; fb[i] = (((x * 255 / fw + f * 8) & 255) << 16) | (((y * 255 / fh) & 255) << 8) | (((x + y) / 4 + (rand() & 3)) & 255); }
			else { int x = i % fw; int b = x * 8 / fw; fb[i] = ((b & 4) ? 0xFF0000 : 0) | ((b & 2) ? 0xFF00 : 0) | ((b & 1) ? 0xFF : 0); }
		}
		hvk_secam_state_t st0 = s->state;
		hvk_secam_frame(s, f, fb, fw, fh, 0, fb, fw, fh, 0, out);
		/* true entry states */
		const int frame = f + 1, parity = frame & 1; const hvk_secam_task_t *T = s->tasks[parity]; const int nprime = f == 0 ? 2 : 0; const int n = s->ntasks[parity] + nprime;
		const hvk_kconst_t *k = &t.k;
		hvk_secam_state_t *tr = calloc(n + 1, sizeof(*tr)); a0_t *A = calloc(n, sizeof(*A));
		{
			hvk_secam_state_t st = st0;
			for(int i = 0; i < n; i++)
			{
				TASK_ARGS(i);
				if(clear_) memset(st.tail, 0, sizeof(st.tail));
				tr[i] = st;
				hvk_secam_chain_line(&s->C, (const hvk_secam_c32_t *) t.secam_lut, (const hvk_secam_c16_t *) t.secam_bell, t.burst_win, &st, s->F + (size_t) i * W, 1, s->acc + (size_t) i * 8, 1, dr_, sr_, pos_, NULL, 1);
				a0_line(s, s->F + (size_t) i * W, dr_, sr_, x1, &A[i]);
			}
			if(memcmp(&st, &s->state, sizeof(st))) printf("replay differs\n");
		}
		for(int ki = 0; ki < 6; ki++)
		{
			const int K = Ks[ki];
			for(int i = K + 1; i < n; i++)
			{
				est_t E; memset(&E, 0, sizeof(E));
				int16_t Tprev[8] = {0};
				for(int m = i - K; m < i; m++)
				{
					TASK_ARGS(m);
					if(clear_) memset(E.tail, 0, sizeof(E.tail));
					memcpy(Tprev, E.tail, sizeof(Tprev));
					b_line(s, s->F + (size_t) m * W, s->acc + (size_t) m * 8, &A[m], dr_, sr_, pos_, x1, &E);
				}
				{ TASK_ARGS(i); if(clear_) memset(E.tail, 0, sizeof(E.tail)); }
				if(memcmp(E.tail, tr[i].tail, 14)) wrongT[ki]++;
				if(ki == 4)
				{
					/* exact pre-walk: IIR over the last P samples of task i-1 with its last 7 from Tprev */
					for(int pi_ = 0; pi_ < 3; pi_++)
					{
						const int P = pres[pi_], m = i - 1; const int16_t *F = s->F + (size_t) m * W; const int32_t *acc = s->acc + (size_t) m * 8;
						double ix = 0, iy = 0;
						for(int x = W - P; x < W; x++)
						{
							double in;
							if(x < W - 7) in = F[x];
							else { int32_t a = acc[x - (W - 7)]; for(int q = 0; q < 7; q++) { int kk = W + 7 + q - x; if(kk <= 14) a += (int32_t) Tprev[q] * s->C.fir[kk]; } a >>= 15; in = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a); }
							iy = (in * 2.90456054 + ix * -2.80912108) - iy * -0.90456054; ix = in;
						}
						if(memcmp(&ix, &tr[i].ix, 8) || memcmp(&iy, &tr[i].iy, 8)) pre_bad[pi_]++;
					}
				}
			}
		}
		total += n;
		free(tr); free(A);
	}
	for(int ki = 0; ki < 6; ki++) printf("K=%2d wrong tails %ld of ~%ld\n", Ks[ki], wrongT[ki], total);
	for(int i = 0; i < 3; i++) printf("prewalk %d: wrong (ix,iy) %ld (K=12 tails)\n", pres[i], pre_bad[i]);
	return 0;
}
