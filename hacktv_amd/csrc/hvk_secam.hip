/* hvk_secam.hip -- the SECAM colour sub-carrier on the device (gfx950).
 *
 * The reference computes it line after line (src/video.c:3068-3233): what a line leaves behind -- the pre-emphasis
 * IIR's two doubles and the values behind the line's end, hvk_secam_state_t -- is the next line's start, for ever.
 * The arithmetic of a line is hvk_secam_chain.h, the same source the host's serial chain compiles (hvk_secam.c,
 * pinned against the reference on the CPU). Here the lines of a batch ("tasks": hvk_secam_tasks()) are worked on
 * all at once:
 *
 *   hvk_k_secam_cells   one workgroup per task, 8 samples per lane: the line's colour-difference cells (its own
 *                       pixels and the line above's through the level table) and the 15-tap low pass, without the
 *                       share of what lies behind the line; written transposed, 8 samples of one task per 16 bytes,
 *                       tasks side by side, so that a wave of the next kernel reads its 64 lines with one load.
 *                       This much depends on the picture and the parity of the frame's number only: where a frame
 *                       shows one picture the rows are kept per picture slot and parity (hvk_secam_args_t.cbase) and
 *                       made again only when the slot gets a new picture (.clist)
 *   hvk_k_secam_chain   one LANE per task: the serial walk over the line (IIR, FM phasor) -- after K warm-up lines,
 *                       the K tasks before it walked from a state of nothing, which leaves the lane with the state
 *                       its own line starts from in all but a few cases per ten thousand (the influence of a
 *                       line's start state on its end state shrinks by a factor of ~3 per line; K = 12). Tasks
 *                       whose warm-up reaches back to the batch's first task start from the true state carried
 *                       over from the batch before. Every lane records the state it assumed and the state it left
 *   hvk_k_secam_est     (round 4) entry states of new pictures' lines by estimate instead of warm-up walks: the values behind
 *                       a line from the summed angle of its FM steps, the IIR pair from a walk of the IIR alone
 *   hvk_k_secam_walk    (round 5) one line per lane from estimated or kept entry states -- no warm-up line anywhere in the
 *                       block --, the FM step computed and the bell filter's gain decoded from LDS (<1>) or both read from
 *                       the table (<0>); hvk_k_secam_chain stays for runs of several lines and for warm-up walks
 *   hvk_k_secam_check   entry state of task t == exit state of task t - 1, bit for bit? By induction from the
 *                       carried state every task that passes is exact
 *   hvk_k_secam_redo    the tasks that failed again from the exit state of the task before; then the check again, until
 *                       nothing fails (the engine gives up after HVK_SECAM_ROUNDS and sends the batch through the host's
 *                       chain). With one line a task: from the record the line's walk left in front of its last eight
 *                       samples (hvk_secam_mid_t) where only the values behind the line changed, the IIR alone from both
 *                       starts until they agree where the IIR pair changed, the whole line otherwise (round 5)
 *   hvk_k_secam_redo_fields  from the second round on: one lane per field, a stretch of any length in one launch
 *
 * Doubles: the IIR is evaluated term by term with contraction off (this file is compiled with -ffp-contract=off),
 * like the reference's x86-64 build. */
#include <hip/hip_runtime.h>
#include "hvk_device.h"         /* level_of(): a colour's levels computed instead of looked up */
#include "hvk_kernels.h"
#include "hvk_secam_chain.h"

typedef struct { short x, y, z, w; } lvl_t;    /* a level-table entry: (Y, U, V, -) */

/* (-, U, V) of a colour: from the 2^24-entry table, or -- pictures with too many colours for the table's lines to
 * be found in the caches again -- worked out on the spot like the table's entries (src/video.c:3917-3958) */
template<int LV>
__device__ __forceinline__ lvl_t lvl_of(const hvk_secam_args_t &a, const uint32_t rgb)
{
	if(LV)
	{
		const short4v q = level_of(rgb, *(const hvk_yuvparams_t *) a.yuvp);
		lvl_t r;
		r.x = q.x; r.y = q.y; r.z = q.z; r.w = 0;
		return(r);
	}
	return(((const lvl_t *) a.yuv)[rgb]);
}

/* task t of the batch -> its record; NULL state of affairs (a padding slot, the priming slots of any frame but the
 * stream's first) comes back as valid = false */
struct task_view {
	bool valid, clear, fid;
	int frame;          /* frame of the batch */
	int64_t fnum;       /* frame number counted from 1 */
	int line, prev_line, sr, dr, phase_pos;
};

__device__ __forceinline__ task_view task_of(const hvk_secam_args_t &a, const int t)
{
	task_view v;
	const int i = t / a.ntasks, slot = t - i * a.ntasks;
	const int64_t findex = a.first_frame + i;
	const int parity = (int) ((findex + 1) & 1);
	const hvk_secam_task_t q = a.tasks[parity * a.ntasks + slot];
	v.frame = i;
	v.fnum = findex + 1;
	v.valid = (q.flags & HVK_SECAM_TASK_VALID) != 0;
	if(slot < 2)
	{
		/* the pipeline's two fill slots before the stream's first line (hvk_secam.c): frame 1, line 0, no picture */
		v.valid = findex == 0;
		v.line = 0;
		v.prev_line = 0;
		v.sr = a.burst_left + a.burst_width;
		v.clear = false;
		v.fid = false;
	}
	else
	{
		v.line = q.line;
		v.prev_line = q.prev_line;
		v.sr = q.sr;
		v.clear = (q.flags & HVK_SECAM_TASK_CLEAR) != 0;
		v.fid = (q.flags & HVK_SECAM_TASK_FID) != 0;
	}
	{
		/* (frame * lines + line) as the reference's int */
		const int n = (int) v.fnum * a.lines + v.line;
		v.dr = n & 1;
		v.phase_pos = n % 3 == 0;
	}
	return(v);
}

/* hvk_secam_round_away() -- to the nearest whole number, halves away from zero -- without branches, comparisons or the
 * condition register: 2 x is exact, k = trunc(2 x) keeps every half, and round(x) = (k + 1) >> 1 for k >= 0, k >> 1 for k < 0
 * (floor(k / 2) = -ceil(|k| / 2)). Needs |2 x| < 2^31: the IIR's output stays below 2^21 (its gain is at most 3 and its
 * impulse response sums to less than 60, over int16 input). */
__device__ __forceinline__ int32_t round_away_nb(const double x)
{
	const int32_t k = (int32_t) (x + x);
	return((k + 1 + (k >> 31)) >> 1);
}

/* ------------------------------------------------------------------ */

#define CELLS_TASKS 1           /* tasks a workgroup works on side by side (whole waves each). Two were measured: 0.86 against 0.73 ms per 512 frames -- more small workgroups overlap better than fewer larger ones */
template<int LV>
__global__ __launch_bounds__(256 * CELLS_TASKS)
void hvk_k_secam_cells(const hvk_secam_args_t a)
{
	extern __shared__ __attribute__((aligned(16))) int16_t lds_all[];  /* per task: 8 zeros, W cells, 24 zeros */
	const int W = a.C.W;
	const int TH = CELLS_TASKS == 1 ? (int) blockDim.x : (int) blockDim.x / CELLS_TASKS;      /* lanes of a task: a multiple of 64 */
	const int sub = CELLS_TASKS == 1 ? 0 : (int) threadIdx.x / TH;
	int16_t *lds = lds_all + sub * ((W + 32 + 7) & ~7);
	/* (workgroups go to the 8 XCDs in turn; the 8 tasks whose outputs share 128 bytes of the transposed store go to ONE,
	 * so that its L2 sees whole lines: workgroup b of XCD b % 8 is the (b / 8)-th there, and 8 / CELLS_TASKS of them make a group) */
	const int bj = (int) blockIdx.x >> 3;
	constexpr int WPG = 8 / CELLS_TASKS;
	const int slot_ = (((bj / WPG) * 8 + ((int) blockIdx.x & 7)) << 3) + (bj % WPG) * CELLS_TASKS + sub;
	const bool in_list = slot_ < a.ntasks;
	if(CELLS_TASKS == 1 && !in_list) return;
	const int slot = in_list ? slot_ : a.ntasks - 1;
	const int i = a.clist[blockIdx.y];
	const int t = i * a.ntasks + slot;
	const int cm = a.cbase[i] + slot;           /* the task's row in the cell stores */
	const task_view v = task_of(a, t);
	const int lane = (int) threadIdx.x - sub * TH, x0 = lane * SPL;
	if(CELLS_TASKS == 1 && !v.valid) return;
	const bool live = in_list && v.valid;       /* (a task that is none takes part in the barriers and writes nothing) */

	int16_t c[SPL];

	if(!live)
	{
		for(int j = 0; j < SPL; j++) c[j] = 0;
	}
	else if(v.fid)
	{
		for(int j = 0; j < SPL; j++) c[j] = x0 + j < W ? a.fid_rows[v.dr * W + x0 + j] : 0;
	}
	else
	{
		const int comp = v.dr ? 1 : 0;
		int pcomp, have_prev;
		bool prime = slot < 2;
		/* geometry of the field's frame: like the raster's (hvk_device.h raster_setup_core) */
		const int second = a.fields == 2 && v.line >= a.hline;
		const hvk_framedesc_t f = a.fdesc[i * (a.fields + 1) + 1 + (second ? 1 : 0)];
		/* (the fill slots as well: they pass while the stream's first picture is the one in force, with its place and width
		 * and no row of it -- src/video.c:3135-3197 with vy = -1) */
		/* (... the second one; the FIRST is processed before the source has been read: the full active width, hvk_secam.c) */
		int fbw = (prime && slot == 0) ? a.active_width : (f.fb_valid ? f.fb_width : 0);
		int p0 = a.active_left + (a.active_width - fbw) / 2;
		int64_t row = -1, prow = -1;
		/* the rows of the (U, V) plane that hold this line's and the line above's levels whole (a half line's has only
		 * the half that is seen) */
		const int *uvr = NULL, *puvr = NULL;

		if(prime)
		{
			have_prev = slot == 1;
			pcomp = v.dr ? 0 : 1;       /* the first fill slot's OTHER component */
		}
		else
		{
			const int fbh = f.fb_valid ? f.fb_height : 0;
			const int vframe_y = (a.active_lines - fbh) / 2;
			const int parity = (int) (v.fnum & 1);
			int vy = a.desc[parity * a.lines + v.line - 1].src_row;
			if(vy >= 0 && a.interlaced != 0 && f.fb_interlaced != a.interlaced) vy += 1;
			vy -= vframe_y;
			if(f.fb_valid && vy >= 0 && vy < fbh) row = f.fb_offset + (int64_t) vy * f.line_stride;
			if(a.uvp && row >= 0)
			{
				const hvk_linedesc_t dl = a.desc[parity * a.lines + v.line - 1];
				if(dl.al <= p0 && dl.ar >= p0 + fbw) uvr = a.uvp + ((size_t) f.plane_row0 + v.line - 1) * W;
			}
			have_prev = v.prev_line != 0;
			{
				const int pn = (int) v.fnum * a.lines + v.prev_line;
				pcomp = (pn & 1) ? 0 : 1;
			}
			if(have_prev)
			{
				int py = a.desc[parity * a.lines + v.prev_line - 1].src_row;
				if(py >= 0 && a.interlaced != 0 && f.fb_interlaced != a.interlaced) py += 1;
				py -= vframe_y;
				if(f.fb_valid && py >= 0 && py < fbh) prow = f.fb_offset + (int64_t) py * f.line_stride;
				if(a.uvp && prow >= 0)
				{
					const hvk_linedesc_t dp = a.desc[parity * a.lines + v.prev_line - 1];
					if(dp.al <= p0 && dp.ar >= p0 + fbw) puvr = a.uvp + ((size_t) f.plane_row0 + v.prev_line - 1) * W;
				}
			}
		}

		const lvl_t black = ((const lvl_t *) a.yuv)[0];
		const int16_t rest = comp ? black.z : black.y;
		/* all of the lane's pixels first (at clamped positions: no load under a lane test, which would be waited for
		 * on the spot), then all of their level look-ups, then the cells */
		lvl_t m[SPL], pm[SPL];
		const bool inw = x0 < W;
		if(uvr)
		{
			/* (the plane's rows hold what lvl_of() gives for the pixels under [p0, p0 + fbw); the rest of a row is not looked at) */
			const int4 q0 = inw ? ((const int4 *) (uvr + x0))[0] : make_int4(0, 0, 0, 0), q1 = inw ? ((const int4 *) (uvr + x0))[1] : make_int4(0, 0, 0, 0);
			const int qq[SPL] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
			for(int j = 0; j < SPL; j++) { m[j].x = 0; m[j].y = (int16_t) qq[j]; m[j].z = (int16_t) (qq[j] >> 16); m[j].w = 0; }
		}
		else
		{
			uint32_t rgb[SPL];
#pragma unroll
			for(int j = 0; j < SPL; j++)
			{
				int xi = x0 + j - p0;
				xi = xi < 0 ? 0 : (xi < fbw ? xi : (fbw > 0 ? fbw - 1 : 0));
				rgb[j] = (row >= 0 && fbw > 0) ? (a.pool[row + xi] & 0xFFFFFF) : 0;
			}
#pragma unroll
			for(int j = 0; j < SPL; j++) m[j] = lvl_of<LV>(a, rgb[j]);
		}
		if(puvr)
		{
			const int4 q0 = inw ? ((const int4 *) (puvr + x0))[0] : make_int4(0, 0, 0, 0), q1 = inw ? ((const int4 *) (puvr + x0))[1] : make_int4(0, 0, 0, 0);
			const int qq[SPL] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
			for(int j = 0; j < SPL; j++) { pm[j].x = 0; pm[j].y = (int16_t) qq[j]; pm[j].z = (int16_t) (qq[j] >> 16); pm[j].w = 0; }
		}
		else
		{
			uint32_t prgb[SPL];
#pragma unroll
			for(int j = 0; j < SPL; j++)
			{
				int xi = x0 + j - p0;
				xi = xi < 0 ? 0 : (xi < fbw ? xi : (fbw > 0 ? fbw - 1 : 0));
				prgb[j] = (have_prev && prow >= 0 && fbw > 0) ? (a.pool[prow + xi] & 0xFFFFFF) : 0;
			}
#pragma unroll
			for(int j = 0; j < SPL; j++) pm[j] = lvl_of<LV>(a, prgb[j]);
		}
#pragma unroll
		for(int j = 0; j < SPL; j++)
		{
			const int x = x0 + j;
			int16_t cell = rest;
			if(x >= p0 && x < p0 + fbw)
			{
				const int held = have_prev ? (pcomp ? pm[j].z : pm[j].y) : 0;
				cell = (int16_t) (((int) (comp ? m[j].z : m[j].y) + held) / 2);
			}
			c[j] = x < W ? cell : 0;
		}
	}

	/* the cells in LDS behind 8 zeros (and in front of 24: the last lane's window), element 8 + x = cell x */
	if(lane == 0)
	{
		*(int4 *) lds = make_int4(0, 0, 0, 0);
		for(int j = 0; j < 24; j += 8) *(int4 *) (lds + 8 + W + j) = make_int4(0, 0, 0, 0);
	}
	if(x0 + SPL <= W)
	{
		int4 pk;
		pk.x = (uint16_t) c[0] | ((uint32_t) (uint16_t) c[1] << 16);
		pk.y = (uint16_t) c[2] | ((uint32_t) (uint16_t) c[3] << 16);
		pk.z = (uint16_t) c[4] | ((uint32_t) (uint16_t) c[5] << 16);
		pk.w = (uint16_t) c[6] | ((uint32_t) (uint16_t) c[7] << 16);
		*(int4 *) (lds + 8 + x0) = pk;
	}
	else for(int j = 0; j < SPL; j++) if(x0 + j < W) lds[8 + x0 + j] = c[j];
	__syncthreads();

	const bool act = live && x0 < W;

	/* 15-tap low pass, zero history (src/video.c:3207): output x reads cells x - 7 .. x + 7 = elements x + 1 .. x + 15 --
	 * the lane's window starts one element behind its 16-byte aligned slice; packed pairs and v_dot2c_i32_i16 */
	int32_t acc[SPL];
	int16_t o[SPL];
	if(act)
	{
		constexpr int ND = SPL / 2 + (15 + 1) / 2 + 1;
		int d[ND], tp[8];
		const int4v *pw = (const int4v *) (lds + x0);
#pragma unroll
		for(int q = 0; q < (ND + 3) / 4; q++)
		{
			const int4v w = pw[q];
			if(q * 4 + 0 < ND) d[q * 4 + 0] = w.x;
			if(q * 4 + 1 < ND) d[q * 4 + 1] = w.y;
			if(q * 4 + 2 < ND) d[q * 4 + 2] = w.z;
			if(q * 4 + 3 < ND) d[q * 4 + 3] = w.w;
		}
#pragma unroll
		for(int q = 0; q < 8; q++) tp[q] = ((int) a.C.fir[2 * q] & 0xFFFF) | ((q < 7 ? (int) a.C.fir[2 * q + 1] : 0) << 16);
		fir8<15, 1>(d, tp, acc);
	}
	else for(int j = 0; j < SPL; j++) acc[j] = 0;

	/* 8 outputs of one task in 16 bytes, tasks side by side */
	{
		for(int j = 0; j < SPL; j++)
		{
			int32_t s = acc[j] >> 15;
			o[j] = (int16_t) (s < INT16_MIN ? INT16_MIN : (s > INT16_MAX ? INT16_MAX : s));
		}
		int4 pk;
		pk.x = (uint16_t) o[0] | ((uint32_t) (uint16_t) o[1] << 16);
		pk.y = (uint16_t) o[2] | ((uint32_t) (uint16_t) o[3] << 16);
		pk.z = (uint16_t) o[4] | ((uint32_t) (uint16_t) o[5] << 16);
		pk.w = (uint16_t) o[6] | ((uint32_t) (uint16_t) o[7] << 16);
		if(act) ((int4 *) a.F)[(size_t) lane * a.cpad + cm] = pk;
	}
	/* the last 7 outputs also as they are before the shift: the share of what lies behind the line comes later */
	if(act && x0 + SPL >= W)
	{
		for(int j = 0; j < SPL; j++)
		{
			const int x = x0 + j;
			if(x >= W - HVK_SECAM_TAIL && x < W) a.acc[(size_t) cm * 8 + (x - (W - HVK_SECAM_TAIL))] = acc[j];
		}
	}

	/* For hvk_k_secam_est: the pre-emphasis IIR over the line from a start of nothing, all lanes at once -- every lane
	 * its eight samples with nothing carried in, the carries by a scan over the lanes (what a lane hands on reaches the
	 * next one times 0.90456054^8), then the samples again with the carry: the outputs to within rounding, which is all
	 * the table indices need. Left behind: the sum of the indices from x1 to the FM window's end short of the line's
	 * last seven samples, and the IIR's output at W - 8. */
	if(a.iya)
	{
		__shared__ double s_y_all[CELLS_TASKS][4];
		__shared__ int s_f_all[CELLS_TASKS][4];
		__shared__ int s_sum_all[CELLS_TASKS];
		double *s_y = s_y_all[sub];
		int *s_f = s_f_all[sub];
		int &s_sum = s_sum_all[sub];
		__shared__ int s_cph_all[CELLS_TASKS], s_cam_all[CELLS_TASKS];
		int &s_cph = s_cph_all[sub], &s_cam = s_cam_all[sub];
		const double CA = 2.90456054, CB = -2.80912108, CC = 0.90456054;
		const int wl = lane & 63, wv = lane >> 6;
		if(lane == 0) { s_sum = 0; s_cph = 0; s_cam = 0; }
		if(wl == 63) s_f[wv] = act ? (int) o[SPL - 1] : 0;
		__syncthreads();
		double yl[SPL];
		{
			int fp = __shfl_up(act ? (int) o[SPL - 1] : 0, 1);
			if(wl == 0) fp = wv ? s_f[wv - 1] : 0;
			double y = 0, xp = (double) fp;
#pragma unroll
			for(int j = 0; j < SPL; j++)
			{
				const double in = act ? (double) o[j] : 0.0;
				y = (in * CA + xp * CB) + y * CC;
				xp = in;
				yl[j] = y;
			}
		}
		/* (the scan inside a wave by shuffles; from the wave before only its last lane's value matters: what lies further
		 * back has shrunk by 0.448^64) */
		double sv = yl[SPL - 1], mul = CC * CC;
		mul *= mul; mul *= mul;         /* CC^8 */
		double pw = 1.0;                /* CC^(8 (wl + 1)) */
		{
			double sq = mul;
			for(int b = 0; b < 7; b++) { if((wl + 1) >> b & 1) pw *= sq; sq *= sq; }
		}
		for(int d = 1; d < 64; d <<= 1)
		{
			const double t = __shfl_up(sv, d);
			if(wl >= d) sv += mul * t;
			mul *= mul;
		}
		if(wl == 63) s_y[wv] = sv;
		__syncthreads();
		const double before = wv ? s_y[wv - 1] : 0.0;
		sv += pw * before;
		double cin = __shfl_up(sv, 1);
		if(wl == 0) cin = before;
		const int fm_end = v.sr < W ? v.sr : W;
		const int hi = fm_end < W - HVK_SECAM_TAIL ? fm_end : W - HVK_SECAM_TAIL;
		const int32_t dmin32 = a.C.dmin[v.dr], dmax32 = a.C.dmax[v.dr];
		int sum = 0;
		double p = CC;
		int uu[SPL];
#pragma unroll
		for(int j = 0; j < SPL; j++)
		{
			const double yj = yl[j] + p * cin;
			const int x = x0 + j;
			p *= CC;
			const int32_t r = round_away_nb(yj);
			uu[j] = r < dmin32 ? dmin32 : (r > dmax32 ? dmax32 : r);
			if(x >= a.x1 && x < hi) sum += uu[j];
			if(act && x == W - 8) a.iya[cm] = yj;
		}
		/* (a lane whose eight samples, all of them FM steps, take ONE table entry: that entry's rounding, eight times) */
		int cph = 0, cam = 0;
		if(a.res && act && x0 >= a.C.sl && x0 + SPL <= hi)
		{
			bool same = true;
#pragma unroll
			for(int j = 1; j < SPL; j++) same = same && uu[j] == uu[0];
			if(same)
			{
				const int rr = ((const int *) a.res)[uu[0] + 32768];
				cph = SPL * (int) (int16_t) rr;
				cam = SPL * (rr >> 16);
			}
		}
		for(int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
		if(wl == 0 && sum) atomicAdd(&s_sum, sum);
		/* (the lanes that have something: in a picture that is not flat that is the blanking's few) */
		if(cph) atomicAdd(&s_cph, cph);
		if(cam) atomicAdd(&s_cam, cam);
		__syncthreads();
		if(live && lane == 0)
		{
			a.acc[(size_t) cm * 8 + 7] = s_sum;
			if(a.corr) { a.corr[(size_t) cm * 2 + 0] = s_cph; a.corr[(size_t) cm * 2 + 1] = s_cam; }
		}
	}
}

/* ------------------------------------------------------------------ */

/* One line's walk by one lane (walk_fast, below): hvk_secam_chain_line() in chunks of 8 samples -- the low pass read 8 at a
 * time from the transposed store (the next chunk's load goes out before this chunk's arithmetic), the IIR over the whole
 * chunk first, then all of the chunk's table reads at once (plain loads: marked non-temporal they take twice as long --
 * what reuse there is happens in L1), then the FM recurrence, then the output 8 at a time. Same arithmetic, same order
 * per sample. (Chunks of 16 with 2 x 16 table reads in flight were the form of rounds 3 and 4: 134 registers a lane and
 * 155 spilled scalar registers; at 8 the kernels hold five to six waves per SIMD, which hides the reads as well.) */
__device__ __forceinline__ void unpack8(const int4 pk, int16_t *f)
{
	f[0] = (int16_t) pk.x; f[1] = (int16_t) (pk.x >> 16);
	f[2] = (int16_t) pk.y; f[3] = (int16_t) (pk.y >> 16);
	f[4] = (int16_t) pk.z; f[5] = (int16_t) (pk.z >> 16);
	f[6] = (int16_t) pk.w; f[7] = (int16_t) (pk.w >> 16);
}

__device__ __forceinline__ int4 pack8(const int16_t *o)
{
	int4 po;
	po.x = (uint16_t) o[0] | ((uint32_t) (uint16_t) o[1] << 16);
	po.y = (uint16_t) o[2] | ((uint32_t) (uint16_t) o[3] << 16);
	po.z = (uint16_t) o[4] | ((uint32_t) (uint16_t) o[5] << 16);
	po.w = (uint16_t) o[6] | ((uint32_t) (uint16_t) o[7] << 16);
	return(po);
}

__device__ __forceinline__ int16_t *out_of(const hvk_secam_args_t &a, const task_view &v)
{
	if(v.line < 1) return(NULL);        /* the fill slots are never seen */
	return(a.chroma + (size_t) (a.orow ? a.orow[v.frame] : v.frame) * a.raster_samples + (size_t) (v.line - 1) * a.C.W);
}

typedef struct { double x, y; } dbl2_t;
template<int PH, bool EMIT = true>
__device__ __forceinline__ void walk_fast(const hvk_secam_args_t &a, const dbl2_t *phc, const uint4 *bz, const int m, const task_view &v, hvk_secam_state_t &S, int16_t *out);

__device__ __forceinline__ void run_task(const hvk_secam_args_t &a, const int m, hvk_secam_state_t &S, const bool emit)
{
	const task_view v = task_of(a, m);
	if(!v.valid) return;
	if(v.clear) for(int i = 0; i < 8; i++) S.tail[i] = 0;
	if(emit) walk_fast<0, true>(a, NULL, NULL, m, v, S, out_of(a, v));
	else walk_fast<0, false>(a, NULL, NULL, m, v, S, NULL);
}

/* the row of task m in the store of kept states */
__device__ __forceinline__ int seed_row(const hvk_secam_args_t &a, const int m)
{
	const int i = m / a.ntasks;
	return(a.sbase[i] + (m - i * a.ntasks));
}

/* ------------------------------------------------------------------ */

#define IIR_STEP(in_) do { const double in__ = (in_); const double t0__ = in__ * 2.90456054, t1__ = ix * -2.80912108, t2__ = iy * -0.90456054; \
                           iy = (t0__ + t1__) - t2__; ix = in__; } while(0)

/* ... and for the estimate, which needs the outputs to within rounding only: the products fused (the chain from sample to
 * sample is one fused multiply-add long), the index by the add of 1.5 * 2^52 */
#ifndef EST_RING
#define EST_RING 4       /* (8 and 12 measured: 3 % and 7 % slower -- the kernel does not wait for these reads) */
#endif
#define EST_STEP(in_) do { const double in__ = (in_); iy = __builtin_fma(iy, 0.90456054, __builtin_fma(in__, 2.90456054, ix * -2.80912108)); ix = in__; } while(0)
#define EST_INDEX(iy_) med3i(__double2loint((iy_) + 6755399441055744.0), dmin32, dmax32)

/* a line's low-pass output x >= W - 7 given the values behind the line (hvk_secam_chain_line) */
__device__ __forceinline__ int32_t tail_output(const hvk_secam_args_t &a, const int cm, const int x, const int16_t *tail)
{
	const int W = a.C.W;
	int32_t s = a.acc[(size_t) cm * 8 + (x - (W - HVK_SECAM_TAIL))];
	for(int i = 0; i < HVK_SECAM_TAIL; i++)
	{
		const int k = W + 7 + i - x;
		if(k <= 14) s += (int32_t) tail[i] * a.C.fir[k];
	}
	s >>= 15;
	return(s < INT16_MIN ? INT16_MIN : (s > INT16_MAX ? INT16_MAX : s));
}

/* One line of the estimate: E on entry (the IIR's state to within rounding, the values behind the line), on exit.
 * The line's head under the entry state at hand; the middle from hvk_k_secam_cells; the last seven samples with the values
 * behind the line at hand; then the FM loop's steps past the line's end with the phasor the summed angle gives --
 * cos and sin of it at the amplitude the floor-after-every-step recurrence has lost one unit per step of -- through
 * the same integer arithmetic as hvk_secam_fm_step(). */
__device__ __forceinline__ void est_line(const hvk_secam_args_t &a, const int m, const task_view &v, hvk_secam_state_t &E)
{
	const int W = a.C.W, sl = a.C.sl;
	const int fm_end = v.sr < W ? v.sr : W;
	const int32_t dmin32 = a.C.dmin[v.dr], dmax32 = a.C.dmax[v.dr];
	const int cm = a.cbase[v.frame] + (m - v.frame * a.ntasks);
	const int4 *F = (const int4 *) a.F + cm;
	double ix = E.ix, iy = E.iy;
	int32_t S = 0;              /* (a line's indices: 942 of at most 14 028) */

	/* (EST_RING chunks in flight: the loads are a lane's own, 16 bytes each; the kernel is not bound by them -- 8 or 12 in flight
	 * change nothing) */
	const int nq = a.x1 / 8;
	int4 ring[EST_RING];
#pragma unroll
	for(int q = 0; q < EST_RING; q++) ring[q] = F[(size_t) (q < nq ? q : 0) * a.cpad];
	const int4 last = F[(size_t) ((W - 8) / 8) * a.cpad];
	const double iya = a.iya[cm];
	int32_t tacc[8];
	{
		const int4 *pa = (const int4 *) (a.acc + (size_t) cm * 8);
		const int4 a0 = pa[0], a1 = pa[1];
		tacc[0] = a0.x; tacc[1] = a0.y; tacc[2] = a0.z; tacc[3] = a0.w;
		tacc[4] = a1.x; tacc[5] = a1.y; tacc[6] = a1.z; tacc[7] = a1.w;
	}
	/* (up to the FM window the IIR alone) */
	const int q0 = sl / 8 < nq ? sl / 8 : nq;
	int q = 0;
	for(; q < q0; q++)
	{
		int16_t f[8];
		unpack8(ring[0], f);
#pragma unroll
		for(int i = 0; i + 1 < EST_RING; i++) ring[i] = ring[i + 1];
		if(q + EST_RING < nq) ring[EST_RING - 1] = F[(size_t) (q + EST_RING) * a.cpad];
#pragma unroll
		for(int j = 0; j < 8; j++) EST_STEP((double) f[j]);
	}
	for(; q < nq; q++)
	{
		int16_t f[8];
		unpack8(ring[0], f);
#pragma unroll
		for(int i = 0; i + 1 < EST_RING; i++) ring[i] = ring[i + 1];
		if(q + EST_RING < nq) ring[EST_RING - 1] = F[(size_t) (q + EST_RING) * a.cpad];
#pragma unroll
		for(int j = 0; j < 8; j++)
		{
			const int x = q * 8 + j;
			EST_STEP((double) f[j]);
			if(x >= sl && x < fm_end)
			{
				S += EST_INDEX(iy);
			}
		}
	}
	S += tacc[7];

	ix = (double) (int16_t) last.x;
	iy = iya;
#pragma unroll
	for(int j = 0; j < HVK_SECAM_TAIL; j++)
	{
		const int x = W - HVK_SECAM_TAIL + j;
		int32_t s = tacc[j];
#pragma unroll
		for(int i = 0; i <= j; i++) s += (int32_t) E.tail[i] * a.C.fir[14 + i - j];     /* (tap W + 7 + i - x) */
		s >>= 15;
		EST_STEP((double) (s < INT16_MIN ? INT16_MIN : (s > INT16_MAX ? INT16_MAX : s)));
		if(x >= sl && x < fm_end)
		{
			S += EST_INDEX(iy);
		}
	}
	E.ix = ix;
	E.iy = iy;

	if(v.sr > W)
	{
		const int n = fm_end > sl ? fm_end - sl : 0;
		double th = (v.phase_pos ? 0.0 : 3.14159265358979323846) + a.kap0 * (double) n + a.kap1 * (double) S;
		double amp = 2147483647.0 - (double) n;
		if(a.corr)
		{
			/* (what the entries' rounding adds up to over the line's stretches of one colour: hvk_secam_args_t.res) */
			th += (double) a.corr[(size_t) cm * 2 + 0] * 0x1p-46;
			amp *= 1.0 + (double) a.corr[(size_t) cm * 2 + 1] * 0x1p-46;
		}
		for(int x = W; x < v.sr; x++)
		{
			int16_t c = E.tail[x - W];
			c = c < (int16_t) dmin32 ? (int16_t) dmin32 : (c > (int16_t) dmax32 ? (int16_t) dmax32 : c);
			th += a.kap0 + a.kap1 * (double) c;
			amp -= 1.0;
			double sn, cs;
			sincos(th, &sn, &cs);
			const int32_t pi = (int32_t) floor(amp * cs), pq = (int32_t) floor(amp * sn);
			const hvk_secam_c16_t g = a.bell[(uint16_t) c];
			const int32_t vi = ((pi >> 16) * a.C.level) >> 15;
			const int32_t vq = ((pq >> 16) * a.C.level) >> 15;
			E.tail[x - W] = (int16_t) (((vi * g.i) >> 15) - ((vq * g.q) >> 15));
		}
	}
}

/* One lane per a.ES consecutive tasks: the values behind the line at each of their entries, from a.EK lines further up
 * walked this way from a state of nothing (the estimate's own errors fade like the walk's: by a factor of about 3 per
 * line). est[m]: [0..7] at task m's entry (before a field's first line clears them), [8..15] what the valid task
 * before it worked with. */
__global__ __launch_bounds__(64)
void hvk_k_secam_est(const hvk_secam_args_t a)
{
	const int g = blockIdx.x * 64 + threadIdx.x;
	const int t0 = g * a.ES;
	if(t0 >= a.total) return;
	const int t1 = t0 + a.ES < a.total ? t0 + a.ES : a.total;
	if(a.kf)
	{
		/* only where a frame asks for it (a run's start looks at most at the task before the segment's first) */
		const int tn = t1 < a.total ? t1 : a.total - 1;
		if(a.kf[t0 / a.ntasks] >= 0 && a.kf[(t1 - 1) / a.ntasks] >= 0 && a.kf[tn / a.ntasks] >= 0) return;
	}

	hvk_secam_state_t E;
	int16_t used[8];
	int m = t0 - a.EK;
	if(m <= 0) { m = 0; E = *a.carry; }
	else { E.ix = 0; E.iy = 0; for(int i = 0; i < 8; i++) E.tail[i] = 0; }
	for(int i = 0; i < 8; i++) used[i] = E.tail[i];

	for(; m < t1; m++)
	{
		if(m >= t0)
		{
			int4 *o = (int4 *) (a.est + (size_t) m * 16);
			o[0] = pack8(E.tail);
			o[1] = pack8(used);
		}
		if(m == t1 - 1) break;
		const task_view v = task_of(a, m);
		if(!v.valid) continue;
		if(v.clear) for(int i = 0; i < 8; i++) E.tail[i] = 0;
		for(int i = 0; i < 8; i++) used[i] = E.tail[i];
		est_line(a, m, v, E);
	}
}

/* A run's entry state from the estimate: the values behind the line as hvk_k_secam_est left them; the IIR's two doubles
 * exactly, by walking the IIR alone over the valid task before, from nothing -- with that task's last seven outputs made
 * with the values IT had behind its line. Over changing input it has arrived bit for bit at the state of the full walk
 * after 450 samples (its pole is 0.90456: 2^-53 after 360 samples, and equal doubles stay equal); over CONSTANT input --
 * a picture of one colour -- it has not: the recurrence then has several stationary values a few units of the last place
 * apart, and which of them a walk settles on depends on the side it comes from. Hence the whole line: the walk from
 * nothing passes the line's blanking level and the picture's left edge like the true one and comes to the picture's
 * level from the same side (tools/secam_est_probe.c, tools/secam_flat_probe.py: pictures of one colour had every line of
 * one kind start wrong with 448 samples, none with the line). */
#define PREWALK 4096
__device__ __forceinline__ void est_entry(const hvk_secam_args_t &a, const int t0, hvk_secam_state_t &S)
{
	int mp = t0 - 1;
	while(mp >= 0 && !task_of(a, mp).valid) mp--;
	if(mp < 0) { S = *a.carry; return; }

	const int W = a.C.W;
	const task_view v = task_of(a, mp);
	const int cm = a.cbase[v.frame] + (mp - v.frame * a.ntasks);
	const int4 *F = (const int4 *) a.F + cm;
	const int4 *e = (const int4 *) (a.est + (size_t) t0 * 16);
	int16_t used[8];
	double ix = 0, iy = 0;
	const int q1 = W / 8 - 1;
	int q = q1 - PREWALK / 8 + 1;
	if(q < 0) q = 0;

	unpack8(e[1], used);
	int4 nx = F[(size_t) q * a.cpad];
	for(; q < q1; q++)
	{
		int16_t f[8];
		unpack8(nx, f);
		nx = F[(size_t) (q + 1) * a.cpad];
#pragma unroll
		for(int j = 0; j < 8; j++) IIR_STEP((double) f[j]);
	}
	IIR_STEP((double) (int16_t) nx.x);
	for(int x = W - HVK_SECAM_TAIL; x < W; x++) IIR_STEP((double) tail_output(a, cm, x, used));

	S.ix = ix;
	S.iy = iy;
	unpack8(e[0], S.tail);
}

/* One lane per RUN of a.R consecutive tasks (a.R = 1 unless the batch has more tasks than eight waves per SIMD hold:
 * then the warm-up is shared by the run's lines) */
__global__ __launch_bounds__(64)
void hvk_k_secam_chain(const hvk_secam_args_t a)
{
	const int r = blockIdx.x * 64 + threadIdx.x;
	if(r >= a.nruns) return;
	const int t0 = r * a.R, t1 = t0 + a.R < a.total ? t0 + a.R : a.total;

	hvk_secam_state_t S;
	const int kk = a.kf ? a.kf[t0 / a.ntasks] : (a.est ? -1 : a.K);
	int m = t0;
	if(kk < 0) est_entry(a, t0, S);
	else
	{
		m = t0 - kk;
		if(m <= 0) { m = 0; S = *a.carry; }
		else if(a.seed) S = a.seed[seed_row(a, m)];
		else { S.ix = 0; S.iy = 0; for(int i = 0; i < 8; i++) S.tail[i] = 0; }
	}

	for(; m < t0; m++) run_task(a, m, S, false);
	a.entry[r] = S;
	for(; m < t1; m++)
	{
		if(a.seed && (!a.owner || a.owner[m / a.ntasks])) a.seed[seed_row(a, m)] = S;
		run_task(a, m, S, true);
	}
	a.exit[r] = S;
}

/* ---- hvk_k_secam_walk: the walk of ONE line per lane from an estimated or kept entry state (no warm-up lines, no runs
 * of several tasks: what a settled stream and a stream of new pictures both end at). The chain kernel above keeps every
 * other case; this one is what the time goes to, so it carries nothing it does not use. PH = 1: no table read from HBM
 * per sample (hvk_secam_args_t.phc). ---- */
/* the FM step of index c, computed: lround(INT32_MAX (cos, sin)(kap0 + kap1 c)) as the product of the coarse phasor of
 * c's 128-index cell and the fine one of its place in the cell (hvk_secam_args_t.phc; tried on every index at open) */
__device__ __forceinline__ void ph_step(const dbl2_t *phc, const double k1, const int c, int32_t &si, int32_t &sq)
{
	const int t = c + (32768 + 64);
	const dbl2_t C = phc[t >> 7];
	const double x = (double) ((t & 127) - 64) * k1, x2 = x * x;
	const double sf = x * __builtin_fma(x2, -1.0 / 6.0, 1.0);
	const double cf = __builtin_fma(x2, __builtin_fma(x2, 1.0 / 24.0, -0.5), 1.0);
	si = __double2int_rn(__builtin_fma(-C.y, sf, C.x * cf));
	sq = __double2int_rn(__builtin_fma(C.x, sf, C.y * cf));
}

/* the bell filter's gain at index c from its 32-index block: the values at the block's first index and a bit per step */
__device__ __forceinline__ void bell_gain(const uint4 *bz, const int c0, const int c, int32_t &gi, int32_t &gq)
{
	const unsigned t = (unsigned) (c - c0);
	const uint4 b = bz[t >> 5];
	const unsigned m = (1u << (t & 31)) - 1u;
	gq = (int32_t) (int16_t) (b.x >> 16) + __popc(b.y & m);
	gi = (int32_t) (int16_t) b.x + __popc(b.z & m) - __popc(b.w & m);
}

/* One FM step with the table's entries at hand (hvk_secam_fm_step's arithmetic) and the sample it leaves. EMIT = false: a
 * warm-up line -- only the state it leaves matters, so the output half of a step (level, bell-filter gain and its table
 * read, burst window) is left out */
#define WALK_FM(si_, sq_, gi_, gq_, bw_) do { \
		const int64_t ni__ = (int64_t) pi * (si_) - (int64_t) pq * (sq_); \
		const int64_t nq__ = (int64_t) pi * (sq_) + (int64_t) pq * (si_); \
		pi = (int32_t) (ni__ >> 31); \
		pq = (int32_t) (nq__ >> 31); \
		if(EMIT) { \
		const int32_t vi__ = ((pi >> 16) * level) >> 15; \
		const int32_t vq__ = ((pq >> 16) * level) >> 15; \
		vv = (int16_t) (((vi__ * (gi_)) >> 15) - ((vq__ * (gq_)) >> 15)); \
		vv = (int16_t) ((vv * (bw_)) >> 15); } } while(0)

template<int PH, bool EMIT>
__device__ __forceinline__ void walk_fast(const hvk_secam_args_t &a, const dbl2_t *phc, const uint4 *bz, const int m, const task_view &v, hvk_secam_state_t &S, int16_t *out)
{
	const int W = a.C.W, sl = a.C.sl;
	const int32_t dmin32 = a.C.dmin[v.dr], dmax32 = a.C.dmax[v.dr];
	const int32_t level = a.C.level;
	const int fm_end = v.sr < W ? v.sr : W;
	const double k1 = a.ph_k1;
	const int c0 = a.bell_c0;
	double ix = S.ix, iy = S.iy;
	int32_t pi = v.phase_pos ? INT32_MAX : -INT32_MAX, pq = 0;
	const int cm = a.cbase[v.frame] + (m - v.frame * a.ntasks);     /* the task's row in the cell stores */
	const int4 *F = (const int4 *) a.F + cm;
	const int chunks = W / 8;

	int4 nx = F[0];
	for(int q = 0; q < chunks; q++)
	{
		int16_t f[8], o[8];
		int32_t c[8];
		unpack8(nx, f);
		if(q + 1 < chunks) nx = F[(size_t) (q + 1) * a.cpad];
		if(q == chunks - 1)
		{
			int32_t ac[7];
			{
				const int4 *pa = (const int4 *) (a.acc + (size_t) cm * 8);
				const int4 a0 = pa[0], a1 = pa[1];
				ac[0] = a0.x; ac[1] = a0.y; ac[2] = a0.z; ac[3] = a0.w; ac[4] = a1.x; ac[5] = a1.y; ac[6] = a1.z;
			}
			if(EMIT && a.mid)
			{
				/* (five 16-byte stores) */
				int4 *pm = (int4 *) (a.mid + m);
				const int2 xi = __builtin_bit_cast(int2, ix), yi = __builtin_bit_cast(int2, iy);
				pm[0] = make_int4(xi.x, xi.y, yi.x, yi.y);
				pm[1] = nx;
				pm[2] = make_int4(pi, pq, ac[0], ac[1]);
				pm[3] = make_int4(ac[2], ac[3], ac[4], ac[5]);
				pm[4] = make_int4(ac[6], 0, 0, 0);
			}
			/* the last 7: with what lies behind the line (hvk_secam_chain_line) */
#pragma unroll
			for(int j = 1; j < 8; j++)
			{
				int32_t s = ac[j - 1];
#pragma unroll
				for(int i = 0; i < j; i++) s += (int32_t) S.tail[i] * a.C.fir[15 + i - j];      /* (tap W + 7 + i - x, x = W - 8 + j) */
				s >>= 15;
				f[j] = (int16_t) (s < INT16_MIN ? INT16_MIN : (s > INT16_MAX ? INT16_MAX : s));
			}
		}
#pragma unroll
		for(int j = 0; j < 8; j++)
		{
			const double in = (double) f[j];
			const double t0 = in * 2.90456054;
			const double t1 = ix * -2.80912108;
			const double t2 = iy * -0.90456054;
			iy = (t0 + t1) - t2;
			ix = in;
			c[j] = med3i(round_away_nb(iy), dmin32, dmax32);     /* (src/fir.c:729-733 and src/video.c:3215-3216 in one: walk_line) */
		}

		const int x0 = q * 8;
		if(x0 + 8 > sl && x0 < fm_end)
		{
			int4 tq[8];
			if(!PH && EMIT)
			{
				/* (step and bell-filter gain of an index side by side: one 16-byte read, one cache line per sample) */
#pragma unroll
				for(int j = 0; j < 8; j++) tq[j] = ((const int4 *) a.lutb)[(unsigned) (c[j] + 32768)];
			}
			if(!PH && !EMIT)
			{
#pragma unroll
				for(int j = 0; j < 8; j++) { const hvk_secam_c32_t st = a.lut[(unsigned) (c[j] + 32768)]; tq[j].x = st.i; tq[j].y = st.q; tq[j].z = 0; }
			}
			const int16_t *bw = a.burst_win + (x0 - sl);
			if(__all(x0 >= sl && x0 + 8 <= fm_end))
			{
				/* the chunk lies inside the sub-carrier window of every line of the wave: no test per sample */
#pragma unroll
				for(int j = 0; j < 8; j++)
				{
					int32_t si, sq, gi, gq;
					int16_t vv = 0;
					if(PH) { ph_step(phc, k1, c[j], si, sq); bell_gain(bz, c0, c[j], gi, gq); }
					else { si = tq[j].x; sq = tq[j].y; gi = (int16_t) tq[j].z; gq = (int16_t) (tq[j].z >> 16); }
					WALK_FM(si, sq, gi, gq, bw[j]);
					o[j] = vv;
				}
			}
			else
			{
#pragma unroll
				for(int j = 0; j < 8; j++)
				{
					const int x = x0 + j;
					int16_t vv = 0;
					if(x >= sl && x < fm_end)
					{
						int32_t si, sq, gi, gq;
						if(PH) { ph_step(phc, k1, c[j], si, sq); bell_gain(bz, c0, c[j], gi, gq); }
						else { si = tq[j].x; sq = tq[j].y; gi = (int16_t) tq[j].z; gq = (int16_t) (tq[j].z >> 16); }
						WALK_FM(si, sq, gi, gq, bw[j]);
					}
					o[j] = vv;
				}
			}
		}
		else
		{
#pragma unroll
			for(int j = 0; j < 8; j++) o[j] = 0;
		}
		if(EMIT && out) *(int4 *) (out + x0) = pack8(o);
	}

	S.ix = ix;
	S.iy = iy;
	/* (past the line the loop works on what lies behind it: two steps at 16 MHz, through the table like hvk_secam_chain_line) */
	const int16_t dmin = (int16_t) dmin32, dmax = (int16_t) dmax32;
#pragma unroll
	for(int i = 0; i < HVK_SECAM_TAIL; i++)
	{
		if(W + i < v.sr) S.tail[i] = hvk_secam_fm_step(a.lut, a.bell, S.tail[i], dmin, dmax, level, &pi, &pq);
	}
}

#define WALK_THREADS 256
#define WALK_PHC 520            /* 16-byte entries of LDS kept for the coarse phasors (513 used) */
template<int PH>
__global__ __launch_bounds__(WALK_THREADS)
void hvk_k_secam_walk(const hvk_secam_args_t a)
{
	extern __shared__ uint4 walk_lds[];
	if(PH)
	{
		for(int i = threadIdx.x; i < 513; i += WALK_THREADS) walk_lds[i] = ((const uint4 *) a.phc)[i];
		for(int i = threadIdx.x; i < a.bell_blocks; i += WALK_THREADS) walk_lds[WALK_PHC + i] = ((const uint4 *) a.bellz)[i];
		__syncthreads();
	}
	const int r = blockIdx.x * WALK_THREADS + threadIdx.x;
	if(r >= a.total) return;
	const int fr = r / a.ntasks;
	const bool last_task = r - fr * a.ntasks == a.ntasks - 1;
	if(a.mflag && a.mflag[fr])
	{
		/* the frame takes what its picture's last walk with this frame number modulo 6 left: its lines' rows lie in the store
		 * already, a line started from the state kept for it and left the one kept for the next (the frame's last: seedx).
		 * Whether the frame really starts from the state that walk started from is the check's to say. */
		const int row = seed_row(a, r);
		a.entry[r] = a.seed[row];
		a.exit[r] = last_task ? a.seedx[a.mrow[fr]] : a.seed[row + 1];
		return;
	}
	const bool keep = !a.owner || a.owner[fr];

	hvk_secam_state_t S;
	const int kk = a.kf ? a.kf[fr] : (a.est ? -1 : 0);
	if(kk < 0) est_entry(a, r, S);
	else if(r == 0) S = *a.carry;
	else if(a.seed) S = a.seed[seed_row(a, r)];
	else { S.ix = 0; S.iy = 0; for(int i = 0; i < 8; i++) S.tail[i] = 0; }
	a.entry[r] = S;
	if(a.seed && keep) a.seed[seed_row(a, r)] = S;
	const task_view v = task_of(a, r);
	if(v.valid)
	{
		if(v.clear) for(int i = 0; i < 8; i++) S.tail[i] = 0;
		walk_fast<PH>(a, (const dbl2_t *) walk_lds, walk_lds + WALK_PHC, r, v, S, out_of(a, v));
	}
	a.exit[r] = S;
	if(a.owner && a.owner[fr] && last_task) a.seedx[a.mrow[fr]] = S;
}

/* every index of the deviation range: the computed FM step and the decoded gain against the tables' entries */
__global__ void hvk_k_secam_check_walk(const hvk_secam_args_t a, const int c_lo, const int c_hi, int *differ)
{
	const int c = c_lo + blockIdx.x * blockDim.x + threadIdx.x;
	if(c > c_hi) return;
	int32_t si, sq, gi, gq;
	ph_step((const dbl2_t *) a.phc, a.ph_k1, c, si, sq);
	bell_gain((const uint4 *) a.bellz, a.bell_c0, c, gi, gq);
	const hvk_secam_c32_t st = a.lut[c + 32768];
	const hvk_secam_c16_t g = a.bell[(uint16_t) (int16_t) c];
	if(si != st.i || sq != st.q || gi != g.i || gq != g.q) atomicAdd(differ, 1);
}

__device__ __forceinline__ bool same_state(const hvk_secam_state_t &p, const hvk_secam_state_t &q)
{
	bool same = __double_as_longlong(p.ix) == __double_as_longlong(q.ix) && __double_as_longlong(p.iy) == __double_as_longlong(q.iy);
	for(int i = 0; i < HVK_SECAM_TAIL; i++) same = same && p.tail[i] == q.tail[i];
	return(same);
}

/* flags[r] = run r started from a state the run before did not leave; count[0] += failures */
__global__ void hvk_k_secam_check(const hvk_secam_args_t a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= a.nruns) return;
	const hvk_secam_state_t want = r ? a.exit[r - 1] : *a.carry;
	const bool bad = !same_state(a.entry[r], want);
	a.flags[r] = bad;
	if(bad) atomicAdd(a.count, 1);
}

/* The runs that failed, again. A lane takes the FIRST run of a stretch of failed ones -- its start is the exit state of
 * a run that passed, which nobody writes in this launch -- and walks on from there, run after run, for as long as the
 * state it leaves is not the one the next run had started from (whether that run had failed the check or not: a
 * corrected exit state makes the next run's start wrong too). It stops in front of a run that passed and is followed
 * by a failed one: that run's exit state is what another lane of this launch starts from. Whatever is left
 * inconsistent -- that case, hvk_k_secam_check finds it -- is next round's. */
/* A line again from a start that differs from the one it had only in the values behind the line (the IIR pair equal to the bit):
 * those values enter the line's last eight samples and what the line leaves, nothing else. From the record its walk left
 * (hvk_secam_mid_t) -- read, with the task's record and the start the line had, while the line BEFORE is worked on (redo_pre):
 * nothing of it depends on the state handed on --, the two table entries of the steps past the line asked for at once (their
 * indices are the values behind the line), so that a line of a stretch costs one dependent read: the eight samples' entries. */
struct redo_pre_t {
	hvk_secam_state_t entry;
	hvk_secam_task_t q;
	int4 m0, m1, m2, m3, m4;
	int flag, flag_next, cbase;
};

__device__ __forceinline__ redo_pre_t redo_pre(const hvk_secam_args_t &a, const int r)
{
	redo_pre_t p;
	const int i = r / a.ntasks, slot = r - i * a.ntasks;
	const int parity = (int) ((a.first_frame + i + 1) & 1);
	p.entry = a.entry[r];
	p.q = a.tasks[parity * a.ntasks + slot];
	const int4 *pm = (const int4 *) (a.mid + r);
	p.m0 = pm[0]; p.m1 = pm[1]; p.m2 = pm[2]; p.m3 = pm[3]; p.m4 = pm[4];
	p.flag = a.flags[r];
	p.flag_next = r + 1 < a.nruns ? a.flags[r + 1] : 0;
	p.cbase = a.cbase[i] + slot;        /* the task's row in the cell stores */
	return(p);
}

__device__ __forceinline__ void redo_tail(const hvk_secam_args_t &a, const redo_pre_t &p, const task_view &v, hvk_secam_state_t &S, int16_t *out)
{
	const int W = a.C.W, sl = a.C.sl;
	const int32_t dmin32 = a.C.dmin[v.dr], dmax32 = a.C.dmax[v.dr];
	const int32_t level = a.C.level;
	const int fm_end = v.sr < W ? v.sr : W;
	const int x0 = W - 8;

	/* the steps past the line: their table entries now */
	hvk_secam_c32_t tst[HVK_SECAM_TAIL];
	hvk_secam_c16_t tg[HVK_SECAM_TAIL];
#pragma unroll
	for(int i = 0; i < HVK_SECAM_TAIL; i++)
	{
		if(W + i < v.sr)
		{
			const int32_t c = med3i((int32_t) S.tail[i], dmin32, dmax32);
			tst[i] = a.lut[c + 32768];
			tg[i] = a.bell[(uint16_t) (int16_t) c];
		}
	}

	double ix = __builtin_bit_cast(double, make_int2(p.m0.x, p.m0.y)), iy = __builtin_bit_cast(double, make_int2(p.m0.z, p.m0.w));
	int32_t pi = p.m2.x, pq = p.m2.y;
	const int32_t ac[7] = { p.m2.z, p.m2.w, p.m3.x, p.m3.y, p.m3.z, p.m3.w, p.m4.x };
	int16_t f[8], o[8];
	int32_t c[8];
	unpack8(p.m1, f);
#pragma unroll
	for(int j = 1; j < 8; j++)
	{
		int32_t s = ac[j - 1];
#pragma unroll
		for(int i = 0; i < j; i++) s += (int32_t) S.tail[i] * a.C.fir[15 + i - j];
		s >>= 15;
		f[j] = (int16_t) (s < INT16_MIN ? INT16_MIN : (s > INT16_MAX ? INT16_MAX : s));
	}
#pragma unroll
	for(int j = 0; j < 8; j++)
	{
		const double in = (double) f[j];
		const double t0 = in * 2.90456054;
		const double t1 = ix * -2.80912108;
		const double t2 = iy * -0.90456054;
		iy = (t0 + t1) - t2;
		ix = in;
		c[j] = med3i(round_away_nb(iy), dmin32, dmax32);
	}
	if(x0 + 8 > sl && x0 < fm_end)
	{
		int4 tq[8];
#pragma unroll
		for(int j = 0; j < 8; j++) tq[j] = ((const int4 *) a.lutb)[(unsigned) (c[j] + 32768)];
		const int16_t *bw = a.burst_win + (x0 - sl);
		constexpr bool EMIT = true;
#pragma unroll
		for(int j = 0; j < 8; j++)
		{
			const int x = x0 + j;
			int16_t vv = 0;
			if(x >= sl && x < fm_end)
			{
				const int32_t si = tq[j].x, sq = tq[j].y, gi = (int16_t) tq[j].z, gq = (int16_t) (tq[j].z >> 16);
				WALK_FM(si, sq, gi, gq, bw[j]);
			}
			o[j] = vv;
		}
	}
	else
	{
#pragma unroll
		for(int j = 0; j < 8; j++) o[j] = 0;
	}
	if(out) *(int4 *) (out + x0) = pack8(o);

	S.ix = ix;
	S.iy = iy;
#pragma unroll
	for(int i = 0; i < HVK_SECAM_TAIL; i++)
	{
		if(W + i < v.sr)
		{
			/* hvk_secam_fm_step() with the entries at hand */
			const int64_t ni = (int64_t) pi * tst[i].i - (int64_t) pq * tst[i].q;
			const int64_t nq = (int64_t) pi * tst[i].q + (int64_t) pq * tst[i].i;
			pi = (int32_t) (ni >> 31);
			pq = (int32_t) (nq >> 31);
			const int32_t vi = ((pi >> 16) * level) >> 15, vq = ((pq >> 16) * level) >> 15;
			S.tail[i] = (int16_t) (((vi * tg[i].i) >> 15) - ((vq * tg[i].q) >> 15));
		}
	}
}

/* A line again from a start whose IIR pair differs from the one it had (by a unit of the last place or two: the seven samples
 * in front of it changed): the IIR alone from both starts side by side. Its pole is 0.90456 -- over changing input the two
 * are equal to the bit after a few hundred samples (equal doubles stay equal), and if every table index of the FM window on the
 * way was the same for both, the line's walk from there on, the FM phasor and every output up to there are what they were: true. A differing
 * index, or no agreement in front of the line's last eight samples: false (the whole line is walked). */
__device__ __forceinline__ bool redo_converges(const hvk_secam_args_t &a, const int cm, const int32_t dmin32, const int32_t dmax32, const int fm_end,
                                               const hvk_secam_state_t &was, const hvk_secam_state_t &S)
{
	const int4 *F = (const int4 *) a.F + cm;
	const int last = a.C.W / 8 - 1;
	double ax = was.ix, ay = was.iy, bx = S.ix, by = S.iy;
	int4 ring[4];
#pragma unroll
	for(int q = 0; q < 4; q++) ring[q] = F[(size_t) (q < last ? q : 0) * a.cpad];
	for(int q = 0; q < last; q++)
	{
		int16_t f[8];
		unpack8(ring[0], f);
		ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = ring[3];
		if(q + 4 < last) ring[3] = F[(size_t) (q + 4) * a.cpad];
		bool differ = false;
#pragma unroll
		for(int j = 0; j < 8; j++)
		{
			const double in = (double) f[j];
			const double t0 = in * 2.90456054;
			const double u1 = ax * -2.80912108, u2 = ay * -0.90456054;
			const double w1 = bx * -2.80912108, w2 = by * -0.90456054;
			ay = (t0 + u1) - u2;
			by = (t0 + w1) - w2;
			ax = in;
			bx = in;
			/* (an index is looked at inside the FM window only: in front of it -- the sync pulse and the porch, where a changed
			 * sample behind the line before still shows -- the line's outputs are zero whatever it is) */
			const int x = q * 8 + j;
			differ = differ || (x >= a.C.sl && x < fm_end && med3i(round_away_nb(ay), dmin32, dmax32) != med3i(round_away_nb(by), dmin32, dmax32));
		}
		if(differ) return(false);
		if(__double_as_longlong(ay) == __double_as_longlong(by)) return(true);
	}
	return(false);
}

/* (task_of() from a record already read) */
__device__ __forceinline__ task_view task_from(const hvk_secam_args_t &a, const int t, const hvk_secam_task_t q)
{
	task_view v;
	const int i = t / a.ntasks, slot = t - i * a.ntasks;
	const int64_t findex = a.first_frame + i;
	v.frame = i;
	v.fnum = findex + 1;
	v.valid = (q.flags & HVK_SECAM_TASK_VALID) != 0;
	if(slot < 2)
	{
		v.valid = findex == 0;
		v.line = 0;
		v.prev_line = 0;
		v.sr = a.burst_left + a.burst_width;
		v.clear = false;
		v.fid = false;
	}
	else
	{
		v.line = q.line;
		v.prev_line = q.prev_line;
		v.sr = q.sr;
		v.clear = (q.flags & HVK_SECAM_TASK_CLEAR) != 0;
		v.fid = (q.flags & HVK_SECAM_TASK_FID) != 0;
	}
	const int n = (int) v.fnum * a.lines + v.line;
	v.dr = n & 1;
	v.phase_pos = n % 3 == 0;
	return(v);
}

__global__ __launch_bounds__(64)
void hvk_k_secam_redo(const hvk_secam_args_t a)
{
	const int r0 = blockIdx.x * 64 + threadIdx.x;
	if(r0 >= a.nruns || !a.flags[r0] || (r0 > 0 && a.flags[r0 - 1])) return;

	hvk_secam_state_t S = r0 ? a.exit[r0 - 1] : *a.carry;
	if(a.mid != NULL && a.R == 1)
	{
		/* one line a run: the stretch with the next line's records on their way while a line is worked on */
		redo_pre_t p = redo_pre(a, r0);
		for(int r = r0; r < a.nruns; r++)
		{
			if(r > r0)
			{
				if(same_state(S, p.entry)) break;
				if(!p.flag && p.flag_next) break;
			}
			const bool iir_same = __double_as_longlong(p.entry.ix) == __double_as_longlong(S.ix) && __double_as_longlong(p.entry.iy) == __double_as_longlong(S.iy);
			const redo_pre_t cur = p;
#ifdef HVK_REDO_TRACE
			printf("redo: stretch from run %d, run %d, %s (ix %d iy %d, tail %d %d -> %d %d)\n", r0, r, iir_same ? "IIR pair equal" : "IIR pair differs", (int) (__double_as_longlong(p.entry.ix) == __double_as_longlong(S.ix)), (int) (__double_as_longlong(p.entry.iy) == __double_as_longlong(S.iy)), (int) p.entry.tail[0], (int) p.entry.tail[1], (int) S.tail[0], (int) S.tail[1]);
#endif
			if(r + 1 < a.nruns) p = redo_pre(a, r + 1);
			a.entry[r] = S;
			if(a.seed) a.seed[seed_row(a, r)] = S;
			const task_view v = task_from(a, r, cur.q);
			if(v.valid)
			{
				hvk_secam_state_t was = cur.entry;
				if(v.clear) for(int i = 0; i < 8; i++) { S.tail[i] = 0; was.tail[i] = 0; }
				if(iir_same || redo_converges(a, cur.cbase, a.C.dmin[v.dr], a.C.dmax[v.dr], v.sr < a.C.W ? v.sr : a.C.W, was, S))
				{
					bool tails_same = true;
					for(int i = 0; i < HVK_SECAM_TAIL; i++) tails_same = tails_same && was.tail[i] == S.tail[i];
#ifdef HVK_REDO_TRACE
					printf("redo:    run %d: %s\n", r, tails_same ? "nothing of the line changes" : "last chunk");
#endif
					if(tails_same) S = a.exit[r];       /* (the line from where the two agree on is the line as it was walked) */
					else redo_tail(a, cur, v, S, out_of(a, v));
				}
				else
				{
#ifdef HVK_REDO_TRACE
					printf("redo:    run %d: whole line\n", r);
#endif
					run_task(a, r, S, true);
				}
			}
			a.exit[r] = S;
		}
		return;
	}
	for(int r = r0; r < a.nruns; r++)
	{
		const int t0 = r * a.R, t1 = t0 + a.R < a.total ? t0 + a.R : a.total;
		if(r > r0)
		{
			if(same_state(S, a.entry[r])) break;                                  /* from here on everything stands */
			if(!a.flags[r] && r + 1 < a.nruns && a.flags[r + 1]) break;           /* its exit state is another lane's start */
		}
		a.entry[r] = S;
		for(int m = t0; m < t1; m++)
		{
			if(a.seed) a.seed[seed_row(a, m)] = S;      /* (the hint the chain kernel left here came from the wrong start) */
			run_task(a, m, S, true);
		}
		a.exit[r] = S;
	}
}

/* The same for one line per lane (a.R = 1), one lane per FIELD of the batch: the field's failed runs in order, each walked
 * on from the exit state of the run before for as long as the state left is not the one the next run had started from --
 * a stretch of any length in ONE launch (where a picture's colours make the values behind the lines carry a difference on
 * from line to line instead of forgetting it, a wrong start is wrong to the field's end: the stretch-by-stretch rounds
 * above would hand such a batch to the host's chain). A run that is consistent with the run before it again ends a stretch;
 * the flags behind it still hold (their runs' predecessors were not touched). What a field's first run takes from the
 * field before may be on its way to change in this very launch: the next check finds that. */
__global__ __launch_bounds__(64)
void hvk_k_secam_redo_fields(const hvk_secam_args_t a)
{
	const int lane = blockIdx.x * 64 + threadIdx.x;
	const int i = lane >> 1, h = lane & 1;
	if(i >= a.nframes) return;
	const int sc = a.half_slot[(int) ((a.first_frame + i + 1) & 1)];
	const int r_lo = i * a.ntasks + (h ? sc : 0), r_hi = i * a.ntasks + (h ? a.ntasks : sc);

	int r = r_lo;
	while(r < r_hi)
	{
		if(!a.flags[r]) { r++; continue; }
		hvk_secam_state_t S = r ? a.exit[r - 1] : *a.carry;
		do
		{
			a.entry[r] = S;
			if(a.seed) a.seed[seed_row(a, r)] = S;
			run_task(a, r, S, true);
			a.exit[r] = S;
			r++;
		}
		while(r < r_hi && !same_state(S, a.entry[r]));
		r++;        /* (that run stands as it is, and so do the flags behind it) */
	}
}

/* the batch is through: its last exit state is the next batch's start */
__global__ void hvk_k_secam_carry(const hvk_secam_args_t a)
{
	if(threadIdx.x == 0 && blockIdx.x == 0) *a.carry = a.exit[a.nruns - 1];
}

extern "C" int hvk_launch_secam_check_walk(const hvk_secam_args_t *a, int *differ, hipStream_t stream)
{
	int lo = a->C.dmin[0] < a->C.dmin[1] ? a->C.dmin[0] : a->C.dmin[1];
	int hi = a->C.dmax[0] > a->C.dmax[1] ? a->C.dmax[0] : a->C.dmax[1];
	if(!a->phc || !a->bellz || lo < a->bell_c0 || hi >= a->bell_c0 + 32 * a->bell_blocks) return(HVK_ERROR);
	if(hipMemsetAsync(differ, 0, sizeof(int), stream) != hipSuccess) return(HVK_ERROR);
	hipLaunchKernelGGL(hvk_k_secam_check_walk, dim3((hi - lo + 256) / 256), dim3(256), 0, stream, *a, lo, hi, differ);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_secam_cells_chain(const hvk_secam_args_t *a, int estimate, int walk, hipStream_t stream)
{
	const int lanes = (a->C.W + SPL - 1) / SPL;
	const int threads = (lanes + 63) & ~63;
	if(threads > 256 || (a->C.W % 16) != 0) return(HVK_UNSUPPORTED);
	if(a->ncells > 0)
	{
		const int gx = ((a->ntasks + 63) & ~63) / CELLS_TASKS;
		const size_t lds_b = (size_t) ((a->C.W + 32 + 7) & ~7) * 2 * CELLS_TASKS;
		/* (with the pictures' (U, V) plane the levels are there; the few rows it does not hold whole -- half lines -- go through the table) */
		if(a->levels_computed && !a->uvp) hipLaunchKernelGGL(hvk_k_secam_cells<1>, dim3(gx, a->ncells), dim3(threads * CELLS_TASKS), lds_b, stream, *a);
		else hipLaunchKernelGGL(hvk_k_secam_cells<0>, dim3(gx, a->ncells), dim3(threads * CELLS_TASKS), lds_b, stream, *a);
	}
	if(estimate && a->est)
	{
		const int lanes_e = (a->total + a->ES - 1) / a->ES;
		hipLaunchKernelGGL(hvk_k_secam_est, dim3((lanes_e + 63) / 64), dim3(64), 0, stream, *a);
	}
	if(walk && a->R == 1 && (a->C.W % 8) == 0)
	{
		const int gx = (a->total + WALK_THREADS - 1) / WALK_THREADS;
		if(walk == 2 && a->phc && a->bellz) hipLaunchKernelGGL(hvk_k_secam_walk<1>, dim3(gx), dim3(WALK_THREADS), (size_t) (WALK_PHC + a->bell_blocks) * 16, stream, *a);
		else hipLaunchKernelGGL(hvk_k_secam_walk<0>, dim3(gx), dim3(WALK_THREADS), 0, stream, *a);
	}
	else hipLaunchKernelGGL(hvk_k_secam_chain, dim3((a->nruns + 63) / 64), dim3(64), 0, stream, *a);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_secam_check(const hvk_secam_args_t *a, hipStream_t stream)
{
	if(hipMemsetAsync(a->count, 0, sizeof(int), stream) != hipSuccess) return(HVK_ERROR);
	hipLaunchKernelGGL(hvk_k_secam_check, dim3((a->nruns + 255) / 256), dim3(256), 0, stream, *a);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

/* round: 1 for the first redo of a stage -- a lane per stretch of failed runs: the few lines an estimate got wrong, side by
 * side --, more for the ones after it: what is still wrong then is a stretch that runs on, a lane per field */
extern "C" int hvk_launch_secam_redo(const hvk_secam_args_t *a, int round, hipStream_t stream)
{
	if(round > 1 && a->R == 1 && a->half_slot[0] > 0 && a->half_slot[1] > 0) hipLaunchKernelGGL(hvk_k_secam_redo_fields, dim3((2 * a->nframes + 63) / 64), dim3(64), 0, stream, *a);
	else hipLaunchKernelGGL(hvk_k_secam_redo, dim3((a->nruns + 63) / 64), dim3(64), 0, stream, *a);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_secam_carry(const hvk_secam_args_t *a, hipStream_t stream)
{
	hipLaunchKernelGGL(hvk_k_secam_carry, dim3(1), dim3(64), 0, stream, *a);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}
