/* hvk_fused.hip -- pictures that change on every frame: the whole scanline pipeline in ONE kernel, from the pixels.
 *
 * hvk_direct.hip splits the reference's per-line work (src/video.c:2864-3066) into a per-picture part (hvk_k_prep: levels,
 * chroma low pass, burst -> picture planes in HBM) and a per-frame part (hvk_k_direct: sub-carrier, video filter, sound).
 * For a picture that stays that is the point: the first part is done once. For a source that shows a new picture on every
 * frame -- every real one -- the planes are written once and read once: 6 + 6 bytes per sample of HBM traffic beside the
 * 3 bytes of pixels and the 4 bytes of output that have to move, and two kernels that cannot overlap.
 *
 * hvk_k_fused does both parts for lines of 1024 samples (the 625-line systems at 16 MHz: the metric configuration's
 * geometry), where a filter tile IS a scanline:
 *
 *   workgroup = 4 consecutive lines of a frame = 4 tiles, 8 main waves (two per line, 8 samples per lane) + 1 halo wave
 *   main wave   pixels -> levels (table or computed) -> luma in registers, U / V through LDS -> 13-tap low pass, burst ->
 *               (V, U) x sub-carrier phasor -> the raster's 8 samples -> the int8 byte planes of the matrix-unit filter
 *               in LDS; then exactly hvk_k_direct's second half: 51-tap filter (mfma_filter), sound carriers, NICAM,
 *               32-byte streaming stores
 *   halo wave   the filter of the group's first outputs reaches 25 samples back into the line BEFORE the group: that
 *               line's last 28 samples are made again here (5 lanes' worth of the same per-line routine) instead of
 *               being fetched from a neighbour; the 36 samples behind the group (the next line's sync edge: no picture,
 *               no chroma) are a table read done by the last line's first lanes
 *
 * The planes never exist; HBM sees the pixels, the phasor table (L2), the carriers and the output. What a lane computes is,
 * stage by stage, what hvk_k_prep8 and hvk_k_direct compute (same device functions, same order of operations); the
 * parity tests render random pictures this way, through the planes and through the raster + filter pair.
 *
 * Window convention: window position 0 of the group is 28 samples before its first output (hvk_k_direct: 26), so that a
 * lane's 8 bytes land 4-byte aligned in the planes; the A operand is laid out for it (hvk_engine.cpp:_mfma_taps, lead 28).
 */
#include "hvk_device.h"
#include <stddef.h>
#include <stdlib.h>

#define FG     4                    /* lines (= tiles) per workgroup */
#define FLEAD  28                   /* window position 0 is this many samples before the group's first output */
#define FW     1024                 /* samples per line */
#define FTL    (FW / SPL)           /* lanes of a line */
#define FHALO0 960                  /* the halo wave's first sample of the line before the group */

/* One scanline's 8 samples per lane, from the pixels to the modulated raster (packed int16 pairs).
 *   part 1 (before the barrier): loads, levels, the chroma channels into LDS (U, V: index j <-> sample j - HVK_CHROMA_LEAD + xbase)
 *   part 2 (behind it): luma over the base line, low pass, burst, modulator
 * xbase: the sample LDS index HVK_CHROMA_LEAD stands for (0 for a whole line; the halo wave stages the line's tail only). */
typedef struct {
	int yp[SPL / 2];            /* luma pairs */
	int4v mkp, mka;             /* run masks: samples that show a pixel, samples that are assigned luma */
	hvk_side_t sd;
	bool lane_ok;               /* the lane's samples lie on the line */
} fline_t;

template<int NT, int LV>
__device__ __forceinline__ void fused_part1(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L, const bool pal,
                                            const int t, const int x0, const int xbase, int16_t *U, int16_t *V, fline_t &F)
{
	constexpr int H = NT / 2;
	constexpr int LEAD = HVK_CHROMA_LEAD;

	const int plo = med3i(L.ax0 - x0, 0, SPL), phi = med3i(L.ax1 - x0, 0, SPL);
	const int alo = L.active ? med3i(L.d.al - x0, 0, SPL) : 0, ahi = L.active ? med3i(L.ar_eff - x0, 0, SPL) : 0;
	const bool lane_pix = phi > plo;
	const uint32_t *row = lane_pix ? P.pool + L.row_off + x0 : P.pool;
	const int4u pa = ((const int4u *) row)[0], pb = ((const int4u *) row)[1];
	const int4v *runs = (const int4v *) ((const char *) P.ghost + HVK_RUNMASK_OFFSET);
	F.mkp = runs[phi > plo ? plo * 9 + phi : 0];
	F.mka = runs[ahi > alo ? alo * 9 + ahi : 0];
	int c[SPL];
	raster_load_side<NT, FW, 0, 1>(k, P, L, x0 / SPL, F.sd, c);
	/* (the over-read samples by the lane's number in its wave pair: the halo wave's lanes stand at the line's end) */
	F.sd.ghost_u = P.ghost[2 * (t < H ? t : H - 1) + 0];
	F.sd.ghost_v = P.ghost[2 * (t < H ? t : H - 1) + 1];

	int up[SPL / 2], vp[SPL / 2];
	if(L.has_pix)
	{
		unsigned px[SPL] = { (unsigned) pa.x, (unsigned) pa.y, (unsigned) pa.z, (unsigned) pa.w, (unsigned) pb.x, (unsigned) pb.y, (unsigned) pb.z, (unsigned) pb.w };
		int2v lv[SPL];
#pragma unroll
		for(int i = 0; i < SPL; i++) asm volatile("" : "+v"(px[i]));
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			if(LV) lv[i] = __builtin_bit_cast(int2v, level_of<0, LV == 3 ? 2 : 0>(px[i] & 0xFFFFFFu, *P.yuvp));
			else lv[i] = ((const int2v *) P.yuv)[px[i] & 0xFFFFFFu];
		}
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			F.yp[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].x, (unsigned) lv[2 * m].x, 0x05040100u);
			up[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].x, (unsigned) lv[2 * m].x, 0x07060302u);
			vp[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].y, (unsigned) lv[2 * m].y, 0x05040100u);
		}
	}
	else
	{
#pragma unroll
		for(int m = 0; m < SPL / 2; m++) F.yp[m] = up[m] = vp[m] = 0;
	}

	if(pal && F.lane_ok)
	{
		const int mp[SPL / 2] = { F.mkp.x, F.mkp.y, F.mkp.z, F.mkp.w };
		*(int4v *) (U + LEAD + x0 - xbase) = (int4v) { up[0] & mp[0], up[1] & mp[1], up[2] & mp[2], up[3] & mp[3] };
		*(int4v *) (V + LEAD + x0 - xbase) = (int4v) { vp[0] & mp[0], vp[1] & mp[1], vp[2] & mp[2], vp[3] & mp[3] };
		/* in front of the line: zeros (src/fir.c:357-375: no history); behind it: the reference's over-read samples */
		if(xbase == 0 && t < LEAD / 8) { *(int4v *) (U + t * 8) = (int4v) { 0, 0, 0, 0 }; *(int4v *) (V + t * 8) = (int4v) { 0, 0, 0, 0 }; }
		if(t < H) { U[LEAD + FW - xbase + t] = (int16_t) F.sd.ghost_u; V[LEAD + FW - xbase + t] = (int16_t) F.sd.ghost_v; }
	}
}

template<int NT>
__device__ __forceinline__ int4u fused_part2(const hvk_kconst_t &k, const hvk_line_t &L, const bool pal, const hvk_packed_taps_t &ctaps,
                                             const int *__restrict__ clut3, const int cb,
                                             const int x0, const int xbase, const int16_t *U, const int16_t *V, const fline_t &F)
{
	/* the sub-carrier's phasors of the lane's samples (a line without chroma: the table's copy of zeros): asked for here, behind
	 * the barrier -- they come from L2 while the low pass runs, and eight registers fewer live across the wait */
	const int4u k0 = ((const int4u *) (clut3 + cb + x0))[0], k1 = ((const int4u *) (clut3 + cb + x0))[1];
	constexpr int H = NT / 2;
	constexpr int LEAD = HVK_CHROMA_LEAD;
	constexpr int BACK = H <= 8 ? 8 : 16;

	/* the line without its sub-carrier: blanking and sync pulses, luma assigned over them (src/video.c:2961-3009) */
	int sp[SPL / 2];
	{
		const int bs[SPL / 2] = { F.sd.base.x, F.sd.base.y, F.sd.base.z, F.sd.base.w };
		const int mp[SPL / 2] = { F.mkp.x, F.mkp.y, F.mkp.z, F.mkp.w }, ma[SPL / 2] = { F.mka.x, F.mka.y, F.mka.z, F.mka.w };
		const int blk = (k.black_y & 0xFFFF) | (k.black_y << 16);
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			const int lum = (F.yp[m] & mp[m]) | (blk & ~mp[m]);
			sp[m] = (lum & ma[m]) | (bs[m] & ~ma[m]);
		}
	}
	int4u s = { sp[0], sp[1], sp[2], sp[3] };
	if(!pal) return(s);

	/* (V, U): zero-history low pass (src/fir.c:357-375), burst written over it (src/video.c:3024-3029) */
	int vu[SPL];
	if(L.has_pix || x0 + SPL + H > FW)
	{
		constexpr int NP = (NT + 1) / 2;
		constexpr int E0 = BACK - H;
		constexpr int ND = E0 / 2 + SPL / 2 + NP + 1;
		constexpr int NQ = (ND + 3) / 4;
		int du[NQ * 4], dv[NQ * 4], u[SPL], v[SPL];
		const int4v *pu = (const int4v *) (U + LEAD - BACK + x0 - xbase), *pv = (const int4v *) (V + LEAD - BACK + x0 - xbase);
#pragma unroll
		for(int m = 0; m < NQ; m++)
		{
			const int4v a = pu[m], b = pv[m];
			du[m * 4 + 0] = a.x; du[m * 4 + 1] = a.y; du[m * 4 + 2] = a.z; du[m * 4 + 3] = a.w;
			dv[m * 4 + 0] = b.x; dv[m * 4 + 1] = b.y; dv[m * 4 + 2] = b.z; dv[m * 4 + 3] = b.w;
		}
		fir8<NT, E0 & 1>(du + E0 / 2, ctaps.p, u);
		fir8<NT, E0 & 1>(dv + E0 / 2, ctaps.p, v);
#pragma unroll
		for(int i = 0; i < SPL; i++) vu[i] = sat_pack16(v[i] >> 15, u[i] >> 15);
	}
	else
	{
#pragma unroll
		for(int i = 0; i < SPL; i++) vu[i] = 0;
	}
	if(x0 + SPL > k.burst_left && x0 < k.burst_left + k.burst_width)
	{
		const int4v bwv = F.sd.bwin;
		const int bw[SPL] = { (int) (short) (bwv.x & 0xFFFF), bwv.x >> 16, (int) (short) (bwv.y & 0xFFFF), bwv.y >> 16,
		                      (int) (short) (bwv.z & 0xFFFF), bwv.z >> 16, (int) (short) (bwv.w & 0xFFFF), bwv.w >> 16 };
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			const int b = x0 + i - k.burst_left;
			if(b >= 0 && b < k.burst_width) vu[i] = (((k.burst_q * bw[i]) >> 15) & 0xFFFF) | (((k.burst_i * bw[i]) >> 15) << 16);
		}
	}

	/* onto the sub-carrier (src/video.c:3032-3040): L + ((i V pal + q U) >> 15) modulo 2^16 -- hvk_k_direct's direct_eval() */
	const int K[SPL] = { k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w };
	int pr[SPL / 2];
#pragma unroll
	for(int m = 0; m < SPL / 2; m++)
	{
		const int t0 = dot2z(K[2 * m], vu[2 * m]) >> 15, t1 = dot2z(K[2 * m + 1], vu[2 * m + 1]) >> 15;
		pr[m] = (int) __builtin_amdgcn_perm((unsigned) t1, (unsigned) t0, 0x05040100u);
	}
	s.x = pk_add16(s.x, pr[0]); s.y = pk_add16(s.y, pr[1]); s.z = pk_add16(s.z, pr[2]); s.w = pk_add16(s.w, pr[3]);
	return(s);
}

/* which line of which picture a wave works on (all scalar): line `rel` of frame y of the batch -- -1: the last line of the
 * frame before, >= lines: a line of the frame behind (no picture there in any mode this kernel takes) */
typedef struct { hvk_line_t L; bool pal; int cb; bool zero; } fsel_t;

__device__ __forceinline__ fsel_t fused_select(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_framedesc_t *__restrict__ fdesc,
                                               const uint32_t *__restrict__ lineoff, const int creg, const int y, const int rel,
                                               const int64_t frame_index)
{
	fsel_t q;
	int line0 = rel, par = (int) ((frame_index + 1) & 1);
	bool own = true;
	if(rel < 0) { line0 = k.lines - 1; par ^= 1; own = false; }
	else if(rel >= k.lines) { line0 = rel - k.lines < k.lines ? rel - k.lines : k.lines - 1; par ^= 1; own = false; }
	q.zero = rel < 0 && frame_index == 0;       /* before the stream: the filter's history is zero, not blanking */
	hvk_framedesc_t f = fdesc[__builtin_amdgcn_readfirstlane(2 * y + (rel < 0 ? 0 : 1))];
	if(rel >= k.lines) f.fb_valid = 0;          /* (the frame behind: its first lines show no picture) */
	const hvk_linedesc_t d = P.desc[__builtin_amdgcn_readfirstlane(par * k.lines + line0)];
	q.L = raster_setup_core<0, 0>(k, P, f, d, y, rel, line0, own || rel < 0, false);
	q.pal = k.colour && d.pal != 0 && !q.zero;
	/* the colour table position advances by one line per line, colour or not; the V switch picks the copy with i negated */
	unsigned coff = fdesc[__builtin_amdgcn_readfirstlane(2 * y + 1)].clut_off0 + lineoff[rel + 1 < k.lines + 3 ? rel + 1 : k.lines + 3];
	if(coff >= k.clw) coff -= k.clw;
	q.cb = !q.pal ? 2 * creg : (d.pal > 0 ? (int) coff : creg + (int) coff);
	return(q);
}

#ifndef FUSED_WAVES_LV3
#define FUSED_WAVES_LV3 5            /* ... the kernel with the short form of the level arithmetic (LV = 3) */
#endif
#ifndef FUSED_WAVES
#define FUSED_WAVES 7                /* waves per SIMD the table-levels kernel is compiled for: three workgroups of nine waves per compute unit */
#endif
template<int NT, int LV>
__global__ __launch_bounds__(FTL * FG + 64, LV == 1 ? 4 : (LV == 3 ? FUSED_WAVES_LV3 : FUSED_WAVES))
void hvk_k_fused(const hvk_kconst_t k, const hvk_packed_taps_t ctaps, const hvk_rptrs_t P,
                 const int *__restrict__ d_clut3, const int d_creg,
                 const hvk_framedesc_t *__restrict__ d_fdesc, const uint32_t *__restrict__ d_lineoff,
                 const int *__restrict__ carriers, const int *__restrict__ tilesyms,
                 const int *__restrict__ nicam_tapd, const int *__restrict__ nicam_cca,
                 const int4v *__restrict__ mfma_a, const int mfma_ci, const int mfma_cq,
                 int *__restrict__ iq, const int64_t out_stride, const int tiles,
                 const int64_t first_frame, const int64_t frame_stride)
{
	constexpr int NP = FG * FW + 80;            /* window positions of the group (+ slack for the last lanes' 8-byte writes) */
	constexpr int CL = FW + 2 * HVK_CHROMA_LEAD;
	__shared__ __attribute__((aligned(16))) unsigned char xh[NP], xl[NP];
	/* the chroma channels of the four lines while they are low-passed; the filter's outputs afterwards (never both) */
	__shared__ __attribute__((aligned(16))) int stage_g[FG][FW + 2 * HVK_CHROMA_LEAD];
	__shared__ __attribute__((aligned(16))) int16_t halo_uv[2][128];
	__shared__ __attribute__((aligned(16))) int16_t tapd[HVK_NICAM_COPIES * HVK_NICAM_TAPD];
	__shared__ int sym_st_g[FG][HVK_NICAM_SYMS];
	__shared__ __attribute__((aligned(16))) int4v sym_ent_g[1 + FG * HVK_NICAM_SYMS];      /* (an entry of slack in front: nicam_add()) */
	static_assert(sizeof(int) * (FW + 2 * HVK_CHROMA_LEAD) >= 2 * sizeof(int16_t) * CL, "U and V of a line fit its filter-output row");

	const int bx = (int) blockIdx.x;
	const int y = (int) blockIdx.y;
	if(bx * FG >= tiles) return;

	const int FS = k.frame_samples;
	const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
	const bool halo = wave == 2 * FG;           /* the ninth wave */
	const int sub = halo ? 0 : wave >> 1;       /* which of the group's lines */
	const int t = halo ? (int) threadIdx.x - 2 * FG * 64 : (int) threadIdx.x % FTL;
	const int rel0 = bx * FG;                   /* the group's first line = tile */
	const int tile_raw = rel0 + sub;
	const bool tile_valid = !halo && tile_raw < tiles;
	const int tile = tile_raw < tiles ? tile_raw : tiles - 1;
	const int n0 = tile_raw * FW;
	const int64_t frame_index = first_frame + (int64_t) y * frame_stride;

	/* ---- which line this wave makes ---- */
	const int rel = halo ? rel0 - 1 : tile_raw;
	const int x0 = halo ? FHALO0 + t * SPL : t * SPL;
	const int xbase = halo ? FHALO0 - 16 : 0;
	int16_t *U = halo ? halo_uv[0] : (int16_t *) stage_g[sub], *V = halo ? halo_uv[1] : (int16_t *) stage_g[sub] + CL;
	const fsel_t q = fused_select(k, P, d_fdesc, d_lineoff, d_creg, y, rel, frame_index);

	/* ---- the second half's loads (hvk_k_direct): NICAM pulse table and symbols, the filter's A operand, carriers ---- */
	const bool tap_mine = k.has_nicam && (int) threadIdx.x < HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8;
	int4v tap_stage = { 0, 0, 0, 0 };
	if(k.has_nicam) tap_stage = ((const int4v *) nicam_tapd)[min((int) threadIdx.x, HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8 - 1)];
	int symv = 0, cc_tile = 0;
	if(k.has_nicam && !halo)
	{
		const int *row = tilesyms + ((size_t) y * tiles + tile) * HVK_NICAM_ROW;
		cc_tile = row[HVK_NICAM_SYMS];
		symv = row[t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1];
	}
	(void) x0;

	/* ---- the line: part 1 ---- */
	fline_t F;
	F.lane_ok = halo ? t < (FW - FHALO0) / SPL : true;
	fused_part1<NT, LV>(k, P, q.L, q.pal, t, F.lane_ok ? x0 : FHALO0, xbase, U, V, F);
	if(tap_mine) ((int4v *) tapd)[threadIdx.x] = tap_stage;
	if(k.has_nicam && !halo && t < HVK_NICAM_SYMS) nicam_symbol_slot(symv, n0, sym_st_g[sub], sym_ent_g + 1 + sub * HVK_NICAM_SYMS, t, tapd);
	__syncthreads();

	/* (asked for behind the first barrier: the line's own reads are through, these have all of part 2 and the filter to arrive in,
	 * and sixteen registers fewer are live while the levels are made) */
	const int4v a_hh = mfma_a[t & 63], a_hl = mfma_a[64 + (t & 63)];
	/* the carriers of the two samples the lane FINISHES in each of the filter's four passes (hvk_k_direct's FIN: samples are
	 * finished in the lane the matrix unit leaves them in, 8 bytes a lane and 512 contiguous bytes a wave instruction) */
	int2u cj[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
	if(k.has_carriers && tile_valid)
	{
		const int fl = t & 63, fg = fl >> 4, fc = fl & 15;
#pragma unroll
		for(int j = 0; j < 4; j++)
		{
			const int nn = n0 + ((t >> 6) * 64 + j * 16 + fc) * 8 + 2 * fg;
			cj[j] = __builtin_nontemporal_load((const int2u *) (carriers + (size_t) y * FS + nn));
		}
	}

	/* ---- part 2, and into the byte planes: window position of sample x of the group's line j is j * 1024 + x + FLEAD ---- */
	{
		int4u g0 = fused_part2<NT>(k, q.L, q.pal, ctaps, d_clut3, q.cb, F.lane_ok ? x0 : FHALO0, xbase, U, V, F);
		if(q.zero) g0 = (int4u) { 0, 0, 0, 0 };
		int2v ph, pl;
		split_planes(g0, ph, pl);
		if(!halo)
		{
			const int w = sub * FW + x0 + FLEAD;
			*(int *) (xh + w) = ph.x; *(int *) (xh + w + 4) = ph.y;
			*(int *) (xl + w) = pl.x; *(int *) (xl + w + 4) = pl.y;
		}
		else if(F.lane_ok)
		{
			/* the line before the group: samples 996 .. 1023 are window positions 0 .. 27 */
			const int w = x0 - (FW - FLEAD);
			if(w >= 0) { *(int *) (xh + w) = ph.x; *(int *) (xl + w) = pl.x; }
			if(w + 4 >= 0) { *(int *) (xh + w + 4) = ph.y; *(int *) (xl + w + 4) = pl.y; }
		}
	}
	if(halo && t >= 16 && t < 16 + 5)
	{
		/* behind the group: the first 40 samples of the next line -- its sync edge: no picture, no chroma (checked by the
		 * host: hvk_fused_supported) -- are its base line's */
		const int tt = t - 16;
		int line0 = rel0 + FG, par = (int) ((frame_index + 1) & 1);
		if(line0 >= k.lines) { line0 -= k.lines; par ^= 1; }
		const hvk_linedesc_t d = P.desc[__builtin_amdgcn_readfirstlane(par * k.lines + line0)];
		const int4v bv = *(const int4v *) (P.linebase + (size_t) (d.secam_fid >> 8) * k.base_stride + tt * SPL);
		int2v ph, pl;
		split_planes((int4u) { bv.x, bv.y, bv.z, bv.w }, ph, pl);
		const int w = FG * FW + tt * SPL + FLEAD;
		*(int *) (xh + w) = ph.x; *(int *) (xh + w + 4) = ph.y;
		*(int *) (xl + w) = pl.x; *(int *) (xl + w + 4) = pl.y;
	}
	__syncthreads();

	/* ---- from here on hvk_k_direct's second half, for the main waves (the halo wave keeps the barriers company) ---- */
	int4u mix[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
	if(k.has_nicam && !halo)
	{
		int cp = cc_tile + x0;
		if(k.nicam_cc_len >= HVK_TILE) { if(cp >= k.nicam_cc_len) cp -= k.nicam_cc_len; }
		else cp %= k.nicam_cc_len;
		nicam_mix_rows(nicam_cca, k.nicam_cc_len + 8, cp, mix);
	}
	int *const outl = stage_g[sub];
	if(halo || !tile_valid) return;         /* (no barrier behind this point: the exchange below is within a wave) */

	/* NICAM on its own (the adds are modulo 2^16 per channel: their order is free), handed through LDS to the lanes that finish the samples */
	if(k.has_nicam)
	{
		int nic[SPL] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		nicam_add(k, x0, sym_st_g[sub], sym_ent_g + 1 + sub * HVK_NICAM_SYMS, tapd, mix, nic);
		((int4v *) (outl + x0))[0] = (int4v) { nic[0], nic[1], nic[2], nic[3] };
		((int4v *) (outl + x0))[1] = (int4v) { nic[4], nic[5], nic[6], nic[7] };
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	int *const frame_out = iq + (size_t) y * out_stride * FS;
	const bool has_nic = k.has_nicam != 0;
	mfma_filter_each(xh + sub * FW, xl + sub * FW, t, a_hh, a_hl, mfma_ci, mfma_cq, [&](const int j, const int seg, const int g, const int2v pk)
	{
		int2v nv = { 0, 0 };
		if(has_nic) nv = *(const int2v *) (outl + seg * 8 + 2 * g);
		int2u ov;
		ov.x = pk_add16(pk_add16(pk.x, cj[j].x), nv.x);
		ov.y = pk_add16(pk_add16(pk.y, cj[j].y), nv.y);
		__builtin_nontemporal_store(ov, (int2u *) (frame_out + n0 + seg * 8 + 2 * g));
	});
}

/* ------------------------------------------------------------------ */

extern "C" void hvk_raster_ptrs(const hvk_raster_args_t *a, hvk_rptrs_t *P);

/* Which configurations render this way: those hvk_k_direct takes, colour by a PAL / NTSC sub-carrier with the 13-tap
 * chroma filter, the video filter on the matrix unit, lines of exactly 1024 samples in whole tiles, and -- the halo
 * shortcuts above -- nothing but sync in the first 40 samples of any line and no burst there */
extern "C" int hvk_fused_supported(const hvk_kconst_t *k, const hvk_linedesc_t *desc)
{
	if(k->width != FW || !k->colour || k->secam || k->chroma_ntaps != 13 || k->vf_type == 0 || k->vf_ntaps != 51) return(0);
	if(k->frame_samples != k->lines * FW || k->frame_samples % HVK_TILE) return(0);
	if(k->burst_left < 40 || k->burst_left + k->burst_width > FHALO0) return(0);
	if(k->active_left < 40) return(0);
	for(int i = 0; i < 2 * k->lines; i++) if(desc[i].ar > desc[i].al && desc[i].al < 40) return(0);
	return(1);
}

extern "C" int hvk_launch_fused(const hvk_raster_args_t *ra, const hvk_direct_args_t *a, const void *mfma_a28, hipStream_t stream)
{
	const int tiles = a->k.frame_samples / HVK_TILE;
	const dim3 grid(((tiles + FG - 1) / FG + 7) & ~7, a->nframes), block(FTL * FG + 64);
	hvk_rptrs_t P;
	hvk_raster_ptrs(ra, &P);
#define FUSED(LVV) hipLaunchKernelGGL((hvk_k_fused<13, LVV>), grid, block, 0, stream, a->k, ra->ctaps, P, a->D.clut3, a->D.creg, a->D.fdesc, a->D.lineoff, \
	(const int *) a->carriers, a->tilesyms, a->nicam_tapd, a->nicam_cca, (const int4v *) mfma_a28, a->mfma_ci, a->mfma_cq, (int *) a->iq, a->out_stride, tiles, a->first_frame, a->frame_stride)
	if(ra->levels_computed >= 3) FUSED(3); else if(ra->levels_computed) FUSED(1); else FUSED(0);
#undef FUSED
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}
