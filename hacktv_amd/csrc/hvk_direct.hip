/* hvk_direct.hip -- the plain configurations' render in ONE kernel, from picture planes.
 *
 * What the reference does per scanline in _vid_next_line_raster (src/video.c:2864-3066) falls into two
 * parts. Everything up to the modulator depends on the PICTURE and the line's place in the frame only:
 * sync pulses, the RGB -> level look-ups, the chroma low pass with its over-read, the burst. The
 * modulator -- multiplying (V, U) by the sub-carrier phasor of the sample's position in the STREAM --
 * and all that follows (video filter, sound carriers, NICAM) depends on where in the stream the frame
 * stands. The engine therefore does the first part once per uploaded picture:
 *
 *   hvk_k_prep     one workgroup per scanline of a picture (the raster stages of hvk_device.h):
 *                  L plane  int16 [lines][width]  the line without its sub-carrier
 *                  C plane  int32 [lines][width]  (V, U) after the low pass, burst written over it
 *                  (parity independent: a frame's parity only decides the phasor's sign and whether a
 *                  line carries chroma at all -- hvk_linedesc_t.pal)
 *
 * and the second part for every frame rendered:
 *
 *   hvk_k_direct   two waves per 1024 output samples, four such tiles per workgroup, 8 samples per
 *                  lane. The raster sample at stream position p of line l, sample x is
 *                      L[l][x] + ((i * V * pal + q * U) >> 15)   modulo 2^16, (i, q) = colour_lookup[coff(l) + x]
 *                  (src/video.c:3032-3040): three vector loads and 8 x (v_dot2c_i32_i16, shift, pack,
 *                  add) per lane. The samples go to LDS as the two byte planes of the int8 matrix
 *                  unit -- ONE pair of planes for the workgroup's 4096 + 64 window positions, so that
 *                  a tile's filter reach into its neighbour is the neighbour's own work --, then the
 *                  51-tap filter (mfma_filter), sound carriers, NICAM and the 32-byte stores exactly
 *                  as in hvk_k_filter. The raster slab in HBM and its 2 + 2 B/sample do not exist.
 *
 * A lane's 8 window positions start 26 samples before a multiple of 8 and may straddle the end of a
 * line: L and C are read where the first position's line has them (rows follow each other in the
 * planes), then the lane reads once more with the next line's parameters -- another picture at a
 * frame's end, the other sign of the V switch, the colour table position after a wrap -- and keeps
 * the samples from the boundary on. A window touches at most three lines (width >= 544).
 *
 * Exactness: the planes hold what raster_compute() computes (the same device code, PREP = 1), the
 * arithmetic above is raster_compute()'s modulator; tests/ compare with the raster + filter kernel
 * pair (HVK_DIRECT=0) and with the reference's digests.
 */
#include "hvk_device.h"
#include <stddef.h>
#include <stdlib.h>

#ifndef DG
#define DG    4                     /* tiles per workgroup (2 and 8 measured: 3 % and 6 % slower) */
#endif
#define DLEAD 26                    /* window position 0 is this many samples before the tile's first output (_mfma_taps) */

/* ------------------------------------------------------------------ */

/* A picture's descriptor as the raster stages want it, from the slot it lies in (hvk_prepgeo_t: src/video.c:4887-4897's
 * centring included) */
__device__ __forceinline__ hvk_framedesc_t prep_fdesc(const hvk_kconst_t &k, const hvk_prepgeo_t &g, const int pic)
{
	hvk_framedesc_t f;
	const int slot = g.slot0 + pic;
	f.frame_index = 0;
	f.fb_offset = (int64_t) slot * g.frame_px;
	f.fb_width = g.fb_valid ? g.fb_width : 0;
	f.fb_height = g.fb_valid ? g.fb_height : 0;
	f.pixel_stride = 1;
	f.line_stride = g.fb_width;
	f.vframe_x = (k.active_width - f.fb_width) / 2;
	f.vframe_y = (k.active_lines - f.fb_height) / 2;
	f.fb_interlaced = g.fb_interlaced;
	f.fb_valid = g.fb_valid;
	f.clut_off0 = 0;
	f.parity = 0;
	f.plane_row0 = slot * k.lines;
	f.chroma_row = 0;
	return(f);
}

template<int NT, int WC, int LV, int SECAM>
__global__ __launch_bounds__(1024)
void hvk_k_prep(const hvk_kconst_t k, const hvk_packed_taps_t ctaps, const hvk_packed_taps_t notch, const hvk_rptrs_t P,
                int16_t *__restrict__ Lp, int *__restrict__ Cp, const hvk_prepgeo_t geo)
{
	extern __shared__ __attribute__((aligned(16))) int16_t lds[];

	if((int) blockIdx.x >= k.lines) return;
	const int W = WC ? WC : k.width;
	const int t = threadIdx.x;
	if(WC) __builtin_assume(t * SPL + SPL <= WC);
	const int nth = blockDim.x;
	const int x0 = t * SPL;
	const int rel = (int) blockIdx.x;
	const int pic = (int) blockIdx.y;

	const hvk_framedesc_t f = prep_fdesc(k, geo, pic);
	hvk_linedesc_t d = P.desc[__builtin_amdgcn_readfirstlane(rel)];
	{
		/* chroma wherever a frame of either parity has it: the parity decides at render time */
		const hvk_linedesc_t d1 = P.desc[__builtin_amdgcn_readfirstlane(k.lines + rel)];
		d.pal = (int16_t) ((d.pal | d1.pal) ? 1 : 0);
	}
	const hvk_line_t L = raster_setup_core<SECAM, 0>(k, P, f, d, pic, rel, rel, true, false);

	const int YL = raster_YL(W), CL = raster_CL(W);
	int16_t *Yb = lds, *U = lds + YL, *V = lds + YL + CL;

	uint32_t rgb[HVK_PIX_PASSES];
	hvk_side_t sd;
	int c[SPL];
	raster_loads<NT, WC, HVK_PIX_PASSES, 1>(k, P, L, t, nth, rgb, sd, c);

	if(L.pal)
	{
		raster_clear(L, t, nth, U, CL);
		__syncthreads();
	}
	if(SECAM)
	{
		/* SECAM: the pixels' colour-difference levels go to the C plane as they are -- what the colour chain's cells are
		 * made of (hvk_k_secam_cells), so that a pixel's levels are worked out once */
		short4v cl[HVK_PIX_PASSES];
		raster_gather<LV>(k, P, L, rgb, cl);
		raster_stage<NT, WC, HVK_PIX_PASSES, 0>(k, L, t, nth, cl, sd.ghost_u, sd.ghost_v, Yb, U, V);
		if(Cp && L.has_pix)
		{
			int *row = Cp + ((size_t) f.plane_row0 + rel) * W;
#pragma unroll
			for(int i = 0; i < HVK_PIX_PASSES; i++)
			{
				const int x = L.ax0 + t + i * nth;
				if(x < L.ax1) row[x] = ((int) cl[i].y & 0xFFFF) | ((int) cl[i].z << 16);
			}
		}
	}
	else raster_pixels<NT, WC, LV>(k, P, L, t, nth, rgb, sd.ghost_u, sd.ghost_v, Yb, U, V);
	if(L.pal || L.has_pix) __syncthreads();

	int s[SPL], cq[SPL];
	raster_compute<NT, SECAM, 0, 0, WC, 1>(k, P, L, ctaps, notch, pic, rel, t, nth, lds, sd, c, s, cq);

	const size_t at = ((size_t) f.plane_row0 + rel) * W + x0;
	if(x0 + SPL <= W)
	{
		int4v o;
		o.x = (s[0] & 0xFFFF) | (s[1] << 16);
		o.y = (s[2] & 0xFFFF) | (s[3] << 16);
		o.z = (s[4] & 0xFFFF) | (s[5] << 16);
		o.w = (s[6] & 0xFFFF) | (s[7] << 16);
		*(int4a2 *) (Lp + at) = (int4a2) { o.x, o.y, o.z, o.w };
		if(NT > 1)
		{
			((int4u *) (Cp + at))[0] = (int4u) { c[0], c[1], c[2], c[3] };
			((int4u *) (Cp + at))[1] = (int4u) { c[4], c[5], c[6], c[7] };
		}
	}
	else
	{
		for(int i = 0; i < SPL; i++)
		{
			if(x0 + i >= W) break;
			Lp[at + i] = (int16_t) s[i];
			if(NT > 1) Cp[at + i] = c[i];
		}
	}
}

/* ------------------------------------------------------------------ */

/* The same planes for the PAL / NTSC / monochrome modes, 8 PIXELS PER LANE: a lane loads the 8 pixels under its own 8
 * samples (two 16-byte loads: the pool's pictures are dense), turns them into levels and keeps the luma in registers;
 * only the two chroma channels pass through LDS, written as one 16-byte vector per lane and channel -- zeros and the
 * reference's over-read samples included, so there is no clearing pass -- and read back as the low pass's
 * windows (index j <-> sample j - HVK_CHROMA_LEAD: writes are aligned, a window is read from the aligned chunk in front of
 * it). One barrier per line. What it computes is raster_compute<..., PREP = 1>'s, stage by stage (hvk_device.h); the parity
 * tests run every plain configuration through it and through the raster + filter kernel pair.
 * src/video.c:2961-3029 (luma, chroma, low pass, burst). */
/* SC: SECAM -- no chroma channels (NT = 1); the picture lines' luma through the 51-tap notch (src/video.c:3206), and the
 * pixels' colour-difference levels as they are into the C plane, for the colour chain's cells (hvk_k_secam_cells) */
template<int NT, int WC, int LV, int SC = 0>
__global__ __launch_bounds__(1024)
void hvk_k_prep8(const hvk_kconst_t k, const hvk_packed_taps_t ctaps, const hvk_packed_taps_t notch, const hvk_rptrs_t P,
                 int16_t *__restrict__ Lp, int *__restrict__ Cp, const hvk_prepgeo_t geo)
{
	extern __shared__ __attribute__((aligned(16))) int16_t lds[];
	constexpr int H = NT / 2;
	constexpr int LEAD = HVK_CHROMA_LEAD;
	constexpr int BACK = H <= 8 ? 8 : 16;       /* a window is read from this many samples in front of the lane's first */
	static_assert(H <= 16 && LEAD >= 16, "chroma window");

	if((int) blockIdx.x >= k.lines) return;
	const int W = WC ? WC : k.width;
	const int t = threadIdx.x;
	if(WC) __builtin_assume(t * SPL + SPL <= WC);
	const int x0 = t * SPL;
	const int rel = (int) blockIdx.x;
	const int pic = (int) blockIdx.y;

	/* computed levels: the 256 gamma values into LDS (a pixel reads three of them, every lane another) */
	__shared__ double s_glut[LV ? 256 : 1];
	if(LV)
	{
		for(int i = t; i < 256; i += (int) blockDim.x) s_glut[i] = P.yuvp->glut[i];
	}

	const hvk_framedesc_t f = prep_fdesc(k, geo, pic);
	hvk_linedesc_t d = P.desc[__builtin_amdgcn_readfirstlane(rel)];
	{
		/* chroma wherever a frame of either parity has it: the parity decides at render time */
		const hvk_linedesc_t d1 = P.desc[__builtin_amdgcn_readfirstlane(k.lines + rel)];
		d.pal = (int16_t) ((d.pal | d1.pal) ? 1 : 0);
	}
	const hvk_line_t L = raster_setup_core<0, 0>(k, P, f, d, pic, rel, rel, true, false);
	const bool pal = NT > 1 && L.pal != 0;
	if(LV) __syncthreads();

	const int CL = raster_CL(W);
	int16_t *U = lds, *V = lds + CL;

	/* ---- loads: the lane's 8 pixels (the pool's first ones where it has none: never used), the line's base, burst
	 * window, over-read samples ---- */
	/* which of the lane's 8 samples show a pixel: [plo, phi); which are assigned luma (or black): [alo, ahi). The 16-bit
	 * element masks of such runs are tabulated behind the over-read samples (hvk_engine.cpp), [lo * 9 + hi] */
	const int plo = med3i(L.ax0 - x0, 0, SPL), phi = med3i(L.ax1 - x0, 0, SPL);
	const int alo = L.active ? med3i(L.d.al - x0, 0, SPL) : 0, ahi = L.active ? med3i(L.ar_eff - x0, 0, SPL) : 0;
	const bool lane_pix = phi > plo;            /* (no picture on the line: ax0 = ax1 = 0) */
	const uint32_t *row = lane_pix ? P.pool + L.row_off + x0 : P.pool;
	const int4u pa = ((const int4u *) row)[0], pb = ((const int4u *) row)[1];
	const int4v *runs = (const int4v *) ((const char *) P.ghost + HVK_RUNMASK_OFFSET);
	const int4v mkp = runs[phi > plo ? plo * 9 + phi : 0], mka = runs[ahi > alo ? alo * 9 + ahi : 0];
	hvk_side_t sd;
	int c[SPL];
	raster_load_side<NT, WC, 0, 1>(k, P, L, t, sd, c);         /* (c[]: zeros -- a line without chroma) */

	/* ---- levels, as pairs of int16: luma, and the two chroma channels where the line has a picture ---- */
	int yp[SPL / 2], up[SPL / 2], vp[SPL / 2];
	if(L.has_pix)
	{
		unsigned px[SPL] = { (unsigned) pa.x, (unsigned) pa.y, (unsigned) pa.z, (unsigned) pa.w, (unsigned) pb.x, (unsigned) pb.y, (unsigned) pb.z, (unsigned) pb.w };
		int2v lv[SPL];
		/* the pixels are first needed HERE (the table's addresses are not prepared, and the loads not waited for, where they were issued) */
#pragma unroll
		for(int i = 0; i < SPL; i++) asm volatile("" : "+v"(px[i]));
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			if(LV) lv[i] = __builtin_bit_cast(int2v, level_from<SC ? 1 : 0, LV ? LV - 1 : 0>(s_glut[(px[i] >> 16) & 0xFF], s_glut[(px[i] >> 8) & 0xFF], s_glut[px[i] & 0xFF], *P.yuvp));
			else lv[i] = ((const int2v *) P.yuv)[px[i] & 0xFFFFFFu];
			/* (computed levels four pixels at a time: twelve doubles in flight instead of twenty-four, a wave more per SIMD) */
			if(LV && i == SPL / 2 - 1) __builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			/* an entry is { y | u << 16, v }: the low halves, the high halves of two neighbours' */
			yp[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].x, (unsigned) lv[2 * m].x, 0x05040100u);
			up[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].x, (unsigned) lv[2 * m].x, 0x07060302u);
			vp[m] = (int) __builtin_amdgcn_perm((unsigned) lv[2 * m + 1].y, (unsigned) lv[2 * m].y, 0x05040100u);
		}
	}
	else
	{
#pragma unroll
		for(int m = 0; m < SPL / 2; m++) yp[m] = up[m] = vp[m] = 0;
	}
	const int mp[SPL / 2] = { mkp.x, mkp.y, mkp.z, mkp.w }, ma[SPL / 2] = { mka.x, mka.y, mka.z, mka.w };
#pragma unroll
	for(int m = 0; m < SPL / 2; m++) { up[m] &= mp[m]; vp[m] &= mp[m]; }

	/* ---- the chroma channels into LDS ---- */
	if(pal)
	{
		if(x0 < W)
		{
			if(!WC && x0 + SPL > W)
			{
				/* the lane that straddles the line's end: the over-read samples behind it are this vector's too */
#pragma unroll
				for(int i = 0; i < SPL; i++)
				{
					const int g = x0 + i - W;
					if(g >= 0 && g < H)
					{
						const int gu = P.ghost[2 * g + 0], gv = P.ghost[2 * g + 1];
						up[i / 2] = (i & 1) ? ((up[i / 2] & 0xFFFF) | (gu << 16)) : ((up[i / 2] & (int) 0xFFFF0000u) | (gu & 0xFFFF));
						vp[i / 2] = (i & 1) ? ((vp[i / 2] & 0xFFFF) | (gv << 16)) : ((vp[i / 2] & (int) 0xFFFF0000u) | (gv & 0xFFFF));
					}
				}
			}
			*(int4v *) (U + LEAD + x0) = (int4v) { up[0], up[1], up[2], up[3] };
			*(int4v *) (V + LEAD + x0) = (int4v) { vp[0], vp[1], vp[2], vp[3] };
		}
		/* what lies in front of the line: zeros (src/fir.c:357-375: no history) */
		if(t < LEAD / 8) { *(int4v *) (U + t * 8) = (int4v) { 0, 0, 0, 0 }; *(int4v *) (V + t * 8) = (int4v) { 0, 0, 0, 0 }; }
		/* ... and behind it, from the next multiple of 8 on: the over-read samples */
		if(t < H)
		{
			const int x = W + t;
			if(x >= ((W + 7) & ~7)) { U[LEAD + x] = (int16_t) sd.ghost_u; V[LEAD + x] = (int16_t) sd.ghost_v; }
		}
	}
	if(pal) __syncthreads();

	/* ---- the line without its sub-carrier: blanking and sync pulses, luma assigned over them -- the picture where the
	 * frame covers the line, black elsewhere (src/video.c:2961-3009) ---- */
	int sp[SPL / 2];
	{
		const int bs[SPL / 2] = { sd.base.x, sd.base.y, sd.base.z, sd.base.w };
		const int blk = (k.black_y & 0xFFFF) | (k.black_y << 16);
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			const int lum = (yp[m] & mp[m]) | (blk & ~mp[m]);
			sp[m] = (lum & ma[m]) | (bs[m] & ~ma[m]);
		}
	}

	/* ---- (V, U): zero-history low pass (src/fir.c:357-375), burst written over it (src/video.c:3024-3029) ---- */
	if(pal && x0 < W)
	{
		int vu[SPL];
		if(L.has_pix || x0 + SPL + H > W)
		{
			constexpr int NP = (NT + 1) / 2;
			constexpr int E0 = BACK - H;                    /* the window's first element within the chunks read */
			constexpr int ND = E0 / 2 + SPL / 2 + NP + 1;   /* dwords fir8 looks at, from the chunks' first */
			constexpr int NQ = (ND + 3) / 4;
			int du[NQ * 4], dv[NQ * 4], u[SPL], v[SPL];
			const int4v *pu = (const int4v *) (U + LEAD - BACK + x0), *pv = (const int4v *) (V + LEAD - BACK + x0);
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
			const int4v bwv = sd.bwin;
			const int bw[SPL] = { (int) (short) (bwv.x & 0xFFFF), bwv.x >> 16, (int) (short) (bwv.y & 0xFFFF), bwv.y >> 16,
			                      (int) (short) (bwv.z & 0xFFFF), bwv.z >> 16, (int) (short) (bwv.w & 0xFFFF), bwv.w >> 16 };
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int b = x0 + i - k.burst_left;
				if(b >= 0 && b < k.burst_width) vu[i] = (((k.burst_q * bw[i]) >> 15) & 0xFFFF) | (((k.burst_i * bw[i]) >> 15) << 16);
			}
		}
#pragma unroll
		for(int i = 0; i < SPL; i++) c[i] = vu[i];
	}

	const size_t at = ((size_t) f.plane_row0 + rel) * W + x0;

	if(SC && L.active)
	{
		/* the luma notch over the active picture (raster_compute()'s stage: a zero-history FIR whose input starts at
		 * active_left and which looks 25 samples past the picture's right edge) */
		constexpr int NH = 25, NLEAD = 26;
		int16_t *Z = lds;                       /* index j <-> sample x = j - NLEAD */
		if(t < 4) *(int4v *) (Z + t * 8) = (int4v) { 0, 0, 0, 0 };
		{
			int w[SPL / 2];
#pragma unroll
			for(int m = 0; m < SPL / 2; m++)
			{
				const int xa = x0 + 2 * m;
				w[m] = (xa >= k.active_left ? (sp[m] & 0xFFFF) : 0) | (xa + 1 >= k.active_left ? (sp[m] & (int) 0xFFFF0000u) : 0);
			}
			*(int4u *) (Z + NLEAD + x0) = (int4u) { w[0], w[1], w[2], w[3] };
		}
		__syncthreads();
		if(x0 + SPL > k.active_left && x0 < k.active_left + k.active_width)
		{
			constexpr int ND = SPL / 2 + (51 + 1) / 2 + 1;
			int dn[ND], a[SPL];
			const int4v *pz = (const int4v *) (Z + x0);
#pragma unroll
			for(int m = 0; m < (ND + 3) / 4; m++)
			{
				const int4v v = pz[m];
				if(m * 4 + 0 < ND) dn[m * 4 + 0] = v.x;
				if(m * 4 + 1 < ND) dn[m * 4 + 1] = v.y;
				if(m * 4 + 2 < ND) dn[m * 4 + 2] = v.z;
				if(m * 4 + 3 < ND) dn[m * 4 + 3] = v.w;
			}
			fir8<51, NLEAD - NH>(dn, notch.p, a);
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int x = x0 + i;
				if(x >= k.active_left && x < k.active_left + k.active_width)
				{
					const int q = clamp16(a[i] >> 15);
					sp[i / 2] = (i & 1) ? ((sp[i / 2] & 0xFFFF) | (q << 16)) : ((sp[i / 2] & (int) 0xFFFF0000u) | (q & 0xFFFF));
				}
			}
		}
	}
	/* A lane's 8 dwords of the C plane as whole kilobytes per wave instruction: with 8 consecutive samples a lane the two 16-byte
	 * stores of a lane interleave -- each instruction writes half of every 32 bytes of the wave's 2 KB (hvk_k_direct has the
	 * measurement: the pattern of an HBM stream is worth a tenth of a launch and more). The dwords change lanes through LDS,
	 * within their wave: lane l then stores samples 4 l .. 4 l + 3 of the wave's first 256, then of its second 256. */
	const bool wave_inside = (((t | 63) + 1) * SPL) <= W;          /* (the same for the wave: every lane's 8 samples lie on the line) */
	int *const xw = (int *) (lds + 2 * CL) + (t >> 6) * 512;        /* (behind U and V: the launch asks for 2 KB a wave more) */
	auto store_c8 = [&](const int (&q)[SPL])
	{
		int *const row = Cp + (at - x0) + (t >> 6) * 512;
		((int4v *) (xw + (t & 63) * 8))[0] = (int4v) { q[0], q[1], q[2], q[3] };
		((int4v *) (xw + (t & 63) * 8))[1] = (int4v) { q[4], q[5], q[6], q[7] };
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const int4v a4 = ((const int4v *) xw)[t & 63], b4 = ((const int4v *) (xw + 256))[t & 63];
		((int4u *) row)[t & 63] = (int4u) { a4.x, a4.y, a4.z, a4.w };
		((int4u *) (row + 256))[t & 63] = (int4u) { b4.x, b4.y, b4.z, b4.w };
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      /* (the wave's next exchange writes the same words) */
		__builtin_amdgcn_wave_barrier();
	};
	if(SC && Cp && L.has_pix && x0 + SPL <= W)
	{
		int q[SPL];
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			q[2 * m] = (up[m] & 0xFFFF) | (vp[m] << 16);
			q[2 * m + 1] = ((up[m] >> 16) & 0xFFFF) | (vp[m] & (int) 0xFFFF0000u);
		}
		if(wave_inside) store_c8(q);
		else
		{
			((int4u *) (Cp + at))[0] = (int4u) { q[0], q[1], q[2], q[3] };
			((int4u *) (Cp + at))[1] = (int4u) { q[4], q[5], q[6], q[7] };
		}
	}

	if(x0 + SPL <= W)
	{
		*(int4a2 *) (Lp + at) = (int4a2) { sp[0], sp[1], sp[2], sp[3] };
		if(NT > 1)
		{
			if(wave_inside) store_c8(c);
			else
			{
				((int4u *) (Cp + at))[0] = (int4u) { c[0], c[1], c[2], c[3] };
				((int4u *) (Cp + at))[1] = (int4u) { c[4], c[5], c[6], c[7] };
			}
		}
	}
	else
	{
		for(int i = 0; i < SPL; i++)
		{
			if(x0 + i >= W) break;
			Lp[at + i] = (int16_t) ((i & 1) ? (sp[i / 2] >> 16) : sp[i / 2]);
			if(NT > 1) Cp[at + i] = c[i];
		}
	}
}

/* ------------------------------------------------------------------ */

/* A line of the window: where window position w finds its L / C entry and its phasor -- at index lb + w, cb + w */
typedef struct { int lb, cb; } dline_t;

/* What a tile reads of the tables for one of its lines, all of it issued before any of it is used (the frame's number
 * and parity come by arithmetic, so nothing here waits for another load): the line's V switch and its share of the
 * colour table position. The values are the same for the whole wave: through v_readfirstlane into scalar registers,
 * where the rest is scalar arithmetic. */
typedef struct { int line0, par, prev, zero, own; int pal; unsigned off; int ovr; } dline_in_t;

template<int COLOUR, int OVR>
__device__ __forceinline__ dline_in_t direct_line_loads(const hvk_kconst_t &k, const hvk_dptrs_t &D, const int par_own, const bool first, const int rel)
{
	/* (raster_line_index(): the line before the frame is the last line of the frame before, of the other
	 * parity; the lines behind it the first ones of the next frame -- no picture there in any mode, so the
	 * frame's own planes have them) */
	dline_in_t q;
	q.line0 = rel; q.par = par_own; q.prev = 0;
	q.own = rel >= 0 && rel < k.lines;
	if(rel < 0) { q.line0 = k.lines - 1; q.par ^= 1; q.prev = 1; }
	else if(rel >= k.lines) { q.line0 = rel - k.lines < k.lines ? rel - k.lines : k.lines - 1; q.par ^= 1; }
	/* before the stream: the filter history is zero, not blanking (src/video.c:4665-4667 with src/fir.c:289, :579) */
	q.zero = rel < 0 && first;
	q.pal = 0;
	q.off = 0;
	q.ovr = -1;
	if(OVR && q.own) q.ovr = D.ovr_idx[q.line0];        /* a line the optional stages can write to: the frame's own row of it */
	if(COLOUR == 1)
	{
		/* hvk_linedesc_t.pal, as the low half of the descriptor's fourth dword */
		static_assert(offsetof(hvk_linedesc_t, pal) == 12 && sizeof(hvk_linedesc_t) == 16, "hvk_linedesc_t layout");
		q.pal = ((const int *) D.desc)[(q.par * k.lines + q.line0) * 4 + 3];
		/* the sub-carrier table position advances by one line per line, colour or not (raster_setup_core():
		 * (clut_off0 + rel * width) mod clw); the line's share is tabulated, so no division here */
		q.off = D.lineoff[rel + 1 < k.lines + 3 ? rel + 1 : k.lines + 3];
	}
	return(q);
}

template<int COLOUR, int OVR>
__device__ __forceinline__ dline_t direct_line(const hvk_kconst_t &k, const hvk_dptrs_t &D, const dline_in_t &q, const int row0_prev, const int row0_own,
                                               const unsigned clut_off0, const int wstart, const int y, const int crow = 0)
{
	dline_t l;
	const int row0 = q.prev ? row0_prev : row0_own;
	l.lb = (q.zero ? D.zero_row : row0 + q.line0) * k.width - wstart;
	l.cb = 2 * D.creg - wstart;                                     /* no chroma: phasors of zero */
	if(OVR)
	{
		const int oi = (int) (short) (__builtin_amdgcn_readfirstlane(q.ovr) & 0xFFFF);
		if(oi >= 0)
		{
			/* rendered whole by the raster kernel for this frame (VBI data, test signals, their sub-carrier): nothing to add */
			l.lb = (D.ovr_row0 + y * D.ovr_n + oi) * k.width - wstart;
			if(COLOUR == 2) l.cb = D.chroma_zero - wstart;
			return(l);
		}
	}
	if(COLOUR == 2)
	{
		/* SECAM: the sub-carrier is the colour chain's, a slab per frame of the batch; the lines around a frame have none */
		l.cb = (q.own && !q.zero ? crow * (int) k.raster_samples + q.line0 * k.width : D.chroma_zero) - wstart;
	}
	if(COLOUR == 1 && !q.zero)
	{
		const int pal = (int) (short) (__builtin_amdgcn_readfirstlane(q.pal) & 0xFFFF);
		unsigned coff = clut_off0 + (unsigned) __builtin_amdgcn_readfirstlane((int) q.off);
		if(coff >= k.clw) coff -= k.clw;
		if(pal > 0) l.cb = (int) coff - wstart;
		else if(pal < 0) l.cb = D.creg + (int) coff - wstart;       /* PAL V switch: the table with i negated */
	}
	return(l);
}

/* 8 raster samples (int16 pairs) at window position w of a line: what is read, and what is made of it */
typedef struct { int4a2 lv, cv; int4u c0, c1, k0, k1; } dload_t;

template<int COLOUR>
__device__ __forceinline__ dload_t direct_load(const hvk_dptrs_t &D, const int lb, const int cb, const int w)
{
	dload_t q;
	q.lv = *(const int4a2 *) (D.Lp + (lb + w));          /* (2-byte aligned where the width is odd) */
	if(COLOUR == 2) q.cv = *(const int4a2 *) (D.chroma + (cb + w));
	if(COLOUR == 1)
	{
		const int4u *cp = (const int4u *) (D.Cp + (lb + w));
		const int4u *kp = (const int4u *) (D.clut3 + (cb + w));
		q.c0 = cp[0]; q.c1 = cp[1]; q.k0 = kp[0]; q.k1 = kp[1];
	}
	return(q);
}

template<int COLOUR>
__device__ __forceinline__ int4u direct_make(const dload_t &q)
{
	int4u s = { q.lv.x, q.lv.y, q.lv.z, q.lv.w };
	if(COLOUR == 2)
	{
		s.x = pk_add16(s.x, q.cv.x); s.y = pk_add16(s.y, q.cv.y); s.z = pk_add16(s.z, q.cv.z); s.w = pk_add16(s.w, q.cv.w);
	}
	if(COLOUR == 1)
	{
		const int C[SPL] = { q.c0.x, q.c0.y, q.c0.z, q.c0.w, q.c1.x, q.c1.y, q.c1.z, q.c1.w };
		const int K[SPL] = { q.k0.x, q.k0.y, q.k0.z, q.k0.w, q.k1.x, q.k1.y, q.k1.z, q.k1.w };
		int pr[SPL / 2];
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			/* (i * V * pal + q * U) >> 15 as one dot2 of the packed table entry with (V, U); modulo 2^16 at the add */
			const int t0 = dot2z(K[2 * m], C[2 * m]) >> 15, t1 = dot2z(K[2 * m + 1], C[2 * m + 1]) >> 15;
			pr[m] = (int) __builtin_amdgcn_perm((unsigned) t1, (unsigned) t0, 0x05040100u);
		}
		s.x = pk_add16(s.x, pr[0]); s.y = pk_add16(s.y, pr[1]); s.z = pk_add16(s.z, pr[2]); s.w = pk_add16(s.w, pr[3]);
	}
	return(s);
}

/* ... whichever lines they lie in: lA from window position 0, lB from b1, lC from b2. A lane whose 8 positions straddle the
 * end of a line reads twice -- with the parameters of the line its first position lies in and with the next line's (another
 * sign of the V switch, another picture at a frame's end) -- and BOTH reads go out before either is used: one round trip,
 * not two, for the wave that has such a lane (at 1024 samples per line every tile's first wave has one). */
/* A lane's 8 positions, read: qa with the parameters of the line its first position lies in. Where a line ends inside them
 * (split: how many lie before the boundary) the rest wants the next line's parameters. Between two lines of one picture
 * those differ in the sub-carrier only -- the rows follow each other in the planes --, so the second read is the phasors'
 * (or SECAM's sub-carrier samples') alone and goes out WITH the first; at a frame's ends, before the stream and on rows of
 * the optional stages the planes differ too (full): read again when the first read has been used (rare: one round more). */
typedef struct { dload_t qa; int4u kb0, kb1; int split; bool full; int lb2, cb2; } dgroup_t;

template<int COLOUR>
__device__ __forceinline__ void direct_group_load(const hvk_dptrs_t &D, const dline_t lA, const dline_t lB, const dline_t lC,
                                                  const int b1, const int b2, const int w, dgroup_t &G)
{
	const bool inB = w >= b1, inC = w >= b2;
	const int lb = inC ? lC.lb : (inB ? lB.lb : lA.lb);
	G.qa = direct_load<COLOUR>(D, lb, inC ? lC.cb : (inB ? lB.cb : lA.cb), w);
	const int d1 = b1 - w, d2 = b2 - w;
	const bool s1 = d1 > 0 && d1 < SPL, s2 = d2 > 0 && d2 < SPL;
	G.split = s1 ? d1 : (s2 ? d2 : 0);
	G.lb2 = s1 ? lB.lb : lC.lb;
	G.cb2 = s1 ? lB.cb : lC.cb;
	G.full = G.split != 0 && G.lb2 != lb;
	if(G.split != 0 && !G.full)
	{
		if(COLOUR == 1)
		{
			const int4u *kp = (const int4u *) (D.clut3 + (G.cb2 + w));
			G.kb0 = kp[0]; G.kb1 = kp[1];
		}
		if(COLOUR == 2)
		{
			const int4a2 cv = *(const int4a2 *) (D.chroma + (G.cb2 + w));
			G.kb0 = (int4u) { cv.x, cv.y, cv.z, cv.w };
		}
	}
}

template<int COLOUR>
__device__ __forceinline__ int4u direct_group_make(const hvk_dptrs_t &D, const dgroup_t &G, const int w)
{
	int4u s = direct_make<COLOUR>(G.qa);
	if(G.split)
	{
		/* a line ends inside the group: the samples from there on with the next line's parameters */
		dload_t q2 = G.qa;
		if(!G.full)
		{
			if(COLOUR == 1) { q2.k0 = G.kb0; q2.k1 = G.kb1; }
			if(COLOUR == 2) q2.cv = (int4a2) { G.kb0.x, G.kb0.y, G.kb0.z, G.kb0.w };
		}
		else q2 = direct_load<COLOUR>(D, G.lb2, G.cb2, w);
		const int4u r = direct_make<COLOUR>(q2);
		const int sv[4] = { s.x, s.y, s.z, s.w }, rv[4] = { r.x, r.y, r.z, r.w };
		int o[4];
#pragma unroll
		for(int m = 0; m < 4; m++)
		{
			const int keep = G.split - 2 * m;           /* samples of this pair that lie before the boundary */
			const unsigned mask = keep <= 0 ? 0u : (keep == 1 ? 0xFFFFu : 0xFFFFFFFFu);
			o[m] = (int) (((unsigned) sv[m] & mask) | ((unsigned) rv[m] & ~mask));
		}
		s = (int4u) { o[0], o[1], o[2], o[3] };
	}
	return(s);
}

template<int COLOUR>
__device__ __forceinline__ int4u direct_group(const hvk_dptrs_t &D, const dline_t lA, const dline_t lB, const dline_t lC,
                                              const int b1, const int b2, const int w)
{
	dgroup_t G;
	direct_group_load<COLOUR>(D, lA, lB, lC, b1, b2, w, G);
	return(direct_group_make<COLOUR>(D, G, w));
}

/* make PHASES=1 (-DHVK_PHASE_TIMES): where a workgroup's time goes. Its first lane reads the shader clock at the marks below and
 * leaves it in a slot of the workgroup's own (tools/phase_times.py prints the averages per workgroup). Results are unchanged,
 * the kernel a little slower: a measuring build, never the product's. */
#ifdef HVK_PHASE_TIMES
#define HVK_PHASE_WGS 32768
__device__ unsigned long long hvk_phase_buf[HVK_PHASE_WGS * 10];     /* per workgroup: the clock at its start and at the eight marks, a count */
#define PT_START() const unsigned pt_wg = (blockIdx.y * gridDim.x + blockIdx.x) % HVK_PHASE_WGS; if(threadIdx.x == 0) hvk_phase_buf[pt_wg * 10 + 8] = clock64()
#define PT(i) do { __builtin_amdgcn_sched_barrier(0); if(threadIdx.x == 0) hvk_phase_buf[pt_wg * 10 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while(0)
extern "C" int hvk_phase_times(unsigned long long *out, int reset)
{
	/* out[0..7]: cycles between the marks, summed over the workgroups of the LAST launch that wrote; out[15]: how many */
	static unsigned long long host[HVK_PHASE_WGS * 10];
	if(hipMemcpyFromSymbol(host, HIP_SYMBOL(hvk_phase_buf), sizeof(host)) != hipSuccess) return(HVK_ERROR);
	for(int i = 0; i < 16; i++) out[i] = 0;
	for(int w = 0; w < HVK_PHASE_WGS; w++)
	{
		const unsigned long long *q = host + (size_t) w * 10;
		if(!q[8] || !q[7]) continue;
		unsigned long long last = q[8];
		for(int i = 0; i < 8; i++) { out[i] += q[i] - last; last = q[i]; }
		out[15]++;
	}
	if(reset) { static unsigned long long z[HVK_PHASE_WGS * 10]; if(hipMemcpyToSymbol(HIP_SYMBOL(hvk_phase_buf), z, sizeof(z)) != hipSuccess) return(HVK_ERROR); }
	return(HVK_OK);
}
#else
#define PT_START() do { } while(0)
#define PT(i) do { } while(0)
#endif

/* SND = 1: the configuration has FM / AM carriers AND NICAM (the metric's), known when the kernel is compiled: no read hangs
 * under a test at whose merge point it would be waited for */
template<int VF, int COLOUR, int EXACT, int OVR, int SND, int TR>
__global__ __launch_bounds__(HVK_TILE / HVK_SPL * DG, 8)
void hvk_k_direct(const hvk_kconst_t k,
                  /* (hvk_dptrs_t member by member: as __restrict__ kernel arguments the descriptor tables are known not to alias
                   * the output, and what this tile reads of them -- uniform addresses -- comes by scalar loads) */
                  const int16_t *__restrict__ d_Lp, const int *__restrict__ d_Cp, const int *__restrict__ d_clut3,
                  const int d_creg, const int d_zero_row,
                  const hvk_linedesc_t *__restrict__ d_desc, const hvk_framedesc_t *__restrict__ d_fdesc,
                  const uint32_t *__restrict__ d_lineoff, const uint32_t d_inv_w,
                  const int16_t *__restrict__ d_chroma, const int d_chroma_zero,
                  const int16_t *__restrict__ d_ovr_idx, const int d_ovr_row0, const int d_ovr_n,
                  const hvk_tilerec_t *__restrict__ d_tilerec, const int tiles_pad,
                  const int *__restrict__ carriers,      /* [frames][frame_samples] int16 pairs */
                  const int *__restrict__ tilesyms,      /* [frames][tiles][HVK_NICAM_ROW] */
                  const int *__restrict__ nicam_tapd,
                  const int *__restrict__ nicam_cca,
                  const int4v *__restrict__ mfma_a,
                  const int mfma_ci, const int mfma_cq,
                  int *__restrict__ iq,                  /* [frames * out_stride][frame_samples] int16 pairs */
                  const int64_t out_stride,
                  const int tiles,
                  const int64_t first_frame,             /* frame y of the batch is stream frame first_frame + y * frame_stride */
                  const int64_t frame_stride)
{
	constexpr int LEAD = VF ? DLEAD : 0;
	/* FIN (every configuration with the video filter): the sample is FINISHED in the lane the matrix unit leaves it in. The NICAM
	 * contribution, computed per 8 consecutive samples as ever, goes through LDS to that lane (the exchange the filter's outputs
	 * used to make in the other direction; none without NICAM), the carriers are read and the samples stored there, 8 bytes a
	 * lane and 512 contiguous bytes a wave instruction. With 8 consecutive samples a lane the two 16-byte accesses of a lane
	 * interleaved: every instruction touched half of every 32 bytes of a wave's 2 KB (tools/ablate_direct.py: whole kilobytes
	 * per instruction are worth 9 % of the metric's launch, the carriers' reads 7 of them; round 6: 0.194-0.202 -> 0.179-0.185 ms). */
#ifdef HVK_V_OLDFIN
	constexpr bool FIN = false;
#else
	constexpr bool FIN = VF != 0;
#endif
	/* FIN0 (no video filter): the lane's 8 finished samples change lanes through LDS -- within their wave -- so that a wave's
	 * carriers are read and its samples stored as whole kilobytes per instruction here too (lane L: samples 4 L .. 4 L + 3 of
	 * the wave's first 256, then of its second 256) */
#ifdef HVK_V_OLDFIN
	constexpr bool FIN0 = false;
#else
	constexpr bool FIN0 = VF == 0;
#endif
	constexpr int NP = DG * HVK_TILE + 64;      /* window positions of the workgroup: its tiles follow each other in the stream */
	constexpr int TL = HVK_TILE / HVK_SPL;      /* lanes of a tile */
	__shared__ __attribute__((aligned(16))) unsigned char xh[VF ? NP : 16], xl[VF ? NP : 16];
	__shared__ __attribute__((aligned(16))) int outl_g[DG][HVK_TILE];
	__shared__ __attribute__((aligned(16))) int16_t tapd[HVK_NICAM_COPIES * HVK_NICAM_TAPD];
	__shared__ int sym_st_g[DG][HVK_NICAM_SYMS];
	__shared__ __attribute__((aligned(16))) int4v sym_ent_g[1 + DG * HVK_NICAM_SYMS];     /* (an entry of slack in front: nicam_add()) */

	/* the grid's x extent is padded to a multiple of 8: with workgroups dealt round-robin to the 8 XCDs, the same
	 * lines of EVERY frame then run on the same XCD, whose L2 keeps their plane rows (a picture that stays) and
	 * their slices of the colour table (the same again every few frames) */
	const int bx = (int) blockIdx.x;
	const int y = (int) blockIdx.y;
	/* (the workgroups the grid's padding adds leave further down, behind the tile's set-up: an early return here would put
	 * a round trip of its own -- the tile count, a kernel argument -- in front of every other scalar load) */
	PT_START();
	/* The kernel arguments the set-up reads, wanted HERE, in the entry block -- one round of scalar loads. Left alone the
	 * compiler loads an argument in the block that first uses it: four dependent round trips before the tile's first vector
	 * load could go out, an eighth of a workgroup's life (tools/phase_times.py). (Pass-through statements without side
	 * effects: a volatile one would count as a write to memory and turn every scalar table read behind it into a vector read.) */
	unsigned k_clw = k.clw;
	int k_w = k.width, k_fs = k.frame_samples, a_tiles = tiles, a_tiles_pad = tiles_pad, a_creg = d_creg, a_zero = d_zero_row;
	int64_t a_ff = first_frame, a_fst = frame_stride;
	/* (numbers only: a pointer that has been through such a statement is one the compiler knows nothing about any more) */
	asm("" : "+s"(k_clw), "+s"(k_w), "+s"(k_fs), "+s"(a_tiles), "+s"(a_tiles_pad), "+s"(a_creg), "+s"(a_zero), "+s"(a_ff), "+s"(a_fst));

	hvk_dptrs_t D;
	D.Lp = d_Lp; D.Cp = d_Cp; D.clut3 = d_clut3; D.creg = a_creg; D.zero_row = a_zero;
	D.desc = d_desc; D.fdesc = d_fdesc; D.lineoff = d_lineoff; D.inv_w = d_inv_w;
	D.chroma = d_chroma; D.chroma_zero = d_chroma_zero;
	D.ovr_idx = d_ovr_idx; D.ovr_row0 = d_ovr_row0; D.ovr_n = d_ovr_n;

	const int FS = k_fs, W = k_w;
	const int sub = __builtin_amdgcn_readfirstlane((int) threadIdx.x / TL);   /* which of the workgroup's tiles: the same for a wave */
	const int t = threadIdx.x % TL;
	const int x0 = t * SPL;
	const int f0_w = (t >> 6) * 512 + (t & 63) * 4;     /* FIN0: the lane's first sample of the tile in the store's order */
	/* a workgroup that reaches past the frame's last tile still fills its planes (the last tile's filter looks into
	 * them); nothing of such a tile is stored */
	const int tile_raw = bx * DG + sub;
	const bool tile_valid = tile_raw < a_tiles;
	const int tile = tile_valid ? tile_raw : a_tiles - 1;
	const int n0 = tile_raw * HVK_TILE;         /* first output sample of the tile, frame local */
	int *const outl = outl_g[sub];
	int *const sym_st = sym_st_g[sub];
	int4v *const sym_ent = sym_ent_g + 1 + sub * HVK_NICAM_SYMS;
	(void) outl;

	/* the NICAM pulse table, staged once per workgroup (hvk_k_filter has the layout) */
	static_assert(TL * DG >= HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8, "one pulse-table vector per thread");
	const bool has_car = SND ? true : k.has_carriers != 0, has_nic = SND ? true : k.has_nicam != 0;
	const bool tap_mine = has_nic && (int) threadIdx.x < HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8;
	int4v tap_stage = { 0, 0, 0, 0 };
	if(has_nic) tap_stage = ((const int4v *) nicam_tapd)[min((int) threadIdx.x, HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8 - 1)];

	int4v a_hh = { 0, 0, 0, 0 }, a_hl = { 0, 0, 0, 0 };
	if(VF)
	{
		a_hh = mfma_a[t & 63];
		a_hl = mfma_a[64 + (t & 63)];
	}

	/* ---- the lines this tile's window lies in (all scalar) ---- */
	const int64_t frame_index = a_ff + (int64_t) y * a_fst;
	const int par_own = (int) ((frame_index + 1) & 1);
	const bool first = frame_index == 0;
	/* (the frame before: the row its LAST line's planes start at less lines - 1; the frame: the row of its line 0) */
	const int row0_prev = __builtin_amdgcn_readfirstlane(D.fdesc[2 * y].plane_row0);
	const int row0_own = __builtin_amdgcn_readfirstlane(D.fdesc[2 * y + 1].plane_row0);
	const unsigned clut_off0 = (unsigned) __builtin_amdgcn_readfirstlane((int) D.fdesc[2 * y + 1].clut_off0);
	const int crow = COLOUR == 2 ? __builtin_amdgcn_readfirstlane(D.fdesc[2 * y + 1].chroma_row) : 0;     /* SECAM: where the frame's sub-carrier lies */
	int b1, b2;
	dline_t lA, lB, lC;
	if(!TR)
	{
		const int p0 = n0 - LEAD;                               /* stream position (frame local) of window position 0 */
		const int lineA = p0 < 0 ? -1 : (int) __builtin_amdgcn_readfirstlane((int) __umulhi((unsigned) p0, D.inv_w));
		const int xA0 = p0 - lineA * W;
		b1 = W - xA0; b2 = b1 + W;                              /* window positions at which the next two lines begin */
		const dline_in_t qA = direct_line_loads<COLOUR, OVR>(k, D, par_own, first, lineA);
		const dline_in_t qB = direct_line_loads<COLOUR, OVR>(k, D, par_own, first, lineA + 1);
		const dline_in_t qC = direct_line_loads<COLOUR, OVR>(k, D, par_own, first, lineA + 2);
		lA = direct_line<COLOUR, OVR>(k, D, qA, row0_prev, row0_own, clut_off0, -xA0, y, crow);
		lB = direct_line<COLOUR, OVR>(k, D, qB, row0_prev, row0_own, clut_off0, b1, y, crow);
		lC = direct_line<COLOUR, OVR>(k, D, qC, row0_prev, row0_own, clut_off0, b2, y, crow);
	}
	else
	{
		/* Which lines the tile's window lies in, where they begin in it, their V switch and their share of the colour table
		 * position depend on the tile and the frame's parity only: tabulated by the host (hvk_engine.cpp:_tile_records, the
		 * arithmetic of direct_line_loads() / direct_line()), ONE 64-byte scalar load instead of a dozen dependent ones. What
		 * is the frame's own -- where its picture's planes lie, its colour table position -- comes from its descriptor. */
		const hvk_tilerec_t R = d_tilerec[__builtin_amdgcn_readfirstlane(par_own * a_tiles_pad + (tile_raw < a_tiles_pad ? tile_raw : a_tiles_pad - 1))];
		b1 = R.b1; b2 = b1 + W;
		dline_t l3[3];
#pragma unroll
		for(int X = 0; X < 3; X++)
		{
			const int meta = R.meta[X];
			const int line0 = meta & 0xFFFF, prev = (meta >> 16) & 1, own = (meta >> 17) & 1, pal = ((meta >> 18) & 3) - 1;
			const bool zero = prev && first;        /* before the stream: the filter's history is zero, not blanking */
			l3[X].lb = zero ? D.zero_row * W + R.nws[X] : (prev ? row0_prev : row0_own) * W + R.lw[X];
			l3[X].cb = 2 * D.creg + R.nws[X];
			if(OVR && R.ovr[X] >= 0)
			{
				/* rendered whole by the raster kernel for this frame (VBI data, test signals, their sub-carrier): nothing to add (direct_line()) */
				l3[X].lb = (D.ovr_row0 + y * D.ovr_n + R.ovr[X]) * W + R.nws[X];
				if(COLOUR == 2) l3[X].cb = D.chroma_zero + R.nws[X];
				continue;
			}
			if(COLOUR == 2) l3[X].cb = (own && !zero ? crow * (int) k.raster_samples + line0 * W : D.chroma_zero) + R.nws[X];
			if(COLOUR == 1 && !zero && pal != 0)
			{
				unsigned coff = clut_off0 + R.off[X];
				if(coff >= k_clw) coff -= k_clw;
				l3[X].cb = (pal < 0 ? D.creg : 0) + (int) coff + R.nws[X];
			}
		}
		lA = l3[0]; lB = l3[1]; lC = l3[2];
	}

	if(bx * DG >= a_tiles) return;
	PT(0);      /* the tile's lines: scalar loads and arithmetic */
	/* ---- loads ---- */
	int symv = 0, cc_tile = 0;
	if(has_nic)
	{
		const int *row = tilesyms + ((size_t) y * a_tiles + tile) * HVK_NICAM_ROW;
		cc_tile = row[HVK_NICAM_SYMS];
		symv = row[t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1];
	}

	const int n = n0 + x0;                      /* this lane's first output, frame local */
	/* the lane's 8 outputs are inside the frame (EXACT: the frame is a whole number of tiles) */
	const bool whole = tile_valid && (EXACT || n + SPL <= FS);

	dgroup_t G0;
	direct_group_load<COLOUR>(D, lA, lB, lC, b1, b2, x0, G0);
	/* The 64 window positions behind the workgroup's last tile (its filter's reach into the next group's first line): ONE
	 * position per lane of that tile's first wave, read in the SAME round trip as the lane's own eight -- three registers a lane,
	 * each lane with the parameters of the line its position lies in (no boundary inside a lane: nothing to merge). As eight
	 * positions for each of eight lanes they cost that wave a second round trip of its own (twenty registers could not be
	 * held beside the first), and the other seven waves of the workgroup waited at the barrier for it: a quarter of a
	 * workgroup's life (tools/phase_times.py). */
	const bool trail = VF && sub == DG - 1 && t < 64;
	int tr_l = 0, tr_c = 0, tr_k = 0;
	if(trail)
	{
		const int w1 = HVK_TILE + t;
		const bool inB = w1 >= b1, inC = w1 >= b2;
		const int lb = inC ? lC.lb : (inB ? lB.lb : lA.lb), cb = inC ? lC.cb : (inB ? lB.cb : lA.cb);
		tr_l = D.Lp[lb + w1];
		if(COLOUR == 1) { tr_c = D.Cp[lb + w1]; tr_k = D.clut3[cb + w1]; }
		if(COLOUR == 2) tr_c = D.chroma[cb + w1];
	}
	const int4u g0 = direct_group_make<COLOUR>(D, G0, x0);

	int4u car0 = { 0, 0, 0, 0 }, car1 = { 0, 0, 0, 0 };
	int2u cj[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
	if(FIN && has_car)
	{
		/* the carriers of the two samples the lane finishes in each of the filter's four passes (mfma_filter_each) */
		const int fl = t & 63, fg = fl >> 4, fc = fl & 15;
#pragma unroll
		for(int j = 0; j < 4; j++)
		{
			const int nn = n0 + ((t >> 6) * 64 + j * 16 + fc) * 8 + 2 * fg;
			const bool ok2 = tile_valid && (EXACT || nn + 2 <= FS);
			cj[j] = __builtin_nontemporal_load((const int2u *) (carriers + (size_t) y * FS + (ok2 ? nn : 0)));
		}
	}
	else if(FIN0 && has_car)
	{
		const int nA = n0 + f0_w, nB = nA + 256;
		const bool okA = tile_valid && (EXACT || nA + 4 <= FS), okB = tile_valid && (EXACT || nB + 4 <= FS);
		car0 = __builtin_nontemporal_load((const int4u *) (carriers + (size_t) y * FS + (okA ? nA : 0)));
		car1 = __builtin_nontemporal_load((const int4u *) (carriers + (size_t) y * FS + (okB ? nB : 0)));
	}
	else if(!FIN && !FIN0 && has_car && (SND || whole) && !ABLATE(256))      /* (ABLATE: measuring builds only, tools/ablate_direct.py) */
	{
		/* (SND: unconditionally -- a lane outside the frame reads the frame's first run instead, and uses nothing of it) */
		const int4u *c = (const int4u *) (carriers + (size_t) y * FS + (whole ? n : 0));
		/* read once, like the output is written once: marked as streaming so that neither pushes the plane rows and the
		 * colour table's slices, which every frame comes back to, out of the XCD's L2 (+5 % on the metric configuration) */
		if(ABLATE(8192))
		{
			/* (timing only: a wave's two reads as two contiguous kilobytes instead of interleaved halves of every 32 bytes) */
			const int4u *cw = (const int4u *) (carriers + (size_t) y * FS + (n - (t & 63) * SPL));
			car0 = __builtin_nontemporal_load(&cw[t & 63]);
			car1 = __builtin_nontemporal_load(&cw[64 + (t & 63)]);
		}
		else
		{
		car0 = __builtin_nontemporal_load(&c[0]);
		car1 = __builtin_nontemporal_load(&c[1]);
		}
	}

	if(tap_mine) ((int4v *) tapd)[threadIdx.x] = tap_stage;

	int o[SPL];                                 /* packed (I, Q) int16 */

	if(VF)
	{
		int2v ph, pl;
		split_planes(g0, ph, pl);
		((int2v *) (xh + sub * HVK_TILE))[t] = ph;
		((int2v *) (xl + sub * HVK_TILE))[t] = pl;
		if(trail)
		{
			/* the 64 positions behind the workgroup's last tile: direct_make()'s arithmetic on one sample, its two bytes */
			int s1 = tr_l;
			if(COLOUR == 1) s1 += dot2z(tr_k, tr_c) >> 15;
			if(COLOUR == 2) s1 += tr_c;
			xh[DG * HVK_TILE + t] = (unsigned char) ((s1 >> 8) & 0xFF);
			xl[DG * HVK_TILE + t] = (unsigned char) ((s1 & 0xFF) ^ 0x80);
		}
	}
	else
	{
		/* no filter: the raster goes straight to I, Q = 0 */
		o[0] = g0.x & 0xFFFF; o[1] = (int) ((unsigned) g0.x >> 16); o[2] = g0.y & 0xFFFF; o[3] = (int) ((unsigned) g0.y >> 16);
		o[4] = g0.z & 0xFFFF; o[5] = (int) ((unsigned) g0.z >> 16); o[6] = g0.w & 0xFFFF; o[7] = (int) ((unsigned) g0.w >> 16);
	}

	if(has_nic && t < HVK_NICAM_SYMS) nicam_symbol_slot(symv, n0, sym_st, sym_ent, t, tapd);
	PT(1);      /* the reads' round trip, the modulator, the byte planes into LDS */
	__syncthreads();
	PT(2);      /* the first barrier */

	/* the mixer row (i, -q) of this lane's samples: on its way while the filter and the pulse sums run */
	int4u mix[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
	if(has_nic)
	{
		int cp = cc_tile + x0;                  /* mixer position of this lane's first sample */
		if(k.nicam_cc_len >= HVK_TILE) { if(cp >= k.nicam_cc_len) cp -= k.nicam_cc_len; }
		else cp %= k.nicam_cc_len;
		nicam_mix_rows(nicam_cca, k.nicam_cc_len + 8, cp, mix);
	}

	if(FIN)
	{
		/* NICAM on its own (the adds are modulo 2^16 per channel: their order is free), handed to the lanes that finish the samples */
		if(has_nic)
		{
			int nic[SPL] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			nicam_add(k, x0, sym_st, sym_ent, tapd, mix, nic);
			((int4v *) (outl + x0))[0] = (int4v) { nic[0], nic[1], nic[2], nic[3] };
			((int4v *) (outl + x0))[1] = (int4v) { nic[4], nic[5], nic[6], nic[7] };
		}
		PT(3);
		/* (the exchange is within a wave -- mfma_filter_each(): wave t >> 6 finishes segments 64 (t >> 6) .. + 63, its own lanes'
		 * samples -- and a wave's LDS operations are carried out in the order it issues them: a fence for the compiler, no barrier) */
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		PT(4);
		int *const frame_out = iq + (size_t) y * out_stride * FS;
		mfma_filter_each(xh + sub * HVK_TILE, xl + sub * HVK_TILE, t, a_hh, a_hl, mfma_ci, mfma_cq,
		                 [&](const int j, const int seg, const int g, const int2v pk)
		{
			int2v nv = { 0, 0 };
			if(has_nic) nv = *(const int2v *) (outl + seg * 8 + 2 * g);
			const int nn = n0 + seg * 8 + 2 * g;
			int2u ov;
			ov.x = pk_add16(pk_add16(pk.x, cj[j].x), nv.x);
			ov.y = pk_add16(pk_add16(pk.y, cj[j].y), nv.y);
			if(tile_valid && (EXACT || nn + 2 <= FS)) __builtin_nontemporal_store(ov, (int2u *) (frame_out + nn));
			else if(tile_valid && nn < FS)
			{
				/* (a frame of an odd number of samples: its last one) */
				frame_out[nn] = pk_add16(pk_add16(pk.x, has_car ? carriers[(size_t) y * FS + nn] : 0), nv.x);
			}
		});
		PT(5); PT(6); PT(7);
		return;
	}

	if(VF)
	{
		if(!ABLATE(2048)) mfma_filter(xh + sub * HVK_TILE, xl + sub * HVK_TILE, outl, t, a_hh, a_hl, mfma_ci, mfma_cq);
		PT(3);  /* the filter on the matrix unit */
		__syncthreads();
		PT(4);  /* the second barrier */
		const int4v oa = ((const int4v *) (outl + x0))[0], ob = ((const int4v *) (outl + x0))[1];
		o[0] = oa.x; o[1] = oa.y; o[2] = oa.z; o[3] = oa.w;
		o[4] = ob.x; o[5] = ob.y; o[6] = ob.z; o[7] = ob.w;
	}

	if(FIN0)
	{
		if(has_nic) nicam_add(k, x0, sym_st, sym_ent, tapd, mix, o);
		((int4v *) (outl + x0))[0] = (int4v) { o[0], o[1], o[2], o[3] };
		((int4v *) (outl + x0))[1] = (int4v) { o[4], o[5], o[6], o[7] };
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		int *const frame_out = iq + (size_t) y * out_stride * FS;
#pragma unroll
		for(int h = 0; h < 2; h++)
		{
			const int nn = n0 + f0_w + h * 256;
			const int4v v = *(const int4v *) (outl + f0_w + h * 256);
			const int4u c4 = h ? car1 : car0;
			if(tile_valid && (EXACT || nn + 4 <= FS))
			{
				int4u ov = { v.x, v.y, v.z, v.w };
				if(has_car) ov = (int4u) { pk_add16(v.x, c4.x), pk_add16(v.y, c4.y), pk_add16(v.z, c4.z), pk_add16(v.w, c4.w) };
				__builtin_nontemporal_store(ov, (int4u *) (frame_out + nn));
			}
			else if(tile_valid)
			{
				const int vv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for(int i = 0; i < 4; i++) if(nn + i < FS) frame_out[nn + i] = has_car ? pk_add16(vv[i], carriers[(size_t) y * FS + nn + i]) : vv[i];
			}
		}
		return;
	}

	/* serial carriers (FM / AM sound), computed on the host: a plain add of int16 pairs with wrap-around
	 * (src/video.c:3431-3432) */
	if(has_car)
	{
		if(whole)
		{
			o[0] = pk_add16(o[0], car0.x); o[1] = pk_add16(o[1], car0.y); o[2] = pk_add16(o[2], car0.z); o[3] = pk_add16(o[3], car0.w);
			o[4] = pk_add16(o[4], car1.x); o[5] = pk_add16(o[5], car1.y); o[6] = pk_add16(o[6], car1.z); o[7] = pk_add16(o[7], car1.w);
		}
		else if(tile_valid)
		{
			const int *c = carriers + (size_t) y * FS + n;
#pragma unroll
			for(int i = 0; i < SPL; i++) if(n + i < FS) o[i] = pk_add16(o[i], c[i]);
		}
	}

	PT(5);      /* the filter's outputs back from LDS, the carriers added */
	if(has_nic && !ABLATE(1024)) nicam_add(k, x0, sym_st, sym_ent, tapd, mix, o);
	PT(6);      /* NICAM */

	/* interleaved int16 I/Q, 32 bytes per lane */
	int *dst = iq + (size_t) y * out_stride * FS + n;
	if(ABLATE(512) && o[0] != 0x12345) return;
	if(ABLATE(4096) && whole)
	{
		/* (timing only: a wave's two stores as two contiguous kilobytes) */
		int4u *dw = (int4u *) (dst - (t & 63) * SPL);
		__builtin_nontemporal_store(((int4u) { o[0], o[1], o[2], o[3] }), &dw[t & 63]);
		__builtin_nontemporal_store(((int4u) { o[4], o[5], o[6], o[7] }), &dw[64 + (t & 63)]);
		return;
	}
	if(whole)
	{
		__builtin_nontemporal_store(((int4u) { o[0], o[1], o[2], o[3] }), &((int4u *) dst)[0]);
		__builtin_nontemporal_store(((int4u) { o[4], o[5], o[6], o[7] }), &((int4u *) dst)[1]);
	}
	else if(tile_valid)
	{
#pragma unroll
		for(int i = 0; i < SPL; i++) if(n + i < FS) dst[i] = o[i];
	}
	PT(7);      /* the stores issued */
}

/* ------------------------------------------------------------------ */
/* launchers                                                           */

extern "C" void hvk_raster_ptrs(const hvk_raster_args_t *a, hvk_rptrs_t *P);

template<int NT>
static int _launch_prep(const hvk_raster_args_t *a, const hvk_prepgeo_t *g, int npics, int16_t *Lp, int *Cp, hipStream_t stream)
{
	const int W = a->k.width;
	int threads = (W + SPL - 1) / SPL;
	threads = (threads + 63) / 64 * 64;
	/* (behind the staging buffers: 2 KB a wave for hvk_k_prep8's C plane exchange) */
	const size_t lds = ((size_t) ((W + 8 + 7) & ~7) + 2 * (size_t) ((W + 2 * HVK_CHROMA_LEAD + 7) & ~7)) * sizeof(int16_t) + 64 + (size_t) (threads / 64) * 2048;
	hvk_rptrs_t P;
	hvk_raster_ptrs(a, &P);
	const dim3 grid((a->k.lines + 7) & ~7, npics), block(threads);
#define PREP(WCV, LVV, SC) hipLaunchKernelGGL((hvk_k_prep<NT, WCV, LVV, SC>), grid, block, lds, stream, a->k, a->ctaps, a->notch, P, Lp, Cp, *g)
#define PREP8(WCV, LVV) hipLaunchKernelGGL((hvk_k_prep8<NT, WCV, LVV>), grid, block, lds, stream, a->k, a->ctaps, a->notch, P, Lp, Cp, *g)
#define PREP8S(LVV) hipLaunchKernelGGL((hvk_k_prep8<NT, 0, LVV, (NT == 1 ? 1 : 0)>), grid, block, lds, stream, a->k, a->ctaps, a->notch, P, Lp, Cp, *g)
	/* (HVK_PREP=1: the one-pixel-per-lane-and-pass kernel built from the raster's stages, kept as the second opinion) */
	static const int old_prep = getenv("HVK_PREP") ? atoi(getenv("HVK_PREP")) : 0;
	if(NT == 1 && a->k.secam && (old_prep == 1 || a->k.width % SPL != 0)) { if(a->levels_computed) PREP(0, 1, (NT == 1 ? 1 : 0)); else PREP(0, 0, (NT == 1 ? 1 : 0)); }
	else if(NT == 1 && a->k.secam) { switch(a->levels_computed) { case 0: PREP8S(0); break; case 2: PREP8S(2); break; case 3: PREP8S(3); break; default: PREP8S(1); } }
	else if(old_prep == 1)
	{
		if(NT == 13 && W == 1024) { if(a->levels_computed) PREP((NT == 13 ? 1024 : 0), 1, 0); else PREP((NT == 13 ? 1024 : 0), 0, 0); }
		else { if(a->levels_computed) PREP(0, 1, 0); else PREP(0, 0, 0); }
	}
	else if(NT == 13 && W == 1024)
	{
		switch(a->levels_computed) { case 0: PREP8((NT == 13 ? 1024 : 0), 0); break; case 2: PREP8((NT == 13 ? 1024 : 0), 2); break; case 3: PREP8((NT == 13 ? 1024 : 0), 3); break; default: PREP8((NT == 13 ? 1024 : 0), 1); }
	}
	else { switch(a->levels_computed) { case 0: PREP8(0, 0); break; case 2: PREP8(0, 2); break; case 3: PREP8(0, 3); break; default: PREP8(0, 1); } }
#undef PREP8S
#undef PREP8
#undef PREP
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_prep(const hvk_raster_args_t *a, const hvk_prepgeo_t *g, int npics, int16_t *Lp, int *Cp, hipStream_t stream)
{
	if(npics < 1) return(HVK_OK);
	switch(a->k.colour && !a->k.secam ? a->k.chroma_ntaps : 1)
	{
	case 1:  return(_launch_prep<1>(a, g, npics, Lp, Cp, stream));
	case 3:  return(_launch_prep<3>(a, g, npics, Lp, Cp, stream));    /* (no chroma low pass: fir8<3>) */
	case 5:  return(_launch_prep<5>(a, g, npics, Lp, Cp, stream));
	case 7:  return(_launch_prep<7>(a, g, npics, Lp, Cp, stream));
	case 9:  return(_launch_prep<9>(a, g, npics, Lp, Cp, stream));
	case 11: return(_launch_prep<11>(a, g, npics, Lp, Cp, stream));
	case 13: return(_launch_prep<13>(a, g, npics, Lp, Cp, stream));
	case 15: return(_launch_prep<15>(a, g, npics, Lp, Cp, stream));
	case 17: return(_launch_prep<17>(a, g, npics, Lp, Cp, stream));
	case 19: return(_launch_prep<19>(a, g, npics, Lp, Cp, stream));
	case 21: return(_launch_prep<21>(a, g, npics, Lp, Cp, stream));
	case 23: return(_launch_prep<23>(a, g, npics, Lp, Cp, stream));
	case 25: return(_launch_prep<25>(a, g, npics, Lp, Cp, stream));
	case 27: return(_launch_prep<27>(a, g, npics, Lp, Cp, stream));
	case 29: return(_launch_prep<29>(a, g, npics, Lp, Cp, stream));
	case 31: return(_launch_prep<31>(a, g, npics, Lp, Cp, stream));
	case 33: return(_launch_prep<33>(a, g, npics, Lp, Cp, stream));
	}
	return(HVK_UNSUPPORTED);
}

/* Which configurations render this way: the plain ones -- PAL / NTSC / SECAM / monochrome at the sample rate, one
 * picture per frame, no inserters -- with the matrix-unit filter or none */
extern "C" int hvk_direct_supported(const hvk_kconst_t *k, const void *mfma_a, int secam_fid, int max_frames)
{
	if(k->s_video || k->rawbb || k->rs_L || k->sis || k->fields != 1 || k->fm_video || k->fsc_mode || k->spill_lines) return(0);      /* (field-sequential colour: what a picture's line looks like depends on the frame's number) */        /* (VBI data lines and test signals: rows of their own per frame, hvk_dptrs_t.ovr_idx) */
	/* SECAM: the sub-carrier comes from the colour chain's slab, indexed with 32 bits */
	(void) secam_fid;       /* (the identification lines: rows of their own per frame, like the VBI data lines) */
	if(k->secam && (int64_t) (max_frames + 1) * k->raster_samples >= 0x7FFFFFFF) return(0);
	if(k->vf_type != 0 && !(k->vf_ntaps == 51 && mfma_a && (k->vf_type == 1 || k->vf_type == 3))) return(0);
	if(k->width < 544) return(0);               /* a tile's window within three lines */
	return(1);
}

template<int VF, int COLOUR>
static int _launch_direct2(const hvk_direct_args_t *a, hipStream_t stream)
{
	const int tiles = (a->k.frame_samples + HVK_TILE - 1) / HVK_TILE;
	const dim3 grid(((tiles + DG - 1) / DG + 7) & ~7, a->nframes), block(HVK_TILE / SPL * DG);
#define DIRECT4(EX, OV, SN, TRV) hipLaunchKernelGGL((hvk_k_direct<VF, COLOUR, EX, OV, SN, TRV>), grid, block, 0, stream, a->k, \
	a->D.Lp, a->D.Cp, a->D.clut3, a->D.creg, a->D.zero_row, a->D.desc, a->D.fdesc, a->D.lineoff, a->D.inv_w, a->D.chroma, a->D.chroma_zero, a->D.ovr_idx, a->D.ovr_row0, a->D.ovr_n, \
	(const hvk_tilerec_t *) a->tilerec, a->tiles_pad, (const int *) a->carriers, a->tilesyms, \
	a->nicam_tapd, a->nicam_cca, (const int4v *) a->mfma_a, a->mfma_ci, a->mfma_cq, (int *) a->iq, a->out_stride, tiles, a->first_frame, a->frame_stride)
/* (no tile records -- HVK_TILEREC=0 -- : the lines of a tile's window worked out by every wave, the second opinion) */
#define DIRECT3(EX, OV, SN) do { if(a->tilerec) DIRECT4(EX, OV, SN, 1); else DIRECT4(EX, OV, SN, 0); } while(0)
#define DIRECT2(EX, OV) do { if(VF && OV == 0 && a->k.has_carriers && a->k.has_nicam) DIRECT3(EX, OV, (VF && OV == 0 ? 1 : 0)); else DIRECT3(EX, OV, 0); } while(0)
#define DIRECT(EX) do { if(a->D.ovr_idx) DIRECT2(EX, 1); else DIRECT2(EX, 0); } while(0)
	if(a->k.frame_samples % HVK_TILE == 0) DIRECT(1); else DIRECT(0);
#undef DIRECT
#undef DIRECT2
#undef DIRECT3
#undef DIRECT4
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_direct(const hvk_direct_args_t *a, hipStream_t stream)
{
	const int vf = a->k.vf_type ? 1 : 0;
	if(a->k.secam) return(vf ? _launch_direct2<1, 2>(a, stream) : _launch_direct2<0, 2>(a, stream));
	if(a->k.colour) return(vf ? _launch_direct2<1, 1>(a, stream) : _launch_direct2<0, 1>(a, stream));
	return(vf ? _launch_direct2<1, 0>(a, stream) : _launch_direct2<0, 0>(a, stream));
}
