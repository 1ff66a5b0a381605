/* hvk_fused.hip -- the whole per-sample path of a render in ONE kernel: raster, video filter,
 * sound carriers, NICAM, interleaved int16 I/Q out (src/video.c:2864-3066 + src/fir.c:564-615 +
 * src/video.c:3431-3432 + src/nicam728.c:342-411). The raster never touches HBM.
 *
 * A workgroup walks a run of R consecutive lines of one frame. Line r is rasterised (the steps of
 * hvk_device.h) into LDS as two BYTE PLANES -- the form the int8 matrix unit multiplies -- while
 * line r - 2, whose successor's leading samples are by then in place, goes through the filter:
 *
 *   iteration:   loads of both lines (source row, sub-carrier phasors | symbol row, carrier samples)
 *                clear chroma staging                       | NICAM symbol table of line r - 2
 *        -- barrier --
 *                pixels -> levels -> Y / U / V staging      | 51-tap filter of line r - 2 on the
 *                                                           | matrix unit, planes -> output exchange
 *        -- barrier --
 *                8 samples per lane: pulses, luma, chroma   | carriers + NICAM onto the filtered
 *                low pass, burst, QAM -> planes of line r   | samples, 32-byte stores
 *        -- barrier --
 *
 * Three plane buffers rotate. Buffer b holds, for its line, the last 32 samples of the line before,
 * the line, and the first 32 samples of the line after (the filter reaches 25 either way): a lane
 * that holds edge samples writes them into the neighbour's buffer as well. The first line of a run
 * needs the tail of the line before the run and the last one the head of the line after it: those
 * two are rasterised too (R + 2 rasters for R lines out).
 *
 * Plane coordinates: sample x of the buffer's line sits at byte x + 34 -- 2 modulo 8, so that the
 * 64-byte window of an 8-output segment (which starts 26 samples before the segment) starts 8-byte
 * aligned; a lane's 8 bytes therefore go out as 2 + 4 + 2.
 *
 * Without a video filter (VF = 0) there are no planes and no halo lines: a lane's raster samples
 * are its outputs.
 *
 * The filter here is the matrix-unit form only (hvk_kernels.hip explains the byte split); taps it
 * cannot express, S-Video, SECAM, raw baseband input and the resampler keep the two-kernel path.
 */
#include "hvk_device.h"

#define FPAD  34        /* plane byte of a line's sample 0 */
#define FEDGE 32        /* samples mirrored into the neighbouring buffers */

typedef struct {
	const int *carriers;      /* [frames][frame_samples] int16 pairs */
	const int *tilesyms;      /* [frames][lines][HVK_NICAM_ROW]: symbols (start << 3 | valid << 2 | dsym) of a LINE, mixer position */
	const int *nicam_tapd;    /* pulse taps: four shifted int16 copies, zero padded (hvk_engine.cpp) */
	const int *nicam_cca;     /* mixer (i, -q), 8 entries past the wrap */
	const int4v *mfma_a;      /* the taps as A operand, [hh, hl][lane] (hvk_engine.cpp:_mfma_taps) */
	int mfma_ci, mfma_cq;     /* 128 * sum of the taps */
	int *iq;                  /* [frames * out_stride][frame_samples] int16 pairs */
	int64_t out_stride;
	int run_lines;            /* R */
} hvk_fptrs_t;

/* a lane's 8 plane bytes (two dwords) to plane byte j .. j + 7, j = 2 (mod 8) */
__device__ __forceinline__ void plane_put8(unsigned char *p, const int j, const int2v v)
{
	*(uint16_t *) (p + j) = (uint16_t) v.x;
	*(uint32_t *) (p + j + 2) = __builtin_amdgcn_alignbit((unsigned) v.y, (unsigned) v.x, 16);
	*(uint16_t *) (p + j + 6) = (uint16_t) ((unsigned) v.y >> 16);
}

/* bytes lo .. hi - 1 of them, wherever */
__device__ __forceinline__ void plane_put_some(unsigned char *p, const int j, const int2v v, const int lo, const int hi)
{
#pragma unroll
	for(int i = 0; i < 8; i++)
	{
		if(i >= lo && i < hi) p[j + i] = (unsigned char) (((unsigned) (i < 4 ? v.x : v.y)) >> ((i & 3) * 8));
	}
}

template<int NT, int VF, int EXTRAS, int WC, int LV>
__global__ __launch_bounds__(256)
void hvk_k_fused(const hvk_kconst_t k,
                 const hvk_packed_taps_t ctaps,
                 const hvk_rptrs_t P,
                 const hvk_fptrs_t Q,
                 const int64_t first_frame,
                 const int64_t frame_stride)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

	const int W = WC ? WC : k.width;
	const int t = threadIdx.x;
	const int nth = WC ? WC / SPL : (int) blockDim.x;
	const int x0 = t * SPL;
	const int y = blockIdx.y;
	const int R = Q.run_lines;
	const int l0 = (int) blockIdx.x * R;
	if(l0 >= k.lines) return;                   /* the grid's x extent is padded to a multiple of 8: run x of every frame on XCD x % 8 */
	const int l1 = min(l0 + R, k.lines);
	const int FS = k.frame_samples;
	const hvk_packed_taps_t notch = { { 0 } };  /* (SECAM keeps the two-kernel path) */

	/* ---- LDS ---- */
	const int YL = raster_YL(W), CL = raster_CL(W);
	const int PB = (nth * SPL + 80 + 15) & ~15;                 /* bytes of one plane */
	int16_t *const rlds = (int16_t *) lds_raw;
	int16_t *const Yb = rlds, *const U = rlds + YL, *const V = rlds + YL + CL;
	unsigned char *const planes = lds_raw + (((YL + 2 * CL) * 2 + 15) & ~15);    /* [3][hi, lo][PB] */
	int *const outl = (int *) (planes + (VF ? 6 * PB : 0));     /* the filter's outputs on their way to the lane that owns them */
	int16_t *const tapd = (int16_t *) (outl + (VF ? nth * SPL : 0));
	int *const sym_st = (int *) (tapd + 4 * HVK_NICAM_TAPD);
	int4v *const sym_ent = (int4v *) (sym_st + HVK_NICAM_SYMS);

	/* four copies of the NICAM pulse table (int16), copy s shifted left by s entries, so that any run of
	 * 8 entries starts 8-byte aligned in one of them; staged once per workgroup */
	if(k.has_nicam)
	{
		for(int j = t; j < HVK_NICAM_TAPD / 2; j += nth) ((int4v *) tapd)[j] = ((const int4v *) Q.nicam_tapd)[j];
	}

	/* this lane's share of the tap matrix, 16 rows (8 outputs x I, Q) by 64 window positions */
	int4v a_hh = { 0, 0, 0, 0 }, a_hl = { 0, 0, 0, 0 };
	if(VF)
	{
		a_hh = Q.mfma_a[t & 63];
		a_hl = Q.mfma_a[64 + (t & 63)];
	}

	/* raster line r = l0 - 1 + it (VF = 0: l0 + it), filtered line fl = r - 2 (VF = 0: r) */
	const int niter = VF ? (l1 - l0) + 3 : (l1 - l0);
	for(int it = 0; it < niter; it++)
	{
		const int r = VF ? l0 - 1 + it : l0 + it;
		const int fl = VF ? r - 2 : r;
		const bool do_r = r <= l1;
		const bool do_f = fl >= l0 && fl < l1;
		const int rs = it % 3;                  /* plane buffer of line r; of line fl: (rs + 1) % 3 */

		/* ---- loads of both lines ---- */
		hvk_line_t L;
		uint32_t rgb[HVK_PIX_PASSES];
		int ghost_u = 0, ghost_v = 0, c[SPL];
		if(do_r)
		{
			L = raster_setup<0, EXTRAS>(k, P, y, r, first_frame, frame_stride);
			if(!L.zero) raster_loads<NT, WC>(k, P, L, t, nth, rgb, ghost_u, ghost_v, c);
		}

		const int n0 = fl * W;                  /* first output sample of the line, frame local */
		int symv = 0, cc_tile = 0;
		int4u car0 = { 0, 0, 0, 0 }, car1 = { 0, 0, 0, 0 };
		const bool whole = WC || x0 + SPL <= W;
		if(do_f)
		{
			if(k.has_nicam)
			{
				/* one dense row per line, prepared by the host: HVK_NICAM_SYMS symbol words then the mixer
				 * position of the line's first sample */
				const int *row = Q.tilesyms + ((size_t) y * k.lines + fl) * HVK_NICAM_ROW;
				cc_tile = row[HVK_NICAM_SYMS];
				symv = row[t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1];
			}
			if(k.has_carriers && whole)
			{
				const int4u *cp = (const int4u *) (Q.carriers + (size_t) y * FS + n0 + x0);
				car0 = cp[0];
				car1 = cp[1];
			}
		}

		if(do_r && !L.zero) raster_clear(L, t, nth, U, CL);

		if(do_f && k.has_nicam)
		{
			/* the symbols whose pulses can touch this line, oldest first: start (relative to the line's
			 * first sample) and sign pair. The schedule (src/nicam728.c:398-407) is tabulated per frame
			 * by the host. */
			if(t < HVK_NICAM_SYMS)
			{
				const int v = symv;
				const int st = (v >> 3) - n0;
				const bool valid = (v & 4) && st < W;
				/* constellation { 0, 1, 3, 2 }: bit 0 -> +I else -I, bit 1 -> +Q else -Q
				 * (src/nicam728.c:33, :386-396) */
				const int cs = (0x2310 >> ((v & 3) * 4)) & 3;
				sym_st[t] = valid ? st : 0x3FFFFFFF;
				const int rel = HVK_NICAM_LEAD - st;
				/* +1 or -1 in both halves: the pulse shapes two samples of a channel per packed multiply-add */
				const int sgi = (cs & 1) ? 0x00010001 : (int) 0xFFFFFFFFu;
				const int sgq = (cs & 2) ? 0x00010001 : (int) 0xFFFFFFFFu;
				sym_ent[t] = valid ? (int4v) { rel, (rel & 3) * HVK_NICAM_TAPD, sgi, sgq }
				                   : (int4v) { 0x10000000, 0, 0, 0 };
			}
		}
		__syncthreads();

		/* ---- pixels of line r into the staging area; line fl through the matrix unit ---- */
		if(do_r && !L.zero) raster_pixels<NT, WC, LV>(k, P, L, t, nth, rgb, ghost_u, ghost_v, Yb, U, V);

		/* the mixer row (i, -q) of this lane's samples: on its way while the filter and the pulse sums run */
		int4u mix_a0 = { 0, 0, 0, 0 }, mix_a1 = { 0, 0, 0, 0 };
		if(do_f && k.has_nicam)
		{
			int cp = cc_tile + x0;              /* mixer position of this lane's first sample */
			if(k.nicam_cc_len >= nth * SPL) { if(cp >= k.nicam_cc_len) cp -= k.nicam_cc_len; }
			else cp %= k.nicam_cc_len;
			mix_a0 = ((const int4u *) (Q.nicam_cca + cp))[0]; mix_a1 = ((const int4u *) (Q.nicam_cca + cp))[1];
		}

		if(VF && do_f)
		{
			/* The FIR as a banded matrix product (hvk_kernels.hip, hvk_k_filter): a wave takes 64
			 * segments of 8 outputs, 16 per v_mfma_i32_16x16x64_i8; lane (g, c) hands over window
			 * positions 16 g .. 16 g + 15 of segment c and gets back outputs 2 g, 2 g + 1 of it, I and Q. */
			const unsigned char *xh = planes + ((rs + 1) % 3) * 2 * PB, *xl = xh + PB;
			const int lane = t & 63, g = lane >> 4, cc = lane & 15;
#pragma unroll
			for(int j = 0; j < 4; j++)
			{
				const int seg = (t >> 6) * 64 + j * 16 + cc;
				const int off = seg * 8 + (FPAD - 26) + g * 16;
				int4v bh, bl;
				bh.xy = *(const int2v *) (xh + off); bh.zw = *(const int2v *) (xh + off + 8);
				bl.xy = *(const int2v *) (xl + off); bl.zw = *(const int2v *) (xl + off + 8);
				int4v p_hh = { 0, 0, 0, 0 }, p_m = { 0, 0, 0, 0 }, p_ll = { Q.mfma_ci, Q.mfma_cq, Q.mfma_ci, Q.mfma_cq };
				p_hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bh, p_hh, 0, 0, 0);
				p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bh, p_m, 0, 0, 0);
				p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bl, p_m, 0, 0, 0);
				p_ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bl, p_ll, 0, 0, 0);
				int yv[4];
#pragma unroll
				for(int i = 0; i < 4; i++) yv[i] = (int) ((((unsigned) p_hh[i] << 8) + (unsigned) p_m[i]) << 8) + p_ll[i];
				int2v pk;
				if(VF == 3)
				{
					pk.x = sat_pack16(yv[0] >> 15, yv[1] >> 15);
					pk.y = sat_pack16(yv[2] >> 15, yv[3] >> 15);
				}
				else
				{
					pk.x = sat_pack16(yv[0] >> 15, 0);
					pk.y = sat_pack16(yv[2] >> 15, 0);
				}
				*(int2v *) (outl + seg * 8 + 2 * g) = pk;
			}
		}
		__syncthreads();

		/* ---- 8 samples per lane of line r; 8 outputs per lane of line fl ---- */
		int s[SPL], cq[SPL];
		if(do_r)
		{
			if(L.zero)
			{
#pragma unroll
				for(int i = 0; i < SPL; i++) s[i] = 0;
			}
			else raster_compute<NT, 0, 0, EXTRAS, WC>(k, P, L, ctaps, notch, y, r + 1, t, nth, rlds, c, s, cq);

			if(VF)
			{
				/* the lane's samples as a plane of high bytes (x >> 8, signed) and a plane of low bytes
				 * less 128 (x & 255, read as signed after ^ 0x80); v_perm_b32 picks the bytes out of the
				 * sample pairs */
				const unsigned d0 = (s[0] & 0xFFFF) | ((unsigned) s[1] << 16), d1 = (s[2] & 0xFFFF) | ((unsigned) s[3] << 16);
				const unsigned d2 = (s[4] & 0xFFFF) | ((unsigned) s[5] << 16), d3 = (s[6] & 0xFFFF) | ((unsigned) s[7] << 16);
				int2v ph, pl;
				ph.x = (int) __builtin_amdgcn_perm(d1, d0, 0x07050301u);
				ph.y = (int) __builtin_amdgcn_perm(d3, d2, 0x07050301u);
				pl.x = (int) (__builtin_amdgcn_perm(d1, d0, 0x06040200u) ^ 0x80808080u);
				pl.y = (int) (__builtin_amdgcn_perm(d3, d2, 0x06040200u) ^ 0x80808080u);

				unsigned char *bh = planes + rs * 2 * PB;               /* line r's buffer */
				unsigned char *nh = planes + ((rs + 1) % 3) * 2 * PB;   /* line r + 1's: wants this line's tail */
				unsigned char *ph_ = planes + ((rs + 2) % 3) * 2 * PB;  /* line r - 1's: wants this line's head */
				if(WC || x0 + SPL <= W)
				{
					plane_put8(bh, x0 + FPAD, ph);
					plane_put8(bh + PB, x0 + FPAD, pl);
				}
				else if(x0 < W)
				{
					plane_put_some(bh, x0 + FPAD, ph, 0, W - x0);
					plane_put_some(bh + PB, x0 + FPAD, pl, 0, W - x0);
				}
				/* tail: samples W - 32 .. W - 1 at bytes 2 .. 33 of the next line's buffer */
				if(x0 + SPL > W - FEDGE && x0 < W)
				{
					const int j = x0 - W + FPAD;
					if((WC || (W & 7) == 0))
					{
						plane_put8(nh, j, ph);
						plane_put8(nh + PB, j, pl);
					}
					else
					{
						const int lo = max(0, W - FEDGE - x0), hi = min(SPL, W - x0);
						plane_put_some(nh, j, ph, lo, hi);
						plane_put_some(nh + PB, j, pl, lo, hi);
					}
				}
				/* head: samples 0 .. 31 behind the previous line's */
				if(x0 < FEDGE)
				{
					const int j = x0 + W + FPAD;
					if((WC || (W & 7) == 0))
					{
						plane_put8(ph_, j, ph);
						plane_put8(ph_ + PB, j, pl);
					}
					else
					{
						plane_put_some(ph_, j, ph, 0, SPL);
						plane_put_some(ph_ + PB, j, pl, 0, SPL);
					}
				}
			}
		}

		if(do_f)
		{
			int o[SPL];                             /* packed (I, Q) int16 */
			if(VF)
			{
				const int4v oa = ((const int4v *) (outl + x0))[0], ob = ((const int4v *) (outl + x0))[1];
				o[0] = oa.x; o[1] = oa.y; o[2] = oa.z; o[3] = oa.w;
				o[4] = ob.x; o[5] = ob.y; o[6] = ob.z; o[7] = ob.w;
			}
			else
			{
				/* no filter: the raster goes straight to I, Q = 0 */
#pragma unroll
				for(int i = 0; i < SPL; i++) o[i] = s[i] & 0xFFFF;
			}

			const int n = n0 + x0;
			/* serial carriers (FM / AM sound), computed on the host: a plain add of
			 * int16 pairs with wrap-around (src/video.c:3431-3432) */
			if(k.has_carriers)
			{
				if(whole)
				{
					o[0] = pk_add16(o[0], car0.x); o[1] = pk_add16(o[1], car0.y); o[2] = pk_add16(o[2], car0.z); o[3] = pk_add16(o[3], car0.w);
					o[4] = pk_add16(o[4], car1.x); o[5] = pk_add16(o[5], car1.y); o[6] = pk_add16(o[6], car1.z); o[7] = pk_add16(o[7], car1.w);
				}
				else if(x0 < W)
				{
					const int *cp = Q.carriers + (size_t) y * FS + n;
#pragma unroll
					for(int i = 0; i < SPL; i++) if(x0 + i < W) o[i] = pk_add16(o[i], cp[i]);
				}
			}

			/* NICAM: sum the pulses of the symbols in flight (int16 wrap-around per
			 * channel, both channels in one packed multiply-add), mix, add
			 * (src/nicam728.c:350-365, :386-396) */
			if(k.has_nicam)
			{
				const int last = x0 + SPL - 1;          /* relative to the line's first sample */
				/* newest symbol that has started by this lane's last sample; slot
				 * HVK_NICAM_BACK - 1 holds the newest one at the line's first sample */
				int idx = HVK_NICAM_BACK - 1 + (int) ((float) last * (1.0f / (float) k.nicam_sps));
				if(idx > HVK_NICAM_SYMS - 2) idx = HVK_NICAM_SYMS - 2;
				while(idx + 1 < HVK_NICAM_SYMS && sym_st[idx + 1] <= last) idx++;
				while(idx > 0 && sym_st[idx] > last) idx--;

				/* I and Q apart while the pulses are summed: (I[2m], I[2m + 1]) and (Q[2m], Q[2m + 1]) */
				int bi[SPL / 2], bq[SPL / 2];
#pragma unroll
				for(int i = 0; i < SPL / 2; i++) bi[i] = bq[i] = 0;

				/* the newest symbol and the six before it: everything older is over. A pulse
				 * that is over (or a slot without a symbol) reads the zero tail of the table:
				 * no branch. idx >= HVK_NICAM_BACK - 1 by construction. */
#pragma unroll 1
				for(int b = 0; b < HVK_NICAM_BACK; b++)
				{
					const int4v en = sym_ent[idx - b];
					int base = x0 + en.x;                                   /* >= 1 */
					base = base < HVK_NICAM_TAPD - SPL ? base : HVK_NICAM_TAPD - SPL;
					const int2v *tp = (const int2v *) (tapd + en.y + (base & ~3));
					const int2v ta = tp[0], tb = tp[1];
					bi[0] = pk_mad16(ta.x, en.z, bi[0]); bi[1] = pk_mad16(ta.y, en.z, bi[1]);
					bi[2] = pk_mad16(tb.x, en.z, bi[2]); bi[3] = pk_mad16(tb.y, en.z, bi[3]);
					bq[0] = pk_mad16(ta.x, en.w, bq[0]); bq[1] = pk_mad16(ta.y, en.w, bq[1]);
					bq[2] = pk_mad16(tb.x, en.w, bq[2]); bq[3] = pk_mad16(tb.y, en.w, bq[3]);
				}

				int bb[SPL];                            /* (I, Q) of each sample */
#pragma unroll
				for(int m = 0; m < SPL / 2; m++)
				{
					bb[2 * m + 0] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x05040100u);
					bb[2 * m + 1] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x07060302u);
				}

				/* mixer: the rotation's first row (i, -q) is tabulated (loaded before the filter) */
				const int4u a0 = mix_a0, a1 = mix_a1;
				const int ca[SPL] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
				/* the second row (q, i) from the first (i, -q): halves swapped, the low one negated (|q| <= 32767) */
				int cqr[SPL];
#pragma unroll
				for(int i = 0; i < SPL; i++) cqr[i] = pk_mad16(shift_pair(ca[i], ca[i]), (int) 0x0001FFFFu, 0);
#pragma unroll
				for(int i = 0; i < SPL; i++)
				{
					const int mi = dot2(bb[i], ca[i], 0);           /* bb.i * cc.i - bb.q * cc.q */
					const int mq = dot2(bb[i], cqr[i], 0);          /* bb.i * cc.q + bb.q * cc.i */
					/* ((mi >> 15) & 0xFFFF) | ((mq >> 15) << 16) */
					const int pk = (int) ((((unsigned) mq << 1) & 0xFFFF0000u) | (((unsigned) mi >> 15) & 0xFFFFu));
					o[i] = pk_add16(o[i], pk);
				}
			}

			/* interleaved int16 I/Q, 32 bytes per lane */
			int *dst = Q.iq + (size_t) y * Q.out_stride * FS + n;
			if(whole)
			{
				((int4u *) dst)[0] = (int4u) { o[0], o[1], o[2], o[3] };
				((int4u *) dst)[1] = (int4u) { o[4], o[5], o[6], o[7] };
			}
			else if(x0 < W)
			{
#pragma unroll
				for(int i = 0; i < SPL; i++) if(x0 + i < W) dst[i] = o[i];
			}
		}
		__syncthreads();
	}
}

/* ------------------------------------------------------------------ */

extern "C" void hvk_raster_ptrs(const hvk_raster_args_t *a, hvk_rptrs_t *P);

extern "C" size_t hvk_fused_lds_bytes(int width, int vf)
{
	const int nth = ((width + SPL - 1) / SPL + 63) / 64 * 64;
	const int YL = (width + 8 + 7) & ~7, CL = (width + 2 * HVK_CHROMA_LEAD + 7) & ~7;
	const int PB = (nth * SPL + 80 + 15) & ~15;
	size_t n = ((size_t) (YL + 2 * CL) * 2 + 15) & ~(size_t) 15;
	if(vf) n += (size_t) 6 * PB + (size_t) nth * SPL * 4;
	n += 4 * HVK_NICAM_TAPD * 2 + HVK_NICAM_SYMS * 4 + HVK_NICAM_SYMS * 16;
	return(n + 64);
}

template<int NT, int VF, int EXTRAS, int WC, int LV>
static int _launch_fused4(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	const int W = ra->k.width;
	const int threads = ((W + SPL - 1) / SPL + 63) / 64 * 64;
	const int runs = (ra->k.lines + run_lines - 1) / run_lines;
	hvk_rptrs_t P;
	hvk_fptrs_t Q;
	hvk_raster_ptrs(ra, &P);
	Q.carriers = (const int *) fa->carriers;
	Q.tilesyms = fa->tilesyms;
	Q.nicam_tapd = fa->nicam_tapd;
	Q.nicam_cca = fa->nicam_cca;
	Q.mfma_a = (const int4v *) fa->mfma_a;
	Q.mfma_ci = fa->mfma_ci;
	Q.mfma_cq = fa->mfma_cq;
	Q.iq = (int *) fa->iq;
	Q.out_stride = fa->out_stride;
	Q.run_lines = run_lines;
	if(threads > 256) return(HVK_UNSUPPORTED);
	hipLaunchKernelGGL((hvk_k_fused<NT, VF, EXTRAS, WC, LV>), dim3((runs + 7) & ~7, ra->nframes), dim3(threads), hvk_fused_lds_bytes(W, VF), stream,
	                   ra->k, ra->ctaps, P, Q, ra->first_frame, ra->frame_stride);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

template<int NT, int VF, int EXTRAS, int WC>
static int _launch_fused3(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	if(ra->levels_computed) return(_launch_fused4<NT, VF, EXTRAS, WC, 1>(ra, fa, run_lines, stream));
	return(_launch_fused4<NT, VF, EXTRAS, WC, 0>(ra, fa, run_lines, stream));
}

template<int NT, int VF>
static int _launch_fused2(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	const bool extras = ra->k.vbi || ra->k.vits;
	if(extras) return(_launch_fused3<NT, VF, 1, 0>(ra, fa, run_lines, stream));
	/* the plain PAL kernel at 1024 samples per line gets the width as a constant */
	if(NT == 13 && ra->k.width == 1024) return(_launch_fused3<NT, VF, 0, NT == 13 ? 1024 : 0>(ra, fa, run_lines, stream));
	return(_launch_fused3<NT, VF, 0, 0>(ra, fa, run_lines, stream));
}

template<int NT>
static int _launch_fused1(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	switch(ra->k.vf_type)
	{
	case 0: return(_launch_fused2<NT, 0>(ra, fa, run_lines, stream));
	case 1: return(_launch_fused2<NT, 1>(ra, fa, run_lines, stream));
	case 3: return(_launch_fused2<NT, 3>(ra, fa, run_lines, stream));
	}
	return(HVK_UNSUPPORTED);
}

/* Can this configuration run as one kernel? */
extern "C" int hvk_fused_supported(const hvk_kconst_t *k, const void *mfma_a)
{
	if(k->secam || k->s_video || k->rawbb || k->rs_L) return(0);
	if(k->vf_type != 0 && (k->vf_ntaps != 51 || !mfma_a)) return(0);
	if(k->vf_type != 0 && k->vf_type != 1 && k->vf_type != 3) return(0);
	if((k->width + SPL - 1) / SPL > 256) return(0);
	if(k->width < 2 * FEDGE + 16) return(0);
	if(k->colour)
	{
		const int nt = k->chroma_ntaps;
		if(nt != 9 && nt != 11 && nt != 13 && nt != 15 && nt != 17 && nt != 21) return(0);
	}
	return(1);
}

extern "C" int hvk_launch_fused(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	if(!hvk_fused_supported(&ra->k, fa->mfma_a) || run_lines < 1) return(HVK_UNSUPPORTED);
	switch(ra->k.colour ? ra->k.chroma_ntaps : 1)
	{
	case 1:  return(_launch_fused1<1>(ra, fa, run_lines, stream));   /* monochrome */
	case 9:  return(_launch_fused1<9>(ra, fa, run_lines, stream));
	case 11: return(_launch_fused1<11>(ra, fa, run_lines, stream));
	case 13: return(_launch_fused1<13>(ra, fa, run_lines, stream));
	case 15: return(_launch_fused1<15>(ra, fa, run_lines, stream));
	case 17: return(_launch_fused1<17>(ra, fa, run_lines, stream));
	case 21: return(_launch_fused1<21>(ra, fa, run_lines, stream));
	}
	return(HVK_UNSUPPORTED);
}
