/* hvk_fused.hip -- the whole per-sample path of a render in ONE kernel: raster, video filter,
 * sound carriers, NICAM, interleaved int16 I/Q out (src/video.c:2864-3066 + src/fir.c:564-615 +
 * src/video.c:3431-3432 + src/nicam728.c:342-411). The raster never touches HBM.
 *
 * A workgroup walks a run of consecutive lines of one frame. Line r is rasterised (the steps of
 * hvk_device.h) into LDS as two BYTE PLANES -- the form the int8 matrix unit multiplies -- while
 * line r - 2, whose successor's leading samples are by then in place, goes through the filter.
 *
 * hvk_k_fusedw, the kernel of the plain configurations (the benchmark's among them), divides the work
 * between the WAVES of the workgroup -- half of them "sample" waves, half "io" waves, about the same
 * number of vector instructions each, ONE barrier per line (the hand-over areas exist twice):
 *
 *        sample waves                                        io waves
 *   ------------------------------------ barrier ------------------------------------
 *        line r - 2 through the 51-tap filter on the         line r + 1's pixel levels (looked up during the
 *        matrix unit: planes -> output exchange;             last round) -> Y / U / V staging in LDS; look-ups
 *        8 samples per lane of line r: base line, luma,      of line r + 2 and source row of line r + 3 go out;
 *        chroma low pass, burst, QAM -> planes; the next     carriers + NICAM onto the filtered samples of line
 *        line's base line / burst / phasor loads go out      r - 3, 32-byte stores; symbols, mixer row and
 *                                                            carriers of line r - 2 go out
 *   ------------------------------------ barrier ------------------------------------
 *
 * A wave is in-order: in one wave the raster's and the filter's dependent loads and LDS round trips
 * queue up behind each other (a wait for the newest vector load is a wait for all older ones), and
 * everything either job keeps in flight has to live in that one wave's registers. Here the register-hungry
 * steps (chroma low pass; matrix products; pulse sums) never share a wave with the look-ups in flight, every
 * load is issued a phase or a line before it is used, and a line's descriptors come out of register lanes
 * (fused_desc_t) instead of memory.
 *
 * hvk_k_fused does both jobs in every wave, one after the other. It serves the configurations with
 * barriers inside the raster (VBI data lines, insertion test signals) and those without a video
 * filter, where a lane's raster samples are its outputs and there are no planes.
 *
 * Three plane buffers rotate (four in hvk_k_fusedw, whose waves write a line's planes while others still read
 * the line two back). Buffer b holds, for its line, the last 32 samples of the line before,
 * the line, and the first 32 samples of the line after (the filter reaches 25 either way): a lane
 * that holds edge samples writes them into the neighbour's buffer as well. The first line of a run
 * needs the tail of the line before the run and the last one the head of the line after it: of those
 * two lines only that much is rasterised.
 *
 * Plane coordinates: sample x of the buffer's line sits at byte x + 34 -- 2 modulo 8, so that the
 * 64-byte window of an 8-output segment (which starts 26 samples before the segment) starts 8-byte
 * aligned; a lane's 8 bytes therefore go out as 2 + 4 + 2.
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
	int nruns;                /* runs per frame: run x covers lines [x * lines / nruns, (x + 1) * lines / nruns) */
	/* hvk_k_fusedw loads unconditionally: where a configuration has no carriers / no NICAM the pointers above
	 * name a block of zeros and these strides are 0, so that every such load reads its first bytes */
	int car_frame, car_line;  /* carrier samples per frame, per line (0: none) */
	int sym_frame, sym_line;  /* symbol row ints per frame, per line (0: none) */
	const void *lstate;       /* hvk_k_fusedw: the lines' states, [frames][lines + 2] records of 32 bytes (hvk_k_linestate writes them) */
} hvk_fptrs_t;

/* entries of one copy of the NICAM pulse table in LDS: the lead, the pulse, and a zero tail a lane's 8 samples wide */
__host__ __device__ __forceinline__ int fused_tapd_len(int nicam_ntaps) { return((HVK_NICAM_LEAD + nicam_ntaps + SPL + 7) & ~7); }

/* the workgroup's LDS. hvk_k_fusedw (ws) has its hand-over areas twice -- staging, output exchange, symbol
 * table: one side fills one while the other side reads the other -- and four plane buffers instead of three */
typedef struct {
	int16_t *rlds;            /* raster staging: Y, U, V */
	int rstride;              /* int16 from one staging area to the other */
	unsigned char *planes;    /* [3 or 4][hi, lo][PB] */
	int PB;
	int *outl;                /* the filter's outputs on their way to the lane that owns them; ostride ints to the other one */
	int ostride;
	int *sym_st;              /* NICAM symbols of the line: start relative to its first sample */
	int4v *sym_ent;           /*   { LEAD - start, copy offset, sign pair I, sign pair Q } */
	int sstride;              /* bytes from one symbol table (sym_st + sym_ent) to the other */
	int16_t *tapd;            /* four copies of the pulse, copy s one entry further left */
	int TL;                   /* entries of one copy */
} fused_lds_t;

#define FUSED_SYM_BYTES (HVK_NICAM_SYMS * 4 + HVK_NICAM_SYMS * 16)

__host__ __device__ __forceinline__ size_t fused_lds_layout(const int W, const int nth, const int vf, const int nicam_ntaps, const int ws,
                                                            int *o_planes, int *o_outl, int *o_sym, int *o_tapd, int *o_PB, int *o_rs)
{
	const int YL = (W + 8 + 7) & ~7, CL = (W + 2 * HVK_CHROMA_LEAD + 7) & ~7;
	const int PB = (nth * SPL + 80 + 15) & ~15;
	const int rbytes = ((YL + 2 * CL) * 2 + 15) & ~15;
	size_t n = (size_t) rbytes * (ws ? 2 : 1);
	*o_rs = rbytes / 2;
	*o_planes = (int) n;
	if(vf) n += (size_t) (ws ? 8 : 6) * PB;
	*o_outl = (int) n;
	if(vf) n += (size_t) nth * SPL * 4 * (ws ? 2 : 1);
	*o_sym = (int) n;
	n += (size_t) FUSED_SYM_BYTES * (ws ? 2 : 1);
	*o_tapd = (int) n;
	n += (size_t) 4 * ((HVK_NICAM_LEAD + nicam_ntaps + SPL + 7) & ~7) * 2;
	*o_PB = PB;
	return(n);
}

__device__ __forceinline__ fused_lds_t fused_lds(unsigned char *raw, const int W, const int nth, const int vf, const int nicam_ntaps, const int ws = 0)
{
	fused_lds_t l;
	int op, oo, os, ot;
	fused_lds_layout(W, nth, vf, nicam_ntaps, ws, &op, &oo, &os, &ot, &l.PB, &l.rstride);
	l.TL = fused_tapd_len(nicam_ntaps);
	l.rlds = (int16_t *) raw;
	l.planes = raw + op;
	l.outl = (int *) (raw + oo);
	l.ostride = nth * SPL;
	l.sym_st = (int *) (raw + os);
	l.sym_ent = (int4v *) (l.sym_st + HVK_NICAM_SYMS);
	l.sstride = FUSED_SYM_BYTES;
	l.tapd = (int16_t *) (raw + ot);
	return(l);
}

/* four copies of the NICAM pulse table (int16), copy s shifted left by s entries, so that any run of
 * 8 entries starts 8-byte aligned in one of them; staged once per workgroup by lanes [0, n) */
__device__ __forceinline__ void fused_stage_tapd(const fused_lds_t &l, const int *nicam_tapd, const int t, const int n)
{
	for(int j = t; j < 4 * l.TL / 8; j += n)
	{
		const int copy = (j * 8) / l.TL, at = j * 8 - copy * l.TL;
		((int4v *) l.tapd)[j] = *(const int4v *) ((const int16_t *) nicam_tapd + copy * HVK_NICAM_TAPD + at);
	}
}

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

/* The lane's samples of line r as a plane of high bytes (x >> 8, signed) and a plane of low bytes less
 * 128 (x & 255, read as signed after ^ 0x80) -- v_perm_b32 picks the bytes out of the sample pairs --
 * into the line's buffer `rs` and, for edge samples, into the neighbours'. part: 1 / 2 only the tail /
 * head is wanted. */
template<int WC>
__device__ __forceinline__ void fused_put_planes(const fused_lds_t &l, const int (&s)[SPL], const int x0, const int W, const int rs, const int part,
                                                 const int nbuf = 3)
{
	const unsigned d0 = (s[0] & 0xFFFF) | ((unsigned) s[1] << 16), d1 = (s[2] & 0xFFFF) | ((unsigned) s[3] << 16);
	const unsigned d2 = (s[4] & 0xFFFF) | ((unsigned) s[5] << 16), d3 = (s[6] & 0xFFFF) | ((unsigned) s[7] << 16);
	int2v ph, pl;
	ph.x = (int) __builtin_amdgcn_perm(d1, d0, 0x07050301u);
	ph.y = (int) __builtin_amdgcn_perm(d3, d2, 0x07050301u);
	pl.x = (int) (__builtin_amdgcn_perm(d1, d0, 0x06040200u) ^ 0x80808080u);
	pl.y = (int) (__builtin_amdgcn_perm(d3, d2, 0x06040200u) ^ 0x80808080u);

	const int PB = l.PB;
	unsigned char *bh = l.planes + rs * 2 * PB;                                     /* line r's buffer */
	unsigned char *nh = l.planes + (rs + 1 == nbuf ? 0 : rs + 1) * 2 * PB;          /* line r + 1's: wants this line's tail */
	unsigned char *vh = l.planes + (rs == 0 ? nbuf - 1 : rs - 1) * 2 * PB;          /* line r - 1's: wants this line's head */
	if(part) { }                    /* a halo line: only its edge is read */
	else if(WC || x0 + SPL <= W)
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
	if(part != 2 && x0 + SPL > W - FEDGE && x0 < W)
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
	if(part != 1 && x0 < FEDGE)
	{
		const int j = x0 + W + FPAD;
		if((WC || (W & 7) == 0))
		{
			plane_put8(vh, j, ph);
			plane_put8(vh + PB, j, pl);
		}
		else
		{
			plane_put_some(vh, j, ph, 0, SPL);
			plane_put_some(vh + PB, j, pl, 0, SPL);
		}
	}
}

/* The FIR as a banded matrix product (hvk_kernels.hip, hvk_k_filter): a wave takes 64 segments of 8
 * outputs, 16 per v_mfma_i32_16x16x64_i8; lane (g, c) hands over window positions 16 g .. 16 g + 15 of
 * segment c and gets back outputs 2 g, 2 g + 1 of it, I and Q. t: lane within the filter's lanes. */
template<int VF>
__device__ __forceinline__ void fused_filter(const fused_lds_t &l, const hvk_fptrs_t &Q, const int4v a_hh, const int4v a_hl, const int t, const int buf,
                                             int *outl = NULL)
{
	if(outl == NULL) outl = l.outl;
	const unsigned char *xh = l.planes + buf * 2 * l.PB, *xl = xh + l.PB;
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

/* The symbols whose pulses can touch the line, oldest first: start (relative to the line's first
 * sample) and sign pair, from the row the host tabulated (src/nicam728.c:398-407). Lanes < HVK_NICAM_SYMS. */
__device__ __forceinline__ void fused_symtab(const fused_lds_t &l0_, const int t, const int symv, const int n0, const int W, const int which = 0)
{
	fused_lds_t l = l0_;
	l.sym_st = (int *) ((unsigned char *) l0_.sym_st + which * l0_.sstride);
	l.sym_ent = (int4v *) ((unsigned char *) l0_.sym_ent + which * l0_.sstride);
	if(t < HVK_NICAM_SYMS)
	{
		const int v = symv;
		const int st = (v >> 3) - n0;
		const bool valid = (v & 4) && st < W;
		/* constellation { 0, 1, 3, 2 }: bit 0 -> +I else -I, bit 1 -> +Q else -Q
		 * (src/nicam728.c:33, :386-396) */
		const int cs = (0x2310 >> ((v & 3) * 4)) & 3;
		l.sym_st[t] = valid ? st : 0x3FFFFFFF;
		const int rel = HVK_NICAM_LEAD - st;
		/* +1 or -1 in both halves: the pulse shapes two samples of a channel per packed multiply-add */
		const int sgi = (cs & 1) ? 0x00010001 : (int) 0xFFFFFFFFu;
		const int sgq = (cs & 2) ? 0x00010001 : (int) 0xFFFFFFFFu;
		l.sym_ent[t] = valid ? (int4v) { rel, (rel & 3) * l.TL, sgi, sgq }
		                     : (int4v) { 0x10000000, 0, 0, 0 };
	}
}

/* mixer table position of the lane's first sample */
__device__ __forceinline__ int fused_mix_pos(const hvk_kconst_t &k, const int cc_line, const int x0, const int nth)
{
	int cp = cc_line + x0;
	if(k.nicam_cc_len >= nth * SPL) { if(cp >= k.nicam_cc_len) cp -= k.nicam_cc_len; }
	else cp %= k.nicam_cc_len;
	return(cp);
}

/* Sound onto the lane's 8 filtered samples o[] (packed I, Q) and out: serial carriers (FM / AM, computed on
 * the host: a plain add of int16 pairs with wrap-around, src/video.c:3431-3432), then NICAM -- the pulses of
 * the symbols in flight summed per channel with int16 wrap-around, mixed, added (src/nicam728.c:350-365,
 * :386-396). x0: the lane's first sample of the line; dst: where it goes. */
template<int WC>
__device__ __forceinline__ void fused_finish(const hvk_kconst_t &k, const fused_lds_t &l_, int (&o)[SPL], const int x0, const int W,
                                             const bool whole, const int4u car0, const int4u car1, const int *car_tail,
                                             const int4u mix_a0, const int4u mix_a1, int *dst, const bool emit = true, const int which = 0)
{
	fused_lds_t l = l_;
	l.sym_st = (int *) ((unsigned char *) l_.sym_st + which * l_.sstride);
	l.sym_ent = (int4v *) ((unsigned char *) l_.sym_ent + which * l_.sstride);
	if(k.has_carriers)
	{
		if(whole || car_tail == NULL)       /* (NULL: the 8 values were loaded from the lane's position whatever it is) */
		{
			o[0] = pk_add16(o[0], car0.x); o[1] = pk_add16(o[1], car0.y); o[2] = pk_add16(o[2], car0.z); o[3] = pk_add16(o[3], car0.w);
			o[4] = pk_add16(o[4], car1.x); o[5] = pk_add16(o[5], car1.y); o[6] = pk_add16(o[6], car1.z); o[7] = pk_add16(o[7], car1.w);
		}
		else if(x0 < W)
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) if(x0 + i < W) o[i] = pk_add16(o[i], car_tail[i]);
		}
	}

	if(k.has_nicam && !ABLATE(2048))
	{
		const int last = x0 + SPL - 1;          /* relative to the line's first sample */
		const int TL = l.TL;
		/* newest symbol that has started by this lane's last sample; slot HVK_NICAM_BACK - 1 holds the newest
		 * one at the line's first sample. The estimate is off by one at most almost everywhere: its slot and
		 * the next are read together. */
		int idx = HVK_NICAM_BACK - 1 + (int) ((float) last * (1.0f / (float) k.nicam_sps));
		if(idx > HVK_NICAM_SYMS - 2) idx = HVK_NICAM_SYMS - 2;
		{
			const int sa = l.sym_st[idx], sb = l.sym_st[idx + 1];
			if(sb <= last) idx++;
			else if(sa > last) idx--;
		}
		while(idx + 1 < HVK_NICAM_SYMS && l.sym_st[idx + 1] <= last) idx++;
		while(idx > 0 && l.sym_st[idx] > last) idx--;

		/* I and Q apart while the pulses are summed: (I[2m], I[2m + 1]) and (Q[2m], Q[2m + 1]) */
		int bi[SPL / 2], bq[SPL / 2];
#pragma unroll
		for(int i = 0; i < SPL / 2; i++) bi[i] = bq[i] = 0;
		/* the newest symbol and the six before it: everything older is over. A pulse that is over (or a slot
		 * without a symbol) reads the zero tail of the table: no branch. idx >= HVK_NICAM_BACK - 1 by
		 * construction. In two batches (4 + 3 symbols): the table entries of a batch are read together,
		 * then its pulse slices -- four LDS round trips instead of fourteen, without holding all seven. */
#pragma unroll
		for(int b0 = 0; b0 < HVK_NICAM_BACK; b0 += 4)
		{
			constexpr int NB = 4;
			int4v en[NB];
			int2v ta[NB], tb[NB];
#pragma unroll
			for(int b = 0; b < NB; b++) if(b0 + b < HVK_NICAM_BACK) en[b] = l.sym_ent[idx - b0 - b];
#pragma unroll
			for(int b = 0; b < NB; b++)
			{
				if(b0 + b >= HVK_NICAM_BACK) continue;
				int base = x0 + en[b].x;                                /* >= 1 */
				base = base < TL - SPL ? base : TL - SPL;
				const int2v *tp = (const int2v *) (l.tapd + en[b].y + (base & ~3));
				ta[b] = tp[0];
				tb[b] = tp[1];
			}
#pragma unroll
			for(int b = 0; b < NB; b++)
			{
				if(b0 + b >= HVK_NICAM_BACK) continue;
				bi[0] = pk_mad16(ta[b].x, en[b].z, bi[0]); bi[1] = pk_mad16(ta[b].y, en[b].z, bi[1]);
				bi[2] = pk_mad16(tb[b].x, en[b].z, bi[2]); bi[3] = pk_mad16(tb[b].y, en[b].z, bi[3]);
				bq[0] = pk_mad16(ta[b].x, en[b].w, bq[0]); bq[1] = pk_mad16(ta[b].y, en[b].w, bq[1]);
				bq[2] = pk_mad16(tb[b].x, en[b].w, bq[2]); bq[3] = pk_mad16(tb[b].y, en[b].w, bq[3]);
			}
		}

		int bb[SPL];                            /* (I, Q) of each sample */
#pragma unroll
		for(int m = 0; m < SPL / 2; m++)
		{
			bb[2 * m + 0] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x05040100u);
			bb[2 * m + 1] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x07060302u);
		}

		/* mixer: the rotation's first row (i, -q) is tabulated */
		const int ca[SPL] = { mix_a0.x, mix_a0.y, mix_a0.z, mix_a0.w, mix_a1.x, mix_a1.y, mix_a1.z, mix_a1.w };
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
	if(!emit) { }
	else if(whole)
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

/* a line on its way to the raster: its state, and how much of it is wanted */
typedef struct {
	hvk_line_t L;
	int part;               /* 0: all of it; 1: only the tail (the line before the run); 2: only the head (the line after it) */
} fused_prep_t;

/* Of the line before the run only the tail is wanted, of the line after it only the head: the waves
 * that hold none of it sit the line out, and one pass of pixels covers what the tail's chroma low
 * pass reaches (1); the head (2) has neither pixels nor sub-carrier when the picture and the burst
 * start beyond it. (Lines with VBI data or test signals are done in full.) */
template<int NT, int VF, int EXTRAS, int WC>
__device__ __forceinline__ fused_prep_t fused_prep(const hvk_kconst_t &k, const hvk_rptrs_t &P, const int y, const int r, const int l0, const int l1,
                                                   const int64_t first_frame, const int64_t frame_stride)
{
	fused_prep_t q;
	const int W = WC ? WC : k.width;
	q.part = (VF && !EXTRAS) ? (r == l0 - 1 ? 1 : (r == l1 ? 2 : 0)) : 0;
	q.L = raster_setup<0, EXTRAS>(k, P, y, r, first_frame, frame_stride);
	if(q.part == 2)
	{
		if((!q.L.active || q.L.d.al >= FEDGE + 16) && (!q.L.pal || k.burst_left >= FEDGE + 16))
		{
			q.L.pal = 0;
			q.L.has_pix = false;
			q.L.ax0 = q.L.ax1 = 0;
		}
		else q.part = 0;
	}
	else if(q.part == 1 && q.L.has_pix)
	{
		const int lo = W - FEDGE - NT / 2 - 8;
		if(q.L.ax0 < lo) q.L.ax0 = lo;
		if(q.L.ax0 >= q.L.ax1) { q.L.has_pix = false; q.L.ax0 = q.L.ax1 = 0; }
	}
	return(q);
}

/* A line's state as hvk_k_linestate tabulates it before the main kernel runs: what raster_setup() derives from
 * the line's and the frame's descriptors, 32 bytes per line of the slab (line -1 first). The main kernel's
 * waves keep the records of their run's lines in the lanes of two registers (lane j: line l0 - 1 + j; a run
 * has 61 lines at most) and pick a line's out with v_readlane: no descriptor fetch, and next to no scalar
 * arithmetic, per line. */
typedef struct {
	int4v a;        /* row_off (lo, hi), coff, ax0 | ax1 << 16 */
	int4v b;        /* al | ar << 16, ar_eff | base row << 16, flags, - */
} fused_rec_t;

#define FREC_PAL_POS 1
#define FREC_PAL_NEG 2
#define FREC_ACTIVE  4
#define FREC_HAS_PIX 8
#define FREC_OWN     16
#define FREC_ZERO    32

__global__ __launch_bounds__(64)
void hvk_k_linestate(const hvk_kconst_t k, const hvk_rptrs_t P, fused_rec_t *__restrict__ out, const int64_t first_frame, const int64_t frame_stride)
{
	const int y = blockIdx.y;
	const int i = blockIdx.x * 64 + threadIdx.x;        /* slab line: line i - 1 of the frame */
	if(i >= k.lines + 2) return;
	const int rel = i - 1;
	int line0, par;
	bool own, zero;
	raster_line_index(k, rel, first_frame + (int64_t) y * frame_stride, line0, par, own, zero);
	const hvk_framedesc_t f = P.fdesc[y * (k.fields + 1) + raster_fdesc_of(k, rel)];
	const hvk_linedesc_t d = P.desc[par * k.lines + line0];
	const hvk_line_t L = raster_setup_core<0, 0>(k, P, f, d, y, rel, line0, own, zero);
	fused_rec_t r;
	r.a.x = (int) (uint32_t) (uint64_t) L.row_off;
	r.a.y = (int) (uint32_t) ((uint64_t) L.row_off >> 32);
	r.a.z = (int) L.coff;
	r.a.w = (L.ax0 & 0xFFFF) | (L.ax1 << 16);
	r.b.x = ((int) d.al & 0xFFFF) | ((int) d.ar << 16);
	r.b.y = (L.ar_eff & 0xFFFF) | ((d.secam_fid >> 8) << 16);
	r.b.z = (L.pal > 0 ? FREC_PAL_POS : 0) | (L.pal < 0 ? FREC_PAL_NEG : 0) | (L.active ? FREC_ACTIVE : 0) | (L.has_pix ? FREC_HAS_PIX : 0) |
	        (L.own ? FREC_OWN : 0) | (L.zero ? FREC_ZERO : 0);
	r.b.w = 0;
	out[(size_t) y * (k.lines + 2) + i] = r;
}

__device__ __forceinline__ fused_rec_t fused_rec_load(const fused_rec_t *lstate, const int lines, const int y, const int l0, const int l1, const int lane)
{
	int rel = l0 - 1 + lane;
	if(rel > l1) rel = l1;
	const fused_rec_t *p = lstate + (size_t) y * (lines + 2) + rel + 1;
	fused_rec_t r;
	r.a = p->a;
	r.b = p->b;
	return(r);
}

template<int NT, int VF, int WC>
__device__ __forceinline__ fused_prep_t fused_prep_lanes(const hvk_kconst_t &k, const fused_rec_t &D, const int r, const int l0, const int l1)
{
	fused_prep_t q;
	const int W = WC ? WC : k.width;
	const int j = r - (l0 - 1);
	const int ax = __builtin_amdgcn_readlane(D.a.w, j), aa = __builtin_amdgcn_readlane(D.b.x, j);
	const int eb = __builtin_amdgcn_readlane(D.b.y, j), fl = __builtin_amdgcn_readlane(D.b.z, j);
	hvk_line_t &L = q.L;
	L.rel = r;
	L.row_off = (int64_t) (((uint64_t) (unsigned) __builtin_amdgcn_readlane(D.a.y, j) << 32) | (unsigned) __builtin_amdgcn_readlane(D.a.x, j));
	L.coff = (unsigned) __builtin_amdgcn_readlane(D.a.z, j);
	L.ax0 = ax & 0xFFFF;
	L.ax1 = (unsigned) ax >> 16;
	L.ar_eff = eb & 0xFFFF;
	L.d.pulse_left = L.d.pulse_mid = L.d.pulse_next = -1;
	L.d.al = (int16_t) (aa & 0xFFFF);
	L.d.ar = (int16_t) ((unsigned) aa >> 16);
	L.d.src_row = 0;
	L.d.pal = 0;
	L.d.secam_fid = (int16_t) (((unsigned) eb >> 16) << 8);
	L.pal = (fl & FREC_PAL_POS) ? 1 : ((fl & FREC_PAL_NEG) ? -1 : 0);
	L.active = (fl & FREC_ACTIVE) != 0;
	L.has_pix = (fl & FREC_HAS_PIX) != 0;
	L.own = (fl & FREC_OWN) != 0;
	L.zero = (fl & FREC_ZERO) != 0;
	L.vbi_op = L.vits_i = -1;

	q.part = VF ? (r == l0 - 1 ? 1 : (r == l1 ? 2 : 0)) : 0;
	if(q.part == 2)
	{
		if((!L.active || L.d.al >= FEDGE + 16) && (!L.pal || k.burst_left >= FEDGE + 16))
		{
			L.pal = 0;
			L.has_pix = false;
			L.ax0 = L.ax1 = 0;
		}
		else q.part = 0;
	}
	else if(q.part == 1 && L.has_pix)
	{
		const int lo = W - FEDGE - NT / 2 - 8;
		if(L.ax0 < lo) L.ax0 = lo;
		if(L.ax0 >= L.ax1) { L.has_pix = false; L.ax0 = L.ax1 = 0; }
	}
	return(q);
}

/* the run of lines of workgroup x: the frame's lines are dealt evenly to nruns runs; the host makes nruns a
 * multiple of 8 so that, with workgroups dealt round-robin to the 8 XCDs, run x of EVERY frame runs on XCD
 * x % 8 (the slices of the colour table and of the source frame a run needs stay in that XCD's L2) and every
 * XCD has the same work */
__device__ __forceinline__ void fused_run(const int lines, const int nruns, int &l0, int &l1)
{
	l0 = (int) (((long) blockIdx.x * lines) / nruns);
	l1 = (int) (((long) (blockIdx.x + 1) * lines) / nruns);
}

/* The workgroup's waves hand data to each other through LDS only: the barrier waits for this wave's LDS
 * traffic, not -- as __syncthreads() does -- for every vector load and store it has in flight (the
 * prefetches of the next lines, the output stores). */
#define FUSED_BARRIER() do { if(!ABLATE(256)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while(0)

/* "The value is wanted here": the wait for a load lands where this stands -- in front of the NEXT lines' loads,
 * when what is waited for came in long ago -- instead of at the first use further down, where (vector loads
 * return in order) it would be a wait for the loads just issued as well. */
__device__ __forceinline__ void touch(int &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void touch(uint32_t &v) { asm volatile("" : "+v"(v)); }
#define touch4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
__device__ __forceinline__ void touch(short4v &v) { int2v w = __builtin_bit_cast(int2v, v); asm volatile("" : "+v"(w.x), "+v"(w.y)); v = __builtin_bit_cast(short4v, w); }

/* profiling build only (HVK_ABLATE bit 16384): cycle counts of a workgroup's phases, written over its first output line */
#if HVK_ENABLE_ABLATE
#define TS_DECL long ts_acc[4] = { 0, 0, 0, 0 }; long ts_last = clock64(); const long ts_start = ts_last
#define TS_MARK(i) do { if(ABLATE(16384)) { const long n_ = clock64(); ts_acc[i] += n_ - ts_last; ts_last = n_; } } while(0)
#define TS_DUMP(role) do { if(ABLATE(16384) && t == 0) { long *o_ = (long *) (Q.iq + (size_t) y * Q.out_stride * FS + (size_t) l0 * W) + (role) * 8; \
	o_[0] = ts_start; o_[1] = clock64(); o_[2] = ts_acc[0]; o_[3] = ts_acc[1]; o_[4] = ts_acc[2]; o_[5] = ts_acc[3]; \
	unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); unsigned xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_)); o_[6] = hw_ | ((long) xc_ << 32); o_[7] = niter; } } while(0)
#else
#define TS_DECL
#define TS_MARK(i)
#define TS_DUMP(role)
#endif

/* ------------------------------------------------------------------ */
/* raster waves and filter waves                                        */

/* The loop bodies below are straight-line on purpose. The compiler places its waits for vector loads
 * conservatively wherever control flow joins, and a wait for one load is a wait for every load issued
 * before it: a branch around a load or around its first use turns the prefetches of the next lines into
 * stalls. So every load goes out unconditionally (at a clamped position where it has no use), the rounds
 * that fill and drain a run's pipeline work on whatever is there, and only stores are masked. */
template<int NT, int VF, int WC, int LV>
__global__ __launch_bounds__(512, 4)
void hvk_k_fusedw(const hvk_kconst_t k,
                  const hvk_packed_taps_t ctaps,
                  const hvk_rptrs_t P,
                  const hvk_fptrs_t Q,
                  const int64_t first_frame,
                  const int64_t frame_stride)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	static_assert(VF != 0, "without a video filter there is nothing to hand over");

	const int W = WC ? WC : k.width;
	const int nth = WC ? WC / SPL : (int) blockDim.x / 2;       /* lanes of each job */
	const bool io = (int) threadIdx.x >= nth;                   /* wave-uniform: nth is a multiple of 64 */
	const int t = io ? (int) threadIdx.x - nth : (int) threadIdx.x;
	const int x0 = t * SPL;
	const int xc = x0 < W ? x0 : 0;             /* where a lane beyond the line loads from */
	const int y = blockIdx.y;
	int l0, l1;
	fused_run(k.lines, Q.nruns, l0, l1);
	if(l1 <= l0) return;
	const int FS = k.frame_samples;
	const fused_lds_t l = fused_lds(lds_raw, W, nth, VF, k.has_nicam ? k.nicam_ntaps : 0, 1);
	/* Round `it`: the sample waves filter line r - 2 and make the samples of line r = l0 - 1 + it; the io
	 * waves stage line r + 1 and send out line r - 3. One barrier per round: every hand-over area exists
	 * twice and the plane buffers rotate by four, so what one side writes in a round the other reads in the next. */
	const int niter = (l1 - l0) + 4;
	fused_rec_t D = fused_rec_load((const fused_rec_t *) Q.lstate, k.lines, y, l0, l1, t & 63);
	touch4(D.a); touch4(D.b);

	if(!io)
	{
		/* ================= sample waves: filter of line r - 2, then the samples of line r ================= */
		const hvk_packed_taps_t notch = { { 0 } };
		/* this lane's share of the tap matrix, 16 rows (8 outputs x I, Q) by 64 window positions */
		int4v a_hh = Q.mfma_a[t & 63], a_hl = Q.mfma_a[64 + (t & 63)];
		touch4(a_hh); touch4(a_hl);

		/* what line r's samples are made from besides pixels is fetched a line ahead */
		fused_prep_t q = fused_prep_lanes<NT, VF, WC>(k, D, l0 - 1, l0, l1);
		hvk_side_t sd;
		int c[SPL];
		raster_load_side<NT, WC, 1>(k, P, q.L, t, sd, c);

		TS_DECL;
		for(int it = 0; it < niter; it++)
		{
			const int r = l0 - 1 + it;
			const bool do_r = r <= l1;
			const int rs = it & 3;                  /* plane buffer of line r; of line r - 2: (rs + 2) & 3 */
			TS_MARK(0);
			FUSED_BARRIER();
			TS_MARK(1);

			/* ---- line r - 2 through the matrix unit: planes -> output exchange ---- */
			if(!ABLATE(1024)) fused_filter<VF>(l, Q, a_hh, a_hl, t, (rs + 2) & 3, l.outl + (it & 1) * l.ostride);

			/* ---- 8 samples per lane of line r (its pixels' levels were staged during the last round) -> planes ---- */
			touch4(sd.base); touch4(sd.bwin);
#pragma unroll
			for(int i = 0; i < SPL; i++) touch(c[i]);
			const int part = q.part;
			/* does this wave hold samples of line r that are wanted? (of the lines before and after the run only the
			 * tail / the head is) */
			const int wx0 = __builtin_amdgcn_readfirstlane(x0);
			const bool mine = part == 0 || (part == 1 ? wx0 + 64 * SPL > W - FEDGE : wx0 < FEDGE);
			if(mine)
			{
				int s[SPL], cq[SPL];
				if(ABLATE(512))
				{
#pragma unroll
					for(int i = 0; i < SPL; i++) s[i] = c[i];
				}
				else raster_compute<NT, 0, 0, 0, WC>(k, P, q.L, ctaps, notch, y, r + 1, t, nth, l.rlds + (it & 1) * l.rstride, sd, c, s, cq);
				if(q.L.zero)
				{
#pragma unroll
					for(int i = 0; i < SPL; i++) s[i] = 0;
				}
				if(do_r && !ABLATE(65536)) fused_put_planes<WC>(l, s, x0, W, rs, part, 4);
			}

			/* ---- the next line's base line, burst window, sub-carrier phasors go out ---- */
			q = fused_prep_lanes<NT, VF, WC>(k, D, min(r + 1, l1), l0, l1);
			raster_load_side<NT, WC, 1>(k, P, q.L, t, sd, c);
			TS_MARK(2);
		}
		TS_DUMP(0);
	}
	else
	{
		/* ================= io waves: pixels in (line r + 1), sound and samples out (line r - 3) ================= */
		constexpr int H = NT / 2;
		const int YL = raster_YL(W), CL = raster_CL(W);
		if(k.has_nicam) fused_stage_tapd(l, Q.nicam_tapd, t, nth);
		const bool whole = WC || x0 + SPL <= W;
		const int ccl = k.has_nicam ? k.nicam_cc_len : 0x40000000;     /* (no NICAM: the mixer position stays where the zeros are) */

		/* the samples the reference reads past its chroma buffer: the same on every line */
		int ghost_u = 0, ghost_v = 0;
		if(NT > 1)
		{
			const int gt = t < H ? t : H - 1;
			ghost_u = P.ghost[2 * gt + 0];
			ghost_v = P.ghost[2 * gt + 1];
		}

		/* Two lines ahead: the source row of line r + 3 is on its way while the levels of line r + 2 are looked up
		 * and line r + 1 is staged. Before the first round: line l0 - 1 into the first staging area. */
		uint32_t rgb[HVK_PIX_PASSES];
		short4v px[HVK_PIX_PASSES];
		fused_prep_t q0 = fused_prep_lanes<NT, VF, WC>(k, D, l0 - 1, l0, l1);
		fused_prep_t q1 = fused_prep_lanes<NT, VF, WC>(k, D, l0, l0, l1);
		fused_prep_t q2 = fused_prep_lanes<NT, VF, WC>(k, D, min(l0 + 1, l1), l0, l1);
		raster_load_rgb<HVK_PIX_PASSES>(k, P, q0.L, t, nth, rgb);
		raster_gather<LV, HVK_PIX_PASSES, 1>(k, P, q0.L, rgb, px);
		raster_load_rgb<HVK_PIX_PASSES>(k, P, q1.L, t, nth, rgb);
		touch(ghost_u); touch(ghost_v);
		if(!q0.L.zero && !ABLATE(4096))
		{
			const int zf = q0.part == 1 ? max(0, W - FEDGE - 2 * HVK_CHROMA_LEAD) : 0;
			raster_stage<NT, WC, HVK_PIX_PASSES, 1>(k, q0.L, t, nth, px, ghost_u, ghost_v, l.rlds, l.rlds + YL, l.rlds + YL + CL, zf);
		}
		raster_gather<LV, HVK_PIX_PASSES, 1>(k, P, q1.L, rgb, px);
		raster_load_rgb<HVK_PIX_PASSES>(k, P, q2.L, t, nth, rgb);
		q0 = q1;            /* q0: the line staged next (r + 1), q1: the line looked up next (r + 2) */
		q1 = q2;

		/* one line ahead: symbol row, carrier samples, mixer row; the first line's symbol table */
		const int *const car_base = Q.carriers + (size_t) y * Q.car_frame + xc;
		const int *const sym_base = Q.tilesyms + (size_t) y * Q.sym_frame + (t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1);
		int cc_line = 0;
		if(k.has_nicam) cc_line = __builtin_amdgcn_readfirstlane(Q.tilesyms[(size_t) y * Q.sym_frame + (size_t) l0 * Q.sym_line + HVK_NICAM_SYMS]);
		/* (every round tabulates the symbols of the line that goes out in the NEXT round: the first rounds send nothing out,
		 * and the round before the first line goes out tabulates it) */
		int4u mix_a0, mix_a1, car0, car1;
		{
			int cp = cc_line + xc;
			if(cp >= ccl) cp -= ccl;
			mix_a0 = ((const int4u *) (Q.nicam_cca + cp))[0]; mix_a1 = ((const int4u *) (Q.nicam_cca + cp))[1];
			const int4u *cq = (const int4u *) (car_base + (unsigned) (l0 * Q.car_line));
			car0 = cq[0];
			car1 = cq[1];
		}

		TS_DECL;
		for(int it = 0; it < niter; it++)
		{
			const int r = l0 - 1 + it;
			const int fo = r - 3;                   /* the line that goes out */
			const bool do_s = r + 1 <= l1;
			const bool do_f = fo >= l0 && fo < l1;
			const int n0 = fo * W;                  /* its first output sample, frame local */
			int16_t *const Yb = l.rlds + ((it + 1) & 1) * l.rstride, *const U = Yb + YL, *const V = U + CL;

			TS_MARK(0);
			FUSED_BARRIER();
			TS_MARK(1);

			/* the symbols of the line that goes out next: in at the end of the round, tabulated for the next one */
			const int fn = min(max(fo + 1, l0), l1 - 1);
			const int symn = sym_base[(unsigned) (fn * Q.sym_line)];

			/* ---- line r + 1's pixel levels (looked up during the last round) into the staging area the sample waves
			 * read next round, with the zeros around them ---- */
#pragma unroll
			for(int i = 0; i < HVK_PIX_PASSES; i++) touch(px[i]);
			if(do_s && !q0.L.zero && !ABLATE(4096))
			{
				/* (of the line before the run only the tail is wanted: no zeros in front of what its chroma low pass reaches) */
				const int zf = q0.part == 1 ? max(0, W - FEDGE - 2 * HVK_CHROMA_LEAD) : 0;
				raster_stage<NT, WC, HVK_PIX_PASSES, 1>(k, q0.L, t, nth, px, ghost_u, ghost_v, Yb, U, V, zf);
			}

			/* ---- look-ups of line r + 2 (its colours came in during the last round), source row of line r + 3 ---- */
			touch4(car0); touch4(car1); touch4(mix_a0); touch4(mix_a1);
#pragma unroll
			for(int i = 0; i < HVK_PIX_PASSES; i++) touch(rgb[i]);
			if(!ABLATE(4096)) raster_gather<LV, HVK_PIX_PASSES, 1>(k, P, q1.L, rgb, px);
			q2 = fused_prep_lanes<NT, VF, WC>(k, D, min(r + 3, l1), l0, l1);
			raster_load_rgb<HVK_PIX_PASSES>(k, P, q2.L, t, nth, rgb);

			/* ---- sound onto the filtered samples of line fo (the sample waves left them last round), and out ---- */
			int o[SPL];                             /* packed (I, Q) int16 */
			const int *ol = l.outl + ((it + 1) & 1) * l.ostride + x0;
			const int4v oa = ((const int4v *) ol)[0], ob = ((const int4v *) ol)[1];
			o[0] = oa.x; o[1] = oa.y; o[2] = oa.z; o[3] = oa.w;
			o[4] = ob.x; o[5] = ob.y; o[6] = ob.z; o[7] = ob.w;
			fused_finish<WC>(k, l, o, x0, W, whole, car0, car1, NULL, mix_a0, mix_a1,
			                 Q.iq + (size_t) y * Q.out_stride * FS + n0 + x0, do_f, it & 1);

			/* the next line's symbol table (in the table the next round reads), mixer row, carrier samples: they have a
			 * round to arrive */
			{
				if(k.has_nicam) fused_symtab(l, t, symn, fn * W, W, (it + 1) & 1);
				/* the mixer position moves on by a line */
				if(fo + 1 > l0 && fo + 1 < l1)
				{
					cc_line += W;
					if(ccl >= W) { if(cc_line >= ccl) cc_line -= ccl; }
					else cc_line %= ccl;
				}
				int cp = cc_line + xc;
				if(ccl >= nth * SPL) { if(cp >= ccl) cp -= ccl; }
				else cp %= ccl;
				mix_a0 = ((const int4u *) (Q.nicam_cca + cp))[0]; mix_a1 = ((const int4u *) (Q.nicam_cca + cp))[1];
				const int4u *cq = (const int4u *) (car_base + (unsigned) (fn * Q.car_line));
				if(!ABLATE(8192)) { car0 = cq[0]; car1 = cq[1]; }
			}
			TS_MARK(2);
			q0 = q1;
			q1 = q2;
		}
		TS_DUMP(1);
	}
}

/* ------------------------------------------------------------------ */
/* both jobs in every wave                                              */

template<int NT, int VF, int EXTRAS, int WC, int LV>
__global__ __launch_bounds__(256, 4)
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
	int l0, l1;
	fused_run(k.lines, Q.nruns, l0, l1);
	if(l1 <= l0) return;
	const int FS = k.frame_samples;
	const hvk_packed_taps_t notch = { { 0 } };  /* (SECAM keeps the two-kernel path) */
	const fused_lds_t l = fused_lds(lds_raw, W, nth, VF, k.has_nicam ? k.nicam_ntaps : 0);
	const int YL = raster_YL(W), CL = raster_CL(W);
	int16_t *const Yb = l.rlds, *const U = l.rlds + YL, *const V = l.rlds + YL + CL;

	/* (the first barrier of the loop comes before the pulse table's first use) */
	if(k.has_nicam) fused_stage_tapd(l, Q.nicam_tapd, t, nth);

	/* this lane's share of the tap matrix, 16 rows (8 outputs x I, Q) by 64 window positions */
	int4v a_hh = { 0, 0, 0, 0 }, a_hl = { 0, 0, 0, 0 };
	if(VF)
	{
		a_hh = Q.mfma_a[t & 63];
		a_hl = Q.mfma_a[64 + (t & 63)];
	}

	/* raster line r = l0 - 1 + it (VF = 0: l0 + it), filtered line fl = r - 2 (VF = 0: r). The source row
	 * of line r + 1 is fetched while line r is worked on. */
	const int niter = VF ? (l1 - l0) + 3 : (l1 - l0);
	const int rlast = VF ? l1 : l1 - 1;         /* last line rasterised */
	uint32_t rgb[HVK_PIX_PASSES];
	fused_prep_t nx = fused_prep<NT, VF, EXTRAS, WC>(k, P, y, VF ? l0 - 1 : l0, l0, l1, first_frame, frame_stride);
	if(!nx.L.zero)
	{
		if(nx.part) raster_load_rgb<1>(k, P, nx.L, t, nth, rgb);
		else raster_load_rgb<HVK_PIX_PASSES>(k, P, nx.L, t, nth, rgb);
	}

	for(int it = 0; it < niter; it++)
	{
		const int r = VF ? l0 - 1 + it : l0 + it;
		const int fl = VF ? r - 2 : r;
		const bool do_r = r <= rlast;
		const bool do_f = fl >= l0 && fl < l1;
		const int rs = it % 3;                  /* plane buffer of line r; of line fl: (rs + 1) % 3 */
		const fused_prep_t cur = nx;
		const hvk_line_t &L = cur.L;
		const int part = cur.part;
		const bool draw = do_r && !L.zero;

		/* ---- line r: level look-ups of its pixels (their colours came in during the last round); the
		 * source row of line r + 1; what else its samples are made from ---- */
		short4v px[HVK_PIX_PASSES];
		hvk_side_t sd;
		int c[SPL];
		if(draw && !ABLATE(4096))
		{
			if(part) raster_gather<LV, 1>(k, P, L, rgb, px);
			else raster_gather<LV, HVK_PIX_PASSES>(k, P, L, rgb, px);
		}
		if(r + 1 <= rlast)
		{
			nx = fused_prep<NT, VF, EXTRAS, WC>(k, P, y, r + 1, l0, l1, first_frame, frame_stride);
			if(nx.part) raster_load_rgb<1>(k, P, nx.L, t, nth, rgb);
			else raster_load_rgb<HVK_PIX_PASSES>(k, P, nx.L, t, nth, rgb);
		}
		if(draw) raster_load_side<NT, WC>(k, P, L, t, sd, c);

		/* ---- line fl: symbol row, carrier samples ---- */
		const int n0 = fl * W;                  /* first output sample of the line, frame local */
		int symv = 0, cc_line = 0;
		int4u car0 = { 0, 0, 0, 0 }, car1 = { 0, 0, 0, 0 };
		const bool whole = WC || x0 + SPL <= W;
		if(do_f)
		{
			if(k.has_nicam)
			{
				/* one dense row per line, prepared by the host: HVK_NICAM_SYMS symbol words then the mixer
				 * position of the line's first sample */
				const int *row = Q.tilesyms + ((size_t) y * k.lines + fl) * HVK_NICAM_ROW;
				cc_line = row[HVK_NICAM_SYMS];
				symv = row[t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1];
			}
			if(k.has_carriers && whole && !ABLATE(8192))
			{
				const int4u *cp = (const int4u *) (Q.carriers + (size_t) y * FS + n0 + x0);
				car0 = cp[0];
				car1 = cp[1];
			}
		}

		/* does this wave hold samples of line r that are wanted? */
		const int wx0 = __builtin_amdgcn_readfirstlane(x0);
		const bool mine = part == 0 || (part == 1 ? wx0 + 64 * SPL > W - FEDGE : wx0 < FEDGE);

		/* ---- line fl through the matrix unit, while the look-ups of line r are on their way ---- */
		if(VF && do_f && !ABLATE(1024)) fused_filter<VF ? VF : 1>(l, Q, a_hh, a_hl, t, (rs + 1) % 3);

		/* ---- pixels of line r into the staging area (with the zeros around them: no clearing pass) ---- */
		if(draw && !ABLATE(4096))
		{
			const int zf = part == 1 ? max(0, W - FEDGE - 2 * HVK_CHROMA_LEAD) : 0;
			if(part) raster_stage<NT, WC, 1, 1>(k, L, t, nth, px, sd.ghost_u, sd.ghost_v, Yb, U, V, zf);
			else raster_stage<NT, WC, HVK_PIX_PASSES, 1>(k, L, t, nth, px, sd.ghost_u, sd.ghost_v, Yb, U, V, zf);
		}

		/* the symbol table of the line */
		if(do_f && k.has_nicam) fused_symtab(l, t, symv, n0, W);
		FUSED_BARRIER();

		/* ---- 8 samples per lane of line r; 8 outputs per lane of line fl ---- */
		/* the mixer row (i, -q) of this lane's samples (a small table: it arrives while line r is computed) */
		int4u mix_a0 = { 0, 0, 0, 0 }, mix_a1 = { 0, 0, 0, 0 };
		if(do_f && k.has_nicam)
		{
			const int cp = fused_mix_pos(k, cc_line, x0, nth);
			mix_a0 = ((const int4u *) (Q.nicam_cca + cp))[0]; mix_a1 = ((const int4u *) (Q.nicam_cca + cp))[1];
		}
		int s[SPL], cq[SPL];
		if(do_r && mine)
		{
			if(L.zero || ABLATE(512))
			{
#pragma unroll
				for(int i = 0; i < SPL; i++) s[i] = ABLATE(512) ? c[i] : 0;
			}
			else raster_compute<NT, 0, 0, EXTRAS, WC>(k, P, L, ctaps, notch, y, r + 1, t, nth, l.rlds, sd, c, s, cq);
			if(VF) fused_put_planes<WC>(l, s, x0, W, rs, part);
		}

		if(do_f)
		{
			int o[SPL];                             /* packed (I, Q) int16 */
			if(VF)
			{
				const int4v oa = ((const int4v *) (l.outl + x0))[0], ob = ((const int4v *) (l.outl + x0))[1];
				o[0] = oa.x; o[1] = oa.y; o[2] = oa.z; o[3] = oa.w;
				o[4] = ob.x; o[5] = ob.y; o[6] = ob.z; o[7] = ob.w;
			}
			else
			{
				/* no filter: the raster goes straight to I, Q = 0 */
#pragma unroll
				for(int i = 0; i < SPL; i++) o[i] = s[i] & 0xFFFF;
			}
			fused_finish<WC>(k, l, o, x0, W, whole, car0, car1, Q.carriers + (size_t) y * FS + n0 + x0, mix_a0, mix_a1,
			                 Q.iq + (size_t) y * Q.out_stride * FS + n0 + x0);
		}
		FUSED_BARRIER();
	}
}

/* ------------------------------------------------------------------ */

extern "C" void hvk_raster_ptrs(const hvk_raster_args_t *a, hvk_rptrs_t *P);

extern "C" size_t hvk_fused_lds_bytes(int width, int vf, int nicam_ntaps, int ws)
{
	const int nth = ((width + SPL - 1) / SPL + 63) / 64 * 64;
	int a, b, c, d, e, f;
	return(fused_lds_layout(width, nth, vf, nicam_ntaps, ws, &a, &b, &c, &d, &e, &f));
}

template<int NT, int VF, int EXTRAS, int WC, int LV>
static int _launch_fused4(const hvk_raster_args_t *ra, const hvk_filter_args_t *fa, int run_lines, hipStream_t stream)
{
	/* raster waves and filter waves (hvk_k_fusedw) wherever the raster has no barriers of its own and there
	 * is a filter to hand over to; HVK_NO_WAVE_ROLES=1 keeps both jobs in every wave (tests run both) */
	constexpr bool CAN_WS = VF != 0 && !EXTRAS;
	static const bool no_ws = getenv("HVK_NO_WAVE_ROLES") != NULL;
	const bool ws = CAN_WS && !no_ws;
	const int W = ra->k.width;
	const int nth = ((W + SPL - 1) / SPL + 63) / 64 * 64;
	const int threads = ws ? 2 * nth : nth;
	const size_t lds = hvk_fused_lds_bytes(W, VF, ra->k.has_nicam ? ra->k.nicam_ntaps : 0, ws);
	const void *fn = (const void *) hvk_k_fused<NT, VF, EXTRAS, WC, LV>;
	if constexpr(CAN_WS) { if(ws) fn = (const void *) hvk_k_fusedw<NT, VF, WC, LV>; }
	if(threads > 512) return(HVK_UNSUPPORTED);

	/* runs per frame: a multiple of 8 (one XCD's share each); run_lines < 0: -run_lines runs, as given */
	int runs;
	if(run_lines < 0) runs = -run_lines;
	else
	{
		/* Each run rasterises (the edges of) two lines more than it puts out and spends three rounds filling
		 * and draining its pipeline: short runs pay that more often, long runs leave the last round of
		 * workgroups half empty. Estimate both from the number of workgroups the chip holds at once. */
		int per_cu = 0, cus = 256;
		hipDeviceProp_t prop;
		int dev = 0;
		(void) hipGetDevice(&dev);
		if(hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
		if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 4;
		const double slots = (double) per_cu * cus;
		const double over = EXTRAS ? 2.0 : (ws ? 3.0 : 1.0);    /* in lines' worth of time */
		double best = 1e30;
		runs = 8;
		for(int n = 8; n <= ra->k.lines / 4; n += 8)
		{
			const double rounds = (double) n * ra->nframes / slots;
			const double cost = (rounds < 1 ? 1 : __builtin_ceil(rounds - 0.02)) * ((double) ra->k.lines / n + over);
			if(cost < best - 1e-9) { best = cost; runs = n; }
		}
	}
	if(runs < 1) runs = 1;
	/* (the waves keep a run's line descriptors in the lanes of a register: 64 lines at most, the two beside the run included) */
	while(ws && (ra->k.lines + runs - 1) / runs + 3 > 64) runs += 8;
	if(runs > ra->k.lines) runs = ra->k.lines;
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
	Q.nruns = runs;
	Q.car_frame = Q.car_line = Q.sym_frame = Q.sym_line = 0;
	Q.lstate = NULL;
	if constexpr(CAN_WS)
	{
		if(ws)
		{
			/* this kernel loads without asking: what a configuration lacks reads as a block of zeros */
			if(!fa->zeros) return(HVK_ERROR);
			if(ra->k.has_carriers) { Q.car_frame = ra->k.frame_samples; Q.car_line = W; }
			else Q.carriers = (const int *) fa->zeros;
			if(ra->k.has_nicam) { Q.sym_frame = ra->k.lines * HVK_NICAM_ROW; Q.sym_line = HVK_NICAM_ROW; }
			else { Q.tilesyms = (const int *) fa->zeros; Q.nicam_cca = (const int *) fa->zeros; }
			if(!fa->lstate) return(HVK_ERROR);
			Q.lstate = fa->lstate;
			hipLaunchKernelGGL(hvk_k_linestate, dim3((ra->k.lines + 2 + 63) / 64, ra->nframes), dim3(64), 0, stream,
			                   ra->k, P, (fused_rec_t *) fa->lstate, ra->first_frame, ra->frame_stride);
			hipLaunchKernelGGL((hvk_k_fusedw<NT, VF, WC, LV>), dim3(runs, ra->nframes), dim3(threads), lds, stream,
			                   ra->k, ra->ctaps, P, Q, ra->first_frame, ra->frame_stride);
			return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
		}
	}
	if(threads > 256) return(HVK_UNSUPPORTED);
	hipLaunchKernelGGL((hvk_k_fused<NT, VF, EXTRAS, WC, LV>), dim3(runs, ra->nframes), dim3(threads), lds, stream,
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
	if(!hvk_fused_supported(&ra->k, fa->mfma_a)) return(HVK_UNSUPPORTED);
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
