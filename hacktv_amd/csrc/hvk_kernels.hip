/* hvk_kernels.hip -- CDNA4 (gfx950) kernels of the composite-video -> IQ engine.
 *
 * All integer except the one-off table expansion. The two kernels of every
 * render are hvk_k_raster and hvk_k_filter; the others serve options:
 *
 *   hvk_k_expand_yuv   once per engine: expands the 2^24-entry RGB -> (Y,U,V)
 *                      level table in HBM from 256 gamma values and a handful
 *                      of doubles, with FP contraction off so that every
 *                      entry equals the reference's (src/video.c:3912-3958).
 *
 *   hvk_k_raster       one workgroup per scanline, 8 consecutive samples per
 *                      lane. Builds the final raster (luma from the frame,
 *                      sync pulses incl. the leading edge of the NEXT line's
 *                      pulse, chroma U/V through the 13-tap zero-history FIR
 *                      staged in LDS, burst, QAM onto the sub-carrier) --
 *                      _vid_next_line_raster, src/video.c:2864-3066 -- and
 *                      writes it as int16, 16 bytes per lane.
 *
 *   hvk_k_filter       two waves per 1024 output samples (= one PAL line at 16 Msps),
 *                      four such tiles per workgroup, 8 consecutive outputs per
 *                      lane. The 51-tap real->complex VSB filter (or real
 *                      low-pass) is a banded matrix product on the int8 matrix
 *                      unit -- taps and samples split into bytes, four
 *                      v_mfma_i32_16x16x64_i8 per 128 outputs recombined exactly
 *                      modulo 2^32 (src/fir.c:564-615, :304-355); taps the split
 *                      cannot express keep the v_dot2c_i32_i16 form. Adds the
 *                      serial-carrier side stream and the NICAM DQPSK signal
 *                      (pulse overlap-add + mixer, src/nicam728.c:342-411),
 *                      and stores interleaved int16 I/Q, 32 bytes per lane.
 *
 *   hvk_k_resample     --pixelrate: rational poly-phase FIR between the two,
 *                      pixel-rate raster -> sample-rate stream (src/fir.c:304-355
 *                      with interpolation / decimation).
 *   hvk_k_tail         --swap-iq / --offset / --passthru, in place on the output
 *                      (src/video.c:3466-3541).
 *   hvk_k_convert      the file sink's sample formats (src/rf_file.c:34-277).
 *
 * Inside hvk_k_raster, behind wave-uniform tests that cost the plain path a
 * scalar compare each: SECAM (luma notch + the FM sub-carrier stream of hvk_secam.hip),
 * insertion test signals, and the VBI data lines (teletext, WSS, VITC, CC608
 * symbols from one table store; anti-copy pulse runs) listed per frame by the
 * host. S-Video has kernel variants of its own (template parameter).
 *
 * The one dense contraction is the video filter. The raster kernel is bound by VALU
 * issue, the filter kernel by its HBM streams (2 + 4 B/sample read, 4 B/sample
 * written) and the NICAM stage.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "hvk_internal.h"
#include "hvk_kernels.h"

#include "hvk_device.h"

#define HVK_FILTER_GROUP 4    /* filter tiles a workgroup works on side by side (two waves each): they share the staged NICAM pulse table */
#define HVK_TILES_PER_WG 1    /* consecutive filter tiles walked by one workgroup (4 measured 10 % slower: fewer independent workgroups to overlap) */

__device__ __forceinline__ int floordiv(int a, int b) { int q = a / b; return((a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q); }

/* ------------------------------------------------------------------ */

__global__ void hvk_k_expand_yuv(short4v *lut, const hvk_yuvparams_t *pp)
{
	unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
	if(c > 0xFFFFFFu) return;
	lut[c] = level_of(c, *pp);
}

/* every colour's levels by the arithmetic `pp` says (the short form) against the table: how many differ */
template<int FAST>
__global__ void hvk_k_check_levels(const short4v *lut, const hvk_yuvparams_t *pp, int *differ)
{
	unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
	if(c > 0xFFFFFFu) return;
	const short4v a = level_of<-1, FAST>(c, *pp), b = lut[c];
	if(a.x != b.x || a.y != b.y || a.z != b.z) atomicAdd(differ, 1);
}

/* ------------------------------------------------------------------ */

/* One workgroup per scanline of the slab (hvk_device.h has the steps): the samples go to the raster
 * slab in HBM, 16 bytes per lane. */
template<int NT, int SECAM, int SV, int EXTRAS, int WC, int LV>
__global__ __launch_bounds__(1024)
void hvk_k_raster(const hvk_kconst_t k,
                  const hvk_packed_taps_t ctaps,
                  const hvk_packed_taps_t notch,        /* SECAM luma notch, 51 taps */
                  const hvk_rptrs_t P,
                  int16_t *__restrict__ S,
                  int16_t *__restrict__ Cq,             /* --s-video: the sub-carrier alone, same slab geometry as S */
                  const int64_t first_frame,            /* frame y of the batch is stream frame first_frame + y * frame_stride */
                  const int64_t frame_stride,
                  const int16_t *__restrict__ linelist, /* only these lines of every frame, row after row (hvk_raster_args_t.linelist); NULL: the slab */
                  const int nlist)
{
	extern __shared__ __attribute__((aligned(16))) int16_t lds[];

	/* the grid's x extent is padded to a multiple of 8 so that, with workgroups
	 * dealt round-robin to the 8 XCDs, line x of EVERY frame runs on XCD x % 8:
	 * the slices of the colour table and of the source frame a line needs then
	 * stay in that XCD's L2 from frame to frame */
	if((int) blockIdx.x >= (linelist ? nlist : k.slab_lines)) return;

	/* WC: the line width when it is known at compile time (1024: PAL at 16 Msps) -- every lane then
	 * holds 8 samples inside the line and the per-sample range tests fold away */
	const int W = WC ? WC : k.width;
	const int t = threadIdx.x;
	if(WC) __builtin_assume(t * SPL + SPL <= WC);
	const int nth = blockDim.x;
	const int x0 = t * SPL;
	const int rel = linelist ? (int) linelist[blockIdx.x] : (int) blockIdx.x - 1;       /* line of the frame; -1 and `lines` (and `lines` + 1 with the resampler) are halo lines */
	int16_t *out = S + ((size_t) blockIdx.y * (linelist ? nlist : k.slab_lines) + blockIdx.x) * W;

	const hvk_line_t L = raster_setup<SECAM, EXTRAS>(k, P, (int) blockIdx.y, rel, first_frame, frame_stride);

	if(L.zero)
	{
		for(int i = 0; i < SPL; i++) if(x0 + i < W) out[x0 + i] = 0;
		return;
	}

	const int YL = raster_YL(W), CL = raster_CL(W);
	int16_t *Yb = lds, *U = lds + YL, *V = lds + YL + CL;

	uint32_t rgb[HVK_PIX_PASSES];
	hvk_side_t sd;
	int c[SPL];
	raster_loads<NT, WC>(k, P, L, t, nth, rgb, sd, c);

	if(L.pal)
	{
		raster_clear(L, t, nth, U, CL);
		__syncthreads();
	}
	raster_pixels<NT, WC, LV>(k, P, L, t, nth, rgb, sd.ghost_u, sd.ghost_v, Yb, U, V);
	if(L.pal || L.has_pix) __syncthreads();

	int s[SPL], cq[SPL];
	raster_compute<NT, SECAM, SV, EXTRAS, WC>(k, P, L, ctaps, notch, (int) blockIdx.y, rel + 1, t, nth, lds, sd, c, s, cq);

	if(ABLATE(128) && s[0] != 12345) return;   /* profiling: no store */
	if(x0 + SPL <= W)
	{
		int4v o;
		o.x = (s[0] & 0xFFFF) | (s[1] << 16);
		o.y = (s[2] & 0xFFFF) | (s[3] << 16);
		o.z = (s[4] & 0xFFFF) | (s[5] << 16);
		o.w = (s[6] & 0xFFFF) | (s[7] << 16);
		*(int4u *) (out + x0) = (int4u) { o.x, o.y, o.z, o.w };
	}
	else
	{
		for(int i = 0; i < SPL; i++) if(x0 + i < W) out[x0 + i] = (int16_t) s[i];
	}

	if(SV)
	{
		int16_t *oc = Cq + ((size_t) blockIdx.y * k.slab_lines + blockIdx.x) * W;
		for(int i = 0; i < SPL; i++) if(x0 + i < W) oc[x0 + i] = (int16_t) cq[i];
	}
}

/* ------------------------------------------------------------------ */


template<int NT, int VF, int SV, int EXACT, int MF>
__global__ __launch_bounds__(HVK_TILE / HVK_SPL * HVK_FILTER_GROUP, MF ? 8 : 1)
void hvk_k_filter(const hvk_kconst_t k,
                  const hvk_packed_taps_t itaps,
                  const hvk_packed_taps_t qtaps,
                  const hvk_framedesc_t *__restrict__ fdesc,
                  const int16_t *__restrict__ S,
                  const int *__restrict__ carriers,      /* [frames][frame_samples] int16 pairs */
                  const int *__restrict__ tilesyms,      /* [frames][tiles][HVK_NICAM_ROW]: symbols (start << 3 | valid << 2 | dsym), mixer position */
                  const int *__restrict__ nicam_tapd,    /* pulse taps: HVK_NICAM_COPIES shifted int16 copies, zero padded (hvk_engine.cpp) */
                  const int *__restrict__ nicam_cca,     /* mixer (i, -q), 8 entries past the wrap */
                  const int16_t *__restrict__ Cq,        /* --s-video: the Q channel, laid out like S */
                  const int4v *__restrict__ mfma_a,      /* MF: the taps as A operand, [hh, hl][lane] (hvk_engine.cpp:_mfma_taps) */
                  const int mfma_ci, const int mfma_cq,  /* MF: 128 * sum of the taps */
                  int *__restrict__ iq,                  /* [frames * out_stride][frame_samples] int16 pairs */
                  const int64_t out_stride,
                  const int tiles)                       /* 1024-sample tiles per frame */
{
	constexpr int H = NT / 2;
	constexpr int LEAD = H + (H & 1);           /* window lead, even */
	constexpr int NWIN = HVK_TILE + 2 * LEAD + 16;
	constexpr int NPL = HVK_TILE + 64;          /* MF: window as two byte planes; position 0 is sample n0 - LEAD */
	static_assert(!MF || (NT == 51 && HVK_TILE / HVK_SPL == 128), "the MFMA filter is laid out for 51 taps and two waves per tile");
	constexpr int G = HVK_FILTER_GROUP;
	__shared__ __attribute__((aligned(16))) int16_t win_g[G][MF ? 8 : NWIN];
	__shared__ __attribute__((aligned(16))) unsigned char xh_g[G][MF ? NPL : 16], xl_g[G][MF ? NPL : 16];
	__shared__ __attribute__((aligned(16))) int outl_g[G][MF ? HVK_TILE : 4];
	__shared__ __attribute__((aligned(16))) int16_t tapd[HVK_NICAM_COPIES * HVK_NICAM_TAPD];   /* HVK_NICAM_COPIES copies of the pulse, copy s one entry further left */
	__shared__ int sym_st_g[G][HVK_NICAM_SYMS];                               /* start, relative to the tile's first sample */
	__shared__ __attribute__((aligned(16))) int4v sym_ent_g[1 + G * HVK_NICAM_SYMS];   /* nicam_symbol_slot(); an entry of slack in front: nicam_add() */

	const int FS = k.frame_samples;
	const int sub = threadIdx.x / (HVK_TILE / HVK_SPL);        /* which of the workgroup's tiles */
	const int t = threadIdx.x % (HVK_TILE / HVK_SPL);
	const int x0 = t * SPL;
	const int16_t *slab = S + (size_t) blockIdx.y * k.s_stride + k.s_lead;    /* frame local sample 0 */
	int16_t *const win = win_g[sub];
	unsigned char *const xh = xh_g[sub], *const xl = xl_g[sub];
	int *const outl = outl_g[sub];
	int *const sym_st = sym_st_g[sub];
	int4v *const sym_ent = sym_ent_g + 1 + sub * HVK_NICAM_SYMS;
	(void) win; (void) xh; (void) xl; (void) outl;

	/* HVK_NICAM_COPIES copies of the NICAM pulse table (int16), copy s shifted left by s entries, so that
	 * any run of 8 entries starts aligned in one of them: one ds_read2_b64 (four copies), lanes side by
	 * side; staged once per workgroup. The load goes out here, the LDS write waits until the first
	 * tile's own loads are on their way. */
	static_assert(HVK_TILE / HVK_SPL * HVK_FILTER_GROUP >= HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8, "one pulse-table vector per thread");
	const bool tap_mine = k.has_nicam && !ABLATE(16) && (int) threadIdx.x < HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8;
	int4v tap_stage = { 0, 0, 0, 0 };
	if(k.has_nicam && !ABLATE(16)) tap_stage = ((const int4v *) nicam_tapd)[min((int) threadIdx.x, HVK_NICAM_COPIES * HVK_NICAM_TAPD / 8 - 1)];

	/* MF: this lane's share of the tap matrix, 16 rows (8 outputs x I, Q) by 64 window positions */
	int4v a_hh = { 0, 0, 0, 0 }, a_hl = { 0, 0, 0, 0 };
	if(MF)
	{
		a_hh = mfma_a[t & 63];
		a_hl = mfma_a[64 + (t & 63)];
	}

	/* a workgroup walks HVK_TILES_PER_WG consecutive tiles: its fixed costs (kernel
	 * arguments into SGPRs, the pulse table) are paid once */
	for(int it = 0; it < HVK_TILES_PER_WG; it++)
	{
	if((blockIdx.x * HVK_TILES_PER_WG + it) * G >= tiles) break;
	/* a group that reaches past the frame's last tile does that one again: same values to the same places */
	const int tile = min((blockIdx.x * HVK_TILES_PER_WG + it) * G + sub, tiles - 1);
	const int n0 = tile * HVK_TILE;             /* first output sample of the tile, frame local */

	/* stage raster samples [n0 - LEAD, n0 + TILE + LEAD) as dwords; the slab
	 * keeps one line before and one after the frame */
	constexpr int NG = NPL / 8;                 /* MF: groups of 8 window samples */
	constexpr int WP = (NG + HVK_TILE / HVK_SPL - 1) / (HVK_TILE / HVK_SPL);
	int4u wd[WP];

	/* ---- this tile's loads, in the order their values are needed; none under a lane test where
	 * it can be helped (such a load is waited for on the spot) ---- */
	int symv = 0, cc_tile = 0;
	if(k.has_nicam)
	{
		/* one dense row per tile, prepared by the host: HVK_NICAM_SYMS symbol words
		 * then the mixer position of the tile's first sample -- a single load
		 * that depends on nothing but the block index */
		const int *row = tilesyms + ((size_t) blockIdx.y * tiles + tile) * HVK_NICAM_ROW;
		cc_tile = row[HVK_NICAM_SYMS];
		symv = row[t < HVK_NICAM_SYMS ? t : HVK_NICAM_SYMS - 1];
	}
	if(VF != 0 && MF)
	{
		/* (2-byte aligned only: a line can have an odd number of samples -- 1135 at 4 f_sc -- and then every other
		 * frame's slab starts on an odd sample; global memory takes the dword loads all the same) */
		const int_a2 *src = (const int_a2 *) (slab + n0 - LEAD);
		const int limit = (k.s_stride - k.s_lead - (n0 - LEAD)) / 2;   /* dwords available in the slab */
#pragma unroll
		for(int i = 0; i < WP; i++)
		{
			const int q = min(t + i * (HVK_TILE / HVK_SPL), NG - 1);
			int4u d = { 0, 0, 0, 0 };
			if(EXACT) { const int4a2 w = ((const int4a2 *) src)[q]; d = (int4u) { w.x, w.y, w.z, w.w }; }
			else if(q * 4 + 3 < limit) { const int4a2 w = ((const int4a2 *) src)[q]; d = (int4u) { w.x, w.y, w.z, w.w }; }
			else
			{
				if(q * 4 + 0 < limit) d.x = src[q * 4 + 0];
				if(q * 4 + 1 < limit) d.y = src[q * 4 + 1];
				if(q * 4 + 2 < limit) d.z = src[q * 4 + 2];
			}
			wd[i] = d;
		}
	}
	/* FIN (the filter on the matrix unit, no S-Video): the samples are FINISHED in the lane the matrix unit leaves them in, as in
	 * hvk_k_direct -- NICAM goes through LDS to that lane, the carriers are read and the samples stored there, 8 bytes a lane and
	 * 512 contiguous bytes a wave instruction (hvk_direct.hip has the measurement: with 8 consecutive samples a lane every
	 * instruction touched half of every 32 bytes of a wave's 2 KB) */
	constexpr bool FIN = VF != 0 && MF && !SV;
	int2u cj[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
	/* the serial-carrier samples of this lane: needed last */
	const int nl = n0 + x0;
	int4u car0 = { 0, 0, 0, 0 }, car1 = { 0, 0, 0, 0 };
	if(FIN && k.has_carriers)
	{
		const int fl = t & 63, fg = fl >> 4, fc = fl & 15;
#pragma unroll
		for(int j = 0; j < 4; j++)
		{
			const int nn = n0 + ((t >> 6) * 64 + j * 16 + fc) * 8 + 2 * fg;
			cj[j] = __builtin_nontemporal_load((const int2u *) (carriers + (size_t) blockIdx.y * FS + ((EXACT || nn + 2 <= FS) ? nn : 0)));
		}
	}
	else if(k.has_carriers && (EXACT || nl + SPL <= FS))
	{
		const int4u *c = (const int4u *) (carriers + (size_t) blockIdx.y * FS + nl);
		car0 = __builtin_nontemporal_load(&c[0]);      /* streaming, like the stores below (hvk_direct.hip has the measurement) */
		car1 = __builtin_nontemporal_load(&c[1]);
	}

	if(it == 0 && tap_mine) ((int4v *) tapd)[threadIdx.x] = tap_stage;

	/* stage raster samples [n0 - LEAD, n0 + TILE + LEAD) as dwords; the slab
	 * keeps one line before and one after the frame */
	if(VF != 0 && MF)
	{
		/* The int8 matrix unit multiplies bytes: the window goes to LDS as a plane of high bytes
		 * (x >> 8, signed) and a plane of low bytes less 128 (x & 255, read as signed after ^ 0x80),
		 * eight samples per lane and pass, v_perm_b32 picking the bytes out of the sample pairs. */
#pragma unroll
		for(int i = 0; i < WP; i++)
		{
			const int q = t + i * (HVK_TILE / HVK_SPL);
			if(q < NG)
			{
				int2v ph, pl;
				split_planes(wd[i], ph, pl);
				((int2v *) xh)[q] = ph;
				((int2v *) xl)[q] = pl;
			}
		}
	}
	else if(VF != 0)
	{
		const int_a2 *src = (const int_a2 *) (slab + n0 - LEAD);   /* (2-byte aligned only, see above) */
		const int limit = (k.s_stride - k.s_lead - (n0 - LEAD)) / 2;   /* dwords available in the slab */
		constexpr int PASSES = (NWIN / 2 + HVK_TILE / HVK_SPL - 1) / (HVK_TILE / HVK_SPL);
		int v[PASSES];
#pragma unroll
		for(int i = 0; i < PASSES; i++)
		{
			const int q = t + i * (HVK_TILE / HVK_SPL);
			v[i] = (q < NWIN / 2 && (EXACT || q < limit)) ? src[q] : 0;
		}
#pragma unroll
		for(int i = 0; i < PASSES; i++)
		{
			const int q = t + i * (HVK_TILE / HVK_SPL);
			if(q < NWIN / 2) ((int *) win)[q] = v[i];
		}
	}

	if(k.has_nicam)
	{
		/* the symbols whose pulses can touch this tile, oldest first: start
		 * (relative to the tile's first sample) and sign pair. The schedule
		 * (src/nicam728.c:398-407) is tabulated per frame by the host. */
		if(t < HVK_NICAM_SYMS) nicam_symbol_slot(symv, n0, sym_st, sym_ent, t, tapd);
	}
	__syncthreads();

	const int n = n0 + x0;                      /* this lane's first output, frame local; lanes past the frame compute and store nothing */

	/* the mixer row (i, -q) of this lane's samples: on its way while the filter and the pulse sums run */
	int4u mix[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
	if(k.has_nicam && !ABLATE(64))
	{
		int cp = cc_tile + x0;                  /* mixer position of this lane's first sample */
		if(k.nicam_cc_len >= HVK_TILE) { if(cp >= k.nicam_cc_len) cp -= k.nicam_cc_len; }
		else cp %= k.nicam_cc_len;
		nicam_mix_rows(nicam_cca, k.nicam_cc_len + 8, cp, mix);
	}

	int o[SPL];                                 /* packed (I, Q) int16 */

	if(FIN)
	{
		const bool has_nic = k.has_nicam != 0, has_car = k.has_carriers != 0;
		if(has_nic)
		{
			/* NICAM on its own (the adds are modulo 2^16 per channel: their order is free), handed to the lanes that finish the samples */
			int nic[SPL] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			nicam_add(k, x0, sym_st, sym_ent, tapd, mix, nic);
			((int4v *) (outl + x0))[0] = (int4v) { nic[0], nic[1], nic[2], nic[3] };
			((int4v *) (outl + x0))[1] = (int4v) { nic[4], nic[5], nic[6], nic[7] };
		}
		/* (the exchange is within a wave: mfma_filter_each()) */
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		int *const frame_out = iq + (size_t) blockIdx.y * out_stride * FS;
		mfma_filter_each(xh, xl, t, a_hh, a_hl, mfma_ci, mfma_cq, [&](const int j, const int seg, const int g, const int2v pk)
		{
			int2v nv = { 0, 0 };
			if(has_nic) nv = *(const int2v *) (outl + seg * 8 + 2 * g);
			const int nn = n0 + seg * 8 + 2 * g;
			int2u ov;
			ov.x = pk_add16(pk_add16(pk.x, cj[j].x), nv.x);
			ov.y = pk_add16(pk_add16(pk.y, cj[j].y), nv.y);
			if(EXACT || nn + 2 <= FS) __builtin_nontemporal_store(ov, (int2u *) (frame_out + nn));
			else if(nn < FS) frame_out[nn] = pk_add16(pk_add16(pk.x, has_car ? carriers[(size_t) blockIdx.y * FS + nn] : 0), nv.x);
		});
		__syncthreads();                            /* the next tile re-uses the LDS window and symbol table */
		continue;
	}
	else if(VF != 0 && MF)
	{
		/* the FIR as a banded matrix product on the matrix unit (hvk_device.h), then through LDS to the lane that owns the 8 outputs */
		mfma_filter(xh, xl, outl, t, a_hh, a_hl, mfma_ci, mfma_cq);
		__syncthreads();
		const int4v oa = ((const int4v *) (outl + x0))[0], ob = ((const int4v *) (outl + x0))[1];
		o[0] = oa.x; o[1] = oa.y; o[2] = oa.z; o[3] = oa.w;
		o[4] = ob.x; o[5] = ob.y; o[6] = ob.z; o[7] = ob.w;
	}
	else if(VF != 0)
	{
		constexpr int ND = SPL / 2 + (NT + 1) / 2 + 1;
		int d[ND];
		int ai[SPL];
		const int4v *p = (const int4v *) (win + x0);
#pragma unroll
		for(int m = 0; m < (ND + 3) / 4; m++)
		{
			const int4v v = p[m];
			if(m * 4 + 0 < ND) d[m * 4 + 0] = v.x;
			if(m * 4 + 1 < ND) d[m * 4 + 1] = v.y;
			if(m * 4 + 2 < ND) d[m * 4 + 2] = v.z;
			if(m * 4 + 3 < ND) d[m * 4 + 3] = v.w;
		}

		/* output i is centred on window element LEAD + x0 + i: first tap at
		 * element LEAD - H + x0 + i */
		fir8<NT, LEAD - H>(d, itaps.p, ai);

		if(VF == 3)
		{
			/* >> 15 and clamp to int16 (src/fir.c:605-608): v_cvt_pk_i16_i32 saturates both halves */
			int aq[SPL];
			fir8<NT, LEAD - H>(d, qtaps.p, aq);
#pragma unroll
			for(int i = 0; i < SPL; i++) o[i] = sat_pack16(ai[i] >> 15, aq[i] >> 15);
		}
		else
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) o[i] = sat_pack16(ai[i] >> 15, 0);
		}
	}
	else
	{
		/* no filter: the raster goes straight to I, Q = 0 */
		const int16_t *p = slab + n;
#pragma unroll
		for(int i = 0; i < SPL; i++) o[i] = (EXACT || n + i < FS) ? ((int) p[i] & 0xFFFF) : 0;
	}

	if(SV)
	{
		/* S-Video: Q is the sub-carrier of the same sample position (the filter only delays the luma
		 * by the line its output slot is shifted by, src/video.c:3235-3248) */
		const int16_t *cp = Cq + (size_t) blockIdx.y * k.s_stride + k.s_lead + n;
#pragma unroll
		for(int i = 0; i < SPL; i++) if(n + i < FS) o[i] = (o[i] & 0xFFFF) | ((int) cp[i] << 16);
	}

	const size_t cbase = (size_t) blockIdx.y * FS + n;
	const size_t obase = (size_t) blockIdx.y * out_stride * FS + n;
	/* EXACT: the frame is a whole number of tiles (and the slab reaches a window's length past it):
	 * every lane's 8 outputs are inside the frame and the tail handling folds away */
	const bool whole = EXACT || n + SPL <= FS;

	/* serial carriers (FM / AM sound), computed on the host: a plain add of
	 * int16 pairs with wrap-around (src/video.c:3431-3432) */
	if(k.has_carriers)
	{
		if(whole)
		{
			o[0] = pk_add16(o[0], car0.x); o[1] = pk_add16(o[1], car0.y); o[2] = pk_add16(o[2], car0.z); o[3] = pk_add16(o[3], car0.w);
			o[4] = pk_add16(o[4], car1.x); o[5] = pk_add16(o[5], car1.y); o[6] = pk_add16(o[6], car1.z); o[7] = pk_add16(o[7], car1.w);
		}
		else
		{
			const int *c = carriers + cbase;
#pragma unroll
			for(int i = 0; i < SPL; i++) if(n + i < FS) o[i] = pk_add16(o[i], c[i]);
		}
	}

	/* NICAM: pulse sums of the symbols in flight, mixer, add (hvk_device.h; src/nicam728.c:350-365, :386-396) */
	if(k.has_nicam) nicam_add(k, x0, sym_st, sym_ent, tapd, mix, o);

	/* interleaved int16 I/Q, 32 bytes per lane */
	int *dst = iq + obase;
	if(whole)
	{
		__builtin_nontemporal_store(((int4u) { o[0], o[1], o[2], o[3] }), &((int4u *) dst)[0]);
		__builtin_nontemporal_store(((int4u) { o[4], o[5], o[6], o[7] }), &((int4u *) dst)[1]);
	}
	else
	{
#pragma unroll
		for(int i = 0; i < SPL; i++) if(n + i < FS) dst[i] = o[i];
	}

	__syncthreads();                            /* the next tile re-uses the LDS window and symbol table */
	}
}

/* ------------------------------------------------------------------ */

/* The file sink's sample formats (src/rf_file.c:34-277), two complex samples or
 * four real ones per lane; grid-stride over the range. */
template<int TYPE, int CPLX>
__global__ void hvk_k_convert(const int *__restrict__ iq, size_t count, void *__restrict__ dst)
{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for(size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
	{
		const int p = iq[i];
		const int vi = (int) (short) (p & 0xFFFF), vq = p >> 16;

		if(TYPE == HVK_UINT8)
		{
			const unsigned a = (unsigned) (vi + 32768) >> 8, b = (unsigned) (vq + 32768) >> 8;
			if(CPLX) ((uint16_t *) dst)[i] = (uint16_t) (a | (b << 8));
			else ((uint8_t *) dst)[i] = (uint8_t) a;
		}
		else if(TYPE == HVK_INT8)
		{
			const int a = vi >> 8, b = vq >> 8;
			if(CPLX) ((uint16_t *) dst)[i] = (uint16_t) ((a & 0xFF) | ((b & 0xFF) << 8));
			else ((int8_t *) dst)[i] = (int8_t) a;
		}
		else if(TYPE == HVK_UINT16)
		{
			const unsigned a = (unsigned) (vi + 32768) & 0xFFFF, b = (unsigned) (vq + 32768) & 0xFFFF;
			if(CPLX) ((unsigned *) dst)[i] = a | (b << 16);
			else ((uint16_t *) dst)[i] = (uint16_t) a;
		}
		else if(TYPE == HVK_INT16)
		{
			if(CPLX) ((int *) dst)[i] = p;
			else ((int16_t *) dst)[i] = (int16_t) vi;
		}
		else if(TYPE == HVK_INT32)
		{
			const int a = (int) ((unsigned) vi << 16) + vi, b = (int) ((unsigned) vq << 16) + vq;
			if(CPLX) ((int2v *) dst)[i] = (int2v) { a, b };
			else ((int *) dst)[i] = a;
		}
		else
		{
			/* (float) v * (1.0 / 32767.0): the product is formed in double */
			const float a = (float) ((double) (float) vi * (1.0 / 32767.0)), b = (float) ((double) (float) vq * (1.0 / 32767.0));
			if(CPLX) ((float2 *) dst)[i] = make_float2(a, b);
			else ((float *) dst)[i] = a;
		}
	}
}

template<int TYPE>
static int _launch_convert(const void *iq, size_t count, int cplx, void *dst, hipStream_t stream)
{
	const int blocks = (int) ((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
	if(cplx) hipLaunchKernelGGL((hvk_k_convert<TYPE, 1>), dim3(blocks), dim3(256), 0, stream, (const int *) iq, count, dst);
	else     hipLaunchKernelGGL((hvk_k_convert<TYPE, 0>), dim3(blocks), dim3(256), 0, stream, (const int *) iq, count, dst);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_convert(const void *iq, size_t count, int type, int cplx, void *dst, hipStream_t stream)
{
	if(count == 0) return(HVK_OK);
	switch(type)
	{
	case HVK_UINT8:  return(_launch_convert<HVK_UINT8>(iq, count, cplx, dst, stream));
	case HVK_INT8:   return(_launch_convert<HVK_INT8>(iq, count, cplx, dst, stream));
	case HVK_UINT16: return(_launch_convert<HVK_UINT16>(iq, count, cplx, dst, stream));
	case HVK_INT16:  return(_launch_convert<HVK_INT16>(iq, count, cplx, dst, stream));
	case HVK_INT32:  return(_launch_convert<HVK_INT32>(iq, count, cplx, dst, stream));
	case HVK_FLOAT:  return(_launch_convert<HVK_FLOAT>(iq, count, cplx, dst, stream));
	}
	return(HVK_ERROR);
}

/* ------------------------------------------------------------------ */
/* --pixelrate: rational poly-phase resampler, pixel-rate raster -> sample-rate
 * stream (src/video.c:3627-3651; arithmetic of fir_int16_process with
 * interpolation L and decimation D, src/fir.c:304-355). Output r of the
 * resampled stream is made from raster sample n = floor(r D / L) and the
 * ataps - 1 before it with the taps of phase (r D) mod L:
 *     out[r] = clamp16((sum_y x[n - ataps + 1 + y] * taps[phase][y]) >> 15)
 * A frame of the raster resamples to a whole number of outputs (hvk_tables.c
 * insists), so phase and position are frame local. One workgroup makes 1024
 * consecutive slab samples, 4 per lane: the raster samples they need and the
 * whole tap table are staged in LDS; stores are 8 bytes per lane, contiguous.
 * The slab written here is what hvk_k_filter reads: s_lead samples before a
 * frame's output sample 0, whose filter centre is resampled sample rs_shift. */
#define HVK_RS_TILE 1024
#define HVK_RS_WIN  (4 * HVK_RS_TILE + 72 + 8)      /* raster samples a tile can need: 1024 D / L + ataps, D <= 4 L */
#define HVK_RS_NP   11                              /* tap pairs per phase: ataps is 21 or 22 for every L (ntaps = 21 L | 1) */
#define HVK_RS_ROW  12                              /* dwords per phase row in LDS: 16-byte aligned rows */
/* BIG: ratios of more than 256 phases (27 MHz <-> 4 x f_sc: 709379 : 1080000) -- the reference's filter then has millions of taps
 * (src/fir.c:404: 21 L | 1) of which an output sample still uses 21 or 22: its phase's row, read from HBM where it is needed
 * instead of from a table in LDS, and positions in 64-bit arithmetic (a frame's index times D leaves 32 bits). */
template<bool BIG>
__global__ __launch_bounds__(256) void hvk_k_resample(const hvk_kconst_t k, const int16_t *__restrict__ Sp, const int *__restrict__ taps,
                                                      int16_t *__restrict__ S2,
                                                      /* frames of two lengths (k.rs_irr): per frame { c, where its samples go in S2 } -- c = B D - f RS L in (-D, D)
                                                       * with B the frame's first output sample: output r of the frame is made from raster sample
                                                       * floor((r D + c) / L) of the frame with the taps of phase (r D + c) mod L. NULL: c = 0, frame y at y * s_stride */
                                                      const int2v *__restrict__ frec)
{
	__shared__ __attribute__((aligned(16))) int win[HVK_RS_WIN / 2];        /* raster samples, two per dword */
	__shared__ __attribute__((aligned(16))) int tp[BIG ? 4 : 256 * HVK_RS_ROW];     /* taps, two per dword, one row per phase */
	typedef typename std::conditional<BIG, unsigned long long, unsigned>::type pos_t;

	const int t = threadIdx.x;
	const unsigned L = k.rs_L, D = k.rs_D;
	const int A = k.rs_ataps;
	const long slab_in = (long) k.slab_lines * k.width;
	const int16_t *in = Sp + (size_t) blockIdx.y * slab_in;     /* raster line -1 of the frame first */
	const int2v fr = frec ? frec[blockIdx.y] : (int2v) { 0, 0 };
	const pos_t cD = BIG ? (pos_t) (long long) fr.x : (pos_t) (unsigned) fr.x;     /* (added modulo 2^32 / 2^64: r D >= 2 D > |c|, hvk_tables.c) */
	int16_t *out = frec ? S2 + fr.y : S2 + (size_t) blockIdx.y * k.s_stride;

	const int q0 = blockIdx.x * HVK_RS_TILE;                    /* first slab sample of the tile */
	const pos_t r0 = (pos_t) (unsigned) (q0 - k.s_lead + k.rs_shift); /* its resampled-stream index, frame local (>= 0; r D < 2^32 unless BIG, hvk_tables.c) */
	/* first raster sample staged, frame local, rounded down to an even index so that pairs are dwords */
	const long n_lo = (((long) ((r0 * D + cD) / L) - (A - 1)) & ~1L);
	const long n_hi = (long) (((r0 + HVK_RS_TILE - 1) * D + cD) / L);
	const int count2 = (int) ((n_hi - n_lo + 2) / 2);           /* dwords */

	/* Loads first, LDS writes after: a load inside a loop with lane-dependent bounds is waited for in
	 * every round. The tap rows come packed from the host (hvk_engine.cpp), 3 x 16 bytes per phase;
	 * the window as two int16 per dword from clamped positions, zero outside the slab. */
	constexpr int TPASS = (256 * HVK_RS_ROW / 4 + 255) / 256;       /* 3 */
	constexpr int WPASS = (HVK_RS_WIN / 2 + 255) / 256;
	int4v trow[TPASS];
	int wlo[WPASS], whi[WPASS];
	const int nrows4 = (int) L * (HVK_RS_ROW / 4);
#pragma unroll
	for(int i = 0; i < TPASS; i++)
	{
		const int j = t + i * 256;
		if(!BIG) trow[i] = ((const int4v *) taps)[j < nrows4 ? j : nrows4 - 1];
	}
#pragma unroll
	for(int i = 0; i < WPASS; i++)
	{
		wlo[i] = whi[i] = 0;
		if(i * 256 < count2 + 1)                                /* the same for every lane */
		{
			const long p = n_lo + 2 * (t + i * 256) + k.width;      /* slab position: one halo line in front */
			const long pa = p < 0 ? 0 : (p < slab_in ? p : slab_in - 1), pb = p + 1 < 0 ? 0 : (p + 1 < slab_in ? p + 1 : slab_in - 1);
			wlo[i] = in[pa];
			whi[i] = in[pb];
		}
	}
#pragma unroll
	for(int i = 0; i < TPASS; i++)
	{
		const int j = t + i * 256;
		if(!BIG && j < nrows4) ((int4v *) tp)[j] = trow[i];
	}
#pragma unroll
	for(int i = 0; i < WPASS; i++)
	{
		const int j = t + i * 256;
		if(i * 256 < count2 + 1 && j < count2 + 1 && j < HVK_RS_WIN / 2)
		{
			const long p = n_lo + 2 * j + k.width;
			const int lo = (p >= 0 && p < slab_in) ? wlo[i] : 0, hi = (p + 1 >= 0 && p + 1 < slab_in) ? whi[i] : 0;
			win[j] = (lo & 0xFFFF) | (hi << 16);
		}
	}
	__syncthreads();

	/* this lane's first output: position and phase by one division, the next three by stepping */
	const pos_t rd = (r0 + (pos_t) (t * 4)) * D + cD;
	long n = (long) (rd / L);
	unsigned ph = (unsigned) (rd - (pos_t) n * L);

	short v[4];
#pragma unroll
	for(int i = 0; i < 4; i++)
	{
		const int ws = (int) (n - n_lo) - (A - 1);              /* window start in samples */
		const int *w = win + (ws >> 1);
		const int sh = (ws & 1) * 16;                           /* odd start: every pair straddles two dwords */
		const int4v *c = BIG ? (const int4v *) (taps + (size_t) ph * HVK_RS_ROW) : (const int4v *) (tp + ph * HVK_RS_ROW);
		const int4v c0 = c[0], c1 = c[1], c2 = c[2];
		const int ct[HVK_RS_ROW] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w };
		int a = 0, cur = w[0];
#pragma unroll
		for(int m = 0; m < HVK_RS_NP; m++)
		{
			const int nxt = w[m + 1];
			a = dot2((int) __builtin_amdgcn_alignbit((unsigned) nxt, (unsigned) cur, sh), ct[m], a);
			cur = nxt;
		}
		a >>= 15;
		v[i] = (short) (a < -32768 ? -32768 : (a > 32767 ? 32767 : a));

		ph += D;
		while(ph >= L) { ph -= L; n++; }
	}

	const int q = q0 + t * 4;
	if(q + 4 <= k.s_stride)
	{
		*(int2u *) (out + q) = (int2u) { ((int) v[0] & 0xFFFF) | ((int) v[1] << 16), ((int) v[2] & 0xFFFF) | ((int) v[3] << 16) };
	}
	else
	{
		for(int i = 0; i < 4; i++) if(q + i < k.s_stride) out[q + i] = v[i];
	}
}

/* S-Video behind the resampler and the video filter where the lines have two widths (hvk_kconst_t.sv_ring; src/video.c:3243,
 * :3578): the Q channel of an emitted line is what its buffer of the reference's ring holds -- the resampled sub-carrier of
 * the line's own content, which has the width of the line BEFORE it: a sample further on (or back) in the sub-carrier stream
 * where that width is not the one the stream's alignment was set by (delta: -1, 0, 1), and a line a sample longer than that
 * ends on what the buffer held (kind 1: the raster's sub-carrier of the line before at that place, src -- downwards; kind 3:
 * the raster's blanking, zero -- upwards; kind 2, an earlier sample of the stream, is no longer made by the host). One
 * workgroup per emitted line; rec[line] = { first output sample of the
 * line in the batch, width | delta << 16 | kind << 20, src, 0 }. C2 and Q in the batch's run of samples, s_lead in front. */
__global__ __launch_bounds__(256) void hvk_k_svq(const int4v *__restrict__ rec, const int16_t *__restrict__ C2, const int16_t *__restrict__ Craster,
                                                 int16_t *__restrict__ Q, const int s_lead)
{
	const int4v r = rec[blockIdx.x];
	const int w = r.y & 0xFFFF, delta = (int) ((unsigned) r.y << 12) >> 28, kind = (r.y >> 20) & 15;     /* (delta: four bits, signed) */
	const int16_t *src = C2 + s_lead + r.x + delta;
	int16_t *dst = Q + s_lead + r.x;
	const int valid = kind ? w - 1 : w;         /* (kind != 0: the line's last sample is the buffer's old content) */
	for(int x = threadIdx.x; x < valid; x += 256) dst[x] = src[x];
	if(kind && threadIdx.x == 0) dst[w - 1] = kind == 1 ? Craster[r.z] : (kind == 2 ? C2[s_lead + r.z] : (int16_t) 0);
}

extern "C" int hvk_launch_svq(const void *rec, int nlines, const void *C2, const void *Craster, void *Q, int s_lead, hipStream_t stream)
{
	if(nlines < 1) return(HVK_OK);
	hipLaunchKernelGGL(hvk_k_svq, dim3(nlines), dim3(256), 0, stream, (const int4v *) rec, (const int16_t *) C2, (const int16_t *) Craster, (int16_t *) Q, s_lead);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_resample(const hvk_kconst_t *k, const void *Sp, const void *taps, void *S2, int nframes, const void *frec, hipStream_t stream)
{
	const int tiles = (k->s_stride + HVK_RS_TILE - 1) / HVK_RS_TILE;
	if(k->rs_L > 256) hipLaunchKernelGGL(hvk_k_resample<true>, dim3(tiles, nframes), dim3(256), 0, stream, *k, (const int16_t *) Sp, (const int *) taps, (int16_t *) S2, (const int2v *) frec);
	else hipLaunchKernelGGL(hvk_k_resample<false>, dim3(tiles, nframes), dim3(256), 0, stream, *k, (const int16_t *) Sp, (const int *) taps, (int16_t *) S2, (const int2v *) frec);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

/* ------------------------------------------------------------------ */
/* The complex tail of the line pipeline for modes whose modulator is on the
 * device (everything but FM video): swap_iq, frequency offset, passthru, in the
 * reference's order (src/video.c:4587-4645). One I/Q pair (one dword) per lane
 * and step: the accesses of a wavefront are 256 contiguous bytes per stream.
 * off: the host's offset phasor, phase >> 16 as int16 pairs (hvk_tail.c);
 * pass: the external samples, zero where the source has ended. Both are dense
 * over the batch; frame i of the output sits at frame slot i * out_stride. */
template<int SWAP, int OFFSET, int PASS>
__global__ __launch_bounds__(256) void hvk_k_tail(int *__restrict__ iq, const int *__restrict__ off, const int *__restrict__ pass,
                                                  long frame_samples, long out_stride, long total)
{
	const long step = (long) gridDim.x * blockDim.x;

	for(long n = (long) blockIdx.x * blockDim.x + threadIdx.x; n < total; n += step)
	{
		const long f = n / frame_samples;
		const long o = f * out_stride * frame_samples + (n - f * frame_samples);
		const int v = iq[o];
		int i = (short) v, q = v >> 16;

		if(SWAP)
		{
			const int x = i;
			i = q;
			q = x;
		}

		if(OFFSET)
		{
			/* cint16_mul, src/common.h:58-67 */
			const int b = off[n];
			const int bi = (short) b, bq = b >> 16;
			const int ri = i * bi - q * bq;
			const int rq = i * bq + q * bi;
			i = (short) (ri >> 15);
			q = (short) (rq >> 15);
		}

		if(PASS)
		{
			/* int16 wrap-around add, src/video.c:3535-3538 */
			const int a = pass[n];
			i = (short) (i + (short) a);
			q = (short) (q + (a >> 16));
		}

		iq[o] = (i & 0xFFFF) | (q << 16);
	}
}

extern "C" int hvk_launch_tail(void *iq, const void *off, const void *pass, int swap, long frame_samples, long out_stride,
                               int nframes, hipStream_t stream)
{
	const long total = frame_samples * nframes;
	long blocks = (total + 255) / 256;
	if(blocks > 256 * 32) blocks = 256 * 32;
	if(total <= 0 || (!swap && !off && !pass)) return(HVK_OK);

#define TAIL(S, O, P) hipLaunchKernelGGL((hvk_k_tail<S, O, P>), dim3(blocks), dim3(256), 0, stream, \
	(int *) iq, (const int *) off, (const int *) pass, frame_samples, out_stride, total)
	switch((swap ? 4 : 0) | (off ? 2 : 0) | (pass ? 1 : 0))
	{
	case 1: TAIL(0, 0, 1); break;
	case 2: TAIL(0, 1, 0); break;
	case 3: TAIL(0, 1, 1); break;
	case 4: TAIL(1, 0, 0); break;
	case 5: TAIL(1, 0, 1); break;
	case 6: TAIL(1, 1, 0); break;
	case 7: TAIL(1, 1, 1); break;
	}
#undef TAIL
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

/* ------------------------------------------------------------------ */
/* launchers                                                           */

extern "C" int hvk_launch_expand_yuv(void *lut, const void *params, hipStream_t stream)
{
	hipLaunchKernelGGL(hvk_k_expand_yuv, dim3(0x1000000 / 256), dim3(256), 0, stream,
	                   (short4v *) lut, (const hvk_yuvparams_t *) params);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_check_levels(const void *lut, const void *params, int fast, int *differ, hipStream_t stream)
{
	if(fast == 2) hipLaunchKernelGGL(hvk_k_check_levels<2>, dim3(0x1000000 / 256), dim3(256), 0, stream, (const short4v *) lut, (const hvk_yuvparams_t *) params, differ);
	else hipLaunchKernelGGL(hvk_k_check_levels<1>, dim3(0x1000000 / 256), dim3(256), 0, stream, (const short4v *) lut, (const hvk_yuvparams_t *) params, differ);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" void hvk_raster_ptrs(const hvk_raster_args_t *a, hvk_rptrs_t *P)
{
	P->chroma = a->chroma;
	P->vbi_sym = a->vbi_sym;
	P->vbi_val = a->vbi_val;
	P->vbi_cov = a->vbi_cov;
	P->vbi_ops = a->vbi_ops;
	P->vbi_map = a->vbi_map;
	P->vits_l = a->vits_l;
	P->vits_c = a->vits_c;
	P->fsc_rows = a->fsc_rows;
	P->sis_dense = a->sis_dense;
	P->sis_win = a->sis_win;
	P->sis_first = a->sis_first;
	P->sis_bits = a->sis_bits;
	P->desc = a->desc;
	P->pulses = a->pulses;
	P->linebase = a->linebase;
	P->yuv = (const short4v *) a->yuv;
	P->yuvp = (const hvk_yuvparams_t *) a->yuvparams;
	P->clut = (const int *) a->clut;
	P->burst_win = a->burst_win;
	P->ghost = a->ghost;
	P->pool = a->pool;
	P->fdesc = a->fdesc;
}

template<int NT, int SECAM, int SV, int EXTRAS, int WC, int LV>
static int _launch_raster3(const hvk_raster_args_t *a, hipStream_t stream)
{
	const int W = a->k.width;
	int threads = (W + SPL - 1) / SPL;
	threads = (threads + 63) / 64 * 64;
	const size_t lds = ((size_t) ((W + 8 + 7) & ~7) + 2 * (size_t) ((W + 2 * HVK_CHROMA_LEAD + 7) & ~7)) * sizeof(int16_t) + 64;
	hvk_rptrs_t P;
	hvk_raster_ptrs(a, &P);
	hipLaunchKernelGGL((hvk_k_raster<NT, SECAM, SV, EXTRAS, WC, LV>), dim3(((a->linelist ? a->nlist : a->k.slab_lines) + 7) & ~7, a->nframes), dim3(threads), lds, stream,
	                   a->k, a->ctaps, a->notch, P, a->S, a->C, a->first_frame, a->frame_stride, a->linelist, a->nlist);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

template<int NT, int SECAM, int SV, int EXTRAS, int WC>
static int _launch_raster2(const hvk_raster_args_t *a, hipStream_t stream)
{
	/* levels computed per pixel or looked up: the engine decides per block of frames (hvk_engine_stage.cpp) */
	if(a->levels_computed) return(_launch_raster3<NT, SECAM, SV, EXTRAS, WC, 1>(a, stream));
	return(_launch_raster3<NT, SECAM, SV, EXTRAS, WC, 0>(a, stream));
}

template<int NT, int SECAM, int SV, int EXTRAS>
static int _launch_raster1(const hvk_raster_args_t *a, hipStream_t stream)
{
	/* the plain PAL kernel at 1024 samples per line gets the width as a constant */
	if(!SECAM && !SV && !EXTRAS && NT == 13 && a->k.width == 1024) return(_launch_raster2<NT, SECAM, SV, EXTRAS, (!SECAM && !SV && !EXTRAS && NT == 13) ? 1024 : 0>(a, stream));
	return(_launch_raster2<NT, SECAM, SV, EXTRAS, 0>(a, stream));
}

/* The plain kernels carry none of the optional stages (VBI data lines, insertion test signals, raw
 * baseband input, SECAM field identification): a render that uses none of them -- the benchmark
 * configuration among them -- does not pay their wave-uniform tests either */
template<int NT, int SECAM, int SV>
static int _launch_raster(const hvk_raster_args_t *a, hipStream_t stream)
{
	const bool extras = a->k.vbi || a->k.vits || a->k.rawbb || a->k.sis || a->k.fsc_mode || (SECAM && a->secam_fid);
	if(SV || extras) return(_launch_raster1<NT, SECAM, SV, 1>(a, stream));
	return(_launch_raster1<NT, SECAM, SV, 0>(a, stream));
}

extern "C" int hvk_launch_raster(const hvk_raster_args_t *a, hipStream_t stream)
{
	/* S-Video (baseband modes only) has kernels of its own: the benchmark path carries none of it */
	if(a->k.s_video)
	{
		if(a->k.secam) return(_launch_raster<1, 1, 1>(a, stream));
		switch(a->k.colour ? a->k.chroma_ntaps : 1)
		{
		case 3:  return(_launch_raster<3, 0, 1>(a, stream));
		case 5:  return(_launch_raster<5, 0, 1>(a, stream));
		case 7:  return(_launch_raster<7, 0, 1>(a, stream));
		case 9:  return(_launch_raster<9, 0, 1>(a, stream));
		case 11: return(_launch_raster<11, 0, 1>(a, stream));
		case 13: return(_launch_raster<13, 0, 1>(a, stream));
		case 15: return(_launch_raster<15, 0, 1>(a, stream));
		case 17: return(_launch_raster<17, 0, 1>(a, stream));
		case 19: return(_launch_raster<19, 0, 1>(a, stream));
		case 21: return(_launch_raster<21, 0, 1>(a, stream));
		case 23: return(_launch_raster<23, 0, 1>(a, stream));
		case 25: return(_launch_raster<25, 0, 1>(a, stream));
		case 27: return(_launch_raster<27, 0, 1>(a, stream));
		case 29: return(_launch_raster<29, 0, 1>(a, stream));
		case 31: return(_launch_raster<31, 0, 1>(a, stream));
		case 33: return(_launch_raster<33, 0, 1>(a, stream));
		}
		return(HVK_UNSUPPORTED);
	}
	if(a->k.secam) return(_launch_raster<1, 1, 0>(a, stream));
	switch(a->k.colour ? a->k.chroma_ntaps : 1)
	{
	case 1:  return(_launch_raster<1, 0, 0>(a, stream));   /* monochrome */
	case 3:  return(_launch_raster<3, 0, 0>(a, stream));   /* (no chroma low pass: fir8<3>) */
	case 5:  return(_launch_raster<5, 0, 0>(a, stream));
	case 7:  return(_launch_raster<7, 0, 0>(a, stream));
	case 9:  return(_launch_raster<9, 0, 0>(a, stream));
	case 11: return(_launch_raster<11, 0, 0>(a, stream));
	case 13: return(_launch_raster<13, 0, 0>(a, stream));
	case 15: return(_launch_raster<15, 0, 0>(a, stream));
	case 17: return(_launch_raster<17, 0, 0>(a, stream));
	case 19: return(_launch_raster<19, 0, 0>(a, stream));
	case 21: return(_launch_raster<21, 0, 0>(a, stream));
	case 23: return(_launch_raster<23, 0, 0>(a, stream));
	case 25: return(_launch_raster<25, 0, 0>(a, stream));
	case 27: return(_launch_raster<27, 0, 0>(a, stream));
	case 29: return(_launch_raster<29, 0, 0>(a, stream));
	case 31: return(_launch_raster<31, 0, 0>(a, stream));
	case 33: return(_launch_raster<33, 0, 0>(a, stream));
	}
	return(HVK_UNSUPPORTED);
}

template<int NT, int VF, int SV, int EXACT, int MF>
static int _launch_filter3(const hvk_filter_args_t *a, hipStream_t stream)
{
	const int tiles = (a->k.frame_samples + HVK_TILE - 1) / HVK_TILE;
	const int per_wg = HVK_TILES_PER_WG * HVK_FILTER_GROUP;
	hipLaunchKernelGGL((hvk_k_filter<NT, VF, SV, EXACT, MF>), dim3((tiles + per_wg - 1) / per_wg, a->nframes), dim3(HVK_TILE / SPL * HVK_FILTER_GROUP), 0, stream,
	                   a->k, a->itaps, a->qtaps, a->fdesc, a->S, (const int *) a->carriers, a->tilesyms,
	                   a->nicam_tapd, a->nicam_cca, a->C, (const int4v *) a->mfma_a, a->mfma_ci, a->mfma_cq,
	                   (int *) a->iq, a->out_stride, tiles);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

template<int NT, int VF, int SV, int EXACT>
static int _launch_filter2(const hvk_filter_args_t *a, hipStream_t stream)
{
	/* the matrix-unit form of the filter whenever the taps have one (hvk_engine.cpp:_mfma_taps) */
	if(NT == 51 && a->mfma_a) return(_launch_filter3<NT, VF, SV, EXACT, NT == 51>(a, stream));
	return(_launch_filter3<NT, VF, SV, EXACT, 0>(a, stream));
}

template<int NT, int VF, int SV>
static int _launch_filter(const hvk_filter_args_t *a, hipStream_t stream)
{
	/* whole tiles only, and at least a window's length of slab behind the frame */
	const bool exact = a->k.frame_samples % HVK_TILE == 0 && a->k.s_stride - a->k.s_lead - a->k.frame_samples >= 128;
	if(exact) return(_launch_filter2<NT, VF, SV, 1>(a, stream));
	return(_launch_filter2<NT, VF, SV, 0>(a, stream));
}

extern "C" int hvk_launch_filter(const hvk_filter_args_t *a, hipStream_t stream)
{
	if(a->k.s_video)
	{
		if(a->k.vf_type == 0) return(_launch_filter<1, 0, 1>(a, stream));
		if(a->k.vf_type == 1 && a->k.vf_ntaps == 51) return(_launch_filter<51, 1, 1>(a, stream));
		return(HVK_UNSUPPORTED);
	}
	if(a->k.vf_type == 0) return(_launch_filter<1, 0, 0>(a, stream));
	/* the FM video pre-emphasis tables: real, 67 or 71 taps (vector-unit form) */
	if(a->k.vf_type == 1 && a->k.vf_ntaps == 67) return(_launch_filter<67, 1, 0>(a, stream));
	if(a->k.vf_type == 1 && a->k.vf_ntaps == 71) return(_launch_filter<71, 1, 0>(a, stream));
	if(a->k.vf_ntaps != 51) return(HVK_UNSUPPORTED);
	if(a->k.vf_type == 1) return(_launch_filter<51, 1, 0>(a, stream));
	if(a->k.vf_type == 3) return(_launch_filter<51, 3, 0>(a, stream));
	return(HVK_UNSUPPORTED);
}

/* ------------------------------------------------------------------ */
/* hvk_block_sums(): s1 = sum w[i], s2 = sum (i + 1) w[i] modulo 2^64 over the words of a run of I/Q pairs */

__global__ __launch_bounds__(256)
void hvk_k_sums(const uint32_t *__restrict__ w, const size_t count, unsigned long long *__restrict__ sums)
{
	unsigned long long s1 = 0, s2 = 0;
	const size_t stride = (size_t) gridDim.x * blockDim.x * 4;
	/* four consecutive words per lane and step (16-byte loads where the run starts 16-byte aligned; the tail word by word) */
	for(size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += stride)
	{
		if(i + 4 <= count)
		{
			const int4u v = *(const int4u *) (w + i);
			const unsigned long long a = (unsigned) v.x, b = (unsigned) v.y, c = (unsigned) v.z, d = (unsigned) v.w;
			s1 += a + b + c + d;
			s2 += (unsigned long long) (i + 1) * a + (unsigned long long) (i + 2) * b + (unsigned long long) (i + 3) * c + (unsigned long long) (i + 4) * d;
		}
		else
		{
			for(size_t j = i; j < count; j++) { s1 += w[j]; s2 += (unsigned long long) (j + 1) * w[j]; }
		}
	}
	for(int o = 32; o > 0; o >>= 1)
	{
		s1 += __shfl_down(s1, o, 64);
		s2 += __shfl_down(s2, o, 64);
	}
	if((threadIdx.x & 63) == 0)
	{
		atomicAdd(&sums[0], s1);
		atomicAdd(&sums[1], s2);
	}
}

extern "C" int hvk_launch_sums(const void *iq, size_t count, unsigned long long *sums, hipStream_t stream)
{
	if(count == 0) return(HVK_OK);
	size_t blocks = (count / 4 + 255) / 256;
	if(blocks > 4096) blocks = 4096;
	if(blocks < 1) blocks = 1;
	hipLaunchKernelGGL(hvk_k_sums, dim3((unsigned) blocks), dim3(256), 0, stream, (const uint32_t *) iq, count, sums);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}
