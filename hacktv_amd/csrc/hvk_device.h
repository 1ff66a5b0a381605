/* hvk_device.h -- device code shared by the gfx950 kernels (hvk_kernels.hip, hvk_fused.hip):
 * integer helpers, the 8-outputs-per-lane FIR on packed int16 pairs, and the raster of one
 * scanline (_vid_next_line_raster, src/video.c:2864-3066) cut into the steps a kernel strings
 * together around its barriers:
 *
 *   raster_setup()    which line of which frame: descriptors, picture geometry       (scalar)
 *   raster_loads()    the loads nothing but the descriptors depends on: source row,
 *                     ghost samples, sub-carrier phasors
 *   raster_clear()    zero the two chroma channels in LDS
 *        -- barrier --
 *   raster_pixels()   RGB -> levels (table look-up or computed) -> Y / U / V in LDS
 *        -- barrier --
 *   raster_compute()  8 consecutive samples per lane: sync pulses, luma, chroma low pass,
 *                     burst, QAM, SECAM notch + sub-carrier, insertion test signals, VBI data
 *                     lines (these last two stages have barriers of their own, under
 *                     workgroup-uniform tests)
 *
 * hvk_k_raster writes the samples to the raster slab in HBM; hvk_k_fused hands them to the
 * video filter through LDS byte planes.
 */
#ifndef HVK_DEVICE_H
#define HVK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hvk_internal.h"
#include "hvk_kernels.h"

typedef short  short2v __attribute__((ext_vector_type(2)));
typedef short  short4v __attribute__((ext_vector_type(4)));
typedef int    int4v   __attribute__((ext_vector_type(4)));
typedef int    int2v   __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
/* four dwords that are only dword aligned: global_load_dwordx4 needs no more */
typedef int    int4u   __attribute__((ext_vector_type(4), aligned(4)));
typedef int    int2u   __attribute__((ext_vector_type(2), aligned(4)));
/* ... and four that are only 2-byte aligned (global memory only) */
typedef int    int4a2  __attribute__((ext_vector_type(4), aligned(2)));
typedef int    int_a2  __attribute__((aligned(2)));

#define SPL HVK_SPL

/* Profiling switches (tools/ablate.py): stages can be skipped to time the rest -- with WRONG output.
 * Compiled in only with -DHVK_ENABLE_ABLATE=1 (make -C hacktv_amd/csrc ABLATE=1); a normal build has none. */
#ifndef HVK_ENABLE_ABLATE
#define HVK_ENABLE_ABLATE 0
#endif
#define ABLATE(bit) (HVK_ENABLE_ABLATE && (k.ablate & (bit)))
#define HVK_PIX_PASSES 8      /* the raster block has >= width / 8 lanes */

__device__ __forceinline__ int wrap16(int v) { return((int) (short) v); }
/* the middle one of three (v_med3_i32): a clamp when lo <= hi */
__device__ __forceinline__ int med3i(int v, int lo, int hi) { return(v < lo ? lo : (v > hi ? hi : v)); }
__device__ __forceinline__ int clamp16(int v) { return(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
/* a.lo * b.lo + a.hi * b.hi with nothing to add to: the three-operand form with the constant 0 (from the builtin the
 * compiler makes the two-operand v_dot2c, which accumulates into its destination -- and a v_mov of 0 in front of every one) */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "hvk_device.h is written for gfx950 (MI355X): dot2z()'s instruction and nicam_add()'s LDS addresses are that ISA's (gfx942 / gfx90a share them); make ARCH=gfx950"
#endif
__device__ __forceinline__ int dot2z(int a, int b)
{
#ifdef HVK_V_NO_DOT2Z
	return(__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), 0, false));
#else
	int d;
	asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
	return(d);
#endif
}

__device__ __forceinline__ int dot2(int a, int b, int c)
{
	return(__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false));
}
/* (lo >> 16) | (hi << 16): the pair of int16 that starts one element later */
__device__ __forceinline__ int shift_pair(int lo, int hi) { return((int) __builtin_amdgcn_alignbit((unsigned) hi, (unsigned) lo, 16)); }
/* (sat16(lo) & 0xFFFF) | (sat16(hi) << 16) */
__device__ __forceinline__ int sat_pack16(int lo, int hi) { return(__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(lo, hi))); }

__device__ __forceinline__ int pk_add16(int a, int b)
{
	return(__builtin_bit_cast(int, (ushort2v) (__builtin_bit_cast(ushort2v, a) + __builtin_bit_cast(ushort2v, b))));
}
__device__ __forceinline__ int pk_mad16(int a, int b, int c)
{
	return(__builtin_bit_cast(int, (ushort2v) (__builtin_bit_cast(ushort2v, a) * __builtin_bit_cast(ushort2v, b) + __builtin_bit_cast(ushort2v, c))));
}

/* 8 consecutive FIR outputs from a register window of packed int16 pairs.
 * d[] holds window elements w[0..], two per dword; output i is
 *   sum_k tap[k] * w[START + i + k]
 * with START in {0,1}. tp[] holds the NT taps packed two per dword and zero
 * padded. Outputs whose first element is dword aligned use d[] directly, the
 * others use the pairs shifted by one element. */
template<int NT, int START>
__device__ __forceinline__ void fir8(const int *d, const int *tp, int (&acc)[SPL])
{
	if(NT == 3)
	{
		/* "three taps" stands for NO filter (a colour mode without a chroma low pass: `ntsc-a`, src/video.c:3016 with
		 * :3998): the window's middle element, scaled so that the callers' >> 15 hands it back as it is */
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			const int e = START + i + 1;
			acc[i] = (int) (((e & 1) ? (d[e / 2] >> 16) : (int) (short) (d[e / 2] & 0xFFFF)) * 32768);
		}
		return;
	}
	constexpr int NP = (NT + 1) / 2;
	constexpr int NS = SPL / 2 + NP;   /* shifted pairs needed */
	int sh[NS];

#pragma unroll
	for(int m = 0; m < NS; m++) sh[m] = shift_pair(d[m], d[m + 1]);

#pragma unroll
	for(int i = 0; i < SPL; i++)
	{
		int a = 0;
		const int e = START + i;        /* first window element of this output */
#pragma unroll
		for(int j = 0; j < NP; j++)
		{
			a = dot2((e & 1) ? sh[e / 2 + j] : d[e / 2 + j], tp[j], a);
		}
		acc[i] = a;
	}
}

/* ------------------------------------------------------------------ */

/* RGB -> (Y, U, V) levels of one colour: src/video.c:3917-3958, same order of operations, no
 * contraction. Used to expand the 2^24-entry table once per engine and, when the pictures have too
 * many colours for the table's cache lines to be found again (moving video), per pixel. */
/* SEC: -1 the mode's kind (SECAM or not) is read from the parameters, 0 / 1: known when the kernel is compiled */
/* FAST: 0 the reference's sequence of operations; 2 the short form (hvk_yuvparams_t.fast); 1 the short form for the two
 * colour-difference levels only, the luma the reference's way (modes whose luma constants make exact ties of many colours:
 * SECAM-L) -- which of them a mode may use is found by trying all 2^24 colours (hvk_engine.cpp) */
/* level_from(): from the three gamma values (a caller that keeps the 256 of them in LDS reads them there itself) */
template<int SEC = -1, int FAST = 0>
__device__ __forceinline__ short4v level_from(const double r, const double g, const double b, const hvk_yuvparams_t &p)
{
	double y, u, v;

	if(FAST)
	{
		const double M = 6755399441055744.0;       /* 1.5 * 2^52: the sum's low dword is the addend rounded to nearest */
		int iy;
		if(FAST == 2)
		{
			y = __builtin_fma(b, p.bw, __builtin_fma(g, p.gw, r * p.rw));
			iy = __double2loint(__builtin_fma(y, p.f_y1, p.f_y0) + M);
			iy = iy < -32767 ? -32767 : (iy > 32767 ? 32767 : iy);
		}
		else
		{
			y = r * p.rw + g * p.gw + b * p.bw;
			double yl = (p.black + (y * p.range)) * p.level;
			yl = fmin(fmax(yl, -1.0), 1.0);
			iy = (int) (short) round(yl * 32767);
		}
		const int iu = __double2loint(__builtin_fma(b - y, p.f_u1, p.f_u0) + M);
		const int iv = __double2loint(__builtin_fma(r - y, p.f_v1, p.f_v0) + M);
		short4v o;
		o.x = (short) iy;
		o.y = (short) (iu < -32767 ? -32767 : (iu > 32767 ? 32767 : iu));
		o.z = (short) (iv < -32767 ? -32767 : (iv > 32767 ? 32767 : iv));
		o.w = 0;
		return(o);
	}

	y = r * p.rw + g * p.gw + b * p.bw;
	u = (b - y) * p.eu;
	v = (r - y) * p.ev;

	y = (p.black + (y * p.range)) * p.level;
	if(SEC < 0 ? !p.secam : !SEC)
	{
		u *= p.chroma_scale;
		v *= p.chroma_scale;
	}
	else
	{
		/* frequency deviation of the D'b / D'r rest frequencies from the FM centre,
		 * in units of the 1 MHz full scale (src/video.c:3951-3952, :45-48) */
		u = (u + 4250000.0 - 4328125.0) / 1000000.0;
		v = (v + 4406250.0 - 4328125.0) / 1000000.0;
	}

	/* limited to [-1, 1] (src/video.c:3954-3956; never NaN): v_max_f64 / v_min_f64 */
	y = fmin(fmax(y, -1.0), 1.0);
	u = fmin(fmax(u, -1.0), 1.0);
	v = fmin(fmax(v, -1.0), 1.0);

	short4v o;
	o.x = (short) round(y * 32767);
	o.y = (short) round(u * 32767);
	o.z = (short) round(v * 32767);
	o.w = 0;
	return(o);
}

template<int SEC = -1, int FAST = 0>
__device__ __forceinline__ short4v level_of(unsigned c, const hvk_yuvparams_t &p)
{
	return(level_from<SEC, FAST>(p.glut[(c & 0xFF0000) >> 16], p.glut[(c & 0x00FF00) >> 8], p.glut[(c & 0x0000FF) >> 0], p));
}

/* ------------------------------------------------------------------ */
/* the raster of one scanline                                           */

/* What the raster reads, besides the engine constants and the filter taps (kernel arguments of
 * their own: they live in SGPRs) */
typedef struct {
	const int16_t *chroma;        /* SECAM: [frames][frame_samples] values to add; raw baseband input: the slab */
	const int *vbi_sym;           /* VBI data lines: every table's symbols, { first sample, length, start } */
	const int16_t *vbi_val;       /*   symbol values */
	const int *vbi_cov;           /*   per table and sample: the (at most HVK_VBI_COVER) symbols that lie over it and their values there */
	const unsigned *vbi_ops;      /*   [frames][HVK_VBI_OPS][16]: symbol base, bits, blank range, -, 12 data words (LSB first) */
	const signed char *vbi_map;   /* [frames][lines]: op of the line or -1 */
	const int16_t *vits_l;        /* VITS: [n][width] luma added */
	const int16_t *vits_c;        /*       [n][width] chroma amplitude */
	const int16_t *fsc_rows;      /* field-sequential colour: [2][width] the flag pulses as dense rows */
	const int16_t *sis_dense;     /* sound-in-syncs: [50][HVK_SIS_SPAN] the half symbols as dense rows */
	const int16_t *sis_win;       /*   the blanking window, k.sis_width values from sample k.sis_left */
	const int16_t *sis_first;     /*   [HVK_SIS_SPAN] what the last never-emitted invocation leaves on the stream's first line */
	const unsigned *sis_bits;     /*   [frames][lines + 1 (+ 2 with the resampler)][2]: a line's burst (the last: the line(s) behind the frame) -- 7 bytes of bits (MSB first), their number in the eighth */
	const hvk_linedesc_t *desc;
	const int16_t *pulses;
	const int16_t *linebase;      /* [rows][k.base_stride]: blanking + sync pulses of every kind of line */
	const short4v *yuv;
	const hvk_yuvparams_t *yuvp;  /* LV: what the table is made from */
	const int *clut;
	const int16_t *burst_win;
	const int16_t *ghost;
	const uint32_t *pool;
	const hvk_framedesc_t *fdesc; /* [frames][1 + fields]: the frame before, then the fields */
} hvk_rptrs_t;

/* What a lane fetches for a line besides pixels and sub-carrier phasors: issued with them, used in raster_compute() */
typedef struct {
	int ghost_u, ghost_v;   /* the samples the reference reads past its chroma buffer (lanes < H) */
	int4v base;             /* the lane's 8 samples of the line's base line (blanking + sync pulses) */
	int4v bwin;             /* ... of the burst window, zero outside it */
} hvk_side_t;

/* One line's state: all of it the same for every lane (SGPRs) */
typedef struct {
	int rel;                /* line of the frame; -1 and `lines` (and `lines` + 1 with the resampler) are halo lines */
	bool own, zero;         /* own: a line of this frame, not a halo line; zero: before the stream -- the filter history is zero, not blanking */
	hvk_linedesc_t d;
	int64_t row_off;        /* the source row in the pool, less the sample of source pixel 0: pixel of sample x at pool[row_off + x] */
	unsigned coff;          /* colour table position of the line's first sample */
	int pal, vbi_op, vits_i;
	int base_row;           /* the line's row of the base-line table */
	int fsc, fsc_flag;      /* field-sequential colour: the channel the line shows (bits to shift a pixel right by), the flag row it carries or -1 */
	int ax0, ax1, ar_eff;   /* samples [ax0, ax1) show a source pixel; luma is assigned up to ar_eff */
	bool active, has_pix;
	bool stream_first;      /* line 1 of the stream's first frame */
	int chroma_row;         /* SECAM: the row of the sub-carrier store the line's frame has its colour chain's output in */
} hvk_line_t;

/* which line of which frame, without dividing the global line number: line of the field table, frame parity,
 * whether the line belongs to the frame at all (the halo lines do not), whether it lies before the stream */
__device__ __forceinline__ void raster_line_index(const hvk_kconst_t &k, const int rel, const int64_t frame_index,
                                                  int &line0, int &par, bool &own, bool &zero)
{
	line0 = rel;
	par = (int) ((frame_index + 1) & 1);
	own = true;
	if(rel < 0) { line0 = k.lines - 1; par ^= 1; own = false; }
	else if(rel >= k.lines) { line0 = rel - k.lines; par ^= 1; own = false; }
	/* before the stream: the filter history is zero, not blanking
	 * (src/video.c:4665-4667 with src/fir.c:289, :579) */
	zero = rel < 0 && frame_index == 0;
}

/* which of a frame's descriptors a line looks at: one per frame, or per field with --interlace -- the second
 * field shows its own source frame. The halo line in front is the last line of the frame BEFORE (entry 0): on
 * 525 lines it shows picture, whose last samples the filter sees from this frame's first outputs. */
__device__ __forceinline__ int raster_fdesc_of(const hvk_kconst_t &k, const int rel)
{
	return(rel < 0 ? 0 : ((k.fields == 2 && rel >= k.hline - 1 && rel < k.lines) ? 2 : 1));
}

template<int SECAM, int EXTRAS>
__device__ __forceinline__ hvk_line_t raster_setup_core(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_framedesc_t &f, const hvk_linedesc_t &d,
                                                        const int y, const int rel, const int line0, const bool own, const bool zero);

template<int SECAM, int EXTRAS>
__device__ __forceinline__ hvk_line_t raster_setup(const hvk_kconst_t &k, const hvk_rptrs_t &P, const int y, const int rel,
                                                   const int64_t first_frame, const int64_t frame_stride)
{
	int line0, par;
	bool own, zero;
	/* frame number and parity by arithmetic: the line descriptor's fetch does not wait for the frame descriptor's */
	raster_line_index(k, rel, first_frame + (int64_t) y * frame_stride, line0, par, own, zero);
	const hvk_framedesc_t f = P.fdesc[__builtin_amdgcn_readfirstlane(y * (k.fields + 1) + raster_fdesc_of(k, rel))];   /* one scalar load of the whole descriptor */
	const hvk_linedesc_t d = P.desc[__builtin_amdgcn_readfirstlane(par * k.lines + line0)];
	return(raster_setup_core<SECAM, EXTRAS>(k, P, f, d, y, rel, line0, own, zero));
}

template<int SECAM, int EXTRAS>
__device__ __forceinline__ hvk_line_t raster_setup_core(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_framedesc_t &f, const hvk_linedesc_t &d,
                                                        const int y, const int rel, const int line0, const bool own, const bool zero)
{
	hvk_line_t L;

	L.rel = rel;
	L.own = own;
	L.zero = zero;
	L.stream_first = own && rel == 0 && f.frame_index == 0;
	L.chroma_row = f.chroma_row;
	L.d = d;
	L.pal = k.colour ? L.d.pal : 0;

	/* a VBI data line (teletext packet, WSS, VITC: the host lists them per frame), an insertion test signal */
	L.vbi_op = -1;
	L.vits_i = -1;
	if(EXTRAS && k.vbi && L.own) L.vbi_op = __builtin_amdgcn_readfirstlane((int) P.vbi_map[(size_t) y * k.lines + line0]);
	if(EXTRAS && k.vits && L.own)
	{
		for(int i = 0; i < 4; i++) if(i < k.vits && line0 == k.vits_line[i]) L.vits_i = i;
	}

	/* blanking + sync pulses: the row of the line's kind -- among the stream's first lines the one without what the line
	 * before would leave behind its end (hvk_kconst_t.spill_lines) */
	L.base_row = L.d.secam_fid >> 8;
	if(k.spill_lines && f.frame_index == 0 && own && rel >= 0 && rel < k.spill_lines) L.base_row = (L.d.secam_fid >> 1) & 0x7F;

	/* field-sequential colour: the frame's number counted from 1, two fields a frame (src/video.c:2919-2930); the flag on
	 * one line of one field of the three (:3043-3063) */
	L.fsc = 0;
	L.fsc_flag = -1;
	/* (lines read from an external baseband stream are not drawn at all -- _vid_next_line_rawbb, src/video.c:2406-2446 -- and
	 * carry no flag: tools/fuzz_parity.py 300 9090 found the engine adding one, round 5) */
	if(k.fsc_mode && !(EXTRAS && k.rawbb))
	{
		const int64_t frame_no = f.frame_index + 1 + (rel < 0 ? -1 : (rel >= k.lines ? 1 : 0));
		const int line = line0 + 1;
		const int fsc = (int) ((frame_no * 2 + (line < k.fsc_split ? 0 : 1)) % 3);
		L.fsc = 8 * fsc;
		if(k.fsc_mode == 1 && fsc == 1 && (line == 18 || line == 281)) L.fsc_flag = 0;
		if(k.fsc_mode == 2 && fsc == 2 && (line == 1 || line == 203)) L.fsc_flag = line == 1 ? 0 : 1;
	}

	/* ---- picture geometry ---- */
	int vy = L.d.src_row;
	if(vy >= 0 && k.interlaced != 0 && f.fb_interlaced != k.interlaced) vy += 1;
	vy -= f.vframe_y;
	if(vy < 0 || vy >= f.fb_height || !(L.own || rel < 0) || !f.fb_valid) vy = -1;

	const int px0 = k.active_left + f.vframe_x;                 /* sample of source pixel 0 */
	L.active = !(EXTRAS && k.rawbb) && L.d.ar > L.d.al;         /* raw baseband input: no picture is drawn */
	L.has_pix = L.active && vy >= 0;
	L.ax0 = L.d.al > px0 ? L.d.al : px0;                        /* samples that show a source pixel */
	L.ax1 = L.d.ar < px0 + f.fb_width ? L.d.ar : px0 + f.fb_width;
	if(!L.has_pix) L.ax1 = L.ax0 = 0;
	/* the reference fills the border left of the picture without looking at the
	 * right end of the active part (src/video.c:2972-2975): on a left-half line a
	 * picture narrow enough to start beyond mid-line pushes the black fill past it */
	L.ar_eff = (px0 > L.d.al && px0 > L.d.ar) ? px0 : L.d.ar;
	/* the pool holds dense pictures (hvk_frame_upload gathers strided and flipped sources): pixel stride 1 */
	L.row_off = f.fb_offset + (int64_t) vy * f.line_stride - px0;

	/* The sub-carrier table position advances by one line per line, colour or not: position relative
	 * to the frame's (fprev[] carries THIS frame's position). */
	{
		const int W = k.width;
		unsigned coff = (f.clut_off0 + (unsigned) (rel + 1) * (unsigned) W) % k.clw;
		L.coff = (coff + k.clw - ((unsigned) W % k.clw)) % k.clw;
	}
	return(L);
}

/* The loads nothing but the descriptors depends on go out first, longest chain first: the source
 * row (its pixels index the level table, whose entries go to LDS), then the samples the reference
 * reads past its chroma buffer, then the sub-carrier phasors. All unconditional, at clamped
 * positions (ax1 > ax0 when there is a picture): a load under a lane test gets a wait of its own
 * from the compiler, and eight round trips in a row. */
template<int PASSES = HVK_PIX_PASSES>
__device__ __forceinline__ void raster_load_rgb(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L, const int t, const int nth,
                                                uint32_t (&rgb)[HVK_PIX_PASSES])
{
	/* without a picture: the pool's first pixel, eight times (never used) -- no branch, no merge */
	const bool pix = L.has_pix && !ABLATE(8);
	const uint32_t *row = pix ? P.pool + L.row_off : P.pool;
#pragma unroll
	for(int i = 0; i < PASSES; i++)
	{
		const int x = L.ax0 + t + i * nth;
		rgb[i] = row[pix ? (x < L.ax1 ? x : L.ax1 - 1) : 0];
	}
}

/* ALWAYS: every load goes out whatever the line is (at a clamped position where it has no use): no branch
 * for the compiler to put a wait behind */
template<int NT, int WC, int ALWAYS = 0, int NOCLUT = 0>
__device__ __forceinline__ void raster_load_side(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L, const int t,
                                                 hvk_side_t &sd, int (&c)[SPL])
{
	constexpr int H = NT / 2;
	const int W = WC ? WC : k.width;
	const int x0 = t * SPL;

	sd.ghost_u = sd.ghost_v = 0;
	if(NT > 1)
	{
		const int gt = t < H ? t : H - 1;
		sd.ghost_u = P.ghost[2 * gt + 0];
		sd.ghost_v = P.ghost[2 * gt + 1];
	}

	/* the line's base: one aligned 16-byte load wherever the lane stands (rows are padded) */
	{
		const int xb = x0 < k.base_stride - SPL ? x0 : k.base_stride - SPL;
		sd.base = *(const int4v *) (P.linebase + (size_t) L.base_row * k.base_stride + xb);
	}
	/* the lane's 8 burst window values in one 16-byte load from the zero-padded table (2-byte aligned:
	 * global memory takes that); lanes away from the burst read zeros */
	sd.bwin = (int4v) { 0, 0, 0, 0 };
	if(NT > 1)
	{
		const int b0 = x0 - k.burst_left;
		const int4a2 w = *(const int4a2 *) (P.burst_win + (b0 < -HVK_PULSE_PAD ? -HVK_PULSE_PAD : (b0 < k.burst_width ? b0 : k.burst_width)));
		sd.bwin = (int4v) { w.x, w.y, w.z, w.w };
	}

	/* sub-carrier phasors of this lane's samples, fetched now so that the read is
	 * in flight during the picture and filter phases */
	if(ALWAYS && NT > 1)
	{
		/* (a lane that straddles the line's end reads on into the table: it is 8 entries longer than it needs to be) */
		const int4u a = ((const int4u *) (P.clut + L.coff + (x0 < W ? x0 : 0)))[0], b = ((const int4u *) (P.clut + L.coff + (x0 < W ? x0 : 0)))[1];
		c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
		return;
	}
#pragma unroll
	for(int i = 0; i < SPL; i++) c[i] = 0;
	if(NOCLUT) return;                      /* the picture planes are made without the sub-carrier (hvk_k_prep) */
	if((L.pal || (L.vits_i >= 0 && k.colour)) && x0 < W && !ABLATE(4))
	{
		const int *cl = P.clut + L.coff + x0;
		if(x0 + SPL <= W)
		{
			const int4u a = ((const int4u *) cl)[0], b = ((const int4u *) cl)[1];
			c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
		}
		else
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) c[i] = (x0 + i < W) ? cl[i] : 0;
		}
	}
}

template<int NT, int WC, int PASSES = HVK_PIX_PASSES, int NOCLUT = 0>
__device__ __forceinline__ void raster_loads(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L, const int t, const int nth,
                                             uint32_t (&rgb)[HVK_PIX_PASSES], hvk_side_t &sd, int (&c)[SPL])
{
	raster_load_rgb<PASSES>(k, P, L, t, nth, rgb);
	raster_load_side<NT, WC, 0, NOCLUT>(k, P, L, t, sd, c);
}

/* LDS layout of the raster (int16 elements):
 *   Y  [YL]  luma of the picture part of the line, index = sample x
 *   U  [CL]  chroma channels, index j <-> sample x = j - H (H = ntaps / 2), so
 *   V  [CL]  a lane's FIR window starts at its own first sample index
 * YL and CL are multiples of 8 elements: every lane's slice is 16-byte aligned. */
__device__ __forceinline__ int raster_YL(int W) { return((W + 8 + 7) & ~7); }
__device__ __forceinline__ int raster_CL(int W) { return((W + 2 * HVK_CHROMA_LEAD + 7) & ~7); }

/* clear both chroma channels (a barrier has to follow before raster_pixels()) */
__device__ __forceinline__ void raster_clear(const hvk_line_t &L, const int t, const int nth, int16_t *U, const int CL)
{
	if(L.pal)
	{
		for(int j = t * 8; j < 2 * CL; j += nth * 8) *(int4v *) (U + j) = (int4v) { 0, 0, 0, 0 };
	}
}

/* RGB -> levels of the pixels loaded by raster_load_rgb(): all look-ups are issued together -- the two
 * dependent global loads per pixel are paid once per line, not once per pass (nth * HVK_PIX_PASSES >= width).
 * PASSES: how many passes of nth pixels cover the line's pixels (ax1 - ax0 <= PASSES * nth). */
template<int LV, int PASSES = HVK_PIX_PASSES, int ALWAYS = 0>
__device__ __forceinline__ void raster_gather(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L,
                                              uint32_t (&rgb)[HVK_PIX_PASSES], short4v (&c)[HVK_PIX_PASSES])
{
	if(ALWAYS || (L.has_pix && !ABLATE(8)))
	{
		/* the pixels are first needed HERE: keeps the compiler from preparing the table addresses
		 * (and waiting for the loads) right where they were issued */
#pragma unroll
		for(int i = 0; i < PASSES; i++) asm volatile("" : "+v"(rgb[i]));
		if(k.fsc_mode)
		{
			/* one colour channel of the pixel as a grey (src/video.c:2995-3000) */
#pragma unroll
			for(int i = 0; i < PASSES; i++) rgb[i] = ((rgb[i] >> L.fsc) & 0xFFu) * 0x010101u;
		}
#pragma unroll
		for(int i = 0; i < PASSES; i++)
		{
			/* LV: the levels computed from the colour instead of looked up -- the same arithmetic that
			 * fills the table. A table entry is 8 bytes somewhere in 128 MiB; pictures with many colours
			 * (moving video) pay an HBM round trip and a 64-byte sector for most pixels. */
			if(LV) c[i] = level_of(rgb[i] & 0xFFFFFFu, *P.yuvp);
			else c[i] = ABLATE(1) ? (short4v) { (short) rgb[i], (short) (rgb[i] >> 8), (short) (rgb[i] >> 12), 0 } : P.yuv[rgb[i] & 0xFFFFFFu];
		}
	}
}

/* ... and into the staging area: Y where there is a pixel, U / V likewise. NOCLEAR = 0: the chroma
 * channels have been cleared (raster_clear() and a barrier); NOCLEAR = 1: the zeros around the
 * pixels are written here, to addresses no pixel goes to, from `zero_from` on (what is in front of
 * that is not read: a line of which only the tail is wanted). */
template<int NT, int WC, int PASSES = HVK_PIX_PASSES, int NOCLEAR = 0>
__device__ __forceinline__ void raster_stage(const hvk_kconst_t &k, const hvk_line_t &L, const int t, const int nth,
                                             const short4v (&c)[HVK_PIX_PASSES], const int ghost_u, const int ghost_v,
                                             int16_t *Yb, int16_t *U, int16_t *V, const int zero_from = 0)
{
	constexpr int H = NT / 2;
	const int W = WC ? WC : k.width;

	if(L.pal)
	{
		if(NOCLEAR)
		{
			const int CL = raster_CL(W);
			/* left of the pixels; right of them (without a picture only the ghost samples' surroundings are read) */
			const int lo_end = L.has_pix ? H + L.ax0 : 0;
			const int hi_beg = L.has_pix ? H + L.ax1 : (W > 2 * HVK_CHROMA_LEAD ? W - 2 * HVK_CHROMA_LEAD : 0);
			for(int j = zero_from + t; j < lo_end; j += nth) U[j] = V[j] = 0;
			for(int j = hi_beg + t; j < CL; j += nth) if(j < H + W || j >= H + W + H) U[j] = V[j] = 0;
		}
		/* the samples the reference reads past the end of its buffer */
		if(t < H)
		{
			U[H + W + t] = (int16_t) ghost_u;
			V[H + W + t] = (int16_t) ghost_v;
		}
	}

	if(L.has_pix && !ABLATE(8))
	{
#pragma unroll
		for(int i = 0; i < PASSES; i++)
		{
			const int x = L.ax0 + t + i * nth;
			if(x < L.ax1)
			{
				Yb[x] = c[i].x;
				if(L.pal)
				{
					U[H + x] = c[i].y;
					V[H + x] = c[i].z;
				}
			}
		}
	}
}

template<int NT, int WC, int LV, int PASSES = HVK_PIX_PASSES>
__device__ __forceinline__ void raster_pixels(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L, const int t, const int nth,
                                              uint32_t (&rgb)[HVK_PIX_PASSES], const int ghost_u, const int ghost_v,
                                              int16_t *Yb, int16_t *U, int16_t *V)
{
	short4v c[HVK_PIX_PASSES];
	raster_gather<LV, PASSES>(k, P, L, rgb, c);
	raster_stage<NT, WC, PASSES, 0>(k, L, t, nth, c, ghost_u, ghost_v, Yb, U, V);
}

/* 8 consecutive samples per lane, from the staged picture: s[] receives the line's samples (only their
 * low 16 bits count), cq[] the Q channel of --s-video. `lds` is the raster's whole LDS area (Y, U, V:
 * the SECAM notch and the VBI data lines re-use it). `slab_line`: the line's index in the raw baseband
 * slab (hvk_k_raster's blockIdx.x). Lanes at or beyond the line's end take part in the barriers. */
/* PREP: the line WITHOUT its sub-carrier, for the picture planes (hvk_k_prep): s[] is what the chroma
 * is added to, c[] receives the (V, U) pairs the modulator multiplies the phasors by (burst included). */
template<int NT, int SECAM, int SV, int EXTRAS, int WC, int PREP = 0>
__device__ __forceinline__ void raster_compute(const hvk_kconst_t &k, const hvk_rptrs_t &P, const hvk_line_t &L,
                                               const hvk_packed_taps_t &ctaps, const hvk_packed_taps_t &notch,
                                               const int y, const int slab_line, const int t, const int nth,
                                               int16_t *lds, const hvk_side_t &sd, int (&c)[SPL], int (&s)[SPL], int (&cq)[SPL])
{
	constexpr int H = NT / 2;
	const int W = WC ? WC : k.width;
	const int x0 = t * SPL;
	const int YL = raster_YL(W), CL = raster_CL(W);
	int16_t *Yb = lds, *U = lds + YL, *V = lds + YL + CL;
	const hvk_linedesc_t &d = L.d;
	const int pal = L.pal;
	const int rel = L.rel;
	const bool own = L.own, active = L.active, has_pix = L.has_pix;
	const int ax0 = L.ax0, ax1 = L.ax1, ar_eff = L.ar_eff;
	(void) rel; (void) own; (void) V;

	/* SECAM: picture lines and field identification lines carry the sub-carrier and get the luma notch */
	const bool sc_line = SECAM && (active || (EXTRAS && (d.secam_fid & 1)));

	/* the samples this WAVE covers, for wave-uniform (scalar) range tests */
	const int wx0 = __builtin_amdgcn_readfirstlane(x0);
	const int wx1 = wx0 + 64 * SPL;
#pragma unroll
	for(int i = 0; i < SPL; i++) cq[i] = 0;

	if(EXTRAS && k.rawbb)
	{
		/* raw baseband input (src/video.c:2431-2436): the line is taken from the external stream
		 * (`chroma` holds it, slab layout) and mapped from its levels onto the mode's; C integer
		 * arithmetic, division truncating */
#pragma unroll
		for(int i = 0; i < SPL; i++) s[i] = k.blanking;
		if(x0 < W)
		{
			const int16_t *in = P.chroma + ((size_t) y * k.slab_lines + slab_line) * W + x0;
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				if(x0 + i < W) s[i] = wrap16(k.blanking + (((int) in[i] - k.rawbb_blank) * (k.white - k.blanking)) / k.rawbb_range);
			}
		}
	}
	else
	{
		/* blanking level and sync pulses -- this line's own, and the part of the next line's left pulse
		 * that starts before its sample 0 (src/vbidata.c:211-216) -- summed modulo 2^16 once per kind of
		 * line by the host (hvk_tables.c:_build_linebase) */
		const int4v bv = sd.base;
		s[0] = (int) (short) (bv.x & 0xFFFF); s[1] = bv.x >> 16; s[2] = (int) (short) (bv.y & 0xFFFF); s[3] = bv.y >> 16;
		s[4] = (int) (short) (bv.z & 0xFFFF); s[5] = bv.z >> 16; s[6] = (int) (short) (bv.w & 0xFFFF); s[7] = bv.w >> 16;
	}

	/* luma is assigned over whatever is there (src/video.c:2961-3009):
	 * the picture where the frame covers the line, black elsewhere */
	if(active && x0 < ar_eff && x0 + SPL > d.al)
	{
		/* one 16-byte read of the lane's 8 luma values whether the picture covers them all or not
		 * (what it does not cover is not used): eight reads under lane tests would each be waited for */
		const int4v yq = *(const int4v *) (Yb + x0);
		const int yv[SPL] = { (int) (short) (yq.x & 0xFFFF), yq.x >> 16, (int) (short) (yq.y & 0xFFFF), yq.y >> 16,
		                      (int) (short) (yq.z & 0xFFFF), yq.z >> 16, (int) (short) (yq.w & 0xFFFF), yq.w >> 16 };
		if(x0 >= ax0 && x0 + SPL <= ax1)
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) s[i] = yv[i];
		}
		else
		{
			int black_y = k.black_y;
			asm volatile("" : "+s"(black_y));        /* one scalar load, not one per sample */
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int x = x0 + i;
				if(x >= d.al && x < ar_eff) s[i] = (x >= ax0 && x < ax1) ? yv[i] : black_y;
			}
		}
	}

	if(pal && x0 < W)
	{
		int vu[SPL];                            /* (V, U) packed int16: the dot2 operand of the modulator */

		/* zero-history low pass of both channels (src/fir.c:357-375), >> 15 and
		 * clamp by the saturating pack. All-zero input (no picture on this line)
		 * only matters where the ghost samples reach. */
		if((has_pix || x0 + SPL + H > W) && !ABLATE(2))
		{
			constexpr int ND = SPL / 2 + (NT + 1) / 2 + 1;
			int du[ND], dv[ND], u[SPL], v[SPL];
			const int4v *pu = (const int4v *) (U + x0), *pv = (const int4v *) (V + x0);
#pragma unroll
			for(int m = 0; m < (ND + 3) / 4; m++)
			{
				const int4v a = pu[m], b = pv[m];
				if(m * 4 + 0 < ND) { du[m * 4 + 0] = a.x; dv[m * 4 + 0] = b.x; }
				if(m * 4 + 1 < ND) { du[m * 4 + 1] = a.y; dv[m * 4 + 1] = b.y; }
				if(m * 4 + 2 < ND) { du[m * 4 + 2] = a.z; dv[m * 4 + 2] = b.z; }
				if(m * 4 + 3 < ND) { du[m * 4 + 3] = a.w; dv[m * 4 + 3] = b.w; }
			}
			fir8<NT, 0>(du, ctaps.p, u);
			fir8<NT, 0>(dv, ctaps.p, v);
#pragma unroll
			for(int i = 0; i < SPL; i++) vu[i] = sat_pack16(v[i] >> 15, u[i] >> 15);
		}
		else
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) vu[i] = 0;
		}

		/* colour burst replaces the filtered samples (src/video.c:3024-3029) */
		if(wx1 > k.burst_left && wx0 < k.burst_left + k.burst_width)
		if(x0 + SPL > k.burst_left && x0 < k.burst_left + k.burst_width)
		{
			const int4v bwv = sd.bwin;
			const int bw[SPL] = { (int) (short) (bwv.x & 0xFFFF), bwv.x >> 16, (int) (short) (bwv.y & 0xFFFF), bwv.y >> 16,
			                      (int) (short) (bwv.z & 0xFFFF), bwv.z >> 16, (int) (short) (bwv.w & 0xFFFF), bwv.w >> 16 };
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int b = x0 + i - k.burst_left;
				if(b >= 0 && b < k.burst_width)
				{
					const int w = bw[i];
					vu[i] = (((k.burst_q * w) >> 15) & 0xFFFF) | (((k.burst_i * w) >> 15) << 16);
				}
			}
		}

		/* quadrature modulation onto the sub-carrier (src/video.c:3032-3040):
		 *   s += (lut.i * V * pal + lut.q * U) >> 15
		 * as one dot2 of the packed table entry (i, q) with (V, U); the PAL switch
		 * negates lut.i, which never is -32768. */
		if(PREP)
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) c[i] = vu[i];
		}
		else if(pal < 0)
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) c[i] = (c[i] & 0xFFFF0000) | ((0 - c[i]) & 0xFFFF);
		}
		if(PREP) { }
		else if(SV)
		{
			/* S-Video: onto the (empty) Q channel instead of the luma (src/video.c:3032) */
#pragma unroll
			for(int i = 0; i < SPL; i++) cq[i] = wrap16(dot2z(c[i], vu[i]) >> 15);
		}
		else
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) s[i] = s[i] + (dot2z(c[i], vu[i]) >> 15);   /* modulo 2^16 at the store (no SECAM here: pal is 0 there) */
		}
	}

	if(sc_line)
	{
		/* SECAM lines with picture (src/video.c:3202-3229): first the luma notch
		 * over the active picture, a zero-history FIR whose input starts at
		 * active_left (everything left of it counts as zero) and which looks 25
		 * samples past the picture's right edge; then the sub-carrier -- the colour chain's
		 * output, hvk_secam.hip (or the host's hvk_secam.c) -- is added. */
		constexpr int NH = 25, NLEAD = 26;
		int16_t *Z = lds + YL;                  /* index j <-> sample x = j - NLEAD */

		if(!SV)                          /* S-Video leaves the luma alone (src/video.c:3206) */
		{
		if(t < 4) *(int4v *) (Z + t * 8) = (int4v) { 0, 0, 0, 0 };   /* Z does not overlap the picture's luma in LDS */
		{
			int4v z;
			int w[SPL];
#pragma unroll
			for(int i = 0; i < SPL; i++) w[i] = (x0 + i >= k.active_left) ? s[i] : 0;
			z.x = (w[0] & 0xFFFF) | (w[1] << 16); z.y = (w[2] & 0xFFFF) | (w[3] << 16);
			z.z = (w[4] & 0xFFFF) | (w[5] << 16); z.w = (w[6] & 0xFFFF) | (w[7] << 16);
			*(int4u *) (Z + NLEAD + x0) = (int4u) { z.x, z.y, z.z, z.w };
		}
		/* windows that run past the line belong to outputs right of the picture, which are not kept */
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
				if(x >= k.active_left && x < k.active_left + k.active_width) s[i] = clamp16(a[i] >> 15);
			}
		}
		}

		if(PREP) { }                   /* (the planes hold the line without its sub-carrier: hvk_k_direct adds the frame's) */
		else if(own && x0 + SPL <= W)
		{
			const int4u cv = *(const int4u *) (P.chroma + (size_t) L.chroma_row * k.raster_samples + (size_t) rel * W + x0);
			const int cw[4] = { cv.x, cv.y, cv.z, cv.w };
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int cs = (i & 1) ? (cw[i / 2] >> 16) : (int) (short) (cw[i / 2] & 0xFFFF);
				if(SV) cq[i] = cs;
				else s[i] = wrap16(s[i] + cs);
			}
		}
		else if(own && x0 < W)
		{
			/* the last, partial group of a line whose width is not a multiple of 8 */
			const int16_t *cp = P.chroma + (size_t) L.chroma_row * k.raster_samples + (size_t) rel * W + x0;
			for(int i = 0; i < SPL; i++)
			{
				if(x0 + i >= W) break;
				if(SV) cq[i] = cp[i];
				else s[i] = wrap16(s[i] + cp[i]);
			}
		}
	}

	if(EXTRAS && L.vits_i >= 0 && x0 < W)
	{
		/* insertion test signal (src/vits.c:270-311): the line's luma waveform is added; its
		 * chroma amplitude rides on the line's sub-carrier, rotated to the insertion phase */
		const int16_t *vl = P.vits_l + (size_t) L.vits_i * W + x0, *vc = P.vits_c + (size_t) L.vits_i * W + x0;
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			if(x0 + i < W)
			{
				s[i] = wrap16(s[i] + vl[i]);
				if(k.colour)
				{
					/* on a V-switched line the chroma stage has negated the table's i half in place */
					const int li = (pal < 0 ? -1 : 1) * (int) (short) (c[i] & 0xFFFF), lq = c[i] >> 16;
					s[i] = wrap16(s[i] + ((((k.vits_pi * lq + k.vits_pq * li) >> 15) * (int) vc[i]) >> 15));
				}
			}
		}
	}

	if(EXTRAS && L.fsc_flag >= 0 && x0 < W)
	{
		/* the field-sequential colour flag: a pulse added to the line (src/video.c:3043-3063) */
		const int16_t *fr = P.fsc_rows + (size_t) L.fsc_flag * W + x0;
#pragma unroll
		for(int i = 0; i < SPL; i++) if(x0 + i < W) s[i] = wrap16(s[i] + fr[i]);
	}

	/* the line's ops, in the reference's process order (an anti-copy line can also carry VITC) */
	if(EXTRAS)
	for(int opi = L.vbi_op; opi >= 0;)
	{
		/* One data line = up to 384 shaped symbols, each a run of samples added for every set
		 * bit (vbidata_render, src/vbidata.c:186-239). Set bits are walked by the whole
		 * workgroup (the data words are wave-uniform); lane t adds the t-th value of the
		 * symbol into an int32 line accumulator in LDS. */
		int *acc = (int *) lds;
		const unsigned *op = P.vbi_ops + ((size_t) y * HVK_VBI_OPS + opi) * HVK_VBI_OPWORDS;
		const int base_next = __builtin_amdgcn_readfirstlane((int) op[0]);
		const int sym_base = base_next & 0xFFFF;
		opi = (base_next >> 16) - 1;            /* next op of this line, -1: none */
		const int bits_mode = __builtin_amdgcn_readfirstlane((int) op[1]);
		const int nbits = bits_mode & 0xFFFF, mode = bits_mode >> 16;
		const int blank = __builtin_amdgcn_readfirstlane((int) op[2]);
		const int blank_lo = blank & 0xFFFF, blank_hi = blank >> 16;

		/* WSS first sets part of the line to black (src/wss.c:176-182) */
		if(blank_hi > blank_lo)
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) if(x0 + i >= blank_lo && x0 + i < blank_hi) s[i] = k.black;
		}

		if(mode == 1)
		{
			/* anti-copy pulse pairs (src/acp.c:113-126): twelve runs of samples SET to one of two levels */
			const int lv = __builtin_amdgcn_readfirstlane((int) op[3]);
			const int la = (int) (short) (lv & 0xFFFF), lb = lv >> 16;
			for(int q = 0; q < 12; q++)
			{
				const int seg = __builtin_amdgcn_readfirstlane((int) op[4 + q]);
				const int lo = seg & 0xFFFF, hi = (unsigned) seg >> 16;
#pragma unroll
				for(int i = 0; i < SPL; i++) if(x0 + i >= lo && x0 + i < hi) s[i] = (q & 1) ? lb : la;
			}
		}
		else if(nbits > 0 && P.vbi_cov && (op[3] >> 31))
		{
			/* The same sum as a GATHER. Which symbols lie over a sample, and with which values, depends on the table alone (a
			 * symbol's place is fixed, src/vbidata.c:36-81), and they are a run of consecutive symbols -- teletext's raised
			 * cosines are 2.3 samples apart and 54 to 59 long: up to 26 over one sample. The host lists per sample the run's
			 * first symbol, its length and the values (hvk_engine.cpp); a lane takes, for each of its 8 samples, the 32 bits of
			 * the line's data from that symbol on and adds the values whose bit is set: no accumulator in LDS, no atomics, no
			 * barrier, and no walk over the set bits with a round of 56 atomic adds each -- that walk was 100 of the 116 us
			 * the teletext lines of a 128-frame block took. The line's bits: word w in lane w of every wave, fetched across the
			 * wave by number (lanes from 12 on hold zeros: what lies in front of and behind the data). */
			const int o3 = __builtin_amdgcn_readfirstlane((int) op[3]);
			const int lut = o3 & 0xFF, first = (o3 >> 8) & 0xFFFF;
			unsigned wl = op[4 + ((t & 63) < 12 ? (t & 63) : 11)];
			{
				const int wlo = (t & 63) * 32;              /* (bits from nbits on are not rendered) */
				if(wlo + 32 > nbits) wl = wlo >= nbits ? 0u : (wl & ((1u << (nbits - wlo)) - 1u));
				if((t & 63) >= 12) wl = 0u;
			}
			/* (the lanes that hold the words, a wave's first twelve, have the wave's lowest samples: active whenever one is) */
			if(x0 < W)
			{
				const int4v *cv = (const int4v *) (P.vbi_cov + ((size_t) lut * W + x0) * 16);
				static_assert(HVK_VBI_COVER == 30, "a sample's entry: the run's start and length, then 30 values in 15 dwords");
#pragma unroll 2
				for(int i = 0; i < SPL; i++)
				{
					const int4v c0 = cv[i * 4 + 0], c1 = cv[i * 4 + 1], c2 = cv[i * 4 + 2], c3 = cv[i * 4 + 3];
					const int idx0 = (c0.x & 0xFFFF) - first;                           /* the run's first symbol as a bit of the line's data (may lie in front of it) */
					const unsigned w_lo = (unsigned) __shfl((int) wl, (idx0 >> 5) & 63), w_hi = (unsigned) __shfl((int) wl, ((idx0 >> 5) + 1) & 63);
					const unsigned mask = __builtin_amdgcn_alignbit(w_hi, w_lo, (unsigned) idx0 & 31u);
					const int vd[15] = { c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w };
					int a = 0;
#pragma unroll
					for(int m = 0; m < 15; m++)
					{
						a += (int) ((mask >> (2 * m)) & 1u) * (int) (short) (vd[m] & 0xFFFF);
						a += (int) ((mask >> (2 * m + 1)) & 1u) * (vd[m] >> 16);
					}
					if(x0 + i < W) s[i] = wrap16(s[i] + a);
				}
			}
		}
		else if(nbits > 0)
		{
			__syncthreads();
			for(int j = t; j < W; j += nth) acc[j] = 0;
			__syncthreads();

			/* set bits are dealt round-robin to the workgroup's waves: a symbol is a run of a few
			 * dozen samples, one wave's worth */
			const int nwaves = nth >> 6, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
			int turn = 0;
			for(int w = 0; w * 32 < nbits && w < 12; w++)
			{
				unsigned word = __builtin_amdgcn_readfirstlane(op[4 + w]);
				if(nbits - w * 32 < 32) word &= (1u << (nbits - w * 32)) - 1;
				while(word)
				{
					const int b = sym_base + w * 32 + __builtin_ctz(word);
					word &= word - 1;
					if(turn++ % nwaves != wave) continue;
					const int off = P.vbi_sym[b * 3 + 0], len = P.vbi_sym[b * 3 + 1];
					const int16_t *v = P.vbi_val + P.vbi_sym[b * 3 + 2];
					for(int j = lane; j < len; j += 64)
					{
						if(off + j >= 0 && off + j < W) atomicAdd(&acc[off + j], (int) v[j]);
					}
				}
			}
			__syncthreads();

			if(x0 < W)
			{
#pragma unroll
				for(int i = 0; i < SPL; i++) if(x0 + i < W) s[i] = wrap16(s[i] + acc[x0 + i]);
			}
		}
	}

	/* Sound-in-syncs (src/sis.c:155-215; behind CC608, in front of teletext, whose symbols lie elsewhere on the line): the
	 * sync area blanked to the sync level through a window, then the burst's half symbols added -- 46 or 50 bits, most
	 * significant first, bit b shaped by entry 50 - nb + b (vbidata_render() passes over the first 50 - nb). All of it
	 * lies in the line's first HVK_SIS_SPAN samples: the first wave's business. */
	if(EXTRAS && k.sis && (own || rel == k.lines || (k.rs_L && rel == k.lines + 1)) && wx0 < HVK_SIS_SPAN)
	{
		/* (also on the line behind the frame: the video filter of the frame's last samples looks into its first ones -- and
		 * behind the resampler, whose output stands a line back (k.rs_shift), into the first ones of the line after that) */
		const unsigned *rec = P.sis_bits + ((size_t) y * (k.lines + (k.rs_L ? 2 : 1)) + rel) * 2;
		const unsigned w0 = __builtin_amdgcn_readfirstlane(rec[0]), w1 = __builtin_amdgcn_readfirstlane(rec[1]);
		const int nb = (int) (w1 >> 24);
		if(x0 < HVK_SIS_SPAN)
		{
			int v[SPL];
#pragma unroll
			for(int i = 0; i < SPL; i++) v[i] = wrap16(s[i]);
			if(L.stream_first)
			{
				/* what the process left here when it ran on the never-emitted slot in front of this line (hvk_tables.c:_build_sis) */
				const int4v f4 = *(const int4v *) (P.sis_first + x0);
				const int fv[SPL] = { (int) (short) (f4.x & 0xFFFF), f4.x >> 16, (int) (short) (f4.y & 0xFFFF), f4.y >> 16,
				                      (int) (short) (f4.z & 0xFFFF), f4.z >> 16, (int) (short) (f4.w & 0xFFFF), f4.w >> 16 };
#pragma unroll
				for(int i = 0; i < SPL; i++) v[i] = wrap16(v[i] + fv[i]);
			}
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int x = x0 + i - k.sis_left;
				if(x >= 0 && x < k.sis_width)
				{
					const int w = P.sis_win[x];
					v[i] = wrap16((v[i] * (32767 - w) + k.sis_sync * w) >> 15);
				}
			}
			for(int b = 0; b < nb; b++)
			{
				const unsigned word = b < 32 ? w0 : w1;
				if(!((word >> (((b >> 3) & 3) * 8 + 7 - (b & 7))) & 1)) continue;       /* (the same for the whole wave) */
				const int4v q = *(const int4v *) (P.sis_dense + (size_t) (50 - nb + b) * HVK_SIS_SPAN + x0);
				v[0] += (int) (short) (q.x & 0xFFFF); v[1] += q.x >> 16; v[2] += (int) (short) (q.y & 0xFFFF); v[3] += q.y >> 16;
				v[4] += (int) (short) (q.z & 0xFFFF); v[5] += q.z >> 16; v[6] += (int) (short) (q.w & 0xFFFF); v[7] += q.w >> 16;
			}
#pragma unroll
			for(int i = 0; i < SPL; i++) s[i] = wrap16(v[i]);
		}
	}
}

/* ------------------------------------------------------------------ */
/* stages of a filter tile shared by hvk_k_filter and hvk_k_direct     */

/* One NICAM symbol slot of a tile's table (src/nicam728.c:33, :386-407): `v` is the host's word for
 * the slot (start << 3 | valid << 2 | value), n0 the tile's first sample. sym_st gets the start
 * relative to the tile, sym_ent { LEAD - start, offset of the shifted pulse copy, sign pair I, sign pair Q }. */
__device__ __forceinline__ void nicam_symbol_slot(const int v, const int n0, int *sym_st, int4v *sym_ent, const int slot, const int16_t *tapd)
{
	const int st = (v >> 3) - n0;
	const bool valid = (v & 4) && st < HVK_TILE;
	/* constellation { 0, 1, 3, 2 }: bit 0 -> +I else -I, bit 1 -> +Q else -Q
	 * (src/nicam728.c:33, :386-396) */
	const int cs = (0x2310 >> ((v & 3) * 4)) & 3;
	sym_st[slot] = valid ? st : 0x3FFFFFFF;
	/* x0 is a multiple of 8, so which of the shifted copies of the pulse table a lane needs depends on the symbol only
	 * (CM = copies - 1), and so does everything else of the address but x0 itself: the lane reads its 8 taps at BYTE
	 * offset min(2 x0 + e.x, e.y) of the table -- e.x = 2 (copy + (rel & ~CM)), rel = LEAD - start (x0 + (rel & ~CM) =
	 * (x0 + rel) & ~CM for a multiple of 8), e.y = 2 (copy + TAPD - 8): a pulse that is over clamps into its copy's zero
	 * tail. A slot without a symbol gets offsets that do so at once. Two vector instructions per symbol and lane instead
	 * of seven. */
	static_assert((HVK_NICAM_COPIES == 8 || HVK_NICAM_COPIES == 4) && HVK_NICAM_LEAD >= HVK_NICAM_COPIES - 1 && SPL == 8, "pulse table copies");
	constexpr int CM = HVK_NICAM_COPIES - 1;
	const int rel = HVK_NICAM_LEAD - st;
	/* (where the table lies in LDS goes into the offsets too: the low half of its flat address) */
	const int copy = (rel & CM) * HVK_NICAM_TAPD + (int) ((unsigned) (size_t) tapd >> 1);
	/* +1 or -1 in both halves: the pulse shapes two samples of a channel per packed multiply-add */
	const int sgi = (cs & 1) ? 0x00010001 : (int) 0xFFFFFFFFu;
	const int sgq = (cs & 2) ? 0x00010001 : (int) 0xFFFFFFFFu;
	const int none = (int) (unsigned) (size_t) tapd + 2 * (HVK_NICAM_TAPD - SPL);
	sym_ent[slot] = valid ? (int4v) { 2 * (copy + (rel & ~CM)), 2 * (copy + HVK_NICAM_TAPD - SPL), sgi, sgq }
	                      : (int4v) { none, none, 0, 0 };
}

/* NICAM onto a lane's 8 packed (I, Q) outputs: sum the pulses of the symbols in flight (int16
 * wrap-around per channel, both channels in one packed multiply-add), mix, add
 * (src/nicam728.c:350-365, :386-396). mix[0..1]: the mixer's first row (i, -q) of the lane's 8 samples,
 * mix[2..3]: its second row (q, i) -- both tabulated (hvk_engine.cpp). */
__device__ __forceinline__ void nicam_add(const hvk_kconst_t &k, const int x0, const int *sym_st, const int4v *sym_ent, const int16_t *tapd,
                                          const int4u (&mix)[4], int (&o)[SPL])
{
	const int last = x0 + SPL - 1;          /* relative to the tile's first sample */
	/* The newest symbol that has started by this lane's last sample. Slot HVK_NICAM_BACK - 1 holds the newest one at the
	 * tile's first sample (start st_b <= 0); the ones behind it follow at sps or sps - 1 samples each (src/nicam728.c:
	 * 398-407), so j of them have started where j = (last - st_b) / sps at least and, over a tile's length, one more at
	 * most (1100 * (1 / (sps - 1) - 1 / sps) < 1 for sps >= 34: every rate hvk_open takes): one look at the next slot
	 * decides. The division is a multiplication by ceil(2^20 / sps), exact below 2^20 / sps. No loop, no lane test. */
	int idx;
	{
		int a = last - sym_st[HVK_NICAM_BACK - 1];
		a = a > 0 ? a : 0;                      /* (a slot without a symbol: nothing has started) */
		const int j = (int) (((unsigned) a * (unsigned) k.nicam_inv20) >> 20);
		int i0 = HVK_NICAM_BACK - 1 + j;
		i0 = i0 < HVK_NICAM_SYMS - 2 ? i0 : HVK_NICAM_SYMS - 2;
		idx = i0 + (sym_st[i0 + 1] <= last ? 1 : 0);
	}

	/* I and Q apart while the pulses are summed: (I[2m], I[2m + 1]) and (Q[2m], Q[2m + 1]) */
	const int xb = 2 * x0;
	int bi[SPL / 2], bq[SPL / 2];
#pragma unroll
	for(int i = 0; i < SPL / 2; i++) bi[i] = bq[i] = 0;

	/* the newest symbol and the six before it: everything older is over. A pulse
	 * that is over (or a slot without a symbol) reads the zero tail of the table:
	 * no branch. idx >= HVK_NICAM_BACK - 1 by construction. */
	/* (the next symbol's entry is asked for before this one's taps are: one LDS round trip per symbol in the chain of
	 * dependent reads, not two) */
#if HVK_NICAM_COPIES == 8
	typedef __attribute__((address_space(3))) const int4v *lds_taps_t;
#else
	typedef __attribute__((address_space(3))) const int2v *lds_taps_t;
#endif
	const int4v *ep = sym_ent + idx;
	int4v en = ep[0];
#pragma unroll 1
	for(int b = 0; b < (ABLATE(32) ? 0 : HVK_NICAM_BACK); b++)
	{
		/* (the entry in front of the oldest symbol -- one in front of the table where idx = HVK_NICAM_BACK - 1: the callers'
		 * tables have an entry of slack there -- is read and not used) */
		ep--;
		const int4v nx = ep[0];
		int at = xb + en.x;                                     /* (an LDS byte address; the symbol has started by the lane's last sample) */
		at = at < en.y ? at : en.y;
		const lds_taps_t tp = (lds_taps_t) (size_t) (unsigned) at;
#if HVK_NICAM_COPIES == 8
		const int4v ta = tp[0];
#else
		const int2v ta0 = tp[0], ta1 = tp[1];
		const int4v ta = { ta0.x, ta0.y, ta1.x, ta1.y };
#endif
		bi[0] = pk_mad16(ta.x, en.z, bi[0]); bi[1] = pk_mad16(ta.y, en.z, bi[1]);
		bi[2] = pk_mad16(ta.z, en.z, bi[2]); bi[3] = pk_mad16(ta.w, en.z, bi[3]);
		bq[0] = pk_mad16(ta.x, en.w, bq[0]); bq[1] = pk_mad16(ta.y, en.w, bq[1]);
		bq[2] = pk_mad16(ta.z, en.w, bq[2]); bq[3] = pk_mad16(ta.w, en.w, bq[3]);
		en = nx;
	}

	int bb[SPL];                            /* (I, Q) of each sample */
#pragma unroll
	for(int m = 0; m < SPL / 2; m++)
	{
		bb[2 * m + 0] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x05040100u);
		bb[2 * m + 1] = (int) __builtin_amdgcn_perm((unsigned) bq[m], (unsigned) bi[m], 0x07060302u);
	}

	if(!ABLATE(64))
	{
		const int ca[SPL] = { mix[0].x, mix[0].y, mix[0].z, mix[0].w, mix[1].x, mix[1].y, mix[1].z, mix[1].w };
		const int cq[SPL] = { mix[2].x, mix[2].y, mix[2].z, mix[2].w, mix[3].x, mix[3].y, mix[3].z, mix[3].w };
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			const int mi = dot2z(bb[i], ca[i]);           /* bb.i * cc.i - bb.q * cc.q */
			const int mq = dot2z(bb[i], cq[i]);           /* bb.i * cc.q + bb.q * cc.i */
			/* ((mi >> 15) & 0xFFFF) | ((mq >> 15) << 16) */
			const int pk = (int) ((((unsigned) mq << 1) & 0xFFFF0000u) | (((unsigned) mi >> 15) & 0xFFFFu));
			o[i] = pk_add16(o[i], pk);
		}
	}
}

/* the mixer rows of a lane's 8 samples from table position cp: row (i, -q) at nicam_cca[cp ..], row (q, i) at
 * nicam_cca[rows + cp ..] (`rows` = cc_len + 8 entries each) */
__device__ __forceinline__ void nicam_mix_rows(const int *nicam_cca, const int rows, const int cp, int4u (&mix)[4])
{
	mix[0] = ((const int4u *) (nicam_cca + cp))[0]; mix[1] = ((const int4u *) (nicam_cca + cp))[1];
	mix[2] = ((const int4u *) (nicam_cca + rows + cp))[0]; mix[3] = ((const int4u *) (nicam_cca + rows + cp))[1];
}

/* A lane's 8 window samples (four dwords of int16 pairs) as 8 bytes of the high-byte plane and 8 of the
 * low-byte plane (low bytes less 128: read as signed after ^ 0x80) -- the operands of the int8 matrix unit */
__device__ __forceinline__ void split_planes(const int4u d, int2v &ph, int2v &pl)
{
	ph.x = (int) __builtin_amdgcn_perm((unsigned) d.y, (unsigned) d.x, 0x07050301u);
	ph.y = (int) __builtin_amdgcn_perm((unsigned) d.w, (unsigned) d.z, 0x07050301u);
	pl.x = (int) (__builtin_amdgcn_perm((unsigned) d.y, (unsigned) d.x, 0x06040200u) ^ 0x80808080u);
	pl.y = (int) (__builtin_amdgcn_perm((unsigned) d.w, (unsigned) d.z, 0x06040200u) ^ 0x80808080u);
}

/* The 51-tap filter of one wave's 512 outputs as a banded matrix product on the matrix unit. A wave takes
 * 64 segments of 8 outputs, 16 segments (the columns of B) per v_mfma_i32_16x16x64_i8: lane (g, c) hands
 * over window positions 16 g .. 16 g + 15 of segment c, which are 16 consecutive bytes of a plane, and gets
 * back rows 4 g .. 4 g + 3 = outputs 2 g, 2 g + 1 of that segment, I and Q. Four products (high / low byte
 * of taps and samples), recombined with two shift-adds; the constant of the low plane's offset starts the
 * low accumulator. Then >> 15 and the saturating pack (src/fir.c:605-608), into `outl` (packed I/Q, indexed
 * by output) for the lane that owns the 8 outputs. xh / xl: the planes, position 0 = 26 samples before the
 * tile's first output; t: lane of the tile (0 .. 127). */
/* emit(j, seg, g, pk): the two finished outputs 2 g, 2 g + 1 of segment seg = 64 (t >> 6) + 16 j + c (packed I/Q each), in the lane the matrix unit leaves them in */
template<class EMIT>
__device__ __forceinline__ void mfma_filter_each(const unsigned char *xh, const unsigned char *xl, const int t,
                                                 const int4v a_hh, const int4v a_hl, const int mfma_ci, const int mfma_cq, EMIT &&emit)
{
	const int lane = t & 63, g = lane >> 4, c = lane & 15;
#pragma unroll
	for(int j = 0; j < 4; j++)
	{
		const int seg = (t >> 6) * 64 + j * 16 + c;
		const int off = seg * 8 + g * 16;
		int4v bh, bl;
		bh.xy = *(const int2v *) (xh + off); bh.zw = *(const int2v *) (xh + off + 8);
		bl.xy = *(const int2v *) (xl + off); bl.zw = *(const int2v *) (xl + off + 8);
		int4v p_hh = { 0, 0, 0, 0 }, p_m = { 0, 0, 0, 0 }, p_ll = { mfma_ci, mfma_cq, mfma_ci, mfma_cq };
		p_hh = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bh, p_hh, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bh, p_m, 0, 0, 0);
		p_m  = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hh, bl, p_m, 0, 0, 0);
		p_ll = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hl, bl, p_ll, 0, 0, 0);
		int y[4];
#pragma unroll
		for(int i = 0; i < 4; i++)
		{
			/* Horner: two v_lshl_add_u32 (left alone the compiler makes it two shifts and a three-operand add) */
			unsigned hm = ((unsigned) p_hh[i] << 8) + (unsigned) p_m[i];
#ifndef HVK_V_NO_HORNER
			asm("" : "+v"(hm));
#endif
			y[i] = (int) ((hm << 8) + (unsigned) p_ll[i]);
		}
		int2v pk;
		pk.x = sat_pack16(y[0] >> 15, y[1] >> 15);
		pk.y = sat_pack16(y[2] >> 15, y[3] >> 15);
		emit(j, seg, g, pk);
	}
}

__device__ __forceinline__ void mfma_filter(const unsigned char *xh, const unsigned char *xl, int *outl, const int t,
                                            const int4v a_hh, const int4v a_hl, const int mfma_ci, const int mfma_cq)
{
	mfma_filter_each(xh, xl, t, a_hh, a_hl, mfma_ci, mfma_cq, [&](const int, const int seg, const int g, const int2v pk) { *(int2v *) (outl + seg * 8 + 2 * g) = pk; });
}

#endif
