/* hvk_kernels.hip -- CDNA4 (gfx950) kernels of the composite-video -> IQ engine.
 *
 * Three kernels, all integer except the one-off table expansion:
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
 *   hvk_k_filter       one workgroup per 1024 output samples (= one PAL line
 *                      at 16 Msps), 8 consecutive outputs per lane. Stages
 *                      1024 + 50 raster samples in LDS, runs the 51-tap
 *                      real->complex VSB filter (or real low-pass) with
 *                      v_dot2c_i32_i16 on packed sample pairs against taps
 *                      held in SGPRs (src/fir.c:564-615, :304-355), adds the
 *                      serial-carrier side stream and the NICAM DQPSK signal
 *                      (pulse overlap-add + mixer, src/nicam728.c:342-411),
 *                      and stores interleaved int16 I/Q, 32 bytes per lane.
 *
 * Nothing here is a dense contraction: no MFMA. The work is bounded by VALU
 * issue (dot2 count) and by the 4 B/sample HBM write.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hvk_internal.h"
#include "hvk_kernels.h"

typedef short  short2v __attribute__((ext_vector_type(2)));
typedef short  short4v __attribute__((ext_vector_type(4)));
typedef int    int4v   __attribute__((ext_vector_type(4)));
typedef int    int2v   __attribute__((ext_vector_type(2)));

#define SPL HVK_SPL

__device__ __forceinline__ int wrap16(int v) { return((int) (short) v); }
__device__ __forceinline__ int clamp16(int v) { return(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
__device__ __forceinline__ int dot2(int a, int b, int c)
{
	return(__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false));
}
/* (lo >> 16) | (hi << 16): the pair of int16 that starts one element later */
__device__ __forceinline__ int shift_pair(int lo, int hi) { return((int) __builtin_amdgcn_alignbit((unsigned) hi, (unsigned) lo, 16)); }
__device__ __forceinline__ int floordiv(int a, int b) { int q = a / b; return((a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q); }

/* 8 consecutive FIR outputs from a register window of packed int16 pairs.
 * d[] holds window elements w[0..], two per dword; output i is
 *   sum_k tap[k] * w[START + i + k]
 * with START in {0,1}. tp[] holds the NT taps packed two per dword and zero
 * padded. Outputs whose first element is dword aligned use d[] directly, the
 * others use the pairs shifted by one element. */
template<int NT, int START>
__device__ __forceinline__ void fir8(const int *d, const int *tp, int (&acc)[SPL])
{
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

__global__ void hvk_k_expand_yuv(short4v *lut, const hvk_yuvparams_t *pp)
{
	const hvk_yuvparams_t &p = *pp;
	unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
	if(c > 0xFFFFFFu) return;

	/* src/video.c:3917-3958, same order of operations, no contraction */
	double r = p.glut[(c & 0xFF0000) >> 16];
	double g = p.glut[(c & 0x00FF00) >> 8];
	double b = p.glut[(c & 0x0000FF) >> 0];
	double y, u, v;

	y = r * p.rw + g * p.gw + b * p.bw;
	u = (b - y) * p.eu;
	v = (r - y) * p.ev;

	y = (p.black + (y * p.range)) * p.level;
	u *= p.chroma_scale;
	v *= p.chroma_scale;

	y = y < -1 ? -1 : (y > 1 ? 1 : y);
	u = u < -1 ? -1 : (u > 1 ? 1 : u);
	v = v < -1 ? -1 : (v > 1 ? 1 : v);

	short4v o;
	o.x = (short) round(y * 32767);
	o.y = (short) round(u * 32767);
	o.z = (short) round(v * 32767);
	o.w = 0;
	lut[c] = o;
}

/* ------------------------------------------------------------------ */

/* LDS layout of the raster kernel: U and V channel arrays of
 * HVK_CHROMA_LEAD + width + HVK_CHROMA_LEAD int16 each. Element index
 * j <-> sample x = j - h (h = ntaps / 2), so a lane's FIR window starts at
 * its own first sample index: 16-byte aligned. */
template<int NT>
__global__ __launch_bounds__(1024)
void hvk_k_raster(const hvk_kconst_t k,
                  const hvk_packed_taps_t ctaps,
                  const hvk_linedesc_t *__restrict__ desc,
                  const int16_t *__restrict__ pulses,
                  const short4v *__restrict__ yuv,
                  const int *__restrict__ clut,
                  const int16_t *__restrict__ burst_win,
                  const int16_t *__restrict__ ghost,
                  const uint32_t *__restrict__ pool,
                  const hvk_framedesc_t *__restrict__ fdesc,
                  int16_t *__restrict__ S)
{
	extern __shared__ __attribute__((aligned(16))) int16_t lds[];

	const int W = k.width;
	const int t = threadIdx.x;
	const int x0 = t * SPL;
	const hvk_framedesc_t &f = fdesc[blockIdx.y];
	const int64_t g = f.frame_index * k.lines + (int) blockIdx.x - 1;   /* global line */
	int16_t *out = S + ((size_t) blockIdx.y * (k.lines + 2) + blockIdx.x) * W;

	int s[SPL];

	if(g < 0)
	{
		/* before the stream: the filter history is zero, not blanking
		 * (src/video.c:4665-4667 with src/fir.c:289, :579) */
		for(int i = 0; i < SPL; i++) if(x0 + i < W) out[x0 + i] = 0;
		return;
	}

	const int64_t frame0 = g / k.lines;
	const int line0 = (int) (g - frame0 * k.lines);
	const hvk_linedesc_t d = desc[((frame0 + 1) & 1) * k.lines + line0];
	const bool own = frame0 == f.frame_index;   /* halo lines carry no picture */

#pragma unroll
	for(int i = 0; i < SPL; i++) s[i] = k.blanking;

	/* sync pulses: this line's own, and the part of the next line's left
	 * pulse that starts before its sample 0 (src/vbidata.c:211-216) */
	{
		const int ids[3] = { d.pulse_left, d.pulse_mid, d.pulse_next };
#pragma unroll
		for(int p = 0; p < 3; p++)
		{
			const int id = ids[p];
			if(id < 0) continue;
			const int off = k.pulse_offset[id] + (p == 2 ? W : 0);
			const int len = k.pulse_length[id];
			const int16_t *v = pulses + k.pulse_start[id];
			if(x0 + SPL <= off || x0 >= off + len) continue;
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int idx = x0 + i - off;
				/* a pulse never crosses into the following line; the part of
				 * the own left pulse before sample 0 belongs to the previous line */
				if(idx >= 0 && idx < len && x0 + i < W) s[i] = wrap16(s[i] + v[idx]);
			}
		}
	}

	const int pal = k.colour ? d.pal : 0;
	constexpr int H = NT / 2;
	const int CL = W + 2 * HVK_CHROMA_LEAD;     /* channel length in LDS */
	int16_t *U = lds, *V = lds + CL;

	if(pal)
	{
		/* clear both channels, then place the samples the reference reads
		 * past the end of its buffer */
		for(int j = t * 8; j < 2 * CL; j += blockDim.x * 8) *(int4v *) (lds + j) = (int4v) { 0, 0, 0, 0 };
		__syncthreads();
		if(t < H)
		{
			U[H + W + t] = ghost[2 * t + 0];
			V[H + W + t] = ghost[2 * t + 1];
		}
	}

	/* active picture: luma is assigned over whatever is there (src/video.c:2961-3009) */
	if(d.ar > d.al && x0 < d.ar && x0 + SPL > d.al)
	{
		int vy = d.src_row;
		if(vy >= 0 && k.interlaced != 0 && f.fb_interlaced != k.interlaced) vy += 1;
		vy -= f.vframe_y;
		if(vy < 0 || vy >= f.fb_height || !own || !f.fb_valid) vy = -1;

		const uint32_t *row = pool + f.fb_offset + (int64_t) vy * f.line_stride;
		const int px0 = k.active_left + f.vframe_x;

#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			const int x = x0 + i;
			if(x < d.al || x >= d.ar) continue;
			const int px = x - px0;
			if(px >= 0 && px < f.fb_width)
			{
				const uint32_t rgb = vy >= 0 ? (row[(int64_t) px * f.pixel_stride] & 0xFFFFFFu) : 0u;
				const short4v c = yuv[rgb];
				s[i] = c.x;
				if(pal)
				{
					U[H + x] = c.y;
					V[H + x] = c.z;
				}
			}
			else s[i] = k.black_y;
		}
	}

	if(pal)
	{
		__syncthreads();

		if(x0 < W)
		{
			int u[SPL], v[SPL];

			/* zero-history low pass of both channels (src/fir.c:357-375) */
			{
				constexpr int ND = SPL / 2 + (NT + 1) / 2 + 1;
				int du[ND], dv[ND];
				const int *pu = (const int *) (U + x0), *pv = (const int *) (V + x0);
#pragma unroll
				for(int m = 0; m < ND; m++) { du[m] = pu[m]; dv[m] = pv[m]; }
				fir8<NT, 0>(du, ctaps.p, u);
				fir8<NT, 0>(dv, ctaps.p, v);
#pragma unroll
				for(int i = 0; i < SPL; i++) { u[i] = clamp16(u[i] >> 15); v[i] = clamp16(v[i] >> 15); }
			}

			/* colour burst replaces the filtered samples (src/video.c:3024-3029) */
			if(x0 + SPL > k.burst_left && x0 < k.burst_left + k.burst_width)
			{
#pragma unroll
				for(int i = 0; i < SPL; i++)
				{
					const int b = x0 + i - k.burst_left;
					if(b >= 0 && b < k.burst_width)
					{
						const int w = burst_win[b];
						u[i] = wrap16((k.burst_i * w) >> 15);
						v[i] = wrap16((k.burst_q * w) >> 15);
					}
				}
			}

			/* quadrature modulation onto the sub-carrier (src/video.c:3032-3040);
			 * the table position advances by one line per line, colour or not */
			const unsigned coff = (unsigned) (((uint64_t) g * (uint64_t) W) % k.clw);
			const int *cl = clut + coff + x0;
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				if(x0 + i < W)
				{
					const int c = cl[i];
					const int ci = (int) (short) (c & 0xFFFF), cq = c >> 16;
					s[i] = wrap16(s[i] + ((ci * v[i] * pal + cq * u[i]) >> 15));
				}
			}
		}
	}

	if(x0 + SPL <= W)
	{
		int4v o;
		o.x = (s[0] & 0xFFFF) | (s[1] << 16);
		o.y = (s[2] & 0xFFFF) | (s[3] << 16);
		o.z = (s[4] & 0xFFFF) | (s[5] << 16);
		o.w = (s[6] & 0xFFFF) | (s[7] << 16);
		if((((size_t) (out + x0)) & 15) == 0) *(int4v *) (out + x0) = o;
		else
		{
			int *q = (int *) (out + x0);
			q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = o.w;
		}
	}
	else
	{
		for(int i = 0; i < SPL; i++) if(x0 + i < W) out[x0 + i] = (int16_t) s[i];
	}
}

/* ------------------------------------------------------------------ */

/* Start of symbol j relative to the frame's anchor symbol:
 * sps * j - floor((ph + j * dsl) / decimation)   (src/nicam728.c:400-407) */
__device__ __forceinline__ int nicam_rel_start(const hvk_kconst_t &k, int ph, int j)
{
	return(k.nicam_sps * j - floordiv(ph + j * k.nicam_dsl, k.nicam_decimation));
}

template<int NT, int VF>
__global__ __launch_bounds__(HVK_TILE / HVK_SPL)
void hvk_k_filter(const hvk_kconst_t k,
                  const hvk_packed_taps_t itaps,
                  const hvk_packed_taps_t qtaps,
                  const hvk_framedesc_t *__restrict__ fdesc,
                  const int16_t *__restrict__ S,
                  const int *__restrict__ carriers,      /* [frames][frame_samples] int16 pairs */
                  const uint8_t *__restrict__ symbols,   /* [frames][symbol_stride] */
                  const int symbol_stride,
                  const int16_t *__restrict__ nicam_taps,
                  const int *__restrict__ nicam_cc,
                  int *__restrict__ iq,                  /* [frames * out_stride][frame_samples] int16 pairs */
                  const int64_t out_stride)
{
	constexpr int H = NT / 2;
	constexpr int LEAD = H + (H & 1);           /* window lead, even */
	constexpr int NWIN = HVK_TILE + 2 * LEAD + 16;
	__shared__ __attribute__((aligned(16))) int16_t win[NWIN];
	__shared__ __attribute__((aligned(16))) int16_t ntaps_lds[256];

	const int W = k.width;
	const int FS = k.frame_samples;
	const int t = threadIdx.x;
	const int n0 = blockIdx.x * HVK_TILE;       /* first output sample of the tile, frame local */
	const int x0 = t * SPL;
	const hvk_framedesc_t &f = fdesc[blockIdx.y];
	const int16_t *slab = S + (size_t) blockIdx.y * (k.lines + 2) * W + W;   /* frame local sample 0 */

	/* stage raster samples [n0 - LEAD, n0 + TILE + LEAD) as dwords; the slab
	 * keeps one line before and one after the frame */
	if(VF != 0)
	{
		const int *src = (const int *) (slab + n0 - LEAD);   /* 4-byte aligned: W, TILE, LEAD even */
		const int limit = (FS + W - (n0 - LEAD)) / 2;         /* dwords available in the slab */
		for(int q = t; q < NWIN / 2; q += blockDim.x) ((int *) win)[q] = q < limit ? src[q] : 0;
	}
	if(k.has_nicam) for(int q = t; q < k.nicam_ntaps; q += blockDim.x) ntaps_lds[q] = nicam_taps[q];
	__syncthreads();

	const int n = n0 + x0;                      /* this lane's first output, frame local */
	if(n >= FS) return;

	int oi[SPL], oq[SPL];

	if(VF != 0)
	{
		constexpr int ND = SPL / 2 + (NT + 1) / 2 + 1;
		int d[ND];
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
		fir8<NT, LEAD - H>(d, itaps.p, oi);
#pragma unroll
		for(int i = 0; i < SPL; i++) oi[i] = clamp16(oi[i] >> 15);

		if(VF == 3)
		{
			fir8<NT, LEAD - H>(d, qtaps.p, oq);
#pragma unroll
			for(int i = 0; i < SPL; i++) oq[i] = clamp16(oq[i] >> 15);
		}
		else
		{
#pragma unroll
			for(int i = 0; i < SPL; i++) oq[i] = 0;
		}
	}
	else
	{
		/* no filter: the raster goes straight to I, Q = 0 */
		const int16_t *p = slab + n;
#pragma unroll
		for(int i = 0; i < SPL; i++) { oi[i] = (n + i < FS) ? p[i] : 0; oq[i] = 0; }
	}

	const size_t cbase = (size_t) blockIdx.y * FS + n;
	const size_t obase = (size_t) blockIdx.y * out_stride * FS + n;

	/* serial carriers (FM / AM sound), computed on the host: a plain add
	 * (src/video.c:3431-3432) */
	if(k.has_carriers)
	{
		const int *c = carriers + cbase;
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			if(n + i < FS)
			{
				const int v = c[i];
				oi[i] = wrap16(oi[i] + (int) (short) (v & 0xFFFF));
				oq[i] = wrap16(oq[i] + (v >> 16));
			}
		}
	}

	/* NICAM: sum the pulses of the symbols in flight, mix, add
	 * (src/nicam728.c:350-365, :386-396) */
	if(k.has_nicam)
	{
		const int r0 = f.nicam_rf + n;          /* position relative to the anchor symbol's start */
		const int period = k.nicam_sps * k.nicam_decimation - k.nicam_dsl;
		const int ph = f.nicam_ph;
		const uint8_t *sym = symbols + (size_t) blockIdx.y * symbol_stride;
		const int kbase = (int) (f.nicam_kf - f.nicam_k0);   /* slab index of the anchor symbol */

		/* newest symbol that has started by this lane's last sample */
		int j = (int) (((int64_t) (r0 + SPL - 1) * k.nicam_decimation) / period);
		while(nicam_rel_start(k, ph, j + 1) <= r0 + SPL - 1) j++;
		while(nicam_rel_start(k, ph, j) > r0 + SPL - 1) j--;

		int bi[SPL], bq[SPL];
#pragma unroll
		for(int i = 0; i < SPL; i++) bi[i] = bq[i] = 0;

		for(int back = 0; back < 7; back++)
		{
			const int jj = j - back;
			const int st = nicam_rel_start(k, ph, jj);
			if(r0 + SPL - 1 - st < 0) continue;
			if(r0 - st >= k.nicam_ntaps) break;
			const int si = kbase + jj;
			if(si < 0) break;
			const unsigned sv = sym[si];
			if(sv == 0xFF) break;               /* before the first symbol of the stream */
			/* constellation { 0, 1, 3, 2 }: bit 0 -> I sign, bit 1 -> Q sign */
			const int cs = (0x2310 >> (sv * 4)) & 3;
			const int sgi = (cs & 1) ? 1 : -1, sgq = (cs & 2) ? 1 : -1;
#pragma unroll
			for(int i = 0; i < SPL; i++)
			{
				const int idx = r0 + i - st;
				if(idx >= 0 && idx < k.nicam_ntaps)
				{
					const int tp = ntaps_lds[idx];
					bi[i] += sgi * tp;
					bq[i] += sgq * tp;
				}
			}
		}

		int cpos = (int) ((f.nicam_cc0 + (int64_t) n) % k.nicam_cc_len);
#pragma unroll
		for(int i = 0; i < SPL; i++)
		{
			const int c = nicam_cc[cpos];
			const int ci = (int) (short) (c & 0xFFFF), cq = c >> 16;
			const int b_i = wrap16(bi[i]), b_q = wrap16(bq[i]);
			oi[i] = wrap16(oi[i] + ((b_i * ci - b_q * cq) >> 15));
			oq[i] = wrap16(oq[i] + ((b_i * cq + b_q * ci) >> 15));
			if(++cpos == k.nicam_cc_len) cpos = 0;
		}
	}

	/* interleaved int16 I/Q, 32 bytes per lane */
	int *o = iq + obase;
	if(n + SPL <= FS && ((((size_t) o) & 15) == 0))
	{
		int4v a, b;
		a.x = (oi[0] & 0xFFFF) | (oq[0] << 16); a.y = (oi[1] & 0xFFFF) | (oq[1] << 16);
		a.z = (oi[2] & 0xFFFF) | (oq[2] << 16); a.w = (oi[3] & 0xFFFF) | (oq[3] << 16);
		b.x = (oi[4] & 0xFFFF) | (oq[4] << 16); b.y = (oi[5] & 0xFFFF) | (oq[5] << 16);
		b.z = (oi[6] & 0xFFFF) | (oq[6] << 16); b.w = (oi[7] & 0xFFFF) | (oq[7] << 16);
		((int4v *) o)[0] = a;
		((int4v *) o)[1] = b;
	}
	else
	{
#pragma unroll
		for(int i = 0; i < SPL; i++) if(n + i < FS) o[i] = (oi[i] & 0xFFFF) | (oq[i] << 16);
	}
}

/* ------------------------------------------------------------------ */
/* launchers                                                           */

extern "C" int hvk_launch_expand_yuv(void *lut, const void *params, hipStream_t stream)
{
	hipLaunchKernelGGL(hvk_k_expand_yuv, dim3(0x1000000 / 256), dim3(256), 0, stream,
	                   (short4v *) lut, (const hvk_yuvparams_t *) params);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

template<int NT>
static int _launch_raster(const hvk_raster_args_t *a, hipStream_t stream)
{
	const int W = a->k.width;
	int threads = (W + SPL - 1) / SPL;
	threads = (threads + 63) / 64 * 64;
	const size_t lds = (size_t) 2 * (W + 2 * HVK_CHROMA_LEAD) * sizeof(int16_t) + 64;
	hipLaunchKernelGGL(hvk_k_raster<NT>, dim3(a->k.lines + 2, a->nframes), dim3(threads), lds, stream,
	                   a->k, a->ctaps, a->desc, a->pulses, (const short4v *) a->yuv, (const int *) a->clut,
	                   a->burst_win, a->ghost, a->pool, a->fdesc, a->S);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_raster(const hvk_raster_args_t *a, hipStream_t stream)
{
	switch(a->k.colour ? a->k.chroma_ntaps : 1)
	{
	case 1:  return(_launch_raster<1>(a, stream));   /* no chroma filter (taps = {32767}) or monochrome */
	case 9:  return(_launch_raster<9>(a, stream));
	case 11: return(_launch_raster<11>(a, stream));
	case 13: return(_launch_raster<13>(a, stream));
	case 15: return(_launch_raster<15>(a, stream));
	case 17: return(_launch_raster<17>(a, stream));
	case 21: return(_launch_raster<21>(a, stream));
	}
	return(HVK_UNSUPPORTED);
}

template<int NT, int VF>
static int _launch_filter(const hvk_filter_args_t *a, hipStream_t stream)
{
	const int tiles = (a->k.frame_samples + HVK_TILE - 1) / HVK_TILE;
	hipLaunchKernelGGL((hvk_k_filter<NT, VF>), dim3(tiles, a->nframes), dim3(HVK_TILE / SPL), 0, stream,
	                   a->k, a->itaps, a->qtaps, a->fdesc, a->S, (const int *) a->carriers, a->symbols,
	                   a->symbol_stride, a->nicam_taps, (const int *) a->nicam_cc, (int *) a->iq, a->out_stride);
	return(hipGetLastError() == hipSuccess ? HVK_OK : HVK_ERROR);
}

extern "C" int hvk_launch_filter(const hvk_filter_args_t *a, hipStream_t stream)
{
	if(a->k.vf_type == 0) return(_launch_filter<1, 0>(a, stream));
	if(a->k.vf_ntaps != 51) return(HVK_UNSUPPORTED);
	if(a->k.vf_type == 1) return(_launch_filter<51, 1>(a, stream));
	if(a->k.vf_type == 3) return(_launch_filter<51, 3>(a, stream));
	return(HVK_UNSUPPORTED);
}
