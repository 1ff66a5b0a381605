/* hvk_engine_launch.cpp -- the launch of a staged batch: which kernels render it (picture planes + hvk_k_direct, hvk_k_fused from
 * the pixels, or the raster [+ resampler] + filter pair; DESIGN.md section 2 has the table), hvk_render*, hvk_sync. */
#include "hvk_engine_priv.h"

extern "C" int hvk_set_stream(hvk_engine_t *e, void *hip_stream)
{
	if(!e) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipStreamSynchronize(e->stream));
	e->stream = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
	return(HVK_OK);
}

/* the kernels' arguments for the staged batch */
/* S-Video behind resampler + video filter, lines of two widths (hvk_kconst_t.sv_ring): the staged batch's Q channel, line by
 * line, as the reference's ring of line buffers pairs it (hvk_k_svq has the rule). Emitted line j of the stream begins at
 * S(j) = ceil((j + s) W L / D) - ceil(s W L / D), s the chunks dropped at start-up (hvk_tables_frame_start()); its content
 * is a chunk of the width of line j - 1, which began delta = w(-1) - w(j - 1) samples behind S(j) in the sub-carrier stream. */
int hvk_e_sv_ring_records(hvk_engine *e, int64_t first_frame, int nframes)
{
	/* Made when the batch is STAGED (hvk_stage_strided*), with its other side inputs: the one pinned buffer is rewritten only
	 * after ev_staged of the stage before -- which this copy is in front of -- so a launch never reads another batch's
	 * records, however far stage / launch / fetch calls are pipelined. */
	const hvk_kconst_t &k = e->t.k;
	const int64_t s = 1 + (k.vf_type ? k.delay_lines : 0), WL = (int64_t) k.width * k.rs_L, D = k.rs_D;
	auto S = [&](int64_t j) { return(((j + s) * WL + D - 1) / D - (s * WL + D - 1) / D); };
	auto width = [&](int64_t j) { return((int) (S(j + 1) - S(j))); };
	const int wmax = e->t.max_width;
	/* (the luma stream lags the chunks by the first chunk the filter was fed and gave nothing for -- the last one dropped at
	 * start-up: ITS width is what a content chunk's width is held against, the longer one or the shorter one as the rates have it) */
	const int wref = s >= 1 ? width(-1) : wmax;
	const int64_t f0 = first_frame, j0 = f0 * k.lines, base = S(j0);
	const long slab_in = (long) k.slab_lines * k.width;
	const int nlines = nframes * k.lines;

	for(int i = 0; i < nlines; i++)
	{
		const int64_t j = j0 + i;
		const int w = width(j), wp = j + s - 1 >= 0 ? width(j - 1) : wref, delta = wref - wp;       /* -1, 0 or 1 */
		int kind = 0, src = 0;
		if(w > wp)
		{
			if(k.rs_L < k.rs_D)
			{
				/* downwards: the raster's sub-carrier of the line before the content's, at the place the content ends */
				const int y = i / k.lines;
				const int64_t pl = S(j) - S((f0 + y) * k.lines);                    /* the line's first sample in its frame */
				const int64_t rr = pl + delta + k.rs_shift;
				const int64_t n0 = (rr * D + e->h_frec[2 * y]) / k.rs_L;
				const int64_t rho = n0 / k.width;
				kind = 1;
				src = (int) ((int64_t) y * slab_in + rho * k.width + wp);
			}
			else
			{
				/* upwards: the raster's blanking -- it clears the whole buffer, max_width samples, before it writes its (shorter)
				 * line there (src/video.c:2934-2939 with :3645-3646), and nothing has written that place since */
				kind = 3;
			}
		}
		e->h_svrec[4 * i + 0] = (int) (S(j) - base);
		e->h_svrec[4 * i + 1] = w | ((delta & 15) << 16) | (kind << 20);
		e->h_svrec[4 * i + 2] = src;
		e->h_svrec[4 * i + 3] = 0;
	}
	HIPCHK(hipMemcpyAsync(e->d_svrec, e->h_svrec, (size_t) nlines * 16, hipMemcpyHostToDevice, e->stream));
	return(HVK_OK);
}

static int _sv_ring_q(hvk_engine *e)
{
	const hvk_kconst_t &k = e->t.k;
	int r = hvk_launch_svq(e->d_svrec, e->staged * k.lines, e->d_C2, e->d_C, e->d_Cq, k.s_lead, e->stream);
	if(r != HVK_OK) return(r);
	e->sv_tail_first = e->staged_first;
	e->sv_tail_total = e->staged_samples;
	e->sv_tail_frames = e->staged;
	return(HVK_OK);
}

/* ... before a NEW batch's sub-carrier stream is made: the end of the stream that lies there (the batch before's) goes in front
 * of it (a batch launched again finds what it found the first time) */
static int _sv_ring_keep(hvk_engine *e)
{
	const hvk_kconst_t &k = e->t.k;
	int16_t *const at = e->d_C2 + k.s_lead;         /* the batch's first sample; the sv_hist samples in front of it are the stream before */
	const int64_t H = e->sv_hist, T = e->sv_tail_total;
	if(e->sv_tail_first < 0 || (e->sv_tail_first == e->staged_first && e->sv_tail_frames == e->staged)) return(HVK_OK);       /* (the same batch launched again) */
	if(e->sv_tail_first + e->sv_tail_frames != e->staged_first)
	{
		/* (not the frames behind the batch before: what lay in the line buffers is not known -- nothing, as at the stream's start) */
		HIPCHK(hipMemsetAsync(at - H, 0, (size_t) H * 2, e->stream));
		return(HVK_OK);
	}
	if(T >= H)
	{
		HIPCHK(hipMemcpyAsync(at - H, at + T - H, (size_t) H * 2, hipMemcpyDeviceToDevice, e->stream));
		return(HVK_OK);
	}
	/* a batch shorter than what is kept: the older part moves up by the batch's length (in pieces no longer than the move: source
	 * and destination of a piece do not overlap), the batch goes behind it */
	for(int64_t off = 0; off < H - T; off += T)
	{
		const int64_t n = H - T - off < T ? H - T - off : T;
		HIPCHK(hipMemcpyAsync(at - H + off, at - H + T + off, (size_t) n * 2, hipMemcpyDeviceToDevice, e->stream));
	}
	HIPCHK(hipMemcpyAsync(at - T, at, (size_t) T * 2, hipMemcpyDeviceToDevice, e->stream));
	return(HVK_OK);
}

void hvk_e_kernel_args(hvk_engine *e, hvk_raster_args_t *pra, hvk_filter_args_t *pfa, void *d_iq, int64_t out_stride)
{
	hvk_raster_args_t &ra = *pra;
	memset(&ra, 0, sizeof(ra));
	ra.k = e->t.k;
	ra.ctaps = e->ctaps;
	ra.notch = e->notch;
	ra.chroma = e->t.k.rawbb ? e->d_raw : e->d_chroma;
	ra.vbi_sym = (const int *) e->d_vbi_sym;
	ra.vbi_val = (const int16_t *) e->d_vbi_val;
	ra.vbi_cov = (const int *) e->d_vbi_cov;
	ra.vbi_ops = e->d_ops;
	ra.vbi_map = (const signed char *) e->d_map;
	ra.fsc_rows = (const int16_t *) e->d_fsc_rows;
	ra.vits_l = (const int16_t *) e->d_vits_l;
	ra.vits_c = (const int16_t *) e->d_vits_c;
	ra.sis_dense = (const int16_t *) e->d_sis_dense;
	ra.sis_win = (const int16_t *) e->d_sis_win;
	ra.sis_first = (const int16_t *) e->d_sis_first;
	ra.sis_bits = e->d_sis_bits;
	ra.desc = (const hvk_linedesc_t *) e->d_desc;
	ra.pulses = (const int16_t *) e->d_pulses;
	ra.linebase = (const int16_t *) e->d_linebase;
	ra.yuv = e->d_yuv;
	ra.yuvparams = e->d_yuvparams;
	ra.levels_computed = e->levels_computed;
	ra.clut = (const hvk_c16_t *) e->d_clut;
	ra.burst_win = (const int16_t *) e->d_burst + HVK_PULSE_PAD;
	ra.ghost = (const int16_t *) e->d_ghost;
	ra.pool = e->d_pool;
	ra.fdesc = e->d_fdesc;
	ra.S = e->d_S;
	ra.C = e->d_C;
	ra.nframes = e->staged;
	ra.secam_fid = e->t.conf.secam_field_id != 0;
	ra.first_frame = e->staged_first;
	ra.frame_stride = e->staged_stride;

	hvk_filter_args_t &fa = *pfa;
	memset(&fa, 0, sizeof(fa));
	fa.k = e->t.k;
	fa.itaps = e->itaps;
	fa.qtaps = e->qtaps;
	fa.fdesc = e->d_fdesc;
	fa.S = e->t.k.rs_L ? e->d_S2 : e->d_S;
	fa.C = e->t.k.rs_L ? (e->t.k.sv_ring ? e->d_Cq : e->d_C2) : e->d_C;
	fa.carriers = (const hvk_c16_t *) e->d_car;
	fa.tilesyms = e->d_tile;
	fa.nicam_tapd = (const int *) e->d_tapd;
	fa.nicam_cca = (const int *) e->d_cca;
	fa.mfma_a = e->d_mfma_a;
	fa.mfma_ci = e->mfma_ci;
	fa.mfma_cq = e->mfma_cq;
	fa.iq = d_iq ? (int16_t *) d_iq : e->d_out;
	fa.nframes = e->staged;
	fa.out_stride = out_stride;

}

extern "C" int hvk_launch(hvk_engine_t *e, void *d_iq)
{
	return(hvk_launch_strided_out(e, d_iq, 1));
}

static int _launch_body(hvk_engine_t *e, void *d_iq, int64_t out_stride);

extern "C" int hvk_launch_strided_out(hvk_engine_t *e, void *d_iq, int64_t out_stride)
{
	int r = _launch_body(e, d_iq, out_stride);
	if(r != HVK_OK || !e->secam_pending) return(r);
	/* SECAM: the colour chain's check of the staged block is still out -- its count is read now that the render is queued behind
	 * it (hvk_engine_stage.cpp); a block whose check failed has been repaired by the time this returns 1 and is rendered again */
	r = hvk_e_secam_resolve(e);
	if(r < 0) return(r);
	return(r > 0 ? _launch_body(e, d_iq, out_stride) : HVK_OK);
}

static int _launch_body(hvk_engine_t *e, void *d_iq, int64_t out_stride)
{
	if(!e || out_stride < 1) return(HVK_ERROR);
	if(out_stride != 1 && d_iq == NULL) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(e->staged < 1) return(HVK_ERROR);
	/* FM video: the device buffer holds the modulator's input; the samples exist on the host only (hvk_fetch) */
	if(e->t.k.fm_video && d_iq != NULL) return(HVK_UNSUPPORTED);

	HIPCHK(hipSetDevice(e->device));

	hvk_raster_args_t ra;
	hvk_filter_args_t fa;
	hvk_e_kernel_args(e, &ra, &fa, d_iq, out_stride);

	const bool timed = e->timing && e->ev_used < HVK_TIMING_SLOTS;
	hipEvent_t *ev = timed ? e->ev[e->ev_used] : NULL;
	int r;

	if(timed) HIPCHK(hipEventRecord(ev[0], e->stream));
	if(e->direct)
	{
		/* one kernel: its time is reported as the second (filter) kernel's; the first one's is nil, or that of the raster
		 * kernel over the few lines the optional stages write to */
		if(e->ovr_n)
		{
			hvk_raster_args_t rl = ra;
			rl.linelist = e->d_ovr_list;
			rl.nlist = e->ovr_n;
			rl.S = e->d_Lp + 16 + (size_t) e->ovr_row0 * e->t.k.width;
			if((r = hvk_launch_raster(&rl, e->stream)) != HVK_OK) return(r);
		}
		if(timed) HIPCHK(hipEventRecord(ev[1], e->stream));
		hvk_direct_args_t da;
		memset(&da, 0, sizeof(da));
		da.k = e->t.k;
		da.D.Lp = e->d_Lp + 16;
		da.D.Cp = e->d_Cp ? e->d_Cp + 16 : NULL;
		da.D.clut3 = e->d_clut3 ? e->d_clut3 + 16 : NULL;
		da.D.creg = e->clut_reg;
		da.D.zero_row = e->plane_zero_row;
		da.D.desc = (const hvk_linedesc_t *) e->d_desc;
		da.D.lineoff = e->d_lineoff;
		da.D.inv_w = e->inv_w;
		da.D.ovr_idx = e->d_ovr_idx;
		da.D.ovr_n = e->ovr_n;
		da.tilerec = e->d_tilerec;
		da.tiles_pad = e->tiles_pad;
		da.nicam_tapd = fa.nicam_tapd;
		da.nicam_cca = fa.nicam_cca;
		da.mfma_a = fa.mfma_a;
		da.mfma_ci = fa.mfma_ci;
		da.mfma_cq = fa.mfma_cq;
		da.out_stride = out_stride;
		da.frame_stride = e->staged_stride;
		/* frames [y0, y0 + n) of the staged block */
		auto direct_range = [&](const int y0, const int n) -> int
		{
			const size_t FS = (size_t) e->t.k.frame_samples;
			da.D.fdesc = e->d_fdesc + 2 * (size_t) y0;
			/* (SECAM: a frame's sub-carrier lies in the row its descriptor names -- hvk_framedesc_t.chroma_row, counted from the store's start) */
			da.D.chroma = e->d_chroma;
			da.D.chroma_zero = (int) (((size_t) e->max_frames + 6 * (size_t) e->secam_memo_slots) * e->t.k.raster_samples + 16);
			da.D.ovr_row0 = e->ovr_row0 + y0 * e->ovr_n;
			da.carriers = fa.carriers ? fa.carriers + (size_t) y0 * FS : NULL;
			da.tilesyms = fa.tilesyms ? fa.tilesyms + (size_t) y0 * e->tiles * HVK_NICAM_ROW : NULL;
			da.iq = fa.iq + (size_t) y0 * (size_t) out_stride * FS * 2;
			da.nframes = n;
			da.first_frame = e->staged_first + (int64_t) y0 * e->staged_stride;
			return(hvk_launch_direct(&da, e->stream));
		};
		bool dirty = false, fused_now = false;
		int ndirty = 0;         /* new pictures among those the block shows (each counted once) */
		if(e->prep_pending)
		{
			std::vector<uint8_t> seen((size_t) e->frame_slots, 0);
			for(int i = 0; i < e->staged; i++)
			{
				const int sl[2] = { e->staged_slots[i], e->staged_prev[i] };
				for(int j = 0; j < 2; j++)
				{
					if(sl[j] < 0 || !e->slots[sl[j]].plane_dirty || seen[sl[j]]) continue;
					seen[sl[j]] = 1;
					dirty = true;
					if(!e->slots[sl[j]].shown) ndirty++;       /* (a picture that was shown before and is still here: its planes are made now) */
				}
			}
		}
		/* (levels by arithmetic -- pictures of many colours -- cost the one kernel more waves per SIMD than they are worth: 128 registers
		 * a lane against 76; such blocks go through the planes, whose hvk_k_prep8 holds the arithmetic alone: measured, profiles/README.md) */
		const bool fused_lv = e->levels_computed && e->t.yuv.fast == 2 && getenv("HVK_FUSED_LV") != NULL;
		if(dirty && e->fused_ok && e->fused_mode != 0 && (e->fused_mode == 1 || (2 * ndirty >= e->staged && (!e->levels_computed || fused_lv))))
		{
			ra.levels_computed = e->levels_computed ? (e->t.yuv.fast == 2 ? 3 : 1) : 0;
			/* most of the block's pictures are new: from the pixels in one kernel (hvk_fused.hip), their planes are not made
			 * (and stay marked: a later block that shows one of them again makes them then) */
			da.D.fdesc = e->d_fdesc;
			da.carriers = fa.carriers;
			da.tilesyms = fa.tilesyms;
			da.iq = fa.iq;
			da.nframes = e->staged;
			da.first_frame = e->staged_first;
			if((r = hvk_launch_fused(&ra, &da, e->d_mfma_a28, e->stream)) != HVK_OK) return(r);
			e->fused_count++;
			fused_now = true;
			for(int i = 0; i < e->staged; i++)
			{
				e->slots[e->staged_slots[i]].shown = 1;
				if(e->staged_prev[i] >= 0) e->slots[e->staged_prev[i]].shown = 1;
			}
		}
		else if(!dirty)
		{
			if((r = direct_range(0, e->staged)) != HVK_OK) return(r);
		}
		else
		{
			/* new pictures: their planes chunk by chunk on the second stream (behind everything queued so far: the pictures'
			 * uploads, the renders that still read the planes' old contents), each chunk's render behind its planes */
			const bool two = e->prep_streams == 2;
			hipStream_t ps = two ? e->prep_stream : e->stream;
			if(two)
			{
				HIPCHK_P(hipEventRecord(e->ev_fork, e->stream));
				HIPCHK_P(hipStreamWaitEvent(e->prep_stream, e->ev_fork, 0));
			}
			int ci = 0;
			for(int y0 = 0; y0 < e->staged; y0 += e->prep_chunk, ci++)
			{
				const int n = std::min(e->prep_chunk, e->staged - y0);
				const int np = hvk_e_prep_staged(e, y0, n, ps);
				if(np < 0) { e->poisoned = 1; return(np); }
				if(np > 0 && two)
				{
					hipEvent_t evp = e->ev_prep[ci % HVK_PREP_EVENTS];
					HIPCHK_P(hipEventRecord(evp, e->prep_stream));
					HIPCHK_P(hipStreamWaitEvent(e->stream, evp, 0));
				}
				if((r = direct_range(y0, n)) != HVK_OK) { e->poisoned = 1; return(r); }
			}
		}
		if(!fused_now)
		{
			e->prep_pending = 0;
			if((r = hvk_e_carry_copy(e)) != HVK_OK) return(r);
		}
	}
	else
	{
		if((r = hvk_launch_raster(&ra, e->stream)) != HVK_OK) return(r);
		if(e->t.k.rs_irr && out_stride != 1) return(HVK_UNSUPPORTED);
		if(e->t.k.rs_L && (r = hvk_launch_resample(&e->t.k, e->d_S, e->d_rs_taps, e->d_S2, e->staged, e->d_frec, e->stream)) != HVK_OK) return(r);
		if(e->t.k.sv_ring && (r = _sv_ring_keep(e)) != HVK_OK) return(r);
		if(e->t.k.rs_L && e->t.k.s_video && (r = hvk_launch_resample(&e->t.k, e->d_C, e->d_rs_taps, e->d_C2, e->staged, e->d_frec, e->stream)) != HVK_OK) return(r);
		if(e->t.k.sv_ring && (r = _sv_ring_q(e)) != HVK_OK) return(r);
		if(timed) HIPCHK(hipEventRecord(ev[1], e->stream));
		if(e->t.k.rs_irr)
		{
			/* frames of two lengths: the resampled frames lie one behind the other, and everything from here on -- the
			 * filter never knew about lines, nor does it need to know about frames -- takes the batch as ONE frame of
			 * staged_samples samples (64 of halo either side, as every frame has them otherwise) */
			fa.k.frame_samples = (int32_t) e->staged_samples;
			fa.k.s_stride = (int32_t) ((e->staged_samples + 2 * 64 + 7) & ~7);
			fa.nframes = 1;
		}
		if((r = hvk_launch_filter(&fa, e->stream)) != HVK_OK) return(r);
	}
	if(timed) { HIPCHK(hipEventRecord(ev[2], e->stream)); e->ev_used++; }
	e->last_direct = e->direct;
	if(!e->t.k.fm_video && (e->t.k.swap_iq || e->d_off || e->d_pass))
	{
		if(e->t.k.rs_irr) r = hvk_launch_tail(fa.iq, e->d_off, e->d_pass, e->t.k.swap_iq, (int) e->staged_samples, 1, 1, e->stream);
		else r = hvk_launch_tail(fa.iq, e->d_off, e->d_pass, e->t.k.swap_iq, e->t.k.frame_samples, out_stride, e->staged, e->stream);
		if(r != HVK_OK) return(r);
	}

	e->last_frames = e->staged;
	e->last_samples = e->staged_samples;
	e->fm_launched = e->t.k.fm_video;

	if(e->fm_prime_pending)
	{
		/* The modulator's input over the start-up samples: the video filter's output while its history is still
		 * zero -- nothing but its last ntaps / 2 outputs, whose windows reach the stream's first samples -- plus the
		 * sound carriers. The stream's first raster samples come from the slab just rendered. */
		const hvk_kconst_t &k = e->t.k;
		const int nt = k.vf_type ? k.vf_ntaps : 0, H = nt / 2, P = k.out_prime;
		std::vector<int16_t> x(H), in((size_t) P);
		if(k.rs_L)
		{
			/* Behind the resampler the start-up samples are not nothing: the resampler's output for raster line N lands in
			 * the slot of line N - 1, so the resampled raster line 1 (and, with the filter on, the filter's output over it
			 * and the line after) passes the modulator before the first emitted sample does (hvk_tables.c: out_prime,
			 * rs_shift). Resampled sample r is made of raster sample floor(r D / L) and the ataps - 1 before it with the
			 * taps of phase (r D) mod L, nothing in front of the stream's first raster sample (hvk_k_resample says the same
			 * of the samples it makes); the stream's sample 0 is the filter's output centred on resampled sample rs_shift. */
			const int64_t L = k.rs_L, D = k.rs_D;
			const int A = k.rs_ataps;
			const int Rn = k.rs_shift + H + 1;
			const int nr = (int) (((int64_t) (Rn - 1) * D) / L) + 1;
			if(nr > k.raster_samples) { e->poisoned = 1; return(HVK_ERROR); }
			std::vector<int16_t> xr((size_t) nr), xs((size_t) Rn);
			HIPCHK(hipMemcpyAsync(xr.data(), e->d_S + (size_t) k.width, (size_t) nr * 2, hipMemcpyDeviceToHost, e->stream));
			HIPCHK(hipStreamSynchronize(e->stream));
			for(int rr = 0; rr < Rn; rr++)
			{
				const int64_t n = ((int64_t) rr * D) / L, ph = ((int64_t) rr * D) % L;
				int32_t acc = 0;
				for(int y = 0; y < A; y++)
				{
					const int64_t xi = n - A + 1 + y;
					if(xi >= 0) acc += (int32_t) xr[(size_t) xi] * e->t.rs_taps[(size_t) ph * A + y];
				}
				acc >>= 15;
				xs[(size_t) rr] = (int16_t) (acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc));
			}
			for(int n = 0; n < P; n++)
			{
				const int c = n - P + k.rs_shift;       /* the resampled sample this output is centred on */
				int32_t acc;
				if(nt)
				{
					acc = 0;
					for(int kk = 0; kk < nt; kk++)
					{
						const int xi = c - H + kk;
						if(xi >= 0 && xi < Rn) acc += (int32_t) e->t.vf_itaps[kk] * xs[(size_t) xi];
					}
					acc >>= 15;
					acc = acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc);
				}
				else acc = c >= 0 && c < Rn ? xs[(size_t) c] : 0;
				in[n] = (int16_t) (acc + e->fm_prime_car[(size_t) n * 2]);
			}
		}
		else
		{
		HIPCHK(hipMemcpyAsync(x.data(), e->d_S + (size_t) k.width, (size_t) H * 2, hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipStreamSynchronize(e->stream));
		for(int n = 0; n < P; n++)
		{
			int32_t acc = 0;
			const int m = n - P;                    /* stream position of this output: -P .. -1 */
			for(int kk = 0; kk < nt; kk++)
			{
				const int xi = m - H + kk;
				if(xi >= 0 && xi < H) acc += (int32_t) e->t.vf_itaps[kk] * x[xi];
			}
			acc >>= 15;
			acc = acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc);
			in[n] = (int16_t) (acc + e->fm_prime_car[(size_t) n * 2]);     /* int16 wrap-around add, src/video.c:3431 */
		}
		}
		r = hvk_tail_fm_prime(e->tail, in.data(), P);
		if(r != HVK_OK) { e->poisoned = 1; return(r); }      /* the sound chain is past these samples: the stream cannot go on */
		e->fm_prime_pending = 0;
	}
	return(HVK_OK);
}

extern "C" int hvk_render_strided(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes,
                                  const int32_t *slots, void *d_iq)
{
	int r = hvk_stage_strided(e, first_frame, stride, nframes, slots);
	if(r != HVK_OK) return(r);
	return(hvk_launch(e, d_iq));
}

extern "C" int hvk_render(hvk_engine_t *e, int nframes, const int32_t *slots, void *d_iq)
{
	if(!e) return(HVK_ERROR);
	int r = hvk_render_strided(e, e->next_frame, 1, nframes, slots, d_iq);
	if(r == HVK_OK) e->next_frame += nframes;
	return(r);
}

extern "C" int hvk_sync(hvk_engine_t *e)
{
	if(!e) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipStreamSynchronize(e->stream));
	return(HVK_OK);
}
