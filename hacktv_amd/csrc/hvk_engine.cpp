/* hvk_engine.cpp -- the C ABI of libhvk (include/hacktv_amd.h): engine
 * life cycle, HBM residency of tables / frames / side streams, and the
 * per-batch launch sequence on one HIP stream.
 *
 * What vid_init() / vid_next_line() / vid_free() are to the reference
 * (src/video.c:3812, :4936, :4706), hvk_open() / hvk_render() / hvk_close()
 * are here, at frame instead of line granularity. There is no CPU rendering
 * path in this library: without a HIP device every render call fails.
 *
 * HBM layout (all allocated once in hvk_open, sized by max_frames):
 *   yuv        2^24 x int16x4   128 MiB   RGB -> level table, expanded on device
 *   clut       (clw + width) x int16x2    colour sub-carrier phasors
 *   pool       frame_slots x active_w x active_h x 4 B   source frames (RGBx)
 *   S          max_frames x (lines + 2 [+ 1]) x width x 2 B  raster stream (int16 I), halo lines either side
 *   carriers   max_frames x frame_samples x 4 B          serial-carrier side stream
 *   symtab     max_frames x symbol_stride x 4 B          NICAM symbols: start sample and value
 *   tileinfo   max_frames x tiles x 8 B                  per filter tile: newest symbol, mixer position
 *   out        max_frames x frame_samples x 4 B          int16 I/Q (if the caller gives no buffer)
 * and, only with the option that needs them: S2 (resampled stream), C (S-Video sub-carrier),
 * off / pass (offset phasor, passthru samples), chroma (SECAM), vbi_sym / vbi_val / ops / map
 * (VBI data lines), vits tables. DESIGN.md section 2 has the sizes.
 */
#include "hvk_engine_priv.h"

/* The video filter as a matrix product (hvk_k_filter, MF = 1). Eight consecutive outputs of a
 * segment and both channels make the 16 rows of A, 64 window positions its columns:
 *
 *   row m = 4 g + i  ->  output b = 2 g + (i >> 1) of the segment, channel i & 1 (I, Q)
 *   A[m][t] = h[t - 1 - b]      (window position 0 is the sample 26 before the segment's first)
 *
 * int16 x int16 on the int8 matrix unit: the taps are split into signed bytes h = 256 hh + hl
 * (hl = the low byte read as signed), the samples into x = 256 xh + (xl - 128) + 128, which gives
 * four int8 products and the constant 128 * sum(h) -- exact modulo 2^32, like the reference's
 * int32 accumulator. Layout: [hh, hl][lane][16 bytes], lane = 16 * (t / 16) + m, byte = t % 16.
 * A tap above 32639 has no such split (hh = 128): the caller keeps the VALU kernel then. */
static bool _mfma_taps(int8_t *a, int *ci, int *cq, const int16_t *hi, const int16_t *hq, int ntaps, int lead = 26)
{
	int64_t si = 0, sq = 0;

	for(int k = 0; k < ntaps; k++)
	{
		si += hi[k];
		if(hq) sq += hq[k];
	}

	for(int lane = 0; lane < 64; lane++)
	{
		const int g = lane >> 4, m = lane & 15, b = 2 * (m >> 2) + ((m & 3) >> 1), q = m & 1;

		for(int j = 0; j < 16; j++)
		{
			const int k = 16 * g + j - (lead - 25) - b;      /* window position 0 lies `lead` samples before the segment's first output */
			const int h = (k >= 0 && k < ntaps) ? (q ? (hq ? hq[k] : 0) : hi[k]) : 0;
			const int lo = (int) (int8_t) (h & 0xFF), hh = (h - lo) >> 8;
			if(hh < -128 || hh > 127) return(false);
			a[(0 * 64 + lane) * 16 + j] = (int8_t) hh;
			a[(1 * 64 + lane) * 16 + j] = (int8_t) lo;
		}
	}

	*ci = (int) (uint32_t) (128 * si);
	*cq = (int) (uint32_t) (128 * sq);
	return(true);
}

static void _pack_taps(hvk_packed_taps_t *p, const int16_t *taps, int ntaps)
{
	memset(p, 0, sizeof(*p));
	for(int k = 0; k < ntaps && k < HVK_MAX_VF_TAPS; k++)
	{
		p->p[k / 2] |= (k & 1) ? ((int) taps[k] << 16) : ((int) taps[k] & 0xFFFF);
	}
}

static int _upload(void **dst, const void *src, size_t bytes)
{
	*dst = NULL;
	if(bytes == 0 || src == NULL) return(HVK_OK);
	HIPCHK(hipMalloc(dst, bytes));
	HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
	return(HVK_OK);
}

extern "C" const char *hvk_version(void) { return(HVK_VERSION); }

extern "C" int hvk_open(hvk_engine_t **pe, const hvk_config_t *conf, unsigned int sample_rate, int device, int max_frames)
{
	return(hvk_open_rates(pe, conf, sample_rate, 0, device, max_frames));
}

extern "C" int hvk_open_rates(hvk_engine_t **pe, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate,
                              int device, int max_frames)
{
	hvk_engine *e;
	int r, ndev = 0;

	if(!pe || !conf) return(HVK_ERROR);
	*pe = NULL;
	if(conf->struct_size != sizeof(hvk_config_t))
	{
		fprintf(stderr, "libhvk: hvk_config_t.struct_size is %u, this library's is %zu: the caller was built against another include/hvk_config.h "
		                "(or did not set it: hvk_config_preset() and HVK_CONFIG_INIT do)\n", conf->struct_size, sizeof(hvk_config_t));
		return(HVK_ERROR);
	}
	if(max_frames < 1) max_frames = 1;

	e = (hvk_engine *) calloc(1, sizeof(hvk_engine));
	if(!e) return(HVK_OUT_OF_MEMORY);
	e->secam_last_frame = -1;
	e->device = device;
	e->max_frames = max_frames;
	/* one source frame slot per frame of a batch, so a batch can show a different
	 * picture on every frame (a static source needs only slot 0) */
	{
		/* --interlace shows a different source frame on each field */
		const int want = max_frames * ((conf->interlace && conf->interlaced) ? 2 : 1);
		e->frame_slots = want < HVK_MIN_FRAME_SLOTS ? HVK_MIN_FRAME_SLOTS : (want > HVK_MAX_FRAME_SLOTS ? HVK_MAX_FRAME_SLOTS : want);
	}
	e->slots = (hvk_slot_t *) calloc(e->frame_slots, sizeof(hvk_slot_t));
	e->staged_slots = (int32_t *) calloc(max_frames, sizeof(int32_t));
	e->staged_slots2 = (int32_t *) calloc(max_frames, sizeof(int32_t));
	if(!e->slots || !e->staged_slots || !e->staged_slots2) { free(e->slots); free(e->staged_slots); free(e->staged_slots2); free(e); return(HVK_OUT_OF_MEMORY); }

	if((r = hvk_tables_build(&e->t, conf, sample_rate, pixel_rate)) != HVK_OK) { hvk_close(e); return(r); }

	/* tools/ablate.py: only a library built with ABLATE=1 has the switches in its kernels; results are WRONG when set */
	if(getenv("HVK_ABLATE")) e->t.k.ablate = atoi(getenv("HVK_ABLATE"));
	if(getenv("HVK_LEVELS"))
	{
		const char *v = getenv("HVK_LEVELS");
		e->levels_mode = !strcmp(v, "table") ? HVK_LEVELS_TABLE : (!strcmp(v, "compute") ? HVK_LEVELS_COMPUTE : HVK_LEVELS_AUTO);
	}

	/* the kernels exist for these chroma filter lengths (5 .. 33 taps: pixel rates of about 5 to 45 MHz) and for
	 * the NICAM pulse lengths the LDS table holds: say so now, not at the first render */
	if(e->t.k.colour)
	{
		const int nt = e->t.k.chroma_ntaps;
		if((nt < 5 || nt > 33 || !(nt & 1)) && !(nt == 3 && e->t.chroma_unfiltered))
		{
			fprintf(stderr, "libhvk: no raster kernel for a %d-tap chroma filter (pixel rate %d Hz)\n", nt, e->t.pixel_rate);
			hvk_close(e);
			return(HVK_UNSUPPORTED);
		}
	}
	if(e->t.k.has_nicam && e->t.sample_rate < 10000000)
	{
		/* Below 10 MHz -- where the 6.552 MHz carrier has long left the band -- the symbols get short against a lane's 8
		 * samples and a tile's 1024: a tile's row of 48 symbol slots and the 7 symbols a lane looks at stop being enough
		 * (9.5 MHz already differs from the oracle in the GPU sweep). Refused, not approximated. */
		fprintf(stderr, "libhvk: NICAM symbols of %d samples at %d Hz are too short for the kernel's tables\n", e->t.k.nicam_sps, e->t.sample_rate);
		hvk_close(e);
		return(HVK_UNSUPPORTED);
	}
	if(e->t.k.has_nicam && HVK_NICAM_LEAD + e->t.k.nicam_ntaps + HVK_SPL > HVK_NICAM_TAPD)
	{
		fprintf(stderr, "libhvk: the NICAM pulse (%d taps at %d Hz) does not fit the kernel's table\n", e->t.k.nicam_ntaps, e->t.sample_rate);
		hvk_close(e);
		return(HVK_UNSUPPORTED);
	}
	if(e->t.k.colour) _pack_taps(&e->ctaps, e->t.chroma_taps, e->t.k.chroma_ntaps);
	if(e->t.k.vf_type) _pack_taps(&e->itaps, e->t.vf_itaps, e->t.k.vf_ntaps);
	if(e->t.k.vf_type == 3) _pack_taps(&e->qtaps, e->t.vf_qtaps, e->t.k.vf_ntaps);

	if(e->t.k.has_carriers || e->t.k.has_nicam || e->t.k.sis)
	{
		e->audio = hvk_audio_new(&e->t);
		if(!e->audio) { hvk_close(e); return(HVK_OUT_OF_MEMORY); }
	}

	if(e->t.k.secam)
	{
		e->secam = hvk_secam_new(&e->t);
		e->host_frames = (uint32_t **) calloc(e->frame_slots, sizeof(uint32_t *));
		if(!e->secam || !e->host_frames) { hvk_close(e); return(HVK_OUT_OF_MEMORY); }
		_pack_taps(&e->notch, e->t.secam_notch, 51);
	}

	if(e->t.k.fm_video || e->t.k.has_offset || e->t.k.has_passthru)
	{
		e->tail = hvk_tail_new(&e->t);
		if(!e->tail) { hvk_close(e); return(HVK_OUT_OF_MEMORY); }
	}

	if(e->t.k.rawbb) e->raw_q = new std::vector<int16_t>();

	if(e->t.conf.cc608)
	{
		e->cc_pairs = (uint8_t *) calloc((size_t) max_frames, 3);
		if(!e->cc_pairs) { hvk_close(e); return(HVK_OUT_OF_MEMORY); }
	}

	if(e->t.k.has_nicam)
	{
		e->symbol_stride = e->t.k.frame_samples / (e->t.k.nicam_sps - 1) + 32;
		e->symbol_stride = (e->symbol_stride + 15) & ~15;
	}

	*pe = e;
	if(device < 0) return(HVK_OK);   /* host tables only: parity tests without a GPU */

	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev)
	{
		fprintf(stderr, "libhvk: no HIP device %d (found %d); this engine has no CPU path\n", device, ndev);
		*pe = NULL;
		e->device = -1;
		hvk_close(e);
		return(HVK_NO_DEVICE);
	}

	const hvk_kconst_t &k = e->t.k;
	const size_t FS = k.frame_samples;

#define OPENCHK(x) do { int _r = (x); if(_r != HVK_OK) { *pe = NULL; hvk_close(e); return(_r); } } while(0)
#define OPENHIP(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s\n", #call, hipGetErrorString(_e)); *pe = NULL; hvk_close(e); \
	return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)

	OPENHIP(hipSetDevice(device));
	OPENHIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
	e->stream = e->own_stream;

	OPENCHK(_upload(&e->d_yuvparams, &e->t.yuv, sizeof(e->t.yuv)));
	if(e->t.k.vf_type && e->t.k.vf_ntaps == 51 && !getenv("HVK_NO_MFMA"))
	{
		std::vector<int8_t> a(HVK_MFMA_A_BYTES);
		if(_mfma_taps(a.data(), &e->mfma_ci, &e->mfma_cq, e->t.vf_itaps, e->t.k.vf_type == 3 ? e->t.vf_qtaps : NULL, 51))
		{
			OPENCHK(_upload(&e->d_mfma_a, a.data(), a.size()));
			int ci2, cq2;
			if(_mfma_taps(a.data(), &ci2, &cq2, e->t.vf_itaps, e->t.k.vf_type == 3 ? e->t.vf_qtaps : NULL, 51, 28)) OPENCHK(_upload(&e->d_mfma_a28, a.data(), a.size()));
		}
	}
	OPENHIP(hipMalloc(&e->d_yuv, 0x1000000UL * 8));
	OPENCHK(hvk_launch_expand_yuv(e->d_yuv, e->d_yuvparams, e->stream));
	if(!getenv("HVK_EXACT_LEVELS"))
	{
		/* Levels computed per pixel (pictures with many colours): the short form of the arithmetic, IF it gives the table's
		 * levels for every one of the 2^24 colours of this mode -- tried here, once; otherwise the reference's sequence of
		 * operations stays (hvk_yuvparams_t.fast) */
		void *d_n = NULL;
		OPENHIP(hipMalloc(&d_n, sizeof(int)));
		for(int fast = 2; fast >= 1 && !e->t.yuv.fast; fast--)
		{
			int differ = -1;
			OPENHIP(hipMemsetAsync(d_n, 0, sizeof(int), e->stream));
			OPENCHK(hvk_launch_check_levels(e->d_yuv, e->d_yuvparams, fast, (int *) d_n, e->stream));
			OPENHIP(hipMemcpyAsync(&differ, d_n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
			OPENHIP(hipStreamSynchronize(e->stream));
			if(differ == 0) e->t.yuv.fast = fast;
			else if(getenv("HVK_SHIM_STATS")) fprintf(stderr, "libhvk: level arithmetic, short form %d, differs on %d colours of this mode\n", fast, differ);
		}
		(void) hipFree(d_n);
	}

	/* One kernel from picture planes (hvk_direct.hip) for the plain configurations; HVK_DIRECT=0 keeps the raster +
	 * filter kernel pair (the parity tests run both). The planes do not depend on a frame's parity: the two
	 * descriptor sets may differ in nothing but `pal`; and the line after a frame must not show picture (its
	 * planes are taken from the frame's own picture). */
	e->direct = hvk_direct_supported(&e->t.k, e->d_mfma_a, e->t.conf.secam_field_id != 0, max_frames) && !(getenv("HVK_DIRECT") && atoi(getenv("HVK_DIRECT")) == 0);
	for(int l = 0; l < k.lines && e->direct; l++)
	{
		hvk_linedesc_t a = e->t.desc[l], b = e->t.desc[k.lines + l];
		a.pal = b.pal = 0;
		if(memcmp(&a, &b, sizeof(a)) != 0) e->direct = 0;
	}
	if(e->t.desc[0].ar > e->t.desc[0].al || e->t.desc[k.lines].ar > e->t.desc[k.lines].al) e->direct = 0;

	OPENCHK(_upload(&e->d_desc, e->t.desc, sizeof(hvk_linedesc_t) * 2 * k.lines));
	OPENCHK(_upload(&e->d_pulses, e->t.pulse_values, sizeof(int16_t) * (e->t.pulse_total + 8)));
	OPENCHK(_upload(&e->d_linebase, e->t.linebase, sizeof(int16_t) * (size_t) e->t.nbase * k.base_stride));
	if(e->t.colour_lookup_len > 0)
	{
		/* 8 entries of slack behind the table: a lane that straddles the end of a line loads 8 entries all the same */
		std::vector<hvk_c16_t> cl((size_t) e->t.colour_lookup_len + 8);
		memcpy(cl.data(), e->t.colour_lookup, sizeof(hvk_c16_t) * e->t.colour_lookup_len);
		memset(cl.data() + e->t.colour_lookup_len, 0, sizeof(hvk_c16_t) * 8);
		OPENCHK(_upload(&e->d_clut, cl.data(), sizeof(hvk_c16_t) * cl.size()));
	}
	{
		/* HVK_PULSE_PAD zeros either side: a lane reads its 8 window values in one load wherever it stands */
		std::vector<int16_t> bw((size_t) k.burst_width + 2 * HVK_PULSE_PAD, 0);
		for(int i = 0; i < k.burst_width; i++) bw[HVK_PULSE_PAD + i] = e->t.burst_win[i];
		OPENCHK(_upload(&e->d_burst, bw.data(), bw.size() * sizeof(int16_t)));
	}
	{
		/* the samples the reference reads past its chroma buffer, and behind them (hvk_k_prep8) the masks of runs of a lane's 8
		 * samples: entry 9 lo + hi has the 16-bit elements lo .. hi - 1 set */
		std::vector<uint8_t> gb(HVK_RUNMASK_OFFSET + 81 * 16, 0);
		static_assert(sizeof(e->t.ghost) <= HVK_RUNMASK_OFFSET, "over-read samples in front of the run masks");
		memcpy(gb.data(), e->t.ghost, sizeof(e->t.ghost));
		for(int lo = 0; lo <= 8; lo++) for(int hi = lo; hi <= 8; hi++)
		{
			uint16_t *m = (uint16_t *) (gb.data() + HVK_RUNMASK_OFFSET + (lo * 9 + hi) * 16);
			for(int i = lo; i < hi; i++) m[i] = 0xFFFF;
		}
		OPENCHK(_upload(&e->d_ghost, gb.data(), gb.size()));
	}
	if(e->direct)
	{
		e->plane_rows = e->frame_slots * k.lines + 3;
		e->plane_carry_row = e->frame_slots * k.lines;
		e->plane_zero_row = e->plane_carry_row + 2;
		if(k.vbi || k.vits || (k.secam && e->t.conf.secam_field_id))
		{
			std::vector<uint8_t> held((size_t) k.lines, 0);
			std::vector<int16_t> list, idx((size_t) k.lines, -1);
			hvk_vbi_lines_held(e, held.data(), k.lines);
			/* teletext's lines: rows 0 .. 15 on lines 7 .. 22, rows 16 .. 31 on lines 320 .. 335 (src/teletext.c:1211-1236) */
			if(k.teletext) for(int r = 0; r < 32; r++) { const int l1 = r < 16 ? 7 + r : 320 + r - 16; if(l1 <= k.lines) held[l1 - 1] = 1; }
			for(int l = 0; l < k.lines; l++) if(held[l]) { idx[l] = (int16_t) list.size(); list.push_back((int16_t) l); }
			/* (a frame's first lines and its last one are also what the frames next to it look into -- from THEIR planes, which
			 * have no such rows: no inserter of the reference writes there, and if one did the kernel pair would render) */
			if(held[0] || held[1] || held[k.lines - 1])
			{
				e->direct = 0;
				fprintf(stderr, "libhvk: an optional stage writes to the frame's first lines or its last: the raster + filter kernel pair renders\n");
			}
			if(!list.empty() && e->direct)
			{
				e->ovr_n = (int) list.size();
				e->ovr_row0 = e->plane_rows;
				OPENCHK(_upload((void **) &e->d_ovr_list, list.data(), list.size() * sizeof(int16_t)));
				OPENCHK(_upload((void **) &e->d_ovr_idx, idx.data(), idx.size() * sizeof(int16_t)));
				e->h_ovr_idx = (int16_t *) malloc(idx.size() * sizeof(int16_t));
				if(!e->h_ovr_idx) { *pe = NULL; hvk_close(e); return(HVK_OUT_OF_MEMORY); }
				memcpy(e->h_ovr_idx, idx.data(), idx.size() * sizeof(int16_t));
			}
		}
		if(e->direct)
		{
			/* a window position's line by a multiplication instead of a division: exact up to the last position a tile can ask
			 * for? (the quotient can only go wrong next to a multiple of the width: those and their neighbours are tried) */
			e->inv_w = (uint32_t) (((1ULL << 32) + k.width - 1) / k.width);
			const uint32_t qmax = (uint32_t) ((k.frame_samples + 8 * HVK_TILE) / k.width + 1);
			for(uint32_t q = 0; q <= qmax && e->direct; q++)
			{
				const uint32_t n0 = q * (uint32_t) k.width, n1 = n0 + (uint32_t) k.width - 1;
				if((uint32_t) (((uint64_t) n0 * e->inv_w) >> 32) != q || (uint32_t) (((uint64_t) n1 * e->inv_w) >> 32) != q) e->direct = 0;
			}
			if(!e->direct) fprintf(stderr, "libhvk: no exact reciprocal of the line width %d: the raster + filter kernel pair renders\n", k.width);
		}
	}
	if(e->direct)
	{
		const size_t pn = ((size_t) e->plane_rows + (size_t) e->ovr_n * max_frames) * k.width + 32;
		OPENHIP(hipMalloc((void **) &e->d_Lp, pn * 2));
		OPENHIP(hipMemset(e->d_Lp, 0, pn * 2));
		if(k.secam && !getenv("HVK_SECAM_NO_UV_PLANE"))
		{
			/* (SECAM: no (V, U) plane for the render and no phasors -- the sub-carrier is the colour chain's -- but the pixels'
			 * colour-difference levels for the chain's cells, if both parities' lines show the same rows) */
			int same = 1;
			for(int l = 0; l < k.lines && same; l++)
			{
				const hvk_linedesc_t *d0 = &e->t.desc[l], *d1 = &e->t.desc[k.lines + l];
				same = d0->src_row == d1->src_row && d0->al == d1->al && d0->ar == d1->ar;
			}
			if(same)
			{
				OPENHIP(hipMalloc((void **) &e->d_UVp, pn * 4));
				OPENHIP(hipMemset(e->d_UVp, 0, pn * 4));
			}
		}
		if(k.colour && !k.secam)
		{
			OPENHIP(hipMalloc((void **) &e->d_Cp, pn * 4));
			OPENHIP(hipMemset(e->d_Cp, 0, pn * 4));
			/* the phasors as they are, with i negated (the PAL switch; i never is -32768), and zeros (a line without
			 * chroma): regions of clw + width + 16 entries, 16 of slack in front */
			e->clut_reg = (int) e->t.colour_lookup_len + 16;
			std::vector<int> c3((size_t) 3 * e->clut_reg + 32, 0);
			for(int64_t i = 0; i < e->t.colour_lookup_len; i++)
			{
				const hvk_c16_t c = e->t.colour_lookup[i];
				c3[16 + (size_t) i] = ((int) c.i & 0xFFFF) | ((int) c.q << 16);
				c3[16 + (size_t) e->clut_reg + i] = ((-(int) c.i) & 0xFFFF) | ((int) c.q << 16);
			}
			OPENCHK(_upload((void **) &e->d_clut3, c3.data(), c3.size() * 4));
		}
		{
			/* what the kernel would divide for: a window position's line by a multiplication (exact up to the
			 * last position a tile can ask for -- checked here), a line's colour table position from a table */
			std::vector<uint32_t> lo((size_t) k.lines + 4, 0);
			for(int j = 0; j < k.lines + 4 && k.colour && !k.secam && k.clw > 0; j++) lo[j] = (uint32_t) ((((int64_t) (j - 1) * k.width) % k.clw + k.clw) % k.clw);
			OPENCHK(_upload((void **) &e->d_lineoff, lo.data(), lo.size() * 4));
			if(e->direct && !(getenv("HVK_TILEREC") && atoi(getenv("HVK_TILEREC")) == 0))
			{
				/* per frame parity and tile of 1024 outputs (the tiles of the last, partial workgroup included): the lines its window
				 * lies in -- the arithmetic of hvk_direct.hip:direct_line_loads() / direct_line(), done once here */
				const int DGT = 4, lead = k.vf_type ? 26 : 0, W = k.width, tiles = (k.frame_samples + HVK_TILE - 1) / HVK_TILE;
				e->tiles_pad = (tiles + DGT - 1) / DGT * DGT;
				std::vector<hvk_tilerec_t> rec((size_t) 2 * e->tiles_pad);
				std::vector<int16_t> ovr_of_line((size_t) k.lines, -1);
				if(e->ovr_n) for(int l = 0; l < k.lines; l++) ovr_of_line[(size_t) l] = e->h_ovr_idx[(size_t) l];
				memset(rec.data(), 0, rec.size() * sizeof(hvk_tilerec_t));
				for(int par_own = 0; par_own < 2; par_own++) for(int tl = 0; tl < e->tiles_pad; tl++)
				{
					hvk_tilerec_t &R = rec[(size_t) par_own * e->tiles_pad + tl];
					const int p0 = tl * HVK_TILE - lead;
					const int lineA = p0 < 0 ? -1 : p0 / W, xA0 = p0 - lineA * W;
					R.b1 = W - xA0;
					for(int X = 0; X < 3; X++)
					{
						const int rel = lineA + X, wstart = X == 0 ? -xA0 : (X == 1 ? R.b1 : R.b1 + W);
						int line0 = rel, par = par_own, prev = 0;
						const int own = rel >= 0 && rel < k.lines;
						if(rel < 0) { line0 = k.lines - 1; par ^= 1; prev = 1; }
						else if(rel >= k.lines) { line0 = rel - k.lines < k.lines ? rel - k.lines : k.lines - 1; par ^= 1; }
						const int pal = (k.colour && !k.secam) ? e->t.desc[(size_t) par * k.lines + line0].pal : 0;
						R.meta[X] = line0 | (prev << 16) | (own << 17) | ((pal + 1) << 18);
						R.lw[X] = line0 * W - wstart;
						R.nws[X] = -wstart;
						R.off[X] = lo[rel + 1 < k.lines + 3 ? rel + 1 : k.lines + 3];
						R.ovr[X] = (own && e->ovr_n) ? (int) ovr_of_line[(size_t) line0] : -1;       /* (a line the optional stages can write to: the frame's own row of it) */
					}
				}
				OPENCHK(_upload(&e->d_tilerec, rec.data(), rec.size() * sizeof(hvk_tilerec_t)));
			}
		}
		OPENHIP(hipStreamCreateWithFlags(&e->prep_stream, hipStreamNonBlocking));
		OPENHIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
		for(int i = 0; i < HVK_PREP_EVENTS; i++) OPENHIP(hipEventCreateWithFlags(&e->ev_prep[i], hipEventDisableTiming));
		e->prep_chunk = getenv("HVK_PREP_CHUNK") ? atoi(getenv("HVK_PREP_CHUNK")) : 0;
		e->prep_streams = getenv("HVK_PREP_STREAMS") ? atoi(getenv("HVK_PREP_STREAMS")) : 1;
		e->fused_mode = getenv("HVK_FUSED") ? atoi(getenv("HVK_FUSED")) : -1;
		e->fused_ok = e->direct && e->d_mfma_a28 && e->d_clut3 && !e->ovr_n && hvk_fused_supported(&e->t.k, e->t.desc);
		if(e->prep_chunk < 1) e->prep_chunk = max_frames;
		e->staged_prev = (int32_t *) malloc(sizeof(int32_t) * (size_t) max_frames);
		if(!e->staged_prev) { *pe = NULL; hvk_close(e); return(HVK_OUT_OF_MEMORY); }
	}
	if(k.has_nicam)
	{
		/* device forms of the NICAM tables: the pulse as int16 behind HVK_NICAM_LEAD
		 * zeros and zero padded (no bounds test is needed), in HVK_NICAM_COPIES copies of which
		 * copy s starts s entries later, so that any eight consecutive entries start
		 * 8-byte aligned in one of them (hvk_kernels.h has why four and not eight); the mixer as the first row of the rotation
		 * matrix, (i, -q), extended by 8 entries past the wrap, and its second row, (q, i), likewise */
		std::vector<int16_t> tapd(HVK_NICAM_COPIES * HVK_NICAM_TAPD, 0);
		std::vector<int> cca(2 * (size_t) (k.nicam_cc_len + 8));
		if(HVK_NICAM_LEAD + k.nicam_ntaps + HVK_SPL > HVK_NICAM_TAPD) { *pe = NULL; hvk_close(e); return(HVK_UNSUPPORTED); }
		for(int i = 0; i < k.nicam_ntaps; i++)
		{
			const int v = e->t.nicam_taps[i];
			/* copy s holds entry j + s at position j */
			for(int sft = 0; sft < HVK_NICAM_COPIES; sft++) tapd[sft * HVK_NICAM_TAPD + HVK_NICAM_LEAD + i - sft] = (int16_t) v;
		}
		for(int i = 0; i < k.nicam_cc_len + 8; i++)
		{
			const hvk_c16_t c = e->t.nicam_cc[i % k.nicam_cc_len];
			cca[i] = ((int) c.i & 0xFFFF) | ((-(int) c.q) << 16);
			cca[(size_t) (k.nicam_cc_len + 8) + i] = ((int) c.q & 0xFFFF) | ((int) c.i << 16);    /* the rotation's second row (|q| <= 32767) */
		}
		OPENCHK(_upload(&e->d_tapd, tapd.data(), tapd.size() * sizeof(int16_t)));
		OPENCHK(_upload(&e->d_cca, cca.data(), cca.size() * 4));
	}

	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	OPENHIP(hipMalloc((void **) &e->d_pool_alloc, frame_px * 4 * e->frame_slots + (size_t) k.active_width * 4 * 2 + 2 * HVK_POOL_PAD * 4));   /* + two kept rows */
	OPENHIP(hipMemset(e->d_pool_alloc, 0, frame_px * 4 * e->frame_slots + (size_t) k.active_width * 4 * 2 + 2 * HVK_POOL_PAD * 4));
	e->d_pool = e->d_pool_alloc + HVK_POOL_PAD;
	OPENHIP(hipMalloc((void **) &e->d_fdesc, sizeof(hvk_framedesc_t) * max_frames * 3));
	OPENHIP(hipMalloc((void **) &e->d_S, (size_t) max_frames * k.slab_lines * k.width * 2 + 256));
	if(k.s_video) OPENHIP(hipMalloc((void **) &e->d_C, (size_t) max_frames * k.slab_lines * k.width * 2 + 256));
	if(k.rs_irr)
	{
		OPENHIP(hipMalloc((void **) &e->d_frec, (size_t) max_frames * 2 * sizeof(int)));
		OPENHIP(hipHostMalloc((void **) &e->h_frec, (size_t) max_frames * 2 * sizeof(int), hipHostMallocDefault));
	}
	if(k.rs_L)
	{
		OPENHIP(hipMalloc((void **) &e->d_S2, (size_t) max_frames * k.s_stride * 2 + 256));
		if(k.s_video && !k.sv_ring) OPENHIP(hipMalloc((void **) &e->d_C2, (size_t) max_frames * k.s_stride * 2 + 256));
		if(k.s_video && k.sv_ring)
		{
			/* (a line's content can begin a sample in front of the line's own place in the stream -- the batch's first line's in
			 * front of the batch: the end of the batch before's stream stays in front of this one's) */
			e->sv_hist = (4 * e->t.max_width + 7) & ~7;
			const size_t n = (size_t) e->sv_hist + (size_t) max_frames * k.s_stride + 128;
			OPENHIP(hipMalloc((void **) &e->d_C2_alloc, n * 2));
			OPENHIP(hipMemset(e->d_C2_alloc, 0, n * 2));
			e->d_C2 = e->d_C2_alloc + e->sv_hist;
			OPENHIP(hipMalloc((void **) &e->d_Cq, ((size_t) max_frames * k.s_stride + 128) * 2));
			OPENHIP(hipMemset(e->d_Cq, 0, ((size_t) max_frames * k.s_stride + 128) * 2));
			OPENHIP(hipHostMalloc((void **) &e->h_svrec, (size_t) max_frames * k.lines * 16, hipHostMallocDefault));
			OPENHIP(hipMalloc((void **) &e->d_svrec, (size_t) max_frames * k.lines * 16));
			e->sv_tail_first = -1;
		}
		/* one 16-byte aligned row of 12 packed dwords per phase: pairs (t[0], t[1]) ... oldest sample
		 * first; a 21-tap phase gets a zero 22nd (hvk_k_resample stages the rows as they are) */
		std::vector<int> rows((size_t) k.rs_L * 12, 0);
		for(int ph = 0; ph < k.rs_L; ph++)
		{
			for(int m = 0; m < 12; m++)
			{
				const int lo = 2 * m < k.rs_ataps ? e->t.rs_taps[ph * k.rs_ataps + 2 * m] : 0;
				const int hi = 2 * m + 1 < k.rs_ataps ? e->t.rs_taps[ph * k.rs_ataps + 2 * m + 1] : 0;
				rows[(size_t) ph * 12 + m] = (lo & 0xFFFF) | (hi << 16);
			}
		}
		OPENCHK(_upload(&e->d_rs_taps, rows.data(), rows.size() * sizeof(int)));
	}
	OPENHIP(hipMalloc((void **) &e->d_out, (size_t) max_frames * FS * 4));
	OPENHIP(hipHostMalloc((void **) &e->h_fdesc, sizeof(hvk_framedesc_t) * max_frames * 3, hipHostMallocDefault));
	for(int i = 0; i < HVK_UPLOAD_RING; i++)
	{
		OPENHIP(hipHostMalloc((void **) &e->h_frame[i], frame_px * 4, hipHostMallocDefault));
		OPENHIP(hipEventCreateWithFlags(&e->up_ev[i], hipEventDisableTiming));
	}
	OPENHIP(hipEventCreateWithFlags(&e->ev_staged, hipEventDisableTiming));
	for(int i = 0; i < HVK_FETCH_TICKETS; i++) OPENHIP(hipEventCreateWithFlags(&e->fetch_ev[i], hipEventDisableTiming));

	if(e->t.k.has_carriers)
	{
		OPENHIP(hipMalloc((void **) &e->d_car, (size_t) max_frames * FS * 4 + 64));    /* (a lane that straddles the end of the last line loads 8 values all the same) */
		OPENHIP(hipMemset(e->d_car, 0, (size_t) max_frames * FS * 4 + 64));
		OPENHIP(hipHostMalloc((void **) &e->h_car, (size_t) max_frames * FS * 4, hipHostMallocDefault));
	}
	if(e->t.k.has_nicam)
	{
		e->tiles = (k.frame_samples + HVK_TILE - 1) / HVK_TILE;
		OPENHIP(hipMalloc((void **) &e->d_sym, (size_t) max_frames * e->symbol_stride * 4));
		OPENHIP(hipHostMalloc((void **) &e->h_sym, (size_t) max_frames * e->symbol_stride * 4, hipHostMallocDefault));
		OPENHIP(hipMalloc((void **) &e->d_tile, (size_t) max_frames * e->tiles * HVK_NICAM_ROW * 4));
		OPENHIP(hipHostMalloc((void **) &e->h_tile, (size_t) max_frames * e->tiles * HVK_NICAM_ROW * 4, hipHostMallocDefault));
		e->sym_tmp = (uint8_t *) malloc((size_t) e->symbol_stride * (e->t.k.rs_irr ? max_frames : 1));      /* (frames of two lengths: a batch's symbols in one go) */
		if(!e->sym_tmp) { *pe = NULL; hvk_close(e); return(HVK_OUT_OF_MEMORY); }
	}

	if(e->t.k.teletext)
	{
		OPENHIP(hipHostMalloc((void **) &e->h_tt_pk, (size_t) max_frames * 32 * 12 * 4, hipHostMallocDefault));
		OPENHIP(hipHostMalloc((void **) &e->h_tt_mask, (size_t) max_frames * 4, hipHostMallocDefault));
		memset(e->h_tt_pk, 0, (size_t) max_frames * 32 * 12 * 4);
		memset(e->h_tt_mask, 0, (size_t) max_frames * 4);
	}

	if(e->t.k.vbi)
	{
		if(e->t.vbi_nsym)
		{
			OPENCHK(_upload(&e->d_vbi_sym, e->t.vbi_sym, sizeof(int32_t) * 3 * e->t.vbi_nsym));
			OPENCHK(_upload(&e->d_vbi_val, e->t.vbi_val, sizeof(int16_t) * (e->t.vbi_total + 8)));
			/* The data lines as a gather (raster_compute()): per table and sample the RUN of consecutive symbols that lie over it
			 * -- its first symbol, its length -- and their values there: 16 dwords a sample. A table in which some sample's
			 * symbols are not one run of at most HVK_VBI_COVER keeps the walk over the set bits. */
			if(!(getenv("HVK_VBI_GATHER") && atoi(getenv("HVK_VBI_GATHER")) == 0))
			{
				const size_t W = (size_t) k.width;
				std::vector<int> cov((size_t) HVK_VBI_LUTS * W * 16 + 8 * 16, 0);
				std::vector<int> b0(W), n(W);
				for(int u = 0; u < HVK_VBI_LUTS; u++)
				{
					e->vbi_cov_ok[u] = e->t.lut_nsym[u] > 0 && e->t.lut_nsym[u] < 0x7000;
					std::fill(b0.begin(), b0.end(), 0);
					std::fill(n.begin(), n.end(), 0);
					for(int sy = 0; sy < e->t.lut_nsym[u] && e->vbi_cov_ok[u]; sy++)
					{
						const int32_t *q = e->t.vbi_sym + (size_t) (e->t.lut_base[u] + sy) * 3;
						for(int j = 0; j < q[1]; j++)
						{
							const long x = (long) q[0] + j;
							if(x < 0 || x >= (long) W) continue;
							if(n[(size_t) x] == 0) b0[(size_t) x] = sy;
							if(sy != b0[(size_t) x] + n[(size_t) x] || n[(size_t) x] >= HVK_VBI_COVER) { e->vbi_cov_ok[u] = 0; break; }
							int *ent = cov.data() + ((size_t) u * W + (size_t) x) * 16;
							const int m = n[(size_t) x]++;
							const unsigned v = (unsigned) (uint16_t) e->t.vbi_val[q[2] + j];
							ent[1 + m / 2] |= (int) ((m & 1) ? (v << 16) : v);
							ent[0] = b0[(size_t) x] | (n[(size_t) x] << 16);
						}
					}
				}
				OPENCHK(_upload(&e->d_vbi_cov, cov.data(), cov.size() * sizeof(int)));
			}
		}
		OPENHIP(hipMalloc((void **) &e->d_ops, (size_t) max_frames * HVK_VBI_OPS * HVK_VBI_OPWORDS * 4));
		OPENHIP(hipHostMalloc((void **) &e->h_ops, (size_t) max_frames * HVK_VBI_OPS * HVK_VBI_OPWORDS * 4, hipHostMallocDefault));
		OPENHIP(hipMalloc((void **) &e->d_map, (size_t) max_frames * k.lines));
		OPENHIP(hipHostMalloc((void **) &e->h_map, (size_t) max_frames * k.lines, hipHostMallocDefault));
	}
	if(e->t.fsc_rows) OPENCHK(_upload(&e->d_fsc_rows, e->t.fsc_rows, sizeof(int16_t) * 2 * (size_t) k.width + 64));
	if(e->t.k.vits)
	{
		OPENCHK(_upload(&e->d_vits_l, e->t.vits_l, sizeof(int16_t) * k.vits * k.width));
		OPENCHK(_upload(&e->d_vits_c, e->t.vits_c, sizeof(int16_t) * k.vits * k.width));
	}

	if(k.sis)
	{
		OPENCHK(_upload(&e->d_sis_dense, e->t.sis_dense, sizeof(int16_t) * 50 * HVK_SIS_SPAN));
		OPENCHK(_upload(&e->d_sis_win, e->t.sis_win, sizeof(int16_t) * k.sis_width));
		OPENCHK(_upload(&e->d_sis_first, e->t.sis_first, sizeof(int16_t) * HVK_SIS_SPAN));
		OPENHIP(hipMalloc((void **) &e->d_sis_bits, (size_t) max_frames * (k.lines + 2) * 8));
		OPENHIP(hipHostMalloc((void **) &e->h_sis_bits, (size_t) max_frames * (k.lines + 2) * 8, hipHostMallocDefault));
	}

	if(k.rawbb)
	{
		const size_t bytes = (size_t) max_frames * k.slab_lines * k.width * 2;
		OPENHIP(hipMalloc((void **) &e->d_raw, bytes));
		OPENHIP(hipHostMalloc((void **) &e->h_raw, bytes, hipHostMallocDefault));
	}

	if(e->t.k.secam)
	{
		const size_t RS = k.raster_samples;   /* the colour side stream is at the pixel rate */
		/* The kept sub-carrier: six rows per picture slot (one per frame number modulo 6) behind the batch's rows, for as many
		 * slots as a 32-bit sample index and a budget of 1 GB allow (hvk_engine_stage.cpp) */
		e->secam_memo_slots = 0;
		if(k.fields == 1 && !(getenv("HVK_SECAM_KEEP") && atoi(getenv("HVK_SECAM_KEEP")) == 0) && !getenv("HVK_SECAM_NO_CELL_CACHE") && !getenv("HVK_SECAM_NO_SEEDS") && !getenv("HVK_SECAM_HOST"))
		{
			int64_t rows = (int64_t) 0x7FFF0000 / (int64_t) RS - max_frames - 2;
			if(rows > (int64_t) (1e9 / (double) (RS * 2))) rows = (int64_t) (1e9 / (double) (RS * 2));
			e->secam_memo_slots = (int) (rows / 6 < e->frame_slots ? (rows / 6 > 0 ? rows / 6 : 0) : e->frame_slots);
		}
		const size_t crows = (size_t) max_frames + 6 * (size_t) e->secam_memo_slots;
		/* (a line's worth of zeros behind the rows: what hvk_k_direct adds to the lines around a frame) */
		OPENHIP(hipMalloc((void **) &e->d_chroma_alloc, (crows * RS + k.width + 128) * 2));
		OPENHIP(hipMemset(e->d_chroma_alloc, 0, (crows * RS + k.width + 128) * 2));
		e->d_chroma = e->d_chroma_alloc + 64;
		e->chroma_par = (signed char *) malloc((size_t) max_frames);
		if(!e->chroma_par) { hvk_close(e); return(HVK_OUT_OF_MEMORY); }
		memset(e->chroma_par, -1, (size_t) max_frames);
		OPENHIP(hipHostMalloc((void **) &e->h_chroma, (size_t) max_frames * RS * 2, hipHostMallocDefault));

		/* the device's own chain (HVK_SECAM_HOST=1: everything through the host's, as before). It needs whole
		 * 8-sample chunks, the FM loop's overhang within the low pass's reach, and no picture that close to the
		 * line's start */
		const int n0 = hvk_secam_tasks(&e->t, 0, NULL, 0), n1 = hvk_secam_tasks(&e->t, 1, NULL, 0);
		const int over = k.burst_left + k.burst_width - k.width;
		if(!getenv("HVK_SECAM_HOST") && k.width % 16 == 0 && k.width <= 2048 && over <= HVK_SECAM_TAIL && k.active_left >= 8 && n0 > 0 && n1 > 0)
		{
			hvk_secam_args_t &a = e->sa;
			memset(&a, 0, sizeof(a));
			a.C.W = k.width;
			a.C.sl = k.burst_left;
			a.C.level = e->t.secam_level;
			for(int i = 0; i < 2; i++) { a.C.dmin[i] = e->t.secam_dmin[i]; a.C.dmax[i] = e->t.secam_dmax[i]; }
			for(int i = 0; i < 15; i++) a.C.fir[i] = e->t.secam_fir[i];
			a.lines = k.lines; a.hline = e->t.conf.hline; a.fields = k.fields; a.interlaced = k.interlaced;
			a.active_left = k.active_left; a.active_width = k.active_width; a.active_lines = k.active_lines;
			a.burst_left = k.burst_left; a.burst_width = k.burst_width;
			a.ntasks = 2 + (n0 > n1 ? n0 : n1);
			a.tpad = (max_frames * a.ntasks + 63) & ~63;
			/* the cell stores: a set of rows per picture slot and frame parity where a frame shows one picture, else
			 * (--interlace: two pictures) a set per frame of the batch */
			e->secam_cell_cache = k.fields == 1 && !getenv("HVK_SECAM_NO_CELL_CACHE");
			a.cpad = e->secam_cell_cache ? ((e->frame_slots * 2 > max_frames ? e->frame_slots * 2 : max_frames) * a.ntasks + 63) & ~63 : a.tpad;
			a.K = HVK_SECAM_WARMUP;
			e->secam_adapt = getenv("HVK_SECAM_WARMUP") == NULL;
			e->secam_patience = 1;
			if(getenv("HVK_SECAM_WARMUP")) a.K = atoi(getenv("HVK_SECAM_WARMUP"));
			if(a.K < 0) a.K = 0;
			a.raster_samples = (int64_t) RS;

			std::vector<hvk_secam_task_t> tasks((size_t) 2 * a.ntasks);
			memset(tasks.data(), 0, tasks.size() * sizeof(hvk_secam_task_t));
			hvk_secam_tasks(&e->t, 0, tasks.data() + 2, n0);
			hvk_secam_tasks(&e->t, 1, tasks.data() + a.ntasks + 2, n1);
			std::vector<int16_t> fid((size_t) 2 * k.width);
			{
				/* rest values of RGB 000000: the level table's first entry, computed like the table (hvk_secam.c) */
				std::vector<int16_t> q(4);
				if(hipStreamSynchronize(e->stream) != hipSuccess || hipMemcpy(q.data(), e->d_yuv, 8, hipMemcpyDeviceToHost) != hipSuccess) { hvk_close(e); return(HVK_ERROR); }
				hvk_secam_fid_row(&e->t, 0, q[1], fid.data());
				hvk_secam_fid_row(&e->t, 1, q[2], fid.data() + k.width);
			}
			/* where a frame's second field begins in the task list of either parity: the second task that clears what lies
			 * behind the line (HVK_SECAM_REDO_FIELDS=0: the stretch-by-stretch redo rounds of before) */
			for(int p = 0; p < 2; p++)
			{
				int seen = 0;
				a.half_slot[p] = 0;
				for(int s_ = 2; s_ < a.ntasks; s_++)
				{
					const hvk_secam_task_t &q = tasks[(size_t) p * a.ntasks + s_];
					if((q.flags & HVK_SECAM_TASK_VALID) && (q.flags & HVK_SECAM_TASK_CLEAR) && ++seen == 2) { a.half_slot[p] = s_; break; }
				}
				if(getenv("HVK_SECAM_REDO_FIELDS") && atoi(getenv("HVK_SECAM_REDO_FIELDS")) == 0) a.half_slot[p] = 0;
			}
			OPENCHK(_upload(&e->d_secam[0], tasks.data(), tasks.size() * sizeof(hvk_secam_task_t)));
			OPENCHK(_upload(&e->d_secam[1], fid.data(), fid.size() * 2));
			OPENCHK(_upload(&e->d_secam[2], e->t.secam_lut, 65536 * sizeof(hvk_c32_t)));
			OPENCHK(_upload(&e->d_secam[3], e->t.secam_bell, 65536 * sizeof(hvk_c16_t)));
			{
				std::vector<int32_t> lb((size_t) 65536 * 4);
				for(int u = 0; u < 65536; u++)
				{
					const hvk_c16_t gq = e->t.secam_bell[(u - 32768) & 0xFFFF];
					lb[(size_t) u * 4 + 0] = e->t.secam_lut[u].i;
					lb[(size_t) u * 4 + 1] = e->t.secam_lut[u].q;
					lb[(size_t) u * 4 + 2] = (int32_t) ((uint32_t) (uint16_t) gq.i | ((uint32_t) (uint16_t) gq.q << 16));
					lb[(size_t) u * 4 + 3] = 0;
				}
				OPENCHK(_upload(&e->d_secam[14], lb.data(), lb.size() * 4));
			}
			OPENHIP(hipMalloc(&e->d_secam[4], (size_t) a.cpad * k.width * 2));
			OPENHIP(hipMalloc(&e->d_secam[5], (size_t) a.cpad * 32));
			OPENHIP(hipMalloc(&e->d_secam[6], (size_t) a.tpad * sizeof(hvk_secam_state_t)));
			OPENHIP(hipMalloc(&e->d_secam[7], (size_t) a.tpad * sizeof(hvk_secam_state_t)));
			OPENHIP(hipMalloc(&e->d_secam[8], sizeof(hvk_secam_state_t) + 64));
			if(!getenv("HVK_SECAM_NO_MID"))
			{
				/* what a line's walk had in hand in front of its last eight samples (a start wrong only in the values behind the
				 * line: hvk_k_secam_redo walks those eight again, not the line) */
				OPENHIP(hipMalloc(&e->d_secam[19], (size_t) a.tpad * sizeof(hvk_secam_mid_t)));
				OPENHIP(hipMemset(e->d_secam[19], 0, (size_t) a.tpad * sizeof(hvk_secam_mid_t)));
				a.mid = (hvk_secam_mid_t *) e->d_secam[19];
			}
			OPENHIP(hipMalloc(&e->d_secam[9], (size_t) a.tpad * 4 + 64));
			OPENHIP(hipMalloc(&e->d_secam[10], (size_t) max_frames * 8 * sizeof(int)));
			OPENHIP(hipMemset(e->d_secam[10], 0, (size_t) max_frames * 8 * sizeof(int)));
			e->secam_seeds = e->secam_cell_cache && !getenv("HVK_SECAM_NO_SEEDS");
			if(e->secam_seeds)
			{
				OPENHIP(hipMalloc(&e->d_secam[11], (size_t) 3 * a.cpad * sizeof(hvk_secam_state_t)));
				OPENHIP(hipMemset(e->d_secam[11], 0, (size_t) 3 * a.cpad * sizeof(hvk_secam_state_t)));    /* (a state of nothing: what every warm-up started from before) */
			}
			/* entry states of new pictures' lines by estimate instead of warm-up walks (hvk_k_secam_est); a pinned warm-up
			 * length (HVK_SECAM_WARMUP) keeps the walks */
			a.x1 = (k.burst_left + 78 + 7) & ~7;
			e->secam_est = e->secam_adapt && a.x1 + 16 <= k.width && k.width >= 512 && !(getenv("HVK_SECAM_EST") && atoi(getenv("HVK_SECAM_EST")) == 0);
			if(e->secam_est)
			{
				OPENHIP(hipMalloc(&e->d_secam[12], (size_t) a.cpad * sizeof(double)));
				OPENHIP(hipMemset(e->d_secam[12], 0, (size_t) a.cpad * sizeof(double)));
				OPENHIP(hipMalloc(&e->d_secam[13], (size_t) a.tpad * 32));
				OPENHIP(hipMemset(e->d_secam[13], 0, (size_t) a.tpad * 32));
				a.iya = (double *) e->d_secam[12];
				a.est = (int16_t *) e->d_secam[13];
				a.ES = getenv("HVK_SECAM_EST_RUN") ? atoi(getenv("HVK_SECAM_EST_RUN")) : 4;
				a.EK = getenv("HVK_SECAM_EST_LINES") ? atoi(getenv("HVK_SECAM_EST_LINES")) : 16;
				e->secam_ek_adapt = getenv("HVK_SECAM_EST_LINES") == NULL;
				e->secam_ek_base = a.EK;
				if(a.ES < 1) a.ES = 1;
				if(a.EK < 1) a.EK = 1;
				/* a step's angle, src/video.c:2236 with :4080 */
				a.kap0 = 2.0 * M_PI / (double) e->t.pixel_rate * 4328125.0;
				a.kap1 = 2.0 * M_PI / (double) e->t.pixel_rate * 1000e3 / (double) INT16_MAX;
				if(!getenv("HVK_SECAM_NO_RES"))
				{
					/* the table entries' rounding (hvk_secam_args_t.res): angle and relative length against the nominal ones */
					std::vector<int16_t> res((size_t) 65536 * 2);
					for(int u = 0; u < 65536; u++)
					{
						const double i_ = e->t.secam_lut[u].i, q_ = e->t.secam_lut[u].q;
						double dphi = atan2(q_, i_) - (a.kap0 + a.kap1 * (double) (u - 32768));
						dphi -= 2.0 * M_PI * floor(dphi / (2.0 * M_PI) + 0.5);
						const double dlen = sqrt(i_ * i_ + q_ * q_) / 2147483647.0 - 1.0;
						const double p_ = round(dphi * 0x1p46), l_ = round(dlen * 0x1p46);
						res[(size_t) u * 2 + 0] = (int16_t) (p_ < -32767 ? -32767 : (p_ > 32767 ? 32767 : p_));
						res[(size_t) u * 2 + 1] = (int16_t) (l_ < -32767 ? -32767 : (l_ > 32767 ? 32767 : l_));
					}
					OPENCHK(_upload(&e->d_secam[15], res.data(), res.size() * 2));
					OPENHIP(hipMalloc(&e->d_secam[16], (size_t) a.cpad * 2 * sizeof(int32_t)));
					OPENHIP(hipMemset(e->d_secam[16], 0, (size_t) a.cpad * 2 * sizeof(int32_t)));
					a.res = (const int16_t *) e->d_secam[15];
					a.corr = (int32_t *) e->d_secam[16];
				}
			}
			{
				/* hvk_k_secam_walk<1> (hvk_secam_args_t.phc): the coarse phasors, the bell filter's gains in 32-index blocks --
				 * and every index of the deviation range tried on the device against the tables before the kernel may be taken */
				const double rate = (double) e->t.pixel_rate, fm_dev = 1000e3, fm_freq = 4328125;       /* src/video.c:45-46 */
				std::vector<double> phc((size_t) 513 * 2);
				for(int i = 0; i < 513; i++)
				{
					const double d = 2.0 * M_PI / rate * (fm_freq + (double) (i * 128 - 32768) / INT16_MAX * fm_dev);
					phc[(size_t) i * 2 + 0] = cos(d) * INT32_MAX;
					phc[(size_t) i * 2 + 1] = sin(d) * INT32_MAX;
				}
				a.ph_k1 = 2.0 * M_PI / rate * (fm_dev / INT16_MAX);
				const int lo = a.C.dmin[0] < a.C.dmin[1] ? a.C.dmin[0] : a.C.dmin[1], hi = a.C.dmax[0] > a.C.dmax[1] ? a.C.dmax[0] : a.C.dmax[1];
				a.bell_c0 = lo & ~31;
				a.bell_blocks = (hi - a.bell_c0) / 32 + 1;
				std::vector<uint32_t> bz((size_t) a.bell_blocks * 4, 0);
				bool ok = lo > INT16_MIN + 64 && hi < INT16_MAX - 64 && a.bell_blocks <= 4096;
				for(int b = 0; b < a.bell_blocks && ok; b++)
				{
					const int first = a.bell_c0 + 32 * b;
					hvk_c16_t g0 = e->t.secam_bell[(uint16_t) (int16_t) first];
					bz[(size_t) b * 4] = (uint32_t) (uint16_t) g0.i | ((uint32_t) (uint16_t) g0.q << 16);
					for(int t = 0; t < 31 && first + t + 1 <= hi; t++)
					{
						const hvk_c16_t g1 = e->t.secam_bell[(uint16_t) (int16_t) (first + t + 1)];
						const int di = g1.i - g0.i, dq = g1.q - g0.q;
						if(dq < 0 || dq > 1 || di < -1 || di > 1) { ok = false; break; }
						if(dq) bz[(size_t) b * 4 + 1] |= 1u << t;
						if(di > 0) bz[(size_t) b * 4 + 2] |= 1u << t;
						if(di < 0) bz[(size_t) b * 4 + 3] |= 1u << t;
						g0 = g1;
					}
				}
				e->secam_walk_ok = 1;
				if(ok)
				{
					OPENCHK(_upload(&e->d_secam[17], phc.data(), phc.size() * sizeof(double)));
					OPENCHK(_upload(&e->d_secam[18], bz.data(), bz.size() * 4));
					a.phc = (const double *) e->d_secam[17];
					a.bellz = (const uint32_t *) e->d_secam[18];
					a.lut = (const hvk_secam_c32_t *) e->d_secam[2];
					a.bell = (const hvk_secam_c16_t *) e->d_secam[3];
					void *d_n = NULL;
					int differ = -1;
					OPENHIP(hipMalloc(&d_n, sizeof(int)));
					OPENCHK(hvk_launch_secam_check_walk(&a, (int *) d_n, e->stream));
					OPENHIP(hipMemcpyAsync(&differ, d_n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
					OPENHIP(hipStreamSynchronize(e->stream));
					(void) hipFree(d_n);
					if(differ == 0) e->secam_walk_ok = 2;
					else { a.phc = NULL; a.bellz = NULL; }
					if(getenv("HVK_SHIM_STATS")) fprintf(stderr, "libhvk: SECAM FM steps computed / gains from blocks: %d of %d indices differ from the tables\n", differ, hi - lo + 1);
				}
				e->secam_walk_mode = getenv("HVK_SECAM_WALK") ? atoi(getenv("HVK_SECAM_WALK")) : -1;
			}
			OPENHIP(hipMemset(e->d_secam[4], 0, (size_t) a.cpad * k.width * 2));
			OPENHIP(hipMemset(e->d_secam[5], 0, (size_t) a.cpad * 32));
			OPENHIP(hipHostMalloc((void **) &e->h_secam_rows, (size_t) max_frames * 8 * sizeof(int), hipHostMallocDefault));
			e->secam_prev_key = (int64_t *) calloc((size_t) max_frames, sizeof(int64_t));
			if(!e->secam_prev_key) OPENCHK(HVK_OUT_OF_MEMORY);
			memset(e->h_secam_rows, 0, (size_t) max_frames * 8 * sizeof(int));
			if(e->secam_memo_slots > 0)
			{
				OPENHIP(hipMalloc(&e->d_secam[20], (size_t) 6 * e->secam_memo_slots * sizeof(hvk_secam_state_t)));
				OPENHIP(hipMemset(e->d_secam[20], 0, (size_t) 6 * e->secam_memo_slots * sizeof(hvk_secam_state_t)));
			}
			OPENHIP(hipMemset(e->d_secam[8], 0, sizeof(hvk_secam_state_t) + 64));
			OPENHIP(hipHostMalloc((void **) &e->h_secam_count, 64, hipHostMallocDefault));
			OPENHIP(hipEventCreateWithFlags(&e->secam_ev, hipEventDisableTiming));
			e->secam_defer = !(getenv("HVK_SECAM_DEFER") && atoi(getenv("HVK_SECAM_DEFER")) == 0);
			OPENHIP(hipHostMalloc((void **) &e->h_secam_carry, sizeof(hvk_secam_state_t), hipHostMallocDefault));
			memset(e->h_secam_carry, 0, sizeof(hvk_secam_state_t));
			a.tasks = (const hvk_secam_task_t *) e->d_secam[0];
			a.fid_rows = (const int16_t *) e->d_secam[1];
			a.lut = (const hvk_secam_c32_t *) e->d_secam[2];
			a.bell = (const hvk_secam_c16_t *) e->d_secam[3];
			a.lutb = e->d_secam[14];
			a.F = (int16_t *) e->d_secam[4];
			a.acc = (int32_t *) e->d_secam[5];
			a.entry = (hvk_secam_state_t *) e->d_secam[6];
			a.exit = (hvk_secam_state_t *) e->d_secam[7];
			a.carry = (hvk_secam_state_t *) e->d_secam[8];
			a.flags = (int *) e->d_secam[9];
			a.count = a.flags + a.tpad;
			a.cbase = (const int *) e->d_secam[10];
			a.clist = a.cbase + max_frames;
			a.seed = (hvk_secam_state_t *) e->d_secam[11];
			a.kf = e->secam_seeds && e->secam_adapt ? a.cbase + 2 * max_frames : NULL;
			a.sbase = a.cbase + 3 * max_frames;
			a.orow = a.cbase + 6 * max_frames;        /* (every frame has a row; which frames take or make a kept set is the stage's to say: mflag, owner) */
			a.mrow = a.cbase + 7 * max_frames;
			a.seedx = (hvk_secam_state_t *) e->d_secam[20];
			a.desc = (const hvk_linedesc_t *) e->d_desc;
			a.pool = e->d_pool;
			a.yuv = e->d_yuv;
			a.burst_win = (const int16_t *) e->d_burst + HVK_PULSE_PAD;
			a.chroma = e->d_chroma;
			{
				hipDeviceProp_t prop;
				e->secam_lanes = 1024 * 64 * 8;
				if(hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) e->secam_lanes = prop.multiProcessorCount * 4 * 64 * 8;
			}
			e->secam_dev = 1;
		}
	}

	if(e->t.k.fm_video)
	{
		OPENHIP(hipHostMalloc((void **) &e->h_fm, (size_t) max_frames * FS * 4, hipHostMallocDefault));
		/* (--passthru: the caller's thread fills the queue the phasor's pass reads -- no thread beside it then. HVK_FM_SYNC=1:
		 * the pass in the caller's thread, as before) */
		if(!e->t.k.has_passthru && !getenv("HVK_FM_SYNC"))
		{
			e->fm_mu = new std::mutex;
			e->fm_cv = new std::condition_variable;
			e->fm_q = new std::deque<hvk_engine::fm_job_t>;
			e->fm_thread = new std::thread(hvk_e_fm_worker, e);
		}
	}
	else
	{
		if(e->t.k.has_offset)
		{
			OPENHIP(hipMalloc((void **) &e->d_off, (size_t) max_frames * FS * 4));
			OPENHIP(hipHostMalloc((void **) &e->h_off, (size_t) max_frames * FS * 4, hipHostMallocDefault));
		}
		if(e->t.k.has_passthru)
		{
			OPENHIP(hipMalloc((void **) &e->d_pass, (size_t) max_frames * FS * 4));
			OPENHIP(hipHostMalloc((void **) &e->h_pass, (size_t) max_frames * FS * 4, hipHostMallocDefault));
		}
	}

	for(int i = 0; i < HVK_TIMING_SLOTS; i++) for(int j = 0; j < 3; j++) OPENHIP(hipEventCreate(&e->ev[i][j]));

	OPENHIP(hipStreamSynchronize(e->stream));
	return(HVK_OK);
}

extern "C" void hvk_close(hvk_engine_t *e)
{
	if(!e) return;

	if(e->fm_thread)
	{
		{ std::lock_guard<std::mutex> lk(*e->fm_mu); e->fm_quit = 1; }
		e->fm_cv->notify_all();
		e->fm_thread->join();
		delete e->fm_thread; delete e->fm_q; delete e->fm_cv; delete e->fm_mu;
		e->fm_thread = NULL;
	}

	if(e->device >= 0)
	{
		(void) hipSetDevice(e->device);
		if(e->stream) (void) hipStreamSynchronize(e->stream);
		for(int i = 0; i < HVK_TIMING_SLOTS; i++) for(int j = 0; j < 3; j++) if(e->ev[i][j]) (void) hipEventDestroy(e->ev[i][j]);
		void *dev[] = { e->d_yuv, e->d_yuvparams, e->d_desc, e->d_pulses, e->d_linebase, e->d_clut, e->d_burst, e->d_ghost, e->d_Lp, e->d_Cp, e->d_UVp, e->d_clut3, e->d_lineoff,
		                e->d_tapd, e->d_cca, e->d_pool_alloc, e->d_fdesc, e->d_S, e->d_car, e->d_sym, e->d_tile, e->d_out, e->d_chroma_alloc, e->d_vbi_sym, e->d_vbi_val, e->d_vbi_cov, e->d_ops, e->d_map, e->d_vits_l, e->d_vits_c, e->d_sis_dense, e->d_sis_win, e->d_sis_first, e->d_sis_bits, e->d_conv, e->d_off, e->d_pass, e->d_S2, e->d_C2_alloc ? (void *) e->d_C2_alloc : (void *) e->d_C2, e->d_Cq, e->d_svrec, e->d_rs_taps, e->d_C, e->d_raw, e->d_mfma_a, e->d_frec, e->d_ovr_list, e->d_ovr_idx, e->d_sums, e->d_mfma_a28, e->d_tilerec, e->d_fsc_rows };
		for(void *p : dev) if(p) (void) hipFree(p);
		for(int i = 0; i < HVK_UPLOAD_RING; i++) { if(e->h_frame[i]) (void) hipHostFree(e->h_frame[i]); if(e->up_ev[i]) (void) hipEventDestroy(e->up_ev[i]); }
		for(void *p : e->d_secam) if(p) (void) hipFree(p);
		if(e->h_secam_count) (void) hipHostFree(e->h_secam_count);
		if(e->secam_ev) (void) hipEventDestroy(e->secam_ev);
		if(e->h_secam_carry) (void) hipHostFree(e->h_secam_carry);
		if(e->ev_staged) (void) hipEventDestroy(e->ev_staged);
		if(e->ev_fork) (void) hipEventDestroy(e->ev_fork);
		for(int i = 0; i < HVK_PREP_EVENTS; i++) if(e->ev_prep[i]) (void) hipEventDestroy(e->ev_prep[i]);
		if(e->prep_stream) { (void) hipStreamSynchronize(e->prep_stream); (void) hipStreamDestroy(e->prep_stream); }
		if(e->copy_stream) { (void) hipStreamSynchronize(e->copy_stream); (void) hipStreamDestroy(e->copy_stream); }
		if(e->copy_out_ev) (void) hipEventDestroy(e->copy_out_ev);
		for(int i = 0; i < HVK_FETCH_TICKETS; i++) if(e->fetch_ev[i]) (void) hipEventDestroy(e->fetch_ev[i]);
		if(e->h_svrec) (void) hipHostFree(e->h_svrec);
		void *host[] = { e->h_fdesc, e->h_car, e->h_sym, e->h_tile, e->h_chroma, e->h_tt_pk, e->h_tt_mask, e->h_ops, e->h_map, e->h_off, e->h_pass, e->h_fm, e->h_raw, e->h_sis_bits, e->h_secam_rows, e->h_frec };
		for(void *p : host) if(p) (void) hipHostFree(p);
		if(e->own_stream) (void) hipStreamDestroy(e->own_stream);
	}

	free(e->sym_tmp);
	free(e->secam_prev_key);
	free(e->chroma_par);
	free(e->h_ovr_idx);
	free(e->fm_prime_car);
	free(e->cc_pairs);
	delete e->raw_q;
	if(e->host_frames) { for(int i = 0; i < e->frame_slots; i++) free(e->host_frames[i]); free(e->host_frames); }
	hvk_secam_free(e->secam);
	hvk_tail_free(e->tail);
	free(e->slots);
	free(e->staged_slots);
	free(e->staged_slots2);
	free(e->staged_prev);
	hvk_audio_free(e->audio);
	hvk_tables_free(&e->t);
	free(e);
}

extern "C" int hvk_get_info(const hvk_engine_t *e, hvk_info_t *info)
{
	if(!e || !info) return(HVK_ERROR);
	if(info->struct_size != sizeof(hvk_info_t)) return(HVK_ERROR);     /* (the caller's layout is not this library's: nothing is written) */
	const hvk_kconst_t &k = e->t.k;
	info->sample_rate = e->t.sample_rate;
	info->width = k.width;
	info->half_width = k.half_width;
	info->active_width = k.active_width;
	info->active_left = k.active_left;
	info->lines = k.lines;
	info->active_lines = k.active_lines;
	info->white_level = e->t.white_level;
	info->black_level = e->t.black_level;
	info->blanking_level = e->t.blanking_level;
	info->sync_level = e->t.sync_level;
	info->delay_lines = k.delay_lines;
	info->frame_samples = k.frame_samples;
	info->max_frames = e->max_frames;
	info->frame_slots = e->frame_slots;
	info->colour_lookup_width = k.clw;
	info->burst_left = k.burst_left;
	info->burst_width = k.burst_width;
	info->has_carriers = k.has_carriers;
	info->has_nicam = k.has_nicam;
	info->pixel_rate = e->t.pixel_rate;
	info->max_width = e->t.max_width;
	info->startup_samples = k.out_prime;
	return(HVK_OK);
}

extern "C" size_t hvk_get_framebuffer_length(const hvk_engine_t *e)
{
	return(sizeof(uint32_t) * e->t.k.active_width * e->t.k.active_lines);
}

extern "C" int hvk_set_chroma_ghost(hvk_engine_t *e, const int16_t *ghost, int n)
{
	if(!e || n < 0 || n > HVK_GHOST_LEN) return(HVK_ERROR);
	memset(e->t.ghost, 0, sizeof(e->t.ghost));
	if(ghost) memcpy(e->t.ghost, ghost, n * sizeof(int16_t));
	else hvk_tables_default_ghost(&e->t);
	for(int i = 0; i < e->frame_slots; i++) e->slots[i].plane_dirty = 1;     /* the chroma low pass reads them (SECAM's cells do not) */
	if(e->device >= 0 && e->d_ghost)
	{
		HIPCHK(hipSetDevice(e->device));
		HIPCHK(hipMemcpyAsync(e->d_ghost, e->t.ghost, sizeof(e->t.ghost), hipMemcpyHostToDevice, e->stream));
		HIPCHK(hipStreamSynchronize(e->stream));
	}
	return(HVK_OK);
}

extern "C" int hvk_get_chroma_ghost(const hvk_engine_t *e, int16_t *ghost, int n)
{
	if(!e || !ghost || n < 0 || n > HVK_GHOST_LEN) return(HVK_ERROR);
	memcpy(ghost, e->t.ghost, n * sizeof(int16_t));
	return(HVK_OK);
}

extern "C" long hvk_table(const hvk_engine_t *e, const char *name, void *dst, long max_bytes)
{
	if(!e || !name) return(-1);

	if(!strcmp(name, "yuv"))
	{
		/* the device-expanded table, read back as the reference's {y,u,v} triples */
		const long bytes = 0x1000000L * 6;
		if(dst == NULL) return(bytes);
		if(e->device < 0) return(-1);
		std::vector<int16_t> q(0x1000000UL * 4);
		if(hipSetDevice(e->device) != hipSuccess) return(-1);
		if(hipMemcpy(q.data(), e->d_yuv, q.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return(-1);
		int16_t *o = (int16_t *) dst;
		long n = max_bytes / 6 < 0x1000000L ? max_bytes / 6 : 0x1000000L;
		for(long i = 0; i < n; i++) { o[i * 3 + 0] = q[i * 4 + 0]; o[i * 3 + 1] = q[i * 4 + 1]; o[i * 3 + 2] = q[i * 4 + 2]; }
		return(n * 6);
	}

	return(hvk_tables_get(&e->t, name, dst, max_bytes));
}

/* ---- inputs ---- */

/* How many colours? 4096 pixels on a regular grid, counted through a small open hash. Graphics and
 * test cards have a few hundred, and the same ones in every frame: their level-table entries stay
 * in cache. Camera pictures have tens of thousands per frame: most look-ups would miss.
 * px: a w x h picture whose rows are `pitch` pixels apart. */
static bool _many_colours(const uint32_t *px, int pitch, int w, int h)
{
	enum { SAMPLES = 4096, SLOTS = 8192, MANY = 1024 };
	static const uint32_t EMPTY = 0xFFFFFFFFu;
	std::vector<uint32_t> seen(SLOTS, EMPTY);
	const size_t npx = (size_t) w * h;
	int distinct = 0;
	for(int i = 0; i < SAMPLES && npx > 0; i++)
	{
		const size_t at = (size_t) ((uint64_t) i * npx / SAMPLES);
		const uint32_t c = px[(at / w) * (size_t) pitch + at % w] & 0xFFFFFFu;
		uint32_t hsh = (c * 2654435761u) >> 19;      /* 13 bits */
		while(seen[hsh] != EMPTY && seen[hsh] != c) hsh = (hsh + 1) & (SLOTS - 1);
		if(seen[hsh] == EMPTY) { seen[hsh] = c; distinct++; }
	}
	return(distinct > MANY);
}

extern "C" int hvk_frame_upload(hvk_engine_t *e, int slot, const uint32_t *fb, int width, int height,
                                int pixel_stride, int line_stride, int interlaced)
{
	if(!e || slot < 0 || slot >= e->frame_slots) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);

	const hvk_kconst_t &k = e->t.k;
	hvk_slot_t *s = &e->slots[slot];

	/* centre crop to the active area: src/video.c:4887-4893, src/av.c:293-303 */
	int x = (width - k.active_width) / 2, y = (height - k.active_lines) / 2;
	int w = k.active_width, h = k.active_lines;
	if(x < 0) { w += x; x = 0; }
	if(y < 0) { h += y; y = 0; }
	if(x + w > width) w = width - x;
	if(y + h > height) h = height - y;
	if(w < 0) w = 0;
	if(h < 0) h = 0;

	s->valid = fb != NULL && w > 0 && h > 0;
	s->width = fb ? w : 0;
	s->height = fb ? h : 0;
	s->interlaced = interlaced;
	s->plane_dirty = 1;
	s->shown = 0;
	s->cells_valid[0] = s->cells_valid[1] = 0;
	memset(s->seeds_valid, 0, sizeof(s->seeds_valid));
	memset(s->memo_valid, 0, sizeof(s->memo_valid));
	s->gen++;
	if(fb == NULL)
	{
		/* av_read_video() past the end hands back an empty frame (src/av.c:55-59) */
		return(HVK_OK);
	}

	HIPCHK(hipSetDevice(e->device));
	/* the next staging buffer of the ring; its last copy (HVK_UPLOAD_RING uploads ago) has to be through */
	const int ub = e->up_next;
	e->up_next = (e->up_next + 1) % HVK_UPLOAD_RING;
	if(e->up_busy[ub]) { HIPCHK(hipEventSynchronize(e->up_ev[ub])); e->up_busy[ub] = 0; }
	uint32_t *const stage = e->h_frame[ub];

	/* gather into a dense w x h image: strides may be negative (flips) */
	const uint32_t *src = fb + (int64_t) y * line_stride + (int64_t) x * pixel_stride;
	for(int r = 0; r < h; r++)
	{
		const uint32_t *p = src + (int64_t) r * line_stride;
		uint32_t *o = stage + (size_t) r * w;
		if(pixel_stride == 1) memcpy(o, p, (size_t) w * 4);
		else for(int c = 0; c < w; c++) o[c] = p[(int64_t) c * pixel_stride];
	}

	s->many_colours = _many_colours(stage, w, w, h);

	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	if(e->secam)
	{
		/* the SECAM pre-pass reads the picture on the host */
		if(!e->host_frames[slot]) e->host_frames[slot] = (uint32_t *) malloc(frame_px * 4);
		if(!e->host_frames[slot]) return(HVK_OUT_OF_MEMORY);
		memcpy(e->host_frames[slot], stage, (size_t) w * h * 4);
	}
	/* on the engine's stream: behind every launch that still reads the slot's old picture, in front of every later one */
	if(e->copy_out_ev) HIPCHK(hipStreamWaitEvent(e->stream, e->copy_out_ev, 0));      /* (another engine may still be reading this pool: hvk_frame_copy) */
	HIPCHK(hipMemcpyAsync(e->d_pool + slot * frame_px, stage, (size_t) w * h * 4, hipMemcpyHostToDevice, e->stream));
	HIPCHK(hipEventRecord(e->up_ev[ub], e->stream));
	e->up_busy[ub] = 1;
	return(HVK_OK);
}

/* The same for a picture that lies in page-locked memory (hvk_host_alloc()), rows `width` pixels apart: no copy on the
 * host, the cropped picture goes from where it lies to the slot by one strided DMA. */
extern "C" int hvk_frame_upload_pinned(hvk_engine_t *e, int slot, const uint32_t *fb, int width, int height, int interlaced)
{
	if(!e || slot < 0 || slot >= e->frame_slots) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	/* SECAM keeps a host copy of every picture for its fall-back chain, and an empty frame has nothing to copy:
	 * the ordinary way */
	if(e->secam || fb == NULL) return(hvk_frame_upload(e, slot, fb, width, height, 1, width, interlaced));

	const hvk_kconst_t &k = e->t.k;
	hvk_slot_t *s = &e->slots[slot];
	int x = (width - k.active_width) / 2, y = (height - k.active_lines) / 2;
	int w = k.active_width, h = k.active_lines;
	if(x < 0) { w += x; x = 0; }
	if(y < 0) { h += y; y = 0; }
	if(x + w > width) w = width - x;
	if(y + h > height) h = height - y;
	if(w < 0) w = 0;
	if(h < 0) h = 0;

	s->valid = w > 0 && h > 0;
	s->width = w;
	s->height = h;
	s->interlaced = interlaced;
	s->plane_dirty = 1;
	s->shown = 0;
	s->cells_valid[0] = s->cells_valid[1] = 0;
	memset(s->seeds_valid, 0, sizeof(s->seeds_valid));
	memset(s->memo_valid, 0, sizeof(s->memo_valid));
	s->gen++;
	if(!s->valid) return(HVK_OK);

	const uint32_t *src = fb + (size_t) y * width + x;
	s->many_colours = _many_colours(src, width, w, h);

	HIPCHK(hipSetDevice(e->device));
	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	if(e->copy_out_ev) HIPCHK(hipStreamWaitEvent(e->stream, e->copy_out_ev, 0));
	HIPCHK(hipMemcpy2DAsync(e->d_pool + slot * frame_px, (size_t) w * 4, src, (size_t) width * 4, (size_t) w * 4, (size_t) h,
	                        hipMemcpyHostToDevice, e->stream));
	return(HVK_OK);
}

/* The picture of another engine's slot (same configuration; any device) into a slot of this one, device to device:
 * what a group does on 525 lines with the picture a block's last frame shows, which the next block's engine needs in
 * front of its first frame. Ordered behind everything queued on the source engine's stream so far; the source engine's
 * next work waits for the copy. */
extern "C" int hvk_frame_copy(hvk_engine_t *e, int slot, hvk_engine_t *from, int from_slot)
{
	if(!e || !from || slot < 0 || slot >= e->frame_slots || from_slot < 0 || from_slot >= from->frame_slots) return(HVK_ERROR);
	if(e->device < 0 || from->device < 0) return(HVK_NO_DEVICE);
	const hvk_kconst_t &k = e->t.k;
	if(k.active_width != from->t.k.active_width || k.active_lines != from->t.k.active_lines || e->secam || from->secam) return(HVK_UNSUPPORTED);
	if(e == from && slot == from_slot) return(HVK_OK);

	hvk_slot_t *s = &e->slots[slot];
	const hvk_slot_t *f = &from->slots[from_slot];
	s->valid = f->valid; s->width = f->width; s->height = f->height; s->interlaced = f->interlaced;
	s->par_num = f->par_num; s->par_den = f->par_den; s->many_colours = f->many_colours;
	s->plane_dirty = 1;
	s->shown = 0;
	s->cells_valid[0] = s->cells_valid[1] = 0;
	memset(s->seeds_valid, 0, sizeof(s->seeds_valid));
	memset(s->memo_valid, 0, sizeof(s->memo_valid));
	s->gen++;
	if(!f->valid) return(HVK_OK);

	/* The copy runs on a stream of the destination's own: behind what the source has queued (the picture's upload: a) and
	 * behind what the destination has queued (a render that still reads the slot: c), in front of what the destination
	 * queues from now on (b). The SOURCE's stream waits for nothing -- a wait for b would put it behind the destination's
	 * renders, block b's launch behind block b + 1 - N's: with two engines one after the other; only an upload INTO the
	 * source's pool waits for the last copy out of it (copy_out_ev, hvk_frame_upload*). */
	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	hipEvent_t a = NULL, b = NULL, c = NULL;
	int ok = 0;
	do
	{
		if(hipSetDevice(e->device) != hipSuccess) break;
		if(!e->copy_stream && hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess) break;
		if(hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c, hipEventDisableTiming) != hipSuccess) break;
		if(hipEventRecord(c, e->stream) != hipSuccess) break;
		if(hipSetDevice(from->device) != hipSuccess) break;
		if(hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventRecord(a, from->stream) != hipSuccess) break;
		if(hipSetDevice(e->device) != hipSuccess) break;
		if(hipStreamWaitEvent(e->copy_stream, a, 0) != hipSuccess || hipStreamWaitEvent(e->copy_stream, c, 0) != hipSuccess) break;
		if(hipMemcpyPeerAsync(e->d_pool + slot * frame_px, e->device, from->d_pool + from_slot * frame_px, from->device,
		                      (size_t) f->width * f->height * 4, e->copy_stream) != hipSuccess) break;
		if(hipEventRecord(b, e->copy_stream) != hipSuccess || hipStreamWaitEvent(e->stream, b, 0) != hipSuccess) break;
		ok = 1;
	} while(0);
	if(a) (void) hipEventDestroy(a);      /* (released once the work queued on them is through) */
	if(c) (void) hipEventDestroy(c);
	if(!ok) { if(b) (void) hipEventDestroy(b); return(HVK_ERROR); }
	if(from->copy_out_ev) (void) hipEventDestroy(from->copy_out_ev);
	from->copy_out_ev = b;
	return(HVK_OK);
}

extern "C" int hvk_set_levels(hvk_engine_t *e, int mode)
{
	if(!e || mode < HVK_LEVELS_AUTO || mode > HVK_LEVELS_COMPUTE) return(HVK_ERROR);
	e->levels_mode = mode;
	for(int i = 0; i < e->frame_slots; i++)     /* (identical either way; made again all the same, so that a test of the mode tests it) */
	{
		e->slots[i].plane_dirty = 1;
		e->slots[i].cells_valid[0] = e->slots[i].cells_valid[1] = 0;
		memset(e->slots[i].seeds_valid, 0, sizeof(e->slots[i].seeds_valid));
		memset(e->slots[i].memo_valid, 0, sizeof(e->slots[i].memo_valid));
	}
	return(HVK_OK);
}

extern "C" int hvk_frame_aspect(hvk_engine_t *e, int slot, int64_t par_num, int64_t par_den)
{
	if(!e || slot < 0 || slot >= e->frame_slots || par_num <= 0 || par_den <= 0) return(HVK_ERROR);
	e->slots[slot].par_num = par_num;
	e->slots[slot].par_den = par_den;
	return(HVK_OK);
}

extern "C" int hvk_teletext_packets(hvk_engine_t *e, int frame_in_batch, const uint8_t *packets, uint32_t mask)
{
	if(!e || !packets || frame_in_batch < 0 || frame_in_batch >= e->max_frames) return(HVK_ERROR);
	if(!e->t.k.teletext) return(HVK_UNSUPPORTED);
	if(e->device < 0) return(HVK_NO_DEVICE);

	/* (queued in host memory that only the next stage's op lists are built from -- _build_vbi_ops --: nothing of the device's
	 * is touched and nothing waited for) */
	uint32_t *dst = e->h_tt_pk + (size_t) frame_in_batch * 32 * 12;
	for(int r = 0; r < 32; r++)
	{
		uint8_t row[48] = { 0 };
		memcpy(row, packets + r * 45, 45);
		memcpy(dst + r * 12, row, 48);     /* little endian: bit b of the packet is bit b & 31 of word b >> 5 */
	}
	e->h_tt_mask[frame_in_batch] = mask;
	return(HVK_OK);
}

/* ... for a run of frames in one call: packets [nframes][32][45], masks [nframes] */
extern "C" int hvk_teletext_packets_block(hvk_engine_t *e, int first_frame_in_batch, int nframes, const uint8_t *packets, const uint32_t *masks)
{
	if(!e || !packets || !masks || nframes < 0 || first_frame_in_batch < 0 || first_frame_in_batch + nframes > e->max_frames) return(HVK_ERROR);
	/* (everything a single frame's call can refuse, before any frame is touched: the block is queued whole or not at all) */
	if(!e->t.k.teletext) return(HVK_UNSUPPORTED);
	if(e->device < 0) return(HVK_NO_DEVICE);
	for(int i = 0; i < nframes; i++)
	{
		const int r = hvk_teletext_packets(e, first_frame_in_batch + i, packets + (size_t) i * 32 * 45, masks[i]);
		if(r != HVK_OK) return(r);
	}
	return(HVK_OK);
}

extern "C" int hvk_rawbb_write(hvk_engine_t *e, const int16_t *samples, size_t nsamples)
{
	if(!e) return(HVK_ERROR);
	if(!e->raw_q) return(HVK_UNSUPPORTED);
	if(nsamples && !samples) return(HVK_ERROR);
	e->raw_q->insert(e->raw_q->end(), samples, samples + nsamples);
	return(HVK_OK);
}

extern "C" int hvk_cc608_write(hvk_engine_t *e, int frame_in_batch, uint8_t c1, uint8_t c2)
{
	if(!e || frame_in_batch < 0 || frame_in_batch >= e->max_frames) return(HVK_ERROR);
	if(!e->cc_pairs) return(HVK_UNSUPPORTED);
	/* empty pairs are not sent (src/cc608.c:60-64) */
	if(((c1 | c2) & 0x7F) == 0) return(HVK_OK);
	e->cc_pairs[(size_t) frame_in_batch * 3 + 0] = 1;
	e->cc_pairs[(size_t) frame_in_batch * 3 + 1] = c1;
	e->cc_pairs[(size_t) frame_in_batch * 3 + 2] = c2;
	return(HVK_OK);
}

extern "C" int hvk_audio_write(hvk_engine_t *e, const int16_t *stereo, size_t nsamples)
{
	if(!e) return(HVK_ERROR);
	if(!e->audio || nsamples == 0) return(HVK_OK);
	return(hvk_audio_push(e->audio, stereo, nsamples));
}

/* The serial sound chains' state (hvk_audio.c): what an engine that renders the frames after this engine's last staged
 * one has to start from, and how much of the stream this engine's chains have worked through themselves. */
extern "C" size_t hvk_sound_state_size(const hvk_engine_t *e)
{
	return((e && e->audio) ? hvk_audio_state_bytes() : 0);
}

extern "C" int hvk_sound_state_export(hvk_engine_t *e, void *buf, size_t bytes)
{
	if(!e || !buf) return(HVK_ERROR);
	if(!e->audio) return(HVK_UNSUPPORTED);
	if(e->poisoned) return(HVK_ERROR);
	return(hvk_audio_state_export(e->audio, buf, bytes));
}

extern "C" int hvk_sound_state_import(hvk_engine_t *e, const void *buf, size_t bytes, int64_t *source_position)
{
	if(!e || !buf) return(HVK_ERROR);
	if(!e->audio) return(HVK_UNSUPPORTED);
	return(hvk_audio_state_import(e->audio, buf, bytes, source_position));
}

/* The colour chain's state between two frames (SECAM): what the last staged frame's last line left -- the pre-emphasis IIR's two
 * doubles and the values behind the line (hvk_secam_state_t; src/video.c:3095-3099, :3160-3165, :3202-3229) -- and the number
 * of the frame that comes next. 40 bytes; a group hands it from the engine of block b to the engine of block b + 1. */
typedef struct { hvk_secam_state_t st; int64_t next_frame; } hvk_secam_handover_t;

extern "C" size_t hvk_secam_state_size(const hvk_engine_t *e)
{
	return((e && e->secam) ? sizeof(hvk_secam_handover_t) : 0);
}

extern "C" int hvk_secam_state_export(hvk_engine_t *e, void *buf, size_t bytes)
{
	if(!e || !buf) return(HVK_ERROR);
	if(!e->secam) return(HVK_UNSUPPORTED);
	if(e->poisoned || bytes < sizeof(hvk_secam_handover_t)) return(HVK_ERROR);
	hvk_secam_handover_t h;
	memset(&h, 0, sizeof(h));
	if(e->secam_dev)
	{
		/* (the stage's last copy brings the carried state to the host: behind it) */
		HIPCHK(hipSetDevice(e->device));
		if(e->secam_pending) { const int r_ = hvk_e_secam_resolve(e); if(r_ < 0) return(r_); }
		HIPCHK(hipStreamSynchronize(e->stream));
		h.st = *e->h_secam_carry;
		h.next_frame = e->secam_next;
	}
	else hvk_secam_get_state(e->secam, &h.st, &h.next_frame);
	memcpy(buf, &h, sizeof(h));
	return(HVK_OK);
}

extern "C" int hvk_secam_state_import(hvk_engine_t *e, const void *buf, size_t bytes)
{
	if(!e || !buf) return(HVK_ERROR);
	if(!e->secam) return(HVK_UNSUPPORTED);
	if(e->poisoned || bytes < sizeof(hvk_secam_handover_t)) return(HVK_ERROR);
	hvk_secam_handover_t h;
	memcpy(&h, buf, sizeof(h));
	if(h.next_frame < 0) return(HVK_ERROR);
	e->secam_next = h.next_frame;
	hvk_secam_set_state(e->secam, &h.st, h.next_frame);         /* (the host's chain: the fall-back goes on from here too) */
	if(e->secam_dev)
	{
		HIPCHK(hipSetDevice(e->device));
		HIPCHK(hipStreamSynchronize(e->stream));                /* (a copy from this pinned word may still be on its way) */
		*e->h_secam_carry = h.st;
		HIPCHK(hipMemcpyAsync(e->sa.carry, e->h_secam_carry, sizeof(hvk_secam_state_t), hipMemcpyHostToDevice, e->stream));
		/* the frame before the block's first was another engine's: its picture counts as new here (warm-up lines, no kept
		 * state taken on trust -- the check decides as ever) */
		e->secam_last_new = 1;
		e->secam_last_frame = -1;       /* (... nor a kept sub-carrier set: behind which picture its first frame stands is not known here) */
	}
	return(HVK_OK);
}

extern "C" int64_t hvk_sound_samples_generated(const hvk_engine_t *e)
{
	return((e && e->audio) ? hvk_audio_generated(e->audio) : 0);
}

extern "C" int64_t hvk_sound_source_end(const hvk_engine_t *e)
{
	return((e && e->audio) ? hvk_audio_source_end(e->audio) : 0);
}

extern "C" size_t hvk_audio_needed(const hvk_engine_t *e, int nframes)
{
	if(!e || !e->audio) return(0);
	const hvk_kconst_t &k = e->t.k;
	int64_t upto = _fstart(e, e->next_frame + nframes) + (int64_t) k.out_prime;
	return(hvk_audio_source_needed(e->audio, upto));
}

extern "C" int hvk_host_side_streams(hvk_engine_t *e, int64_t first, int64_t count,
                                     int16_t *carriers, uint8_t *symbols, int max_symbols, int64_t *k0)
{
	if(!e) return(HVK_ERROR);
	if(!e->audio) return(0);
	return(hvk_audio_generate(e->audio, first, count, carriers, symbols, max_symbols, k0));
}

/* sound-in-syncs, host half on its own: the bursts of stream lines [first_line, first_line + nlines), forward only */
extern "C" int hvk_host_sis_bursts(hvk_engine_t *e, int64_t first_line, int nlines, uint8_t *out)
{
	if(!e || !out || first_line < 0 || nlines < 0) return(HVK_ERROR);
	if(!e->t.k.sis || !e->audio) return(HVK_UNSUPPORTED);
	/* (where line first_line + nlines begins in the stream: behind the resampler emitted line j is chunk j + s, hvk_tables_frame_start()) */
	int64_t upto = (first_line + nlines) * (int64_t) e->t.k.width;
	if(e->t.k.rs_L)
	{
		const hvk_kconst_t &k = e->t.k;
		const int64_t s = 1 + (k.vf_type ? k.delay_lines : 0), g = first_line + nlines + s;
		upto = (g * k.width * k.rs_L + k.rs_D - 1) / k.rs_D - (s * k.width * k.rs_L + k.rs_D - 1) / k.rs_D;
	}
	int r = hvk_audio_advance(e->audio, upto);
	if(r != HVK_OK) return(r);
	return(hvk_audio_sis_fetch(e->audio, first_line, nlines, out));
}

extern "C" int hvk_host_secam_stream(hvk_engine_t *e, const uint32_t *fb, int width, int height, int interlaced, int16_t *out)
{
	if(!e || !out) return(HVK_ERROR);
	if(!e->secam) return(HVK_UNSUPPORTED);

	/* centre crop to the active area, dense copy (as hvk_frame_upload) */
	const hvk_kconst_t &k = e->t.k;
	int x = (width - k.active_width) / 2, y = (height - k.active_lines) / 2;
	int w = k.active_width, h = k.active_lines;
	if(x < 0) { w += x; x = 0; }
	if(y < 0) { h += y; y = 0; }
	if(x + w > width) w = width - x;
	if(y + h > height) h = height - y;
	if(!fb || w <= 0 || h <= 0) { w = h = 0; fb = NULL; }

	std::vector<uint32_t> dense((size_t) w * h + 1);
	for(int r = 0; r < h; r++) memcpy(dense.data() + (size_t) r * w, fb + (size_t) (y + r) * width + x, (size_t) w * 4);

	int r = hvk_secam_frame(e->secam, e->secam_next, fb ? dense.data() : NULL, w, h, interlaced,
	                        fb ? dense.data() : NULL, w, h, interlaced, out);
	if(r == HVK_OK) e->secam_next++;
	return(r);
}

extern "C" int hvk_secam_stats(hvk_engine_t *e, int64_t counts[4])
{
	if(!e || !counts) return(HVK_ERROR);
	if(!e->secam) return(HVK_UNSUPPORTED);
	if(e->secam_pending && hvk_e_secam_resolve(e) < 0) return(HVK_ERROR);    /* (a block staged and not yet launched: its check's count is part of the figures) */
	hvk_secam_counters(e->secam, &counts[0], &counts[1], &counts[2]);
	counts[3] = 0;
	if(e->secam_dev) for(int i = 0; i < 4; i++) counts[i] = e->secam_counts[i];
	return(HVK_OK);
}

extern "C" int hvk_levels_short_form(const hvk_engine_t *e)
{
	return(e && e->device >= 0 ? e->t.yuv.fast : 0);
}

extern "C" int64_t hvk_secam_estimated_stages(const hvk_engine_t *e)
{
	return(e && e->secam_dev ? e->secam_est_stages : 0);
}

extern "C" int hvk_secam_kept(const hvk_engine_t *e, int64_t counts[3])
{
	if(!e || !counts) return(HVK_ERROR);
	if(e->secam_pending && hvk_e_secam_resolve(const_cast<hvk_engine_t *>(e)) < 0) return(HVK_ERROR);
	counts[0] = e->secam_dev ? e->secam_memo_frames : 0;
	counts[1] = e->secam_dev ? e->secam_memo_restarts : 0;
	counts[2] = e->secam_dev ? e->secam_memo_slots : 0;
	return(HVK_OK);
}

extern "C" int hvk_secam_walk_stages(const hvk_engine_t *e, int64_t counts[3])
{
	if(!e || !counts) return(HVK_ERROR);
	for(int i = 0; i < 3; i++) counts[i] = e->secam_dev ? e->secam_walk_stages[i] : 0;
	return(e->secam_dev ? e->secam_walk_ok : 0);
}

extern "C" int64_t hvk_frame_start(const hvk_engine_t *e, int64_t frame)
{
	if(!e || frame < 0) return(HVK_ERROR);
	return(_fstart(e, frame));
}

extern "C" int hvk_secam_warmup_lines(hvk_engine_t *e)
{
	if(!e) return(HVK_ERROR);
	if(!e->secam || !e->secam_dev) return(HVK_UNSUPPORTED);
	if(e->secam_pending && hvk_e_secam_resolve(e) < 0) return(HVK_ERROR);
	return(e->sa.K);
}
extern "C" int hvk_passthru_write(hvk_engine_t *e, const int16_t *iq, size_t nsamples)
{
	if(!e) return(HVK_ERROR);
	if(!e->t.k.has_passthru || !e->tail) return(HVK_UNSUPPORTED);
	return(hvk_tail_passthru_push(e->tail, iq, nsamples));
}

extern "C" int hvk_line_widths(const hvk_engine_t *e, int64_t first_line, int nlines, int32_t *widths)
{
	if(!e || !widths || first_line < 0 || nlines < 0) return(HVK_ERROR);
	hvk_tables_line_widths(&e->t, first_line, nlines, widths);
	return(HVK_OK);
}

extern "C" int hvk_host_offset_stream(hvk_engine_t *e, int64_t first, int64_t count, int16_t *out)
{
	if(!e || !out) return(HVK_ERROR);
	if(!e->t.k.has_offset || !e->tail) return(HVK_UNSUPPORTED);
	return(hvk_tail_offset_stream(e->tail, first, count, out));
}

extern "C" int hvk_host_fm_video(hvk_engine_t *e, int16_t *iq, int64_t count)
{
	if(!e || !iq) return(HVK_ERROR);
	if(!e->t.k.fm_video || !e->tail) return(HVK_UNSUPPORTED);
	return(hvk_tail_fm_apply(e->tail, hvk_tail_fm_position(e->tail), count, iq));
}

extern "C" void *hvk_output_device_ptr(hvk_engine_t *e) { return(e ? e->d_out : NULL); }
extern "C" void *hvk_engine_stream(hvk_engine_t *e) { return(e ? (void *) e->stream : NULL); }
extern "C" int64_t hvk_fused_launches(const hvk_engine_t *e) { return(e ? e->fused_count : 0); }

extern "C" int hvk_last_line_shows_picture(const hvk_engine_t *e)
{
	if(!e) return(0);
	const hvk_linedesc_t *dl = &e->t.desc[e->t.k.lines - 1];
	return(dl->ar > dl->al && !e->t.k.rawbb);
}

extern "C" int hvk_stream_is_one_chain(const hvk_engine_t *e)
{
	if(!e) return(0);
	const hvk_kconst_t &k = e->t.k;
	/* (sound-in-syncs: its burst encoder keeps the sound chains a line or more ahead of the requests, and a state that
	 * stands past the last request is not one hvk_sound_state_export() hands on) */
	/* (SECAM colour is a chain too, but what it hands from frame to frame is 40 bytes: hvk_secam_state_export / _import) */
	return(k.fm_video || k.rs_irr || k.has_passthru || k.rawbb || k.sis);
}


/* The kernels hvk_launch() enqueues for this configuration, as rocprofv3 prints them, ';' between
 * them: the launchers' choice of template arguments restated (hvk_kernels.hip, hvk_direct.hip). */
extern "C" int hvk_kernel_names(const hvk_engine_t *e, char *buf, int n)
{
	if(!e || !buf || n < 1) return(HVK_ERROR);
	const hvk_kconst_t &k = e->t.k;
	const int nt = k.secam ? 1 : (k.colour ? k.chroma_ntaps : 1);
	const int lv = e->levels_computed ? 1 : 0;
	if(e->direct)
	{
		const int exact = k.frame_samples % HVK_TILE == 0 ? 1 : 0, vf = k.vf_type ? 1 : 0, col = k.secam ? 2 : (k.colour ? 1 : 0);
		char dk[96];
		snprintf(dk, sizeof(dk), "hvk_k_direct<%d, %d, %d, %d, %d, %d>", vf, col, exact, e->ovr_n ? 1 : 0, (k.has_carriers && k.has_nicam && vf && !e->ovr_n) ? 1 : 0, e->d_tilerec ? 1 : 0);      /* (tile records with rows of the optional stages too since round 6) */
		if(e->ovr_n) snprintf(buf, n, "hvk_k_raster<%d, %d, 0, 1, 0, %d>;%s", nt, k.secam ? 1 : 0, lv, dk);
		else snprintf(buf, n, "%s", dk);
		return(HVK_OK);
	}
	const int sv = k.s_video ? 1 : 0;
	const int extras = (sv || k.vbi || k.vits || k.rawbb || k.sis || k.fsc_mode || (k.secam && e->t.conf.secam_field_id)) ? 1 : 0;
	const int wc = (!k.secam && !sv && !extras && nt == 13 && k.width == 1024) ? 1024 : 0;
	const int vnt = k.vf_type ? k.vf_ntaps : 1;
	const int exact = (!k.rs_irr && k.frame_samples % HVK_TILE == 0 && k.s_stride - k.s_lead - k.frame_samples >= 128) ? 1 : 0;
	const int mf = (vnt == 51 && e->d_mfma_a) ? 1 : 0;
	snprintf(buf, n, "hvk_k_raster<%d, %d, %d, %d, %d, %d>;%shvk_k_filter<%d, %d, %d, %d, %d>", nt, k.secam ? 1 : 0, sv, extras, wc, lv,
	         k.rs_L ? "hvk_k_resample;" : "", vnt, k.vf_type, sv, exact, mf);
	return(HVK_OK);
}

/* The same with everything beside the per-launch kernels: what runs once per uploaded picture, once per staged block, per
 * launch, behind it -- one line per stage, "when: kernels [(condition)]" (tools/kernel_table.py makes DESIGN.md's table of it) */
extern "C" int hvk_kernel_plan(const hvk_engine_t *e, char *buf, int n)
{
	if(!e || !buf || n < 1) return(HVK_ERROR);
	const hvk_kconst_t &k = e->t.k;
	char launch[512];
	int r = hvk_kernel_names(e, launch, (int) sizeof(launch));
	if(r != HVK_OK) return(r);
	for(char *p = launch; *p; p++) if(*p == ';') *p = '+';
	const int nt = k.secam ? 1 : (k.colour ? k.chroma_ntaps : 1);
	int o = 0;
	buf[0] = 0;
#define PLAN(...) do { if(o < n) o += snprintf(buf + o, (size_t) (n - o), __VA_ARGS__); } while(0)
	if(e->direct) PLAN("per uploaded picture: hvk_k_prep8<%d, %d, LV%s> (picture planes; LV 0 table levels, 1-3 computed)\n", nt, (nt == 13 && k.width == 1024) ? 1024 : 0, k.secam ? ", 1" : "");
	if(k.secam && e->secam_dev)
		PLAN("per staged block: hvk_k_secam_cells (a picture's cells once per slot and parity) + hvk_k_secam_est (new pictures' entry states) + hvk_k_secam_walk<%s> (one line per lane; hvk_k_secam_chain where warm-up lines are walked) + hvk_k_secam_check [+ hvk_k_secam_redo / _redo_fields] + hvk_k_secam_carry\n",
		     e->secam_walk_ok == 2 ? "0 | 1" : "0");
	else if(k.secam) PLAN("per staged block: the host's serial colour chain (hvk_secam.c)\n");
	PLAN("per launch: %s\n", launch);
	if(e->fused_ok) PLAN("per launch of a block of mostly NEW pictures: hvk_k_fused<%d, LV> instead (from the pixels, no planes)\n", nt);
	if(k.sv_ring) PLAN("per launch, between resampler and filter: hvk_k_svq (the Q channel line by line: the reference's ring of line buffers)\n");
	if(!k.fm_video && (k.swap_iq || k.has_offset || k.has_passthru)) PLAN("behind it: hvk_k_tail<%d, %d, %d>\n", k.swap_iq ? 1 : 0, k.has_offset ? 1 : 0, k.has_passthru ? 1 : 0);
	if(k.fm_video) PLAN("behind it: the FM video phasor on the host (hvk_tail.c; behind hvk_fetch_async() on the engine's thread)\n");
	if(k.has_carriers || k.has_nicam) PLAN("side inputs per staged block: the sound carriers' serial chain and the NICAM framing on the host (hvk_audio.c)\n");
#undef PLAN
	return(HVK_OK);
}

extern "C" int hvk_timing_enable(hvk_engine_t *e, int on)
{
	if(!e) return(HVK_ERROR);
	e->timing = on;
	e->ev_used = 0;
	e->t_sum[0] = e->t_sum[1] = 0;
	e->t_n[0] = e->t_n[1] = 0;
	return(HVK_OK);
}

extern "C" int hvk_timing_read(hvk_engine_t *e, int which, double *avg_ms, int64_t *launches)
{
	if(!e || which < 0 || which > 1) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipStreamSynchronize(e->stream));
	for(int i = 0; i < e->ev_used; i++)
	{
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, e->ev[i][0], e->ev[i][1]));
		e->t_sum[0] += ms; e->t_n[0]++;
		HIPCHK(hipEventElapsedTime(&ms, e->ev[i][1], e->ev[i][2]));
		e->t_sum[1] += ms; e->t_n[1]++;
	}
	e->ev_used = 0;
	if(avg_ms) *avg_ms = e->t_n[which] ? e->t_sum[which] / e->t_n[which] : 0;
	if(launches) *launches = e->t_n[which];
	return(HVK_OK);
}

