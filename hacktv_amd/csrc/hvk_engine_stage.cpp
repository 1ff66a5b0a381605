/* hvk_engine_stage.cpp -- staging a batch (hvk_stage_strided*): the host pre-passes and side inputs of its frames, the VBI
 * inserters' op lists, the SECAM colour stage on the device, the picture planes of new pictures. hvk_engine_priv.h has the
 * engine's state; src/video.c:4867-4952 is what a batch stands for. */
#include "hvk_engine_priv.h"


/* ---- render ---- */

/* The VBI data lines of the staged frames (h_fdesc holds their stream frame numbers): per
 * frame a list of ops -- which symbol table, how many bits, the bits -- and a line -> op map.
 * Ops of one line are chained in the reference's process order WSS, ACP, VITC, CC608, teletext
 * (src/video.c:4234-4358). Of the inserters only ACP and teletext yield to a line that is
 * already held (vbialloc, src/acp.c:108, src/teletext.c:1219): ACP's test is done here,
 * teletext's is the caller's business -- it decides which rows carry packets. */
static void _build_vbi_ops(hvk_engine *e, int nframes)
{
	const hvk_tables_t &t = e->t;
	const int lines = t.k.lines;

	memset(e->h_map, 0xFF, (size_t) nframes * lines);
	memset(e->h_ops, 0, (size_t) nframes * HVK_VBI_OPS * HVK_VBI_OPWORDS * 4);

	for(int i = 0; i < nframes; i++)
	{
		uint32_t *ops = e->h_ops + (size_t) i * HVK_VBI_OPS * HVK_VBI_OPWORDS;
		int8_t *map = e->h_map + (size_t) i * lines;
		int n = 0;

		/* hang op n on its line: the first op goes into the map, later ones behind the line's last op
		 * (op word 0: symbol base | (next op + 1) << 16) */
		auto link = [&](int line0)
		{
			if(map[line0] < 0) { map[line0] = (int8_t) n; return; }
			uint32_t *last = ops + (size_t) map[line0] * HVK_VBI_OPWORDS;
			while(last[0] >> 16) last = ops + (size_t) ((last[0] >> 16) - 1) * HVK_VBI_OPWORDS;
			last[0] |= (uint32_t) (n + 1) << 16;
		};

		auto add = [&](int line0, int lut, int first_symbol, int nbits, const uint8_t *lsb_first_bits, int blank_lo, int blank_hi)
		{
			if(n >= HVK_VBI_OPS || line0 < 0 || line0 >= lines) return;
			if(nbits > t.lut_nsym[lut] - first_symbol) nbits = t.lut_nsym[lut] - first_symbol;   /* the table's end stops the render */
			if(nbits > 384) nbits = 384;
			uint32_t *op = ops + (size_t) n * HVK_VBI_OPWORDS;
			uint8_t bytes[48] = { 0 };
			if(nbits < 0) nbits = 0;
			memcpy(bytes, lsb_first_bits, (nbits + 7) / 8);
			op[0] = (uint32_t) (t.lut_base[lut] + first_symbol);
			op[1] = (uint32_t) nbits;
			op[2] = (uint32_t) blank_lo | ((uint32_t) blank_hi << 16);
			/* (the table, the first symbol the data's bit 0 stands for, and whether the table has cover lists: the gather) */
			op[3] = (uint32_t) lut | ((uint32_t) first_symbol << 8) | ((e->d_vbi_cov && e->vbi_cov_ok[lut] && first_symbol < 0x7FFF) ? 0x80000000u : 0u);
			memcpy(op + 4, bytes, 48);
			link(line0);
			n++;
		};

		if(t.conf.wss)
		{
			/* line 23; the table's bits are MSB first (src/wss.c:184) */
			uint8_t rev[18], bits[18];
			const hvk_slot_t &sl = e->slots[e->staged_slots[i]];
			hvk_wss_bits(&t, sl.par_den ? sl.par_num : 1, sl.par_den ? sl.par_den : 1, bits);
			for(int b = 0; b < 18; b++)
			{
				uint8_t v = bits[b], r = 0;
				for(int q = 0; q < 8; q++) if(v & (1 << q)) r |= 0x80 >> q;
				rev[b] = r;
			}
			add(22, 1, 0, 137, rev, t.wss_blank_lo, t.wss_blank_hi > t.wss_blank_lo ? t.wss_blank_hi : t.wss_blank_lo);
		}

		if(t.conf.acp)
		{
			/* six P-sync / AGC pulse pairs on ten lines per field (eight on 525 lines), except where the line
			 * is held already (src/acp.c:93-108): by VITS, or by SECAM's colour process, which marks its field
			 * identification lines (src/video.c:3101-3103, :3135); the AGC level moves with the frame number */
			const int frame = (int) (e->h_fdesc[(size_t) i * (t.k.fields + 1) + 1].frame_index + 1);
			const int agc = hvk_acp_agc_level(&t, frame);
			const int first[2] = { lines == 625 ? 9 : 12, lines == 625 ? 321 : 275 };
			const int count = lines == 625 ? 10 : 8;
			for(int fld = 0; fld < 2; fld++)
			{
				for(int l = first[fld]; l < first[fld] + count; l++)
				{
					bool vits = false;
					for(int q = 0; q < t.k.vits; q++) if(t.k.vits_line[q] == l - 1) vits = true;
					if(vits || (!t.conf.raw_bb && (t.desc[l - 1].secam_fid & 1)) || n >= HVK_VBI_OPS) continue;
					uint32_t *op = ops + (size_t) n * HVK_VBI_OPWORDS;
					op[0] = 0;
					op[1] = 1u << 16;       /* mode 1: assign list */
					op[2] = 0;
					op[3] = ((uint32_t) t.acp_psync_level & 0xFFFF) | ((uint32_t) agc << 16);
					for(int q = 0; q < 6; q++)
					{
						const uint32_t a = t.acp_left[q], b = a + t.acp_psync_width, c = b + t.acp_pagc_width;
						op[4 + q * 2 + 0] = a | (b << 16);
						op[4 + q * 2 + 1] = b | (c << 16);
					}
					link(l - 1);
					n++;
				}
			}
		}

		if(t.conf.vitc)
		{
			const int frame = (int) (e->h_fdesc[(size_t) i * (t.k.fields + 1) + 1].frame_index + 1);
			const int vl[4] = { t.vitc_lines[0], t.vitc_lines[0] + 2, t.vitc_lines[1], t.vitc_lines[1] + 2 };
			for(int q = 0; q < 4; q++)
			{
				uint8_t data[12];
				const int nb = hvk_vitc_bits(&t, frame, vl[q], data);
				add(vl[q] - 1, 2, 21, nb, data, 0, 0);      /* src/vitc.c:193: the first 21 symbols stay empty */
			}
		}

		if(t.conf.cc608)
		{
			/* the frame's byte pair (zeros without one), 17 bits, and the clock run-in: symbol 32 of
			 * the table, whose bit is always set (src/cc608.c:188-221) */
			uint8_t bits[8] = { 0 };
			const uint8_t *pr = e->cc_pairs + (size_t) i * 3;
			hvk_cc608_bits(pr[0] ? pr[1] : 0, pr[0] ? pr[2] : 0, bits);
			bits[2] &= 1;
			bits[4] |= 1;           /* bit 32 */
			add(t.cc608_line - 1, 3, 0, 33, bits, 0, 0);
		}

		if(t.k.teletext && e->h_tt_mask[i])
		{
			for(int r = 0; r < 32; r++)
			{
				if(!((e->h_tt_mask[i] >> r) & 1)) continue;
				add(r < 16 ? 6 + r : 319 + r - 16, 0, 0, 360, (const uint8_t *) (e->h_tt_pk + ((size_t) i * 32 + r) * 12), 0, 0);
			}
		}
	}
}

/* Which lines of a frame the inserters other than teletext write to -- the lines on which the reference's
 * vid_line_t.vbialloc is set by the time the teletext process sees them (src/teletext.c:1219; the processes run in the
 * order VITS, WSS, ACP, VITC, CC608, ..., teletext, src/video.c:4234-4358). From the same tables the op list above is
 * built from, so that a caller who schedules teletext packets (the shim) does not keep a list of its own. */
extern "C" int hvk_vbi_lines_held(const hvk_engine_t *e, uint8_t *held, int nlines)
{
	if(!e || !held || nlines < e->t.k.lines) return(HVK_ERROR);
	const hvk_tables_t &t = e->t;
	const int lines = t.k.lines;
	memset(held, 0, (size_t) nlines);
	auto hold = [&](int line1) { if(line1 >= 1 && line1 <= lines) held[line1 - 1] = 1; };

	for(int q = 0; q < t.k.vits; q++) hold(t.k.vits_line[q] + 1);
	if(t.conf.wss) hold(23);
	if(t.conf.acp)
	{
		const int first[2] = { lines == 625 ? 9 : 12, lines == 625 ? 321 : 275 };
		const int count = lines == 625 ? 10 : 8;
		for(int fld = 0; fld < 2; fld++) for(int l = first[fld]; l < first[fld] + count; l++) hold(l);
	}
	if(t.conf.vitc)
	{
		hold(t.vitc_lines[0]); hold(t.vitc_lines[0] + 2);
		hold(t.vitc_lines[1]); hold(t.vitc_lines[1] + 2);
	}
	if(t.conf.cc608) hold(t.cc608_line);
	/* SECAM field identification lines carry the sub-carrier ramp (src/video.c:3101-3103, :4132-4137); raw baseband has
	 * no colour process to mark them (src/video.c:4180-4190) */
	if(!t.conf.raw_bb) for(int l = 1; l <= lines; l++) if(t.desc[l - 1].secam_fid & 1) hold(l);
	return(HVK_OK);
}

extern "C" int hvk_stage_strided(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots)
{
	return(hvk_e_stage(e, first_frame, stride, nframes, slots, NULL));
}

extern "C" int hvk_stage_strided_prev(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots)
{
	return(hvk_e_stage(e, first_frame, stride, nframes, slots, prev_slots));
}

/* SECAM: the sub-carrier of the staged frames on the device (hvk_secam.hip) -- every line at once from derived entry
 * states, then check / redo rounds until every line started from the state the line before it left. The frame
 * descriptors and pictures are on their way to the device (same stream). */

static int _secam_on_device(hvk_engine_t *e, int64_t first_frame, int nframes);
static int _secam_rounds(hvk_engine_t *e, int64_t first_frame, int nframes, int memo_used, int first_bad);

/* ---- The kept sub-carrier -------------------------------------------------------------------------------------------
 * What the colour chain makes of a frame is a function of three things: the picture's cells (kept per slot and parity already),
 * the frame's number modulo 6 (which colour-difference signal a line carries: modulo 2; the sub-carrier's start phase,
 * (frame * lines + line) mod 3, src/video.c:3211-3212), and the state the frame's first line starts from. A picture that stays
 * -- the test card of BASELINE config 4, a paused source -- meets all three again and again, so the rows of a walk whose every
 * line passed the check are KEPT, six sets per picture slot, with every line's entry state (the seeds, which the walks keep
 * anyway) and the state behind the frame's last line. A later frame of that picture and number takes the set instead of being
 * walked -- hvk_k_secam_walk hands the kept states to the check, which compares the frame's start with the exit of the frame
 * before exactly as it does for every line; the render reads the kept rows (hvk_framedesc_t.chroma_row). The per-picture share
 * of SECAM's work, as the picture planes are PAL's: by induction from the carried state a frame that passes is the walk's, bit
 * for bit. A frame that fails sends the whole block through the chain again without kept sets (_secam_on_device).
 *
 * A set is made by ONE frame per block -- the last that shows the picture with that number: its `owner`, the only one whose walk
 * writes the set's rows, seeds and exit state -- and becomes valid when that block's first check finds nothing wrong. It stops
 * being valid when its slot gets another picture, when a block it took part in had a wrong start, when the seeds are written by
 * anything but an owner's hvk_k_secam_walk (the chain kernel's warm-up walks, the host's chain). */
static void _secam_kept_drop(hvk_engine_t *e, int nframes)
{
	for(int i = 0; i < nframes; i++) memset(e->slots[e->staged_slots[i]].memo_valid, 0, sizeof(e->slots[0].memo_valid));
}

/* decides per frame: takes a set (mflag), makes one (owner), or neither; returns the number of frames that take one */
static int _secam_kept_plan(hvk_engine_t *e, int64_t first_frame, int nframes, int walk)
{
	hvk_secam_args_t &a = e->sa;
	const hvk_kconst_t &k = e->t.k;
	int *rows = e->h_secam_rows, *mflag = rows + 4 * e->max_frames, *owner = rows + 5 * e->max_frames, *orow = rows + 6 * e->max_frames, *mrow = rows + 7 * e->max_frames;
	const int fields = k.fields;
	int taken = 0, any = 0;
	int64_t *pkey = e->secam_prev_key;
	const bool on = e->secam_memo_slots > 0 && !e->secam_memo_off && walk > 0 && e->secam_cell_cache && e->secam_seeds && a.seed && a.seedx && fields == 1;

	for(int i = 0; i < nframes; i++) { mflag[i] = owner[i] = 0; orow[i] = i; mrow[i] = 0; }
	if(!on)
	{
		/* (whatever walks these frames now writes their seeds: the sets they belong to are no longer one walk's) */
		if(e->secam_memo_slots > 0) _secam_kept_drop(e, nframes);
	}
	else
	{
		std::vector<uint8_t> made((size_t) 6 * e->secam_memo_slots, 0);
		/* What a frame starts from is what the frame before it leaves behind, and that is a matter of that frame's picture and number
		 * alone (a line's start state is forgotten within some thirty lines): a set is taken where the frame stands behind the picture
		 * its set's frame stood behind -- a picture that follows ANOTHER one is walked, one frame, instead of being taken on trust,
		 * failing the check and sending the block through the chain again (round 6's fuzzer of staying pictures: one restart in five
		 * blocks before). The check decides as ever: the rule only chooses what is worth trying. */
		const bool keep_any = getenv("HVK_SECAM_KEEP_ANY") != NULL;     /* (the round's first form: a set tried behind any picture -- the check's catch and the restart stay testable) */
		for(int i = 0; i < nframes; i++)
			pkey[i] = i > 0 ? hvk_slot_key(e->slots, e->staged_slots[i - 1]) : (e->secam_last_frame >= 0 && e->secam_last_frame + 1 == first_frame ? e->secam_last_key : -1);
		for(int i = nframes - 1; i >= 0; i--)
		{
			const int slot = e->staged_slots[i];
			const int ph6 = (int) ((first_frame + i + 1) % 6);
			if(slot >= e->secam_memo_slots || first_frame + i == 0) continue;       /* (the stream's first frame has the two fill slots) */
			const int set = slot * 6 + ph6;
			mrow[i] = set;
			if(e->slots[slot].memo_valid[ph6] && (pkey[i] < 0 || e->slots[slot].memo_prev[ph6] != pkey[i]) && !keep_any)
			{
				/* (kept, but behind another picture: walked into the batch's own row, the set stays as it is. The states kept for its
				 * lines are the set's: the first line's would be wrong for certain, so the frame's are estimated as a new picture's are) */
				mrow[i] = 0;
				if(e->secam_est) rows[2 * e->max_frames + i] = -1;
			}
			else if(e->slots[slot].memo_valid[ph6])
			{
				mflag[i] = 1;
				orow[i] = e->max_frames + set;
				rows[2 * e->max_frames + i] = 0;        /* (no entry state to estimate either) */
				taken++;
			}
			else if(!made[(size_t) set])
			{
				/* the last frame of the block with this picture and number makes the set */
				made[(size_t) set] = 1;
				owner[i] = 1;
				orow[i] = e->max_frames + set;
			}
		}
		for(int i = 0; i < nframes; i++) any |= mflag[i];
	}
	a.mflag = any ? a.cbase + 4 * e->max_frames : NULL;
	a.owner = on ? a.cbase + 5 * e->max_frames : NULL;      /* (sets in play: nobody but an owner writes the seeds; none: every walk does, as ever) */
	for(int i = 0; i < nframes; i++) for(int f_ = 0; f_ <= fields; f_++) e->h_fdesc[(size_t) i * (fields + 1) + f_].chroma_row = orow[i];
	/* (the halo descriptor of a frame names the frame before's picture, never its sub-carrier: the lines around a frame carry none) */
	HIPCHK_P(hipMemcpyAsync(e->d_secam[10], rows, (size_t) e->max_frames * 8 * sizeof(int), hipMemcpyHostToDevice, e->stream));
	HIPCHK_P(hipMemcpyAsync(e->d_fdesc, e->h_fdesc, sizeof(hvk_framedesc_t) * nframes * (fields + 1), hipMemcpyHostToDevice, e->stream));
	e->secam_memo_frames += taken;
	return(taken);
}

/* the block's first check found nothing wrong: the sets its owners made are valid */
static void _secam_kept_commit(hvk_engine_t *e, int64_t first_frame, int nframes)
{
	const int *owner = e->h_secam_rows + 5 * e->max_frames;
	for(int i = 0; i < nframes; i++) if(owner[i])
	{
		hvk_slot_t &sl = e->slots[e->staged_slots[i]];
		sl.memo_valid[(first_frame + i + 1) % 6] = 1;
		sl.memo_prev[(first_frame + i + 1) % 6] = e->secam_prev_key[i];
	}
}

static int _secam_on_device(hvk_engine_t *e, int64_t first_frame, int nframes)
{
	const hvk_kconst_t &k = e->t.k;
	hvk_secam_args_t &a = e->sa;
	int r;

	a.nframes = nframes;
	a.total = nframes * a.ntasks;
	a.first_frame = first_frame;
	a.fdesc = e->d_fdesc;
	a.levels_computed = e->levels_computed;
	a.yuvp = e->d_yuvparams;
	a.uvp = NULL;
	if(e->direct && e->d_UVp)
	{
		/* the staged pictures' planes now, not at the launch: the cells are made of them */
		if((r = hvk_e_prep_dirty(e, e->staged_slots, nframes, e->stream)) < 0) return(r);
		a.uvp = e->d_UVp + 16;
	}
	/* A lane's walk is a chain of dependent operations: a SIMD interleaves a few waves of it for free (measured: 1156
	 * waves on 1024 SIMDs take as long as 578). Longer runs per lane only when the batch has more lines than eight
	 * waves per SIMD hold. */
	a.R = (a.total + e->secam_lanes - 1) / e->secam_lanes;
	if(a.R < 1) a.R = 1;
	if(getenv("HVK_SECAM_RUN")) a.R = atoi(getenv("HVK_SECAM_RUN")) > 0 ? atoi(getenv("HVK_SECAM_RUN")) : 1;
	a.nruns = (a.total + a.R - 1) / a.R;

	/* Which rows of the cell stores the frames read, and which of them are made now: a picture's cells depend on the
	 * picture and on the parity of the frame's number only (which of the two colour-difference signals a line carries,
	 * which picture rows a field shows), so a picture that stays has them made once per parity -- the per-picture
	 * share of SECAM's work, as the picture planes are PAL's and NTSC's. (The list's last copy is through: every stage
	 * ends with the check's count read back.) */
	{
		int *rows = e->h_secam_rows, *list = rows + e->max_frames, *kf = rows + 2 * e->max_frames, *srows = rows + 3 * e->max_frames;
		a.ncells = 0;
		for(int i = 0; i < nframes; i++)
		{
			kf[i] = e->secam_est ? -1 : HVK_SECAM_WARMUP;
			srows[i] = 0;
			if(!e->secam_cell_cache)
			{
				rows[i] = i * a.ntasks;
				list[a.ncells++] = i;
				continue;
			}
			const int slot = e->staged_slots[i];
			const int parity = (int) ((first_frame + i + 1) & 1);
			int fresh = 0;
			rows[i] = (slot * 2 + parity) * a.ntasks;
			if(!e->slots[slot].cells_valid[parity] || first_frame + i == 0)      /* (the stream's first frame has the two fill slots) */
			{
				list[a.ncells++] = i;
				e->slots[slot].cells_valid[parity] = 1;
				fresh = 1;
			}
			/* The states kept per row are those the picture's lines had the last time it was shown: good for a picture that
			 * was here before, behind a frame that was here before. A new picture, and the frame behind one (its first
			 * lines' warm-ups start in it), take the full number of warm-up lines; the others the number that follows how
			 * the batches have gone (a.K) */
			/* (what a line starts from also follows its sub-carrier's start phase, (frame * lines + line) mod 3: with the
			 * parity, the frame's number modulo 6) */
			const int ph6 = (int) ((first_frame + i + 1) % 6);
			srows[i] = (slot * 6 + ph6) * a.ntasks;
			if(!e->slots[slot].seeds_valid[ph6]) fresh = 1;
			e->slots[slot].seeds_valid[ph6] = 1;
			/* (while that number is still three or more the estimate is the cheaper start: a fifth of a walk instead of
			 * three and more; the kept states take over below that) */
			if(!fresh && !e->secam_last_new) kf[i] = (e->secam_est && a.K >= 3) ? -1 : a.K;
			e->secam_last_new = fresh;
		}
		/* (the lists go to the device with the kept sets' -- _secam_kept_plan, below, in front of the first kernel that reads them) */
	}

	int memo_used = 0;
	{
		int want = a.kf == NULL;
		for(int i = 0; i < nframes && !want; i++) want = e->h_secam_rows[2 * e->max_frames + i] < 0;
		/* One line per lane and no warm-up line anywhere in the block (entry states estimated, or kept from the picture's
		 * last showing): hvk_k_secam_walk. Its FM steps computed and its gains from LDS where the block shows pictures of many
		 * colours -- their table reads would scatter over a cache line per sample --, both from the table otherwise (the
		 * lines of a wave then read neighbouring entries) */
		int walk = e->secam_walk_ok && a.R == 1;
		if(a.kf == NULL) walk = walk && (a.est != NULL || a.K == 0);
		else for(int i = 0; i < nframes && walk; i++) walk = e->h_secam_rows[2 * e->max_frames + i] <= 0;
		if(walk)
		{
			int many = 0;
			for(int i = 0; i < nframes && !many; i++) many = e->slots[e->staged_slots[i]].many_colours;
			walk = (many && e->secam_walk_ok == 2) ? 2 : 1;
			if(e->secam_walk_mode >= 0) walk = e->secam_walk_mode > e->secam_walk_ok ? e->secam_walk_ok : e->secam_walk_mode;
		}
		e->secam_walk_stages[walk]++;
		memo_used = _secam_kept_plan(e, first_frame, nframes, walk);
		if(memo_used < 0) return(memo_used);
		if(memo_used > 0 && a.kf != NULL)
		{
			want = 0;
			for(int i = 0; i < nframes && !want; i++) want = e->h_secam_rows[2 * e->max_frames + i] < 0;
		}
		/* The chain writes every line on its task list whole and never another; the list follows the frame's parity. A slab
		 * row that was last written with the same parity has nothing to clear (blocks of even length, one after the other:
		 * none of them), the others are cleared in runs. (A frame that takes or makes a kept set leaves its row of the batch
		 * alone; a set's rows are always written with the one parity of its number.) */
		{
			const int *orow = e->h_secam_rows + 6 * e->max_frames;
			auto stale = [&](int j) { return(orow[j] == j && e->chroma_par[j] != (signed char) ((first_frame + j + 1) & 1)); };
			for(int i = 0; i < nframes; )
			{
				int j = i;
				while(j < nframes && stale(j)) j++;
				if(j > i) HIPCHK(hipMemsetAsync(e->d_chroma + (size_t) i * k.raster_samples, 0, (size_t) (j - i) * k.raster_samples * 2, e->stream));
				for(int q = i; q < j; q++) e->chroma_par[q] = (signed char) ((first_frame + q + 1) & 1);
				i = j + 1;
			}
		}
		if((r = hvk_launch_secam_cells_chain(&a, e->secam_est && want, walk, e->stream)) != HVK_OK) return(r);
		if(e->secam_est && want) e->secam_est_stages++;
		e->secam_est_ran = e->secam_est && want;
	}
	e->secam_counts[0] += (int64_t) (nframes - (memo_used > 0 ? memo_used : 0)) * a.ntasks;     /* (lines walked: not those of frames that took a kept set) */

	/* The check's count is not waited for here where the last blocks' checks found nothing: it is queued, hvk_launch queues the
	 * render behind it and only then reads the count (hvk_e_secam_resolve) -- the GPU goes from the check straight into the
	 * render instead of standing idle for the trip to the host and the launches that follow it (40 of config 4's 200 us a block);
	 * the rare block whose check fails is repaired as ever and rendered again. (HVK_SECAM_DEFER=0: the count awaited here.) */
	if(e->secam_defer && !e->secam_resolving && e->secam_memo_off == 0 && e->secam_spec_left == 0 && !k.fm_video
	   && !getenv("HVK_SECAM_DEBUG") && !getenv("HVK_SECAM_FORCE_FALLBACK"))
	{
		if((r = hvk_launch_secam_check(&a, e->stream)) != HVK_OK) return(r);
		HIPCHK(hipMemcpyAsync(e->h_secam_count, a.count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipEventRecord(e->secam_ev, e->stream));
		e->secam_pending = 1;
		e->secam_p_first = first_frame;
		e->secam_p_n = nframes;
		e->secam_p_memo = memo_used;
		return(HVK_OK);
	}
	return(_secam_rounds(e, first_frame, nframes, memo_used, -1));
}

/* The first check's count is in (first_bad >= 0), or the check is still to be made (-1): the rounds of repair, what the stage
 * learns from the count, the kept sets made valid or dropped, the state carried on */
static int _secam_rounds(hvk_engine_t *e, int64_t first_frame, int nframes, int memo_used, int first_bad)
{
	const hvk_kconst_t &k = e->t.k;
	hvk_secam_args_t &a = e->sa;
	int r, rounds = 0;

	/* Where recent blocks had lines that started wrong, the first check is followed at once by the redo round and ITS check (a redo
	 * without failed runs returns at once): one wait for both counts instead of two -- a wrong start costs one trip to the host
	 * less (HVK_SECAM_NO_SPEC=1: one check per wait, as before) */
	int pending = -1;
	for(;;)
	{
		int bad;
		if(pending >= 0) { bad = pending; pending = -1; }
		else if(first_bad >= 0)
		{
			bad = first_bad;
			first_bad = -1;
			if(bad) e->secam_spec_left = getenv("HVK_SECAM_NO_SPEC") ? 0 : 64;
		}
		else
		{
			const bool spec = rounds == 0 && e->secam_spec_left > 0 && !getenv("HVK_SECAM_DEBUG");
			if((r = hvk_launch_secam_check(&a, e->stream)) != HVK_OK) return(r);
			HIPCHK(hipMemcpyAsync(e->h_secam_count, a.count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
			if(spec)
			{
				if((r = hvk_launch_secam_redo(&a, 1, e->stream)) != HVK_OK) return(r);
				if((r = hvk_launch_secam_check(&a, e->stream)) != HVK_OK) return(r);
				HIPCHK(hipMemcpyAsync(e->h_secam_count + 1, a.count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
			}
			HIPCHK(hipStreamSynchronize(e->stream));
			bad = e->h_secam_count[0];
			if(spec && bad) pending = e->h_secam_count[1];
			if(rounds == 0)
			{
				if(bad) e->secam_spec_left = getenv("HVK_SECAM_NO_SPEC") ? 0 : 64;
				else if(e->secam_spec_left > 0) e->secam_spec_left--;
			}
		}
		if(bad && memo_used > 0)
		{
			/* A frame that took a kept set did not start from the state the set's walk had started from (or a line elsewhere started
			 * wrong and the repair would run on into rows that several frames of the block share): every set this block touched is
			 * dropped and the block's chain done again, every frame walked into its own row. Nothing of the first attempt is used,
			 * and it says nothing about how well entry states are guessed: the warm-up length is left alone. */
			_secam_kept_drop(e, nframes);
			e->secam_memo_restarts++;
			e->secam_memo_off++;
			/* (the state the block began with: the pinned word still holds it -- this block's own is copied there only when it is through) */
			e->secam_start = *e->h_secam_carry;
			HIPCHK(hipMemcpyAsync(a.carry, e->h_secam_carry, sizeof(hvk_secam_state_t), hipMemcpyHostToDevice, e->stream));
			memset(e->chroma_par, -1, (size_t) e->max_frames);
			r = _secam_on_device(e, first_frame, nframes);
			e->secam_memo_off--;
			return(r);
		}
		if(rounds == 0 && e->secam_adapt && !(e->secam_est && a.kf == NULL))     /* (no kept states and the estimate for every line: no warm-up length to follow) */
		{
			/* How many warm-up lines a start state needs depends on the pictures and costs a walk each. Exactness never
			 * rests on it -- the check does -- so the number follows what the batches show, carefully: a wrong start costs
			 * a redo round, which is dearer than the walk it saved. One line fewer after a run of clean batches (a run
			 * twice as long after every attempt that failed), two more as soon as anything fails. With the lines' states
			 * kept from the picture's last showing (a.seed) a picture that stays ends at NO warm-up line: its lines start
			 * from what they started from six frames ago, which is what they start from now. */
			if(bad == 0)
			{
				if(++e->secam_clean >= e->secam_patience && a.K > (e->secam_seeds ? 0 : 2)) { a.K--; e->secam_clean = 0; }
			}
			else
			{
				/* a few wrong starts: two lines more; many (a batch at K = 8 can have a quarter of its lines wrong, and the
				 * redo rounds then cost a hundred times what the warm-up saved): back to the full number at once */
				a.K = (int64_t) bad * a.R * 500 > a.total ? HVK_SECAM_WARMUP : (a.K + 2 < HVK_SECAM_WARMUP ? a.K + 2 : HVK_SECAM_WARMUP);
				e->secam_patience = e->secam_patience * 2 < 64 ? e->secam_patience * 2 : 64;
				e->secam_clean = 0;
			}
		}
		if(rounds == 0 && e->secam_est_ran && e->secam_ek_adapt)
		{
			/* How far up an estimate has to start depends on the pictures too: where the values behind the lines forget
			 * slowly (flat colours in the baseband modes) sixteen lines leave one start in a hundred wrong, twenty-four
			 * one in a thousand. More than one in two hundred wrong: eight lines more (up to 48); sixteen clean blocks: eight
			 * fewer again. */
			if((int64_t) bad * a.R * 200 > a.total) { a.EK = a.EK + 8 < 48 ? a.EK + 8 : 48; e->secam_ek_clean = 0; }
			else if(++e->secam_ek_clean >= 16 && a.EK > e->secam_ek_base) { a.EK -= 8; e->secam_ek_clean = 0; }
		}
		if(bad == 0) break;
		if(rounds == 0 && getenv("HVK_SECAM_DEBUG"))
		{
			/* which lines started wrong (a diagnostic: tools/secam_wrong_starts.py) */
			std::vector<int> fl((size_t) a.nruns);
			HIPCHK(hipMemcpy(fl.data(), a.flags, (size_t) a.nruns * sizeof(int), hipMemcpyDeviceToHost));
			int shown = 0;
			for(int r_ = 0; r_ < a.nruns && shown < 24; r_++) if(fl[(size_t) r_])
			{
				const int t_ = r_ * a.R, fr = t_ / a.ntasks, sl = t_ - fr * a.ntasks;
				fprintf(stderr, "libhvk: SECAM wrong start: frame %lld of the stream, task slot %d (run %d)\n", (long long) (first_frame + fr), sl, r_);
				shown++;
			}
		}
		if(rounds == 0) e->secam_counts[1] += (int64_t) bad * a.R;
		if(++rounds > HVK_SECAM_ROUNDS || getenv("HVK_SECAM_FORCE_FALLBACK"))
		{
			/* the host's chain takes the batch over from the state it began with */
			e->secam_start = *e->h_secam_carry;
			hvk_secam_set_state(e->secam, &e->secam_start, first_frame);
			for(int i = 0; i < nframes; i++)
			{
				const int slot = e->staged_slots[i], slot2 = e->staged_slots2[i];
				const hvk_slot_t *s = &e->slots[slot], *s2 = &e->slots[slot2];
				r = hvk_secam_frame(e->secam, first_frame + i, s->valid ? e->host_frames[slot] : NULL, s->valid ? s->width : 0, s->valid ? s->height : 0, s->interlaced,
				                    s2->valid ? e->host_frames[slot2] : NULL, s2->valid ? s2->width : 0, s2->valid ? s2->height : 0, s2->interlaced,
				                    e->h_chroma + (size_t) i * k.raster_samples);
				if(r != HVK_OK) return(r);
			}
			hvk_secam_get_state(e->secam, e->h_secam_carry, NULL);
			_secam_kept_drop(e, nframes);
			{
				/* (the host's chain writes the batch's own rows) */
				const int fields = k.fields;
				for(int i = 0; i < nframes; i++) for(int f_ = 0; f_ <= fields; f_++) e->h_fdesc[(size_t) i * (fields + 1) + f_].chroma_row = i;
				HIPCHK_P(hipMemcpyAsync(e->d_fdesc, e->h_fdesc, sizeof(hvk_framedesc_t) * nframes * (fields + 1), hipMemcpyHostToDevice, e->stream));
			}
			HIPCHK_P(hipMemcpyAsync(e->d_chroma, e->h_chroma, (size_t) nframes * k.raster_samples * 2, hipMemcpyHostToDevice, e->stream));
			HIPCHK(hipMemcpyAsync(a.carry, e->h_secam_carry, sizeof(hvk_secam_state_t), hipMemcpyHostToDevice, e->stream));
			e->secam_counts[3] += nframes;
			memset(e->chroma_par, -1, (size_t) e->max_frames);      /* (the host's chain wrote the rows) */
			return(HVK_OK);
		}
		e->secam_counts[2] += (int64_t) bad * a.R;
		if(pending >= 0) continue;      /* (this round's redo has run, and its check) */
		if((r = hvk_launch_secam_redo(&a, rounds, e->stream)) != HVK_OK) return(r);
	}

	if(rounds == 0) _secam_kept_commit(e, first_frame, nframes);        /* (not a line of the block started wrong: what its owners' walks left is what a later frame may take) */
	else _secam_kept_drop(e, nframes);
	if((r = hvk_launch_secam_carry(&a, e->stream)) != HVK_OK) return(r);
	HIPCHK(hipMemcpyAsync(e->h_secam_carry, a.carry, sizeof(hvk_secam_state_t), hipMemcpyDeviceToHost, e->stream));
	return(HVK_OK);
}

/* A stage whose check is still out (above): the count is read -- behind the render hvk_launch has queued by now -- and what
 * follows from it done. 0: the block's sub-carrier stands as rendered; 1: it was made again (render again); < 0: failure. */
int hvk_e_secam_resolve(hvk_engine *e)
{
	if(!e->secam_pending) return(0);
	e->secam_pending = 0;
	HIPCHK(hipEventSynchronize(e->secam_ev));
	const int bad = e->h_secam_count[0];
	e->secam_resolving = 1;
	const int r = _secam_rounds(e, e->secam_p_first, e->secam_p_n, e->secam_p_memo, bad);
	e->secam_resolving = 0;
	if(r != HVK_OK) { e->poisoned = 1; return(r); }
	return(bad ? 1 : 0);
}

/* The picture planes (hvk_direct.hip) of those of the named slots whose picture is new since its planes were made: one
 * prep launch for the pictures whose levels are looked up, one for those whose levels are computed. On the engine's
 * stream: behind the pictures' uploads, in front of every later render. */
/* The planes of those of the named slots whose picture is new since its planes were made, on `stream`: runs of
 * neighbouring slots with pictures of one geometry go into one launch, which gets the run as an argument -- nothing is
 * copied to the device and nothing waited for. Returns the number of pictures worked on (< 0: failure). */
int hvk_e_prep_dirty(hvk_engine *e, const int32_t *slots, int n, hipStream_t stream)
{
	const hvk_kconst_t &k = e->t.k;
	std::vector<int> todo;

	for(int i = 0; i < n; i++)
	{
		const int sl = slots[i];
		if(sl < 0 || sl >= e->frame_slots || !e->slots[sl].plane_dirty) continue;
		e->slots[sl].plane_dirty = 0;
		todo.push_back(sl);
	}
	if(todo.empty()) return(0);
	std::sort(todo.begin(), todo.end());

	auto kind = [&](int sl, hvk_prepgeo_t *g) -> int
	{
		const hvk_slot_t *ss = &e->slots[sl];
		g->fb_width = ss->valid ? ss->width : 0;
		g->fb_height = ss->valid ? ss->height : 0;
		g->fb_interlaced = ss->interlaced;
		g->fb_valid = ss->valid;
		return(e->levels_mode == HVK_LEVELS_COMPUTE || (e->levels_mode == HVK_LEVELS_AUTO && ss->valid && ss->many_colours));
	};

	hvk_raster_args_t ra;
	hvk_filter_args_t fa;
	hvk_e_kernel_args(e, &ra, &fa, NULL, 1);
	for(size_t i = 0; i < todo.size();)
	{
		hvk_prepgeo_t g, g2;
		memset(&g, 0, sizeof(g));
		const int lv = kind(todo[i], &g);
		size_t j = i + 1;
		for(; j < todo.size() && todo[j] == todo[j - 1] + 1; j++)
		{
			memset(&g2, 0, sizeof(g2));
			if(kind(todo[j], &g2) != lv || g2.fb_width != g.fb_width || g2.fb_height != g.fb_height || g2.fb_interlaced != g.fb_interlaced || g2.fb_valid != g.fb_valid) break;
		}
		g.slot0 = todo[i];
		g.frame_px = (int64_t) k.active_width * k.active_lines;
		ra.levels_computed = lv ? 1 + e->t.yuv.fast : 0;      /* (the plane kernels know the short forms) */
		const int r = hvk_launch_prep(&ra, &g, (int) (j - i), e->d_Lp + 16, e->d_Cp ? e->d_Cp + 16 : (e->d_UVp ? e->d_UVp + 16 : NULL), stream);
		if(r != HVK_OK) return(r);
		e->prep_count += (int64_t) (j - i);
		i = j;
	}
	return((int) todo.size());
}

/* ... of the staged block's frames [y0, y0 + n) (and of the slots named for the frames before them) */
int hvk_e_prep_staged(hvk_engine *e, int y0, int n, hipStream_t stream)
{
	int r = hvk_e_prep_dirty(e, e->staged_slots + y0, n, stream);
	if(r < 0) return(r);
	const int r2 = hvk_e_prep_dirty(e, e->staged_prev + y0, n, stream);
	return(r2 < 0 ? r2 : r + r2);
}

/* the last plane row of the staged block's last frame, kept for the next block's first frame (its slot may hold another
 * picture by then): behind the launch that made the planes */
int hvk_e_carry_copy(hvk_engine *e)
{
	if(!e->carry_copy_pending) return(HVK_OK);
	const size_t W = e->t.k.width;
	HIPCHK(hipMemcpyAsync(e->d_Lp + e->carry_to, e->d_Lp + e->carry_from, W * 2, hipMemcpyDeviceToDevice, e->stream));
	if(e->d_Cp) HIPCHK(hipMemcpyAsync(e->d_Cp + e->carry_to, e->d_Cp + e->carry_from, W * 4, hipMemcpyDeviceToDevice, e->stream));
	e->carry_copy_pending = 0;
	return(HVK_OK);
}

/* A block that was staged and never launched: its planes are made all the same (the next block's first frame may look
 * into its last one's) */
int hvk_e_flush_planes(hvk_engine *e)
{
	if(!e->direct || !e->prep_pending || !e->carry_copy_pending) return(HVK_OK);
	const int r = hvk_e_prep_staged(e, 0, e->staged, e->stream);
	if(r < 0) return(r);
	e->prep_pending = 0;
	return(hvk_e_carry_copy(e));
}

/* The planes of the named slots are made again before the next render shows them: what a caller does who wants the
 * per-picture work inside a clock of its own (bench.py). */
extern "C" int hvk_planes_refresh(hvk_engine_t *e, const int32_t *slots, int n)
{
	if(!e || !slots || n < 0) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	for(int i = 0; i < n; i++) if(slots[i] < 0 || slots[i] >= e->frame_slots) return(HVK_ERROR);
	if(e->secam_dev) for(int i = 0; i < n; i++)
	{
		e->slots[slots[i]].cells_valid[0] = e->slots[slots[i]].cells_valid[1] = 0;
		memset(e->slots[slots[i]].seeds_valid, 0, sizeof(e->slots[slots[i]].seeds_valid));
		memset(e->slots[slots[i]].memo_valid, 0, sizeof(e->slots[slots[i]].memo_valid));
	}
	if(!e->direct) return(HVK_OK);          /* this configuration renders straight from the pictures */
	for(int i = 0; i < n; i++) { e->slots[slots[i]].plane_dirty = 1; e->slots[slots[i]].shown = 0; }
	return(HVK_OK);
}

int hvk_e_stage(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots)
{
	if(!e || nframes < 1 || nframes > e->max_frames || stride < 1 || first_frame < 0) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	/* A stage that fails half way has moved the serial chains (sound carriers, SECAM colour, offset, passthru, FM
	 * video) forward for the frames before the failure; they cannot be rewound, so the stream would go on out of step
	 * without anyone noticing. Everything that can be checked is checked before the first of them is touched, and a
	 * failure after that point poisons the engine: every later call fails too. */
	if(e->poisoned) return(HVK_ERROR);
	if(e->secam_pending) { int r_ = hvk_e_secam_resolve(e); if(r_ < 0) return(r_); }     /* (a block staged and not launched) */
	for(int i = 0; i < nframes * e->t.k.fields; i++)
	{
		if(slots && (slots[i] < 0 || slots[i] >= e->frame_slots)) return(HVK_ERROR);
	}
	if(e->t.k.rs_irr && stride != 1) return(HVK_UNSUPPORTED);        /* frames of two lengths: a batch is one run of samples */
	/* (... whose NICAM symbol starts are kept as 29-bit offsets from its first sample: checked here, before any chain has moved) */
	if(e->t.k.rs_irr && e->audio && (_fstart(e, first_frame + nframes) - _fstart(e, first_frame)) * 8 >= 0x7FFFFFFF) return(HVK_UNSUPPORTED);
	if(e->secam && (stride != 1 || first_frame != e->secam_next)) return(HVK_UNSUPPORTED);   /* one serial chain over the whole stream (hvk_secam.c): frames in order, no gaps */
	{
		/* Where the last line of a frame shows picture (525 lines) it lies within the video filter's reach of the next
		 * frame's first samples: a frame whose predecessor the engine does not have -- a stride, a jump -- needs the
		 * caller to name the slot that holds it (hvk_stage_strided_prev(); the frame's own slot where the picture stays).
		 * Exact or refused: no "nearly". */
		const hvk_linedesc_t *dl = &e->t.desc[e->t.k.lines - 1];
		if(dl->ar > dl->al && !e->t.k.rawbb)
		{
			for(int i = 0; i < nframes; i++)
			{
				if(first_frame + i * stride == 0) continue;
				if(stride == 1 && i > 0) continue;
				if(stride == 1 && e->carry_valid && e->carry_frame + 1 == first_frame) continue;
				if(prev_slots && prev_slots[i] >= 0 && prev_slots[i] < e->frame_slots) continue;
				return(HVK_UNSUPPORTED);
			}
		}
	}

	const hvk_kconst_t &k = e->t.k;
	const int64_t FS = k.frame_samples;
	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	if(k.sv_ring && stride != 1) return(HVK_UNSUPPORTED);       /* the ring of line buffers is walked in stream order: refused before anything is queued */

	HIPCHK(hipSetDevice(e->device));
	{
		int r = hvk_e_flush_planes(e);           /* (a block staged and not launched) */
		if(r != HVK_OK) return(r);
	}
	/* the pinned side buffers are reused: the copies of the stage before have to be through. (Not the whole stream: a
	 * read-back queued with hvk_fetch_async() goes on while this stage's host pre-passes run.) */
	if(k.fm_video) HIPCHK(hipStreamSynchronize(e->stream));
	else if(e->staged_busy) { HIPCHK(hipEventSynchronize(e->ev_staged)); e->staged_busy = 0; }

	if(k.fm_video)
	{
		/* the FM phasor is one serial chain over the stream (hvk_tail.c): finish the
		 * previous batch, then take frames in order, no gaps */
		/* (a batch whose samples have all been handed to the FM thread needs no finishing, and nothing here waits for the
		 * thread: the next batch's host pre-passes run beside it) */
		int r = hvk_e_fm_finish(e);
		if(r != HVK_OK) return(r);
		if(stride != 1 || _fstart(e, first_frame) != e->fm_batch_pos + (int64_t) e->fm_done) return(HVK_UNSUPPORTED);
		e->fm_batch_pos = _fstart(e, first_frame);
		e->fm_done = 0;
		e->fm_async_upto = 0;
	}

	const int fields = k.fields;            /* descriptors (and slots named by the caller) per frame */
	int many = 0;

	if(k.fm_video && (k.vf_type || k.rs_L) && first_frame == 0 && k.out_prime > 0)
	{
		/* The line pipeline's never-emitted start-up samples pass through the FM modulator as well
		 * (src/video.c:4936-4952 drops them only at the output): what the sound carriers add to them is
		 * wanted now, before the audio chain moves on to the first frame (hvk_launch does the rest) */
		free(e->fm_prime_car);
		e->fm_prime_car = (int16_t *) calloc((size_t) k.out_prime * 2, sizeof(int16_t));
		if(!e->fm_prime_car) return(HVK_OUT_OF_MEMORY);
		if(e->audio && k.has_carriers)
		{
			int64_t k0 = 0;
			int n = hvk_audio_generate(e->audio, 0, k.out_prime, e->fm_prime_car, e->sym_tmp, e->sym_tmp ? e->symbol_stride : 0, &k0);
			if(n < 0) { e->poisoned = 1; return(n); }
		}
		e->fm_prime_pending = 1;
	}

	/* The sound chains over a run of samples: the carriers' side stream, and NICAM's symbol schedule as rows per tile.
	 * Per frame -- or, with frames of two lengths, once for the batch, which the filter kernel then takes as one long
	 * frame (tiles counted from the batch's first sample). */
	auto stage_audio = [&](const int64_t a_pos, const int64_t a_len, const size_t a_off, const int a_row, const int a_symcap, const int a_ntiles, const int64_t a_frame) -> int
	{
		const int64_t m0 = a_pos + (int64_t) k.out_prime;
		int64_t k0 = 0;
		int n = hvk_audio_generate(e->audio, m0, a_len,
			e->h_car ? e->h_car + a_off * 2 : NULL,
			e->sym_tmp, a_symcap, &k0);
		if(n < 0) { e->poisoned = 1; return(n); }

		if(k.sis)
		{
			/* the frame's sound-in-syncs bursts: made by the chains' pass just now, line by line (hvk_audio.c) */
			/* (and the first line's of the frame behind it: the filter of this frame's last samples looks into it) */
			/* (behind the resampler the output stands a raster line back, k.rs_shift: the line after that one as well) */
			const int rows = k.lines + (k.rs_L ? 2 : 1);
			int r = hvk_audio_sis_fetch(e->audio, a_frame * k.lines, rows, (uint8_t *) (e->h_sis_bits + (size_t) a_row * rows * 2));
			if(r != HVK_OK) { e->poisoned = 1; return(r); }
		}

		if(k.has_nicam)
		{
			/* tabulate the symbol schedule for the frame (src/nicam728.c:398-407):
			 * symbol k starts at sps * k - floor(k * dsl / decimation); entries are
			 * (start relative to the frame's first sample) << 3 | valid << 2 | value */
			int32_t *tab = e->h_sym + (size_t) a_row * e->symbol_stride;
			int32_t *tile = e->h_tile + (size_t) a_row * e->tiles * HVK_NICAM_ROW;
			int newest = 0;

			for(int j = 0; j < a_symcap; j++)
			{
				const int64_t kk = k0 + j;
				if(j >= n || kk < 0 || e->sym_tmp[j] == 0xFF) { tab[j] = 0; continue; }
				const int64_t start = (int64_t) k.nicam_sps * kk - (kk * k.nicam_dsl) / k.nicam_decimation - m0;
				tab[j] = (int32_t) (start * 8) | 4 | (e->sym_tmp[j] & 3);
			}

			/* per tile a dense row: the HVK_NICAM_SYMS symbols from 6 before the newest
			 * one that has started by the tile's first sample, then the mixer table
			 * position of that sample */
			while(newest + 1 < n && !(tab[newest] & 4)) newest++;   /* slab entries before the stream's first symbol */
			for(int b = 0; b < a_ntiles; b++)
			{
				const int64_t pos = (int64_t) b * HVK_TILE;
				int32_t *row = tile + (size_t) b * HVK_NICAM_ROW;
				while(newest + 1 < n && (tab[newest + 1] & 4) && (tab[newest + 1] >> 3) <= pos) newest++;
				for(int q = 0; q < HVK_NICAM_SYMS; q++)
				{
					const int j = newest - (HVK_NICAM_BACK - 1) + q;
					row[q] = (j >= 0 && j < a_symcap) ? tab[j] : 0;
				}
				row[HVK_NICAM_SYMS] = (int32_t) ((m0 + pos) % k.nicam_cc_len);
			}
		}
		return(HVK_OK);
	};

	for(int i = 0; i < nframes; i++)
	{
		hvk_framedesc_t *f = &e->h_fdesc[(size_t) i * (fields + 1) + 1];
		const int slot = slots ? slots[(size_t) i * fields] : 0;
		const int slot2 = (slots && fields == 2) ? slots[(size_t) i * fields + 1] : slot;
		if(slot < 0 || slot >= e->frame_slots || slot2 < 0 || slot2 >= e->frame_slots) return(HVK_ERROR);
		const hvk_slot_t *s = &e->slots[slot];
		e->staged_slots[i] = slot;
		e->staged_slots2[i] = slot2;

		for(int fld = 0; fld < fields; fld++)
		{
			hvk_framedesc_t *d = f + fld;
			const int sl = fld ? slot2 : slot;
			const hvk_slot_t *ss = &e->slots[sl];

			memset(d, 0, sizeof(*d));
			d->frame_index = first_frame + i * stride;
			d->fb_offset = (int64_t) sl * frame_px;
			d->fb_width = ss->valid ? ss->width : 0;
			d->fb_height = ss->valid ? ss->height : 0;
			d->pixel_stride = 1;
			d->line_stride = ss->width;
			d->vframe_x = (k.active_width - d->fb_width) / 2;      /* src/video.c:4896-4897 */
			d->vframe_y = (k.active_lines - d->fb_height) / 2;
			d->fb_interlaced = ss->interlaced;
			d->fb_valid = ss->valid;
			if(ss->valid && ss->many_colours) many = 1;
			d->parity = (int32_t) ((d->frame_index + 1) & 1);
			d->chroma_row = i;              /* (SECAM: the frame's place in the batch, unless the stage takes or makes a kept set: _secam_on_device) */
			d->plane_row0 = sl * k.lines;
			d->clut_off0 = k.colour ? (uint32_t) (((uint64_t) d->frame_index * (uint64_t) k.raster_samples) % k.clw) : 0;
		}

		if(e->secam && e->secam_dev) e->secam_next++;
		else if(e->secam)
		{
			const hvk_slot_t *s2 = &e->slots[slot2];
			int r = hvk_secam_frame(e->secam, f->frame_index, s->valid ? e->host_frames[slot] : NULL,
			                        f->fb_width, f->fb_height, s->interlaced,
			                        s2->valid ? e->host_frames[slot2] : NULL, s2->valid ? s2->width : 0, s2->valid ? s2->height : 0, s2->interlaced,
			                        e->h_chroma + (size_t) i * k.raster_samples);
			if(r != HVK_OK) return(r);
			e->secam_next++;
		}

		if(e->h_raw)
		{
			/* the lines of this frame's slab: the last line of the frame before, the frame, the first
			 * line of the next (the filter looks 25 samples into it); zeros where nothing is queued */
			const int W = k.width;
			int16_t *dst = e->h_raw + (size_t) i * k.slab_lines * W;
			for(int j = 0; j < k.slab_lines; j++)
			{
				const int64_t g = f->frame_index * k.lines + j - 1;
				const int64_t at = g * W - e->raw_base;
				if(g >= 0 && at >= 0 && at + W <= (int64_t) e->raw_q->size()) memcpy(dst + (size_t) j * W, e->raw_q->data() + at, (size_t) W * 2);
				else memset(dst + (size_t) j * W, 0, (size_t) W * 2);
			}
		}

		/* where the frame's samples stand in the stream, how many they are, where they go in the batch's side buffers */
		const int64_t fpos = _fstart(e, f->frame_index), flen = _fstart(e, f->frame_index + 1) - fpos;
		const size_t foff = k.rs_irr ? (size_t) (fpos - _fstart(e, first_frame)) : (size_t) i * FS;
		if(k.rs_irr)
		{
			/* hvk_k_resample: c = B D - f RS L, and the frame's place in the batch's run */
			e->h_frec[2 * i + 0] = (int) (fpos * k.rs_D - f->frame_index * (int64_t) k.raster_samples * k.rs_L);
			e->h_frec[2 * i + 1] = (int) foff;
		}

		if(e->h_off)
		{
			int r = hvk_tail_offset_stream(e->tail, fpos, flen, e->h_off + foff * 2);
			if(r != HVK_OK) { e->poisoned = 1; return(r); }
		}
		if(e->h_pass)
		{
			int r = hvk_tail_passthru_stream(e->tail, fpos, flen, e->h_pass + foff * 2);
			if(r != HVK_OK) { e->poisoned = 1; return(r); }
		}

		if(e->audio && !k.rs_irr)
		{
			int r = stage_audio(fpos, flen, foff, i, e->symbol_stride, e->tiles, f->frame_index);
			if(r != HVK_OK) return(r);
		}
	}
	if(e->audio && k.rs_irr)
	{
		const int64_t A0 = _fstart(e, first_frame), T = _fstart(e, first_frame + nframes) - A0;
		int r = stage_audio(A0, T, 0, 0, e->symbol_stride * nframes, (int) ((T + HVK_TILE - 1) / HVK_TILE), first_frame);
		if(r != HVK_OK) return(r);
		if(k.sis)
		{
			/* (the raster is made frame by frame: every frame its own rows of bursts, not the batch's first frame's alone) */
			const int rows = k.lines + (k.rs_L ? 2 : 1);
			for(int i = 1; i < nframes; i++)
			{
				r = hvk_audio_sis_fetch(e->audio, (first_frame + i) * k.lines, rows, (uint8_t *) (e->h_sis_bits + (size_t) i * rows * 2));
				if(r != HVK_OK) { e->poisoned = 1; return(r); }
			}
		}
	}

	/* The frame before each frame: the one staged just before it, the last frame of the batch before
	 * (its last line's source row was kept), nothing at the start of the stream. A strided render does
	 * not have the frames in between: it takes the frame's own picture, which is right for a picture
	 * that does not change and wrong by up to the filter's reach (25 samples) otherwise. */
	for(int i = 0; i < nframes; i++)
	{
		hvk_framedesc_t *p = &e->h_fdesc[(size_t) i * (fields + 1)];
		const hvk_framedesc_t *own = &e->h_fdesc[(size_t) i * (fields + 1) + fields];
		const int ps = prev_slots ? prev_slots[i] : -1;
		if(first_frame + i * stride == 0) { memset(p, 0, sizeof(*p)); }     /* nothing before the stream */
		else if(stride == 1 && i > 0) *p = e->h_fdesc[(size_t) (i - 1) * (fields + 1) + fields];
		else if(stride == 1 && e->carry_valid && e->carry_frame + 1 == first_frame) *p = e->carry;
		else if(ps >= 0 && ps < e->frame_slots)
		{
			/* the caller has the frame before in a slot (hvk_stage_strided_prev): its picture on the halo line */
			const hvk_slot_t *ss = &e->slots[ps];
			*p = *own;
			p->fb_offset = (int64_t) ps * frame_px;
			p->fb_width = ss->valid ? ss->width : 0;
			p->fb_height = ss->valid ? ss->height : 0;
			p->line_stride = ss->width;
			p->vframe_x = (k.active_width - p->fb_width) / 2;
			p->vframe_y = (k.active_lines - p->fb_height) / 2;
			p->fb_interlaced = ss->interlaced;
			p->fb_valid = ss->valid;
			p->plane_row0 = ps * k.lines;
		}
		else *p = *own;                                             /* a strided render or a jump without it: the frame's own picture */
		/* the colour table position the kernel counts lines from is this frame's, also on the halo line */
		p->clut_off0 = own->clut_off0;
		p->frame_index = own->frame_index;
		p->parity = own->parity;
	}
	if(e->direct)
	{
		/* the picture planes of every picture this batch shows that is new since its planes were made: when it is launched */
		for(int i = 0; i < nframes; i++) e->staged_prev[i] = (prev_slots && prev_slots[i] >= 0 && prev_slots[i] < e->frame_slots) ? prev_slots[i] : -1;
		e->prep_pending = 1;
	}
	{
		/* keep what the last frame of this batch shows on its last line */
		const hvk_framedesc_t *last = &e->h_fdesc[(size_t) (nframes - 1) * (fields + 1) + fields];
		const hvk_linedesc_t *d = &e->t.desc[(size_t) last->parity * k.lines + k.lines - 1];
		int vy = d->src_row;
		if(vy >= 0 && k.interlaced != 0 && last->fb_interlaced != k.interlaced) vy += 1;
		vy -= last->vframe_y;
		e->carry = *last;
		e->carry_frame = last->frame_index;
		e->carry_valid = 1;
		if(d->ar > d->al)
		{
			/* not the row this batch's first frame is about to read */
			e->carry_row ^= 1;
			if(last->fb_valid && vy >= 0 && vy < last->fb_height)
			{
				const size_t carry_off = frame_px * e->frame_slots + (size_t) e->carry_row * k.active_width;
				HIPCHK(hipMemcpyAsync(e->d_pool + carry_off, e->d_pool + last->fb_offset + (int64_t) vy * last->line_stride,
				                      (size_t) last->fb_width * 4, hipMemcpyDeviceToDevice, e->stream));
				e->carry.fb_offset = (int64_t) carry_off;
				e->carry.line_stride = 0;       /* every row of the kept frame is that one row */
			}
			else e->carry.fb_valid = 0;         /* (no picture on that line: black -- the raster kernel's reading) */
			if(e->direct)
			{
				/* ... and its planes' last row, picture on it or not (hvk_k_direct reads the row whatever the frame showed): the
				 * slot may hold another picture by the time the next batch looks */
				const size_t W = k.width;
				e->carry_from = ((size_t) last->plane_row0 + k.lines - 1) * W + 16;
				e->carry_to = ((size_t) e->plane_carry_row + e->carry_row) * W + 16;
				e->carry_copy_pending = 1;         /* (copied behind the launch that makes the planes) */
				e->carry.plane_row0 = e->plane_carry_row + e->carry_row - (k.lines - 1);
			}
		}
		else e->carry.fb_valid = 0;
	}
	/* (SECAM on the device: the descriptors go once the colour chain's plan has named every frame's sub-carrier rows -- _secam_kept_plan;
	 * nothing in front of it reads them: the planes are made from the slots' geometry) */
	if(!e->secam_dev) HIPCHK_P(hipMemcpyAsync(e->d_fdesc, e->h_fdesc, sizeof(hvk_framedesc_t) * nframes * (fields + 1), hipMemcpyHostToDevice, e->stream));
	if(e->h_ops)
	{
		_build_vbi_ops(e, nframes);
		HIPCHK_P(hipMemcpyAsync(e->d_ops, e->h_ops, (size_t) nframes * HVK_VBI_OPS * HVK_VBI_OPWORDS * 4, hipMemcpyHostToDevice, e->stream));
		HIPCHK_P(hipMemcpyAsync(e->d_map, e->h_map, (size_t) nframes * k.lines, hipMemcpyHostToDevice, e->stream));
		/* teletext packets and caption pairs are consumed by the batch they were queued for */
		if(e->h_tt_mask) memset(e->h_tt_mask, 0, (size_t) e->max_frames * 4);
		if(e->cc_pairs) memset(e->cc_pairs, 0, (size_t) e->max_frames * 3);
	}
	if(e->h_raw)
	{
		HIPCHK_P(hipMemcpyAsync(e->d_raw, e->h_raw, (size_t) nframes * k.slab_lines * k.width * 2, hipMemcpyHostToDevice, e->stream));
		/* what no later frame can need goes: everything before the last line of the last frame staged */
		const int64_t keep = ((first_frame + (int64_t) (nframes - 1) * stride + 1) * k.lines - 1) * k.width;
		if(keep > e->raw_base)
		{
			const int64_t drop = std::min<int64_t>(keep - e->raw_base, (int64_t) e->raw_q->size());
			e->raw_q->erase(e->raw_q->begin(), e->raw_q->begin() + drop);
			e->raw_base += drop;
		}
	}
	e->levels_computed = e->levels_mode == HVK_LEVELS_COMPUTE || (e->levels_mode == HVK_LEVELS_AUTO && many);
	if(e->secam_dev)
	{
		int r = _secam_on_device(e, first_frame, nframes);
		if(r != HVK_OK) { e->poisoned = 1; return(r); }
		e->secam_last_frame = stride == 1 ? first_frame + nframes - 1 : -1;
		e->secam_last_key = hvk_slot_key(e->slots, e->staged_slots[nframes - 1]);
	}
	else if(e->h_chroma) HIPCHK(hipMemcpyAsync(e->d_chroma, e->h_chroma, (size_t) nframes * k.raster_samples * 2, hipMemcpyHostToDevice, e->stream));
	if(e->h_sis_bits) HIPCHK_P(hipMemcpyAsync(e->d_sis_bits, e->h_sis_bits, (size_t) nframes * (k.lines + (k.rs_L ? 2 : 1)) * 8, hipMemcpyHostToDevice, e->stream));
	e->staged_samples = _fstart(e, first_frame + nframes) - _fstart(e, first_frame);     /* (nframes * FS but for frames of two lengths) */
	if(e->h_car) HIPCHK_P(hipMemcpyAsync(e->d_car, e->h_car, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_off) HIPCHK_P(hipMemcpyAsync(e->d_off, e->h_off, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_pass) HIPCHK_P(hipMemcpyAsync(e->d_pass, e->h_pass, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_frec) HIPCHK_P(hipMemcpyAsync(e->d_frec, e->h_frec, (size_t) nframes * 2 * sizeof(int), hipMemcpyHostToDevice, e->stream));
	if(k.sv_ring)
	{
		int r = hvk_e_sv_ring_records(e, first_frame, nframes);        /* (hvk_k_svq's per-line records: in front of ev_staged like every side input) */
		if(r != HVK_OK) { e->poisoned = 1; return(r); }
	}
	if(e->h_sym)
	{
		HIPCHK_P(hipMemcpyAsync(e->d_tile, e->h_tile, (size_t) nframes * e->tiles * HVK_NICAM_ROW * 4, hipMemcpyHostToDevice, e->stream));
	}

	HIPCHK_P(hipEventRecord(e->ev_staged, e->stream));
	e->staged_busy = 1;

	e->levels_computed = e->levels_mode == HVK_LEVELS_COMPUTE || (e->levels_mode == HVK_LEVELS_AUTO && many);
	e->staged = nframes;
	e->staged_first = first_frame;
	e->staged_stride = stride;
	return(HVK_OK);
}

