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
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "hvk_internal.h"
#include "hvk_kernels.h"

#define HVK_VERSION "hacktv-amd 0.1 (gfx950)"
#define HVK_MIN_FRAME_SLOTS 4
#define HVK_MAX_FRAME_SLOTS 1024
#define HVK_TIMING_SLOTS 512
#define HVK_UPLOAD_RING 8
#define HVK_FETCH_TICKETS 4
#define HVK_POOL_PAD 4096
#define HVK_PREP_EVENTS 8

extern "C" {
int hvk_audio_symbol_info(const hvk_audio_t *a, int64_t m, int64_t *k, int64_t *start);
}

struct hvk_slot_t {
	int valid;
	int width, height;      /* after the centre crop */
	int interlaced;
	int64_t par_num, par_den;   /* pixel aspect of the source frame (hvk_frame_aspect), 1:1 unless told */
	int many_colours;           /* a sample of its pixels shows more colours than the level table serves from cache */
	int plane_dirty;            /* the picture planes (hvk_direct.hip) have not been made from this picture yet */
	int shown;                  /* ... although a block has shown it already (from the pixels, hvk_fused.hip): it stays, so its planes are worth making now */
	int cells_valid[2];         /* SECAM: the picture's low-passed colour cells (hvk_secam.hip) stand in the store, by frame parity */
	int seeds_valid[6];         /* SECAM: the picture has been shown with this frame number modulo 6: its lines' entry states are kept */
};

struct hvk_engine {
	hvk_tables_t t;
	hvk_audio_t *audio;
	hvk_secam_t *secam;
	int64_t secam_next;        /* next frame the SECAM pre-pass expects */
	uint32_t **host_frames;     /* SECAM: host copy of every frame slot (cropped, dense) */
	int16_t *d_chroma, *h_chroma;
	signed char *chroma_par;    /* [max_frames] the frame parity the slab's rows were last written with by the device's chain (-1: clear before use) */
	int16_t *d_chroma_alloc;    /* (d_chroma lies 64 entries inside it: a lane of hvk_k_direct whose 8 samples straddle the start of a frame's first line reads up to 7 entries in front) */
	/* SECAM on the device (hvk_secam.hip): tables, the transposed low-pass store, the tasks' states */
	int secam_dev;              /* the sub-carrier is computed by the device; the host's chain is the fall-back */
	hvk_secam_args_t sa;
	void *d_secam[19];          /* what sa points into (freed at close) */
	int secam_walk_ok;          /* 1: hvk_k_secam_walk<0> may be taken; 2: its computed FM steps and decoded gains equal the tables' on every index (tried at open) */
	int secam_walk_mode;        /* HVK_SECAM_WALK: -1 the engine's choice per stage, 0 the chain kernel, 1 / 2 hvk_k_secam_walk<0 / 1> */
	int64_t secam_walk_stages[3];   /* stages that went through the chain kernel / hvk_k_secam_walk<0> / <1> */
	int secam_est_ran, secam_ek_adapt, secam_ek_base, secam_ek_clean;      /* this stage ran the estimate; its reach (a.EK) follows the blocks */
	int secam_est;              /* new pictures' lines start from estimated states (hvk_k_secam_est), not from warm-up walks */
	int64_t secam_est_stages;   /* stages that ran the estimate kernel */
	int *h_secam_rows;          /* [4][max_frames] pinned: the frames' rows in the cell stores, the frames whose cells are made, warm-up lines per frame, rows of the kept states */
	int secam_seeds;            /* warm-ups start from the states the picture's lines had the last time (kept per row) */
	int secam_last_new;         /* the last staged frame showed a picture whose cells had to be made */
	int secam_cell_cache;       /* a picture's cells are kept for the frames that show it again (one picture per frame: no --interlace) */
	int *h_secam_count;         /* pinned: failures of the last check */
	int secam_lanes;            /* lanes of eight waves per SIMD */
	int secam_adapt;            /* the number of warm-up lines follows the pictures (no HVK_SECAM_WARMUP in the environment) */
	int secam_clean, secam_patience;    /* batches without a wrong start in a row; how many of them before a line less is tried */
	hvk_secam_state_t *h_secam_carry;   /* pinned: the state after the last batch */
	hvk_secam_state_t secam_start;      /* ... as the host's chain would need it to take over */
	int64_t secam_counts[4];
	int32_t *staged_slots2;     /* [max_frames] the slot of the second field's picture */
	uint32_t *h_tt_pk;          /* teletext packets queued for the next batch: [max_frames][32][12] */
	uint32_t *h_tt_mask;        /* [max_frames] rows present */
	/* VBI data lines (teletext, WSS, VITC): symbol store, per-frame op list and line map */
	void *d_vbi_sym, *d_vbi_val;
	uint32_t *d_ops, *h_ops;    /* [max_frames][HVK_VBI_OPS][HVK_VBI_OPWORDS] */
	int8_t *d_map, *h_map;      /* [max_frames][lines] */
	void *d_vits_l, *d_vits_c, *d_fsc_rows;
	void *d_sis_dense, *d_sis_win, *d_sis_first;    /* sound-in-syncs tables */
	uint32_t *d_sis_bits, *h_sis_bits;              /* [max_frames][lines][2]: the lines' bursts */
	/* --raw-bb-file: queued stream (raw_q[0] is sample raw_base) and its per-batch slab */
	std::vector<int16_t> *raw_q; int64_t raw_base;
	int16_t *d_raw, *h_raw;
	uint8_t *cc_pairs;          /* CC608: [max_frames][3] { present, c1, c2 } queued for the next batch */
	hvk_packed_taps_t notch;
	hvk_tail_t *tail;           /* FM video / offset / passthru serial state (hvk_tail.c) */
	int16_t *d_off, *h_off;     /* offset phasor side stream, int16 pairs */
	int16_t *d_pass, *h_pass;   /* passthru samples, int16 pairs */
	int16_t *h_fm;              /* FM video: the batch's modulated samples (host) */
	int64_t fm_batch_pos;       /* output position of the staged batch's first sample */
	size_t fm_done;             /* samples of the batch modulated so far */
	size_t fm_async_upto;       /*   ... of which these went through the FM thread into a caller's buffer (not into h_fm) */
	int fm_launched;            /* the staged batch has been rendered and is not fully modulated yet */
	/* FM video behind hvk_fetch_async(): the read-back goes straight into the caller's buffer and a thread of the engine's
	 * runs the phasor over it there, job after job in stream order; hvk_fetch_wait() waits for the job */
	struct fm_job_t { int ticket; int64_t pos, count; int16_t *iq; hipEvent_t ev; };
	std::thread *fm_thread;
	std::mutex *fm_mu;
	std::condition_variable *fm_cv;
	std::deque<fm_job_t> *fm_q;
	int fm_quit;
	int fm_status[4];           /* [HVK_FETCH_TICKETS] */
	int fm_prime_pending;       /* FM video with the video filter: the phasor has yet to run over the pipeline's start-up samples */
	int16_t *fm_prime_car;      /*   their sound carrier samples (out_prime int16 pairs) */
	int device;             /* -1: host tables only */
	int max_frames;
	int frame_slots;
	int symbol_stride;
	hipStream_t stream;         /* stream in use */
	hipStream_t own_stream;

	/* constant tables */
	void *d_yuv, *d_yuvparams, *d_desc, *d_pulses, *d_linebase, *d_clut, *d_burst, *d_ghost, *d_tapd, *d_cca;
	int levels_mode;            /* HVK_LEVELS_AUTO / _TABLE / _COMPUTE (hvk_set_levels) */
	int levels_computed;        /* what the staged block uses */
	void *d_mfma_a;             /* video filter taps as the A operand of v_mfma_i32_16x16x64_i8 (NULL: taps out of its range) */
	void *d_mfma_a28;           /* ... for hvk_k_fused's window (28 samples of lead) */
	int fused_ok;               /* this configuration can render from the pixels in one kernel (hvk_fused.hip) */
	int fused_mode;             /* HVK_FUSED: 0 never, 1 always, unset (-1): when at least half of a block's pictures are new */
	int64_t fused_count;        /* launches that went that way */
	int mfma_ci, mfma_cq;
	/* per batch */
	uint32_t *d_pool;
	uint32_t *d_pool_alloc;     /* (d_pool lies HVK_POOL_PAD pixels inside it and as many lie behind the slots: hvk_k_prep's lanes read the 8 pixels
	                             * under their 8 samples wherever the line's picture begins and ends, and keep what is picture) */
	hvk_framedesc_t *d_fdesc;   /* [max_frames][1 + fields]: the frame before (only its last line is looked at:
	                             * the halo line in front), then one descriptor per field */
	/* the last line's source row of the last frame staged, kept behind the slots: the next batch's first halo */
	hvk_framedesc_t carry; int carry_valid; int64_t carry_frame;
	int carry_row;              /* which of the two kept rows `carry` points at: the batch being staged reads one while the other is written */
	int16_t *d_S;
	int16_t *d_C;           /* --s-video: the sub-carrier slab */
	int16_t *d_C2;          /* --s-video with --pixelrate: the resampled sub-carrier (the resampler's second channel) */
	/* ... where the lines have two widths and the video filter is on (hvk_kconst_t.sv_ring): the Q channel made line by line
	 * the way the reference's ring of line buffers pairs it (hvk_k_svq) */
	int16_t *d_C2_alloc;    /* d_C2 lies sv_hist samples inside it: the end of the batch before's stream, kept in front of this batch's */
	int16_t *d_Cq;          /* what the filter kernel reads as Q */
	int *h_svrec, *d_svrec; /* [max_frames * lines][4] per emitted line: first sample in the batch, width | delta << 16 | kind << 20, source of the last sample */
	int sv_hist;
	int64_t sv_tail_first, sv_tail_total;   /* the batch whose sub-carrier stream lies in d_C2 (first frame; -1: none), its samples */
	int16_t *d_S2; void *d_rs_taps;     /* --pixelrate: the resampled stream the filter kernel reads, the poly-phase taps */
	int16_t *d_car;
	int32_t *d_sym;
	int32_t *d_tile;
	int16_t *d_out;
	void *d_conv; size_t conv_bytes;   /* hvk_fetch_as scratch */
	void *d_sums;               /* hvk_block_sums(): two 64-bit sums */

	/* pinned staging */
	hvk_framedesc_t *h_fdesc;
	int16_t *h_car;
	int32_t *h_sym;
	int32_t *h_tile;
	uint8_t *sym_tmp;
	int tiles;                  /* NICAM symbol rows per frame: one per filter tile */
	int direct;                 /* this configuration renders in one kernel from picture planes (hvk_direct.hip) */
	int last_direct;            /* the last launch did: the raster slab in HBM was not written */
	/* picture planes: [plane_rows][width] each, 16 entries of slack in front; rows: lines per frame slot, two kept
	 * last lines (the halo of the next batch's first frame, 525-line modes), a row of zeros */
	int *d_UVp;                 /* SECAM: the pictures' colour-difference levels, laid out like d_Cp (hvk_k_prep writes them, hvk_k_secam_cells reads them) */
	int16_t *d_Lp; int *d_Cp; int *d_clut3; uint32_t *d_lineoff; uint32_t inv_w;
	void *d_tilerec; int tiles_pad;     /* hvk_tilerec_t [2][tiles_pad] */
	int plane_rows, plane_carry_row, plane_zero_row, clut_reg;
	/* ... and behind them, per frame of a batch, a row for every line the optional stages (VBI data, test signals) can
	 * write to: rendered whole by the raster kernel per frame, taken by hvk_k_direct instead of the planes' rows */
	int ovr_n, ovr_row0;
	int16_t *d_ovr_list, *d_ovr_idx;
	/* The planes of the pictures a staged block shows for the first time are made when the block is LAUNCHED, a chunk of
	 * frames at a time on a stream of their own, each chunk's render behind its planes: hvk_k_prep of chunk c + 1 runs
	 * beside hvk_k_direct of chunk c (one is bound by memory latency, the other by vector issue), and a chunk's planes
	 * are read back while they still lie in the 256 MiB Infinity Cache */
	hipStream_t prep_stream;
	hipEvent_t ev_fork, ev_prep[HVK_PREP_EVENTS];
	int prep_chunk;             /* frames per chunk (HVK_PREP_CHUNK) */
	int prep_streams;           /* 2: the planes on a stream of their own (HVK_PREP_STREAMS) */
	int prep_pending;           /* the staged block's planes have not been made yet */
	int32_t *staged_prev;       /* [max_frames] the slot the caller named for the frame before (hvk_stage_strided_prev), -1: none */
	int carry_copy_pending; size_t carry_from, carry_to;    /* the staged block's last plane row has yet to be kept (525 lines) */
	int64_t prep_count;         /* pictures the planes were made from so far */
	/* pinned staging for source frames: a small ring, each buffer guarded by an event recorded behind its copy, so that
	 * hvk_frame_upload() waits for the copy that last used THAT buffer only -- never for the stream */
	uint32_t *h_frame[HVK_UPLOAD_RING];
	hipEvent_t up_ev[HVK_UPLOAD_RING];
	int up_busy[HVK_UPLOAD_RING];
	int up_next;
	hipEvent_t ev_staged;       /* after the last host-to-device copy of a stage: the pinned side buffers are free again */
	int staged_busy;
	hipEvent_t fetch_ev[HVK_FETCH_TICKETS];   /* hvk_fetch_async() */
	int fetch_busy[HVK_FETCH_TICKETS];        /* handed out and not waited for yet */
	int fetch_next;

	hvk_slot_t *slots;          /* [frame_slots] */
	hvk_packed_taps_t ctaps, itaps, qtaps;

	int64_t next_frame;
	int staged;             /* frames staged for the next launch */
	int64_t staged_samples, last_samples;   /* output samples of the staged / the last launched batch (frames x frame_samples; frames of two lengths: what they add up to) */
	int *h_frec, *d_frec;   /* [max_frames][2] --pixelrate with frames of two lengths: hvk_k_resample's per-frame record */
	int32_t *staged_slots;  /* [max_frames] the slot each of them shows */
	int64_t staged_first, staged_stride;
	int last_frames;        /* frames of the last launch (for fetch) */
	int ghost_dirty;
	int poisoned;           /* a stage failed after the serial chains had moved on: the stream is out of step, nothing more is rendered */

	/* kernel timing with HIP events on the engine's stream */
	int timing;
	hipEvent_t ev[HVK_TIMING_SLOTS][3];
	int ev_used;
	double t_sum[2];
	int64_t t_n[2];
};

static void _fm_worker(hvk_engine *e);


/* ... after the serial chains have moved on for a batch: the failure leaves the stream out of step for good */
#define HIPCHK_P(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
	e->poisoned = 1; return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)
#define HIPCHK(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
	return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)

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

/* first output sample of stream frame f (frames of two lengths with some --pixelrate pairs: hvk_tables.c) */
static inline int64_t _fstart(const hvk_engine *e, int64_t f) { return(hvk_tables_frame_start(&e->t, f)); }

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

	/* the kernels exist for these chroma filter lengths (pixel rates of about 6 to 33 MHz) and for
	 * the NICAM pulse lengths the LDS table holds: say so now, not at the first render */
	if(e->t.k.colour)
	{
		const int nt = e->t.k.chroma_ntaps;
		if((nt < 5 || nt > 25 || !(nt & 1)) && !(nt == 3 && e->t.chroma_unfiltered))
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
			if(e->direct && !e->ovr_n && !(getenv("HVK_TILEREC") && atoi(getenv("HVK_TILEREC")) == 0))
			{
				/* per frame parity and tile of 1024 outputs (the tiles of the last, partial workgroup included): the lines its window
				 * lies in -- the arithmetic of hvk_direct.hip:direct_line_loads() / direct_line(), done once here */
				const int DGT = 4, lead = k.vf_type ? 26 : 0, W = k.width, tiles = (k.frame_samples + HVK_TILE - 1) / HVK_TILE;
				e->tiles_pad = (tiles + DGT - 1) / DGT * DGT;
				std::vector<hvk_tilerec_t> rec((size_t) 2 * e->tiles_pad);
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
			/* (a line's old content can be the last sample of a chunk several turns of the ring back: that much of the stream
			 * stays in front of the batch) */
			e->sv_hist = ((8 * k.sv_ring + 4) * e->t.max_width + 7) & ~7;
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
		/* (a line's worth of zeros behind the frames: what hvk_k_direct adds to the lines around a frame) */
		OPENHIP(hipMalloc((void **) &e->d_chroma_alloc, ((size_t) max_frames * RS + k.width + 128) * 2));
		OPENHIP(hipMemset(e->d_chroma_alloc, 0, ((size_t) max_frames * RS + k.width + 128) * 2));
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
			OPENHIP(hipMalloc(&e->d_secam[9], (size_t) a.tpad * 4 + 64));
			OPENHIP(hipMalloc(&e->d_secam[10], (size_t) max_frames * 4 * sizeof(int)));
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
			OPENHIP(hipHostMalloc((void **) &e->h_secam_rows, (size_t) max_frames * 4 * sizeof(int), hipHostMallocDefault));
			OPENHIP(hipMemset(e->d_secam[8], 0, sizeof(hvk_secam_state_t) + 64));
			OPENHIP(hipHostMalloc((void **) &e->h_secam_count, 64, hipHostMallocDefault));
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
			e->fm_thread = new std::thread(_fm_worker, e);
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
		                e->d_tapd, e->d_cca, e->d_pool_alloc, e->d_fdesc, e->d_S, e->d_car, e->d_sym, e->d_tile, e->d_out, e->d_chroma_alloc, e->d_vbi_sym, e->d_vbi_val, e->d_ops, e->d_map, e->d_vits_l, e->d_vits_c, e->d_sis_dense, e->d_sis_win, e->d_sis_first, e->d_sis_bits, e->d_conv, e->d_off, e->d_pass, e->d_S2, e->d_C2_alloc ? (void *) e->d_C2_alloc : (void *) e->d_C2, e->d_Cq, e->d_svrec, e->d_rs_taps, e->d_C, e->d_raw, e->d_mfma_a, e->d_frec, e->d_ovr_list, e->d_ovr_idx, e->d_sums, e->d_mfma_a28, e->d_tilerec, e->d_fsc_rows };
		for(void *p : dev) if(p) (void) hipFree(p);
		for(int i = 0; i < HVK_UPLOAD_RING; i++) { if(e->h_frame[i]) (void) hipHostFree(e->h_frame[i]); if(e->up_ev[i]) (void) hipEventDestroy(e->up_ev[i]); }
		for(void *p : e->d_secam) if(p) (void) hipFree(p);
		if(e->h_secam_count) (void) hipHostFree(e->h_secam_count);
		if(e->h_secam_carry) (void) hipHostFree(e->h_secam_carry);
		if(e->ev_staged) (void) hipEventDestroy(e->ev_staged);
		if(e->ev_fork) (void) hipEventDestroy(e->ev_fork);
		for(int i = 0; i < HVK_PREP_EVENTS; i++) if(e->ev_prep[i]) (void) hipEventDestroy(e->ev_prep[i]);
		if(e->prep_stream) { (void) hipStreamSynchronize(e->prep_stream); (void) hipStreamDestroy(e->prep_stream); }
		for(int i = 0; i < HVK_FETCH_TICKETS; i++) if(e->fetch_ev[i]) (void) hipEventDestroy(e->fetch_ev[i]);
		if(e->h_svrec) (void) hipHostFree(e->h_svrec);
		void *host[] = { e->h_fdesc, e->h_car, e->h_sym, e->h_tile, e->h_chroma, e->h_tt_pk, e->h_tt_mask, e->h_ops, e->h_map, e->h_off, e->h_pass, e->h_fm, e->h_raw, e->h_sis_bits, e->h_secam_rows, e->h_frec };
		for(void *p : host) if(p) (void) hipHostFree(p);
		if(e->own_stream) (void) hipStreamDestroy(e->own_stream);
	}

	free(e->sym_tmp);
	free(e->chroma_par);
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
	if(!s->valid) return(HVK_OK);

	const uint32_t *src = fb + (size_t) y * width + x;
	s->many_colours = _many_colours(src, width, w, h);

	HIPCHK(hipSetDevice(e->device));
	const size_t frame_px = (size_t) k.active_width * k.active_lines;
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
	if(!f->valid) return(HVK_OK);

	const size_t frame_px = (size_t) k.active_width * k.active_lines;
	hipEvent_t a = NULL, b = NULL;
	HIPCHK(hipSetDevice(from->device));
	HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
	HIPCHK(hipEventRecord(a, from->stream));
	HIPCHK(hipSetDevice(e->device));
	HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
	HIPCHK(hipStreamWaitEvent(e->stream, a, 0));
	HIPCHK(hipMemcpyPeerAsync(e->d_pool + slot * frame_px, e->device, from->d_pool + from_slot * frame_px, from->device,
	                          (size_t) f->width * f->height * 4, e->stream));
	HIPCHK(hipEventRecord(b, e->stream));
	HIPCHK(hipSetDevice(from->device));
	HIPCHK(hipStreamWaitEvent(from->stream, b, 0));
	(void) hipEventDestroy(a);      /* (released once the work queued on them is through) */
	(void) hipEventDestroy(b);
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
	return(e->sa.K);
}

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

/* FM video: bring the host copy of the current batch up to `upto` samples -- fetch the
 * modulator's input from the device and run the serial tail over it (hvk_tail.c) */
static void _fm_worker(hvk_engine *e)
{
	std::unique_lock<std::mutex> lk(*e->fm_mu);
	(void) hipSetDevice(e->device);
	for(;;)
	{
		e->fm_cv->wait(lk, [e] { return(e->fm_quit || !e->fm_q->empty()); });
		if(e->fm_q->empty()) break;
		const hvk_engine::fm_job_t j = e->fm_q->front();
		lk.unlock();
		int r = hipEventSynchronize(j.ev) == hipSuccess ? HVK_OK : HVK_ERROR;
		if(r == HVK_OK) r = hvk_tail_fm_apply(e->tail, j.pos, j.count, j.iq);
		lk.lock();
		e->fm_status[j.ticket] = r;
		if(r != HVK_OK) e->poisoned = 1;        /* the phasor did not run over these samples: every later job would be out of step */
		e->fm_q->pop_front();           /* (behind the work: an empty queue means nothing is being worked on) */
		e->fm_cv->notify_all();
	}
}

/* every queued job through (what comes next works on the phasor itself) */
static void _fm_wait_all(hvk_engine *e)
{
	if(!e->fm_thread) return;
	std::unique_lock<std::mutex> lk(*e->fm_mu);
	e->fm_cv->wait(lk, [e] { return(e->fm_q->empty()); });
}

static int _fm_upto(hvk_engine *e, size_t upto)
{
	_fm_wait_all(e);
	if(upto <= e->fm_done) return(HVK_OK);
	if(upto > (size_t) e->last_samples) return(HVK_ERROR);      /* (frames of two lengths: what the batch's frames add up to, not frames x the longer one) */
	const size_t n = upto - e->fm_done;
	HIPCHK(hipMemcpyAsync(e->h_fm + e->fm_done * 2, e->d_out + e->fm_done * 2, n * 4, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	int r = hvk_tail_fm_apply(e->tail, e->fm_batch_pos + (int64_t) e->fm_done, (int64_t) n, e->h_fm + e->fm_done * 2);
	if(r != HVK_OK) return(r);
	e->fm_done = upto;
	return(HVK_OK);
}

/* ... and to its end, so that the phasor stands at the next batch's first sample */
static int _fm_finish(hvk_engine *e)
{
	if(!e->fm_launched) return(HVK_OK);
	int r = _fm_upto(e, (size_t) e->last_samples);
	if(r == HVK_OK) e->fm_launched = 0;
	return(r);
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

static int _stage(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots);
static void _kernel_args(hvk_engine *e, hvk_raster_args_t *pra, hvk_filter_args_t *pfa, void *d_iq, int64_t out_stride);

extern "C" int hvk_stage_strided(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots)
{
	return(_stage(e, first_frame, stride, nframes, slots, NULL));
}

extern "C" int hvk_stage_strided_prev(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots)
{
	return(_stage(e, first_frame, stride, nframes, slots, prev_slots));
}

/* SECAM: the sub-carrier of the staged frames on the device (hvk_secam.hip) -- every line at once from derived entry
 * states, then check / redo rounds until every line started from the state the line before it left. The frame
 * descriptors and pictures are on their way to the device (same stream). */
static int _prep_dirty(hvk_engine *e, const int32_t *slots, int n, hipStream_t stream);

static int _secam_on_device(hvk_engine_t *e, int64_t first_frame, int nframes)
{
	const hvk_kconst_t &k = e->t.k;
	hvk_secam_args_t &a = e->sa;
	int r, rounds = 0;

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
		if((r = _prep_dirty(e, e->staged_slots, nframes, e->stream)) < 0) return(r);
		a.uvp = e->d_UVp + 16;
	}
	/* A lane's walk is a chain of dependent operations: a SIMD interleaves a few waves of it for free (measured: 1156
	 * waves on 1024 SIMDs take as long as 578). Longer runs per lane only when the batch has more lines than eight
	 * waves per SIMD hold. */
	a.R = (a.total + e->secam_lanes - 1) / e->secam_lanes;
	if(a.R < 1) a.R = 1;
	if(getenv("HVK_SECAM_RUN")) a.R = atoi(getenv("HVK_SECAM_RUN")) > 0 ? atoi(getenv("HVK_SECAM_RUN")) : 1;
	a.nruns = (a.total + a.R - 1) / a.R;
	e->secam_start = *e->h_secam_carry;

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
		HIPCHK_P(hipMemcpyAsync(e->d_secam[10], rows, (size_t) e->max_frames * 4 * sizeof(int), hipMemcpyHostToDevice, e->stream));
	}

	/* The chain writes every line on its task list whole and never another; the list follows the frame's parity. A slab
	 * row that was last written with the same parity has nothing to clear (blocks of even length, one after the other:
	 * none of them), the others are cleared in runs. */
	for(int i = 0; i < nframes; )
	{
		int j = i;
		while(j < nframes && e->chroma_par[j] != (signed char) ((first_frame + j + 1) & 1)) j++;
		if(j > i) HIPCHK(hipMemsetAsync(e->d_chroma + (size_t) i * k.raster_samples, 0, (size_t) (j - i) * k.raster_samples * 2, e->stream));
		for(int q = i; q < j; q++) e->chroma_par[q] = (signed char) ((first_frame + q + 1) & 1);
		i = j + 1;
	}
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
		if((r = hvk_launch_secam_cells_chain(&a, e->secam_est && want, walk, e->stream)) != HVK_OK) return(r);
		if(e->secam_est && want) e->secam_est_stages++;
		e->secam_est_ran = e->secam_est && want;
	}
	e->secam_counts[0] += a.total;

	for(;;)
	{
		if((r = hvk_launch_secam_check(&a, e->stream)) != HVK_OK) return(r);
		HIPCHK(hipMemcpyAsync(e->h_secam_count, a.count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipStreamSynchronize(e->stream));
		const int bad = *e->h_secam_count;
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
		if(rounds == 0) e->secam_counts[1] += (int64_t) bad * a.R;
		if(++rounds > HVK_SECAM_ROUNDS || getenv("HVK_SECAM_FORCE_FALLBACK"))
		{
			/* the host's chain takes the batch over from the state it began with */
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
			HIPCHK_P(hipMemcpyAsync(e->d_chroma, e->h_chroma, (size_t) nframes * k.raster_samples * 2, hipMemcpyHostToDevice, e->stream));
			HIPCHK(hipMemcpyAsync(a.carry, e->h_secam_carry, sizeof(hvk_secam_state_t), hipMemcpyHostToDevice, e->stream));
			e->secam_counts[3] += nframes;
			memset(e->chroma_par, -1, (size_t) e->max_frames);      /* (the host's chain wrote the rows) */
			return(HVK_OK);
		}
		e->secam_counts[2] += (int64_t) bad * a.R;
		if((r = hvk_launch_secam_redo(&a, rounds, e->stream)) != HVK_OK) return(r);
	}

	if((r = hvk_launch_secam_carry(&a, e->stream)) != HVK_OK) return(r);
	HIPCHK(hipMemcpyAsync(e->h_secam_carry, a.carry, sizeof(hvk_secam_state_t), hipMemcpyDeviceToHost, e->stream));
	return(HVK_OK);
}

/* The picture planes (hvk_direct.hip) of those of the named slots whose picture is new since its planes were made: one
 * prep launch for the pictures whose levels are looked up, one for those whose levels are computed. On the engine's
 * stream: behind the pictures' uploads, in front of every later render. */
/* The planes of those of the named slots whose picture is new since its planes were made, on `stream`: runs of
 * neighbouring slots with pictures of one geometry go into one launch, which gets the run as an argument -- nothing is
 * copied to the device and nothing waited for. Returns the number of pictures worked on (< 0: failure). */
static int _prep_dirty(hvk_engine *e, const int32_t *slots, int n, hipStream_t stream)
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
	_kernel_args(e, &ra, &fa, NULL, 1);
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
static int _prep_staged(hvk_engine *e, int y0, int n, hipStream_t stream)
{
	int r = _prep_dirty(e, e->staged_slots + y0, n, stream);
	if(r < 0) return(r);
	const int r2 = _prep_dirty(e, e->staged_prev + y0, n, stream);
	return(r2 < 0 ? r2 : r + r2);
}

/* the last plane row of the staged block's last frame, kept for the next block's first frame (its slot may hold another
 * picture by then): behind the launch that made the planes */
static int _carry_copy(hvk_engine *e)
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
static int _flush_planes(hvk_engine *e)
{
	if(!e->direct || !e->prep_pending || !e->carry_copy_pending) return(HVK_OK);
	const int r = _prep_staged(e, 0, e->staged, e->stream);
	if(r < 0) return(r);
	e->prep_pending = 0;
	return(_carry_copy(e));
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
	}
	if(!e->direct) return(HVK_OK);          /* this configuration renders straight from the pictures */
	for(int i = 0; i < n; i++) { e->slots[slots[i]].plane_dirty = 1; e->slots[slots[i]].shown = 0; }
	return(HVK_OK);
}

static int _stage(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots)
{
	if(!e || nframes < 1 || nframes > e->max_frames || stride < 1 || first_frame < 0) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	/* A stage that fails half way has moved the serial chains (sound carriers, SECAM colour, offset, passthru, FM
	 * video) forward for the frames before the failure; they cannot be rewound, so the stream would go on out of step
	 * without anyone noticing. Everything that can be checked is checked before the first of them is touched, and a
	 * failure after that point poisons the engine: every later call fails too. */
	if(e->poisoned) return(HVK_ERROR);
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

	HIPCHK(hipSetDevice(e->device));
	{
		int r = _flush_planes(e);           /* (a block staged and not launched) */
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
		int r = _fm_finish(e);
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
	HIPCHK_P(hipMemcpyAsync(e->d_fdesc, e->h_fdesc, sizeof(hvk_framedesc_t) * nframes * (fields + 1), hipMemcpyHostToDevice, e->stream));
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
	}
	else if(e->h_chroma) HIPCHK(hipMemcpyAsync(e->d_chroma, e->h_chroma, (size_t) nframes * k.raster_samples * 2, hipMemcpyHostToDevice, e->stream));
	if(e->h_sis_bits) HIPCHK_P(hipMemcpyAsync(e->d_sis_bits, e->h_sis_bits, (size_t) nframes * (k.lines + (k.rs_L ? 2 : 1)) * 8, hipMemcpyHostToDevice, e->stream));
	e->staged_samples = _fstart(e, first_frame + nframes) - _fstart(e, first_frame);     /* (nframes * FS but for frames of two lengths) */
	if(e->h_car) HIPCHK_P(hipMemcpyAsync(e->d_car, e->h_car, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_off) HIPCHK_P(hipMemcpyAsync(e->d_off, e->h_off, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_pass) HIPCHK_P(hipMemcpyAsync(e->d_pass, e->h_pass, (size_t) e->staged_samples * 4, hipMemcpyHostToDevice, e->stream));
	if(e->h_frec) HIPCHK_P(hipMemcpyAsync(e->d_frec, e->h_frec, (size_t) nframes * 2 * sizeof(int), hipMemcpyHostToDevice, e->stream));
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
 * is a chunk of the width of line j - 1. */
static int _sv_ring_q(hvk_engine *e)
{
	const hvk_kconst_t &k = e->t.k;
	const int64_t s = 1 + (k.vf_type ? k.delay_lines : 0), WL = (int64_t) k.width * k.rs_L, D = k.rs_D;
	auto S = [&](int64_t j) { return(((j + s) * WL + D - 1) / D - (s * WL + D - 1) / D); };
	auto width = [&](int64_t j) { return((int) (S(j + 1) - S(j))); };
	const int wmax = e->t.max_width, ring = k.sv_ring;
	const int64_t f0 = e->staged_first, j0 = f0 * k.lines, base = S(j0);
	const long slab_in = (long) k.slab_lines * k.width;
	const int nlines = e->staged * k.lines;
	if(e->staged_stride != 1) return(HVK_UNSUPPORTED);

	for(int i = 0; i < nlines; i++)
	{
		const int64_t j = j0 + i;
		const int w = width(j), wp = j + s - 1 >= 0 ? width(j - 1) : wmax, delta = wmax - wp;
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
				/* upwards: the last sample of the newest chunk of the longer width that lay in this buffer: k turns of the ring back */
				kind = 3;       /* (none: the buffer is as it was allocated) */
				for(int t = 1; t <= 8; t++)
				{
					const int64_t m = j - (int64_t) t * ring;
					if(m + s - 1 < 0) break;
					if(width(m - 1) == wmax)
					{
						const int64_t at = S(m) + (wmax - 1) - base;        /* (its delta is 0) */
						if(at >= -(int64_t) e->sv_hist) { kind = 2; src = (int) at; }
						break;
					}
				}
			}
		}
		e->h_svrec[4 * i + 0] = (int) (S(j) - base);
		e->h_svrec[4 * i + 1] = w | (delta << 16) | (kind << 20);
		e->h_svrec[4 * i + 2] = src;
		e->h_svrec[4 * i + 3] = 0;
	}
	HIPCHK(hipMemcpyAsync(e->d_svrec, e->h_svrec, (size_t) nlines * 16, hipMemcpyHostToDevice, e->stream));
	int r = hvk_launch_svq(e->d_svrec, nlines, e->d_C2, e->d_C, e->d_Cq, k.s_lead, e->stream);
	if(r != HVK_OK) return(r);
	e->sv_tail_first = f0;
	e->sv_tail_total = e->staged_samples;
	return(HVK_OK);
}

/* ... before a NEW batch's sub-carrier stream is made: the end of the stream that lies there (the batch before's) goes in front
 * of it (a batch launched again finds what it found the first time) */
static int _sv_ring_keep(hvk_engine *e)
{
	const hvk_kconst_t &k = e->t.k;
	if(e->sv_tail_first < 0 || e->sv_tail_first == e->staged_first) return(HVK_OK);
	if(e->sv_tail_total >= e->sv_hist)
	{
		HIPCHK(hipMemcpyAsync(e->d_C2 + k.s_lead - e->sv_hist, e->d_C2 + k.s_lead + e->sv_tail_total - e->sv_hist, (size_t) e->sv_hist * 2, hipMemcpyDeviceToDevice, e->stream));
	}
	else HIPCHK(hipMemsetAsync(e->d_C2 + k.s_lead - e->sv_hist, 0, (size_t) e->sv_hist * 2, e->stream));
	return(HVK_OK);
}

static void _kernel_args(hvk_engine *e, hvk_raster_args_t *pra, hvk_filter_args_t *pfa, void *d_iq, int64_t out_stride)
{
	hvk_raster_args_t &ra = *pra;
	memset(&ra, 0, sizeof(ra));
	ra.k = e->t.k;
	ra.ctaps = e->ctaps;
	ra.notch = e->notch;
	ra.chroma = e->t.k.rawbb ? e->d_raw : e->d_chroma;
	ra.vbi_sym = (const int *) e->d_vbi_sym;
	ra.vbi_val = (const int16_t *) e->d_vbi_val;
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

extern "C" int hvk_launch_strided_out(hvk_engine_t *e, void *d_iq, int64_t out_stride)
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
	_kernel_args(e, &ra, &fa, d_iq, out_stride);

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
			da.D.chroma = e->d_chroma ? e->d_chroma + (size_t) y0 * e->t.k.raster_samples : NULL;
			da.D.chroma_zero = (int) ((size_t) (e->max_frames - y0) * e->t.k.raster_samples + 16);
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
				const int np = _prep_staged(e, y0, n, ps);
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
			if((r = _carry_copy(e)) != HVK_OK) return(r);
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

extern "C" int hvk_fetch(hvk_engine_t *e, int16_t *iq, size_t first, size_t count)
{
	if(!e || !iq) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(first + count > (size_t) e->last_samples) return(HVK_ERROR);
	HIPCHK(hipSetDevice(e->device));
	if(e->t.k.fm_video)
	{
		/* (what went out through hvk_fetch_async() was modulated in the caller's buffer: it is not here) */
		if(first < e->fm_async_upto) return(HVK_ERROR);
		int r = _fm_upto(e, first + count);
		if(r != HVK_OK) return(r);
		memcpy(iq, e->h_fm + first * 2, count * 4);
		return(HVK_OK);
	}
	HIPCHK(hipMemcpyAsync(iq, e->d_out + first * 2, count * 4, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	return(HVK_OK);
}

extern "C" int hvk_fetch_async(hvk_engine_t *e, int16_t *iq, size_t first, size_t count)
{
	if(!e || !iq) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(first + count > (size_t) e->last_samples) return(HVK_ERROR);
	HIPCHK(hipSetDevice(e->device));
	const int t = e->fetch_next;
	/* a ticket goes out again only when its last copy has been waited for: a caller with more than HVK_FETCH_TICKETS
	 * copies in flight would otherwise wait on the wrong one */
	if(e->fetch_busy[t]) return(HVK_ERROR);
	e->fetch_next = (e->fetch_next + 1) % HVK_FETCH_TICKETS;
	if(e->t.k.fm_video && e->fm_thread && first == e->fm_done && e->fm_launched && count > 0)
	{
		/* the FM phasor runs on the host (see hvk_fetch()): the modulator's input goes into the caller's buffer, the
		 * engine's FM thread turns it into the output there once the copy is through -- in stream order, behind the
		 * jobs queued before. The caller's thread goes on */
		HIPCHK(hipMemcpyAsync(iq, e->d_out + first * 2, count * 4, hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipEventRecord(e->fetch_ev[t], e->stream));
		{
			std::lock_guard<std::mutex> lk(*e->fm_mu);
			e->fm_status[t] = HVK_OK;
			e->fm_q->push_back({ t, e->fm_batch_pos + (int64_t) first, (int64_t) count, iq, e->fetch_ev[t] });
		}
		e->fm_cv->notify_all();
		e->fm_done = first + count;
		e->fm_async_upto = e->fm_done;
		if(e->fm_done == (size_t) e->last_samples) e->fm_launched = 0;
		e->fetch_busy[t] = 2;
		return(t);
	}
	if(e->t.k.fm_video)
	{
		/* (out of order, or with --passthru, whose queue the caller's thread fills: in this call) */
		int r = hvk_fetch(e, iq, first, count);
		if(r != HVK_OK) return(r);
	}
	/* (one copy moves a block at the link's rate -- 56 GB/s, profiles/r05_d2h_speed.txt; in two halves on two streams it is no
	 * faster. What halves the rate is the FIRST copy into a fresh page-locked buffer: a caller keeps its buffers) */
	else HIPCHK(hipMemcpyAsync(iq, e->d_out + first * 2, count * 4, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipEventRecord(e->fetch_ev[t], e->stream));
	e->fetch_busy[t] = 1;
	return(t);
}

extern "C" int hvk_fetch_wait(hvk_engine_t *e, int ticket)
{
	if(!e || ticket < 0 || ticket >= HVK_FETCH_TICKETS) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(!e->fetch_busy[ticket]) return(HVK_ERROR);
	if(e->fetch_busy[ticket] == 2)
	{
		/* a job of the FM thread's */
		std::unique_lock<std::mutex> lk(*e->fm_mu);
		e->fm_cv->wait(lk, [e, ticket] { for(const auto &j : *e->fm_q) if(j.ticket == ticket) return(false); return(true); });
		e->fetch_busy[ticket] = 0;
		return(e->fm_status[ticket]);
	}
	HIPCHK(hipEventSynchronize(e->fetch_ev[ticket]));
	e->fetch_busy[ticket] = 0;
	return(HVK_OK);
}

extern "C" void *hvk_host_alloc(hvk_engine_t *e, size_t bytes)
{
	void *p = NULL;
	if(!e || e->device < 0 || bytes == 0) return(NULL);
	if(hipSetDevice(e->device) != hipSuccess) return(NULL);
	if(hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return(NULL);
	return(p);
}

extern "C" void hvk_host_free(hvk_engine_t *e, void *p)
{
	(void) e;
	if(p) (void) hipHostFree(p);
}

extern "C" long hvk_fetch_as(hvk_engine_t *e, void *dst, size_t first, size_t count, int type, int complex_out)
{
	if(!e || !dst || type < HVK_UINT8 || type > HVK_FLOAT) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(e->t.k.fm_video) return(HVK_UNSUPPORTED);   /* the final samples are not on the device */
	if(first + count > (size_t) e->last_samples) return(HVK_ERROR);

	const size_t unit = (type <= HVK_INT8 ? 1 : (type <= HVK_INT16 ? 2 : 4)) * (complex_out ? 2 : 1);
	const size_t bytes = count * unit;
	HIPCHK(hipSetDevice(e->device));

	/* converted samples go through a scratch buffer sized on first use */
	if(bytes > e->conv_bytes)
	{
		if(e->d_conv) HIPCHK(hipFree(e->d_conv));
		e->d_conv = NULL;
		e->conv_bytes = 0;
		HIPCHK(hipMalloc(&e->d_conv, bytes));
		e->conv_bytes = bytes;
	}

	int r = hvk_launch_convert(e->d_out + first * 2, count, type, complex_out != 0, e->d_conv, e->stream);
	if(r != HVK_OK) return(r);
	HIPCHK(hipMemcpyAsync(dst, e->d_conv, bytes, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	return((long) bytes);
}

extern "C" int hvk_fetch_raster(hvk_engine_t *e, int16_t *dst, size_t first, size_t count)
{
	/* frame-local raster of the last launch: frame i's samples follow frame
	 * i - 1's; the slab's halo lines are skipped */
	if(!e || !dst) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	const hvk_kconst_t &k = e->t.k;
	const size_t FS = k.raster_samples;
	if(first + count > (size_t) e->last_frames * FS) return(HVK_ERROR);
	HIPCHK(hipSetDevice(e->device));
	if(e->last_direct)
	{
		/* the one-kernel render keeps the raster in LDS: run the raster kernel over the staged batch to have
		 * it in HBM */
		hvk_raster_args_t ra;
		hvk_filter_args_t fa;
		if(e->staged != e->last_frames) return(HVK_ERROR);
		_kernel_args(e, &ra, &fa, NULL, 1);
		int r = hvk_launch_raster(&ra, e->stream);
		if(r != HVK_OK) return(r);
	}
	HIPCHK(hipStreamSynchronize(e->stream));
	while(count > 0)
	{
		const size_t fr = first / FS, off = first % FS;
		const size_t n = count < FS - off ? count : FS - off;
		HIPCHK(hipMemcpy(dst, e->d_S + fr * (size_t) k.slab_lines * k.width + k.width + off, n * 2, hipMemcpyDeviceToHost));
		dst += n; first += n; count -= n;
	}
	return(HVK_OK);
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
	return(k.secam || k.fm_video || k.rs_irr || k.has_passthru || k.rawbb || k.sis);
}

/* hvk_k_sums: a grid-stride pass over the words, a lane's two partial sums folded through the wave and one pair of
 * 64-bit atomic adds per wave (the sums are modulo 2^64: any order gives the same) */
extern "C" int hvk_launch_sums(const void *iq, size_t count, unsigned long long *sums, hipStream_t stream);

extern "C" int hvk_block_sums(hvk_engine_t *e, size_t first, size_t count, uint64_t sums[2])
{
	if(!e || !sums) return(HVK_ERROR);
	if(e->device < 0) return(HVK_NO_DEVICE);
	if(e->t.k.fm_video) return(HVK_UNSUPPORTED);
	if(first + count > (size_t) e->last_samples) return(HVK_ERROR);
	HIPCHK(hipSetDevice(e->device));
	if(!e->d_sums) HIPCHK(hipMalloc((void **) &e->d_sums, 16));
	HIPCHK(hipMemsetAsync(e->d_sums, 0, 16, e->stream));
	int r = hvk_launch_sums(e->d_out + first * 2, count, (unsigned long long *) e->d_sums, e->stream);
	if(r != HVK_OK) return(r);
	HIPCHK(hipMemcpyAsync(sums, e->d_sums, 16, hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	return(HVK_OK);
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
		snprintf(dk, sizeof(dk), "hvk_k_direct<%d, %d, %d, %d, %d, %d>", vf, col, exact, e->ovr_n ? 1 : 0, (k.has_carriers && k.has_nicam && vf && !e->ovr_n) ? 1 : 0, e->d_tilerec ? 1 : 0);
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
