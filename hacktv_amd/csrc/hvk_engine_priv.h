/* hvk_engine_priv.h -- what the translation units of the engine share: the engine's state, the error macros, the
 * helpers that cross a file boundary. Not part of the ABI (include/hacktv_amd.h is).
 *
 *   hvk_engine.cpp         life cycle (hvk_open_rates / hvk_close), tables and residency in HBM, uploads, the small calls
 *   hvk_engine_stage.cpp   staging a batch: host pre-passes, side inputs, VBI op lists, the SECAM colour stage, picture planes
 *   hvk_engine_launch.cpp  the launch policy (planes + hvk_k_direct / hvk_k_fused / raster + filter pair), hvk_render*, hvk_sync
 *   hvk_engine_fetch.cpp   read-back (hvk_fetch*), the FM video thread, sample formats, block sums
 */
#ifndef HVK_ENGINE_PRIV_H
#define HVK_ENGINE_PRIV_H

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
	int memo_valid[6];          /* SECAM: ... and the sub-carrier rows and states that walk left are kept whole (hvk_engine_stage.cpp: kept sub-carrier) */
	int64_t memo_prev[6];       /*   ... behind WHICH picture the set's frame stood (hvk_slot_key() of the frame before it): what the frame started from */
	uint32_t gen;               /* counts the pictures the slot has held */
};

/* a picture by slot and count: the same key, the same pixels */
static inline int64_t hvk_slot_key(const hvk_slot_t *slots, int slot) { return(((int64_t) slot << 32) | slots[slot].gen); }

struct hvk_engine {
	hvk_tables_t t;
	hvk_audio_t *audio;
	hvk_secam_t *secam;
	int64_t secam_next;        /* next frame the SECAM pre-pass expects */
	int64_t *secam_prev_key;    /* [max_frames] the staged frames' pictures before them (kept sub-carrier sets: hvk_engine_stage.cpp) */
	int64_t secam_last_frame, secam_last_key;   /* the last frame the device's colour chain went through and the picture it showed (-1: none) */
	uint32_t **host_frames;     /* SECAM: host copy of every frame slot (cropped, dense) */
	int16_t *d_chroma, *h_chroma;
	signed char *chroma_par;    /* [max_frames] the frame parity the slab's rows were last written with by the device's chain (-1: clear before use) */
	int16_t *d_chroma_alloc;    /* (d_chroma lies 64 entries inside it: a lane of hvk_k_direct whose 8 samples straddle the start of a frame's first line reads up to 7 entries in front) */
	/* SECAM on the device (hvk_secam.hip): tables, the transposed low-pass store, the tasks' states */
	int secam_dev;              /* the sub-carrier is computed by the device; the host's chain is the fall-back */
	hvk_secam_args_t sa;
	void *d_secam[21];          /* what sa points into (freed at close) */
	int secam_walk_ok;          /* 1: hvk_k_secam_walk<0> may be taken; 2: its computed FM steps and decoded gains equal the tables' on every index (tried at open) */
	int secam_walk_mode;        /* HVK_SECAM_WALK: -1 the engine's choice per stage, 0 the chain kernel, 1 / 2 hvk_k_secam_walk<0 / 1> */
	int64_t secam_walk_stages[3];   /* stages that went through the chain kernel / hvk_k_secam_walk<0> / <1> */
	int secam_est_ran, secam_ek_adapt, secam_ek_base, secam_ek_clean;      /* this stage ran the estimate; its reach (a.EK) follows the blocks */
	int secam_est;              /* new pictures' lines start from estimated states (hvk_k_secam_est), not from warm-up walks */
	int64_t secam_est_stages;   /* stages that ran the estimate kernel */
	int *h_secam_rows;          /* [8][max_frames] pinned: the frames' rows in the cell stores, the frames whose cells are made, warm-up lines per frame, rows of the kept states;
	                             * kept sub-carrier: frames that take their set, frames that make it, rows of the sub-carrier store, sets */
	int secam_memo_slots;       /* picture slots whose sub-carrier is kept per frame number modulo 6 (0: none); their rows follow the batch's in d_chroma */
	int secam_memo_off;         /* HVK_SECAM_KEEP=0, or a stage is being done again without them */
	int64_t secam_memo_frames, secam_memo_restarts;   /* frames that took a kept set; stages done again because one did not start where its set had */
	int secam_seeds;            /* warm-ups start from the states the picture's lines had the last time (kept per row) */
	int secam_last_new;         /* the last staged frame showed a picture whose cells had to be made */
	int secam_cell_cache;       /* a picture's cells are kept for the frames that show it again (one picture per frame: no --interlace) */
	int *h_secam_count;         /* pinned: failures of the last check [0], of the check behind a redo queued with it [1] */
	int secam_spec_left;        /* blocks to go for which the first check brings its redo round along (a recent block had a wrong start) */
	int secam_lanes;            /* lanes of eight waves per SIMD */
	int secam_adapt;            /* the number of warm-up lines follows the pictures (no HVK_SECAM_WARMUP in the environment) */
	int secam_clean, secam_patience;    /* batches without a wrong start in a row; how many of them before a line less is tried */
	hvk_secam_state_t *h_secam_carry;   /* pinned: the state after the last batch */
	hvk_secam_state_t secam_start;      /* ... as the host's chain would need it to take over */
	int secam_defer, secam_pending, secam_resolving;    /* the first check's count read in hvk_launch, behind the queued render (hvk_e_secam_resolve) */
	int64_t secam_p_first; int secam_p_n, secam_p_memo;
	hipEvent_t secam_ev;
	int64_t secam_counts[4];
	int32_t *staged_slots2;     /* [max_frames] the slot of the second field's picture */
	uint32_t *h_tt_pk;          /* teletext packets queued for the next batch: [max_frames][32][12] */
	uint32_t *h_tt_mask;        /* [max_frames] rows present */
	/* VBI data lines (teletext, WSS, VITC): symbol store, per-frame op list and line map */
	int16_t *h_ovr_idx;         /* [lines] host copy of hvk_dptrs_t.ovr_idx (the tile records carry it) */
	void *d_vbi_sym, *d_vbi_val;
	void *d_vbi_cov;            /* the tables' cover lists (hvk_rptrs_t.vbi_cov); vbi_cov_ok[u]: table u has one (no sample under more than HVK_VBI_COVER symbols) */
	int vbi_cov_ok[HVK_VBI_LUTS];
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
	int sv_hist;            /* samples of the stream kept in front of a batch */
	int64_t sv_tail_first, sv_tail_total, sv_tail_frames;   /* the batch whose sub-carrier stream lies in d_C2 (first frame; -1: none), its samples, its frames */
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
	hipStream_t copy_stream;    /* hvk_frame_copy(): pictures that come from another engine arrive on it, not behind this engine's renders */
	hipEvent_t copy_out_ev;     /* the last copy OUT of this engine's pool (another engine's hvk_frame_copy): an upload into the pool waits for it */
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

/* ... after the serial chains have moved on for a batch: the failure leaves the stream out of step for good */
#define HIPCHK_P(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
	e->poisoned = 1; return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)
#define HIPCHK(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
	return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)

/* first output sample of stream frame f (frames of two lengths with some --pixelrate pairs: hvk_tables.c) */
static inline int64_t _fstart(const hvk_engine *e, int64_t f) { return(hvk_tables_frame_start(&e->t, f)); }

/* across the files */
void hvk_e_fm_worker(hvk_engine *e);
void hvk_e_fm_wait_all(hvk_engine *e);
int hvk_e_fm_upto(hvk_engine *e, size_t upto);
int hvk_e_fm_finish(hvk_engine *e);
int hvk_e_prep_dirty(hvk_engine *e, const int32_t *slots, int n, hipStream_t stream);
int hvk_e_prep_staged(hvk_engine *e, int y0, int n, hipStream_t stream);
int hvk_e_carry_copy(hvk_engine *e);
int hvk_e_flush_planes(hvk_engine *e);
int hvk_e_secam_resolve(hvk_engine *e);       /* hvk_engine_stage.cpp */
int hvk_e_sv_ring_records(hvk_engine *e, int64_t first_frame, int nframes);  /* hvk_k_svq's per-line records of a batch being staged (hvk_engine_launch.cpp) */
int hvk_e_stage(hvk_engine_t *e, int64_t first_frame, int64_t stride, int nframes, const int32_t *slots, const int32_t *prev_slots);
void hvk_e_kernel_args(hvk_engine *e, hvk_raster_args_t *pra, hvk_filter_args_t *pfa, void *d_iq, int64_t out_stride);

#endif
