/* hvk_kernels.h -- launch interface between the engine (host C++) and the
 * gfx950 kernels in hvk_kernels.hip. */
#ifndef HVK_KERNELS_H
#define HVK_KERNELS_H

#include <hip/hip_runtime_api.h>
#include "hvk_internal.h"

#define HVK_RUNMASK_OFFSET 256 /* bytes into the over-read samples' device buffer: [9 * lo + hi] the 16-bit element masks of a lane's samples lo .. hi - 1 */
#define HVK_CHROMA_LEAD 16   /* int16 of slack either side of a chroma channel in LDS */
#define HVK_NICAM_SYMS  48   /* symbol slots per filter tile */
#define HVK_NICAM_ROW   64   /* ints per tile row: the slots, then the mixer position */
#define HVK_MFMA_A_BYTES (2 * 64 * 16)
#define HVK_NICAM_TAPD  512  /* entries of one copy of the zero padded NICAM pulse table */
#ifndef HVK_NICAM_COPIES
/* copies of it, copy s shifted left by s entries: any run of 8 entries starts 8-byte aligned in one of four (two 8-byte reads
 * a lane), 16-byte aligned in one of eight (one 16-byte read, no bank conflict between the lanes of a wave -- and 4 KB more
 * LDS a workgroup: measured 1 % SLOWER on the metric configuration, profiles/r05_direct_variants.txt; kept as a build variant) */
#define HVK_NICAM_COPIES 4
#endif

/* FIR taps packed two int16 per dword, zero padded: passed by value so they
 * live in SGPRs (wave-uniform operands of v_dot2c_i32_i16) */
typedef struct {
	int p[(HVK_MAX_VF_TAPS + 1) / 2];
} hvk_packed_taps_t;

typedef struct {
	hvk_kconst_t k;
	hvk_packed_taps_t ctaps;
	hvk_packed_taps_t notch;    /* SECAM luma notch */
	const int16_t *chroma;      /* SECAM: [nframes][frame_samples]; raw baseband input: [nframes][slab_lines][width] */
	const int *vbi_sym;         /* VBI data lines: symbol index of every table */
	const int16_t *vbi_val;
	const int *vbi_cov;         /* [HVK_VBI_LUTS][width][16]: { first symbol over the sample | how many << 16, then their values there as int16 pairs }; NULL: no cover lists */
	const unsigned *vbi_ops;    /* [nframes][HVK_VBI_OPS][HVK_VBI_OPWORDS] */
	const signed char *vbi_map; /* [nframes][lines] */
	const int16_t *vits_l, *vits_c;
	const int16_t *fsc_rows;    /* field-sequential colour flag pulses */
	const int16_t *sis_dense, *sis_win, *sis_first;     /* sound-in-syncs tables */
	const unsigned *sis_bits;   /* [nframes][lines + 1 (+ 2 with the resampler)][2] */
	const hvk_linedesc_t *desc;
	const int16_t *pulses;
	const int16_t *linebase;    /* [nbase][k.base_stride] */
	const void *yuv;            /* 2^24 x int16x4 */
	const void *yuvparams;      /* hvk_yuvparams_t on the device */
	int levels_computed;        /* this block's pictures have many colours: compute the levels, do not look them up (the plane kernels: 1 the
	                             * reference's sequence of operations, 2 / 3 the forms hvk_yuvparams_t.fast = 1 / 2 name) */
	const hvk_c16_t *clut;
	const int16_t *burst_win;
	const int16_t *ghost;
	const uint32_t *pool;
	const hvk_framedesc_t *fdesc;   /* [nframes][1 + fields]: the frame before (its last line is this frame's leading halo), then the fields */
	int16_t *S;                 /* [nframes][lines + 2][width] */
	int16_t *C;                 /* --s-video: the sub-carrier, same geometry */
	int nframes;
	int secam_fid;              /* SECAM field identification lines are switched on */
	int64_t first_frame, frame_stride;
	/* Only some lines of every frame, each a row of its own: the lines the optional stages can write to, for the one-kernel
	 * render (hvk_direct.hip), which takes them instead of the picture planes' rows. NULL: the whole slab */
	const int16_t *linelist;    /* [nlist] 0-based line numbers; S is then [nframes][nlist][width] */
	int nlist;
} hvk_raster_args_t;

typedef struct {
	hvk_kconst_t k;
	hvk_packed_taps_t itaps, qtaps;
	const hvk_framedesc_t *fdesc;
	const int16_t *S;
	const int16_t *C;           /* --s-video: the Q channel, laid out like S */
	const hvk_c16_t *carriers;
	const int *tilesyms;        /* [nframes][tiles][HVK_NICAM_ROW] */
	const int *nicam_tapd;      /* HVK_NICAM_COPIES x HVK_NICAM_TAPD int16: the pulse, shifted copies, zero padded */
	const int *nicam_cca;       /* 2 x (nicam_cc_len + 8) dwords: the mixer's rows (cc.i, -cc.q), then (cc.q, cc.i) */
	const void *mfma_a;         /* HVK_MFMA_A_BYTES: the taps as MFMA A operand (hvk_engine.cpp:_mfma_taps), NULL: use the VALU filter */
	int mfma_ci, mfma_cq;       /* 128 * sum of the taps, per channel */
	int16_t *iq;
	int nframes;
	int64_t out_stride;         /* frame i goes to frame slot i * out_stride of iq */
} hvk_filter_args_t;

/* The picture planes and the one-kernel render from them (hvk_direct.hip) */
typedef struct {
	const int16_t *Lp;          /* [rows][width]: a line without its sub-carrier -- blanking, sync pulses, luma */
	const int *Cp;              /* [rows][width]: (V, U) after the chroma low pass, burst included: what the phasors are multiplied by */
	const int *clut3;           /* sub-carrier phasors (i, q), the same with i negated (PAL V switch), zeros: `creg` entries each */
	int creg;
	int zero_row;               /* a plane row of zeros: what lies before the stream's first sample */
	const hvk_linedesc_t *desc;
	const hvk_framedesc_t *fdesc;   /* [nframes][2]: the frame before, the frame */
	const uint32_t *lineoff;    /* [lines + 4]: ((j - 1) * width) mod clw -- the colour table position of line j - 1 of a frame that starts at position 0 */
	uint32_t inv_w;             /* ceil(2^32 / width): n / width == (n * inv_w) >> 32 for every n a frame's window positions take (checked by the host) */
	const int16_t *chroma;      /* SECAM: [nframes][raster_samples] the colour chain's sub-carrier (hvk_secam.hip), added to the frame's own lines */
	int chroma_zero;            /*   index of a run of width + 16 zeros in it: the lines around a frame carry none */
	/* Lines that the optional stages (VBI data lines, insertion test signals) can write to are rendered per frame by the
	 * raster kernel -- whole, sub-carrier and all -- into rows of their own behind the planes' */
	const int16_t *ovr_idx;     /* [lines] -1, or the line's place among a frame's such rows; NULL: none */
	int ovr_row0, ovr_n;        /* their first row in Lp; rows per frame */
} hvk_dptrs_t;

/* What hvk_k_direct needs to know of a tile's 1024 + 64 window positions and does not depend on the frame but for its parity
 * (hvk_engine.cpp:_tile_records): where the two line ends in the window lie, and per line -- the one window position 0 lies in
 * and the two behind it -- its number in the frame, whose picture's planes show it, its V switch, its share of the colour
 * table position */
typedef struct {
	int32_t b1;                 /* window position at which the second line begins (the third: b1 + width) */
	int32_t meta[3];            /* line of the frame | the frame before's (also: before the stream's first frame, zeros) << 16 | a line of the frame itself << 17 | (pal + 1) << 18 */
	int32_t lw[3];              /* line * width - (window position of the line's sample 0): a plane index less the picture's first row's */
	int32_t nws[3];             /* - (window position of the line's sample 0) */
	uint32_t off[3];            /* (line * width) mod clw */
	int32_t ovr[3];             /* the line's place among a frame's rows of the optional stages (hvk_dptrs_t.ovr_idx), -1: a line of the planes */
} __attribute__((aligned(64))) hvk_tilerec_t;

typedef struct {
	hvk_kconst_t k;
	hvk_dptrs_t D;
	const void *tilerec;        /* [2][tiles_pad] hvk_tilerec_t (HVK_TILEREC=0: NULL, the lines worked out by every wave) */
	int tiles_pad;
	const hvk_c16_t *carriers;
	const int *tilesyms;
	const int *nicam_tapd, *nicam_cca;
	const void *mfma_a;
	int mfma_ci, mfma_cq;
	int16_t *iq;
	int nframes;
	int64_t out_stride;
	int64_t first_frame, frame_stride;
} hvk_direct_args_t;

/* SECAM colour sub-carrier on the device (hvk_secam.hip) */
#define HVK_SECAM_WARMUP 12     /* lines walked before a task's own to find its entry state */
#define HVK_SECAM_ROUNDS 16     /* check / redo rounds before the batch goes through the host's chain */
/* What a line's walk had in hand in front of the line's last eight samples: the IIR pair, the FM phasor, those eight low-pass
 * outputs (packed) and the seven sums that the values behind the line are added to -- all that a walk of those eight samples from
 * another set of values behind the line needs (hvk_k_secam_redo), in one 80-byte read */
typedef struct { double ix, iy; int32_t f[4]; int32_t pi, pq, acc[7], pad[3]; } hvk_secam_mid_t;

typedef struct {
	hvk_secam_consts_t C;
	int lines, hline, fields, interlaced, active_left, active_width, active_lines, burst_left, burst_width;
	int ntasks;                 /* task slots per frame: the two fill slots, then the longer of the two parities' lists */
	int nframes, total;         /* frames of the batch, nframes * ntasks */
	int tpad;                   /* tasks the state / flag stores are laid out for (>= total) */
	int cpad;                   /* task rows the cell stores F and acc are laid out for */
	int ncells;                 /* frames of the batch whose cells are made in this stage (clist) */
	int K;
	int R, nruns;               /* tasks per lane of the chain kernel, ceil(total / R) */
	int levels_computed;        /* this batch's pictures have many colours: compute the levels, do not look them up */
	const void *yuvp;           /* hvk_yuvparams_t on the device */
	int64_t first_frame;
	int64_t raster_samples;
	const hvk_secam_task_t *tasks;      /* [2][ntasks] */
	const hvk_linedesc_t *desc;
	const hvk_framedesc_t *fdesc;
	const uint32_t *pool;
	const int *uvp;             /* the pictures' (U, V) plane (hvk_k_prep, SECAM): a line's pixels' levels at plane_row0 + line - 1; NULL: none */
	const void *yuv;
	const int16_t *fid_rows;    /* [2][W] */
	const hvk_secam_c32_t *lut;
	const hvk_secam_c16_t *bell;
	const void *lutb;           /* [65536] {lut[u].i, lut[u].q, bell[(u - 32768) & 0xFFFF] as one dword, 0} */
	const int16_t *burst_win;
	int16_t *F;                 /* [W / 8][cpad][8] */
	int32_t *acc;               /* [cpad][8] */
	/* The low-passed cells of a frame depend on its picture and on the parity of its number, not on where in the stream
	 * it stands: a picture that stays keeps its two sets of rows (the per-picture share of the work, like the picture
	 * planes of hvk_direct.hip); the walk from line to line is every frame's own. */
	const int *cbase;           /* [nframes] the row of F / acc at which the frame's tasks begin */
	const int *clist;           /* [ncells] the frames whose rows are made now */
	/* Where a warm-up starts from. Not from nothing when the picture has been here before: a line's entry state the last
	 * time this picture slot was shown with this frame parity is kept per row (written by every walk of a task of its
	 * own), and a picture that stays meets the same states again -- the derived entry state is then right after a line
	 * or two instead of eleven. A hint, never more: the check decides (hvk_k_secam_check). */
	hvk_secam_state_t *seed;    /* [3 cpad], NULL: warm-ups start from a state of nothing */
	const int *sbase;           /* [nframes] the frame's first row in it: per picture slot and frame number modulo 6 (the parity, and the
	                             * line's sub-carrier start phase, (frame * lines + line) mod 3) */
	const int *kf;              /* [nframes] warm-up lines of the tasks of a frame (new pictures: the full number; < 0: entry states from
	                             * the estimate kernel instead), NULL: K, or the estimates where there are any */
	/* New pictures: a line's entry state without walking the lines before it (hvk_k_secam_est). What a line hands on is
	 * the IIR's two doubles -- which forget their start within 450 samples -- and the values behind the line, which the
	 * FM loop's last steps replace: those need the phasor at the line's end, and the phasor's ANGLE is the sum of the
	 * steps' angles, each linear in the step's table index (src/video.c:2236). hvk_k_secam_cells leaves the sum of a
	 * line's indices from x1 on (acc[7]) and the IIR's output at W - 8 from a start of nothing (iya); the estimate kernel
	 * adds the line's head and its last seven samples under the entry state at hand, turns the angle into the values the
	 * loop leaves behind the line, and goes on to the next line -- a few hundred additions per line instead of a walk.
	 * An estimate, right in all but a few cases per ten thousand (tools/secam_est_probe.c): the check decides. */
	double *iya;                /* [cpad] */
	/* The table's entries are cos / sin rounded to integers: an entry's angle and length differ from the nominal ones by up to
	 * 0.7 / 2^31 -- noise over a line of many different indices, but a stretch of ONE colour takes one entry hundreds of
	 * times and the differences add up to several hundred units of the phasor. res[u] holds them (units of 2^-46: angle,
	 * relative length); hvk_k_secam_cells sums them where a lane's eight samples take the same entry (corr, per line) */
	const int16_t *res;         /* [65536][2] */
	int32_t *corr;              /* [cpad][2] */
	int16_t *est;               /* [tpad][16]: the values behind the line at a task's entry; those the valid task before it used */
	int half_slot[2];           /* per frame parity: the task slot at which the frame's second field begins (hvk_k_secam_redo_fields) */
	int x1;                     /* where a line's head ends: a multiple of 8, the entry state's influence on the indices is gone by then */
	int ES, EK;                 /* tasks per lane of the estimate kernel; lines it walks before them */
	double kap0, kap1;          /* a step's angle: kap0 + kap1 * index */
	hvk_secam_state_t *entry, *exit;    /* [tpad] */
	hvk_secam_state_t *carry;   /* the state the batch starts from; after hvk_launch_secam_carry(): the next batch's */
	int *flags;                 /* [tpad] */
	int *count;
	int16_t *chroma;            /* [nframes][raster_samples] */
	/* hvk_k_secam_walk<1>: a line's walk without a table read from HBM per sample. The FM step of index c is
	 * lround(INT32_MAX cos / sin(kap0 + kap1 c)) (src/video.c:2236): c = 128 kh + kl, the coarse phasors (already times
	 * INT32_MAX) from a 513-entry table that lives in LDS, the fine one by a polynomial in kap1 kl (|kl| <= 64: 7.7e-4 rad at
	 * most), the product rounded -- equal to the table's entry for every index of the deviation range: hvk_k_secam_check_walk
	 * tries them all when the engine is opened, and an engine for which one differs keeps the table. The bell filter's gain
	 * (two int16) moves by at most one unit from index to index: 32 indices a 16-byte block -- the values at the block's
	 * first index, then a bit per step up (q; i) and per step down (i) -- decoded by population counts, 14 KB in LDS
	 * instead of a 256 KB table in HBM. */
	const double *phc;          /* [513][2]: INT32_MAX * (cos, sin) of the angle of index 128 k - 32768 */
	const uint32_t *bellz;      /* [bell_blocks][4]: {gain i | gain q << 16, q steps up, i steps up, i steps down} */
	int bell_c0, bell_blocks;   /* the first index the blocks cover (a multiple of 32 below the deviation limits) */
	double ph_k1;               /* the angle of one index step */
	/* What a walk had in hand in front of its line's last chunk of 8 samples -- the IIR's two doubles and the FM phasor --, per
	 * task. A line that started from a state whose IIR half was right and whose values behind the line were not (the estimate's
	 * usual way of being wrong) differs from the true walk in those eight samples and the state it leaves alone: the redo
	 * goes on from here instead of walking the line again (hvk_k_secam_redo). */
	hvk_secam_mid_t *mid;       /* [tpad], NULL: none kept */
	/* The kept sub-carrier (hvk_engine_stage.cpp). What a frame's walk leaves -- every line's entry state (seed), the state behind
	 * its last line (seedx), its rows of the sub-carrier store -- is a function of the picture's cells, the frame's number modulo 6
	 * and the state the frame starts from. A picture that stays meets all three again: the frame then TAKES the kept rows and
	 * states (mflag), and what is left to do is the check that the state it starts from is the one the kept walk started from
	 * (hvk_k_secam_check, as for every line). orow: the row of `chroma` a frame's lines are written to -- its place in the batch,
	 * or, for the frame that (re)makes a kept set (owner), the set's row; only an owner writes seed / seedx. */
	const int *mflag;           /* [nframes] 1: the frame's lines are not walked, NULL: no frame's */
	const int *owner;           /* [nframes] 1: the frame's walk is the one kept for its picture and frame number modulo 6 */
	const int *orow;            /* [nframes] */
	const int *mrow;            /* [nframes] the set's index (slot * 6 + frame number modulo 6) */
	hvk_secam_state_t *seedx;   /* [sets] the state behind a kept frame's last task */
} hvk_secam_args_t;

#ifdef __cplusplus
extern "C" {
#endif

/* walk: 0 the general chain kernel (warm-up lines, runs of several tasks); 1 hvk_k_secam_walk<0> -- one line per lane from
 * an estimated or kept entry state, FM steps and gains from the table; 2 hvk_k_secam_walk<1> -- steps computed, gains from LDS */
int hvk_launch_secam_cells_chain(const hvk_secam_args_t *a, int estimate, int walk, hipStream_t stream);
int hvk_launch_secam_check_walk(const hvk_secam_args_t *a, int *differ, hipStream_t stream);
int hvk_launch_secam_check(const hvk_secam_args_t *a, hipStream_t stream);
int hvk_launch_secam_redo(const hvk_secam_args_t *a, int round, hipStream_t stream);
int hvk_launch_secam_carry(const hvk_secam_args_t *a, hipStream_t stream);
int hvk_launch_expand_yuv(void *lut, const void *params, hipStream_t stream);
int hvk_launch_check_levels(const void *lut, const void *params, int fast, int *differ, hipStream_t stream);
int hvk_launch_raster(const hvk_raster_args_t *a, hipStream_t stream);
int hvk_launch_filter(const hvk_filter_args_t *a, hipStream_t stream);
/* picture planes of the pictures in slots slot0 .. slot0 + npics - 1, all of one geometry: a kernel argument, nothing is
 * copied to the device for a launch (a slot's picture lies at slot * frame_px of the pool, its planes' rows at slot * lines) */
typedef struct {
	int32_t slot0;
	int32_t fb_width, fb_height, fb_interlaced, fb_valid;
	int64_t frame_px;
} hvk_prepgeo_t;
int hvk_launch_prep(const hvk_raster_args_t *a, const hvk_prepgeo_t *g, int npics, int16_t *Lp, int *Cp, hipStream_t stream);
int hvk_direct_supported(const hvk_kconst_t *k, const void *mfma_a, int secam_fid, int max_frames);
int hvk_launch_direct(const hvk_direct_args_t *a, hipStream_t stream);
/* the whole pipeline from the pixels in one kernel, for pictures that change (hvk_fused.hip); mfma_a28: the filter's A operand for a window that starts 28 samples before a tile */
int hvk_fused_supported(const hvk_kconst_t *k, const hvk_linedesc_t *desc);
int hvk_launch_fused(const hvk_raster_args_t *ra, const hvk_direct_args_t *a, const void *mfma_a28, hipStream_t stream);
int hvk_launch_resample(const hvk_kconst_t *k, const void *Sp, const void *taps, void *S2, int nframes, const void *frec, hipStream_t stream);
int hvk_launch_svq(const void *rec, int nlines, const void *C2, const void *Craster, void *Q, int s_lead, hipStream_t stream);
int hvk_launch_tail(void *iq, const void *off, const void *pass, int swap, long frame_samples, long out_stride,
                    int nframes, hipStream_t stream);
int hvk_launch_convert(const void *iq, size_t count, int type, int cplx, void *dst, hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
