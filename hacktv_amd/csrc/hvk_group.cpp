/* hvk_group.cpp -- ONE stream rendered by several engines, each on a device of its own (include/hacktv_amd.h,
 * "several devices"): BASELINE config 5's sharding as host C inside libhvk -- no Python, no torch.
 *
 * The reference renders line after line on one CPU (src/video.c:4867-4952) and hands every line to one sink
 * (src/hacktv.c:1579-1587 -> rf_write, src/rf.c:23-31). Here the stream is cut into blocks of `block_frames` frames and
 * block b goes to engine b mod N ("block-cyclic"). What crosses a block boundary:
 *   - the colour sub-carrier's table position, the frame parity, the NICAM schedule: closed forms of the frame number
 *     (hvk_stage_strided);
 *   - the serial sound chains (FM / AM phasors, limiter, NICAM framer; src/video.c:2259-2276, :3261-3450): handed from
 *     the engine of block b - 1 to the engine of block b IN PROCESS (hvk_sound_state_export / _import), so every engine
 *     runs them over its own frames only; the 32 kHz source samples are kept here and dealt to the engine whose block
 *     draws them;
 *   - on 525 lines the picture on the last line of the frame before a block's first (within the video filter's reach):
 *     the group uploads that picture into a slot of the next engine as well (hvk_stage_strided_prev).
 * SECAM colour and FM video are chains over every sample of the stream: one engine renders such
 * a stream and a group of more than one is refused for them.
 *
 * Reassembly of the contiguous stream, two shapes:
 *   (i)  host-direct: every engine's block is read back with hvk_fetch_async() straight into its place in the caller's
 *        page-locked stream buffer -- N devices use N PCIe links, the right shape for a sink that lives on the host
 *        (rf_file, rf_hackrf ...);
 *   (ii) gather on one device (north_star: "RCCL gather over xGMI"): hvk_group_gather() moves the blocks of a round into
 *        the root engine's device buffer -- grouped ncclSend / ncclRecv from C (librccl, loaded when first needed; one
 *        xGMI link per peer, never a ring), plain device-to-device copies between engines that share a device.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "hacktv_amd.h"

#define HIPCHK(call) do { hipError_t _e = (call); if(_e != hipSuccess) { \
	fprintf(stderr, "libhvk: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
	return(_e == hipErrorOutOfMemory ? HVK_OUT_OF_MEMORY : HVK_ERROR); } } while(0)

/* what is used of rccl.h (/opt/rocm/include/rccl/rccl.h:236, :260, :339, :459-461, :700, :722, :904), bound at run time:
 * librccl.so is half a gigabyte and only a gather between distinct devices needs it */
typedef struct ncclComm *ncclComm_t;
typedef struct {
	void *lib;
	int (*CommInitAll)(ncclComm_t *, int, const int *);
	int (*CommDestroy)(ncclComm_t);
	const char *(*GetErrorString)(int);
	int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t);
	int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t);
	int (*GroupStart)(void);
	int (*GroupEnd)(void);
	int (*GetVersion)(int *);   /* (not needed by the gather: hvk_rccl_probe() reports it) */
} rccl_t;
#define NCCL_INT32 2

/* how the blocks of a round reach the root device */
enum { GATHER_LOCAL = 0,        /* every engine on the root's device: device-to-device copies on the root's stream */
       GATHER_RCCL = 1,         /* distinct devices: grouped ncclSend / ncclRecv, one communicator per device */
       GATHER_PEER = 2 };       /* hipMemcpyPeerAsync, every sender pushing its block over its own link on its own stream */

struct hvk_group {
	int n, block;
	std::vector<hvk_engine_t *> eng;
	std::vector<int> dev;
	hvk_info_t info;
	int64_t next_frame, next_block;
	int staged;                 /* frames of the block staged and not launched yet */
	int has_sound;
	int needs_prev;             /* 525 lines: the frame before a block's first shows picture within the filter's reach */
	int prev_slot;              /* the slot of every engine that holds it */
	/* the 32 kHz source: pairs [src_base, src_base + src.size() / 2) */
	std::vector<int16_t> src;
	int64_t src_base;
	int64_t fed_to;             /* source position the current block's engine has been fed up to */
	int chains_taken;           /* the block being prepared has taken the chains over already */
	std::vector<uint8_t> state; /* the chains as exported after the last block staged */
	int have_state;
	std::vector<uint8_t> cstate;    /* SECAM: the colour chain's state behind the last block staged (hvk_secam_state_export) */
	int have_cstate;
	/* gather */
	rccl_t rccl;
	std::vector<ncclComm_t> comms;
	int distinct;               /* every engine on a device of its own */
	int mode;                   /* GATHER_* */
	int peer_ready;             /* peer access between the devices asked for (once) */
	std::vector<hipStream_t> gstream;
	std::vector<hipEvent_t> gev;
	char backend[96];
};

static int _rccl_load(rccl_t *out)
{
	if(out->lib) return(HVK_OK);
	void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if(!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if(!lib) { fprintf(stderr, "libhvk: librccl.so.1 not found (%s)\n", dlerror()); return(HVK_UNSUPPORTED); }
	rccl_t r;
	memset(&r, 0, sizeof(r));
	r.lib = lib;
	*(void **) &r.CommInitAll = dlsym(lib, "ncclCommInitAll");
	*(void **) &r.CommDestroy = dlsym(lib, "ncclCommDestroy");
	*(void **) &r.GetErrorString = dlsym(lib, "ncclGetErrorString");
	*(void **) &r.Send = dlsym(lib, "ncclSend");
	*(void **) &r.Recv = dlsym(lib, "ncclRecv");
	*(void **) &r.GroupStart = dlsym(lib, "ncclGroupStart");
	*(void **) &r.GroupEnd = dlsym(lib, "ncclGroupEnd");
	*(void **) &r.GetVersion = dlsym(lib, "ncclGetVersion");
	if(!r.CommInitAll || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
	{
		fprintf(stderr, "libhvk: librccl.so.1 lacks a symbol the gather needs\n");
		dlclose(lib);
		return(HVK_UNSUPPORTED);
	}
	*out = r;
	return(HVK_OK);
}

/* Can the RCCL reassembly be used at all on this machine? Loads librccl.so.1 the way the gather does, binds the seven
 * entry points it calls and writes "rccl <version code>" into msg. Needs no device (a CPU-only test calls it). */
extern "C" int hvk_rccl_probe(char *msg, size_t len)
{
	rccl_t r;
	memset(&r, 0, sizeof(r));
	int rc = _rccl_load(&r);
	if(rc != HVK_OK) { if(msg && len) snprintf(msg, len, "librccl.so.1 could not be loaded or lacks a symbol"); return(rc); }
	int v = 0;
	if(r.GetVersion) (void) r.GetVersion(&v);
	if(msg && len) snprintf(msg, len, "rccl %d: ncclCommInitAll ncclCommDestroy ncclGetErrorString ncclSend ncclRecv ncclGroupStart ncclGroupEnd bound", v);
	dlclose(r.lib);
	return(HVK_OK);
}

static void _name_backend(hvk_group *g)
{
	snprintf(g->backend, sizeof(g->backend), "%s", g->n == 1 ? "none (one engine)" :
		g->mode == GATHER_RCCL ? "rccl (grouped ncclSend / ncclRecv, one communicator per device)" :
		g->mode == GATHER_PEER ? "peer (hipMemcpyPeerAsync, every sender on its own stream)" : "hipMemcpyAsync (engines share a device)");
}

extern "C" int hvk_group_open(hvk_group_t **pg, const hvk_config_t *conf, unsigned int sample_rate, unsigned int pixel_rate,
                              const int *devices, int ndevices, int block_frames)
{
	if(!pg || !conf || !devices || ndevices < 1 || ndevices > 64 || block_frames < 1) return(HVK_ERROR);
	*pg = NULL;
	if(conf->struct_size != sizeof(hvk_config_t)) return(hvk_open_rates((hvk_engine_t **) pg, conf, sample_rate, pixel_rate, -1, 1));   /* (says why, returns HVK_ERROR) */
	hvk_group *g = new hvk_group();
	g->n = ndevices;
	g->block = block_frames;
	g->prev_slot = block_frames;
	g->distinct = 1;
	for(int i = 0; i < ndevices; i++) for(int j = 0; j < i; j++) if(devices[i] == devices[j]) g->distinct = 0;
	if(conf->interlace && ndevices > 1)
	{
		/* (two pictures per frame: the block's slot list, the upload call and the carried picture here all count frames; a group of
		 * ONE engine passes its blocks straight through -- hvk_group_stage()'s two-slots-per-frame list -- and takes it) */
		fprintf(stderr, "libhvk: refused: a group of several engines renders one picture per frame; --interlace (a picture per field, src/video.c:4873) goes through one engine\n");
		delete g;
		return(HVK_UNSUPPORTED);
	}
	for(int i = 0; i < ndevices; i++)
	{
		hvk_engine_t *e = NULL;
		/* one frame more than a block: the slot for the picture of the frame before the block's first */
		int r = hvk_open_rates(&e, conf, sample_rate, pixel_rate, devices[i], block_frames + 1);
		if(r != HVK_OK) { hvk_group_close(g); return(r); }
		g->eng.push_back(e);
		g->dev.push_back(devices[i]);
	}
	g->info.struct_size = (uint32_t) sizeof(g->info);
	hvk_get_info(g->eng[0], &g->info);
	g->has_sound = g->info.has_carriers || g->info.has_nicam || hvk_sound_state_size(g->eng[0]) > 0;
	g->needs_prev = hvk_last_line_shows_picture(g->eng[0]);
	if(ndevices > 1 && hvk_stream_is_one_chain(g->eng[0]))
	{
		fprintf(stderr, "libhvk: refused: this configuration is one serial chain over the stream (FM video, frames of two lengths, passthru, raw "
		                "baseband, or sound-in-syncs, whose burst encoder runs ahead of the sound chains): one engine renders it, a group of %d does not\n", ndevices);
		hvk_group_close(g);
		return(HVK_UNSUPPORTED);
	}
	g->state.resize(hvk_sound_state_size(g->eng[0]));
	g->cstate.resize(hvk_secam_state_size(g->eng[0]));
	/* HVK_GATHER=peer: hipMemcpyPeerAsync instead of RCCL (also between engines that share a device: the one-GPU test of
	 * that branch); HVK_GATHER=rccl (the default between distinct devices) falls back to peer copies when librccl cannot
	 * be loaded or its communicators cannot be made */
	const char *want = getenv("HVK_GATHER");
	g->mode = ndevices == 1 ? GATHER_LOCAL : (want && !strcmp(want, "peer")) ? GATHER_PEER : g->distinct ? GATHER_RCCL : GATHER_LOCAL;
	_name_backend(g);
	*pg = g;
	return(HVK_OK);
}

extern "C" void hvk_group_close(hvk_group_t *g)
{
	if(!g) return;
	for(size_t i = 0; i < g->comms.size(); i++) if(g->comms[i] && g->rccl.CommDestroy) g->rccl.CommDestroy(g->comms[i]);
	for(size_t i = 0; i < g->gstream.size(); i++)
	{
		(void) hipSetDevice(g->dev[i]);
		if(g->gstream[i]) { (void) hipStreamSynchronize(g->gstream[i]); (void) hipStreamDestroy(g->gstream[i]); }
		if(i < g->gev.size() && g->gev[i]) (void) hipEventDestroy(g->gev[i]);
	}
	for(hvk_engine_t *e : g->eng) hvk_close(e);
	if(g->rccl.lib) dlclose(g->rccl.lib);
	delete g;
}

extern "C" int hvk_group_size(const hvk_group_t *g) { return(g ? g->n : 0); }
extern "C" int hvk_group_block_frames(const hvk_group_t *g) { return(g ? g->block : 0); }
extern "C" hvk_engine_t *hvk_group_engine(hvk_group_t *g, int i) { return(g && i >= 0 && i < g->n ? g->eng[i] : NULL); }
extern "C" int64_t hvk_group_next_frame(const hvk_group_t *g) { return(g ? g->next_frame : -1); }
extern "C" int hvk_group_block_index(const hvk_group_t *g) { return(g ? (int) (g->next_block % g->n) : -1); }
extern "C" hvk_engine_t *hvk_group_block_engine(hvk_group_t *g) { return(g ? g->eng[(size_t) (g->next_block % g->n)] : NULL); }
extern "C" const char *hvk_group_gather_backend(const hvk_group_t *g) { return(g ? g->backend : ""); }

/* the engine of the block being prepared takes the sound chains over from the engine of the block before (once) */
static int _take_chains(hvk_group *g)
{
	if(g->chains_taken) return(HVK_OK);
	hvk_engine_t *e = hvk_group_block_engine(g);
	g->fed_to = g->src_base;
	if(g->n > 1 && g->has_sound && g->have_state)
	{
		int64_t pos = 0;
		int r = hvk_sound_state_import(e, g->state.data(), g->state.size(), &pos);
		if(r != HVK_OK) return(r);
		/* what every engine is past can go: the source queue starts where this block's chains go on */
		if(pos > g->src_base)
		{
			const int64_t drop = std::min<int64_t>(pos - g->src_base, (int64_t) (g->src.size() / 2));
			g->src.erase(g->src.begin(), g->src.begin() + drop * 2);
			g->src_base += drop;
		}
		if(pos < g->src_base) return(HVK_ERROR);     /* (cannot happen: nothing behind the chains' position is dropped) */
		/* an engine whose queue still holds `pos` keeps what it was dealt behind it the last time round (hvk_audio.c:
		 * hvk_audio_state_import): it goes on from the END of what it holds, not from pos -- or it would get those pairs twice */
		g->fed_to = std::max<int64_t>(pos, hvk_sound_source_end(e));
	}
	g->chains_taken = 1;
	return(HVK_OK);
}

/* deal the engine of the block being prepared what the queue holds beyond what it has been fed */
static int _deal(hvk_group *g)
{
	hvk_engine_t *e = hvk_group_block_engine(g);
	const int64_t have = g->src_base + (int64_t) (g->src.size() / 2);
	if(have > g->fed_to)
	{
		int r = hvk_audio_write(e, g->src.data() + (size_t) (g->fed_to - g->src_base) * 2, (size_t) (have - g->fed_to));
		if(r != HVK_OK) return(r);
		g->fed_to = have;
	}
	return(HVK_OK);
}

extern "C" int hvk_group_audio_write(hvk_group_t *g, const int16_t *stereo, size_t nsamples)
{
	if(!g || (!stereo && nsamples)) return(HVK_ERROR);
	if(g->n == 1) return(hvk_audio_write(g->eng[0], stereo, nsamples));
	g->src.insert(g->src.end(), stereo, stereo + nsamples * 2);
	return(HVK_OK);
}

/* source pairs still missing before a block of `nframes` frames can be staged */
extern "C" size_t hvk_group_audio_needed(hvk_group_t *g, int nframes)
{
	if(!g || !g->has_sound) return(0);
	hvk_engine_t *e = hvk_group_block_engine(g);
	if(g->n > 1 && (_take_chains(g) != HVK_OK || _deal(g) != HVK_OK)) return(0);
	/* (engines of a group are staged by frame number: hvk_audio_needed() counts from frame 0) */
	return(hvk_audio_needed(e, (int) (g->next_frame + nframes)));
}

extern "C" int hvk_group_frame_upload(hvk_group_t *g, int frame_in_block, const uint32_t *fb, int width, int height,
                                      int pixel_stride, int line_stride, int interlaced)
{
	if(!g || frame_in_block < 0 || frame_in_block >= g->block) return(HVK_ERROR);
	return(hvk_frame_upload(hvk_group_block_engine(g), frame_in_block, fb, width, height, pixel_stride, line_stride, interlaced));
}

/* Stage the next block: nframes <= block_frames frames from the stream's next frame on, on engine (block mod N); `slots`
 * (may be NULL: frame i shows slot i) as in hvk_stage_strided(). */
extern "C" int hvk_group_stage(hvk_group_t *g, int nframes, const int32_t *slots)
{
	if(!g || nframes < 1 || nframes > g->block || g->staged) return(HVK_ERROR);
	hvk_engine_t *e = hvk_group_block_engine(g);
	int r;

	if(g->n == 1)
	{
		std::vector<int32_t> id((size_t) nframes * 2);
		for(size_t i = 0; i < id.size(); i++) id[i] = (int32_t) i;
		r = hvk_stage_strided(e, g->next_frame, 1, nframes, slots ? slots : id.data());
		if(r != HVK_OK) return(r);
		g->staged = nframes;
		return(HVK_OK);
	}

	if((r = _take_chains(g)) != HVK_OK) return(r);
	if(g->has_sound && (r = _deal(g)) != HVK_OK) return(r);

	/* SECAM: the colour chain goes on from where the engine of the block before left it */
	if(!g->cstate.empty() && g->have_cstate && (r = hvk_secam_state_import(e, g->cstate.data(), g->cstate.size())) != HVK_OK) return(r);

	std::vector<int32_t> id((size_t) nframes), prev((size_t) nframes, -1);
	for(int i = 0; i < nframes; i++) id[i] = i;
	if(g->needs_prev && g->next_frame > 0) prev[0] = g->prev_slot;     /* (uploaded there when the block before was staged) */
	r = hvk_stage_strided_prev(e, g->next_frame, 1, nframes, slots ? slots : id.data(), prev.data());
	if(r != HVK_OK) return(r);

	if(g->has_sound)
	{
		if((r = hvk_sound_state_export(e, g->state.data(), g->state.size())) != HVK_OK) return(r);
		g->have_state = 1;
	}
	if(!g->cstate.empty())
	{
		if((r = hvk_secam_state_export(e, g->cstate.data(), g->cstate.size())) != HVK_OK) return(r);
		g->have_cstate = 1;
	}
	if(g->needs_prev)
	{
		/* the picture this block's LAST frame shows -- the slot named for it, whenever and in whatever order it was
		 * uploaded -- for the engine of the block behind this one: device to device, into the slot kept for it */
		hvk_engine_t *nx = g->eng[(size_t) ((g->next_block + 1) % g->n)];
		r = hvk_frame_copy(nx, g->prev_slot, e, slots ? slots[nframes - 1] : nframes - 1);
		if(r != HVK_OK) return(r);
	}
	g->staged = nframes;
	return(HVK_OK);
}

/* Launch the staged block on its engine (d_iq: a buffer on THAT engine's device, NULL: the engine's own) and move on to
 * the next block. Returns the index of the engine that renders it (>= 0), or an error. */
extern "C" int hvk_group_launch(hvk_group_t *g, void *d_iq)
{
	if(!g || !g->staged) return(HVK_ERROR);
	const int idx = (int) (g->next_block % g->n);
	int r = hvk_launch(g->eng[idx], d_iq);
	if(r != HVK_OK) return(r);
	g->next_frame += g->staged;
	g->next_block++;
	g->staged = 0;
	g->chains_taken = 0;
	return(idx);
}

/* Reassembly (ii): the blocks of the engines [0, n) -- each in its engine's own output buffer, `samples` I/Q pairs --
 * into the root engine's device memory, engine i's block at d_root + i * samples pairs. Queued behind each engine's
 * render; *the root engine's stream* is where the gathered round is complete (hvk_sync(root) waits for it). */
extern "C" int hvk_group_gather(hvk_group_t *g, int root, void *d_root, size_t samples)
{
	if(!g || root < 0 || root >= g->n || !d_root) return(HVK_ERROR);
	if(g->gstream.empty())
	{
		g->gstream.resize(g->n, NULL);
		g->gev.resize(g->n, NULL);
		for(int i = 0; i < g->n; i++)
		{
			HIPCHK(hipSetDevice(g->dev[i]));
			HIPCHK(hipStreamCreateWithFlags(&g->gstream[i], hipStreamNonBlocking));
			HIPCHK(hipEventCreateWithFlags(&g->gev[i], hipEventDisableTiming));
		}
	}
	if(g->mode == GATHER_RCCL && g->comms.empty())
	{
		int r = _rccl_load(&g->rccl);
		if(r == HVK_OK)
		{
			g->comms.resize(g->n, NULL);
			int e = g->rccl.CommInitAll(g->comms.data(), g->n, g->dev.data());
			if(e != 0)
			{
				fprintf(stderr, "libhvk: ncclCommInitAll: %s -- the gather goes by hipMemcpyPeerAsync instead\n", g->rccl.GetErrorString(e));
				g->comms.clear();
				r = HVK_ERROR;
			}
		}
		if(r != HVK_OK) { g->mode = GATHER_PEER; _name_backend(g); }
	}
	if(g->mode == GATHER_PEER && !g->peer_ready)
	{
		/* direct access both ways where the devices allow it (a copy between devices without it is staged by the runtime) */
		for(int i = 0; i < g->n; i++) for(int j = 0; j < g->n; j++)
		{
			int can = 0;
			if(g->dev[i] == g->dev[j] || hipDeviceCanAccessPeer(&can, g->dev[i], g->dev[j]) != hipSuccess || !can) continue;
			HIPCHK(hipSetDevice(g->dev[i]));
			hipError_t pe = hipDeviceEnablePeerAccess(g->dev[j], 0);
			if(pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); }
		}
		(void) hipGetLastError();
		g->peer_ready = 1;
	}

	/* the transfers run on streams of their own, behind each engine's render (an event on the engine's stream) */
	for(int i = 0; i < g->n; i++)
	{
		HIPCHK(hipSetDevice(g->dev[i]));
		void *es = hvk_engine_stream(g->eng[i]);
		HIPCHK(hipEventRecord(g->gev[i], (hipStream_t) es));
		HIPCHK(hipStreamWaitEvent(g->gstream[i], g->gev[i], 0));
		if(i != root) { HIPCHK(hipSetDevice(g->dev[root])); HIPCHK(hipStreamWaitEvent(g->gstream[root], g->gev[i], 0)); }
	}
	char *dst = (char *) d_root;
	const size_t bytes = samples * 4;
	if(g->mode == GATHER_RCCL)
	{
		int e = g->rccl.GroupStart();
		for(int i = 0; i < g->n && e == 0; i++)
		{
			if(i == root) continue;
			e = g->rccl.Send(hvk_output_device_ptr(g->eng[i]), samples, NCCL_INT32, root, g->comms[i], g->gstream[i]);
			if(e == 0) e = g->rccl.Recv(dst + (size_t) i * bytes, samples, NCCL_INT32, i, g->comms[root], g->gstream[root]);
		}
		const int e2 = g->rccl.GroupEnd();
		if(e != 0 || e2 != 0) { fprintf(stderr, "libhvk: rccl gather: %s\n", g->rccl.GetErrorString(e ? e : e2)); return(HVK_ERROR); }
	}
	else if(g->mode == GATHER_PEER)
	{
		/* every sender pushes its block on ITS stream (N - 1 links at once); the root's stream then waits for each of them */
		for(int i = 0; i < g->n; i++)
		{
			if(i == root) continue;
			HIPCHK(hipSetDevice(g->dev[i]));
			HIPCHK(hipMemcpyPeerAsync(dst + (size_t) i * bytes, g->dev[root], hvk_output_device_ptr(g->eng[i]), g->dev[i], bytes, g->gstream[i]));
			HIPCHK(hipEventRecord(g->gev[i], g->gstream[i]));
			HIPCHK(hipSetDevice(g->dev[root]));
			HIPCHK(hipStreamWaitEvent(g->gstream[root], g->gev[i], 0));
		}
	}
	else
	{
		for(int i = 0; i < g->n; i++)
		{
			if(i == root) continue;
			HIPCHK(hipSetDevice(g->dev[root]));
			if(g->dev[i] == g->dev[root])
			{
				HIPCHK(hipMemcpyAsync(dst + (size_t) i * bytes, hvk_output_device_ptr(g->eng[i]), bytes, hipMemcpyDeviceToDevice, g->gstream[root]));
			}
			else
			{
				HIPCHK(hipMemcpyPeerAsync(dst + (size_t) i * bytes, g->dev[root], hvk_output_device_ptr(g->eng[i]), g->dev[i], bytes, g->gstream[root]));
			}
		}
	}
	/* the root's own block, and the root engine's stream behind all of it */
	HIPCHK(hipSetDevice(g->dev[root]));
	if((char *) hvk_output_device_ptr(g->eng[root]) != dst + (size_t) root * bytes)
	{
		HIPCHK(hipMemcpyAsync(dst + (size_t) root * bytes, hvk_output_device_ptr(g->eng[root]), bytes, hipMemcpyDeviceToDevice, g->gstream[root]));
	}
	HIPCHK(hipEventRecord(g->gev[root], g->gstream[root]));
	HIPCHK(hipStreamWaitEvent((hipStream_t) hvk_engine_stream(g->eng[root]), g->gev[root], 0));
	/* (a sender's buffer is free again when its block has left it: the next launch on that engine waits for the sender's
	 * own gather stream -- RCCL -- or for the root's, which did the copying) */
	for(int i = 0; i < g->n; i++)
	{
		if(i == root) continue;
		HIPCHK(hipSetDevice(g->dev[i]));
		if(g->mode == GATHER_RCCL) HIPCHK(hipEventRecord(g->gev[i], g->gstream[i]));
		/* (peer: gev[i] was recorded behind the sender's copy; local: the root's stream did the copying) */
		HIPCHK(hipStreamWaitEvent((hipStream_t) hvk_engine_stream(g->eng[i]), g->gev[g->mode == GATHER_LOCAL ? root : i], 0));
	}
	return(HVK_OK);
}
