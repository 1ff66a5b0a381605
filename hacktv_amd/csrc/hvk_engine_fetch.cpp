/* hvk_engine_fetch.cpp -- the samples' way back: hvk_fetch / hvk_fetch_async / hvk_fetch_wait, the FM video phasor's thread
 * (src/video.c:2299-2335 is one recurrence over every sample: hvk_tail.c on the host), the file sink's sample formats
 * (hvk_fetch_as), hvk_fetch_raster, hvk_block_sums. */
#include "hvk_engine_priv.h"


/* FM video: bring the host copy of the current batch up to `upto` samples -- fetch the
 * modulator's input from the device and run the serial tail over it (hvk_tail.c) */
void hvk_e_fm_worker(hvk_engine *e)
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
void hvk_e_fm_wait_all(hvk_engine *e)
{
	if(!e->fm_thread) return;
	std::unique_lock<std::mutex> lk(*e->fm_mu);
	e->fm_cv->wait(lk, [e] { return(e->fm_q->empty()); });
}

int hvk_e_fm_upto(hvk_engine *e, size_t upto)
{
	hvk_e_fm_wait_all(e);
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
int hvk_e_fm_finish(hvk_engine *e)
{
	if(!e->fm_launched) return(HVK_OK);
	int r = hvk_e_fm_upto(e, (size_t) e->last_samples);
	if(r == HVK_OK) e->fm_launched = 0;
	return(r);
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
		int r = hvk_e_fm_upto(e, first + count);
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
		hvk_e_kernel_args(e, &ra, &fa, NULL, 1);
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
