#!/usr/bin/env python3
"""tools/bench_multi.py -- bench.py's N > 1 paths.

headline_group()  `python bench.py --gpus N` as typed: ONE process drives one engine per device through hvk_group_* (host
                  code in C inside libhvk, no torch): blocks of F frames dealt round-robin, the serial sound chains handed
                  from engine to engine in process, a step = every engine's launch + hvk_group_gather of the round onto
                  the root device (grouped ncclSend / ncclRecv from C between distinct devices; device copies between
                  engines that share a device, which is how --devices 0,0 tests the path on one GPU).
headline_ranks()  under `python -m torch.distributed.run --nproc-per-node N` (what the driver launches): one rank per GPU,
                  block b of the stream on rank b mod N, each step ends with the RCCL gather (grouped send / recv over
                  xGMI) that reassembles the contiguous IQ stream on rank 0 for the rf_* sink (src/hacktv.c:1579-1587);
                  K steps between barrier + synchronize, the maximum over ranks. Beside it rank 0 runs headline_group()
                  in a child process with a time limit (a collective that has never met the machine must not be able to
                  hang the measurement) and reports its figures as scalars.

Both gate before timing: the reassembled stream of the first round(s) -- block seams included -- sha256 == the unmodified
reference CLI run in the same job."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from bench import (HBM_PEAK_GBS, BYTES_PER_SAMPLE, SAMPLE_RATE, MODE, METRIC, ref_stream_sha, committed_digest,  # noqa: E402
                   hbm_roofline, traffic_per_launch, clean_env)

FLAT_NOTE = ("with sound a run that stages every round is bound by the host's serial FM chain (one recurrence over every sample, "
             "src/video.c:2259-2276): flat in N by construction")


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipSetDevice.argtypes = [ctypes.c_int]
    return hip


def headline_group(args, devices, log):
    import numpy as np
    import hacktv_amd as H
    import util

    if os.environ.get("BENCH_FAIL_C_GROUP"):      # (tests/test_gpu_block.py: what a failure of this part leaves of the ranks' line)
        raise SystemExit("asked to fail (BENCH_FAIL_C_GROUP)")
    N, F, noaudio = len(devices), args.frames, args.noaudio
    g = util.Golden()
    if len(set(devices)) > 1 and os.environ.get("HVK_GATHER") is None:
        # the collective branch may never have met this machine: a small round in a process of its own, with a time limit, first
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gather_probe.py"), ",".join(str(d) for d in devices)],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180, env=clean_env())
            ok = pr.returncode == 0 and "BACKEND" in pr.stdout
            log("gather probe: " + (pr.stdout.strip().splitlines()[-1] if pr.stdout.strip() else "no output") + ("" if ok else " | " + pr.stderr[-300:]))
        except subprocess.TimeoutExpired:
            ok = False
            log("gather probe: no answer within 180 s")
        if not ok:
            os.environ["HVK_GATHER"] = "peer"
            log("the gather goes by hipMemcpyPeerAsync (HVK_GATHER=peer)")
    conf = H.preset(MODE, H.FLAG_FILTER | (H.FLAG_NOAUDIO if noaudio else 0))
    grp = H.Group(conf, SAMPLE_RATE, devices, F)
    FS = grp.info["frame_samples"]
    hip = _hip()
    hip.hipSetDevice(devices[0])
    root = ctypes.c_void_p()
    if hip.hipMalloc(ctypes.byref(root), N * F * FS * 4) != 0:
        raise SystemExit("no room for the gathered round on the root device")
    for e in grp.engines:
        e.frame_upload(0, g.frame("i_full"))

    def stage_round():
        for _ in range(N):
            if not noaudio:
                while grp.audio_needed(F) > 0:
                    grp.audio_write(g.audio)
            grp.stage(F, slots=[0] * F)
            grp.launch()
        grp.gather(0, root, F * FS)

    def sync_all():
        for e in grp.engines:
            e.sync()

    # ---- round 0 through the group's own stage / launch calls (sound chains handed on), gathered, hashed ----
    t0 = time.perf_counter()
    stage_round()
    sync_all()
    t_round0 = time.perf_counter() - t0
    backend = grp.gather_backend()
    host = np.zeros((N * F * FS, 2), np.int16)
    hip.hipSetDevice(devices[0])
    if hip.hipMemcpy(host.ctypes.data, root, N * F * FS * 4, 2) != 0:
        raise SystemExit("read-back of the gathered round failed")
    got = hashlib.sha256(host.tobytes()).hexdigest()
    del host
    cli = ["--filter"] + (["--noaudio"] if noaudio else [])
    want, how = ref_stream_sha(MODE, SAMPLE_RATE, cli, 0, N * F, FS * 4), None
    if want is not None:
        how = "hacktv_ref run in this job"
        if got != want:
            raise SystemExit("parity gate failed: %d engines x %d frames gathered on the root device differ from the reference CLI's output" % (N, F))
    com = None if noaudio else committed_digest(N * F)
    if com is not None:
        if got != com:
            raise SystemExit("parity gate failed: the first %d frames differ from the committed reference digest" % (N * F))
        how = (how + " + committed digest") if how else "committed digest"
    if how is None:
        raise SystemExit("parity gate: neither oracle/_ref/hacktv_ref nor a committed digest for %d frames -- refusing to report a number" % (N * F))
    gate = "round 0: %d frames over %d engines (sound chains handed on), gathered on device %d: sha256 == %s" % (N * F, N, devices[0], how)
    log("parity gate ok: " + gate + " [" + backend + "]")

    def one(gather=True):
        for e in grp.engines:
            e.launch()
        if gather:
            grp.gather(0, root, F * FS)

    for _ in range(args.warmup):
        one()
    sync_all()
    t_set = time.perf_counter()
    while time.perf_counter() - t_set < args.settle:
        for _ in range(5):
            one()
        sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    sync_all()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one(False)
    sync_all()
    dt_render = time.perf_counter() - t0
    # the kernel's own time on engine 0 (HIP events on its launch stream)
    e0 = grp.engines[0]
    e0.timing_enable(True)
    for _ in range(args.steps):
        one(False)
    sync_all()
    kern_ms, n_k = e0.timing_read(1)
    e0.timing_enable(False)
    # host-direct reassembly: every engine's block read back into its place in one page-locked stream buffer (N PCIe links)
    hb = e0.host_buffer(N * F * FS)
    k_hd = max(2, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(k_hd):
        one(False)
        tk = [(e, e.fetch_async(hb[i * F * FS:(i + 1) * F * FS], 0, F * FS)) for i, e in enumerate(grp.engines)]
        for e, t in tk:
            e.fetch_wait(t)
    dt_host = (time.perf_counter() - t0) / k_hd
    # every round staged anew (the next N blocks of the stream: sound chains, host pre-pass, H2D), launched and gathered
    k_st = 2
    t0 = time.perf_counter()
    for _ in range(k_st):
        stage_round()
    sync_all()
    dt_staged = (time.perf_counter() - t0) / k_st

    samples = N * F * FS
    value = samples * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    names = e0.kernel_names()
    roof = hbm_roofline(names[-1], kern_ms, n_k, F * FS, traffic_per_launch(F))
    roof["path_frac"] = round(BYTES_PER_SAMPLE * samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / N, 4)
    res = {
        "metric": METRIC, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16 data, int32 accumulate",
        "data": "synthetic: built-in test card + 1 kHz tone; a step = every engine renders its staged block of F frames again + the gather "
                "of the round onto the root device; side inputs resident in HBM",
        "config": {"workload": "-m i -s 16000000 --filter test%s (PAL-I AM-VSB + 51-tap FIR, FM mono + NICAM)" % (" --noaudio" if noaudio else ""),
                   "frames_per_gpu_per_step": F, "samples_per_step": samples,
                   "parallelism": "blocks of %d frames round-robin over %d engines (devices %s), one process (hvk_group_*), gathered on the root device: %s"
                                  % (F, N, ",".join(str(d) for d in devices), backend)},
        "parity_gate": gate,
        "roofline": roof,
        "multi_gpu": {
            "mode": "one process, hvk_group_* (C)", "engines": N, "devices": ",".join(str(d) for d in devices), "gather_backend": backend,
            "gathered_Msamples_per_s": round(value, 1),
            "render_only_Msamples_per_s": round(samples * args.steps / dt_render / 1e6, 1),
            "host_direct_Msamples_per_s": round(samples / dt_host / 1e6, 1),
            "staged_every_round_Msamples_per_s": round(samples / dt_staged / 1e6, 1),
            "first_round_staged_s": round(t_round0, 3),
            "note": "value = launch + gather (inputs resident), bound by the root's ingest; " + ("--noaudio: no serial chain" if noaudio else FLAT_NOTE),
        },
    }
    log("value %.1f Msamples/s over %d engines [%s]; render only %.1f, host-direct %.1f, staged every round %.1f" %
        (value, N, backend, res["multi_gpu"]["render_only_Msamples_per_s"], res["multi_gpu"]["host_direct_Msamples_per_s"],
         res["multi_gpu"]["staged_every_round_Msamples_per_s"]))
    hip.hipSetDevice(devices[0])
    hip.hipFree(root)
    grp.close()
    return res


def headline_ranks(args, log):
    import datetime
    import numpy as np   # noqa: F401
    import torch
    import torch.distributed as dist
    import hacktv_amd as H
    from hacktv_amd import sharding
    import util

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    dry = args.dry_run_backend is not None
    if dry:
        local_rank = 0                      # every rank shares GPU 0; transport through host memory
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    long_wait = datetime.timedelta(seconds=max(900.0, 2 * args.group_timeout))
    if dry:
        dist.init_process_group(args.dry_run_backend, timeout=long_wait)
    else:
        dist.init_process_group("nccl", device_id=dev, timeout=long_wait)
    # the sound chains' state travels between the ranks' hosts, in a group of its own: its messages must not queue up
    # between the blocks of the gather (a rank hands the chains on BEFORE it renders and sends its block)
    hostg = dist.new_group(backend="gloo", timeout=long_wait)

    g = util.Golden()
    noaudio = args.noaudio
    conf = H.preset(MODE, H.FLAG_FILTER | (H.FLAG_NOAUDIO if noaudio else 0))
    F = args.frames
    e = H.Engine(conf, SAMPLE_RATE, device=local_rank, max_frames=F)
    FS = e.info["frame_samples"]
    # A stream of our own, made torch's current one: the engine launches on it, and RCCL's point-to-point operations order
    # themselves behind torch's CURRENT stream -- the send of a block has to wait for the render just enqueued there.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    e.set_stream(ctypes.c_void_p(stream.cuda_stream))
    e.frame_upload(0, g.frame("i_full"))
    gather = not args.no_gather
    cli = ["--filter"] + (["--noaudio"] if noaudio else [])

    def ref_sha(first, count):
        return ref_stream_sha(MODE, SAMPLE_RATE, cli, first, count, FS * 4)

    def feed_audio(upto_frame, source_pos=None):
        if source_pos is not None:
            e.audio_write(g.audio[source_pos % len(g.audio):])
        while e.audio_needed(upto_frame) > 0:
            e.audio_write(g.audio)

    def stage_block(block, Fb, last=False):
        """Stage block `block` (Fb frames) on the rank it belongs to: take the sound chains over from the rank that staged the
        block before, run them over this block's frames only, hand them on."""
        first = block * Fb
        pos = None if noaudio else sharding.sound_state_recv(e, N, block, hostg)
        if not noaudio:
            feed_audio(first + Fb, pos)
        e.stage(first, 1, Fb, prev_slots=[0] * Fb)
        if not noaudio:
            sharding.sound_state_send(e, N, block, hostg, last=last)

    # ---- the sharded path end to end on short blocks, BEFORE anything is timed: two rounds of 2-frame blocks through the
    # same calls as the timed loop, rank 0 hashes the reassembled stream -- block seams and round seams included ----
    seam_gate = None
    if not noaudio:
        Fg, rounds = min(2, F), 2
        bufs = [torch.empty((Fg * FS * 2,), dtype=torch.int16, device=dev) for _ in range(2)]
        roots = [torch.empty((N, Fg * FS * 2), dtype=torch.int16, device=dev) for _ in range(2)] if rank == 0 else [None, None]
        host, works = [], []
        for rnd in range(rounds + 1):
            if rnd < rounds:
                stage_block(sharding.block_of(rank, N, rnd), Fg, last=(rnd == rounds - 1 and rank == N - 1))
                e.launch(ctypes.c_void_p(bufs[rnd & 1].data_ptr()))
                torch.cuda.synchronize()
            if rnd > 0:
                if dry:
                    sharding.gather_blocks(bufs[(rnd - 1) & 1], roots[(rnd - 1) & 1], rank, N, via_host=True)
                else:
                    sharding.gather_wait(works)
                if rank == 0:
                    host.append(roots[(rnd - 1) & 1].cpu().numpy().tobytes())
            if rnd < rounds and not dry:
                works = sharding.gather_start(bufs[rnd & 1], roots[rnd & 1], rank, N)
        if rank == 0:
            got = hashlib.sha256(b"".join(host)).hexdigest()
            k = rounds * N * Fg
            want = ref_sha(0, k)
            if want is None:
                cum = g.cases["i_full"]["sha256_cumulative"]
                want = cum[k - 1] if k <= len(cum) else None
            if want is None:
                raise SystemExit("seam gate: no reference to compare %d frames with -- refusing to report a number" % k)
            if got != want:
                raise SystemExit("seam gate failed: the stream reassembled from %d ranks x %d rounds differs from the reference CLI's output" % (N, rounds))
            seam_gate = "%d rounds x %d ranks x %d frames reassembled on rank 0: sha256 == reference CLI" % (rounds, N, Fg)
            log("seam gate ok: " + seam_gate)
        dist.barrier()

    # ---- stage the side inputs of this rank's block (untimed: inputs resident in HBM) ----
    first_frame = sharding.first_frame_of(rank, N, 0, F)   # block-cyclic: block b -> rank b mod N; round 0
    e.close()
    e = H.Engine(conf, SAMPLE_RATE, device=local_rank, max_frames=F)     # (a fresh stream position for the audio pre-pass)
    e.set_stream(ctypes.c_void_p(stream.cuda_stream))
    e.frame_upload(0, g.frame("i_full"))
    t0 = time.perf_counter()
    stage_block(sharding.block_of(rank, N, 0), F, last=(rank == N - 1 and not args.walk_rounds))
    e.sync()
    t_stage = time.perf_counter() - t0
    log("rank 0 staged %d frames (host control path + H2D) in %.2f s = %.1f Msamples/s" % (F, t_stage, F * FS / t_stage / 1e6))

    # two output buffers per rank and two stream buffers on the root: round s is sent while round s + 1 is rendered
    nbuf = 2 if gather else 1
    if rank == 0 and gather:
        outs = [torch.empty((N, F * FS * 2), dtype=torch.int16, device=dev) for _ in range(nbuf)]   # the contiguous stream, block after block
        mines = [o[0] for o in outs]
    else:
        outs = [None] * nbuf
        mines = [torch.empty((F * FS * 2,), dtype=torch.int16, device=dev) for _ in range(nbuf)]
    pending = []
    walk = {"round": 0, "last_round": None}

    def step(i=0):
        """Render this rank's block into buffer i & 1 while the block rendered before travels to rank 0. --walk-rounds:
        every step is the NEXT round's block -- sound chains from the rank before, host pre-pass, H2D, then the render."""
        b = i % nbuf
        if args.walk_rounds and walk["round"] > 0:
            stage_block(sharding.block_of(rank, N, walk["round"]), F, last=(walk["round"] == walk["last_round"] and rank == N - 1))
        if args.walk_rounds:
            walk["round"] += 1
        e.launch(ctypes.c_void_p(mines[b].data_ptr()))
        if gather:
            if dry:
                torch.cuda.synchronize()
                sharding.gather_blocks(mines[b], outs[b], rank, N, via_host=True)
            else:
                sharding.gather_wait(pending)       # the block before this one has arrived: its buffers are free again
                pending[:] = sharding.gather_start(mines[b], outs[b], rank, N)

    def drain():
        if gather and not dry:
            sharding.gather_wait(pending)
            pending[:] = []

    # ---- parity gate before any number: EVERY sample of this rank's block against the unmodified reference ----
    walk["last_round"] = args.warmup + args.steps if args.walk_rounds else 0
    step(0)
    drain()
    torch.cuda.synchronize()
    if not noaudio:
        mine_sha = hashlib.sha256(mines[0].cpu().numpy().tobytes()).hexdigest()
        want, how = ref_sha(first_frame, F), None
        if want is not None:
            how = "hacktv_ref run in this job"
            if mine_sha != want:
                raise SystemExit("parity gate failed on rank %d: frames %d..%d differ from the reference CLI's output" % (rank, first_frame, first_frame + F - 1))
        com = committed_digest(F) if first_frame == 0 else None
        if com is not None:
            if mine_sha != com:
                raise SystemExit("parity gate failed: the first %d frames differ from the committed reference digest" % F)
            how = (how + " + committed digest") if how else "committed digest"
        if how is None:
            raise SystemExit("parity gate: neither oracle/_ref/hacktv_ref nor a committed digest for %d frames -- refusing to report a number" % F)
        gate = "all %d frames x %d samples of every rank's block sha256 == %s" % (F, FS, how)
        if rank == 0 and gather:
            want = ref_sha(0, N * F)     # ... and the whole round as it arrived on rank 0
            if want is not None and hashlib.sha256(outs[0].cpu().numpy().tobytes()).hexdigest() != want:
                raise SystemExit("parity gate failed: the %d blocks gathered on rank 0 differ from the reference CLI's output" % N)
            gate += "; the gathered round of %d frames too" % (N * F)
        log("parity gate ok: " + gate)
    else:
        gate = "skipped (--noaudio is not the metric configuration)"

    for i in range(args.warmup):
        step(i)
    drain()
    settle_steps = 0
    if not args.walk_rounds:
        torch.cuda.synchronize()
        t_set = time.perf_counter()

        def settle_on():
            # rank 0's clock decides for everybody: every step is a send / receive pair between the ranks
            flag = torch.tensor([1 if time.perf_counter() - t_set < args.settle else 0], dtype=torch.int32)
            dist.broadcast(flag, 0, group=hostg)
            return bool(flag.item())
        while settle_on():
            for i in range(20):
                step(settle_steps + i)
            drain()
            torch.cuda.synchronize()
            settle_steps += 20

    def timed(fn_step, fn_drain):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn_step(i)
        fn_drain()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    walk_gate = None
    e.timing_enable(False)
    dt = timed(step, drain)
    if args.walk_rounds:
        if not noaudio:
            # ... and the LAST round walked is the reference's too: this rank's block of it, every sample
            lastb = sharding.block_of(rank, N, walk["last_round"])
            got = hashlib.sha256(mines[(args.steps - 1) % nbuf].cpu().numpy().tobytes()).hexdigest()
            want = ref_sha(lastb * F, F)
            if want is not None and got != want:
                raise SystemExit("parity gate failed on rank %d: block %d (round %d of the walk) differs from the reference CLI's output" % (rank, lastb, walk["last_round"]))
            walk_gate = None if want is None else "round %d (frames %d..%d on rank %d) sha256 == reference CLI" % (walk["last_round"], lastb * F, lastb * F + F - 1, rank)
            log("walk gate: %s" % walk_gate)
        kern_ms, n_k = 0.0, 0
        render_only = None
    else:
        # the kernel's own time: the same steps once more with HIP events around every launch on the launch stream
        e.timing_enable(True)
        timed(step, drain)
        kern_ms, n_k = e.timing_read(1)
        e.timing_enable(False)
        render_only = None
        if gather:
            dt2 = timed(lambda i: e.launch(ctypes.c_void_p(mines[i % nbuf].data_ptr())), lambda: None)
            render_only = N * F * FS * args.steps / dt2 / 1e6

    samples = N * F * FS
    value = samples * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # ---- beside it: the one-process C group over the same devices, in a child process with a time limit (rank 0); the
    # other ranks wait at a host-side barrier, their devices idle ----
    cg = None
    if not args.no_group and not args.walk_rounds:
        torch.cuda.synchronize()
        if rank == 0:
            devs = ",".join(["0"] * N) if dry else ",".join(str(d) for d in range(N))
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--devices", devs, "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--frames", str(F), "--no-cpu-baseline", "--detail-out", ""] + (["--noaudio"] if noaudio else [])
            env = clean_env()
            for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                       "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "OMP_NUM_THREADS"):
                env.pop(k_, None)
            try:
                pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.group_timeout, env=env, cwd=ROOT)
                lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                if pr.returncode == 0 and lines:
                    cg = json.loads(lines[-1])
                    log("c_group (child process): %.1f Msamples/s [%s]" % (cg["value"], cg["multi_gpu"].get("gather_backend")))
                else:
                    cg = {"failed": "rc %d: %s" % (pr.returncode, (pr.stderr.strip().splitlines() or ["no output"])[-1][-200:])}
                    log("c_group FAILED: " + cg["failed"])
            except subprocess.TimeoutExpired:
                cg = {"failed": "no result within %.0f s" % args.group_timeout}
                log("c_group FAILED: " + cg["failed"])
        dist.barrier(group=hostg)

    res = None
    if rank == 0:
        names = e.kernel_names()
        roof = hbm_roofline(names[-1], kern_ms, n_k, F * FS, traffic_per_launch(F))
        roof["path_frac"] = round(BYTES_PER_SAMPLE * samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / N, 4)
        mg = {
            "mode": "one rank per GPU (torch.distributed)", "ranks": N, "world_size": dist.get_world_size(),
            "backend": (args.dry_run_backend + " (dry run: every rank on GPU 0, transport through host memory)") if dry else "nccl (RCCL); sound chains' state between hosts: gloo",
            "gather_in_step": bool(gather), "gather_overlaps_render": bool(gather and not dry),
            "gathered_Msamples_per_s": round(value, 1) if gather else None,
            "render_only_Msamples_per_s": None if render_only is None else round(render_only, 1),
            "walk_rounds": bool(args.walk_rounds), "walk_gate": walk_gate, "seam_gate": seam_gate,
            "stage_one_block_with_sound_Msamples_per_s": round(F * FS / t_stage / 1e6, 1),
            "note": ("every step stages (sound chains from the rank before, host pre-pass, H2D) and renders the NEXT round: " + FLAT_NOTE) if args.walk_rounds else
                    ("value = launch + RCCL gather to rank 0 (inputs resident), bound by the root's ingest; " + FLAT_NOTE),
        }
        if cg is not None:
            if "failed" in cg:
                mg["c_group_failed"] = cg["failed"]
            else:
                for k_ in ("gather_backend", "gathered_Msamples_per_s", "render_only_Msamples_per_s", "host_direct_Msamples_per_s", "staged_every_round_Msamples_per_s"):
                    mg["c_group_" + k_] = cg["multi_gpu"].get(k_)
                mg["c_group_parity_gate"] = cg.get("parity_gate")
        res = {
            "metric": METRIC, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 data, int32 accumulate",
            "data": "synthetic: built-in test card + 1 kHz tone; " + ("every step stages and renders the NEXT round of blocks" if args.walk_rounds else
                    "a step = every rank renders its staged block of F frames again" + (" + the RCCL gather of the round to rank 0" if gather else "") + "; side inputs resident in HBM"),
            "config": {"workload": "-m i -s 16000000 --filter test%s (PAL-I AM-VSB + 51-tap FIR, FM mono + NICAM)" % (" --noaudio" if noaudio else ""),
                       "frames_per_gpu_per_step": F, "samples_per_step": samples,
                       "parallelism": "frames block-cyclic over %d GPUs, one rank each%s" % (N, ", RCCL gather to rank 0 in the step, overlapped with the next block's render" if gather else "")},
            "parity_gate": gate,
            "roofline": roof,
            "multi_gpu": mg,
            "settle": {"seconds": args.settle, "untimed_steps": settle_steps},
        }
        log("value %.1f Msamples/s over %d ranks, %.4f ms per step" % (value, N, ms_per_step))
    e.close()
    dist.barrier()
    dist.destroy_process_group()
    return res
