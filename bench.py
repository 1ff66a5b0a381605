#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s of the MI355X composite-video -> IQ engine.

Workload (BASELINE.json configs[1], the one the metric is quoted on):
  PAL System I, AM-VSB with the 51-tap FIR (`-m i -s 16000000 --filter test`),
  16 MHz sample rate, built-in test card, FM mono + NICAM-728 sound on.

A "step" is one pass of the hot path over one block of F whole frames per GPU,
through the C ABI of libhvk: hvk_k_direct, ONE kernel that composes every raster
sample from the picture's planes and the sub-carrier, filters it on the matrix
unit, adds the sound carriers and NICAM and stores the int16 I/Q (HVK_DIRECT=0:
the raster + filter kernel pair of the earlier rounds). The side inputs of the
block -- serial-carrier stream, NICAM symbols, and the PICTURE PLANES of the test
card (levels, low-passed chroma, burst: what depends on the picture alone is made
once per uploaded picture, hvk_k_prep) -- are staged into HBM before the clock
starts; every step renders all F frames of the staged block again (no sample is
kept between steps). What the per-picture work costs when every frame shows a new
picture is in "moving_pictures" (planes made inside the timed loop). Before any
number is taken EVERY sample of the block is compared with the unmodified
reference CLI run in the same job (and with its committed digest).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

N > 1: one process per GPU (`python -m torch.distributed.run ...`), frames
sharded block-cyclically (block b of F frames -> rank b mod N), each step ends
with the RCCL gather (grouped send/recv over xGMI) that reassembles the
contiguous IQ stream on rank 0 for the rf_* sink; --no-gather leaves it out.

Rank 0 prints ONE JSON line (contract in the task description) with two extra
objects: "roofline" (the dominant kernel against the 8 TB/s HBM peak, timed live
with HIP events on the launch stream; "path_frac" is the same ratio for the
whole step) and "cpu_baseline" (the unmodified reference, oracle/_ref/hacktv_ref,
timed on this box's host cores, median of three).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_SAMPLE = 4        # algorithmic traffic: one int16 I + one int16 Q written per sample
SAMPLE_RATE = 16000000
MODE = "i"


def a2_prepass(H, pcm):
    """The host pre-pass of an A2 stereo system (-m g --a2stereo: two FM carriers, pilot, identification tone -- four serial
    recurrences per sample) on its own, no device: with the tone / pilot pair on a thread of its own (the default) and in
    one thread (HVK_AUDIO_THREADS=0). Same samples either way (tests/test_host_path.py)."""
    import ctypes as C
    from hacktv_amd.engine import lib
    out = {}
    for key, env in (("Msamples_per_s", None), ("one_thread_Msamples_per_s", "0")):
        if env is None:
            os.environ.pop("HVK_AUDIO_THREADS", None)
        else:
            os.environ["HVK_AUDIO_THREADS"] = env
        conf = H.preset("g", H.FLAG_FILTER)
        conf.a2stereo = 1
        best = 0.0
        for _ in range(3):
            e = H.Engine(conf, SAMPLE_RATE, device=-1)
            fs = e.info["frame_samples"]
            n = 16 * fs
            while e.audio_needed(20) > 0:
                e.audio_write(pcm)
            car = np.ones((n, 2), np.int16)
            sym = np.zeros(n // 16 + 64, np.uint8)
            k0 = C.c_int64(0)
            lib().hvk_host_side_streams(e.h, 0, fs, car.ctypes.data, sym.ctypes.data, len(sym), C.byref(k0))
            t0 = time.perf_counter()
            lib().hvk_host_side_streams(e.h, fs, n, car.ctypes.data, sym.ctypes.data, len(sym), C.byref(k0))
            best = max(best, n / (time.perf_counter() - t0) / 1e6)
            e.close()
        out[key] = round(best, 1)
    os.environ.pop("HVK_AUDIO_THREADS", None)
    out["note"] = ("-m g --a2stereo --filter, 16 frames of the serial sound chains alone (no device), best of 3: the identification tone and "
                   "pilot -- constant steps, fed by nothing -- run ahead on a thread of their own; the one-loop form of round 2 measured 215 "
                   "Msamples/s on this host class (profiles/r03_a2_prepass.txt)")
    return out


def cpu_baseline(log):
    """The reference CLI on the host cores: steady state = (t[21 s of signal] - t[1 s]) / 20 s,
    which strips its ~0.5 s table build."""
    ref = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    flags = ["-m", MODE, "-s", str(SAMPLE_RATE), "--filter"]
    if os.path.exists(ref):
        def run(seconds):
            nbytes = seconds * SAMPLE_RATE * 4
            t = time.perf_counter()
            p = subprocess.Popen([ref] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            left = nbytes
            while left > 0:
                chunk = p.stdout.read(min(left, 1 << 22))
                if not chunk:
                    break
                left -= len(chunk)
            dt = time.perf_counter() - t
            p.kill()
            p.wait()
            return dt
        try:
            runs = []
            for _ in range(3):
                t1 = run(1)
                t11 = run(11)
                runs.append(10.0 * SAMPLE_RATE / (t11 - t1) / 1e6)
            runs.sort()
            return {"value": round(runs[1], 2), "unit": "Msamples/s", "cores": 3, "kind": "reference",
                    "runs": [round(v, 2) for v in runs],
                    "sample": "hacktv_ref -m i -s 16000000 --filter -o - test: (t[11 s of signal] - t[1 s]) / 10 s, median of 3; "
                              "3 busy threads (raster, vfilter, audio) of %d host cores" % (os.cpu_count() or 0)}
        except Exception as ex:  # pragma: no cover
            log("reference baseline failed: %r" % (ex,))

    # the oracle restatement, one core
    import oracle
    import util
    g = util.Golden()
    conf, sr = g.conf("i_full")
    with oracle.Oracle(conf, sr) as o:
        o.set_frame(g.frame("i_full"))
        o.set_audio(g.audio, True)
        o.render_lines(625)
        t = time.perf_counter()
        o.render_lines(625 * 20)
        dt = time.perf_counter() - t
    return {"value": round(20 * 640000 / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "oracle/liboracle.so, 20 frames of -m i --filter on one core"}


def ref_stream_sha(mode, sr, cli_flags, first, count, frame_bytes, env_extra=None):
    """sha256 of frames [first, first + count) of the unmodified reference CLI's output for these flags, run now
    (None: oracle/_ref/hacktv_ref is not there)."""
    import hashlib
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    if not os.path.exists(ref_bin):
        return None
    # a clean environment: under rocprofv3 the child would inherit the profiler's LD_PRELOAD and tool settings
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS", "ROCPROFILER"))}
    env.update(env_extra or {})
    p = subprocess.Popen([ref_bin, "-m", mode, "-s", str(sr)] + list(cli_flags) + ["-o", "-", "test"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
    skip, left, h = first * frame_bytes, count * frame_bytes, hashlib.sha256()
    while skip > 0:
        chunk = p.stdout.read(min(skip, 1 << 22))
        if not chunk:
            break
        skip -= len(chunk)
    while left > 0:
        chunk = p.stdout.read(min(left, 1 << 22))
        if not chunk:
            break
        h.update(chunk)
        left -= len(chunk)
    p.kill()
    p.wait()
    return h.hexdigest() if left == 0 else None


def time_steps(step, sync, warmup, steps):
    """`warmup` untimed calls of step(), then `steps` timed ones between two sync()s: seconds per step. The sections' clock
    (the headline has its own: settle phase, barriers, maximum over ranks -- main())."""
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps


def raw_teletext_rows(g, slot_counter):
    """The packets the reference's `raw:` source hands to the 32 teletext lines of the next frame (tests/golden/ttraw.bin,
    256 records): it reads on from where it stood, and the read that hits the end of the file yields NO packet before the
    file starts over (src/teletext.c:1187-1202). slot_counter: [line slots served so far] (updated)."""
    rec = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "ttraw.bin"), "rb").read(), np.uint8).reshape(-1, 42)
    n = len(rec)
    p = np.zeros((32, 45), np.uint8)
    p[:, 0] = 0x55
    p[:, 1] = 0x55
    p[:, 2] = 0x27
    mask = 0
    for r in range(32):
        j = slot_counter[0] % (n + 1)
        slot_counter[0] += 1
        if j < n:
            p[r, 3:] = rec[j]
            mask |= 1 << r
    return p, mask


def case_section(H, g, torch, case, F, steps, warmup, device, stream, label, stage_every_step=False, teletext=False, fresh_e2e=False, noaudio=False):
    """One BASELINE configuration as a bench section: golden case `case` (its preset edits and CLI flags), F-frame blocks.
    Gate: every sample of the first block == the unmodified reference CLI's output for the same flags, run in this job.
    Then `steps` steps: launches of the staged block (inputs resident), or stage + launch of a fresh block each
    (stage_every_step: SECAM, whose colour chain runs when a block is staged)."""
    import hashlib
    import util
    c = g.cases[case]
    conf, sr = g.conf(case)
    if noaudio:
        # the case's configuration without its sound (the device's share of a configuration whose stage is the host's serial sound chain)
        conf = H.preset(c["mode"], c["probe_flags"] | H.FLAG_NOAUDIO)
        conf.teletext = 1 if c.get("teletext") else 0
        for k_, v_ in c.get("extra", {}).items():
            setattr(conf, k_, v_)
    real = bool(c["real"])
    fs = c.get("frame_samples", c["width"] * c["lines"])
    frame_bytes = fs * (2 if real else 4)
    e = H.Engine(conf, sr, device=device, max_frames=F)
    e.set_stream(ctypes.c_void_p(stream.cuda_stream))
    e.frame_upload(0, g.frame(case))
    out = torch.empty((F * fs * 2,), dtype=torch.int16, device=torch.device("cuda", device))
    tt_slots = [0]
    state = {"next": 0}

    tt_blocks = []      # (the packets of every block to come, made before any clock starts: building them is this script's work, not the engine's)

    def stage_block():
        first = state["next"]
        if teletext:
            blk = first // F
            if blk >= len(tt_blocks):
                rm = [raw_teletext_rows(g, tt_slots) for _ in range(F)]
                tt_blocks.append((np.stack([r for r, _ in rm]), np.array([m for _, m in rm], np.uint32)))
            e.teletext_packets_block(0, *tt_blocks[blk])        # (one call per block: hvk_teletext_packets_block)
        e.stage(first, 1, F)
        state["next"] = first + F

    def feed(upto_blocks):
        # audio for the blocks to come (hvk_audio_needed counts from the engine's own next frame, which stage() does not
        # advance: feed by position instead)
        need = upto_blocks * F
        while e.audio_needed(need) > 0:
            e.audio_write(g.audio)

    nblocks = 1 + ((warmup + steps + 1) if stage_every_step else 0) + (1 if fresh_e2e else 0)
    if teletext:
        for _ in range(nblocks):
            rm = [raw_teletext_rows(g, tt_slots) for _ in range(F)]
            tt_blocks.append((np.stack([r for r, _ in rm]), np.array([m for _, m in rm], np.uint32)))
    feed(nblocks)
    t0 = time.perf_counter()
    stage_block()
    e.sync()
    t_stage = time.perf_counter() - t0
    e.launch(ctypes.c_void_p(out.data_ptr()))
    torch.cuda.synchronize()
    got = hashlib.sha256(util.stream_bytes(out.cpu().numpy().reshape(-1, 2), real)).hexdigest()
    flags = g.cli_flags(case) + (["--noaudio"] if noaudio else [])
    want = ref_stream_sha(c["mode"], sr, flags, 0, F, frame_bytes)
    if want is None and not noaudio:
        cum = c["sha256_cumulative"]
        if F <= len(cum):
            want = cum[F - 1]
    if want is None:
        gate = "no reference to compare %d frames with (oracle/_ref/hacktv_ref missing): NOT gated" % F
    elif got != want:
        raise SystemExit("parity gate failed for %s: %d frames differ from the reference CLI's output" % (label, F))
    else:
        gate = "all %d frames x %d samples sha256 == hacktv_ref %s run in this job" % (F, fs, " ".join(["-m", c["mode"], "-s", str(sr)] + flags))

    def step():
        if stage_every_step:
            stage_block()
        e.launch(ctypes.c_void_p(out.data_ptr()))

    dt = time_steps(step, torch.cuda.synchronize, warmup, steps)
    res = {
        "workload": " ".join(["-m", c["mode"], "-s", str(sr)] + [f if not f.startswith("raw:") else "raw:tests/golden/ttraw.bin" for f in flags] + ["test"]),
        "frames_per_step": F,
        "step": "stage (host pre-passes, colour chain on the device) + launch of a fresh block" if stage_every_step else "launch of the staged block (side inputs resident)",
        "Msamples_per_s": round(F * fs / dt / 1e6, 1),
        "ms_per_step": round(dt * 1e3, 4),
        "path_frac": round(BYTES_PER_SAMPLE * F * fs / dt / 1e9 / HBM_PEAK_GBS, 4),
        "parity_gate": gate,
        "kernels": e.kernel_names(),
        "first_block_stage_s": round(t_stage, 4),
    }
    try:
        res["secam_lines"] = e.secam_stats()        # (SECAM only: how the colour chain's speculation went)
    except Exception:
        pass
    if fresh_e2e:
        host_out = e.host_buffer(F * fs)
        e.fetch_wait(e.fetch_async(host_out, 0, F * fs))     # (a buffer that has been written to once: see end_to_end)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stage_block()
        e.launch()
        e.fetch_wait(e.fetch_async(host_out, 0, F * fs))
        t1 = time.perf_counter() - t0
        res["fresh_block_end_to_end_Msamples_per_s"] = round(F * fs / t1 / 1e6, 1)
        res["fresh_block_note"] = "one fresh block, nothing overlapped: stage (host pre-passes + H2D) + render + D2H of the int16 IQ into pinned host memory"
    e.close()
    return res


def dropin_section(flags, seconds=4, pin_clock=False, devnull_s=0):
    """The drop-in binary (the reference's own main(), av_test.c, rf_file.c, teletext.c + the video.h shim + libhvk) on
    these CLI flags: its first frames against the reference CLI's (both with the wall clock pinned where teletext needs
    it), then its steady-state rate from two run lengths."""
    import hashlib
    ref = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    hvk = os.path.join(ROOT, "oracle", "_ref", "hacktv_hvk")
    pin = os.path.join(ROOT, "oracle", "_ref", "pin_time.so")
    if not (os.path.exists(ref) and os.path.exists(hvk)):
        return None
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS", "ROCPROFILER"))}
    env["HVK_BATCH"] = "32"
    if pin_clock:
        env["LD_PRELOAD"] = pin
        env["TZ"] = "UTC"
    flags = [f.replace("@REF@", os.path.join(ROOT, "oracle", "_ref")) for f in flags]

    def run(binary, nbytes, digest=False):
        t = time.perf_counter()
        p = subprocess.Popen([binary] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        left, h = nbytes, hashlib.sha256()
        while left > 0:
            chunk = p.stdout.read(min(left, 1 << 22))
            if not chunk:
                break
            if digest:
                h.update(chunk)
            left -= len(chunk)
        dt = time.perf_counter() - t
        p.kill()
        p.wait()
        return dt, (h.hexdigest() if digest and left == 0 else None)

    fb = 640000 * 4
    nfr = 40
    _, a = run(ref, nfr * fb, True)
    _, b = run(hvk, nfr * fb, True)
    if a is None or b is None or a != b:
        raise SystemExit("drop-in gate failed: hacktv_hvk %s differs from hacktv_ref within the first %d frames" % (" ".join(flags), nfr))
    sr = 16000000
    t1 = min(run(hvk, 1 * sr * 4)[0], run(hvk, 1 * sr * 4)[0])
    t2, _ = run(hvk, (1 + seconds) * sr * 4)
    while t2 - t1 < 0.6 and seconds < 200:
        # too fast for the difference of two process lifetimes to mean anything: a longer run (the pipe carries 64 MB per second of signal)
        seconds *= 4
        t2, _ = run(hvk, (1 + seconds) * sr * 4)
    r1, _ = run(ref, 1 * sr * 4)
    r2, _ = run(ref, 3 * sr * 4)
    devnull = None
    if devnull_s:
        # the same binary writing to /dev/null for a few seconds, stopped by SIGINT: the shim's own count at exit (no pipe, no reader)
        import re, signal
        p = subprocess.Popen([hvk] + flags + ["-o", "/dev/null", "test"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(env, HVK_SHIM_STATS="1"), text=True)
        time.sleep(devnull_s)
        p.send_signal(signal.SIGINT)
        try:
            err = p.communicate(timeout=30)[1]
        except subprocess.TimeoutExpired:
            p.kill()
            err = p.communicate()[1]
        m = re.search(r"(\d+) frames in ([0-9.]+) s from the first line on = ([0-9.]+) Msamples/s", err or "")
        devnull = float(m.group(3)) if m else None
    return {
        **({"to_dev_null_Msamples_per_s": devnull,
            "to_dev_null_note": "-o /dev/null for %d s, the shim's count at exit: the unchanged main()'s loop -- one rf_write -> fwrite -> write(2) per line -- is what is left" % devnull_s}
           if devnull_s else {}),
        "workload": "hacktv_hvk " + " ".join(os.path.basename(f) if f.endswith(".tti") else f for f in flags) + " -o - test" + (" (time() pinned for both binaries: oracle/pin_time.c)" if pin_clock else ""),
        "parity_gate": "first %d frames of the drop-in binary's output sha256 == the reference CLI's, both run in this job" % nfr,
        "Msamples_per_s": round(seconds * sr / (t2 - t1) / 1e6, 1),
        "reference_cli_Msamples_per_s": round(2 * sr / (r2 - r1) / 1e6, 1),
        "note": "end to end through a pipe: the reference's main() and file sink, the shim's read-ahead worker (host sound pre-pass, uploads), "
                "render, D2H; (t[%d s of signal] - t[1 s]) / %d s" % (1 + seconds, seconds),
    }


def c_group_section(H, g, devices, Fb, rounds, log):
    """The several-devices path in C (hvk_group_*, hvk_group.cpp): blocks of Fb frames dealt round-robin to one engine per
    device named, the sound chains handed on in process, and both reassemblies of the contiguous stream -- (i) every engine's
    block read back into its place in one page-locked host buffer (N PCIe links: the shape a host rf_* sink wants), (ii) the
    blocks of a round gathered into the root engine's device memory (RCCL between distinct devices, device copies between
    engines that share one). Gate: the first round's 2 x Fb frames against the reference (committed digest where there is one)."""
    import hashlib
    N = len(devices)
    res = {"devices": list(devices), "engines": N, "block_frames": Fb}
    for sound in (True, False):
        conf = H.preset(MODE, H.FLAG_FILTER | (0 if sound else H.FLAG_NOAUDIO))
        key = "with_sound" if sound else "noaudio"
        with H.Group(conf, SAMPLE_RATE, devices, Fb) as grp:
            fs = grp.info["frame_samples"]
            res["gather_backend"] = grp.gather_backend()
            host = [grp.engines[0].host_buffer(N * Fb * fs) for _ in range(2)]
            def one_round(hb, gather_to=None):
                tk = []
                for b in range(N):
                    e = grp.block_engine()
                    if sound:
                        while grp.audio_needed(Fb) > 0:
                            grp.audio_write(g.audio)
                    grp.stage(Fb, slots=[0] * Fb)
                    grp.launch()
                    if gather_to is None:
                        tk.append((e, e.fetch_async(hb[b * Fb * fs:(b + 1) * Fb * fs], 0, Fb * fs)))
                if gather_to is not None:
                    grp.gather(0, gather_to, Fb * fs)
                return tk

            for e in grp.engines:
                e.frame_upload(0, g.frame("i_full"))
            # round 0, host-direct, gated
            for e, t in one_round(host[0]):
                e.fetch_wait(t)
            got = hashlib.sha256(host[0].tobytes()).hexdigest()
            gate = None
            if sound:
                long_file = os.path.join(ROOT, "tests", "golden", "ref_long.json")
                committed = json.load(open(long_file))["i_full"]["sha256_at_frames"] if os.path.exists(long_file) else {}
                want = committed.get(str(N * Fb))
                if want is None:
                    want = ref_stream_sha(MODE, SAMPLE_RATE, ["--filter"], 0, N * Fb, fs * 4)
                if want is not None:
                    if got != want:
                        raise SystemExit("c_group gate failed: %d engines x %d frames reassembled on the host differ from the reference CLI's output" % (N, Fb))
                    gate = "round 0 (%d frames over %d engines, sound chains handed on in process) sha256 == reference" % (N * Fb, N)
            res.setdefault("parity_gate", gate)
            # host-direct rounds: two host buffers, the read-back of a round runs beside the next round's stage + render
            t0 = time.perf_counter()
            pend = []
            for r in range(rounds):
                tk = one_round(host[r & 1])
                for e, t in pend:
                    e.fetch_wait(t)
                pend = tk
            for e, t in pend:
                e.fetch_wait(t)
            dt = time.perf_counter() - t0
            hd = N * Fb * fs * rounds / dt / 1e6
            # gathered rounds: into the root engine's device memory (a buffer of the test's own would need torch on that device;
            # the root's output buffer holds a block + a frame, so gather into a scratch allocation of the HIP runtime)
            import ctypes as C_
            hip = C_.CDLL("libamdhip64.so")
            hip.hipMalloc.argtypes = [C_.POINTER(C_.c_void_p), C_.c_size_t]
            hip.hipFree.argtypes = [C_.c_void_p]
            hip.hipSetDevice.argtypes = [C_.c_int]
            hip.hipSetDevice(devices[0])
            root = C_.c_void_p()
            gd = None
            if hip.hipMalloc(C_.byref(root), N * Fb * fs * 4) == 0:
                one_round(None, gather_to=root)
                grp.engines[0].sync()
                t0 = time.perf_counter()
                for r in range(rounds):
                    one_round(None, gather_to=root)
                for e in grp.engines:
                    e.sync()
                gd = N * Fb * fs * rounds / (time.perf_counter() - t0) / 1e6
                hip.hipFree(root)
            res[key] = {"host_direct_Msamples_per_s": round(hd, 1), "gathered_on_root_device_Msamples_per_s": None if gd is None else round(gd, 1)}
            log("c_group %s: host-direct %.1f, gathered %s Msamples/s" % (key, hd, gd))
    res["note"] = ("host code in C inside libhvk (no torch, no Python in the path): HVK_DEVICES=0,1,... makes the drop-in binary take it. WITH SOUND THE CURVE IS FLAT BY "
                   "CONSTRUCTION: the FM / AM phasor chain is one recurrence over every sample of the stream (src/video.c:2259-2276) -- each engine has to wait for the "
                   "state of the one before it, so N devices stage at the pace of one host core (about 0.5 Gsamples/s) whatever N is; --noaudio has no such chain and "
                   "scales with the devices and their PCIe links. Host-direct is the reassembly a host rf_* sink wants (src/hacktv.c:1579-1587 -> rf_write): one xGMI link "
                   "moves about 38 Gsamples/s, so a gather on one GPU is bound by the root's ingest before the samples have even started towards the host")
    return res


def c_group_timed(H, g, dist, torch, rank, devices, F, steps, warmup, noaudio, log):
    """N > 1, the path of record: ONE process (rank 0) drives one engine per device through hvk_group_* -- host code in C,
    as north_star asks -- and a step is a round: every engine renders its block of F frames, the blocks are reassembled on
    the root device (hvk_group_gather: RCCL between distinct devices, peer copies where RCCL is not to be had). K steps
    between barriers over ALL ranks (the others hold their devices idle), the maximum over ranks taken by the caller's
    all_reduce. Round 0 goes through the group's own stage / launch calls with the sound chains handed from engine to
    engine, and is hashed against the reference before anything is timed; the timed steps launch the staged blocks again
    (inputs resident in HBM, like the headline at N = 1)."""
    import ctypes as C_
    import hashlib
    N = len(devices)
    res = None
    t_steps = t_render = t_host = 0.0
    failed = None
    grp = fs = hip = root = gate = one = sync_all = t0 = None

    def guarded(fn):
        # (rank 0's part may fail -- a gate, a device call, a collective this machine has never run -- without leaving the other ranks
        # at a barrier: the failure is reported in the JSON, `value` then stays what the ranks measured through torch.distributed)
        nonlocal failed
        if rank != 0 or failed:
            return
        try:
            fn()
        except (Exception, SystemExit) as ex:
            failed = "%s: %s" % (type(ex).__name__, ex)
            log("c_group FAILED: " + failed)

    def setup():
        nonlocal grp, fs, hip, root, gate, one, sync_all, t0, t_steps, t_render, t_host, res
        if os.environ.get("BENCH_FAIL_C_GROUP"):      # (tests/test_gpu_block.py: what a failure of this part leaves of the line)
            raise RuntimeError("asked to fail (BENCH_FAIL_C_GROUP)")
        if len(set(devices)) > 1 and os.environ.get("HVK_GATHER") is None:
            # the collective branch has never met this machine: a small round in a process of its own, with a time limit, first
            try:
                pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gather_probe.py"), ",".join(str(d) for d in devices)],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
                ok = pr.returncode == 0 and "BACKEND" in pr.stdout
                log("c_group gather probe: " + (pr.stdout.strip().splitlines()[-1] if pr.stdout.strip() else "no output") + ("" if ok else " | " + pr.stderr[-300:]))
            except subprocess.TimeoutExpired:
                ok = False
                log("c_group gather probe: no answer within 180 s")
            if not ok:
                os.environ["HVK_GATHER"] = "peer"
                log("c_group: the gather goes by hipMemcpyPeerAsync (HVK_GATHER=peer)")
        conf = H.preset(MODE, H.FLAG_FILTER | (H.FLAG_NOAUDIO if noaudio else 0))
        grp = H.Group(conf, SAMPLE_RATE, devices, F)
        fs = grp.info["frame_samples"]
        hip = C_.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [C_.POINTER(C_.c_void_p), C_.c_size_t]
        hip.hipMemcpy.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_size_t, C_.c_int]
        hip.hipFree.argtypes = [C_.c_void_p]
        hip.hipSetDevice.argtypes = [C_.c_int]
        hip.hipSetDevice(devices[0])
        root = C_.c_void_p()
        if hip.hipMalloc(C_.byref(root), N * F * fs * 4) != 0:
            raise SystemExit("c_group: no room for the gathered round on the root device")
        for e in grp.engines:
            e.frame_upload(0, g.frame("i_full"))
        for b in range(N):
            if not noaudio:
                while grp.audio_needed(F) > 0:
                    grp.audio_write(g.audio)
            grp.stage(F, slots=[0] * F)
            grp.launch()
        grp.gather(0, root, F * fs)
        grp.engines[0].sync()
        gate = "skipped (--noaudio is not the metric configuration)"
        if not noaudio:
            host = np.zeros((N * F * fs, 2), np.int16)
            hip.hipSetDevice(devices[0])
            assert hip.hipMemcpy(host.ctypes.data, root, N * F * fs * 4, 2) == 0
            got = hashlib.sha256(host.tobytes()).hexdigest()
            del host
            want = ref_stream_sha(MODE, SAMPLE_RATE, ["--filter"], 0, N * F, fs * 4)
            if want is None:
                raise SystemExit("c_group gate: oracle/_ref/hacktv_ref is missing -- refusing to report a number")
            if got != want:
                raise SystemExit("c_group gate failed: %d engines x %d frames gathered on the root device differ from the reference CLI's output" % (N, F))
            gate = "round 0: %d frames over %d engines, sound chains handed on in process, gathered on device %d (%s): sha256 == hacktv_ref run in this job" % (N * F, N, devices[0], grp.gather_backend())
            log("c_group gate ok: " + gate)

        def one(gather=True):
            for e in grp.engines:
                e.launch()
            if gather:
                grp.gather(0, root, F * fs)

        def sync_all():
            for e in grp.engines:
                e.sync()
        for _ in range(warmup):
            one()
        sync_all()

    def timed_steps():
        nonlocal grp, fs, hip, root, gate, one, sync_all, t0, t_steps, t_render, t_host, res
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        sync_all()

    def rest():
        nonlocal grp, fs, hip, root, gate, one, sync_all, t0, t_steps, t_render, t_host, res
        t_steps = time.perf_counter() - t0
        # beside it: the same launches without the reassembly, and with the host-direct reassembly (every engine's block
        # read back into its place in one page-locked stream buffer: N PCIe links, what a host rf_* sink wants)
        t0 = time.perf_counter()
        for _ in range(steps):
            one(False)
        sync_all()
        t_render = time.perf_counter() - t0
        hb = grp.engines[0].host_buffer(N * F * fs)
        k_hd = max(2, min(steps, 10))
        t0 = time.perf_counter()
        for _ in range(k_hd):
            one(False)
            tk = [(e, e.fetch_async(hb[i * F * fs:(i + 1) * F * fs], 0, F * fs)) for i, e in enumerate(grp.engines)]
            for e, t in tk:
                e.fetch_wait(t)
        t_host = (time.perf_counter() - t0) / k_hd
        res = {"devices": list(devices), "engines": N, "block_frames": F, "gather_backend": grp.gather_backend(), "parity_gate": gate,
               "gathered_on_root_device_Msamples_per_s": round(N * F * fs * steps / t_steps / 1e6, 1),
               "render_only_Msamples_per_s": round(N * F * fs * steps / t_render / 1e6, 1),
               "host_direct_Msamples_per_s": round(N * F * fs / t_host / 1e6, 1),
               "note": "one process, N devices, host code in C (hvk_group_*): `value` is the gathered figure -- bound by the root's ingest, about 38 Gsamples/s per "
                       "xGMI link, not by the kernels; render_only is the same launches without the reassembly; host_direct reads every engine's block back into "
                       "its place in one page-locked stream buffer (N PCIe links)"}
        hip.hipSetDevice(devices[0])
        hip.hipFree(root)
        grp.close()

    guarded(setup)
    dist.barrier()
    guarded(timed_steps)
    torch.cuda.synchronize()
    dist.barrier()
    guarded(rest)
    if failed:
        return -1.0, {"failed": failed, "devices": list(devices)}
    return t_steps, res

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the RCCL reassembly out of the step")
    ap.add_argument("--walk-rounds", action="store_true",
                    help="every step takes the NEXT round of blocks: sound chains handed from rank to rank, host pre-pass, H2D and render of a fresh block inside the timed loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--noaudio", action="store_true", help="render the --noaudio variant instead")
    ap.add_argument("--no-moving", action="store_true", help="skip the moving-picture section")
    ap.add_argument("--no-configs", action="store_true", help="skip the sections for BASELINE configs 1, 3, 4 and --noaudio")
    ap.add_argument("--hour-sound", action="store_true", help="(the default since round 5) 5_one_hour: the whole hour WITH sound as well (two minutes: the host's serial FM chain)")
    ap.add_argument("--no-hour-sound", action="store_true", help="5_one_hour: leave the run with sound out (it takes two minutes)")
    ap.add_argument("--no-hour", action="store_true", help="skip the one-hour section")
    ap.add_argument("--settle", type=float, default=0.6, help="seconds of untimed launches before the clock starts (sustained clocks)")
    ap.add_argument("--dry-run-backend", default=None, help="gloo: dry-run the N > 1 path with every rank on GPU 0 (no RCCL peers needed)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hacktv_amd as H
    from hacktv_amd import sharding
    import util

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    N = world

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    dry = args.dry_run_backend is not None
    if dry:
        local_rank = 0                      # every rank shares GPU 0; transport through host memory
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(args.dry_run_backend)
        else:
            dist.init_process_group("nccl", device_id=dev)
        # the sound chains' state travels between the ranks' hosts, in a group of its own: its messages must not queue up
        # between the blocks of the gather (a rank hands the chains on BEFORE it renders and sends its block)
        hostg = dist.new_group(backend="gloo")

    if N == 1:
        hostg = None
    g = util.Golden()
    flags = H.FLAG_FILTER | (H.FLAG_NOAUDIO if args.noaudio else 0)
    conf = H.preset(MODE, flags)
    F = args.frames
    e = H.Engine(conf, SAMPLE_RATE, device=local_rank, max_frames=F)
    FS = e.info["frame_samples"]
    # A stream of our own, made torch's current one: the engine launches on it (a null handle would send the engine
    # back to its private stream), and RCCL's point-to-point operations order themselves behind torch's CURRENT stream --
    # the send of a block has to wait for the render that was just enqueued there.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    e.set_stream(ctypes.c_void_p(stream.cuda_stream))
    e.frame_upload(0, g.frame("i_full"))
    gather = N > 1 and not args.no_gather
    import hashlib
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")

    def ref_sha(first, count):
        """sha256 of frames [first, first + count) of the unmodified reference CLI's output for the metric configuration, run now (None: no binary)."""
        return ref_stream_sha(MODE, SAMPLE_RATE, ["--filter"] + (["--noaudio"] if args.noaudio else []), first, count, FS * 4)

    def feed_audio(upto_frame, source_pos=None):
        """32 kHz source samples up to frame `upto_frame`; source_pos: the engine has just taken over another rank's sound
        chains and its (empty) queue goes on at that position of the source -- the test tone is a loop."""
        if source_pos is not None:
            e.audio_write(g.audio[source_pos % len(g.audio):])
        while e.audio_needed(upto_frame) > 0:
            e.audio_write(g.audio)

    def stage_block(block, Fb, last=False):
        """Stage block `block` (Fb frames) on the rank it belongs to: take the sound chains over from the rank that staged the
        block before, run them over this block's frames only, hand them on."""
        first = block * Fb
        pos = None if args.noaudio else sharding.sound_state_recv(e, N, block, hostg)
        if not args.noaudio:
            feed_audio(first + Fb, pos)
        e.stage(first, 1, Fb, prev_slots=[0] * Fb)
        if not args.noaudio:
            sharding.sound_state_send(e, N, block, hostg, last=last)

    # ---- N > 1: the sharded path end to end on short blocks, BEFORE anything is timed: every rank stages, renders and
    # sends two rounds of 2-frame blocks through the same calls as the timed loop (stage with the predecessor slot,
    # double-buffered gather), rank 0 hashes the reassembled stream -- block seams and round seams included -- against
    # the reference CLI's output ----
    seam_gate = None
    if N > 1 and not args.noaudio:
        Fg, rounds = min(2, F), 2
        bufs = [torch.empty((Fg * FS * 2,), dtype=torch.int16, device=dev) for _ in range(2)]
        roots = [torch.empty((N, Fg * FS * 2), dtype=torch.int16, device=dev) for _ in range(2)] if rank == 0 else [None, None]
        host = []
        works = []
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
            want = ref_sha(0, rounds * N * Fg)
            if want is None:
                k = rounds * N * Fg
                cum = g.cases["i_full"]["sha256_cumulative"]
                want = cum[k - 1] if k <= len(cum) else None
            if want is None:
                raise SystemExit("seam gate: no reference to compare %d frames with -- refusing to report a number" % (rounds * N * Fg))
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
    mine = mines[0]
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
                # (the communicator's stream waits for the render just enqueued on the current stream before it sends)
                pending[:] = sharding.gather_start(mines[b], outs[b], rank, N)

    def drain():
        if gather and not dry:
            sharding.gather_wait(pending)
            pending[:] = []

    # ---- parity gate before any number: EVERY sample of this rank's block against the unmodified reference ----
    # (--walk-rounds: rounds 0 [this gate], then warm-up and timed steps one round each, then nothing: the last rank of the
    # last round keeps the chains' state to itself)
    walk["last_round"] = args.warmup + args.steps if args.walk_rounds else 0
    step(0)
    drain()
    torch.cuda.synchronize()
    if not args.noaudio:
        mine_sha = hashlib.sha256(mines[0].cpu().numpy().tobytes()).hexdigest()
        want, how = ref_sha(first_frame, F), None
        if want is not None:
            how = "hacktv_ref run in this job"
            if mine_sha != want:
                raise SystemExit("parity gate failed on rank %d: frames %d..%d differ from the reference CLI's output" % (rank, first_frame, first_frame + F - 1))
        long_file = os.path.join(ROOT, "tests", "golden", "ref_long.json")
        committed = json.load(open(long_file))["i_full"]["sha256_at_frames"] if os.path.exists(long_file) else {}
        if first_frame == 0 and str(F) in committed:
            if mine_sha != committed[str(F)]:
                raise SystemExit("parity gate failed: the first %d frames differ from the committed reference digest" % F)
            how = (how + " + committed digest") if how else "committed digest"
        if how is None:
            raise SystemExit("parity gate: neither oracle/_ref/hacktv_ref nor a committed digest for %d frames -- refusing to report a number" % F)
        gate = "all %d frames x %d samples of rank %d's block sha256 == %s" % (F, FS, rank, how)
        log("parity gate ok: " + gate)
        if rank == 0 and gather:
            # ... and the whole round as it arrived on rank 0
            want = ref_sha(0, N * F)
            if want is not None and hashlib.sha256(outs[0].cpu().numpy().tobytes()).hexdigest() != want:
                raise SystemExit("parity gate failed: the %d blocks gathered on rank 0 differ from the reference CLI's output" % N)
    else:
        gate = "skipped (--noaudio is not the metric configuration)"

    for i in range(args.warmup):
        step(i)
    drain()
    # ... and, untimed, the same launches for --settle seconds more: a short timed region (20 steps are 5 ms) then sees the
    # clocks a long run has, not the first milliseconds after an idle period
    settle_steps = 0
    if not args.walk_rounds:
        torch.cuda.synchronize()
        t_set = time.perf_counter()

        def settle_on():
            """Another group of settling steps? Rank 0's clock decides for everybody: every step is a send / receive pair
            between the ranks, and two ranks that looked at their own clocks a millisecond apart would leave the loop a
            group apart -- one of them waiting for a block nobody sends."""
            more = time.perf_counter() - t_set < args.settle
            if N > 1:
                flag = torch.tensor([1 if more else 0], dtype=torch.int32)
                dist.broadcast(flag, 0, group=hostg)
                more = bool(flag.item())
            return more
        while settle_on():
            for i in range(20):
                step(settle_steps + i)
            drain()
            torch.cuda.synchronize()
            settle_steps += 20

    def timed(fn_step, fn_drain):
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn_step(i)
        fn_drain()
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if N > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    walk_gate = None
    if args.walk_rounds:
        # one pass: every step stages and renders the next round (the events' cost is nothing beside a stage)
        e.timing_enable(True)
        dt = timed(step, drain)
        raster_ms, n_r = e.timing_read(0)
        filter_ms, n_f = e.timing_read(1)
        e.timing_enable(False)
        if not args.noaudio:
            # ... and the LAST round walked is the reference's too: this rank's block of it, every sample
            lastb = sharding.block_of(rank, N, walk["last_round"])
            got = hashlib.sha256(mines[(args.steps - 1) % nbuf].cpu().numpy().tobytes()).hexdigest()   # (the timed loop counts its steps from 0)
            want = ref_sha(lastb * F, F)
            if want is not None and got != want:
                raise SystemExit("parity gate failed on rank %d: block %d (round %d of the walk) differs from the reference CLI's output" % (rank, lastb, walk["last_round"]))
            walk_gate = None if want is None else "round %d (frames %d..%d on rank %d) sha256 == reference CLI" % (walk["last_round"], lastb * F, lastb * F + F - 1, rank)
            log("walk gate: %s" % walk_gate)
    else:
        # the timed region: K steps, nothing but launches (and the gather at N > 1) between the barriers
        dt = timed(step, drain)

        # the kernels' own time, for the roofline object: the same steps once more with HIP events recorded around every
        # launch on the launch stream (their recording costs a little: not inside the region `value` comes from)
        e.timing_enable(True)
        timed(step, drain)
        raster_ms, n_r = e.timing_read(0)
        filter_ms, n_f = e.timing_read(1)
        e.timing_enable(False)

    samples_per_step = N * F * FS
    value = samples_per_step * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # the spread: ten more runs of a tenth of the steps each (at least 10), every run between synchronisations
    sub_ms = []
    if not args.walk_rounds:
        nsub = max(10, args.steps // 10)
        for _ in range(10):
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            for i in range(nsub):
                step(i)
            drain()
            torch.cuda.synchronize()
            sub_ms.append((time.perf_counter() - t_s) / nsub * 1e3)
        sub_ms.sort()

    # N > 1: the same steps without the reassembly, for the record (the ranks share nothing then)
    render_only = None
    if gather and not args.walk_rounds:
        dt2 = timed(lambda i: e.launch(ctypes.c_void_p(mines[i % nbuf].data_ptr())), lambda: None)
        render_only = samples_per_step * args.steps / dt2 / 1e6

    # ---- N > 1: the path of record is the C group (one process, N devices); what the ranks measured above through
    # torch.distributed stays beside it as the cross-check ----
    cg_timed = None
    harness = None
    if N > 1 and not args.walk_rounds:
        harness = {"gathered_Msamples_per_s": round(value, 1) if gather else None, "ms_per_step": round(ms_per_step, 4),
                   "render_only_Msamples_per_s": None if render_only is None else round(render_only, 1)}
        dt_c, cg_timed = c_group_timed(H, g, dist, torch, rank, [0] * N if dry else list(range(N)), F, args.steps, args.warmup, args.noaudio, log)
        tt = torch.tensor([dt_c], dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        c_failed = float(tt.item()) < 0
        tt = torch.tensor([dt_c], dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if not c_failed:
            dt = float(tt.item())
            value = samples_per_step * args.steps / dt / 1e6
            ms_per_step = dt / args.steps * 1e3
            sub_ms = []
        else:
            # (reported, not hidden: `value` stays the gathered figure of the torch.distributed harness above, which has its own gate)
            log("c_group failed: `value` is the harness's")

    # ---- one FRESH block end to end: host pre-pass + H2D of the side inputs, render, D2H of the samples ----
    e2e = None
    if N == 1 and not args.walk_rounds:
        host_out = e.host_buffer(F * FS)
        e.fetch_wait(e.fetch_async(host_out, 0, F * FS))     # (the first copy into a fresh page-locked buffer runs at half the link's rate: not what a sink that keeps its buffers sees)
        nxt = first_frame + F
        t0 = time.perf_counter()
        while e.audio_needed(nxt + F) > 0:
            e.audio_write(g.audio)
        e.stage(nxt, 1, F)
        e.sync()                    # (the side inputs' copies to the device are part of the stage, not of the render: 328 MB of carriers over the same link)
        t1 = time.perf_counter()
        e.launch()
        e.fetch_wait(e.fetch_async(host_out, 0, F * FS))
        t2 = time.perf_counter()
        e2e = {"stage_s": round(t1 - t0, 4), "render_and_d2h_s": round(t2 - t1, 4),
               "Msamples_per_s": round(F * FS / (t2 - t0) / 1e6, 1),
               "render_and_d2h_Msamples_per_s": round(F * FS / (t2 - t1) / 1e6, 1),
               "note": "one fresh block, nothing overlapped: host audio control path (the serial FM phasor chain, one core) + H2D of the "
                       "side inputs (waited for: stage_s), then render + D2H of the int16 IQ into page-locked host memory that has been written to before "
                       "(render_and_d2h_s: the link's 56 GB/s; rounds 3-4 counted the tail of the side inputs' H2D in it); the PCIe-inclusive rate, never `value`"}

    # ---- pictures that change every frame (the 7 B/sample regime): F new pictures per step, uploaded
    # inside the timed loop (pinned ring, asynchronous copies), --noaudio so that the serial sound pre-pass does not
    # hide what is being measured; beside it the same launches with the pictures resident ----
    moving = None
    if N == 1 and not args.no_moving:
        Fm = min(F, 64)
        rng = np.random.default_rng(1)
        yy, xx = np.mgrid[0:576, 0:832]
        pics = []
        for i in range(8):
            r = (xx * 255 // 831 + 31 * i) & 255
            gch = (yy * 255 // 575 + 17 * i) & 255
            b = ((xx + yy) // 6 + 53 * i) & 255
            noise = rng.integers(0, 4, (576, 832, 3))
            pics.append((((r + noise[..., 0]) & 255) << 16 | ((gch + noise[..., 1]) & 255) << 8 | ((b + noise[..., 2]) & 255)).astype(np.uint32))
        em = H.Engine(H.preset(MODE, H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=local_rank, max_frames=Fm)
        em.set_stream(ctypes.c_void_p(stream.cuda_stream))
        slots = list(range(Fm))
        outm = torch.empty((Fm * FS * 2,), dtype=torch.int16, device=dev)

        # the same pictures once more in page-locked memory (a source that decodes into hvk_host_alloc() memory)
        pinned = [em.host_picture(576, 832) for _ in pics]
        for hp, pic in zip(pinned, pics):
            hp[:] = pic

        def mstep(k, upload):
            if upload == 3:
                em.planes_refresh(slots)        # pictures resident, their planes made again: the per-picture work without PCIe
            if upload == 1:
                for i in range(Fm):
                    em.frame_upload(i, pics[(k * Fm + i) % len(pics)])
            elif upload == 2:
                for i in range(Fm):
                    em.frame_upload_pinned(i, pinned[(k * Fm + i) % len(pinned)])
            em.stage(k * Fm, 1, Fm, slots=slots)
            em.launch(ctypes.c_void_p(outm.data_ptr()))

        for k in range(2):
            mstep(k, True)
        torch.cuda.synchronize()
        ksteps = 5
        t0 = time.perf_counter()
        for k in range(ksteps):
            mstep(2 + k, True)
        torch.cuda.synchronize()
        t_up = time.perf_counter() - t0
        t0 = time.perf_counter()
        for k in range(ksteps):
            mstep(2 + ksteps + k, False)
        torch.cuda.synchronize()
        t_res = time.perf_counter() - t0
        t0 = time.perf_counter()
        for k in range(ksteps):
            mstep(2 + ksteps + k, 3)
        torch.cuda.synchronize()
        t_prep = time.perf_counter() - t0
        for k in range(2):
            mstep(2 + 2 * ksteps + k, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(ksteps):
            mstep(4 + 2 * ksteps + k, 2)
        torch.cuda.synchronize()
        t_pin = time.perf_counter() - t0
        fused_used = em.fused_launches()

        def new_pictures(levels, fused, card):
            """Fm new pictures per step, resident in HBM: through the picture planes (HVK_FUSED=0: hvk_k_prep8 + hvk_k_direct) or
            from the pixels in one kernel (hvk_k_fused, what the engine takes by itself when most of a block's pictures are new)."""
            os.environ["HVK_FUSED"] = "1" if fused else "0"
            try:
                ex = H.Engine(H.preset(MODE, H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=local_rank, max_frames=Fm)
            finally:
                del os.environ["HVK_FUSED"]
            ex.set_stream(ctypes.c_void_p(stream.cuda_stream))
            ex.set_levels(levels)
            for i in range(Fm):
                ex.frame_upload(i, np.roll(g.frame("i_full"), 13 * i, axis=1) if card else pics[i % len(pics)])
            nxt = [0]

            def one():
                ex.planes_refresh(slots); ex.stage(nxt[0] * Fm, 1, Fm, slots=slots); ex.launch(ctypes.c_void_p(outm.data_ptr()))
                nxt[0] += 1
            dt_ = time_steps(one, torch.cuda.synchronize, 2, ksteps * 2)
            nf = ex.fused_launches()
            ex.close()
            return round(Fm * FS / dt_ / 1e6, 1), nf

        def new_pictures_m(levels):
            """The same at BASELINE config 3's geometry (-m m -s 13500000 --filter --noaudio: 858-sample lines, 11-tap chroma): the one kernel
            from the pixels exists for 1024-sample lines only, so new pictures go through hvk_k_prep8 + hvk_k_direct there."""
            ex = H.Engine(H.preset("m", H.FLAG_FILTER | H.FLAG_NOAUDIO), 13500000, device=local_rank, max_frames=Fm)
            ex.set_stream(ctypes.c_void_p(stream.cuda_stream))
            ex.set_levels(levels)
            base = g.frame("m_full")
            rr = np.random.default_rng(2)
            for i in range(Fm):
                pic = np.roll(base, 13 * i, axis=1)
                if levels == 2:
                    pic = (pic ^ (rr.integers(0, 4, base.shape, dtype=np.uint32) * np.uint32(0x010101))).astype(np.uint32)     # (low-bit noise: many colours)
                ex.frame_upload(i, pic)
            fsm = ex.info["frame_samples"]
            outm_m = torch.empty((Fm * fsm * 2,), dtype=torch.int16, device=dev)
            nxt = [0]

            def one():
                ex.planes_refresh(slots); ex.stage(nxt[0] * Fm, 1, Fm, slots=slots); ex.launch(ctypes.c_void_p(outm_m.data_ptr()))
                nxt[0] += 1
            dt_ = time_steps(one, torch.cuda.synchronize, 2, ksteps * 2)
            names_m = ex.kernel_names()
            ex.close()
            return round(Fm * fsm / dt_ / 1e6, 1), names_m

        np_m_tab, names_m = new_pictures_m(1)
        np_m_cmp, _ = new_pictures_m(2)
        np_tab_f, nf1 = new_pictures(1, True, True)
        np_tab_p, _ = new_pictures(1, False, True)
        np_cmp_f, nf2 = new_pictures(2, True, False)
        np_cmp_p, _ = new_pictures(2, False, False)
        moving = {
            "new_pictures_every_frame": {
                "table_levels_Msamples_per_s": max(np_tab_f, np_tab_p), "computed_levels_Msamples_per_s": max(np_cmp_f, np_cmp_p),
                "one_kernel_from_the_pixels": {"table_levels": np_tab_f, "computed_levels": np_cmp_f, "kernel": "hvk_k_fused<13, LV>", "launches_that_way": [nf1, nf2]},
                "through_picture_planes": {"table_levels": np_tab_p, "computed_levels": np_cmp_p, "kernels": "hvk_k_prep8<13, 1024, LV> + hvk_k_direct"},
                "ntsc_m": {"table_levels_Msamples_per_s": np_m_tab, "computed_levels_Msamples_per_s": np_m_cmp, "kernels": "hvk_k_prep8<11, 0, LV> + " + names_m[-1],
                           "workload": "-m m -s 13500000 --filter --noaudio (BASELINE config 3's geometry: 858-sample lines), %d new pictures per step; through the picture planes: "
                                       "the one kernel from the pixels (hvk_k_fused) is 1024 samples a line" % Fm},
                "note": "%d pictures resident in HBM, every one NEW in every step (hvk_planes_refresh): table levels = shifted test cards (few colours: the 2^24-entry "
                        "level table serves from cache), computed levels = gradients + noise (levels by FP64 arithmetic per pixel). The engine takes the one kernel "
                        "by itself for a block whose pictures are mostly new (HVK_FUSED unset); the first figure of each pair is the faster of the two ways" % Fm,
            },
            "workload": "-m i -s 16000000 --filter --noaudio, a different 832 x 576 picture on every frame (smooth gradients + noise), %d frames per step" % Fm,
            "computed_levels_arithmetic": {"short_form": em.levels_short_form(),
                                           "note": "hvk_levels_short_form(): 2 = levels computed per pixel take the short form of the FP64 arithmetic (11 operations "
                                                   "a pixel instead of 38), which hvk_open() TRIED on all 2^24 colours of the mode against the table made with the "
                                                   "reference's sequence of operations; 1 = for the colour-difference levels only; 0 = the reference's sequence"},
            "with_uploads_Msamples_per_s": round(Fm * FS * ksteps / t_up / 1e6, 1),
            "with_uploads_from_pinned_memory_Msamples_per_s": round(Fm * FS * ksteps / t_pin / 1e6, 1),
            "pictures_resident_Msamples_per_s": round(Fm * FS * ksteps / t_res / 1e6, 1),
            "pictures_resident_planes_made_every_step_Msamples_per_s": round(Fm * FS * ksteps / t_prep / 1e6, 1),
            "kernels": em.kernel_names() + (["hvk_k_fused<13, 1> (%d launches of this engine rendered from the pixels)" % fused_used] if fused_used else []),
            "note": "with uploads: every picture goes host -> pinned ring -> HBM inside the timed loop (1.9 MB per frame over PCIe, plus the copy "
                    "into pinned memory on one host core); from pinned memory: the pictures already lie in page-locked memory "
                    "(hvk_frame_upload_pinned: one DMA per picture, no host copy); resident: the same launches re-using the uploaded pictures AND their planes; planes_made_every_step: the "
                    "pictures stay in HBM but hvk_k_prep (levels, chroma low pass) runs for every one of them in every step -- the device-side cost "
                    "of a new picture on every frame. Levels are computed per pixel (many colours: the 2^24-entry table would miss)",
        }
        em.close()

    # ---- SECAM-L (BASELINE config 4's mode): the colour sub-carrier's line-to-line chain runs on the device when a
    # block is staged (hvk_secam.hip), so here a step is stage + launch of a fresh block; beside it the host's serial
    # chain on one short block ----
    secam = None
    if N == 1 and not args.no_moving:
        def secam_run(Fs, ksteps, wsteps=16, pics=None, refresh=False):
            # (the warm-up steps also let the number of warm-up LINES per start state settle: it follows the pictures, one
            # line down per clean block, two up per block with a wrong start -- hvk_engine.cpp)
            es = H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=local_rank, max_frames=Fs)
            slots = None
            if pics is None:
                es.frame_upload(0, g.frame("l_full"))
            else:
                for i_, p_ in enumerate(pics):
                    es.frame_upload(i_, p_)
                slots = [i_ % len(pics) for i_ in range(Fs)]
            nxt = [0]

            def one():
                if refresh:
                    es.planes_refresh(slots)        # (every picture's luma and (U, V) planes made again: hvk_k_prep8)
                es.stage(nxt[0] * Fs, 1, Fs, slots=slots)
                es.launch()
                nxt[0] += 1
            time_steps(one, es.sync, 0, wsteps)     # (untimed: lets the number of warm-up lines settle)
            st0 = es.secam_stats()
            est0 = es.secam_estimated_stages()
            t_dev = time_steps(one, es.sync, 0, ksteps)
            st = es.secam_stats()
            st = {kk: st[kk] - st0[kk] for kk in st}        # the timed steps' lines
            st["warmup_lines_per_start_state"] = es.secam_warmup_lines()
            st["stages_with_estimated_entry_states"] = es.secam_estimated_stages() - est0
            names_s = es.kernel_names()
            es.close()
            return t_dev, st, names_s

        t_dev, st, names_s = secam_run(F, 5)
        t_big, st_big, _ = secam_run(4 * F, 5)
        # pictures that change: the cells (levels, vertical average, low pass) are every frame's own work again, and
        # noisy pictures make the walk's table reads scatter
        rngs = np.random.default_rng(3)
        yy_, xx_ = np.mgrid[0:576, 0:832]
        noisy = []
        for i_ in range(4):
            p_ = (((xx_ * 255 // 831 + i_ * 17) % 256).astype(np.uint32) << 16) | (((yy_ * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx_ + yy_) // 3 % 256).astype(np.uint32))
            noisy.append(np.where(rngs.random(p_.shape) < 0.2, rngs.integers(0, 1 << 24, p_.shape, dtype=np.uint32), p_).astype(np.uint32))
        os.environ["HVK_SECAM_NO_CELL_CACHE"] = "1"
        t_mov, st_mov, _ = secam_run(4 * F, 3, wsteps=4, pics=noisy)
        # ... and with a picture slot per frame whose planes (luma through the notch, the pixels' colour-difference levels)
        # are made again in every step as well: everything a new picture on every frame costs on the device
        t_new, st_new, names_new = secam_run(4 * F, 3, wsteps=3, pics=[noisy[i_ % 4] for i_ in range(4 * F)], refresh=True)
        del os.environ["HVK_SECAM_NO_CELL_CACHE"]
        os.environ["HVK_SECAM_HOST"] = "1"
        eh = H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=local_rank, max_frames=8)
        eh.frame_upload(0, g.frame("l_full"))
        t0 = time.perf_counter()
        eh.stage(0, 1, 8)
        eh.launch()
        eh.sync()
        t_host = time.perf_counter() - t0
        eh.close()
        del os.environ["HVK_SECAM_HOST"]
        secam = {
            "workload": "-m l -s 16000000 --filter --noaudio test, %d frames per step, every step stages (= runs the colour chain: every line of every frame walked and checked) "
                        "and renders a fresh block. Per-picture work is done once per picture, like the headline's picture planes: the test card's low-passed colour cells (per frame parity) "
                        "and its luma planes; and a line's walk starts from the state the line had the last time the picture was shown with this frame number modulo 6, "
                        "which for a picture that stays is the state it has now -- no warm-up lines (lines.warmup_lines_per_start_state), every line still walked "
                        "once and its start state checked bit for bit" % (4 * F),
            "Msamples_per_s": round(4 * F * FS / t_big / 1e6, 1),
            "ms_per_step": round(t_big * 1e3, 3),
            "lines": st_big,
            "blocks_of_%d_frames" % F: {"Msamples_per_s": round(F * FS / t_dev / 1e6, 1), "ms_per_step": round(t_dev * 1e3, 3), "lines": st,
                                         "note": "the block size of the PAL-I headline: a quarter of the lines, and the chain -- one lane per line, bound by the latency "
                                                 "of its dependent steps -- takes nearly as long: about one wave per SIMD instead of four"},
            "pictures_change_every_frame": {"Msamples_per_s": round(4 * F * FS / t_mov / 1e6, 1), "ms_per_step": round(t_mov * 1e3, 3), "lines": st_mov,
                                            "note": "noisy pictures (gradients, a fifth of the pixels random colours), resident in HBM, the cells made for EVERY frame "
                                                    "(HVK_SECAM_NO_CELL_CACHE=1) and every line's entry state new (no state kept from a last showing): the colour chain's "
                                                    "share of a moving source -- the measure of rounds 2 and 3. With the test card a picture's cells are "
                                                    "made once per frame parity and kept (per-picture work, like the picture planes of the PAL-I headline); the walk "
                                                    "from line to line, the check and the render are every frame's in both. Since round 4 the entry states of new "
                                                    "pictures' lines are estimated (hvk_k_secam_est: the values behind a line from the summed angle of the FM steps, "
                                                    "the IIR's state from a walk of the IIR alone) instead of derived by walking the twelve lines before, and the cells "
                                                    "are made from the pictures' (U, V) plane"},
            "new_picture_every_frame": {"Msamples_per_s": round(4 * F * FS / t_new / 1e6, 1), "ms_per_step": round(t_new * 1e3, 3), "lines": st_new,
                                        "kernels": names_new,
                                        "note": "%d picture slots, one per frame of the block, and in every step every slot's planes are made again too "
                                                "(hvk_planes_refresh -> hvk_k_prep8<1, 0, LV, 1>: levels computed per pixel, luma through the 51-tap notch, (U, V) plane) before "
                                                "cells, estimate, walk, check and render: the whole device-side cost of a new picture on every frame, uploads apart" % (4 * F)},
            "host_chain_Msamples_per_s": round(8 * FS / t_host / 1e6, 1),
            "kernels": ["hvk_k_secam_cells", "hvk_k_secam_est (new pictures)", "hvk_k_secam_walk<0 / 1> (hvk_k_secam_chain where warm-up lines are walked)", "hvk_k_secam_check", "hvk_k_secam_redo (lines that started wrong)"] + names_s,
            "note": "lines (of the timed steps): worked on from derived entry states / found to have started wrong / redone / frames sent through the host's chain; "
                    "the number of warm-up lines per start state follows the pictures (exactness rests on the check, not on it) and has settled over the untimed blocks",
        }

    configs = None
    if N == 1 and not args.no_configs:
        ksteps = max(10, min(args.steps, 100))
        configs = {
            "1_pal_baseband": case_section(H, g, torch, "pal_bb", F, ksteps, 3, local_rank, stream, "config 1"),
            "3_ntsc_m": case_section(H, g, torch, "m_full", F, ksteps, 3, local_rank, stream, "config 3"),
            "4_secam_l_teletext_device": case_section(H, g, torch, "l_tt", F, 5, 2, local_rank, stream, "config 4 (raw packets)",
                                                      stage_every_step=True, teletext=True),
            "4_secam_l_teletext_noaudio_device": case_section(H, g, torch, "l_tt", F, 5, 16, local_rank, stream, "config 4 --noaudio (raw packets)",
                                                              stage_every_step=True, teletext=True, noaudio=True),
            "4_secam_l_teletext_demo_tti_dropin": dropin_section(["-m", "l", "-s", "16000000", "--filter", "--teletext", "@REF@/demo.tti"], pin_clock=True),
            "2_noaudio": case_section(H, g, torch, "i_vsb", F, ksteps, 3, local_rank, stream, "config 2 --noaudio", fresh_e2e=True),
            "2_noaudio_dropin": dropin_section(["-m", "i", "-s", "16000000", "--filter", "--noaudio"], devnull_s=5),
            "2_dropin": dropin_section(["-m", "i", "-s", "16000000", "--filter"]),
        }
        for k2, v2 in configs.items():
            if v2:
                log("%s: %s Msamples/s" % (k2, v2.get("Msamples_per_s")))

    # ---- the several-devices path in C: at N = 1 two engines on this one device (everything but the second PCIe link is
    # exercised), at N > 1 rank 0 drives one engine per device of the node after the ranks' own measurement ----
    cgroup = None
    if rank == 0 and not args.no_configs and not dry and os.environ.get("HVK_BENCH_CGROUP", "1") != "0":
        try:
            cgroup = c_group_section(H, g, [local_rank, local_rank] if N == 1 else list(range(N)), 64, 3, log)
        except SystemExit:
            raise
        except Exception as ex:        # (never lets the headline fall: reported instead)
            cgroup = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if N > 1:
        dist.barrier()

    hour = None
    if rank == 0 and N == 1 and not args.no_hour and not args.no_configs:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import hour as hour_mod
        hour = {"noaudio": hour_mod.run(H, g.frame("i_full"), g.audio, device=local_rank, sound=False, log=log)}
        if not args.no_hour_sound:
            # the metric configuration HAS sound (src/video.c:2259-2276): the hour as written, every block's sums and the
            # cumulative sha256 at 9 000 / 45 000 / 90 000 frames against the reference's -- two minutes, the host's serial FM chain
            hour["with_sound"] = hour_mod.run(H, g.frame("i_full"), g.audio, device=local_rank, sound=True, log=log)
        else:
            hour["with_sound"] = {"not_run": "--no-hour-sound (two minutes: the host's serial FM chain over 57.6 G samples)"}

    if rank == 0:
        names = e.kernel_names()
        one_kernel = len(names) == 1                        # hvk_k_direct (picture planes): the whole render in one kernel
        samples = F * FS                                    # per launch (one launch per kernel per step)
        alg = BYTES_PER_SAMPLE * samples
        tj = {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("frames") != F:
                    tj = {}
            except Exception:
                tj = {}

        def hbm_roofline(name, ms, n, key):
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tj.get(key),
                    "algorithmic_bytes_per_launch": int(alg), "launches_per_step": 1,
                    "avg_launch_ms": round(ms, 4), "launches_timed": int(n)}

        # the whole step against the same roofline: what the PATH achieves (kernels back to back, launch gaps, the gather)
        path_ach = BYTES_PER_SAMPLE * samples_per_step / (ms_per_step * 1e-3) / 1e9 / N
        if one_kernel:
            roof = hbm_roofline(names[0], filter_ms, n_f, "hvk_k_direct_bytes_per_launch")
            kernels = {"one_kernel": True, names[0]: round(filter_ms, 4)}
            other = None
        else:
            roof = hbm_roofline(names[-1], filter_ms, n_f, "hvk_k_filter_bytes_per_launch")
            kernels = {"one_kernel": False, names[0]: round(raster_ms, 4), names[-1]: round(filter_ms, 4),
                       "note": "average per launch; the kernels of a step run back to back on one stream"}
            # the raster kernel writes 2 B per sample and is bound by vector-ALU issue, not by HBM: no HBM fraction for it
            other = {"bound": "valu", "kernel": names[0], "avg_launch_ms": round(raster_ms, 4), "launches_timed": int(n_r),
                     "algorithmic_bytes_per_launch": int(2 * samples), "traffic": tj.get("hvk_k_raster_bytes_per_launch"),
                     "note": "VALU-issue bound (profiles/): its time is not an HBM figure"}
        roof["path_frac"] = round(path_ach / HBM_PEAK_GBS, 4)
        roof["path_achieved"] = round(path_ach, 1)
        roof["path_note"] = "4 B x samples of a step / ms_per_step / n_gpus against the same 8 TB/s: the fraction the whole path achieves per GPU"
        roof_also = True
        # the handful of numbers a reader of the headline wants beside it (their sections below have the detail)
        def _g(d, *keys):
            for k_ in keys:
                if not isinstance(d, dict) or k_ not in d:
                    return None
                d = d[k_]
            return d
        also = {
            "new_pictures_every_frame_table_levels_Msamples_per_s": _g(moving, "new_pictures_every_frame", "table_levels_Msamples_per_s"),
            "new_pictures_every_frame_computed_levels_Msamples_per_s": _g(moving, "new_pictures_every_frame", "computed_levels_Msamples_per_s"),
            "new_pictures_every_frame_ntsc_m_table_levels_Msamples_per_s": _g(moving, "new_pictures_every_frame", "ntsc_m", "table_levels_Msamples_per_s"),
            "secam_l_test_card_Msamples_per_s": _g(secam, "Msamples_per_s"),
            "secam_l_pictures_change_every_frame_Msamples_per_s": _g(secam, "pictures_change_every_frame", "Msamples_per_s"),
            "secam_l_new_picture_every_frame_planes_too_Msamples_per_s": _g(secam, "new_picture_every_frame", "Msamples_per_s"),
            "config1_path_frac": _g(configs, "1_pal_baseband", "path_frac"), "config3_path_frac": _g(configs, "3_ntsc_m", "path_frac"),
            "config4_noaudio_device_path_frac": _g(configs, "4_secam_l_teletext_noaudio_device", "path_frac"),
            "config2_noaudio_path_frac": _g(configs, "2_noaudio", "path_frac"),
            "end_to_end_Msamples_per_s": _g(e2e, "Msamples_per_s"),
            "dropin_config2_Msamples_per_s": _g(configs, "2_dropin", "Msamples_per_s"), "dropin_config2_noaudio_Msamples_per_s": _g(configs, "2_noaudio_dropin", "Msamples_per_s"), "dropin_config2_noaudio_to_dev_null_Msamples_per_s": _g(configs, "2_noaudio_dropin", "to_dev_null_Msamples_per_s"),
            "one_hour_noaudio_wall_s": _g(hour, "noaudio", "wall_s"),
            "one_hour_with_sound_wall_s": _g(hour, "with_sound", "wall_s"),
            "one_hour_with_sound_gate": _g(hour, "with_sound", "gate"),
            "c_group_noaudio_host_direct_Msamples_per_s": _g(cgroup, "noaudio", "host_direct_Msamples_per_s"),
            "c_group_noaudio_gathered_Msamples_per_s": _g(cgroup, "noaudio", "gathered_on_root_device_Msamples_per_s"),
            "c_group_with_sound_host_direct_Msamples_per_s": _g(cgroup, "with_sound", "host_direct_Msamples_per_s"),
            "c_group_gather_backend": _g(cgroup, "gather_backend"),
            "render_and_d2h_Msamples_per_s": _g(e2e, "render_and_d2h_Msamples_per_s"),
        }
        res = {
            "metric": "IQ Msamples/s (PAL-I AM-VSB, 16 MHz SR)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16 data, int32 accumulate",
            "data": "synthetic: built-in test card + 1 kHz tone (hacktv test source); " + ("every step stages (sound chains, host pre-pass, H2D) and renders the NEXT round of blocks. " if args.walk_rounds else "every step renders all frames of the staged block again. ") +
                    "The test card's picture planes (levels, low-passed chroma, burst: per-picture work, hvk_k_prep) are made once when the picture "
                    "is uploaded, OUTSIDE the timed loop, like the other side inputs; with a new picture on every frame that work is per frame: "
                    "moving_pictures.pictures_resident_planes_made_every_step",
            "ms_per_step_min": round(sub_ms[0], 4) if sub_ms else None,
            "ms_per_step_median": round(sub_ms[len(sub_ms) // 2], 4) if sub_ms else None,
            "settle": {"seconds": args.settle, "untimed_steps": settle_steps, "note": "the same launches, untimed, after the warm-up steps and before the clock starts"},
            "config": {
                "workload": "-m i -s 16000000 --filter test%s (PAL-I AM-VSB + 51-tap FIR, FM mono + NICAM)" % (" --noaudio" if args.noaudio else ""),
                "frames_per_gpu_per_step": F,
                "samples_per_step": samples_per_step,
                "also_measured": also,
                **{("also_" + k_): v_ for k_, v_ in also.items()},       # (the same scalars as keys of `config` itself: a record that keeps only flat keys keeps them)
                "parallelism": "frames block-cyclic over %d GPU(s)%s" % (N, (", one process driving an engine per device (hvk_group_*), blocks gathered on the root device in the step: " + cg_timed["gather_backend"]) if (cg_timed and "gather_backend" in cg_timed) else
                                                                         (", RCCL gather to rank 0 in the step, overlapped with the next block's render" if gather else "")),
            },
            "parity_gate": gate,
            "multi_gpu": {"ranks": 1, "backend": "none (one process, one device)", "c_group": cgroup,
                          "reassembly": "host-direct (every engine's block straight into the host stream buffer) and gathered on a root device (hvk_group_gather): both in c_group",
                          "sound_chains": "one recurrence over every sample of the stream: with sound the scaling curve is flat by construction (about 0.5 Gsamples/s, one host core), only --noaudio scales"} if N == 1 else {
                "c_group": cgroup,
                "c_group_timed": cg_timed,
                "value_from": "walk over rounds through the torch.distributed harness (--walk-rounds)" if args.walk_rounds else
                              ("the torch.distributed harness (c_group_timed FAILED: see c_group_timed.failed)" if (cg_timed and "failed" in cg_timed) else "") or
                              "c_group_timed: rank 0 drives one engine per device through hvk_group_* (C inside libhvk), K rounds of N blocks + hvk_group_gather between barriers over all ranks",
                "torch_harness": harness,
                "ranks": N, "world_size": dist.get_world_size(),
                "backend": (args.dry_run_backend + " (dry run: every rank on GPU 0, transport through host memory)") if dry else "nccl (RCCL); the sound chains' state between hosts: gloo",
                "walk_rounds": bool(args.walk_rounds), "walk_gate": walk_gate,
                "gathered_Msamples_per_s": round(value, 1) if (gather or cg_timed) else None,
                "sound_chains": "handed from rank to rank (hvk_sound_state_export / _import): every rank runs them over its own frames only",
                "gather_in_step": bool(gather), "gather_overlaps_render": bool(gather and not dry),
                "seam_gate": seam_gate,
                "render_only_Msamples_per_s": None if render_only is None else round(render_only, 1),
                "note": "value includes the reassembly of the contiguous stream on the root device (hvk_group_gather: one xGMI link per peer): it is bound by "
                        "the root's ingest (about 38 Gsamples/s per link), not by the kernels; render_only is the same steps without it. torch_harness: the same "
                        "sharding with one process per GPU over torch.distributed (the earlier rounds' path of record), kept as the cross-check. With sound on, a run "
                        "that also stages every round is bound by the serial host pre-pass (host_prepass), whatever the number of GPUs",
            },
            "roofline": roof,
            "kernels": kernels,
            "host_prepass": {
                "note": "staging one block before the clock: host audio control path (serial FM phasor chain on one core) and H2D of the side streams",
                "stage_s": round(t_stage, 3),
                "Msamples_per_s": round(F * FS / t_stage / 1e6, 1),
                "a2_stereo": a2_prepass(H, g.audio) if N == 1 and not args.no_moving else None,
            },
        }
        if roof_also:
            for k_ in ("secam_l_pictures_change_every_frame_Msamples_per_s", "secam_l_new_picture_every_frame_planes_too_Msamples_per_s",
                       "new_pictures_every_frame_table_levels_Msamples_per_s", "new_pictures_every_frame_computed_levels_Msamples_per_s",
                       "config3_path_frac", "config4_noaudio_device_path_frac", "config2_noaudio_path_frac", "one_hour_with_sound_wall_s",
                       "c_group_noaudio_host_direct_Msamples_per_s", "c_group_noaudio_gathered_Msamples_per_s"):
                res["roofline"]["also_" + k_] = also.get(k_)
        if other:
            res["roofline_other_kernel"] = other
        if e2e:
            res["end_to_end"] = e2e
        if moving:
            res["moving_pictures"] = moving
        if secam:
            res["secam_l"] = secam
        if configs:
            res["baseline_configs"] = configs
        if hour:
            res["5_one_hour"] = hour
        if not args.no_cpu_baseline and N == 1:
            res["cpu_baseline"] = cpu_baseline(log)
        print(json.dumps(res), flush=True)

    e.close()
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
