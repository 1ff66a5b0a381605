#!/usr/bin/env python3
"""tools/bench_sections.py -- the sections of bench.py's sidecar (bench_detail.json): the other BASELINE
configurations, pictures that change on every frame, SECAM-L, the drop-in binary, the C group with two engines on one
device, the one-hour run. bench.py's headline does not depend on any of them; quick_sections() runs by default (a few
seconds each, every one gated against the reference CLI run in the same job), full_sections() under --full (minutes).
No torch: output buffers and streams are the engine's own."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from bench import HBM_PEAK_GBS, BYTES_PER_SAMPLE, SAMPLE_RATE, MODE, ref_stream_sha, clean_env  # noqa: E402


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



def time_steps(step, sync, warmup, steps, settle=0.0):
    """`warmup` untimed calls of step(), then -- settle > 0 -- the same calls untimed for that many seconds (sustained clocks: a
    section whose timed region is a few milliseconds otherwise sees the first moments after an idle period, as the headline's
    --settle says), then `steps` timed ones between two sync()s: seconds per step."""
    for _ in range(warmup):
        step()
    sync()
    t_set = time.perf_counter()
    while time.perf_counter() - t_set < settle:
        for _ in range(10):
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



def case_section(H, g, case, F, steps, warmup, device, label, stage_every_step=False, teletext=False, fresh_e2e=False, noaudio=False):
    """One BASELINE configuration as a bench section: golden case `case` (its preset edits and CLI flags), F-frame blocks.
    Gate: every sample of the first block == the unmodified reference CLI's output for the same flags, run in this job.
    Then `steps` steps: launches of the staged block (inputs resident), or stage + launch of a fresh block each
    (stage_every_step: SECAM, whose colour chain runs when a block is staged)."""
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
    e.frame_upload(0, g.frame(case))
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
    e.launch()
    e.sync()
    got = hashlib.sha256(util.stream_bytes(e.fetch(0, F * fs), real)).hexdigest()
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
        e.launch()

    dt = time_steps(step, e.sync, warmup, steps, settle=0.0 if stage_every_step else 0.3)
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
        e.sync()
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
    ref = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    hvk = os.path.join(ROOT, "oracle", "_ref", "hacktv_hvk")
    pin = os.path.join(ROOT, "oracle", "_ref", "pin_time.so")
    if not (os.path.exists(ref) and os.path.exists(hvk)):
        return None
    env = clean_env({"HVK_BATCH": "32"})
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



def moving_section(H, g, F, FS, device, log):
    """Pictures that change every frame (the 7 B/sample regime): F new pictures per step, uploaded inside the
    timed loop (pinned ring, asynchronous copies), --noaudio so that the serial sound pre-pass does not hide what is being
    measured; beside it the same launches with the pictures resident."""
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
    em = H.Engine(H.preset(MODE, H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=device, max_frames=Fm)
    slots = list(range(Fm))

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
        em.launch()

    for k in range(2):
        mstep(k, True)
    em.sync()
    ksteps = 5
    t0 = time.perf_counter()
    for k in range(ksteps):
        mstep(2 + k, True)
    em.sync()
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(ksteps):
        mstep(2 + ksteps + k, False)
    em.sync()
    t_res = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(ksteps):
        mstep(2 + ksteps + k, 3)
    em.sync()
    t_prep = time.perf_counter() - t0
    for k in range(2):
        mstep(2 + 2 * ksteps + k, 2)
    em.sync()
    t0 = time.perf_counter()
    for k in range(ksteps):
        mstep(4 + 2 * ksteps + k, 2)
    em.sync()
    t_pin = time.perf_counter() - t0
    fused_used = em.fused_launches()

    def new_pictures(levels, fused, card):
        """Fm new pictures per step, resident in HBM: through the picture planes (HVK_FUSED=0: hvk_k_prep8 + hvk_k_direct) or
        from the pixels in one kernel (hvk_k_fused, what the engine takes by itself when most of a block's pictures are new)."""
        os.environ["HVK_FUSED"] = "1" if fused else "0"
        try:
            ex = H.Engine(H.preset(MODE, H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=device, max_frames=Fm)
        finally:
            del os.environ["HVK_FUSED"]
        ex.set_levels(levels)
        for i in range(Fm):
            ex.frame_upload(i, np.roll(g.frame("i_full"), 13 * i, axis=1) if card else pics[i % len(pics)])
        nxt = [0]

        def one():
            ex.planes_refresh(slots); ex.stage(nxt[0] * Fm, 1, Fm, slots=slots); ex.launch()
            nxt[0] += 1
        dt_ = time_steps(one, ex.sync, 2, ksteps * 2)
        nf = ex.fused_launches()
        ex.close()
        return round(Fm * FS / dt_ / 1e6, 1), nf

    def new_pictures_m(levels):
        """The same at BASELINE config 3's geometry (-m m -s 13500000 --filter --noaudio: 858-sample lines, 11-tap chroma): the one kernel
        from the pixels exists for 1024-sample lines only, so new pictures go through hvk_k_prep8 + hvk_k_direct there."""
        ex = H.Engine(H.preset("m", H.FLAG_FILTER | H.FLAG_NOAUDIO), 13500000, device=device, max_frames=Fm)
        ex.set_levels(levels)
        base = g.frame("m_full")
        rr = np.random.default_rng(2)
        for i in range(Fm):
            pic = np.roll(base, 13 * i, axis=1)
            if levels == 2:
                pic = (pic ^ (rr.integers(0, 4, base.shape, dtype=np.uint32) * np.uint32(0x010101))).astype(np.uint32)     # (low-bit noise: many colours)
            ex.frame_upload(i, pic)
        fsm = ex.info["frame_samples"]
        nxt = [0]

        def one():
            ex.planes_refresh(slots); ex.stage(nxt[0] * Fm, 1, Fm, slots=slots); ex.launch()
            nxt[0] += 1
        dt_ = time_steps(one, ex.sync, 2, ksteps * 2)
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
    return moving


def secam_section(H, g, F, FS, device, log):
    """SECAM-L (BASELINE config 4's mode): the colour sub-carrier's line-to-line chain runs on the device when a block is
    staged (hvk_secam.hip), so here a step is stage + launch of a fresh block; beside it the host's serial chain on one
    short block. A frame that went through the host's chain in a device section is a FAILURE (host_frames != 0)."""
    def secam_run(Fs, ksteps, wsteps=16, pics=None, refresh=False):
        # (the warm-up steps also let the number of warm-up LINES per start state settle: it follows the pictures, one
        # line down per clean block, two up per block with a wrong start -- hvk_engine.cpp)
        es = H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=device, max_frames=Fs)
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
    eh = H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), SAMPLE_RATE, device=device, max_frames=8)
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
    for name_, st_ in (("test card", st_big), ("blocks of F", st), ("pictures change", st_mov), ("new pictures", st_new)):
        if st_.get("host_frames"):
            raise SystemExit("SECAM section '%s': %d frames fell back to the host's chain (hvk_secam.c) -- the device chain must carry them" % (name_, st_["host_frames"]))
    return secam


def _g(d, *keys):
    for k_ in keys:
        if not isinstance(d, dict) or k_ not in d:
            return None
        d = d[k_]
    return d


def quick_sections(H, g, args, res, log):
    """The other BASELINE configurations as short sections (a block gated against the reference CLI, then a few timed
    steps): 1 (PAL baseband), 2 without sound, 3 (NTSC-M), 4 without sound (SECAM-L + teletext: the device's share of it).
    Results into res["baseline_configs"], one scalar each into res["also"]."""
    F, dev = args.frames, args.device
    k = max(10, min(args.steps, 50))
    configs = {
        "1_pal_baseband": case_section(H, g, "pal_bb", F, k, 3, dev, "config 1"),
        "2_noaudio": case_section(H, g, "i_vsb", F, k, 3, dev, "config 2 --noaudio", fresh_e2e=True),
        "3_ntsc_m": case_section(H, g, "m_full", F, k, 3, dev, "config 3"),
        "4_secam_l_teletext_noaudio_device": case_section(H, g, "l_tt", F, 20, 16, dev, "config 4 --noaudio (raw packets)",
                                                          stage_every_step=True, teletext=True, noaudio=True),
    }
    hf = _g(configs, "4_secam_l_teletext_noaudio_device", "secam_lines", "host_frames")
    if hf:
        raise SystemExit("config 4: %d frames fell back to the host's SECAM chain (hvk_secam.c) -- the device chain must carry them" % hf)
    for k2, v2 in configs.items():
        log("%s: %s Msamples/s (path_frac %s)" % (k2, v2.get("Msamples_per_s"), v2.get("path_frac")))
    res["baseline_configs"] = configs
    res["also"].update({
        "config1_pal_baseband_Msamples_per_s": _g(configs, "1_pal_baseband", "Msamples_per_s"),
        "config2_noaudio_Msamples_per_s": _g(configs, "2_noaudio", "Msamples_per_s"),
        "config2_noaudio_path_frac": _g(configs, "2_noaudio", "path_frac"),
        "config2_noaudio_fresh_block_end_to_end_Msamples_per_s": _g(configs, "2_noaudio", "fresh_block_end_to_end_Msamples_per_s"),
        "config3_ntsc_m_Msamples_per_s": _g(configs, "3_ntsc_m", "Msamples_per_s"),
        "config3_path_frac": _g(configs, "3_ntsc_m", "path_frac"),
        "config4_secam_noaudio_device_Msamples_per_s": _g(configs, "4_secam_l_teletext_noaudio_device", "Msamples_per_s"),
        "config4_secam_noaudio_device_path_frac": _g(configs, "4_secam_l_teletext_noaudio_device", "path_frac"),
    })


def full_sections(H, g, args, res, log):
    """--full: minutes. Moving pictures, SECAM in depth, config 4 with sound and through the drop-in binary, the drop-in
    binary on the metric configuration, two engines on this one device through hvk_group_*, the one-hour run."""
    F, dev = args.frames, args.device
    FS = 640000
    res["moving_pictures"] = moving_section(H, g, F, FS, dev, log)
    res["secam_l"] = secam_section(H, g, F, FS, dev, log)
    cfg = res.setdefault("baseline_configs", {})
    cfg["4_secam_l_teletext_device"] = case_section(H, g, "l_tt", F, 5, 2, dev, "config 4 (raw packets)", stage_every_step=True, teletext=True)
    cfg["4_secam_l_teletext_demo_tti_dropin"] = dropin_section(["-m", "l", "-s", "16000000", "--filter", "--teletext", "@REF@/demo.tti"], pin_clock=True)
    cfg["2_noaudio_dropin"] = dropin_section(["-m", "i", "-s", "16000000", "--filter", "--noaudio"], devnull_s=5)
    cfg["2_dropin"] = dropin_section(["-m", "i", "-s", "16000000", "--filter"])
    res["host_prepass"]["a2_stereo"] = a2_prepass(H, g.audio)
    try:
        res["c_group_two_engines_one_device"] = c_group_section(H, g, [dev, dev], 64, 3, log)
    except SystemExit:
        raise
    except Exception as ex:
        res["c_group_two_engines_one_device"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    import hour as hour_mod
    hour = {"noaudio": hour_mod.run(H, g.frame("i_full"), g.audio, device=dev, sound=False, log=log)}
    hour["with_sound"] = hour_mod.run(H, g.frame("i_full"), g.audio, device=dev, sound=True, log=log)
    res["5_one_hour"] = hour
    res["also"].update({
        "new_pictures_every_frame_table_levels_Msamples_per_s": _g(res, "moving_pictures", "new_pictures_every_frame", "table_levels_Msamples_per_s"),
        "new_pictures_every_frame_computed_levels_Msamples_per_s": _g(res, "moving_pictures", "new_pictures_every_frame", "computed_levels_Msamples_per_s"),
        "new_pictures_every_frame_ntsc_m_table_levels_Msamples_per_s": _g(res, "moving_pictures", "new_pictures_every_frame", "ntsc_m", "table_levels_Msamples_per_s"),
        "secam_l_test_card_Msamples_per_s": _g(res, "secam_l", "Msamples_per_s"),
        "secam_l_pictures_change_every_frame_Msamples_per_s": _g(res, "secam_l", "pictures_change_every_frame", "Msamples_per_s"),
        "secam_l_new_picture_every_frame_Msamples_per_s": _g(res, "secam_l", "new_picture_every_frame", "Msamples_per_s"),
        "dropin_config2_Msamples_per_s": _g(cfg, "2_dropin", "Msamples_per_s"),
        "dropin_config2_noaudio_Msamples_per_s": _g(cfg, "2_noaudio_dropin", "Msamples_per_s"),
        "one_hour_noaudio_wall_s": _g(hour, "noaudio", "wall_s"),
        "one_hour_with_sound_wall_s": _g(hour, "with_sound", "wall_s"),
    })
    for k_, v_ in res["also"].items():
        log("%s: %s" % (k_, v_))
