#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s of the MI355X composite-video -> IQ engine.

Workload (BASELINE.json configs[1], the one the metric is quoted on):
  PAL System I, AM-VSB with the 51-tap FIR (`-m i -s 16000000 --filter test`),
  16 MHz sample rate, built-in test card, FM mono + NICAM-728 sound on.

A "step" is one pass of the hot path over one block of F whole frames per GPU,
through the C ABI of libhvk (hvk_launch): hvk_k_direct, ONE kernel that composes
every raster sample from the picture's planes and the sub-carrier, filters it,
adds the sound carriers and NICAM and stores the int16 I/Q. The block's side
inputs (sound-carrier stream, NICAM symbols, the test card's picture planes) are
staged into HBM before the clock starts; every step renders all F frames again.
Before any number is taken EVERY sample of the block is compared with the
unmodified reference CLI run in the same job (and with its committed digest).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--full]

N = 1            one process, one engine, no torch (HIP through libhvk only).
N > 1, as typed  one process drives an engine per device through hvk_group_* (C inside
                 libhvk): blocks dealt round-robin, a step = every engine's launch + the
                 gather of the round onto the root device (RCCL between distinct devices).
                 --devices 0,0 puts several engines on one GPU (tests).
N > 1, torchrun  (WORLD_SIZE = N: what the driver launches) one rank per GPU, blocks
                 block-cyclic over the ranks, RCCL gather to rank 0 inside the step,
                 barrier + synchronize either side, maximum over ranks (tools/bench_multi.py).

stdout carries ONE line, the last thing printed: a compact JSON object (< 4 KB) with the
contract's keys plus "roofline" and "cpu_baseline". Everything else -- the other BASELINE
configurations, moving pictures, SECAM, the drop-in binary, the one-hour run (--full) --
goes to the sidecar bench_detail.json (--detail-out) and to stderr.
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_SAMPLE = 4        # algorithmic traffic: one int16 I + one int16 Q written per sample
SAMPLE_RATE = 16000000
MODE = "i"
METRIC = "IQ Msamples/s (PAL-I AM-VSB, 16 MHz SR)"
LINE_LIMIT = 4096           # bytes of the final stdout line (tests/test_bench_line.py)

# the keys of the final line, in order; everything else a result dict holds stays in the sidecar
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "parity_gate", "roofline", "cpu_baseline", "multi_gpu", "also", "detail")
CONFIG_KEYS = ("workload", "frames_per_gpu_per_step", "samples_per_step", "parallelism")
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
             "avg_launch_ms", "launches_timed", "path_frac")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_line(res):
    """The one line stdout carries: the contract's keys, `roofline`, `cpu_baseline`, a handful of scalars under `also`.
    Strings are clipped, nested detail is dropped; never longer than LINE_LIMIT bytes."""
    out = {}
    for k in LINE_KEYS:
        if k not in res or res[k] is None and k not in ("vs_baseline",):
            continue
        v = res[k]
        if k == "config":
            v = {kk: (_clip(v[kk], 160) if isinstance(v[kk], str) else v[kk]) for kk in CONFIG_KEYS if kk in v}
        elif k == "roofline":
            v = {kk: v[kk] for kk in ROOF_KEYS if kk in v}
        elif k == "cpu_baseline":
            v = {kk: (_clip(v[kk], 120) if isinstance(v[kk], str) else v[kk]) for kk in CPU_KEYS if kk in v}
        elif k in ("multi_gpu", "also"):
            v = {kk: (_clip(vv, 120) if isinstance(vv, str) else vv) for kk, vv in v.items()
                 if isinstance(vv, (int, float, str, bool)) or vv is None}
        elif k in ("data", "parity_gate"):
            v = _clip(v, 240)
        out[k] = v
    line = json.dumps(out, separators=(",", ":"))
    while len(line.encode()) > LINE_LIMIT and out.get("also"):
        out["also"].popitem()
        line = json.dumps(out, separators=(",", ":"))
    if len(line.encode()) > LINE_LIMIT:
        raise AssertionError("bench line is %d bytes" % len(line.encode()))
    return line


def emit(res, detail_path, log):
    """Sidecar first (the whole result), then the compact line as the last thing on stdout."""
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(res, f, indent=1)
            res.setdefault("detail", os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path)
        except OSError as ex:
            log("could not write %s: %s" % (detail_path, ex))
    sys.stderr.flush()
    sys.stdout.write(compact_line(res) + "\n")
    sys.stdout.flush()


def clean_env(extra=None):
    """For child processes: under rocprofv3 a child would inherit the profiler's LD_PRELOAD and tool settings."""
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS", "ROCPROFILER"))}
    env.update(extra or {})
    return env


def ref_stream_sha(mode, sr, cli_flags, first, count, frame_bytes, env_extra=None):
    """sha256 of frames [first, first + count) of the unmodified reference CLI's output for these flags, run now
    (None: oracle/_ref/hacktv_ref is not there)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    if not os.path.exists(ref_bin):
        return None
    p = subprocess.Popen([ref_bin, "-m", mode, "-s", str(sr)] + list(cli_flags) + ["-o", "-", "test"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=clean_env(env_extra))
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


def committed_digest(nframes):
    """sha256 of the first nframes frames of the metric configuration's reference stream (tests/golden/ref_long.json)."""
    long_file = os.path.join(ROOT, "tests", "golden", "ref_long.json")
    if not os.path.exists(long_file):
        return None
    return json.load(open(long_file))["i_full"]["sha256_at_frames"].get(str(nframes))


def cpu_baseline(log):
    """The reference CLI on the host cores: steady state = (t[11 s of signal] - t[1 s]) / 10 s, which strips its
    ~0.5 s table build; median of three. About 12 s of CPU work."""
    ref = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")
    flags = ["-m", MODE, "-s", str(SAMPLE_RATE), "--filter"]
    if os.path.exists(ref):
        def run(seconds):
            nbytes = seconds * SAMPLE_RATE * 4
            t = time.perf_counter()
            p = subprocess.Popen([ref] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=clean_env())
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
                    "sample": "hacktv_ref -m i -s 16000000 --filter: (t[11 s signal]-t[1 s])/10 s, median of 3; 3 busy threads of %d cores" % (os.cpu_count() or 0)}
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


def traffic_per_launch(F, key="hvk_k_direct_bytes_per_launch"):
    """HBM bytes per launch from the PMC passes kept in profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, the gfx950 correction applied: tools/pmc_summary.py); None when the file is for another block size."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
        return tj.get(key) if tj.get("frames") == F else None
    except Exception:
        return None


def hbm_roofline(kernel, avg_ms, launches, samples_per_launch, traffic):
    alg = BYTES_PER_SAMPLE * samples_per_launch
    ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": int(alg),
            "avg_launch_ms": round(avg_ms, 4), "launches_timed": int(launches)}


def device_sync():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipDeviceSynchronize()


def headline_one(args, log):
    """N = 1: one engine on one device. Returns the result dict (every key; compact_line() picks)."""
    import numpy as np   # noqa: F401
    import hacktv_amd as H
    import util

    g = util.Golden()
    flags = H.FLAG_FILTER | (H.FLAG_NOAUDIO if args.noaudio else 0)
    conf = H.preset(MODE, flags)
    F = args.frames
    e = H.Engine(conf, SAMPLE_RATE, device=args.device, max_frames=F)
    FS = e.info["frame_samples"]
    e.frame_upload(0, g.frame("i_full"))
    while e.audio_needed(F) > 0:
        e.audio_write(g.audio)
    t0 = time.perf_counter()
    e.stage(0, 1, F)
    e.sync()
    t_stage = time.perf_counter() - t0
    log("staged %d frames (host control path + H2D) in %.2f s = %.1f Msamples/s" % (F, t_stage, F * FS / t_stage / 1e6))

    # ---- parity gate before any number: EVERY sample of the block against the unmodified reference ----
    e.launch()
    e.sync()
    cli = ["--filter"] + (["--noaudio"] if args.noaudio else [])
    got = hashlib.sha256(e.fetch(0, F * FS).tobytes()).hexdigest()
    want, how = ref_stream_sha(MODE, SAMPLE_RATE, cli, 0, F, FS * 4), None
    if want is not None:
        how = "hacktv_ref run in this job"
        if got != want:
            raise SystemExit("parity gate failed: frames 0..%d differ from the reference CLI's output" % (F - 1))
    com = None if args.noaudio else committed_digest(F)
    if com is not None:
        if got != com:
            raise SystemExit("parity gate failed: the first %d frames differ from the committed reference digest" % F)
        how = (how + " + committed digest") if how else "committed digest"
    if how is None:
        raise SystemExit("parity gate: neither oracle/_ref/hacktv_ref nor a committed digest for %d frames -- refusing to report a number" % F)
    gate = "all %d frames x %d samples sha256 == %s" % (F, FS, how)
    log("parity gate ok: " + gate)

    # ---- warm-up, then --settle seconds of the same launches untimed (sustained clocks), then the timed region ----
    for _ in range(args.warmup):
        e.launch()
    e.sync()
    t_set, settle_steps = time.perf_counter(), 0
    while time.perf_counter() - t_set < args.settle:
        for _ in range(20):
            e.launch()
        e.sync()
        settle_steps += 20
    device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e.launch()
    e.sync()
    device_sync()
    dt = time.perf_counter() - t0

    # the kernel's own time, for the roofline object: the same steps once more with HIP events recorded around every
    # launch on the launch stream (recording costs a little: not inside the region `value` comes from)
    e.timing_enable(True)
    for _ in range(args.steps):
        e.launch()
    e.sync()
    raster_ms, n_r = e.timing_read(0)
    kern_ms, n_k = e.timing_read(1)
    e.timing_enable(False)

    # the spread: five more runs of a fifth of the steps each
    sub = []
    nsub = max(10, args.steps // 5)
    for _ in range(5):
        e.sync()
        ts = time.perf_counter()
        for _ in range(nsub):
            e.launch()
        e.sync()
        sub.append((time.perf_counter() - ts) / nsub * 1e3)
    sub.sort()

    names = e.kernel_names()
    samples = F * FS
    value = samples * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    roof = hbm_roofline(names[-1], kern_ms, n_k, samples, traffic_per_launch(F))
    roof["path_frac"] = round(BYTES_PER_SAMPLE * samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if len(names) > 1:
        roof["other_kernel"] = {"kernel": names[0], "avg_launch_ms": round(raster_ms, 4), "launches_timed": int(n_r)}
    res = {
        "metric": METRIC, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16 data, int32 accumulate",
        "data": "synthetic: built-in test card + 1 kHz tone (hacktv test source); every step renders all frames of the staged block again; "
                "side inputs and the card's picture planes resident in HBM (made once, outside the clock)",
        "config": {"workload": "-m i -s 16000000 --filter test%s (PAL-I AM-VSB + 51-tap FIR, FM mono + NICAM)" % (" --noaudio" if args.noaudio else ""),
                   "frames_per_gpu_per_step": F, "samples_per_step": samples, "parallelism": "one engine on one GPU"},
        "parity_gate": gate,
        "roofline": roof,
        "ms_per_step_min": round(sub[0], 4), "ms_per_step_median": round(sub[len(sub) // 2], 4),
        "settle": {"seconds": args.settle, "untimed_steps": settle_steps},
        "host_prepass": {"stage_s": round(t_stage, 3), "Msamples_per_s": round(F * FS / t_stage / 1e6, 1),
                         "note": "staging one block before the clock: the host's serial sound chains (one core) + H2D of the side streams"},
        "also": {"stage_with_sound_Msamples_per_s": round(F * FS / t_stage / 1e6, 1)},
    }
    log("value %.1f Msamples/s, %.4f ms per step; %s %.4f ms per launch = %.4f of the HBM roofline" %
        (value, ms_per_step, names[-1], kern_ms, roof["frac"]))

    # ---- one FRESH block end to end (the PCIe-inclusive rate, never `value`) ----
    host_out = e.host_buffer(F * FS)
    e.fetch_wait(e.fetch_async(host_out, 0, F * FS))
    t0 = time.perf_counter()
    while e.audio_needed(2 * F) > 0:
        e.audio_write(g.audio)
    e.stage(F, 1, F)
    e.sync()
    t1 = time.perf_counter()
    e.launch()
    e.fetch_wait(e.fetch_async(host_out, 0, F * FS))
    t2 = time.perf_counter()
    res["end_to_end"] = {"stage_s": round(t1 - t0, 4), "render_and_d2h_s": round(t2 - t1, 4),
                         "Msamples_per_s": round(F * FS / (t2 - t0) / 1e6, 1),
                         "render_and_d2h_Msamples_per_s": round(F * FS / (t2 - t1) / 1e6, 1),
                         "note": "one fresh block, nothing overlapped: host sound chains + H2D, then render + D2H into page-locked memory"}
    res["also"]["fresh_block_end_to_end_Msamples_per_s"] = res["end_to_end"]["Msamples_per_s"]
    res["also"]["render_and_d2h_Msamples_per_s"] = res["end_to_end"]["render_and_d2h_Msamples_per_s"]
    e.close()

    # ---- the other BASELINE configurations, briefly (each gated against the reference CLI run in this job) ----
    if not args.no_configs:
        import bench_sections as S
        S.quick_sections(H, g, args, res, log)
    if args.full:
        import bench_sections as S
        S.full_sections(H, g, args, res, log)
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(log)
        log("cpu_baseline: %s %s (%s)" % (res["cpu_baseline"]["value"], res["cpu_baseline"]["unit"], res["cpu_baseline"]["kind"]))
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--device", type=int, default=0, help="N = 1: the HIP device")
    ap.add_argument("--devices", default=None, help="N > 1 as typed: one HIP device ordinal per engine, e.g. 0,0 (default 0..N-1)")
    ap.add_argument("--full", action="store_true", help="also: moving pictures, SECAM, the drop-in binary, the C group at N = 1, the one-hour run (minutes)")
    ap.add_argument("--no-configs", action="store_true", help="skip the brief sections for BASELINE configs 1, 3, 4 and --noaudio")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--noaudio", action="store_true", help="render the --noaudio variant instead (not the metric configuration)")
    ap.add_argument("--settle", type=float, default=0.6, help="seconds of untimed launches before the clock starts (sustained clocks)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"), help="sidecar with every section ('' = none)")
    # N > 1 under torch.distributed.run
    ap.add_argument("--no-gather", action="store_true", help="ranks: leave the RCCL reassembly out of the step")
    ap.add_argument("--walk-rounds", action="store_true", help="ranks: every step stages and renders the NEXT round of blocks (sound chains handed from rank to rank)")
    ap.add_argument("--dry-run-backend", default=None, help="ranks: gloo = every rank on GPU 0, transport through host memory (tests)")
    ap.add_argument("--no-group", action="store_true", help="ranks: do not run the one-process C group beside the harness")
    ap.add_argument("--group-timeout", type=float, default=420.0, help="ranks: seconds the C group's child process may take")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if world > 1:
        import bench_multi
        res = bench_multi.headline_ranks(args, log)      # (None on ranks other than 0)
    elif args.gpus > 1 or args.devices:
        import bench_multi
        devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
        res = bench_multi.headline_group(args, devices, log)
    else:
        res = headline_one(args, log)
    if rank == 0 and res is not None:
        emit(res, args.detail_out, log)


if __name__ == "__main__":
    main()
