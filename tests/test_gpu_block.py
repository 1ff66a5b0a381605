"""Whole bench blocks and BASELINE config 4 as written, on the GPU, against the unmodified reference.

* The metric configuration (`-m i -s 16000000 --filter test`, FM + NICAM on) rendered in blocks of 128
  frames (what bench.py times) and of 37 (an odd batch), every sample hashed: against the committed
  digests of the reference CLI (tests/golden/ref_long.json, oracle/make_golden_long.py) and, where the
  reference binary is present (it travels to the GPU box), against its output produced in the same job.
* `-m l -s 16000000 --filter --teletext demo.tti` through the drop-in binary (the reference's main(),
  av_test.c, rf_file.c, teletext.c + the video.h shim + libhvk) with the wall clock pinned for both
  (oracle/pin_time.c)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
from conftest import require_ref

import hacktv_amd as H
import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
LONG = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_long.json")))


def _ref_digests(flags, frame_bytes, marks, env=None):
    """sha256 of the first m frames of the reference CLI's output, for every m in marks (None: no binary)."""
    exe = os.path.join(REF, "hacktv_ref")
    if not os.path.exists(exe):
        return None
    p = subprocess.Popen([exe] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
    h = hashlib.sha256()
    out, done = {}, 0
    for m in sorted(marks):
        want = m * frame_bytes - done
        while want > 0:
            chunk = p.stdout.read(min(want, 1 << 22))
            assert chunk, "reference ended early"
            h.update(chunk)
            want -= len(chunk)
            done += len(chunk)
        out[m] = h.copy().hexdigest()
    p.kill()
    p.wait()
    return out


@pytest.mark.parametrize("batches,env", [((128, 37), {}), ((37, 37, 37, 17), {}),
                                         ((128, 37), {"HVK_DIRECT": "0"}), ((37, 91), {"HVK_DIRECT": "0"})])
def test_whole_blocks_equal_the_reference(golden, batches, env, monkeypatch):
    """Every sample of 128-frame and 37-frame blocks with sound on: the one-kernel render from picture planes
    (default) and the raster + filter kernel pair (HVK_DIRECT=0)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    require_ref(os.path.join(REF, "hacktv_ref"))
    conf, sr = golden.conf("i_full")
    marks, total = [], 0
    for b in batches:
        total += b
        marks.append(total)
    want = LONG["i_full"]["sha256_at_frames"]
    live = _ref_digests(LONG["i_full"]["flags"], LONG["i_full"]["frame_bytes"], marks)
    h = hashlib.sha256()
    with H.Engine(conf, sr, device=0, max_frames=max(batches)) as e:
        e.frame_upload(0, golden.frame("i_full"))
        fs = e.info["frame_samples"]
        done = 0
        for b in batches:
            while e.audio_needed(b) > 0:
                e.audio_write(golden.audio)
            e.render(b)
            h.update(e.fetch(0, b * fs).tobytes())
            done += b
            got = h.copy().hexdigest()
            if str(done) in want:
                assert got == want[str(done)], "first %d frames differ from the committed reference digest" % done
            if live is not None:
                assert got == live[done], "first %d frames differ from the reference run in this job" % done
            assert str(done) in want or live is not None


def test_config4_teletext_from_demo_tti_with_the_clock_pinned():
    """BASELINE config 4 as written: the TTI page goes through the reference's own parser and packet
    scheduler (unchanged, in the drop-in binary), the shim hands the packets to the engine, the device
    renders them. time() is pinned for both binaries."""
    hvk = os.path.join(REF, "hacktv_hvk")
    ref = os.path.join(REF, "hacktv_ref")
    pin = os.path.join(REF, "pin_time.so")
    tti = os.path.join(REF, "demo.tti")
    for f in (hvk, pin, tti):
        require_ref(f)
    env = dict(os.environ, LD_PRELOAD=pin, TZ="UTC", HVK_BATCH="2")
    env.pop("HVK_PIN_TIME", None)
    flags = [f.replace("@REF@", REF) for f in LONG["l_tti"]["flags"]]
    fb = LONG["l_tti"]["frame_bytes"]
    nframes = 5

    def run(binary):
        p = subprocess.Popen([binary] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        out = bytearray()
        while len(out) < nframes * fb:
            chunk = p.stdout.read(nframes * fb - len(out))
            if not chunk:
                break
            out += chunk
        p.kill()
        p.wait()
        return bytes(out)

    got = run(hvk)
    assert len(got) == nframes * fb
    for m, d in LONG["l_tti"]["sha256_at_frames"].items():
        assert util.sha256(got[: int(m) * fb]) == d, "first %s frames differ from the committed digest of the reference" % m
    if os.path.exists(ref):
        want = run(ref)
        if got != want:
            a = np.frombuffer(got, np.int16).reshape(-1, 2)
            b = np.frombuffer(want, np.int16).reshape(-1, 2)
            bad = np.nonzero((a != b).any(axis=1))[0]
            raise AssertionError("%d samples differ, first at %d (line %d)" % (bad.size, bad[0], bad[0] // 1024))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _last_line(stdout):
    """The driver's view: the LAST line of stdout is the JSON object, and it is short."""
    lines = stdout.decode().splitlines()
    assert lines and lines[-1].startswith("{"), lines[-3:]
    assert len(lines[-1].encode()) <= 4096
    assert sum(1 for l in lines if l.startswith("{")) == 1
    return json.loads(lines[-1])


def test_bench_gpus_2_as_typed_runs_the_group_in_process():
    """`python3 bench.py --gpus 2 --devices 0,0 ...` exactly as typed (no torch.distributed.run): one process, two engines on
    the one GPU through hvk_group_*, round 0 gated against the reference CLI, the compact line printed last."""
    import sys
    require_ref(os.path.join(REF, "hacktv_ref"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--steps", "2", "--frames", "3", "--detail-out", ""]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    d = _last_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 2
    assert "sha256 == hacktv_ref" in d["parity_gate"]
    mg = d["multi_gpu"]
    assert mg["engines"] == 2 and mg["gather_backend"] and mg["render_only_Msamples_per_s"] > 0
    assert mg["host_direct_Msamples_per_s"] > 0 and mg["staged_every_round_Msamples_per_s"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["avg_launch_ms"] > 0


def test_bench_default_line_parses_and_carries_roofline_and_cpu_baseline():
    """N = 1 as the driver runs it (short: 3 frames, 5 steps): one compact JSON line, last on stdout, with `roofline` and
    `cpu_baseline`; the detail in the sidecar."""
    import sys
    import tempfile
    require_ref(os.path.join(REF, "hacktv_ref"))
    with tempfile.TemporaryDirectory() as td:
        side = os.path.join(td, "detail.json")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--frames", "3", "--settle", "0", "--detail-out", side]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
        d = _last_line(r.stdout)
        full = json.load(open(side))
    assert d["n_gpus"] == 1 and d["metric"].startswith("IQ Msamples/s") and d["unit"] == "Msamples/s"
    assert "sha256 == hacktv_ref" in d["parity_gate"]
    assert d["roofline"]["kernel"].startswith("hvk_k_direct") and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0
    assert d["config"]["frames_per_gpu_per_step"] == 3
    assert "baseline_configs" in full and full["baseline_configs"]["4_secam_l_teletext_noaudio_device"]["secam_lines"]["host_frames"] == 0


def test_a_failure_of_the_c_group_leaves_the_harness_value_and_says_so():
    """bench.py under torch.distributed.run: the one-process group beside the harness (a child process of rank 0) made to
    fail -- the other rank is not left at a barrier, the line is printed with the harness's gathered figure, and says so."""
    import sys
    require_ref(os.path.join(REF, "hacktv_ref"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "3",
           "--dry-run-backend", "gloo", "--no-cpu-baseline", "--detail-out", ""]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT, env=dict(os.environ, BENCH_FAIL_C_GROUP="1"))
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    d = _last_line(r.stdout)
    assert "asked to fail" in d["multi_gpu"]["c_group_failed"]
    assert abs(d["value"] - d["multi_gpu"]["gathered_Msamples_per_s"]) < 0.2 and d["value"] > 0
    assert "sha256 == reference CLI" in d["multi_gpu"]["seam_gate"]


@pytest.mark.parametrize("walk", [False, True])
def test_two_ranks_on_one_gpu_reassemble_the_reference_stream(walk):
    """bench.py's torch.distributed path with the engine in it: two ranks (both on GPU 0, gloo through host memory) stage,
    render and send block-cyclic blocks -- the serial sound chains handed from rank to rank, each running them over its own
    frames only; before timing anything rank 0 hashes the stream reassembled from two rounds -- block seams and round seams
    included -- against the reference CLI run in the same job, every rank hashes its timed block, and rank 0 the gathered
    round. The JSON line must carry the seam gate's verdict."""
    import sys
    require_ref(os.path.join(REF, "hacktv_ref"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "3",
           "--dry-run-backend", "gloo", "--no-cpu-baseline", "--detail-out", ""] + (["--walk-rounds"] if walk else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    d = _last_line(r.stdout)
    assert d["n_gpus"] == 2
    assert d["multi_gpu"]["seam_gate"] and "sha256 == reference CLI" in d["multi_gpu"]["seam_gate"]
    assert "sha256 ==" in d["parity_gate"]
    assert d["multi_gpu"]["world_size"] == 2 and d["multi_gpu"]["walk_rounds"] == walk
    if not walk:
        # `value` is the ranks' own (launch + gather between barriers); beside it the one-process group over the same devices
        assert abs(d["value"] - d["multi_gpu"]["gathered_Msamples_per_s"]) < 0.2
        assert d["multi_gpu"]["c_group_gathered_Msamples_per_s"] > 0 and d["multi_gpu"]["c_group_host_direct_Msamples_per_s"] > 0
        assert "sha256 == hacktv_ref" in d["multi_gpu"]["c_group_parity_gate"]
    if walk:
        # every step staged and rendered the next round, the sound chains went from rank to rank, and the last round
        # walked (round 3: frames 18 .. 20 on rank 0) still is the reference's
        assert "sha256 == reference CLI" in d["multi_gpu"]["walk_gate"]


def test_queued_read_back_into_page_locked_memory(golden):
    """hvk_fetch_async() / hvk_fetch_wait() / hvk_host_alloc(): batches read back behind the next batch's staging
    are the same samples hvk_fetch() hands over, and the reference's."""
    FS = 640000
    c = H.preset("i", H.FLAG_FILTER)
    with H.Engine(c, 16000000, max_frames=3) as e, H.Engine(c, 16000000, max_frames=3) as plain:
        bufs = [e.host_buffer(3 * FS), e.host_buffer(3 * FS)]
        for eng in (e, plain):
            eng.frame_upload(0, golden.frame("i_full"))
        tickets, got, want = [None, None], [], []
        for b in range(4):
            for eng in (e, plain):
                while eng.audio_needed(3 * (b + 1)) > 0:
                    eng.audio_write(golden.audio)
            if tickets[b & 1] is not None:
                e.fetch_wait(tickets[b & 1])
                got.append(bufs[b & 1].copy())
            e.stage(3 * b, 1, 3)                 # host pre-passes of batch b while batch b - 1 is still on its way
            e.launch()
            tickets[b & 1] = e.fetch_async(bufs[b & 1], 0, 3 * FS)
            plain.render(3)
            want.append(plain.fetch(0, 3 * FS))
        for b in (2, 3):
            e.fetch_wait(tickets[b & 1])
            got.append(bufs[b & 1].copy())
        for b in range(4):
            assert np.array_equal(got[b], want[b]), "batch %d" % b
        # and the reference's first frame (committed digest of the reference CLI's output)
        assert hashlib.sha256(got[0][:FS].tobytes()).hexdigest() == LONG["i_full"]["sha256_at_frames"]["1"]
    assert e.h is None


@pytest.mark.parametrize("case,sync", [("pal_fm", False), ("palfm_f14", False), ("pal_fm", True)])
def test_fm_video_read_back_through_the_engines_fm_thread(golden, monkeypatch, case, sync):
    """FM video: the phasor is a host pass over every sample (hvk_tail.c). Behind hvk_fetch_async() the engine's own
    thread runs it in the caller's buffer, job after job, while the caller stages the next batch (HVK_FM_SYNC=1: in the
    call, as before): batches of three and two frames, whole and in two pieces, against the reference CLI's digests
    and hvk_fetch()."""
    if sync:
        monkeypatch.setenv("HVK_FM_SYNC", "1")
    c = golden.cases[case]
    conf, sr = golden.conf(case)
    FS = c["width"] * c["lines"]
    nframes = 8                                  # (the digests hold the first two or three; hvk_fetch() all of them)
    with H.Engine(conf, sr, max_frames=3) as e, H.Engine(conf, sr, max_frames=3) as plain:
        bufs = [e.host_buffer(3 * FS), e.host_buffer(3 * FS)]
        for eng in (e, plain):
            eng.frame_upload(0, golden.frame(case))
        got, want, pending, done = [], [], [], 0
        b = 0
        while done < nframes:
            n = min(3 if b % 2 == 0 else 2, nframes - done)
            for eng in (e, plain):
                while eng.audio_needed(n) > 0:
                    eng.audio_write(golden.audio)
            e.stage(done, 1, n)                  # (nothing here waits for the FM thread's work on the batch before)
            for t_, buf_, cnt_ in pending:
                e.fetch_wait(t_)
            got += [buf_[:cnt_].copy() for _, buf_, cnt_ in pending]
            e.launch()
            buf = bufs[b & 1]
            if b % 2 == 0:
                pending = [(e.fetch_async(buf, 0, n * FS), buf, n * FS)]
            else:
                half = (n * FS) // 2 + 7
                pending = [(e.fetch_async(buf, 0, half), buf, half), (e.fetch_async(buf[half:], half, n * FS - half), buf[half:], n * FS - half)]
            plain.render(n)
            want.append(plain.fetch(0, n * FS))
            done += n
            b += 1
        for t_, buf_, cnt_ in pending:
            e.fetch_wait(t_)
        got += [buf_[:cnt_].copy() for _, buf_, cnt_ in pending]
        got, want = np.concatenate(got), np.concatenate(want)
        assert np.array_equal(got, want)
        for n in range(c["frames"]):
            assert util.sha256(util.stream_bytes(got[: (n + 1) * FS], c["real"])) == c["sha256_cumulative"][n], "frame %d" % (n + 1)


@pytest.mark.parametrize("flags", [["-m", "i", "-s", "16000000", "--filter"],
                                   ["-m", "l", "-s", "16000000", "--filter", "--vits", "--vitc"]])
def test_dropin_resident_set_does_not_grow_with_the_run(flags):
    """120 s of signal through the drop-in binary: its queues (32 kHz sound kept for line->audio, NICAM symbols,
    the SECAM chain's stores, the read-back ring) are emptied as the stream goes by -- after the first half minute of
    signal (one-off growth of the runtime's pools) the resident set stays where it is."""
    psutil = pytest.importorskip("psutil")
    exe = os.path.join(REF, "hacktv_hvk")
    require_ref(exe)
    S = 120
    n = S * 16000000 * 4
    p = subprocess.Popen([exe] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         env=dict(os.environ, HVK_BATCH="16"))
    try:
        ps = psutil.Process(p.pid)
        got, rss = 0, []
        while got < n:
            chunk = p.stdout.read(min(1 << 24, n - got))
            assert chunk, "the drop-in ended early"
            got += len(chunk)
            if len(rss) < got // (n // 8 + 1) + 1:
                rss.append(ps.memory_info().rss >> 20)
    finally:
        p.kill()
        p.wait()
    assert len(rss) == 8 and rss[-1] <= rss[3] + 16, rss


@pytest.mark.parametrize("shape", [(576, 832), (600, 900), (300, 500)])
def test_pictures_uploaded_from_page_locked_memory(golden, shape):
    """hvk_frame_upload_pinned(): a picture that lies in page-locked memory goes to its slot by one strided copy --
    exact size, larger than the active area (centre crop) and smaller (borders): the same samples as through the
    ordinary upload."""
    rng = np.random.default_rng(shape[0])
    pics = [rng.integers(0, 1 << 24, size=shape, dtype=np.uint32) for _ in range(2)]
    c = H.preset("i", H.FLAG_FILTER | H.FLAG_NOAUDIO)
    with H.Engine(c, 16000000, max_frames=2) as a, H.Engine(c, 16000000, max_frames=2) as b:
        host = [b.host_picture(*shape) for _ in range(2)]
        for i in range(2):
            a.frame_upload(i, pics[i])
            host[i][:] = pics[i]
            b.frame_upload_pinned(i, host[i])
        a.render(2, slots=[0, 1])
        b.render(2, slots=[0, 1])
        assert np.array_equal(a.fetch(0, 2 * 640000), b.fetch(0, 2 * 640000))


@pytest.mark.parametrize("mode,first", [("i", 3350), ("l", 1000)])
def test_two_minutes_in_the_stream_still_equals_the_reference(mode, first):
    """Frames 3350 .. 3361 of `-m i -s 16000000 --filter test` -- where the sample index passes 2^31, after 65 000
    re-normalisations of the sound phasor and 134 000 NICAM frames -- from the drop-in binary and from the unmodified
    reference CLI: the same bytes. And frames 1000 .. 1011 of SECAM-L: 578 000 lines into the colour sub-carrier's
    chain, batch after batch on the device."""
    ref, hvk = os.path.join(REF, "hacktv_ref"), os.path.join(REF, "hacktv_hvk")
    require_ref(ref)
    require_ref(hvk)
    flags = ["-m", mode, "-s", "16000000", "--filter", "-o", "-", "test"]
    skip, take = first * 2560000, 12 * 2560000

    def digest(exe, env=None):
        p = subprocess.Popen([exe] + flags, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        try:
            left = skip
            while left > 0:
                n = len(p.stdout.read(min(left, 1 << 24)))
                assert n, "ended early"
                left -= n
            h, left = hashlib.sha256(), take
            while left > 0:
                chunk = p.stdout.read(min(left, 1 << 24))
                assert chunk, "ended early"
                h.update(chunk)
                left -= len(chunk)
            return h.hexdigest()
        finally:
            p.kill()
            p.wait()

    got = digest(hvk, dict(os.environ, HVK_BATCH="32"))
    assert got == digest(ref)
