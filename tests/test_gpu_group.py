"""One stream rendered by several engines (hvk_group_*, hvk_group.cpp; the shim's HVK_DEVICES): blocks of frames dealt
round-robin to N engines -- here N engines on the one GPU of the box --, the serial sound chains handed from engine to
engine in process, the stream reassembled (i) on the host, every engine's read-back into its block's place, and (ii) on
one device (hvk_group_gather). Every sample against the unmodified reference: committed digests
(tests/golden/ref_long.json, ref_digests.json) and, where the binary travels with the snapshot, its output made in the job."""
import ctypes
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


def _feed(g, golden, n):
    while g.audio_needed(n) > 0:
        g.audio_write(golden.audio)


def test_three_engines_with_sound_host_direct(golden):
    """-m i --filter, FM + NICAM on: 128 frames in blocks of 8 (two short ones) over three engines -- five rounds and a
    bit --, each block fetched from the engine that rendered it: cumulative digests at 37 and 128 frames."""
    conf, sr = golden.conf("i_full")
    want = LONG["i_full"]["sha256_at_frames"]
    blocks = [8, 8, 8, 8, 5] + [8] * 11 + [3]
    h = hashlib.sha256()
    done = 0
    with H.Group(conf, sr, [0, 0, 0], 8) as g:
        fs = g.info["frame_samples"]
        used = set()
        for b in blocks:
            e = g.block_engine()
            for i in range(b):
                g.frame_upload(i, golden.frame("i_full"))
            _feed(g, golden, b)
            g.stage(b)
            used.add(g.launch())
            h.update(e.fetch(0, b * fs).tobytes())
            done += b
            if str(done) in want:
                assert h.copy().hexdigest() == want[str(done)], "first %d frames differ from the reference" % done
        assert done == 128 and used == {0, 1, 2}
        # every engine ran the chains over its own frames only
        gen = [e.sound_samples_generated() for e in g.engines]
        assert sum(gen) <= 129 * fs and max(gen) < 60 * fs, gen


def test_sound_written_far_ahead_of_the_blocks(golden):
    """The caller writes the 32 kHz source a long way ahead (one 204 800-pair chunk = 160 frames at a time) and the
    stream runs PAST it: 165 frames in blocks of 5 over three engines. An engine that comes round again still holds what
    it was dealt behind its last block (hvk_sound_state_import keeps a queue that holds the position) -- the group must go
    on from the END of what the engine holds, not deal those pairs a second time (they would be played, time-shifted,
    once the first chunk is used up: frames 160 on). Cumulative digests at 37, 128 and 165 frames; and no engine's
    queue grows beyond what the stream has written."""
    conf, sr = golden.conf("i_full")
    want = LONG["i_full"]["sha256_at_frames"]
    h = hashlib.sha256()
    done = 0
    written = 0
    with H.Group(conf, sr, [0, 0, 0], 5) as g:
        fs = g.info["frame_samples"]
        blocks = [5] * 7 + [2] + [5] * 18 + [1] + [5] * 7 + [2]
        assert sum(blocks) == 165
        for b in blocks:
            e = g.block_engine()
            for i in range(b):
                g.frame_upload(i, golden.frame("i_full"))
            while g.audio_needed(b) > 0:
                g.audio_write(golden.audio)
                written += len(golden.audio)
            g.stage(b)
            g.launch()
            h.update(e.fetch(0, b * fs).tobytes())
            done += b
            assert H.lib().hvk_sound_source_end(e.h) <= written, "an engine holds source pairs the caller never wrote: dealt twice"
            if str(done) in want:
                assert h.copy().hexdigest() == want[str(done)], "first %d frames differ from the reference" % done
        assert done == 165 and written >= 2 * len(golden.audio)


def _gather_rounds(golden, case, devices, expect, monkeypatch=None):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipSetDevice.argtypes = [ctypes.c_int]
    conf, sr = golden.conf(case)
    want = golden.cases[case]["sha256_cumulative"]
    n = len(devices)
    with H.Group(conf, sr, devices, 1) as g:
        fs = g.info["frame_samples"]
        root = ctypes.c_void_p()
        hip.hipSetDevice(devices[0])
        assert hip.hipMalloc(ctypes.byref(root), n * fs * 4) == 0
        host = np.zeros((n * fs, 2), np.int16)
        h = hashlib.sha256()
        for rnd in range(len(want) // n):           # (the committed cumulative digests hold four frames)
            for k in range(n):
                g.frame_upload(0, golden.frame(case))
                _feed(g, golden, 1)
                g.stage(1)
                g.launch()
            g.gather(0, root, fs)
            assert expect in g.gather_backend(), g.gather_backend()
            g.engines[0].sync()
            hip.hipSetDevice(devices[0])
            assert hip.hipMemcpy(host.ctypes.data, root, n * fs * 4, 2) == 0
            h.update(host.tobytes())
            assert h.copy().hexdigest() == want[n * (rnd + 1) - 1], (case, rnd, g.gather_backend())
        hip.hipFree(root)


def test_gather_by_peer_copies(golden, monkeypatch):
    """HVK_GATHER=peer: hipMemcpyPeerAsync, every sender pushing its block on a stream of its own and the root's stream
    waiting for each -- the reassembly for a machine without a usable librccl, here between engines on one device."""
    monkeypatch.setenv("HVK_GATHER", "peer")
    for case in ("i_vsb", "i_full"):
        _gather_rounds(golden, case, [0, 0], "peer")


def _device_count():
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0


def test_gather_between_distinct_devices(golden, monkeypatch):
    """More than one HIP device (an 8-GPU node; one MI355X in CPX / DPX partition mode): the RCCL branch -- one
    communicator per device, grouped ncclSend / ncclRecv -- and the peer-copy branch, each against the reference's
    digests, with sound (the chains handed on between engines on different devices) and without."""
    nd = _device_count()
    if nd < 2:
        pytest.skip("one HIP device: hvk_group_gather's RCCL branch needs two (the pool's boxes offer one and refuse partitioning: profiles/r05_partition_probe.txt)")
    for devs in ([0, 1], list(range(min(nd, 4)))):
        monkeypatch.delenv("HVK_GATHER", raising=False)
        for case in ("i_vsb", "i_full"):
            _gather_rounds(golden, case, devs, "rccl")
        monkeypatch.setenv("HVK_GATHER", "peer")
        for case in ("i_vsb", "i_full"):
            _gather_rounds(golden, case, devs, "peer")


def test_two_engines_gathered_on_one_device(golden):
    """--noaudio and with sound: rounds of two 1-frame blocks, gathered into the root engine's device memory
    (hvk_group_gather: engines that share a device -> device-to-device copies; distinct devices -> RCCL), read back from
    there: the reference's first four frames."""
    hip = ctypes.CDLL("libamdhip64.so")      # (the runtime libhvk is linked with: already in the process)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    for case in ("i_vsb", "i_full"):
        conf, sr = golden.conf(case)
        want = golden.cases[case]["sha256_cumulative"]
        with H.Group(conf, sr, [0, 0], 1) as g:
            fs = g.info["frame_samples"]
            root = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(root), 2 * fs * 4) == 0
            host = np.zeros((2 * fs, 2), np.int16)
            h = hashlib.sha256()
            assert "hipMemcpy" in g.gather_backend()
            for rnd in range(2):
                for k in range(2):
                    g.frame_upload(0, golden.frame(case))
                    _feed(g, golden, 1)
                    g.stage(1)
                    g.launch()
                g.gather(0, root, fs)
                g.engines[0].sync()          # the gathered round is complete where the root engine's stream is
                assert hip.hipMemcpy(host.ctypes.data, root, 2 * fs * 4, 2) == 0      # hipMemcpyDeviceToHost
                h.update(host.tobytes())
                assert h.copy().hexdigest() == want[2 * rnd + 1], (case, rnd)
            hip.hipFree(root)


def test_525_lines_the_picture_before_a_block(golden):
    """NTSC-M: the last line of a frame shows picture and lies within the video filter's reach of the next frame's first
    samples -- the engine of a block needs the picture of the frame before it. Pictures that change on every frame, blocks
    of two frames over two engines, against one engine that renders the stream alone; and the test card against the
    reference's digests."""
    conf, sr = golden.conf("m_full")
    rng = np.random.default_rng(5)
    base = golden.frame("m_full")
    pics = [np.where(rng.random(base.shape) < 0.3, rng.integers(0, 1 << 24, base.shape, dtype=np.uint32), np.roll(base, 7 * i, axis=1)).astype(np.uint32) for i in range(6)]
    with H.Engine(conf, sr, device=0, max_frames=6) as e:
        for i, p in enumerate(pics):
            e.frame_upload(i, p)
        while e.audio_needed(6) > 0:
            e.audio_write(golden.audio)
        e.render(6, slots=list(range(6)))
        fs = e.info["frame_samples"]
        want = e.fetch(0, 6 * fs)
    got = []
    with H.Group(conf, sr, [0, 0], 2) as g:
        for b in range(3):
            e = g.block_engine()
            for i in range(2):
                g.frame_upload(i, pics[2 * b + i])
            _feed(g, golden, 2)
            g.stage(2)
            g.launch()
            got.append(e.fetch(0, 2 * fs))
    got = np.concatenate(got)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "first difference at sample %d (frame %d)" % (bad[0], bad[0] // fs)

    ref = golden.cases["m_full"]["sha256_cumulative"]
    h = hashlib.sha256()
    with H.Group(conf, sr, [0, 0, 0], 1) as g:
        for f in range(4):
            e = g.block_engine()
            g.frame_upload(0, base)
            _feed(g, golden, 1)
            g.stage(1)
            g.launch()
            h.update(e.fetch(0, fs).tobytes())
            assert h.copy().hexdigest() == ref[f], f


def test_chains_over_every_sample_are_refused(golden):
    for case in ("pal_fm",):
        conf, sr = golden.conf(case)
        with pytest.raises(H.HvkError) as ei:
            H.Group(conf, sr, [0, 0], 2)
        assert ei.value.code == H.HVK_UNSUPPORTED
        with H.Group(conf, sr, [0], 2) as g:      # one engine: an ordinary stream
            assert g.n == 1


@pytest.mark.parametrize("case,block,nblocks", [("l_full", 1, 4), ("l_full", 3, 7), ("d_full", 2, 6), ("secam_bb", 2, 5), ("l_fid", 2, 4)])
def test_secam_colour_over_three_engines_equals_the_reference(golden, case, block, nblocks):
    """SECAM over a group: the colour chain's state -- the IIR pair the reference never resets and the values behind a frame's last
    line (src/video.c:3095-3099, :3160-3165, :3202-3229) -- handed from the engine of block b to the engine of block b + 1
    (hvk_secam_state_export / _import) like the sound chains'. Blocks of 1, 2 and 3 frames dealt to three engines on the one
    GPU, with sound (-m l --filter) and without, with the video filter and in the baseband: every frame's samples equal the
    reference's digests where the golden file has them and ONE engine's stream throughout; no frame went through the host's
    chain."""
    if case not in golden.cases:
        pytest.skip("no golden case " + case)
    conf, sr = golden.conf(case)
    c = golden.cases[case]
    fs = c.get("frame_samples", c["width"] * c["lines"])
    real = bool(c["real"])
    n = block * nblocks
    with H.Engine(conf, sr, device=0, max_frames=n) as one:
        one.frame_upload(0, golden.frame(case))
        while one.audio_needed(n) > 0:
            one.audio_write(golden.audio)
        one.render(n)
        want = one.fetch(0, n * fs)
    got = []
    with H.Group(conf, sr, [0, 0, 0], block) as g:
        for e in g.engines:
            e.frame_upload(0, golden.frame(case))
        for b in range(nblocks):
            e = g.block_engine()
            while g.audio_needed(block) > 0:
                g.audio_write(golden.audio)
            g.stage(block, slots=[0] * block)
            g.launch()
            got.append(e.fetch(0, block * fs))
        hosts = sum(e.secam_stats()["host_frames"] for e in g.engines)
    got = np.concatenate(got)
    assert np.array_equal(got, want)
    assert hosts == 0
    cum = c["sha256_cumulative"]
    for m in range(1, min(n, len(cum)) + 1):
        assert util.sha256(util.stream_bytes(got[: m * fs], real)) == cum[m - 1], "first %d frames differ from the reference's digest" % m


@pytest.mark.parametrize("flags,key", [(["-m", "i", "-s", "16000000", "--filter"], "i_full"),
                                       (["-m", "i", "-s", "16000000", "--filter", "--noaudio"], None),
                                       (["-m", "m", "-s", "13500000", "--filter"], None)])
def test_dropin_binary_on_three_engines(flags, key):
    """hacktv_hvk with HVK_DEVICES=0,0,0 (batches of 4 frames: 37 frames are three rounds and a bit) writes what the
    reference CLI writes -- with sound (the committed digest at 37 frames, and the reference run in the job), without, and
    on 525 lines."""
    hvk = os.path.join(REF, "hacktv_hvk")
    ref = os.path.join(REF, "hacktv_ref")
    require_ref(hvk)

    def run(binary, nbytes, env):
        p = subprocess.Popen([binary] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        h = hashlib.sha256()
        left = nbytes
        while left > 0:
            chunk = p.stdout.read(min(left, 1 << 22))
            if not chunk:
                break
            h.update(chunk)
            left -= len(chunk)
        p.kill()
        p.wait()
        assert left == 0, "%s ended early" % binary
        return h.hexdigest()

    fb = 2560000 if "i" in flags[1] else 450450 * 4
    got = run(hvk, 37 * fb, dict(os.environ, HVK_BATCH="4", HVK_DEVICES="0,0,0"))
    if key:
        assert got == LONG[key]["sha256_at_frames"]["37"]
    if os.path.exists(ref):
        assert got == run(ref, 37 * fb, dict(os.environ))
    else:
        assert got == run(hvk, 37 * fb, dict(os.environ, HVK_BATCH="4"))
