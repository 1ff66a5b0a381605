#!/usr/bin/env python3
"""tools/hour.py -- BASELINE config 5 on ONE device: `-m i -s 16000000 --filter test` for one hour of signal (90 000
frames, 57.6 G samples, 230.4 GB of int16 I/Q) rendered in blocks of 128 frames, every block checked against the
unmodified reference CLI's output as reduced by oracle/make_golden_hour.py (tests/golden/ref_hour.json):

  * every block: the two 64-bit sums over its samples, computed on the device (hvk_block_sums) -- every sample of the hour
    is compared, 16 bytes per block cross PCIe;
  * the blocks around frames 0, 9 000, 45 000 and 90 000: fetched and hashed (sha256);
  * full_sha: the whole stream read back block by block and hashed on the way out, the cumulative sha256 after 9 000,
    45 000 and 90 000 frames against the reference's (the hashing thread is what bounds that run: one core at about 1-2 GB/s).

What the hour exercises that shorter runs do not: 1.76 M re-normalisations of the FM phasor (src/video.c:2266-2275), a frame
counter past 2^16 and sample positions past 2^35 (src/video.c:4927-4931), the test source's audio loop wrapping 14 000 times
(src/av_test.c:46-52, :156-196).

bench.py's `5_one_hour` section and tests/test_gpu_hour.py call run(); stand-alone: python tools/hour.py [--sound] [--full-sha]."""
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "ref_hour.json")


def golden(sound):
    if not os.path.exists(GOLD):
        return None
    return json.load(open(GOLD)).get("i_hour" if sound else "i_hour_noaudio")


def run(H, frame, audio, device=0, sound=False, full_sha=False, max_blocks=None, log=None):
    """Renders the hour (or its first max_blocks blocks). Raises AssertionError at the first block that differs from the
    reference. Returns a dict for the bench JSON."""
    import numpy as np
    ref = golden(sound)
    if ref is None:
        return {"skipped": "tests/golden/ref_hour.json has no %s case (oracle/make_golden_hour.py)" % ("i_hour" if sound else "i_hour_noaudio")}
    F = ref["block_frames"]
    frames = ref["frames"]
    nblocks = (frames + F - 1) // F
    if max_blocks is not None:
        nblocks = min(nblocks, max_blocks)
    flags = H.FLAG_FILTER | (0 if sound else H.FLAG_NOAUDIO)
    marks = {int(k): v for k, v in ref["sha256_at_frames"].items()}
    sha_blocks = {int(k): v for k, v in ref["sha256_of_block"].items()}
    checked_sha, checked_marks = [], []
    cum = hashlib.sha256()
    q = []
    qlock = threading.Condition()
    fail = []

    def hasher():
        # the stream on its way out: blocks in order, the cumulative digest at the marks
        done = 0
        while True:
            with qlock:
                while not q:
                    qlock.wait()
                item = q.pop(0)
                qlock.notify_all()
            if item is None:
                return
            b, nb, buf = item
            view = memoryview(buf.reshape(-1).view(np.uint8))[:nb * fs * 4]
            pos = 0
            for m in sorted(marks):
                if done < m <= done + nb:
                    cum.update(view[pos:(m - done) * fs * 4])
                    pos = (m - done) * fs * 4
                    if cum.copy().hexdigest() != marks[m]:
                        fail.append("the stream's first %d frames hash differently from the reference's" % m)
                    checked_marks.append(m)
            cum.update(view[pos:])
            done += nb
            with qlock:
                free.append(buf)
                qlock.notify_all()

    with H.Engine(H.preset("i", flags), 16000000, device=device, max_frames=F) as e:
        fs = e.info["frame_samples"]
        e.frame_upload(0, frame)
        free = []
        th = None
        if full_sha:
            free = [e.host_buffer(F * fs) for _ in range(3)]
            th = threading.Thread(target=hasher, daemon=True)
            th.start()
        t0 = time.perf_counter()
        t_stage = t_check = 0.0
        for b in range(nblocks):
            nb = min(F, frames - b * F)
            ta = time.perf_counter()
            if sound:
                while e.audio_needed((b * F) + nb) > 0:
                    e.audio_write(audio)
            e.stage(b * F, 1, nb)
            e.launch()
            tb = time.perf_counter()
            s1, s2 = e.block_sums(0, nb * fs)
            want = ref["sums"][b]
            assert ("%016x" % s1, "%016x" % s2) == (want[0], want[1]), "block %d (frames %d..%d) differs from the reference: sums %016x %016x, wanted %s %s" % (b, b * F, b * F + nb - 1, s1, s2, want[0], want[1])
            if b in sha_blocks and not full_sha:
                got = hashlib.sha256(e.fetch(0, nb * fs).tobytes()).hexdigest()
                assert got == sha_blocks[b], "block %d: sha256 differs from the reference's" % b
                checked_sha.append(b)
            if full_sha:
                with qlock:
                    while not free and not fail:
                        qlock.wait(1.0)
                    buf = free.pop(0)
                e.fetch_wait(e.fetch_async(buf, 0, nb * fs))
                if b in sha_blocks:
                    assert hashlib.sha256(memoryview(buf.reshape(-1).view(np.uint8))[:nb * fs * 4]).hexdigest() == sha_blocks[b], "block %d: sha256 differs from the reference's" % b
                    checked_sha.append(b)
                with qlock:
                    q.append((b, nb, buf))
                    qlock.notify_all()
            tc = time.perf_counter()
            t_stage += tb - ta
            t_check += tc - tb
            if fail:
                break
            if log and b % 100 == 0:
                log("hour (%s): block %d / %d, %.1f s" % ("sound" if sound else "noaudio", b, nblocks, tc - t0))
        e.sync()
        t_render = time.perf_counter() - t0
        if th:
            with qlock:
                q.append(None)
                qlock.notify_all()
            th.join()
        wall = time.perf_counter() - t0
        assert not fail, fail[0]
        done_frames = min(frames, nblocks * F)
        return {
            "workload": "-m i -s 16000000 --filter%s test, %d frames (%.1f s of signal, %.1f GB of int16 I/Q) on one device in blocks of %d" % ("" if sound else " --noaudio", done_frames, done_frames / 25.0, done_frames * fs * 4 / 1e9, F),
            "frames": done_frames, "blocks": nblocks,
            "wall_s": round(wall, 2),
            "Msamples_per_s": round(done_frames * fs / wall / 1e6, 1),
            "stage_and_launch_s": round(t_stage, 2), "checks_s": round(t_check, 2),
            "gate": "every block's two 64-bit sums over its samples (computed on the device) == the reference CLI's (tests/golden/ref_hour.json); sha256 of blocks %s == the reference's%s" % (
                checked_sha, ("; cumulative sha256 of the whole stream after %s frames == the reference's" % sorted(checked_marks)) if checked_marks else ""),
            "note": "with sound the serial FM phasor chain on one host core bounds the run; the full-sha run is bounded by one host core hashing 230 GB" if (sound or full_sha) else
                    "--noaudio: nothing serial on the host; every block is rendered, summed on the device and compared before the next one is staged",
        }


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hacktv_amd as H
    import util
    g = util.Golden()
    res = run(H, g.frame("i_full"), g.audio, sound="--sound" in sys.argv, full_sha="--full-sha" in sys.argv,
              max_blocks=int(os.environ["HOUR_BLOCKS"]) if os.environ.get("HOUR_BLOCKS") else None, log=lambda m: print(m, file=sys.stderr, flush=True))
    print(json.dumps(res))
