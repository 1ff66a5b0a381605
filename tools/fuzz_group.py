#!/usr/bin/env python3
"""tools/fuzz_group.py [cases] [seed] [seconds] -- one stream over SEVERAL engines (hvk_group_*: block b on engine b mod N, the
serial chains' states handed from engine to engine) against ONE engine's stream, every sample: random configurations (any mode, a
rate it takes, --filter / --noaudio / --nonicam, A2 stereo, the VBI inserters, sound-in-syncs, SECAM with field identification,
S-Video, --pixelrate, --offset / --swap-iq), 2 .. 4 engines on the one device, blocks of 1 .. 3 frames, a new random picture on
every frame, loud sound. Configurations a group refuses (chains over every sample: FM video, --passthru ...) are counted. The one
engine is what tools/fuzz_parity.py holds against the oracle. Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 240
rng = np.random.default_rng(SEED)
MODES = ["i", "b", "g", "pal-d", "pal-k", "pal-fm", "pal", "pal-m", "pal-n", "525pal", "m", "ntsc-i", "ntsc-fm", "ntsc", "pal60-i", "pal60", "l", "d", "k", "secam-i", "secam-b",
         "secam-g", "secam-fm", "secam", "e", "819", "a", "ntsc-a", "405-i", "405", "ntsc-405", "240-am", "240", "30-am", "30", "nbtv-am", "nbtv",
         "apollo-fsc-fm", "apollo-fsc", "apollo-fm", "apollo", "m-cbs405", "cbs405"]
RATES = {625: [16000000, 13500000, 14000000, 18000000, 20250000, 17734475, 27000000], 525: [13500000, 16000000, 14318181, 18000000, 27000000], 819: [24570000, 16380000],
         405: [8100000, 16200000, 12150000], 240: [4800000], 30: [750000], 32: [800000], 320: [3200000, 8000000, 13500000]}
done = refused = bad = 0
t_start = time.time()
while done < N and time.time() - t_start < LIMIT:
    mode = MODES[int(rng.integers(len(MODES)))]
    base = H.preset(mode, 0)
    lines = int(base.lines)
    rates = [17496000] if mode in ("m-cbs405", "cbs405") else RATES.get(lines, [16000000])
    sr = int(rates[int(rng.integers(len(rates)))])
    flags = 0
    for f, p in ((H.FLAG_FILTER, 0.5), (H.FLAG_NOAUDIO, 0.35), (H.FLAG_NONICAM, 0.2)):
        if rng.random() < p: flags |= f
    conf = H.preset(mode, flags)
    opts = []
    def maybe(name, value, p):
        if rng.random() < p:
            setattr(conf, name, value); opts.append("%s=%s" % (name, value)); return True
        return False
    if lines in (625, 525):
        maybe("vits", 1, 0.2); maybe("vitc", 1, 0.2); maybe("acp", 1, 0.15)
        if lines == 625:
            maybe("wss", int(rng.integers(1, 9)), 0.2); maybe("sis", 1, 0.15)
        if mode in ("g", "b", "m") and not (flags & H.FLAG_NOAUDIO): maybe("a2stereo", 1, 0.3)
    if mode in ("l", "d", "k", "secam", "secam-fm", "secam-i", "secam-b", "secam-g"):
        if maybe("secam_field_id", 1, 0.5) and rng.random() < 0.5:
            conf.secam_field_id_lines = int(rng.integers(1, 10)); opts.append("secam_field_id_lines=%d" % conf.secam_field_id_lines)
    if mode in ("pal", "ntsc", "secam", "pal60", "525pal"): maybe("s_video", 1, 0.25)
    if rng.random() < 0.1: conf.swap_iq = 1; opts.append("swap_iq")
    if rng.random() < 0.1: conf.offset = int(rng.integers(-8, 9)) * 50000 or 250000; opts.append("offset=%d" % conf.offset)
    pr = 0
    if lines in (625, 525) and rng.random() < 0.25:
        cand = [r for r in RATES[lines] if r != sr]
        pr = int(cand[int(rng.integers(len(cand)))])
    ne = int(rng.integers(2, 5)); block = int(rng.integers(1, 4)); nblocks = int(rng.integers(ne, 2 * ne + 2))
    desc = "%-13s %9d px %9d flags %d %s: %d engines, %d blocks of %d" % (mode, sr, pr, flags, " ".join(opts), ne, nblocks, block)
    print("case     ", desc, flush=True)
    n = block * nblocks
    try:
        one = H.Engine(conf, sr, device=0, max_frames=n, pixel_rate=pr)
    except H.HvkError:
        refused += 1; print("refused  ", desc, "(one engine)", flush=True); continue
    try:
        with one:
            w, h = one.info["active_width"], one.info["active_lines"]
            def picture(i):
                kind = int(rng.integers(4))
                if kind == 0: return rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
                if kind == 1: return np.full((h, w), int(rng.integers(0, 1 << 24)), np.uint32)
                if kind == 2:
                    yy, xx = np.mgrid[0:h, 0:w]
                    return np.ascontiguousarray((((xx * 255 // max(w - 1, 1) + i * 9) % 256).astype(np.uint32) << 16) | (((yy * 255 // max(h - 1, 1)) % 256).astype(np.uint32) << 8) | ((xx + yy) % 256).astype(np.uint32))
                ww, hh = int(rng.integers(2, w + 1)), int(rng.integers(1, h + 1))
                return rng.integers(0, 1 << 24, (hh, ww), dtype=np.uint32)
            pics = [picture(i) for i in range(n)]
            audio = rng.integers(-32768, 32768, (65536, 2)).astype(np.int16)
            for i, p in enumerate(pics): one.frame_upload(i, p, 0)
            while one.audio_needed(n) > 0: one.audio_write(audio)
            one.render(n, slots=list(range(n)))
            want = one.fetch(0, one.frame_start(n))
            starts = [one.frame_start(i) for i in range(n + 1)]
        try:
            g = H.Group(conf, sr, [0] * ne, block, pixel_rate=pr)
        except H.HvkError:
            refused += 1; print("refused  ", desc, "(group)", flush=True); continue
        got = []
        with g:
            for b in range(nblocks):
                e = g.block_engine()
                for i in range(block): g.frame_upload(i, pics[b * block + i], 0)
                while g.audio_needed(block) > 0: g.audio_write(audio)
                g.stage(block)
                g.launch()
                got.append(e.fetch(0, starts[(b + 1) * block] - starts[b * block]))
            hosts = sum(e.secam_stats()["host_frames"] for e in g.engines) if mode in ("l", "d", "k", "secam", "secam-fm", "secam-i", "secam-b", "secam-g") else 0
        got = np.concatenate(got)
        if got.shape != want.shape or not np.array_equal(got, want):
            bad += 1
            d = np.nonzero((got != want).any(axis=1))[0] if got.shape == want.shape else np.array([-1])
            print("DIFFERENT", desc, "first at sample %d (frame %d), %d samples; shapes %s %s" % (d[0], int(np.searchsorted(starts, d[0], side="right")) - 1, d.size, got.shape, want.shape), flush=True)
        else:
            print("equal    ", desc, ("host frames %d" % hosts) if hosts else "", flush=True)
        done += 1
    except Exception as ex:
        bad += 1
        print("ERROR    ", desc, repr(ex)[:200], flush=True)
        done += 1
print("%d compared, %d refused, %d bad, %.0f s" % (done, refused, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
