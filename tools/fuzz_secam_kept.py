#!/usr/bin/env python3
"""tools/fuzz_secam_kept.py [cases] [seed] [seconds] -- SECAM pictures that STAY: random SECAM configurations (mode, rate, --filter,
field identification, the VBI inserters, teletext, sound or none), a handful of picture slots shown in runs of random length over
20 .. 40 frames in batches of 1 .. 8, a slot's picture now and then replaced between batches -- the kept sub-carrier sets of round 6
(made, taken, dropped, met from another state) against the oracle, every sample; the engine's counters say how many frames took a
set. tools/fuzz_parity.py shows a new picture on every frame and never gets there. Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 240
rng = np.random.default_rng(SEED)
MODES = ["l", "d", "k", "secam-i", "secam-b", "secam-g", "secam"]
RATES = [16000000, 18000000, 20250000, 17734475, 27000000]
done = refused = bad = taken = restarts = 0
t_start = time.time()
while done < N and time.time() - t_start < LIMIT:
    mode = MODES[int(rng.integers(len(MODES)))]
    sr = int(RATES[int(rng.integers(len(RATES)))])
    flags = 0
    for f, p in ((H.FLAG_FILTER, 0.6), (H.FLAG_NOAUDIO, 0.6)):
        if rng.random() < p: flags |= f
    conf = H.preset(mode, flags)
    opts = []
    def maybe(name, value, p):
        if rng.random() < p:
            setattr(conf, name, value); opts.append("%s=%s" % (name, value)); return True
        return False
    maybe("vits", 1, 0.2); maybe("vitc", 1, 0.2); maybe("acp", 1, 0.15); maybe("wss", int(rng.integers(1, 9)), 0.2)
    if maybe("secam_field_id", 1, 0.5) and rng.random() < 0.5:
        conf.secam_field_id_lines = int(rng.integers(1, 10)); opts.append("secam_field_id_lines=%d" % conf.secam_field_id_lines)
    tt = maybe("teletext", 1, 0.3)
    if mode == "secam": maybe("s_video", 1, 0.2)
    nfr = int(rng.integers(20, 41))
    nslots = int(rng.integers(1, 5))
    maxb = int(rng.integers(2, 9))
    levels = int(rng.integers(1, 3))
    desc = "%-8s %9d flags %d %s levels %d, %d frames, %d slots, batches of up to %d" % (mode, sr, flags, " ".join(opts), levels, nfr, nslots, maxb)
    print("case     ", desc, flush=True)
    try:
        e = H.Engine(conf, sr, device=0, max_frames=maxb)
    except H.HvkError:
        refused += 1
        print("refused  ", desc, flush=True)
        continue
    try:
        with e:
            w, h = e.info["active_width"], e.info["active_lines"]
            L = e.info["lines"]
            def picture():
                kind = int(rng.integers(4))
                if kind == 0: return rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
                if kind == 1: return np.full((h, w), int(rng.integers(0, 1 << 24)), np.uint32)
                if kind == 2:
                    yy, xx = np.mgrid[0:h, 0:w]
                    return np.ascontiguousarray((((xx * 255 // max(w - 1, 1) + int(rng.integers(256))) % 256).astype(np.uint32) << 16) | (((yy * 255 // max(h - 1, 1)) % 256).astype(np.uint32) << 8) | ((xx + yy) % 256).astype(np.uint32))
                ww, hh = int(rng.integers(2, w + 1)), int(rng.integers(1, h + 1))
                return rng.integers(0, 1 << 24, (hh, ww), dtype=np.uint32)
            pics = [picture() for _ in range(nslots)]
            # the batches, the slot every frame shows (runs of random length), the slots that get another picture in front of a batch
            plan, left, cur = [], nfr, 0
            while left > 0:
                n = int(min(left, rng.integers(1, maxb + 1)))
                slots = []
                for _ in range(n):
                    if rng.random() < 0.12: cur = int(rng.integers(nslots))
                    slots.append(cur)
                new = [(int(s), picture()) for s in range(nslots) if rng.random() < 0.04]
                plan.append((n, slots, new))
                left -= n
            audio = rng.integers(-32768, 32768, (65536, 2)).astype(np.int16)
            ttp = [(rng.integers(0, 256, (32, 45), dtype=np.uint8), int(rng.integers(0, 1 << 32))) for _ in range(nfr)] if tt else None
            shown = [p.copy() for p in pics]
            want, f = [], 0
            with oracle.Oracle(conf, sr, 0) as o:
                o.set_audio(audio, True)
                for n, slots, new in plan:
                    for s, p in new: shown[s] = p
                    for i in range(n):
                        o.set_frame(shown[slots[i]], 0)
                        if ttp is not None: o.teletext_packets(f, ttp[f][0], ttp[f][1])
                        want.append(o.render_lines(L))
                        f += 1
            want = np.concatenate(want)
            e.set_levels(levels)
            for s, p in enumerate(pics): e.frame_upload(s, p, 0)
            got, f = [], 0
            for n, slots, new in plan:
                for s, p in new: e.frame_upload(s, p, 0)
                if ttp is not None:
                    for i in range(n): e.teletext_packets(i, ttp[f + i][0], ttp[f + i][1])
                while e.audio_needed(n) > 0: e.audio_write(audio)
                e.render(n, slots=slots)
                got.append(e.fetch(0, e.frame_start(f + n) - e.frame_start(f)))
                f += n
            got = np.concatenate(got)
            kept = e.secam_kept(); st = e.secam_stats()
        taken += kept["frames_taken"]; restarts += kept["restarts"]
        if got.shape != want.shape or not np.array_equal(got, want):
            bad += 1
            d = np.nonzero((got != want).any(axis=1))[0] if got.shape == want.shape else np.array([-1])
            print("DIFFERENT", desc, "first at sample %d, %d samples; kept %s stats %s; plan %s" % (d[0], d.size, kept, st, [(n, s, [x[0] for x in nw]) for n, s, nw in plan]), flush=True)
        else:
            print("equal    ", desc, "kept", kept, "host frames", st["host_frames"], flush=True)
        done += 1
    except Exception as ex:
        bad += 1
        print("ERROR    ", desc, repr(ex)[:200], flush=True)
        done += 1
print("%d compared, %d refused, %d bad, %d frames took a kept set, %d restarts, %.0f s" % (done, refused, bad, taken, restarts, time.time() - t_start))
sys.exit(1 if bad else 0)
