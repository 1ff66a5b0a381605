#!/usr/bin/env python3
"""tools/fuzz_parity.py [cases] [seed] [seconds] -- random configurations (any of the 44 mode ids, a rate the mode takes, a random
set of the options that need no side input: --filter, --noaudio, --nonicam, A2 stereo, --pixelrate, S-Video, sound-in-syncs,
VITS / VITC / WSS (a mode, or auto with one of six pixel aspects) / ACP / CC608, field identification (1 .. 9 lines), --nocolour, --interlace, --offset, --swap-iq, --gamma / --level / --invert-video /
--volume, teletext packets, --passthru, --raw-bb-file, levels computed or looked up), random pictures that change every frame
(noise, flat, gradients, none, other sizes, either field-order flag), loud sound: the engine on the GPU against the oracle (which tests/ pin to the reference), three frames
in a random batch split, every sample. FUZZ_ONLY=text compares only the cases whose description holds the text. Configurations the engine refuses are counted, not failed. Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 240
EXTRA = os.environ.get("FUZZ_PLAIN", "") == ""     # pictures' interlace flags, caption bytes, other batch splits (draws more random numbers)
SIDE = os.environ.get("FUZZ_NO_SIDE", "") == ""    # --gamma / --level / --invert-video / --volume, teletext packets, --passthru, --raw-bb-file
ONLY = os.environ.get("FUZZ_ONLY", "")      # compare only the cases whose description holds this
MORE = os.environ.get("FUZZ_NO_MORE", "") == ""    # --nocolour, --wss auto with one of six pixel aspects, --secam-field-id-lines 1 .. 9 (drawn since the end of round 5: FUZZ_NO_MORE=1 redraws the earlier seeds' cases)
PARS = [(1, 1), (12, 13), (16, 11), (64, 45), (4, 3), (16, 15)]
FRAMES = int(os.environ.get("FUZZ_FRAMES", "3"))   # frames a case (3: the batch splits of old; more: random batches of 1 .. 6 frames -- the serial chains over a longer run)
rng = np.random.default_rng(SEED)
MODES = ["i", "b", "g", "pal-d", "pal-k", "pal-fm", "pal", "pal-m", "pal-n", "525pal", "m", "ntsc-i", "ntsc-fm", "ntsc", "pal60-i", "pal60", "l", "d", "k", "secam-i", "secam-b",
         "secam-g", "secam-fm", "secam", "e", "819", "a", "ntsc-a", "405-i", "405", "ntsc-405", "240-am", "240", "30-am", "30", "nbtv-am", "nbtv",
         "apollo-fsc-fm", "apollo-fsc", "apollo-fm", "apollo", "m-cbs405", "cbs405"]
RATES = {625: [16000000, 13500000, 14000000, 18000000, 20250000, 17734475, 27000000], 525: [13500000, 16000000, 14318181, 18000000, 27000000], 819: [24570000, 16380000],
         405: [8100000, 16200000, 12150000], 240: [4800000], 30: [750000], 32: [800000], 320: [3200000, 8000000, 13500000]}
WIDE_RATES = {625: [8000000, 9000000, 10000000, 12000000, 15000000, 21000000, 24000000, 30000000, 36000000, 35468950, 38000000, 16384000],
              525: [8000000, 9000000, 10000000, 12272727, 15000000, 21000000, 24545454, 28636362, 30000000, 36000000]}
done = refused = bad = 0
bad_lines = []
t_start = time.time()
case = 0
while done < N and time.time() - t_start < LIMIT:
    case += 1
    mode = MODES[int(rng.integers(len(MODES)))]
    try:
        base = H.preset(mode, 0)
    except H.HvkError:
        continue
    lines = int(base.lines)
    if mode in ("m-cbs405", "cbs405"):
        rates = [17496000]
    else:
        rates = RATES.get(lines, [16000000])
    sr = int(rates[int(rng.integers(len(rates)))])
    if os.environ.get("FUZZ_WIDE_RATES") and lines in WIDE_RATES and rng.random() < 0.5:
        sr = int(WIDE_RATES[lines][int(rng.integers(len(WIDE_RATES[lines])))])     # (round 6: chroma low passes of 5 .. 33 taps, lines that are not a whole number of samples)
    flags = 0
    for f, p in ((H.FLAG_FILTER, 0.5), (H.FLAG_NOAUDIO, 0.35), (H.FLAG_NONICAM, 0.2)):
        if rng.random() < p:
            flags |= f
    if MORE and rng.random() < 0.1:
        flags |= H.FLAG_NOCOLOUR
    conf = H.preset(mode, flags)
    opts = []
    par = (12, 13)
    def maybe(name, value, p):
        if rng.random() < p:
            setattr(conf, name, value); opts.append("%s=%s" % (name, value)); return True
        return False
    if lines in (625, 525):
        maybe("vits", 1, 0.2); maybe("vitc", 1, 0.2); maybe("acp", 1, 0.15)
        if lines == 525 and EXTRA: maybe("cc608", 1, 0.25)
        if lines == 625:
            maybe("wss", int(rng.integers(1, 9)), 0.2); maybe("sis", 1, 0.2)
        maybe("interlace", 1, 0.12)
        if mode in ("g", "b", "m") and not (flags & H.FLAG_NOAUDIO):
            maybe("a2stereo", 1, 0.3)
    if mode in ("l", "d", "k", "secam", "secam-fm", "secam-i", "secam-b", "secam-g"):
        if maybe("secam_field_id", 1, 0.5) and MORE and rng.random() < 0.5:
            conf.secam_field_id_lines = int(rng.integers(1, 10)); opts.append("secam_field_id_lines=%d" % conf.secam_field_id_lines)
    if mode in ("pal", "ntsc", "secam", "pal60", "525pal"):
        maybe("s_video", 1, 0.3)
    if base.output_type != 0 if hasattr(base, "output_type") else False:
        pass
    tt = pt = rb = False
    if SIDE:
        if rng.random() < 0.15: conf.gamma = float(rng.uniform(0.4, 2.6)); opts.append("gamma=%.3f" % conf.gamma)
        if rng.random() < 0.15: conf.level = float(conf.level * rng.uniform(0.3, 1.0)); opts.append("level=%.3f" % conf.level)
        maybe("invert_video", 1, 0.1)
        maybe("volume", int(rng.integers(64, 700)), 0.15)
        if lines == 625: tt = maybe("teletext", 1, 0.15)
        pt = maybe("passthru", 1, 0.1)
        if lines in (625, 525) and not tt and rng.random() < 0.08:
            conf.raw_bb = 1; conf.raw_bb_blanking_level = 2000; conf.raw_bb_white_level = 21000; rb = True; opts.append("raw_bb")
    if MORE and lines == 625 and not conf.wss and rng.random() < 0.15:
        conf.wss = 0xFF; opts.append("wss=auto")
    if MORE:
        par = PARS[int(rng.integers(len(PARS)))]
        if conf.wss == 0xFF: opts.append("par=%d:%d" % par)
    if rng.random() < 0.15: conf.swap_iq = 1; opts.append("swap_iq")
    if rng.random() < 0.15: conf.offset = int(rng.integers(-8, 9)) * 50000 or 250000; opts.append("offset=%d" % conf.offset)
    pr = 0
    if lines in (625, 525) and rng.random() < 0.3:
        # (FUZZ_FSC_PIXELS=1: 4 x the sub-carrier as a PIXEL rate as well -- resamplers of hundreds of thousands of phases, taken since round 6)
        cand = [r for r in RATES[lines] if r != sr and (os.environ.get("FUZZ_FSC_PIXELS") or r not in (17734475, 14318181))]
        pr = int(cand[int(rng.integers(len(cand)))])
    levels = int(rng.integers(1, 3))
    desc = "%-13s %9d px %9d flags %d %s levels %d" % (mode, sr, pr, flags, " ".join(opts), levels)
    if os.environ.get("FUZZ_ANNOUNCE"): print("case     ", desc, flush=True)       # (before anything is made: a case that ends the process is the last one named)
    try:
        e = H.Engine(conf, sr, device=0, max_frames=3 if FRAMES == 3 else 6, pixel_rate=pr)
    except H.HvkError as err:
        refused += 1
        print("refused  ", desc, flush=True)
        continue
    try:
        with e:
            w, h = e.info["active_width"], e.info["active_lines"]
            fs = e.info["frame_samples"]
            L = e.info["lines"]
            nfr = FRAMES
            npic = nfr * (2 if conf.interlace else 1)
            pics = []
            for i in range(npic):
                kind = int(rng.integers(5))
                if kind == 0: p = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
                elif kind == 1: p = np.full((h, w), int(rng.integers(0, 1 << 24)), np.uint32)
                elif kind == 2:
                    yy, xx = np.mgrid[0:h, 0:w]
                    p = ((((xx * 255 // max(w - 1, 1) + i * 9) % 256).astype(np.uint32) << 16) | (((yy * 255 // max(h - 1, 1)) % 256).astype(np.uint32) << 8) | ((xx + yy) % 256).astype(np.uint32))
                elif kind == 3: p = None
                else:
                    # a picture of another size: narrower, shorter (centred, src/video.c:4896-4897)
                    ww, hh = int(rng.integers(2, w + 1)), int(rng.integers(1, h + 1))
                    p = rng.integers(0, 1 << 24, (hh, ww), dtype=np.uint32)
                pics.append(None if p is None else np.ascontiguousarray(p))
            # (the pictures' interlace flag: which field the source says comes first, src/video.c:3081-3084)
            ilace = [int(rng.integers(3)) if EXTRA else 0 for _ in range(npic)]
            cc = rng.integers(0, 256, (nfr, 2)) if (EXTRA and conf.cc608) else None
            split = [(2, 1), (1, 2), (3,), (1, 1, 1)][int(rng.integers(4))] if EXTRA else (2, 1)
            if FRAMES != 3:
                split, left = [], nfr
                while left > 0:
                    split.append(int(min(left, rng.integers(1, 7))))
                    left -= split[-1]
                split = tuple(split)
            audio = rng.integers(-32768, 32768, (65536, 2)).astype(np.int16)
            ttp = [(rng.integers(0, 256, (32, 45), dtype=np.uint8), int(rng.integers(0, 1 << 32))) for _ in range(nfr)] if tt else None
            pti = rng.integers(-3000, 3000, (int(fs * 2.4) + 17, 2)).astype(np.int16) if pt else None
            rbs = None
            if rb:
                rbs = rng.integers(300, 24000, (e.info["width"] * L + 311,)).astype(np.int16)
                rbs = np.tile(rbs, (nfr + 2))
            if ONLY and ONLY not in desc:       # (the random draws made: the cases behind it are the same ones)
                done += 1
                continue
            with oracle.Oracle(conf, sr, pr) as o:
                o.set_audio(audio, True)
                o.set_frame_aspect(*par)
                if pti is not None: o.set_passthru(pti)
                if rbs is not None: o.set_rawbb(rbs)
                want = []
                # (the oracle rasters one line ahead with the picture set at that moment: where a frame's first line shows
                # picture the next picture is set before the frame's last line is asked for -- tests/ref_random_check.py)
                early = mode in ("30", "30-am", "nbtv", "nbtv-am")
                for f in range(nfr):
                    if conf.interlace:
                        o.set_frame(pics[2 * f] if pics[2 * f] is not None else np.zeros((0, 0), np.uint32), ilace[2 * f]); o.set_frame2(pics[2 * f + 1] if pics[2 * f + 1] is not None else np.zeros((0, 0), np.uint32), ilace[2 * f + 1])
                    else:
                        o.set_frame(pics[f] if pics[f] is not None else np.zeros((0, 0), np.uint32), ilace[f])
                    if ttp is not None: o.teletext_packets(f, ttp[f][0], ttp[f][1])
                    if cc is not None and (int(cc[f][0]) | int(cc[f][1])) & 0x7F:
                        o.set_cc608(f, int(cc[f][0]), int(cc[f][1]))
                    want.append(o.render_lines((L - 1 if f == 0 else L) if early else L))
                if early:
                    want.append(o.render_lines(1))
                want = np.concatenate(want)
            e.set_levels(levels)
            if pti is not None: e.passthru_write(pti)
            if rbs is not None: e.rawbb_write(rbs)
            got, fdone = [], 0
            for n in split:
                per = 2 if conf.interlace else 1
                for i in range(n * per):
                    e.frame_upload(i, pics[fdone * per + i], ilace[fdone * per + i])
                    e.frame_aspect(i, *par)
                if cc is not None:
                    for i in range(n):
                        e.cc608_write(i, int(cc[fdone + i][0]), int(cc[fdone + i][1]))
                if ttp is not None:
                    for i in range(n):
                        e.teletext_packets(i, ttp[fdone + i][0], ttp[fdone + i][1])
                while e.audio_needed(n) > 0:
                    e.audio_write(audio)
                e.render(n, slots=list(range(n * per)))
                cnt = e.frame_start(fdone + n) - e.frame_start(fdone)
                got.append(e.fetch(0, cnt))
                fdone += n
            got = np.concatenate(got)
        if got.shape != want.shape or not np.array_equal(got, want):
            bad += 1
            if got.shape == want.shape:
                d = np.nonzero((got != want).any(axis=1))[0]
                bad_lines.append("(differed) " + desc)
                print("DIFFERENT", desc, "first at sample %d (line %d), last %d, %d samples; got %s want %s; pictures %s" % (d[0], d[0] // max(e.info["width"], 1), d[-1], d.size, got[d[0]].tolist(), want[d[0]].tolist(),
                      ["none" if p is None else ("flat" if (p == p.flat[0]).all() else "varied") + " %dx%d" % (p.shape[1], p.shape[0]) for p in pics]), "interlace flags", ilace, "batches", split, flush=True)
            else:
                bad_lines.append("(differed in shape) " + desc)
                print("DIFFERENT", desc, "shapes", got.shape, want.shape, flush=True)
        else:
            print("equal    ", desc, flush=True)
        done += 1
    except Exception as ex:
        bad += 1
        bad_lines.append("(error) " + desc + " " + repr(ex)[:120])
        print("ERROR    ", desc, repr(ex)[:200], flush=True)
        done += 1
for ln in bad_lines: print(ln)
print("%d compared, %d refused, %d bad, %.0f s" % (done, refused, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
