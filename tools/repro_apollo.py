#!/usr/bin/env python3
"""tools/repro_apollo.py -- the case tools/fuzz_parity.py 300 9090 found (apollo-fsc at 27 MHz behind the resampler from 13.5 MHz,
raw baseband lines, captions): engine against oracle, with members taken away one at a time; where the samples differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import oracle

def run(mode, sr, pr, flags, members, split=(2, 1), seed=5):
    rng = np.random.default_rng(seed)
    conf = H.preset(mode, flags)
    for k, v in members.items():
        setattr(conf, k, v)
    with H.Engine(conf, sr, device=0, max_frames=3, pixel_rate=pr) as e:
        w, h, fs, L, W = e.info["active_width"], e.info["active_lines"], e.info["frame_samples"], e.info["lines"], e.info["width"]
        pics = [None, rng.integers(0, 1 << 24, (h, w), dtype=np.uint32), None]
        audio = rng.integers(-32768, 32768, (65536, 2)).astype(np.int16)
        cc = rng.integers(0, 256, (3, 2))
        rbs = None
        if conf.raw_bb:
            rbs = np.tile(rng.integers(300, 24000, (W * L + 311,)).astype(np.int16), 5)
        with oracle.Oracle(conf, sr, pr) as o:
            o.set_audio(audio, True)
            o.set_frame_aspect(12, 13)
            if rbs is not None: o.set_rawbb(rbs)
            want = []
            for f in range(3):
                o.set_frame(pics[f] if pics[f] is not None else np.zeros((0, 0), np.uint32), 1)
                if conf.cc608 and (int(cc[f][0]) | int(cc[f][1])) & 0x7F: o.set_cc608(f, int(cc[f][0]), int(cc[f][1]))
                want.append(o.render_lines(L))
            want = np.concatenate(want)
        if rbs is not None: e.rawbb_write(rbs)
        got, fdone = [], 0
        for n in split:
            for i in range(n):
                e.frame_upload(i, pics[fdone + i], 1)
                e.frame_aspect(i, 12, 13)
                if conf.cc608: e.cc608_write(i, int(cc[fdone + i][0]), int(cc[fdone + i][1]))
            while e.audio_needed(n) > 0:
                e.audio_write(audio)
            e.render(n, slots=list(range(n)))
            cnt = e.frame_start(fdone + n) - e.frame_start(fdone)
            got.append(e.fetch(0, cnt))
            fdone += n
        got = np.concatenate(got)
        widths = e.line_widths(0, 3 * L)
    if got.shape != want.shape:
        return "shapes %s %s" % (got.shape, want.shape)
    d = np.nonzero((got != want).any(axis=1))[0]
    if d.size == 0:
        return "equal"
    starts = np.concatenate([[0], np.cumsum(widths)])
    ln = np.searchsorted(starts, d, side="right") - 1
    xs = d - starts[ln]
    diff = (got[d, 0].astype(int) - want[d, 0].astype(int))
    return "%d differ: lines %s..%s (%d distinct), x %d..%d, got - want in %s; first %s" % (d.size, ln.min(), ln.max(), len(set(ln.tolist())), xs.min(), xs.max(), sorted(set(diff.tolist()))[:6],
            [(int(ln[i]), int(xs[i]), int(got[d[i], 0]), int(want[d[i], 0])) for i in range(min(6, d.size))])

base = dict(cc608=1, raw_bb=1, raw_bb_blanking_level=2000, raw_bb_white_level=21000)
cases = [("as found", "apollo-fsc", 27000000, 13500000, 0, base),
         ("no captions", "apollo-fsc", 27000000, 13500000, 0, {k: v for k, v in base.items() if k != "cc608"}),
         ("no raw baseband", "apollo-fsc", 27000000, 13500000, 0, {"cc608": 1}),
         ("neither", "apollo-fsc", 27000000, 13500000, 0, {}),
         ("noaudio", "apollo-fsc", 27000000, 13500000, H.FLAG_NOAUDIO, base),
         ("no resampler", "apollo-fsc", 27000000, 0, 0, base),
         ("18 MHz from 13.5", "apollo-fsc", 18000000, 13500000, 0, base),
         ("ntsc instead", "ntsc", 27000000, 13500000, 0, base),
         ("m instead", "m", 27000000, 13500000, 0, base),
         ("one batch", "apollo-fsc", 27000000, 13500000, 0, base)]
for name, mode, sr, pr, flags, members in cases:
    try:
        print("%-18s %s" % (name, run(mode, sr, pr, flags, members, split=(3,) if name == "one batch" else (2, 1))), flush=True)
    except H.HvkError as ex:
        print("%-18s refused (%d)" % (name, ex.code), flush=True)
