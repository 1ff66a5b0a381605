#!/usr/bin/env python3
"""tools/repro_sv.py -- S-Video behind the resampler (two line widths): engine against oracle on variants of the case the parity
fuzzer's seed 2718 found (-m ntsc -s 27000000 --pixelrate 16000000 --filter --noaudio --vits --cc608 --s-video). Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import oracle

def run(mode, sr, pr, flags, opts, split=(3,), levels=2, pics_kind="noise"):
    conf = H.preset(mode, flags)
    for k, v in opts.items(): setattr(conf, k, v)
    rng = np.random.default_rng(5)
    desc = "%s sr %d px %d flags %d %s split %s levels %d %s" % (mode, sr, pr, flags, opts, split, levels, pics_kind)
    try:
        e = H.Engine(conf, sr, device=0, max_frames=3, pixel_rate=pr)
    except H.HvkError as err:
        print("refused", desc); return
    with e:
        w, h = e.info["active_width"], e.info["active_lines"]
        L = e.info["lines"]
        nfr = 3
        pics = [rng.integers(0, 1 << 24, (h, w), dtype=np.uint32) if pics_kind == "noise" else np.full((h, w), 0x808080, np.uint32) for _ in range(nfr)]
        cc = rng.integers(0, 256, (nfr, 2)) if conf.cc608 else None
        audio = rng.integers(-32768, 32768, (65536, 2)).astype(np.int16)
        with oracle.Oracle(conf, sr, pr) as o:
            o.set_audio(audio, True); o.set_frame_aspect(12, 13)
            want = []
            for f in range(nfr):
                o.set_frame(pics[f], 0)
                if cc is not None and (int(cc[f][0]) | int(cc[f][1])) & 0x7F: o.set_cc608(f, int(cc[f][0]), int(cc[f][1]))
                want.append(o.render_lines(L))
            want = np.concatenate(want)
        e.set_levels(levels)
        got, fdone = [], 0
        for n in split:
            for i in range(n):
                e.frame_upload(i, pics[fdone + i], 0); e.frame_aspect(i, 12, 13)
            if cc is not None:
                for i in range(n): e.cc608_write(i, int(cc[fdone + i][0]), int(cc[fdone + i][1]))
            while e.audio_needed(n) > 0: e.audio_write(audio)
            e.render(n, slots=list(range(n)))
            cnt = e.frame_start(fdone + n) - e.frame_start(fdone)
            got.append(e.fetch(0, cnt)); fdone += n
        got = np.concatenate(got)
        if got.shape != want.shape:
            print("SHAPES", desc, got.shape, want.shape); return
        di = np.nonzero(got[:, 0] != want[:, 0])[0]; dq = np.nonzero(got[:, 1] != want[:, 1])[0]
        fs = e.frame_start(1) - e.frame_start(0)
        def where(d):
            if d.size == 0: return "-"
            return "%d differ, first %d (frame %d), got %s want %s, max |d| %d" % (d.size, d[0], d[0] // fs, got[d[0]].tolist(), want[d[0]].tolist(), int(np.abs(got[d].astype(int) - want[d].astype(int)).max()))
        print(("equal   " if di.size + dq.size == 0 else "DIFFERS ") + desc + " | I: " + where(di) + " | Q: " + where(dq), flush=True)

F, NA = H.FLAG_FILTER, H.FLAG_NOAUDIO
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1, vits=1, cc608=1), split=(1, 2))
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1))
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1), pics_kind="grey")
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1, vits=1))
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1, cc608=1))
run("ntsc", 27000000, 16000000, NA, dict(s_video=1))
run("ntsc", 27000000, 16000000, F | NA, dict(s_video=1), levels=1)
run("ntsc", 27000000, 13500000, F | NA, dict(s_video=1))
run("ntsc", 27000000, 18000000, F | NA, dict(s_video=1))
run("ntsc", 18000000, 16000000, F | NA, dict(s_video=1))
run("ntsc", 16000000, 13500000, F | NA, dict(s_video=1))
run("pal", 27000000, 16000000, F | NA, dict(s_video=1))
run("pal", 20250000, 16000000, F | NA, dict(s_video=1))
