#!/usr/bin/env python3
"""oracle/make_golden_rasters.py -- TEST INFRASTRUCTURE (not product code).

The rasters of src/video.c:2592-2862 other than 625 and 525 lines -- 819, 405, CBS 405, Apollo 320, Baird 240 and 30, NBTV 32 --
and the field-sequential colour modes, through the presets of src/video.c:1988-2006: digests and line excerpts of the unmodified
reference CLI's output (run three times: it has to say the same thing every time), its tables and its test source's pictures,
added to tests/golden/ref_digests.json, ref_lines.npz and testsrc.npz without touching the other cases.

The mechanical systems scan vertically: main() opens the test source with width and height exchanged (src/hacktv.c:1520-1526)
and the engine turns every picture (src/video.c:4883-4885: rotate by 270 degrees, mirror). The fixture keeps the picture as
the raster shows it -- turned --, which is what an engine that "shows what it is given" is handed; the turning is done here with
the reference's own stride arithmetic (src/av.c:242-290).

Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_rasters.py [case ...]
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refprobe  # noqa: E402
from make_golden import GOLD, TABLES, ref_cli  # noqa: E402

F = refprobe.FLAG_FILTER
CASES = [
    # id, mode, sample rate, CLI flags, probe flags, real, frames
    ("e_full",       "e",             24570000, ["--filter"], F, False, 2),   # 819 lines of 1200 samples, VSB, AM sound at 11.15 MHz
    ("819_bb",       "819",           16380000, [],           0, True,  2),   # 800-sample lines
    ("a_full",       "a",              8100000, ["--filter"], F, False, 2),   # 405 lines of 800 samples, AM sound BELOW the vision carrier
    ("405i_full",    "405-i",         16200000, ["--filter"], F, False, 2),
    ("405_bb",       "405",            8100000, [],           0, True,  2),
    ("ntsc405_bb",   "ntsc-405",       8100000, [],           0, True,  2),   # NTSC colour on 405 lines
    ("ntsca_full",   "ntsc-a",         8100000, ["--filter"], F, False, 3),   # ... with VSB and sound, and NO chroma low pass (the preset sets none)
    ("240am",        "240-am",         4800000, [],           0, False, 2),   # Baird 240: the broad pulse at mid-line runs on into the next line
    ("240_bb",       "240",            4800000, [],           0, True,  3),
    ("30_bb",        "30",              750000, [],           0, True,  3),   # Baird 30 lines: no sync, scanned vertically
    ("30am",         "30-am",           750000, [],           0, False, 2),
    ("nbtv_bb",      "nbtv",            800000, [],           0, True,  3),
    ("nbtvam",       "nbtv-am",         800000, [],           0, False, 2),
    ("apollo_bb",    "apollo",         3200000, [],           0, True,  2),
    ("apollofm",     "apollo-fm",      8000000, [],           0, False, 2),
    ("apollofsc_bb", "apollo-fsc",    13500000, [],           0, True,  4),   # field-sequential colour: three fields make the sequence
    ("apollofscfm",  "apollo-fsc-fm", 13500000, [],           0, False, 3),
    ("cbs405_bb",    "cbs405",        17496000, [],           0, True,  4),   # 600-sample lines, 72 frames a second
    ("mcbs405_full", "m-cbs405",      17496000, ["--filter"], F, False, 4),
]


def oriented(src, orientation):
    """The picture the raster shows of source picture `src` ([h][w]): av_rotate_frame / av_hflip_frame / av_vflip_frame
    (src/av.c:242-290) as the pointer and stride arithmetic they are."""
    h, w = src.shape
    base, ps, ls = 0, 1, w
    a = orientation & 3
    if a in (1, 3):
        base += (h - 1) * ls
        w, h = h, w
        ps, ls = -ls, ps
    if a in (2, 3):
        base += (w - 1) * ps; ps = -ps
        base += (h - 1) * ls; ls = -ls
    if orientation & 4:
        base += (w - 1) * ps; ps = -ps
    if orientation & 8:
        base += (h - 1) * ls; ls = -ls
    flat = src.reshape(-1)
    yy, xx = np.mgrid[0:h, 0:w]
    return flat[base + yy * ls + xx * ps].copy()


def main():
    only = sys.argv[1:]
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    lines = dict(np.load(os.path.join(GOLD, "ref_lines.npz")))
    src = dict(np.load(os.path.join(GOLD, "testsrc.npz")))
    sys.path.insert(0, ROOT)
    import hacktv_amd as H
    for cid, mode, sr, flags, pflags, real, nframes in CASES:
        if only and cid not in only:
            continue
        conf = H.preset(mode, pflags)
        with refprobe.RefProbe(mode, sr, pflags) as r:
            info = dict(r.info)
            n = refprobe.lib().ref_test_frame(r.p, None, 0)
            raw = np.zeros(n, np.uint32)
            refprobe.lib().ref_test_frame(r.p, raw.ctypes.data, n)
            rot = (conf.frame_orientation & 3) in (1, 3)
            raw = raw.reshape((info["active_width"], info["active_lines"]) if rot else (info["active_lines"], info["active_width"]))
            key = "frame_%dx%d" % (info["active_width"], info["active_lines"])
            pic = oriented(raw, conf.frame_orientation)
            assert pic.shape == (info["active_lines"], info["active_width"]), (cid, pic.shape)
            if conf.frame_orientation:
                key += "_o%d" % conf.frame_orientation
            src[key] = pic
            tabs = {}
            for name, dt in TABLES:
                a = r.table(name, dt)
                tabs[name] = {"len": int(a.size), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
        W, L = info["width"], info["lines"]
        fs = W * L
        bps = 2 if real else 4
        runs = [ref_cli(mode, sr, flags, nframes * fs * bps) for _ in range(3)]
        assert all(len(d) == nframes * fs * bps and d == runs[0] for d in runs), cid + ": the reference's output changes from run to run"
        data = runs[0]
        per_frame = [hashlib.sha256(data[: (i + 1) * fs * bps]).hexdigest() for i in range(nframes)]
        a = np.frombuffer(data, np.int16)
        a = a.reshape(-1, 1) if real else a.reshape(-1, 2)
        pick = sorted(set([0, 1, 2, 3, 4, 5, 8, 12, 13, 17, 20, 21, 40, 100, 201, 202, 203, 206, 217, 280, 405, 408, 446, L // 2, L - 3, L - 2, L - 1, L, L + 1, L + 4, L + 17, L + 100, 2 * L, 2 * L + 17]))
        pick = [g for g in pick if 0 <= g < nframes * L]
        lines[cid + "_idx"] = np.array(pick, np.int32)
        lines[cid] = np.stack([a[g * W:(g + 1) * W] for g in pick])
        digests[cid] = {
            "mode": mode, "sample_rate": sr, "cli_flags": flags, "probe_flags": pflags, "real": real,
            "width": W, "lines": L, "frames": nframes, "teletext": False, "extra": {}, "pixel_rate": 0, "frame_samples": fs,
            "sha256_cumulative": per_frame, "info": info, "tables": tabs, "frame_key": key,
        }
        print(cid, W, L, info.get("olines"), per_frame[-1][:16], flush=True)
    np.savez_compressed(os.path.join(GOLD, "testsrc.npz"), **src)
    np.savez_compressed(os.path.join(GOLD, "ref_lines.npz"), **lines)
    with open(dfile, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
