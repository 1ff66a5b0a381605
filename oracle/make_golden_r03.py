#!/usr/bin/env python3
"""oracle/make_golden_r03.py -- TEST INFRASTRUCTURE (not product code).

Adds this round's resampler cases to tests/golden/ref_digests.json / ref_lines.npz without touching the others
(oracle/make_golden.py regenerates everything): --pixelrate together with --raw-bb-file (down and up), with
--passthru, with --s-video; rate pairs with frames of two lengths.

Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_r03.py
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
import util  # noqa: E402
from make_golden import GOLD, ref_cli  # noqa: E402

# id, base case (info / tables are taken from it: same mode at the same PIXEL rate), mode, sample rate, pixel rate, CLI flags, probe flags, real, frames, extra
CASES = [
    ("pal_rawbb_px135", "pal_px135_s136", "pal", 16000000, 13500000, ["--raw-bb-file", "@RAWBB@", "--pixelrate", "13500000"], 0, True, 2,
     {"raw_bb": 1, "raw_bb_blanking_level": 0, "raw_bb_white_level": 32767}),
    ("i_rawbb_px16", "i_full", "i", 13500000, 16000000, ["--filter", "--raw-bb-file", "@RAWBB@", "--raw-bb-blanking", "2000", "--raw-bb-white", "21000", "--pixelrate", "16000000", "--vits"],
     refprobe.FLAG_FILTER, False, 2, {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000, "vits": 1}),
    # --passthru behind the resampler: lines of varying width (870 / 871 at 13.5 -> 13.6 MHz), a source that ends inside frame 3
    ("i_pass_px135", "i_px135", "i", 16000000, 13500000, ["--filter", "--passthru", "@PASS@", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 4, {"passthru": 1}),
    ("pal_pass_px135_s136", "pal_px135_s136", "pal", 13600000, 13500000, ["--passthru", "@PASS@", "--pixelrate", "13500000"], 0, True, 4, {"passthru": 1}),
    # S-Video behind the resampler: its second channel (src/video.c:4361-4367)
    ("pal_sv_px135", "pal_sv", "pal", 16000000, 13500000, ["--s-video", "--pixelrate", "13500000"], 0, False, 2, {"s_video": 1}),
    ("ntsc_sv_f_px18", "ntsc_sv_f", "ntsc", 13500000, 18000000, ["--s-video", "--filter", "--pixelrate", "18000000"], refprobe.FLAG_FILTER, False, 2, {"s_video": 1}),
    ("secam_sv_f_px2025", "secam_sv", "secam", 16000000, 20250000, ["--s-video", "--filter", "--pixelrate", "20250000"], refprobe.FLAG_FILTER, False, 2, {"s_video": 1}),
    # rate pairs at which a raster frame is not a whole number of samples (858 x 525 x 32 / 27; 1017 x 525 x 27 / 32): frames of two lengths
    ("m_px135_s16", "m_full", "m", 16000000, 13500000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 5, {}),
    ("ntsc_px16_s135", "ntsc_bb", "ntsc", 13500000, 16000000, ["--pixelrate", "16000000"], 0, True, 4, {}),
]


def main():
    only = sys.argv[1:]
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    lines = dict(np.load(os.path.join(GOLD, "ref_lines.npz")))
    src = dict(np.load(os.path.join(GOLD, "testsrc.npz")))
    passfile = "/tmp/hvk_passthru.bin"
    util.passthru_signal().tofile(passfile)
    rawfile = "/tmp/hvk_rawbb.bin"
    util.rawbb_signal().tofile(rawfile)
    for cid, base, mode, sr, pr, flags, pflags, real, nframes, extra in CASES:
        if only and cid not in only:
            continue
        b = digests[base]
        L = b["lines"]
        with refprobe.RefProbe(mode, sr, pflags, pixel_rate=pr) as r:
            info = dict(r.info)
            key = "frame_%dx%d" % (info["active_width"], info["active_lines"])
            if key not in src:
                src[key] = r.test_frame()
        rs = info["width"] * L
        # frame f's first sample: where its first emitted line begins -- emitted line j is the resampler's chunk j + s, chunk g
        # begins at ceil(g W L / D) (hvk_tables.c:hvk_tables_frame_start) --, whether a frame is a whole number of samples or not
        sh = 1 + (1 if "--filter" in flags else 0)      # (the video filter: one line of latency, src/video.c:3620-3625)
        cut = lambda g: (g * info["width"] * sr + pr - 1) // pr
        ends = [cut((i + 1) * L + sh) - cut(sh) for i in range(nframes)]
        irregular = (rs * sr) % pr != 0
        fs = rs * sr // pr + (1 if irregular else 0)
        W = (rs * sr // pr) // L
        bps = 2 if real else 4
        cli = [f.replace("@PASS@", passfile).replace("@RAWBB@", rawfile) for f in flags]
        data = ref_cli(mode, sr, cli, ends[-1] * bps)
        again = ref_cli(mode, sr, cli, ends[-1] * bps)
        assert len(data) == ends[-1] * bps and data == again, cid
        per_frame = [hashlib.sha256(data[: ends[i] * bps]).hexdigest() for i in range(nframes)]
        a = np.frombuffer(data, np.int16)
        a = a.reshape(-1, 1) if real else a.reshape(-1, 2)
        pick = sorted(set([0, 1, 5, 6, 22, 23, 100, 309, 310, 312, 313, 335, 622, 623, L - 1, L, L + 1, L + 6, L + 100]))
        pick = [g for g in pick if g < nframes * L]
        lines[cid + "_idx"] = np.array(pick, np.int32)
        lines[cid] = np.stack([a[g * W:(g + 1) * W] for g in pick])
        digests[cid] = {
            "mode": mode, "sample_rate": sr, "cli_flags": flags, "probe_flags": pflags, "real": real,
            "width": W, "lines": L, "frames": nframes, "teletext": False, "extra": extra, "pixel_rate": pr, "frame_samples": fs,
            "sha256_cumulative": per_frame, "info": info, "tables": b["tables"],
        }
        if irregular:
            digests[cid]["frame_ends"] = ends
        print(cid, per_frame[-1][:16], flush=True)
    np.savez_compressed(os.path.join(GOLD, "ref_lines.npz"), **lines)
    np.savez_compressed(os.path.join(GOLD, "testsrc.npz"), **src)
    with open(dfile, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
