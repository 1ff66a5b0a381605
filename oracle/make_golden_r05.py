#!/usr/bin/env python3
"""oracle/make_golden_r05.py -- TEST INFRASTRUCTURE (not product code).

Round 5: S-Video behind the resampler AND the video filter where the lines have two widths (525 lines at 16 MHz: 1017, 1017,
..., 1016) -- the reference gives a line's luma the width of the chunk the filter was last fed and pairs it with the
sub-carrier its ring of line buffers holds (src/video.c:3243, :3578): the last combination the engine refused. Adds to
tests/golden/ref_digests.json / ref_lines.npz:

  ntsc_sv_f_px135_s16   hacktv_ref -m ntsc -s 16000000 --s-video --filter --pixelrate 13500000   (upwards: a line's old content is the raster's blanking)
  ntsc_sv_f_px18_s16    hacktv_ref -m ntsc -s 16000000 --s-video --filter --pixelrate 18000000   (downwards: the raster's sub-carrier of the line before)
  pal60_sv_f_px27_s16   hacktv_ref -m pal60 -s 16000000 --s-video --filter --pixelrate 27000000
  ntsc_sv_f_px16_s27    hacktv_ref -m ntsc -s 27000000 --s-video --filter --pixelrate 16000000   (upwards, lines of 1716 / 1717: found by tools/fuzz_parity.py seed 2718)
  ntsc_sv_f_px16_s18    hacktv_ref -m ntsc -s 18000000 --s-video --filter --pixelrate 16000000   (upwards, lines of 1144 / 1145)

Every case is run RUNS times: one output. Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_r05.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refprobe  # noqa: E402
import make_golden_r03 as r03  # noqa: E402
from make_golden import GOLD, ref_cli  # noqa: E402

RUNS = 5
CASES = [
    ("ntsc_sv_f_px135_s16", "ntsc_sv_f", "ntsc", 16000000, 13500000, ["--s-video", "--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 4, {"s_video": 1}),
    ("ntsc_sv_f_px18_s16", "ntsc_sv_f", "ntsc", 16000000, 18000000, ["--s-video", "--filter", "--pixelrate", "18000000"], refprobe.FLAG_FILTER, False, 4, {"s_video": 1}),
    ("pal60_sv_f_px27_s16", "pal60_bb", "pal60", 16000000, 27000000, ["--s-video", "--filter", "--pixelrate", "27000000"], refprobe.FLAG_FILTER, False, 3, {"s_video": 1}),
    # (upwards from the raster's 1017 samples a line: most lines the SHORTER of two widths, and the first chunk the filter is fed a short one)
    ("ntsc_sv_f_px16_s27", "ntsc_sv_f", "ntsc", 27000000, 16000000, ["--s-video", "--filter", "--pixelrate", "16000000"], refprobe.FLAG_FILTER, False, 3, {"s_video": 1}),
    ("ntsc_sv_f_px16_s18", "ntsc_sv_f", "ntsc", 18000000, 16000000, ["--s-video", "--filter", "--pixelrate", "16000000"], refprobe.FLAG_FILTER, False, 3, {"s_video": 1}),
]


def main():
    only = sys.argv[1:]
    for cid, base, mode, sr, pr, flags, pflags, real, nframes, extra in CASES:
        if only and cid not in only:
            continue
        d = [hashlib.sha256(ref_cli(mode, sr, flags, 534000 * 4 * 3)).hexdigest() for _ in range(RUNS)]
        assert len(set(d)) == 1, (cid, d)
        print(cid, "%d identical runs" % RUNS, flush=True)
    r03.CASES = CASES
    r03.main()
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    for c in CASES:
        if c[0] in digests:
            digests[c[0]]["reference_runs"] = "%d runs of the reference CLI, one output" % RUNS
    json.dump(digests, open(dfile, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
