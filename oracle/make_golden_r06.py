#!/usr/bin/env python3
"""oracle/make_golden_r06.py -- TEST INFRASTRUCTURE (not product code).

Round 6: configurations the engine refused until now for a kernel table's size, not for anything in the reference
(Missing 5). Adds to tests/golden/ref_digests.json / ref_lines.npz / testsrc.npz:

  pal_36m        hacktv_ref -m pal -s 36000000                       (chroma low pass of more than 25 taps: 2304-sample lines)
  pal_8fsc       hacktv_ref -m pal -s 35468950                       (8 x the PAL sub-carrier: 2270-sample lines)
  i_36m          hacktv_ref -m i -s 36000000 --filter                (... with the video filter, FM sound and a NICAM pulse of 495 taps)
  i_27m          hacktv_ref -m i -s 27000000 --filter                (the base of the next)
  pal_27m        hacktv_ref -m pal -s 27000000                       (the base of the next)
  pal_px27_s4fsc hacktv_ref -m pal -s 17734475 --pixelrate 27000000  (a resampler of 709379 phases, src/fir.c:393-428: refused above 256 until now)
  i_px27_s4fsc   hacktv_ref -m i -s 17734475 --filter --pixelrate 27000000
  apollofm_f     hacktv_ref -m apollo-fm -s 8000000 --filter         (FM video pre-emphasis on 320 lines: the 625-line 20.25 MHz table, src/video.c:3711-3734)
  apollofscfm_f  hacktv_ref -m apollo-fsc-fm -s 13500000 --filter    (525 lines, field-sequential colour: the 525-line 20.25 MHz table)
  i_sis_27m      hacktv_ref -m i -s 27000000 --filter --sis dcsis    (sound-in-syncs bursts longer than 128 samples, src/sis.c:155-201;
                                                                      "skip_samples": 64 -- the stream's first 28 samples are not the reference's to say)

Rates at which the reference's own output changes from run to run (its chroma low pass reads past its buffer into allocator
words, SURVEY.md H2: 34, 40 MHz PAL, 36 MHz NTSC among them) have nothing to pin and are not in the list; every case here is
run several times and gave one output. Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_r06.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refprobe  # noqa: E402
import make_golden_rates as rates  # noqa: E402
import make_golden_sis as sis  # noqa: E402
import make_golden_r03 as r03  # noqa: E402
import make_golden_rasters as rasters  # noqa: E402

RATE_CASES = [
    ("pal_36m",  "pal", 36000000, [],           0,                    True,  2),
    ("pal_8fsc", "pal", 35468950, [],           0,                    True,  2),
    ("i_36m",    "i",   36000000, ["--filter"], refprobe.FLAG_FILTER, False, 2),
    ("i_27m",    "i",   27000000, ["--filter"], refprobe.FLAG_FILTER, False, 2),
    ("pal_27m",  "pal", 27000000, [],           0,                    True,  2),     # (the base of pal_px27_s4fsc)
]
# --pixelrate pairs of more than 256 phases (27 MHz -> 4 x the PAL sub-carrier is 709379 : 1080000 in lowest terms: a resampler of
# 14 896 959 taps, src/fir.c:404): id, base case (same mode at the same PIXEL rate), mode, sample rate, pixel rate, CLI flags, probe flags, real, frames, extra
PIXELRATE_CASES = [
    ("pal_px27_s4fsc", "pal_27m", "pal", 17734475, 27000000, ["--pixelrate", "27000000"], 0, True, 3, {}),
    ("i_px27_s4fsc",   "i_27m",   "i",   17734475, 27000000, ["--filter", "--pixelrate", "27000000"], refprobe.FLAG_FILTER, False, 3, {}),
    # sound-in-syncs where the frames have two lengths (the fuzzer's find of round 6: every frame of a batch its own bursts; SECAM's
    # chains four lines ahead of the requests)
    ("i_sis_px2025_s4fsc", "i_20m",    "i", 17734475, 20250000, ["--filter", "--sis", "dcsis", "--pixelrate", "20250000"], refprobe.FLAG_FILTER, False, 4, {"sis": 1}),
    # S-Video behind resampler + filter where the LINES have two widths and the frames one length (the fuzzer's second find of round 6)
    # (the find itself was from 18 MHz pixels; PAL at 18 MHz -- as a sample rate or a pixel rate, with or without S-Video -- is one of the rates at
    # which the reference's own runs differ from each other, in one to two samples of a hundred from the first colour line on: what its chroma
    # low pass reads past its buffer is another thread's memory there. Nothing to pin: the cases below are from 16 and 27 MHz pixels, three runs one output)
    ("pal_sv_f_px27_s4fsc", "pal_sv", "pal", 17734475, 27000000, ["--s-video", "--filter", "--pixelrate", "27000000"], refprobe.FLAG_FILTER, False, 3, {"s_video": 1}),
    ("pal_sv_f_px16_s4fsc", "pal_sv", "pal", 17734475, 16000000, ["--s-video", "--filter", "--pixelrate", "16000000"], refprobe.FLAG_FILTER, False, 3, {"s_video": 1}),
    ("l_sis_px2025_s4fsc", "l_px2025", "l", 17734475, 20250000, ["--filter", "--sis", "dcsis", "--pixelrate", "20250000"], refprobe.FLAG_FILTER, False, 4, {"sis": 1}),
]
# FM video's pre-emphasis filter on a raster that has neither 625 nor 525 lines: the reference takes its 625-line tables for every
# count but 525 (src/video.c:3693, :3711) -- Apollo's 320 lines at 8 MHz get the 20.25 MHz table and a warning
RASTER_CASES = [
    ("apollofm_f",    "apollo-fm",      8000000, ["--filter"], refprobe.FLAG_FILTER, False, 3),
    ("apollofscfm_f", "apollo-fsc-fm", 13500000, ["--filter"], refprobe.FLAG_FILTER, False, 3),
]
SIS_CASES = [
    # (the reference's first 28 samples differ from run to run at this rate -- its burst renderer's first, never-emitted invocation
    # reads a line buffer nobody has written: hashed from sample 64 on, every run the same from there)
    ("i_sis_27m", "i_27m", "i", 27000000, ["--filter", "--sis", "dcsis"], refprobe.FLAG_FILTER, 6, {"sis": 1}, False, 64),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    rates.CASES = [c for c in RATE_CASES if not only or c[0] in only]
    sys.argv = sys.argv[:1]
    if rates.CASES:
        rates.main()
    sis.CASES = [c for c in SIS_CASES if not only or c[0] in only]
    if sis.CASES:
        sis.main()
    rasters.CASES = [c for c in RASTER_CASES if not only or c[0] in only]
    if rasters.CASES:
        rasters.main()
    r03.CASES = [c for c in PIXELRATE_CASES if not only or c[0] in only]
    if r03.CASES:
        r03.main()
