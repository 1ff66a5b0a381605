#!/usr/bin/env python3
"""oracle/make_golden_r04.py -- TEST INFRASTRUCTURE (not product code).

Round 4: the combinations the engine used to refuse. Adds to tests/golden/ref_digests.json / ref_lines.npz:

  pal_sv_sis     hacktv_ref -m pal -s 16000000 --s-video --sis dcsis            (sound-in-syncs beside the sub-carrier's own channel)
  i_rawbb_sis    hacktv_ref -m i -s 16000000 --filter --raw-bb-file ... --sis dcsis   (the burst on a line that was not drawn from a picture)
  i_sis_px135    hacktv_ref -m i -s 16000000 --filter --sis dcsis --pixelrate 13500000   (the burst drawn at the pixel rate, resampled)
  l_sis_px16_s14 hacktv_ref -m l -s 14000000 --filter --sis dcsis --pixelrate 16000000   (downwards: the filter reaches the next frame's second line)

and writes tests/golden/ref_undefined.json: configurations whose output the reference does not define -- the runs'
digests differ from one run of the same binary to the next (uninitialised memory the FM modulator then carries along
as a phase for ever). Every case here is run RUNS times; the digests of a defined case must all be equal.

Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_r04.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refprobe  # noqa: E402
import util  # noqa: E402
import make_golden_r03 as r03  # noqa: E402
from make_golden import GOLD, ref_cli  # noqa: E402

RUNS = 10
RAWBB = ["--raw-bb-file", "@RAWBB@", "--raw-bb-blanking", "2000", "--raw-bb-white", "21000"]
CASES = [
    ("pal_sv_sis", "pal_sv", "pal", 16000000, 16000000, ["--s-video", "--sis", "dcsis"], 0, False, 4, {"s_video": 1, "sis": 1}),
    ("i_rawbb_sis", "i_full", "i", 16000000, 16000000, ["--filter"] + RAWBB + ["--sis", "dcsis"], refprobe.FLAG_FILTER, False, 4,
     {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000, "sis": 1}),
    ("i_sis_px135", "i_px135", "i", 16000000, 13500000, ["--filter", "--sis", "dcsis", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 6, {"sis": 1}),
    ("i_sis_px2025", "i_px2025", "i", 16000000, 20250000, ["--sis", "dcsis", "--pixelrate", "20250000"], 0, False, 6, {"sis": 1}),
    # ... downwards with the filter on: the filter of a frame's last samples looks past the resampler's one-line lag into the burst of the
    # SECOND line of the next frame
    ("l_sis_px16_s14", "l_full", "l", 14000000, 16000000, ["--filter", "--sis", "dcsis", "--pixelrate", "16000000"], refprobe.FLAG_FILTER, False, 4, {"sis": 1}),
    # --raw-bb-file in a SECAM mode: the reference adds no colour process at all beside the line reader (src/video.c:4190 against :4206-4212)
    ("l_rawbb", "l_full", "l", 16000000, 16000000, ["--filter"] + RAWBB + ["--secam-field-id"], refprobe.FLAG_FILTER, False, 3,
     {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000, "secam_field_id": 1}),
    # FM video behind the resampler: the never-emitted start-up samples the modulator runs over are the resampled first raster line
    # (and the filter's output over it) -- up and down, with and without the pre-emphasis filter
    ("palfm_px135", "pal_fm", "pal-fm", 16000000, 13500000, ["--pixelrate", "13500000"], 0, False, 3, {}),
    ("palfm_f14_px135", "palfm_f14", "pal-fm", 14000000, 13500000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 3, {}),
    ("secamfm_px18", "secam_fm_tail", "secam-fm", 16000000, 18000000, ["--pixelrate", "18000000"], 0, False, 3, {}),
    ("ntscfm_f18_px135", "ntscfm_f18", "ntsc-fm", 18000000, 13500000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER, False, 3, {}),
    ("palfm_s14_px16", "pal_fm", "pal-fm", 14000000, 16000000, ["--pixelrate", "16000000"], 0, False, 3, {}),
    # ... with --passthru (lines of varying width behind the resampler: the sum is made frame by frame, behind the modulator)
    ("palfm_pass_px135", "pal_fm", "pal-fm", 16000000, 13500000, ["--passthru", "@PASS@", "--pixelrate", "13500000"], 0, False, 4, {"passthru": 1}),
    # ... and at a rate pair with frames of two lengths (1017 x 525 x 9 / 8): the modulator's place in the stream is what the frames add up to
    ("ntscfm_s18_px16", "ntsc_fm", "ntsc-fm", 18000000, 16000000, ["--pixelrate", "16000000"], 0, False, 5, {}),
]
# mode, sample rate, flags: what differs from run to run
UNDEFINED = [
    ("pal-fm", 16000000, ["--pixelrate", "18000000"]),
    ("pal-fm", 16000000, ["--filter", "--pixelrate", "18000000"]),
]
DEFINED_BESIDE = [
    ("pal-fm", 16000000, ["--pixelrate", "13500000"]),
    ("secam-fm", 16000000, ["--pixelrate", "18000000"]),
]


def runs_of(mode, sr, flags, nbytes, n):
    return [hashlib.sha256(ref_cli(mode, sr, flags, nbytes)).hexdigest() for _ in range(n)]


ONLY = sys.argv[1:]


def main():
    util.rawbb_signal().tofile("/tmp/hvk_rawbb.bin")
    util.passthru_signal().tofile("/tmp/hvk_passthru.bin")
    for cid, base, mode, sr, pr, flags, pflags, real, nframes, extra in CASES:
        cli = [f.replace("@RAWBB@", "/tmp/hvk_rawbb.bin").replace("@PASS@", "/tmp/hvk_passthru.bin") for f in flags]
        if ONLY and cid not in ONLY:
            continue
        d = runs_of(mode, sr, cli, 640000 * 4 * 3, RUNS)
        assert len(set(d)) == 1, (cid, d)
        print(cid, "%d identical runs" % RUNS, flush=True)
    r03.CASES = CASES
    sys.argv = sys.argv[:1] + ONLY
    r03.main()
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    for c in CASES:
        if c[0] in digests:
            digests[c[0]]["reference_runs"] = "%d runs of the reference CLI, one output" % RUNS
    json.dump(digests, open(dfile, "w"), indent=1, sort_keys=True)

    if ONLY:
        return
    und = {"note": "sha256 of the first 5 frames' worth of samples of RUNS runs of oracle/_ref/hacktv_ref (the unmodified reference), test source; "
                   "`undefined`: the runs differ from each other -- nothing to be equal to; `defined_beside_them`: neighbours whose runs agree",
           "runs": RUNS, "undefined": [], "defined_beside_them": []}
    for key, lst in (("undefined", UNDEFINED), ("defined_beside_them", DEFINED_BESIDE)):
        for mode, sr, flags in lst:
            d = runs_of(mode, sr, flags, 640000 * 4 * 5, RUNS)
            und[key].append({"cli": "hacktv_ref -m %s -s %d %s -o - test" % (mode, sr, " ".join(flags)), "sha256_of_each_run": d, "distinct": len(set(d))})
            print(key, mode, flags, len(set(d)), "distinct of", RUNS, flush=True)
    assert all(u["distinct"] > 1 for u in und["undefined"]) and all(u["distinct"] == 1 for u in und["defined_beside_them"])
    json.dump(und, open(os.path.join(GOLD, "ref_undefined.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
