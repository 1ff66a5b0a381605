#!/usr/bin/env python3
"""oracle/make_golden_sis.py -- TEST INFRASTRUCTURE (not product code).

Adds the sound-in-syncs cases to tests/golden/ref_digests.json / ref_lines.npz without touching the others:

  i_sis          hacktv_ref -m i -s 16000000 --sis dcsis -o - test
  i_sis_filter   ... --filter
  l_sis_tt       hacktv_ref -m l -s 16000000 --filter --sis dcsis --teletext raw:ttraw.bin  (SECAM-L, AM sound, teletext behind it)

The reference hands the audio block to its SiS process without a lock (src/sis.c:217-221, src/video.c:3370-3373), so
every case is run TEN times here and all ten outputs must be the same bytes; if they ever are not, the two outputs'
first frames and the ranges of samples that differ are written under tests/diag/sis_race/ and the script fails --
that, and not a sentence in a document, is what a refusal to render --sis would have to rest on.

Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_sis.py
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
from make_golden import GOLD, ref_cli  # noqa: E402

RUNS = 10
# id, base case (tables / info: --sis changes none of them), mode, rate, CLI flags, probe flags, frames, extra members, teletext
CASES = [
    ("i_sis",        "i_audio", "i", 16000000, ["--sis", "dcsis"],                          0,                    12, {"sis": 1}, False),
    ("i_sis_filter", "i_full",  "i", 16000000, ["--filter", "--sis", "dcsis"],              refprobe.FLAG_FILTER, 12, {"sis": 1}, False),
    ("l_sis_tt",     "l_tt",    "l", 16000000, ["--filter", "--sis", "dcsis", "--teletext", "raw:@TTRAW@"], refprobe.FLAG_FILTER, 3, {"sis": 1}, True),
]


def main():
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    lines = dict(np.load(os.path.join(GOLD, "ref_lines.npz")))
    ttraw = os.path.join(GOLD, "ttraw.bin")
    for case in CASES:
        cid, base, mode, sr, flags, pflags, nframes, extra, teletext = case[:9]
        skip = case[9] if len(case) > 9 else 0      # samples at the stream's start in which the reference's own runs differ (make_golden_r06.py)
        b = digests[base]
        W, L, fs = b["width"], b["lines"], b["frame_samples"]
        cli = [f.replace("@TTRAW@", ttraw) for f in flags]
        outs = [ref_cli(mode, sr, cli, nframes * fs * 4) for _ in range(RUNS)]
        for i, o in enumerate(outs[1:], 1):
            if o[skip * 4:] != outs[0][skip * 4:]:
                d = os.path.join(ROOT, "tests", "diag", "sis_race")
                os.makedirs(d, exist_ok=True)
                a = np.frombuffer(outs[0], np.int16).reshape(-1, 2)
                c = np.frombuffer(o, np.int16).reshape(-1, 2)
                bad = np.nonzero((a != c).any(axis=1))[0]
                np.savez_compressed(os.path.join(d, cid + ".npz"), run0=a[:fs], run=c[:fs], differing=bad)
                raise SystemExit("%s: run %d of the reference differs from run 0 at %d samples (first %d): written to %s" % (cid, i, bad.size, bad[0], d))
        data = outs[0]
        per_frame = [hashlib.sha256(data[skip * 4: (i + 1) * fs * 4]).hexdigest() for i in range(nframes)]
        a = np.frombuffer(data, np.int16).reshape(-1, 2)
        pick = sorted(set([0, 1, 2, 5, 6, 14, 15, 16, 22, 23, 31, 100, 309, 310, 312, 313, 335, 622, 623, L - 1, L, L + 1, L + 6, L + 100]))
        pick = [g for g in pick if g < nframes * L and (g + 1) * W > skip and g * W >= skip]
        lines[cid + "_idx"] = np.array(pick, np.int32)
        lines[cid] = np.stack([a[g * W:(g + 1) * W] for g in pick])
        digests[cid] = {
            "mode": mode, "sample_rate": sr, "cli_flags": flags, "probe_flags": pflags, "real": False,
            "width": W, "lines": L, "frames": nframes, "teletext": teletext, "extra": extra, "pixel_rate": 0, "frame_samples": fs,
            "sha256_cumulative": per_frame, "info": b["info"], "tables": b["tables"],
            "reference_runs": "%d runs of the reference CLI, one output" % RUNS,
        }
        if skip:
            digests[cid]["skip_samples"] = skip
            digests[cid]["reference_runs"] = "%d runs of the reference CLI: one output from sample %d on (the %d before it differ from run to run)" % (RUNS, skip, skip)
        print(cid, per_frame[-1][:16], "(%d identical runs)" % RUNS, flush=True)
    np.savez_compressed(os.path.join(GOLD, "ref_lines.npz"), **lines)
    with open(dfile, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
