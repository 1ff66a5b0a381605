#!/usr/bin/env python3
"""oracle/make_golden_rates.py -- TEST INFRASTRUCTURE (not product code).

Adds cases at sample rates outside the 11 .. 28 MHz the first rounds' kernels covered (chroma filters of 7, 19 and 23
taps: 8, 9, 24, 28.6 and 30 MHz) to tests/golden/ref_digests.json, ref_lines.npz and testsrc.npz without touching the
other cases. Every case is run twice: at rates where the reference's chroma over-read lands on allocator pointers
 its own output changes from run to run and there is nothing to pin -- those rates are not in the list.

Run from the repository root after `make -C oracle ref`:  python oracle/make_golden_rates.py
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

CASES = [
    ("pal_8m",     "pal",  8000000,  [],           0,                    True,  2),   # 512-sample lines: the raster + filter kernel pair
    ("pal_9m",     "pal",  9000000,  [],           0,                    True,  2),   # 7-tap chroma filter, 576-sample lines
    ("i_24m",      "i",    24000000, ["--filter"], refprobe.FLAG_FILTER, False, 2),   # 19 taps, NICAM pulse of 331 taps
    ("ntsc_24m",   "ntsc", 24000000, [],           0,                    True,  2),   # 19 taps, 1525-sample lines
    ("m_4fsc",     "m",    28636360, ["--filter"], refprobe.FLAG_FILTER, False, 2),   # 23 taps, 4 x the NTSC sub-carrier: 1820-sample lines
    ("pal_30m",    "pal",  30000000, [],           0,                    True,  2),   # 23 taps, 1920-sample lines
]


def main():
    only = sys.argv[1:]
    dfile = os.path.join(GOLD, "ref_digests.json")
    digests = json.load(open(dfile))
    lines = dict(np.load(os.path.join(GOLD, "ref_lines.npz")))
    src = dict(np.load(os.path.join(GOLD, "testsrc.npz")))
    for cid, mode, sr, flags, pflags, real, nframes in CASES:
        if only and cid not in only:
            continue
        with refprobe.RefProbe(mode, sr, pflags) as r:
            info = dict(r.info)
            key = "frame_%dx%d" % (info["active_width"], info["active_lines"])
            if key not in src:
                src[key] = r.test_frame()
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
        pick = sorted(set([0, 1, 5, 6, 22, 23, 100, 309, 310, 312, 313, 335, L - 3, L - 2, L - 1, L, L + 1, L + 6, L + 100]))
        pick = [g for g in pick if 0 <= g < nframes * L]
        lines[cid + "_idx"] = np.array(pick, np.int32)
        lines[cid] = np.stack([a[g * W:(g + 1) * W] for g in pick])
        digests[cid] = {
            "mode": mode, "sample_rate": sr, "cli_flags": flags, "probe_flags": pflags, "real": real,
            "width": W, "lines": L, "frames": nframes, "teletext": False, "extra": {}, "pixel_rate": 0, "frame_samples": fs,
            "sha256_cumulative": per_frame, "info": info, "tables": tabs,
        }
        print(cid, W, info.get("chroma_ntaps"), per_frame[-1][:16], flush=True)
    np.savez_compressed(os.path.join(GOLD, "testsrc.npz"), **src)
    np.savez_compressed(os.path.join(GOLD, "ref_lines.npz"), **lines)
    with open(dfile, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
