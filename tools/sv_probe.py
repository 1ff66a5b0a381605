#!/usr/bin/env python3
"""tools/sv_probe.py -- S-Video behind resampler + filter where the lines have two widths: the engine's Q channel as it is
(the sub-carrier at the luma's own position, HVK_SV_EXPERIMENT=1 lifts the refusal) against the oracle's ring of line buffers,
line by line: with which shift does each line agree, and which samples remain."""
import os, sys
os.environ["HVK_SV_EXPERIMENT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import oracle, util
g = util.Golden()
for mode, sr, pr in (("ntsc", 16000000, 27000000), ("ntsc", 16000000, 13500000), ("pal60", 16000000, 18000000)):
    conf = H.preset(mode, H.FLAG_FILTER)
    conf.s_video = 1
    nf = 3
    with H.Engine(conf, sr, device=0, max_frames=nf, pixel_rate=pr) as e:
        fb = g.frame("m_full")
        e.frame_upload(0, fb)
        e.render(nf)
        total = e.frame_start(nf) - e.frame_start(0)
        got = e.fetch(0, total)
        lines = e.info["lines"]
        widths = e.line_widths(0, nf * lines)
    with oracle.Oracle(conf, sr, pixel_rate=pr) as o:
        o.set_frame(fb)
        want = o.render_lines(nf * lines)
    print(mode, sr, pr, "samples", len(got), len(want), "I equal:", np.array_equal(got[:, 0], want[:len(got), 0]))
    pos = 0
    hist = {}
    rest = []
    gq, wq = got[:, 1].astype(np.int32), want[:, 1].astype(np.int32)
    for ln, w in enumerate(widths):
        seg_w = wq[pos:pos + w]
        best = None
        for sh in (-2, -1, 0, 1, 2):
            a, b = pos + sh, pos + sh + w
            if a < 0 or b > len(gq):
                continue
            seg_g = gq[a:b]
            nbad = int((seg_g != seg_w).sum())
            if best is None or nbad < best[1]:
                best = (sh, nbad, np.nonzero(seg_g != seg_w)[0][:4].tolist())
        hist[(int(w), best[0], best[1])] = hist.get((int(w), best[0], best[1]), 0) + 1
        if best[1] and len(rest) < 12:
            rest.append((ln, int(w), int(widths[ln - 1]) if ln else None, best))
        pos += w
    print("  (width, shift, mismatches) -> lines:", sorted(hist.items()))
    print("  first lines with a rest:", rest)
