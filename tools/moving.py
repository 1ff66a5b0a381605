#!/usr/bin/env python3
"""tools/moving.py -- device-only rate when every frame of the block is a different picture
(the 7 B/sample regime): the static test card, a natural-looking moving picture
(smooth gradients + noise of a few LSB) and uniform noise (every pixel a random 24-bit colour:
the worst case for the 2^24-entry level table). Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(1)
h, w = 576, 832
yy, xx = np.mgrid[0:h, 0:w]


def natural(i):
    r = (xx * 255 // w + 3 * i) % 256
    gch = (yy * 255 // h + 5 * i) % 256
    b = ((xx + yy) * 255 // (w + h) + 7 * i) % 256
    n = rng.integers(-3, 4, size=(3, h, w))
    r, gch, b = [np.clip(c + d, 0, 255).astype(np.uint32) for c, d in zip((r, gch, b), n)]
    return (r << 16) | (gch << 8) | b


kinds = [("static test card", lambda i: g.frame("i_full")),
         ("moving, natural-looking", natural),
         ("moving, uniform noise", lambda i: rng.integers(0, 1 << 24, size=(h, w), dtype=np.uint32))]
conf = H.preset("i", H.FLAG_FILTER)
for name, make in kinds:
  for mode, mname in ((1, "table"), (2, "computed"), (0, "auto")):
    with H.Engine(conf, 16000000, device=0, max_frames=F) as e:
        e.set_levels(mode)
        slots = min(F, e.info["frame_slots"])
        for s in range(slots):
            e.frame_upload(s, make(s))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F, slots=[i % slots for i in range(F)])
        for _ in range(2):
            e.launch()
        e.sync()
        e.timing_enable(True)
        for _ in range(10):
            e.launch()
        r, _ = e.timing_read(0)
        f, _ = e.timing_read(1)
        fs = e.info["frame_samples"]
        print("%-26s levels %-9s raster %.4f ms  filter %.4f ms  -> %.1f Gsamples/s" % (name, mname, r, f, F * fs / (r + f) / 1e6), flush=True)
