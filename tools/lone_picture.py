#!/usr/bin/env python3
"""tools/lone_picture.py -- a batch of ONE frame whose picture is new (hvk_planes_refresh of its slot, stage, launch, waited for), PAL-I
--filter --noaudio and NTSC-M at 13.5 MHz (no fused kernel: 858 samples a line), levels from the table and computed: wall time per
frame here; the kernels' own durations from  tools/kstats.sh python $PWD/tools/lone_picture.py .  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util
g = util.Golden()
for mode, sr in (("i", 16000000), ("m", 13500000)):
    for lv, name in ((1, "table"), (2, "computed")):
        with H.Engine(H.preset(mode, H.FLAG_FILTER | H.FLAG_NOAUDIO), sr, device=0, max_frames=4) as e:
            e.set_levels(lv)
            hh, ww = e.info["active_lines"], e.info["active_width"]
            e.frame_upload(0, np.ascontiguousarray(g.frame("i_full")[:hh, :ww]))
            e.stage(0, 1, 1, slots=[0]); e.launch(); e.sync()
            n = 300
            t0 = time.perf_counter()
            for k in range(n):
                e.planes_refresh([0]); e.stage(1 + k, 1, 1, slots=[0]); e.launch(); e.sync()
            print("-m %s levels %-8s: one frame with a new picture, staged, rendered and waited for: %.1f us" % (mode, name, (time.perf_counter() - t0) / n * 1e6), flush=True)
