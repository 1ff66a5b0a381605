#!/usr/bin/env python3
"""tools/d2h_speed.py -- one engine, one 128-frame block of -m i --filter --noaudio: render + read-back into page-locked memory,
five times into the same buffer, with the read-back whole and -- an experiment of round 5, since removed: HVK_FETCH_SPLIT_MB -- in two halves on two streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H
import util
g = util.Golden()
F, FS = 128, 640000
for split in ("0", "16"):
    os.environ["HVK_FETCH_SPLIT_MB"] = split
    with H.Engine(H.preset("i", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        hb = e.host_buffer(F * FS)
        e.stage(0, 1, F); e.launch(); e.sync()
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            e.launch()
            e.fetch_wait(e.fetch_async(hb, 0, F * FS))
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        e.fetch_wait(e.fetch_async(hb, 0, F * FS))
        tc = time.perf_counter() - t0
        print("HVK_FETCH_SPLIT_MB=%s: render + D2H per block (ms): %s -> %.1f Gsamples/s = %.1f GB/s steady; the copy alone %.2f ms = %.1f GB/s"
              % (split, " ".join("%.2f" % (t * 1e3) for t in ts), F * FS / min(ts) * 1e-9, F * FS * 4 / min(ts) * 1e-9, tc * 1e3, F * FS * 4 / tc * 1e-9))
