#!/usr/bin/env python3
"""tools/secam_speed.py [frames] -- SECAM-L (-m l -s 16000000 --filter --noaudio): time of staging a block (the colour
sub-carrier chain: on the device by default, on the host with HVK_SECAM_HOST=1) and of rendering it. Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FS = 640000
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:576, 0:832]
pics = []
for i in range(4):
    p = (((xx * 255 // 831 + i * 17) % 256).astype(np.uint32) << 16) | (((yy * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx + yy) // 3 % 256).astype(np.uint32))
    pics.append(np.where(rng.random(p.shape) < 0.2, rng.integers(0, 1 << 24, p.shape, dtype=np.uint32), p).astype(np.uint32))
CARD = len(sys.argv) > 2 and sys.argv[2] == "card"
if CARD:
    pics = [g.frame("l_full")] * 4
for label, env in (("device chain", {}), ("host chain", {"HVK_SECAM_HOST": "1"})):
    for k in ("HVK_SECAM_HOST",):
        os.environ.pop(k, None)
    os.environ.update(env)
    with H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=F) as e:
        for s in range(4):
            e.frame_upload(s, pics[s] if label else None)
        slots = [i % 4 for i in range(F)]
        ts = []
        for b in range(3 if not env else 1):
            e.sync()
            t0 = time.time()
            e.stage(b * F, 1, F, slots=slots)
            e.sync()
            ts.append(time.time() - t0)
            t1 = time.time()
            e.launch()
            e.sync()
            tl = time.time() - t1
        st = e.secam_stats()
        print("%-13s stage %s s -> %.1f Msamples/s; launch %.2f ms; %s" % (label, " ".join("%.4f" % t for t in ts), F * FS / min(ts) * 1e-6, tl * 1e3, st))
