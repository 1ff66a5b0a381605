#!/usr/bin/env python3
"""tools/secam_warmup_sweep.py [frames] [card|noisy] -- SECAM-L: time of stage + launch of a fresh block and the lines
that started wrong, for every fixed number of warm-up lines (HVK_SECAM_WARMUP=K). Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "card"
FS = 640000
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:576, 0:832]
pics = []
for i in range(4):
    p = (((xx * 255 // 831 + i * 17) % 256).astype(np.uint32) << 16) | (((yy * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx + yy) // 3 % 256).astype(np.uint32))
    pics.append(np.where(rng.random(p.shape) < 0.2, rng.integers(0, 1 << 24, p.shape, dtype=np.uint32), p).astype(np.uint32))
if kind == "card":
    pics = [g.frame("l_full")] * 4
for K in (12, 11, 10, 9, 8, 7, 6, 5, 4, 3):
    os.environ["HVK_SECAM_WARMUP"] = str(K)
    with H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=F) as e:
        for s in range(4):
            e.frame_upload(s, pics[s])
        slots = [i % 4 for i in range(F)]
        for b in range(2):
            e.stage(b * F, 1, F, slots=slots); e.launch()
        e.sync()
        st0 = e.secam_stats()
        t0 = time.perf_counter()
        n = 4
        for b in range(2, 2 + n):
            e.stage(b * F, 1, F, slots=slots); e.launch()
        e.sync()
        t = (time.perf_counter() - t0) / n
        st = e.secam_stats()
        d = {k: st[k] - st0[k] for k in st}
        print("K=%2d  %.3f ms per block = %.1f Gsamples/s; %s" % (K, t * 1e3, F * FS / t * 1e-9, d), flush=True)
