#!/usr/bin/env python3
"""tools/secam_blocks.py [frames] [card|noisy|new] [steps] -- SECAM-L (-m l -s 16000000 --filter --noaudio): stage + launch of
fresh blocks, the warm-up length left to the engine; what tools/profile_round.sh runs under rocprofv3 for the tracked
SECAM kernel summaries. `noisy`: four noisy pictures and HVK_SECAM_NO_CELL_CACHE=1 (the cells made for every frame, as
with a moving source). `new`: a picture slot per frame of the block and every slot's planes made again in every step
(hvk_planes_refresh) as well -- everything a new picture on every frame costs on the device. Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "card"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
FS = 640000
if kind == "card":
    pics = [g.frame("l_full")]
else:
    os.environ["HVK_SECAM_NO_CELL_CACHE"] = "1"
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:576, 0:832]
    pics = []
    for i in range(4):
        p = (((xx * 255 // 831 + i * 17) % 256).astype(np.uint32) << 16) | (((yy * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx + yy) // 3 % 256).astype(np.uint32))
        pics.append(np.where(rng.random(p.shape) < 0.2, rng.integers(0, 1 << 24, p.shape, dtype=np.uint32), p).astype(np.uint32))
with H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=F) as e:
    new = kind == "new"
    if new:
        for s in range(F):
            e.frame_upload(s, pics[s % len(pics)])
        slots = list(range(F))
    else:
        for s, p in enumerate(pics):
            e.frame_upload(s, p)
        slots = [i % len(pics) for i in range(F)]
    for b in range(4):
        if new: e.planes_refresh(slots)
        e.stage(b * F, 1, F, slots=slots); e.launch()
    e.sync()
    st0 = e.secam_stats()
    t0 = time.perf_counter()
    for b in range(4, 4 + steps):
        if new: e.planes_refresh(slots)
        e.stage(b * F, 1, F, slots=slots); e.launch()
    e.sync()
    t = (time.perf_counter() - t0) / steps
    st = e.secam_stats()
    print("%s, %d frames per block: %.3f ms per block = %.1f Gsamples/s; warm-up lines %d; lines of the timed blocks %s; walk kernels (ok, [chain, walk<0>, walk<1>]) %s"
          % (kind, F, t * 1e3, F * FS / t * 1e-9, e.secam_warmup_lines(), {k: st[k] - st0[k] for k in st}, e.secam_walk_stages()))
