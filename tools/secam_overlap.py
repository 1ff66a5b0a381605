#!/usr/bin/env python3
"""tools/secam_overlap.py [frames] [engines] [steps] -- SECAM-L with a new picture on every frame (tools/secam_blocks.py `new`),
the block's frames split over ENGINES engines on the one device, each driven by its own host thread: how much of the colour
chain's kernels (vector-issue bound) hides behind the planes' and the render's (HBM bound) when their launches overlap.
Every engine renders frames of its own (a stream each; not one SECAM stream split in blocks). Run on the GPU box."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
FS = 640000
os.environ["HVK_SECAM_NO_CELL_CACHE"] = "1"
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:576, 0:832]
pics = []
for i in range(4):
    p = (((xx * 255 // 831 + i * 17) % 256).astype(np.uint32) << 16) | (((yy * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx + yy) // 3 % 256).astype(np.uint32))
    pics.append(np.where(rng.random(p.shape) < 0.2, rng.integers(0, 1 << 24, p.shape, dtype=np.uint32), p).astype(np.uint32))
Fe = F // N
engines = [H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=Fe) for _ in range(N)]
slots = list(range(Fe))
for e in engines:
    for s in range(Fe):
        e.frame_upload(s, pics[s % len(pics)])

def run(e, b0, n):
    for b in range(b0, b0 + n):
        e.planes_refresh(slots)
        e.stage(b * Fe, 1, Fe, slots=slots); e.launch()
    e.sync()

def all_run(b0, n):
    th = [threading.Thread(target=run, args=(e, b0, n)) for e in engines]
    for t in th: t.start()
    for t in th: t.join()

all_run(0, 4)
t0 = time.perf_counter()
all_run(4, steps)
t = (time.perf_counter() - t0) / steps
print("new picture on every frame, %d frames over %d engines on one device: %.3f ms per round = %.1f Gsamples/s; host frames %s"
      % (Fe * N, N, t * 1e3, Fe * N * FS / t * 1e-9, [e.secam_stats().get("host_frames") for e in engines]))
for e in engines: e.close()
