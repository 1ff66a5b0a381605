#!/usr/bin/env python3
"""tools/secam_wrong_starts.py -- which lines of bench.py's noisy SECAM pictures start from a wrong estimated state
(HVK_SECAM_DEBUG=1 makes the engine name them): one 64-frame block of the four pictures."""
import os, sys
os.environ["HVK_SECAM_DEBUG"] = "1"
os.environ["HVK_SECAM_NO_CELL_CACHE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
rngs = np.random.default_rng(3)
yy_, xx_ = np.mgrid[0:576, 0:832]
noisy = []
for i_ in range(4):
    p_ = (((xx_ * 255 // 831 + i_ * 17) % 256).astype(np.uint32) << 16) | (((yy_ * 255 // 575) % 256).astype(np.uint32) << 8) | (((xx_ + yy_) // 3 % 256).astype(np.uint32))
    noisy.append(np.where(rngs.random(p_.shape) < 0.2, rngs.integers(0, 1 << 24, p_.shape, dtype=np.uint32), p_).astype(np.uint32))
F = 64
with H.Engine(H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO), 16000000, device=0, max_frames=F) as e:
    for i, p in enumerate(noisy):
        e.frame_upload(i, p)
    slots = [i % 4 for i in range(F)]
    for b in range(3):
        print("block", b, flush=True)
        e.stage(b * F, 1, F, slots=slots); e.launch(); e.sync()
    print(e.secam_stats(), e.secam_walk_stages())
    # the task list: slot -> line
    print("(task slot s is line tasks[s]: slots 0, 1 are the stream's fill slots)")
