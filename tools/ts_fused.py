#!/usr/bin/env python3
"""tools/ts_fused.py [frames] [extra ablate bits] -- per-phase cycle counts of hvk_k_fusedw's workgroups
(profiling build, HVK_ABLATE bit 16384: WRONG output). Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HVK_FUSE"] = "1"
os.environ.setdefault("HVK_LIB", os.path.join(ROOT, "hacktv_amd", "libhvk_ablate.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
runs = int(os.environ.get("HVK_RUNS", "8"))
os.environ["HVK_RUNS"] = str(runs)
os.environ["HVK_ABLATE"] = str(16384 | extra)
conf = H.preset("i", H.FLAG_FILTER)
with H.Engine(conf, 16000000, device=0, max_frames=F) as e:
    e.frame_upload(0, g.frame("i_full"))
    while e.audio_needed(F) > 0:
        e.audio_write(g.audio)
    e.stage(0, 1, F)
    for _ in range(3):
        e.launch()
    e.sync()
    fs = e.info["frame_samples"]
    out = e.fetch(0, F * fs).reshape(F, 625, 1024, 2)
rec = []
for y in range(F):
    for x in range(runs):
        l0 = x * 625 // runs
        a = out[y, l0].reshape(-1).view(np.int64)[:16].reshape(2, 8)
        rec.append(a)
rec = np.array(rec)            # [wg][role][8]
t0 = rec[:, :, 0].min()
dur = rec[:, :, 1] - rec[:, :, 0]
print("workgroups %d, kernel span %d cycles (s_memtime units)" % (len(rec), rec[:, :, 1].max() - t0))
for role, name in enumerate(("raster", "filter")):
    r = rec[:, role]
    n = r[:, 7].mean()
    print("%s waves: duration %.0f (min %d max %d), per iteration: phaseA %.0f  wait1 %.0f  phaseB %.0f  wait2 %.0f  (iterations %.1f)" %
          (name, dur[:, role].mean(), dur[:, role].min(), dur[:, role].max(), (r[:, 2] / r[:, 7]).mean(), (r[:, 3] / r[:, 7]).mean(),
           (r[:, 4] / r[:, 7]).mean(), (r[:, 5] / r[:, 7]).mean(), n))
start = rec[:, 0, 0] - t0
print("start times: 10%% %d  50%% %d  90%% %d  max %d" % tuple(np.percentile(start, [10, 50, 90, 100])))
hw = rec[:, 0, 6]
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | ((hw >> 32) << 8)     # cu_id, se_id, xcc
uniq, cnt = np.unique(cu, return_counts=True)
print("distinct (xcc, se, cu): %d, workgroups per CU min %d max %d" % (len(uniq), cnt.min(), cnt.max()))
