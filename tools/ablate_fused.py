#!/usr/bin/env python3
"""tools/ablate_fused.py [frames] -- time of the fused kernel with stages switched off (WRONG output; profiling
only). Needs the library with the switches compiled in:
    make -C hacktv_amd/csrc ABLATE=1 OUT=../libhvk_ablate.so B=build_ablate
Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HVK_FUSE"] = "1"
os.environ.setdefault("HVK_LIB", os.path.join(ROOT, "hacktv_amd", "libhvk_ablate.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SW = [("everything", 0), ("no barriers", 256), ("no raster compute", 512), ("no matrix-unit filter", 1024), ("no NICAM", 2048),
      ("no pixel gathers / staging", 4096), ("no carrier loads", 8192), ("no colour-table read", 4), ("no chroma FIR", 2),
      ("no compute, no NICAM", 512 | 2048), ("no compute, NICAM, filter", 512 | 2048 | 1024),
      ("none of the stages", 512 | 2048 | 1024 | 4096 | 8192 | 4), ("none of the stages, no barriers", 512 | 2048 | 1024 | 4096 | 8192 | 4 | 256)]
for name, bits in SW:
    os.environ["HVK_ABLATE"] = str(bits)
    conf = H.preset("i", H.FLAG_FILTER)
    with H.Engine(conf, 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F)
        for _ in range(2):
            e.launch()
        e.sync()
        e.timing_enable(True)
        for _ in range(10):
            e.launch()
        f, _ = e.timing_read(1)
        print("%-36s %.4f ms -> %.1f Gsamples/s" % (name, f, F * e.info["frame_samples"] / f / 1e6), flush=True)
