#!/usr/bin/env python3
"""tools/ablate_direct.py -- the metric kernel (hvk_k_direct<1,1,1,0,1,1>, 128 frames) with stages switched off, by HIP events: what
each stage's ABSENCE buys. Needs a library built with the switches: make -C hacktv_amd/csrc ABLATE=1 B=/tmp/build_abl OUT=../libhvk_abl.so
and HVK_LIB=hacktv_amd/libhvk_abl.so. The variants' output is NOT the signal's. Run on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ABL_CHILD"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hacktv_amd as H, util
    g = util.Golden()
    F = 128
    with H.Engine(H.preset("i", H.FLAG_FILTER), 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F)
        for _ in range(300):
            e.launch()
        e.sync()
        e.timing_enable(True)
        for _ in range(300):
            e.launch()
        e.sync()
        print("%.4f" % e.timing_read(1)[0])
    sys.exit(0)
CASES = [("as built", 0), ("stores: two contiguous KB per wave", 4096), ("carrier reads: two contiguous KB per wave", 8192), ("both contiguous", 4096 + 8192), ("no NICAM symbol loop", 32), ("no NICAM mixer", 64), ("no NICAM stage at all", 1024), ("no carrier reads", 256),
         ("no carriers, no NICAM", 256 + 1024), ("no filter (matrix unit)", 2048), ("no stores", 512), ("no carriers, no stores", 256 + 512),
         ("no carriers, NICAM, filter", 256 + 1024 + 2048), ("nothing but reads + modulator + LDS planes", 256 + 1024 + 2048 + 512)]
if os.environ.get("ABL_FIRST"):
    CASES = CASES[:int(os.environ["ABL_FIRST"])]
for rnd in range(int(os.environ.get("ABL_ROUNDS", "2"))):
    for name, v in CASES:
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, ABL_CHILD="1", HVK_ABLATE=str(v)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print("%-46s %s ms" % (name, out.stdout.strip() or "FAILED " + out.stderr[-200:]), flush=True)
