#!/usr/bin/env python3
"""tools/time_direct.py [frames] [launches] [rounds] -- the metric configuration's kernel alone (no gate: for compile-time
experiments whose output is not the signal's): average launch in ms by the engine's own events, once per library named in
TIME_LIBS (comma separated paths, "" = the library as built), alternating. Run on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("TIME_CHILD"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hacktv_amd as H
    import util
    g = util.Golden()
    F = int(sys.argv[1]); N = int(sys.argv[2])
    with H.Engine(H.preset("i", H.FLAG_FILTER), 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F)
        for _ in range(300):
            e.launch()
        e.sync()
        e.timing_enable(True)
        for _ in range(N):
            e.launch()
        e.sync()
        ms, n = e.timing_read(1)
        print("%.4f" % ms)
    sys.exit(0)
F = sys.argv[1] if len(sys.argv) > 1 else "128"
N = sys.argv[2] if len(sys.argv) > 2 else "400"
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
libs = os.environ.get("TIME_LIBS", "").split(",")
for r in range(R):
    for lib in libs:
        env = dict(os.environ, TIME_CHILD="1")
        if lib:
            env["HVK_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, __file__, F, N], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print("%-32s %s ms" % (lib or "(as built)", out.stdout.strip() or ("FAILED " + out.stderr[-300:])), flush=True)
