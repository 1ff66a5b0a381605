#!/usr/bin/env python3
"""tools/ab_direct.py [env assignments ...] -- the metric configuration's timed loop (bench.py without its other sections) once per
environment given, e.g.  python tools/ab_direct.py HVK_DIRECT_V=1 HVK_DIRECT_V=2 : value, ms per step, the kernel's average launch.
Run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            env[k] = v
    steps = env.get("AB_STEPS", "200")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-configs", "--detail-out", "", "--steps", steps] + (["--noaudio"] if env.get("AB_NOAUDIO") else []),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-40s value %.1f Msamples/s  %.4f ms/step  kernel %.4f ms  %s" % (spec, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"]), flush=True)
    except Exception as ex:
        print(spec, "FAILED", ex, r.stderr[-800:], flush=True)
