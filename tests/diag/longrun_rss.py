#!/usr/bin/env python3
"""tests/diag/longrun_rss.py [SECONDS_OF_SIGNAL] -- run the drop-in binary for minutes of signal and watch its
resident set: queues (audio, passthru, NICAM symbols) must not grow with the length of the run."""
import os
import subprocess
import sys
import time
import psutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for flags in (["-m", "i", "-s", "16000000", "--filter"], ["-m", "l", "-s", "16000000", "--filter", "--vits", "--vitc"]):
    n = S * 16000000 * 4 if flags[1] == "i" else S * 16000000 * 4 // 4
    p = subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", "hacktv_hvk")] + flags + ["-o", "-", "test"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, HVK_BATCH="16"))
    ps = psutil.Process(p.pid)
    got, t0, rss = 0, time.time(), []
    while got < n:
        chunk = p.stdout.read(min(1 << 24, n - got))
        if not chunk:
            break
        got += len(chunk)
        if len(rss) < got // (n // 8 + 1) + 1:
            rss.append(ps.memory_info().rss >> 20)
    dt = time.time() - t0
    p.kill()
    p.wait()
    print("%-50s %5.1f s of signal in %5.1f s; RSS MiB at eighths of the run: %s" % (" ".join(flags), got / 64e6, dt, rss), flush=True)
