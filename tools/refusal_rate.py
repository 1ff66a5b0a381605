#!/usr/bin/env python3
"""tools/refusal_rate.py [cases] [seed] -- how many of tools/fuzz_parity.py's random configurations hvk_open() refuses, and why: the
same draws (modes, rates, options), engines opened without a device (host tables only: every refusal is decided there), the reasons
counted by the first words of the line libhvk prints. `--pixelrate` here also draws the 4 x f_sc rates (17734475, 14318181) the
GPU fuzzer leaves out. No GPU needed."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("REFUSAL_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np
    import hacktv_amd as H
    N, SEED = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(SEED)
    MODES = ["i", "b", "g", "pal-d", "pal-k", "pal-fm", "pal", "pal-m", "pal-n", "525pal", "m", "ntsc-i", "ntsc-fm", "ntsc", "pal60-i", "pal60", "l", "d", "k", "secam-i", "secam-b",
             "secam-g", "secam-fm", "secam", "e", "819", "a", "ntsc-a", "405-i", "405", "ntsc-405", "240-am", "240", "30-am", "30", "nbtv-am", "nbtv",
             "apollo-fsc-fm", "apollo-fsc", "apollo-fm", "apollo", "m-cbs405", "cbs405"]
    RATES = {625: [16000000, 13500000, 14000000, 18000000, 20250000, 17734475, 27000000], 525: [13500000, 16000000, 14318181, 18000000, 27000000], 819: [24570000, 16380000],
             405: [8100000, 16200000, 12150000], 240: [4800000], 30: [750000], 32: [800000], 320: [3200000, 8000000, 13500000]}
    for case in range(N):
        mode = MODES[int(rng.integers(len(MODES)))]
        base = H.preset(mode, 0)
        lines = int(base.lines)
        rates = [17496000] if mode in ("m-cbs405", "cbs405") else RATES.get(lines, [16000000])
        sr = int(rates[int(rng.integers(len(rates)))])
        flags = 0
        for f, p in ((H.FLAG_FILTER, 0.5), (H.FLAG_NOAUDIO, 0.35), (H.FLAG_NONICAM, 0.2)):
            if rng.random() < p:
                flags |= f
        conf = H.preset(mode, flags)
        opts = []
        def maybe(name, value, p):
            if rng.random() < p:
                setattr(conf, name, value); opts.append("%s=%s" % (name, value)); return True
            return False
        if lines in (625, 525):
            maybe("vits", 1, 0.2); maybe("vitc", 1, 0.2); maybe("acp", 1, 0.15)
            if lines == 625:
                maybe("wss", int(rng.integers(1, 9)), 0.2); maybe("sis", 1, 0.2)
            maybe("interlace", 1, 0.12)
            if mode in ("g", "b", "m") and not (flags & H.FLAG_NOAUDIO):
                maybe("a2stereo", 1, 0.3)
        if mode in ("pal", "ntsc", "secam", "pal60", "525pal"):
            maybe("s_video", 1, 0.3)
        if lines == 625: maybe("teletext", 1, 0.15)
        maybe("passthru", 1, 0.1)
        pr = 0
        if lines in (625, 525) and rng.random() < 0.3:
            cand = [r for r in RATES[lines] if r != sr]
            pr = int(cand[int(rng.integers(len(cand)))])
        desc = "%s %d px %d flags %d %s" % (mode, sr, pr, flags, " ".join(opts))
        try:
            H.Engine(conf, sr, device=-1, pixel_rate=pr).close()
            print("CASE ok " + desc, flush=True)
        except H.HvkError:
            print("CASE refused " + desc, flush=True)
    sys.exit(0)
N = sys.argv[1] if len(sys.argv) > 1 else "2000"
SEED = sys.argv[2] if len(sys.argv) > 2 else "6"
r = subprocess.run([sys.executable, __file__, N, SEED], env=dict(os.environ, REFUSAL_CHILD="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
why = collections.Counter()
last = None
n = ref = 0
for line in r.stdout.splitlines():
    if line.startswith("libhvk:"):
        last = re.sub(r"\d+", "N", line)[:110]
    elif line.startswith("CASE"):
        n += 1
        if line.startswith("CASE refused"):
            ref += 1
            why[last or "(no line)"] += 1
        last = None
print("%d configurations drawn (seed %s), %d refused = 1 in %.1f" % (n, SEED, ref, n / max(ref, 1)))
for k, v in why.most_common():
    print("%5d  %s" % (v, k))
