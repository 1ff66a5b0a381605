#!/usr/bin/env python3
"""tools/fuzz_secam_host.py [cases] [seed] -- TEST INFRASTRUCTURE, runs without a GPU.

The engine's SECAM colour chain as the host runs it (hvk_secam.c + hvk_secam_chain.h: the arithmetic the device kernels share,
and what HVK_SECAM_HOST=1 selects) against the oracle, which tools/fuzz_oracle_ref.py and tests/test_oracle_vs_ref.py pin to the
unmodified reference: `-m secam --s-video` -- the Q channel IS the sub-carrier there, so every sample of it is compared -- at
random sample rates (13.5 .. 30 MHz, the ones nothing was tuned for among them), with and without field identification lines
(1 .. 9 of them), --gamma / --level / --invert-video, and pictures that change every frame: noise, flat colours, gradients,
none at all, narrower and shorter than the raster, either field-order flag. Three to five frames a case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H  # noqa: E402
import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
RATES = [13500000, 14000000, 15000000, 16000000, 16384000, 17000000, 17734475, 18000000, 20250000, 21000000, 22500000, 24000000, 25000000, 27000000, 30000000]
rng = np.random.default_rng(SEED)
bad = refused = 0
t0 = time.time()
for case in range(N):
    sr = int(RATES[int(rng.integers(len(RATES)))])
    conf = H.preset("secam", H.FLAG_NOAUDIO)
    conf.s_video = 1
    opts = []
    if rng.random() < 0.6:
        conf.secam_field_id = 1
        if rng.random() < 0.5:
            conf.secam_field_id_lines = int(rng.integers(1, 10))
        opts.append("fid=%d" % conf.secam_field_id_lines)
    if rng.random() < 0.25:
        conf.gamma = float(rng.uniform(0.4, 2.6)); opts.append("gamma=%.3f" % conf.gamma)
    if rng.random() < 0.25:
        conf.level = float(conf.level * rng.uniform(0.3, 1.0)); opts.append("level=%.3f" % conf.level)
    if rng.random() < 0.15:
        conf.invert_video = 1; opts.append("invert")
    nfr = int(rng.integers(3, 6))
    desc = "secam s-video %9d %s" % (sr, " ".join(opts))
    try:
        e = H.Engine(conf, sr, device=-1)
    except H.HvkError:
        refused += 1
        print("refused  ", desc, flush=True)
        continue
    with e, oracle.Oracle(conf, sr) as o:
        w, h, L, W = o.info["active_width"], o.info["active_lines"], o.info["lines"], o.info["width"]
        pics, flags = [], []
        for f in range(nfr):
            kind = int(rng.integers(5))
            if kind == 0: p = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
            elif kind == 1: p = np.full((h, w), int(rng.integers(0, 1 << 24)), np.uint32)
            elif kind == 2:
                yy, xx = np.mgrid[0:h, 0:w]
                p = ((((xx * 255 // max(w - 1, 1) + f * 9) % 256).astype(np.uint32) << 16) | (((yy * 255 // max(h - 1, 1)) % 256).astype(np.uint32) << 8) | ((xx + yy) % 256).astype(np.uint32))
            elif kind == 3: p = None
            else: p = rng.integers(0, 1 << 24, (int(rng.integers(1, h + 1)), int(rng.integers(2, w + 1))), dtype=np.uint32)
            pics.append(None if p is None else np.ascontiguousarray(p))
            flags.append(int(rng.integers(3)))
        kinds = ["none" if p is None else "%dx%d" % (p.shape[1], p.shape[0]) for p in pics]
        wrong = None
        for f in range(nfr):
            o.set_frame(pics[f] if pics[f] is not None else np.zeros((0, 0), np.uint32), flags[f])
            want = o.render_lines(L)[:, 1].reshape(L, W)
            got = e.host_secam_stream(pics[f], flags[f]).reshape(L, W)
            d = np.nonzero((got != want).any(axis=1))[0]
            if d.size:
                wrong = "frame %d: %d lines differ, first line %d" % (f, d.size, d[0] + 1)
                break
    if wrong:
        bad += 1
        print("DIFFERENT", desc, wrong, "pictures", kinds, "flags", flags, flush=True)
    else:
        print("equal    ", desc, kinds, flush=True)
print("%d cases, %d refused, %d different, %.0f s" % (N, refused, bad, time.time() - t0))
sys.exit(1 if bad else 0)
