#!/usr/bin/env python3
"""tools/prep_speed.py [frames] -- what a new picture on every frame costs on the device (PAL-I --filter --noaudio):
hvk_k_prep alone (hvk_planes_refresh over all slots), prep + render, a lone picture's prep; for pictures of few colours
(levels from the table) and noisy ones (levels computed), and -- HVK_PATHS="0 1" -- both prep kernels in processes of
their own (HVK_PREP is read once). Run on the GPU box."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(F):
    import numpy as np
    import hacktv_amd as H
    import util
    g = util.Golden()
    rng = np.random.default_rng(1)
    h, w = 576, 832
    yy, xx = np.mgrid[0:h, 0:w]

    def natural(i):
        r = (xx * 255 // w + 3 * i) % 256
        gch = (yy * 255 // h + 5 * i) % 256
        b = ((xx + yy) * 255 // (w + h) + 7 * i) % 256
        n = rng.integers(-3, 4, size=(3, h, w))
        r, gch, b = [np.clip(c + d, 0, 255).astype(np.uint32) for c, d in zip((r, gch, b), n)]
        return (r << 16) | (gch << 8) | b

    def bars(i):
        return np.roll(g.frame("i_full"), 13 * i, axis=1)

    for mode, flags in (("i", H.FLAG_FILTER | H.FLAG_NOAUDIO), ("m", H.FLAG_FILTER | H.FLAG_NOAUDIO)):
        sr = 16000000 if mode == "i" else 13500000
        for name, make, lv in (("few colours, table", bars, 1), ("noisy, computed", natural, 2), ("noisy, table", natural, 1)):
            res = []
            for chunk in [int(c) for c in os.environ.get("HVK_CHUNKS", "0 32 16 8 4").split()]:
                os.environ["HVK_PREP_CHUNK"] = str(chunk)
                with H.Engine(H.preset(mode, flags), sr, device=0, max_frames=F) as e:
                    e.set_levels(lv)
                    hh, ww = e.info["active_lines"], e.info["active_width"]
                    slots = list(range(F))
                    for s in slots:
                        e.frame_upload(s, np.ascontiguousarray(make(s)[:hh, :ww]))
                    fs = e.info["frame_samples"]
                    e.stage(0, 1, F, slots=slots); e.launch(); e.sync()
                    nxt = F
                    for what in ("prep+render", "render") if chunk == 0 else ("prep+render",):
                        n = 10
                        t0 = time.perf_counter()
                        for k in range(n):
                            if what != "render":
                                e.planes_refresh(slots)
                            e.stage(nxt, 1, F, slots=slots); e.launch()
                            nxt += F
                        e.sync()
                        dt = (time.perf_counter() - t0) / n
                        res.append("%s%s %.2f us/frame (%.0f Gsamples/s)" % (what, " in chunks of %d" % chunk if chunk else " (one chunk)", dt / F * 1e6, F * fs / dt / 1e9))
                    if chunk == 0:
                        n = 200
                        t0 = time.perf_counter()
                        for k in range(n):
                            e.planes_refresh([0])
                            e.stage(nxt, 1, 1, slots=[0]); e.launch()
                            nxt += 1
                        e.sync()
                        res.append("one new picture, staged and rendered alone: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
            print("HVK_PREP=%s %s %-20s F=%d: %s" % (os.environ.get("HVK_PREP", "-"), mode, name, F, "; ".join(res)), flush=True)


if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    if os.environ.get("HVK_PREP_CHILD"):
        one(F)
    else:
        for pth in os.environ.get("HVK_PATHS", "0 1").split():
            env = dict(os.environ, HVK_PREP=pth, HVK_PREP_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), str(F)], env=env)
