#!/usr/bin/env python3
"""tools/secam_soak3.py -- the SECAM colour chain with new pictures on every frame (entry states by estimate, cells from the
(U, V) plane) against the host's serial chain, over modes, sample rates, picture kinds and block sizes that the parity tests
do not all visit: every sample compared, the counters of wrong starts printed. Run on the GPU box."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H

def pictures(kind, n, w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        if kind == "noise":
            p = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
        elif kind == "bars":
            b = (xx * 8 // w + i) % 8
            p = (np.where(b & 4, 0xFF0000, 0) | np.where(b & 2, 0xFF00, 0) | np.where(b & 1, 0xFF, 0)).astype(np.uint32)
        elif kind == "flat":
            p = np.full((h, w), int(rng.integers(0, 1 << 24)), np.uint32)
        else:   # gradients that move, a tenth of the pixels noise
            p = ((((xx * 255 // (w - 1) + i * 9) % 256).astype(np.uint32) << 16) | (((yy * 255 // (h - 1) + i * 5) % 256).astype(np.uint32) << 8)
                 | (((xx + yy + i * 29) // 3) % 256).astype(np.uint32))
            p = np.where(rng.random(p.shape) < 0.1, rng.integers(0, 1 << 24, p.shape, dtype=np.uint32), p).astype(np.uint32)
        out.append(np.ascontiguousarray(p))
    return out

ONLY = os.environ.get("SOAK_ONLY")      # e.g. "secam:bars,l:flat"
CASES = [("l", 16000000, H.FLAG_FILTER, 1), ("l", 20250000, H.FLAG_FILTER, 0), ("secam", 16000000, 0, 1), ("secam-fm", 16000000, 0, 0), ("d", 18000000, H.FLAG_FILTER, 1),
         ("secam-i", 14000000, 0, 0), ("l", 17734475, 0, 1), ("secam-b", 27000000, H.FLAG_FILTER, 0)]
bad = 0
for mode, sr, flags, fid in CASES:
    conf = H.preset(mode, flags | H.FLAG_NOAUDIO)
    conf.secam_field_id = fid
    for kind in ("noise", "moving", "bars", "flat"):
        if ONLY and ("%s:%s" % (mode, kind)) not in ONLY.split(","):
            continue
        for B, NB in ((7, 3), (32, 2)):
            def run():
                out = []
                try:
                    e = H.Engine(conf, sr, device=0, max_frames=B)
                except H.HvkError as err:
                    return None, str(err), 0
                with e:
                    w, h = e.info["active_width"], e.info["active_lines"]
                    pics = pictures(kind, B * NB, w, h, 7)
                    fs = e.info["frame_samples"]
                    for b in range(NB):
                        for s in range(B):
                            e.frame_upload(s, pics[b * B + s] if (b * B + s) % 5 else None)     # (every fifth frame: no picture)
                        e.render(B, slots=list(range(B)))
                        out.append(e.fetch(0, B * fs).copy())
                    return np.concatenate(out), e.secam_stats(), e.secam_estimated_stages()
            os.environ["HVK_SECAM_HOST"] = "1"
            want, st_h, _ = run()
            del os.environ["HVK_SECAM_HOST"]
            if want is None:
                print(mode, sr, "not opened:", st_h); break
            got, st, est = run()
            ok = np.array_equal(got, want)
            bad += not ok
            print("%-9s %9d %-6s B=%2d: %s, estimate stages %d, %s" % (mode, sr, kind, B, "equal" if ok else "DIFFERENT", est, st), flush=True)
        else:
            continue
        break
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
