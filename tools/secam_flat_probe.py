#!/usr/bin/env python3
"""tools/secam_flat_probe.py [colour ...] -- SECAM-L, pictures of ONE colour: wrong starts per batch of 7 frames (the estimate's
systematic error shows per colour and line class). Run on the GPU box."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H
conf = H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO); conf.secam_field_id = 1
def run(pics):
    with H.Engine(conf, 16000000, device=0, max_frames=7) as e:
        per = []
        for b in range(3):
            for s in range(7):
                e.frame_upload(s, pics[b * 7 + s])
            st0 = e.secam_stats()
            e.render(7, slots=list(range(7)))
            e.fetch(0, 16)
            st = e.secam_stats()
            per.append(st["mismatches"] - st0["mismatches"])
        return per
for c in [int(x, 16) for x in sys.argv[1:]] or [0xa00641, 0xaf266a, 0xe5afcd, 0x000000]:
    print("colour %06x x21:" % c, run([np.full((576, 832), c, np.uint32)] * 21), flush=True)
