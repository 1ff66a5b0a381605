#!/usr/bin/env python3
"""tools/secam_soak.py -- 128 noisy moving pictures through SECAM-L in batches of 32: the device's colour chain (default
warm-up, a short one, four lines per lane) against the host's serial chain, with the counters of wrong starts and redone
lines. Run on the GPU box."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H
from test_gpu_parity import _secam_noisy
conf = H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO)
conf.secam_field_id = 1
B, NB = 32, 4
pics = _secam_noisy(B * NB, seed=3)
def run():
    out = []
    with H.Engine(conf, 16000000, device=0, max_frames=B) as e:
        for b in range(NB):
            for s in range(B):
                e.frame_upload(s, pics[b * B + s])
            e.render(B, slots=list(range(B)))
            out.append(e.fetch(0, B * 640000).copy())
        return np.concatenate(out), e.secam_stats()
os.environ["HVK_SECAM_HOST"] = "1"
want, _ = run()
del os.environ["HVK_SECAM_HOST"]
for env in ({}, {"HVK_SECAM_WARMUP": "8"}, {"HVK_SECAM_RUN": "4"}):
    os.environ.update(env)
    got, st = run()
    print(env, "equal" if np.array_equal(got, want) else "DIFFERENT", st)
    for k in env: del os.environ[k]
