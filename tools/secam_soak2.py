#!/usr/bin/env python3
"""tools/secam_soak2.py [seed] -- SECAM-L through the one-kernel render with the kept cells and start states at work: 240
frames in batches of 1 .. 7, drawn from 6 picture slots in random order (runs of the same picture, pictures taking
turns), slots that get new pictures or none every now and then -- against the host's serial chain. Run on the GPU box."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H, util
from test_gpu_parity import _secam_noisy
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = util.Golden()
conf = H.preset("l", H.FLAG_FILTER | H.FLAG_NOAUDIO)
rng = np.random.default_rng(seed)
pics = _secam_noisy(10, seed=seed) + [g.frame("l_full"), None]
plan, f = [], 0
while f < 240:
    n = int(rng.integers(1, 8))
    mode = rng.integers(0, 3)
    if mode == 0: slots = [int(rng.integers(0, 6))] * n                      # a picture that stays
    elif mode == 1: slots = [int(s) for s in rng.integers(0, 6, n)]           # any order
    else: a, b = rng.integers(0, 6, 2); slots = [int(a if i & 1 else b) for i in range(n)]
    ups = [(int(rng.integers(0, 6)), int(rng.integers(0, len(pics)))) for _ in range(int(rng.integers(0, 3)) if rng.random() < 0.3 else 0)]
    plan.append((slots, ups)); f += n
def run():
    out = []
    with H.Engine(conf, 16000000, device=0, max_frames=7) as e:
        for s in range(6): e.frame_upload(s, pics[s])
        for slots, ups in plan:
            for s, p in ups: e.frame_upload(s, pics[p])
            e.render(len(slots), slots=slots)
            out.append(e.fetch(0, len(slots) * 640000).copy())
        return np.concatenate(out), e.secam_stats(), e.kernel_names()
os.environ["HVK_SECAM_HOST"] = "1"
want, _, _ = run()
del os.environ["HVK_SECAM_HOST"]
for env in ({}, {"HVK_SECAM_NO_SEEDS": "1"}, {"HVK_DIRECT": "0"}):
    os.environ.update(env)
    got, st, names = run()
    print("seed", seed, env, "equal" if np.array_equal(got, want) else "DIFFERENT", st, names[-1], flush=True)
    for k in env: del os.environ[k]
