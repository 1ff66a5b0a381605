import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hacktv_amd as H, oracle, util
g = util.Golden()
conf, sr = g.conf("l_raster")
rng = np.random.default_rng(5)
fbs = [rng.integers(0, 1 << 24, size=(576, 832), dtype=np.uint32), rng.integers(0, 1 << 24, size=(300, 500), dtype=np.uint32), None, g.frame("l_full")]
with H.Engine(conf, sr, device=0, max_frames=2) as e, oracle.Oracle(conf, sr) as o:
    for i in (0, 2):
        for s in range(2): e.frame_upload(s, fbs[i + s])
        e.render(2, slots=[0, 1])
        got = e.fetch(0, 2 * 640000)[:, 0].astype(np.int64).reshape(2, 625, 1024)
        for s in range(2):
            o.set_frame(fbs[i + s]); want = o.render_lines(625)[:, 0].astype(np.int64).reshape(625, 1024)
            bad = np.nonzero((got[s] != want).any(axis=1))[0]
            print('frame', i + s, 'bad lines', len(bad), (bad[:8] + 1).tolist())
            if len(bad):
                l = bad[0]; xs = np.nonzero(got[s][l] != want[l])[0]
                print('   line', l + 1, 'nx', len(xs), 'x', xs[:8].tolist(), 'got', got[s][l][xs[:4]].tolist(), 'want', want[l][xs[:4]].tolist())
