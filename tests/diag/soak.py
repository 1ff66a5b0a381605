#!/usr/bin/env python3
"""tests/diag/soak.py [FRAMES] -- long-run parity: the device path against the oracle over many frames
(several FM re-normalisations, thousands of NICAM frames, the colour table wrapping many times), in
batches of uneven size. One-off diagnostic (minutes of oracle time); the suite's own long-run cases
are tests/test_gpu_parity.py::test_late_frames_* and tests/test_oracle_vs_ref.py::test_oracle_long_run_*."""
import hashlib
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H  # noqa: E402
import oracle  # noqa: E402
import util  # noqa: E402

g = util.Golden()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for case, n in (("i_full", N), ("m_full", N // 2), ("l_tt", N // 6), ("i_px135", N // 6), ("g_a2", N // 6)):
    c = g.cases[case]
    conf, sr = g.conf(case)
    pr = c.get("pixel_rate", 0)
    skip = g.teletext_skip(case)
    L = c["lines"]
    batches = [7, 1, 13, 32, 5]
    with oracle.Oracle(conf, sr, pr) as o, H.Engine(conf, sr, device=0, max_frames=32, pixel_rate=pr) as e:
        o.set_frame(g.frame(case))
        o.set_audio(g.audio, True)
        e.frame_upload(0, g.frame(case))
        done, bi, bad = 0, 0, None
        while done < n and bad is None:
            b = min(batches[bi % len(batches)], n - done)
            bi += 1
            while e.audio_needed(b) > 0:
                e.audio_write(g.audio)
            if c.get("teletext"):
                for f in range(b):
                    rows, mask = g.teletext_rows((done + f) % 8, skip)
                    e.teletext_packets(f, rows, mask)
            e.render(b)
            got = e.fetch(0, b * e.info["frame_samples"])
            want = []
            for f in range(b):      # the oracle queues at most 16 frames of packets ahead
                if c.get("teletext"):
                    o.teletext_packets(done + f, *g.teletext_rows((done + f) % 8, skip))
                want.append(o.render_lines(L))
            want = np.concatenate(want)
            if not np.array_equal(got, want):
                d = np.nonzero((got != want).any(axis=1))[0]
                bad = (done + d[0] // e.info["frame_samples"], d[0] % e.info["frame_samples"], len(d))
            done += b
        print("%-8s %4d frames: %s" % (case, done, "EQUAL" if bad is None else "DIFFERENT first at frame %d sample %d (%d samples in that batch)" % bad), flush=True)
