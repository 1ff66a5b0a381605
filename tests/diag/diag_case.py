#!/usr/bin/env python3
"""tests/diag/diag_case.py CASE [NFRAMES] -- render a golden case on the GPU and with the oracle; list the lines that differ."""
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
case = sys.argv[1]
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = g.cases[case]
conf, sr = g.conf(case)
pr = c.get("pixel_rate", 0)
skip = g.teletext_skip(case)
with oracle.Oracle(conf, sr, pr) as o:
    o.set_frame(g.frame(case))
    o.set_audio(g.audio, True)
    if c.get("teletext"):
        for f in range(nfr + 1):
            o.teletext_packets(f, *g.teletext_rows(f, skip))
    if conf.passthru:
        o.set_passthru(util.passthru_signal())
    want = o.render_lines(nfr * c["lines"])
with H.Engine(conf, sr, device=0, max_frames=nfr, pixel_rate=pr) as e:
    e.frame_upload(0, g.frame(case))
    while e.audio_needed(nfr) > 0:
        e.audio_write(g.audio)
    if c.get("teletext"):
        for f in range(nfr):
            e.teletext_packets(f, *g.teletext_rows(f, skip))
    if conf.passthru:
        e.passthru_write(util.passthru_signal())
    e.render(nfr)
    got = e.fetch(0, nfr * e.info["frame_samples"])
W = len(want) // (nfr * c["lines"])
bad = np.nonzero((got != want).any(axis=1))[0]
print(case, "samples differing:", len(bad))
for ln in sorted(set((bad // W).tolist()))[:40]:
    d = bad[bad // W == ln]
    print("  line %d (frame %d line %d): %d samples, x = %s got %s want %s" % (ln, ln // c["lines"] + 1, ln % c["lines"] + 1, len(d), (d[:4] % W).tolist(), got[d[0]].tolist(), want[d[0]].tolist()))
