#!/usr/bin/env python3
"""tests/diag/ghost_sweep.py -- TEST INFRASTRUCTURE. The oracle with its default model of the reference's heap
over-read against the reference CLI over a range of sample rates: where the freed
filter-design chunk is handed to the burst window the model is exact; elsewhere the over-read picks
up allocator pointers and the reference CLI's own output changes from run to run (ASLR), e.g.
`hacktv -m pal -s 15000000`: nothing to be exact to."""
import numpy as np, subprocess, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util, oracle, hacktv_amd as H, refprobe
def check(mode, sr):
    conf = H.preset(mode, 0)
    with refprobe.RefProbe(mode, sr, 0) as rp: fr = rp.test_frame()
    with oracle.Oracle(conf, sr) as o:
        W = o.info['width']; L = o.info['lines']
        o.set_frame(fr)
        iq = o.render_lines(L)[:,0]
    out = subprocess.run(["bash","-c","oracle/_ref/hacktv_ref -m %s -s %d -o - test 2>/dev/null | head -c %d" % (mode, sr, len(iq)*2)],capture_output=True).stdout
    ref = np.frombuffer(out,np.int16)
    d = np.nonzero(iq != ref)[0]
    print(mode, sr, 'W', W, 'taps', o.info['chroma_ataps'], 'mismatches', len(d), sorted(set((d % W).tolist()))[:8])
for mode, rates in (("pal", (12000000, 13500000, 14000000, 15000000, 16000000, 17734475, 18000000, 20250000, 27000000)),
                    ("ntsc", (12272727, 13500000, 14318181, 16000000, 18000000, 20250000, 27000000))):
    for sr in rates: check(mode, sr)
