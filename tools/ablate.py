#!/usr/bin/env python3
"""tools/ablate.py -- per-kernel time of the two kernels for configurations that
switch stages off (HIP events on the launch stream). Run on the GPU box. The stage switches need a
library built with them: `make -C hacktv_amd/csrc clean all ABLATE=1` (the `modes` table does not)."""
import os, sys
os.environ.setdefault("HVK_DIRECT", "0")     # the stage switches live in the raster + filter kernel pair (the default path since round 3 is hvk_k_direct)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cases = [("full: filter+fm+nicam", H.FLAG_FILTER), ("filter only (noaudio)", H.FLAG_FILTER | H.FLAG_NOAUDIO),
         ("filter+fm (nonicam)", H.FLAG_FILTER | H.FLAG_NONICAM), ("audio only (no filter)", 0),
         ("raster only", H.FLAG_NOAUDIO), ("mono + filter, noaudio", H.FLAG_FILTER | H.FLAG_NOAUDIO | H.FLAG_NOCOLOUR)]
abl = [("", 0)]
if len(sys.argv) > 2 and sys.argv[2] == "modes":
    abl = []
elif len(sys.argv) > 2:
    abl = [("filter: no pulse-table staging", 16), ("filter: no shaping loop", 32), ("filter: no mixer", 64), ("filter: no staging+loop+mixer", 112)] if sys.argv[2] == "filter" else [("raster: no level-table gather", 1), ("raster: no chroma FIR", 2), ("raster: no colour-table read", 4), ("raster: no picture phase at all", 8), ("raster: none of the four", 15), ("raster: none of the four, no store", 143), ("raster: no store", 128)]
    cases = cases[:1]
for aname, aval in abl:
  os.environ["HVK_ABLATE"] = str(aval)
  for name, flags in cases:
    name = aname or name
    conf = H.preset("i", flags)
    with H.Engine(conf, 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F)
        for _ in range(2):
            e.launch()
        e.sync()
        e.timing_enable(True)
        for _ in range(10):
            e.launch()
        r, _ = e.timing_read(0)
        f, _ = e.timing_read(1)
        fs = e.info["frame_samples"]
        print("%-34s raster %.4f ms  filter %.4f ms  -> %.1f Gsamples/s" % (name, r, f, F * fs / (r + f) / 1e6), flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "modes":
    # device-only rates of the other BASELINE configurations (parity-test cases, not bench lines)
    os.environ["HVK_ABLATE"] = "0"
    for case in ("pal_bb", "i_full", "m_full", "l_full", "l_tt", "i_px135", "i_vbi"):
        c = g.cases[case]
        conf, sr = g.conf(case)
        pr = c.get("pixel_rate", 0)
        skip = g.teletext_skip(case)
        Fm = min(F, 32) if conf.colour_mode == 3 else F      # SECAM: the host colour pre-pass is slow
        with H.Engine(conf, sr, device=0, max_frames=Fm, pixel_rate=pr) as e:
            e.frame_upload(0, g.frame(case))
            while e.audio_needed(Fm) > 0:
                e.audio_write(g.audio)
            if c.get("teletext"):
                for f in range(Fm):
                    e.teletext_packets(f, *g.teletext_rows(f % 4, skip))
            e.stage(0, 1, Fm)
            for _ in range(2):
                e.launch()
            e.sync()
            e.timing_enable(True)
            for _ in range(10):
                e.launch()
            r, _ = e.timing_read(0)
            f, _ = e.timing_read(1)
            fs = e.info["frame_samples"]
            print("%-10s %-44s frames %3d  raster(+resample) %.4f ms  filter %.4f ms  -> %.1f Gsamples/s" %
                  (case, "-m %s -s %d %s" % (c["mode"], sr, " ".join(c["cli_flags"][:4])), Fm, r, f, Fm * fs / (r + f) / 1e6), flush=True)
