#!/usr/bin/env python3
"""tools/cfg4_blocks.py [frames] [steps] -- BASELINE config 4 without its sound (-m l -s 16000000 --filter --teletext raw:... --noaudio):
stage + launch of fresh blocks, as bench.py's `4_secam_l_teletext_noaudio_device` section does (gate included); what
tools/profile_round.sh runs under rocprofv3 for the per-kernel table of that section. Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hacktv_amd as H
import util
import bench_sections as S
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = util.Golden()
r = S.case_section(H, g, "l_tt", F, steps, 16, 0, "config 4 --noaudio (raw packets)", stage_every_step=True, teletext=True, noaudio=True)
print("%d frames per block: %.4f ms per step = %.1f Gsamples/s; %s; kernels %s" % (F, r["ms_per_step"], r["Msamples_per_s"] / 1e3, r.get("secam_lines"), r["kernels"]))
