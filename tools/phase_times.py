#!/usr/bin/env python3
"""tools/phase_times.py [frames] [launches] -- where a workgroup of hvk_k_direct spends its time: needs the measuring build
(make -C hacktv_amd/csrc PHASES=1 B=build_phases OUT=../libhvk_phases.so; HVK_LIB=hacktv_amd/libhvk_phases.so). The metric
configuration (-m i --filter, FM + NICAM) and the same without sound; average shader-clock cycles per workgroup between the
kernel's marks (its first lane's clock: what the slowest of its eight waves makes it wait for shows at the barriers)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H
import util

g = util.Golden()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = ctypes.CDLL(H.LIB_PATH)
if not hasattr(L, "hvk_phase_times"):
    raise SystemExit("%s is not the measuring build (make PHASES=1)" % H.LIB_PATH)
names = ["tile's lines (scalar)", "reads + modulator + byte planes", "barrier 1", "filter (matrix unit)", "barrier 2",
         "outputs from LDS + carriers", "NICAM", "stores issued"]
for label, flags in (("-m i --filter (FM + NICAM)", H.FLAG_FILTER), ("-m i --filter --noaudio", H.FLAG_FILTER | H.FLAG_NOAUDIO)):
    with H.Engine(H.preset("i", flags), 16000000, device=0, max_frames=F) as e:
        e.frame_upload(0, g.frame("i_full"))
        while e.audio_needed(F) > 0:
            e.audio_write(g.audio)
        e.stage(0, 1, F)
        for _ in range(5):
            e.launch()
        e.sync()
        acc = (ctypes.c_ulonglong * 16)()
        L.hvk_phase_times(acc, 1)
        e.timing_enable(True)
        for _ in range(N):
            e.launch()
        e.sync()
        ms, n = e.timing_read(1)
        L.hvk_phase_times(acc, 1)
        wgs = acc[15]
        tot = sum(acc[i] for i in range(8))
        print("%s: %.4f ms per launch (measuring build), %d workgroups of the last launch, %.0f cycles per workgroup" % (label, ms, wgs, tot / wgs))
        for i, nm in enumerate(names):
            print("   %-34s %8.0f cycles  %5.1f %%" % (nm, acc[i] / wgs, 100.0 * acc[i] / tot))
