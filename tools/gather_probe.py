#!/usr/bin/env python3
"""tools/gather_probe.py dev0,dev1,... -- a small hvk_group_* round on the devices named (PAL-I --filter --noaudio, 2 frames a block)
gathered on the first of them with the backend HVK_GATHER asks for (default: RCCL between distinct devices), compared with every
engine's own read-back; prints the backend that ran. bench.py runs it in a process of its own, with a time limit, before it takes the
same path in-process at N > 1: a collective that has never met the machine it runs on must not be able to hang the measurement."""
import ctypes as C_, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hacktv_amd as H
import util
devices = [int(x) for x in sys.argv[1].split(",")]
g = util.Golden()
F, N = 2, len(devices)
conf = H.preset("i", H.FLAG_FILTER | H.FLAG_NOAUDIO)
with H.Group(conf, 16000000, devices, F) as grp:
    fs = grp.info["frame_samples"]
    hip = C_.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C_.POINTER(C_.c_void_p), C_.c_size_t]
    hip.hipMemcpy.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_size_t, C_.c_int]
    hip.hipSetDevice.argtypes = [C_.c_int]
    hip.hipSetDevice(devices[0])
    root = C_.c_void_p()
    assert hip.hipMalloc(C_.byref(root), N * F * fs * 4) == 0
    for e in grp.engines:
        e.frame_upload(0, g.frame("i_full"))
    for b in range(N):
        grp.stage(F, slots=[0] * F)
        grp.launch()
    grp.gather(0, root, F * fs)
    for e in grp.engines:
        e.sync()
    host = np.zeros((N * F * fs, 2), np.int16)
    hip.hipSetDevice(devices[0])
    assert hip.hipMemcpy(host.ctypes.data, root, N * F * fs * 4, 2) == 0
    for i, e in enumerate(grp.engines):
        own = e.fetch(0, F * fs)
        if not np.array_equal(own, host[i * F * fs:(i + 1) * F * fs]):
            print("MISMATCH engine %d" % i)
            sys.exit(1)
    print("BACKEND " + grp.gather_backend())
