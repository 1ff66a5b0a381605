import sys, time, hashlib
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hacktv_amd as H, util, ctypes as C
from hacktv_amd.engine import lib
g=util.Golden()
def conf(mode,flags,a2=0):
    c=H.preset(mode,flags); c.a2stereo=a2; return c
for c,sr,name in ((conf("i",H.FLAG_FILTER),16000000,"PAL-I FM+NICAM"),(conf("g",H.FLAG_FILTER,1),16000000,"PAL-G A2"),(conf("m",0,1),13500000,"NTSC-M A2"),(conf("l",H.FLAG_FILTER),16000000,"SECAM-L AM+NICAM")):
    e=H.Engine(c,sr,device=-1)
    FS=e.info["frame_samples"]; n=FS*16
    while e.audio_needed(20)>0: e.audio_write(g.audio)
    car = np.zeros((n, 2), np.int16); car[:]=1
    sym = np.zeros(n // 16 + 64, np.uint8); k0=C.c_int64(0)
    best=0; pos=0
    for rep in range(1):
        t0=time.perf_counter()
        r=lib().hvk_host_side_streams(e.h, pos, n, car.ctypes.data, sym.ctypes.data, len(sym), C.byref(k0)); pos+=n
        t=time.perf_counter()-t0
    print(name, "%.1f Msamples/s"%(n/t/1e6), hashlib.sha256(car.tobytes()).hexdigest()[:16])
    e.close()
