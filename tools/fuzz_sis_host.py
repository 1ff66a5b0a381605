#!/usr/bin/env python3
"""tools/fuzz_sis_host.py <seed> <cases> -- TEST INFRASTRUCTURE, runs without a GPU.

Sound-in-syncs only: the burst records the engine's host half makes (hvk_audio.c: the NICAM framer behind the hand-over of
32-sample blocks; which symbols each line's burst carries) against the oracle's, on loud random sound -- 14 modes x 12 sample
rates x 6 pixel rates, --filter / --noaudio / --nonicam, A2 stereo, S-Video, --volume; 1400 lines a case, asked for in three
pieces. (What the reference does with sound that differs from block to block is its threads' race: DESIGN.md section 3.)"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, hacktv_amd as H, oracle
rng=np.random.default_rng(int(sys.argv[1]))
MODES=["i","b","g","pal-d","pal-k","pal","pal-n","l","d","k","secam-i","secam-b","secam-g","secam"]
RATES=[16000000,13500000,14000000,18000000,20250000,17734475,15000000,17000000,21000000,22500000,24000000,12000000]
bad=ref=0; N=int(sys.argv[2])
for c in range(N):
    mode=MODES[int(rng.integers(len(MODES)))]; sr=int(RATES[int(rng.integers(len(RATES)))])
    flags=(H.FLAG_FILTER if rng.random()<0.5 else 0)|(H.FLAG_NOAUDIO if rng.random()<0.3 else 0)|(H.FLAG_NONICAM if rng.random()<0.2 else 0)
    pr=0
    if rng.random()<0.35:
        cand=[r for r in (13500000,16000000,18000000,20250000,27000000,14000000) if r!=sr]; pr=int(cand[int(rng.integers(len(cand)))])
    conf=H.preset(mode,flags); conf.sis=1
    if rng.random()<0.3: conf.volume=int(rng.integers(64,700))
    if mode=="secam" and rng.random()<0.5: conf.s_video=1
    if mode in("g","b") and not flags&H.FLAG_NOAUDIO and rng.random()<0.3: conf.a2stereo=1
    desc="%s %d px %d flags %d vol %d sv %d a2 %d"%(mode,sr,pr,flags,conf.volume,conf.s_video,conf.a2stereo)
    try:
        e=H.Engine(conf,sr,device=-1,pixel_rate=pr)
    except H.HvkError:
        ref+=1; continue
    with e, oracle.Oracle(conf,sr,pr) as o:
        audio=rng.integers(-32768,32768,(4096+37,2),dtype=np.int64).astype(np.int16)
        n=1400
        o.set_audio(audio,True); o.set_frame(np.zeros((0,0),np.uint32)); o.render_lines(n)
        want=o.sis_bursts(0,n)
        for _ in range(4): e.audio_write(audio)
        got=np.concatenate([e.host_sis_bursts(0,700),e.host_sis_bursts(700,1),e.host_sis_bursts(701,n-701)])
        if got.shape!=want.shape or not np.array_equal(got,want):
            bad+=1; d=np.nonzero((got!=want).any(axis=1))[0]; print("DIFFERENT",desc,"first line",d[0],got[d[0]],want[d[0]],flush=True)
print(N,"cases",ref,"refused",bad,"different")
