#!/usr/bin/env python3
"""tools/fuzz_tail_host.py [cases] [seed] -- TEST INFRASTRUCTURE, runs without a GPU.

The engine's serial tail as the host runs it (hvk_tail.c: the FM video phasor with its 32767-sample renormalisation, then
--swap-iq, the --offset phasor, --passthru) against the oracle (pinned to the unmodified reference by tests/ and
tools/fuzz_oracle_ref.py): the oracle's un-modulated composite of an FM video mode, given to hvk_host_fm_video() in uneven
pieces, has to come out as the oracle's FM output -- random FM modes, sample rates, --filter, offsets, swap, passthru signals
that end inside the run, random pictures and loud sound, a little over a frame. And for the AM / VSB modes the offset
stream alone: out_with_offset == out_without * hvk_host_offset_stream(), sample for sample."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H  # noqa: E402
import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
FM = {"pal-fm": [14000000, 16000000, 18000000, 20250000, 27000000], "secam-fm": [16000000, 18000000, 20250000], "ntsc-fm": [13500000, 14318181, 18000000],
      "apollo-fm": [3200000, 8000000], "apollo-fsc-fm": [13500000]}
AM = {"i": [16000000, 14000000, 20250000], "m": [13500000, 16000000], "g": [16000000, 18000000], "b": [13500000], "l": [16000000], "pal-n": [17734475]}
rng = np.random.default_rng(SEED)
bad = refused = 0
t0 = time.time()


def frames_of(conf, sr, nl, pic, audio, sig):
    with oracle.Oracle(conf, sr) as o:
        o.set_frame(pic)
        o.set_audio(audio, True)
        if sig is not None:
            o.set_passthru(sig)
        return o.render_lines(nl), o.info["width"], o.info["lines"]


for case in range(N):
    fm = rng.random() < 0.6
    table = FM if fm else AM
    mode = list(table)[int(rng.integers(len(table)))]
    sr = int(table[mode][int(rng.integers(len(table[mode])))])
    # (FM video without --filter here: with it the modulator has run over the filter's never-emitted start-up line when the
    # stream's first sample is made, which hvk_host_fm_video() is not given -- the engine primes it from the slab, and the
    # GPU tests and tools/fuzz_parity.py have those)
    flags = (H.FLAG_FILTER if rng.random() < 0.5 and not fm else 0) | (H.FLAG_NOAUDIO if rng.random() < 0.3 else 0)
    conf = H.preset(mode, flags)
    opts = []
    if rng.random() < (0.5 if fm else 1.0):
        conf.offset = int(rng.integers(-8, 9)) * 50000 or 250000; opts.append("offset=%d" % conf.offset)
    if rng.random() < 0.3:
        conf.swap_iq = 1; opts.append("swap")
    use_pass = rng.random() < 0.4
    if use_pass:
        conf.passthru = 1; opts.append("passthru")
    desc = "%-13s %9d flags %d %s" % (mode, sr, flags, " ".join(opts))
    try:
        e = H.Engine(conf, sr, device=-1)
    except H.HvkError:
        refused += 1
        print("refused  ", desc, flush=True)
        continue
    with e:
        w, h, L, W = e.info["active_width"], e.info["active_lines"], e.info["lines"], e.info["width"]
        nl = L + int(rng.integers(20, 200))
        pic = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
        audio = rng.integers(-32768, 32768, (4096 + 37, 2), dtype=np.int64).astype(np.int16)
        sig = rng.integers(-3000, 3000, (int(W * L * rng.uniform(0.3, 1.4)) + 17, 2), dtype=np.int64).astype(np.int16) if use_pass else None
        want, _, _ = frames_of(conf, sr, nl, pic, audio, sig)
        pre_conf = H.preset(mode, flags)
        if fm:
            pre_conf.modulation = 0         # HVK_NONE: the same levels, no modulator
            pre, _, _ = frames_of(pre_conf, sr, nl, pic, audio, None)
            if sig is not None:
                e.passthru_write(sig[:1000]); e.passthru_write(sig[1000:])
            cuts = sorted(set([0, nl * W] + [int(c) * W for c in rng.integers(1, nl, 3)]))
            got = np.concatenate([e.host_fm_video(pre[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        else:
            a, _, _ = frames_of(pre_conf, sr, nl, pic, audio, None)
            if conf.swap_iq:
                a = a[:, ::-1]
            a = a.astype(np.int32)
            cut = int(rng.integers(1, nl)) * W
            b = np.concatenate([e.host_offset_stream(0, cut), e.host_offset_stream(cut, nl * W - cut)]).astype(np.int32)
            got = np.empty_like(a)
            got[:, 0] = (a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1]) >> 15
            got[:, 1] = (a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]) >> 15
            got = got.astype(np.int16)
            if sig is not None:
                # whole lines only, and the process has run over the filter's never-emitted start-up line(s) already
                prime = e.info["delay_lines"] * W
                n = max(min(len(sig) // W * W - prime, nl * W), 0)
                got[:n] = (got[:n].astype(np.int32) + sig[prime:prime + n].astype(np.int32)).astype(np.int16)
    if got.shape != want.shape or not np.array_equal(got, want):
        bad += 1
        d = np.nonzero((got != want).any(axis=1))[0] if got.shape == want.shape else [-1]
        print("DIFFERENT", desc, "first at sample %d (line %d), %d samples" % (d[0], d[0] // W, len(d)), flush=True)
    else:
        print("equal    ", desc, flush=True)
print("%d cases, %d refused, %d different, %.0f s" % (N, refused, bad, time.time() - t0))
sys.exit(1 if bad else 0)
