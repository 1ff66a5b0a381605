#!/usr/bin/env python3
"""tools/fuzz_oracle_ref.py [cases] [seed] [seconds] [jobs] -- TEST INFRASTRUCTURE, runs without a GPU.

Random configurations -- any of the 43 mode ids; a rate the mode takes or one nothing was tuned for; a random set of --filter /
--noaudio / --nonicam / --nocolour / A2 stereo / --pixelrate / S-Video / VITS / VITC / ACP / CC608 / WSS auto or one of its eight
modes / SECAM field identification with 1 .. 9 lines / --interlace / --gamma / --level / --invert-video / --volume / --offset /
--swap-iq / --raw-bb-file / --passthru / --teletext raw: (random files) / sound-in-syncs (not with FM video, and with sound whose
blocks are alike: tests/ref_random_check.py says why); frames the source has no picture for, pictures narrower and shorter
than the raster, either field-order flag, six pixel aspects; two to seven frames (FUZZ_LONG=1: 20 .. 55, hundreds on the
30-line rasters) -- each given to the UNMODIFIED reference in-process (oracle/_ref/libhacktv_ref.so) and to the oracle on the
same random pictures and loud sound, every sample compared (tests/ref_random_check.py '@{json}', one process per case). Before
that the engine's host half is opened on the configuration (device -1): what it refuses is counted; what it takes has its
host tables and -- with sound -- its serial sound pre-pass compared with the oracle's. A case that differs is run four more
times: where the reference's own runs differ from each other it is "undefined", not a failure. The third side of the triangle
tools/fuzz_parity.py draws on the GPU (engine against oracle over the same family of configurations)."""
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H  # noqa: E402
import oracle  # noqa: E402
import refprobe as R  # noqa: E402
import util  # noqa: E402
from test_host_path import HOST_TABLES  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 600
JOBS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
LONG = os.environ.get("FUZZ_LONG", "") != ""      # a few dozen frames per case instead of two to seven
WIDE = os.environ.get("FUZZ_PLAIN", "") == ""    # pictures of other sizes, their field-order flag and pixel aspect (draws more random numbers)

MODES = ["i", "b", "g", "pal-d", "pal-k", "pal-fm", "pal", "pal-m", "pal-n", "525pal", "m", "ntsc-i", "ntsc-fm", "ntsc", "pal60-i", "pal60", "l", "d", "k", "secam-i", "secam-b",
         "secam-g", "secam-fm", "secam", "e", "819", "a", "ntsc-a", "405-i", "405", "ntsc-405", "240-am", "240", "30-am", "30", "nbtv-am", "nbtv",
         "apollo-fsc-fm", "apollo-fsc", "apollo-fm", "apollo", "m-cbs405", "cbs405"]
RATES = {625: [16000000, 13500000, 14000000, 18000000, 20250000, 17734475, 27000000], 525: [13500000, 16000000, 14318181, 18000000, 27000000], 819: [24570000, 16380000],
         405: [8100000, 16200000, 12150000], 240: [4800000], 30: [750000], 32: [800000], 320: [3200000, 8000000, 13500000]}


ODD_RATES = {625: [10000000, 12000000, 15000000, 17000000, 21000000, 22500000, 24000000, 25000000, 30000000, 16384000], 525: [10000000, 12272727, 15000000, 20250000, 21000000, 24545454, 12000000],
             819: [20475000, 27300000, 18000000], 405: [10125000, 6075000, 9000000], 240: [3000000, 6000000], 320: [5000000, 6400000]}
WSS_MODES = [("4:3", 0x08), ("14:9-letterbox", 0x01), ("14:9-top", 0x02), ("16:9-letterbox", 0x0B), ("16:9-top", 0x04), ("16:9+-letterbox", 0x0D), ("14:9-window", 0x0E), ("16:9", 0x07)]


def draw(rng, case):
    mode = MODES[int(rng.integers(len(MODES)))]
    base = H.preset(mode, 0)
    lines = int(base.lines)
    rates = [17496000] if mode in ("m-cbs405", "cbs405") else RATES.get(lines, [16000000])
    sr = int(rates[int(rng.integers(len(rates)))])
    if WIDE and mode not in ("m-cbs405", "cbs405") and lines in ODD_RATES and rng.random() < 0.2:
        sr = int(ODD_RATES[lines][int(rng.integers(len(ODD_RATES[lines])))])     # lines that are not a whole number of samples, rates nothing was tuned for
    pf = hf = 0
    members, over = {}, {}
    for p_, h_, prob in ((R.FLAG_FILTER, H.FLAG_FILTER, 0.5), (R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, 0.35), (R.FLAG_NONICAM, H.FLAG_NONICAM, 0.2)):
        if rng.random() < prob:
            pf |= p_
            hf |= h_

    def maybe(flag, name, value, prob):
        nonlocal pf
        if rng.random() < prob:
            pf |= flag
            members[name] = value
            return True
        return False
    if lines in (625, 525):
        maybe(R.FLAG_VITS, "vits", 1, 0.2)
        maybe(R.FLAG_VITC, "vitc", 1, 0.2)
        maybe(R.FLAG_ACP, "acp", 1, 0.15)
        if lines == 525:
            maybe(R.FLAG_CC608, "cc608", 1, 0.25)
        if lines == 625:
            maybe(R.FLAG_WSS_AUTO, "wss", 0xFF, 0.2)
            # (not with FM video: the modulator carries what the never-emitted start-up lines held along as a phase for ever, and
            # what their bursts hold is the reference's threads' race even with silence as sound -- its runs differ from each
            # other there, two to three digests in five runs at most rates: DESIGN.md section 3)
            if not mode.endswith("-fm") and maybe(R.FLAG_SIS, "sis", 1, 0.15):
                over["flat_audio"] = int(rng.integers(-20000, 20000))
        maybe(R.FLAG_INTERLACE, "interlace", 1, 0.12)
        if mode in ("g", "b", "m") and not (hf & H.FLAG_NOAUDIO):
            maybe(R.FLAG_A2STEREO, "a2stereo", 1, 0.3)
    if mode in ("l", "d", "k", "secam", "secam-fm", "secam-i", "secam-b", "secam-g"):
        maybe(R.FLAG_SECAM_FID, "secam_field_id", 1, 0.5)
    if mode in ("pal", "ntsc", "secam", "pal60", "525pal"):
        maybe(R.FLAG_SVIDEO, "s_video", 1, 0.3)
    if rng.random() < 0.15:
        over["gamma"] = members["gamma"] = round(float(rng.uniform(0.4, 2.6)), 3)
    if rng.random() < 0.15:
        over["level"] = round(float(rng.uniform(0.3, 1.0)), 3)
        members["level"] = float(base.level) * over["level"]
    if rng.random() < 0.1:
        over["invert"] = members["invert_video"] = 1
    if rng.random() < 0.15:
        over["volume"] = members["volume"] = int(rng.integers(64, 700))
    if rng.random() < 0.15:
        over["blank"] = int(rng.integers(1, 8))
    if WIDE:
        # --offset / --swap-iq (the complex tail, src/video.c:4587-4645), an explicit --wss mode, --secam-field-id-lines, --nocolour
        if int(base.output_type) == 0:       # HVK_INT16_COMPLEX
            if rng.random() < 0.2:
                over["offset"] = members["offset"] = (int(rng.integers(-8, 9)) * 50000) or 250000
            if rng.random() < 0.15:
                over["swap_iq"] = members["swap_iq"] = 1
        if lines == 625 and "wss" not in members and rng.random() < 0.15:
            name_, code = WSS_MODES[int(rng.integers(len(WSS_MODES)))]
            over["wss"] = name_
            members["wss"] = code
        if members.get("secam_field_id") and rng.random() < 0.4:
            over["fid_lines"] = members["secam_field_id_lines"] = int(rng.integers(1, 10))
        if rng.random() < 0.08:
            pf |= R.FLAG_NOCOLOUR
            hf |= H.FLAG_NOCOLOUR
        # --raw-bb-file (a little more than a frame of samples, not a whole number of lines) and --passthru (2.4 frames: it ends inside the run)
        if lines in (625, 525) and rng.random() < 0.08:
            members.update({"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000})
            over["rawbb"] = int(sr if False else 0) or None
        if rng.random() < 0.1:
            members["passthru"] = 1
            over["passthru"] = -1
        if lines == 625 and "raw_bb" not in members and rng.random() < 0.12:
            members["teletext"] = 1
            over["teletext"] = 1
        if rng.random() < 0.2:
            over["pic"] = [int(rng.integers(2, 1200)), int(rng.integers(1, 600))]
        if rng.random() < 0.2:
            over["src_ilace"] = int(rng.integers(1, 3))
        if rng.random() < 0.2:
            over["par"] = [[16, 11], [12, 11], [64, 45], [1, 1], [10, 11], [40, 33]][int(rng.integers(6))]
    pr = 0
    if lines in (625, 525) and rng.random() < 0.3:
        cand = [r for r in RATES[lines] if r != sr and (os.environ.get("FUZZ_FSC_PIXELS") or r not in (17734475, 14318181))]       # (FUZZ_FSC_PIXELS=1: 4 x the sub-carrier as a pixel rate as well, taken since round 6)
        pr = int(cand[int(rng.integers(len(cand)))])
    nfr = (2 if lines >= 405 else 4) + (int(rng.integers(0, 4)) if WIDE and rng.random() < 0.25 else 0)     # (PAL's sub-carrier sequence is four frames long)
    if LONG:
        nfr = int(rng.integers(20, 56)) if lines >= 405 else int(rng.integers(100, 400))      # slow counters: time code seconds, the anti-copy level, NICAM's frame count, sub-carrier sequences
    fs_raster = int(round((pr or sr) * float(base.frame_rate.den) / float(base.frame_rate.num)))     # samples per frame at the raster's rate
    fs_out = int(round(sr * float(base.frame_rate.den) / float(base.frame_rate.num)))
    if "rawbb" in over:
        over["rawbb"] = fs_raster + 311
    if "passthru" in over:
        over["passthru"] = int(fs_out * 2.4) + 17
    name = "fz%d_%d" % (SEED, case)
    desc = "%-13s %9d px %9d %s %s" % (mode, sr, pr, " ".join(n for n, b in (("filter", H.FLAG_FILTER), ("noaudio", H.FLAG_NOAUDIO), ("nonicam", H.FLAG_NONICAM)) if hf & b),
                                       " ".join("%s=%s" % kv for kv in list(members.items()) + [(k, v) for k, v in over.items() if k in ("blank", "flat_audio", "pic", "src_ilace", "par", "rawbb", "passthru", "teletext")] + ([("nocolour", 1)] if hf & H.FLAG_NOCOLOUR else [])))
    return name, desc, [mode, sr, pf, hf, members, nfr, pr, over]


def run(item):
    name, desc, setup = item
    # the engine's host half (tables only, device -1): refused configurations are counted, not compared; the tables it
    # builds for the others against the oracle's (what tests/test_host_path.py does for the golden cases)
    try:
        conf = H.preset(setup[0], setup[3])
        for k, v in setup[4].items():
            setattr(conf, k, v)
        with H.Engine(conf, setup[1], device=-1, pixel_rate=setup[6]) as e, oracle.Oracle(conf, setup[1], setup[6]) as o:
            if setup[4].get("teletext"):
                o.teletext_packets(0, np.zeros((32, 45), np.uint8), 0)      # (the oracle builds its teletext symbols when first asked for a row)
            for t in HOST_TABLES:
                if t == "chroma_taps" and setup[0] == "ntsc-a":
                    continue        # (colour without a chroma low pass: the engine's table holds the three taps that stand for "none")
                if not np.array_equal(e.table(t, util.TABLE_DTYPES[t]), o.table(t, util.TABLE_DTYPES[t])):
                    return "TABLES", desc, t
            # ... and the host's serial sound pre-pass (FM / AM phasor chains, limiter, A2's pilot, the 32 kHz tick; hvk_audio.c)
            # on loud random sound against the oracle's per-sample loop, a few hundred lines (tests/test_host_path.py does it
            # for ten golden cases) -- where the audio process sees the raster's own lines
            if not (setup[3] & H.FLAG_NOAUDIO) and not setup[6] and e.info.get("has_carriers", 1):
                arng = np.random.default_rng(len(desc))
                audio = arng.integers(-32768, 32768, (4096 + 37, 2), dtype=np.int64).astype(np.int16)
                audio[1000:1400] = 32767
                W, nl = o.info["width"], 260
                o.set_audio(audio, True)
                o.set_frame(np.zeros((0, 0), np.uint32))
                o.render_lines(nl)
                want = o.last_carrier()
                for _ in range(3):
                    e.audio_write(audio)
                got, _, _ = e.host_side_streams(e.info["delay_lines"] * W, nl * W)
                if got.shape != want.shape or not np.array_equal(got, want):
                    return "CARRIERS", desc, "the host's sound pre-pass differs from the oracle's carriers"
        # ... and the sound-in-syncs burst records (which symbols every line's burst carries: the NICAM framer fed through the
        # hand-over of 32-sample blocks, hvk_audio.c) on LOUD random sound -- the oracle's reading of the hand-over is the
        # engine's, whatever the reference's threads do (tests/test_host_path.py has eight golden cases)
        if setup[4].get("sis") and not setup[4].get("raw_bb") and not setup[4].get("teletext"):
            with H.Engine(conf, setup[1], device=-1, pixel_rate=setup[6]) as e, oracle.Oracle(conf, setup[1], setup[6]) as o:
                arng = np.random.default_rng(len(desc) + 1)
                audio = arng.integers(-32768, 32768, (4096 + 37, 2), dtype=np.int64).astype(np.int16)
                n = 1400
                o.set_audio(audio, True)
                o.set_frame(np.zeros((0, 0), np.uint32))
                o.render_lines(n)
                want = o.sis_bursts(0, n)
                for _ in range(4):
                    e.audio_write(audio)
                got = np.concatenate([e.host_sis_bursts(0, 700), e.host_sis_bursts(700, 1), e.host_sis_bursts(701, n - 701)])
                if got.shape != want.shape or not np.array_equal(got, want):
                    return "SIS", desc, "the host's sound-in-syncs burst records differ from the oracle's"
    except H.HvkError as err:
        return "refused", desc, str(err)[:80]
    def once():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_random_check.py"), "@" + json.dumps({"name": name, "setup": setup})],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000, env=dict(os.environ, REF_CHECK_SHA="1"))
        out = r.stdout.strip().splitlines()
        sha = [l for l in out if l.startswith("REFSHA")]
        return r, (out[-1] if out else ""), (sha[-1] if sha else "")
    r, last, sha = once()
    if last.startswith("EQUAL"):
        return "equal", desc, last if last != "EQUAL" else ""
    if r.returncode != 0:
        err = (r.stderr.strip().splitlines() or ["?"])[-1]
        return ("refused" if "HvkError" in err or "ref_open failed" in err else "ERROR"), desc, err[:200]
    # different: is the reference equal to itself? (FM video carries whatever its never-emitted start-up lines held along for
    # ever, and some combinations leave that to the threads' race: tests/golden/ref_undefined.json)
    shas = {sha}
    for _ in range(4):
        r2, last2, sha2 = once()
        shas.add(sha2)
        if last2.startswith("EQUAL") or len(shas) > 1:
            return "undefined", desc, "the reference's runs differ from each other (%s)" % ("one of them is the oracle's" if last2.startswith("EQUAL") else "%d digests" % len(shas))
    return "DIFFERENT", desc, last + "\n          rerun: python tests/ref_random_check.py '@%s'" % json.dumps({"name": name, "setup": setup})


def main():
    if not R.available():
        sys.exit("oracle/_ref/libhacktv_ref.so has not been built (make -C oracle ref)")
    rng = np.random.default_rng(SEED)
    items = [draw(rng, c) for c in range(N)]
    t0 = time.time()
    counts = {}
    with ThreadPoolExecutor(JOBS) as ex:
        for verdict, desc, note in ex.map(run, items):
            counts[verdict] = counts.get(verdict, 0) + 1
            print("%-9s %s %s" % (verdict, desc, note), flush=True)
            if time.time() - t0 > LIMIT:
                break
    print(" ".join("%d %s" % (v, k) for k, v in sorted(counts.items())), "%.0f s" % (time.time() - t0))
    sys.exit(1 if counts.get("DIFFERENT") or counts.get("ERROR") or counts.get("TABLES") or counts.get("CARRIERS") or counts.get("SIS") else 0)


if __name__ == "__main__":
    main()
