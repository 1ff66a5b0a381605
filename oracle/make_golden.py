#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE (not product code).

Generates the committed fixtures under tests/golden/ from the UNMODIFIED
reference built by oracle/Makefile (oracle/_ref). The reference ships no tests
or golden vectors of its own, so these are outputs of the
reference itself, run in the build container:

  testsrc.npz        the built-in test source (src/av_test.c): the 832x576
                     and 715x480 RGBx test cards and the 6.4 s stereo tone.
  ref_digests.json   for every case: sha256 of the first N frames the
                     reference CLI writes (`hacktv_ref <flags> -o - test`),
                     and sha256 of every table vid_init() builds.
  ref_lines.npz      for every case: a few whole lines of reference output
                     (first line, a VBI line, picture lines, the line pair
                     around a frame boundary) for diagnosable comparisons.

Run from the repository root, after `make -C oracle ref`:
    python oracle/make_golden.py
/root/reference is needed only to BUILD oracle/_ref; this script and the
tests never read it.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refprobe  # noqa: E402
import util  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# id, mode, sample rate, CLI flags, probe flags, real output, frames hashed
CASES = [
    ("pal_bb",        "pal",  16000000, [],                        0,                                      True,  4),
    ("pal_bb_filter", "pal",  16000000, ["--filter"],              refprobe.FLAG_FILTER,                   True,  2),
    ("i_raster",      "i",    16000000, ["--noaudio"],             refprobe.FLAG_NOAUDIO,                  False, 4),
    ("i_vsb",         "i",    16000000, ["--noaudio", "--filter"], refprobe.FLAG_NOAUDIO | refprobe.FLAG_FILTER, False, 4),
    ("i_fm",          "i",    16000000, ["--nonicam"],             refprobe.FLAG_NONICAM,                  False, 2),
    ("i_audio",       "i",    16000000, [],                        0,                                      False, 2),
    ("i_full",        "i",    16000000, ["--filter"],              refprobe.FLAG_FILTER,                   False, 4),
    ("i_mono",        "i",    16000000, ["--nocolour", "--filter"], refprobe.FLAG_NOCOLOUR | refprobe.FLAG_FILTER, False, 2),
    ("g_full",        "g",    16000000, ["--filter"],              refprobe.FLAG_FILTER,                   False, 2),
    ("m_full",        "m",    13500000, ["--filter"],              refprobe.FLAG_FILTER,                   False, 4),
    ("ntsc_bb",       "ntsc", 13500000, [],                        0,                                      True,  2),
    ("i_20m",         "i",    20250000, ["--filter"],              refprobe.FLAG_FILTER,                   False, 2),
    ("secam_bb",      "secam", 16000000, [],                       0,                                      True,  3),
    ("l_raster",      "l",    16000000, ["--noaudio"],             refprobe.FLAG_NOAUDIO,                  False, 2),
    ("l_full",        "l",    16000000, ["--filter"],              refprobe.FLAG_FILTER,                   False, 4),
    # the other 625 / 525-line presets of src/video.c:1956-2008
    ("pald_full",     "pal-d",   16000000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("palm_full",     "pal-m",   13500000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("paln_full",     "pal-n",   16000000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("pal525_bb",     "525pal",  13500000, [],                     0,                                      True,  2),
    ("d_full",        "d",       16000000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("secami_full",   "secam-i", 16000000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("secamb_raster", "secam-b", 16000000, ["--noaudio"],          refprobe.FLAG_NOAUDIO,                  False, 2),
    ("ntsci_full",    "ntsc-i",  13500000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("pal60i_full",   "pal60-i", 13500000, ["--filter"],           refprobe.FLAG_FILTER,                   False, 2),
    ("pal60_bb",      "pal60",   13500000, [],                     0,                                      True,  2),
    # FM video with its pre-emphasis filter (fixed tap tables, src/video.c:2017-2113, picked by :3690-3730)
    ("palfm_f14",     "pal-fm",   14000000, ["--filter"],          refprobe.FLAG_FILTER,                   False, 2),
    ("ntscfm_f18",    "ntsc-fm",  18000000, ["--filter"],          refprobe.FLAG_FILTER,                   False, 2),
    ("secamfm_f2025", "secam-fm", 20250000, ["--filter"],          refprobe.FLAG_FILTER,                   False, 2),
    # ... and the rest of the tail behind it: the pipeline's start-up samples go through swap, offset and passthru too
    ("palfm_f14_tail", "pal-fm",  14000000, ["--filter", "--swap-iq", "--offset", "400000", "--passthru", "@PASS@"], refprobe.FLAG_FILTER, False, 3, {"swap_iq": 1, "offset": 400000, "passthru": 1}),
    # NICAM at the top of the range of sample rates (its pulse is 373 taps long there)
    ("i_27m",         "i",        27000000, ["--filter"],          refprobe.FLAG_FILTER,                   False, 2),
    # teletext from a raw packet file (tests/golden/ttraw.bin: 42-byte records, no wall clock involved)
    ("i_tt",          "i",    16000000, ["--noaudio", "--teletext", "raw:@TTRAW@"], refprobe.FLAG_NOAUDIO,  False, 3),
    ("l_tt",          "l",    16000000, ["--filter", "--teletext", "raw:@TTRAW@"],  refprobe.FLAG_FILTER,   False, 3),
    # the complex tail of the pipeline: swap_iq, frequency offset, passthru (util.passthru_signal written to a
    # scratch file), and FM video -- an 8th element names the hvk_config_t members the flags set
    ("i_offset",      "i",    16000000, ["--filter", "--offset", "1000000"], refprobe.FLAG_FILTER,          False, 3, {"offset": 1000000}),
    ("i_swap_pass",   "i",    16000000, ["--filter", "--swap-iq", "--passthru", "@PASS@"], refprobe.FLAG_FILTER, False, 4, {"swap_iq": 1, "passthru": 1}),
    ("m_offset_pass", "m",    13500000, ["--offset", "-250000", "--passthru", "@PASS@"], 0,                 False, 4, {"offset": -250000, "passthru": 1}),
    ("pal_fm",        "pal-fm", 16000000, [],                      0,                                      False, 3),
    ("ntsc_fm",       "ntsc-fm", 13500000, [],                     0,                                      False, 2),
    ("secam_fm_tail", "secam-fm", 16000000, ["--swap-iq", "--offset", "500000"], 0,                        False, 3, {"swap_iq": 1, "offset": 500000}),
    ("pal_fm_pass",   "pal-fm", 16000000, ["--offset", "300000", "--passthru", "@PASS@"], 0,               False, 4, {"offset": 300000, "passthru": 1}),
    # --pixelrate: raster at the pixel rate, poly-phase resampler to the sample rate (9th element: the pixel rate)
    ("i_px135",       "i",    16000000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER,      False, 3, {}, 13500000),
    ("i_px2025",      "i",    16000000, ["--noaudio", "--pixelrate", "20250000"], refprobe.FLAG_NOAUDIO,    False, 2, {}, 20250000),
    ("l_px2025",      "l",    16000000, ["--filter", "--pixelrate", "20250000"], refprobe.FLAG_FILTER,      False, 2, {}, 20250000),
    ("pal_px16_s14",  "pal",  14000000, ["--pixelrate", "16000000"], 0,                                    True,  2, {}, 16000000),
    ("m_px135_s27",   "m",    27000000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER,      False, 2, {}, 13500000),
    # VBI inserters: insertion test signals, widescreen signalling, time code (with teletext they take lines from it)
    ("i_vbi",         "i",    16000000, ["--filter", "--wss", "16:9", "--vitc", "--vits"], refprobe.FLAG_FILTER, False, 3, {"wss": 0x07, "vitc": 1, "vits": 1}),
    ("i_vbi_tt",      "i",    16000000, ["--noaudio", "--wss", "4:3", "--vitc", "--vits", "--teletext", "raw:@TTRAW@"], refprobe.FLAG_NOAUDIO, False, 2, {"wss": 0x08, "vitc": 1, "vits": 1}),
    ("m_vbi",         "m",    13500000, ["--filter", "--vitc", "--vits"], refprobe.FLAG_FILTER,             False, 3, {"vitc": 1, "vits": 1}),
    ("l_vbi",         "l",    16000000, ["--filter", "--wss", "14:9-window", "--vitc", "--vits"], refprobe.FLAG_FILTER, False, 2, {"wss": 0x0E, "vitc": 1, "vits": 1}),
    ("pal_vbi_px",    "pal",  16000000, ["--vits", "--vitc", "--wss", "16:9-top", "--pixelrate", "13500000"], 0, True, 2, {"wss": 0x04, "vitc": 1, "vits": 1}, 13500000),
    # S-Video: luma on I, the colour sub-carrier on Q (the file sink then writes pairs)
    ("pal_sv",        "pal",  16000000, ["--s-video"],               0,                                     False, 2, {"s_video": 1}),
    ("ntsc_sv_f",     "ntsc", 13500000, ["--s-video", "--filter"],   refprobe.FLAG_FILTER,                  False, 2, {"s_video": 1}),
    ("secam_sv",      "secam", 16000000, ["--s-video", "--filter"],  refprobe.FLAG_FILTER,                  False, 2, {"s_video": 1}),
    # raw baseband input instead of the raster (inserters, filter, sound still apply)
    ("i_rawbb",       "i",    16000000, ["--filter", "--raw-bb-file", "@RAWBB@", "--raw-bb-blanking", "2000", "--raw-bb-white", "21000", "--vits", "--wss", "16:9"],
                      refprobe.FLAG_FILTER, False, 3, {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000, "vits": 1, "wss": 7}),
    ("pal_rawbb",     "pal",  16000000, ["--raw-bb-file", "@RAWBB@"], 0,                                   True,  2, {"raw_bb": 1, "raw_bb_blanking_level": 0, "raw_bb_white_level": 32767}),
    # SECAM field identification lines in the vertical interval
    ("l_fid",         "l",    16000000, ["--filter", "--secam-field-id"], refprobe.FLAG_FILTER,            False, 3, {"secam_field_id": 1}),
    ("secam_fid4",    "secam", 16000000, ["--secam-field-id", "--secam-field-id-lines", "4"], 0,            True,  2, {"secam_field_id": 1, "secam_field_id_lines": 4}),
    # --wss auto: the test source is 4:3 (pixel aspect 12:13 at 832 x 576)
    ("i_wss_auto",    "i",    16000000, ["--noaudio", "--wss", "auto"], refprobe.FLAG_NOAUDIO,             False, 2, {"wss": 0xFF}),
    # anti-copy pulses and the CEA-608 caption line (no captions from the test source: parity-only codes)
    ("i_acp_cc",      "i",    16000000, ["--noaudio", "--acp", "--cc608", "--vits", "--teletext", "raw:@TTRAW@"], refprobe.FLAG_NOAUDIO, False, 3, {"acp": 1, "cc608": 1, "vits": 1}),
    ("m_acp_cc",      "m",    13500000, ["--filter", "--acp", "--cc608", "--vitc"], refprobe.FLAG_FILTER,   False, 2, {"acp": 1, "cc608": 1, "vitc": 1}),
    # Zweikanalton: second FM carrier with pilot instead of NICAM (system M carries L - R)
    ("g_a2",          "g",    16000000, ["--filter", "--a2stereo"], refprobe.FLAG_FILTER,                   False, 3, {"a2stereo": 1}),
    ("m_a2",          "m",    13500000, ["--a2stereo"],             0,                                      False, 2, {"a2stereo": 1}),
    ("pal_px135_s136", "pal", 13600000, ["--filter", "--pixelrate", "13500000"], refprobe.FLAG_FILTER,      True,  3, {}, 13500000),   # lines of 870 / 871 samples
]

TABLES = [
    ("syncs", np.int16), ("yuv", np.int16), ("colour_lookup", np.int16), ("burst_win", np.int16),
    ("chroma_taps", np.int16), ("vfilter_itaps", np.int16), ("vfilter_qtaps", np.int16),
    ("fm_mono_lut", np.int32), ("nicam_taps", np.int16), ("nicam_cc", np.int16),
    ("limiter_shape", np.int16), ("limiter_vtaps", np.int32), ("limiter_ftaps", np.int32),
    ("fm_secam_lut", np.int32), ("fm_secam_bell", np.int16), ("fm_secam_fir", np.int16), ("secam_l_fir", np.int16),
    ("teletext_lut", np.int16), ("fm_video_lut", np.int32), ("resampler_taps", np.int16),
]


def ref_cli(mode, sr, flags, nbytes):
    p = subprocess.Popen([refprobe.BIN_PATH, "-m", mode, "-s", str(sr)] + flags + ["-o", "-", "test"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    out = bytearray()
    while len(out) < nbytes:
        chunk = p.stdout.read(nbytes - len(out))
        if not chunk:
            break
        out += chunk
    p.kill()
    p.wait()
    return bytes(out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    digests = {}
    lines = {}
    src = {}

    ttraw = os.path.join(GOLD, "ttraw.bin")
    passfile = "/tmp/hvk_passthru.bin"
    util.passthru_signal().tofile(passfile)
    rawfile = "/tmp/hvk_rawbb.bin"
    util.rawbb_signal().tofile(rawfile)
    for case in CASES:
        cid, mode, sr, flags, pflags, real, nframes = case[:7]
        extra = case[7] if len(case) > 7 else {}
        pixel_rate = case[8] if len(case) > 8 else 0
        teletext = any("@TTRAW@" in f for f in flags)
        with refprobe.RefProbe(mode, sr, pflags, pixel_rate=pixel_rate, teletext=("raw:" + ttraw) if teletext else None) as r:
            info = dict(r.info)
            key = "frame_%dx%d" % (info["active_width"], info["active_lines"])
            if key not in src:
                src[key] = r.test_frame()
            if "audio" not in src:
                src["audio"] = r.test_audio()
            tabs = {}
            for name, dt in TABLES:
                a = r.table(name, dt)
                tabs[name] = {"len": int(a.size), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}

        W, L = info["width"], info["lines"]
        fs = W * L
        if pixel_rate:
            # the cases are chosen so that a frame and a line resample to whole numbers of samples
            assert (fs * sr) % pixel_rate == 0
            fs = fs * sr // pixel_rate
            W = fs // L     # the nominal line; where lines vary in width the excerpts are just windows of the stream
        bps = 2 if real else 4
        data = ref_cli(mode, sr, [f.replace("@TTRAW@", ttraw).replace("@PASS@", passfile).replace("@RAWBB@", rawfile) for f in flags], nframes * fs * bps)
        assert len(data) == nframes * fs * bps, (cid, len(data))
        per_frame = [hashlib.sha256(data[: (i + 1) * fs * bps]).hexdigest() for i in range(nframes)]

        a = np.frombuffer(data, np.int16)
        a = a.reshape(-1, 1) if real else a.reshape(-1, 2)
        pick = sorted(set([0, 1, 5, 6, 22, 23, 100, 309, 310, 312, 313, 335, 622, 623, L - 1, L, L + 1, L + 6, L + 100]))
        pick = [g for g in pick if g < nframes * L]
        lines[cid + "_idx"] = np.array(pick, np.int32)
        lines[cid] = np.stack([a[g * W:(g + 1) * W] for g in pick])

        digests[cid] = {
            "mode": mode, "sample_rate": sr, "cli_flags": flags, "probe_flags": pflags, "real": real,
            "width": W, "lines": L, "frames": nframes, "teletext": teletext, "extra": extra, "pixel_rate": pixel_rate, "frame_samples": fs,
            "sha256_cumulative": per_frame,   # sha256 of the first 1, 2, ... frames
            "info": info, "tables": tabs,
        }
        print(cid, per_frame[-1][:16], flush=True)

    # the file sink's sample formats (src/rf_file.c): first frame of a complex and of a real mode
    sink = {}
    for cid, mode, sr, flags, bytes_per in (("i_full", "i", 16000000, ["--filter"], 2), ("pal_bb", "pal", 16000000, [], 1)):
        for tname, size in (("uint8", 1), ("int8", 1), ("uint16", 2), ("int16", 2), ("int32", 4), ("float", 4)):
            n = 640000 * size * bytes_per
            data = ref_cli(mode, sr, flags + ["-t", tname], n)
            assert len(data) == n
            sink["%s:%s" % (cid, tname)] = hashlib.sha256(data).hexdigest()
    digests["_sink_formats"] = sink
    print("sink formats", len(sink), flush=True)

    np.savez_compressed(os.path.join(GOLD, "testsrc.npz"), **src)
    np.savez_compressed(os.path.join(GOLD, "ref_lines.npz"), **lines)
    with open(os.path.join(GOLD, "ref_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
