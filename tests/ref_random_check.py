#!/usr/bin/env python3
"""tests/ref_random_check.py SETUP -- TEST INFRASTRUCTURE.

Runs the UNMODIFIED reference in-process (oracle/_ref/libhacktv_ref.so through oracle/ref_probe.c)
on a source of our own -- random pictures that change every frame (every field with --interlace),
saturated colours, full-scale noise and bursts as audio, caption pairs, an anamorphic pixel aspect --
and the oracle on the same input. Prints "EQUAL" or the first difference. One setup per process:
the reference's heap over-read is read out of this process's heap, which has to be
the same at read-out and at render time. Used by tests/test_oracle_vs_ref.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H  # noqa: E402
import oracle  # noqa: E402
import refprobe as R  # noqa: E402

SETUPS = {
    # name: (mode, sample rate, probe flags, hvk flags, conf members, frames)
    "i_loud":      ("i", 16000000, R.FLAG_FILTER, H.FLAG_FILTER, {}, 3),
    "m_loud":      ("m", 13500000, R.FLAG_FILTER, H.FLAG_FILTER, {}, 3),
    "l_moving":    ("l", 16000000, R.FLAG_FILTER, H.FLAG_FILTER, {}, 3),
    "g_a2_loud":   ("g", 16000000, R.FLAG_FILTER | R.FLAG_A2STEREO, H.FLAG_FILTER, {"a2stereo": 1}, 3),
    "i_interlace": ("i", 16000000, R.FLAG_NOAUDIO | R.FLAG_INTERLACE, H.FLAG_NOAUDIO, {"interlace": 1}, 3),
    "l_interlace": ("l", 16000000, R.FLAG_NOAUDIO | R.FLAG_INTERLACE, H.FLAG_NOAUDIO, {"interlace": 1}, 3),
    "m_vbi_cc":    ("m", 13500000, R.FLAG_NOAUDIO | R.FLAG_CC608 | R.FLAG_ACP | R.FLAG_VITS | R.FLAG_VITC, H.FLAG_NOAUDIO,
                    {"cc608": 1, "acp": 1, "vits": 1, "vitc": 1}, 3),
    "i_wss_auto":  ("i", 16000000, R.FLAG_NOAUDIO | R.FLAG_WSS_AUTO, H.FLAG_NOAUDIO, {"wss": 0xFF}, 2),
    # the resampler on moving pictures with loud sound; FM video (a serial modulator over all of it)
    "i_px_moving": ("i", 16000000, R.FLAG_FILTER, H.FLAG_FILTER, {}, 3, 13500000),
    "palfm_loud":  ("pal-fm", 16000000, 0, 0, {}, 2),
    "secamfm_mov": ("secam-fm", 16000000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 2),
    # away from 16 MHz: the inserters' tables, NICAM and the rasters scale with the rate
    "i_vbi_135":   ("i", 13500000, R.FLAG_FILTER | R.FLAG_CC608 | R.FLAG_ACP | R.FLAG_VITS | R.FLAG_VITC | R.FLAG_WSS_AUTO, H.FLAG_FILTER,
                    {"cc608": 1, "acp": 1, "vits": 1, "vitc": 1, "wss": 0xFF}, 2),
    "m_16m":       ("m", 16000000, R.FLAG_FILTER | R.FLAG_VITS | R.FLAG_VITC, H.FLAG_FILTER, {"vits": 1, "vitc": 1}, 2),    # 1017-sample lines
    "l_2025":      ("l", 20250000, R.FLAG_FILTER | R.FLAG_VITS, H.FLAG_FILTER, {"vits": 1}, 2),
    "g_a2_2025":   ("g", 20250000, R.FLAG_FILTER | R.FLAG_A2STEREO, H.FLAG_FILTER, {"a2stereo": 1}, 2),
    # --gamma / --level / --invert-video / --volume (8th element: the probe's overrides, mirrored in the members)
    "i_gamma_lvl": ("i", 16000000, R.FLAG_FILTER, H.FLAG_FILTER, {"gamma": 2.2, "level": 0.7, "volume": 700}, 2, 0, {"gamma": 2.2, "level": 0.7, "volume": 700}),
    "m_invert":    ("m", 13500000, R.FLAG_FILTER, H.FLAG_FILTER, {"invert_video": 1, "volume": 64}, 2, 0, {"invert": 1, "volume": 64}),
    "l_level":     ("l", 16000000, R.FLAG_FILTER, H.FLAG_FILTER, {"level": 0.5, "gamma": 0.45}, 2, 0, {"level": 0.5, "gamma": 0.45}),
    # 44 frames: the anti-copy AGC level starts to move at frame 39; time code minutes stay 0 but seconds tick
    "i_acp_long":  ("i", 16000000, R.FLAG_NOAUDIO | R.FLAG_ACP | R.FLAG_VITC, H.FLAG_NOAUDIO, {"acp": 1, "vitc": 1}, 44),
    # the other rasters with pictures that change: which picture a frame's first and last lines show (the mechanical ones
    # scan vertically: the source's dimensions change places and the reference turns every picture, src/video.c:4883-4885)
    "30_moving":   ("30", 750000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 4),
    "nbtv_moving": ("nbtv", 800000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 4),
    "240_moving":  ("240", 4800000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 3),
    "405_moving":  ("405", 8100000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 3),
    "819_moving":  ("819", 16380000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 3),
    "apollo_mov":  ("apollo-fsc", 13500000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 4),
    "cbs_moving":  ("cbs405", 17496000, R.FLAG_NOAUDIO, H.FLAG_NOAUDIO, {}, 4),
    # frames the source has no picture for (src/av.c:50-53), S-Video's second channel, SECAM's field identification lines
    "secam_sv_blank": ("secam", 14000000, R.FLAG_NONICAM | R.FLAG_SVIDEO | R.FLAG_SECAM_FID, H.FLAG_NONICAM, {"s_video": 1, "secam_field_id": 1}, 3, 0, {"blank": 0b101}),
    "secam_sv_16":    ("secam", 16000000, R.FLAG_NONICAM | R.FLAG_SVIDEO | R.FLAG_SECAM_FID, H.FLAG_NONICAM, {"s_video": 1, "secam_field_id": 1}, 3, 0, {"blank": 0b101}),
    # NOT in the test list: sound-in-syncs with sound that differs from block to block. The reference hands 32-sample blocks from
    # its audio thread to the burst encoder on the main thread without a lock (src/sis.c:217-221, src/video.c:3370-3373); which
    # block a frame encode sees depends on which thread is further into the step. The oracle's reading (hand-overs of earlier
    # steps only) is the reference's through the first seven encodes here (lines 1-124, three runs alike) and differs in about a
    # third of the lines after that; with the test tone, whose blocks are alike, both are the same for good (tests/golden).
    "l_sis_px16_14":  ("l", 14000000, R.FLAG_FILTER | R.FLAG_SIS, H.FLAG_FILTER, {"sis": 1}, 3, 16000000),
    "i_sis_loud":     ("i", 16000000, R.FLAG_FILTER | R.FLAG_SIS, H.FLAG_FILTER, {"sis": 1}, 3),
    # found by tools/fuzz_oracle_ref.py (random configurations, reference against oracle):
    # anti-copy pulses leave SECAM's field identification lines alone -- the colour process marks them first (src/video.c:3135, src/acp.c:108)
    "l_acp_fid":      ("l", 16000000, R.FLAG_FILTER | R.FLAG_NONICAM | R.FLAG_ACP | R.FLAG_SECAM_FID, H.FLAG_FILTER | H.FLAG_NONICAM, {"acp": 1, "secam_field_id": 1}, 2),
    "secamfm_acp_fid_px": ("secam-fm", 14000000, R.FLAG_FILTER | R.FLAG_NOAUDIO | R.FLAG_ACP | R.FLAG_SECAM_FID, H.FLAG_FILTER | H.FLAG_NOAUDIO, {"acp": 1, "secam_field_id": 1}, 2, 16000000),
    # S-Video behind the resampler AND the video filter where the lines are not all of one width: a line ends on what its buffer
    # held before (oracle_video.c; the engine refuses these, hvk_tables.c)
    "ntsc_sv_f_down": ("ntsc", 16000000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 2, 27000000),
    "ntsc_sv_f_up":   ("ntsc", 16000000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 13500000),
    "pal60_sv_f_18":  ("pal60", 16000000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 18000000),
    # (the raster's 1017 samples a line resampled UP to rates whose lines are mostly the SHORTER of two widths: 1716 / 1717, 1144 / 1145 --
    # tools/fuzz_parity.py seed 2718 found the engine's Q channel a sample off there)
    "ntsc_sv_f_16_27": ("ntsc", 27000000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 16000000),
    "ntsc_sv_f_16_18": ("ntsc", 18000000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 16000000),
    # (... and DOWN from them: 858 / 859, the longer line the rare one)
    "ntsc_sv_f_16_135": ("ntsc", 13500000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 16000000),
    "pal60_sv_f_16_135": ("pal60", 13500000, R.FLAG_FILTER | R.FLAG_SVIDEO, H.FLAG_FILTER, {"s_video": 1}, 3, 16000000),
    # field-sequential colour on lines that are read from a raw baseband file: no flag pulse -- the line reader takes the raster's
    # place (src/video.c:2406-2446 against :3043-3063); tools/fuzz_parity.py found the engine drawing one (round 5)
    "apollofsc_rawbb": ("apollo-fsc", 13500000, 0, 0, {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000}, 7, 0, {"rawbb": 500000}),
    "cbs405_rawbb":    ("cbs405", 17496000, 0, 0, {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000}, 7, 0, {"rawbb": 300000}),
    # caption pairs queue as the pictures are read -- two a frame with --interlace, none for a frame without a picture -- and leave one a frame
    "m_cc_ilace":     ("m", 13500000, R.FLAG_NOAUDIO | R.FLAG_CC608 | R.FLAG_INTERLACE, H.FLAG_NOAUDIO, {"cc608": 1, "interlace": 1}, 4, 0, {"blank": 0b100010}),
    # SECAM's first fill slot is processed before the source is read (the full active width), the second with the stream's first
    # picture: a narrow or missing first picture at 13.5 / 14 MHz, where the next line's low pass reads the sub-carrier's overrun
    "secam_sv_narrow":   ("secam", 13500000, R.FLAG_NONICAM | R.FLAG_SVIDEO | R.FLAG_SECAM_FID, H.FLAG_NONICAM, {"s_video": 1, "secam_field_id": 1}, 3, 0, {"pic": [301, 100]}),
    "secam_sv_narrow14": ("secam", 14000000, R.FLAG_NONICAM | R.FLAG_SVIDEO, H.FLAG_NONICAM, {"s_video": 1}, 2, 0, {"pic": [175, 246]}),
    "secam_sv_blank135": ("secam", 13500000, R.FLAG_NONICAM | R.FLAG_SVIDEO | R.FLAG_SECAM_FID, H.FLAG_NONICAM, {"s_video": 1, "secam_field_id": 1}, 3, 0, {"blank": 0b001}),
    "secami_ilace_blank": ("secam-i", 27000000, R.FLAG_INTERLACE | R.FLAG_SECAM_FID, 0, {"interlace": 1, "secam_field_id": 1}, 2, 20250000, {"blank": 0b0101}),
}


def main():
    name = sys.argv[1]
    if name.startswith("@"):
        # a setup of the caller's own (tools/fuzz_oracle_ref.py): @{"name": .., "setup": [mode, rate, probe flags, hvk flags, members, frames, pixel rate, overrides]}
        import json
        spec = json.loads(name[1:])
        name = spec["name"]
        SETUPS[name] = tuple(spec["setup"])
    mode, sr, pflags, hflags, members, nframes = SETUPS[name][:6]
    pixel_rate = SETUPS[name][6] if len(SETUPS[name]) > 6 else 0
    override = dict(SETUPS[name][7]) if len(SETUPS[name]) > 7 else {}
    blank = override.pop("blank", 0)
    pic = override.pop("pic", None)                      # pictures narrower / shorter than the raster's (centred, src/video.c:4888-4897)
    src_ilace = override.pop("src_ilace", 0)             # the pictures' field-order flag (src/video.c:3081-3084)
    par_o = override.pop("par", None)                    # their pixel aspect (WSS auto, src/wss.c)
    rawbb_n = override.pop("rawbb", 0)                   # --raw-bb-file: that many random samples (not a whole number of lines: the rewind falls inside one)
    pass_n = override.pop("passthru", 0)                 # --passthru: that many random I/Q pairs (the source ends inside the run if short)
    flat_audio = override.pop("flat_audio", None)        # every sample alike: what sound-in-syncs reads does not depend on the threads' race then
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31) if False else sum(map(ord, name)))
    conf = H.preset(mode, hflags)
    for k, v in members.items():
        setattr(conf, k, v)

    rawbb = passiq = ttrec = None
    tmp = []
    if override.pop("teletext", 0):
        # --teletext raw:<file> (src/teletext.c:1082, :1188-1201): 42-byte records, one per teletext line that no other inserter
        # holds, enough of them that the file does not wrap
        ttrec = rng.integers(0, 256, (nframes * 32, 42), dtype=np.int64).astype(np.uint8)
        tmp.append("/tmp/hvk_fuzz_tt_%d.bin" % os.getpid())
        ttrec.tofile(tmp[-1])
        override["teletext"] = "raw:" + tmp[-1]
    if rawbb_n:
        rawbb = rng.integers(300, 24000, (rawbb_n,)).astype(np.int16)
        tmp.append("/tmp/hvk_fuzz_rawbb_%d.bin" % os.getpid())
        rawbb.tofile(tmp[-1])
        override["raw_bb"] = tmp[-1]
        override["raw_bb_levels"] = (int(conf.raw_bb_blanking_level), int(conf.raw_bb_white_level))
    if pass_n:
        passiq = rng.integers(-3000, 3000, (pass_n, 2)).astype(np.int16)
        tmp.append("/tmp/hvk_fuzz_pass_%d.bin" % os.getpid())
        passiq.tofile(tmp[-1])
        override["passthru"] = tmp[-1]
    import atexit
    atexit.register(lambda: [os.path.exists(t_) and os.remove(t_) for t_ in tmp])

    with R.RefProbe(mode, sr, pflags, pixel_rate=pixel_rate, **override) as r:
        info = dict(r.info)
        w, h, L = info["active_width"], info["active_lines"], info["lines"]
        fields = 2 if members.get("interlace") else 1
        nsrc = min(nframes * fields + 2, 5)                  # shown in turn, over and over
        turned = (int(conf.frame_orientation) & 3) in (1, 3)
        if turned:
            w, h = h, w                                      # (the source's dimensions: src/hacktv.c:1520-1526)
        if pic:
            w, h = min(w, int(pic[0])), min(h, int(pic[1]))
        frames = rng.integers(0, 1 << 24, (nsrc, h, w), dtype=np.uint32)
        frames[1, : h // 2] = 0xFFFFFF                       # white / saturated primaries: the level clamps
        frames[1, h // 2:, : w // 3] = 0xFF0000
        frames[1, h // 2:, w // 3: 2 * w // 3] = 0x00FF00
        frames[1, h // 2:, 2 * w // 3:] = 0x0000FF
        frames[2] = (np.arange(w, dtype=np.uint32) * 0x010101 % 0xFFFFFF)[None, :]
        audio = rng.integers(-32768, 32768, (4096 + 37, 2), dtype=np.int64).astype(np.int16)
        audio[1000:1400] = 32767                              # a clipped burst into the limiter
        audio[2000:2300, 0] = -32768
        if flat_audio is not None:
            audio[:] = flat_audio
        cc = rng.integers(0, 256, (nsrc, 2), dtype=np.int64).astype(np.uint8)
        cc[1] = 0
        par = (16, 11) if name == "i_wss_auto" else (tuple(par_o) if par_o else (1, 1))
        r.set_source(frames, audio, interlaced=src_ilace, par=par, cc=cc, blank=blank)
        ghost = r.table("chroma_ghost", np.int16)
        # ... and kept what they are: in some heap layouts an object of the reference's own lies there and changes while it
        # runs (found by tools/fuzz_oracle_ref.py: the same case equal in one build of the probe, different in the next)
        r.pin_ghost(ghost)
        # (the same kind of thing at the stream's first samples with sound-in-syncs: the burst encoder's first invocation reads
        # what the heap holds in front of its symbol table -- the allocation's own chunk header, constant, and the last 8 bytes
        # of the chunk before it, which are this process's and not the CLI's)
        sis_heap = r.table("sis_heap", np.int16) if members.get("sis") else None
        ref = r.render_lines(nframes * L)
        ghost_after = r.table("chroma_ghost", np.int16)
    if os.environ.get("REF_CHECK_SHA"):
        import hashlib
        print("REFSHA", hashlib.sha256(ref.tobytes()).hexdigest())       # (tools/fuzz_oracle_ref.py: do two runs of the reference agree?)

    if int(conf.frame_orientation):
        # the oracle (like the engine) is given the picture as the raster shows it
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from make_golden_rasters import oriented
        frames = np.stack([oriented(f_, int(conf.frame_orientation)) for f_ in frames])

    with oracle.Oracle(conf, sr, pixel_rate) as o:
        o.set_ghost(ghost)
        if sis_heap is not None and len(sis_heap) == 8:
            o.set_sis_heap(sis_heap)
        o.set_audio(audio, True)
        if rawbb is not None:
            o.set_rawbb(rawbb)
        if passiq is not None:
            o.set_passthru(passiq)
        if os.environ.get("REF_CHECK_VISIBLE"):
            o.set_sis_visible(int(os.environ["REF_CHECK_VISIBLE"]))
        out = []
        # The oracle rasters ONE line ahead of what it hands out (a line's left sync pulse can begin in the line before it),
        # with the picture set at that moment. Where a frame's first line shows picture (the 30- and 32-line rasters: every
        # line does) the next frame's picture therefore has to be set before the frame's LAST line is asked for -- the
        # reference reads it when it starts the frame's first line (src/video.c:4873-4881), which is the same moment.
        early = mode in ("30", "30-am", "nbtv", "nbtv-am")
        fifo, late = [], []
        tt_used = 0
        if ttrec is not None:
            # the lines the other inserters (and SECAM's field identification) hold: teletext leaves them alone and keeps the
            # record for the next free line (src/teletext.c:1219) -- from the engine's own list, which the shim schedules by
            with H.Engine(conf, sr, device=-1, pixel_rate=pixel_rate) as eh:
                held = set(eh.vbi_lines_held())       # (1-based line numbers)
        for f in range(nframes):
            if ttrec is not None:
                rows = np.zeros((32, 45), np.uint8)
                rows[:, 0:2] = 0x55
                rows[:, 2] = 0x27
                mask = 0
                for row in range(32):
                    line1 = 7 + row if row < 16 else 320 + row - 16
                    if line1 in held:
                        continue
                    rows[row, 3:] = ttrec[tt_used]
                    tt_used += 1
                    mask |= 1 << row
                o.teletext_packets(f, rows, mask)
            # (`blank` counts the source's reads: one per frame, one per field with --interlace)
            none = np.zeros((0, 0), np.uint32)
            o.set_frame(frames[(f * fields) % nsrc] if not (blank >> (f * fields)) & 1 else none, src_ilace)
            if fields == 2:
                o.set_frame2(frames[(f * fields + 1) % nsrc] if not (blank >> (f * fields + 1)) & 1 else none, src_ilace)
            # (line 23 is in the first field: the picture in force there; a frame without a picture has square pixels, src/av.c:21-33)
            o.set_frame_aspect(*(par if not (blank >> (f * fields)) & 1 else (1, 1)))
            # caption pairs queue up as the pictures are read (empty pairs and frames without a picture add none,
            # src/cc608.c:47-75, src/video.c:4900-4903) and leave one per frame on the caption line; a field's second
            # picture is read behind that line
            for k in range(fields):
                c = cc[(f * fields + k) % nsrc]
                if not (blank >> (f * fields + k)) & 1 and (int(c[0]) | int(c[1])) & 0x7F:
                    (fifo if k == 0 else late).append(c)
            if fifo:
                c = fifo.pop(0)
                o.set_cc608(f, int(c[0]), int(c[1]))
            fifo += late
            del late[:]
            out.append(o.render_lines((L - 1 if f == 0 else L) if early else L))
        if early:
            out.append(o.render_lines(1))
        mine = np.concatenate(out)

    W = len(ref) // (nframes * L)          # the output line (differs from info["width"] with the resampler)
    # (the samples the chroma filter reads past its buffer are pinned -- ref_pin_ghost -- so nothing about the comparison
    # depends on the heap any more: every sample counts, the line ends too. Until the pin this script compared everything but
    # the line ends where the read-outs before and after the run differed -- and so overlooked what sits at a line's start.)
    assert np.array_equal(ghost, ghost_after)
    if np.array_equal(ref, mine):
        print("EQUAL")
        return
    d = np.nonzero((ref != mine).any(axis=1))[0]
    if os.environ.get("REF_CHECK_VERBOSE"):
        x = d % W
        print("x range of the differences %d..%d; lines %d; largest difference %d" % (x.min(), x.max(), len(np.unique(d // W)), np.abs(ref[d].astype(int) - mine[d].astype(int)).max()))
    print("DIFFERENT %d samples, first at line %d x %d: ref %s oracle %s" % (len(d), d[0] // W, d[0] % W, ref[d[0]].tolist(), mine[d[0]].tolist()))


if __name__ == "__main__":
    main()
