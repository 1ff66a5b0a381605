"""The oracle (oracle/liboracle.so, the CPU restatement) against the committed
outputs of the unmodified reference: tests/golden/ref_digests.json (sha256 of
the reference CLI's first frames, sha256 of every vid_init() table) and
ref_lines.npz (whole lines). This is what pins the oracle on a box that has no
/root/reference."""
import numpy as np
import pytest

import oracle
import util

CASES_FAST = ["pal_bb", "i_raster", "i_vsb", "i_fm", "i_audio", "i_full", "m_full", "ntsc_bb", "i_mono", "g_full",
              "pal_bb_filter", "i_20m", "secam_bb", "l_raster", "l_full", "i_tt", "l_tt"]
# the complex tail (swap_iq, offset, passthru) and FM video; the passthru source ends inside frame 3
# --pixelrate: raster at the pixel rate + poly-phase resampler (the last one has lines of 870 / 871 samples)
CASES_PIXELRATE = ["i_px135", "i_px2025", "l_px2025", "pal_px16_s14", "m_px135_s27", "pal_px135_s136", "pal_rawbb_px135", "i_rawbb_px16", "pal_sv_px135", "ntsc_sv_f_px18", "secam_sv_f_px2025", "i_pass_px135", "pal_pass_px135_s136", "m_px135_s16", "ntsc_px16_s135",
                   # S-Video behind resampler + filter, lines of two widths: the ring of line buffers (oracle/make_golden_r05.py)
                   "ntsc_sv_f_px135_s16", "ntsc_sv_f_px18_s16", "pal60_sv_f_px27_s16", "ntsc_sv_f_px16_s27", "ntsc_sv_f_px16_s18",
                   # a resampler of 709379 phases: 27 MHz -> 4 x the PAL sub-carrier (oracle/make_golden_r06.py)
                   "pal_px27_s4fsc", "i_px27_s4fsc", "i_sis_px2025_s4fsc", "l_sis_px2025_s4fsc", "pal_sv_f_px27_s4fsc", "pal_sv_f_px16_s4fsc"]
# VBI inserters (insertion test signals, widescreen signalling, time code)
CASES_VBI = ["i_vbi", "i_vbi_tt", "m_vbi", "l_vbi", "pal_vbi_px", "i_acp_cc", "m_acp_cc", "i_wss_auto"]
CASES_A2 = ["g_a2", "m_a2", "pal_sv", "ntsc_sv_f", "secam_sv", "l_fid", "secam_fid4", "i_rawbb", "pal_rawbb", "l_rawbb"]
# the other 625 / 525-line presets; FM video with its pre-emphasis filter
CASES_PRESETS = ["pald_full", "palm_full", "paln_full", "pal525_bb", "d_full", "secami_full", "secamb_raster", "ntsci_full",
                 "pal60i_full", "pal60_bb", "palfm_f14", "ntscfm_f18", "secamfm_f2025", "i_27m"]
# sound-in-syncs: a NICAM stream of its own inside every sync pulse (oracle/make_golden_sis.py: ten runs of the reference, one output)
CASES_SIS = ["i_sis", "i_sis_filter", "l_sis_tt", "pal_sv_sis", "i_rawbb_sis", "i_sis_px135", "i_sis_px2025", "l_sis_px16_s14", "i_sis_27m"]
# rates outside the first rounds' 11 .. 28 MHz: chroma filters of 7, 19, 23 taps (oracle/make_golden_rates.py)
CASES_RATES = ["pal_8m", "pal_9m", "i_24m", "ntsc_24m", "m_4fsc", "pal_30m",
               "pal_36m", "pal_8fsc", "i_36m", "pal_27m"]       # (round 6: chroma low pass of 27 .. 31 taps, oracle/make_golden_r06.py)
# the rasters other than 625 / 525 lines and field-sequential colour (oracle/make_golden_rasters.py)
CASES_RASTERS = ["e_full", "819_bb", "a_full", "405i_full", "405_bb", "ntsc405_bb", "ntsca_full", "240am", "240_bb", "30_bb", "30am", "nbtv_bb", "nbtvam",
                 "apollo_bb", "apollofm", "apollofsc_bb", "apollofscfm", "cbs405_bb", "mcbs405_full", "apollofm_f", "apollofscfm_f"]
CASES_TAIL = ["i_offset", "i_swap_pass", "m_offset_pass", "pal_fm", "ntsc_fm", "secam_fm_tail", "pal_fm_pass", "palfm_f14_tail",
              "palfm_px135", "palfm_pass_px135", "palfm_f14_px135", "secamfm_px18", "ntscfm_f18_px135", "palfm_s14_px16", "ntscfm_s18_px16"]


@pytest.mark.parametrize("case", CASES_FAST + CASES_TAIL + CASES_PIXELRATE + CASES_VBI + CASES_A2 + CASES_PRESETS + CASES_SIS + CASES_RATES + CASES_RASTERS)
def test_oracle_stream_matches_reference_cli(golden, case):
    c = golden.cases[case]
    conf, sr = golden.conf(case)
    W, L = c["width"], c["lines"]
    nframes = c["frames"] if (c.get("extra", {}).get("passthru") or case in CASES_RASTERS) else min(2, c["frames"])
    with oracle.Oracle(conf, sr, c.get("pixel_rate", 0)) as o:
        o.set_frame(golden.frame(case))
        o.set_frame_aspect(12, 13)       # the test source: 4:3 on 832 x 576 (src/av_test.c:50)
        o.set_audio(golden.audio, True)
        if conf.passthru:
            o.set_passthru(util.passthru_signal())
        if conf.raw_bb:
            o.set_rawbb(util.rawbb_signal())
        if c.get("teletext"):
            for f in range(nframes + 1):
                o.teletext_packets(f, *golden.teletext_rows(f, golden.teletext_skip(case)))
        iq = o.render_lines(nframes * L)
    fs = c.get("frame_samples", W * L)
    ends = c.get("frame_ends") or [(n + 1) * fs for n in range(nframes)]       # (rate pairs with frames of two lengths list them)
    assert iq.shape[0] == ends[nframes - 1]
    for n in range(nframes):
        got = util.cum_sha(iq, ends[n], c)
        assert got == c["sha256_cumulative"][n], "frame %d of %s differs from the reference" % (n + 1, case)
    # the excerpted lines, for a readable failure
    idx = golden.lines[case + "_idx"]
    ref = golden.lines[case]
    for j, g in enumerate(idx):
        if g >= nframes * L:
            continue
        mine = iq[g * W:(g + 1) * W, : (1 if c["real"] else 2)]
        assert np.array_equal(mine, ref[j]), "line %d of %s" % (g, case)


@pytest.mark.parametrize("case", ["i_full", "m_full", "pal_bb_filter", "g_full", "i_20m", "l_full", "l_tt", "pal_fm", "ntsc_fm",
                                  "i_px135", "l_px2025", "pal_px135_s136"])
def test_oracle_tables_match_reference(golden, case):
    c = golden.cases[case]
    conf, sr = golden.conf(case)
    with oracle.Oracle(conf, sr, c.get("pixel_rate", 0)) as o:
        if c.get("teletext"):
            o.teletext_packets(0, golden.teletext_rows(0)[0], 0)
        for k in ("width", "half_width", "active_width", "active_left", "white_level", "black_level",
                  "blanking_level", "sync_level", "colour_lookup_width", "burst_left", "burst_width",
                  "burst_phase_i", "burst_phase_q", "chroma_ataps", "fm_mono_level", "nicam_ntaps",
                  "nicam_sps", "nicam_dsl", "nicam_decimation", "nicam_cc_len"):
            assert o.info[k] == c["info"][k], k
        for name, ref in c["tables"].items():
            if name == "fm_secam_bell" and ref["len"]:
                # 65535 entries; the reference writes (and the probe cannot read) a 65536th
                ref = dict(ref)
            a = o.table(name, util.TABLE_DTYPES[name])
            assert a.size == ref["len"], name
            assert util.sha256(a.tobytes()) == ref["sha256"], name


@pytest.mark.parametrize("case", ["i_full", "pal_bb"])
def test_oracle_sink_formats_match_reference(golden, case):
    """rf_file.c's six sample formats (oracle/oracle_sink.c) against `hacktv_ref -t <type>`."""
    c = golden.cases[case]
    conf, sr = golden.conf(case)
    with oracle.Oracle(conf, sr) as o:
        o.set_frame(golden.frame(case))
        o.set_audio(golden.audio, True)
        iq = o.render_lines(625)
    for tname in oracle.SINK_TYPES:
        out = oracle.sink_convert(iq, tname, not c["real"])
        assert util.sha256(out.tobytes()) == golden.sink_formats["%s:%s" % (case, tname)], tname
