"""Host side of the engine (no GPU): the table builder and the audio-rate
control path of libhvk against the oracle."""
import numpy as np
import pytest

import hacktv_amd as H
import oracle
import util

HOST_TABLES = ["syncs", "colour_lookup", "burst_win", "chroma_taps", "chroma_ghost", "vfilter_itaps",
               "vfilter_qtaps", "fm_mono_lut", "nicam_taps", "nicam_cc", "limiter_shape", "limiter_vtaps",
               "limiter_ftaps", "fm_secam_lut", "fm_secam_bell", "fm_secam_fir", "secam_l_fir", "teletext_lut"]


@pytest.mark.parametrize("case", ["i_full", "m_full", "pal_bb_filter", "g_full", "ntsc_bb", "i_20m", "l_full", "l_tt"])
def test_host_tables_equal_oracle(golden, case):
    conf, sr = golden.conf(case)
    with H.Engine(conf, sr, device=-1) as e, oracle.Oracle(conf, sr) as o:
        if golden.cases[case].get("teletext"):
            o.teletext_packets(0, golden.teletext_rows(0), 0)
        for name in HOST_TABLES:
            assert np.array_equal(e.table(name, util.TABLE_DTYPES[name]), o.table(name, util.TABLE_DTYPES[name])), name
        for k in ("width", "half_width", "active_width", "active_left", "lines", "active_lines", "white_level",
                  "black_level", "blanking_level", "sync_level", "colour_lookup_width", "burst_left", "burst_width"):
            assert e.info[k] == o.info[k], k


@pytest.mark.parametrize("case", ["i_full", "m_full", "i_audio", "l_full"])
def test_serial_carrier_stream_equals_oracle(golden, case):
    """The host pre-pass (FM/AM phasor chains, limiter, 32 kHz tick) produces the same
    per-sample contribution as the oracle's per-sample loop, including the
    delay_lines * width offset."""
    conf, sr = golden.conf(case)
    nl = 800
    with H.Engine(conf, sr, device=-1) as e, oracle.Oracle(conf, sr) as o:
        W = o.info["width"]
        o.set_audio(golden.audio, True)
        o.render_lines(nl)
        want = o.last_carrier()
        for _ in range(2):
            e.audio_write(golden.audio)
        got, _, _ = e.host_side_streams(e.info["delay_lines"] * W, nl * W)
    assert np.array_equal(got, want)


def _nicam_from_symbols(e, sym, k0, first, count):
    """numpy model of the filter kernel's NICAM stage (src/nicam728.c:342-411)."""
    taps = e.table("nicam_taps", np.int16).astype(np.int64)
    cc = e.table("nicam_cc", np.int16).reshape(-1, 2).astype(np.int64)
    sps, dsl, dec = 44, 4, 91   # 16 Msps: asserted below
    bi = np.zeros(count, np.int64)
    bq = np.zeros(count, np.int64)
    for j, sv in enumerate(sym):
        if sv == 0xFF:
            continue
        k = k0 + j
        st = sps * k - (k * dsl) // dec
        lo, hi = max(st, first), min(st + len(taps), first + count)
        if hi <= lo:
            continue
        cs = [0, 1, 3, 2][sv]
        t = taps[lo - st: hi - st]
        bi[lo - first: hi - first] += t if cs & 1 else -t
        bq[lo - first: hi - first] += t if cs & 2 else -t
    bi = ((bi + 32768) % 65536) - 32768
    bq = ((bq + 32768) % 65536) - 32768
    m = (first + np.arange(count)) % len(cc)
    oi = (bi * cc[m, 0] - bq * cc[m, 1]) >> 15
    oq = (bi * cc[m, 1] + bq * cc[m, 0]) >> 15
    return np.stack([oi, oq], axis=1)


def test_nicam_symbols_reproduce_the_reference_signal(golden):
    """Host NICAM framing + symbol schedule: the signal rebuilt from the emitted
    symbols equals (oracle with NICAM) - (oracle without NICAM), mod 2^16, over
    5 NICAM frames."""
    conf_a, sr = golden.conf("i_audio")      # FM + NICAM, no filter
    conf_b, _ = golden.conf("i_fm")          # FM only
    nl = 90
    with oracle.Oracle(conf_a, sr) as a, oracle.Oracle(conf_b, sr) as b:
        for o in (a, b):
            o.set_audio(golden.audio, True)
        W = a.info["width"]
        d = a.render_lines(nl).astype(np.int64) - b.render_lines(nl).astype(np.int64)
    with H.Engine(conf_a, sr, device=-1) as e:
        assert (e.table("nicam_taps", np.int16).size, e.table("nicam_cc", np.int16).size) == (221, 4000)
        e.audio_write(golden.audio)
        _, sym, k0 = e.host_side_streams(0, nl * W)
        got = _nicam_from_symbols(e, sym, k0, 0, nl * W)
    assert np.array_equal((d - got) % 65536, np.zeros_like(d))


def test_secam_host_prepass_equals_oracle(golden):
    """The SECAM colour pre-pass (serial on the host: IIR state for ever, FM tail into the
    next line's filter, the two pipeline-fill passes) against the oracle. Outside the
    active picture the luma notch does not act, so there
        oracle raster (SECAM) - oracle raster (no colour) == the added sub-carrier, mod 2^16;
    any slip in the serial state shows up everywhere after it."""
    conf_c, sr = golden.conf("l_raster")
    import hacktv_amd as H2
    conf_m = H2.preset("l", H2.FLAG_NOAUDIO | H2.FLAG_NOCOLOUR)
    frame = golden.frame("l_raster")
    nframes = 2
    with oracle.Oracle(conf_c, sr) as a, oracle.Oracle(conf_m, sr) as b:
        a.set_frame(frame)
        b.set_frame(frame)
        a.render_lines(625 * nframes)
        b.render_lines(625 * nframes)
        d = (a.last_raster().astype(np.int64) - b.last_raster().astype(np.int64)).reshape(nframes * 625, 1024)
    with H.Engine(conf_c, sr, device=-1) as e:
        got = np.concatenate([e.host_secam_stream(frame) for _ in range(nframes)]).astype(np.int64).reshape(nframes * 625, 1024)
        al, aw = e.info["active_left"], e.info["active_width"]
    outside = np.r_[0:al, al + aw:1024]
    assert not ((d[:, outside] - got[:, outside]) % 65536).any()
    # and nothing is added outside the sub-carrier window
    assert not got[:, :82].any()
