"""Host side of the engine (no GPU): the table builder and the audio-rate
control path of libhvk against the oracle."""
import numpy as np
import pytest

import hacktv_amd as H
import oracle
import util

HOST_TABLES = ["syncs", "colour_lookup", "burst_win", "chroma_taps", "chroma_ghost", "vfilter_itaps",
               "vfilter_qtaps", "fm_mono_lut", "nicam_taps", "nicam_cc", "limiter_shape", "limiter_vtaps",
               "limiter_ftaps", "fm_secam_lut", "fm_secam_bell", "fm_secam_fir", "secam_l_fir", "teletext_lut", "fm_video_lut", "resampler_taps"]


@pytest.mark.parametrize("case", ["i_full", "m_full", "pal_bb_filter", "g_full", "ntsc_bb", "i_20m", "l_full", "l_tt",
                                  "pal_fm", "ntsc_fm", "secam_fm_tail", "i_px135", "i_px2025", "l_px2025", "pal_px16_s14",
                                  "m_px135_s27", "pal_px135_s136", "m_px135_s16", "ntsc_px16_s135",
                                  "e_full", "a_full", "405i_full", "ntsc405_bb", "240_bb", "30_bb", "nbtv_bb", "apollofm", "apollofsc_bb", "mcbs405_full"])
def test_host_tables_equal_oracle(golden, case):
    conf, sr = golden.conf(case)
    pr = golden.cases[case].get("pixel_rate", 0)
    with H.Engine(conf, sr, device=-1, pixel_rate=pr) as e, oracle.Oracle(conf, sr, pr) as o:
        if golden.cases[case].get("teletext"):
            o.teletext_packets(0, golden.teletext_rows(0)[0], 0)
        for name in HOST_TABLES:
            assert np.array_equal(e.table(name, util.TABLE_DTYPES[name]), o.table(name, util.TABLE_DTYPES[name])), name
        for k in ("width", "half_width", "active_width", "active_left", "lines", "active_lines", "white_level",
                  "black_level", "blanking_level", "sync_level", "colour_lookup_width", "burst_left", "burst_width"):
            assert e.info[k] == o.info[k], k
        assert e.info["max_width"] == o.info["max_width"]
        if pr:
            # geometry of the resampled stream: frame length, line widths, start-up samples
            assert e.info["frame_samples"] == golden.cases[case]["frame_samples"]
            o.set_frame(golden.frame(case))
            o.render_lines(700)
            w = o.last_widths()
            assert np.array_equal(e.line_widths(0, 700), w)
            assert np.array_equal(e.line_widths(650, 50), w[650:])
            # a frame begins where its first line begins (frames of two lengths with some rate pairs)
            L = e.info["lines"]
            assert e.frame_start(1) == int(np.sum(w[:L])) and e.frame_start(0) == 0


@pytest.mark.parametrize("case", ["i_full", "m_full", "i_audio", "l_full", "g_a2", "m_a2", "e_full", "a_full", "apollofm", "mcbs405_full"])
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


@pytest.mark.parametrize("mode,sr", [("g", 16000000), ("m", 13500000)])
def test_a2_tone_thread_changes_nothing(golden, monkeypatch, mode, sr):
    """A2 stereo: the identification tone and pilot run ahead on a thread of their own (hvk_audio.c). Loud noise as sound,
    6 frames (every phasor passes its amplitude correction several times, the ring of blocks wraps): the carriers equal
    those of the same chains in one thread (HVK_AUDIO_THREADS=0), piecewise requests equal one request, and an engine
    that takes the state over in the middle -- the thread's state is a replay inside a block -- continues bit for bit."""
    conf = H.preset(mode, H.FLAG_FILTER)
    conf.a2stereo = 1
    rng = np.random.default_rng(17)
    pcm = (rng.integers(-32768, 32768, size=(32000, 2))).astype(np.int16)
    def stream(pieces, handover_at=None):
        out = []
        e = H.Engine(conf, sr, device=-1)
        fs, prime = e.info["frame_samples"], e.info["startup_samples"]
        total = 6 * fs
        for _ in range(8):
            e.audio_write(pcm)
        pos = prime
        for n in pieces(total):
            if handover_at is not None and pos - prime >= handover_at:
                st = e.sound_state_export()
                e2 = H.Engine(conf, sr, device=-1)
                src = e2.sound_state_import(st)
                e.close()
                e = e2
                e.audio_write(pcm[src % len(pcm):])
                for _ in range(8):
                    e.audio_write(pcm)
                handover_at = None
            out.append(e.host_side_streams(pos, n)[0])
            pos += n
        e.close()
        return np.concatenate(out)
    W = 1024 if mode == "g" else 858
    whole = lambda total: [total]
    ragged = lambda total: [70001] * (total // 70001) + ([total % 70001] if total % 70001 else [])
    lines = lambda total: [67 * W] * (total // (67 * W)) + ([total % (67 * W)] if total % (67 * W) else [])   # (a state goes over between lines)
    want = stream(whole)
    assert np.array_equal(stream(ragged), want)
    assert np.array_equal(stream(lines, handover_at=3 * 67 * W), want)
    monkeypatch.setenv("HVK_AUDIO_THREADS", "0")
    assert np.array_equal(stream(whole), want)
    assert np.array_equal(stream(lines, handover_at=5 * 67 * W), want)


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


def test_secam_lines_from_derived_entry_states(golden, monkeypatch):
    """What the device does (hvk_secam.hip), run by the host code that shares its arithmetic (hvk_secam_chain.h,
    HVK_SECAM_SPEC=K): every line on its own from an entry state derived by walking K lines before it from nothing,
    then checked against the true chain. The result equals the serial chain's whatever K is (wrong starts are
    found and redone); with 12 warm-up lines next to none are wrong, with 4 most are."""
    frames = [golden.frame("l_raster")]
    rng = np.random.default_rng(3)
    frames += [rng.integers(0, 1 << 24, size=(576, 832), dtype=np.uint32) for _ in range(2)]
    conf = H.preset("l", H.FLAG_NOAUDIO)

    def run():
        with H.Engine(conf, 16000000, device=-1) as e:
            return np.concatenate([e.host_secam_stream(f) for f in frames]), e.secam_stats()
    want, _ = run()
    monkeypatch.setenv("HVK_SECAM_SPEC", "12")
    got, st = run()
    assert np.array_equal(got, want) and st["tasks"] > 1700 and st["mismatches"] <= 5, st
    monkeypatch.setenv("HVK_SECAM_SPEC", "4")
    got, st = run()
    assert np.array_equal(got, want) and st["mismatches"] > st["tasks"] // 2, st


# ---- the complex tail: offset phasor, passthru queue, FM video (hvk_tail.c) ----

def _oracle_frames(golden, conf, sr, case, nlines, passthru=False):
    with oracle.Oracle(conf, sr) as o:
        o.set_frame(golden.frame(case))
        o.set_audio(golden.audio, True)
        if passthru:
            o.set_passthru(util.passthru_signal())
        return o.render_lines(nlines)


@pytest.mark.parametrize("case", ["i_offset", "m_offset_pass"])
def test_offset_stream_reproduces_the_oracle(golden, case):
    """out_with_offset == cint16_mul(out_without, host offset stream): the host phasor chain
    (incl. its start at INT16_MAX, the 32767-sample re-normalisation and the start-up line it
    runs over when the filter is on) against the oracle's per-line process."""
    conf, sr = golden.conf(case)
    c = golden.cases[case]
    W, nl = c["width"], 700
    want = _oracle_frames(golden, conf, sr, case, nl, passthru=bool(conf.passthru)).astype(np.int32)
    plain_conf, _ = golden.conf(case)
    plain_conf.offset = 0
    plain_conf.passthru = 0
    a = _oracle_frames(golden, plain_conf, sr, case, nl).astype(np.int32)
    with H.Engine(conf, sr, device=-1) as e:
        # forward only, in two pieces
        b = np.concatenate([e.host_offset_stream(0, 300 * W), e.host_offset_stream(300 * W, (nl - 300) * W)]).astype(np.int32)
        prime = e.info["delay_lines"] * W
    assert np.abs(b[:32766 - prime]).max() <= 1 and np.abs(b[40000:41000]).max() > 30000   # the reference's quirk
    got = np.empty_like(a)
    got[:, 0] = (a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1]) >> 15
    got[:, 1] = (a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]) >> 15
    got = got.astype(np.int16).astype(np.int32)
    if conf.passthru:
        p = util.passthru_signal().astype(np.int32)
        n = min(len(p) // W, nl) * W     # whole lines only
        got[:n] = (got[:n] + p[:n]).astype(np.int16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", ["pal_fm", "ntsc_fm", "secam_fm_tail", "pal_fm_pass"])
def test_fm_video_host_tail_equals_oracle(golden, case):
    """hvk_host_fm_video() over the oracle's un-modulated composite == the oracle's FM output
    (pinned to the reference CLI by test_oracle_golden): FM phasor, then swap / offset /
    passthru on the host, fed in uneven pieces."""
    conf, sr = golden.conf(case)
    c = golden.cases[case]
    W, L = c["width"], c["lines"]
    nl = 3 * L if conf.passthru else 400
    want = _oracle_frames(golden, conf, sr, case, nl, passthru=bool(conf.passthru))
    pre_conf, _ = golden.conf(case)
    pre_conf.modulation = 0          # HVK_NONE: same levels (conf.level is 1), no modulator
    pre_conf.swap_iq = pre_conf.passthru = 0
    pre_conf.offset = 0
    pre = _oracle_frames(golden, pre_conf, sr, case, nl)
    with H.Engine(conf, sr, device=-1) as e:
        if conf.passthru:
            sig = util.passthru_signal()
            e.passthru_write(sig[:1000])
            e.passthru_write(sig[1000:])
        cuts = [0, 7 * W, 8 * W, 250 * W, nl * W]
        got = np.concatenate([e.host_fm_video(pre[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(got, want)


def test_audio_runs_over_the_resamplers_startup_lines(golden):
    """--pixelrate: the audio process has run over the never-emitted start-up chunks
    (info.startup_samples) when the first output sample is made."""
    case = "i_px135"
    conf, sr = golden.conf(case)
    pr = golden.cases[case]["pixel_rate"]
    nl = 400
    with H.Engine(conf, sr, device=-1, pixel_rate=pr) as e, oracle.Oracle(conf, sr, pr) as o:
        o.set_frame(golden.frame(case))
        o.set_audio(golden.audio, True)
        o.render_lines(nl)
        want = o.last_carrier()
        for _ in range(2):
            e.audio_write(golden.audio)
        assert e.info["startup_samples"] == 2048
        got, _, _ = e.host_side_streams(e.info["startup_samples"], len(want))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", ["l_sis_px2025_s4fsc", "i_sis_px2025_s4fsc"])
def test_sound_chains_ahead_of_the_requests_where_lines_have_two_widths(golden, case):
    """Sound-in-syncs keeps the sound chains lines AHEAD of the requests (three behind SECAM's threaded colour process, one more
    behind the resampler), counted in lines of the widest width; at 4 x the PAL sub-carrier most lines are a sample narrower, so the
    chains stand up to five lines past a request's end -- and the next request starts inside lines that must still be kept. Four
    were (round 6's GPU fuzzer: SECAM-L's first request was refused); the carriers of frame after frame, asked for in one piece
    and in pieces that end inside lines, equal the oracle's, and every line's burst is there."""
    conf, sr = golden.conf(case)
    c = golden.cases[case]
    pr, fs, L = c["pixel_rate"], c["frame_samples"], c["lines"]
    with oracle.Oracle(conf, sr, pr) as o:
        o.set_frame(golden.frame(case))
        o.set_audio(golden.audio, True)
        o.render_lines(2 * L)
        want = o.last_carrier()
    def run(pieces):
        with H.Engine(conf, sr, device=-1, pixel_rate=pr) as e:
            for _ in range(3):
                e.audio_write(golden.audio)
            pos, out = e.info["startup_samples"], []
            for n in pieces:
                out.append(e.host_side_streams(pos, n)[0])
                pos += n
            bursts = e.host_sis_bursts(0, 2 * L)
        return np.concatenate(out), bursts
    whole, b0 = run([fs, fs])
    assert np.array_equal(whole[: len(want)], want[: len(whole)])
    parts, b1 = run([fs - 777, 777 + 5, fs - 5 - 1135 * 3, 1135 * 3])
    assert np.array_equal(parts, whole) and np.array_equal(b0, b1)
    assert (b0[:, 7] >= 44).all()           # (every line has its burst: 44 or 48 bits)


def test_passthru_needs_whole_lines_and_stays_ended(golden):
    conf, sr = golden.conf("pal_fm_pass")
    W = golden.cases["pal_fm_pass"]["width"]
    with H.Engine(conf, sr, device=-1) as e:
        e.passthru_write(np.ones((W + 10, 2), np.int16))
        z = np.zeros((3 * W, 2), np.int16)
        a = e.host_fm_video(z)
        e.passthru_write(np.ones((4 * W, 2), np.int16))     # too late: the source has ended
        b = e.host_fm_video(z)
    conf.passthru = 0
    with H.Engine(conf, sr, device=-1) as e:
        ref = np.concatenate([e.host_fm_video(z), e.host_fm_video(z)])
    assert np.array_equal(a[:W], ref[:W] + 1) and np.array_equal(a[W:], ref[W:3 * W]) and np.array_equal(b, ref[3 * W:])


@pytest.mark.parametrize("case", ["i_sis", "i_sis_filter", "l_sis_tt", "pal_sv_sis", "i_rawbb_sis", "i_sis_px135", "i_sis_px2025", "l_sis_px16_s14"])
@pytest.mark.parametrize("loud", [False, True])
def test_sound_in_syncs_bursts_equal_the_oracles(golden, case, loud):
    """--sis: the bits of every line's burst (hvk_host_sis_bursts(): the host half, framing in step with the sound chains)
    against the oracle's restatement of src/sis.c:155-201, which equals the reference CLI on these cases
    (test_oracle_golden.py) -- on the test tone, and on full-scale noise, whose 32-sample blocks all differ: which block
    a NICAM frame carries then shows in every frame (the oracle's and the engine's reading of the reference's unlocked
    hand-over: the newest block of an earlier pipeline step). PAL-I: one never-emitted invocation in front of line 1;
    SECAM-L: three, and the chains run two lines ahead of the requests; --raw-bb-file: none (the process that reads the
    lines in has one line, not the raster's three); behind the resampler the audio process's lines are the resampler's
    chunks, of two widths."""
    conf, sr = golden.conf(case)
    audio = golden.audio
    if loud:
        audio = np.random.default_rng(5).integers(-32768, 32768, (len(golden.audio), 2)).astype(np.int16)
    n = 625 * 2 + 100
    pr = golden.cases[case].get("pixel_rate", 0)
    with oracle.Oracle(conf, sr, pr) as o:
        o.set_frame(golden.frame(case))
        o.set_audio(audio, True)
        if conf.raw_bb:
            o.set_rawbb(util.rawbb_signal())
        if golden.cases[case].get("teletext"):
            for f in range(4):
                o.teletext_packets(f, *golden.teletext_rows(f))
        o.render_lines(n)
        want = o.sis_bursts(0, n)
    with H.Engine(conf, sr, device=-1, pixel_rate=pr) as e:
        for _ in range(3):
            e.audio_write(audio)
        got = np.concatenate([e.host_sis_bursts(0, 700), e.host_sis_bursts(700, 1), e.host_sis_bursts(701, n - 701)])
    assert set(want[:, 7]) == {46, 50}
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "line %d: %s != %s" % (bad[0], got[bad[0]], want[bad[0]])


@pytest.mark.parametrize("mode,sr,pr,flags,members,words", [
    ("d", 14000000, 0, H.FLAG_FILTER, {}, "luma notch"),                                  # the reference reads past its line buffer there
    ("pal-k", 40000001, 13500000, 0, {}, "lowest terms"),                                  # 40 000 001 phases (27 MHz -> 4 f_sc, 709379 phases and this row until round 6, is rendered now)
    ("pal", 48000000, 0, 0, {}, "reads further past the line"),                           # a 37-tap chroma low pass: the over-read model holds 32 samples (SiS at 27 MHz, this row until round 6, is rendered now)
    ("m", 13500000, 0, 0, {"wss": 8}, "625-line"),
])
def test_refusals_say_why(capfd, mode, sr, pr, flags, members, words):
    """A configuration the engine does not render is refused at open with HVK_UNSUPPORTED and one line on stderr that says
    why (the reference prints its own refusals the same way); nothing is rendered approximately."""
    conf = H.preset(mode, flags)
    for k, v in members.items():
        setattr(conf, k, v)
    with pytest.raises(H.HvkError) as err:
        H.Engine(conf, sr, device=-1, pixel_rate=pr)
    assert err.value.code == H.HVK_UNSUPPORTED
    said = capfd.readouterr().err
    assert "libhvk: refused: " in said and words in said, said


@pytest.mark.parametrize("mode,sr,pr,flags,members", [
    ("ntsc", 16000000, 27000000, H.FLAG_FILTER, {"s_video": 1}),       # lines of two widths AND the filter: rendered since round 5 (hvk_k_svq: the ring of line buffers)
    ("ntsc", 16000000, 13500000, H.FLAG_FILTER, {"s_video": 1}),
    ("ntsc", 16000000, 27000000, 0, {"s_video": 1}),                   # lines of two widths, but no filter to give a line another line's width
    ("ntsc", 13500000, 18000000, H.FLAG_FILTER, {"s_video": 1}),       # the filter, but every line 858 samples
    ("pal", 16000000, 13500000, H.FLAG_FILTER, {"s_video": 1}),
])
def test_s_video_behind_the_resampler_where_it_is_defined(mode, sr, pr, flags, members):
    conf = H.preset(mode, flags)
    for k, v in members.items():
        setattr(conf, k, v)
    with H.Engine(conf, sr, device=-1, pixel_rate=pr) as e:
        assert e.info["width"] > 0


@pytest.mark.parametrize("sr", [13500000, 14000000, 16000000])
@pytest.mark.parametrize("first", ["narrow", "none", "full"])
def test_secam_fill_slots_see_the_init_frame_then_the_first_picture(sr, first):
    """The colour process's two never-emitted fill slots: the first is taken before the source has been read -- the full
    active width of vid_init()'s frame, no pixels (src/video.c:4169-4177) --, the second with the place and width of the
    stream's first picture. With a first picture narrower than the raster the filter state they leave differs from a
    full-width one's, and at 13.5 / 14 MHz -- where the sub-carrier's last samples run past the line into what the next
    line's low pass reads -- the first field identification line shows it (tests/ref_random_check.py secam_sv_narrow has
    the unmodified reference on it). S-Video: the Q channel IS the sub-carrier, so the host's chain (hvk_secam.c, the
    arithmetic the device kernels share) is compared with the oracle sample for sample."""
    conf = H.preset("secam", H.FLAG_NOAUDIO)
    conf.s_video = 1
    conf.secam_field_id = 1
    rng = np.random.default_rng(11)
    with oracle.Oracle(conf, sr) as o:
        w, h, L, W = o.info["active_width"], o.info["active_lines"], o.info["lines"], o.info["width"]
        full = rng.integers(0, 1 << 24, (h, w), dtype=np.uint32)
        pics = [{"narrow": np.ascontiguousarray(full[:100, :301]), "none": None, "full": full}[first], full]
        want = []
        for p in pics:
            o.set_frame(p if p is not None else np.zeros((0, 0), np.uint32))
            want.append(o.render_lines(L)[:, 1].reshape(L, W))
    with H.Engine(conf, sr, device=-1) as e:
        got = [e.host_secam_stream(p).reshape(L, W) for p in pics]
    for f in range(2):
        bad = np.nonzero((got[f] != want[f]).any(axis=1))[0]
        assert bad.size == 0, "frame %d: %d lines differ, first line %d" % (f, bad.size, bad[0] + 1)
