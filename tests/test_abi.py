"""The C-ABI library loads and exports every entry point include/hacktv_amd.h
declares (no device needed, nothing is computed)."""
import ctypes
import os
import re

import hacktv_amd as H
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "hacktv_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hvk_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(H.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_lists_every_declared_symbol():
    from hacktv_amd.engine import SYMBOLS
    assert sorted(SYMBOLS) == declared_functions()


def test_presets_match_the_reference_mode_ids():
    ids = []
    i = 0
    while H.lib().hvk_preset_id(i):
        ids.append(H.lib().hvk_preset_id(i).decode())
        i += 1
    # every mode of the reference's vid_configs[] (src/video.c:1956-2008) but ntsc-bs (DANCE digital audio) and the six MAC
    # modes (a packet multiplex, not a raster): the 625 / 525-line ones, then 819, 405, Baird, NBTV, Apollo, CBS
    assert ids == ["i", "b", "g", "pal", "l", "secam", "m", "ntsc", "pal-fm", "secam-fm", "ntsc-fm",
                   "pal-d", "pal-k", "pal-m", "pal-n", "525pal", "d", "k", "secam-i", "secam-b", "secam-g", "ntsc-i",
                   "pal60-i", "pal60",
                   "e", "819", "a", "ntsc-a", "405-i", "405", "ntsc-405", "240-am", "240", "30-am", "30", "nbtv-am", "nbtv",
                   "apollo-fsc-fm", "apollo-fsc", "apollo-fm", "apollo", "m-cbs405", "cbs405"]
    assert H.lib().hvk_config_preset(ctypes.byref(H.HvkConfig()), b"nope") == -1


def test_no_cpu_path_without_a_device():
    """device -1 builds host tables only; rendering must fail loudly."""
    c = H.preset("i", H.FLAG_FILTER)
    with H.Engine(c, 16000000, device=-1) as e:
        assert e.info["width"] == 1024 and e.info["frame_samples"] == 640000
        try:
            e.render(1)
        except H.HvkError as err:
            assert err.code == H.HVK_NO_DEVICE
        else:
            raise AssertionError("render without a device did not fail")


def test_unsupported_configurations_are_refused():
    """Configurations outside the engine's scope fail at open with HVK_UNSUPPORTED."""
    bad = []
    c = H.preset("i"); c.fm_mono_preemph = 3; bad.append((c, 16000000))       # J.17 FM pre-emphasis
    c = H.preset("i"); c.type = 2; bad.append((c, 16000000))                  # a raster type whose number of lines the configuration does not have
    c = H.preset("i"); c.type = 8; bad.append((c, 16000000))                  # not a raster (the reference's VID_MAC)
    for conf, sr in bad:
        try:
            H.Engine(conf, sr, device=-1)
        except H.HvkError as err:
            assert err.code == -4
        else:
            raise AssertionError("an unsupported configuration was accepted")


def test_bad_arguments_fail_with_codes_not_crashes():
    """Every entry point checks its arguments (host-tables engine: no device needed)."""
    L = H.lib()
    c = H.preset("i", H.FLAG_FILTER)
    with H.Engine(c, 16000000, device=-1) as e:
        h = e.h
        buf = (ctypes.c_int16 * 64)()
        w = (ctypes.c_int32 * 4)()
        assert L.hvk_get_info(h, None) == H.HVK_ERROR
        assert L.hvk_frame_upload(h, 99, None, 0, 0, 0, 0, 0) == H.HVK_ERROR          # no such slot
        assert L.hvk_teletext_packets(h, 0, buf, 1) == H.HVK_UNSUPPORTED              # not a teletext configuration
        assert L.hvk_passthru_write(h, buf, 16) == H.HVK_UNSUPPORTED                  # no --passthru
        assert L.hvk_host_offset_stream(h, 0, 16, buf) == H.HVK_UNSUPPORTED           # no --offset
        assert L.hvk_host_fm_video(h, buf, 16) == H.HVK_UNSUPPORTED                   # not FM video
        assert L.hvk_line_widths(h, -1, 4, w) == H.HVK_ERROR
        assert L.hvk_line_widths(h, 0, 4, w) == H.HVK_OK and list(w) == [1024] * 4
        assert L.hvk_render(h, 1, None, None) == H.HVK_NO_DEVICE
        assert L.hvk_fetch(h, buf, 0, 16) == H.HVK_NO_DEVICE
        assert L.hvk_fetch_as(h, buf, 0, 16, 99, 0) == H.HVK_ERROR                    # no such sample type
        assert L.hvk_sync(h) == H.HVK_NO_DEVICE
        assert L.hvk_set_chroma_ghost(h, buf, 1000) == H.HVK_ERROR
        assert L.hvk_set_levels(h, 3) == H.HVK_ERROR and L.hvk_set_levels(h, -1) == H.HVK_ERROR
        assert L.hvk_set_levels(h, 2) == H.HVK_OK and L.hvk_set_levels(h, 0) == H.HVK_OK
        assert L.hvk_audio_write(h, None, 0) in (H.HVK_OK, H.HVK_ERROR)
    assert L.hvk_get_info(None, None) == H.HVK_ERROR
    assert L.hvk_render(None, 1, None, None) == H.HVK_ERROR
    L.hvk_close(None)


def test_rate_pairs_the_resampler_refuses():
    """--pixelrate: up to 20 000 000 phases (up to 256 until round 6; beyond that a sample's taps come from HBM, hvk_k_resample<true>)
    and a decimation of up to four times the interpolation. (A pair at which a raster frame is not a whole number of samples --
    450450 * 32 / 27 -- is taken since round 3: frames of two lengths, hvk_frame_start().)"""
    c = H.preset("m", 0)
    with H.Engine(c, 16000000, device=-1, pixel_rate=13500000) as e:
        assert e.info["frame_samples"] == 533867 and [e.frame_start(i) for i in range(4)] == [0, 533867, 1067734, 1601600]
    with H.Engine(c, 13500000, device=-1) as e:
        assert [e.frame_start(i) for i in range(3)] == [0, 450450, 900900]
    with H.Engine(H.preset("pal", 0), 17734475, device=-1, pixel_rate=27000000) as e:      # 709379 : 1080000
        assert e.info["frame_samples"] == 709379
    for sr, pr in ((40000001, 13500000), (3000000, 27000000)):     # L = 40000001; D = 9 L
        try:
            H.Engine(c, sr, device=-1, pixel_rate=pr)
        except H.HvkError as err:
            assert err.code == H.HVK_UNSUPPORTED
        else:
            raise AssertionError("accepted %d / %d" % (sr, pr))
    # together with --passthru, --s-video and --raw-bb-file the resampler is taken since round 3
    g = util.Golden()
    for case in ("i_pass_px135", "pal_sv_px135", "pal_rawbb_px135"):
        conf, sr = g.conf(case)
        assert conf.passthru or conf.s_video or conf.raw_bb
        H.Engine(conf, sr, device=-1, pixel_rate=g.cases[case]["pixel_rate"]).close()
    with H.Engine(H.preset("i", 0), 16000000, device=-1, pixel_rate=16000000) as e:   # same rate: no resampler
        assert e.info["max_width"] == 1024 and e.info["startup_samples"] == 0


def test_sample_rates_without_a_kernel_are_refused_at_open():
    """Rates whose chroma filter or NICAM pulse has no kernel fail in hvk_open, not at the first render."""
    for mode, flags, sr in (("i", H.FLAG_NOAUDIO, 64000000), ("i", 0, 48000000), ("pal", 0, 3000000), ("pal", 0, 48000000), ("i", 0, 40000000),     # (36 MHz -- 27 chroma taps -- is rendered since round 6)
                            ("l", H.FLAG_NOAUDIO, 13500000), ("secam", 0, 14750000)):   # SECAM: the notch would leave the line
        try:
            H.Engine(H.preset(mode, flags), sr, device=-1)
        except H.HvkError as err:
            assert err.code == H.HVK_UNSUPPORTED
        else:
            raise AssertionError("%s at %d Hz accepted" % (mode, sr))
    for sr in (7000000, 9000000, 12000000, 13500000, 14000000, 16000000, 17734475, 20250000, 24000000, 27000000, 30000000, 33000000, 36000000, 38000000):
        with H.Engine(H.preset("i", H.FLAG_NOAUDIO), sr, device=-1) as e:
            assert e.info["sample_rate"] == sr
    with H.Engine(H.preset("i"), 27000000, device=-1) as e:      # NICAM's 373-tap pulse at the top of the range
        assert e.info["has_nicam"] == 1


def test_lines_held_by_the_other_inserters():
    """hvk_vbi_lines_held(): where a teletext packet must not go (the reference's vbialloc, src/teletext.c:1219) --
    the 625-line positions of VITS (src/vits.c), VITC (src/vitc.c), ACP (src/acp.c:93-108), CC608 (src/cc608.c),
    WSS (line 23) and the SECAM field identification lines."""
    c = H.preset("i", H.FLAG_NOAUDIO)
    with H.Engine(c, 16000000, device=-1) as e:
        assert e.vbi_lines_held() == []
    c.vits, c.vitc, c.acp, c.cc608, c.wss = 1, 1, 1, 1, 1
    with H.Engine(c, 16000000, device=-1) as e:
        want = set([17, 18, 330, 331]) | set([19, 21, 332, 334]) | set(range(9, 19)) | set(range(321, 331)) | set([22, 23])
        assert set(e.vbi_lines_held()) == want
    c = H.preset("l", H.FLAG_NOAUDIO)
    c.secam_field_id, c.secam_field_id_lines = 1, 5
    with H.Engine(c, 16000000, device=-1) as e:
        assert e.vbi_lines_held() == list(range(7, 12)) + list(range(320, 325))


def test_a_caller_built_against_another_layout_is_told_so():
    """hvk_config_t.struct_size / hvk_info_t.struct_size: an embedder whose structs are not this library's (a shorter,
    older hvk_config_t; an info struct of another size) gets HVK_ERROR from hvk_open() / hvk_get_info() -- it is not
    misread, and nothing is written into a struct of the wrong size."""
    L = H.lib()
    c = H.preset("i", H.FLAG_FILTER)
    assert c.struct_size == ctypes.sizeof(H.HvkConfig)
    h = ctypes.c_void_p()
    for bad in (0, ctypes.sizeof(H.HvkConfig) - 32, ctypes.sizeof(H.HvkConfig) + 8):
        c.struct_size = bad
        assert L.hvk_open(ctypes.byref(h), ctypes.byref(c), 16000000, -1, 1) == H.HVK_ERROR and not h.value
        assert L.hvk_open_rates(ctypes.byref(h), ctypes.byref(c), 16000000, 0, -1, 1) == H.HVK_ERROR and not h.value
        devs = (ctypes.c_int * 1)(-1)
        assert L.hvk_group_open(ctypes.byref(h), ctypes.byref(c), 16000000, 0, devs, 1, 1) == H.HVK_ERROR and not h.value
    c.struct_size = ctypes.sizeof(H.HvkConfig)
    with H.Engine(c, 16000000, device=-1) as e:
        info = H.HvkInfo()
        info.struct_size = ctypes.sizeof(info) - 4
        info.width = -7
        assert L.hvk_get_info(e.h, ctypes.byref(info)) == H.HVK_ERROR and info.width == -7
        info.struct_size = ctypes.sizeof(info)
        assert L.hvk_get_info(e.h, ctypes.byref(info)) == H.HVK_OK and info.width == 1024


def test_rccl_loads_and_has_the_gathers_entry_points():
    """hvk_rccl_probe(): librccl.so.1 is found the way hvk_group_gather() finds it and all seven entry points the grouped
    ncclSend / ncclRecv reassembly calls are bound (the container and the GPU box carry /opt/rocm/lib/librccl.so.1). No
    device needed: communicators are made at the first gather between distinct devices."""
    from hacktv_amd.engine import rccl_probe
    code, msg = rccl_probe()
    assert code == H.HVK_OK, msg
    assert msg.startswith("rccl ") and int(msg.split()[1].rstrip(":")) >= 20000
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGetErrorString", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
        assert name in msg


def test_groups_refuse_what_they_cannot_cut_into_blocks(capfd):
    """For N > 1: --interlace (a picture per field; one engine takes it) and the streams that are one serial chain over every sample --
    FM video, sound-in-syncs (its burst encoder keeps the sound chains ahead of the requests: the state cannot be handed
    on) -- are refused when the group is opened, with a line that says why; SECAM colour is not (its state between two frames
    is 40 bytes and travels with the blocks); host tables only, no device."""
    def try_open(conf, sr, devices):
        h = ctypes.c_void_p()
        devs = (ctypes.c_int * len(devices))(*devices)
        r = H.lib().hvk_group_open(ctypes.byref(h), ctypes.byref(conf), sr, 0, devs, len(devices), 2)
        if r == H.HVK_OK:
            H.lib().hvk_group_close(h)
        return r

    c = H.preset("i", H.FLAG_FILTER)
    assert try_open(c, 16000000, [-1, -1]) == H.HVK_OK
    c.interlace = 1
    assert try_open(c, 16000000, [-1]) == H.HVK_OK and try_open(c, 16000000, [-1, -1]) == H.HVK_UNSUPPORTED
    assert "--interlace" in capfd.readouterr().err
    c = H.preset("i", H.FLAG_FILTER)
    c.sis = 1
    assert try_open(c, 16000000, [-1]) == H.HVK_OK
    assert try_open(c, 16000000, [-1, -1]) == H.HVK_UNSUPPORTED
    assert "sound-in-syncs" in capfd.readouterr().err
    assert try_open(H.preset("pal-fm", 0), 14000000, [-1, -1]) == H.HVK_UNSUPPORTED
    assert try_open(H.preset("l", 0), 16000000, [-1, -1]) == H.HVK_OK
