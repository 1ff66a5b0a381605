"""The C-ABI library loads and exports every entry point include/hacktv_amd.h
declares (no device needed, nothing is computed)."""
import ctypes
import os
import re

import hacktv_amd as H

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
    assert ids == ["i", "b", "g", "pal", "l", "secam", "m", "ntsc", "pal-fm", "secam-fm", "ntsc-fm"]
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
    c = H.preset("l"); c.secam_field_id = 1; bad.append((c, 16000000))        # SECAM field identification lines
    c = H.preset("i"); c.fm_mono_preemph = 3; bad.append((c, 16000000))       # J.17 FM pre-emphasis
    c = H.preset("pal-fm", H.FLAG_FILTER); bad.append((c, 16000000))          # FM video with the fixed pre-emphasis tap tables
    c = H.preset("i"); c.type = 2; bad.append((c, 16000000))                  # a raster other than 625 / 525
    for conf, sr in bad:
        try:
            H.Engine(conf, sr, device=-1)
        except H.HvkError as err:
            assert err.code == -4
        else:
            raise AssertionError("an unsupported configuration was accepted")
