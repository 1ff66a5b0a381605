"""BASELINE config 5 on one device (tools/hour.py): the metric configuration over ONE HOUR of signal -- 90 000 frames in 704
blocks -- every block's samples summed on the device (hvk_block_sums) and compared with the reference CLI's sums
(tests/golden/ref_hour.json, oracle/make_golden_hour.py), the blocks around frames 0, 9 000, 45 000 and 90 000 hashed.
--noaudio runs the whole hour (a few seconds: nothing serial on the host); with sound the host's FM chain takes two minutes
for the hour, so the test walks the first 48 blocks (6 144 frames, 4 minutes of signal, 120 000 re-normalisations of the
phasor) and bench.py --hour-sound the whole of it (profiles/r04_hour.json)."""
import os
import sys

import numpy as np
import pytest

import hacktv_amd as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hour  # noqa: E402


def test_block_sums_equal_the_oracles(golden):
    """hvk_block_sums against oracle_sink.c:orc_block_sums on one rendered block (and on a run that is no multiple of 4 words)."""
    import ctypes
    import oracle
    conf, sr = golden.conf("i_full")
    with H.Engine(conf, sr, device=0, max_frames=2) as e:
        e.frame_upload(0, golden.frame("i_full"))
        while e.audio_needed(2) > 0:
            e.audio_write(golden.audio)
        e.render(2)
        fs = e.info["frame_samples"]
        iq = e.fetch(0, 2 * fs)
        for first, count in ((0, 2 * fs), (3, fs + 5), (fs - 1, 7)):
            w = np.ascontiguousarray(iq[first:first + count]).view(np.uint32).reshape(-1)
            out = (ctypes.c_uint64 * 2)()
            oracle.lib().orc_block_sums(w.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(w.size), out)
            assert e.block_sums(first, count) == (int(out[0]), int(out[1]))


def test_one_hour_without_sound(golden):
    if hour.golden(False) is None:
        pytest.skip("tests/golden/ref_hour.json has no i_hour_noaudio case")
    res = hour.run(H, golden.frame("i_full"), golden.audio, sound=False)
    assert res["frames"] == 90000 and res["blocks"] == 704


def test_four_minutes_with_sound(golden):
    if hour.golden(True) is None:
        pytest.skip("tests/golden/ref_hour.json has no i_hour case")
    res = hour.run(H, golden.frame("i_full"), golden.audio, sound=True, max_blocks=48)
    assert res["frames"] == 48 * 128
