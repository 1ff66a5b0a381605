"""Live comparison of the oracle with the unmodified reference (oracle/_ref,
built by oracle/Makefile where /root/reference exists). Skipped where the
reference build is absent; tests/test_oracle_golden.py covers that case
through committed digests."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import refprobe
import util

pytestmark = pytest.mark.skipif(not refprobe.available() or not os.path.exists(refprobe.BIN_PATH),
                                reason="oracle/_ref not built")


def _cli(mode, sr, flags, nbytes):
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
    return np.frombuffer(bytes(out), np.int16)


def test_oracle_tables_equal_the_probe(golden):
    """Every table, entry for entry, against the tables vid_init() built in-process."""
    case = "i_full"
    conf, sr = golden.conf(case)
    with refprobe.RefProbe("i", sr, refprobe.FLAG_FILTER) as r, oracle.Oracle(conf, sr) as o:
        for name, dt in util.TABLE_DTYPES.items():
            if name == "chroma_ghost":
                continue
            assert np.array_equal(r.table(name, dt), o.table(name, dt)), name


def test_ghost_samples_follow_the_heap(golden):
    """the reference's colour-line tails depend on what follows its
    chrominance buffer on the heap. In-process (a different heap from the CLI's) the
    probe sees other values; feeding THOSE to the oracle reproduces the in-process
    reference, which shows the ghost input is the right abstraction."""
    conf, sr = golden.conf("i_raster")
    with refprobe.RefProbe("i", sr, refprobe.FLAG_NOAUDIO) as r:
        ghost = r.table("chroma_ghost", np.int16)
        ref = r.render_lines(40)
    with oracle.Oracle(conf, sr) as o:
        o.set_ghost(ghost)
        o.set_frame(golden.frame("i_raster"))
        mine = o.render_lines(40)
    assert np.array_equal(ref, mine)


@pytest.mark.parametrize("mode,sr,flags,pflags", [
    ("i", 16000000, ["--filter"], refprobe.FLAG_FILTER),
    ("m", 13500000, ["--filter"], refprobe.FLAG_FILTER),
])
def test_oracle_long_run_equals_cli(golden, mode, sr, flags, pflags):
    """5 frames, crossing several FM re-normalisations and >200 NICAM frames."""
    import hacktv_amd as H
    conf = H.preset(mode, pflags)
    with oracle.Oracle(conf, sr) as o:
        key = "frame_%dx%d" % (o.info["active_width"], o.info["active_lines"])
        o.set_frame(golden.src[key])
        o.set_audio(golden.audio, True)
        iq = o.render_lines(o.info["lines"] * 5)
    ref = _cli(mode, sr, flags, iq.size * 2)
    assert np.array_equal(ref, iq.reshape(-1))
