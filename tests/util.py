"""Shared helpers of the test-suite: golden fixtures, case table, hashing."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

TABLE_DTYPES = {
    "syncs": np.int16, "yuv": np.int16, "colour_lookup": np.int16, "burst_win": np.int16,
    "chroma_taps": np.int16, "chroma_ghost": np.int16, "vfilter_itaps": np.int16, "vfilter_qtaps": np.int16,
    "fm_mono_lut": np.int32, "nicam_taps": np.int16, "nicam_cc": np.int16,
    "limiter_shape": np.int16, "limiter_vtaps": np.int32, "limiter_ftaps": np.int32,
    "fm_secam_lut": np.int32, "fm_secam_bell": np.int16, "fm_secam_fir": np.int16, "secam_l_fir": np.int16,
    "teletext_lut": np.int16, "fm_video_lut": np.int32, "resampler_taps": np.int16,
}


def rawbb_signal(nsamples=700 * 1024 + 311):
    """The external baseband stream of the --raw-bb-file cases: a fixed integer pattern (levels 2000 ..
    21000, with excursions below and above), a little more than one PAL frame, not a whole number of
    lines -- so the reference's rewind falls inside a line."""
    n = np.arange(nsamples, dtype=np.int64)
    v = 2000 + (n * 37) % 19001
    v[(n % 1024) < 75] = 300          # something like a sync tip
    v[(n % 4099) == 0] = 32000
    return v.astype(np.int16)


def passthru_signal(nsamples=1600300):
    """The external I/Q signal of the --passthru cases: a fixed integer pattern (int16 pairs), 2.5 PAL
    frames and a bit long so that runs end inside it -- in the middle of a line."""
    n = np.arange(nsamples, dtype=np.int64)
    iq = np.empty((nsamples, 2), np.int16)
    iq[:, 0] = (n * 7919) % 4001 - 2000
    iq[:, 1] = (n * 104729) % 3001 - 1500
    return iq


class Golden:
    """tests/golden/: outputs of the unmodified reference (oracle/make_golden.py)."""

    def __init__(self):
        with open(os.path.join(GOLD, "ref_digests.json")) as f:
            self.cases = json.load(f)
        self.sink_formats = self.cases.pop("_sink_formats", {})
        self.src = np.load(os.path.join(GOLD, "testsrc.npz"))
        self.lines = np.load(os.path.join(GOLD, "ref_lines.npz"))

    def frame(self, case):
        """The picture the raster shows: the test source's (for the systems that scan vertically turned the way the
        reference turns every picture it reads: oracle/make_golden_rasters.py)."""
        c = self.cases[case]
        i = c["info"]
        return self.src[c.get("frame_key") or "frame_%dx%d" % (i["active_width"], i["active_lines"])]

    @property
    def audio(self):
        return self.src["audio"]

    def conf(self, case):
        import hacktv_amd as H
        c = self.cases[case]
        conf = H.preset(c["mode"], c["probe_flags"])
        conf.teletext = 1 if c.get("teletext") else 0
        for k, v in c.get("extra", {}).items():
            setattr(conf, k, v)
        return conf, c["sample_rate"]

    def teletext_rows(self, frame, skip=()):
        """The packets the reference's raw: source yields for a frame from tests/golden/ttraw.bin:
        55 55 27 + the next 42-byte record (src/teletext.c:1188-1201), one per teletext line
        (rows 0..15 lines 7..22, rows 16..31 lines 320..335). Rows in `skip` are lines another
        inserter holds: teletext leaves them alone and keeps the packet for the next line
        (src/teletext.c:1219). Returns (rows, mask)."""
        rec = np.frombuffer(open(os.path.join(GOLD, "ttraw.bin"), "rb").read(), np.uint8).reshape(-1, 42)
        use = [r for r in range(32) if r not in skip]
        p = np.zeros((32, 45), np.uint8)
        p[:, 0] = 0x55
        p[:, 1] = 0x55
        p[:, 2] = 0x27
        p[use, 3:] = rec[frame * len(use):(frame + 1) * len(use)]
        mask = 0
        for r in use:
            mask |= 1 << r
        return p, mask

    def teletext_skip(self, case):
        """Teletext rows (625 lines) held by the case's other inserters: VITS 17/18/330/331, VITC
        19/21/332/334, ACP 9-18/321-330, CC608 22."""
        x = self.cases[case].get("extra", {})
        lines = ([17, 18, 330, 331] if x.get("vits") else []) + ([19, 21, 332, 334] if x.get("vitc") else [])
        lines += (list(range(9, 19)) + list(range(321, 331)) if x.get("acp") else []) + ([22] if x.get("cc608") else [])
        return tuple(sorted(set(l - 7 if l < 300 else 16 + l - 320 for l in lines)))

    def cli_flags(self, case, passfile="/tmp/hvk_passthru.bin"):
        """The reference CLI's flags for the case; passthru cases expect passthru_signal() at `passfile`."""
        return [f.replace("@TTRAW@", os.path.join(GOLD, "ttraw.bin")).replace("@PASS@", passfile).replace("@RAWBB@", "/tmp/hvk_rawbb.bin")
                for f in self.cases[case]["cli_flags"]]


def stream_bytes(iq, real):
    """The bytes the reference's file sink writes for these samples
    (src/rf_file.c: int16 real writes I only, int16 complex writes pairs)."""
    iq = np.ascontiguousarray(iq, np.int16)
    return (iq[:, 0].copy() if real else iq).tobytes()


def cum_sha(iq, end, case):
    """sha256 of the stream's first `end` samples as the golden file has it: the file sink's bytes, from sample
    case["skip_samples"] on where the reference's own runs differ before that (oracle/make_golden_r06.py)."""
    return sha256(stream_bytes(iq[case.get("skip_samples", 0): end], case["real"]))


def sha256(b):
    return hashlib.sha256(b).hexdigest()
