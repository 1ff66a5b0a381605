"""ctypes wrapper around oracle/_ref/libhacktv_ref.so (TEST INFRASTRUCTURE).

The shared object is the UNMODIFIED reference engine plus oracle/ref_probe.c,
built by oracle/Makefile where /root/reference exists. Tests that use it skip
when it has not been built.
"""
import ctypes
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HVK_REF_LIB") or os.path.join(ROOT, "oracle", "_ref", "libhacktv_ref.so")      # (HVK_REF_LIB: another build of it, e.g. one with -fsanitize=address)
BIN_PATH = os.path.join(ROOT, "oracle", "_ref", "hacktv_ref")

FLAG_FILTER, FLAG_NOAUDIO, FLAG_NONICAM, FLAG_NOCOLOUR = 1, 2, 4, 8
FLAG_INTERLACE, FLAG_A2STEREO, FLAG_CC608, FLAG_WSS_AUTO, FLAG_ACP, FLAG_VITS, FLAG_VITC = 16, 32, 64, 128, 256, 512, 1024
FLAG_SVIDEO, FLAG_SECAM_FID, FLAG_SIS = 2048, 4096, 8192

INFO_NAMES = [
    "width", "half_width", "active_width", "active_left", "lines", "active_lines",
    "white_level", "black_level", "blanking_level", "sync_level",
    "colour_lookup_width", "burst_left", "burst_width", "burst_phase_i", "burst_phase_q",
    "chroma_ataps", "olines", "max_width",
    "fm_mono_level", "nicam_ntaps", "nicam_sps", "nicam_dsl", "nicam_decimation", "nicam_cc_len",
    "am_mono_level", "am_mono_delta_i", "am_mono_delta_q",
    "fm_secam_level", "secam_dmin0", "secam_dmax0", "secam_dmin1", "secam_dmax1",
    "secam_fsync_level", "secam_field_id_lines",
]

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        L.ref_open.restype = ctypes.c_void_p
        L.ref_open.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_char_p]
        L.ref_close.argtypes = [ctypes.c_void_p]
        L.ref_override.restype = None
        L.ref_override.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.ref_override2.restype = None
        L.ref_override2.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.ref_override_files.restype = None
        L.ref_override_files.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        L.ref_pin_ghost.restype = None
        L.ref_pin_ghost.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ref_set_source.restype = None
        L.ref_set_source.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        L.ref_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ref_pipeline_depths.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.ref_render_lines.restype = ctypes.c_long
        L.ref_render_lines.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        L.ref_table.restype = ctypes.c_long
        L.ref_table.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
        L.ref_test_frame.restype = ctypes.c_long
        L.ref_test_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        L.ref_test_audio.restype = ctypes.c_long
        L.ref_test_audio.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        _lib = L
    return _lib


class RefProbe:
    def __init__(self, mode, sample_rate, flags=0, pixel_rate=0, teletext=None, gamma=0.0, level=0.0, invert=0, volume=0,
                 offset=0, swap_iq=0, wss=None, fid_lines=0, raw_bb=None, raw_bb_levels=(0, 32767), passthru=None):
        if gamma or level or invert or volume:
            lib().ref_override(gamma, level, invert, volume)
        if offset or swap_iq or wss or fid_lines:
            lib().ref_override2(offset, swap_iq, wss.encode() if wss else None, fid_lines)
        if raw_bb or passthru:
            lib().ref_override_files(raw_bb.encode() if raw_bb else None, int(raw_bb_levels[0]), int(raw_bb_levels[1]), passthru.encode() if passthru else None)
        self.p = lib().ref_open(mode.encode(), sample_rate, pixel_rate, flags,
                                teletext.encode() if teletext else None)
        if not self.p:
            raise RuntimeError("ref_open failed for mode %r" % mode)
        v = np.zeros(64, np.int32)
        n = lib().ref_info(self.p, v.ctypes.data, 64)
        self.info = dict(zip(INFO_NAMES, v[:n].tolist()))

    def set_source(self, frames, audio, interlaced=0, par=(1, 1), cc=None, blank=0):
        """Replace the test source: frames [n][h][w] RGBx shown in turn, audio [m][2] looped; `blank`: bit f set = the stream's frame f has no picture."""
        lib().ref_blank_frames(blank)
        f = np.ascontiguousarray(frames, np.uint32)
        a = np.ascontiguousarray(audio, np.int16)
        c = np.ascontiguousarray(cc, np.uint8) if cc is not None else None
        self._keep = (f, a, c)
        lib().ref_set_source(self.p, f.ctypes.data, f.shape[0], f.shape[2], f.shape[1], interlaced, par[0], par[1],
                             c.ctypes.data if c is not None else None, a.ctypes.data, a.shape[0])

    def pin_ghost(self, ghost):
        """Keep the samples the chroma filter reads past its buffer what `ghost` says (oracle/ref_probe.c:ref_pin_ghost)."""
        g = np.ascontiguousarray(ghost, np.int16)
        lib().ref_pin_ghost(self.p, g.ctypes.data, min(len(g), 64))

    def pipeline_depths(self):
        """(reference, shim): lines held back by the reference's line pipeline, and the shim's count of them."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        lib().ref_pipeline_depths(self.p, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def table(self, name, dtype):
        n = lib().ref_table(self.p, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        a = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        if n:
            lib().ref_table(self.p, name.encode(), a.ctypes.data, n)
        return a

    def test_frame(self):
        n = lib().ref_test_frame(self.p, None, 0)
        a = np.zeros(n, np.uint32)
        lib().ref_test_frame(self.p, a.ctypes.data, n)
        return a.reshape(self.info["active_lines"], self.info["active_width"])

    def test_audio(self):
        n = lib().ref_test_audio(self.p, None, 0)
        a = np.zeros(n * 2, np.int16)
        lib().ref_test_audio(self.p, a.ctypes.data, n)
        return a.reshape(n, 2)

    def render_lines(self, nlines):
        w = self.info["max_width"]
        buf = np.zeros(nlines * w * 2, np.int16)
        got = lib().ref_render_lines(self.p, buf.ctypes.data, nlines)
        return buf[: got * 2].reshape(got, 2)

    def close(self):
        if self.p:
            lib().ref_close(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
