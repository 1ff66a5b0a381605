"""ctypes wrapper around oracle/liboracle.so -- the CPU restatement of the
reference's composite-video -> IQ path (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_void_p, C.c_uint]
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_table.restype = C.c_long
        L.orc_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        L.orc_set_ghost.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_set_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_set_audio.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int]
        L.orc_set_frame2.restype = None
        L.orc_set_frame2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_teletext_packets.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_uint32]
        L.orc_open_rates.restype = C.c_void_p
        L.orc_open_rates.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        L.orc_last_widths.restype = C.c_long
        L.orc_last_widths.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_set_frame_aspect.restype = None
        L.orc_set_frame_aspect.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong]
        L.orc_set_rawbb.restype = None
        L.orc_set_rawbb.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_set_cc608.restype = None
        L.orc_set_cc608.argtypes = [C.c_void_p, C.c_long, C.c_uint8, C.c_uint8]
        L.orc_set_sis_heap.restype = None
        L.orc_set_sis_heap.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_sis_visible.restype = None
        L.orc_set_sis_visible.argtypes = [C.c_void_p, C.c_int]
        L.orc_sis_bursts.restype = C.c_long
        L.orc_sis_bursts.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p]
        L.orc_set_passthru.restype = None
        L.orc_set_passthru.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_render_lines.restype = C.c_long
        L.orc_render_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_last_raster.restype = C.c_long
        L.orc_last_raster.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_last_carrier.restype = C.c_long
        L.orc_last_carrier.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_sink_convert.restype = C.c_long
        L.orc_sink_convert.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


SINK_TYPES = {"uint8": (0, np.uint8), "int8": (1, np.int8), "uint16": (2, np.uint16),
              "int16": (3, np.int16), "int32": (4, np.int32), "float": (5, np.float32)}


def sink_convert(iq, type_name, complex_out):
    """rf_file.c's sample-format conversion (oracle/oracle_sink.c)."""
    iq = np.ascontiguousarray(iq, np.int16)
    code, dt = SINK_TYPES[type_name]
    n = iq.shape[0]
    out = np.zeros(n * (2 if complex_out else 1), dt)
    r = lib().orc_sink_convert(iq.ctypes.data, n, code, 1 if complex_out else 0, out.ctypes.data)
    assert r == out.nbytes
    return out


INFO_NAMES = [
    "width", "half_width", "active_width", "active_left", "lines", "active_lines",
    "white_level", "black_level", "blanking_level", "sync_level",
    "colour_lookup_width", "burst_left", "burst_width", "burst_phase_i", "burst_phase_q",
    "chroma_ataps", "olines", "max_width",
    "fm_mono_level", "nicam_ntaps", "nicam_sps", "nicam_dsl", "nicam_decimation", "nicam_cc_len",
    "am_mono_level", "am_mono_delta_i", "am_mono_delta_q",
]


class Oracle:
    def __init__(self, conf, sample_rate, pixel_rate=0):
        self.conf = conf
        self.p = lib().orc_open_rates(C.addressof(conf), sample_rate, pixel_rate)
        if not self.p:
            raise RuntimeError("orc_open failed")
        v = np.zeros(64, np.int32)
        n = lib().orc_info(self.p, v.ctypes.data, 64)
        self.info = dict(zip(INFO_NAMES, v[:n].tolist()))
        self._keep = []

    def table(self, name, dtype):
        n = lib().orc_table(self.p, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        a = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        if n:
            lib().orc_table(self.p, name.encode(), a.ctypes.data, n)
        return a

    def set_ghost(self, ghost):
        g = np.ascontiguousarray(ghost, np.int16)
        lib().orc_set_ghost(self.p, g.ctypes.data, len(g))

    def set_frame(self, fb, interlaced=0):
        if fb is None:
            # an empty 0 x 0 frame, what av_read_video() hands out past the end (src/av.c:55-59)
            lib().orc_set_frame(self.p, None, 0, 0, 0, 0, 0)
            return
        fb = np.ascontiguousarray(fb, np.uint32)
        self._keep.append(fb)
        h, w = fb.shape
        lib().orc_set_frame(self.p, fb.ctypes.data, w, h, 1, w, interlaced)

    def set_frame2(self, fb, interlaced=0):
        fb = np.ascontiguousarray(fb, np.uint32)
        self._keep.append(fb)
        h, w = fb.shape
        lib().orc_set_frame2(self.p, fb.ctypes.data, w, h, 1, w, interlaced)

    def set_audio(self, stereo, loop=True):
        a = np.ascontiguousarray(stereo, np.int16)
        self._keep.append(a)
        lib().orc_set_audio(self.p, a.ctypes.data, a.shape[0], 1 if loop else 0)

    def set_frame_aspect(self, num, den):
        lib().orc_set_frame_aspect(self.p, num, den)

    def set_rawbb(self, samples):
        a = np.ascontiguousarray(samples, np.int16)
        self._keep.append(a)
        lib().orc_set_rawbb(self.p, a.ctypes.data, a.shape[0])

    def set_cc608(self, frame_index, c1, c2):
        lib().orc_set_cc608(self.p, frame_index, c1, c2)

    def set_sis_heap(self, h8):
        """What lies in front of the reference's sound-in-syncs symbol table on ITS heap (8 samples; the stream's first
        samples show it, oracle_sis.c)."""
        a = np.ascontiguousarray(h8, np.int16)
        assert a.shape == (8,)
        lib().orc_set_sis_heap(self.p, a.ctypes.data)

    def set_sis_visible(self, samples):
        """--sis: samples of a step's audio line the reference's audio thread is taken to have behind it when the SiS
        process picks its block (oracle_sis.c); 0: none."""
        lib().orc_set_sis_visible(self.p, samples)

    def sis_bursts(self, first_line, nlines):
        out = np.zeros((nlines, 8), np.uint8)
        assert lib().orc_sis_bursts(self.p, first_line, nlines, out.ctypes.data) == nlines
        return out

    def set_passthru(self, iq):
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2)
        self._keep.append(a)
        lib().orc_set_passthru(self.p, a.ctypes.data, a.shape[0])

    def teletext_packets(self, frame_index, packets, mask):
        p = np.ascontiguousarray(packets, np.uint8)
        assert p.shape == (32, 45)
        if lib().orc_teletext_packets(self.p, frame_index, p.ctypes.data, mask) != 0:
            raise RuntimeError("orc_teletext_packets failed")

    def render_lines(self, nlines):
        w = self.info["max_width"]
        buf = np.zeros(nlines * w * 2, np.int16)
        got = lib().orc_render_lines(self.p, buf.ctypes.data, nlines)
        return buf[: got * 2].reshape(got, 2)

    def last_widths(self):
        n = lib().orc_last_widths(self.p, None, 0)
        a = np.zeros(n, np.int32)
        lib().orc_last_widths(self.p, a.ctypes.data, n)
        return a

    def last_raster(self):
        n = lib().orc_last_raster(self.p, None, 0)
        a = np.zeros(n, np.int16)
        lib().orc_last_raster(self.p, a.ctypes.data, n)
        return a

    def last_carrier(self):
        n = lib().orc_last_carrier(self.p, None, 0)
        a = np.zeros(n * 2, np.int16)
        lib().orc_last_carrier(self.p, a.ctypes.data, n)
        return a.reshape(n, 2)

    def close(self):
        if self.p:
            lib().orc_close(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
