"""ctypes mirrors of the plain-data structs in include/hvk_config.h and
include/hacktv_amd.h. Used by the Python binding (hacktv_amd.engine) and by the
test harness for the oracle, which takes the same hvk_config_t."""
import ctypes as C


class HvkRational(C.Structure):
    _fields_ = [("num", C.c_int64), ("den", C.c_int64)]


class HvkConfig(C.Structure):
    # field order == include/hvk_config.h
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("output_type", C.c_int),
        ("modulation", C.c_int),
        ("video_bw", C.c_double),
        ("vsb_upper_bw", C.c_double),
        ("vsb_lower_bw", C.c_double),
        ("level", C.c_double),
        ("video_level", C.c_double),
        ("fm_mono_level", C.c_double),
        ("am_audio_level", C.c_double),
        ("nicam_level", C.c_double),
        ("type", C.c_int),
        ("frame_rate", HvkRational),
        ("lines", C.c_int),
        ("hline", C.c_int),
        ("interlaced", C.c_int),
        ("interlace", C.c_int),
        ("active_lines", C.c_int),
        ("hsync_width", C.c_double),
        ("vsync_short_width", C.c_double),
        ("vsync_long_width", C.c_double),
        ("sync_rise", C.c_double),
        ("invert_video", C.c_int),
        ("white_level", C.c_double),
        ("black_level", C.c_double),
        ("blanking_level", C.c_double),
        ("sync_level", C.c_double),
        ("active_width", C.c_double),
        ("active_left", C.c_double),
        ("gamma", C.c_double),
        ("rw_co", C.c_double),
        ("gw_co", C.c_double),
        ("bw_co", C.c_double),
        ("colour_mode", C.c_int),
        ("colour_carrier", HvkRational),
        ("colour_bw", C.c_double),
        ("burst_width", C.c_double),
        ("burst_left", C.c_double),
        ("burst_level", C.c_double),
        ("burst_rise", C.c_double),
        ("ev_co", C.c_double),
        ("eu_co", C.c_double),
        ("secam_field_id", C.c_int),
        ("secam_field_id_lines", C.c_int),
        ("volume", C.c_int),
        ("fm_mono_carrier", C.c_double),
        ("fm_mono_deviation", C.c_double),
        ("fm_mono_preemph", C.c_int),
        ("nicam_carrier", C.c_double),
        ("nicam_beta", C.c_double),
        ("am_mono_carrier", C.c_double),
        ("a2stereo", C.c_int),
        ("vfilter", C.c_int),
        ("raw_bb", C.c_int),
        ("raw_bb_blanking_level", C.c_int),
        ("raw_bb_white_level", C.c_int),
        ("s_video", C.c_int),
        ("teletext", C.c_int),
        ("wss", C.c_int),
        ("vits", C.c_int),
        ("vitc", C.c_int),
        ("acp", C.c_int),
        ("cc608", C.c_int),
        ("sis", C.c_int),
        ("fm_level", C.c_double),
        ("fm_deviation", C.c_double),
        ("swap_iq", C.c_int),
        ("offset", C.c_int64),
        ("passthru", C.c_int),
        ("fsc_flag_width", C.c_double),
        ("fsc_flag_left", C.c_double),
        ("fsc_flag_level", C.c_double),
        ("frame_orientation", C.c_int),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)      # (HVK_CONFIG_INIT)


class HvkInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in (
        "sample_rate", "width", "half_width", "active_width", "active_left",
        "lines", "active_lines",
        "white_level", "black_level", "blanking_level", "sync_level",
        "delay_lines", "frame_samples", "max_frames", "frame_slots",
        "colour_lookup_width", "burst_left", "burst_width",
        "has_carriers", "has_nicam", "pixel_rate", "max_width", "startup_samples")]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "struct_size"}


FLAG_FILTER, FLAG_NOAUDIO, FLAG_NONICAM, FLAG_NOCOLOUR = 1, 2, 4, 8

HVK_OK, HVK_ERROR, HVK_OUT_OF_MEMORY, HVK_NO_DEVICE, HVK_UNSUPPORTED = 0, -1, -2, -3, -4

# hvk_set_levels()
LEVELS_AUTO, LEVELS_TABLE, LEVELS_COMPUTE = 0, 1, 2
