"""ctypes binding of libhvk.so (include/hacktv_amd.h).

The class mirrors the reference's engine interface at frame granularity:
Engine(...) is vid_init(), render() is a batch of vid_next_line() calls,
close() is vid_free() (src/video.h:510-516). There is no Python or CPU
implementation behind it: if the shared library or the HIP device is missing
the calls raise."""
import ctypes as C
import os
import numpy as np

from .ctypes_defs import HvkConfig, HvkInfo, HVK_OUT_OF_MEMORY

# HVK_LIB: another build of the library (tools/ablate.py uses one with the profiling switches compiled in)
LIB_PATH = os.environ.get("HVK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhvk.so")

SYMBOLS = [
    "hvk_config_preset", "hvk_config_apply_flags", "hvk_preset_id", "hvk_preset_desc",
    "hvk_open", "hvk_open_rates", "hvk_line_widths", "hvk_frame_start", "hvk_close", "hvk_get_info", "hvk_get_framebuffer_length",
    "hvk_set_chroma_ghost", "hvk_get_chroma_ghost", "hvk_frame_upload", "hvk_teletext_packets", "hvk_audio_write",
    "hvk_passthru_write", "hvk_host_offset_stream", "hvk_host_fm_video", "hvk_cc608_write", "hvk_frame_aspect", "hvk_rawbb_write",
    "hvk_audio_needed", "hvk_render", "hvk_render_strided", "hvk_stage_strided", "hvk_stage_strided_prev", "hvk_launch",
    "hvk_launch_strided_out", "hvk_set_stream", "hvk_set_levels", "hvk_planes_refresh",
    "hvk_host_sis_bursts", "hvk_sound_state_size", "hvk_sound_state_export", "hvk_sound_state_import", "hvk_sound_samples_generated",
    "hvk_host_side_streams", "hvk_host_secam_stream", "hvk_secam_stats", "hvk_secam_warmup_lines", "hvk_vbi_lines_held", "hvk_sync", "hvk_fetch", "hvk_fetch_async", "hvk_fetch_wait", "hvk_host_alloc", "hvk_host_free", "hvk_frame_upload_pinned", "hvk_fetch_as", "hvk_output_device_ptr",
    "hvk_timing_enable", "hvk_timing_read", "hvk_kernel_names", "hvk_table", "hvk_fetch_raster", "hvk_version",
    "hvk_group_open", "hvk_group_close", "hvk_group_size", "hvk_group_block_frames", "hvk_group_engine", "hvk_group_block_engine", "hvk_group_block_index",
    "hvk_group_next_frame", "hvk_group_frame_upload", "hvk_group_audio_write", "hvk_group_audio_needed", "hvk_group_stage", "hvk_group_launch",
    "hvk_group_gather", "hvk_group_gather_backend", "hvk_engine_stream", "hvk_last_line_shows_picture", "hvk_stream_is_one_chain", "hvk_block_sums", "hvk_fused_launches", "hvk_secam_estimated_stages", "hvk_levels_short_form",
    "hvk_sound_source_end", "hvk_secam_kept", "hvk_secam_state_size", "hvk_secam_state_export", "hvk_secam_state_import", "hvk_frame_copy", "hvk_rccl_probe", "hvk_secam_walk_stages", "hvk_teletext_packets_block", "hvk_kernel_plan",
]

_lib = None


class HvkError(RuntimeError):
    def __init__(self, what, code):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


def lib():
    """Load libhvk.so (built by __graft_entry__.build() / hacktv_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.hvk_config_preset.argtypes = [vp, C.c_char_p]
        L.hvk_config_apply_flags.argtypes = [vp, i32]
        L.hvk_config_apply_flags.restype = None
        L.hvk_preset_id.restype = C.c_char_p
        L.hvk_preset_id.argtypes = [i32]
        L.hvk_preset_desc.restype = C.c_char_p
        L.hvk_preset_desc.argtypes = [i32]
        L.hvk_open.argtypes = [C.POINTER(vp), vp, C.c_uint, i32, i32]
        L.hvk_open_rates.argtypes = [C.POINTER(vp), vp, C.c_uint, C.c_uint, i32, i32]
        L.hvk_line_widths.argtypes = [vp, i64, i32, vp]
        L.hvk_close.argtypes = [vp]
        L.hvk_close.restype = None
        L.hvk_get_info.argtypes = [vp, vp]
        L.hvk_get_framebuffer_length.argtypes = [vp]
        L.hvk_get_framebuffer_length.restype = C.c_size_t
        L.hvk_set_chroma_ghost.argtypes = [vp, vp, i32]
        L.hvk_set_levels.argtypes = [vp, i32]
        L.hvk_get_chroma_ghost.argtypes = [vp, vp, i32]
        L.hvk_frame_upload.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32]
        L.hvk_teletext_packets.argtypes = [vp, i32, vp, C.c_uint32]
        L.hvk_audio_write.argtypes = [vp, vp, C.c_size_t]
        L.hvk_passthru_write.argtypes = [vp, vp, C.c_size_t]
        L.hvk_cc608_write.argtypes = [vp, i32, C.c_uint8, C.c_uint8]
        L.hvk_frame_aspect.argtypes = [vp, i32, i64, i64]
        L.hvk_rawbb_write.argtypes = [vp, vp, C.c_size_t]
        L.hvk_host_offset_stream.argtypes = [vp, i64, i64, vp]
        L.hvk_host_fm_video.argtypes = [vp, vp, i64]
        L.hvk_audio_needed.argtypes = [vp, i32]
        L.hvk_audio_needed.restype = C.c_size_t
        L.hvk_render.argtypes = [vp, i32, vp, vp]
        L.hvk_render_strided.argtypes = [vp, i64, i64, i32, vp, vp]
        L.hvk_stage_strided.argtypes = [vp, i64, i64, i32, vp]
        L.hvk_stage_strided_prev.argtypes = [vp, i64, i64, i32, vp, vp]
        L.hvk_launch.argtypes = [vp, vp]
        L.hvk_launch_strided_out.argtypes = [vp, vp, i64]
        L.hvk_set_stream.argtypes = [vp, vp]
        L.hvk_host_side_streams.argtypes = [vp, i64, i64, vp, vp, i32, vp]
        L.hvk_host_secam_stream.argtypes = [vp, vp, i32, i32, i32, vp]
        L.hvk_secam_stats.argtypes = [vp, vp]
        L.hvk_secam_estimated_stages.argtypes = [vp]
        L.hvk_secam_estimated_stages.restype = C.c_int64
        L.hvk_frame_start.argtypes = [vp, C.c_int64]
        L.hvk_frame_start.restype = C.c_int64
        L.hvk_secam_warmup_lines.argtypes = [vp]
        L.hvk_vbi_lines_held.argtypes = [vp, vp, i32]
        L.hvk_sync.argtypes = [vp]
        L.hvk_fetch.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.hvk_fetch_async.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.hvk_fetch_wait.argtypes = [vp, i32]
        L.hvk_frame_upload_pinned.argtypes = [vp, i32, vp, i32, i32, i32]
        L.hvk_host_alloc.argtypes = [vp, C.c_size_t]
        L.hvk_host_alloc.restype = vp
        L.hvk_host_free.argtypes = [vp, vp]
        L.hvk_host_free.restype = None
        L.hvk_fetch_as.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i32, i32]
        L.hvk_fetch_as.restype = C.c_long
        L.hvk_output_device_ptr.argtypes = [vp]
        L.hvk_output_device_ptr.restype = vp
        L.hvk_timing_enable.argtypes = [vp, i32]
        L.hvk_planes_refresh.argtypes = [vp, vp, i32]
        L.hvk_host_sis_bursts.argtypes = [vp, C.c_int64, i32, vp]
        L.hvk_sound_state_size.argtypes = [vp]
        L.hvk_sound_state_size.restype = C.c_size_t
        L.hvk_sound_state_export.argtypes = [vp, vp, C.c_size_t]
        L.hvk_sound_state_import.argtypes = [vp, vp, C.c_size_t, vp]
        L.hvk_sound_samples_generated.argtypes = [vp]
        L.hvk_sound_samples_generated.restype = C.c_int64
        L.hvk_timing_read.argtypes = [vp, i32, vp, vp]
        L.hvk_kernel_names.argtypes = [vp, C.c_char_p, i32]
        L.hvk_table.argtypes = [vp, C.c_char_p, vp, C.c_long]
        L.hvk_table.restype = C.c_long
        L.hvk_fetch_raster.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.hvk_version.restype = C.c_char_p
        L.hvk_group_open.argtypes = [C.POINTER(vp), vp, C.c_uint, C.c_uint, vp, i32, i32]
        L.hvk_group_close.argtypes = [vp]
        L.hvk_group_close.restype = None
        L.hvk_group_size.argtypes = [vp]
        L.hvk_group_block_frames.argtypes = [vp]
        L.hvk_group_engine.argtypes = [vp, i32]
        L.hvk_group_engine.restype = vp
        L.hvk_group_block_engine.argtypes = [vp]
        L.hvk_group_block_engine.restype = vp
        L.hvk_group_block_index.argtypes = [vp]
        L.hvk_group_next_frame.argtypes = [vp]
        L.hvk_group_next_frame.restype = i64
        L.hvk_group_frame_upload.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32]
        L.hvk_group_audio_write.argtypes = [vp, vp, C.c_size_t]
        L.hvk_group_audio_needed.argtypes = [vp, i32]
        L.hvk_group_audio_needed.restype = C.c_size_t
        L.hvk_group_stage.argtypes = [vp, i32, vp]
        L.hvk_group_launch.argtypes = [vp, vp]
        L.hvk_group_gather.argtypes = [vp, i32, vp, C.c_size_t]
        L.hvk_group_gather_backend.argtypes = [vp]
        L.hvk_group_gather_backend.restype = C.c_char_p
        L.hvk_engine_stream.argtypes = [vp]
        L.hvk_engine_stream.restype = vp
        L.hvk_last_line_shows_picture.argtypes = [vp]
        L.hvk_stream_is_one_chain.argtypes = [vp]
        L.hvk_block_sums.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
        L.hvk_fused_launches.argtypes = [vp]
        L.hvk_fused_launches.restype = i64
        L.hvk_sound_source_end.argtypes = [vp]
        L.hvk_sound_source_end.restype = i64
        L.hvk_frame_copy.argtypes = [vp, i32, vp, i32]
        L.hvk_rccl_probe.argtypes = [C.c_char_p, C.c_size_t]
        L.hvk_secam_walk_stages.argtypes = [vp, vp]
        L.hvk_secam_kept.argtypes = [vp, vp]
        L.hvk_secam_state_size.argtypes = [vp]
        L.hvk_secam_state_size.restype = C.c_size_t
        L.hvk_secam_state_export.argtypes = [vp, vp, C.c_size_t]
        L.hvk_secam_state_import.argtypes = [vp, vp, C.c_size_t]
        L.hvk_teletext_packets_block.argtypes = [vp, i32, i32, vp, vp]
        L.hvk_kernel_plan.argtypes = [vp, C.c_char_p, i32]
        _lib = L
    return _lib


def preset(mode, flags=0):
    """vid_configs[] lookup + the CLI's preset edits (src/hacktv.c:1078-1171)."""
    c = HvkConfig()
    r = lib().hvk_config_preset(C.byref(c), mode.encode())
    if r != 0:
        raise HvkError("hvk_config_preset(%r)" % mode, r)
    lib().hvk_config_apply_flags(C.byref(c), flags)
    return c


def rccl_probe():
    """(code, message): does librccl.so.1 load and hold the seven entry points hvk_group_gather() calls? No device needed."""
    buf = C.create_string_buffer(256)
    r = lib().hvk_rccl_probe(buf, len(buf))
    return r, buf.value.decode()


class Engine:
    def __init__(self, conf, sample_rate, device=0, max_frames=4, pixel_rate=0):
        self.h = C.c_void_p()
        self.conf = conf
        r = lib().hvk_open_rates(C.byref(self.h), C.byref(conf), sample_rate, pixel_rate, device, max_frames)
        if r != 0:
            self.h = None
            raise HvkError("hvk_open", r)
        self._host_bufs = []
        info = HvkInfo()
        lib().hvk_get_info(self.h, C.byref(info))
        self.info = info.as_dict()
        self.device = device

    def _chk(self, what, r):
        if r < 0:
            raise HvkError(what, r)
        return r

    def close(self):
        if self.h:
            for p in self._host_bufs:
                lib().hvk_host_free(self.h, p)
            self._host_bufs = []
            lib().hvk_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def table(self, name, dtype):
        n = lib().hvk_table(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        a = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        if n:
            got = lib().hvk_table(self.h, name.encode(), a.ctypes.data, n)
            if got != n:
                raise HvkError("hvk_table(%s)" % name, got)
        return a

    def set_chroma_ghost(self, ghost):
        g = np.ascontiguousarray(ghost, np.int16)
        self._chk("hvk_set_chroma_ghost", lib().hvk_set_chroma_ghost(self.h, g.ctypes.data, len(g)))

    def chroma_ghost(self):
        g = np.zeros(32, np.int16)
        self._chk("hvk_get_chroma_ghost", lib().hvk_get_chroma_ghost(self.h, g.ctypes.data, 32))
        return g

    def frame_upload(self, slot, fb, interlaced=0):
        if fb is None:
            return self._chk("hvk_frame_upload", lib().hvk_frame_upload(self.h, slot, None, 0, 0, 0, 0, 0))
        fb = np.ascontiguousarray(fb, np.uint32)
        h, w = fb.shape
        return self._chk("hvk_frame_upload", lib().hvk_frame_upload(self.h, slot, fb.ctypes.data, w, h, 1, w, interlaced))

    def teletext_packets(self, frame_in_batch, packets, mask=0xFFFFFFFF):
        p = np.ascontiguousarray(packets, np.uint8)
        assert p.shape == (32, 45)
        return self._chk("hvk_teletext_packets", lib().hvk_teletext_packets(self.h, frame_in_batch, p.ctypes.data, mask))

    def teletext_packets_block(self, first_frame_in_batch, packets, masks):
        """packets (n, 32, 45) uint8, masks (n,) uint32: hvk_teletext_packets() for n frames in one call"""
        p = np.ascontiguousarray(packets, np.uint8)
        m = np.ascontiguousarray(masks, np.uint32)
        assert p.ndim == 3 and p.shape[1:] == (32, 45) and m.shape == (p.shape[0],)
        return self._chk("hvk_teletext_packets_block", lib().hvk_teletext_packets_block(self.h, first_frame_in_batch, p.shape[0], p.ctypes.data, m.ctypes.data))

    def line_widths(self, first_line, nlines):
        w = np.zeros(nlines, np.int32)
        self._chk("hvk_line_widths", lib().hvk_line_widths(self.h, first_line, nlines, w.ctypes.data))
        return w

    def audio_write(self, stereo):
        a = np.ascontiguousarray(stereo, np.int16)
        return self._chk("hvk_audio_write", lib().hvk_audio_write(self.h, a.ctypes.data, a.shape[0]))

    def rawbb_write(self, samples):
        a = np.ascontiguousarray(samples, np.int16)
        return self._chk("hvk_rawbb_write", lib().hvk_rawbb_write(self.h, a.ctypes.data, a.shape[0]))

    def frame_aspect(self, slot, num, den):
        return self._chk("hvk_frame_aspect", lib().hvk_frame_aspect(self.h, slot, num, den))

    def cc608_write(self, frame_in_batch, c1, c2):
        return self._chk("hvk_cc608_write", lib().hvk_cc608_write(self.h, frame_in_batch, c1, c2))

    def passthru_write(self, iq):
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2)
        return self._chk("hvk_passthru_write", lib().hvk_passthru_write(self.h, a.ctypes.data, a.shape[0]))

    def host_offset_stream(self, first, count):
        out = np.zeros((count, 2), np.int16)
        self._chk("hvk_host_offset_stream", lib().hvk_host_offset_stream(self.h, first, count, out.ctypes.data))
        return out

    def host_fm_video(self, iq):
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2).copy()
        self._chk("hvk_host_fm_video", lib().hvk_host_fm_video(self.h, a.ctypes.data, a.shape[0]))
        return a

    def audio_needed(self, nframes):
        return lib().hvk_audio_needed(self.h, nframes)

    def host_side_streams(self, first, count):
        car = np.zeros((count, 2), np.int16)
        sym = np.zeros(count // 16 + 64, np.uint8)
        k0 = C.c_int64(0)
        n = self._chk("hvk_host_side_streams", lib().hvk_host_side_streams(
            self.h, first, count, car.ctypes.data, sym.ctypes.data, len(sym), C.byref(k0)))
        return car, sym[:n], k0.value

    def host_sis_bursts(self, first_line, nlines):
        out = np.zeros((nlines, 8), np.uint8)
        self._chk("hvk_host_sis_bursts", lib().hvk_host_sis_bursts(self.h, first_line, nlines, out.ctypes.data))
        return out

    def host_secam_stream(self, fb, interlaced=0):
        out = np.zeros(self.info["frame_samples"], np.int16)
        if fb is None:
            r = lib().hvk_host_secam_stream(self.h, None, 0, 0, 0, out.ctypes.data)
        else:
            fb = np.ascontiguousarray(fb, np.uint32)
            h, w = fb.shape
            r = lib().hvk_host_secam_stream(self.h, fb.ctypes.data, w, h, interlaced, out.ctypes.data)
        self._chk("hvk_host_secam_stream", r)
        return out

    def vbi_lines_held(self):
        """1-based numbers of the lines the inserters other than teletext write to"""
        n = self.info["lines"]
        held = np.zeros(n, np.uint8)
        self._chk("hvk_vbi_lines_held", lib().hvk_vbi_lines_held(self.h, held.ctypes.data, n))
        return [i + 1 for i in range(n) if held[i]]

    def secam_stats(self):
        c = (C.c_int64 * 4)()
        self._chk("hvk_secam_stats", lib().hvk_secam_stats(self.h, c))
        return dict(zip(("tasks", "mismatches", "redone", "host_frames"), list(c)))

    def kernel_plan(self):
        buf = C.create_string_buffer(4096)
        self._chk("hvk_kernel_plan", lib().hvk_kernel_plan(self.h, buf, len(buf)))
        return buf.value.decode()

    def secam_walk_stages(self):
        """(what hvk_open() allows: 0 / 1 / 2, [stages through hvk_k_secam_chain, hvk_k_secam_walk<0>, hvk_k_secam_walk<1>])"""
        c = (C.c_int64 * 3)()
        ok = lib().hvk_secam_walk_stages(self.h, c)
        return ok, list(c)

    def secam_kept(self):
        """{frames that took a kept sub-carrier set, stages done again without them, picture slots sets are kept for} (hvk_secam_kept)"""
        c = (C.c_int64 * 3)()
        self._chk("hvk_secam_kept", lib().hvk_secam_kept(self.h, c))
        return dict(zip(("frames_taken", "restarts", "slots"), list(c)))

    def levels_short_form(self):
        """1: computed levels use the short arithmetic, checked at open on all 2^24 colours (hvk_levels_short_form)."""
        return int(lib().hvk_levels_short_form(self.h))

    def secam_estimated_stages(self):
        """Stages whose new pictures' lines started from estimated states (hvk_k_secam_est) instead of warm-up walks."""
        return int(lib().hvk_secam_estimated_stages(self.h))

    def frame_start(self, frame):
        """First output sample of a stream frame (frame * frame_samples but for rate pairs with frames of two lengths)."""
        return lib().hvk_frame_start(self.h, frame)

    def secam_warmup_lines(self):
        return self._chk("hvk_secam_warmup_lines", lib().hvk_secam_warmup_lines(self.h))

    def render(self, nframes, slots=None, d_iq=None):
        s = np.ascontiguousarray(slots if slots is not None else np.zeros(nframes * 2), np.int32)
        return self._chk("hvk_render", lib().hvk_render(self.h, nframes, s.ctypes.data, d_iq))

    def stage(self, first_frame, stride, nframes, slots=None, prev_slots=None):
        s = np.ascontiguousarray(slots if slots is not None else np.zeros(nframes * 2), np.int32)
        if prev_slots is not None:
            p = np.ascontiguousarray(prev_slots, np.int32)
            return self._chk("hvk_stage_strided_prev", lib().hvk_stage_strided_prev(self.h, first_frame, stride, nframes, s.ctypes.data, p.ctypes.data))
        return self._chk("hvk_stage_strided", lib().hvk_stage_strided(self.h, first_frame, stride, nframes, s.ctypes.data))

    def launch(self, d_iq=None, out_stride=1):
        if out_stride == 1:
            return self._chk("hvk_launch", lib().hvk_launch(self.h, d_iq))
        return self._chk("hvk_launch_strided_out", lib().hvk_launch_strided_out(self.h, d_iq, out_stride))

    def set_stream(self, hip_stream):
        return self._chk("hvk_set_stream", lib().hvk_set_stream(self.h, hip_stream))

    def sync(self):
        return self._chk("hvk_sync", lib().hvk_sync(self.h))

    def fetch(self, first, count):
        out = np.zeros((count, 2), np.int16)
        self._chk("hvk_fetch", lib().hvk_fetch(self.h, out.ctypes.data, first, count))
        return out

    def host_picture(self, height, width):
        """(height, width) uint32 array over page-locked memory, for frame_upload_pinned()"""
        return self.host_buffer(height * width).view(np.uint32).reshape(height, width)

    def frame_upload_pinned(self, slot, fb, interlaced=0):
        """fb: an array from host_picture(); it must stay unchanged until the copy is through (sync / a later fetch)"""
        h, w = fb.shape
        return self._chk("hvk_frame_upload_pinned", lib().hvk_frame_upload_pinned(self.h, slot, fb.ctypes.data, w, h, interlaced))

    def host_buffer(self, count):
        """(count, 2) int16 array over page-locked memory (hvk_host_alloc). The array is a VIEW of memory that the
        engine's close() releases: it (and host_picture()'s) must not be touched after that."""
        p = lib().hvk_host_alloc(self.h, count * 4)
        if not p:
            raise HvkError("hvk_host_alloc", HVK_OUT_OF_MEMORY)
        self._host_bufs.append(p)
        return np.ctypeslib.as_array((C.c_int16 * (count * 2)).from_address(p)).reshape(count, 2)

    def fetch_async(self, out, first, count):
        """Queue the read-back of samples [first, first + count) into `out` (an int16 array, best from host_buffer());
        returns the ticket for fetch_wait()."""
        return self._chk("hvk_fetch_async", lib().hvk_fetch_async(self.h, out.ctypes.data, first, count))

    def fetch_wait(self, ticket):
        return self._chk("hvk_fetch_wait", lib().hvk_fetch_wait(self.h, ticket))

    FILE_TYPES = {"uint8": (0, np.uint8), "int8": (1, np.int8), "uint16": (2, np.uint16),
                  "int16": (3, np.int16), "int32": (4, np.int32), "float": (5, np.float32)}

    def fetch_as(self, first, count, type_name, complex_out=True):
        code, dt = self.FILE_TYPES[type_name]
        out = np.zeros(count * (2 if complex_out else 1), dt)
        self._chk("hvk_fetch_as", lib().hvk_fetch_as(self.h, out.ctypes.data, first, count, code, 1 if complex_out else 0))
        return out

    def fetch_raster(self, first, count):
        out = np.zeros(count, np.int16)
        self._chk("hvk_fetch_raster", lib().hvk_fetch_raster(self.h, out.ctypes.data, first, count))
        return out

    def timing_enable(self, on=True):
        return self._chk("hvk_timing_enable", lib().hvk_timing_enable(self.h, 1 if on else 0))

    def set_levels(self, mode):
        """0 auto, 1 table look-up, 2 computed per pixel (hvk_set_levels)."""
        return self._chk("hvk_set_levels", lib().hvk_set_levels(self.h, mode))

    def sound_state_size(self):
        return lib().hvk_sound_state_size(self.h)

    def sound_state_export(self):
        """The serial sound chains' state after the last frame staged (bytes; hvk_sound_state_export)."""
        n = lib().hvk_sound_state_size(self.h)
        buf = C.create_string_buffer(n)
        self._chk("hvk_sound_state_export", lib().hvk_sound_state_export(self.h, buf, n))
        return buf.raw

    def sound_state_import(self, state):
        """... into this engine; returns the position in the 32 kHz source stream the chains go on from."""
        pos = C.c_int64(0)
        self._chk("hvk_sound_state_import", lib().hvk_sound_state_import(self.h, state, len(state), C.byref(pos)))
        return pos.value

    def secam_state_export(self):
        """The colour chain's state behind the last frame staged (bytes; hvk_secam_state_export)."""
        n = lib().hvk_secam_state_size(self.h)
        buf = C.create_string_buffer(n)
        self._chk("hvk_secam_state_export", lib().hvk_secam_state_export(self.h, buf, n))
        return buf.raw

    def secam_state_import(self, state):
        return self._chk("hvk_secam_state_import", lib().hvk_secam_state_import(self.h, state, len(state)))

    def sound_samples_generated(self):
        return lib().hvk_sound_samples_generated(self.h)

    def planes_refresh(self, slots):
        """Make the picture planes of these slots now (hvk_planes_refresh)."""
        arr = (C.c_int32 * len(slots))(*slots)
        return self._chk("hvk_planes_refresh", lib().hvk_planes_refresh(self.h, arr, len(slots)))

    def kernel_names(self):
        """The kernels a launch enqueues for this configuration (one where it renders from picture planes)."""
        buf = C.create_string_buffer(512)
        self._chk("hvk_kernel_names", lib().hvk_kernel_names(self.h, buf, 512))
        return buf.value.decode().split(";")

    def timing_read(self, which):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._chk("hvk_timing_read", lib().hvk_timing_read(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def fused_launches(self):
        return int(lib().hvk_fused_launches(self.h))

    def block_sums(self, first, count):
        """(sum w[i], sum (i + 1) w[i]) modulo 2^64 over the I/Q pairs [first, first + count) of the last render, on the device."""
        out = (C.c_uint64 * 2)()
        self._chk("hvk_block_sums", lib().hvk_block_sums(self.h, first, count, out))
        return int(out[0]), int(out[1])


class _Member(Engine):
    """An engine that belongs to a group: the group opens and closes it."""

    def __init__(self, handle, device):
        self.h = C.c_void_p(handle)
        self._host_bufs = []
        info = HvkInfo()
        lib().hvk_get_info(self.h, C.byref(info))
        self.info = info.as_dict()
        self.device = device

    def close(self):
        for p in self._host_bufs:
            lib().hvk_host_free(self.h, p)
        self._host_bufs = []
        self.h = None


class Group:
    """hvk_group_*: one stream rendered by several engines, block b of `block_frames` frames on engine b mod N
    (include/hacktv_amd.h, hvk_group.cpp). devices: one HIP device ordinal per engine, repeats allowed."""

    def __init__(self, conf, sample_rate, devices, block_frames, pixel_rate=0):
        self.h = C.c_void_p()
        self.conf = conf
        devs = (C.c_int * len(devices))(*devices)
        r = lib().hvk_group_open(C.byref(self.h), C.byref(conf), sample_rate, pixel_rate, devs, len(devices), block_frames)
        if r != 0:
            self.h = None
            raise HvkError("hvk_group_open", r)
        self.n = len(devices)
        self.block = block_frames
        self.engines = [_Member(lib().hvk_group_engine(self.h, i), devices[i]) for i in range(self.n)]
        self.info = self.engines[0].info

    def _chk(self, what, r):
        if r < 0:
            raise HvkError(what, r)
        return r

    def close(self):
        if self.h:
            for e in self.engines:
                e.close()
            lib().hvk_group_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def block_engine(self):
        return self.engines[lib().hvk_group_block_index(self.h)]

    def next_frame(self):
        return int(lib().hvk_group_next_frame(self.h))

    def frame_upload(self, frame_in_block, fb, interlaced=0):
        if fb is None:
            return self._chk("hvk_group_frame_upload", lib().hvk_group_frame_upload(self.h, frame_in_block, None, 0, 0, 1, 0, interlaced))
        fb = np.ascontiguousarray(fb, dtype=np.uint32)
        return self._chk("hvk_group_frame_upload", lib().hvk_group_frame_upload(self.h, frame_in_block, fb.ctypes.data, fb.shape[1], fb.shape[0], 1, fb.shape[1], interlaced))

    def audio_write(self, stereo):
        a = np.ascontiguousarray(stereo, dtype=np.int16)
        return self._chk("hvk_group_audio_write", lib().hvk_group_audio_write(self.h, a.ctypes.data, a.size // 2))

    def audio_needed(self, nframes):
        return int(lib().hvk_group_audio_needed(self.h, nframes))

    def stage(self, nframes, slots=None):
        s = None if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        return self._chk("hvk_group_stage", lib().hvk_group_stage(self.h, nframes, None if s is None else s.ctypes.data))

    def launch(self, d_iq=None):
        return self._chk("hvk_group_launch", lib().hvk_group_launch(self.h, d_iq))

    def gather(self, root, d_root, samples):
        return self._chk("hvk_group_gather", lib().hvk_group_gather(self.h, root, d_root, samples))

    def gather_backend(self):
        return lib().hvk_group_gather_backend(self.h).decode()
