"""Loader and ctypes bindings of libhermespy_rt_amd.so (include/hermespy_rt.h, hrt_device.h).

There is no fallback: if the library has not been built (make -C hermespy-rt_amd, or
__graft_entry__.build()) loading raises, and every compute entry point needs a HIP device.
"""
import ctypes as C
import os

from . import LIB_DIR, abi

LIB_PATH = os.path.join(LIB_DIR, "libhermespy_rt_amd.so")

#: every symbol include/hermespy_rt.h and include/hrt_device.h declare
EXPORTED = (
    "compute_paths", "scene_load", "scene_save", "hrt_compute_paths_ex", "hrt_compute_paths_interleaved",
    "hrt_compute_paths_list", "hrt_path_list_free", "hrt_last_error",
    "hrt_version", "hrt_cache_clear", "hrt_problem_create", "hrt_problem_destroy", "hrt_fallback_state", "hrt_problem_num_triangles",
    "hrt_problem_num_rx", "hrt_problem_num_tx", "hrt_problem_device", "hrt_problem_eta_table",
    "hrt_problem_normals", "hrt_problem_tri_ids", "hrt_problem_tri_order", "hrt_shard_num_local",
    "hrt_shard_global_path", "hrt_launch_dirs_host", "hrt_launch_order_host", "hrt_launch_dirs_device", "hrt_launch_order_device", "hrt_layout_query", "hrt_trace",
    "hrt_work_from_counts", "hrt_timer_create", "hrt_timer_destroy", "hrt_trace_timed",
    "hrt_trace_flags",
    "hrt_timer_read", "hrt_device_count", "hrt_device_malloc", "hrt_device_free",
    "hrt_device_upload", "hrt_device_download", "hrt_device_sync", "hrt_device_mem_info",
    "hrt_selftest_math", "hrt_debug_kernel_stats", "hrt_scene_import_sionna",
    "hrt_export_meta_words", "hrt_export_words", "hrt_export_locate", "hrt_gather_create", "hrt_gather_destroy",
    "hrt_gather_meta_words", "hrt_gather_meta_device", "hrt_gather_prepare", "hrt_gather_set_meta", "hrt_gather_meta",
    "hrt_gather_pack", "hrt_gather_recv_buffer", "hrt_gather_export", "hrt_gather_received", "hrt_rccl_unique_id",
    "hrt_rccl_comm_create", "hrt_rccl_comm_destroy", "hrt_gather_rccl", "hrt_stats_size", "hrt_layout_size",
)

HIT_FIELDS = ("ray", "tri", "theta", "fs0", "ox", "oy", "oz", "dx", "dy", "dz",
              "a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau")
REC_FIELDS = ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau", "dirx", "diry", "dirz", "dfs")
LOS_FLOATS = 8
DIRS_IN_LAUNCH_ORDER = 1   # hrt_trace_flags: dirs[i] belongs to launch position i


class Shard(C.Structure):
    _fields_ = [("num_paths", C.c_uint64), ("rank", C.c_uint32), ("count", C.c_uint32),
                ("chunk", C.c_uint32), ("num_bounces", C.c_uint32)]


class Layout(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "total_bytes", "cap", "off_counts", "off_los", "off_hits", "hit_block_bytes",
        "off_recs", "rec_block_bytes", "off_masks", "off_chunk_cnt", "off_super_cnt", "off_res",
        "num_super", "off_sort_scratch", "off_sort_keys", "off_sort_tmp", "sort_tmp_bytes",
        "off_lb", "lb_stride", "off_wide_q", "off_wide_key", "wide_cap")]


class KernelTimes(C.Structure):
    _fields_ = [("los_ms", C.c_float), ("trace_ms", C.c_float * 33), ("shade_ms", C.c_float * 33),
                ("records_ms", C.c_float * 33), ("num_bounce_launches", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("live", C.c_uint64 * 34), ("records", C.c_uint64),
                ("records_unblocked", C.c_uint64), ("tests", C.c_uint64),
                ("t_setup_s", C.c_double), ("t_launch_dirs_s", C.c_double),
                ("t_device_s", C.c_double), ("t_readback_s", C.c_double),
                ("t_total_s", C.c_double), ("device", C.c_int), ("num_devices", C.c_int),
                ("num_batches", C.c_uint32), ("dev_id", C.c_int * 16), ("dev_batches", C.c_uint32 * 16),
                ("dev_t_device_s", C.c_double * 16), ("dev_t_readback_s", C.c_double * 16)]


class ExportPart(C.Structure):   # hrt_export_part
    _fields_ = [(k, C.c_uint64) for k in ("hits", "unblocked", "records", "off_hit", "off_index", "off_rec", "off_mask")]


class RcclId(C.Structure):       # ncclUniqueId
    _fields_ = [("internal", C.c_char * 128)]


EXPORT_FULL, EXPORT_UNBLOCKED = 0, 1


class HrtError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the bound library.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same soname, libamdhip64.so.7).  Two runtimes in one
    # process do not work (the second sees no device), so when torch is installed it is
    # imported FIRST: the library's DT_NEEDED then binds to the runtime torch already loaded
    # and both share streams and allocations.  C callers without torch get /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise HrtError("%s not found: build it first (make -C hermespy-rt_amd). There is no "
                       "CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    abi.bind_c_abi(L)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    V3 = C.POINTER(abi.Vec3)
    f32p = C.POINTER(C.c_float)
    L.hrt_last_error.restype = C.c_char_p
    L.hrt_version.restype = C.c_char_p
    L.hrt_cache_clear.restype = None
    L.hrt_compute_paths_ex.restype = C.c_int
    L.hrt_compute_paths_ex.argtypes = [
        C.POINTER(abi.Scene), V3, V3, V3, V3, C.c_float, C.c_size_t, C.c_size_t, C.c_size_t,
        C.c_size_t, C.POINTER(abi.ChannelInfo), C.POINTER(abi.RaysInfo),
        C.POINTER(abi.ChannelInfo), C.POINTER(abi.RaysInfo), C.POINTER(Stats)]
    L.hrt_compute_paths_interleaved.restype = C.c_int
    L.hrt_compute_paths_interleaved.argtypes = L.hrt_compute_paths_ex.argtypes
    L.hrt_problem_create.restype = C.c_int
    L.hrt_problem_create.argtypes = [C.POINTER(abi.Scene), V3, V3, V3, V3, C.c_float, C.c_size_t,
                                     C.c_size_t, C.c_int, C.POINTER(vp)]
    L.hrt_problem_destroy.argtypes = [vp]
    L.hrt_problem_destroy.restype = None
    L.hrt_fallback_state.argtypes = []
    L.hrt_fallback_state.restype = C.c_int
    for n in ("hrt_problem_num_triangles", "hrt_problem_num_rx", "hrt_problem_num_tx"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = u32
    L.hrt_problem_device.argtypes = [vp]
    L.hrt_problem_device.restype = C.c_int
    L.hrt_problem_eta_table.argtypes = [vp, f32p]
    L.hrt_problem_normals.argtypes = [vp, f32p]
    L.hrt_problem_tri_ids.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.hrt_problem_tri_order.argtypes = [vp, C.POINTER(u32)]
    L.hrt_shard_num_local.argtypes = [C.POINTER(Shard)]
    L.hrt_shard_num_local.restype = u64
    L.hrt_shard_global_path.argtypes = [C.POINTER(Shard), u64]
    L.hrt_shard_global_path.restype = u64
    L.hrt_launch_dirs_host.argtypes = [C.POINTER(Shard), f32p, C.c_int]
    L.hrt_launch_dirs_host.restype = C.c_int
    L.hrt_layout_query.argtypes = [vp, C.POINTER(Shard), C.POINTER(Layout)]
    L.hrt_layout_query.restype = C.c_int
    L.hrt_launch_order_host.argtypes = [C.POINTER(Shard), f32p, C.POINTER(u32)]
    L.hrt_launch_order_host.restype = C.c_int
    L.hrt_launch_dirs_device.argtypes = [C.POINTER(Shard), vp, C.c_int, vp, C.POINTER(u64)]
    L.hrt_launch_dirs_device.restype = C.c_int
    L.hrt_launch_order_device.argtypes = [C.POINTER(Shard), vp, C.c_int, vp]
    L.hrt_launch_order_device.restype = C.c_int
    L.hrt_trace.argtypes = [vp, C.POINTER(Shard), vp, vp, vp, u64, vp, C.POINTER(KernelTimes)]
    L.hrt_trace.restype = C.c_int
    L.hrt_timer_create.argtypes = [u32, C.POINTER(vp)]
    L.hrt_timer_create.restype = C.c_int
    L.hrt_timer_destroy.argtypes = [vp]
    L.hrt_timer_destroy.restype = None
    L.hrt_trace_timed.argtypes = [vp, C.POINTER(Shard), vp, vp, vp, u64, vp, vp]
    L.hrt_trace_timed.restype = C.c_int
    L.hrt_trace_flags.argtypes = [vp, C.POINTER(Shard), vp, vp, vp, u64, vp, vp, C.c_uint32]
    L.hrt_trace_flags.restype = C.c_int
    L.hrt_timer_read.argtypes = [vp, C.POINTER(KernelTimes)]
    L.hrt_timer_read.restype = C.c_int
    L.hrt_work_from_counts.argtypes = [vp, C.POINTER(Shard), C.POINTER(u32), C.POINTER(Stats)]
    L.hrt_work_from_counts.restype = None
    L.hrt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.hrt_device_malloc.argtypes = [C.c_int, C.POINTER(vp), u64]
    L.hrt_device_free.argtypes = [C.c_int, vp]
    L.hrt_device_upload.argtypes = [C.c_int, vp, vp, u64]
    L.hrt_device_download.argtypes = [C.c_int, vp, vp, u64]
    L.hrt_device_sync.argtypes = [C.c_int, vp]
    L.hrt_device_mem_info.argtypes = [C.c_int, C.POINTER(u64), C.POINTER(u64)]
    L.hrt_selftest_math.argtypes = [C.c_int, C.c_int, f32p, f32p, u64]
    L.hrt_selftest_math.restype = C.c_int
    L.hrt_debug_kernel_stats.argtypes = [C.c_int, C.POINTER(u64), C.c_int]
    L.hrt_debug_kernel_stats.restype = C.c_int
    L.hrt_scene_import_sionna.argtypes = [C.c_char_p, C.POINTER(abi.Scene)]
    L.hrt_scene_import_sionna.restype = C.c_int
    # packed export and gather (include/hrt_device.h)
    u32p = C.POINTER(u32)
    L.hrt_export_meta_words.argtypes = [u32, u32]
    L.hrt_export_meta_words.restype = u32
    L.hrt_export_words.argtypes = [u32p, u32, u32, u32]
    L.hrt_export_words.restype = u64
    L.hrt_export_locate.argtypes = [u32p, u32, u32, u32, u32, u32, C.POINTER(ExportPart)]
    L.hrt_gather_create.argtypes = [vp, C.POINTER(Shard), C.c_int, u32, C.POINTER(vp)]
    L.hrt_gather_destroy.argtypes = [vp]
    L.hrt_gather_destroy.restype = None
    L.hrt_gather_meta_words.argtypes = [vp]
    L.hrt_gather_meta_words.restype = u32
    L.hrt_gather_meta_device.argtypes = [vp]
    L.hrt_gather_meta_device.restype = vp
    L.hrt_gather_prepare.argtypes = [vp, vp, vp]
    L.hrt_gather_set_meta.argtypes = [vp, u32, u32p]
    L.hrt_gather_meta.argtypes = [vp, u32]
    L.hrt_gather_meta.restype = u32p
    L.hrt_gather_pack.argtypes = [vp, vp, vp, C.POINTER(vp), C.POINTER(u64)]
    L.hrt_gather_recv_buffer.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
    L.hrt_gather_export.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
    L.hrt_gather_received.argtypes = [vp, u32]
    L.hrt_gather_received.restype = vp
    L.hrt_rccl_unique_id.argtypes = [C.POINTER(RcclId)]
    L.hrt_rccl_comm_create.argtypes = [C.POINTER(RcclId), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.hrt_rccl_comm_destroy.argtypes = [vp]
    L.hrt_gather_rccl.argtypes = [vp, vp, vp, vp, C.c_int]
    L.hrt_stats_size.restype = u64
    L.hrt_layout_size.restype = u64
    # the library writes hrt_stats / hrt_layout in full: a mirror of another size would be overrun
    if int(L.hrt_stats_size()) != C.sizeof(Stats) or int(L.hrt_layout_size()) != C.sizeof(Layout):
        raise HrtError("lib.py's mirrors of hrt_stats / hrt_layout (%d / %d bytes) do not match the library's (%d / %d)"
                       % (C.sizeof(Stats), C.sizeof(Layout), int(L.hrt_stats_size()), int(L.hrt_layout_size())))
    _lib = L
    return L


def check(rc, what="hermespy-rt_amd"):
    if rc != 0:
        raise HrtError("%s failed (%d): %s" % (what, rc, load().hrt_last_error().decode()))
