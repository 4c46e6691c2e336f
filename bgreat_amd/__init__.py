"""bgreat_amd -- ctypes view of the C-ABI in include/bgreat_gpu.h (lib/libbgreat_gpu.so).

The product is the shared library + the `bgreat` CLI (C++/HIP).  This module only lets Python (tests,
bench.py, __graft_entry__) call the same entry points; it contains no algorithm and no CPU fallback: if
the library is missing it raises, and mapping calls fail when there is no HIP device.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGR_LIB_PATH") or os.path.join(_HERE, "lib", "libbgreat_gpu.so")  # (the override: A/B builds of the kernels, tools only)
CLI_PATH = os.path.join(_HERE, "bin", "bgreat")

MODE_GREEDY, MODE_EXHAUSTIVE, MODE_ANCHORS = 0, 1, 2
ST_NOANCHOR, ST_FAILED, ST_ALIGNED, ST_MASK, ST_RC = 0, 1, 2, 3, 4
BUILD_ANCHORS = 1
BUILD_NO_EVICTIONS = 2  # test hook: keys whose two buckets are full go to the fallback list

# every symbol include/bgreat_gpu.h declares (checked by tests/test_cabi.py)
SYMBOLS = [
    "bgr_last_error", "bgr_device_count", "bgr_graph_build", "bgr_graph_build_from_fasta", "bgr_graph_blob",
    "bgr_graph_from_blob", "bgr_graph_info", "bgr_graph_destroy", "bgr_graph_upload", "bgr_graph_device_blob",
    "bgr_graph_adopt_device_blob", "bgr_aligner_create", "bgr_aligner_destroy", "bgr_align_batch", "bgr_align_device",
    "bgr_aligner_sync", "bgr_aligner_device_results", "bgr_aligner_fetch", "bgr_aligner_counters",
    "bgr_aligner_reset_counters", "bgr_aligner_kernel_time", "bgr_aligner_reset_kernel_time", "bgr_aligner_launch_info",
    "bgr_aligner_configure", "bgr_readset_load", "bgr_readset_count", "bgr_readset_view", "bgr_readset_destroy",
    "bgr_write_records", "bgr_graph_unitigs", "bgr_readset_load_parallel", "bgr_align_all", "bgr_host_alloc", "bgr_host_free",
    "bgr_set_build_threads", "bgr_graph_build_ex", "bgr_graph_build_from_fasta_ex", "bgr_graph_anchor_lookup", "bgr_graph_key_lookup",
    "bgr_aligner_set_knob", "bgr_aligner_pass_counts", "bgr_aligner_last_pass_runs", "bgr_set_option", "bgr_get_option", "bgr_option_name", "bgr_plan_launch", "bgr_aligner_kernel_times", "bgr_devices_init", "bgr_devices_method", "bgr_packed_plane_words", "bgr_pack_reads", "bgr_align_batch_packed",
    "bgr_align_fasta_text", "bgr_aligner_fetch_text", "bgr_host_cache_release", "bgr_device_local_cpus", "bgr_text_stage_create", "bgr_text_stage_destroy", "bgr_text_stage_upload",
    "bgr_align_batch_begin", "bgr_align_batch_test", "bgr_align_batch_wait", "bgr_text_stage_device", "bgr_text_stage_upload_parts",
    "bgr_device_alloc", "bgr_device_free", "bgr_device_upload", "bgr_device_download",
]
KNOB_EXH_FRAME_CAP, KNOB_EXH_SEARCH, KNOB_BATCH_SPLIT_LIMIT, KNOB_DEBUG_STOP, KNOB_GREEDY_FAST, KNOB_EXH_FAST, KNOB_ANCHORS_FAST, KNOB_BATCH_OVERLAP, KNOB_EXH_MEMO_CAP, KNOB_GREEDY_PREPASS, KNOB_KERNEL_EVENTS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
SEARCH_AUTO, SEARCH_DEPTH_FIRST, SEARCH_BY_LEVEL = 0, 1, 2


class _Borrowed(np.ndarray):
    """ndarray view of library-owned memory; `_owner` pins the handle that owns it."""
    _owner = None


class PlanInput(C.Structure):  # bgr_plan_input
    _fields_ = [("k", C.c_uint32), ("slot_fill_x100", C.c_uint32), ("table_bytes", C.c_uint32), ("has_exceptions", C.c_uint32), ("anchors", C.c_uint32), ("anchor_levels", C.c_uint32),
                ("graph_bases", C.c_uint64), ("n_unitigs", C.c_uint64), ("max_unitig_len", C.c_uint64),
                ("num_cus", C.c_uint32), ("resident_waves", C.c_uint32 * 7), ("lds_per_cu", C.c_uint64),
                ("cfg_waves", C.c_uint32), ("cfg_blocks_per_cu", C.c_uint32), ("cfg_lds_mphf", C.c_uint32),
                ("mode", C.c_uint32), ("max_mismatch", C.c_uint32), ("partial", C.c_uint32), ("max_read_len", C.c_uint32),
                ("n_reads", C.c_uint64), ("total_bases", C.c_uint64)]


class PlanPass(C.Structure):
    _fields_ = [("used", C.c_uint32), ("blocks", C.c_uint32), ("waves_per_block", C.c_uint32), ("lds_bytes", C.c_uint32), ("table_staged", C.c_uint32)]


class PlanOutput(C.Structure):  # bgr_plan_output
    _fields_ = [("pass_", PlanPass * 6), ("level_search", C.c_uint32), ("deep_only", C.c_uint32), ("x4_levels", C.c_uint32), ("memo_cap", C.c_uint32),
                ("deep_scratch_bytes", C.c_uint64), ("arena_ints", C.c_uint64)]


class BgrError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("max_mismatch", C.c_uint32), ("effort", C.c_uint32), ("partial", C.c_uint32)]


class RunOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("n_gpus", C.c_uint32), ("threads", C.c_uint32), ("batch_reads", C.c_uint64), ("chunk_bytes", C.c_uint64),
                ("fastq", C.c_uint32), ("write_exhaustive", C.c_uint32), ("echo_files", C.c_uint32), ("correction", C.c_uint32),
                ("no_overlap_file", C.c_char_p), ("first_device", C.c_uint32), ("route", C.c_uint32), ("numa", C.c_uint32), ("split_output", C.c_uint32)]


class Ticket(C.Structure):
    _fields_ = [("aligner", C.c_void_p), ("n_reads", C.c_uint64), ("serial", C.c_uint64)]


class TextBatch(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("text", C.c_void_p), ("text_bytes", C.c_uint64), ("want_output", C.c_uint32), ("irregular", C.c_uint32), ("paths_out", C.c_void_p),
                ("paths_cap", C.c_uint64), ("notaligned_out", C.c_void_p), ("notaligned_cap", C.c_uint64), ("n_records", C.c_uint64),
                ("n_accepted", C.c_uint64), ("paths_bytes", C.c_uint64), ("notaligned_bytes", C.c_uint64), ("stage", C.c_void_p),
                ("fastq", C.c_uint32), ("reserved", C.c_uint32), ("record_info_out", C.c_void_p), ("record_info_cap", C.c_uint64)]


class PackedReads(C.Structure):
    _fields_ = [("read_offsets", C.c_void_p), ("fw3", C.c_void_p), ("hasn", C.c_void_p), ("nm_index", C.c_void_p), ("nm_value", C.c_void_p),
                ("nm_count", C.c_uint64), ("max_read_len", C.c_uint32)]


class GraphInfo(C.Structure):
    _fields_ = [("k", C.c_uint32), ("n_levels", C.c_uint32), ("n_unitigs", C.c_uint64), ("n_keys", C.c_uint64),
                ("n_left_keys", C.c_uint64), ("n_right_keys", C.c_uint64), ("n_fallback", C.c_uint64),
                ("total_bases", C.c_uint64), ("blob_bytes", C.c_uint64), ("mphf_bytes", C.c_uint64),
                ("max_unitig_len", C.c_uint64), ("has_exceptions", C.c_uint32), ("has_anchors", C.c_uint32),
                ("gamma", C.c_double)]


_lib = None


def build(force=False):
    """Compile lib/libbgreat_gpu.so and bin/bgreat for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def _pin_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's): whichever is loaded
    first serves the whole process.  Device pointers only make sense inside ONE runtime, so when torch is
    installed its copy is loaded first (without importing torch); a later `import torch` then shares it."""
    if "torch" in sys.modules or os.environ.get("BGREAT_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _pin_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise BgrError("libbgreat_gpu.so is not built (%s); run `make -C bgreat_amd` -- there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.bgr_last_error.restype = C.c_char_p
    L.bgr_set_build_threads.restype = None
    L.bgr_set_build_threads.argtypes = [C.c_uint32]
    L.bgr_device_count.restype = i32
    L.bgr_graph_build.argtypes = [u32, u64, vp, vp, C.c_double, C.POINTER(vp)]
    L.bgr_graph_build_from_fasta.argtypes = [C.c_char_p, u32, C.c_double, C.POINTER(vp)]
    L.bgr_graph_build_ex.argtypes = [u32, u64, vp, vp, C.c_double, u32, C.POINTER(vp)]
    L.bgr_graph_build_from_fasta_ex.argtypes = [C.c_char_p, u32, C.c_double, u32, C.POINTER(vp)]
    L.bgr_graph_anchor_lookup.argtypes = [vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.bgr_graph_key_lookup.argtypes = [vp, u64, C.POINTER(C.c_uint32)]
    L.bgr_graph_blob.restype = vp
    L.bgr_graph_blob.argtypes = [vp, C.POINTER(u64)]
    L.bgr_graph_from_blob.argtypes = [vp, u64, C.POINTER(vp)]
    L.bgr_graph_info.argtypes = [vp, C.POINTER(GraphInfo)]
    L.bgr_graph_unitigs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.bgr_graph_destroy.argtypes = [vp]
    L.bgr_graph_destroy.restype = None
    L.bgr_graph_upload.argtypes = [vp, i32]
    L.bgr_graph_device_blob.restype = vp
    L.bgr_graph_device_blob.argtypes = [vp, i32]
    L.bgr_graph_adopt_device_blob.argtypes = [i32, vp, u64, C.POINTER(vp)]
    L.bgr_devices_init.argtypes = [vp, i32, u32, u32]
    L.bgr_devices_method.argtypes = [vp]
    L.bgr_devices_method.restype = u32
    L.bgr_aligner_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.bgr_aligner_destroy.argtypes = [vp]
    L.bgr_aligner_destroy.restype = None
    L.bgr_align_batch.argtypes = [vp, C.POINTER(Params), vp, vp, u64, vp, u64, vp, vp]
    L.bgr_align_device.argtypes = [vp, C.POINTER(Params), vp, vp, u64, u64, u32]
    L.bgr_packed_plane_words.argtypes = [u64, u64]
    L.bgr_packed_plane_words.restype = u64
    L.bgr_pack_reads.argtypes = [vp, vp, u64, vp, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u32)]
    L.bgr_align_batch_packed.argtypes = [vp, C.POINTER(Params), C.POINTER(PackedReads), u64, vp, u64, vp, vp]
    L.bgr_align_fasta_text.argtypes = [vp, C.POINTER(Params), C.POINTER(TextBatch)]
    L.bgr_align_batch_begin.argtypes = [vp, C.POINTER(Params), vp, vp, u64, C.POINTER(Ticket)]
    L.bgr_align_batch_test.argtypes = [C.POINTER(Ticket)]
    L.bgr_align_batch_wait.argtypes = [C.POINTER(Ticket), vp, u64, vp, vp]
    L.bgr_aligner_fetch_text.argtypes = [vp, C.POINTER(TextBatch)]
    L.bgr_text_stage_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.bgr_text_stage_destroy.argtypes = [vp]
    L.bgr_text_stage_destroy.restype = None
    L.bgr_text_stage_upload.argtypes = [vp, vp, u64]
    L.bgr_aligner_sync.argtypes = [vp]
    L.bgr_aligner_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.bgr_aligner_fetch.argtypes = [vp, u64, vp, u64, vp, vp]
    L.bgr_aligner_counters.argtypes = [vp, vp]
    L.bgr_aligner_reset_counters.argtypes = [vp]
    L.bgr_aligner_kernel_time.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double)]
    L.bgr_aligner_reset_kernel_time.argtypes = [vp]
    L.bgr_aligner_kernel_times.argtypes = [vp, C.POINTER(u64), vp, vp]
    L.bgr_aligner_launch_info.argtypes = [vp, vp]
    L.bgr_aligner_configure.argtypes = [vp, u32, u32, u32]
    L.bgr_aligner_set_knob.argtypes = [vp, u32, u64]
    L.bgr_aligner_pass_counts.argtypes = [vp, vp]
    L.bgr_aligner_last_pass_runs.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.bgr_set_option.argtypes = [C.c_char_p, C.c_int64]
    L.bgr_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    L.bgr_option_name.restype = C.c_char_p
    L.bgr_option_name.argtypes = [u32, C.POINTER(C.c_char_p)]
    L.bgr_plan_launch.argtypes = [C.POINTER(PlanInput), C.POINTER(PlanOutput)]
    L.bgr_readset_load.argtypes = [C.c_char_p, i32, u32, C.POINTER(vp)]
    L.bgr_readset_load_parallel.argtypes = [C.c_char_p, i32, u32, u32, u64, C.POINTER(vp)]
    L.bgr_align_all.argtypes = [vp, C.POINTER(Params), C.POINTER(RunOptions), C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(C.c_double)]
    L.bgr_text_stage_device.argtypes = [vp]
    L.bgr_text_stage_upload_parts.argtypes = [vp, u32, vp, vp]
    L.bgr_device_alloc.argtypes = [i32, u64, C.POINTER(vp)]
    L.bgr_device_free.argtypes = [i32, vp]
    L.bgr_device_upload.argtypes = [i32, vp, vp, u64]
    L.bgr_device_download.argtypes = [i32, vp, vp, u64]
    L.bgr_host_alloc.argtypes = [u64, C.POINTER(vp)]
    L.bgr_host_free.argtypes = [vp]
    L.bgr_readset_count.restype = u64
    L.bgr_readset_count.argtypes = [vp]
    L.bgr_readset_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.bgr_readset_destroy.argtypes = [vp]
    L.bgr_readset_destroy.restype = None
    L.bgr_write_records.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise BgrError("bgreat_gpu error %d: %s" % (rc, lib().bgr_last_error().decode("utf-8", "replace")))


def device_count():
    return lib().bgr_device_count()


def set_option(name, value):
    """Process-wide library option (bgr_set_option): the library reads no environment variables."""
    _check(lib().bgr_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int64(0)
    _check(lib().bgr_get_option(name.encode(), C.byref(v)))
    return int(v.value)


def option_names():
    out, i = [], 0
    while True:
        what = C.c_char_p()
        n = lib().bgr_option_name(i, C.byref(what))
        if n is None:
            return out
        out.append((n.decode(), what.value.decode()))
        i += 1


def set_options_from_string(spec):
    """"name=value,name=value" -> bgr_set_option (tools: their BGR_FUZZ_OPTIONS / --options; the LIBRARY reads no environment)."""
    for kv in filter(None, (spec or "").split(",")):
        k, v = kv.split("=")
        set_option(k.strip(), int(v))


class options:
    """with B.options(build_filter=2, **{"test.bases_cap": 5000}): ...  -- set, and put back on exit (tests)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def plan_launch(**kw):
    """bgr_plan_launch: the launch geometry for a graph header / device / batch given as numbers (no device needed).  -> dict, or raises BgrError."""
    i, o = PlanInput(), PlanOutput()
    for k, v in kw.items():
        if k == "resident_waves":
            for j, w in enumerate(v):
                i.resident_waves[j] = int(w)
        else:
            setattr(i, k, int(v))
    _check(lib().bgr_plan_launch(C.byref(i), C.byref(o)))
    names = ["general", "greedy16", "exhaustive8", "anchors4", "depth_first_mid", "last"]
    d = {n: dict(used=bool(p.used), blocks=p.blocks, waves_per_block=p.waves_per_block, lds_bytes=p.lds_bytes, table_staged=bool(p.table_staged)) for n, p in zip(names, o.pass_)}
    d.update(level_search=bool(o.level_search), deep_only=bool(o.deep_only), x4_levels=o.x4_levels, memo_cap=o.memo_cap, deep_scratch_bytes=o.deep_scratch_bytes, arena_ints=o.arena_ints)
    return d


class DeviceBuffer:
    """A numpy array parked in a device's HBM through the C-ABI (bgr_device_alloc / upload): input of bgr_align_device."""

    def __init__(self, device, array):
        a = np.ascontiguousarray(array)
        self.device, self.nbytes = device, a.nbytes
        p = C.c_void_p()
        _check(lib().bgr_device_alloc(device, a.nbytes, C.byref(p)))
        self.ptr = p.value
        _check(lib().bgr_device_upload(device, self.ptr, a.ctypes.data, a.nbytes))

    def data_ptr(self):
        return self.ptr

    def free(self):
        if getattr(self, "ptr", None):
            lib().bgr_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:   # (interpreter shutdown: the library may be gone already)
            pass


def _as_u8(a):
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(a, dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def pack_reads(reads, offsets):
    """ASCII reads -> the 2-bit planes of bgr_align_batch_packed (host side; bgr_pack_reads).  -> dict of arrays."""
    reads = _as_u8(reads)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    rel = offsets - offsets[0]
    total = int(rel[n])
    words = lib().bgr_packed_plane_words(n, total)
    fw3 = np.zeros(words, dtype=np.uint64)
    hasn = np.zeros((n + 31) // 32 + 1, dtype=np.uint32)
    cap = max(64, total // 16 + 64)
    nm_index = np.zeros(cap, dtype=np.uint32)
    nm_value = np.zeros(cap, dtype=np.uint64)
    cnt, mx = C.c_uint64(), C.c_uint32()
    _check(lib().bgr_pack_reads(reads.ctypes.data, offsets.ctypes.data, n, fw3.ctypes.data, hasn.ctypes.data, nm_index.ctypes.data,
                                nm_value.ctypes.data, cap, C.byref(cnt), C.byref(mx)))
    return {"read_offsets": rel, "fw3": fw3, "hasn": hasn, "nm_index": nm_index[: cnt.value].copy(), "nm_value": nm_value[: cnt.value].copy(),
            "max_read_len": mx.value, "n": n}


class Graph:
    """The immutable index (Aligner::indexUnitigs, aligner.cpp:407-547)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def build(cls, k, seqs, offsets, gamma=0.0, anchors=False, no_evictions=False):
        """anchors=True also builds the k-mer anchors index of -G mode (MODE_ANCHORS); gamma = key table slots per key (0 = default)."""
        seqs = _as_u8(seqs)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().bgr_graph_build_ex(k, len(offsets) - 1, seqs.ctypes.data, offsets.ctypes.data, gamma,
                                        (BUILD_ANCHORS if anchors else 0) | (BUILD_NO_EVICTIONS if no_evictions else 0), C.byref(h)))
        return cls(h)

    @classmethod
    def from_fasta(cls, path, k, gamma=0.0, anchors=False):
        h = C.c_void_p()
        _check(lib().bgr_graph_build_from_fasta_ex(path.encode(), k, gamma, BUILD_ANCHORS if anchors else 0, C.byref(h)))
        return cls(h)

    def key_lookup(self, key):
        """slot of a canonical (k-1)-mer in the overlap key table, None for a non-member"""
        slot = C.c_uint32()
        _check(lib().bgr_graph_key_lookup(self.h, int(key), C.byref(slot)))
        return None if slot.value == 0xFFFFFFFF else int(slot.value)

    def anchor_lookup(self, kmer):
        """(index, unitig, offset) of boomphf::mphf::lookup(kmer) on the anchors index; index None for ULLONG_MAX."""
        idx, pos = C.c_uint64(), C.c_uint64()
        _check(lib().bgr_graph_anchor_lookup(self.h, int(kmer), C.byref(idx), C.byref(pos)))
        if idx.value == 0xFFFFFFFFFFFFFFFF:
            return None, 0, 0
        return idx.value, pos.value >> 32, pos.value & 0xFFFFFFFF

    @classmethod
    def from_blob(cls, blob):
        blob = _as_u8(blob)
        h = C.c_void_p()
        _check(lib().bgr_graph_from_blob(blob.ctypes.data, blob.size, C.byref(h)))
        return cls(h)

    @classmethod
    def adopt_device_blob(cls, device, dev_ptr, nbytes):
        h = C.c_void_p()
        _check(lib().bgr_graph_adopt_device_blob(device, dev_ptr, nbytes, C.byref(h)))
        return cls(h)

    def blob(self):
        n = C.c_uint64()
        p = lib().bgr_graph_blob(self.h, C.byref(n))
        if not p:
            return np.zeros(0, dtype=np.uint8)
        v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).view(_Borrowed)
        v._owner = self   # the bytes belong to the graph: keep it alive as long as the view (or a view of it) is
        return v

    def info(self):
        gi = GraphInfo()
        _check(lib().bgr_graph_info(self.h, C.byref(gi)))
        return {f[0]: getattr(gi, f[0]) for f in GraphInfo._fields_ if f[0] != "reserved"}

    def upload(self, device=0):
        _check(lib().bgr_graph_upload(self.h, device))

    def devices_init(self, first_device=0, n_devices=1, how=0):
        """Graph resident on n devices: one upload, then RCCL broadcast / xGMI peer copies (bgr_devices_init) -> method used."""
        _check(lib().bgr_devices_init(self.h, first_device, n_devices, how))
        return lib().bgr_devices_method(self.h)

    def device_blob(self, device=0):
        return lib().bgr_graph_device_blob(self.h, device)

    def close(self):
        if self.h:
            lib().bgr_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Aligner:
    """Batch form of alignReadGreedy / alignReadExhaustive on one GPU."""

    def __init__(self, graph, device=0):
        self.graph = graph
        self.h = C.c_void_p()
        _check(lib().bgr_aligner_create(graph.h, device, C.byref(self.h)))

    def configure(self, waves_per_block=0, blocks_per_cu=0, lds_mphf=0):
        _check(lib().bgr_aligner_configure(self.h, waves_per_block, blocks_per_cu, lds_mphf))

    def set_knob(self, knob, value):
        """Test / diagnostic hooks (KNOB_*), see include/bgreat_gpu.h."""
        _check(lib().bgr_aligner_set_knob(self.h, knob, int(value)))

    def align(self, reads, offsets, m=2, effort=2, mode=MODE_GREEDY, partial=False):
        """-> (paths int32[], path_offsets uint64[n+1], status uint8[n]) in input order."""
        reads = _as_u8(reads)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = int(offsets[-1] - offsets[0]) + 8 * n + 8
        paths = np.empty(cap, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(max(n, 1), dtype=np.uint8)
        p = Params(mode, m, effort, int(partial))
        _check(lib().bgr_align_batch(self.h, C.byref(p), reads.ctypes.data, offsets.ctypes.data, n, paths.ctypes.data, cap,
                                     poffs.ctypes.data, status.ctypes.data))
        return paths[: int(poffs[n])].copy(), poffs, status[:n]

    def align_packed(self, pk, m=2, effort=2, mode=MODE_GREEDY, partial=False, out=None):
        """bgr_align_batch_packed on the dict pack_reads() returns -> (paths, path_offsets, status)."""
        n = pk["n"]
        cap = int(pk["read_offsets"][n]) + 8 * n + 8
        paths = np.empty(cap, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(max(n, 1), dtype=np.uint8)
        s = PackedReads(pk["read_offsets"].ctypes.data, pk["fw3"].ctypes.data, pk["hasn"].ctypes.data,
                        pk["nm_index"].ctypes.data if len(pk["nm_index"]) else None, pk["nm_value"].ctypes.data if len(pk["nm_value"]) else None,
                        len(pk["nm_index"]), pk["max_read_len"])
        p = Params(mode, m, effort, int(partial))
        _check(lib().bgr_align_batch_packed(self.h, C.byref(p), C.byref(s), n, paths.ctypes.data, cap, poffs.ctypes.data, status.ctypes.data))
        return paths[: int(poffs[n])].copy(), poffs, status[:n]

    def align_begin(self, reads, offsets, m=2, effort=2, mode=MODE_GREEDY, partial=False):
        """bgr_align_batch_begin: copy + launch enqueued, returns at once -> ticket (keeps the host arrays alive until align_wait)."""
        reads = _as_u8(reads)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        t = Ticket()
        p = Params(mode, m, effort, int(partial))
        _check(lib().bgr_align_batch_begin(self.h, C.byref(p), reads.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(t)))
        return (t, reads, offsets)

    def align_test(self, ticket):
        rc = lib().bgr_align_batch_test(C.byref(ticket[0]))
        if rc < 0:
            _check(rc)
        return bool(rc)

    def align_wait(self, ticket):
        """bgr_align_batch_wait -> (paths, path_offsets, status) of the batch the ticket was given for."""
        t, reads, offsets = ticket
        n = len(offsets) - 1
        cap = int(offsets[-1] - offsets[0]) + 8 * n + 8
        paths = np.empty(cap, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(max(n, 1), dtype=np.uint8)
        _check(lib().bgr_align_batch_wait(C.byref(t), paths.ctypes.data, cap, poffs.ctypes.data, status.ctypes.data))
        return paths[: int(poffs[n])].copy(), poffs, status[:n]

    def align_fasta_text(self, text, m=2, effort=2, mode=MODE_GREEDY, partial=False, want_output=True, paths_cap=None, staged=False, fastq=False, parts=None, record_info=False):
        """bgr_align_fasta_text: a piece of a FASTA file (bytes) -> (paths bytes, notAligned bytes, info dict); info["irregular"] = the
        device left the piece to the host parser (nothing mapped).  fastq: True / 1 = the piece is whole four-line FASTQ records instead, 2 = their
        header and read lines only.  A too small `paths_cap` is grown through bgr_aligner_fetch_text.  parts (with staged): byte offsets at which
        the piece is cut into host ranges sent with bgr_text_stage_upload_parts (the call itself then gets no host pointer)."""
        text = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else _as_u8(text)
        n = len(text)
        pcap = n + 64 if paths_cap is None else paths_cap
        pout = np.empty(max(pcap, 1), dtype=np.uint8)
        nout = np.empty(n + 64, dtype=np.uint8)
        b = TextBatch(C.sizeof(TextBatch), text.ctypes.data if n else None, n, int(want_output), 0, pout.ctypes.data, pcap, nout.ctypes.data, n + 64, 0, 0, 0, 0, None)
        b.fastq = int(fastq)
        rinfo = None
        if record_info:
            rinfo = np.zeros(n // 24 + 1024, dtype=np.uint32)
            b.record_info_out, b.record_info_cap = rinfo.ctypes.data, len(rinfo)
        p = Params(mode, m, effort, int(partial))
        stage = C.c_void_p()
        if staged:  # the piece sent ahead on a copy stream of its own (bgr_text_stage_upload); the call orders itself behind it
            _check(lib().bgr_text_stage_create(0, C.byref(stage)))
            if parts is None:
                _check(lib().bgr_text_stage_upload(stage, text.ctypes.data if n else None, n))
            else:
                cuts = [0] + sorted(int(x) for x in parts) + [n]
                keep = [np.ascontiguousarray(text[cuts[i]: cuts[i + 1]]).copy() for i in range(len(cuts) - 1)]   # separate host ranges
                ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data if len(k) else None for k in keep])
                lens = (C.c_uint64 * len(keep))(*[len(k) for k in keep])
                _check(lib().bgr_text_stage_upload_parts(stage, len(keep), ptrs, lens))
                b.text = None
            b.stage = stage
        try:
            rc = lib().bgr_align_fasta_text(self.h, C.byref(p), C.byref(b))
            if rc == -4:  # BGR_E_CAPACITY: the mapping is done, the bytes did not fit (the stage still holds the text the records are cut from)
                pout = np.empty(int(b.paths_bytes) + 64, dtype=np.uint8)
                b.paths_out, b.paths_cap = pout.ctypes.data, len(pout)
                rc = lib().bgr_aligner_fetch_text(self.h, C.byref(b))
        finally:
            if staged:
                lib().bgr_text_stage_destroy(stage)
        _check(rc)
        info = {"irregular": bool(b.irregular), "n_records": int(b.n_records), "n_accepted": int(b.n_accepted)}
        if rinfo is not None:  # one word per record: kept << 31 | mapped << 30 | read length
            info["records"] = rinfo[: int(b.n_records)].copy()
        return pout[: int(b.paths_bytes)].tobytes(), nout[: int(b.notaligned_bytes)].tobytes(), info

    def align_device(self, d_reads_ptr, d_offsets_ptr, n, total_bases, max_len, m=2, effort=2, mode=MODE_GREEDY, partial=False):
        p = Params(mode, m, effort, int(partial))
        _check(lib().bgr_align_device(self.h, C.byref(p), d_reads_ptr, d_offsets_ptr, n, total_bases, max_len))

    def fetch(self, n, cap):
        paths = np.empty(cap, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(max(n, 1), dtype=np.uint8)
        _check(lib().bgr_aligner_fetch(self.h, n, paths.ctypes.data, cap, poffs.ctypes.data, status.ctypes.data))
        return paths[: int(poffs[n])].copy(), poffs, status[:n]

    def sync(self):
        _check(lib().bgr_aligner_sync(self.h))

    def counters(self):
        out = np.zeros(5, dtype=np.uint64)
        _check(lib().bgr_aligner_counters(self.h, out.ctypes.data))
        return dict(zip(["reads", "no_overlap", "aligned", "not_aligned", "overlaps"], (int(x) for x in out)))

    def reset_counters(self):
        _check(lib().bgr_aligner_reset_counters(self.h))

    def kernel_time(self):
        n, ms = C.c_uint64(), C.c_double()
        _check(lib().bgr_aligner_kernel_time(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_times(self):
        """-> (launches, [(kernel name, summed ms), ...]) per kernel of a launch, in launch order."""
        n = C.c_uint64()
        ms = (C.c_double * 8)()
        names = (C.c_char_p * 8)()
        _check(lib().bgr_aligner_kernel_times(self.h, C.byref(n), ms, names))
        return n.value, [(names[i].decode(), ms[i]) for i in range(8) if names[i]]

    def reset_kernel_time(self):
        _check(lib().bgr_aligner_reset_kernel_time(self.h))

    def launch_info(self):
        out = np.zeros(4, dtype=np.uint32)
        _check(lib().bgr_aligner_launch_info(self.h, out.ctypes.data))
        return {"blocks": int(out[0]), "threads": int(out[1]), "lds_bytes": int(out[2]), "mphf_in_lds": bool(out[3] & 1),
                "level_search": bool(out[3] & 2), "four_reads_per_wave": bool(out[3] & 4)}

    def pass_counts(self):
        """Reads each pass of the last launch handed on (see bgr_aligner_pass_counts): 4 ints."""
        out = np.zeros(4, dtype=np.uint32)
        _check(lib().bgr_aligner_pass_counts(self.h, out.ctypes.data))
        return tuple(int(x) for x in out)

    def last_pass_runs(self):
        """Exhaustive mode: (runs of the last pass for the launch last settled, entries per wave of its table of remembered calls in the final run)."""
        r, c = C.c_uint32(0), C.c_uint32(0)
        _check(lib().bgr_aligner_last_pass_runs(self.h, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    def close(self):
        if self.h:
            lib().bgr_aligner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_all(graph, reads_csv, paths_file, notaligned_file, m=2, effort=2, mode=MODE_GREEDY, partial=False, n_gpus=1, threads=1,
              batch_reads=0, chunk_bytes=0, fastq=False, write_exhaustive=False, correction=False, no_overlap_file=None, first_device=0, route=0, numa=0, split_output=False):
    """Aligner::alignAll (aligner.cpp:550-597) as one call -> (counters dict, mapping seconds).  route: 0 = FASTA goes through the device as
    text when it can (bgr_align_fasta_text), 1 = host parser + host formatter always.  split_output: one pipeline per device, device d
    writing `<paths_file>.<d>` / `<notaligned_file>.<d>` (their concatenation = the single-file bytes)."""
    p = Params(mode, m, effort, int(partial))
    o = RunOptions(C.sizeof(RunOptions), n_gpus, threads, batch_reads, chunk_bytes, int(fastq), int(write_exhaustive), 0, int(correction),
                   no_overlap_file.encode() if no_overlap_file else None, first_device, route, numa, int(split_output))
    out = np.zeros(5, dtype=np.uint64)
    secs = C.c_double()
    _check(lib().bgr_align_all(graph.h, C.byref(p), C.byref(o), reads_csv.encode(), paths_file.encode(), notaligned_file.encode(),
                               out.ctypes.data, C.byref(secs)))
    return dict(zip(["reads", "no_overlap", "aligned", "not_aligned", "overlaps"], (int(x) for x in out))), secs.value


def load_reads(path, k, fastq=False, threads=1, chunk_bytes=0):
    """getReads (aligner.cpp:46-117) over a whole file -> (reads u8[], read_offs u64[n+1], headers u8[], header_offs u64[n+1])."""
    h = C.c_void_p()
    _check(lib().bgr_readset_load_parallel(path.encode(), int(fastq), k, threads, chunk_bytes, C.byref(h)))
    try:
        n = lib().bgr_readset_count(h)
        r, ro, hd, ho = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().bgr_readset_view(h, C.byref(r), C.byref(ro), C.byref(hd), C.byref(ho)))
        roffs = np.ctypeslib.as_array(C.cast(ro, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        hoffs = np.ctypeslib.as_array(C.cast(ho, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        def _bytes(ptr, total):  # an empty vector hands out a null pointer
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(total,)).copy() if total else np.zeros(0, np.uint8)
        reads = _bytes(r, int(roffs[n]))
        heads = _bytes(hd, int(hoffs[n]))
        return reads, roffs, heads, hoffs
    finally:
        lib().bgr_readset_destroy(h)
