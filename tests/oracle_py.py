"""ctypes view of oracle/liboracle.so -- the CPU restatement used as the CHECKER.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "bgreat_oracle.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = C.CDLL(_SO)
        vp, u64 = C.c_void_p, C.c_uint64
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.c_int, vp, vp, u64]
        L.orc_create_from_file.restype = vp
        L.orc_create_from_file.argtypes = [C.c_char_p, C.c_int]
        L.orc_create2.restype = vp
        L.orc_create2.argtypes = [C.c_int, vp, vp, u64, C.c_int]
        L.orc_create_from_file2.restype = vp
        L.orc_create_from_file2.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.orc_anchor_lookup.restype = C.c_int
        L.orc_anchor_lookup.argtypes = [vp, u64, vp, vp, vp]
        L.orc_destroy.argtypes = [vp]
        L.orc_unitig_count.restype = u64
        L.orc_unitig_count.argtypes = [vp]
        L.orc_align.restype = C.c_int64
        L.orc_align.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, u64, vp, u64, vp, vp]
        L.orc_counters.argtypes = [vp, vp]
        L.orc_reset_counters.argtypes = [vp]
        L.orc_work.restype = C.c_int
        L.orc_work.argtypes = [vp, vp, C.c_int]
        L.orc_alg_bytes.restype = C.c_double
        L.orc_alg_bytes.argtypes = [vp]
        L.orc_parse_file.restype = C.c_int
        L.orc_parse_file.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


WORK_FIELDS = ["reads", "read_bases", "lookups", "probes_all", "probes_nonempty", "level_hits", "rank_words", "final_finds",
               "tab_records", "unitig_fetch", "mm_calls", "mm_bases", "path_ints"]


class Oracle:
    def __init__(self, k, seqs=None, offsets=None, fasta=None, anchors=False):
        """anchors=True: index with dogMode (-G), needed by mode 2 and anchor_lookup."""
        if fasta is not None:
            self.h = lib().orc_create_from_file2(fasta.encode(), k, int(anchors))
        else:
            seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            self.h = lib().orc_create2(k, seqs.ctypes.data, offsets.ctypes.data, len(offsets) - 1, int(anchors))
        if not self.h:
            raise RuntimeError("oracle: cannot create")

    def anchor_lookup(self, kmer):
        """(index, unitig, offset) of anchorsMPHF.lookup(kmer) / anchorsPosition; index None for ULLONG_MAX."""
        idx, u, o = C.c_uint64(), C.c_uint32(), C.c_uint32()
        rc = lib().orc_anchor_lookup(self.h, int(kmer), C.byref(idx), C.byref(u), C.byref(o))
        assert rc >= 0
        return (None, 0, 0) if rc == 0 else (idx.value, u.value, o.value)

    def align(self, reads, offsets, m=2, effort=2, mode=0, partial=False):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = int(offsets[-1]) + 8 * n + 8
        paths = np.empty(cap, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(max(n, 1), dtype=np.uint8)
        w = lib().orc_align(self.h, mode, m, effort, int(partial), reads.ctypes.data, offsets.ctypes.data, n, paths.ctypes.data, cap,
                            poffs.ctypes.data, status.ctypes.data)
        assert w >= 0
        return paths[:w].copy(), poffs, status[:n]

    def counters(self):
        out = np.zeros(5, dtype=np.uint64)
        lib().orc_counters(self.h, out.ctypes.data)
        return dict(zip(["reads", "no_overlap", "aligned", "not_aligned", "overlaps"], (int(x) for x in out)))

    def reset(self):
        lib().orc_reset_counters(self.h)

    def work(self):
        out = np.zeros(16, dtype=np.uint64)
        n = lib().orc_work(self.h, out.ctypes.data, 16)
        return dict(zip(WORK_FIELDS, (int(x) for x in out[:n])))

    def alg_bytes(self):
        return lib().orc_alg_bytes(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None


def parse_file(path, k, fastq=False):
    sizes = np.zeros(3, dtype=np.uint64)
    assert lib().orc_parse_file(path.encode(), int(fastq), k, sizes.ctypes.data, None, None, None, None) == 0
    n, rb, hb = (int(x) for x in sizes)
    reads = np.empty(max(rb, 1), dtype=np.uint8)
    heads = np.empty(max(hb, 1), dtype=np.uint8)
    roffs = np.empty(n + 1, dtype=np.uint64)
    hoffs = np.empty(n + 1, dtype=np.uint64)
    assert lib().orc_parse_file(path.encode(), int(fastq), k, sizes.ctypes.data, reads.ctypes.data, roffs.ctypes.data, heads.ctypes.data, hoffs.ctypes.data) == 0
    return reads[:rb], roffs, heads[:hb], hoffs
