"""ctypes binding of tools/libsynth.so -- seeded synthetic graph + reads.  Test/bench input only."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsynth.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "synth.cpp")):
            subprocess.check_call(["make", "-C", _HERE, "libsynth.so"])
        L = C.CDLL(so)
        L.syn_create.restype = C.c_void_p
        L.syn_create.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.syn_destroy.argtypes = [C.c_void_p]
        L.syn_unitig_count.restype = C.c_uint64
        L.syn_unitig_count.argtypes = [C.c_void_p]
        L.syn_unitig_bases.restype = C.c_uint64
        L.syn_unitig_bases.argtypes = [C.c_void_p]
        L.syn_unitigs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.syn_write_unitigs.argtypes = [C.c_void_p, C.c_char_p]
        L.syn_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        L.syn_read_starts.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        L.syn_write_reads.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
        L.syn_write_reads_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class Synth:
    """Random genome of G bases with a variant site every ~d bases (`alleles` alleles each), cut into unitigs."""

    def __init__(self, G, d, alleles, k, seed):
        self.k = k
        self.h = lib().syn_create(G, d, alleles, k, seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib().syn_destroy(self.h)
            self.h = None

    def unitigs(self):
        """-> (seqs uint8[total], offs uint64[n+1])"""
        n = lib().syn_unitig_count(self.h)
        tot = lib().syn_unitig_bases(self.h)
        seqs = np.empty(tot, dtype=np.uint8)
        offs = np.empty(n + 1, dtype=np.uint64)
        lib().syn_unitigs(self.h, seqs.ctypes.data, offs.ctypes.data)
        return seqs, offs

    def write_unitigs(self, path):
        assert lib().syn_write_unitigs(self.h, path.encode()) == 0

    def reads(self, first, n, L, max_sub, seed, threads=8):
        """-> (reads uint8[n*L], offs uint64[n+1]) fixed-length reads first..first+n"""
        out = np.empty(n * L, dtype=np.uint8)
        lib().syn_reads(self.h, first, n, L, max_sub, seed, out.ctypes.data, threads)
        offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
        return out, offs

    def read_starts(self, first, n, L, seed):
        """-> uint64[n]: the genome position each of reads first..first+n was drawn from (diagnostics: locality experiments)"""
        out = np.empty(n, dtype=np.uint64)
        lib().syn_read_starts(self.h, first, n, L, seed, out.ctypes.data)
        return out

    def write_reads(self, path, first, n, L, max_sub, seed, fastq=False, threads=1):
        if threads > 1:
            assert lib().syn_write_reads_mt(self.h, path.encode(), first, n, L, max_sub, seed, int(fastq), threads) == 0
        else:
            assert lib().syn_write_reads(self.h, path.encode(), first, n, L, max_sub, seed, int(fastq)) == 0
