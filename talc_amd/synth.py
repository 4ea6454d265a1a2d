"""ctypes binding of libtalc_synth.so — deterministic synthetic TALC inputs."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None


class SynthSpec(C.Structure):
    _fields_ = [
        ("target_kmers", C.c_uint64),
        ("k", C.c_uint32),
        ("mean_len", C.c_uint32),
        ("sd_len", C.c_uint32),
        ("min_len", C.c_uint32),
        ("mixed_lengths", C.c_int32),
        ("sub_rate", C.c_double),
        ("ins_rate", C.c_double),
        ("del_rate", C.c_double),
        ("frac_short", C.c_double),
        ("frac_random", C.c_double),
        ("extra_error_frac", C.c_double),
        ("count1_frac", C.c_double),
        ("junction_period", C.c_int32),
        ("seed", C.c_uint64),
        ("paralog_frac", C.c_double),
        ("paralog_div", C.c_double),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_build.OUT, "libtalc_synth.so")
        if not os.path.exists(path):
            _build.build_synth()
        L = C.CDLL(path)
        L.synth_create.restype = C.c_void_p
        L.synth_create.argtypes = [C.POINTER(SynthSpec)]
        L.synth_destroy.argtypes = [C.c_void_p]
        for f in ("synth_transcriptome_bases", "synth_num_transcripts", "synth_dump_size", "synth_junction_size"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.synth_dump_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_dump_release.argtypes = [C.c_void_p]
        L.synth_junction_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.synth_write_dump.argtypes = [C.c_void_p, C.c_char_p]
        L.synth_write_junctions.argtypes = [C.c_void_p, C.c_char_p]
        L.synth_write_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32]
        L.synth_spec_default.argtypes = [C.POINTER(SynthSpec)]
        _LIB = L
    return _LIB


class Synth:
    """A synthetic transcriptome + k-mer dump + read generator."""

    def __init__(self, target_kmers=5_000_000, k=21, seed=0, **kw):
        L = lib()
        sp = SynthSpec()
        L.synth_spec_default(C.byref(sp))
        sp.target_kmers = target_kmers
        sp.k = k
        sp.seed = seed
        for key, val in kw.items():
            if not hasattr(sp, key):
                raise TypeError("unknown synth option " + key)
            setattr(sp, key, val)
        self.spec = sp
        self.k = k
        self._h = C.c_void_p(L.synth_create(C.byref(sp)))

    def close(self):
        if self._h:
            lib().synth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dump_arrays(self, release=True):
        """(keys u64 packed 2-bit k-mers, counts u32) in dump line order (shuffled, unfiltered)."""
        L = lib()
        n = L.synth_dump_size(self._h)
        keys = np.empty(n, dtype=np.uint64)
        counts = np.empty(n, dtype=np.uint32)
        L.synth_dump_arrays(self._h, keys.ctypes.data, counts.ctypes.data)
        if release:
            L.synth_dump_release(self._h)
        return keys, counts

    def junction_arrays(self):
        L = lib()
        n = L.synth_junction_size(self._h)
        keys = np.empty(n, dtype=np.uint64)
        jc = np.empty(n, dtype=np.int64)
        L.synth_junction_arrays(self._h, keys.ctypes.data, jc.ctypes.data)
        return keys, jc

    def reads(self, first, n):
        """(bases uint8 ASCII concatenated, offsets u64[n+1]) for reads [first, first+n)."""
        L = lib()
        offsets = np.empty(n + 1, dtype=np.uint64)
        L.synth_reads(self._h, first, n, None, offsets.ctypes.data)
        bases = np.empty(int(offsets[n]), dtype=np.uint8)
        L.synth_reads(self._h, first, n, bases.ctypes.data, offsets.ctypes.data)
        return bases, offsets

    def read_lengths(self, first, n):
        """Lengths (int64[n]) of reads [first, first+n) without materialising them."""
        offsets = np.empty(n + 1, dtype=np.uint64)
        lib().synth_reads(self._h, first, n, None, offsets.ctypes.data)
        return np.diff(offsets.astype(np.int64))

    def write_dump(self, path):
        if lib().synth_write_dump(self._h, path.encode()) != 0:
            raise IOError(path)

    def write_junctions(self, path):
        if lib().synth_write_junctions(self._h, path.encode()) != 0:
            raise IOError(path)

    def write_fasta(self, path, first, n):
        if lib().synth_write_fasta(self._h, path.encode(), first, n) != 0:
            raise IOError(path)
