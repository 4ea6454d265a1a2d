"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference algorithm (oracle/talc_oracle.cpp, parity
unpinned).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OrcParams(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("min_count", C.c_uint32),
        ("alpha", C.c_double),
        ("window_size", C.c_uint32),
        ("sr_error_rate", C.c_double),
        ("min_inner_score", C.c_double),
        ("min_border_score", C.c_double),
        ("max_nb_competing_paths", C.c_uint32),
        ("use_junctions", C.c_int32),
        ("reverse", C.c_int32),
        ("min_start_anchors", C.c_uint32),
        ("max_start_anchors", C.c_uint32),
        ("max_in_count", C.c_uint32),
        ("max_nb_border_paths", C.c_uint32),
        ("max_nb_inner_paths", C.c_uint32),
        ("check_interval", C.c_uint32),
        ("allowed_failure_rate", C.c_double),
        ("max_nb_border_failures", C.c_int32),
        ("coloured_count_thr", C.c_uint32),
        ("max_border_length", C.c_uint32),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_table_new.restype = C.c_void_p
        L.orc_table_new.argtypes = [C.c_int]
        L.orc_table_free.argtypes = [C.c_void_p]
        L.orc_table_size.restype = C.c_uint64
        L.orc_table_size.argtypes = [C.c_void_p]
        L.orc_table_build.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(OrcParams), C.c_void_p]
        L.orc_table_insert_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcParams), C.c_int]
        L.orc_table_colour_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcParams)]
        L.orc_table_decolour.argtypes = [C.c_void_p, C.POINTER(OrcParams)]
        L.orc_table_lookup_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_next_counts.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_coverage.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_structure.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.orc_correct_batch.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_correct_batch_stats.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_trace_read.restype = C.c_int64
        L.orc_trace_read.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_ub_counters.argtypes = [C.c_void_p]
        L.orc_global_alignment.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 7
        L.orc_local_alignment.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 3
        L.orc_extend_seed.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p] + [C.c_int] * 5
        L.orc_seed_and_extension.restype = C.c_double
        L.orc_seed_and_extension.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_is_expected_by_model.argtypes = [C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_int]
        L.orc_is_expected_by_last_node.argtypes = [C.POINTER(OrcParams), C.c_uint32, C.c_uint32]
        L.orc_tag_next_nodes.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_gardening.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
        L.orc_seq_error_threshold.restype = C.c_double
        L.orc_seq_error_threshold.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_uint64]
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        _LIB = L
    return _LIB


def params(**kw):
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(k)
        setattr(p, k, v)
    return p


STATUS_NAMES = {0: "CORRECTED", 1: "SKIPPED_SHORT", 2: "NO_SOLID_KMER", 3: "NO_STRUCTURE"}


class OracleTable:
    MAP, FLAT = 0, 1

    def __init__(self, p, backend=FLAT):
        self.p = p
        self._h = C.c_void_p(lib().orc_table_new(backend))

    def close(self):
        if self._h:
            lib().orc_table_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().orc_table_size(self._h))

    def build_from_files(self, dump, jdump=None):
        st = np.zeros(3, dtype=np.int64)
        lib().orc_table_build(self._h, dump.encode(), (jdump or "").encode(), C.byref(self.p), st.ctypes.data)
        return st

    def insert_packed(self, keys, counts, sorted_hint=False):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        lib().orc_table_insert_packed(self._h, keys.ctypes.data, counts.ctypes.data, len(keys), C.byref(self.p), int(sorted_hint))

    def colour_packed(self, jkeys, jcounts):
        jkeys = np.ascontiguousarray(jkeys, dtype=np.uint64)
        jcounts = np.ascontiguousarray(jcounts, dtype=np.int64)
        lib().orc_table_colour_packed(self._h, jkeys.ctypes.data, jcounts.ctypes.data, len(jkeys), C.byref(self.p))

    def decolour(self):
        lib().orc_table_decolour(self._h, C.byref(self.p))

    def lookup_packed(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        c = np.empty(len(keys), dtype=np.uint32)
        j = np.empty(len(keys), dtype=np.uint32)
        lib().orc_table_lookup_packed(self._h, keys.ctypes.data, len(keys), self.p.k, c.ctypes.data, j.ctypes.data)
        return c, j

    def next_counts(self, kmer, direction):
        c = np.empty(4, dtype=np.uint32)
        j = np.empty(4, dtype=np.uint32)
        lib().orc_next_counts(self._h, C.byref(self.p), kmer.encode(), int(direction), c.ctypes.data, j.ctypes.data)
        return c, j

    def coverage(self, seq):
        b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
        n = max(0, len(b) - self.p.k + 1)
        c = np.zeros(n, dtype=np.uint32)
        j = np.zeros(n, dtype=np.uint32)
        nin = lib().orc_coverage(self._h, C.byref(self.p), b.ctypes.data, len(b), c.ctypes.data, j.ctypes.data)
        return c, j, nin

    def structure(self, seq, max_regions=100000):
        b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
        reg = np.zeros(2 * max_regions, dtype=np.uint32)
        thr = C.c_double()
        ok = C.c_int32()
        n = lib().orc_structure(self._h, C.byref(self.p), b.ctypes.data, len(b), reg.ctypes.data, max_regions, C.byref(thr), C.byref(ok))
        return reg[: 2 * n].reshape(-1, 2).copy(), thr.value, bool(ok.value)

    def correct_batch(self, bases, offsets, nthreads=1):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = int(len(bases)) * 2 + 1024
        while True:
            out = np.empty(cap, dtype=np.uint8)
            oo = np.empty(n + 1, dtype=np.uint64)
            st = np.empty(n, dtype=np.int32)
            rc = lib().orc_correct_batch(self._h, C.byref(self.p), bases.ctypes.data, offsets.ctypes.data, n,
                                         out.ctypes.data, cap, oo.ctypes.data, st.ctypes.data, nthreads)
            if rc == 0:
                return out[: int(oo[n])].copy(), oo, st
            cap = int(oo[n]) + 16

    def correct_batch_stats(self, bases, offsets, nthreads=1):
        """correct_batch plus the rows of Read::outputBasicReadStats: int64[n, 5]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = int(len(bases)) * 2 + 1024
        while True:
            out = np.empty(cap, dtype=np.uint8)
            oo = np.empty(n + 1, dtype=np.uint64)
            st = np.empty(n, dtype=np.int32)
            rows = np.zeros((n, 5), dtype=np.int64)
            rc = lib().orc_correct_batch_stats(self._h, C.byref(self.p), bases.ctypes.data, offsets.ctypes.data, n,
                                               out.ctypes.data, cap, oo.ctypes.data, st.ctypes.data, nthreads, rows.ctypes.data)
            if rc == 0:
                return out[: int(oo[n])].copy(), oo, st, rows
            cap = int(oo[n]) + 16

    def trace(self, seq, steps=False):
        b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
        cap = 1 << 20
        while True:
            buf = C.create_string_buffer(cap)
            need = lib().orc_trace_read(self._h, C.byref(self.p), b.ctypes.data, len(b), int(steps), buf, cap)
            if need <= cap:
                return buf.value.decode()
            cap = int(need) + 16


def ub_counters():
    a = np.zeros(3, dtype=np.int64)
    lib().orc_ub_counters(a.ctypes.data)
    return dict(infixClamped=int(a[0]), seedTooShort=int(a[1]), gardeningOOB=int(a[2]))
