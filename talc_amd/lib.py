"""ctypes binding of libtalc_hip.so (include/talc_hip.h) — the product's C ABI.

There is no CPU fallback: every compute entry point needs a MI355X and fails loudly
(TalcError) without one.  Loading the library and resolving its symbols works on any host.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None

READ_CORRECTED, READ_SKIPPED_SHORT, READ_NO_SOLID_KMER, READ_NO_STRUCTURE, READ_ERROR = range(5)
LOG_MESSAGES = {
    READ_NO_SOLID_KMER: "No solid kmer could be found.",          # main.cpp:294
    READ_NO_STRUCTURE: "Unable to define convenient structure.",  # main.cpp:290
}

# every symbol include/talc_hip.h declares
ABI_SYMBOLS = [
    "talc_abi_version", "talc_last_error", "talc_params_default", "talc_device_count", "talc_pinned_alloc", "talc_pinned_free",
    "talc_table_build", "talc_table_from_arrays", "talc_table_build_device", "talc_table_from_arrays_device",
    "talc_table_colour", "talc_table_decolour_repeats",
    "talc_table_size", "talc_table_device_bytes", "talc_table_upload", "talc_table_capacity", "talc_table_image_bytes",
    "talc_table_export_device", "talc_table_import_device", "talc_table_lookup_batch",
    "talc_table_next_counts_batch", "talc_table_lookup_host_batch", "talc_table_destroy",
    "talc_ctx_create", "talc_ctx_destroy", "talc_batch_create", "talc_batch_destroy",
    "talc_batch_coverage", "talc_batch_fetch_coverage", "talc_batch_num_kmers", "talc_batch_num_bases",
    "talc_batch_correct", "talc_batch_corrected_bytes", "talc_batch_fetch_corrected", "talc_batch_fetch_read_stats",
    "talc_batch_copy_corrected_device", "talc_correct_batch",
    "talc_ctx_get_timing", "talc_batch_trace_read", "talc_test_dp",
]


class TalcError(RuntimeError):
    pass


class Params(C.Structure):
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


class Timing(C.Structure):
    _fields_ = [
        ("encode_ms", C.c_float),
        ("coverage_ms", C.c_float),
        ("structure_ms", C.c_float),
        ("search_ms", C.c_float),
        ("emit_ms", C.c_float),
        ("retry_ms", C.c_float),
        ("n_kmers", C.c_uint64),
        ("n_bases", C.c_uint64),
        ("n_trail_steps", C.c_uint64),
        ("n_dp_cells", C.c_uint64),
        ("n_retried", C.c_uint32),
        ("n_failed", C.c_uint32),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def lib_path():
    # TALC_LIB selects another build of the same library (e.g. the -DTALC_PROF diagnostic build)
    return os.environ.get("TALC_LIB") or os.path.join(_build.OUT, "libtalc_hip.so")


def lib():
    """Load libtalc_hip.so (built in-tree by talc_amd.build); raises if it is missing."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise TalcError("libtalc_hip.so is missing: run `python -m talc_amd.build` (needs hipcc); "
                            "there is no CPU fallback for the correction path")
        L = C.CDLL(path)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.talc_last_error.restype = C.c_char_p
        L.talc_params_default.argtypes = [C.POINTER(Params)]
        L.talc_pinned_alloc.restype = vp
        L.talc_pinned_alloc.argtypes = [u64]
        L.talc_pinned_free.argtypes = [vp]
        L.talc_table_build.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Params), C.POINTER(vp), vp]
        L.talc_table_from_arrays.argtypes = [vp, vp, u64, C.POINTER(Params), C.POINTER(vp)]
        L.talc_table_build_device.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Params), i32, C.POINTER(vp), vp]
        L.talc_table_from_arrays_device.argtypes = [vp, vp, u64, C.POINTER(Params), i32, C.POINTER(vp)]
        L.talc_table_colour.argtypes = [vp, vp, vp, u64]
        L.talc_table_decolour_repeats.argtypes = [vp]
        L.talc_table_size.restype = u64
        L.talc_table_size.argtypes = [vp]
        L.talc_table_device_bytes.restype = u64
        L.talc_table_device_bytes.argtypes = [vp]
        L.talc_table_upload.argtypes = [vp, i32]
        L.talc_table_capacity.restype = u64
        L.talc_table_capacity.argtypes = [vp]
        L.talc_table_image_bytes.restype = u64
        L.talc_table_image_bytes.argtypes = [vp]
        L.talc_table_export_device.argtypes = [vp, i32, vp, vp]
        L.talc_table_import_device.argtypes = [C.POINTER(Params), u64, u64, vp, vp, i32, C.POINTER(vp)]
        L.talc_table_lookup_batch.argtypes = [vp, i32, vp, u64, vp, vp]
        L.talc_table_next_counts_batch.argtypes = [vp, i32, vp, u64, i32, vp, vp]
        L.talc_table_lookup_host_batch.argtypes = [vp, vp, u64, vp, vp]
        L.talc_table_destroy.argtypes = [vp]
        L.talc_ctx_create.argtypes = [vp, C.POINTER(Params), i32, C.POINTER(vp)]
        L.talc_ctx_destroy.argtypes = [vp]
        L.talc_batch_create.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.talc_batch_destroy.argtypes = [vp]
        L.talc_batch_coverage.argtypes = [vp, vp]
        L.talc_batch_fetch_coverage.argtypes = [vp, vp, vp, vp, vp, vp]
        L.talc_batch_num_kmers.restype = u64
        L.talc_batch_num_kmers.argtypes = [vp]
        L.talc_batch_num_bases.restype = u64
        L.talc_batch_num_bases.argtypes = [vp]
        L.talc_batch_correct.argtypes = [vp, vp]
        L.talc_batch_corrected_bytes.restype = u64
        L.talc_batch_corrected_bytes.argtypes = [vp]
        L.talc_batch_fetch_corrected.argtypes = [vp, vp, vp, u64, vp, vp]
        L.talc_batch_copy_corrected_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.talc_batch_fetch_read_stats.argtypes = [vp, vp, vp]
        L.talc_correct_batch.argtypes = [vp, vp, vp, u32, vp, u64, vp, vp]
        L.talc_ctx_get_timing.argtypes = [vp, C.POINTER(Timing)]
        L.talc_batch_trace_read.restype = C.c_int64
        L.talc_batch_trace_read.argtypes = [vp, vp, u32, vp, u64]
        L.talc_test_dp.argtypes = [vp, i32, C.c_char_p, i32, C.c_char_p, i32, i32, i32, i32, i32, vp]
        _LIB = L
    return _LIB


WARN_READ_ERRORS = 1   # TALC_WARN_READ_ERRORS: the batch is valid, some reads carry READ_ERROR


def _chk(rc):
    """Negative codes are errors; positive ones (TALC_WARN_READ_ERRORS) are returned to the caller."""
    if rc < 0:
        raise TalcError("libtalc_hip error %d: %s" % (rc, lib().talc_last_error().decode(errors="replace")))
    return rc


def default_params(**kw):
    p = Params()
    _chk(lib().talc_params_default(C.byref(p)))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown parameter " + k)
        setattr(p, k, v)
    return p


def device_count():
    return int(lib().talc_device_count())


class PinnedArray:
    """A uint8 numpy view over page-locked host memory of the library (talc_pinned_alloc)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = lib().talc_pinned_alloc(max(self.nbytes, 1))
        if not self._p:
            raise TalcError("talc_pinned_alloc(%d) failed: %s" % (nbytes, lib().talc_last_error().decode(errors="replace")))
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self._p))[: self.nbytes]

    def close(self):
        if self._p:
            self.array = None
            lib().talc_pinned_free(C.c_void_p(self._p))
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Table:
    """The SR k-mer table (replaces buildCDBG, Jellyfish.cpp:236-295)."""

    def __init__(self, handle, params):
        self._h = handle
        self.params = params

    @classmethod
    def from_arrays(cls, kmers, counts, params, device=None):
        """device=None: host builder; device=d: insertion on GPU d (same content)."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        h = C.c_void_p()
        if device is None:
            _chk(lib().talc_table_from_arrays(kmers.ctypes.data, counts.ctypes.data, len(kmers), C.byref(params), C.byref(h)))
        else:
            _chk(lib().talc_table_from_arrays_device(kmers.ctypes.data, counts.ctypes.data, len(kmers), C.byref(params),
                                                     int(device), C.byref(h)))
        return cls(h, params)

    @classmethod
    def from_files(cls, dump, junctions, params, device=None):
        h = C.c_void_p()
        st = np.zeros(3, dtype=np.int64)
        if device is None:
            _chk(lib().talc_table_build(dump.encode(), junctions.encode() if junctions else None, C.byref(params), C.byref(h), st.ctypes.data))
        else:
            _chk(lib().talc_table_build_device(dump.encode(), junctions.encode() if junctions else None, C.byref(params),
                                               int(device), C.byref(h), st.ctypes.data))
        t = cls(h, params)
        t.build_stats = st
        return t

    def colour(self, jkmers, jcounts):
        jkmers = np.ascontiguousarray(jkmers, dtype=np.uint64)
        jcounts = np.ascontiguousarray(jcounts, dtype=np.int64)
        _chk(lib().talc_table_colour(self._h, jkmers.ctypes.data, jcounts.ctypes.data, len(jkmers)))

    def decolour_repeats(self):
        _chk(lib().talc_table_decolour_repeats(self._h))

    def __len__(self):
        return int(lib().talc_table_size(self._h))

    @property
    def device_bytes(self):
        return int(lib().talc_table_device_bytes(self._h))

    def upload(self, device=0):
        _chk(lib().talc_table_upload(self._h, device))

    @property
    def capacity(self):
        return int(lib().talc_table_capacity(self._h))

    @property
    def image_bytes(self):
        """Bytes of each of the two bucket tables (RIGHT, LEFT) of the device image."""
        return int(lib().talc_table_image_bytes(self._h))

    def export_device(self, device, ptr_right, ptr_left):
        """Copy the image on `device` into two caller-owned device buffers of image_bytes each."""
        _chk(lib().talc_table_export_device(self._h, int(device), C.c_void_p(ptr_right), C.c_void_p(ptr_left)))

    @classmethod
    def import_device(cls, params, capacity, n_kmers, ptr_right, ptr_left, device):
        """A table on `device` from an exported image (two device buffers; they are copied)."""
        h = C.c_void_p()
        _chk(lib().talc_table_import_device(C.byref(params), int(capacity), int(n_kmers), C.c_void_p(ptr_right),
                                            C.c_void_p(ptr_left), int(device), C.byref(h)))
        return cls(h, params)

    def lookup(self, kmers, device=0):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        c = np.empty(len(kmers), dtype=np.uint32)
        j = np.empty(len(kmers), dtype=np.uint32)
        _chk(lib().talc_table_lookup_batch(self._h, device, kmers.ctypes.data, len(kmers), c.ctypes.data, j.ctypes.data))
        return c, j

    def lookup_host(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        c = np.empty(len(kmers), dtype=np.uint32)
        j = np.empty(len(kmers), dtype=np.uint32)
        _chk(lib().talc_table_lookup_host_batch(self._h, kmers.ctypes.data, len(kmers), c.ctypes.data, j.ctypes.data))
        return c, j

    def next_counts(self, kmers, direction, device=0):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        c = np.empty((len(kmers), 4), dtype=np.uint32)
        j = np.empty((len(kmers), 4), dtype=np.uint32)
        _chk(lib().talc_table_next_counts_batch(self._h, device, kmers.ctypes.data, len(kmers), int(direction), c.ctypes.data, j.ctypes.data))
        return c, j

    def close(self):
        if self._h:
            lib().talc_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, table, params=None, device=0):
        self.table = table
        self.params = params or table.params
        self.device = device
        h = C.c_void_p()
        _chk(lib().talc_ctx_create(table._h, C.byref(self.params), device, C.byref(h)))
        self._h = h

    def timing(self):
        t = Timing()
        _chk(lib().talc_ctx_get_timing(self._h, C.byref(t)))
        return t

    def batch(self, bases, offsets):
        return Batch(self, bases, offsets)

    def test_dp(self, mode, a, b, p0=0, p1=0, p2=0, p3=0):
        out = np.zeros(12, dtype=np.int32)
        a = a if isinstance(a, bytes) else a.encode()
        b = b if isinstance(b, bytes) else b.encode()
        _chk(lib().talc_test_dp(self._h, mode, a, len(a), b, len(b), p0, p1, p2, p3, out.ctypes.data))
        return out

    def correct(self, bases, offsets, out=None):
        """One-shot: returns (out uint8 ASCII, out_offsets, status)."""
        b = self.batch(bases, offsets)
        try:
            b.correct()
            return b.fetch_corrected(out)
        finally:
            b.close()

    def close(self):
        if self._h:
            lib().talc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A batch of reads resident in HBM."""

    def __init__(self, ctx, bases, offsets):
        self.ctx = ctx
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n_reads = len(offsets) - 1
        h = C.c_void_p()
        _chk(lib().talc_batch_create(ctx._h, bases.ctypes.data, offsets.ctypes.data, self.n_reads, C.byref(h)))
        self._h = h

    @property
    def n_kmers(self):
        return int(lib().talc_batch_num_kmers(self._h))

    @property
    def n_bases(self):
        return int(lib().talc_batch_num_bases(self._h))

    def coverage(self):
        _chk(lib().talc_batch_coverage(self.ctx._h, self._h))

    def fetch_coverage(self):
        n = self.n_kmers
        c = np.empty(n, dtype=np.uint32)
        j = np.empty(n, dtype=np.uint32)
        ko = np.empty(self.n_reads + 1, dtype=np.uint64)
        nin = np.empty(self.n_reads, dtype=np.int32)
        _chk(lib().talc_batch_fetch_coverage(self.ctx._h, self._h, c.ctypes.data, j.ctypes.data, ko.ctypes.data, nin.ctypes.data))
        return c, j, ko, nin

    def correct(self):
        """0, or WARN_READ_ERRORS when some reads exhausted the device scratch (status READ_ERROR, passed through)."""
        return _chk(lib().talc_batch_correct(self.ctx._h, self._h))

    def fetch_corrected(self, out=None):
        """(records uint8 ASCII, offsets, status); `out`: a caller's uint8 buffer to fill (e.g. PinnedArray.array)."""
        total = int(lib().talc_batch_corrected_bytes(self._h))
        if out is None or len(out) < total:
            out = np.empty(max(total, 1), dtype=np.uint8)
        oo = np.empty(self.n_reads + 1, dtype=np.uint64)
        st = np.empty(self.n_reads, dtype=np.int32)
        _chk(lib().talc_batch_fetch_corrected(self.ctx._h, self._h, out.ctypes.data, total, oo.ctypes.data, st.ctypes.data))
        return out[:total], oo, st

    def fetch_read_stats(self):
        """int64[n, 5]: {row written, raw length, IN-region span, IN regions, corrected length} (Read.cpp:418-433)."""
        st = np.zeros((self.n_reads, 5), dtype=np.int64)
        _chk(lib().talc_batch_fetch_read_stats(self.ctx._h, self._h, st.ctypes.data))
        return st

    @property
    def corrected_bytes(self):
        return int(lib().talc_batch_corrected_bytes(self._h))

    def copy_corrected_to_device(self, device_ptr, capacity):
        """Device-to-device copy of the dense corrected records into a caller-owned buffer."""
        oo = np.empty(self.n_reads + 1, dtype=np.uint64)
        st = np.empty(self.n_reads, dtype=np.int32)
        _chk(lib().talc_batch_copy_corrected_device(self.ctx._h, self._h, C.c_void_p(device_ptr), capacity, oo.ctypes.data, st.ctypes.data))
        return oo, st

    def trace(self, read_index):
        cap = 1 << 22
        while True:
            buf = C.create_string_buffer(cap)
            need = lib().talc_batch_trace_read(self.ctx._h, self._h, read_index, buf, cap)
            if need < 0:
                _chk(int(need))
            if need <= cap:
                return buf.value.decode()
            cap = int(need) + 16

    def close(self):
        if self._h:
            lib().talc_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
