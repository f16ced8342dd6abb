"""Python binding of libemspec's C ABI (include/emspec.h) via ctypes.

Used by tests/, bench.py and __graft_entry__.py.  The production host side is
the Node addon in em-spec_amd/js/; this module exposes the same calls for the
Python tooling.  There is NO fallback: if libemspec.so is missing or no gfx950
device is present, loading / Engine() raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)                 # em-spec_amd/
LIB_PATH = os.path.join(ROOT, "libemspec.so")
# the same sources built with -DEMSPEC_DIAG: adds include/emspec_debug.h, the stamped / A-B kernel builds and the
# EMSPEC_* environment switches.  Tools and the tests that probe internals load this one; the product never does.
DIAG_LIB_PATH = os.path.join(ROOT, "libemspec_diag.so")

ABI_VERSION = 2
OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_STATE, ERR_COMM = -1, -2, -3, -4, -5, -6
MODE_FAST, MODE_EXACT = 0, 1
COMM_ID_BYTES = 128
GATHER_LOOPBACK = 1
GATHER_PACKED = 2

SYMBOLS = [
    "emspec_default_config", "emspec_create", "emspec_destroy", "emspec_last_error", "emspec_set_colormap",
    "emspec_num_columns", "emspec_latency_columns", "emspec_column", "emspec_column_flush", "emspec_reset",
    "emspec_batch", "emspec_batch_device", "emspec_parity_dump", "emspec_parity_dump_device",
    "emspec_get_tables", "emspec_device_arch", "emspec_uses_fused", "emspec_set_row_edges_hz",
    "emspec_get_row_edges_hz", "emspec_host_alloc", "emspec_host_free", "emspec_set_display",
    "emspec_push_samples", "emspec_push_columns", "emspec_warped_edges_hz", "emspec_make_colormap",
    "emspec_comm_unique_id", "emspec_comm_init", "emspec_comm_destroy", "emspec_comm_rank", "emspec_comm_world",
    "emspec_gather_columns", "emspec_wire_bound", "emspec_wire_pack", "emspec_wire_unpack", "emspec_batch_gather",
    "emspec_parity_dump_exact", "emspec_gather_packed_layout", "emspec_mode", "emspec_build_info", "emspec_device_status",
    "emspec_comm_set_timeout", "emspec_batch_packed", "emspec_wire_unpack_host",
    "emspec_columns", "emspec_columns_flush", "emspec_push_columns_multi", "emspec_push_samples_multi",
    "emspec_reset_stream", "emspec_live_streams",
]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("rows", C.c_int32), ("mode", C.c_int32),
                ("sample_rate", C.c_float), ("fmin_hz", C.c_float), ("fmax_hz", C.c_float), ("gain", C.c_float),
                ("db_top", C.c_float), ("db_range", C.c_float), ("gate_db", C.c_float), ("power_floor", C.c_float)]


class Out(C.Structure):
    _fields_ = [("db", C.c_void_p), ("rgba", C.c_void_p), ("index", C.c_void_p)]


def wire_unpack_host(image, columns, rows, out=None, diag=False):
    """Expand one wire image (numpy uint8) into palette indices [columns, rows] on the host's own cores
    (emspec_wire_unpack_host: plain C, no device, no engine)."""
    image = np.ascontiguousarray(image, np.uint8)
    if out is None:
        out = np.empty((columns, rows), np.uint8)
    assert out.dtype == np.uint8 and out.flags.c_contiguous and out.size == columns * rows
    lib = load(diag=diag)
    rc = lib.emspec_wire_unpack_host(_np_ptr(image), C.c_int64(image.size), C.c_int64(columns), C.c_int32(rows), _np_ptr(out))
    if rc != 0:
        raise EmspecError(rc, "emspec_wire_unpack_host: the image does not match (columns, rows) or is damaged")
    return out


class EmspecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"emspec error {code}: {msg}")
        self.code = code


_libs = {}


def load(diag=False):
    """Load libemspec.so (or, diag=True, libemspec_diag.so), built in-tree by __graft_entry__.build().
    Raises if absent: there is no fallback."""
    if diag in _libs:
        return _libs[diag]
    path = DIAG_LIB_PATH if diag else LIB_PATH
    # PyTorch ships its own HIP runtime.  If libemspec.so (linked against /opt/rocm's) is loaded first, a later
    # `import torch` brings a second runtime into the process and finds no device ("No HIP GPUs are available"); with
    # torch first, libemspec's libamdhip64 dependency resolves to the copy already loaded.  The Python tooling (tests,
    # bench, smoke) uses torch for device buffers, so settle the order here; hosts without torch are unaffected.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.emspec_last_error.restype = C.c_char_p
    lib.emspec_last_error.argtypes = [C.c_void_p]
    lib.emspec_device_arch.restype = C.c_char_p
    lib.emspec_device_arch.argtypes = [C.c_void_p]
    lib.emspec_num_columns.restype = C.c_int64
    lib.emspec_num_columns.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.emspec_latency_columns.restype = C.c_int32
    lib.emspec_latency_columns.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.emspec_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.emspec_destroy.argtypes = [C.c_void_p]
    lib.emspec_destroy.restype = None
    lib.emspec_reset.argtypes = [C.c_void_p]
    lib.emspec_set_colormap.argtypes = [C.c_void_p, C.c_void_p]
    lib.emspec_column.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.POINTER(C.c_int64)]
    lib.emspec_column_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    lib.emspec_push_columns.restype = C.c_int64
    lib.emspec_push_columns.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    lib.emspec_push_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.emspec_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(Out)]
    lib.emspec_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emspec_parity_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emspec_parity_dump_device.argtypes = lib.emspec_parity_dump.argtypes + [C.c_void_p]
    lib.emspec_parity_dump_exact.argtypes = lib.emspec_parity_dump.argtypes + [C.c_void_p]
    lib.emspec_uses_fused.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.emspec_set_row_edges_hz.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.emspec_get_row_edges_hz.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.emspec_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.emspec_host_free.argtypes = [C.c_void_p]
    lib.emspec_host_free.restype = None
    lib.emspec_set_display.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.emspec_get_tables.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.emspec_comm_unique_id.argtypes = [C.c_void_p]
    lib.emspec_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.emspec_comm_destroy.argtypes = [C.c_void_p]
    lib.emspec_comm_rank.argtypes = [C.c_void_p]
    lib.emspec_comm_world.argtypes = [C.c_void_p]
    lib.emspec_gather_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_uint32,
                                          C.c_void_p, C.POINTER(C.c_int64)]
    lib.emspec_gather_packed_layout.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.emspec_batch_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.emspec_wire_bound.restype = C.c_int64
    lib.emspec_wire_bound.argtypes = [C.c_int64, C.c_int32]
    lib.emspec_wire_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    lib.emspec_wire_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.emspec_mode.argtypes = [C.c_void_p]
    lib.emspec_mode.restype = C.c_int32
    lib.emspec_build_info.argtypes = []
    lib.emspec_build_info.restype = C.c_char_p
    lib.emspec_device_status.argtypes = [C.c_void_p]
    lib.emspec_comm_set_timeout.argtypes = [C.c_void_p, C.c_double]
    lib.emspec_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_void_p]
    lib.emspec_columns_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.emspec_push_columns_multi.restype = C.c_int64
    lib.emspec_push_columns_multi.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    lib.emspec_push_samples_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    lib.emspec_reset_stream.argtypes = [C.c_void_p, C.c_int32]
    lib.emspec_live_streams.argtypes = [C.c_void_p]
    lib.emspec_live_streams.restype = C.c_int32
    _libs[diag] = lib
    return lib


def build_info(diag=False):
    """What the loaded library was built from: 'emspec abi=2 sources=<sha16> arch=gfx950' (emspec_build_info)."""
    return load(diag).emspec_build_info().decode()


class PinnedArray:
    """numpy array backed by page-locked host memory from emspec_host_alloc (freed on close/del)."""

    def __init__(self, shape, dtype):
        lib = load()
        self._lib = lib
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        self._ptr = C.c_void_p()
        rc = lib.emspec_host_alloc(n, C.byref(self._ptr))
        if rc != OK:
            raise EmspecError(rc, "emspec_host_alloc failed")
        buf = (C.c_char * n).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dt).reshape(shape)

    def close(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            self.array = None
            self._lib.emspec_host_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_config(**kw):
    cfg = Config()
    load().emspec_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def num_columns(L, n, hop):
    return int(load().emspec_num_columns(L, n, hop))


def warped_edges_hz(rows, fmin_hz, fmax_hz, low_end_boost=1.0, freq_scale=1.0):
    """rows+1 edges (Hz) of the [BUILD-DEFINED] Frequency Scale / Low-End Boost law, for Engine.set_row_edges_hz."""
    lib = load()
    out = np.empty(rows + 1, np.float32)
    lib.emspec_warped_edges_hz.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    if lib.emspec_warped_edges_hz(rows, fmin_hz, fmax_hz, low_end_boost, freq_scale, _np_ptr(out)) != OK:
        raise EmspecError(ERR_INVALID_ARG, "invalid axis parameters")
    return out


def make_colormap(brightness=0.5):
    """The reference ramp scaled by Brightness: uint8 [256][4] for Engine.set_colormap."""
    lib = load()
    out = np.empty((256, 4), np.uint8)
    lib.emspec_make_colormap.argtypes = [C.c_float, C.c_void_p]
    if lib.emspec_make_colormap(brightness, _np_ptr(out)) != OK:
        raise EmspecError(ERR_INVALID_ARG, "invalid brightness")
    return out


def comm_unique_id():
    """128-byte communicator id: rank 0 creates it and hands it to the other ranks (any host channel)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = load().emspec_comm_unique_id(buf)
    if rc != OK:
        raise EmspecError(rc, load().emspec_last_error(None).decode())
    return bytes(buf)


def wire_bound(columns, rows):
    return int(load().emspec_wire_bound(columns, rows))


def latency_columns(n, hop, reassign=True):
    return int(load().emspec_latency_columns(n, hop, int(bool(reassign))))


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Engine:
    """One engine = one HIP device + stream (not thread-safe), see emspec.h."""

    def __init__(self, cfg=None, diag=False, **kw):
        self._lib = load(diag)
        self.cfg = cfg if cfg is not None else default_config(**kw)
        h = C.c_void_p()
        rc = self._lib.emspec_create(C.byref(self.cfg), C.byref(h))
        if rc != OK:
            raise EmspecError(rc, self._lib.emspec_last_error(None).decode())
        self._h = h
        self.rows = int(self.cfg.rows)
        # the library must run the mode that was asked for (a version-1 library ignored the field: include/emspec.h)
        if self._lib.emspec_mode(h) != int(self.cfg.mode):
            self._lib.emspec_destroy(h)
            self._h = None
            raise EmspecError(ERR_STATE, "the library did not honour the requested arithmetic mode")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.emspec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise EmspecError(rc, self._lib.emspec_last_error(self._h).decode())

    @property
    def mode(self):
        return int(self._lib.emspec_mode(self._h))

    def device_status(self):
        """Synchronise the device and raise if a kernel flagged a protocol error (emspec_device_status)."""
        self._chk(self._lib.emspec_device_status(self._h))

    @property
    def arch(self):
        return self._lib.emspec_device_arch(self._h).decode()

    def fused(self, n, hop, reassign=True):
        return bool(self._lib.emspec_uses_fused(self._h, n, hop, int(bool(reassign))))

    def set_colormap(self, lut):
        lut = np.ascontiguousarray(lut, np.uint8)
        assert lut.shape == (256, 4)
        self._chk(self._lib.emspec_set_colormap(self._h, _np_ptr(lut)))

    def set_display(self, smoothing=0.0, agc_strength=0.0):
        """Temporal smoothing in [0,0.95] and adaptive-brightness strength in [0,1]; (0,0) = off."""
        self._chk(self._lib.emspec_set_display(self._h, smoothing, agc_strength))

    def set_row_edges_hz(self, edges_hz):
        """rows+1 strictly increasing edges in Hz, or None for the configured log axis."""
        if edges_hz is None:
            self._chk(self._lib.emspec_set_row_edges_hz(self._h, None, 0))
        else:
            edges_hz = np.ascontiguousarray(edges_hz, np.float32)
            self._chk(self._lib.emspec_set_row_edges_hz(self._h, _np_ptr(edges_hz), edges_hz.size))

    def row_edges_hz(self):
        out = np.empty(self.rows + 1, np.float32)
        self._chk(self._lib.emspec_get_row_edges_hz(self._h, _np_ptr(out), out.size))
        return out

    def tables(self, n):
        eb = np.empty(self.rows + 1, np.float32)
        tw = np.empty(n, np.float32)
        self._chk(self._lib.emspec_get_tables(self._h, n, _np_ptr(eb), _np_ptr(tw)))
        return tw, eb

    # -- batch, host buffers ------------------------------------------------
    def batch(self, pcm, n, hop, reassign=True, want=("db",), db_out=None):
        pcm = np.ascontiguousarray(pcm, np.float32)
        if pcm.ndim == 1:
            pcm = pcm[None]
        S, L = pcm.shape
        Cn = num_columns(L, n, hop)
        db = (db_out if db_out is not None else np.empty((S, Cn, self.rows), np.float32)) if "db" in want else None
        assert db is None or (db.shape == (S, Cn, self.rows) and db.dtype == np.float32 and db.flags.c_contiguous)
        rgba = np.empty((S, Cn, self.rows, 4), np.uint8) if "rgba" in want else None
        idx = np.empty((S, Cn, self.rows), np.uint8) if "index" in want else None
        out = Out(_np_ptr(db), _np_ptr(rgba), _np_ptr(idx))
        self._chk(self._lib.emspec_batch(self._h, _np_ptr(pcm), S, L, n, hop, int(bool(reassign)), C.byref(out)))
        return {"db": db, "rgba": rgba, "index": idx}

    def batch_packed(self, pcm, n, hop, reassign=True, wire=None):
        """Host buffers in, the palette-index columns out as ONE lossless wire image per stream (emspec_batch_packed):
        returns (wire uint8 array, offsets int64 [S+1]); stream s is wire[offsets[s]:offsets[s+1]], expand it with
        wire_unpack_host(image, columns, rows).  `wire`: optional preallocated (pinned) uint8 array."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        if pcm.ndim == 1:
            pcm = pcm[None]
        S, L = pcm.shape
        Cn = num_columns(L, n, hop)
        if wire is None:
            wire = np.empty(S * wire_bound(Cn, self.rows), np.uint8)
        offsets = np.zeros(S + 1, np.int64)
        self._chk(self._lib.emspec_batch_packed(self._h, _np_ptr(pcm), S, L, n, hop, int(bool(reassign)), _np_ptr(wire),
                                                C.c_int64(wire.size), offsets.ctypes.data_as(C.c_void_p)))
        return wire, offsets

    # -- batch, device-resident torch tensors ---------------------------------
    def batch_device(self, pcm_t, n, hop, reassign=True, db=None, rgba=None, index=None, stream=None):
        """pcm_t: contiguous float32 CUDA tensor [S, L]; outputs: preallocated CUDA tensors or None.
        Enqueues on `stream` (a torch.cuda.Stream; default: the current torch stream); does not synchronise."""
        import torch
        assert pcm_t.is_cuda and pcm_t.dtype == torch.float32 and pcm_t.is_contiguous() and pcm_t.dim() == 2
        S, L = pcm_t.shape
        st = stream if stream is not None else torch.cuda.current_stream(pcm_t.device)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for t in (db, rgba, index):
            assert t is None or (t.is_cuda and t.is_contiguous())
        self._chk(self._lib.emspec_batch_device(self._h, ptr(pcm_t), S, L, n, hop, int(bool(reassign)), ptr(db),
                                                ptr(rgba), ptr(index), C.c_void_p(st.cuda_stream)))

    # -- multi-GPU: RCCL communicator + gather of finished palette-index columns ----
    def comm_init(self, comm_id, rank, world):
        """Collective over all `world` ranks (one engine / process per GPU); comm_id from comm_unique_id() of rank 0."""
        assert len(comm_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._chk(self._lib.emspec_comm_init(self._h, buf, rank, world))

    def comm_destroy(self):
        self._chk(self._lib.emspec_comm_destroy(self._h))

    def comm_set_timeout(self, seconds):
        """Bound on the host wait inside gather_columns (the size exchange); 0 = none.  On expiry the communicator is aborted."""
        self._chk(self._lib.emspec_comm_set_timeout(self._h, float(seconds)))

    @property
    def comm_rank(self):
        return int(self._lib.emspec_comm_rank(self._h))

    @property
    def comm_world(self):
        return int(self._lib.emspec_comm_world(self._h))

    def gather_columns(self, index_t, root=0, out=None, stream=None, loopback=False, packed=False):
        """packed=True: the root keeps the ranks' wire images packed in `out` (directory + images; expand on demand with
        wire_unpack at gather_packed_layout(rank)); capacity 256*(world+1) + sum of wire_bound over the ranks.
        index_t: contiguous uint8 CUDA tensor [..., rows] of this rank's finished columns; out (root only):
        uint8 CUDA tensor holding the ranks' blocks in rank order ([world, ...same shape...] when the shards are equal;
        the library checks that the announced shards fit).  Returns the bytes this rank put on the wire."""
        import torch
        assert index_t.is_cuda and index_t.dtype == torch.uint8 and index_t.is_contiguous() and index_t.shape[-1] == self.rows
        columns = index_t.numel() // self.rows
        st = stream if stream is not None else torch.cuda.current_stream(index_t.device)
        if out is not None:
            assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous()
        sent = C.c_int64(0)
        self._chk(self._lib.emspec_gather_columns(self._h, C.c_void_p(index_t.data_ptr()), columns, root,
                                                  C.c_void_p(out.data_ptr()) if out is not None else None,
                                                  out.numel() if out is not None else 0,
                                                  (GATHER_LOOPBACK if loopback else 0) | (GATHER_PACKED if packed else 0),
                                                  C.c_void_p(st.cuda_stream), C.byref(sent)))
        return int(sent.value)

    def gather_packed_layout(self, rank):
        """Root, after gather_columns(packed=True): (offset in the gathered buffer, image bytes, columns) of `rank`'s image."""
        off, nb, cols = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._lib.emspec_gather_packed_layout(self._h, rank, C.byref(off), C.byref(nb), C.byref(cols)))
        return int(off.value), int(nb.value), int(cols.value)

    def batch_gather(self, pcm, n, hop, reassign=True, root=0, want_db=False):
        """Host buffers: this rank's streams -> (gathered index [world,S,C,rows] on root else None, own dB or None, wire bytes)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        S, L = pcm.shape
        Cn = num_columns(L, n, hop)
        allidx = np.empty((self.comm_world, S, Cn, self.rows), np.uint8) if self.comm_rank == root else None
        db = np.empty((S, Cn, self.rows), np.float32) if want_db else None
        sent = C.c_int64(0)
        self._chk(self._lib.emspec_batch_gather(self._h, _np_ptr(pcm), S, L, n, hop, int(bool(reassign)), root, _np_ptr(allidx),
                                                _np_ptr(db), C.byref(sent)))
        return allidx, db, int(sent.value)

    def wire_pack(self, index_t, wire_t, stream=None, want_size=True):
        """index_t uint8 CUDA [columns, rows] -> wire_t (uint8 CUDA, >= wire_bound bytes); returns the image size."""
        import torch
        columns = index_t.numel() // self.rows
        assert wire_t.numel() >= wire_bound(columns, self.rows)
        st = stream if stream is not None else torch.cuda.current_stream(index_t.device)
        n = C.c_int64(-1)
        self._chk(self._lib.emspec_wire_pack(self._h, C.c_void_p(index_t.data_ptr()), columns, C.c_void_p(wire_t.data_ptr()),
                                             C.byref(n) if want_size else None, C.c_void_p(st.cuda_stream)))
        return int(n.value)

    def wire_unpack(self, wire_t, wire_bytes, out_t, stream=None):
        import torch
        columns = out_t.numel() // self.rows
        st = stream if stream is not None else torch.cuda.current_stream(out_t.device)
        self._chk(self._lib.emspec_wire_unpack(self._h, C.c_void_p(wire_t.data_ptr()), wire_bytes, columns,
                                               C.c_void_p(out_t.data_ptr()), C.c_void_p(st.cuda_stream)))

    # -- parity dump ------------------------------------------------------------
    def parity_dump(self, pcm, n, hop, reassign=True, frame0=0, nframes=None):
        pcm = np.ascontiguousarray(pcm, np.float32)
        if pcm.ndim == 1:
            pcm = pcm[None]
        S, L = pcm.shape
        if nframes is None:
            nframes = num_columns(L, n, hop) - frame0
        K = n // 2 + 1
        pw = np.empty((S, nframes, K), np.float32)
        col = np.empty((S, nframes, K), np.int32)
        row = np.empty((S, nframes, K), np.int32)
        self._chk(self._lib.emspec_parity_dump(self._h, _np_ptr(pcm), S, L, n, hop, int(bool(reassign)), frame0,
                                               nframes, _np_ptr(pw), _np_ptr(col), _np_ptr(row)))
        return pw, col, row

    def parity_dump_exact(self, pcm, n, hop, reassign=True, frame0=0, nframes=None):
        """EXACT-mode engine: (power float64, col, row, q int64), each [S][nframes][n/2+1]."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        if pcm.ndim == 1:
            pcm = pcm[None]
        S, L = pcm.shape
        if nframes is None:
            nframes = num_columns(L, n, hop) - frame0
        K = n // 2 + 1
        pw = np.empty((S, nframes, K), np.float64)
        col = np.empty((S, nframes, K), np.int32)
        row = np.empty((S, nframes, K), np.int32)
        q = np.empty((S, nframes, K), np.int64)
        self._chk(self._lib.emspec_parity_dump_exact(self._h, _np_ptr(pcm), S, L, n, hop, int(bool(reassign)), frame0,
                                                     nframes, _np_ptr(pw), _np_ptr(col), _np_ptr(row), _np_ptr(q)))
        return pw, col, row, q

    # -- streaming: the renderer's computeSpectrogramColumn -------------------------
    def column(self, frame, hop, reassign=True, want_rgba=False):
        frame = np.ascontiguousarray(frame, np.float32)
        n = frame.size
        db = np.empty(self.rows, np.float32)
        rgba = np.empty((self.rows, 4), np.uint8) if want_rgba else None
        c = C.c_int64(-2)
        self._chk(self._lib.emspec_column(self._h, _np_ptr(frame), n, hop, int(bool(reassign)), _np_ptr(db),
                                          _np_ptr(rgba), self.rows, C.byref(c)))
        return (db, rgba, int(c.value)) if want_rgba else (db, int(c.value))

    def push_samples(self, samples, n, hop, reassign=True, want_rgba=False):
        """Feed a block of samples of any length; returns (db [k][rows], first_column) (+ rgba) for the k columns
        the block completed (k may be 0)."""
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        k = int(self._lib.emspec_push_columns(self._h, samples.size, n, hop, int(bool(reassign))))
        if k < 0:
            raise EmspecError(ERR_INVALID_ARG, "invalid fft size / hop")
        db = np.empty((k, self.rows), np.float32)
        rgba = np.empty((k, self.rows, 4), np.uint8) if want_rgba else None
        cnt, first = C.c_int64(-1), C.c_int64(-2)
        self._chk(self._lib.emspec_push_samples(self._h, _np_ptr(samples), samples.size, n, hop, int(bool(reassign)),
                                                _np_ptr(db), _np_ptr(rgba), self.rows, k, C.byref(cnt), C.byref(first)))
        assert cnt.value == k
        return (db, rgba, int(first.value)) if want_rgba else (db, int(first.value))

    def flush(self, want_rgba=False):
        db = np.empty(self.rows, np.float32)
        rgba = np.empty((self.rows, 4), np.uint8) if want_rgba else None
        c = C.c_int64(-2)
        self._chk(self._lib.emspec_column_flush(self._h, _np_ptr(db), _np_ptr(rgba), self.rows, C.byref(c)))
        return (db, rgba, int(c.value)) if want_rgba else (db, int(c.value))

    def reset(self):
        self._chk(self._lib.emspec_reset(self._h))

    # -- live multi-stream streaming: S streams per call, one launch ------------------
    @property
    def live_streams(self):
        return int(self._lib.emspec_live_streams(self._h))

    def _live_out(self, S, cols, want_db, want_rgba, db, rgba):
        shape = (S, self.rows) if cols is None else (S, cols, self.rows)
        if db is None and want_db:
            db = np.empty(shape, np.float32)
        if rgba is None and want_rgba:
            rgba = np.empty(shape + (4,), np.uint8)
        assert db is None or (db.dtype == np.float32 and db.flags.c_contiguous and db.shape == shape)
        assert rgba is None or (rgba.dtype == np.uint8 and rgba.flags.c_contiguous and rgba.shape == shape + (4,))
        return db, rgba

    def columns(self, frames, hop, reassign=True, want_db=True, want_rgba=False, db=None, rgba=None):
        """Per-frame form (emspec_columns): frames [S][n], one frame of every stream -> (db [S][rows] or None,
        rgba [S][rows][4] or None, columns int64 [S] (-1 = the empty column)).  db / rgba: optional preallocated
        (page-locked: PinnedArray.array) outputs."""
        frames = frames if isinstance(frames, np.ndarray) and frames.dtype == np.float32 and frames.flags.c_contiguous \
            else np.ascontiguousarray(frames, np.float32)
        S, n = frames.shape
        db, rgba = self._live_out(S, None, want_db, want_rgba, db, rgba)
        cols = np.empty(S, np.int64)
        self._chk(self._lib.emspec_columns(self._h, _np_ptr(frames), S, n, hop, int(bool(reassign)), _np_ptr(db), _np_ptr(rgba),
                                           self.rows, _np_ptr(cols)))
        return db, rgba, cols

    def columns_flush(self, want_db=True, want_rgba=False, db=None, rgba=None):
        S = self.live_streams
        db, rgba = self._live_out(S, None, want_db, want_rgba, db, rgba)
        cols = np.empty(S, np.int64)
        self._chk(self._lib.emspec_columns_flush(self._h, _np_ptr(db), _np_ptr(rgba), self.rows, _np_ptr(cols)))
        return db, rgba, cols

    def push_columns_multi(self, count, n, hop, reassign=True):
        return int(self._lib.emspec_push_columns_multi(self._h, count, n, hop, int(bool(reassign))))

    def push_samples_multi(self, samples, n, hop, reassign=True, want_db=True, want_rgba=False, db=None, rgba=None,
                           count=None, offset=0):
        """Per-sample-block form (emspec_push_samples_multi): samples [S][>= offset + count] (a window [offset, offset + count)
        of every row is fed) -> (db [S][k][rows], rgba, counts int64 [S], first_columns int64 [S]); k = the largest per-stream
        count (or the second dimension of the preallocated db / rgba)."""
        samples = samples if isinstance(samples, np.ndarray) and samples.dtype == np.float32 and samples.flags.c_contiguous \
            else np.ascontiguousarray(samples, np.float32)
        S, width = samples.shape
        if count is None:
            count = width - offset
        assert 0 <= offset and offset + count <= width
        k = self.push_columns_multi(count, n, hop, reassign)
        if k < 0:
            raise EmspecError(ERR_INVALID_ARG, "invalid fft size / hop")
        if db is not None:
            k = db.shape[1]
        elif rgba is not None:
            k = rgba.shape[1]
        db, rgba = self._live_out(S, k, want_db, want_rgba, db, rgba)
        counts, firsts = np.empty(S, np.int64), np.empty(S, np.int64)
        base = C.c_void_p(samples.ctypes.data + 4 * offset)
        self._chk(self._lib.emspec_push_samples_multi(self._h, base, S, count, width, n, hop, int(bool(reassign)), _np_ptr(db),
                                                      _np_ptr(rgba), self.rows, k, _np_ptr(counts), _np_ptr(firsts)))
        return db, rgba, counts, firsts

    def reset_stream(self, stream):
        self._chk(self._lib.emspec_reset_stream(self._h, stream))
