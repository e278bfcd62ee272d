"""ctypes binding of include/bce_hip.h, shaped like the reference's own objects.

reference                                   here
------------------------------------------  ---------------------------------------------
RankFile(name)  (bce.cpp:932-984)           RankFile(data)      rotate+BWT (K1), planes (K2)
  .size() .offset() .status() .ranks          .size() .offset() .status() .zeros / .rank1()
BCE<AdaptiveCoder<31>,...>::encode(file)    BCE(config).encode(rank_file) -> archive bytes
  (bce.cpp:1117-1167)
AdaptiveCoder::load_config (bce.cpp:626)    BCE(config=288 bytes) / BCE.load_config(path)
main -c (bce.cpp:1403-1427)                 compress(data, config) ; bce_amd/bin/bce
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BCE_HIP_LIB") or os.path.join(_HERE, "lib", "libbcehip.so")   # (BCE_HIP_LIB: kernel-variant experiments)
_lib = None

CONFIG_BYTES = 288


class BceError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: status %d (%s)%s" % (where, status, _strerror(status), (" -- " + detail) if detail else ""))


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("nodes", C.c_uint64), ("symbols", C.c_uint64), ("rounds", C.c_uint32),
                ("sort_rounds", C.c_uint32), ("flushes", C.c_uint32), ("spine_levels", C.c_uint32),
                ("t_load", C.c_double), ("t_bwt", C.c_double), ("t_planes", C.c_double), ("t_enum", C.c_double),
                ("t_model", C.c_double), ("t_coder", C.c_double), ("t_total", C.c_double),
                ("k3_ms", C.c_double), ("k3_launches", C.c_double), ("t_coder_busy", C.c_double),
                ("list_grows", C.c_double), ("list_nodes", C.c_double), ("split_rounds", C.c_double),
                ("reg_maps", C.c_double), ("reg_unmaps", C.c_double), ("dec_restarts", C.c_double), ("t_model_kernels", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/bce_hip.h declares: (name, restype, argtypes)
_u8p, _u32p, _vp = C.c_void_p, C.c_void_p, C.c_void_p
SYMBOLS = [
    ("bce_hip_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    ("bce_hip_create_sized", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint64]),
    ("bce_hip_destroy", None, [C.c_void_p]),
    ("bce_hip_strerror", C.c_char_p, [C.c_int]),
    ("bce_hip_last_error", C.c_char_p, [C.c_void_p]),
    ("bce_hip_set_config", C.c_int, [C.c_void_p, _u8p]),
    ("bce_hip_set_symbol_capacity", C.c_int, [C.c_void_p, C.c_uint64]),
    ("bce_hip_set_gated", C.c_int, [C.c_void_p, C.c_int]),
    ("bce_hip_set_progress", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bce_hip_debug_set", C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    ("bce_hip_load_host", C.c_int, [C.c_void_p, _u8p, C.c_uint32]),
    ("bce_hip_load_device", C.c_int, [C.c_void_p, _vp, C.c_uint32]),
    ("bce_hip_bwt", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("bce_hip_set_bwt", C.c_int, [C.c_void_p, _u8p, C.c_uint32, C.c_uint32]),
    ("bce_hip_get_bwt", C.c_int, [C.c_void_p, _u8p]),
    ("bce_hip_divbwt", C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("bce_hip_inverse_bwt", C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint32, C.c_uint32]),
    ("bce_hip_build_planes", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("bce_hip_get_plane_bits", C.c_int, [C.c_void_p, C.c_int, _u8p]),
    ("bce_hip_rank1", C.c_int, [C.c_void_p, C.c_int, _u32p, C.c_uint32, _u32p]),
    ("bce_hip_encode", C.c_int, [C.c_void_p]),
    ("bce_hip_archive_size", C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    ("bce_hip_archive_copy", C.c_int, [C.c_void_p, _u8p, C.c_size_t]),
    ("bce_hip_set_plane_mask", C.c_int, [C.c_void_p, C.c_uint32]),
    ("bce_hip_plane_stream_size", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    ("bce_hip_plane_stream_copy", C.c_int, [C.c_void_p, C.c_int, _vp, C.c_size_t]),
    ("bce_hip_plane_stream_set", C.c_int, [C.c_void_p, C.c_int, _vp, C.c_size_t]),
    ("bce_hip_compress", C.c_int, [C.c_void_p, _u8p, C.c_uint32, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("bce_hip_compress_device", C.c_int, [C.c_void_p, _vp, C.c_uint32, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("bce_hip_enum_begin", C.c_int, [C.c_void_p]),
    ("bce_hip_enum_nodes", C.c_int, [C.c_void_p, C.c_int, _u32p, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("bce_hip_enum_round", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("bce_hip_enum_symbols", C.c_int, [C.c_void_p, _u32p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("bce_hip_enum_model", C.c_int, [C.c_void_p, _u32p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("bce_hip_scan", C.c_int, [C.c_void_p, _u8p, C.POINTER(C.c_double)]),
    ("bce_hip_decompress", C.c_int, [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("bce_hip_decompress_device", C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("bce_hip_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("bce_hip_synth_text", None, [C.c_uint64, _u8p, C.c_size_t]),
    ("bce_hip_synth_rand", None, [C.c_uint64, _u8p, C.c_size_t]),
]


def library_path():
    return _LIB_PATH


def load_library():
    """Load libbcehip.so (no fallback: raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C bce_amd/csrc) first; there is no CPU fallback" % _LIB_PATH)
        lib = C.CDLL(_LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _strerror(status):
    try:
        return load_library().bce_hip_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "?"


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class _Ctx:
    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.bce_hip_create(C.byref(h), device)
        if rc != 0:
            raise BceError(rc, "bce_hip_create", "no usable HIP device %d (a GPU is required; there is no CPU path)" % device)
        self.h = h

    def check(self, rc, where):
        if rc != 0:
            raise BceError(rc, where, self.lib.bce_hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.bce_hip_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class RankFile:
    """The reference's RankFile (bce.cpp:932-984) on the GPU: input -> rotate + BWT (K1) -> 8 planes (K2).

    `data` may be bytes / a numpy u8 array (host) or an int device pointer with `n` (input resident in HBM).
    """

    def __init__(self, data=None, n=None, device=0, device_ptr=None, bwt=None, offset=None, ctx=None, build=True):
        self._c = ctx or _Ctx(device)
        lib = self._c.lib
        self._status = 0
        if bwt is not None:                      # test hook: inject a BWT, skip K1
            a = _as_u8(bwt)
            self._c.check(lib.bce_hip_set_bwt(self._c.h, a.ctypes.data, len(a), int(offset)), "bce_hip_set_bwt")
            self._n, self._offset = len(a), int(offset)
        else:
            if device_ptr is not None:
                self._c.check(lib.bce_hip_load_device(self._c.h, int(device_ptr), int(n)), "bce_hip_load_device")
                self._n = int(n)
            else:
                a = _as_u8(data)
                if len(a) == 0:
                    self._status = 1               # reference: `Error loading file` path; (empty input crashes it, Q12)
                    raise BceError(-1, "RankFile", "empty input")
                self._c.check(lib.bce_hip_load_host(self._c.h, a.ctypes.data, len(a)), "bce_hip_load_host")
                self._n = len(a)
            off = C.c_uint32()
            self._c.check(lib.bce_hip_bwt(self._c.h, C.byref(off)), "bce_hip_bwt")
            self._offset = off.value
        self.zeros = None
        if build:
            z = (C.c_uint32 * 8)()
            self._c.check(lib.bce_hip_build_planes(self._c.h, z), "bce_hip_build_planes")
            self.zeros = list(z)

    def size(self):
        return self._n

    def offset(self):
        return self._offset

    def status(self):
        return self._status

    def bwt(self):
        out = np.empty(self._n, dtype=np.uint8)
        self._c.check(self._c.lib.bce_hip_get_bwt(self._c.h, out.ctypes.data), "bce_hip_get_bwt")
        return out

    def plane_bits(self, plane):
        out = np.empty(self._n, dtype=np.uint8)
        self._c.check(self._c.lib.bce_hip_get_plane_bits(self._c.h, plane, out.ctypes.data), "bce_hip_get_plane_bits")
        return out

    def rank1(self, plane, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.empty(len(idx), dtype=np.uint32)
        self._c.check(self._c.lib.bce_hip_rank1(self._c.h, plane, idx.ctypes.data, len(idx), out.ctypes.data), "bce_hip_rank1")
        return out

    def close(self):
        self._c.close()


class BCE:
    """The reference's BCE<AdaptiveCoder<31>, ...> (bce.cpp:1111-1374), encode side."""

    max = 31

    def __init__(self, config=None, symbol_capacity=0):
        if config is not None and len(config) != CONFIG_BYTES:
            raise ValueError("Config not found or wrong size.")   # bce.cpp:629-631
        self.config = None if config is None else bytes(config)
        self.symbol_capacity = symbol_capacity

    @staticmethod
    def load_config(path):
        with open(path, "rb") as f:
            return f.read()

    def _apply(self, rf):
        lib, h = rf._c.lib, rf._c.h
        cfg = None
        if self.config is not None:
            cfg = _as_u8(self.config)
        rf._c.check(lib.bce_hip_set_config(h, cfg.ctypes.data if cfg is not None else None), "bce_hip_set_config")
        rf._c.check(lib.bce_hip_set_symbol_capacity(h, self.symbol_capacity), "bce_hip_set_symbol_capacity")

    def encode(self, rf: RankFile, out=None):
        """-> the archive (a bytearray).  With `out` (a writable C-contiguous uint8 numpy array that is large enough) the
        library lays the archive out there -- the C ABI's own shape: the caller's buffer, e.g. pinned memory a collective
        sends from -- and a memoryview of its first len(archive) bytes comes back; too small an `out` is ignored."""
        self._apply(rf)
        lib, h = rf._c.lib, rf._c.h
        rf._c.check(lib.bce_hip_encode(h), "bce_hip_encode")
        n = C.c_size_t()
        rf._c.check(lib.bce_hip_archive_size(h, C.byref(n)), "bce_hip_archive_size")
        if out is not None and isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous and out.flags.writeable and out.size >= n.value:
            rf._c.check(lib.bce_hip_archive_copy(h, out.ctypes.data, out.size), "bce_hip_archive_copy")
            return memoryview(out.reshape(-1))[:n.value]
        # the library lays the archive out straight into this buffer (one copy of the coded streams, none in Python):
        # a bytearray compares, hashes, slices and writes like bytes
        out = bytearray(n.value)
        view = (C.c_uint8 * n.value).from_buffer(out)
        rf._c.check(lib.bce_hip_archive_copy(h, C.addressof(view), n.value), "bce_hip_archive_copy")
        del view
        return out

    # --- BCE::code one round at a time (parity tests) ---
    def code_begin(self, rf: RankFile):
        self._apply(rf)
        rf._c.check(rf._c.lib.bce_hip_enum_begin(rf._c.h), "bce_hip_enum_begin")

    def code_nodes(self, rf: RankFile, plane):
        cap = rf.size() // 2 + 2
        out = np.empty((cap, 3), dtype=np.uint32)
        cnt = C.c_uint32()
        rf._c.check(rf._c.lib.bce_hip_enum_nodes(rf._c.h, plane, out.ctypes.data, cap, C.byref(cnt)), "bce_hip_enum_nodes")
        return out[:cnt.value].copy()

    def code_round(self, rf: RankFile):
        nxt = C.c_uint64()
        rf._c.check(rf._c.lib.bce_hip_enum_round(rf._c.h, C.byref(nxt)), "bce_hip_enum_round")
        return nxt.value

    def code_symbols(self, rf: RankFile, cap):
        out = np.empty((max(cap, 1), 6), dtype=np.uint32)
        cnt = C.c_uint64()
        rf._c.check(rf._c.lib.bce_hip_enum_symbols(rf._c.h, out.ctypes.data, cap, C.byref(cnt)), "bce_hip_enum_symbols")
        return out[:cnt.value].copy()

    def code_model(self, rf: RankFile, cap):
        out = np.empty((max(cap, 1), 3), dtype=np.uint32)
        cnt = C.c_uint64()
        rf._c.check(rf._c.lib.bce_hip_enum_model(rf._c.h, out.ctypes.data, cap, C.byref(cnt)), "bce_hip_enum_model")
        return out[:cnt.value].copy()


def stats(rf: RankFile) -> dict:
    st = Stats()
    rf._c.check(rf._c.lib.bce_hip_get_stats(rf._c.h, C.byref(st)), "bce_hip_get_stats")
    return st.as_dict()


def stats_of(ctx) -> dict:
    """The statistics of the context's last compression (as stats(rf), without the RankFile)."""
    st = Stats()
    ctx.check(ctx.lib.bce_hip_get_stats(ctx.h, C.byref(st)), "bce_hip_get_stats")
    return st.as_dict()


def compress(data, config=None, device=0, ctx=None) -> bytes:
    """`bce -c` on an in-memory buffer (bce.cpp:1403-1427 minus file I/O) -> archive bytes."""
    rf = RankFile(data, device=device, ctx=ctx)
    try:
        return BCE(config).encode(rf)
    finally:
        if ctx is None:
            rf.close()


def compress_device(device_ptr, n, config=None, device=0, ctx=None, out=None):
    """Same with the input already resident in HBM.  Returns (archive bytes, stats dict); `out`: see BCE.encode."""
    own = ctx is None
    c = ctx or _Ctx(device)
    try:
        rf = RankFile(n=n, device_ptr=device_ptr, ctx=c)
        arch = BCE(config).encode(rf, out=out)
        return arch, stats(rf)
    finally:
        if own:
            c.close()


def set_plane_mask(ctx, mask):
    """The planes whose range coders context `ctx` runs from now on (bit p = plane p; 0xFF = all: the default)."""
    ctx.check(ctx.lib.bce_hip_set_plane_mask(ctx.h, int(mask) & 0xFF), "bce_hip_set_plane_mask")


def plane_stream(ctx, plane) -> np.ndarray:
    """The finished coded stream of one plane of the context's last encode (u16 words)."""
    n = C.c_size_t()
    ctx.check(ctx.lib.bce_hip_plane_stream_size(ctx.h, plane, C.byref(n)), "bce_hip_plane_stream_size")
    out = np.empty(n.value, dtype=np.uint16)
    ctx.check(ctx.lib.bce_hip_plane_stream_copy(ctx.h, plane, out.ctypes.data, n.value), "bce_hip_plane_stream_copy")
    return out


def set_plane_stream(ctx, plane, words):
    """Put another context's finished stream of `plane` (same input, same config) in the place of this context's own."""
    w = np.ascontiguousarray(words, dtype=np.uint16)
    ctx.check(ctx.lib.bce_hip_plane_stream_set(ctx.h, plane, w.ctypes.data, w.size), "bce_hip_plane_stream_set")


def archive_of(ctx) -> bytearray:
    """The archive of the context's last encode as it stands (after set_plane_stream: with the streams put in)."""
    n = C.c_size_t()
    ctx.check(ctx.lib.bce_hip_archive_size(ctx.h, C.byref(n)), "bce_hip_archive_size")
    out = bytearray(n.value)
    view = (C.c_uint8 * n.value).from_buffer(out)
    ctx.check(ctx.lib.bce_hip_archive_copy(ctx.h, C.addressof(view), n.value), "bce_hip_archive_copy")
    del view
    return out


class ContextPool:
    """`contexts` gated contexts on one device (bce_hip_set_gated), driven by one host thread each: their GPU phases take
    turns while the eight coder threads of the context that has just left the GPU finish its last batches, so the step
    time of a stream of inputs is the GPU phase instead of GPU phase + coding tail.  Buffers are kept between calls."""

    def __init__(self, contexts=2, device=0):
        self.ctxs = [_Ctx(device) for _ in range(max(1, int(contexts)))]

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def compress_many(self, inputs, config=None, on_device=False, with_stats=False, symbol_capacity=0):
        """Independent inputs (files, or the blocks of `bce -cN`) -> their archives, in order; each is the archive
        `compress` gives for the same input.  `inputs`: buffers, or (device_ptr, n) pairs with on_device=True."""
        import threading
        inputs = list(inputs)
        results = [None] * len(inputs)
        errors = []
        ctxs = self.ctxs[:max(1, min(len(self.ctxs), len(inputs)))]
        lock = threading.Lock()
        nxt = [0]

        def worker(c):
            try:
                c.check(c.lib.bce_hip_set_gated(c.h, 1), "bce_hip_set_gated")
                while True:
                    with lock:
                        if errors or nxt[0] >= len(inputs):
                            return
                        i = nxt[0]
                        nxt[0] += 1
                    if on_device:
                        ptr, n = inputs[i]
                        rf = RankFile(n=n, device_ptr=ptr, ctx=c)
                    else:
                        rf = RankFile(inputs[i], ctx=c)
                    arch = BCE(config, symbol_capacity).encode(rf)
                    results[i] = (arch, stats(rf)) if with_stats else arch
            except Exception as e:  # the other workers stop at their next input
                with lock:
                    errors.append(e)
            finally:
                c.lib.bce_hip_set_gated(c.h, 1)      # (gives the gate back if this context still holds it)

        threads = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results


def compress_many(inputs, config=None, device=0, contexts=2, on_device=False, with_stats=False, symbol_capacity=0):
    """ContextPool.compress_many with a pool of its own."""
    with ContextPool(contexts, device) as pool:
        return pool.compress_many(inputs, config=config, on_device=on_device, with_stats=with_stats, symbol_capacity=symbol_capacity)


def scan(data, device=0):
    """`bce -s` on an in-memory buffer (bce.cpp:1384-1402 minus file I/O) -> (288-byte config, nine result sizes)."""
    rf = RankFile(data, device=device)
    try:
        cfg = np.zeros(CONFIG_BYTES, dtype=np.uint8)
        res = (C.c_double * 9)()
        rf._c.check(rf._c.lib.bce_hip_scan(rf._c.h, cfg.ctypes.data, res), "bce_hip_scan")
        return cfg.tobytes(), list(res)
    finally:
        rf.close()


def decompress(archive) -> bytes:
    """`bce -d` on an in-memory archive (bce.cpp:1428-1472 minus file I/O).  Host C++ decoder; needs no GPU."""
    lib = load_library()
    a = _as_u8(archive)
    n = C.c_size_t()
    rc = lib.bce_hip_decompress(a.ctypes.data, len(a), None, 0, C.byref(n))
    if rc != 0:
        raise BceError(rc, "bce_hip_decompress")
    out = np.empty(n.value, dtype=np.uint8)
    rc = lib.bce_hip_decompress(a.ctypes.data, len(a), out.ctypes.data, n.value, C.byref(n))
    if rc != 0:
        raise BceError(rc, "bce_hip_decompress")
    return out.tobytes()


def decompress_device(archive, device=0, ctx=None, out=None):
    """`bce -d` with the GPU doing everything but the eight sequential range decoders (kd_decode.hip) -> bytes.

    With `out` (a writable C-contiguous uint8 array at least as large as the original) the bytes are written there, as a
    caller of the C ABI would have them, and the number of bytes is returned: no allocation and no copy on this side."""
    own = ctx is None
    c = ctx or _Ctx(device)
    try:
        a = _as_u8(archive)
        n = C.c_size_t()
        if out is not None:
            if out.dtype != np.uint8 or not out.flags.c_contiguous or not out.flags.writeable:
                raise ValueError("out must be a writable C-contiguous uint8 array")
            c.check(c.lib.bce_hip_decompress_device(c.h, a.ctypes.data, len(a), out.ctypes.data, out.size, C.byref(n)),
                    "bce_hip_decompress_device")
            return n.value
        c.check(c.lib.bce_hip_decompress_device(c.h, a.ctypes.data, len(a), None, 0, C.byref(n)), "bce_hip_decompress_device")
        buf = np.empty(n.value, dtype=np.uint8)
        c.check(c.lib.bce_hip_decompress_device(c.h, a.ctypes.data, len(a), buf.ctypes.data, n.value, C.byref(n)),
                "bce_hip_decompress_device")
        return buf.tobytes()
    finally:
        if own:
            c.close()


def synth_text(seed, n) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    load_library().bce_hip_synth_text(seed, out.ctypes.data, n)
    return out


def synth_rand(seed, n) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    load_library().bce_hip_synth_rand(seed, out.ctypes.data, n)
    return out
