"""ctypes binding of oracle/_build/libbce_oracle.so (built by oracle/Makefile with gcc)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbce_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (no GPU toolchain involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("bce_oracle.c", "scan_oracle.cpp", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        L.bce_oracle_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.bce_oracle_compress.restype = C.c_int
        L.bce_oracle_free.argtypes = [C.c_void_p]
        L.bce_oracle_stage_seconds.argtypes = [C.POINTER(C.c_double)]
        L.bce_oracle_bwt_stage.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, u32p]
        L.bce_oracle_bwt_stage.restype = C.c_int
        L.bce_oracle_encode_from_bwt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), u32p]
        L.bce_oracle_encode_from_bwt.restype = C.c_int
        L.bce_oracle_plane_bits.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.bce_oracle_plane_bits.restype = C.c_int
        for name in ("nodes", "syms", "ops"):
            f = getattr(L, "bce_oracle_trace_" + name)
            f.argtypes = [C.POINTER(u32p)]
            f.restype = C.c_size_t
        L.bce_oracle_trace_rounds.restype = C.c_uint32
        L.bce_oracle_scan.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]
        L.bce_oracle_scan.restype = C.c_int
        L.bce_oracle_divbwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.bce_oracle_divbwt.restype = C.c_int32
        L.bce_oracle_inverse_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.bce_oracle_inverse_bwt.restype = C.c_int
        L.bce_oracle_context_index.argtypes = [C.c_uint32] * 4
        L.bce_oracle_context_index.restype = C.c_uint32
        L.bce_oracle_synth_rand.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        L.bce_oracle_synth_text.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _cfg(config):
    if config is None:
        return None, None
    if len(config) != 288:
        raise ValueError("config must be 288 bytes (9 rows x 32)")
    a, p = _buf(config)
    return a, p


def set_threads(t: int):
    """Threads of the round loop: 1 = reference built without OpenMP, 8 = with (one per plane, bce.cpp:1250-1252)."""
    lib().bce_oracle_set_threads(int(t))


def compress(data, config=None) -> bytes:
    """Reference `bce -c` on an in-memory buffer -> archive bytes."""
    a, p = _buf(data)
    _c, cp = _cfg(config)
    out, n = C.c_void_p(), C.c_size_t()
    rc = lib().bce_oracle_compress(p, len(a), cp, C.byref(out), C.byref(n))
    if rc != 0:
        raise ValueError("oracle compress failed rc=%d" % rc)
    try:
        return C.string_at(out, n.value)
    finally:
        lib().bce_oracle_free(out)


def stage_seconds():
    """Stage times of the last compress() as the reference's -DM_TIME build prints them: rotate, bwt, rank (planes), encode."""
    t = (C.c_double * 4)()
    lib().bce_oracle_stage_seconds(t)
    return dict(zip(("rotate", "bwt", "rank", "encode"), t))


def bwt_stage(data):
    """-> (bwt bytes as np.uint8[n], offset) per File::rotate + File::bwt."""
    a, p = _buf(data)
    out = np.empty(len(a), dtype=np.uint8)
    off = C.c_uint32()
    rc = lib().bce_oracle_bwt_stage(p, len(a), out.ctypes.data_as(C.c_void_p), C.byref(off))
    if rc != 0:
        raise ValueError("oracle bwt failed")
    return out, off.value


def encode_from_bwt(bwt, offset, config=None):
    """-> (archive bytes, C[8]) from BWT bytes (skips rotate/bwt)."""
    a, p = _buf(bwt)
    _c, cp = _cfg(config)
    out, n = C.c_void_p(), C.c_size_t()
    Cz = (C.c_uint32 * 8)()
    rc = lib().bce_oracle_encode_from_bwt(p, len(a), offset, cp, C.byref(out), C.byref(n), Cz)
    if rc != 0:
        raise ValueError("oracle encode failed")
    try:
        return C.string_at(out, n.value), list(Cz)
    finally:
        lib().bce_oracle_free(out)


def plane_bits(bwt):
    """-> np.uint8[8, n] of the wavelet-matrix plane bits in the reference's order."""
    a, p = _buf(bwt)
    out = np.empty((8, len(a)), dtype=np.uint8)
    lib().bce_oracle_plane_bits(p, len(a), out.ctypes.data_as(C.c_void_p))
    return out


def trace_encode_from_bwt(bwt, offset, config=None):
    """Run encode_from_bwt with tracing.

    -> dict(archive, C, rounds, nodes[N,5]=(round,plane,s,x0,x1), syms[S,6]=(plane,s,k,c1,c2,cs),
            ops[O,4]=(coder,cum,freq,total))
    """
    L = lib()
    L.bce_oracle_trace_begin()
    try:
        arch, Cz = encode_from_bwt(bwt, offset, config)
        res = {"archive": arch, "C": Cz, "rounds": L.bce_oracle_trace_rounds()}
        for name, w in (("nodes", 5), ("syms", 6), ("ops", 4)):
            ptr = C.POINTER(C.c_uint32)()
            cnt = getattr(L, "bce_oracle_trace_" + name)(C.byref(ptr))
            if cnt:
                res[name] = np.ctypeslib.as_array(ptr, shape=(cnt, w)).copy()
            else:
                res[name] = np.zeros((0, w), dtype=np.uint32)
        return res
    finally:
        L.bce_oracle_trace_end()
        L.bce_oracle_trace_free()


def divbwt(data):
    """libdivsufsort's divbwt(T, U, NULL, n) as restated for the oracle -> (U bytes, primary index)."""
    a, p = _buf(data)
    out = np.empty(len(a), dtype=np.uint8)
    pidx = lib().bce_oracle_divbwt(p, out.ctypes.data_as(C.c_void_p), len(a))
    return out.tobytes(), pidx


def inverse_bwt(bwt, idx) -> bytes:
    """The inverse (inverse_bw_transform(T, U, NULL, n, idx)) by the textbook LF walk."""
    a, p = _buf(bwt)
    out = np.empty(len(a), dtype=np.uint8)
    rc = lib().bce_oracle_inverse_bwt(p, out.ctypes.data_as(C.c_void_p), len(a), idx)
    if rc != 0:
        raise ValueError("oracle inverse_bwt failed rc=%d" % rc)
    return out.tobytes()


def context_index(bits: int, c1: int, c2: int, cs: int) -> int:
    """get_context's slot number (bce.cpp:675) in the reference's uint32 arithmetic."""
    return lib().bce_oracle_context_index(bits, c1, c2, cs)


def scan(data):
    """Reference `bce -s` on an in-memory buffer -> (288-byte config, the nine "Result size" values in bytes)."""
    a, p = _buf(data)
    cfg = np.zeros(288, dtype=np.uint8)
    res = (C.c_double * 9)()
    rc = lib().bce_oracle_scan(p, len(a), cfg.ctypes.data_as(C.c_void_p), res)
    if rc != 0:
        raise ValueError("oracle scan failed")
    return cfg.tobytes(), list(res)


def synth_rand(seed: int, n: int) -> bytes:
    out = np.empty(n, dtype=np.uint8)
    lib().bce_oracle_synth_rand(seed, out.ctypes.data_as(C.c_void_p), n)
    return out.tobytes()


def synth_text(seed: int, n: int) -> bytes:
    out = np.empty(n, dtype=np.uint8)
    lib().bce_oracle_synth_text(seed, out.ctypes.data_as(C.c_void_p), n)
    return out.tobytes()
