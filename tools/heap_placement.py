"""Where glibc puts the 16 MB, 2 MB-aligned posix_memalign blocks that round 4's decoder registered with the runtime
(DESIGN.md 4.5).  A request above the mmap threshold (128 KB at start) gets a mapping of its own -- but free() of any
mmapped block up to 32 MB RAISES the threshold to that block's size (glibc's dynamic threshold), and a host program that
has freed one block between 18 and 32 MB -- numpy arrays and bytes objects of a test suite -- gets the same request carved
out of the brk heap from then on: next to everything else malloc hands out, trimmed and reused as malloc sees fit.
No GPU needed: python tools/heap_placement.py"""
import ctypes

import numpy as np

libc = ctypes.CDLL("libc.so.6")
libc.sbrk.restype = ctypes.c_void_p
libc.free.argtypes = [ctypes.c_void_p]
libc.posix_memalign.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t]


def where():
    p = ctypes.c_void_p()
    assert libc.posix_memalign(ctypes.byref(p), 2 << 20, 16 << 20) == 0
    brk = libc.sbrk(0)
    s = "%#x: %s (program break %#x)" % (p.value, "INSIDE THE BRK HEAP" if p.value < brk else "a mapping of its own", brk)
    libc.free(p)
    return s


print("fresh process:            ", where())
a = np.ones(24 << 20, dtype=np.uint8)
del a
print("after one freed 24 MB array:", where())
