"""Multi-block container for block-sharded compression.

The reference has no multi-block format (one block per archive, bce.cpp:1151-1157); this is an extension for
the sharded path only.  A plain single-block `.bce` archive is NOT wrapped, so it stays reference-compatible.

layout (little-endian):  b"BCEM" | u32 version=1 | u32 nblocks | nblocks x (u64 raw_bytes, u64 archive_bytes) | archives...
Every embedded archive is exactly what `bce -c` produces for that block alone (parity is per block).
"""
import struct

MAGIC = b"BCEM"


def pack_blocks(archives, raw_sizes):
    if len(archives) != len(raw_sizes):
        raise ValueError("one raw size per archive")
    out = [MAGIC, struct.pack("<II", 1, len(archives))]
    for a, r in zip(archives, raw_sizes):
        out.append(struct.pack("<QQ", r, len(a)))
    out.extend(archives)
    return b"".join(out)


def unpack_blocks(blob):
    if blob[:4] != MAGIC:
        raise ValueError("not a BCEM container")
    ver, nb = struct.unpack_from("<II", blob, 4)
    if ver != 1:
        raise ValueError("unknown container version %d" % ver)
    pos = 12
    meta = []
    for _ in range(nb):
        meta.append(struct.unpack_from("<QQ", blob, pos))
        pos += 16
    archives = []
    for _raw, alen in meta:
        archives.append(bytes(blob[pos:pos + alen]))
        pos += alen
    if pos != len(blob):
        raise ValueError("trailing bytes in container")
    return archives, [m[0] for m in meta]
