"""numpy restatement of the gather's wire image (em-spec_amd/csrc/pack.hip.inc) — TEST INFRASTRUCTURE ONLY.

The reference has no multi-GPU path, so the format is [BUILD-DEFINED]; this file pins the byte layout
the HIP pack / unpack kernels must produce:
    header 32 B : u32 magic 'EMW2', u32 rows, u64 columns, u64 payload bytes, 8 B zero
    offsets     : columns x u32, start of each column's non-zero indices in the payload (exclusive prefix sum)
    masks       : columns x ceil(rows/32) u32 words, bit (r % 32) of word r // 32 set <=> index[r] != 0
    payload     : the non-zero indices, column after column, rows ascending; zero-padded to a multiple of 16 B
"""
import numpy as np

MAGIC = 0x32574D45


def mask_words(rows):
    return (rows + 31) // 32


def fixed_bytes(columns, rows):
    return 32 + columns * 4 + columns * mask_words(rows) * 4


def bound(columns, rows):
    return fixed_bytes(columns, rows) + columns * rows + 32


def pack(index):
    """index: uint8 [columns][rows] -> uint8 wire image (exact size)."""
    index = np.ascontiguousarray(index, np.uint8)
    columns, rows = index.shape
    mw = mask_words(rows)
    nz = index != 0
    bits = np.zeros((columns, mw * 32), bool)
    bits[:, :rows] = nz
    words = (bits.reshape(columns, mw, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)
    payload = index[nz]                      # row-major order = column after column, rows ascending
    pad = (-payload.size) % 16
    hdr = np.zeros(8, np.uint32)
    hdr[0], hdr[1] = MAGIC, rows
    hdr[2], hdr[3] = columns & 0xFFFFFFFF, columns >> 32
    hdr[4], hdr[5] = payload.size & 0xFFFFFFFF, payload.size >> 32
    counts = nz.sum(axis=1).astype(np.uint64)
    offsets = (np.cumsum(counts) - counts).astype(np.uint32)
    return np.concatenate([hdr.view(np.uint8), offsets.view(np.uint8), words.reshape(-1).view(np.uint8), payload,
                           np.zeros(pad, np.uint8)])


def unpack(wire, columns, rows):
    wire = np.ascontiguousarray(wire, np.uint8)
    hdr = wire[:32].view(np.uint32)
    assert hdr[0] == MAGIC and hdr[1] == rows and (int(hdr[2]) | int(hdr[3]) << 32) == columns
    npay = int(hdr[4]) | int(hdr[5]) << 32
    mw = mask_words(rows)
    offsets = wire[32:32 + columns * 4].view(np.uint32)
    m0 = 32 + columns * 4
    words = wire[m0:m0 + columns * mw * 4].view(np.uint32).reshape(columns, mw)
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(columns, mw * 32)[:, :rows]
    counts = bits.sum(axis=1)
    assert int(counts.sum()) == npay and np.array_equal(offsets, np.cumsum(counts) - counts)
    p0 = fixed_bytes(columns, rows)
    out = np.zeros((columns, rows), np.uint8)
    out[bits] = wire[p0:p0 + npay]
    return out
