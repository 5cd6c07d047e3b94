#!/usr/bin/env python3
"""Differential fuzz of the library's own inflate (csrc/epg_inflate.h) against zlib, in memory (epgio_inflate_mem):

    python tools/fuzz_inflate.py --streams 100000 [--seed 1]

Seeds: a table text compressed at several zlib levels / strategies, stored blocks, the library's own compressor, two-member
files.  Mutations: bit flips near the block headers and anywhere, truncation, random splices, spliced tails of other seeds,
duplicated and dropped ranges.  Rule (the product's: an accepted stream is used, a declined one goes to zlib): whenever the own
inflate ACCEPTS a stream, zlib must accept it too and give the same bytes.  Run under AddressSanitizer by tools/asan_io.sh."""
import argparse
import sys
import time
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from epilogos_amd import _io  # noqa: E402


def table_text(rng, R, N, S=18):
    rows = []
    x = rng.integers(1, S + 1, size=(R, N))
    for r in range(R):
        rows.append("chr1\t%d\t%d\t" % (200 * r, 200 * r + 200) + "\t".join(str(v) for v in x[r]))
    return ("\n".join(rows) + "\n").encode()


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    raw = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + raw + (zlib.crc32(data) & 0xffffffff).to_bytes(4, "little")
            + (len(data) & 0xffffffff).to_bytes(4, "little"))


BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf(data, level=6, block=60000, eof=True):
    """The blocked gzip of htslib: members of <= 64 KiB with their compressed size in a 'BC' extra subfield."""
    out = b""
    for o in range(0, len(data), block):
        ch = data[o:o + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(ch) + c.flush()
        tot = 18 + len(raw) + 8
        out += (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, (tot - 1) & 255, (tot - 1) >> 8]) + raw
                + (zlib.crc32(ch) & 0xffffffff).to_bytes(4, "little") + len(ch).to_bytes(4, "little"))
    return out + (BGZF_EOF if eof else b"")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    texts = [table_text(rng, 60, 12), table_text(rng, 9, 40), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), b"A" * 5000 + b"\n", b""]
    seeds = []
    for t in texts:
        half = len(t) // 2
        seeds += [member(t, 6), member(t, 1), member(t, 9), member(t, 6, zlib.Z_FIXED), member(t, 6, zlib.Z_RLE),
                  member(t, 6, zlib.Z_HUFFMAN_ONLY), member(t, 0), member(t, 9, memlevel=1), _io.gzip_fast(t),
                  member(t[:half]) + member(t[half:], 1), member(t[:half], 0) + _io.gzip_fast(t[half:])]
    for t in texts[:2]:                                # members with empty stored blocks (Z_SYNC_FLUSH) near their end
        for pad in (0, 3, 9):
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            raw = c.compress(t) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(b"x" * pad) + c.flush(zlib.Z_SYNC_FLUSH) + c.flush()
            w = t + b"x" * pad
            seeds.append(b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + raw + (zlib.crc32(w) & 0xffffffff).to_bytes(4, "little") + len(w).to_bytes(4, "little"))
    n_plain = len(seeds)
    for t in texts[:3]:                                # BGZF: read by the parallel block reader (own=2) as well
        seeds += [bgzf(t, 6, 700), bgzf(t, 1, 2000, eof=False), bgzf(t, 0, 512)]
    for k, s in enumerate(seeds[n_plain:]):
        assert _io.inflate_mem(s, own=2) == _io.inflate_mem(s, own=False) is not None, "BGZF seed %d" % k
    # a BGZF member WITHOUT a deflate stream (empty payload, CRC 0, ISIZE 0): declined by every reader, before and
    # after a good member -- it is not a seed (the unmutated seeds must be accepted), its neighbours are
    empty = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 25, 0]) + bytes(8)
    for blob in (empty + seeds[n_plain], seeds[n_plain][:-len(BGZF_EOF)] + empty + BGZF_EOF):
        assert _io.inflate_mem(blob, own=2) is None and _io.inflate_mem(blob, own=False) is None, "empty BGZF member accepted"
    for k, s in enumerate(seeds):                      # the unmutated seeds must be accepted by both, identically
        a, b = _io.inflate_mem(s, own=True), _io.inflate_mem(s, own=False)
        assert a is not None and a == b, "seed %d: own %s zlib %s" % (k, a is None, b is None)
    t0 = time.time()
    accepted = declined_both = own_declined_only = bgzf_accepted = 0
    for i in range(args.streams):
        b = bytearray(seeds[int(rng.integers(0, len(seeds)))])
        kind = int(rng.integers(0, 8))
        n = len(b)
        if kind == 0:
            b[int(rng.integers(10, min(n, 140)))] ^= 1 << int(rng.integers(0, 8))           # block header / code lengths
        elif kind == 1:
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            b = b[: int(rng.integers(1, n))]
        elif kind == 3:
            j = int(rng.integers(10, max(11, n - 8)))
            b[j:j + int(rng.integers(1, 9))] = bytes(rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8))
        elif kind == 4:
            other = seeds[int(rng.integers(0, len(seeds)))]
            b = b[: int(rng.integers(10, n))] + other[int(rng.integers(0, len(other))):]
        elif kind == 5:
            j, k = sorted(int(v) for v in rng.integers(10, n, 2))
            b = b[:j] + b[j:k] + b[j:]                                                       # a range duplicated
        elif kind == 6:
            j, k = sorted(int(v) for v in rng.integers(10, n, 2))
            b = b[:j] + b[k:]                                                                # a range dropped
        else:
            b = b + bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))   # trailing bytes
        if len(b) < 18:
            b = b + b"\0" * (18 - len(b))
        b = bytes(b)
        blocked = _io.inflate_mem(b, own=2, cap=1 << 20)       # the BGZF reader: whatever it accepts, zlib must read the same
        if blocked is not None:
            theirs = _io.inflate_mem(b, own=False, cap=1 << 20)
            if theirs is None or theirs != blocked:
                Path("fuzz_counterexample.gz").write_bytes(b)
                print("MISMATCH at stream %d (kind %d): the BGZF reader accepted %d bytes, zlib %s -> fuzz_counterexample.gz"
                      % (i, kind, len(blocked), "declined" if theirs is None else "%d other bytes" % len(theirs)))
                return 1
            bgzf_accepted += 1
        mine = _io.inflate_mem(b, own=True, cap=1 << 20)
        if mine is None:
            theirs = _io.inflate_mem(b, own=False, cap=1 << 20)
            declined_both += theirs is None
            own_declined_only += theirs is not None
            continue
        theirs = _io.inflate_mem(b, own=False, cap=1 << 20)
        if theirs is None or theirs != mine:
            Path("fuzz_counterexample.gz").write_bytes(b)
            print("MISMATCH at stream %d (kind %d): own accepted %d bytes, zlib %s -> fuzz_counterexample.gz"
                  % (i, kind, len(mine), "declined" if theirs is None else "%d other bytes" % len(theirs)))
            return 1
        accepted += 1
    print("fuzz_inflate: %d streams in %.1f s: %d accepted by both with equal bytes, %d declined by both, %d declined by the own "
          "inflate only (zlib reads those), %d accepted by the BGZF block reader; no mismatch" % (args.streams, time.time() - t0, accepted, declined_both, own_declined_only, bgzf_accepted))
    return 0


if __name__ == "__main__":
    sys.exit(main())
