"""CPU tests of the native input codec and score writer (libepilogos_io.so) against pandas / Python formatting and the
reference's golden text."""
import gzip
import struct

import numpy as np
import pandas as pd
import pytest

from epilogos_amd import _io, helpers
from tests.test_host_logic import write_tsv

S = 18


def _pandas_states(path, lo, hi):
    ncols = pd.read_table(path, nrows=1, header=None, sep="\t").shape[1]
    df = pd.read_table(path, usecols=range(3, ncols), skiprows=lo, nrows=hi - lo, header=None, sep="\t")
    return (df.to_numpy(dtype=int) - 1).astype(np.int8)


@pytest.mark.parametrize("suffix", [".txt", ".txt.gz"])
def test_parser_matches_pandas(tmp_path, suffix):
    rng = np.random.default_rng(5)
    x = rng.integers(0, S, size=(5003, 37)).astype(np.int8)
    f = tmp_path / ("matrix_chr7" + suffix)
    write_tsv(f, x, chrom="chr7", start0=1000)
    assert helpers.countRows(f) == 5003
    st, loc = _io.read_table(f)
    assert st.dtype == np.int8 and np.array_equal(st, x)
    assert np.array_equal(st, _pandas_states(f, 0, 5003))
    la = loc.to_object_array()
    ref = pd.read_table(f, header=None, sep="\t", usecols=[0, 1, 2]).to_numpy()
    assert la.shape == (5003, 3) and (la == ref).all()
    for lo, hi in ((0, 1), (17, 4100), (5002, 5003), (100, 100)):
        st, loc = _io.read_table(f, (lo, hi), threads=3)
        assert np.array_equal(st, x[lo:hi]) and len(loc) == hi - lo
    st, _ = _io.read_table(f, (4000, 99999), threads=1)        # hi clamps to the file
    assert np.array_equal(st, x[4000:])
    st, _ = _io.read_table(f, ldx=48)                            # padded rows: pad bytes are -1
    assert st.shape == (5003, 48) and np.array_equal(st[:, :37], x) and (st[:, 37:] == -1).all()


def test_parser_edge_cases(tmp_path):
    x = np.arange(12, dtype=np.int8).reshape(3, 4) % S
    write_tsv(tmp_path / "nonl.txt", x, trailing_newline=False)
    st, loc = _io.read_table(tmp_path / "nonl.txt")
    assert st.shape == (2, 4)                                    # quirk Q6: the unterminated last line does not exist
    (tmp_path / "bad.txt").write_text("chr1\t0\t200\t1\t2\nchr1\t200\t400\t1\n")
    with pytest.raises(_io.EpilogosIOError):
        _io.read_table(tmp_path / "bad.txt")
    (tmp_path / "alpha.txt").write_text("chr1\t0\t200\t1\tx\n")
    with pytest.raises(_io.EpilogosIOError):
        _io.read_table(tmp_path / "alpha.txt")
    with pytest.raises(_io.EpilogosIOError):
        _io.read_table(tmp_path / "missing.txt")
    (tmp_path / "zero.txt").write_text("chrX\t0\t200\t0\t19\n")     # 0 -> -1 and 19 -> 18: kept, rejected downstream
    st, _ = _io.read_table(tmp_path / "zero.txt")
    assert st.tolist() == [[-1, 18]]
    (tmp_path / "crlf.txt").write_text("chr1\t0\t200\t3\t4\r\nchr1\t200\t400\t5\t6\r\n")
    st, _ = _io.read_table(tmp_path / "crlf.txt")
    assert st.tolist() == [[2, 3], [4, 5]]
    # a row that is one column short but holds a junk byte inside a field must not pass as a full row (round-2 advisory:
    # the scalar fast path stopped on the junk byte and the general loop swallowed it as if it were the separator)
    for k, junk in enumerate(("1.5\t7", "1x5\t7", "1 5\t7", "1\t5.7", "15\t7x", "1\t\t7", "1\r5\t7")):
        f = tmp_path / ("junk%d.txt" % k)
        f.write_text("chr1\t0\t200\t1\t2\t3\nchr1\t200\t400\t%s\n" % junk)
        with pytest.raises(_io.EpilogosIOError):
            _io.read_table(f)
        with pytest.raises(_io.EpilogosIOError):
            _io.read_table(f, threads=1)
    # the same with more columns than one AVX-512 step covers, junk far from either end
    good = "\t".join(str(1 + i % 18) for i in range(200))
    for k, (a, b) in enumerate((("\t7\t8\t", "\t7x8\t"), ("\t17\t18\t", "\t17.18\t"), ("\t3\t4\t", "\t3 4\t"))):
        bad = good.replace(a, b, 1)
        assert bad != good
        f = tmp_path / ("wide%d.txt" % k)
        f.write_text("chr1\t0\t200\t%s\nchr1\t200\t400\t%s\n" % (good, bad))
        with pytest.raises(_io.EpilogosIOError):
            _io.read_table(f)


def test_writer_reproduces_reference_bytes(tmp_path, golden_real, golden_edge):
    g = golden_real
    R = g["x"].shape[0]
    f = tmp_path / "in.txt"
    write_tsv(f, g["x"], start0=int(g["start0"]))
    _, loc = _io.read_table(f)
    out = tmp_path / "scores.txt.gz"
    _io.write_scores(out, loc, g["s1_f32"], threads=3)
    with gzip.open(out, "rb") as fh:
        assert fh.read() == g["s1_text"].tobytes()              # the reference's writeScores output, byte for byte
    # object-array locations (the reference's locationArr) and the '-0.00000' case
    from epilogos_amd.scores import writeScores
    writeScores(golden_edge["fmt_vals"], tmp_path / "fmt.txt.gz", np.array([["chrX", 200, 400]], dtype=object))
    with gzip.open(tmp_path / "fmt.txt.gz", "rb") as fh:
        assert fh.read() == golden_edge["fmt_text"].tobytes()
    _io.write_scores(tmp_path / "empty.txt.gz", _io.Locations(np.zeros(0, np.uint8), np.zeros(1, np.int64)),
                     np.zeros((0, S), np.float32))
    with gzip.open(tmp_path / "empty.txt.gz", "rb") as fh:
        assert fh.read() == b""


def test_many_gzip_members(tmp_path):
    R = 70000                                                    # > 2 members of 32768 rows
    rng = np.random.default_rng(1)
    sc = rng.normal(size=(R, 3)).astype(np.float32)
    loc = _io.Locations.from_object_array([["chr2", 200 * i, 200 * i + 200] for i in range(R)])
    _io.write_scores(tmp_path / "m.txt.gz", loc, sc, threads=4)
    with gzip.open(tmp_path / "m.txt.gz", "rt") as fh:
        lines = fh.read().split("\n")
    assert len(lines) == R + 1 and lines[-1] == ""
    assert lines[12345] == "chr2\t%d\t%d\t" % (200 * 12345, 200 * 12345 + 200) + "\t".join("%.5f" % float(v) for v in sc[12345])


def test_format_f5_is_pythons_percent_5f():
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=400000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    special = np.array([0.0, -0.0, 1e-7, -1e-7, 5e-6, -5e-6, 4.9999998e-6, 5.0000002e-6, 1.5e-5, 2.5e-5, 0.5, 1.0, -1.0, 11.999996,
                        2.675, 3.0769148, 123456.789, 1e10, 3.4e38, -3.4e38, 1e-45, np.inf, -np.inf, np.nan, 0.000015, 0.000025,
                        0.125, 0.3125e-4, 8388608.5, 16777216.0], dtype=np.float32)
    scores_like = (rng.random(200000) * 12 - 0.5).astype(np.float32)
    for arr in (special, vals[:200000], scores_like):
        got = _io.format_f5(arr, sep="\n").decode().split("\n")[:-1]
        want = ["%.5f" % float(v) for v in arr]
        bad = [(float(a), g, w) for a, g, w in zip(arr, got, want) if g != w]
        assert not bad, bad[:5]


def test_write_metrics_matches_python_formatting(tmp_path):
    """pairwiseMetrics lines of the native writer == the reference's str.format template
    (roiAndVisualPairwise.py:560-569), including -0.0, huge / tiny p-values and more rows than one gzip member."""
    rng = np.random.default_rng(11)
    R = 70001
    chroms = ["chr1", "chr10", "chrX"]
    ci = np.sort(rng.integers(0, 3, R)).astype(np.int32)
    start = np.arange(R, dtype=np.int64) * 200
    end = start + 200
    names = ["TssA", "Enh G1", "Quies"]
    md = rng.integers(1, 4, R).astype(np.int32)
    dist = (rng.normal(0, 3, R) * rng.choice([1e-7, 1e-3, 1, 1e4], R)).astype(np.float32)
    dist[:4] = [0.0, -0.0, np.float32(2.5e-6), np.float32(-123456.789)]
    pv = rng.random(R) ** 20
    pv[:5] = [0.0, 1.0, 1e-300, 5e-324, 0.999995]
    mh = np.minimum(pv * 3, 1.0)
    sign = lambda x: "+" if x >= 0 else "-"
    for with_p in (False, True):
        out = tmp_path / ("m%d.txt.gz" % with_p)
        _io.write_metrics(out, chroms, ci, start, end, names, md, dist, pv if with_p else None, mh if with_p else None)
        got = gzip.open(out, "rb").read().decode().splitlines()
        assert len(got) == R
        for i in list(range(200)) + list(rng.integers(0, R, 300)) + [R - 1]:
            want = "{}\t{}\t{}\t{}\t{:.5f}\t{}".format(chroms[ci[i]], start[i], end[i], names[md[i] - 1], abs(dist[i]), sign(dist[i]))
            if with_p:
                want += "\t{:.5e}\t{:.5e}".format(pv[i], mh[i])
            assert got[i] == want
    _io.write_metrics(tmp_path / "empty.txt.gz", chroms, [], [], [], names, [], [])
    assert gzip.open(tmp_path / "empty.txt.gz", "rb").read() == b""
    with pytest.raises(ValueError):
        _io.write_metrics(tmp_path / "bad.txt.gz", chroms, [5], [0], [200], names, [1], [0.5])


def test_binary_input_cache(tmp_path, monkeypatch):
    """the parsed matrix kept as int8 .npy + coordinate side-car; keyed by path, size and mtime."""
    rng = np.random.default_rng(2)
    x = rng.integers(0, 18, size=(300, 11)).astype(np.int8)
    f = tmp_path / "matrix_chr7.txt.gz"
    write_tsv(f, x, chrom="chr7", start0=1000)
    plain_states, plain_loc = helpers.readTable(f, (17, 203))
    monkeypatch.setenv("EPILOGOS_CACHE_DIR", str(tmp_path / "cache"))
    for _ in range(2):                                           # first call fills the cache, second one maps it
        st, loc = helpers.readTable(f, (17, 203))
        assert st.dtype == np.int8 and st.flags["C_CONTIGUOUS"] and np.array_equal(st, plain_states)
        assert np.array_equal(loc.blob, plain_loc.blob) and np.array_equal(loc.offsets, plain_loc.offsets)
        helpers.flushCacheWrites()                               # the cache is written behind the first call's back
    files = sorted(p.name for p in (tmp_path / "cache").iterdir())
    assert len(files) == 4 and all(n.startswith("matrix_chr7_") for n in files)
    full, floc = helpers.readTable(f)
    assert np.array_equal(full, x) and len(floc) == 300
    assert helpers.readTable(f, (290, 400))[0].shape == (10, 11)           # clipped like the parser clips
    # a rewritten input is parsed again
    import os
    import time
    x2 = x.copy(); x2[0, 0] = (x2[0, 0] + 1) % 18
    write_tsv(f, x2, chrom="chr7", start0=1000)
    os.utime(f, ns=(time.time_ns() + 10**9, time.time_ns() + 10**9))
    assert np.array_equal(helpers.readTable(f)[0], x2)
    helpers.flushCacheWrites()
    assert len(list((tmp_path / "cache").iterdir())) == 8


def test_fast_gzip_round_trips():
    """The writer's own DEFLATE compressor (csrc/epg_deflate.h, gzip_level 0, the default): any inflate must read its
    members back -- empty input, one byte, incompressible bytes (stored blocks), long runs (length-258 matches at distance
    1), inputs around the 512 KiB block size, all byte values, Fibonacci-distributed symbols (a Huffman tree deeper than
    15 bits before it is flattened), repeats beyond the 32 KiB window, and score-like text."""
    import gzip
    import zlib
    rng = np.random.default_rng(1)

    def check(data):
        z = _io.gzip_fast(data)
        assert gzip.decompress(z) == data
        d = zlib.decompressobj(31)
        assert d.decompress(z) == data and d.eof and d.unused_data == b""
        return len(z)

    assert check(b"") <= 20 and check(b"a") <= 24
    assert check(bytes(3_000_000)) < 4000
    n = check(rng.integers(0, 256, 1_000_000, dtype=np.uint8).tobytes())
    assert 1_000_000 <= n <= 1_000_200                                   # stored: five bytes per 65535
    for size in (5, 6, 7, 8, 9, 15, 16, 17, 258, 259, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, (1 << 20) + 3):
        check((b"0.00000\t0.12345\t" * (size // 16 + 1))[:size])
        check(rng.integers(48, 58, size, dtype=np.uint8).tobytes())
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    data = np.frombuffer(b"".join(bytes([i + 1]) * f for i, f in enumerate(fib)), dtype=np.uint8)
    assert check(bytes(rng.permutation(data))) < data.size
    blk = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    check(blk + bytes(40000) + blk + blk)
    vals = (rng.random((20000, 18)) ** 6 * 2).astype(np.float32)
    text = b"".join(b"chr1\t%d\t%d\t" % (200 * r, 200 * r + 200) + _io.format_f5(vals[r]).rstrip(b"\t") + b"\n" for r in range(len(vals)))
    assert check(text) < 0.40 * len(text)


def test_writer_levels_agree(tmp_path):
    """gzip_level 0 (own compressor), 1 and 6 (zlib) hold the same text; EPILOGOS_GZIP_LEVEL picks the default."""
    import gzip
    rng = np.random.default_rng(2)
    R = 70000                                                              # three members of 32768 rows
    sc = (rng.random((R, S)) ** 4).astype(np.float32)
    blob = "".join("chr2\t%d\t%d\n" % (200 * r, 200 * r + 200) for r in range(R)).encode()
    off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum([len(l) + 1 for l in blob.decode().split("\n")[:-1]], out=off[1:])
    loc = _io.Locations(np.frombuffer(blob, dtype=np.uint8).copy(), off)
    texts = []
    for lvl in (0, 1, 6, None):
        _io.write_scores(tmp_path / "s.gz", loc, sc, gzip_level=lvl)
        texts.append(gzip.open(tmp_path / "s.gz", "rb").read())
    assert texts[0] == texts[1] == texts[2] == texts[3] and texts[0].count(b"\n") == R
    assert _io.default_gzip_level() == 0


def _table_text(R=3000, N=12, seed=5, crlf=False):
    rng = np.random.default_rng(seed)
    x = rng.integers(1, S + 1, size=(R, N))
    nl = "\r\n" if crlf else "\n"
    text = "".join("chr7\t%d\t%d\t" % (200 * r, 200 * r + 200) + "\t".join(map(str, x[r])) + nl for r in range(R)).encode()
    return x, text


def test_own_inflate_reads_what_any_deflate_wrote(tmp_path):
    """The reader's own inflate (csrc/epg_inflate.h) on members written by zlib at every strategy that changes the block types
    -- stored (level 0), fixed Huffman (Z_FIXED), dynamic (levels 1, 6, 9), Huffman-only and RLE --, by the library's own
    compressor, with a file name and a comment in the header, several members in one file, and trailing garbage; every result
    equals the text, and equals what the zlib path (EPGIO_INFLATE=zlib, exercised through a corrupt CRC below) returns."""
    import gzip
    import zlib
    x, text = _table_text()
    want = (x - 1).astype(np.int8)

    def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, header=b""):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        raw = c.compress(data) + c.flush()
        flg = 0
        extra = b""
        if header:
            flg = 8 | 16                                            # FNAME + FCOMMENT
            extra = header + b"\0" + b"a comment\0"
        return (b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\xff" + extra + raw
                + (zlib.crc32(data) & 0xffffffff).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little"))

    half = text.index(b"\n", len(text) // 2) + 1
    files = {
        "stored": member(text, 0), "fixed": member(text, 6, zlib.Z_FIXED), "l1": member(text, 1), "l9": member(text, 9),
        "huff": member(text, 6, zlib.Z_HUFFMAN_ONLY), "rle": member(text, 6, zlib.Z_RLE), "own": _io.gzip_fast(text),
        "named": member(text, 6, header=b"matrix_chr7.txt"),
        "two": member(text[:half], 6) + member(text[half:], 1, header=b"x"),
        "garbage": member(text, 6) + b"\0\0\0\0 not a gzip member",
        "empty_first": member(b"", 6) + member(text, 6),
    }
    for name, blob in files.items():
        p = tmp_path / (name + ".txt.gz")
        p.write_bytes(blob)
        st, loc = _io.read_table(p)
        assert np.array_equal(st, want), name
        assert len(loc) == len(want)


def test_own_inflate_never_trusts_itself(tmp_path):
    """A wrong CRC-32 or ISIZE, a flipped bit in the DEFLATE data and a truncated file are errors (the fast path hands such a
    file to zlib, whose verdict is reported), never silently different states."""
    import zlib
    x, text = _table_text(R=2000)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(text) + c.flush()
    head = b"\x1f\x8b\x08\0\0\0\0\0\0\xff"
    good = head + raw + (zlib.crc32(text) & 0xffffffff).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    p = tmp_path / "t.txt.gz"
    p.write_bytes(good)
    st, _ = _io.read_table(p)
    assert np.array_equal(st, (x - 1).astype(np.int8))
    bad_crc = good[:-8] + bytes([good[-8] ^ 1]) + good[-7:]
    bad_size = good[:-4] + (len(text) + 1).to_bytes(4, "little")
    flipped = bytearray(good)
    flipped[len(head) + len(raw) // 2] ^= 0x10
    for blob in (bad_crc, bad_size, bytes(flipped), good[: len(good) // 2], good[:-9]):
        p.write_bytes(blob)
        with pytest.raises(_io.EpilogosIOError):
            _io.read_table(p)


def _fuzz_outcomes(paths):
    import hashlib
    out = []
    for p in paths:
        try:
            st, loc = _io.read_table(p)
            out.append(hashlib.sha256(st.tobytes() + np.asarray(loc.blob).tobytes()).hexdigest())
        except _io.EpilogosIOError:
            out.append("error")
    return out


def test_own_inflate_differential_fuzz(tmp_path):
    """Differential fuzz of csrc/epg_inflate.h against zlib (round-2 advisory): a few hundred mutated gzip files -- bit flips
    in the header, the Huffman tables, the symbols and the trailer, truncations, spliced members, members whose matches would
    reach back into the PREVIOUS member's output -- are read in this process (own inflate first) and in a child process with
    EPGIO_INFLATE=zlib; every file must give the same verdict and the same bytes in both."""
    import json
    import os
    import subprocess
    import sys
    import zlib
    rng = np.random.default_rng(77)
    _, text = _table_text(R=150, N=24)

    def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, raw=None, crc_of=None):
        if raw is None:
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            raw = c.compress(data) + c.flush()
        crc_of = data if crc_of is None else crc_of
        return (b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + raw + (zlib.crc32(crc_of) & 0xffffffff).to_bytes(4, "little")
                + (len(crc_of) & 0xffffffff).to_bytes(4, "little"))

    half = text.index(b"\n", len(text) // 2) + 1
    seeds = [member(text, 6), member(text, 1), member(text, 9), member(text, 6, zlib.Z_FIXED), member(text, 6, zlib.Z_RLE),
             member(text, 6, zlib.Z_HUFFMAN_ONLY), member(text, 0), _io.gzip_fast(text), member(text[:half]) + member(text[half:], 1)]
    # second member compressed WITH the first as preset dictionary: its matches point before its own first byte; the CRC and
    # ISIZE in its trailer are those of the bytes a window-sharing decoder would produce, so only the distance check stops it
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, text[:half][-32768:])
    raw2 = c.compress(text[half:]) + c.flush()
    blobs = [member(text[:half]) + member(None, raw=raw2, crc_of=text[half:])] + seeds
    for k in range(260):
        b = bytearray(seeds[k % len(seeds)])
        kind = k % 5
        if kind == 0:
            b[int(rng.integers(10, min(len(b), 120)))] ^= 1 << int(rng.integers(0, 8))      # block header / code lengths
        elif kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            b = b[: int(rng.integers(1, len(b)))]
        elif kind == 3:
            i = int(rng.integers(10, len(b) - 8))
            b[i:i + int(rng.integers(1, 9))] = bytes(rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8))
        else:
            b[-int(rng.integers(1, 9))] ^= 1 << int(rng.integers(0, 8))                      # trailer
        blobs.append(bytes(b))
    paths = []
    for k, b in enumerate(blobs):
        p = tmp_path / ("f%03d.txt.gz" % k)
        p.write_bytes(b)
        paths.append(str(p))
    mine = _fuzz_outcomes(paths)
    (tmp_path / "paths.json").write_text(json.dumps(paths))
    env = dict(os.environ, EPGIO_INFLATE="zlib")
    code = ("import json, sys; sys.path.insert(0, %r); from tests.test_native_io import _fuzz_outcomes; "
            "print(json.dumps(_fuzz_outcomes(json.load(open(%r)))))" % (str(__import__("tests.conftest").conftest.ROOT), str(tmp_path / "paths.json")))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    theirs = json.loads(res.stdout.strip().splitlines()[-1])
    assert mine == theirs, [k for k in range(len(mine)) if mine[k] != theirs[k]]
    assert mine[0] == "error"                                   # the dictionary member: no decoder may accept it
    assert all(m != "error" for m in mine[1:1 + len(seeds)]) and len(set(mine[1:1 + len(seeds)])) == 1   # the unmutated files


def test_row_sums_and_rolling_max_have_numpys_and_pandas_bits():
    """STEP 4 helpers: epgio_row_sums_f32 equals ndarray.sum(axis=1) bit for bit (numpy adds a contiguous float32 row with
    eight accumulators, their tree sum, then the remainder) for every width up to 128 and for row-strided views;
    epgio_rolling_max_f64 equals Series.rolling(W, center=True).max() including the NaN edges, for odd and even windows,
    windows longer than the series, ties, and any thread count."""
    import pandas as pd
    rng = np.random.default_rng(0)
    for width in (1, 3, 7, 8, 9, 15, 16, 17, 18, 25, 31, 40, 127, 128):
        a = (rng.standard_normal((5000, width)) * rng.random((5000, 1)) * 100).astype(np.float32)
        assert np.array_equal(a.sum(axis=1), _io.row_sums(a)), width
        b = np.ascontiguousarray(rng.standard_normal((700, width + 5)).astype(np.float32))[:, :width]
        assert np.array_equal(b.sum(axis=1), _io.row_sums(b)), width
    assert np.array_equal(_io.row_sums(np.zeros((0, 18), dtype=np.float32)), np.zeros(0, dtype=np.float32))
    for W in (50, 7, 10, 125, 3, 2, 1, 64):
        for n in (0, 1, W - 1, W, W + 1, 1000, 1_200_000):
            x = rng.standard_normal(n)
            if n > 10:
                x[rng.integers(0, n, n // 20)] = 1.5                  # equal maxima
            want = pd.Series(x).rolling(W, center=True).max().to_numpy()
            for th in (1, 0):
                assert np.array_equal(want, _io.rolling_max(x, W, threads=th), equal_nan=True), (W, n, th)


def test_stored_block_in_the_last_bytes_of_a_member():
    """A stored block met while the decoder reads from its zero-padded copy of the last 16 input bytes (zlib's Z_SYNC_FLUSH
    leaves an empty one, `00 00 ff ff`, in front of the final block): the bytes given back from the bit buffer may stem from
    before that copy.  tools/asan_io.sh found the read of tail[-6..]; the own inflate now goes back to the real input there
    and ACCEPTS such members (it used to decline them -- on garbage -- and leave them to zlib)."""
    import zlib
    data = b"chr1\t0\t200\t1\t2\t3\n" * 50
    for pad in range(12):
        for level in (1, 6, 9):
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            raw = c.compress(data) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(b"x" * pad) + c.flush(zlib.Z_SYNC_FLUSH) + c.flush()
            whole = data + b"x" * pad
            g = (b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + raw + (zlib.crc32(whole) & 0xffffffff).to_bytes(4, "little")
                 + len(whole).to_bytes(4, "little"))
            assert _io.inflate_mem(g, own=True) == whole, (pad, level)
            assert _io.inflate_mem(g, own=False) == whole
            assert _io.inflate_mem(g[:-9], own=True) is None            # truncated trailer: declined, not over-read


def test_in_memory_differential_fuzz_short():
    """tools/fuzz_inflate.py (the harness tools/asan_io.sh runs for 10^5 streams under AddressSanitizer), a short leg."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_inflate.py"), "--streams", "4000", "--seed", "9"], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "no mismatch" in res.stdout, res.stdout + res.stderr


def test_sanitizer_recipe_quick(tmp_path):
    """The committed AddressSanitizer / UBSan recipe for the native host library, quick leg: builds
    csrc/epg_io.cpp with -fsanitize=address,undefined, runs this file's parser / writer / codec tests and 3000 fuzz streams
    against that build.  Skipped where g++ has no libasan.  The full leg (10^5 streams) is tools/asan_io.sh; its log is kept
    in profiles/."""
    import os
    import subprocess
    from tests.conftest import ROOT
    if os.environ.get("EPILOGOS_IO_LIB"):
        pytest.skip("already running against another build of the library")
    res = subprocess.run([str(ROOT / "tools" / "asan_io.sh"), "--quick", str(tmp_path)], capture_output=True, text=True, timeout=1500)
    if res.returncode == 77:
        pytest.skip("no libasan for g++ here")
    assert res.returncode == 0 and "asan_io: clean" in res.stdout, (res.stdout + res.stderr)[-4000:]


def test_bgzf_blocks_are_read_in_parallel_and_written_on_request(tmp_path, monkeypatch):
    """BGZF (htslib's blocked gzip: members of <= 64 KiB carrying their compressed size in a 'BC' extra subfield).  Reader: such a
    file's blocks are inflated in parallel (epg_io.cpp inflate_bgzf) -- same states and coordinates as the plain-gzip file, for
    any thread count, with or without the end marker; a block that is not what its header says sends the file to the general
    reader and zlib, with their verdict.  Writer: EPILOGOS_BGZF=1 makes every writer emit BGZF -- the same decompressed bytes,
    'BC' subfield, <= 64 KiB blocks, the 28-byte end marker."""
    import zlib
    sys_path = __import__("sys").path
    sys_path.insert(0, str(__import__("tests.conftest").conftest.ROOT / "tools"))
    from fuzz_inflate import BGZF_EOF, bgzf
    x, text = _table_text(R=900, N=61)
    plain = tmp_path / "p_chr1.txt.gz"
    plain.write_bytes(gzip.compress(text, 6))
    want_s, want_l = _io.read_table(plain)
    for k, blob in enumerate((bgzf(text, 6, 60000), bgzf(text, 1, 3000, eof=False), bgzf(text, 0, 65280), bgzf(text[:0]))):
        f = tmp_path / ("b%d_chr1.txt.gz" % k)
        f.write_bytes(blob)
        if k < 3:
            assert _io.inflate_mem(blob, own=2, cap=1 << 24) == text
            for threads in (0, 1, 3):
                s, l = _io.read_table(f, threads=threads)
                assert np.array_equal(s, want_s) and np.array_equal(l.blob, want_l.blob) and np.array_equal(l.offsets, want_l.offsets)
        else:
            s, l = _io.read_table(f)
            assert s.shape[0] == 0 and len(l) == 0
    # a block whose trailer lies about its size / whose payload is damaged: not accepted by the block reader, judged like zlib does
    good = bgzf(text, 6, 5000)
    for pos in (len(good) // 2, 30, len(good) - 40):
        bad = bytearray(good)
        bad[pos] ^= 0x20
        bad = bytes(bad)
        mine, theirs = _io.inflate_mem(bad, own=2, cap=1 << 24), _io.inflate_mem(bad, own=False, cap=1 << 24)
        assert mine is None or mine == theirs
        (tmp_path / "bad_chr1.txt.gz").write_bytes(bad)
        if theirs is None:
            with pytest.raises(_io.EpilogosIOError):
                _io.read_table(tmp_path / "bad_chr1.txt.gz")
    assert _io.inflate_mem(plain.read_bytes(), own=2) is None          # plain gzip is not blocked gzip
    # the writers
    monkeypatch.setenv("EPILOGOS_BGZF", "1")
    st = (x - 1).astype(np.int8)
    _io.write_states(tmp_path / "w_chr1.txt.gz", "chr1", st, gzip_level=1)
    sc = np.random.default_rng(1).standard_normal((900, 18)).astype(np.float32)
    _io.write_scores(tmp_path / "w_scores.txt.gz", want_l, sc)
    monkeypatch.delenv("EPILOGOS_BGZF")
    _io.write_states(tmp_path / "v_chr1.txt.gz", "chr1", st, gzip_level=1)
    _io.write_scores(tmp_path / "v_scores.txt.gz", want_l, sc)
    for name in ("chr1", "scores"):
        b = (tmp_path / ("w_%s.txt.gz" % name)).read_bytes()
        assert b[:4] == b"\x1f\x8b\x08\x04" and b[12:16] == b"BC\x02\x00" and b.endswith(BGZF_EOF)
        assert gzip.decompress(b) == gzip.decompress((tmp_path / ("v_%s.txt.gz" % name)).read_bytes())
        pos = 0
        while pos < len(b):                                            # every block says how long it is, and is <= 64 KiB
            size = int.from_bytes(b[pos + 16:pos + 18], "little") + 1
            assert b[pos:pos + 4] == b"\x1f\x8b\x08\x04" and size <= 65536
            pos += size
        assert pos == len(b)
    s, _l = _io.read_table(tmp_path / "w_chr1.txt.gz")
    assert np.array_equal(s, st)


def test_bgzf_member_with_an_empty_payload_is_declined():
    """a BGZF member whose deflate payload is EMPTY (BSIZE says 26 bytes in all, CRC 0, ISIZE 0) is not a deflate
    stream -- zlib rejects it -- and must not pass the block reader's checks as "0 bytes used of 0" (a truncated or corrupt
    block would be read as empty text).  The regular end-of-file block (two payload bytes) still passes."""
    import zlib
    from tools.fuzz_inflate import BGZF_EOF, bgzf
    good = bgzf(b"chr1\t0\t200\t1\t2\n" * 50, 6, 300)
    assert _io.inflate_mem(good, own=2) == b"chr1\t0\t200\t1\t2\n" * 50
    empty = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 25, 0]) + (0).to_bytes(4, "little") + (0).to_bytes(4, "little")
    for blob in (good[:-len(BGZF_EOF)] + empty + BGZF_EOF, empty + good, good[:-len(BGZF_EOF)] + empty):
        with pytest.raises((zlib.error, EOFError, gzip.BadGzipFile)):
            gzip.decompress(blob)                              # (every member is read: the one without a deflate stream fails)
        assert _io.inflate_mem(blob, own=2) is None, "the block reader accepted a member without a deflate stream"
        assert _io.inflate_mem(blob, own=False) is None


def test_host_budget_reaches_the_library_through_the_setter(monkeypatch):
    """the native library must not read the environment from its reader threads while Python rewrites it.  The
    variable is read once; _io.host_budget() hands later values over with epgio_set_host_threads."""
    lib = _io.load()
    before = lib.epgio_default_threads()
    try:
        monkeypatch.setenv("EPILOGOS_NUM_CORES", "3")
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
        assert _io.host_budget() == min(3, _io.node_cores()) == lib.epgio_default_threads()
        monkeypatch.setenv("EPILOGOS_HOST_THREADS", "57")      # changing the variable behind the library's back does nothing
        assert lib.epgio_default_threads() == min(3, _io.node_cores())
        lib.epgio_set_host_threads(5)
        assert lib.epgio_default_threads() == 5
    finally:
        monkeypatch.delenv("EPILOGOS_NUM_CORES", raising=False)
        monkeypatch.delenv("EPILOGOS_HOST_THREADS", raising=False)
        _io._budget_told = None
        lib.epgio_set_host_threads(0)
        assert lib.epgio_default_threads() >= 1
        _io.host_budget()


def test_lenient_inputs_are_read_like_the_reference_reads_them(tmp_path, capsys):
    """Inputs pandas reads and the strict native parser refuses -- a blank line inside the file, blanks around a number, "+1",
    "1.0" -- must give the arrays the REFERENCE's helpers.readStates gives (helpers.py:152-155: pandas.read_table), and the
    locations it would print (scores.py:161,526-531): helpers.readTable re-reads such a file through pandas and says so.  The
    expected arrays were produced by the real reference (tests/golden/make_golden_lenient.py)."""
    from tests.conftest import load_golden
    g = load_golden("lenient.npz")
    names = sorted(k[:-5] for k in g if k.endswith("_text"))
    assert {"blank_line", "blanks_around", "plus_sign", "float_state", "all_together", "crlf"} <= set(names)
    for name in names:
        f = tmp_path / (name + ".txt")
        f.write_bytes(g[name + "_text"].tobytes())
        want = g[name + "_states"]
        strict_ok = True
        try:
            _io.read_table(f)
        except _io.EpilogosIOError as e:
            strict_ok = False
            assert "malformed line" in str(e)
        assert strict_ok == (name == "crlf")                       # (CRLF line ends are read natively)
        st, loc = helpers.readTable(f)
        assert st.dtype == np.int8 and np.array_equal(st, want), name
        assert loc.blob.tobytes().decode().splitlines() == list(g[name + "_loc"]), name
        assert ("through pandas" in capsys.readouterr().out) == (not strict_ok)
        # a row range, a caller-owned padded destination and the file's value range, as the driver asks for them
        R = want.shape[0]
        got = {}

        def alloc(r, n):
            got["buf"] = np.full((r, 16), 99, dtype=np.int8)
            return got["buf"]
        st2, loc2 = helpers.readTable(f, (1, R), alloc=alloc)
        assert st2 is got["buf"] and np.array_equal(st2[:, :want.shape[1]], want[1:]) and (st2[:, want.shape[1]:] == -1).all()
        assert len(loc2) == R - 1
        assert helpers.readTable(f, with_range=True)[2] == (int(want.min()) + 1, int(want.max()) + 1)
        assert np.array_equal(helpers.readStates(f, rowsToCalc=(0, int(g[name + "_countRows"]))), want)
        capsys.readouterr()
    # what pandas cannot turn into integers raises, as it does in the reference
    (tmp_path / "alpha.txt").write_text("chr1\t0\t200\t1\tx\nchr1\t200\t400\t1\t2\n")
    with pytest.raises(ValueError):
        helpers.readTable(tmp_path / "alpha.txt")
    (tmp_path / "short.txt").write_text("chr1\t0\t200\t1\t2\nchr1\t200\t400\t1\n")
    with pytest.raises(ValueError):
        helpers.readTable(tmp_path / "short.txt")
