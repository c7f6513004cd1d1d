"""The device BGZF codec (crumble_gz_* of include/crumble_hip.h; csrc/gz_kernels.h): SURVEY.md 8 f1 "and/or GPU inflate;
symmetric BGZF deflate writer".

The checker is zlib (Python's binding): members written by zlib at levels 0 / 1 / 6 / 9, with fixed and dynamic Huffman
codes, stored blocks, several DEFLATE blocks per member, and by the crumble-hip host writer (libdeflate when the box has it)
must inflate on the device to zlib's bytes with the CRC-32 checked; what the device deflates must inflate under zlib to the
input, member by member, with gzip's own CRC-32 / ISIZE check.  Corrupt members are refused with an error code.
The emulated tier (no GPU) runs the literal kernel sources on small inputs; the GPU tier runs full-size blocks."""
import struct
import zlib

import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import gz

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flushes=()):
    """one BGZF member around zlib's raw DEFLATE of `raw`; flushes: byte positions at which zlib is made to end a block
    (Z_FULL_FLUSH: a member with several DEFLATE blocks, stored empty ones between them)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    payload, last = b"", 0
    for f in flushes:
        payload += co.compress(raw[last:f]) + co.flush(zlib.Z_FULL_FLUSH)
        last = f
    payload += co.compress(raw[last:]) + co.flush()
    total = 18 + len(payload) + 8
    assert total <= 65536
    return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", total - 1) + payload +
            struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))


def bgzf(data, level=6, blk=0xff00, **kw):
    return b"".join(member(data[i:i + blk], level, **kw) for i in range(0, len(data), blk))


def gunzip_members(comp):
    """zlib's gzip decoder member by member (checks every CRC-32 and ISIZE) -> (bytes, number of members)"""
    out, pos, n = [], 0, 0
    while pos < len(comp):
        d = zlib.decompressobj(31)
        out.append(d.decompress(comp[pos:]))
        assert d.eof
        pos = len(comp) - len(d.unused_data)
        n += 1
    return b"".join(out), n


def samples(rng, n):
    """byte strings shaped like what a BAM holds: packed bases (sixteen byte values), quality runs as crumble leaves them,
    names / tags that repeat at a record's distance, noise, zeros"""
    acgt = np.frombuffer(bytes(a << 4 | b for a in (1, 2, 4, 8) for b in (1, 2, 4, 8)), np.uint8)
    rec = []
    for i in range(max(1, n // 300)):
        name = b"r%07d\0" % i
        seq = rng.choice(acgt, 75).tobytes()
        q = bytearray(b"I" * 150)
        for k in rng.integers(0, 150, 3):
            q[k] = 38
        rec.append(struct.pack("<iiBBHHHiiii", 0, 1000 + 3 * i, len(name), 60, 4680, 1, 99, 150, 0, 1200 + 3 * i, 350) + name +
                   struct.pack("<I", 150 << 4) + seq + bytes(q) + b"NMC\0XSc\x05")
    recs = b"".join(struct.pack("<i", len(r)) + r for r in rec)[:n]
    return {
        "records": recs + bytes(n - len(recs)),
        "bases": rng.choice(acgt, n).tobytes(),
        "runs": b"".join(bytes([int(rng.integers(33, 75))]) * int(rng.integers(1, 400)) for _ in range(n // 100 + 1))[:n],
        "noise": rng.integers(0, 256, n, dtype=np.uint8).tobytes(),
        "zeros": bytes(n),
        "period3": (b"abc" * (n // 3 + 1))[:n],
    }


def _inflate_cases(codec, sizes, levels, kinds=None):
    rng = np.random.default_rng(11)
    for n in sizes:
        for kind, data in samples(rng, n).items():
            if kinds and n > 1000 and kind not in kinds:         # (the emulated tier: the large sizes on the kinds that differ there)
                continue
            for level in levels:
                comp = bgzf(data, level)
                got = codec.inflate(comp)
                assert got.tobytes() == data, (kind, n, level)
    # fixed Huffman codes; several DEFLATE blocks (and empty stored ones) in one member; empty members between full ones
    data = samples(rng, 9000)["records"]
    assert codec.inflate(bgzf(data, 6, strategy=zlib.Z_FIXED)).tobytes() == data
    assert codec.inflate(member(data, 6, flushes=(100, 100, 4000, 8999))).tobytes() == data
    assert codec.inflate(member(data[:3000], 1) + BGZF_EOF + member(data[3000:], 9) + BGZF_EOF).tobytes() == data
    assert codec.inflate(b"").size == 0 and codec.inflate(BGZF_EOF).size == 0


def _corrupt_cases(codec):
    """a flipped payload bit, a wrong CRC-32, a wrong ISIZE, a truncated payload: error codes, and the codec stays usable"""
    rng = np.random.default_rng(12)
    data = samples(rng, 5000)["records"]
    good = member(data, 6)
    assert codec.inflate(good).tobytes() == data
    bad_crc = bytearray(good); bad_crc[-8] ^= 1
    bad_size = bytearray(good); bad_size[-4] ^= 1
    with pytest.raises(ca.CrumbleError) as ei:
        codec.inflate(bytes(bad_crc))
    assert ei.value.code == -1 and "CRC" in str(ei.value)
    with pytest.raises(ca.CrumbleError):
        codec.inflate(bytes(bad_size))
    assert codec.inflate(bytes(bad_crc), check_crc=False).tobytes() == data
    n_refused = 0
    for k in range(40):
        b = bytearray(good)
        b[18 + int(rng.integers(0, len(good) - 26))] ^= 1 << int(rng.integers(0, 8))
        try:
            out = codec.inflate(bytes(b))
            assert out.tobytes() != data or bytes(b) == good        # (a flip in unused header bits may be harmless)
        except ca.CrumbleError as e:
            assert e.code == -1
            n_refused += 1
    assert n_refused >= 35                                          # CRC-32 catches what the decoder does not
    # the payload cut short: clen says less than the stream needs
    blk, consumed, total = gz.bgzf_scan(good, codec.L)
    blk = blk.copy(); blk["clen"][0] -= 40
    with pytest.raises(ca.CrumbleError):
        codec.inflate(good, blocks=blk)
    assert codec.inflate(good).tobytes() == data
    with pytest.raises(ValueError):
        gz.bgzf_scan(b"\x1f\x8b\x08\x00" + bytes(30), codec.L)     # gzip without the BC field is not BGZF


def _deflate_cases(codec, sizes):
    rng = np.random.default_rng(13)
    for n in sizes:
        for kind, data in samples(rng, n).items():
            for level in (0, 6):
                comp = codec.deflate(data, level).tobytes()
                back, nm = gunzip_members(comp)
                assert back == data, (kind, n, level)
                assert nm == (n + 0xff00 - 1) // 0xff00
                blk, consumed, total = gz.bgzf_scan(comp, codec.L)
                assert consumed == len(comp) and total == n and len(blk) == nm
                assert all(int(b["ulen"]) == min(0xff00, n - i * 0xff00) for i, b in enumerate(blk))
                assert codec.inflate(comp).tobytes() == data                  # and the device reads its own output
                if level and kind in ("runs", "zeros", "period3", "records") and n >= 1000:
                    assert len(comp) < 0.5 * n, (kind, n, len(comp))
                if level and kind == "noise":
                    assert len(comp) <= n + 31 * nm                         # stored when nothing is gained
                if n >= 2500:
                    assert codec.deflate(data, level).tobytes() == comp       # deterministic
    assert codec.deflate(b"", 6).size == 0
    data = samples(rng, 3000)["records"]                                 # zlib's "default" level (-1) compresses, 0 alone stores
    assert codec.deflate(data, -1).tobytes() == codec.deflate(data, 6).tobytes() and len(codec.deflate(data, 0)) > len(data)


# ------------------------------------------------------------------------------------------ emulated tier (no GPU)
@pytest.fixture(scope="module")
def emu_codec():
    import emu_lib
    c = gz.GzCodec(0, emu_lib.load())
    yield c
    c.close()


def _walk(codec, tokens):
    """tokens: wave | lane | lane16 | lane64 (the lane walk with 8 -- its choice for few members --, 16 or 64 walking lanes per wave)"""
    codec.set_option("tokens", tokens[:4])
    codec.set_option("lanes_per_wave", tokens[4:] or "auto")


@pytest.mark.parametrize("tokens", ["wave", "lane", "lane16", "lane64"])
def test_emu_inflate_zlib_members(emu_codec, tokens):
    """both walks: a wave per member (k_gz_tokens) and a lane per member (k_gz_tokens_lanes)"""
    _walk(emu_codec, tokens)
    if tokens in ("lane16", "lane64"):                       # (the same kernel source as "lane", another stride: a shorter list)
        _inflate_cases(emu_codec, (7, 20000), (1, 6), kinds=("records",))
    else:
        _inflate_cases(emu_codec, (1, 7, 700, 70000), (0, 1, 6), kinds=("records", "runs"))
    _walk(emu_codec, "auto")


@pytest.mark.parametrize("tokens", ["wave", "lane"])
def test_emu_inflate_refuses_corrupt_members(emu_codec, tokens):
    _walk(emu_codec, tokens)
    _corrupt_cases(emu_codec)
    _walk(emu_codec, "auto")


def test_emu_deflate_round_trip(emu_codec):
    _deflate_cases(emu_codec, (1, 300, 2500))


def test_emu_deflate_full_block(emu_codec):
    """one block of the full 0xff00 bytes and the byte behind it (two members): every sub-block step, the carry across
    them, the bit buffer's far end (slow under emulation: 1024 fibers)"""
    rng = np.random.default_rng(14)
    data = samples(rng, 0xff00 + 1)["records"]
    comp = emu_codec.deflate(data, 6).tobytes()
    back, nm = gunzip_members(comp)
    assert back == data and nm == 2
    assert len(comp) < len(bgzf(data, 6)) * 1.15        # within 15 % of zlib -6 on record-shaped bytes


def test_no_gpu_no_codec(product):
    """like the engine: without a HIP device the codec refuses to exist (no CPU fallback in the product)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ca.CrumbleError) as ei:
        gz.GzCodec(0)
    assert ei.value.code == -6


# ------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def gpu_codec(product):
    c = gz.GzCodec(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tokens", ["wave", "lane", "lane16", "lane64"])
def test_gpu_inflate_zlib_members(gpu_codec, tokens):
    _walk(gpu_codec, tokens)
    _inflate_cases(gpu_codec, (1, 7, 700, 70000, 1 << 20), (0, 1, 6, 9))
    _walk(gpu_codec, "auto")


@pytest.mark.gpu
@pytest.mark.parametrize("tokens", ["wave", "lane", "lane16", "lane64"])
def test_gpu_inflate_refuses_corrupt_members(gpu_codec, tokens):
    _walk(gpu_codec, tokens)
    _corrupt_cases(gpu_codec)
    _walk(gpu_codec, "auto")


@pytest.mark.gpu
def test_gpu_deflate_round_trip(gpu_codec):
    _deflate_cases(gpu_codec, (1, 5, 300, 2500, 0xff00, 0xff00 + 1, 3 << 20))


@pytest.mark.gpu
def test_gpu_codec_on_a_bam(gpu_codec, tmp_path):
    """a BAM written by the host writer (libdeflate where the box has it, zlib otherwise; both codecs, three levels):
    inflated on the device = gzip's bytes; deflated again on the device, the host READER takes the file and finds the
    same records; the size stays within 12 % of the host codec's level 6"""
    import gzip
    import os
    import synth
    from crumble_amd import bamio
    hb = synth.generate(synth.config("chr20", scale=0.004))
    sizes = {}
    for zl in ("0", "1"):
        os.environ["CRUMBLE_BAM_ZLIB"] = zl
        for level in (1, 6, 9):
            path = str(tmp_path / f"in_{zl}_{level}.bam")
            bamio.write_batches(path, [("chr20", 70000000)], [hb], level=level)
            comp = open(path, "rb").read()
            want = gzip.decompress(comp)
            got = gpu_codec.inflate(comp)
            assert got.tobytes() == want, (zl, level)
            sizes[(zl, level)] = len(comp)
    os.environ.pop("CRUMBLE_BAM_ZLIB", None)
    comp2 = gpu_codec.deflate(want, 6).tobytes() + BGZF_EOF
    assert gzip.decompress(comp2) == want
    p2 = str(tmp_path / "dev.bam")
    open(p2, "wb").write(comp2)
    rd = bamio.Reader(p2)
    b = rd.next_contig()
    assert b is not None and b.n_reads == hb.n_reads and np.array_equal(b.quals, hb.quals) and np.array_equal(b.pos, hb.pos)
    rd.close()
    assert len(comp2) < 1.12 * sizes[("1", 6)], (len(comp2), sizes)
