"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

import crumble_amd
from crumble_amd.capi import ROOT


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(crumble_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol(product):
    L = C.CDLL(crumble_amd.LIB_PATH)
    names = _declared("crumble_hip.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"


def test_abi_version_handshake(product):
    """crumble_hip_abi_version() is the header's CRUMBLE_HIP_ABI, crumble_hip_struct_size() the sizes the ctypes mirrors have
    (a host built against an older header must be able to notice), and crumble_set_host_threads exists."""
    from crumble_amd import capi
    L = crumble_amd.lib()
    txt = open(os.path.join(ROOT, "include", "crumble_hip.h")).read()
    abi = int(re.search(r"#define\s+CRUMBLE_HIP_ABI\s+(\d+)", txt).group(1))
    assert L.crumble_hip_abi_version() == abi == capi.ABI
    want = [C.sizeof(capi.Batch), C.sizeof(capi.Params), C.sizeof(capi.Coded), 32, 32, C.sizeof(capi.Column)]
    assert [L.crumble_hip_struct_size(i) for i in range(6)] == want and L.crumble_hip_struct_size(6) == 0
    L.crumble_set_host_threads(3); L.crumble_set_host_threads(0)


def test_bamio_exports_every_declared_symbol(product):
    L = C.CDLL(os.path.join(ROOT, "crumble_amd", "libcrumble_bamio.so"))
    names = [n for n in _declared("crumble_bamio.h") if n.startswith("crumble_bam_")]
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/crumble_bamio.h but not exported"


def test_presets_match_reference_table(product):
    """SURVEY.md App. E / snp_score.c:2380-2482."""
    want = {
        9: (8, 70, 125, 1.5, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.0, 2, 0),
        8: (0, 70, 125, 1.5, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.0, 2, 0),
        7: (0, 75, 150, 1.0, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.1, 2, 0),
        5: (0, 75, 150, 1.0, 0.5, 0.1, 0.5, 3.0, 0.0, 0, 1.1, 2, 0),
        3: (0, 75, 150, 1.0, 0.5, 0.1, 0.5, 3.0, 1.0, 0, 1.1, 2, 0),
        1: (0, 75, 150, 1.0, 0.5, 0.1, 0.5, 3.0, 1.0, 5, 2.0, 1, 5),
    }
    for lvl, w in want.items():
        p = crumble_amd.preset_params(lvl)
        got = (p.pblock, p.min_qual_B, p.min_indel_B, p.min_discrep_B, p.low_mqual_perc,
               p.ins_len_perc, p.indel_ov_perc, p.over_depth, p.sSTR_mul, p.sSTR_add,
               p.iSTR_mul, p.iSTR_add, p.min_mqual)
        assert got == w, lvl
        assert (p.qlow, p.qcutoff, p.qhigh, p.qcap, p.min_qual_A) == (5, 25, 40, 60, 0)
    for bad in (2, 4, 6, 0):
        p = crumble_amd.default_params()
        assert crumble_amd.lib().crumble_params_preset(C.byref(p), bad) != 0
    p = crumble_amd.preset_params(None, tech="pbccs")
    assert (p.indel_fract, p.min_discrep_B, p.qcutoff, p.qhigh, p.pblock, p.perfect_col,
            p.preserve_qual[93]) == (0.1, 0.8, 40, 60, 16, 1, 1)
    p = crumble_amd.preset_params(9, keep="2-4,40", really_keep="93")
    assert [p.preserve_qual[i] for i in (1, 2, 3, 4, 5, 40, 93)] == [0, 1, 1, 1, 0, 1, 2]


def test_create_fails_loudly_without_gpu(product):
    """No CPU fallback: without a HIP device the engine refuses to exist."""
    from parity_util import gpu_count
    if gpu_count():
        return
    try:
        crumble_amd.Engine(crumble_amd.preset_params(9))
    except crumble_amd.CrumbleError as e:
        assert e.code in (-6, -3)
    else:
        raise AssertionError("engine was created without a GPU")


def test_bed_collapse_semantics(product):
    import numpy as np
    L = crumble_amd.lib()
    tid = np.array([0, 0, 0, 1], np.int32)
    st = np.array([25, 0, 10, 5], np.int32)
    en = np.array([50, 30, 20, 9], np.int32)
    n = L.crumble_bed_collapse(tid.ctypes.data, st.ctypes.data, en.ctypes.data, 4)
    # bed.c:20-40 compares with the previous INPUT region's end: [25,50] is not merged into [0,30]
    assert n == 3
    assert list(zip(tid[:n], st[:n], en[:n])) == [(0, 0, 30), (0, 25, 50), (1, 5, 9)]


def test_no_exec_in_the_product():
    """crumble-hip must never replace its process image: under a profiler's preload the GPU is initialised before main(),
    and an exec from such a process takes the machine down on this pool (round-4 verdict / ADVICE).  Idle host threads
    sleep on a condition variable instead (csrc/team.h), so no OMP_WAIT_POLICY re-exec is needed."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crumble_amd")
    pat = re.compile(r"\b(execv|execve|execvp|execvpe|execl|execlp|execle|fexecve|posix_spawn|posix_spawnp|system|popen)\s*\(")
    hits = []
    for f in glob.glob(os.path.join(root, "csrc", "*")):
        if f.endswith((".c", ".h", ".hip", ".cpp")):
            for i, line in enumerate(open(f, errors="replace"), 1):
                if pat.search(line):
                    hits.append((os.path.basename(f), i, line.strip()))
    assert not hits, hits
    # and no OpenMP runtime behind the host libraries (its wait policy is fixed when it is loaded)
    import subprocess
    for so in ("libcrumble_hip.so", "libcrumble_bamio.so", "crumble-hip"):
        path = os.path.join(root, so)
        if os.path.exists(path):
            out = subprocess.run(["ldd", path], stdout=subprocess.PIPE, text=True).stdout
            assert "libgomp" not in out and "libomp" not in out, (so, out)


def test_coded_expansion_host_side(product):
    """crumble_decode_seek / _run / _quals (plain C, no device): a hand-made coded result -- every code, escapes on both
    sides of the 4096-base block borders, a last partial word -- expands to what numpy computes; through the cursor from
    arbitrary starts, in place, with skipped runs (records without qualities)."""
    import numpy as np
    from crumble_amd import Coded, DecodeCursor
    L = crumble_amd.lib()
    rng = np.random.default_rng(5)
    for nb in (0, 1, 15, 16, 17, 4095, 4096, 4097, 3 * 4096 + 5, 70001):
        codes_b = rng.choice(4, nb, p=[0.6, 0.15, 0.2, 0.05]).astype(np.uint32)
        if nb > 4100:
            codes_b[4090:4102] = 3                                     # escapes across a block border
            codes_b[8192 - 16:8192 + 16] = [0, 1] * 16                 # whole words of the two values
        qin = rng.integers(0, 94, nb).astype(np.uint8)
        cap = np.minimum(np.arange(256), 60).astype(np.uint8)
        cap[70] = 70                                                   # a preserved value above the cap
        esc_vals = rng.integers(0, 100, int((codes_b == 3).sum())).astype(np.uint8)
        want = np.where(codes_b == 0, 40, np.where(codes_b == 1, 5, cap[qin])).astype(np.uint8)
        want[codes_b == 3] = esc_vals
        nw, nblk = (nb + 15) // 16, (nb + 4095) // 4096
        words = np.zeros(max(1, nw), np.uint32)
        for i in range(16):
            part = codes_b[i::16]
            words[:len(part)] |= part << np.uint32(2 * i)
        ne_before = np.concatenate([[0], np.cumsum(codes_b == 3)])
        esc_off = np.array([ne_before[min(nb, b * 4096)] for b in range(max(1, nblk))], np.int64)
        c = Coded()
        c.n_bases, c.n_words, c.n_blocks, c.n_escapes = nb, nw, nblk, len(esc_vals)
        c.value[0], c.value[1] = 40, 5
        for q in range(256):
            c.cap[q] = int(cap[q])
        esc = np.concatenate([esc_vals, np.zeros(1, np.uint8)])
        c.codes, c.esc_off, c.esc = words.ctypes.data, esc_off.ctypes.data, esc.ctypes.data
        out = np.full(max(1, nb), 0xEE, np.uint8)
        if nb:
            L.crumble_decode_quals(C.byref(c), qin.ctypes.data, out.ctypes.data, 3)
            assert np.array_equal(out[:nb], want), (nb, np.nonzero(out[:nb] != want)[0][:10])
        # the cursor: runs of random lengths from a random start, in place, some skipped
        buf = qin.copy()
        start = int(rng.integers(0, nb + 1))
        cur = DecodeCursor()
        L.crumble_decode_seek(C.byref(c), start, C.byref(cur))
        b = start
        while b < nb:
            n = int(min(nb - b, rng.integers(0, 300)))
            skip = bool(rng.integers(0, 4) == 0)
            L.crumble_decode_run(C.byref(c), C.byref(cur), n, buf.ctypes.data + b, None if skip else buf.ctypes.data + b)
            assert np.array_equal(buf[b:b + n], qin[b:b + n] if skip else want[b:b + n]), (nb, b, n, skip)
            b += n
        assert cur.base == max(start, nb) or nb == 0


def test_pack_quals_host_side(product):
    """crumble_pack_quals (plain C, no device): 2-bit codes for up to four distinct qualities, 4-bit codes for up to sixteen,
    0 (hand over the bytes) for more; every length around the byte and thread-piece borders; the table holds exactly the
    values that occur, and table[code(i)] is quality i -- what k_p0_quals computes on the device."""
    import numpy as np
    L = crumble_amd.lib()
    L.crumble_pack_quals.restype = C.c_int
    L.crumble_pack_quals.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    pools = {1: [37], 2: [0, 255], 3: [2, 11, 40], 4: [2, 11, 25, 37], 5: [2, 11, 25, 37, 40], 8: list(range(20, 28)),
             16: list(range(0, 64, 4)), 17: list(range(17)), 40: list(range(2, 42))}
    for nv, vals in pools.items():
        for nb in (0, 1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 4099, (1 << 16) + 3, (1 << 20) + 5, 3 * (1 << 20) + 1):
            if nb and nb < nv:
                continue
            q = np.asarray(vals, np.uint8)[rng.integers(0, len(vals), nb)]
            if nb:
                q[rng.permutation(nb)[:len(vals)]] = vals              # every value of the pool occurs
            out = np.full((nb * 4 + 7) // 8 + 16, 0xEE, np.uint8)
            lut = np.full(16, 0xEE, np.uint8)
            bits = L.crumble_pack_quals(nb, q.ctypes.data if nb else None, out.ctypes.data, lut.ctypes.data)
            # out == NULL is the look alone (a caller decides before it allocates): the same width, nothing written
            assert L.crumble_pack_quals(nb, q.ctypes.data if nb else None, None, np.zeros(16, np.uint8).ctypes.data) == bits
            if nv > 16 and nb:
                assert bits == 0, (nv, nb)
                continue
            assert bits == (2 if (nv <= 4 or nb == 0) else 4), (nv, nb, bits)
            if not nb:
                continue
            assert sorted(lut[:nv].tolist()) == sorted(vals) and not lut[nv:].any(), (nv, nb, lut)
            idx = np.arange(nb, dtype=np.int64)
            codes = (out[(idx * bits) >> 3] >> ((idx * bits) & 7).astype(np.uint8)) & ((1 << bits) - 1)
            assert np.array_equal(lut[codes], q), (nv, nb)
            assert not out[(nb * bits + 7) // 8:(nb * bits + 7) // 8 + 16].any()      # the slack behind the codes is zeroed
