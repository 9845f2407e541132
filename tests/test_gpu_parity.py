"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.
Bit-exact everywhere (all stages are integer / byte / index work)."""
import numpy as np
import pytest

import dark_amd
from conftest import seeded_inputs

pytestmark = pytest.mark.gpu
MODELS = ("dark", "exp", "ybs", "simple")
CAP = 6 << 20


@pytest.fixture(scope="module")
def ctx():
    c = dark_amd.Context(CAP)
    yield c
    c.close()


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if len(a) != len(b):
        return "len %d vs %d" % (len(a), len(b))
    bad = np.flatnonzero(a != b)
    if len(bad) == 0:
        return None
    i = int(bad[0])
    return "%d mismatches, first at %d: got %s want %s" % (len(bad), i, a[i:i + 8], b[i:i + 8])


def text_like(rng, n, vocab=2000):
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(vocab)]
    ids = rng.zipf(1.25, size=n // 3 + 16) % vocab
    out = b" ".join(words[int(i)] for i in ids)
    return np.frombuffer(out[:n], dtype=np.uint8)


def test_sort_pairs(ctx):
    rng = np.random.default_rng(1)
    for count, lo, hi in ((1, 0, 64), (2, 0, 64), (257, 0, 64), (4096, 0, 64), (4097, 0, 64), (100003, 0, 64),
                          (1 << 20, 0, 64), (50000, 0, 8), (50000, 8, 24), (300000, 0, 40), (8191, 16, 64)):
        keys = rng.integers(0, 1 << 63, size=count, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=count, dtype=np.uint64)
        if count % 2 == 1:
            keys &= np.uint64(0x00FF00FF00FF00FF)  # many duplicates: stability matters
        vals = np.arange(count, dtype=np.uint32)
        k2, v2 = ctx.dbg_sort_pairs(keys, vals, lo, hi)
        mask = np.uint64(((1 << hi) - 1) ^ ((1 << lo) - 1)) if hi < 64 else np.uint64(~((1 << lo) - 1) & 0xFFFFFFFFFFFFFFFF)
        order = np.argsort(keys & mask, kind="stable")
        assert first_diff(v2, vals[order]) is None, (count, lo, hi, first_diff(v2, vals[order]))
        assert first_diff(k2, keys[order]) is None, (count, lo, hi)


def _fib_word(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], np.uint8)


def test_five_megabyte_blocks(orc):
    """blocks above 2^22 bytes: the prefix probe picks the key length and the rank array is built through LDS windows -- text, {A,C,G,T},
    random bytes, a period-2 block, two identical halves and a Fibonacci word"""
    from dark_amd import datagen
    rng = np.random.default_rng(3)
    n = 5_000_011
    cases = [datagen.wiki_like(n, 7), datagen.acgt(n, 8), datagen.random_bytes(n, 9), np.frombuffer(b"ab" * (n // 2) + b"a", np.uint8),
             np.concatenate([datagen.wiki_like(n // 2, 10)] * 2), _fib_word(n)]
    # what each case must have gone through: (suffix array call, BWT call)
    routes = [({"isa_windows", "text_round", "inplace_rounds", "pair_chains"}, {"lfirst", "lfirst_deep"}),   # text: long repeats end as pair chains / deep groups
              ({"short_prefix", "narrow_keys", "text_round", "packed_pairs"}, {"short_prefix", "narrow_keys", "packed_pairs"}),  # {A,C,G,T}: the probe shortens the key; 32 key bits + position in one word
              ({"short_prefix", "narrow_keys", "packed_pairs"}, {"short_prefix", "narrow_keys", "packed_pairs"}),                # random bytes (5 MB: code 8 bits + position 23)
              ({"period_round", "big_groups"}, {"period_round"}),                                             # period 2: one round on the tokens of the stretch's end
              ({"isa_windows", "inplace_rounds"}, {"lfirst", "lfirst_deep"}),                                 # two identical halves
              ({"isa_windows", "isa_marked", "general_round", "big_groups"}, {"general_round"})]              # Fibonacci word (repetitive, no period to find): giant groups, doubling
    with dark_amd.Context(n) as c:
        for t, (sa_route, bwt_route) in zip(cases, routes):
            t = np.ascontiguousarray(t)
            want_sa = orc.sa_sais(t)
            got = c.suffix_array(t)
            assert first_diff(got, want_sa) is None, first_diff(got, want_sa)
            assert sa_route <= c.stats()["routes"] and "lfirst" not in c.stats()["routes"], (sa_route, c.stats()["routes"])
            want_bwt, want_origin = orc.bwt_forward(t, want_sa)
            bwt, origin = c.bwt_forward(t)
            assert origin == want_origin and first_diff(bwt, want_bwt) is None
            assert bwt_route <= c.stats()["routes"], (bwt_route, c.stats()["routes"])
    del rng


def test_pathological_inputs(orc):
    """Worst cases of prefix doubling (the reference's SA-IS is linear: README.md:12, src/saca.rs:5-6) at suite size; tools/pathological.py
    runs them at 1e8 bytes (profiles/r03_pathological.json).  Besides parity: the radix passes per round stay bounded -- round 2 keyed
    the global sort by the group's OFFSET in the big list (24 bits for two giant groups) instead of its dense index (1 bit): 142 passes
    on (ab)^2^23 where 83 do."""
    from dark_amd import datagen
    half = datagen.wiki_like(1_500_000, 3)
    a, b = b"a", b"ab"
    while len(b) < (1 << 21):
        a, b = b, b + a
    cases = {"(ab)^2^20": np.frombuffer(b"ab" * (1 << 20), np.uint8), "a^n b": np.concatenate([np.zeros((1 << 21) - 1, np.uint8), np.ones(1, np.uint8)]),
             "fibonacci word": np.frombuffer(b[:1 << 21], np.uint8), "two identical halves": np.concatenate([half, half]),
             "period 1000": np.tile(np.random.default_rng(1).integers(0, 256, 1000, dtype=np.uint8), 2500)}
    with dark_amd.Context(3 << 20) as c:
        for name, t in cases.items():
            t = np.ascontiguousarray(t)
            want = orc.sa_sais(t)
            assert first_diff(c.suffix_array(t), want) is None, name
            st = c.stats()
            assert st["sort_passes"] <= 5 * st["rounds"] + 8, (name, st["sort_passes"], st["rounds"])
            bwt, origin = c.bwt_forward(t)
            wb, wo = orc.bwt_forward(t, want)
            assert origin == wo and first_diff(bwt, wb) is None, name
            assert first_diff(c.bwt_inverse(bwt, origin), t) is None, name


def test_pair_chains_through_groups_of_three_and_four(orc):
    """dk_suffix_array on copies of copies.  Two identical halves of a text that repeats itself inside leave groups of two AND of
    four; three copies leave groups of three; more copies leave groups the chains must walk THROUGH.  Every pair inside a group of at most four
    members is a chain record; the suffix array must equal SA-IS's, and what doubling needed log2(length of the repeat) rounds for is done in a few
    (two identical halves: 24 rounds at 1e8 bytes before round 5, 11 since)."""
    from dark_amd import datagen
    base = datagen.wiki_like(1_200_000, 21)
    inner = base.copy()
    inner[400_000:460_000] = inner[100_000:160_000]       # a 60 KB repeat inside the text: groups of four once the text is doubled
    five = np.concatenate([base[:300_000]] * 5)          # groups of five: no records, later chains have to walk through them
    mixed = np.concatenate([base[:500_000], base[100_000:400_000], base[:500_000], base[200_000:450_000], datagen.wiki_like(1000, 5), base[:500_000]])
    cases = {"two halves": (np.concatenate([base, base]), 12), "two halves with a repeat inside": (np.concatenate([inner, inner]), 12),
             "three copies": (np.concatenate([base[:800_000]] * 3), 13), "four copies": (np.concatenate([base[:600_000]] * 4), 14),
             "five copies": (five, 40), "copies of parts of copies": (mixed, 40),
             "two halves, the second one cut short": (np.concatenate([base, base[:-7]]), 12), "one odd byte in front": (np.concatenate([[7], base, base]).astype(np.uint8), 12)}
    with dark_amd.Context(max(len(t) for t, _ in cases.values())) as c:
        for name, (t, max_rounds) in cases.items():
            t = np.ascontiguousarray(t)
            want = orc.sa_sais(t)
            got = c.suffix_array(t)
            assert first_diff(got, want) is None, (name, first_diff(got, want))
            st = c.stats()
            assert "pair_chains" in st["routes"], (name, st["routes"])
            assert st["rounds"] <= max_rounds, (name, st["rounds"])


def test_word_like_text(orc):
    """Text of a few frequent words: most suffixes sit in big groups after the initial sort, the text rounds are skipped, and the general
    rounds on ranks do the work -- the active suffixes get their head's position from the inverse permutation itself (marked SA entries),
    and members of big groups carry the symbol in front of their suffix through the global sort (tools/zipf_text.py is this at 1e8 bytes)."""
    rng = np.random.default_rng(21)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) + b" " for _ in range(60)]
    picks = np.minimum(rng.zipf(1.3, size=1_400_000) - 1, len(words) - 1)
    t = np.frombuffer(b"".join(words[i] for i in picks)[:6_000_000], np.uint8).copy()
    t[rng.integers(0, len(t), size=len(t) // 90)] = 10
    with dark_amd.Context(len(t)) as c:
        want = orc.sa_sais(t)
        assert first_diff(c.suffix_array(t), want) is None
        # the route this test is named after (a threshold change must not turn it into a duplicate of another test)
        assert {"isa_windows", "isa_marked", "general_round", "big_groups"} <= c.stats()["routes"], c.stats()["routes"]
        print("suffix array:", sorted(c.stats()["routes"]))
        bwt, origin = c.bwt_forward(t)
        print("bwt:", sorted(c.stats()["routes"]))
        wb, wo = orc.bwt_forward(t, want)
        assert origin == wo and first_diff(bwt, wb) is None
        assert first_diff(c.bwt_inverse(bwt, origin), t) is None


def test_known_answers_saca_rs_411(ctx, vectors):
    # /root/reference/src/saca.rs:409-413 `detailed`, through the GPU path
    for v in vectors["reference"]["saca_rs_411_412"]:
        t = v["input"].encode()
        assert list(ctx.suffix_array(t)) == v["sa"]
        bwt, origin = ctx.bwt_forward(t)
        assert bwt.tobytes() == v["bwt"].encode() and origin == v["origin"]
        assert ctx.bwt_inverse(bwt, origin).tobytes() == t


def test_golden_vectors(ctx, vectors, license_bytes):
    for name, data in (("abracababra", b"abracababra"), ("LICENSE", license_bytes)):
        g = vectors["oracle"][name]
        assert list(ctx.suffix_array(data)) == g["sa"]
        bwt, origin = ctx.bwt_forward(data)
        assert bwt.tobytes().hex() == g["bwt_hex"] and origin == g["origin"]
        dc = ctx.dc_encode(bwt)
        assert list(dc["init"]) == g["dc_init"] and list(dc["d"]) == g["dc_d"]
        assert dc["sym"].tobytes().hex() == g["dc_sym_hex"] and dc["rank"].tobytes().hex() == g["dc_rank_hex"]
        assert ctx.block_encode("rawdc", data).hex() == g["rawdc_records_hex"]
        for m in MODELS:
            s = ctx.block_encode(m, data)
            assert s.hex() == g["streams_hex"][m]
            assert ctx.block_decode(m, s, len(data)) == data  # src/block/dc.rs:187-192 roundtrips


def check_all_stages(ctx, orc, t, models=("dark",), naive=False):
    n = len(t)
    want_sa = orc.sa_naive(t) if (naive or n == 1) else orc.sa_sais(t)
    got_sa = ctx.suffix_array(t)
    assert first_diff(got_sa, want_sa) is None, ("sa", n, first_diff(got_sa, want_sa))
    want_bwt, want_origin = orc.bwt_forward(t, want_sa)
    bwt, origin = ctx.bwt_forward(t)
    assert origin == want_origin and first_diff(bwt, want_bwt) is None, ("bwt", n, origin, want_origin)
    back = ctx.bwt_inverse(bwt, origin)
    assert first_diff(back, t) is None, ("ibwt", n, first_diff(back, t))
    want = orc.dc_encode(want_bwt)
    got = ctx.dc_encode(bwt)
    for key in ("init", "d", "sym", "rank"):
        assert first_diff(got[key], want[key]) is None, ("dc." + key, n, first_diff(got[key], want[key]))
    for m in models:
        s = ctx.block_encode(m, t)
        assert s == orc.block_dc_encode_bwt(m, want_bwt, want_origin), ("stream", m, n)
        if not (t == 255).any():
            assert ctx.block_decode(m, s, n) == t.tobytes(), ("decode", m, n)


def test_seeded_small_inputs(ctx, orc):
    for t in seeded_inputs(seed=41, count=50):
        check_all_stages(ctx, orc, t, models=MODELS, naive=len(t) < 600)


def test_structured_inputs(ctx, orc):
    rng = np.random.default_rng(43)
    cases = [
        text_like(rng, 300000),
        rng.choice(np.frombuffer(b"ACGT", np.uint8), size=262144 + 77),          # small-alphabet packing (32 symbols / key)
        rng.integers(0, 256, size=200001, dtype=np.uint8),                         # random bytes incl. 0xFF
        np.frombuffer(b"ab" * 70000, np.uint8),                                    # period 2: log n rounds, all active
        np.zeros(100000, np.uint8),                                                # one symbol
        np.concatenate([np.zeros(5000, np.uint8), np.ones(5000, np.uint8)] * 7),   # long runs
        np.frombuffer(bytes(range(256)) * 300, np.uint8),                          # period 256
        rng.integers(0, 2, size=150000, dtype=np.uint8),                           # binary alphabet (64 symbols / key)
    ]
    seg = text_like(rng, 9000)
    cases.append(np.concatenate([text_like(rng, 50000), seg, text_like(rng, 30000), seg, seg, text_like(rng, 1000)]))
    for t in cases:
        check_all_stages(ctx, orc, np.ascontiguousarray(t), models=("dark",))


def test_megabyte_text(ctx, orc):
    rng = np.random.default_rng(47)
    t = text_like(rng, 4 << 20, vocab=20000)
    check_all_stages(ctx, orc, t, models=("dark", "ybs"))


def test_tile_boundaries(ctx, orc):
    # sizes around the kernels' tile sizes (2048 / 4096 slots) and a BWT whose runs straddle DC tiles
    rng = np.random.default_rng(53)
    for n in (2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 65536 + 1):
        t = rng.integers(0, 3, size=n, dtype=np.uint8)
        check_all_stages(ctx, orc, t)
    for n in (4096, 8192, 20000):
        for bwt in (np.repeat(rng.integers(0, 5, size=n // 512 + 1, dtype=np.uint8), 512)[:n],
                    rng.integers(0, 200, size=n, dtype=np.uint8)):
            bwt = np.ascontiguousarray(bwt)
            want = orc.dc_encode(bwt)
            got = ctx.dc_encode(bwt)
            for key in ("init", "d", "sym", "rank"):
                assert first_diff(got[key], want[key]) is None, ("dc." + key, n, first_diff(got[key], want[key]))


def test_first_call_of_a_fresh_context(orc):
    """A context's very first call, on a tiny block, right after creation -- forty times.  dk_ctx_create used to clear the mailbox with a memset on the
    null stream, which does not order the context's non-blocking stream: the first call's symbol histogram could be wiped after its kernel had written it
    (one symbol "seen" -> the suffix array of a^n), one run in four where a big context had been destroyed just before (round 5)."""
    from dark_amd import datagen
    with dark_amd.Context(5_000_011) as big:
        big.suffix_array(datagen.random_bytes(5_000_011, 9))
    rng = np.random.default_rng(53)
    t = rng.integers(0, 3, size=2047, dtype=np.uint8)
    want = orc.sa_sais(t)
    wb, wo = orc.bwt_forward(t, want)
    for k in range(40):
        with dark_amd.Context(1 << 20) as c:
            if k & 1:
                bwt, origin = c.bwt_forward(t)
                assert origin == wo and first_diff(bwt, wb) is None, k
            else:
                assert first_diff(c.suffix_array(t), want) is None, k


def test_errors_and_limits(ctx):
    with pytest.raises(dark_amd.DarkError) as e:
        ctx.suffix_array(b"")
    assert e.value.code == dark_amd._lib.DK_E_ARG  # n == 0: the reference panics (src/saca.rs:107)
    with pytest.raises(dark_amd.DarkError) as e:
        ctx.suffix_array(np.zeros(CAP + 1, np.uint8))
    assert e.value.code == dark_amd._lib.DK_E_ARG
    with pytest.raises(dark_amd.DarkError) as e:
        ctx.bwt_inverse(b"abc", 3)
    assert e.value.code == dark_amd._lib.DK_E_ARG
    with pytest.raises(dark_amd.DarkError) as e:
        ctx.block_encode("nope", b"abc")
    assert e.value.code == dark_amd._lib.DK_E_MODEL
    s = ctx.block_encode("dark", b"hello\xffworld")  # encodes (bit-exact with the reference) ...
    with pytest.raises(dark_amd.DarkError):           # ... but the format cannot carry symbol 0xFF back
        ctx.block_decode("dark", s, 11)
    assert ctx.block_decode("dark", ctx.block_encode("dark", b"aaaaaaa"), 7) == b"aaaaaaa"  # one-symbol quirk handled
    assert list(ctx.suffix_array(b"z")) == [0]  # n == 1: reference asserts (saca.rs:300); the GPU path just answers


def test_mirror_interfaces(vectors, license_bytes):
    # the reference's own tests, written against the mirrored interface (src/saca.rs:393-407, src/block/dc.rs:176-192)
    from dark_amd import block, saca
    con = saca.Constructor(len(b"banana"))
    assert con.capacity() == 6
    assert list(con.compute(b"banana")) == [5, 3, 1, 0, 4, 2]
    for m, data in (("exp", b"abracababra"), ("exp", license_bytes), ("ybs", license_bytes)):
        enc = block.dc.Encoder(len(data), m)
        stream = enc.encode(data)
        dec = block.dc.Decoder(len(data), enc.model)
        assert dec.decode(stream) == data
    enc = block.raw.Encoder(len(license_bytes), "bbb")    # src/block/raw.rs with the coding model of src/main.rs:73,104
    assert block.raw.Decoder(len(license_bytes), enc.model).decode(enc.encode(license_bytes)) == license_bytes
    dump = block.raw.Encoder(len(license_bytes), "raw")
    assert dump.encode(license_bytes) == b"\0\0\0\0" and len(dump.dumped) == len(license_bytes) + 4


def test_cli_container(tmp_path, orc, license_bytes, monkeypatch):
    # src/main.rs:87-113 / :56-83 through the CLI: FILE -> FILE.dark = [u32 LE n][stream]; FILE.dark -> FILE.orig
    import struct
    from dark_amd import cli
    monkeypatch.chdir(tmp_path)
    src = tmp_path / "book"
    data = license_bytes * 40
    src.write_bytes(data)
    out = cli.encode_file(str(src), "dark", 0, 0)
    blob = open(out, "rb").read()
    assert out == "book.dark" and blob[:4] == struct.pack("<I", len(data))
    assert blob[4:] == orc.block_dc_encode("dark", data)  # single block: byte-identical to the oracle's stream, no footer
    assert open(cli.decode_file("book.dark", "dark", 0), "rb").read() == data
    # names follow PathBuf::set_extension (main.rs:64-66,96-98): the extension is REPLACED
    (tmp_path / "notes.txt").write_bytes(data[:5000])
    assert cli.encode_file(str(tmp_path / "notes.txt"), "exp") == "notes.dark"
    assert cli.decode_file("notes.dark", "exp") == "notes.orig" and open("notes.orig", "rb").read() == data[:5000]
    # multi-block extension (streamed + batched): records are concatenated, each is what a single-block run writes; index footer
    out = cli.encode_file(str(src), "ybs", 10000, 0)
    blob = open(out, "rb").read()
    assert blob[:4] == struct.pack("<I", 10000)
    first = orc.block_dc_encode("ybs", data[:10000])
    assert blob[4:4 + len(first)] == first
    offsets, end = cli.read_footer(out)
    assert len(offsets) == -(-len(data) // 10000) and offsets[1] == 4 + len(first)
    pos = 0
    for k, off in enumerate(offsets):  # every record, not only the first, equals the oracle's stream of that block
        blk = data[10000 * k:10000 * (k + 1)]
        rec = struct.pack("<I", len(blk)) + orc.block_dc_encode("ybs", blk)
        assert off == pos and blob[off:off + len(rec)] == rec
        pos += len(rec)
    assert pos == end
    assert open(cli.decode_file("book.dark", "ybs", 0), "rb").read() == data
    # the same archive without its footer decodes by walking the records (bytes consumed per record)
    open("legacy.dark", "wb").write(blob[:end])
    assert open(cli.decode_file("legacy.dark", "ybs", 0), "rb").read() == data
    # two worker processes (block b -> worker b mod 2; both on GPU 0 here): identical archive, identical round trip
    out2 = cli.encode_file(str(src), "ybs", 10000, 0, gpus=2, devices="0,0")
    assert open(out2, "rb").read() == blob
    assert open(cli.decode_file("book.dark", "ybs", 0, gpus=2, devices="0,0"), "rb").read() == data
    cli.main(["-m", "exp", str(src)])
    cli.main(["-m", "exp", "book.dark"])
    assert open("book.orig", "rb").read() == data
    # -m bbb: block::raw with the coding model (src/main.rs:73,104); single block and walked multi-record form
    out = cli.encode_file(str(src), "bbb")
    assert open(out, "rb").read() == struct.pack("<I", len(data)) + orc.raw_bbb_encode(data)
    assert open(cli.decode_file(out, "bbb"), "rb").read() == data
    out = cli.encode_file(str(src), "bbb", 12345)
    assert open(cli.decode_file(out, "bbb"), "rb").read() == data
    # dump models (src/model/raw.rs): out.raw = origin (4 bytes, big-endian order) + BWT; out-dc.raw = 10-byte records
    cli.encode_file(str(src), "raw", 0, 0)
    bwt, origin = orc.bwt_forward(data)
    assert open("out.raw", "rb").read() == struct.pack(">I", origin) + bwt.tobytes()
    assert open("book.dark", "rb").read() == struct.pack("<I", len(data)) + b"\0\0\0\0"
    cli.encode_file(str(src), "rawdc", 0, 0)
    assert open("out-dc.raw", "rb").read() == orc.block_dc_encode("rawdc", data)


def test_cli_constant_blocks_and_0xff(tmp_path, license_bytes, monkeypatch):
    # a one-symbol block (zero padding in tar / disk images) inside a multi-record archive; and blocks with byte 0xFF
    from dark_amd import cli
    monkeypatch.chdir(tmp_path)
    data = license_bytes * 9 + b"\0" * 8192 + license_bytes * 3 + b"\x07" * 5000
    (tmp_path / "disk.img").write_bytes(data)
    for model in ("dark", "exp", "ybs", "simple"):
        out = cli.encode_file(str(tmp_path / "disk.img"), model, 4096, 0)
        assert out == "disk.dark"
        assert open(cli.decode_file(out, model, 0), "rb").read() == data
        offsets, end = cli.read_footer(out)
        open("walk.dark", "wb").write(open(out, "rb").read()[:end])  # no index: records walked by consumed bytes
        assert open(cli.decode_file("walk.dark", model, 0), "rb").read() == data
    (tmp_path / "bin").write_bytes(license_bytes + b"\xff" + license_bytes)
    with pytest.raises(SystemExit) as e:
        cli.encode_file(str(tmp_path / "bin"), "dark")
    assert "0xFF" in str(e.value)
    with pytest.raises(SystemExit):
        cli.encode_file(str(tmp_path / "bin"), "dark", 700)
    out = cli.encode_file(str(tmp_path / "bin"), "dark", force=True)  # what the reference writes, undecodable by its own format
    with pytest.raises(dark_amd.DarkError):
        cli.decode_file(out, "dark")


def test_cli_refusal_and_failure_leave_no_damage(tmp_path, license_bytes, monkeypatch):
    """the archive is written to a temporary file and renamed only when complete.  A refusal (byte 0xFF, with or without -b) or
    a failure mid-file must leave an existing good archive of the same name untouched, and no partial file behind."""
    import os
    from dark_amd import cli
    monkeypatch.chdir(tmp_path)
    good = license_bytes * 12
    (tmp_path / "data.txt").write_bytes(good)
    out = cli.encode_file(str(tmp_path / "data.txt"), "dark", 5000)
    before = open(out, "rb").read()
    (tmp_path / "data.bin").write_bytes(good + b"\xff" + good)  # same stem -> same archive name; 0xFF sits in block 2 of 3
    for bs in (0, len(good) // 2 + 11):
        with pytest.raises(SystemExit):
            cli.encode_file(str(tmp_path / "data.bin"), "dark", bs)
        assert open(out, "rb").read() == before
        assert sorted(os.listdir(".")) == ["data.bin", "data.dark", "data.txt"], os.listdir(".")  # no temporary or partial file
    assert open(cli.decode_file(out, "dark"), "rb").read() == good


def test_batch_error_path_is_safe(orc):
    """a push that fails (here: a block larger than the context) must not leave coding threads running over buffers that are
    about to be freed: the batch is finished before the error is raised, the context stays usable, and a batch left open is finished by
    Context.close() / dk_ctx_destroy."""
    torch = pytest.importorskip("torch")
    blocks = [np.ascontiguousarray(text_like(np.random.default_rng(5 + i), 150_000)) for i in range(3)]
    big = np.zeros(400_001, np.uint8)
    with dark_amd.Context(400_000) as c:
        b = c.batch_begin("dark", 2)
        for blk in blocks:
            b.push(torch.from_numpy(blk).cuda(), len(blk))
        with pytest.raises(dark_amd.DarkError) as e:  # while a batch is open the context serves nothing else
            c.block_encode("dark", blocks[0])
        assert e.value.code == dark_amd._lib.DK_E_ARG
        with pytest.raises(dark_amd.DarkError):
            b.push(torch.from_numpy(big).cuda(), len(big))
        with pytest.raises(dark_amd.DarkError):       # the failed push closed the batch
            b.finish()
        assert c.block_encode("dark", blocks[1]) == orc.block_dc_encode("dark", blocks[1])
        with c.batch_begin("dark", 2) as b2:           # context-manager form; results through finish()
            for blk in blocks:
                b2.push(torch.from_numpy(blk).cuda(), len(blk))
            streams = b2.finish()
        assert [s.tobytes() for s in streams] == [orc.block_dc_encode("dark", blk) for blk in blocks]
        b3 = c.batch_begin("dark", 2)                  # left open on purpose: Context.close() must join its coders
        b3.push(torch.from_numpy(blocks[2]).cuda(), len(blocks[2]))
    assert b3._h is None


def test_raw_block_codec(ctx, orc, license_bytes):
    # block::raw::{Encoder,Decoder} with the dump model Out (src/block/raw.rs:35-104, src/model/raw.rs:46-76)
    import struct
    dump = ctx.raw_block_encode_dump(license_bytes)
    bwt, origin = orc.bwt_forward(license_bytes)
    assert dump == struct.pack(">I", origin) + bwt.tobytes()
    assert ctx.raw_block_decode(b"\0\0\0\0", 10) == b"\0" * 10  # Out::decode "not supported": symbol 0 for everything
    with pytest.raises(dark_amd.DarkError) as e:
        ctx.raw_block_encode_dump(license_bytes, raw_model=7)
    assert e.value.code == dark_amd._lib.DK_E_MODEL
    # the coding RawModel bbb (src/model/bbb.rs; gates after etc/bbb/main.cpp, PARITY UNPINNED): GPU BWT + host coder == oracle, round trip
    rng = np.random.default_rng(73)
    for t in (np.frombuffer(license_bytes * 5, np.uint8), rng.integers(0, 256, size=30000, dtype=np.uint8), np.zeros(5000, np.uint8),
              text_like(rng, 200000)):
        t = np.ascontiguousarray(t)
        s = ctx.raw_block_encode(t, 1)
        assert s == orc.raw_bbb_encode(t), len(t)
        assert ctx.raw_block_decode(s, len(t), 1) == t.tobytes()   # bytes 0xFF included: block::raw has no header quirk
        assert ctx.last_consumed() == len(s)
    lib = dark_amd.load_library()
    assert ctx.last_block_flags() == 0
    ctx.block_encode("dark", b"abc\xffdef")
    assert ctx.last_block_flags() & dark_amd._lib.DK_FLAG_HAS_FF
    ctx.block_encode("dark", b"zzzzzz")
    assert ctx.last_block_flags() == dark_amd._lib.DK_FLAG_SINGLE_SYMBOL
    del lib


def test_group_size_classes(ctx, orc):
    # groups around the small/big threshold (64 members) at every alignment relative to the 2048-slot tiles of the local
    # round kernel, plus deep repeats that stay unresolved for many doubling rounds
    rng = np.random.default_rng(59)
    parts = []
    for g in list(range(60, 70)) + [2, 3, 127, 128, 129, 300, 2047, 2048, 2049] + [int(x) for x in rng.integers(2, 200, size=60)]:
        motif = rng.integers(0, 250, size=int(rng.integers(12, 40)), dtype=np.uint8)
        for _ in range(g):
            parts.append(motif)
            parts.append(rng.integers(0, 250, size=int(rng.integers(1, 4)), dtype=np.uint8))  # separator: copies diverge after the motif
    deep = rng.integers(0, 4, size=20000, dtype=np.uint8)
    parts += [deep, rng.integers(0, 250, size=100, dtype=np.uint8), deep, rng.integers(0, 250, size=7, dtype=np.uint8), deep]
    order = rng.permutation(len(parts) - 5)
    t = np.concatenate([parts[i] for i in order] + parts[-5:])
    check_all_stages(ctx, orc, np.ascontiguousarray(t), models=("dark",))


def test_sizes_around_2p22(ctx, orc):
    # 2^22 bytes: the prefix probe starts to run, and the inverse permutation's windows grow beyond 1024 words (two levels, wbits 10 -> 11)
    rng = np.random.default_rng(61)
    base = text_like(rng, (1 << 22) + 8, vocab=30000)
    for n in ((1 << 22) - 1, 1 << 22, (1 << 22) + 1):
        t = np.ascontiguousarray(base[:n])
        sa = ctx.suffix_array(t)
        want = orc.sa_sais(t)
        assert first_diff(sa, want) is None, (n, first_diff(sa, want))
        assert "isa_windows" in ctx.stats()["routes"], ctx.stats()["routes"]


def test_batch_encode_matches_single(ctx, orc):
    # dk_dev_batch_encode: device stages pipelined against host coding threads; every stream equals the single-block result
    import torch
    rng = np.random.default_rng(67)
    blocks = [text_like(rng, int(n)) for n in (50000, 120000, 777, 300001, 64000, 9)]
    d_blocks = [torch.from_numpy(b.copy()).cuda() for b in blocks]
    streams = ctx.dev_batch_encode("dark", d_blocks, [len(b) for b in blocks], host_threads=3)
    for b, s in zip(blocks, streams):
        assert s.tobytes() == orc.block_dc_encode("dark", b)
    d_outs = [torch.empty(len(b), dtype=torch.uint8, device="cuda") for b in blocks]
    ctx.dev_batch_decode("dark", streams, [len(b) for b in blocks], d_outs, host_threads=4)
    for b, o in zip(blocks, d_outs):
        assert o.cpu().numpy().tobytes() == b.tobytes()
    streams = ctx.dev_batch_encode("ybs", d_blocks[:2], [len(b) for b in blocks[:2]], host_threads=8)
    assert streams[1].tobytes() == ctx.block_encode("ybs", blocks[1])
    with pytest.raises(dark_amd.DarkError):  # a stream cut short fails its block and the call, it does not hang the pool
        ctx.dev_batch_decode("dark", [streams[0][:40]] + list(streams), [len(blocks[0])] + [len(b) for b in blocks[:2]],
                             [d_outs[0]] + d_outs[:2], host_threads=2)


def test_dc_path_mixtures(ctx, orc):
    """k_dc_main picks its way per tile (wide: more than 20 distinct symbols in the tile's first 64 positions) and per chunk of 64
    (how many lanes are the first / second / a later occurrence of their symbol, how many first occurrences): byte arrays built to
    land on every combination, compared with the oracle's bwt::dc::encode restatement (any byte array is a valid input of the stage)."""
    rng = np.random.default_rng(4711)
    T = 4096

    def tiles(head, body, count=5, tail=1234):
        out = []
        for k in range(count):
            out.append(head(64))
            out.append(body(T - 64))
        out.append(body(tail))
        return np.concatenate(out).astype(np.uint8)

    rnd = lambda a: (lambda m: rng.integers(0, a, m))
    period = lambda p: (lambda m: (np.arange(m) % p) + 7)
    runs = lambda m: np.repeat(rng.integers(0, 9, m // 5 + 1), 5)[:m]
    cases = [
        tiles(rnd(256), period(2)),             # wide tile, then chunks where 60 lanes are a third or later occurrence (bit-matching fallback)
        tiles(rnd(256), period(3)),
        tiles(rnd(256), period(40)),            # wide tile, every chunk: 40 firsts, 24 seconds
        tiles(rnd(256), runs),                  # wide tile, then long runs (chunks without any run start)
        tiles(period(2), rnd(256)),             # narrow tile whose later chunks have ~57 first occurrences (whole-table route)
        tiles(rnd(256), lambda m: np.where(np.arange(m) // 64 % 2 == 0, rng.integers(0, 256, m), 5)),  # random chunks between constant ones
        rng.integers(0, 40, 3 * T + 17),        # 40 symbols: ~14 lanes per chunk beyond the second occurrence, either side of the threshold
        rng.integers(0, 24, 3 * T + 999),       # 24 symbols: wide or not from tile to tile
        rng.integers(0, 90, 2 * T + 63),
        rng.integers(0, 256, 8 * T + 1),        # the bitmap route proper
        np.concatenate([rng.integers(0, 256, 70), np.full(3 * T, 200)]),  # a wide tile whose symbols never come back
        np.concatenate([rng.integers(0, 256, T), rng.integers(0, 4, T), rng.integers(0, 256, T + 5)]),
    ]
    for k, L in enumerate(cases):
        L = np.ascontiguousarray(L, dtype=np.uint8)
        want = orc.dc_encode(L)
        got = ctx.dc_encode(L)
        for key in ("init", "d", "sym", "rank"):
            assert first_diff(got[key], want[key]) is None, (k, "dc." + key, len(L), first_diff(got[key], want[key]))
        out, used = orc.dc_decode(got["init"], got["d"], len(L))
        assert used == len(got["d"]) and first_diff(out, L) is None, (k, "dc round trip")


def test_lfirst_bwt_without_the_suffix_array(orc):
    """dk_bwt_forward on text: only groups of suffixes with different symbols in front are refined, from the text alone (csrc/lfirst.inc);
    the result is the BWT the reference's TransformIterator gives (src/block/dc.rs:45-50).  Cases built for its parts: tile-crossing groups
    (runs of equal 8-grams), groups that take the deep way (copies of long passages, also three and four of them), suffix 0 inside a
    repeat (the origin needs its exact place), text ending in zero bytes (suffixes that end inside a key), blocks around the tile sizes."""
    from dark_amd import datagen
    rng = np.random.default_rng(77)
    base = datagen.wiki_like(3_000_000, 12)
    cases = []
    t = base.copy()
    seg = t[5000:45000].copy()
    for o in (700_000, 1_900_000, 2_800_000):  # four copies of 40 KB
        t[o:o + len(seg)] = seg
    cases.append(("copies of copies", t, {"lfirst", "lfirst_deep"}))
    t = base[:2_000_000].copy()
    t[1_200_000:1_200_000 + 300_000] = t[:300_000]  # suffix 0 starts a 300 KB repeat
    cases.append(("suffix 0 in a repeat", t, {"lfirst", "lfirst_deep"}))
    t = np.concatenate([base[:1_500_000], np.zeros(9, np.uint8)])
    cases.append(("ends in a few zeros", t, {"lfirst"}))
    t = base[:1_000_000].copy()
    t[300_000:300_000 + 5000] = 65  # a run of 5000 equal bytes: one group that never splits on text -- it rides in the big list until the round
    cases.append(("a run inside text", t, {"lfirst", "lfirst_big_round", "period_round"}))  # stalls, then one token round places it by where the run ends
    t = base[:1_000_000].copy()
    for o in range(0, 900_000, 30_000):  # thirty runs of 700 bytes, 2 % of the block: tokens in the first round place them by where they end (round 5;
        t[o + 100:o + 800] = 65          # until then the run probe sent such a block the suffix-array way)
    cases.append(("many runs inside text", t, {"lfirst", "period_round"}))
    t = base[:1_000_000].copy()
    for o in range(0, 900_000, 30_000):  # thirty runs of 2400 bytes, 7 % of the block: more than the L-first path takes (run probe): suffix-array path
        t[o + 100:o + 2500] = 65
    cases.append(("a lot of runs inside text", t, {"general_round"}))
    t = base[:1_000_000].copy()
    for k, o in enumerate(range(0, 900_000, 100_000)):  # nine runs of the same length with the same byte behind them: the tokens tie, the rounds go on
        t[o + 100:o + 500] = 66
        t[o + 500] = 67
    cases.append(("runs of one length inside text", t, {"lfirst", "period_round"}))
    t = base[:1_000_000].copy()
    t[300_033:300_033 + 300] = 65   # a run the probe does not see (it starts one byte behind a 256-byte border): a group of ~290 members that does not
    cases.append(("a short run inside text", t, {"lfirst", "lfirst_big_round", "lfirst_deep"}))  # split on text -- k_lf_deep orders it by common extensions
    t = base[:1_000_000].copy()
    t[500_001:500_001 + 3 * 160] = np.frombuffer(b"xyz" * 160, np.uint8)  # a periodic stretch: the same, period 3
    cases.append(("a periodic stretch inside text", t, {"lfirst"}))
    cases.append(("two identical halves", np.concatenate([base[:1_200_000]] * 2), {"lfirst", "lfirst_giant"}))   # one pair with 1.2 MB in common
    cases.append(("three copies", np.concatenate([base[:800_000]] * 3), {"lfirst", "lfirst_giant"}))            # ... and a triple: two giant rounds
    t = np.concatenate([base[:700_000], base[100_000:700_000], base[:700_000]])                                   # copies inside copies
    cases.append(("copies inside copies", t, {"lfirst", "lfirst_giant"}))
    for n in (65536, 65537, 2048 * 33, 2048 * 33 + 1, 1024 * 1024 + 3):
        cases.append(("n = %d" % n, base[:n].copy(), {"lfirst"}))
    # groups that cross many 2048-slot tiles (k_lf_straddle): 9000 copies of a 40-byte passage, each behind the same byte -- except one copy,
    # the first / a middle / the last in the text (= in its group after the stable initial sort): the one group that starts with the
    # passage's first byte is live because of a single member far from the tiles most of it lies in; the 39 others are not
    seg = rng.integers(97, 123, size=40, dtype=np.uint8)
    for odd in (0, 4500, 8999):
        parts = []
        for k in range(9000):
            parts += [rng.integers(0, 256, size=int(rng.integers(20, 60)), dtype=np.uint8), np.array([66 if k == odd else 65], np.uint8), seg]
        t = np.concatenate(parts + [rng.integers(0, 256, size=100_000, dtype=np.uint8)])
        t[t == 255] = 0
        # (the live group is big and holds nearly everything that is live: the first rerank decides all forty groups across their tiles, then
        # the suffix-array path goes on; tests/test_env_variants.py runs the same shape with the L-first path forced)
        cases.append(("tile-crossing groups, odd copy %d" % odd, t, {"lfirst_fallback"}))
    with dark_amd.Context(max(len(t) for _, t, _ in cases)) as c:
        for name, t, route in cases:
            wb, wo = orc.bwt_forward(t)
            bwt, origin = c.bwt_forward(t)
            assert origin == wo and first_diff(bwt, wb) is None, name
            assert route <= c.stats()["routes"], (name, c.stats()["routes"])
            assert first_diff(c.bwt_inverse(bwt, origin), t) is None, name
            stream = c.block_encode("dark", t)
            assert stream == orc.block_dc_encode("dark", t), name
    del rng


def test_workspace_accounting(orc):
    """a context sized exactly to its block must hold every stage's temporaries (csrc/abi.cpp workspace_bytes lists them) --
    text, word-like text (marked SA entries, big groups), {A,C,G,T}, a period-2 block (pair-chain-free giant groups) and text with long
    repeats (pair chains / deep groups), through the suffix array, the BWT, a block encode and a decode; the margin is printed."""
    from dark_amd import datagen
    n = 3_000_000
    rng = np.random.default_rng(5)
    cases = [datagen.wiki_like(n, 3), datagen.word_like(n, 8), datagen.acgt(n, 4), np.frombuffer(b"ab" * (n // 2), np.uint8),
             np.concatenate([datagen.wiki_like(n // 2, 6)] * 2), rng.integers(0, 255, size=n, dtype=np.uint8)]
    for t in cases:
        t = np.ascontiguousarray(t)
        with dark_amd.Context(len(t)) as c:
            c.suffix_array(t)
            bwt, origin = c.bwt_forward(t)
            stream = c.block_encode("dark", t)
            assert c.block_decode("dark", stream, len(t)) == t.tobytes()
            st = c.stats()
            assert 0 < st["ws_peak_bytes"] <= st["ws_size_bytes"]
            print("n=%d peak %.1f n of %.1f n" % (len(t), st["ws_peak_bytes"] / len(t), st["ws_size_bytes"] / len(t)))


def test_period_round_settles_periodic_blocks_in_one_round(orc):
    """a^n b, (ab)^n, (abc)^n with one odd byte, 32-bit words of one value with one odd word: every suffix lies inside a stretch of one short
    period, and one round on the tokens of the stretch's end (direction, length to the break) settles all of them -- the reference's SA-IS is
    linear on such input (README.md:12), prefix doubling alone needs log2(n) rounds over everything (missing 4)"""
    n = 3_000_001
    abc = np.frombuffer(b"abc" * (n // 3 + 1), np.uint8)[:n].copy()
    abc[n // 2] = ord("b")
    words = np.tile(np.array([0, 0, 1, 7], np.uint8), n // 4 + 1)[:n].copy()
    words[123_456] = 9
    rows = np.tile(np.random.default_rng(5).integers(0, 256, 1000, dtype=np.uint8), n // 1000 + 1)[:n].copy()  # a table of one 1000-byte row: the long-period
    rows[n // 3] ^= 1                                                                                            # search finds it, the token round waits for depth 1000
    cases = {"a^n b": np.concatenate([np.zeros(n - 1, np.uint8), np.ones(1, np.uint8)]), "(ab)^n": np.frombuffer(b"ab" * (n // 2), np.uint8),
             "b^n a": np.concatenate([np.full(n - 1, 98, np.uint8), np.full(1, 97, np.uint8)]), "(abc)^n, one odd byte": abc, "words, one odd": words,
             "rows of 1000 bytes, one odd byte": rows}
    with dark_amd.Context(n) as c:
        for name, t in cases.items():
            t = np.ascontiguousarray(t)
            want = orc.sa_sais(t)
            got = c.suffix_array(t)
            assert first_diff(got, want) is None, (name, first_diff(got, want))
            st = c.stats()
            assert "period_round" in st["routes"] and st["rounds"] <= (12 if name.startswith("rows") else 3), (name, st["rounds"], st["routes"])
            wb, wo = orc.bwt_forward(t, want)
            bwt, origin = c.bwt_forward(t)
            assert origin == wo and first_diff(bwt, wb) is None, name
            assert "period_round" in c.stats()["routes"], (name, c.stats()["routes"])
