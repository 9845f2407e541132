"""Seeded synthetic blocks for the BASELINE.json configs: the real corpora (book1, enwik8, enwik9)
are not in the image and there is no network, so every config has a deterministic stand-in of the same size and flavour."""
import numpy as np

ALPHA = 96  # printable bytes 32..127


def _markov(n, seed, order, chains):
    """order-k Markov text over 96 printable bytes, generated as `chains` independent chains advanced in lock-step
    (vectorised over chains; every chain is a contiguous stretch of the output)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nctx = ALPHA ** order
    cand = rng.integers(0, ALPHA, size=(nctx, 4), dtype=np.uint8)  # four candidate successors per context
    # make common contexts point at space / 'e' / 't' a bit more often: word-like structure
    cand[rng.integers(0, nctx, size=nctx // 5), 0] = 0
    steps = -(-n // chains)
    state = rng.integers(0, nctx, size=chains, dtype=np.int64)
    out = np.empty((steps, chains), dtype=np.uint8)
    thresholds = np.array([141, 205, 238], dtype=np.uint8)  # P = .55 .25 .13 .07
    for s in range(steps):
        u = rng.integers(0, 256, size=chains, dtype=np.uint8)
        choice = (u > thresholds[0]).astype(np.int64) + (u > thresholds[1]) + (u > thresholds[2])
        sym = cand[state, choice]
        out[s] = sym
        state = (state * ALPHA + sym) % nctx
    text = np.ascontiguousarray(out.T).reshape(-1)[:n]
    text += 32
    return text, rng


def wiki_like(n, seed=2):
    """config 2 / 4 stand-in for enwik8 / enwik9 blocks: order-3 Markov over 96 printable bytes, 5 % of the length
    overwritten by repeated 256 B - 8 KiB segments (forces deep LCPs like wiki boilerplate)."""
    chains = int(min(65536, max(1, n // 512)))
    text, rng = _markov(n, seed, 3, chains)
    budget = n // 20
    while budget > 0 and n > 16384:
        ln = int(rng.integers(256, 8193))
        src = int(rng.integers(0, n - ln))
        dst = int(rng.integers(0, n - ln))
        text[dst:dst + ln] = text[src:src + ln].copy()
        budget -= ln
    return text


def english_like(n=768771, seed=1):
    """config 1 stand-in for Calgary book1: order-2 Markov English-like text"""
    text, _ = _markov(n, seed, 2, int(min(4096, max(1, n // 256))))
    return text


def acgt(n=1 << 28, seed=3):
    """config 3: i.i.d. uniform over {A,C,G,T}"""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)]


def random_bytes(n=1 << 30, seed=50):
    """config 5: i.i.d. uniform bytes (contains 0xFF: encodes, but the reference format cannot decode it)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=n, dtype=np.uint8)


def word_like(n=100_000_000, seed=5, vocab=200_000, alpha=180):
    """word-like text (what real enwik8 does to a suffix sorter and the Markov stand-in does not): words of a `vocab`-word
    vocabulary (2..10 letters each) drawn with Zipf(1.15) frequencies and separated by spaces, a line end every 80 bytes on average, 3 % of
    the length overwritten by copied 64 B - 4 KiB segments.  alpha = number of distinct letters: 180 gives 182 distinct bytes, so symbol codes
    take 8 bits like on real enwik8 (seven symbols in the initial sort key); 26 gives a 5-bit alphabet.  No byte 0xFF."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 11, size=vocab)
    letters = (rng.integers(0, alpha, size=int(lens.sum())) + (97 if alpha <= 26 else 48)).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)])
    nwords = max(n // 5, 16)
    ranks = np.minimum(rng.zipf(1.15, size=nwords) - 1, vocab - 1)
    wl = lens[ranks] + 1
    total = int(wl.sum())
    if total < n:
        raise ValueError("word_like: %d words make %d bytes, fewer than n = %d" % (nwords, total, n))
    out = np.full(total, 32, dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    for L in range(2, 11):  # fill the letters of all words of one length at a time
        sel = np.nonzero(lens[ranks] == L)[0]
        if len(sel) == 0:
            continue
        src = offs[ranks[sel]][:, None] + np.arange(L)[None, :]
        dst = starts[sel][:, None] + np.arange(L)[None, :]
        out[dst.reshape(-1)] = letters[src.reshape(-1)]
    out = np.ascontiguousarray(out[:n])
    if n >= 80:
        out[rng.integers(0, n, size=n // 80)] = 10
    budget = n // 33
    while budget > 0 and n > 8192:
        ln = int(rng.integers(64, 4097))
        s = int(rng.integers(0, n - ln))
        d = int(rng.integers(0, n - ln))
        out[d:d + ln] = out[s:s + ln].copy()
        budget -= ln
    return out


def real_text(n=50_000_000, seed=0, exts=(".py", ".pyi", ".txt", ".rst", ".md", ".h", ".hpp")):
    """REAL text, the only kind the image holds (no book1 / enwik8, no network): the text files under the interpreter's library directories
    and the ROCm headers, walked in sorted order and concatenated to n bytes -- words, indentation runs, licence boilerplate repeated hundreds of
    times ("no real byte has been through the path").  Files with a byte 0xFF are left out (the reference format cannot
    decode such a block, src/block/dc.rs:57-73).  The same image on the build container and on the GPU box gives the same bytes; `seed` rotates
    the list of files.  Raises when the directories hold fewer than n bytes."""
    import os
    roots = [os.path.dirname(os.__file__), "/usr/local/lib/python3.10/dist-packages", "/usr/lib/python3/dist-packages", "/opt/rocm/include"]
    paths = []
    for root in roots:
        for dirpath, dirnames, files in os.walk(root):
            dirnames.sort()
            paths += [os.path.join(dirpath, f) for f in sorted(files) if f.endswith(tuple(exts))]
    if seed:
        k = (seed * 7919) % max(1, len(paths))
        paths = paths[k:] + paths[:k]
    out, total = [], 0
    for p in paths:
        try:
            with open(p, "rb") as fh:
                b = fh.read()
        except OSError:
            continue
        if not b or b"\xff" in b:
            continue
        out.append(b)
        total += len(b)
        if total >= n:
            break
    if total < n:
        raise ValueError("real_text: the image holds %d bytes of text files, fewer than n = %d" % (total, n))
    return np.frombuffer(b"".join(out)[:n], dtype=np.uint8).copy()


WORKLOADS = {
    "book1_like_768771": lambda seed=1: english_like(768771, seed),
    "enwik8_like_1e8": lambda seed=2: wiki_like(100_000_000, seed),
    "acgt_2p28": lambda seed=3: acgt(1 << 28, seed),
    "enwik9_block_125e6": lambda seed=40: wiki_like(125_000_000, seed),
    "random_2p30": lambda seed=50: random_bytes(1 << 30, seed),
    "wordlike_1e8": lambda seed=5: word_like(100_000_000, seed),
    "realtext_5e7": lambda seed=0: real_text(50_000_000, seed),
}
