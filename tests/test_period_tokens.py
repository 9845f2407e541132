"""The period round's token (dark_amd/csrc/suffix_array.hip: k_period_fill, k_round_local's `second()`, k_lf_tokens), restated in Python and checked
against the suffix order by brute force -- the model the kernels were written from.

next_break[i] = min { j >= i : j + p >= n or T[j] != T[j+p] };  e = next_break[x] - x = LCE(x, x + p) (cut at the text's end).  For the members of a
group -- suffixes equal in their first h >= p symbols -- that follow period p that far (e >= h - p, a test on those h symbols alone, so all members of
a group agree on it), and that are at least h long:
    token(x) = (0, e, bytes from the break on)            if the break goes DOWN: the text ends at x + e + p, or T[x+e+p] < T[x+e]
               (1, MAX - e, bytes from the break on)      if it goes UP
orders them like the suffixes themselves; a member shorter than h keeps the text key (nothing behind h: zero) and sorts first.  Equal tokens mean
"equal up to the break and in the bytes compared behind it", never a wrong order -- the round only makes groups finer.  The GPU tests run the kernels
(tests/test_gpu_parity.py::test_period_round_settles_periodic_blocks_in_one_round, tools/period_check.py through tests/test_env_variants.py)."""
import random

MAXE = (1 << 31) - 1
TAIL = 4  # bytes of text behind the break that the small groups' 64-bit token carries


def next_break(t, p):
    n = len(t)
    nb = [0] * n
    nxt = n  # (never read: position n - p, or the last position, is always a break)
    for j in range(n - 1, -1, -1):
        if j + p >= n or t[j] != t[j + p]:
            nxt = j
        nb[j] = nxt
    return nb


def token(t, nb, p, x):
    n = len(t)
    b = nb[x]
    e = b - x
    brk = b + p
    down = brk >= n or t[brk] < t[b]
    tail = tuple(t[brk + k] if brk + k < n else 0 for k in range(TAIL))
    return (0, e, tail) if down else (1, MAXE - e, tail)


def check(t, p, h):
    """every group of suffixes equal in h symbols: periodic members by token, the others by the next bytes -- never a contradiction with the suffix order"""
    n = len(t)
    nb = next_break(t, p)
    groups = {}
    for x in range(n):
        groups.setdefault(bytes(t[x:x + h]) + b"\x00" * max(0, x + h - n), []).append(x)  # (zero padding merges "ends here" with "goes on with zeros", like the keys)
    settled = 0
    for key, members in groups.items():
        if len(members) < 2:
            continue
        keyed = []
        for x in members:
            if x + h <= n and nb[x] - x + p >= h:
                k = (1,) + token(t, nb, p, x)          # a periodic member of full length
            elif x + h > n:
                k = (0,)                               # shorter than h: the text key is zero, it sorts first
            else:
                k = None
            keyed.append((x, k))
        kinds = {k is None for _, k in keyed if k != (0,)}
        assert len(kinds) <= 1, "members of one group disagree on being periodic"
        if True in kinds:
            continue  # not a periodic group: the round sorts it by text
        for i in range(len(keyed)):
            for j in range(i + 1, len(keyed)):
                (x, kx), (y, ky) = keyed[i], keyed[j]
                if kx == ky:
                    continue  # a tie: stays a group
                want = bytes(t[x:]) < bytes(t[y:])  # (a proper prefix is smaller: Python's order on bytes)
                assert (kx < ky) == want, (p, h, x, y, kx, ky)
                settled += 1
    return settled


def test_tokens_order_periodic_groups_like_their_suffixes():
    rng = random.Random(5)
    settled = 0
    for case in range(300):
        p = rng.randint(1, 6)
        sigma = rng.randint(2, 4)
        parts = []
        for _ in range(rng.randint(1, 6)):
            u = [rng.randrange(sigma) for _ in range(p)]
            ln = rng.choice([3, 8, 20, 60])
            phase = rng.randrange(p)
            parts += (u * (ln // p + 2))[phase:phase + ln]
            parts += [rng.randrange(sigma) for _ in range(rng.randint(0, 5))]
        t = parts[:rng.randint(max(8, len(parts) // 2), len(parts))] if len(parts) > 8 else parts
        h = rng.randint(p, p + 6)
        settled += check(t, p, h)
    assert settled > 20000  # (the cases do exercise the comparison)


def test_tokens_at_the_ends_of_the_alphabet_and_of_the_text():
    assert check([0] * 40 + [1], 1, 3) > 0           # a^n b: every suffix inside the run, one round
    assert check([1] * 40 + [0], 1, 3) > 0           # b^n a
    assert check([0, 255] * 30, 2, 4) > 0            # (ab)^n with the extreme byte values, ending inside the stretch
    assert check([7] * 50, 1, 5) > 0                 # one symbol: shorter is smaller
    assert check(([1, 2, 3] * 20) + [1, 2, 4] + ([1, 2, 3] * 20), 3, 6) > 0   # two stretches of one period string, different ends
