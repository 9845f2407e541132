// every thread form of the dark model's host coder gives the same bytes as one thread, on random distance streams of several flavours
#include "../dark_amd/csrc/entropy.hpp"
// (a form whose threads the host cannot give -- no last-level-cache group with that many usable cores -- falls back to a narrower one inside
// encode_block_stream: that is reported per form, "ran" / "not available here", so a log shows which forms were really compared)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace dk;
int main(int argc, char **argv) {
    std::mt19937_64 rng(7);
    int bad = 0;
    const int cases = argc > 1 ? atoi(argv[1]) : 24;
    int ran[6] = {0, 0, 0, 0, 0, 0}, unavailable[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < cases; ++c) {
        const size_t m = 50000 + rng() % 400000, n = 1u << 27;
        std::vector<uint32_t> d(m); std::vector<uint8_t> s(m); uint32_t init[256];
        const int flavour = c % 4;
        for (size_t k = 0; k < m; ++k) {
            const unsigned sh = flavour == 0 ? rng() % 27 : flavour == 1 ? 20 + rng() % 7 : flavour == 2 ? rng() % 4 : (rng() % 10 ? 26 : rng() % 27);
            d[k] = static_cast<uint32_t>((rng() % (1u << 27)) >> sh);  // flavour 1: tiny, 2: huge (long unary extensions), 3: mostly zero with outliers
            s[k] = static_cast<uint8_t>(flavour == 3 ? rng() % 3 : rng() % 200);
        }
        for (int i = 0; i < 256; ++i) init[i] = i < 200 ? i : static_cast<uint32_t>(n);
        DcStream st; st.n = n; st.init = init; st.dist = d.data(); st.sym = s.data(); st.m = m; st.origin = 12345;
        std::vector<uint8_t> ref(8 * m + 8192), out(8 * m + 8192); size_t rl = 0, ol = 0;
        int rc = encode_block_stream(0, st, ref.data(), ref.size(), &rl, 1);
        if (rc) { printf("case %d: one thread rc=%d\n", c, rc); ++bad; continue; }
        for (int mode : {2, 4, 5}) {
            rc = encode_block_stream(0, st, out.data(), out.size(), &ol, mode);
            const bool same = rc == 0 && ol == rl && std::equal(ref.begin(), ref.begin() + rl, out.begin());
            if (!same) { printf("case %d mode %d: rc=%d len %zu vs %zu DIFFERENT\n", c, mode, rc, ol, rl); ++bad; }
            if (last_entropy_threads() == mode) ++ran[mode];
            else if (host_l3_groups(mode) > 0) { printf("case %d mode %d: the host has a group for it, but %d threads coded\n", c, mode, last_entropy_threads()); ++bad; }
            else ++unavailable[mode];
        }
        printf("case %d flavour %d m=%zu len=%zu threads of the last form %d\n", c, flavour, m, rl, last_entropy_threads());
    }
    for (int mode : {2, 4, 5}) printf("form with %d threads: ran in %d cases, not available here in %d\n", mode, ran[mode], unavailable[mode]);
    printf("bad: %d\n", bad);
    return bad != 0;
}
