// Robustness of the host decoders against corrupt input, meant to be built with -fsanitize=address,undefined (CPU only):
//   clang++ -O1 -g -fsanitize=address,undefined -std=c++17 -march=x86-64-v3 -Iinclude -o /tmp/ent_fuzz tools/ent_fuzz.cpp dark_amd/csrc/entropy.cpp -lpthread
// Encodes a sample with every model, checks the round trip, then decodes thousands of mutated / truncated streams and random distance
// arrays: every call must return (an error code or garbage bytes), never read or write out of bounds, never hang.
#include "../dark_amd/csrc/entropy.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace dk;

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937_64 rng(12345);
    // a synthetic BWT-like byte string: runs over a small alphabet (never 0xFF)
    const size_t n = 60000;
    std::vector<uint8_t> bwt(n);
    for (size_t i = 0; i < n;) {
        const uint8_t sym = static_cast<uint8_t>(rng() % 40);
        size_t len = 1 + rng() % 6;
        while (len-- && i < n) bwt[i++] = sym;
    }
    // distance coding on the host: take it from the decoder's own inverse -- build (init, dist) by brute force
    std::vector<uint32_t> dist, init(256, static_cast<uint32_t>(n));
    std::vector<uint8_t> sym;
    {
        // straightforward restatement of bwt::dc::encode for the fuzzer's own use (O(n * sigma), small n)
        std::vector<long> last(256, -1);
        std::vector<uint32_t> sparse(n, static_cast<uint32_t>(n));
        for (size_t i = 0; i < n; ++i) {
            const uint8_t c = bwt[i];
            if (i > 0 && bwt[i - 1] == c) { last[c] = static_cast<long>(i); continue; }
            if (last[c] < 0) init[c] = static_cast<uint32_t>(i);
            else {
                unsigned rank = 0;
                for (int o = 0; o < 256; ++o) rank += (o != c && last[o] > last[c]);
                sparse[static_cast<size_t>(last[c])] = static_cast<uint32_t>(i - static_cast<size_t>(last[c]) - rank - 1);
            }
            last[c] = static_cast<long>(i);
        }
        for (int c = 0; c < 256; ++c)
            if (last[c] >= 0) {
                unsigned rank = 0;
                for (int o = 0; o < 256; ++o) rank += (o != c && last[o] > last[c]);
                sparse[static_cast<size_t>(last[c])] = static_cast<uint32_t>(n - static_cast<size_t>(last[c]) - rank - 1);
            }
        for (size_t i = 0; i < n; ++i)
            if (sparse[i] != n) { dist.push_back(sparse[i]); sym.push_back(bwt[i]); }
    }
    DcStream st;
    st.n = n; st.init = init.data(); st.dist = dist.data(); st.sym = sym.data(); st.m = dist.size(); st.origin = 777;
    size_t failures = 0, decoded_ok = 0, errors = 0;
    for (int model = 0; model < 4; ++model) {
        std::vector<uint8_t> out(8 * dist.size() + 8192);
        size_t len = 0;
        int rc = encode_block_stream(model, st, out.data(), out.size(), &len, 1);
        if (rc) { printf("model %d: encode rc=%d\n", model, rc); ++failures; continue; }
        std::vector<uint8_t> back(n);
        uint32_t origin = 0; int single = 0; size_t consumed = 0;
        rc = decode_block_stream(model, out.data(), len, n, back.data(), &origin, &single, &consumed);
        if (rc || origin != 777 || back != bwt || consumed != len) { printf("model %d: round trip failed rc=%d\n", model, rc); ++failures; }
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> bad(out.begin(), out.begin() + static_cast<long>(len));
            const int kind = static_cast<int>(rng() % 4);
            if (kind == 0) bad.resize(rng() % (len + 1));                                   // truncation
            else if (kind == 1) for (int k = 0; k < 1 + static_cast<int>(rng() % 4); ++k) bad[rng() % len] ^= static_cast<uint8_t>(1u << (rng() % 8));
            else if (kind == 2) for (size_t k = rng() % len; k < len && k < len; k += 1 + rng() % 97) bad[k] = static_cast<uint8_t>(rng());
            else { bad.resize(len + rng() % 64); for (size_t k = len / 2; k < bad.size(); ++k) bad[k] = static_cast<uint8_t>(rng()); }
            const size_t claim_n = (rng() % 8 == 0) ? 1 + rng() % (2 * n) : n;  // the header's n may lie too
            std::vector<uint8_t> o2(claim_n);
            rc = decode_block_stream(model, bad.data(), bad.size(), claim_n, o2.data(), &origin, &single, &consumed);
            if (rc) ++errors; else ++decoded_ok;
        }
    }
    // random distance arrays through dc::decode
    for (int r = 0; r < rounds; ++r) {
        std::vector<uint32_t> d(1 + rng() % 5000), in(256);
        const size_t nn = 1 + rng() % 20000;
        for (auto &v : d) v = static_cast<uint32_t>(rng() % (rng() % 2 ? 50 : nn + 5));
        for (auto &v : in) v = (rng() % 4) ? static_cast<uint32_t>(nn) : static_cast<uint32_t>(rng() % (nn + 3));
        std::vector<uint8_t> o(nn);
        size_t used = 0;
        const int rc = dc_decode_array(in.data(), d.data(), d.size(), o.data(), nn, &used);
        if (rc) ++errors; else ++decoded_ok;
    }
    printf("failures %zu, corrupt inputs: %zu rejected, %zu decoded to something\n", failures, errors, decoded_ok);
    return failures ? 1 : 0;
}
