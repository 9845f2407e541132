// The reference's own unit tests, written against the C++ mirror (include/dark.hpp) of its Rust interface.
//   saca::test::detailed / roundtrips        /root/reference/src/saca.rs:393-433
//   block::dc::test::roundtrips              /root/reference/src/block/dc.rs:176-192
// Built and run by tests/test_gpu_cpp_mirror.py on the GPU box.  argv[1] = path of the LICENSE fixture.
#include <cassert>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "dark.hpp"

using Bytes = std::vector<uint8_t>;
static Bytes bytes(const char *s) { Bytes b; for (; *s; ++s) b.push_back(static_cast<uint8_t>(*s)); return b; }
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static int some_detail(const Bytes &input, const std::vector<uint32_t> &suf_expected, size_t origin_expected, const Bytes &out_expected) {
    dark::saca::Constructor con(input.size());
    const auto &suf = con.compute(input);
    CHECK(suf == suf_expected);
    auto [output, origin] = dark::bwt::transform(con.context(), input);
    CHECK(origin == origin_expected);
    CHECK(output == out_expected);
    CHECK(dark::bwt::decode(con.context(), output, origin) == input);
    return 0;
}

static int some_roundtrip(const Bytes &input) {
    dark::saca::Constructor con(input.size());
    con.compute(input);
    auto [output, origin] = dark::bwt::transform(con.context(), input);
    CHECK(dark::bwt::decode(con.context(), output, origin) == input);
    return 0;
}

template <class M>
static int roundtrip(M model, const Bytes &data) {
    dark::block::dc::Encoder<M> enc(data.size(), model);
    auto [writer, err] = enc.encode(data, Bytes());
    err.unwrap();
    dark::block::dc::Decoder<M> dec(data.size(), enc.model);
    auto [reader, output, err2] = dec.decode(writer, Bytes());
    (void)reader;
    err2.unwrap();
    CHECK(output == data);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    Bytes text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    CHECK(text.size() == 1083);
    // saca::test::detailed (src/saca.rs:409-413)
    if (some_detail(bytes("abracadabra"), {10, 7, 0, 3, 5, 8, 1, 4, 6, 9, 2}, 2, bytes("rdarcaaaabb"))) return 1;
    if (some_detail(bytes("banana"), {5, 3, 1, 0, 4, 2}, 3, bytes("nnbaaa"))) return 1;
    // saca::test::roundtrips (src/saca.rs:429-433)
    if (some_roundtrip(text)) return 1;
    // block::dc::test::roundtrips (src/block/dc.rs:187-192)
    if (roundtrip(dark::model::exp::Model(), bytes("abracababra"))) return 1;
    if (roundtrip(dark::model::exp::Model(), text)) return 1;
    if (roundtrip(dark::model::ybs::Model(), text)) return 1;
    if (roundtrip(dark::model::dark::Model(), text)) return 1;
    if (roundtrip(dark::model::simple::Model(), text)) return 1;
    // block::raw with the coding model bbb (src/block/raw.rs:35-104; the reference has no unit test for it, its Makefile packs with -m bbb)
    {
        dark::block::raw::Encoder<dark::model::bbb::Model> enc(text.size(), dark::model::bbb::Model());
        auto [writer, err] = enc.encode(text, Bytes());
        err.unwrap();
        dark::block::raw::Decoder<dark::model::bbb::Model> dec(text.size(), enc.model);
        auto [reader, output, err2] = dec.decode(writer, Bytes());
        (void)reader;
        err2.unwrap();
        CHECK(output == text);
        CHECK(writer.size() < text.size());
        // the dump model: origin (4 bytes) + L land in the model, the coder writes its 4-byte tail only
        dark::block::raw::Encoder<dark::model::raw::Out> dump(text.size(), dark::model::raw::Out());
        auto [w2, e2] = dump.encode(text, Bytes());
        e2.unwrap();
        CHECK(w2.size() == 4 && dump.model.dumped.size() == text.size() + 4);
    }
    // Constructor::compute asserts the exact size (src/saca.rs:369)
    try { dark::saca::Constructor c(5); c.compute(bytes("banana")); return 1; } catch (const dark::Error &) {}
    std::puts("cpp mirror tests ok");
    return 0;
}
