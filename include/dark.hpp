// dark.hpp -- C++ mirror of the reference's Rust interface for the hot path, a thin layer over the C ABI (dark_amd.h).
// The reference is compiled code (Rust); its toolchain is not in this image, so the host side a Rust caller would write is
// given here in C++ with the same names, argument meaning and error behaviour:
//   dark::saca::Constructor            src/saca.rs:344-384   new(max_n) / capacity() / compute(input)
//   dark::block::dc::Encoder<Model>    src/block/dc.rs:21-92  new(n, model) / encode(input, writer) -> (writer, result)
//   dark::block::dc::Decoder<Model>    src/block/dc.rs:96-161 new(n, model) / decode(reader, writer) -> (reader, writer, result)
//   dark::block::raw::Encoder<Model> / Decoder<Model>   src/block/raw.rs:17-105 (Model = model::bbb::Model or model::raw::Out)
//   dark::model::{dark,exp,ybs,simple,bbb}::Model, dark::model::raw::Out   src/model/*.rs  (the state lives in the library; the type selects it)
// Where the Rust code panics (assert!/unwrap) this throws dark::Error; io::Result becomes dark::Result{ok, message}.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "dark_amd.h"

namespace dark {

struct Error : std::runtime_error { int code; Error(int c, const std::string &m) : std::runtime_error(m), code(c) {} };
struct Result { bool ok = true; std::string message; explicit operator bool() const { return ok; } void unwrap() const { if (!ok) throw Error(DK_E_STREAM, message); } };

namespace detail {
class Ctx {
public:
    Ctx(size_t max_n, int device) {
        int rc = dk_ctx_create(device, max_n, &h_);
        if (rc != DK_OK) throw Error(rc, "dk_ctx_create failed (no GPU, or out of memory)");
    }
    ~Ctx() { dk_ctx_destroy(h_); }
    Ctx(const Ctx &) = delete;
    Ctx &operator=(const Ctx &) = delete;
    dk_ctx *get() const { return h_; }
    std::string error() const { return dk_last_error(h_); }
private:
    dk_ctx *h_ = nullptr;
};
}  // namespace detail

namespace model {
namespace dark { struct Model { static constexpr int ID = DK_MODEL_DARK; void reset() {} }; }
namespace exp { struct Model { static constexpr int ID = DK_MODEL_EXP; void reset() {} }; }
namespace ybs { struct Model { static constexpr int ID = DK_MODEL_YBS; void reset() {} }; }
namespace simple { struct Model { static constexpr int ID = DK_MODEL_SIMPLE; void reset() {} }; }
namespace bbb { struct Model { static constexpr int RAW_ID = DK_RAWMODEL_BBB; void reset() {} }; }   // src/model/bbb.rs
namespace raw { struct Out { static constexpr int RAW_ID = DK_RAWMODEL_OUT; std::vector<uint8_t> dumped; void reset() {} }; }  // src/model/raw.rs:46-76: "out.raw"
}  // namespace model

namespace saca {
using Symbol = uint8_t;   // src/saca.rs:18
using Suffix = uint32_t;  // src/saca.rs:20
class Constructor {
public:
    explicit Constructor(size_t max_n, int device = 0) : ctx_(max_n, device), suffixes_(max_n), n_(max_n) {}  // Constructor::new
    size_t capacity() const { return dk_capacity(ctx_.get()); }
    // Constructor::compute: asserts input.len() == n (src/saca.rs:369)
    const std::vector<Suffix> &compute(const std::vector<Symbol> &input) {
        if (input.size() != n_) throw Error(DK_E_ARG, "assertion failed: input.len() == self.n");
        int rc = dk_suffix_array(ctx_.get(), input.data(), input.size(), suffixes_.data());
        if (rc != DK_OK) throw Error(rc, ctx_.error());
        return suffixes_;
    }
    detail::Ctx &context() { return ctx_; }  // plays reuse(): later stages share the device workspace through it
private:
    detail::Ctx ctx_;
    std::vector<Suffix> suffixes_;
    size_t n_;
};
}  // namespace saca

namespace block {
namespace dc {
template <class M>
class Encoder {
public:
    M model;  // public like the reference's field (src/block/dc.rs:25)
    Encoder(size_t n, M m, int device = 0) : model(m), ctx_(n, device) { model.reset(); }
    // encode(&input, writer) -> (writer, io::Result<()>); the writer here is any byte container with insert()
    template <class W>
    std::pair<W, Result> encode(const std::vector<uint8_t> &input, W writer) {
        if (input.size() > dk_capacity(ctx_.get())) throw Error(DK_E_ARG, "assertion failed: block_size <= self.sac.capacity()");
        std::vector<uint8_t> out(2 * input.size() + 4096);
        size_t len = 0;
        int rc = dk_block_encode(ctx_.get(), M::ID, input.data(), input.size(), out.data(), out.size(), &len);
        if (rc != DK_OK) return {std::move(writer), Result{false, ctx_.error()}};
        writer.insert(writer.end(), out.begin(), out.begin() + static_cast<std::ptrdiff_t>(len));
        return {std::move(writer), Result{}};
    }
private:
    detail::Ctx ctx_;
};

template <class M>
class Decoder {
public:
    M model;
    Decoder(size_t n, M m, int device = 0) : model(m), ctx_(n, device), n_(n) { model.reset(); }
    // decode(reader, writer) -> (reader, writer, io::Result<()>)
    template <class W>
    std::tuple<std::vector<uint8_t>, W, Result> decode(std::vector<uint8_t> reader, W writer) {
        std::vector<uint8_t> out(n_);
        int rc = dk_block_decode(ctx_.get(), M::ID, reader.data(), reader.size(), n_, out.data());
        if (rc != DK_OK) return {std::move(reader), std::move(writer), Result{false, ctx_.error()}};
        writer.insert(writer.end(), out.begin(), out.end());
        return {std::move(reader), std::move(writer), Result{}};
    }
private:
    detail::Ctx ctx_;
    size_t n_;
};
}  // namespace dc

namespace raw {  // src/block/raw.rs
template <class M>
class Encoder {
public:
    M model;
    Encoder(size_t n, M m, int device = 0) : model(m), ctx_(n, device) { model.reset(); }
    template <class W>
    std::pair<W, Result> encode(const std::vector<uint8_t> &input, W writer) {
        if (input.size() > dk_capacity(ctx_.get())) throw Error(DK_E_ARG, "assertion failed: block_size <= self.sac.capacity()");
        std::vector<uint8_t> out(2 * input.size() + 4096), dump(input.size() + 4);
        size_t len = 0, dumped = 0;
        int rc = dk_raw_block_encode(ctx_.get(), M::RAW_ID, input.data(), input.size(), out.data(), out.size(), &len, dump.data(), dump.size(), &dumped);
        if (rc != DK_OK) return {std::move(writer), Result{false, ctx_.error()}};
        keep_dump(model, dump, dumped);
        writer.insert(writer.end(), out.begin(), out.begin() + static_cast<std::ptrdiff_t>(len));
        return {std::move(writer), Result{}};
    }
private:
    static void keep_dump(model::raw::Out &m, const std::vector<uint8_t> &d, size_t k) { m.dumped.assign(d.begin(), d.begin() + static_cast<std::ptrdiff_t>(k)); }
    template <class Other> static void keep_dump(Other &, const std::vector<uint8_t> &, size_t) {}
    detail::Ctx ctx_;
};
template <class M>
class Decoder {
public:
    M model;
    Decoder(size_t n, M m, int device = 0) : model(m), ctx_(n, device), n_(n) { model.reset(); }
    template <class W>
    std::tuple<std::vector<uint8_t>, W, Result> decode(std::vector<uint8_t> reader, W writer) {
        std::vector<uint8_t> out(n_);
        int rc = dk_raw_block_decode(ctx_.get(), M::RAW_ID, reader.data(), reader.size(), n_, out.data());
        if (rc != DK_OK) return {std::move(reader), std::move(writer), Result{false, ctx_.error()}};
        writer.insert(writer.end(), out.begin(), out.end());
        return {std::move(reader), std::move(writer), Result{}};
    }
private:
    detail::Ctx ctx_;
    size_t n_;
};
}  // namespace raw
}  // namespace block

namespace bwt {  // the pieces of crate `compress` the reference's tests call next to saca (src/saca.rs:398-405)
inline std::pair<std::vector<uint8_t>, size_t> transform(detail::Ctx &ctx, const std::vector<uint8_t> &input) {
    std::vector<uint8_t> out(input.size());
    uint32_t origin = 0;
    int rc = dk_bwt_forward(ctx.get(), input.data(), input.size(), out.data(), &origin);
    if (rc != DK_OK) throw Error(rc, ctx.error());
    return {out, origin};
}
inline std::vector<uint8_t> decode(detail::Ctx &ctx, const std::vector<uint8_t> &bwt, size_t origin) {
    std::vector<uint8_t> out(bwt.size());
    int rc = dk_bwt_inverse(ctx.get(), bwt.data(), bwt.size(), static_cast<uint32_t>(origin), out.data());
    if (rc != DK_OK) throw Error(rc, ctx.error());
    return out;
}
}  // namespace bwt

}  // namespace dark
