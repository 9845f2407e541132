"""Mirror of the reference's `block` module (src/block/mod.rs:27-36, src/block/dc.rs): `dc.Encoder(n, model).encode(bytes)`
and `dc.Decoder(n, model).decode(stream)`.  Writers/readers of the Rust signature become return values."""
from .context import Context, model_id


class _Dc:
    class Encoder:
        def __init__(self, n, model, device=0, ctx=None):
            self.model = model  # public like block::dc::Encoder.model (src/block/dc.rs:25)
            self._n = n
            self._ctx = ctx or Context(n, device)
            model_id(model)

        def encode(self, data):
            if len(data) > self._ctx.capacity():  # src/block/dc.rs:43
                raise ValueError("block larger than the encoder capacity")
            return self._ctx.block_encode(self.model, data)

    class Decoder:
        def __init__(self, n, model, device=0, ctx=None):
            self.model = model
            self._n = n
            self._ctx = ctx or Context(n, device)
            model_id(model)

        def decode(self, stream):
            return self._ctx.block_decode(self.model, stream, self._n)


dc = _Dc


RAW_MODELS = {"raw": 0, "out": 0, "bbb": 1}  # model::raw::Out, model::bbb::Model (src/main.rs:73,76,104,107)


class _Raw:
    """src/block/raw.rs: `raw.Encoder(n, "bbb").encode(bytes)` -> coded stream; with the dump model "raw" the symbols the reference
    would append to ./out.raw are kept in `.dumped` and the stream is the idle coder's 4-byte tail."""

    class Encoder:
        def __init__(self, n, model, device=0, ctx=None):
            self.model = model
            self._id = RAW_MODELS[model]
            self._ctx = ctx or Context(n, device)
            self.dumped = b""

        def encode(self, data):
            if len(data) > self._ctx.capacity():  # src/block/raw.rs:37
                raise ValueError("block larger than the encoder capacity")
            if self._id == 0:
                self.dumped = self._ctx.raw_block_encode_dump(data)
                return b"\0\0\0\0"
            return self._ctx.raw_block_encode(data, self._id)

    class Decoder:
        def __init__(self, n, model, device=0, ctx=None):
            self.model = model
            self._id = RAW_MODELS[model]
            self._n = n
            self._ctx = ctx or Context(n, device)

        def decode(self, stream):
            return self._ctx.raw_block_decode(stream, self._n, self._id)


raw = _Raw
