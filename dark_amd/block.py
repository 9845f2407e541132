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
