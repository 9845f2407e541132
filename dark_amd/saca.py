"""Mirror of the reference's `saca` module (src/saca.rs:344-384)."""
from .context import Context


class Constructor:
    """saca::Constructor: `new(max_n)`, `capacity()`, `compute(input) -> suffix array`."""

    def __init__(self, max_n, device=0):
        self._ctx = Context(max_n, device)
        self._n = max_n

    def capacity(self):
        return self._ctx.capacity()

    def compute(self, data):
        if len(data) != self._n:  # src/saca.rs:369 assert_eq!(input.len(), self.n)
            raise ValueError("Constructor sized for %d bytes got %d" % (self._n, len(data)))
        return self._ctx.suffix_array(data)

    def context(self):
        """the analogue of reuse(): the device workspace is lent to the later stages through the context"""
        return self._ctx
