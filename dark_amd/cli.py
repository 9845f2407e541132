"""Command line front end with the reference's conventions (src/main.rs:34-116).

    python -m dark_amd.cli [-m dark|exp|ybs|simple|rawdc] FILE          -> ./FILE.dark   = [u32 LE n][coded stream]
    python -m dark_amd.cli [-m MODEL] FILE.dark                         -> ./FILE.orig

The default model is `exp`, as in src/main.rs:52.  A file made here with one block is byte-for-byte what the reference writes
(to the extent DESIGN.md section 2 pins it).  Extension beyond the reference: `-b BYTES` cuts the input into blocks and
concatenates their records; decoding walks the records (each decode reports how many stream bytes it consumed).  The reference
reads such a file as its first block only.
"""
import argparse
import os
import struct
import sys

import numpy as np

EXTENSION = ".dark"


def encode_file(path, model, block_size, device):
    from .context import Context
    data = np.fromfile(path, dtype=np.uint8)
    n_total = len(data)
    if n_total == 0:
        raise SystemExit("empty input: the reference panics on it (src/saca.rs:107)")
    bs = block_size or n_total
    out_path = os.path.basename(path) + EXTENSION  # like main.rs:96-98: next to the CWD, input name + .dark
    with Context(min(bs, n_total), device) as ctx, open(out_path, "wb") as out:
        for off in range(0, n_total, bs):
            block = data[off:off + bs]
            out.write(struct.pack("<I", len(block)))          # main.rs:102
            if model == "raw":
                # block::raw::Encoder with model::raw::Out (src/block/raw.rs:35-59, src/model/raw.rs:46-76): origin as four
                # symbols then every BWT byte go to ./out.raw; nothing reaches the coder, whose tail is four zero bytes
                bwt, origin = ctx.bwt_forward(block)
                with open("out.raw", "ab" if off else "wb") as dump:
                    dump.write(struct.pack(">I", origin))
                    dump.write(bwt.tobytes())
                out.write(b"\0\0\0\0")
            elif model == "rawdc":
                # block::dc::Encoder with model::raw::DcOut (src/model/raw.rs:12-44): 10-byte records go to ./out-dc.raw
                with open("out-dc.raw", "ab" if off else "wb") as dump:
                    dump.write(ctx.block_encode("rawdc", block))
                out.write(b"\0\0\0\0")
            else:
                out.write(ctx.block_encode(model, block))     # main.rs:104-113
    return out_path


def decode_file(path, model, device):
    from .context import Context
    blob = np.fromfile(path, dtype=np.uint8)
    base = os.path.basename(path)
    out_path = base[:-len(EXTENSION)] + ".orig"             # main.rs:64-66
    pos, ctx = 0, None
    with open(out_path, "wb") as out:
        while pos < len(blob):
            if pos + 4 > len(blob):
                raise SystemExit("truncated record header at byte %d" % pos)
            (n,) = struct.unpack("<I", blob[pos:pos + 4].tobytes())   # main.rs:70
            pos += 4
            if ctx is None or ctx.capacity() < n:
                if ctx is not None:
                    ctx.close()
                ctx = Context(n, device)
            out.write(ctx.block_decode(model, blob[pos:], n))
            pos += ctx.last_consumed()
    if ctx is not None:
        ctx.close()
    return out_path


def main(argv=None):
    ap = argparse.ArgumentParser(prog="dark_amd.cli", description="Dark compressor usage: [options] input_file[.dark]")
    ap.add_argument("-m", "--model", default="exp", help="dark|exp|ybs|simple|raw|rawdc (default exp, like the reference; raw and rawdc are dump-only)")
    ap.add_argument("-b", "--block-size", type=int, default=0, help="cut the input into blocks of this many bytes (extension)")
    ap.add_argument("-d", "--device", type=int, default=0)
    ap.add_argument("file")
    args = ap.parse_args(argv)
    if args.file.endswith(EXTENSION):
        print(decode_file(args.file, args.model, args.device))
    else:
        print(encode_file(args.file, args.model, args.block_size, args.device))


if __name__ == "__main__":
    main(sys.argv[1:])
