"""Command line front end with the reference's conventions (src/main.rs:34-116).

    python -m dark_amd.cli [-m dark|exp|ybs|simple|bbb|raw|rawdc] FILE       -> ./FILE with its extension replaced by .dark
    python -m dark_amd.cli [-m MODEL] FILE.dark                           -> ./FILE.orig

The default model is `exp`, as in src/main.rs:52.  Output names follow PathBuf::set_extension (src/main.rs:64-66,96-98):
book.txt -> book.dark, book -> book.dark, book.dark -> book.orig; the output lands in the current directory.  A file made here with
one block is byte-for-byte what the reference writes: [u32 LE n][coded stream] (to the extent DESIGN.md section 2 pins it).

Extensions beyond the reference (whole file = one block, read into memory, one thread):
  -b BYTES   cut the input into blocks; records [u32 LE n][stream] are concatenated, each byte-identical to a single-block run on that
             block, followed by an index footer (record offsets) so that decoding can be batched and spread over GPUs.  The reference
             reads such a file as its first block only.  Blocks are STREAMED: one block of input is in host memory at a time; each goes
             through dk_batch_push as soon as it is in HBM (file reading, upload, GPU stages and the host coding of earlier blocks overlap).
  --gpus G   block b -> GPU b mod G, one worker process per GPU; the parent never touches a GPU and stitches the records in order.
  --force    encode blocks that contain byte 0xFF.  The reference's header cannot carry that symbol (src/block/dc.rs:57,60,73,127):
             it writes such an archive without complaint and can never decode it.  This front end refuses unless --force is given.
"""
import argparse
import os
import time
import struct
import subprocess
import sys
import threading
import queue

import numpy as np

EXTENSION = "dark"
STATS = {}  # --stats: when the library and its context were ready (imports, workspace allocation done)
FOOTER_MAGIC = b"DKIX"
DUMP_MODELS = ("raw", "rawdc")
RAW_CODING_MODELS = ("bbb",)  # block::raw with a coding RawModel (src/main.rs:73,104): block after block through dk_raw_block_encode


def set_extension(name, ext):
    """std::path::PathBuf::set_extension on a bare file name: the part after the last '.' is replaced (a leading '.' alone does not start
    an extension); a name without extension gets one appended."""
    i = name.rfind(".")
    stem = name[:i] if i > 0 else name
    return stem + "." + ext


def output_name(path, ext):
    return set_extension(os.path.basename(path), ext)  # main.rs:96-98 / 64-66: file name only -> current directory


def has_extension(path, ext):
    name = os.path.basename(path)
    i = name.rfind(".")
    return i > 0 and name[i + 1:] == ext


class Refused(SystemExit):
    pass


class _AtomicOutput:
    """The archive is written to a temporary file next to its final name and renamed over it only when it is complete (records and
    footer written, file closed): a refusal, a GPU error or a failed worker leaves an existing archive of that name untouched and never
    leaves a truncated one behind -- a footer-less prefix of records would otherwise decode as a valid, shorter file."""

    def __init__(self, out_path):
        self.final = out_path
        self.tmp = os.path.join(os.path.dirname(out_path) or ".", ".%s.tmp%d" % (os.path.basename(out_path), os.getpid()))
        self.f = None

    def __enter__(self):
        self.f = open(self.tmp, "wb")
        return self.f

    def __exit__(self, exc_type, exc, tb):
        self.f.close()
        if exc_type is None:
            os.replace(self.tmp, self.final)
        else:
            try:
                os.remove(self.tmp)
            except OSError:
                pass
        return False


_NOTED = set()


def _note(key, text):
    """one line on stderr, once per run"""
    if key not in _NOTED:
        _NOTED.add(key)
        print("dark_amd: " + text, file=sys.stderr)


def _note_model(model):
    if model in RAW_CODING_MODELS:
        _note("bbb", "-m bbb is EXPERIMENTAL here: the model's gates restate compress::entropy::ari::apm from an in-repo analogue (parity "
                     "unpinned, DESIGN.md section 7); archives decode with this program, compatibility with the Rust crate's bytes is not claimed")


def _note_single_symbol(block, where):
    if len(block) and int(block.min()) == int(block.max()):
        _note("single", "%s holds one distinct byte value: this program decodes it, the reference's decoder mis-reads `origin` for such a "
                        "block and returns one byte (DESIGN.md, reference quirks)" % where)


def _check_ff(block, force, where):
    if not force and bool((block == 255).any()):
        raise Refused("%s contains byte 0xFF: the reference format cannot carry that symbol (src/block/dc.rs:57-73), the archive could "
                      "never be decoded.  Pass --force to write it anyway (bit-exact with what the reference writes)." % where)


# ---- single block: the reference's own behaviour ------------------------------------------------------------------------------------
def _encode_single(path, model, device, force, out_path):
    from .context import Context
    data = np.fromfile(path, dtype=np.uint8)  # main.rs:87-95 reads the whole file: one block
    n = len(data)
    if n == 0:
        raise SystemExit("empty input: the reference panics on it (src/saca.rs:107)")
    if model not in DUMP_MODELS and model not in RAW_CODING_MODELS:
        _check_ff(data, force, "the block")               # before the output is touched
    _note_single_symbol(data, "the input")
    with Context(n, device) as ctx, _AtomicOutput(out_path) as out:
        out.write(struct.pack("<I", n))                   # main.rs:102
        _write_block(ctx, model, data, out, True, force)
    return out_path


def _write_block(ctx, model, block, out, first, force):
    if model == "raw":
        # block::raw::Encoder with model::raw::Out (src/block/raw.rs:35-59, src/model/raw.rs:46-76): origin as four symbols, then every
        # BWT byte, go to ./out.raw; nothing reaches the coder, whose tail is four zero bytes
        with open("out.raw", "wb" if first else "ab") as dump:
            dump.write(ctx.raw_block_encode_dump(block))
        out.write(b"\0\0\0\0")
    elif model == "rawdc":
        # block::dc::Encoder with model::raw::DcOut (src/model/raw.rs:12-44): 10-byte records go to ./out-dc.raw
        with open("out-dc.raw", "wb" if first else "ab") as dump:
            dump.write(ctx.block_encode("rawdc", block))
        out.write(b"\0\0\0\0")
    elif model in RAW_CODING_MODELS:
        # block::raw::Encoder with model::bbb::Model (src/main.rs:104): every byte value can be carried, no 0xFF restriction
        out.write(ctx.raw_block_encode(block, 1))
    else:
        _check_ff(block, force, "the block")
        out.write(ctx.block_encode(model, block))         # main.rs:104-113


# ---- streamed, pipelined encode on one GPU ---------------------------------------------------------------------------------------------
def _reader(f, block_size, first_block, step, total_blocks, q, force, device):
    """reads blocks first_block, first_block+step, ... one at a time and uploads each to HBM; at most two are waiting to be pushed"""
    import torch
    try:
        for b in range(first_block, total_blocks, step):
            f.seek(b * block_size)
            block = np.fromfile(f, dtype=np.uint8, count=block_size)  # ONE block of input in host memory
            _check_ff(block, force, "block %d" % b)
            _note_single_symbol(block, "block %d" % b)
            q.put((b, len(block), torch.from_numpy(block).to("cuda:%d" % device)))
            del block
        q.put(None)
    except BaseException as e:  # noqa: BLE001 -- handed to the consumer, which re-raises
        q.put(e)


def _encode_blocks(path, model, block_size, device, first_block, step, total_blocks, out, index, force, host_threads):
    """encode blocks first_block, first_block+step, ... of `path` in order into `out`; index gets (block number, record length).
    Every block goes through dk_batch_push as soon as it is in HBM: its device stages run at once, its coding on one of the host threads;
    file reading, upload, GPU stages and coding all overlap.  Records are written out every `flush_every` blocks."""
    import torch
    from .context import Context
    torch.cuda.set_device(device)
    flush_every = int(max(2 * host_threads, min(256, (1 << 32) // max(1, block_size))))  # at most about 4 GB of input between flushes
    q = queue.Queue(maxsize=2)
    with open(path, "rb") as f, Context(block_size, device) as ctx:
        STATS["t_ready"] = time.perf_counter()
        t = threading.Thread(target=_reader, args=(f, block_size, first_block, step, total_blocks, q, force, device), daemon=True)
        t.start()
        batch, pending = None, []

        def flush():
            nonlocal batch, pending
            if batch is None:
                return
            streams = batch.finish()
            for (b, n), s in zip(pending, streams):
                out.write(struct.pack("<I", n))
                out.write(memoryview(s))
                index.append((b, 4 + len(s)))
            batch, pending = None, []

        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                b, n, d = item
                if batch is None:
                    batch = ctx.batch_begin(model, host_threads)
                batch.push(d, n)  # a failed push closes the batch (joins the coders of the blocks before it) and raises
                pending.append((b, n))
                del d, item
                if len(pending) >= flush_every:
                    flush()
            flush()
        finally:
            if batch is not None:
                batch.close()  # error path: the coding threads are joined before their output buffers and the context go away
        t.join()


def _write_footer(out, offsets):
    """index of a multi-record file: 'DKIX' u32 count, count x u64 record offsets, u64 offset of this footer, 'DKIX'"""
    start = out.tell()
    out.write(FOOTER_MAGIC + struct.pack("<I", len(offsets)))
    out.write(struct.pack("<%dQ" % len(offsets), *offsets))
    out.write(struct.pack("<Q", start) + FOOTER_MAGIC)


def read_footer(path):
    """-> list of record offsets + the offset where the records end, or None for a file without index (single block / legacy)"""
    size = os.path.getsize(path)
    if size < 24:
        return None
    with open(path, "rb") as f:
        f.seek(size - 12)
        tail = f.read(12)
        if tail[8:] != FOOTER_MAGIC:
            return None
        (start,) = struct.unpack("<Q", tail[:8])
        if start + 20 > size:
            return None
        f.seek(start)
        head = f.read(8)
        if head[:4] != FOOTER_MAGIC:
            return None
        (count,) = struct.unpack("<I", head[4:])
        if start + 8 + 8 * count + 12 != size:
            return None
        offs = list(struct.unpack("<%dQ" % count, f.read(8 * count)))
    return offs, start


def _host_threads():
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        cpus = max(1, int(q) // int(period)) if q != "max" else (os.cpu_count() or 1)
    except (OSError, ValueError):
        cpus = os.cpu_count() or 1
    return cpus


def _devices(device, gpus, devices):
    """GPU of worker r: --devices a,b,... (e.g. 0,0 rehearses two workers on a one-GPU box), else device + r"""
    if devices:
        ids = [int(x) for x in str(devices).split(",")]
        if len(ids) != gpus:
            raise SystemExit("--devices needs exactly --gpus entries")
        return ids
    return [device + r for r in range(gpus)]


def encode_file(path, model, block_size=0, device=0, gpus=1, force=False, host_threads=0, devices=None):
    out_path = output_name(path, EXTENSION)
    total = os.path.getsize(path)
    if total == 0:
        raise SystemExit("empty input: the reference panics on it (src/saca.rs:107)")
    _note_model(model)
    if not block_size or block_size >= total:
        return _encode_single(path, model, device, force, out_path)
    nblocks = -(-total // block_size)
    if model in DUMP_MODELS or model in RAW_CODING_MODELS:  # block after block through the host entry points
        from .context import Context
        with open(path, "rb") as f, Context(block_size, device) as ctx, _AtomicOutput(out_path) as out:
            for b in range(nblocks):
                block = np.fromfile(f, dtype=np.uint8, count=block_size)
                _note_single_symbol(block, "block %d" % b)
                out.write(struct.pack("<I", len(block)))
                _write_block(ctx, model, block, out, b == 0, force)
        return out_path
    threads = host_threads or max(1, _host_threads() // max(1, gpus) - 1)
    if gpus <= 1:
        index = []
        with _AtomicOutput(out_path) as out:
            _encode_blocks(path, model, block_size, device, 0, 1, nblocks, out, index, force, threads)
            offsets, pos = [], 0
            for _, ln in index:
                offsets.append(pos)
                pos += ln
            _write_footer(out, offsets)
        return out_path
    # one worker process per GPU (block b -> GPU b mod G); this parent has not touched and does not touch a GPU
    parts = ["%s.part%d" % (out_path, r) for r in range(gpus)]
    procs = []
    devs = _devices(device, gpus, devices)
    for r in range(gpus):
        cmd = [sys.executable, "-m", "dark_amd.cli", "--worker", "%d/%d" % (r, gpus), "--part", parts[r], "-m", model, "-b", str(block_size),
               "-d", str(devs[r]), "--host-threads", str(threads)] + (["--force"] if force else []) + [path]
        procs.append(subprocess.Popen(cmd, env=dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                                                                               os.environ.get("PYTHONPATH", "").split(os.pathsep)))))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:
        for part in parts:
            for name in (part, part + ".idx"):
                if os.path.exists(name):
                    os.remove(name)
        raise SystemExit("a GPU worker failed (exit code %d)" % rc)
    # stitch: records in block order, each worker's part holds its blocks in its own order
    lens = [[int(x) for x in open(part + ".idx").read().split()] for part in parts]
    files = [open(part, "rb") for part in parts]
    offsets, pos = [], 0
    with _AtomicOutput(out_path) as out:
        for b in range(nblocks):
            r, k = b % gpus, b // gpus
            ln = lens[r][k]
            offsets.append(pos)
            out.write(files[r].read(ln))
            pos += ln
        _write_footer(out, offsets)
    for fobj, part in zip(files, parts):
        fobj.close()
        os.remove(part)
        os.remove(part + ".idx")
    return out_path


def _worker_encode(args):
    r, g = (int(x) for x in args.worker.split("/"))
    total = os.path.getsize(args.file)
    nblocks = -(-total // args.block_size)
    index = []
    with open(args.part, "wb") as out:
        _encode_blocks(args.file, args.model, args.block_size, args.device, r, g, nblocks, out, index, args.force, args.host_threads or 1)
    with open(args.part + ".idx", "w") as f:
        f.write(" ".join(str(ln) for _, ln in index))


# ---- decode -----------------------------------------------------------------------------------------------------------------------------
def _decode_records_batched(path, model, device, offsets, end, which, out_write, host_threads):
    """decode records `which` (indices into offsets) through dk_dev_batch_decode, in batches; out_write(k, bytes) receives them in order"""
    import torch
    from .context import Context
    torch.cuda.set_device(device)
    bounds = offsets + [end]
    with open(path, "rb") as f:
        sizes = []
        for k in which:
            f.seek(offsets[k])
            sizes.append(struct.unpack("<I", f.read(4))[0])
        if not sizes:
            return
        batch = int(max(2, min(4 * host_threads, (1 << 30) // max(sizes))))  # two batches of outputs live in HBM at a time
        with Context(max(sizes), device) as ctx:
            STATS["t_ready"] = time.perf_counter()
            # a writer thread downloads and writes the blocks of one batch while the next batch is being decoded
            wq = queue.Queue(maxsize=1)
            werr = []

            def writer():
                while True:
                    item = wq.get()
                    if item is None:
                        return
                    try:
                        if not werr:
                            for k, d in zip(*item):
                                out_write(k, d.cpu().numpy().tobytes())
                    except BaseException as e:  # noqa: BLE001 -- re-raised by the main thread
                        werr.append(e)

            wt = threading.Thread(target=writer, daemon=True)
            wt.start()
            try:
                for lo in range(0, len(which), batch):
                    ks = which[lo:lo + batch]
                    ns = sizes[lo:lo + batch]
                    streams = []
                    for k in ks:
                        f.seek(offsets[k] + 4)
                        streams.append(np.fromfile(f, dtype=np.uint8, count=bounds[k + 1] - offsets[k] - 4))
                    d_outs = [torch.empty(n, dtype=torch.uint8, device="cuda:%d" % device) for n in ns]
                    ctx.dev_batch_decode(model, streams, ns, d_outs, host_threads)
                    wq.put((ks, d_outs))
                    del d_outs, streams
                    if werr:
                        break
            finally:
                wq.put(None)
                wt.join()
            if werr:
                raise werr[0]


def decode_file(path, model, device=0, gpus=1, host_threads=0, devices=None):
    from .context import Context
    out_path = output_name(path, "orig")                    # main.rs:64-66
    if model in DUMP_MODELS:
        raise SystemExit("model %s is dump-only: there is nothing to decode (src/model/raw.rs:39-43 panics)" % model)
    footer = read_footer(path) if model not in RAW_CODING_MODELS else None  # bbb archives carry no index: records are walked
    threads = host_threads or max(1, _host_threads() // max(1, gpus) - 1)
    if footer is None:
        # no index: the reference's single-block file (main.rs:70), or concatenated records walked by the bytes each decode consumed
        size = os.path.getsize(path)
        with open(path, "rb") as f, open(out_path, "wb") as out:
            blob = np.fromfile(f, dtype=np.uint8)
            pos, ctx, records = 0, None, 0
            while pos < size:
                if pos + 4 > size:
                    raise SystemExit("truncated record header at byte %d" % pos)
                (n,) = struct.unpack("<I", blob[pos:pos + 4].tobytes())   # main.rs:70
                pos += 4
                if ctx is None or ctx.capacity() < n:
                    if ctx is not None:
                        ctx.close()
                    ctx = Context(n, device)
                if model in RAW_CODING_MODELS:
                    out.write(ctx.raw_block_decode(blob[pos:], n, 1))   # block::raw::Decoder, main.rs:73
                else:
                    out.write(ctx.block_decode(model, blob[pos:], n))
                pos += ctx.last_consumed()
                records += 1
            if ctx is not None:
                ctx.close()
        if records > 1:
            _note("walked", "%d records without an index footer: every record decoded, but a file cut off at a record boundary cannot be "
                            "told from a complete one (archives written by -b carry a footer; bbb archives and plain concatenations do not)" % records)
        return out_path
    offsets, end = footer
    if gpus <= 1:
        with open(out_path, "wb") as out:
            _decode_records_batched(path, model, device, offsets, end, list(range(len(offsets))), lambda k, data: out.write(data), threads)
        return out_path
    parts = ["%s.part%d" % (out_path, r) for r in range(gpus)]
    procs = []
    devs = _devices(device, gpus, devices)
    for r in range(gpus):
        cmd = [sys.executable, "-m", "dark_amd.cli", "--worker", "%d/%d" % (r, gpus), "--part", parts[r], "-m", model, "-d", str(devs[r]),
               "--host-threads", str(threads), path]
        procs.append(subprocess.Popen(cmd, env=dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                                                                               os.environ.get("PYTHONPATH", "").split(os.pathsep)))))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:
        raise SystemExit("a GPU worker failed (exit code %d)" % rc)
    files = [open(part, "rb") for part in parts]
    with open(path, "rb") as f, open(out_path, "wb") as out:
        for k in range(len(offsets)):
            f.seek(offsets[k])
            (n,) = struct.unpack("<I", f.read(4))
            out.write(files[k % gpus].read(n))
    for fobj, part in zip(files, parts):
        fobj.close()
        os.remove(part)
    return out_path


def _worker_decode(args):
    r, g = (int(x) for x in args.worker.split("/"))
    offsets, end = read_footer(args.file)
    with open(args.part, "wb") as out:
        _decode_records_batched(args.file, args.model, args.device, offsets, end, list(range(r, len(offsets), g)), lambda k, data: out.write(data),
                                args.host_threads or 1)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="dark_amd.cli", description="Dark compressor usage: [options] input_file[.dark]")
    ap.add_argument("-m", "--model", default="exp", help="dark|exp|ybs|simple|bbb|raw|rawdc (default exp, like the reference; raw and rawdc are dump-only; bbb: DESIGN.md section 7)")
    ap.add_argument("-b", "--block-size", type=int, default=0, help="cut the input into blocks of this many bytes (extension; streamed and batched)")
    ap.add_argument("-d", "--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1, help="block b -> GPU b mod G, one worker process per GPU (needs -b / an indexed archive)")
    ap.add_argument("--force", action="store_true", help="write archives of blocks containing byte 0xFF (undecodable in the reference format)")
    ap.add_argument("--host-threads", type=int, default=0, help="host coding threads per GPU (default: CPU quota / gpus - 1)")
    ap.add_argument("--devices", default="", help="GPU ids of the workers, comma separated (default: device, device+1, ...)")
    ap.add_argument("--stats", action="store_true", help="print a JSON line with the wall time of the work and the peak RSS to stderr")
    ap.add_argument("--worker", default="", help=argparse.SUPPRESS)
    ap.add_argument("--part", default="", help=argparse.SUPPRESS)
    ap.add_argument("file")
    args = ap.parse_args(argv)
    decode = has_extension(args.file, EXTENSION)           # main.rs:56: direction by extension
    if args.worker:
        return _worker_decode(args) if decode else _worker_encode(args)
    t0 = time.perf_counter()
    if decode:
        out = decode_file(args.file, args.model, args.device, args.gpus, args.host_threads, args.devices)
    else:
        out = encode_file(args.file, args.model, args.block_size, args.device, args.gpus, args.force, args.host_threads, args.devices)
    print(out)
    if args.stats:  # wall time of the work itself (interpreter start and imports of this front end excluded) and this process's peak RSS
        import json
        import resource
        t1 = time.perf_counter()
        print(json.dumps({"seconds": round(t1 - t0, 3), "seconds_after_setup": round(t1 - STATS.get("t_ready", t0), 3), "input_bytes": os.path.getsize(args.file),
                          "output_bytes": os.path.getsize(out), "peak_rss_bytes": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024}),
              file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
