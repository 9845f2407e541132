#!/usr/bin/env python3
"""File-level throughput of the streamed / batched CLI path: a synthetic file of `--blocks` x `--block-bytes`
(wiki-like text, BASELINE configs[3] shape) is written to a scratch directory, encoded with `python -m dark_amd.cli -b ...` in a child
process, decoded again and compared.  Reports MB/s of both directions and the child's peak RSS (the input must not be slurped).

    python tools/cli_throughput.py [--blocks 8] [--block-bytes 125000000] [--model dark] [--gpus 1]   (on the GPU box)"""
import argparse
import hashlib
import json
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(cmd, cwd):
    t = time.perf_counter()
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    r = subprocess.run(cmd, cwd=cwd, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    dt = time.perf_counter() - t
    if r.returncode != 0:
        sys.exit(r.stdout + r.stderr)
    inside = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
    # (process wall, the CLI's own --stats line, high-water RSS mark of all children so far)
    return dt, (inside[-1] if inside else {}), max(before, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss) * 1024


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--block-bytes", type=int, default=125_000_000)
    ap.add_argument("--model", default="dark")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", default="")
    ap.add_argument("--repeat", type=int, default=1, help="write the generated blocks this many times (bigger file, same generation time)")
    args = ap.parse_args()
    from dark_amd import datagen
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        src = os.path.join(d, "enwik9_like.txt")
        with open(src, "wb") as f:
            made = [datagen.wiki_like(args.block_bytes, 40 + b) for b in range(args.blocks)]
            for _ in range(args.repeat):
                for blk in made:
                    blk.tofile(f)
            del made
        total = os.path.getsize(src)
        extra = ["--gpus", str(args.gpus)] + (["--devices", args.devices] if args.devices else [])
        cmd = [sys.executable, "-m", "dark_amd.cli", "--stats", "-m", args.model, "-b", str(args.block_bytes)] + extra
        t_enc, in_enc, rss_enc = run(cmd + [src], d)
        packed = os.path.join(d, "enwik9_like.dark")
        t_dec, in_dec, rss_dec = run([sys.executable, "-m", "dark_amd.cli", "--stats", "-m", args.model] + extra + [packed], d)
        ok = sha(os.path.join(d, "enwik9_like.orig")) == sha(src)
        print(json.dumps({"file_bytes": total, "blocks": args.blocks * args.repeat, "block_bytes": args.block_bytes, "model": args.model, "gpus": args.gpus,
                          "encode_s": round(t_enc, 2), "encode_MBps": round(total / t_enc / 1e6, 1), "decode_s": round(t_dec, 2),
                          "decode_MBps": round(total / t_dec / 1e6, 1),
                          "encode_inside": in_enc, "encode_MBps_inside": round(total / in_enc["seconds"] / 1e6, 1) if in_enc else None,
                          "encode_MBps_after_setup": round(total / in_enc["seconds_after_setup"] / 1e6, 1) if in_enc else None,
                          "decode_MBps_after_setup": round(total / in_dec["seconds_after_setup"] / 1e6, 1) if in_dec else None,
                          "decode_inside": in_dec, "decode_MBps_inside": round(total / in_dec["seconds"] / 1e6, 1) if in_dec else None, "packed_bytes": os.path.getsize(packed), "roundtrip_ok": ok,
                          "peak_rss_bytes_children": rss_dec, "peak_rss_in_blocks": round(rss_dec / args.block_bytes, 2), "input_unique_blocks": args.blocks,
                          "note": "wall time of the whole CLI process (python start, library load, context creation, file I/O included); "
                                  "RSS includes the library's pinned staging of the distance streams (about 2.5 bytes per input byte per "
                                  "block in flight), not only the CLI's own buffers"}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
