"""Real text from the box itself (dark_amd.datagen.real_text): the Python sources (and other text files) under the interpreter's library
directories, concatenated to a block of N bytes -- the image holds no corpus (book1, enwik8), but it holds tens of megabytes of real source
text: words, indentation runs, licence boilerplate repeated hundreds of times.  BWT + origin against the oracle (TEST INFRASTRUCTURE use, like tests/), and suffix sort + BWT times on the
product's path, with the L-first path forced and with it off (tuning build).   python tools/real_text.py [N] [suffixes, default .py,.pyi,.txt,.rst,.md,.h,.hpp]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    exts = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else (".py", ".pyi", ".txt", ".rst", ".md", ".h", ".hpp")
    if len(sys.argv) > 3:  # child: time one configuration
        import torch, dark_amd
        t = np.fromfile(sys.argv[3], dtype=np.uint8)
        d = torch.from_numpy(t).cuda(); out = torch.empty(len(t), dtype=torch.uint8, device="cuda")
        with dark_amd.Context(len(t)) as ctx:
            ctx.dev_bwt_forward(d, len(t), out)
            ms = []
            for _ in range(3):
                ctx.dev_bwt_forward(d, len(t), out); st = ctx.stats(); ms.append(st["ms_sa"] + st["ms_bwt"])
            print("%-28s suffix sort + BWT %.2f ms  rounds %d passes %d routes %s" % (os.environ.get("LABEL", ""), sorted(ms)[1], st["rounds"], st["sort_passes"], sorted(st["routes"])), flush=True)
            np.save(sys.argv[3] + ".bwt.npy", out.cpu().numpy())
        return
    t0 = time.time()
    from dark_amd import datagen
    t = datagen.real_text(n, 0, exts)  # the bench / parity workload realtext_5e7 is real_text(50_000_000)
    print("gathered %d bytes of real text in %.1f s, %d distinct byte values" % (len(t), time.time() - t0, len(np.unique(t))), flush=True)
    path = "/tmp/real_text.bin"
    t.tofile(path)
    from oracle import orc
    t0 = time.time(); wb, wo = orc.bwt_forward(t); print("oracle SA-IS + BWT %.1f s" % (time.time() - t0), flush=True)
    tuning = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark_amd", "libdark_amd_tuning.so")
    for label, env in (("product", {}), ("L-first forced", {"DARK_AMD_LIB": tuning, "DK_LFIRST": "2"}), ("suffix-array path", {"DARK_AMD_LIB": tuning, "DK_LFIRST": "0"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(n), ",".join(exts), path], env=dict(os.environ, LABEL=label, **env), check=True)
        got = np.load(path + ".bwt.npy")
        print("    equal to the oracle's BWT:", bool((got == np.frombuffer(wb, np.uint8)).all()), flush=True)


if __name__ == "__main__":
    main()
