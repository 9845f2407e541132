"""Builds dark_amd/libdark_amd.so (HIP kernels for gfx950 + host entropy stage + C ABI) with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
SO = os.path.join(HERE, "libdark_amd.so")
SOURCES = ["abi.cpp", "context.cpp", "entropy.cpp", "bbb.cpp", "radix_sort.hip", "suffix_array.hip", "bwt.hip", "dc.hip"]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "--offload-arch=" + ARCH]


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    out.append(os.path.join(HERE, "..", "include", "dark_amd.h"))
    return out


SO_TUNING = os.path.join(HERE, "libdark_amd_tuning.so")
# sources that hold A/B switches (DK_KNOB, csrc/context.hpp): the tuning build compiles them with -DDK_TUNING, everything else is shared
TUNING_SOURCES = ("context.cpp", "radix_sort.hip", "suffix_array.hip", "bwt.hip")


def build(force=False, verbose=False, tuning=False):
    """the product library; tuning=True: libdark_amd_tuning.so as well -- the same code with its A/B switches (DK_XCD, DK_PREFIX,
    DK_RADIX_WIDE ...) readable from the environment, for tests/test_env_variants.py and the tools/ (DARK_AMD_LIB selects it)"""
    so = _build_one(force, verbose, False)
    if tuning:
        _build_one(force, verbose, True)
    return so


def _build_one(force, verbose, tuning):
    os.makedirs(OBJ, exist_ok=True)
    newest_dep = max(os.path.getmtime(p) for p in _deps())
    objs, rebuilt = [], False
    SO = SO_TUNING if tuning else globals()["SO"]
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        special = tuning and src in TUNING_SOURCES
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + (".tuning.o" if special else ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < newest_dep:
            # host-only sources: x86-64-v3 (AVX2, BMI2, LZCNT, MOVBE) -- every host that carries an MI355X has it
            extra = ["-x", "hip"] if src.endswith(".hip") else ["-march=x86-64-v3"]
            if src == "entropy.cpp":
                # the models, the sinks and the coder are small functions calling each other once per coded decision: with the
                # default threshold clang leaves some of them out of line (measured on the GPU box: 22.5 -> 18.4 ns per distance, decode 42 -> 37)
                extra += ["-mllvm", "-inline-threshold=20000"]
                # the pipeline's loops are a few dozen instructions each and their speed moves by 2-3 % with where they happen to start
                # (profiles/r04_entropy_ab.json: the same source 450.8 / 441.6 / 399.4 ms aligned against 456.6 / 454.9 / 407.4 ms)
                extra += ["-falign-loops=32", "-falign-functions=64"]
            if special:
                extra += ["-DDK_TUNING"]
            cmd = [HIPCC] + CXXFLAGS + extra + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", SO] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv)
    print(SO)
