"""A short run of the randomized parity campaign (tools/fuzz_gpu.py): suffix array, BWT + origin, DC arrays and whole-block streams
against the oracle on inputs built to cross the thresholds of the suffix sort and of the distance coder."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lfirst", ["1", "2", "switch"])
@pytest.mark.parametrize("prefix_mode", ["1", "2"])
def test_randomized_campaign(prefix_mode, lfirst):
    """DK_PREFIX is a switch of the TUNING build (csrc/context.hpp: the product library has every switch compiled in and never reads the
    environment): mode 1 is the product's own choice of the initial key length, mode 2 forces the shortest candidate -- every input then
    takes the short-prefix / narrow-key / text-round route.  The tool is pointed at the tuning library so that the two are different
    campaigns (with the product library they were the same ten cases twice).  DK_LFIRST=2 forces the L-first BWT path on
    every input (the product takes it for text-like blocks only)."""
    tuning = os.path.join(ROOT, "dark_amd", "libdark_amd_tuning.so")
    assert os.path.exists(tuning), "build with tuning=True (__graft_entry__.build does)"
    env = dict(os.environ, DK_PREFIX=prefix_mode, DARK_AMD_LIB=tuning)
    env.update({"DK_LFIRST": "1", "DK_LF_SWITCH": "60"} if lfirst == "switch" else {"DK_LFIRST": lfirst})  # switch: the L-first path takes over mid-way
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "8", "5"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_dc_stage_campaign():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "dc", "150", "9"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]
