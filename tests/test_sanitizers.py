"""Host entropy stage under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not available on the pool):
tools/ent_fuzz.cpp round-trips every model and decodes thousands of mutated / truncated streams and random distance arrays."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++")
def test_host_decoders_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "ent_fuzz")
    cmd = [CLANG, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-march=x86-64-v3",
           "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "ent_fuzz.cpp"),
           os.path.join(ROOT, "dark_amd", "csrc", "entropy.cpp"), "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitizer" in (build.stderr or "").lower() and "not found" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, "250"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "failures 0" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-4000:]
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++")
def test_every_thread_form_codes_the_same_bytes(tmp_path):
    """tools/ent_forms_fuzz.cpp: two threads, the four-stage and the five-stage pipeline against one thread on random distance streams of
    four flavours (tiny, huge with long unary extensions, mostly zero, mixed) -- the halves of the five-stage form must add up to the
    decision the one-thread model codes (src/model/dark.rs:180-214)."""
    exe = str(tmp_path / "ent_forms_fuzz")
    cmd = [CLANG, "-O2", "-std=c++17", "-march=x86-64-v3", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tools", "ent_forms_fuzz.cpp"), os.path.join(ROOT, "dark_amd", "csrc", "entropy.cpp"),
           os.path.join(ROOT, "dark_amd", "csrc", "bbb.cpp"), "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "bad: 0" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
    print(run.stdout[-400:])  # which forms really ran on this host (a form without an L3 group of enough cores falls back to a narrower one)
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++")
def test_thread_forms_under_thread_sanitizer(tmp_path):
    """the same tool under -fsanitize=thread (the pipeline's four lock-free rings): no data race report, same bytes in every form"""
    exe = str(tmp_path / "ent_forms_tsan")
    cmd = [CLANG, "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-march=x86-64-v3", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tools", "ent_forms_fuzz.cpp"), os.path.join(ROOT, "dark_amd", "csrc", "entropy.cpp"),
           os.path.join(ROOT, "dark_amd", "csrc", "bbb.cpp"), "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "tsan" in (build.stderr or "").lower():
        pytest.skip("thread sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"))
    assert "bad: 0" in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
    assert "ThreadSanitizer: data race" not in run.stderr, run.stderr[-6000:]
    shutil.rmtree(tmp_path, ignore_errors=True)
