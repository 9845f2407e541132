"""Builds tests/cpp_mirror_tests.cpp (the reference's unit tests written against include/dark.hpp, the C++ mirror of its Rust
interface) with g++ against the in-tree libdark_amd.so and runs it on the GPU."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_reference_unit_tests_through_cpp_mirror(tmp_path):
    exe = str(tmp_path / "cpp_mirror_tests")
    lib_dir = os.path.join(ROOT, "dark_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp_mirror_tests.cpp"),
                           "-L", lib_dir, "-ldark_amd", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, os.path.join(GOLDEN, "LICENSE.txt")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp mirror tests ok" in out.stdout
