"""C++ host layer (include/rten_hip_ops.hpp): the reference's Operator interface above the C ABI.

The test program tests/cpp/test_host_ops.cpp checks the reference's error messages on the host and, on a GPU, every
operator's output bits against the CPU oracle (linked directly as a C library: the oracle is test infrastructure).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_ops")


def build_binary():
    from oracle import ref
    from rten_amd import lib as L
    ref.build()
    L.load()  # raises if librten_hip.so is missing: the C++ layer has no other backend
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_host_ops.cpp")
    deps = [src, os.path.join(ROOT, "include", "rten_hip_ops.hpp"), os.path.join(ROOT, "include", "rten_hip.h")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L" + os.path.join(ROOT, "rten_amd"), "-lrten_hip", "-L" + os.path.join(ROOT, "oracle", "_build"), "-lrten_oracle",
                               "-Wl,-rpath,$ORIGIN/../../../rten_amd", "-Wl,-rpath,$ORIGIN/../../../oracle/_build",
                               "-Wl,-rpath," + os.path.join(ROOT, "rten_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build"),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_cpp_host_layer_validation_messages():
    out = subprocess.run([build_binary(), "--host-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_layer_operators_bit_exact():
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
