"""The multiply-shift divider of the kernels' prologues (rten_make_div / rten_div in rten_amd/csrc/internal.h), host side: the same two functions compiled
by hipcc for the host and compared with `/` over 26 M (n, d) pairs -- every divisor up to 4096, the ends of every range the launchers ask for.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_multiply_shift_division_is_exact_on_the_promised_range():
    src = os.path.join(ROOT, "tests", "cpp", "test_div.hip")
    out = os.path.join(ROOT, "tests", "cpp", "_build", "test_div")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", src, "-o", out])
    r = subprocess.run([out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout + r.stderr
