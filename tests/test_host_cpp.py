"""The C++ host mirror (go-blosc_amd/host/blosc.hpp) compiles against the C ABI and behaves like the reference API.
CPU run: header helpers + loud failure without a device.  GPU run: round trips."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "go-blosc_amd", "lib")
EXE = os.path.join(ROOT, "tests", "tools", "host_mirror_check")


def _build():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(LIBDIR, "libhipblosc.so")):
        g.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", EXE, os.path.join(ROOT, "tests", "tools", "host_mirror_check.cpp"),
                           "-L" + LIBDIR, "-lhipblosc", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])


def test_host_mirror_cpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout


@pytest.mark.gpu
def test_host_mirror_gpu(hb):
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device round trips" in out.stdout
