"""Build and run the C++ host-mirror KAT program (tests/cpp/kat_host_mirror.cpp) against
liblaser_hip.so: the compiled-caller view of the drop-in boundary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    exe = os.path.join(tmp, "kat_host_mirror")
    lib = os.path.join(ROOT, "laser_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "kat_host_mirror.cpp"), "-o", exe, "-L", lib,
                    "-llaser_hip", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"],
                   check=True)
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    _build(str(tmp_path))


@pytest.mark.gpu
def test_cpp_mirror_kats(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SUCCESS" in r.stdout, r.stdout + r.stderr
