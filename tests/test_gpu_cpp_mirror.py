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


def _build_small(tmp):
    exe = os.path.join(tmp, "small_gemm_bench")
    lib = os.path.join(ROOT, "laser_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "cpp", "small_gemm_bench.cpp"), "-o", exe, "-L", lib,
                    "-llaser_hip", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"],
                   check=True)
    return exe


def test_small_gemm_bench_compiles_and_links(tmp_path):
    _build_small(str(tmp_path))


@pytest.mark.gpu
def test_small_gemm_from_a_compiled_caller(tmp_path):
    """BASELINE configs[0] (fp32 128^3) through the C-ABI from compiled code: bit-exact vs an fmaf chain, and the
    small-matrix path is not slower than the tiled kernels + staged copies it replaces (timings are printed for the
    record: profiles/r02/small_gemm.jsonl)."""
    import json
    exe = _build_small(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["mismatches"] == 0
    print(r.stdout)
    assert d["host_us"] <= d["tiled_kernels"]["host_us"] * 1.1, d      # zero-copy staging vs three blocking copies
