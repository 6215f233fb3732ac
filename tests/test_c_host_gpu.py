"""The C ABI from a plain C host: tests/c_host/hotpath_host.c (HIP runtime + include/retake_hip.h, no Python, no torch)
runs DPSelect and a PivotKV score + select through libretake_hip.so and checks them against the CPU oracle's C functions."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_runs_the_hot_path(tmp_path):
    from oracle import oracle as orc

    orc.build()
    lib = os.path.join(ROOT, "video-retake_amd", "retake", "_lib")
    obuild = os.path.join(ROOT, "oracle", "_build")
    exe = str(tmp_path / "hotpath_host")
    subprocess.check_call(["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_host", "hotpath_host.c"), "-o", exe,
                           "-L", lib, "-lretake_hip", "-L", obuild, "-lretake_oracle", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           f"-Wl,-rpath,{lib}", f"-Wl,-rpath,{obuild}", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "C_HOST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("bit-exact") >= 5
