"""csrc/gmm_model.cpp (text format, number conversions, every packer) under AddressSanitizer + UBSan, and the threaded
packers under ThreadSanitizer: tests/host/host_checks.cpp, built here with g++ (host code only, no GPU, no HIP runtime)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speaker-recognition_amd", "csrc")


def _build_and_run(tmp_path, flags, args, env=None):
    exe = str(tmp_path / "host_checks")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", *flags, "-I", CSRC, "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "host", "host_checks.cpp"), os.path.join(CSRC, "gmm_model.cpp"),
           "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0 and "host checks ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_model_host_code_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], [],
                   env={"ASAN_OPTIONS": "detect_leaks=1"})


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_packers_under_tsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=thread"], ["threads"])
