"""The C++ drop-in headers (include/gsplat_cuda/*.cuh) compile a host program written against the reference's
operator signatures; on the GPU box the program runs the reference's known-answer cases through them."""
import os
import subprocess

import pytest

from conftest import ROOT, pkg

EXE = os.path.join(ROOT, "tests", "cpp", "shim_test")


def _build():
    lib = pkg("_lib").build()
    src = os.path.join(ROOT, "tests", "cpp", "shim_test.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", "-I",
                               os.path.join(ROOT, "include"), src, "-x", "none", lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-o", EXE])
    return EXE


def test_host_written_against_reference_headers_compiles():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_shim_known_answers_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
