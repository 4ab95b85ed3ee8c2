"""CPU: the C-ABI library loads, exports every symbol include/vilo_gpu.h declares, and refuses to run
without a GPU (no CPU fallback). No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vilo_[a-z0-9_]+)\s*\(", txt)))


def test_gpu_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "cerberus_amd", "lib", "libvilo_gpu.so"))
    names = _declared("vilo_gpu.h")
    assert len(names) >= 24
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_synth_library_exports():
    lib = C.CDLL(os.path.join(ROOT, "cerberus_amd", "lib", "libvilo_synth.so"))
    for n in _declared("vilo_synth.h"):
        assert hasattr(lib, n), n


def test_struct_sizes_match_header():
    from cerberus_amd import _ctypes as T
    assert C.sizeof(T.Sample) == 35 * 8
    assert C.sizeof(T.Preint) == (33 + 2 * 961) * 8
    assert C.sizeof(T.PreintImu) == (17 + 2 * 225) * 8
    assert C.sizeof(T.Config) == 8 * 18 + 8 + 8 * (16 + 3 + 9 + 2)
    assert C.sizeof(T.SolveSummary) == 16 + 16 + 2 * 64 * 8


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    from cerberus_amd import api, synth
    with pytest.raises(api.ViloError):
        api.Context(synth.default_config(), 0)


def test_product_does_not_reference_oracle():
    """The shipped package must not import / link anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cerberus_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle_py|liboracle|#include\s+\".*oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
