"""The C-ABI library loads and exports every symbol include/sessd_hip.h declares (no compute)."""
import os
import re

from conftest import ROOT


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "sessd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sessd_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    import sessd_hip
    names = _header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(sessd_hip.lib, n), "libsessd_hip.so does not export %s" % n
        assert n in sessd_hip.SIGNATURES, "sessd_hip/_lib.py has no binding for %s" % n
    for n in sessd_hip.SIGNATURES:
        assert n in names, "%s is bound but not declared in include/sessd_hip.h" % n
    assert b"gfx950" in sessd_hip.lib.sessd_version()


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "se-ssd_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from sessd_hip import ops
    with pytest.raises(ValueError):
        ops.boxes_pairwise(1, torch.zeros(2, 5), torch.zeros(2, 5))
