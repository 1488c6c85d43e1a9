"""The product path must not import, link or execute the oracle (it is test infrastructure)."""
import os
import re
import subprocess

from common import ROOT


def test_product_sources_do_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "surfelmeshing_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\boracle\b|smx_oracle|orc_", text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_library_does_not_link_the_oracle():
    so = os.path.join(ROOT, "surfelmeshing_amd", "libsmx.so")
    if not os.path.exists(so):
        import pytest
        pytest.skip("libsmx.so not built")
    out = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "orc_" not in syms


def test_tools_do_not_use_the_oracle():
    """Development tools outside tests/ stay on the product path; checker-side utilities live in tests/tools/."""
    offenders = []
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            if re.search(r"^\s*(import oracle|from oracle)|oracle_pipeline|ref_binding", text, flags=re.M):
                offenders.append(f)
    assert not offenders, offenders
