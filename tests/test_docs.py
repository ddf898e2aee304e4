"""Guards for the hand-maintained documents (a doc-update substitution once blew BASELINE.md up to 2.9 MB)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baseline_md_is_a_document():
    path = os.path.join(ROOT, "BASELINE.md")
    assert os.path.getsize(path) < 64 * 1024, "BASELINE.md has grown beyond 64 KB: a substitution went wrong"
    text = open(path, encoding="utf-8").read()
    assert text.startswith("# BASELINE")
    heads = re.findall(r"^## (\d)\. ", text, flags=re.M)
    assert heads[:6] == ["1", "2", "3", "4", "5", "6"], heads
    assert len(text.splitlines()) < 400


def test_design_and_integration_present():
    for name in ("DESIGN.md", "INTEGRATION.md", "SURVEY.md"):
        path = os.path.join(ROOT, name)
        assert 1024 < os.path.getsize(path) < 256 * 1024, name
