"""Repeated solves must repeat bit for bit (tools/soak.py, short): the whole-tree K3 launches hand update matrices and solutions from workgroup to
workgroup through flags in global memory, the trial launch carries the next linearisation -- a missing fence or a hand-over race would show
up as a different LM trace once in a while, alone or with eight host threads solving on the same device."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_repeated_solves_are_identical_alone_and_under_concurrency(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "80"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "every repeat identical" in r.stdout, r.stdout
