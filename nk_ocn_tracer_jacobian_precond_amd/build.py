"""In-tree build of every native piece (gfx950 cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)


def _make(directory, *args):
    subprocess.run(["make", "-C", directory, *args], check=True)


def build_all(jobs=4, oracle=True):
    _make(os.path.join(_PKG, "host"))
    _make(os.path.join(_PKG, "csrc"), f"-j{jobs}")
    os.makedirs(os.path.join(_PKG, "bin"), exist_ok=True)
    _make(os.path.join(_PKG, "cli"))
    if oracle and os.path.isdir(os.path.join(_ROOT, "oracle")):
        _make(os.path.join(_ROOT, "oracle"))


if __name__ == "__main__":
    build_all()
