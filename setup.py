"""pip install -e .   (the reference installs the same way: setup.py:1-6, name gym_lmaze, one dependency `gym`).

Installs the alias package `gym_lmaze` (the reference's import name and env ids) and the MI355X-native
implementation under the import name `gym_lmaze_amd` (the source directory is `gym-lmaze_amd/`; a hyphen cannot be
an import name, so a checkout reaches it with importlib.import_module("gym-lmaze_amd") and an installed copy by its
underscore name -- gym_lmaze/__init__.py tries both).  liblmaze_hip.so is built with hipcc for gfx950 before the
files are collected (`make -C gym-lmaze_amd/csrc`), and travels as package data together with the kernel sources and
the C header.  gym / gymnasium stay optional (gym-lmaze_amd/compat.py has a built-in registry), torch and numpy are
expected from the ROCm image and are not pulled from an index.
"""
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
IMPL = os.path.join(ROOT, "gym-lmaze_amd")


class build_py_with_hip(build_py):
    """Build liblmaze_hip.so (hipcc cross-compiles gfx950 without a GPU) and copy include/lmaze.h next to it."""

    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(IMPL, "csrc"), "-j4"])
        shutil.copyfile(os.path.join(ROOT, "include", "lmaze.h"), os.path.join(IMPL, "lmaze.h"))
        build_py.run(self)


setup(
    name="gym_lmaze",
    version="0.2.0",
    description="MI355X-native batched L-maze step path; drop-in for gkm2708/gym-lmaze (ids lmaze-v0 ... lmaze-v6)",
    packages=["gym_lmaze", "gym_lmaze.envs", "gym_lmaze_amd"],
    package_dir={"gym_lmaze": "gym_lmaze", "gym_lmaze_amd": "gym-lmaze_amd"},
    package_data={"gym_lmaze_amd": ["liblmaze_hip.so", "lmaze.h", "csrc/*.hip", "csrc/*.h", "csrc/Makefile"]},
    python_requires=">=3.8",
    install_requires=[],          # numpy + torch (ROCm build) come with the image; gym / gymnasium optional
    extras_require={"gym": ["gym"]},
    cmdclass={"build_py": build_py_with_hip},
    zip_safe=False,
)
