"""Packaging for gym_reinmav_amd: builds librmav.so with hipcc (gfx950) in-tree and installs the package.

    pip install -e reinmav-gym_amd        # or: python setup.py build_ext --inplace

The library is a plain C-ABI shared object loaded with ctypes (no Python extension module), so "building the
extension" is one hipcc invocation driven by the Makefile next to this file."""
import os
import subprocess

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _make():
    subprocess.run(["make", "-C", HERE], check=True)


class BuildExt(Command):
    description = "compile librmav.so for gfx950 (hipcc)"
    user_options = [("inplace", "i", "kept for compatibility; the library is always built in-tree")]

    def initialize_options(self):
        self.inplace = 1

    def finalize_options(self):
        pass

    def run(self):
        _make()


class BuildPy(build_py):
    def run(self):
        _make()
        super().run()


setup(
    name="gym_reinmav_amd",
    version="0.1.0",
    description="MI355X-native batched drop-in for reinmav-gym's native quadrotor environments",
    packages=find_packages(HERE),
    package_dir={"": "."},
    package_data={"gym_reinmav_amd": ["librmav.so"]},
    python_requires=">=3.9",
    install_requires=["numpy"],
    extras_require={"torch": ["torch"]},
    cmdclass={"build_ext": BuildExt, "build_py": BuildPy},
)
