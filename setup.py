"""Packaging shim for setuptools older than 61 (Ubuntu 22.04 ships 59.6, which ignores pyproject.toml's [project] table):
the same metadata as pyproject.toml.  Reference: setup.py:14-35 (the `epilogos` console script)."""
import re
from pathlib import Path

from setuptools import setup

version = re.search(r'__version__ = "([^"]+)"', (Path(__file__).parent / "epilogos_amd" / "__init__.py").read_text()).group(1)

setup(
    name="epilogos-amd",
    version=version,
    description="epilogos scoring hot path (S1/S2/S3 saliency, paired mode) on AMD MI355X: hand-written HIP kernels behind a C ABI",
    packages=["epilogos_amd"],
    package_data={"epilogos_amd": ["csrc/*", "_lib/*.so"]},
    python_requires=">=3.10",
    install_requires=["numpy", "pandas", "scipy", "click", "torch"],
    entry_points={"console_scripts": ["epilogos = epilogos_amd.run:cli"]},
)
