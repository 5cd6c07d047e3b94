"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports exactly what the header
declares.  No compute calls (no GPU here)."""
import ctypes
import shutil
import subprocess

import pytest

from epilogos_amd import _abi, build


@pytest.fixture(scope="module")
def lib():
    if build.is_stale():
        if shutil.which("hipcc") is None:
            pytest.skip("hipcc not available and library not prebuilt")
        build.build_library()
    return _abi.load()


def test_header_and_binding_agree():
    hdr = _abi.header_symbols()
    assert hdr, "no prototypes parsed from include/epilogos_amd.h"
    assert sorted(_abi.PROTOTYPES) == hdr


def test_library_exports_every_header_symbol(lib):
    for name in _abi.header_symbols():
        assert hasattr(lib, name), "libepilogos_hip.so does not export " + name


def test_exports_are_c_abi(lib):
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    out = subprocess.run([nm, "-D", "--defined-only", str(_abi.lib_path())], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in _abi.header_symbols():
        assert name in exported          # unmangled => extern "C"


def test_version_and_argument_validation_without_gpu(lib):
    assert lib.epg_version() == _abi.ABI_VERSION == 2          # (the header's EPG_ABI_VERSION; bumped with every change of the ABI)
    # pure argument validation happens before any HIP call
    rc = lib.epg_bin_hist(None, -1, 10, 10, 18, None, None, None)
    assert rc == -1 and b"bad shape" in lib.epg_last_error()
    rc = lib.epg_bin_hist(None, 10, 10, 5, 18, None, None, None)       # ldx < N
    assert rc == -1
    rc = lib.epg_bin_hist(None, 10, 10, 16, 128, None, None, None)     # S > 127: states are int8
    assert rc == -2
    rc = lib.epg_bin_hist(None, 10, 10, 16, 40, None, None, None)      # a wide model is taken; X is NULL
    assert rc == -1
    assert lib.epg_ws_bytes(4, 10, 10, 18) == -1
    assert lib.epg_ws_bytes(1, 0, 833, 18) >= 834 * 18 * 12


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "lib_path", lambda: tmp_path / "nope.so")
    with pytest.raises(_abi.EpilogosHipError):
        _abi.load()


def test_engine_refuses_to_run_without_gpu():
    import torch
    from epilogos_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_abi.EpilogosHipError):
        engine.require_gpu()


def test_io_library_exports_its_header():
    import re
    from pathlib import Path
    from epilogos_amd import _io
    if build.io_is_stale():
        build.build_io_library()
    lib = _io.load()
    hdr = (Path(__file__).resolve().parents[1] / "include" / "epilogos_io.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(epgio_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n


def test_console_script_of_pyproject_resolves():
    """pyproject.toml declares the reference's `epilogos` command (setup.py:28-33) as epilogos_amd.run:cli."""
    import importlib
    from pathlib import Path

    import tomli
    meta = tomli.loads((Path(__file__).resolve().parents[1] / "pyproject.toml").read_text())
    target = meta["project"]["scripts"]["epilogos"]
    mod, fn = target.split(":")
    assert callable(getattr(importlib.import_module(mod), fn))
    assert meta["tool"]["setuptools"]["dynamic"]["version"]["attr"] == "epilogos_amd.__version__"
