"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports every
symbol that include/t2s_hip.h declares; the ctypes table covers them all.  No compute calls."""
import os
import re

import pytest

from text2speech_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "t2s_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t2s_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libt2s_hip.so does not export %s" % n


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_pure_host_entry_points(lib):
    assert lib.t2s_abi_version() >= 1
    assert lib.t2s_plane_rows(2000, 128) == 2048 + 256
    assert lib.t2s_plane_rows(256, 0) == 256
    assert lib.t2s_padded_rows(1024) == 1024 and lib.t2s_padded_rows(130) == 256
    assert lib.t2s_error_string(0) == b"ok"
    assert lib.t2s_error_string(-1) == b"invalid argument"


def test_argument_validation_without_gpu(lib):
    # null pointers / bad geometry are rejected before anything touches the device
    assert lib.t2s_wg_convinv(None, None, 1, 8, 0, 8, 10, None) == -1
    assert lib.t2s_small_logdet_inv(None, 4, 1.0, None, None, None) == -1
    assert lib.t2s_pack_conv_weight(None, None, 0, None, 4, 4, 1, 0, 0, 0, 256, 0, 32, None, None, None, 0, None) == -1
