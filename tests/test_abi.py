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
    assert lib.t2s_abi_version() == 4      # round 4: t2s_taco_decoder grew (gate_part, w_pre2T); INTEGRATION.md lists what changed per version
    import ctypes
    from text2speech_amd.tacotron.tacotron import _DecoderStruct
    assert lib.t2s_sizeof_taco_decoder() == ctypes.sizeof(_DecoderStruct)       # the ctypes mirrors against the compiled structs
    from text2speech_amd.tacotron.autograd import _Bptt
    assert lib.t2s_sizeof_taco_bptt() == ctypes.sizeof(_Bptt)
    assert lib.t2s_plane_rows(2000, 128) == 2048 + 256
    assert lib.t2s_plane_rows(256, 0) == 256
    assert lib.t2s_padded_rows(1024) == 1024 and lib.t2s_padded_rows(130) == 256
    assert lib.t2s_error_string(0) == b"ok"
    assert lib.t2s_error_string(-1) == b"invalid argument"
    # gate-GEMM tile height by grid size: 256-row tiles above 128 workgroups, 128-row tiles (twice the fold slots) below
    assert lib.t2s_wg_gate_fold_slots(8, 512, 2000) == 8          # 4 x 8 x 8 = 256 workgroups of 256-row tiles
    assert lib.t2s_wg_gate_fold_slots(1, 512, 6400) == 16         # 4 x 25 = 100 workgroups -> 128-row tiles
    assert lib.t2s_wg_gate_fold_slots(1, 512, 32000) == 8
    assert lib.t2s_wg_gate_fold_slots(0, 512, 10) == -1


def test_argument_validation_without_gpu(lib):
    # null pointers / bad geometry are rejected before anything touches the device
    assert lib.t2s_wg_convinv(None, None, 1, 8, 0, 8, 10, None) == -1
    assert lib.t2s_small_logdet_inv(None, 4, 1.0, None, None, None) == -1
    assert lib.t2s_pack_conv_weight(None, None, 0, None, 4, 4, 1, 0, 0, 0, 256, 0, 32, None, None, None, 0, None) == -1
    assert lib.t2s_wg_melwin_planes(None, 1, 80, 10, 4, 256, None, None, None) == -1
    assert lib.t2s_wg_upsample_basis(None, None, 80, 1024, 256, 8, 0, 0, None, None, None) == -1
    assert lib.t2s_wg_compose_cond(None, None, 1024, 1024, 32, 320, 10241, None, None, None, None) == -1


def test_composed_conditioning_is_opt_in(monkeypatch):
    """Host logic of the inverse flow's composed-conditioning path: off unless T2S_COND_COMPOSE=1, and
    only for geometries where the upsampler's hop is a whole number of plane rows and the mel window fills whole 32-channel chunks."""
    from text2speech_amd import synth
    from text2speech_amd.glow import WaveGlow
    m = WaveGlow(**synth.WAVEGLOW_SMALL)
    eng = m._eng()
    monkeypatch.delenv("T2S_COND_COMPOSE", raising=False)
    assert eng.compose_geom() is None
    monkeypatch.setenv("T2S_COND_COMPOSE", "1")
    assert eng.compose_geom() == (32, 4, 320)          # hop 256 / n_group 8 phases, 1024 / 256 lags, 4 x 80 window channels
    monkeypatch.setenv("T2S_COND_COMPOSE", "0")
    assert eng.compose_geom() is None
