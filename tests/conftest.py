"""pytest configuration: `gpu` marker, import path of the product package and the oracle, shared fixtures."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")

# Kernel-family switches of the C library (read on every call): the per-kernel parity tests run each case once per family,
# because the product's default dispatch picks a family per shape by measured speed and would leave the others untested.
KERNEL_FAMILIES = {
    "default": {},
    "general": {"SSDSEG_NO_WRES": "1", "SSDSEG_DW_FWD": "lds", "SSDSEG_DW_BWD": "lds", "SSDSEG_CONV3_WGRAD": "taps",
                "SSDSEG_CONV3_NARROW": "0", "SSDSEG_PW_TILE": "0", "SSDSEG_CONV3_TILE": "0", "SSDSEG_PW_WGRAD": "0", "SSDSEG_DW_FWD_DEPTH": "1"},
    "general-reg": {"SSDSEG_NO_WRES": "1", "SSDSEG_DW_FWD": "lds", "SSDSEG_DW_BWD": "reg", "SSDSEG_PW_TILE": "0", "SSDSEG_PW_WGRAD": "0"},
    "resident-fused": {"SSDSEG_WRES_FORCE": "1", "SSDSEG_PW_FUSED": "1", "SSDSEG_PW_TILE": "0"},
    # round 2: the double-buffered tile GEMMs (pw_tile.h) for every shape they take, OCC-limited rowA instantiations for the rest;
    # the row-naming weight-gradient kernel (pw_wgrad.h) and the two-rows-ahead depthwise forward are on in "default",
    # "resident-fused" and "tile", off in the others
    "tile": {"SSDSEG_PW_TILE": "1", "SSDSEG_OCC_ROWS": "0"},
    # the tile GEMM's column-tile rule for layers below 65,536 rows (64 / 32 columns by shape in "tile"): every such layer on
    # 32-column tiles, and on one tile of <= 160 columns (the rule before the short-M table)
    "tile-32": {"SSDSEG_PW_TILE": "1", "SSDSEG_PWT_SMALL": "32"},
    "tile-wide": {"SSDSEG_PW_TILE": "1", "SSDSEG_PWT_SMALL": "0"},
}
_FAMILY_VARS = ("SSDSEG_NO_WRES", "SSDSEG_WRES_FORCE", "SSDSEG_PW_FUSED", "SSDSEG_DW_FWD", "SSDSEG_DW_BWD", "SSDSEG_CONV3_WGRAD",
                "SSDSEG_CONV3_NARROW", "SSDSEG_PW_TILE", "SSDSEG_CONV3_TILE", "SSDSEG_CONV3_WINOGRAD", "SSDSEG_OCC_ROWS", "SSDSEG_PW_WGRAD",
                "SSDSEG_DW_FWD_DEPTH", "SSDSEG_PWT_SMALL")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (fails loudly if the library or the GPU is missing)."""
    from ssdseglib import _hip
    c = _hip.Context(0)
    yield c
    c.sync()
    c.close()


@pytest.fixture()
def rng():
    return np.random.default_rng(1993)


@pytest.fixture(params=list(KERNEL_FAMILIES))
def kernel_family(request, monkeypatch):
    for k in _FAMILY_VARS:
        monkeypatch.delenv(k, raising=False)
    for k, v in KERNEL_FAMILIES[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param
