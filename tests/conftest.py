import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def pair4k():
    """~4k-point HDL-64E pair (64 azimuth steps) with reference normals from the product filter."""
    from laser_slam_amd import synth
    ref, rd, T_true, T_init = synth.scan_pair(64)
    return dict(ref=ref, rd=rd, T_true=T_true, T_init=T_init)


@pytest.fixture(scope="session")
def pair64k():
    from laser_slam_amd import synth
    ref, rd, T_true, T_init = synth.scan_pair(1024)
    return dict(ref=ref, rd=rd, T_true=T_true, T_init=T_init)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
