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
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    from defslam_amd import sft
    ctx = sft.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def lab_ctx():
    """A context of the lab build (libdefslam_hip_lab.so: the product ABI + include/defslam_hip_debug.h)."""
    from defslam_amd import sft
    ctx = sft.Context(0, lab=True)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def host_ctx():
    from defslam_amd import sft
    ctx = sft.Context(-1)
    yield ctx
    ctx.close()


def oracle_args(oracle, tmpl, fr, regs=None):
    from defslam_amd import synth
    tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
    regs = regs or (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    return tc, (tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz) + tuple(regs)
