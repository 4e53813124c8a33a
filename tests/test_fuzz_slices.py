"""A seeded slice of every randomised sweep, inside `pytest -m gpu` (round-5 review: the fuzzers were scripts the driver never
saw).  Each runs as a child process -- the scripts draw models, batches, launch options and math modes per case, compare with
the oracle (tests/fuzz_*.py) or a float64 product (tools/fuzz_gemm.py) and exit non-zero on the first mismatch.  40 cases
each, fixed seeds: the long soaks stay manual (`python tests/fuzz_layerwise.py 1000 <seed>`)."""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from gnnbuilder_amd import runtime

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CASES = 40


@pytest.fixture(scope="module")
def dev():
    runtime.load_library(require_gpu=True)  # fails loudly: no fallback
    return torch.device("cuda:0")


@pytest.mark.parametrize("script,extra", [("tests/fuzz_fused.py", ()), ("tests/fuzz_fused.py", ("zf",)), ("tests/fuzz_layerwise.py", ()),
                                          ("tests/fuzz_layerwise.py", ("math",)), ("tools/fuzz_gemm.py", ())],
                         ids=["fused_stacks", "gcn2_zf", "layerwise", "layerwise_math_modes", "large_k_gemm"])
def test_seeded_slice_of_the_randomised_sweeps(dev, script, extra):
    seed = 600 + sum(map(ord, script + "".join(extra))) % 97
    proc = subprocess.run([sys.executable, str(ROOT / script), str(CASES), str(seed), *extra], capture_output=True, text=True, timeout=900,
                          cwd=str(ROOT))
    tail = "\n".join((proc.stdout + proc.stderr).splitlines()[-12:])
    assert proc.returncode == 0, tail
    assert f"{CASES} cases" in proc.stdout.splitlines()[-1], tail
