"""Single-rank RCCL self test of gpmi_comm_* (world = 1) and a 2-process gloo-bootstrapped run on ONE GPU is not
possible (RCCL rejects duplicate devices), so the N>1 gather is covered by this test + the CPU gloo tests."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
from inference_amd._engine import GpEngine
from inference_amd import sharding
eng = GpEngine(np.random.rand(64, 2), np.random.rand(64))
sharding.init_device_comm(eng)
out = eng.comm_allgather(np.array([1.5, 2.5, 3.5, 4.5]))
print("rccl allgather world=1:", out)
assert out.shape == (1, 4) and np.array_equal(out[0], [1.5, 2.5, 3.5, 4.5])
print("OK")
