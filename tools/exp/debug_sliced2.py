import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_parity as T
from arreau_amd.checkpoint import make_synthetic_model
dev = torch.device("cuda", 0)
m = make_synthetic_model(S=90, seed=1234).to(dev)
if os.environ.get("ORACLE"):
    from tests.helpers import oracle_from_module
    oracle_from_module(m, torch.float32)
try:
    T.test_sliced_multi_stream_execution_is_bitwise_the_whole_batch(dev, (m, None), 2)
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:200])
