"""What arreau_model_create's calibration batch measures for several weight sets (GPU box): ARREAU_VERBOSE_CALIBRATION=1 python3 tools/exp/calib_probe.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from arreau_amd.checkpoint import make_synthetic_model
from tests.helpers import make_heavy_tailed
dev = torch.device("cuda", 0)
for tag, mk in (("default S=90", lambda: make_synthetic_model(S=90, seed=1234)), ("seed 7", lambda: make_synthetic_model(S=90, seed=7)),
                ("S=12", lambda: make_synthetic_model(S=12, seed=1234, num_timesteps=100)),
                ("heavy", lambda: make_heavy_tailed(make_synthetic_model(S=90, seed=1234), seed=5)),
                ("heavy df5 b10", lambda: make_heavy_tailed(make_synthetic_model(S=90, seed=1234), seed=5, df=5.0, boost=10.0)),
                ("df10 b1", lambda: make_heavy_tailed(make_synthetic_model(S=90, seed=1234), seed=6, df=10.0, boost=1.0))):
    m = mk().to(dev)
    print(tag, flush=True)
    eng = m.engine()
    print("   ", {k: v for k, v in eng.status().items() if "share" in k or k == "flags"}, flush=True)
    eng.close()
