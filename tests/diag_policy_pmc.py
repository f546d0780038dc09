"""diagnostic (not a test): a fixed number of policy forwards for counter collection:
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ... -- python tests/diag_policy_pmc.py N"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.policy import DevicePolicy, random_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pol = DevicePolicy("RMA_full", random_weights("RMA_full", 3))
obs = torch.randn((n, 22), device="cuda"); prev = torch.rand((n, 4), device="cuda"); out = torch.empty((n, 4), device="cuda")
for _ in range(60):
    pol.forward(obs, prev, out=out)
torch.cuda.synchronize()
print("done", n, pol.kernel)
