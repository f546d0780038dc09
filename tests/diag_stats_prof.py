"""diagnostic (not a test): the column-statistics pass over one BASELINE fragment's observations, for rocprofv3
(--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE in separate runs)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.custom_logging import BatchStatistics
x = torch.randn((1024, 4096, 22), device="cuda")
st = BatchStatistics()
for _ in range(40):
    st.column_stats_tensor(x)
torch.cuda.synchronize()
