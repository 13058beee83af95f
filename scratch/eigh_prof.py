import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nutpie_amd import _lib as hip
torch.manual_seed(0)
for s in (64, 128):
    A = torch.randn(4, s, s, dtype=torch.float64, device="cuda"); A = A @ A.transpose(1, 2)
    w, V = hip.batched_eigh(A, _stage=3)
    c = w[0, :7].cpu().numpy()
    print(f"order {s}: cycles load+tridiagonal {c[0]:.0f}, form Q {c[1]:.0f}, QL scalar {c[2]:.0f}, QL apply {c[3]:.0f}, rotations {c[4]:.0f}, sweeps {c[5]:.0f}, total {c[6]:.0f}  ->  {c[2] / max(c[4], 1):.0f} + {c[3] / max(c[4], 1):.0f} cycles per rotation")
