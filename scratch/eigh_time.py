"""Batched symmetric eigendecomposition, 512 / 64 / 8 problems of order 42..128: torch.linalg.eigh (rocSOLVER), the same embedded in an
order-128 problem, and the engine's own routine (nphip_batched_eigh, nutpie_amd/csrc/linalg.hip)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nutpie_amd import _lib as hip
hip.lib()
torch.manual_seed(0)
for nb in (512, 64, 8):
    for s in (16, 42, 64, 66, 96, 128):
        A = torch.randn(nb, s, s, dtype=torch.float64, device="cuda"); A = A @ A.transpose(1, 2)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e, U = torch.linalg.eigh(A); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        P = torch.zeros(nb, 128, 128, dtype=torch.float64, device="cuda"); P[:, :s, :s] = A
        idx = torch.arange(s, 128, device="cuda"); P[:, idx, idx] = -1.0 - idx.double()
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e2, U2 = torch.linalg.eigh(P); torch.cuda.synchronize(); dp = time.perf_counter() - t0
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e3, U3 = hip.batched_eigh(A); torch.cuda.synchronize(); dn = time.perf_counter() - t0
        err = (e3 - e).abs().max().item() / e.abs().max().item()
        res = (A @ U3 - U3 * e3[:, None, :]).abs().max().item() / A.abs().max().item()
        print(f"batch {nb} order {s}: rocSOLVER {dt*1e3:.1f} ms, embedded in order 128 {dp*1e3:.1f} ms, nphip_batched_eigh {dn*1e3:.2f} ms (eigenvalues within {err:.1e}, residual {res:.1e})")
