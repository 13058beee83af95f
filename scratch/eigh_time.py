import time, torch
torch.manual_seed(0)
for nb in (512, 64, 8):
    for s in (42, 64, 65, 66, 72, 96, 128):
        A = torch.randn(nb, s, s, dtype=torch.float64, device="cuda"); A = A @ A.transpose(1, 2)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e, U = torch.linalg.eigh(A); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # padded into a 128 x 128 block-diagonal problem
        P = torch.zeros(nb, 128, 128, dtype=torch.float64, device="cuda"); P[:, :s, :s] = A
        idx = torch.arange(s, 128, device="cuda"); P[:, idx, idx] = -1.0 - idx.double()
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e2, U2 = torch.linalg.eigh(P); torch.cuda.synchronize(); dp = time.perf_counter() - t0
        err = (e2[:, 128 - s:] - e).abs().max().item() / e.abs().max().item()
        print(f"batch {nb} size {s}: eigh {dt*1e3:.1f} ms, padded to 128 {dp*1e3:.1f} ms (rel err {err:.1e})")
