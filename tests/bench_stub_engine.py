"""Stand-in for ``nutpie_amd._lib`` used ONLY by tests/test_bench_cpu.py to exercise bench.py's multi-rank plumbing on a machine
without a GPU (``bench.py --engine-stub``; torch.distributed over gloo).  It samples nothing: chains "advance" by a fixed number
of leapfrogs per launch and their "draws" are a function of the GLOBAL chain id, so that the test can tell whether shards,
offsets and the gather's order are right.  The line bench.py prints with it is marked ``"data": "stub"``."""
import numpy as np

LEAPFROGS_PER_DRAW = 4


def lib():
    return None


def default_evals_per_launch(dim):
    return 8


class PyNutsSettings:
    def __init__(self, seed):
        self.seed = seed
        self.num_tune, self.num_draws, self.num_chains = 400, 1000, 6

    @staticmethod
    def Diag(seed=None):
        return PyNutsSettings(seed)

    def update(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class TridiagGaussianModel:
    def __init__(self, diag, offdiag=None, mu=None):
        self.dim = len(diag)


class _Progress:
    def __init__(self, steps, draws, tune):
        self.total_num_steps = int(steps)
        self.finished_draws = int(draws)
        self.tuning = draws < tune


class PySampler:
    def __init__(self, settings, model, *, device=0, waves_per_chain=0, chain_offset=0, n_local_chains=0, store_draws=True,
                 evals_per_launch=0, manual=False, **_kw):
        self.s, self.dim = settings, model.dim
        self.offset = int(chain_offset)
        self.n = int(n_local_chains) or int(settings.num_chains)
        assert self.offset + self.n <= int(settings.num_chains), "shard outside the job (the engine rejects this too)"
        self.E = int(evals_per_launch) or default_evals_per_launch(self.dim)
        self.total_draws = int(settings.num_tune) + int(settings.num_draws)
        self.steps = np.zeros(self.n, dtype=np.int64)
        self.waves_per_chain = int(waves_per_chain) or (1 if self.dim <= 1024 else 4)
        self.launches = 0

    def _draws(self):
        return np.minimum(self.steps // LEAPFROGS_PER_DRAW, self.total_draws)

    def step(self, n_launches=1):
        done_launches = 0
        for _ in range(int(n_launches)):
            if (self._draws() >= self.total_draws).all():
                break
            self.steps += self.E
            done_launches += 1
        self.launches += done_launches
        return bool((self._draws() >= self.total_draws).all()), done_launches, 0.5 * done_launches

    def wait(self):
        while not self.step(64)[0]:
            pass

    def progress(self):
        return [_Progress(min(s, self.total_draws * LEAPFROGS_PER_DRAW), d, int(self.s.num_tune)) for s, d in zip(self.steps, self._draws())]

    def device_ptr(self, name):
        return 1

    def _copy(self, name, dtype, vec=False):
        T = self.total_draws
        gid = (self.offset + np.arange(self.n))[:, None]
        t = np.arange(T)[None, :]
        if vec:
            return (gid[:, :, None] + 1e-3 * t[:, :, None] + np.zeros((1, 1, self.dim))).astype(dtype)
        if name == "tuning":
            return np.broadcast_to(t < int(self.s.num_tune), (self.n, T)).astype(dtype)
        if name == "n_steps":
            return np.full((self.n, T), LEAPFROGS_PER_DRAW, dtype=dtype)
        return np.broadcast_to(gid, (self.n, T)).astype(dtype)

    def close(self):
        pass
