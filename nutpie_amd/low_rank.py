"""``adaptation="low_rank"`` — a low-rank modified mass matrix on the HIP engine.

What the reference does (``PyNutsSettings::LowRank``, src/wrapper.rs:307-334, 725-729; python/nutpie/sample.py:921-933;
docs/sampling-options.qmd:124-144): nuts-rs adapts a mass matrix ``M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2`` — a diagonal
scaling plus a rank-k correction whose eigenvalues are those of the geometric mean of the draw covariance and the inverse
gradient covariance that lie outside ``[1/cutoff, cutoff]`` (``mass_matrix_eigval_cutoff``), regularised by
``mass_matrix_gamma`` — and integrates, draws momenta and tests U-turns under it.

Here (round 3) the metric lives where the reference has it, INSIDE the sampler: with the setting ``low_rank_metric`` the engine's
kernels apply ``v = M^-1 p`` in the leapfrog, draw ``p ~ N(0, M)`` and carry the velocity with every tree state
(``kernels.hip``: ``lf1`` / ``lf2`` / ``sample_momentum_lr`` / ``turning``; round 4: also the register-resident leaf, ``Machine<..., LR>``
— fused models up to D = 4096 and compiled densities; the oracle restates the same arithmetic and the two are
compared bit for bit, tests/test_gpu_low_rank.py) — for EVERY model flavour: fused, raw C callbacks, BridgeStan, device
callbacks, runtime-compiled densities.  The DRIVER of the adaptation is here: chains stop between two draws at the window
boundaries (``nphip_settings_set_pause_draws``), every stopped chain's ``(sigma^2, V, lambda)`` is estimated from the window's
draws and gradients — round 6: by ONE kernel below the C-ABI (``nphip_low_rank_estimate``, nutpie_amd/csrc/lowrank_est.hip: a
workgroup per chain, straight out of the engine's trace) for models of up to 256 dimensions; above that by :func:`estimate`,
the torch formulation the kernel restates (batched eigendecompositions on the engine's ``nphip_batched_eigh`` up to order 64,
rocSOLVER above) —, on a worker thread while the other chains run, and handed to the engine (``nphip_sampler_set_metric``);
the chains go on from where they are, through a new step-size search, as nuts-rs does when the mass matrix changes.  A chain
that needs no low-rank part and still adapts its own diagonal is released unchanged (``nphip_sampler_release``).  The estimator follows the published description (Seyboldt et al., "Preconditioning Hamiltonian Monte
Carlo by minimizing Fisher divergence"); the crate is not in the tree — PARITY UNPINNED, like the rest of the sampler.

(Round 2 obtained the same sampler as a linear re-parametrisation around the density — ``y = L^-1 (x - m)`` with the engine's
diag-NUTS on ``y``; that needed a batched torch density, two contractions per evaluation and a rewrite of the trace at every
switch.  The :class:`Transform` below is what is left of it: the algebra the estimator's tests are written against.)
"""

from __future__ import annotations

import contextlib
import threading
import time
from dataclasses import dataclass

import numpy as np

K_MAX = 16            # columns kept per chain (the reference has no such knob)
BASIS_DRAWS = None    # draws (thinned evenly from the window) that span the subspace of the low-rank part; the DIAGONAL part uses every
                      # draw.  None: 32 (with their gradients 64 directions: the order the engine's own batched eigensolver is fastest at,
                      # 1.8 ms per 512 problems) for models of up to 256 dimensions, 64 (order 128: 12 ms) above — measured on the D = 500
                      # demo: 0.31 relative error of the posterior sd with 32, 0.09 with 64 (profiles/r5_low_rank_schedule.txt)
WINDOW_MAX = 256      # draws of a foreground window that are read at all (the most recent ones)
MIN_WINDOW = 12       # a window shorter than this says too little: no hand-in
SETTLE = 4            # draws after a hand-in that are not used (the chain re-runs its step-size search there)
HOLD_LAUNCHES = 3     # looks (a launch, or a few ms of launches) a stopped chain waits for company before it is handed in on its own
ESTIMATOR_CAP = None   # workgroups of the estimator kernel beside a running engine kernel: None = the CUs that kernel leaves idle, 0 = one per chain
SHORT_LAUNCH_DIV = 4  # launches while chains can still stop at a boundary: the default length divided by this (the fast driver)
MAX_HAND_INS = 15     # pause draws the engine holds (engine_types.h: pause_draws[16])


def window_schedule(num_tune: int, early_window: float = 0.3, step_size_window: float = 0.15, switch_freq: int = 80, early_switch_freq: int = 10,
                    update_freq: int = 10, early_hand_ins: bool = True):
    """The reference's warm-up windows for ``LowRank`` settings — the SAME foreground / background schedule as ``diag``
    (``src/wrapper.rs:198-240``: ``window_switch_freq`` / ``mass_matrix_switch_freq``, ``early_window_switch_freq`` are set on Diag AND
    LowRank settings; nuts-rs' ``GlobalStrategy::adapt``): the background estimator becomes the foreground one every
    ``early_switch_freq`` draws during the first ``early_window`` of warm-up and every ``switch_freq`` draws afterwards, as long as a
    whole background window still fits before the final step-size window; the metric is refreshed from the foreground — the draws
    since the SECOND-LAST switch — every ``update_freq`` draws (10 for low-rank settings, recalled) and stays fixed for the last
    ``step_size_window`` of warm-up.

    Returns ``[(pause, start)]``: the chain stops after ``pause`` finished draws and its metric is estimated from ``draws[start:pause]``.
    What differs from the reference, and why: (1) a hand-in stops the chain until the host has estimated (an engine launch or two),
    so the refreshes are THINNED — during the early windows the engine's own diagonal adaptation runs (same windows, on the device,
    free) and the first low-rank metric is handed in at the first switch of the main phase, i.e. from a window that starts after
    the early phase and after the step size has settled; from there every switch, refreshes no closer than ``switch_freq / 2`` draws,
    and the last refresh before the final window — at most MAX_HAND_INS; (2) the draws right after a hand-in (SETTLE) are not used."""
    T = int(num_tune)
    switch_freq, early_switch_freq, update_freq = max(1, int(switch_freq)), max(1, int(early_switch_freq)), max(1, int(update_freq))
    early_end = int(np.ceil(T * early_window))
    final = max(0, T - int(np.ceil(T * step_size_window))) + 1          # the engine's bound: draws d < final adapt the metric
    final = min(final, T)
    switches, bg = [], 1                                                 # (the estimators see the initial point)
    for d in range(final):
        freq = early_switch_freq if d < early_end else switch_freq
        bg += 1
        if bg >= freq and not (freq + d > final):
            switches.append(d + 1)
            bg = 0
    late = [c for c in switches if c > early_end]
    out, last = [], None

    def window_start(p):
        before = [c for c in switches if c <= p]
        return before[-2] if len(before) >= 2 else 0

    cands = sorted(set(late) | {c for c in range(update_freq, final, update_freq) if late and c > late[0]} | ({final - 1} if final - 1 > early_end else set()))
    early_pick = None
    if early_hand_ins:
        # the early phase: the reference refreshes the metric there too, every few draws from the 10 - 20 draws since the second-last
        # early switch.  Here ONE hand-in, at the last early switch, from the second half of the early phase (measured, profiles/
        # r5_low_rank_schedule.txt: hand-ins from 16 - 20 draws at 19 / 39 / 79 / 119 cost radon 2 x the engine time and its warm-up
        # 40 % more leapfrogs; none at all leaves a hard target — D = 500, six directions of variance x 400 — in transit when
        # the main phase starts, 0.42 relative error of the posterior sd against 0.25)
        early = [c for c in switches if c <= early_end]
        if early and early[-1] >= 4 * MIN_WINDOW:
            early_pick = early[-1]
            cands = sorted(set(cands) | {early_pick})
    for p in cands:
        if p <= 0 or p >= T:
            continue
        start = window_start(p)
        if p == early_pick:
            start = p // 2
        if last is not None and last <= start < last + SETTLE:
            start = min(last + SETTLE, p - MIN_WINDOW)      # (a window that begins right at a hand-in skips the draws of its step-size search)
        if p - start < MIN_WINDOW:
            continue
        is_switch, is_last = (p in late) or p == early_pick, p == cands[-1]
        if last is not None and not is_switch and not is_last and p - last < max(switch_freq // 2, update_freq):
            continue
        if is_last and last is not None and p - last < update_freq:
            continue
        out.append((p, start))
        last = p
    if len(out) > MAX_HAND_INS:
        keep = np.unique(np.round(np.linspace(0, len(out) - 1, MAX_HAND_INS)).astype(int))
        out = [out[i] for i in keep]
    if not out and T >= 2 * MIN_WINDOW:
        # a warm-up too short for a main-phase switch: one hand-in before the final window, from everything after the early phase
        p = max(MIN_WINDOW, final - 1)
        if p < T:
            out = [(p, max(0, min(early_end, p - MIN_WINDOW)))]
    return out


@dataclass
class Transform:
    """x = mean + stds * (y + V (d * (V' y))), batched over chains; ``d = sqrt(lambda) - 1`` (0 = column unused)."""

    mean: "object"   # [n, D]
    stds: "object"   # [n, D]
    V: "object"      # [n, D, k]
    d: "object"      # [n, k]

    @staticmethod
    def identity(n, D, device):
        import torch

        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=device)  # noqa: E731
        return Transform(z(n, D), torch.ones(n, D, dtype=torch.float64, device=device), z(n, D, K_MAX), z(n, K_MAX))

    def _apply(self, v, coef):
        # v + V (coef * (V' v)), v: [n, (m,) D]
        import torch

        if v.dim() == 2:
            t = torch.einsum("ndk,nd->nk", self.V, v) * coef
            return v + torch.einsum("ndk,nk->nd", self.V, t)
        t = torch.einsum("ndk,nmd->nmk", self.V, v) * coef[:, None, :]
        return v + torch.einsum("ndk,nmk->nmd", self.V, t)

    def _b(self, a, like):
        return a if like.dim() == 2 else a[:, None, :]

    def forward(self, y):
        return self._b(self.mean, y) + self._b(self.stds, y) * self._apply(y, self.d)

    def inverse(self, x):
        dinv = 1.0 / (1.0 + self.d) - 1.0
        return self._apply((x - self._b(self.mean, x)) / self._b(self.stds, x), dinv)

    def grad_to_y(self, gx):   # L' gx
        return self._apply(self._b(self.stds, gx) * gx, self.d)

    def grad_to_x(self, gy):   # L'^-1 gy
        dinv = 1.0 / (1.0 + self.d) - 1.0
        return self._apply(gy, dinv) / self._b(self.stds, gy)

    def eigenvalues(self):
        return (1.0 + self.d) ** 2


NATIVE_EIGH_MAX = 64   # measured on MI355X, 512 problems (scratch/eigh_time.py, profiles/r4_low_rank_register_kernel.txt §5):
                       #   order      16     42     64     66     96    128
                       #   rocSOLVER  12.1   89.0  208.9   4.3    6.7   11.6 ms   (its small-matrix path below 65)
                       #   engine      0.26   1.10   1.83  2.11   7.0   11.9 ms   (one workgroup per matrix; the QL sweep is one wave's sequential
                       #                                                       work, 540 cycles per rotation: 6.1 ms for ONE order-128 problem, rocSOLVER 2.5)


def _eigh_psd(A):
    """Eigendecomposition of a batch of symmetric matrices ``[n, s, s]`` in the shapes of ``torch.linalg.eigh``.  On the GPU, orders
    up to NATIVE_EIGH_MAX go to the engine's own batched routine (``nphip_batched_eigh``, nutpie_amd/csrc/linalg.hip: one workgroup
    per matrix, the matrix in LDS; results independent of the batch), larger ones to rocSOLVER through torch; on the CPU (tests) it
    is LAPACK through torch."""
    import torch

    if A.is_cuda and A.shape[-1] <= NATIVE_EIGH_MAX:
        from nutpie_amd import _lib

        return _lib.batched_eigh(A)
    return torch.linalg.eigh(A)


def estimate(x, gx, gamma: float, cutoff: float, k_max: int = K_MAX, basis_draws: int | None = None) -> Transform:
    """The low-rank metric of every chain from its window: ``x``, ``gx``: [n, m, D] draws and gradients in model space.

    scaling      s_i = sqrt(std(x_i) / std(g_i)) over ALL m draws        (the diagonal "diag" adaptation uses the same ratio,
                                                                        python/nutpie/normalizing_flow.py:1906-1915)
    subspace     orthonormal Q of span{scaled draws, scaled gradients} of ``basis_draws`` draws thinned evenly from the window
                 (None: all of them) — eigh of the 2b x 2b Gram matrix; the whole space when b - 1 >= D
    projected    Cx = Px'Px / b + gamma I,  Cg = Pg'Pg / b + gamma I
    metric       S = Cx # Cg^-1 (geometric mean: S Cg S = Cx), eigh(S) -> the k_max eigenvalues furthest from 1 in log scale
                 among those outside [1/cutoff, cutoff]; V = Q W
    """
    import torch

    n, m, D = x.shape
    # a window with a non-finite entry says nothing: that chain gets the identity (sigma = 1, no columns) instead of a NaN metric
    ok = (torch.isfinite(x).flatten(1).all(1) & torch.isfinite(gx).flatten(1).all(1))[:, None, None]
    x, gx = torch.where(ok, x, torch.zeros_like(x)), torch.where(ok, gx, torch.zeros_like(gx))
    mean = x.mean(1)
    sx = x.std(1, unbiased=True)
    sg = gx.std(1, unbiased=True)
    stds = torch.sqrt(sx / sg)
    stds = torch.where(torch.isfinite(stds) & (stds > 0), stds, torch.ones_like(stds)).clamp(1e-10, 1e10)
    gmean = gx.mean(1, keepdim=True)
    if basis_draws is not None and m > basis_draws:
        pick = torch.as_tensor(basis_pick(m, basis_draws), device=x.device)
        x, gx = x.index_select(1, pick), gx.index_select(1, pick)
        m = int(pick.numel())
    X = (x - mean[:, None, :]) / stds[:, None, :]
    G = (gx - gmean) * stds[:, None, :]
    if m - 1 >= D:
        # draws and gradients each span the whole space: work in it directly (Q = I)
        r = D
        Q = None
        Px, Pg = X, G
        keep = torch.ones(n, r, dtype=torch.bool, device=x.device)
    else:
        r = 2 * m
        Z = torch.cat([X, G], 1)                                   # [n, 2m, D]
        ev, U = _eigh_psd(Z @ Z.transpose(1, 2))                   # [n, 2m], [n, 2m, 2m]
        keep = ev > (1e-10 * ev[:, -1:].clamp_min(1e-300))
        scale = torch.where(keep, ev.clamp_min(1e-300).rsqrt(), torch.zeros_like(ev))
        Q = Z.transpose(1, 2) @ (U * scale[:, None, :])            # [n, D, 2m], orthonormal columns (zero where dropped)
        Px, Pg = X @ Q, G @ Q                                      # [n, m, 2m]
    eye = torch.eye(r, dtype=x.dtype, device=x.device)
    Cx = Px.transpose(1, 2) @ Px / m + gamma * eye
    Cg = Pg.transpose(1, 2) @ Pg / m + gamma * eye
    # a Gram matrix that overflowed (large finite draws) says as little as a non-finite window: the identity for that chain
    fin = (torch.isfinite(Cx).flatten(1).all(1) & torch.isfinite(Cg).flatten(1).all(1))
    if not bool(fin.all()):
        Cx = torch.where(fin[:, None, None], Cx, eye.expand_as(Cx))
        Cg = torch.where(fin[:, None, None], Cg, eye.expand_as(Cg))
        stds = torch.where(fin[:, None], stds, torch.ones_like(stds))
    # S = Cg^-1/2 (Cg^1/2 Cx Cg^1/2)^1/2 Cg^-1/2
    eg, Ug = _eigh_psd(Cg)
    eg = eg.clamp_min(1e-300)
    half = (Ug * eg.sqrt()[:, None, :]) @ Ug.transpose(1, 2)
    ihalf = (Ug * eg.rsqrt()[:, None, :]) @ Ug.transpose(1, 2)
    em, Um = _eigh_psd(half @ Cx @ half)
    mid = (Um * em.clamp_min(0).sqrt()[:, None, :]) @ Um.transpose(1, 2)
    S = ihalf @ mid @ ihalf
    es, W = _eigh_psd(0.5 * (S + S.transpose(1, 2)))
    # The geometric mean of A = Cx (eigenvalues in [gamma, |Cx|]) and B = Cg^-1 (in [1 / |Cg|, 1 / gamma]) has its eigenvalues in
    # [sqrt(gamma / |Cg|), sqrt(|Cx| / gamma)]: anything outside is rounding — half Cx half has a condition number of 1 / gamma^2 and an
    # eigensolver is accurate to eps |A|, so its smallest eigenvalues can come out zero or negative, S singular, and a column with
    # lambda = 1e-300 would be "the most extreme direction" (measured: radon, a chain still in transit at the first boundary: every
    # draw of the next window diverged with a NaN energy; one chain of 96 with rocSOLVER, two with the engine's solver)
    lo = (gamma / eg[:, -1].clamp_min(gamma)).sqrt()
    hi = (Cx.diagonal(dim1=1, dim2=2).sum(1).clamp_min(gamma) / gamma).sqrt()      # (the trace bounds the largest eigenvalue)
    es = torch.maximum(torch.minimum(es, hi[:, None]), lo[:, None])
    # Re-centre the spectrum on its bulk.  With strong correlations sqrt(std(x) / std(g)) over-scales EVERY coordinate (std(x)
    # carries the shared directions, std(g) the conditional precisions): a few eigenvalues end up large and all the others small
    # (measured: 3 at ~500, 57 at ~0.03) — more directions than k_max columns can repair.  Dividing by the median eigenvalue
    # (and folding it into the diagonal scaling) leaves the bulk at 1 and only the genuine outliers for the low-rank part.
    live = ((W * W) * keep[:, :, None].to(W.dtype)).sum(1) > 0.5     # eigenvectors inside the span of the window (not the dropped columns of Q)
    log_es = torch.where(live, es.log(), torch.full_like(es, float("nan")))
    centre = torch.nan_to_num(torch.nanmedian(log_es, dim=1).values, nan=0.0).exp()       # [n]
    centre = torch.where(ok[:, 0, 0], centre, torch.ones_like(centre))                     # (a window that was thrown away: exactly the identity)
    es = es / centre[:, None]
    stds = stds * centre.sqrt()[:, None]
    score = es.log().abs()
    outside = (score > float(np.log(cutoff))) & live
    score = torch.where(outside, score, torch.full_like(score, -1.0))
    k = min(k_max, r)
    top = torch.topk(score, k, dim=1)
    sel = top.indices                                           # [n, k]
    lam = torch.gather(es, 1, sel)
    used = top.values > 0
    Wsel = torch.gather(W, 2, sel[:, None, :].expand(n, r, k))
    V = Wsel if Q is None else Q @ Wsel                         # [n, D, k]
    d = torch.where(used, lam.sqrt() - 1.0, torch.zeros_like(lam))
    V = V * used[:, None, :].to(V.dtype)
    if k < k_max:
        V = torch.cat([V, torch.zeros(n, D, k_max - k, dtype=V.dtype, device=V.device)], 2)
        d = torch.cat([d, torch.zeros(n, k_max - k, dtype=d.dtype, device=d.device)], 1)
    return Transform(mean, stds, V.contiguous(), d.contiguous())


def basis_draws_for(dim: int) -> int:
    return int(BASIS_DRAWS) if BASIS_DRAWS else (32 if dim <= 256 else 64)


def basis_pick(m: int, basis_draws: int | None):
    """Indices (into a window of ``m`` draws) of the draws that span the subspace of the low-rank part: all of them, or ``basis_draws``
    thinned evenly — what :func:`estimate` selects."""
    if basis_draws is not None and m > basis_draws:
        return np.unique(np.round(np.linspace(0, m - 1, basis_draws)).astype(np.int64))
    return np.arange(m, dtype=np.int64)


RELEASE_WITHOUT_COLUMNS = True   # a chain on its own diagonal metric whose window needs no low-rank part is released, not handed a frozen diagonal


def torch_index(mask, like):
    import torch

    return torch.as_tensor(np.nonzero(mask)[0], device=like.device)


NATIVE_ESTIMATOR = True   # the estimator as one kernel below the C-ABI (nphip_low_rank_estimate) wherever it covers the shape


def estimate_window(draws, grads, chains, lo: int, hi: int, gamma: float, cutoff: float, k_max: int = K_MAX, basis_draws: int | None = None,
                    max_workgroups: int = 0):
    """(sigma^2 [n, D], V rows [n, k_max, D], lambda [n, k_max]) of ``chains`` (indices, or None for all) from the window
    ``[lo, hi)`` of the trace arrays ``draws`` / ``grads`` ``[n_all, T, D]`` — :func:`estimate` + :func:`metric_of`, computed by the engine's
    own kernel (``nphip_low_rank_estimate``: one workgroup per chain, nothing copied out of the trace, one launch) where that covers
    the shape: up to 512 dimensions with at most 32 basis draws."""
    import torch

    D, m = int(draws.shape[2]), int(hi) - int(lo)
    pick = basis_pick(m, basis_draws)
    if NATIVE_ESTIMATOR and draws.is_cuda and k_max <= 16 and len(pick) <= D:
        from nutpie_amd import _lib

        if _lib.low_rank_estimate_supported(D, m, len(pick), k_max):
            idx = None if chains is None else torch.as_tensor(np.asarray(chains, dtype=np.int64), device=draws.device)
            sig2, V, lam, _ = _lib.low_rank_estimate(draws, grads, idx, lo, hi, pick, gamma, cutoff, k_max, max_workgroups)
            return sig2, V, lam
    if chains is None:
        x, g = draws[:, lo:hi], grads[:, lo:hi]
    else:
        idx = torch.as_tensor(np.asarray(chains, dtype=np.int64), device=draws.device)
        x, g = draws[idx, lo:hi], grads[idx, lo:hi]
    return metric_of(estimate(x, g, gamma, cutoff, k_max=k_max, basis_draws=basis_draws))


def schedule_of(settings):
    """:func:`window_schedule` from a settings object (the keys of ``src/wrapper.rs:198-240``; ``PyNutsSettings.LowRank`` sets
    ``mass_matrix_update_freq`` to its low-rank default of 10, and whatever the user sets afterwards — 1 included — is used as given)"""
    return window_schedule(int(settings.num_tune), float(settings.early_window), float(settings.step_size_window), int(settings.mass_matrix_switch_freq),
                           int(settings.early_window_switch_freq), max(1, int(settings.mass_matrix_update_freq)))


def pause_draws(num_tune: int):
    return [p for p, _ in window_schedule(num_tune)]


def metric_of(T: Transform):
    """(sigma^2 [n, D], V rows [n, k, D], lambda [n, k]) of a :class:`Transform`: ``M^-1 = L L'`` with ``L = D^1/2 (I + V d V')``,
    i.e. eigenvalues ``(1 + d)^2`` on the columns of V (unused columns: V = 0, lambda = 1)."""
    return T.stds * T.stds, T.V.transpose(1, 2).contiguous(), ((1.0 + T.d) ** 2).contiguous()


class LowRankSampler:
    """A ``PySampler`` in manual mode (settings ``low_rank_metric``, pause draws at the window boundaries, gradients stored) plus
    the host thread that drives it and hands the engine a new metric at every boundary.  Same handle surface as ``PySampler``
    (wait / pause / resume / abort / is_finished / progress / inspect / take_results); the trace is in model space throughout."""

    def __init__(self, inner, device, gamma, cutoff, schedule):
        self._inner = inner
        self._device = device
        self._gamma, self._cutoff = float(gamma), float(cutoff)
        self._schedule = [(int(p), int(a)) for p, a in schedule]    # (pause draw, first draw of its window): window_schedule
        self._pauses = [p for p, _ in self._schedule]
        self._had_columns = np.zeros(inner.num_chains, dtype=bool)  # per chain: its current metric has a low-rank part
        self._handed = np.zeros(inner.num_chains, dtype=bool)       # per chain: it has been handed a metric (its own diagonal adaptation is off)
        self.fallbacks = 0            # hand-ins that kept only the diagonal part because the last low-rank metric deepened the chain's trees
        self._chain_next = np.zeros(inner.num_chains, dtype=np.int64)   # per chain: index of the next boundary it stops at
        self._stream = None           # the estimator's stream (made on its thread)
        self._lock = threading.Lock()
        self._cv = threading.Condition(self._lock)
        self._step_lock = threading.Lock()   # held while the engine steps or a metric is being installed: readers take it
        self._paused = False
        self._abort = False
        self._done = False
        self._error = None
        # the estimator kernel's launch is capped at the CUs the engine's kernel leaves idle (one wave per chain, four chains per workgroup
        # from 257 chains on; a CU per chain below that): 0 = no cap (the other drivers estimate while nothing of the engine's needs a CU)
        self._estimator_workgroups = 0
        self.switch_log = []          # (boundary draw, mean number of columns used, seconds spent estimating, chains handed in, seconds since the start)
        self._t_start = time.perf_counter()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ driver
    def _run(self):
        """Drive the engine one launch at a time while chains can still stop at a window boundary.  A chain that has stopped gets
        ITS new metric without waiting for the other chains: the estimate is per chain anyway, and in lockstep a single chain
        with a poor early metric (max-depth trees for a whole window) held 511 finished chains for most of the job (measured:
        radon, 512 chains: 284 launches for a mean of 50 launches' worth of work per chain,
        profiles/r4_low_rank_register_kernel.txt §4).  The estimate runs on a worker thread and its own stream WHILE the engine
        does its next launch for the chains that have not stopped, and is installed right after that launch — always that one,
        so that which chains are estimated together (rocSOLVER's batched results depend on the batch in the last bits) depends
        on the engine's state alone and a job is reproducible from its seed.  Stopped chains are handed in together: when nothing
        else is running, when they are a quarter of the unfinished chains, or after HOLD_LAUNCHES looks."""
        from concurrent.futures import ThreadPoolExecutor

        pool = ThreadPoolExecutor(1)
        try:
            held = 0
            # launches between two looks at the chains: one launch of a kernel that evaluates inside the launch (hundreds of gradient
            # evaluations per chain), sixteen of a model whose launches are one evaluation each.  Decided from the model's kind, not
            # from a clock: when the looks happen decides which chains are estimated together, and a job must be reproducible
            kind = type(getattr(self._inner, "_model", None)).__name__
            resident = kind in ("TridiagGaussianModel", "JitDensityModel") or (kind in ("HostCallbackModel", "BridgeStanModel") and self._inner.host_mode == "resident")
            per_look = int(getattr(self._inner, "launches_per_look", 0)) or (1 if resident else 16)
            if self._native_estimator():   # (the windows go to the engine's own estimator kernel: a driver without lock-step rules)
                return self._run_fast(pool, kind, resident, per_look)
            job = None                                             # the estimate in flight
            busy = np.zeros(self._inner.num_chains, dtype=bool)    # chains whose estimate is in flight
            while True:
                with self._cv:
                    while self._paused and not self._abort:
                        self._cv.wait()
                    if self._abort:
                        break
                with self._step_lock:
                    pending = bool((self._chain_next < len(self._pauses)).any())
                    done, cnt, ms = self._inner.step(per_look if pending else 16)
                    if job is not None:                            # its estimate ran while the engine did that launch
                        self._install(job.result())
                        busy[:] = False
                        job = None
                    if done:
                        break
                    if pending:
                        code = self._inner.waiting_codes()
                        wait = (code == 1) & ~busy
                        if wait.any():
                            n_wait, n_run = int(wait.sum()), int((code == 0).sum())
                            if n_run == 0 or 4 * n_wait >= n_wait + n_run or held >= HOLD_LAUNCHES:
                                grp = np.nonzero(wait)[0]
                                busy[grp] = True
                                job = pool.submit(self._estimate, grp, self._chain_next[grp].copy())
                                held = 0
                            else:
                                held += 1
        except BaseException as e:  # noqa: BLE001 - reported by wait()
            self._error = e
        finally:
            pool.shutdown(wait=True)
            with self._cv:
                self._done = True
                self._cv.notify_all()

    def _run_fast(self, pool, kind, resident, per_look):
        """The driver when the windows go to the engine's own estimator kernel (``estimate_window``: its result for a chain does not
        depend on what else is in the batch, so WHO is handed in together may depend on the clock — what a chain is handed, and at
        which draw, does not: the job stays reproducible from its seed).  Nothing waits for anything it does not need: the engine
        takes one launch after the other — a quarter of their usual length while chains can still stop, so that a chain that stops
        is seen within a millisecond —; whenever no estimate is in flight, every chain that has stopped by then goes into the next
        one (worker thread, own stream); a finished estimate is installed between two launches.  (Measured and rejected,
        scratch/r6_lr_wall.py: the estimate BETWEEN two launches with the device to itself — radon, 512 chains: 118 hand-ins of
        4.4 ms with nothing else running, 0.84 s of wall against 0.41.)"""
        short = 0
        if resident and per_look == 1 and kind in ("TridiagGaussianModel", "JitDensityModel") and hasattr(self._inner, "set_evals_per_launch"):
            from nutpie_amd import _lib

            short = max(16, (512 if kind == "JitDensityModel" else _lib.default_evals_per_launch(self._inner.dim)) // SHORT_LAUNCH_DIV)
            self._inner.set_evals_per_launch(short)
        n_ch = int(self._inner.num_chains)
        self._estimator_workgroups = max(8, 256 - (n_ch if n_ch <= 256 else (n_ch + 3) // 4) - 8) if ESTIMATOR_CAP is None else int(ESTIMATOR_CAP)
        job = None
        while True:
            with self._cv:
                while self._paused and not self._abort:
                    self._cv.wait()
                if self._abort:
                    break
            with self._step_lock:
                pending = bool((self._chain_next < len(self._pauses)).any())
                if short and not pending:
                    self._inner.set_evals_per_launch(0)   # (nothing stops any more: the default launch length)
                    short = 0
                done, cnt, ms = self._inner.step(per_look if pending else 16)
                if done:
                    break
                if not pending:
                    continue
                code = self._inner.waiting_codes()
                if job is not None and (job.done() or not (code == 0).any()):
                    self._install(job.result())
                    job = None
                    code = self._inner.waiting_codes()
                if job is None:
                    wait = code == 1
                    if wait.any():
                        grp = np.nonzero(wait)[0]
                        job = pool.submit(self._estimate, grp, self._chain_next[grp].copy())

    def _native_estimator(self):
        """Whether this job's windows go to ``nphip_low_rank_estimate`` (a GPU trace of a shape the kernel covers)."""
        try:
            import torch

            from nutpie_amd import _lib

            D = int(self._inner.dim)
            return bool(NATIVE_ESTIMATOR and torch.cuda.is_available() and hasattr(self._inner, "device_ptr") and
                        _lib.low_rank_estimate_supported(D, WINDOW_MAX, min(basis_draws_for(D), D), K_MAX) and basis_draws_for(D) <= 32)
        except Exception:  # noqa: BLE001 - (a stand-in engine of the CPU tests)
            return False

    def _views(self):
        from nutpie_amd.distributed import device_tensor

        n, T, D = self._inner.num_chains, self._inner.total_draws, self._inner.dim
        draws = device_tensor(self._inner.device_ptr("draws"), (n, T, D), "float64", self._device)
        grads = device_tensor(self._inner.device_ptr("gradient"), (n, T, D), "float64", self._device)
        return draws, grads

    def _n_steps(self):
        from nutpie_amd.distributed import device_tensor

        return device_tensor(self._inner.device_ptr("n_steps"), (self._inner.num_chains, self._inner.total_draws), "int64", self._device)

    def _estimate(self, chains, at):
        """(worker thread) New metrics for the stopped ``chains``; ``at``: the index of the boundary each one is at.  Chains at the
        same boundary are one batch.  -> [(chains, boundary index, sigma^2, V, lambda, log entry)], ready to install."""
        import torch

        draws, grads = self._views()
        out = []
        cuda = draws.is_cuda
        if cuda:
            torch.cuda.set_device(self._device)
            if self._stream is None:
                # a HIGH-PRIORITY stream: streams of one priority share the device's few hardware queues round-robin, and an estimator
                # stream that lands on the engine stream's queue serialises every estimate with the launch it should run beside
                # (measured, scratch/r6_lr_outliers.py: two jobs of ten took 0.44 s instead of 0.33, every hand-in twice as long)
                self._stream = torch.cuda.Stream(self._device, priority=-1)
        with torch.no_grad(), (torch.cuda.stream(self._stream) if cuda else contextlib.nullcontext()):
            n_steps = None
            for i in np.unique(at):
                t0 = time.perf_counter()
                grp = chains[at == i]
                hi, lo = self._schedule[i]
                lo = max(lo, hi - WINDOW_MAX)
                sig2, V, lam = estimate_window(draws, grads, None if len(grp) == draws.shape[0] else grp, lo, hi, self._gamma, self._cutoff,
                                               basis_draws=basis_draws_for(draws.shape[2]), max_workgroups=self._estimator_workgroups)
                # A chain whose LAST low-rank metric deepened its trees (mean leapfrogs per draw of the window just finished against
                # the window before that hand-in) keeps only the diagonal part this time: a metric estimated from a chain still in
                # transit can point the columns the wrong way, and the next window would be estimated from max-depth draws
                bad = np.zeros(len(grp), dtype=bool)
                if i >= 1 and self._had_columns[grp].any():
                    if n_steps is None:
                        n_steps = self._n_steps()
                    p_prev, a_prev = self._schedule[i - 1]
                    idx = torch.as_tensor(grp, device=draws.device)
                    after = n_steps[idx, min(p_prev + SETTLE, hi - 1):hi].double().mean(1)
                    before = n_steps[idx, max(a_prev, p_prev - (hi - p_prev)):p_prev].double().mean(1)
                    bad = ((after > 1.3 * before + 1.0).cpu().numpy()) & self._had_columns[grp]
                    if bad.any():
                        b = torch.as_tensor(bad, device=lam.device)
                        lam = torch.where(b[:, None], torch.ones_like(lam), lam)
                        V = torch.where(b[:, None, None], torch.zeros_like(V), V)
                # only the columns some chain uses (estimate() puts a chain's used columns first): every column handed in costs
                # every leapfrog of every chain a dot product and an update in both halves of the step.  Counted on the host from one
                # small copy (the one synchronisation; no torch kernel in the path of a hand-in without columns: a torch kernel's
                # first launch in a process costs tens of milliseconds, which a 0.3 s job sees)
                n_cols = (lam.cpu().numpy() != 1.0).sum(1) if lam.numel() else np.zeros(len(grp), dtype=np.int64)
                k_used = int(n_cols.max()) if len(n_cols) else 0
                has = n_cols > 0
                if k_used:
                    V, lam = V[:, :k_used].contiguous(), lam[:, :k_used].contiguous()
                cols = float(n_cols.mean()) if k_used else 0.0
                if cuda:
                    self._stream.synchronize()
                out.append((grp, int(i), sig2, V if k_used else None, lam if k_used else None, (hi, cols, time.perf_counter() - t0, len(grp), t0 - self._t_start), has, int(bad.sum())))
        return out

    def _install(self, metrics):
        """(driver thread, between two launches) hand the estimated metrics to the engine"""
        for grp, i, sig2, V, lam, entry, has, n_bad in metrics:
            # A chain whose window shows no direction outside the cutoff, and that runs on the diagonal metric it adapts itself, simply
            # goes on: that metric is the estimator's diagonal part, refreshed every draw on the device instead of at six boundaries
            # (measured, radon, 512 chains: a chain still in transit at its first boundary was frozen on the metric of those 59 draws
            # for the next 80 — 120 leapfrogs per draw — and the job ended with it: scratch/r6_lr_wall.py).  A chain that has been
            # handed a metric before keeps being handed one (its own estimators stopped then).
            free = ~np.asarray(has, dtype=bool) & ~self._handed[grp] if RELEASE_WITHOUT_COLUMNS and hasattr(self._inner, "release") else np.zeros(len(grp), dtype=bool)
            if free.all():
                self._inner.release(grp)
            else:
                if free.any():
                    self._inner.release(grp[free])
                    keep = torch_index(~free, sig2)
                    sig2, V, lam = sig2[keep], (None if V is None else V[keep]), (None if lam is None else lam[keep])
                self._inner.set_metric(grp[~free], sig2, V, lam)
                self._handed[grp[~free]] = True
            self._chain_next[grp] = i + 1
            self._had_columns[grp] = has
            self.fallbacks += n_bad
            self.switch_log.append(entry)

    def _adapt(self, chains):
        """Estimate and install in one go (what the driver does, without the overlap)."""
        chains = np.asarray(chains)
        self._install(self._estimate(chains, self._chain_next[chains].copy()))

    # ------------------------------------------------------------------ handle surface
    def wait(self, timeout_seconds=None):
        with self._cv:
            if self._paused:
                self._paused = False
                self._cv.notify_all()
            end = None if timeout_seconds is None else time.monotonic() + timeout_seconds
            while not self._done:
                left = 0.1 if end is None else min(0.1, end - time.monotonic())
                if end is not None and left <= 0:
                    raise TimeoutError("Timeout while waiting for sampler to finish")
                self._cv.wait(left)
        if self._error is not None:
            raise RuntimeError(str(self._error)) from self._error

    def pause(self):
        with self._cv:
            self._paused = True

    def resume(self):
        with self._cv:
            self._paused = False
            self._cv.notify_all()

    def abort(self):
        with self._cv:
            self._abort = True
            self._cv.notify_all()
        self._thread.join()
        self._inner.abort()

    def is_finished(self):
        return self._done

    def is_empty(self, ignore_error=False):
        return self._inner.is_empty(ignore_error)

    def inspect(self):
        with self._step_lock:   # (between two engine steps, never inside an adaptation)
            return self._inner.inspect()

    def progress(self):
        with self._step_lock:
            return self._inner.progress()

    def take_results(self):
        self._thread.join()
        return self._inner.take_results()

    def close(self):
        if not self._done:
            self.abort()
        self._inner.close()

    def __getattr__(self, name):   # num_chains, dim, device_ptr(), seconds, ...
        return getattr(self._inner, name)

    def __setattr__(self, name, value):
        if name.startswith("_") and name in ("_keep_host_draws", "_device_expand", "_keep_tensors"):
            setattr(self._inner, name, value)
        else:
            object.__setattr__(self, name, value)


def make_sampler(compiled_model, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
    """``compiled_model._make_sampler`` for ``adaptation="low_rank"``: any model flavour."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("adaptation='low_rank' needs a GPU: the nutpie-hip engine has no CPU fallback")
    if not engine_kw.get("store_draws", True):
        raise ValueError("adaptation='low_rank' estimates the metric from the stored draws: store_draws=False cannot be combined with it")
    schedule = schedule_of(settings)
    pauses = [p for p, _ in schedule]
    inner_settings = settings.clone()
    # (the estimator needs the gradients of the window's draws; ``mass_matrix_update_freq`` paces the ESTIMATES — the engine's own diagonal part
    # is refreshed with every draw, as before the key had a low-rank default)
    inner_settings.update(low_rank_metric=True, store_gradient=True, mass_matrix_update_freq=1)
    inner_settings.set_pause_draws(pauses)
    device = int(engine_kw.get("device", 0) or 0)
    inner = compiled_model._make_sampler(inner_settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store,
                                         **{**engine_kw, "manual": True})
    return LowRankSampler(inner, device, settings._low_rank["mass_matrix_gamma"], settings._low_rank["mass_matrix_eigval_cutoff"], schedule)
