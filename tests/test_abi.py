"""The C-ABI library loads on a CPU-only box and exports every symbol include/nutpie_hip.h declares;
settings semantics follow the reference's PyNutsSettings (src/wrapper.rs:210-451, 563-620)."""
import ctypes
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nutpie_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(nphip_[a-z0-9_]+)\s*\(", text))
    typedefs = set(re.findall(r"\(\*(nphip_[a-z0-9_]+)\)", text))
    return sorted(names - typedefs)


def test_library_exports_every_declared_symbol():
    from nutpie_amd import _lib

    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f"{s} is declared in include/nutpie_hip.h but not exported"


def test_oracle_library_exports():
    import oracle

    L = oracle.lib()
    for s in ("oracle_sample_tridiag", "oracle_sample_callback", "oracle_default_settings", "oracle_philox", "oracle_dot"):
        assert hasattr(L, s)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nutpie_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("the oracle/", ""), f


def test_settings_defaults_and_json():
    from nutpie_amd._lib import PyNutsSettings

    s = PyNutsSettings.Diag(123)
    d = s.as_dict()
    assert d["sampler"] == "nuts" and d["adaptation"] == "diag"
    st = d["settings"]
    # defaults visible in the reference: 400 tune + 1000 draws, 6 chains (docs/_freeze/index; tests/test_stan.py:241),
    # maxdepth 10 (sample.py:896-899), target_accept 0.8, max_energy_error 1000 (docs/sampling-options.qmd:71-73)
    assert (st["num_tune"], st["num_draws"], st["num_chains"], st["maxdepth"], st["seed"]) == (400, 1000, 6, 10, 123)
    assert st["max_energy_error"] == 1000 and st["adapt_options"]["step_size_settings"]["target_accept"] == 0.8
    assert st["adapt_options"]["mass_matrix_options"] == {"store_mass_matrix": False, "use_grad_based_estimate": True}
    json.dumps(d)
    # no seed -> random seed (wrapper.rs:453-458)
    assert PyNutsSettings.Diag().as_dict()["settings"]["seed"] != PyNutsSettings.Diag().as_dict()["settings"]["seed"]


def test_settings_flat_attribute_names():
    from nutpie_amd._lib import PyNutsSettings

    s = PyNutsSettings.Diag(1)
    s.update({"num_tune": 10, "num_draws": 20, "num_chains": 3, "maxdepth": 7, "target_accept": 0.9, "initial_step": 0.5,
              "max_energy_error": 50.0, "store_gradient": True, "store_mass_matrix": True, "store_divergences": True,
              "store_unconstrained": True, "window_switch_freq": 33, "early_window_switch_freq": 5, "step_size_jitter": 0.1,
              "use_grad_based_mass_matrix": False, "check_turning": False, "mindepth": 1, "max_step_size": 2.0})
    st = s.as_dict()["settings"]
    assert st["num_tune"] == 10 and st["maxdepth"] == 7 and st["store_gradient"] and st["store_divergences"]
    ao = st["adapt_options"]
    assert ao["mass_matrix_switch_freq"] == 33 and ao["early_mass_matrix_switch_freq"] == 5
    assert ao["step_size_settings"]["jitter"] == 0.1 and ao["step_size_settings"]["initial_step"] == 0.5
    assert ao["mass_matrix_options"] == {"store_mass_matrix": True, "use_grad_based_estimate": False}
    assert ao["step_size_settings"]["adapt_options"]["dual_average"]["max_step_size"] == 2.0
    s.mass_matrix_switch_freq = 44          # alias of window_switch_freq for diag (wrapper.rs:214-229, 291-303)
    assert s.as_dict()["settings"]["adapt_options"]["mass_matrix_switch_freq"] == 44
    s.step_size_jitter = 0.0                # 0 => None (wrapper.rs:397-399)
    assert s.as_dict()["settings"]["adapt_options"]["step_size_settings"]["jitter"] is None
    s.step_size_adapt_method = "0.25"       # "<float>" => fixed step size (wrapper.rs:350-357)
    assert s.as_dict()["settings"]["adapt_options"]["step_size_settings"]["adapt_options"]["method"] == {"fixed": 0.25}
    s.step_size_adapt_method = "adam"       # wrapper.rs:344-349
    s.step_size_adam_learning_rate = 0.07
    sss = s.as_dict()["settings"]["adapt_options"]["step_size_settings"]["adapt_options"]
    assert sss["method"] == "adam" and sss["adam"]["learning_rate"] == 0.07
    s.step_size_adapt_method = "dual_average"
    assert s.as_dict()["settings"]["adapt_options"]["step_size_settings"]["adapt_options"]["method"] == "dual_average"
    assert s.num_tune == 10 and s.seed == 1 and s.store_gradient is True


def test_settings_error_classes():
    from nutpie_amd._lib import PyMclmcSettings, PyNutsSettings

    s = PyNutsSettings.Diag(1)
    with pytest.raises(AttributeError, match="Unknown settings attribute: nonsense"):   # wrapper.rs:610-614
        s.nonsense = 3
    with pytest.raises(AttributeError, match="Unknown settings attribute"):
        s.update({"step_size": 0.1})  # an MCLMC-only key (wrapper.rs:623-640)
    for key in ("mass_matrix_eigval_cutoff", "mass_matrix_gamma"):                      # low-rank only
        with pytest.raises(ValueError, match=f"Option {key} not available for diag adaptation"):  # wrapper.rs:138-145
            setattr(s, key, 2.0)
        setattr(s, key, None)  # None is ignored (Option<f64>)
    with pytest.raises(ValueError, match="Option train_on_orbit not available for diag adaptation"):
        s.train_on_orbit = True
    with pytest.raises(ValueError, match="step_size_jitter must be positive"):          # wrapper.rs:394-396
        s.step_size_jitter = -0.1
    with pytest.raises(ValueError, match="must be a positive float"):                   # wrapper.rs:358-362
        s.step_size_adapt_method = "banana"
    with pytest.raises(ValueError, match="must be a string"):
        s.step_size_adapt_method = 0.3
    with pytest.raises(ValueError):
        s.maxdepth = 40
    with pytest.raises(TypeError):
        s.num_tune = -5
    with pytest.raises(TypeError):
        s.store_gradient = 1
    lr = PyNutsSettings.LowRank(1)                                                      # wrapper.rs:725-729, 307-334
    assert lr.as_dict()["adaptation"] == "low_rank" and lr.mass_matrix_gamma == 1e-5
    lr.update(mass_matrix_eigval_cutoff=3, mass_matrix_gamma=1e-4)
    assert lr.as_dict()["settings"]["adapt_options"]["mass_matrix_options"] == {"store_mass_matrix": False, "gamma": 1e-4, "eigval_cutoff": 3.0}
    with pytest.raises(ValueError, match="greater than one"):
        lr.mass_matrix_eigval_cutoff = 1.0
    with pytest.raises(ValueError, match="Option train_on_orbit not available"):
        lr.train_on_orbit = True
    assert lr.clone().mass_matrix_eigval_cutoff == 3.0
    with pytest.raises(NotImplementedError):
        PyNutsSettings.Flow(1)
    with pytest.raises(NotImplementedError):
        PyMclmcSettings.Diag(1)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never fall back to the oracle or the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import numpy as np

    import nutpie_amd

    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        nutpie_amd.sample(nutpie_amd.std_normal(4), chains=2, progress_bar=False)
    with pytest.raises(RuntimeError, match="GPU"):
        m = nutpie_amd.from_torchfunc(3, lambda: (lambda x: (-(x * x).sum(-1) / 2, -x)))
        nutpie_amd.sample(m, chains=2, progress_bar=False)
    from nutpie_amd import _lib

    assert _lib.lib().nphip_device_count() == 0
    with pytest.raises(RuntimeError):
        _lib.test_detmath("exp", np.zeros(3))


def test_model_validation():
    import numpy as np

    from nutpie_amd import _lib

    with pytest.raises(ValueError):
        _lib.TridiagGaussianModel(np.ones(4), offdiag=np.ones(4))
    m = _lib.TridiagGaussianModel(np.ones(4), offdiag=np.ones(3), mu=np.zeros(4))
    with pytest.raises(ValueError, match="incorrect length"):   # src/pyfunc.rs:561-563
        m.set_init("explicit", np.zeros((2, 5)))
    m.set_init("explicit", np.zeros((2, 4)))
    m.set_init("normal")
    cb = _lib.RAW_LOGP_FN(lambda d, x, g, lp, u: 0)
    _lib.HostCallbackModel(3, cb)
    _lib.HostCallbackModel(3, ctypes.cast(cb, ctypes.c_void_p).value, keep_alive=cb)


def test_default_launch_length_by_dimension():
    """About 10 ms of kernel per launch of a fused model (host.hip: default_evals_per_launch); bench.py asks for the same number."""
    from nutpie_amd import _lib

    assert [_lib.default_evals_per_launch(d) for d in (1, 1000, 1024, 1025, 4096, 4097, 10000, 100000)] == [2048, 2048, 2048, 1024, 1024, 512, 512, 512]


def test_build_configuration_of_the_families_measured_without_interprocedural_allocation():
    """DESIGN.md §4 / profiles/r6_call_placement_and_draw_end.txt: the one-wave kernels with 2 .. 8 chunks per lane (kernels.hip part 12), the dense
    Gaussian's resident kernels (part 11) and the compiled densities are built with ``-mllvm -enable-ipra=0`` — chosen by
    same-box A/B per family; the lean kernels and the fused models' low-rank leaf lose by it.  A pin on the build files, so that the choice is not lost by accident."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = open(os.path.join(root, "nutpie_amd", "csrc", "Makefile")).read()
    assert "kernels_p12.o" in re.search(r"^PARTS = (.*)$", mk, re.M).group(1)
    rule = re.search(r"^kernels_p11\.o kernels_p12\.o: kernels_p%\.o:.*\n\t(.*)$", mk, re.M)
    assert rule and "-mllvm -enable-ipra=0" in rule.group(1) and "-DNPHIP_PART=$*" in rule.group(1)
    generic = re.search(r"^kernels_p%\.o: kernels\.hip.*\n\t(.*)$", mk, re.M)
    assert generic and "enable-ipra" not in generic.group(1)
    src = open(os.path.join(root, "nutpie_amd", "csrc", "kernels.hip")).read()
    assert "#if NPHIP_HAS(12) && !defined(NPHIP_DEV_BUILD)\nhipError_t launch_w1_noipra(" in src
    dens = open(os.path.join(root, "nutpie_amd", "density.py")).read()
    assert 'flags = (_FLAGS + ["-mllvm", "-enable-ipra=0", "-DNPHIP_JIT_DENSITY=1"' in dens
