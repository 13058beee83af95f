"""Runtime-compiled device densities, the parts that run without a GPU: the generated data block, the source generator, and
the cross-compilation of a model's library (hipcc builds gfx950 code here; nothing is launched)."""
import ctypes
import struct

import numpy as np
import pytest


def test_data_block_layout_and_struct_source():
    from nutpie_amd import density

    data = {"y": np.arange(5.0), "idx": np.array([2, 0, 1], dtype=np.int64), "scale": 1.5, "k": 7}
    layout = density.data_layout(data)
    # pointers, then doubles, then ints (array lengths first): no padding anywhere
    assert layout == [("y", "double*"), ("idx", "int*"), ("scale", "double"), ("n_y", "int"), ("n_idx", "int"), ("k", "int")]
    src = density.struct_source(layout)
    assert "const double* y;" in src and "const int* idx;" in src and "double scale;" in src and "int n_idx;" in src and "int k;" in src
    assert struct.calcsize("<QQdiii") == 8 + 8 + 8 + 12
    with pytest.raises(ValueError, match="defined twice"):
        density.data_layout({"y": np.zeros(2), "n_y": 3})
    with pytest.raises(TypeError, match="boolean"):
        density.data_layout({"flag": True})
    with pytest.raises(TypeError, match="unsupported"):
        density.data_layout({"s": np.array(["a"])})


def test_front_end_contract_without_a_gpu():
    import nutpie_amd
    from nutpie_amd.radon import radon_density_model, synthetic_radon_data, radon_density_data

    m = radon_density_model()
    assert m.n_dim == 173 and m.shapes["county_effect"] == (85,) and m.shapes["sigma"] == ()
    m2 = m.with_data(**radon_density_data(synthetic_radon_data(seed=4)))
    assert m2 is not m and not np.array_equal(m2.data["y"], m.data["y"]) and np.array_equal(m.data["y"], radon_density_data()["y"])
    with pytest.raises(ValueError, match="Unknown data variable"):
        m.with_data(bogus=1)
    ex = m._expand_draws(np.zeros((2, 3, 173)))
    assert ex["county_effect"].shape == (2, 3, 85) and np.all(ex["sigma"] == 1.0)
    with pytest.raises(ValueError, match="nphip_density"):
        nutpie_amd.from_density_source(3, "// nothing here")


def test_density_library_cross_compiles_and_exports_its_entry_points(tmp_path, monkeypatch):
    from nutpie_amd import density
    from nutpie_amd.radon import radon_density_model

    monkeypatch.setenv("NUTPIE_AMD_CACHE", str(tmp_path))
    m = radon_density_model()
    path = density.compile_density(m._source, density.data_layout(m._data), m.n_dim)
    assert path.startswith(str(tmp_path)) and density.compile_density(m._source, density.data_layout(m._data), m.n_dim) == path   # cached
    lib = ctypes.CDLL(path)
    for sym in ("nphip_jit_launch", "nphip_jit_logp", "nphip_jit_nv"):
        assert hasattr(lib, sym), sym
    assert lib.nphip_jit_nv() == 2            # 173 dimensions: two chunks of 128
    with pytest.raises(RuntimeError, match="compiling the density failed"):
        density.compile_density("__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* g, double* l, const double* sh, int lane) { return nope; }", [], 2)


def test_device_memory_scratch_is_a_field_of_the_data_block():
    """``scratch_doubles_per_chain``: the data block carries ``scratch__`` (allocated on the device when a sampler is created, one block
    per resident chain of a launch); the name is reserved"""
    from nutpie_amd import density

    src = "__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* g, double* l, const double* sh, int lane) { return 0.0; }"
    m = density.from_density_source(3, src, {"y": np.zeros(5)}, scratch_doubles_per_chain=lambda d: 2 * len(d["y"]))
    layout = density.data_layout(m._data)
    assert ("scratch__", "double*") in layout and ("n_scratch__", "int") in layout
    assert "const double* scratch__;" in density.struct_source(layout) and "NPHIP_CHAIN_SLOT" in density.generated_source(src, layout)
    assert m._scratch(m.with_data(y=np.zeros(9))._data) == 18
    with pytest.raises(ValueError, match="reserved"):
        density.from_density_source(3, src, {"scratch__": np.zeros(2)}, scratch_doubles_per_chain=4)
    assert "scratch__" not in density.from_density_source(3, src, {"y": np.zeros(5)})._data
