"""Ahead-of-time compilation of the runtime-compiled density libraries the GPU tests and bench.py's config-3 legs use (into the in-tree
cache nutpie_amd/_density_cache, which travels to the GPU box).  Run by ``__graft_entry__.build()`` as a separate process; safe to run by hand:
    python tests/prebuild_density_cache.py
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from nutpie_amd import density
    from nutpie_amd.radon import radon_density_model

    m = radon_density_model()
    keep = {m.library_path()}
    # the generated densities of the front-end's tests (nutpie_amd/symbolic.py)
    import symbolic_models

    for name, make in symbolic_models.ALL.items():
        cm = make().compile()
        keep.add(cm.library_path())   # (with the number of waves per chain compile() chose for the model)
        if name in symbolic_models.LOW_RANK:
            keep.add(cm.library_path(low_rank=True))   # the variant whose resident kernel integrates under the low-rank metric
    # torch log-densities traced and compiled (nutpie_amd/torch_trace.py): BASELINE config 3 as it is named, and the tracer's test models
    from nutpie_amd.compiled_pyfunc import from_torch_density
    from nutpie_amd.radon import radon_traced_model
    import torch_models

    keep.add(radon_traced_model().library_path())
    for name, make in torch_models.ALL.items():
        D, fn, batched, shared = make()
        keep.add(from_torch_density(D, fn, compile=True, batched=batched, shared_data=shared).library_path())
    # the models of the reference's frozen docs that are sampled through the front-end (tests/test_gpu_reference_fixtures.py)
    from nutpie_amd.compile_pymc import compile_pymc_model

    for make in symbolic_models.DOC_MODELS.values():
        keep.add(compile_pymc_model(make()).library_path())
    cm = symbolic_models.radon().compile(waves_per_chain=2)   # tests/test_gpu_density.py: the metric with two waves per chain
    keep.update((cm.library_path(), cm.library_path(low_rank=True)))
    # Libraries of EARLIER engine sources (the cache key hashes the engine's sources: nothing of this tree can find them any more) are removed —
    # only files older than the engine library this tree has just built, so that a model another process is compiling right now, or a user's
    # own models compiled against this engine, are left alone.
    cache = density.cache_dir()
    engine = os.path.join(ROOT, "nutpie_amd", "libnutpie_hip.so")
    horizon = os.path.getmtime(engine) if os.path.exists(engine) else 0.0
    for f in os.listdir(cache):
        path = os.path.join(cache, f)
        if f.startswith("density_") and path not in keep and os.path.getmtime(path) < horizon:
            os.remove(path)
        elif f.startswith(".build_") and os.path.isdir(path) and os.path.getmtime(path) < horizon:
            shutil.rmtree(path, ignore_errors=True)   # (what an interrupted compilation left behind)


if __name__ == "__main__":
    main()
