"""vaex_amd — MI355X-native N-d binned statistics / groupby aggregation for vaex.

The only thing in here is vaex's one data-parallel hot path (the `vaex.superagg` kernels behind
df.count/sum/mean/std(binby=...) and df.groupby().agg()), as hand-written HIP kernels in
libvaexhip.so (C-ABI: include/vaex_hip.h) behind the reference's own class surface:

    vaex_amd.superagg      pybind11 shim with the classes of `vaex.superagg`
    vaex_amd.install()     swap it into an (unmodified) vaex installation
    vaex_amd.vaexfast      the legacy `vaex.vaexfast.statisticNd_f8` entry on the same kernels
    vaex_amd.binned        host-side driver mirroring df.count/sum/mean/...(binby=, limits=, shape=)
    vaex_amd.dist          row-sharded multi-GPU reduce of the grids over RCCL

There is no CPU fallback: without a GPU every compute call raises.
"""
# torch (when present) must be imported BEFORE libvaexhip.so is loaded: torch ships its own
# libamdhip64.so (SONAME libamdhip64.so.7); loading it first makes our library bind to the same
# HIP runtime instead of bringing a second one into the process.
try:  # pragma: no cover - plumbing
    import torch as _torch  # noqa: F401
except Exception:  # torch is only needed for device tensors / torch.distributed
    _torch = None

from . import superagg  # noqa: E402  (fails loudly if the extension was not built)

__all__ = ["superagg", "install"]


def install(vaex_module=None, legacy=True):
    """Make vaex use the HIP kernels: replaces the module attribute `vaex.superagg`, which vaex looks
    classes up on by name at call time (vaex/utils.py:754-791, vaex/cpu.py:49-53, :646, vaex/agg.py:286-313).
    legacy=True also points the legacy statistic task (df.minmax / limits=None: vaex/cpu.py:533-538) at
    vaex_amd.vaexfast.statisticNd_f8; the float32 variant keeps the reference's CPU code (it scales in float32)."""
    import sys
    if vaex_module is None:
        import vaex as vaex_module
    vaex_module.superagg = superagg
    sys.modules["vaex.superagg"] = superagg
    if legacy:
        from . import vaexfast as _vf
        legacy_mod = getattr(vaex_module, "vaexfast", None) or sys.modules.get("vaex.vaexfast")
        if legacy_mod is not None:
            legacy_mod.statisticNd_f8 = _vf.statisticNd_f8
    return superagg
