"""pytest plugin of the subprocess tests/test_vaex_reference_suite.py starts over the reference's own test files (TEST
INFRASTRUCTURE; loaded with `-p reftest_plugin`, never by the product).

* the reference's tests/common.py imports vaex-server and pytest-asyncio for its remote fixtures, which exist in neither this image
  nor the oracle build of the reference: empty stand-ins are registered (VAEX_TEST_SKIP_REMOTE=1 keeps those fixtures out of the
  parametrisations, tests/common.py:225-236);
* VAEX_AMD_REFTEST_INSTALL=1: vaex_amd.install() before the first test — from there on every aggregation task part, statistic,
  selection, filter and groupby of the reference's tests runs on the HIP classes (or on the per-task fallback install() decides);
* every test's outcome goes to the JSON file VAEX_AMD_REFTEST_REPORT names, with install()'s counters."""
import json
import os
import sys
import types

import pytest

import vaex

for _name in ("vaex.server", "vaex.server.service", "vaex.server.tornado_server", "vaex.server.dummy", "vaex.server.fastapi"):
    sys.modules[_name] = types.ModuleType(_name)
vaex.server = sys.modules["vaex.server"]
for _sub in ("service", "tornado_server", "dummy", "fastapi"):
    setattr(vaex.server, _sub, sys.modules["vaex.server." + _sub])
_asyncio = types.ModuleType("pytest_asyncio")
_asyncio.fixture = pytest.fixture
sys.modules.setdefault("pytest_asyncio", _asyncio)

INSTALL = os.environ.get("VAEX_AMD_REFTEST_INSTALL") == "1"
HOST_LOGIC = os.environ.get("VAEX_AMD_REFTEST_INSTALL") in ("host", "hostgb")   # install()'s host logic alone over vaex's own classes (no GPU needed)
HOST_GROUPBY = os.environ.get("VAEX_AMD_REFTEST_INSTALL") == "hostgb"                # ... with the groupby wrapper's PLANNING on (which calls would the device groupby take?): without a device every planned call then declines with "device groupby failed" and vaex answers
_outcomes = {}
_why = {}

if INSTALL:
    import time
    import vaex_amd
    for _attempt in range(20):       # (this process may be the first user of the GPU on a fresh box)
        if vaex_amd.superagg.device_count() > 0:
            break
        time.sleep(0.5)
    assert vaex_amd.superagg.device_count() > 0, "vaex_amd.install() needs a HIP device"
    vaex_amd.install()
elif HOST_LOGIC:
    import vaex_amd
    _backend = vaex_amd.install(hash_sets=False, legacy=False, groupby=HOST_GROUPBY)

    class _NoHip:
        def __getattr__(self, name):
            raise NotImplementedError("reftest: HIP classes switched off")
    _backend.__dict__["_hip"] = _NoHip()
    if HOST_GROUPBY:
        # the device groupby's stand-in without a GPU (as in tests/test_vaex_groupby.py): vaex_amd.binned.Frame driving the reference's own C++
        # (dense key ranges; predicates as host masks) — so that which calls the wrapper TAKES, and what it hands back for them, can be checked here
        import threading
        import numpy as _np
        from vaex_amd import binned as _binned, vaex_groupby as _vg
        from tests.test_golden_api import RefAdapter
        _ref = RefAdapter(vaex_amd._installed["cpu_module"])

        def _no_pack(*a, **kw):   # (several keys are packed on the device: no CPU stand-in — such calls decline here)
            raise NotImplementedError("reftest: multi-key packing needs the device")
        _ref.pack_keys = _no_pack

        class HostMaskFrame(_binned.Frame):
            def _selection_mask(self, selection):
                sel = super()._selection_mask(selection)
                if isinstance(sel, _binned._predicate.Predicate):
                    key = ("mask", sel.key())
                    if key not in self._predicates:
                        self._predicates[key] = sel.numpy_mask({c: self.columns[c] for c in sel.columns})
                    return self._predicates[key]
                return sel

            def groupby(self, *a, **kw):   # (scattered key ranges go through the HIP hash set's set_keys: no CPU stand-in — such calls decline here)
                try:
                    return super().groupby(*a, **kw)
                except AttributeError as e:
                    raise NotImplementedError(f"reftest: scattered keys need the device ({e})")

        class HostCollector:
            def __init__(self, plan, capacity):
                self.parts, self.lock, self.rows, self.capacity = [], threading.Lock(), 0, capacity

            def append(self, chunks):
                with self.lock:
                    self.parts.append({k: _np.array(v) for k, v in chunks.items()})   # (process() has turned the blocks into plain numpy: _block_as_numpy)
                    self.rows += len(next(iter(chunks.values())))

            def frame(self):
                return HostMaskFrame({k: _np.concatenate([p[k] for p in self.parts]) for k in self.parts[0]}, chunk_size=50_000, nthreads=2, superagg=_ref)
        _vg._frame_for = lambda df, columns: HostMaskFrame(dict(columns), chunk_size=50_000, nthreads=2, superagg=_ref)
        _vg._collector_for = lambda plan, capacity: HostCollector(plan, capacity)


def pytest_runtest_logreport(report):
    # one outcome per test: a failing setup / teardown counts as "error", a skip in any phase as "skipped"
    prev = _outcomes.get(report.nodeid)
    if report.when == "call":
        out = report.outcome if not hasattr(report, "wasxfail") else "xfail"
    elif report.failed:
        out = "error"
    elif report.skipped:
        out = "skipped" if not hasattr(report, "wasxfail") else "xfail"
    else:
        return
    if prev in ("failed", "error"):
        return
    _outcomes[report.nodeid] = out
    if out in ("failed", "error"):
        _why[report.nodeid] = str(report.longrepr)[-1500:]


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("VAEX_AMD_REFTEST_REPORT")
    if not path:
        return
    doc = {"install": INSTALL or HOST_LOGIC, "outcomes": _outcomes, "why": _why}
    if INSTALL or HOST_LOGIC:
        from vaex_amd import vaex_groupby, vaex_selection, vaex_filter
        doc["task_stats"] = {k: v for k, v in vaex_amd.task_stats.items() if isinstance(v, (int, float, str, dict))}
        doc["groupby"] = {"device": vaex_groupby.stats.get("device", 0), "task": vaex_groupby.stats.get("task", 0), "vaex": vaex_groupby.stats.get("vaex", 0), "why": vaex_groupby.stats.get("why", {}), "by_test": vaex_groupby.stats.get("by_test", {})}
        doc["selection"] = dict(vaex_selection.stats)
        doc["filter"] = dict(vaex_filter.stats)
    with open(path, "w") as f:
        json.dump(doc, f, default=str)
