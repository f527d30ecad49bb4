"""The reference's OWN test files of this path, unmodified, as the parity test ("the golden vectors, known-answer tests
and fixtures the reference's own tests hold for this path").

oracle/build_ref.sh copies seventy-seven of /root/reference/tests/*_test.py and vaex/test/cmodule.py (aggregations, count, groupby, selections, limits, percentiles,
grid, first, correlation, mutual information, filters, describe, countna, masked values, unique / value_counts / hashmap, concat,
slice; the executor, its progress and task cache; categories, datetimes, isin, join, map, sort; the rest of the frame API; arrow/) with their common.py / conftest.py into the git-ignored oracle/_ref/reftests/ — a build product like oracle/_ref/vaexpy, which is
the reference's unmodified Python package.  They run in a subprocess (tests/reftest_plugin.py stands in for vaex-server / pytest-asyncio,
which only the remote fixtures need):

  * here (no GPU) on vaex's C++ classes (oracle/_ref/*.so): the harness works, and most of each file passes — what does not is the
    oracle build's stubbed string classes, `vaex.example()` (a download) and xarray;
  * -m gpu: once on vaex's C++ (the baseline of THAT box) and once under vaex_amd.install(): every test that passes on the reference's
    classes must pass on the HIP classes — same assertions, same fixtures (big-endian and masked columns, filtered / sliced /
    concatenated / Arrow / parquet frames, 3-row chunks through `buffer_size`).  The report (gpurun_out/.../reference_suite.json when
    VAEX_AMD_REPORT_DIR is set) says how many task parts, selections, filters and groupbys ran on the device meanwhile."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFTESTS = os.path.join(ROOT, "oracle", "_ref", "reftests")
VAEXPY = os.path.join(ROOT, "oracle", "_ref", "vaexpy")
OVERLAY = os.path.join(ROOT, "oracle", "_ref", "overlay")
FAKE = os.path.join(ROOT, "oracle", "fake")
PKG = VAEXPY if os.path.isdir(os.path.join(VAEXPY, "vaex")) else OVERLAY

FILES = ["agg_test.py", "count_test.py", "groupby_test.py", "selection_test.py", "limits_test.py", "percentile_approx_test.py", "grid_test.py",
         "first_test.py", "correlation_test.py", "mutual_information_test.py", "filter_test.py", "describe_test.py", "countna_test.py",
         "masked_values_filters_test.py", "unique_test.py", "value_counts_test.py", "hashmap_unique_test.py", "concat_test.py", "slice_test.py",
         # the executor around the task parts (passes, progress, cancellation, the task cache), categoricals / datetimes as binners and values,
         # and what else reads the hash sets (isin, join, map, sort)
         "execution_test.py", "progress_test.py", "cache_test.py", "category_test.py", "datetime_test.py", "timedelta_test.py", "isin_test.py", "join_test.py",
         "dtypes_test.py", "nop_test.py", "trim_test.py", "dropna_test.py", "sort_test.py", "stack_test.py", "materialize_test.py", "map_test.py", "sparse_test.py",
         "fingerprint_test.py", "cornercases_test.py", "shape_test.py", "values_test.py", "internal/groupby_test.py", "internal/hash_test.py",
         # the rest of the frame API (nothing of it may change under install(); fillna / dropna / describe-like helpers schedule aggregations)
         "apply_test.py", "astype_test.py", "cast_to_array_test.py", "compute_test.py", "copy_test.py", "dataset_test.py", "derivative_test.py", "dot_product_test.py",
         "drop_test.py", "dropinf_test.py", "expression_variables_test.py", "extract_test.py", "fillna_test.py", "getattr_test.py", "indexing_test.py", "isna_test.py",
         "propagate_uncertainty_test.py", "rename_test.py", "rolling_test.py", "row_test.py", "split_test.py", "struct_test.py", "to_test.py", "utils_test.py",
         "variables_test.py", "evaluate_test.py", "column_test.py",
         "arrow/assumptions_test.py", "arrow/compute_test.py", "arrow/conversion_test.py", "arrow/convert_test.py", "arrow/dataset_test.py", "arrow/dict_test.py",
         "arrow/io_test.py", "arrow/to_arrow_table_test.py",
         "legacy/cmodule.py"]   # packages/vaex-core/vaex/test/cmodule.py: the unittest of vaexfast.statisticNd_f8 (install() puts the HIP entry there)

_SUBSET = [f for f in os.environ.get("VAEX_AMD_REFTEST_FILES", "").split(",") if f]   # (a quick check of the harness itself: a few files, no thresholds)
if _SUBSET:
    FILES = _SUBSET

pytestmark = pytest.mark.skipif(not (os.path.isfile(os.path.join(REFTESTS, "agg_test.py")) and os.path.isdir(os.path.join(PKG, "vaex"))),
                                reason="oracle/_ref/reftests or the reference's Python package not built (oracle/build_ref.sh needs /root/reference)")

#: tests that pass on vaex's C++ and are NOT expected to pass under install(), with the reason (none: the list is the claim)
EXPECTED_DIFFERENT = {}


def run_files(install, tmp_path, files=FILES, threads=4):
    report = str(tmp_path / ("report_%s.json" % (install if isinstance(install, str) else ("hip" if install else "cpp"))))
    env = dict(os.environ)
    env.update(PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), PKG, FAKE, ROOT]), PYTHONDONTWRITEBYTECODE="1", VAEX_TEST_SKIP_REMOTE="1",
               VAEX_NUM_THREADS=str(threads), VAEX_AMD_REFTEST_INSTALL=install if isinstance(install, str) else ("1" if install else "0"), VAEX_AMD_REFTEST_REPORT=report,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VAEX_AMD_REPORT_DIR", None)
    # (a copy of its own per run: the reference's parquet fixture writes data/unittest.parquet next to the tests, and two runs go side by side)
    import shutil
    import uuid
    work = str(tmp_path / ("reftests_" + uuid.uuid4().hex[:8]))
    shutil.copytree(REFTESTS, work)
    cmd = [sys.executable, "-m", "pytest", "-p", "reftest_plugin", "-q", "-p", "no:cacheprovider", "--tb=short", "--rootdir", work] + list(files)
    p = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=3000)
    shutil.rmtree(work, ignore_errors=True)
    assert os.path.exists(report), (p.stdout[-3000:], p.stderr[-3000:])
    doc = json.load(open(report))
    doc["tail"] = p.stdout[-600:]
    return doc


_cache = {}


def cached_run(install, tmp_path):
    """one run per mode and pytest process (a run is ~2 minutes; without a GPU the two modes of this file's tests run side by side)"""
    if install not in _cache:
        import torch
        if not torch.cuda.is_available() and not _cache:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(2) as pool:
                both = {mode: pool.submit(run_files, mode, tmp_path) for mode in (False, "host")}
                for mode, f in both.items():
                    _cache[mode] = f.result()
        else:
            _cache[install] = run_files(install, tmp_path)
    return _cache[install]


def counts(doc):
    c = {}
    for o in doc["outcomes"].values():
        c[o] = c.get(o, 0) + 1
    return c


def test_reference_files_run_against_the_reference_classes(tmp_path):
    doc = cached_run(False, tmp_path)
    c = counts(doc)
    # (this image: 1653 pass; the rest need the string hash classes the oracle build of the reference stubs out, or vaex.example(): a download)
    assert c.get("passed", 0) >= 1600, (c, doc["tail"])
    per_file = {}
    for node, o in doc["outcomes"].items():
        per_file.setdefault(node.split("::")[0], {}).setdefault(o, 0)
        per_file[node.split("::")[0]][o] += 1
    for f in ("selection_test.py", "limits_test.py", "grid_test.py", "first_test.py", "filter_test.py", "describe_test.py", "countna_test.py", "concat_test.py"):
        assert per_file[f].get("failed", 0) + per_file[f].get("error", 0) == 0, (f, per_file[f])


def test_what_passes_on_the_reference_classes_passes_through_the_host_logic_of_install(tmp_path):
    """no GPU: vaex_amd.install()'s HOST logic alone — the task part wrappers, filtered runs in the keep-mask form (here with host masks), named
    selections resolved at scheduling time, the per-task fallback — over vaex's own classes (the HIP classes switched off)"""
    base = cached_run(False, tmp_path)
    host = cached_run("host", tmp_path)
    passed = [n for n, o in base["outcomes"].items() if o == "passed"]
    regressions = {n: host["why"].get(n, host["outcomes"].get(n, "not run"))[-700:] for n in passed if host["outcomes"].get(n) != "passed"}
    allowed = sorted(n for n in regressions if any(a in n for a in SECOND_RUN_ALLOWED))   # (see the GPU test: only the reference's known nondeterministic tests get a second run)
    if allowed:
        again = run_files("host", tmp_path, files=allowed)
        regressions = {n: w for n, w in regressions.items() if again["outcomes"].get(n) != "passed"}
    assert host["filter"]["runs_switched"] > 500 and host["task_stats"]["cpu"] > 1000, (host.get("filter"), host.get("task_stats"))
    assert not regressions, (len(regressions), dict(list(regressions.items())[:8]))


#: tests of the reference that are nondeterministic ON THE REFERENCE'S OWN CLASSES and may take a second run alone (the list is the claim: anything
#: else that needs one fails the test — ADVICE r5: a blanket second run would hide an intermittent fault of install()).
#: execution_test.py::test_thread_safe enters the executor from four threads while the chunk size is being changed.
SECOND_RUN_ALLOWED = ("execution_test.py::test_thread_safe",)


@pytest.mark.gpu
def test_what_passes_on_the_reference_classes_passes_on_the_hip_classes(tmp_path):
    """ONE run on the reference's classes, ONE under install(): no run is repeated (round 5 repeated a run that died early and re-ran every
    regression once; round 6 looped install() as the first GPU user of 360 fresh processes without a failure — profiles/r06_first_user.txt —
    and took the allowances out).  A subprocess that dies leaves its whole output in the report directory."""
    out_dir = os.environ.get("VAEX_AMD_REPORT_DIR")
    try:
        base = cached_run(False, tmp_path)
        hip = run_files(True, tmp_path)
    except AssertionError as e:          # (the subprocess died before it wrote its report: keep everything it said)
        if out_dir:
            with open(os.path.join(out_dir, "reference_suite_lost_run.txt"), "w") as f:
                f.write(str(e))
        raise
    passed = [n for n, o in base["outcomes"].items() if o == "passed"]
    assert len(passed) >= 1600 or _SUBSET, counts(base)
    regressions = {n: hip["why"].get(n, hip["outcomes"].get(n, "not run"))[-700:] for n in passed if hip["outcomes"].get(n) != "passed" and n not in EXPECTED_DIFFERENT}
    retried = {}
    allowed = sorted(n for n in regressions if any(a in n for a in SECOND_RUN_ALLOWED))
    if allowed:
        again = run_files(True, tmp_path, files=allowed)
        retried = {n: again["outcomes"].get(n, "not run") for n in allowed}
        regressions = {n: w for n, w in regressions.items() if retried.get(n) != "passed"}
    fixed = [n for n, o in hip["outcomes"].items() if o == "passed" and base["outcomes"].get(n) in ("failed", "error")]
    summary = {"reference_classes": counts(base), "hip_classes": counts(hip), "pass_on_both": len(passed) - len(regressions), "regressions": regressions,
               "pass_only_under_install": fixed, "expected_different": EXPECTED_DIFFERENT, "second_runs": retried, "second_run_allowed": list(SECOND_RUN_ALLOWED),
               "task_parts": hip.get("task_stats"), "groupby": hip.get("groupby"), "selection": hip.get("selection"), "filter": hip.get("filter")}
    if out_dir:
        with open(os.path.join(out_dir, "reference_suite.json"), "w") as f:
            json.dump(summary, f, indent=1)
    assert hip["task_stats"]["hip"] > (20 if _SUBSET else 500), hip["task_stats"]       # the tests DID run on the HIP classes
    assert not regressions, (len(regressions), dict(list(regressions.items())[:8]))
