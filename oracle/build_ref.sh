#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds the reference's own native code, in place, as the parity oracle.
#
# Compiles vaex-core's superagg / superutils / vaexfast (+ superstrings for `import vaex`)
# straight from /root/reference/packages/vaex-core/src (nothing is copied into this repo)
# with g++ and the reference's own flags (-std=c++17 -O3 -funroll-loops -DVAEX_USE_TSL,
# packages/vaex-core/setup.py:101-123), against the shim headers in oracle/shim that stand
# in for the un-vendored git submodules (string-view-lite, hopscotch-map, pcre).
#
# Outputs ONLY into oracle/_ref/ (git-ignored, travels to the GPU box with gpurun):
#   oracle/_ref/superagg<ext>.so  superutils<ext>.so  vaexfast<ext>.so  superstrings<ext>.so
#   oracle/_ref/overlay/          symlink overlay of the reference's pure-Python package
#                                 (needed only HERE by oracle/make_goldens.py; gpurun-ignored)
#   oracle/_ref/vaexpy/           the same package as plain files (travels to the GPU box: the drop-in test there)
#   oracle/_ref/reftests/         the reference's own test files of this path (travel too: tests/test_vaex_reference_suite.py)
#
# Usage: oracle/build_ref.sh [--minimal]   (--minimal: superagg only)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/packages/vaex-core
SRC=$REF/src
OUT=$HERE/_ref
OBJ=$OUT/obj
if [ ! -d "$SRC" ]; then
    echo "build_ref: $SRC not present (GPU box?) - using prebuilt files in $OUT if any"; exit 0
fi
mkdir -p "$OBJ"
PYINC=$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')
NPINC=$(python3 -c 'import numpy;print(numpy.get_include())')
PBINC=$(python3 -c 'import pybind11;print(pybind11.get_include())')
EXT=$(python3 -c 'import sysconfig;print(sysconfig.get_config_var("EXT_SUFFIX"))')
CXX="g++ -std=c++17 -O3 -funroll-loops -fPIC -w -DVAEX_USE_TSL -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I$HERE/shim -I$PYINC -I$NPINC -I$PBINC"

compile() { # $1 = source path, $2 = object name; skipped when up to date
    if [ ! -f "$OBJ/$2.o" ] || [ "$1" -nt "$OBJ/$2.o" ]; then
        $CXX ${3:-} -c "$1" -o "$OBJ/$2.o"
    fi
}

AGG="agg agg_count agg_sum agg_minmax agg_first agg_list agg_nunique binners binner_ordinal binner_combined binner_hash string_utils"
pids=()
for f in $AGG; do compile $SRC/$f.cpp $f & pids+=($!); done
compile $HERE/stubs/stub_superagg.cpp stub_superagg & pids+=($!)
if [ "${1:-}" != "--minimal" ]; then
    compile $SRC/superutils.cpp superutils & pids+=($!)
    compile $SRC/hash_primitives_pot.cpp hash_primitives_pot & pids+=($!)
    compile $HERE/stubs/stub_superutils.cpp stub_superutils & pids+=($!)
    compile $SRC/strings.cpp strings "-I$REF/vendor/boost" & pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done

g++ -shared -o $OUT/superagg$EXT $(for f in $AGG stub_superagg; do echo $OBJ/$f.o; done)
if [ "${1:-}" != "--minimal" ]; then
    g++ -shared -o $OUT/superutils$EXT $OBJ/superutils.o $OBJ/hash_primitives_pot.o $OBJ/string_utils.o $OBJ/stub_superutils.o
    g++ -shared -o $OUT/superstrings$EXT $OBJ/strings.o $OBJ/string_utils.o
    if [ ! -f $OUT/vaexfast$EXT ] || [ $SRC/vaexfast.cpp -nt $OUT/vaexfast$EXT ]; then
        $CXX -shared $SRC/vaexfast.cpp -o $OUT/vaexfast$EXT
    fi
    # pure-Python overlay (symlinks into /root/reference; never committed, never shipped)
    rm -rf $OUT/overlay
    mkdir -p $OUT/overlay
    cp -rs $REF/vaex $OUT/overlay/vaex
    for m in superagg superutils superstrings vaexfast; do ln -sf $OUT/$m$EXT $OUT/overlay/vaex/$m$EXT; done
    DI=$OUT/overlay/vaex_core-4.19.0.dist-info
    mkdir -p $DI
    printf 'Metadata-Version: 2.1\nName: vaex-core\nVersion: 4.19.0\n' > $DI/METADATA
    cat > $DI/entry_points.txt <<'EOF'
[vaex.memory.tracker]
default = vaex.memory:MemoryTracker

[vaex.progressbar]
vaex = vaex.progress:simple
simple = vaex.progress:simple
widget = vaex.progress:widget
rich = vaex.progress:rich

[vaex.dataframe.accessor]
struct = vaex.struct:DataFrameAccessorStruct

[vaex.dataset.opener]
csv = vaex.csv:DatasetCsvLazy
arrow = vaex.arrow.opener:ArrowOpener
parquet = vaex.arrow.opener:ParquetOpener
feather = vaex.arrow.opener:FeatherOpener
EOF
    # The same package with the files themselves instead of symlinks (pure-Python modules only, no tests / images /
    # datasets): what travels to the GPU box, where /root/reference does not exist, so that tests/test_vaex_dropin.py can
    # drive the HIP classes through an UNMODIFIED vaex there.  Build output like the .so files above: under the
    # git-ignored oracle/_ref/, never committed; test infrastructure, never imported by the product.
    rm -rf $OUT/vaexpy
    mkdir -p $OUT/vaexpy
    (cd $REF && find vaex -name '*.py' -not -path 'vaex/test/*' -print0 | xargs -0 cp --parents -t $OUT/vaexpy)
    for m in superagg superutils superstrings vaexfast; do ln -sf ../../$m$EXT $OUT/vaexpy/vaex/$m$EXT; done
    cp -r $DI $OUT/vaexpy/
    # The reference's OWN test files of this path (aggregations, binby, groupby, selections, limits, percentiles, unique /
    # value_counts), as they are, for tests/test_vaex_reference_suite.py: run there against vaex's C++ (the baseline) and, on the
    # GPU box, under vaex_amd.install() — what passes on the reference's classes must pass on the HIP classes.  Build output like
    # vaexpy above: git-ignored, never committed, never imported by the product.  (They are RUN from this copy, never in place:
    # the fixtures write a parquet file next to themselves.)
    rm -rf $OUT/reftests
    mkdir -p $OUT/reftests/data
    for t in common.py conftest.py agg_test.py count_test.py groupby_test.py selection_test.py limits_test.py percentile_approx_test.py \
             grid_test.py first_test.py correlation_test.py mutual_information_test.py filter_test.py describe_test.py countna_test.py \
             masked_values_filters_test.py unique_test.py value_counts_test.py hashmap_unique_test.py concat_test.py slice_test.py \
             execution_test.py progress_test.py cache_test.py category_test.py datetime_test.py timedelta_test.py isin_test.py join_test.py \
             dtypes_test.py nop_test.py trim_test.py dropna_test.py sort_test.py stack_test.py materialize_test.py map_test.py sparse_test.py \
             fingerprint_test.py cornercases_test.py shape_test.py values_test.py \
             apply_test.py astype_test.py cast_to_array_test.py compute_test.py copy_test.py dataset_test.py derivative_test.py dot_product_test.py \
             drop_test.py dropinf_test.py expression_variables_test.py extract_test.py fillna_test.py getattr_test.py indexing_test.py isna_test.py \
             propagate_uncertainty_test.py rename_test.py rolling_test.py row_test.py split_test.py struct_test.py to_test.py utils_test.py \
             variables_test.py evaluate_test.py column_test.py; do
        cp /root/reference/tests/$t $OUT/reftests/
    done
    mkdir -p $OUT/reftests/internal $OUT/reftests/arrow $OUT/reftests/legacy
    cp $REF/vaex/test/cmodule.py $OUT/reftests/legacy/   # the unittest of vaexfast.statisticNd_f8: add / weights / moments / edges
    for t in __init__.py groupby_test.py hash_test.py; do cp /root/reference/tests/internal/$t $OUT/reftests/internal/; done
    for t in __init__.py assumptions_test.py compute_test.py conversion_test.py convert_test.py dataset_test.py dict_test.py io_test.py to_arrow_table_test.py; do
        cp /root/reference/tests/arrow/$t $OUT/reftests/arrow/
    done
fi
echo "build_ref: done -> $OUT"
