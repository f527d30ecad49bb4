"""Diagnosis run of tests/test_vaex_random_groupby.py's script on the GPU box: VAEX_AMD_RANDOM_SEEDS=a,b,c (these seeds only; tracebacks, both results printed)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_vaex_random_groupby as t
try:
    print(t._run(1, 1, 800)[-8000:])
except AssertionError as e:
    print(str(e)[-12000:])
