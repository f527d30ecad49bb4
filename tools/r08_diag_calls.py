"""Diagnosis runs of tests/test_vaex_random_calls.py's script on the GPU box: VAEX_AMD_RANDOM_CALL_RANGE=lo:hi (these calls only, what execute() raised printed),
VAEX_AMD_RANDOM_FORCE_MOVED=i (the reference's two answers of call i spoilt on purpose: exercises the one-pool-thread tie-break)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_vaex_random_calls as t
try:
    out = t._run(1, int(os.environ.get("VAEX_AMD_RANDOM_CALLS", "1")), 800)
    print(out[-6000:])
except AssertionError as e:
    print(str(e)[-9000:])
