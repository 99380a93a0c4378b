"""'Which kernel ran' expectations of the GPU tests, judged AFTER the numeric comparisons.

A GPU test makes two kinds of statements: (i) the numbers agree with the oracle / the fixture / another kernel, and
(ii) the launch it is about really took the kernel it names (``n_chain_blk``, ``chain_fused``, ``n_lowsync`` ...).  With a
measurement switch flipped (``KRYPY_AMD_MGS_CHAIN=0``, ``KRYPY_AMD_CHAIN_BLK=0`` ...: tools/r05_evidence.sh fallback) statements of kind (ii)
are false BY CONSTRUCTION, and as plain ``assert`` lines in front of the comparisons they used to end the test before any
number had been compared - a fallback run then proved nothing.  ``expect_kernel`` records a false expectation instead of
raising; the autouse fixture in ``tests/conftest.py`` reports the recorded ones at teardown, i.e. only after the whole
test body - every numeric comparison included - has run and passed.  In a pytest summary they read

    ERROR tests/test_gpu_x.py::test_y - Failed: KERNEL-PATH EXPECTATION (all numeric comparisons of this test passed): ...

while a numeric failure stays a plain ``FAILED ... AssertionError``.  On the default configuration both are failures of
the run."""
_pending = []


def expect_kernel(cond, what):
    if not cond:
        _pending.append(str(what))
    return bool(cond)


def drain():
    out = list(_pending)
    del _pending[:]
    return out
