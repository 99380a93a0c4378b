"""bench.py with the NumPy test double standing in for the device - TEST INFRASTRUCTURE, CPU suite only.

``python tests/support/bench_double.py --gpus 2 ...`` is ``python bench.py --gpus 2 ...`` in a container without a GPU:
the double is installed as the process-wide context and the visible-device count is faked (``BENCH_DOUBLE_DEVICES``,
default 8), then ``bench.main()`` runs unchanged.  Because bench.py's own launcher re-executes ``sys.argv`` for each rank,
every rank process it starts comes through this file again and gets the double too: the launcher-free path
(`bench._launch`) is exercised exactly as the driver would hit it, process tree and all.  The product has no hook
for this - it is all done from the outside, here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    # (a rank process that dies before the rendezvous, for the launcher's failed-rank test)
    if os.environ.get("BENCH_DOUBLE_DIE_RANK") is not None and os.environ.get("RANK") == os.environ["BENCH_DOUBLE_DIE_RANK"]:
        sys.stderr.write("bench_double: rank %s told to die\n" % os.environ["RANK"])
        sys.exit(3)
    from krypy_amd import _hip
    from tests.support.numpy_context import NumpyContext

    fail_at = os.environ.get("BENCH_DOUBLE_FAIL_MGS_AT")
    if fail_at is not None:
        # every rank's reference-order step fails at the same k (what a timed-out in-launch sum looks like from the host: an
        # error on all ranks of the communicator) - for the test of `--ortho auto`'s way back to the panel form
        orig = NumpyContext.arnoldi_step

        def failing(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1=0.0):
            if gs_mode == _hip.GS_MGS and k == int(fail_at):
                raise _hip.BackendError("bench_double: the reference-order step was told to fail at k = %d" % k)
            return orig(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1)

        NumpyContext.arnoldi_step = failing
    fail_after = os.environ.get("BENCH_DOUBLE_FAIL_MGS_AFTER")
    if fail_after is not None:
        # every rank's reference-order steps start failing after the same number of calls - a sum over the mailboxes that times out
        # INSIDE the timed region (the probe, if any, has passed by then): for the test of the timed region's way back to the panel form
        orig2 = NumpyContext.arnoldi_step
        calls = [0]

        def failing2(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1=0.0):
            if gs_mode == _hip.GS_MGS:
                calls[0] += 1
                if calls[0] > int(fail_after):
                    raise _hip.BackendError("bench_double: reference-order step %d of this process was told to fail" % calls[0])
            return orig2(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1)

        NumpyContext.arnoldi_step = failing2
    _hip._install_context_for_testing(NumpyContext())
    _hip.device_count = lambda: int(os.environ.get("BENCH_DOUBLE_DEVICES", "8"))
    import bench

    bench.main()


if __name__ == "__main__":
    main()
