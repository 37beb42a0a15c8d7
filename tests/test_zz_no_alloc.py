"""hipexec_any_hits_static (device pointers, the library's fixed work memory, no allocation) against
hipexec_any_hits and against the CPU oracle on the same rays: identical occlusion flags, on two calls in a
row, the second with a smaller input (the fixed array keeps the first call's contents), for both algorithms
and both rule sets, plus the size limit and the free device memory before and after a call.  Device memory for the rays and flags is the test's own (plain hipMalloc)."""
import numpy as np
import pytest

from raytracer_amd import _abi, scenegen

pytestmark = pytest.mark.gpu

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL


from tests.util import HipMem as _Hip


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform([-8, 0.5, -14], [8, 8, 6], size=(n, 3))
    tgt = rng.uniform([-10, -0.5, -18], [10, 3, 2], size=(n, 3))
    d = (tgt - o) * rng.uniform(0.2, 3.0, size=(n, 1))
    return np.concatenate([o, d], axis=1).astype(np.float32)


def _static(ex, hip, d_rays, d_hits, rays, pre, mode):
    hip.h2d(d_rays, rays)
    hip.h2d(d_hits, pre)
    hip.sync()                                   # the executor's stream is not ordered against the copies above
    ex.any_hits_static(d_rays, rays.shape[0], d_hits, mode)
    ex.sync()
    out = np.empty(rays.shape[0], np.int8)
    hip.d2h(out, d_hits)
    return out


@pytest.mark.parametrize("algo", ["bvh", "scan"])
@pytest.mark.parametrize("mode", [CPU, CL])
def test_static_any_hits_equals_any_hits_twice(mode, algo):
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    hip = _Hip()
    fs = scenegen.small_variant("c2").to_flat()
    orc = Oracle(fs)
    n1, n2 = 70001, 4099                          # neither a multiple of the block size; second call smaller
    d_rays, d_hits = hip.malloc(n1 * 24), hip.malloc(n1)
    try:
        with HipExecutor(fs, device=0) as ex:
            ex.set_algorithm(algo)
            for n, seed in ((n1, 5), (n2, 6)):
                rays = _rays(n, seed)
                pre = (np.arange(n) % 5 == 0).astype(np.int8)
                want = ex.any_hits(rays, mode, hits=pre.copy())
                ref = orc.any_hits(rays, mode, hits=pre.copy())
                free_before = hip.mem_free()
                got = _static(ex, hip, d_rays, d_hits, rays, pre, mode)
                assert hip.mem_free() == free_before, "the call changed the free device memory"
                assert 0 < int(want.sum()) < n
                assert np.array_equal(got, ref), f"{algo} mode {mode}: {(got != ref).sum()} of {n} flags differ from the oracle's"
                assert np.array_equal(got, want), f"{algo} mode {mode}: {(got != want).sum()} of {n} flags differ"
    finally:
        hip.free(d_rays); hip.free(d_hits)


def test_static_any_hits_limits():
    from raytracer_amd.executor import HipExecutor, HipExecutorError
    hip = _Hip()
    fs = scenegen.small_variant("c1").to_flat()
    d = hip.malloc(64)
    try:
        with HipExecutor(fs, device=0) as ex:
            ex.any_hits_static(d, 0, d, CL)                                  # n = 0 is legal and does nothing
            with pytest.raises(HipExecutorError, match="exceed"):
                ex.any_hits_static(d, _abi.STATIC_MAX_RAYS + 1, d, CL)
            with pytest.raises(HipExecutorError):
                ex.any_hits_static(0, 8, d, CL)
            # the largest legal size
            n = _abi.STATIC_MAX_RAYS
            rays = _rays(n, 9)
            d_rays, d_hits = hip.malloc(n * 24), hip.malloc(n)
            try:
                pre = np.zeros(n, np.int8)
                want = ex.any_hits(rays, CPU, hits=pre.copy())
                got = _static(ex, hip, d_rays, d_hits, rays, pre, CPU)
                assert np.array_equal(got, want)
            finally:
                hip.free(d_rays); hip.free(d_hits)
    finally:
        hip.free(d)
