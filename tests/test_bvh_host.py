"""Host-side surface-area topology of the hierarchy (raytracer_amd/csrc/rt_bvh_host.hpp): the header is plain C++, so
its invariants are checked on the CPU by a small program compiled from tests/cpp/bvh_host_check.cpp (leaf order is a
permutation, inner nodes 0 .. n-2 all used with the root at 0, ranges contiguous and consistent with the children,
cross-block list exact, the same tree for any number of threads; soups, a grid, coincident triangles, n = 2 .. 70 000).
No counterpart in the reference, which has no hierarchy."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_topology_invariants(tmp_path):
    exe = tmp_path / "bvh_host_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread",
                    f"-I{ROOT}/raytracer_amd/csrc", "-o", str(exe), f"{ROOT}/tests/cpp/bvh_host_check.cpp"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().endswith("ok"), out.stdout + out.stderr
