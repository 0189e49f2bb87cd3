"""The hot kernels must fit eight waves on a SIMD: 64 VGPRs, 80 SGPRs (a SIMD has 800 and the hardware keeps 16 more per wave
for the trap handler, so a wave that names 96 leaves room for seven), and LDS for two path-kernel workgroups or eight packet
workgroups per CU.  A kernel that outgrows this still computes the same pixels -- only slower, and no other test notices.
Read from the compiler's own metadata (make print-resources compiles the device code, nothing runs)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opengl-raytracer_amd", "csrc")


def kernel_resources():
    out = subprocess.run(["make", "-s", "-C", CSRC, "print-resources"], check=True, capture_output=True, text=True, timeout=600).stdout
    kernels, cur = {}, {}
    for line in out.splitlines():
        key, _, val = line.strip().lstrip(".").partition(":")
        cur[key] = val.strip()
        if key == "vgpr_count":   # the last of a kernel's fields in the metadata (alphabetical order)
            kernels[cur["name"]] = {k: int(v) for k, v in cur.items() if k != "name"}
            cur = {}
    return kernels


@pytest.mark.skipif(shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs make and hipcc")
def test_hot_kernels_fit_eight_waves_per_simd():
    ks = kernel_resources()
    paths = {n: r for n, r in ks.items() if n.startswith("_Z7k_pathsI")}
    primary = {n: r for n, r in ks.items() if n.startswith("_Z9k_primaryI")}
    assert len(paths) >= 16 and len(primary) == 2, (sorted(paths), sorted(primary))
    for n, r in {**paths, **primary}.items():
        assert r["vgpr_count"] <= 64, (n, r)
        assert r["sgpr_count"] <= 80, (n, r)
    for n, r in paths.items():
        assert 2 * r["group_segment_fixed_size"] <= 160 * 1024, (n, r)
    for n, r in primary.items():
        assert 8 * r["group_segment_fixed_size"] <= 160 * 1024, (n, r)
    # the timed instantiations (the packet kernel and the path kernel fed from its hit queues) keep nothing in scratch
    assert ks["_Z9k_primaryILb0EEv6DScene7DCamera9PathsArgs"]["private_segment_fixed_size"] == 0
    assert ks["_Z7k_pathsILb0ELb0ELi3EEv6DScene7DCamera9PathsArgs"]["private_segment_fixed_size"] == 0
