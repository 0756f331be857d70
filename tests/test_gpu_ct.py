"""Constant-time evidence for the secret-scalar kernels (the discipline being claimed is the
reference's scale16, lib/ed.c:346-391: no branch and no address depends on a secret digit): the hardware counters
of k_x25519_base_point, k_genpub_point and k_sign_point - VALU / SALU / LDS / SMEM / VMEM instruction counts and
LDS bank-conflict cycles - must be IDENTICAL whether the 2^16 (one lane per item) or 2^12 (four lanes per item) secrets are all zero, all ones, random, or a
different class in every lane.  Runs tools/ct_counters.sh (rocprofv3 --pmc passes, no tracing) as a child."""
import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_point_kernels_counters_do_not_depend_on_the_secrets(engine, tmp_path):
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 is not installed on this box")
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT)
    stale = os.path.join(ROOT, "gpurun_out", "profiles_out", "pytest_ct_counters.json")
    if os.path.exists(stale):
        os.remove(stale)
    r = subprocess.run([os.path.join(ROOT, "tools", "ct_counters.sh"), "pytest"], env=env, capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    out = os.path.join(ROOT, "gpurun_out", "profiles_out", "pytest_ct_counters.json")
    if r.returncode != 0 or not os.path.exists(out):        # the profiler could not run here: nothing to compare
        pytest.skip("rocprofv3 --pmc did not produce counters on this box: " + (r.stdout + r.stderr)[-300:])
    d = json.load(open(out))
    if not all(d["counters"].get(c) for c in ("zero", "ones", "random", "mixed")):
        pytest.skip("rocprofv3 collected no counters for some class on this box")
    same = d["identical_across_secret_classes"]
    assert set(same) == {k + w for k in ("ed::k_x25519_base_point", "ed::k_genpub_point", "ed::k_sign_point")
                         for w in ("<1>", "<4>")}, same             # one lane per item, and the four-lane form of small passes
    assert all(same.values()), d["counters"]
    c = d["counters"]["random"]["ed::k_sign_point<1>"]
    assert c["SQ_INSTS_VALU"][0] > 1e6 and len(c["SQ_LDS_BANK_CONFLICT"]) == 1     # one value over all launches
