"""The reference's OWN unit tests, run against this build's modules (CPU; build container only -- skipped where /root/reference
does not exist, e.g. on the GPU box; nothing of the reference is stored in the repo: the harness loads the test files from the
reference tree at run time and points their imports at ``param_amd``)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "train", "comms", "pt", "tests")), reason="needs /root/reference")
def test_reference_unit_tests_pass_against_this_build():
    """``train/comms/pt/tests/comms_utils_tests.py`` (22 tests: parsesize, parseRankList, getAlgBW, getSizes, fixBeginSize,
    get_rank_details, env2int, read_comms_env_vars, paramToCommName, ensureTensorFlush), ``commsTraceReplay_tests.py`` (13 tests:
    prepComms with and without shrinking, replayTrace, runComms blocking / non-blocking, initTraceStat, initBench, rebalance) and
    ``train/compute/python/test/test_register.py`` (operator / iterator / generator registries; the operator registry is this
    build's) -- every one passes with ``param_bench.train.comms.pt.*`` resolving to ``param_amd.comms.pt.*``."""
    cmd = [sys.executable, os.path.join(ROOT, "tests", "ref_unittest_harness.py"), ROOT, REF,
           "comms:comms_utils_tests", "comms:commsTraceReplay_tests", "compute:test_register"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert {k: v["run"] for k, v in out.items()} == {"comms_utils_tests": 22, "commsTraceReplay_tests": 13, "test_register": 3}
    for name, res in out.items():
        assert res["failures"] == [] and res["errors"] == [], (name, res)
