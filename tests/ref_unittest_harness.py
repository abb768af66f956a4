"""Runs the REFERENCE's own unit tests against THIS build's modules (build container only: needs /root/reference).

``param_bench.train.comms.pt.<module>`` is made to resolve to ``param_amd.comms.pt.<module>`` and
``param_bench.train.compute.python.lib.operator`` to ``param_amd.compute.python.operator``; only the test files, their mocks and
fixtures -- and, for the compute tests, the framework modules this build does not replace -- are loaded from the reference tree.
Prints one JSON object: per test module, tests run / failures / errors.  Used by tests/test_reference_unit_tests.py.

    python tests/ref_unittest_harness.py <repo> <reference> comms:comms_utils_tests comms:commsTraceReplay_tests compute:test_register
"""
import importlib
import json
import sys
import types
import unittest


def pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    if "." in name:
        parent, leaf = name.rsplit(".", 1)
        setattr(sys.modules[parent], leaf, m)
    return m


def alias(ref_name, mine):
    mod = importlib.import_module(mine)
    sys.modules[ref_name] = mod
    parent, leaf = ref_name.rsplit(".", 1)
    setattr(sys.modules[parent], leaf, mod)


def main():
    repo, ref = sys.argv[1], sys.argv[2]
    sys.path.insert(0, repo)
    for n in ("param_bench", "param_bench.train", "param_bench.train.comms", "param_bench.train.compute"):
        pkg(n)
    pkg("param_bench.train.comms.pt")
    for sub in ("comms_utils", "pytorch_backend_utils", "commsTraceReplay", "commsTraceParser", "comms", "logger_utils"):
        alias(f"param_bench.train.comms.pt.{sub}", f"param_amd.comms.pt.{sub}")
    pkg("param_bench.train.comms.pt.tests", f"{ref}/train/comms/pt/tests")
    pkg("param_bench.train.comms.pt.tests.mocks", f"{ref}/train/comms/pt/tests/mocks")
    # compute/python: the reference's framework package, with the operator registry replaced by this build's
    pkg("param_bench.train.compute.python", f"{ref}/train/compute/python")
    pkg("param_bench.train.compute.python.lib", f"{ref}/train/compute/python/lib")
    alias("param_bench.train.compute.python.lib.operator", "param_amd.compute.python.operator")
    pkg("param_bench.train.compute.python.test", f"{ref}/train/compute/python/test")
    prefix = {"comms": "param_bench.train.comms.pt.tests.", "compute": "param_bench.train.compute.python.test."}
    out = {}
    for item in sys.argv[3:]:
        kind, name = item.split(":")
        mod = importlib.import_module(prefix[kind] + name)
        suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
        res = unittest.TextTestRunner(verbosity=0, stream=open("/dev/null", "w")).run(suite)
        out[name] = {"run": res.testsRun, "failures": [[str(t), e.splitlines()[-1]] for t, e in res.failures],
                     "errors": [[str(t), e.splitlines()[-1]] for t, e in res.errors], "skipped": len(res.skipped)}
    sys.__stdout__.write("\n" + json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
