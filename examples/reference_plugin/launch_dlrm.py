#!/usr/bin/env python3
"""Runs the REFERENCE's ``train/comms/pt/dlrm.py`` (the DLRM comms benchmark) on this build's backend.

    PYTHONPATH=<dir holding param_bench/>:<PARAM>/train/comms/pt:<this repo> \
    python examples/reference_plugin/launch_dlrm.py --master-ip 127.0.0.1 --master-port 29500 --backend rccl_xgmi \
        --device cuda --mini-batch-size 8192 --arch-sparse-feature-size 128 --arch-embedding-size 10000000-...-10000000 \
        --num-indices-per-lookup 20 --num-indices-per-lookup-fixed

Two things the reference's own ``__main__`` cannot do by itself:
  * ``dlrm.py:1327`` instantiates ``PyTorchDistBackend`` by name (there is no ``customized_backend`` look-up in this
    driver): the module attribute is re-bound to ``MI355XBackend`` -- the one line a maintainer would change;
  * ``dlrm.py`` reads ``args.use_device_time`` but never declares the flag (SURVEY.md bug R1): it is pre-declared here.
``PARAM_AMD_HOST_TABLES=1`` (used by tests/golden/gen_ref_plugin_rows.py, which has no GPU) replaces the embedding tables
by torch's CPU EmbeddingBag so that the collectives of the DLRM iteration can be exercised on gloo; the product lookup has
no CPU path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from param_amd.comms.pt import mi355_backend  # noqa: E402

import dlrm  # noqa: E402  (the reference's train/comms/pt/dlrm.py, found on PYTHONPATH)
from param_bench.train.comms.pt import comms_utils  # noqa: E402

backend_class = mi355_backend.MI355XBackend
if os.environ.get("PARAM_AMD_HOST_TABLES") == "1":
    class backend_class(mi355_backend.MI355XBackend):   # noqa: F811  (test stand-in: see the module docstring)
        def alloc_embedding_tables(self, n, m, curRankDevice, dtype):
            return torch.nn.EmbeddingBag(n, m, mode="sum", sparse=False).to(curRankDevice)

dlrm.PyTorchDistBackend = backend_class
mi355_backend.register()
env = comms_utils.read_comms_env_vars()
bench = dlrm.commsDLRMBench()
parser = argparse.ArgumentParser()
parser.add_argument("--use-device-time", action="store_true", default=False)
args = bench.readArgs(parser)
bench.checkArgs(args)
bench.initBench(args, env)
info = comms_utils.bootstrap_info_holder(args.master_ip, args.master_port, args.num_tpu_cores, env)
bench.runBench(info, comms_utils.commsDlrmParamsHolder(args, env), args)
