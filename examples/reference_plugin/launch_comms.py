#!/usr/bin/env python3
"""Runs the REFERENCE's unmodified ``train/comms/pt/comms.py`` with this build's comms plug-in registered.

    PYTHONPATH=<dir holding param_bench/ -> the PARAM checkout>:<this repo> \
    python examples/reference_plugin/launch_comms.py <path to PARAM>/train/comms/pt/comms.py \
        --master-ip 127.0.0.1 --master-port 29500 --b 8 --e 256M --f 2 --n 100 --z 1 \
        --collective all_to_all --backend rccl_xgmi --device cuda

One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_SIZE / MASTER_ADDR / MASTER_PORT in the environment, as for
the reference itself).  ``--device cpu`` moves host tensors over gloo instead of RCCL: that is how
tests/golden/gen_ref_plugin_rows.py runs exactly this file on 2 ranks in a container without GPUs.

What it does: ``register()`` puts ``MI355XBackend`` into ``param_bench.train.comms.pt.pytorch_backend_utils.
customized_backend`` under the name ``rccl_xgmi`` (reference ``comms.py:1507-1524`` looks a non-c10d ``--backend`` up
there), then the reference's script runs as ``__main__``.  Nothing of the reference is modified."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from param_amd.comms.pt import mi355_backend  # noqa: E402

if len(sys.argv) < 2 or not sys.argv[1].endswith(".py"):
    raise SystemExit(__doc__)
mi355_backend.register()
script = sys.argv.pop(1)
sys.argv[0] = script
runpy.run_path(script, run_name="__main__")
