#!/usr/bin/env python3
"""Generates tests/golden/comms_rows.json: the COMMS-RES header and the deterministic columns (collective, dtype, message
size, elements per rank) of the rows the REFERENCE's ``train/comms/pt/comms.py`` prints on 2 gloo ranks for
``--b 64 --e 1024 --f 4 --z 1 --c 1 --collective all_to_allv,all_reduce``.  Needs /root/reference (build container only)."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
work = tempfile.mkdtemp()
os.makedirs(os.path.join(work, "pb"))
os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
port = "29563"
env = dict(os.environ, PYTHONPATH=os.path.join(work, "pb"), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", LOCAL_SIZE="2")
cmd = [sys.executable, "/root/reference/train/comms/pt/comms.py", "--master-ip", "127.0.0.1", "--master-port", port, "--b", "64",
       "--e", "1024", "--f", "4", "--n", "5", "--w", "2", "--z", "1", "--c", "1", "--collective", "all_to_allv,all_reduce",
       "--backend", "gloo", "--device", "cpu"]
procs = [subprocess.Popen(cmd, cwd=work, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                          stderr=subprocess.DEVNULL, text=True) for r in (0, 1)]
outs = [p.communicate(timeout=600)[0] for p in procs]
assert [p.returncode for p in procs] == [0, 0]
lines = [ln for ln in outs[0].splitlines() if "COMMS-RES" in ln]
header = [ln for ln in lines if "total-size" in ln][0]
rows = []
for ln in lines:
    m = re.match(r"\tCOMMS-RES-(\w+)-(\w+)\s+(\d+)\s+(\d+)\s", ln)
    if m:
        rows.append({"collective": m.group(1), "dtype": m.group(2), "size": int(m.group(3)), "elements_per_rank": int(m.group(4)),
                     "columns": len(ln.split())})
json.dump({"args": cmd[2:], "header": header, "rows": rows, "rank1_prints_rows": any("COMMS-RES" in ln for ln in outs[1].splitlines())},
          open(os.path.join(HERE, "comms_rows.json"), "w"), indent=1)
print(header)
print(rows)
