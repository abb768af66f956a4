#!/bin/bash
# fifth GPU visit: driver-layer tests, N>1 code path on a 1-rank RCCL group, DLRM + overlap drivers from the CLI
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_drivers.py -m gpu -q > $OUT/pytest_gpu_e.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_e.log
timeout 900 python bench.py --steps 20 --warmup 3 --dist-debug --no-cpu-baseline > $OUT/bench_r1e_distdebug.json 2> $OUT/bench_r1e_distdebug.err; echo "rc=$?" >> $OUT/bench_r1e_distdebug.err
# the driver's own launch form at N=1 under torch.distributed.run (env-provided rank/world)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-uniform --tables 8 > $OUT/bench_r1e_torchrun.json 2> $OUT/bench_r1e_torchrun.err; echo "rc=$?" >> $OUT/bench_r1e_torchrun.err
timeout 600 python -m param_amd.comms.pt.comms --master-port 29612 --b 1M --e 64M --f 4 --n 20 --w 5 --z 1 --c 1 \
    --collective all_to_all,all_to_allv --backend rccl_xgmi --device rocm > $OUT/comms_cli.log 2>&1; echo "rc=$?" >> $OUT/comms_cli.log
timeout 600 python -m param_amd.compute.pt.pytorch_emb --features 14000000 --embdim 128 --nnz 30 --batch 16384 --steps 50 --warmups 5 -d gpu --json > $OUT/emb_cli.log 2>&1; echo "rc=$?" >> $OUT/emb_cli.log
timeout 600 python -m param_amd.compute.pt.pytorch_emb --features 14000000 --embdim 128 --nnz 30 --batch 16384 --steps 20 --warmups 2 -d cpu --no-grad >> $OUT/emb_cli.log 2>&1
tail -4 $OUT/pytest_gpu_e.log
python - <<EOF
import json
for f in ("bench_r1e_distdebug","bench_r1e_torchrun"):
    try:
        r=json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.3g" % r["value"], "groups/workload:", r["config"]["workload"][-90:], "a2a", r.get("all_to_all"))
    except Exception as e:
        print(f, "FAILED", e); print(open("$OUT/%s.err" % f).read()[-2000:])
EOF
tail -6 $OUT/comms_cli.log; tail -8 $OUT/emb_cli.log
