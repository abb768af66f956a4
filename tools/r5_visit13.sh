#!/bin/bash
# tools/r5_visit13.sh: the N > 1 flow of bench.py with REAL ranks on a one-GPU box (PARAM_AMD_BENCH_SHARED_GPU=1: every rank on GPU 0, gloo):
# 2 ranks x 8 tables, 4 ranks x 26 tables (uneven partition [7,7,6,6]), Criteo on 2 ranks, the blocked send layout -- a flow check
t=${1:-r5_v13}; out=gpurun_out/$t; mkdir -p $out
export PARAM_AMD_BENCH_SHARED_GPU=1
run() { name=$1; n=$2; shift 2
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $n --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  echo "== $name rc=$?"; tail -c 600 $out/$name.json; echo; grep -iE "error|Traceback|assert" $out/$name.err | head -5; }
run n2_16tables 2 --tables 16 --rows 2000000
run n4_26tables 4 --tables 26 --rows 1000000
run n2_criteo 2 --workload criteo --batch 2048
run n2_blocked 2 --tables 16 --rows 2000000 --send-layout blocked
# the self-launch form: no launcher, bench.py starts its own ranks
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --tables 16 --rows 2000000 > $out/selflaunch.json 2> $out/selflaunch.err; echo "== selflaunch rc=$?"; tail -c 300 $out/selflaunch.json; echo
