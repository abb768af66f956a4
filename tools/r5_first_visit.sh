#!/bin/bash
# tools/r5_first_visit.sh [tag]: first GPU visit of round 5.  Round 4 ended without GPU minutes for its last additions (the host
# side of comms.py / dlrm.py / commsComputeBench.py / commsTraceReplay.py / run_benchmark.py ran on gloo ranks only), so this
# visit (1) runs tools/r4_visit.sh's record (GPU tests, smoke, bench twice, dist-debug, rocprofv3 kernel stats) and (2) drives
# the new command-line surface once on the device (1-rank RCCL group): every collective of the backend table with --c 1, the
# perf logger, graph launches next to the new collectives, the overlap bench's reference-format rows, the DLRM driver's report,
# the replay's profiler ranges.  Each step has its own log and time limit; a failing step does not stop the others.
tag=${1:-r5_first}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
bash tools/r4_visit.sh "$tag" 2>&1 | tail -40
step() { name=$1; shift; timeout 300 "$@" > "$out/cli_$name.log" 2>&1; echo "== $name rc=$? $(grep -c 'COMMS-RES-' "$out/cli_$name.log") rows"; tail -3 "$out/cli_$name.log"; }
common="--master-ip 127.0.0.1 --device rocm --backend rccl_xgmi --n 10 --w 2"
step table       python -m param_amd.comms.pt.comms $common --master-port 29701 --z 1 --c 1 --b 1K --e 1M --f 32 \
                 --collective all_gather,all_gather_base,reduce_scatter,reduce_scatter_base,broadcast,reduce,gather,scatter,all_to_all,all_to_allv,all_to_all_single,all_reduce
step nonblocking python -m param_amd.comms.pt.comms $common --master-port 29702 --z 0 --b 1K --e 1M --f 32 --num-coll 4 --tag nb \
                 --collective all_to_allv,all_gather_base,reduce_scatter_base --use-perf-logger jsonl
step sizes_dtype python -m param_amd.comms.pt.comms $common --master-port 29703 --z 1 --c 1 --ss 4096,65536,256 --data-types float32,bfloat16,int32 \
                 --collective all_to_all_single,all_reduce
step graph       python -m param_amd.comms.pt.comms $common --master-port 29704 --z 1 --c 1 --b 1K --e 64K --f 8 --graph-launches 5 \
                 --collective all_gather_base,reduce_scatter_base,broadcast,all_to_allv
step profiler    env PARAM_COMMS_PROFILE_DIR=$out/comms_profile python -m param_amd.comms.pt.comms $common --master-port 29705 --z 1 \
                 --b 64K --e 64K --collective all_to_allv --size-start-profiler 64K --use-device-time
step overlap     python -m param_amd.comms.pt.commsComputeBench $common --master-port 29706 --b 1M --e 16M --f 4 --collective all_to_allv \
                 --kernel emb_lookup --num-compute 3 --ntables 8 --num-embs 200000 --emb-dim 128 --batch-size 2048 --bag-size 20 --tag ov
step dlrm        python -m param_amd.comms.pt.dlrm --master-ip 127.0.0.1 --master-port 29707 --device rocm --mini-batch-size 2048 --num-batches 8 \
                 --warmup-batches 2 --arch-mlp-bot 64-32 --arch-mlp-top 32-1 --arch-sparse-feature-size 128 \
                 --arch-embedding-size 200000-300000-400000-500000 --num-indices-per-lookup 20 --num-indices-per-lookup-fixed --perf-debug \
                 --arch-interaction-op cat
step replay      python -m param_amd.comms.pt.commsTraceReplay --trace-path examples/trace_replay --device rocm --master-ip 127.0.0.1 \
                 --master-port 29708 --num-replays 4 --do-warm-up --z 1 --use-one-trace --disable-parallel-read --enable-profiler \
                 --profiler-num-replays 2 --output-path $out/replay
PARAM_AMD_R5_CLI=1 timeout 900 python -m pytest tests/test_gpu_cli_surface.py -q -m gpu > "$out/pytest_cli_surface.log" 2>&1
echo "== gated GPU tests of the command-line surface (green -> drop the gate in tests/test_gpu_cli_surface.py)"; tail -5 "$out/pytest_cli_surface.log"
ls "$out" | head -60
