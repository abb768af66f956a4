#!/bin/bash
# tools/r5_visit1.sh [tag]: round 5, first GPU visit -- (1) the persistent forward's parity tests, (2) its same-box A/B against the
# classic kernel, (3) the command-line surface round 4 could only run on gloo ranks (tools/r5_first_visit.sh's steps) + its gated tests.
tag=${1:-r5_v1}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fwd_persist.py -x -q -m gpu > "$out/pytest_persist.log" 2>&1; echo "== persist tests rc=$?"; tail -5 "$out/pytest_persist.log"
timeout 900 python tools/r5_fwd_ab.py --rounds 2 > "$out/fwd_ab_fp32.jsonl" 2> "$out/fwd_ab_fp32.err"; echo "== fwd A/B rc=$?"; tail -3 "$out/fwd_ab_fp32.err"
python - "$out/fwd_ab_fp32.jsonl" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
bad = [r for r in rows if r.get("bit_identical_to_classic") is False]
print("bit-identity failures:", bad)
agg = collections.defaultdict(list)
for r in rows:
    if "avg_launch_us" in r: agg[r["config"], r["layout"], r["indices"]].append(r["avg_launch_us"])
for k in sorted(agg): print(k, agg[k])
PY
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
echo "== gated GPU tests of the command-line surface"; tail -5 "$out/pytest_cli_surface.log"
ls "$out" | head -60
