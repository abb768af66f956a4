#!/bin/bash
# tools/multi_gpu_sweep.sh N [tag]: the N > 1 measurements on one node with N MI355X (never run on hardware: one GPU per gpurun
# visit; the round-end driver runs bench.py --gpus N only).  One process per GPU under torch.distributed.run; the drivers open
# their own TCP store, so --master-port given to a driver differs from the launcher's rendezvous port.
#   DEVICE=cpu bash tools/multi_gpu_sweep.sh 2      dry run of the comms.py steps on gloo ranks (build container)
# Steps: (1) bench.py --gpus N (the contract's line: lookups/s, exchange busBW vs the xGMI bound, overlap, self-check);
# (2) the all-to-all sweep 8 B .. 256 MB, eager and as hipGraph replays; (3) one xGMI link: --pt2pt one2one 0 -> 1, and all links of
# GPU pairs at once: --pt2pt pairwise; (4) configs[3]: dlrm.py 26 tables, batch 8192 per rank; (5) lookup || all-to-all overlap bench.
N=${1:?ranks}
tag=${2:-multi_n$N}
DEVICE=${DEVICE:-rocm}
BACKEND=$([ "$DEVICE" = cpu ] && echo gloo || echo rccl_xgmi)
out=gpurun_out/$tag
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29600
run() {  # run <name> <module or script> <args...>: N ranks under torch.distributed.run
  name=$1; shift; target=$1; shift
  port=$((port + 2))
  if [[ "$target" == *.py ]]; then launch=("$target"); else launch=(-m "$target" --); fi
  timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$port" \
      "${launch[@]}" "$@" > "$out/$name.log" 2> "$out/$name.err"
  echo "== $name rc=$? ($(grep -c 'COMMS-RES-' "$out/$name.log") COMMS-RES rows)"; tail -4 "$out/$name.log"
}
drv="--master-ip 127.0.0.1 --device $DEVICE --backend $BACKEND"
if [ "$DEVICE" != cpu ]; then
  run bench bench.py --gpus "$N" --steps 50 --warmup 5
  run bench_26 bench.py --gpus "$N" --tables 26 --steps 50 --warmup 5
  run bench_criteo bench.py --gpus "$N" --workload criteo --steps 50 --warmup 5
  run bench_criteo_mixed bench.py --gpus "$N" --workload criteo --mixed-dims --steps 50 --warmup 5
  run bench_masked bench.py --gpus "$N" --lookup-cus 224 --steps 50 --warmup 5     # (the default picks the compute stream by a trial in the run:
  run bench_unmasked bench.py --gpus "$N" --lookup-cus 0 --steps 50 --warmup 5     #  both fixed settings beside it, for the record)
  big=256M; graphs="--graph-launches 10"
else
  big=1M; graphs=""
fi
run a2a_sweep param_amd.comms.pt.comms $drv --master-port $((port + 1001)) --b 8 --e $big --f 2 --n 100 --w 10 --z 1 --collective all_to_all,all_to_allv
[ -n "$graphs" ] && run a2a_graph param_amd.comms.pt.comms $drv --master-port $((port + 1001)) --b 8 --e 1M --f 4 --n 20 --w 5 --z 1 --collective all_to_allv $graphs
run link_one2one param_amd.comms.pt.comms $drv --master-port $((port + 1001)) --pt2pt one2one --src-ranks 0 --dst-ranks 1 --b 1K --e $big --f 8 --n 20 --w 5 --window 16
half=$((N / 2))
[ "$half" -ge 1 ] && run links_pairwise param_amd.comms.pt.comms $drv --master-port $((port + 1001)) --pt2pt pairwise --src-ranks 0:$((half - 1)) --dst-ranks $half:$((2 * half - 1)) --b 1M --e $big --f 16 --n 20 --w 5 --window 16
if [ "$DEVICE" != cpu ]; then
  tables=$(python -c "print('-'.join(['10000000'] * 26))")
  run dlrm_cfg3 param_amd.comms.pt.dlrm --master-ip 127.0.0.1 --master-port $((port + 1001)) --device rocm --arch-embedding-size "$tables" \
      --arch-sparse-feature-size 128 --mini-batch-size 8192 --num-indices-per-lookup 20 --num-indices-per-lookup-fixed \
      --num-batches 30 --warmup-batches 5 --arch-mlp-bot 512-256-128 --arch-mlp-top 1024-1024-512-256-1 --print-comms
  run overlap param_amd.comms.pt.commsComputeBench $drv --master-port $((port + 1001)) --b 16M --e 256M --f 4 --collective all_to_allv \
      --kernel emb_lookup --num-compute 4 --ntables 8 --num-embs 10000000 --emb-dim 128 --batch-size 8192 --bag-size 20
fi
ls "$out"
