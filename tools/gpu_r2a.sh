#!/bin/bash
# round-2 visit A: GPU tests (incl. the new full-size Zipf test), bench.py smoke on every code path, the default bench line,
# the [T,B,D] layout line, the contiguous-allocation probe.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
(timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -25) > $O/r2a_pytest.log
timeout 300 python bench.py --tables 8 --rows 200000 --steps 5 --warmup 2 --no-cpu-baseline > $O/r2a_small.json 2> $O/r2a_small.err
timeout 300 python bench.py --dist-debug --tables 8 --rows 200000 --steps 5 --warmup 2 > $O/r2a_small_dd.json 2> $O/r2a_small_dd.err
timeout 300 python bench.py --dist-debug --workload criteo --steps 5 --warmup 2 --lookup-cus 224 > $O/r2a_criteo_dd.json 2> $O/r2a_criteo_dd.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r2a_bench.json 2> $O/r2a_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --layout tbd --no-cpu-baseline > $O/r2a_bench_tbd.json 2> $O/r2a_bench_tbd.err
timeout 600 python tools/contig_probe.py > $O/r2a_contig.jsonl 2> $O/r2a_contig.err
tail -3 $O/r2a_pytest.log; for f in small small_dd criteo_dd bench bench_tbd; do echo "== $f"; head -c 600 $O/r2a_$f.json; tail -3 $O/r2a_$f.err; done; cat $O/r2a_contig.jsonl; tail -3 $O/r2a_contig.err
