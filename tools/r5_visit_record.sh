#!/bin/bash
# tools/r5_visit_record.sh [tag]: round 5's record on one GPU box -- GPU tests, smoke, the default bench line twice, the N > 1 path on a
# 1-rank RCCL group (26 tables, Criteo), rocprofv3 kernel stats of the headline launches (Zipf, uniform) and of every phase of the
# fp32 / bf16 / Criteo blocks.  Everything lands in gpurun_out/<tag>/; what is to be judged is copied to profiles/r05_*.
tag=${1:-r5_record}
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 900 python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
timeout 900 python bench.py > "$out/bench_line_run2.json" 2> "$out/bench_line_run2.err"
timeout 900 python bench.py --dist-debug --tables 26 --no-cpu-baseline --steps 20 > "$out/distdebug_26tables.json" 2> "$out/dd26.err"
timeout 900 python bench.py --dist-debug --workload criteo --no-cpu-baseline --steps 20 > "$out/distdebug_criteo.json" 2> "$out/ddc.err"
for v in zipf:--only-headline uniform:--only-headline,--alpha,0; do
  name=${v%%:*}; bargs=${v#*:}
  d=/tmp/${tag}_$name; rm -rf "$d"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 5 ${bargs//,/ } > "$GRAFT_REPO_ROOT/$out/headline_${name}_under_rocprofv3.json" 2> "$d.err")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/headline_${name}_kernel_stats.csv"
  python tools/r4_kstats.py stats "$out/headline_${name}_kernel_stats.csv" | head -3
done
bash tools/r5_kstats.sh $tag fp32,bf16,criteo
for n in bench_line bench_line_run2; do echo "== $n"; python tools/r5_bench_summary.py "$out/$n.json"; done
