#!/bin/bash
# tools/r4_visit.sh <tag>: round 4's record on one GPU box -- GPU tests, smoke, the default bench line twice in a row (it now carries
# the bf16 64-table and Criteo blocks), the N > 1 path on a 1-rank RCCL group (26 tables, Criteo), rocprofv3 kernel stats of the
# headline launches and of the whole default command.  Everything lands in gpurun_out/<tag>/; copy what is to be judged to profiles/.
tag=${1:-r4_official}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 900 python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
timeout 900 python bench.py > "$out/bench_line_run2.json" 2> "$out/bench_line_run2.err"
timeout 900 python bench.py --dist-debug --tables 26 --no-cpu-baseline --steps 20 > "$out/distdebug_26tables.json" 2> "$out/dd26.err"
timeout 900 python bench.py --dist-debug --workload criteo --no-cpu-baseline --steps 20 > "$out/distdebug_criteo.json" 2> "$out/ddc.err"
for v in zipf:--only-headline uniform:--only-headline,--alpha,0 full:--no-cpu-baseline,--no-extra; do
  name=${v%%:*}; bargs=${v#*:}
  d=/tmp/r4off_$name; rm -rf "$d"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 ${bargs//,/ } > "$GRAFT_REPO_ROOT/$out/${name}_under_rocprofv3.json" 2> "$d.err")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${name}_kernel_stats.csv"
  python tools/r4_kstats.py stats "$out/${name}_kernel_stats.csv" > "$out/${name}_kernel_stats_pm.txt"
done
for n in bench_line bench_line_run2 distdebug_26tables distdebug_criteo uniform_under_rocprofv3; do echo "== $n"; python tools/r4_bench_summary.py "$out/$n.json"; done
echo "== headline kernels under rocprofv3"; head -3 "$out/zipf_kernel_stats_pm.txt"; head -3 "$out/uniform_kernel_stats_pm.txt"; echo "== full"; cat "$out/full_kernel_stats_pm.txt"
