# k_read_stats compiled for 7 waves per SIMD against the library (bash tools/variants.sh sw7 "-DPG_STATS_WAVES_PER_EU=7" first), C1 both stream modes, three passes
C="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for r in 1 2 3; do for v in base sw7; do
  L=""; [ $v != base ] && L="--lib build/$v/libpgmove.so"
  for m in "" "--one-stream"; do
    timeout -k 10 200 python3 bench.py $C $m $L 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['ms_per_step_blocks']; print('rep $r $v ${m:-default}: %.4f (median %.4f) stats %.1f us' % (d['ms_per_step'], b['median'], d['kernels_ms_per_step']['k_read_stats']*1e3))"
  done
done; done
