C="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for r in 1 2 3; do for w in 48 56 64 72 80; do
  PGMOVE_STATS_CU_WITHHELD=$w timeout -k 10 200 python3 bench.py $C 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['ms_per_step_blocks']; print('rep $r withheld $w: %.4f (median %.4f)' % (d['ms_per_step'], b['median']))"
done; done
