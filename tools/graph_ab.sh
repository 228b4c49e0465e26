for conc in 1 0; do for g in "" "--no-graph"; do
echo "conc=$conc graph=$g"; SVA_CONCURRENCY=$conc python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline $g | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
