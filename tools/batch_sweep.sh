for b in 512 64 128 192 256 384 768 512; do
  timeout -k 10 150 python bench.py --steps 12 --warmup 4 --batch $b --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['mfma_peak_fraction'])" || exit 1
done
