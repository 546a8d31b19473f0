set -e
mkdir -p gpurun_out
for rev in 0 1 0 1; do
  echo "== REVERSE=$rev iso 4096"
  SFHIP_REVERSE=$rev timeout -k 10 120 python bench.py --no-traffic-pass --no-cpu-baseline --steps 200 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms'))"
done
for rev in 0 1; do
  echo "== REVERSE=$rev gam 4096"
  SFHIP_REVERSE=$rev timeout -k 10 120 python bench.py --solver gam --no-traffic-pass --no-cpu-baseline --steps 100 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms'))"
done
for rev in 0 1; do
  echo "== REVERSE=$rev iso 2048"
  SFHIP_REVERSE=$rev timeout -k 10 120 python bench.py --resolution 2048 --no-traffic-pass --no-cpu-baseline --steps 200 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms'))"
  echo "== REVERSE=$rev iso 8192"
  SFHIP_REVERSE=$rev timeout -k 10 120 python bench.py --resolution 8192 --no-traffic-pass --no-cpu-baseline --steps 50 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms'))"
done
