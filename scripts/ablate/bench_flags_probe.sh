# repeated default runs: value next to the host interpreter's garbage collections inside the timed region
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/step %.1f scenes/s  gc %s' % (j['ms_per_step'], j['value'], j['config']['host_gc']))"
done
