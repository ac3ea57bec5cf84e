set -e
for k in ${KS:-1 4 8 16}; do
  RAFTGPU_RELAY_CHUNKS=$k timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('chunks $k', 'value %.4e' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'flags', d['safety_flags'], d['flagged_groups'], 'committed', d['counters_per_step']['entries_committed'])"
done
