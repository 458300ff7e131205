#!/bin/bash
# chunk x waves_per_cu sweep, serial kernel timing (C2)
for c in 75 25; do for w in 8 12 16 24; do
  python bench.py --no-cpu-baseline --serial --steps 40 --chunk $c --waves-per-cu $w | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c wpc $w segs', d['config']['segments'], 'kernel_ms', d['roofline']['kernel_ms_mean'], 'min', d['roofline']['kernel_ms_min'], 'step', d['ms_per_step'])"
done; done
