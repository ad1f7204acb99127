#!/bin/bash
# same-box A/B of the LDS-DMA cache-policy bits: the shipped library (sc1) against one built with ENH_GLDS_AUX_OVERRIDE=0, interleaved
export TMPDIR=/tmp
for rep in 1 2; do
for lib in sc1 none; do
  [ $lib = none ] && export ENH_HIP_LIB=$PWD/enhancing-transformers_amd/lib/libenh_hip_aux0.so || unset ENH_HIP_LIB
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib headline', d['value'], 'img/s;', ' '.join(f\"{v['achieved']:.0f}\" for n,v in sorted(k.items(), key=lambda kv:-kv[1]['share_of_step'])[:9]))"
done; done
for lib in sc1 none; do
  [ $lib = none ] && export ENH_HIP_LIB=$PWD/enhancing-transformers_amd/lib/libenh_hip_aux0.so || unset ENH_HIP_LIB
  timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib adv16', d['value'], 'img/s')"
done
