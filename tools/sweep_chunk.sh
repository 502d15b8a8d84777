#!/bin/bash
for c in "$@"; do
  C=$c STENOS_CHUNK_MIB=$c python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-entropy 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('chunk_MiB', os.environ.get('C'), 'value', d['value'], 'enc', d['encode_gbps'], 'dec', d['decode_gbps'], 'kernel_ms', d['roofline']['kernel_ms'])" 
done
