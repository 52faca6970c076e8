#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{ echo "== 3d tests"; python -m pytest tests/test_phiseg3d.py tests/test_b16_storage_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
  for t in 1 0 1 0; do echo "== UZ_WGRAD_TABLE_VOL=$t"; UZ_WGRAD_TABLE_VOL=$t python bench.py --model phiseg3d --skip-cpu --no-profile 2>/dev/null | tail -1 | cut -c1-200; done
  echo "== f32 storage"; for t in 1 0; do UZ_WGRAD_TABLE_VOL=$t python bench.py --model phiseg3d --storage f32 --skip-cpu --no-profile 2>/dev/null | tail -1 | cut -c1-200; done
} > gpurun_out/r4_call86.txt 2>&1
