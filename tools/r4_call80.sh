#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
echo "== quad determinism"
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from unet_zoo_amd import _ffi
L = _ffi.lib(); st = torch.cuda.current_stream().cuda_stream
for (N, C, H, W, ac, acc) in [(32, 192, 64, 64, 1, 1), (32, 192, 32, 32, 1, 0), (8, 64, 64, 32, 0, 1), (32, 224, 64, 64, 1, 1)]:
    dy = torch.randn(N, C, 2 * H, 2 * W, device="cuda"); base = torch.randn(N, C, H, W, device="cuda")
    outs = []
    for _ in range(20):
        dx = base.clone()
        _ffi.check(L.uz_bilinear2x_bwd(dy.data_ptr(), C, C, dx.data_ptr(), C, N, H, W, ac, acc, st), "b")
        outs.append(dx)
    torch.cuda.synchronize()
    print((N, C, H, W, ac, acc), "all equal:", all(torch.equal(outs[0], o) for o in outs[1:]))
PY
echo "== dp test, quad"; python -m pytest tests/test_dp_gpu.py -q -p no:cacheprovider -k world_size_one 2>&1 | grep "bit-identical\|passed\|failed"
echo "== dp test, pair"; UZ_BILINEAR_BWD_PAIR=1 python -m pytest tests/test_dp_gpu.py -q -p no:cacheprovider -k world_size_one 2>&1 | grep "bit-identical\|passed\|failed"
echo "== dp test, pair again"; UZ_BILINEAR_BWD_PAIR=1 python -m pytest tests/test_dp_gpu.py -q -p no:cacheprovider -k world_size_one 2>&1 | grep "bit-identical\|passed\|failed"
echo "== split probunet graph, quad"; UZ_CONV_MATH=split python -m pytest tests/test_unet_probunet_gpu.py -q -p no:cacheprovider -k "graph_replay" 2>&1 | grep "passed\|failed\|diff"
echo "== split probunet graph, pair"; UZ_BILINEAR_BWD_PAIR=1 UZ_CONV_MATH=split python -m pytest tests/test_unet_probunet_gpu.py -q -p no:cacheprovider -k "graph_replay" 2>&1 | grep "passed\|failed\|diff"
} > gpurun_out/r4_call80.txt 2>&1
