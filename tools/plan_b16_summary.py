"""CPU: which buffers of the config-5 plan go to bf16 storage (UZ_STORE_B16=1), what stays fp32 and why."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UZ_STORE_B16", "1")
from unet_zoo_amd.models.phiseg3D import PHISeg3D
from unet_zoo_amd import _ffi
L = _ffi.lib(); L.uz_set_conv_math(3)
net = PHISeg3D(4, 3, [32, 64, 128, 192, 192], latent_levels=5, image_size=(4, 128, 128, 64), device="cpu"); net.train()
plan = net._build(128, 128, 64, True, True)
print(plan.b16_info)
print("arena GB", plan.arena_floats * 4 / 1e9)
big = [b for b in plan.bufs if b.vol and b.W >= 32]
print("volume buffers with planes >= 64 x 32:", len(big), "in bf16:", sum(b.b16 for b in big), "| GB kept fp32", sum(b.numel * 4 for b in big if not b.b16) / 1e9,
      "| GB in bf16", sum(b.numel * 2 for b in big if b.b16) / 1e9)
for b in sorted([b for b in big if not b.b16], key=lambda b: -b.numel)[:int(sys.argv[1]) if len(sys.argv) > 1 else 10]:
    print("  fp32:", b.name, (b.N, b.C, b.H, b.W), round(b.numel * 4 / 1e6, 1), "MB")
