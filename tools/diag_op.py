"""Time selected backward ops of the PHiSeg plan in isolation and print their arguments (diagnostic)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from unet_zoo_amd import _ffi
from unet_zoo_amd.synthetic import synthetic_batch
net = bench.build("phiseg"); net.train()
x, m, _ = synthetic_batch(32)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
for _ in range(2):
    net.forward(x, m); net.loss(m).backward()
torch.cuda.synchronize()
plan = net._cur; L = _ffi.lib(); st = C.c_void_p(net._stream())
print("flags after two training steps", net.check_bounds())
arr, n = plan.tapes["bwd"]
for k in [int(v) for v in sys.argv[1:]]:
    o = plan.bwd_ops[k]
    j = k
    while j > 0 and plan.bwd_ops[j - 1]["gid"] == o["gid"]:
        j -= 1
    if j < k and os.environ.get("PREFIX", "1") == "1":
        _ffi.check(L.uz_run_tape((type(arr[0]) * (k - j))(*[arr[i] for i in range(j, k)]), k - j, st), "prefix")
    one = (type(arr[0]) * 1)(arr[k]); best = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _ffi.check(L.uz_run_tape(one, 1, st), "op"); e1.record(); e1.synchronize(); best.append(round(e0.elapsed_time(e1) * 1e3, 1))
    print(k, o["code"], o["i"], "us:", best)
    for j, r in enumerate(o["p"]):
        if r is None: continue
        d = getattr(r, "buf", None)
        print("    p[%d]" % j, (d.name, d.C, r.c0, r.C) if d is not None else (type(r).__name__, getattr(r, "view", None) and r.view.buf.name) if not isinstance(r, tuple) else r)
    for j in (4, 5):
        r = o["p"][j] if len(o["p"]) > j else None
        if isinstance(r, tuple) and r[0] == "amax":
            addr = plan._resolve(r, 0)
            t = torch.empty(256, device="cuda")
            _ffi.check(L.uz_copy_f32(t.data_ptr(), addr, 256, st), "copy")
            torch.cuda.synchronize(); print("    slot", r, "max", float(t.max()))
print("flags", net.check_bounds())
