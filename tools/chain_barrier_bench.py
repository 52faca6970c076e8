"""GPU: what a phase of the chain launch costs when it holds (almost) no work - a table of N phases with one trivial sub-op each
(a 1-slab slab-sum over 64 floats), for several workgroup counts; stamps of workgroup 0 give the per-phase time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib()
codes = _ffi.chain_codes()
NPH = 100
src = torch.zeros(4096, device="cuda"); dst = torch.zeros(4096, device="cuda")
arr = (_ffi.uz_chain_op * NPH)()
for k in range(NPH):
    a = arr[k]
    a.code = codes["UZ_CH_SLAB_SUM"]
    for j, v in enumerate([1, 1, 1, 1, 64, 0]):
        a.i[j] = v
    a.p[0], a.p[1] = src.data_ptr(), dst.data_ptr()
    a.tile0, a.ntiles = 0, 1
ops = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().cuda()
phases = torch.tensor([[k, 1] for k in range(NPH)], dtype=torch.int32).reshape(-1).cuda()
state = torch.zeros(L.uz_chain_state_bytes() // 4, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for G in [int(g) for g in os.environ.get("GS", "32,64,128,256").split(",")]:
    for _ in range(3):
        _ffi.check(L.uz_chain_run(ops.data_ptr(), phases.data_ptr(), NPH, NPH, G, state.data_ptr(), st), "run")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.uz_chain_run(ops.data_ptr(), phases.data_ptr(), NPH, NPH, G, state.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    stamps = state.cpu()[32:].view(torch.int64)
    span = (stamps[NPH - 1].item() - stamps[0].item()) / 100.0
    out = C.c_int(0); L.uz_chain_status(state.data_ptr(), C.byref(out), st)
    print(f"G={G:4d}  launch {e0.elapsed_time(e1) * 100:.1f} us for {NPH} phases  -> {e0.elapsed_time(e1) * 100 / NPH:.2f} us / phase (stamped {span / (NPH - 1):.2f})  status {out.value}")
