"""Small-grid 3x3 convolutions of the DPT head: plain grid vs the K split (edv_conv3x3 vs edv_conv3x3_ws), warm, interleaved."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
GWS = torch.zeros(lib.edv_gemm_workspace() // 4, device=dev)
def timed(fn, iters=200):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3
big = [torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev) * 0.05, torch.empty(8192, 8192, device=dev)]
timed(lambda: _lib.check(lib.edv_gemm(big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr(), 8192, 8192, 1024, None, 0, None, None, None, 0, st())), 300)
for (F, H, W, Cin, Cout, s, what) in ((8, 19, 19, 384, 64, 1, "layer4_rn"), (8, 37, 37, 192, 64, 1, "layer3_rn"), (8, 37, 37, 384, 384, 2, "resize_layers.3"),
                                      (8, 37, 37, 64, 64, 1, "RCU 37x37"), (8, 19, 19, 64, 64, 1, "RCU 19x19"), (8, 74, 74, 64, 64, 1, "RCU 74x74"),
                                      (8, 74, 74, 96, 64, 1, "layer2_rn"), (8, 148, 148, 64, 64, 1, "RCU 148x148")):
    x = torch.randn(F, H, W, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    y = torch.empty(F, OH, OW, Cout, device=dev)
    ts = {False: [], True: []}
    for rep in range(3):
        ts[False].append(timed(lambda: _lib.check(lib.edv_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), F, H, W, Cin, Cout, s, 1, 0, None, None, st()))))
        ts[True].append(timed(lambda: _lib.check(lib.edv_conv3x3_ws(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), F, H, W, Cin, Cout, s, 1, 0, None, None, GWS.data_ptr(), GWS.numel() * 4, st()))))
    tiles = (F * OH * OW + 63) // 64 * ((Cout + 63) // 64)
    print(f"{what:16s} {F}x{H}x{W} {Cin}->{Cout} s{s}: tiles {tiles:5d}  plain {min(ts[False]):6.1f} us   split {min(ts[True]):6.1f} us", flush=True)
