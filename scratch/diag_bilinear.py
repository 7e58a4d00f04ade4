import torch, numpy as np, torch.nn.functional as F
print("cpu capability:", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
torch.manual_seed(0)
for (H, OH) in ((296, 518), (129, 64), (518, 259)):
    x = torch.rand(1,1,H,1)
    ref = F.interpolate(x, size=(OH,1), mode="bilinear", align_corners=True)[0,0,:,0].numpy()
    xs = x[0,0,:,0].numpy().astype(np.float64)
    r32 = np.float32(H-1)/np.float32(OH-1)
    def run(lam_fn):
        out = np.zeros(OH, np.float32)
        for o in range(OH):
            src = np.float32(r32*np.float32(o)); i0 = int(src); i1 = i0 + (1 if i0 < H-1 else 0)
            l1 = lam_fn(o, src, i0)
            out[o] = np.float32((1-l1)*xs[i0] + l1*xs[i1])
        return np.abs(out-ref).max()
    print(H, OH, "plain", run(lambda o,src,i0: float(np.float32(src-np.float32(i0)))),
          "fma", run(lambda o,src,i0: float(np.float32(float(r32)*o - i0))))
    # 2-D, multi-channel
    x2 = torch.rand(1,4,H,H)
    a = F.interpolate(x2, size=(OH,OH), mode="bilinear", align_corners=True)
    if torch.cuda.is_available():
        b = F.interpolate(x2.cuda(), size=(OH,OH), mode="bilinear", align_corners=True).cpu()
        print("   torch cpu vs torch gpu 2-D:", float((a-b).abs().max()))
