"""Every kernel of the forward (and of the fine-tune step) beside another instance's kernels: two models on two streams, each round's outputs must be
bit-identical to that model's solo run.  (The attention LDS race of round 2 only showed under co-residency; this sweeps all kernels for siblings.)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import endodav_amd
from endodav_amd import synth
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 518
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 60
def make(seed):
    m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(size, size), lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(m)
    return m.to(dev), torch.from_numpy(synth.synth_clip(1, T, size, size, seed=seed, kind="tissue")).to(dev), torch.cuda.Stream()
A, B = make(1), make(2)
def fwd(x):
    m, clip, s = x
    with torch.cuda.stream(s), torch.no_grad():
        return [o.clone() for o in m(clip).values()]
torch.cuda.synchronize()
ra = fwd(A); torch.cuda.synchronize(); rb = fwd(B); torch.cuda.synchronize()
bad = 0
for it in range(rounds):
    oa = fwd(A); ob = fwd(B); oa2 = fwd(A); ob2 = fwd(B)
    torch.cuda.synchronize()
    for got, ref in ((oa, ra), (ob, rb), (oa2, ra), (ob2, rb)):
        bad += sum(not torch.equal(a, b) for a, b in zip(got, ref))
print(f"inference, two models on two streams, T={T} {size}x{size}: {rounds} rounds x 4 forwards, {bad} of {rounds * 16} outputs differ from the solo run")
# fine-tune step gradients under the same co-residency
for m in (A[0], B[0]):
    endodav_amd.mark_only_part_as_trainable(m, warm_up=True); m.train()
def grads(x):
    m, clip, s = x
    with torch.cuda.stream(s):
        m.zero_grad(set_to_none=True)
        sum(o.mean() for o in m(clip).values()).backward()
        return [p.grad.clone() for p in m.parameters() if p.requires_grad]
ga = grads(A); torch.cuda.synchronize(); gb = grads(B); torch.cuda.synchronize()
badg = 0
for it in range(max(rounds // 4, 5)):
    xa = grads(A); xb = grads(B)
    torch.cuda.synchronize()
    badg += sum(not torch.equal(a, b) for a, b in zip(xa, ga)) + sum(not torch.equal(a, b) for a, b in zip(xb, gb))
print(f"fine-tune step, two models on two streams: {max(rounds // 4, 5)} rounds, {badg} gradient tensors differ from the solo run (of {len(ga) + len(gb)} per round)")
