import sys, torch, numpy as np
sys.path.insert(0, ".")
import torch.nn.functional as F
from tests import helpers as H
from oracle import endodav_oracle as orc
cuda = torch.device("cuda:0")
name = "micro_clstoken"
model, kwargs, shape, kind, store = H.build_model(name)
x = H.case_input(name)
sd = {k: v.detach() for k, v in model.state_dict().items()}
cfg = H.oracle_config(kwargs)
xr = x.flatten(0, 1)
mean = torch.tensor(orc.IMAGENET_MEAN)[None, :, None, None]; std = torch.tensor(orc.IMAGENET_STD)[None, :, None, None]
feats = orc.encoder_taps(sd, (xr - mean) / std, cfg)
model = model.to(cuda); model.set_capture(True)
with torch.no_grad(): out = model(x.to(cuda))
print("cfg use_clstoken", model._config().use_clstoken)
for j in range(4):
    got = model.stage(f"tapcls{j}").cpu().reshape(feats[j][1].shape)
    print("tapcls", j, H.rel_err(got.numpy(), feats[j][1].numpy()), "tap", H.rel_err(model.stage(f"tap{j}").cpu().reshape(feats[j][0].shape).numpy(), feats[j][0].numpy()))
tok, cls = feats[3]
W, b = sd["head.readout_projects.3.0.weight"], sd["head.readout_projects.3.0.bias"]
D = W.shape[0]
fb = F.linear(cls, W[:, D:], b)
print("fbias", H.rel_err(model.stage("fbias").cpu().reshape(fb.shape).numpy(), fb.numpy()))
ro = F.gelu(F.linear(torch.cat((tok, cls.unsqueeze(1).expand_as(tok)), -1), W, b))
print("readout", H.rel_err(model.stage("readout").cpu().reshape(ro.shape).numpy(), ro.numpy()))
