import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
from oracle.unet_ref import tiny_config
cfg = tiny_config()
sd = random_state_dict({**DEFAULT_CONFIG, **cfg}, 0)
m1 = UNetSpatioTemporalConditionModel(**cfg).load_state_dict(sd, device="cuda")
m2 = UNetSpatioTemporalConditionModel(**cfg).load_state_dict({k: v.cuda() for k, v in sd.items()}, device="cuda")
bad = [k for k in m1.w if isinstance(m1.w[k], torch.Tensor) and not torch.equal(m1.w[k], m2.w[k])]
for k in m1.w:
    if isinstance(m1.w[k], dict):
        for kk in m1.w[k]:
            a, b = m1.w[k][kk], m2.w[k][kk]
            if isinstance(a, torch.Tensor):
                if not torch.equal(a, b): bad.append((k, kk))
                if a.data_ptr() % 16 or b.data_ptr() % 16 or not b.is_contiguous(): print("ALIGN/CONTIG", k, kk, a.data_ptr() % 16, b.data_ptr() % 16, b.is_contiguous(), b.shape, b.stride())
            elif a != b: bad.append((k, kk))
print("mismatching packed entries:", bad[:10], len(bad))
