#!/usr/bin/env python3
"""Debug aid: the two-launch heads against the multi-launch heads at small shapes, repeated; prints which outputs deviate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import model as MM
from msa_amd.model import MMBertConfig, MMBertForPretraining
dev = torch.device("cuda", 0)
for B, H in ((2, 128), (4, 128), (16, 768)):
    torch.manual_seed(0)
    m = MMBertForPretraining(MMBertConfig(hidden_size=H, num_hidden_layers=1, num_attention_heads=H // 64, intermediate_size=4 * H, vocab_size=512))
    m.bert.set_joint_embeddings("mosei"); m.to(dev)
    m._ensure_ready(dev)
    with torch.no_grad():
        for n, q in m.named_parameters():
            if not n.startswith(("bert.embeddings", "bert.encoder", "cls.predictions", "bert.jointEmbeddings")):
                q.mul_(6.0)
    m._flat.maybe_refresh()
    bad = {}
    for it in range(40):
        first = torch.randn(3 * B, H, device=dev)
        ap = torch.randint(0, 2, (2 * B,), device=dev); sent = torch.rand(B, device=dev) * 6 - 3
        mlm = torch.tensor([7.0, 7.1, 6.9], device=dev)
        res = []
        for fn in (MM._HeadsFn, MM._HeadsStepFn):
            m._flat.grads.zero_()
            f = first.clone().requires_grad_(True)
            loss, aux, logits, t_rel, rel = fn.apply(f, m, ap, sent, mlm)
            loss.backward()
            torch.cuda.synchronize()
            res.append(dict(loss=loss.detach().view(1), aux=aux.clone(), logits=logits.clone().view(-1), t_rel=t_rel.clone().view(-1), rel=rel.clone().view(-1),
                            dfirst=f.grad.clone().view(-1), grads=m._flat.grads.clone()))
        for k in res[0]:
            a, b = res[0][k], res[1][k]
            e = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)
            if e > 1e-4:
                bad.setdefault(k, []).append((it, round(e, 5)))
    print(f"B={B} H={H}: deviations > 1e-4 (relative to the largest entry) over 40 runs: {({k: v[:4] for k, v in bad.items()}) or 'none'}", flush=True)
