#!/usr/bin/env python3
"""Stand-alone GPU time of the pretraining heads, forward and backward: the level-launch form (csrc/heads_coop.hip, model._HeadsStepFn: 7 + 6 launches) against the
multi-launch form (csrc/heads.hip, model._HeadsFn) at the headline shape (B = 16, H = 768) and at the reference's default (B = 32, H = 1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import model as MM
from msa_amd.model import MMBertConfig, MMBertForPretraining
dev = torch.device("cuda", 0)
for B, H in ((16, 768), (32, 1024)):
    torch.manual_seed(0)
    m = MMBertForPretraining(MMBertConfig(hidden_size=H, num_hidden_layers=1, num_attention_heads=H // 64, intermediate_size=4 * H, vocab_size=512))
    m.bert.set_joint_embeddings("mosei"); m.to(dev)
    m._ensure_ready(dev)
    first = torch.randn(3 * B, H, device=dev)
    ap = torch.randint(0, 2, (2 * B,), device=dev); sent = torch.rand(B, device=dev) * 6 - 3
    mlm = torch.tensor([7.0, 7.1, 6.9], device=dev)
    for name, fn in (("level launches", MM._HeadsStepFn), ("multi-launch", MM._HeadsFn)):
        for _ in range(3):
            f = first.clone().requires_grad_(True)
            fn.apply(f, m, ap, sent, mlm)[0].backward()
        torch.cuda.synchronize()
        tf, tb = [], []
        for _ in range(20):
            f = first.clone().requires_grad_(True)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); loss = fn.apply(f, m, ap, sent, mlm)[0]; e1.record(); loss.backward(); e2.record()
            torch.cuda.synchronize()
            tf.append(e0.elapsed_time(e1) * 1e3); tb.append(e1.elapsed_time(e2) * 1e3)
        tf.sort(); tb.sort()
        print(f"B={B:3d} H={H:5d} {name:14s}: forward {tf[10]:7.1f} us  backward {tb[10]:7.1f} us", flush=True)
