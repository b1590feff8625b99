#!/usr/bin/env python3
"""Inference (eval mode, no autograd) forward time of the headline model, with and without the masked-row dedupe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
torch.manual_seed(0)
m = MMBertForPretraining(MMBertConfig()); m.bert.set_joint_embeddings("mosei"); m.set_alpha_beta(1.0, 1.0); m.cuda().eval()
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), "cuda") for i in range(4)]
for dd in (True, False, True, False):
    m.dedupe_masked_rows = dd
    with torch.no_grad():
        for i in range(5): m(**pool[i % 4])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): m(**pool[i % 4])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"dedupe={dd}: {dt * 1e3:.2f} ms per forward batch of 16 = {16 / dt:.0f} samples/s")
