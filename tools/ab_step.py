#!/usr/bin/env python3
"""Same-process, interleaved A/B of the headline train step under in-process toggles (box-to-box clocks differ by 5-10 %, so only
ratios measured in one process count).

    python tools/ab_step.py base: paired:attr.defer_wgrads=False nt256:nt_mode=256      # each variant = name:KEY=VAL[,KEY=VAL...]

Keys: attr.NAME=<python literal> (a model attribute), ops.NAME=<literal> (a msa_amd.ops module attribute), nt_mode=<mmbert_gemm_nt_force
mode>, tn_splits=<mmbert_gemm_tn_force_splits>, tn_one=<mmbert_gemm_tn_force_one_launch>.  The library reads no environment variable (round 5); two BUILDS of it are compared with
alternating processes (tools/ab_lib.sh, MMBERT_LIB_PATH)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args

variants = []
for a in sys.argv[1:]:
    name, _, envs = a.partition(":")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
# 40-step windows (0.55 s): an 8-step window right behind a switch of variants still carries the previous variant's clock / power state --
# variants that differ in power draw read +-1-2 % wrong that way (round 4: three "+-0" results turned into -0.7 ... -2.5 % with longer
# windows and with alternating 600-step processes)
steps, rounds = int(os.environ.get("STEPS", 40)), int(os.environ.get("ROUNDS", 4))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig())
model.bert.set_joint_embeddings("mosei")
model.to(dev).train()
model.manual_seed(1234)
model.async_prologue = os.environ.get("ASYNC_PROLOGUE", "1") != "0"     # (the pool below is resident; attr.async_prologue=False for the A/B)
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
all_keys = {k for _, env in variants for k in env}


_hi = None


def step(i):
    global _hi
    if os.environ.get("AB_HIGH_PRIO_MAIN") == "1":          # the whole step on a high-priority stream (side streams stay at the default = lower)
        if _hi is None:
            _hi = torch.cuda.Stream(priority=-1)
        _hi.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(_hi):
            out, _ = model(**pool[i % 4])
            out[0].mean().backward()
            opt.step(); sched.step(); opt.zero_grad()
        torch.cuda.current_stream().wait_stream(_hi)
        return
    out, _ = model(**pool[i % 4])
    out[0].mean().backward()
    opt.step(); sched.step(); opt.zero_grad()


from msa_amd import ops as _ops
attr_defaults = {k[5:]: getattr(model, k[5:], None) for k in all_keys if k.startswith("attr.")}
ops_defaults = {k[4:]: getattr(_ops, k[4:]) for k in all_keys if k.startswith("ops.")}


def setenv(env):
    for k in all_keys:
        os.environ.pop(k, None)
    for k, v in attr_defaults.items():
        setattr(model, k, v)
    for k, v in ops_defaults.items():
        setattr(_ops, k, v)
    os.environ.update({k: v for k, v in env.items() if not k.startswith(("attr.", "ops.")) and k not in ("tn_splits", "nt_mode", "tn_one")})
    from msa_amd import _lib as _l
    _l.load().mmbert_gemm_tn_force_splits(int(env.get("tn_splits", 0)))
    _l.load().mmbert_gemm_tn_force_one_launch(int(env.get("tn_one", 0)))
    _l.load().mmbert_gemm_nt_force(int(env.get("nt_mode", 0)))
    for k, v in env.items():
        if k.startswith("attr."):                      # model attribute toggles: attr.NAME=python-literal
            setattr(model, k[5:], eval(v))
        elif k.startswith("ops."):                     # msa_amd.ops module toggles: ops.NAME=python-literal
            setattr(_ops, k[4:], eval(v))


ts = {n: [] for n, _ in variants}
for n, env in variants:
    setenv(env)
    for i in range(3):
        step(i)
torch.cuda.synchronize()
for r in range(rounds):
    for n, env in variants:
        setenv(env)
        step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        ts[n].append((time.perf_counter() - t0) / steps * 1e3)
base = sorted(ts[variants[0][0]])[rounds // 2]
for n, _ in variants:
    t = sorted(ts[n])
    print(f"{n:24s} median {t[rounds // 2]:7.3f} ms/step  (min {t[0]:.3f} max {t[-1]:.3f})  {16 / t[rounds // 2] * 1e3:7.1f} samples/s   x{t[rounds // 2] / base:.4f}", flush=True)
