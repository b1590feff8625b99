#!/usr/bin/env python3
"""Headline benchmark: MMBert train-step samples/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = forward of the three passes (text S=50, text+visual S=550, text+speech S=550: the
reference's ``MMBertForPretraining.forward``) in TRAIN mode (all dropouts on) + every loss +
backward + (N>1: RCCL gradient all-reduce) + the AdamW step, on synthetic MMBertDataset-shaped
batches already resident in HBM.  Workload = BASELINE.json configs[1]: 12 layers, d=768, 12 heads,
T=50, A=V=500, batch 16 per GPU, bf16 MFMA compute (fp32 master weights / accumulation).
Weak scaling: per-GPU batch is fixed, global batch = 16*N.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (gemm_nt: forward projections + input gradients, 2/3 of the
                  step's FLOPs): algorithmic FLOPs of its launches / their summed durations, taken live
                  with HIP events on the launch stream over a timed region of the SAME K steps that follows
                  the headline region (the ~340 event records per step cost ~0.5 ms per step of queue time,
                  so they are kept out of `value`; `ms_per_step_instrumented` is that second region).
  cpu_baseline -- the CPU oracle (oracle/mmbert_oracle.py, kind "port") timed on this box's host
                  cores on a bounded sample: the same model / shapes at batch 2, 1 warm-up + median of 3
                  fwd+bwd steps, plus BASELINE configs[0] exactly.
  fused1050    -- secondary: the same train step on ONE sequence text|visual|speech (S = 1050), a declared extension
                  (model.forward_fused); never the headline value.
config.tflop_per_sample is the dense algorithmic count; config.tflop_per_sample_executed subtracts the
work whose result is exactly zero and therefore skipped (MLM-head backward of unlabelled rows, backward of rows behind a
sequence's last unmasked key, attention over masked-out keys: DESIGN.md S2/S3); step_mfma_frac uses the executed count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch


def flop_parts(L, H, I, V, T, Pv, Pa, dv=35, ds=74):
    """Algorithmic forward FLOPs of one sample over the three passes, by component."""
    def one(S, P, D):
        return dict(proj=2 * S * L * (4 * H * H + 2 * H * I), attn=4 * S * S * H * L, head=2 * S * H * H + 2 * S * H * V, joint=2 * P * D * H)
    parts = [one(T, 0, 0), one(T + Pv, Pv, dv), one(T + Pa, Pa, ds)]
    return {k: sum(p[k] for p in parts) for k in parts[0]}


def flops_per_sample(L, H, I, V, T, Pv, Pa, dv=35, ds=74):
    """Algorithmic forward FLOPs of one sample over the three passes; train = 3x."""
    return sum(flop_parts(L, H, I, V, T, Pv, Pa, dv, ds).values())


def smi_readings():
    """sclk / mclk / power cap of GPU 0 as `rocm-smi` reports them, or the reason it did not answer.  Called at the very start of main(), BEFORE
    this process touches the GPU: the tool is a child process, and a process that has initialised HIP must not spawn-and-exec on this pool."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    try:
        r = subprocess.run([exe, "-d", "0", "--showclocks", "--showmaxpower", "--showpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=8)
        card = next(iter(json.loads(r.stdout).values()))
    except Exception as e:                                         # no permission, a timeout, text instead of JSON: recorded, never fatal
        return {"error": f"{type(e).__name__}: {str(e)[:120]}"}
    pick = {}
    for k, v in card.items():
        kl = k.lower()
        for tag, name in (("sclk", "sclk"), ("mclk", "mclk"), ("fclk", "fclk"), ("max graphics package power", "power_cap_w"), ("package power", "power_w"),
                          ("performance level", "perf_level")):
            if tag in kl and name not in pick:
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                pick[name] = v if name == "perf_level" else (float(m.group(0)) if m else str(v))
    return pick or {"error": "no clock fields in rocm-smi's answer", "keys": list(card)[:8]}


def box_probe(dev, ops, smi):
    """Which box is this?  The boxes of the pool differ by 4-5 % on the train step and by ~10 % on attention backward, so
    a driver line read against last round's means nothing without a yardstick taken on the same box.  Stand-alone microseconds (median of 5 after
    2 warm-up launches, HIP events on the launch stream) of three FIXED launches -- the ones that separated the box classes in
    profiles/r4_bench_final_profile_box.json against r4_bench_final.json -- on seeded operands, before anything else runs:
      vocab_nt_us   the vocabulary NT GEMM of the step: [18400, 768] x [30592, 768]^T + bias -> bf16                 (0.865 TFLOP)
      wgrad_tn44_us the 44-problem weight-gradient call: 11 layers x (W1, W2, Wqkv, Wo) at 13 850 rows               (2.15 TFLOP)
      attn_bwd_us   attention backward at the step's layout: 16 x 50 + 32 x 550 tokens, 12 heads, dropout 0.1, all keys
    `index` = the geometric mean of (reference microseconds / measured) over the three, reference = a FAST-class box of round 6
    (REF_US below): 1.00 on such a box, ~0.95 on the slow class; `value_normalised` on the bench line is value / index."""
    # (round 6, gpurun_out/r6_bench_mid.json: a box on which the round-5 tree reads 1 262 samples/s -- the fast class; the slow class reads ~1 195-1 220)
    REF_US = {"vocab_nt_us": 731.0, "wgrad_tn44_us": 1893.0, "attn_bwd_us": 257.0}
    H, I, heads, Vp, M, ra = 768, 3072, 12, 30592, 18400, 13850
    g = torch.Generator(device=dev)
    g.manual_seed(20260601)
    bf = torch.bfloat16
    rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev, generator=g) * scale).to(bf)

    def med_us(fn, warm=2, reps=7):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1))
        return round(sorted(ts)[len(ts) // 2], 1)
    out = {}
    t0 = time.perf_counter()
    A, W, bias = rnd(M, H), rnd(Vp, H, scale=0.02), torch.zeros(Vp, device=dev)
    C = torch.empty((M, Vp), device=dev, dtype=bf)
    # the GPU idles at a few hundred MHz when the process starts (rocm-smi above: sclk 185 MHz): ~0.15 s of the first yardstick itself
    # bring it to its working clocks before anything is timed (round 6's first probe read 983 us for a launch that takes 760 once warm)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.15:
        for _ in range(8):
            ops.gemm_nt(A, W, bias=bias, out=C)
        torch.cuda.synchronize()
    out["vocab_nt_us"] = med_us(lambda: ops.gemm_nt(A, W, bias=bias, out=C))
    del C, W
    probs = []
    for _ in range(11):
        du, y1, dz, gg, dqkv, x = rnd(ra, I, scale=0.01), rnd(ra, H), rnd(ra, H, scale=0.01), rnd(ra, I), rnd(ra, 3 * H, scale=0.01), rnd(ra, H)
        f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        probs += [(du, y1, f32(I, H), f32(I)), (dz, gg, f32(H, I), f32(H)), (dqkv, x, f32(3 * H, H), f32(3 * H)), (dz, y1, f32(H, H), f32(H))]
    out["wgrad_tn44_us"] = med_us(lambda: ops.gemm_tn_grouped(probs, accumulate=False), warm=2, reps=5)
    del probs
    lay = ops.SeqLayout([50] * 16 + [550] * 32, heads, dev)
    qkv, dctx = rnd(lay.tokens, 3 * H), rnd(lay.tokens, H, scale=0.01)
    kb = ops.pad_key_bias(torch.zeros(lay.tokens, device=dev), lay)
    drop = ops.make_drop(0.1, 1234, 7)
    ctx, lse = ops.attn_fwd(qkv, kb, lay, H, drop=drop)
    dqkv = torch.empty_like(qkv)
    out["attn_bwd_us"] = med_us(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, kb, lay, H, drop=drop, dqkv=dqkv))
    idx = 1.0
    for k, ref in REF_US.items():
        idx *= ref / out[k]
    out["index"] = round(idx ** (1.0 / 3.0), 4)
    out["reference_us"] = REF_US
    out["note"] = ("stand-alone us of three fixed launches taken before the headline (median of 3-5, HIP events); index = geometric mean of reference / measured, "
                   "reference = a fast-class box of round 6 (the round-5 tree reads 1 262 samples/s there)")
    props = torch.cuda.get_device_properties(dev)
    out["device"] = {"name": props.name, "cus": props.multi_processor_count, "clock_rate_khz": getattr(props, "clock_rate", None)}
    out["smi"] = smi
    torch.cuda.synchronize()
    out["probe_seconds"] = round(time.perf_counter() - t0, 3)
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--text", type=int, default=50)
    ap.add_argument("--pair", type=int, default=500)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--vocab", type=int, default=30522)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary fused-sequence (S = T+V+A) measurement")
    ap.add_argument("--no-dense-reference", action="store_true", help="skip the secondary measurement with the exact-zero short cuts off")
    ap.add_argument("--no-train-only", action="store_true", help="skip the secondary measurement without the returned prediction scores")
    ap.add_argument("--eval-dropout-off", action="store_true", help="diagnostic only: not a valid headline number")
    ap.add_argument("--no-skip-masked-keys", action="store_true", help="A/B: attention also visits the key tiles that are entirely masked out")
    ap.add_argument("--no-skip-padded-backward", action="store_true", help="A/B: backward also runs on the rows whose gradients are exactly zero")
    ap.add_argument("--no-sparse-top-layer", action="store_true", help="A/B: dense backward of the top encoder layer")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel path (RCCL process group, DataParallel wrapper, bucketed "
                    "all-reduce hooks, dynamic tile queue) even with ONE process: the N = 1 execution of the code the driver launches at N = 2/4/8")
    ap.add_argument("--bucket-mb", type=float, default=32.0, help="data parallel: smallest gradient slice handed to an all-reduce")
    ap.add_argument("--dp-wire", choices=["fp32", "bf16"], default="fp32", help="data parallel: dtype on the links (bf16: all-to-all + fp32 "
                    "accumulation + all-gather instead of the fp32 all-reduce; opt-in, halves the bytes)")
    ap.add_argument("--no-early-word", action="store_true", help="A/B: the word-embedding table's gradient reduced with the tail (round-2 form)")
    ap.add_argument("--sync-prologue", action="store_true", help="A/B: the step prologue on the compute stream (model.async_prologue = False)")
    ap.add_argument("--scores-fp32", action="store_true", help="return the prediction scores as fp32 (model.scores_dtype = torch.float32)")
    ap.add_argument("--no-dp-reference-legs", action="store_true", help="skip the N > 1 readiness legs (paired weight gradients on the plain step; bf16 wire under the wrapper)")
    ap.add_argument("--no-deterministic", action="store_true", help="skip the secondary measurement in deterministic mode")
    ap.add_argument("--deterministic", action="store_true", help="run the HEADLINE in deterministic mode (model.deterministic = True)")
    ap.add_argument("--no-scores-fp32", action="store_true", help="skip the secondary measurement with fp32 prediction scores (the reference's dtype)")
    ap.add_argument("--no-reference-default", action="store_true", help="skip the secondary measurement of the reference's own default model "
                    "(bert-large: 24-layer d=1024, T=P=40, batch 32: REF:train.py:28,32,38)")
    ap.add_argument("--preset", choices=["headline", "reference-default"], default="headline",
                    help="reference-default: measure ONLY the reference's default model and print its record (tools; never the driver's line)")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the three fixed yardstick launches in front of the headline (the `box` object)")
    ap.add_argument("--rank-report", action="store_true", help="N > 1 diagnostics even on one process: per-rank step times, exposed tail (on by default for N > 1)")
    a = ap.parse_args()
    # (rocm-smi is a child process: started before anything here initialises HIP, and only with the box yardstick)
    smi = smi_readings() if (int(os.environ.get("RANK", 0)) == 0 and not a.no_box_probe) else None

    from msa_amd import ops, parallel
    from msa_amd.data import synthetic_batch, batch_to, to_fused
    from msa_amd.model import MMBertConfig, MMBertForPretraining
    from msa_amd.trainer import build_optimizer, default_args

    # RCCL prints a version banner through C stdio on stdout when its first communicator comes up; the contract is ONE JSON line on
    # stdout, so stdout is parked on stderr while the process group exists and restored (C buffers flushed) right before the line
    saved_stdout = None
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 or a.force_dp:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    rccl_log = None
    if (int(os.environ.get("WORLD_SIZE", 1)) > 1 or a.force_dp) and os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN", ""):
        # N > 1 must be diagnosable from ONE run: RCCL's own account of what it built (rings / trees, channels, transport per peer) and
        # of what it picked per message size (algorithm / protocol / channels) goes to a per-process file; rank 0 condenses it into
        # config.dp.rccl and onto stderr at the end (the full file stays in /tmp).  (An NCCL_DEBUG of VERSION / WARN from the image's
        # environment is raised to INFO; an INFO / TRACE the caller set is left alone -- it then goes wherever the caller sent it.)
        rccl_log = f"/tmp/mmbert_rccl_{os.getpid()}.log"
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS=os.environ.get("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING"), NCCL_DEBUG_FILE=rccl_log)
    rank, local, world = parallel.init_from_env(force=a.force_dp)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    box = box_probe(dev, ops, smi) if (rank == 0 and not a.no_box_probe) else None

    timing = {"nt": [], "tn": [], "attn_fwd": [], "attn_bwd": []}
    record = [False]

    def wrap(name, fn, flop_fn, shape_fn=None):
        def inner(*args, **kw):
            if not record[0]:
                return fn(*args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            timing[name].append((flop_fn(*args, **kw), e0, e1, shape_fn(*args, **kw) if shape_fn else None))
            return out
        return inner

    # The per-launch wrappers are installed only around the instrumented legs: the headline region runs the model's default host path
    # (one composite C call per encoder layer, mmbert_layer_fwd / _bwd), the instrumented region the per-launch path -- the same kernels
    # with the same arguments (tests/test_model_gpu.py::test_composite_layer_calls_are_bit_identical), each launch between two events.
    originals, wrappers = {}, {}
    if not a.no_kernel_timing:
        originals = dict(gemm_nt=ops.gemm_nt, gemm_tn=ops.gemm_tn, gemm_tn_grouped=ops.gemm_tn_grouped, attn_fwd=ops.attn_fwd, attn_bwd=ops.attn_bwd)
        wrappers["gemm_nt"] = wrap("nt", ops.gemm_nt, lambda A, B, **kw: 2.0 * A.shape[0] * A.shape[1] * B.shape[0],
                                   lambda A, B, **kw: (A.shape[0], B.shape[0], A.shape[1]))
        wrappers["gemm_tn"] = wrap("tn", ops.gemm_tn, lambda A, B, W, **kw: 2.0 * A.shape[0] * A.shape[1] * B.shape[1])
        wrappers["gemm_tn_grouped"] = wrap("tn", ops.gemm_tn_grouped, lambda probs, **kw: sum(2.0 * p[0].shape[0] * p[0].shape[1] * p[1].shape[1] for p in probs))
        # executed attention FLOPs: with the valid-first packing forward visits all queries x the unmasked keys, backward the
        # unmasked rows only (lay.valid_host); otherwise the full S x S
        def attn_fl(lay, bwd):
            v = getattr(lay, "valid_host", None)
            if v is None:
                return sum(4.0 * n * n * 64 * lay.heads for n in lay.lens)
            return sum(4.0 * (k if bwd else n) * k * 64 * lay.heads for n, k in zip(lay.lens, v))
        wrappers["attn_fwd"] = wrap("attn_fwd", ops.attn_fwd, lambda qkv, kb, lay, H, **kw: attn_fl(lay, False))
        wrappers["attn_bwd"] = wrap("attn_bwd", ops.attn_bwd, lambda qkv, c, d, l, kb, lay, H, **kw: 2.5 * attn_fl(lay, True))

    def instrument(on):
        for k, fn in (wrappers if on else originals).items():
            setattr(ops, k, fn)
        record[0] = bool(on)

    if a.preset == "reference-default":
        rec = reference_default_leg(a, dev, ops, wrap_state=(timing, instrument))
        if saved_stdout is not None:
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(rec), flush=True)
        return

    H, L, I, V = a.hidden, a.layers, 4 * a.hidden, a.vocab
    torch.manual_seed(0)
    cfg = MMBertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=a.heads, intermediate_size=I)
    model = MMBertForPretraining(cfg)
    model.bert.set_joint_embeddings("mosei")
    model.set_alpha_beta(1.0, 1.0)
    model.to(dev)
    model.train(not a.eval_dropout_off)
    model.manual_seed(1234 + rank)
    # the batches of this benchmark are resident and complete before the timed region (the metric's definition): the step prologue may
    # read them on the model's input stream, ahead of the previous step's tail (model._prologue_stream)
    model.async_prologue = not a.sync_prologue
    model.skip_masked_keys = not a.no_skip_masked_keys
    model.skip_padded_backward = not a.no_skip_padded_backward
    model.sparse_top_layer_backward = not a.no_sparse_top_layer
    model.return_scores = True            # the reference returns the six score tensors; keep them materialised
    if a.scores_fp32:
        model.scores_dtype = torch.float32
    if a.deterministic:
        model.deterministic = True
    targs = default_args(train_batch_size=a.batch, learning_rate=5e-5)
    opt, sched = build_optimizer(model, targs, num_train_optimization_steps=10 * (a.steps + a.warmup))
    dp = (parallel.DataParallel(model, opt, bucket_mb=a.bucket_mb, force_dynamic_queue=a.force_dp, wire_dtype=torch.bfloat16 if a.dp_wire == "bf16" else None,
                                early_word_embedding=not a.no_early_word, equal_batch_shapes=True) if (world > 1 or a.force_dp) else None)
    pool = [batch_to(synthetic_batch(a.batch, a.text, a.pair, a.pair, vocab=V, seed=1 + i + 1000 * rank), dev) for i in range(4)]

    row_frac = []
    tail_events = []       # instrumented leg only: (before, after) finish_backward on the compute stream = the EXPOSED part of the exchange

    def step(i):
        out, _ = model(**pool[i % len(pool)])
        out[0].mean().backward()
        row_frac.append(getattr(model, "last_backward_row_fraction", 1.0))      # (known on the host once backward has sized its launches)
        if dp is not None:
            if record[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dp.finish_backward()
                e1.record()
                tail_events.append((e0, e1))
            else:
                dp.finish_backward()
        opt.step()
        sched.step()
        opt.zero_grad()
        return out[0]

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(a.steps):
        last = step(a.warmup + i)
    host_enqueued = time.perf_counter() - t0              # every call of the K steps has RETURNED: the host's side of the region (queue busy)
    torch.cuda.synchronize()
    own_done = time.perf_counter() - t0                   # this rank's GPU is done (the contract's value is taken behind the barrier)
    barrier()
    elapsed = time.perf_counter() - t0
    main_row_frac = row_frac[-a.steps:]                   # (the secondary loops below append their own)
    per_rank_ms = None
    if world > 1 or a.rank_report:
        # per-rank view of the SAME region: when this rank's GPU finished its K steps (before the barrier), i.e. who the others waited for
        t_own = torch.tensor([own_done], device=dev, dtype=torch.float64)
        if world > 1:
            allt = [torch.zeros_like(t_own) for _ in range(world)]
            torch.distributed.all_gather(allt, t_own)
            per_rank_ms = [round(1e3 * float(x) / a.steps, 3) for x in allt]
        else:
            per_rank_ms = [round(1e3 * own_done / a.steps, 3)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(last.detach())
    # The roofline leg: the SAME K steps once more, bracketed the same way, with a HIP event pair around every GEMM / attention launch
    # on the launch stream.  Kept out of the headline region because the instrumentation is not free: ~340 event records per step
    # are ~340 extra packets in the queue between kernels (round 2 timed both in one pass: 16.4 ms per step against 15.9 without the
    # events, same box, same process); `ms_per_step_instrumented` reports what this leg took.
    elapsed_instr = None
    if not a.no_kernel_timing:
        instrument(True)
        ti0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)
        torch.cuda.synchronize()
        barrier()
        elapsed_instr = time.perf_counter() - ti0
        instrument(False)

    def timed_leg(fn, nwarm=None):
        """A secondary leg: ``nwarm`` untimed calls of ``fn(i)``, then the contract's bracket (synchronize + barrier on both sides,
        max over ranks) around a.steps calls.  Returns seconds."""
        for i in range(min(a.warmup, 3) + 1 if nwarm is None else nwarm):
            fn(i)
        torch.cuda.synchronize()
        barrier()
        tl0 = time.perf_counter()
        for i in range(a.steps):
            fn(i)
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - tl0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        return el

    def leg_record(el, **extra):
        return dict({"value": round(a.steps * a.batch * world / el, 2), "unit": "samples/s", "ms_per_step": round(1e3 * el / a.steps, 3)}, **extra)

    # For reference: the same step with every exact-zero short cut switched off (attention visits the masked-out keys, backward
    # runs on all rows and densely through the top layer); identical gradients up to fp32 summation order, never the headline.
    dense_ref = None
    if not a.no_dense_reference and (model.skip_masked_keys or model.skip_padded_backward or model.sparse_top_layer_backward):
        saved_flags = (model.skip_masked_keys, model.skip_padded_backward, model.sparse_top_layer_backward)
        model.skip_masked_keys = model.skip_padded_backward = model.sparse_top_layer_backward = False
        delapsed = timed_leg(step)
        model.skip_masked_keys, model.skip_padded_backward, model.sparse_top_layer_backward = saved_flags
        dense_ref = leg_record(delapsed, note="same step with the exact-zero short cuts off: all keys, all rows in backward (MLM-head backward still on labelled rows)")

    # Secondary: the prediction scores in the REFERENCE's dtype.  REF:MMBertForPretraining.py:445-449 returns fp32 score tensors; the
    # headline step returns bf16 views of the materialised logits (config.scores_dtype).  model.scores_dtype = torch.float32 makes the
    # vocabulary GEMM store its fp32 accumulators instead: bit-identical losses and gradients (test_scores_dtype_float32_for_numpy_consumers),
    # twice the bytes written by that launch.
    scores_fp32 = None
    if not a.no_scores_fp32 and model.scores_dtype != torch.float32:
        saved_dtype = model.scores_dtype
        model.scores_dtype = torch.float32
        sel = timed_leg(step)
        model.scores_dtype = saved_dtype
        scores_fp32 = leg_record(sel, note="model.scores_dtype = torch.float32: the six returned prediction-score tensors in the reference's dtype "
                                           "(REF:MMBertForPretraining.py:445-449); same losses and gradients as the headline step")

    # Secondary: model.deterministic = True -- ordered sums instead of fp32 atomics (CE loss sums, the heads' skinny products, bias /
    # LayerNorm partial sums, the embedding scatter through per-id run sums): bit-identical losses and gradients run to run
    # (tests/test_train_gpu.py::test_deterministic_mode_gives_bit_identical_steps); what it costs in the step is this leg against the headline.
    det_leg = None
    if not a.no_deterministic:
        was_det = model.deterministic
        model.deterministic = True
        try:
            dl = timed_leg(step)
        finally:
            model.deterministic = was_det
        det_leg = leg_record(dl, note="model.deterministic = True (mmbert_set_deterministic): every fp32 sum in a schedule-independent order; "
                                      "bit-identical losses and gradients run to run")

    # Secondary (N > 1 readiness): the weight gradients per layer PAIR instead of all layers in one call -- the form the data-parallel
    # wrapper has to use (its hook needs a layer's gradient slice when the layer is done), measured on the plain step so that a SCALE
    # point can be read against a world-1 number of the same code path.
    paired = None
    if not a.no_dp_reference_legs and dp is None:
        was_defer = getattr(model, "defer_wgrads", None)
        model.defer_wgrads = False
        pel = timed_leg(step)
        model.defer_wgrads = was_defer
        paired = leg_record(pel, note="model.defer_wgrads = False: one weight-gradient launch per encoder-layer pair (what DataParallel's per-layer hooks need)")
    # Secondary under the data-parallel wrapper: bf16 on the links (all-to-all + fp32 sum in rank order + all-gather instead of the fp32
    # all-reduce: DataParallel(wire_dtype=torch.bfloat16)); at world size 1 the wire path is forced so that its kernels run
    wire_leg = None
    # (only in the forced single-process form by default: at N > 1 the headline's collectives are the ones to be measured, and a leg that
    # switches the exchange protocol between timed regions is one more thing that could go wrong on the first real multi-GPU run;
    # --dp-wire bf16 measures that protocol as the headline of its own run)
    if not a.no_dp_reference_legs and dp is not None and a.dp_wire == "fp32" and world == 1:
        bk = dp.bucketer
        saved_wire = (bk.wire_dtype, bk.force_wire_path)
        bk.wire_dtype, bk.force_wire_path = torch.bfloat16, world == 1
        try:
            wel = timed_leg(step)
        finally:
            bk.wire_dtype, bk.force_wire_path = saved_wire
        wire_leg = leg_record(wel, note="DataParallel wire dtype bf16 (all-to-all + fp32 accumulation in rank order + all-gather); "
                                        + ("forced through both stages at world size 1" if world == 1 else f"world size {world}"))

    # Secondary: the train step as trainer.py consumes it -- model.return_scores = False: the six prediction-score tensors that
    # the reference's forward returns and its trainer never reads are not produced, so the MLM head runs on the labelled rows
    # only (forward too) and the encoder leaves out the rows that only those scores would read.  Same losses and gradients
    # (tests/test_model_gpu.py); not the headline, which keeps every returned tensor.
    train_only = None
    if not a.no_train_only:
        model.return_scores = False
        telapsed = timed_leg(step)
        model.return_scores = True
        train_only = leg_record(telapsed, note="model.return_scores = False: losses and gradients as in the headline step, the returned prediction scores are None")

    # Secondary: the same train step on the fused single sequence text | visual | speech (S = T + V + A = 1050), the shape
    # BASELINE.json's metric name quotes.  The reference never builds that sequence (its step is the three passes above), so
    # this is a declared extension (model.forward_fused, checked against oracle.fused_forward) and never the headline value.
    fused = None
    if not a.no_fused:
        fpool = [to_fused(b) for b in pool]

        def fstep(i):
            out, _ = model.forward_fused(**fpool[i % len(fpool)])
            out[0].mean().backward()
            if dp is not None:
                dp.finish_backward()
            opt.step()
            sched.step()
            opt.zero_grad()

        felapsed = timed_leg(fstep)
        S = a.text + 2 * a.pair
        ffl = 3.0 * (2 * S * L * (4 * H * H + 2 * H * I) + 4 * S * S * H * L + 2 * S * H * H + 2 * S * H * V + 2 * a.pair * (35 + 74) * H)
        fused = leg_record(felapsed, seq_len=S, tflop_per_sample=round(ffl / 1e12, 4),
                           note="declared extension (one pass over text|visual|speech; not in the reference, no reference parity): model.forward_fused")

    # Secondary: the reference's OWN default model (REF:train.py:28,32,38,70 -- bert-large-uncased: 24 layers, d = 1024, 16 heads, I = 4096,
    # max_seq_length 40, pair length == text length, train_batch_size 32), same train step.  Never the headline (BASELINE.json quotes d = 768).
    ref_default = None
    if not a.no_reference_default and world == 1 and dp is None:
        ref_default = reference_default_leg(a, dev, ops, wrap_state=(timing, instrument))

    samples = a.steps * a.batch * world
    value = samples / elapsed
    fps = 3.0 * flops_per_sample(L, H, I, V, a.text, a.pair, a.pair)
    # the MLM head's backward runs on the labelled rows only (their CE gradient is the only non-zero one; ~2 % of the
    # rows with 15 % masking of the text tokens): FLOPs actually executed = dense count - the skipped backward products
    tokens = a.text + 2 * (a.text + a.pair)
    # ... and the rest of backward runs on the rows that can have a gradient (valid-first packing): f = their share,
    # measured per step; attention forward skips the masked-out key tiles (~ the same share of its keys)
    f = sum(main_row_frac) / max(1, len(main_row_frac))
    fp = flop_parts(L, H, I, V, a.text, a.pair, a.pair)
    head_bwd = 2.0 * fp["head"] * (0.15 * 3 * a.text / tokens)
    fps_exec = (fp["proj"] + fp["attn"] * f + fp["head"] + fp["joint"]) + 2.0 * (fp["proj"] * f + fp["attn"] * f * f + fp["joint"] * f) + head_bwd
    # ... and what the dense_backward_reference leg executes: every row and every key (f = 1); its MLM-head backward still runs on the
    # labelled rows only (round 4 divided the DENSE count by that leg's time: 15 % too much)
    fps_exec_dense = (fp["proj"] + fp["attn"] + fp["head"] + fp["joint"]) + 2.0 * (fp["proj"] + fp["attn"] + fp["joint"]) + head_bwd
    res = {
        "metric": "train-step samples/sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "ms_per_step_instrumented": (round(1e3 * elapsed_instr / a.steps, 3) if elapsed_instr else None), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"MMBertForPretraining train step, 3 passes S={a.text}/{a.text + a.pair}/{a.text + a.pair}, "
                               f"{L}-layer d={H} heads={a.heads} vocab={V}, T={a.text} A={a.pair} V={a.pair}, dropout on, AdamW",
                   "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                   "tflop_per_sample": round(fps / 1e12, 4), "tflop_per_sample_executed": round(fps_exec / 1e12, 4),
                   "scores_dtype": {torch.bfloat16: "bf16", torch.float32: "fp32"}[model.scores_dtype],
                   "mlm_backward": "labelled rows only (exact: unlabelled rows have zero CE gradient)",
                   "backward_row_fraction": round(f, 4),
                   "backward_rows": "rows behind a sequence's last unmasked key and without a label have zero gradients in every layer: backward skips them (exact)",
                   # round 6: how the step is scheduled (the work itself is unchanged: bit-identical with every one of these off, tests/test_train_gpu.py)
                   "scheduling": {"few_row_weight_gradients_in_the_deferred_call": bool(getattr(model, "late_wgrads", False) and dp is None),
                                  "side_streams": {"heads": bool(getattr(model, "heads_side_stream", False)),
                                                   "deferred_weight_gradient_call": bool(getattr(model, "wgrad_side_stream", False) and dp is None),
                                                   "pair_projections": bool(getattr(model, "pairs_side_stream", False)),
                                                   "transposed_weight_copies": bool(getattr(ops, "SIDE_TRANSPOSES", False))}}},
        "final_loss": round(loss, 4),
        # the host's share of the headline region: wall time until the last step() call returned (the GPU still has queued work then);
        # ms_per_step - host_enqueue_ms = how far ahead of the GPU the host runs.  Under the data-parallel wrapper this includes its hooks
        # and RCCL's host calls (config.dp.host_enqueue_ms: VERDICT r5 item 8)
        "host_enqueue_ms": round(1e3 * host_enqueued / a.steps, 3),
        "step_mfma_frac": round(fps_exec * value / world / 2.5e15, 4),
    }
    if dp is not None:
        res["config"]["dp"] = {"backend": torch.distributed.get_backend(), "forced_single_process": bool(a.force_dp and world == 1),
                               "gemm_tile_queue": "dynamic, one counter per XCD", "bucket_mb": a.bucket_mb, "wire_dtype": a.dp_wire,
                               "word_embedding_table": ("reduced right after the MLM head's backward; lookup rows exchanged in compact form"
                                                        if dp.early_word else "with the tail"),
                               "all_reduce_calls_per_step": dp.bucketer.calls_per_step if hasattr(dp.bucketer, "calls_per_step") else None,
                               "per_rank_ms_per_step": per_rank_ms,
                               "host_enqueue_ms": round(1e3 * host_enqueued / a.steps, 3),
                               "exposed_tail_us": exposed_tail(tail_events, world, dev),
                               "exposed_tail_note": "GPU time of the compute stream inside finish_backward() (instrumented leg): what of the "
                                                    "gradient exchange is NOT hidden under backward -- the last bucket, the compact row exchange, the waits"}
    if box is not None:
        res["box"] = box
        res["value_normalised"] = round(value / box["index"], 2)      # what this step would read on the reference (fast) box class
    if dense_ref is not None:
        res["dense_backward_reference"] = dense_ref
    if dense_ref is not None:
        dense_ref["tflop_per_sample_executed"] = round(fps_exec_dense / 1e12, 4)
    if scores_fp32 is not None:
        res["scores_fp32"] = scores_fp32
    if det_leg is not None:
        res["deterministic"] = det_leg
    if paired is not None:
        res["paired_wgrads"] = paired
    if wire_leg is not None:
        res["dp_wire_bf16"] = wire_leg
    res["config"]["deterministic"] = bool(a.deterministic)
    if train_only is not None:
        res["train_only"] = train_only
    if fused is not None:
        res["fused1050"] = fused
    if ref_default is not None:
        res["reference_default"] = ref_default
    if saved_stdout is not None:
        import ctypes
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if dp is not None and rccl_log is not None and rank == 0:
        res["config"]["dp"]["rccl"] = rccl_summary(rccl_log)
    if rank == 0:
        if not a.no_kernel_timing and timing["nt"]:
            kern = {}
            for k, lst in timing.items():
                if lst:
                    ms = sum(e0.elapsed_time(e1) for _, e0, e1, _ in lst)
                    fl = sum(f for f, _, _, _ in lst)
                    kern[k] = (fl, ms, len(lst))
            fl, ms, n = kern["nt"]
            ach = fl / (ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic_per_launch("gemm_nt8_kernel")
            res["roofline"] = {"bound": "mfma", "kernel": "the NT GEMM family behind mmbert_gemm_nt: gemm_nt8_kernel (bf16 MFMA 16x16x32, every fused epilogue; 64-deep K tiles in "
                                                          "8 phases, LDS-DMA half-tiles 3 ahead: one 192/224/256x256 tile per workgroup for single-round launches, a stream of "
                                                          "224x256 tiles per workgroup for multi-round ones)",
                               "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
                               "traffic": traffic, "traffic_source": traffic_src, "launches": n, "avg_launch_us": round(1e3 * ms / n, 2),
                               "share_of_step_time": round(ms * 1e-3 / elapsed_instr, 3),
                               "timed_in": f"a second pass of the same {a.steps} steps with an event pair around every GEMM / attention launch (ms_per_step_instrumented)",
                               # whole-step MFMA fractions of 2.5 PF, side by side: FLOPs actually executed by the headline step, and the
                               # FLOPs executed by the step with every exact-zero short cut switched off at that step's speed
                               "frac_step_executed": round(fps_exec * value / world / 2.5e15, 4),
                               "frac_step_dense_shortcuts_off": (round(fps_exec_dense * dense_ref["value"] / world / 2.5e15, 4) if dense_ref else None)}
            for k in ("tn", "attn_fwd", "attn_bwd"):
                if k in kern:
                    fl2, ms2, n2 = kern[k]
                    res["roofline"]["gemm_tn" if k == "tn" else k] = {"achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 1), "launches": n2,
                                                                      "share_of_step_time": round(ms2 * 1e-3 / elapsed_instr, 3)}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a, L, H, I, V)
        print(json.dumps(res), flush=True)


def reference_default_leg(a, dev, ops, wrap_state):
    """The reference's default configuration (bert-large, T = P = 40, batch 32) through the same train step: samples/s, the persistent NT
    family's in-situ rate at its {1024, 3072, 4096} shapes (HIP events, a second pass like the headline's roofline leg) and which
    kernel / tile / tile walk every shape was dispatched to (mmbert_gemm_nt_describe)."""
    from msa_amd.data import synthetic_batch, batch_to
    from msa_amd.model import MMBertConfig, MMBertForPretraining
    from msa_amd.trainer import build_optimizer, default_args
    timing, instrument = wrap_state
    L, H, heads, I, V, T, B = 24, 1024, 16, 4096, a.vocab, 40, 32
    torch.manual_seed(0)
    model = MMBertForPretraining(MMBertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=I))
    model.bert.set_joint_embeddings("mosei")
    model.set_alpha_beta(1.0, 1.0)
    model.to(dev)
    model.train()
    model.manual_seed(4321)
    model.async_prologue = not a.sync_prologue
    steps, warm = max(4, a.steps // 2), min(a.warmup, 3) + 1
    opt, sched = build_optimizer(model, default_args(train_batch_size=B, learning_rate=5e-5), 10 * (2 * steps + warm))
    pool = [batch_to(synthetic_batch(B, T, T, T, vocab=V, seed=50 + i), dev) for i in range(4)]

    def step(i):
        out, _ = model(**pool[i % len(pool)])
        out[0].mean().backward()
        opt.step(); sched.step(); opt.zero_grad()
        return out[0]
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rec = {"value": round(steps * B / el, 2), "unit": "samples/s", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
           "final_loss": round(float(last.detach()), 4),
           "workload": f"REF:train.py:28,32,38 defaults: bert-large ({L}-layer d={H} heads={heads} I={I}), T=P={T} (S = {T}/{2 * T}/{2 * T}), batch {B}, "
                       "train mode (dropout on), AdamW; pair-position MLM labels = copy of the text labels (REF:trainer.py:50,53)"}
    from msa_amd.model import _auto_defer_wgrads
    rec["weight_gradients"] = ("all layers in one call of whole 256-tile rounds at the end of backward" if _auto_defer_wgrads(H, I, L, dev)
                               else "one launch per layer pair") + " (chosen by shape: model._auto_defer_wgrads)"
    S3 = T + 4 * T
    fwd = 2 * S3 * L * (4 * H * H + 2 * H * I) + 4 * (T * T + 2 * (2 * T) ** 2) * H * L + 2 * S3 * H * H + 2 * S3 * H * V + 2 * T * (35 + 74) * H
    rec["tflop_per_sample"] = round(3.0 * fwd / 1e12, 4)
    rec["step_mfma_frac_dense"] = round(3.0 * fwd * rec["value"] / 2.5e15, 4)
    if not a.no_kernel_timing:
        marks = {k: len(v) for k, v in timing.items()}
        mark = marks["nt"]
        instrument(True)
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        instrument(False)
        by = {}
        for fl, e0, e1, shp in timing["nt"][mark:]:
            d = by.setdefault((shp[1], shp[2]), [0.0, 0.0, 0, 0])
            d[0] += fl; d[1] += e0.elapsed_time(e1); d[2] += 1; d[3] = max(d[3], shp[0])
        for k, n0 in marks.items():                              # (the headline's roofline records stay what they were)
            del timing[k][n0:]
        shapes = {}
        tot_f = tot_ms = 0.0
        for (N, K), (fl, ms, n, M) in sorted(by.items()):
            big = ops.gemm_nt_describe(M, N, K)
            if M < 512:                                           # the few-row products of the sparse paths are not this family
                continue
            tot_f += fl; tot_ms += ms
            shapes[f"N{N}_K{K}"] = {"launches_per_step": round(n / steps, 1), "rows_max": M, "avg_us": round(1e3 * ms / n, 1),
                                    "tflops": round(fl / (ms * 1e-3) / 1e12, 1), "dispatch": big}
        if tot_ms:
            rec["gemm_nt"] = {"achieved": round(tot_f / (tot_ms * 1e-3) / 1e12, 1), "frac": round(tot_f / (tot_ms * 1e-3) / 2.5e15, 4), "shapes": shapes,
                              "note": "all mmbert_gemm_nt launches of at least 512 rows, by (N, K); dispatch = mmbert_gemm_nt_describe at the largest row count seen"}
    del model, opt, pool
    torch.cuda.empty_cache()
    return rec


def exposed_tail(events, world, dev):
    """Mean GPU microseconds inside finish_backward() per step, per rank (list over ranks)."""
    if not events:
        return None
    torch.cuda.synchronize()
    us = 1e3 * sum(e0.elapsed_time(e1) for e0, e1 in events) / len(events)
    if world > 1:
        t = torch.tensor([us], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        return [round(float(x), 1) for x in allt]
    return [round(us, 1)]


def rccl_summary(path, limit=24):
    """RCCL's own INFO lines, condensed: how many channels / rings / trees it built, the transports per peer, and the distinct
    (collective, algorithm, protocol, channels) choices it logged -- also echoed to stderr so that the driver's log carries them."""
    import re
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError as e:
        return {"error": str(e)}
    pick = [l for l in lines if re.search(r"(Channel \d+/\d+ *:|Ring \d+ *:|Trees? |nChannels|via P2P|via SHM|via NET|Algo|algorithm|protocol|Using network|comm 0x.* rank .* nranks|Connected all)", l)]
    tuning = sorted({re.sub(r"^.*?NCCL INFO ", "", l) for l in lines if "TUNING" in l or re.search(r"(AllReduce|AllGather|AllToAll|ReduceScatter|Broadcast).*(Algo|algo|proto)", l)})
    # (one line per ring / tree / channel would crowd everything else out of the record: counts, plus the first two of each kind)
    seen, brief = {}, []
    for l in pick:
        body = re.sub(r"^.*?NCCL INFO ", "", l)
        kind = re.match(r"(Tree|Ring|Channel)\b", body)
        if kind:
            seen[kind.group(1)] = seen.get(kind.group(1), 0) + 1
            if seen[kind.group(1)] > 2:
                continue
        brief.append(body)
    nch = len({m.group(1) for l in lines for m in [re.search(r"Channel (\d+)/\d+", l)] if m})
    nranks = max([int(m.group(1)) for l in lines for m in [re.search(r"nranks (\d+)", l)] if m] or [1])
    # xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce over ONE channel moves a 32-MiB bucket over one link per hop
    # (~5 ms for the step's 465 MB); RCCL spreads a collective over its channels, each routed over another link, so the
    # fp32 buckets need >= 7 channels (or a non-ring algorithm) at 8 ranks to use the fabric.  Recorded, and loud on stderr -- not fatal:
    # the bench line must still be printed.
    multi_link = None if nranks <= 2 else bool(nch >= min(7, nranks - 1))
    if multi_link is False:
        print(f"[bench] WARNING: RCCL built {nch} channel(s) for {nranks} ranks: the gradient all-reduce is bound by {nch} xGMI link(s) per hop", file=sys.stderr, flush=True)
    out = {"log": path, "lines": len(lines), "channels": nch, "nranks": nranks, "all_reduce_uses_all_links": multi_link,
           "p2p_links": sum("via P2P" in l for l in lines), "shm_links": sum("via SHM" in l for l in lines),
           "tree_ring_channel_lines": seen, "choices": tuning[:limit], "init": brief[:limit]}
    print("[bench] RCCL summary:", json.dumps(out)[:4000], file=sys.stderr, flush=True)
    return out


def csrc_digest():
    """sha256 over the kernel sources: PMC summaries under profiles/ carry the digest of the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "msa_amd", "csrc")
    for f in sorted(x for x in os.listdir(d) if x.endswith((".hip", ".h"))):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(kernel_substr, path=None):
    """HBM/fabric bytes per launch of the dominant kernel family from the committed PMC summary (two separate rocprofv3 --pmc
    passes over this same bench command, FETCH_SIZE doubled as the gfx950 correction asks: tools/pmc_traffic.py).  Counters
    cannot be collected from inside the timed run, so this is the measured figure of record -- REFUSED (None, with the reason)
    when the summary was measured on other kernel sources than the ones in the tree (its ``# csrc_sha256:`` line)."""
    import glob
    cands = [path] if path else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.csv")), reverse=True)
    for path in cands:
        try:
            import csv
            lines = open(path).read().splitlines()
            stamp = [l.split(":", 1)[1].strip() for l in lines if l.startswith("# csrc_sha256:")]
            rel = os.path.relpath(path, ROOT)
            if not stamp:
                return None, f"{rel}: no csrc_sha256 stamp (measured on unknown sources): refused"
            if stamp[0] != csrc_digest():
                return None, f"{rel}: stale (measured on kernel sources {stamp[0]}, tree has {csrc_digest()}): refused"
            subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
            rows = [r for r in csv.DictReader(l for l in lines if not l.startswith("#")) if any(k in r["kernel"] for k in subs)]
            n = sum(int(r["launches"]) for r in rows)
            if not n:
                continue
            mb = sum(float(r["total_MB"]) * int(r["launches"]) for r in rows) / n
            return round(mb * 1e6), f"{rel} (mean over {n} launches, read x2-corrected + written; sources {stamp[0]})"
        except Exception as e:
            return None, f"{path}: {e}"
    return None, None


def usable_cores(cap=32):
    """Threads for the CPU baseline: the cgroup CPU quota if there is one, else the affinity mask, capped at 32
    (256 torch threads on the GPU box's 256-core host took 429 s for the same step that needs 12 s on 8 cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(a, L, H, I, V):
    """the CPU oracle on the host cores -- C2 shapes at batch 2 (same 12-layer d=768 model, S = 50/550/550), fp32,
    dropout on, forward+loss and forward+backward, 1 warm-up then the median of 3; and C1 exactly (BASELINE.json configs[0])."""
    from oracle import mmbert_oracle as O
    from msa_amd.data import synthetic_batch
    cores = usable_cores()
    torch.set_num_threads(cores)

    def timed(cfg, B, T, P, iters):
        p = {k: v.requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
        batch = synthetic_batch(B, T, P, P, vocab=V, seed=1)
        tf, tb = [], []
        for it in range(iters + 1):
            for v in p.values():
                v.grad = None
            t1 = time.perf_counter()
            out, _ = O.pretraining_forward(p, cfg, **batch, train=True)
            t2 = time.perf_counter()
            out[0].mean().backward()
            t3 = time.perf_counter()
            if it:                                              # iteration 0 is the warm-up
                tf.append(t2 - t1); tb.append(t3 - t1)
        return sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2]
    t0 = time.perf_counter()
    B = 2
    f2, b2 = timed(dict(hidden=H, layers=L, heads=a.heads, intermediate=I, vocab=V, dataset="mosei", alpha=1.0, beta=1.0), B, a.text, a.pair, 3)
    f1, b1 = timed(dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=V, dataset="mosei", alpha=1.0, beta=1.0), 2, 50, 64, 3)
    c1 = {"fwd_loss_samples_per_s": round(2 / f1, 2), "fwd_bwd_samples_per_s": round(2 / b1, 2),
          "sample": "BASELINE configs[0]: 2-layer d=128, batch 2, T=50 A=V=64, 1 warm-up + median of 3"}
    return {"value": round(B / b2, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+loss+bwd (fp32, dropout on), same model and S={a.text}/{a.text + a.pair}/{a.text + a.pair} shapes at batch {B} "
                      f"(BASELINE.md S4: C2 shapes at B=2), 1 warm-up + median of 3: fwd+loss {f2:.2f} s, fwd+bwd {b2:.2f} s; "
                      f"whole leg {time.perf_counter() - t0:.0f} s",
            "fwd_loss_samples_per_s": round(B / f2, 4), "configs0": c1}


if __name__ == "__main__":
    main()
