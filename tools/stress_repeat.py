#!/usr/bin/env python3
"""Repeats one GPU test function in ONE process with allocator churn in between and prints every assertion message: finds tolerances that
sit inside the run-to-run noise of the fp32 atomics (a bound that fails one suite run in three passes every isolated run).
    cd tests && python ../tools/stress_repeat.py test_model_gpu test_training_without_returned_scores_equals_the_faithful_step 25 False True"""
import importlib, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
mod, fn, reps = importlib.import_module(sys.argv[1]), sys.argv[2], int(sys.argv[3])
params = [eval(a) for a in sys.argv[4:]] or [None]
fails = 0
for rep in range(reps):
    for prm in params:
        try:
            getattr(mod, fn)(*(() if prm is None else (prm,)))
        except AssertionError as e:
            fails += 1
            print("FAIL rep", rep, "param", prm, str(e)[:300].replace("\n", " "))
    junk = [torch.randn(1 + 37 * rep, 1000 + rep, device="cuda") for _ in range(3)]      # churn the caching allocator
    del junk
print("failures", fails, "of", reps * len(params))
