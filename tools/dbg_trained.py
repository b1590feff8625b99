"""Diagnostic: test_trained_state_gradients_match_oracle_without_calibrator repeated on clean / poisoned torch.empty memory, printing the
state each run ended in (the alignment head's gradient norm and error, the joint loss on the unseen batch): 26 steps at lr 5e-4 do not
reach one state.    python tools/dbg_trained.py clean poison clean poison"""
import os, sys, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import test_train_gpu as TT
import poison_empty
real = (torch.empty, torch.empty_like, torch.Tensor.new_empty)
for mode in sys.argv[1:]:
    if mode == "poison": poison_empty.install()
    else: torch.empty, torch.empty_like, torch.Tensor.new_empty = real
    try:
        TT.test_trained_state_gradients_match_oracle_without_calibrator()
        ok = "pass"
    except AssertionError as e:
        ok = "FAIL " + str(e)[:160].replace("\n", " ")
    r = json.load(open("gpurun_out/parity_trained.json"))
    a = r["grads"]["cls.align.weight"]
    print(mode, ok, "| align rel_err %.4f norm %.4f | moved %.5f | joint %.5f" % (a["rel_err"], a["norm"], r["moved_mean_abs"], r["losses"]["joint"]["ours"]), flush=True)
