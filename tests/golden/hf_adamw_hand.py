"""G11: a HAND-COMPUTED known-answer vector for the transformers-2.8.0 ``AdamW`` update (the optimizer REF:train.py:10,92 imports;
absent from the installed transformers 5.15.0, so no fixture can be generated from it here).

The numbers below were obtained by evaluating the PUBLISHED update rule

    m_t = b1 m_{t-1} + (1 - b1) g_t                     v_t = b2 v_{t-1} + (1 - b2) g_t^2
    p  <- p - lr sqrt(1 - b2^t) / (1 - b1^t) * m_t / (sqrt(v_t) + eps)           (bias correction in the step size, eps OUTSIDE it)
    p  <- p - lr wd p                                                            (decoupled decay AFTER the update, on the updated p)

in 40-digit decimal arithmetic, scalar by scalar (no tensor library, no optimizer implementation), with lr = 0.1, b1 = 0.9,
b2 = 0.999, eps = 1e-6.  First step of trajectory "a" by hand: m = 0.05, v = 2.5e-4, sqrt(v) + eps = 0.0158123883..., step size
0.1 sqrt(0.001) / 0.1 = 0.0316227766..., update 0.0316227766 x 0.05 / 0.0158123883 = 0.0999936762..., p = 0.9000063238...,
decayed by (1 - 0.1 x 0.01): 0.8991063178...

Each trajectory: (p0, weight decay, [g_1, g_2, g_3], [p_1, p_2, p_3], [m_1..3], [v_1..3]).  "b" has no decay, gradients ~ eps-scale
second moments and a ZERO gradient at step 2 (where eps inside / outside the bias correction differ most); "c" is a constant
gradient (|update| -> lr, decay visible).  Used by tests/test_oracle_golden.py (the oracle's mode "hf", CPU) and
tests/test_kernels_gpu.py (the HIP AdamW kernel's mode 0, through the C ABI)."""

LR, BETA1, BETA2, EPS = 0.1, 0.9, 0.999, 1e-6

TRAJECTORIES = {
    "a": (1.0, 0.01, [0.5, -0.25, 0.125],
          [0.89910631783119028775, 0.87160164689167443819, 0.83672305231357477175],
          [0.05, 0.02, 0.0305],
          [2.5e-4, 3.1225e-4, 3.2756275e-4]),
    "b": (-2.0, 0.0, [0.001, 0.0, -0.002],
          [-2.09693465699682844912, -2.16188552779872584918, -2.12835634975886816163],
          [1.0e-4, 9.0e-5, -1.19e-4],
          [1.0e-9, 9.99e-10, 4.998001e-9]),
    "c": (0.5, 0.01, [-3.0, -3.0, -3.0],
          [0.59939894697263904693, 0.69869880323436013997, 0.79789949615867791075],
          [-0.3, -0.57, -0.813],
          [9.0e-3, 1.7991e-2, 2.6973009e-2]),
}
