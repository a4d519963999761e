"""Golden vectors for the permuter, BUILD container only.

(1) the reference's own known-answer self-test (modules/dynamic_modules/permuter.py:139-307) is
    executed with runpy and its fixture (inputs) and results captured as data;
(2) synthetic cases (ragged lengths, an all-coarse and an all-fine image, both orders) are run
    through the imported reference class.
Usage: python oracle/gen_golden_permuter.py
"""
import io
import os
import runpy
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

refimport.setup()
from modules.dynamic_modules.permuter import DualGrainSeperatePermuter  # noqa: E402


def run(perm, idx, grain):
    out = perm(torch.from_numpy(idx), torch.from_numpy(grain))
    back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
    return {k: v.numpy() for k, v in out.items()}, back.numpy()


if __name__ == "__main__":
    # (1) the reference's self-test: capture its fixture tensors and outputs
    with redirect_stdout(io.StringIO()):
        g = runpy.run_path(os.path.join(refimport.REF, "modules/dynamic_modules/permuter.py"), run_name="__main__")
    idx, grain = g["original_indices"].numpy(), g["grain_indices"].numpy()
    assert bool(torch.all(g["target_fine"] == g["original_indices"]))
    kw = dict(indices=idx.astype(np.int16), grain=grain.astype(np.int8))
    for order in ("region-first", "row-first"):
        perm = DualGrainSeperatePermuter(fine_position_order=order)
        out, back = run(perm, idx, grain)
        assert np.array_equal(back, idx)
        tag = order.split("-")[0]
        for k, v in out.items():
            kw["%s_%s" % (tag, k)] = v.astype(np.int16)
    save("permuter_reference_selftest", **kw)

    # (2) synthetic: B = 4, image 0 all coarse, image 1 all fine, images 2-3 Bernoulli(0.5 / 0.2)
    B = 4
    grain = np.stack([np.zeros((16, 16), np.int64), np.ones((16, 16), np.int64),
                      synth.bernoulli(4101, (16, 16), 0.5).astype(np.int64),
                      synth.bernoulli(4102, (16, 16), 0.2).astype(np.int64)])
    fine_codes = synth.randint(4103, (B, 32, 32), 1024)
    coarse_codes = synth.randint(4104, (B, 16, 16), 1024).repeat(2, axis=-1).repeat(2, axis=-2)
    gf = grain.repeat(2, axis=-1).repeat(2, axis=-2)
    idx = np.where(gf == 1, fine_codes, coarse_codes)
    kw = dict(indices=idx.astype(np.int16), grain=grain.astype(np.int8))
    for order in ("region-first", "row-first"):
        perm = DualGrainSeperatePermuter(fine_position_order=order)
        out, back = run(perm, idx, grain)
        assert np.array_equal(back, idx)
        tag = order.split("-")[0]
        for k, v in out.items():
            kw["%s_%s" % (tag, k)] = v.astype(np.int16)
    # forward_back with duplicated positions (later entry wins) and pads after EOS
    perm = DualGrainSeperatePermuter()
    cc = np.array([[5, 6, 7, 1025, 1024, 1024]], np.int64)
    cp = np.array([[3, 3, 10, 257, 256, 256]], np.int64)
    fc = np.array([[11, 12, 13, 14, 99, 1025, 1024]], np.int64)
    fp = np.array([[6, 7, 38, 39, 6, 1025, 1024]], np.int64)
    kw["dup_cc"], kw["dup_cp"], kw["dup_fc"], kw["dup_fp"] = cc, cp, fc, fp
    kw["dup_back"] = perm.forward_back(*(torch.from_numpy(a) for a in (cc, fc, cp, fp))).numpy().astype(np.int16)
    save("permuter_synthetic", **kw)
