"""GPU (-m gpu): the persistent role-alternating form of pass 1 (csrc/vq_assign_pipe.hip -- tuning build only, not the
default: it is bit-exact but not yet faster, see its header) against the product kernel and the oracle: dense, select fused
(dual and triple), codes-only, special values.  Runs in a child process with DVQ_LIBRARY = libdvq_tuning.so."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual, vq_assign_routed_triple
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, D = 1024, 256
E = synth.codebook_trained(K, D)
def both(fn):
    out = []
    for pipe in (0, 1):
        assert _lib.lib.dvq_tuning_set(b"pipe", pipe) == 0
        out.append(fn())
    _lib.lib.dvq_tuning_set(b"pipe", 0)
    return out
# dense, B = 128 (1024 blocks = 4 per workgroup), masked, NaN / Inf / huge tokens (exact list) -- vs the product kernel and the oracle
B = 128
z = synth.z_tokens(E, B, 32, 32, 8801)
z[3, 5, 0, 0] = np.nan; z[4, :, 1, 1] = np.inf; z[5, :, 3, 3] *= np.float32(1e6); z[6, :, 0, 1] = 0.0
mask = np.where(synth.bernoulli(8802, (B, 1, 32, 32), 0.5), 1.0, 0.25).astype(np.float32)
(zq0, c0, l0), (zq1, c1, l1) = both(lambda: vq_assign(t(z), t(E), _CodebookPrep(), t(mask)))
assert torch.equal(c0, c1) and bool(((zq0 == zq1) | (torch.isnan(zq0) & torch.isnan(zq1))).all())
sel = np.arange(0, B, 16)
o = oracle.vq_assign_nchw(z[sel], E, mask[sel])
assert np.array_equal(c1.cpu().numpy()[sel].reshape(len(sel), -1), o["codes"])
assert np.array_equal(zq1.cpu().numpy()[sel], o["zq"], equal_nan=True)
(_, c2, _), (_, c3, _) = both(lambda: vq_assign(t(z), t(E), _CodebookPrep(), None, want_zq=False, want_loss=False))
assert torch.equal(c2, c3)
# dual, select fused, B = 128: all images vs the oracle
hf, hc = synth.z_tokens(E, B, 32, 32, 8811), synth.z_tokens(E, B, 16, 16, 8812)
ent = synth.entropy_map(8813, B, 16, 16)
THR = 1.6777750253677368
r0, r1 = both(lambda: vq_assign_routed_dual(t(hc), t(hf), t(E), _CodebookPrep(), entropy=t(ent), threshold=THR))
for k in ("zq", "codes", "indices", "codebook_mask", "gate"):
    assert torch.equal(r0[k], r1[k]), k
og = oracle.entropy_gate(ent, THR)
osel = oracle.route_select_dual(og, hc, hf)
o = oracle.vq_assign_nchw(osel["h_dual"], E, osel["codebook_mask"])
assert np.array_equal(r1["codes"].cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(r1["zq"].cpu().numpy(), o["zq"])
ol = float(oracle.vq_loss(o["sqerr"], o["numel"], 0.25))
assert abs(float(r1["loss"][1]) - ol) <= 1e-5 * abs(ol)
# triple, select fused, f32 logits, B = 128
hm, hco = synth.z_tokens(E, B, 16, 16, 8821), synth.z_tokens(E, B, 8, 8, 8822)
lg = synth.grain_logits_triple(8823, B, 8, 8)
q0, q1 = both(lambda: vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), _CodebookPrep(), t(lg)))
for k in ("zq", "codes", "indices", "codebook_mask"):
    assert torch.equal(q0[k], q1[k]), k
osel = oracle.route_select_triple(lg, hco, hm, hf)
o = oracle.vq_assign_nchw(osel["h_triple"], E, osel["codebook_mask"])
assert np.array_equal(q1["codes"].cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(q1["zq"].cpu().numpy(), o["zq"])
print("PIPE_FORM_OK")
"""


def test_pipe_form_bit_exact(dev):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune = os.path.join(root, "dynamicvectorquantization_amd", "csrc", "libdvq_tuning.so")
    if not os.path.exists(tune):
        pytest.skip("libdvq_tuning.so not built (make -C dynamicvectorquantization_amd/csrc tuning)")
    r = subprocess.run([sys.executable, "-c", CODE % {"root": root}], env=dict(os.environ, DVQ_LIBRARY=tune),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PIPE_FORM_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
