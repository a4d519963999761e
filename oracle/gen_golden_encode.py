"""Golden for the model order of the encode path, from the IMPORTED reference (BUILD container only):
the reference's own `DualGrainVQModel.encode` (models/stage1_dynamic/dqvae_dual_entropy.py:124-134) is run on CPU with the
encoder, router and quantizer instantiated from the reference's YAML (configs/stage1/dqvae-entropy-dual-r05_imagenet.yml) through
its `instantiate_from_config`, a seeded 1x1 quant_conv and a trained-like codebook, on one synthetic image.  The two encoder
branch outputs (the inputs of the hot path) are captured with forward hooks and stored with the reference's outputs:
  inputs   h_fine [1, 256, 32, 32] f32, h_coarse [1, 256, 16, 16] f32, x_entropy [1, 16, 16] f32  (captured), conv weight / bias and
           codebook regenerated from seeds (CRCs stored)
  outputs  grain indices, gate, codes, emb_loss, quant (float16 copy: the conv's summation order differs between back ends, codes
           may differ at near-ties only)
Only data is stored.  Usage: python oracle/gen_golden_encode.py
"""
import json
import os
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main(B=1):
    refimport.setup()
    # modules/dynamic_modules/utils.py (drawing helpers, not on the encode path) builds a torchvision transform at import time:
    # two more attributes on the in-memory torchvision stand-in of this process let it import
    tv = sys.modules["torchvision.transforms"]
    tv.Compose = lambda ts: None
    tv.ToPILImage = lambda *a, **k: None
    tv.ToTensor = lambda *a, **k: None
    import yaml
    from utils.utils import instantiate_from_config
    from models.stage1_dynamic.dqvae_dual_entropy import DualGrainVQModel, Entropy
    cfg = yaml.safe_load(open(os.path.join(refimport.REF, "configs/stage1/dqvae-entropy-dual-r05_imagenet.yml")))["model"]["params"]
    torch.manual_seed(20260301)
    torch.set_grad_enabled(False)
    cwd = os.getcwd()
    os.chdir(refimport.REF)                                   # the router's json_path is relative to the reference root
    try:
        encoder = instantiate_from_config(cfg["encoderconfig"]).eval()
        quantize = instantiate_from_config(cfg["vqconfig"]).eval()
    finally:
        os.chdir(cwd)
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    quantize.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    conv = torch.nn.Conv2d(cfg["quant_before_dim"], cfg["quant_after_dim"], 1).eval()
    cw = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0)
    cb = synth.normal(9502, (D,), 0.0, 0.1)
    conv.weight.data.copy_(torch.from_numpy(cw))
    conv.bias.data.copy_(torch.from_numpy(cb))
    model = types.SimpleNamespace(encoder=encoder, quantize=quantize, quant_conv=conv,
                                  quant_sample_temperature=cfg["quant_sample_temperature"],
                                  entropy_calculation=Entropy(16, 256, 256).eval())
    img, noisy = synth.images_flat_noise(9503, B)
    cap = {}
    h1 = encoder.conv_out_fine.register_forward_hook(lambda m, i, o: cap.__setitem__("h_fine", o.detach().clone()))
    h2 = encoder.conv_out_coarse.register_forward_hook(lambda m, i, o: cap.__setitem__("h_coarse", o.detach().clone()))
    # the trunk's output is a random network's: scale it to the codebook's magnitude so that tokens and codes interact
    # (done by scaling the two output convs, i.e. still the reference's forward)
    quant, emb_loss, info, grain, gate, x_entropy = DualGrainVQModel.encode(model, torch.from_numpy(img))
    h1.remove(); h2.remove()
    codes = info[2]
    meta = json.dumps(dict(torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
                           reference="Corleone-Huang/DynamicVectorQuantization @ /root/reference: DualGrainVQModel.encode on CPU",
                           yaml="configs/stage1/dqvae-entropy-dual-r05_imagenet.yml", seeds=dict(conv_w=9501, conv_b=9502, image=9503),
                           images="synth.images_flat_noise(9503, %d)[0]: the test regenerates the PIXELS and runs them through the "
                                  "Entropy kernel" % B, image_crc=int(crc(img))))
    out = os.path.join(ROOT, "tests", "golden", "encode_dual_entropy_model_B%d.npz" % B)
    np.savez_compressed(out, meta=np.array(meta), h_fine=cap["h_fine"].numpy(), h_coarse=cap["h_coarse"].numpy(),
                        x_entropy=x_entropy.numpy().astype(np.float32), conv_w_crc=crc(cw), conv_b_crc=crc(cb), cb_crc=crc(E),
                        grain=grain.numpy().astype(np.int8), gate=gate.numpy().astype(np.int8), codes=codes.numpy().astype(np.int16),
                        emb_loss=np.float32(float(emb_loss)), quant_f16=quant.numpy().astype(np.float16),
                        fine_ratio=np.float32(float(grain.float().mean())))
    print("wrote", out, "%.1f KiB" % (os.path.getsize(out) / 1024), "fine ratio", float(grain.float().mean()),
          "loss", float(emb_loss), "|h_fine| max", float(cap["h_fine"].abs().max()), "codes used", int(codes.unique().numel()))


def feature_model(kind, B=1):
    """the feature-router models: DualGrainVQModel (dqvae_dual_feat.py:59-68) / TripleGrainVQModel (dqvae_triple_feat.py:68-77) with
    the encoder + feature router + quantizer of the reference's YAMLs; the router's MLP gets seeded weights (regenerated in the
    test, CRCs stored) so that nothing but the captured branch features has to be stored"""
    refimport.setup()
    tv = sys.modules["torchvision.transforms"]
    tv.Compose = lambda ts: None
    tv.ToPILImage = lambda *a, **k: None
    tv.ToTensor = lambda *a, **k: None
    import yaml
    from utils.utils import instantiate_from_config
    if kind == "dual":
        from models.stage1_dynamic.dqvae_dual_feat import DualGrainVQModel as Model
        yml, G = "configs/stage1/dqvae-dual-r-05_imagenet.yml", 2
    else:
        from models.stage1_dynamic.dqvae_triple_feat import TripleGrainVQModel as Model
        yml, G = "configs/stage1/dqvae-triple-r-03-03_imagenet.yml", 3
    cfg = yaml.safe_load(open(os.path.join(refimport.REF, yml)))["model"]["params"]
    torch.manual_seed(20260302 + G)
    torch.set_grad_enabled(False)
    cwd = os.getcwd()
    os.chdir(refimport.REF)
    try:
        encoder = instantiate_from_config(cfg["encoderconfig"]).eval()
        quantize = instantiate_from_config(cfg["vqconfig"]).eval()
    finally:
        os.chdir(cwd)
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    quantize.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    F = G * D
    w1, b1 = synth.normal(9600 + G, (F, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9610 + G, (F,), 0.0, 0.1)
    w2, b2 = synth.normal(9620 + G, (G, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9630 + G, (G,), 0.0, 0.1)
    gate_mlp = encoder.router.gate
    gate_mlp[0].weight.data.copy_(torch.from_numpy(w1)); gate_mlp[0].bias.data.copy_(torch.from_numpy(b1))
    gate_mlp[2].weight.data.copy_(torch.from_numpy(w2)); gate_mlp[2].bias.data.copy_(torch.from_numpy(b2))
    conv = torch.nn.Conv2d(cfg["quant_before_dim"], cfg["quant_after_dim"], 1).eval()
    cw = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0)
    cb = synth.normal(9502, (D,), 0.0, 0.1)
    conv.weight.data.copy_(torch.from_numpy(cw)); conv.bias.data.copy_(torch.from_numpy(cb))
    model = types.SimpleNamespace(encoder=encoder, quantize=quantize, quant_conv=conv,
                                  quant_sample_temperature=cfg["quant_sample_temperature"])
    img, _ = synth.images_flat_noise(9640 + G, B)
    cap = {}
    hooks = [getattr(encoder, "conv_out_" + n).register_forward_hook(lambda m, i, o, n=n: cap.__setitem__("h_" + n, o.detach().clone()))
             for n in (("fine", "coarse") if G == 2 else ("fine", "median", "coarse"))]
    quant, emb_loss, info, grain, gate = Model.encode(model, torch.from_numpy(img))
    for h in hooks:
        h.remove()
    codes = info[2]
    # margin of the gate's argmax: the test asks for equal grain maps only where it exceeds the logits' tolerance
    top2 = torch.topk(gate, 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    meta = json.dumps(dict(torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(), yaml=yml,
                           reference="Corleone-Huang/DynamicVectorQuantization @ /root/reference: %s.encode on CPU" % Model.__name__,
                           seeds=dict(conv_w=9501, conv_b=9502, w1=9600 + G, b1=9610 + G, w2=9620 + G, b2=9630 + G, image=9640 + G)))
    out = os.path.join(ROOT, "tests", "golden", "encode_%s_feature_model_B%d.npz" % (kind, B))
    np.savez_compressed(out, meta=np.array(meta), **{k: v.numpy() for k, v in cap.items()},
                        conv_w_crc=crc(cw), conv_b_crc=crc(cb), cb_crc=crc(E), w1_crc=crc(w1), b1_crc=crc(b1), w2_crc=crc(w2), b2_crc=crc(b2),
                        gate=gate.numpy().astype(np.float32), grain=grain.numpy().astype(np.int8), gate_margin_min=np.float32(margin.min()),
                        codes=codes.numpy().astype(np.int16), emb_loss=np.float32(float(emb_loss)), quant_f16=quant.numpy().astype(np.float16))
    print("wrote", out, "%.1f KiB" % (os.path.getsize(out) / 1024), "grain histogram", np.bincount(grain.numpy().reshape(-1), minlength=G),
          "min gate margin", float(margin.min()), "loss", float(emb_loss), "codes used", int(codes.unique().numel()))


if __name__ == "__main__":
    # usage: gen_golden_encode.py [entropy|dual|triple ...] [B=<batch>]   (round 3: B = 1 each; round 4: entropy B=4, dual / triple B=2)
    args = [a for a in sys.argv[1:] if not a.startswith("B=")]
    Bs = [int(a[2:]) for a in sys.argv[1:] if a.startswith("B=")]
    B = Bs[0] if Bs else 1
    which = args or ["entropy", "dual", "triple"]
    if "entropy" in which:
        main(B)
    for kind in ("dual", "triple"):
        if kind in which:
            feature_model(kind, B)
