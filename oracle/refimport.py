"""Import the upstream reference (read-only, /root/reference) in the BUILD container.

Used only by oracle/gen_golden.py and oracle/validate_against_reference.py to
pin the oracle.  /root/reference does not exist on the GPU box; nothing in
tests/, bench.py or the product package imports this module.
"""
import os
import sys
import types

REF = os.environ.get("DVQ_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "modules", "vector_quantization"))


def setup():
    """Make `import modules.…` / `import models.…` resolve to the reference.

    pytorch_lightning and torchvision are absent from the image; the stage-1
    modules only need `LightningModule` as a base class and never call
    torchvision on the encode path, so empty stand-in modules are enough to
    import them (these stubs exist only in this process)."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    sys.dont_write_bytecode = True
    import torch.nn as nn

    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = nn.Module
        pl.Callback = object
        sys.modules["pytorch_lightning"] = pl
        for sub in ("callbacks", "utilities", "utilities.distributed"):
            m = types.ModuleType("pytorch_lightning." + sub)
            m.Callback = object
            m.rank_zero_only = lambda f: f
            sys.modules["pytorch_lightning." + sub] = m
    if "torchvision" not in sys.modules:
        for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                     "torchvision.utils", "torchvision.models"):
            sys.modules[name] = types.ModuleType(name)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def quantizers():
    setup()
    from modules.vector_quantization.quantize2_mask import VectorQuantize2
    from modules.vector_quantization.quantize_vqgan import VectorQuantizer2
    return VectorQuantize2, VectorQuantizer2


def routers():
    setup()
    from modules.dynamic_modules.RouterDual import (DualGrainFeatureRouter,
                                                    DualGrainFixedEntropyRouter)
    from modules.dynamic_modules.RouterTriple import TripleGrainFeatureRouter
    return DualGrainFeatureRouter, DualGrainFixedEntropyRouter, TripleGrainFeatureRouter
