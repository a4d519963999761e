"""CRC of every section of the codebook prep buffer (dvq_codebook_prepare_f32) over a set of codebooks: run once per library
(DVQ_LIBRARY=...), diff the JSON lines.  The 32 padding floats of each f32 tile are masked (scratch of the build)."""
import os, sys, json, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, synth
dev = torch.device("cuda:0")
out = []
def case(name, E):
    K, D = E.shape
    Et = torch.from_numpy(np.ascontiguousarray(E, dtype=np.float32)).to(dev)
    nb = _lib.lib.dvq_codebook_prep_bytes(K, D)
    buf = torch.full((nb,), 0xA5, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.dvq_codebook_prepare_f32(Et.data_ptr(), K, D, buf.data_ptr(), nb, _lib.stream_ptr(dev)), "prep")
    torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    T = (K + 31) // 32
    tf = 32 * D + 64
    tiles = raw[:T * tf * 4].view(np.float32).reshape(T, tf).copy()
    tiles[:, 32 * D + 32:] = 0
    en_off = T * tf * 4
    en = raw[en_off:en_off + T * 128]
    f16 = (en_off + T * 128 + 255) // 256 * 256
    meta = raw[f16:f16 + 24]
    tb = D * 64 + 256
    img8 = raw[f16 + 256:f16 + 256 + T * tb]
    o16 = (T * tb + 255) // 256 * 256
    img16 = raw[f16 + 256 + o16:f16 + 256 + o16 + T * tb]
    c = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    out.append({"case": name, "K": K, "D": D, "tiles": c(tiles), "en": c(en), "meta": meta.view(np.int32).tolist(), "img8": c(img8), "img16": c(img16)})
rng = np.random.default_rng(3)
case("trained 1024x256", synth.codebook_trained(1024, 256))
case("trained 1000x256", synth.codebook_trained(1000, 256))
case("normal 33x64", rng.standard_normal((33, 64)) * 1e-3)
case("normal 512x128", rng.standard_normal((512, 128)) * 40.0)
case("normal 8200x64", rng.standard_normal((8200, 64)))
case("trained 16384x256", synth.codebook_trained(16384, 256))
e = rng.standard_normal((100, 256)); e[57, 13] = np.nan; case("nan 100x256", e)
e = rng.standard_normal((100, 256)); e[99, 255] = np.inf; case("inf 100x256", e)
case("zeros 64x128", np.zeros((64, 128)))
e = rng.standard_normal((40, 64)) * 1e-30; case("tiny 40x64", e)
e = rng.standard_normal((40, 64)) * 1e30; case("huge 40x64 (norm overflows)", e)
print(json.dumps({"lib": os.environ.get("DVQ_LIBRARY", "product"), "cases": out}))
