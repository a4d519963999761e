"""MI355X-native DQ-VAE vector-quantization hot path (drop-in for the reference's
modules/vector_quantization + modules/dynamic_modules router path).

Submodules:
  quantize  VectorQuantize2 / VQEmbedding / VectorQuantizer2  (HIP: dvq_vq_assign_nchw_f32)
  router    routers + route_select_dual/triple + entropy_gate (HIP: dvq_route_select_*, dvq_entropy_gate_f32)
  encode    encode glue + image-parallel all-gather of codes
  synth     deterministic synthetic inputs (numpy only)
  _lib      ctypes binding of csrc/libdvq.so -- raises if the library is missing (no CPU fallback)
`quantize`, `router` and `encode` import `_lib`; `synth` does not, so host-only tooling can use it.
"""
__version__ = "0.1.0"
