// vq_assign_routed.hip -- routing prepass of the routed assign in EXACT mode.
//
// Routed assign (dvq_vq_assign_routed_{dual,triple}_f32): the routing tail of the reference encoders
// (EncoderDual.py:134-149, EncoderTriple.py:148-176) and VectorQuantize2.forward
// (quantize2_mask.py:157-191) as ONE op that never materialises h_dual / h_triple.  In DVQ_MODE_FILTER the select
// is fused into pass 1 (vq_assign_filter.hip), which derives every position's grain from the gate and writes the
// by-products itself.  DVQ_MODE_EXACT scores every position with the exact kernel, which re-derives a position's
// source from `indices`; this prepass writes them first:
//   routed_prepass_kernel   one workgroup per image: argmax of the gate (or entropy > threshold) per coarse cell
//                           -> indices [B, hc, wc], codebook_mask [B, 1, H, W], and (entropy gate) the router's
//                           int64 gate [B, hc, wc, 2].
#include "dvq_filter.h"

template <int G>
__global__ __launch_bounds__(256) void routed_prepass_kernel(
    const void *__restrict__ gate, int gate_mode, float thr, int hc, int wc,
    long long *__restrict__ indices, float *__restrict__ cmask, long long *__restrict__ gate_out)
{
    constexpr int SC = (G == 2) ? 2 : 4;
    __shared__ unsigned char grain[DVQ_ROUTE_MAX_CELLS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, ncell = hc * wc;
    const int H = SC * hc, W = SC * wc;
    for (int cell = tid; cell < ncell; cell += 256) {
        const DvqGateRaw raw = dvq_gate_fetch(gate, gate_mode, G, (size_t)b * ncell + cell);
        const int g = dvq_gate_reduce(raw, gate_mode, G, thr);
        grain[cell] = (unsigned char)g;
        indices[(size_t)b * ncell + cell] = g;
        if (gate_mode == 2 && gate_out != nullptr) {          // the router's int64 gate, a by-product
            longlong2 gg; gg.x = (raw.f[0] <= thr) ? 1 : 0; gg.y = (raw.f[0] > thr) ? 1 : 0;
            *(longlong2 *)(gate_out + 2 * ((size_t)b * ncell + cell)) = gg;
        }
    }
    __syncthreads();
    for (int i = tid; i < H * W; i += 256) {                   // codebook_mask [B, 1, H, W]
        const int y = i / W, x = i - y * W;
        const int g = grain[(y / SC) * wc + x / SC];
        cmask[(size_t)b * H * W + i] = (G == 2) ? (g == 0 ? 0.25f : 1.0f) : (g == 0 ? 0.0625f : (g == 1 ? 0.25f : 1.0f));
    }
}

int dvq_launch_routed_prepass(int G, int gate_mode, const void *gate, float thr, int B, int hc, int wc,
                              long long *indices, float *cmask, long long *gate_out, hipStream_t st)
{
    if (G == 2) hipLaunchKernelGGL(routed_prepass_kernel<2>, dim3(B), dim3(256), 0, st, gate, gate_mode, thr, hc, wc, indices, cmask, gate_out);
    else hipLaunchKernelGGL(routed_prepass_kernel<3>, dim3(B), dim3(256), 0, st, gate, gate_mode, thr, hc, wc, indices, cmask, gate_out);
    return (int)hipGetLastError();
}
