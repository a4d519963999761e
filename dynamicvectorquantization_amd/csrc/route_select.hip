// route_select.hip -- dual / triple granularity router select ("scatter/gather") for gfx950.
//
// Replaces the routing tails of the reference encoders in eval mode:
//   modules/dynamic_modules/EncoderDual.py:134-149   (argmax, x2 repeat_interleave, where, mask)
//   modules/dynamic_modules/EncoderTriple.py:148-176 (argmax, x4/x2 repeat_interleave, 3 wheres, mask)
//   modules/dynamic_modules/RouterDual.py:53-57      (entropy threshold gate)
// One pass: every output float4 reads exactly one source (the branch that won its cell), so HBM
// traffic is the algorithmic 1 read + 1 write per element; the reference materialises two
// upsampled copies and reads all branches.  Pure data movement -> HBM-bound: each thread moves
// 16 B, consecutive lanes consecutive addresses; LDS only holds the image's grain map (1 B / cell).
#include "dvq_common.h"

// MODE 0: f32 gate logits, 1: int64 gate, 2: f32 entropy map + threshold (the fixed-entropy router fused in:
// gate = [(ent <= thr), (ent > thr)], whose argmax is (ent > thr); NaN compares false twice -> 0)
template <int G, int MODE>
__device__ __forceinline__ int gate_argmax(const void *gate, size_t cell, float thr)
{
    // torch.argmax semantics: first maximal value wins, NaN counts as the maximum
    if (MODE == 2) {
        return (((const float *)gate)[cell] > thr) ? 1 : 0;
    } else if (MODE == 1) {
        const long long *g = (const long long *)gate + cell * G;
        long long best = g[0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < G; ++i) {
            long long v = g[i];
            if (v > best) { best = v; bi = i; }
        }
        return bi;
    } else {
        const float *g = (const float *)gate + cell * G;
        float best = g[0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < G; ++i) {
            float v = g[i];
            if ((v > best) || (v != v && best == best)) { best = v; bi = i; }
        }
        return bi;
    }
}

// G = 2: dual (fine grid = 2x coarse); G = 3: triple (fine grid = 4x coarse, median 2x).
// One workgroup = one image b and a run of PLANES_PER_BLOCK planes p in [0, C]: planes 0..C-1 are
// the feature channels, plane C is the codebook_mask plane (which also emits `indices`).  The
// image's grain map (argmax per coarse cell) is evaluated once per workgroup into LDS; after that
// every thread streams float4s: one 16-B load from the branch that won the cell, one 16-B store.
constexpr int PLANES_PER_BLOCK = 8;
constexpr int MAX_CELLS = 4096;                 // grain map bytes kept in LDS (64 x 64 coarse cells)

// V = floats per thread and access: 4 (rows are whole float4s), or 2 for the dual select on an odd coarse width (rows of 2 wc floats
// are then only 8-byte pieces; both floats of a piece lie in one coarse cell)
template <int V> struct VecOf;
template <> struct VecOf<4> { typedef f32x4 type; };
template <> struct VecOf<2> { typedef f32x2 type; };
template <int G, int MODE, int V = 4>
__global__ __launch_bounds__(256) void route_select_kernel(
    const void *__restrict__ gate, const float *__restrict__ h_coarse,
    const float *__restrict__ h_median, const float *__restrict__ h_fine,
    int B, int C, int hc, int wc,
    float *__restrict__ h_out, long long *__restrict__ indices, float *__restrict__ cmask,
    float thr, long long *__restrict__ gate_out)
{
    constexpr int SC = (G == 2) ? 2 : 4;          // fine pixels per coarse cell edge
    __shared__ unsigned char grain[MAX_CELLS];
    typedef typename VecOf<V>::type vec_t;
    const int H = SC * hc, W = SC * wc, W4 = W / V;
    const int per_plane = H * W4;
    const int ncell = hc * wc;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const int p0 = blockIdx.x * PLANES_PER_BLOCK;
    const int p1 = (p0 + PLANES_PER_BLOCK < C + 1) ? p0 + PLANES_PER_BLOCK : C + 1;
    const bool in_lds = ncell <= MAX_CELLS;
    if (in_lds) {
        for (int cell = threadIdx.x; cell < ncell; cell += 256)
            grain[cell] = (unsigned char)gate_argmax<G, MODE>(gate, (size_t)b * ncell + cell, thr);
        __syncthreads();
    }
    auto grain_of = [&](int cell) -> int {
        return in_lds ? (int)grain[cell] : gate_argmax<G, MODE>(gate, (size_t)b * ncell + cell, thr);
    };
    for (int p = p0; p < p1; ++p) {
        const size_t plane = (size_t)b * C + p;
        for (int i = threadIdx.x; i < per_plane; i += 256) {
            const int y = i / W4, x = (i - y * W4) * V;
            const int cy = y / SC;
            const int cx0 = x / SC, cx1 = (x + V - 1) / SC;   // the (up to two) coarse cells of this piece
            const int cell0 = cy * wc + cx0;
            const int g0 = grain_of(cell0);
            const int g1 = (cx1 != cx0) ? grain_of(cell0 + 1) : g0;
            if (p == C) {
                vec_t m;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    int g = ((x + j) / SC == cx0) ? g0 : g1;
                    m[j] = (G == 2) ? (g == 0 ? 0.25f : 1.0f)
                                    : (g == 0 ? 0.0625f : (g == 1 ? 0.25f : 1.0f));
                }
                *(vec_t *)(cmask + ((size_t)b * H + y) * W + x) = m;
                if (y % SC == 0) {
                    if (x % SC == 0) indices[(size_t)b * ncell + cell0] = g0;
                    if (cx1 != cx0) indices[(size_t)b * ncell + cell0 + 1] = g1;
                    if (MODE == 2 && gate_out != nullptr) {        // the router's int64 gate, a by-product
                        const float *ent = (const float *)gate + (size_t)b * ncell;
                        if (x % SC == 0) {
                            const float e = ent[cell0];
                            longlong2 gg; gg.x = (e <= thr) ? 1 : 0; gg.y = (e > thr) ? 1 : 0;
                            *(longlong2 *)(gate_out + 2 * ((size_t)b * ncell + cell0)) = gg;
                        }
                        if (cx1 != cx0) {
                            const float e = ent[cell0 + 1];
                            longlong2 gg; gg.x = (e <= thr) ? 1 : 0; gg.y = (e > thr) ? 1 : 0;
                            *(longlong2 *)(gate_out + 2 * ((size_t)b * ncell + cell0 + 1)) = gg;
                        }
                    }
                }
                continue;
            }
            const size_t o = (plane * H + y) * W + x;
            vec_t v;
            if (g0 == G - 1 && g1 == G - 1) {
                v = *(const vec_t *)(h_fine + o);
            } else {
                vec_t f = {};
                if (g0 == G - 1 || g1 == G - 1) f = *(const vec_t *)(h_fine + o);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int xx = x + j;
                    const int g = (xx / SC == cx0) ? g0 : g1;
                    float sv;
                    if (g == 0)
                        sv = h_coarse[(plane * hc + cy) * wc + xx / SC];
                    else if (G == 3 && g == 1)
                        sv = h_median[(plane * (2 * hc) + y / 2) * (2 * wc) + xx / 2];
                    else
                        sv = f[j];
                    v[j] = sv;
                }
            }
            *(vec_t *)(h_out + o) = v;
        }
    }
    if (in_lds) __syncthreads();                  // the next image overwrites the grain map
    }
}

__global__ __launch_bounds__(256) void entropy_gate_kernel(const float *__restrict__ ent, long n,
                                                           float thr, long long *__restrict__ gate)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float e = ent[i];
        longlong2 g;
        g.x = (e <= thr) ? 1 : 0;
        g.y = (e > thr) ? 1 : 0;
        *(longlong2 *)(gate + 2 * i) = g;
    }
}

// out[n, :] = E[idx[n], :], D % 4 == 0; invalid index -> NaN row
__global__ __launch_bounds__(256) void embed_gather_kernel(const float *__restrict__ E, int K, int D,
                                                           const long long *__restrict__ idx, long n,
                                                           float *__restrict__ out)
{
    const int D4 = D / 4;
    const size_t total = (size_t)n * D4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t row = i / D4;
        int q = (int)(i - row * D4);
        long long j = idx[row];
        f32x4 v;
        if (j >= 0 && j < K) {
            v = *(const f32x4 *)(E + (size_t)j * D + 4 * q);
        } else {
            float nanv = __builtin_nanf("");
            v = (f32x4){nanv, nanv, nanv, nanv};
        }
        *(f32x4 *)(out + row * D + 4 * q) = v;
    }
}

static int grid_for(size_t items)
{
    size_t blocks = (items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;     // 256 CUs x 16 resident 256-thread blocks
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int dvq_launch_route_select(int G, int gate_mode, const void *gate, const float *h_coarse,
                            const float *h_median, const float *h_fine, int B, int C, int hc, int wc,
                            float *h_out, long long *indices, float *cmask, float thr, long long *gate_out,
                            hipStream_t st)
{
    dim3 grid((C + 1 + PLANES_PER_BLOCK - 1) / PLANES_PER_BLOCK, B < 65535 ? B : 65535), block(256);
#define DVQ_SEL(GG, MM) hipLaunchKernelGGL((route_select_kernel<GG, MM>), grid, block, 0, st, gate, h_coarse, h_median, h_fine, B, C, hc, wc, h_out, indices, cmask, thr, gate_out)
#define DVQ_SEL2(MM) hipLaunchKernelGGL((route_select_kernel<2, MM, 2>), grid, block, 0, st, gate, h_coarse, h_median, h_fine, B, C, hc, wc, h_out, indices, cmask, thr, gate_out)
    if (G == 2 && (wc & 1)) {                                // odd coarse width: 8-byte pieces
        if (gate_mode == 2) DVQ_SEL2(2);
        else if (gate_mode == 1) DVQ_SEL2(1);
        else DVQ_SEL2(0);
    } else
    if (G == 2 && gate_mode == 2) DVQ_SEL(2, 2);
    else if (G == 2 && gate_mode == 1) DVQ_SEL(2, 1);
    else if (G == 2) DVQ_SEL(2, 0);
    else if (gate_mode == 1) DVQ_SEL(3, 1);
    else DVQ_SEL(3, 0);
#undef DVQ_SEL
#undef DVQ_SEL2
    return (int)hipGetLastError();
}

int dvq_launch_entropy_gate(const float *ent, long n, float thr, long long *gate, hipStream_t st)
{
    hipLaunchKernelGGL(entropy_gate_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, ent, n, thr, gate);
    return (int)hipGetLastError();
}

int dvq_launch_embed_gather(const float *E, int K, int D, const long long *idx, long n, float *out,
                            hipStream_t st)
{
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for((size_t)n * (D / 4))), dim3(256), 0, st, E, K, D, idx, n, out);
    return (int)hipGetLastError();
}
