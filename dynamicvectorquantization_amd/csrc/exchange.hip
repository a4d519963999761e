// exchange.hip -- wire format of the image-parallel exchange: ONE all-gather per
// batch of a packed byte buffer per rank
//     [codes of the rank's images as int16 (num_codes <= 32768) or int32, zero padded to b_max images, then to 8 B]
//     [grain indices as int8, padded the same way]
//     [(loss numerator, element count) as 2 x float64]
// pack: one launch on the producing rank; unpack: one launch on every rank turns the gathered
// [world x bytes] buffer into int64 codes / grain indices of the GLOBAL batch and the global mean
// (the per-rank pairs are added in rank order, so every rank gets the same bits).  No allocation, no
// host round trip: the step issues pack + all_gather + unpack and nothing else.
#include "dvq_common.h"

struct XchLayout {
    long cpi, gpi;          // codes / grain indices per image
    int esize;              // 2 or 4 bytes per code on the wire
    int b_max;              // images per rank on the wire (largest shard)
    long off_grain, off_pair, nbytes;
};

__host__ __device__ inline XchLayout xch_layout(long cpi, long gpi, int b_max, int num_codes)
{
    XchLayout L;
    L.cpi = cpi; L.gpi = gpi; L.b_max = b_max;
    L.esize = (num_codes <= 32768) ? 2 : 4;
    const long nc = (long)b_max * cpi * L.esize, ng = (long)b_max * gpi;
    L.off_grain = (nc + 7) / 8 * 8;
    L.off_pair = L.off_grain + (ng + 7) / 8 * 8;
    L.nbytes = L.off_pair + 16;
    return L;
}

__global__ __launch_bounds__(256) void xch_pack_kernel(const long long *__restrict__ codes,
                                                       const long long *__restrict__ grain,
                                                       const float *__restrict__ loss, double numel,
                                                       int b_local, XchLayout L, char *__restrict__ buf)
{
    const long nc = (long)L.b_max * L.cpi, ncl = (long)b_local * L.cpi;
    const long ng = (long)L.b_max * L.gpi, ngl = (long)b_local * L.gpi;
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = t0; i < nc; i += stride) {
        const long long v = (i < ncl) ? codes[i] : 0;
        if (L.esize == 2) ((short *)buf)[i] = (short)v;
        else ((int *)buf)[i] = (int)v;
    }
    if (grain != nullptr)
        for (long i = t0; i < ng; i += stride) buf[L.off_grain + i] = (char)((i < ngl) ? grain[i] : 0);
    if (t0 == 0) {
        double *pair = (double *)(buf + L.off_pair);
        pair[0] = (loss != nullptr) ? (double)loss[0] * numel : 0.0;       // loss[0] = local mean
        pair[1] = (loss != nullptr) ? numel : 0.0;
    }
}

// images of rank r: [start, start + size): the first (global_batch % world) ranks hold one extra
__device__ __forceinline__ void xch_shard(int global_batch, int world, int r, int &start, int &size)
{
    const int base = global_batch / world, extra = global_batch - base * world;
    start = r * base + (r < extra ? r : extra);
    size = base + (r < extra ? 1 : 0);
}

__global__ __launch_bounds__(256) void xch_unpack_kernel(const char *__restrict__ gathered, int world,
                                                         int global_batch, XchLayout L,
                                                         long long *__restrict__ codes,
                                                         long long *__restrict__ grain,
                                                         float *__restrict__ mean)
{
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int base = global_batch / world, extra = global_batch - base * world;
    const long total_c = (long)global_batch * L.cpi, total_g = (long)global_batch * L.gpi;
    for (long i = t0; i < total_c; i += stride) {
        const int img = (int)(i / L.cpi);
        // rank of image `img`: the first `extra` ranks hold base + 1 images
        const int split = extra * (base + 1);
        const int r = (img < split) ? img / (base + 1) : extra + (base > 0 ? (img - split) / base : 0);
        int start, size;
        xch_shard(global_batch, world, r, start, size);
        const long local = i - (long)start * L.cpi;
        const char *src = gathered + (long)r * L.nbytes;
        codes[i] = (L.esize == 2) ? (long long)((const short *)src)[local] : (long long)((const int *)src)[local];
    }
    if (grain != nullptr)
        for (long i = t0; i < total_g; i += stride) {
            const int img = (int)(i / L.gpi);
            const int split = extra * (base + 1);
            const int r = (img < split) ? img / (base + 1) : extra + (base > 0 ? (img - split) / base : 0);
            int start, size;
            xch_shard(global_batch, world, r, start, size);
            grain[i] = (long long)(gathered + (long)r * L.nbytes + L.off_grain)[i - (long)start * L.gpi];
        }
    if (t0 == 0 && mean != nullptr) {
        double s = 0.0, n = 0.0;
        for (int r = 0; r < world; ++r) {                      // rank order: identical on every rank
            const double *pair = (const double *)(gathered + (long)r * L.nbytes + L.off_pair);
            s += pair[0];
            n += pair[1];
        }
        mean[0] = (float)(s / n);
    }
}

size_t dvq_xch_bytes(long cpi, long gpi, int b_max, int num_codes) { return (size_t)xch_layout(cpi, gpi, b_max, num_codes).nbytes; }

static int xch_grid(long items)
{
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int dvq_launch_xch_pack(const long long *codes, const long long *grain, const float *loss, double numel, int b_local,
                        int b_max, long cpi, long gpi, int num_codes, void *buf, hipStream_t st)
{
    const XchLayout L = xch_layout(cpi, gpi, b_max, num_codes);
    hipLaunchKernelGGL(xch_pack_kernel, dim3(xch_grid((long)b_max * cpi)), dim3(256), 0, st, codes, grain, loss, numel,
                       b_local, L, (char *)buf);
    return (int)hipGetLastError();
}

int dvq_launch_xch_unpack(const void *gathered, int world, int global_batch, long cpi, long gpi, int num_codes,
                          long long *codes, long long *grain, float *mean, hipStream_t st)
{
    const int b_max = (global_batch + world - 1) / world;
    const XchLayout L = xch_layout(cpi, gpi, b_max, num_codes);
    hipLaunchKernelGGL(xch_unpack_kernel, dim3(xch_grid((long)global_batch * cpi)), dim3(256), 0, st,
                       (const char *)gathered, world, global_batch, L, codes, grain, mean);
    return (int)hipGetLastError();
}
