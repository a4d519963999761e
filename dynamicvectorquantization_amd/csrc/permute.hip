// permute.hip -- dense codes + grain map <-> variable-length coarse / fine code streams (gfx950).
//
// Replaces DualGrainSeperatePermuter (reference modules/dynamic_modules/permuter.py):
//   forward      :50-109  per-image Python list comprehensions with boolean masking, .cpu() hops of
//                         the position tables and pad_sequence
//   forward_back :111-135 a triple-nested Python loop, one tensor element per iteration
// Here: one workgroup per image.  forward = a segmented stream compaction (ballot/popcount scan of
// the grain map; the exclusive ranks stay in LDS so "row-first" order needs no second pass), EOS and
// PAD written by the same kernel.  forward_back keeps the reference's sequential semantics (a later
// sequence entry overwrites an earlier one at the same position, entries after EOS are ignored)
// with an LDS "last writer" table built by atomicMax.  Pure integer work: bit-exact.
#include "dvq_common.h"

constexpr int PERM_MAX_CELLS = 1024;      // coarse cells per image (reference: 16 x 16 = 256)

__device__ __forceinline__ int block_excl_scan_256(int flag, int *wave_tot, int &total)
{
    // exclusive prefix count of `flag` over the 256 threads of the block (4 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return off + in_wave;
}

// counts[b] = (#coarse cells, #fine cells); maxes = batch maxima (must be zero on entry)
__global__ __launch_bounds__(256) void permute_count_kernel(const long long *__restrict__ grain, int ncell,
                                                            int *__restrict__ counts, int *__restrict__ maxes)
{
    __shared__ int tot[2];
    if (threadIdx.x < 2) tot[threadIdx.x] = 0;
    __syncthreads();
    const long long *g = grain + (size_t)blockIdx.x * ncell;
    int c = 0, f = 0;
    for (int i = threadIdx.x; i < ncell; i += 256) {
        long long v = g[i];
        c += (v == 0);
        f += (v == 1);
    }
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off); f += __shfl_xor(f, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&tot[0], c); atomicAdd(&tot[1], f); }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[2 * blockIdx.x] = tot[0];
        counts[2 * blockIdx.x + 1] = tot[1];
        atomicMax(&maxes[0], tot[0]);
        atomicMax(&maxes[1], tot[1]);
    }
}

struct PermSpecial { long long content_pad, content_eos, cpos_pad, cpos_eos, fpos_pad, fpos_eos; };

__global__ __launch_bounds__(256) void permute_forward_kernel(
    const long long *__restrict__ codes, const long long *__restrict__ grain, int hc, int wc, int row_first,
    int Lc, int Lf, PermSpecial sp,
    long long *__restrict__ cc, long long *__restrict__ cp, long long *__restrict__ cs,
    long long *__restrict__ fc, long long *__restrict__ fp, long long *__restrict__ fs)
{
    __shared__ int rank_f[PERM_MAX_CELLS + 1];          // exclusive rank of each cell among the fine cells
    __shared__ int wave_tot[4];
    const int b = blockIdx.x, ncell = hc * wc, W = 2 * wc;
    const long long *g = grain + (size_t)b * ncell;
    const long long *cd = codes + (size_t)b * 4 * ncell;
    long long *occ = cc + (size_t)b * Lc, *ocp = cp + (size_t)b * Lc, *ocs = cs + (size_t)b * Lc;
    long long *ofc = fc + (size_t)b * Lf, *ofp = fp + (size_t)b * Lf, *ofs = fs + (size_t)b * Lf;
    int base_c = 0, base_f = 0;
    for (int c0 = 0; c0 < ncell; c0 += 256) {
        const int ci = c0 + threadIdx.x;
        const long long v = (ci < ncell) ? g[ci] : -1;
        int tc, tf;
        const int rc = block_excl_scan_256(v == 0, wave_tot, tc) + base_c;
        const int rf = block_excl_scan_256(v == 1, wave_tot, tf) + base_f;
        if (ci < ncell) {
            rank_f[ci] = rf;
            const int cy = ci / wc, cx = ci - cy * wc;
            if (v == 0 && rc < Lc) {                       // coarse: the (h2, w2) = (0, 0) code of the cell
                occ[rc] = cd[(size_t)(2 * cy) * W + 2 * cx];
                ocp[rc] = ci;
                ocs[rc] = 0;
            }
            if (v == 1 && !row_first) {                   // region-first: the 4 codes of the cell together
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 2 * cy + (q >> 1), c = 2 * cx + (q & 1);
                    const int k = 4 * rf + q;
                    if (k < Lf) { ofc[k] = cd[(size_t)r * W + c]; ofp[k] = (long long)r * W + c; ofs[k] = 1; }
                }
            }
        }
        base_c += tc;
        base_f += tf;
    }
    if (threadIdx.x == 0) rank_f[ncell] = base_f;
    __syncthreads();
    if (row_first) {                                      // fine pixels in row-major order
        for (int ci = threadIdx.x; ci < ncell; ci += 256) {
            if (g[ci] != 1) continue;
            const int cy = ci / wc, cx = ci - cy * wc;
            const int row0 = rank_f[cy * wc], rowcnt = rank_f[(cy + 1) * wc] - row0, within = rank_f[ci] - row0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dy = q >> 1, dx = q & 1;
                const int r = 2 * cy + dy, c = 2 * cx + dx;
                const int k = 4 * row0 + dy * 2 * rowcnt + 2 * within + dx;
                if (k < Lf) { ofc[k] = cd[(size_t)r * W + c]; ofp[k] = (long long)r * W + c; ofs[k] = 1; }
            }
        }
    }
    // EOS, then PAD up to the batch-wide length (pad_sequence)
    for (int k = base_c + threadIdx.x; k < Lc; k += 256) {
        occ[k] = (k == base_c) ? sp.content_eos : sp.content_pad;
        ocp[k] = (k == base_c) ? sp.cpos_eos : sp.cpos_pad;
        ocs[k] = 0;
    }
    for (int k = 4 * base_f + threadIdx.x; k < Lf; k += 256) {
        ofc[k] = (k == 4 * base_f) ? sp.content_eos : sp.content_pad;
        ofp[k] = (k == 4 * base_f) ? sp.fpos_eos : sp.fpos_pad;
        ofs[k] = 1;
    }
}

__global__ __launch_bounds__(256) void permute_backward_kernel(
    const long long *__restrict__ cc, const long long *__restrict__ fc,
    const long long *__restrict__ cp, const long long *__restrict__ fp,
    int Lc, int Lf, int hc, int wc, long long cpos_eos, long long fpos_eos, long long *__restrict__ target)
{
    __shared__ int win_c[PERM_MAX_CELLS];               // last sequence entry that wrote each coarse cell
    __shared__ int win_f[4 * PERM_MAX_CELLS];           // ... each fine position
    __shared__ int eos[2];
    const int b = blockIdx.x, ncell = hc * wc, nfine = 4 * ncell, W = 2 * wc;
    const long long *icc = cc + (size_t)b * Lc, *icp = cp + (size_t)b * Lc;
    const long long *ifc = fc + (size_t)b * Lf, *ifp = fp + (size_t)b * Lf;
    for (int i = threadIdx.x; i < ncell; i += 256) win_c[i] = -1;
    for (int i = threadIdx.x; i < nfine; i += 256) win_f[i] = -1;
    if (threadIdx.x == 0) { eos[0] = Lc; eos[1] = Lf; }
    __syncthreads();
    for (int k = threadIdx.x; k < Lc; k += 256) if (icp[k] == cpos_eos) atomicMin(&eos[0], k);
    for (int k = threadIdx.x; k < Lf; k += 256) if (ifp[k] == fpos_eos) atomicMin(&eos[1], k);
    __syncthreads();
    const int ec = eos[0], ef = eos[1];
    const bool coarse_applied = ec < Lc;                 // the upsample happens when EOS is met (:122-125)
    for (int k = threadIdx.x; k < ec; k += 256) {
        long long p = icp[k];
        if (p >= 0 && p < ncell) atomicMax(&win_c[(int)p], k);
    }
    for (int k = threadIdx.x; k < ef; k += 256) {
        long long p = ifp[k];
        if (p >= 0 && p < nfine) atomicMax(&win_f[(int)p], k);
    }
    __syncthreads();
    long long *out = target + (size_t)b * nfine;
    for (int i = threadIdx.x; i < nfine; i += 256) {
        const int r = i / W, c = i - r * W;
        long long v = 0;
        const int wf = win_f[i];
        if (wf >= 0) {
            v = ifc[wf];
        } else if (coarse_applied) {
            const int wc_ = win_c[(r >> 1) * wc + (c >> 1)];
            if (wc_ >= 0) v = icc[wc_];
        }
        out[i] = v;
    }
}

__global__ void permute_zero_maxes_kernel(int *__restrict__ maxes)
{
    if (threadIdx.x < 2) maxes[threadIdx.x] = 0;
}

int dvq_launch_permute_count(const long long *grain, int B, int ncell, int *counts, int *maxes, hipStream_t st)
{
    // zeroed by a kernel, not hipMemsetAsync: memset nodes misbehave under hipGraph replay on ROCm 7.2
    hipLaunchKernelGGL(permute_zero_maxes_kernel, dim3(1), dim3(64), 0, st, maxes);
    hipLaunchKernelGGL(permute_count_kernel, dim3(B), dim3(256), 0, st, grain, ncell, counts, maxes);
    return (int)hipGetLastError();
}

int dvq_launch_permute_forward(const long long *codes, const long long *grain, int B, int hc, int wc,
                               int row_first, int Lc, int Lf, const long long *special,
                               long long *cc, long long *cp, long long *cs, long long *fc, long long *fp,
                               long long *fs, hipStream_t st)
{
    PermSpecial sp = {special[0], special[1], special[2], special[3], special[4], special[5]};
    hipLaunchKernelGGL(permute_forward_kernel, dim3(B), dim3(256), 0, st, codes, grain, hc, wc, row_first, Lc, Lf,
                       sp, cc, cp, cs, fc, fp, fs);
    return (int)hipGetLastError();
}

int dvq_launch_permute_backward(const long long *cc, const long long *fc, const long long *cp,
                                const long long *fp, int B, int Lc, int Lf, int hc, int wc,
                                long long cpos_eos, long long fpos_eos, long long *target, hipStream_t st)
{
    hipLaunchKernelGGL(permute_backward_kernel, dim3(B), dim3(256), 0, st, cc, fc, cp, fp, Lc, Lf, hc, wc,
                       cpos_eos, fpos_eos, target);
    return (int)hipGetLastError();
}

int dvq_permute_max_cells(void) { return PERM_MAX_CELLS; }
