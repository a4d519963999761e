// dvq_common.h -- shared device helpers and the codebook "prep" buffer layout (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DVQ_CODE_TILE 32  // codes per MFMA tile (rows of a 32x32 MFMA)

// ---------------------------------------------------------------------------------------------
// Prep buffer (built once per codebook by dvq_codebook_prepare_f32), all offsets in bytes,
// T = ceil(K/32) code tiles:
//   [f32 tile images]  T x (32*D + 64) floats.  Image of tile t:
//        img[kg][c][p], kg < D/8, c < 32, p < 8  =  E[32t + c][8kg + 2(p&3) + (p>>2)]
//        (lane (c, h) of a wave reads its 4 next f32-MFMA A operands k = 8kg + {0,2,4,6} + h with
//         one ds_read_b128 at float offset (kg*32 + c)*8 + 4h), then en[32] (exact ATen-order
//        squared norms of the tile's codes; 0 for padded codes) and 32 floats of padding (no kernel reads them; the build parks
//        its partial maxima there: vq_assign_exact.hip, codebook_prep_f32_kernel).  tests/test_codebook_prep.py restates every section.
//   [en]               Kpad floats (Kpad = 32T): all squared norms, same values.
//   [f16 section]      used by the filter kernel, see vq_assign_filter.hip.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int dvq_num_tiles(int K) { return (K + DVQ_CODE_TILE - 1) / DVQ_CODE_TILE; }
__host__ __device__ inline size_t dvq_tile_floats(int D) { return (size_t)32 * D + 64; }
__host__ __device__ inline size_t dvq_prep_f32_tiles_bytes(int K, int D)
{
    return (size_t)dvq_num_tiles(K) * dvq_tile_floats(D) * sizeof(float);
}
__host__ __device__ inline size_t dvq_prep_en_offset(int K, int D) { return dvq_prep_f32_tiles_bytes(K, D); }
__host__ __device__ inline size_t dvq_prep_f16_offset(int K, int D)
{
    return dvq_prep_en_offset(K, D) + (size_t)dvq_num_tiles(K) * 32 * sizeof(float);
}

// XCD-aware block remap (bijective for any grid size): consecutive tiles go to ONE XCD, so the
// adjacent pieces of a DRAM row / the shared operands are requested close together in time by
// one L2.  blockIdx % 8 labels the blocks that share an XCD (dispatch is round-robin); this is a
// speed heuristic only -- any placement gives the same results.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc_lane,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// the same with an instruction offset OFF (bytes, < 4096): it is added to BOTH addresses -- global gsrc_lane + OFF, LDS
// lds_wave_base + OFF + lane*16 (measured: tools/micro/glds_offset.hip) -- so the pieces of one contiguous copy share one
// address register pair and one M0 value
template <int OFF>
__device__ __forceinline__ void glds16_off(const void *gsrc_lane, void *lds_wave_base)
{
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed instruction offset");
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc_lane,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, OFF, 0);
}
__device__ __forceinline__ void glds4(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc_lane,
        (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}

// ATen-order squared norm pieces (see oracle/dvq_oracle.c): explicit roundings, never contracted.
__device__ __forceinline__ float sq_rn(float x) { return __fmul_rn(x, x); }

// torch CPU argmin step over candidates visited in ascending index:
// take if d < best, or d is NaN and best is not.
__device__ __forceinline__ bool argmin_take(float d, float best)
{
    return !(d >= best) && (best == best);
}

// merge two (distance, index) candidates of disjoint ascending scans
__device__ __forceinline__ void argmin_merge(float &d, int &i, float d2, int i2)
{
    bool n1 = d != d, n2 = d2 != d2;
    bool other;
    if (n1 || n2)
        other = n2 && (!n1 || i2 < i);
    else
        other = (d2 < d) || (d2 == d && i2 < i);
    if (other) { d = d2; i = i2; }
}

// Loss finalize fused into the last kernel of the filter path (vq_assign_exact_kernel in list
// mode): the block that draws the last ticket sums partials[0 .. nparts) and writes loss[0..1].
struct DvqLossTail {
    float *loss;              // nullptr = no fused loss finalize
    const double *partials;
    int *ticket;              // zeroed at the start of the op
    int nparts;
    double inv_numel;
    float beta;
    int *counters;            // nullptr = no queue bookkeeping; else counters[0] = total queued tokens
    int shard_cap;
};
#define DVQ_QSHARDS 64        // the pass-1 -> resolver queue is sharded this many ways (power of two)
#define DVQ_QCOUNT0 8         // counters[DVQ_QCOUNT0 + shard] = tokens queued in that shard
#define DVQ_COUNTER_BYTES 1024
// The counter block (ints) at the start of the filter path's workspace.  "report" words are only ever overwritten (what
// dvq_vq_assign_fallback_count_offset points at); "live" words are zero between ops: the list kernel's finishing workgroup puts
// them back (so does each chunk's last resolver slice with its ticket pair), and the zero kernel in front of the op is only
// launched for a caller that does not declare the workspace clean (DVQ_MODE_WS_CLEAN).
#define DVQ_C_QUEUED   0      // report: tokens queued for the resolver
#define DVQ_C_NEXACT   1      // report: tokens handed to the exact list
#define DVQ_C_EXACT    2      // live: exact-list append counter
#define DVQ_C_TICKET   4      // live: finalize ticket of the list kernel
#define DVQ_C_PREPASS  5
#define DVQ_SPLIT_TICKET0 128     // live: counters[DVQ_SPLIT_TICKET0 + token block] = slices done (split form of pass 1, small batches)
#ifndef DVQ_SPLIT_MAX_SLICES
#define DVQ_SPLIT_MAX_SLICES 8    // slices per token block at most (a power of two; 16: no faster at B = 4, 16 x 16, and the merge spills)
#endif
#define DVQ_SPLIT_MAX_BLOCKS 64   // ... which serves up to this many token blocks of 128 (beyond: no gain measured, profiles/r05_small_batch.json)
// Cache policy of the streaming reads of a batch's latents / branch features: up to this many bytes of the FINEST tensor (the coarser
// branches add a third) they are read with plain loads -- the 256-MB memory-side cache then serves what the producer wrote or the router
// gate just read -- above it with the non-temporal hint.  profiles/archive/r04_cache_policy.json: one batch at a time, plain loads pay up to
// ~150 MB (configs[1] -2.4 %, the gate op -5 %); with three batches in flight on three streams (how bench.py and a serving loop drive
// the op) they pay only while all three working sets fit the cache together, hence 64 MiB (B = 64 at 32 x 32 x 256); B = 128 under
// three streams loses 3 % with plain loads, B = 256 3 - 6 % even alone.
#ifndef DVQ_CACHED_MAX_BYTES
#define DVQ_CACHED_MAX_BYTES ((size_t)64 << 20)
#endif
#define DVQ_EXACT_LIST_BLOCKS 512   // grid of the list-mode exact kernel (2 per CU; it walks the list in chunks)

// Kernels with more than 64 KiB of dynamic LDS need the per-device opt-in once; `done` is the caller's
// static bitmask (one bit per device ordinal, so a process driving several GPUs opts in on each; relaxed
// atomics: two host threads may both apply the attribute, which is idempotent).  Returns the HIP error
// of hipFuncSetAttribute (0 = ok).
static inline int dvq_allow_dynamic_lds(const void *kernel, int bytes, unsigned long long *done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;     // shared bit: always re-apply
    if (dev != 63 && ((__atomic_load_n(done, __ATOMIC_RELAXED) >> dev) & 1ull)) return 0;
    const hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc != hipSuccess) return (int)rc;
    if (dev != 63) __atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELAXED);
    return 0;
}

// per-launch host-side error plumbing (dvq_abi.hip)
void dvq_set_error(const char *fmt, ...);
