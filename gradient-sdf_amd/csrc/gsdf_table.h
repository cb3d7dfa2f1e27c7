/*
 * gsdf_table.h -- the voxel hash map in HBM (device side).
 *
 * Replaces the reference's two CPU containers
 *   tsdf_ : phmap::parallel_node_hash_map<Vector3i, SdfVoxel>   (MapGradPixelSdf.h:65-68)
 *   vis_  : phmap::parallel_flat_hash_map<Vec3i, vector<bool>>  (MapGradPixelSdf.h:70)
 * with ONE open-addressed table of 32-byte slots:
 *
 *   +0  u64  key    x,y,z + 2^20 packed 21 bits each (x low, z high); ~0 = empty
 *   +8  f32  w      sum of weights                      (SdfVoxel::weight)
 *   +12 f32  s      sum of w * truncated sdf            (SdfVoxel::dist  = s / w)
 *   +16 f32  gx,gy,gz  sum of w * R n                   (SdfVoxel::grad)
 *   +28 u32  aux    last frame index that touched the voxel + 1 (vis_ stand-in)
 *
 * Four slots form one 128-byte bucket = one HBM/L2 line, so a probe fetches a
 * whole bucket with one coalesced line read; probing is linear over buckets.
 * The running mean of the reference (MapGradPixelSdf.cpp:111) equals s / w, so
 * storing the additive sums makes fusion order-free (atomics) and shard-mergeable.
 * The packed key orders like (z, y, x), the order exports are sorted in.
 */
#ifndef GSDF_TABLE_H_
#define GSDF_TABLE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSDF_KEY_EMPTY   0xFFFFFFFFFFFFFFFFull
#define GSDF_KEY_OFF     (1 << 20)
#define GSDF_KEY_MASK    0x1FFFFFull
#define GSDF_BUCKET      4            /* slots per 128-byte bucket */
#define GSDF_MAX_PROBE   128          /* buckets probed before reporting TABLE_FULL */

struct __attribute__((aligned(32))) gsdf_slot {
    unsigned long long key;
    float w, s, gx, gy, gz;
    uint32_t aux;
};

struct gsdf_table {
    gsdf_slot* slots;
    uint32_t bucket_mask;             /* number of buckets - 1 */
};

__host__ __device__ __forceinline__ bool gsdf_key_in_range(int x, int y, int z) {
    return x >= -GSDF_KEY_OFF && x < GSDF_KEY_OFF && y >= -GSDF_KEY_OFF && y < GSDF_KEY_OFF &&
           z >= -GSDF_KEY_OFF && z < GSDF_KEY_OFF;
}
__host__ __device__ __forceinline__ unsigned long long gsdf_key_pack(int x, int y, int z) {
    return (unsigned long long)(uint32_t)(x + GSDF_KEY_OFF) |
           ((unsigned long long)(uint32_t)(y + GSDF_KEY_OFF) << 21) |
           ((unsigned long long)(uint32_t)(z + GSDF_KEY_OFF) << 42);
}
__host__ __device__ __forceinline__ void gsdf_key_unpack(unsigned long long k, int* x, int* y, int* z) {
    *x = (int)(k & GSDF_KEY_MASK) - GSDF_KEY_OFF;
    *y = (int)((k >> 21) & GSDF_KEY_MASK) - GSDF_KEY_OFF;
    *z = (int)((k >> 42) & GSDF_KEY_MASK) - GSDF_KEY_OFF;
}
/* hash choice is free: std::hash<Vec3i> (hash_map.h:44-52) only fixes phmap's iteration order */
__host__ __device__ __forceinline__ uint32_t gsdf_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}

#if defined(__HIPCC__)
/* tsdf_[vi] (operator[]: find, insert zero-initialised if absent) -- MapGradPixelSdf.cpp:109.
 * Returns the slot index or -1 when the probe budget is exhausted.  Keys never change once
 * written, so a stale read can only show EMPTY, and then the CAS is authoritative.
 * *inserted is set when this call created the voxel. */
__device__ __forceinline__ long long gsdf_find_or_insert(const gsdf_table& T, unsigned long long key, bool* inserted) {
    uint32_t b = gsdf_hash(key) & T.bucket_mask;
    *inserted = false;
    for (int probe = 0; probe < GSDF_MAX_PROBE; ++probe) {
        gsdf_slot* base = T.slots + (size_t)b * GSDF_BUCKET;
#pragma unroll
        for (int j = 0; j < GSDF_BUCKET; ++j) {
            unsigned long long k = __hip_atomic_load(&base[j].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == GSDF_KEY_EMPTY) {
                k = atomicCAS(&base[j].key, GSDF_KEY_EMPTY, key);
                if (k == GSDF_KEY_EMPTY) { *inserted = true; return (long long)((size_t)b * GSDF_BUCKET + j); }
            }
            if (k == key) return (long long)((size_t)b * GSDF_BUCKET + j);
        }
        b = (b + 1) & T.bucket_mask;
    }
    return -1;
}

/* tsdf_.find(idx) -- MapGradPixelSdf.h:119.  Read-only kernels only (plain loads). */
__device__ __forceinline__ const gsdf_slot* gsdf_find(const gsdf_table& T, unsigned long long key) {
    uint32_t b = gsdf_hash(key) & T.bucket_mask;
    for (int probe = 0; probe < GSDF_MAX_PROBE; ++probe) {
        const gsdf_slot* base = T.slots + (size_t)b * GSDF_BUCKET;
#pragma unroll
        for (int j = 0; j < GSDF_BUCKET; ++j) {
            const unsigned long long k = base[j].key;
            if (k == key) return base + j;
            if (k == GSDF_KEY_EMPTY) return nullptr;
        }
        b = (b + 1) & T.bucket_mask;
    }
    return nullptr;
}
#endif

#endif /* GSDF_TABLE_H_ */
