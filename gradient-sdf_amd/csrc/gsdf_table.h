/*
 * gsdf_table.h -- the voxel hash map in HBM (device side).
 *
 * Replaces the reference's two CPU containers
 *   tsdf_ : phmap::parallel_node_hash_map<Vector3i, SdfVoxel>   (MapGradPixelSdf.h:65-68)
 *   vis_  : phmap::parallel_flat_hash_map<Vec3i, vector<bool>>  (MapGradPixelSdf.h:70)
 * with ONE open-addressed table of 128-byte buckets (= one HBM / L2 line) of 4 voxels:
 *
 *   +0   u64 key[4]     x,y,z + 2^20 packed 21 bits each (x low, z high); ~0 = empty
 *   +32  payload[4]     6 x 4 bytes each:
 *          f32 w          sum of weights                      (SdfVoxel::weight)
 *          f32 s          sum of w * truncated sdf            (SdfVoxel::dist  = s / w)
 *          f32 gx,gy,gz   sum of w * R n                      (SdfVoxel::grad)
 *          u32 aux        index+1 of the last frame that touched the voxel (vis_ stand-in,
 *                         and the per-frame ownership tag of the fusion flush)
 *
 * A probe reads the 4 keys of a bucket with two 16-byte loads of ONE line (coalesced probe);
 * the payload of a hit sits in the same line.  Probing is linear over buckets; a capacity of
 * 2^c "slots" means 2^(c-2) buckets.  The running mean of the reference
 * (MapGradPixelSdf.cpp:111) equals s / w, so storing the additive sums makes fusion
 * order-free (atomics) and shard-mergeable.  Packed keys order like (z, y, x), the order
 * exports are sorted in.
 */
#ifndef GSDF_TABLE_H_
#define GSDF_TABLE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSDF_KEY_EMPTY   0xFFFFFFFFFFFFFFFFull
#define GSDF_KEY_OFF     (1 << 20)
#define GSDF_KEY_MASK    0x1FFFFFull
#define GSDF_BUCKET      4            /* voxels per 128-byte bucket */
#define GSDF_MAX_PROBE   128          /* buckets probed before reporting TABLE_FULL */

struct gsdf_payload {
    float w, s, gx, gy, gz;
    uint32_t aux;
};
struct __attribute__((aligned(128))) gsdf_bucket {
    unsigned long long key[GSDF_BUCKET];
    gsdf_payload pay[GSDF_BUCKET];
};

struct gsdf_table {
    gsdf_bucket* buckets;
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
 * Returns the payload slot or nullptr when the probe budget is exhausted.  Keys never change
 * once written and payloads are zeroed by the table clear, so the 4 keys are read with plain
 * 16-byte loads: a stale read can only show EMPTY, and then the CAS is authoritative. */
__device__ __forceinline__ gsdf_payload* gsdf_find_or_insert(const gsdf_table& T, unsigned long long key) {
    uint32_t b = gsdf_hash(key) & T.bucket_mask;
    for (int probe = 0; probe < GSDF_MAX_PROBE; ++probe) {
        gsdf_bucket* B = T.buckets + b;
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&B->key[0]);
        const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&B->key[2]);
        const unsigned long long ks[4] = { k01.x, k01.y, k23.x, k23.y };
#pragma unroll
        for (int j = 0; j < GSDF_BUCKET; ++j) {
            unsigned long long k = ks[j];
            if (k == GSDF_KEY_EMPTY) {
                k = atomicCAS(&B->key[j], GSDF_KEY_EMPTY, key);
                if (k == GSDF_KEY_EMPTY) return &B->pay[j];
            }
            if (k == key) return &B->pay[j];
        }
        b = (b + 1) & T.bucket_mask;
    }
    return nullptr;
}

/* tsdf_.find(idx) -- MapGradPixelSdf.h:119.  Read-only kernels only. */
__device__ __forceinline__ const gsdf_payload* gsdf_find(const gsdf_table& T, unsigned long long key) {
    uint32_t b = gsdf_hash(key) & T.bucket_mask;
    for (int probe = 0; probe < GSDF_MAX_PROBE; ++probe) {
        const gsdf_bucket* B = T.buckets + b;
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&B->key[0]);
        const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&B->key[2]);
        if (k01.x == key) return &B->pay[0];
        if (k01.y == key) return &B->pay[1];
        if (k23.x == key) return &B->pay[2];
        if (k23.y == key) return &B->pay[3];
        /* slots fill in order and are never freed: an empty slot ends the probe sequence */
        if (k23.y == GSDF_KEY_EMPTY) return nullptr;
        b = (b + 1) & T.bucket_mask;
    }
    return nullptr;
}
#endif

#endif /* GSDF_TABLE_H_ */
