/*
 * gsdf_table.h -- the voxel hash map in HBM (device side).
 *
 * Replaces the reference's two CPU containers
 *   tsdf_ : phmap::parallel_node_hash_map<Vector3i, SdfVoxel>   (MapGradPixelSdf.h:65-68)
 *   vis_  : phmap::parallel_flat_hash_map<Vec3i, vector<bool>>  (MapGradPixelSdf.h:70)
 * with a hash map of 4x4x4 VOXEL BLOCKS (the surface band is ~21 voxels thick, so blocks that exist
 * are ~70 % full):
 *
 *   bkeys[n_blocks]      u64 block key: (x>>2, y>>2, z>>2) + 2^18, 19 bits each (x low); ~0 = empty.
 *                        Open addressing, double hashing; entry i owns block i.  65536 blocks (the
 *                        2^22-voxel default) = 512 KB of keys: the probe runs out of the L2.
 *                        Capacity: 2^c voxel records = 2^(c-6) blocks; a surface map fills its blocks to
 *                        ~70 %, random-depth clutter to ~30 %; TABLE_FULL is reported when a probe
 *                        sequence of GSDF_MAX_PROBE entries finds no room (block load > ~0.95).
 *   vox[n_blocks * 64]   32-byte voxel records, index = block * 64 + (x&3 | (y&3)<<2 | (z&3)<<4):
 *          f32 w          sum of weights                      (SdfVoxel::weight)
 *          f32 s          sum of w * truncated sdf            (SdfVoxel::dist  = s / w)
 *          f32 gx,gy,gz   sum of w * R n                      (SdfVoxel::grad)
 *          u32 aux        serial of the last fusion launch that wrote the voxel
 *          u32 pad[2]
 *                        4 x-adjacent voxels share a 128-byte line, a block is 16 consecutive lines.
 *
 * A voxel EXISTS iff its w > 0: the reference creates tsdf_[vi] only for a sample with w > 0
 * (MapGradPixelSdf.cpp:108-109), weights are positive and only ever added, and records are zeroed by
 * the table clear.  The running mean of the reference (MapGradPixelSdf.cpp:111) equals s / w, so
 * storing the additive sums makes fusion order-free and shard-mergeable.
 *
 * Why blocks: a camera tile touches ~1000 voxels per frame but only ~50 blocks and ~400 lines, so
 * the fusion flush and the tracker's gather move 2.6x fewer lines than with per-voxel hashing, the
 * key probe is an L2 hit, and full-map sweeps (export, PhotoBA) stream dense memory.
 */
#ifndef GSDF_TABLE_H_
#define GSDF_TABLE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSDF_KEY_EMPTY   0xFFFFFFFFFFFFFFFFull
#define GSDF_KEY_OFF     (1 << 20)
#define GSDF_KEY_MASK    0x1FFFFFull
#define GSDF_BLOCK_VOX   64           /* voxels per 4x4x4 block */
#define GSDF_MAX_PROBE   1024         /* block keys probed before reporting TABLE_FULL */

struct __attribute__((aligned(32))) gsdf_payload {
    float w, s, gx, gy, gz;
    uint32_t aux;
    uint32_t pad[2];
};

struct gsdf_table {
    unsigned long long* bkeys;        /* [block_mask + 1] */
    gsdf_payload* vox;                /* [(block_mask + 1) * 64] */
    uint32_t block_mask;              /* number of blocks - 1 (power of two) */
    /* Block filter: one bit per block KEY, at a hashed position among 64 x n_blocks bits (512 KB at the 2^22-voxel default).
     * A clear bit proves the block absent with ONE load and a few instructions -- no key packing, no 64-bit hash, no walk of
     * the probe sequence to the first empty entry; a set bit (present, or 1 in ~150 absent blocks at 40 % load) is followed by
     * the normal probe.  Its user is the raycaster, most of whose samples lie in empty space.
     * The filters are REBUILT from the key array when a raycast finds the map changed (k_occ_rebuild, a few us): setting the
     * bits where blocks are inserted -- inside the fusion kernel's flush -- cost that kernel 24 vector registers and 16 more
     * spilled scalar registers (measured: 97 -> 121 VGPRs, +2 us per fusion), for a structure the frame loop never reads. */
    uint32_t* occ;                    /* 64 x n_blocks bits, then the cell filter below */
    /* The same one level up: one bit per CELL of 8x8x8 blocks (32^3 voxels) that holds a block, among n_blocks bits (8 KB at
     * the default: L1-resident), stored behind the block filter.  A ray crossing empty space tests the cell of its sample and,
     * when the bit is clear, knows every sample up to the cell's far face to be missing without looking at any of them (the
     * raycaster's empty-space skip).
     * Sizes and the second pointer are derived from block_mask (gsdf_occ_mask / gsdf_occ2_mask / gsdf_occ2): the struct is a
     * kernel argument of the fusion kernel, which has no scalar registers to spare. */
};
__host__ __device__ __forceinline__ uint32_t gsdf_occ_mask(const gsdf_table& T) { return (T.block_mask << 6) | 63u; }   /* filter bits - 1 */
__host__ __device__ __forceinline__ uint32_t gsdf_occ2_mask(const gsdf_table& T) { return T.block_mask | 31u; }         /* cell filter bits - 1 (>= one word) */
__host__ __device__ __forceinline__ uint32_t* gsdf_occ2(const gsdf_table& T) { return T.occ + 2 * ((size_t)T.block_mask + 1); }
#define GSDF_CELL_SHIFT 5             /* voxels per cell edge = 32 */

__host__ __device__ __forceinline__ bool gsdf_key_in_range(int x, int y, int z) {
    return x >= -GSDF_KEY_OFF && x < GSDF_KEY_OFF && y >= -GSDF_KEY_OFF && y < GSDF_KEY_OFF &&
           z >= -GSDF_KEY_OFF && z < GSDF_KEY_OFF;
}
/* packed VOXEL key: x,y,z + 2^20, 21 bits each (x low, z high); orders like (z, y, x), the export order */
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
/* block key and in-block index of a packed voxel key (the 2^20 bias is a multiple of 4: floor semantics) */
__host__ __device__ __forceinline__ unsigned long long gsdf_block_key(unsigned long long k) {
    return ((k >> 2) & 0x7FFFFull) | (((k >> 23) & 0x7FFFFull) << 19) | (((k >> 44) & 0x7FFFFull) << 38);
}
__host__ __device__ __forceinline__ uint32_t gsdf_block_local(unsigned long long k) {
    return (uint32_t)(k & 3ull) | ((uint32_t)((k >> 21) & 3ull) << 2) | ((uint32_t)((k >> 42) & 3ull) << 4);
}
/* packed voxel key of voxel `local` of the block with key `bk` */
__host__ __device__ __forceinline__ unsigned long long gsdf_voxel_key(unsigned long long bk, uint32_t local) {
    const unsigned long long ux = ((bk & 0x7FFFFull) << 2) | (local & 3u);
    const unsigned long long uy = (((bk >> 19) & 0x7FFFFull) << 2) | ((local >> 2) & 3u);
    const unsigned long long uz = (((bk >> 38) & 0x7FFFFull) << 2) | ((local >> 4) & 3u);
    return ux | (uy << 21) | (uz << 42);
}
/* Hash choice is free: std::hash<Vec3i> (hash_map.h:44-52) only fixes phmap's iteration order.
 * The 64-bit murmur finaliser of the packed block key; low half = home entry, high half = probe step.
 * Cheaper forms were measured in round 3 (a wave64 instruction occupies its SIMD for 4 cycles, the finaliser's two 64 x 64
 * multiplies are ~50 issue slots per lookup, a tracker pass does three lookups per lane):
 *  - a lattice form of the block coordinates + one xor-shift (~10 slots): same probe counts in simulation (1.23 per key) --
 *    and the tracker 4 us per pass SLOWER: the home entry decides where a block's 2 KB of voxel records live, neighbouring
 *    blocks landed in neighbouring entries, and the records of a wall became one contiguous run of memory that all 256
 *    workgroups hit at once (slowest workgroup's gather 5 -> 12 us: memory channels hot-spot).  Placement must be random;
 *  - the lattice form + the 32-bit murmur finaliser (~20 slots, random placement again): gather median -0.1 us, launch span
 *    unchanged within noise -- a pass waits for its slowest wave, not for instruction issue.  Not worth a change of the map's
 *    layout function, so the finaliser stays.  (Round 4, the same form measured on the whole bench: fusion 59.5 against 59.3 us,
 *    tracker launches 8.97 against 9.0, raycast 116.7 against 115.9 -- nothing.) */
__host__ __device__ __forceinline__ unsigned long long gsdf_hash64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
__host__ __device__ __forceinline__ uint32_t gsdf_hash(unsigned long long k) { return (uint32_t)gsdf_hash64(k); }
/* Probe sequence of a block key with home entry h0: double hashing -- h0, h0 + step, h0 + 2 step, ... with an odd
 * step, so every entry is visited.  (Measured alternative: the 16 entries of the home entry's 128-byte line first, so
 * that later probes are L1 hits -- slower at every load factor: in-line clustering makes the longest chain of a wave,
 * which is what a wave waits for, several times longer.) */
__host__ __device__ __forceinline__ uint32_t gsdf_probe_step(unsigned long long bk) {
    return (uint32_t)(gsdf_hash64(bk) >> 32) | 1u;
}

/* Position of a block in the filter, from its biased block coordinates ((v + 2^20) >> 2, 19 bits each -- the fields of a block
 * key): a LATTICE hash, bx + C2 by + C3 bz (24-bit odd constants: two full-rate 24-bit multiply-adds; the walk through empty
 * space evaluates it once per sample).  Neighbouring blocks never share a bit; blocks far apart alias at random, which only
 * costs a wasted probe (a set bit is always followed by the real lookup). */
__host__ __device__ __forceinline__ uint32_t gsdf_occ_index(const gsdf_table& T, uint32_t bx, uint32_t by, uint32_t bz) {
    return (bx + by * 0x9E3779u + bz * 0x85EBCBu) & gsdf_occ_mask(T);
}
__host__ __device__ __forceinline__ uint32_t gsdf_occ_bit(const gsdf_table& T, unsigned long long bk) {
    return gsdf_occ_index(T, (uint32_t)(bk & 0x7FFFFull), (uint32_t)((bk >> 19) & 0x7FFFFull), (uint32_t)((bk >> 38) & 0x7FFFFull));
}
/* the same from voxel indices that passed gsdf_key_in_range */
__host__ __device__ __forceinline__ uint32_t gsdf_occ_bit_vox(const gsdf_table& T, int x, int y, int z) {
    return gsdf_occ_index(T, (uint32_t)(x + GSDF_KEY_OFF) >> 2, (uint32_t)(y + GSDF_KEY_OFF) >> 2, (uint32_t)(z + GSDF_KEY_OFF) >> 2);
}

/* position of a cell in the cell filter, from its biased cell coordinates ((v + 2^20) >> 5, 16 bits each) */
__host__ __device__ __forceinline__ uint32_t gsdf_occ2_index(const gsdf_table& T, uint32_t cx, uint32_t cy, uint32_t cz) {
    return (cx + cy * 0x6C8E95u + cz * 0xB5297Bu) & gsdf_occ2_mask(T);
}
__host__ __device__ __forceinline__ uint32_t gsdf_occ2_bit_vox(const gsdf_table& T, int x, int y, int z) {
    return gsdf_occ2_index(T, (uint32_t)(x + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT, (uint32_t)(y + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT,
                           (uint32_t)(z + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT);
}

#if defined(__HIPCC__)
/* the bits of one existing block: block filter and cell filter (k_occ_rebuild) */
__device__ __forceinline__ void gsdf_occ_set(const gsdf_table& T, unsigned long long bk) {
    const uint32_t bx = (uint32_t)(bk & 0x7FFFFull), by = (uint32_t)((bk >> 19) & 0x7FFFFull), bz = (uint32_t)((bk >> 38) & 0x7FFFFull);
    const uint32_t b = gsdf_occ_index(T, bx, by, bz);
    atomicOr(&T.occ[b >> 5], 1u << (b & 31u));
    /* up to 512 blocks share a cell, 32 cells a word: only the first block of a cell needs the atomic (same-address atomics
     * serialise: 25 us for the bench map without this look; a stale look only costs a redundant atomic) */
    const uint32_t c = gsdf_occ2_index(T, bx >> (GSDF_CELL_SHIFT - 2), by >> (GSDF_CELL_SHIFT - 2), bz >> (GSDF_CELL_SHIFT - 2));
    uint32_t* w = &gsdf_occ2(T)[c >> 5];
    if (!((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (c & 31u)) & 1u)) atomicOr(w, 1u << (c & 31u));
}
/* false: the block is certainly absent */
__device__ __forceinline__ bool gsdf_occ_test(uint32_t word, uint32_t bit) { return (word >> (bit & 31u)) & 1u; }

/* Looks N block keys up TOGETHER: every round evaluates probe r of all pending keys and then issues probe r + 1 of those
 * still pending back to back, so a lane's N chains of dependent loads overlap (a wave's round count is the longest
 * chain of any of its lanes, not the sum over the N keys).  k[e] = the key already loaded from the home entry h[e]
 * (callers load it early; h and k are scratch afterwards); bit e of `pend` = key e takes part.  b[e] = block index or -1 (INSERT: probe budget
 * exhausted = table full; else: block absent).
 * Keys never change once written and records are zeroed by the table clear, so keys are read with plain loads: a stale
 * read can only show EMPTY, and then the CAS is authoritative (INSERT) -- kernels that only look up run after the
 * inserting kernel has ended. */
template <int N, bool INSERT>
__device__ __forceinline__ void gsdf_block_lookup_n(const gsdf_table& T, const unsigned long long (&bk)[N], uint32_t (&h)[N],
                                                    unsigned long long (&k)[N], uint32_t pend, int (&b)[N]) {
    uint32_t step[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { b[e] = -1; step[e] = 1u; }
    if constexpr (!INSERT) {
        /* Lookups only: TWO entries of the probe sequence per round.  A wave waits for the longest chain among its lanes' keys
         * (64 x N of them: 4-6 entries at 40 % load, measured per wave with tools/track_waves.py: the lookups are 60 % of the
         * tracker's gather) and every round is a dependent L2 round trip; the second entry's load costs nothing but a request.
         * The order of the sequence is kept (first the entry, then its successor), so the result is the same. */
        unsigned long long k2[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            step[e] = gsdf_probe_step(bk[e]);
            k2[e] = ((pend >> e) & 1u) ? T.bkeys[(h[e] + step[e]) & T.block_mask] : GSDF_KEY_EMPTY;
        }
        for (int r = 0; r < GSDF_MAX_PROBE; r += 2) {
#pragma unroll
            for (int e = 0; e < N; ++e) {
                if (!((pend >> e) & 1u)) continue;
                if (k[e] == bk[e]) { b[e] = (int)h[e]; pend &= ~(1u << e); }
                else if (k[e] == GSDF_KEY_EMPTY) pend &= ~(1u << e);           /* entries are never freed: an empty one ends the chain */
                else if (k2[e] == bk[e]) { b[e] = (int)((h[e] + step[e]) & T.block_mask); pend &= ~(1u << e); }
                else if (k2[e] == GSDF_KEY_EMPTY) pend &= ~(1u << e);
            }
            if (!__any(pend != 0u)) break;
#pragma unroll
            for (int e = 0; e < N; ++e)
                if ((pend >> e) & 1u) {
                    h[e] = (h[e] + 2u * step[e]) & T.block_mask;
                    k[e] = T.bkeys[h[e]];
                    k2[e] = T.bkeys[(h[e] + step[e]) & T.block_mask];
                }
        }
        return;
    }
    for (int r = 0; r < GSDF_MAX_PROBE; ++r) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            if (!((pend >> e) & 1u)) continue;
            unsigned long long kk = k[e];
            if (INSERT && kk == GSDF_KEY_EMPTY) {
                kk = atomicCAS(&T.bkeys[h[e]], GSDF_KEY_EMPTY, bk[e]);
                if (kk == GSDF_KEY_EMPTY) kk = bk[e];
            }
            if (kk == bk[e]) { b[e] = (int)h[e]; pend &= ~(1u << e); }
            else if (!INSERT && kk == GSDF_KEY_EMPTY) pend &= ~(1u << e);   /* entries are never freed: an empty one ends the chain */
        }
        if (!__any(pend != 0u)) break;
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < N; ++e) step[e] = gsdf_probe_step(bk[e]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e)
            if ((pend >> e) & 1u) { h[e] = (h[e] + step[e]) & T.block_mask; k[e] = T.bkeys[h[e]]; }
    }
}

/* Single-key forms.  `first` is the key already loaded from the home entry `h`.  Return the block index or -1. */
__device__ __forceinline__ int gsdf_block_find_or_insert(const gsdf_table& T, unsigned long long bk, uint32_t h,
                                                         unsigned long long first) {
    const unsigned long long bks[1] = { bk };
    uint32_t hs[1] = { h };
    unsigned long long ks[1] = { first };
    int bs[1];
    gsdf_block_lookup_n<1, true>(T, bks, hs, ks, 1u, bs);
    return bs[0];
}
__device__ __forceinline__ int gsdf_block_find(const gsdf_table& T, unsigned long long bk, uint32_t h,
                                               unsigned long long first) {
    const unsigned long long bks[1] = { bk };
    uint32_t hs[1] = { h };
    unsigned long long ks[1] = { first };
    int bs[1];
    gsdf_block_lookup_n<1, false>(T, bks, hs, ks, 1u, bs);
    return bs[0];
}

/* tsdf_[vi] (operator[]: find, insert zero-initialised if absent) -- MapGradPixelSdf.cpp:109.
 * `key` is the packed voxel key.  nullptr when the table is full. */
__device__ __forceinline__ gsdf_payload* gsdf_find_or_insert(const gsdf_table& T, unsigned long long key) {
    const unsigned long long bk = gsdf_block_key(key);
    const uint32_t h = gsdf_hash(bk) & T.block_mask;
    const int b = gsdf_block_find_or_insert(T, bk, h, T.bkeys[h]);
    return b < 0 ? nullptr : T.vox + ((size_t)b * GSDF_BLOCK_VOX + gsdf_block_local(key));
}
/* tsdf_.find(idx) -- MapGradPixelSdf.h:119.  nullptr when the block does not exist; the caller
 * tests w > 0 for the voxel itself. */
__device__ __forceinline__ const gsdf_payload* gsdf_find(const gsdf_table& T, unsigned long long key) {
    const unsigned long long bk = gsdf_block_key(key);
    const uint32_t h = gsdf_hash(bk) & T.block_mask;
    const int b = gsdf_block_find(T, bk, h, T.bkeys[h]);
    return b < 0 ? nullptr : T.vox + ((size_t)b * GSDF_BLOCK_VOX + gsdf_block_local(key));
}
#endif

#endif /* GSDF_TABLE_H_ */
