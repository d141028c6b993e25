// kernels.hip -- the wavefront path tracer's device code for gfx950 (CDNA4, wave64).
//
// Replaces the OptiX pipeline of the reference (shader/Shader.cu: raygen :246-273, miss
// :276-287, closest-hit :94-242, :297-310; RT-core traversal requested at
// src/Global/RendererImpl.cu:295-314) with explicit kernels:
//
//   k_rng_init      XORWOW curand_init restated (HostFunctions.cu:122-127)
//   k_generate      primary rays                 (Shader.cu:246-267)
//   k_traverse      persistent-wave BVH8 closest-hit / any-hit traversal (optixTrace, Shader.cu:70)
//   k_bin_hits      sort hits by program with wave ballot + prefix count (SBT dispatch)
//   k_shade<P>      one kernel per closest-hit program (Shader.cu:108-233)
//   k_accumulate    path termination: miss colour / depth cut-off, innermost-first albedo fold
//                   (Shader.cu:102-107, :236-238, :276-287)
//   k_finalize      mean over samples + colorToFloat4 + AOVs (Shader.cu:270-272)
//   k_to_rgba8      convertFloat4ToUchar4Kernel (RendererImpl.cu:672-678)
//
// Arithmetic that feeds control flow is written with a fixed operation order and compiled
// with -ffp-contract=off so that it is bit-identical to the CPU oracle; fused multiply-adds
// appear only where written explicitly (fmaf) in the conservative box test.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "srgb_pow.h"
#include "trav_common.h"

#pragma clang fp contract(off)

namespace hrt {
// curand_init(seed = tid ^ salt, subsequence = tid, offset = 0): src/Global/HostFunctions.cu:122-127,
// bounds-checked and indexed by the frame width (quirk Q9), clock64() pinned to salt (Q8).
// jump[k] = T^(2^(67+k)) as 160 columns of 5 words.
__global__ __launch_bounds__(256) void k_rng_init(RngState *states, uint32_t n, uint64_t salt,
                                                  const uint32_t *__restrict__ jump) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const uint64_t seed = (uint64_t)tid ^ salt;
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    uint32_t v[5];
    const uint32_t d = 6615241u + t1 + t0;
    v[0] = 123456789u + t0; v[1] = 362436069u ^ t0; v[2] = 521288629u + t1; v[3] = 88675123u ^ t1; v[4] = 5783321u + t0;
    for (uint32_t k = 0; k < 32 && (tid >> k) != 0; ++k) {
        if (((tid >> k) & 1u) == 0) continue;
        const uint32_t *m = jump + (size_t)k * 800;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
        for (int w = 0; w < 5; ++w) {
            const uint32_t word = v[w];
            for (int b = 0; b < 32; ++b) {
                const uint32_t msk = 0u - ((word >> b) & 1u);
                const uint32_t *c = m + (w * 32 + b) * 5;
                r0 ^= c[0] & msk; r1 ^= c[1] & msk; r2 ^= c[2] & msk; r3 ^= c[3] & msk; r4 ^= c[4] & msk;
            }
        }
        v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4;
    }
    RngState out;
    out.d = d; out.v[0] = v[0]; out.v[1] = v[1]; out.v[2] = v[2]; out.v[3] = v[3]; out.v[4] = v[4];
    out.boxmuller_flag = 0; out.boxmuller_flag_double = 0; out.boxmuller_extra = 0.0f; out._pad = 0.0f;
    out.boxmuller_extra_double = 0.0;
    states[tid] = out;
}

// ---------------------------------------------------------------------------------------
// generate: __raygen__raygenProgram up to the trace call, shader/Shader.cu:246-267
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_generate(GenerateArgs a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n_tile_pixels) return;
    const uint32_t j = a.first_pixel + q;
    const uint32_t row = j / a.width;
    const uint32_t ix = j - row * a.width;
    const uint32_t iy = a.rows[row];
    const V3 dir = primary_direction(ix, iy, a.width, a.height, a.U, a.V, a.W);
    RayRec r;
    r.o = make_float4(a.center[0], a.center[1], a.center[2], __uint_as_float(j));
    r.d = make_float4(dir.x, dir.y, dir.z, __uint_as_float(iy * a.width + ix));
    a.rays[q] = r;
}

template <bool COUNT, bool HAS_SPHERES, bool DMA, bool FUSED>
// 5 waves per SIMD: the register allocator is held to 96 VGPRs (the fused instantiations spill 27 / 71 dwords to scratch,
// which costs less than the fifth wave gives: profiles/r01_sweep_occupancy.txt)
__global__ __launch_bounds__(kTraverseBlock, 5) void k_traverse(TraverseArgs a) {
    static_assert(kTraverseBlock == 64, "one wave per workgroup: staging images and mailboxes are per wave");
    __shared__ uint2 s_stack[kLdsStack][kTraverseBlock];
    __shared__ uint4 s_node_stage[DMA ? 5 * 64 : 1];          // 64 node slots x 80 B, filled by LDS-DMA
    __shared__ uint4 s_prim_stage[DMA ? 3 * 64 : 1];          // 64 primitive slots x 48 B
    // tail splitting: one mailbox per lane that owns a split ray (indexed by its home lane)
    __shared__ float s_mb_t[kTraverseBlock], s_mb_u[kTraverseBlock], s_mb_v[kTraverseBlock];
    __shared__ uint32_t s_mb_prim[kTraverseBlock], s_mb_inst[kTraverseBlock], s_mb_pending[kTraverseBlock];
    __shared__ uint32_t s_pair[kTraverseBlock];
    uint2 spill[kSpillStack];

    // up to two queue segments per launch (e.g. the depth-4 rays of sample s and the primary rays of
    // sample s+1): ray i < n_a comes from segment 0, the others from segment 1
    const uint32_t n_a = a.seg[0].n_ptr ? (a.seg[0].n_ptr[0] + a.seg[0].n_ptr[1] + a.seg[0].n_ptr[2] + a.seg[0].n_ptr[3]) : a.seg[0].n;
    const uint32_t n_b = a.seg[1].rays ? (a.seg[1].n_ptr ? (a.seg[1].n_ptr[0] + a.seg[1].n_ptr[1] + a.seg[1].n_ptr[2] + a.seg[1].n_ptr[3]) : a.seg[1].n) : 0u;
    const uint32_t n_rays = FUSED ? a.path.n_tile_pixels : n_a + n_b;     // fused: the queue is the tile's pixel list
    const char *__restrict__ node_bytes = reinterpret_cast<const char *>(a.nodes);
    const char *__restrict__ prim_bytes = reinterpret_cast<const char *>(a.prims);
    const float tmin = a.tmin, tmax_ray = a.tmax;
    const uint32_t tx = threadIdx.x;                // = lane

    // gather geometry: in DMA instruction i this lane fetches piece g_off[i] of the slot owned by lane g_own[i]
    uint32_t gn_own[5], gn_off[5], gp_own[3], gp_off[3];
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) { const uint32_t x = 64u * i + tx; gn_own[i] = x / 5u; gn_off[i] = (x % 5u) * 16u; }
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) { const uint32_t x = 64u * i + tx; gp_own[i] = x / 3u; gp_off[i] = (x % 3u) * 16u; }
    const uint32_t node_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&s_node_stage[0]));
    const uint32_t prim_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&s_prim_stage[0]));

    TravState s;
    bool alive = false;
    bool exhausted = false;                 // wave-uniform
    bool shared = false;                    // this lane works on a ray that has been split across lanes
    uint32_t home = tx;                     // lane whose mailbox collects the split ray's result
    uint32_t cnt_nodes = 0, cnt_prims = 0, cnt_nodes_b = 0, cnt_prims_b = 0;
    bool any = false;                       // this lane's ray only needs to know whether anything is hit
    bool in_b = false;                      // ... and belongs to segment 1

    // FUSED (path mode): the lane owns a pixel and carries its path state; a finished ray is shaded in
    // place and the next ray (bounce, next sample, next pixel) starts in the same lane -- no queues, no
    // per-stage launches, no stage barriers.  The default mode of the production build.
    bool have_pixel = false, waiting = false, px_first = true;
    uint32_t px_local = 0u, px_tid = 0u, px_sample = 0u, px_depth = 1u;
    uint32_t px_chain[4] = {0u, 0u, 0u, 0u};
    float px_ax = 0.0f, px_ay = 0.0f, px_az = 0.0f;
    uint32_t px_t0 = 0u;                                    // probe launch: clock at the pixel's start
    float px_pdx = 0.0f, px_pdy = 0.0f, px_pdz = 1.0f;      // the pixel's primary direction: the same for every sample (no jitter, Shader.cu:249-261)
    Xorwow px_rng{};
    uint32_t px_rays_closest = 0u, px_rays_any = 0u;
#ifdef HRT_LANE_STATS
    unsigned long long ls_iter = 0, ls_alive = 0, ls_node = 0, ls_prim = 0, ls_ppass = 0, ls_regen = 0;
#endif

    // what the next iteration gathers for this lane
    bool has_node = false, has_prim = false;
    uint32_t nidx = 0u, pidx = 0u;

    // pop the next node group / leaf group from the stack as needed, then take the nearest child
    // of the node group in hand: sets has_node / nidx
    auto advance_select = [&]() {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (s.cur.y <= 0x00ffffffu && s.sp > s.base) {
                const uint2 top = (s.sp - 1) < kLdsStack ? s_stack[s.sp - 1][tx] : spill[s.sp - 1 - kLdsStack];
                if (top.y > 0x00ffffffu) { s.cur = top; --s.sp; }
                else if (s.ptri.y == 0u) { s.ptri = top; --s.sp; }
            }
        }
        if (s.cur.y > 0x00ffffffu) {
            const uint32_t hits_imask = s.cur.y;
            const uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
            s.cur.y &= ~(1u << bit);
            const uint32_t slot_index = (bit - 24u) ^ (s.oct_inv4 & 0xffu);
            const uint32_t rel = (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot_index));
            nidx = s.cur.x + rel;
            if (s.cur.y > 0x00ffffffu) {            // siblings still to visit: keep them on the stack
                if (s.sp < kLdsStack) s_stack[s.sp][tx] = s.cur; else spill[s.sp - kLdsStack] = s.cur;
                ++s.sp;
            }
            s.cur = make_uint2(0u, 0u);
            has_node = true;
        } else { has_node = false; nidx = 0u; }
    };

    // wave-local slice of the queue: [wbeg, wend).  Slices of fetch_chunk rays are handed out by
    // kFetchShards counters (chunk c of shard s covers rays (c * kFetchShards + s) * fetch_chunk ...),
    // so one launch costs n_rays / fetch_chunk atomics spread over 8 addresses instead of one
    // atomic per refill on a single word (measured: the single word capped the kernel).
    uint32_t wbeg = 0, wend = 0;
    const uint32_t home_shard = blockIdx.x & (kFetchShards - 1);
    uint32_t kstart = 0;

    for (;;) {
        const uint64_t idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (FUSED) {
            // ---- regenerate: shade finished rays in place, start the next sample / pixel ----
            if (n_idle >= (uint32_t)a.refill_threshold || idle == ~0ull) {
#ifdef HRT_LANE_STATS
                ++ls_regen;
#endif
                auto start_ray = [&](V3 o, V3 d) {
                    s.ox = o.x; s.oy = o.y; s.oz = o.z; s.dx = d.x; s.dy = d.y; s.dz = d.z;
                    s.idx = safe_rcp_dir<COUNT>(s.dx); s.idy = safe_rcp_dir<COUNT>(s.dy); s.idz = safe_rcp_dir<COUNT>(s.dz);
                    const uint32_t oct = (s.dx < 0.0f ? 4u : 0u) | (s.dy < 0.0f ? 2u : 0u) | (s.dz < 0.0f ? 1u : 0u);
                    s.oct_inv4 = (7u - oct) * 0x01010101u;
                    s.bt = tmax_ray; s.bu = 0.0f; s.bv = 0.0f; s.bprim = kMissPrim; s.binst = kMissPrim;
                    s.cur = make_uint2(0u, 0x80000000u);
                    s.ptri = make_uint2(0u, 0u);
                    s.sp = 0; s.base = 0;
                    any = px_depth >= kRayTraceDepth;      // a hit at the depth limit is black whatever it is
                    if (any) ++px_rays_any; else ++px_rays_closest;
                    alive = true; has_prim = false; pidx = 0u;
                    advance_select();
                };
                bool want_primary = false;
                if (!alive && waiting) {
                    waiting = false;
                    const bool miss = s.bprim == kMissPrim;
                    if (a.path.trace_rays) {           // hrt_trace_rays on this kernel: the "pixel" is a caller's ray, its hit record the result
                        a.path.trace_tuvp[px_local] = make_float4(s.bt, s.bu, s.bv, __uint_as_float(s.bprim));
                        a.path.trace_inst[px_local] = s.binst;
                        have_pixel = false;
                    } else if (miss || px_depth >= kRayTraceDepth) {
                        const V3 r = fold_chain(miss, a.path.bg, px_chain, px_depth, a.path.hitgroups);
                        if (px_first) { px_ax = r.x; px_ay = r.y; px_az = r.z; px_first = false; }
                        else { px_ax += r.x; px_ay += r.y; px_az += r.z; }
                        ++px_sample;
                        if (a.path.slice_cost)     // probe launch: how long this pixel's sample took, start of its primary ray to here
                            atomicAdd(a.path.slice_cost + px_local / a.fetch_chunk, ((uint32_t)__builtin_amdgcn_s_memtime() - px_t0) >> 4);
                        if (px_sample >= a.path.spp) {
                            a.path.accum[px_local] = make_float4(px_ax, px_ay, px_az, 0.0f);
                            rng_store(a.path.states + px_tid, px_rng);
                            have_pixel = false;
                        } else want_primary = true;
                    } else {
                        const uint32_t inst = s.binst;
                        const HitGroup hg = a.path.hitgroups[inst];
                        const uint32_t program = a.path.inst_program[inst];
                        const V3 ro = mk3(s.ox, s.oy, s.oz), rd = mk3(s.dx, s.dy, s.dz);
                        V3 hp, nd;
                        if (program == (uint32_t)kProgramTriangleRough) scatter<false, true>(hg, ro, rd, s.bt, s.bu, s.bv, s.bprim, px_rng, hp, nd);
                        else if (program == (uint32_t)kProgramTriangleMetal) scatter<false, false>(hg, ro, rd, s.bt, s.bu, s.bv, s.bprim, px_rng, hp, nd);
                        else if (program == (uint32_t)kProgramSphereRough) scatter<true, true>(hg, ro, rd, s.bt, s.bu, s.bv, s.bprim, px_rng, hp, nd);
                        else scatter<true, false>(hg, ro, rd, s.bt, s.bu, s.bv, s.bprim, px_rng, hp, nd);
                        px_chain[px_depth - 1u] = inst;
                        ++px_depth;
                        start_ray(hp, nd);
                    }
                }
                // lanes without a pixel take the next ones of the wave's slice of the tile
                const uint64_t need = __ballot(!alive && !have_pixel && !want_primary);
                if (need != 0ull && !exhausted) {
                    if (wbeg >= wend) {
                        for (uint32_t k = kstart; k < kFetchShards && wbeg >= wend; ++k) {
                            const uint32_t shard = (home_shard + k) & (kFetchShards - 1);
                            uint32_t c = 0;
                            if (tx == 0u) c = atomicAdd(a.fetch_counter + shard * kFetchShardStride, 1u);
                            c = (uint32_t)__shfl((int)c, 0);
                            const uint64_t q = (uint64_t)c * kFetchShards + shard;          // the q-th slice handed out ...
                            if (q * (uint64_t)a.fetch_chunk < (uint64_t)n_rays) {
                                // ... is slice slice_order[q] of the tile: the expensive slices first, so that the render
                                // ends on cheap pixels (longest-processing-time-first; a pixel's samples run one after the other)
                                const uint64_t beg = (a.path.slice_order ? (uint64_t)a.path.slice_order[q] : q) * (uint64_t)a.fetch_chunk;
                                wbeg = (uint32_t)beg;
                                wend = (uint32_t)(beg + a.fetch_chunk < (uint64_t)n_rays ? beg + a.fetch_chunk : (uint64_t)n_rays);
                            } else kstart = k + 1;
                        }
                        if (wbeg >= wend) exhausted = true;
                    }
                    if (!exhausted) {
                        const uint32_t n_need = (uint32_t)__popcll(need);
                        const uint32_t take = n_need < wend - wbeg ? n_need : wend - wbeg;
                        const uint32_t rank = lane_prefix(need);
                        const uint32_t mine = wbeg + rank;
                        wbeg += take;
                        if (!alive && !have_pixel && !want_primary && rank < take) {
                            const uint32_t j = a.path.first_pixel + mine;
                            px_local = j;
                            have_pixel = true; want_primary = true;
                            if (!a.path.trace_rays) {
                                const uint32_t row = j / a.path.width;
                                const uint32_t ix = j - row * a.path.width;
                                const uint32_t iy = a.path.rows[row];
                                px_tid = iy * a.path.width + ix;
                                px_sample = 0u; px_rng = rng_load(a.path.states + px_tid);
                                px_first = a.path.continue_sum == 0u;         // later launches of a long render continue the pixel's sum
                                if (!px_first) { const float4 acc = a.path.accum[px_local]; px_ax = acc.x; px_ay = acc.y; px_az = acc.z; }
                                if (a.path.slice_cost) px_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
                                const V3 pd = primary_direction(ix, iy, a.path.width, a.path.height, a.path.U, a.path.V, a.path.W);
                                px_pdx = pd.x; px_pdy = pd.y; px_pdz = pd.z;
                            }
                        }
                    }
                }
                if (!alive && want_primary && a.path.trace_rays) {
                    const RayRec r = a.path.trace_rays[px_local];
                    px_depth = a.path.trace_any ? kRayTraceDepth : 1u;      // any-hit queries take the depth-limit ray's early exit
                    start_ray(mk3(r.o.x, r.o.y, r.o.z), mk3(r.d.x, r.d.y, r.d.z));
                } else if (!alive && want_primary) {
                    px_depth = 1u;
                    V3 pd = mk3(px_pdx, px_pdy, px_pdz);
                    if (HAS_SPHERES) {      // the sphere build is at its register limit (4 waves per SIMD): recompute instead of keeping
                        const uint32_t iy = px_tid / a.path.width, ix = px_tid - iy * a.path.width;
                        pd = primary_direction(ix, iy, a.path.width, a.path.height, a.path.U, a.path.V, a.path.W);
                    }
                    start_ray(mk3(a.path.center[0], a.path.center[1], a.path.center[2]), pd);
                }
            }
        } else
        // ---- refill idle lanes from the wave's slice ----
        if (!exhausted && (n_idle >= (uint32_t)a.refill_threshold || idle == ~0ull)) {
            if (wbeg >= wend) {
                for (uint32_t k = kstart; k < kFetchShards && wbeg >= wend; ++k) {
                    const uint32_t shard = (home_shard + k) & (kFetchShards - 1);
                    uint32_t c = 0;
                    if (tx == 0u) c = atomicAdd(a.fetch_counter + shard * kFetchShardStride, 1u);
                    c = (uint32_t)__shfl((int)c, 0);
                    const uint64_t beg = ((uint64_t)c * kFetchShards + shard) * (uint64_t)a.fetch_chunk;
                    if (beg < (uint64_t)n_rays) {
                        wbeg = (uint32_t)beg;
                        wend = (uint32_t)(beg + a.fetch_chunk < (uint64_t)n_rays ? beg + a.fetch_chunk : (uint64_t)n_rays);
                    } else kstart = k + 1;             // this shard is drained for good
                }
                if (wbeg >= wend) exhausted = true;
            }
            if (!exhausted) {
                const uint32_t take = n_idle < wend - wbeg ? n_idle : wend - wbeg;
                const uint32_t rank = lane_prefix(idle);
                const uint32_t mine = wbeg + rank;
                wbeg += take;
                if (!alive && rank < take) {
                    in_b = mine >= n_a;
                    any = (in_b ? a.seg[1].any_hit : a.seg[0].any_hit) != 0u;
                    const RayRec r = in_b ? a.seg[1].rays[mine - n_a] : a.seg[0].rays[mine];
                    s.ox = r.o.x; s.oy = r.o.y; s.oz = r.o.z;
                    s.dx = r.d.x; s.dy = r.d.y; s.dz = r.d.z;
                    s.idx = safe_rcp_dir<COUNT>(s.dx); s.idy = safe_rcp_dir<COUNT>(s.dy); s.idz = safe_rcp_dir<COUNT>(s.dz);
                    const uint32_t oct = (s.dx < 0.0f ? 4u : 0u) | (s.dy < 0.0f ? 2u : 0u) | (s.dz < 0.0f ? 1u : 0u);
                    s.oct_inv4 = (7u - oct) * 0x01010101u;
                    s.bt = tmax_ray; s.bu = 0.0f; s.bv = 0.0f; s.bprim = kMissPrim; s.binst = kMissPrim;
                    s.cur = make_uint2(0u, 0x80000000u);
                    s.ptri = make_uint2(0u, 0u);
                    s.sp = 0; s.base = 0;
                    s.slot = in_b ? mine - n_a : mine;
                    alive = true; has_prim = false; pidx = 0u;
                    advance_select();                   // the root becomes this lane's next node
                }
            }
        }
        if (__ballot(alive) == 0ull) break;     // queue drained and every lane finished (fused: nothing waits after a full regeneration)

        // ---- traverse until enough lanes have finished to make a refill worthwhile ----
        for (;;) {
            // ---- tail: the queue is drained, lanes are idle and a few long rays remain.  A busy lane
            //      gives the BOTTOM entry of its stack (the largest pending subtree) to an idle lane
            //      of the same wave, which continues with a copy of the ray and of the best hit so
            //      far; results are merged through an LDS mailbox when the pieces finish. ----
            if (!COUNT && !FUSED && a.tail_split && exhausted) {
                const uint64_t idle2 = __ballot(!alive);
                const uint64_t donors = __ballot(alive && s.sp > s.base);
                const uint32_t n_idle2 = (uint32_t)__popcll(idle2), n_don = (uint32_t)__popcll(donors);
                const uint32_t n_pairs = n_idle2 < n_don ? n_idle2 : n_don;
                if (n_pairs) {
                    const uint32_t drank = lane_prefix(donors), irank = lane_prefix(idle2);
                    const bool is_donor = alive && s.sp > s.base && drank < n_pairs;
                    const bool is_recv = !alive && irank < n_pairs;
                    uint2 give = make_uint2(0u, 0u);
                    if (is_donor) {
                        give = s.base < kLdsStack ? s_stack[s.base][tx] : spill[s.base - kLdsStack];
                        ++s.base;
                        if (!shared) {
                            shared = true; home = tx;
                            s_mb_t[tx] = s.bt; s_mb_u[tx] = s.bu; s_mb_v[tx] = s.bv; s_mb_prim[tx] = s.bprim; s_mb_inst[tx] = s.binst;
                            s_mb_pending[tx] = 2u;
                        } else atomicAdd(&s_mb_pending[home], 1u);
                        s_pair[drank] = tx;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int src = is_recv ? (int)s_pair[irank] : (int)tx;
                    // every lane shuffles; only receivers keep what they read
                    const float r_ox = __shfl(s.ox, src), r_oy = __shfl(s.oy, src), r_oz = __shfl(s.oz, src);
                    const float r_dx = __shfl(s.dx, src), r_dy = __shfl(s.dy, src), r_dz = __shfl(s.dz, src);
                    const float r_ix = __shfl(s.idx, src), r_iy = __shfl(s.idy, src), r_iz = __shfl(s.idz, src);
                    const float r_bt = __shfl(s.bt, src), r_bu = __shfl(s.bu, src), r_bv = __shfl(s.bv, src);
                    const uint32_t r_bp = (uint32_t)__shfl((int)s.bprim, src), r_bi = (uint32_t)__shfl((int)s.binst, src);
                    const uint32_t r_oct = (uint32_t)__shfl((int)s.oct_inv4, src), r_slot = (uint32_t)__shfl((int)s.slot, src);
                    const uint32_t r_home = (uint32_t)__shfl((int)home, src);
                    const uint32_t r_flags = (uint32_t)__shfl((int)((any ? 1u : 0u) | (in_b ? 2u : 0u)), src);
                    const uint32_t r_gx = (uint32_t)__shfl((int)give.x, src), r_gy = (uint32_t)__shfl((int)give.y, src);
                    if (is_recv) {
                        s.ox = r_ox; s.oy = r_oy; s.oz = r_oz; s.dx = r_dx; s.dy = r_dy; s.dz = r_dz;
                        s.idx = r_ix; s.idy = r_iy; s.idz = r_iz;
                        s.bt = r_bt; s.bu = r_bu; s.bv = r_bv; s.bprim = r_bp; s.binst = r_bi;
                        s.oct_inv4 = r_oct; s.slot = r_slot;
                        if (r_gy > 0x00ffffffu) { s.cur = make_uint2(r_gx, r_gy); s.ptri = make_uint2(0u, 0u); }
                        else { s.cur = make_uint2(0u, 0u); s.ptri = make_uint2(r_gx, r_gy); }
                        s.sp = 0; s.base = 0;
                        home = r_home; shared = true; alive = true; has_prim = false; pidx = 0u;
                        any = (r_flags & 1u) != 0u; in_b = (r_flags & 2u) != 0u;
                        advance_select();
                    }
                }
            }

            // ---- G. fetch what every lane needs next: primitives first, nodes second ----
            f32x4 rpa, rpb, rpc;                      // !DMA: the lane's own primitive / node in registers
            u32x4 rn0, rn1, rn2, rn3, rn4;
            if (DMA) {
                const void *pp[3], *np[5];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    pp[i] = prim_bytes + (size_t)(uint32_t)__shfl((int)pidx, (int)gp_own[i]) * a.prim_stride + gp_off[i];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    np[i] = node_bytes + (size_t)(uint32_t)__shfl((int)nidx, (int)gn_own[i]) * a.node_stride + gn_off[i];
                gather_prim_pieces(prim_lds, pp[0], pp[1], pp[2]);
                gather_node_pieces(node_lds, np[0], np[1], np[2], np[3], np[4]);
            } else {
                issue_prim_loads(prim_bytes + (size_t)pidx * a.prim_stride, rpa, rpb, rpc);
                issue_node_loads(node_bytes + (size_t)nidx * a.node_stride, rn0, rn1, rn2, rn3, rn4);
            }

            bool done = false;
            if (any && alive && shared && s_mb_prim[home] != kMissPrim) done = true;   // another piece already found a hit

#ifdef HRT_LANE_STATS
            { const uint64_t mp = __ballot(alive && has_prim), mn = __ballot(alive && has_node); ++ls_iter; ls_alive += __popcll(__ballot(alive)); ls_node += __popcll(mn); ls_prim += __popcll(mp); ls_ppass += mp != 0ull; }
#endif
            // ---- C. leaf test: waits for the primitive pieces only ----
            if (DMA) wait_prim_gather(); else wait_prim_loads(rpa, rpb, rpc);
            if (!COUNT && alive && !done && has_prim) {
                float4 pa, pb, pc;
                if (DMA) {
                    pa = reinterpret_cast<const float4 *>(s_prim_stage)[3 * tx + 0];
                    pb = reinterpret_cast<const float4 *>(s_prim_stage)[3 * tx + 1];
                    pc = reinterpret_cast<const float4 *>(s_prim_stage)[3 * tx + 2];
                } else {
                    pa = make_float4(rpa.x, rpa.y, rpa.z, rpa.w); pb = make_float4(rpb.x, rpb.y, rpb.z, rpb.w);
                    pc = make_float4(rpc.x, rpc.y, rpc.z, rpc.w);
                }
                const bool better = test_prim<HAS_SPHERES>(pa, pb, pc, s, tmin, tmax_ray, a.inst_inv, a.inst_identity);
                if (any && better) done = true;
            }
            has_prim = false; pidx = 0u;

            // ---- A. node phase ----
            uint2 tri = make_uint2(0u, 0u);
            if (DMA) wait_node_gather(); else wait_node_loads(rn0, rn1, rn2, rn3, rn4);
            if (alive && !done && has_node) {
                uint4 n0, n1, n2, n3, n4;
                if (DMA) {
                    n0 = s_node_stage[5 * tx + 0]; n1 = s_node_stage[5 * tx + 1]; n2 = s_node_stage[5 * tx + 2];
                    n3 = s_node_stage[5 * tx + 3]; n4 = s_node_stage[5 * tx + 4];
                } else {
                    n0 = make_uint4(rn0.x, rn0.y, rn0.z, rn0.w); n1 = make_uint4(rn1.x, rn1.y, rn1.z, rn1.w);
                    n2 = make_uint4(rn2.x, rn2.y, rn2.z, rn2.w); n3 = make_uint4(rn3.x, rn3.y, rn3.z, rn3.w);
                    n4 = make_uint4(rn4.x, rn4.y, rn4.z, rn4.w);
                }
                if (COUNT) { if (in_b) ++cnt_nodes_b; else ++cnt_nodes; }
                const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
                const uint32_t e_imask = n0.w;
                const float aix = __uint_as_float((e_imask & 0xffu) << 23) * s.idx;
                const float aiy = __uint_as_float(((e_imask >> 8) & 0xffu) << 23) * s.idy;
                const float aiz = __uint_as_float(((e_imask >> 16) & 0xffu) << 23) * s.idz;
                const float aox = (px - s.ox) * s.idx, aoy = (py - s.oy) * s.idy, aoz = (pz - s.oz) * s.idz;
                const bool nx = s.dx < 0.0f, ny = s.dy < 0.0f, nz = s.dz < 0.0f;
                uint32_t hitmask = 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t meta4 = h ? n1.w : n1.z;
                    const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
                    const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
                    const uint32_t bit_index4 = (meta4 ^ (s.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
                    const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
                    const uint32_t qlox = h ? n2.y : n2.x, qloy = h ? n2.w : n2.z, qloz = h ? n3.y : n3.x;
                    const uint32_t qhix = h ? n3.w : n3.z, qhiy = h ? n4.y : n4.x, qhiz = h ? n4.w : n4.z;
                    const uint32_t xn = nx ? qhix : qlox, xf = nx ? qlox : qhix;
                    const uint32_t yn = ny ? qhiy : qloy, yf = ny ? qloy : qhiy;
                    const uint32_t zn = nz ? qhiz : qloz, zf = nz ? qloz : qhiz;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float tnx = fmaf(HRT_BYTE_F(xn, j), aix, aox), tfx = fmaf(HRT_BYTE_F(xf, j), aix, aox);
                        const float tny = fmaf(HRT_BYTE_F(yn, j), aiy, aoy), tfy = fmaf(HRT_BYTE_F(yf, j), aiy, aoy);
                        const float tnz = fmaf(HRT_BYTE_F(zn, j), aiz, aoz), tfz = fmaf(HRT_BYTE_F(zf, j), aiz, aoz);
                        const float tlo = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                        const float thi = fminf(fminf(tfx, tfy), fminf(tfz, s.bt));
                        const uint32_t cb = (child_bits4 >> (8 * j)) & 0xffu;
                        const uint32_t bi = (bit_index4 >> (8 * j)) & 0xffu;
                        // conservative: the builder pads and rounds the child boxes outwards
                        if (tlo <= thi) hitmask |= cb << bi;
                    }
                }
                s.cur = make_uint2(n1.x, (hitmask & 0xff000000u) | (e_imask >> 24));
                tri = make_uint2(n1.y, hitmask & 0x00ffffffu);
            }
            has_node = false; nidx = 0u;

            // COUNT builds walk in the canonical order (every leaf of a node is tested before the next
            // node is chosen, no postponing, no splitting): the counters then equal a CPU walk of the
            // same BVH bytes and define the algorithmic traffic per ray.
            if (COUNT) {
                while (alive && !done && tri.y != 0u) {
                    const uint32_t k = (uint32_t)__ffs((int)tri.y) - 1u;
                    tri.y &= tri.y - 1u;
                    if (in_b) ++cnt_prims_b; else ++cnt_prims;
                    const float4 *pp = reinterpret_cast<const float4 *>(prim_bytes + (size_t)(tri.x + k) * a.prim_stride);
                    const bool better = test_prim<HAS_SPHERES>(pp[0], pp[1], pp[2], s, tmin, tmax_ray, a.inst_inv, a.inst_identity);
                    if (any && better) done = true;
                }
            }

            // ---- B. bookkeeping: pending leaf group, the primitive and the node of the next iteration ----
            if (alive && !done && tri.y != 0u) {
                if (s.ptri.y == 0u) s.ptri = tri;
                else {
                    if (s.sp < kLdsStack) s_stack[s.sp][tx] = tri; else spill[s.sp - kLdsStack] = tri;
                    ++s.sp;
                }
            }
            {
                // leaf pass: ONE per iteration, one primitive per lane out of its pending leaf group.  It
                // is skipped (wave-uniform) while few lanes have leaf work and none depends on it.
                const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
                const bool has = alive && !done && s.ptri.y != 0u;
                const uint64_t m = __ballot(has);
                const uint64_t must = __ballot(has && s.cur.y <= 0x00ffffffu);     // no node work in hand
                if (m != 0ull && (must != 0ull || (uint32_t)__popcll(m) * 100u >= n_alive * (uint32_t)a.postpone_pct)) {
                    if (has) {
                        const uint32_t k = (uint32_t)__ffs((int)s.ptri.y) - 1u;
                        s.ptri.y &= s.ptri.y - 1u;
                        pidx = s.ptri.x + k;
                        has_prim = true;
                    }
                }
            }
            if (alive && !done) advance_select();

            // ---- finished? ----
            if (alive) {
                if (!done && !has_node && !has_prim && s.ptri.y == 0u) done = true;
                if (FUSED && done) {
                    alive = false; waiting = true; has_node = false; has_prim = false; nidx = 0u; pidx = 0u;   // shaded at the next regeneration
                } else if (done && !shared) {
                    (in_b ? a.seg[1].hit_tuvp : a.seg[0].hit_tuvp)[s.slot] = make_float4(s.bt, s.bu, s.bv, __uint_as_float(s.bprim));
                    (in_b ? a.seg[1].hit_inst : a.seg[0].hit_inst)[s.slot] = s.binst;
                    alive = false; has_node = false; has_prim = false; nidx = 0u; pidx = 0u;
                }
            }
            // pieces of split rays finish one lane at a time: merge into the home mailbox, the last one writes
            uint64_t fin = __ballot(alive && done && shared);
            while (fin) {
                const uint32_t l = (uint32_t)__ffsll((long long)fin) - 1u;
                fin &= fin - 1ull;
                if (tx == l) {
                    const float mt = s_mb_t[home];
                    const uint64_t mid = ((uint64_t)s_mb_inst[home] << 32) | s_mb_prim[home];
                    const uint64_t id = ((uint64_t)s.binst << 32) | s.bprim;
                    const bool better = any ? (s.bprim != kMissPrim && s_mb_prim[home] == kMissPrim)
                                                : (s.bt < mt || (s.bt == mt && id < mid));
                    if (better) { s_mb_t[home] = s.bt; s_mb_u[home] = s.bu; s_mb_v[home] = s.bv; s_mb_prim[home] = s.bprim; s_mb_inst[home] = s.binst; }
                    const uint32_t pend = s_mb_pending[home] - 1u;
                    s_mb_pending[home] = pend;
                    if (pend == 0u) {
                        (in_b ? a.seg[1].hit_tuvp : a.seg[0].hit_tuvp)[s.slot] = make_float4(s_mb_t[home], s_mb_u[home], s_mb_v[home], __uint_as_float(s_mb_prim[home]));
                        (in_b ? a.seg[1].hit_inst : a.seg[0].hit_inst)[s.slot] = s_mb_inst[home];
                    }
                    alive = false; shared = false; home = tx; has_node = false; has_prim = false; nidx = 0u; pidx = 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint64_t act = __ballot(alive);
            if (act == 0ull) break;
            if ((FUSED || !exhausted) && (64u - (uint32_t)__popcll(act)) >= (uint32_t)a.refill_threshold) break;
        }
    }
#ifdef HRT_LANE_STATS
    if (FUSED && tx == 0u) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.path.rays_closest);
        atomicAdd(d + 6, ls_iter); atomicAdd(d + 7, ls_alive); atomicAdd(d + 8, ls_node); atomicAdd(d + 9, ls_prim); atomicAdd(d + 2, ls_ppass); atomicAdd(d + 3, ls_regen);
    }
#endif
    if (FUSED) {
        for (int off = 32; off > 0; off >>= 1) {
            px_rays_closest += (uint32_t)__shfl_down((int)px_rays_closest, off);
            px_rays_any += (uint32_t)__shfl_down((int)px_rays_any, off);
        }
        if (tx == 0u) {
            atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_closest), (unsigned long long)px_rays_closest);
            atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_any), (unsigned long long)px_rays_any);
        }
    }
    if (COUNT) {
        // wave-level reduction, one atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            cnt_nodes += (uint32_t)__shfl_down((int)cnt_nodes, off);
            cnt_prims += (uint32_t)__shfl_down((int)cnt_prims, off);
            cnt_nodes_b += (uint32_t)__shfl_down((int)cnt_nodes_b, off);
            cnt_prims_b += (uint32_t)__shfl_down((int)cnt_prims_b, off);
        }
        if (tx == 0u) {
            atomicAdd(reinterpret_cast<unsigned long long *>(a.seg[0].count_nodes), (unsigned long long)cnt_nodes);
            atomicAdd(reinterpret_cast<unsigned long long *>(a.seg[0].count_prims), (unsigned long long)cnt_prims);
            if (a.seg[1].rays) {
                atomicAdd(reinterpret_cast<unsigned long long *>(a.seg[1].count_nodes), (unsigned long long)cnt_nodes_b);
                atomicAdd(reinterpret_cast<unsigned long long *>(a.seg[1].count_prims), (unsigned long long)cnt_prims_b);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// bin hits by closest-hit program: SBT dispatch as a wave-ballot compaction.
// bin 0 = path ends here (miss, or any hit at depth >= rayTraceDepth), bin 1+P = program P.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bin_hits(BinArgs a) {
    // Each block owns one contiguous slice of the queue: pass 1 counts its rays per bin (wave
    // ballots), ONE atomic per bin per block reserves the output ranges, pass 2 writes the ray
    // indices with ballot prefix counts.  (One atomic per wave on a single word serialised at
    // ~88 atomics/us and cost 0.45 ms per launch.)
    __shared__ uint32_t s_wave_cnt[4][kNumBins];
    __shared__ uint32_t s_base[kNumBins];
    const uint32_t n = a.n_rays_ptr ? (a.n_rays_ptr[0] + a.n_rays_ptr[1] + a.n_rays_ptr[2] + a.n_rays_ptr[3]) : a.n_rays;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(a.total_rays), (unsigned long long)n);
    const uint32_t per_block = ((n + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
    const uint64_t begin64 = (uint64_t)blockIdx.x * per_block;
    if (begin64 >= n) return;
    const uint32_t begin = (uint32_t)begin64;
    const uint32_t end = begin + per_block < n ? begin + per_block : n;
    const uint32_t wave = threadIdx.x >> 6;

    auto bin_of = [&](uint32_t i) -> uint32_t {
        if (i >= end) return 0xffu;
        const uint32_t inst = a.hit_inst[i];
        return (inst == kMissPrim || a.depth >= kRayTraceDepth) ? 0u : 1u + a.inst_program[inst];
    };

    uint32_t cnt[kNumBins];
#pragma unroll
    for (uint32_t b = 0; b < kNumBins; ++b) cnt[b] = 0;
    for (uint32_t base = begin; base < end; base += 256u) {
        const uint32_t bin = bin_of(base + threadIdx.x);
#pragma unroll
        for (uint32_t b = 0; b < kNumBins; ++b) cnt[b] += (uint32_t)__popcll(__ballot(bin == b));
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (uint32_t b = 0; b < kNumBins; ++b) s_wave_cnt[wave][b] = cnt[b];
    }
    __syncthreads();
    if (threadIdx.x < kNumBins) {
        const uint32_t tot = s_wave_cnt[0][threadIdx.x] + s_wave_cnt[1][threadIdx.x] + s_wave_cnt[2][threadIdx.x] + s_wave_cnt[3][threadIdx.x];
        s_base[threadIdx.x] = tot ? atomicAdd(&a.bin_count[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    // running write offsets, per bin: block base + everything earlier waves / earlier chunks wrote
    uint32_t run[kNumBins];
#pragma unroll
    for (uint32_t b = 0; b < kNumBins; ++b) run[b] = s_base[b];
    for (uint32_t base = begin; base < end; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t bin = bin_of(i);
        uint64_t m[kNumBins];
#pragma unroll
        for (uint32_t b = 0; b < kNumBins; ++b) m[b] = __ballot(bin == b);
        __syncthreads();                       // previous chunk's s_wave_cnt fully consumed
        if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
            for (uint32_t b = 0; b < kNumBins; ++b) s_wave_cnt[wave][b] = (uint32_t)__popcll(m[b]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t b = 0; b < kNumBins; ++b) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) { const uint32_t c = s_wave_cnt[w][b]; total += c; if (w < wave) before += c; }
            if (bin == b) a.bin_items[(size_t)b * a.bin_stride + run[b] + before + lane_prefix(m[b])] = i;
            run[b] += total;
        }
    }
}

// ---------------------------------------------------------------------------------------
// shade: closesthitImpl for one (geometry, material) program, shader/Shader.cu:108-233
// ---------------------------------------------------------------------------------------
template <int PROGRAM>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) {
    constexpr bool kSphere = PROGRAM == kProgramSphereRough || PROGRAM == kProgramSphereMetal;
    constexpr bool kRough = PROGRAM == kProgramSphereRough || PROGRAM == kProgramTriangleRough;
    const uint32_t n = a.bin_count[1 + PROGRAM];
    uint32_t out_base = 0;                                  // programs are laid out one after another
    for (int p = 0; p < PROGRAM; ++p) out_base += a.bin_count[1 + p];
    const uint32_t *items = a.bin_items + (size_t)(1 + PROGRAM) * a.bin_stride;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) {
        const uint32_t i = items[q];
        const RayRec ray = a.rays_in[i];
        const float4 hit = a.hit_tuvp[i];
        const uint32_t inst = a.hit_inst[i];
        const uint32_t local = __float_as_uint(ray.o.w), tid = __float_as_uint(ray.d.w);
        const HitGroup hg = a.hitgroups[inst];               // SBT record of the instance, :108

        const V3 rayOrigin = mk3(ray.o.x, ray.o.y, ray.o.z), rayDirection = mk3(ray.d.x, ray.d.y, ray.d.z);
        RngState *state = a.states + tid;                                  // params.stateArray + tid
        const bool draws = program_draws(PROGRAM, hg);
        Xorwow rng{};
        if (draws) rng = rng_load(state);
        V3 hitPoint, reflectDirection;
        scatter<kSphere, kRough>(hg, rayOrigin, rayDirection, hit.x /* optixGetRayTmax :111 */, hit.y, hit.z,
                                 __float_as_uint(hit.w) /* optixGetPrimitiveIndex :117 */, rng, hitPoint, reflectDirection);
        if (draws) rng_store(state, rng);

        a.chain[(size_t)local * 4 + (a.depth - 1u)] = inst;               // albedo applied on the way back, :236-238
        RayRec nr;                                                        // recursive rayTrace, :230-233
        nr.o = make_float4(hitPoint.x, hitPoint.y, hitPoint.z, ray.o.w);
        nr.d = make_float4(reflectDirection.x, reflectDirection.y, reflectDirection.z, ray.d.w);
        a.rays_out[out_base + q] = nr;
    }
}

// ---------------------------------------------------------------------------------------
// accumulate: paths that end at this depth.  __miss__ (Shader.cu:276-287) or the depth
// cut-off (:102-107), then the albedo products of the unwinding recursion (:236-238),
// innermost bounce first.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_accumulate(AccumArgs a) {
    const uint32_t n = a.bin_count[0];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) {
        const uint32_t i = a.bin_items[q];
        const uint32_t local = __float_as_uint(a.rays_in[i].o.w);
        const bool miss = a.hit_inst[i] == kMissPrim;
        const uint4 ch = reinterpret_cast<const uint4 *>(a.chain)[local];
        const uint32_t chain[4] = {ch.x, ch.y, ch.z, ch.w};
        const V3 r = fold_chain(miss, a.bg, chain, a.depth, a.hitgroups);
        const float rx = r.x, ry = r.y, rz = r.z;
        // every pixel ends exactly once per sample: a plain store.  k_sum adds the samples in sample order.
        a.result[local] = make_float4(rx, ry, rz, 0.0f);
    }
}

// accum (+)= result, once per sample and strictly in sample order (bit-exact mean): the stages of two
// consecutive samples overlap in time, their terminations must not be added in stage order.
__global__ __launch_bounds__(256) void k_sum(float4 *accum, const float4 *result, uint32_t n, uint32_t first_sample) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = result[i];
    if (first_sample) accum[i] = make_float4(r.x, r.y, r.z, 0.0f);
    else { float4 acc = accum[i]; acc.x += r.x; acc.y += r.y; acc.z += r.z; accum[i] = acc; }
}

// ---------------------------------------------------------------------------------------
// finalize: colorToFloat4(mean) and the (always zero, quirk Q3) AOVs, Shader.cu:270-272
// ---------------------------------------------------------------------------------------
// powf(cx, 1/2.4f) of DeviceFunctions.cuh:196-198 is pinned as the correctly rounded float of cx^y: srgb_pow.h
__device__ __forceinline__ float srgb_channel(float c) {
    const float cx = fmaxf(0.0f, fminf(c, 1.0f));
    const float px = pow_inv_gamma(cx);
    const float sx = cx < 0.0031308f ? 12.92f * cx : 1.055f * px - 0.055f;
    return fmaxf(0.0f, fminf(sx, 1.0f));
}

__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    // (the path kernel that ran before this one has used up its slice counters: zero again for the next launch, which then needs no
    // memset of its own in front of it -- one operation less per frame of the reference's loop)
    if (a.reset_counters && j < a.n_reset) a.reset_counters[j] = 0u;
    if (j >= a.n_tile_pixels) return;
    const uint32_t row = j / a.width;
    const uint32_t ix = j - row * a.width;
    const uint32_t p = a.rows[row] * a.width + ix;
    float4 acc = a.accum[j];
    if (a.spp > 1u) { const float n = (float)a.spp; acc.x /= n; acc.y /= n; acc.z /= n; }
    if (a.linear) a.linear[p] = make_float4(acc.x, acc.y, acc.z, 1.0f);
    a.color[p] = make_float4(srgb_channel(acc.x), srgb_channel(acc.y), srgb_channel(acc.z), 1.0f);
    if (a.albedo) a.albedo[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.normal) a.normal[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// convertFloat4ToUchar4Kernel, src/Global/RendererImpl.cu:672-678 (colorToUchar4, DeviceFunctions.cuh:153-183)
__global__ __launch_bounds__(256) void k_to_rgba8(const float4 *src, uchar4 *dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = src[i];
    const float s[3] = {srgb_channel(c.x), srgb_channel(c.y), srgb_channel(c.z)};
    uchar4 o;
    uint32_t q;
    q = (uint32_t)(s[0] * 256.0f); o.x = (unsigned char)(q < 255u ? q : 255u);
    q = (uint32_t)(s[1] * 256.0f); o.y = (unsigned char)(q < 255u ? q : 255u);
    q = (uint32_t)(s[2] * 256.0f); o.z = (unsigned char)(q < 255u ? q : 255u);
    o.w = 255u;
    dst[i] = o;
}

// colorToFloat4 over an array (DeviceFunctions.cuh:188-209): what raygen applies to its result (Shader.cu:270)
__global__ __launch_bounds__(256) void k_color_to_float4(const float4 *src, float4 *dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = src[i];
    dst[i] = make_float4(srgb_channel(c.x), srgb_channel(c.y), srgb_channel(c.z), 1.0f);
}

// ---------------------------------------------------------------------------------------
// helpers for hrt_trace_rays (parity tests run the production traverse kernel on their own rays)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_rays(const float *o, const float *d, uint32_t n, RayRec *rays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RayRec r;
    r.o = make_float4(o[3 * (size_t)i], o[3 * (size_t)i + 1], o[3 * (size_t)i + 2], __uint_as_float(i));
    r.d = make_float4(d[3 * (size_t)i], d[3 * (size_t)i + 1], d[3 * (size_t)i + 2], __uint_as_float(i));
    rays[i] = r;
}
__global__ __launch_bounds__(256) void k_unpack_hits(const float4 *tuvp, const uint32_t *inst, uint32_t n,
                                                     float *t, float *u, float *v, uint32_t *prim, uint32_t *oinst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = tuvp[i];
    t[i] = h.x; u[i] = h.y; v[i] = h.z; prim[i] = __float_as_uint(h.w); oinst[i] = inst[i];
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void launch_rng_init(RngState *states, uint32_t n, uint64_t salt, const uint32_t *d_jump, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_rng_init, dim3(ceil_div(n, 256)), dim3(256), 0, s, states, n, salt, d_jump);
}
void launch_generate(const GenerateArgs &a, hipStream_t s) {
    if (a.n_tile_pixels) hipLaunchKernelGGL(k_generate, dim3(ceil_div(a.n_tile_pixels, 256)), dim3(256), 0, s, a);
}
void launch_traverse(const TraverseArgs &a, bool count, bool has_spheres, bool dma, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    const int sel = (dma ? 4 : 0) | (count ? 2 : 0) | (has_spheres ? 1 : 0);
    switch (sel) {
        case 0: hipLaunchKernelGGL((k_traverse<false, false, false, false>), g, b, 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_traverse<false, true, false, false>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_traverse<true, false, false, false>), g, b, 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_traverse<true, true, false, false>), g, b, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_traverse<false, false, true, false>), g, b, 0, s, a); break;
        case 5: hipLaunchKernelGGL((k_traverse<false, true, true, false>), g, b, 0, s, a); break;
        case 6: hipLaunchKernelGGL((k_traverse<true, false, true, false>), g, b, 0, s, a); break;
        default: hipLaunchKernelGGL((k_traverse<true, true, true, false>), g, b, 0, s, a); break;
    }
}
// fused path mode: one launch renders every sample of every pixel of the tile
void launch_paths_v1(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    if (has_spheres) hipLaunchKernelGGL((k_traverse<false, true, false, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_traverse<false, false, false, true>), g, b, 0, s, a);
}
void launch_bin(const BinArgs &a, uint32_t grid_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_bin_hits, dim3(grid_blocks), dim3(256), 0, s, a);
}
void launch_shade(const ShadeArgs &a, int program, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(256);
    switch (program) {
        case 0: hipLaunchKernelGGL((k_shade<0>), g, b, 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_shade<1>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_shade<2>), g, b, 0, s, a); break;
        default: hipLaunchKernelGGL((k_shade<3>), g, b, 0, s, a); break;
    }
}
void launch_accumulate(const AccumArgs &a, uint32_t grid_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_blocks), dim3(256), 0, s, a);
}
void launch_sum(float4 *accum, const float4 *result, uint32_t n, uint32_t first_sample, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_sum, dim3(ceil_div(n, 256)), dim3(256), 0, s, accum, result, n, first_sample);
}
void launch_finalize(const FinalizeArgs &a, hipStream_t s) {
    if (a.n_tile_pixels) hipLaunchKernelGGL(k_finalize, dim3(ceil_div(a.n_tile_pixels, 256)), dim3(256), 0, s, a);
}
void launch_to_rgba8(const float4 *src, uchar4 *dst, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_to_rgba8, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, dst, n);
}
void launch_color_to_float4(const float4 *src, float4 *dst, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_color_to_float4, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, dst, n);
}
void launch_pack_rays(const float *o, const float *d, uint32_t n, RayRec *rays, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pack_rays, dim3(ceil_div(n, 256)), dim3(256), 0, s, o, d, n, rays);
}
void launch_unpack_hits(const float4 *tuvp, const uint32_t *inst, uint32_t n, float *t, float *u, float *v,
                        uint32_t *prim, uint32_t *oinst, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_unpack_hits, dim3(ceil_div(n, 256)), dim3(256), 0, s, tuvp, inst, n, t, u, v, prim, oinst);
}

}  // namespace hrt
