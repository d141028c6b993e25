// fused.hip -- k_fused, the production kernel of hrt_render_launch: the whole render in ONE persistent launch.
//
// Replaces the OptiX launch of the reference (raygen -> optixTrace -> closest-hit / miss, recursively, shader/Shader.cu:46-287;
// launched at src/Global/RendererMesh.cu:416-419).  Same execution model as round 1's fused mode of k_traverse (kernels.hip,
// still there as HRT_FUSED=2): every lane OWNS a pixel and carries its path state in registers -- RNG state, sample and depth
// counters, the albedo chain, the running sum; waves take 16-pixel slices of the tile from sharded counters; a lane whose ray
// has finished waits, and once enough lanes of the wave wait a regeneration phase (wave-uniform branch) shades them in place
// with the shared device functions of trav_common.h and starts the next ray in the same lane: the bounce, the next sample's
// primary ray, or the next pixel.  No ray queues, no hit records, no stage barriers.
// What is new is the traversal step between two regenerations: trav_lean.h, written for instruction count (the SIMD issues one
// instruction of any kind per ~2.4 cycles and this kernel is bound by that: DESIGN.md section 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "trav_common.h"
#include "trav_lean.h"

#pragma clang fp contract(off)

namespace hrt {

template <bool HAS_SPHERES>
#ifndef HRT_FUSED_WAVES_PER_SIMD
#define HRT_FUSED_WAVES_PER_SIMD 4      // 125 VGPRs and no spills: 3122 Mrays/s on C4; 5 waves (96 VGPRs, 95 spilled around the shading) 2560
#endif
__global__ __launch_bounds__(kTraverseBlock, HRT_FUSED_WAVES_PER_SIMD) void k_fused(TraverseArgs a) {
    static_assert(kTraverseBlock == 64, "one wave per workgroup: the stacks are per wave");
    __shared__ uint2 s_nodes[kNodeStackLds][kTraverseBlock];     // sibling groups: one per tree level (hrt_api.cpp sends deeper trees to k_traverse)
    __shared__ uint2 s_leaves[kLeafStackLds][kTraverseBlock];    // leaf groups

    const uint32_t n_pixels = a.path.n_tile_pixels;
    const char *__restrict__ node_bytes = reinterpret_cast<const char *>(a.nodes);
    const char *__restrict__ prim_bytes = reinterpret_cast<const char *>(a.prims);
    const float tmin = a.tmin, tmax_ray = a.tmax;
    const uint32_t tx = threadIdx.x;
    const uint32_t ldsn = (uint32_t)reinterpret_cast<uintptr_t>(&s_nodes[0][tx]), ldsl = (uint32_t)reinterpret_cast<uintptr_t>(&s_leaves[0][tx]);

    LeanLane L;
    lean_reset(L);
    L.s.bt = tmax_ray; L.s.bu = 0.0f; L.s.bv = 0.0f; L.s.bprim = kMissPrim; L.s.binst = kMissPrim;
    L.s.ox = L.s.oy = L.s.oz = 0.0f; L.s.dx = L.s.dy = 0.0f; L.s.dz = 1.0f; L.s.idx = L.s.idy = L.s.idz = 1.0f; L.s.oct_inv4 = 0u;
    bool alive = false;                     // a ray is being traversed in this lane
    bool waiting = false;                   // ... has finished and waits for the next regeneration
    bool any = false;                       // this lane's ray only needs to know whether anything is hit
    bool exhausted = false;                 // wave-uniform: no pixels left to start

    // the lane's pixel
    bool have_pixel = false, px_first = true;
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

    // the wave's slice of the tile: [wbeg, wend); slices of fetch_chunk pixels are handed out by kFetchShards counters
    uint32_t wbeg = 0, wend = 0, kstart = 0;
    const uint32_t home_shard = blockIdx.x & (kFetchShards - 1);

    for (;;) {
        const uint64_t idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        // ---- regenerate: shade finished rays in place, start the next sample / pixel ----
        if (n_idle >= (uint32_t)a.refill_threshold || idle == ~0ull) {
#ifdef HRT_LANE_STATS
            ++ls_regen;
#endif
            bool launch = false, want_primary = false;      // launch: this lane starts the ray (ro, rd) below
            V3 ro = mk3(0.0f, 0.0f, 0.0f), rd = mk3(0.0f, 0.0f, 1.0f);
            if (!alive && waiting) {
                waiting = false;
                const TravState &s = L.s;
                const bool miss = s.bprim == kMissPrim;
                if (a.path.trace_rays) {           // hrt_trace_rays on this kernel: the "pixel" is a caller's ray, its hit record the result
                    a.path.trace_tuvp[px_local] = make_float4(s.bt, s.bu, s.bv, __uint_as_float(s.bprim));
                    a.path.trace_inst[px_local] = s.binst;
                    have_pixel = false;
                } else if (miss || px_depth >= kRayTraceDepth) {
                    // the path ends: miss colour or black at the depth limit, folded through the albedo chain (Shader.cu:102-107, :236-238, :276-287)
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
                    const V3 o = mk3(s.ox, s.oy, s.oz), d = mk3(s.dx, s.dy, s.dz);
                    V3 hp, nd;
                    scatter_programs<HAS_SPHERES>(program, hg, o, d, s.bt, s.bu, s.bv, s.bprim, px_rng, hp, nd);
                    px_chain[px_depth - 1u] = inst;
                    ++px_depth;
                    ro = hp; rd = nd; launch = true;
                }
            }
            // lanes without a pixel take the next ones of the wave's slice of the tile
            const uint64_t need = __ballot(!alive && !have_pixel && !want_primary && !launch);
            if (need != 0ull && !exhausted) {
                if (wbeg >= wend) {
                    for (uint32_t k = kstart; k < kFetchShards && wbeg >= wend; ++k) {
                        const uint32_t shard = (home_shard + k) & (kFetchShards - 1);
                        uint32_t c = 0;
                        if (tx == 0u) c = atomicAdd(a.fetch_counter + shard * kFetchShardStride, 1u);
                        c = (uint32_t)__shfl((int)c, 0);
                        const uint64_t q = (uint64_t)c * kFetchShards + shard;          // the q-th slice handed out ...
                        if (q * (uint64_t)a.fetch_chunk < (uint64_t)n_pixels) {
                            // ... is slice slice_order[q] of the tile: the expensive slices first, so that the render
                            // ends on cheap pixels (longest-processing-time-first; a pixel's samples run one after the other)
                            const uint64_t beg = (a.path.slice_order ? (uint64_t)a.path.slice_order[q] : q) * (uint64_t)a.fetch_chunk;
                            wbeg = (uint32_t)beg;
                            wend = (uint32_t)(beg + a.fetch_chunk < (uint64_t)n_pixels ? beg + a.fetch_chunk : (uint64_t)n_pixels);
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
                    if (!alive && !have_pixel && !want_primary && !launch && rank < take) {
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
            if (want_primary) {
                if (a.path.trace_rays) {
                    const RayRec r = a.path.trace_rays[px_local];
                    px_depth = a.path.trace_any ? kRayTraceDepth : 1u;      // any-hit queries take the depth-limit ray's early exit
                    ro = mk3(r.o.x, r.o.y, r.o.z); rd = mk3(r.d.x, r.d.y, r.d.z);
                } else {
                    px_depth = 1u;
                    ro = mk3(a.path.center[0], a.path.center[1], a.path.center[2]); rd = mk3(px_pdx, px_pdy, px_pdz);
                }
                launch = true;
            }
            if (launch) {
                any = px_depth >= kRayTraceDepth;      // a hit at the depth limit is black whatever it is (Shader.cu:102-107)
                if (any) ++px_rays_any; else ++px_rays_closest;
                lean_start(L, ro, rd, tmax_ray);
                alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;     // the tile is used up and every lane has finished (nothing waits after a full regeneration)

        // ---- traverse until enough lanes have finished to make a regeneration worthwhile ----
        // the registers the loads land in: "defined" without an instruction (lanes that load nothing never look at theirs)
        f32x4 rpa, rpb, rpc;
        u32x4 rn0, rn1, rn2, rn3, rn4;
        asm volatile("" : "=v"(rpa), "=v"(rpb), "=v"(rpc), "=v"(rn0), "=v"(rn1), "=v"(rn2), "=v"(rn3), "=v"(rn4));
        for (;;) {
            // ---- G. fetch what the lanes need next: primitives first, nodes second -- for the lanes that need one only (the
            //      instruction slots of the loads are not saved, but their L1 / TA cycles are).  The node loads are issued even when
            //      no lane wants one: they are then ALWAYS the five youngest vector-memory operations at the primitives' wait,
            //      whose vmcnt(5) is counted by hand. ----
            const uint64_t mask_p = __ballot(L.pidx != kNoWork), mask_n0 = __ballot(L.nidx != kNoWork);
            {
                uint32_t po = L.pidx * a.prim_stride, no = L.nidx * a.node_stride;      // (garbage for kNoWork: masked out)
                asm volatile("" : "+v"(po), "+v"(no));          // both offsets before the first load
                if (mask_p != 0ull) issue_prim_loads_off(mask_p, prim_bytes, po, rpa, rpb, rpc);
                issue_node_loads_off(mask_n0, node_bytes, no, rn0, rn1, rn2, rn3, rn4);
            }
#ifdef HRT_LANE_STATS
            { ++ls_iter; ls_alive += __popcll(__ballot(alive)); ls_node += __popcll(mask_n0); ls_prim += __popcll(mask_p); ls_ppass += mask_p != 0ull; }
#endif
            // ---- C. leaf test: waits for the primitive pieces only (the node loads issued behind them stay in flight) ----
            bool hit_any = false;
            if (mask_p != 0ull) {
                wait_prim_loads(rpa, rpb, rpc);
                if (L.pidx != kNoWork) {
                    const float4 pa = make_float4(rpa.x, rpa.y, rpa.z, rpa.w), pb = make_float4(rpb.x, rpb.y, rpb.z, rpb.w),
                                 pc = make_float4(rpc.x, rpc.y, rpc.z, rpc.w);
                    const bool better = test_prim<HAS_SPHERES>(pa, pb, pc, L.s, tmin, tmax_ray, a.inst_inv, a.inst_identity);
                    hit_any = any && better;
                }
            }
            // ---- A. node step ----
            uint2 child = make_uint2(0u, 0u), tri = make_uint2(0u, 0u);
            wait_node_loads(rn0, rn1, rn2, rn3, rn4);
            if (L.nidx != kNoWork && !hit_any) lean_node(L.s, tmin, rn0, rn1, rn2, rn3, rn4, child, tri);
            // ---- B. bookkeeping (trav_lean.h: one hand-written sequence): file the new groups; the leaf pass (ONE per iteration, one
            //      primitive per lane, skipped while few lanes have leaf work and none depends on it); the primitive and the node of
            //      the next iteration; finished? ----
            if (hit_any) { lean_reset(L); alive = false; waiting = true; }     // an any-hit ray is done with its first accepted intersection
            if (alive) {
                const uint32_t fin = lean_bookkeeping_asm(L, child, tri, ldsn, ldsl, (uint32_t)a.postpone_pct, (uint32_t)a.leaf_quorum);
                if (fin != 0u) { alive = false; waiting = true; }
            }
            const uint64_t act = __ballot(alive);
            if (act == 0ull) break;
            if ((64u - (uint32_t)__popcll(act)) >= (uint32_t)a.refill_threshold) break;
        }
    }
#ifdef HRT_LANE_STATS
    if (tx == 0u) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.path.rays_closest);
        atomicAdd(d + 6, ls_iter); atomicAdd(d + 7, ls_alive); atomicAdd(d + 8, ls_node); atomicAdd(d + 9, ls_prim); atomicAdd(d + 2, ls_ppass); atomicAdd(d + 3, ls_regen);
    }
#endif
    for (int off = 32; off > 0; off >>= 1) {
        px_rays_closest += (uint32_t)__shfl_down((int)px_rays_closest, off);
        px_rays_any += (uint32_t)__shfl_down((int)px_rays_any, off);
    }
    if (tx == 0u) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_closest), (unsigned long long)px_rays_closest);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_any), (unsigned long long)px_rays_any);
    }
}

// one launch renders every sample of every pixel of the tile
void launch_fused(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    if (has_spheres) hipLaunchKernelGGL((k_fused<true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_fused<false>), g, b, 0, s, a);
}

}  // namespace hrt
