// fused_queue.hip -- k_trace_queue: the traverse kernel of the WAVEFRONT pipeline (generate / traverse / bin / shade / accumulate as
// separate launches, hrt_api.cpp) with the traversal step of the production path kernel: trav_lean.h -- two small LDS stacks, the
// hand-written bookkeeping, hand-counted vmcnt, issue priorities per phase, tail splitting with a shared best hit -- exactly the loop
// of k_fused (fused.hip), fed from ray queues instead of pixels: a lane takes a RayRec from one of up to two queue segments
// (their lengths are read from device memory, so a render is a fixed sequence of launches), traverses it, and writes the hit
// record; no path state, no RNG.  Replaces optixTrace = RT-core traversal + built-in intersection (shader/Shader.cu:70,
// src/Global/RendererImpl.cu:295-314) for the rays of one wavefront stage.  Round 1's k_traverse (kernels.hip) stays for the
// counting build, the LDS-DMA gather mode and trees deeper than k_fused's node stack.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "device_types.h"
#include "trav_common.h"
#include "trav_lean.h"

#pragma clang fp contract(off)

namespace hrt {

#ifndef HRT_PRIO_BOOK
#define HRT_PRIO_BOOK 2      // bookkeeping + address arithmetic + load issue (fused.hip has the measurements)
#define HRT_PRIO_PRIM 0      // primitive wait + test
#define HRT_PRIO_NODE 1      // node wait + slab tests
#define HRT_PRIO_REGEN 2     // hit stores, new rays
#endif

template <bool HAS_SPHERES>
__global__ __launch_bounds__(kTraverseBlock, 4) void k_trace_queue(TraverseArgs a) {
    static_assert(kTraverseBlock == 64, "one wave per workgroup: the stacks are per wave");
    __shared__ uint2 s_nodes[kNodeStackLds][kTraverseBlock];
    __shared__ uint2 s_leaves[kLeafStackLds][kTraverseBlock];
    __shared__ float s_mb_t[kTraverseBlock], s_mb_u[kTraverseBlock], s_mb_v[kTraverseBlock];
    __shared__ uint32_t s_mb_prim[kTraverseBlock], s_mb_inst[kTraverseBlock], s_mb_pending[kTraverseBlock];
    __shared__ uint32_t s_pair[kTraverseBlock];

    // up to two queue segments per launch (e.g. the depth-4 rays of sample s and the primary rays of sample s + 1): ray i < n_a
    // comes from segment 0, the others from segment 1
    const uint32_t n_a = a.seg[0].n_ptr ? (a.seg[0].n_ptr[0] + a.seg[0].n_ptr[1] + a.seg[0].n_ptr[2] + a.seg[0].n_ptr[3]) : a.seg[0].n;
    const uint32_t n_b = a.seg[1].rays ? (a.seg[1].n_ptr ? (a.seg[1].n_ptr[0] + a.seg[1].n_ptr[1] + a.seg[1].n_ptr[2] + a.seg[1].n_ptr[3]) : a.seg[1].n) : 0u;
    const uint32_t n_rays = n_a + n_b;
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
    bool waiting = false;                   // ... has finished: its hit record is written at the next refill
    bool any = false;                       // this lane's ray only needs to know whether anything is hit
    bool exhausted = false;                 // wave-uniform: the queue is used up
    bool shared = false;                    // this lane works on a piece of a ray that has been split across lanes (tail splitting)
    uint32_t home = tx;                     // ... whose owner is this lane
    uint32_t q_index = 0u;                  // the lane's ray: position in the concatenated queue

    uint32_t wbeg = 0, wend = 0, kstart = 0;
    const uint32_t home_shard = blockIdx.x & (kFetchShards - 1);

    for (;;) {
        const uint64_t idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (idle == ~0ull || (exhausted ? (uint32_t)__popcll(__ballot(waiting)) >= (uint32_t)a.tail_regen : n_idle >= (uint32_t)a.refill_threshold)) {
            __builtin_amdgcn_s_setprio(HRT_PRIO_REGEN);
            if (!alive && waiting) {
                waiting = false;
                const TravState &s = L.s;
                const bool in_b = q_index >= n_a;
                const uint32_t k = in_b ? q_index - n_a : q_index;
                float4 *tuvp = in_b ? a.seg[1].hit_tuvp : a.seg[0].hit_tuvp;
                uint32_t *hinst = in_b ? a.seg[1].hit_inst : a.seg[0].hit_inst;
                tuvp[k] = make_float4(s.bt, s.bu, s.bv, __uint_as_float(s.bprim));
                hinst[k] = s.binst;
            }
            const uint64_t need = __ballot(!alive);
            bool launch = false;
            if (need != 0ull && !exhausted) {
                if (wbeg >= wend) {
                    for (uint32_t k = kstart; k < kFetchShards && wbeg >= wend; ++k) {
                        const uint32_t shard = (home_shard + k) & (kFetchShards - 1);
                        uint32_t c = 0;
                        if (tx == 0u) c = atomicAdd(a.fetch_counter + shard * kFetchShardStride, 1u);
                        c = (uint32_t)__shfl((int)c, 0);
                        const uint64_t q = (uint64_t)c * kFetchShards + shard;
                        if (q * (uint64_t)a.fetch_chunk < (uint64_t)n_rays) {
                            const uint64_t beg = q * (uint64_t)a.fetch_chunk;
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
                    if (!alive && rank < take) { q_index = mine; launch = true; }
                }
            }
            if (launch) {
                const bool in_b = q_index >= n_a;
                const RayRec r = in_b ? a.seg[1].rays[q_index - n_a] : a.seg[0].rays[q_index];
                any = (in_b ? a.seg[1].any_hit : a.seg[0].any_hit) != 0u;
                lean_start(L, mk3(r.o.x, r.o.y, r.o.z), mk3(r.d.x, r.d.y, r.d.z), tmax_ray);
                alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;

        // ---- the traversal loop of k_fused (fused.hip), unchanged ----
        auto traverse = [&](auto tail_tag) {
            constexpr bool kTail = decltype(tail_tag)::value;
            // the registers the loads land in: "defined" without an instruction (lanes that load nothing never look at theirs)
            f32x4 rpa, rpb, rpc;
            u32x4 rn0, rn1, rn2, rn3, rn4;
            asm volatile("" : "=v"(rpa), "=v"(rpb), "=v"(rpc), "=v"(rn0), "=v"(rn1), "=v"(rn2), "=v"(rn3), "=v"(rn4));
            for (;;) {
                // ---- tail: the tile is used up, lanes have no pixel any more and a few pixels' sample chains remain, one ray after
                //      the other.  A busy lane gives the BOTTOM entry of its node stack (the largest pending subtree) to a free lane of
                //      the wave, which continues with a copy of the ray.  The pieces of a split ray share ONE best hit, the mailbox of
                //      the lane that owns the ray: a piece publishes every improvement there (canonical order: the result does not
                //      depend on who found what, or when) and adopts what the others found closer, so every piece culls with the
                //      ray's best hit so far. ----
                if (kTail && a.tail_split) {          // (one donation per busy lane and iteration: more rounds of this change nothing, r02_sweep_tile_tail.txt)
                    const bool is_free = !alive && !waiting && !shared;
                    const uint64_t free_m = __ballot(is_free);
                    const uint64_t donors = __ballot(alive && L.nsp > L.base);
                    const uint32_t n_free = (uint32_t)__popcll(free_m), n_don = (uint32_t)__popcll(donors);
                    const uint32_t n_pairs = n_free < n_don ? n_free : n_don;
                    if (n_pairs) {
                        const uint32_t drank = lane_prefix(donors), irank = lane_prefix(free_m);
                        const bool is_donor = alive && L.nsp > L.base && drank < n_pairs;
                        const bool is_recv = is_free && irank < n_pairs;
                        uint2 give = make_uint2(0u, 0u);
                        if (is_donor) {
                            give = s_nodes[L.base][tx];
                            ++L.base;
                            if (!shared) {
                                shared = true; home = tx;
                                s_mb_t[tx] = L.s.bt; s_mb_u[tx] = L.s.bu; s_mb_v[tx] = L.s.bv; s_mb_prim[tx] = L.s.bprim; s_mb_inst[tx] = L.s.binst;
                                s_mb_pending[tx] = 2u;
                            } else atomicAdd(&s_mb_pending[home], 1u);
                            s_pair[drank] = tx;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int src = is_recv ? (int)s_pair[irank] : (int)tx;
                        // every lane shuffles; only receivers keep what they read
                        TravState &s = L.s;
                        const float r_ox = __shfl(s.ox, src), r_oy = __shfl(s.oy, src), r_oz = __shfl(s.oz, src);
                        const float r_dx = __shfl(s.dx, src), r_dy = __shfl(s.dy, src), r_dz = __shfl(s.dz, src);
                        const uint32_t r_home = (uint32_t)__shfl((int)(home | (any ? 0x100u : 0u)), src);
                        const uint32_t r_gx = (uint32_t)__shfl((int)give.x, src), r_gy = (uint32_t)__shfl((int)give.y, src);
                        if (is_recv) {
                            lean_start(L, mk3(r_ox, r_oy, r_oz), mk3(r_dx, r_dy, r_dz), tmax_ray);     // the same reciprocals and octant as the owner's
                            home = r_home & 0xffu; any = (r_home & 0x100u) != 0u; shared = true; alive = true;
                            s.bt = s_mb_t[home]; s.bu = s_mb_u[home]; s.bv = s_mb_v[home]; s.bprim = s_mb_prim[home]; s.binst = s_mb_inst[home];
                            s.cur = make_uint2(r_gx, r_gy);           // a sibling group with hits: only those are pushed
                            lean_pick_node(L);                        // (replaces the root lean_start chose)
                        }
                    }
                }

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
                    __builtin_amdgcn_s_setprio(HRT_PRIO_PRIM);
                }
                // ---- C. leaf test: waits for the primitive pieces only (the node loads issued behind them stay in flight) ----
                bool hit_any = false, improved = false;
                if (mask_p != 0ull) {
                    wait_prim_loads(rpa, rpb, rpc);
                    if (L.pidx != kNoWork) {
                        const float4 pa = make_float4(rpa.x, rpa.y, rpa.z, rpa.w), pb = make_float4(rpb.x, rpb.y, rpb.z, rpb.w),
                                     pc = make_float4(rpc.x, rpc.y, rpc.z, rpc.w);
                        improved = test_prim<HAS_SPHERES>(pa, pb, pc, L.s, tmin, tmax_ray, a.inst_inv, a.inst_identity);
                        hit_any = any && improved;
                    }
                }
                if (kTail && a.tail_split) {
                    // pieces of split rays publish their improvements one lane at a time (rare: a few per ray) ...
                    uint64_t pub = __ballot(improved && shared);
                    while (pub) {
                        const uint32_t l = (uint32_t)__ffsll((long long)pub) - 1u;
                        pub &= pub - 1ull;
                        if (tx == l) {
                            const TravState &s = L.s;
                            const float mt = s_mb_t[home];
                            const uint64_t mid = ((uint64_t)s_mb_inst[home] << 32) | s_mb_prim[home];
                            const uint64_t id = ((uint64_t)s.binst << 32) | s.bprim;
                            if (any ? s_mb_prim[home] == kMissPrim : (s.bt < mt || (s.bt == mt && id < mid))) {
                                s_mb_t[home] = s.bt; s_mb_u[home] = s.bu; s_mb_v[home] = s.bv; s_mb_prim[home] = s.bprim; s_mb_inst[home] = s.binst;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    // ... and take over what another piece has found closer; an any-hit ray is done once any piece has hit
                    if (alive && shared) {
                        if (any) hit_any = hit_any || s_mb_prim[home] != kMissPrim;
                        else if (s_mb_t[home] < L.s.bt) {
                            L.s.bt = s_mb_t[home]; L.s.bu = s_mb_u[home]; L.s.bv = s_mb_v[home]; L.s.bprim = s_mb_prim[home]; L.s.binst = s_mb_inst[home];
                        }
                    }
                }
                // ---- A. node step ----
                uint2 child = make_uint2(0u, 0u), tri = make_uint2(0u, 0u);
                __builtin_amdgcn_s_setprio(HRT_PRIO_NODE);
                wait_node_loads(rn0, rn1, rn2, rn3, rn4);
                if (L.nidx != kNoWork && !hit_any) lean_node(L.s, tmin, rn0, rn1, rn2, rn3, rn4, child, tri);
                // ---- B. bookkeeping (trav_lean.h: one hand-written sequence): file the new groups; the leaf pass (ONE per iteration, one
                //      primitive per lane, skipped while few lanes have leaf work and none depends on it); the primitive and the node of
                //      the next iteration; finished? ----
                __builtin_amdgcn_s_setprio(HRT_PRIO_BOOK);
                bool done = false;
                // an any-hit ray is done with its first accepted intersection: nothing more to fetch (what is left on its stacks is
                // dropped when the lane's next ray starts, lean_start)
                if (hit_any) { L.nidx = kNoWork; L.pidx = kNoWork; done = true; }
                if (alive && !done) done = lean_bookkeeping_asm(L, child, tri, ldsn, ldsl, (uint32_t)a.postpone_pct, (uint32_t)a.leaf_quorum) != 0u;
                if (done && (!kTail || !shared)) { alive = false; waiting = true; }
                if (kTail && a.tail_split) {
                    // a piece that has finished has nothing left to merge: the mailbox holds the ray's best hit
                    if (alive && done && shared) {
                        atomicSub(&s_mb_pending[home], 1u);
                        alive = false;
                        if (home != tx) { shared = false; home = tx; }        // a helper is free again; the owner waits for the last piece
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // the owner of a split ray picks the merged hit up once its last piece has finished
                    if (shared && !alive && home == tx && s_mb_pending[tx] == 0u) {
                        L.s.bt = s_mb_t[tx]; L.s.bu = s_mb_u[tx]; L.s.bv = s_mb_v[tx]; L.s.bprim = s_mb_prim[tx]; L.s.binst = s_mb_inst[tx];
                        shared = false; waiting = true;
                    }
                }
                const uint64_t act = __ballot(alive);
                if (act == 0ull) break;
                if (kTail ? (uint32_t)__popcll(__ballot(waiting)) >= (uint32_t)a.tail_regen : (64u - (uint32_t)__popcll(act)) >= (uint32_t)a.refill_threshold) break;
            }
        };
        if (exhausted) traverse(std::true_type{}); else traverse(std::false_type{});
    }
}

void launch_trace_queue(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    if (has_spheres) hipLaunchKernelGGL((k_trace_queue<true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace_queue<false>), g, b, 0, s, a);
}

}  // namespace hrt
