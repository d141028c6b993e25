// fused.hip -- k_fused, the production kernel of hrt_render_launch: the whole render in ONE persistent launch.
//
// Replaces the OptiX launch of the reference (raygen -> optixTrace -> closest-hit / miss, recursively, shader/Shader.cu:46-287;
// launched at src/Global/RendererMesh.cu:416-419).  Same execution model as round 1's fused mode of k_traverse (kernels.hip,
// still there as HRT_FUSED=2): every lane OWNS a pixel and carries its path state in registers -- RNG state, sample and depth
// counters, the albedo chain, the running sum; waves take 16-pixel slices of the tile from sharded counters; a lane whose ray
// has finished waits, and once enough lanes of the wave wait a regeneration phase (wave-uniform branch) shades them in place
// with the shared device functions of trav_common.h and starts the next ray in the same lane: the bounce, the next sample's
// primary ray, or the next pixel.  No ray queues, no hit records, no stage barriers.
// What is new: the traversal step between two regenerations -- trav_lean.h, written for instruction
// count: the SIMD issues one instruction of any kind per ~2.4 cycles and this kernel is bound by exactly that --, one scatter body
// for all programs, an issue priority per loop phase, and tail splitting once the tile is used up (lanes without a pixel take
// subtrees off the busy lanes' stacks; the pieces of a ray share one best hit in LDS).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "device_types.h"
#include "trav_common.h"
#include "trav_lean.h"

#pragma clang fp contract(off)

namespace hrt {

// Issue priority of a wave per phase of its loop (s_setprio: among the waves of a SIMD that are ready, the highest goes first).
// A wave that is about to fetch -- the bookkeeping that chooses its next node and primitive, the loads themselves -- and a wave
// in a regeneration go before a wave in its node step, and that before a wave in its primitive test: the fetches go out as
// early as possible and the waves drift apart instead of queueing for memory together.  C4: 3180 -> 3390 Mrays/s
// (profiles/r02_sweep_wave_priority.txt; every other assignment of the levels tried is within 3 % of this one, no priorities at
// all 6 % below).
#ifndef HRT_PRIO_BOOK
#define HRT_PRIO_BOOK 2      // bookkeeping + address arithmetic + load issue
#define HRT_PRIO_PRIM 0      // primitive wait + test
#define HRT_PRIO_NODE 1      // node wait + slab tests
#define HRT_PRIO_REGEN 2     // shading, next pixel, new ray
#endif

#ifndef HRT_SPHERE_CULL
#define HRT_SPHERE_CULL 1     // INSTANCED: a ray that misses an instance's bounding sphere does not enter it (0: enters every instance whose box it crosses)
#endif
struct InstLane { uint32_t inst_cur = kNoWork, frame = 0u; };
struct NoInstLane { static constexpr uint32_t inst_cur = 0u; };

// INSTANCED: the tree has two levels (bvh8.h: transform nodes; the reference's IAS over shared GASes, RendererImpl.cu:174-206).  A lane
// whose next node turns out to be a transform node leaves its world ray in LDS, goes on with the ray in the instance's object space
// (row-major 3x4 inverse from the node; identity: copied) and the BLAS's root as the only child; its node stack continues ABOVE what it
// held (`base` = the frame's bottom), so the bookkeeping sequence is the one-level kernel's, unchanged.  When it reports the frame
// empty the lane takes its world ray back and pops what it had left in world space.  A separate instantiation: the one-level kernels
// and their register budget are untouched.
// REUSE (HRT_CTX_REUSE_PRIMARY): the reference's raygen has no pixel jitter (Shader.cu:249-261), so the primary ray of a pixel -- and its
// hit -- is the same for every sample.  The first sample a launch takes of a pixel traces it and leaves the hit record in the lane's slot of
// `primary_cache`; the later ones are shaded from there (same record, same shading, same random numbers: the same bits) and only their
// bounces are traversed.  Rays that are not traversed are not counted.  A separate instantiation, like INSTANCED.
template <bool HAS_SPHERES, bool INSTANCED, bool REUSE>
#ifndef HRT_FUSED_WAVES_PER_SIMD
#define HRT_FUSED_WAVES_PER_SIMD 4      // 128 VGPRs, a dozen kernel constants spilled; 5 waves (96 VGPRs, 95 spilled around the shading): 2560 against 3122 Mrays/s
#endif
#ifndef HRT_INST_WAVES_PER_SIMD
#define HRT_INST_WAVES_PER_SIMD 3       // the INSTANCED instantiation: 160 VGPRs, nothing spilled, 12 waves per CU (at 4 waves per SIMD it spills 47 registers around
                                        // the shading: 2 % slower on both particle clouds, profiles/r04_two_level_sweep.txt)
#endif
__global__ __launch_bounds__(kTraverseBlock, INSTANCED ? HRT_INST_WAVES_PER_SIMD : HRT_FUSED_WAVES_PER_SIMD) void k_fused(TraverseArgs a) {
    static_assert(kTraverseBlock == 64, "one wave per workgroup: the stacks are per wave");
    __shared__ uint2 s_nodes[kNodeStackLds][kTraverseBlock];     // sibling groups: one per tree level (hrt_api.cpp sends deeper trees to k_traverse)
    __shared__ uint2 s_leaves[kLeafStackLds][kTraverseBlock];    // leaf groups
    // tail splitting: one mailbox per lane that owns a split ray (indexed by its home lane) collects the pieces' hits
    __shared__ float s_mb_t[kTraverseBlock], s_mb_u[kTraverseBlock], s_mb_v[kTraverseBlock];
    __shared__ uint32_t s_mb_prim[kTraverseBlock], s_mb_inst[kTraverseBlock], s_mb_pending[kTraverseBlock];
    __shared__ uint32_t s_pair[kTraverseBlock];
    // INSTANCED: the rest of a world ray that waits while its lane is inside an instance (reciprocals, octant)
    __shared__ float s_park_idx[kTraverseBlock], s_park_idy[kTraverseBlock], s_park_idz[kTraverseBlock];
    __shared__ uint32_t s_park_oct[kTraverseBlock];

    const uint32_t n_pixels = a.path.n_tile_pixels;
    const char *__restrict__ node_bytes = reinterpret_cast<const char *>(a.nodes);
    const char *__restrict__ prim_bytes = reinterpret_cast<const char *>(a.prims);
    const float tmin = a.tmin, tmax_ray = a.tmax;
    const uint32_t leaf_hold = a.leaf_hold >= 1 && a.leaf_hold <= 4 ? (uint32_t)a.leaf_hold : 4u;      // (a lane that could never take a node would never finish)
    const uint32_t tx = threadIdx.x;
    const uint32_t ldsn = (uint32_t)reinterpret_cast<uintptr_t>(&s_nodes[0][tx]), ldsl = (uint32_t)reinterpret_cast<uintptr_t>(&s_leaves[0][tx]);

    // INSTANCED: the instance whose BLAS this lane is in (kNoWork: none), and what its node stack looked like when it went in
    // (base | entries << 8); the lane's world ray waits in LDS meanwhile (the mailboxes' memory: tail splitting is off for two-level trees)
    std::conditional_t<INSTANCED, InstLane, NoInstLane> I;

    LeanLane L;
    lean_reset(L);
    L.s.bt = tmax_ray; L.s.bu = 0.0f; L.s.bv = 0.0f; L.s.bprim = kMissPrim; L.s.binst = kMissPrim;
    L.s.ox = L.s.oy = L.s.oz = 0.0f; L.s.dx = L.s.dy = 0.0f; L.s.dz = 1.0f; L.s.idx = L.s.idy = L.s.idz = 1.0f; L.s.oct_inv4 = 0u;
    bool alive = false;                     // a ray is being traversed in this lane
    bool waiting = false;                   // ... has finished and waits for the next regeneration
    bool any = false;                       // this lane's ray only needs to know whether anything is hit
    bool exhausted = false;                 // wave-uniform: no pixels left to start
    bool shared = false;                    // this lane works on a piece of a ray that has been split across lanes (tail splitting)
    uint32_t home = tx;                     // ... whose owner is this lane

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
    unsigned long long ls_iter = 0, ls_alive = 0, ls_node = 0, ls_prim = 0, ls_ppass = 0, ls_regen = 0, ls_enter = 0;
#endif

    // the wave's slice of the tile: [wbeg, wend); slices of fetch_chunk pixels are handed out by kFetchShards counters
    uint32_t wbeg = 0, wend = 0, kstart = 0;
    const uint32_t home_shard = blockIdx.x & (kFetchShards - 1);

    [[maybe_unused]] bool force_regen = false;      // (REUSE, wave-uniform)
    [[maybe_unused]] bool cached = false;           // (REUSE) this lane waits to be shaded with its pixel's cached primary hit
    for (;;) {
        const uint64_t idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        // ---- regenerate: shade finished rays in place, start the next sample / pixel ----
        // (once the tile is used up the lanes without a pixel stay idle and the render ends with the slowest pixels' sample chains:
        // what counts then is how soon a finished ray's successor starts, against what a regeneration costs the rays still under
        // way -- a dozen waiting rays, or nothing else left to do: 1/8 of the C4 frame 142 ms with 1, 129 ms with 8 to 16)
        if (force_regen || idle == ~0ull || (exhausted ? (uint32_t)__popcll(__ballot(waiting)) >= (uint32_t)a.tail_regen : n_idle >= (uint32_t)a.refill_threshold)) {
#ifdef HRT_LANE_STATS
            ++ls_regen;
#endif
            __builtin_amdgcn_s_setprio(HRT_PRIO_REGEN);
            bool launch = false, want_primary = false;      // launch: this lane starts the ray (ro, rd) below
            [[maybe_unused]] bool reshade = false;
            [[maybe_unused]] const bool was_forced = force_regen;
            force_regen = false;
            V3 ro = mk3(0.0f, 0.0f, 0.0f), rd = mk3(0.0f, 0.0f, 1.0f);
            if (!alive && waiting) {
                waiting = false;
                const TravState &s = L.s;
                // the finished ray and what it hit
                V3 o = mk3(s.ox, s.oy, s.oz), d = mk3(s.dx, s.dy, s.dz);
                float bt = s.bt, bu = s.bu, bv = s.bv; uint32_t bprim = s.bprim, binst = s.binst;
                if constexpr (REUSE) {
                    float4 *slot = a.path.primary_cache + 2u * (blockIdx.x * kTraverseBlock + tx);
                    if (cached) {                   // ... or, for a primary ray that was not traversed again, what the pixel's first sample found
                        const float4 c0 = slot[0]; const float4 c1 = slot[1];
                        o = mk3(a.path.center[0], a.path.center[1], a.path.center[2]); d = mk3(px_pdx, px_pdy, px_pdz);
                        bt = c0.x; bu = c0.y; bv = c0.z; bprim = __float_as_uint(c0.w); binst = __float_as_uint(c1.x);
                        cached = false;
                    } else if (!a.path.trace_rays && px_depth == 1u && px_sample == 0u) {      // the primary hit of the pixel's first sample in this launch
                        slot[0] = make_float4(bt, bu, bv, __uint_as_float(bprim)); slot[1] = make_float4(__uint_as_float(binst), 0.0f, 0.0f, 0.0f);
                    }
                }
                const bool miss = bprim == kMissPrim;
                if (a.path.trace_rays) {           // hrt_trace_rays on this kernel: the "pixel" is a caller's ray, its hit record the result
                    a.path.trace_tuvp[px_local] = make_float4(bt, bu, bv, __uint_as_float(bprim));
                    a.path.trace_inst[px_local] = binst;
                    have_pixel = false;
                } else if (miss || px_depth >= kRayTraceDepth) {
                    // the path ends: miss colour or black at the depth limit, folded through the albedo chain (Shader.cu:102-107, :236-238, :276-287)
                    const V3 r = fold_chain(miss, a.path.bg, px_chain, px_depth, a.path.hitgroups);
                    if (px_first) { px_ax = r.x; px_ay = r.y; px_az = r.z; px_first = false; }
                    else { px_ax += r.x; px_ay += r.y; px_az += r.z; }
                    ++px_sample;
                    if constexpr (REUSE) {
                        // a primary ray that leaves the scene: every sample of the pixel is the background colour, added one by one
                        if (miss && px_depth == 1u && !a.path.slice_cost)
                            for (; px_sample < a.path.spp; ++px_sample) { px_ax += r.x; px_ay += r.y; px_az += r.z; }
                    }
                    if (a.path.slice_cost)     // probe launch: how long this pixel's sample took, start of its primary ray to here
                        atomicAdd(a.path.slice_cost + px_local / a.fetch_chunk, ((uint32_t)__builtin_amdgcn_s_memtime() - px_t0) >> 4);
                    if (px_sample >= a.path.spp) {
                        a.path.accum[px_local] = make_float4(px_ax, px_ay, px_az, 0.0f);
                        rng_store(a.path.states + px_tid, px_rng);
                        have_pixel = false;
                    } else want_primary = true;
                } else {
                    const uint32_t inst = binst;
                    const HitGroup hg = a.path.hitgroups[inst];
                    const uint32_t program = a.path.inst_program[inst];
                    V3 hp, nd;
                    scatter_programs<HAS_SPHERES>(program, hg, o, d, bt, bu, bv, bprim, px_rng, hp, nd);
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
                    launch = true;
                } else {
                    px_depth = 1u;
                    ro = mk3(a.path.center[0], a.path.center[1], a.path.center[2]); rd = mk3(px_pdx, px_pdy, px_pdz);
                    launch = true;
                    if constexpr (REUSE) {
                        // not traversed again: the lane waits for its shading as if the ray had just finished (the hit is in the cache)
                        if (px_sample > 0u) { launch = false; waiting = true; cached = true; reshade = true; }
                    }
                }
            }
            if (launch) {
                any = px_depth >= kRayTraceDepth;      // a hit at the depth limit is black whatever it is (Shader.cu:102-107)
                if (any) ++px_rays_any; else ++px_rays_closest;
                lean_start(L, ro, rd, tmax_ray);
                if constexpr (INSTANCED) I.inst_cur = kNoWork;      // (an any-hit ray may have ended inside an instance)
                alive = true;
            }
            // REUSE: lanes that have just taken their primary hit from the cache are shaded in a second regeneration, at once, so that
            // their bounces start together with the other lanes' rays (one extra round, not more: the others are waiting)
            if constexpr (REUSE) { if (!was_forced && __ballot(reshade) != 0ull) { force_regen = true; continue; } }
        }
        // the tile is used up and every lane has finished (nothing waits after a full regeneration -- but, REUSE, a lane with a cached hit to shade:
        // an empty pass through the loop below brings it back here)
        if (__ballot(REUSE ? alive || waiting : alive) == 0ull) break;

        // ---- traverse until enough lanes have finished to make a regeneration worthwhile ----
        // Two copies of the loop: the second one, with tail splitting and the drained phase's exit rule, runs once the tile is used
        // up -- the first pays nothing for either.
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
                if (kTail && !INSTANCED && a.tail_split) {          // (one donation per busy lane and iteration: more rounds of this change nothing, r02_sweep_tile_tail.txt)
                    const bool is_free = !alive && !waiting && !have_pixel && !shared;
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
    #ifdef HRT_LANE_STATS
                { ++ls_iter; ls_alive += __popcll(__ballot(alive)); ls_node += __popcll(mask_n0); ls_prim += __popcll(mask_p); ls_ppass += mask_p != 0ull; }
    #endif
                // ---- C. leaf test: waits for the primitive pieces only (the node loads issued behind them stay in flight) ----
                bool hit_any = false, improved = false;
                if (mask_p != 0ull) {
                    wait_prim_loads(rpa, rpb, rpc);
                    if (L.pidx != kNoWork) {
                        const float4 pa = make_float4(rpa.x, rpa.y, rpa.z, rpa.w), pb = make_float4(rpb.x, rpb.y, rpb.z, rpb.w),
                                     pc = make_float4(rpc.x, rpc.y, rpc.z, rpc.w);
                        improved = test_prim<HAS_SPHERES, INSTANCED>(pa, pb, pc, L.s, tmin, tmax_ray, a.inst_inv, a.inst_identity, I.inst_cur);
                        hit_any = any && improved;
                    }
                }
                if (kTail && !INSTANCED && a.tail_split) {
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
                [[maybe_unused]] bool enter = false;
                if constexpr (!INSTANCED) {
                    if (L.nidx != kNoWork && !hit_any) lean_node(L.s, tmin, rn0, rn1, rn2, rn3, rn4, child, tri);
                } else {
                    enter = L.nidx != kNoWork && !hit_any && rn0.w == 0u;          // a transform node: word 3 == 0
                    if (L.nidx != kNoWork && !hit_any && !enter) lean_node(L.s, tmin, rn0, rn1, rn2, rn3, rn4, child, tri);
                    if (__ballot(enter) != 0ull) {
                        float ox = L.s.ox, oy = L.s.oy, oz = L.s.oz, dx = L.s.dx, dy = L.s.dy, dz = L.s.dz;      // the ray in the instance's object space
                        if (enter) {
                            const TravState &s = L.s;
                            if (rn1.z == 0u) {          // not the identity: xf_point / xf_vector of the oracle, operation for operation
                                const float m0 = __uint_as_float(rn2.x), m1 = __uint_as_float(rn2.y), m2 = __uint_as_float(rn2.z), m3 = __uint_as_float(rn2.w);
                                const float m4 = __uint_as_float(rn3.x), m5 = __uint_as_float(rn3.y), m6 = __uint_as_float(rn3.z), m7 = __uint_as_float(rn3.w);
                                const float m8 = __uint_as_float(rn4.x), m9 = __uint_as_float(rn4.y), m10 = __uint_as_float(rn4.z), m11 = __uint_as_float(rn4.w);
                                ox = ((m0 * s.ox + m1 * s.oy) + m2 * s.oz) + m3; oy = ((m4 * s.ox + m5 * s.oy) + m6 * s.oz) + m7; oz = ((m8 * s.ox + m9 * s.oy) + m10 * s.oz) + m11;
                                dx = (m0 * s.dx + m1 * s.dy) + m2 * s.dz; dy = (m4 * s.dx + m5 * s.dy) + m6 * s.dz; dz = (m8 * s.dx + m9 * s.dy) + m10 * s.dz;
                            }
                            // The object-space ray against the BLAS's bounding sphere (words 0-2: centre, word 7: radius, negative: none) -- the
                            // instance's box in the top level is the box of that sphere under a rotation nobody knows in advance, and half the rays
                            // that cross such a box miss the sphere.  Culling only, with slack for the rounding of every term: misses the line of
                            // the ray by more than the radius, or starts outside and points away.
                            const float R = __uint_as_float(rn1.w);
                            const float cx = ox - __uint_as_float(rn0.x), cy = oy - __uint_as_float(rn0.y), cz = oz - __uint_as_float(rn0.z);
                            const float cc = fmaf(cx, cx, fmaf(cy, cy, cz * cz)), aa = fmaf(dx, dx, fmaf(dy, dy, dz * dz)), b = fmaf(cx, dx, fmaf(cy, dy, cz * dz));
                            const float R2 = R * R * 1.0001f, ca = cc * aa;
                            // (branch-free on purpose, & and | instead of && and ||: with a branch on R >= 0 inside this block hipcc 7.2 carries child.x and
                            // I.inst_cur of the block below through the registers it also uses for cc and b here, and the lanes that took the branch
                            // entered node 0 instead of their BLAS, for ever -- found in the ISA, tools/debug_two_level.py)
                            const bool beside = fmaf(-b, b, ca) > fmaf(R2, aa, 4e-6f * ca), behind = (b > 0.0f) & (cc > fmaf(4e-6f, cc, R2));
                            if (HRT_SPHERE_CULL && ((R >= 0.0f) & (beside | behind))) enter = false;
                        }
                        if (enter) {
                            TravState &s = L.s;
                            s_mb_t[tx] = s.ox; s_mb_u[tx] = s.oy; s_mb_v[tx] = s.oz;
                            s_mb_prim[tx] = __float_as_uint(s.dx); s_mb_inst[tx] = __float_as_uint(s.dy); s_mb_pending[tx] = __float_as_uint(s.dz);
                            s_park_idx[tx] = s.idx; s_park_idy[tx] = s.idy; s_park_idz[tx] = s.idz; s_park_oct[tx] = s.oct_inv4;
                            I.inst_cur = rn1.y;
                            s.ox = ox; s.oy = oy; s.oz = oz; s.dx = dx; s.dy = dy; s.dz = dz;
                            s.idx = safe_rcp_dir<false>(dx); s.idy = safe_rcp_dir<false>(dy); s.idz = safe_rcp_dir<false>(dz);
                            const uint32_t oct = (dx < 0.0f ? 4u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 1u : 0u);
                            s.oct_inv4 = (7u - oct) * 0x01010101u;
                            child = make_uint2(rn1.x, 0x01000000u);      // one child, no inner-mask bits: the pick below is child base + 0 = the BLAS's root
                        }
    #ifdef HRT_LANE_STATS
                        ls_enter += __popcll(__ballot(enter));
    #endif
                    }
                }
                // ---- B. bookkeeping (trav_lean.h: one hand-written sequence): file the new groups; the leaf pass (ONE per iteration, one
                //      primitive per lane, skipped while few lanes have leaf work and none depends on it); the primitive and the node of
                //      the next iteration; finished? ----
                __builtin_amdgcn_s_setprio(HRT_PRIO_BOOK);
                bool done = false;
                // an any-hit ray is done with its first accepted intersection: nothing more to fetch (what is left on its stacks is
                // dropped when the lane's next ray starts, lean_start)
                if (hit_any) { L.nidx = kNoWork; L.pidx = kNoWork; done = true; }
                [[maybe_unused]] uint32_t lane = 0u;
                if constexpr (!INSTANCED) {
                    if (alive && !done) done = lean_bookkeeping_asm(L, child, tri, ldsn, ldsl, (uint32_t)a.postpone_pct, (uint32_t)a.leaf_quorum, leaf_hold) != 0u;
                } else {
                    // (this instantiation is two registers over its budget and the compiler's choice of what to keep in scratch is the two
                    // stack addresses, reloaded here in every iteration: they are a constant plus eight times the lane number -- two
                    // instructions to make again)
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
                    const uint32_t ldsn_i = (uint32_t)reinterpret_cast<uintptr_t>(&s_nodes[0][0]) + 8u * lane, ldsl_i = (uint32_t)reinterpret_cast<uintptr_t>(&s_leaves[0][0]) + 8u * lane;
                    if (alive && !done) done = lean_bookkeeping_asm(L, child, tri, ldsn_i, ldsl_i, (uint32_t)a.postpone_pct, (uint32_t)a.leaf_quorum, leaf_hold) != 0u;
                }
                if constexpr (INSTANCED) {
                    // in: the frame starts above what the lane holds (siblings still in hand have just been pushed, step 2 of the sequence)
                    if (enter) { I.frame = (uint32_t)L.base | ((uint32_t)L.nsp << 8); L.base = L.nsp; }
                    // out: the frame is empty -- nothing in hand, on the node stack above `base`, or in the leaf stack -- but the ray is not done
                    const bool leave = done && !hit_any && I.inst_cur != kNoWork;
                    if (__ballot(leave) != 0ull) {
                        if (leave) {
                            TravState &s = L.s;
                            s.ox = s_mb_t[tx]; s.oy = s_mb_u[tx]; s.oz = s_mb_v[tx];
                            s.dx = __uint_as_float(s_mb_prim[tx]); s.dy = __uint_as_float(s_mb_inst[tx]); s.dz = __uint_as_float(s_mb_pending[tx]);
                            s.idx = s_park_idx[tx]; s.idy = s_park_idy[tx]; s.idz = s_park_idz[tx]; s.oct_inv4 = s_park_oct[tx];
                            I.inst_cur = kNoWork;
                            L.base = (int)(I.frame & 0xffu); L.nsp = (int)(I.frame >> 8);
                            if (L.nsp != L.base) { --L.nsp; s.cur = s_nodes[L.nsp][lane]; }      // (only groups with hits are ever pushed)
                            if (s.cur.y > 0x00ffffffu) { lean_pick_node(L); done = false; }
                        }
                    }
                }
                if (done && (!kTail || !shared)) { alive = false; waiting = true; }
                if (kTail && !INSTANCED && a.tail_split) {
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
#ifdef HRT_LANE_STATS
    if (tx == 0u) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.path.rays_closest);
        atomicAdd(d + 6, ls_iter); atomicAdd(d + 7, ls_alive); atomicAdd(d + 8, ls_node); atomicAdd(d + 9, ls_prim); atomicAdd(d + 2, ls_ppass); atomicAdd(d + 3, ls_regen); atomicAdd(d + 4, ls_enter);
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
    const bool reuse = a.path.primary_cache != nullptr;
    if (has_spheres && reuse) hipLaunchKernelGGL((k_fused<true, false, true>), g, b, 0, s, a);
    else if (has_spheres) hipLaunchKernelGGL((k_fused<true, false, false>), g, b, 0, s, a);
    else if (reuse) hipLaunchKernelGGL((k_fused<false, false, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_fused<false, false, false>), g, b, 0, s, a);
}
// ... through a two-level tree (transform nodes over shared BLASes)
void launch_fused_instanced(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    const bool reuse = a.path.primary_cache != nullptr;
    if (has_spheres && reuse) hipLaunchKernelGGL((k_fused<true, true, true>), g, b, 0, s, a);
    else if (has_spheres) hipLaunchKernelGGL((k_fused<true, true, false>), g, b, 0, s, a);
    else if (reuse) hipLaunchKernelGGL((k_fused<false, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_fused<false, true, false>), g, b, 0, s, a);
}

}  // namespace hrt
