// paths.hip -- k_paths, the production kernel of hrt_render_launch: a whole render in ONE persistent launch, organised as
// a wavefront pipeline that lives inside each wave ("slot pipeline").
//
// Replaces the OptiX launch of the reference (raygen -> optixTrace -> closest-hit / miss, recursively, shader/Shader.cu:46-287;
// launched at src/Global/RendererMesh.cu:416-419) like the round-1 fused kernel (k_traverse<.., FUSED> in kernels.hip) did, but
// with the two things that kernel measured as its limits taken apart:
//
//   * there, a lane OWNED a pixel: when its ray finished the lane sat idle until enough lanes waited to make shading them
//     worthwhile (47 of 64 lanes alive on average), the shading ran for 17-24 lanes of 64, and the pixel's state (RNG, albedo
//     chain, running sum ...) occupied 25 registers of every lane all the time;
//   * here, a wave keeps kPipeSlots = 128 pixels in flight.  A pixel is a SLOT: its state lives in global memory (PathSlot,
//     112 B, L2 / Infinity-Cache resident) and its current ray is in one of three places -- a lane (being traversed), the
//     wave's RAY QUEUE (waiting for a lane) or its HIT QUEUE (traversal finished, waiting to be shaded); both queues and
//     the 32-byte ray / hit record of every slot are in LDS, private to the wave, so nothing is shared, locked or fenced.
//     A lane whose ray ends drops the hit record into the hit queue and takes the next ray from the ray queue at once
//     (a few LDS operations); when the ray queue has run dry the wave shades the whole hit queue in one full-width batch:
//     every lane takes one hit, loads the slot, runs the closest-hit / miss program (the same device functions as the
//     wavefront kernels: one rounding behaviour), writes the slot back and puts the new ray -- the bounce, the pixel's
//     next sample, or the next pixel of the wave's slice -- into the ray queue.
//
// Every pixel still has exactly one ray in flight and draws from its own XORWOW stream in path order, so the image, the final
// RNG states and the ray counts are those of the oracle, bit for bit, whatever the scheduling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "trav_common.h"

#pragma clang fp contract(off)

namespace hrt {

// ring-buffer position (head, k < SLOTS)
template <int SLOTS>
__device__ __forceinline__ uint32_t ring_pos(uint32_t head, uint32_t k) { const uint32_t i = head + k; return i >= (uint32_t)SLOTS ? i - (uint32_t)SLOTS : i; }

constexpr uint32_t kPackedFirst = 1u << 20;
__device__ __forceinline__ uint32_t packed_sample(uint32_t p) { return p & 0xffffu; }
__device__ __forceinline__ uint32_t packed_depth(uint32_t p) { return (p >> 16) & 0xfu; }

template <bool HAS_SPHERES, int kPipeSlots>
__global__ __launch_bounds__(kTraverseBlock, 5) void k_paths(TraverseArgs a) {
    static_assert(kTraverseBlock == 64 && kPipeSlots <= kMaxPipeSlots && kPipeSlots > 64, "one wave per workgroup; queue entries are bytes");
    auto ring_at = [](uint32_t head, uint32_t k) { return ring_pos<kPipeSlots>(head, k); };
    __shared__ uint2 s_stack[kLdsStack][kTraverseBlock];
    // per slot, 48 B: the ray waiting for a lane {o.xyz, any}{d.xyz, -}{-}  or its hit record {hit point.xyz, prim}{d.xyz, inst}{u, v, -, -}
    __shared__ uint4 s_rec[kPipeSlots][3];
    __shared__ uint8_t s_rayq[kPipeSlots], s_hitq[kPipeSlots];     // ring buffers of slot numbers
    uint2 spill[kSpillStack];

    const uint32_t tx = threadIdx.x;
    const uint32_t n_pixels = a.path.n_tile_pixels;
    const char *__restrict__ node_bytes = reinterpret_cast<const char *>(a.nodes);
    const char *__restrict__ prim_bytes = reinterpret_cast<const char *>(a.prims);
    const float tmin = a.tmin, tmax_ray = a.tmax;
    uint4 *__restrict__ gslots = reinterpret_cast<uint4 *>(a.path.slots + (size_t)blockIdx.x * kMaxPipeSlots);     // 4 pieces per slot

    // ---- lane state: traversal only ----
    TravState s;
    bool alive = false;            // a ray is being traversed in this lane
    bool finished = false;         // ... has ended; its hit record goes to the hit queue at the next exchange
    bool any = false;
    uint32_t slot = 0u;
    bool has_node = false, has_prim = false;
    uint32_t nidx = 0u, pidx = 0u;

    // ---- wave state (uniform) ----
    uint32_t rq_head = 0u, rq_count = 0u, hq_head = 0u, hq_count = 0u;
    uint32_t n_alloc = 0u;                       // slots handed out so far (a slot carries pixel after pixel until the tile is used up)
    uint32_t wbeg = 0u, wend = 0u, kstart = 0u;  // the wave's current slice of the tile
    bool exhausted = false;                      // no pixels left to start
    const uint32_t home_shard = blockIdx.x & (kFetchShards - 1);
    uint32_t cnt_closest = 0u, cnt_any = 0u;     // rays started, by kind (per lane; reduced at the end)
#ifdef HRT_LANE_STATS
    unsigned long long ls_iter = 0, ls_alive = 0, ls_node = 0, ls_prim = 0, ls_ppass = 0, ls_batches = 0, ls_batch_lanes = 0;
#endif

    auto advance_select = [&]() {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (s.cur.y <= 0x00ffffffu && s.sp > 0) {
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

    for (;;) {
        // =====================================================================================================
        // exchange: finished lanes -> hit queue; [shading batch]; idle lanes <- ray queue
        // =====================================================================================================
        {
            const uint64_t fin = __ballot(finished);
            if (fin != 0ull) {
                if (finished) {
                    const V3 hp = hit_point(mk3(s.ox, s.oy, s.oz), mk3(s.dx, s.dy, s.dz), s.bt);      // Shader.cu:111-114
                    s_rec[slot][0] = make_uint4(__float_as_uint(hp.x), __float_as_uint(hp.y), __float_as_uint(hp.z), s.bprim);
                    s_rec[slot][1] = make_uint4(__float_as_uint(s.dx), __float_as_uint(s.dy), __float_as_uint(s.dz), s.binst);
                    s_rec[slot][2] = make_uint4(__float_as_uint(s.bu), __float_as_uint(s.bv), 0u, 0u);
                    s_hitq[ring_at(hq_head, hq_count + lane_prefix(fin))] = (uint8_t)slot;
                    finished = false;
                }
                hq_count += (uint32_t)__popcll(fin);
            }
        }
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(!alive));
        // A shading batch is due BEFORE the ray queue runs dry -- it is down to what the idle lanes are about to take plus a
        // few (low_water) -- provided the hit queue holds a worthwhile number of hits (min_batch); or when the hit queue fills a
        // wave; or, whatever the counts, when lanes starve and nothing else can feed them (start and end of the render).  The
        // lanes of the batch without a hit open new slots while slots and pixels last.
        const bool can_start = !exhausted && n_alloc < (uint32_t)kPipeSlots;
        const bool starving = rq_count < n_idle && (n_idle >= (uint32_t)a.path.shade_threshold || rq_count + (64u - n_idle) == 0u || exhausted);
        const bool low = rq_count <= n_idle + (uint32_t)a.path.low_water && hq_count >= (uint32_t)a.path.min_batch;
        if (hq_count >= 64u || ((low || starving) && (hq_count > 0u || can_start))) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_hits = hq_count < 64u ? hq_count : 64u;
            const bool has_hit = tx < n_hits;
            uint32_t bslot = 0u;                     // the slot this lane works on in the batch
            if (has_hit) bslot = s_hitq[ring_at(hq_head, tx)];
            hq_head = ring_at(hq_head, n_hits); hq_count -= n_hits;
#ifdef HRT_LANE_STATS
            ++ls_batches; ls_batch_lanes += n_hits;
#endif
            uint32_t rng_w[6] = {0u, 0u, 0u, 0u, 0u, 0u}, chain[4] = {0u, 0u, 0u, 0u};
            uint32_t packed = 0u, px_local = 0u, px_tid = 0u, t0 = 0u;
            V3 ro = mk3(0.0f, 0.0f, 0.0f), rd = mk3(0.0f, 0.0f, 1.0f);      // the ray this lane emits
            bool emit = false;                       // this lane puts a ray into the ray queue
            bool need_pixel = false;                 // ... needs a (new) pixel first
            bool next_sample = false;                // ... starts the next sample of its pixel
            bool have_slot = has_hit;
            bool store_chain = false;
            if (has_hit) {
                const uint4 g0 = gslots[4 * bslot + 0], g1 = gslots[4 * bslot + 1], g2 = gslots[4 * bslot + 2];
                const uint4 h0 = s_rec[bslot][0], h1 = s_rec[bslot][1], h2 = s_rec[bslot][2];
                rng_w[0] = g0.x; rng_w[1] = g0.y; rng_w[2] = g0.z; rng_w[3] = g0.w; rng_w[4] = g1.x; rng_w[5] = g1.y;
                packed = g1.z; px_local = g1.w;
                chain[0] = g2.x; chain[1] = g2.y; chain[2] = g2.z; chain[3] = g2.w;
                const uint32_t hprim = h0.w, hinst = h1.w;
                const uint32_t depth = packed_depth(packed);
                const bool miss = hprim == kMissPrim;
                if (miss || depth >= kRayTraceDepth) {
                    // the path ends: miss colour or black at the depth limit, folded through the albedo chain (Shader.cu:102-107, :236-238, :276-287)
                    const V3 r = fold_chain(miss, a.path.bg, chain, depth, a.path.hitgroups);
                    const uint4 g3 = gslots[4 * bslot + 3];
                    px_tid = g3.x; t0 = g3.y;
                    float4 acc;
                    if (packed & kPackedFirst) { acc = make_float4(r.x, r.y, r.z, 0.0f); packed &= ~kPackedFirst; }
                    else { acc = a.path.accum[px_local]; acc.x += r.x; acc.y += r.y; acc.z += r.z; }
                    a.path.accum[px_local] = acc;
                    packed = (packed & ~0xfffffu) | (packed_sample(packed) + 1u);             // one more sample; no ray in flight
                    if (a.path.slice_cost)      // probe launch: how long this pixel's sample took, start of the pixel to here
                        atomicAdd(a.path.slice_cost + px_local / a.fetch_chunk, ((uint32_t)__builtin_amdgcn_s_memtime() - t0) >> 4);
                    if (packed_sample(packed) >= a.path.spp) {
                        uint2 *sp = reinterpret_cast<uint2 *>(a.path.states + px_tid);
                        sp[0] = make_uint2(rng_w[0], rng_w[1]); sp[1] = make_uint2(rng_w[2], rng_w[3]); sp[2] = make_uint2(rng_w[4], rng_w[5]);
                        need_pixel = true;
                    } else next_sample = true;
                } else {
                    const HitGroup hg = a.path.hitgroups[hinst];
                    const uint32_t program = a.path.inst_program[hinst];
                    Xorwow rng; rng.d = rng_w[0]; rng.v0 = rng_w[1]; rng.v1 = rng_w[2]; rng.v2 = rng_w[3]; rng.v3 = rng_w[4]; rng.v4 = rng_w[5];
                    const V3 hp = mk3(__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h0.z));
                    const V3 dir = mk3(__uint_as_float(h1.x), __uint_as_float(h1.y), __uint_as_float(h1.z));
                    const float hu = __uint_as_float(h2.x), hv = __uint_as_float(h2.y);
                    V3 nd;
                    if (program == (uint32_t)kProgramTriangleRough) scatter_at<false, true>(hg, hp, dir, hu, hv, hprim, rng, nd);
                    else if (program == (uint32_t)kProgramTriangleMetal) scatter_at<false, false>(hg, hp, dir, hu, hv, hprim, rng, nd);
                    else if (HAS_SPHERES && program == (uint32_t)kProgramSphereRough) scatter_at<true, true>(hg, hp, dir, hu, hv, hprim, rng, nd);
                    else if (HAS_SPHERES) scatter_at<true, false>(hg, hp, dir, hu, hv, hprim, rng, nd);
                    rng_w[0] = rng.d; rng_w[1] = rng.v0; rng_w[2] = rng.v1; rng_w[3] = rng.v2; rng_w[4] = rng.v3; rng_w[5] = rng.v4;
                    chain[depth - 1u] = hinst; store_chain = true;
                    packed += 1u << 16;                                                        // depth + 1
                    ro = hp; rd = nd;
                    emit = true;
                }
            }
            // lanes of the batch without a hit open new slots while there are slots and pixels left
            {
                const uint64_t spare = __ballot(!has_hit);
                const uint32_t room = (uint32_t)kPipeSlots - n_alloc;
                if (!exhausted && room > 0u && spare != 0ull) {
                    const uint32_t rank = lane_prefix(spare);
                    if (!has_hit && rank < room) { bslot = n_alloc + rank; need_pixel = true; have_slot = true; }
                    const uint32_t n_spare = (uint32_t)__popcll(spare);
                    n_alloc += n_spare < room ? n_spare : room;
                }
            }
            // pixels for the lanes that need one: the next ones of the wave's slice of the tile
            bool got_pixel = false;
            for (;;) {
                const uint64_t need = __ballot(need_pixel && !got_pixel);
                if (need == 0ull || exhausted) break;
                if (wbeg >= wend) {
                    for (uint32_t k = kstart; k < kFetchShards && wbeg >= wend; ++k) {
                        const uint32_t shard = (home_shard + k) & (kFetchShards - 1);
                        uint32_t c = 0;
                        if (tx == 0u) c = atomicAdd(a.fetch_counter + shard * kFetchShardStride, 1u);
                        c = (uint32_t)__shfl((int)c, 0);
                        const uint64_t q = (uint64_t)c * kFetchShards + shard;          // the q-th slice handed out ...
                        if (q * (uint64_t)a.fetch_chunk < (uint64_t)n_pixels) {
                            // ... is slice slice_order[q] of the tile: the expensive slices first (longest-processing-time-first)
                            const uint64_t beg = (a.path.slice_order ? (uint64_t)a.path.slice_order[q] : q) * (uint64_t)a.fetch_chunk;
                            wbeg = (uint32_t)beg;
                            wend = (uint32_t)(beg + a.fetch_chunk < (uint64_t)n_pixels ? beg + a.fetch_chunk : (uint64_t)n_pixels);
                        } else kstart = k + 1;
                    }
                    if (wbeg >= wend) { exhausted = true; break; }
                }
                const uint32_t n_need = (uint32_t)__popcll(need);
                const uint32_t take = n_need < wend - wbeg ? n_need : wend - wbeg;
                const uint32_t rank = lane_prefix(need);
                const uint32_t mine = wbeg + rank;
                wbeg += take;
                if (need_pixel && !got_pixel && rank < take) {
                    const uint32_t j = a.path.first_pixel + mine;
                    const uint32_t row = j / a.path.width;
                    const uint32_t ix = j - row * a.path.width;
                    const uint32_t iy = a.path.rows[row];
                    px_local = j; px_tid = iy * a.path.width + ix;
                    const uint2 *sp = reinterpret_cast<const uint2 *>(a.path.states + px_tid);
                    const uint2 r0 = sp[0], r1 = sp[1], r2 = sp[2];
                    rng_w[0] = r0.x; rng_w[1] = r0.y; rng_w[2] = r1.x; rng_w[3] = r1.y; rng_w[4] = r2.x; rng_w[5] = r2.y;
                    packed = a.path.continue_sum == 0u ? kPackedFirst : 0u;       // later launches of a long render continue the pixel's sum
                    t0 = a.path.slice_cost ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
                    gslots[4 * bslot + 3] = make_uint4(px_tid, t0, 0u, 0u);
                    got_pixel = true; next_sample = true;
                }
            }
            // (a lane that needed a pixel and got none lets its slot run out: the tile is used up)
            if (next_sample) {          // the primary ray of the pixel's next sample: the same for every sample (no jitter, Shader.cu:249-261)
                const uint32_t iy = px_tid / a.path.width, ix = px_tid - iy * a.path.width;
                rd = primary_direction(ix, iy, a.path.width, a.path.height, a.path.U, a.path.V, a.path.W);
                ro = mk3(a.path.center[0], a.path.center[1], a.path.center[2]);
                packed = (packed & ~0xf0000u) | (1u << 16);                     // depth 1
                emit = true;
            }
            // the new rays: slot state back to memory, ray record into LDS, slot number into the ray queue
            {
                const uint64_t em = __ballot(emit && have_slot);
                if (emit && have_slot) {
                    const bool ray_any = packed_depth(packed) >= kRayTraceDepth;   // a hit at the depth limit is black whatever it is (Shader.cu:102-107)
                    if (ray_any) ++cnt_any; else ++cnt_closest;
                    gslots[4 * bslot + 0] = make_uint4(rng_w[0], rng_w[1], rng_w[2], rng_w[3]);
                    gslots[4 * bslot + 1] = make_uint4(rng_w[4], rng_w[5], packed, px_local);
                    if (store_chain) gslots[4 * bslot + 2] = make_uint4(chain[0], chain[1], chain[2], chain[3]);
                    s_rec[bslot][0] = make_uint4(__float_as_uint(ro.x), __float_as_uint(ro.y), __float_as_uint(ro.z), ray_any ? 1u : 0u);
                    s_rec[bslot][1] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), 0u);
                    s_rayq[ring_at(rq_head, rq_count + lane_prefix(em))] = (uint8_t)bslot;
                }
                rq_count += (uint32_t)__popcll(em);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    // idle lanes take the next rays of the ray queue
        if (rq_count > 0u && n_idle > 0u) {
            const uint64_t idle = __ballot(!alive);
            const uint32_t take = n_idle < rq_count ? n_idle : rq_count;
            const uint32_t rank = lane_prefix(idle);
            if (!alive && rank < take) {
                slot = s_rayq[ring_at(rq_head, rank)];
                const uint4 r0 = s_rec[slot][0], r1 = s_rec[slot][1];
                s.ox = __uint_as_float(r0.x); s.oy = __uint_as_float(r0.y); s.oz = __uint_as_float(r0.z);
                s.dx = __uint_as_float(r1.x); s.dy = __uint_as_float(r1.y); s.dz = __uint_as_float(r1.z);
                any = r0.w != 0u;
                s.idx = safe_rcp_dir<false>(s.dx); s.idy = safe_rcp_dir<false>(s.dy); s.idz = safe_rcp_dir<false>(s.dz);
                const uint32_t oct = (s.dx < 0.0f ? 4u : 0u) | (s.dy < 0.0f ? 2u : 0u) | (s.dz < 0.0f ? 1u : 0u);
                s.oct_inv4 = (7u - oct) * 0x01010101u;
                s.bt = tmax_ray; s.bu = 0.0f; s.bv = 0.0f; s.bprim = kMissPrim; s.binst = kMissPrim;
                s.cur = make_uint2(0u, 0x80000000u);
                s.ptri = make_uint2(0u, 0u);
                s.sp = 0; s.base = 0;
                alive = true; has_prim = false; pidx = 0u;
                advance_select();                   // the root becomes this lane's next node
            }
            rq_head = ring_at(rq_head, take); rq_count -= take;
        }
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
        if (n_alive == 0u) {
            if (hq_count == 0u && rq_count == 0u && (exhausted || n_alloc >= (uint32_t)kPipeSlots)) break;    // nothing in flight, nothing to start: done
            continue;      // the next exchange shades what is queued or opens new slots (each such batch hands out a slot or finds the tile used up)
        }
        // the next exchange comes when this many lanes have finished: a few per cent of the lanes in the bulk of the render,
        // every single one at its end, when the wave has only a few pixels left and each waits for its own chain of rays
        uint32_t exchange_after = (n_alive + rq_count + hq_count) / 8u;
        exchange_after = exchange_after < 1u ? 1u : (exchange_after > (uint32_t)a.refill_threshold ? (uint32_t)a.refill_threshold : exchange_after);

        // =====================================================================================================
        // traverse until enough lanes have finished to make an exchange worthwhile
        // =====================================================================================================
        // destinations of the traversal loads: a lane's registers change only when it takes part in a load (dead across the exchange)
        f32x4 rpa = {0.0f, 0.0f, 0.0f, 0.0f}, rpb = rpa, rpc = rpa;
        u32x4 rn0 = {0u, 0u, 0u, 0u}, rn1 = rn0, rn2 = rn0, rn3 = rn0, rn4 = rn0;
        for (;;) {
            // ---- G. fetch what the lanes need next: primitives first, nodes second -- for the lanes that need one only ----
            // (wave-uniform masks; when no lane has a node to fetch, lane 0 fetches one all the same: the node loads are then
            // ALWAYS the five youngest vector-memory operations at the primitives' wait, whose vmcnt(5) is counted by hand)
            const uint64_t mask_p = __ballot(has_prim), mask_n0 = __ballot(has_node);
            const uint64_t mask_n = mask_n0 != 0ull ? mask_n0 : 1ull;
            {
                // both addresses first: whatever they depend on (a stack entry read back from scratch ...) is waited for
                // here, not between the two groups of loads
                const char *pp = prim_bytes + (size_t)pidx * a.prim_stride, *np = node_bytes + (size_t)nidx * a.node_stride;
                asm volatile("" : "+v"(pp), "+v"(np));
                if (mask_p != 0ull) issue_prim_loads_masked(mask_p, pp, rpa, rpb, rpc);
                issue_node_loads_masked(mask_n, np, rn0, rn1, rn2, rn3, rn4);
            }
#ifdef HRT_EXP_LOAD2
            // experiment: five more 16-byte loads per lane and iteration; waited for with the node
            u32x4 xn0, xn1, xn2, xn3, xn4;
#ifdef HRT_EXP_LOAD2_SAME
            issue_node_loads(node_bytes + (size_t)nidx * a.node_stride, xn0, xn1, xn2, xn3, xn4);     // the same node again: L1 hits only
#else
            issue_node_loads(node_bytes + (size_t)((nidx * 2654435761u) % 140000u) * a.node_stride, xn0, xn1, xn2, xn3, xn4);
#endif
#endif
            bool done = false;
#ifdef HRT_LANE_STATS
            { ++ls_iter; ls_alive += __popcll(__ballot(alive)); ls_node += __popcll(mask_n0); ls_prim += __popcll(mask_p); ls_ppass += mask_p != 0ull; }
#endif
            // ---- C. leaf test: waits for the primitive pieces only (the node loads issued behind them stay in flight) ----
            if (mask_p != 0ull) {
#ifdef HRT_EXP_LOAD2
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rpa), "+v"(rpb), "+v"(rpc) :: "memory");
#else
                wait_prim_loads(rpa, rpb, rpc);
#endif
                if (alive && has_prim) {
                    const float4 pa = make_float4(rpa.x, rpa.y, rpa.z, rpa.w), pb = make_float4(rpb.x, rpb.y, rpb.z, rpb.w),
                                 pc = make_float4(rpc.x, rpc.y, rpc.z, rpc.w);
                    const bool better = test_prim<HAS_SPHERES>(pa, pb, pc, s, tmin, tmax_ray, a.inst_inv, a.inst_identity);
                    if (any && better) done = true;
                }
            }
            has_prim = false; pidx = 0u;

            // ---- A. node phase ----
            uint2 tri = make_uint2(0u, 0u);
#ifdef HRT_EXP_LOAD2
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn0), "+v"(xn1), "+v"(xn2), "+v"(xn3), "+v"(xn4) :: "memory");
            if (xn0.x == 0xdeadbeefu && xn1.x == xn2.y && xn3.x == xn4.w) nidx = 1u;     // never: keeps the loads
#endif
            wait_node_loads(rn0, rn1, rn2, rn3, rn4);
            if (alive && !done && has_node) {
                const uint4 n0 = make_uint4(rn0.x, rn0.y, rn0.z, rn0.w), n1 = make_uint4(rn1.x, rn1.y, rn1.z, rn1.w),
                            n2 = make_uint4(rn2.x, rn2.y, rn2.z, rn2.w), n3 = make_uint4(rn3.x, rn3.y, rn3.z, rn3.w),
                            n4 = make_uint4(rn4.x, rn4.y, rn4.z, rn4.w);
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
#ifdef HRT_EXP_VALU2
                // experiment: the same slab arithmetic once more on slightly different inputs; the result only feeds a bit that is always zero
                {
                    uint32_t hm2 = 0u; float acc2 = 0.0f;
                    const float bx = aox + 1.0f, by = aoy + 1.0f, bz = aoz + 1.0f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t qlox = h ? n2.y : n2.x, qloy = h ? n2.w : n2.z, qloz = h ? n3.y : n3.x;
                        const uint32_t qhix = h ? n3.w : n3.z, qhiy = h ? n4.y : n4.x, qhiz = h ? n4.w : n4.z;
                        const uint32_t xn = nx ? qhix : qlox, xf = nx ? qlox : qhix;
                        const uint32_t yn = ny ? qhiy : qloy, yf = ny ? qloy : qhiy;
                        const uint32_t zn = nz ? qhiz : qloz, zf = nz ? qloz : qhiz;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float tnx = fmaf(HRT_BYTE_F(xn, j), aix, bx), tfx = fmaf(HRT_BYTE_F(xf, j), aix, bx);
                            const float tny = fmaf(HRT_BYTE_F(yn, j), aiy, by), tfy = fmaf(HRT_BYTE_F(yf, j), aiy, by);
                            const float tnz = fmaf(HRT_BYTE_F(zn, j), aiz, bz), tfz = fmaf(HRT_BYTE_F(zf, j), aiz, bz);
                            const float tlo = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                            const float thi = fminf(fminf(tfx, tfy), fminf(tfz, s.bt));
                            if (tlo <= thi) hm2 |= 1u << (4 * h + j);
                            acc2 += tlo;
                        }
                    }
                    if (hm2 == 0x5au && acc2 == 12345.678f) hitmask |= 1u;        // practically never
                }
#endif
                s.cur = make_uint2(n1.x, (hitmask & 0xff000000u) | (e_imask >> 24));
                tri = make_uint2(n1.y, hitmask & 0x00ffffffu);
            }
            has_node = false; nidx = 0u;

            // ---- B. bookkeeping: pending leaf group, the primitive and the node of the next iteration ----
            if (alive && !done && tri.y != 0u) {
                if (s.ptri.y == 0u) s.ptri = tri;
                else {
                    if (s.sp < kLdsStack) s_stack[s.sp][tx] = tri; else spill[s.sp - kLdsStack] = tri;
                    ++s.sp;
                }
            }
            {
                // leaf pass: ONE per iteration, one primitive per lane out of its pending leaf group.  It is skipped
                // (wave-uniform) while few lanes have leaf work and none depends on it.
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
                if (done) { alive = false; finished = true; has_node = false; has_prim = false; nidx = 0u; pidx = 0u; }
            }
            const uint64_t act = __ballot(alive);
            if (act == 0ull) break;
            if ((uint32_t)__popcll(__ballot(finished)) >= exchange_after) break;
        }
    }
#ifdef HRT_LANE_STATS
    if (tx == 0u) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.path.rays_closest);
        atomicAdd(d + 6, ls_iter); atomicAdd(d + 7, ls_alive); atomicAdd(d + 8, ls_node); atomicAdd(d + 9, ls_prim);
        atomicAdd(d + 2, ls_ppass); atomicAdd(d + 3, ls_batches); atomicAdd(d + 4, ls_batch_lanes);
    }
#endif
    for (int off = 32; off > 0; off >>= 1) {
        cnt_closest += (uint32_t)__shfl_down((int)cnt_closest, off);
        cnt_any += (uint32_t)__shfl_down((int)cnt_any, off);
    }
    if (tx == 0u) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_closest), (unsigned long long)cnt_closest);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.path.rays_any), (unsigned long long)cnt_any);
    }
}

// one launch renders every sample of every pixel of the tile
// slots: pixels in flight per wave (64 in lanes + the rest in the queues): 80, 96, 112 or 128
void launch_paths(const TraverseArgs &a, bool has_spheres, int slots, uint32_t grid_blocks, hipStream_t s) {
    const dim3 g(grid_blocks), b(kTraverseBlock);
    if (has_spheres) {
        if (slots <= 80) hipLaunchKernelGGL((k_paths<true, 80>), g, b, 0, s, a);
        else if (slots <= 96) hipLaunchKernelGGL((k_paths<true, 96>), g, b, 0, s, a);
        else if (slots <= 112) hipLaunchKernelGGL((k_paths<true, 112>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_paths<true, 128>), g, b, 0, s, a);
    } else {
        if (slots <= 80) hipLaunchKernelGGL((k_paths<false, 80>), g, b, 0, s, a);
        else if (slots <= 96) hipLaunchKernelGGL((k_paths<false, 96>), g, b, 0, s, a);
        else if (slots <= 112) hipLaunchKernelGGL((k_paths<false, 112>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_paths<false, 128>), g, b, 0, s, a);
    }
}
// one-wave workgroups of k_paths that fit a CU's 160 KB of LDS
uint32_t paths_blocks_that_fit(int slots) {
    const int n = slots <= 80 ? 80 : slots <= 96 ? 96 : slots <= 112 ? 112 : 128;
    const uint32_t lds = (uint32_t)(kLdsStack * kTraverseBlock * 8 + n * 48 + 2 * n);
    return 163840u / lds;
}

}  // namespace hrt
