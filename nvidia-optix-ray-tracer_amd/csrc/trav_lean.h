// trav_lean.h -- the traversal step of the production path kernel (fused.hip), written for INSTRUCTION COUNT.
//
// What round 2 measured (profiles/r02_valu_issue_patterns_microbench.txt, r02_exp_bounds.txt): a gfx950 SIMD issues about one
// instruction of ANY kind per 2.4 cycles -- scalar instructions, mask saves / restores and branches count like vector ones --
// and the path kernel is bound by exactly that.  The traversal loop of k_traverse spends a third of its 580 instructions on
// scalar mask bookkeeping: every `if` on per-lane data is a mask save, a branch and a restore, every per-lane bool carried
// round the loop three scalar instructions to merge, the two-level stack (LDS, overflow in scratch) two branches per access and
// a FLAT load per pop.  This step does the same work -- same nodes, same primitives, same canonical hit -- with:
//   * two small LDS stacks instead of one mixed LDS + scratch stack: sibling groups (node work) and leaf groups (primitive
//     work) are popped independently, one pop each per iteration, no "what kind is the top entry" loop; the leaf stack needs no
//     overflow path at all (a lane whose leaf stack is full does not take a new node until a group has been consumed), the
//     node stack holds one group per tree level and trees deeper than that take round 1's kernel (hrt_api.cpp);
//   * the bookkeeping after the node step as one hand-written instruction sequence;
//   * "no work" encoded in the work index itself (nidx / pidx = kNoWork) instead of separate bools;
//   * the straight-line primitive test of trav_common.h;
//   * lazy pushes: the sibling group in hand goes to the stack only when a child group arrives while siblings remain.
#pragma once
#include "trav_common.h"

namespace hrt {

constexpr uint32_t kNoWork = 0xffffffffu;
static_assert(kFusedMaxDepth == 12, "the node stack below is sized for it");
constexpr int kNodeStackLds = kFusedMaxDepth;      // sibling groups per lane in LDS: one per tree level above the current node (deeper trees: another kernel)
constexpr int kLeafStackLds = 4;       // leaf groups per lane in LDS (never more: see lean_select)

struct LeanLane {
    TravState s;                 // ray, reciprocal direction, octant, best hit; s.cur = sibling group in hand, s.ptri = leaf group in hand
    int nsp, lsp;                // top of the node stack / entries on the leaf stack
    int base;                    // bottom of the node stack: entries below have been given away (tail splitting, fused.hip)
    uint32_t nidx, pidx;         // node / primitive to fetch next (kNoWork: none)
};

__device__ __forceinline__ void lean_reset(LeanLane &L) {
    L.s.cur = make_uint2(0u, 0u); L.s.ptri = make_uint2(0u, 0u);
    L.nsp = 0; L.base = 0; L.lsp = 0; L.nidx = kNoWork; L.pidx = kNoWork;
}

// a new ray: the root is its first node
__device__ __forceinline__ void lean_start(LeanLane &L, V3 o, V3 d, float tmax_ray) {
    TravState &s = L.s;
    s.ox = o.x; s.oy = o.y; s.oz = o.z; s.dx = d.x; s.dy = d.y; s.dz = d.z;
    s.idx = safe_rcp_dir<false>(s.dx); s.idy = safe_rcp_dir<false>(s.dy); s.idz = safe_rcp_dir<false>(s.dz);
    const uint32_t oct = (s.dx < 0.0f ? 4u : 0u) | (s.dy < 0.0f ? 2u : 0u) | (s.dz < 0.0f ? 1u : 0u);
    s.oct_inv4 = (7u - oct) * 0x01010101u;
    s.bt = tmax_ray; s.bu = 0.0f; s.bv = 0.0f; s.bprim = kMissPrim; s.binst = kMissPrim;
    lean_reset(L);
    L.nidx = 0u;
}

// the node step: slab test of the eight children (identical arithmetic to k_traverse); returns the children's sibling group and
// leaf group (y == 0 / <= 0xffffff: none)
__device__ __forceinline__ void lean_node(const TravState &s, float tmin, const u32x4 rn0, const u32x4 rn1, const u32x4 rn2, const u32x4 rn3,
                                          const u32x4 rn4, uint2 &child, uint2 &tri) {
    const float px = __uint_as_float(rn0.x), py = __uint_as_float(rn0.y), pz = __uint_as_float(rn0.z);
    const uint32_t e_imask = rn0.w;
    const float aix = __uint_as_float((e_imask & 0xffu) << 23) * s.idx;
    const float aiy = __uint_as_float(((e_imask >> 8) & 0xffu) << 23) * s.idy;
    const float aiz = __uint_as_float(((e_imask >> 16) & 0xffu) << 23) * s.idz;
    const float aox = (px - s.ox) * s.idx, aoy = (py - s.oy) * s.idy, aoz = (pz - s.oz) * s.idz;
    const bool nx = s.dx < 0.0f, ny = s.dy < 0.0f, nz = s.dz < 0.0f;
    uint32_t hitmask = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t meta4 = h ? rn1.w : rn1.z;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
        const uint32_t bit_index4 = (meta4 ^ (s.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t qlox = h ? rn2.y : rn2.x, qloy = h ? rn2.w : rn2.z, qloz = h ? rn3.y : rn3.x;
        const uint32_t qhix = h ? rn3.w : rn3.z, qhiy = h ? rn4.y : rn4.x, qhiz = h ? rn4.w : rn4.z;
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
    child = make_uint2(rn1.x, (hitmask & 0xff000000u) | (e_imask >> 24));
    tri = make_uint2(rn1.y, hitmask & 0x00ffffffu);
}

// the next node of a lane that has just been handed a sibling group (tail splitting): its nearest child -- step (7) of
// lean_bookkeeping_asm in C++
__device__ __forceinline__ void lean_pick_node(LeanLane &L) {
    TravState &s = L.s;
    const uint32_t hits_imask = s.cur.y;
    const uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
    s.cur.y &= ~(1u << bit);
    const uint32_t slot_index = (bit - 24u) ^ (s.oct_inv4 & 0xffu);
    L.nidx = s.cur.x + (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot_index));
}

// Hand-issued loads like issue_*_loads_masked (trav_common.h), addressed as uniform base + 32-bit byte offset per lane (one
// v_mul_lo_u32 per record instead of a 64-bit multiply-add and its operand moves).  Lanes outside `mask` load nothing and keep
// their registers; an empty mask is fine: the loads still count in vmcnt, in order.
__device__ __forceinline__ void issue_prim_loads_off(uint64_t mask, const void *base, uint32_t off, f32x4 &a, f32x4 &b, f32x4 &c) {
    uint64_t save;
    asm volatile("s_mov_b64 %3, exec\n\t"
                 "s_mov_b64 exec, %6\n\t"
                 "global_load_dwordx4 %0, %4, %5\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:32\n\t"
                 "s_mov_b64 exec, %3"
                 : "+v"(a), "+v"(b), "+v"(c), "=&s"(save) : "v"(off), "s"(base), "s"(mask) : "memory");
}
__device__ __forceinline__ void issue_node_loads_off(uint64_t mask, const void *base, uint32_t off, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
    uint64_t save;
    asm volatile("s_mov_b64 %5, exec\n\t"
                 "s_mov_b64 exec, %8\n\t"
                 "global_load_dwordx4 %0, %6, %7\n\t"
                 "global_load_dwordx4 %1, %6, %7 offset:16\n\t"
                 "global_load_dwordx4 %2, %6, %7 offset:32\n\t"
                 "global_load_dwordx4 %3, %6, %7 offset:48\n\t"
                 "global_load_dwordx4 %4, %6, %7 offset:64\n\t"
                 "s_mov_b64 exec, %5"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "=&s"(save) : "v"(off), "s"(base), "s"(mask) : "memory");
}

// Everything after the node step as ONE hand-written instruction sequence: file the node's two groups, decide on the leaf pass,
// choose the primitive and the node of the next iteration, report finished rays.  (Written in C++ this is ~250 instructions,
// two thirds of them mask bookkeeping: every `if` on per-lane data a mask save, a branch and a restore; here it is 97.)
//   (1) the node's leaf group (tri: first primitive, bit per primitive that may be hit) goes into the hand (s.ptri) if that is
//       empty, else onto the leaf stack -- room is guaranteed by (7)
//   (2) the node's sibling group (child: first child, hit bits 31..24 | inner mask) becomes the group in hand (s.cur); siblings
//       still in hand go to the node stack first (lazy push)
//   (3) leaf pass?  (4) lanes with a leaf group in hand take one primitive of it  (5) an empty hand takes the top of the leaf stack
//   (6) a group in hand without hits left is replaced by the top of the node stack (entries [base, nsp) are the lane's own:
//       the bottom may have been given away, tail splitting)
//   (7) the next node: the nearest child of the group in hand -- unless the leaf stack could not take another group: such a
//       lane waits with its node work until primitives have been consumed, which is why the leaf stack needs no overflow path
//   (8) nothing left to do?
// Runs for the lanes that are active at the call (alive, not finished by an any-hit); inactive lanes keep their registers.  The
// node stack has NO overflow path either: the caller guarantees a tree of at most kNodeStackLds levels below the root.
//   ldsn / ldsl: byte address in LDS of this lane's column of the node / leaf stack (entries are 512 bytes apart)
//   pct, quorum: leaf passes are skipped while fewer than pct % of the active lanes have leaf work and fewer than `quorum`
//   lanes have nothing else to do (those wait)
// Returns 1 in the lanes whose ray has nothing left to do.
__device__ __forceinline__ uint32_t lean_bookkeeping_asm(LeanLane &L, uint2 child, uint2 tri, uint32_t ldsn, uint32_t ldsl, uint32_t pct, uint32_t quorum, uint32_t hold = 4u) {
    static_assert(kLeafStackLds == 4 && kTraverseBlock == 64 && sizeof(uint2) == 8,
                  "the sequence below has the leaf stack's depth (4, 'about to fill' = 3) and the stacks' row pitch (64 lanes x 8 bytes = 1 << 9) as literals");
    uint32_t fin, t0, t1, t2, c0, c1;
    uint64_t sv0, sv1, sv2, c2;
    const uint32_t k24 = 0x00ffffffu;
    asm volatile(
        "s_mov_b64 %[sv0], exec\n\t"
        // (1) the leaf group of this node: into the hand if it is free, else onto the leaf stack
        "v_cmp_ne_u32_e32 vcc, 0, %[ty]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[py]\n\t"
        "v_cndmask_b32_e32 %[px], %[px], %[tx], vcc\n\t"
        "v_cndmask_b32_e32 %[py], %[py], %[ty], vcc\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "v_lshl_add_u32 %[t0], %[lsp], 9, %[ldsl]\n\t"
        "ds_write2_b32 %[t0], %[tx], %[ty] offset1:1\n\t"
        "v_add_u32_e32 %[lsp], 1, %[lsp]\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        // (2) the sibling group of this node's children becomes the group in hand; siblings still in hand go to the stack first
        "v_cmp_lt_u32_e32 vcc, %[k24], %[chy]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_cmp_lt_u32_e32 vcc, %[k24], %[cy]\n\t"
        "s_mov_b64 %[sv1], exec\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_lshl_add_u32 %[t0], %[nsp], 9, %[ldsn]\n\t"
        "ds_write2_b32 %[t0], %[cx], %[cy] offset1:1\n\t"
        "v_add_u32_e32 %[nsp], 1, %[nsp]\n\t"
        "s_mov_b64 exec, %[sv1]\n\t"
        "v_mov_b32_e32 %[cx], %[chx]\n\t"
        "v_mov_b32_e32 %[cy], %[chy]\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        // (3) leaf pass?  sv1: lanes with leaf work; sv2: lanes WITHOUT node work.  Yes when pct % of the lanes have leaf work, or
        //     `quorum` lanes have nothing else to do, or a leaf stack is about to fill, or no lane has node work
        "v_cmp_ne_u32_e32 vcc, 0, %[py]\n\t"
        "s_mov_b64 %[sv1], vcc\n\t"
        "v_cmp_ge_u32_e32 vcc, %[k24], %[cy]\n\t"
        "v_cmp_eq_u32_e64 %[sv2], %[base], %[nsp]\n\t"
        "s_and_b64 %[sv2], %[sv2], vcc\n\t"
        "s_bcnt1_i32_b64 %[c0], %[sv1]\n\t"
        "s_bcnt1_i32_b64 %[c1], exec\n\t"
        "s_mulk_i32 %[c0], 0x64\n\t"
        "s_mul_i32 %[c1], %[c1], %[pct]\n\t"
        "s_cmp_ge_u32 %[c0], %[c1]\n\t"
        "s_cselect_b64 vcc, -1, 0\n\t"
        "s_xor_b64 %[c2], %[sv2], exec\n\t"                // lanes with node work
        "s_cmp_eq_u64 %[c2], 0\n\t"
        "s_cselect_b64 %[c2], -1, 0\n\t"
        "s_or_b64 vcc, vcc, %[c2]\n\t"
        "s_and_b64 %[sv2], %[sv2], %[sv1]\n\t"             // lanes with nothing but leaf work
        "s_bcnt1_i32_b64 %[c0], %[sv2]\n\t"
        "s_cmp_ge_u32 %[c0], %[quorum]\n\t"
        "s_cselect_b64 %[sv2], -1, 0\n\t"
        "s_or_b64 %[sv2], %[sv2], vcc\n\t"
        "s_lshr_b32 %[c0], %[hold], 1\n\t"                 // "about to fill": one group below the hold (4 -> more than 2 queued, as ever; 1 -> any)
        "v_cmp_lt_u32_e32 vcc, %[c0], %[lsp]\n\t"
        "s_and_b64 vcc, vcc, %[sv1]\n\t"                   // a leaf stack about to fill
        "s_cmp_lg_u64 vcc, 0\n\t"
        "s_cselect_b64 vcc, -1, 0\n\t"
        "s_or_b64 vcc, vcc, %[sv2]\n\t"
        "s_and_b64 vcc, vcc, %[sv1]\n\t"
        // (4) one primitive of the leaf group in hand
        "v_mov_b32_e32 %[pidx], -1\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_ffbl_b32_e32 %[t0], %[py]\n\t"
        "v_add_u32_e32 %[t1], -1, %[py]\n\t"
        "v_add_u32_e32 %[pidx], %[px], %[t0]\n\t"
        "v_and_b32_e32 %[py], %[py], %[t1]\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        // (5) an empty hand takes the top of the leaf stack
        "v_cmp_eq_u32_e32 vcc, 0, %[py]\n\t"
        "v_cmp_ne_u32_e64 %[sv1], 0, %[lsp]\n\t"
        "s_and_b64 exec, vcc, %[sv1]\n\t"
        "v_add_u32_e32 %[lsp], -1, %[lsp]\n\t"
        "v_lshl_add_u32 %[t0], %[lsp], 9, %[ldsl]\n\t"
        "ds_read_b32 %[px], %[t0]\n\t"
        "ds_read_b32 %[py], %[t0] offset:4\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        // (6) a group in hand without hits left is replaced by the top of the node stack
        "v_cmp_ge_u32_e32 vcc, %[k24], %[cy]\n\t"
        "v_cmp_ne_u32_e64 %[sv1], %[base], %[nsp]\n\t"
        "s_and_b64 exec, vcc, %[sv1]\n\t"
        "v_add_u32_e32 %[nsp], -1, %[nsp]\n\t"
        "v_lshl_add_u32 %[t0], %[nsp], 9, %[ldsn]\n\t"
        "ds_read_b32 %[cx], %[t0]\n\t"
        "ds_read_b32 %[cy], %[t0] offset:4\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // (7) the next node: the nearest child (highest hit bit, octant order) of the group in hand -- unless the leaf stack is full
        "v_mov_b32_e32 %[nidx], -1\n\t"
        "v_cmp_lt_u32_e32 vcc, %[k24], %[cy]\n\t"
        "v_cmp_gt_u32_e64 %[sv1], %[hold], %[lsp]\n\t"
        "s_and_b64 exec, vcc, %[sv1]\n\t"
        "v_ffbh_u32_e32 %[t0], %[cy]\n\t"
        "v_sub_u32_e32 %[t0], 31, %[t0]\n\t"
        "v_lshlrev_b32_e64 %[t1], %[t0], 1\n\t"
        "v_add_u32_e32 %[t0], -24, %[t0]\n\t"
        "v_and_b32_e32 %[t2], 0xff, %[oct]\n\t"
        "v_xor_b32_e32 %[t0], %[t0], %[t2]\n\t"
        "v_bfm_b32 %[t2], %[t0], 0\n\t"
        "v_and_b32_e32 %[t2], %[t2], %[cy]\n\t"
        "v_bcnt_u32_b32 %[nidx], %[t2], %[cx]\n\t"
        "v_xor_b32_e32 %[cy], %[cy], %[t1]\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        // (8) nothing left?
        "v_and_b32_e32 %[t0], %[nidx], %[pidx]\n\t"
        "v_sub_u32_e32 %[t2], %[nsp], %[base]\n\t"
        "v_or3_b32 %[t1], %[py], %[t2], %[lsp]\n\t"
        "v_cmp_eq_u32_e32 vcc, -1, %[t0]\n\t"
        "v_cmp_eq_u32_e64 %[sv1], 0, %[t1]\n\t"
        "s_and_b64 %[sv1], %[sv1], vcc\n\t"
        "v_cmp_ge_u32_e32 vcc, %[k24], %[cy]\n\t"
        "s_and_b64 vcc, vcc, %[sv1]\n\t"
        "v_cndmask_b32_e64 %[fin], 0, 1, vcc"
        : [cx] "+v"(L.s.cur.x), [cy] "+v"(L.s.cur.y), [px] "+v"(L.s.ptri.x), [py] "+v"(L.s.ptri.y), [nsp] "+v"(L.nsp), [lsp] "+v"(L.lsp),
          [nidx] "+v"(L.nidx), [pidx] "+v"(L.pidx), [fin] "=&v"(fin), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [sv0] "=&s"(sv0), [sv1] "=&s"(sv1), [sv2] "=&s"(sv2), [c2] "=&s"(c2), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [chx] "v"(child.x), [chy] "v"(child.y), [tx] "v"(tri.x), [ty] "v"(tri.y), [oct] "v"(L.s.oct_inv4), [base] "v"(L.base), [ldsn] "v"(ldsn), [ldsl] "v"(ldsl),
          [k24] "s"(k24), [pct] "s"(pct), [quorum] "s"(quorum), [hold] "s"(hold)
        : "vcc", "scc", "memory");
    return fin;
}

}  // namespace hrt
