// fmi_seed.hip — FM-index SMEM search for gfx950 (MI355X).
//
// What is computed (reference semantics, /root/reference):
//   round 1  getSMEMsAllPosOneThread        src/FMI_search.cpp:1608-1660
//   round 2  getSMEMsOnePosOneThread        src/FMI_search.cpp:1372-1606 on the pivots chosen
//            by mem_collect_smem            src/bwamem.cpp:721-751
//   round 3  bwtSeedStrategyAllPosOneThread src/FMI_search.cpp:1662-1816
//   each step is one backwardExt            src/FMI_search.cpp:2029-2056 over CP_OCC blocks
//
// How it is mapped to the machine (DESIGN.md §"SMEM kernel"):
//   The search is a chain of dependent random 64-byte block reads (one or two per
//   extension), ~460 per read.  Throughput therefore comes from the number of
//   independent chains in flight, not from lanes cooperating on one chain: every
//   LANE owns one read and runs a small state machine whose every iteration performs
//   exactly one extension, whatever phase (forward / backward / round 3) the lane is
//   in, so divergent phases still share one memory round trip.  Lanes pull the next
//   read from a global cursor when they finish (no tail of idle lanes), the grid is
//   persistent and sized to the chip, and the per-lane list of "previous" intervals
//   keeps its 8 newest entries in an LDS ring, the rest in a lane-contiguous HBM list.
#include "fmi_kernels.h"
#include "wave_ops.h"

namespace bwams {

namespace {

constexpr int kBlock = 256;
constexpr int kBlocksPerCU = 6;

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) {
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

struct Occ4 {
    int64_t v[4];
};

// ---- quad-cooperative block fetch --------------------------------------------------
// A lane that reads its own 64-byte block with four 16-byte loads costs four L2 requests
// per block; the memory system then tops out at ~28 G blocks/s (tools/ubench_gather, mode 0).
// Instead the four lanes of a quad fetch four blocks together: in load j every lane of the
// quad reads the 16-byte piece (lane & 3) of quad member j's block, so one wave instruction
// is 16 fully used 64-byte requests; a 4x4 register transpose inside the quad (DPP
// quad_perm, no LDS) then hands every lane the whole block it asked for.  This shape
// reaches the random-line ceiling of HBM (~50 G blocks/s, mode 1 of the same benchmark).
template <int CTRL>
__device__ __forceinline__ uint32_t qdpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int J>
__device__ __forceinline__ int64_t quad_bcast64(int64_t v) {
    const uint32_t lo = qdpp<J * 0x55>((uint32_t)v), hi = qdpp<J * 0x55>((uint32_t)((uint64_t)v >> 32));
    return (int64_t)mk64(lo, hi);
}

// In place 4x4 transpose across the quad: on entry lane q holds m[j] = element (q, j);
// on exit lane q holds m[j] = element (j, q).
__device__ __forceinline__ void quad_transpose(uint32_t &m0, uint32_t &m1, uint32_t &m2, uint32_t &m3, int q) {
    const bool hi2 = (q & 2) != 0, hi1 = (q & 1) != 0;
    // exchange 2x2 blocks with lane ^ 2
    uint32_t t0 = hi2 ? m0 : m2, t1 = hi2 ? m1 : m3;
    t0 = qdpp<0x4E>(t0);           // quad_perm [2,3,0,1]
    t1 = qdpp<0x4E>(t1);
    if (hi2) { m0 = t0; m1 = t1; } else { m2 = t0; m3 = t1; }
    // exchange inside the 2x2 blocks with lane ^ 1
    t0 = hi1 ? m0 : m1;
    t1 = hi1 ? m2 : m3;
    t0 = qdpp<0xB1>(t0);           // quad_perm [1,0,3,2]
    t1 = qdpp<0xB1>(t1);
    if (hi1) { m0 = t0; m2 = t1; } else { m1 = t0; m3 = t1; }
}

__device__ __forceinline__ void quad_transpose4(uint4 &a0, uint4 &a1, uint4 &a2, uint4 &a3, int q) {
    quad_transpose(a0.x, a1.x, a2.x, a3.x, q);
    quad_transpose(a0.y, a1.y, a2.y, a3.y, q);
    quad_transpose(a0.z, a1.z, a2.z, a3.z, q);
    quad_transpose(a0.w, a1.w, a2.w, a3.w, q);
}

__device__ __forceinline__ void occ_from_block(const uint4 &c01, const uint4 &c23, const uint4 &h01,
                                               const uint4 &h23, int64_t pos, Occ4 &o) {
    const int y = (int)(pos & 63);
    const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
    o.v[0] = (int64_t)mk64(c01.x, c01.y) + __popcll(mk64(h01.x, h01.y) & mask);
    o.v[1] = (int64_t)mk64(c01.z, c01.w) + __popcll(mk64(h01.z, h01.w) & mask);
    o.v[2] = (int64_t)mk64(c23.x, c23.y) + __popcll(mk64(h23.x, h23.y) & mask);
    o.v[3] = (int64_t)mk64(c23.z, c23.w) + __popcll(mk64(h23.z, h23.w) & mask);
}

// ---- the compact table (BWAMS_CP2=1: a resident layout for the search kernels; the files and every other kernel keep CP_OCC) -----
// CpOcc2: the same information as two adjacent CP_OCC blocks in 64 bytes — the four counts at the start of a 128-base block and the
// bases as two bit planes (bit 63 - j of word 0 = base j, of word 1 = base 64 + j; hi: G or T, lo: C or T):
//     piece 0 = {count A, count C}, piece 1 = {count G, count T}, piece 2 = {hi word 0, hi word 1}, piece 3 = {lo word 0, lo word 1}.
// k and k + s then share a block whenever s < 128 (one request instead of two), and the table is half as large.  The row of the sentinel
// has no bit in any one-hot string; in the planes it reads as an A: Occ(A) is corrected in its block.  Algorithmic bytes stay the
// reference layout's count: the kernels count blocks of 64 rows whatever table they read.
__device__ __forceinline__ void occ_from_block2(const uint4 &c01, const uint4 &c23, const uint4 &hp, const uint4 &lp, int64_t pos,
                                                int64_t sentinel, Occ4 &o) {
    const int y = (int)(pos & 127);
    const int y0 = y < 64 ? y : 64, y1 = y - y0;
    const uint64_t m0 = y0 ? (~0ull << (64 - y0)) : 0ull, m1 = y1 ? (~0ull << (64 - y1)) : 0ull;
    const uint64_t h0 = mk64(hp.x, hp.y) & m0, h1 = mk64(hp.z, hp.w) & m1;
    const uint64_t l0 = mk64(lp.x, lp.y), l1 = mk64(lp.z, lp.w);
    const int nt = __popcll(h0 & l0) + __popcll(h1 & l1);
    const int ng = __popcll(h0 & ~l0) + __popcll(h1 & ~l1);
    const int nc = __popcll(~mk64(hp.x, hp.y) & l0 & m0) + __popcll(~mk64(hp.z, hp.w) & l1 & m1);
    int na = y - nt - ng - nc;
    if ((pos >> 7) == (sentinel >> 7) && y > (int)(sentinel & 127)) --na;
    o.v[0] = (int64_t)mk64(c01.x, c01.y) + na;
    o.v[1] = (int64_t)mk64(c01.z, c01.w) + nc;
    o.v[2] = (int64_t)mk64(c23.x, c23.y) + ng;
    o.v[3] = (int64_t)mk64(c23.z, c23.w) + nt;
}
template <int TAB>
__device__ __forceinline__ void occ_any(const DevFmi &f, const uint4 &p0, const uint4 &p1, const uint4 &p2, const uint4 &p3, int64_t pos, Occ4 &o) {
    if (TAB == 1) occ_from_block2(p0, p1, p2, p3, pos, f.sentinel, o);
    else occ_from_block(p0, p1, p2, p3, pos, o);
}

// ---- the interleaved table (TAB == 2; BWAMS_CP2=2, the default) ---------------------------------------------------------------------
// The same 64 bytes per 64 rows as CP_OCC, the fields permuted: piece b (16 bytes) = {cp_count[b], one_hot_bwt_str[b]}, b = A, C, G, T.
// backwardExt(a) needs Occ of base a at both ends (k' and s') and, for l', the sizes of the bases ABOVE a — or, because the four sizes add
// up to s minus the sentinel, the bases up to a:
//     l'[0] = l + s - s0        l'[1] = l + s - s0 - s1        l'[2] = l + [sentinel] + s3        l'[3] = l + [sentinel]
// (FMI_search.cpp:2029-2056 with s0 + s1 + s2 + s3 = s - [sentinel in range]: exact integer identities).  So an extension by a reads only
// the HALF of each block that holds a's pair of bases — {A, C} or {G, T}: 32 bytes — fetched by a PAIR of lanes (lane q of the pair
// reads piece q of both members' halves; one exchange with the neighbour gives a lane both pieces of its own half).  Against the
// quad-cooperative whole-block fetch: half the load instructions, a 2 x 2 exchange of 4 words instead of a 4 x 4 transpose of 16, two
// masked popcounts per end instead of four: 5.74 G -> 4.27 G vector instructions per round-1 launch, 16.8 -> 15.9 ms (profiles/r04_notes.md).
// The files and every other kernel keep the reference layout; the algorithmic byte count stays the reference layout's.
__device__ __forceinline__ int64_t cnt_at4(const DevFmi &f, int i) { return i == 0 ? f.count[0] : i == 1 ? f.count[1] : i == 2 ? f.count[2] : f.count[3]; }
struct HalfBlk { uint4 x, y; };            // piece (lane & 1) and piece 1 - (lane & 1) of the lane's half: bases 2h + q, 2h + 1 - q
template <int J>
__device__ __forceinline__ int64_t pair_bcast64(int64_t v) {      // member J of every pair: quad_perm [J, J, 2 + J, 2 + J]
    constexpr int C = J ? 0xF5 : 0xA0;
    const uint32_t lo = qdpp<C>((uint32_t)v), hi = qdpp<C>((uint32_t)((uint64_t)v >> 32));
    return (int64_t)mk64(lo, hi);
}
__device__ __forceinline__ uint4 pair_swap(const uint4 &v) {       // the neighbour's value: quad_perm [1, 0, 3, 2]
    return make_uint4(qdpp<0xB1>(v.x), qdpp<0xB1>(v.y), qdpp<0xB1>(v.z), qdpp<0xB1>(v.w));
}
// MUST be called by all 64 lanes.  idx = (block << 1) | half of the lanes that fetch; the lanes of a pair fetch both members' halves
__device__ __forceinline__ void pair_fetch(const uint4 *__restrict__ tab, bool fetch, int64_t idx, int q, HalfBlk &out) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 P0 = zero, P1 = zero;
    const int64_t i0 = pair_bcast64<0>(idx), i1 = pair_bcast64<1>(idx);
    const uint32_t n = fetch ? 1u : 0u;
    if (qdpp<0xA0>(n)) P0 = tab[(i0 << 1) + q];
    if (qdpp<0xF5>(n)) P1 = tab[(i1 << 1) + q];
    // the piece this lane holds for itself (load q) and the one it holds for its neighbour (load 1 - q)
    const uint4 own = q ? P1 : P0, give = q ? P0 : P1;
    const uint4 got = pair_swap(give);
    if (fetch) { out.x = own; out.y = got; }
}
__device__ __forceinline__ void occ_pair(const HalfBlk &h, int64_t pos, int64_t &ox, int64_t &oy) {
    const int y = (int)(pos & 63);
    const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
    ox = (int64_t)mk64(h.x.x, h.x.y) + __popcll(mk64(h.x.z, h.x.w) & mask);
    oy = (int64_t)mk64(h.y.x, h.y.y) + __popcll(mk64(h.y.z, h.y.w) & mask);
}
__device__ __forceinline__ void pair_finish(const DevFmi &f, int q, int64_t k, int64_t l, int64_t s, int a, int64_t oxs, int64_t oys, int64_t oxe,
                                            int64_t oye, int64_t &nk, int64_t &nl, int64_t &ns) {
    const bool a_is_x = (a & 1) == q;                            // x holds base 2h + q
    const int64_t occ_a = a_is_x ? oxs : oys;
    const int64_t sx = oxe - oxs, sy = oye - oys;
    const int64_t s_a = a_is_x ? sx : sy, s_o = a_is_x ? sy : sx;
    const int64_t sent = (k <= f.sentinel && k + s > f.sentinel) ? 1 : 0;
    nk = cnt_at4(f, a) + occ_a;
    ns = s_a;
    nl = a == 0 ? l + s - s_a : a == 1 ? l + s - s_a - s_o : a == 2 ? l + sent + s_o : l + sent;
}
struct HalfCache {
    HalfBlk a, b;
    int32_t ta, tb;                     // (block << 1 | half) held, -1 = none
};
// backwardExt over the interleaved table for every lane of the wave.  MUST be called by all 64 lanes.
__device__ __forceinline__ void backward_ext_pair(const DevFmi &f, bool need, int64_t k, int64_t l, int64_t s, int a,
                                                  int64_t &nk, int64_t &nl, int64_t &ns) {
    const int q = (int)(threadIdx.x & 1);
    const int64_t sp = need ? k : 0, ep = need ? k + s : 0;
    const int h = (a >> 1) & 1;
    const int64_t is = ((sp >> 6) << 1) | h, ie = ((ep >> 6) << 1) | h;
    const bool two = need && is != ie;
    HalfBlk A, B;
    A.x = A.y = B.x = B.y = make_uint4(0, 0, 0, 0);
    pair_fetch(f.cp2, need, is, q, A);
    if (__any(two)) pair_fetch(f.cp2, two, ie, q, B);
    int64_t oxs, oys, oxe, oye;
    occ_pair(A, sp, oxs, oys);
    occ_pair(two ? B : A, ep, oxe, oye);
    pair_finish(f, q, k, l, s, a, oxs, oys, oxe, oye, nk, nl, ns);
}
// ... with the lane's two most recent half blocks kept in registers (BlkCache's rule: start against previous start, end against previous end)
__device__ __forceinline__ void backward_ext_pair_cached(const DevFmi &f, HalfCache &c, bool need, int64_t k, int64_t l, int64_t s, int a,
                                                         int64_t &nk, int64_t &nl, int64_t &ns) {
    const int q = (int)(threadIdx.x & 1);
    const int64_t sp = need ? k : 0, ep = need ? k + s : 0;
    const int h = (a >> 1) & 1;
    const int32_t is = (int32_t)(((sp >> 6) << 1) | h), ie = (int32_t)(((ep >> 6) << 1) | h);
    const bool two = need && is != ie;
    const bool fa = need && is != c.ta, fb = two && ie != c.tb;
    if (__any(fa)) { pair_fetch(f.cp2, fa, (int64_t)is, q, c.a); if (fa) c.ta = is; }
    if (__any(fb)) { pair_fetch(f.cp2, fb, (int64_t)ie, q, c.b); if (fb) c.tb = ie; }
    int64_t oxs, oys, oxe, oye;
    occ_pair(c.a, sp, oxs, oys);
    occ_pair(two ? c.b : c.a, ep, oxe, oye);
    pair_finish(f, q, k, l, s, a, oxs, oys, oxe, oye, nk, nl, ns);
}

// backwardExt for every lane of the wave at once.  MUST be called by all 64 lanes
// (wave-uniform control flow); lanes without work pass need = false.
template <int TAB = 0>
__device__ __forceinline__ void backward_ext_coop(const DevFmi &f, bool need, int64_t k, int64_t l, int64_t s,
                                                  int a, int64_t &nk, int64_t &nl, int64_t &ns) {
    if (TAB == 2) { backward_ext_pair(f, need, k, l, s, a, nk, nl, ns); return; }
    constexpr int BS = TAB == 1 ? 7 : 6;
    const uint4 *const tab = TAB ? f.cp2 : f.cp;
    const int q = (int)(threadIdx.x & 3);
    const int64_t sp = need ? k : 0, ep = need ? k + s : 0;
    const bool two = need && ((sp >> BS) != (ep >> BS));
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 A0 = zero, A1 = zero, A2 = zero, A3 = zero;
    {
        const int64_t b0 = quad_bcast64<0>(sp) >> BS, b1 = quad_bcast64<1>(sp) >> BS;
        const int64_t b2 = quad_bcast64<2>(sp) >> BS, b3 = quad_bcast64<3>(sp) >> BS;
        const uint32_t n = need ? 1u : 0u;
        if (qdpp<0x00>(n)) A0 = tab[(b0 << 2) + q];
        if (qdpp<0x55>(n)) A1 = tab[(b1 << 2) + q];
        if (qdpp<0xAA>(n)) A2 = tab[(b2 << 2) + q];
        if (qdpp<0xFF>(n)) A3 = tab[(b3 << 2) + q];
    }
    uint4 B0 = zero, B1 = zero, B2 = zero, B3 = zero;
    const bool any_two = __any(two);
    if (any_two) {
        const int64_t b0 = quad_bcast64<0>(ep) >> BS, b1 = quad_bcast64<1>(ep) >> BS;
        const int64_t b2 = quad_bcast64<2>(ep) >> BS, b3 = quad_bcast64<3>(ep) >> BS;
        const uint32_t n = two ? 1u : 0u;
        if (qdpp<0x00>(n)) B0 = tab[(b0 << 2) + q];
        if (qdpp<0x55>(n)) B1 = tab[(b1 << 2) + q];
        if (qdpp<0xAA>(n)) B2 = tab[(b2 << 2) + q];
        if (qdpp<0xFF>(n)) B3 = tab[(b3 << 2) + q];
    }
    // after the transpose A0..A3 are pieces 0..3 of this lane's own block
    quad_transpose4(A0, A1, A2, A3, q);
    if (any_two) quad_transpose4(B0, B1, B2, B3, q);
    if (!two) { B0 = A0; B1 = A1; B2 = A2; B3 = A3; }
    Occ4 osp, oep;
    occ_any<TAB>(f, A0, A1, A2, A3, sp, osp);
    occ_any<TAB>(f, B0, B1, B2, B3, ep, oep);
    const int64_t s0 = oep.v[0] - osp.v[0], s1 = oep.v[1] - osp.v[1];
    const int64_t s2 = oep.v[2] - osp.v[2], s3 = oep.v[3] - osp.v[3];
    const int64_t l3 = l + ((k <= f.sentinel && k + s > f.sentinel) ? 1 : 0);
    const int64_t l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
    nk = (a == 0 ? f.count[0] + osp.v[0] : a == 1 ? f.count[1] + osp.v[1]
          : a == 2 ? f.count[2] + osp.v[2] : f.count[3] + osp.v[3]);
    ns = a == 0 ? s0 : a == 1 ? s1 : a == 2 ? s2 : s3;
    nl = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}

// backwardExt with the lane's two most recent blocks kept in registers.  Four extensions in five belong to the backward
// phase (642 M of 818 M per million reads in rounds 1-2), where the entries of a column are NESTED intervals visited from the
// innermost outwards: the block holding k is one of the previous entry's blocks for 43 % of them, the block holding k + s for
// 35 % of those that need a second block (counted on the bench reads, profiles/r03_notes.md).  Those fetches were L1 / L2 hits,
// but requests all the same, and requests per second — not bytes, not lines — are what the memory system runs out of under this
// kernel.  The cache is role-bound (start block against the previous start block, end block against the previous end block), so
// a hit moves no data: the lane simply takes no part in the quad's fetch of its own block.  MUST be called by all 64 lanes.
struct BlkCache {
    uint4 a0, a1, a2, a3, b0, b1, b2, b3;
    int32_t ta, tb;                     // block numbers held (rows >> 6 < 2^30), -1 = none
};
template <int TAB = 0>
__device__ __forceinline__ void backward_ext_cached(const DevFmi &f, BlkCache &c, bool need, int64_t k, int64_t l, int64_t s,
                                                    int a, int64_t &nk, int64_t &nl, int64_t &ns) {
    constexpr int BS = TAB == 1 ? 7 : 6;
    const uint4 *const tab = TAB ? f.cp2 : f.cp;
    const int q = (int)(threadIdx.x & 3);
    const int64_t sp = need ? k : 0, ep = need ? k + s : 0;
    const int32_t bs = (int32_t)(sp >> BS), be = (int32_t)(ep >> BS);
    const bool two = need && bs != be;
    const bool fa = need && bs != c.ta, fb = two && be != c.tb;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (__any(fa)) {
        uint4 A0 = zero, A1 = zero, A2 = zero, A3 = zero;
        const int64_t b0 = quad_bcast64<0>(sp) >> BS, b1 = quad_bcast64<1>(sp) >> BS;
        const int64_t b2 = quad_bcast64<2>(sp) >> BS, b3 = quad_bcast64<3>(sp) >> BS;
        const uint32_t n = fa ? 1u : 0u;
        if (qdpp<0x00>(n)) A0 = tab[(b0 << 2) + q];
        if (qdpp<0x55>(n)) A1 = tab[(b1 << 2) + q];
        if (qdpp<0xAA>(n)) A2 = tab[(b2 << 2) + q];
        if (qdpp<0xFF>(n)) A3 = tab[(b3 << 2) + q];
        quad_transpose4(A0, A1, A2, A3, q);
        if (fa) { c.a0 = A0; c.a1 = A1; c.a2 = A2; c.a3 = A3; c.ta = bs; }
    }
    if (__any(fb)) {
        uint4 B0 = zero, B1 = zero, B2 = zero, B3 = zero;
        const int64_t b0 = quad_bcast64<0>(ep) >> BS, b1 = quad_bcast64<1>(ep) >> BS;
        const int64_t b2 = quad_bcast64<2>(ep) >> BS, b3 = quad_bcast64<3>(ep) >> BS;
        const uint32_t n = fb ? 1u : 0u;
        if (qdpp<0x00>(n)) B0 = tab[(b0 << 2) + q];
        if (qdpp<0x55>(n)) B1 = tab[(b1 << 2) + q];
        if (qdpp<0xAA>(n)) B2 = tab[(b2 << 2) + q];
        if (qdpp<0xFF>(n)) B3 = tab[(b3 << 2) + q];
        quad_transpose4(B0, B1, B2, B3, q);
        if (fb) { c.b0 = B0; c.b1 = B1; c.b2 = B2; c.b3 = B3; c.tb = be; }
    }
    Occ4 osp, oep;
    occ_any<TAB>(f, c.a0, c.a1, c.a2, c.a3, sp, osp);
    occ_any<TAB>(f, two ? c.b0 : c.a0, two ? c.b1 : c.a1, two ? c.b2 : c.a2, two ? c.b3 : c.a3, ep, oep);
    const int64_t s0 = oep.v[0] - osp.v[0], s1 = oep.v[1] - osp.v[1];
    const int64_t s2 = oep.v[2] - osp.v[2], s3 = oep.v[3] - osp.v[3];
    const int64_t l3 = l + ((k <= f.sentinel && k + s > f.sentinel) ? 1 : 0);
    const int64_t l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
    nk = (a == 0 ? f.count[0] + osp.v[0] : a == 1 ? f.count[1] + osp.v[1]
          : a == 2 ? f.count[2] + osp.v[2] : f.count[3] + osp.v[3]);
    ns = a == 0 ? s0 : a == 1 ? s1 : a == 2 ? s2 : s3;
    nl = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}

// the register cache that goes with a table kind, and the cached extension over it
template <int TAB> struct CacheOf { using type = BlkCache; };
template <> struct CacheOf<2> { using type = HalfCache; };
__device__ __forceinline__ void cache_init(BlkCache &c) {
    c.a0 = c.a1 = c.a2 = c.a3 = c.b0 = c.b1 = c.b2 = c.b3 = make_uint4(0, 0, 0, 0);
    c.ta = c.tb = -1;
}
__device__ __forceinline__ void cache_init(HalfCache &c) {
    c.a.x = c.a.y = c.b.x = c.b.y = make_uint4(0, 0, 0, 0);
    c.ta = c.tb = -1;
}
template <int TAB>
__device__ __forceinline__ void ext_cached(const DevFmi &f, typename CacheOf<TAB>::type &c, bool need, int64_t k, int64_t l, int64_t s, int a,
                                           int64_t &nk, int64_t &nl, int64_t &ns) {
    if constexpr (TAB == 2) backward_ext_pair_cached(f, c, need, k, l, s, a, nk, nl, ns);
    else backward_ext_cached<TAB>(f, c, need, k, l, s, a, nk, nl, ns);
}

// count[i] without dynamic indexing of the kernel argument (keeps it in SGPRs)
__device__ __forceinline__ int64_t cnt_at(const DevFmi &f, int i) {
    return i == 0 ? f.count[0] : i == 1 ? f.count[1] : i == 2 ? f.count[2] : i == 3 ? f.count[3] : f.count[4];
}

// Work tickets: a wave reserves kTicketChunk item indices with ONE atomic and hands them to its
// lanes as they finish (ballot-ranked); a lane that finds the reservation empty retries in the
// next iteration.  (One atomic per item on a single word tops out near 90 M/s.)
//
// `guided` (round 1): the reservation shrinks towards the end of the queue.  A wave that took the last 64 reads just before the
// queue ran dry keeps 63 of them for its own lanes while its neighbours leave; a reservation of what is left, shared among
// all waves twice over (between kMinTicketChunk and 64; `seen` is the cursor after the wave's previous reservation, a stale
// but safe estimate), ends round 1 0.7 ms earlier (profiles/r03_notes.md 87).  Rounds 2 and 3 keep 64: their items are a
// third of a read's work, every reservation is an atomic round trip the whole wave waits for, and the same rule cost them
// 1.0 and 1.4 ms.
constexpr int kTicketChunk = 64;
#ifndef BWAMS_MIN_TICKET_CHUNK
#define BWAMS_MIN_TICKET_CHUNK 2
#endif
constexpr int kMinTicketChunk = BWAMS_MIN_TICKET_CHUNK;
struct WaveTickets {
    unsigned long long next;   // wave-uniform
    int left;                  // wave-uniform
    unsigned long long seen;   // wave-uniform: the cursor after this wave's last reservation
};
// MUST be called by all 64 lanes.  Returns true and sets `ticket` for the lanes that were served.
__device__ __forceinline__ bool take_ticket(unsigned long long *head, WaveTickets &wt, bool want,
                                            unsigned long long &ticket, int64_t n_items, bool guided) {
    const unsigned long long m = __ballot(want);
    if (!m) return false;
    const int lane = (int)(threadIdx.x & 63);
    if (wt.left == 0) {
        const long long rem = (long long)n_items - (long long)wt.seen;
        const long long share = rem / (2ll * (long long)gridDim.x * (kBlock / 64));
        const int chunk = !guided || share >= kTicketChunk ? kTicketChunk : share <= kMinTicketChunk ? kMinTicketChunk : (int)share;
        wt.next = wave_ticket(head, (unsigned long long)chunk);          // out of line: see wave_ops.h
        wt.left = chunk;
        wt.seen = wt.next + (unsigned long long)chunk;
    }
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int cnt = __popcll(m);
    const int served = cnt < wt.left ? cnt : wt.left;
    const bool got = want && rank < served;
    ticket = wt.next + (unsigned long long)rank;
    wt.next += (unsigned long long)served;
    wt.left -= served;
    return got;
}

// ---- SMEM output: per-wave chunks ----------------------------------------------------
// One global cursor bumped per emitted SMEM serialises the whole chip on a single address
// (measured: 2/3 of the round-3 kernel).  Instead a wave reserves chunks of kChunk pool
// slots with ONE atomic and fills them with ballot-ranked stores; the unused tail of an
// abandoned chunk is stamped with rid = kHoleRid, which sorts behind every real read and
// is dropped after the sort.  The number of real SMEMs is added once per wave at exit.
constexpr int kChunk = 64;
constexpr uint32_t kHoleRid = 0xffffffffu;

struct WaveOut {
    long long base;      // first slot of the current chunk (wave-uniform), -1 = none
    int used;            // slots used in it (wave-uniform)
    unsigned long long emitted;   // this lane's emitted count (summed at exit)
};

__device__ __forceinline__ void wave_close_chunk(const SeedLaunch &a, WaveOut &w) {
    const int lane = (int)(threadIdx.x & 63);
    if (w.base >= 0) {
        const long long slot = w.base + w.used + lane;
        if (w.used + lane < kChunk && slot < a.pool_cap) a.pool[slot].rid = kHoleRid;
    }
    w.base = -1;
    w.used = 0;
}

// MUST be called by all 64 lanes at a wave-uniform point.
__device__ __forceinline__ void wave_emit(const SeedLaunch &a, WaveOut &w, bool flag, uint32_t rid, uint32_t m,
                                          uint32_t n, int64_t k, int64_t l, int64_t s) {
    if (a.debug & 1) flag = false;             // diagnostic ablation only (BWAMS_DEBUG=1)
    const unsigned long long mask = __ballot(flag);
    if (!mask) return;
    const int lane = (int)(threadIdx.x & 63);
    const int cnt = __popcll(mask);
    if (w.base < 0 || w.used + cnt > kChunk) {
        wave_close_chunk(a, w);
        w.base = (long long)wave_ticket(&a.ctr->n_smem_total, (unsigned long long)kChunk);   // out of line: see wave_ops.h
    }
    if (flag) {
        const long long slot = w.base + w.used + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < a.pool_cap) {
            bwams_smem_t r;
            r.rid = rid; r.m = m; r.n = n; r.pad_ = 0;
            r.k = k; r.l = l; r.s = s;
            a.pool[slot] = r;
        }
        w.emitted++;
    }
    w.used += cnt;
}

__device__ __forceinline__ void wave_emit_finish(const SeedLaunch &a, WaveOut &w, bool round3 = false) {
    wave_close_chunk(a, w);
    unsigned long long e = w.emitted;
    for (int o = 32; o > 0; o >>= 1) e += mk64(__shfl_down((uint32_t)e, o), __shfl_down((uint32_t)(e >> 32), o));
    if ((threadIdx.x & 63) == 0 && e) atomicAdd(round3 ? &a.ctr->n_smem3 : &a.ctr->n_smem_valid, e);
}

__device__ __forceinline__ void flush_counters(DevCounters *ctr, unsigned long long n_ext,
                                               unsigned long long n_blk, bool round3 = false) {
    for (int o = 32; o > 0; o >>= 1) {
        n_ext += mk64(__shfl_down((uint32_t)n_ext, o), __shfl_down((uint32_t)(n_ext >> 32), o));
        n_blk += mk64(__shfl_down((uint32_t)n_blk, o), __shfl_down((uint32_t)(n_blk >> 32), o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(round3 ? &ctr->n_ext3 : &ctr->n_ext, n_ext);
        atomicAdd(round3 ? &ctr->n_blk3 : &ctr->n_ext_blocks, n_blk);
    }
}

// Previous-interval list of one lane: `cap` entries of 16 bytes, contiguous per lane so that
// consecutive entries share cache lines and pages (a lane-interleaved layout costs four
// scattered requests per access and thrashes the TLB: profiles/r01_notes.md).  One entry is
// one 16-byte request:  x = k[31:0]  y = l[31:0]  z = s[31:0]
//                       w = n[15:0] | k[35:32] << 16 | l[35:32] << 20 | s[35:32] << 24
// (36-bit rows: texts up to 2^36 = 68 G rows; GRCh38 has 6.4 G.)
__device__ __forceinline__ uint4 prev_pack(int64_t k, int64_t l, int64_t s, int n) {
    const uint32_t w = (uint32_t)(n & 0xffff) | (((uint32_t)((uint64_t)k >> 32) & 0xf) << 16) |
                       (((uint32_t)((uint64_t)l >> 32) & 0xf) << 20) | (((uint32_t)((uint64_t)s >> 32) & 0xf) << 24);
    return make_uint4((uint32_t)k, (uint32_t)l, (uint32_t)s, w);
}
__device__ __forceinline__ void prev_unpack(const uint4 a, int64_t &k, int64_t &l, int64_t &s, int &n) {
    k = (int64_t)mk64(a.x, (a.w >> 16) & 0xf);
    l = (int64_t)mk64(a.y, (a.w >> 20) & 0xf);
    s = (int64_t)mk64(a.z, (a.w >> 24) & 0xf);
    n = (int)(a.w & 0xffff);
}

// The list with its head in LDS.  Entries are pushed at decreasing physical indices during the
// forward phase (ph = cap-1, cap-2, ...); the backward phase addresses them logically
// (p = physical - base, base = cap - num_prev, p = 0 is the last one pushed).  The kPrevLds most
// recently pushed entries — logical indices [0, kPrevLds) — live in an LDS ring (slot = physical
// index mod kPrevLds, column of the owning lane); older ones are spilled to the HBM list when the
// ring wraps.  The backward phase compacts towards logical 0, so once a list has shrunk below
// kPrevLds entries it never touches HBM again.  (A third of the round-1 kernel's HBM traffic was
// this list: profiles/r01_notes.md.)
#ifndef BWAMS_PREV_LDS
#define BWAMS_PREV_LDS 8
#endif
#ifndef BWAMS_SEARCH_MIN_BLOCKS
#define BWAMS_SEARCH_MIN_BLOCKS 1
#endif
constexpr int kPrevLds = BWAMS_PREV_LDS;
__device__ __forceinline__ int ring_slot(int i) { return (kPrevLds & (kPrevLds - 1)) == 0 ? (i & (kPrevLds - 1)) : (int)((unsigned)i % (unsigned)kPrevLds); }
struct PrevList {
    uint4 *glob;          // this lane's HBM list
    uint4 *ring;          // this lane's LDS column: ring[slot * kBlock]
};
__device__ __forceinline__ void prev_push(const PrevList &pl, int ph, int num_prev, int64_t k, int64_t l, int64_t s, int n) {
    uint4 *slot = pl.ring + ring_slot(ph) * kBlock;
    if (num_prev >= kPrevLds) pl.glob[ph + kPrevLds] = *slot;      // the entry pushed kPrevLds pushes ago leaves the ring
    *slot = prev_pack(k, l, s, n);
}
__device__ __forceinline__ void prev_get(const PrevList &pl, int base, int p, int64_t &k, int64_t &l, int64_t &s, int &n) {
    const uint4 a = p < kPrevLds ? pl.ring[ring_slot(base + p) * kBlock] : pl.glob[base + p];
    prev_unpack(a, k, l, s, n);
}
__device__ __forceinline__ void prev_put(const PrevList &pl, int base, int p, int64_t k, int64_t l, int64_t s, int n) {
    const uint4 a = prev_pack(k, l, s, n);
    if (p < kPrevLds) pl.ring[ring_slot(base + p) * kBlock] = a;
    else pl.glob[base + p] = a;
}

__device__ __forceinline__ uint4 prev_raw(const PrevList &pl, int base, int p) {
    return p < kPrevLds ? pl.ring[ring_slot(base + p) * kBlock] : pl.glob[base + p];
}

// ---- long interval lists leave the lane ------------------------------------------------------
// A backward phase costs (entries of the list) x (columns until the list dies) dependent extensions, all of them on one
// lane.  Measured on the bench reads (profiles/r03_notes.md 86): the forward phases cost 149 extensions per read whatever
// the read, the backward phases 71 % of the work with a median of 135 extensions, a 99.99th percentile of 1145 and a
// maximum of 4345 (112 entries x 85 columns) — and the launch ends when the lane holding that pivot does.  The entries of
// a column are independent extensions (FMI_search.cpp:1529-1590 decides on them in order, but computes them one by one only
// because it is scalar code), so a pivot whose forward phase leaves `bwd_min_list` entries or more is written out — pivot,
// min_intv, the packed entries — and its backward phase is run by smem_bwd_wave_kernel, one lane per entry and one memory
// round trip per column.  The lane moves on to its next pivot at once.  Returns false (nothing handed over: run the
// backward phase here) when the list is short, too long for the wave kernel's LDS, or the buffers are full.
constexpr int kBwdMaxList = 256;
constexpr int kBwdShortMax = 32;           // lists up to here go to smem_bwd_group_kernel (16 lanes per pivot), longer ones to smem_bwd_wave_kernel
constexpr int kPivotQueue = 64;            // per wavefront, power of two (smem_search_kernel<true>)

// Hand-over slots are reserved per WAVE, in chunks: one atomic per chunk of item slots and one per chunk of entry slots instead
// of two per pivot.  (Round 3's first attempt to hand over every backward phase once the queue was dry put 1 M atomics on two
// words and cost 4 ms, profiles/r03_notes.md 87; a hot word serves ~90 M atomics/s.)  Item slots a wave reserved but did not fill
// are stamped num_prev = 0 when the chunk is closed — the wave kernels skip them; unused entry slots are simply never read.
// While the launch is in its steady state a wave hands over a pivot now and then: small chunks, few holes.  Once it drains,
// every backward phase leaves its lane (see below): large chunks.
struct HoState {
    long long it_base[2];   // first slot of the current chunk per class (0 long lists, 1 short lists), -1 = none   (wave-uniform)
    int it_used[2], it_cap[2];
    long long en_base;      // next free entry slot of the current chunk
    int en_left;
};
__device__ __forceinline__ void ho_init(HoState &h) {
    h.it_base[0] = h.it_base[1] = -1;
    h.it_used[0] = h.it_used[1] = h.it_cap[0] = h.it_cap[1] = 0;
    h.en_base = 0;
    h.en_left = 0;
}
template <int CLS>
__device__ __forceinline__ void ho_close_items(const SeedLaunch &a, HoState &h) {
    const int lane = (int)(threadIdx.x & 63);
    if (h.it_base[CLS] >= 0) {
        const long long slot = h.it_base[CLS] + h.it_used[CLS] + lane;       // chunks hold at most 64 slots
        if (h.it_used[CLS] + lane < h.it_cap[CLS] && slot < a.bwd_items_cap) (CLS ? a.bwd_items_s : a.bwd_items)[slot].num_prev = 0;
    }
    h.it_base[CLS] = -1;
    h.it_used[CLS] = h.it_cap[CLS] = 0;
}
// MUST be called by all 64 lanes.  Returns the slot of a requesting lane (-1: none, or the array is full).
template <int CLS>
__device__ __forceinline__ long long ho_item_slot(const SeedLaunch &a, HoState &h, bool req, bool dry) {
    const unsigned long long m = __ballot(req);
    if (!m) return -1;
    const int lane = (int)(threadIdx.x & 63);
    const int cnt = __popcll(m);
    if (h.it_base[CLS] < 0 || h.it_used[CLS] + cnt > h.it_cap[CLS]) {
        ho_close_items<CLS>(a, h);
        const int chunk = dry ? 64 : (cnt > 8 ? cnt : 8);
        h.it_base[CLS] = (long long)wave_ticket(CLS ? &a.ctr->bwd_items_s : &a.ctr->bwd_items, (unsigned long long)chunk);
        h.it_cap[CLS] = chunk;
    }
    const long long slot = h.it_base[CLS] + h.it_used[CLS] + __popcll(m & ((1ull << lane) - 1ull));
    h.it_used[CLS] += cnt;
    return req && slot < a.bwd_items_cap ? slot : -1;
}

// ---- backward phases leave the lane ------------------------------------------------------------
// A backward phase costs (entries of the list) x (columns until the list dies) dependent extensions, all of them on one
// lane.  Measured on the bench reads (profiles/r03_notes.md 86): the forward phases cost 149 extensions per read whatever
// the read, the backward phases 71 % of the work with a median of 135 extensions, a 99.99th percentile of 1145 and a
// maximum of 4345 (112 entries x 85 columns) — and the launch ends when the lane holding that pivot does.  The entries of
// a column are independent extensions (FMI_search.cpp:1529-1590 decides on them in order, but computes them one by one only
// because it is scalar code), so a pivot can be written out — pivot, min_intv, the packed entries — and its backward phase
// run by a kernel that gives every entry a lane: one memory round trip per column instead of one per entry and column.
// Lanes that want to hand their pivot over arrive here with `req` set; the wave reserves the slots together (HoState).
// Returns true for the lanes whose pivot has left; false (run the backward phase here) when the buffers are full.
// MUST be called by all 64 lanes.
__device__ __forceinline__ bool bwd_hand_over(const SeedLaunch &a, HoState &h, const PrevList &pl, bool req, int base, int num_prev,
                                              uint32_t rid, int x, int min_intv, bool dry, const uint4 *src_list = nullptr) {
    if (!__any(req)) return false;
    const int lane = (int)(threadIdx.x & 63);
    const bool is_short = num_prev <= kBwdShortMax;
    const long long sl = ho_item_slot<0>(a, h, req && !is_short, dry);
    const long long ss = ho_item_slot<1>(a, h, req && is_short, dry);
    const long long slot = is_short ? ss : sl;
    // entry slots: an exclusive prefix sum of the list lengths over the requesting lanes
    const int mine = req ? num_prev : 0;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63);
    if (h.en_left < total) {
        const int chunk = dry ? 2048 : 256;
        const int step = total > chunk ? total : chunk;
        h.en_base = (long long)wave_ticket(&a.ctr->bwd_entries, (unsigned long long)step);
        h.en_left = step;
    }
    const long long eo = h.en_base + (long long)(incl - mine);
    h.en_base += total;
    h.en_left -= total;
    const bool fits = req && slot >= 0 && eo + (long long)mine <= a.bwd_ent_cap;
    if (req && slot >= 0) {
        BwdItem w;
        w.rid = rid; w.x = x; w.min_intv = min_intv; w.num_prev = 0; w.off = 0;
        if (fits) {
            if (src_list) { for (int p = 0; p < num_prev; ++p) a.bwd_ent[eo + p] = src_list[num_prev - 1 - p]; }     // a forward kernel's list: push order
            else for (int p = 0; p < num_prev; ++p) a.bwd_ent[eo + p] = prev_raw(pl, base, p);
            w.num_prev = num_prev;
            w.off = (int64_t)eo;
        }
        (is_short ? a.bwd_items_s : a.bwd_items)[slot] = w;       // num_prev = 0: the slot stays empty
    }
    return fits;
}

// ---- FMA table builders --------------------------------------------------------------------
// One lane per table entry walks its k-mer with plain per-lane block reads (an offline step).
__device__ __forceinline__ void backward_ext_lane(const DevFmi &f, int64_t k, int64_t l, int64_t s, int a,
                                                  int64_t &nk, int64_t &nl, int64_t &ns) {
    const int64_t sp = k, ep = k + s;
    const uint4 *p = f.cp + ((sp >> 6) << 2);
    const uint4 *q = f.cp + ((ep >> 6) << 2);
    Occ4 osp, oep;
    occ_from_block(p[0], p[1], p[2], p[3], sp, osp);
    occ_from_block(q[0], q[1], q[2], q[3], ep, oep);
    const int64_t s0 = oep.v[0] - osp.v[0], s1 = oep.v[1] - osp.v[1];
    const int64_t s2 = oep.v[2] - osp.v[2], s3 = oep.v[3] - osp.v[3];
    const int64_t l3 = l + ((k <= f.sentinel && k + s > f.sentinel) ? 1 : 0);
    const int64_t l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
    nk = (a == 0 ? f.count[0] + osp.v[0] : a == 1 ? f.count[1] + osp.v[1]
          : a == 2 ? f.count[2] + osp.v[2] : f.count[3] + osp.v[3]);
    ns = a == 0 ? s0 : a == 1 ? s1 : a == 2 ? s2 : s3;
    nl = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}

__global__ void build_all_smem_kernel(DevFmi f, int bp, uint32_t *tab) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((int64_t)1 << (2 * bp))) return;
    uint32_t *ent = tab + idx * 32;
    for (int w = 0; w < 32; ++w) ent[w] = 0;
    int a = (int)((idx >> (2 * (bp - 1))) & 3);
    int64_t k = cnt_at(f, a), l = cnt_at(f, 3 - a), s = cnt_at(f, a + 1) - k;
    uint32_t last_avail = 0;
    for (int i = 1; i < bp; ++i) {
        a = (int)((idx >> (2 * (bp - 1 - i))) & 3);
        int64_t nk, nl, ns;
        backward_ext_lane(f, l, k, s, 3 - a, nk, nl, ns);      // forward = backward on the other strand
        const int64_t fk = nl, fl = nk;
        ent[1 + 3 * (i - 1)] = (uint32_t)(fk - k);
        ent[2 + 3 * (i - 1)] = (uint32_t)(fl - cnt_at(f, 3 - a));
        ent[3 + 3 * (i - 1)] = (uint32_t)ns;
        if (ns > 0) last_avail = (uint32_t)i; else break;
        k = fk; l = fl; s = ns;
    }
    ent[0] = last_avail;
}

__global__ void build_last_smem_kernel(DevFmi f, int bp, uint4 *tab) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((int64_t)1 << (2 * bp))) return;
    int a = (int)((idx >> (2 * (bp - 1))) & 3);
    int64_t k = cnt_at(f, a), l = cnt_at(f, 3 - a), s = cnt_at(f, a + 1) - k;
    int i;
    for (i = 1; i < bp; ++i) {
        a = (int)((idx >> (2 * (bp - 1 - i))) & 3);
        int64_t nk, nl, ns;
        backward_ext_lane(f, l, k, s, 3 - a, nk, nl, ns);
        if (ns == 0) break;
        k = nl; l = nk; s = ns;
    }
    uint4 e;
    e.x = (uint32_t)i | (((uint32_t)((uint64_t)k >> 32) & 0xff) << 8) | (((uint32_t)((uint64_t)l >> 32) & 0xff) << 16) |
          (((uint32_t)((uint64_t)s >> 32) & 0xff) << 24);
    e.y = (uint32_t)k; e.z = (uint32_t)l; e.w = (uint32_t)s;
    tab[idx] = e;
}

// ---- reads: packed once per batch, then resident in LDS ---------------------------------
// pack_reads_kernel turns the byte-per-base enc_qdb into `W` 32-bit words per read (W a
// multiple of 4): words [0, cw) hold 2-bit codes (16 bases per word, base j at bits 2*(j&15)),
// words [cw, cw+mw) hold the N mask (1 bit per base).  A lane that takes a read copies its W
// words into its own LDS column (word w of thread t at lds[w * 256 + t]: conflict-free for
// any per-lane w) and from then on every base fetch is an LDS read instead of a
// divergent global byte load (which costs one L2 request per lane and iteration).
struct ReadView {
    const uint32_t *lds_col;   // this thread's LDS column (nullptr -> read packed words from global)
    const uint32_t *gl;        // packed words of the current read in global memory
    int cw;
};
__device__ __forceinline__ uint32_t read_word(const ReadView &r, int w) {
    return r.lds_col ? r.lds_col[w * kBlock] : r.gl[w];
}
__device__ __forceinline__ int base_at(const ReadView &r, int j) {
    const uint32_t code = (read_word(r, j >> 4) >> ((j & 15) * 2)) & 3u;
    const uint32_t isn = (read_word(r, r.cw + (j >> 5)) >> (j & 31)) & 1u;
    return isn ? 4 : (int)code;
}
// copy the packed read `rid` into the LDS column (or just point at it)
__device__ __forceinline__ void read_take(ReadView &r, uint32_t *lds_col_w, const uint32_t *packed, int W,
                                          uint32_t rid) {
    const uint4 *src = reinterpret_cast<const uint4 *>(packed + (int64_t)rid * W);
    r.gl = packed + (int64_t)rid * W;
    if (lds_col_w) {
        for (int w4 = 0; w4 < (W >> 2); ++w4) {
            const uint4 v = src[w4];
            lds_col_w[(4 * w4 + 0) * kBlock] = v.x;
            lds_col_w[(4 * w4 + 1) * kBlock] = v.y;
            lds_col_w[(4 * w4 + 2) * kBlock] = v.z;
            lds_col_w[(4 * w4 + 3) * kBlock] = v.w;
        }
    }
}

__global__ void pack_reads_kernel(const uint8_t *__restrict__ enc, const int64_t *__restrict__ cum, int64_t nseq,
                                  int W, int cw, uint32_t *__restrict__ packed) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nseq * W) return;
    const int64_t rid = g / W;
    const int w = (int)(g - rid * W);
    const int64_t off = cum[rid];
    const int len = (int)(cum[rid + 1] - off);
    uint32_t v = 0;
    if (w < cw) {
        for (int b = 0; b < 16; ++b) {
            const int j = w * 16 + b;
            if (j < len) v |= (uint32_t)(enc[off + j] & 3u) << (2 * b);
        }
    } else {
        const int mwi = w - cw;
        for (int b = 0; b < 32; ++b) {
            const int j = mwi * 32 + b;
            if (j < len && enc[off + j] >= 4) v |= 1u << b;
        }
    }
    packed[g] = v;
}

enum : int { PH_FETCH = 0, PH_LOAD, PH_HOLD, PH_PIVOT, PH_FWD, PH_FWD_END, PH_BWD, PH_BWD_END, PH_EXIT, PH_HO, PH_HO_LATE };

// Rounds 1 and 2.  ALL_POS: work item = read, walk every pivot (round 1).
// !ALL_POS: work item = (read, pivot, min_intv), one pivot (round 2).
template <bool ALL_POS, int TAB = 0>
__global__ __launch_bounds__(kBlock, TAB == 1 ? 3 : BWAMS_SEARCH_MIN_BLOCKS) void smem_search_kernel(SeedLaunch a, const Round2Work *work) {
    const DevFmi &f = a.fmi;
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int cap = a.prev_cap;
    extern __shared__ uint32_t lds_reads[];
    uint32_t *const lds_col = a.reads_in_lds ? lds_reads + threadIdx.x : nullptr;
    PrevList prev;
    prev.glob = a.prev + slot * (int64_t)cap;
    prev.ring = reinterpret_cast<uint4 *>(lds_reads + (a.reads_in_lds ? a.read_w * kBlock : 0)) + threadIdx.x;
    ReadView rv;
    rv.lds_col = lds_col;
    rv.gl = a.packed;
    rv.cw = a.read_cw;
    const int64_t n_work = ALL_POS ? a.nseq : (int64_t)a.ctr->n_work2;
    // round 1: the wave's queue of pivots waiting for a lane (see the push at the end of the loop body)
    uint2 *const pq = reinterpret_cast<uint2 *>(reinterpret_cast<uint4 *>(lds_reads + (a.reads_in_lds ? a.read_w * kBlock : 0)) +
                                                kPrevLds * kBlock) + (threadIdx.x >> 6) * kPivotQueue;
    int pq_head = 0, pq_n = 0;                // wave-uniform
    bool spawned = false;                     // this lane's read goes on in another lane: leave the read after this pivot
    bool dry = false;                         // wave-uniform: the work queue has run out, the launch is draining
    bool ho_tried = false;                    // this pivot's late hand-over has been decided (once per pivot and state of the queue)
    int ho_x = 0;                             // PH_HO / PH_HO_LATE: the column the wave kernel resumes in front of
    HoState ho;
    ho_init(ho);
    const unsigned long long lanes_below = (1ull << (threadIdx.x & 63)) - 1ull;

#ifdef BWAMS_BWDDBG
    const unsigned long long tk_start = wall_clock64();
    unsigned long long tk_dry = 0, n_iter = 0, n_act = 0, n_tail = 0, n_single = 0, n_few = 0;
#endif
    int phase = PH_FETCH;
    uint32_t rid = 0;
    int64_t qoff = 0;
    int len = 0, x = 0, next_x = 0, min_intv = 1;
    int64_t ck = 0, cl = 0, cs = 0;      // current interval (forward phase)
    int cn = 0;                           // its end position n
    int j = 0;                            // position being extended to
    int num_prev = 0, base = 0, p = 0, num_curr = 0, cur_m = 0;
    int32_t curr_s = -1;
    bool first = true;
    int bwd_a = 0;
    unsigned long long n_ext = 0, n_blk = 0;
    WaveOut wo;
    wo.base = -1; wo.used = 0; wo.emitted = 0;
    WaveTickets wt;
    wt.next = 0; wt.left = 0; wt.seen = 0;
    typename CacheOf<TAB>::type bc;
    cache_init(bc);
#ifdef BWAMS_LIST_PREFETCH                   // (measured: no gain, 17.1 against 16.9 ms — the entry's round trip is not what an iteration waits for; profiles/r04_notes.md)
    uint4 pf_ent = make_uint4(0, 0, 0, 0);    // the list entry requested one iteration ahead (logical index pf_p of the current column, -1: none)
    int pf_p = -1;
#endif

    while (true) {
        // at most one SMEM per lane and iteration; written at the wave-uniform point below
        bool em = false;
        uint32_t em_m = 0, em_n = 0;
        int64_t em_k = 0, em_l = 0, em_s = 0;
        // ---- leave a finished pivot -------------------------------------------------
        if (phase == PH_BWD_END) {
            if (num_prev != 0) {
                int64_t qk, ql, qs;
                int qn;
                prev_get(prev, base, 0, qk, ql, qs, qn);
                if (qn - cur_m + 1 >= a.min_seed_len) {
                    em = true; em_m = (uint32_t)cur_m; em_n = (uint32_t)qn; em_k = qk; em_l = ql; em_s = qs;
                }
            }
            x = next_x;
            phase = (ALL_POS && !spawned) ? PH_PIVOT : PH_FETCH;
        }
        wave_emit(a, wo, em, rid, em_m, em_n, em_k, em_l, em_s);
        em = false;
        // ---- a pivot of this wave's queue before a new read ----------------------------
        if (ALL_POS && pq_n) {
            const bool idle = phase == PH_FETCH || phase == PH_EXIT;
            const unsigned long long fm = __ballot(idle);
            if (fm) {
                const int rank = __popcll(fm & lanes_below);
                if (idle && rank < pq_n) {
                    const uint2 it = pq[(pq_head + rank) & (kPivotQueue - 1)];
                    rid = it.x;
                    x = (int)it.y;
                    min_intv = 1;
                    spawned = false;
                    qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    read_take(rv, lds_col, a.packed, a.read_w, rid);
                    phase = PH_PIVOT;
                }
                const int cnt = __popcll(fm);
                const int n = cnt < pq_n ? cnt : pq_n;
                pq_head = (pq_head + n) & (kPivotQueue - 1);
                pq_n -= n;
            }
        }
        // ---- take the next work item ------------------------------------------------
        {
            unsigned long long t = 0;
            if (take_ticket(&a.ctr->work_head, wt, phase == PH_FETCH, t, n_work, ALL_POS)) {
                if ((int64_t)t >= n_work) {
                    phase = PH_EXIT;
#ifdef BWAMS_BWDDBG
                    if (!tk_dry) tk_dry = wall_clock64();
#endif
                } else {
                    if (ALL_POS) {
                        rid = (uint32_t)t;
                        x = 0;
                        min_intv = 1;
                        spawned = false;
                    } else {
                        const Round2Work wk = work[t];
                        rid = wk.rid;
                        x = wk.x;
                        min_intv = wk.min_intv;
                    }
                    qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    phase = PH_PIVOT;
                    if (ALL_POS && a.skip && a.skip[rid]) phase = PH_FETCH;
                    else read_take(rv, lds_col, a.packed, a.read_w, rid);
                }
            }
        }
#ifdef BWAMS_BWDDBG
        if (!tk_dry && __any(phase == PH_EXIT)) tk_dry = wall_clock64();
#endif
        if (__all(phase == PH_EXIT)) break;
        if (!dry && __any(phase == PH_EXIT)) {
            dry = true;
            ho_tried = false;                   // the thresholds have just dropped: running backward phases are asked again
        }

        // ---- open a pivot -----------------------------------------------------------
        if (phase == PH_PIVOT) {
            if (x >= len) {
                phase = PH_FETCH;
            } else {
                const int c = base_at(rv, x);
                if (c >= 4) {
                    x = x + 1;                          // query_pos = next_x = x + 1
                    if (!ALL_POS) phase = PH_FETCH;
                } else {
                    ck = cnt_at(f, c);
                    cl = cnt_at(f, 3 - c);
                    cs = cnt_at(f, c + 1) - ck;
                    cn = x;
                    j = x + 1;
                    next_x = x + 1;
                    num_prev = 0;
                    ho_tried = false;
                    phase = PH_FWD;
                    if (f.all_smem && len - x >= f.all_bp) {
                        // FMA: the first forward steps come from one all_smem entry (FMI_search.cpp:1414-1463)
                        const int bp = f.all_bp;
                        uint32_t tix = 0;
                        int kk = 0;
                        for (; kk < bp; ++kk) {
                            const int bb = base_at(rv, x + kk);
                            if (bb >= 4) break;
                            tix |= (uint32_t)bb << ((bp - 1 - kk) * 2);
                        }
                        const uint32_t *ent = f.all_smem + (int64_t)tix * 32;
                        const int last_avail = (int)ent[0];
                        const int last_idx = (kk > last_avail ? last_avail : kk) - 1;
                        for (int t = 0; t < last_idx; ++t, ++j) {
                            const int bb = base_at(rv, j);
                            next_x = j + 1;
                            const int64_t tk = ck + ent[1 + 3 * t];
                            const int64_t tl = cnt_at(f, 3 - bb) + ent[2 + 3 * t];
                            const int64_t ts = ent[3 + 3 * t];
                            if (ts != cs) {
                                prev_push(prev, cap - 1 - num_prev, num_prev, ck, cl, cs, cn);
                                num_prev++;
                            }
                            if (ts < min_intv) {
                                next_x = j;
                                j = len;                       // no further forward steps
                                break;
                            }
                            ck = tk; cl = tl; cs = ts; cn = j;
                        }
                        if (kk < bp) {                         // an N inside the window (reference quirk kept)
                            next_x = j + 1;
                            j = len;
                        }
                    }
                }
            }
        }

        bool do_ext = false, want_push = false;
        int64_t ek = 0, el = 0, es = 0;
        int ea = 0;
        int64_t pk = 0, pl = 0, ps = 0;
        int pn = 0;

        // ---- forward phase: pre ------------------------------------------------------
        if (phase == PH_FWD) {
            phase = PH_FWD_END;
            if (j < len) {
                const int c = base_at(rv, j);
                next_x = j + 1;
                if (c < 4) {
                    phase = PH_FWD;
                    do_ext = true;
                    ek = cl; el = ck; es = cs;         // forward = backward on the other strand
                    ea = 3 - c;
                }
            }
        }
        if (phase == PH_FWD_END) {
            if (cs >= min_intv) {
                prev_push(prev, cap - 1 - num_prev, num_prev, ck, cl, cs, cn);
                num_prev++;
            }
            base = cap - num_prev;                      // entry p lives at base + p, longest first
            if (a.bwd_min_list > 0 && num_prev >= (dry ? a.bwd_dry_min_list : a.bwd_min_list) && num_prev <= kBwdMaxList) {
                ho_x = x;
                phase = PH_HO;                          // decided at the wave-uniform point below
            } else {
                j = x - 1;
                p = 0; num_curr = 0; curr_s = -1; first = true;
                cur_m = x;
                phase = PH_BWD;
                want_push = ALL_POS && next_x < len;
            }
        }
        // ---- backward phase: pre -----------------------------------------------------
        if (phase == PH_BWD && !do_ext) {
            bool go = true;
            if (p == 0) {
                go = false;
                if (num_prev != 0 && j >= 0) {
                    bwd_a = base_at(rv, j);
                    go = bwd_a < 4;
                }
            }
            if (!go) {
                phase = PH_BWD_END;
            } else {
                // an entry beyond the LDS ring comes from the lane's HBM list: a memory round trip IN FRONT of the block fetch that depends
                // on it — two dependent round trips in one iteration, for the whole wave.  The entry the lane will need NEXT (p + 1: the
                // compaction writes at most index p) is requested here, together with this iteration's blocks, and is in registers when
                // the next iteration asks for it
#ifdef BWAMS_LIST_PREFETCH
                if (pf_p == p) prev_unpack(pf_ent, pk, pl, ps, pn);
                else prev_get(prev, base, p, pk, pl, ps, pn);
                pf_p = -1;
                if (p + 1 < num_prev && p + 1 >= kPrevLds) { pf_ent = prev.glob[base + p + 1]; pf_p = p + 1; }
#else
                prev_get(prev, base, p, pk, pl, ps, pn);
#endif
                do_ext = true;
                ek = pk; el = pl; es = ps; ea = bwd_a;
            }
        }

        // ---- the one extension of this iteration -------------------------------------
        int64_t nk = 0, nl = 0, ns = 0;
#ifdef BWAMS_NO_BLKCACHE
        backward_ext_coop<TAB>(f, do_ext, ek, el, es, ea, nk, nl, ns);
#else
        ext_cached<TAB>(f, bc, do_ext, ek, el, es, ea, nk, nl, ns);
#endif
        if (do_ext) {
            n_ext++;
            n_blk += ((ek >> 6) == ((ek + es) >> 6)) ? 1 : 2;
        }
#ifdef BWAMS_BWDDBG
        { const int na = __popcll(__ballot(do_ext)); n_iter++; n_act += (unsigned long long)na; if (tk_dry) { n_tail++; if (na <= 1) n_single++; else if (na <= 4) n_few++; } }
#endif

        // ---- post ---------------------------------------------------------------------
        if (do_ext && phase == PH_FWD) {
            // the extended interval is (k, l) = (nl, nk) after swapping strands back
            if (ns != cs) {
                prev_push(prev, cap - 1 - num_prev, num_prev, ck, cl, cs, cn);
                num_prev++;
            }
            if (ns < min_intv) {
                next_x = j;
                phase = PH_FWD_END;
                // FWD_END pushes cur only if cs >= min_intv: cur is still the old interval
            } else {
                ck = nl; cl = nk; cs = ns; cn = j;
                j++;
            }
            if (phase == PH_FWD_END) {
                if (cs >= min_intv) {
                    prev_push(prev, cap - 1 - num_prev, num_prev, ck, cl, cs, cn);
                    num_prev++;
                }
                base = cap - num_prev;
                if (a.bwd_min_list > 0 && num_prev >= (dry ? a.bwd_dry_min_list : a.bwd_min_list) && num_prev <= kBwdMaxList) {
                    ho_x = x;
                    phase = PH_HO;
                } else {
                    j = x - 1;
                    p = 0; num_curr = 0; curr_s = -1; first = true;
                    cur_m = x;
                    phase = PH_BWD;
                    want_push = ALL_POS && next_x < len;
                }
            }
        } else if (do_ext && phase == PH_BWD) {
            bool keep = false;
            if (first) {
                if (ns < min_intv && (pn - cur_m + 1) >= a.min_seed_len) {
                    em = true; em_m = (uint32_t)cur_m; em_n = (uint32_t)pn; em_k = pk; em_l = pl; em_s = ps;
                    first = false;
                } else if (ns >= min_intv && ns != (int64_t)curr_s) {
                    keep = true;
                    first = false;
                }
            } else {
                keep = ns >= min_intv && ns != (int64_t)curr_s;
            }
            if (keep) {
                curr_s = (int32_t)ns;
                prev_put(prev, base, num_curr, nk, nl, ns, pn);
                num_curr++;
            }
            p++;
            if (p == num_prev) {                         // this column is done
                num_prev = num_curr;
                if (num_curr == 0) {
                    phase = PH_BWD_END;
                } else {
                    cur_m = j;
                    j--;
                    p = 0; num_curr = 0; curr_s = -1; first = true;
                    // a backward phase that has proven long: the rest of it goes to the wave kernel (which resumes at column cur_m - 1)
                    // ... sooner while the launch drains: the lanes are mostly idle by then, and a backward phase just under the
                    // thresholds (39 entries x 24 columns) is 900 steps on one lane
                    if (!ho_tried && x - cur_m >= (dry ? a.bwd_dry_cols : a.bwd_cols)) {
                        ho_tried = true;
                        if (a.bwd_min_list > 0 && num_prev >= (dry ? a.bwd_dry_late_list : a.bwd_late_list) && num_prev <= kBwdMaxList) {
                            ho_x = cur_m;
                            phase = PH_HO_LATE;
                        }
                    }
                }
            }
        }
        // ---- backward phases that leave the lane (bwd_hand_over) ---------------------------------
        {
            const bool rq = phase == PH_HO || phase == PH_HO_LATE;
            if (__any(rq)) {
                const bool gone = bwd_hand_over(a, ho, prev, rq, base, num_prev, rid, ho_x, min_intv, dry);
                if (rq) {
                    if (gone) {
                        x = next_x;
                        want_push = false;      // a one-entry column ends in the iteration of its forward end: the lane goes on with the read itself
                        phase = (ALL_POS && !spawned) ? PH_PIVOT : PH_FETCH;
                    } else if (phase == PH_HO) {            // the buffers are full: the backward phase runs here
                        j = x - 1;
                        p = 0; num_curr = 0; curr_s = -1; first = true;
                        cur_m = x;
                        phase = PH_BWD;
                        want_push = ALL_POS && next_x < len;
                    } else {
                        phase = PH_BWD;                     // ... goes on here: its state is untouched
                    }
                }
            }
        }
        // ---- round 1: the read's next pivot does not wait for this backward phase --------
        // The pivots of a read are found one after the other (the next one starts where this forward phase ended), but
        // only the FORWARD phases depend on each other: once next_x is known the backward phase of this pivot and the
        // whole next pivot are independent work.  The lane keeps the backward phase and leaves (rid, next_x) in its
        // wave's queue (LDS, kPivotQueue entries); any lane of the wave that runs out of work takes it before a new
        // read.  A read with seven pivots — a third of the bench reads, 70 % of the work — then occupies several lanes
        // for the length of its forward phases plus one backward phase instead of one lane for the sum of all of them,
        // which is what the launch's tail was made of.  A full queue just means the lane goes on with the read itself.
        if (ALL_POS) {
            const unsigned long long pm = __ballot(want_push);
            if (pm) {
                const int rank = __popcll(pm & lanes_below);
                const int space = kPivotQueue - pq_n;
                if (want_push && rank < space) {
                    pq[(pq_head + pq_n + rank) & (kPivotQueue - 1)] = make_uint2(rid, (uint32_t)next_x);
                    spawned = true;
                }
                const int cnt = __popcll(pm);
                pq_n += cnt < space ? cnt : space;
            }
        }
        wave_emit(a, wo, em, rid, em_m, em_n, em_k, em_l, em_s);
    }
#ifdef BWAMS_BWDDBG
    if (ALL_POS && (threadIdx.x & 63) == 0) {
        const unsigned long long tk_end = wall_clock64();
        unsigned long long dry = 0;
        for (int l = 0; l < 64; ++l) { const unsigned long long v = __shfl(tk_dry, l); if (v && (!dry || v < dry)) dry = v; }
        atomicMax(&a.ctr->dbg[8], ~tk_start);                // earliest start
        if (dry) atomicMax(&a.ctr->dbg[9], ~dry);            // first time the read queue was found empty
        atomicMax(&a.ctr->dbg[10], tk_end);                  // last wave out
        atomicAdd(&a.ctr->dbg[11], tk_end - tk_start);       // wave-time
        atomicAdd(&a.ctr->dbg[12], 1ull);
        atomicAdd(&a.ctr->dbg[13], n_iter); atomicAdd(&a.ctr->dbg[14], n_act);
        if (dry) atomicAdd(&a.ctr->dbg[15], tk_end - dry);   // wave-time after the queue ran dry
        {
            const unsigned long long t0 = ~a.ctr->dbg[8];
            int bin = (int)((tk_end - t0) / 40000ull);            // 0.4 ms bins
            if (bin > 47) bin = 47;
            atomicAdd(&a.ctr->dbg[16 + bin], 1ull);
            atomicAdd(&a.ctr->dbg[64], n_tail); atomicAdd(&a.ctr->dbg[65], n_single); atomicAdd(&a.ctr->dbg[66], n_few);
            atomicMax(&a.ctr->dbg[67], n_single);
        }
    }
#endif
    ho_close_items<0>(a, ho);
    ho_close_items<1>(a, ho);
    wave_emit_finish(a, wo);
    flush_counters(a.ctr, n_ext, n_blk);
}

// ---- rounds 1 and 2 as TWO lane kernels (round 4) ------------------------------------------------------------------------------
// smem_search_kernel runs forward and backward phases of different lanes in one loop: every iteration executes the forward lanes'
// code AND the backward lanes' code (the wave diverges), about 700 vector + 290 scalar instructions for one extension per lane — the
// launch is bound by instruction issue (profiles/r04_notes.md), not by the memory round trip.  Split by role, each loop carries one
// role's code and state:
//   smem_fwd_kernel   a lane owns a read (round 1) or a pivot (round 2) and runs FORWARD phases only; every interval the reference
//                     would push on prevArray goes straight to the pivot's list in HBM (no ring, no copy); at the end of a forward
//                     phase the pivot becomes an item (rid, x, min_intv, entries, list) and the lane opens the read's next pivot at once.
//                     Forward phases cost the same for every read (149 extensions): no tail.
//   smem_bwdl_kernel  a lane owns an item and runs its BACKWARD phase: the first column reads the list where the forward kernel left
//                     it, the survivors are compacted into the lane's LDS ring / HBM list as before.  Long phases leave the lane for
//                     the kernels behind (bwd_hand_over), as in the one-kernel form.
// Lists in HBM: round 1, pivot x of read r at 2 (cum[r] + r) + 2 x (a read's forward phases tile it: list i ends before list i + 1
// begins); round 2, work item t at t (max_len + 2).  Same extensions, same SMEMs, same counts as smem_search_kernel.
enum : int { FW_FETCH = 0, FW_PIVOT, FW_FWD, FW_EXIT };

template <bool ALL_POS, int TAB>
__global__ __launch_bounds__(kBlock) void smem_fwd_kernel(SeedLaunch a, const Round2Work *work) {
    const DevFmi &f = a.fmi;
    extern __shared__ uint32_t lds_reads[];
    uint32_t *const lds_col = a.reads_in_lds ? lds_reads + threadIdx.x : nullptr;
    ReadView rv;
    rv.lds_col = lds_col;
    rv.gl = a.packed;
    rv.cw = a.read_cw;
    const int64_t n_work = ALL_POS ? a.nseq : (int64_t)a.ctr->n_work2;
    const unsigned long long lanes_below = (1ull << (threadIdx.x & 63)) - 1ull;
    int phase = FW_FETCH;
    uint32_t rid = 0;
    int len = 0, x = 0, next_x = 0, min_intv = 1;
    int64_t ck = 0, cl = 0, cs = 0;
    int cn = 0, j = 0, num_prev = 0;
    int64_t lbase = 0;                    // round 1: 2 (cum[rid] + rid); round 2: the item's list
    int64_t fl = 0;                       // the list of the pivot in progress
    bool lst_ok = true;                   // its list lies inside the buffer (round 2 beyond the buffer: counted, the caller re-runs unsplit)
    unsigned long long n_ext = 0, n_blk = 0;
    WaveTickets wt;
    wt.next = 0; wt.left = 0; wt.seen = 0;
    long long it_base = -1;               // the wave's chunk of item slots
    int it_used = 0;

    auto push = [&](int64_t k, int64_t l, int64_t s_, int n) {
        if (lst_ok) a.fl_ent[fl + num_prev] = prev_pack(k, l, s_, n);
        num_prev++;
    };
    while (true) {
        {
            unsigned long long t = 0;
            if (take_ticket(&a.ctr->work_head, wt, phase == FW_FETCH, t, n_work, ALL_POS)) {
                if ((int64_t)t >= n_work) phase = FW_EXIT;
                else {
                    if (ALL_POS) {
                        rid = (uint32_t)t; x = 0; min_intv = 1;
                    } else {
                        const Round2Work wk = work[t];
                        rid = wk.rid; x = wk.x; min_intv = wk.min_intv;
                    }
                    const int64_t qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    lbase = ALL_POS ? 2 * (qoff + (int64_t)rid) : (int64_t)t * (int64_t)(a.fl_item_stride);
                    lst_ok = ALL_POS || lbase + a.fl_item_stride <= a.fl_cap;
                    phase = FW_PIVOT;
                    if (ALL_POS && a.skip && a.skip[rid]) phase = FW_FETCH;
                    else if (!lst_ok) { atomicAdd(&a.ctr->f_overflow, 1ull); phase = FW_FETCH; }
                    else read_take(rv, lds_col, a.packed, a.read_w, rid);
                }
            }
        }
        if (__all(phase == FW_EXIT)) break;

        if (phase == FW_PIVOT) {
            if (x >= len) phase = FW_FETCH;
            else {
                const int c = base_at(rv, x);
                if (c >= 4) {
                    x = x + 1;
                    if (!ALL_POS) phase = FW_FETCH;
                } else {
                    ck = cnt_at(f, c);
                    cl = cnt_at(f, 3 - c);
                    cs = cnt_at(f, c + 1) - ck;
                    cn = x;
                    j = x + 1;
                    next_x = x + 1;
                    num_prev = 0;
                    fl = ALL_POS ? lbase + 2 * (int64_t)x : lbase;
                    phase = FW_FWD;
                    if (f.all_smem && len - x >= f.all_bp) {
                        // FMA: the first forward steps come from one all_smem entry (FMI_search.cpp:1414-1463)
                        const int bp = f.all_bp;
                        uint32_t tix = 0;
                        int kk = 0;
                        for (; kk < bp; ++kk) {
                            const int bb = base_at(rv, x + kk);
                            if (bb >= 4) break;
                            tix |= (uint32_t)bb << ((bp - 1 - kk) * 2);
                        }
                        const uint32_t *ent = f.all_smem + (int64_t)tix * 32;
                        const int last_avail = (int)ent[0];
                        const int last_idx = (kk > last_avail ? last_avail : kk) - 1;
                        for (int t = 0; t < last_idx; ++t, ++j) {
                            const int bb = base_at(rv, j);
                            next_x = j + 1;
                            const int64_t tk = ck + ent[1 + 3 * t];
                            const int64_t tl = cnt_at(f, 3 - bb) + ent[2 + 3 * t];
                            const int64_t ts = ent[3 + 3 * t];
                            if (ts != cs) push(ck, cl, cs, cn);
                            if (ts < min_intv) {
                                next_x = j;
                                j = len;                       // no further forward steps
                                break;
                            }
                            ck = tk; cl = tl; cs = ts; cn = j;
                        }
                        if (kk < bp) {                         // an N inside the window (reference quirk kept)
                            next_x = j + 1;
                            j = len;
                        }
                    }
                }
            }
        }
        // ---- one forward extension ----------------------------------------------------------------
        bool do_ext = false, fin = false;
        int ea = 0;
        if (phase == FW_FWD) {
            fin = true;
            if (j < len) {
                const int c = base_at(rv, j);
                next_x = j + 1;
                if (c < 4) { fin = false; do_ext = true; ea = 3 - c; }
            }
        }
        int64_t nk = 0, nl = 0, ns = 0;
        backward_ext_coop<TAB>(f, do_ext, cl, ck, cs, ea, nk, nl, ns);          // forward = backward on the other strand
        if (do_ext) {
            n_ext++;
            n_blk += ((cl >> 6) == ((cl + cs) >> 6)) ? 1 : 2;
            if (ns != cs) push(ck, cl, cs, cn);
            if (ns < min_intv) {
                next_x = j;
                fin = true;                                   // cur is still the old interval
            } else {
                ck = nl; cl = nk; cs = ns; cn = j;
                j++;
            }
        }
        bool item = false;
        if (phase == FW_FWD && fin) {
            if (cs >= min_intv) push(ck, cl, cs, cn);
            item = num_prev > 0;                              // an empty list leaves nothing for the backward phase to do
        }
        // ---- the pivots whose forward phase ended become items (slots reserved per wave, 64 at a time) --------
        {
            const unsigned long long m = __ballot(item);
            if (m) {
                const int cnt = __popcll(m);
                if (it_base < 0 || it_used + cnt > 64) {
                    const int lane = (int)(threadIdx.x & 63);
                    if (it_base >= 0) { const long long sl = it_base + it_used + lane; if (it_used + lane < 64 && sl < a.f_items_cap) a.f_items[sl].num_prev = 0; }
                    it_base = (long long)wave_ticket(&a.ctr->f_items, 64ull);
                    it_used = 0;
                }
                const long long sl = it_base + it_used + __popcll(m & lanes_below);
                if (item) {
                    if (sl < a.f_items_cap) {
                        BwdItem w;
                        w.rid = rid; w.x = x; w.min_intv = min_intv; w.num_prev = num_prev; w.off = fl;
                        a.f_items[sl] = w;
                    } else atomicAdd(&a.ctr->f_overflow, 1ull);
                }
                it_used += cnt;
            }
        }
        if (phase == FW_FWD && fin) {
            x = next_x;
            phase = ALL_POS ? FW_PIVOT : FW_FETCH;
        }
    }
    {   // the unused slots of the wave's last chunk
        const int lane = (int)(threadIdx.x & 63);
        if (it_base >= 0) { const long long sl = it_base + it_used + lane; if (it_used + lane < 64 && sl < a.f_items_cap) a.f_items[sl].num_prev = 0; }
    }
    flush_counters(a.ctr, n_ext, n_blk);
}

enum : int { BL_FETCH = 0, BL_BWD, BL_BWD_END, BL_HO, BL_HO_LATE, BL_EXIT };

template <int TAB>
__global__ __launch_bounds__(kBlock, TAB == 1 ? 3 : BWAMS_SEARCH_MIN_BLOCKS) void smem_bwdl_kernel(SeedLaunch a) {
    const DevFmi &f = a.fmi;
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    extern __shared__ uint32_t lds_reads[];
    uint32_t *const lds_col = a.reads_in_lds ? lds_reads + threadIdx.x : nullptr;
    PrevList prev;
    prev.glob = a.prev + slot * (int64_t)a.prev_cap;
    prev.ring = reinterpret_cast<uint4 *>(lds_reads + (a.reads_in_lds ? a.read_w * kBlock : 0)) + threadIdx.x;
    ReadView rv;
    rv.lds_col = lds_col;
    rv.gl = a.packed;
    rv.cw = a.read_cw;
    unsigned long long n_items = a.f_items_fixed >= 0 ? (unsigned long long)a.f_items_fixed : a.ctr->f_items;
    if ((int64_t)n_items > a.f_items_cap) n_items = (unsigned long long)a.f_items_cap;
    int phase = BL_FETCH;
    uint32_t rid = 0;
    int x = 0, min_intv = 1, j = 0, num_prev = 0, p = 0, num_curr = 0, cur_m = 0, bwd_a = 0, ho_x = 0;
    int32_t curr_s = -1;
    bool first = true, dry = false, ho_tried = false;
    const uint4 *src = nullptr;           // the forward kernel's list while the first column reads it (push order: shortest match first)
    unsigned long long n_ext = 0, n_blk = 0;
    WaveOut wo;
    wo.base = -1; wo.used = 0; wo.emitted = 0;
    WaveTickets wt;
    wt.next = 0; wt.left = 0; wt.seen = 0;
    typename CacheOf<TAB>::type bc;
    cache_init(bc);
    HoState ho;
    ho_init(ho);

    auto entry = [&](int q, int64_t &k, int64_t &l, int64_t &s_, int &n) {
        if (src) prev_unpack(src[num_prev - 1 - q], k, l, s_, n);
        else prev_get(prev, 0, q, k, l, s_, n);
    };
    while (true) {
        bool em = false;
        uint32_t em_m = 0, em_n = 0;
        int64_t em_k = 0, em_l = 0, em_s = 0;
        if (phase == BL_BWD_END) {
            if (num_prev != 0) {
                int64_t qk, ql, qs;
                int qn;
                entry(0, qk, ql, qs, qn);
                if (qn - cur_m + 1 >= a.min_seed_len) { em = true; em_m = (uint32_t)cur_m; em_n = (uint32_t)qn; em_k = qk; em_l = ql; em_s = qs; }
            }
            phase = BL_FETCH;
        }
        wave_emit(a, wo, em, rid, em_m, em_n, em_k, em_l, em_s);
        em = false;
        {
            unsigned long long t = 0;
            if (take_ticket(&a.ctr->f_ticket, wt, phase == BL_FETCH, t, (int64_t)n_items, true)) {
                if (t >= n_items) phase = BL_EXIT;
                else {
                    const BwdItem it = a.f_items[t];
                    if (it.num_prev > 0) {
                        rid = it.rid; x = it.x; min_intv = it.min_intv; num_prev = it.num_prev;
                        src = a.fl_ent + it.off;
                        read_take(rv, lds_col, a.packed, a.read_w, rid);
                        j = x - 1; p = 0; num_curr = 0; curr_s = -1; first = true; cur_m = x; ho_tried = false;
                        phase = BL_BWD;
                        if (a.bwd_min_list > 0 && num_prev >= (dry ? a.bwd_dry_min_list : a.bwd_min_list) && num_prev <= kBwdMaxList) { ho_x = x; phase = BL_HO; }
                    }
                }
            }
        }
        if (__all(phase == BL_EXIT)) break;
        if (!dry && __any(phase == BL_EXIT)) { dry = true; ho_tried = false; }

        bool do_ext = false;
        int64_t pk = 0, pl = 0, ps = 0;
        int pn = 0;
        if (phase == BL_BWD) {
            bool go = true;
            if (p == 0) {
                go = false;
                if (num_prev != 0 && j >= 0) { bwd_a = base_at(rv, j); go = bwd_a < 4; }
            }
            if (!go) phase = BL_BWD_END;
            else { entry(p, pk, pl, ps, pn); do_ext = true; }
        }
        int64_t nk = 0, nl = 0, ns = 0;
#ifdef BWAMS_NO_BLKCACHE
        backward_ext_coop<TAB>(f, do_ext, pk, pl, ps, bwd_a, nk, nl, ns);
#else
        ext_cached<TAB>(f, bc, do_ext, pk, pl, ps, bwd_a, nk, nl, ns);
#endif
        if (do_ext) {
            n_ext++;
            n_blk += ((pk >> 6) == ((pk + ps) >> 6)) ? 1 : 2;
            bool keep = false;
            if (first) {
                if (ns < min_intv && (pn - cur_m + 1) >= a.min_seed_len) {
                    em = true; em_m = (uint32_t)cur_m; em_n = (uint32_t)pn; em_k = pk; em_l = pl; em_s = ps;
                    first = false;
                } else if (ns >= min_intv && ns != (int64_t)curr_s) { keep = true; first = false; }
            } else keep = ns >= min_intv && ns != (int64_t)curr_s;
            if (keep) {
                curr_s = (int32_t)ns;
                prev_put(prev, 0, num_curr, nk, nl, ns, pn);
                num_curr++;
            }
            p++;
            if (p == num_prev) {                         // this column is done: the list now lives in the lane's ring / HBM list
                src = nullptr;
                num_prev = num_curr;
                if (num_curr == 0) phase = BL_BWD_END;
                else {
                    cur_m = j;
                    j--;
                    p = 0; num_curr = 0; curr_s = -1; first = true;
                    if (!ho_tried && x - cur_m >= (dry ? a.bwd_dry_cols : a.bwd_cols)) {
                        ho_tried = true;
                        if (a.bwd_min_list > 0 && num_prev >= (dry ? a.bwd_dry_late_list : a.bwd_late_list) && num_prev <= kBwdMaxList) { ho_x = cur_m; phase = BL_HO_LATE; }
                    }
                }
            }
        }
        {
            const bool rq = phase == BL_HO || phase == BL_HO_LATE;
            if (__any(rq)) {
                const bool gone = bwd_hand_over(a, ho, prev, rq, 0, num_prev, rid, ho_x, min_intv, dry, phase == BL_HO ? src : nullptr);
                if (rq) phase = gone ? BL_FETCH : BL_BWD;       // the buffers are full: the backward phase runs (on) here, its state untouched
            }
        }
        wave_emit(a, wo, em, rid, em_m, em_n, em_k, em_l, em_s);
    }
    ho_close_items<0>(a, ho);
    ho_close_items<1>(a, ho);
    wave_emit_finish(a, wo);
    flush_counters(a.ctr, n_ext, n_blk);
}

// The backward phase of one pivot per wavefront (see bwd_hand_over): the list lives in LDS, lane p of a batch of 64 owns entry
// p, every column is one cooperative extension of all entries followed by the reference's in-order decisions
// (FMI_search.cpp:1529-1590) taken with ballots:
//   * while nothing has been kept or emitted in this column (`first`), the first entry that either dies long enough
//     (ns < min_intv, length >= min_seed_len: it is an SMEM) or survives (ns >= min_intv) settles it;
//   * a surviving entry is kept unless its size equals the 32-bit `curr_s` of the last entry kept — which, entries being
//     dropped only when they equal it, is the truncated size of the closest surviving entry before it.
// Kept entries are compacted to the front of the list (writes land below the batch being read).  Extensions and blocks are
// counted as the lane-per-read kernel counts them.
constexpr int kBwdItemsPerTicket = 4;
// LDS of the launch behind a search kernel, per WAVEFRONT (its waves move from the first role to the second at different times,
// so a wave's two roles share its own region and nothing else): kBwdMaxList 16-byte entries — the wave role's list, or the four
// groups' lists of kBwdShortMax entries each — followed by four copies of a packed read (the wave role uses the first).  Reads
// beyond kBwdReadLds words stay in global memory.
constexpr int kBwdReadLds = 256;
__device__ __forceinline__ int bwd_read_words(const SeedLaunch &a) { return a.read_w <= kBwdReadLds ? a.read_w : 0; }
__device__ __forceinline__ int bwd_wave_words(const SeedLaunch &a) { return kBwdMaxList * 4 + 4 * bwd_read_words(a); }

template <int TAB>
__device__ __forceinline__ void bwd_wave_role(const SeedLaunch &a, uint32_t *lds_reads, WaveOut &wo, unsigned long long &n_ext_io,
                                              unsigned long long &n_blk_io) {
    const DevFmi &f = a.fmi;
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    uint32_t *const wreg = lds_reads + wv * bwd_wave_words(a);
    uint4 *const lst = reinterpret_cast<uint4 *>(wreg);
    uint32_t *const rd = wreg + kBwdMaxList * 4;
    const bool rd_lds = bwd_read_words(a) != 0;
    ReadView rv;
    rv.lds_col = nullptr;
    rv.gl = rd;
    rv.cw = a.read_cw;
    unsigned long long n_items = a.ctr->bwd_items;
    if ((int64_t)n_items > a.bwd_items_cap) n_items = (unsigned long long)a.bwd_items_cap;
    unsigned long long n_ext = 0, n_blk = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#ifdef BWAMS_BWDDBG
    const unsigned long long tk_start = wall_clock64();
    unsigned long long d_items = 0, d_cols = 0, d_setup = 0, tk_last = tk_start;
#endif

    for (;;) {
        const unsigned long long t0 = wave_ticket(&a.ctr->bwd_ticket, (unsigned long long)kBwdItemsPerTicket);
        if (t0 >= n_items) break;
        for (unsigned long long t = t0; t < t0 + kBwdItemsPerTicket && t < n_items; ++t) {
            const BwdItem it = a.bwd_items[t];
            int num_prev = it.num_prev;
            if (num_prev == 0) continue;
            const uint32_t rid = it.rid;
            const int min_intv = it.min_intv;
            for (int p = lane; p < num_prev; p += 64) lst[p] = a.bwd_ent[it.off + p];
            if (rd_lds) { for (int w = lane; w < a.read_w; w += 64) rd[w] = a.packed[(int64_t)rid * a.read_w + w]; }
            else rv.gl = a.packed + (int64_t)rid * a.read_w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int j = it.x - 1, cur_m = it.x;      // a pivot's first column, or the column a lane stopped in front of: x is the last position matched
#ifdef BWAMS_BWDDBG
            d_items++;
            const unsigned long long tk_a = wall_clock64();
#endif
            for (;;) {
                int ba = 4;
                if (num_prev != 0 && j >= 0) ba = base_at(rv, j);
                if (ba >= 4) {                              // the pivot is left (PH_BWD_END of the lane kernel)
                    bool em = false;
                    int64_t qk = 0, ql = 0, qs = 0;
                    int qn = 0;
                    if (num_prev != 0) {
                        prev_unpack(lst[0], qk, ql, qs, qn);
                        em = lane == 0 && qn - cur_m + 1 >= a.min_seed_len;
                    }
                    wave_emit(a, wo, em, rid, (uint32_t)cur_m, (uint32_t)qn, qk, ql, qs);
                    break;
                }
                bool first = true;
                int32_t curr_s = -1;                        // (int32_t) size of the closest surviving entry so far
                int num_curr = 0;
                for (int b = 0; b < num_prev; b += 64) {
#ifdef BWAMS_BWDDBG
                    d_cols++;
#endif
                    const int p = b + lane;
                    const bool need = p < num_prev;
                    int64_t pk = 0, pl = 0, ps = 0, nk = 0, nl = 0, ns = 0;
                    int pn = 0;
                    if (need) prev_unpack(lst[p], pk, pl, ps, pn);
                    backward_ext_coop<TAB>(f, need, pk, pl, ps, ba, nk, nl, ns);
                    if (need) {
                        n_ext++;
                        n_blk += ((pk >> 6) == ((pk + ps) >> 6)) ? 1 : 2;
                    }
                    const bool alive = need && ns >= min_intv;
                    const bool dies_long = need && ns < min_intv && (pn - cur_m + 1) >= a.min_seed_len;
                    const unsigned long long m_alive = __ballot(alive);
                    bool em = false;
                    if (first) {
                        const unsigned long long m_dies = __ballot(dies_long);
                        if (m_alive | m_dies) {
                            const int fa = m_alive ? __ffsll((long long)m_alive) - 1 : 64;
                            const int fd = m_dies ? __ffsll((long long)m_dies) - 1 : 64;
                            em = fd < fa && lane == fd;
                            first = false;
                        }
                    }
                    wave_emit(a, wo, em, rid, (uint32_t)cur_m, (uint32_t)pn, pk, pl, ps);
                    // size of the closest surviving entry before this one (this batch, else the carry)
                    const unsigned long long lower = m_alive & below;
                    const int src = lower ? 63 - __clzll((long long)lower) : lane;
                    const int32_t s_here = (int32_t)ns;
                    const int32_t s_src = __shfl(s_here, src);          // by every lane: a source lane inside a branch not taken reads as 0
                    const int32_t s_before = lower ? s_src : curr_s;
                    const bool keep = alive && ns != (int64_t)s_before;
                    const unsigned long long m_keep = __ballot(keep);
                    __builtin_amdgcn_wave_barrier();        // every lane has read its entry before the compaction writes
                    if (keep) lst[num_curr + __popcll(m_keep & below)] = prev_pack(nk, nl, ns, pn);
                    num_curr += __popcll(m_keep);
                    if (m_alive) curr_s = __shfl(s_here, 63 - __clzll((long long)m_alive));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                num_prev = num_curr;
                if (num_curr == 0) break;                   // nothing survived: PH_BWD_END with an empty list emits nothing
                cur_m = j;
                j--;
            }
#ifdef BWAMS_BWDDBG
            { const unsigned long long tk_b = wall_clock64(); d_setup += tk_a - tk_last; tk_last = tk_b; }
#endif
        }
    }
#ifdef BWAMS_BWDDBG
    if (lane == 0 && d_items) {
        const unsigned long long busy = tk_last - tk_start;
        atomicAdd(&a.ctr->dbg[0], d_items); atomicAdd(&a.ctr->dbg[1], d_cols); atomicAdd(&a.ctr->dbg[2], busy);
        atomicAdd(&a.ctr->dbg[3], d_setup); atomicMax(&a.ctr->dbg[4], busy); atomicAdd(&a.ctr->dbg[5], 1ull);
        atomicMax(&a.ctr->dbg[6], tk_last); atomicMax(&a.ctr->dbg[7], ~tk_start);
    }
#endif
    n_ext_io += n_ext;
    n_blk_io += n_blk;
}

// The same procedure with SIXTEEN lanes per pivot: four pivots share a wavefront.  A typical list has 16-23 entries at the forward
// end and shrinks from there, so a whole wavefront per pivot (above) leaves three lanes in four idle and is bound by vector issue
// (2.7 us per column whatever the list holds); here a column of a list of up to kBwdShortMax entries is one or two batches of
// sixteen, the four groups of a wavefront advance independently (every loop iteration is one batch of every live group), and a
// group that finishes its pivot takes the next one of the wave's reservation while the others carry on.  The in-order decisions
// are the ones above, taken on the group's sixteen bits of each ballot.
constexpr int kGrp = 16;
constexpr int kGroupItemsPerTicket = 16;
template <int TAB>
__device__ __forceinline__ void bwd_group_role(const SeedLaunch &a, uint32_t *lds_reads, WaveOut &wo, unsigned long long &n_ext_io,
                                               unsigned long long &n_blk_io) {
    const DevFmi &f = a.fmi;
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    const int grp = lane >> 4, gl = lane & 15;
    uint32_t *const wreg = lds_reads + wv * bwd_wave_words(a);
    uint4 *const lst = reinterpret_cast<uint4 *>(wreg) + grp * kBwdShortMax;
    uint32_t *const rd = wreg + kBwdMaxList * 4 + grp * bwd_read_words(a);
    const bool rd_lds = bwd_read_words(a) != 0;
    ReadView rv;
    rv.lds_col = nullptr;
    rv.gl = rd;
    rv.cw = a.read_cw;
    unsigned long long n_items = a.ctr->bwd_items_s;
    if ((int64_t)n_items > a.bwd_items_cap) n_items = (unsigned long long)a.bwd_items_cap;
    unsigned long long n_ext = 0, n_blk = 0;
    const uint32_t gbelow = (1u << gl) - 1u;
    const unsigned long long leaders = 0x0001000100010001ull;
    // the wave's reservation of item indices (wave-uniform)
    unsigned long long tk_next = 0;
    int tk_left = 0;
    bool exhausted = false;
    // the group's pivot (uniform over the group's lanes)
    bool live = false;
    uint32_t rid = 0;
    int min_intv = 1, num_prev = 0, j = 0, cur_m = 0, b = 0, num_curr = 0;
    int32_t curr_s = -1;
    bool first = true;
#ifdef BWAMS_BWDDBG
    const unsigned long long tk_start = wall_clock64();
    unsigned long long d_items = 0, d_iter = 0, d_live = 0;
#endif

    for (;;) {
        // ---- groups without a pivot take the next item -------------------------------------------
        const unsigned long long want = __ballot(!live) & leaders;
        if (want && !exhausted) {
            if (tk_left == 0) {
                tk_next = wave_ticket(&a.ctr->bwd_ticket_s, (unsigned long long)kGroupItemsPerTicket);
                tk_left = kGroupItemsPerTicket;
                if (tk_next >= n_items) { exhausted = true; tk_left = 0; }
            }
            if (!exhausted) {
                const int rank = __popcll(want & ((1ull << (lane & 48)) - 1ull));     // groups in front of this one that want an item
                const int cnt = __popcll(want);
                const int served = cnt < tk_left ? cnt : tk_left;
                const unsigned long long t = tk_next + (unsigned long long)rank;
                if (!live && rank < served && t < n_items) {
                    const BwdItem it = a.bwd_items_s[t];
                    if (it.num_prev > 0) {
                        rid = it.rid;
                        min_intv = it.min_intv;
                        num_prev = it.num_prev;
                        for (int p = gl; p < num_prev; p += kGrp) lst[p] = a.bwd_ent[it.off + p];
                        if (rd_lds) { for (int w = gl; w < a.read_w; w += kGrp) rd[w] = a.packed[(int64_t)rid * a.read_w + w]; }
                        else rv.gl = a.packed + (int64_t)rid * a.read_w;
                        j = it.x - 1; cur_m = it.x;      // x is the last position matched (a pivot, or the column a lane stopped in front of)
                        b = 0; num_curr = 0; curr_s = -1; first = true;
                        live = true;
#ifdef BWAMS_BWDDBG
                        if (gl == 0) d_items++;
#endif
                    }
                }
                tk_next += (unsigned long long)served;
                tk_left -= served;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!__any(live)) {
            if (exhausted) break;
            continue;
        }
#ifdef BWAMS_BWDDBG
        d_iter++; d_live += (unsigned long long)__popcll(__ballot(live) & leaders);
#endif
        // ---- a column opens: the base to extend by, or the end of the pivot ------------------------
        bool em = false;
        uint32_t em_n = 0;
        int64_t em_k = 0, em_l = 0, em_s = 0;
        int ba = 4;
        if (live) {
            if (j >= 0) ba = base_at(rv, j);
            if (ba >= 4) {                                  // the pivot is left (PH_BWD_END of the lane kernel); lists here are never empty
                int64_t qk, ql, qs;
                int qn;
                prev_unpack(lst[0], qk, ql, qs, qn);
                em = gl == 0 && qn - cur_m + 1 >= a.min_seed_len;
                em_n = (uint32_t)qn; em_k = qk; em_l = ql; em_s = qs;
                live = false;
            }
        }
        const int em_m = cur_m;
        // ---- one batch of sixteen entries per live group ------------------------------------------------
        const int p = b + gl;
        const bool need = live && p < num_prev;
        int64_t pk = 0, pl = 0, ps = 0, nk = 0, nl = 0, ns = 0;
        int pn = 0;
        if (need) prev_unpack(lst[p], pk, pl, ps, pn);
        backward_ext_coop<TAB>(f, need, pk, pl, ps, ba & 3, nk, nl, ns);
        if (need) {
            n_ext++;
            n_blk += ((pk >> 6) == ((pk + ps) >> 6)) ? 1 : 2;
        }
        const bool alive = need && ns >= min_intv;
        const bool dies_long = need && ns < min_intv && (pn - cur_m + 1) >= a.min_seed_len;
        const uint32_t m_alive = (uint32_t)(__ballot(alive) >> (lane & 48)) & 0xffffu;
        const uint32_t m_dies = (uint32_t)(__ballot(dies_long) >> (lane & 48)) & 0xffffu;
        if (live && first && (m_alive | m_dies)) {
            const int fa = m_alive ? __ffs((int)m_alive) - 1 : 64;
            const int fd = m_dies ? __ffs((int)m_dies) - 1 : 64;
            if (fd < fa && gl == fd) {
                em = true;
                em_n = (uint32_t)pn; em_k = pk; em_l = pl; em_s = ps;
            }
            first = false;
        }
        wave_emit(a, wo, em, rid, (uint32_t)em_m, em_n, em_k, em_l, em_s);
        // size of the closest surviving entry before this one (this batch, else the carry)
        const uint32_t lower = m_alive & gbelow;
        const int src = (lane & 48) + (lower ? 31 - __clz((int)lower) : gl);
        const int32_t s_here = (int32_t)ns;
        const int32_t s_src = __shfl(s_here, src);           // by every lane: a source lane inside a branch not taken reads as 0
        const int32_t s_before = lower ? s_src : curr_s;
        const bool keep = alive && ns != (int64_t)s_before;
        const uint32_t m_keep = (uint32_t)(__ballot(keep) >> (lane & 48)) & 0xffffu;
        const int32_t s_last = __shfl(s_here, (lane & 48) + (m_alive ? 31 - __clz((int)m_alive) : 0));
        __builtin_amdgcn_wave_barrier();                     // every lane has read its entry before the compaction writes
        if (keep) lst[num_curr + __popc(m_keep & gbelow)] = prev_pack(nk, nl, ns, pn);
        if (live) {
            num_curr += __popc(m_keep);
            if (m_alive) curr_s = s_last;
            b += kGrp;
            if (b >= num_prev) {                             // this column is done
                num_prev = num_curr;
                if (num_curr == 0) {
                    live = false;                            // nothing survived: PH_BWD_END with an empty list emits nothing
                } else {
                    cur_m = j;
                    j--;
                    b = 0; num_curr = 0; curr_s = -1; first = true;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef BWAMS_BWDDBG
    {
        const unsigned long long tk_end = wall_clock64();
        for (int o = 32; o > 0; o >>= 1) d_items += mk64(__shfl_down((uint32_t)d_items, o), __shfl_down((uint32_t)(d_items >> 32), o));
        if (lane == 0 && d_iter) {
            atomicAdd(&a.ctr->dbg[68], d_items); atomicAdd(&a.ctr->dbg[69], d_iter); atomicAdd(&a.ctr->dbg[70], d_live);
            atomicAdd(&a.ctr->dbg[71], tk_end - tk_start); atomicAdd(&a.ctr->dbg[72], 1ull); atomicMax(&a.ctr->dbg[73], tk_end - tk_start);
            atomicMax(&a.ctr->dbg[74], tk_end); atomicMax(&a.ctr->dbg[75], ~tk_start);
        }
        { unsigned long long e = n_ext; for (int o = 32; o > 0; o >>= 1) e += mk64(__shfl_down((uint32_t)e, o), __shfl_down((uint32_t)(e >> 32), o));
          if (lane == 0) atomicAdd(&a.ctr->dbg[76], e); }
    }
#endif
    n_ext_io += n_ext;
    n_blk_io += n_blk;
}

// The launch behind rounds 1 and 2: every wavefront first takes pivots with long lists (a wavefront each, the longest-running
// items), then pivots with short lists (four at a time) — one launch, one tail.
template <int TAB>
__global__ __launch_bounds__(kBlock) void smem_bwd_kernel(SeedLaunch a) {
    extern __shared__ uint32_t lds_reads[];
    unsigned long long n_ext = 0, n_blk = 0;
    WaveOut wo;
    wo.base = -1; wo.used = 0; wo.emitted = 0;
    bwd_wave_role<TAB>(a, lds_reads, wo, n_ext, n_blk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bwd_group_role<TAB>(a, lds_reads, wo, n_ext, n_blk);
    wave_emit_finish(a, wo);
    flush_counters(a.ctr, n_ext, n_blk);
}
template <int ROLE, int TAB>
__global__ __launch_bounds__(kBlock) void smem_bwd_role_kernel(SeedLaunch a) {
    extern __shared__ uint32_t lds_reads[];
    unsigned long long n_ext = 0, n_blk = 0;
    WaveOut wo;
    wo.base = -1; wo.used = 0; wo.emitted = 0;
    if (ROLE == 0) bwd_wave_role<TAB>(a, lds_reads, wo, n_ext, n_blk);
    else bwd_group_role<TAB>(a, lds_reads, wo, n_ext, n_blk);
    wave_emit_finish(a, wo);
    flush_counters(a.ctr, n_ext, n_blk);
}

// Select round-2 pivots from the round-1 SMEMs (src/bwamem.cpp:721-738).
__global__ void round2_work_kernel(const bwams_smem_t *pool, DevCounters *ctr, Round2Work *work,
                                   int64_t work_cap, int split_len, int split_width) {
    const int64_t n1 = (int64_t)ctr->n_after_r1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1;
         i += (int64_t)gridDim.x * blockDim.x) {
        const bwams_smem_t s = pool[i];
        if (s.rid == kHoleRid) continue;
        const int start = (int)s.m, end = (int)s.n + 1;
        if (end - start < split_len || s.s > split_width) continue;
        const unsigned long long pos = atomicAdd(&ctr->n_work2, 1ull);
        if ((int64_t)pos < work_cap) {
            Round2Work w;
            w.rid = s.rid;
            w.x = (end + start) >> 1;
            w.min_intv = (int32_t)(s.s + 1);
            work[pos] = w;
        }
    }
}

// round 3 ran with a pool and counters of its own: its slots (chunk holes included) go behind the main pool's ...
__global__ void append_r3_kernel(bwams_smem_t *__restrict__ pool, int64_t pool_cap, const bwams_smem_t *__restrict__ pool3, int64_t pool3_cap,
                                 const DevCounters *ctr, const DevCounters *ctr3) {
    const int64_t n12 = (int64_t)ctr->n_smem_total;
    int64_t n3 = (int64_t)ctr3->n_smem_total;
    if (n3 > pool3_cap) n3 = pool3_cap;                   // (the overflow is reported by the finish kernel)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x)
        if (n12 + i < pool_cap) pool[n12 + i] = pool3[i];
}
// ... and its counts into the fields mark_kernel(3) folds in.  A pool that could not hold round 3's records shows as a slot count
// beyond the main pool's capacity: the host grows the pools and runs the stage again.
__global__ void append_r3_finish_kernel(int64_t pool_cap, int64_t pool3_cap, DevCounters *ctr, const DevCounters *ctr3) {
    const unsigned long long n3 = ctr3->n_smem_total;
    ctr->n_smem_total += n3;
    if ((int64_t)n3 > pool3_cap && (int64_t)ctr->n_smem_total <= pool_cap) ctr->n_smem_total = (unsigned long long)pool_cap + (n3 - (unsigned long long)pool3_cap);
    ctr->n_smem3 += ctr3->n_smem3; ctr->n_ext3 += ctr3->n_ext3; ctr->n_blk3 += ctr3->n_blk3;
}

// bookkeeping between rounds (single thread): snapshot the pool cursor, reset the queue
__global__ void mark_kernel(DevCounters *ctr, int which) {
    if (which == 1) { ctr->n_after_r1 = ctr->n_smem_total; ctr->valid_after[0] = ctr->n_smem_valid; }
    if (which == 2) { ctr->n_after_r2 = ctr->n_smem_total; ctr->valid_after[1] = ctr->n_smem_valid; }
    if (which == 3) {                          // round 3 counted apart (it may have run beside round 2): fold it in
        ctr->n_smem_valid += ctr->n_smem3; ctr->n_ext += ctr->n_ext3; ctr->n_ext_blocks += ctr->n_blk3;
        ctr->n_smem3 = ctr->n_ext3 = ctr->n_blk3 = 0;
        ctr->valid_after[2] = ctr->n_smem_valid;
    }
    if (which >= 1 && which <= 3) {
        ctr->ext_after[which - 1] = ctr->n_ext;
        ctr->blk_after[which - 1] = ctr->n_ext_blocks;
    }
    ctr->work_head = 0;
    ctr->bwd_items = ctr->bwd_entries = ctr->bwd_ticket = 0;
    ctr->bwd_items_s = ctr->bwd_ticket_s = 0;
    if (which == 1) ctr->f_items_r[0] = ctr->f_items;
    if (which == 2) ctr->f_items_r[1] = ctr->f_items;
    ctr->f_items = ctr->f_ticket = 0;
    if (which != 2) ctr->work_head3 = 0;       // (mark 2 may run while round 3 is in flight on its own stream... it has joined; kept for symmetry)
}

// Round 3: forward-only seeds.
template <int TAB>
__global__ __launch_bounds__(kBlock) void seed_strategy_kernel(SeedLaunch a, int max_intv) {
    const DevFmi &f = a.fmi;
    extern __shared__ uint32_t lds_reads[];
    uint32_t *const lds_col = a.reads_in_lds ? lds_reads + threadIdx.x : nullptr;
    ReadView rv;
    rv.lds_col = lds_col;
    rv.gl = a.packed;
    rv.cw = a.read_cw;
    int phase = PH_FETCH;
    uint32_t rid = 0;
    int64_t qoff = 0;
    int len = 0, x = 0, next_x = 0, j = 0;
    int64_t ck = 0, cl = 0, cs = 0;
    unsigned long long n_ext = 0, n_blk = 0;
    WaveOut wo;
    wo.base = -1; wo.used = 0; wo.emitted = 0;
    WaveTickets wt;
    wt.next = 0; wt.left = 0; wt.seen = 0;

    while (true) {
        {
            unsigned long long t = 0;
            if (take_ticket(&a.ctr->work_head3, wt, phase == PH_FETCH, t, a.nseq, false)) {
                if ((int64_t)t >= a.nseq) {
                    phase = PH_EXIT;
                } else {
                    rid = (uint32_t)t;
                    qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    x = 0;
                    phase = PH_PIVOT;
                    if (a.skip && a.skip[rid]) phase = PH_FETCH;
                    else read_take(rv, lds_col, a.packed, a.read_w, rid);
                }
            }
        }
        if (__all(phase == PH_EXIT)) break;

        bool em = false;
        uint32_t em_m = 0, em_n = 0;
        int64_t em_k = 0, em_l = 0, em_s = 0;
        if (phase == PH_PIVOT) {
            if (x >= len) {
                phase = PH_FETCH;
            } else {
                const int c = base_at(rv, x);
                next_x = x + 1;
                if (c >= 4) {
                    x = next_x;
                } else {
                    ck = cnt_at(f, c);
                    cl = cnt_at(f, 3 - c);
                    cs = cnt_at(f, c + 1) - ck;
                    j = x + 1;
                    phase = PH_FWD;
                    if (f.last_smem && len - x >= f.last_bp) {
                        // FMA: jump over the longest non-empty prefix of the next last_bp bases (FMI_search.cpp:1705-1750)
                        const int bp = f.last_bp;
                        uint32_t tix = 0;
                        int with_n = 0;
                        for (int kk = 0; kk < bp; ++kk) {
                            const int bb = base_at(rv, x + kk);
                            tix |= (uint32_t)(bb & 3) << ((bp - 1 - kk) * 2);
                            with_n += bb >> 2;
                        }
                        if (with_n == 0) {
                            const uint4 e = f.last_smem[tix];
                            const int ebp = (int)(e.x & 0xff);
                            j = x + ebp;
                            next_x = j;
                            ck = (int64_t)(((uint64_t)(uint32_t)(int32_t)(int8_t)((e.x >> 8) & 0xff) << 32) | e.y);
                            cl = (int64_t)(((uint64_t)(uint32_t)(int32_t)(int8_t)((e.x >> 16) & 0xff) << 32) | e.z);
                            cs = (int64_t)(((uint64_t)(uint32_t)(int32_t)(int8_t)((e.x >> 24) & 0xff) << 32) | e.w);
                            if (cs < max_intv && ebp >= a.min_seed_len && cs > 0) {
                                // emitted WITHOUT ending the pivot (reference quirk); extension resumes next iteration
                                em = true; em_m = (uint32_t)x; em_n = (uint32_t)(j - 1); em_k = ck; em_l = cl; em_s = cs;
                            }
                            phase = PH_HOLD;
                        }
                    }
                }
            }
        }
        bool do_ext = false;
        int ea = 0;
        if (phase == PH_FWD) {
            bool stop = true;
            if (j < len) {
                const int c = base_at(rv, j);
                next_x = j + 1;
                if (c < 4) {
                    do_ext = true;
                    ea = 3 - c;
                    stop = false;
                }
            }
            if (stop) {
                x = next_x;
                phase = PH_PIVOT;
            }
        }
        int64_t nk = 0, nl = 0, ns = 0;
        backward_ext_coop<TAB>(f, do_ext, cl, ck, cs, ea, nk, nl, ns);
        if (do_ext) {
            n_ext++;
            n_blk += ((cl >> 6) == ((cl + cs) >> 6)) ? 1 : 2;
            ck = nl; cl = nk; cs = ns;
            if (cs < max_intv && (j - x + 1) >= a.min_seed_len) {
                em = cs > 0;
                em_m = (uint32_t)x;
                em_n = (uint32_t)j;
                em_k = ck; em_l = cl; em_s = cs;
                x = next_x;
                phase = PH_PIVOT;
            }
            j++;
        }
        if (phase == PH_HOLD) phase = PH_FWD;
        wave_emit(a, wo, em, rid, em_m, em_n, em_k, em_l, em_s);
    }
    wave_emit_finish(a, wo, true);
    flush_counters(a.ctr, n_ext, n_blk, true);
}

// (rid, m, n) sort key of each pooled SMEM
__global__ void make_keys_kernel(const bwams_smem_t *pool, int64_t n, uint64_t *keys, uint32_t *vals,
                                 uint32_t hole_key_rid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bwams_smem_t s = pool[i];
        const uint32_t rid = s.rid == kHoleRid ? hole_key_rid : s.rid;     // holes behind every read
        keys[i] = ((uint64_t)rid << 32) | ((uint64_t)(s.m & 0xffff) << 16) | (uint64_t)(s.n & 0xffff);
        vals[i] = (uint32_t)i;
    }
}

__global__ void gather_sorted_kernel(const bwams_smem_t *pool, const uint32_t *order, int64_t n,
                                     bwams_smem_t *sorted, int64_t *sa_cnt, int max_occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bwams_smem_t s = pool[order[i]];
        sorted[i] = s;
        if (sa_cnt) sa_cnt[i] = s.s < (int64_t)max_occ ? s.s : (int64_t)max_occ;
    }
}

// CpOcc2 from CP_OCC: compact block B = reference blocks 2B and 2B + 1 (the second may not exist)
__global__ void cp2_build_kernel(const uint4 *__restrict__ cp, int64_t n_blk, uint4 *__restrict__ cp2) {
    const int64_t n2 = (n_blk + 1) >> 1;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n2; b += (int64_t)gridDim.x * blockDim.x) {
        const uint4 *p = cp + (2 * b) * 4;
        const uint4 c01 = p[0], c23 = p[1], ac = p[2], gt = p[3];         // one-hot strings: {A, C}, {G, T}
        uint4 ac2 = make_uint4(0, 0, 0, 0), gt2 = make_uint4(0, 0, 0, 0);
        if (2 * b + 1 < n_blk) { ac2 = p[6]; gt2 = p[7]; }
        uint4 hp, lp;
        hp.x = gt.x | gt.z; hp.y = gt.y | gt.w;            // G | T of bases 0-63
        hp.z = gt2.x | gt2.z; hp.w = gt2.y | gt2.w;
        lp.x = ac.z | gt.z; lp.y = ac.w | gt.w;            // C | T
        lp.z = ac2.z | gt2.z; lp.w = ac2.w | gt2.w;
        uint4 *o = cp2 + b * 4;
        o[0] = c01; o[1] = c23; o[2] = hp; o[3] = lp;
    }
}

// the interleaved table from CP_OCC: piece b = {cp_count[b], one_hot_bwt_str[b]}
__global__ void cpi_build_kernel(const uint4 *__restrict__ cp, int64_t n_blk, uint4 *__restrict__ out) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blk; b += (int64_t)gridDim.x * blockDim.x) {
        const uint4 *p = cp + b * 4;
        const uint4 c01 = p[0], c23 = p[1], h01 = p[2], h23 = p[3];
        uint4 *o = out + b * 4;
        o[0] = make_uint4(c01.x, c01.y, h01.x, h01.y);
        o[1] = make_uint4(c01.z, c01.w, h01.z, h01.w);
        o[2] = make_uint4(c23.x, c23.y, h23.x, h23.y);
        o[3] = make_uint4(c23.z, c23.w, h23.z, h23.w);
    }
}

int grid_for(int64_t n_items, int cu_count) {
    int64_t blocks = (n_items + kBlock - 1) / kBlock;
    const int64_t maxb = (int64_t)cu_count * kBlocksPerCU;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

size_t cp2_bytes(int64_t n_blk, int kind) { return kind == 2 ? (size_t)n_blk * 64 + 64 : (size_t)((n_blk + 1) >> 1) * 64 + 64; }
void launch_cp2_build(const uint4 *cp, int64_t n_blk, uint4 *cp2, int kind, hipStream_t st) {
    if (n_blk <= 0) return;
    if (kind == 2) cpi_build_kernel<<<256 * 16, 256, 0, st>>>(cp, n_blk, cp2);
    else cp2_build_kernel<<<256 * 16, 256, 0, st>>>(cp, n_blk, cp2);
}

static size_t lds_bytes(const SeedLaunch &a) { return a.reads_in_lds ? (size_t)a.read_w * kBlock * 4 : 0; }
static size_t lds_bytes_search(const SeedLaunch &a) { return lds_bytes(a) + (size_t)kPrevLds * kBlock * 16 + (size_t)(kBlock / 64) * kPivotQueue * 8; }

void launch_pack_reads(const uint8_t *enc, const int64_t *cum, int64_t nseq, int W, int cw, uint32_t *packed,
                       hipStream_t st) {
    const int64_t n = nseq * W;
    if (n <= 0) return;
    pack_reads_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(enc, cum, nseq, W, cw, packed);
}

void launch_build_fma(const DevFmi &f, int all_bp, uint32_t *all_tab, int last_bp, uint4 *last_tab, hipStream_t st) {
    const int64_t na = (int64_t)1 << (2 * all_bp), nl = (int64_t)1 << (2 * last_bp);
    build_all_smem_kernel<<<(unsigned)((na + 255) / 256), 256, 0, st>>>(f, all_bp, all_tab);
    build_last_smem_kernel<<<(unsigned)((nl + 255) / 256), 256, 0, st>>>(f, last_bp, last_tab);
}

int seed_block_threads() { return kBlock; }
int64_t seed_max_threads(int cu_count) { return (int64_t)cu_count * kBlocksPerCU * kBlock; }
// pool slots the launches of one seeding pass can leave unused in partly filled chunks: every wave of the three search launches
// (rounds 1, 2, 3) and of the launches behind rounds 1 and 2 (smem_bwd_kernel: cu * 8 workgroups, sized for two roles each) may
// abandon one chunk of kChunk slots
int64_t seed_pool_slack(int cu_count) {
    return 5 * (seed_max_threads(cu_count) / 64 * kChunk) + 2 * ((int64_t)cu_count * 8 * (kBlock / 64) * kChunk);
}

void launch_mark(DevCounters *ctr, int which, hipStream_t st) { mark_kernel<<<1, 1, 0, st>>>(ctr, which); }
void launch_append_r3(bwams_smem_t *pool, int64_t pool_cap, const bwams_smem_t *pool3, int64_t pool3_cap, DevCounters *ctr, const DevCounters *ctr3,
                      hipStream_t st) {
    append_r3_kernel<<<256, 256, 0, st>>>(pool, pool_cap, pool3, pool3_cap, ctr, ctr3);
    append_r3_finish_kernel<<<1, 1, 0, st>>>(pool_cap, pool3_cap, ctr, ctr3);
}

void launch_smem_round1(const SeedLaunch &a, int cu_count, hipStream_t st) {
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (tab == 2) smem_search_kernel<true, 2><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, nullptr);
    else if (tab == 1) smem_search_kernel<true, 1><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, nullptr);
    else smem_search_kernel<true, 0><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, nullptr);
}

void launch_round2_work(const SeedLaunch &a, Round2Work *work, int64_t work_cap, int split_len,
                        int split_width, int cu_count, hipStream_t st) {
    round2_work_kernel<<<cu_count * 4, 256, 0, st>>>(a.pool, a.ctr, work, work_cap, split_len, split_width);
}

void launch_smem_round2(const SeedLaunch &a, const Round2Work *work, int cu_count, hipStream_t st) {
    // the number of items is only known on the device: launch the persistent grid at chip size
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (tab == 2) smem_search_kernel<false, 2><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, work);
    else if (tab == 1) smem_search_kernel<false, 1><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, work);
    else smem_search_kernel<false, 0><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes_search(a), st>>>(a, work);
}

void launch_smem_fwd(const SeedLaunch &a, const Round2Work *work, int cu_count, hipStream_t st) {
    const int grid = cu_count * knobs().fwd_bpc;
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (!work) {
        if (tab == 2) smem_fwd_kernel<true, 2><<<grid, kBlock, lds_bytes(a), st>>>(a, nullptr);
        else if (tab == 1) smem_fwd_kernel<true, 1><<<grid, kBlock, lds_bytes(a), st>>>(a, nullptr);
        else smem_fwd_kernel<true, 0><<<grid, kBlock, lds_bytes(a), st>>>(a, nullptr);
    } else {
        if (tab == 2) smem_fwd_kernel<false, 2><<<grid, kBlock, lds_bytes(a), st>>>(a, work);
        else if (tab == 1) smem_fwd_kernel<false, 1><<<grid, kBlock, lds_bytes(a), st>>>(a, work);
        else smem_fwd_kernel<false, 0><<<grid, kBlock, lds_bytes(a), st>>>(a, work);
    }
}

void launch_smem_bwdl(const SeedLaunch &a, int cu_count, hipStream_t st) {
    const size_t lds = lds_bytes(a) + (size_t)kPrevLds * kBlock * 16;
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (tab == 2) smem_bwdl_kernel<2><<<cu_count * knobs().bwdl_bpc, kBlock, lds, st>>>(a);
    else if (tab == 1) smem_bwdl_kernel<1><<<cu_count * knobs().bwdl_bpc, kBlock, lds, st>>>(a);
    else smem_bwdl_kernel<0><<<cu_count * knobs().bwdl_bpc, kBlock, lds, st>>>(a);
}

void launch_smem_bwd_wave(const SeedLaunch &a, int cu_count, hipStream_t st) {
    if (a.bwd_min_list <= 0) return;
    // the number of items is only known on the device; waves without an item leave at their first ticket
    const size_t lds = (size_t)(kBlock / 64) * (kBwdMaxList * 16 + 16 * (size_t)(a.read_w <= kBwdReadLds ? a.read_w : 0));
    const size_t lds_g = lds;
    const bool fused = knobs().bwd_fused != 0;
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (fused) {
        if (tab == 2) smem_bwd_kernel<2><<<cu_count * 8, kBlock, lds > lds_g ? lds : lds_g, st>>>(a);
        else if (tab == 1) smem_bwd_kernel<1><<<cu_count * 8, kBlock, lds > lds_g ? lds : lds_g, st>>>(a);
        else smem_bwd_kernel<0><<<cu_count * 8, kBlock, lds > lds_g ? lds : lds_g, st>>>(a);
    } else if (tab == 2) {                     // (A-B switch: the two roles as two launches)
        smem_bwd_role_kernel<0, 2><<<cu_count * 8, kBlock, lds, st>>>(a);
        smem_bwd_role_kernel<1, 2><<<cu_count * 8, kBlock, lds_g, st>>>(a);
    } else {
        smem_bwd_role_kernel<0, 0><<<cu_count * 8, kBlock, lds, st>>>(a);
        smem_bwd_role_kernel<1, 0><<<cu_count * 8, kBlock, lds_g, st>>>(a);
    }
}

void launch_smem_round3(const SeedLaunch &a, int max_intv, int cu_count, hipStream_t st) {
    const int tab = a.fmi.cp2 ? a.fmi.tab_kind : 0;
    if (tab == 2) seed_strategy_kernel<2><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes(a), st>>>(a, max_intv);
    else if (tab == 1) seed_strategy_kernel<1><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes(a), st>>>(a, max_intv);
    else seed_strategy_kernel<0><<<grid_for(a.nseq, cu_count), kBlock, lds_bytes(a), st>>>(a, max_intv);
}

void launch_make_keys(const bwams_smem_t *pool, int64_t n, uint64_t *keys, uint32_t *vals, uint32_t hole_key_rid,
                      hipStream_t st) {
    if (n <= 0) return;
    make_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pool, n, keys, vals, hole_key_rid);
}

void launch_gather_sorted(const bwams_smem_t *pool, const uint32_t *order, int64_t n, bwams_smem_t *sorted,
                          int64_t *sa_cnt, int max_occ, hipStream_t st) {
    if (n <= 0) return;
    gather_sorted_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pool, order, n, sorted, sa_cnt, max_occ);
}

}  // namespace bwams
