// usim_kernels.hip -- CDNA4 (gfx950) step/reset kernel of the batched Ultrasound simulator.
//
// The step replaces, per environment:
//   a1 robosuite MujocoEnv.step driver            a2 OSC_POSE controller (rl_config.yaml:33-51)
//   a3 MuJoCo mj_step (forward dynamics + soft constraints + Euler)
//   a4 Ultrasound.reward  ultrasound.py:230-269   a5 sensors ultrasound.py:363-401
//   a6 _post_action / _check_terminated ultrasound.py:512-551, 635-670
//   a7 probe<->torso contact predicate ultrasound.py:673-736          a8 utils/quaternion.py
//   a10 reset ultrasound.py:416-478 (trajectory sampling, initial-pose IK, noise, solref randomisation)
// Model and algorithm are specified in DESIGN.md; this file is written independently of oracle/.
//
// Mapping.  G lanes of a wave64 form the group of one environment.
//   rigid torso  G = 1 : one environment per lane, 64 per workgroup, everything in VGPRs, no LDS.
//   soft torso   G = 8 / 16 : 16 environments per workgroup (4096 envs -> 256 workgroups, one per CU, G/4 waves
//                each).  All lanes of a group carry the same 7-DoF arm state (the arm mathematics is replicated, those
//                lanes would otherwise idle) and split the 99-element lattice, the collision tests and the contacts:
//                  - the lattice inverse (99 x 100 fp32) and the element tables are workgroup-resident in LDS; per-environment
//                    scratch is a 548-word LDS block (548 mod 64 = 36 puts the 16-byte windows of the 16 environments on
//                    disjoint bank groups);
//                  - the lattice right-hand side is a 5-point stencil on a zero-bordered grid, the lattice solve
//                    A~ = Linv X (X = the right-hand sides of the wave's environments) runs on the matrix cores
//                    (v_mfma_f32_4x4x1, step kernel) -- the one dense contraction of the path;
//                  - contacts keep ascending shell-id order through a wave ballot;
//                  - contact k lives in the registers of lane k; the dual problem is solved on 3 x 3 Delassus blocks held
//                    per lane, a Gauss-Seidel visit broadcasts three force increments with DPP row_newbcast.
// Per-environment state is read and written once per step as rows of the SoA state block in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "usim_device.h"
#include "usim_devmath.h"

namespace usim {

// lattice tables, laid out exactly as their workgroup-resident LDS copy.  One device buffer per handle (DevModel::tables, built and
// uploaded by usim_create): handles with different torso shapes can live side by side on one GPU
constexpr int LROW = 100;                             // row stride of the lattice inverse (pad word zero)
constexpr int TB_LINV = 0;                            // float [99][100]
constexpr int TB_POS = N_TOP * LROW;                  // float [99][3] nominal surface point rel. torso centre (padded to 300)
constexpr int TB_AXIS = TB_POS + 300;                 // float [99][3] slide axis
constexpr int TB_SHELL = TB_AXIS + 300;               // int   [99]    shell id (contact-pair index convention)
constexpr int TB_WORDS = TB_SHELL + 100;              // 10600 words (the lattice topology itself is implicit: 9 x 11 grid stencil)
constexpr int TB_ARM = TB_WORDS;                      // behind the lattice block (not copied to LDS): the arm table of the 16-lane step kernel
constexpr int TB_TOTAL = TB_ARM + A16_LANES * AT_STRIDE;

// per-environment LDS block (word offsets); GE_X must stay 16-byte aligned
constexpr int GE_X = 0;                               // rhs[100] of the lattice solve
constexpr int GE_S = 100;                             // (free: element positions stay in their owners' registers)
constexpr int GE_SD = 200;                            // sdot[99]
constexpr int GE_A = 300;                             // lattice acceleration a~[99]
constexpr int GE_U = 300;                             // zero-bordered 11 x 13 grid of u = k_t s + b_t sdot; dead before a~ and the contact records land
constexpr int GE_U_WORDS = 144;                       // (LAT_NA + 2) * (LAT_NC + 2) = 143, rounded to 16 bytes
constexpr int CG_WORDS = 8;                           // contact record: n3, r3, element, distance
constexpr int GE_CG = 400;                            // contact records 8 x 8
constexpr int GE_WS = 464;                            // per-contact wrench + element impulse 8 x 8 (before that: candidate records 8..15)
constexpr int MAXCAND = 16;                           // penetrating elements recorded before the MAXC deepest are kept (+ one spare record for misses)
constexpr int GE_STRIDE = 548;                        // 548 mod 64 = 36: disjoint 16-byte bank windows for 16 environments
static_assert(GE_WS + MAXC * 8 <= GE_STRIDE && GE_CG + (MAXCAND + 1) * CG_WORDS <= GE_STRIDE && GE_U + GE_U_WORDS <= GE_STRIDE, "per-environment LDS block overflow");
static_assert((LAT_NA + 2) * (LAT_NC + 2) <= GE_U_WORDS && LAT_NA * LAT_NC == N_TOP, "padded lattice grid");

template <int G> struct GroupGeom {
    static constexpr int EPW = 64 / G;                // environments per wave
    static constexpr int WAVES = (G >= 4) ? G / 4 : 1;
    static constexpr int EPB = EPW * WAVES;           // environments per workgroup (64 for G = 1, else 16)
    static constexpr int NT = 64 * WAVES;
    static constexpr int LDS_WORDS = TB_WORDS + EPB * GE_STRIDE;
};
// two workgroups of the 8-lane mapping share a CU at 8192 envs/GPU (2 waves each): both must fit the 160 KB of LDS
static_assert(2 * GroupGeom<8>::LDS_WORDS * 4 <= 160 * 1024, "LDS block too large for two workgroups per CU");

DI void group_sync() {
    // the lanes of a group exchange data through LDS inside one wave: order the LDS traffic, no s_barrier needed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// value of lane k (0..7) of every group, delivered to all lanes of the group (DPP row_newbcast, gfx90a+).  k is a
// compile-time constant after unrolling, so the switch folds to one v_mov_b32_dpp (two for G = 8).
// (__builtin_amdgcn_mov_dpp has no tied `old` operand: one v_mov_b32_dpp; update_dpp(iv, iv, ...) costs a register copy first -- three per pair and iteration of the
//  contact solve's matrix-vector product, 6 - 10 % of an iteration)
#define USIM_BCAST_CASE(K)                                                                               \
    case K:                                                                                              \
        if (G == 16) iv = __builtin_amdgcn_mov_dpp(iv, 0x150 + K, 0xf, 0xf, false);                      \
        else if (G == 8) {                                                                               \
            int t = __builtin_amdgcn_mov_dpp(iv, 0x150 + K, 0xf, 0x3, false);            /* lanes 0-7 of the row <- lane K (lanes 8-15: overwritten next) */   \
            iv = __builtin_amdgcn_update_dpp(t, iv, 0x150 + 8 + K, 0xf, 0xc, false);     /* lanes 8-15 <- lane 8 + K */          \
        }                                                                                                \
        break;
template <int G>
DI float group_bcast(float v, int k) {
    int iv = __float_as_int(v);
    switch (k) {
        USIM_BCAST_CASE(0) USIM_BCAST_CASE(1) USIM_BCAST_CASE(2) USIM_BCAST_CASE(3)
        USIM_BCAST_CASE(4) USIM_BCAST_CASE(5) USIM_BCAST_CASE(6) USIM_BCAST_CASE(7)
        default: break;
    }
    return __int_as_float(iv);
}
#undef USIM_BCAST_CASE
// 16-lane groups: lanes 0-7 of the row receive the value of lane J, lanes 8-15 that of lane 4 + J (J = 0..3, a compile-time constant after unrolling): two bank-masked
// v_mov_b32_dpp row_newbcast
DI float half_bcast(float v, int J) {
    const int iv = __float_as_int(v);
    int r = iv;
    switch (J) {
        case 0: { const int t = __builtin_amdgcn_mov_dpp(iv, 0x150 + 0, 0xf, 0x3, false); r = __builtin_amdgcn_update_dpp(t, iv, 0x150 + 4, 0xf, 0xc, false); } break;
        case 1: { const int t = __builtin_amdgcn_mov_dpp(iv, 0x150 + 1, 0xf, 0x3, false); r = __builtin_amdgcn_update_dpp(t, iv, 0x150 + 5, 0xf, 0xc, false); } break;
        case 2: { const int t = __builtin_amdgcn_mov_dpp(iv, 0x150 + 2, 0xf, 0x3, false); r = __builtin_amdgcn_update_dpp(t, iv, 0x150 + 6, 0xf, 0xc, false); } break;
        case 3: { const int t = __builtin_amdgcn_mov_dpp(iv, 0x150 + 3, 0xf, 0x3, false); r = __builtin_amdgcn_update_dpp(t, iv, 0x150 + 7, 0xf, 0xc, false); } break;
        default: break;
    }
    return __int_as_float(r);
}

// phase timeline probe (diagnostics only; profiling build): wave 0 of workgroup 0 stamps the shader clock when a buffer is given
#if !defined(USIM_TSTAMP) && !defined(USIM_TSTAMP_NOWAIT)
#define USIM_STAMP(dbg, k) do { } while (0)
#elif defined(USIM_TSTAMP_NOWAIT)
#define USIM_STAMP(dbg, k) do { if ((dbg) && blockIdx.x == 0 && threadIdx.x == 0) (dbg)[k] = __builtin_readcyclecounter(); } while (0)
#else
#define USIM_STAMP(dbg, k) do { if ((dbg) && blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); (dbg)[k] = __builtin_readcyclecounter(); } } while (0)
#endif

// contact-solve phases of the lattice wave of the split kernel (first lattice wave of workgroup 0): dbg[40..45]
#if !defined(USIM_TSTAMP) && !defined(USIM_TSTAMP_NOWAIT)
#define USIM_CSTAMP(dbg, k) do { } while (0)
#else
#define USIM_CSTAMP(dbg, k) do { if ((dbg) && blockIdx.x == 0 && threadIdx.x == 256) (dbg)[40 + (k)] = __builtin_readcyclecounter(); } while (0)
#endif

// prescribed torso base motion (usim_config.torso_drop): none since round 4 (drop = 0: the torso stands on its rim capsules at the spawn height);
// torso_drop = 1: free fall over the 4.7 mm spawn gap of ultrasound.py:313, then rest
DI void torso_motion(const DevCfg& C, int tsim, float& dz, float& vz, float& az) {
    dz = -C.drop; vz = 0.f; az = 0.f;
    if (C.torso_drop) {
        float tt = (float)tsim * C.dt, zf = -0.5f * GRAV * tt * tt;
        if (zf > -C.drop) { dz = zf; vz = -GRAV * tt; az = -GRAV; }
    }
}

// Probe collision geometry (stand-in for the missing mesh, ultrasound_probe_gripper.xml:3,8; MuJoCo collides the convex hull of a mesh): a flared
// blade = convex hull of two parallel capsules of half-length probe_hl along the site x axis -- the tip capsule (radius probe_r, axis probe_r above
// the tip = grip_site) and an upper capsule (radius probe_r2, axis probe_h above the tip capsule's).  Signed distance of a point given in the site
// frame (site z points from the tip away from the probe body) and its gradient: the round-cone distance on the cross-section.
DI float probe_sdf(const DevCfg& C, const f3 p, f3& g) {
    // (v_rsq_f32 is a quarter-rate instruction like v_sqrt_f32 / v_rcp_f32: every length and its reciprocal come from one of them)
    // round 4: the cross-section is swept sideways by +-probe_hw as it is lengthways by +-probe_hl (a flat 2 hl x 2 hw face with edges of radius probe_r), and the
    // lowest point lies probe_tip beyond the site along its z axis
    const float py = C.probe_tip - p.z - C.probe_r, e = fmaxf(fabsf(p.x) - C.probe_hl, 0.f), el = fmaxf(fabsf(p.y) - C.probe_hw, 0.f);
    const float px2 = fmaf(el, el, e * e);
    const bool pxok = px2 > 1e-18f;
    const float ipx = pxok ? rsq_(px2) : 0.f, px = px2 * ipx;
    const float kk = fmaf(py, C.probe_ca, -(px * C.probe_cb));
    const bool low = kk < 0.f, flank = !low && !(kk > C.probe_cah);
    const float qy = py - C.probe_h;
    const float lc2 = fmaf(qy, qy, px2), ll2 = fmaf(py, py, px2);
    const bool okc = lc2 > 1e-18f, okl = ll2 > 1e-18f;
    const float ilc = okc ? rsq_(lc2) : 0.f, ill = okl ? rsq_(ll2) : 0.f;             // seen from the centre of the upper circle of the cross-section / of the tip circle
    const float cx = px * ilc, cy = qy * ilc;
    float d = low ? ll2 * ill - C.probe_r : lc2 * ilc - C.probe_r2;
    float gx = low ? px * ill : cx, gy = low ? (okl ? py * ill : -1.f) : (okc ? cy : -1.f);
    if (flank) { d = fmaf(px, C.probe_ca, fmaf(py, C.probe_cb, -C.probe_r)); gx = C.probe_ca; gy = C.probe_cb; }
    // direction field: the distance gradient is undefined on the medial axis of the body (the tip capsule's axis, probe_r below the surface, and
    // the centre plane above it); between 2/3 and 0.96 probe_r below the surface the direction turns into the one seen from the upper centre
    // (like the centre-to-centre search direction of a convex collider).  The distance stays exact.
    const float beta = okc ? clampf((-d - C.probe_deep0) * C.probe_inv_band, 0.f, 1.f) : 0.f;
    const float bx = fmaf(beta, cx - gx, gx), by = fmaf(beta, cy - gy, gy);
    const float rn = beta > 0.f ? rsq_(fmaf(bx, bx, by * by)) : 1.f;
    gx = bx * rn; gy = by * rn;
    const float gxi = gx * ipx;
    // (on the probe's axis the lateral direction is undefined: the lateral part of the direction, if any -- the blended field of the deep band -- goes to the site's y axis, so that g stays a unit vector)
    g = mk(copysignf(gxi * e, p.x), pxok ? copysignf(gxi * el, p.y) : gx, -gy);
    return d;
}

// One collision round: the probe blade against element i G + gl of this lane's environment (one element per lane); the wave ballot gives every
// hit its slot in the record area `recs` (MAXCAND records + one spare for misses), so that the list stays sorted by ascending shell id.
// Straight-line code (a miss writes its record to the spare slot): the rounds can be scheduled between the matrix instructions of the lattice
// solve.  Element = capsule (soft_box.xml:10): axis segment from the cap centre `tip` (t = 0) to the inner end (t = 1); the probe distance d(t)
// is convex along it.  With the slopes s0, s1 at the two ends, t minimises the quadratic model d0 + s0 t + (s1 - s0 + eps) t^2 / 2 on [0, 1];
// eps settles the point near the cap when the shaft lies flat against a flank of the probe (every point equally close: the plain minimiser
// would be ill-conditioned).
template <int G>
DI void collide_elem(const float* lds, float* recs, const bool valid, const int e, const int gl, const int gbase, const DevModel& M, const DevCfg& C, const float se, const float dz,
                     const f3 Kx, const f3 Ksx, const f3 Ksy, const f3 Ksz, int& nc) {
    const f3 ax = mk(lds[TB_AXIS + 3 * e], lds[TB_AXIS + 3 * e + 1], lds[TB_AXIS + 3 * e + 2]);
    const f3 tip = mk(M.torso[0] + lds[TB_POS + 3 * e], M.torso[1] + lds[TB_POS + 3 * e + 1], M.torso[2] + lds[TB_POS + 3 * e + 2] + dz) + ax * (se - ELEM_R);
    const f3 rel = tip - Kx;
    const f3 p0 = mk(dot(Ksx, rel), dot(Ksy, rel), dot(Ksz, rel));               // site frame
    const f3 us = mk(dot(Ksx, ax), dot(Ksy, ax), dot(Ksz, ax)) * (-2.f * ELEM_HL);
    f3 g0, g1, gs;
    (void)probe_sdf(C, p0, g0);
    (void)probe_sdf(C, p0 + us, g1);
    const float s0 = dot(g0, us), s1 = dot(g1, us);
    const float tt = clampf(-s0 * rcp_(fmaxf(s1 - s0, 0.f) + SHAFT_EPS), 0.f, 1.f);
    const float dist = probe_sdf(C, madd(p0, us, tt), gs) - ELEM_R;
    const bool hit = valid && (dist < 0.f);
    const f3 nn = (Ksx * gs.x + Ksy * gs.y + Ksz * gs.z) * -1.f;                 // from the element towards the probe
    const f3 rr = madd(tip, ax, -2.f * ELEM_HL * tt) + nn * (ELEM_R + 0.5f * dist) - Kx;
    const unsigned long long bal = __ballot(hit);
    const unsigned gm = (unsigned)(bal >> gbase) & ((1u << G) - 1u);
    const int slot = nc + __popc(gm & ((1u << gl) - 1u));
    const int sl = (hit && slot < MAXCAND) ? slot : MAXCAND;                      // MAXCAND = the spare record
    float4* rec = reinterpret_cast<float4*>(&recs[sl * CG_WORDS]);
    rec[0] = make_float4(nn.x, nn.y, nn.z, rr.x);
    rec[1] = make_float4(rr.y, rr.z, __int_as_float(e), dist);
    nc += __popc(gm);
}

// round form: element i G + gl of this lane's environment (the single-wave kernels walk all rounds)
template <int G>
DI void collide_one(const float* lds, float* recs, const int i, const int gl, const int gbase, const DevModel& M, const DevCfg& C, const float se, const float dz,
                    const f3 Kx, const f3 Ksx, const f3 Ksy, const f3 Ksz, int& nc) {
    const int eraw = i * G + gl;
    collide_elem<G>(lds, recs, eraw < N_TOP, eraw < N_TOP ? eraw : N_TOP - 1, gl, gbase, M, C, se, dz, Kx, Ksx, Ksy, Ksz, nc);
}

// Broad phase (split kernel).  The probe -- convex hull of two parallel capsules whose axes are probe_h apart -- lies inside the capsule of
// radius probe_r + probe_h around its UPPER axis; an element's capsule lies inside the ball of radius ELEM_HL + ELEM_R around the midpoint of its
// axis segment.  An element whose ball misses that capsule cannot touch the probe: 80 % of the 99 elements, for 20 instructions each instead
// of three distance evaluations.  The survivors go, in ascending element order, into a queue that the lanes of the group then walk together
// (collide_queue): typically one pass of 16 instead of seven rounds.
DI bool collide_cull(const float* lds, const int e, const DevModel& M, const DevCfg& C, const float se, const float dz, const f3 Kx, const f3 Ksx, const f3 Ksz) {
    const f3 ax = mk(lds[TB_AXIS + 3 * e], lds[TB_AXIS + 3 * e + 1], lds[TB_AXIS + 3 * e + 2]);
    const f3 mid = mk(M.torso[0] + lds[TB_POS + 3 * e], M.torso[1] + lds[TB_POS + 3 * e + 1], M.torso[2] + lds[TB_POS + 3 * e + 2] + dz) + ax * (se - ELEM_R - ELEM_HL);
    const f3 d = mid - (Kx - Ksz * (C.probe_r + C.probe_h - C.probe_tip));         // from the centre of the upper axis (site z points away from the probe body)
    const f3 v = d - Ksx * clampf(dot(d, Ksx), -C.probe_hl, C.probe_hl);
    return dot(v, v) < C.probe_cull2;
}
template <int G>
DI void collide_queue(const float* lds, const float* queue, const int q0, const int q1, float* recs, const int gl, const int gbase, const DevModel& M, const DevCfg& C,
                      const float* s_lds, const float dz, const f3 Kx, const f3 Ksx, const f3 Ksy, const f3 Ksz, int& nc) {
    for (int j0 = q0; __any(j0 < q1); j0 += G) {
        const int j = j0 + gl;
        const bool valid = j < q1;
        const int e = valid ? __float_as_int(queue[j]) : 0;
        collide_elem<G>(lds, recs, valid, e, gl, gbase, M, C, s_lds[e], dz, Kx, Ksx, Ksy, Ksz, nc);
    }
}

// More penetrating elements than contact slots (rare in a mixed batch, common right after a synchronous reset): keep the MAXC deepest of the
// first MAXCAND candidates (ties keep the lower id), list still in ascending shell id.  The caller clamps its count to MAXC afterwards.
template <int G>
DI void contact_overflow(float* lds, const int eb, const int gl, const int gbase, const int nc) {
#define EBF(off) lds[TB_WORDS + eb * GE_STRIDE + (off)]
    if (nc > MAXC) {
                        // more penetrating elements than contact slots (rare in a mixed batch, common right after a synchronous reset: 4 % of
                        // the environments).  Keep the MAXC deepest of the first MAXCAND candidates (ties keep the lower id), list still in
                        // ascending shell id.
                        group_sync();
                        if constexpr (G == 16) {
                            // one candidate per lane of the group: rank by depth with sixteen row broadcasts, compact with a ballot
                            const int m = nc < MAXCAND ? nc : MAXCAND;
                            const bool cand = gl < m;
                            const float4* rec = reinterpret_cast<const float4*>(&EBF(GE_CG + gl * CG_WORDS));
                            const float4 r0 = rec[0], r1 = rec[1];
                            const float d = cand ? r1.w : 1.0f;
                            int rank = 0;
#define USIM_RANK_STEP(I) { const float di = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x150 + I, 0xf, 0xf, true)); \
                            rank += (di < d || (di == d && I < gl)) ? 1 : 0; }
                            USIM_RANK_STEP(0) USIM_RANK_STEP(1) USIM_RANK_STEP(2) USIM_RANK_STEP(3) USIM_RANK_STEP(4) USIM_RANK_STEP(5) USIM_RANK_STEP(6) USIM_RANK_STEP(7)
                            USIM_RANK_STEP(8) USIM_RANK_STEP(9) USIM_RANK_STEP(10) USIM_RANK_STEP(11) USIM_RANK_STEP(12) USIM_RANK_STEP(13) USIM_RANK_STEP(14) USIM_RANK_STEP(15)
#undef USIM_RANK_STEP
                            const bool keep = cand && rank < MAXC;
                            const unsigned gm = (unsigned)(__ballot(keep) >> gbase) & 0xffffu;
                            const int slot = __popc(gm & ((1u << gl) - 1u));
                            group_sync();                                  // every record is in registers before any slot is overwritten
                            if (keep) {
                                float4* dst = reinterpret_cast<float4*>(&EBF(GE_CG + slot * CG_WORDS));
                                dst[0] = r0; dst[1] = r1;
                            }
                        } else if (gl == 0) {
                            // (8 lanes per environment: one lane of the group edits the records in place)
                            const int m = nc < MAXCAND ? nc : MAXCAND;
                            for (int drop = m - MAXC; drop > 0; --drop) {
                                int worst = 0; float wd = -1.0e30f;
                                for (int j = 0; j < m; ++j) {
                                    const float dj = EBF(GE_CG + j * CG_WORDS + 7);
                                    if (dj < 0.f && dj >= wd) { wd = dj; worst = j; }
                                }
                                EBF(GE_CG + worst * CG_WORDS + 7) = 1.0f;                       // dropped
                            }
                            int wpos = 0;
                            for (int j = 0; j < m; ++j) {
                                if (EBF(GE_CG + j * CG_WORDS + 7) < 0.f) {
                                    if (wpos != j)
                                        for (int a = 0; a < CG_WORDS; ++a) EBF(GE_CG + wpos * CG_WORDS + a) = EBF(GE_CG + j * CG_WORDS + a);
                                    ++wpos;
                                }
                            }
                        }
                    }
#undef EBF
}

// Lattice front end of one forward pass, executed by the G lanes of a group on the group's LDS block: stage (s, sdot), build
// the right-hand side of the soft-equality system, a~ = Linv rhs, collide the probe capsule with the 99 cap spheres and
// leave the contact records (ascending shell id; the MAXC deepest when more were found) in LDS.  Returns the number found (may exceed MAXC).
// PART 0: everything; 1: staging + right-hand side only (needs no arm quantity); 2: solve + collision only (after a PART 1 call); 3: solve only
// (needs no arm quantity either: the split kernel's lattice side runs it while it would otherwise wait for the site pose); 4: collision only (after PART 3).
template <int G, int NE, bool MM, int PART = 0, bool QM = false>
DI int lattice_front(float* lds, const int eb, const int gl, const int gbase, const DevModel& M, const DevCfg& C, const int tsim,
                     const float kst, const float kdmp, const bool live, const float* s_pre, const float* sd_pre,
                     const f3 Kx, const f3 Ksy, const f3 Ksz, unsigned long long* dbg, const float* queue = nullptr, const int q0 = 0, const int q1 = 0) {
#if !defined(USIM_TSTAMP) && !defined(USIM_TSTAMP_NOWAIT)
#define LSTAMP(k) do { } while (0)
#else
#define LSTAMP(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = __builtin_readcyclecounter(); } while (0)
#endif
#define EBF(off) lds[TB_WORDS + eb * GE_STRIDE + (off)]
    float dz, vz, az;
    torso_motion(C, tsim, dz, vz, az);
                    if constexpr (PART == 0 || PART == 1) {
                                    // ---- stage s, sdot and the spring-damper potential u = k_t s + b_t sdot: lane gl of the group owns elements
                    //      gl, gl+G, ...  u goes into a zero-bordered 11 x 13 copy of the 9 x 11 grid, so that the four neighbours of an
                    //      element are four unconditional reads; a pinned rim neighbour (s = 0) is a border cell.
                    // (the loads of s, sdot were issued together with the scalar state at the top of the kernel)
                    const float kfix = 1.0f / (SI_DMAX * SR_TC * SR_TC), bfix = 2.0f / (SI_DMAX * SR_TC);
                    const float kten = kst * (1.0f / SI_DMAX), bten = kdmp * (1.0f / SI_DMAX);
                    {
                        float4* uz = reinterpret_cast<float4*>(&EBF(GE_U));
    #pragma unroll
                        for (int v = gl; v < GE_U_WORDS / 4; v += G) uz[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    int up[NE];
                    float u_own[NE];
    #pragma unroll
                    for (int i = 0; i < NE; ++i) {
                        const int e = gl + i * G;
                        const int ix = (e * 373) >> 12, iz = e - LAT_NC * ix;            // e / 11 for e < 682
                        up[i] = (ix + 1) * (LAT_NC + 2) + iz + 1;
                        const float se = live ? s_pre[i] : 0.f, sde = live ? sd_pre[i] : 0.f;
                        u_own[i] = fmaf(kten, se, bten * sde);
                        if (e < N_TOP) { EBF(GE_SD + e) = sde; EBF(GE_U + up[i]) = u_own[i]; }     // (s itself stays in the owner's registers)
                    }
                    group_sync();
                                    // ---- lattice right-hand side: a_s + w_fix aref_fix + w_ten sum_j aref_ij
                    //      = a_s - w_fix (k_fix s + b_fix sdot) - w_ten (deg u - sum of the four neighbour cells), deg = 4 (3 at the corners) ----
    #pragma unroll
                    for (int i = 0; i < NE; ++i) {
                        const int e = gl + i * G;
                        if (e >= N_TOP) continue;
                        const float se = live ? s_pre[i] : 0.f, sde = live ? sd_pre[i] : 0.f;
                        const float* uc = &EBF(GE_U + up[i]);
                        const float nb = (uc[-1] + uc[1]) + (uc[-(LAT_NC + 2)] + uc[LAT_NC + 2]);
                        const bool corner = (e == 0) | (e == LAT_NC - 1) | (e == N_TOP - LAT_NC) | (e == N_TOP - 1);
                        float r = -(GRAV + az) * lds[TB_AXIS + 3 * e + 2] - M.wfix * fmaf(bfix, sde, kfix * se);
                        r = fmaf(-M.wten, fmaf(corner ? 3.f : 4.f, u_own[i], -nb), r);
                        EBF(GE_X + e) = r;
                    }
                    if (gl == 0) EBF(GE_X + N_TOP) = 0.f;          // pad word read by the 16-byte row chunks
                    group_sync();
                    }
                    if constexpr (PART == 1) return 0;
                    LSTAMP(5);
                                    // ---- collision: seven rounds of one element per lane (collide_one), scheduled between the pieces of the matrix-core
                    //      solve below -- or, in the split kernel (QM), this wave's share [q0, q1) of the broad phase's queue (collide_cull / collide_queue;
                    //      the arm wave, which has the site pose first, builds the queue and takes the other share; the two hit lists are merged
                    //      after hand-off (2)) ----
                    const f3 Ksx = cross(Ksy, Ksz);
                    int nc = 0;
                    // single-wave kernels: broad phase and narrow phase in one go -- element positions into the environment's LDS block, the ids
                    // that pass collide_cull (ascending) into a queue that overlays the a~ area (free until the solve stores its result), then the
                    // queue G elements at a time.  Same per-element arithmetic and the same ballot slots as walking all rounds.
                    auto collide_all = [&]() {
                        float* const q = &EBF(GE_A);
                        int nq = 0;
#pragma unroll
                        for (int i = 0; i < NE; ++i) {
                            const int eraw = i * G + gl, e = eraw < N_TOP ? eraw : N_TOP - 1;
                            const float se = live ? s_pre[i] : 0.f;
                            if (eraw < N_TOP) EBF(GE_S + eraw) = se;
                            const bool cand = (eraw < N_TOP) && collide_cull(lds, e, M, C, se, dz, Kx, Ksx, Ksz);
                            const unsigned gm = (unsigned)(__ballot(cand) >> gbase) & ((1u << G) - 1u);
                            if (cand) q[nq + __popc(gm & ((1u << gl) - 1u))] = __int_as_float(e);
                            nq += __popc(gm);
                        }
                        group_sync();
                        collide_queue<G>(lds, q, 0, nq, &EBF(GE_CG), gl, gbase, M, C, &EBF(GE_S), dz, Kx, Ksx, Ksy, Ksz, nc);
                        group_sync();                                      // the queue area is rewritten by the solve's result
                    };
                    auto collide_round = [&](const int i) {
                        if (i != 0 || PART == 3) return;
                        if constexpr (QM) collide_queue<G>(lds, queue, q0, q1, &EBF(GE_CG), gl, gbase, M, C, &EBF(GE_S), dz, Kx, Ksx, Ksy, Ksz, nc);
                        else collide_all();
                    };
                                    // ---- a~ = Linv * rhs ----
                    if constexpr (PART == 4) {
                        collide_round(0);
                    } else if constexpr (MM) {
                        // Matrix-core form (every lane of the wave is active here): the wave's environments are the columns of one dense
                        // product A~[99 x EPW] = Linv[99 x 100] X[100 x EPW], issued as v_mfma_f32_4x4x1 (16 blocks of 4 rows x 4 columns
                        // per instruction).  Lane l feeds Linv row l (and row 64 + l) as the A operand and the rhs of environment l % 4 as
                        // the B operand; it receives rows 4 (l / 4) .. + 3 of that environment (layout: tools/probe/mfma_4x4x1_layout.hip).
                        // The k loop is unrolled in NE pieces with one collision round after each: the rounds do not depend on the
                        // product and fill the issue slots the matrix pipeline leaves free.
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        constexpr int EPW = 64 / G, NSET = (EPW >= 4 && EPW <= 8) ? EPW / 4 : 1, NCH = LROW / 4;
                        const int lane = gbase + gl, ebw = eb - gbase / G, blk = lane >> 2;
                        const int r1 = (64 + lane < N_TOP) ? 64 + lane : N_TOP - 1;
                        const float4* la0 = reinterpret_cast<const float4*>(&lds[TB_LINV + lane * LROW]);
                        const float4* la1 = reinterpret_cast<const float4*>(&lds[TB_LINV + r1 * LROW]);
                        const float4* xb[NSET];
                        v4f acc0[NSET], acc1[NSET];
    #pragma unroll
                        for (int u = 0; u < NSET; ++u) {
                            xb[u] = reinterpret_cast<const float4*>(&lds[TB_WORDS + (ebw + 4 * u + (lane & 3)) * GE_STRIDE + GE_X]);
                            acc0[u] = (v4f){0.f, 0.f, 0.f, 0.f}; acc1[u] = (v4f){0.f, 0.f, 0.f, 0.f};
                        }
    #pragma unroll
                        for (int i = 0; i < NE; ++i) {
    #pragma unroll
                            for (int c = (i * NCH) / NE; c < ((i + 1) * NCH) / NE; ++c) {
                                const float4 a0 = la0[c], a1 = la1[c];
    #pragma unroll
                                for (int u = 0; u < NSET; ++u) {
                                    const float4 b = xb[u][c];
                                    acc0[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, b.x, acc0[u], 0, 0, 0);
                                    acc1[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, b.x, acc1[u], 0, 0, 0);
                                    acc0[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, b.y, acc0[u], 0, 0, 0);
                                    acc1[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, b.y, acc1[u], 0, 0, 0);
                                    acc0[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, b.z, acc0[u], 0, 0, 0);
                                    acc1[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, b.z, acc1[u], 0, 0, 0);
                                    acc0[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, b.w, acc0[u], 0, 0, 0);
                                    acc1[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, b.w, acc1[u], 0, 0, 0);
                                }
                            }
                            collide_round(i);
                        }
    #pragma unroll
                        for (int u = 0; u < NSET; ++u) {
                            float* dst = &lds[TB_WORDS + (ebw + 4 * u + (lane & 3)) * GE_STRIDE + GE_A];
                            *reinterpret_cast<float4*>(&dst[4 * blk]) = make_float4(acc0[u][0], acc0[u][1], acc0[u][2], acc0[u][3]);
                            if (64 + 4 * blk < LROW)                       // rows 64..99 (word 99 is padding)
                                *reinterpret_cast<float4*>(&dst[64 + 4 * blk]) = make_float4(acc1[u][0], acc1[u][1], acc1[u][2], acc1[u][3]);
                        }
                        group_sync();
                    } else {
                        // VALU form for launches whose environments may be masked: lane gl computes rows gl, gl+G, ...; Linv rows and rhs
                        // are read as 16-byte chunks.  (The collision runs first: its queue overlays the a~ area this product fills.)
                        collide_round(0);
                        const float4* xv = reinterpret_cast<const float4*>(&EBF(GE_X));
                        constexpr int RB = 4;                              // rows per pass share one read of the rhs chunk
    #pragma unroll
                        for (int i0 = 0; i0 < NE; i0 += RB) {
                            const float4* lr[RB];
                            float acc[RB], bcc[RB];
    #pragma unroll
                            for (int j = 0; j < RB; ++j) {
                                int r = gl + (i0 + j) * G; if (r >= N_TOP) r = N_TOP - 1;
                                lr[j] = reinterpret_cast<const float4*>(&lds[TB_LINV + r * LROW]);
                                acc[j] = 0.f; bcc[j] = 0.f;
                            }
    #pragma unroll 5
                            for (int c = 0; c < LROW / 4; ++c) {
                                const float4 x = xv[c];
    #pragma unroll
                                for (int j = 0; j < RB; ++j) {
                                    const float4 u = lr[j][c];
                                    acc[j] = fmaf(u.x, x.x, acc[j]); bcc[j] = fmaf(u.y, x.y, bcc[j]);
                                    acc[j] = fmaf(u.z, x.z, acc[j]); bcc[j] = fmaf(u.w, x.w, bcc[j]);
                                }
                            }
    #pragma unroll
                            for (int j = 0; j < RB; ++j) {
                                const int r = gl + (i0 + j) * G;
                                if (i0 + j < NE && r < N_TOP) EBF(GE_A + r) = acc[j] + bcc[j];
                            }
                        }
                        LSTAMP(6);
                    }
                    if constexpr (!QM && PART != 3) contact_overflow<G>(lds, eb, gl, gbase, nc);
    return nc;
#undef EBF
#undef LSTAMP
}

// sum over the lanes of a group, delivered to every lane (three or four DPP steps; lane k + 8 first, so that a 16-lane group whose halves carry the two contacts
// of a pair adds in the order of an 8-lane group that holds both in one lane: the same bits)
template <int G>
DI float group_allsum(float x) {
    if constexpr (G == 16) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));   // row_ror:8
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));                         // row_half_mirror: k <-> 7 - k
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));                          // quad_perm [1 0 3 2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));                          // quad_perm [2 3 0 1]
    return x;
}

// Two floats per lane, handled by the packed float32 instructions of the part (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32: both halves in one issue slot): with 8-lane
// groups a lane carries BOTH contacts of a probe-element pair, and their visits are the same instruction sequence on different data.  Every operation below is the
// scalar one per component (IEEE fma / mul / add; v_rcp, v_rsq, min, max and the selects run once per half), so a pair visited in one lane has the bits of a pair
// visited in lanes k and 8 + k of a 16-lane group.
typedef float v2f __attribute__((ext_vector_type(2)));
struct b2 { bool x, y; };
template <class T> struct LaneVec;
template <> struct LaneVec<float> {
    typedef bool mask;
    static DI float splat(float a) { return a; }
    static DI float fma(float a, float b, float c) { return fmaf(a, b, c); }
    static DI float rcp(float a) { return rcp_(a); }
    static DI float rsq(float a) { return rsq_(a); }
    static DI float max(float a, float b) { return fmaxf(a, b); }
    static DI float min(float a, float b) { return fminf(a, b); }
    static DI bool gt(float a, float b) { return a > b; }
    static DI bool lt(float a, float b) { return a < b; }
    static DI bool both(bool a, bool b) { return bool(int(a) & int(b)); }
    static DI float sel(bool c, float a, float b) { return c ? a : b; }
    static DI float hsum(float a) { return a; }                       // sum over the lane's virtual contacts, in their order
    static DI float nsum(float a) { return -a; }                      // minus that sum, as the scalar code accumulates it: (-a0) - a1
};
template <> struct LaneVec<v2f> {
    typedef b2 mask;
    static DI v2f splat(float a) { return (v2f)(a); }
    static DI v2f fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
    static DI v2f rcp(v2f a) { v2f r; r.x = rcp_(a.x); r.y = rcp_(a.y); return r; }
    static DI v2f rsq(v2f a) { v2f r; r.x = rsq_(a.x); r.y = rsq_(a.y); return r; }
    static DI v2f max(v2f a, v2f b) { v2f r; r.x = fmaxf(a.x, b.x); r.y = fmaxf(a.y, b.y); return r; }
    static DI v2f min(v2f a, v2f b) { v2f r; r.x = fminf(a.x, b.x); r.y = fminf(a.y, b.y); return r; }
    static DI b2 gt(v2f a, v2f b) { return b2{a.x > b.x, a.y > b.y}; }
    static DI b2 lt(v2f a, v2f b) { return b2{a.x < b.x, a.y < b.y}; }
    static DI b2 both(b2 a, b2 b) { return b2{bool(int(a.x) & int(b.x)), bool(int(a.y) & int(b.y))}; }
    static DI v2f sel(b2 c, v2f a, v2f b) { v2f r; r.x = c.x ? a.x : b.x; r.y = c.y ? a.y : b.y; return r; }
    static DI float hsum(v2f a) { return a.x + a.y; }
    static DI float nsum(v2f a) { return -a.x - a.y; }
};

// One contact's block of the Jacobi iteration (oracle: cone_local_solve): from the force f, the residual r of its three rows and its block B (regulariser included),
// a better force h of the cone |h_t| <= mu h_n for the block's own problem.
//   (1) ray: exact line minimisation along the current force, f <- (1 + x) f, x >= -1;
//   (2) second ray, always, from the new point and its residual: along (1, 0, 0) or, when friction alone makes a force pay (r_n < mu |r_t|), along
//       (1, -mu r_t / |r_t|), x2 >= 0 -- a contact whose force the first ray has taken to zero starts again within the same visit, and the visit is a continuous
//       function of its inputs (no switch that float32 and float64 could take differently);
//   (3) friction with the normal fixed: the minimiser of the tangential 2 x 2 problem on the disc |t| <= mu n, t = -(B_tt + lambda I)^-1 r~ in adjugate form, one
//       Newton step on the secular equation from the contact's lambda of the iteration before, radial clamp.
// T = float: one contact per lane; T = v2f: the two contacts of a pair (same block, own residual, force, cone and multiplier) in the two halves.
// Returns whether the contact has a friction disc (lambda is meaningful).
template <class T>
DI typename LaneVec<T>::mask cone_local(const T b00, const T b01, const T b02, const T b11, const T b12, const T b22, T r0, T r1, T r2,
                                        const T f0, const T f1, const T f2, const T mu, T& lam, T& h0, T& h1, T& h2) {
    typedef LaneVec<T> V;
    const T Bf0 = V::fma(b02, f2, V::fma(b01, f1, b00 * f0)), Bf1 = V::fma(b12, f2, V::fma(b11, f1, b01 * f0)), Bf2 = V::fma(b22, f2, V::fma(b12, f1, b02 * f0));
    const T vr = V::fma(f2, r2, V::fma(f1, r1, f0 * r0)), vBv = V::fma(f2, Bf2, V::fma(f1, Bf1, f0 * Bf0));
    // (a force below 1e-10 N is left to the second ray: the damped steps of the iteration shrink a force that has to vanish geometrically, and once f0^2 underflows in
    //  float32 the quotient is inf -- oracle: same threshold)
    const T x = V::sel(V::gt(f0, V::splat(1e-10f)), V::max(-vr * V::rcp(vBv), V::splat(-1.f)), V::splat(0.f));
    r0 = V::fma(x, Bf0, r0); r1 = V::fma(x, Bf1, r1); r2 = V::fma(x, Bf2, r2);
    T n0 = V::fma(x, f0, f0), n1 = V::fma(x, f1, f1), n2 = V::fma(x, f2, f2);
    const T rt2 = V::fma(r1, r1, r2 * r2);
    const T irt = V::rsq(V::max(rt2, V::splat(1e-30f))), rtn = rt2 * irt;
    const T sl = V::sel(V::both(V::gt(rt2, V::splat(0.f)), V::lt(r0, mu * rtn)), -mu * irt, V::splat(0.f));
    const T u1 = sl * r1, u2 = sl * r2;                                            // second direction (1, u1, u2)
    const T Bu0 = V::fma(b02, u2, V::fma(b01, u1, b00)), Bu1 = V::fma(b12, u2, V::fma(b11, u1, b01)), Bu2 = V::fma(b22, u2, V::fma(b12, u1, b02));
    const T ur = V::fma(u2, r2, V::fma(u1, r1, r0)), uBu = V::fma(u2, Bu2, V::fma(u1, Bu1, Bu0));
    const T x2 = V::max(-ur * V::rcp(uBu), V::splat(0.f));
    n0 += x2; n1 = V::fma(x2, u1, n1); n2 = V::fma(x2, u2, n2);
    r1 = V::fma(x2, Bu1, r1); r2 = V::fma(x2, Bu2, r2);
    // friction on the disc |t| <= mu n0
    const T lim = mu * n0;
    // (a friction disc below 1e-7 N is no friction: the multiplier of such a disc is ~ |r~| / lim, and beyond 1e10 the squares below leave float32 -- oracle: same threshold)
    const typename V::mask haslim = V::gt(lim, V::splat(1e-7f));
    const T q1 = r1 - V::fma(b12, n2, b11 * n1), q2 = r2 - V::fma(b22, n2, b12 * n1);
    T m11 = b11 + lam, m22 = b22 + lam, det = V::fma(m11, m22, -(b12 * b12));
    T a1 = V::fma(m22, q1, -(b12 * q2)), a2 = V::fma(m11, q2, -(b12 * q1));
    {
        const T aa = V::fma(a1, a1, a2 * a2);
        const T aAa = V::fma(m11 * a2, a2, V::fma(m22 * a1, a1, -2.f * b12 * a1 * a2));
        const T an = aa * V::rsq(V::max(aa, V::splat(1e-30f)));
        lam = V::max(V::fma(V::fma(-det, lim, an) * aa, V::rcp(V::max(lim * aAa, V::splat(1e-30f))), lam), V::splat(0.f));
    }
    m11 = b11 + lam; m22 = b22 + lam; det = V::fma(m11, m22, -(b12 * b12));
    a1 = V::fma(m22, q1, -(b12 * q2)); a2 = V::fma(m11, q2, -(b12 * q1));
    const T aa = V::fma(a1, a1, a2 * a2);
    const T sc = -V::min(lim * V::rsq(aa), V::rcp(det));                           // (v_min keeps the number when aa = 0 makes the product inf or NaN)
    h0 = n0; h1 = V::sel(haslim, a1 * sc, V::splat(0.f)); h2 = V::sel(haslim, a2 * sc, V::splat(0.f));
    return haslim;
}

// Contact solve of one forward pass (called when some environment of the wave has a contact): contact k of an environment lives in the
// registers of lane k of its group.  Inputs: contact records in LDS (lattice_front), element indices cel[], the site-space operator
// Lambda^-1 (packed lower 6 x 6), the site acceleration / velocity of the unconstrained arm (alpha, vs).  Outputs: net contact wrench on the
// site W[6] (accumulated) and the impulse gf[k] along each contact's element axis.
// The arm-independent half of a contact lane's set-up (everything the row needs from the record, the lattice tables and the element state): the split
// kernel's lattice side runs it before hand-off (2), while the arm side still forms Lambda^-1.
struct ContactRows {
    float w[3][6], g[3], Rd[3], Km[MAXC], vrel0[3], ae0, kdist;
};
template <int G>
DI void contact_rows(float* lds, const int eb, const int gl, const DevModel& M, const DevCfg& C, const int nc, const int* cel, const float vz, ContactRows& P) {
#define EB(off) lds[TB_WORDS + eb * GE_STRIDE + (off)]
    const bool own = gl < nc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) P.Km[c] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        P.g[d] = 0.f; P.Rd[d] = 0.f; P.vrel0[d] = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) P.w[d][a] = 0.f;
    }
    P.ae0 = 0.f; P.kdist = 0.f;
    if (own) {
        const int b = GE_CG + gl * CG_WORDS;
        f3 nn = mk(EB(b + 0), EB(b + 1), EB(b + 2)), rr = mk(EB(b + 3), EB(b + 4), EB(b + 5));
        const int e = __float_as_int(EB(b + 6));
        const float dist = EB(b + 7);
        // tangent frame without a case distinction (Frisvad 2012: continuous except at n.z = -1; contact normals point from the element towards the probe,
        // and no element sits above it): the iterate of a fixed number of row-by-row sweeps depends on the frame, so float32 and float64 must not be able
        // to choose different ones (oracle: same lines)
        const float aa = -rcp_(1.f + nn.z), bb = nn.x * nn.y * aa;
        f3 t1 = mk(1.f + nn.x * nn.x * aa, bb, -nn.x);
        f3 t2 = mk(bb, 1.f + nn.y * nn.y * aa, -nn.y);
        f3 ax = mk(lds[TB_AXIS + 3 * e], lds[TB_AXIS + 3 * e + 1], lds[TB_AXIS + 3 * e + 2]);
        const float sde = EB(GE_SD + e);
        float xx = fminf(-dist * (1.0f / SI_WIDTH), 1.f);
        float yy = (xx < 0.5f) ? 2.f * xx * xx : 1.f - 2.f * (1.f - xx) * (1.f - xx);
        float dimp = SI_D0 + yy * (SI_DMAX - SI_D0);
        float kk = dimp * (1.0f / (SI_DMAX * SI_DMAX * SR_TC * SR_TC));
        float Rn = (1.f - dimp) * rcp_(dimp) * M.invw;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { const float t = lds[TB_LINV + e * LROW + cel[c]] * (1.0f / ELEM_MASS); P.Km[c] = (c < nc) ? t : 0.f; }
        P.ae0 = EB(GE_A + e);
        P.kdist = kk * dist;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            f3 dir = (d == 0) ? nn : (d == 1 ? t1 : t2);
            f3 rx = cross(rr, dir);
            P.w[d][0] = dir.x; P.w[d][1] = dir.y; P.w[d][2] = dir.z; P.w[d][3] = rx.x; P.w[d][4] = rx.y; P.w[d][5] = rx.z;
            P.g[d] = -dot(dir, ax);
            P.vrel0[d] = P.g[d] * sde - dir.z * vz;
            P.Rd[d] = (d == 0) ? Rn * C.rn_scale : Rn * (1.0f / IMPRATIO);    // (two colliding probe geoms: two equal normal rows in parallel = half the regulariser)
        }
    }
#undef EB
}

// PRE: the arm-independent half of the rows comes from contact_rows (P); otherwise the whole set-up is formed here (vz and the records; P is not read) -- one
// statement sequence for the kernels that do not split it: handing the rows through the struct cost the 16-lane split kernel 0.5 us per step.
template <int G, bool PRE>
DI void contact_solve(float* lds, const int eb, const int gl, const DevModel& M, const DevCfg& C, const int nc, const int ncmax, const int* cel,
                      const float* Li, const float* alpha, const float* vs, const float mu, const float vz, const ContactRows& P, float* W, float* gf,
                      unsigned long long* dbg) {
#define EB(off) lds[TB_WORDS + eb * GE_STRIDE + (off)]
    // ---- contact k lives in the registers of lane k of its group.  Set-up: row directions w, Lambda^-1 w, element
    //      coupling g, reference acceleration, regulariser; Km[c] = Linv[e_own][e_c] / m ----
    USIM_CSTAMP(dbg, 0);
    // 16 lanes per environment, eight contact slots: lanes 8-15 would idle through the set-up.  They CLONE lanes 0-7 instead (lane 8 + k forms the same rows
    // of contact k, at no cost: same instructions) and take half of the Delassus blocks off them below.  Everything that leaves this function is masked to
    // lanes 0-7 or to lane k: forces stay zero in the clones (only lane k keeps an increment), the wrench sum and the publications read lanes < MAXC.
    constexpr bool CLONE = (G == 16) && !PRE;
    const int cl = CLONE ? (gl & 7) : gl;
    const bool own = cl < nc;
    float w[3][6], Liw[3][6], g[3], Rd[3], f[3] = {0.f, 0.f, 0.f}, cres[3] = {0.f, 0.f, 0.f}, Km[MAXC];
    if constexpr (PRE) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) Km[c] = P.Km[c];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        g[d] = P.g[d]; Rd[d] = P.Rd[d];
#pragma unroll
        for (int a = 0; a < 6; ++a) { w[d][a] = P.w[d][a]; Liw[d][a] = 0.f; }
    }
    if (own) {
        const float bcon = 2.0f / (SI_DMAX * SR_TC);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float vrel = P.vrel0[d], wa = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float s = 0.f;
#pragma unroll
                for (int bb = 0; bb < 6; ++bb) s = fmaf((a >= bb) ? Li[PK(a, bb)] : Li[PK(bb, a)], w[d][bb], s);
                Liw[d][a] = s;
                vrel = fmaf(w[d][a], vs[a], vrel);
                wa = fmaf(w[d][a], alpha[a], wa);
            }
            const float aref = -bcon * vrel - (d == 0 ? P.kdist : 0.f);
            cres[d] = fmaf(g[d], P.ae0, wa) - aref;           // residual of row d at zero force
        }
    }
    } else {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) Km[c] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        g[d] = 0.f; Rd[d] = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) { w[d][a] = 0.f; Liw[d][a] = 0.f; }
    }
    if (own) {
        const int b = GE_CG + cl * CG_WORDS;
        f3 nn = mk(EB(b + 0), EB(b + 1), EB(b + 2)), rr = mk(EB(b + 3), EB(b + 4), EB(b + 5));
        const int e = __float_as_int(EB(b + 6));
        const float dist = EB(b + 7);
        // tangent frame without a case distinction (Frisvad 2012: continuous except at n.z = -1; contact normals point from the element towards the probe,
        // and no element sits above it): the iterate of a fixed number of row-by-row sweeps depends on the frame, so float32 and float64 must not be able
        // to choose different ones (oracle: same lines)
        const float aa = -rcp_(1.f + nn.z), bb = nn.x * nn.y * aa;
        f3 t1 = mk(1.f + nn.x * nn.x * aa, bb, -nn.x);
        f3 t2 = mk(bb, 1.f + nn.y * nn.y * aa, -nn.y);
        f3 ax = mk(lds[TB_AXIS + 3 * e], lds[TB_AXIS + 3 * e + 1], lds[TB_AXIS + 3 * e + 2]);
        const float sde = EB(GE_SD + e);
        const float bcon = 2.0f / (SI_DMAX * SR_TC);
        float xx = fminf(-dist * (1.0f / SI_WIDTH), 1.f);
        float yy = (xx < 0.5f) ? 2.f * xx * xx : 1.f - 2.f * (1.f - xx) * (1.f - xx);
        float dimp = SI_D0 + yy * (SI_DMAX - SI_D0);
        float kk = dimp * (1.0f / (SI_DMAX * SI_DMAX * SR_TC * SR_TC));
        float Rn = (1.f - dimp) * rcp_(dimp) * M.invw;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { const float t = lds[TB_LINV + e * LROW + cel[c]] * (1.0f / ELEM_MASS); Km[c] = (c < nc) ? t : 0.f; }   // (cel[c] = 0 beyond the count: the read is always in range, and a select costs less than a branch around it)
        const float ae0 = EB(GE_A + e);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            f3 dir = (d == 0) ? nn : (d == 1 ? t1 : t2);
            f3 rx = cross(rr, dir);
            w[d][0] = dir.x; w[d][1] = dir.y; w[d][2] = dir.z; w[d][3] = rx.x; w[d][4] = rx.y; w[d][5] = rx.z;
            g[d] = -dot(dir, ax);
            float vrel = g[d] * sde - dir.z * vz, wa = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float s = 0.f;
#pragma unroll
                for (int bb = 0; bb < 6; ++bb) s = fmaf((a >= bb) ? Li[PK(a, bb)] : Li[PK(bb, a)], w[d][bb], s);
                Liw[d][a] = s;
                vrel = fmaf(w[d][a], vs[a], vrel);
                wa = fmaf(w[d][a], alpha[a], wa);
            }
            const float aref = -bcon * vrel - (d == 0 ? kk * dist : 0.f);
            Rd[d] = (d == 0) ? Rn * C.rn_scale : Rn * (1.0f / IMPRATIO);      // (two colliding probe geoms: two equal normal rows in parallel = half the regulariser)
            cres[d] = fmaf(g[d], ae0, wa) - aref;             // residual of row d at zero force
        }
    }
    }
    USIM_STAMP(dbg, 9);
    USIM_CSTAMP(dbg, 1);
    // ---- Delassus blocks: B[k][d][d'] = d(residual of row d of this lane's contact) / d(force on row d' of contact k)
    //      = w_d . Lambda^-1 w^k_d' + g_d Km[k] g^k_d' (+ the regulariser on the diagonal of the lane's own block); the sweeps below then
    //      need three broadcasts per visit.
    //      Lane k publishes Lambda^-1 w^k (18 words) and g^k (3) once in the environment's LDS block (the right-hand-side / staging area is
    //      free by now); every lane then reads contact k's record with six 16-byte broadcast reads -- a quarter of the issue slots the 21 DPP
    //      broadcasts took --, the reads of contact k + 1 in flight while the block of contact k is formed.
    const int ncr = ncmax;                                        // (the iterations run exactly the wave's largest contact count)
    // B[k] = d(residual of this lane's rows) / d(force on contact k), WITHOUT the regulariser; written and read only under k < ncr.  16-lane groups keep FOUR blocks per
    // lane (round 6): those of contacts 0-3 in both halves while the wave has at most four contacts, those of contacts 0-3 in lanes 0-7 and of contacts 4-7 in lanes 8-15
    // beyond -- each half then sums its own four products of an iteration and one rotation adds the halves (below).
    constexpr int NB = CLONE ? 4 : MAXC;
    float B[NB][3][3];
    float b00 = 0.f, b01 = 0.f, b02 = 0.f, b11 = 0.f, b12 = 0.f, b22 = 0.f;      // the lane's own diagonal block (regulariser added below)
    static_assert(MAXC * 24 <= GE_SD, "Delassus records overlay the rhs / staging area");
    if (gl < MAXC) {
        float4* pub = reinterpret_cast<float4*>(&EB(gl * 24));
        pub[0] = make_float4(Liw[0][0], Liw[0][1], Liw[0][2], Liw[0][3]); pub[1] = make_float4(Liw[0][4], Liw[0][5], Liw[1][0], Liw[1][1]);
        pub[2] = make_float4(Liw[1][2], Liw[1][3], Liw[1][4], Liw[1][5]); pub[3] = make_float4(Liw[2][0], Liw[2][1], Liw[2][2], Liw[2][3]);
        pub[4] = make_float4(Liw[2][4], Liw[2][5], g[0], g[1]); pub[5] = make_float4(g[2], 0.f, 0.f, 0.f);
    }
    group_sync();
    if constexpr (CLONE) {
        // Lane k carries contact A of pair k, lane 8 + k contact B: the same rows.  Up to four contacts in the wave: both halves form blocks 0-3 (the same instructions:
        // no cost).  More: lanes 0-7 form the blocks of contacts 0-3, lanes 8-15 those of contacts 4-7 -- and keep them (rounds 3-5 rotated every block to the other half:
        // 36 rotations, 72 selects and 36 more registers per lane; the iteration below no longer needs them).  A slot beyond an environment's count holds zeros (its lane
        // published zero rows), so its block is exactly zero.
        const bool hi = gl >= 8;
        const bool wide = ncr > 4;                                       // (wave-uniform)
        const int k0 = (hi && wide) ? 4 : 0;
        float4 rk[6];
        {
            const float4* src = reinterpret_cast<const float4*>(&EB(k0 * 24));
#pragma unroll
            for (int v = 0; v < 6; ++v) rk[v] = src[v];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < ncr) {
                const float Kmj = (hi && wide) ? Km[4 + j] : Km[j];
                const float Lk[3][6] = {{rk[0].x, rk[0].y, rk[0].z, rk[0].w, rk[1].x, rk[1].y}, {rk[1].z, rk[1].w, rk[2].x, rk[2].y, rk[2].z, rk[2].w},
                                        {rk[3].x, rk[3].y, rk[3].z, rk[3].w, rk[4].x, rk[4].y}};
                const float gk[3] = {rk[4].z * Kmj, rk[4].w * Kmj, rk[5].x * Kmj};
                if (j + 1 < 4) {
                    const float4* src = reinterpret_cast<const float4*>(&EB((k0 + j + 1) * 24));
#pragma unroll
                    for (int v = 0; v < 6; ++v) rk[v] = src[v];
                }
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float r1 = fmaf(w[d][4], Lk[dd][4], fmaf(w[d][2], Lk[dd][2], w[d][0] * Lk[dd][0]));
                        float r2 = fmaf(w[d][5], Lk[dd][5], fmaf(w[d][3], Lk[dd][3], w[d][1] * Lk[dd][1]));
                        B[j][d][dd] = fmaf(g[d], gk[dd], r1 + r2);
                    }
                }
                // the lane's own diagonal block, if this half formed it (cl - k0 == j); otherwise the partner lane (the other half, same cl) did: one rotation below
                const bool me = (cl - k0) == j;
                b00 = me ? B[j][0][0] : b00; b01 = me ? B[j][0][1] : b01; b02 = me ? B[j][0][2] : b02;
                b11 = me ? B[j][1][1] : b11; b12 = me ? B[j][1][2] : b12; b22 = me ? B[j][2][2] : b22;
            }
        }
        if (wide) {
            // (exactly one lane of a pair holds the block, the other holds zeros: x + 0)
            b00 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b00), 0x128, 0xf, 0xf, true)); b01 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b01), 0x128, 0xf, 0xf, true));
            b02 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b02), 0x128, 0xf, 0xf, true)); b11 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b11), 0x128, 0xf, 0xf, true));
            b12 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b12), 0x128, 0xf, 0xf, true)); b22 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b22), 0x128, 0xf, 0xf, true));
        }
    } else {
    float4 rk[6];
    {
        const float4* src = reinterpret_cast<const float4*>(&EB(0));
#pragma unroll
        for (int v = 0; v < 6; ++v) rk[v] = src[v];
    }
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        if (k < ncr) {
            const float Lk[3][6] = {{rk[0].x, rk[0].y, rk[0].z, rk[0].w, rk[1].x, rk[1].y}, {rk[1].z, rk[1].w, rk[2].x, rk[2].y, rk[2].z, rk[2].w},
                                    {rk[3].x, rk[3].y, rk[3].z, rk[3].w, rk[4].x, rk[4].y}};
            const float gk[3] = {rk[4].z * Km[k], rk[4].w * Km[k], rk[5].x * Km[k]};
            if (k + 1 < MAXC) {                                          // next contact's record (a slot beyond the count holds stale words: unused)
                const float4* src = reinterpret_cast<const float4*>(&EB((k + 1) * 24));
#pragma unroll
                for (int v = 0; v < 6; ++v) rk[v] = src[v];
            }
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float r1 = fmaf(w[d][4], Lk[dd][4], fmaf(w[d][2], Lk[dd][2], w[d][0] * Lk[dd][0]));
                    float r2 = fmaf(w[d][5], Lk[dd][5], fmaf(w[d][3], Lk[dd][3], w[d][1] * Lk[dd][1]));
                    B[k][d][dd] = fmaf(g[d], gk[dd], r1 + r2);
                }
            }
        }
    }
    }
    USIM_CSTAMP(dbg, 2);
    // ---- BLOCK JACOBI WITH AN EXACT LINE SEARCH on the dual  min 1/2 f'(A + R) f + b'f,  f_v in {|f_t| <= mu_v f_n}  (round 5; oracle: constrained_forward,
    //      cone_solver 2 -- same optimum as the exact-cone Gauss-Seidel of round 4, i.e. MuJoCo's Newton solver's).  A Gauss-Seidel sweep is one visit per contact, one
    //      after the other, in which one lane of sixteen does useful work -- and two visits per probe-element pair once its two coincident contacts
    //      (ultrasound_probe_gripper.xml:8-9: probe_collision, mu 0.01 after MuJoCo's max rule, and probe_visual, mu 1) are modelled explicitly.  A wave issues one
    //      instruction per ~4 cycles whatever its lanes do (profiles/r05/micro_two_wave.txt), so here EVERY virtual contact runs its visit at the same time: contact A of
    //      pair k in lane k, contact B in lane 8 + k (16-lane groups; both in lane k with 8-lane groups), each on its own 3 x 3 block from the current residual
    //      (cone_local: ray update with an immediate restart from zero, friction QCQP with one Newton step on the carried multiplier).  The step along d = f^ - f is
    //      t = (sum_v d_v'B_v d_v) / (d'Qd) <= 1 (the exact minimiser of the quadratic when every block is solved exactly, never longer) -- a convex combination of feasible points, no projection --; the shared residual moves by t A D,
    //      D_k = d_Ak + d_Bk (three row broadcasts and nine multiply-adds per pair: the only part that grows with the contact count).  pgs_iters iterations, cold start.
    // Virtual contacts per lane: one (16-lane groups: contact A of pair k in lane k, contact B in lane 8 + k) or two (8-lane groups: both in lane k, in the two halves
    // of packed registers -- one visit for the pair at the issue cost of ~1.2 instead of two: round 6).
    typedef typename std::conditional<CLONE, float, v2f>::type VT;
    typedef LaneVec<VT> V;
    const bool pairB = C.pair != 0;
    const float muB = fmaxf(C.probe_fric2, C.elem_fric);
    typename V::mask ownv; VT muv, fv[3], lamv = V::splat(0.f);
    if constexpr (CLONE) { ownv = own && (gl < 8 || pairB); muv = (gl < 8) ? mu : muB; }
    else { ownv = b2{own, own && pairB}; muv.x = mu; muv.y = muB; }
    fv[0] = fv[1] = fv[2] = V::splat(0.f);
    // the lane's own diagonal block, regulariser included
    if constexpr (!CLONE) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            if (k < ncr) {
                const bool me = cl == k;
                b00 = me ? B[k][0][0] : b00; b01 = me ? B[k][0][1] : b01; b02 = me ? B[k][0][2] : b02;
                b11 = me ? B[k][1][1] : b11; b12 = me ? B[k][1][2] : b12; b22 = me ? B[k][2][2] : b22;
            }
        }
    }
    b00 += Rd[0]; b11 += Rd[1]; b22 += Rd[2];
    auto iterations = [&](auto NCM_) {
        constexpr int NCM = decltype(NCM_)::value;
        const VT vb00 = V::splat(b00), vb01 = V::splat(b01), vb02 = V::splat(b02), vb11 = V::splat(b11), vb12 = V::splat(b12), vb22 = V::splat(b22);
        const VT vR0 = V::splat(Rd[0]), vR1 = V::splat(Rd[1]), vR2 = V::splat(Rd[2]);
        for (int it = 0; it < C.pgs_iters; ++it) {
            VT dv[3];
            float num, D0, D1, D2;
            {
                const VT r0 = V::fma(vR0, fv[0], V::splat(cres[0])), r1 = V::fma(vR1, fv[1], V::splat(cres[1])), r2 = V::fma(vR2, fv[2], V::splat(cres[2]));
                VT h0, h1, h2, lam = lamv;
                const typename V::mask haslim = cone_local<VT>(vb00, vb01, vb02, vb11, vb12, vb22, r0, r1, r2, fv[0], fv[1], fv[2], muv, lam, h0, h1, h2);
                lamv = V::sel(V::both(ownv, haslim), lam, lamv);
                dv[0] = V::sel(ownv, h0 - fv[0], V::splat(0.f)); dv[1] = V::sel(ownv, h1 - fv[1], V::splat(0.f)); dv[2] = V::sel(ownv, h2 - fv[2], V::splat(0.f));
                // slope of the cost along d, block by block: -d'B d (for the minimiser of a block r.d <= -d'B d, with equality inside the cone) -- a sum of squares
                // instead of r.d, whose products cancel to second order for a sliding contact and are float32 noise once |d| < 5e-3 N (oracle: same lines)
                const VT e0 = dv[0], e1 = dv[1], e2 = dv[2];
                const VT Be0 = V::fma(vb02, e2, V::fma(vb01, e1, vb00 * e0)), Be1 = V::fma(vb12, e2, V::fma(vb11, e1, vb01 * e0)), Be2 = V::fma(vb22, e2, V::fma(vb12, e1, vb02 * e0));
                num = V::nsum(V::fma(e2, Be2, V::fma(e1, Be1, e0 * Be0)));          // (no 0 + x: the compiler may not fold it -- signed zeros -- and an instruction here is paid 24 times per step)
                D0 = V::hsum(dv[0]); D1 = V::hsum(dv[1]); D2 = V::hsum(dv[2]);
            }
            if constexpr (CLONE) {                                       // D_k = d_Ak + d_Bk in both halves
                D0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D0), 0x128, 0xf, 0xf, true));
                D1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D1), 0x128, 0xf, 0xf, true));
                D2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D2), 0x128, 0xf, 0xf, true));
            }
            // q = A D.  Up to four contacts: one running sum over them.  More: the sum over contacts 0-3 plus the sum over contacts 4-7 -- in that association in EVERY mapping
            // (an environment's bits must not depend on its wave's neighbours: for one with at most four contacts the second sum is exact zeros) -- which a 16-lane group
            // evaluates in its two halves at once: lanes 0-7 take D_j, lanes 8-15 D_(4+j) from the same row (two bank-masked broadcasts per word), each half multiplies with
            // the four blocks it formed, one rotation by eight lanes adds the halves.  Eight contacts: 63 instructions instead of 96 (round 5), six: 63 / 72.
            float q0 = 0.f, q1 = 0.f, q2 = 0.f;
            if constexpr (NCM <= 4) {
#pragma unroll
                for (int k = 0; k < NCM; ++k) {
                    const float e0 = group_bcast<G>(D0, k), e1 = group_bcast<G>(D1, k), e2 = group_bcast<G>(D2, k);
                    q0 = fmaf(B[k][0][2], e2, fmaf(B[k][0][1], e1, fmaf(B[k][0][0], e0, q0)));
                    q1 = fmaf(B[k][1][2], e2, fmaf(B[k][1][1], e1, fmaf(B[k][1][0], e0, q1)));
                    q2 = fmaf(B[k][2][2], e2, fmaf(B[k][2][1], e1, fmaf(B[k][2][0], e0, q2)));
                }
            } else if constexpr (CLONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = half_bcast(D0, j), e1 = half_bcast(D1, j), e2 = half_bcast(D2, j);
                    q0 = fmaf(B[j][0][2], e2, fmaf(B[j][0][1], e1, fmaf(B[j][0][0], e0, q0)));
                    q1 = fmaf(B[j][1][2], e2, fmaf(B[j][1][1], e1, fmaf(B[j][1][0], e0, q1)));
                    q2 = fmaf(B[j][2][2], e2, fmaf(B[j][2][1], e1, fmaf(B[j][2][0], e0, q2)));
                }
                q0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q0), 0x128, 0xf, 0xf, true));
                q1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q1), 0x128, 0xf, 0xf, true));
                q2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q2), 0x128, 0xf, 0xf, true));
            } else {
                float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
                for (int k = 0; k < NCM; ++k) {
                    const float e0 = group_bcast<G>(D0, k), e1 = group_bcast<G>(D1, k), e2 = group_bcast<G>(D2, k);
                    float& a0 = (k < 4) ? q0 : h0; float& a1 = (k < 4) ? q1 : h1; float& a2 = (k < 4) ? q2 : h2;
                    a0 = fmaf(B[k][0][2], e2, fmaf(B[k][0][1], e1, fmaf(B[k][0][0], e0, a0)));
                    a1 = fmaf(B[k][1][2], e2, fmaf(B[k][1][1], e1, fmaf(B[k][1][0], e0, a1)));
                    a2 = fmaf(B[k][2][2], e2, fmaf(B[k][2][1], e1, fmaf(B[k][2][0], e0, a2)));
                }
                q0 += h0; q1 += h1; q2 += h2;
            }
            float den = V::hsum(V::fma(dv[2], V::fma(vR2, dv[2], V::splat(q2)), V::fma(dv[1], V::fma(vR1, dv[1], V::splat(q1)), dv[0] * V::fma(vR0, dv[0], V::splat(q0)))));
            num = group_allsum<G>(num); den = group_allsum<G>(den);
            const float t = (den > 0.f) ? fminf(-num * rcp_(den), 1.f) : 0.f;
            fv[0] = V::fma(V::splat(t), dv[0], fv[0]); fv[1] = V::fma(V::splat(t), dv[1], fv[1]); fv[2] = V::fma(V::splat(t), dv[2], fv[2]);
            cres[0] = fmaf(t, q0, cres[0]); cres[1] = fmaf(t, q1, cres[1]); cres[2] = fmaf(t, q2, cres[2]);
        }
    };
    // (one straight-line instantiation per wave-uniform contact count)
    switch (ncr) {
        case 1: iterations(std::integral_constant<int, 1>{}); break;
        case 2: iterations(std::integral_constant<int, 2>{}); break;
        case 3: iterations(std::integral_constant<int, 3>{}); break;
        case 4: iterations(std::integral_constant<int, 4>{}); break;
        case 5: iterations(std::integral_constant<int, 5>{}); break;
        case 6: iterations(std::integral_constant<int, 6>{}); break;
        case 7: iterations(std::integral_constant<int, 7>{}); break;
        default: iterations(std::integral_constant<int, 8>{}); break;
    }
    // the pair's total force (lanes 0-7 of a 16-lane group: contact A's own force plus contact B's from lane 8 + k)
    if constexpr (CLONE) {
#pragma unroll
        for (int d = 0; d < 3; ++d) f[d] = fv[d] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fv[d]), 0x128, 0xf, 0xf, true));
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) f[d] = V::hsum(fv[d]);
    }
    USIM_STAMP(dbg, 10);
    USIM_CSTAMP(dbg, 3);
    // ---- contact wrench on the site and impulse along each element axis (lanes without a contact hold w = g = f = 0, i.e. contribute
    //      zeros) ----
    {
        const float Fw[6] = {w[0][0] * f[0] + w[1][0] * f[1] + w[2][0] * f[2], w[0][1] * f[0] + w[1][1] * f[1] + w[2][1] * f[2],
                             w[0][2] * f[0] + w[1][2] * f[1] + w[2][2] * f[2], w[0][3] * f[0] + w[1][3] * f[1] + w[2][3] * f[2],
                             w[0][4] * f[0] + w[1][4] * f[1] + w[2][4] * f[2], w[0][5] * f[0] + w[1][5] * f[1] + w[2][5] * f[2]};
        const float gfo = (g[0] * f[0] + g[1] * f[1] + g[2] * f[2]) * (1.0f / ELEM_MASS);
        {
            // the sum over the eight contact lanes is a shifted-add reduction over the DPP row (16 lanes per environment: lanes 8-15 are masked;
            // 8 lanes per environment: the shifts are fenced at the group boundary, same tree, same bits), the per-contact impulses are
            // group broadcasts
            auto shr_add = [&](float v, auto Dc) {
                constexpr int D = decltype(Dc)::value;
                float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true));      // row_shr:D
                if constexpr (G == 8) t = (gl < D) ? 0.f : t;
                return v + t;
            };
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float v = (gl < MAXC) ? Fw[a] : 0.f;
                v = shr_add(v, std::integral_constant<int, 1>{});
                v = shr_add(v, std::integral_constant<int, 2>{});
                v = shr_add(v, std::integral_constant<int, 4>{});
                W[a] += group_bcast<G>(v, 7);
            }
#pragma unroll
            for (int k = 0; k < MAXC; ++k) gf[k] = group_bcast<G>(gfo, k);          // (no test per slot: a lane without a contact holds zeros)
        }
    }
    USIM_CSTAMP(dbg, 4);
#undef EB
}

#include "usim_full.h"

struct StepOut {               // results of one forward pass that the env logic needs
    float fc[3];               // net contact force on the probe (cfrc_ext[probe][3:6])
    float tq[3];               // torque sensor at ft_frame (site frame)
    int ncon;
    int con_shell[MAXC];
    int overflow;
};

// MODE 0: one env.step() per environment; a finished environment takes its next initial state from the reset bank.
// MODE 1: reset computation (draws, initial-pose IK, zero-torque forward pass) for the environments selected by the mask
//         (written to the live state) or for the (env, episode) items of the refill work list (written to the reset bank).
template <int TORSO, int G, int MODE>
__global__ __launch_bounds__(GroupGeom<G>::NT, (TORSO == 2) ? 2 : 1) void usim_step_kernel(const DevModel M, const DevCfg C, float* __restrict__ st, int n, int npad,
                                                                      const DevIO io, int flags, long long rstep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // step waves outrank the background refill waves that may share their SIMD (priority, then age, arbitrates VALU issue)
    __builtin_amdgcn_s_setprio(MODE == 0 ? 3 : 0);
    constexpr int EPW = GroupGeom<G>::EPW, EPB = GroupGeom<G>::EPB, NT = GroupGeom<G>::NT;
    constexpr int NTT = NT;
    static_assert(!TORSO || G >= MAXC, "the contact solver gives every contact its own lane of the group");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, ge = lane / G;           // lane within the group, group (= environment) within the wave
    const int gbase = lane - gl;                      // ballot bit of the group's first lane
    const int eb = wave * EPW + ge;                   // environment within the workgroup
    // refill launches walk the work list with a grid-stride loop; every other launch runs the body once
    const bool refill = (MODE == 1) && io.refill != 0;
    const int item_cnt = refill ? io.count[0] : 1;
    for (int item0 = refill ? (int)blockIdx.x * EPB : 0; item0 < item_cnt; item0 += refill ? (int)gridDim.x * EPB : 1) {
    int env = blockIdx.x * EPB + eb;
    bool valid = env < n;                             // lattice rows are stored by every lane of the group
    int item_ep = 0;
    if (refill) {
        valid = item0 + eb < item_cnt;
        const int2 it = valid ? io.items[item0 + eb] : make_int2(0, 0);
        env = it.x; item_ep = it.y;
    }
    const bool store = valid && gl == 0;              // per-environment scalars and outputs by its first lane
    const int ei = valid ? env : (refill ? 0 : n - 1);               // clamp so that every lane has something to read; stores are guarded
    constexpr bool reset_only = (MODE == 1);
    const bool auto_reset = (flags & LF_AUTO_RESET) != 0;
#define ST(f) st[scalar_index((f), (size_t)ei)]
#define STI(f) (reinterpret_cast<int*>(st))[scalar_index((f), (size_t)ei)]
#define LAT(w) st[(size_t)F_LAT * npad + (size_t)ei * (TORSO == 2 ? LATF_ENV_WORDS : LAT_ENV_WORDS) + (w)]
#define EB(off) lds[TB_WORDS + eb * GE_STRIDE + (off)]
// phase timeline probe (diagnostics only): wave 0 of workgroup 0 stamps the shader clock when io.dbg is set
// The stamps exist only in the profiling build (make prof -> libusim_prof.so, -DUSIM_TSTAMP): each one is a branch, and sixteen of them
// cost the production kernel 2 % (25.6 vs 25.1 us/step).  -DUSIM_TSTAMP_NOWAIT additionally keeps the stamps from draining memory traffic.
#if !defined(USIM_TSTAMP) && !defined(USIM_TSTAMP_NOWAIT)
#define TSTAMP(k) do { } while (0)
#elif defined(USIM_TSTAMP_NOWAIT)
#define TSTAMP(k) do { if (io.dbg && blockIdx.x == 0 && threadIdx.x == 0) { io.dbg[k] = __builtin_readcyclecounter(); } } while (0)
#else
#define TSTAMP(k) do { if (io.dbg && blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); io.dbg[k] = __builtin_readcyclecounter(); } } while (0)
#endif
#define BK(slot, f) st[(size_t)io.bank_row0 * npad + ((size_t)ei * BANK_DEPTH + (slot)) * BANK_STRIDE + (f)]
#define BKI(slot, f) (reinterpret_cast<int*>(st))[(size_t)io.bank_row0 * npad + ((size_t)ei * BANK_DEPTH + (slot)) * BANK_STRIDE + (f)]
    if (TORSO == 1 && item0 == (refill ? (int)blockIdx.x * EPB : 0)) {
        // workgroup-resident copy of the lattice tables (inverse 99 x 100, element positions/axes/neighbours/shell ids):
        // 16-byte loads, all issued before the first LDS store
        const float4* src = reinterpret_cast<const float4*>(M.tables);
        float4* dst = reinterpret_cast<float4*>(lds);
        constexpr int NV = TB_WORDS / 4, PER = (NV + NTT - 1) / NTT;
        float4 tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) { int idx = threadIdx.x + i * NTT; tmp[i] = (idx < NV) ? src[idx] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < PER; ++i) { int idx = threadIdx.x + i * NTT; if (idx < NV) dst[idx] = tmp[i]; }
        __syncthreads();
    }

    TSTAMP(0);
    // ---------------- load state ----------------
    float sv[F_NSCALAR];                               // the 40 scalar words of the environment, in Field order: ten 16-byte loads
    {
        const float4* sp = reinterpret_cast<const float4*>(st + scalar_index(0, (size_t)ei));
#pragma unroll
        for (int v = 0; v < F_NSCALAR / 4; ++v) { const float4 x = sp[v]; sv[4 * v] = x.x; sv[4 * v + 1] = x.y; sv[4 * v + 2] = x.z; sv[4 * v + 3] = x.w; }
    }
    float q[NJ], qd[NJ], q0[NJ], dq[NJ];             // the joint words of the state hold dq = q - q0 (usim_device.h)
#pragma unroll
    for (int i = 0; i < NJ; ++i) { dq[i] = sv[F_Q + i]; qd[i] = sv[F_QD + i]; q0[i] = sv[F_Q0 + i]; q[i] = q0[i] + dq[i]; }
    f3 ts = mk(sv[F_TS], sv[F_TS + 1], sv[F_TS + 2]), te = mk(sv[F_TE], sv[F_TE + 1], sv[F_TE + 2]);
    float u0 = sv[F_U0], vbar = sv[F_VBAR], fzbar = sv[F_FZBAR], fzprev = sv[F_FZPREV], dfz = sv[F_DFZ];
    float kst = sv[F_KST], kdmp = sv[F_KDMP], mu = sv[F_MU], epret = sv[F_EPRET];
    int t = __float_as_int(sv[F_T]), touched = __float_as_int(sv[F_TOUCH]), episode = __float_as_int(sv[F_EPISODE]), status = __float_as_int(sv[F_STATUS]);
    // lattice rows of this lane (elements gl, gl+G, ...): prefetched now, consumed after the arm phase
    constexpr int NE = (TORSO == 2) ? FE : (TORSO ? (N_TOP + G - 1) / G : 1);
    float s_pre[NE], sd_pre[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = (TORSO == 2) ? FE * gl + i : gl + i * G;          // (full torso: lane l owns elements 5 l .. 5 l + 4)
        s_pre[i] = 0.f; sd_pre[i] = 0.f;
        if (TORSO == 1 && MODE == 0 && e < N_TOP) { s_pre[i] = LAT(LAT_S + e); sd_pre[i] = LAT(LAT_SD + e); }
        if (TORSO == 2 && MODE == 0 && e < NSH) { s_pre[i] = LAT(LATF_S + e); sd_pre[i] = LAT(LATF_SD + e); }
    }
    // full torso: the free body (spawn pose at a reset: ultrasound.py:426-431)
    FullBody body;
    body.p = mk(M.torso[0], M.torso[1], M.torso[2]); body.q[0] = 1.f; body.q[1] = body.q[2] = body.q[3] = 0.f; body.v = mk(0.f, 0.f, 0.f); body.w = mk(0.f, 0.f, 0.f);
    if (TORSO == 2 && MODE == 0) {
        float bw[13];
#pragma unroll
        for (int a = 0; a < 13; ++a) bw[a] = LAT(LATF_BODY + a);
        body.p = mk(bw[0], bw[1], bw[2]); body.q[0] = bw[3]; body.q[1] = bw[4]; body.q[2] = bw[5]; body.q[3] = bw[6];
        body.v = mk(bw[7], bw[8], bw[9]); body.w = mk(bw[10], bw[11], bw[12]);
    }

    TSTAMP(1);
    // ---------------- action ----------------
    float act[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!reset_only) {
        if (flags & LF_RANDOM_ACT) {
            uint32_t gid = (uint32_t)(C.env_offset + ei);
            uint32_t rr[8];
            if constexpr (G == 16) {
                // the two counter blocks are evaluated side by side by the even and odd lanes of the group, then shared
                // (not for G = 8, where the extra live values push the kernel into scratch)
                u4 r = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 1u + (uint32_t)(gl & 1), C.key0, C.key1);
                const float ra = __uint_as_float(r.a), rb = __uint_as_float(r.b), rc = __uint_as_float(r.c), rd = __uint_as_float(r.d);
                rr[0] = __float_as_uint(group_bcast<G>(ra, 0)); rr[1] = __float_as_uint(group_bcast<G>(rb, 0));
                rr[2] = __float_as_uint(group_bcast<G>(rc, 0)); rr[3] = __float_as_uint(group_bcast<G>(rd, 0));
                rr[4] = __float_as_uint(group_bcast<G>(ra, 1)); rr[5] = __float_as_uint(group_bcast<G>(rb, 1));
                rr[6] = __float_as_uint(group_bcast<G>(rc, 1)); rr[7] = 0u;
            } else {
                u4 r1 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 1u, C.key0, C.key1);
                u4 r2 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 2u, C.key0, C.key1);
                rr[0] = r1.a; rr[1] = r1.b; rr[2] = r1.c; rr[3] = r1.d; rr[4] = r2.a; rr[5] = r2.b; rr[6] = r2.c; rr[7] = r2.d;
            }
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                float u = u01(rr[a]);
                bool sgn = (C.mode == 1) || (C.mode == 3) || (C.mode == 2 && a == 6);
                act[a] = sgn ? 2.f * u - 1.f : u;
                if (C.mode == 3) act[a] *= WRENCH_MAX;
                if (io.act_out && store && a < C.adim) io.act_out[(size_t)ei * C.adim + a] = act[a];
            }
        } else {
#pragma unroll
            for (int a = 0; a < 7; ++a) if (a < C.adim) {
                float v = io.act[(size_t)ei * C.adim + a];
                act[a] = (v == v && fabsf(v) <= 3.0e38f) ? v : 0.f;      // a non-finite action component is treated as 0
            }
        }
    }

    bool need = reset_only ? (io.mask ? io.mask[ei] != 0 : true) : false;   // lanes that (re)initialise in pass 1
    if (refill) need = valid;
    bool done = false;
    const float dt = C.dt, inv_h = rcp_((float)C.horizon);

    constexpr int pass = MODE;                        // 0: step from the live state, 1: reset computation
    int ep_t = episode;                               // episode index the reset draws are keyed on
    do {
        const bool active = (pass == 0) ? true : need;
        if (pass == 1) {
            if (!__any(need)) break;
            if (need) {
                // ================= reset draws (ultrasound.py:416-478) =================
                ep_t = refill ? item_ep : episode + 1;                   // listed bank episode, or the live reset
                if (!refill) episode = ep_t;
                uint32_t gid = (uint32_t)(C.env_offset + ei);
                u4 A = philox(gid, (uint32_t)ep_t, 0u, 0u, C.key0, C.key1);
                u4 B = philox(gid, (uint32_t)ep_t, 1u, 0u, C.key0, C.key1);
                u4 Cc = philox(gid, (uint32_t)ep_t, 2u, 0u, C.key0, C.key1);
                const float tz = M.torso[2] + M.base[2] + C.top_off;      // ultrasound.py:184,807
                f3 noise = mk(0, 0, 0);
                kst = C.stiffness; kdmp = C.damping;
                if (io.reset_params) {
                    const float* p = io.reset_params + (size_t)ei * 13;
                    ts = mk(p[0], p[1], p[2]); te = mk(p[3], p[4], p[5]); u0 = p[6]; noise = mk(p[7], p[8], p[9]);
                    kst = p[10]; kdmp = p[11]; mu = p[12];
                } else {
                    if (C.det_traj) { ts = mk(0.062f, -0.020f, 0.896f); te = mk(-0.032f, -0.075f, 0.896f); }   // ultrasound.py:763-764
                    else {
                        // ultrasound.py:787-788: np.linspace grids over the torso top, 50 points each
                        const float tx = M.torso[0] + M.base[0], ty = M.torso[1] + M.base[1];
                        const float xs = -0.15f + tx + 0.03f, xstep = (0.15f + tx - xs) / 49.f;
                        const float ys = -C.y_range + ty, ystep = 2.f * C.y_range / 49.f;
                        ts = mk(xs + (float)urange(A.a, 50u) * xstep, ys + (float)urange(A.b, 50u) * ystep, tz);
                        te = mk(xs + (float)urange(A.c, 50u) * xstep, ys + (float)urange(A.d, 50u) * ystep, tz);
                    }
                    u0 = u01(B.a);                                           // ultrasound.py:443
                    if (C.rand_pos) {                                        // ultrasound.py:880-881
                        float r1 = sqrtf(-2.f * logf(u01_open(B.b))), th1 = 2.f * PI_F * u01(B.c);
                        float r2 = sqrtf(-2.f * logf(u01_open(B.d))), th2 = 2.f * PI_F * u01(Cc.a);
                        noise = mk(r1 * cosf(th1) * 0.0025f, r1 * sinf(th1) * 0.0025f, r2 * cosf(th2) * 0.010f);
                    }
                    if (C.rand_solref) { kst = 1300.f + (float)urange(Cc.b, 300u); kdmp = 17.f + (float)urange(Cc.c, 24u); }   // ultrasound.py:293-294
                    float pf = C.probe_fric;
                    if (C.rand_fric) pf *= 0.5f + 1.5f * u01(Cc.d);
                    mu = fmaxf(pf, C.elem_fric);
            if (C.probe_geoms == 2 && !C.pair) mu = 0.5f * (mu + fmaxf(C.probe_fric2, C.elem_fric));   // two coincident contacts per pair restated as one (usim_config.probe_geoms)
                }
                // ================= initial pose: damped-least-squares IK from init_qpos (ultrasound.py:812-844) ==========
                float uu = clampf(u0, 0.f, 1.f);
                f3 tp0 = ts + (te - ts) * uu;
                f3 target = mk(tp0.x + noise.x + 0.0028f - M.base[0], tp0.y + noise.y + 0.0008f - M.base[1], tp0.z + noise.z + 0.0066f - M.base[2]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) q[i] = INITQ[i];
                for (int it = 0; it < C.ik_iters; ++it) {
                    Kin K; fk<G == 1>(M, q, K);
                    f3 gx = mk(M.grot[0], M.grot[3], M.grot[6]), gy = mk(M.grot[1], M.grot[4], M.grot[7]), gz = mk(M.grot[2], M.grot[5], M.grot[8]);
                    f3 eo = (cross(K.sx, gx) + cross(K.sy, gy) + cross(K.sz, gz)) * 0.5f;
                    f3 ep = target - K.x;
                    float e[6] = {ep.x, ep.y, ep.z, eo.x, eo.y, eo.z};
                    float J[6][NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        f3 jv = cross(K.z[j], K.x - K.o[j]);
                        J[0][j] = jv.x; J[1][j] = jv.y; J[2][j] = jv.z; J[3][j] = K.z[j].x; J[4][j] = K.z[j].y; J[5][j] = K.z[j].z;
                    }
                    float A6[21], id6[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) {
                            float s = (a == b) ? 1e-6f : 0.f;
#pragma unroll
                            for (int j = 0; j < NJ; ++j) s = fmaf(J[a][j], J[b][j], s);
                            A6[PK(a, b)] = s;
                        }
                    chol_packed<6>(A6, id6);
                    chol_solve<6>(A6, id6, e);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float s = 0.f;
#pragma unroll
                        for (int a = 0; a < 6; ++a) s = fmaf(J[a][j], e[a], s);
                        q[j] += s;
                    }
                }
#pragma unroll
                for (int i = 0; i < NJ; ++i) { q0[i] = q[i]; qd[i] = 0.f; dq[i] = 0.f; }
                t = 0; touched = 0; fzprev = 0.f; dfz = 0.f; vbar = 0.f; epret = 0.f; status = 0;
            }
        } else {
            t += 1;                                                  // MujocoEnv.step: timestep += 1
        }

        // =====================================================================================================
        // forward pass at (q, qd): kinematics, dynamics, controller, constrained accelerations, sensors
        // =====================================================================================================
        StepOut R;
        R.ncon = 0; R.overflow = 0;
        float qacc[NJ];
        f3 hv = mk(0, 0, 0);
        float obs[OBS_DIM];
        float pos_err_norm = 0.f, ori_err = 0.f;
        if (active) {
            Kin K; fk<G == 1>(M, q, K);
            Dyn D; dynamics(M, K, qd, D);
#pragma unroll
            for (int i = 0; i < NJ; ++i) D.M[PK(i, i)] += M.armature[i];          // rotor inertias (usim_config.armature_scale)
            // site Jacobian J = [Jv; Jw]
            float J[6][NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f3 jv = cross(K.z[j], K.x - K.o[j]);
                J[0][j] = jv.x; J[1][j] = jv.y; J[2][j] = jv.z; J[3][j] = K.z[j].x; J[4][j] = K.z[j].y; J[5][j] = K.z[j].z;
            }
            float Lm[28], idm[NJ];
#pragma unroll
            for (int k = 0; k < 28; ++k) Lm[k] = D.M[k];
            chol_packed<NJ>(Lm, idm);
            // Lambda^-1 = J M^-1 J^T = Y^T Y with Y = Lm^-1 J^T (six forward substitutions only; packed lower 6x6)
            float Li[21];
            {
                float Y[6][NJ];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) Y[a][j] = J[a][j];
                    chol_forward<NJ>(Lm, idm, Y[a]);
                }
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) s = fmaf(Y[a][j], Y[b][j], s);
                        Li[PK(a, b)] = s;
                    }
            }
            TSTAMP(2);
            // ---------------- OSC_POSE torque (robosuite osc.py run_controller; rl_config.yaml:33-51) ----------------
            float tau[NJ];
            if (pass == 0) {
                float kp[6], kd[6];
                f3 gpos, gx, gy, gz;
                float up = clampf((float)(t - 1) * inv_h + u0, 0.f, 1.f);   // controller.traj_pos from the previous _post_action
                f3 tpw = ts + (te - ts) * up;
                if (C.mode == 1) {
                    float d[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a) d[a] = clampf(act[a], -1.f, 1.f) * (a < 3 ? C.out_pos : C.out_ori);
                    gpos = K.x + mk(d[0], d[1], d[2]);
                    float ang = sqrt_(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                    if (ang < 1e-12f) { gx = K.sx; gy = K.sy; gz = K.sz; }
                    else {
                        float hh = 0.5f * ang, sh = sinf(hh) * rcp_(ang), qw = cosf(hh), qx = d[3] * sh, qy = d[4] * sh, qz = d[5] * sh;
                        // rotation matrix of the delta quaternion, applied on the left of the current orientation
                        f3 e0 = mk(1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy + qw * qz), 2.f * (qx * qz - qw * qy));
                        f3 e1 = mk(2.f * (qx * qy - qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz + qw * qx));
                        f3 e2 = mk(2.f * (qx * qz + qw * qy), 2.f * (qy * qz - qw * qx), 1.f - 2.f * (qx * qx + qy * qy));
                        gx = e0 * K.sx.x + e1 * K.sx.y + e2 * K.sx.z;
                        gy = e0 * K.sy.x + e1 * K.sy.y + e2 * K.sy.z;
                        gz = e0 * K.sz.x + e1 * K.sz.y + e2 * K.sz.z;
                    }
#pragma unroll
                    for (int a = 0; a < 6; ++a) { kp[a] = C.kp_fixed; kd[a] = 2.f * sqrt_(C.kp_fixed) * C.damping_ratio; }
                } else {
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        float v = (C.mode == 3) ? 0.f : clampf(act[a], 0.f, 1.f);     // wrench mode: no impedance term
                        kp[a] = C.kp_min + v * (C.kp_max - C.kp_min);
                        kd[a] = 2.f * sqrt_(kp[a]) * C.damping_ratio;
                    }
                    gpos = mk(tpw.x - M.base[0], tpw.y - M.base[1], tpw.z - M.base[2]);
                    if (C.mode == 2) gpos.z += clampf(act[6], -1.f, 1.f) * C.out_pos;
                    gx = mk(M.grot[0], M.grot[3], M.grot[6]); gy = mk(M.grot[1], M.grot[4], M.grot[7]); gz = mk(M.grot[2], M.grot[5], M.grot[8]);
                }
                float v6[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf(J[a][j], qd[j], s);
                    v6[a] = s;
                }
                f3 eo = (cross(K.sx, gx) + cross(K.sy, gy) + cross(K.sz, gz)) * 0.5f;
                f3 ep = gpos - K.x;
                float Fp[3] = {ep.x * kp[0] - v6[0] * kd[0], ep.y * kp[1] - v6[1] * kd[1], ep.z * kp[2] - v6[2] * kd[2]};
                float Tp[3] = {eo.x * kp[3] - v6[3] * kd[3], eo.y * kp[4] - v6[4] * kd[4], eo.z * kp[5] - v6[5] * kd[5]};
                if (C.mode == 3) {
                    // fork-only "wrench" baseline (utils/plot.py:267-268; checkpoint action box [-10,10]^6): the action takes the place
                    // of desired_force / desired_torque in the OSC law, i.e. wrench = [Lambda_pos a_f; Lambda_ori a_t].  Inferred; the
                    // shipped `wrench` policy replayed under this reading earns 9.2 reward/step (8.6 on MuJoCo), under "action =
                    // wrench" it fails within 80 steps (tests/test_gpu_policy_replay.py)
#pragma unroll
                    for (int a = 0; a < 3; ++a) { Fp[a] = clampf(act[a], -WRENCH_MAX, WRENCH_MAX); Tp[a] = clampf(act[3 + a], -WRENCH_MAX, WRENCH_MAX); }
                }
                // lambda_pos F, lambda_ori T : solves with the 3x3 diagonal blocks of Li (uncouple_pos_ori, rl_config.yaml:48)
                {
                    float P3[6] = {Li[PK(0, 0)], Li[PK(1, 0)], Li[PK(1, 1)], Li[PK(2, 0)], Li[PK(2, 1)], Li[PK(2, 2)]}, ip[3];
                    chol_packed<3>(P3, ip); chol_solve<3>(P3, ip, Fp);
                    float O3[6] = {Li[PK(3, 3)], Li[PK(4, 3)], Li[PK(4, 4)], Li[PK(5, 3)], Li[PK(5, 4)], Li[PK(5, 5)]}, io3[3];
                    chol_packed<3>(O3, io3); chol_solve<3>(O3, io3, Tp);
                }
                float wr[6] = {Fp[0], Fp[1], Fp[2], Tp[0], Tp[1], Tp[2]};
                // nullspace torque N^T M (10 (q0 - q) - 2 sqrt(10) qd)
                float pt[NJ], y[NJ];
#pragma unroll
                for (int i = 0; i < NJ; ++i) pt[i] = 10.f * (q0[i] - q[i]) - 6.3245553203367586f * qd[i];
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf((i >= j) ? D.M[PK(i, j)] : D.M[PK(j, i)], pt[j], s);
                    y[i] = s;
                }
                float jb[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf(J[a][j], pt[j], s);      // (M^-1 J^T)^T (M pt) = J pt
                    jb[a] = s;
                }
                {
                    float L6[21], i6[6];
#pragma unroll
                    for (int k = 0; k < 21; ++k) L6[k] = Li[k];
                    chol_packed<6>(L6, i6); chol_solve<6>(L6, i6, jb);
                }
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    float s = D.bias[i] + y[i];
#pragma unroll
                    for (int a = 0; a < 6; ++a) s = fmaf(J[a][i], wr[a] - jb[a], s);
                    tau[i] = clampf(s, -TAUMAX[i], TAUMAX[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NJ; ++i) tau[i] = 0.f;      // reset: sim.forward() with zero ctrl
            }
            if (pass == 0 && store && io.log) {
                // torque and action channels of the episode record leave the registers here instead of living to the end of the step
                float* L = io.log + (size_t)ei * LOG_WIDTH;
#pragma unroll
                for (int i = 0; i < NJ; ++i) L[33 + i] = tau[i];
#pragma unroll
                for (int a = 0; a < 7; ++a) L[46 + a] = act[a];
            }
            TSTAMP(3);
            // ---------------- smooth acceleration ----------------
            float qs[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) qs[i] = tau[i] - D.bias[i] - JOINT_DAMP * qd[i];
            chol_solve<NJ>(Lm, idm, qs);
            joint_friction(D.M, Lm, idm, qd, C.frictionloss, qs);

            float W[6] = {0, 0, 0, 0, 0, 0};          // site-space wrench of the contact forces
            float full_chk = 0.f;                     // full torso: |free body| + sum |sliders| after the integration, for the numerical fault guard
            if constexpr (TORSO == 2) {
                // ---------------- full torso (usim_full.h): 270 sliders on the free body, probe and table contacts ----------------
                float alpha[6], vs[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float s = 0.f, u = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { s = fmaf(J[a][j], qs[j], s); u = fmaf(J[a][j], qd[j], u); }
                    alpha[a] = s; vs[a] = u;
                }
                float acc_e[FE], ab[6], lat_chk = 0.f;
                int cel[MAXC], nc = 0, ovf = 0;
                // (the contact solve starts from the forces of the previous physics step, kept in the environment's lattice block; a reset pass starts cold and leaves none)
                float* const latp = &LAT(0);
                full_forward(lds, gl, M, C, kst, kdmp, mu, s_pre, sd_pre, body, K.x, K.sx, K.sy, K.sz, Li, alpha, vs, W, acc_e, ab, nc, cel, ovf,
                             (pass == 0) ? latp : nullptr, (pass == 0 && valid) ? latp : nullptr);
                // The arm quantities the rest of the pass needs (kinematics, mass matrix and its factor, bias, site Jacobian: ~250 words) are formed AGAIN here, from joint
                // state the compiler cannot recognise, instead of living through the contact solve: 2.5 k instructions against the solve's 700 k, and without them the
                // kernel fits 256 registers with no scratch -- two environments per SIMD.  Same inputs, same instructions: the same bits.
#pragma unroll
                for (int i = 0; i < NJ; ++i) asm volatile("" : "+v"(q[i]), "+v"(qd[i]));
                fk<G == 1>(M, q, K);
                dynamics(M, K, qd, D);
#pragma unroll
                for (int i = 0; i < NJ; ++i) D.M[PK(i, i)] += M.armature[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f3 jv = cross(K.z[j], K.x - K.o[j]);
                    J[0][j] = jv.x; J[1][j] = jv.y; J[2][j] = jv.z; J[3][j] = K.z[j].x; J[4][j] = K.z[j].y; J[5][j] = K.z[j].z;
                }
#pragma unroll
                for (int k = 0; k < 28; ++k) Lm[k] = D.M[k];
                chol_packed<NJ>(Lm, idm);
                R.ncon = nc; R.overflow = ovf;
#pragma unroll
                for (int k = 0; k < MAXC; ++k) R.con_shell[k] = cel[k];
                // semi-implicit Euler: sliders; free body (linear part in world axes, angular velocity in the body frame, quaternion by the exponential of dt w / 2)
#pragma unroll
                for (int i = 0; i < FE; ++i) {
                    const int e = FE * gl + i;
                    float sdn = 0.f, sn = 0.f;
                    if (pass == 0) { sdn = fmaf(dt, acc_e[i], sd_pre[i]); sn = fmaf(dt, sdn, s_pre[i]); }
                    if (valid && e < NSH && (pass == 0 || !refill)) { LAT(LATF_SD + e) = sdn; LAT(LATF_S + e) = sn; }
                    if (e < NSH) lat_chk += fabsf(sn) + 1e-3f * fabsf(sdn);
                }
                if (pass == 0) {
                    const float qw = body.q[0], qx = body.q[1], qy = body.q[2], qz = body.q[3];
                    const f3 abl = mk(ab[0], ab[1], ab[2]);
                    const f3 aw = mk((1.f - 2.f * (qy * qy + qz * qz)) * abl.x + 2.f * (qx * qy - qw * qz) * abl.y + 2.f * (qx * qz + qw * qy) * abl.z,
                                     2.f * (qx * qy + qw * qz) * abl.x + (1.f - 2.f * (qx * qx + qz * qz)) * abl.y + 2.f * (qy * qz - qw * qx) * abl.z,
                                     2.f * (qx * qz - qw * qy) * abl.x + 2.f * (qy * qz + qw * qx) * abl.y + (1.f - 2.f * (qx * qx + qy * qy)) * abl.z);
                    body.v = madd(body.v, aw, dt); body.p = madd(body.p, body.v, dt);
                    body.w = madd(body.w, mk(ab[3], ab[4], ab[5]), dt);
                    const float wn = sqrt_(dot(body.w, body.w)), hh = 0.5f * dt * wn;
                    float shh, chh; sincosf(hh, &shh, &chh);
                    const float sh = (wn > 1e-12f) ? shh * rcp_(wn) : 0.5f * dt;
                    const float dx = body.w.x * sh, dy = body.w.y * sh, dz2 = body.w.z * sh;
                    const float n0 = qw * chh - qx * dx - qy * dy - qz * dz2, n1 = qw * dx + qx * chh + qy * dz2 - qz * dy;
                    const float n2 = qw * dy - qx * dz2 + qy * chh + qz * dx, n3 = qw * dz2 + qx * dy - qy * dx + qz * chh;
                    const float irn = rsq_(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
                    body.q[0] = n0 * irn; body.q[1] = n1 * irn; body.q[2] = n2 * irn; body.q[3] = n3 * irn;
                    // what the numerical fault guard below sees of the torso: the free body's 13 words and every slider (a non-finite word makes the sum non-finite; without
                    // this a torso gone NaN fails every comparison of full_forward, its contacts vanish silently and the arm -- all the guard used to look at -- stays finite)
                    full_chk = wave_sum(lat_chk) + fabsf(body.p.x) + fabsf(body.p.y) + fabsf(body.p.z) + fabsf(body.q[0]) + fabsf(body.q[1]) + fabsf(body.q[2]) + fabsf(body.q[3])
                             + 1e-3f * (fabsf(body.v.x) + fabsf(body.v.y) + fabsf(body.v.z) + fabsf(body.w.x) + fabsf(body.w.y) + fabsf(body.w.z));
                }
                if (store && (pass == 0 || !refill)) {
                    const float bw[13] = {body.p.x, body.p.y, body.p.z, body.q[0], body.q[1], body.q[2], body.q[3], body.v.x, body.v.y, body.v.z, body.w.x, body.w.y, body.w.z};
#pragma unroll
                    for (int a = 0; a < 13; ++a) LAT(LATF_BODY + a) = bw[a];
                }
                if (pass == 1 && valid && !refill) {                         // a reset of the live state: the episode starts without a warm start
                    for (int w = gl; w < LATF_WARM_WORDS; w += G) LAT(LATF_WTAB + w) = (w >= 4 * NSH && w < 4 * NSH + 8) ? __int_as_float(-1) : 0.f;
                }
            } else if (TORSO) {
                const int* tb_shell = reinterpret_cast<const int*>(lds + TB_SHELL);
                const int tsim = (t > 0) ? t - 1 : 0;
                float dz, vz, az;
                torso_motion(C, tsim, dz, vz, az);
                TSTAMP(4);
                int nc = lattice_front<G, NE, MODE == 0>(lds, eb, gl, gbase, M, C, tsim, kst, kdmp, pass == 0, s_pre, sd_pre, K.x, K.sy, K.sz, io.dbg);
                TSTAMP(7);
                if (nc > MAXC) { R.overflow = 1; nc = MAXC; }
                R.ncon = nc;
                group_sync();
                int ncmax = 0;                                       // wave-uniform bound on the contact count
#pragma unroll
                for (int k = MAXC; k >= 1; --k) if (ncmax == 0 && __any(nc >= k)) ncmax = k;
                float gf[MAXC];
                int cel[MAXC];
#pragma unroll
                for (int k = 0; k < MAXC; ++k) { gf[k] = 0.f; cel[k] = (k < nc) ? __float_as_int(EB(GE_CG + k * CG_WORDS + 6)) : 0; }
                if (ncmax > 0) {
                    float alpha[6], vs[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        float s = 0.f, u = 0.f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) { s = fmaf(J[a][j], qs[j], s); u = fmaf(J[a][j], qd[j], u); }
                        alpha[a] = s; vs[a] = u;
                    }
                    TSTAMP(8);
                    contact_solve<G, false>(lds, eb, gl, M, C, nc, ncmax, cel, Li, alpha, vs, mu, vz, ContactRows{}, W, gf, io.dbg);
                }
                TSTAMP(11);
                // ---- element accelerations a = a~ + Linv[:, e_c] gf_c, semi-implicit Euler, write back ----
                {
                    float acc_e[NE];
#pragma unroll
                    for (int i = 0; i < NE; ++i) { const int e = gl + i * G; acc_e[i] = (pass == 0 && e < N_TOP) ? EB(GE_A + e) : 0.f; }
                    if (pass == 0) {
#pragma unroll
                        for (int k = 0; k < MAXC; ++k) {     // contact outer (one wave-uniform test per slot), elements inner; slots beyond this
                            if (k < ncmax) {                 // environment's count carry gf = 0 and element 0
#pragma unroll
                                for (int i = 0; i < NE; ++i) {
                                    const int e = (gl + i * G < N_TOP) ? gl + i * G : N_TOP - 1;
                                    acc_e[i] = fmaf(lds[TB_LINV + e * LROW + cel[k]], gf[k], acc_e[i]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NE; ++i) {
                        const int e = gl + i * G;
                        if (e >= N_TOP) continue;
                        float sdn = 0.f, sn = 0.f;
                        if (pass == 0) {
                            sdn = sd_pre[i] + dt * acc_e[i];
                            sn = s_pre[i] + dt * sdn;
                        }
                        if (valid && (pass == 0 || !refill)) { LAT(LAT_SD + e) = sdn; LAT(LAT_S + e) = sn; }
                    }
                }
#pragma unroll
                for (int k = 0; k < MAXC; ++k) R.con_shell[k] = (k < nc) ? tb_shell[cel[k]] : -1;
            } else {
#pragma unroll
                for (int k = 0; k < MAXC; ++k) R.con_shell[k] = -1;
            }
            TSTAMP(12);
            // ---------------- constrained arm acceleration: qacc = qs + M^-1 J^T W ----------------
#pragma unroll
            for (int i = 0; i < NJ; ++i) qacc[i] = qs[i];
            if (TORSO) {
                float z[NJ];
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int a = 0; a < 6; ++a) s = fmaf(J[a][i], W[a], s);
                    z[i] = s;
                }
                chol_solve<NJ>(Lm, idm, z);
#pragma unroll
                for (int i = 0; i < NJ; ++i) qacc[i] += z[i];
            }
            R.fc[0] = W[0]; R.fc[1] = W[1]; R.fc[2] = W[2];
            // ---------------- torque sensor at ft_frame (MuJoCo cfrc_int of the probe body, site frame) ----------------
            {
                // link-7 accelerations from the site Jacobian: alpha = alpha_bias + Jw qacc, a(o7) = a_bias + Jv qacc - (Jw qacc) x (x - o7)
                float aq[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float sacc = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sacc = fmaf(J[a][j], qacc[j], sacc);
                    aq[a] = sacc;
                }
                const f3 alq = mk(aq[3], aq[4], aq[5]);
                f3 al = D.al7 + alq;
                f3 a7 = D.a7 + mk(aq[0], aq[1], aq[2]) - cross(alq, K.x - K.o[NJ - 1]);
                f3 rc = K.r7x * M.pcom7[0] + K.r7y * M.pcom7[1] + K.r7z * M.pcom7[2];
                f3 ac = a7 + cross(al, rc) + cross(D.w7, cross(D.w7, rc));
                f3 N = rot_inertia_mul(K, M.pI7, al) + cross(D.w7, rot_inertia_mul(K, M.pI7, D.w7));
                f3 Fp = ac * PROBE_MASS;
                f3 tw = N + cross(K.o[NJ - 1] + rc - K.x, Fp) - mk(W[3], W[4], W[5]);
                R.tq[0] = dot(K.sx, tw); R.tq[1] = dot(K.sy, tw); R.tq[2] = dot(K.sz, tw);
            }
            TSTAMP(13);
            // ---------------- integrate the arm: mj_Euler with implicit joint damping ----------------
            if (pass == 0) {
                // (M + h D) x = M qacc with D = d I, h d = 2e-5: x = qacc - h d M^-1 x.  One step from x = qacc reuses the factor of M; the
                // contraction is h d / lambda_min(M) = 2.8e-4 (lambda_min(M) = 0.071 kg m^2 over the workspace), so the remainder
                // is 8e-8 relative -- fp32 rounding.  No second factorisation, and the mass matrix is dead before the contact phase.
                float rhs[NJ];
                {
                    const float hd = dt * JOINT_DAMP;
                    float xk[NJ];
#pragma unroll
                    for (int i = 0; i < NJ; ++i) xk[i] = qacc[i];
                    chol_solve<NJ>(Lm, idm, xk);
#pragma unroll
                    for (int i = 0; i < NJ; ++i) rhs[i] = fmaf(-hd, xk[i], qacc[i]);
                }
#pragma unroll
                for (int i = 0; i < NJ; ++i) { qd[i] = fmaf(dt, rhs[i], qd[i]); dq[i] = fmaf(dt, qd[i], dq[i]); q[i] = q0[i] + dq[i]; }
                // hand velocity: Jacobian from before the integration, qvel from after (mj_step data semantics)
                float vs2[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf(J[a][j], qd[j], s);
                    vs2[a] = s;
                }
                hv = mk(vs2[0], vs2[1], vs2[2]) + cross(mk(vs2[3], vs2[4], vs2[5]), K.hand - K.x);
            }
            TSTAMP(14);
            // ---------------- observation (ultrasound.py:363-401) ----------------
            {
                const int tprev = (pass == 0) ? t - 1 : 0;
                float up = clampf((float)tprev * inv_h + u0, 0.f, 1.f);
                f3 tpw = ts + (te - ts) * up;
                if (pass == 1) fzbar = R.fc[2];                          // ultrasound.py:477
                obs[0] = R.fc[0]; obs[1] = R.fc[1]; obs[2] = R.fc[2];
                obs[3] = R.tq[0]; obs[4] = R.tq[1]; obs[5] = R.tq[2];
                obs[6] = hv.x; obs[7] = hv.y; obs[8] = hv.z;
                obs[9] = fzbar - 5.0f; obs[10] = dfz - 0.0f; obs[11] = vbar - 0.04f;
                f3 xw = mk(K.x.x + M.base[0], K.x.y + M.base[1], K.x.z + M.base[2]);
                obs[12] = xw.x - tpw.x; obs[13] = xw.y - tpw.y; obs[14] = xw.z - tpw.z;
                float qe[4]; mat2quat_xyzw(K.sx, K.sy, K.sz, qe);
                difference_quat(qe, M.gquat, obs + 15);               // xyzw arrays through the wxyz routine (ultrasound.py:390)
                if (pass == 0) {
                    // ---------------- reward (ultrasound.py:230-269) ----------------
                    const bool contact = R.ncon > 0;
                    if (contact) touched = 1;
                    float pe0 = 90.f * (xw.x - tpw.x), pe1 = 90.f * (xw.y - tpw.y);
                    pe0 *= pe0; pe1 *= pe1;
                    pos_err_norm = sqrt_(pe0 * pe0 + pe1 * pe1);
                    float pos_rew = 5.f * exp_(-pos_err_norm);
                    float qc[4] = {qe[3], qe[0], qe[1], qe[2]};
                    ori_err = 0.2f * distance_quat_goal(qc, M.ghat, M.geps);
                    float ori_rew = exp_(-ori_err);
                    float ve = 45.f * (vbar - 0.04f); ve *= ve;
                    float vel_rew = exp_(-ve);
                    float fe = 0.7f * (fzbar - 5.f); fe *= fe;
                    float force_rew = contact ? 3.f * exp_(-fe) : 0.f;
                    float de = 0.01f * dfz; de *= de;
                    float dforce_rew = contact ? 2.f * exp_(-de) : 0.f;
                    float reward = pos_rew + ori_rew + vel_rew + force_rew + dforce_rew;
                    done = t >= C.horizon;
                    // ---------------- bookkeeping (ultrasound.py:528-546) ----------------
                    float hvn = sqrt_(dot(hv, hv));
                    vbar += (hvn - vbar) * rcp_((float)t);
                    float fz = R.fc[2];
                    dfz = (fz - fzprev) * rcp_(dt);
                    fzprev = fz;
                    fzbar = 0.1f * fz + 0.9f * fzbar;
                    if (C.early_term) {                                // ultrasound.py:635-670
                        bool term = false;
#pragma unroll
                        for (int i = 0; i < NJ; ++i) term = term || (q[i] < QMIN[i] + 0.1f) || (q[i] > QMAX[i] - 0.1f);
                        term = term || (pos_err_norm > 1.0f) || (contact && ori_err > 0.10f) || (touched && !contact);
                        done = done || term;
                    }
                    epret += reward;
                    if (store && io.log) {
                        // per-step episode record in the order of the reference's CSV dump (ultrasound.py:552-614)
                        float* L = io.log + (size_t)ei * LOG_WIDTH;
                        const float upn = clampf((float)t * inv_h + u0, 0.f, 1.f);
                        const f3 tpn = ts + (te - ts) * upn;                                    // trajectory point after this step's update (:532)
                        L[0] = xw.x; L[1] = xw.y; L[2] = xw.z; L[3] = tpn.x; L[4] = tpn.y; L[5] = tpn.z;
                        L[6] = hv.x; L[7] = hv.y; L[8] = hv.z; L[9] = 0.04f; L[10] = vbar;
                        L[11] = qe[0]; L[12] = qe[1]; L[13] = qe[2]; L[14] = qe[3];
                        L[15] = M.gquat[0]; L[16] = M.gquat[1]; L[17] = M.gquat[2]; L[18] = M.gquat[3];
                        L[19] = ori_err * 5.0f;                                                 // distance_quat (ori_err = 0.2 * distance)
                        L[20] = fz; L[21] = 5.0f; L[22] = fzbar; L[23] = dfz; L[24] = 0.f; L[25] = contact ? 1.f : 0.f;
#pragma unroll
                        for (int i = 0; i < NJ; ++i) L[26 + i] = q[i];
                        L[40] = (float)(t - 1) * inv_h * 100.f;
                        L[41] = pos_rew; L[42] = ori_rew; L[43] = vel_rew; L[44] = force_rew; L[45] = dforce_rew;
                    }
                    if (R.overflow) status |= (TORSO == 2) ? (R.overflow & 3) : 1;      // (full torso: bit 1 = more element-table contacts than the kernel keeps)
                    {
                        // numerical fault guard: a non-finite or run-away state ends the episode and is flagged
                        float chk = (TORSO == 2) ? full_chk : 0.f;
#pragma unroll
                        for (int i = 0; i < NJ; ++i) chk += fabsf(q[i]) + 1e-3f * fabsf(qd[i]);
                        if (!(chk < 1.0e3f)) { status |= 4; done = true; epret -= reward; reward = 0.f; if (!(epret == epret)) epret = 0.f; }
                    }
                    if (store) {
                        io.rew[ei] = reward;
                        if (io.status_out) io.status_out[ei] = status;
                        io.done[ei] = done ? 1 : 0;
                        if (io.contacts) {
                            io.contacts[(size_t)ei * (1 + MAXC)] = R.ncon;
#pragma unroll
                            for (int k = 0; k < MAXC; ++k) io.contacts[(size_t)ei * (1 + MAXC) + 1 + k] = R.con_shell[k];
                        }
                        if (done) {
                            if (io.term_obs) {
#pragma unroll
                                for (int a = 0; a < OBS_DIM; ++a) io.term_obs[(size_t)ei * OBS_DIM + a] = obs[a];
                            }
                            if (io.ep_ret) io.ep_ret[ei] = epret;
                            if (io.ep_len) io.ep_len[ei] = t;
                        }
                    }
                    need = done && auto_reset;
                }
                if (pass == 1 && refill) {
                    // reset computed ahead of time: park it in the bank slot of episode ep_t
                    if (store && need) {
                        const int sl = ep_t & (BANK_DEPTH - 1);
#pragma unroll
                        for (int i = 0; i < NJ; ++i) BK(sl, BQ0 + i) = q[i];
                        BK(sl, BTS) = ts.x; BK(sl, BTS + 1) = ts.y; BK(sl, BTS + 2) = ts.z; BK(sl, BTE) = te.x; BK(sl, BTE + 1) = te.y; BK(sl, BTE + 2) = te.z;
                        BK(sl, BU0) = u0; BK(sl, BKST) = kst; BK(sl, BKDMP) = kdmp; BK(sl, BMU) = mu; BK(sl, BFZ) = fzbar;
#pragma unroll
                        for (int a = 0; a < OBS_DIM; ++a) BK(sl, BOBS + a) = obs[a];
                        BKI(sl, BSTATUS) = (TORSO == 2) ? (R.overflow & 3) : (R.overflow ? 1 : 0);
                    }
                } else if (store && io.obs && (pass == 1 ? need : !need)) {
#pragma unroll
                    for (int a = 0; a < OBS_DIM; ++a) io.obs[(size_t)ei * OBS_DIM + a] = obs[a];
                }
                if (pass == 1 && R.overflow) status |= (TORSO == 2) ? (R.overflow & 3) : 1;
            }
        }
    } while (0);

    if (MODE == 0 && need) {
        // ================= auto-reset: adopt the initial state prepared in the reset bank (SB3 VecEnv semantics: the
        // observation returned for a finished environment is its reset observation) and queue the slot for refill ==========
        episode += 1;
        const int sl = episode & (BANK_DEPTH - 1);
#pragma unroll
        for (int i = 0; i < NJ; ++i) { q[i] = BK(sl, BQ0 + i); q0[i] = q[i]; qd[i] = 0.f; dq[i] = 0.f; }
        ts = mk(BK(sl, BTS), BK(sl, BTS + 1), BK(sl, BTS + 2)); te = mk(BK(sl, BTE), BK(sl, BTE + 1), BK(sl, BTE + 2));
        u0 = BK(sl, BU0); kst = BK(sl, BKST); kdmp = BK(sl, BKDMP); mu = BK(sl, BMU); fzbar = BK(sl, BFZ);
        t = 0; touched = 0; fzprev = 0.f; dfz = 0.f; vbar = 0.f; epret = 0.f; status = BKI(sl, BSTATUS);
        if (store && io.obs) {
#pragma unroll
            for (int a = 0; a < OBS_DIM; ++a) io.obs[(size_t)ei * OBS_DIM + a] = BK(sl, BOBS + a);
        }
        if (TORSO == 1 && valid) for (int e = gl; e < N_TOP; e += G) { LAT(LAT_S + e) = 0.f; LAT(LAT_SD + e) = 0.f; }
        if (TORSO == 2 && valid) {
            // (every word by the lane that wrote it in the step above)
            for (int i = 0; i < FE; ++i) { const int e = FE * gl + i; if (e < NSH) { LAT(LATF_S + e) = 0.f; LAT(LATF_SD + e) = 0.f; } }
            if (gl == 0) {
                const float bw[13] = {M.torso[0], M.torso[1], M.torso[2], 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int a = 0; a < 13; ++a) LAT(LATF_BODY + a) = bw[a];
            }
            group_sync();                                                    // (the step above left its warm start through other lanes)
            for (int w = gl; w < LATF_WARM_WORDS; w += G) LAT(LATF_WTAB + w) = (w >= 4 * NSH && w < 4 * NSH + 8) ? __int_as_float(-1) : 0.f;
        }
        // the slot just consumed is free again: order the episode that will occupy it (computed by the next bulk refill,
        // which runs at least every BANK_DEPTH steps, i.e. before this environment can come round to the slot again)
        if (store) { const int idx = atomicAdd(io.count, 1); io.items[idx] = make_int2(env, episode + BANK_DEPTH); }
    }

    TSTAMP(15);
    // ---------------- store state ----------------
    if (store && !(MODE == 1 && (refill || !need))) {
        {
            // 16-byte stores of the quads that hold a changed word (q, qd | running statistics | counters); the quads of per-episode
            // constants only when an episode starts
            float o[F_NSCALAR];
#pragma unroll
            for (int i = 0; i < NJ; ++i) { o[F_Q + i] = dq[i]; o[F_QD + i] = qd[i]; o[F_Q0 + i] = q0[i]; }
            o[F_TS] = ts.x; o[F_TS + 1] = ts.y; o[F_TS + 2] = ts.z; o[F_TE] = te.x; o[F_TE + 1] = te.y; o[F_TE + 2] = te.z;
            o[F_U0] = u0; o[F_VBAR] = vbar; o[F_FZBAR] = fzbar; o[F_FZPREV] = fzprev; o[F_DFZ] = dfz;
            o[F_KST] = kst; o[F_KDMP] = kdmp; o[F_MU] = mu; o[F_EPRET] = epret;
            o[F_T] = __int_as_float(t); o[F_TOUCH] = __int_as_float(touched); o[F_EPISODE] = __int_as_float(episode); o[F_STATUS] = __int_as_float(status);
            float4* sp = reinterpret_cast<float4*>(st + scalar_index(0, (size_t)ei));
            const bool all = (MODE == 1) || need;
#pragma unroll
            for (int v = 0; v < F_NSCALAR / 4; ++v) {
                const bool changed = (v <= 3) || v == 7 || v == 8 || v == 9;      // words 0-15 (q, qd, q0[0..1]), 28-31, 32-39
                if (changed || all) sp[v] = make_float4(o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]);
            }
        }
    }
#undef ST
#undef STI
#undef LAT
#undef EB
    if (refill) group_sync();                         // next item reuses the per-environment LDS block
    }   // item loop
    if (MODE == 1 && refill) {
        // the last workgroup to finish empties the work list for the step kernels that follow on the stream
        __syncthreads();
        if (threadIdx.x == 0) {
            // (no device-scope fence: the list is read by the launches that FOLLOW on the stream, and a fence costs an L2 write-back per wave)
            if (atomicAdd(io.count + 1, 1) == (int)gridDim.x - 1) { io.count[0] = 0; io.count[1] = 0; }
        }
    }
    TSTAMP(16);
#undef BK
#undef BKI
#undef TSTAMP
}

// work items (env, episode + k), k = 1..BANK_DEPTH, for the environments selected by mask (reset / set_state paths)
__global__ void usim_bank_items_kernel(const float* __restrict__ st, int n, const uint8_t* __restrict__ mask, int2* items, int* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * BANK_DEPTH) return;
    const int env = i / BANK_DEPTH, k = i % BANK_DEPTH + 1;
    if (mask && !mask[env]) return;
    const int episode = reinterpret_cast<const int*>(st)[scalar_index(F_EPISODE, (size_t)env)];
    items[atomicAdd(count, 1)] = make_int2(env, episode + k);
}


// synthetic actions of BASELINE.md section 4 (same stream as the in-kernel LF_RANDOM_ACT path)
__global__ void usim_random_actions_kernel(const DevCfg C, int n, long long rstep, float* __restrict__ act) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t gid = (uint32_t)(C.env_offset + i);
    u4 r1 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 1u, C.key0, C.key1);
    u4 r2 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 2u, C.key0, C.key1);
    uint32_t rr[8] = {r1.a, r1.b, r1.c, r1.d, r2.a, r2.b, r2.c, r2.d};
    for (int a = 0; a < C.adim; ++a) {
        float u = u01(rr[a]);
        bool sgn = (C.mode == 1) || (C.mode == 3) || (C.mode == 2 && a == 6);
        float v = sgn ? 2.f * u - 1.f : u;
        act[(size_t)i * C.adim + a] = (C.mode == 3) ? v * WRENCH_MAX : v;
    }
}

}  // namespace usim

#include "usim_step16.h"
