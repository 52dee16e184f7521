// usim_kernels.hip -- CDNA4 (gfx950) kernels of the batched Ultrasound simulator.
//
// One environment per lane, one wave64 per workgroup.  All 7-DoF arm mathematics (forward kinematics,
// composite-rigid-body mass matrix, recursive Newton-Euler bias, 7x7/6x6/3x3 Cholesky solves, the OSC torque
// law) lives in VGPRs, fully unrolled; the 99-element torso lattice and the contact rows are staged in LDS as
// [word][lane] rows (bank = lane, conflict-free for wave-uniform word indices); per-environment state is read
// and written once per step as coalesced 256-byte rows of the SoA state block in HBM.
//
// The step replaces, per environment:
//   a1 robosuite MujocoEnv.step driver            a2 OSC_POSE controller (rl_config.yaml:33-51)
//   a3 MuJoCo mj_step (forward dynamics + soft constraints + Euler)
//   a4 Ultrasound.reward  ultrasound.py:230-269   a5 sensors ultrasound.py:363-401
//   a6 _post_action / _check_terminated ultrasound.py:512-551, 635-670
//   a7 probe<->torso contact predicate ultrasound.py:673-736          a8 utils/quaternion.py
//   a10 reset ultrasound.py:416-478 (trajectory sampling, initial-pose IK, noise, solref randomisation)
// Model and algorithm are specified in DESIGN.md; this file is written independently of oracle/.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usim_device.h"

namespace usim {

// ------------------------------------------------------------------------------------------------------------
// small vector helpers
// ------------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
#define DI __device__ __forceinline__
DI f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DI f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DI f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DI f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DI f3 operator*(float s, f3 a) { return mk(a.x * s, a.y * s, a.z * s); }
DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DI f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DI f3 madd(f3 a, f3 b, float s) { return mk(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)); }
DI float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
// symmetric 3x3 (xx,xy,xz,yy,yz,zz) times vector
DI f3 symmul(const float* I, f3 v) {
    return mk(I[0] * v.x + I[1] * v.y + I[2] * v.z, I[1] * v.x + I[3] * v.y + I[4] * v.z, I[2] * v.x + I[4] * v.y + I[5] * v.z);
}

// ------------------------------------------------------------------------------------------------------------
// Panda chain (robosuite asset, un-vendored; SURVEY.md Appendix B.4 -- the build's own model definition)
// link i: fixed translation, fixed rotation about x by ROTX[i]*90 deg, then the joint rotation about z
// ------------------------------------------------------------------------------------------------------------
__device__ constexpr float LPOS[NJ][3] = {{0.f, 0.f, 0.333f}, {0.f, 0.f, 0.f}, {0.f, -0.316f, 0.f}, {0.0825f, 0.f, 0.f},
                                          {-0.0825f, 0.384f, 0.f}, {0.f, 0.f, 0.f}, {0.088f, 0.f, 0.f}};
__device__ constexpr int ROTX[NJ] = {0, -1, 1, 1, -1, 1, 1};
__device__ constexpr float LCOM[NJ][3] = {{0.f, 0.f, -0.07f}, {0.f, -0.1f, 0.f}, {0.04f, 0.f, -0.05f}, {-0.04f, 0.05f, 0.f},
                                          {0.f, 0.f, -0.15f}, {0.06f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // [6] comes from DevModel
__device__ constexpr float LMASS[NJ] = {3.f, 3.f, 2.f, 2.f, 2.f, 1.5f, 0.f};                       // [6] comes from DevModel
__device__ constexpr float LISO[NJ] = {0.3f, 0.3f, 0.2f, 0.2f, 0.2f, 0.1f, 0.f};                     // isotropic inertias
__device__ constexpr float QMIN[NJ] = {-2.8973f, -1.7628f, -2.8973f, -3.0718f, -2.8973f, -0.0175f, -2.8973f};
__device__ constexpr float QMAX[NJ] = {2.8973f, 1.7628f, 2.8973f, -0.0698f, 2.8973f, 3.7525f, 2.8973f};
__device__ constexpr float TAUMAX[NJ] = {80.f, 80.f, 80.f, 80.f, 12.f, 12.f, 12.f};
__device__ constexpr float INITQ[NJ] = {0.f, 0.19634954084936207f, 0.f, -2.6179938779914944f, 0.f, 2.941592653589793f, 0.7853981633974483f};
constexpr float JOINT_DAMP = 0.1f;
constexpr float GRAV = 9.81f;
constexpr float PROBE_MASS = 1.0f;
constexpr float ELEM_R = 0.0075f, ELEM_MASS = 0.01f;
constexpr float TORSO_DROP = 0.0047f;
// MuJoCo default soft-constraint parameters (solref 0.02 1, solimp 0.9 0.95 0.001 0.5 2) and robosuite's impratio
constexpr float SR_TC = 0.02f, SI_D0 = 0.9f, SI_DMAX = 0.95f, SI_WIDTH = 0.001f, IMPRATIO = 20.f;
constexpr float PI_F = 3.14159265358979323846f;

// ------------------------------------------------------------------------------------------------------------
// torso lattice tables (identical for every handle; uploaded once per device by usim_create).  They live in the
// constant address space so that wave-uniform indices turn into scalar loads (s_load_dword*) and the
// coefficients are consumed straight from SGPRs.
// ------------------------------------------------------------------------------------------------------------
__constant__ float c_el_pos[N_TOP * 3];                       // nominal surface point rel. torso centre
__constant__ float c_el_axis[N_TOP * 3];                      // slide axis
__constant__ int c_el_nbr[N_TOP * 4];                         // neighbour element, -1 pinned (side face), -2 none
__constant__ int c_el_shell[N_TOP];                           // shell id (contact-pair index convention)
__constant__ float c_linv_blk[LINV_NBLK * N_TOP * LINV_BLK];  // blocked inverse of the lattice normal matrix
__constant__ float c_linv[N_TOP * N_TOP];                     // same, plain row-major

// ------------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al. 2011)
// ------------------------------------------------------------------------------------------------------------
struct u4 { uint32_t a, b, c, d; };
DI u4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u4 o; o.a = c0; o.b = c1; o.c = c2; o.d = c3; return o;
}
DI float u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
DI float u01_open(uint32_t u) { return (float)((u >> 8) + 1u) * (1.0f / 16777216.0f); }
DI uint32_t urange(uint32_t u, uint32_t n) { return __umulhi(u, n); }

// ------------------------------------------------------------------------------------------------------------
// packed-lower Cholesky helpers, fully unrolled (registers only)
// ------------------------------------------------------------------------------------------------------------
#define PK(i, j) ((i) * ((i) + 1) / 2 + (j))
template <int N>
DI void chol_packed(float* L, float* invd) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float d = L[PK(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fmaf(-L[PK(j, k)], L[PK(j, k)], d);
        d = sqrtf(fmaxf(d, 1e-30f));
        float inv = 1.0f / d;
        L[PK(j, j)] = d; invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            float s = L[PK(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fmaf(-L[PK(i, k)], L[PK(j, k)], s);
            L[PK(i, j)] = s * inv;
        }
    }
}
template <int N>
DI void chol_solve(const float* L, const float* invd, float* b) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fmaf(-L[PK(i, k)], b[k], s);
        b[i] = s * invd[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        float s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s = fmaf(-L[PK(k, i)], b[k], s);
        b[i] = s * invd[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// arm kinematics + dynamics
// ------------------------------------------------------------------------------------------------------------
struct Kin {
    f3 o[NJ], z[NJ];          // joint origins / axes (base-centred world axes)
    f3 c[NJ];                 // link COMs
    f3 r7x, r7y, r7z;         // link-7 rotation columns
    f3 x, sx, sy, sz;         // eef site position and rotation columns
    f3 hand;                  // right_hand body origin
};

DI void fk(const DevModel& M, const float* q, Kin& K) {
    f3 px = mk(1.f, 0.f, 0.f), py = mk(0.f, 1.f, 0.f), pz = mk(0.f, 0.f, 1.f), po = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        f3 o = po + px * LPOS[i][0] + py * LPOS[i][1] + pz * LPOS[i][2];
        f3 ax = px, ay, az;
        if (ROTX[i] == 0) { ay = py; az = pz; }
        else if (ROTX[i] > 0) { ay = pz; az = mk(-py.x, -py.y, -py.z); }
        else { ay = mk(-pz.x, -pz.y, -pz.z); az = py; }
        float s, c;
        sincosf(q[i], &s, &c);
        f3 nx = ax * c + ay * s, ny = ay * c - ax * s;
        K.o[i] = o; K.z[i] = az;
        if (i < NJ - 1) K.c[i] = o + nx * LCOM[i][0] + ny * LCOM[i][1] + az * LCOM[i][2];
        else K.c[i] = o + nx * M.c7[0] + ny * M.c7[1] + az * M.c7[2];
        px = nx; py = ny; pz = az; po = o;
    }
    K.r7x = px; K.r7y = py; K.r7z = pz;
    K.x = po + px * M.site7[0] + py * M.site7[1] + pz * M.site7[2];
    K.hand = po + px * M.hand7[0] + py * M.hand7[1] + pz * M.hand7[2];
    // site frame = link-7 frame rotated by -45 deg about z (robosuite right_hand quat 0.924 0 0 -0.383)
    const float h = 0.70710678118654752f;
    K.sx = (px - py) * h; K.sy = (px + py) * h; K.sz = pz;
}

struct Dyn {
    float M[28];              // mass matrix, packed lower
    float bias[NJ];           // qfrc_bias (gravity + Coriolis/centrifugal)
    f3 w7, al7, a7;           // link 7: angular velocity, bias angular accel., bias accel. of its origin (incl. +g)
};

// R7 * I7 * R7^T * v for the link-7 frame symmetric inertia I7
DI f3 rot_inertia_mul(const Kin& K, const float* I, f3 v) {
    f3 l = mk(dot(K.r7x, v), dot(K.r7y, v), dot(K.r7z, v));
    f3 t = symmul(I, l);
    return K.r7x * t.x + K.r7y * t.y + K.r7z * t.z;
}

DI void dynamics(const DevModel& M, const Kin& K, const float* qd, Dyn& D) {
    // ---- recursive Newton-Euler with qdd = 0, gravity as base acceleration +g ----
    f3 F[NJ], Nc[NJ];
    f3 w = mk(0, 0, 0), al = mk(0, 0, 0), a = mk(0, 0, GRAV), op = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        f3 r = K.o[i] - op;
        a = a + cross(al, r) + cross(w, cross(w, r));
        al = al + cross(w, K.z[i]) * qd[i];
        w = w + K.z[i] * qd[i];
        f3 rc = K.c[i] - K.o[i];
        f3 ac = a + cross(al, rc) + cross(w, cross(w, rc));
        float mi = (i < NJ - 1) ? LMASS[i] : M.m7;
        F[i] = ac * mi;
        f3 N;
        if (i < NJ - 1) N = al * LISO[i];
        else N = rot_inertia_mul(K, M.I7, al) + cross(w, rot_inertia_mul(K, M.I7, w));
        Nc[i] = N + cross(K.c[i], F[i]);        // moment about the base origin
        op = K.o[i];
    }
    D.w7 = w; D.al7 = al; D.a7 = a;
    // ---- backward pass: bias torques and composite-rigid-body mass matrix (inertia about the base origin) ----
    f3 vo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) vo[j] = cross(K.o[j], K.z[j]);
    f3 fa = mk(0, 0, 0), na = mk(0, 0, 0);
    float cm = 0.f; f3 ch = mk(0, 0, 0);
    float Io[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = NJ - 1; i >= 0; --i) {
        fa = fa + F[i]; na = na + Nc[i];
        D.bias[i] = dot(K.z[i], na - cross(K.o[i], fa));
        float mi = (i < NJ - 1) ? LMASS[i] : M.m7;
        f3 c = K.c[i];
        cm += mi; ch = madd(ch, c, mi);
        float cc = dot(c, c);
        if (i < NJ - 1) { Io[0] += LISO[i]; Io[3] += LISO[i]; Io[5] += LISO[i]; }
        else {
            // R7 I7 R7^T, six unique entries
            f3 e0 = rot_inertia_mul(K, M.I7, mk(1, 0, 0)), e1 = rot_inertia_mul(K, M.I7, mk(0, 1, 0)), e2 = rot_inertia_mul(K, M.I7, mk(0, 0, 1));
            Io[0] += e0.x; Io[1] += e0.y; Io[2] += e0.z; Io[3] += e1.y; Io[4] += e1.z; Io[5] += e2.z;
        }
        Io[0] += mi * (cc - c.x * c.x); Io[1] -= mi * c.x * c.y; Io[2] -= mi * c.x * c.z;
        Io[3] += mi * (cc - c.y * c.y); Io[4] -= mi * c.y * c.z; Io[5] += mi * (cc - c.z * c.z);
        f3 n = symmul(Io, K.z[i]) + cross(ch, vo[i]);
        f3 f = vo[i] * cm + cross(K.z[i], ch);
#pragma unroll
        for (int j = 0; j <= i; ++j) D.M[PK(i, j)] = dot(K.z[j], n) + dot(vo[j], f);
    }
}

// ------------------------------------------------------------------------------------------------------------
// quaternion helpers (src/utils/quaternion.py; robosuite transform_utils.mat2quat sign convention w >= 0)
// ------------------------------------------------------------------------------------------------------------
DI void mat2quat_xyzw(f3 cx, f3 cy, f3 cz, float* q) {
    // rotation matrix columns cx, cy, cz: m_rc = (column c).component r
    float m00 = cx.x, m10 = cx.y, m20 = cx.z, m01 = cy.x, m11 = cy.y, m21 = cy.z, m02 = cz.x, m12 = cz.y, m22 = cz.z;
    float tr = m00 + m11 + m22, w, x, y, z;
    if (tr > 0.f) { float s = sqrtf(tr + 1.f) * 2.f; w = 0.25f * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s; }
    else if (m00 > m11 && m00 > m22) { float s = sqrtf(1.f + m00 - m11 - m22) * 2.f; w = (m21 - m12) / s; x = 0.25f * s; y = (m01 + m10) / s; z = (m02 + m20) / s; }
    else if (m11 > m22) { float s = sqrtf(1.f + m11 - m00 - m22) * 2.f; w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25f * s; z = (m12 + m21) / s; }
    else { float s = sqrtf(1.f + m22 - m00 - m11) * 2.f; w = (m10 - m01) / s; x = (m02 + m20) / s; y = (m12 + m21) / s; z = 0.25f * s; }
    if (w < 0.f) { w = -w; x = -x; y = -y; z = -z; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
// transforms3d qmult(a, qconjugate(b)) with index 0 treated as the scalar part (quaternion.py:23-35)
DI void difference_quat(const float* a, const float* b, float* o) {
    float bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
    o[0] = a[0] * bw - a[1] * bx - a[2] * by - a[3] * bz;
    o[1] = a[0] * bx + a[1] * bw + a[2] * bz - a[3] * by;
    o[2] = a[0] * by - a[1] * bz + a[2] * bw + a[3] * bx;
    o[3] = a[0] * bz + a[1] * by - a[2] * bx + a[3] * bw;
}
// distance_quat(q, goal) (quaternion.py:38-59) for a unit quaternion q (w,x,y,z) and the goal quaternion as written at
// ultrasound.py:174, whose norm is 1 - eps_g (eps_g = 1.2e-9).  The reference evaluates 2 arccos(clip(w)) with
// w = q . g and folds distances above pi to |2 pi - d|, i.e. d = 2 arccos(|w|).  arccos loses half the mantissa near
// |w| = 1 (where the probe spends its life), so 1 - |w| is formed without cancellation from the chord to the unit goal
// g^:  q . g^ = 1 - |q - g^|^2 / 2  =>  1 - |w| = eps_g + (1 - eps_g) min(|q - g^|^2, |q + g^|^2) / 2,
// and d = 4 arcsin(sqrt((1 - |w|) / 2)).  Same value as the reference formula, accurate to fp32 rounding.
DI float distance_quat_goal(const float* q, const float* ghat, float eps_g) {
    float dm = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { float a = q[i] - ghat[i], b = q[i] + ghat[i]; dm = fmaf(a, a, dm); dp = fmaf(b, b, dp); }
    float m = fminf(dm, dp);
    if (m == 0.f) return 0.f;                           // q_log: zero vector part (quaternion.py:17-18)
    float h = eps_g + 0.5f * m * (1.f - eps_g);
    return 4.f * asinf(sqrtf(fminf(0.5f * h, 1.f)));
}

// closest point of the segment p1 + s d1 (s in [0,1]) to the point c
DI f3 seg_point(f3 p1, f3 d1, f3 c) {
    float s = clampf(dot(d1, c - p1) / dot(d1, d1), 0.f, 1.f);
    return madd(p1, d1, s);
}

// ------------------------------------------------------------------------------------------------------------
// LDS plan (words per lane; rows of 64 lanes).  TOP torso only.
//   S   [0,99)        slide coordinates s[e]           (dead after collision -> contact coupling matrix Kc[8][8])
//   Y   [99,198)      sdot[e] during the rhs build, then the lattice acceleration a~[e] (live to the end)
//   X   [198,534)     rhs[e] for the lattice solve (first 99 words), then the contact scratch (8 slots x 42 words)
// ------------------------------------------------------------------------------------------------------------
constexpr int L_S = 0, L_Y = N_TOP, L_X = 2 * N_TOP;
constexpr int CS_WORDS = 42;       // per contact slot: n3 t3 r3 Liw18 g3 aref3 Ad3 Rn1 f3 (ae, elem in registers)
constexpr int CS_N = 0, CS_T = 3, CS_R = 6, CS_LIW = 9, CS_G = 27, CS_AREF = 30, CS_AD = 33, CS_RN = 36, CS_F = 37, CS_E = 40;
constexpr int LDS_WORDS_TOP = 2 * N_TOP + MAXC * CS_WORDS;   // 534 words/lane = 136704 B per workgroup
#define LDSW(base, idx) lds[((base) + (idx)) * WG + lane]

struct StepOut {               // results of one forward pass that the env logic needs
    float fc[3];               // net contact force on the probe (cfrc_ext[probe][3:6])
    float tq[3];               // torque sensor at ft_frame (site frame)
    int ncon;
    int con_shell[MAXC];
    int overflow;
};

template <int TORSO>
__global__ __launch_bounds__(WG) void usim_step_kernel(const DevModel M, const DevCfg C, float* __restrict__ st, int n, int npad,
                                                       const DevIO io, int flags, long long rstep) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int env = blockIdx.x * WG + lane;
    const bool valid = env < n;
    const int ei = valid ? env : n - 1;           // clamp so that every lane has something to read; stores are guarded
    const bool reset_only = (flags & LF_RESET_ONLY) != 0;
    const bool auto_reset = (flags & LF_AUTO_RESET) != 0;
#define ST(f) st[(size_t)(f) * npad + ei]
#define STI(f) (reinterpret_cast<int*>(st))[(size_t)(f) * npad + ei]

    // ---------------- load state ----------------
    float q[NJ], qd[NJ], q0[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) { q[i] = ST(F_Q + i); qd[i] = ST(F_QD + i); q0[i] = ST(F_Q0 + i); }
    f3 ts = mk(ST(F_TS), ST(F_TS + 1), ST(F_TS + 2)), te = mk(ST(F_TE), ST(F_TE + 1), ST(F_TE + 2));
    float u0 = ST(F_U0), vbar = ST(F_VBAR), fzbar = ST(F_FZBAR), fzprev = ST(F_FZPREV), dfz = ST(F_DFZ);
    float kst = ST(F_KST), kdmp = ST(F_KDMP), mu = ST(F_MU), epret = ST(F_EPRET);
    int t = STI(F_T), touched = STI(F_TOUCH), episode = STI(F_EPISODE), status = STI(F_STATUS);

    // ---------------- action ----------------
    float act[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!reset_only) {
        if (flags & LF_RANDOM_ACT) {
            uint32_t gid = (uint32_t)(C.env_offset + ei);
            u4 r1 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 1u, C.key0, C.key1);
            u4 r2 = philox(gid, (uint32_t)rstep, (uint32_t)((unsigned long long)rstep >> 32), 2u, C.key0, C.key1);
            uint32_t rr[8] = {r1.a, r1.b, r1.c, r1.d, r2.a, r2.b, r2.c, r2.d};
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                float u = u01(rr[a]);
                bool sgn = (C.mode == 1) || (C.mode == 2 && a == 6);
                act[a] = sgn ? 2.f * u - 1.f : u;
                if (io.act_out && valid && a < C.adim) io.act_out[(size_t)ei * C.adim + a] = act[a];
            }
        } else {
#pragma unroll
            for (int a = 0; a < 7; ++a) if (a < C.adim) act[a] = io.act[(size_t)ei * C.adim + a];
        }
    }

    bool need = reset_only ? (io.mask ? io.mask[ei] != 0 : true) : false;   // lanes that (re)initialise in pass 1
    bool done = false;
    const float dt = C.dt;

#pragma nounroll
    for (int pass = reset_only ? 1 : 0; pass < 2; ++pass) {
        const bool active = (pass == 0) ? true : need;
        if (pass == 1) {
            if (!__any(need)) break;
            if (need) {
                // ================= reset draws (ultrasound.py:416-478) =================
                episode += 1;
                uint32_t gid = (uint32_t)(C.env_offset + ei);
                u4 A = philox(gid, (uint32_t)episode, 0u, 0u, C.key0, C.key1);
                u4 B = philox(gid, (uint32_t)episode, 1u, 0u, C.key0, C.key1);
                u4 Cc = philox(gid, (uint32_t)episode, 2u, 0u, C.key0, C.key1);
                const float tz = M.torso[2] + M.base[2] + 0.039f;          // ultrasound.py:184,807
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
                        const float ys = -0.09f + ty, ystep = 0.18f / 49.f;
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
                }
                // ================= initial pose: damped-least-squares IK from init_qpos (ultrasound.py:812-844) ==========
                float uu = clampf(u0, 0.f, 1.f);
                f3 tp0 = ts + (te - ts) * uu;
                f3 target = mk(tp0.x + noise.x + 0.0028f - M.base[0], tp0.y + noise.y + 0.0008f - M.base[1], tp0.z + noise.z + 0.0066f - M.base[2]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) q[i] = INITQ[i];
                for (int it = 0; it < C.ik_iters; ++it) {
                    Kin K; fk(M, q, K);
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
                for (int i = 0; i < NJ; ++i) { q0[i] = q[i]; qd[i] = 0.f; }
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
            Kin K; fk(M, q, K);
            Dyn D; dynamics(M, K, qd, D);
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
            // B = M^-1 J^T (column a = M^-1 J[a]), Li = J B  (6x6, packed lower) = Lambda^-1
            float Bm[6][NJ];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) Bm[a][j] = J[a][j];
                chol_solve<NJ>(Lm, idm, Bm[a]);
            }
            float Li[21];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf(J[a][j], Bm[b][j], s);
                    Li[PK(a, b)] = s;
                }
            // ---------------- OSC_POSE torque (robosuite osc.py run_controller; rl_config.yaml:33-51) ----------------
            float tau[NJ];
            if (pass == 0) {
                float kp[6], kd[6];
                f3 gpos, gx, gy, gz;
                float up = clampf((float)(t - 1) / (float)C.horizon + u0, 0.f, 1.f);   // controller.traj_pos from the previous _post_action
                f3 tpw = ts + (te - ts) * up;
                if (C.mode == 1) {
                    float d[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a) d[a] = clampf(act[a], -1.f, 1.f) * (a < 3 ? C.out_pos : C.out_ori);
                    gpos = K.x + mk(d[0], d[1], d[2]);
                    float ang = sqrtf(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                    if (ang < 1e-12f) { gx = K.sx; gy = K.sy; gz = K.sz; }
                    else {
                        float hh = 0.5f * ang, sh = sinf(hh) / ang, qw = cosf(hh), qx = d[3] * sh, qy = d[4] * sh, qz = d[5] * sh;
                        // rotation matrix of the delta quaternion, applied on the left of the current orientation
                        f3 e0 = mk(1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy + qw * qz), 2.f * (qx * qz - qw * qy));
                        f3 e1 = mk(2.f * (qx * qy - qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz + qw * qx));
                        f3 e2 = mk(2.f * (qx * qz + qw * qy), 2.f * (qy * qz - qw * qx), 1.f - 2.f * (qx * qx + qy * qy));
                        gx = e0 * K.sx.x + e1 * K.sx.y + e2 * K.sx.z;
                        gy = e0 * K.sy.x + e1 * K.sy.y + e2 * K.sy.z;
                        gz = e0 * K.sz.x + e1 * K.sz.y + e2 * K.sz.z;
                    }
#pragma unroll
                    for (int a = 0; a < 6; ++a) { kp[a] = C.kp_fixed; kd[a] = 2.f * sqrtf(C.kp_fixed) * C.damping_ratio; }
                } else {
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        float v = clampf(act[a], 0.f, 1.f);
                        kp[a] = C.kp_min + v * (C.kp_max - C.kp_min);
                        kd[a] = 2.f * sqrtf(kp[a]) * C.damping_ratio;
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
                    for (int j = 0; j < NJ; ++j) s = fmaf(Bm[a][j], y[j], s);
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
            // ---------------- smooth acceleration ----------------
            float qs[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) qs[i] = tau[i] - D.bias[i] - JOINT_DAMP * qd[i];
            chol_solve<NJ>(Lm, idm, qs);

            float W[6] = {0, 0, 0, 0, 0, 0};          // site-space wrench of the contact forces
            if (TORSO) {
                // prescribed torso base motion: free fall over the 4.7 mm spawn gap, then rest
                const int tsim = (t > 0) ? t - 1 : 0;
                float dz = -TORSO_DROP, vz = 0.f, az = 0.f;
                if (C.torso_drop) {
                    float tt = (float)tsim * dt, zf = -0.5f * GRAV * tt * tt;
                    if (zf > -TORSO_DROP) { dz = zf; vz = -GRAV * tt; az = -GRAV; }
                }
                // ---- stage s, sdot (coalesced rows) ----
                if (pass == 0) {
                    for (int e = 0; e < N_TOP; ++e) { LDSW(L_S, e) = ST(F_S + e); LDSW(L_Y, e) = ST(F_SD + e); }
                } else {
                    for (int e = 0; e < N_TOP; ++e) { LDSW(L_S, e) = 0.f; LDSW(L_Y, e) = 0.f; }
                }
                // ---- lattice right-hand side: a_s + w_fix aref_fix + w_ten sum_j aref_ij ----
                const float kfix = 1.0f / (SI_DMAX * SR_TC * SR_TC), bfix = 2.0f / (SI_DMAX * SR_TC);
                const float kten = kst / SI_DMAX, bten = kdmp / SI_DMAX;
                for (int e = 0; e < N_TOP; ++e) {
                    float se = LDSW(L_S, e), sde = LDSW(L_Y, e);
                    float r = -(GRAV + az) * c_el_axis[3 * e + 2] + M.wfix * (-bfix * sde - kfix * se);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        int j = c_el_nbr[4 * e + d];
                        if (j >= -1) {
                            float sj = (j >= 0) ? LDSW(L_S, j) : 0.f, sdj = (j >= 0) ? LDSW(L_Y, j) : 0.f;
                            r += M.wten * (-bten * (sde - sdj) - kten * (se - sj));
                        }
                    }
                    LDSW(L_X, e) = r;
                }
                // ---- a~ = Linv * rhs: 13 row blocks of 8, coefficients wave-uniform (scalar loads) ----
                for (int blk = 0; blk < LINV_NBLK; ++blk) {
                    float acc[LINV_BLK] = {0, 0, 0, 0, 0, 0, 0, 0};
                    const float* lb = c_linv_blk + blk * N_TOP * LINV_BLK;
                    for (int j = 0; j < N_TOP; ++j) {
                        float xj = LDSW(L_X, j);
#pragma unroll
                        for (int k = 0; k < LINV_BLK; ++k) acc[k] = fmaf(lb[j * LINV_BLK + k], xj, acc[k]);
                    }
#pragma unroll
                    for (int k = 0; k < LINV_BLK; ++k) if (blk * LINV_BLK + k < N_TOP) LDSW(L_Y, blk * LINV_BLK + k) = acc[k];
                }
                // ---- collision: probe capsule vs the 99 element capsules, ascending shell id ----
                f3 cc = K.x - K.sz * C.probe_r;                       // capsule centre one radius behind the tip
                f3 p1 = cc - K.sy * C.probe_hl, d1 = K.sy * (2.f * C.probe_hl);
                int nc = 0;
                int cel[MAXC];
                float cdist[MAXC];
#pragma unroll
                for (int k = 0; k < MAXC; ++k) { cel[k] = 0; cdist[k] = 0.f; }
                for (int e = 0; e < N_TOP; ++e) {
                    f3 ax = mk(c_el_axis[3 * e], c_el_axis[3 * e + 1], c_el_axis[3 * e + 2]);
                    float se = LDSW(L_S, e);
                    f3 tip = mk(M.torso[0] + c_el_pos[3 * e], M.torso[1] + c_el_pos[3 * e + 1], M.torso[2] + c_el_pos[3 * e + 2] + dz) + ax * (se - ELEM_R);
                    // element collision geometry = the cap sphere (centre `tip`, radius ELEM_R); DESIGN.md section 2
                    f3 c2 = tip, c1 = seg_point(p1, d1, tip);
                    f3 dd = c1 - c2;
                    float len = sqrtf(dot(dd, dd)), dist = len - (C.probe_r + ELEM_R);
                    if (dist < 0.f) {
                        if (nc < MAXC) {
                            f3 nn = (len > 1e-9f) ? dd * (1.f / len) : mk(0, 0, 1);
                            f3 pc = c2 + nn * (ELEM_R + 0.5f * dist);
                            f3 rr = pc - K.x;
                            f3 ref = (fabsf(nn.x) > 0.9f) ? mk(0, 1, 0) : mk(1, 0, 0);
                            f3 t1 = cross(nn, ref); t1 = t1 * (1.f / sqrtf(dot(t1, t1)));
                            const int b = L_X + nc * CS_WORDS;
                            LDSW(b, CS_N) = nn.x; LDSW(b, CS_N + 1) = nn.y; LDSW(b, CS_N + 2) = nn.z;
                            LDSW(b, CS_T) = t1.x; LDSW(b, CS_T + 1) = t1.y; LDSW(b, CS_T + 2) = t1.z;
                            LDSW(b, CS_R) = rr.x; LDSW(b, CS_R + 1) = rr.y; LDSW(b, CS_R + 2) = rr.z;
                            LDSW(b, CS_E) = __int_as_float(e);
                            LDSW(b, CS_E + 1) = dist;
#pragma unroll
                            for (int k = 0; k < MAXC; ++k) if (k == nc) { cel[k] = e; cdist[k] = dist; }
                            nc += 1;
                        } else R.overflow = 1;
                    }
                }
                R.ncon = nc;
                int ncmax = 0;                                       // wave-uniform bound on the contact count
#pragma unroll
                for (int k = MAXC; k >= 1; --k) if (ncmax == 0 && __any(nc >= k)) ncmax = k;
                float gf[MAXC];
#pragma unroll
                for (int k = 0; k < MAXC; ++k) gf[k] = 0.f;
                if (ncmax > 0) {
                    // contact coupling through the lattice: Kc[c2][k] = Linv[e_c2][e_k] / m   (S region is dead now)
#pragma unroll
                    for (int k = 0; k < MAXC; ++k)
#pragma unroll
                        for (int c2 = 0; c2 < MAXC; ++c2)
                            if (k < ncmax && c2 < ncmax) LDSW(L_S, k * MAXC + c2) = (k < nc && c2 < nc) ? c_linv[cel[c2] * N_TOP + cel[k]] * (1.0f / ELEM_MASS) : 0.f;
                    float alpha[6], vs[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        float s = 0.f, u = 0.f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) { s = fmaf(J[a][j], qs[j], s); u = fmaf(J[a][j], qd[j], u); }
                        alpha[a] = s; vs[a] = u;
                    }
                    float ae[MAXC];
                    const float bcon = 2.0f / (SI_DMAX * SR_TC);
                    // ---- per-contact row data ----
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        ae[k] = 0.f;
                        if (k < ncmax && k < nc) {
                            const int b = L_X + k * CS_WORDS;
                            const bool on = true;
                            f3 nn = mk(LDSW(b, CS_N), LDSW(b, CS_N + 1), LDSW(b, CS_N + 2));
                            f3 t1 = mk(LDSW(b, CS_T), LDSW(b, CS_T + 1), LDSW(b, CS_T + 2));
                            f3 rr = mk(LDSW(b, CS_R), LDSW(b, CS_R + 1), LDSW(b, CS_R + 2));
                            f3 t2 = cross(nn, t1);
                            const int e = on ? cel[k] : 0;
                            f3 ax = mk(c_el_axis[3 * e], c_el_axis[3 * e + 1], c_el_axis[3 * e + 2]);
                            float sde = on ? ST(F_SD + e) : 0.f;
                            if (pass == 1) sde = 0.f;
                            ae[k] = on ? LDSW(L_Y, e) : 0.f;
                            float dist = cdist[k];
                            float xx = fminf(-dist / SI_WIDTH, 1.f);
                            float yy = (xx < 0.5f) ? 2.f * xx * xx : 1.f - 2.f * (1.f - xx) * (1.f - xx);
                            float dimp = SI_D0 + yy * (SI_DMAX - SI_D0);
                            float kk = dimp / (SI_DMAX * SI_DMAX * SR_TC * SR_TC);
                            float Rn = (1.f - dimp) / dimp * M.invw;
                            float linv_ee = on ? LDSW(L_S, k * MAXC + k) : 0.f;      // Linv[e][e]/m
                            LDSW(b, CS_RN) = Rn;
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                f3 dir = (d == 0) ? nn : (d == 1 ? t1 : t2);
                                f3 rx = cross(rr, dir);
                                float w[6] = {dir.x, dir.y, dir.z, rx.x, rx.y, rx.z};
                                float g = -dot(dir, ax);
                                float vrel = g * sde - dir.z * vz;
                                float Aii = g * g * linv_ee;
#pragma unroll
                                for (int a = 0; a < 6; ++a) {
                                    float s = 0.f;
#pragma unroll
                                    for (int bb = 0; bb < 6; ++bb) s = fmaf((a >= bb) ? Li[PK(a, bb)] : Li[PK(bb, a)], w[bb], s);
                                    LDSW(b, CS_LIW + d * 6 + a) = s;
                                    Aii = fmaf(w[a], s, Aii);
                                    vrel = fmaf(w[a], vs[a], vrel);
                                }
                                LDSW(b, CS_G + d) = g;
                                LDSW(b, CS_AREF + d) = -bcon * vrel - (d == 0 ? kk * dist : 0.f);
                                LDSW(b, CS_AD + d) = Aii;
                                LDSW(b, CS_F + d) = 0.f;
                            }
                        }
                    }
                    // ---- projected Gauss-Seidel on the dual over the contact rows (fixed sweeps, cold start) ----
                    for (int it = 0; it < C.pgs_iters; ++it) {
#pragma unroll
                        for (int k = 0; k < MAXC; ++k) {
                            if (k < ncmax && k < nc) {
                                const int b = L_X + k * CS_WORDS;
                                const bool on = true;
                                f3 nn = mk(LDSW(b, CS_N), LDSW(b, CS_N + 1), LDSW(b, CS_N + 2));
                                f3 t1 = mk(LDSW(b, CS_T), LDSW(b, CS_T + 1), LDSW(b, CS_T + 2));
                                f3 rr = mk(LDSW(b, CS_R), LDSW(b, CS_R + 1), LDSW(b, CS_R + 2));
                                f3 t2 = cross(nn, t1);
                                float Rn = LDSW(b, CS_RN);
                                float f[3] = {LDSW(b, CS_F), LDSW(b, CS_F + 1), LDSW(b, CS_F + 2)};
                                float g[3] = {LDSW(b, CS_G), LDSW(b, CS_G + 1), LDSW(b, CS_G + 2)};
#pragma unroll
                                for (int d = 0; d < 3; ++d) {
                                    f3 dir = (d == 0) ? nn : (d == 1 ? t1 : t2);
                                    f3 rx = cross(rr, dir);
                                    float Rd = (d == 0) ? Rn : Rn * (1.0f / IMPRATIO);
                                    float res = g[d] * ae[k] - LDSW(b, CS_AREF + d) + Rd * f[d];
                                    res += dir.x * alpha[0] + dir.y * alpha[1] + dir.z * alpha[2] + rx.x * alpha[3] + rx.y * alpha[4] + rx.z * alpha[5];
                                    float fn = f[d] - res / (LDSW(b, CS_AD + d) + Rd);
                                    if (d == 0) fn = fmaxf(fn, 0.f);
                                    float df = on ? fn - f[d] : 0.f;
                                    f[d] += df;
#pragma unroll
                                    for (int a = 0; a < 6; ++a) alpha[a] = fmaf(LDSW(b, CS_LIW + d * 6 + a), df, alpha[a]);
                                    float gd = g[d] * df;
#pragma unroll
                                    for (int c2 = 0; c2 < MAXC; ++c2) if (c2 < ncmax) ae[c2] = fmaf(LDSW(L_S, k * MAXC + c2), gd, ae[c2]);
                                }
                                // elliptic cone: |f_t| <= mu f_n
                                float ft = sqrtf(f[1] * f[1] + f[2] * f[2]), lim = mu * f[0];
                                if (on && ft > lim) {
                                    float sc = (ft > 0.f) ? lim / ft : 0.f;
#pragma unroll
                                    for (int d = 1; d < 3; ++d) {
                                        float df = f[d] * sc - f[d];
                                        f[d] += df;
#pragma unroll
                                        for (int a = 0; a < 6; ++a) alpha[a] = fmaf(LDSW(b, CS_LIW + d * 6 + a), df, alpha[a]);
                                        float gd = g[d] * df;
#pragma unroll
                                        for (int c2 = 0; c2 < MAXC; ++c2) if (c2 < ncmax) ae[c2] = fmaf(LDSW(L_S, k * MAXC + c2), gd, ae[c2]);
                                    }
                                }
                                LDSW(b, CS_F) = f[0]; LDSW(b, CS_F + 1) = f[1]; LDSW(b, CS_F + 2) = f[2];
                            }
                        }
                    }
                    // ---- contact wrench on the site, force along each element axis ----
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        if (k < ncmax && k < nc) {
                            const int b = L_X + k * CS_WORDS;
                            const bool on = true;
                            f3 nn = mk(LDSW(b, CS_N), LDSW(b, CS_N + 1), LDSW(b, CS_N + 2));
                            f3 t1 = mk(LDSW(b, CS_T), LDSW(b, CS_T + 1), LDSW(b, CS_T + 2));
                            f3 rr = mk(LDSW(b, CS_R), LDSW(b, CS_R + 1), LDSW(b, CS_R + 2));
                            f3 t2 = cross(nn, t1);
                            float f0 = on ? LDSW(b, CS_F) : 0.f, f1 = on ? LDSW(b, CS_F + 1) : 0.f, f2 = on ? LDSW(b, CS_F + 2) : 0.f;
                            f3 Fw = nn * f0 + t1 * f1 + t2 * f2;
                            f3 Tw = cross(rr, Fw);
                            W[0] += Fw.x; W[1] += Fw.y; W[2] += Fw.z; W[3] += Tw.x; W[4] += Tw.y; W[5] += Tw.z;
                            gf[k] = LDSW(b, CS_G) * f0 + LDSW(b, CS_G + 1) * f1 + LDSW(b, CS_G + 2) * f2;
                            if (!on) gf[k] = 0.f;
                        }
                    }
                }
                // ---- element accelerations a = a~ + Linv[:, e_c] gf_c / m, semi-implicit Euler, write back ----
                if (pass == 0) {
                    for (int e = 0; e < N_TOP; ++e) {
                        float a = LDSW(L_Y, e);
#pragma unroll
                        for (int k = 0; k < MAXC; ++k)
                            if (k < ncmax && k < nc) a = fmaf(c_linv[cel[k] * N_TOP + e] * (1.0f / ELEM_MASS), gf[k], a);   // Linv symmetric: row cel[k]
                        float sdn = ST(F_SD + e) + dt * a;
                        float sn = ST(F_S + e) + dt * sdn;
                        if (valid) { ST(F_SD + e) = sdn; ST(F_S + e) = sn; }
                    }
                } else if (valid) {
                    for (int e = 0; e < N_TOP; ++e) { ST(F_SD + e) = 0.f; ST(F_S + e) = 0.f; }
                }
#pragma unroll
                for (int k = 0; k < MAXC; ++k) R.con_shell[k] = (k < nc) ? c_el_shell[cel[k]] : -1;
            } else {
#pragma unroll
                for (int k = 0; k < MAXC; ++k) R.con_shell[k] = -1;
            }
            // ---------------- constrained arm acceleration: qacc = qs + M^-1 J^T W ----------------
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                float s = qs[i];
#pragma unroll
                for (int a = 0; a < 6; ++a) s = fmaf(Bm[a][i], W[a], s);
                qacc[i] = s;
            }
            R.fc[0] = W[0]; R.fc[1] = W[1]; R.fc[2] = W[2];
            // ---------------- torque sensor at ft_frame (MuJoCo cfrc_int of the probe body, site frame) ----------------
            {
                f3 al = D.al7, a7 = D.a7;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    al = madd(al, K.z[j], qacc[j]);
                    a7 = madd(a7, cross(K.z[j], K.o[NJ - 1] - K.o[j]), qacc[j]);
                }
                f3 rc = K.r7x * M.pcom7[0] + K.r7y * M.pcom7[1] + K.r7z * M.pcom7[2];
                f3 ac = a7 + cross(al, rc) + cross(D.w7, cross(D.w7, rc));
                f3 N = rot_inertia_mul(K, M.pI7, al) + cross(D.w7, rot_inertia_mul(K, M.pI7, D.w7));
                f3 Fp = ac * PROBE_MASS;
                f3 tw = N + cross(K.o[NJ - 1] + rc - K.x, Fp) - mk(W[3], W[4], W[5]);
                R.tq[0] = dot(K.sx, tw); R.tq[1] = dot(K.sy, tw); R.tq[2] = dot(K.sz, tw);
            }
            // ---------------- integrate the arm: mj_Euler with implicit joint damping ----------------
            if (pass == 0) {
                float rhs[NJ], Ld[28], idd[NJ];
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s = fmaf((i >= j) ? D.M[PK(i, j)] : D.M[PK(j, i)], qacc[j], s);
                    rhs[i] = s;
                }
#pragma unroll
                for (int k = 0; k < 28; ++k) Ld[k] = D.M[k];
#pragma unroll
                for (int i = 0; i < NJ; ++i) Ld[PK(i, i)] += dt * JOINT_DAMP;
                chol_packed<NJ>(Ld, idd); chol_solve<NJ>(Ld, idd, rhs);
#pragma unroll
                for (int i = 0; i < NJ; ++i) { qd[i] = fmaf(dt, rhs[i], qd[i]); q[i] = fmaf(dt, qd[i], q[i]); }
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
            // ---------------- observation (ultrasound.py:363-401) ----------------
            {
                const int tprev = (pass == 0) ? t - 1 : 0;
                float up = clampf((float)tprev / (float)C.horizon + u0, 0.f, 1.f);
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
                    pos_err_norm = sqrtf(pe0 * pe0 + pe1 * pe1);
                    float pos_rew = 5.f * expf(-pos_err_norm);
                    float qc[4] = {qe[3], qe[0], qe[1], qe[2]};
                    ori_err = 0.2f * distance_quat_goal(qc, M.ghat, M.geps);
                    float ori_rew = expf(-ori_err);
                    float ve = 45.f * (vbar - 0.04f); ve *= ve;
                    float vel_rew = expf(-ve);
                    float fe = 0.7f * (fzbar - 5.f); fe *= fe;
                    float force_rew = contact ? 3.f * expf(-fe) : 0.f;
                    float de = 0.01f * dfz; de *= de;
                    float dforce_rew = contact ? 2.f * expf(-de) : 0.f;
                    float reward = pos_rew + ori_rew + vel_rew + force_rew + dforce_rew;
                    done = t >= C.horizon;
                    // ---------------- bookkeeping (ultrasound.py:528-546) ----------------
                    float hvn = sqrtf(dot(hv, hv));
                    vbar += (hvn - vbar) / (float)t;
                    float fz = R.fc[2];
                    dfz = (fz - fzprev) / dt;
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
                    if (R.overflow) status |= 1;
                    if (valid) {
                        io.rew[ei] = reward;
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
                if (valid && io.obs && (pass == 1 || !need)) {
#pragma unroll
                    for (int a = 0; a < OBS_DIM; ++a) io.obs[(size_t)ei * OBS_DIM + a] = obs[a];
                }
            }
        }
    }

    // ---------------- store state ----------------
    if (valid) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) { ST(F_Q + i) = q[i]; ST(F_QD + i) = qd[i]; ST(F_Q0 + i) = q0[i]; }
        ST(F_TS) = ts.x; ST(F_TS + 1) = ts.y; ST(F_TS + 2) = ts.z; ST(F_TE) = te.x; ST(F_TE + 1) = te.y; ST(F_TE + 2) = te.z;
        ST(F_U0) = u0; ST(F_VBAR) = vbar; ST(F_FZBAR) = fzbar; ST(F_FZPREV) = fzprev; ST(F_DFZ) = dfz;
        ST(F_KST) = kst; ST(F_KDMP) = kdmp; ST(F_MU) = mu; ST(F_EPRET) = epret;
        STI(F_T) = t; STI(F_TOUCH) = touched; STI(F_EPISODE) = episode; STI(F_STATUS) = status;
    }
#undef ST
#undef STI
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
        bool sgn = (C.mode == 1) || (C.mode == 2 && a == 6);
        act[(size_t)i * C.adim + a] = sgn ? 2.f * u - 1.f : u;
    }
}

// translational inverse weight of the probe at init_qpos: tr(Jv M^-1 Jv^T) / 3 (MuJoCo body_invweight0 analogue)
__global__ void usim_invweight_kernel(const DevModel M, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float q[NJ], qd[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) { q[i] = INITQ[i]; qd[i] = 0.f; }
    Kin K; fk(M, q, K);
    Dyn D; dynamics(M, K, qd, D);
    float idm[NJ];
    chol_packed<NJ>(D.M, idm);
    float tr = 0.f;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        float jt[NJ], y[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { f3 jv = cross(K.z[j], K.x - K.o[j]); jt[j] = (ax == 0) ? jv.x : (ax == 1 ? jv.y : jv.z); y[j] = jt[j]; }
        chol_solve<NJ>(D.M, idm, y);
#pragma unroll
        for (int j = 0; j < NJ; ++j) tr = fmaf(jt[j], y[j], tr);
    }
    out[0] = tr * (1.0f / 3.0f);
}

}  // namespace usim
