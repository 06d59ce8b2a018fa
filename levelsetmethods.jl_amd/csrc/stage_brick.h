// stage_brick.h — the RK stage of a narrow band, brick by brick (round 3).
//
// stage_tile (stage_kernel.h) marches a 32 x 8 tile through the planes of a listed brick with one thread per NODE OF THE
// TILE: a wave computes a plane when any of its 64 nodes belongs to the band, every lane at full price.  On a band that is
// a shell a few nodes thick, 25 % of the lanes that execute the stencil arithmetic are band nodes (profiles/r3/band_step:
// 41 M vector instructions per 768³ stage against 10 M for the band's 3.6 M nodes), and every plane of every brick pays
// the march's loads, LDS writes and barrier whether it holds a band node or not.
//
// Here a workgroup takes BZ = 8 (or 16) planes of a brick of the band's active-tile list and
//   1. issues its loads — the mask bytes of its nodes first, then the planes with their halo: (BZ + 2G) x (8 + 2G) rows of 40
//      elements, 16-byte loads from x0 - 4,
//   2. turns the mask bytes into a list of its band nodes while the values are in flight (wave scans of the per-thread
//      counts; x fastest), then writes the values to LDS in the field's storage type,
//   3. deals the list to its waves 64 nodes at a time: ONE LANE PER BAND NODE.  A lane reads its stencil from the LDS
//      brick through BrickView — the interface node_update expects of a NodeView, with the march axis as a third LDS
//      stride — so the arithmetic, and every bit of the result, is stage_tile's.
// Lanes are idle only in the last batch of a wave.  Plain variants of the FAST build (constant or ROTATION advection
// coefficient, constant speed / curvature coefficient, one output, terms in slot order) on the aligned layout; everything
// else stays with stage_tile.  LsmTuning::band_bricks = 0 (LSM_BAND_BRICKS) keeps the tiled stage for everything.
#pragma once
#include "stage_kernel.h"

namespace lsm {
namespace LSM_NS {

template <class LT>
struct LdsElems {              // a position in the LDS brick; reads widen exactly to double
    const LT* p;
    LSM_DEV double operator[](int i) const { return (double)p[i]; }
};
template <int W, int HW, class LT>
struct BrickView {
    LdsElems<LT> T0;
    double c;
    LSM_DEV double z(int k) const { return T0[k * HW]; }
    template <int D>
    LSM_DEV double at(int k) const { return T0[k * (D == 0 ? 1 : (D == 1 ? W : HW))]; }
    template <int A, int B>
    LSM_DEV double corner(int sa, int sb) const { return T0[sa * (A == 0 ? 1 : W) + sb * (B == 1 ? W : HW)]; }
};

#ifndef LSM_BRICK_PITCH
#define LSM_BRICK_PITCH 44
#endif
struct BrickCfg {
    static constexpr int TX = 32, TY = 8, XL = 4;     // XL: elements in front of the tile's first node (>= G, rows start 16-byte aligned)
    static constexpr int W = TX + 2 * XL;             // elements of a row that are loaded
    // Row pitch of the LDS brick.  A lane reads ITS band node's stencil: the 32 lanes of a half-wave are consecutive entries of the
    // brick's node list (x fastest), i.e. runs of 8-12 nodes of three or four successive rows where the band's normal has an x
    // component — and a run of row r + 1 lands on the banks of row r's run shifted by the pitch mod 32 (32 banks of 4 bytes for
    // ds_read_b32, 64 for ds_read_b64 with 8-byte elements: the same in elements).  44 (= 12 mod 32) keeps runs of up to 12 nodes of
    // three successive rows apart where 40 (= 8) stacks them; measured −0.5 % per step, and SQ_LDS_BANK_CONFLICT unchanged at 1.3-1.4
    // cycles per LDS-active cycle: 32 scattered addresses on 32 banks always find a pair, whatever the pitch.
    static constexpr int WP = LSM_BRICK_PITCH;
    static_assert(WP >= W && WP % 4 == 0, "rows hold the loaded elements and start on 16-byte boundaries");
};

// NT threads take BZ planes of a listed brick (StageArgs::brick_list: the band's active tiles, one brick of `mc` planes each;
// ceil(mc / BZ) workgroups per brick).  float fields: 256 threads x 8 planes — 35 KB of LDS, four workgroups per CU in
// different phases (copy / list / arithmetic) — measured against 512 x 16 (57 KB, two per CU).  fp64 fields: 512 x 8 (67 KB, two per CU;
// 256 x 4 — 47 KB, three per CU, 2.5-fold halo — measured slower: 0.665 against 0.642 ms per 768³ step).
template <int ADV, int NM, int CURV, int EIK, class ST, int AK, int NT, int BZ>
__global__ void __launch_bounds__(NT) brick_kernel(const StageArgs a, const unsigned sub_per) {
    constexpr int NDIM = 3, TX = BrickCfg::TX, TY = BrickCfg::TY, XL = BrickCfg::XL, W = BrickCfg::W, WP = BrickCfg::WP;
    constexpr int G = halo_of(ADV, NM, CURV, EIK);
    constexpr int SEG = 16 / (int)sizeof(ST), NSEG = W / SEG, PSEG = WP / SEG;      // 16-byte segments of a row: loaded / pitch
    constexpr int H = TY + 2 * G, HW = H * WP, D = BZ + 2 * G;
    constexpr int NROW = D * H, NV4 = NROW * PSEG;
    constexpr int BPT = TX * TY * BZ / NT;                // mask bytes per thread: 8 or 4
    static_assert(XL >= G && XL % SEG == 0 && W % SEG == 0 && (BPT == 8 || BPT == 4) && NT % 64 == 0, "");
    __shared__ lsm_v4u vbrick[NV4];
    __shared__ unsigned short nodes[TX * TY * BZ];
    __shared__ int wsum[NT / 64];
    // gfx950 has 160 KB of LDS per CU (64 KB per workgroup on its predecessors: the fp64 instance would not launch there);
    // two workgroups of the largest instance must fit a CU
    static_assert(2 * (sizeof(lsm_v4u) * NV4 + sizeof(unsigned short) * TX * TY * BZ + 64) <= 160 * 1024, "brick kernel: LDS budget of a gfx950 CU");
    const ST* brick = reinterpret_cast<const ST*>(vbrick);

    // the launch's workgroups in list order, dealt to the XCDs in contiguous ranges (TileOrder): the BZ-plane parts of a brick
    // and the bricks of a neighbourhood meet in one L2
    const unsigned nwg = a.nbrick_list * sub_per, per = (nwg + 7u) / 8u;
    const unsigned wg = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (blockIdx.x / 8u >= per || wg >= nwg) return;
    const unsigned tile_id = (unsigned)a.brick_list[wg / sub_per], sub = wg % sub_per;
    const unsigned tbx = tile_id % a.nb[0], tby = (tile_id / a.nb[0]) % a.nb[1], tbm = tile_id / (a.nb[0] * a.nb[1]);
    const int bx0 = tbx * TX, by0 = tby * TY;
    const int nx = a.n[0], ny = a.n[1], nm = a.n[2];
    const long long sy = a.s1, sm = a.s2;
    const int m0 = a.mb + (int)tbm * a.mc;
    const int m1 = m0 + a.mc < a.me ? m0 + a.mc : a.me;
    const int zb = m0 + (int)sub * BZ;
    if (zb >= m1) return;
    const int nz = m1 - zb < BZ ? m1 - zb : BZ;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- 1. loads: the mask bytes first (vector loads return in order: the node list is built while the brick is in flight),
    //         then the brick with its halo, row by row: RPI rows of NSEG 16-byte segments per round of the workgroup
    unsigned long long mbytes;
    const int mrow = tid / (TX / BPT), mq = tid % (TX / BPT), mzr = mrow / TY, myr = mrow % TY;   // NT x BPT mask bytes = the 32 x 8 x BZ nodes
    {
        const unsigned char* mb_ = uniform_ptr(a.mask + (a.origin + (long long)zb * sm + (long long)by0 * sy + bx0));
        const bool rv = mzr < nz && by0 + myr < ny;
        const unsigned moff = rv ? (unsigned)mzr * (unsigned)sm + (unsigned)myr * (unsigned)sy + (unsigned)(mq * BPT) : LSM_OOB_OFFSET;
        if constexpr (BPT == 8) mbytes = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(plane_rsrc(mb_), moff, 0, 0));
        else mbytes = __builtin_amdgcn_raw_buffer_load_b32(plane_rsrc(mb_), moff, 0, 0);
    }
    constexpr int RPI = NT / NSEG, NIT = (NROW + RPI - 1) / RPI, QZ = RPI / H, RY = RPI % H;
    lsm_v4u v[NIT];
    const int crow = tid / NSEG, cseg = tid - crow * NSEG;
    {
        const ST* cb = uniform_ptr(reinterpret_cast<const ST*>(a.psi) + (a.origin + (long long)(zb - G) * sm + (long long)(by0 - G) * sy + (bx0 - XL)));
        const bool cv = crow < RPI && bx0 - XL + cseg * SEG <= nx + G - 1;     // beyond: nothing a band node reads (zeros, no access)
        int zr = crow / H, yr = crow - zr * H;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int Y = by0 - G + yr, Z = zb - G + zr;
            Y = Y > ny + G - 1 ? ny + G - 1 : Y;
            Z = Z > nm + G - 1 ? nm + G - 1 : Z;
            const unsigned eo = (unsigned)(Z - (zb - G)) * (unsigned)sm + (unsigned)(Y - (by0 - G)) * (unsigned)sy + (unsigned)(cseg * SEG);
            const unsigned off = (cv && zr < D) ? (unsigned)sizeof(ST) * eo : LSM_OOB_OFFSET;
            v[k] = __builtin_bit_cast(lsm_v4u, __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc(cb), off, 0, 0));
            yr += RY; zr += QZ;
            if (yr >= H) { yr -= H; ++zr; }
        }
    }
    // ---- 2. the band nodes of the brick, x fastest
    int N;
    {
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j)
            if (((mbytes >> (8 * j)) & 0xffull) != 0 && bx0 + mq * BPT + j < nx) bits |= 1u << j;
        const int cnt = __builtin_popcount(bits);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            const int s = wsum[w];
            base += w < wv ? s : 0;
            tot += s;
        }
        N = tot;
        int pos = base + incl - cnt;
        const unsigned code0 = ((unsigned)mzr << 8) | ((unsigned)myr << 5) | (unsigned)(mq * BPT);
        while (bits) {
            const int j = __builtin_ctz(bits);
            bits &= bits - 1;
            nodes[pos++] = (unsigned short)(code0 + j);
        }
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k)
        if (crow < RPI && crow + k * RPI < NROW) vbrick[(crow + k * RPI) * PSEG + cseg] = v[k];
    __syncthreads();
    // ---- 3. one lane per band node
    const long long po = a.origin + (long long)zb * sm + (long long)by0 * sy + bx0;     // the brick's first node
    PlaneTab pt;
#pragma unroll
    for (int k = 0; k < 3; ++k) pt.adv[k] = pt.nm[k] = pt.curv[k] = 1.0;               // no SEPARABLE coefficients here
    for (int b0 = wv * 64; b0 < N; b0 += NT) {
        const int i = b0 + lane;
        const bool on = i < N;
        const unsigned code = nodes[on ? i : N - 1];
        const int x = code & 31, y = (code >> 5) & 7, z = code >> 8;
        const int l = ((z + G) * H + (y + G)) * WP + x + XL;
        const unsigned eo = (unsigned)z * (unsigned)sm + (unsigned)y * (unsigned)sy + (unsigned)x;
        const NodeIO io{po, (unsigned)sizeof(ST) * eo, 8u * eo, zb + z + a.goff[2]};
        double pre_adv[3] = {0, 0, 0}, pre_nm[3] = {0, 0, 0}, pre_curv[3] = {0, 0, 0};
        const int gi0 = bx0 + x + a.goff[0], gi1 = by0 + y + a.goff[1];
        if constexpr (ADV != 0) coeff_prep<NDIM, NDIM, ADV_SCALED, AK>(a.adv, a, gi0, gi1, pre_adv);
        if constexpr (NM != 0) coeff_prep<NDIM, 1, false, LSM_COEFF_CONST>(a.nm, a, gi0, gi1, pre_nm);
        if constexpr (CURV != 0) coeff_prep<NDIM, 1, false, LSM_COEFF_CONST>(a.curv, a, gi0, gi1, pre_curv);
        NodeOps op;
        node_operands<NDIM, ADV, NM, CURV, EIK, ST, AK>(a, io, pre_adv, pre_nm, pre_curv, pt, op);
        const BrickView<WP, HW, ST> nv{{brick + l}, (double)brick[l]};
        double r1 = 0.0, r2 = 0.0;
        node_update<NDIM, ADV, NM, CURV, EIK, G, WP, ST, true>(a, nv, op, r1, r2);
        node_store<ST, true>(a, io, on, r1, r2);
    }
}

// is this launch a case for the brick kernel?  (adv / nm / curv: the pass's combination)
inline bool bricks_applicable(int ADV, int NM, int CURV, const StageArgs& a) {
    if (!a.tune->band_bricks) return false;                          // LsmTuning: the tiled band stage
    if (!a.mask || !a.brick_list || a.mc <= 0 || a.nbrick_list == 0 || a.out2 || !a.natural || a.xredirect || a.yredirect) return false;
    if ((NM && a.nm.kind != LSM_COEFF_CONST) || (CURV && a.curv.kind != LSM_COEFF_CONST)) return false;
    const int ak = ADV ? a.adv.kind : (int)LSM_COEFF_CONST;
    if (ak != LSM_COEFF_CONST && ak != LSM_COEFF_ROTATION) return false;
    if (a.me <= a.mb || a.tune->stage_generic) return false;
    // 16-byte rows: the aligned layout (lsm_create), 16-byte aligned arrays; 32-bit byte offsets inside a pass
    const long long seg = a.f32 ? 4 : 2, lead = a.origin - LSM_GHOST * a.s2 - LSM_GHOST * a.s1;
    if (lead < BrickCfg::XL || lead % seg || a.s1 % 8 || a.s2 % 8 || a.origin % 8) return false;
    if (((unsigned long long)a.psi | (unsigned long long)a.mask) % 16ull) return false;
    if ((long long)(16 + 2 * LSM_GHOST + 1) * a.s2 * 8 >= (1ll << 31)) return false;
    return true;
}

// 0 = launched, -1 = not a case for the brick kernel (the caller goes on to stage_tile)
template <int ADV, int NM, int CURV, int EIK>
int launch_bricks(const StageArgs& a, hipStream_t s) {
    if (!bricks_applicable(ADV, NM, CURV, a)) return -1;
    const int ak = ADV ? a.adv.kind : (int)LSM_COEFF_CONST;
    StageArgs b = a;
    b.nb[0] = (a.n[0] + BrickCfg::TX - 1) / BrickCfg::TX;
    b.nb[1] = (a.n[1] + BrickCfg::TY - 1) / BrickCfg::TY;
    b.nb[2] = (a.me - a.mb + a.mc - 1) / a.mc;
    b.nbig = 0; b.mc_tail = 0; b.tail_wgs = 0;
    // 8 planes per workgroup: 256 threads for float fields (35 KB of LDS, four workgroups per CU in different phases), 512 for fp64
    // (67 KB, two per CU).  Measured against it: 512 threads x 16 planes for float (0.559 against 0.537-0.544 ms per 768³ step),
    // 256 x 4 for fp64 (0.665 against 0.642).
    constexpr int bz = 8;
    const unsigned sub_per = (unsigned)((a.mc + bz - 1) / bz);
    const unsigned nwg = a.nbrick_list * sub_per;
    const dim3 grid(((nwg + 7u) / 8u) * 8u);
#define LSM_BRICK(STT, AKK, NTT, BZZ) hipLaunchKernelGGL((brick_kernel<ADV, NM, CURV, EIK, STT, AKK, NTT, BZZ>), grid, dim3(NTT), 0, s, b, sub_per)
#define LSM_BRICK_ST(AKK) do { if (!b.f32) LSM_BRICK(double, AKK, 512, 8); else LSM_BRICK(float, AKK, 256, 8); } while (0)
    if constexpr (ADV != 0) {
        if (ak == LSM_COEFF_ROTATION) { LSM_BRICK_ST(LSM_COEFF_ROTATION); return hipGetLastError() == hipSuccess ? 0 : -1; }
    }
    LSM_BRICK_ST(LSM_COEFF_CONST);
#undef LSM_BRICK_ST
#undef LSM_BRICK
    return hipGetLastError() == hipSuccess ? 0 : -1;     // a launch the device refuses (LDS, grid) goes to the tiled kernel
}

}  // namespace LSM_NS
}  // namespace lsm
