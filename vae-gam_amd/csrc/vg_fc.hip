// Fully connected layers of the encoder head and the decoder stem (vae_reg_GP.py:197-210, 243-259) on the matrix cores.
//
// Twelve tiny layers (3072->200->100->3x50->3x32, 41->50->100->200->3840) sit in the middle of the step's forward and backward chains:
// as library GEMMs + ATen glue they were ~32 GEMM / GEMV launches and ~15 elementwise ones per step, 5-16 us each with nothing beside
// them (0.13 ms forward, 0.33 ms backward of a 7 ms step: rocprofv3 kernel trace, profiles/round3_*).  One kernel covers all of them:
//
//   C[m][n] (+)= epilogue( sum_k A(m,k) * B(k,n) )      A, B addressed by element strides, one of each pair = 1
//
//   operand options   A <- max(A, 0)                     the layer input is stored as a pre-activation (conv5's output)
//                     A <- A * [amask > 0]               ReLU backward of the incoming gradient, folded into both of its consumers
//                     B <- max(B, 0)
//                     B gets one more column of ones     the bias gradient = the extra output column of the weight-gradient product
//   epilogue options  + bias[n], ReLU, * [cmask > 0], accumulate into C (the .grad views of the flat gradient buffer)
//   split-K           long reductions onto few output tiles (fc1: K = 3072 -> 64 x 200; fc8's data gradient: K = 3840 -> 576 x 200) are
//                     cut into pieces whose partial tiles go to a workspace; vg_fc_reduce_k sums them in a fixed order (no atomics:
//                     results do not depend on scheduling) and applies the epilogue
//   batch             the three 50 -> 32 heads as one launch (strides between the products)
//
//   jobs              up to 4 independent products in ONE launch (a layer's data gradient and weight gradient; blocks are dealt to the
//                     jobs by block index)
//
// These products are LATENCY-bound, not bandwidth- or MFMA-bound: a block's whole reduction is a handful of global load -> LDS -> MFMA
// round trips (first version, one step in flight: fc8's weight gradient 55 us, the small layers 15-20 us each, 0.26 ms for the backward
// chain).  So: tile 32x32 (4 waves x one MFMA tile of v_mfma_f32_16x16x4_f32, exact fp32; thousands of blocks, several per CU), K in
// steps of 64, and TWO steps of both operands in flight in registers.
// Both operands are staged K-CONTIGUOUS in LDS (As[m][k], Bs[n][k], pitch 68 floats) whatever their layout in memory, so a lane
// fetches the operands of four MFMAs with one 16-byte LDS read (8 consecutive lanes cover the 32 banks); the reduction index a lane
// feeds to MFMA s is k = 16*blk + 4*(lane>>4) + s for A and B alike (any pairing of the k's is a valid dot product).  A source that is
// contiguous along k is read as 16-byte loads, one that is contiguous along m / n as 4 coalesced dword loads (lanes along m / n) that
// become one 16-byte LDS write.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

constexpr int FC_TK = 64;
constexpr int FC_PITCH = FC_TK + 4;
constexpr int FC_DEPTH = 2;              // k-steps in flight in registers
constexpr int FC_MAXJOBS = 4;

struct FcArgs {
    vg_fc_desc d;
    const float* A; const float* amask; const float* B; const float* bias; const float* cmask;
    float* C; float* cx; float* ws;
    int kchunk;          // reduction indices per split (a multiple of FC_TK); K when ksplit == 1
    int a_vec, b_vec;    // load form of the operand (see fc_body): 0 dwords, 1 16 bytes along k, 2 16 bytes along m / n
    int gx, gy;          // tiles along m and n
    int big;             // 64 x 64 tiles (2 x 2 MFMA tiles per wave) instead of 32 x 32
};
struct FcJobs { int n; int first[FC_MAXJOBS + 1]; FcArgs job[FC_MAXJOBS]; };

__device__ __forceinline__ void fc_store(const FcArgs& p, int z, int m, int n, float v) {
    const vg_fc_desc& d = p.d;
    const int NW = d.N + ((d.flags & VG_FC_B_ONES) ? 1 : 0);
    if (m >= d.M || n >= NW) return;
    if (d.ksplit > 1) { p.ws[((size_t)z * d.M + m) * NW + n] = v; return; }
    if (n == d.N) {                                        // the ones column: row sums of A -> the bias gradient
        float* q = p.cx + (size_t)z * d.cx_sb + m;
        *q = (d.flags & VG_FC_C_ACCUM) ? *q + v : v;
        return;
    }
    if (d.flags & VG_FC_C_BIAS) v += p.bias[(size_t)z * d.bias_sb + n];
    if (d.flags & VG_FC_C_RELU) v = v > 0.f ? v : 0.f;
    const size_t o = (size_t)z * d.c_sb + (size_t)m * d.c_sm + n;
    if ((d.flags & VG_FC_C_MASK) && !(p.cmask[o] > 0.f)) v = 0.f;
    p.C[o] = (d.flags & VG_FC_C_ACCUM) ? p.C[o] + v : v;
}

// AV / BV: how an operand is read.  1: 16-byte loads ALONG K (k-contiguous, aligned, K % 4 == 0) -> one 16-byte LDS write.  2: 16-byte
// loads ALONG M / N (the operand is contiguous along its row index: every weight-gradient operand, the weights of a data gradient;
// aligned, row count % 4 == 0): a unit is 4 consecutive rows at one k -> four dword LDS writes; a quarter of the load instructions and
// address arithmetic of form 0.  0: dword loads (any layout).  Compile-time, one body per combination: with the load forms as run-time
// alternatives inside one body the register allocator shared registers between them and every unit of the dword form began with
// s_waitcnt vmcnt(0).
// W: MFMA tiles per wave and direction (tile edge 32 * W).
template <int W, int AV, int BV>
__device__ __forceinline__ void fc_body(const FcArgs& p, int b, float* __restrict__ As, float* __restrict__ Bs) {
    constexpr int FC_T = 32 * W;
    constexpr int FC_U = FC_T * (FC_TK / 4) / 256;     // 16-byte units per thread, operand and step
    const vg_fc_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id();
    const int wm = wave & 1, wn = wave >> 1;
    const int bx = b % p.gx; b /= p.gx;
    const int by = b % p.gy;
    const int z = b / p.gy;
    const int m0 = bx * FC_T, n0 = by * FC_T;
    const bool split = d.ksplit > 1;
    const int kb = split ? z * p.kchunk : 0, ke = split ? min(d.K, kb + p.kchunk) : d.K;
    const float* __restrict__ A = p.A + (split ? 0 : (size_t)z * d.a_sb);
    // no mask: the "mask" is A itself against a threshold of -inf (always open) -- a conditional second load per element made the
    // compiler wait between the loads
    const float* __restrict__ Am = p.amask ? p.amask + (split ? 0 : (size_t)z * d.a_sb) : A;
    const float mthr = p.amask ? 0.f : -__builtin_inff();
    const float* __restrict__ B = p.B + (split ? 0 : (size_t)z * d.b_sb);
    const bool a_relu = d.flags & VG_FC_A_RELU, b_relu = d.flags & VG_FC_B_RELU, b_ones = d.flags & VG_FC_B_ONES;
    const bool a_kc = d.a_sk == 1, b_kc = d.b_sk == 1;

    // unit u of an operand tile: row r (m or n inside the tile), k-quad q; lanes run along the operand's contiguous index
    int ar[FC_U], aq[FC_U], br[FC_U], bq[FC_U];
#pragma unroll
    for (int i = 0; i < FC_U; ++i) {
        const int u = tid + i * 256;
        ar[i] = a_kc ? u / (FC_TK / 4) : u % FC_T; aq[i] = a_kc ? u % (FC_TK / 4) : u / FC_T;
        br[i] = b_kc ? u / (FC_TK / 4) : u % FC_T; bq[i] = b_kc ? u % (FC_TK / 4) : u / FC_T;
        // form 2: ar = first of the unit's 4 rows, aq = its k inside the step (lanes along the row groups: 16 contiguous bytes each)
        if (AV == 2) { ar[i] = 4 * (u % (FC_T / 4)); aq[i] = u / (FC_T / 4); }
        if (BV == 2) { br[i] = 4 * (u % (FC_T / 4)); bq[i] = u / (FC_T / 4); }
    }
    // fetch() only LOADS (raw operand values, raw mask values; indices clamped into the operand so that no load needs a predicate whose
    // result is consumed at once); zero fill, mask, ReLU and the ones column are applied in put(), four steps later.  (With the
    // selects next to the loads every dword load was followed by its own s_waitcnt: a 9-step weight gradient took 36 us.)
    float ra[FC_DEPTH][FC_U][4], rm[FC_DEPTH][FC_U][4], rb[FC_DEPTH][FC_U][4];
    // (no divergent choice between load forms either: two exec-masked paths that write the same registers make the compiler wait
    // between them.  The 16-byte form is chosen per operand for the whole launch -- K a multiple of 4 -- and reads a clamped quad.)
    // Address arithmetic and selects are most of a step's instructions (first version: 475 vector instructions per step beside 16 MFMAs,
    // and the SIMD that issues them is the one that runs the MFMAs): row pointers are formed once, offsets are 32-bit, and a step whose
    // tile lies inside both operands and inside K skips every validity select.
    const int klast = ke - 1, qlast = max(ke - 4, 0) & ~3;
    const int ask = (int)d.a_sk, bsk = (int)d.b_sk;
    const ptrdiff_t mdelta = Am - A;
    const float* pa[FC_U]; const float* pb[FC_U];
#pragma unroll
    for (int i = 0; i < FC_U; ++i) {
        pa[i] = A + (size_t)min(m0 + ar[i], d.M - 1) * d.a_sm;
        pb[i] = B + (size_t)min(n0 + br[i], d.N - 1) * d.b_sn;
    }
    auto fetch = [&](int k0, float (&xa)[FC_U][4], float (&xm)[FC_U][4], float (&xb)[FC_U][4]) {
#pragma unroll
        for (int i = 0; i < FC_U; ++i) {
            const int k = k0 + 4 * aq[i];
            if (AV == 2) {
                const float* q = A + min(m0 + ar[i], d.M - 4) + (size_t)min(k0 + aq[i], klast) * d.a_sk;
                const float4 t = *reinterpret_cast<const float4*>(q);
                xa[i][0] = t.x; xa[i][1] = t.y; xa[i][2] = t.z; xa[i][3] = t.w;
                const float4 q4 = *reinterpret_cast<const float4*>(q + mdelta); xm[i][0] = q4.x; xm[i][1] = q4.y; xm[i][2] = q4.z; xm[i][3] = q4.w;
            } else if (AV == 1) {
                const float* q = pa[i] + min(k, qlast);
                const float4 t = *reinterpret_cast<const float4*>(q);
                xa[i][0] = t.x; xa[i][1] = t.y; xa[i][2] = t.z; xa[i][3] = t.w;
                const float4 q4 = *reinterpret_cast<const float4*>(q + mdelta); xm[i][0] = q4.x; xm[i][1] = q4.y; xm[i][2] = q4.z; xm[i][3] = q4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* q = pa[i] + min(k + e, klast) * ask;
                    xa[i][e] = *q;
                    xm[i][e] = q[mdelta];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FC_U; ++i) {
            const int k = k0 + 4 * bq[i];
            if (BV == 2) {
                const float4 t = *reinterpret_cast<const float4*>(B + min(n0 + br[i], d.N - 4) + (size_t)min(k0 + bq[i], klast) * d.b_sk);
                xb[i][0] = t.x; xb[i][1] = t.y; xb[i][2] = t.z; xb[i][3] = t.w;
            } else if (BV == 1) {
                const float4 t = *reinterpret_cast<const float4*>(pb[i] + min(k, qlast));
                xb[i][0] = t.x; xb[i][1] = t.y; xb[i][2] = t.z; xb[i][3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[i][e] = pb[i][min(k + e, klast) * bsk];
            }
        }
    };
    const bool inside = m0 + FC_T <= d.M && n0 + FC_T <= d.N;          // (the ones column, n = N, is never in such a tile)
    const float alo = a_relu ? 0.f : -__builtin_inff(), blo = b_relu ? 0.f : -__builtin_inff();
    auto put = [&](int k0, const float (&xa)[FC_U][4], const float (&xm)[FC_U][4], const float (&xb)[FC_U][4]) {
        if (inside && k0 + FC_TK <= ke) {
#pragma unroll
            for (int i = 0; i < FC_U; ++i) {
                float4 t, u4;
                t.x = vg_max((xm[i][0] > mthr) ? xa[i][0] : 0.f, alo); t.y = vg_max((xm[i][1] > mthr) ? xa[i][1] : 0.f, alo);
                t.z = vg_max((xm[i][2] > mthr) ? xa[i][2] : 0.f, alo); t.w = vg_max((xm[i][3] > mthr) ? xa[i][3] : 0.f, alo);
                u4.x = vg_max(xb[i][0], blo); u4.y = vg_max(xb[i][1], blo); u4.z = vg_max(xb[i][2], blo); u4.w = vg_max(xb[i][3], blo);
                if (AV == 2) { float* q = &As[ar[i] * FC_PITCH + aq[i]]; q[0] = t.x; q[FC_PITCH] = t.y; q[2 * FC_PITCH] = t.z; q[3 * FC_PITCH] = t.w; }
                else *reinterpret_cast<float4*>(&As[ar[i] * FC_PITCH + 4 * aq[i]]) = t;
                if (BV == 2) { float* q = &Bs[br[i] * FC_PITCH + bq[i]]; q[0] = u4.x; q[FC_PITCH] = u4.y; q[2 * FC_PITCH] = u4.z; q[3 * FC_PITCH] = u4.w; }
                else *reinterpret_cast<float4*>(&Bs[br[i] * FC_PITCH + 4 * bq[i]]) = u4;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < FC_U; ++i) {
            float va[4], vb[4];
            // element e of a unit: (row, k) = (r, k + e) in forms 0 / 1, (r + e, k) in form 2
            const int m = m0 + ar[i], ka = k0 + (AV == 2 ? aq[i] : 4 * aq[i]);
            const int n = n0 + br[i], kq = k0 + (BV == 2 ? bq[i] : 4 * bq[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int me = AV == 2 ? m + e : m, kae = AV == 2 ? ka : ka + e;
                float v = (me < d.M && kae < ke) ? xa[i][e] : 0.f;
                if (!(xm[i][e] > mthr)) v = 0.f;
                va[e] = a_relu ? (v > 0.f ? v : 0.f) : v;
                const int ne = BV == 2 ? n + e : n, kbe = BV == 2 ? kq : kq + e;
                float w = (ne < d.N && kbe < ke) ? xb[i][e] : 0.f;
                if (b_relu) w = w > 0.f ? w : 0.f;
                if (b_ones && ne == d.N) w = (kbe < ke) ? 1.f : 0.f;
                vb[e] = w;
            }
            if (AV == 2) { float* q = &As[ar[i] * FC_PITCH + aq[i]]; q[0] = va[0]; q[FC_PITCH] = va[1]; q[2 * FC_PITCH] = va[2]; q[3 * FC_PITCH] = va[3]; }
            else { float4 t; t.x = va[0]; t.y = va[1]; t.z = va[2]; t.w = va[3]; *reinterpret_cast<float4*>(&As[ar[i] * FC_PITCH + 4 * aq[i]]) = t; }
            if (BV == 2) { float* q = &Bs[br[i] * FC_PITCH + bq[i]]; q[0] = vb[0]; q[FC_PITCH] = vb[1]; q[2 * FC_PITCH] = vb[2]; q[3 * FC_PITCH] = vb[3]; }
            else { float4 u4; u4.x = vb[0]; u4.y = vb[1]; u4.z = vb[2]; u4.w = vb[3]; *reinterpret_cast<float4*>(&Bs[br[i] * FC_PITCH + 4 * bq[i]]) = u4; }
        }
    };

    vg_f32x4 acc[W][W];
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j2 = 0; j2 < W; ++j2) acc[i][j2].v[0] = acc[i][j2].v[1] = acc[i][j2].v[2] = acc[i][j2].v[3] = 0.f;
    // straight-line pipeline: every fetch is unconditional (clamped addresses are always inside the operand), the step count is rounded
    // up to the depth (a step past the end multiplies zeros) -- conditionally defined register arrays turned into whole-array copies
    // and waits in the generated code
#pragma unroll
    for (int s = 0; s < FC_DEPTH; ++s) fetch(kb + s * FC_TK, ra[s], rm[s], rb[s]);
    for (int k0 = kb; k0 < ke; k0 += FC_DEPTH * FC_TK) {
#pragma unroll
        for (int s = 0; s < FC_DEPTH; ++s) {
            const int ks = k0 + s * FC_TK;
            __syncthreads();                               // the previous step's operand reads are done
            put(ks, ra[s], rm[s], rb[s]);
            __syncthreads();
            fetch(ks + FC_DEPTH * FC_TK, ra[s], rm[s], rb[s]);          // FC_DEPTH steps ahead, behind the MFMAs
#pragma unroll
            for (int blk = 0; blk < FC_TK / 16; ++blk) {
                float4 af[W], bf[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    af[i] = *reinterpret_cast<const float4*>(&As[((wm * W + i) * 16 + (lane & 15)) * FC_PITCH + blk * 16 + 4 * (lane >> 4)]);
                    bf[i] = *reinterpret_cast<const float4*>(&Bs[((wn * W + i) * 16 + (lane & 15)) * FC_PITCH + blk * 16 + 4 * (lane >> 4)]);
                }
#pragma unroll
                for (int i = 0; i < W; ++i)
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) {
                        vg_mfma16(af[i].x, bf[j2].x, acc[i][j2]); vg_mfma16(af[i].y, bf[j2].y, acc[i][j2]);
                        vg_mfma16(af[i].z, bf[j2].z, acc[i][j2]); vg_mfma16(af[i].w, bf[j2].w, acc[i][j2]);
                    }
            }
        }
    }
    // epilogue: every load it needs (bias, output mask, the old C / cx when accumulating) is issued before the first use -- element by
    // element (load, wait, store) it was 4 W^2 serial round trips, longer than the product itself
    const int NW = d.N + (b_ones ? 1 : 0);
    if (split) {
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
            for (int j2 = 0; j2 < W; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * W + i) * 16 + (lane >> 4) * 4 + r, n = n0 + (wn * W + j2) * 16 + (lane & 15);
                    if (m < d.M && n < NW) p.ws[((size_t)z * d.M + m) * NW + n] = acc[i][j2].v[r];
                }
        return;
    }
    float bv[W], cmv[W][W][4], cold[W][W][4];
    float* dst[W][W][4];
#pragma unroll
    for (int j2 = 0; j2 < W; ++j2) {
        const int n = min(n0 + (wn * W + j2) * 16 + (lane & 15), d.N - 1);
        bv[j2] = (d.flags & VG_FC_C_BIAS) ? p.bias[(size_t)z * d.bias_sb + n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j2 = 0; j2 < W; ++j2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + (wm * W + i) * 16 + (lane >> 4) * 4 + r, d.M - 1), n = min(n0 + (wn * W + j2) * 16 + (lane & 15), NW - 1);
                dst[i][j2][r] = (n == d.N) ? p.cx + (size_t)z * d.cx_sb + m : p.C + (size_t)z * d.c_sb + (size_t)m * d.c_sm + n;
                cmv[i][j2][r] = 1.f; cold[i][j2][r] = 0.f;
            }
    if (d.flags & VG_FC_C_MASK) {
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
            for (int j2 = 0; j2 < W; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) cmv[i][j2][r] = p.cmask[dst[i][j2][r] - p.C];       // (no ones column together with an output mask)
    }
    if (d.flags & VG_FC_C_ACCUM) {
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
            for (int j2 = 0; j2 < W; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) cold[i][j2][r] = *dst[i][j2][r];
    }
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j2 = 0; j2 < W; ++j2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * W + i) * 16 + (lane >> 4) * 4 + r, n = n0 + (wn * W + j2) * 16 + (lane & 15);
                float v = acc[i][j2].v[r];
                if (n < d.N) {
                    v += bv[j2];
                    if (d.flags & VG_FC_C_RELU) v = v > 0.f ? v : 0.f;
                    if (!(cmv[i][j2][r] > 0.f)) v = 0.f;
                }
                if (m < d.M && n < NW) *dst[i][j2][r] = cold[i][j2][r] + v;
            }
}

// one kernel per tile size (the 64 x 64 bodies need twice the registers: in one kernel they would halve the occupancy of the small jobs)
template <int W>
__global__ void __launch_bounds__(256)
fc_gemm_k(FcJobs js) {
    __shared__ __attribute__((aligned(16))) float As[32 * W * FC_PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[32 * W * FC_PITCH];
    int j = 0;
    while (j + 1 < js.n && (int)blockIdx.x >= js.first[j + 1]) ++j;
    const FcArgs& p = js.job[j];
    const int b = (int)blockIdx.x - js.first[j];
    switch (p.a_vec * 3 + p.b_vec) {
        case 0: fc_body<W, 0, 0>(p, b, As, Bs); break;
        case 1: fc_body<W, 0, 1>(p, b, As, Bs); break;
        case 2: fc_body<W, 0, 2>(p, b, As, Bs); break;
        case 3: fc_body<W, 1, 0>(p, b, As, Bs); break;
        case 4: fc_body<W, 1, 1>(p, b, As, Bs); break;
        case 5: fc_body<W, 1, 2>(p, b, As, Bs); break;
        case 6: fc_body<W, 2, 0>(p, b, As, Bs); break;
        case 7: fc_body<W, 2, 1>(p, b, As, Bs); break;
        default: fc_body<W, 2, 2>(p, b, As, Bs); break;
    }
}

// split-K: C = epilogue(sum_z ws[z]) in a fixed order, for every job that was split
__global__ void __launch_bounds__(256)
fc_reduce_k(FcJobs js) {
    for (int j = 0; j < js.n; ++j) {
        const FcArgs& p = js.job[j];
        const vg_fc_desc& d = p.d;
        if (d.ksplit <= 1) continue;
        const int NW = d.N + ((d.flags & VG_FC_B_ONES) ? 1 : 0);
        const size_t tot = (size_t)d.M * NW;
        FcArgs q = p; q.d.ksplit = 1;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int z = 0; z < d.ksplit; ++z) v += p.ws[(size_t)z * tot + i];
            fc_store(q, 0, (int)(i / NW), (int)(i % NW), v);
        }
    }
}

int fc_prepare(const vg_fc_job* jb, FcArgs* p) {
    const vg_fc_desc* d = &jb->d;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->ksplit <= 0) { vg_set_error("vg_fc_gemm: bad shape"); return VG_ERR_ARG; }
    if ((d->a_sm != 1 && d->a_sk != 1) || (d->b_sk != 1 && d->b_sn != 1)) { vg_set_error("vg_fc_gemm: one stride of each operand must be 1"); return VG_ERR_ARG; }
    if (d->ksplit > 1 && d->batch != 1) { vg_set_error("vg_fc_gemm: split-K and batch exclude each other"); return VG_ERR_ARG; }
    if ((d->flags & VG_FC_C_MASK) && (d->flags & VG_FC_B_ONES)) { vg_set_error("vg_fc_gemm: VG_FC_C_MASK and VG_FC_B_ONES exclude each other"); return VG_ERR_ARG; }
    if (!jb->A || !jb->B || !jb->C) { vg_set_error("vg_fc_gemm: null operand"); return VG_ERR_ARG; }
    if (((d->flags & VG_FC_A_MASK) && !jb->amask) || ((d->flags & VG_FC_C_BIAS) && !jb->bias) || ((d->flags & VG_FC_C_MASK) && !jb->cmask) ||
        ((d->flags & VG_FC_B_ONES) && !jb->cx) || (d->ksplit > 1 && !jb->ws)) { vg_set_error("vg_fc_gemm: a flag is set whose buffer is null"); return VG_ERR_ARG; }
    const int kc = (d->K + d->ksplit - 1) / d->ksplit;
    const int kchunk = (kc + FC_TK - 1) / FC_TK * FC_TK;
    if ((long)kchunk * (d->ksplit - 1) >= d->K) { vg_set_error("vg_fc_gemm: ksplit=%d leaves an empty piece of K=%d", d->ksplit, d->K); return VG_ERR_ARG; }
    p->d = *d; p->A = jb->A; p->amask = (d->flags & VG_FC_A_MASK) ? jb->amask : nullptr; p->B = jb->B; p->bias = jb->bias; p->cmask = jb->cmask;
    p->C = jb->C; p->cx = jb->cx; p->ws = jb->ws; p->kchunk = kchunk;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool a_al = al16(jb->A) && (!p->amask || al16(p->amask)) && d->a_sb % 4 == 0, b_al = al16(jb->B) && d->b_sb % 4 == 0;
    p->a_vec = (d->a_sk == 1 && d->K % 4 == 0 && d->a_sm % 4 == 0 && a_al) ? 1 : (d->a_sm == 1 && d->a_sk != 1 && d->M % 4 == 0 && d->a_sk % 4 == 0 && a_al) ? 2 : 0;
    p->b_vec = (d->b_sk == 1 && d->K % 4 == 0 && d->b_sn % 4 == 0 && b_al) ? 1 : (d->b_sn == 1 && d->b_sk != 1 && d->N % 4 == 0 && d->b_sk % 4 == 0 && b_al) ? 2 : 0;
    const int NW = d->N + ((d->flags & VG_FC_B_ONES) ? 1 : 0);
    // 32 x 32 tiles (several blocks per CU hide the round trips) until they number >= 768 per launch-job; then 64 x 64 (4x the MFMAs per
    // LDS byte and barrier: fc8 forward / weight gradient / split data gradient)
    const long t32 = (long)((d->M + 31) / 32) * ((NW + 31) / 32) * (d->ksplit > 1 ? d->ksplit : d->batch);
    p->big = t32 >= 768;
    return VG_OK;
}

}  // namespace

extern "C" int64_t vg_fc_ws_bytes(const vg_fc_desc* d) {
    if (!d || d->ksplit <= 1) return 0;
    return (int64_t)d->ksplit * d->M * (d->N + ((d->flags & VG_FC_B_ONES) ? 1 : 0)) * (int64_t)sizeof(float);
}

extern "C" int vg_fc_gemm_jobs(const vg_fc_job* jobs, int32_t njobs, void* stream) {
    if (!jobs || njobs < 1 || njobs > FC_MAXJOBS) { vg_set_error("vg_fc_gemm_jobs: 1..%d jobs", FC_MAXJOBS); return VG_ERR_ARG; }
    FcJobs js; js.n = njobs; js.first[0] = 0;
    bool any_split = false; size_t red = 0;
    bool big = true;
    for (int j = 0; j < njobs; ++j) {
        const int rc = fc_prepare(&jobs[j], &js.job[j]);
        if (rc != VG_OK) return rc;
        big = big && js.job[j].big;
    }
    for (int j = 0; j < njobs; ++j) {
        const vg_fc_desc& d = js.job[j].d;
        const int T = big ? 64 : 32;
        js.job[j].gx = (d.M + T - 1) / T; js.job[j].gy = (d.N + ((d.flags & VG_FC_B_ONES) ? 1 : 0) + T - 1) / T;
        const long nb = (long)js.job[j].gx * js.job[j].gy * (d.ksplit > 1 ? d.ksplit : d.batch);
        if (js.first[j] + nb > (1L << 30)) { vg_set_error("vg_fc_gemm: too many tiles"); return VG_ERR_ARG; }
        js.first[j + 1] = js.first[j] + (int)nb;
        if (d.ksplit > 1) { any_split = true; const size_t t = (size_t)d.M * (d.N + 1); if (t > red) red = t; }
    }
    hipStream_t s = (hipStream_t)stream;
    if (big) vg_launch(fc_gemm_k<2>, dim3(js.first[njobs]), dim3(256), 0, s, js);
    else vg_launch(fc_gemm_k<1>, dim3(js.first[njobs]), dim3(256), 0, s, js);
    const int r2 = vg_check_launch("fc_gemm");
    if (r2 != VG_OK || !any_split) return r2;
    int nb = (int)((red + 255) / 256); if (nb > 1024) nb = 1024;
    vg_launch(fc_reduce_k, dim3(nb), dim3(256), 0, s, js);
    return vg_check_launch("fc_reduce");
}

extern "C" int vg_fc_gemm(const vg_fc_desc* d, const float* A, const float* amask, const float* B, const float* bias, const float* cmask,
                          float* C, float* cx, float* ws, void* stream) {
    if (!d) { vg_set_error("vg_fc_gemm: null descriptor"); return VG_ERR_ARG; }
    vg_fc_job jb; jb.d = *d; jb.A = A; jb.amask = amask; jb.B = B; jb.bias = bias; jb.cmask = cmask; jb.C = C; jb.cx = cx; jb.ws = ws;
    return vg_fc_gemm_jobs(&jb, 1, stream);
}
