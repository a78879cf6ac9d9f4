// vg_conv_mfma.hip -- stride-1 3x3x3 correlation on the fp32 matrix cores (gfx950), the engine behind vg_corr3d for
// conv1/3/5 and convt1/3/5 (forward and data gradient) of vae_reg_GP.py:189-215.
//
//   y[co][od][oh][ow] = bias[co] + sum_{ci,kd,kh,kw} W[ci][kd][kh][kw][co] * P(x)[ci][od+kd-pd][oh+kh-ph][ow+kw-pw]
//
// One v_mfma_f32_16x16x4_f32 (exact fp32) covers ONE (ci, kd, kh) and the FOUR input columns kw' = 0..3:
//   B[kw'][j]  = the input value at column (ow_j + kw')              -- one ds_read_b32 at  tile row offset + lane constant
//   A[row][kw'] = W[.., kw'][co = row]                               -- 16-channel layers: 3 of the 4 k's carry weights
//   A[row][kw'] = W[.., kw' - (row >> 3)][co = row & 7]              -- 8-channel layers ("Toeplitz pair"): rows 8..15 hold
//                  the weights shifted by one column, so they produce the NEXT output column from the same B operand:
//                  a tile is 8 channels x 32 positions and no matrix row is idle.
// The input tile lives in LDS as a ZERO-PADDED image (row pitch IW + 2 pad): it is filled by LDS-DMA with per-lane source
// addresses (halo lanes are simply inactive), the ReLU / batch-norm affine is applied once per staged element by the wave
// that fetched it, and the inner loop is  v_add + ds_read + mfma  with no masks, no divisions and no branches.
// A wave keeps NG accumulator tiles; every per-lane offset (operand reads, output stores) is computed once per kernel,
// because every item of the persistent grid has the same tile shape.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

constexpr int QMAX = 8;                 // LDS-DMA instructions per staged (channel, plane): rows * pitch <= 64 * QMAX cells

struct S1mParams {
    vg_conv_desc d;
    int TD, TH;                         // output planes / rows per item
    int LD, LR, RWP, PLP, CHP;          // staged planes, rows per plane, row pitch, plane pitch, channel pitch (floats)
    int Q;                              // DMA instructions per (channel, plane)
    int cc;                             // input channels per chunk
    int t_off;                          // float offset of the input tile (the weight image comes first)
    int lds_floats;
    int items, odb, ohb;
};

template <bool TOEP, int NG>
__global__ void __launch_bounds__(256)
corr3d_s1m_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
             const float* __restrict__ in_scale, const float* __restrict__ in_shift,
             const float* __restrict__ mask_src, float* __restrict__ y, S1mParams p) {
    VG_DYN_SMEM(float, lds);
    const vg_conv_desc& d = p.d;
    const int CI = d.CI, CO = d.CO;
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id();
    const int kq = lane >> 4, jl = lane & 15;
    float* tile = lds + p.t_off;

    for (int i = tid; i < p.lds_floats; i += blockDim.x) lds[i] = 0.f;       // halo cells are never written again: they stay zero
    __syncthreads();
    // ---- weight image wl[ci][kd*3+kh][kw'][row]  (zero where a row has no weight for that kw').  Gathers are issued in
    //      batches of 8 before their LDS writes: one load -> wait -> write per iteration costs a full memory latency each
    //      (36 of them for 16 channels was ~20 us of every launch).
    {
        const int nel = CI * 9 * 64;
        for (int i0 = tid; i0 < nel; i0 += 8 * 256) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int row = i & 15, kwp = (i >> 4) & 3, r2 = i >> 6;
                const int kk = r2 % 9, ci = r2 / 9;
                const int co = TOEP ? (row & 7) : row;
                const int kw = TOEP ? kwp - (row >> 3) : kwp;
                const bool ok = i < nel && kw >= 0 && kw < 3 && co < CO;
                wv[u] = ok ? wpk[((size_t)ci * 27 + kk * 3 + kw) * CO + co] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * 256 < nel) lds[i0 + u * 256] = wv[u];
        }
    }
    // ---- staging geometry of this lane: cell q*64 + lane of a plane image -> (row, column)
    int srcoff[QMAX], rowq[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int e = q * VG_WAVE + lane;
        const int r = e / p.RWP, c = e - r * p.RWP;
        const int iw = c - d.pad_w;
        const bool ok = q < p.Q && r < p.LR && iw >= 0 && iw < d.IW;
        rowq[q] = ok ? r : (1 << 20);                                       // an out-of-range row never passes the ih test
        srcoff[q] = ok ? r * d.IW + iw : 0;
    }
    // ---- compute geometry of this lane: for each of its NG position groups the LDS offset of its B operand and the
    //      offset / validity of the outputs it owns (column jl of the group; rows (kq*4 + r) of the tile)
    const int PR = TOEP ? (d.OW + 1) / 2 : d.OW;                            // columns of the position grid per output row
    const int oplane = d.OH * d.OW;
    const size_t ovol = (size_t)oplane * d.OD;
    int posOff[NG], odl[NG], ohl[NG], owl[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int pp = ((g * 4 + wave) * 16) + jl;
        const int r = pp / PR, pc = pp - r * PR;
        const bool ok = r < p.TD * p.TH;
        odl[g] = ok ? r / p.TH : (1 << 20);
        ohl[g] = ok ? r % p.TH : 0;
        owl[g] = TOEP ? 2 * pc : pc;
        posOff[g] = ok ? odl[g] * p.PLP + ohl[g] * p.RWP + owl[g] + kq : kq;
    }
    const int par = TOEP ? (lane >> 5) : 0;                                  // rows 8..15 of a Toeplitz tile: the odd column
    const int cob = TOEP ? ((lane >> 4) & 1) * 4 : (lane >> 4) * 4;          // first of this lane's 4 output channels
    const float lo = d.relu_in ? 0.f : -__builtin_inff();
    const bool has_pro = d.relu_in || in_scale != nullptr;
    const int plane = d.IH * d.IW;
    const size_t vol = (size_t)plane * d.ID;
    __syncthreads();

    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
        const int n = item / (p.odb * p.ohb); const int rem = item - n * (p.odb * p.ohb);
        const int od0 = (rem / p.ohb) * p.TD, oh0 = (rem % p.ohb) * p.TH;
        const int id0 = od0 - d.pad_d, ih0 = oh0 - d.pad_h;
        const int gaff = (in_scale != nullptr) ? n / d.per_group : 0;
        unsigned okq = 0;                                                 // bit q: this lane's cell q of a plane image is a real input row
#pragma unroll
        for (int q = 0; q < QMAX; ++q) { const int ih = ih0 + rowq[q]; okq |= (ih >= 0 && ih < d.IH ? 1u : 0u) << q; }
        vg_f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g].v[0] = 0.f; acc[g].v[1] = 0.f; acc[g].v[2] = 0.f; acc[g].v[3] = 0.f; }

        for (int c0 = 0; c0 < CI; c0 += p.cc) {
            const int cc = min(p.cc, CI - c0);
            __syncthreads();                                              // previous chunk / item fully consumed
            // ---- stage: (channel, plane) images dealt over the 4 waves; inactive lanes (halo, outside the tensor) write 0
            for (int cl = wave; cl < cc * p.LD; cl += 4) {
                const int c = cl / p.LD, l = cl - c * p.LD;
                const int id = id0 + l;
                const bool pl_ok = id >= 0 && id < d.ID;
                const float* src = x + ((size_t)n * CI + c0 + c) * vol + (size_t)(pl_ok ? id : 0) * plane + (ptrdiff_t)ih0 * d.IW;
                float* dst = tile + c * p.CHP + l * p.PLP;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    if (q >= p.Q) break;
                    const bool ok = pl_ok && ((okq >> q) & 1u);
                    if (ok) vg_dma4(src + srcoff[q], dst + q * VG_WAVE);
                    else dst[q * VG_WAVE + lane] = 0.f;
                }
            }
            vg_dma_wait();
            if (has_pro) {
                // ReLU / batch-norm affine, once per element, by the wave that fetched it (its own DMAs have landed)
                for (int cl = wave; cl < cc * p.LD; cl += 4) {
                    const int c = cl / p.LD, l = cl - c * p.LD;
                    const int id = id0 + l;
                    if (id < 0 || id >= d.ID) continue;
                    float sc = 1.f, sh = 0.f;
                    if (in_scale != nullptr) { sc = in_scale[gaff * CI + c0 + c]; sh = in_shift[gaff * CI + c0 + c]; }
                    float* dst = tile + c * p.CHP + l * p.PLP + lane;
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        if (q >= p.Q) break;
                        if ((okq >> q) & 1u) dst[q * VG_WAVE] = fmaf(vg_max(dst[q * VG_WAVE], lo), sc, sh);
                    }
                }
            }
            __syncthreads();
            // ---- matrix work of the chunk: (channel, kd, kh) k-steps x NG position groups
            for (int c = 0; c < cc; ++c) {
                const float* wl = lds + (size_t)(c0 + c) * 9 * 64 + lane;
                const float* tc = tile + c * p.CHP;
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) {
                    const float aw = wl[kk * 64];
                    const float* tr = tc + (kk / 3) * p.PLP + (kk % 3) * p.RWP;
                    float bv[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) bv[g] = tr[posOff[g]];
#pragma unroll
                    for (int g = 0; g < NG; ++g) vg_mfma16(aw, bv[g], acc[g]);
                }
            }
        }
        // ---- epilogue: lane owns column jl of each group, channels cob..cob+3, (Toeplitz) the even or the odd position.
        //      All mask loads are issued before the first store (a load -> select -> store chain per element would pay
        //      the full memory latency NG*4 times).
        const size_t ybase = (size_t)n * CO * ovol;
        size_t o0[NG]; bool ok[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int od = od0 + odl[g], oh = oh0 + ohl[g], ow = owl[g] + par;
            ok[g] = odl[g] < p.TD && od < d.OD && oh < d.OH && ow < d.OW;
            o0[g] = ybase + (size_t)(ok[g] ? od : 0) * oplane + (size_t)(ok[g] ? oh : 0) * d.OW + (ok[g] ? ow : 0);
        }
        float mk[NG][4];
        if (mask_src) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[g][r] = mask_src[o0[g] + (size_t)min(cob + r, CO - 1) * ovol];
        }
        float bv4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv4[r] = (bias && cob + r < CO) ? bias[cob + r] : 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[g].v[r] + bv4[r];
                if (mask_src) v = (mk[g][r] > 0.f) ? v : 0.f;
                if (ok[g] && cob + r < CO) y[o0[g] + (size_t)(cob + r) * ovol] = v;
            }
    }
}

template <bool TOEP, int NG>
int launch_s1m(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
               const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    S1mParams p; p.d = *d;
    const int PR = TOEP ? (d->OW + 1) / 2 : d->OW;
    p.RWP = d->IW + 2 * d->pad_w;
    const int maxgroups = 4 * NG;
    p.cc = d->CI < 4 ? d->CI : 4;
    const size_t wfl = (size_t)d->CI * 9 * 64;
    const size_t budget = 64 * 1024;
    // largest tile (TD planes x TH rows of full-width output rows) whose positions fit the wave's accumulators, whose plane
    // image fits QMAX DMA instructions and whose chunk fits LDS; rows are split evenly over the row blocks
    int best_td = 0, best_th = 0; long best_pos = 0;
    for (int td = 1; td <= d->OD && td <= 4; ++td)
        for (int nb = 1; nb <= d->OH; ++nb) {
            const int th = (d->OH + nb - 1) / nb;
            const int LR = th + 2, LD = td + 2;
            if ((long)LR * p.RWP > 64 * QMAX) continue;
            const int Q = (LR * p.RWP + 63) / 64;
            if (((long)td * th * PR + 15) / 16 > maxgroups) continue;
            const size_t fl = wfl + (size_t)p.cc * LD * Q * 64 + 64;
            if (fl * 4 > budget) continue;
            const long pos = (long)td * th * PR;
            // prefer more positions per tile, but never a tile that leaves the last depth block mostly empty -- and never
            // fewer items than CUs while a smaller tile would still give every wave a group
            const double util = ((double)d->OD / (((d->OD + td - 1) / td) * td)) * ((double)d->OH / (nb * th));
            const long items = (long)d->N * ((d->OD + td - 1) / td) * nb;
            const double fill = items >= 256 ? 1.0 : (double)items / 256.0;
            const long score = (long)(pos * util * fill * fill);
            if (score > best_pos) { best_pos = score; best_td = td; best_th = th; }
            break;                                                        // smaller th for this td only lowers the score
        }
    if (!best_td) return -1;
    p.TD = best_td; p.TH = best_th; p.LD = p.TD + 2; p.LR = p.TH + 2;
    p.Q = (p.LR * p.RWP + 63) / 64; p.PLP = p.Q * 64; p.CHP = p.LD * p.PLP;
    p.t_off = (int)wfl;
    // as many channels per chunk as LDS holds (one fill + one barrier pair per chunk: the small layers then stage once per item)
    for (int cc = d->CI; cc > p.cc; cc = (cc + 1) / 2)
        if ((wfl + (size_t)cc * p.CHP + 64) * 4 <= budget) { p.cc = cc; break; }
    p.lds_floats = (int)(wfl + (size_t)p.cc * p.CHP + 64);
    p.odb = (d->OD + p.TD - 1) / p.TD; p.ohb = (d->OH + p.TH - 1) / p.TH;
    p.items = d->N * p.odb * p.ohb;
    const size_t shmem = (size_t)p.lds_floats * sizeof(float);
    int per_cu = vg_blocks_per_cu((const void*)corr3d_s1m_k<TOEP, NG>, 256, shmem);
    if (per_cu > 6) per_cu = 6;
    int grid = 256 * per_cu; if (grid > p.items) grid = p.items;
    vg_launch(corr3d_s1m_k<TOEP, NG>, dim3(grid), dim3(256), shmem, s, x, wpk, bias, in_scale, in_shift, mask_src, y, p);
    return vg_check_launch("corr3d_s1m");
}

}  // namespace

// -1: geometry not covered here (the caller falls back to the VALU kernels)
int vg_corr3d_s1_mfma(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                      const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    if (d->KD != 3 || d->KH != 3 || d->KW != 3 || d->stride != 1) return -1;
    if (d->pad_d != d->pad_h || d->pad_h != d->pad_w || (d->pad_w != 0 && d->pad_w != 2)) return -1;
    if (d->OD != d->ID + 2 * d->pad_d - 2 || d->OH != d->IH + 2 * d->pad_h - 2 || d->OW != d->IW + 2 * d->pad_w - 2) return -1;
    if (d->CI > 16 || d->CI < 8) return -1;
    // measured on MI355X (tools/layer_bench.py): the staged-tile MFMA form wins on the launches below ~2 GFLOP (conv3, conv5,
    // convt1: 35-50 us instead of 40-80); on the large decoder layers the register-tiled VALU kernels are still faster
    // (convt3 141 vs 182 us), because there the per-tile staging (halo re-reads, two barriers per 4 channels) dominates
    const double gflop = 2.0 * 27.0 * d->CI * d->CO * (double)d->N * d->OD * d->OH * d->OW * 1e-9;
    if (gflop > 2.0) return -1;
    // accumulator tiles per wave: the smallest instance whose 4*NG groups still hold the tile the planner would pick
    // (a tiny layer on a 4-tile instance issues 4 matrix instructions per k-step for 1 useful one)
    const int PRr = d->CO == 8 ? (d->OW + 1) / 2 : d->OW;
    const long plane_groups = ((long)d->OH * PRr + 15) / 16;
    const bool few = (long)d->N * d->OD < 256;                   // one plane per item is already more than enough work per item
    if (d->CO == 8) {
        if (few && plane_groups <= 4) return launch_s1m<true, 1>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
        if (few && plane_groups <= 8) return launch_s1m<true, 2>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
        return launch_s1m<true, 4>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
    }
    if (d->CO == 16) {
        if (few && plane_groups <= 4) return launch_s1m<false, 1>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
        if (few && plane_groups <= 8) return launch_s1m<false, 2>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
        return launch_s1m<false, 4>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
    }
    return -1;
}
