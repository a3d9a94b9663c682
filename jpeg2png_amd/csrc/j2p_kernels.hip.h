// jpeg2png_amd — gfx950 device code for the deblocking solver's hot path.
//
// Two device-wide phases per iteration (the global ||g|| of compute.c:209-211
// forces the split):
//
//   k_gradient : FISTA point y = x_k + factor*(x_k - x_{k-1}) formed on the fly
//                (compute.c:433-439), staged with a 2-pixel halo in LDS, then
//                the prob (compute.c:53-66), TV (compute.c:73-125) and TGV2
//                (compute.c:128-197) subgradients in GATHER form, g written to
//                HBM plus one double sum(g*g) per workgroup.
//   k_project  : norm + step (compute.c:200-216) fused in front of the 8x8
//                DCT -> clamp -> IDCT projection (compute.c:334-404); also
//                emits the next iteration's prob gradient block
//                IDCT((clamped - d*q)/q^2) (compute.c:47-51) so the clamped
//                coefficients never go to HBM.
//
// Bit-exactness rules (SURVEY.md §8a "exactness recipe"): this TU is compiled
// with -ffp-contract=off and no fast-math; `/` and sqrtf are the correctly
// rounded forms; the DCT butterflies evaluate CONSTANT*float products and their
// sums in double and round to float once per assignment exactly like
// ooura/dct.c:39-66,103-130; gather sums run in the reference's raster order.
//
// No MFMA anywhere: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace j2p {

constexpr int kMaxCh = 3;
constexpr int kHalo = 2;

// ---------------------------------------------------------------------------
// 8-point orthonormal DCT-II / DCT-III, one lane owns the whole 8-vector.
// ---------------------------------------------------------------------------
// sqrt(2/8)*cos(k*pi/16), sqrt(2/8)*sin(k*pi/16), and cos(pi/4) — the values of
// ooura/dct.c:24-31, kept as double so that products promote like the reference.
constexpr double K1c = 0.49039264020161522456, K1s = 0.09754516100806413392;
constexpr double K2c = 0.46193976625564337806, K2s = 0.19134171618254488586;
constexpr double K3c = 0.41573480615127261854, K3s = 0.27778511650980111237;
constexpr double K4 = 0.35355339059327376220, KH = 0.70710678118654752440;

__device__ __forceinline__ float mix_add(double ka, float a, double kb, float b)
{
        return (float)(ka * (double)a + kb * (double)b);
}
__device__ __forceinline__ float mix_sub(double ka, float a, double kb, float b)
{
        return (float)(ka * (double)a - kb * (double)b);
}
__device__ __forceinline__ float scale(double k, float a) { return (float)(k * (double)a); }

// one pass of dct8x8s (ooura/dct.c:103-130)
__device__ __forceinline__ void fdct8(float (&v)[8])
{
        float e0 = v[0] + v[7], o0 = v[0] - v[7];
        float e1 = v[2] + v[5], o1 = v[2] - v[5];
        float e2 = v[4] + v[3], o2 = v[4] - v[3];
        float e3 = v[6] + v[1], o3 = v[6] - v[1];
        float p = e0 + e2, q = e1 + e3;
        v[0] = scale(K4, p + q);
        v[4] = scale(K4, p - q);
        p = e0 - e2;
        q = e1 - e3;
        v[2] = mix_sub(K2c, p, K2s, q);
        v[6] = mix_add(K2c, q, K2s, p);
        float r = scale(KH, o1 - o3);
        float t = scale(KH, o1 + o3);
        float u3 = t - o2;
        float u1 = t + o2;
        float w3 = o0 - r;
        float w1 = o0 + r;
        v[1] = mix_sub(K1c, w1, K1s, u1);
        v[7] = mix_add(K1c, u1, K1s, w1);
        v[3] = mix_sub(K3c, w3, K3s, u3);
        v[5] = mix_add(K3c, u3, K3s, w3);
}

// one pass of idct8x8s (ooura/dct.c:39-66)
__device__ __forceinline__ void idct8(float (&v)[8])
{
        float a1 = mix_add(K1c, v[1], K1s, v[7]);
        float b1 = mix_sub(K1c, v[7], K1s, v[1]);
        float a3 = mix_add(K3c, v[3], K3s, v[5]);
        float b3 = mix_sub(K3c, v[5], K3s, v[3]);
        float dr = a1 - a3;
        float di = b1 + b3;
        a1 = a1 + a3;
        b3 = b3 - b1;
        b1 = scale(KH, dr + di);
        a3 = scale(KH, dr - di);
        float cr = mix_add(K2c, v[2], K2s, v[6]);
        float ci = mix_sub(K2c, v[6], K2s, v[2]);
        float s0 = scale(K4, v[0] + v[4]);
        float d0 = scale(K4, v[0] - v[4]);
        float m2r = s0 - cr;
        float m2i = d0 - ci;
        s0 = s0 + cr;
        d0 = d0 + ci;
        v[0] = s0 + a1;
        v[7] = s0 - a1;
        v[2] = d0 + b1;
        v[5] = d0 - b1;
        v[4] = m2r - b3;
        v[3] = m2r + b3;
        v[6] = m2i - a3;
        v[1] = m2i + a3;
}

// ---------------------------------------------------------------------------
// 8x8 transpose inside each group of 8 lanes through wave-private LDS.
// Lane (b = lane>>3, j = lane&7) owns 8 values v[0..7] of line j of block b and
// receives element j of every line: out[i] = v_of_lane(b,i)[j].
// Layout b*104 + line*12 + elem: the two 16-byte stores of a lane group hit 32
// distinct banks (12*j mod 32 covers all 4-bank slots), and the dword reads of a
// 32-lane half hit 32 distinct banks (104 mod 32 = 8).
// ---------------------------------------------------------------------------
constexpr int kTpLine = 12, kTpBlock = 104, kTpWave = 8 * kTpBlock;  // floats

__device__ __forceinline__ void transpose8(float (&v)[8], float *scratch, int lane)
{
        const int b = lane >> 3, j = lane & 7;
        float4 *dst = reinterpret_cast<float4 *>(scratch + b * kTpBlock + j * kTpLine);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float *src = scratch + b * kTpBlock + j;
#pragma unroll
        for(int i = 0; i < 8; i++) { v[i] = src[i * kTpLine]; }
        __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Kernel arguments
// ---------------------------------------------------------------------------
struct ChanDev {
        float *xcur;        // x_k, own row 0 (halo rows at negative offsets)
        float *xprev;       // x_{k-1}; receives x_{k+1}
        float *grad;        // objective gradient, own rows
        float *pg;          // carried prob-gradient state, coefficient raster, band-local
        const int16_t *d;   // quantised coefficients, block-major, band-local
        const float *q;     // 64 floats
        unsigned cw, ch;    // coefficient plane size (whole image)
        unsigned ws, hs;    // subsampling
        unsigned crow0;     // first coefficient row held in d / pg  (= row0 / hs clipped)
        float p_alpha;      // pweight*2*255*sqrtf(2)  (compute.c:245)
        int prob_on;        // pweight != 0
};

struct Geo {
        unsigned W, H;      // canvas
        unsigned row0;      // first canvas row of the band
        unsigned rows;      // band rows
        unsigned ntx;       // gradient tiles per tile row
};

struct GradArgs {
        ChanDev ch[kMaxCh];
        Geo geo;
        float factor;       // FISTA (t-1)/tnext  (compute.c:432)
        float a_tv;         // 1/sqrt(nchannel)   (compute.c:90)
        float a_tgv;        // alpha/sqrt(nchannel) (compute.c:154)
        double *part_g2;    // [c][local tile row][tile col]
        double *part_tv;    // [local tile row][tile col][2]  (LOG only)
};

struct ProjArgs {
        ChanDev ch[kMaxCh];
        Geo geo;
        float factor;
        float step;         // radius / sqrtf(1 + iterations)  (compute.c:443)
        const float *norm;  // [c] ||g||  (compute.c:210)
        double *part_prob;  // [c][strip]  (LOG only)
        unsigned strips_per_chan;   // stride of part_prob
};

// gradient tile
constexpr int kTX = 64, kTY = 16;
constexpr int kYW = kTX + 2 * kHalo, kYH = kTY + 2 * kHalo;   // 68 x 20 pixels of y
constexpr int kSW = kTX + 2, kSH = kTY + 2;                   // 66 x 18 sources

// deterministic block-wide sum of one double per thread (256 threads); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *red /* >= 4 doubles */)
{
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if(lane == 0) { red[wave] = v; }
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------
// Phase A: gradient
// ---------------------------------------------------------------------------
template <int NCH, bool TGV, bool LOG>
__global__ __launch_bounds__(256) void k_gradient(GradArgs a)
{
        extern __shared__ __attribute__((aligned(16))) float smem[];
        float *Y = smem;                                   // [NCH][kYH][kYW]
        float *N1 = Y + NCH * kYH * kYW;                   // [kSH][kSW]
        float *TG = N1 + kSH * kSW;                        // [NCH][4][kSH][kSW]  (TGV only)
        double *red = reinterpret_cast<double *>(TG + (TGV ? NCH * 4 * kSH * kSW : 0)) ;

        const int W = (int)a.geo.W, H = (int)a.geo.H;
        const int tx0 = (int)blockIdx.x * kTX;
        const int ty0 = (int)blockIdx.y * kTY;             // band-local
        const int gy0 = (int)a.geo.row0 + ty0;             // canvas row of tile row 0
        const int tid = (int)threadIdx.x;

        // ---- stage 1: FISTA point for tile + halo (compute.c:433-439) ----
        for(int i = tid; i < kYH * kYW; i += 256) {
                const int lx = i % kYW - kHalo, ly = i / kYW - kHalo;
                const int gx = tx0 + lx, gy = gy0 + ly;
                const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                const ptrdiff_t off = (ptrdiff_t)(ty0 + ly) * W + gx;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        float y = 0.f;
                        if(in) {
                                const float xc = a.ch[c].xcur[off], xp = a.ch[c].xprev[off];
                                y = xc + a.factor * (xc - xp);
                        }
                        Y[c * kYH * kYW + i] = y;
                }
        }
        __syncthreads();

        auto yat = [&](int c, int lx, int ly) -> float {
                return Y[c * kYH * kYW + (ly + kHalo) * kYW + lx + kHalo];
        };
        // forward differences with the reference's border rule (compute.c:79,81)
        auto dxf = [&](int c, int lx, int ly) -> float {
                return tx0 + lx >= W - 1 ? 0.f : yat(c, lx + 1, ly) - yat(c, lx, ly);
        };
        auto dyf = [&](int c, int lx, int ly) -> float {
                return gy0 + ly >= H - 1 ? 0.f : yat(c, lx, ly + 1) - yat(c, lx, ly);
        };

        // ---- stage 2: per-source norms and TGV2 terms for tile + 1-pixel ring ----
        double tv_acc = 0., tv2_acc = 0.;
        for(int i = tid; i < kSH * kSW; i += 256) {
                const int lx = i % kSW - 1, ly = i / kSW - 1;
                const int gx = tx0 + lx, gy = gy0 + ly;
                const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                const bool own = lx >= 0 && lx < kTX && ly >= 0 && ly < kTY;
                float n1 = 0.f;
                if(in) {
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                const float gxv = dxf(c, lx, ly), gyv = dyf(c, lx, ly);
                                n1 += gxv * gxv;
                                n1 += gyv * gyv;
                        }
                        n1 = sqrtf(n1);
                        if(LOG && own) { tv_acc += (double)(a.a_tv * n1); }
                }
                N1[i] = n1;
                if(TGV) {
                        float xx[NCH], sy[NCH], yy[NCH];
                        float n2 = 0.f;
                        if(in) {
#pragma unroll
                                for(int c = 0; c < NCH; c++) {
                                        // backward differences of the forward differences (compute.c:136-146)
                                        const float gxv = dxf(c, lx, ly), gyv = dyf(c, lx, ly);
                                        xx[c] = gx == 0 ? 0.f : gxv - dxf(c, lx - 1, ly);
                                        const float gyx = gx == 0 ? 0.f : gyv - dyf(c, lx - 1, ly);
                                        const float gxy = gy == 0 ? 0.f : gxv - dxf(c, lx, ly - 1);
                                        yy[c] = gy == 0 ? 0.f : gyv - dyf(c, lx, ly - 1);
                                        sy[c] = (gxy + gyx) / 2.f;
                                        n2 += xx[c] * xx[c] + 2 * (sy[c] * sy[c]) + yy[c] * yy[c];
                                }
                                n2 = sqrtf(n2);
                                if(LOG && own) { tv2_acc += (double)(a.a_tgv * n2); }
                        }
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                float tA = 0.f, tB = 0.f, tC = 0.f, tO = 0.f;
                                if(in && n2 != 0.f) {
                                        // compute.c:165-183: a2 * (expr / n2), division first
                                        tA = a.a_tgv * ((sy[c] + xx[c]) / n2);                    // to (x-1,y), (x+1,y)
                                        tB = a.a_tgv * ((yy[c] + sy[c]) / n2);                    // to (x,y-1), (x,y+1)
                                        tC = a.a_tgv * ((-sy[c]) / n2);                           // to (x+1,y-1), (x-1,y+1)
                                        tO = a.a_tgv * (-(2 * xx[c] + 2 * sy[c] + 2 * yy[c]) / n2); // own
                                }
                                float *t = TG + (c * 4) * kSH * kSW + i;
                                t[0] = tA;
                                t[kSH * kSW] = tB;
                                t[2 * kSH * kSW] = tC;
                                t[3 * kSH * kSW] = tO;
                        }
                }
        }
        __syncthreads();

        // ---- stage 3: gather per target pixel, raster order of the sources ----
        double g2[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c++) { g2[c] = 0.; }
        const int lx = tid & 63;
        const int gx = tx0 + lx;
#pragma unroll
        for(int r = 0; r < kTY / 4; r++) {
                const int ly = (tid >> 6) + 4 * r;
                const int gy = gy0 + ly;
                if(gx < W && gy < H && ty0 + ly < (int)a.geo.rows) {
                        const int si = (ly + 1) * kSW + lx + 1;      // own source slot
                        const float nU = N1[si - kSW], nL = N1[si - 1], nO = N1[si];
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                const ChanDev &k = a.ch[c];
                                float g = 0.f;
                                if(k.prob_on && (unsigned)gx < k.cw * k.ws && (unsigned)gy < k.ch * k.hs) {
                                        const unsigned cy = (unsigned)gy / k.hs - k.crow0, cx = (unsigned)gx / k.ws;
                                        g += k.p_alpha * k.pg[(size_t)cy * k.cw + cx];
                                }
                                // TV (compute.c:97-104): (a*v)/n
                                if(nU != 0.f) { g += a.a_tv * dyf(c, lx, ly - 1) / nU; }
                                if(nL != 0.f) { g += a.a_tv * dxf(c, lx - 1, ly) / nL; }
                                if(nO != 0.f) { g += a.a_tv * -(dxf(c, lx, ly) + dyf(c, lx, ly)) / nO; }
                                if(TGV) {
                                        const float *t = TG + (c * 4) * kSH * kSW + si;
                                        const float *tA = t, *tB = t + kSH * kSW, *tC = t + 2 * kSH * kSW, *tO = t + 3 * kSH * kSW;
                                        g += tB[-kSW];          // (x,  y-1)
                                        g += tC[-kSW + 1];      // (x+1,y-1)
                                        g += tA[-1];            // (x-1,y)
                                        g += tO[0];             // own
                                        g += tA[1];             // (x+1,y)
                                        g += tC[kSW - 1];       // (x-1,y+1)
                                        g += tB[kSW];           // (x,  y+1)
                                }
                                k.grad[(size_t)(ty0 + ly) * W + gx] = g;
                                g2[c] += (double)(g * g);       // compute.c:203
                        }
                }
        }
        const size_t tile = (size_t)blockIdx.y * a.geo.ntx + blockIdx.x;
        const size_t ntiles = (size_t)gridDim.y * a.geo.ntx;
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                const double s = block_sum(g2[c], red);
                if(tid == 0) { a.part_g2[c * ntiles + tile] = s; }
        }
        if(LOG) {
                const double s1 = block_sum(tv_acc, red);
                const double s2 = block_sum(tv2_acc, red);
                if(tid == 0) {
                        a.part_tv[2 * tile] = s1;
                        a.part_tv[2 * tile + 1] = s2;
                }
        }
}


// ---------------------------------------------------------------------------
// Phase A, register-marching form (the one that ships).
//
// One wavefront owns a strip of 64*CPL consecutive columns (CPL columns per
// lane) and walks down RPW rows.  Everything the gather needs from neighbouring
// COLUMNS moves between lanes with DPP wave shifts; everything it needs from
// neighbouring ROWS is carried in registers from one loop trip to the next:
// no LDS, no barriers.  The two outermost columns on each side of the strip are
// halo (loaded and differenced, never stored), so a strip yields 64*CPL-4 output
// columns; likewise each strip recomputes the source terms of one row above and
// below its RPW rows.
//
// Per loop trip (source row r, with y rows r-1, r, r+1 in registers):
//   S_r  = per-pixel terms every neighbour will need from pixel (x,r):
//          TV  : tvx=(a*gx)/n  tvy=(a*gy)/n  tvo=(a*-(gx+gy))/n       (compute.c:98-103)
//          TGV2: A=a2*((s+gxx)/n2) B=a2*((gyy+s)/n2) C=a2*((-s)/n2)
//                O=a2*(-(2gxx+2s+2gyy)/n2)                              (compute.c:165-182)
//   then target row t=r-1 is complete:
//          g = p_alpha*P  + S_{t-1}.tvy + S_t.tvx(x-1) + S_t.tvo
//              + S_{t-1}.B + S_{t-1}.C(x+1) + S_t.A(x-1) + S_t.O + S_t.A(x+1)
//              + S_{t+1}.C(x-1) + S_{t+1}.B          (the reference's raster order)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lane_from_left(float v)    // value held by lane-1 (0 in lane 0)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_from_right(float v)   // value held by lane+1 (0 in lane 63)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int CPL>
struct Cols {
        float v[CPL];
};
// column x-1 / x+1 of a per-lane group of CPL adjacent columns
template <int CPL>
__device__ __forceinline__ Cols<CPL> left_of(const Cols<CPL> &a)
{
        Cols<CPL> r;
        r.v[0] = lane_from_left(a.v[CPL - 1]);
#pragma unroll
        for(int j = 1; j < CPL; j++) { r.v[j] = a.v[j - 1]; }
        return r;
}
template <int CPL>
__device__ __forceinline__ Cols<CPL> right_of(const Cols<CPL> &a)
{
        Cols<CPL> r;
        r.v[CPL - 1] = lane_from_right(a.v[0]);
#pragma unroll
        for(int j = 0; j < CPL - 1; j++) { r.v[j] = a.v[j + 1]; }
        return r;
}

constexpr int kRPW = 32;         // rows per wavefront strip (multiple of kTY)

template <int NCH, bool TGV, bool LOG, int CPL>
__global__ __launch_bounds__(256) void k_gradient_march(GradArgs a)
{
        constexpr int VALIDW = 64 * CPL - 4;
        const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
        const int wcol = (int)blockIdx.x * 4 + wave;
        if(wcol >= (int)a.geo.ntx) { return; }
        const int W = (int)a.geo.W, H = (int)a.geo.H;
        const int rows = (int)a.geo.rows, row0 = (int)a.geo.row0;
        const int t0 = (int)blockIdx.y * kRPW;                 // band-local target rows [t0, t1)
        const int t1 = t0 + kRPW < rows ? t0 + kRPW : rows;
        const int xl = wcol * VALIDW - 2 + lane * CPL;         // canvas column of v[0]

        bool col_in[CPL], col_own[CPL];
#pragma unroll
        for(int j = 0; j < CPL; j++) {
                const int x = xl + j, rel = lane * CPL + j;
                col_in[j] = x >= 0 && x < W;
                col_own[j] = col_in[j] && rel >= 2 && rel < 64 * CPL - 2;
        }

        // FISTA point of one row for this lane's columns (compute.c:433-439); 0 outside the image
        auto load_y = [&](int lr, Cols<CPL> (&y)[NCH]) {
                const int gr = row0 + lr;
                const bool rin = gr >= 0 && gr < H;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const ptrdiff_t off = (ptrdiff_t)lr * W + xl;
                        if(CPL == 2) {
                                float2 xc = make_float2(0.f, 0.f), xp = make_float2(0.f, 0.f);
                                if(rin && col_in[0]) {          // W and xl are even: the pair is in or out together
                                        xc = *reinterpret_cast<const float2 *>(a.ch[c].xcur + off);
                                        xp = *reinterpret_cast<const float2 *>(a.ch[c].xprev + off);
                                }
                                y[c].v[0] = xc.x + a.factor * (xc.x - xp.x);
                                y[c].v[CPL - 1] = xc.y + a.factor * (xc.y - xp.y);
                        } else {
#pragma unroll
                                for(int j = 0; j < CPL; j++) {
                                        float xc = 0.f, xp = 0.f;
                                        if(rin && col_in[j]) {
                                                xc = a.ch[c].xcur[off + j];
                                                xp = a.ch[c].xprev[off + j];
                                        }
                                        y[c].v[j] = xc + a.factor * (xc - xp);
                                }
                        }
                }
        };
        // forward differences of row r given rows r and r+1 (compute.c:79,81)
        auto diffs = [&](int gr, const Cols<CPL> (&yc)[NCH], const Cols<CPL> (&yn)[NCH], Cols<CPL> (&gx)[NCH], Cols<CPL> (&gy)[NCH]) {
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const Cols<CPL> yr = right_of<CPL>(yc[c]);
#pragma unroll
                        for(int j = 0; j < CPL; j++) {
                                gx[c].v[j] = xl + j >= W - 1 ? 0.f : yr.v[j] - yc[c].v[j];
                                gy[c].v[j] = gr >= H - 1 ? 0.f : yn[c].v[j] - yc[c].v[j];
                        }
                }
        };

        Cols<CPL> yc[NCH], yn[NCH], gxp[NCH], gyp[NCH];
        {
                Cols<CPL> ym[NCH];
                load_y(t0 - 2, ym);
                load_y(t0 - 1, yc);
                load_y(t0, yn);
                diffs(row0 + t0 - 2, ym, yc, gxp, gyp);
        }
        // carried source terms: row t (p1*) and the part of row t-1 the next target still needs (p2*)
        Cols<CPL> p1_tvxL[NCH], p1_tvo[NCH], p1_tvy[NCH], p1_AL[NCH], p1_O[NCH], p1_AR[NCH], p1_B[NCH], p1_CR[NCH];
        Cols<CPL> p2_tvy[NCH], p2_B[NCH], p2_CR[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c++) {
#pragma unroll
                for(int j = 0; j < CPL; j++) {
                        p1_tvxL[c].v[j] = p1_tvo[c].v[j] = p1_tvy[c].v[j] = 0.f;
                        p1_AL[c].v[j] = p1_O[c].v[j] = p1_AR[c].v[j] = p1_B[c].v[j] = p1_CR[c].v[j] = 0.f;
                        p2_tvy[c].v[j] = p2_B[c].v[j] = p2_CR[c].v[j] = 0.f;
                }
        }
        double g2[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c++) { g2[c] = 0.; }
        double tv_acc = 0., tv2_acc = 0.;
        const size_t ntiles_row = a.geo.ntx;
        const size_t nparts = (size_t)((rows + kTY - 1) / kTY) * ntiles_row;

#pragma unroll 1
        for(int r = t0 - 1; r <= t1; r++) {
                const int gr = row0 + r;
                Cols<CPL> ynn[NCH];
                load_y(r + 2 <= t1 + 1 ? r + 2 : -(int)kHalo - 1 - row0, ynn);   // prefetch for the next trip; past the
                                                                                 // halo nothing is needed (maps to a row < 0)
                // ---- source terms of row r ----
                const bool rin = gr >= 0 && gr < H;
                Cols<CPL> gx[NCH], gy[NCH];
                diffs(gr, yc, yn, gx, gy);
                Cols<CPL> n1;
#pragma unroll
                for(int j = 0; j < CPL; j++) {
                        float n = 0.f;
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                n += gx[c].v[j] * gx[c].v[j];
                                n += gy[c].v[j] * gy[c].v[j];
                        }
                        n1.v[j] = rin && col_in[j] ? sqrtf(n) : 0.f;
                        if(LOG && col_own[j] && r >= t0 && r < t1) { tv_acc += (double)(a.a_tv * n1.v[j]); }
                }
                Cols<CPL> s_tvx[NCH], s_tvy[NCH], s_tvo[NCH];
#pragma unroll
                for(int c = 0; c < NCH; c++) {
#pragma unroll
                        for(int j = 0; j < CPL; j++) {
                                const float n = n1.v[j];
                                const bool nz = n != 0.f;
                                s_tvx[c].v[j] = nz ? a.a_tv * gx[c].v[j] / n : 0.f;
                                s_tvy[c].v[j] = nz ? a.a_tv * gy[c].v[j] / n : 0.f;
                                s_tvo[c].v[j] = nz ? a.a_tv * -(gx[c].v[j] + gy[c].v[j]) / n : 0.f;
                        }
                }
                Cols<CPL> s_A[NCH], s_B[NCH], s_C[NCH], s_O[NCH];
                if(TGV) {
                        Cols<CPL> xx[NCH], sy[NCH], yy[NCH];
                        Cols<CPL> n2;
#pragma unroll
                        for(int j = 0; j < CPL; j++) { n2.v[j] = 0.f; }
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                const Cols<CPL> gxl = left_of<CPL>(gx[c]), gyl = left_of<CPL>(gy[c]);
#pragma unroll
                                for(int j = 0; j < CPL; j++) {
                                        const bool x0 = xl + j == 0;
                                        xx[c].v[j] = x0 ? 0.f : gx[c].v[j] - gxl.v[j];
                                        const float gyx = x0 ? 0.f : gy[c].v[j] - gyl.v[j];
                                        const float gxy = gr == 0 ? 0.f : gx[c].v[j] - gxp[c].v[j];
                                        yy[c].v[j] = gr == 0 ? 0.f : gy[c].v[j] - gyp[c].v[j];
                                        sy[c].v[j] = (gxy + gyx) / 2.f;
                                        n2.v[j] += xx[c].v[j] * xx[c].v[j] + 2 * (sy[c].v[j] * sy[c].v[j]) + yy[c].v[j] * yy[c].v[j];
                                }
                        }
#pragma unroll
                        for(int j = 0; j < CPL; j++) {
                                n2.v[j] = rin && col_in[j] ? sqrtf(n2.v[j]) : 0.f;
                                if(LOG && col_own[j] && r >= t0 && r < t1) { tv2_acc += (double)(a.a_tgv * n2.v[j]); }
                        }
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
#pragma unroll
                                for(int j = 0; j < CPL; j++) {
                                        const float n = n2.v[j];
                                        const bool nz = n != 0.f;
                                        s_A[c].v[j] = nz ? a.a_tgv * ((sy[c].v[j] + xx[c].v[j]) / n) : 0.f;
                                        s_B[c].v[j] = nz ? a.a_tgv * ((yy[c].v[j] + sy[c].v[j]) / n) : 0.f;
                                        s_C[c].v[j] = nz ? a.a_tgv * ((-sy[c].v[j]) / n) : 0.f;
                                        s_O[c].v[j] = nz ? a.a_tgv * (-(2 * xx[c].v[j] + 2 * sy[c].v[j] + 2 * yy[c].v[j]) / n) : 0.f;
                                }
                        }
                }
                // ---- target row t = r-1 ----
                const int t = r - 1;
                if(t >= t0) {
                        const int gt = row0 + t;
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                const ChanDev &k = a.ch[c];
                                Cols<CPL> g;
#pragma unroll
                                for(int j = 0; j < CPL; j++) { g.v[j] = 0.f; }
                                if(k.prob_on && (unsigned)gt < k.ch * k.hs) {
                                        const size_t prow = (size_t)((unsigned)gt / k.hs - k.crow0) * k.cw;
#pragma unroll
                                        for(int j = 0; j < CPL; j++) {
                                                const int x = xl + j;
                                                if(col_own[j] && (unsigned)x < k.cw * k.ws) {
                                                        g.v[j] += k.p_alpha * k.pg[prow + (unsigned)x / k.ws];
                                                }
                                        }
                                }
                                Cols<CPL> cL, aL_unused;
                                (void)aL_unused;
                                if(TGV) { cL = left_of<CPL>(s_C[c]); }
#pragma unroll
                                for(int j = 0; j < CPL; j++) {
                                        float v = g.v[j];
                                        v += p2_tvy[c].v[j];        // TV from (x, t-1)
                                        v += p1_tvxL[c].v[j];       // TV from (x-1, t)
                                        v += p1_tvo[c].v[j];        // TV own
                                        if(TGV) {
                                                v += p2_B[c].v[j];  // (x,   t-1)
                                                v += p2_CR[c].v[j]; // (x+1, t-1)
                                                v += p1_AL[c].v[j]; // (x-1, t)
                                                v += p1_O[c].v[j];  // own
                                                v += p1_AR[c].v[j]; // (x+1, t)
                                                v += cL.v[j];       // (x-1, t+1)
                                                v += s_B[c].v[j];   // (x,   t+1)
                                        }
                                        g.v[j] = v;
                                        if(col_own[j]) { g2[c] += (double)(v * v); }    // compute.c:203
                                }
                                float *dst = k.grad + (size_t)t * W + xl;
                                if(CPL == 2) {
                                        if(col_own[0]) { *reinterpret_cast<float2 *>(dst) = make_float2(g.v[0], g.v[CPL - 1]); }
                                } else {
#pragma unroll
                                        for(int j = 0; j < CPL; j++) {
                                                if(col_own[j]) { dst[j] = g.v[j]; }
                                        }
                                }
                        }
                        // one partial per 16-row tile row and strip: the granularity of the GPU-count
                        // invariant norm reduction
                        if((t & (kTY - 1)) == kTY - 1 || t == t1 - 1) {
#pragma unroll
                                for(int c = 0; c < NCH; c++) {
                                        double v = g2[c];
#pragma unroll
                                        for(int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); }
                                        if(lane == 0) { a.part_g2[c * nparts + (size_t)(t / kTY) * ntiles_row + wcol] = v; }
                                        g2[c] = 0.;
                                }
                        }
                }
                // ---- rotate the carried state ----
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        p2_tvy[c] = p1_tvy[c];
                        p1_tvy[c] = s_tvy[c];
                        p1_tvxL[c] = left_of<CPL>(s_tvx[c]);
                        p1_tvo[c] = s_tvo[c];
                        if(TGV) {
                                p2_B[c] = p1_B[c];
                                p2_CR[c] = p1_CR[c];
                                p1_B[c] = s_B[c];
                                p1_CR[c] = right_of<CPL>(s_C[c]);
                                p1_AL[c] = left_of<CPL>(s_A[c]);
                                p1_AR[c] = right_of<CPL>(s_A[c]);
                                p1_O[c] = s_O[c];
                        }
                        gxp[c] = gx[c];
                        gyp[c] = gy[c];
                        yc[c] = yn[c];
                        yn[c] = ynn[c];
                }
        }
        if(LOG) {
#pragma unroll
                for(int off = 32; off > 0; off >>= 1) {
                        tv_acc += __shfl_down(tv_acc, off, 64);
                        tv2_acc += __shfl_down(tv2_acc, off, 64);
                }
                if(lane == 0) {
                        const size_t w = (size_t)blockIdx.y * ntiles_row + wcol;
                        a.part_tv[2 * w] = tv_acc;
                        a.part_tv[2 * w + 1] = tv2_acc;
                }
        }
}

template <int NCH, bool TGV>
constexpr size_t gradient_lds_bytes()
{
        return sizeof(float) * (NCH * kYH * kYW + kSH * kSW + (TGV ? NCH * 4 * kSH * kSW : 0)) + 4 * sizeof(double) + 16;
}

// ---------------------------------------------------------------------------
// Norm reduction.  Level 1: per row-of-tiles sums (sequential over tile columns).
// Level 2: pairwise tree over the row-of-tiles array padded to a power of two —
// a function of the GLOBAL array only, so the value does not depend on how many
// GPUs produced the rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowsums(const double *part, double *rowsum, unsigned ntx, unsigned nrows_local, unsigned nch)
{
        // part: [c][local tile row][tile col] -> rowsum: [local tile row][c]  (tile-row major, so that
        // concatenating the bands of consecutive GPUs yields the global array)
        const unsigned i = blockIdx.x * 256 + threadIdx.x;
        if(i >= nrows_local * nch) { return; }
        const unsigned c = i / nrows_local, r = i % nrows_local;
        const double *p = part + (size_t)i * ntx;
        double s = 0.;
        for(unsigned t = 0; t < ntx; t++) { s += p[t]; }
        rowsum[(size_t)r * nch + c] = s;
}

constexpr int kMaxTileRows = 4096;   // canvas height <= 65536 (JPEG limit) / kTY

__device__ __forceinline__ double tree_sum_lds(double *buf, unsigned n, unsigned P)
{
        // buf[0..P) holds the n values followed by zeros; all 256 threads participate
        for(unsigned s = P >> 1; s > 0; s >>= 1) {
                __syncthreads();
                for(unsigned i = threadIdx.x; i < s; i += 256) { buf[i] = buf[i] + buf[i + s]; }
        }
        __syncthreads();
        (void)n;
        return buf[0];
}

// one block per channel: norm[c] = sqrtf((float) sum)   (compute.c:200-207)
__global__ __launch_bounds__(256) void k_norm_finish(const double *rowsum_all, unsigned nrows_global, unsigned nch, float *norm)
{
        extern __shared__ __attribute__((aligned(16))) float smem[];
        double *buf = reinterpret_cast<double *>(smem);
        unsigned P = 1;
        while(P < nrows_global) { P <<= 1; }
        const double *src = rowsum_all + blockIdx.x;      // [tile row][c]
        for(unsigned i = threadIdx.x; i < P; i += 256) { buf[i] = i < nrows_global ? src[(size_t)i * nch] : 0.; }
        const double s = tree_sum_lds(buf, nrows_global, P);
        if(threadIdx.x == 0) { norm[blockIdx.x] = sqrtf((float)s); }
}

// log sums: tv / tv2 from the gradient tiles and per-channel prob distance from the
// projection strips, plain fixed-order tree (values only feed the CSV log)
__global__ __launch_bounds__(256) void k_log_sums(const double *part_tv, unsigned ntiles,
                                                  const double *part_prob, unsigned nstrips, unsigned strips_stride,
                                                  unsigned nch, double *out /* [2 + kMaxCh] */, int which)
{
        __shared__ double red[4];
        if(which == 0) {
                for(int k = 0; k < 2; k++) {
                        double v = 0.;
                        for(unsigned i = threadIdx.x; i < ntiles; i += 256) { v += part_tv[2 * (size_t)i + k]; }
                        const double s = block_sum(v, red);
                        if(threadIdx.x == 0) { out[k] = s; }
                }
        } else {
                for(unsigned c = 0; c < nch; c++) {
                        double v = 0.;
                        for(unsigned i = threadIdx.x; i < nstrips; i += 256) { v += part_prob[(size_t)c * strips_stride + i]; }
                        const double s = block_sum(v, red);
                        if(threadIdx.x == 0) { out[2 + c] = s; }
                }
        }
}

// ---------------------------------------------------------------------------
// Phase B: step + projection.  One wavefront = one strip of 8 coefficient
// blocks (64 coefficient columns x 8 coefficient rows); a lane owns one
// coefficient column for the column passes and one block row for the row passes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float stepped(const ChanDev &k, ptrdiff_t off, float factor, float step, float norm)
{
        const float xc = k.xcur[off], xp = k.xprev[off];
        const float y = xc + factor * (xc - xp);                 // compute.c:435
        if(norm != 0.f) { return y - step * (k.grad[off] / norm); }   // compute.c:213
        return y;
}

template <bool LOG>
__global__ __launch_bounds__(256) void k_project(ProjArgs a)
{
        __shared__ __attribute__((aligned(16))) float tp[4 * kTpWave];
        __shared__ __attribute__((aligned(16))) float qs[64];

        const int c = (int)blockIdx.z;
        const ChanDev &k = a.ch[c];
        const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
        if(threadIdx.x < 64) { qs[threadIdx.x] = k.q[threadIdx.x]; }
        __syncthreads();

        const unsigned W = a.geo.W, H = a.geo.H;
        const unsigned ws = k.ws, hs = k.hs;
        const unsigned strips_x = (W + 64 * ws - 1) / (64 * ws);      // strips across the canvas
        const unsigned brows = (a.geo.rows + 8 * hs - 1) / (8 * hs);  // block rows in the band
        const unsigned strip = blockIdx.x * 4 + wave;
        if(strip >= strips_x * brows) { return; }
        const unsigned by = strip / strips_x, sx = strip % strips_x;
        float *scratch = tp + wave * kTpWave;

        const float norm = a.norm[c];
        const unsigned cx = sx * 64 + lane;                           // coefficient column of this lane
        const unsigned cy0 = (a.geo.row0 / hs) + by * 8;              // first coefficient row (global)
        const bool covered = cx < k.cw && cy0 < k.ch;                 // block-granular: cw, ch multiples of 8
        const unsigned ly0 = by * 8 * hs;                             // band-local canvas row
        const bool direct = ws == 1 && hs == 1;

        float v[8];
        if(direct) {
                const bool inside = cx < W;
#pragma unroll
                for(int r = 0; r < 8; r++) {
                        v[r] = 0.f;
                        if(inside && ly0 + r < a.geo.rows) {
                                const ptrdiff_t off = (ptrdiff_t)(ly0 + r) * W + cx;
                                v[r] = stepped(k, off, a.factor, a.step, norm);
                                if(!covered) { k.xprev[off] = v[r]; }  // stepped but never projected (SURVEY §7 hard part 5)
                        }
                }
        } else {
                // block-mean down-sample (compute.c:348-360); residual handled in the second sweep
#pragma unroll 1
                for(int r = 0; r < 8; r++) {
                        float mean = 0.f;
                        for(unsigned sy = 0; sy < hs; sy++) {
                                for(unsigned sxx = 0; sxx < ws; sxx++) {
                                        const unsigned x = cx * ws + sxx, ly = ly0 + r * hs + sy;
                                        if(x < W && ly < a.geo.rows) {
                                                const ptrdiff_t off = (ptrdiff_t)ly * W + x;
                                                const float f = stepped(k, off, a.factor, a.step, norm);
                                                if(covered) { mean += f; } else { k.xprev[off] = f; }
                                        }
                                }
                        }
                        v[r] = mean / (float)(ws * hs);
                }
        }
        float mean_old[8];
#pragma unroll
        for(int r = 0; r < 8; r++) { mean_old[r] = v[r]; }

        // ---- forward DCT: columns pass (lane = column), transpose, rows pass (lane = block row) ----
        fdct8(v);
        transpose8(v, scratch, lane);
        fdct8(v);

        // ---- clamp to the quantisation interval (compute.c:323-331) and prob state (compute.c:47-49) ----
        const int b = lane >> 3, rr = lane & 7;                      // block in strip, row in block
        const unsigned bx = sx * 8 + b;
        const bool bcov = bx * 8 < k.cw && cy0 < k.ch;
        float e[8];
        double dist = 0.;
        {
                short dd[8];
                if(bcov) {
                        const size_t blk = (size_t)(cy0 / 8 - k.crow0 / 8) * (k.cw / 8) + bx;
                        const int4 raw = *reinterpret_cast<const int4 *>(k.d + blk * 64 + rr * 8);
                        dd[0] = (short)(raw.x & 0xffff); dd[1] = (short)(raw.x >> 16);
                        dd[2] = (short)(raw.y & 0xffff); dd[3] = (short)(raw.y >> 16);
                        dd[4] = (short)(raw.z & 0xffff); dd[5] = (short)(raw.z >> 16);
                        dd[6] = (short)(raw.w & 0xffff); dd[7] = (short)(raw.w >> 16);
                } else {
#pragma unroll
                        for(int u = 0; u < 8; u++) { dd[u] = 0; }
                }
#pragma unroll
                for(int u = 0; u < 8; u++) {
                        const float q = qs[rr * 8 + u];
                        const float df = (float)dd[u];
                        const float lo = (df - 0.5f) * q, hi = (df + 0.5f) * q;
                        float x = v[u];
                        x = x > hi ? hi : (x < lo ? lo : x);
                        v[u] = x;
                        const float t = x - df * q;
                        if(LOG) { const float tq = t / q; dist += (double)(tq * tq); }   // compute_simd_step.c:22-26
                        e[u] = t / (q * q);
                }
        }

        // ---- inverse DCT of the clamped coefficients ----
        transpose8(v, scratch, lane);
        idct8(v);
        transpose8(v, scratch, lane);
        idct8(v);                                                    // lane (b, rr): row rr of block b

        if(direct) {
                if(bcov && ly0 + rr < a.geo.rows) {
                        float4 *dst = reinterpret_cast<float4 *>(k.xprev + (size_t)(ly0 + rr) * W + bx * 8);
                        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
        } else {
                // back to lane = coefficient column, then add the new mean onto the residual (compute.c:365,398)
                transpose8(v, scratch, lane);
                if(covered) {
#pragma unroll 1
                        for(int r = 0; r < 8; r++) {
                                for(unsigned sy = 0; sy < hs; sy++) {
                                        for(unsigned sxx = 0; sxx < ws; sxx++) {
                                                const unsigned x = cx * ws + sxx, ly = ly0 + r * hs + sy;
                                                const ptrdiff_t off = (ptrdiff_t)ly * W + x;
                                                float f = stepped(k, off, a.factor, a.step, norm);
                                                f = f - mean_old[r];
                                                k.xprev[off] = f + v[r];
                                        }
                                }
                        }
                }
        }

        // ---- next iteration's prob gradient block (compute.c:49-51) ----
        if(k.prob_on) {
                transpose8(e, scratch, lane);
                idct8(e);
                transpose8(e, scratch, lane);
                idct8(e);
                if(bcov) {
                        float4 *dst = reinterpret_cast<float4 *>(k.pg + (size_t)(cy0 - k.crow0 + rr) * k.cw + bx * 8);
                        dst[0] = make_float4(e[0], e[1], e[2], e[3]);
                        dst[1] = make_float4(e[4], e[5], e[6], e[7]);
                }
                if(LOG) {
                        if(!bcov) { dist = 0.; }
#pragma unroll
                        for(int off = 32; off > 0; off >>= 1) { dist += __shfl_down(dist, off, 64); }
                        if(lane == 0) { a.part_prob[(size_t)c * a.strips_per_chan + strip] = dist; }
                }
        }
}

// ---------------------------------------------------------------------------
// aux_init on the device (compute.c:295-309): x_k = x_{k-1} = replicate-upsample,
// prob state = 0 (cos = d*q  =>  IDCT(0)).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_state(ChanDev k, Geo geo, const float *decoded /* band-local coefficient raster */,
                                                    int fill_halo)
{
        // rows [-halo, rows+halo) of the band when fill_halo, else own rows only
        const int lo = fill_halo ? -kHalo : 0, hi = (int)geo.rows + (fill_halo ? kHalo : 0);
        const size_t n = (size_t)(hi - lo) * geo.W;
        for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
                const int ly = lo + (int)(i / geo.W);
                const unsigned x = (unsigned)(i % geo.W);
                const long gy = (long)geo.row0 + ly;
                if(gy < 0 || gy >= (long)geo.H) { continue; }
                unsigned cy = (unsigned)gy / k.hs, cx = x / k.ws;
                if(cy > k.ch - 1) { cy = k.ch - 1; }
                if(cx > k.cw - 1) { cx = k.cw - 1; }
                const float v = decoded[(size_t)(cy - k.crow0) * k.cw + cx];
                const ptrdiff_t off = (ptrdiff_t)ly * geo.W + x;
                k.xcur[off] = v;
                k.xprev[off] = v;
        }
}

__global__ __launch_bounds__(256) void k_fill_zero(float *p, size_t n)
{
        for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { p[i] = 0.f; }
}

// decode_coefficients + unbox (jpeg.c:83-92, box.c:5-19): one wavefront per 8 blocks
__global__ __launch_bounds__(256) void k_decode(const int16_t *d, const float *q, float *out, unsigned cw, unsigned nblocks_y)
{
        __shared__ __attribute__((aligned(16))) float tp[4 * kTpWave];
        const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
        const unsigned bw = cw / 8;
        const unsigned groups_x = (bw + 7) / 8;
        const unsigned grp = blockIdx.x * 4 + wave;
        if(grp >= groups_x * nblocks_y) { return; }
        const unsigned by = grp / groups_x, bx = (grp % groups_x) * 8 + (lane >> 3);
        const int rr = lane & 7;
        const bool ok = bx < bw;
        float v[8];
#pragma unroll
        for(int u = 0; u < 8; u++) {
                const int dv = ok ? (int)d[((size_t)by * bw + bx) * 64 + rr * 8 + u] : 0;
                v[u] = (float)(dv * (int)(unsigned)q[rr * 8 + u]);   // int product, then to float (jpeg.c:88)
        }
        float *scratch = tp + wave * kTpWave;
        transpose8(v, scratch, lane);      // lane = column
        idct8(v);
        transpose8(v, scratch, lane);      // lane = row
        idct8(v);
        if(ok) {
                float4 *dst = reinterpret_cast<float4 *>(out + (size_t)(by * 8 + rr) * cw + bx * 8);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
}

// plain 8x8 transforms on a block-major array (parity tests of the butterflies)
__global__ __launch_bounds__(256) void k_dct_blocks(float *blocks, size_t nblocks, int inverse)
{
        __shared__ __attribute__((aligned(16))) float tp[4 * kTpWave];
        const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
        const size_t blk = ((size_t)blockIdx.x * 4 + wave) * 8 + (lane >> 3);
        const int rr = lane & 7;
        const bool ok = blk < nblocks;
        float v[8];
#pragma unroll
        for(int u = 0; u < 8; u++) { v[u] = ok ? blocks[blk * 64 + rr * 8 + u] : 0.f; }
        float *scratch = tp + wave * kTpWave;
        transpose8(v, scratch, lane);
        if(inverse) { idct8(v); } else { fdct8(v); }
        transpose8(v, scratch, lane);
        if(inverse) { idct8(v); } else { fdct8(v); }
        if(ok) {
#pragma unroll
                for(int u = 0; u < 8; u++) { blocks[blk * 64 + rr * 8 + u] = v[u]; }
        }
}

}  // namespace j2p
