// jpeg2png_amd — gfx950 device code for the deblocking solver's hot path.
//
// Two device-wide phases per iteration (the global ||g|| of compute.c:209-211
// forces the split):
//
//   k_gradient : FISTA point y = x_k + factor*(x_k - x_{k-1}) formed on the fly
//                (compute.c:433-439) by wavefronts that march down 128-column
//                strips with the row neighbourhood in registers and the column
//                neighbourhood through DPP lane shifts (no LDS), then the prob
//                (compute.c:53-66), TV (compute.c:73-125) and TGV2
//                (compute.c:128-197) subgradients in GATHER form, g written to
//                HBM plus one double sum(g*g) per strip and 16-row tile row.
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
//
// (The timing-only decomposition builds of rounds 3-5 — arithmetic / traffic / halo rows / mantissa test switched off one at
// a time — are gone from this file; their measurements are profiles/r03_decomposition.jsonl and r05_decomposition.jsonl.)
// -DJ2P_TRACE -DJ2P_TRACE_CLOCK: the trace record's middle stamp becomes the wavefront's life in CORE-clock ticks (s_memtime)
// next to its life on the constant 100 MHz clock: the shader clock the kernels actually run at (tools/core_clock.py)
// (and one that stays correct: J2P_PROJECT_MAXWAVES=N caps k_project's wavefronts per SIMD through its LDS footprint)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace j2p {

constexpr int kMaxCh = 3;
constexpr int kHalo = 2;

// ---------------------------------------------------------------------------
// -DJ2P_DEBUG: the analogue of the reference's DEBUG build, whose pixel indexer p() asserts every access
// (utils.h:68-81).  Every global load and store of the two phase kernels is checked against the byte range the
// access is MEANT to stay in — x: the band's own rows plus the halo rows that exist in the image (phase A) / the
// own rows (phase B); g: the own rows; prob state and d: the coefficient rows the band holds.  A violation does
// not trap (the kernels' loads are unconditional by design, so a trap would hide the count): it is counted, and
// the first offender's site code and offset are kept; j2p_debug_violations() reads the counters.  The release
// build compiles the checks out.
// ---------------------------------------------------------------------------
#ifdef J2P_DEBUG
struct DbgRange {
        const char *lo, *hi;
};
struct DbgChan {
        DbgRange x_read[2];    // [0] = xcur, [1] = xprev: rows phase A may read
        DbgRange x_own[2];     // ... rows phase B reads and writes
        DbgRange grad, pg, d;
        unsigned long long *counters;   // [0] violations, [1] first site code, [2] first offset (bytes from lo)
};
__device__ __forceinline__ void dbg_check(const DbgChan &g, const DbgRange &r, const void *p, unsigned bytes, unsigned site)
{
        const char *c = static_cast<const char *>(p);
        if(c < r.lo || c + bytes > r.hi) {
                if(atomicAdd(g.counters, 1ull) == 0ull) {
                        g.counters[1] = site;
                        g.counters[2] = (unsigned long long)(c - r.lo);
                }
        }
}
#define J2P_CHK(k, range, ptr, bytes, site) dbg_check((k).dbg, (k).dbg.range, (ptr), (bytes), (site))
#else
#define J2P_CHK(k, range, ptr, bytes, site) ((void)0)
#endif

// ---------------------------------------------------------------------------
// -DJ2P_TRACE (tools/wave_trace.py; timing tool, never shipped): every wavefront of the two phase kernels leaves
// one record {start, first data, end} in 10 ns ticks of the constant clock plus its hardware id (XCD, SE, CU, SIMD),
// so that a launch can be laid out wave by wave: when wavefronts start, how long the first loads take, how long
// a wavefront lives and how they are spread over the CUs.  Record 0 of the buffer is the append counter.
// ---------------------------------------------------------------------------
#ifdef J2P_TRACE
struct TraceRec {
        unsigned long long t_start, t_data, t_end, id;    // id = kernel tag << 56 | launch seq << 32 | xcc << 24 | hw_id bits
};
__device__ __forceinline__ unsigned long long trace_now() { return wall_clock64(); }
__device__ __forceinline__ unsigned trace_hwid()
{
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        return (hw & 0xffffffu) | ((xcc & 0xfu) << 24);
}
// slot = the launch's first record (reserved by the host) + the wavefront's index within the launch: no atomic —
// a counter shared by every wavefront of the chip would serialise their exits at ~12 ns each and stretch the launch
__device__ __forceinline__ void trace_put(unsigned long long *buf, unsigned cap, unsigned base, unsigned tag, unsigned seq,
                                          unsigned long long t0, unsigned long long t1, unsigned long long t2)
{
        if(!buf || (threadIdx.x & 63) != 0) { return; }
        const unsigned waves = (blockDim.x + 63) >> 6;
        const unsigned long long slot = (unsigned long long)base +
                ((unsigned long long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * waves + (threadIdx.x >> 6);
        if(slot + 1 >= cap) { return; }
        TraceRec *r = reinterpret_cast<TraceRec *>(buf) + 1 + slot;
        r->t_start = t0;
        r->t_data = t1;
        r->t_end = t2;
        r->id = ((unsigned long long)tag << 56) | ((unsigned long long)(seq & 0xffffffu) << 32) | trace_hwid();
}
#endif

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
        const uint8_t *d8;  // the same as d + 128 in one byte each when every |d| of the channel is <= 127 (k_narrow_coefficients), else NULL
        const float *q;     // 64 floats
        unsigned cw, ch;    // coefficient plane size (whole image)
        unsigned ws, hs;    // subsampling
        unsigned crow0;     // first coefficient row held in d / pg  (= row0 / hs clipped)
        unsigned crows;     // coefficient rows held (pg always has at least one row allocated)
        float p_alpha;      // pweight*2*255*sqrtf(2)  (compute.c:245)
        int prob_on;        // pweight != 0
#ifdef J2P_DEBUG
        DbgChan dbg;
#endif
};

struct Geo {
        unsigned W, H;      // canvas
        unsigned row0;      // first canvas row of the band
        unsigned rows;      // band rows
        unsigned ntx;       // gradient strips (wavefronts) per row of strips
        unsigned rpw;       // rows per gradient strip = rows per norm partial: 16, or 8 / 4 on small canvases
        // row segments of a gradient launch: segment i covers band-local rows [i * rpw, min((i + 1) * rpw, rows));
        // launch row j of the grid handles segment seg_off + j * seg_mul (all: 0,1; interior only: 1,1; the two
        // edge ones: 0, nseg - 1).  Plain arithmetic on kernel arguments: a table in memory would put two dependent
        // loads in front of every wavefront's first row fetch
        unsigned seg_off, seg_mul;
        // how a gradient launch's workgroups map to strips and rows (grad_item): units (four strips x a pair of tile rows)
        // and tile rows of the launch, and the shares (in 1/256) of every XCD's run that are dealt as double, as half and as
        // quarter tile rows
        unsigned units, ntr_launch, zone_d, zone_b, zone_c;
        // 1: the launch walks the canvas bottom-up (unit n - 1 - u instead of u).  k_project walks top-down, so each
        // phase then STARTS on the rows the phase before touched last — what the Infinity Cache still holds of planes
        // that do not fit it (j2p_solver_create: canvases whose two planes exceed the cache)
        unsigned reverse;
#ifdef J2P_TRACE
        unsigned long long *trace;   // TraceRec buffer (record 0 unused) or NULL
        unsigned trace_cap, trace_seq, trace_base;
#endif
};

// Row-tiled runs whose bands can write each other's memory (j2p_tiled, exchange "direct"): the level-1 sums of
// ||g||^2 go straight into EVERY band's copy of the global [tile row][channel] array (the
// band's own included), so that each band finishes the norm from memory of its own (norm_tree_wave in k_project, or
// k_norm_finish) without a reduction launch on a root band and the event hop behind it
constexpr int kMaxBands = 32;
struct RowsumPush {
        double *dst[kMaxBands];             // band b's global array of this iteration's parity
        unsigned n;                         // 0 = no push
        unsigned first_tr;                  // this band's first global tile row
        // J2P_TILED_WAIT=counter: when the band's LAST tile row has been pushed, one count goes to every band's counter
        // (coherent pinned host memory): a band's projection then waits for ONE value (hipStreamWaitValue64: N counts per
        // iteration) instead of N - 1 events
        unsigned long long *count[kMaxBands];
        unsigned ncount;                    // 0 = no counters
};

struct GradArgs {
        ChanDev ch[kMaxCh];
        Geo geo;
        float factor;       // FISTA (t-1)/tnext  (compute.c:432)
        float a_tv;         // 1/sqrt(nchannel)   (compute.c:90)
        float a_tgv;        // alpha/sqrt(nchannel) (compute.c:154)
        double *part_g2;    // [c][local tile row][tile col]
        double *part_tv;    // [local tile row][tile col][2]  (LOG only)
        // norm reduction folded into this kernel (see fold_tile_row): the last strip to deliver its partial of a
        // 16-row tile row sums that row's partials; the last tile row of the launch to finish runs the tree
        unsigned *row_ticket;   // [local tile rows], zero between launches; NULL = no fold (separate kernels reduce)
        unsigned *done_ticket;  // one counter, zero between launches
        double *rowsum;         // [local tile row][channel]: the level-1 sums (what the bands of a tiled run exchange)
        float *norm_out;        // [channel]; NULL = stop after level 1 (band solvers, canvases above kFoldMaxRows tile rows)
        unsigned nch_total;     // channels of the solver (partials per strip and tile row)
        unsigned fold_phase;    // 0 / 1: the iteration's parity, carried in the SIGN BIT of every partial of a folding launch (fold_tile_row)
        unsigned fold_rows;     // tile rows this launch completes
        unsigned ntr_global;    // tile rows of the whole canvas (length of the tree's input)
        // linked bands: where every tile row's sum also goes, in DEVICE memory (one wavefront per tile row reads it; as
        // 272 bytes of kernel arguments it cost every launch of every solver 0.5 us, same-box A/B); NULL = no push
        const RowsumPush *push;
};

struct ProjArgs {
        ChanDev ch[kMaxCh];
        Geo geo;
        float factor;
        float step;         // radius / sqrtf(1 + iterations)  (compute.c:443)
        const float *norm;  // [c] ||g||  (compute.c:210)
        double *part_prob;  // [c][strip]  (LOG only)
        unsigned strips_per_chan;   // stride of part_prob
        unsigned chan_of_z[kMaxCh]; // channel handled by blockIdx.z (channels are launched grouped by sampling)
        // block rows of the band handled by this launch, per blockIdx.z (they depend on the channel's vertical
        // sampling): by = by_offset + i * by_mul for i < nby
        // (all: 0,1,brows; first and last only: 0,brows-1,2; all but those: 1,1,brows-2)
        unsigned by_offset[kMaxCh], by_mul[kMaxCh], nby[kMaxCh];
        // non-NULL: the norm is not read from `norm` but reduced by every wavefront itself from the level-1 row sums
        // [tile row][channel] the gradient launch left behind (norm_tree_wave) — no reduction kernel between the phases
        const double *norm_rowsums;
        unsigned norm_rows, norm_nch;
        // Row-tiled runs whose bands can write each other's memory (NIP == 2 instantiations only): the band's first /
        // last kHalo rows of the new iterate ALSO go where the neighbouring bands' next gradient phase reads them — the
        // lower halo rows of the band above, the upper halo rows of the band below, in the buffer that becomes x_{k+1}
        // there too — instead of being copied or sent after the launch.  (Those rows hold the neighbour's halo of
        // x_{k-1} until now, which its gradient launch of THIS iteration read: no projection starts before every
        // band's gradient launch has finished, it needs ||g||.)  NULL = nobody to tell.
        float *halo_up[kMaxCh];      // row 0 of the lower halo rows of the band above
        float *halo_down[kMaxCh];    // row 0 of the upper halo rows of the band below
};

// rows per norm partial on canvases large enough to fill the chip: the granularity of the GPU-count invariant
// reduction, and what band boundaries are aligned to (J2P_TILE_ROWS).  Small canvases use 8 or 4 (Geo::rpw).
constexpr int kTY = 16;

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
// Phase A: gradient, register-marching form.
//
// One wavefront owns a strip of 128 consecutive columns (2 per lane, handled as
// packed float2 so that the arithmetic issues as v_pk_* ops) and walks down kRPW
// rows.  Everything the gather needs from neighbouring COLUMNS moves between
// lanes with DPP wave shifts; everything it needs from neighbouring ROWS is
// carried in registers from one loop trip to the next: no LDS, no barriers.
// The two outermost columns on each side of the strip are halo (loaded and
// differenced, never stored), so a strip yields 124 output columns; likewise
// each strip recomputes the source terms of one row above and below its rows.
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
// The loop is unrolled by hand three times over a ring of row slots so that the
// carried state never has to be copied.
//
// Division and square root.  Both are IEEE-correct.  The fast path is the compiler's own
// f32 sequence with the range scaling left out (division: rcp, one Newton step,
// quotient, two fma residual corrections, the refined reciprocal shared by the 3-4
// numerators of a pixel; square root: v_sqrt_f32 then the two fma residual tests
// against the neighbouring floats), written on float2 so that it issues as v_pk_fma.
// It is bit-identical to `/` and sqrtf() whenever the compiler's version would not
// have rescaled its operands, and that is guaranteed by screening the INPUT of the
// whole stencil once per loaded pixel: if every y is 0 or 2^-20 <= |y| < 2^41, all
// first/second differences are multiples of 2^-44, hence 0 or >= 2^-44, their
// squares are normal, the norms lie in [2^-43, 2^43] and no quotient is subnormal.
// A wavefront that loads a pixel outside that range (in practice: rounding noise around 0 in
// the chroma of a flat grey area, where the all-zero coefficient blocks decode to exactly 0)
// takes the plain `/` and sqrtf() path for the rows that pixel touches.
// ---------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Row accesses of the phase kernels as BUFFER instructions: a wave-uniform resource (the plane's base for this
// wavefront's rows, in scalar registers), a loop-invariant 32-bit lane offset and a scalar row offset —
// buffer_load_dword v, v_off, s[rsrc], s_row offen — so that no vector arithmetic goes into addresses.  As flat
// pointers the same accesses cost one 64-bit vector addition each (v_lshl_add_u64, chained row to row: 24 of them in
// front of k_project's 24 loads, 4.5 per row trip in k_gradient).  The resource covers "everything from the base on"
// (no range checking intended: the kernels clamp their own addresses; J2P_DEBUG checks the equivalent pointers);
// row offsets stay far below 4 GiB because the base is the wavefront's first row, not the plane's.
// NT: the non-temporal hint of nt_policy (the `nt` bit of the instruction).
// ---------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_from(const void *base)
{
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)0xfffffffe, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)row_off, NT ? 2 : 0));
}
template <bool NT>
__device__ __forceinline__ v2f buf_load2(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u raw = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)row_off, NT ? 2 : 0);
        return __builtin_bit_cast(v2f, raw);
}
template <bool NT>
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)lane_off, (int)row_off, NT ? 2 : 0);
}
template <bool NT>
__device__ __forceinline__ void buf_store(v2f v, __amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)lane_off, (int)row_off, NT ? 2 : 0);
}
template <bool NT>
__device__ __forceinline__ void buf_store4(float a, float b, float c, float d, __amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u raw = v4u{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                            __builtin_bit_cast(unsigned, d)};
        __builtin_amdgcn_raw_buffer_store_b128(raw, r, (int)lane_off, (int)row_off, NT ? 2 : 0);
}
template <bool NT, class V>
__device__ __forceinline__ V buf_load(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned row_off)
{
        if constexpr(sizeof(V) == 8) { return buf_load2<NT>(r, lane_off, row_off); }
        else { return buf_load1<NT>(r, lane_off, row_off); }
}

// Everything below is written once for a lane's PIXEL VECTOR V: v2f = two neighbouring columns per lane, the arithmetic
// issuing as packed operations (128-column strips: what canvases that fill the chip use), or float = one column per
// lane (64-column strips: twice the wavefronts with half the work each, for canvases whose launches would otherwise
// leave most wavefront slots empty — packed f32 operations cost two plain ones on gfx950, so nothing is lost per pixel
// but the halo columns).  The few places that care which of the two they are dealing with are these overloads.
template <class V>
__device__ __forceinline__ V splat(float s);
template <>
__device__ __forceinline__ float splat<float>(float s) { return s; }
template <>
__device__ __forceinline__ v2f splat<v2f>(float s) { return v2f{s, s}; }

// wave_shr:1 / wave_shl:1 with bound_ctrl: the lane without a source reads 0, and because no "old" value
// has to be supplied the compiler does not spend a v_mov on initialising the destination
__device__ __forceinline__ float lane_from_left(float v)    // value held by lane-1 (0 in lane 0)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_from_right(float v)   // value held by lane+1 (0 in lane 63)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// columns x-1 / x+1 of a lane's pixel vector
__device__ __forceinline__ v2f left_of(v2f a) { return v2f{lane_from_left(a.y), a.x}; }
__device__ __forceinline__ v2f right_of(v2f a) { return v2f{a.y, lane_from_right(a.x)}; }
__device__ __forceinline__ float left_of(float a) { return lane_from_left(a); }
__device__ __forceinline__ float right_of(float a) { return lane_from_right(a); }

// The lane shifts feed one addition or subtraction each.  Written per element, the shifted operand can go into the
// instruction itself (v_add_f32_dpp / v_subrev_f32_dpp) with one plain v_add_f32 as its partner
__device__ __forceinline__ v2f add_left_of(v2f g, v2f t) { return v2f{g.x + lane_from_left(t.y), g.y + t.x}; }     // g + left_of(t)
__device__ __forceinline__ v2f add_right_of(v2f g, v2f t) { return v2f{g.x + t.y, g.y + lane_from_right(t.x)}; }  // g + right_of(t)
__device__ __forceinline__ v2f minus_left_of(v2f a) { return v2f{a.x - lane_from_left(a.y), a.y - a.x}; }         // a - left_of(a)
__device__ __forceinline__ v2f right_of_minus(v2f a) { return v2f{a.y - a.x, lane_from_right(a.x) - a.y}; }       // right_of(a) - a
__device__ __forceinline__ float add_left_of(float g, float t) { return g + lane_from_left(t); }
__device__ __forceinline__ float add_right_of(float g, float t) { return g + lane_from_right(t); }
__device__ __forceinline__ float minus_left_of(float a) { return a - lane_from_left(a); }
__device__ __forceinline__ float right_of_minus(float a) { return lane_from_right(a) - a; }

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float pk_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// hardware approximations (1 ulp) and the IEEE forms, per element
__device__ __forceinline__ v2f hw_rcp(v2f d) { return v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}; }
__device__ __forceinline__ float hw_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ v2f hw_rsq(v2f x) { return v2f{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ v2f ieee_sqrt(v2f x) { return v2f{sqrtf(x.x), sqrtf(x.y)}; }
__device__ __forceinline__ float ieee_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ v2f ieee_div(v2f x, v2f d) { return v2f{x.x / d.x, x.y / d.y}; }
__device__ __forceinline__ float ieee_div(float x, float d) { return x / d; }
// a where n != 0, 0 where n == 0 (compute.c:97,158); n where n != 0, 1 where n == 0; max(n, m)
__device__ __forceinline__ v2f weight_unless_zero(v2f n, float a) { return v2f{n.x == 0.f ? 0.f : a, n.y == 0.f ? 0.f : a}; }
__device__ __forceinline__ float weight_unless_zero(float n, float a) { return n == 0.f ? 0.f : a; }
__device__ __forceinline__ v2f one_if_zero(v2f n) { return v2f{n.x == 0.f ? 1.f : n.x, n.y == 0.f ? 1.f : n.y}; }
__device__ __forceinline__ float one_if_zero(float n) { return n == 0.f ? 1.f : n; }
__device__ __forceinline__ v2f at_least(v2f n, float m) { return v2f{fmaxf(n.x, m), fmaxf(n.y, m)}; }
__device__ __forceinline__ float at_least(float n, float m) { return fmaxf(n, m); }
// acc += (double)(a * n) element by element, in column order (the log sums, compute.c:92,156)
__device__ __forceinline__ void add_scaled(double &acc, float a, v2f n)
{
        acc += (double)(a * n.x);
        acc += (double)(a * n.y);
}
__device__ __forceinline__ void add_scaled(double &acc, float a, float n) { acc += (double)(a * n); }
// acc += (double)v element by element (compute.c:203)
__device__ __forceinline__ void add_elements(double &acc, v2f v)
{
        acc += (double)v.x;
        acc += (double)v.y;
}
__device__ __forceinline__ void add_elements(double &acc, float v) { acc += (double)v; }
// The operand screen of the short division / sqrt sequences, on bit patterns and for a whole row at once (make_y in
// k_gradient): a loaded pixel is outside the range for which the short paths are exact when 0 < |y| < 2^-20, |y| >= 2^41
// or it is a NaN.  Flat regions whose pixels are rounding noise around 0 (1e-17 in the chroma of a grey area) do occur
// in image data, so the lower bound has to hold all the way down to the subnormals.  hi = largest |y|, lo = smallest
// NON-ZERO |y| minus one ulp (0 - 1 wraps to the top, so zeros drop out of the minimum)
__device__ __forceinline__ void screen_update(unsigned &hi, unsigned &lo, v2f y)
{
        // (bit-cast the VECTOR: hipcc 7.2 turns element-wise casts of .x and .y into two reads of .x)
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u u = __builtin_bit_cast(v2u, y) & 0x7fffffffu;
        const v2u um = u - 1u;
        hi = max(hi, max(u.x, u.y));
        lo = min(lo, min(um.x, um.y));
}
__device__ __forceinline__ void screen_update(unsigned &hi, unsigned &lo, float y)
{
        const unsigned u = __builtin_bit_cast(unsigned, y) & 0x7fffffffu;
        hi = max(hi, u);
        lo = min(lo, u - 1u);
}

// shared part of the division: reciprocal refined by one Newton step
template <class V>
__device__ __forceinline__ V div_prepare(V d)
{
        const V r = hw_rcp(d);
        const V e = pk_fma(-d, r, splat<V>(1.f));
        return pk_fma(e, r, r);
}
// x / d given r = div_prepare(d)
template <class V>
__device__ __forceinline__ V div_shared(V x, V d, V r)
{
        const V q0 = x * r;
        const V q1 = pk_fma(pk_fma(-d, q0, x), r, q0);
        return pk_fma(pk_fma(-d, q1, x), r, q1);
}
// ---- the SHORT division ----
// With r the CORRECTLY ROUNDED reciprocal of d, one residual correction is enough (Markstein): q0 = RN(x r),
// e = x - d q0 (exact, one fma), q = RN(q0 + e r) is RN(x / d).  Three operations per quotient instead of five.
// Not taken on trust: j2p_division_exhaustive enumerates EVERY denominator mantissa against EVERY numerator mantissa
// (2^46 quotients, GPU test) with r = 1.f / d; operand ranges as for div_shared (no intermediate can be subnormal).
template <class V>
__device__ __forceinline__ V recip_exact(V d, V seed)
{
        const V one = splat<V>(1.f);
        const V r1 = pk_fma(pk_fma(-d, seed, one), seed, seed);
        return pk_fma(pk_fma(-d, r1, one), r1, r1);
}
template <class V>
__device__ __forceinline__ V div_exact_recip(V x, V d, V r)
{
        const V q0 = x * r;
        return pk_fma(pk_fma(-d, q0, x), r, q0);
}

// the same for N numerators over one denominator, written breadth-first so that the N
// independent fma chains are interleaved instead of issued back to back
template <int N, class V>
__device__ __forceinline__ void div_shared_n(const V (&x)[N], V d, V r, V (&q)[N])
{
        V q0[N], e0[N], q1[N], e1[N];
#pragma unroll
        for(int i = 0; i < N; i++) { q0[i] = x[i] * r; }
#pragma unroll
        for(int i = 0; i < N; i++) { e0[i] = pk_fma(-d, q0[i], x[i]); }
#pragma unroll
        for(int i = 0; i < N; i++) { q1[i] = pk_fma(e0[i], r, q0[i]); }
#pragma unroll
        for(int i = 0; i < N; i++) { e1[i] = pk_fma(-d, q1[i], x[i]); }
#pragma unroll
        for(int i = 0; i < N; i++) { q[i] = pk_fma(e1[i], r, q1[i]); }
}

// sqrtf for 0 or 2^-96 <= x < 2^126: v_sqrt_f32 is within 1 ulp; pick the correctly rounded
// neighbour with two exact residuals (the compiler's own expansion without its rescaling)
__device__ __forceinline__ v2f sqrt_fast(v2f x)
{
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2f s = v2f{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
        // the neighbouring floats, through integer VECTOR arithmetic: written element-wise
        // (v2f{bitcast(int(s.x)-1), bitcast(int(s.y)-1)}) hipcc 7.2 reuses the .x result for .y
        const v2i si = __builtin_bit_cast(v2i, s);
        const v2f dn = __builtin_bit_cast(v2f, si - 1);
        const v2f up = __builtin_bit_cast(v2f, si + 1);
        const v2f vp = pk_fma(-dn, s, x), vs = pk_fma(-up, s, x);
        v2f r = v2f{vp.x <= 0.f ? dn.x : s.x, vp.y <= 0.f ? dn.y : s.y};
        r = v2f{vs.x > 0.f ? up.x : r.x, vs.y > 0.f ? up.y : r.y};
        return r;
}
__device__ __forceinline__ float sqrt_fast(float x)
{
        const float s = __builtin_amdgcn_sqrtf(x);
        const int si = __builtin_bit_cast(int, s);
        const float dn = __builtin_bit_cast(float, si - 1), up = __builtin_bit_cast(float, si + 1);
        const float vp = pk_fma(-dn, s, x), vs = pk_fma(-up, s, x);
        float r = vp <= 0.f ? dn : s;
        r = vs > 0.f ? up : r;
        return r;
}

constexpr int kStripCols = 124;   // output columns per wavefront strip with two columns per lane (one: 60)
#ifndef J2P_RING
#define J2P_RING 4
#endif
constexpr int kRing = J2P_RING;   // row slots = hand-unroll factor of the marching loop (1 channel; 3 otherwise: registers);
                                  // rows are fetched (slots - 1) trips ahead (3 slots: 12 % slower; 2 or 5 ring turns per
                                  // loop iteration: slower too)

// compile-time description of a strip for k_gradient's march: `value` = it touches no image / band / coverage
// edge (clamps and masks are the identity), `unit` = additionally every channel of the wavefront is sampled 1x1
// and covers all of the strip's columns (the prob state of a lane's pixel vector is one load at a row offset)
template <bool FREE, bool UNIT>
struct MarchTag {
        static constexpr bool value = FREE;
        static constexpr bool unit = UNIT;
};

template <int NCH, bool TGV, class V = v2f>
struct SourceTerms {
        V tvx[NCH], tvo[NCH], tvy[NCH];                           // TV: to (x+1) (shifted at its use), own, to the row below
        V A[NCH], O[NCH], B[NCH], C[NCH];                         // TGV2 (A and C are shifted left / right at their uses)
};

// sqrtf for 2^-100 <= x < 2^127 through the reciprocal square root: the compiler's own expansion
// for the flush-denormal mode (one v_rsq_f32, then a coupled Newton step on sqrt and 1/(2 sqrt)
// and a final residual correction).  Correctly rounded on that whole range — checked
// EXHAUSTIVELY against sqrtf() by j2p_sqrt_exhaustive (every float, GPU test).  Not valid for 0.
// `r` receives the v_rsq_f32 values the root was made of.
template <class V>
__device__ __forceinline__ V sqrt_rsq(V x, V &r)
{
        r = hw_rsq(x);
        V s = x * r;
        V h = r * 0.5f;
        const V e = pk_fma(-h, s, splat<V>(0.5f));
        h = pk_fma(h, e, h);
        s = pk_fma(s, e, s);
        const V d = pk_fma(-s, s, x);
        return pk_fma(d, h, s);
}
template <class V>
__device__ __forceinline__ V sqrt_rsq(V x)
{
        V r;
        return sqrt_rsq(x, r);
}

// EXACT_ZERO: result must be sqrtf(x) including x == 0 (the log sums read the norm itself).
// Otherwise only a positive divisor is needed when x == 0 (every numerator is 0 then): on the
// screened path x is 0 or >= 2^-88, so x + 2^-120 is x itself unless x == 0, where the root comes
// out as exactly 2^-60 (checked by j2p_math_selftest) — one add instead of clamping both
// the radicand and the root.
template <bool FAST, bool EXACT_ZERO, class V>
__device__ __forceinline__ V sqrt_pair(V x)
{
        if(!FAST) { return ieee_sqrt(x); }
        if(EXACT_ZERO) { return sqrt_fast(x); }
        return sqrt_rsq(x + splat<V>(0x1p-120f));
}
// divisor for the quotients of a pixel whose norm is n (see sqrt_pair)
template <bool FAST, bool EXACT_ZERO, class V>
__device__ __forceinline__ V divisor_of(V n)
{
        if(!FAST) { return one_if_zero(n); }                                       // divide by 1, scale by 0
        if(EXACT_ZERO) { return at_least(n, 0x1p-60f); }
        return n;                                                                  // already >= 2^-60
}
// what the source terms need of a sum of squares x: the norm n = sqrtf(x), the divisor d made from it and the
// reciprocal r of d that the quotients are refined from (unscreened path: r is not used).
// SHORT (the screened path without log sums): r is refined TWICE from the v_rsq_f32 value the root was made of, which
// makes it the correctly rounded 1 / n for every norm but those with an all-ones mantissa (k_recip_exhaustive: all
// norms enumerated), and each quotient then needs ONE residual correction (div_exact_recip; k_div_exhaustive: every
// radicand x every numerator enumerated, clean except at those norms).  The march sends rows that may hold such a
// norm down the unscreened path (allones_candidate).  Per pixel pair 4 + 7 x 3 packed operations instead of
// 2 + 2 transcendental + 7 x 5.
template <bool FAST, bool EXACT_ZERO, class V>
__device__ __forceinline__ void norm_and_reciprocal(V x, V &n, V &d, V &r)
{
        if constexpr(FAST && !EXACT_ZERO) {
                V seed;
                n = sqrt_rsq(x + splat<V>(0x1p-120f), seed);
                d = n;
                r = recip_exact(d, seed);
        } else {
                n = sqrt_pair<FAST, EXACT_ZERO>(x);
                d = divisor_of<FAST, EXACT_ZERO>(n);
                if constexpr(FAST) { r = div_prepare(d); }
                else { r = d; }
        }
}
template <bool FAST, bool EXACT_ZERO, int N, class V>
__device__ __forceinline__ void div_n(const V (&x)[N], V d, V r, V (&q)[N])
{
        if constexpr(FAST && !EXACT_ZERO) {
#pragma unroll
                for(int i = 0; i < N; i++) { q[i] = div_exact_recip(x[i], d, r); }
        } else if constexpr(FAST) {
                div_shared_n<N>(x, d, r, q);
        } else {
#pragma unroll
                for(int i = 0; i < N; i++) { q[i] = ieee_div(x[i], d); }
        }
}
// true when one of the radicands of a lane's pixels MAY give a norm with an all-ones mantissa: such a norm is the root
// of a radicand whose own mantissa ends in 0x7ffffe or 0x7fffff, so "the low 16 bits are >= 0xfffe" is necessary — one
// row in ~120 trips of a wavefront, which then takes the unscreened (IEEE) path for that row; a few 16-bit maxima and
// one compare per lane
__device__ __forceinline__ bool allones_candidate(v2f r1, v2f r2)
{
        typedef unsigned short v2h __attribute__((ext_vector_type(2)));
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u a = __builtin_bit_cast(v2u, r1), b = __builtin_bit_cast(v2u, r2);
        const v2h m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(v2h, a.x), __builtin_bit_cast(v2h, a.y)),
                                                __builtin_elementwise_max(__builtin_bit_cast(v2h, b.x), __builtin_bit_cast(v2h, b.y)));
        return m.x >= (unsigned short)0xfffe;
}
__device__ __forceinline__ bool allones_candidate(float r1, float r2)
{
        typedef unsigned short v2h __attribute__((ext_vector_type(2)));
        const v2h m = __builtin_elementwise_max(__builtin_bit_cast(v2h, r1), __builtin_bit_cast(v2h, r2));
        return m.x >= (unsigned short)0xfffe;
}

// The four TGV2 numerators of a pixel (compute.c:165-182): s + gxx, gyy + s, s (its sign goes onto the quotient)
// and the own term 2 gxx + 2 s + 2 gyy.  On the screened path every operand is a multiple of 2^-44 below 2^43, so
// doubling commutes with every rounding involved — (2 gxx + 2 s) + 2 gyy == 2 ((gxx + s) + gyy), and gxx + s is the
// first numerator (float addition commutes) — and the factor 2 moves through the division and onto the weight:
// a2 * -((2 m) / n) == (2 a2) * -(m / n), all of it exact scaling.  Three multiplications and one addition less
// per pixel; the unscreened path (subnormal operands possible) keeps the reference's expression.
template <bool FAST, class V>
__device__ __forceinline__ void tgv_numerators(V xx, V sy, V yy, V (&num)[4])
{
        num[0] = sy + xx;
        num[1] = yy + sy;
        num[2] = sy;
        if(FAST) { num[3] = num[0] + yy; }
        else { num[3] = 2.f * xx + 2.f * sy + 2.f * yy; }
}
template <bool FAST, class V>
__device__ __forceinline__ V own_term(V a2, V q3)
{
        if(FAST) { return (2.f * a2) * -q3; }       // 2 a2: exact and wave-uniform (hoisted out of the march)
        return a2 * -q3;
}

// Source terms of one image row for a lane's pixel vector, in two steps.  gx,gy: forward differences of this row,
// gxp,gyp: of the row above.  m_hx / m_hy zero the second differences on the first column / first row
// (compute.c:137-143).
//   source_prepare : the second differences and the two sums of squares under the norms (the same arithmetic on every
//                    path) — from which the march decides which path the row takes
//   source_finish  : norms, quotients, weights.  FAST: the screened path (short sequences, see norm_and_reciprocal);
//                    otherwise plain `/` and sqrtf().  tv / tv2 receive the log sums when `log_row`.
template <int NCH, bool TGV, class V = v2f>
struct SourcePrep {
        V n1r, n2r;                                // gx^2 + gy^2 summed over the channels; the TGV2 counterpart
        V xx[NCH], sy[NCH], yy[NCH];
};

template <int NCH, bool TGV, bool MASKED, class V>
__device__ __forceinline__ void source_prepare(const V (&gx)[NCH], const V (&gy)[NCH], const V (&gxp)[NCH],
                                               const V (&gyp)[NCH], V m_hx, V m_hy, SourcePrep<NCH, TGV, V> &p)
{
        // ---- TV (compute.c:84-89) ----
        // (the reference starts the sum at 0.f; a square is never -0, so 0.f + gx * gx is gx * gx bit for bit)
        V n1 = gx[0] * gx[0];
        n1 += gy[0] * gy[0];
#pragma unroll
        for(int c = 1; c < NCH; c++) {
                n1 += gx[c] * gx[c];
                n1 += gy[c] * gy[c];
        }
        p.n1r = n1;
        p.n2r = splat<V>(0.f);
        // ---- TGV2 (compute.c:136-152) ----
        if(TGV) {
                V n2 = splat<V>(0.f);
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        // MASKED == false: the caller knows every mask is 1 here (v * 1.f is v, so dropping
                        // the products changes no bit)
                        p.xx[c] = MASKED ? minus_left_of(gx[c]) * m_hx : minus_left_of(gx[c]);
                        const V gyx = MASKED ? minus_left_of(gy[c]) * m_hx : minus_left_of(gy[c]);
                        const V gxy = MASKED ? (gx[c] - gxp[c]) * m_hy : gx[c] - gxp[c];
                        p.yy[c] = MASKED ? (gy[c] - gyp[c]) * m_hy : gy[c] - gyp[c];
                        p.sy[c] = (gxy + gyx) * 0.5f;                     // (g_xy + g_yx) / 2.
                        const V term = p.xx[c] * p.xx[c] + 2.f * (p.sy[c] * p.sy[c]) + p.yy[c] * p.yy[c];
                        if(c == 0) { n2 = term; }                       // 0.f + term is term: never -0
                        else { n2 += term; }
                }
                p.n2r = n2;
        }
}

// source_finish on the screened path without log sums (the hot one), LEVEL BY LEVEL.  The arithmetic is that of
// norm_and_reciprocal<true, false> + div_exact_recip, operation for operation — sqrt_rsq and recip_exact for the two norms,
// then one residual correction per quotient — but written breadth-first: the two norm chains side by side, the 7 quotient
// chains of a channel side by side, with a scheduling barrier for vector instructions between the levels.  Why: on gfx950 a
// packed f32 operation needs a wait state before an instruction that reads its result, and left to itself the compiler
// emits each chain depth-first — 30 of the hot march's 250 instructions per row trip were `s_nop 0` between a v_pk_fma and
// the v_pk_fma that consumes it (tools/isa_count.py), each one an issue slot of the wavefront.  Same bits, fewer slots.
#ifndef J2P_LEVELS
#define J2P_LEVELS 1
#endif
#define J2P_LEVEL_END() __builtin_amdgcn_sched_barrier(0x0094)      /* SALU, VMEM and DS may cross; vector ALU work may not */
template <int NCH, bool TGV, class V>
__device__ __forceinline__ void source_finish_levels(const V (&gx)[NCH], const V (&gy)[NCH], const SourcePrep<NCH, TGV, V> &p, float a_tv,
                                                     float a_tgv, SourceTerms<NCH, TGV, V> &s)
{
        constexpr int K = TGV ? 2 : 1;                         // norms: TV, TGV2
        const V one = splat<V>(1.f), half = splat<V>(0.5f), tiny = splat<V>(0x1p-120f);
        V x[K], r[K], sq[K], h[K], e[K], n[K], rc[K];
        x[0] = p.n1r + tiny;
        if(TGV) { x[K - 1] = p.n2r + tiny; }
        // ---- sqrt_rsq ----
#pragma unroll
        for(int k = 0; k < K; k++) { r[k] = hw_rsq(x[k]); }
#pragma unroll
        for(int k = 0; k < K; k++) { sq[k] = x[k] * r[k]; }
#pragma unroll
        for(int k = 0; k < K; k++) { h[k] = r[k] * 0.5f; }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { e[k] = pk_fma(-h[k], sq[k], half); }
        // (numerators have nothing to do with the norms: they fill the slots between the levels of the two chains)
        V num[NCH][TGV ? 7 : 3];
        const V a1 = splat<V>(a_tv), a2 = splat<V>(a_tgv);
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                num[c][0] = a1 * gx[c];
                num[c][1] = a1 * gy[c];
                num[c][2] = a1 * -(gx[c] + gy[c]);
        }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { h[k] = pk_fma(h[k], e[k], h[k]); }
#pragma unroll
        for(int k = 0; k < K; k++) { sq[k] = pk_fma(sq[k], e[k], sq[k]); }
        if(TGV) {
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        V t4[4];
                        tgv_numerators<true>(p.xx[c], p.sy[c], p.yy[c], t4);
#pragma unroll
                        for(int i = 0; i < 4; i++) { num[c][(TGV ? 3 : 0) + i] = t4[i]; }
                }
        }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { e[k] = pk_fma(-sq[k], sq[k], x[k]); }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { n[k] = pk_fma(e[k], h[k], sq[k]); }          // the norms
        J2P_LEVEL_END();
        // ---- recip_exact(n, seed = r) ----
#pragma unroll
        for(int k = 0; k < K; k++) { e[k] = pk_fma(-n[k], r[k], one); }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { h[k] = pk_fma(e[k], r[k], r[k]); }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { e[k] = pk_fma(-n[k], h[k], one); }
        J2P_LEVEL_END();
#pragma unroll
        for(int k = 0; k < K; k++) { rc[k] = pk_fma(e[k], h[k], h[k]); }
        J2P_LEVEL_END();
        // ---- the quotients: q0 = RN(x r), q = RN(q0 + (x - d q0) r)  (div_exact_recip) ----
        constexpr int Q = TGV ? 7 : 3;
        V q0[NCH][Q], er[NCH][Q];
#pragma unroll
        for(int c = 0; c < NCH; c++) {
#pragma unroll
                for(int i = 0; i < Q; i++) { q0[c][i] = num[c][i] * rc[i < 3 ? 0 : K - 1]; }
        }
        J2P_LEVEL_END();
#pragma unroll
        for(int c = 0; c < NCH; c++) {
#pragma unroll
                for(int i = 0; i < Q; i++) { er[c][i] = pk_fma(-n[i < 3 ? 0 : K - 1], q0[c][i], num[c][i]); }
        }
        J2P_LEVEL_END();
#pragma unroll
        for(int c = 0; c < NCH; c++) {
#pragma unroll
                for(int i = 0; i < Q; i++) { q0[c][i] = pk_fma(er[c][i], rc[i < 3 ? 0 : K - 1], q0[c][i]); }
        }
        J2P_LEVEL_END();
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                s.tvx[c] = q0[c][0];
                s.tvy[c] = q0[c][1];
                s.tvo[c] = q0[c][2];
                if(TGV) {
                        s.A[c] = a2 * q0[c][Q - 4];                                     // to (x-1,y), (x+1,y)
                        s.B[c] = a2 * q0[c][Q - 3];                                     // to (x,y-1), (x,y+1)
                        s.C[c] = a2 * -q0[c][Q - 2];                                    // to (x+1,y-1), (x-1,y+1)
                        s.O[c] = own_term<true>(a2, q0[c][Q - 1]);                      // own
                }
        }
}

template <int NCH, bool TGV, bool LOG, bool FAST, class V>
__device__ __forceinline__ void source_finish(const V (&gx)[NCH], const V (&gy)[NCH], const SourcePrep<NCH, TGV, V> &p, float a_tv,
                                              float a_tgv, bool log_row, double &tv, double &tv2, SourceTerms<NCH, TGV, V> &s)
{
#if J2P_LEVELS
        if constexpr(FAST && !LOG && NCH == 1) {
                (void)log_row; (void)tv; (void)tv2;
                source_finish_levels<NCH, TGV, V>(gx, gy, p, a_tv, a_tgv, s);
                return;
        }
#endif
        // ---- TV (compute.c:90-104) ----
        V n1, d1, r1;
        norm_and_reciprocal<FAST, LOG>(p.n1r, n1, d1, r1);
        if(LOG && log_row) { add_scaled(tv, a_tv, n1); }
        // A pixel with zero norm contributes nothing (compute.c:97).  Screened path: n == 0 implies
        // every numerator is exactly 0 (no square can underflow), so any positive divisor gives 0.
        // Unscreened path: divide by 1, scale by 0.
        const V a1 = FAST ? splat<V>(a_tv) : weight_unless_zero(n1, a_tv);
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                if(NCH > 1) { __builtin_amdgcn_sched_barrier(0); }   // one channel at a time: bounds the live ranges
                const V num[3] = {a1 * gx[c], a1 * gy[c], a1 * -(gx[c] + gy[c])};
                V q[3];
                div_n<FAST, LOG, 3>(num, d1, r1, q);
                s.tvx[c] = q[0];
                s.tvy[c] = q[1];
                s.tvo[c] = q[2];
        }
        // ---- TGV2 (compute.c:153-183) ----
        if(TGV) {
                V n2, d2, r2;
                norm_and_reciprocal<FAST, LOG>(p.n2r, n2, d2, r2);
                if(LOG && log_row) { add_scaled(tv2, a_tgv, n2); }
                const V a2 = FAST ? splat<V>(a_tgv) : weight_unless_zero(n2, a_tgv);   // compute.c:158
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        if(NCH > 1) { __builtin_amdgcn_sched_barrier(0); }
                        // a2 * (expr / n2): division first (compute.c:165-182)
                        // the two negative numerators are divided as positives and the sign goes onto the quotient:
                        // (-v) / n == -(v / n) bit for bit, in the IEEE and in the short sequences alike
                        V num[4], q[4];
                        tgv_numerators<FAST>(p.xx[c], p.sy[c], p.yy[c], num);
                        div_n<FAST, LOG, 4>(num, d2, r2, q);
                        s.A[c] = a2 * q[0];                                             // to (x-1,y), (x+1,y)
                        s.B[c] = a2 * q[1];                                             // to (x,y-1), (x,y+1)
                        s.C[c] = a2 * -q[2];                                            // to (x+1,y-1), (x-1,y+1)
                        s.O[c] = own_term<FAST>(a2, q[3]);                              // own
                }
        }
}

// source_prepare for ONE channel of a jointly optimised image whose other channels live in the other
// wavefronts of the workgroup (J wavefronts = J channels, same strip).  Only the two norms couple
// the channels (compute.c:84-89, 148-152): every wavefront publishes the squares of its own
// differences, and after one barrier every wavefront adds them up in the reference's order
// ((((0 + gx0^2) + gy0^2) + gx1^2) + ... ; per-channel Hessian terms likewise), so all of them
// hold bit-identical sums — and take the same path — and the rest of the work (source_finish<1, ...>)
// stays private to the channel.  `xchg` is a double-buffered LDS area: [2][J][64 lanes][3] pixel vectors.
template <int J, bool TGV, bool MASKED, class V>
__device__ __forceinline__ void source_prepare_joint(int cidx, int lane, int parity, V *xchg, V gx, V gy, V gxp, V gyp,
                                                     V m_hx, V m_hy, SourcePrep<1, TGV, V> &p)
{
        V xx = splat<V>(0.f), sy = xx, yy = xx, tq = xx;
        if(TGV) {
                xx = MASKED ? minus_left_of(gx) * m_hx : minus_left_of(gx);
                const V gyx = MASKED ? minus_left_of(gy) * m_hx : minus_left_of(gy);
                const V gxy = MASKED ? (gx - gxp) * m_hy : gx - gxp;
                yy = MASKED ? (gy - gyp) * m_hy : gy - gyp;
                sy = (gxy + gyx) * 0.5f;
                tq = xx * xx + 2.f * (sy * sy) + yy * yy;
        }
        p.xx[0] = xx;
        p.sy[0] = sy;
        p.yy[0] = yy;
        V *mine = xchg + ((size_t)(parity * J + cidx) * 64 + lane) * 3;
        mine[0] = gx * gx;
        mine[1] = gy * gy;
        mine[2] = tq;
        __syncthreads();
        // (sums start with channel 0's terms: 0.f + a square is the square, see source_prepare)
        const V *o0 = xchg + ((size_t)(parity * J) * 64 + lane) * 3;
        V n1 = o0[0], n2 = o0[2];
        n1 += o0[1];
#pragma unroll
        for(int c = 1; c < J; c++) {
                const V *o = xchg + ((size_t)(parity * J + c) * 64 + lane) * 3;
                n1 += o[0];
                n1 += o[1];
                n2 += o[2];
        }
        p.n1r = n1;
        p.n2r = n2;
}

// ---------------------------------------------------------------------------
// Norm reduction folded into k_gradient (no separate launch, nothing serial between the two phases).
// Same arithmetic as strip_sum / tree_sum_lds below — a fixed function of the partial array, whoever
// evaluates it — so the norm is bit-identical to the stand-alone kernels' and independent of the order in
// which wavefronts arrive.  Cross-workgroup hand-over: see fold_arrive.
// ---------------------------------------------------------------------------
constexpr unsigned kFoldMaxRows = 1024;      // tile rows the in-kernel tree handles (canvas height <= 16384)

// lanes 8c..8c+7 of the calling wavefront: the eight interleaved running sums of strip_sum for channel c
__device__ __forceinline__ void fold_tile_row(const GradArgs &a, unsigned tr, size_t nparts, int lane)
{
        const unsigned ntx = a.geo.ntx, nch = a.nch_total;
        const int c = lane >> 3, j = lane & 7;
        double s = 0.;
        if(c < (int)nch) {
                const double *p = a.part_g2 + (size_t)c * nparts + (size_t)tr * ntx;
                // element i belongs to running sum i % 8, added in increasing i (strip_sum's order); loads batched
                for(unsigned i0 = (unsigned)j; i0 < ntx; i0 += 64) {
                        double v[8];
                        // The strips drew their tickets WITHOUT waiting for their partials to be acknowledged (that wait
                        // held every wavefront of the launch until its last row of g was written: 3 us of k_gradient at
                        // 4096^2).  A partial that has not landed yet shows: every partial of this launch carries the
                        // iteration's parity in its sign bit (a sum of squares has none of its own), and what the slot
                        // holds until then — the previous launch's partial, complete since that kernel ended, or the
                        // fill pattern of reset — carries the other parity.  Read again until the parities are right.
                        bool landed;
                        do {
                                landed = true;
#pragma unroll
                                for(int u = 0; u < 8; u++) {
                                        const unsigned i = i0 + 8u * u;
                                        v[u] = __hip_atomic_load(p + (i < ntx ? i : i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        landed = landed && (unsigned)(__builtin_bit_cast(unsigned long long, v[u]) >> 63) == a.fold_phase;
                                }
                        } while(!landed);
#pragma unroll
                        for(int u = 0; u < 8; u++) {
                                if(i0 + 8u * u < ntx) { s += __builtin_fabs(v[u]); }
                        }
                }
        }
        // ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)), evaluated by the lane with j == 0
        double t[8];
#pragma unroll
        for(int u = 0; u < 8; u++) { t[u] = __shfl(s, (lane & ~7) + u, 64); }
        const double sum = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        if(j == 0 && c < (int)nch) {
                __hip_atomic_store(a.rowsum + (size_t)tr * nch + c, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // row-tiled, bands in each other's reach: the sum into every band's copy of the global array (all eight lanes
        // of channel c's group hold it; lane j serves bands j, j + 8, ...).  System-scope stores: performed at the
        // destination, visible to the peers' projection launches through the event recorded behind this launch.
        if(a.push && c < (int)nch) {
                const RowsumPush &push = *a.push;
                const size_t slot = (size_t)(push.first_tr + tr) * nch + (unsigned)c;
                for(unsigned b = (unsigned)j; b < push.n; b += 8) {
                        __hip_atomic_store(push.dst[b] + slot, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
        }
}

// one wavefront: norm[c] = sqrtf((float) tree(rowsums of channel c))  (compute.c:200-207); buf >= P doubles of LDS
__device__ __forceinline__ void fold_tree(const GradArgs &a, double *buf, int lane)
{
        const unsigned n = a.ntr_global, nch = a.nch_total;
        unsigned P = 1;
        while(P < n) { P <<= 1; }
        for(unsigned c = 0; c < nch; c++) {
                for(unsigned i = (unsigned)lane; i < P; i += 64) {
                        buf[i] = i < n ? __hip_atomic_load(a.rowsum + (size_t)i * nch + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.;
                }
                for(unsigned st = P >> 1; st > 0; st >>= 1) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        for(unsigned i = (unsigned)lane; i < st; i += 64) { buf[i] = buf[i] + buf[i + st]; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if(lane == 0) { a.norm_out[c] = sqrtf((float)buf[0]); }
                __builtin_amdgcn_wave_barrier();
        }
}

// Hand-over between wavefronts on different CUs / XCDs WITHOUT cache maintenance.  A release fence at agent scope
// costs a write-back of the XCD's whole L2 (buffer_wbl2) per wavefront — measured: k_gradient 64 -> 379 us at
// 4096^2 with one fence per strip.  Instead every value that crosses (partials, row sums, tickets) is written and
// read with agent-scope atomic accesses (sc1: performed at the device-coherent level, not in the XCD's L2), the
// producer waits for its stores to be acknowledged (s_waitcnt) before it draws its ticket, and the consumer's
// loads depend on the ticket's value — so no other cached data has to move.
__device__ __forceinline__ void publish_double(double *p, double v)
{
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stores_acknowledged()
{
        // vmcnt(0) lgkmcnt(0): every store of this wavefront has been acknowledged.  Inline asm with a memory clobber,
        // not __builtin_amdgcn_s_waitcnt: the builtin is no barrier for the COMPILER (it may sink a publishing store
        // below it or hoist the ticket above it), and the pass that drops "redundant" waitcnts does not see inside asm
        // (MI355X_MICROARCH.md, compiler hazard).  ISA assumption: an sc1 store that has been acknowledged is visible
        // to every later sc1 load of the device.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// called by a wavefront whose lane 0 has just published `mine` partials of tile row tr (wave-uniform arguments)
__device__ __forceinline__ void fold_arrive(const GradArgs &a, unsigned tr, unsigned mine, size_t nparts, double *buf, int lane)
{
        unsigned old = 0;
        // (no wait for the partial to be acknowledged: the reader checks for itself, see fold_tile_row)
        if(lane == 0) { old = __hip_atomic_fetch_add(a.row_ticket + tr, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if(old + mine != a.geo.ntx * a.nch_total) { return; }
        // last strip of this tile row: every other strip has ISSUED its partials
        fold_tile_row(a, tr, nparts, lane);
        if(lane == 0) { __hip_atomic_store(a.row_ticket + tr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // ready for the next launch
        const unsigned ncount = a.push ? a.push->ncount : 0u;         // (wave-uniform)
        if(!a.norm_out && !ncount) { return; }
        unsigned done = 0;
        stores_acknowledged();                                       // the row sums of lanes 0, 8, 16 (and their pushed copies) before the ticket
        if(lane == 0) { done = __hip_atomic_fetch_add(a.done_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        done = (unsigned)__builtin_amdgcn_readfirstlane((int)done);
        if(done + 1 != a.fold_rows) { return; }
        if(a.norm_out) { fold_tree(a, buf, lane); }
        // the band's last tile row: every row sum of this launch has been acknowledged at its destinations (each finisher
        // waited for its stores before it drew its ticket), and no strip of the band reads a halo row any more — tell
        // every band (lane b: band b's counter)
#ifndef J2P_EXP_DROP_COUNTS   // (fault injection, tests/test_tiled_verify_gpu.py: the counts never come — the value form HANGS, and the verification's deadline has to catch it)
        if((unsigned)lane < ncount) { __hip_atomic_fetch_add(a.push->count[lane], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif
        if(lane == 0) { __hip_atomic_store(a.done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

#ifndef J2P_GRAD_WAVES
#define J2P_GRAD_WAVES 4
#endif
constexpr int kGradWaves1 = J2P_GRAD_WAVES;    // waves per SIMD the 1-channel gradient kernel is register-limited to (5 needs <= 96 VGPRs: spills)
// ... except the hot instantiation — one channel per workgroup wavefront, no logging (Y-only planes, the components of `-s`):
// with a ring of three row slots it fits 96 registers without a spill, i.e. five wavefronts per SIMD (round 5; the same
// ring at four wavefronts and either ring in the joint / logging kernels do not: 20-56 bytes of scratch)
#ifndef J2P_HOT_WAVES
#define J2P_HOT_WAVES 4
#endif
#ifndef J2P_HOT_RING
#define J2P_HOT_RING 4
#endif
constexpr int kHotWaves = J2P_HOT_WAVES, kHotRing = J2P_HOT_RING;
// ... separately for the planes whose working set exceeds the Infinity Cache (NT >= 1: from ~15 Mpixel; no instantiation
// more).  Measured with 5 / 3 there (profiles/r05_ab_five_wavefronts.jsonl): 4096^2 -0.2 %, 8192x4096 and 16384x2048
// -1.0 %, 8192^2 +1.2 % — a percent either way again; stays 4 / 4
#ifndef J2P_BIG_WAVES
#define J2P_BIG_WAVES 4
#endif
#ifndef J2P_BIG_RING
#define J2P_BIG_RING 4
#endif
constexpr int kBigWaves = J2P_BIG_WAVES, kBigRing = J2P_BIG_RING;
constexpr int kGradWaves3 = 2;    // ... the three-channels-in-one-wavefront schedule
// NCH channels are handled inside one wavefront (J == 1, workgroup = 4 strips), or — for a
// jointly optimised image — J wavefronts of a workgroup take one channel each of the same strip
// and only exchange their norm contributions through LDS (J > 1, NCH == 1).
// NT (0..3): which streams bypass the caches' retention (non-temporal loads / stores), chosen by the solver from the
// size of its working set against the 256 MiB Infinity Cache (nt_policy in j2p_solver_create).  x_k and x_{k-1} are
// each touched two or three times per iteration and never get the hint; g (written by this kernel, read once by
// k_project) gets it at level >= 1, the prob state (written by k_project, read once here) at >= 2, the coefficients
// d (read once per iteration by k_project) at 3.  What is left without the hint is what should stay cache-resident:
// 4096^2 Y (288 MiB): level 1, 137 -> 127 us per iteration; 16384x2048 (576 MiB, the planes x_k, x_{k-1} are exactly
// 256 MiB): level 3, 292 -> 240 us; when everything fits the hint costs 1-2 %.
// PX: columns per lane (2: packed arithmetic, 128-column strips; 1: 64-column strips, see the pixel-vector overloads above)
// What one wavefront of a gradient launch works on (wave-uniform; grad_item): strip `wcol`, band-local target rows
// [t0, t0 + nrows) of tile row `tr` — the whole tile row (kind 0), one of its halves (kind 1) or quarters (kind 2; `sub`
// says which), or tile rows tr AND tr + 1 (kind 3).  Half and quarter items exist so that the LAST workgroups of a launch
// are short: a launch is as long as its last wavefront, and a whole tile row is a wavefront life of ~17 us at 4096^2
// (profiles/r06_wave_trace.jsonl).  Double items are for the workgroups dispatched FIRST: 34 row trips for 32 rows
// instead of 2 x 18 — the two source rows above a strip are recomputed half as often, and read half as often.
constexpr int kKindWhole = 0, kKindHalf = 1, kKindQuarter = 2, kKindDouble = 3;
struct StripItem {
        int wcol, t0, nrows, tile0;
        unsigned tr;
        int kind, sub;
        bool active;            // false: beyond the last strip / the band's last row — nothing to march, but the wavefront
                                // still takes part in its workgroup's hand-over
};

// The march of one wavefront over its item's rows: FISTA point, gradient, g stored; the sums of g^2 come back per lane in
// the tile row's canonical order — lo = a0 + a1, hi = a2 + a3 with a_i the running sum over the tile row's i-th group of
// FOUR rows (an item that covers only part of the tile row leaves the others 0) — so that the partial of a tile row,
// (a0 + a1) + (a2 + a3) summed over the lanes, has the same bits whether one, two or four wavefronts marched it — or one
// wavefront marched it together with the tile row below (second index of the sums: which of the item's tile rows).
// xchg = the workgroup's LDS (joint images).
template <int NCH, int J>
constexpr int kItemTiles = NCH == 1 && J == 1 ? 2 : 1;       // tile rows an item can cover (double items: one channel per wavefront)
template <int NCH, bool TGV, bool LOG, int J, int NT, int PX, class V>
__device__ __forceinline__ void march_rows(const GradArgs &a, V *xchg, const StripItem &it, double (&g2_lo)[NCH][kItemTiles<NCH, J>],
                                           double (&g2_hi)[NCH][kItemTiles<NCH, J>], double &tv_acc, double &tv2_acc, unsigned long long &tr_data)
{
        constexpr int kCols = 64 * PX - 4;                      // output columns per strip: 2 halo columns on each side
        const int lane = (int)threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // uniform: keeps row/strip arithmetic scalar
        const int wcol = it.wcol;
        const int cbase = J == 1 ? 0 : wave;                    // first channel of this wavefront
        const int W = (int)a.geo.W, H = (int)a.geo.H;
        const int rows = (int)a.geo.rows, row0 = (int)a.geo.row0;
        const int t0 = it.t0;                                   // band-local target rows [t0, t1)
        const int t1 = t0 + it.nrows < rows ? t0 + it.nrows : rows;
        const int tile0 = it.tile0;
        // (the three scalars by value: read through `a` inside the lambdas they ended up in an LDS-promoted alloca)
        const float fista_factor = a.factor, w_tv = a.a_tv, w_tgv = a.a_tgv;
        // Strip i loads columns [kCols i, kCols i + 64 PX); its two outermost columns on each side are halo — except
        // at the image's left and right edges, where the neighbour beyond the edge contributes nothing anyway, so the
        // first strip also owns its left halo lanes and the last strip its right ones: n strips cover kCols n + 4
        // columns (two columns per lane: 124 n + 4, 33 strips for W = 4096).
        const int xl = wcol * kCols + lane * PX;               // canvas column of the lane's first pixel (W is a multiple of 8)

        const bool pair_in = xl >= 0 && xl < W;                // the lane's columns are in the image (both, or neither)
        const bool pair_own = pair_in && (lane * PX >= 2 || wcol == 0) && (lane * PX + PX <= 64 * PX - 2 || wcol * kCols + 64 * PX >= W);
        // per-lane constant masks (1.f / 0.f), multiplied instead of selected: v*1 is exact, v*0 = +-0
        const float in_f = pair_in ? 1.f : 0.f;
        V m_gx, m_hx;
        if constexpr(PX == 2) {
                m_gx = v2f{in_f, xl + 1 >= W - 1 ? 0.f : in_f};     // gx = 0 on the last column (compute.c:79)
                m_hx = v2f{xl == 0 ? 0.f : in_f, in_f};             // gxx, gyx = 0 on the first column
        } else {
                m_gx = xl >= W - 1 ? 0.f : in_f;
                m_hx = xl == 0 ? 0.f : in_f;
        }

        // Rows are fetched kRing-1 loop trips before they are needed: `fetch_row` only issues the
        // loads of x_k / x_{k-1} (raw values stay in the ring), `make_y` turns them into the FISTA
        // point (compute.c:433-439) when the row is first used.  All loads are UNCONDITIONAL —
        // lanes left/right of the image and rows above/below it read a clamped, valid address and
        // are zeroed by a mask afterwards — so the loop body is straight-line code and the compiler
        // can keep the younger loads in flight (counted s_waitcnt) instead of draining them.
        const int xl_c = xl < 0 ? 0 : (xl > W - PX ? W - PX : xl);
        const unsigned xoff = (unsigned)xl_c * 4u;             // byte offset of the lane's pixel vector within a row
        // (rows t0 - 2 ... t1 + 1 are what the segment touches; the x buffers have 2 halo rows above the band's row 0)
        const int seg_base = t0 - 2, grad_base = t0;
        __amdgpu_buffer_rsrc_t res_cur[NCH], res_prev[NCH], res_grad[NCH], res_pg[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                const ChanDev &k = a.ch[cbase + c];
                res_cur[c] = rows_from(k.xcur + (ptrdiff_t)seg_base * W);
                res_prev[c] = rows_from(k.xprev + (ptrdiff_t)seg_base * W);
                res_grad[c] = rows_from(k.grad + (ptrdiff_t)grad_base * W);
                // (the prob state of a unit-sampled channel, see load_p: coefficient row = canvas row)
                res_pg[c] = rows_from(k.pg + (size_t)((unsigned)(row0 + grad_base) - k.crow0) * k.cw);
        }
        const int lr_lo = -(row0 < (int)kHalo ? row0 : (int)kHalo);                      // first readable band-local row
        const int lr_hi = rows - 1 + (H - row0 - rows < (int)kHalo ? H - row0 - rows : (int)kHalo);
        // FREE (a std::bool_constant, see `march` below): the strip is known to lie inside the image and the
        // band with room to spare, so the row clamps and every 0/1 mask are the identity and are left out
        auto fetch_row = [&](auto free_tag, int lr, V (&rc)[NCH], V (&rp)[NCH]) {
                constexpr bool FREE = decltype(free_tag)::value;
                // rows past the strip's last needed row (t1+1) re-read that row: a cache hit, not HBM traffic
                int lm = lr > t1 + 1 ? t1 + 1 : lr;
                int lc = FREE ? lm : (lm < lr_lo ? lr_lo : (lm > lr_hi ? lr_hi : lm));
                // wave-uniform resource at the segment's first row + loop-invariant 32-bit lane offset + scalar row offset:
                // no vector arithmetic in the address (see rows_from)
                const ptrdiff_t roff = (ptrdiff_t)lc * W;              // (the pointer form: what J2P_DEBUG checks)
                (void)roff;
                const unsigned row_off = (unsigned)(lc - seg_base) * (unsigned)W * 4u;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        J2P_CHK(a.ch[cbase + c], x_read[0], reinterpret_cast<const char *>(a.ch[cbase + c].xcur + roff) + xoff, 4 * PX, 101);
                        J2P_CHK(a.ch[cbase + c], x_read[1], reinterpret_cast<const char *>(a.ch[cbase + c].xprev + roff) + xoff, 4 * PX, 102);
                        rc[c] = buf_load<false, V>(res_cur[c], xoff, row_off);
                        rp[c] = buf_load<false, V>(res_prev[c], xoff, row_off);
                }
        };
        auto make_y = [&](auto free_tag, int lr, const V (&rc)[NCH], const V (&rp)[NCH], V (&y)[NCH], unsigned &suspect) {
                constexpr bool FREE = decltype(free_tag)::value;
                const int gr = row0 + lr;
                const float m = gr >= 0 && gr < H ? in_f : 0.f;   // 0 outside the image
                // the operand screen of the short division / sqrt sequences (see screen_update) for the whole row at
                // once, on the bit patterns: hi = largest |y|, lo = smallest NON-ZERO |y| minus one ulp (0 - 1
                // wraps to the top, so zeros drop out of the minimum).  Two compares per row, each feeding a
                // ballot directly, so the flag is born in scalar registers.
                unsigned hi = 0u, lo = ~0u;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const V yy = rc[c] + fista_factor * (rc[c] - rp[c]);     // compute.c:435
                        y[c] = FREE ? yy : yy * m;
                        screen_update(hi, lo, y[c]);
                }
                constexpr unsigned kLo = 0x35800000u, kHi = 0x54000000u;      // bits of 2^-20 and 2^41
                const unsigned long long out_of_range = __builtin_amdgcn_ballot_w64(hi >= kHi) | __builtin_amdgcn_ballot_w64(lo < kLo - 1u);
                suspect = (unsigned)out_of_range | (unsigned)(out_of_range >> 32);   // wave-uniform, non-zero = suspect
        };
        // forward differences of row gr given rows gr and gr+1 (compute.c:79,81)
        auto diffs = [&](auto free_tag, int gr, const V (&yc)[NCH], const V (&yn)[NCH], V (&gx)[NCH], V (&gy)[NCH]) {
                constexpr bool FREE = decltype(free_tag)::value;
                const float m_gy = gr >= 0 && gr < H - 1 ? 1.f : 0.f;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        gx[c] = FREE ? right_of_minus(yc[c]) : right_of_minus(yc[c]) * m_gx;
                        gy[c] = FREE ? yn[c] - yc[c] : (yn[c] - yc[c]) * m_gy;
                }
        };

        // prob-gradient state of one target row (compute.c:53-66: replicated over the sample's
        // footprint).  Loaded unconditionally from a clamped address; `pmask` (per lane) and the row
        // test at the point of use decide whether it contributes.  A channel with pweight == 0 has an
        // all-zero state buffer, so it needs no special case.
        V p_scale[NCH];
        int p_col[NCH][PX];
#pragma unroll
        for(int c = 0; c < NCH; c++) {
                const ChanDev &k = a.ch[cbase + c];
                const bool on = k.prob_on && pair_own && (unsigned)xl < k.cw * k.ws;
                p_scale[c] = splat<V>(on ? k.p_alpha : 0.f);
                const unsigned cmax = k.cw - 1;
#pragma unroll
                for(int e = 0; e < PX; e++) {
                        const unsigned ce = (unsigned)(xl_c + e) / k.ws;
                        p_col[c][e] = (int)(ce > cmax ? cmax : ce) * 4;   // byte offsets within a coefficient row
                }
        }
        auto load_p = [&](auto free_tag, int lt, V (&pv)[NCH]) {
                constexpr bool FREE = decltype(free_tag)::value;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const ChanDev &k = a.ch[cbase + c];
                        // coefficient row of canvas row lt, clamped into the rows this band holds
                        const int ltc = lt > t1 - 1 ? t1 - 1 : lt;                     // past the strip: re-read its last row
                        const int gt = row0 + (FREE ? ltc : (ltc < 0 ? 0 : ltc));
                        if constexpr(decltype(free_tag)::unit) {
                                const float *prow = k.pg + (size_t)((unsigned)gt - k.crow0) * k.cw;
                                J2P_CHK(k, pg, reinterpret_cast<const char *>(prow) + xoff, 4 * PX, 103);
                                (void)prow;
                                pv[c] = buf_load<(NT >= 2), V>(res_pg[c], xoff, (unsigned)(gt - row0 - grad_base) * k.cw * 4u);
                                continue;
                        }
                        unsigned cr;
                        if(k.hs == 1) { cr = (unsigned)gt; }                            // (uniform branch: skips the scalar division)
                        else { cr = (unsigned)gt / k.hs; }
                        if(!FREE) {
                                const unsigned cr_hi = k.crow0 + (k.crows ? k.crows - 1 : 0);
                                cr = cr < k.crow0 ? k.crow0 : (cr > cr_hi ? cr_hi : cr);
                        }
                        const float *prow = k.pg + (size_t)(cr - k.crow0) * k.cw;
                        J2P_CHK(k, pg, reinterpret_cast<const char *>(prow) + (unsigned)p_col[c][0], 4, 104);
                        if constexpr(PX == 2) {
                                J2P_CHK(k, pg, reinterpret_cast<const char *>(prow) + (unsigned)p_col[c][1], 4, 105);
                                pv[c] = v2f{*reinterpret_cast<const float *>(reinterpret_cast<const char *>(prow) + (unsigned)p_col[c][0]),
                                            *reinterpret_cast<const float *>(reinterpret_cast<const char *>(prow) + (unsigned)p_col[c][1])};   // two dword loads whatever the sampling: no branch
                        } else {
                                pv[c] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(prow) + (unsigned)p_col[c][0]);
                        }
                }
        };

        double g2[NCH];                                  // sum of g*g over the running group of four rows
#pragma unroll
        for(int c = 0; c < NCH; c++) { g2[c] = 0.; }
        // a group of four rows is complete (target row t was its last): into the tile row's lower or upper pair sum
        const int tile_rows = (int)a.geo.rpw;
        auto close_group = [&](int t) {
                int rel = t - tile0;                             // (everything here is wave-uniform)
                const bool second = kItemTiles<NCH, J> == 2 && rel >= tile_rows;
                if(second) { rel -= tile_rows; }
                const bool upper = (rel & 8) != 0;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        if constexpr(kItemTiles<NCH, J> == 2) {
                                if(second) {
                                        if(upper) { g2_hi[c][1] += g2[c]; }
                                        else { g2_lo[c][1] += g2[c]; }
                                } else {
                                        if(upper) { g2_hi[c][0] += g2[c]; }
                                        else { g2_lo[c][0] += g2[c]; }
                                }
                        } else {
                                if(upper) { g2_hi[c][0] += g2[c]; }
                                else { g2_lo[c][0] += g2[c]; }
                        }
                        g2[c] = 0.;
                }
        };
        constexpr int R = NCH == 1 ? (J == 1 && !LOG && PX == 2 ? (NT >= 1 ? kBigRing : kHotRing) : kRing) : 3;

        // The march over the strip's rows, compiled twice: once general, once for strips that touch neither an
        // image edge, a band edge nor a channel's coverage limit (all but the outermost strips and segments).
        // In the second form the clamps, compares and multiplications by 1.f disappear — about a seventh of the
        // instructions of a trip, most of them scalar — and nothing else changes, so the bits are the same.
        auto march = [&](auto free_tag) {
                constexpr bool FREE = decltype(free_tag)::value;
                // rings of kRing row slots, slot = (row - (t0-1)) mod kRing = phase of the trip that owns the row
                V RC[R][NCH], RP[R][NCH], Y[R][NCH], GX[R][NCH], GY[R][NCH], PV[R][NCH];
                unsigned bad1, bad2;   // screen results (non-zero = suspect) of the last two rows made; plain scalars
                SourceTerms<NCH, TGV, V> S[R];
                {
                        // rows t0-2, t0-1, t0 are needed at once; rows up to t0+R-2 are put in flight
                        V mc[NCH], mp[NCH], ym[NCH];
                        unsigned bm;
                        fetch_row(free_tag, t0 - 2, mc, mp);
                        fetch_row(free_tag, t0 - 1, RC[0], RP[0]);
        #pragma unroll
                        for(int i = 1; i <= R - 1; i++) { fetch_row(free_tag, t0 - 1 + i, RC[i], RP[i]); }
        #pragma unroll
                        for(int i = 1; i <= R - 3; i++) { load_p(free_tag, t0 - 1 + i, PV[i]); }
                        unsigned b0;
                        make_y(free_tag, t0 - 2, mc, mp, ym, bm);
                        make_y(free_tag, t0 - 1, RC[0], RP[0], Y[0], b0);
#ifdef J2P_TRACE
                        // (the screen flag depends on the loaded rows: the stamp cannot be taken before they arrived)
                        if((bm | b0) != 0xffffffffu) { tr_data = trace_now(); }
#endif
                        bad2 = bm;
                        bad1 = b0;
                        diffs(free_tag, row0 + t0 - 2, ym, Y[0], GX[R - 1], GY[R - 1]);
                }
                // one trip: source terms of row r into slot P, then target row r-1
                auto trip = [&](auto phase, int r) {
                        constexpr int P = decltype(phase)::value, P1 = (P + 1) % R, PM1 = (P + R - 1) % R, PM2 = (P + R - 2) % R;
                        const int gr = row0 + r;
                        // put row r+R in flight (its slot held row r, whose raw values became y last trip), and the
                        // prob state of target row r+R-2; then finish row r+1, fetched R-1 trips ago
                        fetch_row(free_tag, r + R, RC[P], RP[P]);
                        load_p(free_tag, r + R - 2, PV[PM2]);
                        unsigned bnew;
                        make_y(free_tag, r + 1, RC[P1], RP[P1], Y[P1], bnew);
                        const unsigned badmask = bad2 | bad1 | bnew;        // rows r-1, r, r+1
                        bad2 = bad1;
                        bad1 = bnew;
                        SourceTerms<NCH, TGV, V> &s = S[P];
                        diffs(free_tag, gr, Y[P], Y[P1], GX[P], GY[P]);
                        {
                                // A row above or below the image needs no special case: its y is 0, m_gy zeroes
                                // its gy, and hy = 0 zeroes its gxy/gyy, so every term comes out 0.
                                const bool log_row = LOG && pair_own && r >= t0 && r < t1 && cbase == 0;
                                const float hy = gr <= 0 || gr >= H ? 0.f : in_f;  // gxy, gyy = 0 on the first row (compute.c:141-143)
                                const V m_hy = splat<V>(hy);
                                // second differences and the sums under the two norms (the same on every path), then the
                                // path: screened unless the rows involved hold a value outside the screen's range or one
                                // of the norms may have an all-ones mantissa (see norm_and_reciprocal)
                                SourcePrep<NCH, TGV, V> prep;
                                if constexpr(J > 1) {
                                        const int parity = (r - t0 + 1) & 1;
                                        source_prepare_joint<J, TGV, !FREE>(cbase, lane, parity, xchg, GX[P][0], GY[P][0], GX[PM1][0], GY[PM1][0],
                                                                            m_hx, m_hy, prep);
                                } else {
                                        source_prepare<NCH, TGV, !FREE>(GX[P], GY[P], GX[PM1], GY[PM1], m_hx, m_hy, prep);
                                }
                                bool slow = badmask != 0;
                                if constexpr(!LOG) { slow = slow || __builtin_amdgcn_ballot_w64(allones_candidate(prep.n1r, prep.n2r)) != 0; }
                                if(!slow) { source_finish<NCH, TGV, LOG, true>(GX[P], GY[P], prep, w_tv, w_tgv, log_row, tv_acc, tv2_acc, s); }
                                else { source_finish<NCH, TGV, LOG, false>(GX[P], GY[P], prep, w_tv, w_tgv, log_row, tv_acc, tv2_acc, s); }
                        }
                        // ---- target row t = r-1: rows t-1, t, t+1 live in slots PM2, PM1, P ----
                        const int t = r - 1;
                        if(t >= t0) {
                                const SourceTerms<NCH, TGV, V> &up = S[PM2], &mid = S[PM1];
                                const int gt = row0 + t;
        #pragma unroll
                                for(int c = 0; c < NCH; c++) {
                                        const ChanDev &k = a.ch[cbase + c];
                                        V g = splat<V>(0.f);
                                        if(FREE || (unsigned)gt < k.ch * k.hs) { g += p_scale[c] * PV[PM1][c]; }   // row t, fetched R-1 trips ago
                                        g += up.tvy[c];                  // TV from (x, t-1)
                                        g = add_left_of(g, mid.tvx[c]);  // TV from (x-1, t)
                                        g += mid.tvo[c];                 // TV own
                                        if(TGV) {
                                                g += up.B[c];                    // (x,   t-1)
                                                g = add_right_of(g, up.C[c]);    // (x+1, t-1)
                                                g = add_left_of(g, mid.A[c]);    // (x-1, t)
                                                g += mid.O[c];                   // own
                                                g = add_right_of(g, mid.A[c]);   // (x+1, t)
                                                g = add_left_of(g, s.C[c]);      // (x-1, t+1)
                                                g += s.B[c];                     // (x,   t+1)
                                        }
                                        if(pair_own) {
                                                J2P_CHK(k, grad, reinterpret_cast<char *>(k.grad + (size_t)t * W) + (unsigned)xl * 4u, 4 * PX, 106);
                                                buf_store<(NT >= 1)>(g, res_grad[c], (unsigned)xl * 4u, (unsigned)(t - grad_base) * (unsigned)W * 4u);
                                                add_elements(g2[c], g * g);      // compute.c:203
                                        }
                                }
                                // t0 is a multiple of 4 and the trip of phase 1 handles t = t0 - 1 + 4 i: with a ring of four
                                // the place where groups end is known at compile time
                                if constexpr(R == 4) {
                                        if constexpr(P == 1) { close_group(t); }
                                } else {
                                        if(((t - t0) & 3) == 3) { close_group(t); }
                                }
                        }
                };

                // one turn of the ring = R trips; returns false once the last row (t1) has been done
                auto ring = [&](int r) -> bool {
                        trip(std::integral_constant<int, 0>{}, r);
                        if(r + 1 > t1) { return false; }
                        trip(std::integral_constant<int, 1>{}, r + 1);
                        if(r + 2 > t1) { return false; }
                        trip(std::integral_constant<int, 2>{}, r + 2);
                        if(R > 3) {
                                if(r + 3 > t1) { return false; }
                                trip(std::integral_constant<int, 3 % R>{}, r + 3);
                        }
                        if(R > 4) {
                                if(r + 4 > t1) { return false; }
                                trip(std::integral_constant<int, 4 % R>{}, r + 4);
                        }
                        return r + R <= t1;
                };
                for(int r = t0 - 1; r <= t1; r += R) {
                        if(!ring(r)) { break; }
                }
        };
        {
                bool seg_free = wcol > 0 && wcol * kCols + 64 * PX <= W - 1 &&           // no lane on the first / last column
                                row0 + t0 - 2 >= 0 && row0 + t1 + 1 < H &&                // rows t0-2 .. t1+1 inside the image
                                t0 - 2 >= lr_lo && t1 + 1 <= lr_hi;                       // ... and readable in this band
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const ChanDev &k = a.ch[cbase + c];
                        seg_free = seg_free && (unsigned)(row0 + t1) <= k.ch * k.hs;      // every target row is covered by the channel
                }
                bool unit = seg_free;
#pragma unroll
                for(int c = 0; c < NCH; c++) {
                        const ChanDev &k = a.ch[cbase + c];
                        unit = unit && k.ws == 1 && k.hs == 1 && (unsigned)(wcol * kCols + 64 * PX) <= k.cw;
                }
                if(__builtin_amdgcn_readfirstlane(unit ? 1 : 0)) { march(MarchTag<true, true>{}); }
                else if(__builtin_amdgcn_readfirstlane(seg_free ? 1 : 0)) { march(MarchTag<true, false>{}); }
                else { march(MarchTag<false, false>{}); }
        }
        // (rows per strip that are no multiple of four — J2P_RPW experiments only — leave a group open)
        if(((t1 - t0) & 3) != 0) { close_group(t1 - 1); }
        (void)tr_data;          // (J2P_TRACE builds: stamped when the first rows have arrived)
}

// One wavefront of a gradient launch: march the item's rows, then turn the per-lane sums of g^2 into the partial of
// (tile row, strip) — one partial per strip and tile row (16 rows; 8 or 4 on small canvases), the granularity of the
// GPU-count invariant norm reduction — and report in (fold_arrive); the last strip of a tile row / of the launch to do so
// finishes the reduction.  The wavefronts of a half / quarter item hand their sums to the workgroup's wavefront that
// marched the tile row's first rows, through LDS (fold_buf).
template <int NCH, bool TGV, bool LOG, int J, int NT, int PX, class V>
__device__ __forceinline__ void gradient_strip(const GradArgs &a, V *xchg, double *fold_buf, const StripItem &it)
{
        static_assert(NCH == 1 || J == 1, "one channel per wavefront when a workgroup holds several");
        const int lane = (int)threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        const int cbase = J == 1 ? 0 : wave;
#ifdef J2P_TRACE
        const unsigned long long tr_start = trace_now();
#ifdef J2P_TRACE_CLOCK
        const unsigned long long tr_core = clock64();
#endif
#endif
        unsigned long long tr_data = 0;
        constexpr int kTiles = kItemTiles<NCH, J>;
        double lo[NCH][kTiles], hi[NCH][kTiles], tv_acc = 0., tv2_acc = 0.;
#pragma unroll
        for(int c = 0; c < NCH; c++) {
#pragma unroll
                for(int k = 0; k < kTiles; k++) { lo[c][k] = hi[c][k] = 0.; }
        }
        if(it.active) { march_rows<NCH, TGV, LOG, J, NT, PX, V>(a, xchg, it, lo, hi, tv_acc, tv2_acc, tr_data); }
        if(LOG) {
#pragma unroll
                for(int off = 32; off > 0; off >>= 1) {
                        tv_acc += __shfl_down(tv_acc, off, 64);
                        tv2_acc += __shfl_down(tv2_acc, off, 64);
                }
        }
        bool publisher = it.active;
        if constexpr(NCH == 1 && J == 1) {
                if(it.kind == kKindHalf || it.kind == kKindQuarter) {
                        // half items: wavefronts (0, 1) and (2, 3) of the workgroup share a strip; quarter items: all four do
                        const int first = it.kind == kKindHalf ? (wave & ~1) : 0;
                        if(wave != first) {
                                fold_buf[wave * 64 + lane] = lo[0][0] + hi[0][0];        // (one of the two is 0)
                                if(LOG && lane == 0) { fold_buf[256 + 2 * wave] = tv_acc; fold_buf[257 + 2 * wave] = tv2_acc; }
                        }
                        __syncthreads();
                        publisher = wave == first && it.active;
                        if(publisher) {
                                if(it.kind == kKindHalf) {
                                        hi[0][0] = fold_buf[(wave + 1) * 64 + lane];
                                        if(LOG) { tv_acc += fold_buf[256 + 2 * (wave + 1)]; tv2_acc += fold_buf[257 + 2 * (wave + 1)]; }
                                } else {
                                        lo[0][0] = lo[0][0] + fold_buf[64 + lane];
                                        hi[0][0] = fold_buf[128 + lane] + fold_buf[192 + lane];
                                        if(LOG) {
#pragma unroll
                                                for(int w = 1; w < 4; w++) { tv_acc += fold_buf[256 + 2 * w]; tv2_acc += fold_buf[257 + 2 * w]; }
                                        }
                                }
                        }
                }
        }
        if(publisher) {
                const size_t ntiles_row = a.geo.ntx;
                const size_t nparts = (size_t)((a.geo.rows + a.geo.rpw - 1) / a.geo.rpw) * ntiles_row;
                // (a double item: its second tile row too, if the band has it)
                const int ntile = kTiles == 2 && it.kind == kKindDouble && it.tile0 + (int)a.geo.rpw < (int)a.geo.rows ? 2 : 1;
                for(int k = 0; k < ntile; k++) {
                        const unsigned tr = it.tr + (unsigned)k;
#pragma unroll
                        for(int c = 0; c < NCH; c++) {
                                double v = kTiles == 2 && k == 1 ? lo[c][kTiles - 1] + hi[c][kTiles - 1] : lo[c][0] + hi[c][0];
#pragma unroll
                                for(int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); }
                                // (a folding launch marks its partials with the iteration's parity, see fold_tile_row: the sign bit is
                                // SET to it, whatever it was — a sum of squares is >= +0, and a NaN must not make the reader wait for ever)
                                if(a.row_ticket) {
                                        const unsigned long long bits = (__builtin_bit_cast(unsigned long long, v) & ~(1ull << 63)) | ((unsigned long long)a.fold_phase << 63);
                                        v = __builtin_bit_cast(double, bits);
                                }
                                if(lane == 0) { publish_double(&a.part_g2[(cbase + c) * nparts + (size_t)tr * ntiles_row + it.wcol], v); }
                        }
                        if(LOG && lane == 0 && cbase == 0) {
                                // (the CSV sums of a double item sit with its first tile row; the second one's slot holds 0)
                                const size_t w = (size_t)tr * ntiles_row + it.wcol;
                                a.part_tv[2 * w] = k == 0 ? tv_acc : 0.;
                                a.part_tv[2 * w + 1] = k == 0 ? tv2_acc : 0.;
                        }
                        if(a.row_ticket) { fold_arrive(a, tr, (unsigned)NCH, nparts, fold_buf, lane); }
                }
        }
#ifdef J2P_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wavefront's stores have been acknowledged
#ifdef J2P_TRACE_CLOCK
        tr_data = clock64() - tr_core;                            // core-clock ticks of the wavefront's life (tools/core_clock.py)
#endif
        trace_put(a.geo.trace, a.geo.trace_cap, a.geo.trace_base, 1u, a.geo.trace_seq, tr_start, tr_data, trace_now());
#endif
}

// item l of n, dealt to 8 queues in contiguous runs (queue q holds items [chunk_base(n, q), chunk_base(n, q + 1))): what
// "workgroup b runs on XCD b % 8" turns into when every XCD is to work on one contiguous region of the canvas
__host__ __device__ __forceinline__ unsigned chunk_base(unsigned n, unsigned q) { return q * (n >> 3) + (q < (n & 7) ? q : (n & 7)); }

// The order in which a gradient launch hands out its work.  Tile rows are taken in PAIRS (2 r, 2 r + 1) and the launch's
// (pair row, strip) positions are numbered row-major; a UNIT is four consecutive ones (J == 1: a workgroup's four
// wavefronts never idle because a row of strips is no multiple of four — 33 strips at W = 4096 — and every workgroup
// gives each SIMD of its CU one wavefront) or — joint images, a wavefront per channel — one.  Workgroup b runs on XCD
// b % 8 and the workgroups of an XCD start in the order of their numbers, so every XCD gets a contiguous, row-major run
// of units — vertically adjacent strips then meet in one L2 and their shared halo rows are fetched from HBM once — and
// deals the run in four zones, long items first: a share zone_d / 256 of the units as DOUBLES (one workgroup per unit: 4
// strips x both tile rows, 34 row trips for 32 rows), then whole tile rows (two workgroups per unit: 4 strips x 16 rows),
// then zone_b / 256 as HALVES (four workgroups: 2 strips x 2 halves of 8 rows), the last zone_c / 256 as QUARTERS (eight
// workgroups: 1 strip x 4 quarters of 4 rows).  The launch ends when its last wavefront does, and the wavefronts
// dispatched last are then the short ones.  Zones change who marches which rows, never a bit of the result (march_rows).
struct ZoneShares {
        unsigned d, b, c;       // doubles, halves, quarters in 1/256 of a run; the rest whole
};
__host__ __device__ __forceinline__ unsigned zone_workgroups(unsigned run, ZoneShares z)
{
        const unsigned quarters = (run * z.c) >> 8, halves = (run * z.b) >> 8, doubles = (run * z.d) >> 8;
        return doubles + 2 * (run - doubles - halves - quarters) + 4 * halves + 8 * quarters;
}
// workgroups a gradient launch needs: 8 x the longest run's
__host__ __device__ __forceinline__ unsigned grad_grid(unsigned n /* units */, ZoneShares z)
{
        unsigned most = 0;
        for(unsigned q = 0; q < 8; q++) {
                const unsigned w = zone_workgroups(chunk_base(n, q + 1) - chunk_base(n, q), z);
                most = w > most ? w : most;
        }
        return 8 * most;
}

// workgroup -> item of the calling wavefront (everything wave-uniform); false: the workgroup has nothing to do
// (host too: j2p_debug_grad_items enumerates a launch's items for the CPU test of this map, tests/test_capi.py)
template <int J>
__host__ __device__ __forceinline__ bool grad_item(const Geo &g, unsigned b, int wave, StripItem &it)
{
        const unsigned q = b & 7, j = b >> 3;
        const unsigned n = g.units;
        const unsigned first = chunk_base(n, q);
        // (plain scalars, no struct for the split: as an object it ended up in LDS, 12 bytes per thread, via the alloca promotion)
        const unsigned run = chunk_base(n, q + 1) - first;
        const unsigned quarters = (run * g.zone_c) >> 8, halves = (run * g.zone_b) >> 8, doubles = (run * g.zone_d) >> 8;
        const unsigned whole = run - doubles - halves - quarters;
        unsigned u = 0, strip_in_group = (unsigned)wave, tile_in_pair = 0;
        int kind = kKindWhole, sub = 0;
        if(j < doubles) {
                u = first + j;
                kind = kKindDouble;
        } else if(j < doubles + 2 * whole) {
                const unsigned jj = j - doubles;
                u = first + doubles + (jj >> 1);
                tile_in_pair = jj & 1;
        } else if(j < doubles + 2 * whole + 4 * halves) {
                const unsigned jj = j - doubles - 2 * whole;
                u = first + doubles + whole + (jj >> 2);
                kind = kKindHalf;
                tile_in_pair = (jj >> 1) & 1;
                strip_in_group = 2 * (jj & 1) + ((unsigned)wave >> 1);
                sub = wave & 1;
        } else if(j < doubles + 2 * whole + 4 * halves + 8 * quarters) {
                const unsigned jj = j - doubles - 2 * whole - 4 * halves;
                u = first + doubles + whole + halves + (jj >> 3);
                kind = kKindQuarter;
                tile_in_pair = (jj >> 2) & 1;
                strip_in_group = jj & 3;
                sub = wave;
        } else {
                return false;
        }
        if(g.reverse) { u = n - 1 - u; }
        const unsigned id = J == 1 ? 4 * u + strip_in_group : u;        // (pair row, strip) of the launch, row-major
        const unsigned pair_row = id / g.ntx;
        const unsigned tr_launch = 2 * pair_row + tile_in_pair;
        it.tr = g.seg_off + tr_launch * g.seg_mul;
        it.wcol = (int)(id - pair_row * g.ntx);
        it.kind = kind;
        it.sub = sub;
        it.tile0 = (int)(it.tr * g.rpw);
        it.nrows = kind == kKindDouble ? (int)(2 * g.rpw) : kind == kKindHalf ? (int)(g.rpw >> 1) : kind == kKindQuarter ? (int)(g.rpw >> 2) : (int)g.rpw;
        it.t0 = it.tile0 + sub * it.nrows;
        it.active = tr_launch < g.ntr_launch && it.t0 < (int)g.rows;
        return true;
}

// (the checked build carries its range descriptors in registers: at four wavefronts per SIMD it spilled 132-164 bytes to
// scratch, and kernels that need scratch on several streams waiting for each other's events — a row-tiled run with its
// bands on one GPU — hung the queue every other run, tests/test_debug_build_gpu.py; two wavefronts per SIMD, no scratch)
#ifdef J2P_DEBUG
constexpr int kDebugWaveCap = 2;
#else
constexpr int kDebugWaveCap = 64;
#endif
constexpr int grad_waves(int want) { return want < kDebugWaveCap ? want : kDebugWaveCap; }
template <int NCH, bool TGV, bool LOG, int J = 1, int NT = 0, int PX = 2>
__global__ __launch_bounds__((J == 1 ? 256 : 64 * J), grad_waves(PX == 1 ? (J == 1 && !LOG ? 6 : 2) : NCH == 1 ? (J == 1 && !LOG ? (NT >= 1 ? kBigWaves : kHotWaves) : kGradWaves1) : NCH == 2 ? 3 : kGradWaves3))
void k_gradient(GradArgs a)
{
        static_assert(J == 1 || NCH == 1, "channel-per-wavefront mode keeps one channel per wavefront");
        static_assert(PX == 2 || NCH == 1, "one column per lane: one channel per wavefront");
        typedef typename std::conditional<PX == 2, v2f, float>::type V;
        __shared__ __attribute__((aligned(16))) V xchg[J == 1 ? 1 : 2 * J * 64 * 3];
        __shared__ double fold_buf[kFoldMaxRows];               // the norm tree of the launch's last wavefront; sub-item hand-over
        StripItem it;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        if(!grad_item<J>(a.geo, blockIdx.x, wave, it)) { return; }
        gradient_strip<NCH, TGV, LOG, J, NT, PX, V>(a, xchg, fold_buf, it);
}

// ---------------------------------------------------------------------------
// Norm reduction.  Level 1: per row-of-tiles sums (sequential over tile columns).
// Level 2: pairwise tree over the row-of-tiles array padded to a power of two —
// a function of the GLOBAL array only, so the value does not depend on how many
// GPUs produced the rows.
// ---------------------------------------------------------------------------
// level 1: the strips of one row of tiles.  Eight interleaved running sums (so that the
// loads pipeline), combined pairwise — a fixed function of the array, whoever evaluates it.
__device__ __forceinline__ double strip_sum(const double *p, unsigned n)
{
        double s[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
        unsigned t = 0;
        for(; t + 8 <= n; t += 8) {
#pragma unroll
                for(int j = 0; j < 8; j++) { s[j] += p[t + j]; }
        }
        for(unsigned j = 0; t + j < n; j++) { s[j] += p[t + j]; }
        return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// Both level-1 kernels first stage the partials into LDS with coalesced loads (every load of the launch
// in flight at once) and then let one thread per tile row run strip_sum over its LDS copy: the per-thread
// strided global reads of the direct form cost one dependent L2/fabric round trip per 8 strips.
constexpr unsigned kStageDoubles = 2048;          // 16 KB of static LDS per k_rowsums block: many small blocks, one round each
constexpr unsigned kNormLdsBytes = 156 * 1024;    // dynamic LDS k_norm_whole may ask for

// copy n doubles global -> LDS with all 256 threads; loads are issued in batches of 40 per thread before
// the first store so that the whole copy is one or two memory round trips, not one per element
__device__ __forceinline__ void stage_copy(double *dst, const double *src, unsigned n)
{
        constexpr unsigned kBatch = 40;
        for(unsigned base = threadIdx.x; base < n; base += 256 * kBatch) {
                double v[kBatch];
#pragma unroll
                for(unsigned j = 0; j < kBatch; j++) {
                        const unsigned i = base + j * 256;
                        v[j] = src[i < n ? i : n - 1];
                }
#pragma unroll
                for(unsigned j = 0; j < kBatch; j++) {
                        const unsigned i = base + j * 256;
                        if(i < n) { dst[i] = v[j]; }
                }
        }
}

__global__ __launch_bounds__(256) void k_rowsums(const double *part, double *rowsum, unsigned ntx, unsigned nrows_local, unsigned nch,
                                                 unsigned items_per_block)
{
        // part: [c][local tile row][strip] -> rowsum: [local tile row][c]  (tile-row major, so that
        // concatenating the bands of consecutive GPUs yields the global array).  Item i = c * nrows_local + r;
        // a block owns items_per_block consecutive items (items_per_block * ntx <= kStageDoubles).
        __shared__ double st[kStageDoubles];
        const unsigned total = nrows_local * nch;
        const unsigned i0 = blockIdx.x * items_per_block;
        const unsigned items = i0 + items_per_block <= total ? items_per_block : total - i0;
        const double *src = part + (size_t)i0 * ntx;
        stage_copy(st, src, items * ntx);
        __syncthreads();
        for(unsigned k = threadIdx.x; k < items; k += 256) {
                const unsigned i = i0 + k, c = i / nrows_local, r = i % nrows_local;
                rowsum[(size_t)r * nch + c] = strip_sum(st + (size_t)k * ntx, ntx);
        }
}

constexpr int kMaxTileRows = 4096;   // canvas height <= 65536 (JPEG limit) / kTY; shorter tile rows only on small canvases

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

// whole-canvas solver: both levels in one launch (one block per channel), same arithmetic.
// Dynamic LDS: P doubles for the tree followed by stage_doubles for staging (kNormLdsBytes together at most;
// gfx950 lets one workgroup have all 160 KiB of a CU).
__global__ __launch_bounds__(256) void k_norm_whole(const double *part, unsigned ntx, unsigned nrows, unsigned nch, float *norm,
                                                    unsigned stage_doubles)
{
        extern __shared__ __attribute__((aligned(16))) float smem[];
        double *buf = reinterpret_cast<double *>(smem);
        const unsigned c = blockIdx.x;
        unsigned P = 1;
        while(P < nrows) { P <<= 1; }
        double *st = buf + P;
        const double *src = part + (size_t)c * nrows * ntx;
        if(stage_doubles == 0) {
                // few strips per tile row: one thread per tile row straight from global memory (the strided
                // reads cost one round trip per 8 strips, cheaper than funnelling everything through one CU's LDS)
                for(unsigned r = threadIdx.x; r < nrows; r += 256) { buf[r] = strip_sum(src + (size_t)r * ntx, ntx); }
        } else {
                const unsigned group = stage_doubles / ntx;         // tile rows staged per round (ntx <= 529 < stage_doubles)
                for(unsigned r0 = 0; r0 < nrows; r0 += group) {
                        const unsigned rows = r0 + group <= nrows ? group : nrows - r0;
                        __syncthreads();                            // the previous round's readers are done
                        stage_copy(st, src + (size_t)r0 * ntx, rows * ntx);
                        __syncthreads();
                        for(unsigned r = threadIdx.x; r < rows; r += 256) { buf[r0 + r] = strip_sum(st + (size_t)r * ntx, ntx); }
                }
        }
        for(unsigned r = nrows + threadIdx.x; r < P; r += 256) { buf[r] = 0.; }
        const double s = tree_sum_lds(buf, nrows, P);
        if(threadIdx.x == 0) { norm[c] = sqrtf((float)s); }
        (void)nch;
}

// Level 2 evaluated by ONE wavefront without LDS or barriers: the same padded pairwise tree (buf[i] + buf[i + s] for
// s = P/2 ... 1, P = the power of two >= n) — lane l holds elements l, l + 64, l + 128, ...; the levels with s >= 64
// add registers of one lane, the levels below move partner values between lanes.  Identical additions, hence the
// identical double, as tree_sum_lds.  n <= 1024.  (Padding P up to 64 only adds exact zeros to sums that are >= 0.)
constexpr unsigned kWaveTreeMax = 1024;
// in two steps, so that the caller can put its own loads in flight between them: the row sums are requested first,
// the planes' rows right behind them, and the tree runs while those are still on their way (one memory round trip
// in front of the projection's arithmetic instead of two: 512x512 4:2:0 k_project 2.96 -> see profiles/ us to first data)
struct WaveTreeRows {
        double v[kWaveTreeMax / 64];
        unsigned P;
};
__device__ __forceinline__ void norm_tree_load(const double *rowsum, unsigned n, unsigned nch, unsigned c, int lane, WaveTreeRows &t)
{
        unsigned P = 64;
        while(P < n) { P <<= 1; }
        t.P = P;
#pragma unroll
        for(unsigned j = 0; j < kWaveTreeMax / 64; j++) {
                const unsigned i = (unsigned)lane + 64 * j;
                t.v[j] = (j * 64 < P && i < n) ? rowsum[(size_t)i * nch + c] : 0.;
        }
}
__device__ __forceinline__ float norm_tree_reduce(WaveTreeRows &t)
{
        const unsigned P = t.P;
#pragma unroll
        for(unsigned half = kWaveTreeMax / 128; half >= 1; half >>= 1) {      // s = 64 * half
                if(64 * half < P) {
#pragma unroll
                        for(unsigned j = 0; j < half; j++) { t.v[j] = t.v[j] + t.v[j + half]; }
                }
        }
        double m = t.v[0];
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) { m = m + __shfl_down(m, off, 64); }
        m = __shfl(m, 0, 64);
        return sqrtf((float)m);                                               // compute.c:206
}
__device__ __forceinline__ float norm_tree_wave(const double *rowsum, unsigned n, unsigned nch, unsigned c, int lane)
{
        WaveTreeRows t;
        norm_tree_load(rowsum, n, nch, c, lane, t);
        return norm_tree_reduce(t);
}

// ---------------------------------------------------------------------------
// Row-tiled runs inside one process (j2p_tiled): the two per-iteration exchanges as kernels that READ the other
// bands' memory directly (peer access over xGMI, or plain device memory when bands share a GPU).
// ---------------------------------------------------------------------------
struct BandRowsums {
        const double *rowsum[kMaxBands];   // band b's [tile row][channel] level-1 sums (GradArgs::rowsum)
        unsigned first[kMaxBands];         // its first global tile row
        unsigned count[kMaxBands];         // its tile rows
        unsigned nband;
        float *out[kMaxBands];             // where the norm goes: [channel] words of nout solvers (peers' memory included)
        unsigned nout;
};

// level 2 of the norm reduction over the bands' row sums: the same padded pairwise tree as k_norm_finish over the
// same global array, so the norm — and the result — does not depend on how the canvas was cut.  One block per channel.
__global__ __launch_bounds__(256) void k_norm_bands(BandRowsums t, unsigned nrows_global, unsigned nch)
{
        extern __shared__ __attribute__((aligned(16))) float smem[];
        double *buf = reinterpret_cast<double *>(smem);
        unsigned P = 1;
        while(P < nrows_global) { P <<= 1; }
        const unsigned c = blockIdx.x;
        for(unsigned i = threadIdx.x; i < P; i += 256) { buf[i] = 0.; }
        __syncthreads();
        for(unsigned b = 0; b < t.nband; b++) {
                const double *src = t.rowsum[b];
                for(unsigned i = threadIdx.x; i < t.count[b]; i += 256) { buf[t.first[b] + i] = src[(size_t)i * nch + c]; }
        }
        const double s = tree_sum_lds(buf, nrows_global, P);
        // one band reduces for all: the float goes into every band's own norm word (a store over xGMI for bands on
        // other GPUs; visible to their projection kernels through the event recorded behind this launch)
        const float nrm = sqrtf((float)s);
        if(threadIdx.x < t.nout) { t.out[threadIdx.x][c] = nrm; }
}

// up to 2 * kMaxCh row blocks copied into this band's halo rows from the neighbours' edge rows
struct RowCopies {
        float *dst[2 * kMaxCh];
        const float *src[2 * kMaxCh];
        unsigned n;
        unsigned floats;                   // per copy; a multiple of 2 (W is even)
};
__global__ __launch_bounds__(256) void k_copy_rows(RowCopies t)
{
        const unsigned pairs = t.floats / 2;
        for(unsigned k = 0; k < t.n; k++) {
                const v2f *src = reinterpret_cast<const v2f *>(t.src[k]);
                v2f *dst = reinterpret_cast<v2f *>(t.dst[k]);
                for(unsigned i = blockIdx.x * 256 + threadIdx.x; i < pairs; i += gridDim.x * 256) { dst[i] = src[i]; }
        }
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
//
// The three divisions of this phase — g/||g|| (compute.c:213), (cos-dq)/q^2
// (compute.c:49) and, when logging, (cos-dq)/q (compute_simd_step.c:22) — have
// a wave-uniform or per-coefficient-position denominator, so the refined
// reciprocal of the IEEE division sequence is computed once (per wave / per
// table entry) and every quotient costs five packed fma.  As in phase A the
// short sequence equals `/` bit for bit whenever the compiler's version would
// not rescale: denominators in [2^-20, 2^26], numerators 0 or in [2^-100, 2^61)
// (then quotients are normal and every residual is exact).  Numerators are
// screened per wavefront, denominators per launch; anything else takes `/`.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float stepped(const ChanDev &k, ptrdiff_t off, float factor, float step, float norm)
{
        J2P_CHK(k, x_own[0], k.xcur + off, 4, 201);
        J2P_CHK(k, x_own[1], k.xprev + off, 4, 202);
        J2P_CHK(k, grad, k.grad + off, 4, 203);
        const float xc = k.xcur[off], xp = k.xprev[off];
        const float y = xc + factor * (xc - xp);                 // compute.c:435
        if(norm != 0.f) { return y - step * (k.grad[off] / norm); }   // compute.c:213
        return y;
}

// numerator screen of phase B's short division — suspect: 0 < |x| < 2^-100, |x| >= 2^61, or NaN — for a batch of values at
// once, on the bit patterns (as k_gradient's make_y does): hi = largest |x|, lo = smallest NON-ZERO |x| minus one ulp
// (0 - 1 wraps to the top, so zeros drop out of the minimum); two compares at the end
struct NumScreen {
        unsigned hi = 0u, lo = ~0u;
        __device__ __forceinline__ void add(v2f x)
        {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u u = __builtin_bit_cast(v2u, x) & 0x7fffffffu;
                const v2u um = u - 1u;
                hi = max(hi, max(u.x, u.y));
                lo = min(lo, min(um.x, um.y));
        }
        __device__ __forceinline__ bool suspect() const
        {
                constexpr unsigned kLo = 0x0d800000u, kHi = 0x5e000000u;     // bits of 2^-100 and 2^61 (NaN and infinity lie above)
                return hi >= kHi || lo < kLo - 1u;
        }
};
__device__ __forceinline__ bool den_ok(float d) { return d >= 0x1p-20f && d <= 0x1p26f; }


// Stores into a NEIGHBOURING BAND's halo rows (ProjArgs::halo_up / halo_down: another GPU's memory over xGMI, or this
// GPU's) are system-scope write-through stores (sc0 sc1: performed at the destination, nothing left dirty in this XCD's
// L2), so that they are visible to the peer as soon as the kernel has ended whatever publishes that fact — an event
// record with its system-scope release, or a value written behind the kernel (J2P_TILED_WAIT=counter), which releases
// nothing by itself.  Only the strips of a band's first / last block row get here: 4- and 8-byte stores are fast enough.
// -DJ2P_EXP_DROP_HALO_PUSH (fault injection, tests/test_tiled_verify_gpu.py): the rows are NOT pushed — the exchange is
// broken on purpose, and j2p_tiled_create's verification has to notice and demote it.
__device__ __forceinline__ void peer_store(float *p, float v)
{
#ifndef J2P_EXP_DROP_HALO_PUSH
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
        (void)p; (void)v;
#endif
}
__device__ __forceinline__ void peer_store2(float *p, float a, float b)      // p is 8-byte aligned
{
#ifndef J2P_EXP_DROP_HALO_PUSH
        const v2f v = v2f{a, b};
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
#else
        (void)p; (void)a; (void)b;
#endif
}
template <int WS, class VWS>
__device__ __forceinline__ void peer_store_ws(float *p, VWS o)
{
        if constexpr(WS == 2) { peer_store2(p, o[0], o[1]); }
        else {
#pragma unroll
                for(int i = 0; i < WS; i++) { peer_store(p + i, o[i]); }
        }
}

// Phase B front/back end for a SUBSAMPLED channel whose 64 x 8 coefficient strip lies wholly
// inside the canvas: lane = coefficient column = WS canvas columns, 8*HS canvas rows.
// Keeps the stepped pixels in registers between the block-mean (compute.c:348-360) and the
// add-back of the new mean onto the residual (compute.c:361-368, 390-402).
template <int WS, int HS>
struct SubTile {
        float f[8 * HS][WS];
};

template <int WS, int HS>
__device__ __forceinline__ void sub_load_step_mean(const ChanDev &k, size_t base /* (row, col) of the lane's first pixel */,
                                                   unsigned W, float factor, float step, float norm,
                                                   SubTile<WS, HS> &t, float (&mean)[8])
{
        typedef float vws __attribute__((ext_vector_type(WS)));
        const bool have_norm = norm != 0.f;                                   // compute.c:212
        const bool fast = den_ok(norm);
        const float rn = fast ? 1.f / norm : 0.f;                             // correctly rounded: what div_exact_recip needs
#pragma unroll
        for(int half = 0; half < 2; half++) {
                // half the rows at a time: bounds the registers held by loads in flight
                vws gv[4 * HS], xc[4 * HS], xp[4 * HS];
#pragma unroll
                for(int i = 0; i < 4 * HS; i++) {
                        const size_t off = base + (size_t)(half * 4 * HS + i) * W;
                        J2P_CHK(k, grad, k.grad + off, 4 * WS, 204);
                        J2P_CHK(k, x_own[0], k.xcur + off, 4 * WS, 205);
                        J2P_CHK(k, x_own[1], k.xprev + off, 4 * WS, 206);
                        gv[i] = *reinterpret_cast<const vws *>(k.grad + off);
                        xc[i] = *reinterpret_cast<const vws *>(k.xcur + off);
                        xp[i] = *reinterpret_cast<const vws *>(k.xprev + off);
                }
                NumScreen scr;
                if(WS == 2) {
#pragma unroll
                        for(int i = 0; i < 4 * HS; i++) { scr.add(v2f{gv[i][0], gv[i][WS - 1]}); }
                } else {
#pragma unroll
                        for(int i = 0; i < 4 * HS; i += 2) { scr.add(v2f{gv[i][0], gv[i + 1][0]}); }
                }
                const bool use_fast = fast && __builtin_amdgcn_ballot_w64(scr.suspect()) == 0;
#pragma unroll
                for(int i = 0; i < 4 * HS; i++) {
                        vws y = xc[i] + factor * (xc[i] - xp[i]);                 // compute.c:435
                        if(have_norm) {
                                vws q;
                                if(use_fast) {
                                        if(WS == 2) {
                                                const v2f qq = div_exact_recip(v2f{gv[i][0], gv[i][WS - 1]}, v2f{norm, norm}, v2f{rn, rn});
                                                q[0] = qq.x;
                                                q[WS - 1] = qq.y;
                                        } else {
                                                const v2f qq = div_exact_recip(v2f{gv[i][0], 0.f}, v2f{norm, norm}, v2f{rn, rn});
                                                q[0] = qq.x;
                                        }
                                } else {
#pragma unroll
                                        for(int j = 0; j < WS; j++) { q[j] = gv[i][j] / norm; }
                                }
                                y = y - step * q;                                     // compute.c:213
                        }
#pragma unroll
                        for(int j = 0; j < WS; j++) { t.f[half * 4 * HS + i][j] = y[j]; }
                }
        }
#pragma unroll
        for(int r = 0; r < 8; r++) {
                float m = 0.f;
#pragma unroll
                for(int sy = 0; sy < HS; sy++) {
#pragma unroll
                        for(int sx = 0; sx < WS; sx++) { m += t.f[r * HS + sy][sx]; }
                }
                mean[r] = m / (float)(WS * HS);
        }
}

// halo_up / halo_down: non-NULL when the strip is the band's first / last block row and a neighbouring band wants its
// first / last kHalo rows too (ProjArgs::halo_up, halo_down), already offset to the lane's first column
template <int WS, int HS>
__device__ __forceinline__ void sub_store_residual(const ChanDev &k, size_t base, unsigned W, const SubTile<WS, HS> &t,
                                                   const float (&mean_old)[8], const float (&mean_new)[8],
                                                   float *halo_up = nullptr, float *halo_down = nullptr)
{
        typedef float vws __attribute__((ext_vector_type(WS)));
#pragma unroll
        for(int r = 0; r < 8; r++) {
#pragma unroll
                for(int sy = 0; sy < HS; sy++) {
                        vws o;
#pragma unroll
                        for(int sx = 0; sx < WS; sx++) {
                                const float res = t.f[r * HS + sy][sx] - mean_old[r];     // compute.c:365
                                o[sx] = res + mean_new[r];                                // compute.c:398
                        }
                        J2P_CHK(k, x_own[1], k.xprev + base + (size_t)(r * HS + sy) * W, 4 * WS, 207);
                        *reinterpret_cast<vws *>(k.xprev + base + (size_t)(r * HS + sy) * W) = o;
                        // (the strip covers whole block rows of the band: its rows 0, 1 / 8 HS - 2, 8 HS - 1 are the band's)
                        const int row = r * HS + sy;
                        if(row < kHalo && halo_up) { peer_store_ws<WS>(halo_up + (size_t)row * W, o); }
                        if(row >= 8 * HS - kHalo && halo_down) { peer_store_ws<WS>(halo_down + (size_t)(row - (8 * HS - kHalo)) * W, o); }
                }
        }
}

// WS, HS: the subsampling this instantiation has a register-resident fast path for
// (1,1 = full-resolution channel; 0,0 = any other sampling, generic path only).  Strips that
// stick out of the canvas or of the channel's coverage always take the generic path.
struct __attribute__((aligned(16))) ProjShared {
        float tp[4 * kTpWave];
        float qs[64];    // q
        float qq[64];    // q*q
        float rqq[64];   // 1/(q*q), correctly rounded
        float rq[64];    // 1/q, correctly rounded   (log only)
        int q_fast;
        float norm_ws;   // NIP == 2: ||g|| as the workgroup's first wavefront reduced it
};

// NIP: 0 = ||g|| is read from ProjArgs::norm; otherwise it is reduced here from the level-1 row sums the gradient launch
// left behind (ProjArgs::norm_rowsums), so that no reduction launch stands between the phases — 1: by EVERY wavefront, its
// loads in flight together with the strip's rows (small canvases: one memory round trip in front of the arithmetic);
// 2: by the workgroup's FIRST wavefront, before anything else, and handed to the other three through LDS at the barrier
// that publishes the quantisation tables anyway (bands of a row-tiled run: up to 1024 row sums, a quarter of the
// loads and none of the registers of form 1 — the tree's values are dead before the row loads are issued)
template <bool LOG, int WS, int HS, int NT, int NIP, bool PTR = false>
__device__ __forceinline__ void project_strip(const ProjArgs &a, ProjShared &sh)
{
        const unsigned wg = blockIdx.x, nwg = gridDim.x;
        float *const tp = sh.tp;
        float *const qs = sh.qs, *const qq = sh.qq, *const rqq = sh.rqq, *const rq = sh.rq;
        int &q_fast = sh.q_fast;
#ifdef J2P_TRACE
        const unsigned long long tr_start = trace_now();
        unsigned long long tr_data = 0;
#ifdef J2P_TRACE_CLOCK
        const unsigned long long tr_core = clock64();
#endif
#endif

        const unsigned zi = blockIdx.z;
        const int c = (int)a.chan_of_z[zi];
        const ChanDev &k = a.ch[c];
        const int lane = (int)threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // uniform: strip / row arithmetic and the row resources stay scalar
        const unsigned W = a.geo.W;
        const unsigned ws = k.ws, hs = k.hs;
        const unsigned strips_x = (W + 64 * ws - 1) / (64 * ws);      // strips across the canvas
        const unsigned brows = (a.geo.rows + 8 * hs - 1) / (8 * hs);  // block rows in the band
        // workgroup b runs on XCD b % 8: give every XCD a contiguous run of strips (as k_gradient does) — the eight
        // L2s then each stream one region of the planes instead of interleaving at 1 KB (68.8 -> 67.8 us at 4096^2)
        unsigned lstrip;
        {
                const unsigned b = wg, xcd = b & 7, q = nwg >> 3, rem = nwg & 7;
                const unsigned l = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
                lstrip = l * 4 + wave;                                // index within this launch
        }
        // (wavefronts past the launch's last strip stay until the workgroup barrier below)
        const bool active = lstrip < strips_x * a.nby[zi];
        if(!active) { lstrip = 0; }
        const unsigned by = a.by_offset[zi] + (lstrip / strips_x) * a.by_mul[zi], sx = lstrip % strips_x;
        const unsigned strip = by * strips_x + sx;                    // index within the band
        (void)brows;
        float *scratch = tp + wave * kTpWave;

        // (NIP is a template parameter because the tree costs two registers: 82 instead of 80, i.e. five instead of
        // six wavefronts per SIMD, 1.3 us per launch at 4096^2, where it is not used)
        float norm;
        WaveTreeRows tree;
        if constexpr(NIP == 1) { norm_tree_load(a.norm_rowsums, a.norm_rows, a.norm_nch, (unsigned)c, lane, tree); }   // reduced below, behind the row loads
        else if constexpr(NIP == 2) {
                if(wave == 0) {
                        const float nrm = norm_tree_wave(a.norm_rowsums, a.norm_rows, a.norm_nch, (unsigned)c, lane);
                        if(lane == 0) { sh.norm_ws = nrm; }
                }
        }
        else { norm = a.norm[c]; }
        const unsigned cx = sx * 64 + lane;                           // coefficient column of this lane
        const unsigned cy0 = (a.geo.row0 / hs) + by * 8;              // first coefficient row (global)
        const bool covered = cx < k.cw && cy0 < k.ch;                 // block-granular: cw, ch multiples of 8
        const unsigned ly0 = by * 8 * hs;                             // band-local canvas row
        // A full-resolution channel whose coefficient plane is smaller than the canvas (the chroma planes pad
        // further than the luma plane: most 4:2:0 images) still goes through the reference's resampling code
        // with a 1 x 1 footprint (compute.c:348-370, 390-403): the DCT sees 0.f + x and the result is
        // (x - (0.f + x)) + p.  That only differs from p itself in the sign of a zero, but it does differ.
        const bool unit = ws == 1 && hs == 1;
        const bool resample1 = unit && (k.cw != W || k.ch != a.geo.H);
        const bool direct = unit && !resample1;
        // wave-uniform: the whole 64 x 8 strip is inside the canvas and projected
        const bool full = active && WS == 1 && HS == 1 && unit && sx * 64 + 64 <= k.cw && sx * 64 + 64 <= W && cy0 < k.ch &&
                          ly0 + 8 <= a.geo.rows;
        // band instantiations: strips of the band's first / last block row also store into the neighbours' halo rows
        // (ProjArgs::halo_up / halo_down).  Wave-uniform pointers; NULL for every other strip.
        constexpr bool PUSH = NIP == 2;
        float *push_up = nullptr, *push_down = nullptr;
        if constexpr(PUSH) {
                if(ly0 < (unsigned)kHalo) { push_up = a.halo_up[c]; }
                if(ly0 + 8 * hs + (unsigned)kHalo > a.geo.rows) { push_down = a.halo_down[c]; }
        }
        // where band-local canvas row ly of the new iterate goes besides the band's own plane (row start), or NULL
        auto halo_copy_of = [&](unsigned ly) -> float * {
                if(push_up && ly < (unsigned)kHalo) { return push_up + (size_t)ly * W; }
                if(push_down && ly + (unsigned)kHalo >= a.geo.rows && ly < a.geo.rows) { return push_down + (size_t)(ly + kHalo - a.geo.rows) * W; }
                return nullptr;
        };

        // The common path puts its 24 loads in flight FIRST; the quantisation tables (one more dependent global load,
        // then LDS and a workgroup barrier) are set up while they fly: one memory round trip in front of the
        // arithmetic instead of two.
        float gv[8], xcv[8], xpv[8];
        if(full) {
                const size_t base = (size_t)ly0 * W + cx;             // (the pointer form: what J2P_DEBUG checks)
                (void)base;
                // wave-uniform resources at the strip's first row; lane = column, rows by scalar offset (see rows_from)
                const __amdgpu_buffer_rsrc_t rc_ = rows_from(k.xcur + (size_t)ly0 * W), rp_ = rows_from(k.xprev + (size_t)ly0 * W),
                                             rg_ = rows_from(k.grad + (size_t)ly0 * W);
                const unsigned lane_off = cx * 4u;
                // (array by array: 0.5 % faster than row by row.)  PTR: the same loads through flat pointers — what canvases
                // whose rows lie >= 64 KiB apart take: there, and only there, the buffer form measures 5 % SLOWER
                // (16384 x 2048: 116.7 -> 122.9 us; 8192 x 4096, same bytes: 116.4 -> 115.9; 4096^2 62.3 -> 61.2;
                // 2048^2 22.4 -> 20.8), whatever the order of the 24 loads and at six or seven wavefronts per SIMD
                // (profiles/r03_ab_buffer_addressing.jsonl)
#pragma unroll
                for(int r = 0; r < 8; r++) {
                        J2P_CHK(k, x_own[0], &k.xcur[base + (size_t)r * W], 4, 209);
                        if constexpr(PTR) { xcv[r] = k.xcur[base + (size_t)r * W]; }
                        else { xcv[r] = buf_load1<false>(rc_, lane_off, (unsigned)r * W * 4u); }
                }
#pragma unroll
                for(int r = 0; r < 8; r++) {
                        J2P_CHK(k, x_own[1], &k.xprev[base + (size_t)r * W], 4, 210);
                        if constexpr(PTR) { xpv[r] = k.xprev[base + (size_t)r * W]; }
                        else { xpv[r] = buf_load1<false>(rp_, lane_off, (unsigned)r * W * 4u); }
                }
#pragma unroll
                for(int r = 0; r < 8; r++) {
                        J2P_CHK(k, grad, &k.grad[base + (size_t)r * W], 4, 208);
                        if constexpr(PTR) {
                                if constexpr(NT >= 1) { gv[r] = __builtin_nontemporal_load(&k.grad[base + (size_t)r * W]); }
                                else { gv[r] = k.grad[base + (size_t)r * W]; }
                        } else { gv[r] = buf_load1<(NT >= 1)>(rg_, lane_off, (unsigned)r * W * 4u); }
                }
        }
        if constexpr(NIP == 1) { norm = norm_tree_reduce(tree); }
        if(threadIdx.x < 64) {
                const float q = k.q[threadIdx.x];
                qs[threadIdx.x] = q;
                qq[threadIdx.x] = q * q;
                rqq[threadIdx.x] = 1.f / (q * q);                    // correctly rounded reciprocals (div_exact_recip)
                rq[threadIdx.x] = 1.f / q;
                const bool ok = den_ok(q * q) && den_ok(q);
                const unsigned long long all_ok = __builtin_amdgcn_ballot_w64(ok);
                if(threadIdx.x == 0) { q_fast = all_ok == ~0ull; }
        }
        __syncthreads();
        if constexpr(NIP == 2) { norm = sh.norm_ws; }

        if(!active) { return; }

        // subsampled channel, strip wholly inside canvas and coverage: register-resident fast path
        constexpr bool kSub = WS * HS > 1;
        const bool fullsub = kSub && ws == (unsigned)WS && hs == (unsigned)HS && sx * 64 + 64 <= k.cw &&
                             (sx * 64 + 64) * ws <= W && cy0 + 8 <= k.ch && ly0 + 8 * hs <= a.geo.rows;
        const size_t sub_base = (size_t)ly0 * W + (size_t)cx * ws;
        SubTile<(kSub ? WS : 1), (kSub ? HS : 1)> tile;

        float v[8];
        float st1[8];                                                   // stepped pixels of a `full && resample1` strip
        if(fullsub) {
                sub_load_step_mean<(kSub ? WS : 1), (kSub ? HS : 1)>(k, sub_base, W, a.factor, a.step, norm, tile, v);
        } else if(full) {
                v2f y2[4], g2[4];
                NumScreen scr;
#pragma unroll
                for(int p = 0; p < 4; p++) {
                        const v2f xc = v2f{xcv[2 * p], xcv[2 * p + 1]}, xp = v2f{xpv[2 * p], xpv[2 * p + 1]};
                        g2[p] = v2f{gv[2 * p], gv[2 * p + 1]};
                        y2[p] = xc + a.factor * (xc - xp);                  // compute.c:435
                        scr.add(g2[p]);
                }
                if(norm != 0.f) {                                           // compute.c:212
                        if(den_ok(norm) && __builtin_amdgcn_ballot_w64(scr.suspect()) == 0) {
                                const v2f nn = v2f{norm, norm};
                                const float rn1 = 1.f / norm;               // correctly rounded, once per wavefront
                                const v2f rn = v2f{rn1, rn1};
#pragma unroll
                                for(int p = 0; p < 4; p++) { y2[p] = y2[p] - a.step * div_exact_recip(g2[p], nn, rn); }
                        } else {
#pragma unroll
                                for(int p = 0; p < 4; p++) { y2[p] = y2[p] - a.step * v2f{g2[p].x / norm, g2[p].y / norm}; }
                        }
                }
#pragma unroll
                for(int p = 0; p < 4; p++) {
                        v[2 * p] = y2[p].x;
                        v[2 * p + 1] = y2[p].y;
                }
                if(resample1) {
#pragma unroll
                        for(int r = 0; r < 8; r++) {
                                st1[r] = v[r];
                                v[r] = 0.f + v[r];                          // mean of one sample (compute.c:352-358)
                        }
                }
        } else if(direct) {
                const bool inside = cx < W;
#pragma unroll
                for(int r = 0; r < 8; r++) {
                        v[r] = 0.f;
                        if(inside && ly0 + r < a.geo.rows) {
                                const ptrdiff_t off = (ptrdiff_t)(ly0 + r) * W + cx;
                                v[r] = stepped(k, off, a.factor, a.step, norm);
                                if(!covered) {
                                        k.xprev[off] = v[r];           // stepped but never projected (SURVEY §7 hard part 5); address checked by stepped()
                                        if constexpr(PUSH) {
                                                if(float *h = halo_copy_of(ly0 + r)) { peer_store(h + cx, v[r]); }
                                        }
                                }
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
                                                if(covered) { mean += f; }
                                                else {
                                                        k.xprev[off] = f;
                                                        if constexpr(PUSH) {
                                                                if(float *h = halo_copy_of(ly)) { peer_store(h + x, f); }
                                                        }
                                                }
                                        }
                                }
                        }
                        v[r] = mean / (float)(ws * hs);
                }
        }
        float mean_old[8];
        if(!direct) {
#pragma unroll
                for(int r = 0; r < 8; r++) { mean_old[r] = v[r]; }
        }
#ifdef J2P_TRACE
        asm volatile("; stepped pixels exist" ::"v"(v[0]) : "memory");    // (needs the loads: the stamp cannot move above them)
        tr_data = trace_now();
#endif

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
                // the lane's eight coefficients d as floats.  Wide form: 16 bytes of int16, two per dword (low half = even
                // coefficient).  Narrow form (ChanDev::d8, wave-uniform: channels all of whose |d| are <= 127 — every
                // coefficient of a Q <= 50 image): 8 bytes of d + 128, (float)byte - 128.f is d exactly; one byte per pixel
                // less for the phase to read
                v2f dfs[4];
                const size_t blk = bcov ? (size_t)(cy0 / 8 - k.crow0 / 8) * (k.cw / 8) + bx : 0;
                if(k.d8) {
                        typedef unsigned v2u __attribute__((ext_vector_type(2)));
                        v2u rb = v2u{0x80808080u, 0x80808080u};
                        if(bcov) {
                                J2P_CHK(k, d, k.d8 + blk * 64 + rr * 8, 8, 214);
                                if constexpr(NT >= 3) { rb = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(k.d8 + blk * 64 + rr * 8)); }
                                else { rb = *reinterpret_cast<const v2u *>(k.d8 + blk * 64 + rr * 8); }
                        }
                        const unsigned rw[2] = {rb.x, rb.y};
#pragma unroll
                        for(int p = 0; p < 4; p++) {
                                const unsigned w = rw[p >> 1] >> (16 * (p & 1));
                                dfs[p] = v2f{(float)(w & 0xffu), (float)((w >> 8) & 0xffu)} - 128.f;
                        }
                } else {
                        int4 raw = make_int4(0, 0, 0, 0);
                        if(bcov) {
                                J2P_CHK(k, d, k.d + blk * 64 + rr * 8, 16, 211);
                                if constexpr(NT >= 3) {
                                        typedef int v4i __attribute__((ext_vector_type(4)));
                                        const v4i rv = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(k.d + blk * 64 + rr * 8));
                                        raw = make_int4(rv.x, rv.y, rv.z, rv.w);
                                } else {
                                        raw = *reinterpret_cast<const int4 *>(k.d + blk * 64 + rr * 8);
                                }
                        }
                        const int rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                        for(int p = 0; p < 4; p++) { dfs[p] = v2f{(float)(short)(rw[p] & 0xffff), (float)(rw[p] >> 16)}; }
                }
                v2f t2[4], q2[4];
                NumScreen scr;
#pragma unroll
                for(int p = 0; p < 4; p++) {
                        const v2f df = dfs[p];
                        const v2f q = *reinterpret_cast<const v2f *>(&qs[rr * 8 + 2 * p]);
                        const v2f lo = (df - 0.5f) * q, hi = (df + 0.5f) * q;
                        v2f x = v2f{v[2 * p], v[2 * p + 1]};
                        // x > hi ? hi : (x < lo ? lo : x)  (compute.c:327-329) as the median of the three: lo < hi, neither is
                        // a zero (q >= 1, d integer), so whichever operand is returned carries the same bits.
                        // PRECONDITION: x is not a NaN.  The reference's expression hands a NaN through (both compares are
                        // false) and the image goes NaN from there; v_med3_f32 would return lo instead.  A NaN can only get
                        // here from non-finite input planes (the iteration itself produces none from finite state: every
                        // division is guarded by a norm != 0 test, compute.c:97,158,212), and what the reference makes of
                        // such input — NaN everywhere after two iterations — is not a result worth reproducing; the parity
                        // statements are for finite input.
                        x = v2f{__builtin_amdgcn_fmed3f(x.x, lo.x, hi.x), __builtin_amdgcn_fmed3f(x.y, lo.y, hi.y)};
                        v[2 * p] = x.x;
                        v[2 * p + 1] = x.y;
                        t2[p] = x - df * q;
                        q2[p] = q;
                        scr.add(t2[p]);
                }
                if(k.prob_on) {
                        if(q_fast && __builtin_amdgcn_ballot_w64(scr.suspect()) == 0) {
#pragma unroll
                                for(int p = 0; p < 4; p++) {
                                        const v2f d2 = *reinterpret_cast<const v2f *>(&qq[rr * 8 + 2 * p]);
                                        const v2f r2 = *reinterpret_cast<const v2f *>(&rqq[rr * 8 + 2 * p]);
                                        const v2f ev = div_exact_recip(t2[p], d2, r2);
                                        e[2 * p] = ev.x;
                                        e[2 * p + 1] = ev.y;
                                        if(LOG) {                                    // compute_simd_step.c:22-26
                                                const v2f r1 = *reinterpret_cast<const v2f *>(&rq[rr * 8 + 2 * p]);
                                                const v2f tq = div_exact_recip(t2[p], q2[p], r1);
                                                const v2f sq = tq * tq;
                                                dist += (double)sq.x;
                                                dist += (double)sq.y;
                                        }
                                }
                        } else {
#pragma unroll
                                for(int p = 0; p < 4; p++) {
                                        const v2f d2 = q2[p] * q2[p];
                                        e[2 * p] = t2[p].x / d2.x;
                                        e[2 * p + 1] = t2[p].y / d2.y;
                                        if(LOG) {
                                                const v2f tq = v2f{t2[p].x / q2[p].x, t2[p].y / q2[p].y};
                                                const v2f sq = tq * tq;
                                                dist += (double)sq.x;
                                                dist += (double)sq.y;
                                        }
                                }
                        }
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
                        J2P_CHK(k, x_own[1], dst, 32, 212);
                        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                        if constexpr(PUSH) {
                                if(float *h = halo_copy_of(ly0 + rr)) {
#pragma unroll
                                        for(int u = 0; u < 8; u += 2) { peer_store2(h + bx * 8 + u, v[u], v[u + 1]); }
                                }
                        }
                }
        } else {
                // back to lane = coefficient column, then add the new mean onto the residual (compute.c:365,398)
                transpose8(v, scratch, lane);
                if(fullsub) {
                        sub_store_residual<(kSub ? WS : 1), (kSub ? HS : 1)>(k, sub_base, W, tile, mean_old, v,
                                                                             push_up ? push_up + (size_t)cx * ws : nullptr,
                                                                             push_down ? push_down + (size_t)cx * ws : nullptr);
                } else if(full) {
                        // full && resample1: residual (x - mean) + new mean, lane = column again
                        const size_t base = (size_t)ly0 * W + cx;
#pragma unroll
                        for(int r = 0; r < 8; r++) {
                                J2P_CHK(k, x_own[1], &k.xprev[base + (size_t)r * W], 4, 213);
                                const float o = (st1[r] - mean_old[r]) + v[r];
                                k.xprev[base + (size_t)r * W] = o;
                                if constexpr(PUSH) {
                                        if(float *h = halo_copy_of(ly0 + r)) { peer_store(h + cx, o); }
                                }
                        }
                } else if(covered) {
#pragma unroll 1
                        for(int r = 0; r < 8; r++) {
                                for(unsigned sy = 0; sy < hs; sy++) {
                                        for(unsigned sxx = 0; sxx < ws; sxx++) {
                                                const unsigned x = cx * ws + sxx, ly = ly0 + r * hs + sy;
                                                const ptrdiff_t off = (ptrdiff_t)ly * W + x;
                                                float f = stepped(k, off, a.factor, a.step, norm);
                                                f = f - mean_old[r];
                                                k.xprev[off] = f + v[r];
                                                if constexpr(PUSH) {
                                                        if(float *h = halo_copy_of(ly)) { peer_store(h + x, f + v[r]); }
                                                }
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
                        J2P_CHK(k, pg, dst, 32, 214);
                        if constexpr(NT >= 2) {
                                typedef float v4f __attribute__((ext_vector_type(4)));
                                __builtin_nontemporal_store(v4f{e[0], e[1], e[2], e[3]}, reinterpret_cast<v4f *>(dst));
                                __builtin_nontemporal_store(v4f{e[4], e[5], e[6], e[7]}, reinterpret_cast<v4f *>(dst) + 1);
                        } else {
                                dst[0] = make_float4(e[0], e[1], e[2], e[3]);
                                dst[1] = make_float4(e[4], e[5], e[6], e[7]);
                        }
                }
                if(LOG) {
                        if(!bcov) { dist = 0.; }
#pragma unroll
                        for(int off = 32; off > 0; off >>= 1) { dist += __shfl_down(dist, off, 64); }
                        if(lane == 0) { a.part_prob[(size_t)c * a.strips_per_chan + strip] = dist; }
                }
        }
#ifdef J2P_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef J2P_TRACE_CLOCK
        tr_data = clock64() - tr_core;
#endif
        trace_put(a.geo.trace, a.geo.trace_cap, a.geo.trace_base, 2u, a.geo.trace_seq, tr_start, tr_data, trace_now());
#endif
}

// NIP: the norm comes from norm_tree_wave (ProjArgs::norm_rowsums) instead of ProjArgs::norm
// (70 registers for the 1x1 form with buffer addressing = seven wavefronts per SIMD, 78 = six in the pointer form; capping
// seven back to six — J2P_PROJECT_MAXWAVES=6 — changes nothing, forcing eight spills)
#ifndef J2P_PROJECT_WAVES
#define J2P_PROJECT_WAVES 0
#endif
template <bool LOG, int WS, int HS, int NT = 0, int NIP = 0, bool PTR = false>
__global__ __launch_bounds__(256, (J2P_PROJECT_WAVES && WS == 1 && HS == 1 && !LOG && !NIP ? J2P_PROJECT_WAVES : 1)) void k_project(ProjArgs a)
{
        __shared__ ProjShared sh;
#ifdef J2P_PROJECT_MAXWAVES     // (experiment: cap the wavefronts per SIMD by the workgroup's LDS footprint instead of raising them)
        __shared__ float occupancy_pad[(160 * 1024 / (J2P_PROJECT_MAXWAVES + 1) - sizeof(ProjShared)) / 4 + 64];
        if(a.geo.W == 0xffffffffu) {            // (never: keeps the allocation alive)
                occupancy_pad[threadIdx.x] = (float)a.geo.H;
                __syncthreads();
                if(occupancy_pad[threadIdx.x ^ 1] == 3.f) { __builtin_trap(); }
        }
#endif
        project_strip<LOG, WS, HS, NT, NIP, PTR>(a, sh);
}

// Small canvases are bound by the number of dependent launches per iteration, not by bytes: there ALL channels
// of an image go into one launch whatever their sampling (blockIdx.z = channel; 1x1 and 2x2 keep their
// register-resident paths, everything else takes the generic one).  Not for large images: the kernel needs the
// registers of its hungriest path for every wavefront.
template <bool LOG, bool NIP>
__global__ __launch_bounds__(256) void k_project_mixed(ProjArgs a)
{
        __shared__ ProjShared sh;
        const ChanDev &k = a.ch[a.chan_of_z[blockIdx.z]];
        if(k.ws == 1 && k.hs == 1) { project_strip<LOG, 1, 1, 0, (NIP ? 1 : 0)>(a, sh); }
        else if(k.ws == 2 && k.hs == 2) { project_strip<LOG, 2, 2, 0, (NIP ? 1 : 0)>(a, sh); }
        else { project_strip<LOG, 0, 0, 0, (NIP ? 1 : 0)>(a, sh); }
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

// The coefficients of a channel once more as one byte each (d + 128), and the largest |d| met: when that is <= 127 the
// projection reads the bytes (ChanDev::d8) — the values it computes with are the same floats.  16 bytes in, 8 out per step.
__global__ __launch_bounds__(256) void k_narrow_coefficients(const int16_t *d, uint8_t *d8, size_t cells, unsigned *maxabs)
{
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        unsigned m = 0;
        const size_t n8 = cells / 8;                                   // (cells is a multiple of 64: whole blocks)
        for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
                const v4i r = *reinterpret_cast<const v4i *>(d + i * 8);
                const int w[4] = {r.x, r.y, r.z, r.w};
                unsigned b[8];
#pragma unroll
                for(int p = 0; p < 4; p++) {
                        const int lo = (short)(w[p] & 0xffff), hi = w[p] >> 16;
                        m = max(m, (unsigned)(lo < 0 ? -lo : lo));
                        m = max(m, (unsigned)(hi < 0 ? -hi : hi));
                        b[2 * p] = (unsigned)(lo + 128) & 0xffu;
                        b[2 * p + 1] = (unsigned)(hi + 128) & 0xffu;
                }
                *reinterpret_cast<v2u *>(d8 + i * 8) = v2u{b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24)};
        }
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) { m = max(m, (unsigned)__shfl_down((int)m, off, 64)); }
        if((threadIdx.x & 63) == 0 && m) { atomicMax(maxabs, m); }
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

// ---------------------------------------------------------------------------
// YCbCr -> RGB of the PNG writer (png.c:37-62) with the luma +128 fix-up of
// jpeg2png.c:156-159, cropped to the image size.  The reference evaluates the
// colour matrix in double, narrows to float for the clamp, scales by
// (1 << bits) / 256 in float and truncates to unsigned; the same here.
// out: 3 bytes per pixel (bits == 8) or 6 bytes, big-endian samples (bits == 16).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned to_sample(double v, float bitfactor)
{
        float x = (float)v;
        x = (double)x > 255. ? 255.f : ((double)x < 0. ? 0.f : x);      // CLAMP(x, 0., 255.), png.c:15-17
        return (unsigned)(x * bitfactor);
}

__global__ __launch_bounds__(256) void k_to_rgb(const float *yp, unsigned ys, const float *cbp, unsigned cbs, const float *crp,
                                                unsigned crs, unsigned w, unsigned h, unsigned bits, uint8_t *out)
{
        const size_t n = (size_t)w * h;
        const float bitfactor = (float)((double)(1 << bits) / 256.);
        for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
                const unsigned x = (unsigned)(i % w), y = (unsigned)(i / w);
                const float yi = (float)((double)yp[(size_t)y * ys + x] + 128.);   // jpeg2png.c:158
                const float cbi = cbp[(size_t)y * cbs + x], cri = crp[(size_t)y * crs + x];
                const unsigned r = to_sample((double)yi + 1.402 * (double)cri, bitfactor);
                const unsigned g = to_sample((double)yi - 0.34414 * (double)cbi - 0.71414 * (double)cri, bitfactor);
                const unsigned b = to_sample((double)yi + 1.772 * (double)cbi, bitfactor);
                if(bits == 8) {
                        uint8_t *o = out + i * 3;
                        o[0] = (uint8_t)(r & 0xff);
                        o[1] = (uint8_t)(g & 0xff);
                        o[2] = (uint8_t)(b & 0xff);
                } else {
                        uint8_t *o = out + i * 6;
                        o[0] = (uint8_t)((r >> 8) & 0xff); o[1] = (uint8_t)(r & 0xff);
                        o[2] = (uint8_t)((g >> 8) & 0xff); o[3] = (uint8_t)(g & 0xff);
                        o[4] = (uint8_t)((b >> 8) & 0xff); o[5] = (uint8_t)(b & 0xff);
                }
        }
}

// ---------------------------------------------------------------------------
// Self-test of the fast division / square root against the compiler's IEEE forms
// on n pseudo-random operand pairs inside the screened range (tests/ only).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned x)
{
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
}
// random float with exponent in [elo, ehi] (biased) and random sign / mantissa
__device__ __forceinline__ float rnd_float(unsigned h, unsigned elo, unsigned ehi)
{
        const unsigned e = elo + (h >> 9) % (ehi - elo + 1);
        return __builtin_bit_cast(float, (h << 31) | (e << 23) | (mix32(h) & 0x7fffffu));
}
__global__ __launch_bounds__(256) void k_math_selftest(size_t n, unsigned seed, unsigned long long *mism /* [2] */)
{
        unsigned long long bad_div = 0, bad_sqrt = 0;
        for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
                const unsigned h0 = mix32((unsigned)i * 2654435761u + seed), h1 = mix32(h0 + 0x9e3779b9u);
                const unsigned h2 = mix32(h1 + 0x9e3779b9u), h3 = mix32(h2 + 0x9e3779b9u);
                // denominators: norms in [2^-44, 2^44]; numerators: 0 or magnitude in [2^-45, 2^45]
                v2f d = v2f{fabsf(rnd_float(h0, 83, 171)), fabsf(rnd_float(h1, 83, 171))};
                v2f x = v2f{rnd_float(h2, 82, 172), rnd_float(h3, 82, 172)};
                if((h3 & 0xff) == 0) { x.x = 0.f; }
                if((h2 & 0xff) == 1) { x.y = d.y; }
                const v2f q = div_shared(x, d, div_prepare(d));
                const v2f qi = v2f{x.x / d.x, x.y / d.y};
                bad_div += __builtin_bit_cast(unsigned, q.x) != __builtin_bit_cast(unsigned, qi.x) && !(q.x == 0.f && qi.x == 0.f);
                bad_div += __builtin_bit_cast(unsigned, q.y) != __builtin_bit_cast(unsigned, qi.y) && !(q.y == 0.f && qi.y == 0.f);
                // square roots: sums of squares in [2^-90, 2^90], or exactly 0, or perfect squares
                v2f sx = v2f{fabsf(rnd_float(h1 ^ h2, 37, 217)), fabsf(rnd_float(h0 ^ h3, 37, 217))};
                if((h0 & 0xff) == 0) { sx.x = 0.f; }
                if((h1 & 0xff) == 1) { sx.y = d.y * d.y; }
                const v2f sf = sqrt_fast(sx);
                const v2f si = v2f{sqrtf(sx.x), sqrtf(sx.y)};
                bad_sqrt += __builtin_bit_cast(unsigned, sf.x) != __builtin_bit_cast(unsigned, si.x);
                bad_sqrt += __builtin_bit_cast(unsigned, sf.y) != __builtin_bit_cast(unsigned, si.y);
                // phase B's short division over its whole operand range: denominators in [2^-20, 2^26] with their IEEE
                // reciprocal, numerators 0 or in [2^-100, 2^61)
                {
                        const v2f dd = v2f{fabsf(rnd_float(h0 ^ h3, 107, 152)), fabsf(rnd_float(h1 ^ h2, 107, 152))};
                        v2f nn = v2f{rnd_float(h3 + h0, 27, 187), rnd_float(h2 + h1, 27, 187)};
                        if((h3 & 0xff) == 9) { nn.x = 0.f; }
                        if((h2 & 0xff) == 11) { nn.y = dd.y; }
                        const v2f qm = div_exact_recip(nn, dd, v2f{1.f / dd.x, 1.f / dd.y});
                        const v2f qi2 = v2f{nn.x / dd.x, nn.y / dd.y};
                        bad_div += __builtin_bit_cast(unsigned, qm.x) != __builtin_bit_cast(unsigned, qi2.x) && !(qm.x == 0.f && qi2.x == 0.f);
                        bad_div += __builtin_bit_cast(unsigned, qm.y) != __builtin_bit_cast(unsigned, qi2.y) && !(qm.y == 0.f && qi2.y == 0.f);
                }
                // the norm -> reciprocal -> quotient chain of source_finish: root through v_rsq_f32, the reciprocal refined
                // twice from that same v_rsq_f32 value (recip_exact), one-correction quotients against `/` by the IEEE root.
                // Radicands as the kernel sees them: 0 or in [2^-88, 2^87); numerators 0 or within 2^45 of the norm.
                {
                        v2f rx = v2f{fabsf(rnd_float(h2 ^ h0, 39, 213)), fabsf(rnd_float(h3 ^ h1, 39, 213))};
                        if((h2 & 0xff) == 7) { rx.x = 0.f; }
                        v2f nn, dd, rr;
                        norm_and_reciprocal<true, false>(rx, nn, dd, rr);
                        const v2f ni = v2f{rx.x == 0.f ? 0x1p-60f : sqrtf(rx.x), rx.y == 0.f ? 0x1p-60f : sqrtf(rx.y)};
                        bad_sqrt += __builtin_bit_cast(unsigned, nn.x) != __builtin_bit_cast(unsigned, ni.x);
                        bad_sqrt += __builtin_bit_cast(unsigned, nn.y) != __builtin_bit_cast(unsigned, ni.y);
                        // numerator = norm * a random factor in [2^-40, 2^2) (differences never exceed their norm by much)
                        v2f num = v2f{ni.x * rnd_float(h1 ^ h3, 87, 128), ni.y * rnd_float(h0 ^ h2, 87, 128)};
                        if((h1 & 0x7f) == 3) { num.x = 0.f; }
                        if((h0 & 0x7f) == 5) { num.y = ni.y; }
                        // (rows whose radicands could give an all-ones norm take the IEEE path in the kernel: skipped here too)
                        const bool skip = allones_candidate(rx + v2f{0x1p-120f, 0x1p-120f}, v2f{1.f, 1.f});
                        const v2f qs = div_exact_recip(num, dd, rr);
                        const v2f qd = v2f{num.x / ni.x, num.y / ni.y};
                        bad_div += !skip && __builtin_bit_cast(unsigned, qs.x) != __builtin_bit_cast(unsigned, qd.x) && !(qs.x == 0.f && qd.x == 0.f);
                        bad_div += !skip && __builtin_bit_cast(unsigned, qs.y) != __builtin_bit_cast(unsigned, qd.y) && !(qs.y == 0.f && qd.y == 0.f);
                }
        }
        if(blockIdx.x == 0 && threadIdx.x == 0) {
                // the zero-norm divisor of sqrt_pair<true, false>
                const v2f z = sqrt_pair<true, false>(v2f{0.f, 0x1p-88f});
                bad_sqrt += z.x != 0x1p-60f;
                bad_sqrt += z.y != 0x1p-44f;
        }
        if(bad_div) { atomicAdd(&mism[0], bad_div); }
        if(bad_sqrt) { atomicAdd(&mism[1], bad_sqrt); }
}

// every float in [2^-100, 2^127): sqrt_rsq and sqrt_fast against sqrtf()
__global__ __launch_bounds__(256) void k_sqrt_exhaustive(unsigned long long *mism /* [2] */)
{
        constexpr unsigned lo = 27u << 23, hi = 254u << 23;          // biased exponents 27 .. 253
        unsigned long long bad_rsq = 0, bad_fast = 0;
        for(unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; 2 * i + 1 < (unsigned long long)(hi - lo);
            i += (unsigned long long)gridDim.x * 256) {
                const v2f x = v2f{__builtin_bit_cast(float, (unsigned)(lo + 2 * i)), __builtin_bit_cast(float, (unsigned)(lo + 2 * i + 1))};
                const v2f want = v2f{sqrtf(x.x), sqrtf(x.y)};
                const v2f a = sqrt_rsq(x), b = sqrt_fast(x);
                bad_rsq += (__builtin_bit_cast(unsigned, a.x) != __builtin_bit_cast(unsigned, want.x)) +
                           (__builtin_bit_cast(unsigned, a.y) != __builtin_bit_cast(unsigned, want.y));
                bad_fast += (__builtin_bit_cast(unsigned, b.x) != __builtin_bit_cast(unsigned, want.x)) +
                            (__builtin_bit_cast(unsigned, b.y) != __builtin_bit_cast(unsigned, want.y));
        }
        if(bad_rsq) { atomicAdd(&mism[0], bad_rsq); }
        if(bad_fast) { atomicAdd(&mism[1], bad_fast); }
}

// ---------------------------------------------------------------------------
// Exhaustive checks behind the SHORT division of the gradient kernel (div_exact_recip): the reciprocal of a norm
// refined by TWO Newton steps from the v_rsq_f32 seed is the correctly rounded 1 / n, and with a correctly rounded
// reciprocal ONE residual correction yields the correctly rounded quotient (Markstein: q0 = RN(a r), e = a - n q0
// exactly, q = RN(q0 + e r)).  Theorems with side conditions — so both steps are enumerated instead of trusted:
//   pass 1: every radicand in [2^-100, 2^127) (all the norms the kernel can meet): n, seed -> r; counts r != 1.f / n
//   pass 2: every radicand in [1, 4) (every mantissa of n, both exponent parities) x every numerator mantissa
//           in [1, 2) — 2^47 quotients; scaling either operand by a power of two scales every step exactly, and the
//           kernel's operand screen keeps all of it normal — counts q != a / n (sign handled by symmetry of RN)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_recip_exhaustive(unsigned long long *mism /* [0] count, [1..8] first offenders (bits of the radicand) */)
{
        constexpr unsigned lo = 27u << 23, hi = 254u << 23;
        unsigned long long bad = 0;
        for(unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; 2 * i + 1 < (unsigned long long)(hi - lo);
            i += (unsigned long long)gridDim.x * 256) {
                const v2f x = v2f{__builtin_bit_cast(float, (unsigned)(lo + 2 * i)), __builtin_bit_cast(float, (unsigned)(lo + 2 * i + 1))};
                v2f seed;
                const v2f n = sqrt_rsq(x, seed);
                const v2f r = recip_exact(n, seed);
                const v2f want = v2f{1.f / n.x, 1.f / n.y};
                const bool bx = __builtin_bit_cast(unsigned, r.x) != __builtin_bit_cast(unsigned, want.x);
                const bool by = __builtin_bit_cast(unsigned, r.y) != __builtin_bit_cast(unsigned, want.y);
                if(bx || by) {
                        const unsigned long long k = atomicAdd(&mism[0], (unsigned long long)bx + by);
                        if(k < 8) { mism[1 + k] = __builtin_bit_cast(unsigned, bx ? x.x : x.y); }
                }
                bad += 0;
        }
        (void)bad;
}

// radicands [first, first + count) of the 2^24 floats of [1, 4), each against all 2^23 numerator mantissas
// DIRECT: the denominators themselves are enumerated — [first, first + count) of the 2^23 floats of [1, 2) — with
// r = 1.f / d, the form phase B uses (reciprocals of the norm and of the quantisation table by IEEE division)
template <bool DIRECT>
__global__ __launch_bounds__(256) void k_div_exhaustive(unsigned first, unsigned count, unsigned long long *mism /* [0] count, [1..8]: radicand bits << 32 | numerator bits */)
{
        const unsigned idx = blockIdx.x * 4 + (threadIdx.x >> 6);        // one wavefront per radicand / denominator
        if(idx >= count) { return; }
        const int lane = (int)threadIdx.x & 63;
        const float x = __builtin_bit_cast(float, 0x3f800000u + first + idx);
        v2f seed, n, r;
        if(DIRECT) {
                n = v2f{x, x};
                r = v2f{1.f / x, 1.f / x};
        } else {
                n = sqrt_rsq(v2f{x, x}, seed);
                r = recip_exact(n, seed);
        }
        unsigned long long bad = 0;
        // lane l takes numerator mantissas 2 l, 2 l + 1, then + 128, ...
        for(unsigned m = (unsigned)lane * 2; m < (1u << 23); m += 128) {
                const v2f a = v2f{__builtin_bit_cast(float, 0x3f800000u + m), __builtin_bit_cast(float, 0x3f800000u + m + 1)};
                const v2f q = div_exact_recip(a, n, r);
                const v2f want = v2f{a.x / n.x, a.y / n.y};
                const bool bx = __builtin_bit_cast(unsigned, q.x) != __builtin_bit_cast(unsigned, want.x);
                const bool by = __builtin_bit_cast(unsigned, q.y) != __builtin_bit_cast(unsigned, want.y);
                if(bx || by) {
                        bad += (unsigned long long)bx + by;
                        const unsigned long long k = atomicAdd(&mism[9], 1ull);
                        if(k < 8) { mism[1 + k] = ((unsigned long long)(0x3f800000u + first + idx) << 32) | (0x3f800000u + m + (bx ? 0 : 1)); }
                }
        }
        if(bad) { atomicAdd(&mism[0], bad); }
}

}  // namespace j2p
