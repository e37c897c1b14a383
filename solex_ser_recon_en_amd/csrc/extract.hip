// Pass B: per-frame column extraction along the fitted line -> raw disks.
// Replaces the frame loop of read_video_improved (reference solex_util.py:93-144).
//
// Algorithmic traffic is tiny (2 samples in, 1 sample out per frame, shift and slit
// row) and the access pattern is a gather, so the design goal is sector efficiency:
//  * a lane owns one slit row y; consecutive lanes are consecutive y.  For rotated
//    files (Width > Height, the usual SER) y runs along the file's column axis, so
//    the 64 lanes of a wave read one contiguous 128-byte run of a file row;
//  * a workgroup builds a [shift][64 rows][64 frames] tile in LDS (row stride padded to
//    66 elements -> conflict-free 2-byte column writes) and writes it out as 128-byte
//    row segments, 16 bytes per lane, instead of 2-byte scattered stores;
//  * the gather is latency bound, so what matters is how many requests a wave keeps in
//    flight: all loads are branch-free (stand-in addresses instead of predicates) and are
//    issued BATCH frames at a time before the first use (S=21 at C2: 211 -> 95 us).
// Arithmetic is float64 with separately rounded products (compile with
// -ffp-contract=off) and a truncating store, to be bit-exact with NumPy.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "shg_common.h"

namespace shg {
thread_local bool t_minmax_slots_zeroed = false;
thread_local void* t_zero_with_fold = nullptr;
thread_local size_t t_zero_with_fold_words = 0;
thread_local void* t_prezeroed = nullptr;
}

namespace {

constexpr int TY = 64;        // slit rows per workgroup (= lanes of a wave)
constexpr int TK = 64;        // output columns (frames) per workgroup
constexpr int TKP = TK + 2;   // padded LDS row stride (33 dwords: odd)
constexpr int SC_MAX = 4;     // shifts per workgroup (2 when the scan has only the two implicit shifts: half the registers)

struct ExtractArgs {
    const void* stack;
    int n_frames;
    int64_t height, width, fstride;
    const int32_t* ind_l;
    const double *lw, *rw;
    int n_shifts;
    uint16_t* disks;
    int64_t row_pitch, plane_stride, n_cols, k_offset;
    int flip_x, vec_store;
    uint32_t* mm;
};

template <typename T, bool ROT, int BATCH, int SC> __global__ __launch_bounds__(256) void k_extract(const ExtractArgs kargs) {
    const T* __restrict__ stack = static_cast<const T*>(kargs.stack);
    const int n_frames = kargs.n_frames, n_shifts = kargs.n_shifts, flip_x = kargs.flip_x, vec_store = kargs.vec_store;
    const int64_t height = kargs.height, width = kargs.width, fstride = kargs.fstride, row_pitch = kargs.row_pitch, plane_stride = kargs.plane_stride,
                  n_cols = kargs.n_cols, k_offset = kargs.k_offset;
    const int32_t* __restrict__ ind_l = kargs.ind_l;
    const double* __restrict__ lw = kargs.lw;
    const double* __restrict__ rw = kargs.rw;
    uint16_t* __restrict__ disks = kargs.disks;
    uint32_t* __restrict__ mm = kargs.mm;
    __shared__ uint16_t tile[SC][TY][TKP];
    const int64_t ih = ROT ? width : height;
    const int lane = threadIdx.x & 63;
    // (the wave's number through readfirstlane: the compiler does not know threadIdx.x >> 6 is the same in all lanes, and without it
    // every frame's index, range test and 64-bit base address below are per-lane VALU work -- 7 of the kernel's 24 instructions per sample)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c0 = (int64_t)blockIdx.x * TK;           // first output column of this workgroup
    const int64_t y = (int64_t)blockIdx.y * TY + lane;
    const int s0 = blockIdx.z * SC;
    const int ns = min(SC, n_shifts - s0);
    const bool y_ok = y < ih;
    constexpr int scale = sizeof(T) == 1 ? 256 : 1;        // video_reader.py:121-122

    // Every load below is unconditional: out-of-range rows / shifts / frames read a valid stand-in address and the
    // result is dropped.  A predicated load (`ok ? f[i] : 0`) becomes a branch with an s_waitcnt at its join, which
    // left two requests in flight per wave; branch-free, a batch issues all its loads back to back and waits once.
    const int64_t yc = y_ok ? y : ih - 1;
    int il[SC];
    const double wl = lw[yc], wr = rw[yc];
    // sample * weight with ONE instruction: D = 2^52 + sample is exact (the integer sits in the low mantissa bits), D * w - 2^52 * w is
    // sample * w exactly, and the fma rounds it once -- the product fl((double)sample * w) the reference forms, without the conversion
    // (v_cvt_f64_u32 runs at a quarter of the rate: this kernel's time was its float64 instructions, not its gather --
    // tools/probes/gather_width_probe.hip reads the same bytes in a third of the time).  Needs 2^52 * w finite: any fit's weights
    // (they lie in [0, 1]); other weights through the C ABI take the plain form, wave by wave.
    const double kl = wl * 0x1p+52, kr = wr * 0x1p+52;
    const bool fast = __all(fabs(wl) < 0x1p+900 && fabs(wr) < 0x1p+900) != 0;
#pragma unroll
    for (int s = 0; s < SC; ++s) il[s] = ind_l[(int64_t)(s0 + min(s, ns - 1)) * ih + yc];

    // element offsets of the left and the right sample inside a frame (32 bits: a frame is far below 4 G samples; with the frame's
    // base address in scalar registers the loads need no address arithmetic of their own)
    // BYTE offsets: `scalar base + 32-bit lane offset` is an addressing mode of the load (a scaled element index is not: its doubling could overflow)
    uint32_t off[SC], offr[SC];
    const int64_t step = ROT ? width : 1;
#pragma unroll
    for (int s = 0; s < SC; ++s) {
        off[s] = (uint32_t)((ROT ? (int64_t)il[s] * width + (width - 1 - yc) : yc * width + il[s]) * (int64_t)sizeof(T));
        offr[s] = off[s] + (uint32_t)(step * (int64_t)sizeof(T));
    }

    __shared__ uint32_t wred[4][SC][2];
    uint32_t vlo[SC], vhi[SC];
#pragma unroll
    for (int s = 0; s < SC; ++s) { vlo[s] = 0xffffu; vhi[s] = 0u; }

    // Frames of this wave: cc = wave, wave + 4, ...  BATCH of them at a time.
    for (int cb = wave; cb < TK; cb += 4 * BATCH) {
        T lv[BATCH][SC], rv[BATCH][SC];
        bool ok[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int cc = cb + 4 * i;
            const int64_t col = c0 + cc;
            const int64_t k = (flip_x ? (n_cols - 1 - col) : col) - k_offset;   // wave-uniform
            ok[i] = cc < TK && col < n_cols && k >= 0 && k < n_frames;
            const char* f = reinterpret_cast<const char*>(stack + (ok[i] ? k : 0) * fstride);
#pragma unroll
            for (int s = 0; s < SC; ++s) {
                lv[i][s] = *reinterpret_cast<const T*>(f + off[s]);
                rv[i][s] = *reinterpret_cast<const T*>(f + offr[s]);
            }
        }
        auto blend = [&](auto fast_form) {                   // (the branch on `fast` sits outside the unrolled loops)
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int cc = cb + 4 * i;
                if (!ok[i] || !y_ok) continue;
#pragma unroll
                for (int s = 0; s < SC; ++s) {
                    if (s < ns) {
                        double v;
                        if (decltype(fast_form)::value) {
                            const double pl = __builtin_fma(__hiloint2double(0x43300000, (int)lv[i][s] * scale), wl, -kl);
                            const double pr = __builtin_fma(__hiloint2double(0x43300000, (int)rv[i][s] * scale), wr, -kr);
                            v = pl + pr;
                        } else {
                            const double l = (double)((int)lv[i][s] * scale);
                            const double r = (double)((int)rv[i][s] * scale);
                            v = l * wl + r * wr;
                        }
                        const uint32_t q = (uint32_t)(int)v & 0xffffu;   // what the stored pixel is (weights outside [0, 1] wrap as the reference's uint16 cast does)
                        tile[s][lane][cc] = (uint16_t)q;
                        vlo[s] = q < vlo[s] ? q : vlo[s];
                        vhi[s] = q > vhi[s] ? q : vhi[s];
                    }
                }
            }
        };
        if (fast) blend(std::true_type{});
        else blend(std::false_type{});
    }

    if (mm) {
        // the planes' minimum and maximum (what the warp clips to, ellipse_to_circle.py:112-114), gathered from the values
        // while they were in registers: wave reduction, then one atomic pair per workgroup and plane into one of 64 slots
        // (same-address atomics serialise chip-wide; k_fold_minmax folds the slots)
#pragma unroll
        for (int s = 0; s < SC; ++s) {
            uint32_t lo = vlo[s], hi = vhi[s];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
                lo = ol < lo ? ol : lo;
                hi = oh > hi ? oh : hi;
            }
            if (lane == 0) { wred[wave][s][0] = lo; wred[wave][s][1] = hi; }
        }
    }
    __syncthreads();
    if (mm && threadIdx.x < ns) {
        const int s = threadIdx.x;
        uint32_t lo = wred[0][s][0], hi = wred[0][s][1];
        for (int wv = 1; wv < 4; ++wv) { lo = wred[wv][s][0] < lo ? wred[wv][s][0] : lo; hi = wred[wv][s][1] > hi ? wred[wv][s][1] : hi; }
        if (hi >= lo) {                                   // the workgroup produced at least one value for this plane
            const int slot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) & 63u);
            uint32_t* m = mm + ((int64_t)(s0 + s) * 64 + slot) * 2;
            atomicMax(&m[0], 0xffffu - lo);              // the minimum as the maximum of the complement: both slots start at zero
            atomicMax(&m[1], hi);
        }
    }

    // write-out: one 16-byte segment (8 columns) per lane, 8 lanes per row
    const int seg = threadIdx.x & 7;
    const int r0 = threadIdx.x >> 3;      // 0..31
    for (int s = 0; s < ns; ++s) {
        uint16_t* plane = disks + (int64_t)(s0 + s) * plane_stride;
        for (int r = r0; r < TY; r += 32) {
            const int64_t yy = (int64_t)blockIdx.y * TY + r;
            if (yy >= ih) break;
            const int64_t col = c0 + seg * 8;
            uint16_t* dst = plane + yy * row_pitch + col;
            const uint16_t* src = &tile[s][r][seg * 8];
            if (vec_store && col + 8 <= n_cols) {
                // the tile only holds columns whose frame this rank owns
                const int64_t ka = (flip_x ? (n_cols - 1 - col) : col) - k_offset;
                const int64_t kb = (flip_x ? (n_cols - 1 - (col + 7)) : (col + 7)) - k_offset;
                if (ka >= 0 && ka < n_frames && kb >= 0 && kb < n_frames) {
                    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
                    *reinterpret_cast<uint4*>(dst) = make_uint4(s32[0], s32[1], s32[2], s32[3]);
                    continue;
                }
            }
            for (int j = 0; j < 8; ++j) {
                const int64_t cj = col + j;
                const int64_t kj = (flip_x ? (n_cols - 1 - cj) : cj) - k_offset;
                if (cj < n_cols && kj >= 0 && kj < n_frames) dst[j] = src[j];
            }
        }
    }
}

// ---- a Doppler stack of consecutive shifts (-w a:b:1): every distinct sample is loaded once ------------------------------------------
// The S shifts of such a scan read the columns base .. base + S of a slit row (base = the fitted column plus the smallest
// shift): S + 1 distinct samples per (row, frame), which the general kernel fetches as 2 S loads in groups of SC shifts (48
// loads for 22 samples at S = 21).  Here a lane loads the S + 1 samples of a frame once and forms all S values from registers;
// the tile is [S][64 rows][DK frames].  Measured (tools/bench_extract.py, tools/pmc_extract.sh, C4's shape): on UN-ROTATED files,
// where every lane's samples sit in a cache line of their own, 335 us against the general kernel's 684; on rotated files 185
// against 107 -- the general kernel's re-reads of shared lines hit L2 (FETCH_SIZE 129 against 154 MB), and with 114 VGPRs and
// 34 KB of LDS it keeps 16 waves per CU where this one (200 VGPRs, 48 KB) keeps 8 for the same total of wave cycles.  So the
// stage uses it for un-rotated files only (shg_stage_extract).
// Rows whose line lies within S columns of the frame's edge (the clamps of solex_util.py:114-119 bite) are redone by the
// general rule from the clamped indices, after the main loop: rare, and bit-identical either way.
constexpr int DS_MAX = 24;             // most shifts of the dense path
struct PlaneOfOffset { int v[DS_MAX]; };

// NW waves per workgroup, DK frames per workgroup (LDS row stride DK + 2: an odd number of dwords)
template <typename T, bool ROT, int BATCH, int NW, int DK>
__global__ __launch_bounds__(64 * NW) void k_extract_dense(const T* __restrict__ stack, int n_frames, int64_t height, int64_t width, int64_t fstride,
                                                       const int32_t* __restrict__ ind_l, const int32_t* __restrict__ base_col,
                                                       const double* __restrict__ lw, const double* __restrict__ rw, int S, PlaneOfOffset plane_of,
                                                       uint16_t* __restrict__ disks, int64_t row_pitch, int64_t plane_stride,
                                                       int64_t n_cols, int64_t k_offset, int flip_x, int vec_store, uint32_t* __restrict__ mm) {
    constexpr int DKP = DK + 2;
    extern __shared__ uint16_t dtile[];                    // [S][TY][DKP]
    __shared__ uint32_t wred[NW][DS_MAX][2];
    __shared__ int any_edge;
    const int64_t ih = ROT ? width : height, iw = ROT ? height : width;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * DK;
    const int64_t y = (int64_t)blockIdx.y * TY + lane;
    const bool y_ok = y < ih;
    const int64_t yc = y_ok ? y : ih - 1;
    constexpr int scale = sizeof(T) == 1 ? 256 : 1;
    const double wl = lw[yc], wr = rw[yc];
    const int base = base_col[yc];
    const bool plain = base >= 0 && (int64_t)base + S <= iw - 1;          // columns base .. base + S all lie inside the frame
    if (threadIdx.x == 0) any_edge = 0;
    // element offset of column `base` inside a frame and the distance between neighbouring columns; a lane near the edge
    // reads stand-in addresses (column 0 ..) here and is redone below
    const int64_t b0 = plain ? base : 0;
    const int64_t step = ROT ? width : 1;
    const int64_t off0 = ROT ? b0 * width + (width - 1 - yc) : yc * width + b0;
    uint32_t vlo[DS_MAX], vhi[DS_MAX];
#pragma unroll
    for (int d = 0; d < DS_MAX; ++d) { vlo[d] = 0xffffu; vhi[d] = 0u; }
    __syncthreads();
    if (y_ok && !plain) any_edge = 1;

    for (int cb = wave; cb < DK; cb += NW * BATCH) {
        T v[BATCH][DS_MAX + 1];
        bool ok[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int cc = cb + NW * i;
            const int64_t col = c0 + cc;
            const int64_t k = (flip_x ? (n_cols - 1 - col) : col) - k_offset;   // wave-uniform
            ok[i] = cc < DK && col < n_cols && k >= 0 && k < n_frames;
            const T* f = stack + (ok[i] ? k : 0) * fstride + off0;
            // (unconditional: a load under `if (d <= S)` becomes a branch with a wait at its join -- one request in flight;
            // the columns beyond S re-read column S)
#pragma unroll
            for (int d = 0; d <= DS_MAX; ++d) v[i][d] = f[(int64_t)(d < S ? d : S) * step];
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int cc = cb + NW * i;
            if (!ok[i] || !y_ok) continue;
#pragma unroll
            for (int d = 0; d < DS_MAX; ++d) {
                if (d < S) {
                    const double l = (double)((int)v[i][d] * scale);
                    const double r = (double)((int)v[i][d + 1] * scale);
                    const double val = l * wl + r * wr;
                    const uint32_t q = (uint32_t)(uint16_t)(int)val;
                    dtile[((size_t)d * TY + lane) * DKP + cc] = (uint16_t)q;
                    if (plain) {
                        vlo[d] = q < vlo[d] ? q : vlo[d];
                        vhi[d] = q > vhi[d] ? q : vhi[d];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (any_edge) {
        // the general rule for the rows the clamps touch: left sample at the clamped index of every shift, right one beside it
        for (int cc = wave; cc < DK; cc += NW) {
            const int64_t col = c0 + cc;
            const int64_t k = (flip_x ? (n_cols - 1 - col) : col) - k_offset;
            if (!(col < n_cols && k >= 0 && k < n_frames) || !y_ok || plain) continue;
            const T* f = stack + k * fstride;
            for (int d = 0; d < S; ++d) {
                const int il = ind_l[(int64_t)plane_of.v[d] * ih + yc];
                const int64_t off = ROT ? (int64_t)il * width + (width - 1 - yc) : yc * width + il;
                const double l = (double)((int)f[off] * scale);
                const double r = (double)((int)f[off + step] * scale);
                const double val = l * wl + r * wr;
                const uint32_t q = (uint32_t)(uint16_t)(int)val;
                dtile[((size_t)d * TY + lane) * DKP + cc] = (uint16_t)q;
                vlo[d] = q < vlo[d] ? q : vlo[d];
                vhi[d] = q > vhi[d] ? q : vhi[d];
            }
        }
    }
    if (mm) {
#pragma unroll
        for (int d = 0; d < DS_MAX; ++d) {
            if (d < S) {
                uint32_t lo = vlo[d], hi = vhi[d];
#pragma unroll
                for (int e = 32; e >= 1; e >>= 1) {
                    const uint32_t ol = __shfl_xor(lo, e), oh = __shfl_xor(hi, e);
                    lo = ol < lo ? ol : lo;
                    hi = oh > hi ? oh : hi;
                }
                if (lane == 0) { wred[wave][d][0] = lo; wred[wave][d][1] = hi; }
            }
        }
    }
    __syncthreads();
    if (mm && threadIdx.x < S) {
        const int d = threadIdx.x;
        uint32_t lo = wred[0][d][0], hi = wred[0][d][1];
        for (int wv = 1; wv < NW; ++wv) { lo = wred[wv][d][0] < lo ? wred[wv][d][0] : lo; hi = wred[wv][d][1] > hi ? wred[wv][d][1] : hi; }
        if (hi >= lo) {
            const int slot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) & 63u);
            uint32_t* m = mm + ((int64_t)plane_of.v[d] * 64 + slot) * 2;
            atomicMax(&m[0], 0xffffu - lo);
            atomicMax(&m[1], hi);
        }
    }
    // write-out: one 16-byte segment (8 columns) per lane, DK / 8 lanes per row; the threads cover (plane, row) pairs
    constexpr int SEGS = DK / 8, ROWS = 64 * NW / SEGS;
    const int seg = threadIdx.x % SEGS;
    const int64_t col = c0 + seg * 8;
    for (int pr = threadIdx.x / SEGS; pr < S * TY; pr += ROWS) {
        const int d = pr / TY, r = pr % TY;
        const int64_t yy = (int64_t)blockIdx.y * TY + r;
        if (yy >= ih) continue;
        uint16_t* dst = disks + (int64_t)plane_of.v[d] * plane_stride + yy * row_pitch + col;
        const uint16_t* src = &dtile[((size_t)d * TY + r) * DKP + seg * 8];
        if (vec_store && col + 8 <= n_cols) {
            const int64_t ka = (flip_x ? (n_cols - 1 - col) : col) - k_offset;
            const int64_t kb = (flip_x ? (n_cols - 1 - (col + 7)) : (col + 7)) - k_offset;
            if (ka >= 0 && ka < n_frames && kb >= 0 && kb < n_frames) {
                const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
                *reinterpret_cast<uint4*>(dst) = make_uint4(s32[0], s32[1], s32[2], s32[3]);
                continue;
            }
        }
        for (int j = 0; j < 8; ++j) {
            const int64_t cj = col + j;
            const int64_t kj = (flip_x ? (n_cols - 1 - cj) : cj) - k_offset;
            if (cj < n_cols && kj >= 0 && kj < n_frames) dst[j] = src[j];
        }
    }
}

// ---- rotated files (Width > Height, the usual SER), round 6: the band kernel --------------------------------------------------------
// On a rotated file the samples a slit row needs from a frame lie in consecutive FILE rows, and the 64 slit rows of a wave in one
// 128-byte run of each.  The general kernel above was thought to be bound by the DRAM efficiency of that gather (round 5); taken
// apart (tools/sweep_band.py: the same launch without its loads, without its stores, without its arithmetic) it was bound by
// instruction issue -- a branch per load and per value, 64-bit address arithmetic per load, a mask and a move per sample, a scalar
// prologue of 585 instructions a wave -- and, once that was gone, by its stores displacing from L2 the file rows that neighbouring
// workgroups share.  What this kernel does about each:
//  * a group of G shifts per workgroup, every sample loaded once: for a Doppler stack of consecutive shifts (CONSEC) the G + 1 file
//    rows the group spans (the sample a shift takes on its right is the one the next takes on its left: 24 loads per slit row and
//    frame at S = 21 instead of 48); for any other list the two rows of each shift;
//  * all of a wave's loads (FPW frames x R rows <= 64, what one wave can have in flight) issued back to back, no branch: a frame's
//    address comes from scalar registers (lane i forms the i-th frame's offset, v_readlane hands it over), a row's offset inside
//    the frame is one lane register shared by all frames -- `scalar base + lane offset` is an addressing mode, no arithmetic per load;
//  * the groups of one (row block, column block) next to each other in dispatch order on ONE XCD (grid (8 ng, nx / 8, ny), x
//    fastest: ids i, i + 8, ... share an XCD), so the file row two groups share -- and the cache lines a 128-byte run of a 4000-byte
//    file row straddles, shared with the row block next door -- are L2 hits: FETCH_SIZE 1.02 x the distinct bytes at C4 (general
//    kernel: 1.47 x);
//  * sample x weight as one fma on the double whose low word IS the sample (2^52 + sample; 2^60 + 256 sample for 8-bit files: the
//    x 256 of video_reader.py:121-122 for free), two neighbouring frames' values packed into one LDS dword with v_perm;
//  * NONTEMPORAL 16-byte stores of the tile's rows (94 -> 80 us at C4: the disks no longer displace the shared file rows), the
//    planes' extrema taken on the way out of LDS with packed minima / maxima.
// C4 (2000 x 2000 x 200, S = 21): 72 us against the general kernel's 97 - 105 (profiles/r06_sweeps.txt has every step, and the
// shapes and the walk-the-shifts kernel that lost).  What bounds it now is memory: loads alone 41 us, stores alone <= 30, both 71.
constexpr int BAND_MAX_S = DS_MAX;
constexpr int BAND_NW = 8;           // waves of a workgroup
struct BandArgs {
    const void* stack;
    int n_frames;
    int64_t height, width, fstride;
    const int32_t* ind_l;            // [S][ih] clamped left columns (solex_util.py:114-119)
    const int32_t* base_col;         // CONSEC: [ih] column of the smallest shift, not clamped
    const double *lw, *rw;
    int S;
    uint16_t* disks;
    int64_t row_pitch, plane_stride, n_cols, k_offset;
    int flip_x, vec_store;
    uint32_t* mm;
    int nx;                          // column blocks (the grid's y is rounded up to eights)
    uint8_t plane_of[BAND_MAX_S];    // CONSEC: plane of the d-th smallest shift
};
typedef unsigned short __attribute__((ext_vector_type(2))) ushort2_t;
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4_t;   // native vector: nontemporal-storable
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}

template <typename T, int G, bool CONSEC>
__global__ __launch_bounds__(64 * BAND_NW) void k_extract_band(const BandArgs a) {
    constexpr int NW = BAND_NW;
    constexpr int DKW = TK / 2 + 1;        // dwords of a tile row (odd: the blend's column writes touch every bank)
    constexpr int FPW = TK / NW;           // frames of a wave: pairs of neighbours
    constexpr int PAIRS = FPW / 2;
    constexpr int R = CONSEC ? G + 1 : 2 * G;     // file rows a group loads per frame
    static_assert(FPW % 2 == 0 && FPW * R <= 64, "pairs of frames; a wave has at most 64 loads in flight");
    extern __shared__ uint32_t btile[];    // [G][TY][DKW]
    const int g = (int)(blockIdx.x >> 3);
    const int bx = (int)(blockIdx.y * 8u + (blockIdx.x & 7u)), by = (int)blockIdx.z;
    if (bx >= a.nx) return;
    const T* __restrict__ stack = static_cast<const T*>(a.stack);
    const int n_frames = a.n_frames, flip_x = a.flip_x, S = a.S;
    const int64_t width = a.width, ih = a.width, iw = a.height, fstride = a.fstride, n_cols = a.n_cols, k_offset = a.k_offset;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c0 = (int64_t)bx * TK;
    const int64_t y = (int64_t)by * TY + lane;
    const bool y_ok = y < ih;
    const int64_t yc = y_ok ? y : ih - 1;
    const int g0 = g * G, ns = min(G, S - g0);
    constexpr int scale = sizeof(T) == 1 ? 256 : 1;        // video_reader.py:121-122
    const double wl = a.lw[yc], wr = a.rw[yc];
    // sample * weight with ONE instruction (see k_extract): the sample as the low word of the double 2^52 + sample (high word
    // 0x43300000); an 8-bit file's as the low word of 2^60 + 256 sample (0x43B00000).  Weights that 2^60 w cannot take (never a fit's:
    // they lie in [0, 1]) get the plain products, workgroup by workgroup.
    constexpr int HI = sizeof(T) == 1 ? 0x43B00000 : 0x43300000;
    const double kl = wl * (sizeof(T) == 1 ? 0x1p+60 : 0x1p+52), kr = wr * (sizeof(T) == 1 ? 0x1p+60 : 0x1p+52);
    const bool fast = __all(fabs(wl) < 0x1p+900 && fabs(wr) < 0x1p+900) != 0;
    // byte offsets of the group's samples inside a frame.  CONSEC: rows base + g0 .. base + g0 + G; a slit row whose line lies within
    // S columns of the frame's edge (the clamps of solex_util.py:114-119 bite) reads stand-in rows g0 .. and is redone by the general
    // rule below.  Otherwise: the clamped left column of each shift and the one beside it.  Shifts beyond a short last group re-read
    // its last one.
    const uint32_t rowb = (uint32_t)(width * (int64_t)sizeof(T));
    const uint32_t colb = (uint32_t)((width - 1 - yc) * (int64_t)sizeof(T));
    uint32_t voff[R];
    bool plain = true;
    if (CONSEC) {
        const int base = a.base_col[yc];
        plain = base >= 0 && (int64_t)base + S <= iw - 1;           // columns base .. base + S all lie inside the frame
        voff[0] = (uint32_t)((plain ? base : 0) + g0) * rowb + colb;
#pragma unroll
        for (int d = 1; d < R; ++d) voff[d] = voff[0] + (uint32_t)min(d, ns) * rowb;
    } else {
#pragma unroll
        for (int d = 0; d < G; ++d) {
            voff[2 * d] = (uint32_t)a.ind_l[(int64_t)(g0 + min(d, ns - 1)) * ih + yc] * rowb + colb;
            voff[2 * d + 1] = voff[2 * d] + rowb;
        }
    }

    // the wave's frames (pairs of neighbours wave, wave + NW, ...): lane i forms the offset of the i-th one (a frame this rank does
    // not hold: frame 0 stands in, its column is never stored)
    const int ncols32 = (int)n_cols;
    uint32_t f_lo, f_hi;
    {
        const int i = lane < FPW ? lane : 0;
        const int col = (int)c0 + 2 * (wave + NW * (i >> 1)) + (i & 1);
        const int k = (flip_x ? (ncols32 - 1 - col) : col) - (int)k_offset;
        const bool ok = col < ncols32 && (unsigned)k < (unsigned)n_frames;
        const uint64_t f = (uint64_t)(uint32_t)(ok ? k : 0) * (uint64_t)(fstride * (int64_t)sizeof(T));
        f_lo = (uint32_t)f;
        f_hi = (uint32_t)(f >> 32);
    }
    uint32_t v[FPW][R];                    // (32-bit words: the load zero-extends, no mask at the use)
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const char* f = reinterpret_cast<const char*>(stack) + (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)f_hi, i) << 32) |
                                                                (uint32_t)__builtin_amdgcn_readlane((int)f_lo, i));
#pragma unroll
        for (int d = 0; d < R; ++d) v[i][d] = (uint32_t)*reinterpret_cast<const T*>(f + voff[d]);
    }

    uint32_t* const my = btile + lane * DKW + wave;        // tile[d][lane][pair]: + d * TY * DKW + NW * p, constants
    auto blend = [&](auto fast_form) {
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            uint32_t qv[2][G];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * p + h;
#pragma unroll
                for (int d = 0; d < G; ++d) {
                    const uint32_t xl = v[i][CONSEC ? d : 2 * d], xr = v[i][CONSEC ? d + 1 : 2 * d + 1];
                    double val;
                    if (decltype(fast_form)::value)
                        val = __builtin_fma(__hiloint2double(HI, (int)xl), wl, -kl) + __builtin_fma(__hiloint2double(HI, (int)xr), wr, -kr);
                    else
                        val = (double)(int)(xl * scale) * wl + (double)(int)(xr * scale) * wr;
                    qv[h][d] = (uint32_t)(int)val;
                }
            }
            // low halves of the two values side by side (weights outside [0, 1] wrap as the reference's uint16 cast does)
#pragma unroll
            for (int d = 0; d < G; ++d) my[d * TY * DKW + NW * p] = __builtin_amdgcn_perm(qv[1][d], qv[0][d], 0x05040100u);
        }
    };
    if (fast) blend(std::true_type{});
    else blend(std::false_type{});

    if (CONSEC && y_ok && !plain) {
        // the general rule for the rows the clamps touch: left sample at the clamped index of every shift, right one beside it.
        // This thread wrote the same tile entries above: no barrier between the two.
        uint16_t* t16 = reinterpret_cast<uint16_t*>(btile);
        for (int i = 0; i < FPW; ++i) {
            const int cc = 2 * (wave + NW * (i >> 1)) + (i & 1);
            const int64_t col = c0 + cc;
            const int64_t k = (flip_x ? (n_cols - 1 - col) : col) - k_offset;
            if (!(col < n_cols && k >= 0 && k < n_frames)) continue;
            const T* f = stack + k * fstride;
            for (int d = 0; d < ns; ++d) {
                const int il = a.ind_l[(int64_t)a.plane_of[g0 + d] * ih + yc];
                const int64_t off = (int64_t)il * width + (width - 1 - yc);
                const double l = (double)((int)f[off] * scale);
                const double r = (double)((int)f[off + width] * scale);
                t16[((size_t)(d * TY + lane) * DKW) * 2 + cc] = (uint16_t)(uint32_t)(int)(l * wl + r * wr);
            }
        }
    }
    __syncthreads();

    // write-out: one 16-byte segment (8 columns) per lane, 8 lanes per row, a wave per shift; the planes' minimum and maximum
    // (what the warp clips to, ellipse_to_circle.py:112-114) are taken here, from the values as stored
    constexpr int SEGS = TK / 8;
    const int seg = lane % SEGS;
    const int64_t col = c0 + seg * 8;
    const int64_t ka = (flip_x ? (n_cols - 1 - col) : col) - k_offset;
    const int64_t kb = (flip_x ? (n_cols - 1 - (col + 7)) : (col + 7)) - k_offset;
    const bool whole = a.vec_store && col + 8 <= n_cols && ka >= 0 && ka < n_frames && kb >= 0 && kb < n_frames;
    const int rows_here = (int)min((int64_t)TY, ih - (int64_t)by * TY);
    for (int d = wave; d < ns; d += NW) {
        const int plane = CONSEC ? a.plane_of[g0 + d] : g0 + d;
        uint16_t* dst0 = a.disks + (int64_t)plane * a.plane_stride + ((int64_t)by * TY) * a.row_pitch + col;
        uint32_t lo = 0xffffffffu, hi = 0u;
        if (whole) {
#pragma unroll 4
            for (int r = lane / SEGS; r < rows_here; r += 64 / SEGS) {
                const uint32_t* s32 = &btile[(d * TY + r) * DKW + seg * 4];
                const u32x4_t o = {s32[0], s32[1], s32[2], s32[3]};
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(dst0 + r * a.row_pitch));
                lo = pk_min_u16(pk_min_u16(lo, o.x), pk_min_u16(pk_min_u16(o.y, o.z), o.w));
                hi = pk_max_u16(pk_max_u16(hi, o.x), pk_max_u16(pk_max_u16(o.y, o.z), o.w));
            }
        } else {
            // (the tile only holds columns whose frame this rank owns)
            for (int r = lane / SEGS; r < rows_here; r += 64 / SEGS) {
                const uint16_t* src = reinterpret_cast<const uint16_t*>(&btile[(d * TY + r) * DKW + seg * 4]);
                uint16_t* dst = dst0 + r * a.row_pitch;
                for (int j = 0; j < 8; ++j) {
                    const int64_t cj = col + j;
                    const int64_t kj = (flip_x ? (n_cols - 1 - cj) : cj) - k_offset;
                    if (cj < n_cols && kj >= 0 && kj < n_frames) {
                        dst[j] = src[j];
                        lo = pk_min_u16(lo, src[j] * 0x10001u);
                        hi = pk_max_u16(hi, src[j] * 0x10001u);
                    }
                }
            }
        }
        if (a.mm) {
            lo = shg::wave_fold_u32(lo, [](uint32_t x, uint32_t y) { return pk_min_u16(x, y); });
            hi = shg::wave_fold_u32(hi, [](uint32_t x, uint32_t y) { return pk_max_u16(x, y); });
            const uint32_t l16 = min(lo & 0xffffu, lo >> 16), h16 = max(hi & 0xffffu, hi >> 16);
            if (lane == 0 && h16 >= l16) {                    // this workgroup stored at least one value of the plane
                // one atomic pair per wave and plane into one of 64 slots (same-address atomics serialise chip-wide; k_fold_minmax folds them)
                const int slot = (int)((bx * 5u + by * 3u) & 63u);
                uint32_t* m = a.mm + ((int64_t)plane * 64 + slot) * 2;
                atomicMax(&m[0], 0xffffu - l16);              // the minimum as the maximum of the complement: both slots start at zero
                atomicMax(&m[1], h16);
            }
        }
    }
}

// one launch of the band kernel: groups of G shifts (CONSEC: 4 .. 7, as few groups as 7 allow, then as even as they come; otherwise 2 or 4)
template <bool CONSEC>
int launch_band(const BandArgs& ba, int bytes_per_px, int64_t ih, hipStream_t st) {
    const int S = ba.S;
    int G;
    if (CONSEC) {
        const int ng7 = (S + 6) / 7;
        G = std::max(4, (S + ng7 - 1) / ng7);
    } else {
        G = S <= 2 ? 2 : 4;
    }
    const int ng = (S + G - 1) / G;
    const dim3 grid(8u * (unsigned)ng, ((unsigned)ba.nx + 7u) / 8u, (unsigned)((ih + TY - 1) / TY));
    int status = SHG_E_UNSUPPORTED;
#define SHG_BAND(T, GV)                                                                                                                   \
    if (G == GV && bytes_per_px == (int)sizeof(T)) {                                                                                      \
        constexpr size_t lds = (size_t)GV * TY * (TK / 2 + 1) * 4;                                                                        \
        static const bool ok_ = lds <= 64 * 1024 || hipFuncSetAttribute(reinterpret_cast<const void*>(k_extract_band<T, GV, CONSEC>),    \
                                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; \
        if (!ok_) (void)hipGetLastError();                                                                                                \
        status = shg::launch(k_extract_band<T, GV, CONSEC>, grid, dim3(64 * BAND_NW), lds, st, ba, "k_extract_band");                     \
    }
    if constexpr (CONSEC) {
        SHG_BAND(uint16_t, 4) SHG_BAND(uint16_t, 5) SHG_BAND(uint16_t, 6) SHG_BAND(uint16_t, 7)
        SHG_BAND(uint8_t, 4) SHG_BAND(uint8_t, 5) SHG_BAND(uint8_t, 6) SHG_BAND(uint8_t, 7)
    } else {
        SHG_BAND(uint16_t, 2) SHG_BAND(uint16_t, 4) SHG_BAND(uint8_t, 2) SHG_BAND(uint8_t, 4)
    }
#undef SHG_BAND
    return status;
}

// fold the 64 slots of every plane: out[s] = {min, max}
struct FoldMinmaxArgs {
    const uint32_t* slots;
    uint32_t* out;
    int n_planes;                    // workgroups beyond these clear `zero` (shg::t_zero_with_fold)
    uint32_t* zero;
    size_t zero_words;
};

__global__ __launch_bounds__(64) void k_fold_minmax(const FoldMinmaxArgs kargs) {
    const uint32_t* __restrict__ slots = kargs.slots;
    uint32_t* __restrict__ out = kargs.out;
    if ((int)blockIdx.x >= kargs.n_planes) {
        uint32_t* __restrict__ z = kargs.zero;
        const size_t stride = (size_t)(gridDim.x - kargs.n_planes) * 64;
        for (size_t i = (size_t)(blockIdx.x - kargs.n_planes) * 64 + threadIdx.x; i < kargs.zero_words; i += stride) z[i] = 0;
        return;
    }
    uint32_t a = slots[((int64_t)blockIdx.x * 64 + threadIdx.x) * 2], b = slots[((int64_t)blockIdx.x * 64 + threadIdx.x) * 2 + 1];
    a = shg::wave_fold_u32(a, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    b = shg::wave_fold_u32(b, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = 0xffffu - a; out[blockIdx.x * 2 + 1] = b; }
}

// k_fold_minmax, and with it the clearing of the area the next stage asked for (shg::t_zero_with_fold)
int launch_fold(uint32_t* minmax_slots, int n_shifts, hipStream_t st) {
    FoldMinmaxArgs fa{minmax_slots, minmax_slots + (int64_t)n_shifts * 64 * 2, n_shifts, nullptr, 0};
    unsigned extra = 0;
    if (shg::t_zero_with_fold && shg::t_zero_with_fold_words > 0) {
        fa.zero = static_cast<uint32_t*>(shg::t_zero_with_fold);
        fa.zero_words = shg::t_zero_with_fold_words;
        extra = (unsigned)std::min<size_t>((fa.zero_words + 1023) / 1024, 64);
    }
    const int e = shg::launch(k_fold_minmax, dim3((unsigned)n_shifts + extra), dim3(64), 0, st, fa, "k_fold_minmax");
    if (e == 0 && extra) shg::t_prezeroed = shg::t_zero_with_fold;
    shg::t_zero_with_fold = nullptr;
    shg::t_zero_with_fold_words = 0;
    return e;
}

}  // namespace

extern "C" int shg_extract_columns(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                                   int bytes_per_px, int64_t frame_stride_px, const int32_t* ind_l, const double* lw, const double* rw,
                                   int n_shifts, uint16_t* disks, int64_t row_pitch, int64_t plane_stride,
                                   int64_t n_cols, int64_t k_offset, int flip_x, shg_stream_t stream) {
    return shg_extract_columns_minmax(stack, n_frames, height, width, bytes_per_px, frame_stride_px, ind_l, lw, rw, n_shifts, disks,
                                      row_pitch, plane_stride, n_cols, k_offset, flip_x, nullptr, stream);
}

extern "C" int shg_extract_columns_minmax(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                                          int bytes_per_px, int64_t frame_stride_px, const int32_t* ind_l, const double* lw, const double* rw,
                                          int n_shifts, uint16_t* disks, int64_t row_pitch, int64_t plane_stride,
                                          int64_t n_cols, int64_t k_offset, int flip_x, uint32_t* minmax_slots, shg_stream_t stream) {
    const bool slots_zeroed = shg::t_minmax_slots_zeroed;        // (first thing: whatever this call returns, the hint is spent)
    shg::t_minmax_slots_zeroed = false;
    SHG_REQUIRE(stack && ind_l && lw && rw && disks, SHG_E_ARG, "shg_extract_columns: null pointer");
    SHG_REQUIRE(n_frames > 0 && height > 0 && width > 0 && n_shifts > 0, SHG_E_ARG, "shg_extract_columns: empty input");
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_extract_columns: bytes_per_px must be 1 or 2");
    SHG_REQUIRE(n_frames < (1ll << 31), SHG_E_UNSUPPORTED, "shg_extract_columns: too many frames");
    SHG_REQUIRE(n_cols >= n_frames && k_offset >= 0 && k_offset + n_frames <= n_cols, SHG_E_ARG,
                "shg_extract_columns: frames [%lld, %lld) do not fit %lld columns", (long long)k_offset,
                (long long)(k_offset + n_frames), (long long)n_cols);
    SHG_REQUIRE(row_pitch >= n_cols, SHG_E_ARG, "shg_extract_columns: row_pitch < n_cols");
    SHG_REQUIRE((height < width ? height : width) >= 2, SHG_E_ARG, "shg_extract_columns: spectral axis needs >= 2 pixels");
    SHG_REQUIRE(frame_stride_px == 0 || frame_stride_px >= height * width, SHG_E_ARG, "shg_extract_columns: frame stride smaller than a frame");
    // (the kernel addresses a sample as `frame base + 32-bit byte offset`)
    SHG_REQUIRE(height * width * bytes_per_px < (1ll << 32), SHG_E_UNSUPPORTED, "shg_extract_columns: a frame of %lld x %lld samples is larger than 4 GiB",
                (long long)height, (long long)width);
    const int64_t fstride = frame_stride_px > 0 ? frame_stride_px : height * width;
    const bool rot = width > height;
    const int64_t ih = rot ? width : height;
    // the tile's 16-byte row segments land on 16-byte boundaries when every row does
    const int vec_store = ((reinterpret_cast<uintptr_t>(disks) & 15) == 0) && (row_pitch % 8 == 0) && (plane_stride % 8 == 0);
    static const int sc_env = [] { const char* e = getenv("SHG_EXT_SC"); return e ? atoi(e) : 0; }();           // tuning override: 2 or 4
    const int sc = n_shifts <= 2 ? 2 : (sc_env == 2 ? 2 : SC_MAX);
    dim3 grid((unsigned)((n_cols + TK - 1) / TK), (unsigned)((ih + TY - 1) / TY), (unsigned)((n_shifts + sc - 1) / sc));
    hipStream_t st = shg::as_stream(stream);
    const int n = (int)n_frames;
    const ExtractArgs xa{stack, n, height, width, fstride, ind_l, lw, rw, n_shifts, disks, row_pitch, plane_stride, n_cols, k_offset, flip_x, vec_store, minmax_slots};
    int launch_status = 0;
#define EXT_LAUNCH_BS(T, ROT, B, SCV) launch_status = shg::launch(k_extract<T, ROT, B, SCV>, grid, dim3(256), 0, st, xa, "k_extract")
#define EXT_LAUNCH_B(T, ROT, B) do { if (sc == 2) EXT_LAUNCH_BS(T, ROT, B, 2); else EXT_LAUNCH_BS(T, ROT, B, 4); } while (0)
#define EXT_LAUNCH(T, ROT)                                                 \
    switch (batch) {                                                       \
        case 1: EXT_LAUNCH_B(T, ROT, 1); break;                            \
        case 2: EXT_LAUNCH_B(T, ROT, 2); break;                            \
        case 8: EXT_LAUNCH_B(T, ROT, 8); break;                            \
        case 16: EXT_LAUNCH_B(T, ROT, 16); break;                          \
        default: EXT_LAUNCH_B(T, ROT, 4); break;                           \
    }
    static const int batch_env = [] { const char* e = getenv("SHG_EXT_BATCH"); return e ? atoi(e) : 0; }();   // tuning override
    // measured at C2 (tools/sweep_extract.sh), batch 1 / 2 / 4 / 8 / 16: S=21 124 / 104 / 104 / 95 / 124 us; S=2 with the two-shift
    // instantiation 14 / 13 / 14 / 14 us for batch 2 / 4 / 8 / 16 (16 us with the four-shift one)
    const int batch2 = 4;
    const int batch = batch_env > 0 ? batch_env : (n_shifts > SC_MAX ? 8 : (n_shifts <= 2 ? batch2 : 4));
    if (minmax_slots && !slots_zeroed) {
        if (hipError_t e = hipMemsetAsync(minmax_slots, 0, (size_t)n_shifts * 64 * 2 * sizeof(uint32_t), st)) {
            shg::set_error("shg_extract_columns: memset: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    // rotated files: the band kernel, groups of two or four shifts in plane order (SHG_EXT_GENERAL=1: the kernel above, for A / B runs)
    const bool general_only = [] { const char* e = getenv("SHG_EXT_GENERAL"); return e && atoi(e) != 0; }();     // (read per call: tools/sweep_band.py)
    if (rot && !general_only) {
        BandArgs ba{stack, n, height, width, fstride, ind_l, nullptr, lw, rw, n_shifts, disks, row_pitch, plane_stride, n_cols, k_offset, flip_x, vec_store,
                    minmax_slots, (int)((n_cols + TK - 1) / TK), {}};
        {
            SHG_PROF("extract", st);
            launch_status = launch_band<false>(ba, bytes_per_px, ih, st);
        }
        if (launch_status) return launch_status;
        if (minmax_slots) return launch_fold(minmax_slots, n_shifts, st);
        return 0;
    }
    {
        SHG_PROF("extract", st);
        if (bytes_per_px == 2) {
            if (rot) EXT_LAUNCH(uint16_t, true) else EXT_LAUNCH(uint16_t, false)
        } else {
            if (rot) EXT_LAUNCH(uint8_t, true) else EXT_LAUNCH(uint8_t, false)
        }
    }
#undef EXT_LAUNCH_B
#undef EXT_LAUNCH_BS
#undef EXT_LAUNCH
    if (launch_status) return launch_status;
    if (minmax_slots) return launch_fold(minmax_slots, n_shifts, st);
    return 0;
}

// The same for a Doppler stack whose shifts are consecutive integers (any order): host_shifts[n_shifts] as the planes are laid
// out; base_col [ih] = the column of the smallest shift, NOT clamped (fit[:, 0] + min shift); ind_l as above (clamped), used for
// the rows near the frame's edge.  3 <= n_shifts <= 24.  Bit-identical to shg_extract_columns_minmax.
extern "C" int shg_extract_dense_fits(const int32_t* host_shifts, int n_shifts) {
    if (!host_shifts || n_shifts < 3 || n_shifts > DS_MAX) return 0;
    int lo = host_shifts[0], hi = host_shifts[0];
    for (int i = 1; i < n_shifts; ++i) { lo = host_shifts[i] < lo ? host_shifts[i] : lo; hi = host_shifts[i] > hi ? host_shifts[i] : hi; }
    if ((int64_t)hi - lo != n_shifts - 1) return 0;
    bool seen[DS_MAX] = {};
    for (int i = 0; i < n_shifts; ++i) {
        if (seen[host_shifts[i] - lo]) return 0;
        seen[host_shifts[i] - lo] = true;
    }
    return 1;
}

extern "C" int shg_extract_columns_dense(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                                         int64_t frame_stride_px, const int32_t* ind_l, const int32_t* base_col, const double* lw,
                                         const double* rw, const int32_t* host_shifts, int n_shifts, uint16_t* disks, int64_t row_pitch,
                                         int64_t plane_stride, int64_t n_cols, int64_t k_offset, int flip_x, uint32_t* minmax_slots,
                                         shg_stream_t stream) {
    const bool slots_zeroed = shg::t_minmax_slots_zeroed;        // (first thing: whatever this call returns, the hint is spent)
    shg::t_minmax_slots_zeroed = false;
    SHG_REQUIRE(stack && ind_l && base_col && lw && rw && host_shifts && disks, SHG_E_ARG, "shg_extract_columns_dense: null pointer");
    SHG_REQUIRE(n_frames > 0 && height > 0 && width > 0, SHG_E_ARG, "shg_extract_columns_dense: empty input");
    SHG_REQUIRE(shg_extract_dense_fits(host_shifts, n_shifts), SHG_E_UNSUPPORTED, "shg_extract_columns_dense: the shifts are not 3..%d consecutive integers", DS_MAX);
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_extract_columns_dense: bytes_per_px must be 1 or 2");
    SHG_REQUIRE(n_frames < (1ll << 31), SHG_E_UNSUPPORTED, "shg_extract_columns_dense: too many frames");
    SHG_REQUIRE(n_cols >= n_frames && k_offset >= 0 && k_offset + n_frames <= n_cols, SHG_E_ARG, "shg_extract_columns_dense: frames do not fit the columns");
    SHG_REQUIRE(row_pitch >= n_cols, SHG_E_ARG, "shg_extract_columns_dense: row_pitch < n_cols");
    SHG_REQUIRE((height < width ? height : width) > n_shifts, SHG_E_UNSUPPORTED, "shg_extract_columns_dense: the spectral axis must be wider than the shift range");
    SHG_REQUIRE(frame_stride_px == 0 || frame_stride_px >= height * width, SHG_E_ARG, "shg_extract_columns_dense: frame stride smaller than a frame");
    const int64_t fstride = frame_stride_px > 0 ? frame_stride_px : height * width;
    const bool rot = width > height;
    const int64_t ih = rot ? width : height;
    const int vec_store = ((reinterpret_cast<uintptr_t>(disks) & 15) == 0) && (row_pitch % 8 == 0) && (plane_stride % 8 == 0);
    int lo = host_shifts[0];
    for (int i = 1; i < n_shifts; ++i) lo = host_shifts[i] < lo ? host_shifts[i] : lo;
    PlaneOfOffset po = {};
    for (int i = 0; i < n_shifts; ++i) po.v[host_shifts[i] - lo] = i;
    hipStream_t st = shg::as_stream(stream);
    if (minmax_slots && !slots_zeroed) {
        if (hipError_t e = hipMemsetAsync(minmax_slots, 0, (size_t)n_shifts * 64 * 2 * sizeof(uint32_t), st)) {
            shg::set_error("shg_extract_columns_dense: memset: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    const int n = (int)n_frames;
    if (rot) {
        // the band kernel: every sample once per group of up to seven shifts
        BandArgs ba{stack, n, height, width, fstride, ind_l, base_col, lw, rw, n_shifts, disks, row_pitch, plane_stride, n_cols, k_offset, flip_x, vec_store,
                    minmax_slots, (int)((n_cols + TK - 1) / TK), {}};
        for (int i = 0; i < n_shifts; ++i) ba.plane_of[host_shifts[i] - lo] = (uint8_t)i;
        int launch_status;
        {
            SHG_PROF("extract", st);
            launch_status = launch_band<true>(ba, bytes_per_px, ih, st);
        }
        if (launch_status) return launch_status;
        if (minmax_slots) return launch_fold(minmax_slots, n_shifts, st);
        return 0;
    }
    // launch shape (tools/bench_extract.py): waves per workgroup x frames per workgroup x frames in flight per wave
    static const int shape = [] { const char* e = getenv("SHG_EXT_DENSE_SHAPE"); return e ? atoi(e) : 0; }();
#define SHG_DENSE(T, ROT, B, NW, DKV)                                                                                                               \
    do {                                                                                                                                            \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(k_extract_dense<T, ROT, B, NW, DKV>),                            \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;                         \
        if (!ok_) (void)hipGetLastError();                                                                                                          \
        const size_t lds = (size_t)n_shifts * TY * (DKV + 2) * sizeof(uint16_t);                                                                    \
        dim3 grid((unsigned)((n_cols + DKV - 1) / DKV), (unsigned)((ih + TY - 1) / TY));                                                            \
        k_extract_dense<T, ROT, B, NW, DKV><<<grid, 64 * NW, lds, st>>>(static_cast<const T*>(stack), n, height, width, fstride, ind_l, base_col, lw, rw, \
                                                                       n_shifts, po, disks, row_pitch, plane_stride, n_cols, k_offset, flip_x, vec_store, minmax_slots); \
    } while (0)
#define SHG_DENSE_S(T, ROT)                                                                  \
    do {                                                                                     \
        switch (shape) {                                                                     \
            case 1: SHG_DENSE(T, ROT, 4, 8, 32); break;                                      \
            case 2: SHG_DENSE(T, ROT, 2, 16, 32); break;                                     \
            case 3: SHG_DENSE(T, ROT, 4, 4, 16); break;                                      \
            case 4: SHG_DENSE(T, ROT, 2, 8, 16); break;                                      \
            case 5: SHG_DENSE(T, ROT, 2, 4, 8); break;                                       \
            case 6: SHG_DENSE(T, ROT, 4, 4, 32); break;                                      \
            default: SHG_DENSE(T, ROT, 4, 4, 16); break;                                     \
        }                                                                                    \
    } while (0)
    {
        SHG_PROF("extract", st);
        if (bytes_per_px == 2) { if (rot) SHG_DENSE_S(uint16_t, true); else SHG_DENSE_S(uint16_t, false); }
        else { if (rot) SHG_DENSE_S(uint8_t, true); else SHG_DENSE_S(uint8_t, false); }
    }
#undef SHG_DENSE_S
#undef SHG_DENSE
    if (int e = shg::check_launch("k_extract_dense")) return e;
    if (minmax_slots) return launch_fold(minmax_slots, n_shifts, st);
    return 0;
}
